#!/usr/bin/env python
"""Per-kernel "algorithmic bytes / microseconds / fraction of 8 TB/s" table for the HBM-bound kernels of the train step, from a
rocprofv3 --kernel-trace CSV of `python bench.py ...`.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o b -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-events --no-overlap-dw
    python tools/hbm_kernels.py gpurun_out/trace/b_kernel_trace.csv > profiles/r02_hbm_kernels.txt

Algorithmic bytes are derived from the launch geometry of each (kernel, grid) group at the bench workload (32 clips, sdt_bp):
the tensor each launch streams is identified by its grid, and the bytes are what the operator has to move once (reads + writes of
the fp32 activations; statistics and parameters are negligible).  A kernel group the table does not know is listed with "?"."""
import collections
import csv
import sys

HBM_PEAK = 8.0e12
B = 32
# activation sizes (floats) of the audio-encoder outputs at B=32: name -> B*H*W*C
ACT = collections.OrderedDict([
    ("L0", B * 80 * 427 * 64), ("L1", B * 40 * 213 * 64), ("L2", B * 40 * 213 * 128), ("L3", B * 20 * 106 * 128),
    ("L4", B * 20 * 106 * 256), ("L5", B * 10 * 53 * 256), ("L6", B * 10 * 53 * 256), ("L7", B * 5 * 51 * 256)])
MEL = B * 80 * 427
G_PARAMS = 7075122 + 32 * 4096  # generator + clip-code table (two Adam launches; the big one dominates)


def algorithmic_bytes(name, rows_elems):
    """bytes moved once by the operator, given the number of fp32 activation elements the launch covers"""
    n = rows_elems
    if name.startswith("colnorm_apply_fwd"):
        return 2 * 4 * n            # read y, write z
    if name.startswith("colnorm_apply_bwd"):
        return 3 * 4 * n            # read dz, y; write dy
    if name.startswith("colstats_kernel<true") or name.startswith("colstats_kernel<1"):
        return 2 * 4 * n            # read dz, y
    if name.startswith("colstats_kernel"):
        return 4 * n                # read y
    if name.startswith("colnorm_bwd_fused") or name.startswith("colnorm_bwd_onepass"):
        return 3 * 4 * n
    return None


UNR = 4  # csrc/norm.hip


def rows_per_block(C, G, R, stats):
    """mirror of colnorm_rows_per_block (csrc/norm.hip): the grid identifies the layer(s) a launch group belongs to"""
    rpp = 256 // (C >> 2)
    target = 512 if (stats and G == 1) else 2048
    want = -(-R // max(1, target // G))
    unit = UNR * rpp
    rpb = -(-want // unit) * unit
    return min(max(rpb, unit), rpp * 64)


# 2-D layers that own a normalisation launch (L0 is fused into l0_*): name -> (rows per clip, C)
NORM2D = collections.OrderedDict([("L1", (40 * 213, 64)), ("L2", (40 * 213, 128)), ("L3", (20 * 106, 128)), ("L4", (20 * 106, 256)),
                                  ("L5", (10 * 53, 256)), ("L6", (10 * 53, 256)), ("L7", (5 * 51, 256))])


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    groups = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        wg = int(r["Workgroup_Size_X"])
        key = (name, int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        g = groups.setdefault(key, [0, 0.0])
        g[0] += 1
        g[1] += d
    if steps is None:  # one Adam launch of the generator group per step
        steps = float(max(n for (name, gx, _, _), (n, _) in groups.items() if name.startswith("adam_kernel")))
    # layers by the grid their normalisation launches use: several layers can share a grid, so a group's bytes are the SUM over
    # its layers and the rate is (bytes per step) / (time per step) of the whole group
    by_grid = collections.defaultdict(list)
    for lname, (R, C) in NORM2D.items():
        by_grid[(-(-R // rows_per_block(C, B, R, False)), B)].append(lname)
    rows, small = [], collections.defaultdict(lambda: [0, 0.0])
    for (name, gx, gy, gz), (n, t) in groups.items():
        per_step_us = t / steps / 1e3
        calls = n / steps
        fam2d = name.startswith(("colnorm_apply", "colstats", "colnorm_bwd"))
        if fam2d and gy == B and (gx, gy) in by_grid:
            layers = by_grid[(gx, gy)]
            if abs(calls - len(layers)) > 1e-6 and abs(calls - 1) < 1e-6:
                layers = layers[-1:]  # a group that only the last layer of the chain launches (L7: no fused statistics)
            ok = abs(calls - len(layers)) < 1e-6
            nb = sum(algorithmic_bytes(name, B * NORM2D[l][0] * NORM2D[l][1]) for l in layers) if ok else None
            rows.append((name, (gx, gy, gz), calls, per_step_us, nb, "+".join(layers) if ok else "?"))
        elif fam2d or name.startswith(("rownorm_kernel", "splitk_reduce", "upsample_add", "c1d_")):
            s_ = small[name.split("<")[0]]
            s_[0] += calls
            s_[1] += per_step_us
        elif name.startswith("l0_fwd"):
            rows.append((name, (gx, gy, gz), calls, per_step_us, 4 * (ACT["L0"] + MEL), "L0 fwd: mel in, z out"))
        elif name.startswith("l0_bwd_sums"):
            rows.append((name, (gx, gy, gz), calls, per_step_us, 4 * (ACT["L0"] + MEL), "L0 bwd: dz + mel in"))
        elif name.startswith("l0_moments"):
            rows.append((name, (gx, gy, gz), calls, per_step_us, 4 * MEL, "mel moments (latency-bound)"))
        elif name.startswith("adam_kernel") and gx >= 1024:
            rows.append((name, (gx, gy, gz), calls, per_step_us, 7 * 4 * 7075122, "generator group: p,g,m,v in; p,m,v out"))
        elif name.startswith("weight_transpose_batched"):
            rows.append((name, (gx, gy, gz), calls, per_step_us, 2 * 4 * (7075122 - 848626 + 848384), "all mirrors, one launch"))
        elif name.startswith("final_metrics_kernel"):
            rows.append((name, (gx, gy, gz), calls, per_step_us, 2 * B * 64 * 242 * 4, "f64 metrics (latency-bound)"))
        elif name.startswith("mel_fb"):
            rows.append((name, (gx, gy, gz), calls, per_step_us, 4 * (B * 427 * 514 + MEL), "power spectrum in, log-mel out"))
    print("HBM-bound kernels of one train step (B=32, voice2pose_sdt_bp): algorithmic bytes per step / time per step, %d steps in the trace" % steps)
    print("%-30s %-14s %6s %9s %9s %7s %8s  %s" % ("kernel", "grid", "calls", "us/step", "alg_MB", "TB/s", "of 8TB/s", "tensors"))
    for name, grid, calls, us, nb, what in sorted(rows, key=lambda r: -r[3]):
        if nb is None:
            print("%-30s %-14s %6.1f %9.1f %9s %7s %8s  %s" % (name[:30], str(grid), calls, us, "?", "?", "?", what))
        else:
            tbs = nb / (us * 1e-6) / 1e12
            print("%-30s %-14s %6.1f %9.1f %9.1f %7.2f %7.0f%%  %s" % (name[:30], str(grid), calls, us, nb / 1e6, tbs, 100 * tbs * 1e12 / HBM_PEAK, what))
    print()
    print("latency-bound 1-D / small launches (<= 3 MB each, a launch is ~4 us whatever it moves), per step:")
    for name, (calls, us) in sorted(small.items(), key=lambda kv: -kv[1][1]):
        print("  %-34s %6.1f launches %8.1f us  (%.1f us each)" % (name, calls, us, us / max(calls, 1e-9)))


if __name__ == "__main__":
    main()
