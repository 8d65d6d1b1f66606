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
    if name.startswith("colstats_kernel<true>") or name.startswith("colstats_kernel<1"):
        return 2 * 4 * n            # read dz, y
    if name.startswith("colstats_kernel"):
        return 4 * n                # read y
    if name.startswith("colnorm_bwd_fused") or name.startswith("colnorm_bwd_onepass"):
        return 3 * 4 * n
    return None


def main():
    path = sys.argv[1]
    groups = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        wg = int(r["Workgroup_Size_X"])
        key = (name, int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        g = groups.setdefault(key, [0, 0.0])
        g[0] += 1
        g[1] += d
    # match a (kernel, grid) group to the activation it streams by elapsed-time rank within the kernel family: the families'
    # launches come in the layer sizes of ACT, so sort both by size
    rows = []
    fam = collections.defaultdict(list)
    for (name, gx, gy, gz), (n, t) in groups.items():
        fam[name].append(((gx, gy, gz), n, t / n / 1e3))
    for name, lst in fam.items():
        if name.startswith(("colnorm_apply", "colstats", "colnorm_bwd")):
            # 2-D layers: grid.y == B (InstanceNorm groups) -> sizes L1..L7 (L0 is fused into l0_*); sort by avg time descending
            two_d = sorted([x for x in lst if x[0][1] == B], key=lambda x: -x[2])
            sizes = sorted((v for k, v in ACT.items() if k != "L0"), reverse=True)
            # distinct sizes only (L5 == L6)
            uniq = sorted(set(sizes), reverse=True)
            for (grid, n, us), elems in zip(two_d, uniq):
                rows.append((name, grid, n, us, algorithmic_bytes(name, elems)))
            for x in lst:
                if x[0][1] != B:
                    rows.append((name, x[0], x[1], x[2], None))
        elif name.startswith("l0_fwd"):
            for grid, n, us in lst:
                rows.append((name, grid, n, us, 4 * (ACT["L0"] + MEL)))
        elif name.startswith("l0_bwd_sums"):
            for grid, n, us in lst:
                rows.append((name, grid, n, us, 4 * (ACT["L0"] + MEL)))
        elif name.startswith("l0_moments"):
            for grid, n, us in lst:
                rows.append((name, grid, n, us, 4 * MEL))
        elif name.startswith("adam_kernel"):
            for grid, n, us in sorted(lst, key=lambda x: -x[2])[:1]:
                rows.append((name, grid, n, us, 7 * 4 * 7075122))
        elif name.startswith("weight_transpose_batched"):
            for grid, n, us in lst:
                rows.append((name, grid, n, us, 2 * 4 * (7075122 - 848626 + 848384)))
        elif name.startswith("final_metrics_kernel"):
            for grid, n, us in lst:
                rows.append((name, grid, n, us, 2 * B * 64 * 242 * 4))
        elif name.startswith("mel_fb"):
            for grid, n, us in lst:
                rows.append((name, grid, n, us, 4 * (B * 427 * 514 + MEL)))
    print("%-34s %-18s %7s %9s %10s %9s %8s" % ("kernel", "grid", "calls", "avg_us", "alg_MB", "TB/s", "of 8TB/s"))
    for name, grid, n, us, nb in sorted(rows, key=lambda r: -(r[3] * r[2])):
        if nb is None:
            print("%-34s %-18s %7d %9.1f %10s %9s %8s" % (name[:34], str(grid), n, us, "?", "?", "?"))
        else:
            tbs = nb / (us * 1e-6) / 1e12
            print("%-34s %-18s %7d %9.1f %10.1f %9.2f %7.0f%%" % (name[:34], str(grid), n, us, nb / 1e6, tbs, 100 * tbs * 1e12 / HBM_PEAK))


if __name__ == "__main__":
    main()
