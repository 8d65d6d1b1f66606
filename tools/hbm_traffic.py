#!/usr/bin/env python
"""HBM-side traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass,
MI355X_MICROARCH.md: TCC counter budget) of the same command:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o b -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o b -- python bench.py ...
    python tools/hbm_traffic.py gpurun_out/pmc_fetch/b_counter_collection.csv gpurun_out/pmc_write/b_counter_collection.csv out.json "<command>"

Both counters are in KiB.  gfx950 correction (same guide): FETCH_SIZE counts 128-byte requests as 64 bytes on wide coalesced
streams -> traffic = 2 * FETCH_SIZE + WRITE_SIZE.  The counters sit on the fabric side of the L2 (Infinity-Cache hits
included): an upper bound on HBM bytes."""
import collections
import csv
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def per_kernel(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        # the three epilogue instantiations (EPI = plain / + forward statistics / + backward statistics) are one kernel for bench.py
        name = re.sub(r"^(conv_taps_kernel<\d+, \d+, \w+, \d+), \d+>", r"\1>", name)
        # <TX, TY, BM, BN, EPI, workgroups per CU> -> bench.py's name (the bf16 instantiations stay mangled: the demangler does not know DF16b)
        name = re.sub(r"^convsk_kernel<float, float, (\d+, \d+), \d+, \d+>", r"convsk_kernel<\1>", name)
        name = re.sub(r"^_Z13convsk_kernelIDF16bDF16bLi(\d+)ELi(\d+)E.*", r"convsk_kernel<\1, \2> bf16", name)
        # <ET, BM, BN, WGM, WGN, EPI> of the 8-wave kernel (csrc/convbf.hip): float = the split-fp32 form, mangled = bf16 tensors
        name = re.sub(r"^convbf2_kernel<float, (\d+, \d+), \d+, \d+, \d+(?:, (?:true|false))?>", r"convbf2_kernel<float, \1>", name)
        name = re.sub(r"^_Z14convbf2_kernelIDF16bLi(\d+)ELi(\d+)E.*", r"convbf2_kernel<\1, \2>", name)
        tot[name] += float(r["Counter_Value"]) * 1024.0
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[name] += 1
    return tot, cnt


def main():
    fetch_csv, write_csv, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    f, nf = per_kernel(fetch_csv, "FETCH_SIZE")
    w, nw = per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in f:
        if nf[k] == 0 or nw.get(k, 0) == 0:
            continue
        kernels[k] = {"launches": nf[k], "fetch_raw_mb_per_launch": f[k] / nf[k] / 1e6, "fetch_x2_mb_per_launch": 2 * f[k] / nf[k] / 1e6,
                      "write_mb_per_launch": w[k] / nw[k] / 1e6}
    total = {k: (v["fetch_x2_mb_per_launch"] + v["write_mb_per_launch"]) * v["launches"] for k, v in kernels.items()}
    order = sorted(kernels, key=lambda k: -total[k])
    want = sys.argv[5] if len(sys.argv) > 5 else None  # bench.py's roofline.kernel: the traffic of THAT kernel is what the bench line cites
    dom = want if want in kernels else next(k for k in order if k.startswith(("convbf2_kernel<float", "convsk_kernel", "conv_taps_kernel")))
    from bench import kernel_source_digest
    try:
        head = subprocess.run(["git", "-C", REPO, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        head = None
    if head is None and os.path.exists(os.path.join(REPO, ".git_head")):  # the GPU box gets a snapshot without .git/
        head = open(os.path.join(REPO, ".git_head")).read().strip()
    res = {"command": cmd,
           # bench.py cites this file only while the kernel sources still hash to this digest
           "kernel_source_digest": kernel_source_digest(), "git_head": head,
           "note": "TCC fabric-side counters (include Infinity-Cache hits): upper bound on HBM bytes. FETCH_SIZE under-reports wide "
                   "coalesced streams by 2x on gfx950 (MI355X_MICROARCH.md): traffic = 2*FETCH + WRITE.",
           "dominant_kernel": dom,
           "traffic_mb_per_launch": kernels[dom]["fetch_x2_mb_per_launch"] + kernels[dom]["write_mb_per_launch"],
           "kernels": {k: kernels[k] for k in order[:12]}}
    json.dump(res, open(out, "w"), indent=1)
    print(dom, res["traffic_mb_per_launch"])


if __name__ == "__main__":
    main()
