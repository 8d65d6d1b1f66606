#!/usr/bin/env python
"""Trace-based roofline fraction of one kernel family: mean launch duration from a rocprofv3 --kernel-trace CSV against the algorithmic
FLOPs per launch that bench.py reports for it (roofline.algorithmic_gflop_per_launch).  Written next to the round's trace summaries so that
bench.py can cite it beside its event-based figure (roofline.trace_based), guarded by the kernel-source digest.
usage: python tools/trace_fraction.py <kernel_trace.csv> <name regex> <bench kernel name> <gflop per launch> <peak TFLOP/s> <out.json>"""
import csv
import json
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    path, rx, kname, gflop, peak, out = sys.argv[1], re.compile(sys.argv[2]), sys.argv[3], float(sys.argv[4]), float(sys.argv[5]), sys.argv[6]
    n, tot = 0, 0.0
    for r in csv.DictReader(open(path)):
        if rx.search(r["Kernel_Name"]):
            n += 1
            tot += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    assert n > 0, "no launch matches %r" % sys.argv[2]
    import bench
    avg_us = tot / n / 1e3
    try:
        head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__))).decode().strip()
    except Exception:
        head = None
    d = {"kernel": kname, "name_regex": sys.argv[2], "launches": n, "avg_launch_us": avg_us, "algorithmic_gflop_per_launch": gflop,
         "achieved_tflops": gflop * 1e9 / (avg_us * 1e-6) / 1e12, "peak_tflops": peak,
         "frac": gflop * 1e9 / (avg_us * 1e-6) / 1e12 / peak, "trace": os.path.basename(path), "kernel_source_digest": bench.kernel_source_digest(), "git_head": head}
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d))


if __name__ == "__main__":
    main()
