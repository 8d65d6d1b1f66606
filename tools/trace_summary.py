#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid) so that launches can be matched to layers.
usage: python tools/trace_summary.py <bench_kernel_trace.csv> <steps_in_trace> [top_n]"""
import collections
import csv
import sys


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    agg = collections.OrderedDict()
    total = 0.0
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        wg = int(r["Workgroup_Size_X"])
        key = (name, int(r["Grid_Size_X"]) // wg, r["Grid_Size_Y"], r["Grid_Size_Z"])
        d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += d
        total += d
    print("total kernel time per step: %.3f ms" % (total / steps / 1e6))
    for (name, gx, gy, gz), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-40s grid(%6d,%4s,%3s)  n/step %5.1f  avg_us %8.1f  ms/step %6.3f" % (name[:40], gx, gy, gz, n / steps, t / n / 1e3, t / steps / 1e6))


if __name__ == "__main__":
    main()
