#!/usr/bin/env python
"""Per-queue (HIP stream) busy time and the top kernels of each queue from a rocprofv3 --kernel-trace CSV of bench.py.
usage: python tools/stream_summary.py <kernel_trace.csv> <steps_in_trace> [top_n]"""
import collections
import csv
import sys


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    q = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    spans = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        a = q[r["Queue_Id"]][name]
        a[0] += 1
        a[1] += e - s
        spans[r["Queue_Id"]].append((s, e))
    allspans = sorted(x for v in spans.values() for x in v)
    busy, cs, ce = 0, None, None
    for s, e in allspans:
        if ce is None or s > ce:
            if ce is not None:
                busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print("wall (first start .. last end) %.3f ms/step, GPU busy (union of all queues) %.3f ms/step" % (
        (allspans[-1][1] - allspans[0][0]) / steps / 1e6, busy / steps / 1e6))
    for qid, ks in sorted(q.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
        tot = sum(v[1] for v in ks.values())
        print("queue %s: %.3f ms/step of kernel time, %.1f launches/step" % (qid, tot / steps / 1e6, sum(v[0] for v in ks.values()) / steps))
        for name, (n, t) in sorted(ks.items(), key=lambda kv: -kv[1][1])[:top]:
            print("    %-44s n/step %5.1f  avg_us %8.1f  ms/step %6.3f" % (name[:44], n / steps, t / n / 1e3, t / steps / 1e6))


if __name__ == "__main__":
    main()
