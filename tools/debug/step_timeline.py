#!/usr/bin/env python
"""One train step of a rocprofv3 kernel trace as a timeline: every main-queue gap above a threshold with the kernels around it and
what the other queue was doing meanwhile.  usage: python tools/debug/step_timeline.py <kernel_trace.csv> [gap_us=40]"""
import csv
import sys

path = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "")[:50])
        for r in csv.DictReader(open(path))]
rows.sort()
adam = [i for i, r in enumerate(rows) if r[3].startswith("adam_kernel")]
# last full step: between the 4th-last and 2nd-last adam launches (two adam launches per step)
a, b = adam[-5], adam[-3]
seg = rows[a + 1:b + 1]
t0 = seg[0][0]
main = max(set(r[2] for r in seg), key=lambda q: sum(1 for r in seg if r[2] == q))
mq = [r for r in seg if r[2] == main]
print("step: %.3f ms, %d kernels on the main queue %s, %d on others" % ((seg[-1][1] - t0) / 1e6, len(mq), main, len(seg) - len(mq)))
for (s0, e0, _, n0), (s1, e1, _, n1) in zip(mq, mq[1:]):
    g = (s1 - e0) / 1e3
    if g >= thr:
        other = [r for r in seg if r[2] != main and r[1] > e0 and r[0] < s1]
        busy = sum(min(r[1], s1) - max(r[0], e0) for r in other) / 1e3
        print("  t=%8.1f us  gap %7.1f us after %-40s before %-40s | other queues busy %6.1f us: %s" % (
            (e0 - t0) / 1e3, g, n0, n1, busy, ", ".join(sorted(set(r[3][:28] for r in other)))[:120]))
