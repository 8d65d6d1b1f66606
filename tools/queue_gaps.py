#!/usr/bin/env python
"""Idle gaps between consecutive kernels of each HIP queue in a rocprofv3 --kernel-trace CSV of bench.py: where the main stream
waits (host enqueue, cross-stream events) rather than computes.
usage: python tools/queue_gaps.py <kernel_trace.csv> <steps_in_trace> [top_n]"""
import collections
import csv
import sys


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    q = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        q[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]))
    for qid, ks in sorted(q.items(), key=lambda kv: -len(kv[1])):
        ks.sort()
        # drop everything before the timed region: keep the last `steps` occurrences of the step delimiter (the big Adam launch)
        gaps = collections.defaultdict(lambda: [0, 0.0])
        tot_gap = tot_busy = 0.0
        hist = collections.Counter()
        for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
            g = max(0, s1 - e0) / 1e3
            if g > 2000:  # between steps of the warm-up / region boundaries
                continue
            tot_gap += g
            tot_busy += (e0 - s0) / 1e3
            a = gaps[(n0, n1)]
            a[0] += 1
            a[1] += g
            hist[min(int(g // 2) * 2, 20)] += 1
        n = len(ks)
        print("queue %s: %d launches, kernel time %.3f ms/step, idle between launches %.3f ms/step (%.1f us per launch)" % (
            qid, n, tot_busy / steps / 1e3, tot_gap / steps / 1e3, tot_gap / max(n, 1)))
        print("   gap histogram (us -> launches/step): " + "  ".join("%d-%d: %.1f" % (k, k + 2, v / steps) if k < 20 else ">=20: %.1f" % (v / steps) for k, v in sorted(hist.items())))
        for (n0, n1), (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:top]:
            print("   %-44s -> %-44s n/step %5.1f  avg gap %7.1f us  us/step %7.1f" % (n0, n1, c / steps, g / c, g / steps))


if __name__ == "__main__":
    main()
