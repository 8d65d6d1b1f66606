#!/usr/bin/env python
"""The kernels of ONE train step in start order, per queue, from a rocprofv3 --kernel-trace CSV of bench.py: what runs between which kernels, and the
gaps.  usage: python tools/debug/step_sequence.py <kernel_trace.csv> [step_index_from_end=3] [anchor=mel_fb_kernel]"""
import csv
import sys


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    anchor = sys.argv[3] if len(sys.argv) > 3 else "mel_fb_kernel"
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", ""),
                     r.get("Grid_Size", ""), r.get("Workgroup_Size", "")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if anchor in r[3]]
    a, b = marks[-back - 1], marks[-back]
    t0 = rows[a][0]
    main_q = rows[a][2]
    prev_end = {}
    print("step of %.1f us from %s to the next one; main queue %s" % ((rows[b][0] - t0) / 1e3, anchor, main_q))
    for s, e, q, name, grid, wg in rows[a:b]:
        gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
        prev_end[q] = e
        print("%9.1f %8.1f %s gap %6.1f  %-60s grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, "M" if q == main_q else "s" + q[-1], gap, name[:60], grid))


if __name__ == "__main__":
    main()
