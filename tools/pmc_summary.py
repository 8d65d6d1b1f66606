#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 --pmc counter_collection.csv: launches, mean duration, and the mean of every counter;
derives the shader clock (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / duration) and MFMA pipe occupancy
(SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x clock x duration)) when those counters are present.

    python tools/pmc_summary.py gpurun_out/pmcX/*_counter_collection.csv [--match conv_] [--min-us 50]
"""
import argparse
import collections
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--match", default="")
    ap.add_argument("--min-us", type=float, default=0.0)
    ap.add_argument("--per-dispatch", action="store_true", help="one line per dispatch, in dispatch order")
    a = ap.parse_args()
    disp = {}
    for row in csv.DictReader(open(a.csv)):
        name = row["Kernel_Name"]
        if a.match and a.match not in name:
            continue
        d = disp.setdefault(row["Dispatch_Id"], {"name": name.split("(")[0].replace("void ", ""), "grid": int(row["Grid_Size"]),
                                                  "us": (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3, "c": {}})
        d["c"][row["Counter_Name"]] = d["c"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    if a.per_dispatch:
        for k in sorted(disp, key=int):
            d = disp[k]
            if d["us"] >= a.min_us:
                print("%-40s WGs %6d  %8.1f us  %s" % (d["name"][:40], d["grid"] // 256, d["us"],
                                                      "  ".join("%s=%.4g" % kv for kv in sorted(d["c"].items()))))
        return
    groups = collections.OrderedDict()
    for d in disp.values():
        if d["us"] < a.min_us:
            continue
        groups.setdefault((d["name"], d["grid"]), []).append(d)
    for (name, grid), ds in groups.items():
        n = len(ds)
        us = sum(d["us"] for d in ds) / n
        c = {k: sum(d["c"].get(k, 0.0) for d in ds) / n for k in ds[0]["c"]}
        extra = ""
        if "GRBM_GUI_ACTIVE" in c:
            clk = c["GRBM_GUI_ACTIVE"] / 8 / (us * 1e-6) / 1e9
            extra += "  clock %.2f GHz" % clk
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                extra += "  MFMA busy %.0f%%" % (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * clk * 1e9 * us * 1e-6))
        print("%-44s grid %8d  n %3d  %8.1f us%s" % (name[:44], grid, n, us, extra))
        print("      " + "  ".join("%s=%.4g" % kv for kv in sorted(c.items())))


if __name__ == "__main__":
    main()
