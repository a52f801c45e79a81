#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch).
usage: pmc_summary.py <counter_collection.csv> [more.csv ...]"""
import collections
import csv
import sys


def load(paths):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in paths:
        seen = set()
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
            key = name.split("(")[0] + " g" + r["Grid_Size"]
            d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if (path, r["Dispatch_Id"]) not in seen:
                seen.add((path, r["Dispatch_Id"]))
                d[key]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            d[key]["_regs"] = [r["VGPR_Count"] + "+" + r["Accum_VGPR_Count"] + " lds" + r["LDS_Block_Size"]]
    return d


def main():
    d = load(sys.argv[1:])
    rows = []
    for k, v in d.items():
        dur = sum(v["_dur_us"]) / len(v["_dur_us"])
        rows.append((sum(v["_dur_us"]), k, len(v["_dur_us"]), dur, v))
    rows.sort(reverse=True, key=lambda r: r[0])
    ctrs = sorted({c for _, _, _, _, v in rows for c in v if not c.startswith("_")})
    print("kernel | calls | avg_us | " + " | ".join(ctrs) + " | regs")
    for tot, k, n, dur, v in rows[:30]:
        vals = ["%.4g" % (sum(v[c]) / len(v[c])) if c in v else "-" for c in ctrs]
        print("%-58s %3d %8.1f  " % (k[:58], n, dur) + "  ".join(vals) + "  " + v["_regs"][0])


if __name__ == "__main__":
    main()
