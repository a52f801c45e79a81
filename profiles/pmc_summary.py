#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch).
Derived columns (when the pass holds GRBM_GUI_ACTIVE / SQ_VALU_MFMA_BUSY_CYCLES):
  clk_GHz  = GRBM_GUI_ACTIVE / 8 XCDs / avg_us          (effective shader clock under the power cap)
  mfma_%   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x avg_us x clk)   (share of SIMD-cycles with the matrix pipe busy)
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
    print("kernel | calls | avg_us | clk_GHz | mfma_% | " + " | ".join(ctrs) + " | regs")
    for tot, k, n, dur, v in rows[:30]:
        vals = ["%.4g" % (sum(v[c]) / len(v[c])) if c in v else "-" for c in ctrs]
        mean = lambda c: sum(v[c]) / len(v[c])
        clk = mean("GRBM_GUI_ACTIVE") / 8.0 / dur / 1e3 if "GRBM_GUI_ACTIVE" in v and dur > 0 else 0.0
        busy = 100.0 * mean("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * dur * 1e3 * clk) if ("SQ_VALU_MFMA_BUSY_CYCLES" in v and clk > 0) else 0.0
        print("%-58s %3d %8.1f  %5.2f  %5.1f  " % (k[:58], n, dur, clk, busy) + "  ".join(vals) + "  " + v["_regs"][0])


if __name__ == "__main__":
    main()
