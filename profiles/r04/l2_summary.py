#!/usr/bin/env python
"""L2 side of the gather-bound passes from the `--pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum` pass of
profiles/r04/collect.sh: requests the CUs' vector caches (TCP) send to the XCD L2s (TCC) per launch, the bytes they stand for,
the rate over the kernel's duration in that pass and the fraction of the 34.5 TB/s aggregate L2 figure of MI355X_MICROARCH.md.
Request size: 128 B (calibrated on bn_act_kreduce<edge> at F = 128: 4.31e6 requests for 4 (R k F + R F) = 528 MB of gathered rows).
usage: l2_summary.py <counter_collection.csv>"""
import collections
import csv
import sys

KEEP = ("bn_act_kreduce_kernel<4, true>", "bn_bwd_apply_kernel<4, true>", "edge_bwd_apply_wgrad_kernel", "edge_gather_add_kernel",
        "csr_gather_sum_kernel", "edge_mlp_bf16_kernel", "bn1_act_kernel", "bn1_bwd_kernel<true>")
d = collections.defaultdict(lambda: collections.defaultdict(list))
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    if not name.startswith(KEEP):
        continue
    key = name + " g" + r["Grid_Size"]
    d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        d[key]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-46s %5s %8s %12s %9s %9s %8s %8s" % ("kernel (grid)", "calls", "avg_us", "TCP->TCC req", "MB(128B)", "GB/s", "of 34.5T", "L2 hit"))
rows = []
for k, v in d.items():
    m = lambda c: sum(v[c]) / len(v[c])
    us, req = m("us"), m("TCP_TCC_READ_REQ_sum")
    mb = req * 128 / 1e6
    rows.append((us * len(v["us"]), "%-46s %5d %8.1f %12.3e %9.1f %9.0f %8.3f %8.2f" % (
        k, len(v["us"]), us, req, mb, mb / us * 1e3 if us else 0, mb / us * 1e3 / 34500 if us else 0,
        m("TCC_HIT_sum") / max(m("TCC_HIT_sum") + m("TCC_MISS_sum"), 1))))
for _, line in sorted(rows, reverse=True):
    print(line)
