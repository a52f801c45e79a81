#!/usr/bin/env python
"""Turn one profiles/collect.sh output directory (gpurun_out/prof_<tag>) into the committed
summaries: kernel-trace table, SQ PMC table, and pmc_traffic.json (HBM bytes per launch per kernel
tag, read by bench.py for roofline.traffic).

HBM bytes follow MI355X_MICROARCH.md "HBM": bytes = FETCH_SIZE*1024*2 (gfx950 rocprofv3 reports
half the bytes of wide coalesced reads) + WRITE_SIZE*1024, FETCH and WRITE from separate passes.
usage: make_summaries.py gpurun_out/prof_r01a r01a"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
AS = {0: "A_ROW", 1: "A_COL", 2: "A_EDGE", 3: "A_EDGE_T"}
BS = {0: "B_ROW", 1: "B_COL"}
EP = {0: "STORE", 1: "SCATTER"}


def tag_of(name):
    """rocprof kernel name -> the tag dgcnn/_engine.py gives the same launch."""
    n = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
    m = re.match(r"gemm_kernel<(\d), (\d), (\d), (\d+), (\d+), (true|false)>", n)
    if m:
        return "gemm_kernel<%s,%s,%s,%s,%s>" % (AS[int(m.group(1))], BS[int(m.group(2))], EP[int(m.group(3))],
                                                  m.group(4), m.group(5))
    m = re.match(r"gemm_x3w2_kernel<(\d), (\d), (\d+)>", n)
    if m:
        kinds = {0: "KCONTIG", 1: "KSTRIDED"}
        return "gemm_x3w2_kernel<%s,%s,bf16x%s>" % (kinds[int(m.group(1))], kinds[int(m.group(2))], m.group(3))
    m = re.match(r"gemm_x3q_kernel<(\d), (\d), (\d+), (\d+)>", n)
    if m:
        kinds = {0: "KCONTIG", 1: "KSTRIDED"}
        return "gemm_x3q_kernel<%s,%s,%s,bf16x%s>" % (kinds[int(m.group(1))], kinds[int(m.group(2))], m.group(4), m.group(3))
    m = re.match(r"gemm_x3_kernel<(\d), (\d), (\d+), (\d+)(?:, \d+)?>", n)
    if m:
        kinds = {0: "KCONTIG", 1: "KSTRIDED"}
        return "gemm_x3_kernel<%s,%s,%s,bf16x%s>" % (kinds[int(m.group(1))], kinds[int(m.group(2))], m.group(3), m.group(4))
    m = re.match(r"gemm_pl_kernel<(\d), (\d), (\d+), (\d+)>", n)
    if m:
        return "gemm_pl_kernel<%s,%s>" % ("KC" if m.group(1) == "0" else "TR", "bf16x3" if m.group(2) == "0" else "f16x2")
    m = re.match(r"knn_(mfma_)?kernel<(\d+), (\d+)(, \w+)?>", n)
    if m:
        return "knn_kernel<C%s,k%s>" % (m.group(2), m.group(3))
    return re.sub(r"<.*", "", n)


def per_launch(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[tag_of(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    one = lambda pat: max(glob.glob(os.path.join(src, pat)), key=os.path.getmtime)   # gpurun MERGES runs into the directory: newest
    steps = 10.0   # collect.sh: --steps 8 --warmup 2
    with open(os.path.join(HERE, "%s_kernel_trace.txt" % tag), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline (1x MI355X; the default launch mode: calibration steps eager, timed steps replayed from the launch plan); per step = per\n"
                "# adam_kernel launch (bench.py also runs 13 EdgeConv-stack-only passes after the steps: their launches are in the totals)\n")
        f.write(subprocess.check_output([sys.executable, os.path.join(HERE, "trace_summary.py"),
                                         one("trace/*/*kernel_trace.csv"), str(steps)]).decode())
    with open(os.path.join(HERE, "%s_kernel_stats.csv" % tag), "w") as f:
        f.write(open(one("trace/*/*kernel_stats.csv")).read())
    with open(os.path.join(HERE, "%s_pmc_sq.txt" % tag), "w") as f:
        f.write("# rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE (own pass) -- python bench.py --steps 2 --warmup 2; mean per dispatch\n")
        f.write(subprocess.check_output([sys.executable, os.path.join(HERE, "pmc_summary.py"),
                                         one("pmc_sq/*/*counter_collection.csv")]).decode())
    fetch = per_launch(one("pmc_fetch/*/*counter_collection.csv"), "FETCH_SIZE")
    write = per_launch(one("pmc_write/*/*counter_collection.csv"), "WRITE_SIZE")
    traffic = {}
    for k in sorted(set(fetch) | set(write)):
        traffic[k] = int(fetch.get(k, 0.0) * 1024 * 2 + write.get(k, 0.0) * 1024)
    json.dump(traffic, open(os.path.join(HERE, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    # mean duration per tag from the (counter-free) kernel-trace pass -> achieved HBM GB/s and its fraction of 8 TB/s
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(one("trace/*/*kernel_trace.csv"))):
        dur[tag_of(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(os.path.join(HERE, "%s_pmc_hbm_bytes.txt" % tag), "w") as f:
        f.write("# HBM bytes per launch (mean over launches of the tag): FETCH_SIZE*1024*2 + WRITE_SIZE*1024, separate --pmc passes;\n"
                "# avg_us from the kernel-trace pass (no counters; kernels of the two streams overlap there) -> GB/s, fraction of the 8 TB/s HBM3E peak\n")
        for k, v in sorted(traffic.items(), key=lambda kv: -kv[1]):
            us = sum(dur[k]) / len(dur[k]) if dur.get(k) else 0.0
            gbs = v / 1e3 / us if us > 0 else 0.0
            f.write("%-48s fetch_KB %12.0f  write_KB %12.0f  hbm_MB %10.1f  avg_us %8.1f  GB/s %7.0f  of_peak %5.3f\n"
                    % (k, fetch.get(k, 0), write.get(k, 0), v / 1e6, us, gbs, gbs / 8000.0))
    # the bench line of the same collect run; its roofline.traffic was read from the PREVIOUS pmc_traffic.json when it printed:
    # replace it with this run's own counter passes so that the committed line and tables describe one build
    b = json.load(open(os.path.join(src, "bench.json")))
    k = b.get("roofline", {}).get("kernel")
    if k in traffic:
        b["roofline"]["traffic"] = traffic[k]
        b["roofline"]["traffic_note"] = "HBM bytes per launch of this tag from the --pmc passes of the same profiles collect run (FETCH_SIZE*1024*2 + WRITE_SIZE*1024)"
    json.dump(b, open(os.path.join(HERE, "%s_bench.json" % tag), "w"), indent=1)


if __name__ == "__main__":
    main()
