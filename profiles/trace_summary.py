#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 --kernel-trace --output-format csv run.
usage: trace_summary.py <kernel_trace.csv> <steps_in_run>"""
import collections
import csv
import sys


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        name = name.split("(")[0]
        d[name][0] += 1
        d[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "adam_kernel" in d:            # one optimizer step per training step: the run's real step count (warm-up, timed,
        steps = float(d["adam_kernel"][0])  # instrumented passes); the EdgeConv-stack-only passes of bench.py add launches without one
    tot = sum(v[1] for v in d.values())
    print("# total kernel time per step %.3f ms (%d steps in the run)" % (tot / steps / 1e3, steps))
    for k, v in sorted(d.items(), key=lambda kv: -kv[1][1]):
        print("%-62s calls/step %5.1f  avg_us %9.1f  ms/step %6.3f  %5.1f%%" % (
            k[:62], v[0] / steps, v[1] / v[0], v[1] / steps / 1e3, 100 * v[1] / tot))


if __name__ == "__main__":
    main()
