#!/usr/bin/env python
"""One training step of a rocprofv3 --kernel-trace run as a TIMELINE: every kernel in start order with its queue, start offset,
duration and the gap since the previous kernel on the SAME queue ended; then, per queue, busy time and idle time inside the step.
The step is delimited by consecutive adam_kernel launches (one per optimizer step); the one in the middle of the run is taken.

usage: timeline.py <dir with *_kernel_trace.csv> [step index from the end, default: middle]   (output: text on stdout)"""
import csv
import glob
import os
import re
import sys


def short(name):
    n = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:64]


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if r[3].startswith("adam_kernel")]
    if len(adam) < 3:
        raise SystemExit("fewer than 3 optimizer steps in the trace")
    which = int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) // 2
    lo, hi = adam[which - 1] + 1, adam[which] + 1
    step = rows[lo:hi]
    t0 = rows[adam[which - 1]][1]                       # end of the previous step's Adam
    tend = step[-1][1]
    print("# step %d of %d: %d kernels, %.3f ms from the previous Adam's end to this Adam's end" % (which, len(adam), len(step), (tend - t0) / 1e6))
    last_end = {}
    busy = {}
    qnames = {}
    print("# %9s %8s %7s  q  kernel" % ("start_us", "dur_us", "gap_us"))
    for s, e, q, n in step:
        qi = qnames.setdefault(q, len(qnames))
        gap = (s - last_end[q]) / 1e3 if q in last_end else (s - t0) / 1e3
        print("  %9.1f %8.1f %7.1f  %d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, qi, n))
        last_end[q] = e
        busy[q] = busy.get(q, 0) + (e - s)
    for q, qi in qnames.items():
        print("# queue %d: busy %.3f ms of %.3f" % (qi, busy[q] / 1e6, (tend - t0) / 1e6))
    # union of busy intervals over all queues: time with NO kernel running anywhere
    iv = sorted((s, e) for s, e, _, _ in step)
    cur_s, cur_e, tot = iv[0][0], iv[0][1], 0
    for s, e in iv[1:]:
        if s > cur_e:
            tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    tot += cur_e - cur_s
    print("# some kernel running: %.3f ms; nothing running: %.3f ms" % (tot / 1e6, (tend - t0 - tot) / 1e6))


if __name__ == "__main__":
    main()
