#!/usr/bin/env python
"""Timeline of ONE training step from a rocprofv3 --kernel-trace csv: per kernel start / duration / queue, the idle gap before
it on its own queue, and how much of the step the GPU spent with nothing running.  Steps are delimited by adam_kernel.
usage: step_timeline.py <kernel_trace.csv> [step_index_from_end=1] [--all]"""
import csv
import sys


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    n = n.replace("void at::native::", "at::")
    return n.split("(")[0][:58]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"])))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if r[3].startswith("adam_kernel")]
    back = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 1
    lo, hi = ends[-back - 1] + 1, ends[-back] + 1
    step = rows[lo:hi]
    t0 = step[0][0]
    wall = (step[-1][1] - t0) / 1e3
    # union of busy intervals
    busy, cur_s, cur_e = 0.0, step[0][0], step[0][1]
    for s, e, q, n in step[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    busy /= 1e3
    print("# step: %d kernels, wall %.1f us, GPU busy (union) %.1f us, idle %.1f us (%.1f%%), sum of kernel time %.1f us"
          % (len(step), wall, busy, wall - busy, 100 * (wall - busy) / wall, sum(e - s for s, e, _, _ in step) / 1e3))
    last_end = {}
    prev_end_any = t0
    small = 0.0
    for s, e, q, n in step:
        gap_q = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        gap_any = (s - prev_end_any) / 1e3
        if "--all" in sys.argv or (e - s) / 1e3 < 12 or gap_any > 3:
            print("%9.1f  q%-2d %-58s %8.1f us   gap(queue) %6.1f  idle-before %6.1f" % ((s - t0) / 1e3, q, n, (e - s) / 1e3, gap_q, max(gap_any, 0.0)))
        last_end[q] = e
        prev_end_any = max(prev_end_any, e)
        if (e - s) / 1e3 < 12:
            small += (e - s) / 1e3
    print("# kernels shorter than 12 us: total %.1f us" % small)


if __name__ == "__main__":
    main()
