"""Timeline of ONE training step from a rocprofv3 kernel trace (csv): kernels in start order with start offset, duration and
the idle gap before each, plus totals (busy / idle).  The step is delimited by occurrences of k_train_march.

    python tools/step_timeline.py <kernel_trace.csv> [--step 10]"""
import argparse
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--step", type=int, default=10)
    args = ap.parse_args()
    rows = list(csv.DictReader(open(args.trace)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
    marks = [i for i, e in enumerate(ev) if "k_train_march" in e[2]]
    a, b = marks[args.step], marks[args.step + 1]
    t0 = ev[a][0]
    busy_end = t0
    busy = idle = 0
    for s, e, n in ev[a:b]:
        gap = s - busy_end
        if gap > 0:
            idle += gap
        busy += max(0, e - max(s, busy_end))
        print("%9.1f us  +%7.1f us  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, n[:100]))
        busy_end = max(busy_end, e)
    print("step %.1f us: busy %.1f us, idle %.1f us, %d kernels" % ((ev[b][0] - t0) / 1e3, busy / 1e3, idle / 1e3, b - a))


if __name__ == "__main__":
    main()
