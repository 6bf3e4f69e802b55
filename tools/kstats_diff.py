"""Per-step kernel cost from two rocprofv3 kernel_stats.csv of the same script run for n1 and n2 steps:
(total2 - total1) / (n2 - n1) per kernel name -- set-up kernels cancel."""
import csv
import sys

from kstats import short


def load(path):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(path))}


def main(p1, n1, p2, n2, top=70):
    a, b = load(p1), load(p2)
    dn = float(n2) - float(n1)
    rows = []
    for k, (c2, t2) in b.items():
        c1, t1 = a.get(k, (0, 0.0))
        if c2 != c1:
            rows.append(((t2 - t1) / dn / 1e3, (c2 - c1) / dn, short(k)))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    for us, calls, name in rows[:top]:
        print("%9.1f us/step %6.2f calls/step  %s" % (us, calls, name))
    print("total %.3f ms/step, %.1f launches/step, %d kernels" % (tot / 1e3, sum(r[1] for r in rows), len(rows)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4])
