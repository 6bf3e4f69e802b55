"""Condense a rocprofv3 *_kernel_stats.csv: name (template arguments and parameter list dropped), calls, avg us, total ms."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.replace("at::native::", "").replace("(anonymous namespace)::", "")
    return name[:88]


def main(path, top=40, div=1):
    rows = list(csv.DictReader(open(path)))
    for r in rows[:top]:
        print("%-88s %5d  %9.1f us  %8.3f ms/step" % (short(r["Name"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                                                      float(r["TotalDurationNs"]) / 1e6 / div))
    print("total %.3f ms/step over %d kernels" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / div, len(rows)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40, float(sys.argv[3]) if len(sys.argv) > 3 else 1)
