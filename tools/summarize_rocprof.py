"""Condense a rocprofv3 --kernel-trace --stats CSV (…_kernel_stats.csv) into the table kept under profiles/."""
import csv
import sys


def main(src, dst):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w") as f:
        f.write("kernel,calls,avg_us,min_us,max_us,total_ms,percent\n")
        for r in rows:
            name = r["Name"].replace("atdn::", "").replace('"', "'")
            f.write('"%s",%s,%.2f,%.2f,%.2f,%.3f,%s\n' % (name, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                       float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                       float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
