"""Effective shader clock per kernel from one rocprofv3 pass (MI355X_MICROARCH.md, "DVFS give-back"):
    rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -- python3 bench.py --streams 1 ...
    python tools/summarize_clock.py DIR/**/counter_collection.csv out.csv
clock ~= GRBM_GUI_ACTIVE / 8 XCDs / dispatch wall time (reads high on dispatches shorter than ~0.3 ms)."""
import collections
import csv
import sys


def main(path, out):
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        dt = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        a = acc[r["Kernel_Name"].replace("atdn::", "").replace("(anonymous namespace)::", "")]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += dt
    rows = sorted(acc.items(), key=lambda kv: -kv[1][2])
    with open(out, "w") as fh:
        fh.write("kernel,launches,avg_us,effective_clock_GHz\n")
        for k, (n, cyc, ns) in rows:
            fh.write('"%s",%d,%.1f,%.3f\n' % (k, n, ns / n / 1e3, cyc / 8.0 / ns))
    for k, (n, cyc, ns) in rows[:24]:
        print("%8.1f us x %5d  %.3f GHz  %s" % (ns / n / 1e3, n, cyc / 8.0 / ns, k[:120]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
