"""Average every collected counter per kernel over the passes written by tools/pmc_microbench.sh (diagnostic)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "conv_sf" not in k:
        continue
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    print(k.replace("atdn::", "")[:110])
    wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    for n in sorted(c):
        print("    %-28s %14.0f   /wave_cycles %.3f" % (n, c[n], c[n] / wc))
