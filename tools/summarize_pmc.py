"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as
MI355X_MICROARCH.md prescribes). Units: the counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B read
requests at 64 B for 16-B-per-lane loads, so the read side is doubled (guide §HBM); WRITE_SIZE is exact.

    python tools/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out_prefix> [batch]

`batch` (pairs per launch of the profiled run, default bench.DEFAULT_BATCH) is recorded as `_batch`: bench.py only
quotes a traffic figure measured at the batch it runs.
"""
import collections
import csv
import json
import sys


def load(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


def _default_batch():
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.DEFAULT_BATCH


def main(fetch_csv, write_csv, prefix, batch=None):
    batch = _default_batch() if batch is None else batch
    f, w = load(fetch_csv), load(write_csv)
    out = {}
    for k in sorted(set(f) | set(w)):
        fa = sum(f.get(k, [0.0])) / max(1, len(f.get(k, [])))
        wa = sum(w.get(k, [0.0])) / max(1, len(w.get(k, [])))
        # (a kernel launched at several sizes — the correlation kernel runs once per pyramid level — also gets its LARGEST launch)
        fm, wm = max(f.get(k, [0.0])), max(w.get(k, [0.0]))
        out[k.replace("atdn::", "")] = {"launches": len(f.get(k, [])), "fetch_size_kib_avg": fa, "write_size_kib_avg": wa,
                                        "hbm_bytes_per_launch": (2.0 * fa + wa) * 1024.0,
                                        "hbm_bytes_largest_launch": (2.0 * fm + wm) * 1024.0}
    js = dict(out)
    js["_batch"] = int(batch)
    json.dump(js, open(prefix + ".json", "w"), indent=1, sort_keys=True)
    with open(prefix + ".csv", "w") as fh:
        fh.write("kernel,launches,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg,hbm_MB_per_launch(2xFETCH+WRITE)\n")
        for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
            fh.write('"%s",%d,%.1f,%.1f,%.2f\n' % (k, v["launches"], v["fetch_size_kib_avg"], v["write_size_kib_avg"],
                                                  v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:5])
