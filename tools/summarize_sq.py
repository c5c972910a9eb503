"""Per-kernel averages of the SQ counters collected by tools/sq_counters.sh or tools/pmc_microbench.sh (one rocprofv3
--pmc pass per counter group, merged here by kernel name).

    python tools/summarize_sq.py <dir with p*/ passes> [--csv out.csv]

Derived columns (MI355X_MICROARCH.md units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave;
SQ_VALU_MFMA_BUSY_CYCLES = cycles the matrix pipes are busy = 32 per 32x32x16 f16 MFMA; the SQ counters of this
pool sample a subset of the chip, so only ratios between them are meaningful):
  mfma_busy_per_wave_cycle = MFMA_BUSY / (4 * WAVE_CYCLES): share of a resident wave's lifetime spent issuing MFMAs
  wait_any, wait_inst      = SQ_WAIT_ANY, SQ_WAIT_INST_ANY / WAVE_CYCLES
  lds_conflict             = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  valu_per_mfma            = SQ_INSTS_VALU / SQ_INSTS_MFMA
"""
import collections
import csv
import glob
import sys


def main():
    root = sys.argv[1]
    out_csv = sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for k in acc:
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        launches = max(len(v) for v in acc[k].values())
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        rows.append({
            "kernel": k.replace("atdn::", ""), "launches": launches, "wave_cycles_total": wc * launches,
            "mfma_busy_per_wave_cycle": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * wc),
            "wait_any": c.get("SQ_WAIT_ANY", 0.0) / wc, "wait_inst": c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
            "active_inst": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
            "lds_conflict": c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0.0)),
            "valu_per_mfma": c.get("SQ_INSTS_VALU", 0.0) / max(1.0, c.get("SQ_INSTS_MFMA", 0.0)),
        })
    rows.sort(key=lambda r: -r["wave_cycles_total"])
    cols = ["kernel", "launches", "mfma_busy_per_wave_cycle", "wait_any", "wait_inst", "active_inst", "lds_conflict",
            "valu_per_mfma"]
    lines = [",".join(cols)]
    for r in rows[:24]:
        lines.append('"%s",%d,%.3f,%.3f,%.3f,%.3f,%.3f,%.1f' % tuple(r[c] for c in cols))
    text = "\n".join(lines) + "\n"
    if out_csv:
        open(out_csv, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
