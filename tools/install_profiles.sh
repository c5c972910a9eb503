#!/bin/bash
# Copy a profile set produced by tools/profile_round.sh + tools/sq_counters.sh from gpurun_out/ into profiles/ under its tag and
# print the figures the documents quote.   usage: bash tools/install_profiles.sh r05_v8
set -e
t=$1; o=gpurun_out/$t
cp $o/bench_default.json profiles/${t}_bench_default.json
cp $o/bench_under_rocprof.json profiles/${t}_bench_under_rocprof.json
cp $o/bench_under_rocprof_1stream.json profiles/${t}_bench_under_rocprof_1stream.json
cp $o/trace2_kernel_stats.csv profiles/${t}_kernel_stats.csv
cp $o/trace1_kernel_stats.csv profiles/${t}_kernel_stats_1stream.csv
cp $o/pmc.csv profiles/${t}_pmc.csv
cp $o/pmc.json profiles/${t}_pmc.json
cp gpurun_out/${t}_sq/sq_counters.csv profiles/${t}_sq_counters.csv
python3 - "$t" <<'PY'
import csv, json, sys
t = sys.argv[1]
rows = list(csv.DictReader(open('profiles/%s_kernel_stats_1stream.csv' % t)))
tot = sum(float(r['total_ms']) for r in rows)
n = [int(r['calls']) for r in rows if 'attn_v3' in r['kernel']][0] / 12
print("kernel ms per forward (one stream): %.2f over %d forwards" % (tot / n, n))
for nm in ('attn_v3', 'lookup_conv', 'SfGruZR', 'SfGruQ', 'FlowHead', 'stem_sf', 'in_apply_sf', 'corr_bricks', 'qk_softmax'):
    print("  ", nm, [(r['calls'], r['avg_us']) for r in rows if nm in r['kernel']])
d = json.loads(open('profiles/%s_bench_default.json' % t).read().strip().splitlines()[-1])
print("value %.1f  ms/step %.2f  config3 %.1f (%.4f, scan %.1f ms)  h2d %.1f  f16 %.1f (%.3f px)  cpu %.2f / %.2f  roofline frac %.3f at %.1f us" % (
    d['value'], d['ms_per_step'], d['config3']['value'], d['config3']['ratio_to_value'], d['config3']['scan_rel2abs_ms'],
    d['h2d_inclusive']['value'], d['f16_fast']['value'], d['f16_fast']['flow_up_abs_diff_vs_split_f16_px']['max'],
    d['cpu_baseline']['value'], d.get('cpu_baseline_1thread', {}).get('value', 0), d['roofline']['frac'], 1e3 * d['roofline']['launch_ms']))
print("stages:", {k: round(x, 2) for k, x in d['stages_ms_per_forward'].items()})
PY
