# A/B of two library builds in one job: bash tools/diag/ab_bench.sh <old.so>   (new = the in-tree library)
export TMPDIR=/tmp
LEGS="--no-cpu-baseline --no-config3 --no-f16-leg --no-h2d-leg"
for r in 1 2; do
  python3 bench.py --steps 10 --warmup 2 $LEGS > gpurun_out/ab_new_$r.json 2> gpurun_out/ab_new_$r.err
  ATDN_LIB_PATH=$1 python3 bench.py --steps 10 --warmup 2 $LEGS > gpurun_out/ab_old_$r.json 2> gpurun_out/ab_old_$r.err
done
python3 - <<'PY'
import json
for t in ("new_1", "old_1", "new_2", "old_2"):
    d = json.loads([l for l in open("gpurun_out/ab_%s.json" % t) if l.startswith("{")][-1])
    st = d["stages_ms_per_forward"]
    print(t, "%.1f pairs/s" % d["value"], {k: st[k] for k in ("gru_zr", "gru_q", "gru_zr_v", "gru_q_v", "motion_encoder", "flow_head", "fnet", "cnet")})
PY
