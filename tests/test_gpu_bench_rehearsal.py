"""bench.py's N > 1 code path (pairs sharded over ranks, barrier, one all-gather of the 512-d features, replicated
ordered scan, max-over-ranks timing, rank 0 prints ONE JSON line) rehearsed on a one-GPU box: two ranks on cuda:0 over
gloo (`ATDN_BENCH_REHEARSAL=1`; RCCL refuses two ranks on one device). Checks the plumbing the driver's 2/4/8-GPU
runs go through, not a number."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_rehearsal_prints_one_valid_line():
    env = dict(os.environ, ATDN_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "2", "--no-cpu-baseline", "--no-h2d-leg"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] > 0 and d["unit"] == "frame-pairs/s" and "REHEARSAL" in d["data"]
    B = d["config"]["pairs_per_step_per_gpu"]
    assert "x2" in d["config"]["parallelism"] and B >= 8
    # whole-job aggregate: 2 ranks x 2 steps x B pairs over the max-over-ranks time
    assert abs(d["value"] - 2 * 2 * B / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"] + 1e-3
