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
           "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "2", "--no-cpu-baseline", "--no-h2d-leg", "--no-f16-leg", "--no-f32-leg", "--no-per-frame-leg", "--config3-frames", "44"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    # --steps 3 with two streams per GPU: a step count the streams do not divide
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] > 0 and d["unit"] == "frame-pairs/s" and "REHEARSAL" in d["data"]
    B = d["config"]["pairs_per_step_per_gpu"]
    assert "x2" in d["config"]["parallelism"] and B >= 8
    # whole-job aggregate: 2 ranks x 3 steps x B pairs over the max-over-ranks time
    assert abs(d["value"] - 2 * 3 * B / (d["ms_per_step"] * 3 / 1e3)) < 1e-6 * d["value"] + 1e-3
    # config3 leg (BASELINE configs[2]): 43 pairs over 2 ranks = shards of 22 and 21 pairs: two lanes per rank, the second
    # lane SHORTER than one clip of 16 (6 / 5 pairs); one all-gather, replicated scan
    c3 = d["config3"]
    assert c3["pairs"] == 43 and c3["lanes_per_gpu"] == 2 and c3["value"] > 0
    assert abs(c3["value"] - 43 / c3["wall_s"]) < 1e-6 * c3["value"]
    assert c3["encode_s"] > 0 and c3["allgather_ms"] >= 0 and c3["scan_rel2abs_ms"] > 0


def test_bench_config3_with_a_shard_shorter_than_one_clip():
    """A rank whose whole shard is shorter than one clip (9 pairs over 2 ranks: 5 and 4 pairs, clips of 16): one lane per
    rank, a single short clip, the gather is ragged."""
    env = dict(os.environ, ATDN_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29619", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "2", "--no-cpu-baseline", "--no-h2d-leg", "--no-f16-leg", "--no-f32-leg", "--no-per-frame-leg", "--config3-frames", "10"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config3"]["pairs"] == 9 and d["config3"]["value"] > 0


def test_bench_collectives_on_real_rccl_at_world_size_one():
    """As much of the N > 1 path as ONE GPU can run on the real backend: `bench.py` under torch.distributed.run with one
    rank, process group = nccl (RCCL), every barrier / all-reduce / all-gather of the multi-GPU path executed
    (ATDN_BENCH_FORCE_DIST, ATDN_FORCE_COLLECTIVE), `config3` through run_sequence's gather included. Catches device / dtype /
    ordering mistakes in the collective plumbing that gloo forgives; two ranks on real RCCL need the driver's 8-GPU node."""
    env = dict(os.environ, ATDN_BENCH_FORCE_DIST="1", ATDN_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29621", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "2", "--no-cpu-baseline", "--no-h2d-leg", "--no-f16-leg", "--no-f32-leg", "--no-per-frame-leg", "--config3-frames", "40"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config3"]["pairs"] == 39 and d["config3"]["value"] > 0
    assert "REHEARSAL" not in d["data"]


def test_bench_starts_its_own_ranks():
    """The line a harness types: `python bench.py --gpus 2 ...` with NO launcher and no WORLD_SIZE. bench.py's parent starts
    the two ranks before any GPU call (atdn_vslam_amd/launch.py), relays rank 0's one JSON line and the exit code. On this
    one-GPU box both ranks share cuda:0 over gloo (ATDN_BENCH_REHEARSAL); the N > 1 line carries `cpu_baseline: null` + why."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ATDN_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2", "--no-h2d-leg",
           "--no-f16-leg", "--no-f32-leg", "--no-per-frame-leg", "--config3-frames", "40"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line reaches the parent's stdout"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config3"]["pairs"] == 39
    assert d["cpu_baseline"] is None and "N = 1" in d["cpu_baseline_reason"]


def test_bench_line_carries_the_secondary_legs():
    """One GPU, no launcher: the line the driver records. Beside `value` it carries `roofline`, the legs round 6 added — `f32_exact`
    (what the saturation fallback costs) and `per_frame` (the reference's one-frame-per-call pattern) — and `f16_fast`; none of
    them is `value`."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-h2d-leg",
           "--no-config3"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["vs_baseline"] is None and d["scaling"] == "weak"
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    f32 = d["f32_exact"]
    assert 0 < f32["value"] < d["value"] and abs(f32["ratio_to_value"] - f32["value"] / d["value"]) < 1e-9
    assert f32["flow_up_abs_diff_vs_split_f16_px"]["max"] < 1e-3        # the two fp32-grade modes agree inside the stated tolerance
    pf = d["per_frame"]
    assert 0 < pf["value"] < d["value"] and abs(pf["value"] * pf["ms_per_frame"] - 1e3) < 1e-6 * 1e3
    assert d["f16_fast"]["value"] > d["value"]
    assert d["cpu_baseline"] is None and d["cpu_baseline_reason"] == "--no-cpu-baseline"
