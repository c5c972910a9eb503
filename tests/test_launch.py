"""`python bench.py --gpus N` (N > 1) with NO launcher around it must start its own N ranks (atdn_vslam_amd/launch.py): the
harness that types `python bench.py --gpus 1` types the same line with `--gpus 8`. CPU tier: the launch plumbing only
(`--launch-check`: gloo group, one all-gather of the ranks, no GPU call); the same line through the GPU path is
tests/test_gpu_bench_rehearsal.py::test_bench_starts_its_own_ranks."""
import json
import os
import subprocess
import sys

from atdn_vslam_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS", "TORCHELASTIC_RUN_ID")}
    env.update(kw)
    return env


def _json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_requested_gpus_parses_both_spellings():
    assert launch.requested_gpus(["--steps", "3"]) == 1
    assert launch.requested_gpus(["--gpus", "8", "--steps", "3"]) == 8
    assert launch.requested_gpus(["--steps", "3", "--gpus=4"]) == 4


def test_no_spawn_for_one_gpu_or_inside_a_launched_job():
    assert launch.spawn_ranks_if_needed("bench.py", ["--gpus", "1"], env={}) is None
    assert launch.spawn_ranks_if_needed("bench.py", [], env={}) is None
    # already a rank of somebody's torch.distributed.run: never a second level of ranks
    assert launch.spawn_ranks_if_needed("bench.py", ["--gpus", "8"], env={"WORLD_SIZE": "8"}) is None


def test_launch_command_is_the_drivers_own_line():
    cmd = launch.launch_command("/x/bench.py", ["--gpus", "8", "--steps", "20"], 8, 1234)
    assert cmd[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                       "--master-port", "1234", "/x/bench.py", "--gpus", "8", "--steps", "20"]


def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], cwd=ROOT,
                         env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = _json_lines(out.stdout)
    assert len(lines) == 1, "rank 0 prints exactly one JSON line through the parent's stdout"
    d = lines[0]
    assert d["launch_check"] and d["n_gpus"] == 2 and d["ranks"] == [0, 1] and d["local_ranks"] == [0, 1]
    assert d["launched_by"] == "bench.py itself"
    # every rank got its share of the host's cores, not torch.distributed.run's default of one thread
    share = max(1, launch.host_cores() // 2)
    assert d["cpu_threads_per_rank"] == [share, share]


def test_a_users_thread_count_is_kept():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], cwd=ROOT,
                         env=_env(OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert _json_lines(out.stdout)[0]["cpu_threads_per_rank"] == [1, 1]


def test_a_failing_rank_ends_the_self_launched_job_non_zero():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], cwd=ROOT,
                         env=_env(ATDN_LAUNCH_CHECK_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert not _json_lines(out.stdout)
    # both ranks raised (sharding.rendezvous): nobody waited for a watchdog
    assert out.stderr.count("ShardError: rank 1 failed") >= 2


def test_train_bench_starts_its_own_ranks_too():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_train.py"), "--gpus", "2", "--launch-check"],
                         cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _json_lines(out.stdout)
    assert len(d) == 1 and d[0]["n_gpus"] == 2 and d[0]["ranks"] == [0, 1]
