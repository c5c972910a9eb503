"""Data-parallel CLVO training step on CPU (gloo, world size 2): gradient averaging and rank-identical updates."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_data_parallel_step_averages_gradients(tmp_path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(ROOT, "tests", "_dist_train_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=850,
                       env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert r.returncode == 0, r.stdout[-3000:]
    a = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    b = np.load(os.path.join(str(tmp_path), "rank1.npz"))
    assert not np.allclose(a["local"], b["local"])                        # different shards, different gradients
    np.testing.assert_allclose(a["mean"], (a["local"] + b["local"]) / 2, rtol=0, atol=1e-7)
    assert np.array_equal(a["mean"], b["mean"])                           # every rank holds the same averaged gradient
    assert np.array_equal(a["w"], b["w"])                                 # ... and applies the same update
    assert a["loss"] != b["loss"]


@pytest.mark.timeout(420)
def test_world_8_gradient_exchange(tmp_path):
    """VERDICT r4 #5c: the first 8-GPU training run should be boring. Eight gloo ranks average a flat gradient of the trainer's
    size (5.06 M fp32) with `allreduce_mean_` and apply the scheduled AdamW step: the mean is the mean of the eight local
    gradients and every rank ends with the same weights."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29741", os.path.join(ROOT, "tests", "_dist_train_worker.py"), str(tmp_path), "synthetic"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=400,
                       env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-3000:]
    ranks = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(8)]
    want = sum(x["local_head"].astype(np.float64) for x in ranks) / 8
    for x in ranks:
        np.testing.assert_allclose(x["mean_head"], want, rtol=0, atol=1e-6)
        assert np.array_equal(x["mean_head"], ranks[0]["mean_head"]) and x["mean_sum"] == ranks[0]["mean_sum"]
        assert np.array_equal(x["w_head"], ranks[0]["w_head"]) and x["w_sum"] == ranks[0]["w_sum"]
    assert not np.allclose(ranks[0]["local_head"], ranks[7]["local_head"])


def test_allreduce_is_a_noop_without_a_process_group():
    import torch
    from atdn_vslam_amd.training import allreduce_mean_
    g = torch.arange(6, dtype=torch.float32)
    assert torch.equal(allreduce_mean_(g.clone()), g)
