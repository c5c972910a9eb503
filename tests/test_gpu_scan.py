"""The persistent LSTM scan (csrc/lstm_scan.hip; SURVEY K17, atdn_vslam/odometry/network.py:137-146): ONE launch per sequence with
the recurrent weights resident in registers and tagged-granule hand-offs between the 64 workgroups, against the per-step kernel
it replaces for sequences of one batch row (`ATDN_SCAN_PERSISTENT=0` keeps that one) and against the CPU oracle."""
import os
import time

import numpy as np
import pytest
import torch

from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import ATDNVO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _head(persistent):
    old = os.environ.get("ATDN_SCAN_PERSISTENT")
    os.environ["ATDN_SCAN_PERSISTENT"] = "1" if persistent else "0"   # read when the handle is finalized
    try:
        h = ATDNVO()
        h.load_state_dict(syn.to_torch(syn.make_clvo_state(seed=1)))
        h = h.to(DEV).eval()
        h.scan(torch.zeros(2, 1, 512, device=DEV))   # the handle exists now
    finally:
        if old is None:
            os.environ.pop("ATDN_SCAN_PERSISTENT")
        else:
            os.environ["ATDN_SCAN_PERSISTENT"] = old
    return h


def _feats(T, seed):
    r = np.random.RandomState(seed)
    base = r.normal(0, 0.12, (1, 512)).astype(np.float32)
    walk = np.cumsum(r.normal(0, 0.01, (T, 512)).astype(np.float32), axis=0)
    return torch.from_numpy(base + walk + r.normal(0, 0.03, (T, 512)).astype(np.float32)).to(DEV)[:, None, :]


@pytest.mark.parametrize("T", [16, 17, 100, 1000])
def test_persistent_scan_matches_the_per_step_kernel(T):
    """Same recurrence, different schedule (and v_exp / v_rcp gate functions instead of libm): poses of every step within 2e-6,
    the carried state within 5e-6; a non-zero initial state; the state is carried exactly like the per-step kernel carries it."""
    per, one = _head(False), _head(True)
    f = _feats(T, 11 + T)
    st0 = torch.from_numpy(np.random.RandomState(3).normal(0, 0.2, (4, 1, 512)).astype(np.float32)).to(DEV)
    r0, t0, s0 = per.scan(f, state=st0)
    r1, t1, s1 = one.scan(f, state=st0)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(r1).all()) and bool(torch.isfinite(s1).all())
    assert float((r1 - r0).abs().max()) < 2e-6 and float((t1 - t0).abs().max()) < 2e-6, (float((r1 - r0).abs().max()), float((t1 - t0).abs().max()))
    assert float((s1 - s0).abs().max()) < 5e-6
    assert float(r0.std()) > 1e-5
    # deterministic: a second run gives the same bits (fixed summation order, no atomics on the data path)
    r2, t2, s2 = one.scan(f, state=st0)
    assert torch.equal(r1, r2) and torch.equal(t1, t2) and torch.equal(s1, s2)
    # chunks of 50 steps with the state carried (each one a persistent launch of its own) = the single call, bit for bit
    if T >= 100:
        st, rr = st0, []
        for c in range(0, T, 50):
            ro, _, st = one.scan(f[c:c + 50], state=st)
            rr.append(ro)
        assert torch.equal(torch.cat(rr), r1) and torch.equal(st, s1)


def test_short_sequences_and_batched_rows_keep_the_per_step_kernel():
    """T < 16 (the per-frame callers: T = 1) and batch rows > 1 never take the persistent kernel: same bits from both handles."""
    per, one = _head(False), _head(True)
    f = _feats(15, 5)
    a, b = per.scan(f), one.scan(f)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    f2 = _feats(40, 6).repeat(1, 3, 1).contiguous()
    a, b = per.scan(f2), one.scan(f2)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])


def test_persistent_scan_of_a_kitti00_length_sequence_is_faster_and_finite():
    """4,540 steps (BASELINE configs[2]): finite poses, agreement with the per-step kernel at the END of the sequence (drift over
    4,540 steps of the two gate-function implementations stays below 1e-5), and the wall time of both printed (-s shows it)."""
    per, one = _head(False), _head(True)
    f = _feats(4540, 9)
    for h in (per, one):
        h.scan(f)   # warm: the per-step path graphs a length on second sight
        h.scan(f)
    torch.cuda.synchronize()
    out = {}
    for name, h in (("per_step", per), ("persistent", one)):
        t0 = time.perf_counter()
        for _ in range(3):
            r, t, s = h.scan(f)
        torch.cuda.synchronize()
        out[name] = ((time.perf_counter() - t0) / 3 * 1e3, r, t, s)
    print("scan of 4540 steps: per-step %.2f ms, persistent %.2f ms" % (out["per_step"][0], out["persistent"][0]))
    assert bool(torch.isfinite(out["persistent"][1]).all())
    assert float((out["persistent"][1] - out["per_step"][1]).abs().max()) < 1e-5
    assert float((out["persistent"][2] - out["per_step"][2]).abs().max()) < 1e-5
    assert out["persistent"][0] < out["per_step"][0]


def test_persistent_scans_share_the_gpu_with_each_other_and_with_a_flow_forward():
    """Residency is not a given: two persistent scans on two streams at once (256 workgroups), and a scan that starts while another
    stream's flow network holds the chip, must give the bits of a scan that ran alone — a workgroup that waits for a CU is late,
    not lost (the spins are bounded by wall time, 0.5 s)."""
    from atdn_vslam_amd.modules import RAFTGMA
    a, b = _head(True), _head(True)
    fa, fb = _feats(600, 21), _feats(700, 22)
    ra = a.scan(fa)
    rb = b.scan(fb)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        ra2 = a.scan(fa)
    with torch.cuda.stream(s2):
        rb2 = b.scan(fb)
    torch.cuda.synchronize()
    assert torch.equal(ra[0], ra2[0]) and torch.equal(rb[0], rb2[0]) and torch.equal(ra[2], ra2[2])
    net = RAFTGMA(max_batch=4)
    net.load_state_dict(syn.to_torch(syn.make_gma_state(seed=1)))
    net = net.to(DEV).eval()
    fr = torch.from_numpy(syn.make_frames(5, 376, 1232, seed=3)).to(DEV)
    net.forward_sequence(fr, iters=4)
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        for _ in range(3):
            net.forward_sequence(fr, iters=4)
    with torch.cuda.stream(s2):
        ra3 = a.scan(fa)
    torch.cuda.synchronize()
    assert torch.equal(ra[0], ra3[0]) and torch.equal(ra[1], ra3[1]) and torch.equal(ra[2], ra3[2])


def test_a_persistent_launch_that_gives_up_is_repeated_on_the_per_step_kernel(capfd):
    """The persistent launch is verified before its results are used (ClvoNet::step synchronises and reads the abort word): a
    launch that gave up — here forced by ATDN_SCAN_TEST_ABORT — leaves NaN behind, and the caller must never see it: the sequence is
    repeated on the per-step kernel from the saved state (same bits as a handle that never had the persistent kernel), a line goes
    to stderr, and the handle stays on the per-step kernel afterwards."""
    per, one = _head(False), _head(True)
    f = _feats(200, 33)
    st0 = torch.from_numpy(np.random.RandomState(4).normal(0, 0.2, (4, 1, 512)).astype(np.float32)).to(DEV)
    want = per.scan(f, state=st0)
    os.environ["ATDN_SCAN_TEST_ABORT"] = "1"
    try:
        got = one.scan(f, state=st0)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("ATDN_SCAN_TEST_ABORT")
    assert bool(torch.isfinite(got[0]).all())
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
    assert "repeating the sequence on the per-step kernel" in capfd.readouterr().err
    again = one.scan(f, state=st0)                      # the hook is gone, the handle stays where it is
    assert torch.equal(again[0], want[0])
