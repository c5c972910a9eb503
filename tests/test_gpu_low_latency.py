"""The low-latency form of the flow network (RAFTGMA(low_latency=True), atdn_gma_set_low_latency): the reference's per-frame call
pattern runs ONE pair per call (neural_slam.py:202), which leaves attention x V with 29 blocks on 256 CUs; the split form cuts its
key axis into up to 8 ranges with fp32 partial sums. Another summation order, so: within rounding of the default path and inside
the stated tolerances of the oracle, not bit-identical — and the default path must not change."""
import time

import numpy as np
import pytest
import torch

from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import RAFTGMA

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _net(low_latency, max_batch=1):
    n = RAFTGMA(max_batch=max_batch, low_latency=low_latency)
    n.load_state_dict(syn.to_torch(syn.make_gma_state(seed=1)))
    return n.to(DEV).eval()


def test_low_latency_matches_the_oracle_at_c1():
    """BASELINE config 1 (512x160, 8 iterations): 5 tiles x 5 key ranges; flows inside the stated tolerances of the CPU oracle."""
    from oracle import gma_ref
    gsd = syn.to_torch(syn.make_gma_state(seed=1))
    fr = torch.from_numpy(syn.make_frames(2, 160, 512, seed=3))
    ref_low, ref_up = gma_ref.gma_forward(gsd, fr[0:1], fr[1:2], iters=8)
    low, up = _net(True)(fr[0:1].to(DEV), fr[1:2].to(DEV), iters=8, test_mode=True)
    assert float((low.cpu() - ref_low).abs().max()) < 2e-4 and float((up.cpu() - ref_up).abs().max()) < 1e-3


@pytest.mark.parametrize("B", [1, 2, 4])
def test_low_latency_agrees_with_the_default_path_at_kitti_size(B):
    """376x1232, 12 iterations, B = 1 / 2 / 4 pairs per call (8 / 4 / 2 key ranges): the same flow within 2.5e-4 px of the default
    path — a quarter of the stated flow_up tolerance; measured 8e-5 px on flows of up to 70 px = 11 ulp: twelve GRU iterations
    amplify the last-bit differences of the aggregate, finite, nothing clamped; the
    default handle gives the same bits before and after a low-latency handle has run on the device."""
    fr = torch.from_numpy(syn.make_frames(B + 1, 376, 1232, seed=40 + B)).to(DEV)
    base, fast = _net(False, B), _net(True, B)
    low0, up0 = base.forward_sequence(fr, iters=12)
    low1, up1 = fast.forward_sequence(fr, iters=12)
    low2, up2 = base.forward_sequence(fr, iters=12)
    torch.cuda.synchronize()
    assert torch.equal(up0, up2) and torch.equal(low0, low2)
    assert bool(torch.isfinite(up1).all())
    d = float((up1 - up0).abs().max())
    assert 0.0 < d < 2.5e-4, d        # not the same bits (another summation order), the same flow
    assert fast.check_saturation(raise_on_clamp=False) == 0
    # repeatable: the partial sums are added in a fixed order
    low3, up3 = fast.forward_sequence(fr, iters=12)
    assert torch.equal(up1, up3)


def test_low_latency_is_faster_for_one_pair_per_call():
    fr = torch.from_numpy(syn.make_frames(2, 376, 1232, seed=9)).to(DEV)
    out = {}
    for name, net in (("default", _net(False)), ("low_latency", _net(True))):
        for _ in range(3):
            net(fr[0:1], fr[1:2], iters=12, test_mode=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            net(fr[0:1], fr[1:2], iters=12, test_mode=True)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / 10 * 1e3
    print("single-pair forward, 376x1232, 12 iterations: default %.2f ms, low-latency %.2f ms" % (out["default"], out["low_latency"]))
    assert out["low_latency"] < out["default"]


@pytest.mark.parametrize("low_latency", [False, True])
def test_forward_consecutive_is_pair_mode_minus_one_feature_pass(low_latency):
    """RAFTGMA.forward_consecutive(prev, cur): what the per-frame callers use. While the chain of calls is unbroken only `cur`
    passes the feature network (the continued form of the sequence call, one pair); the flows are pair mode's, bit for bit.
    A pair-mode call in between, a frame modified in place, or a `prev` that is not the previous `cur` end the chain — the next
    call encodes both frames again and still gives pair mode's bits."""
    net = _net(low_latency)
    ref = _net(low_latency)
    fr = [f.clone() for f in torch.from_numpy(syn.make_frames(7, 160, 512, seed=31)).to(DEV)]

    def pair(a, b):
        return ref(a[None], b[None], iters=6, test_mode=True)[1]

    for k in range(1, 4):                                   # unbroken chain: calls 2 and 3 are continued
        up = net.forward_consecutive(fr[k - 1], fr[k], iters=6)[1]
        assert torch.equal(up, pair(fr[k - 1], fr[k])), k
    assert net._stream_tail is not None and net._stream_tail[0] is fr[3]
    net(fr[0][None], fr[1][None], iters=6, test_mode=True)  # something else runs on the module: the chain ends
    assert net._stream_tail is None
    up = net.forward_consecutive(fr[3], fr[4], iters=6)[1]
    assert torch.equal(up, pair(fr[3], fr[4]))
    fr[4].mul_(0.5)                                         # the previous frame is modified in place: its features are stale
    up = net.forward_consecutive(fr[4], fr[5], iters=6)[1]
    assert torch.equal(up, pair(fr[4], fr[5]))
    other = fr[5].clone()                                   # equal content, another tensor: not trusted, encoded again
    up = net.forward_consecutive(other, fr[6], iters=6)[1]
    assert torch.equal(up, pair(other, fr[6]))
