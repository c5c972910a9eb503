"""Round-2 parity tests (`-m gpu`, through the C ABI): exactly what bench.py times, the full-resolution C2 flow, a long
recurrent scan (BASELINE config 3's regime) and the domain of validity of the split-f16 arithmetic.

Tolerances are the stated ones of the default (fp32-grade) path: flow_low <= 2e-4 px, flow_up <= 1e-3 px, features /
pose <= 2e-5 / 1e-5 against the CPU oracle, which tests/test_oracle_golden.py pins to the imported reference."""
import os

import numpy as np
import pytest
import torch

from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import ATDNVO, RAFTGMA

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _maxerr(a, b):
    return float((a.double() - b.double()).abs().max())


@pytest.fixture(scope="module")
def gsd():
    return syn.to_torch(syn.make_gma_state(seed=1))


@pytest.fixture(scope="module")
def hsd():
    return syn.to_torch(syn.make_clvo_state(seed=1))


def _u8_sequence(n, seed):
    return torch.from_numpy(syn.make_frames(n, 376, 1241, seed=seed)).round().clamp(0, 255).to(torch.uint8)


def test_bench_step_pattern_matches_pair_mode_and_oracle(gsd, hsd):
    """The launch pattern bench.py times: two OdometryPipelines on two HIP streams, each walking its own sequence in
    clips of B = bench.DEFAULT_BATCH pairs, continued clips reusing the shared frame's features, uint8 host frames ingested (H2D + resize)
    inside the loop. Every clip's flow and features must equal pair mode on a fresh handle (up to kernel-selection
    rounding: the feature network sees 8 instead of 16 images), and two pairs — the first pair of a continued clip, whose
    image1 features are the reused ones, and the last pair of the last clip — must match the CPU oracle."""
    from oracle import clvo_ref, gma_ref
    from atdn_vslam_amd.pipeline import FrameIngest, OdometryPipeline, resize_frames
    import bench
    B, S, CLIPS = bench.DEFAULT_BATCH, 2, 2
    pipes = [OdometryPipeline(gsd, hsd, device=DEV, max_batch=B, iters=12) for _ in range(S)]
    ingests = [FrameIngest((376, 1241), max_frames=B + 1, device=DEV) for _ in range(S)]
    streams = [torch.cuda.Stream(device=DEV) for _ in range(S)]
    seqs = [_u8_sequence(CLIPS * B + 1, seed=300 + p).pin_memory() for p in range(S)]
    feats = [[None] * CLIPS for _ in range(S)]
    flows = [[None] * CLIPS for _ in range(S)]
    for j in range(CLIPS):
        for p in range(S):
            with torch.cuda.stream(streams[p]):
                frames = ingests[p](seqs[p][j * B:(j + 1) * B + 1])
                feats[p][j], flows[p][j] = pipes[p].features_clip(frames, continued=(j > 0))
    torch.cuda.synchronize()
    ref = OdometryPipeline(gsd, hsd, device=DEV, max_batch=B, iters=12)
    for p in range(S):
        for j in range(CLIPS):
            fr = resize_frames(seqs[p][j * B:(j + 1) * B + 1].to(DEV))
            f2, up2 = ref.features(fr[:-1], fr[1:])            # pair mode: both images of every pair encoded
            assert torch.isfinite(flows[p][j]).all()
            assert _maxerr(flows[p][j], up2) < 2e-4, (p, j, _maxerr(flows[p][j], up2))
            assert _maxerr(feats[p][j], f2) < 1e-5, (p, j)
    for (p, j, b) in ((0, 1, 0), (1, CLIPS - 1, B - 1)):
        fr = resize_frames(seqs[p][j * B + b:j * B + b + 2].to(DEV)).cpu()
        _, ref_up = gma_ref.gma_forward(gsd, fr[0:1], fr[1:2], iters=12)
        assert _maxerr(flows[p][j][b:b + 1].cpu(), ref_up) < 1e-3, (p, j, b)
        ref_feat = clvo_ref.clvo_encode(hsd, ref_up)
        assert _maxerr(feats[p][j][b:b + 1].cpu(), ref_feat) < 2e-5, (p, j, b)
    # nothing on this path left the split-f16 range
    assert float(pipes[0].flow_net.debug_read("sf_clamped", (1,), 376, 1232)[0]) == 0.0


def test_c2_full_resolution_flow_matches_oracle_and_golden(golden_dir, gsd):
    """BASELINE configs[1] (376x1232, 12 iterations): the WHOLE flow_up map against the CPU oracle, element by element
    (round 1 compared a stride-4 subsample and channel sums), and the reference's golden flow_low / flow_up samples."""
    from oracle import gma_ref
    g = np.load(os.path.join(golden_dir, "gma_c2.npz"))
    fr = torch.from_numpy(syn.make_frames(2, 376, 1232, seed=int(g["seed_frames"])))
    net = RAFTGMA(max_batch=1)
    net.load_state_dict(gsd)
    net = net.to(DEV).eval()
    low, up = net(fr[0:1].to(DEV), fr[1:2].to(DEV), iters=int(g["iters"]), test_mode=True)
    ref_low, ref_up = gma_ref.gma_forward(gsd, fr[0:1], fr[1:2], iters=int(g["iters"]))
    assert _maxerr(low.cpu(), ref_low) < 2e-4
    assert _maxerr(up.cpu(), ref_up) < 1e-3
    assert _maxerr(low.cpu()[0], torch.from_numpy(g["flow_low"])) < 2e-4
    assert _maxerr(up.cpu()[0, :, ::4, ::4], torch.from_numpy(g["flow_up_s4"])) < 1e-3
    assert float(ref_up.abs().max()) > 10.0      # a real flow field, not a trivial one


def test_lstm_scan_512_steps_matches_oracle(hsd):
    """BASELINE config 3 never resets the LSTM state over a 4,540-pair sequence (evaluate_odometry.py:60-75): a 512-step
    scan on the GPU (one atdn_clvo_step call, and the same in four chunks with the state carried) against 512 sequential
    clvo_ref.clvo_step calls — outputs at every step and the final state (drift would show here)."""
    from oracle import clvo_ref
    T = 512
    r = np.random.RandomState(7)
    # features of the magnitude the encoder produces (|feat| <= 0.4 in clvo.npz), slowly varying like a real sequence
    base = r.normal(0, 0.12, (1, 512)).astype(np.float32)
    walk = np.cumsum(r.normal(0, 0.01, (T, 512)).astype(np.float32), axis=0)
    feats = torch.from_numpy(base + walk + r.normal(0, 0.03, (T, 512)).astype(np.float32))
    state = clvo_ref.zero_state(1)
    ref_rot, ref_tr = [], []
    for t in range(T):
        ro, tr, state = clvo_ref.clvo_step(hsd, feats[t:t + 1], state)
        ref_rot.append(ro)
        ref_tr.append(tr)
    ref_rot, ref_tr = torch.cat(ref_rot), torch.cat(ref_tr)
    head = ATDNVO()
    head.load_state_dict(hsd)
    head = head.to(DEV).eval()
    rot, tr, st = head.scan(feats.to(DEV)[:, None, :])
    assert _maxerr(rot[:, 0].cpu(), ref_rot) < 1e-5 and _maxerr(tr[:, 0].cpu(), ref_tr) < 1e-5
    for k in range(4):
        assert _maxerr(st[k].cpu(), state[k]) < 2e-5, k
    # chunked with the state carried: identical to the single call
    st2, rots = None, []
    for c in range(0, T, 128):
        ro, _, st2 = head.scan(feats[c:c + 128].to(DEV)[:, None, :], state=st2)
        rots.append(ro)
    assert torch.equal(torch.cat(rots), rot) and torch.equal(st2, st)
    assert float(ref_rot.std()) > 1e-4      # the outputs do move along the sequence


def _scaled_state(gsd, scale):
    """The synthetic checkpoint with the context network's stem scaled: BatchNorm is folded, so every activation of
    cnet.layer1 (and, through the ReLUs, of the layers behind it) grows / shrinks by about that factor."""
    sd = {k: v.clone() for k, v in gsd.items()}
    sd["cnet.conv1.weight"] = sd["cnet.conv1.weight"] * scale
    sd["cnet.conv1.bias"] = sd["cnet.conv1.bias"] * scale
    return sd


@pytest.mark.parametrize("scale", [16.0, 1.0 / 256.0])
def test_split_f16_domain_of_validity(gsd, scale):
    """Split-f16 storage holds |x| <= 65504 and floors the absolute error at 3e-8 below |x| ~ 0.06 (sf.h). The C1 forward
    with the context network's activations scaled x16 (heavier tails) and x1/256 (deep in the subnormal-residual range):
    split_f16 must stay as close to the CPU oracle as the exact-fp32 MFMA mode does (within the stated tolerances, and
    within 3x the fp32 mode's own error + 1e-5), and must report zero clamped values."""
    from oracle import gma_ref
    sd = _scaled_state(gsd, scale)
    fr = torch.from_numpy(syn.make_frames(2, 160, 512, seed=71))
    ref_low, ref_up = gma_ref.gma_forward(sd, fr[0:1], fr[1:2], iters=8)
    errs = {}
    for prec in ("split_f16", "f32"):
        net = RAFTGMA(max_batch=1, precision=prec)
        net.load_state_dict(sd)
        net = net.to(DEV).eval()
        net(fr[0:1].to(DEV), fr[1:2].to(DEV), iters=1, test_mode=True)                        # builds the handle
        net.debug_read("sf_clamped", (1,), 160, 512)                                          # reset the counter
        low, up = net(fr[0:1].to(DEV), fr[1:2].to(DEV), iters=8, test_mode=True)
        errs[prec] = (_maxerr(low.cpu(), ref_low), _maxerr(up.cpu(), ref_up))
        if prec == "split_f16":
            assert float(net.debug_read("sf_clamped", (1,), 160, 512)[0]) == 0.0
    assert errs["split_f16"][0] < 2e-4 and errs["split_f16"][1] < 1e-3, errs
    assert errs["split_f16"][1] < 3.0 * errs["f32"][1] + 1e-5, errs


def test_split_f16_saturation_is_counted(gsd):
    """Activations beyond the f16 range: the counter behind atdn_gma_debug_read("sf_clamped") must see them (a real
    checkpoint with a hot channel would otherwise be silently wrong), and the exact-fp32 mode is the way out."""
    sd = _scaled_state(gsd, 3.0e5)
    fr = torch.from_numpy(syn.make_frames(2, 160, 512, seed=71))
    net = RAFTGMA(max_batch=1)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    # round 3: the module reads the counter itself after the first forward of a fresh checkpoint and raises
    from atdn_vslam_amd.modules import SplitF16RangeError
    with pytest.raises(SplitF16RangeError):
        net(fr[0:1].to(DEV), fr[1:2].to(DEV), iters=1, test_mode=True)
    assert float(net.debug_read("sf_clamped", (1,), 160, 512)[0]) == 0.0   # reading resets it
    net._sat_pending = False                                               # the raw counter, without the guard
    net.saturation_check_every = 0
    net(fr[0:1].to(DEV), fr[1:2].to(DEV), iters=1, test_mode=True)
    assert float(net.debug_read("sf_clamped", (1,), 160, 512)[0]) > 0.0


def test_fused_attention_agrees_with_the_separate_passes(gsd, monkeypatch):
    """Round-2 attention (QK^T with the softmax fused in, fragment-major storage, streaming attention x V) against the
    round-1 path it replaced (logits GEMM -> softmax pass -> generic GEMM kernel), which stays selectable
    (ATDN_ATTN_LEGACY=1): attention rows, the aggregated motion features and the flow, at the plumbing size and a
    ragged one (N = 23 * 41 = 943: partial last strip, odd chunk count)."""
    sd = {"module." + k: v for k, v in gsd.items()}

    def run(h, w):
        m = RAFTGMA(max_batch=2, precision="split_f16")
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        fr = torch.from_numpy(syn.make_frames(3, h, w, seed=29)).to(DEV)
        low, up = m.forward_sequence(fr, iters=4)
        n = (h // 8) * (w // 8)
        ldn = (n + 31) // 32 * 32
        attn = m.debug_read("attn", (2, n, ldn), h, w)[:, :, :n]
        x = m.debug_read("x", (2 * n, 384), h, w)
        return low.cpu(), up.cpu(), attn, x[:, 256:384]

    for (h, w) in ((160, 512), (184, 328)):
        new = run(h, w)
        monkeypatch.setenv("ATDN_ATTN_LEGACY", "1")
        old = run(h, w)
        monkeypatch.delenv("ATDN_ATTN_LEGACY")
        assert _maxerr(new[2].sum(-1), torch.ones_like(new[2].sum(-1))) < 1e-5
        assert _maxerr(new[2], old[2]) < 1e-6 + 1e-4 * float(old[2].max())
        assert _maxerr(new[3], old[3]) < 1e-4          # motion_features_global after the last iteration
        assert _maxerr(new[0], old[0]) < 1e-4 and _maxerr(new[1], old[1]) < 5e-4


def test_bricked_pyramid_and_fused_lookup_agree_with_the_separate_kernels(gsd, monkeypatch):
    """Round-2 lookup (bricked pyramid, every level a GEMM against pooled features, lookup fused with convc1) against the
    round-1 path (row-major pyramid, pooled volume, separate lookup and 1x1 convolution; ATDN_LOOKUP_LEGACY=1): pyramid
    levels, the 324 samples, and the flow, at the plumbing size and a ragged one with a non-zero flow_init (windows that
    leave the map on every side)."""
    sd = {"module." + k: v for k, v in gsd.items()}

    def run(h, w):
        m = RAFTGMA(max_batch=2, precision="split_f16")
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        fr = torch.from_numpy(syn.make_frames(3, h, w, seed=31)).to(DEV)
        h8, w8 = h // 8, w // 8
        r = np.random.RandomState(5)
        fi = torch.from_numpy(r.uniform(-0.7, 0.7, (2, 2, h8, w8)).astype(np.float32) * np.array([w8, h8], np.float32).reshape(1, 2, 1, 1))
        low, up = m.forward_sequence(fr, iters=3, flow_init=fi.to(DEV))
        n = h8 * w8
        pyr = [m.debug_read("pyr%d" % l, (2 * n, (h8 >> l) * (w8 >> l)), h, w) for l in range(4)]
        look = m.debug_read("corrfeat", (2 * n, 352), h, w)[:, :324]
        return low.cpu(), up.cpu(), pyr, look

    for (h, w) in ((160, 512), (184, 328)):
        new = run(h, w)
        monkeypatch.setenv("ATDN_LOOKUP_LEGACY", "1")
        old = run(h, w)
        monkeypatch.delenv("ATDN_LOOKUP_LEGACY")
        for l in range(4):
            assert _maxerr(new[2][l], old[2][l]) < 2e-5, l      # pooled features vs pooled volume: fp32 rounding only
        assert _maxerr(new[3], old[3]) < 1e-4
        assert float(old[3].abs().max()) > 0.1 and float((old[3] == 0).float().mean()) > 0.01   # inside AND outside the map
        assert _maxerr(new[0], old[0]) < 1e-4 and _maxerr(new[1], old[1]) < 5e-4


def test_split_f16_stem_agrees_with_the_exact_fp32_stem(gsd, monkeypatch):
    """The 7x7 stems on the split-f16 engine (stem_sf.hip; InstanceNorm = statistics pass + recompute-and-normalise
    pass, nothing raw in memory) against the exact-fp32 ROW-mode stem + separate normalisation they replaced
    (ATDN_STEM_LEGACY=1): feature maps (fnet, InstanceNorm), hidden state / context (cnet, folded BatchNorm) and flow.
    Sizes: the plumbing size, one whose half-resolution map has partial tiles in both directions (92 x 164: 92 % 8 = 4,
    164 % 32 = 4), and KITTI (188 x 616)."""
    sd = {"module." + k: v for k, v in gsd.items()}

    def run(h, w):
        m = RAFTGMA(max_batch=2, precision="split_f16")
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        fr = torch.from_numpy(syn.make_frames(3, h, w, seed=37)).to(DEV)
        low, up = m.forward_sequence(fr, iters=2)
        n = (h // 8) * (w // 8)
        fmap = m.debug_read("fmap", (3 * n, 256), h, w)
        x = m.debug_read("x", (2 * n, 384), h, w)[:, :128]      # relu half of the context network's output
        return low.cpu(), up.cpu(), fmap, x

    for (h, w) in ((160, 512), (184, 328), (376, 1232)):
        new = run(h, w)
        monkeypatch.setenv("ATDN_STEM_LEGACY", "1")
        old = run(h, w)
        monkeypatch.delenv("ATDN_STEM_LEGACY")
        assert float(old[2].abs().max()) > 0.1 and float(old[3].abs().max()) > 0.01
        assert _maxerr(new[2], old[2]) < 2e-5 * max(1.0, float(old[2].abs().max())), (h, w)
        assert _maxerr(new[3], old[3]) < 2e-5 * max(1.0, float(old[3].abs().max())), (h, w)
        assert _maxerr(new[0], old[0]) < 1e-4 and _maxerr(new[1], old[1]) < 5e-4, (h, w)


def test_fused_flow_head_agrees_with_the_two_convolutions(gsd, monkeypatch):
    """Flow head with conv2 folded into conv1's epilogue (18 partial sums per pixel + a 3 x 3 gather; used when one
    256-wide block holds all of conv1's channels, i.e. at B = 8 KITTI) against conv1 -> sf tensor -> conv2 kernel
    (ATDN_FLOWHEAD_FUSED=0): coordinates after 1 and after 4 iterations, the flow channels of the GRU input and the
    upsampled flow. Batch 8 at KITTI size selects the fused path; a ragged size with B = 8 checks partial tiles (the
    map's border pixels get fewer than nine taps)."""
    sd = {"module." + k: v for k, v in gsd.items()}

    def run(h, w, iters):
        m = RAFTGMA(max_batch=8, precision="split_f16")
        m.load_state_dict(sd)
        m = m.to(DEV).eval()
        fr = torch.from_numpy(syn.make_frames(9, h, w, seed=43)).to(DEV)
        low, up = m.forward_sequence(fr, iters=iters)
        n = (h // 8) * (w // 8)
        c1 = m.debug_read("coords1", (8 * n, 2), h, w)
        xf = m.debug_read("x", (8 * n, 384), h, w)[:, 254:256]
        return low.cpu(), up.cpu(), c1, xf

    for (h, w, iters) in ((376, 1232, 1), (376, 1232, 4), (360, 1000, 2)):
        new = run(h, w, iters)
        monkeypatch.setenv("ATDN_FLOWHEAD_FUSED", "0")
        old = run(h, w, iters)
        monkeypatch.delenv("ATDN_FLOWHEAD_FUSED")
        assert float(old[0].abs().max()) > 1e-3
        assert _maxerr(new[2], old[2]) < 1e-4 and _maxerr(new[3], old[3]) < 1e-4, (h, w, iters)
        assert _maxerr(new[0], old[0]) < 1e-4 and _maxerr(new[1], old[1]) < 5e-4, (h, w, iters)


def test_sequence_driver_from_host_uint8_matches_frame_by_frame(gsd, hsd):
    """The sequence driver on a 40-frame uint8 camera sequence in (pinned) host memory — ingest (H2D on the copy stream
    + convert + resize) clip by clip, continued clips, one ordered scan — against the reference's call pattern frame by
    frame (VisualOdometry = NeuralSLAM.__call__ in odometry mode, neural_slam.py:192-227). World size 1 here; the
    sharding of the same driver over ranks is covered on gloo (tests/test_sharding.py)."""
    from atdn_vslam_amd.pipeline import OdometryPipeline, VisualOdometry
    frames = _u8_sequence(40, seed=77).pin_memory()
    pipe = OdometryPipeline(gsd, hsd, device=DEV, max_batch=8)
    poses = pipe.run_sequence(frames, batch=8)
    assert poses.dtype == torch.float64 and tuple(poses.shape) == (40, 4, 4)
    # a clip length that does not divide the sequence (39 pairs = 5 x 7 + 4) gives the same trajectory
    poses7 = pipe.run_sequence(frames, batch=7)
    assert _maxerr(poses7, poses) < 1e-5
    vo = VisualOdometry(gsd, hsd, device=DEV)
    for i in range(40):
        p = vo(frames[i])
        if i in (1, 17, 39):
            assert _maxerr(poses[i].float(), p) < 2e-4, i      # fp64 vs fp32 pose accumulation over up to 39 steps
    assert float(poses[-1][:3, 3].norm()) > 0.1
    # device-resident uint8 frames take the same path minus the copy
    assert _maxerr(pipe.run_sequence(frames.to(DEV), batch=8), poses) == 0.0
