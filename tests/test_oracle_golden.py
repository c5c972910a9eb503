"""The CPU oracle (oracle/) against fixtures produced by the imported reference
(tests/golden/make_golden.py).  This is what pins the oracle; the GPU parity
tests then compare the HIP path with the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.weights_spec import clvo_state_spec, gma_state_spec
from oracle import clvo_ref, gma_ref, pose_ref


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_state_dict_layout(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "state_keys.json")))
    assert [[k, list(v[0])] for k, v in gma_state_spec().items()] == ref["gma"]
    assert [[k, list(v[0])] for k, v in clvo_state_spec().items()] == ref["clvo"]


@pytest.fixture(scope="module")
def gsd():
    return syn.to_torch(syn.make_gma_state(seed=1))


@pytest.fixture(scope="module")
def hsd():
    return syn.to_torch(syn.make_clvo_state(seed=1))


def test_gma_c1_stages_and_flow(golden_dir, gsd):
    g = _load(golden_dir, "gma_c1.npz")
    fr = torch.from_numpy(syn.make_frames(2, 160, 512, seed=int(g["seed_frames"])))
    taps = {}
    flow_low, flow_up = gma_ref.gma_forward(gsd, fr[0:1], fr[1:2], iters=int(g["iters"]), taps=taps)
    tol = dict(rtol=0, atol=2e-5)
    np.testing.assert_allclose(taps["fmap1"][0, :, ::3, ::5].numpy(), g["fmap1"], **tol)
    np.testing.assert_allclose(taps["fmap2"][0, :, ::3, ::5].numpy(), g["fmap2"], **tol)
    np.testing.assert_allclose(taps["net0"][0, :, ::3, ::5].numpy(), g["net0"], **tol)
    np.testing.assert_allclose(taps["inp"][0, :, ::3, ::5].numpy(), g["inp"], **tol)
    pyr = taps["pyramid"]
    np.testing.assert_allclose(pyr[3].reshape(20, 64, 2, 8)[::3, ::5].numpy(), g["pyr3"], **tol)
    np.testing.assert_allclose(pyr[1].reshape(20, 64, 10, 32)[::7, ::9].numpy(), g["pyr1"], **tol)
    np.testing.assert_allclose(pyr[0].reshape(1280, 1280)[[0, 77, 640, 1279]].numpy(), g["pyr0_rows"], **tol)
    np.testing.assert_allclose(taps["attn"].reshape(1280, 1280)[[0, 77, 640, 1279]].numpy(), g["attn_rows"],
                               rtol=1e-4, atol=1e-8)
    look = gma_ref.corr_lookup(pyr, torch.from_numpy(g["probe"])[None])
    np.testing.assert_allclose(look[0].numpy(), g["lookup"], **tol)
    # out-of-range probe positions sample zero padding only
    assert np.all(g["lookup"][:81, 0, 1] == 0.0)  # level 0 window fully outside
    np.testing.assert_allclose(taps["net1"][0, :, ::3, ::5].numpy(), g["net1"], **tol)
    np.testing.assert_allclose(taps["delta0"][0].numpy(), g["delta1"], **tol)
    # flow after 8 recurrent iterations: stated tolerance 1e-3 px
    np.testing.assert_allclose(flow_low[0].numpy(), g["flow_low"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(flow_up[0].numpy(), g["flow_up"], rtol=0, atol=1e-3)
    assert float(np.abs(g["flow_up"]).max()) > 1.0  # non-trivial flow


def test_gma_flow_predictions_of_every_iteration(golden_dir, gsd):
    """RAFTGMA.forward(test_mode=False) (network.py:106-129): the oracle's per-iteration upsampled flows against the reference's
    own list (tests/golden/make_golden_preds.py), with and without a flow_init."""
    g = _load(golden_dir, "gma_preds.npz")
    fr = torch.from_numpy(syn.make_frames(2, 160, 512, seed=int(g["seed_frames"])))
    preds = []
    _, up = gma_ref.gma_forward(gsd, fr[0:1], fr[1:2], iters=int(g["iters"]), predictions=preds)
    assert len(preds) == int(g["iters"]) and torch.equal(preds[-1], up)
    p = torch.stack(preds, 0)[:, 0]
    np.testing.assert_allclose(p[:, :, ::4, ::4].numpy(), g["preds_s4"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(p.double().abs().sum(dim=(2, 3)).numpy(), g["preds_abs"], rtol=1e-5)
    # the iterations differ from one another (a test that compared eight copies of the last flow would pass nothing)
    assert float(np.abs(g["preds_s4"][0] - g["preds_s4"][-1]).max()) > 0.5
    preds2 = []
    gma_ref.gma_forward(gsd, torch.cat([fr[0:1], fr[1:2]]), torch.cat([fr[1:2], fr[0:1]]), iters=3,
                        flow_init=torch.from_numpy(g["flow_init"]), predictions=preds2)
    p2 = torch.stack(preds2, 0)
    np.testing.assert_allclose(p2[:, :, :, ::4, ::4].numpy(), g["preds2_s4"], rtol=0, atol=1e-3)


@pytest.mark.timeout(600)
def test_gma_c2_flow_and_head(golden_dir, gsd, hsd):
    g = _load(golden_dir, "gma_c2.npz")
    fr = torch.from_numpy(syn.make_frames(2, 376, 1232, seed=int(g["seed_frames"])))
    flow_low, flow_up = gma_ref.gma_forward(gsd, fr[0:1], fr[1:2], iters=int(g["iters"]))
    np.testing.assert_allclose(flow_low[0].numpy(), g["flow_low"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(flow_up[0, :, ::4, ::4].numpy(), g["flow_up_s4"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(flow_up.double().sum(dim=(0, 2, 3)).numpy(), g["flow_up_sum"], rtol=1e-5, atol=1.0)
    rot, tr, _ = clvo_ref.clvo_forward(hsd, flow_up, clvo_ref.zero_state(1))
    np.testing.assert_allclose(rot.numpy(), g["rot"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(tr.numpy(), g["tr"], rtol=0, atol=1e-5)


def test_clvo_head(golden_dir, hsd):
    g = _load(golden_dir, "clvo.npz")
    fl = torch.from_numpy(syn.make_flow(3, 376, 1232, seed=6))
    tol = dict(rtol=0, atol=1e-5)
    np.testing.assert_allclose(clvo_ref.clvo_encode(hsd, fl).numpy(), g["feat"], **tol)
    state = clvo_ref.zero_state(1)
    for t in range(3):
        rot, tr, state = clvo_ref.clvo_forward(hsd, fl[t:t + 1], state)
        np.testing.assert_allclose(rot.numpy(), g["rot%d" % t], **tol)
        np.testing.assert_allclose(tr.numpy(), g["tr%d" % t], **tol)
    rot, tr, _ = clvo_ref.clvo_forward(hsd, fl[1:2], clvo_ref.zero_state(1))
    np.testing.assert_allclose(rot.numpy(), g["rot_after_reset"], **tol)
    assert np.abs(g["rot_after_reset"] - g["rot1"]).max() > 1e-4  # the state matters
    fw = torch.from_numpy(syn.make_flow(1, 376, 1241, seed=7))
    rot, tr, _ = clvo_ref.clvo_forward(hsd, fw[:, :, :, 4:4 + 1232], clvo_ref.zero_state(1))
    np.testing.assert_allclose(rot.numpy(), g["rot_crop"], **tol)
    np.testing.assert_allclose(tr.numpy(), g["tr_crop"], **tol)
    fl4 = torch.from_numpy(syn.make_flow(4, 376, 1232, seed=8))
    rot, tr, st = clvo_ref.clvo_forward(hsd, fl4, clvo_ref.zero_state(4))
    np.testing.assert_allclose(rot.numpy(), g["rot_b4"], **tol)
    rot, tr, st = clvo_ref.clvo_forward(hsd, fl4.flip(0), st)
    np.testing.assert_allclose(rot.numpy(), g["rot_b4_step2"], **tol)
    np.testing.assert_allclose(tr.numpy(), g["tr_b4_step2"], **tol)


def test_pose_algebra(golden_dir):
    g = _load(golden_dir, "pose.npz")
    for i in range(16):
        m = pose_ref.transform(g["rots"][i], g["trs"][i], dtype=np.float32)
        np.testing.assert_allclose(m, g["transform"][i], rtol=0, atol=1e-6)
        np.testing.assert_allclose(pose_ref.matrix2euler(g["transform"][i][:3, :3]), g["euler"][i], rtol=0, atol=1e-6)
    absolute = pose_ref.rel2abs(g["rots"], g["trs"])
    assert absolute.dtype == np.float64 and absolute.shape == (17, 4, 4)
    np.testing.assert_allclose(absolute, g["rel2abs"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(pose_ref.kitti_rows(absolute), g["kitti_rows"], rtol=0, atol=1e-12)
    for (h, w), pad in zip(g["pad_dims"], g["pads"]):
        assert pose_ref.pad_amounts(int(h), int(w)) == list(pad)


def test_vae_encoder_oracle_matches_reference(golden_dir):
    """oracle/vae_ref.py against the imported reference's MappingVAE on the same synthetic weights and frames."""
    from oracle import vae_ref
    g = np.load(os.path.join(golden_dir, "vae.npz"))
    sd = syn.to_torch(syn.make_vae_state(seed=int(g["seed_weights"])))
    frames = torch.from_numpy(syn.make_frames(5, 376, 1232, seed=int(g["seed_frames"])))[:2]
    taps = {}
    mu = vae_ref.vae_encode(sd, frames, taps)
    assert tuple(mu.shape) == (2, 128, 6, 20)
    np.testing.assert_allclose(mu.numpy(), g["mu"], rtol=0, atol=2e-5)
    for k, v in taps.items():
        sub = v[:, :, ::max(1, v.shape[2] // 12), ::max(1, v.shape[3] // 16)].numpy()
        np.testing.assert_allclose(sub, g[k], rtol=0, atol=2e-5, err_msg=k)


def test_vae_state_layout_matches_reference(golden_dir):
    import json
    from atdn_vslam_amd.weights_spec import vae_state_spec
    keys = {k: tuple(s) for k, s in json.load(open(os.path.join(golden_dir, "state_keys.json")))["vae"]}
    spec = vae_state_spec()
    for k, (shape, _) in spec.items():
        assert k in keys and keys[k] == tuple(shape), k
    # everything the spec leaves out belongs to the decoder (training loss only)
    assert all(k.startswith("decoder.") for k in keys if k not in spec)
