"""GPU parity of the relocalisation row: MappingVAE embedding and the NeuralSLAM drop-in, through
the C ABI, against the committed reference outputs (tests/golden/vae.npz, reloc.npz, slam.npz, keyframes.npz) and the
CPU oracle."""
import os

import numpy as np
import pytest
import torch

from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import MappingVAE

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def vsd():
    return syn.to_torch(syn.make_vae_state(seed=2))


def test_vae_embedding_matches_golden_and_oracle(golden_dir, vsd):
    from oracle import vae_ref
    g = np.load(os.path.join(golden_dir, "vae.npz"))
    frames = torch.from_numpy(syn.make_frames(5, 376, 1232, seed=int(g["seed_frames"])))
    net = MappingVAE()
    net.load_state_dict(vsd)
    mu, logvar, latent, decoded = net(frames[:2].to(DEV))
    assert logvar is None and decoded is None and latent is mu
    assert tuple(mu.shape) == (2, 128, 6, 20)
    # fp32 exact-MFMA path: the reference's values to 5e-5 abs (|mu| <= 10)
    assert float((mu.cpu() - torch.from_numpy(g["mu"])).abs().max()) < 5e-5
    # other batch / single image [3,H,W] / another geometry against the oracle
    one = net(frames[3].to(DEV))[0]
    ref = vae_ref.vae_encode(vsd, frames[3:4])
    assert float((one.cpu() - ref).abs().max()) < 5e-5
    small = torch.from_numpy(syn.make_frames(3, 200, 333, seed=14))
    out = net(small.to(DEV))[0]
    ref = vae_ref.vae_encode(vsd, small)
    assert out.shape == ref.shape and float((out.cpu() - ref).abs().max()) < 5e-5


def test_vae_checkpoint_with_decoder_keys_loads(vsd):
    full = dict(vsd)
    full["decoder.0.conv.0.conv.weight"] = torch.zeros(128, 128, 3, 3)
    net = MappingVAE()
    net.load_state_dict(full)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 376, 1232))  # CPU tensor: no fallback


class _Args:
    def __init__(self, path):
        self.device = DEV
        self.keyframes_path = path


@pytest.fixture(scope="module")
def gsd():
    return syn.to_torch(syn.make_gma_state(seed=1))


@pytest.fixture(scope="module")
def hsd():
    return syn.to_torch(syn.make_clvo_state(seed=1))


def test_neuralslam_odometry_mode_matches_reference(golden_dir, gsd, hsd, tmp_path):
    """Same call sequence as the reference run behind tests/golden/slam.npz: poses per call, keyframe count, and
    the files the reference writes (rgb/000000.pth uint8 [3,376,1232]; poses.pth [K,12] at end_odometry)."""
    from atdn_vslam_amd.slam import NeuralSLAM
    g = np.load(os.path.join(golden_dir, "slam.npz"))
    kf = os.path.join(str(tmp_path), "kf")
    slam = NeuralSLAM(_Args(kf), odometry_weights=hsd, flow_weights={"module." + k: v for k, v in gsd.items()})
    assert slam.mode() == "idle"
    with pytest.raises(Exception):
        slam(torch.zeros(3, 376, 1241))          # called in an invalid state
    slam.start_odometry()
    assert slam.mode() == "odometry"
    frames = torch.from_numpy(syn.make_frames(4, 376, 1241, seed=int(g["seed_frames"])))
    for i in range(4):
        pose = slam(frames[i])
        assert float((pose - torch.from_numpy(g["poses"][i])).abs().max()) < 2e-5, i
    assert len(slam) == int(g["n_keyframes"])
    first = torch.load(os.path.join(kf, "rgb", "000000.pth"))
    assert first.dtype == torch.uint8 and tuple(first.shape) == (3, 376, 1232)
    assert torch.equal(slam[0].pose, torch.eye(4)) and slam.get_keyframe(0).rgb_file_name.endswith("000000.pth")
    # no trained MappingVAE and fewer keyframes than one training batch: poses are persisted, map creation refuses
    # (the reference divides by zero there), the state machine stays in "mapping"
    with pytest.raises(RuntimeError, match="at least 16 keyframes"):
        slam.end_odometry()
    assert slam.mode() == "mapping"
    saved = torch.load(os.path.join(kf, "poses.pth"))
    assert tuple(saved.shape) == (len(slam), 12) and torch.equal(saved[0], torch.eye(4).flatten()[:12])
    # with weights it embeds the keyframes and relocalises
    slam.end_odometry(mapping_weights=syn.to_torch(syn.make_vae_state(seed=2)))
    assert slam.mode() == "relocalization" and slam[0].embedding is not None
    init, refined, dist = slam(first.float())
    assert float(dist[0]) < 1e-3 and torch.equal(init, slam[0].pose)


def test_neuralslam_relocalization_matches_reference(golden_dir, gsd, hsd, vsd, tmp_path):
    """The keyframe directory of tests/golden/make_golden_slam.py rebuilt from seeds; NeuralSLAM started in
    "relocalization" mode must return the reference's distances, initial and refined poses for both queries
    (the second one with the head's LSTM state carried over from the first, as in the reference)."""
    from atdn_vslam_amd.slam import NeuralSLAM
    g = np.load(os.path.join(golden_dir, "reloc.npz"))
    frames = torch.from_numpy(syn.make_frames(5, 376, 1232, seed=int(g["seed_frames"])))
    kf = os.path.join(str(tmp_path), "kf")
    os.makedirs(os.path.join(kf, "rgb"))
    for i in range(3):
        torch.save(frames[i].byte(), os.path.join(kf, "rgb", "%06d.pth" % i))
    torch.save(torch.from_numpy(g["keyframe_poses"]), os.path.join(kf, "poses.pth"))
    torch.save(vsd, os.path.join(kf, "MappingVAE_weights.pth"))
    slam = NeuralSLAM(_Args(kf), odometry_weights=hsd, flow_weights=gsd, start_mode="relocalization")
    assert slam.mode() == "relocalization" and len(slam) == 3
    for name, q in (("near1", frames[1].byte().float()), ("new", frames[4].byte().float())):
        init, refined, dist = slam(q)
        np.testing.assert_allclose(dist.numpy(), g[name + "_distances"], rtol=0, atol=2e-3)   # |mu| ~ 10, 15360 dims
        assert int(torch.argmin(dist)) == int(np.argmin(g[name + "_distances"]))
        np.testing.assert_allclose(init.numpy(), g[name + "_initial"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(refined.numpy(), g[name + "_refined"], rtol=0, atol=5e-5)


def test_neuralslam_creates_its_map_and_relocalizes(gsd, hsd, vsd, tmp_path):
    """BASELINE config 5 end to end: a keyframe directory as odometry leaves it, `start_mode="mapping"` -> the
    MappingVAE is trained on the keyframes (stock PyTorch, two epochs here), every keyframe is embedded on the HIP
    path and queries relocalise. The HIP embedding of the trained weights must agree with the torch module that was
    trained."""
    from atdn_vslam_amd import mapping
    from atdn_vslam_amd.slam import NeuralSLAM
    frames = torch.from_numpy(syn.make_frames(16, 376, 1232, seed=77))
    kf = os.path.join(str(tmp_path), "kf")
    os.makedirs(os.path.join(kf, "rgb"))
    poses = torch.eye(4).flatten()[:12].repeat(16, 1)
    poses[:, 3] = torch.arange(16, dtype=torch.float32)       # x translation = keyframe index
    for i in range(16):
        torch.save(frames[i].byte(), os.path.join(kf, "rgb", "%06d.pth" % i))
    torch.save(poses, os.path.join(kf, "poses.pth"))
    # map weights left behind by an earlier session in the same directory: the reference always retrains and
    # overwrites them (neural_slam.py:160,305-352), so they must not be picked up
    torch.save(vsd, os.path.join(kf, "MappingVAE_weights.pth"))
    cwd = os.getcwd()
    os.chdir(str(tmp_path))   # mapping_loss.pth goes to the working directory, as in the reference
    try:
        torch.manual_seed(11)
        slam = NeuralSLAM(_Args(kf), odometry_weights=hsd, flow_weights=gsd, start_mode="mapping",
                          map_options={"num_epochs": 2, "miopen": False})
    finally:
        os.chdir(cwd)
    assert slam.mode() == "relocalization" and len(slam) == 16
    assert os.path.exists(os.path.join(kf, "MappingVAE_weights.pth"))
    assert list(torch.load(os.path.join(str(tmp_path), "mapping_loss.pth")).shape) == [2]
    # the torch module with the trained weights, eval mode, against the HIP embedding stored on the keyframes
    net = mapping.MappingVAENet()
    trained = torch.load(os.path.join(kf, "MappingVAE_weights.pth"))
    key = "mean_lin.weight" if "mean_lin.weight" in trained else next(k for k in trained if k.endswith("weight"))
    assert not torch.equal(trained[key].cpu(), torch.as_tensor(vsd[key]).cpu()), "stale map weights were reused"
    net.load_state_dict(trained)
    net = net.to(DEV).eval()
    with torch.no_grad():
        mu = net(frames[5:6].byte().float().to(DEV))[0]
    emb = slam[5].embedding
    scale = float(mu.abs().max())
    assert float((emb.to(DEV) - mu).abs().max()) < 2e-3 * max(scale, 1.0)
    init, refined, dist = slam(frames[5].byte().float())
    assert int(torch.argmin(dist)) == 5 and float(dist[5]) < 1e-3 * max(scale, 1.0)
    assert torch.equal(init, slam[5].pose) and float(init[0, 3]) == 5.0
