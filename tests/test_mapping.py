"""Map creation (`atdn_vslam_amd.mapping`, the reference's NeuralSLAM.__create_map) against `tests/golden/map.npz`:
what the reference's own loop produced on the same synthetic keyframes (tests/golden/make_golden_map.py)."""
import json
import os

import numpy as np
import pytest
import torch

from atdn_vslam_amd import mapping
from atdn_vslam_amd import synthetic as syn

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "map.npz"))


def _keyframes(tmp_path):
    base = str(tmp_path / "kf")
    os.makedirs(os.path.join(base, "rgb"))
    frames = syn.make_frames(int(G["nkey"]), int(G["hk"]), int(G["wk"]), seed=int(G["frames_seed"]))
    for i in range(frames.shape[0]):
        torch.save(torch.from_numpy(frames[i]).byte(), os.path.join(base, "rgb", "%06d.pth" % i))
    return base, frames


def _digest(sd):
    vals = [v.double().flatten() for v in sd.values()]
    return (np.array([float(v.sum()) for v in vals]), np.array([float(v.norm()) for v in vals]),
            np.stack([np.resize(v[:4].numpy(), 4) for v in vals]))


def test_state_dict_layout_is_the_references():
    keys = json.load(open(os.path.join(HERE, "golden", "state_keys.json")))["vae"]
    sd = mapping.MappingVAENet().state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == keys


def test_first_epochs_follow_the_reference_run(tmp_path):
    """Same seed, same keyframes, jitter off (the reference ran with its ColorJitter stubbed): the first epochs'
    losses and the checkpoint after them must be the reference's (same initial weights, same batch order, same
    optimiser and schedule)."""
    base, _ = _keyframes(tmp_path)
    n = int(G["snap_epoch"])
    torch.manual_seed(int(G["seed"]))
    net, losses = mapping.create_map(base, device="cpu", num_epochs=50, augment=False, stop_after=n,
                                     loss_file=str(tmp_path / "loss.pth"))
    assert len(losses) == n and not net.training
    np.testing.assert_allclose(losses, G["losses"][:n], rtol=1e-4)   # measured: 5e-8
    saved = torch.load(os.path.join(base, "MappingVAE_weights.pth"))
    sums, norms, heads = _digest(saved)
    np.testing.assert_allclose(norms, G["snap_norms"], rtol=1e-3, atol=1e-5)   # measured: bit-exact
    np.testing.assert_allclose(heads, G["snap_heads"], rtol=1e-2, atol=1e-3)
    assert list(torch.load(str(tmp_path / "loss.pth")).shape) == [n]


def test_needs_a_full_batch(tmp_path):
    base = str(tmp_path / "kf")
    os.makedirs(os.path.join(base, "rgb"))
    torch.save(torch.zeros(3, 192, 256, dtype=torch.uint8), os.path.join(base, "rgb", "000000.pth"))
    with pytest.raises(RuntimeError, match="at least 16 keyframes"):
        mapping.create_map(base, device="cpu", num_epochs=1)


def test_color_jitter_ranges_and_identity():
    g = torch.Generator().manual_seed(3)
    x = torch.from_numpy(syn.make_frames(2, 64, 96, seed=9))
    y = mapping.color_jitter(x, generator=g)
    assert y.shape == x.shape and float(y.min()) >= 0.0 and float(y.max()) <= 255.0
    assert float((y - x).abs().mean()) < 0.12 * 255.0
    z = mapping.color_jitter(x, brightness=0.0, saturation=0.0, hue=0.0, generator=g)
    assert float((z - x).abs().max()) < 1e-2
    # the reference's clamp (torchvision treats float images as 0..1): everything >= 1 saturates
    w = mapping.color_jitter(x, bound=1.0, generator=g)
    assert float(w.max()) <= 1.0


def test_gaussian_blur_is_normalised_and_symmetric():
    x = torch.zeros(1, 3, 9, 9)
    x[:, :, 4, 4] = 1.0
    y = mapping.gaussian_blur5(x)
    assert abs(float(y.sum()) - 3.0) < 1e-5
    assert torch.allclose(y, y.flip(-1)) and torch.allclose(y, y.transpose(-1, -2))
    assert float(y[0, 0, 4, 4]) > float(y[0, 0, 4, 5]) > float(y[0, 0, 4, 6]) > 0.0
