"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/atdn_hip.h declares,
and its host-only entry points (pose algebra, argument validation) behave like the reference."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from atdn_vslam_amd import _lib, transforms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "atdn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(atdn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 20
    handle = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), "libatdn_hip.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert _lib.lib().atdn_version() >= 100


def test_pose_algebra_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "pose.npz"))
    for i in range(16):
        m = transforms.transform(torch.from_numpy(g["rots"][i]), torch.from_numpy(g["trs"][i]))
        np.testing.assert_allclose(m.numpy(), g["transform"][i], rtol=0, atol=1e-6)
        e = transforms.matrix2euler(torch.from_numpy(g["transform"][i][:3, :3]))
        np.testing.assert_allclose(e.numpy(), g["euler"][i], rtol=0, atol=1e-6)
    rots = [torch.from_numpy(g["rots"][i:i + 1]) for i in range(16)]
    trs = [torch.from_numpy(g["trs"][i:i + 1]) for i in range(16)]
    absolute = transforms.rel2abs(rots, trs)
    assert absolute.dtype == torch.float64 and tuple(absolute.shape) == (17, 4, 4)
    np.testing.assert_allclose(absolute.numpy(), g["rel2abs"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(transforms.kitti_rows(absolute).numpy(), g["kitti_rows"], rtol=0, atol=1e-12)
    # empty sequence: identity only
    assert tuple(transforms.rel2abs(np.zeros((0, 3)), np.zeros((0, 3))).shape) == (1, 4, 4)
    # float32 running pose as NeuralSLAM keeps it
    pose = torch.eye(4)
    for i in range(16):
        pose = transforms.accumulate(pose, g["rots"][i], g["trs"][i])
    np.testing.assert_allclose(pose.numpy(), g["rel2abs"][16], rtol=0, atol=2e-5)


def test_padder_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "pose.npz"))
    for (h, w), pad in zip(g["pad_dims"], g["pads"]):
        p = transforms.InputPadder((3, int(h), int(w)))
        assert p._pad == list(pad)
        x = torch.arange(3 * int(h) * int(w), dtype=torch.float32).view(1, 3, int(h), int(w))
        y = p.pad(x)[0]
        assert y.shape[-2] % 8 == 0 and y.shape[-1] % 8 == 0
        assert torch.equal(p.unpad(y), x)


def test_argument_errors_are_reported():
    L = _lib.lib()
    h = C.c_void_p()
    assert L.atdn_gma_create(C.byref(h), 375, 1232, 1, 0) != 0  # not a multiple of 8
    assert b"multiple of 8" in L.atdn_last_error()
    assert L.atdn_clvo_create(C.byref(h), 160, 512, 1) != 0  # cannot reduce to 16x4x13
    assert b"16x4x13" in L.atdn_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(L.atdn_pose_rel2abs(None, None, 3, None))
