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


def _prototypes():
    """{name: (return type, [parameter types])} parsed from include/atdn_hip.h (comments stripped; types as written, with the
    parameter names removed)."""
    text = open(os.path.join(ROOT, "include", "atdn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for ret, name, params in re.findall(r"([A-Za-z_][A-Za-z0-9_ ]*?\**)\s*\b(atdn_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        plist = []
        for prm in [x.strip() for x in " ".join(params.split()).split(",")]:
            if prm in ("void", ""):
                continue
            m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", prm)      # the last identifier is the parameter's name
            plist.append(m.group(1).replace(" ", ""))
        protos[name] = (ret.replace(" ", ""), plist)
    return protos


def _ctype_kind(t):
    """A ctypes argtype / restype as the C type class it can bind: 'ptr' or the scalar's C name."""
    if t is None:
        return "void"
    if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or getattr(t, "_type_", None) == "P":
        return "ptr"
    return {C.c_int: "int", C.c_long: "long", C.c_float: "float", C.c_double: "double", C.c_size_t: "size_t",
            C.c_int64: "int64_t"}[t]


def _c_kind(t):
    if t.endswith("*"):
        return "ptr"
    t = t.replace("const", "")
    return {"int": "int", "long": "long", "float": "float", "double": "double", "size_t": "size_t", "int64_t": "int64_t",
            "void": "void"}[t]


def test_ctypes_argument_lists_match_the_header_prototypes():
    """names alone do not catch a drifted `argtypes` (it would only surface as a crash in a GPU test). Every
    prototype of include/atdn_hip.h is parsed and compared with _lib.SIGNATURES parameter by parameter: count, pointer vs
    scalar, and the scalar's C type; the return type too."""
    protos = _prototypes()
    assert sorted(protos) == sorted(_lib.SIGNATURES)
    for name, (ret, params) in sorted(protos.items()):
        res, args = _lib.SIGNATURES[name]
        assert len(args) == len(params), "%s: header has %d parameters, ctypes table %d" % (name, len(params), len(args))
        for i, (ct, pt) in enumerate(zip(args, params)):
            a, b = _ctype_kind(ct), _c_kind(pt)
            assert a == b or {a, b} <= {"long", "int64_t"}, "%s: parameter %d is `%s` in the header, %s in the ctypes table" % (name, i, pt, ct)
        # (on LP64, size_t / long / int64_t all bind 8-byte integers; ctypes aliases c_int64 to c_long there)
        want = _c_kind(ret)
        got = _ctype_kind(res)
        assert got == want or {got, want} <= {"long", "int64_t"}, "%s: returns `%s` in the header, %s in the table" % (name, ret, res)


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
