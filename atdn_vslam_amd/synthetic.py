"""Deterministic synthetic checkpoints and KITTI-shaped frames.

Both reference checkpoints are absent (`/root/reference/.MISSING_LARGE_BLOBS`),
so parity and benchmarks run on seeded synthetic weights of the exact
state-dict layout (weights_spec.py) and on synthetic frames of KITTI geometry
(3x376x1241, uint8-valued float32, RGB 0..255 — what `NeuralSLAM.__call__`
receives, neural_slam.py:196-199).

Generators use `numpy.random.RandomState` (frozen legacy stream → identical
values on every box) and never torch RNG.
"""
import zlib

import numpy as np

from .weights_spec import F32, clvo_state_spec, gma_state_spec, vae_state_spec


def _rs(seed, key):
    return np.random.RandomState((seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF)


def _fill(key, shape, seed, gain):
    r = _rs(seed, key)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.array(100, dtype=np.int64)
    if leaf == "rel_ind":
        n = shape[0]
        d = np.arange(n)[None, :] - np.arange(n)[:, None]
        return (d + n - 1).astype(np.int64)
    if leaf == "running_var":
        return r.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == "running_mean":
        return r.uniform(-0.2, 0.2, shape).astype(np.float32)
    if leaf == "gamma":
        return np.full(shape, 0.5, dtype=np.float32)
    if len(shape) == 1:
        if leaf == "weight":  # norm scale
            return r.uniform(0.5, 1.5, shape).astype(np.float32)
        return r.uniform(-0.1, 0.1, shape).astype(np.float32)  # biases / norm shift
    fan_in = int(np.prod(shape[1:]))
    a = gain * np.sqrt(3.0 / fan_in)
    return r.uniform(-a, a, shape).astype(np.float32)


def make_gma_state(seed=0, prefix=""):
    """Synthetic RAFTGMA state (numpy arrays). `prefix="module."` mimics the
    DataParallel checkpoint the reference loads (neural_slam.py:51-52)."""
    out = {}
    for key, (shape, _) in gma_state_spec().items():
        out[prefix + key] = _fill(key, shape, seed, gain=1.4)
    # norm3 and downsample.1 are one module in the reference (extractor.py:44-45)
    for key in list(out):
        if ".downsample.1." in key:
            out[key] = out[key.replace(".downsample.1.", ".norm3.")].copy()
    # Trained checkpoints keep activations O(1)-O(10); plain fan-in scaling does not (features reach the
    # hundreds and every sigmoid/softmax saturates, which amplifies rounding noise ~1000x). These per-layer
    # factors bring the synthetic network to trained-like statistics: |fmap| ~ 4, |corr| ~ 25, |inp| ~ 6,
    # motion features O(1), ~0.5 px flow update per iteration.
    for frag, fac in _GMA_RESCALE:
        for key in out:
            if frag in key and (key.endswith(".weight") or key.endswith(".bias")) and "norm" not in key:
                out[key] = (out[key] * fac).astype(np.float32)
    return out


_GMA_RESCALE = (
    ("fnet.conv2.", 0.25),
    ("cnet.conv2.", 0.1),
    ("update_block.encoder.convc1.", 0.1),
    ("update_block.encoder.convf1.", 0.2),
    ("update_block.flow_head.conv2.", 0.5),
)


def make_clvo_state(seed=0):
    """Synthetic ATDNVO state (numpy arrays)."""
    out = {}
    for key, (shape, _) in clvo_state_spec().items():
        out[key] = _fill(key, shape, seed + 17, gain=1.0)
    out["encoder_CNN.0.weight"] = _rs(seed, "dw").uniform(0.5, 1.5, (2, 1, 1, 1)).astype(np.float32)
    return out


def make_vae_state(seed=0):
    """Synthetic MappingVAE encoder + mean_lin state (numpy arrays)."""
    out = {}
    for key, (shape, _) in vae_state_spec().items():
        out[key] = _fill(key, shape, seed + 29, gain=1.2)
    return out


def to_torch(state):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}


def _value_noise(r, h, w, cell):
    gh, gw = h // cell + 3, w // cell + 3
    g = r.uniform(0.0, 1.0, (3, gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = ys.astype(np.int64)
    x0 = xs.astype(np.int64)
    fy = (ys - y0)[None, :, None]
    fx = (xs - x0)[None, None, :]
    a = g[:, y0][:, :, x0]
    b = g[:, y0][:, :, x0 + 1]
    c = g[:, y0 + 1][:, :, x0]
    d = g[:, y0 + 1][:, :, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_canvas(seed, h, w):
    r = np.random.RandomState(seed & 0x7FFFFFFF)
    img = 0.55 * _value_noise(r, h, w, 48) + 0.30 * _value_noise(r, h, w, 12) + 0.15 * _value_noise(r, h, w, 4)
    return img  # [3,h,w] in 0..1


def make_frames(n_frames, height=376, width=1241, seed=0, max_shift=6):
    """`n_frames` consecutive frames [n,3,H,W] float32 with integer values 0..255.
    Frame t is a window of one textured canvas moved by a seeded random walk of
    at most `max_shift` px per step, plus +-2 LSB noise, so the flow between
    consecutive frames is non-trivial and stays inside the lookup radius."""
    margin = max_shift * n_frames + 8
    canvas = make_canvas(seed, height + 2 * margin, width + 2 * margin)
    r = np.random.RandomState((seed + 991) & 0x7FFFFFFF)
    oy, ox = margin, margin
    out = np.empty((n_frames, 3, height, width), dtype=np.float32)
    for t in range(n_frames):
        if t:
            oy += int(r.randint(-max_shift // 2, max_shift // 2 + 1))
            ox += int(r.randint(-max_shift, max_shift + 1))
        win = canvas[:, oy:oy + height, ox:ox + width] * 255.0
        win = win + r.randint(-2, 3, win.shape)
        out[t] = np.clip(np.rint(win), 0, 255)
    return out


def make_flow(batch, height=376, width=1232, seed=0):
    """Synthetic optical flow [B,2,H,W] with KITTI-like anisotropic scale for
    head-only (evaluate_odometry.py-style) runs."""
    r = np.random.RandomState((seed + 4242) & 0x7FFFFFFF)
    base = np.stack([_value_noise(r, height, width, 64)[:2] for _ in range(batch)])
    flow = (base - 0.5) * np.array([120.0, 36.0], dtype=np.float32).reshape(1, 2, 1, 1)
    flow = flow + r.normal(0.0, 0.5, flow.shape)
    return flow.astype(np.float32)
