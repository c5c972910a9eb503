"""Pose algebra of the odometry path (atdn_vslam/utils/transforms.py) and the frame padder
(whl:GMA/core/utils/utils.py:8-25), host side of libatdn_hip."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _np32(x):
    if isinstance(x, torch.Tensor):
        x = x.detach().to("cpu")
        x = x.numpy()
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def transform(rot, tr):
    """transform(rot, tr) -> 4x4 float32 CPU tensor (transforms.py:97-119, Euler "yxz")."""
    r, t = _np32(rot).reshape(3), _np32(tr).reshape(3)
    out = np.empty(16, dtype=np.float32)
    _lib.check(_lib.lib().atdn_pose_transform_f32(r.ctypes.data, t.ctypes.data, out.ctypes.data))
    return torch.from_numpy(out.reshape(4, 4))


def rel2abs(rotations, translations):
    """rel2abs -> [T+1,4,4] float64 CPU tensor, identity first (transforms.py:147-170). Accepts the
    reference's lists of [1,3] tensors or [T,3] arrays."""
    if isinstance(rotations, (list, tuple)):
        rotations = np.stack([_np32(r).reshape(3) for r in rotations]) if len(rotations) else np.zeros((0, 3))
        translations = np.stack([_np32(t).reshape(3) for t in translations]) if len(translations) else np.zeros((0, 3))
    r, t = _np32(rotations).reshape(-1, 3), _np32(translations).reshape(-1, 3)
    if r.shape != t.shape:
        raise RuntimeError("rotations and translations differ in length")
    out = np.empty((r.shape[0] + 1, 4, 4), dtype=np.float64)
    _lib.check(_lib.lib().atdn_pose_rel2abs(r.ctypes.data, t.ctypes.data, r.shape[0], out.ctypes.data))
    return torch.from_numpy(out)


def accumulate(pose, rot, tr):
    """pose @ transform(rot, tr) in float32, as NeuralSLAM keeps its running pose (neural_slam.py:204-207)."""
    p = _np32(pose).reshape(16).copy()
    r, t = _np32(rot).reshape(3), _np32(tr).reshape(3)
    _lib.check(_lib.lib().atdn_pose_accumulate_f32(p.ctypes.data, r.ctypes.data, t.ctypes.data))
    return torch.from_numpy(p.reshape(4, 4))


def matrix2euler(R):
    """yxz Euler angles of a rotation matrix (transforms.py:41-44)."""
    R = torch.as_tensor(R)
    a = torch.atan2(R[0, 2], R[2, 2])
    b = torch.atan2(-R[1, 2], torch.sqrt(1 - R[1, 2] ** 2))
    g = torch.atan2(R[1, 0], R[1, 1])
    return torch.stack([a, b, g])


def kitti_rows(poses):
    """[T,4,4] -> [T,12] rows of the KITTI pose format (evaluate_odometry.py:86-90)."""
    p = torch.as_tensor(poses)
    return p[:, :3, :].reshape(p.shape[0], 12)


class InputPadder:
    """Replicate-pads frames to multiples of 8 ('sintel' mode splits the padding on both sides)."""

    def __init__(self, dims, mode="sintel"):
        self.ht, self.wd = dims[-2:]
        ph = (((self.ht // 8) + 1) * 8 - self.ht) % 8
        pw = (((self.wd // 8) + 1) * 8 - self.wd) % 8
        if mode == "sintel":
            self._pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]
        else:
            self._pad = [pw // 2, pw - pw // 2, 0, ph]

    def pad(self, *inputs):
        if not any(self._pad):  # 376x1232: nothing to pad (neural_slam.py:54) — skip the copy
            return list(inputs)
        return [self._pad_one(x) for x in inputs]

    def _pad_one(self, x):
        """F.pad(x, pad, mode="replicate") (utils.py:19-20). Device tensors go through libatdn_hip's pad kernel; host
        tensors (the reference pads on whatever device the frame is on) use torch."""
        if not x.is_cuda:
            return torch.nn.functional.pad(x, self._pad, mode="replicate")
        import ctypes as C
        from . import _lib
        l, r, t, b = self._pad
        src = x.float().contiguous()
        H, W = src.shape[-2:]
        out = torch.empty(tuple(src.shape[:-2]) + (H + t + b, W + l + r), dtype=torch.float32, device=src.device)
        planes = int(src.numel() // (H * W))
        with torch.cuda.device(src.device):
            _lib.check(_lib.lib().atdn_pad_frames(C.c_void_p(src.data_ptr()), planes, H, W, l, r, t, b,
                                                  C.c_void_p(out.data_ptr()),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        return x[..., self._pad[2]:ht - self._pad[3], self._pad[0]:wd - self._pad[1]]
