"""CPU ORACLE (test infrastructure, not product code) — pose algebra.

numpy restatement of atdn_vslam/utils/transforms.py: `euler2matrix` ("yxz",
:54-94, matrix at :79-81), `matrix2euler` (:26-51), `transform` (:97-119),
`rel2abs` (:147-170, float64, identity prepended) and of the reference's
frame padder `InputPadder` (whl:GMA/core/utils/utils.py:8-25, 'sintel' mode).

Parity pin: tests/golden/pose.npz (outputs of the imported reference).
"""
import numpy as np


def euler2matrix(r, dtype=np.float64):
    r = np.asarray(r, dtype=dtype)
    c1, c2, c3 = np.cos(r[0]), np.cos(r[1]), np.cos(r[2])
    s1, s2, s3 = np.sin(r[0]), np.sin(r[1]), np.sin(r[2])
    return np.array([[c1 * c3 + s1 * s2 * s3, c3 * s1 * s2 - c1 * s3, c2 * s1],
                     [c2 * s3, c2 * c3, -s2],
                     [c1 * s2 * s3 - c3 * s1, c1 * c3 * s2 + s1 * s3, c1 * c2]], dtype=dtype)


def matrix2euler(R):
    R = np.asarray(R)
    a = np.arctan2(R[0, 2], R[2, 2])
    b = np.arctan2(-R[1, 2], np.sqrt(1 - R[1, 2] ** 2))
    g = np.arctan2(R[1, 0], R[1, 1])
    return np.array([a, b, g], dtype=R.dtype)


def transform(rot, tr, dtype=np.float64):
    m = np.eye(4, dtype=dtype)
    m[:3, :3] = euler2matrix(rot, dtype)
    m[:3, 3] = np.asarray(tr, dtype=dtype)
    return m


def rel2abs(rotations, translations):
    """[T,3],[T,3] → [T+1,4,4] float64 absolute poses."""
    poses = [np.eye(4)]
    for r, t in zip(rotations, translations):
        poses.append(poses[-1] @ transform(np.asarray(r).reshape(3), np.asarray(t).reshape(3)))
    return np.stack(poses)


def kitti_rows(poses):
    """evaluate_odometry.py:86-90: pose[:3,:].view(12) per row."""
    return np.asarray(poses)[:, :3, :].reshape(len(poses), 12)


def pad_amounts(ht, wd):
    """[left, right, top, bottom] of InputPadder(mode='sintel')."""
    ph = (((ht // 8) + 1) * 8 - ht) % 8
    pw = (((wd // 8) + 1) * 8 - wd) % 8
    return [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]
