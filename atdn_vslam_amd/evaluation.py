"""Trajectory tail of the odometry path: KITTI pose files and trajectory error.

* `save_kitti_poses` writes what evaluate_odometry.py:84-99 writes (`np.savetxt` of `pose[:3,:].view(12)` rows);
  `load_kitti_poses` reads that format back (also the reference's shipped `atdn_vslam/eval/GT.txt`).
* `ate_rmse` is the absolute trajectory error the reference obtains through `evo` (eval/visualizer.py:85-91, APE on
  translations with SE(3) / Sim(3) Umeyama alignment); evo is not in the image, so the alignment is restated here.
"""
import numpy as np


def kitti_rows(poses):
    p = np.asarray(poses, dtype=np.float64)
    return p[:, :3, :].reshape(len(p), 12)


def save_kitti_poses(path, poses):
    """poses [T,4,4] (torch or numpy) -> text file, one 12-float row per pose."""
    np.savetxt(path, kitti_rows(np.asarray(poses)))


def load_kitti_poses(path):
    rows = np.loadtxt(path).reshape(-1, 12)
    out = np.tile(np.eye(4), (len(rows), 1, 1))
    out[:, :3, :] = rows.reshape(-1, 3, 4)
    return out


def umeyama(src, dst, with_scale):
    """Least-squares similarity (R, t, s) with dst ~ s R src + t (Umeyama 1991). src, dst [T,3]."""
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / len(src)
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    s = float(np.trace(np.diag(D) @ S) / xs.var(0).sum()) if with_scale else 1.0
    t = mu_d - s * R @ mu_s
    return R, t, s


def ate_rmse(pred, gt, align="se3"):
    """RMSE of the translation error after aligning `pred` to `gt`; align in {"none", "se3", "sim3"}."""
    p = np.asarray(pred, dtype=np.float64)[:, :3, 3]
    g = np.asarray(gt, dtype=np.float64)[:, :3, 3]
    if p.shape != g.shape:
        raise ValueError("trajectories differ in length")
    if align != "none":
        R, t, s = umeyama(p, g, with_scale=(align == "sim3"))
        p = (s * (R @ p.T)).T + t
    return float(np.sqrt(((p - g) ** 2).sum(1).mean()))
