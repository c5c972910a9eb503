"""Trajectory tail of the odometry path: KITTI pose files and trajectory error.

* `save_kitti_poses` writes what evaluate_odometry.py:84-99 writes (`np.savetxt` of `pose[:3,:].view(12)` rows);
  `load_kitti_poses` reads that format back (also the reference's shipped `atdn_vslam/eval/GT.txt`).
* `ate_rmse` is the absolute trajectory error the reference obtains through `evo` (eval/visualizer.py:85-91, APE on
  translations with SE(3) / Sim(3) Umeyama alignment); evo is not in the image, so the alignment is restated here.
  `rpe` is evo's relative pose error for a fixed frame delta.
* `relative_motions`, `reverse_backward_run`, `motion_std`, `kalman_fuse`, `fuse_forward_backward` are the
  forward/backward trajectory fusion of eval/kalman.py (45-88, 91-128): per-axis inverse-variance weighting of the
  relative motions of a forward and a time-reversed run, re-integrated with rel2abs. All float64 on the host.
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


def _hom(poses):
    p = np.asarray(poses, dtype=np.float64)
    if p.ndim == 2 and p.shape[1] == 12:
        p = p.reshape(-1, 3, 4)
    if p.shape[1] == 3:
        out = np.tile(np.eye(4), (len(p), 1, 1))
        out[:, :3, :] = p
        return out
    return p


def _euler_yxz(R):
    """matrix2euler (utils/transforms.py:41-44), float64."""
    return np.array([np.arctan2(R[0, 2], R[2, 2]), np.arctan2(-R[1, 2], np.sqrt(1 - R[1, 2] ** 2)),
                     np.arctan2(R[1, 0], R[1, 1])])


def _matrix_yxz(r):
    """euler2matrix, yxz convention (utils/transforms.py:72-75), float64."""
    c1, c2, c3 = np.cos(r)
    s1, s2, s3 = np.sin(r)
    return np.array([[c1 * c3 + s1 * s2 * s3, c3 * s1 * s2 - c1 * s3, c2 * s1],
                     [c2 * s3, c2 * c3, -s2],
                     [c1 * s2 * s3 - c3 * s1, c1 * c3 * s2 + s1 * s3, c1 * c2]])


def relative_motions(poses):
    """Absolute poses [T,3|4,4] (or KITTI rows [T,12]) -> (euler [T-1,3], translation [T-1,3]) of
    inverse(P_i) @ P_{i+1} (kalman.py:9-28)."""
    p = _hom(poses)
    rot, tr = [], []
    for i in range(len(p) - 1):
        d = np.linalg.inv(p[i]) @ p[i + 1]
        rot.append(_euler_yxz(d[:3, :3]))
        tr.append(d[:3, 3])
    return np.array(rot).reshape(-1, 3), np.array(tr).reshape(-1, 3)


def reverse_backward_run(poses):
    """Trajectory estimated on the time-reversed sequence -> the same motion in forward time, starting at identity:
    inverse(last) @ P_i for every pose, order flipped (kalman.py:64-69). Returns [T,4,4]."""
    p = _hom(poses)
    inv = np.linalg.inv(p[-1])
    return np.stack([inv @ m for m in p])[::-1].copy()


def integrate_motions(rot, tr):
    """rel2abs in float64 (utils/transforms.py:147-170): [T,3], [T,3] -> [T+1,4,4], identity first."""
    out = [np.eye(4)]
    for r, t in zip(np.asarray(rot, dtype=np.float64), np.asarray(tr, dtype=np.float64)):
        m = np.eye(4)
        m[:3, :3] = _matrix_yxz(r)
        m[:3, 3] = t
        out.append(out[-1] @ m)
    return np.stack(out)


def kalman_fuse(x1, x2, s1, s2):
    """Inverse-variance fusion of two estimates with standard deviations s1, s2 (kalman.py:45-50)."""
    v1, v2 = np.asarray(s1, dtype=np.float64) ** 2, np.asarray(s2, dtype=np.float64) ** 2
    return (np.asarray(x1) * v2 + np.asarray(x2) * v1) / (v1 + v2)


def motion_std(real, forward, backward):
    """Per-axis std (unbiased, as torch.std) of the relative-motion errors of the forward and of the (already
    time-reversed) backward run against ground truth: [std_rot_f, std_rot_b, std_tr_f, std_tr_b] (kalman.py:91-128)."""
    rr, tr = relative_motions(real)
    rf, tf = relative_motions(forward)
    rb, tb = relative_motions(backward)
    return [np.std(rf - rr, axis=0, ddof=1), np.std(rb - rr, axis=0, ddof=1),
            np.std(tf - tr, axis=0, ddof=1), np.std(tb - tr, axis=0, ddof=1)]


def fuse_forward_backward(forward, backward_raw, std):
    """process_kalman (kalman.py:53-88): `forward` and `backward_raw` are pose files of the two runs (the backward
    one as estimated on the reversed sequence); returns the fused absolute trajectory [T,4,4]."""
    back = reverse_backward_run(backward_raw)
    rf, tf = relative_motions(forward)
    rb, tb = relative_motions(back)
    return integrate_motions(kalman_fuse(rf, rb, std[0], std[1]), kalman_fuse(tf, tb, std[2], std[3]))


def rpe(pred, gt, delta=1):
    """Relative pose error over a fixed frame delta (evo's RPE, all pairs): returns (translation RMSE in the units of
    the poses, rotation-angle RMSE in radians) of inverse(Q_i^-1 Q_{i+d}) @ (P_i^-1 P_{i+d})."""
    p, g = _hom(pred), _hom(gt)
    if p.shape != g.shape:
        raise ValueError("trajectories differ in length")
    if len(p) <= delta:
        raise ValueError("trajectory shorter than the frame delta")
    te, re = [], []
    for i in range(len(p) - delta):
        dp = np.linalg.inv(p[i]) @ p[i + delta]
        dg = np.linalg.inv(g[i]) @ g[i + delta]
        e = np.linalg.inv(dg) @ dp
        te.append(np.linalg.norm(e[:3, 3]))
        re.append(np.arccos(np.clip((np.trace(e[:3, :3]) - 1.0) / 2.0, -1.0, 1.0)))
    return float(np.sqrt(np.mean(np.square(te)))), float(np.sqrt(np.mean(np.square(re))))
