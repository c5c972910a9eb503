"""KITTI pose files and trajectory error (the eval tail of evaluate_odometry.py / eval/visualizer.py)."""
import os

import numpy as np

from atdn_vslam_amd import evaluation as ev
from oracle import pose_ref


def _trajectory(n, seed):
    r = np.random.RandomState(seed)
    return pose_ref.rel2abs(r.uniform(-0.05, 0.05, (n, 3)), r.uniform(-1.0, 1.0, (n, 3)) + np.array([0, 0, 1.0]))


def test_kitti_file_round_trip(tmp_path, golden_dir):
    g = np.load(os.path.join(golden_dir, "pose.npz"))
    path = os.path.join(str(tmp_path), "traj.txt")
    ev.save_kitti_poses(path, g["rel2abs"])
    rows = np.loadtxt(path)
    np.testing.assert_allclose(rows, g["kitti_rows"], rtol=0, atol=1e-15)   # the reference's own row layout
    back = ev.load_kitti_poses(path)
    np.testing.assert_allclose(back, g["rel2abs"], rtol=0, atol=1e-15)
    assert np.all(back[:, 3] == np.array([0, 0, 0, 1.0]))


def test_ate_alignment_properties():
    gt = _trajectory(300, 1)
    # a rigidly moved copy has zero SE(3)-aligned error, a scaled one zero Sim(3)-aligned error
    Rz = pose_ref.euler2matrix([0.3, -0.2, 0.5])
    moved = gt.copy()
    moved[:, :3, 3] = (Rz @ gt[:, :3, 3].T).T + np.array([5.0, -2.0, 1.0])
    assert ev.ate_rmse(moved, gt, "se3") < 1e-9 and ev.ate_rmse(moved, gt, "none") > 1.0
    scaled = moved.copy()
    scaled[:, :3, 3] *= 0.9
    assert ev.ate_rmse(scaled, gt, "sim3") < 1e-9 and ev.ate_rmse(scaled, gt, "se3") > 0.1
    noisy = gt.copy()
    noisy[:, :3, 3] += np.random.RandomState(2).normal(0, 0.1, (301, 3))
    e = ev.ate_rmse(noisy, gt, "se3")
    assert 0.1 < e < 0.2
    assert abs(ev.ate_rmse(gt, gt, "none")) == 0.0
