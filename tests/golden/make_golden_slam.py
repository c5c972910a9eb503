"""Golden fixtures for the rows next to the hot path, produced by running the REFERENCE itself:

* `kalman.npz`   — forward/backward trajectory fusion of atdn_vslam/eval/kalman.py (its functions are executed from
                   the reference file with the argparse `main()` call cut off) on the first poses of the reference's
                   own data fixtures eval/GT.txt and eval/ATDN_prediction.txt;
* `keyframes.npz` — NeuralSLAM's keyframe decisions for a sequence of relative motions (neural_slam.py:268-283);
* `vae.npz`      — MappingVAE (localization/network.py) state-dict layout and encoder outputs on synthetic weights;
* `reloc.npz`    — NeuralSLAM started in "relocalization" mode on a synthetic keyframe directory: embedding
                   distances, initial and refined pose for a query frame (neural_slam.py:355-399).

Run only in the build container (needs /root/reference):   python tests/golden/make_golden_slam.py
Fixtures are numbers only; nothing of the reference's text is stored.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"

from make_golden import install_stubs  # noqa: E402


def kalman_fixture():
    from atdn_vslam.utils import transforms as T
    src = open(os.path.join(REF, "atdn_vslam/eval/kalman.py")).read()
    src = src.replace("from ..utils.transforms import matrix2euler, rel2abs", "")
    src = src[:src.rindex("main()")]  # the module ends with a bare call that parses sys.argv
    plt = types.ModuleType("matplotlib.pyplot")
    ns = {"matrix2euler": T.matrix2euler, "rel2abs": T.rel2abs}
    sys.modules.setdefault("matplotlib", types.ModuleType("matplotlib"))
    sys.modules["matplotlib.pyplot"] = plt
    exec(compile(src, "kalman_ref", "exec"), ns)

    n = 48
    real = np.loadtxt(os.path.join(REF, "atdn_vslam/eval/GT.txt"))[:n]
    forward = np.loadtxt(os.path.join(REF, "atdn_vslam/eval/ATDN_prediction.txt"))[:n]
    # a "backward run": the forward relative motions perturbed, accumulated from the last frame backwards
    r = np.random.RandomState(21)
    mat_f = torch.from_numpy(forward).view(n, 3, 4)
    rot_f, tr_f = ns["preprocess_poses_euler"](mat_f)
    rot_n = rot_f + torch.from_numpy(r.normal(0, 2e-3, rot_f.shape))
    tr_n = tr_f + torch.from_numpy(r.normal(0, 3e-2, tr_f.shape))
    cur = torch.eye(4, dtype=torch.float64)
    back = [cur[:3].clone()]
    for i in reversed(range(n - 1)):
        cur = cur @ torch.inverse(T.transform(rot_n[i], tr_n[i]).double())
        back.append(cur[:3].clone())
    backward = torch.stack(back).reshape(n, 12).numpy()

    # what determine_std / process_kalman compute, without the file and plot side effects
    mat_r = torch.from_numpy(real).view(n, 3, 4)
    mat_b = torch.from_numpy(backward).view(n, 3, 4)
    h_ext = torch.tensor([0, 0, 0, 1], dtype=mat_b.dtype).view(1, 1, 4).repeat(n, 1, 1)
    mat_b2 = torch.cat([mat_b, h_ext], dim=1)
    inv = torch.inverse(mat_b2[-1])
    mat_bt = torch.flip(torch.stack([torch.matmul(inv, m)[:3, :] for m in mat_b2]), dims=[0])
    rot_b, tr_b = ns["preprocess_poses_euler"](mat_bt)
    rot_r, tr_r = ns["preprocess_poses_euler"](mat_r)
    std = [(rot_f - rot_r).std(0), (rot_b - rot_r).std(0), (tr_f - tr_r).std(0), (tr_b - tr_r).std(0)]
    opt_rot = ns["kalman"](rot_f, rot_b, std[0], std[1])
    opt_tr = ns["kalman"](tr_f, tr_b, std[2], std[3])
    opt = T.rel2abs(opt_rot, opt_tr)
    np.savez_compressed(os.path.join(HERE, "kalman.npz"), real=real, forward=forward, backward=backward,
                        backward_transformed=mat_bt.reshape(n, 12).numpy(),
                        rot_f=rot_f.numpy(), tr_f=tr_f.numpy(), rot_b=rot_b.numpy(), tr_b=tr_b.numpy(),
                        std=np.stack([s.numpy() for s in std]), opt_rot=opt_rot[0].numpy(), opt_tr=opt_tr[0].numpy(),
                        fused_rows=opt[:, :3, :].reshape(-1, 12).numpy())


def slam_fixtures():
    from atdn_vslam.utils import transforms as T
    from atdn_vslam.utils.arguments import Arguments
    from atdn_vslam.localization.network import MappingVAE
    from atdn_vslam_amd import synthetic as syn

    vae = MappingVAE().eval()
    keys = json.load(open(os.path.join(HERE, "state_keys.json")))
    keys["vae"] = [[k, list(v.shape)] for k, v in vae.state_dict().items()]
    with open(os.path.join(HERE, "state_keys.json"), "w") as f:
        json.dump(keys, f)
    # the spec covers encoder + mean_lin (what relocalisation uses); the decoder keeps its constructor values
    vsd = dict(vae.state_dict())
    vsd.update(syn.to_torch(syn.make_vae_state(seed=2)))
    vae.load_state_dict(vsd)
    frames = torch.from_numpy(syn.make_frames(5, 376, 1232, seed=12))
    with torch.no_grad():
        x = vae.normalization(frames[:2])
        taps = {}
        for i, layer in enumerate(vae.encoder):
            x = layer(x)
            taps["enc%d" % i] = x[:, :, ::max(1, x.shape[2] // 12), ::max(1, x.shape[3] // 16)].numpy()
        mu = vae(frames[:2])[0]
    np.savez_compressed(os.path.join(HERE, "vae.npz"), seed_weights=2, seed_frames=12, mu=mu.numpy(), **taps)

    gsd = syn.to_torch(syn.make_gma_state(seed=1))
    hsd = syn.to_torch(syn.make_clvo_state(seed=1))
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            os.makedirs("atdn_vslam/checkpoints")
            torch.save({"module." + k: v for k, v in gsd.items()}, "atdn_vslam/checkpoints/gma-kitti.pth")
            torch.save(hsd, "odo.pth")
            from atdn_vslam.slam_framework.neural_slam import NeuralSLAM
            args = Arguments()
            args.device = "cpu"
            args.keyframes_path = os.path.join(tmp, "kf")
            args.data_path = tmp

            # ---- keyframe policy: the private decision method fed with a scripted sequence of relative motions
            slam = NeuralSLAM(args, odometry_weights="odo.pth")
            r = np.random.RandomState(31)
            rots = r.uniform(-0.05, 0.05, (60, 3)).astype(np.float32)
            trs = r.uniform(-1.0, 2.5, (60, 3)).astype(np.float32)
            rots[20:24] *= 4.0  # a sharp turn: the rotation threshold fires
            decisions = []
            for i in range(60):
                m = T.transform(torch.from_numpy(rots[i]), torch.from_numpy(trs[i]))
                decisions.append(bool(slam._NeuralSLAM__decide_keyframe(m)))
            np.savez_compressed(os.path.join(HERE, "keyframes.npz"), rots=rots, trs=trs,
                                decisions=np.array(decisions, dtype=np.uint8))

            # ---- relocalisation on a synthetic keyframe directory
            kf = args.keyframes_path
            os.makedirs(os.path.join(kf, "rgb"), exist_ok=True)
            for f in os.listdir(os.path.join(kf, "rgb")):
                os.remove(os.path.join(kf, "rgb", f))
            kposes = []
            cur = torch.eye(4)
            pr = np.random.RandomState(32)
            for i in range(3):
                torch.save(frames[i].byte(), os.path.join(kf, "rgb", "%06d.pth" % i))
                kposes.append(cur.flatten()[:12].clone())
                cur = cur @ T.transform(torch.from_numpy(pr.uniform(-0.1, 0.1, 3).astype(np.float32)),
                                        torch.from_numpy(pr.uniform(-2, 8, 3).astype(np.float32)))
            torch.save(torch.stack(kposes), os.path.join(kf, "poses.pth"))
            torch.save(vsd, os.path.join(kf, "MappingVAE_weights.pth"))
            slam = NeuralSLAM(args, odometry_weights="odo.pth", start_mode="relocalization")
            assert slam.mode() == "relocalization" and len(slam) == 3
            out = {}
            for name, q in (("near1", frames[1].byte().float()), ("new", frames[4].byte().float())):
                init, refined, dist = slam(q)
                out[name + "_initial"] = init.numpy()
                out[name + "_refined"] = refined.numpy()
                out[name + "_distances"] = dist.numpy()
            np.savez_compressed(os.path.join(HERE, "reloc.npz"), seed_weights_gma=1, seed_weights_clvo=1,
                                seed_weights_vae=2, seed_frames=12, keyframe_poses=torch.stack(kposes).numpy(), **out)
        finally:
            os.chdir(cwd)


def main():
    install_stubs()
    sys.path.insert(0, os.path.join(REF, "GMA-1.0.0-py3-none-any.whl"))
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    kalman_fixture()
    slam_fixtures()
    print("slam/eval golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
