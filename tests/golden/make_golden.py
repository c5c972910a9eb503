"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Run only in the build container (needs /root/reference; it never travels):

    python tests/golden/make_golden.py

It imports the reference's own modules — the GMA wheel through zipimport and
`atdn_vslam` from /root/reference — with stubs for the two packages the image
lacks (torchvision, cv2), feeds them the seeded synthetic checkpoints/frames of
`atdn_vslam_amd.synthetic`, and stores inputs-by-seed + expected outputs.
Nothing of the reference's text is stored: fixtures are numbers only.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def install_stubs():
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")

    class Normalize:  # torchvision.transforms.Normalize on tensors: (x - mean) / std per channel
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
            self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean.to(x.device)) / self.std.to(x.device)

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    def resize(img, size, *a, **k):  # torchvision tensor resize: bilinear, antialias
        lead = img.dim() == 3
        x = img[None] if lead else img
        y = F.interpolate(x, size=list(size), mode="bilinear", align_corners=False, antialias=True)
        return y[0] if lead else y

    tvt.Normalize, tvt.Compose, tvt.Resize, tvt.ColorJitter = Normalize, Compose, object, object
    tvf.resize = resize
    tvt.functional = tvf
    tv.transforms = tvt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf,
                        "cv2": types.ModuleType("cv2")})


def main():
    install_stubs()
    sys.path.insert(0, os.path.join(REF, "GMA-1.0.0-py3-none-any.whl"))
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from GMA.core.network import RAFTGMA
    from GMA.core.corr import CorrBlock
    from GMA.core.utils.utils import InputPadder, coords_grid
    from atdn_vslam.utils.gma_parameters import GMA_Parameters
    from atdn_vslam.odometry.network import ATDNVO
    from atdn_vslam.utils import transforms as T

    from atdn_vslam_amd import synthetic as syn

    # ---------------------------------------------------------------- state-dict layouts
    gma = RAFTGMA(GMA_Parameters()).eval()
    head = ATDNVO().eval()
    keys = {"gma": [[k, list(v.shape)] for k, v in gma.state_dict().items()],
            "clvo": [[k, list(v.shape)] for k, v in head.state_dict().items()]}
    with open(os.path.join(HERE, "state_keys.json"), "w") as f:
        json.dump(keys, f)

    gsd = syn.to_torch(syn.make_gma_state(seed=1))
    gma.load_state_dict(gsd)
    hsd = syn.to_torch(syn.make_clvo_state(seed=1))
    head.load_state_dict(hsd)

    with torch.no_grad():
        # ------------------------------------------------------------ C1: 160x512, 8 iterations, per-stage taps
        fr = torch.from_numpy(syn.make_frames(2, 160, 512, seed=3))
        im1, im2 = fr[0:1], fr[1:2]
        flow_low, flow_up = gma(im1, im2, iters=8, test_mode=True)
        a = 2 * (im1 / 255.0) - 1.0
        b = 2 * (im2 / 255.0) - 1.0
        fmap1, fmap2 = gma.fnet([a, b])
        cb = CorrBlock(fmap1.float(), fmap2.float(), radius=4)
        cnet = gma.cnet(a)
        net, inp = torch.split(cnet, [128, 128], dim=1)
        net, inp = torch.tanh(net), torch.relu(inp)
        attn = gma.att(inp)
        c0 = coords_grid(1, 20, 64)
        # lookup probe: a smooth field plus a few coordinates far outside the map (zero padding)
        r = np.random.RandomState(5)
        probe = c0 + torch.from_numpy(r.uniform(-6, 6, (1, 2, 20, 64)).astype(np.float32))
        probe[0, :, 0, 0] = torch.tensor([-7.5, 3.25])
        probe[0, :, 0, 1] = torch.tensor([70.0, 25.0])
        probe[0, :, 0, 2] = torch.tensor([63.0, 19.0])
        probe[0, :, 0, 3] = torch.tensor([0.0, 0.0])
        look = cb(probe)
        corr0 = cb(c0)
        net1, mask1, delta1 = gma.update_block(net, inp, corr0, c0 - c0, attn)
        np.savez_compressed(
            os.path.join(HERE, "gma_c1.npz"),
            seed_weights=1, seed_frames=3, iters=8,
            flow_low=flow_low[0].numpy(), flow_up=flow_up[0].numpy(),
            fmap1=fmap1[0, :, ::3, ::5].numpy(), fmap2=fmap2[0, :, ::3, ::5].numpy(),
            net0=net[0, :, ::3, ::5].numpy(), inp=inp[0, :, ::3, ::5].numpy(),
            pyr3=cb.corr_pyramid[3].reshape(20, 64, 2, 8)[::3, ::5].numpy(),
            pyr1=cb.corr_pyramid[1].reshape(20, 64, 10, 32)[::7, ::9].numpy(),
            pyr0_rows=cb.corr_pyramid[0].reshape(1280, 1280)[[0, 77, 640, 1279]].numpy(),
            attn_rows=attn.reshape(1280, 1280)[[0, 77, 640, 1279]].numpy(),
            probe=probe[0].numpy(), lookup=look[0].numpy(),
            net1=net1[0, :, ::3, ::5].numpy(), delta1=delta1[0].numpy(), mask1=mask1[0, :, ::3, ::5].numpy())

        # ------------------------------------------------------------ C2: 376x1232, 12 iterations, + head on that flow
        fr = torch.from_numpy(syn.make_frames(2, 376, 1232, seed=4))
        flow_low, flow_up = gma(fr[0:1], fr[1:2], iters=12, test_mode=True)
        head.reset_lstm()
        rot, tr = head(flow_up)
        np.savez_compressed(
            os.path.join(HERE, "gma_c2.npz"),
            seed_weights=1, seed_frames=4, iters=12,
            flow_low=flow_low[0].numpy(), flow_up_s4=flow_up[0, :, ::4, ::4].numpy(),
            flow_up_sum=flow_up.double().sum(dim=(0, 2, 3)).numpy(),
            flow_up_abs=flow_up.double().abs().sum(dim=(0, 2, 3)).numpy(),
            rot=rot.numpy(), tr=tr.numpy())

        # ------------------------------------------------------------ head alone
        out = {}
        fl = torch.from_numpy(syn.make_flow(3, 376, 1232, seed=6))
        head.reset_lstm()
        feats = []
        for t in range(3):  # state carried across calls (network.py:137-139)
            rot, tr = head(fl[t:t + 1])
            out["rot%d" % t], out["tr%d" % t] = rot.numpy(), tr.numpy()
        feat = head.encoder_CNN(head.normalize_flow(fl))
        out["feat"] = feat.numpy()
        head.reset_lstm()
        rot, tr = head(fl[1:2])
        out["rot_after_reset"], out["tr_after_reset"] = rot.numpy(), tr.numpy()
        # 376x1241 flow centre-cropped to 1232 as FlowKittiDataset2 does (datasets.py:120-122)
        fw = torch.from_numpy(syn.make_flow(1, 376, 1241, seed=7))
        lo = (1241 - 1232) // 2
        head.reset_lstm()
        rot, tr = head(fw[:, :, :, lo:lo + 1232])
        out["rot_crop"], out["tr_crop"] = rot.numpy(), tr.numpy()
        head4 = ATDNVO(batch_size=4).eval()
        head4.load_state_dict(hsd)
        fl4 = torch.from_numpy(syn.make_flow(4, 376, 1232, seed=8))
        r4, t4 = head4(fl4)
        r4b, t4b = head4(fl4.flip(0))
        out.update(rot_b4=r4.numpy(), tr_b4=t4.numpy(), rot_b4_step2=r4b.numpy(), tr_b4_step2=t4b.numpy())
        np.savez_compressed(os.path.join(HERE, "clvo.npz"), seed_weights=1, **out)

        # ------------------------------------------------------------ pose algebra + padder
        r = np.random.RandomState(9)
        rots = r.uniform(-0.2, 0.2, (16, 3)).astype(np.float32)
        trs = r.uniform(-1.5, 1.5, (16, 3)).astype(np.float32)
        mats = np.stack([T.transform(torch.from_numpy(rots[i]), torch.from_numpy(trs[i])).numpy() for i in range(16)])
        eul = np.stack([T.matrix2euler(torch.from_numpy(mats[i, :3, :3])).numpy() for i in range(16)])
        absolute = T.rel2abs([torch.from_numpy(rots[i:i + 1]) for i in range(16)],
                             [torch.from_numpy(trs[i:i + 1]) for i in range(16)]).numpy()
        pads = np.array([InputPadder((3, h, w))._pad for h, w in ((376, 1232), (376, 1241), (370, 1226), (375, 1242))])
        np.savez_compressed(os.path.join(HERE, "pose.npz"), rots=rots, trs=trs, transform=mats, euler=eul,
                            rel2abs=absolute, kitti_rows=absolute[:, :3, :].reshape(17, 12),
                            pad_dims=np.array([(376, 1232), (376, 1241), (370, 1226), (375, 1242)]), pads=pads)

    # ---------------------------------------------------------------- NeuralSLAM caller behaviour (odometry mode)
    from atdn_vslam.utils.arguments import Arguments
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
            slam = NeuralSLAM(args, odometry_weights="odo.pth")
            slam.start_odometry()
            frames = torch.from_numpy(syn.make_frames(4, 376, 1241, seed=10))
            poses = [slam(frames[i]).clone().numpy() for i in range(4)]
            np.savez_compressed(os.path.join(HERE, "slam.npz"), seed_weights=1, seed_frames=10,
                                poses=np.stack(poses), n_keyframes=len(slam))
        finally:
            os.chdir(cwd)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
