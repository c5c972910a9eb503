"""Golden fixture for map creation: runs the REFERENCE's own `NeuralSLAM.__create_map`
(atdn_vslam/slam_framework/neural_slam.py:305-352) on a small synthetic keyframe directory and records the per-epoch
losses, a digest of every trained tensor and the embedding of one frame.

torchvision is not installed in the build container, so the three torchvision calls of that method are stubs here:
`TF.resize` (bilinear, antialias — what torchvision does for tensors), `TF.gaussian_blur` (torchvision's kernel:
sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8, reflect padding) and `ColorJitter` = identity. The fixture therefore pins the
model, loss, optimiser, schedule and data order of the reference's loop, not its colour augmentation.

Run only in the build container (needs /root/reference):   python tests/golden/make_golden_map.py
Fixtures are numbers only; nothing of the reference's text is stored.
"""
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
sys.path.insert(0, HERE)
REF = "/root/reference"

from make_golden import install_stubs  # noqa: E402

SEED, NKEY, HK, WK = 5, 32, 192, 256
SNAP_EPOCH = 6   # the per-epoch checkpoint whose digest is recorded besides the final one (a short test re-trains to here)


def write_keyframes(base, syn):
    os.makedirs(os.path.join(base, "rgb"), exist_ok=True)
    frames = syn.make_frames(NKEY, HK, WK, seed=41)
    for i in range(NKEY):
        torch.save(torch.from_numpy(frames[i]).byte(), os.path.join(base, "rgb", "%06d.pth" % i))
    return frames


def digest(sd):
    names = list(sd.keys())
    vals = [sd[k].double().flatten() for k in names]
    return (np.array([float(v.sum()) for v in vals]), np.array([float(v.norm()) for v in vals]),
            np.stack([np.resize(v[:4].numpy(), 4) for v in vals]))


def main():
    install_stubs()
    tvt = sys.modules["torchvision.transforms"]
    tvf = sys.modules["torchvision.transforms.functional"]

    class ColorJitter:  # identity: see the module docstring
        def __init__(self, **kw):
            pass

        def __call__(self, x):
            return x

    def gaussian_blur(img, kernel_size, sigma=None):
        ks = kernel_size[0]
        sg = 0.3 * ((ks - 1) * 0.5 - 1) + 0.8
        lim = (ks - 1) * 0.5
        t = torch.linspace(-lim, lim, ks)
        pdf = torch.exp(-0.5 * (t / sg).pow(2))
        k1 = pdf / pdf.sum()
        k2 = torch.mm(k1[:, None], k1[None, :]).expand(img.shape[-3], 1, ks, ks)
        pad = [ks // 2] * 4
        return F.conv2d(F.pad(img, pad, mode="reflect"), k2, groups=img.shape[-3])

    tvt.ColorJitter = ColorJitter
    tvf.gaussian_blur = gaussian_blur
    sys.path.insert(0, os.path.join(REF, "GMA-1.0.0-py3-none-any.whl"))
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    from atdn_vslam.slam_framework.neural_slam import NeuralSLAM
    from atdn_vslam.utils.normalizations import get_rgb_norm
    from atdn_vslam_amd import synthetic as syn

    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            base = os.path.join(tmp, "kf")
            frames = write_keyframes(base, syn)
            me = types.SimpleNamespace()
            me._NeuralSLAM__args = types.SimpleNamespace(device="cpu")
            me._NeuralSLAM__keyframes_base_path = base
            me._NeuralSLAM__norm_rgb = get_rgb_norm()
            # the method overwrites its checkpoint every epoch: keep a digest of the SNAP_EPOCH-th one
            snap = {}
            real_save, count = torch.save, [0]

            def save(obj, f, *a, **k):
                if isinstance(f, str) and f.endswith("MappingVAE_weights.pth"):
                    count[0] += 1
                    if count[0] == SNAP_EPOCH:
                        snap["digest"] = digest({kk: vv.detach().clone() for kk, vv in obj.items()})
                return real_save(obj, f, *a, **k)

            torch.save = save
            torch.manual_seed(SEED)
            try:
                NeuralSLAM._NeuralSLAM__create_map(me)
            finally:
                torch.save = real_save
            net = me._NeuralSLAM__mapping_net
            losses = torch.load("mapping_loss.pth").numpy()
            saved = torch.load(os.path.join(base, "MappingVAE_weights.pth"))
            sums, norms, heads = digest(saved)
            with torch.no_grad():
                mu = net(torch.from_numpy(frames[:2]))[0].numpy()
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "map.npz"), seed=SEED, nkey=NKEY, hk=HK, wk=WK, frames_seed=41, losses=losses,
                        sums=sums, norms=norms, heads=heads, mu=mu, snap_epoch=SNAP_EPOCH, snap_sums=snap["digest"][0],
                        snap_norms=snap["digest"][1], snap_heads=snap["digest"][2])
    print("losses", losses[:3], "...", losses[-3:])


if __name__ == "__main__":
    main()
