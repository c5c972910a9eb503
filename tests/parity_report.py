"""Test infrastructure (may use oracle/): prints the per-stage error of the HIP path against the CPU oracle (run on the GPU box).
Used to calibrate the tolerances stated in tests/test_gpu_parity.py and DESIGN.md."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repository root
sys.path.insert(0, ROOT)
from atdn_vslam_amd import synthetic as syn  # noqa: E402
from atdn_vslam_amd.modules import ATDNVO, RAFTGMA  # noqa: E402
from oracle import clvo_ref, gma_ref  # noqa: E402

DEV = "cuda:0"


def rep(name, got, ref):
    d = (got.double() - ref.double()).abs()
    s = float(ref.abs().max())
    print("%-28s max|ref| %10.4f  maxerr %.3e  rel %.3e  meanerr %.3e" % (name, s, float(d.max()), float(d.max()) / max(s, 1e-30), float(d.mean())))


def main():
    gsd = syn.to_torch(syn.make_gma_state(seed=1))
    net = RAFTGMA(max_batch=1)
    net.load_state_dict(gsd)
    net = net.to(DEV)
    for (H, W, iters, seed) in ((160, 512, 8, 3), (376, 1232, 12, 4)):
        print("=== %dx%d" % (H, W))
        fr = torch.from_numpy(syn.make_frames(2, H, W, seed=seed))
        H8, W8 = H // 8, W // 8
        N = H8 * W8
        taps = {}
        ref_low1, ref_up1 = gma_ref.gma_forward(gsd, fr[0:1], fr[1:2], iters=1, taps=taps)
        low1, up1 = net(fr[0:1].to(DEV), fr[1:2].to(DEV), iters=1, test_mode=True)
        torch.cuda.synchronize()
        nchw = lambda b, c: b.reshape(1, H8, W8, c).permute(0, 3, 1, 2)
        fmap = net.debug_read("fmap", (2, N, 256), H, W)
        rep("fmap1", nchw(fmap[0], 256), taps["fmap1"])
        rep("fmap2", nchw(fmap[1], 256), taps["fmap2"])
        for l in range(4):
            hl, wl = H8 >> l, W8 >> l
            rep("pyr%d" % l, net.debug_read("pyr%d" % l, (N, hl * wl), H, W), taps["pyramid"][l].reshape(N, hl * wl))
        x = net.debug_read("x", (N, 384), H, W)
        rep("inp", nchw(x[:, :128], 128), taps["inp"])
        ldn = (N + 31) // 32 * 32
        attn = net.debug_read("attn", (N, ldn), H, W)[:, :N]
        rep("attn", attn, taps["attn"].reshape(N, N))
        rep("lookup0", nchw(net.debug_read("corrfeat", (N, 352), H, W)[:, :324], 324), taps["lookup0"])
        rep("mf0[:126]", nchw(x[:, 128:256], 128)[:, :126], taps["mf0"][:, :126])
        rep("mfg0", nchw(x[:, 256:384], 128), taps["mfg0"])
        rep("net1", nchw(net.debug_read("net", (N, 128), H, W), 128), taps["net1"])
        rep("delta0", low1.cpu(), ref_low1)
        rep("mask(it1)", nchw(net.debug_read("mask", (N, 576), H, W), 576), taps["mask"])
        rep("flow_up(it1)", up1.cpu(), ref_up1)
        for it in (2, 4, 8, 12):
            if it > iters:
                break
            rl, ru = gma_ref.gma_forward(gsd, fr[0:1], fr[1:2], iters=it)
            gl, gu = net(fr[0:1].to(DEV), fr[1:2].to(DEV), iters=it, test_mode=True)
            rep("flow_low  iters=%d" % it, gl.cpu(), rl)
            rep("flow_up   iters=%d" % it, gu.cpu(), ru)
    hsd = syn.to_torch(syn.make_clvo_state(seed=1))
    head = ATDNVO()
    head.load_state_dict(hsd)
    head = head.to(DEV)
    fl = torch.from_numpy(syn.make_flow(2, 376, 1232, seed=6))
    rep("clvo feat", head.encode(fl.to(DEV)).cpu(), clvo_ref.clvo_encode(hsd, fl))
    st = clvo_ref.zero_state(1)
    for t in range(2):
        rot, tr = head(fl[t:t + 1].to(DEV))
        rr, rt, st = clvo_ref.clvo_forward(hsd, fl[t:t + 1], st)
        rep("rot step %d" % t, rot.cpu(), rr)
        rep("tr  step %d" % t, tr.cpu(), rt)


if __name__ == "__main__":
    main()
