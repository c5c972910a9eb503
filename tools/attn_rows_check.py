"""Diagnostic: stored attention probabilities (decoded) of the H3 and SF4 formats against the CPU oracle's softmax."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import RAFTGMA
from oracle import gma_ref

gsd = syn.to_torch(syn.make_gma_state(seed=1))
for scale in (1.0, 16.0):
    sd = {k: v.clone() for k, v in gsd.items()}
    sd["cnet.conv1.weight"] *= scale
    sd["cnet.conv1.bias"] *= scale
    fr = torch.from_numpy(syn.make_frames(2, 160, 512, seed=71))
    taps = {}
    gma_ref.gma_forward(sd, fr[0:1], fr[1:2], iters=1, taps=taps)
    ref = taps["attn"].reshape(1280, 1280).double()
    for fmt in ("h3", "sf4"):
        os.environ["ATDN_ATTN_FMT"] = fmt
        net = RAFTGMA(max_batch=1)
        net.load_state_dict(sd)
        net = net.to("cuda:0").eval()
        net(fr[0:1].cuda(), fr[1:2].cuda(), iters=1, test_mode=True)
        a = net.debug_read("attn", (1280, 1280), 160, 512).double()
        err = (a - ref).abs()
        rowmax = ref.max(dim=1, keepdim=True).values
        print("scale %-4g %s: max abs err %.3e, max err / rowmax %.3e, row-sum err %.3e, peak prob %.3f, rel err of entries > 1e-3*rowmax: %.3e"
              % (scale, fmt, float(err.max()), float((err / rowmax).max()), float((a.sum(1) - 1).abs().max()), float(ref.max()),
                 float((err / ref.clamp_min(1e-30))[ref > 1e-3 * rowmax].max())))
