"""Flow error against the CPU oracle for the attention paths (diagnostic):
    ATDN_ATTN_LEGACY=1 (logits GEMM + softmax pass + 4-byte matrix) | default (fused, H3 storage)   x   cnet scale 1 / 16"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import RAFTGMA
from oracle import gma_ref

label = sys.argv[1] if len(sys.argv) > 1 else "default"
gsd = syn.to_torch(syn.make_gma_state(seed=1))
for scale, (h, w), iters in ((1.0, (160, 512), 8), (16.0, (160, 512), 8), (1.0, (376, 1232), 12)):
    sd = {k: v.clone() for k, v in gsd.items()}
    sd["cnet.conv1.weight"] *= scale
    sd["cnet.conv1.bias"] *= scale
    fr = torch.from_numpy(syn.make_frames(2, h, w, seed=71))
    ref_low, ref_up = gma_ref.gma_forward(sd, fr[0:1], fr[1:2], iters=iters)
    out = []
    for prec in ("split_f16", "f32"):
        net = RAFTGMA(max_batch=1, precision=prec)
        net.load_state_dict(sd)
        net = net.to("cuda:0").eval()
        low, up = net(fr[0:1].cuda(), fr[1:2].cuda(), iters=iters, test_mode=True)
        out.append("%s low %.2e up %.2e" % (prec, float((low.cpu() - ref_low).abs().max()), float((up.cpu() - ref_up).abs().max())))
    print("%-8s scale %-4g %dx%d: %s" % (label, scale, h, w, " | ".join(out)))
