"""Diagnostic (run under rocprofv3 --kernel-trace --stats): the fused lookup+convc1 kernel and the sampling-only kernel
(lookup_conv_kernel<false, ...>, launched by the corrfeat debug read) on one 16-pair batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import RAFTGMA
H, W, B = 376, 1232, 16
N = (H // 8) * (W // 8)
net = RAFTGMA(max_batch=B)
net.load_state_dict(syn.to_torch(syn.make_gma_state(seed=1)))
net = net.to("cuda:0").eval()
fr = torch.from_numpy(syn.make_frames(B + 1, H, W, seed=100)).cuda()
for _ in range(3):
    net.forward_sequence(fr, iters=12)
for _ in range(5):
    net.debug_read("corrfeat", (B * N, 352), H, W)
torch.cuda.synchronize()
