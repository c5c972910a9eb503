"""Diagnostic (run under rocprofv3 --kernel-trace --stats): attention x V on a 16-pair clip, the kernel alone on its stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import RAFTGMA
H, W, B = 376, 1232, 16
net = RAFTGMA(max_batch=B, saturation_check_every=0)
net.load_state_dict(syn.to_torch(syn.make_gma_state(seed=1)))
net = net.to("cuda:0").eval()
fr = torch.from_numpy(syn.make_frames(B + 1, H, W, seed=100)).cuda()
for _ in range(4):
    net.forward_sequence(fr, iters=12)
torch.cuda.synchronize()
