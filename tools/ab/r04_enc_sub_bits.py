"""Depth-first encoder sub-batches must not change a bit: flow of one 16-pair clip with ATDN_ENC_SUB = 0 / 3 / 4 / 5."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import RAFTGMA

B = 16
sd = syn.to_torch(syn.make_gma_state(seed=1))
fr = torch.from_numpy(syn.make_frames(B + 1, 376, 1232, seed=100)).to("cuda:0")
ref = None
for sub in (0, 3, 4, 5):
    os.environ["ATDN_ENC_SUB"] = str(sub)
    net = RAFTGMA(max_batch=B)
    net.load_state_dict(sd)
    net = net.to("cuda:0").eval()
    low, up = net.forward_sequence(fr, iters=12)
    low2, up2 = net.forward_sequence(torch.cat([fr[-1:], fr[:B]]), iters=12, continued=True)
    torch.cuda.synchronize()
    out = (low.clone(), up.clone(), low2.clone(), up2.clone())
    if ref is None:
        ref = out
    else:
        d = [float((a - b).abs().max()) for a, b in zip(out, ref)]
        print("sub", sub, "max abs diff vs sub 0:", d)
        assert all(x == 0.0 for x in d), d
    del net
print("bit-identical")
