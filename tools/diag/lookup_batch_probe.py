"""Diagnostic: sampled correlation features of a 2-pair batch against the single-pair runs (batch invariance of the lookup)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import RAFTGMA
H, W = 376, 1232
N = (H // 8) * (W // 8)
gsd = syn.to_torch(syn.make_gma_state(seed=1))
net = RAFTGMA(max_batch=2)
net.load_state_dict(gsd)
net = net.to("cuda:0").eval()
fr = torch.from_numpy(syn.make_frames(2, H, W, seed=7)).cuda()
it = int(os.environ.get("ITERS", "2"))
net(torch.cat([fr[0:1], fr[1:2]]), torch.cat([fr[1:2], fr[0:1]]), iters=it, test_mode=True)
cb = net.debug_read("corrfeat", (2 * N, 352), H, W).clone()
cu = net.debug_read("coords1", (2 * N, 2), H, W).clone()
net(fr[1:2], fr[0:1], iters=it, test_mode=True)
c1 = net.debug_read("corrfeat", (2 * N, 352), H, W)[:N].clone()
cu1 = net.debug_read("coords1", (2 * N, 2), H, W)[:N].clone()
print("coords equal:", bool(torch.equal(cu[N:], cu1)), float((cu[N:] - cu1).abs().max()))
d = (cb[N:] - c1).abs()
bad = (d > 0).nonzero()
print("mismatching samples:", bad.shape[0], "max", float(d.max()))
if bad.shape[0]:
    px = bad[:, 0].unique()
    print("pixels:", px[:40].tolist(), "... count", px.numel())
    ch = bad[:, 1].unique()
    print("channels:", ch[:60].tolist(), "count", ch.numel())
    print("pos in block (batch):", sorted(set(((px + N) % 32).tolist())))
    print("pos in block (single):", sorted(set((px % 32).tolist())))
