"""Per-stage device time of one B-pair forward (B, MODE = pair | sequence | continued, REPS from the environment) (atdn_gma_profile: eager launches, HIP events), for A/B runs of kernel
variants inside ONE gpurun call (the boxes of the pool differ by a few %):
    python tools/stage_profile.py [label]      (environment switches such as ATDN_ATTN_LEGACY=1 select the variant)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import RAFTGMA

B = int(os.environ.get("B", "8"))
net = RAFTGMA(max_batch=B, precision=os.environ.get("PRECISION", "split_f16"))
net.load_state_dict(syn.to_torch(syn.make_gma_state(seed=1)))
net = net.to("cuda:0").eval()
fr = torch.from_numpy(syn.make_frames(B + 1, 376, 1232, seed=100)).to("cuda:0")
for _ in range(2):
    net.forward_sequence(fr, iters=12)
torch.cuda.synchronize()
st = net.profile(376, 1232, B, iters=12, reps=int(os.environ.get("REPS", "5")), mode=os.environ.get("MODE", "pair"))
tot = sum(st.values())
print("%-14s total %.3f ms | " % (sys.argv[1] if len(sys.argv) > 1 else "default", tot) +
      " ".join("%s %.3f" % (k, v) for k, v in st.items()))
