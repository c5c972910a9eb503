import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import ATDNVO
DEV="cuda:0"
def head(p):
    os.environ["ATDN_SCAN_PERSISTENT"] = "1" if p else "0"
    h = ATDNVO(); h.load_state_dict(syn.to_torch(syn.make_clvo_state(seed=1))); h = h.to(DEV).eval(); h.scan(torch.zeros(2,1,512,device=DEV)); return h
r = np.random.RandomState(9); T=20
f = torch.from_numpy(r.normal(0,0.12,(T,1,512)).astype(np.float32)).to(DEV)
per, one = head(False), head(True)
r0,t0,s0 = per.scan(f); r1,t1,s1 = one.scan(f)
print("per-step rot diff by step:", [(float((r1[t]-r0[t]).abs().max())) for t in range(T)])
for k,n in enumerate(["h1","c1","h2","c2"]):
    print(n, float((s1[k]-s0[k]).abs().max()), float(s0[k].abs().max()))
