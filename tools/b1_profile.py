"""Stage times of ONE frame pair per forward (B = 1: the per-frame callers VisualOdometry / NeuralSLAM) next to the 16-pair clip."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.pipeline import OdometryPipeline, VisualOdometry

dev = torch.device("cuda", 0)
gsd = syn.to_torch(syn.make_gma_state(seed=1))
hsd = syn.to_torch(syn.make_clvo_state(seed=1))
out = {}
for B in (1, 2, 4, 16):
    pipe = OdometryPipeline(gsd, hsd, device=dev, max_batch=B, iters=12)
    st = pipe.flow_net.profile(376, 1232, B, iters=12, reps=3, mode="continued")
    out[B] = {k: round(v, 3) for k, v in st.items()}
    out[B]["sum"] = round(sum(st.values()), 3)
    del pipe
print(json.dumps(out))
vo = VisualOdometry(gsd, hsd, device=dev, iters=12)
fr = torch.from_numpy(syn.make_frames(12, 376, 1241, seed=21))
for k in range(4):
    vo(fr[k])
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(4, 12):
    vo(fr[k])
torch.cuda.synchronize()
print("VisualOdometry per frame (host uint8->pose, synchronous): %.2f ms" % ((time.perf_counter() - t0) / 8 * 1e3))
