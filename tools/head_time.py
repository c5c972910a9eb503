"""Device time of the CLVO encoder (ATDNVO.encode) on a 16-pair clip of 376x1232 flows (diagnostic; HIP events)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import ATDNVO
B = int(os.environ.get("B", "16"))
vo = ATDNVO(batch_size=1)
vo.load_state_dict(syn.to_torch(syn.make_clvo_state(seed=2)))
vo = vo.to("cuda:0").eval()
g = torch.Generator().manual_seed(3)
fl = (torch.randn(B, 2, 376, 1232, generator=g) * 8.0).to("cuda:0")
for _ in range(3):
    f0 = vo.encode(fl)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    f = vo.encode(fl)
b.record()
torch.cuda.synchronize()
print("%-10s encode %.3f ms per %d pairs; feature checksum %.9e max|f| %.4f" % (sys.argv[1] if len(sys.argv) > 1 else "lib", a.elapsed_time(b) / 20, B, float(f.double().sum()), float(f.abs().max())))
