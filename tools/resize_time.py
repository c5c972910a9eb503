"""Device time of the fused uint8 -> fp32 + resize of a 17-frame KITTI clip (diagnostic)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atdn_vslam_amd.pipeline import resize_frames
fr = torch.randint(0, 256, (17, 3, 376, 1241), dtype=torch.uint8, device="cuda:0")
for _ in range(5):
    o = resize_frames(fr)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    o = resize_frames(fr)
b.record()
torch.cuda.synchronize()
print("resize of 17 frames: %.1f us; checksum %.6e" % (a.elapsed_time(b) * 1e3 / 50, float(o.double().sum())))
