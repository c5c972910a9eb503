"""cProfile of pipeline.VisualOdometry's per-frame call (host side): where the milliseconds that are not kernels go."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.pipeline import VisualOdometry

dev = torch.device("cuda", 0)
vo = VisualOdometry(syn.to_torch(syn.make_gma_state(seed=1)), syn.to_torch(syn.make_clvo_state(seed=1)), device=dev, iters=12)
fr = torch.from_numpy(syn.make_frames(12, 376, 1241, seed=21)).round().clamp(0, 255).to(torch.uint8).pin_memory()
for k in range(4):
    vo(fr[k])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    for k in range(4, 12):
        vo(fr[k])
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
