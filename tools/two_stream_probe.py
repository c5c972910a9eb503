"""Probe: does running two independent clips concurrently on two HIP streams raise throughput (tail filling)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.pipeline import OdometryPipeline, resize_frames

dev = torch.device("cuda", 0)
gsd = syn.to_torch(syn.make_gma_state(seed=1)); hsd = syn.to_torch(syn.make_clvo_state(seed=1))
for B, nstreams in ((8, 1), (4, 2), (8, 2), (4, 3)):
    pipes = [OdometryPipeline(gsd, hsd, device=dev, max_batch=B) for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    raw = resize_frames(torch.from_numpy(syn.make_frames(B + 1, 376, 1241, seed=7)).to(dev))
    def run(n):
        for i in range(n):
            for p, s in zip(pipes, streams):
                with torch.cuda.stream(s):
                    p.features_clip(raw)
    run(2); torch.cuda.synchronize()
    t0 = time.time(); n = 8; run(n); torch.cuda.synchronize(); dt = time.time() - t0
    print("B=%d x %d streams: %.1f pairs/s" % (B, nstreams, n * nstreams * B / dt))
    del pipes
    torch.cuda.empty_cache()
