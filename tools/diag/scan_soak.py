"""Soak of the persistent LSTM scan: thousands of launches, alone and beside a flow network running on another stream; every result
compared with the first one bit for bit. A launch that gives up would print a line to stderr (and repeat on the per-step kernel)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import ATDNVO, RAFTGMA

DEV = "cuda:0"
head = ATDNVO()
head.load_state_dict(syn.to_torch(syn.make_clvo_state(seed=1)))
head = head.to(DEV).eval()
net = RAFTGMA(max_batch=8)
net.load_state_dict(syn.to_torch(syn.make_gma_state(seed=1)))
net = net.to(DEV).eval()
fr = torch.from_numpy(syn.make_frames(9, 376, 1232, seed=3)).to(DEV)
r = np.random.RandomState(1)
s2 = torch.cuda.Stream()
for T, n in ((320, 1500), (4540, 150), (17, 1500)):
    f = torch.from_numpy(r.normal(0, 0.15, (T, 1, 512)).astype(np.float32)).to(DEV)
    ref = head.scan(f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bad = 0
    for i in range(n):
        if i % 2 == 1:
            with torch.cuda.stream(s2):
                net.forward_sequence(fr, iters=2)
        out = head.scan(f)
        if not (torch.equal(out[0], ref[0]) and torch.equal(out[2], ref[2])):
            bad += 1
    torch.cuda.synchronize()
    print("T = %5d: %d scans (every second one beside an 8-pair flow forward), %d differ from the first, %.2f ms per scan"
          % (T, n, bad, (time.perf_counter() - t0) / n * 1e3), flush=True)
