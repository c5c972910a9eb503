"""Diagnostic: run_sequence throughput on a long synthetic sequence, host (ingest) vs device-resident frames, 1 vs 2 lanes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.pipeline import OdometryPipeline

B = 16
dev = torch.device("cuda", 0)
gsd = syn.to_torch(syn.make_gma_state(seed=1)); hsd = syn.to_torch(syn.make_clvo_state(seed=1))
pipes = [OdometryPipeline(gsd, hsd, device=dev, max_batch=B, iters=12) for _ in range(2)]
clip = 2 * B + 1
period = 2 * (clip - 1)
base = torch.from_numpy(syn.make_frames(clip, 376, 1241, seed=100)).round().clamp(0, 255).to(torch.uint8)
order = [(k if k < clip else period - k) for k in range(period)] + list(range(B + 1))
seq_host = base[order].contiguous().pin_memory()
T = int(os.environ.get("T", "1601"))
cyc = bench.CyclicSequence(seq_host, period, T)
class DevSeq(bench.CyclicSequence):
    def __init__(self, *a):
        super().__init__(*a); self.is_cuda = True; self.buf = self.buf.to(dev)
dcyc = DevSeq(seq_host, period, T)
for name, frames, lanes in (("host 2 lanes", cyc, pipes[1:]), ("device 2 lanes", dcyc, pipes[1:]), ("host 1 lane", cyc, None), ("host 2 lanes again", cyc, pipes[1:])):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        timing = {}
        pipes[0].run_sequence(frames, batch=B, lanes=lanes, timing=timing)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("%-20s %7.1f pairs/s  encode %.3f s scan %.1f ms" % (name, (T - 1) / dt, timing["encode_s"], timing["scan_s"] * 1e3))

# host-side cost of enqueueing one clip (device-resident frames, nothing to wait for): if this approaches half a step, two
# lanes make the driver host-bound
import time as _t
fr = dcyc[0:B + 1]
from atdn_vslam_amd.pipeline import resize_frames
torch.cuda.synchronize()
ts = []
for k in range(12):
    t0 = _t.perf_counter()
    x = resize_frames(fr, pipes[0].size)
    f, _ = pipes[0].features_clip(x, continued=(k > 0))
    ts.append(_t.perf_counter() - t0)
torch.cuda.synchronize()
print("host enqueue time per clip (ms):", " ".join("%.2f" % (t * 1e3) for t in ts))
