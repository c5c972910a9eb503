"""Diagnostic: bench-style two-stream loop vs run_sequence lanes on device-resident frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.pipeline import OdometryPipeline, resize_frames
B = 16
dev = torch.device("cuda", 0)
gsd = syn.to_torch(syn.make_gma_state(seed=1)); hsd = syn.to_torch(syn.make_clvo_state(seed=1))
pipes = [OdometryPipeline(gsd, hsd, device=dev, max_batch=B, iters=12) for _ in range(2)]
clip = 2 * B + 1
period = 2 * (clip - 1)
base = torch.from_numpy(syn.make_frames(clip, 376, 1241, seed=100)).round().clamp(0, 255).to(torch.uint8)
order = [(k if k < clip else period - k) for k in range(period)] + list(range(B + 1))
seq_dev = base[order].contiguous().to(dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
NC = 50   # clips per lane
def loop(variant):
    keep = []
    main = torch.cuda.current_stream()
    for st in streams: st.wait_stream(main)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    feats = torch.empty((2 * NC * B, 512), device=dev)
    for k in range(NC):
        for p in range(2):
            key = ((k + p) * B) % period
            with torch.cuda.stream(streams[p]):
                fr = resize_frames(seq_dev[key:key + B + 1])
                f, _ = pipes[p].features_clip(fr, continued=(k > 0))
                if variant == "copy":
                    feats[(k * 2 + p) * B:(k * 2 + p + 1) * B] = f
                elif variant == "keep":
                    keep.append(f)
                elif variant == "keep+record":
                    f.record_stream(main); keep.append(f)
    for st in streams: main.wait_stream(st)
    torch.cuda.synchronize()
    return 2 * NC * B / (time.perf_counter() - t0)
for v in ("copy", "keep", "keep+record", "copy"):
    loop(v)
    print("%-12s %.1f pairs/s" % (v, loop(v)))
