"""Throughput of the CLVO training iteration (BASELINE config 4 shape: batch 24, sequence length 6, 376x1232 flows)
on one GPU, or data-parallel under torch.distributed.run (one process per GPU, RCCL all-reduce of the flat gradient).

    python tools/bench_train.py [--gpus N] [--batch 24] [--seq 6] [--steps 10] [--warmup 2]

`--gpus N` with N > 1 and no launcher around it starts its own N ranks (atdn_vslam_amd/launch.py), exactly as bench.py does.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":   # before torch is imported / any GPU call: the parent only starts the ranks and relays their exit code
    from atdn_vslam_amd.launch import spawn_ranks_if_needed
    _rc = spawn_ranks_if_needed(__file__)
    if _rc is not None:
        sys.exit(_rc)

import numpy as np
import torch

from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.training import CLVOTrainer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--seq", type=int, default=6)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gpus", type=int, default=None, help="ranks of the data-parallel job (default: WORLD_SIZE, else 1)")
    ap.add_argument("--launch-check", action="store_true", help="launch plumbing only (gloo all-gather of the ranks, no GPU call)")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert a.gpus is None or a.gpus == world, "%d rank(s) for --gpus %s" % (world, a.gpus)
    if a.launch_check:
        import torch.distributed as dist
        rows = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        if world > 1:
            dist.init_process_group("gloo")
            dist.all_gather(rows, torch.tensor([rank], dtype=torch.int64))
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks": [int(r[0]) for r in rows]}))
        return
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    tr = CLVOTrainer(syn.to_torch(syn.make_clvo_state(seed=1)), a.batch, a.seq, device=dev, total_steps=a.steps + a.warmup)
    r = np.random.RandomState(7 + rank)
    flows = torch.from_numpy(syn.make_flow(a.batch * a.seq, 376, 1232, seed=50 + rank)).view(a.batch, a.seq, 2, 376, 1232).to(dev)
    rot = torch.from_numpy(r.uniform(-0.02, 0.02, (a.batch, a.seq, 3)).astype(np.float32)).to(dev)
    trn = torch.from_numpy(r.uniform(-0.5, 1.5, (a.batch, a.seq, 3)).astype(np.float32)).to(dev)
    losses = []
    for _ in range(a.warmup):
        tr.step(flows, rot, trn)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses.append(tr.step(flows, rot, trn))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        # roofline of the iteration: the head's convolutions are 0.635 GMAC per flow frame forward (SURVEY appendix
        # A) on the exact-fp32 matrix core (v_mfma_f32_16x16x4_f32), the backward pass is two GEMMs of the same size per layer
        # (data and weight gradients), the fully connected / LSTM tail 5.03 MMAC x 3; bytes: every flow frame is read once
        # (2 x 376 x 1232 fp32) and the stem's 16-channel activation at 188 x 616 is written forward and read backward
        frames = a.batch * a.seq
        flop = frames * 3.0 * 2.0 * (0.635e9 + 5.03e6)
        sec = dt / a.steps
        by = frames * (2 * 376 * 1232 * 4.0 + 2 * 16 * 188 * 616 * 4.0)
        print(json.dumps({"metric": "CLVO training flow-frames/s (forward + backward + AdamW)", "value": world * a.batch * a.seq * a.steps / dt,
                          "unit": "flow frames/s", "n_gpus": world, "ms_per_iteration": dt * 1e3 / a.steps, "batch_per_gpu": a.batch,
                          "sequence_length": a.seq, "loss_first": losses[0], "loss_last": losses[-1],
                          "roofline": {"bound": "mfma", "achieved": flop / sec / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                                       "frac": flop / sec / 1e12 / 157.3, "algorithmic_flop_per_iteration": flop,
                                       "note": "fp32 MFMA peak (the trainer computes in exact fp32); per GPU"},
                          "hbm": {"algorithmic_bytes_per_iteration": by, "achieved_GBps": by / sec / 1e9, "peak_GBps": 8000.0,
                                  "frac": by / sec / 1e9 / 8000.0}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
