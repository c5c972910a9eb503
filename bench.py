"""Headline benchmark: frame-pairs/s of KITTI-shaped odometry inference on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic frame pairs per GPU: GMA flow
(12 GRU iterations, 376x1232 after the reference's resize of 376x1241 KITTI frames) -> CLVO CNN
encoder -> 512-d feature. After the K steps the timed region also holds the sequence tail: ONE RCCL
all-gather of the features and the ordered LSTM/MLP scan + rel2abs that turns them into the 6-DoF
trajectory (every rank ends up with all N*K*B poses). Frames are resident in HBM before timing starts.

Prints ONE JSON line on rank 0 (see DESIGN.md §Measurement for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from atdn_vslam_amd import synthetic as syn  # noqa: E402
from atdn_vslam_amd import transforms  # noqa: E402
from atdn_vslam_amd.pipeline import OdometryPipeline, resize_frames  # noqa: E402
from atdn_vslam_amd.sharding import gather_features  # noqa: E402

H_KITTI, W_KITTI = 376, 1241
H, W = 376, 1232
N8 = (H // 8) * (W // 8)
ITERS = 12
# Algorithmic work per frame pair (SURVEY §8d / BASELINE.md §3): mask head + upsampling counted once.
FLOP_PER_PAIR = 0.951e12
# Dominant kernel: the fused z|r convolution of the separable ConvGRU (1x5 / 5x1, 256 output channels) over the
# 384 iteration-dependent input channels [h | motion | motion_global] (the 128 context channels are hoisted out
# of the loop): 2 * N8 * 256 * (5*384) FLOP per pair and launch, 24 launches per forward.
GRU_ZR_FLOP = 2.0 * N8 * 256 * 1920
LOOKUP_BYTES = N8 * (400 + 324) * 4.0  # <=400 cells read + 324 samples written per source pixel (fp32)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: f16/bf16 MFMA, dense
PEAK_HBM_GBS = 8000.0


def pmc_traffic(kernel_fragment, batch):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc.json: 2 x FETCH_SIZE + WRITE_SIZE, collected in separate --pmc runs at B = 4)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    if not files:
        return None
    data = json.load(open(files[-1]))
    if data.get("_batch") != batch:
        return None
    for name, v in data.items():
        if kernel_fragment in name:
            return v["hbm_bytes_per_launch"]
    return None


def usable_cores():
    """Cores this process may actually use: affinity mask and cgroup quota, capped at 64 (more threads only
    slow the oneDNN convolutions of this path down)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(gsd, hsd, frames, budget_s=25.0):
    """The CPU oracle (a port of the reference's PyTorch-CPU op sequence) on this host's cores, bounded sample."""
    from oracle import clvo_ref, gma_ref
    cores = usable_cores()
    torch.set_num_threads(cores)
    fr = frames.cpu()
    state = clvo_ref.zero_state(1)
    times = []
    t_all = time.time()
    for i in range(min(4, fr.shape[0] - 1)):
        t0 = time.time()
        _, up = gma_ref.gma_forward(gsd, fr[i:i + 1], fr[i + 1:i + 2], iters=ITERS)
        _, _, state = clvo_ref.clvo_forward(hsd, up, state)
        times.append(time.time() - t0)
        if time.time() - t_all > budget_s:
            break
    timed = times[1:] if len(times) > 1 else times  # first pair is warm-up
    sec = float(np.median(timed))
    return {"value": 1.0 / sec, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
            "sample": "%d pair(s) after 1 warm-up, 376x1232, %d iters, fp32, torch CPU ops, median %.3f s/pair"
                      % (len(timed), ITERS, sec)}


MFMA_PER_PRODUCT = {"split_f16": 3, "f16": 1, "f32": 1}   # MFMAs the engine executes per algorithmic product
DTYPE_LABEL = {"split_f16": "f32 via 3xf16 split MFMA (fp32 accumulate)",
               "f16": "f16 operands, fp32 accumulate (fast mode, own tolerance)",
               "f32": "f32 (exact fp32 MFMA)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="frame pairs per step per GPU")
    ap.add_argument("--streams", type=int, default=2, help="independent clips in flight per GPU (HIP streams)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="split_f16", choices=["split_f16", "f16", "f32"],
                    help="arithmetic of the flow network; the headline is split_f16 (fp32-grade). f16 = fast mode")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert world == args.gpus, "launch with --nproc-per-node == --gpus"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    B, K, Wm = args.batch, args.steps, args.warmup

    gsd = syn.to_torch(syn.make_gma_state(seed=1))
    hsd = syn.to_torch(syn.make_clvo_state(seed=1))
    # S pipelines on S streams: consecutive steps (clips) overlap, which fills the tails of each other's kernels
    S = max(1, args.streams)
    pipes = [OdometryPipeline(gsd, hsd, device=dev, max_batch=B, iters=ITERS, precision=args.precision) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    pipe = pipes[0]
    # synthetic clip, different per rank (each rank owns its own stretch of the sequence); resized once, resident
    clip = 2 * B + 1
    raw = torch.from_numpy(syn.make_frames(clip, H_KITTI, W_KITTI, seed=100 + rank)).to(dev)  # 376x1241, resident
    torch.cuda.synchronize()

    # Every pipeline walks its own long sequence clip by clip. The sequence is the resident clip played forwards and
    # backwards (frame k of the sequence = raw[triangle(k)]), so consecutive frames are always neighbours of the clip
    # and each frame of the sequence passes through the feature network once (continued clips reuse the shared frame).
    period = 2 * (clip - 1)
    calls = [0] * S
    idx_cache = {}

    def tri(k):
        k %= period
        return k if k < clip else period - k

    def step(i, feats, slot=None):
        nonlocal calls
        p = i % S
        j = calls[p]
        calls[p] += 1
        slot = i if slot is None else slot
        key = ((j + p) * B) % period
        if key not in idx_cache:
            idx_cache[key] = torch.tensor([tri(key + t) for t in range(B + 1)], device=dev)
        idx = idx_cache[key]
        with torch.cuda.stream(streams[p]):
            frames = resize_frames(raw[idx])             # the reference's per-frame resize to 376x1232, on the GPU
            f, _ = pipes[p].features_clip(frames, continued=(j > 0))   # B consecutive pairs of the sequence
            feats[slot * B:(slot + 1) * B] = f

    def join():
        for st_ in streams:
            torch.cuda.current_stream().wait_stream(st_)

    feats = torch.empty((max(K, Wm) * B, 512), device=dev)
    # W warm-up steps, and at least two on EVERY pipeline (the graphs of a first and of a continued clip) before timing
    for i in range(max(Wm, 2 * S)):
        step(i, feats, slot=i % max(Wm, 1))
    calls = [0] * S   # the timed region starts a fresh sequence on every pipeline: nothing computed earlier is reused
    join()
    if Wm:
        pipe.scan(feats[:B])  # warm the tail kernels too
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(K):
        step(i, feats)
        ev[i + 1].record(streams[i % S])
    join()
    torch.cuda.current_stream().synchronize()
    t_tail = time.perf_counter()   # sequence tail: all-gather + ordered LSTM scan over all N*K*B features + rel2abs
    allf = gather_features(feats[:K * B], world * K * B) if world > 1 else feats[:K * B]
    rot, tr = pipe.scan(allf)
    poses = transforms.rel2abs(rot.cpu().numpy(), tr.cpu().numpy())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tail_ms = (time.perf_counter() - t_tail) * 1e3
    assert tuple(poses.shape) == (world * K * B + 1, 4, 4) and bool(torch.isfinite(poses).all())
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # completion-to-completion intervals on the launch streams (steps overlap when S > 1)
    step_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(S, K)] if K > S else [dt * 1e3 / K]

    if rank == 0:
        total_pairs = world * K * B
        # per-stage device time of the same forward, eager with HIP events on the launch stream
        reps = 3
        st = pipe.flow_net.profile(H, W, B, iters=ITERS, reps=reps)
        zr_launch_ms = st["gru_zr"] / (2 * ITERS)
        zr_tflops = GRU_ZR_FLOP * B / (zr_launch_ms * 1e-3) / 1e12
        lookup_ms = st["lookup"] / ITERS
        fwd_ms = float(np.median(step_ms))
        out = {
            "metric": "frame-pairs/sec, KITTI 1241x376 odometry inference at 1/2/4/8 MI355X",
            "value": total_pairs / dt, "unit": "frame-pairs/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": dt * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_LABEL[args.precision], "data": "synthetic",
            "config": {"workload": "KITTI seq-03-shaped 376x1241 frames resized to 376x1232, GMA flow 12 GRU iters + "
                                   "CLVO head -> 6-DoF trajectory (BASELINE configs[1])",
                       "pairs_per_step_per_gpu": B, "streams_per_gpu": S, "gru_iters": ITERS, "parallelism": "pairs sharded x%d, one "
                       "all-gather of 512-d features, replicated LSTM scan" % world},
            # achieved = ALGORITHMIC flops / launch time; the kernel executes 3 f16 MFMAs per algorithmic product, so
            # the matrix pipe runs at 3x `achieved` (mfma_executed_*); peak = dense f16 MFMA.
            "roofline": {"bound": "mfma", "kernel": "conv_sf6_kernel<8,16,256,1,8,1,5|5,1,SfGruZR> (fused z|r ConvGRU convolution: 8x16-pixel x 256-channel blocks, fragment-major weights straight to registers)",
                         "achieved": zr_tflops, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": zr_tflops / PEAK_F16_MFMA_TFLOPS, "traffic": pmc_traffic("SfGruZR", B),
                         "launch_ms": zr_launch_ms, "flop_per_launch": GRU_ZR_FLOP * B,
                         "mfma_executed_tflops": MFMA_PER_PRODUCT[args.precision] * zr_tflops,
                         "mfma_executed_frac": MFMA_PER_PRODUCT[args.precision] * zr_tflops / PEAK_F16_MFMA_TFLOPS,
                         "vs_f32_mfma_peak": zr_tflops / PEAK_F32_MFMA_TFLOPS},
            "forward": {"ms_per_batch_median": fwd_ms, "tflops": FLOP_PER_PAIR * B / (fwd_ms * 1e-3) / 1e12,
                        "frac_of_f32_mfma_peak": FLOP_PER_PAIR * B / (fwd_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS},
            "lookup": {"ms_per_launch": lookup_ms, "achieved_GBps": LOOKUP_BYTES * B / (lookup_ms * 1e-3) / 1e9,
                       "peak_GBps": PEAK_HBM_GBS},
            "stages_ms_per_forward": {k: round(v, 4) for k, v in st.items()},
            # inside the timed region, after the K steps: grows with N (every rank scans all N*K*B features in order)
            "sequence_tail_ms": round(tail_ms, 3),
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N = 1 only (rank 0)
            out["cpu_baseline"] = cpu_baseline(gsd, hsd, resize_frames(raw[:6]))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
