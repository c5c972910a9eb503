"""Headline benchmark: frame-pairs/s of KITTI-shaped odometry inference on N MI355X.

    python bench.py --gpus N --steps 20 --warmup 3          (N > 1: starts its own N ranks, see atdn_vslam_amd/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic frame pairs per GPU: GMA flow
(12 GRU iterations, 376x1232 after the reference's resize of 376x1241 KITTI frames) -> CLVO CNN
encoder -> 512-d feature. After the K steps the timed region also holds the sequence tail: ONE RCCL
all-gather of the features and the ordered LSTM/MLP scan + rel2abs that turns them into the 6-DoF
trajectory (every rank ends up with all N*K*B poses). The uint8 frames are resident in HBM before timing starts
(`value`); a second timed pass ingests them from pinned host memory instead (`h2d_inclusive`, never the headline).
Secondary legs in the same JSON line (none of them is `value`): `config3` = BASELINE configs[2] as a benchmark — ONE
synthetic 4,541-frame uint8 sequence in pinned host memory, sharded over the N ranks, through
OdometryPipeline.run_sequence (ingest, flow, head, one all-gather, replicated scan, rel2abs): 4540 / max-over-ranks wall;
`f16_fast` = the opt-in plain-f16 mode with its measured flow error against the split-f16 output of the same clip.

Prints ONE JSON line on rank 0 (see DESIGN.md §Measurement for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# `python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks (python -m torch.distributed.run, one per
# GPU) from THIS process before torch is imported or any GPU call is made, relay their output and exit with their code
# (atdn_vslam_amd/launch.py; the parent never touches the GPU and never re-execs). Under an external launcher WORLD_SIZE is
# set and this is a no-op.
if __name__ == "__main__":
    from atdn_vslam_amd.launch import spawn_ranks_if_needed
    _rc = spawn_ranks_if_needed(__file__)
    if _rc is not None:
        sys.exit(_rc)

# N > 1: the ranks of one node share its host cores. Cap every rank's CPU thread pools BEFORE torch (OpenMP / MKL) starts
# them — 8 ranks x torch's default of one thread per core would oversubscribe a 16-core box 8-fold around every host-side
# step of the timed loop. The variable is overridden only when nobody chose it: unset, or the 1 that an external
# torch.distributed.run exports by default when it is unset (TORCHELASTIC_RUN_ID marks that launcher; this file's own launcher
# sets the share itself and flags a user's explicit figure with ATDN_OMP_FROM_USER).
_WORLD = int(os.environ.get("WORLD_SIZE", "1"))
if _WORLD > 1:
    try:
        _cores = len(os.sched_getaffinity(0))
    except AttributeError:
        _cores = os.cpu_count() or 1
    _share = str(max(1, _cores // _WORLD))
    _omp = os.environ.get("OMP_NUM_THREADS")
    if _omp is None or (_omp == "1" and "TORCHELASTIC_RUN_ID" in os.environ and "ATDN_OMP_FROM_USER" not in os.environ):
        os.environ["OMP_NUM_THREADS"] = _share
    os.environ.setdefault("MKL_NUM_THREADS", os.environ["OMP_NUM_THREADS"])

import numpy as np
import torch
import torch.distributed as dist

from atdn_vslam_amd import synthetic as syn  # noqa: E402
from atdn_vslam_amd import transforms  # noqa: E402
from atdn_vslam_amd.pipeline import FrameIngest, OdometryPipeline, resize_frames  # noqa: E402
from atdn_vslam_amd.sharding import gather_features, rendezvous  # noqa: E402

H_KITTI, W_KITTI = 376, 1241
H, W = 376, 1232
N8 = (H // 8) * (W // 8)
ITERS = 12
# Frame pairs per step (clip length) per GPU. Measured on one box: 8 pairs 348.3 pairs/s, 12: 354.5, 16: 356.5 (a second
# box: 16: 354.9, 24: 358.0, 32: 359.3) — longer clips fill the tails of every launch a little better and encode the
# shared frame of consecutive clips less often. 16 takes most of that at half the step latency of 32.
DEFAULT_BATCH = 16
# Algorithmic work per frame pair (SURVEY §8d / BASELINE.md §3): mask head + upsampling counted once.
FLOP_PER_PAIR = 0.951e12
# ... of which the path no longer EXECUTES 70.6 GMAC per pair: one feature-network pass per pair (shared frames, 31.5 GMAC)
# and the context-channel part of the ConvGRU convolutions in iterations 2..12 (39.1 GMAC, computed once per pair)
FLOP_PER_PAIR_EXECUTED = FLOP_PER_PAIR - 2.0 * (31.48e9 + 39.13e9)
LOOKUP_BYTES = N8 * (400 + 324) * 4.0  # <=400 cells read + 324 samples written per source pixel (fp32), SURVEY §8d
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: f16/bf16 MFMA, dense
PEAK_HBM_GBS = 8000.0


class CyclicSequence:
    """A T-frame uint8 camera sequence in pinned host memory without T frames of memory: frame k of the sequence is frame
    k mod `period` of a (2B + 1)-frame clip played forwards and backwards, laid out in sequence order with B + 1 extra
    frames at the end, so that any window of at most B + 1 consecutive frames is ONE contiguous slice of the pinned buffer
    (consecutive frames are always neighbours of the base clip: the flow between them is real motion, not a jump).
    Duck-types what OdometryPipeline.run_sequence needs: .shape, .dtype, .is_cuda, slice indexing."""

    def __init__(self, seq_host, period, T):
        self.buf, self.period, self.T = seq_host, period, T
        self.shape = (T,) + tuple(seq_host.shape[1:])
        self.dtype, self.is_cuda = seq_host.dtype, False
        self.window = seq_host.shape[0] - period

    def __getitem__(self, sl):
        a, b, step = sl.indices(self.T)
        assert step == 1 and 0 < b - a <= self.window, "windows of at most %d consecutive frames" % self.window
        k = a % self.period
        return self.buf[k:k + (b - a)]


def kernel_table(stages, B):
    """Per-kernel rows from the SAME-RUN per-stage HIP-event timing (atdn_gma_profile; stages that hold launches of one
    kernel only): name, rocprof name fragment, launches per forward, us per launch, bounding roofline, algorithmic bytes
    or FLOP per launch (SURVEY §8d figures x B pairs), achieved, fraction of the gfx950 peak. Sorted by time per forward."""
    nn = float(N8) * N8
    spec = [
        # (stage, label, rocprof fragment, launches, bound, algorithmic work per launch)
        ("aggregate", "attention x V (gma.py:102-115): streams the [N x N] attention matrix of every pair once "
                      "(algorithmic bytes = fp32 storage, SURVEY 8d; the kernel stores 3 bytes per element)",
         "attn_v", ITERS, "hbm", nn * 4.0 * B),
        ("gru_zr", "fused z|r ConvGRU convolution, horizontal 1x5 pass (update.py:48-55), K = 5*384",
         "1, 5, SfGruZR", ITERS, "mfma", 2.0 * N8 * 256 * 1920 * B),
        ("gru_zr_v", "fused z|r ConvGRU convolution, vertical 5x1 pass (update.py:57-63), K = 5*384",
         "5, 1, SfGruZR", ITERS, "mfma", 2.0 * N8 * 256 * 1920 * B),
        ("gru_q", "q ConvGRU convolution, horizontal 1x5 pass, K = 5*384", "1, 5, SfGruQ", ITERS, "mfma", 2.0 * N8 * 128 * 1920 * B),
        ("gru_q_v", "q ConvGRU convolution, vertical 5x1 pass, K = 5*384", "5, 1, SfGruQ", ITERS, "mfma", 2.0 * N8 * 128 * 1920 * B),
        ("lookup", "correlation-pyramid lookup fused with convc1 (corr.py:32-53 + update.py:76-78); bytes = SURVEY's "
                   "lookup figure (<=400 cells read + 324 samples per pixel)", "lookup_conv_kernel", ITERS, "hbm", LOOKUP_BYTES * B),
        ("corr", "all-pairs correlation volume, level 0 (corr.py:55-63), K = 256, written in brick order (corr_bricks.hip; the stage "
                 "also holds the ~40 us brick_rows copy of the target features)", "corr_bricks_kernel", 1, "mfma", 2.0 * nn * 256 * B),
        ("attention", "Q K^T with the row softmax fused in, full-precision sweep (gma.py:60-74), K = 128",
         "qk_softmax_kernel<false", 1, "mfma", 2.0 * nn * 128 * B),
    ]
    rows = []
    for stage, label, frag, launches, bound, work in spec:
        ms = stages.get(stage, 0.0)
        if ms <= 0.0:
            continue
        us = ms * 1e3 / launches
        if bound == "hbm":
            ach, peak, unit = work / (us * 1e-6) / 1e9, PEAK_HBM_GBS, "GB/s"
        else:
            ach, peak, unit = work / (us * 1e-6) / 1e12, PEAK_F16_MFMA_TFLOPS, "TFLOP/s"
        row = {"stage": stage, "kernel": label, "rocprof_name_contains": frag, "launches_per_forward": launches,
               "us_per_launch": round(us, 2), "ms_per_forward": round(ms, 4), "bound": bound,
               "algorithmic_per_launch": work, "algorithmic_unit": "bytes" if bound == "hbm" else "FLOP",
               "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak}
        if stage == "aggregate":
            # the kernel is co-bound by its own matrix loop: 3 f16 MFMAs per product of the [N x N] x [N x 128] GEMM
            # (fast mode: 1), and it moves 3 bytes per element, not the 4 the algorithmic figure counts
            row["mfma_executed_tflops"] = MFMA_PER_PRODUCT_NOW[0] * 2.0 * nn * 128 * B / (us * 1e-6) / 1e12
            row["mfma_executed_frac"] = row["mfma_executed_tflops"] / PEAK_F16_MFMA_TFLOPS
            row["stored_bytes_per_launch"] = nn * 3.0 * B
            row["frac_stored_bytes"] = nn * 3.0 * B / (us * 1e-6) / 1e9 / PEAK_HBM_GBS
        if stage == "corr":
            # what the level-0 correlation must move: the fp32 volume written once + the two feature maps read once
            row["algorithmic_bytes_per_launch"] = (nn * 4.0 + 2.0 * N8 * 256 * 4.0) * B
        rows.append(row)
    rows.sort(key=lambda r: -r["ms_per_forward"])
    return rows


def pmc_traffic(kernel_fragment, batch, largest=False):
    """HBM-side bytes per launch of a kernel from the newest committed rocprofv3 PMC passes (profiles/*_pmc.json:
    2 x FETCH_SIZE + WRITE_SIZE, separate --pmc runs). Returns (bytes, file) or (None, None): NOT measured in this run."""
    import glob
    import re
    # newest by name, the numbers in a name compared as numbers (r05_v10 is newer than r05_v9)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")),
                   key=lambda f: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(f))])
    if not files:
        return None, None
    data = json.load(open(files[-1]))
    if data.get("_batch") != batch:
        return None, None
    for name, v in data.items():
        if kernel_fragment in name:
            key = "hbm_bytes_largest_launch" if largest and "hbm_bytes_largest_launch" in v else "hbm_bytes_per_launch"
            return v[key], "profiles/" + os.path.basename(files[-1])
    return None, None


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


def cpu_baseline(gsd, hsd, frames, budget_s=12.0, budget_1t_s=16.0):
    """The CPU oracle (a port of the reference's PyTorch-CPU op sequence) on this host's cores, bounded sample: all usable
    cores first (warm-up pair + up to 3 timed pairs), then ONE pair on a single thread (SURVEY 8d asks for both figures; a
    pair takes ~15 s on one core, so that leg is one pair, warmed by the all-cores leg). Returns (all_cores, one_thread)."""
    from oracle import clvo_ref, gma_ref
    cores = usable_cores()
    fr = frames.cpu()

    def run(threads, max_pairs, budget):
        torch.set_num_threads(threads)
        state = clvo_ref.zero_state(1)
        times = []
        t_all = time.time()
        for i in range(min(max_pairs, fr.shape[0] - 1)):
            t0 = time.time()
            _, up = gma_ref.gma_forward(gsd, fr[i:i + 1], fr[i + 1:i + 2], iters=ITERS)
            _, _, state = clvo_ref.clvo_forward(hsd, up, state)
            times.append(time.time() - t0)
            if time.time() - t_all > budget:
                break
        return times

    times = run(cores, 4, budget_s)
    timed = times[1:] if len(times) > 1 else times  # first pair is warm-up
    sec = float(np.median(timed))
    allc = {"value": 1.0 / sec, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
            "sample": "%d pair(s) after 1 warm-up, 376x1232, %d iters, fp32, torch CPU ops, median %.3f s/pair"
                      % (len(timed), ITERS, sec)}
    t1 = run(1, 1, budget_1t_s)
    one = {"value": 1.0 / t1[0], "unit": "frame-pairs/s", "cores": 1, "kind": "port",
           "sample": "1 pair on 1 thread (no separate warm-up: the all-cores leg ran first), 376x1232, %d iters, fp32, "
                     "torch CPU ops, %.3f s/pair" % (ITERS, t1[0])}
    torch.set_num_threads(cores)
    return allc, one


MFMA_PER_PRODUCT = {"split_f16": 3, "f16": 1, "f32": 1}   # MFMAs the engine executes per algorithmic product
MFMA_PER_PRODUCT_NOW = [3]                                 # ... of the precision this run's headline uses (set in main)
DTYPE_LABEL = {"split_f16": "f32 via 3xf16 split MFMA (fp32 accumulate)",
               "f16": "f16 operands, fp32 accumulate (fast mode, own tolerance)",
               "f32": "f32 (exact fp32 MFMA)"}


def launch_check(args, world, rank, local):
    """`--launch-check`: what a harness can run on any box (no GPU needed) to see that `python bench.py --gpus N` reaches N
    ranks that can talk to each other: gloo group over the launcher's rendezvous, one all-gather of (rank, local rank, CPU
    threads), one JSON line on rank 0. ATDN_LAUNCH_CHECK_FAIL_RANK=r makes rank r fail BEFORE the collective, the way a rank
    fails in the bench (through sharding.rendezvous: every rank raises, the job ends non-zero)."""
    assert world == args.gpus, "launched with %d rank(s) for --gpus %d" % (world, args.gpus)
    err = None
    if world > 1:
        dist.init_process_group("gloo")
    if os.environ.get("ATDN_LAUNCH_CHECK_FAIL_RANK") == str(rank):
        err = RuntimeError("launch check: rank %d told to fail" % rank)
    rendezvous(err)
    mine = torch.tensor([rank, local, torch.get_num_threads()], dtype=torch.int64)
    rows = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(rows, mine)
    else:
        rows = [mine]
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks": [int(r[0]) for r in rows],
                          "local_ranks": [int(r[1]) for r in rows], "cpu_threads_per_rank": [int(r[2]) for r in rows],
                          "launched_by": "bench.py itself" if "ATDN_SELF_LAUNCHED" in os.environ else "external launcher"
                          if "TORCHELASTIC_RUN_ID" in os.environ else "none (single process)"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=DEFAULT_BATCH, help="frame pairs per step per GPU")
    ap.add_argument("--streams", type=int, default=2, help="independent clips in flight per GPU (HIP streams)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d-leg", action="store_true", help="skip the second timed pass that ingests host frames")
    ap.add_argument("--precision", default="split_f16", choices=["split_f16", "f16", "f32"],
                    help="arithmetic of the flow network; the headline is split_f16 (fp32-grade). f16 = fast mode")
    ap.add_argument("--no-config3", action="store_true", help="skip the sharded-sequence leg (BASELINE configs[2])")
    ap.add_argument("--config3-frames", type=int, default=4541, help="frames of the synthetic sequence of the config3 leg")
    ap.add_argument("--no-f16-leg", action="store_true", help="skip the secondary timed pass in the f16 fast mode")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the secondary timed pass in the exact-fp32 mode")
    ap.add_argument("--no-per-frame-leg", action="store_true", help="skip the one-pair-per-call leg (the reference's per-frame call pattern)")
    ap.add_argument("--launch-check", action="store_true",
                    help="dry run of the launch plumbing only: every rank joins a gloo group, the ranks all-gather their "
                         "(rank, local rank, threads), rank 0 prints one JSON line; no GPU call is made")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # ATDN_BENCH_REHEARSAL=1: the N > 1 code path on a ONE-GPU box — every rank on cuda:0, gloo instead of RCCL (which
    # refuses two ranks on one device). Checks the sharding / gather / scan plumbing only; its numbers mean nothing.
    rehearsal = os.environ.get("ATDN_BENCH_REHEARSAL") == "1"
    if args.launch_check:
        return launch_check(args, world, rank, local)
    if rehearsal:
        local = 0
    # ATDN_BENCH_FORCE_DIST=1: a ONE-rank launch still initialises the process group and runs every collective of the N > 1
    # path (barriers, the max all-reduce, the all-gather of features) — on a one-GPU box that is RCCL itself at world size 1
    # (tests/test_gpu_bench_rehearsal.py); with ATDN_BENCH_REHEARSAL it is gloo
    dist_on = world > 1 or os.environ.get("ATDN_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert world == args.gpus, ("%d rank(s) for --gpus %d: start `python bench.py --gpus N` without a launcher (it starts its "
                                "own ranks) or with --nproc-per-node == --gpus" % (world, args.gpus))
    if world > 1:
        torch.set_num_threads(max(1, usable_cores() // world))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    B, K, Wm = args.batch, args.steps, args.warmup
    MFMA_PER_PRODUCT_NOW[0] = MFMA_PER_PRODUCT[args.precision]

    gsd = syn.to_torch(syn.make_gma_state(seed=1))
    hsd = syn.to_torch(syn.make_clvo_state(seed=1))
    # S pipelines on S streams: consecutive steps (clips) overlap, which fills the tails of each other's kernels
    S = max(1, args.streams)
    pipes = [OdometryPipeline(gsd, hsd, device=dev, max_batch=B, iters=ITERS, precision=args.precision) for _ in range(S)]
    ingests = [FrameIngest((H_KITTI, W_KITTI), max_frames=B + 1, device=dev) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    pipe = pipes[0]
    # Synthetic uint8 camera frames (what a KITTI png decodes to), different per rank (each rank owns its own stretch of
    # the sequence). Every pipeline walks its own long sequence clip by clip; the sequence is a 2B+1-frame clip played
    # forwards and backwards, so consecutive frames are always neighbours and each frame of the sequence passes through
    # the feature network once (continued clips reuse the shared frame). Laid out in sequence order, once in pinned
    # host memory (the H2D-inclusive leg) and once resident in HBM (the headline leg).
    clip = 2 * B + 1
    period = 2 * (clip - 1)
    base = torch.from_numpy(syn.make_frames(clip, H_KITTI, W_KITTI, seed=100 + rank)).round().clamp(0, 255).to(torch.uint8)
    order = [(k if k < clip else period - k) for k in range(period)] + list(range(B + 1))
    seq_host = base[order].contiguous().pin_memory()      # [period + B + 1, 3, 376, 1241] uint8
    seq_dev = seq_host.to(dev)
    torch.cuda.synchronize()

    calls = [0] * S
    active = [pipes]   # the pipelines step() drives (the f16 leg swaps in its own)

    def step(i, feats, slot=None, host=False):
        pipes = active[0]
        p = i % S
        j = calls[p]
        calls[p] += 1
        slot = i if slot is None else slot
        key = ((j + p) * B) % period
        with torch.cuda.stream(streams[p]):
            if host:   # uint8 frames in host memory: async H2D on the ingest's copy stream, then convert + resize
                frames = ingests[p](seq_host[key:key + B + 1])
            else:      # uint8 frames resident in HBM: the reference's per-frame resize to 376x1232, fused with the conversion
                frames = resize_frames(seq_dev[key:key + B + 1])
            f, _ = pipes[p].features_clip(frames, continued=(j > 0))   # B consecutive pairs of the sequence
            feats[slot * B:(slot + 1) * B] = f

    def join():
        for st_ in streams:
            torch.cuda.current_stream().wait_stream(st_)

    feats = torch.empty((max(K, Wm, 2 * S) * B, 512), device=dev)

    def timed(host, K=K):
        """W warm-up steps (at least two on EVERY pipeline: the graphs of a first and of a continued clip), then exactly
        K timed steps + the sequence tail, bracketed by barrier + synchronize. Returns (seconds, step_ms, tail_ms)."""
        nonlocal calls
        calls = [0] * S
        nw = max(Wm, 2 * S)
        err = None
        try:
            for i in range(nw):
                step(i, feats, slot=i % nw, host=host)
            calls = [0] * S   # the timed region starts a fresh sequence on every pipeline: nothing computed earlier is reused
            join()
            # warm the tail too, at the length the timed region scans (the recurrent scan graphs a length on second sight:
            # two calls, so that the timed scan REPLAYS — ADVICE r3: capture used to land inside the timed region)
            for _ in range(2):
                pipe.scan(torch.zeros((world * K * B, 512), device=dev))
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001 — one rank failing in warm-up must not leave the others in the barrier
            err = e
        # ... and the collective of the tail at the size the timed region uses (the first all-gather of a size sets up RCCL's
        # buffers and protocol for it: that belongs in the warm-up). Every rank takes part, failed or not (sharding.gather_features
        # carries a failure to every rank), so it doubles as the barrier that agrees on failure.
        if dist_on:
            gather_features(None if err is not None else feats[:K * B], world * K * B, error=err, device=dev)
            err = None
        # barrier that agrees on failure (sharding.rendezvous: one small all-gather of status rows; every rank raises if
        # any rank did), then the contract's plain barrier
        rendezvous(err, device=dev)
        if dist_on:
            dist.barrier()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        try:
            for i in range(K):
                step(i, feats, host=host)
                ev[i + 1].record(streams[i % S])
            join()
            torch.cuda.current_stream().synchronize()
        except Exception as e:   # noqa: BLE001 — carried through the all-gather below and raised on EVERY rank
            err = e
        t_tail = time.perf_counter()   # sequence tail: all-gather + ordered LSTM scan over all N*K*B features + rel2abs
        if dist_on:
            allf = gather_features(None if err is not None else feats[:K * B], world * K * B, error=err, device=dev)
        elif err is not None:
            raise err
        else:
            allf = feats[:K * B]
        rot, tr = pipe.scan(allf)
        poses = transforms.rel2abs(rot.cpu().numpy(), tr.cpu().numpy())
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        dt = time.perf_counter() - t0
        tail_ms = (time.perf_counter() - t_tail) * 1e3
        assert tuple(poses.shape) == (world * K * B + 1, 4, 4) and bool(torch.isfinite(poses).all())
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        if dist_on:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # completion-to-completion intervals on the launch streams (steps overlap when S > 1)
        step_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(S, K)] if K > S else [dt * 1e3 / K]
        return float(tmax.item()), step_ms, tail_ms

    dt, step_ms, tail_ms = timed(host=False)
    h2d = None if args.no_h2d_leg else timed(host=True)

    # ---- config3: ONE long sequence sharded over the ranks (evaluate_odometry.py:60-75: the LSTM state is never reset
    # inside a sequence), uint8 frames in pinned host memory, through the product's sequence driver; both streams of every
    # GPU walk their own contiguous sub-shard (lanes). Timed host-clock, barrier + synchronize on both sides, max over ranks.
    c3 = None
    if not args.no_config3:
        T3 = max(2, args.config3_frames)
        cyc = CyclicSequence(seq_host, period, T3)
        lanes = pipes[1:]
        # warm every (clip length, continued) graph the plan needs, on every lane, outside the timed region
        from atdn_vslam_amd.sharding import clip_plan, lane_ranges, shard_range
        lo3, hi3 = shard_range(T3 - 1, rank, world)
        err3 = None
        try:
            for lane, (a, b) in enumerate(lane_ranges(lo3, hi3, B, S)):
                need = sorted({(e - s, c) for (s, e, c) in clip_plan(a, b, B)}, key=lambda t: (t[1], -t[0]))
                with torch.cuda.stream(streams[lane]):
                    for (n, cont) in need:
                        pipes[lane].features_clip(resize_frames(seq_dev[:n + 1]), continued=cont)
            join()
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001
            err3 = e
        rendezvous(err3, device=dev)
        if dist_on:
            dist.barrier()
        timing = {}
        t0 = time.perf_counter()
        poses3 = pipe.run_sequence(cyc, batch=B, lanes=lanes, timing=timing)
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        dt3 = time.perf_counter() - t0
        assert tuple(poses3.shape) == (T3, 4, 4) and bool(torch.isfinite(poses3).all())
        t3 = torch.tensor([dt3, timing["encode_s"], timing["gather_s"], timing["scan_s"]], device=dev, dtype=torch.float64)
        if dist_on:
            dist.all_reduce(t3, op=dist.ReduceOp.MAX)
        c3 = [float(x) for x in t3.tolist()] + [T3]

    # ---- f16 fast mode as a secondary leg: same steps, same frames, precision="f16" handles (never the headline)
    f16 = None
    if not args.no_f16_leg and args.precision == "split_f16":
        fast = [OdometryPipeline(gsd, hsd, device=dev, max_batch=B, iters=ITERS, precision="f16") for _ in range(S)]
        frames0 = resize_frames(seq_dev[:B + 1])
        _, up_ref = pipe.features_clip(frames0)
        _, up_fast = fast[0].features_clip(frames0)
        d = (up_fast - up_ref).abs()
        f16_err = (float(d.max()), float(d.mean()), float(up_ref.abs().max()))
        active[0] = fast
        f16 = timed(host=False) + (f16_err,)
        active[0] = pipes
        del fast

    # ---- exact-fp32 mode as a secondary leg: what `RAFTGMA(saturation_fallback=True)` (modules.py) drops to when a checkpoint's
    # activations leave the split-f16 range — every GEMM on v_mfma_f32_32x32x2_f32, the reference's data flow (row-major
    # pyramid, separate lookup, logits + softmax). Same clip and frames, a quarter of the steps (never the headline).
    f32 = None
    if not args.no_f32_leg and args.precision == "split_f16":
        exact = [OdometryPipeline(gsd, hsd, device=dev, max_batch=B, iters=ITERS, precision="f32") for _ in range(S)]
        frames0 = resize_frames(seq_dev[:B + 1])
        _, up_ref = pipe.features_clip(frames0)
        _, up_exact = exact[0].features_clip(frames0)
        d = (up_exact - up_ref).abs()
        f32_err = (float(d.max()), float(d.mean()), float(up_ref.abs().max()))
        K32 = max(2, K // 4)
        active[0] = exact
        f32 = timed(host=False, K=K32) + (f32_err, K32)
        active[0] = pipes
        del exact

    # ---- the reference's own call pattern as a secondary leg: ONE frame per call (NeuralSLAM.__call__ in odometry mode,
    # neural_slam.py:192-227 = pipeline.VisualOdometry): host uint8 frame -> H2D -> resize -> flow (one pair, the low-latency form) ->
    # head (one LSTM step) -> pose on the host, synchronous; rank 0 only, never the headline
    per_frame = None
    if not args.no_per_frame_leg and rank == 0 and args.precision == "split_f16":
        from atdn_vslam_amd.pipeline import VisualOdometry
        vo = VisualOdometry(gsd, hsd, device=dev, iters=ITERS)
        nf = 24
        for k in range(4):
            vo(seq_host[k])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(4, 4 + nf):
            pose = vo(seq_host[k])
        torch.cuda.synchronize()
        per_frame = ((time.perf_counter() - t0) / nf, nf)
        assert bool(torch.isfinite(pose).all())
        del vo

    if rank == 0:
        total_pairs = world * K * B
        # per-stage device time of the same forward in this run, eager with HIP events on the launch stream
        reps = 3
        # in the call form the timed loop uses: a CONTINUED clip (B feature-network passes, not the 2B of pair mode)
        st = pipe.flow_net.profile(H, W, B, iters=ITERS, reps=reps, mode="continued")
        rows = kernel_table(st, B)
        # HBM-side bytes per launch from the committed PMC passes (NOT this run: `traffic_source` names the file) on every row
        # whose kernel the profile holds, and their ratio to the row's algorithmic bytes — the lookup and the level-0
        # correlation are the two kernels that move more than they must
        for r in rows:
            # (the correlation kernel runs once per pyramid level: level 0 is its largest launch)
            tb, src = pmc_traffic(r["rocprof_name_contains"], B, largest=(r["stage"] == "corr"))
            if tb is not None:
                r["traffic"], r["traffic_source"] = tb, src
                ab = r.get("algorithmic_bytes_per_launch", r["algorithmic_per_launch"] if r["bound"] == "hbm" else None)
                if ab:
                    r["traffic_ratio"] = tb / ab
        top = rows[0]
        traffic, traffic_src = pmc_traffic(top["rocprof_name_contains"], B)
        fwd_ms = float(np.median(step_ms))
        mfma_x = MFMA_PER_PRODUCT[args.precision]
        out = {
            "metric": "frame-pairs/sec, KITTI 1241x376 odometry inference at 1/2/4/8 MI355X",
            "value": total_pairs / dt, "unit": "frame-pairs/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": dt * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_LABEL[args.precision], "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo)" if rehearsal else ""),
            "config": {"workload": "KITTI seq-03-shaped 376x1241 uint8 frames (resident in HBM) resized to 376x1232, GMA flow "
                                   "12 GRU iters + CLVO head -> 6-DoF trajectory (BASELINE configs[1])",
                       "pairs_per_step_per_gpu": B, "streams_per_gpu": S, "gru_iters": ITERS, "parallelism": "pairs sharded x%d, one "
                       "all-gather of 512-d features, replicated LSTM scan" % world},
            # the dominant kernel = the top row of `kernels` (same-run HIP-event time per forward). For an hbm-bound kernel
            # achieved = algorithmic bytes / launch time; for an mfma-bound one algorithmic FLOP / launch time (the split-f16
            # engine executes 3 f16 MFMAs per algorithmic product: mfma_executed_*). `traffic` is NOT measured in this run:
            # it is the PMC figure (2 x FETCH_SIZE + WRITE_SIZE per launch) of the committed profile named in traffic_source.
            "roofline": {"bound": top["bound"], "kernel": top["kernel"], "rocprof_name_contains": top["rocprof_name_contains"],
                         "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
                         "traffic": traffic, "traffic_source": traffic_src, "launch_ms": top["us_per_launch"] / 1e3,
                         "algorithmic_per_launch": top["algorithmic_per_launch"],
                         "algorithmic_unit": top["algorithmic_unit"],
                         # share of the eager stage sum of one continued-clip forward (one stream, events between launches)
                         # and of the timed step (two streams overlapping: completion-to-completion interval)
                         "share_of_forward": top["ms_per_forward"] / sum(st.values()),
                         "share_of_timed_step": top["ms_per_forward"] / fwd_ms},
            "kernels": rows[:5] + [r for r in rows[5:] if r["stage"] in ("lookup", "corr")],
            "stages_mode": "continued clip (B feature-network passes per B pairs): the call form the timed loop uses",
            "forward": {"ms_per_batch_median": fwd_ms,
                        "tflops_algorithmic_0.951_per_pair": FLOP_PER_PAIR * B / (fwd_ms * 1e-3) / 1e12,
                        "tflops_executed_work": FLOP_PER_PAIR_EXECUTED * B / (fwd_ms * 1e-3) / 1e12,
                        "mfma_executed_tflops": mfma_x * FLOP_PER_PAIR_EXECUTED * B / (fwd_ms * 1e-3) / 1e12,
                        "frac_of_f16_mfma_peak_algorithmic": FLOP_PER_PAIR * B / (fwd_ms * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS,
                        "frac_of_f32_mfma_peak": FLOP_PER_PAIR * B / (fwd_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS},
            "lookup": {"ms_per_launch": st["lookup"] / ITERS, "achieved_GBps": LOOKUP_BYTES * B / (st["lookup"] / ITERS * 1e-3) / 1e9,
                       "peak_GBps": PEAK_HBM_GBS},
            "stages_ms_per_forward": {k: round(v, 4) for k, v in st.items()},
            # inside the timed region, after the K steps: grows with N (every rank scans all N*K*B features in order)
            "sequence_tail_ms": round(tail_ms, 3),
        }
        out["stages_ms_per_forward"]["attention_total"] = round(st["attention"] + st["attn_logits"], 4)
        for k in ("mfma_executed_tflops", "mfma_executed_frac", "stored_bytes_per_launch", "frac_stored_bytes"):
            if k in top:
                out["roofline"][k] = top[k]
        if traffic is not None and top["bound"] == "hbm":
            # PMC bytes (committed profile) over this run's launch time: the fraction of the HBM peak in REAL bytes
            out["roofline"]["frac_real_bytes"] = traffic / (top["us_per_launch"] * 1e-6) / 1e9 / PEAK_HBM_GBS
        if c3 is not None:
            out["config3"] = {"workload": "ONE synthetic KITTI-00-shaped sequence of %d uint8 frames (pinned host memory) sharded over %d "
                                          "rank(s), OdometryPipeline.run_sequence: ingest + flow + head, one all-gather, replicated "
                                          "scan, rel2abs (BASELINE configs[2])" % (c3[4], world),
                              "pairs": c3[4] - 1, "value": (c3[4] - 1) / c3[0], "unit": "frame-pairs/s", "wall_s": c3[0],
                              "encode_s": c3[1], "allgather_ms": c3[2] * 1e3, "scan_rel2abs_ms": c3[3] * 1e3,
                              "lanes_per_gpu": S, "ratio_to_value": ((c3[4] - 1) / c3[0]) / (total_pairs / dt)}
        if f16 is not None:
            out["f16_fast"] = {"value": total_pairs / f16[0], "unit": "frame-pairs/s", "ms_per_step": f16[0] * 1e3 / K,
                               "ratio_to_value": (total_pairs / f16[0]) / (total_pairs / dt),
                               "dtype": DTYPE_LABEL["f16"],
                               "flow_up_abs_diff_vs_split_f16_px": {"max": f16[3][0], "mean": f16[3][1], "max_abs_flow": f16[3][2],
                                                                     "sample": "%d pairs of the bench clip, 12 iterations" % B}}
        if f32 is not None:
            n32 = world * f32[4] * B
            out["f32_exact"] = {"value": n32 / f32[0], "unit": "frame-pairs/s", "ms_per_step": f32[0] * 1e3 / f32[4], "steps": f32[4],
                                "ratio_to_value": (n32 / f32[0]) / (total_pairs / dt), "dtype": DTYPE_LABEL["f32"],
                                "role": "the mode RAFTGMA(saturation_fallback=True) switches to when the split-f16 range guard trips",
                                "flow_up_abs_diff_vs_split_f16_px": {"max": f32[3][0], "mean": f32[3][1], "max_abs_flow": f32[3][2],
                                                                     "sample": "%d pairs of the bench clip, 12 iterations" % B}}
        if per_frame is not None:
            out["per_frame"] = {"value": 1.0 / per_frame[0], "unit": "frame-pairs/s", "ms_per_frame": per_frame[0] * 1e3, "frames": per_frame[1],
                                "workload": "pipeline.VisualOdometry: one host uint8 frame per call -> pose on the host, synchronous "
                                            "(NeuralSLAM.__call__'s odometry branch, neural_slam.py:192-227); flow network in its "
                                            "low-latency form (one pair per launch)",
                                "ratio_to_value": (1.0 / per_frame[0]) / (total_pairs / dt)}
        if h2d is not None:
            # second timed pass of the same K steps with the uint8 frames in pinned HOST memory: H2D (copy stream,
            # double-buffered) + convert + resize inside the timed region. Never the headline `value`.
            out["h2d_inclusive"] = {"value": total_pairs / h2d[0], "unit": "frame-pairs/s", "ms_per_step": h2d[0] * 1e3 / K,
                                    "ratio_to_value": (total_pairs / h2d[0]) / (total_pairs / dt),
                                    "host_bytes_per_step": (B + 1) * 3 * H_KITTI * W_KITTI}
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N = 1 only (rank 0)
            out["cpu_baseline"], out["cpu_baseline_1thread"] = cpu_baseline(gsd, hsd, resize_frames(seq_dev[:6]))
        else:
            out["cpu_baseline"] = None
            out["cpu_baseline_reason"] = ("--no-cpu-baseline" if world == 1 else "the CPU leg is timed at N = 1 only (rank 0 "
                                          "would hold the other %d ranks in the closing barrier and share its cores with them); "
                                          "the N = 1 line of the same build carries it" % (world - 1))
        print(json.dumps(out))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
