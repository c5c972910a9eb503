"""Frame-pair sharding of a sequence over the GPUs of one node.

The flow network and the CLVO CNN encoder depend on one frame pair only, so pairs shard
embarrassingly: rank r owns the contiguous range `shard_range(P, r, G)`. The head's LSTM tail
(odometry/network.py:137-140) is a recurrence over the WHOLE sequence whose state the reference
never resets inside a sequence (evaluate_odometry.py:60-75), so the ranks exchange exactly one
thing: an all-gather of the per-pair 512-d features (RCCL over xGMI; [P,512] fp32 = 9.3 MB for
KITTI-00), after which every rank runs the same ordered scan and holds the full 6-DoF trajectory.
Backend-agnostic: the same code runs on `gloo` CPU tensors in the tests.
"""
import torch
import torch.distributed as dist


def shard_range(n_pairs, rank, world):
    """Contiguous, balanced split: the first (n_pairs % world) ranks get one extra pair."""
    base, extra = divmod(n_pairs, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_features(local, n_pairs, group=None):
    """local [p_r, D] (this rank's shard, in sequence order) -> [n_pairs, D] in global sequence order on every
    rank. One all_gather of equally sized (padded) blocks; ragged and empty shards are handled."""
    import os
    # (ATDN_FORCE_COLLECTIVE=1: a one-rank group still goes through the collective — a plumbing test of the backend)
    if not (dist.is_available() and dist.is_initialized()) or \
            (dist.get_world_size(group) == 1 and os.environ.get("ATDN_FORCE_COLLECTIVE") != "1"):
        assert local.shape[0] == n_pairs
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(n_pairs, rank, world)
    assert local.shape[0] == hi - lo, "shard length does not match shard_range"
    width = -(-n_pairs // world)  # ceil
    block = torch.zeros((width, local.shape[1]), dtype=local.dtype, device=local.device)
    block[: hi - lo] = local
    out = torch.empty((world * width, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, block, group=group)
    parts = []
    for r in range(world):
        a, b = shard_range(n_pairs, r, world)
        parts.append(out[r * width: r * width + (b - a)])
    return torch.cat(parts, dim=0)


def sharded_odometry(n_pairs, encode_pairs, scan, group=None):
    """Runs a sequence of `n_pairs` frame pairs over the ranks of `group`.

    encode_pairs(start, stop) -> [stop-start, 512] features of pairs start..stop-1 (flow + CNN encoder; local)
    scan(features [P,512])    -> (rot [P,3], tr [P,3])   the ordered LSTM/MLP tail from a zero state
    Returns (rot, tr) for the whole sequence, identical on every rank.
    """
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_range(n_pairs, rank, world)
    local = encode_pairs(lo, hi)
    feats = gather_features(local, n_pairs, group)
    return scan(feats)


def clip_plan(lo, hi, batch):
    """Clips of at most `batch` consecutive pairs covering pairs [lo, hi) of a sequence, in order:
    [(first_pair, stop_pair, continued)]. Clip (s, e) needs frames s..e inclusive; `continued` is True when frame s was
    the last frame of the previous clip of the SAME shard (its features can be reused). A shard need not start on a
    multiple of `batch`: its first clip is simply not continued."""
    return [(s, min(s + batch, hi), s > lo) for s in range(lo, hi, batch)]


def lane_ranges(lo, hi, batch, lanes):
    """Splits the shard [lo, hi) into at most `lanes` contiguous sub-ranges for independent streams of one GPU: every
    sub-range but the last is a multiple of `batch` long (so only one lane ends on a short clip). Empty lanes are dropped."""
    n = hi - lo
    lanes = max(1, int(lanes))
    clips = -(-n // batch) if n > 0 else 0            # ceil
    out, s = [], lo
    for k in range(lanes):
        c = clips // lanes + (1 if k < clips % lanes else 0)
        e = min(hi, s + c * batch)
        if e > s:
            out.append((s, e))
        s = e
    return out


def sharded_sequence(n_frames, encode_clip, scan, batch, group=None, lanes=1, timing=None):
    """The sequence driver: frames 0..n_frames-1 -> (rot, tr) of the n_frames-1 pairs, identical on every rank.

    encode_clip(first_pair, stop_pair, continued) -> [stop_pair-first_pair, 512] features of that clip (local work:
    frame ingest, flow, CNN encoder); clips of one shard are requested in order. scan as in sharded_odometry.
    lanes > 1: the rank's shard is cut into `lanes` contiguous sub-ranges (lane_ranges) that are walked round-robin, one clip
    at a time, as encode_clip(first_pair, stop_pair, continued, lane) — independent streams of one GPU, each with its own
    handles; `encode_clip.join()` (if present) is called once after the last clip. `continued` refers to the lane's own
    previous clip. timing: a dict that receives host-clock seconds of the three phases (`encode_s`, `gather_s`, `scan_s`;
    needs `encode_clip.sync()` to fence the device between them)."""
    import time

    def encode_pairs(lo, hi):
        if lanes <= 1:
            parts = [encode_clip(s, e, c) for (s, e, c) in clip_plan(lo, hi, batch)]
        else:
            plans = [clip_plan(a, b, batch) for (a, b) in lane_ranges(lo, hi, batch, lanes)]
            got = [[] for _ in plans]
            for k in range(max((len(p) for p in plans), default=0)):
                for lane, p in enumerate(plans):
                    if k < len(p):
                        got[lane].append(encode_clip(p[k][0], p[k][1], p[k][2], lane))
            if hasattr(encode_clip, "join"):
                encode_clip.join()
            parts = [f for g in got for f in g]
        if not parts:
            return None
        return torch.cat(parts, dim=0)

    fence = getattr(encode_clip, "sync", lambda: None)
    n_pairs = max(n_frames - 1, 0)
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_range(n_pairs, rank, world)
    t0 = time.perf_counter()
    local = encode_pairs(lo, hi)
    if local is None:   # empty shard: a [0, 512] block on the device / dtype the scan expects
        local = torch.zeros((0, 512), dtype=torch.float32, device=getattr(encode_clip, "device", "cpu"))
    if timing is not None:
        fence()
        t1 = time.perf_counter()
    feats = gather_features(local, n_pairs, group)
    if timing is not None:
        fence()
        t2 = time.perf_counter()
    out = scan(feats)
    if timing is not None:
        fence()
        timing.update(encode_s=t1 - t0, gather_s=t2 - t1, scan_s=time.perf_counter() - t2, local_pairs=hi - lo)
    return out
