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


class ShardError(RuntimeError):
    """Raised on EVERY rank when some rank's local work failed before a collective: `rank` is the first failing rank,
    `remote_type` / `remote_message` the class name and text of its exception (on that rank itself the original exception
    is chained as __cause__)."""

    def __init__(self, rank, remote_type, remote_message):
        super().__init__("rank %d failed before the all-gather: %s: %s" % (rank, remote_type, remote_message))
        self.rank, self.remote_type, self.remote_message = rank, remote_type, remote_message


# The status row that travels INSIDE the padded block of the one all-gather (no extra collective): [0] = 1.0 if the rank's
# local work raised, [1] = number of message bytes, [2:] = "ClassName: text" as one byte value per float. A rank that failed
# still joins the collective with a zero block, so no peer is left waiting in it (the reference has no collectives,
# neural_slam.py:51; this contract is the build's own).
def _status_row(error, width, dtype, device):
    row = torch.zeros((1, width), dtype=dtype, device="cpu")
    if error is not None and width >= 1:
        row[0, 0] = 1.0
        msg = ("%s: %s" % (type(error).__name__, error)).encode("utf-8", "replace")[: max(0, width - 2)]
        if width >= 2:
            row[0, 1] = float(len(msg))
            if msg:
                row[0, 2:2 + len(msg)] = torch.tensor(list(msg), dtype=dtype)
    return row.to(device)


def _first_failure(status_rows):
    """status_rows [world, width] (host) -> None, or (rank, type, message) of the first rank whose flag is set."""
    flags = status_rows[:, 0]
    for r in range(status_rows.shape[0]):
        if float(flags[r]) != 0.0:
            n = int(status_rows[r, 1]) if status_rows.shape[1] >= 2 else 0
            raw = bytes(int(v) & 0xFF for v in status_rows[r, 2:2 + n].tolist())
            text = raw.decode("utf-8", "replace")
            typ, _, msg = text.partition(": ")
            return r, (typ or "Exception"), msg
    return None


def _raise_agreed(status_rows, rank, error):
    hit = _first_failure(status_rows)
    if hit is None:
        return
    err = ShardError(*hit)
    if error is not None:
        raise err from error     # this rank failed too: keep its own traceback
    raise err


def rendezvous(error=None, group=None, device="cpu"):
    """Barrier that also agrees on failure: every rank contributes a status row to ONE small all-gather and, if any rank
    passed an exception, EVERY rank raises ShardError naming the first failing rank (instead of the healthy ranks waiting in
    the next collective until the watchdog fires). Without a process group: re-raises `error`, if any."""
    if not (dist.is_available() and dist.is_initialized()):
        if error is not None:
            raise error
        return
    world = dist.get_world_size(group)
    row = _status_row(error, 256, torch.float32, device)
    out = torch.empty((world, 256), dtype=torch.float32, device=row.device)
    dist.all_gather_into_tensor(out, row, group=group)
    _raise_agreed(out.cpu(), dist.get_rank(group), error)


FEATURE_DIM, FEATURE_DTYPE = 512, torch.float32   # the block every rank of the sequence drivers contributes: [*, 512] fp32


def gather_features(local, n_pairs, group=None, error=None, device=None, dim=None, dtype=None):
    """local [p_r, D] (this rank's shard, in sequence order) -> [n_pairs, D] in global sequence order on every
    rank. One all_gather of equally sized (padded) blocks; ragged and empty shards are handled.
    error: the exception this rank's local work raised, or None. A failed rank passes it (with local=None) and still joins
    the collective; its status travels in one extra row of the SAME padded block, and every rank then raises ShardError with
    the first failing rank's message. Nobody blocks in the all-gather because a peer raised before reaching it.
    device / dim / dtype: where, how wide and of which type a rank WITHOUT a local tensor builds its block (default: the
    [*, 512] fp32 contract of the pose head's features). Every rank's block must have ONE shape and type or the collective
    hangs or corrupts — the very thing this path exists to prevent — so when dim / dtype are given (the sequence drivers pass
    them) a healthy rank's `local` is checked against them BEFORE the collective, and a mismatch travels as that rank's
    error."""
    import os
    # (ATDN_FORCE_COLLECTIVE=1: a one-rank group still goes through the collective — a plumbing test of the backend)
    if not (dist.is_available() and dist.is_initialized()) or \
            (dist.get_world_size(group) == 1 and os.environ.get("ATDN_FORCE_COLLECTIVE") != "1"):
        if error is not None:
            raise error
        assert local.shape[0] == n_pairs
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(n_pairs, rank, world)
    width = -(-n_pairs // world)  # ceil
    want_D = int(dim) if dim is not None else None
    if error is None and local is not None:
        # checked here, on the healthy path, so that a wrong block becomes a carried error instead of a mismatched collective
        if local.dim() != 2 or local.shape[0] != hi - lo:
            error = ValueError("shard of rank %d has shape %s, shard_range says %d rows" % (rank, tuple(local.shape), hi - lo))
        elif (want_D is not None and local.shape[1] != want_D) or (dtype is not None and local.dtype != dtype):
            error = ValueError("features of rank %d are %s %s, the gather contract is [*, %s] %s"
                               % (rank, tuple(local.shape), local.dtype, want_D, dtype))
    if error is not None or local is None:
        # a failed (or empty-handed) rank: a zero block of the agreed shape, NOT of whatever its half-finished tensor has
        D = want_D if want_D is not None else (local.shape[1] if local is not None and local.dim() == 2 else FEATURE_DIM)
        dtype = dtype if dtype is not None else (local.dtype if local is not None else FEATURE_DTYPE)
        dev = local.device if local is not None else torch.device(device if device is not None else "cpu")
        block = torch.zeros((width + 1, D), dtype=dtype, device=dev)
    else:
        D, dtype, dev = local.shape[1], local.dtype, local.device
        block = torch.zeros((width + 1, D), dtype=dtype, device=dev)
        block[: hi - lo] = local
    block[width:] = _status_row(error, D, dtype, dev)
    out = torch.empty((world * (width + 1), D), dtype=dtype, device=dev)
    dist.all_gather_into_tensor(out, block, group=group)
    blocks = out.view(world, width + 1, D)
    _raise_agreed(blocks[:, width, :].float().cpu(), rank, error)   # (one small D2H per sequence: world x D floats)
    parts = []
    for r in range(world):
        a, b = shard_range(n_pairs, r, world)
        parts.append(blocks[r, : b - a])
    return torch.cat(parts, dim=0)


def sharded_odometry(n_pairs, encode_pairs, scan, group=None, device=None):
    """Runs a sequence of `n_pairs` frame pairs over the ranks of `group`.

    encode_pairs(start, stop) -> [stop-start, 512] features of pairs start..stop-1 (flow + CNN encoder; local)
    scan(features [P,512])    -> (rot [P,3], tr [P,3])   the ordered LSTM/MLP tail from a zero state
    Returns (rot, tr) for the whole sequence, identical on every rank. If encode_pairs raises on any rank, every rank
    raises ShardError after the all-gather (gather_features); `device` is where a failed rank builds its empty block.
    """
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_range(n_pairs, rank, world)
    local, error = None, None
    try:
        local = encode_pairs(lo, hi)
    except Exception as e:   # noqa: BLE001 — carried to every rank through the gather
        error = e
    feats = gather_features(local, n_pairs, group, error=error, device=device, dim=FEATURE_DIM, dtype=FEATURE_DTYPE)
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
    frame ingest, flow, CNN encoder); clips of one shard are requested in order. scan as in sharded_odometry. If it raises
    on any rank, EVERY rank raises ShardError after the all-gather (no rank is left blocked in the collective).
    lanes > 1: the rank's shard is cut into `lanes` contiguous sub-ranges (lane_ranges) that are walked round-robin, one clip
    at a time, as encode_clip(first_pair, stop_pair, continued, lane) — independent streams of one GPU, each with its own
    handles; `encode_clip.join()` (if present) is called once after the last clip, `encode_clip.finish()` (if present) after
    the whole shard, still before the gather. `continued` refers to the lane's own
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
    # an exception of this rank's shard (a bad frame, out of memory, SplitF16RangeError of the saturation guard) must not
    # leave the other ranks waiting in the all-gather: it is carried through it and raised on EVERY rank (gather_features)
    local, error = None, None
    dev = getattr(encode_clip, "device", "cpu")
    try:
        local = encode_pairs(lo, hi)
        if local is None:   # empty shard: a [0, 512] block on the device / dtype the scan expects
            local = torch.zeros((0, 512), dtype=torch.float32, device=dev)
        if hasattr(encode_clip, "finish"):
            encode_clip.finish()   # end-of-shard checks that may raise (the saturation guard of every lane)
    except Exception as e:   # noqa: BLE001
        local, error = None, e
        if hasattr(encode_clip, "join"):
            try:
                encode_clip.join()   # lane streams of the failed walk: back in order behind the main stream
            except Exception:        # noqa: BLE001
                pass
    if timing is not None:
        try:
            fence()   # (a device error of the failed walk re-raised by this synchronisation must not skip the gather either)
        except Exception as e:   # noqa: BLE001
            local, error = None, (error or e)
        t1 = time.perf_counter()
    feats = gather_features(local, n_pairs, group, error=error, device=dev, dim=FEATURE_DIM, dtype=FEATURE_DTYPE)
    if timing is not None:
        fence()
        t2 = time.perf_counter()
    out = scan(feats)
    if timing is not None:
        fence()
        timing.update(encode_s=t1 - t0, gather_s=t2 - t1, scan_s=time.perf_counter() - t2, local_pairs=hi - lo)
    return out
