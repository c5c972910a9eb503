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
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
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
