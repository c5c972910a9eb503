"""Worker for tests/test_sharding.py: runs the sharded odometry driver on `gloo` CPU tensors. The per-pair
feature function and the scan are the CPU oracle (tests may use it); what is under test is the driver:
shard ranges, ragged/empty shards, gather order and rank-identical results."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from atdn_vslam_amd import synthetic as syn  # noqa: E402
from atdn_vslam_amd.sharding import clip_plan, gather_features, shard_range, sharded_odometry, sharded_sequence  # noqa: E402
from oracle import clvo_ref  # noqa: E402


def main():
    out_dir = sys.argv[1]
    n_pairs = int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.set_num_threads(2)
    hsd = syn.to_torch(syn.make_clvo_state(seed=1))
    flows = torch.from_numpy(syn.make_flow(n_pairs, 376, 1232, seed=31))  # stands in for the per-pair flow

    calls = []

    def encode(lo, hi):
        calls.append((lo, hi))
        if hi == lo:
            return torch.zeros((0, 512))
        return clvo_ref.clvo_encode(hsd, flows[lo:hi])

    def scan(feats):
        state = clvo_ref.zero_state(1)
        rots, trs = [], []
        for t in range(feats.shape[0]):
            r, x, state = clvo_ref.clvo_step(hsd, feats[t:t + 1], state)
            rots.append(r)
            trs.append(x)
        return torch.cat(rots), torch.cat(trs)

    rot, tr = sharded_odometry(n_pairs, encode, scan)
    assert calls == [shard_range(n_pairs, rank, world)]
    # gather_features alone, with a recognisable payload
    lo, hi = shard_range(n_pairs, rank, world)
    tag = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 3)
    full = gather_features(tag, n_pairs)
    assert torch.equal(full[:, 0], torch.arange(n_pairs, dtype=torch.float32))
    # the sequence driver: clips of `batch` pairs inside each shard; a shard that does not start on a multiple of the
    # clip length begins with a non-continued clip, every later clip of the shard continues the previous one
    batch = 4
    seen = []

    def encode_clip(s, e, continued):
        seen.append((s, e, continued))
        assert 0 < e - s <= batch
        return clvo_ref.clvo_encode(hsd, flows[s:e])     # pair p = flow p (frames p, p + 1)

    rot2, tr2 = sharded_sequence(n_pairs + 1, encode_clip, scan, batch)
    assert seen == clip_plan(lo, hi, batch)
    assert all(c == (s > lo) for (s, e, c) in seen) and (not seen or (seen[0][0] == lo and seen[-1][1] == hi))
    # (the CPU oracle's convolutions round differently for different batch sizes: equal to fp32 noise, not bit for bit)
    assert float((rot2 - rot).abs().max()) < 1e-6 and float((tr2 - tr).abs().max()) < 1e-6
    # two lanes per rank (two HIP streams on a GPU): the shard is cut into contiguous sub-ranges walked round-robin; every
    # lane's first clip is not continued, the concatenated features are in sequence order, results unchanged
    from atdn_vslam_amd.sharding import lane_ranges
    seen2, joined = [], []

    def encode_clip_lane(s, e, continued, lane):
        seen2.append((s, e, continued, lane))
        return clvo_ref.clvo_encode(hsd, flows[s:e])

    encode_clip_lane.join = lambda: joined.append(len(seen2))
    timing = {}
    rot3, tr3 = sharded_sequence(n_pairs + 1, encode_clip_lane, scan, 2, lanes=2, timing=timing)
    ranges = lane_ranges(lo, hi, 2, 2)
    for lane, (a, b) in enumerate(ranges):
        mine = [(s, e, c) for (s, e, c, l) in seen2 if l == lane]
        assert mine == clip_plan(a, b, 2)
    assert sorted(p for (s, e, _, _) in seen2 for p in range(s, e)) == list(range(lo, hi))
    assert joined == ([len(seen2)] if hi > lo else [0]) or (hi == lo and joined == [])
    assert float((rot3 - rot).abs().max()) < 1e-6 and float((tr3 - tr).abs().max()) < 1e-6
    assert timing["local_pairs"] == hi - lo and timing["encode_s"] >= 0 and timing["scan_s"] > 0
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), rot=rot.numpy(), tr=tr.numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
