"""Worker for tests/test_sharding.py::test_a_failing_rank_raises_on_every_rank: one rank's local work raises before the
all-gather; every rank must raise ShardError within seconds (nobody blocks in the collective), and the process group must
still be usable afterwards. gloo, CPU tensors."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from atdn_vslam_amd.modules import SplitF16RangeError  # noqa: E402
from atdn_vslam_amd.sharding import ShardError, gather_features, rendezvous, shard_range, sharded_odometry, sharded_sequence  # noqa: E402


def main():
    out_dir, bad_rank, kind = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_pairs = int(sys.argv[4]) if len(sys.argv) > 4 else 4 * world + 1          # default: ragged, the first rank has one pair more
    batch = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    exc = SplitF16RangeError if kind == "SplitF16RangeError" else RuntimeError
    text = "synthetic failure of rank %d (%s)" % (bad_rank, kind)
    report = {"rank": rank}

    def scan(feats):
        return feats[:, :3].clone(), feats[:, 3:6].clone()

    def make_encode(fail):
        calls = []

        def encode_clip(s, e, continued, lane=0):
            calls.append((s, e))
            if fail and rank == bad_rank and len(calls) == 2:      # second clip of the failing rank
                raise exc(text)
            return torch.arange(s, e, dtype=torch.float32)[:, None].repeat(1, 512)
        return encode_clip

    # 1) the sequence driver (one lane, then two lanes)
    for lanes in (1, 2):
        t0 = time.time()
        try:
            sharded_sequence(n_pairs + 1, make_encode(True), scan, batch, lanes=lanes)
            report["seq%d" % lanes] = "no error"
        except ShardError as e:
            report["seq%d" % lanes] = {"rank": e.rank, "type": e.remote_type, "msg": e.remote_message, "s": time.time() - t0,
                                       "cause": type(e.__cause__).__name__ if e.__cause__ is not None else None}
    # 2) sharded_odometry and a bare gather_features with an error (bench.py's timed loop)
    def encode_pairs(lo, hi):
        if rank == bad_rank:
            raise exc(text)
        return torch.zeros((hi - lo, 512))
    try:
        sharded_odometry(n_pairs, encode_pairs, scan)
        report["odo"] = "no error"
    except ShardError as e:
        report["odo"] = {"rank": e.rank, "type": e.remote_type}
    lo, hi = shard_range(n_pairs, rank, world)
    try:
        gather_features(None if rank == bad_rank else torch.zeros((hi - lo, 512)), n_pairs,
                        error=exc(text) if rank == bad_rank else None)
        report["gather"] = "no error"
    except ShardError as e:
        report["gather"] = {"rank": e.rank, "type": e.remote_type}
    # 2b) a rank whose features have the wrong width / type (mismatched blocks hang or corrupt the collective): the
    # contract [*, 512] fp32 is checked before the gather and the mismatch travels like any other failure
    def encode_wrong(lo, hi):
        return torch.zeros((hi - lo, 256), dtype=torch.float64) if rank == bad_rank else torch.zeros((hi - lo, 512))
    try:
        sharded_odometry(n_pairs, encode_wrong, scan)
        report["wrong_block"] = "no error"
    except ShardError as e:
        report["wrong_block"] = {"rank": e.rank, "type": e.remote_type}
    # 2c) timing mode: the device fence after a failed walk raises too (a sticky device error) — the rank must still reach
    # the gather
    enc = make_encode(True)

    def bad_sync():
        if rank == bad_rank:
            raise RuntimeError("device fence failed")
    enc.sync = bad_sync
    try:
        sharded_sequence(n_pairs + 1, enc, scan, batch, timing={})
        report["fence"] = "no error"
    except ShardError as e:
        report["fence"] = {"rank": e.rank, "type": e.remote_type}
    # 3) the failure-agreeing barrier
    try:
        rendezvous(exc(text) if rank == bad_rank else None)
        report["rendezvous"] = "no error"
    except ShardError as e:
        report["rendezvous"] = {"rank": e.rank, "type": e.remote_type, "msg": e.remote_message}
    rendezvous(None)     # and it is a plain barrier when nobody failed
    # 4) the group is still in step: a clean run gives the right answer on every rank
    timing = {}
    rot, tr = sharded_sequence(n_pairs + 1, make_encode(False), scan, batch, lanes=2, timing=timing)
    report["clean_ok"] = bool(torch.equal(rot[:, 0], torch.arange(n_pairs, dtype=torch.float32)))
    report["local_pairs"] = timing["local_pairs"]
    json.dump(report, open(os.path.join(out_dir, "fail_rank%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
