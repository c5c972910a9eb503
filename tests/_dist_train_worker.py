"""Worker for tests/test_train_dp.py: the data-parallel step of atdn_vslam_amd.training on `gloo` CPU tensors.
Each rank computes the gradients of ITS clips with the CPU training oracle (stands in for the HIP iteration, which
needs a GPU); what is under test is the exchange: the flat gradient is averaged over ranks by `allreduce_mean_`, every
rank applies the same AdamW update at the scheduled rate, and ends up with identical weights."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from atdn_vslam_amd import synthetic as syn  # noqa: E402
from atdn_vslam_amd.training import allreduce_mean_, cosine_lr  # noqa: E402
from oracle import clvo_train_ref as tr  # noqa: E402


def main():
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.set_num_threads(2)
    if len(sys.argv) > 2 and sys.argv[2] == "synthetic":
        # world-8 rehearsal of the exchange alone (VERDICT r4 #5c): a flat gradient of the trainer's size (5.06 M fp32 = 20 MB,
        # what one RCCL all-reduce carries per iteration), seeded per rank, averaged, and the same AdamW update on every rank
        n = 5_063_880
        g = torch.Generator().manual_seed(1000 + rank)
        local = torch.randn(n, generator=g)
        flat = local.clone()
        allreduce_mean_(flat)
        w = torch.zeros(n)
        lr = cosine_lr(0, 1e-3, 10, 1e-9)
        with torch.no_grad():
            tr.adamw_step(w, flat, torch.zeros_like(w), torch.zeros_like(w), 1, lr, 1e-3, 1e-8)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), local_head=local[:4096].numpy(), mean_head=flat[:4096].numpy(),
                 mean_sum=float(flat.double().sum()), w_sum=float(w.double().sum()), w_head=w[:4096].numpy())
        dist.destroy_process_group()
        return
    B, T = 2, 2
    P, S = tr.split_state(syn.to_torch(syn.make_clvo_state(seed=1)))
    keys = [k for k in P if not k.startswith("polar_norm.")]
    r = np.random.RandomState(100 + rank)
    fl = torch.from_numpy(syn.make_flow(B * T, 376, 1232, seed=60 + rank)).view(B, T, 2, 376, 1232)
    rot = torch.from_numpy(r.uniform(-0.02, 0.02, (B, T, 3)).astype(np.float32))
    trn = torch.from_numpy(r.uniform(-0.5, 1.5, (B, T, 3)).astype(np.float32))
    loss, _, _ = tr.train_iteration(P, S, fl, rot, trn)
    flat = torch.cat([P[k].grad.flatten() for k in keys])           # the trainer's flat gradient buffer
    local = flat.clone()
    allreduce_mean_(flat)
    lr = cosine_lr(0, 1e-3, 10, 1e-9)
    off = 0
    with torch.no_grad():
        for k in keys:
            n = P[k].numel()
            g = flat[off:off + n].view_as(P[k])
            tr.adamw_step(P[k], g, torch.zeros_like(P[k]), torch.zeros_like(P[k]), 1, lr, 1e-3, 1e-8)
            off += n
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), local=local.numpy(), mean=flat.numpy(), loss=float(loss),
             w=torch.cat([P[k].detach().flatten() for k in keys]).numpy())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
