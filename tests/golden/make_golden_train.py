"""Golden fixture for the CLVO training step, produced by running the REFERENCE:
`ATDNVO(batch_size=B).train()` stepped over a T-frame clip exactly as train_odometry.py:21-49 does, `CLVO_Loss`
(odometry/loss.py), `loss.backward()`, one `AdamW` + `CosineAnnealingLR` step (train_odometry.py:99-105), and a
second iteration on the updated weights. Stored: the loss of both iterations, every gradient's L2 norm and a few
elements, the updated parameters' checksums/elements and the BatchNorm running statistics.

Run only in the build container (needs /root/reference):   python tests/golden/make_golden_train.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"

from make_golden import install_stubs  # noqa: E402

B, T = 2, 3
HP = dict(lr=1e-3, wd=1e-3, eps=1e-8, total_steps=10, eta_min=1e-9, alpha=1.0, w=3)


def sample_idx(n, k=6):
    return np.unique(np.linspace(0, n - 1, k).astype(np.int64))


def main():
    install_stubs()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from atdn_vslam.odometry.network import ATDNVO
    from atdn_vslam.odometry.loss import CLVO_Loss
    from atdn_vslam_amd import synthetic as syn

    model = ATDNVO(batch_size=B, in_channels=2)
    model.load_state_dict(syn.to_torch(syn.make_clvo_state(seed=1)))
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=HP["lr"], weight_decay=HP["wd"], eps=HP["eps"])
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, HP["total_steps"], eta_min=HP["eta_min"])
    loss_fn = CLVO_Loss(HP["alpha"], w=HP["w"], device="cpu")

    r = np.random.RandomState(41)
    out = {"B": B, "T": T, "seed_weights": 1, "seed_flow": 40, "seed_targets": 41}
    for it in range(2):
        fl = torch.from_numpy(syn.make_flow(B * T, 376, 1232, seed=40 + it)).view(B, T, 2, 376, 1232)
        true_rot = torch.from_numpy(r.uniform(-0.02, 0.02, (B, T, 3)).astype(np.float32))
        true_tr = torch.from_numpy(r.uniform(-0.5, 1.5, (B, T, 3)).astype(np.float32))
        out["true_rot%d" % it], out["true_tr%d" % it] = true_rot.numpy(), true_tr.numpy()
        opt.zero_grad()
        rots, trs = [], []
        for j in range(T):
            pr, pt = model(fl[:, j].float())
            rots.append(pr)
            trs.append(pt)
        pred_rots, pred_trs = torch.stack(rots, dim=1), torch.stack(trs, dim=1)
        loss = loss_fn(pred_rots, pred_trs, true_rot, true_tr, device="cpu")
        loss.backward()
        out["loss%d" % it] = np.float64(loss.item())
        out["pred_rot%d" % it], out["pred_tr%d" % it] = pred_rots.detach().numpy(), pred_trs.detach().numpy()
        out["lr%d" % it] = np.float64(sched.get_last_lr()[0])
        if it == 0:
            for name, p in model.named_parameters():
                if p.grad is None:
                    out["nograd/" + name] = np.int64(1)
                    continue
                g = p.grad.detach().flatten().double()
                out["gnorm/" + name] = np.float64(g.norm().item())
                idx = sample_idx(g.numel())
                out["gidx/" + name] = idx
                out["gval/" + name] = g[idx].numpy()
        opt.step()
        sched.step()
        model.reset_lstm()
        for name, p in model.named_parameters():
            v = p.detach().flatten().double()
            out["pnorm%d/" % it + name] = np.float64(v.norm().item())
            out["pval%d/" % it + name] = v[sample_idx(v.numel())].numpy()
        for name, b in model.named_buffers():
            if name.endswith("running_mean") or name.endswith("running_var"):
                out["stat%d/" % it + name] = b.detach().double().numpy()
    np.savez_compressed(os.path.join(HERE, "train.npz"), **{k: v for k, v in out.items()},
                        **{"hp_" + k: np.float64(v) for k, v in HP.items()})
    print("train golden written; losses", out["loss0"], out["loss1"])


if __name__ == "__main__":
    main()
