"""The CLVO training-iteration oracle against the imported reference's numbers (tests/golden/train.npz)."""
import os

import numpy as np
import torch

from atdn_vslam_amd import synthetic as syn
from oracle import clvo_train_ref as tr


def test_training_iteration_oracle_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "train.npz"))
    B, T = int(g["B"]), int(g["T"])
    P, S = tr.split_state(syn.to_torch(syn.make_clvo_state(seed=int(g["seed_weights"]))))
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    V = {k: torch.zeros_like(v) for k, v in P.items()}
    lr0, wd, eps = float(g["hp_lr"]), float(g["hp_wd"]), float(g["hp_eps"])
    for it in range(2):
        fl = torch.from_numpy(syn.make_flow(B * T, 376, 1232, seed=int(g["seed_flow"]) + it)).view(B, T, 2, 376, 1232)
        loss, pr, pt = tr.train_iteration(P, S, fl, torch.from_numpy(g["true_rot%d" % it]),
                                          torch.from_numpy(g["true_tr%d" % it]), float(g["hp_alpha"]), int(g["hp_w"]))
        assert abs(float(loss) - float(g["loss%d" % it])) < 2e-5 * max(1.0, float(g["loss%d" % it])), it
        np.testing.assert_allclose(pr.numpy(), g["pred_rot%d" % it], rtol=0, atol=2e-6)
        np.testing.assert_allclose(pt.numpy(), g["pred_tr%d" % it], rtol=0, atol=2e-6)
        lr = tr.cosine_lr(it, lr0, int(g["hp_total_steps"]), float(g["hp_eta_min"]))
        assert abs(lr - float(g["lr%d" % it])) < 1e-12
        for k, p in P.items():
            if "nograd/" + k in g.files:
                assert p.grad is None, k   # polar_norm is never used by forward()
                continue
            if it == 0:
                gr = p.grad.flatten().double()
                ref_n = float(g["gnorm/" + k])
                assert abs(float(gr.norm()) - ref_n) <= 2e-4 * ref_n + 1e-7, (k, float(gr.norm()), ref_n)
                np.testing.assert_allclose(gr[g["gidx/" + k]].numpy(), g["gval/" + k], rtol=2e-3,
                                           atol=2e-5 * ref_n + 1e-8, err_msg=k)
            with torch.no_grad():
                tr.adamw_step(p, p.grad, M[k], V[k], it + 1, lr, wd, eps)
        for k, p in P.items():
            ref_n = float(g["pnorm%d/" % it + k])
            assert abs(float(p.detach().double().norm()) - ref_n) <= 1e-5 * ref_n + 1e-7, (it, k)
        for k, s in S.items():
            if "stat%d/" % it + k in g.files:
                np.testing.assert_allclose(s.double().numpy(), g["stat%d/" % it + k], rtol=2e-5, atol=1e-6, err_msg=k)
