"""GPU parity of the CLVO training iteration through the C ABI, against what the imported reference
produced (tests/golden/train.npz: two iterations of train_odometry.py's loop body at B = 2, T = 3)."""
import os

import numpy as np
import pytest
import torch

from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.training import CLVOTrainer, cosine_lr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_training_iteration_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "train.npz"))
    B, T = int(g["B"]), int(g["T"])
    tr = CLVOTrainer(syn.to_torch(syn.make_clvo_state(seed=int(g["seed_weights"]))), B, T, device=DEV, lr=float(g["hp_lr"]),
                     weight_decay=float(g["hp_wd"]), eps=float(g["hp_eps"]), total_steps=int(g["hp_total_steps"]),
                     eta_min=float(g["hp_eta_min"]))
    for it in range(2):
        fl = torch.from_numpy(syn.make_flow(B * T, 376, 1232, seed=int(g["seed_flow"]) + it)).view(B, T, 2, 376, 1232)
        loss, pr, pt = tr.forward_backward(fl.to(DEV), torch.from_numpy(g["true_rot%d" % it]), torch.from_numpy(g["true_tr%d" % it]))
        ref = float(g["loss%d" % it])
        assert abs(loss - ref) < 1e-4 * max(1.0, ref), (it, loss, ref)
        np.testing.assert_allclose(pr.cpu().numpy(), g["pred_rot%d" % it], rtol=0, atol=2e-5)
        np.testing.assert_allclose(pt.cpu().numpy(), g["pred_tr%d" % it], rtol=0, atol=2e-5)
        assert abs(tr.current_lr() - float(g["lr%d" % it])) < 1e-12
        if it == 0:
            worst = 0.0
            for k in [f[6:] for f in g.files if f.startswith("gnorm/")]:
                gr = tr.gradient(k).flatten().double()
                ref_n = float(g["gnorm/" + k])
                rel = abs(float(gr.norm()) - ref_n) / (ref_n + 1e-12)
                worst = max(worst, rel)
                assert rel < 2e-3, (k, float(gr.norm()), ref_n)
                np.testing.assert_allclose(gr[g["gidx/" + k]].numpy(), g["gval/" + k], rtol=5e-3, atol=2e-3 * ref_n + 1e-7,
                                           err_msg=k)
            for k in [f[7:] for f in g.files if f.startswith("nograd/")]:
                assert float(tr.gradient(k).abs().max()) == 0.0      # polar_norm: never used by forward()
        tr.optimizer_step()
        for k in [f[7:] for f in g.files if f.startswith("pnorm%d/" % it)]:
            v = tr.parameter(k).flatten().double()
            ref_n = float(g["pnorm%d/%s" % (it, k)])
            assert abs(float(v.norm()) - ref_n) <= 2e-5 * ref_n + 1e-6, (it, k)
        sd = tr.state_dict()
        for f in g.files:
            if f.startswith("stat%d/" % it):
                np.testing.assert_allclose(sd[f[6:]].double().numpy(), g[f], rtol=1e-4, atol=1e-5, err_msg=f)


def test_cosine_schedule_closed_form():
    sched_ref = []
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-3)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 7, eta_min=1e-9)
    for i in range(7):
        sched_ref.append(sch.get_last_lr()[0])
        opt.step()
        sch.step()
    for i, r in enumerate(sched_ref):
        assert abs(cosine_lr(i, 1e-3, 7, 1e-9) - r) < 1e-15


def test_gradient_buffer_all_reduces_through_rccl(golden_dir):
    """Data-parallel path, as far as ONE GPU can show it (two ranks cannot share the leased card, the 8-GPU run is the
    driver's): the library-owned flat gradient buffer, seen by torch through __cuda_array_interface__, goes through a
    real RCCL all-reduce (world size 1) between forward_backward and the AdamW kernel. The collective must leave the
    gradients bit for bit unchanged, be ordered after the backward kernels and before the optimiser (the update equals
    the one of an identical trainer that skipped the collective), and a SUM over a 1-rank group followed by the mean
    division must be the identity."""
    import torch.distributed as dist
    g = np.load(os.path.join(golden_dir, "train.npz"))
    B, T = int(g["B"]), int(g["T"])
    kw = dict(device=DEV, lr=float(g["hp_lr"]), weight_decay=float(g["hp_wd"]), eps=float(g["hp_eps"]),
              total_steps=int(g["hp_total_steps"]), eta_min=float(g["hp_eta_min"]))
    sd = syn.to_torch(syn.make_clvo_state(seed=int(g["seed_weights"])))
    fl = torch.from_numpy(syn.make_flow(B * T, 376, 1232, seed=int(g["seed_flow"]))).view(B, T, 2, 376, 1232).to(DEV)
    rot, trn = torch.from_numpy(g["true_rot0"]), torch.from_numpy(g["true_tr0"])
    ref = CLVOTrainer(sd, B, T, **kw)
    ref.forward_backward(fl, rot, trn)
    want_grads = ref.grads.clone()
    ref.optimizer_step()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29731", world_size=1, rank=0,
                            device_id=torch.device(DEV))
    try:
        tr = CLVOTrainer(sd, B, T, **kw)
        tr.forward_backward(fl, rot, trn)
        dist.all_reduce(tr.grads, op=dist.ReduceOp.SUM)       # what allreduce_mean_ issues when world > 1
        tr.grads.div_(dist.get_world_size())
        assert torch.equal(tr.grads, want_grads)
        tr.optimizer_step()
        for k in ("lstm1.weight_hh", "encoder_CNN.1.conv.weight", "rotation_regressor.2.weight"):
            assert torch.equal(tr.parameter(k), ref.parameter(k)), k
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_training_iteration_at_the_baseline_shape_matches_the_oracle():
    """BASELINE configs[3]: batch 24 x sequence 6 at 376x1232 (the golden fixture is B = 2, T = 3; this shape only ran in
    tools/bench_train.py before). One iteration on the HIP path against the CPU oracle (torch autograd on the same
    restatement that the golden vectors pin): loss, every prediction, the gradient norm of every parameter, and the
    weights after the AdamW step."""
    from oracle import clvo_train_ref as ref
    B, T = 24, 6
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    sd = syn.to_torch(syn.make_clvo_state(seed=3))
    flows = torch.from_numpy(syn.make_flow(B * T, 376, 1232, seed=77)).view(B, T, 2, 376, 1232)
    r = np.random.RandomState(5)
    true_rot = torch.from_numpy(r.normal(0, 0.01, (B, T, 3)).astype(np.float32))
    true_tr = torch.from_numpy(r.normal(0, 0.5, (B, T, 3)).astype(np.float32))
    hp = dict(lr=1e-3, weight_decay=1e-3, eps=1e-8, total_steps=100, eta_min=1e-9)
    tr = CLVOTrainer(sd, B, T, device=DEV, **hp)
    loss, pr, pt = tr.forward_backward(flows.to(DEV), true_rot, true_tr)
    P, S = ref.split_state(sd)
    ref_loss, ref_pr, ref_pt = ref.train_iteration(P, S, flows, true_rot, true_tr)
    assert abs(loss - float(ref_loss)) < 1e-4 * max(1.0, float(ref_loss)), (loss, float(ref_loss))
    np.testing.assert_allclose(pr.cpu().numpy(), ref_pr.numpy(), rtol=0, atol=5e-5)
    np.testing.assert_allclose(pt.cpu().numpy(), ref_pt.numpy(), rtol=0, atol=5e-5)
    worst = 0.0
    for k, p in P.items():
        if p.grad is None:
            continue
        ref_n = float(p.grad.double().norm())
        got_n = float(tr.gradient(k).flatten().double().norm())
        rel = abs(got_n - ref_n) / (ref_n + 1e-12)
        worst = max(worst, rel)
        assert rel < 3e-3, (k, got_n, ref_n)
    # one AdamW step (the oracle's single-tensor update at the trainer's learning rate)
    lr0 = tr.current_lr()
    tr.optimizer_step()
    for k in ("lstm1.weight_hh", "encoder_CNN.1.conv.weight", "encoder_CNN.3.conv.1.conv.weight", "rotation_regressor.2.weight"):
        p = P[k].detach().clone()
        ref.adamw_step(p, P[k].grad, torch.zeros_like(p), torch.zeros_like(p), 1, lr0, hp["weight_decay"], hp["eps"])
        got = tr.parameter(k).cpu()
        # the first AdamW step moves a weight by lr * g / (|g| + eps): where |g| ~ eps (1e-8) that is ill-conditioned in g, so
        # the element-wise check covers the weights whose gradient is well above eps; the rest is covered by the norm
        ok = P[k].grad.abs() > 1e-5
        assert int(ok.sum()) > 0.2 * ok.numel(), k
        assert float((got - p)[ok].abs().max()) <= 2e-5 * float(p.abs().max()) + 2e-6, k
        assert abs(float(got.double().norm()) - float(p.double().norm())) <= 2e-5 * float(p.double().norm()), k
    print("24x6 iteration: loss %.6f (oracle %.6f), worst gradient-norm deviation %.2e" % (loss, float(ref_loss), worst))


def test_fused_batchnorm_statistics_match_the_separate_reduction(monkeypatch):
    """Round 5: the BatchNorm statistics of the training forward come out of the kernel that writes the layer's input (the
    16-channel convolutions' epilogue, or bn_apply for out_block). ATDN_TRAIN_FUSED_STATS=0 is the reduction pass of their own they
    replaced: same values, another summation order — loss, predictions, running statistics and gradients agree to fp32 rounding."""
    B, T = 3, 4
    sd = syn.to_torch(syn.make_clvo_state(seed=3))
    flows = torch.from_numpy(syn.make_flow(B * T, 376, 1232, seed=78)).view(B, T, 2, 376, 1232).to(DEV)
    r = np.random.RandomState(6)
    true_rot = torch.from_numpy(r.normal(0, 0.01, (B, T, 3)).astype(np.float32))
    true_tr = torch.from_numpy(r.normal(0, 0.5, (B, T, 3)).astype(np.float32))
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ATDN_TRAIN_FUSED_STATS", mode)   # read when the trainer is constructed
        tr = CLVOTrainer(sd, B, T, device=DEV, lr=1e-3, weight_decay=1e-3, eps=1e-8, total_steps=10, eta_min=1e-9)
        loss, pr, pt = tr.forward_backward(flows, true_rot, true_tr)
        out[mode] = (loss, pr.cpu(), pt.cpu(),
                     {k: tr.gradient(k).cpu() for k in ("encoder_CNN.1.conv.weight", "encoder_CNN.2.conv.0.bn.weight",
                                                        "encoder_CNN.5.out_block.1.bias", "lstm1.weight_ih")},
                     {k: v.cpu() for k, v in tr.state_dict().items()
                      if k in ("encoder_CNN.1.bn.running_mean", "encoder_CNN.3.out_block.1.running_var", "encoder_CNN.6.bn.running_var")})
    a, b = out["1"], out["0"]
    assert abs(a[0] - b[0]) < 1e-5 * max(1.0, abs(b[0]))
    assert float((a[1] - b[1]).abs().max()) < 1e-5 and float((a[2] - b[2]).abs().max()) < 1e-5
    for k in a[3]:
        na, d = float(b[3][k].double().norm()), float((a[3][k] - b[3][k]).double().norm())
        assert d <= 1e-3 * na + 1e-9, (k, d, na)
    assert len(a[4]) == 3
    for k in a[4]:
        assert float((a[4][k] - b[4][k]).abs().max()) <= 1e-5 * float(b[4][k].abs().max()) + 1e-7, k
