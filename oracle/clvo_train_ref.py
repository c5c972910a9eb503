"""CPU ORACLE (test infrastructure, not product code) — one CLVO training iteration.

Functional restatement, on stock torch CPU ops with autograd, of what train_odometry.py:21-49 does per batch:
`ATDNVO` in train mode stepped over the T frames of a clip (odometry/network.py:122-146; BatchNorm layers use the
statistics of the current call and update their running averages with momentum 0.1, unbiased variance; the LSTM state
is carried across the T calls and the graph is kept, i.e. back-propagation through time over the clip), `CLVO_Loss`
(odometry/loss.py:25-118), `backward`, `AdamW` (train_odometry.py:99) and `CosineAnnealingLR` (101-105).
With alpha = 1 (the shipped configuration, README) the composite term is multiplied by 0; it is restated anyway —
as in the reference it is built from `transform()` outputs created with `torch.tensor(...)`, which detaches it from
the graph, so it never contributes gradient.

Parity pin: tests/golden/train.npz (the imported reference's losses, gradients, updated weights, running stats).
"""
import math

import torch
import torch.nn.functional as F

from .clvo_ref import FLOW_STD
from .pose_ref import euler2matrix  # noqa: F401  (float64 helpers for the detached composite term)

ROT_WEIGHT_UNUSED = (0.0175, 0.0031, 0.0027)   # loss.py:15-16 builds these weights but never applies them
DELTA, KHI = 1.0, 100.0                        # loss.py:20-21


def _bn_train(x, params, stats, p, momentum=0.1, eps=1e-5):
    """BatchNorm2d in training mode; `stats` (dict of running_mean / running_var) is updated in place."""
    return F.batch_norm(x, stats[p + ".running_mean"], stats[p + ".running_var"], params[p + ".weight"],
                        params[p + ".bias"], training=True, momentum=momentum, eps=eps)


def _conv_block(x, P, S, p, stride, padding):
    y = F.mish(F.conv2d(x, P[p + ".conv.weight"], P[p + ".conv.bias"], stride=stride, padding=padding))
    return _bn_train(y, P, S, p + ".bn")


def _res_block(x, P, S, p):
    y = _conv_block(x, P, S, p + ".conv.0", 1, 1)
    y = _conv_block(y, P, S, p + ".conv.1", 2, 1)
    skip = F.conv2d(x, P[p + ".skip_layer.weight"], P[p + ".skip_layer.bias"], stride=2)
    return _bn_train(F.mish(y + skip), P, S, p + ".out_block.1")


def _lin_mish(x, P, p):
    return F.mish(F.linear(x, P[p + ".linear.weight"], P[p + ".linear.bias"]))


def forward_train(P, S, flows, state):
    """One call of ATDNVO.forward in train mode. flows [B,2,H,W]; state = [h1,c1,h2,c2] (graph kept)."""
    std = torch.tensor(FLOW_STD, dtype=flows.dtype).view(1, 2, 1, 1)
    x = F.conv2d(flows / std, P["encoder_CNN.0.weight"], P["encoder_CNN.0.bias"], groups=2)
    x = _conv_block(x, P, S, "encoder_CNN.1", 2, 3)
    for i in (2, 3, 4, 5):
        x = _res_block(x, P, S, "encoder_CNN.%d" % i)
    x = _conv_block(x, P, S, "encoder_CNN.6", 3, 0)
    feat = _lin_mish(x.flatten(1), P, "encoder_CNN.8")
    h1, c1, h2, c2 = state
    h1, c1 = torch._VF.lstm_cell(feat, (h1, c1), P["lstm1.weight_ih"], P["lstm1.weight_hh"], P["lstm1.bias_ih"],
                                 P["lstm1.bias_hh"])
    x2 = _lin_mish(h1, P, "lstm_linear")
    h2, c2 = torch._VF.lstm_cell(x2, (h2, c2), P["lstm2.weight_ih"], P["lstm2.weight_hh"], P["lstm2.bias_ih"],
                                 P["lstm2.bias_hh"])
    outs = []
    for head in ("rotation_regressor", "translation_regressor"):
        y = _lin_mish(h2, P, head + ".0")
        y = _lin_mish(y, P, head + ".1")
        outs.append(F.linear(y, P[head + ".2.weight"]))
    return outs[0], outs[1], [h1, c1, h2, c2]


def transform_loss(pr, pt, tr_, tt):
    return DELTA * ((pt - tt) ** 2).sum(-1) + KHI * ((pr - tr_) ** 2).sum(-1)


def clvo_loss(pred_rot, pred_tr, true_rot, true_tr, alpha=1.0, w=3):
    """loss.py:25-58. Tensors [B,T,3]. The composite term is evaluated without graph, as the reference's is."""
    l_rel = transform_loss(pred_rot, pred_tr, true_rot, true_tr).sum(-1)   # [B]
    if alpha == 1.0:
        return l_rel.mean()
    from . import pose_ref
    com = []
    with torch.no_grad():
        for b in range(pred_rot.shape[0]):
            pm = [torch.from_numpy(pose_ref.transform(pred_rot[b, i].numpy(), pred_tr[b, i].numpy())).float()
                  for i in range(pred_rot.shape[1])]
            tm = [torch.from_numpy(pose_ref.transform(true_rot[b, i].numpy(), true_tr[b, i].numpy())).float()
                  for i in range(pred_rot.shape[1])]
            tot = 0.0
            for j in range(len(pm) - w + 1):
                a, c = pm[j], tm[j]
                for i in range(j + 1, j + w):
                    a, c = a @ pm[i], c @ tm[i]
                pr_, tr2 = torch.from_numpy(pose_ref.matrix2euler(a[:3, :3].numpy())).float(), \
                    torch.from_numpy(pose_ref.matrix2euler(c[:3, :3].numpy())).float()
                tot = tot + transform_loss(pr_, a[:3, 3], tr2, c[:3, 3])
            com.append(tot)
    return (alpha * l_rel + (1 - alpha) * torch.stack(com)).mean()


def cosine_lr(step, base_lr, total_steps, eta_min):
    """Learning rate CosineAnnealingLR(T_max=total_steps) reports after `step` scheduler steps (closed form)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * step / total_steps)) / 2


def adamw_step(p, g, m, v, t, lr, wd, eps, b1=0.9, b2=0.999):
    """torch.optim.AdamW single-tensor update, in place; t is the 1-based step count."""
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))


def split_state(sd):
    """state_dict -> (trainable parameters with requires_grad, BatchNorm running statistics)."""
    P, S = {}, {}
    for k, v in sd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            S[k] = v.clone()
        elif k.endswith("num_batches_tracked"):
            continue
        else:
            P[k] = v.clone().requires_grad_(True)
    return P, S


def train_iteration(P, S, flows, true_rot, true_tr, alpha=1.0, w=3):
    """flows [B,T,2,H,W]; returns (loss, pred_rot [B,T,3], pred_tr [B,T,3]) with gradients left in P[k].grad."""
    B, T = flows.shape[:2]
    state = [torch.zeros(B, 512) for _ in range(4)]
    rots, trs = [], []
    for j in range(T):
        r, t, state = forward_train(P, S, flows[:, j].float(), state)
        rots.append(r)
        trs.append(t)
    pred_rot, pred_tr = torch.stack(rots, dim=1), torch.stack(trs, dim=1)
    loss = clvo_loss(pred_rot, pred_tr, true_rot, true_tr, alpha, w)
    for p in P.values():
        p.grad = None
    loss.backward()
    return loss.detach(), pred_rot.detach(), pred_tr.detach()
