"""CPU ORACLE (test infrastructure, not product code) — CLVO pose head `ATDNVO`.

Functional fp32 restatement on stock torch CPU ops of
`ATDNVO.forward` (atdn_vslam/odometry/network.py:122-146), its blocks
(`Conv.forward` layers/conv.py:36-37 = BN(Mish(conv)); `ResidualConv.forward`
layers/conv.py:83-90; `Linear.forward` layers/linear.py:35-42) and the flow
normalisation (utils/normalizations.py:8-10).  The LSTM state is explicit
(h1, c1, h2, c2) instead of hidden module attributes.

Parity pin: tests/golden/clvo.npz (outputs of the imported reference).
"""
import torch
import torch.nn.functional as F

FLOW_STD = (58.1837, 17.7647)


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=1e-5)


def _conv_block(x, sd, p, stride, padding):
    return _bn(F.mish(F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=stride, padding=padding)),
               sd, p + ".bn")


def _res_block(x, sd, p):
    y = _conv_block(x, sd, p + ".conv.0", 1, 1)
    y = _conv_block(y, sd, p + ".conv.1", 2, 1)
    skip = F.conv2d(x, sd[p + ".skip_layer.weight"], sd[p + ".skip_layer.bias"], stride=2)
    return _bn(F.mish(y + skip), sd, p + ".out_block.1")


def _lin_mish(x, sd, p):
    return F.mish(F.linear(x, sd[p + ".linear.weight"], sd[p + ".linear.bias"]))


@torch.no_grad()
def clvo_encode(sd, flows):
    """flows [B,2,H,W] → 512-d feature (the stateless, shardable part; network.py:131-134)."""
    std = torch.tensor(FLOW_STD, dtype=flows.dtype).view(1, 2, 1, 1)
    x = flows / std
    x = F.conv2d(x, sd["encoder_CNN.0.weight"], sd["encoder_CNN.0.bias"], groups=2)
    x = _conv_block(x, sd, "encoder_CNN.1", 2, 3)
    for i in (2, 3, 4, 5):
        x = _res_block(x, sd, "encoder_CNN.%d" % i)
    x = _conv_block(x, sd, "encoder_CNN.6", 3, 0)
    x = x.flatten(1)
    return _lin_mish(x, sd, "encoder_CNN.8")


def zero_state(batch=1):
    return [torch.zeros(batch, 512) for _ in range(4)]


def _lstm_cell(x, h, c, sd, p):
    return torch._VF.lstm_cell(x, (h, c), sd[p + ".weight_ih"], sd[p + ".weight_hh"], sd[p + ".bias_ih"],
                               sd[p + ".bias_hh"])


@torch.no_grad()
def clvo_step(sd, feat, state):
    """One recurrent step (network.py:137-146). state = [h1,c1,h2,c2]; returns (rot, tr, new_state)."""
    h1, c1, h2, c2 = state
    h1, c1 = _lstm_cell(feat, h1, c1, sd, "lstm1")
    x2 = _lin_mish(h1, sd, "lstm_linear")
    h2, c2 = _lstm_cell(x2, h2, c2, sd, "lstm2")
    outs = []
    for head in ("rotation_regressor", "translation_regressor"):
        y = _lin_mish(h2, sd, head + ".0")
        y = _lin_mish(y, sd, head + ".1")
        outs.append(F.linear(y, sd[head + ".2.weight"]))
    return outs[0], outs[1], [h1, c1, h2, c2]


@torch.no_grad()
def clvo_forward(sd, flows, state):
    return clvo_step(sd, clvo_encode(sd, flows), state)
