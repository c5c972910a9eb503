"""CPU ORACLE (test infrastructure, not product code) — GMA optical-flow forward.

Functional fp32 restatement, on stock torch CPU ops, of the op sequence the
reference executes in `RAFTGMA.forward(..., test_mode=True)`
(whl:GMA/core/network.py:72-129).  Weights come in as a flat {key: tensor}
dict in the reference's state-dict layout.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this
package; the product path never does.

Parity pin: `tests/golden/gma_c1.npz` / `gma_c2.npz` hold outputs of the
*imported reference itself* (tests/golden/make_golden.py) on the same seeded
weights and inputs; tests/test_oracle_golden.py checks this file against them.
"""
import math

import torch
import torch.nn.functional as F


def _w(sd, name):
    return sd[name + ".weight"], sd.get(name + ".bias")


def _norm(x, sd, name, kind):
    """extractor.py:16-37: InstanceNorm2d (no affine, eps 1e-5) or eval BatchNorm2d."""
    if kind == "instance":
        return F.instance_norm(x, eps=1e-5)
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], training=False, eps=1e-5)


def residual_block(x, sd, p, kind, stride):
    """extractor.py:47-55."""
    w1, b1 = _w(sd, p + "conv1")
    w2, b2 = _w(sd, p + "conv2")
    y = F.relu(_norm(F.conv2d(x, w1, b1, stride=stride, padding=1), sd, p + "norm1", kind))
    y = F.relu(_norm(F.conv2d(y, w2, b2, padding=1), sd, p + "norm2", kind))
    if stride != 1:
        wd, bd = _w(sd, p + "downsample.0")
        x = _norm(F.conv2d(x, wd, bd, stride=stride), sd, p + "norm3", kind)
    return F.relu(x + y)


def encoder(x, sd, p, kind):
    """BasicEncoder.forward, extractor.py:165-189 (eval: no dropout)."""
    w, b = _w(sd, p + "conv1")
    x = F.relu(_norm(F.conv2d(x, w, b, stride=2, padding=3), sd, p + "norm1", kind))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        x = residual_block(x, sd, "%slayer%d.0." % (p, li), kind, stride)
        x = residual_block(x, sd, "%slayer%d.1." % (p, li), kind, 1)
    w, b = _w(sd, p + "conv2")
    return F.conv2d(x, w, b)


def corr_pyramid(fmap1, fmap2, levels=4):
    """corr.py:16-30,55-63: all-pairs dot / sqrt(C), then 2x2 average pooling."""
    b, c, h, w = fmap1.shape
    corr = torch.matmul(fmap1.view(b, c, h * w).transpose(1, 2), fmap2.view(b, c, h * w))
    corr = corr / math.sqrt(float(c))
    corr = corr.reshape(b * h * w, 1, h, w)
    pyr = [corr]
    for _ in range(levels - 1):
        corr = F.avg_pool2d(corr, 2, stride=2)
        pyr.append(corr)
    return pyr


def corr_lookup(pyr, coords, radius=4):
    """corr.py:32-53 + utils.py:59-73.  coords [B,2,H,W] (x,y) → [B,L*(2r+1)^2,H,W]."""
    b, _, h, w = coords.shape
    c = coords.permute(0, 2, 3, 1).reshape(b * h * w, 1, 1, 2)
    d = torch.linspace(-radius, radius, 2 * radius + 1)
    # reference quirk: meshgrid(dy, dx) stacked then added to (x, y) → first window
    # axis steps x, second steps y
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).view(1, 2 * radius + 1, 2 * radius + 1, 2)
    out = []
    for lvl, corr in enumerate(pyr):
        hh, ww = corr.shape[-2:]
        pos = c / 2 ** lvl + delta
        gx = 2 * pos[..., 0:1] / (ww - 1) - 1
        gy = 2 * pos[..., 1:2] / (hh - 1) - 1
        s = F.grid_sample(corr, torch.cat([gx, gy], dim=-1), align_corners=True)
        out.append(s.view(b, h, w, -1))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().float()


def attention(inp, sd):
    """gma.py:54-76 with heads=1, content-only."""
    b, c, h, w = inp.shape
    qk = F.conv2d(inp, sd["att.to_qk.weight"])
    q, k = qk.chunk(2, dim=1)
    q = q.reshape(b, c, h * w).transpose(1, 2) * (c ** -0.5)
    k = k.reshape(b, c, h * w)
    return torch.softmax(torch.matmul(q, k), dim=-1)  # [B, N, N]


def aggregate(attn, mf, sd):
    """gma.py:102-115 (project is None for dim == inner_dim)."""
    b, c, h, w = mf.shape
    v = F.conv2d(mf, sd["update_block.aggregator.to_v.weight"]).reshape(b, c, h * w)
    out = torch.matmul(attn, v.transpose(1, 2)).transpose(1, 2).reshape(b, c, h, w)
    return mf + sd["update_block.aggregator.gamma"] * out


def motion_encoder(flow, corr, sd):
    """update.py:76-84."""
    p = "update_block.encoder."
    cor = F.relu(F.conv2d(corr, *_w(sd, p + "convc1")))
    cor = F.relu(F.conv2d(cor, *_w(sd, p + "convc2"), padding=1))
    flo = F.relu(F.conv2d(flow, *_w(sd, p + "convf1"), padding=3))
    flo = F.relu(F.conv2d(flo, *_w(sd, p + "convf2"), padding=1))
    out = F.relu(F.conv2d(torch.cat([cor, flo], 1), *_w(sd, p + "conv"), padding=1))
    return torch.cat([out, flow], 1)


def sep_conv_gru(h, x, sd):
    """update.py:48-63."""
    p = "update_block.gru."
    for tag, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(F.conv2d(hx, *_w(sd, p + "convz" + tag), padding=pad))
        r = torch.sigmoid(F.conv2d(hx, *_w(sd, p + "convr" + tag), padding=pad))
        q = torch.tanh(F.conv2d(torch.cat([r * h, x], 1), *_w(sd, p + "convq" + tag), padding=pad))
        h = (1 - z) * h + z * q
    return h


def flow_head(net, sd):
    """update.py:7-15."""
    p = "update_block.flow_head."
    return F.conv2d(F.relu(F.conv2d(net, *_w(sd, p + "conv1"), padding=1)), *_w(sd, p + "conv2"), padding=1)


def up_mask(net, sd):
    """update.py:120-123,138."""
    p = "update_block.mask."
    return 0.25 * F.conv2d(F.relu(F.conv2d(net, *_w(sd, p + "0"), padding=1)), *_w(sd, p + "2"))


def convex_upsample(flow, mask):
    """network.py:59-70."""
    n, _, h, w = flow.shape
    m = torch.softmax(mask.view(n, 1, 9, 8, 8, h, w), dim=2)
    uf = F.unfold(8 * flow, [3, 3], padding=1).view(n, 2, 9, 1, 1, h, w)
    return torch.sum(m * uf, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(n, 2, 8 * h, 8 * w)


def coords_grid(b, h, w):
    """utils.py:76-79: channel 0 = x, channel 1 = y."""
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([xs, ys], 0).float()[None].repeat(b, 1, 1, 1)


def strip_prefix(sd):
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


@torch.no_grad()
def gma_forward(sd, image1, image2, iters=12, flow_init=None, taps=None, predictions=None):
    """RAFTGMA.forward(test_mode=True) → (flow_low [B,2,H/8,W/8], flow_up [B,2,H,W]).
    `taps`, if a dict, receives intermediate tensors for per-stage parity tests.
    `predictions`, if a list, receives what test_mode=False returns (network.py:106-129): every iteration's flow, upsampled
    with that iteration's mask."""
    sd = strip_prefix(sd)
    im1 = (2 * (image1 / 255.0) - 1.0).contiguous()
    im2 = (2 * (image2 / 255.0) - 1.0).contiguous()
    bsz = im1.shape[0]
    fmaps = encoder(torch.cat([im1, im2], 0), sd, "fnet.", "instance").float()
    fmap1, fmap2 = fmaps[:bsz], fmaps[bsz:]
    pyr = corr_pyramid(fmap1, fmap2)
    cnet = encoder(im1, sd, "cnet.", "batch")
    net, inp = torch.split(cnet, [128, 128], dim=1)
    net = torch.tanh(net)
    inp = torch.relu(inp)
    attn = attention(inp, sd)
    h8, w8 = im1.shape[2] // 8, im1.shape[3] // 8
    coords0 = coords_grid(bsz, h8, w8)
    coords1 = coords_grid(bsz, h8, w8)
    if flow_init is not None:
        coords1 = coords1 + flow_init
    if taps is not None:
        taps.update(fmap1=fmap1, fmap2=fmap2, pyramid=pyr, net0=net, inp=inp, attn=attn)
    for it in range(iters):
        corr = corr_lookup(pyr, coords1)
        flow = coords1 - coords0
        mf = motion_encoder(flow, corr, sd)
        mfg = aggregate(attn, mf, sd)
        net = sep_conv_gru(net, torch.cat([inp, mf, mfg], 1), sd)
        delta = flow_head(net, sd)
        coords1 = coords1 + delta
        if predictions is not None:   # network.py:118-124
            predictions.append(convex_upsample(coords1 - coords0, up_mask(net, sd)))
        if taps is not None and it == 0:
            taps.update(lookup0=corr, mf0=mf, mfg0=mfg, net1=net, delta0=delta)
    mask = up_mask(net, sd)
    flow_low = coords1 - coords0
    flow_up = convex_upsample(flow_low, mask)
    if taps is not None:
        taps.update(net_final=net, mask=mask)
    return flow_low, flow_up
