"""State-dict layouts of the two networks on the odometry hot path.

The drop-in modules must load the very same checkpoints as the reference
(`gma-kitti.pth` → RAFTGMA, 185 entries, `module.`-prefixed;
`*_atdnvo_c.pth` → ATDNVO, 127 entries), so the key/shape tables are rebuilt
here from the architecture description:

* RAFTGMA  — whl:GMA/core/network.py:26-48, extractor.py:6-55,116-163,
             update.py:7-15,36-46,66-75,112-125, gma.py:6-19,34-52,79-100
* ATDNVO   — atdn_vslam/odometry/network.py:20-119,
             atdn_vslam/layers/conv.py:7-37,40-81, layers/linear.py:5-33

`tests/golden/state_keys.json` (captured from the imported reference) pins both
tables.
"""
from collections import OrderedDict

F32 = "f32"
I64 = "i64"


def _conv(spec, name, cout, cin, kh, kw, bias=True):
    spec[name + ".weight"] = ((cout, cin, kh, kw), F32)
    if bias:
        spec[name + ".bias"] = ((cout,), F32)


def _bn(spec, name, c):
    spec[name + ".weight"] = ((c,), F32)
    spec[name + ".bias"] = ((c,), F32)
    spec[name + ".running_mean"] = ((c,), F32)
    spec[name + ".running_var"] = ((c,), F32)
    spec[name + ".num_batches_tracked"] = ((), I64)


def _encoder(spec, p, batchnorm, out_dim):
    """BasicEncoder: stem 7x7/2, three stages of two residual blocks, 1x1 head."""
    if batchnorm:
        _bn(spec, p + "norm1", 64)
    _conv(spec, p + "conv1", 64, 3, 7, 7)
    cin = 64
    for li, (dim, stride) in enumerate(((64, 1), (96, 2), (128, 2)), start=1):
        for bi in range(2):
            q = "%slayer%d.%d." % (p, li, bi)
            s = stride if bi == 0 else 1
            _conv(spec, q + "conv1", dim, cin, 3, 3)
            _conv(spec, q + "conv2", dim, dim, 3, 3)
            if batchnorm:
                _bn(spec, q + "norm1", dim)
                _bn(spec, q + "norm2", dim)
                if s != 1:
                    _bn(spec, q + "norm3", dim)
            if s != 1:
                _conv(spec, q + "downsample.0", dim, cin, 1, 1)
                if batchnorm:
                    # downsample = Sequential(conv, norm3): norm3 is registered twice
                    _bn(spec, q + "downsample.1", dim)
            cin = dim
    _conv(spec, p + "conv2", out_dim, 128, 1, 1)


def gma_state_spec():
    """Ordered {key: (shape, dtype)} of RAFTGMA.state_dict() (no `module.` prefix)."""
    s = OrderedDict()
    _encoder(s, "fnet.", False, 256)
    _encoder(s, "cnet.", True, 256)
    e = "update_block.encoder."
    _conv(s, e + "convc1", 256, 324, 1, 1)
    _conv(s, e + "convc2", 192, 256, 3, 3)
    _conv(s, e + "convf1", 128, 2, 7, 7)
    _conv(s, e + "convf2", 64, 128, 3, 3)
    _conv(s, e + "conv", 126, 256, 3, 3)
    g = "update_block.gru."
    for nm in ("convz1", "convr1", "convq1"):
        _conv(s, g + nm, 128, 512, 1, 5)
    for nm in ("convz2", "convr2", "convq2"):
        _conv(s, g + nm, 128, 512, 5, 1)
    _conv(s, "update_block.flow_head.conv1", 256, 128, 3, 3)
    _conv(s, "update_block.flow_head.conv2", 2, 256, 3, 3)
    _conv(s, "update_block.mask.0", 256, 128, 3, 3)
    _conv(s, "update_block.mask.2", 576, 256, 1, 1)
    s["update_block.aggregator.gamma"] = ((1,), F32)
    _conv(s, "update_block.aggregator.to_v", 128, 128, 1, 1, bias=False)
    _conv(s, "att.to_qk", 256, 128, 1, 1, bias=False)
    s["att.pos_emb.rel_ind"] = ((160, 160), I64)
    s["att.pos_emb.rel_height.weight"] = ((319, 128), F32)
    s["att.pos_emb.rel_width.weight"] = ((319, 128), F32)
    return s


def _lin(spec, name, cout, cin, bias=True):
    spec[name + ".weight"] = ((cout, cin), F32)
    if bias:
        spec[name + ".bias"] = ((cout,), F32)


def clvo_state_spec():
    """Ordered {key: (shape, dtype)} of ATDNVO().state_dict() (compressor variant)."""
    s = OrderedDict()
    _bn(s, "polar_norm", 2)
    s["encoder_CNN.0.weight"] = ((2, 1, 1, 1), F32)
    s["encoder_CNN.0.bias"] = ((2,), F32)
    _conv(s, "encoder_CNN.1.conv", 16, 2, 7, 7)
    _bn(s, "encoder_CNN.1.bn", 16)
    for i in (2, 3, 4, 5):
        p = "encoder_CNN.%d." % i
        _conv(s, p + "conv.0.conv", 16, 16, 3, 3)
        _bn(s, p + "conv.0.bn", 16)
        _conv(s, p + "conv.1.conv", 16, 16, 3, 3)
        _bn(s, p + "conv.1.bn", 16)
        _conv(s, p + "skip_layer", 16, 16, 1, 1)
        _bn(s, p + "out_block.1", 16)
    _conv(s, "encoder_CNN.6.conv", 16, 16, 3, 3)
    _bn(s, "encoder_CNN.6.bn", 16)
    _lin(s, "encoder_CNN.8.linear", 512, 832)
    for nm in ("lstm1",):
        s[nm + ".weight_ih"] = ((2048, 512), F32)
        s[nm + ".weight_hh"] = ((2048, 512), F32)
        s[nm + ".bias_ih"] = ((2048,), F32)
        s[nm + ".bias_hh"] = ((2048,), F32)
    _lin(s, "lstm_linear.linear", 512, 512)
    s["lstm2.weight_ih"] = ((2048, 512), F32)
    s["lstm2.weight_hh"] = ((2048, 512), F32)
    s["lstm2.bias_ih"] = ((2048,), F32)
    s["lstm2.bias_hh"] = ((2048,), F32)
    for head in ("translation_regressor", "rotation_regressor"):
        _lin(s, head + ".0.linear", 128, 512)
        _lin(s, head + ".1.linear", 64, 128)
        _lin(s, head + ".2", 3, 64, bias=False)
    return s


VAE_CHANNELS = (3, 16, 16, 32, 64, 128, 128)


def vae_state_spec():
    """Ordered {key: (shape, dtype)} of the part of MappingVAE().state_dict() relocalisation uses: the encoder and
    `mean_lin` (atdn_vslam/localization/network.py:29-45). The decoder only serves the VAE's training loss."""
    s = OrderedDict()
    _conv(s, "encoder.0.conv", 3, 3, 7, 7)
    _bn(s, "encoder.0.bn", 3)
    for i in range(1, 7):
        cin, cout = VAE_CHANNELS[i - 1], VAE_CHANNELS[i]
        p = "encoder.%d." % i
        _conv(s, p + "conv.0.conv", cin, cin, 3, 3)
        _bn(s, p + "conv.0.bn", cin)
        _conv(s, p + "conv.1.conv", cout, cin, 3, 3)
        _bn(s, p + "conv.1.bn", cout)
        _conv(s, p + "skip_layer", cout, cin, 1, 1)
        _bn(s, p + "out_block.1", cout)
    _conv(s, "mean_lin", 128, 128, 1, 1)
    return s
