"""CPU ORACLE (test infrastructure, not product code) — MappingVAE encoder used by relocalisation.

Functional fp32 restatement on stock torch CPU ops of the part of `MappingVAE.forward`
(atdn_vslam/localization/network.py:57-77, non-variational) that produces the embedding `mu`:
`get_rgb_norm` (utils/normalizations.py:4-6: x/255 then ImageNet mean/std), `encoder` = Conv 7x7 (3->3) + six
stride-2 `ResidualConv` (3->16->16->32->64->128->128; network.py:29-41, blocks as in oracle/clvo_ref.py), and
`mean_lin` (1x1 conv 128->128, network.py:45). The decoder only feeds the training loss and is not restated.
`nearest_keyframe` restates NeuralSLAM.__get_closest_keyframe (slam_framework/neural_slam.py:372-383).

Parity pin: tests/golden/vae.npz (outputs of the imported reference, tests/golden/make_golden_slam.py).
"""
import torch
import torch.nn.functional as F

from .clvo_ref import _conv_block

RGB_MEAN = (0.485, 0.456, 0.406)
RGB_STD = (0.229, 0.224, 0.225)


def normalize_rgb(image):
    x = image / 255.0
    return (x - torch.tensor(RGB_MEAN).view(1, 3, 1, 1)) / torch.tensor(RGB_STD).view(1, 3, 1, 1)


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=False, eps=1e-5)


def _res_block(x, sd, p):
    y = _conv_block(x, sd, p + ".conv.0", 1, 1)
    y = _conv_block(y, sd, p + ".conv.1", 2, 1)
    skip = F.conv2d(x, sd[p + ".skip_layer.weight"], sd[p + ".skip_layer.bias"], stride=2)
    return _bn(F.mish(y + skip), sd, p + ".out_block.1")


@torch.no_grad()
def vae_encode(sd, image, taps=None):
    """image [B,3,H,W] float (0..255) -> mu [B,128,H/64,W/64]. `taps`: optional dict filled with every block's output."""
    x = normalize_rgb(image.float())
    x = _conv_block(x, sd, "encoder.0", 1, 3)
    if taps is not None:
        taps["enc0"] = x
    for i in range(1, 7):
        x = _res_block(x, sd, "encoder.%d" % i)
        if taps is not None:
            taps["enc%d" % i] = x
    return F.conv2d(x, sd["mean_lin.weight"], sd["mean_lin.bias"])


def nearest_keyframe(embeddings, code):
    """embeddings: list of mu tensors; code: mu of the query. Returns (index of the closest, distances [K])."""
    d = torch.stack([torch.norm(e - code, p=2) for e in embeddings])
    return int(torch.argmin(d)), d
