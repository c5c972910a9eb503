"""Map creation at the end of odometry: trains the MappingVAE auto-encoder on the stored keyframes.

Mirrors `NeuralSLAM.__create_map` (atdn_vslam/slam_framework/neural_slam.py:305-352). SURVEY.md §8f row 3 keeps
this step on stock PyTorch-ROCm, as the reference does: 50 epochs of AdamW over the keyframe images are a one-off
at the end of a mapping run, not part of the per-frame hot path. What it produces —
`<keyframes_path>/MappingVAE_weights.pth` with the reference's state-dict layout — is what the HIP embedding path
(`modules.MappingVAE` -> `atdn_vae_*`) loads for relocalisation.

* `MappingVAENet` — the full auto-encoder (encoder + decoder) of atdn_vslam/localization/network.py:9-77 as a torch
  module for training; sub-modules are created in the reference's order, so the same `torch.manual_seed` gives the
  same initial weights and the state dict has the reference's 263 keys (tests/golden/state_keys.json).
* `create_map` — the training loop: batch 16, shuffled, incomplete batch dropped; AdamW(lr 1e-3, weight decay 1e-3);
  cosine schedule to 1e-5 over all steps; loss = MSE(prediction, target) + mean |saturation(target) -
  saturation(prediction)| with target = normalise(gaussian_blur5(resize(image -> prediction size))); weights saved
  after every epoch; per-epoch losses to `mapping_loss.pth` in the working directory.

Colour augmentation. The reference applies torchvision's `ColorJitter(brightness=0.1, saturation=0.1, hue=1e-3)` to
float images in 0..255. torchvision clamps float images to [0, 1] inside every jitter step, so in the reference the
network input collapses to ~1/255 of full scale wherever a pixel is >= 1 (a reference quirk).
Here the same jitter (same factor ranges, random order of the three operations) runs on the image scaled to [0, 1]
and the result is scaled back: `jitter_bound=1.0` reproduces the reference's clamp instead. `augment=False` switches
the jitter off (used to pin the loop against the reference run with its ColorJitter stubbed out).
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

RGB_MEAN = (0.485, 0.456, 0.406)
RGB_STD = (0.229, 0.224, 0.225)


# ------------------------------------------------------------------------------------------- model
class _Conv(nn.Module):
    """conv -> activation -> BatchNorm (layers/conv.py:7-37)."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=padding, bias=True)
        self.activation = nn.Mish()
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return self.bn(self.activation(self.conv(x)))


class _ResidualConv(nn.Module):
    """Two conv blocks + strided 1x1 skip, then activation -> BatchNorm (layers/conv.py:40-90)."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv = nn.Sequential(_Conv(cin, cin, 3, 1, 1), _Conv(cin, cout, 3, stride, 1))
        self.skip_layer = nn.Conv2d(cin, cout, 1, stride=stride, bias=True)
        self.out_block = nn.Sequential(nn.Mish(), nn.BatchNorm2d(cout))

    def forward(self, x):
        return self.out_block(self.conv(x) + self.skip_layer(x))


def _resize(x, size):
    """torchvision's tensor resize: bilinear, antialiased, half-pixel centres."""
    return F.interpolate(x, size=list(size), mode="bilinear", align_corners=False, antialias=True)


class _TransposedConv(nn.Module):
    """conv block -> ConvTranspose2d -> activation -> BatchNorm, plus a 1x1 skip on the resized input, then
    activation -> BatchNorm (layers/conv.py:93-141)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(_Conv(cin, cout, 3, 1, 1),
                                  nn.ConvTranspose2d(cout, cout, 3, stride=2, padding=1, output_padding=0, bias=True),
                                  nn.Mish(), nn.BatchNorm2d(cout))
        self.skip_layer = nn.Conv2d(cin, cout, 1)
        self.out_layer = nn.Sequential(nn.Mish(), nn.BatchNorm2d(cout))

    def forward(self, x):
        direct = self.conv(x)
        return self.out_layer(direct + self.skip_layer(_resize(x, direct.shape[-2:])))


def normalize_rgb(x):
    """get_rgb_norm() (utils/normalizations.py:4-6): /255, then ImageNet mean / std."""
    mean = torch.tensor(RGB_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(RGB_STD, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    return ((x - 0.0) / 255.0 - mean) / std


class MappingVAENet(nn.Module):
    """The complete (non-variational) MappingVAE for training; `forward` returns the reference's 4-tuple."""

    CHANNELS = (16, 16, 32, 64, 128, 128)

    def __init__(self):
        super().__init__()
        c = self.CHANNELS
        enc = [_Conv(3, 3, [7, 7], 1, [3, 3])]
        cin = 3
        for cout in c:
            enc.append(_ResidualConv(cin, cout, 2))
            cin = cout
        self.encoder = nn.Sequential(*enc)
        self.mean_lin = nn.Conv2d(c[5], c[5], 1, 1)
        dec = []
        for cout in (c[4], c[3], c[2], c[1], c[0], 8):
            dec.append(_TransposedConv(cin, cout))
            cin = cout
        dec.append(nn.Conv2d(8, 3, 3, padding=1))
        self.decoder = nn.Sequential(*dec)

    def forward(self, image):
        mu = self.mean_lin(self.encoder(normalize_rgb(image)))
        return mu, None, mu, self.decoder(mu)


# ------------------------------------------------------------------------------------------- image ops
def gaussian_blur5(x):
    """torchvision.transforms.functional.gaussian_blur(x, [5, 5]): sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8 = 1.1,
    reflect padding, separable kernel applied per channel."""
    k, sigma = 5, 0.3 * ((5 - 1) * 0.5 - 1) + 0.8
    t = torch.linspace(-(k - 1) * 0.5, (k - 1) * 0.5, k, dtype=x.dtype, device=x.device)
    g = torch.exp(-0.5 * (t / sigma) ** 2)
    g = g / g.sum()
    k2 = (g[:, None] * g[None, :]).expand(x.shape[1], 1, k, k)
    return F.conv2d(F.pad(x, [2, 2, 2, 2], mode="reflect"), k2, groups=x.shape[1])


def _gray(x):
    return (0.2989 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3])


def _rgb2hsv(img):
    r, g, b = img.unbind(dim=-3)
    maxc = torch.max(img, dim=-3).values
    minc = torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    cr_div = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / cr_div, (maxc - g) / cr_div, (maxc - b) / cr_div
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img):
    h, s, v = img.unbind(dim=-3)
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int32) % 6
    p = torch.clamp(v * (1.0 - s), 0.0, 1.0)
    q = torch.clamp(v * (1.0 - s * f), 0.0, 1.0)
    t = torch.clamp(v * (1.0 - s * (1.0 - f)), 0.0, 1.0)
    mask = i.unsqueeze(dim=-3) == torch.arange(6, device=i.device).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(dtype=img.dtype), a4)


def color_jitter(img, brightness=0.1, saturation=0.1, hue=1e-3, bound=255.0, generator=None):
    """torchvision's ColorJitter for a float batch [B,3,H,W]: one set of factors per call, the enabled operations in
    random order; brightness / saturation factors uniform in [1 - a, 1 + a], hue shift uniform in [-hue, hue].
    The arithmetic runs on img * (1 / bound) clamped to [0, 1] (`bound=1.0`: torchvision's own behaviour on 0..255
    floats, which the reference inherits) and is scaled back by `bound`."""
    x = (img / bound).clamp(0.0, 1.0)
    # torchvision draws a permutation of its four operations (contrast is disabled here and skipped), then the factors
    order = torch.randperm(4, generator=generator)
    bf = float(torch.empty(1).uniform_(1.0 - brightness, 1.0 + brightness, generator=generator))
    sf = float(torch.empty(1).uniform_(1.0 - saturation, 1.0 + saturation, generator=generator))
    hf = float(torch.empty(1).uniform_(-hue, hue, generator=generator))
    for op in order.tolist():
        if op == 0:
            x = (bf * x).clamp(0.0, 1.0)
        elif op == 1:
            continue
        elif op == 2:
            x = (sf * x + (1.0 - sf) * _gray(x)).clamp(0.0, 1.0)
        else:
            hsv = _rgb2hsv(x)
            h = (hsv[:, 0] + hf) % 1.0
            x = _hsv2rgb(torch.stack((h, hsv[:, 1], hsv[:, 2]), dim=1))
    return x * bound


class KeyframeImages(torch.utils.data.Dataset):
    """`<keyframes_path>/rgb/%06d.pth` as float images (ColorDataset(pth=True), localization/datasets.py:8-62)."""

    def __init__(self, keyframes_path):
        self.dir = os.path.join(keyframes_path, "rgb")
        self.n = len([f for f in os.listdir(self.dir) if f.endswith(".pth")])

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return torch.load(os.path.join(self.dir, "%06d.pth" % i)).float()


def map_loss(im_pred, im):
    """neural_slam.py:331-339: the target is the image at the prediction's size, blurred and normalised."""
    tgt = normalize_rgb(gaussian_blur5(_resize(im, im_pred.shape[-2:])))
    loss1 = ((im_pred - tgt) ** 2).mean()
    sat_true = tgt.amax(dim=1) - tgt.amin(dim=1)
    sat_pred = im_pred.amax(dim=1) - im_pred.amin(dim=1)
    return loss1 + (sat_true - sat_pred).abs().mean()


def create_map(keyframes_path, device="cuda", num_epochs=50, batch_size=16, augment=True, jitter_bound=255.0,
               loss_file="mapping_loss.pth", progress=None, stop_after=None, miopen=True):
    """Train the auto-encoder on the keyframes under `keyframes_path`; returns (net in eval mode, per-epoch losses).
    Needs at least `batch_size` keyframes (the reference's loader drops the incomplete batch: with fewer it would
    divide by zero at neural_slam.py:346). `stop_after=n` ends the run after n epochs of the `num_epochs` schedule
    (the learning-rate curve is the full run's). `miopen=False` runs the convolutions on PyTorch's native GEMM-based
    kernels: MIOpen compiles a kernel per layer shape on first use (~2 minutes on a fresh machine), which a 50-epoch
    run amortises and a two-epoch smoke run does not."""
    if not miopen:
        with torch.backends.cudnn.flags(enabled=False):
            return create_map(keyframes_path, device, num_epochs, batch_size, augment, jitter_bound, loss_file, progress,
                              stop_after, True)
    net = MappingVAENet().to(device).train()
    data = KeyframeImages(keyframes_path)
    loader = torch.utils.data.DataLoader(dataset=data, batch_size=batch_size, shuffle=True, drop_last=True)
    if len(loader) == 0:
        raise RuntimeError("map creation needs at least %d keyframes, found %d" % (batch_size, len(data)))
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=1e-3)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, num_epochs * len(loader), eta_min=1e-5)
    losses = []
    out = os.path.join(keyframes_path, "MappingVAE_weights.pth")
    for epoch in range(num_epochs):
        running = 0.0
        for im in loader:
            opt.zero_grad()
            im = im.to(device)
            im_in = color_jitter(im, bound=jitter_bound) if augment else im
            _, _, _, im_pred = net(im_in)
            loss = map_loss(im_pred, im)
            running += loss.item()
            loss.backward()
            opt.step()
            sched.step()
        torch.save(net.state_dict(), out)
        losses.append(running / len(loader))
        if progress is not None:
            progress(epoch, losses[-1])
        if stop_after is not None and epoch + 1 >= stop_after:
            break
    net.eval()
    if loss_file:
        torch.save(torch.tensor(losses), loss_file)
    return net, losses
