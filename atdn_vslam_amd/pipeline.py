"""Odometry-mode pipeline: frames -> flow -> pose head -> 6-DoF trajectory.

`OdometryPipeline` is the batch/sequence form used by the benchmark and the multi-GPU driver;
`VisualOdometry` mirrors `NeuralSLAM.__call__` in odometry mode frame by frame
(atdn_vslam/slam_framework/neural_slam.py:192-227): resize to 376x1232, pad, flow (12 iterations),
head, transform, float32 pose accumulation.
"""
import torch
import torch.nn.functional as F

from . import transforms
from .modules import ATDNVO, RAFTGMA
from .sharding import sharded_odometry

SLAM_SIZE = (376, 1232)  # neural_slam.py:198


def resize_frames(frames, size=SLAM_SIZE):
    """torchvision's tensor resize (bilinear, antialias) as NeuralSLAM applies it (neural_slam.py:198,220), on the
    GPU through libatdn_hip (atdn_resize_frames)."""
    import ctypes as C
    from . import _lib
    if tuple(frames.shape[-2:]) == tuple(size):
        return frames
    if not frames.is_cuda:
        raise RuntimeError("resize_frames: the MI355X path needs tensors on a HIP device")
    x = frames.float().contiguous()
    out = torch.empty(tuple(x.shape[:-2]) + tuple(size), dtype=torch.float32, device=x.device)
    planes = int(x.numel() // (x.shape[-2] * x.shape[-1]))
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().atdn_resize_frames(C.c_void_p(x.data_ptr()), planes, x.shape[-2], x.shape[-1], size[0],
                                                 size[1], C.c_void_p(out.data_ptr()),
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


class OdometryPipeline:
    """Holds the two networks on one device and runs batches of frame pairs."""

    def __init__(self, gma_state, clvo_state, device="cuda:0", max_batch=4, iters=12, size=SLAM_SIZE, precision=None):
        self.device = torch.device(device)
        self.iters = iters
        self.size = size
        self.flow_net = RAFTGMA(max_batch=max_batch, precision=precision)
        self.flow_net.load_state_dict(gma_state)
        self.flow_net = self.flow_net.to(self.device).eval()
        self.head = ATDNVO()
        self.head.load_state_dict(clvo_state)
        self.head = self.head.to(self.device).eval()
        self.padder = transforms.InputPadder((3,) + tuple(size))

    @torch.no_grad()
    def features(self, im1, im2):
        """im1, im2 [B,3,H,W] (already at `size`) -> (feat [B,512], flow_up [B,2,H,W])."""
        im1, im2 = self.padder.pad(im1, im2)
        _, flow = self.flow_net(im1, im2, iters=self.iters, test_mode=True)
        return self.head.encode(flow), flow

    @torch.no_grad()
    def features_clip(self, frames, continued=False):
        """frames [B+1,3,H,W]: B consecutive pairs of one clip -> (feat [B,512], flow_up [B,2,H,W]); the shared
        frames go through the feature network once (RAFTGMA.forward_sequence). `continued`: frames[0] was the last
        frame of the previous call (next clip of the same sequence) and its features are reused."""
        frames = self.padder.pad(frames)[0]
        _, flow = self.flow_net.forward_sequence(frames, iters=self.iters, continued=continued)
        return self.head.encode(flow), flow

    @torch.no_grad()
    def scan(self, feats):
        """feats [P,512] in sequence order -> (rot [P,3], tr [P,3]) from a zero LSTM state."""
        rot, tr, _ = self.head.scan(feats[:, None, :], hw=self.size)
        return rot[:, 0], tr[:, 0]

    @torch.no_grad()
    def run_sequence(self, frames, batch=4, group=None):
        """frames [T,3,H,W] on the device (every rank holds the clip or at least its shard + 1 frame).
        Returns absolute poses [T,4,4] float64 (identity first), identical on every rank."""
        n_pairs = frames.shape[0] - 1

        def encode(lo, hi):
            out = []
            for s in range(lo, hi, batch):
                e = min(s + batch, hi)
                f, _ = self.features_clip(frames[s:e + 1], continued=(s > lo))   # consecutive clips of this shard
                out.append(f)
            return torch.cat(out) if out else torch.zeros((0, 512), device=self.device)

        rot, tr = sharded_odometry(n_pairs, encode, self.scan, group)
        return transforms.rel2abs(rot.cpu().numpy(), tr.cpu().numpy())


class VisualOdometry:
    """Frame-at-a-time odometry with the reference's call pattern: `pose = vo(frame)`."""

    def __init__(self, gma_state, clvo_state, device="cuda:0", iters=12):
        self.pipe = OdometryPipeline(gma_state, clvo_state, device=device, max_batch=1, iters=iters)
        self.device = self.pipe.device
        self.reset()

    def reset(self):
        self._prev = None
        self.current_pose = torch.eye(4, dtype=torch.float32)
        self.pipe.head.reset_lstm()

    @torch.no_grad()
    def __call__(self, im):
        im = resize_frames(im.to(self.device).float())
        im = self.pipe.padder.pad(im)[0]
        if self._prev is not None:
            _, flow = self.pipe.flow_net(self._prev[None], im[None], iters=self.pipe.iters, test_mode=True)
            rot, tr = self.pipe.head(flow)
            self.current_pose = transforms.accumulate(self.current_pose, rot.squeeze().cpu(), tr.squeeze().cpu())
        self._prev = im
        return self.current_pose
