"""Odometry-mode pipeline: frames -> flow -> pose head -> 6-DoF trajectory.

`OdometryPipeline` is the batch/sequence form used by the benchmark and the multi-GPU driver;
`VisualOdometry` mirrors `NeuralSLAM.__call__` in odometry mode frame by frame
(atdn_vslam/slam_framework/neural_slam.py:192-227): resize to 376x1232, pad, flow (12 iterations),
head, transform, float32 pose accumulation.
"""
import torch

from . import transforms
from .modules import ATDNVO, RAFTGMA

SLAM_SIZE = (376, 1232)  # neural_slam.py:198


def resize_frames(frames, size=SLAM_SIZE, antialias=True):
    """torchvision's tensor resize as NeuralSLAM applies it (neural_slam.py:198,220), on the GPU through libatdn_hip:
    bilinear, align_corners=False; `antialias=True` is what torchvision >= 0.17 does for tensors, `False` what older
    versions do (the reference pins none). uint8 frames are converted inside the resize kernel."""
    import ctypes as C
    from . import _lib
    if tuple(frames.shape[-2:]) == tuple(size):
        return frames
    if not frames.is_cuda:
        raise RuntimeError("resize_frames: the MI355X path needs tensors on a HIP device")
    u8 = frames.dtype == torch.uint8
    x = frames.contiguous() if u8 else frames.float().contiguous()
    out = torch.empty(tuple(x.shape[:-2]) + tuple(size), dtype=torch.float32, device=x.device)
    planes = int(x.numel() // (x.shape[-2] * x.shape[-1]))
    fn = _lib.lib().atdn_resize_frames_u8 if u8 else _lib.lib().atdn_resize_frames_mode
    with torch.cuda.device(x.device):
        _lib.check(fn(C.c_void_p(x.data_ptr()), planes, x.shape[-2], x.shape[-1], size[0], size[1], int(bool(antialias)),
                      C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


class FrameIngest:
    """uint8 camera frames in host memory -> fp32 frames at the network size on the device (atdn_ingest_*): async H2D
    on the handle's copy stream into alternating staging slots, resize fused with the uint8 -> fp32 conversion on the
    current stream. Replaces `im.to(device)` + `TF.resize` of neural_slam.py:197-199,219-221 for whole clips."""

    def __init__(self, in_size, max_frames, size=SLAM_SIZE, antialias=True, device="cuda:0"):
        import ctypes as C
        from . import _lib
        self.device = torch.device(device)
        self.in_size, self.size, self.max_frames = tuple(in_size), tuple(size), int(max_frames)
        self._h = C.c_void_p()
        # host buffers of the copies that may still be in flight: one per staging slot. The native call waits (on the host)
        # for the copy that last used the slot it is about to reuse, so the buffers of the last TWO calls are the only ones
        # an asynchronous copy can still read (include/atdn_hip.h, LIFETIME).
        self._keep = [None, None]
        self._calls = 0
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().atdn_ingest_create(C.byref(self._h), self.in_size[0], self.in_size[1], self.size[0],
                                                     self.size[1], self.max_frames, int(bool(antialias))))

    def __call__(self, host_frames):
        """host_frames [n,3,Hin,Win] uint8 CPU tensor (pinned for an asynchronous copy) -> [n,3,H,W] fp32 on the device."""
        import ctypes as C
        from . import _lib
        if host_frames.is_cuda or host_frames.dtype != torch.uint8 or host_frames.dim() != 4 or host_frames.shape[1] != 3 \
                or tuple(host_frames.shape[-2:]) != self.in_size:
            raise RuntimeError("FrameIngest expects uint8 host frames [n,3,%d,%d], got %s %s on %s"
                               % (self.in_size + (tuple(host_frames.shape), host_frames.dtype, host_frames.device)))
        x = host_frames.contiguous()
        n = x.shape[0]
        out = torch.empty((n, 3) + self.size, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().atdn_ingest_frames_u8(self._h, C.c_void_p(x.data_ptr()), n, C.c_void_p(out.data_ptr()),
                                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        self._keep[self._calls & 1] = x   # replaces the buffer of two calls ago, whose copy the native call just waited for
        self._calls += 1
        return out

    def __del__(self):
        try:
            from . import _lib
            if self._h:
                _lib.lib().atdn_ingest_destroy(self._h)
                self._h = None
        except Exception:
            pass


class OdometryPipeline:
    """Holds the two networks on one device and runs batches of frame pairs."""

    def __init__(self, gma_state, clvo_state, device="cuda:0", max_batch=4, iters=12, size=SLAM_SIZE, precision=None,
                 saturation_fallback=False, low_latency=False):
        self.device = torch.device(device)
        self.iters = iters
        self.size = size
        self.flow_net = RAFTGMA(max_batch=max_batch, precision=precision, saturation_fallback=saturation_fallback,
                                low_latency=low_latency)
        self.flow_net.load_state_dict(gma_state)
        self.flow_net = self.flow_net.to(self.device).eval()
        self.head = ATDNVO()
        self.head.load_state_dict(clvo_state)
        self.head = self.head.to(self.device).eval()
        self.padder = transforms.InputPadder((3,) + tuple(size))
        self._lane_cache = {}   # run_sequence: lane streams and FrameIngest objects, kept for the life of the pipeline

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
    def run_sequence(self, frames, batch=4, group=None, antialias=True, lanes=None, timing=None):
        """A whole sequence -> absolute poses [T,4,4] float64 (identity first), identical on every rank of `group`.

        frames [T,3,Hin,Win]: uint8 in HOST memory (pinned for asynchronous copies; every rank needs at least its own
        shard + 1 frame: the camera sequence as `evaluate_odometry.py` / NeuralSLAM walk it, neural_slam.py:196-221), or a
        tensor already on the device (uint8 or float); any object with `.shape`, `.dtype`, `.is_cuda` and slice indexing
        that returns such tensors works (a memory-mapped or cyclic sequence). Each rank takes the contiguous shard
        `shard_range` gives it, walks it in clips of `batch` pairs (host frames: H2D on the ingest's copy stream + convert +
        resize; device frames: resize), reuses the features of the frame two clips share, and the ranks exchange one
        all-gather of 512-d features before the replicated ordered scan (sharding.py).
        lanes: further OdometryPipeline objects on the SAME device (own handles and workspaces). The rank's shard is then
        cut into 1 + len(lanes) contiguous sub-ranges that are walked concurrently on separate HIP streams, one clip at a
        time round-robin: the tails of one stream's kernels are filled by the other's (what bench.py's two streams do).
        timing: dict receiving the host-clock seconds of encode / gather / scan (adds three device synchronisations)."""
        from .sharding import sharded_sequence
        T = frames.shape[0]
        host = not frames.is_cuda
        if host and frames.dtype != torch.uint8:
            raise RuntimeError("run_sequence: host frames must be uint8 (camera frames); move float frames to the device")
        pipes = [self] + list(lanes or [])
        for p in pipes:
            if p.device != self.device or tuple(p.size) != tuple(self.size) or p.iters != self.iters:
                raise RuntimeError("run_sequence: every lane must be an OdometryPipeline on %s with the same size / iterations"
                                   % (self.device,))
        L = len(pipes)
        # lane streams and ingest objects live as long as the pipeline (a new stream per call would also mean a new pool of
        # the caching allocator per call)
        cache = self._lane_cache
        with torch.cuda.device(self.device):
            if L == 1:
                streams = [torch.cuda.current_stream()]
            else:
                while len(cache.setdefault("streams", [])) < L:
                    cache["streams"].append(torch.cuda.Stream(device=self.device))
                streams = cache["streams"][:L]
        ingests = None
        if host:
            key = (tuple(frames.shape[-2:]), batch + 1, tuple(self.size), bool(antialias))
            pool = cache.setdefault("ingests", {}).setdefault(key, [])
            while len(pool) < L:
                pool.append(FrameIngest(key[0], key[1], size=self.size, antialias=antialias, device=self.device))
            ingests = pool[:L]
        main = torch.cuda.current_stream(self.device)

        def encode_clip(s, e, continued, lane=0):
            with torch.cuda.stream(streams[lane]):
                clip = frames[s:e + 1]
                fr = ingests[lane](clip) if host else resize_frames(clip, self.size, antialias=antialias)
                f, _ = pipes[lane].features_clip(fr, continued=continued)
                if L > 1:
                    f.record_stream(main)
            return f

        def join():
            for st in streams:
                main.wait_stream(st)

        def finish():
            # split-f16 saturation guard, once per sequence and BEFORE the all-gather: EVERY lane's module is checked (the
            # counter is per device and whoever reads first takes the count: modules.RAFTGMA.check_saturation publishes it in a
            # per-device ledger, so no lane can swallow another lane's clamp), and a SplitF16RangeError raised here travels
            # through the gather to every rank (sharding.gather_features) instead of leaving them blocked in it
            # (every lane is read before anything is raised: a lane left unread would keep a stale ledger position and raise
            # for this same event at the end of the NEXT, clean, sequence — ADVICE r4)
            join()
            first = None
            for p in pipes:
                try:
                    p.flow_net.check_saturation()
                except Exception as e:   # noqa: BLE001 - re-raised below, after the other lanes have been read
                    first = first or e
            if first is not None:
                raise first

        encode_clip.device = self.device
        encode_clip.join = join
        encode_clip.finish = finish
        encode_clip.sync = lambda: torch.cuda.synchronize(self.device)
        if L > 1:
            for st in streams:
                st.wait_stream(main)
        rot, tr = sharded_sequence(T, encode_clip, self.scan, batch, group, lanes=L, timing=timing)
        return transforms.rel2abs(rot.cpu().numpy(), tr.cpu().numpy())


class VisualOdometry:
    """Frame-at-a-time odometry with the reference's call pattern: `pose = vo(frame)`."""

    def __init__(self, gma_state, clvo_state, device="cuda:0", iters=12):
        # one pair per call: the low-latency form of the flow network (modules.RAFTGMA)
        self.pipe = OdometryPipeline(gma_state, clvo_state, device=device, max_batch=1, iters=iters, low_latency=True)
        # frame-by-frame caller: every call ends in a device synchronisation anyway (the pose goes to the host), so the
        # split-f16 range guard is read on EVERY forward — no pose computed from clamped activations is ever handed out
        self.pipe.flow_net.saturation_check_every = 1
        self.device = self.pipe.device
        self.reset()

    def reset(self):
        self._prev = None
        self.current_pose = torch.eye(4, dtype=torch.float32)
        self.pipe.head.reset_lstm()

    @torch.no_grad()
    def __call__(self, im):
        im = im.to(self.device)
        im = resize_frames(im if im.dtype == torch.uint8 else im.float())   # (uint8 frames are converted inside the resize kernel)
        if im.dtype != torch.float32:
            im = im.float()
        im = self.pipe.padder.pad(im)[0]
        if self._prev is not None:
            # (pair mode's bits; the previous frame's features are reused when the chain of calls is unbroken)
            _, flow = self.pipe.flow_net.forward_consecutive(self._prev, im, iters=self.pipe.iters)
            rot, tr = self.pipe.head(flow)
            rt = torch.cat([rot.reshape(-1), tr.reshape(-1)]).cpu()   # one trip to the host for both vectors
            self.current_pose = transforms.accumulate(self.current_pose, rt[:3], rt[3:])
        self._prev = im
        return self.current_pose
