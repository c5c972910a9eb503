"""Drop-in replacements for the two reference modules on the odometry path.

* `RAFTGMA`  — same constructor / forward / state_dict contract as
  whl:GMA/core/network.py:26-129 (used at neural_slam.py:51-53,202).
* `ATDNVO`   — same contract as atdn_vslam/odometry/network.py:11-162
  (used at evaluate_odometry.py:124,66 and neural_slam.py:57-59,203).

They own parameters with the reference's exact state-dict keys (so the same
checkpoints load) and run every forward through libatdn_hip's C ABI. PyTorch
only supplies device memory and the stream.
"""
import ctypes as C
import os

import torch
from torch import nn

from . import _lib
from .weights_spec import F32, clvo_state_spec, gma_state_spec, vae_state_spec

_BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked", "rel_ind")


class _Node(nn.Module):
    """Container whose only job is to reproduce the reference's parameter names."""


def _build_tree(root, spec):
    for key, (shape, kind) in spec.items():
        parts = key.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, _Node())
            mod = mod._modules[p]
        leaf = parts[-1]
        if kind == F32:
            t = torch.zeros(shape, dtype=torch.float32)
        else:
            t = torch.zeros(shape, dtype=torch.int64)
        if leaf in _BUFFER_LEAVES:
            mod.register_buffer(leaf, t)
        else:
            mod.register_parameter(leaf, nn.Parameter(t, requires_grad=False))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class SplitF16RangeError(RuntimeError):
    """Activations left the range of the split-f16 format (|x| > 65504, or NaN): results are no longer fp32-grade."""


def _require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s: the MI355X path needs tensors on a HIP device (got %s); there is no CPU fallback"
                           % (what, t.device))


class _NativeModule(nn.Module):
    """Shared plumbing: parameter fingerprinting and per-shape native handles."""

    def __init__(self):
        super().__init__()
        self._handles = {}
        self._fp_slots = None

    def _fingerprint(self):
        """Version counters of every parameter and buffer: a handle is rebuilt when any of them was written (load_state_dict,
        an in-place edit) or replaced. Called on EVERY forward, so it must be cheap: walking the module tree
        (`self.parameters()`, 185 + 127 tensors through `named_modules`) cost 0.4 ms per call — 1.5 ms of a 5.9 ms per-frame
        VisualOdometry call went into four of these (tools/diag/vo_cprofile.py). The (dict, key) slots of the leaves are
        collected once — the tree of these modules is fixed at construction — and read directly."""
        if self._fp_slots is None:
            slots = []
            for m in self.modules():
                slots += [(m._parameters, k) for k in m._parameters]
                slots += [(m._buffers, k) for k in m._buffers]
            self._fp_slots = slots
        return tuple((id(d[k]), d[k]._version) if d[k] is not None else None for d, k in self._fp_slots)

    def _drop_handles(self):
        for ent in self._handles.values():  # (handle, fingerprint, destroy_fn, max_batch)
            ent[2](ent[0])
        self._handles = {}

    def _device(self):
        """The device the module's parameters live on (NOT torch's current device: on a multi-GPU rank the two can differ)."""
        p = next(self.parameters(), None)
        if p is None or not p.is_cuda:
            # parameters still on the host (the native handle uploads its own copy): the device torch has current —
            # every forward runs under torch.cuda.device(input.device), so that is the input's device
            return torch.device("cuda", torch.cuda.current_device())
        return p.device

    def _require_input(self, t, what):
        """Inputs must be on a HIP device, and on the SAME device as the module's parameters once those are on a GPU: handles
        are keyed and created on the parameters' device (with parameters on cuda:0 and an input on cuda:1 the
        handle used to be built on device 1 and cached under device 0's key)."""
        _require_gpu(t, what)
        p = next(self.parameters(), None)
        if p is not None and p.is_cuda and p.device != t.device:
            raise RuntimeError("%s: input on %s but the module's parameters are on %s — move one of them (.to()) first"
                               % (what, t.device, p.device))

    def _key(self, H, W):
        """Native handles own weights and workspace on ONE device: keyed by (device index of the module's parameters, H, W),
        so a module moved with .to(other_gpu) builds a new handle there instead of launching on the old device's memory."""
        return (self._device().index, H, W)

    def _apply(self, fn, *args, **kw):
        # .to() / .cuda() / .float(): parameters move, handles built from the old ones are stale
        out = super()._apply(fn, *args, **kw)
        self._fp_slots = None   # (collected again on the next forward: cheap, and indifferent to how _apply replaced the leaves)
        self._drop_handles()
        return out

    def load_state_dict(self, state_dict, strict=True, **kw):
        # DataParallel checkpoints carry a "module." prefix (neural_slam.py:51-52)
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        out = super().load_state_dict(sd, strict=strict, **kw)
        self._drop_handles()
        return out

    def __del__(self):
        try:
            self._drop_handles()
        except Exception:
            pass


# device index -> {"events", "count", "reader"}: what the per-device saturation counter has reported so far (check_saturation)
_SAT_LEDGER = {}


class RAFTGMA(_NativeModule):
    """GMA optical flow; `args` is the reference's GMA_Parameters-like object (only
    `num_heads`, `position_only`, `position_and_content` are consulted)."""

    PRECISIONS = {"f32": 0, "split_f16": 1, "f16": 2}

    def __init__(self, args=None, max_batch=1, precision=None, saturation_check_every=512, saturation_fallback=False,
                 low_latency=False):
        """`low_latency=True`: the form for the reference's per-frame call pattern (neural_slam.py:202: ONE pair per call) — launches
        that would leave most of the chip idle at one to four pairs are cut finer (attention x V along its key axis) and independent
        chains of the captured graph run as parallel branches: a single-pair forward 6.9 -> 4.9 ms. Same flow within rounding
        (8e-5 px at KITTI size after 12 iterations), NOT bit-identical to the default path, whose clip,
        continued-clip and pair modes are bit-identical to each other; pipeline.VisualOdometry and slam.NeuralSLAM ask for it.
        `saturation_check_every`: in the split-f16 modes the module reads the library's saturation counter after the
        FIRST forward of a freshly loaded checkpoint (and whenever the weights changed), then every that many forwards
        (0: never again), and `check_saturation()` can be called at any time (OdometryPipeline.run_sequence does, at the end
        of a sequence). A non-zero count raises SplitF16RangeError — or, with `saturation_fallback=True`, switches THIS
        module to precision="f32" (a new handle in the same process, a warning, `self.fell_back = True`) and recomputes the
        forward that detected it. What that guarantees: the forward that DETECTED the clamp is never returned clamped; forwards
        between two checks (up to `saturation_check_every` - 1 of them) are only covered by the next check, which raises /
        falls back then — OdometryPipeline.run_sequence checks every lane before the all-gather of a sequence."""
        super().__init__()
        self.args = args
        self.saturation_fallback = bool(saturation_fallback)
        self.low_latency = bool(low_latency) or os.environ.get("ATDN_LOW_LATENCY") == "1"   # (the variable: experiments only)
        self._stream_tail = None        # forward_consecutive: (last frame tensor, its version, handle) of the chain in progress
        self.fell_back = False
        self.saturation_check_every = int(saturation_check_every)
        self.saturation_checks = 0      # how many times the counter has been read (tests)
        self._sat_pending = True        # a checkpoint whose first forward has not been checked yet
        self._sat_calls = 0
        self._sat_seen = 0              # ledger count (per device) this module has already answered for
        self.precision = precision or os.environ.get("ATDN_PRECISION", "split_f16")
        if self.precision not in self.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(self.PRECISIONS))
        self.hidden_dim = 128
        self.context_dim = 128
        if args is not None:
            args.corr_levels = 4   # network.py:33-34 writes these back into args
            args.corr_radius = 4
            if getattr(args, "num_heads", 1) != 1 or getattr(args, "position_only", False) or \
                    getattr(args, "position_and_content", False):
                raise NotImplementedError("only the configuration ATDN vSLAM ships (1 head, content-only attention)")
        self.max_batch = max_batch
        _build_tree(self, gma_state_spec())
        n = 160
        d = torch.arange(n).view(1, -1) - torch.arange(n).view(-1, 1)
        self.att.pos_emb.rel_ind.copy_(d + n - 1)

    def _drop_handles(self):
        super()._drop_handles()
        self._stream_tail = None
        self._sat_pending = True        # new weights (load_state_dict / .to()): the next forward is checked again

    def check_saturation(self, raise_on_clamp=True):
        """Reads (and atomically resets) the split-f16 saturation counter of this module's device: the number of values that
        had to be clamped to +-65504 (or were NaN) since the last read. Synchronises the current stream.
        The hardware counter is ONE per device, shared by every handle of the process on that device (lanes of run_sequence,
        a f16 handle beside a split-f16 one), so whichever module reads first takes everybody's count. The count is
        therefore published in a per-device ledger (`_SAT_LEDGER`): every split-f16 / f16 module of that device that has a
        live handle sees an event it has not yet reacted to at ITS next check and raises (or falls back) too — a clamp
        can be attributed too widely, never swallowed. Returns the count this module must answer for (the ledger entries it
        had not seen); an f32 module still reads and publishes, and returns 0."""
        if not self._handles:
            return 0
        dev = self._device()
        out = torch.empty(1, dtype=torch.float32)
        with torch.cuda.device(dev):
            key, ent = next((key, ent) for key, ent in self._handles.items() if key[0] == dev.index)
            n = _lib.lib().atdn_gma_debug_read(ent[0], b"sf_clamped", C.c_void_p(out.data_ptr()), 1, _stream())
        if n < 0:
            _lib.check(1)
        led = _SAT_LEDGER.setdefault(dev.index, {"events": 0, "count": 0, "reader": None})
        fresh = int(out[0])
        if fresh:
            led["events"] += 1
            led["count"] += fresh
            led["reader"] = "RAFTGMA handle %#x (%s, %dx%d)" % (ent[0].value or 0, self.precision, key[1], key[2])
        if self.precision == "f32":
            self._sat_seen = led["count"]
            return 0
        self.saturation_checks += 1
        self._sat_pending = False
        self._sat_calls = 0
        clamped = led["count"] - self._sat_seen
        self._sat_seen = led["count"]
        if clamped and raise_on_clamp:
            raise SplitF16RangeError(
                "%d activation(s) of the flow network left the range of the split-f16 format (|x| > 65504, or NaN) on %s "
                "(counter is per device; last read through %s): with this checkpoint the default arithmetic is not fp32-grade. "
                "Construct the module with precision=\"f32\" (or set ATDN_PRECISION=f32; the C ABI calls this mode "
                "ATDN_PRECISION_F32) to run every GEMM on the exact-fp32 matrix core instead." % (clamped, dev, led["reader"]))
        return clamped

    def _after_forward(self):
        """Returns True when the forward just issued must be recomputed (the module has switched itself to f32)."""
        if self.precision == "f32":
            return False
        self._sat_calls += 1
        if self._sat_pending or (self.saturation_check_every > 0 and self._sat_calls >= self.saturation_check_every):
            if not self.saturation_fallback:
                self.check_saturation()
                return False
            clamped = self.check_saturation(raise_on_clamp=False)
            if clamped:
                import warnings
                warnings.warn("RAFTGMA: %d activation(s) left the split-f16 range; this module now runs in precision=\"f32\" "
                              "(exact-fp32 MFMA) and the forward is recomputed" % clamped, RuntimeWarning)
                self.precision = "f32"
                self.fell_back = True
                self._drop_handles()
                return True
        return False

    def _handle(self, H, W, B):
        key = self._key(H, W)
        fp = self._fingerprint()
        ent = self._handles.get(key)
        if ent is not None and (ent[1] != fp or ent[3] < B):
            ent[2](ent[0])
            ent = None
            self._sat_pending = True
        if ent is None:
            L = _lib.lib()
            h = C.c_void_p()
            mb = max(B, self.max_batch)
            _lib.check(L.atdn_gma_create(C.byref(h), H, W, mb, self.PRECISIONS[self.precision]))
            if self.low_latency:
                _lib.check(L.atdn_gma_set_low_latency(h, 1))
            _lib.load_state(L.atdn_gma_load, h, self.state_dict())
            _lib.check(L.atdn_gma_finalize(h))
            ent = (h, fp, L.atdn_gma_destroy, mb)
            if not any(k[0] == key[0] for k in self._handles):
                # first handle of this module on the device: clamps recorded before it existed are not its own
                self._sat_seen = _SAT_LEDGER.get(key[0], {"count": 0})["count"]
            self._handles[key] = ent
        return ent[0]

    @torch.no_grad()
    def forward(self, image1, image2, iters=12, flow_init=None, upsample=True, test_mode=False):
        """`test_mode=True`: (flow_low, flow_up) of the last iteration (network.py:126-127). `test_mode=False`: the reference's
        `flow_predictions` (network.py:106-129) — a list of `iters` tensors [B,2,H,W], the upsampled flow after every iteration;
        values only (this module is inference-only: the list carries no autograd graph)."""
        self._require_input(image1, "RAFTGMA.forward")
        self._stream_tail = None   # pair mode overwrites the handle's feature maps: a forward_consecutive chain ends here
        if image1.shape != image2.shape or image1.dim() != 4 or image1.shape[1] != 3:
            raise RuntimeError("expected two [B,3,H,W] frames, got %s and %s" % (tuple(image1.shape), tuple(image2.shape)))
        B, _, H, W = image1.shape
        with torch.cuda.device(image1.device):
            im1 = image1.float().contiguous()
            im2 = image2.float().contiguous()
            fi = None
            if flow_init is not None:
                fi = flow_init.to(image1.device).float().contiguous()
                if tuple(fi.shape) != (B, 2, H // 8, W // 8):
                    raise RuntimeError("flow_init must be [B,2,H/8,W/8]")
            h = self._handle(H, W, B)
            if not test_mode:
                preds = torch.empty((int(iters), B, 2, H, W), dtype=torch.float32, device=image1.device)
                _lib.check(_lib.lib().atdn_gma_forward_predictions(h, _ptr(im1), _ptr(im2), B, int(iters), _ptr(fi), _ptr(preds),
                                                                   _stream()))
            else:
                flow_low = torch.empty((B, 2, H // 8, W // 8), dtype=torch.float32, device=image1.device)
                flow_up = torch.empty((B, 2, H, W), dtype=torch.float32, device=image1.device)
                _lib.check(_lib.lib().atdn_gma_forward(h, _ptr(im1), _ptr(im2), B, int(iters), _ptr(fi), _ptr(flow_low),
                                                       _ptr(flow_up), _stream()))
            retry = self._after_forward()
        if retry:
            return self.forward(image1, image2, iters=iters, flow_init=flow_init, upsample=upsample, test_mode=test_mode)
        if not test_mode:
            return list(preds.unbind(0))
        return flow_low, flow_up

    @torch.no_grad()
    def forward_sequence(self, frames, iters=12, flow_init=None, continued=False):
        """Flow of the B consecutive pairs of a clip `frames` [B+1,3,H,W] (pair b = frames[b] -> frames[b+1]), as
        NeuralSLAM walks a sequence; each frame passes through the feature network once. Returns (flow_low, flow_up)
        exactly as `forward(frames[:-1], frames[1:], test_mode=True)` does.
        `continued=True`: frames[0] is the frame that was frames[-1] of the previous call on this module (the next
        clip of the same sequence); its features are reused and only frames[1:] go through the feature network.
        Same bits as the non-continued call and as pair mode (round 3: the kernels' statistics grouping no longer depends on
        how many images share a launch; tests/test_gpu_round3.py)."""
        self._require_input(frames, "RAFTGMA.forward_sequence")
        self._stream_tail = None
        if frames.dim() != 4 or frames.shape[1] != 3 or frames.shape[0] < 2:
            raise RuntimeError("expected frames [B+1,3,H,W] with B >= 1, got %s" % (tuple(frames.shape),))
        if self.precision not in ("split_f16", "f16"):
            return self.forward(frames[:-1], frames[1:], iters=iters, flow_init=flow_init, test_mode=True)
        B, H, W = frames.shape[0] - 1, frames.shape[2], frames.shape[3]
        with torch.cuda.device(frames.device):
            fr = frames.float().contiguous()
            fi = None
            if flow_init is not None:
                fi = flow_init.to(frames.device).float().contiguous()
                if tuple(fi.shape) != (B, 2, H // 8, W // 8):
                    raise RuntimeError("flow_init must be [B,2,H/8,W/8]")
            flow_low = torch.empty((B, 2, H // 8, W // 8), dtype=torch.float32, device=frames.device)
            flow_up = torch.empty((B, 2, H, W), dtype=torch.float32, device=frames.device)
            h = self._handle(H, W, B)
            fn = _lib.lib().atdn_gma_forward_sequence_continued if continued else _lib.lib().atdn_gma_forward_sequence
            _lib.check(fn(h, _ptr(fr), B, int(iters), _ptr(fi), _ptr(flow_low), _ptr(flow_up), _stream()))
            retry = self._after_forward()
        if retry:   # (the f32 mode has no sequence form: pair mode on the same frames)
            return self.forward(frames[:-1], frames[1:], iters=iters, flow_init=flow_init, test_mode=True)
        return flow_low, flow_up

    @torch.no_grad()
    def forward_consecutive(self, prev, cur, iters=12):
        """`flow_net(prev[None], cur[None], iters, test_mode=True)` for a caller that walks a sequence one frame per call
        (NeuralSLAM.__call__, neural_slam.py:192-227: frame t is image2 of one pair and image1 of the next): when `prev` IS the
        tensor that was `cur` of the previous call of this method — same object, not modified since, same handle, nothing else
        run on the module in between — its features are still in the handle and only `cur` goes through the feature network
        (atdn_gma_forward_sequence_continued with one pair). The continued form gives the bits of pair mode
        (tests/test_gpu_round3.py), so this is pair mode minus one feature-network pass; anything that breaks the chain (another
        forward, new weights, .to(), another frame size) just makes the next call encode both frames again."""
        self._require_input(cur, "RAFTGMA.forward_consecutive")
        if prev.shape != cur.shape or cur.dim() != 3 or cur.shape[0] != 3:
            raise RuntimeError("expected two [3,H,W] frames, got %s and %s" % (tuple(prev.shape), tuple(cur.shape)))
        if self.precision not in ("split_f16", "f16"):
            return self.forward(prev[None], cur[None], iters=iters, test_mode=True)
        key = self._key(cur.shape[1], cur.shape[2])
        ent, tail = self._handles.get(key), self._stream_tail
        cont = (tail is not None and ent is not None and tail[0] is prev and tail[1] == prev._version and tail[2] == ent[0].value
                and ent[1] == self._fingerprint())
        low, up = self.forward_sequence(torch.stack([prev, cur]), iters=iters, continued=cont)
        ent = self._handles.get(key)
        if ent is not None and not self.fell_back:
            self._stream_tail = (cur, cur._version, ent[0].value)   # (holds `cur`: its storage cannot be handed to another tensor)
        return low, up

    def debug_read(self, name, shape, H, W):
        """Copy an internal activation of the (H, W) handle to a CPU tensor (parity tests)."""
        out = torch.empty(shape, dtype=torch.float32)
        with torch.cuda.device(self._device()):
            h = self._handles[self._key(H, W)][0]
            n = _lib.lib().atdn_gma_debug_read(h, name.encode(), C.c_void_p(out.data_ptr()), out.numel(), _stream())
        if n < 0:
            _lib.check(1)
        return out

    def profile(self, H, W, B, iters=12, reps=1, mode="pair"):
        """Per-stage device milliseconds of one eager forward (HIP events on the current stream). `mode`: "pair" (two
        feature-network passes per pair), "sequence" (B + 1 frames) or "continued" (B frames: what a continued clip of a
        long sequence costs — the mode bench.py times)."""
        ms = (C.c_float * len(_lib.GMA_STAGES))()
        with torch.cuda.device(self._device()):
            h = self._handle(H, W, B)
            _lib.check(_lib.lib().atdn_gma_profile_mode(h, B, int(iters), int(reps), {"pair": 0, "sequence": 1, "continued": 2}[mode],
                                                        ms, _stream()))
        return {k: float(v) / reps for k, v in zip(_lib.GMA_STAGES, ms)}

    def freeze_bn(self):  # network.py:45-48; inference-only module: nothing to freeze
        return None


class ATDNVO(_NativeModule):
    """CLVO pose head. Stateful like the reference: two LSTM cell states persist across calls."""

    def __init__(self, batch_size=1, in_channels=2, compressor=True, use_dropout=False, use_layernorm=False):
        super().__init__()
        if in_channels != 2 or not compressor or use_layernorm:
            raise NotImplementedError("only the shipped configuration ATDNVO() (2 channels, compressor, no layernorm)")
        self.batch_size = batch_size
        self.in_channels = in_channels
        self.device = "cpu"
        self.lstm_out_size = 512
        self.suffix = "_c" + ("d" if use_dropout else "")  # network.py:52-60 (dropout is Identity in eval)
        _build_tree(self, clvo_state_spec())
        self._state = torch.zeros(4, batch_size, 512)

    # the reference exposes the four state tensors as attributes
    lstm1_h = property(lambda self: self._state[0])
    lstm1_c = property(lambda self: self._state[1])
    lstm2_h = property(lambda self: self._state[2])
    lstm2_c = property(lambda self: self._state[3])

    def reset_lstm(self):
        self._state = torch.zeros(4, self.batch_size, 512, device=self.device)

    def to(self, device):
        super().to(device)
        self.device = device
        self.reset_lstm()  # network.py:156-162: moving the module also resets the state
        return self

    def _handle(self, H, W, B):
        key = self._key(H, W)
        fp = self._fingerprint()
        ent = self._handles.get(key)
        if ent is not None and (ent[1] != fp or ent[3] < B):
            ent[2](ent[0])
            ent = None
        if ent is None:
            L = _lib.lib()
            h = C.c_void_p()
            _lib.check(L.atdn_clvo_create(C.byref(h), H, W, B))
            _lib.load_state(L.atdn_clvo_load, h, self.state_dict())
            _lib.check(L.atdn_clvo_finalize(h))
            ent = (h, fp, L.atdn_clvo_destroy, B)
            self._handles[key] = ent
        return ent[0]

    @torch.no_grad()
    def encode(self, flows):
        """Stateless part: flows [B,2,H,W] -> 512-d features (shardable across frame pairs)."""
        self._require_input(flows, "ATDNVO.encode")
        if flows.dim() != 4 or flows.shape[1] != 2:
            raise RuntimeError("expected flows [B,2,H,W], got %s" % (tuple(flows.shape),))
        B, _, H, W = flows.shape
        with torch.cuda.device(flows.device):
            fl = flows.float().contiguous()
            feat = torch.empty((B, 512), dtype=torch.float32, device=flows.device)
            _lib.check(_lib.lib().atdn_clvo_encode(self._handle(H, W, B), _ptr(fl), B, _ptr(feat), _stream()))
        return feat

    @torch.no_grad()
    def scan(self, feats, state=None, hw=(376, 1232)):
        """Ordered recurrence over feats [T,Bs,512]; returns (rot [T,Bs,3], tr [T,Bs,3], state [4,Bs,512])."""
        self._require_input(feats, "ATDNVO.scan")
        T, Bs, _ = feats.shape
        with torch.cuda.device(feats.device):
            f = feats.float().contiguous()
            st = torch.zeros(4, Bs, 512, device=feats.device) if state is None else state.float().contiguous().clone()
            rot = torch.empty((T, Bs, 3), dtype=torch.float32, device=feats.device)
            tr = torch.empty((T, Bs, 3), dtype=torch.float32, device=feats.device)
            h = self._handle(hw[0], hw[1], Bs)
            _lib.check(_lib.lib().atdn_clvo_step(h, _ptr(f), T, Bs, _ptr(st), _ptr(rot), _ptr(tr), _stream()))
        return rot, tr, st

    @torch.no_grad()
    def forward(self, flows):
        if flows.shape[0] != self.batch_size:
            raise RuntimeError("ATDNVO was built for batch_size=%d, got %d flows" % (self.batch_size, flows.shape[0]))
        feat = self.encode(flows)
        if self._state.device != flows.device:
            self._state = self._state.to(flows.device)
        rot, tr, self._state = self.scan(feat[None], self._state, hw=tuple(flows.shape[2:]))
        return rot[0], tr[0]


class MappingVAE(_NativeModule):
    """Embedding half of the reference's MappingVAE (atdn_vslam/localization/network.py): `forward(image)` returns
    the reference's 4-tuple `(mu, logvar, latent, decoded)` with `logvar = decoded = None` — the non-variational
    model has no logvar, and the decoder only feeds the training loss (NeuralSLAM.__create_map), which is not on
    this path. `mu` [B,128,H/64,W/64] is what relocalisation compares (neural_slam.py:355-383)."""

    def __init__(self, variational=False):
        super().__init__()
        if variational:
            raise NotImplementedError("only the shipped configuration MappingVAE() (non-variational)")
        self.var = False
        _build_tree(self, vae_state_spec())

    def load_state_dict(self, state_dict, strict=True, **kw):
        # checkpoints written by the reference also hold the decoder: not needed for the embedding
        sd = {k: v for k, v in state_dict.items() if not k.startswith("decoder.") and not k.startswith("module.decoder.")}
        return super().load_state_dict(sd, strict=strict, **kw)

    def _handle(self, H, W, B):
        key = self._key(H, W)
        fp = self._fingerprint()
        ent = self._handles.get(key)
        if ent is not None and (ent[1] != fp or ent[3] < B):
            ent[2](ent[0])
            ent = None
        if ent is None:
            L = _lib.lib()
            h = C.c_void_p()
            _lib.check(L.atdn_vae_create(C.byref(h), H, W, B))
            _lib.load_state(L.atdn_vae_load, h, self.state_dict())
            _lib.check(L.atdn_vae_finalize(h))
            ent = (h, fp, L.atdn_vae_destroy, B)
            self._handles[key] = ent
        return ent[0]

    @torch.no_grad()
    def forward(self, image):
        self._require_input(image, "MappingVAE.forward")
        if image.dim() == 3:
            image = image.unsqueeze(0)
        if image.dim() != 4 or image.shape[1] != 3:
            raise RuntimeError("expected image [B,3,H,W], got %s" % (tuple(image.shape),))
        B, _, H, W = image.shape
        with torch.cuda.device(image.device):
            im = image.float().contiguous()
            h = self._handle(H, W, B)
            oh, ow = C.c_int(), C.c_int()
            _lib.check(_lib.lib().atdn_vae_embedding_shape(h, C.byref(oh), C.byref(ow)))
            mu = torch.empty((B, oh.value * ow.value, 128), dtype=torch.float32, device=image.device)
            _lib.check(_lib.lib().atdn_vae_encode(h, _ptr(im), B, _ptr(mu), _stream()))
        mu = mu.view(B, oh.value, ow.value, 128).permute(0, 3, 1, 2).contiguous()
        return mu, None, mu, None
