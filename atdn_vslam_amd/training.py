"""CLVO training on the MI355X path (BASELINE config 4).

`CLVOTrainer.step(flows, true_rot, true_tr)` is one iteration of the reference's loop body (train_odometry.py:21-49):
train-mode `ATDNVO` over the T frames of every clip, `CLVO_Loss` (alpha = 1), backward, `AdamW` with the
`CosineAnnealingLR` schedule (train_odometry.py:99-105), LSTM reset — all in libatdn_hip (`atdn_clvo_trainer_*`).

Data-parallel training: one process per GPU, each with its own batch shard; the flat gradient buffer is all-reduced
(RCCL through torch.distributed, one collective of 5.06 M floats) and averaged before the optimiser step, so every
rank applies the same update. BatchNorm statistics stay per rank (what torch DDP does without SyncBatchNorm).
"""
import ctypes as C
import math

import torch

from . import _lib
from .weights_spec import clvo_state_spec


class _DeviceView:
    """Zero-copy torch view of a device range owned by the library (through __cuda_array_interface__)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def cosine_lr(step, base_lr, total_steps, eta_min):
    """Learning rate `CosineAnnealingLR(T_max=total_steps, eta_min)` holds after `step` scheduler steps."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * step / total_steps)) / 2


def allreduce_mean_(flat, group=None):
    """In-place average of a flat gradient tensor over the ranks of `group` (no-op without an initialised group)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
    return flat


class CLVOTrainer:
    def __init__(self, state_dict, batch_size, sequence_length, hw=(376, 1232), device="cuda:0", lr=1e-3, weight_decay=1e-3,
                 eps=1e-8, total_steps=1000, eta_min=1e-9, group=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("CLVOTrainer: the MI355X path needs a HIP device; there is no CPU fallback")
        self.B, self.T, self.hw = batch_size, sequence_length, tuple(hw)
        self.lr, self.wd, self.eps, self.total_steps, self.eta_min = lr, weight_decay, eps, total_steps, eta_min
        self.group = group
        self.iteration = 0
        self._spec = clvo_state_spec()
        L = _lib.lib()
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.atdn_clvo_trainer_create(C.byref(self._h), self.hw[0], self.hw[1], batch_size, sequence_length))
            sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
            missing = [k for k in self._spec if k not in sd]
            if missing:
                raise KeyError("state dict lacks %s" % missing[:3])
            _lib.load_state(L.atdn_clvo_trainer_load, self._h, {k: sd[k] for k in self._spec})
            _lib.check(L.atdn_clvo_trainer_finalize(self._h))
            ptr, cnt = C.c_void_p(), C.c_long()
            _lib.check(L.atdn_clvo_trainer_gradients(self._h, C.byref(ptr), C.byref(cnt)))
            self.grads = torch.as_tensor(_DeviceView(ptr.value, cnt.value), device=self.device)

    def __del__(self):
        try:
            if self._h:
                _lib.lib().atdn_clvo_trainer_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def current_lr(self):
        return cosine_lr(self.iteration, self.lr, self.total_steps, self.eta_min)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @torch.no_grad()
    def forward_backward(self, flows, true_rot, true_tr):
        """flows [B,T,2,H,W], targets [B,T,3] on the device -> (loss, pred_rot [B,T,3], pred_tr [B,T,3])."""
        B, T = self.B, self.T
        if tuple(flows.shape) != (B, T, 2) + self.hw or not flows.is_cuda:
            raise RuntimeError("expected device flows of shape %s" % ((B, T, 2) + self.hw,))
        fl = flows.float().contiguous()
        tr_, tt = true_rot.to(self.device).float().contiguous(), true_tr.to(self.device).float().contiguous()
        pr = torch.empty((B, T, 3), device=self.device)
        pt = torch.empty((B, T, 3), device=self.device)
        loss = C.c_float()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().atdn_clvo_trainer_forward_backward(
                self._h, C.c_void_p(fl.data_ptr()), C.c_void_p(tr_.data_ptr()), C.c_void_p(tt.data_ptr()),
                C.c_void_p(pr.data_ptr()), C.c_void_p(pt.data_ptr()), C.byref(loss), self._stream()))
        return float(loss.value), pr, pt

    @torch.no_grad()
    def optimizer_step(self):
        """All-reduce (mean) of the gradients over the group, AdamW at the scheduled rate, scheduler step."""
        allreduce_mean_(self.grads, self.group)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().atdn_clvo_trainer_adamw_step(self._h, self.current_lr(), self.wd, self.eps,
                                                              self.iteration + 1, self._stream()))
        self.iteration += 1

    def step(self, flows, true_rot, true_tr):
        loss, _, _ = self.forward_backward(flows, true_rot, true_tr)
        self.optimizer_step()
        return loss

    def _read(self, key, kind):
        shape = self._spec[key][0]
        n = 1
        for d in shape:
            n *= d
        out = torch.empty(max(n, 1), dtype=torch.float32)
        with torch.cuda.device(self.device):
            got = _lib.lib().atdn_clvo_trainer_read(self._h, key.encode(), kind, C.c_void_p(out.data_ptr()), out.numel(),
                                                    self._stream())
        if got < 0:
            raise RuntimeError(_lib.lib().atdn_last_error().decode())
        return out[:n].view(shape)

    def parameter(self, key):
        return self._read(key, 0)

    def gradient(self, key):
        return self._read(key, 1)

    def state_dict(self):
        """Current weights in the reference's checkpoint layout (what train_odometry.py:142 saves)."""
        out = {}
        for k, (shape, kind) in self._spec.items():
            if k.endswith("num_batches_tracked"):
                out[k] = torch.tensor(self.iteration * self.T, dtype=torch.int64)
            elif k.endswith("running_mean") or k.endswith("running_var"):
                out[k] = self._read(k, 2)
            else:
                out[k] = self._read(k, 0)
        return out
