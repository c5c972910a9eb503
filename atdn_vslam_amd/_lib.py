"""ctypes binding of libatdn_hip.so (include/atdn_hip.h). There is no fallback: if the
library is missing or a call fails, the caller gets an exception."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (ATDN_LIB_PATH: another build of the same library, for A/B timing of two builds inside one GPU job; diagnostics only)
LIB_PATH = os.environ.get("ATDN_LIB_PATH") or os.path.join(_HERE, "libatdn_hip.so")

_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)
_vp = C.c_void_p

# name -> (restype, argtypes); the single source for the loader and the "exports every symbol" test
SIGNATURES = {
    "atdn_version": (C.c_int, []),
    "atdn_last_error": (C.c_char_p, []),
    "atdn_gma_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int]),
    "atdn_gma_set_low_latency": (C.c_int, [_vp, C.c_int]),
    "atdn_gma_load": (C.c_int, [_vp, C.c_char_p, _vp, _i64p, C.c_int]),
    "atdn_gma_finalize": (C.c_int, [_vp]),
    "atdn_gma_forward": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "atdn_gma_forward_predictions": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "atdn_gma_forward_sequence": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "atdn_gma_forward_sequence_continued": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "atdn_gma_debug_read": (C.c_long, [_vp, C.c_char_p, _vp, C.c_long, _vp]),
    "atdn_gma_profile": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _f32p, _vp]),
    "atdn_gma_profile_mode": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _vp]),
    "atdn_gma_workspace_bytes": (C.c_size_t, [_vp]),
    "atdn_gma_destroy": (None, [_vp]),
    "atdn_clvo_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int]),
    "atdn_clvo_load": (C.c_int, [_vp, C.c_char_p, _vp, _i64p, C.c_int]),
    "atdn_clvo_finalize": (C.c_int, [_vp]),
    "atdn_clvo_encode": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "atdn_clvo_step": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "atdn_clvo_destroy": (None, [_vp]),
    "atdn_vae_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int]),
    "atdn_vae_load": (C.c_int, [_vp, C.c_char_p, _vp, _i64p, C.c_int]),
    "atdn_vae_finalize": (C.c_int, [_vp]),
    "atdn_vae_embedding_shape": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "atdn_vae_encode": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "atdn_vae_destroy": (None, [_vp]),
    "atdn_clvo_trainer_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int]),
    "atdn_clvo_trainer_load": (C.c_int, [_vp, C.c_char_p, _vp, _i64p, C.c_int]),
    "atdn_clvo_trainer_finalize": (C.c_int, [_vp]),
    "atdn_clvo_trainer_forward_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f32p, _vp]),
    "atdn_clvo_trainer_gradients": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_long)]),
    "atdn_clvo_trainer_adamw_step": (C.c_int, [_vp, C.c_float, C.c_float, C.c_float, C.c_int, _vp]),
    "atdn_clvo_trainer_read": (C.c_long, [_vp, C.c_char_p, C.c_int, _vp, C.c_long, _vp]),
    "atdn_clvo_trainer_destroy": (None, [_vp]),
    "atdn_pose_transform_f32": (C.c_int, [_vp, _vp, _vp]),
    "atdn_pose_rel2abs": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "atdn_pose_accumulate_f32": (C.c_int, [_vp, _vp, _vp]),
    "atdn_resize_frames": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "atdn_resize_frames_mode": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "atdn_resize_frames_u8": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "atdn_pad_frames": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "atdn_ingest_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "atdn_ingest_frames_u8": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "atdn_ingest_destroy": (None, [_vp]),
    "atdn_corr_lookup": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp]),
    "atdn_corr_pyramid": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "atdn_corr_lookup_bricks": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _vp]),
    "atdn_conv2d_nhwc": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "atdn_conv2d_nhwc_sf": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "atdn_conv2d_nhwc_sf_epi": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
}

GMA_STAGES = ("fnet", "corr", "pool", "cnet", "attention", "lookup", "motion_encoder", "aggregate", "gru_zr", "gru_q",
              "flow_head", "mask", "gru_ctx", "attn_logits", "agg_vt", "convc1", "gru_zr_v", "gru_q_v")

_lib = None


def lib():
    """The loaded library (loads on first use)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s is missing: the HIP extension has not been built (python -m atdn_vslam_amd.build). "
                "There is no CPU fallback for the product path." % LIB_PATH)
        # PyTorch-ROCm ships its own HIP runtime: it must be the one already mapped when this library (linked against
        # libamdhip64 by soname) is loaded, or the process ends up with two runtimes and the second one finds no device
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("libatdn_hip: " + lib().atdn_last_error().decode("utf-8", "replace"))


def load_state(load_fn, handle, state):
    """Feed a {key: tensor/ndarray} state dict to atdn_*_load (float entries only)."""
    import numpy as np
    import torch
    for key, val in state.items():
        if isinstance(val, torch.Tensor):
            if not val.dtype.is_floating_point:
                continue
            arr = val.detach().to("cpu", torch.float32).contiguous().numpy()
        else:
            arr = np.asarray(val)
            if arr.dtype.kind != "f":
                continue
            arr = np.ascontiguousarray(arr, dtype=np.float32)
        shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape) if arr.ndim else (C.c_int64 * 1)(1)
        check(load_fn(handle, key.encode(), arr.ctypes.data_as(_vp), shape, arr.ndim))
