"""ctypes binding of libksr.so (the C ABI declared in include/ksr.h).

There is exactly one compute path: the HIP library.  Importing this module without the
built library, or calling an op on a non-GPU tensor, raises; nothing falls back to
PyTorch or CPU code.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p, POINTER

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libksr.so")

_lib = None


class KsrError(RuntimeError):
    pass


# When set to a dict, every library call is bracketed by two events on torch's current stream (the
# stream the kernels are launched on) and appended under the entry point's name: bench.py uses this
# to time individual kernels live.  None (the default) adds no overhead beyond one attribute test.
PROFILE = None


class _Proxy:
    """Attribute access returns the ctypes function, wrapped with event timing while PROFILE is set."""

    def __init__(self, cdll):
        self._cdll = cdll

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if PROFILE is None or not name.endswith(("_fwd", "_bwd", "_up", "_wgrad", "_sum", "colreduce", "_token", "_attn", "_loss", "_step", "outer2", "_batched", "_apply", "_entries", "_ycbcr", "_rgb", "_grad", "_finalize", "_mix", "_merge", "_dgrad", "_batch", "_wg", "_v", "_l1", "_input", "_img", "_kan", "_flush", "_groups", "_lnf", "_bwd_dx")):
            return fn

        def timed(*args):
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            rc = fn(*args)
            e.record()
            PROFILE.setdefault(name, []).append((s, e))
            return rc

        return timed


def lib() -> "_Proxy":
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KsrError(
                f"{LIB_PATH} is missing: build it with `make -C super-resolution_amd/csrc` "
                "(or python -c 'import __graft_entry__ as g; g.build()'). There is no fallback path.")
        cdll = ctypes.CDLL(LIB_PATH)
        _declare(cdll)
        _lib = _Proxy(cdll)
    return _lib


_FP = POINTER(c_float)


def _declare(L: ctypes.CDLL) -> None:
    for name in EXPORTS:
        getattr(L, name).restype = c_int
    L.ksr_last_error.restype = c_char_p
    L.ksr_gram_slab_floats.restype = c_int64
    L.ksr_ssim_slab_floats.restype = c_int64
    L.ksr_ssim_l1_slab_floats.restype = c_int64
    L.ksr_unet_conv_partials.restype = c_int64
    L.ksr_unet_wgrad_slab_floats.restype = c_int64
    L.ksr_unet_bn_bwd_partials.restype = c_int64
    L.ksr_kan_lin_slab_floats.restype = c_int64
    L.ksr_fused_wgrad_slab_floats.restype = c_int64
    L.ksr_in_proj_bwd_wg_slab_floats.restype = c_int64
    L.ksr_tx_bwd_wg_slab_floats.restype = c_int64


# every symbol include/ksr.h declares (tests/test_host_logic.py compares the two lists)
EXPORTS = [
    "ksr_version", "ksr_last_error", "ksr_set_precision", "ksr_get_precision", "ksr_set_f32_split", "ksr_get_f32_split", "ksr_tx2_single_fwd",
    "ksr_reduce_defer", "ksr_reduce_flush", "ksr_reduce_pending", "ksr_dwconv_wgrad_groups",
    "ksr_batch_tx_layer_bwd_mlp_wg", "ksr_batch_tx_layer_bwd_ln1_wg", "ksr_feed_kan_bwd_lnf",
    "ksr_sm_conv_fwd", "ksr_sm_layernorm_fwd", "ksr_sm_resize_add_fwd", "ksr_sm_route_fwd", "ksr_sm_experts_fwd",
    "ksr_kan_cell_fwd", "ksr_kan_cell_bwd", "ksr_bilinear_up",
    "ksr_stem_fwd", "ksr_stem_gate_fwd", "ksr_pointwise_fwd", "ksr_dwrelu_pw_fwd", "ksr_gate_pw_fwd",
    "ksr_batch_tx_layer_fwd", "ksr_batch_attn_fwd", "ksr_batch_qkv_attn_fwd", "ksr_batch_tx_layer_post_fwd",
    "ksr_feed_kan_supported", "ksr_feed_kan_fwd", "ksr_feed_kan_fwd_img", "ksr_feed_kan_bwd_img", "ksr_gram_batched_kan", "ksr_feed_kan_dx", "ksr_feed_kan_dx_img", "ksr_stem_dx", "ksr_bilinear_up_bwd",
    "ksr_feed_kan_lane_len", "ksr_feed_kan_bwd", "ksr_batch_tx_layer_bwd_token", "ksr_batch_tx_layer_bwd_attn",
    "ksr_batch_attn_bwd", "ksr_tx_bwd_wg_slab_floats", "ksr_batch_tx_layer_bwd_out_wg", "ksr_wgrad_affine_input",
    "ksr_gate_pw_bwd_v", "ksr_fn1_bwd_stem_v", "ksr_dwconv_wgrad_v", "ksr_in_proj_bwd", "ksr_in_proj_bwd_wg", "ksr_in_proj_bwd_wg_slab_floats", "ksr_gate_pw_bwd", "ksr_gate_pw_bwd_wg", "ksr_pw2_bwd_wg", "ksr_pw1_bwd_wg", "ksr_fused_wgrad_slab_floats", "ksr_fn1_bwd", "ksr_pw2_bwd", "ksr_pw1_bwd",
    "ksr_wgrad_slab_floats", "ksr_wgrad", "ksr_colreduce", "ksr_dwconv_wgrad", "ksr_stem_scalar_wgrad", "ksr_outer2",
    "ksr_l1_y_loss", "ksr_l1_y_loss_static", "ksr_adam_step", "ksr_adam_step_dev", "ksr_set_dropout_seed_source", "ksr_comm_unique_id", "ksr_comm_init", "ksr_allreduce_sum", "ksr_comm_nranks",
    "ksr_comm_destroy",
    "ksr_pointwise_bwd", "ksr_chan_dw_fwd", "ksr_chan_dw_bwd", "ksr_chan_dw_bwd_dx", "ksr_gram_slab_floats", "ksr_gram_batched",
    "ksr_chan_soft_fwd", "ksr_chan_soft_bwd", "ksr_sum_entries", "ksr_chan_apply",
    "ksr_rgb_to_ycbcr", "ksr_ycbcr_to_rgb", "ksr_bilinear_up_to_rgb",
    "ksr_ssim_slab_floats", "ksr_ssim_fwd", "ksr_ssim_y_grad", "ksr_ssim_l1_slab_floats", "ksr_ssim_y_grad_l1",
    "ksr_unet_conv_partials", "ksr_unet_conv_fwd", "ksr_unet_bn_finalize", "ksr_unet_bn_relu_fwd", "ksr_unet_convt_fwd",
    "ksr_unet_att_mix",
    "ksr_unet_wgrad_slab_floats", "ksr_unet_wgrad", "ksr_unet_bn_bwd_partials", "ksr_unet_bn_relu_bwd", "ksr_unet_grad_merge",
    "ksr_unet_s2_dgrad", "ksr_unet_att_mix_bwd", "ksr_gather_batch",
    "ksr_kan_lin_slab_floats", "ksr_kan_lin_fwd", "ksr_kan_lin_bwd",
]


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().ksr_last_error().decode(errors="replace")
        raise KsrError(f"{what} failed (code {rc}): {msg}")


def ptr(t: torch.Tensor | None) -> c_void_p:
    """Device pointer of a contiguous GPU tensor (None -> NULL).  float32 everywhere, or bfloat16 for the [N, C]
    activation tensors of the bf16-storage mode; any other element type would be read as garbage."""
    if t is None:
        return c_void_p(0)
    if not t.is_cuda:
        raise KsrError("libksr ops need GPU tensors (the HIP path is the only path)")
    if t.dtype is not torch.float32 and t.dtype is not torch.bfloat16:
        raise KsrError(f"libksr ops need float32 tensors (bfloat16 for bf16-storage activations), got {t.dtype}")
    if not t.is_contiguous():
        raise KsrError("libksr ops need contiguous tensors")
    return c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream() -> c_void_p:
    """HIP stream handle of torch's current stream on the current device (called once per kernel launch: the raw
    accessor avoids building a torch.cuda.Stream object each time)."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return c_void_p(_RAW_STREAM(_RAW_DEVICE()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def host_floats(values) -> ctypes.Array:
    arr = (c_float * len(values))(*[float(v) for v in values])
    return arr
