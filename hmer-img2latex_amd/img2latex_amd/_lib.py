"""ctypes binding of libimg2latex_hip.so (the C ABI declared in include/img2latex_hip.h).

There is NO CPU fallback: every op raises if the HIP library is missing or if a
tensor is not on a ROCm device.  PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_uint64, c_void_p
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libimg2latex_hip.so")

OK = 0
STOP_NONE, STOP_STICKY = 0, 1
SELECT_LOGITS, SELECT_SOFTMAX, SELECT_SAMPLE = 0, 1, 2
PREP_WEIGHTS, PREP_ROWS, PREP_ALL = 1, 2, 3
MAX_LSTM_LAYERS, MAX_BEAM = 4, 8
# kernel-selection flags (include/img2latex_hip.h I2L_FLAG_*): explicit arguments, the library reads no environment
FLAG_EXACT_FP32, FLAG_NO_GROUP, FLAG_RESNET_NO_RING, FLAG_RESNET_IM2COL_STEM = 0x1, 0x2, 0x4, 0x8
FLAG_WEIGHTS_PACKED = 0x10          # conv forward: the workspace still holds the packed filters of these weights
FLAG_AGENT_SCOPE_EXCHANGE = 0x20
FLAG_CONV_NO_SPARSE_WGRAD = 0x4000   # first conv block's weight gradient by the implicit-im2col GEMM (A/B)
FLAG_CONV_COL_READY = 0x10000        # i2l_conv_f32_bwd: the workspace still holds the forward call's column image
FLAG_DECODE_REGION_CLEARED = 0x2000000  # i2l_greedy_decode_ex: the caller zeroed the group region (status + exchange granules) itself
FLAG_RESNET_WIDE_TILES = 0x40000      # i2l_conv_bn_act_bf16_fwd: 128-column tiles regardless of balance (A/B switch)
FLAG_RESNET_NO_PATCH = 0x20000       # i2l_conv_bn_act_bf16_fwd: 3x3 convs on the implicit-GEMM ring kernel (A/B switch)
FLAG_DECODE_GROUP16 = 0x8000         # greedy decode: 16 members x 16 rows per group, per-step products on the matrix cores (split-bf16 MFMA)
FLAG_DECODE_GROUP8 = 0x1000          # greedy decode: 8 members x 8 rows per group (co-resident with a conv workgroup)
FLAG_TEST_SHORT_TIMEOUT, FLAG_TEST_DROP_MEMBER = 0x40, 0x80       # test hooks of the grouped kernels


def flag_resnet_ring_depth(n: int) -> int:
    return (int(n) & 0xF) << 8


def flag_resnet_patch_shape(n: int) -> int:
    """I2L_FLAG_RESNET_PATCH_SHAPE(n): force tile shape n (1..5) of the 3x3 patch kernel; 0 = picked for balance."""
    return (int(n) & 0xF) << 20


class DecoderWeights(ctypes.Structure):
    """struct i2l_decoder_weights."""
    _fields_ = [("embedding", c_void_p), ("w_ih", POINTER(c_void_p)), ("w_hh", POINTER(c_void_p)),
                ("b_ih", POINTER(c_void_p)), ("b_hh", POINTER(c_void_p)), ("w_out", c_void_p),
                ("b_out", c_void_p), ("vocab", c_int), ("embed", c_int), ("hidden", c_int), ("layers", c_int)]


class DecoderGrads(ctypes.Structure):
    """struct i2l_decoder_grads."""
    _fields_ = [("embedding", c_void_p), ("w_ih", POINTER(c_void_p)), ("w_hh", POINTER(c_void_p)),
                ("b_ih", POINTER(c_void_p)), ("b_hh", POINTER(c_void_p)), ("w_out", c_void_p), ("b_out", c_void_p)]


_SIGNATURES = {
    "i2l_version": (c_int, []),
    "i2l_error_string": (c_char_p, [c_int]),
    "i2l_lanes_create": (c_int, [POINTER(c_void_p), c_int, POINTER(c_void_p)]),
    "i2l_lanes_destroy": (c_int, [c_void_p]),
    "i2l_lanes_join": (c_int, [c_void_p, c_void_p]),
    "i2l_stream_wait_value32": (c_int, [c_void_p, ctypes.c_uint32, ctypes.c_float, c_void_p]),
    "i2l_conv_workspace_bytes": (c_size_t, [c_int, c_int]),
    "i2l_conv3x3_relu_pool2_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                           c_int, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "i2l_conv_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "i2l_conv3x3_relu_pool2_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_int,
                                           c_void_p, c_void_p]),
    "i2l_linear_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "i2l_linear_bias_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                        c_int, c_int, c_int, c_void_p, c_size_t, c_int, c_void_p, c_void_p]),
    "i2l_linear_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "i2l_linear_bias_act_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                        c_void_p, c_size_t, c_int, c_void_p]),
    "i2l_conv_bf16_packed_bytes": (c_size_t, [c_int] * 4),
    "i2l_conv_bn_bf16_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_size_t,
                                      c_int, c_int, c_int, c_int, c_void_p]),
    "i2l_conv_bf16_workspace_bytes": (c_size_t, [c_int] * 10),
    "i2l_conv_bn_act_bf16_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 10 +
                                 [c_void_p, c_size_t, c_int, c_void_p]),
    "i2l_bottleneck_join_bf16_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_long,
                                             c_int, c_int, c_int, c_void_p]),
    "i2l_maxpool3x3s2_bf16_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "i2l_global_avgpool_bf16_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "i2l_decoder_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "i2l_decoder_prepare": (c_int, [POINTER(DecoderWeights), c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "i2l_greedy_decode": (c_int, [POINTER(DecoderWeights), c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_float, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    "i2l_greedy_decode_ex": (c_int, [POINTER(DecoderWeights), c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_int, c_void_p, ctypes.c_uint32, c_void_p]),
    "i2l_sample_decode": (c_int, [POINTER(DecoderWeights), c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float,
                                  c_int, c_float, c_uint64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    "i2l_beam_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "i2l_beam_decode": (c_int, [POINTER(DecoderWeights), c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                c_size_t, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "i2l_attention_context_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                          c_int, c_int, c_void_p]),
    "i2l_decoder_train_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "i2l_decoder_train_fwd": (c_int, [POINTER(DecoderWeights), c_void_p, c_void_p, c_int, c_int, c_float, c_uint64,
                                      c_int, c_void_p, c_size_t, c_void_p, c_int, c_void_p]),
    "i2l_decoder_train_bwd": (c_int, [POINTER(DecoderWeights), c_void_p, c_int, c_int, c_float, c_uint64, c_int,
                                      c_void_p, c_size_t, c_void_p, POINTER(DecoderGrads), c_void_p, c_int, c_void_p,
                                      c_void_p]),
    "i2l_optimizer_workspace_bytes": (c_size_t, []),
    "i2l_grad_clip_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_float, c_float,
                                        c_float, c_float, c_float, c_float, c_int, c_void_p, c_size_t, c_void_p,
                                        c_void_p]),
    "i2l_ce_workspace_bytes": (c_size_t, [c_int]),
    "i2l_ce_label_smooth_fwd_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_size_t,
                                            c_void_p, c_void_p, c_void_p]),
}

_SIGNATURES.update({
    "i2l_sequence_metrics": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "i2l_compact_ids": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "i2l_masked_accuracy": (c_int, [c_void_p, c_void_p, ctypes.c_int64, c_int, ctypes.c_int64, c_void_p, c_void_p]),
    "i2l_lanczos_ksize": (c_int, [c_int, c_int]),
    "i2l_lanczos_coeffs": (c_int, [c_int, c_int, c_void_p, c_void_p]),
    "i2l_preprocess_images": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                      c_void_p, c_void_p]),
    "i2l_bn_train_workspace_bytes": (c_size_t, [ctypes.c_int64, c_int]),
    "i2l_bn_train_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int,
                                     c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "i2l_bn_train_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int, ctypes.c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "i2l_conv_f32_workspace_bytes": (c_size_t, [c_int] * 11),
    "i2l_conv_f32_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_size_t, c_int, c_void_p]),
    "i2l_conv_f32_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 9 +
                         [c_void_p, c_size_t, c_int, c_void_p]),
    "i2l_im2col_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "i2l_col2im_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "i2l_maxpool3x3s2_f32_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "i2l_maxpool3x3s2_f32_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "i2l_global_avgpool_f32_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "i2l_global_avgpool_bwd_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "i2l_decoder_group_status_offset": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "i2l_decoder_group_region_bytes": (c_size_t, [c_int] * 5),
    "i2l_resample_ksize": (c_int, [c_int, c_int, c_int]),
    "i2l_resample_coeffs": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p]),
    "i2l_resample_coeffs_batch": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "i2l_scores_from_statistics": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_double), POINTER(c_double)]),
    "i2l_resample_coeffs_device": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "i2l_pack_host": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int]),
    "i2l_resize_bilinear_f32": (c_int, [c_void_p, c_void_p, ctypes.c_int64, c_int, c_int, c_int, c_int, c_void_p]),
})
FILTER_LANCZOS, FILTER_BICUBIC = 1, 3          # include/img2latex_hip.h I2L_FILTER_* (= PIL.Image.Resampling values)
EXPORTED_SYMBOLS = tuple(_SIGNATURES)
_LIB: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    """Load the HIP library (once).  Fails loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"img2latex_amd: HIP extension {LIB_PATH} is missing -- build it with "
                "`make -C hmer-img2latex_amd/csrc` (or __graft_entry__.build()); there is no CPU fallback")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


def check(rc: int, what: str) -> None:
    if rc != OK:
        raise RuntimeError(f"img2latex_amd: {what} failed: {lib().i2l_error_string(rc).decode()} (code {rc})")


def ids_timed_out(ids_host: torch.Tensor) -> bool:
    """True when the grouped decode kernel gave up on one of its bounded inter-workgroup waits: it then fills the
    whole id row with -3 (decode_group.inc.h), so the first column tells (a pinned buffer is slow to scan)."""
    first = ids_host[:, 0] if ids_host.dim() == 2 else ids_host
    return bool(first.numel()) and bool((first == -3).any())


def check_ids(ids_host: torch.Tensor) -> torch.Tensor:
    """The grouped decode kernel marks every id of a row with -3 when one of its bounded inter-workgroup
    polls expired (decode_group.inc.h); surface that as an error instead of returning garbage."""
    # a timed-out workgroup fills its whole row, so the first column tells (and a pinned buffer is slow to scan)
    first = ids_host[:, 0] if ids_host.dim() == 2 else ids_host
    if first.numel() and bool((first == -3).any()):
        raise RuntimeError("img2latex_amd: grouped decode timed out waiting for a peer workgroup "
                           "(GPU oversubscribed?); pass rows_per_workgroup=1 to use the row-per-workgroup kernel")
    return ids_host


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_gpu(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    """The product path runs on the MI355X only."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"img2latex_amd: {name} is on {t.device}; the HIP path needs a ROCm device "
                           "tensor and there is no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class Lanes:
    """Caller-owned side lanes of the training backward pass (include/img2latex_hip.h: i2l_lanes): `n` torch streams of
    `device` plus the library's fork / join events on them.  ``handle`` goes into the ``lanes`` argument of the backward
    entry points; ``join()`` makes the current stream wait for everything they put on the lanes."""

    def __init__(self, device, n: int = 2):
        self.streams = [torch.cuda.Stream(device=device) for _ in range(n)]
        arr = (c_void_p * n)(*[s.cuda_stream for s in self.streams])
        out = c_void_p()
        with torch.cuda.device(device):
            check(lib().i2l_lanes_create(arr, n, ctypes.byref(out)), "lanes_create")
        self.handle = out.value

    def join(self) -> None:
        check(lib().i2l_lanes_join(self.handle, stream_ptr()), "lanes_join")

    def close(self) -> None:
        if getattr(self, "handle", None):
            lib().i2l_lanes_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pointer_array(tensors: Sequence[torch.Tensor]):
    arr = (c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


# ---- optional stage marks (bench.py records a HIP event on the current stream at each mark)
_STAGE_HOOK = None


def set_stage_hook(fn):
    """Installs ``fn(name)`` (or None) as the stage hook and returns the one it replaces."""
    global _STAGE_HOOK
    old, _STAGE_HOOK = _STAGE_HOOK, fn
    return old


def mark(name: str) -> None:
    if _STAGE_HOOK is not None:
        _STAGE_HOOK(name)
