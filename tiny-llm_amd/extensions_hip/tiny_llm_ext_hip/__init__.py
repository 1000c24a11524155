"""ctypes binding of ``libtinyllm_hip.so`` with the reference extension's Python surface.

Mirrors the nanobind module ``tiny_llm_ext_ref._ext``
(reference: src/extensions_ref/bindings.cpp:11-65): same function names, keyword
names and defaults, but the arrays are PyTorch-ROCm tensors and the kernels are
hand-written gfx950 HIP behind the C ABI declared in ``include/tinyllm_hip.h``.

Like the reference, the extension is GPU-only: every op raises ``RuntimeError``
for host tensors (reference ``eval_cpu`` throws, quantized_matmul.cpp:103-109),
and importing this module fails loudly when the shared library is missing.
There is no CPU fallback.
"""

from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch

_LIB_NAME = "libtinyllm_hip.so"
_LIB_PATH = Path(__file__).resolve().parent / _LIB_NAME

if not _LIB_PATH.exists():
    raise ImportError(
        f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C tiny-llm_amd/csrc` (hipcc, gfx950). There is no CPU fallback."
    )

_lib = ctypes.CDLL(str(_LIB_PATH), mode=os.RTLD_LOCAL if hasattr(os, "RTLD_LOCAL") else 0)

_c_void_p = ctypes.c_void_p
_c_int = ctypes.c_int
_c_float = ctypes.c_float
_c_size_t = ctypes.c_size_t

TL_F32, TL_F16, TL_BF16 = 0, 1, 2
_DTYPES = {torch.float32: TL_F32, torch.float16: TL_F16, torch.bfloat16: TL_BF16}

# name -> (restype, argtypes); every symbol declared in include/tinyllm_hip.h and
# include/tinyllm_engine.h must appear here (tests/test_abi.py checks the headers).
_SIGNATURES = {
    "tl_last_error": (ctypes.c_char_p, []),
    "tl_abi_version": (_c_int, []),
    "tl_load_library": (_c_int, [ctypes.c_char_p]),
    "tl_quantized_matmul": (
        _c_int,
        [_c_void_p] * 5 + [_c_int] * 5 + [_c_int, _c_int, _c_int, _c_void_p, _c_size_t, _c_void_p],
    ),
    "tl_quantized_matmul_workspace_bytes": (_c_size_t, [_c_int] * 6),
    "tl_gather_quantized_matvec": (_c_int, [_c_void_p] * 6 + [_c_int] * 7 + [_c_void_p]),
    "tl_quantized_matmul_split_k": (_c_int, [_c_int] * 5),
    "tl_quantized_embedding": (_c_int, [_c_void_p, _c_int] + [_c_void_p] * 4 + [_c_int] * 6 + [_c_void_p]),
    "tl_rms_norm": (_c_int, [_c_void_p] * 3 + [_c_int, _c_int, _c_float, _c_int, _c_void_p]),
    "tl_rope": (_c_int, [_c_void_p] * 3 + [_c_int] * 5 + [_c_float, _c_int, _c_int, _c_void_p]),
    "tl_swiglu": (_c_int, [_c_void_p] * 3 + [_c_size_t, _c_int, _c_void_p]),
    "tl_decode_attention": (_c_int, [_c_void_p] * 5 + [_c_int] * 6 + [_c_float, _c_int, _c_int, _c_int, _c_void_p]),
    "tl_paged_cache_update": (_c_int, [_c_void_p] * 2 + [_c_int] * 8 + [_c_void_p]),
    "tl_paged_attention": (
        _c_int,
        [_c_void_p] * 6 + [_c_int] * 8 + [_c_float, _c_int, _c_int, _c_int, _c_void_p, _c_size_t, _c_void_p],
    ),
    "tl_paged_attention_workspace_bytes": (_c_size_t, [_c_int] * 8),
    "tl_paged_attention_waves": (_c_int, [_c_int]),
    # FP8 (E4M3) KV pages (include/tinyllm_hip.h, last operator section; no reference counterpart)
    "tl_kv_fp8_quantize_rows": (_c_int, [_c_void_p, _c_void_p, _c_void_p, ctypes.c_long, _c_int, _c_void_p]),
    "tl_kv_fp8_dequantize_rows": (_c_int, [_c_void_p, _c_void_p, _c_void_p, ctypes.c_long, _c_int, _c_void_p]),
    "tl_paged_cache_update_fp8": (_c_int, [_c_void_p] * 3 + [_c_int] * 7 + [_c_void_p]),
    "tl_paged_attention_fp8": (
        _c_int,
        [_c_void_p] * 8 + [_c_int] * 8 + [_c_float, _c_int, _c_int, _c_void_p, _c_size_t, _c_void_p],
    ),
}



# ---- decode engine ABI (include/tinyllm_engine.h) ------------------------------------------------
class TlW4(ctypes.Structure):
    _fields_ = [("weight_dev", _c_void_p), ("scales_dev", _c_void_p), ("biases_dev", _c_void_p),
                ("rows", _c_int), ("cols", _c_int)]


class TlLayerWeights(ctypes.Structure):
    _fields_ = [("wqkv", TlW4), ("wo", TlW4), ("wgu", TlW4), ("wdown", TlW4),
                ("input_norm_dev", _c_void_p), ("post_norm_dev", _c_void_p),
                ("q_norm_dev", _c_void_p), ("k_norm_dev", _c_void_p)]


class TlMoeWeights(ctypes.Structure):
    _fields_ = [("router", TlW4), ("gate_dev", _c_void_p), ("up_dev", _c_void_p), ("down_dev", _c_void_p),
                ("gate_scales_dev", _c_void_p), ("gate_biases_dev", _c_void_p), ("up_scales_dev", _c_void_p),
                ("up_biases_dev", _c_void_p), ("down_scales_dev", _c_void_p), ("down_biases_dev", _c_void_p),
                ("num_experts", _c_int), ("experts_per_token", _c_int), ("intermediate_size", _c_int),
                ("norm_topk_prob", _c_int)]


class TlEngineConfig(ctypes.Structure):
    _fields_ = [("hidden_size", _c_int), ("num_layers", _c_int), ("num_heads", _c_int), ("num_kv_heads", _c_int),
                ("head_dim", _c_int), ("intermediate_size", _c_int), ("vocab_size", _c_int),
                ("rope_theta", _c_float), ("rms_norm_eps", _c_float),
                ("page_size", _c_int), ("num_pages", _c_int), ("max_batch", _c_int), ("max_pages_per_seq", _c_int),
                ("max_prefill_rows", _c_int)]


class TlEngineStats(ctypes.Structure):
    _fields_ = [("pages_in_use", _c_int), ("pages_free", _c_int), ("peak_pages_in_use", _c_int),
                ("page_allocations", ctypes.c_long), ("reused_page_allocations", ctypes.c_long),
                ("decode_steps", ctypes.c_long), ("graph_captures", ctypes.c_long), ("graph_replays", ctypes.c_long),
                ("prefill_tokens", ctypes.c_long), ("kv_bytes", _c_size_t), ("workspace_bytes", _c_size_t),
                ("graph_cache_flushes", ctypes.c_long), ("aql_steps", ctypes.c_long)]


class TlStepProfile(ctypes.Structure):
    _fields_ = [("kernel_us", ctypes.c_double * 8), ("launches", _c_int * 8), ("gemv_bytes", ctypes.c_double * 5),
                ("span_us", ctypes.c_double), ("clock_khz", _c_int), ("n_splits", _c_int)]


class TlStepCheck(ctypes.Structure):
    _fields_ = [("launches", _c_int), ("written_once_plan", _c_int), ("n_splits", _c_int), ("double_writes", ctypes.c_long),
                ("elements_written", ctypes.c_long), ("first_launch", _c_int), ("first_kind", _c_int), ("first_region", _c_int),
                ("first_offset", ctypes.c_long)]


class TlLinearInfo(ctypes.Structure):
    _fields_ = [("kernel", _c_int), ("launches", _c_int), ("rows_per_pass", _c_int), ("p", _c_int * 5)]


class TlLinearEx(ctypes.Structure):
    _fields_ = [("merge_ws_dev", _c_void_p), ("n_splits", _c_int), ("ss_in_dev", _c_void_p), ("ss_in_n", _c_int),
                ("ss_out_dev", _c_void_p), ("norm_out_dev", _c_void_p), ("out_w_dev", _c_void_p), ("fragment_order", _c_int)]


class TlAttentionInfo(ctypes.Structure):
    _fields_ = [("n_splits", _c_int), ("tokens_per_split", _c_int), ("heads_per_workgroup", _c_int),
                ("launches", _c_int)]


_P = ctypes.POINTER
_SIGNATURES.update({
    "tl_tiled_w4_create": (_c_int, [_P(TlW4), _c_void_p, _P(_c_void_p)]),
    "tl_tiled_w4_destroy": (None, [_c_void_p]),
    "tl_decode_linear_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "tl_decode_linear": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_float,
                                  _c_int, _c_void_p, _c_size_t, _c_void_p, _P(TlLinearInfo)]),
    "tl_decode_linear_ex": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_float,
                                     _c_int, _c_void_p, _c_size_t, _c_void_p, _P(TlLinearEx), _P(TlLinearInfo)]),
    "tl_prefill_weights_bf16": (_c_int, [_P(TlW4), _c_void_p, _c_void_p]),
    "tl_prefill_matmul_bf16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "tl_decode_gemv_variant_compiled": (_c_int, [_c_int, _c_int, _c_int, _c_int]),
    "tl_decode_attention_fused_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "tl_decode_attention_fused": (_c_int, [_c_void_p] * 8 + [_c_int] * 6 + [_c_float, _c_float, _c_int, _c_void_p, _c_size_t,
                                                             _c_void_p, _P(TlAttentionInfo)]),
    "tl_engine_profile_step": (_c_int, [_c_void_p, _c_int, _P(TlStepProfile)]),
    "tl_engine_check_step": (_c_int, [_c_void_p, _c_int, _P(TlStepCheck)]),
    "tl_engine_create": (_c_int, [_P(TlEngineConfig), _P(TlLayerWeights), _P(TlW4), _c_void_p, _P(TlW4), _c_void_p,
                                  _P(_c_void_p)]),
    "tl_engine_create_kv": (_c_int, [_P(TlEngineConfig), _P(TlLayerWeights), _P(TlW4), _c_void_p, _P(TlW4), _c_void_p, _c_int,
                                     _P(_c_void_p)]),
    "tl_engine_kv_format": (_c_int, [_c_void_p]),
    "tl_decode_attention_fused_fp8": (_c_int, [_c_void_p] * 10 + [_c_int] * 6 + [_c_float, _c_float, _c_int, _c_void_p, _c_size_t,
                                                                  _c_void_p, _P(TlAttentionInfo)]),
    "tl_engine_set_moe_layer": (_c_int, [_c_void_p, _c_int, _P(TlMoeWeights)]),
    "tl_decode_gemv_plan": (_c_int, [_c_int, _c_int, _c_int, _P(_c_int)]),
    "tl_decode_batched_plan": (_c_int, [_c_int, _c_int, _c_int, _P(_c_int)]),
    "tl_decode_batched_variant_compiled": (_c_int, [_c_int, _c_int]),
    "tl_decode_streaming_plan": (_c_int, [_c_int, _c_int, _c_int, _P(_c_int)]),
    "tl_decode_streaming_variant_compiled": (_c_int, [_c_int, _c_int]),
    "tl_decode_attention_plan": (_c_int, [_c_int, _c_int, _c_int, _c_int, _P(_c_int)]),
    "tl_engine_set_option": (_c_int, [_c_void_p, ctypes.c_char_p, _c_int]),
    "tl_engine_destroy": (None, [_c_void_p]),
    "tl_engine_synchronize": (_c_int, [_c_void_p]),
    "tl_engine_begin": (_c_int, [_c_void_p, _c_int]),
    "tl_engine_reserve": (_c_int, [_c_void_p, _c_int, _c_int]),
    "tl_engine_release": (_c_int, [_c_void_p, _c_int]),
    "tl_engine_rewind": (_c_int, [_c_void_p, _c_int, _c_int]),
    "tl_engine_context_len": (_c_int, [_c_void_p, _c_int]),
    "tl_engine_move": (_c_int, [_c_void_p, _c_int, _c_int]),
    "tl_engine_fork": (_c_int, [_c_void_p, _c_int, _c_int]),
    "tl_engine_read_pending": (_c_int, [_c_void_p, _c_int, _P(ctypes.c_int32)]),
    "tl_engine_prefill": (_c_int, [_c_void_p, _c_int, _P(ctypes.c_int32), _c_int, _c_int]),
    "tl_engine_prefill_packed": (_c_int, [_c_void_p, _c_int, _P(ctypes.c_int), _P(ctypes.c_int32), _P(ctypes.c_int), _P(ctypes.c_int)]),
    "tl_engine_verify": (_c_int, [_c_void_p, _c_int, _P(ctypes.c_int32), _c_int, _P(ctypes.c_int32)]),
    "tl_engine_set_token": (_c_int, [_c_void_p, _c_int, ctypes.c_int32]),
    "tl_engine_decode": (_c_int, [_c_void_p, _c_int, _c_int, _c_int]),
    "tl_engine_read_tokens": (_c_int, [_c_void_p, _c_int, _c_int, _P(ctypes.c_int32)]),
    "tl_engine_logits_dev": (_c_void_p, [_c_void_p]),
    "tl_engine_copy_logits": (_c_int, [_c_void_p, _c_void_p, _c_int]),
    "tl_engine_tokens_dev": (_c_void_p, [_c_void_p]),
    "tl_engine_get_stats": (_c_int, [_c_void_p, _P(TlEngineStats)]),
    "tl_engine_replay_route": (ctypes.c_char_p, [_c_void_p]),
    "tl_engine_step_bytes": (_c_size_t, [_c_void_p, _c_int]),
})

for _name, (_res, _args) in _SIGNATURES.items():
    _fn = getattr(_lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


def _bind_optional(name: str, restype, argtypes) -> bool:
    fn = getattr(_lib, name, None)
    if fn is None:
        return False
    fn.restype = restype
    fn.argtypes = argtypes
    return True


def lib() -> ctypes.CDLL:
    """The loaded shared library (used by the decode engine binding)."""
    return _lib


def check(status: int) -> None:
    """Raise RuntimeError(tl_last_error()) for a non-zero tl_status."""
    _check(status)


def library_path() -> str:
    return str(_LIB_PATH)


def _check(status: int) -> None:
    if status != 0:
        message = _lib.tl_last_error()
        raise RuntimeError(message.decode() if message else f"tinyllm_hip error {status}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_gpu(op: str, *tensors: torch.Tensor) -> None:
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(f"{op}: the course extension is GPU-only")


def _ptr(t: torch.Tensor) -> int:
    return t.data_ptr()


_workspaces: dict[tuple[int, int], torch.Tensor] = {}


def _workspace(nbytes: int, device: torch.device) -> torch.Tensor | None:
    """Per-(device, stream) scratch buffer; grows geometrically, reused across calls."""
    if nbytes <= 0:
        return None
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def load_library(path: str) -> None:
    """reference: load_library(path) registers the metallib (utils.cpp:9-14). Here: verify a gfx950 device."""
    _check(_lib.tl_load_library(str(path).encode()))


def quantized_matmul(
    scales: torch.Tensor,
    biases: torch.Tensor,
    group_size: int,
    bits: int,
    a: torch.Tensor,
    b: torch.Tensor,
    transpose_b: bool = False,
    use_simdgroup: bool = True,
    use_split_k: bool = False,
    stream=None,
) -> torch.Tensor:
    """W4A16 ``a[M,N] @ dequant(b[K,N/8]).T`` (reference quantized_matmul.cpp:14-80 for the checks)."""
    if scales.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("quantized_matmul: scales must be float16 or bfloat16")
    if scales.dtype != biases.dtype:
        raise RuntimeError("quantized_matmul: scales and biases must be the same dtype")
    if b.dtype not in (torch.uint32, torch.int32):
        raise RuntimeError("quantized_matmul: b must be uint32")
    if a.dtype != scales.dtype:
        raise RuntimeError("quantized_matmul: a must be the same dtype as scales")
    if a.dim() != 2:
        raise RuntimeError("quantized_matmul: a must be a 2D array")
    if b.dim() != 2:
        raise RuntimeError("quantized_matmul: b must be a 2D array")
    if bits != 4:
        raise RuntimeError("quantized_matmul: bits must be 4")
    if group_size != 128:
        raise RuntimeError("quantized_matmul: group_size must be 128")
    if not transpose_b:
        raise RuntimeError("quantized_matmul: b must be transposed")
    if scales.shape != biases.shape:
        raise RuntimeError("quantized_matmul: scales and biases must have the same shape")
    if b.shape[0] != scales.shape[0]:
        raise RuntimeError("quantized_matmul: b must have the same number of rows as scales")
    M, N = a.shape
    K = b.shape[0]
    if N % group_size != 0:
        raise RuntimeError("quantized_matmul: a columns must be divisible by group_size")
    if scales.dim() != 2 or scales.shape[1] != N // group_size:
        raise RuntimeError("quantized_matmul: scales must have one column per input group")
    if b.shape[1] != N // 8:
        raise RuntimeError("quantized_matmul: a must have the same number of columns as b")
    _require_gpu("quantized_matmul", scales, biases, a, b)
    if not a.is_contiguous():
        raise RuntimeError("quantized_matmul: a must be contiguous")
    if not b.is_contiguous():
        raise RuntimeError("quantized_matmul: b must be contiguous")
    scales = scales.contiguous()
    biases = biases.contiguous()
    dt = _DTYPES[a.dtype]
    out = torch.empty((M, K), dtype=a.dtype, device=a.device)
    ws_bytes = _lib.tl_quantized_matmul_workspace_bytes(M, N, K, dt, int(use_simdgroup), int(use_split_k))
    ws = _workspace(ws_bytes, a.device)
    _check(
        _lib.tl_quantized_matmul(
            _ptr(scales), _ptr(biases), _ptr(a), _ptr(b), _ptr(out), M, N, K, group_size, bits, dt,
            int(use_simdgroup), int(use_split_k), _ptr(ws) if ws is not None else None,
            ws.numel() if ws is not None else 0, _stream(),
        )
    )
    return out


def gather_quantized_matvec(scales: torch.Tensor, biases: torch.Tensor, group_size: int, bits: int, a: torch.Tensor,
                            b: torch.Tensor, expert_ids: torch.Tensor, stream=None) -> torch.Tensor:
    """Grouped-expert W4A16 product ``out[m] = a[m] @ dequant(b[expert_ids[m]]).T`` (what the reference gets from
    ``mx.gather_qmm`` in grouped_expert_linear, src/tiny_llm_ref/moe.py:7-36).  b [E,K,N/8], scales/biases [E,K,N/128],
    a [M,N], expert_ids [M] int32 on the device."""
    if scales.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("gather_quantized_matvec: scales must be float16 or bfloat16")
    if scales.dtype != biases.dtype or a.dtype != scales.dtype:
        raise RuntimeError("gather_quantized_matvec: scales, biases and a must share one dtype")
    if b.dtype not in (torch.uint32, torch.int32):
        raise RuntimeError("gather_quantized_matvec: b must be uint32")
    if bits != 4 or group_size != 128:
        raise RuntimeError("gather_quantized_matvec: only 4-bit weights in groups of 128 are supported")
    if a.dim() != 2 or b.dim() != 3 or scales.dim() != 3 or scales.shape != biases.shape:
        raise RuntimeError("gather_quantized_matvec: expected a [M,N], b [E,K,N/8], scales/biases [E,K,N/128]")
    M, N = a.shape
    E, K = b.shape[0], b.shape[1]
    if N % group_size != 0 or b.shape[2] != N // 8 or tuple(scales.shape) != (E, K, N // group_size):
        raise RuntimeError("gather_quantized_matvec: shapes of a, b and scales do not describe one [E,K,N] weight stack")
    if expert_ids.dtype != torch.int32 or tuple(expert_ids.shape) != (M,):
        raise RuntimeError("gather_quantized_matvec: expert_ids must be int32 with one entry per row of a")
    _require_gpu("gather_quantized_matvec", scales, biases, a, b, expert_ids)
    for name, t in (("a", a), ("b", b), ("scales", scales), ("biases", biases), ("expert_ids", expert_ids)):
        if not t.is_contiguous():
            raise RuntimeError(f"gather_quantized_matvec: {name} must be contiguous")
    out = torch.empty((M, K), dtype=a.dtype, device=a.device)
    _check(_lib.tl_gather_quantized_matvec(_ptr(scales), _ptr(biases), _ptr(a), _ptr(b), _ptr(expert_ids), _ptr(out), M, N,
                                           K, E, int(group_size), int(bits), _DTYPES[a.dtype], _stream()))
    return out


def quantized_embedding(
    indices: torch.Tensor,
    scales: torch.Tensor,
    biases: torch.Tensor,
    weight: torch.Tensor,
    group_size: int,
    bits: int,
    stream=None,
) -> torch.Tensor:
    """Gather + dequantize rows of a packed table (reference quantized_matmul.cpp:82-101)."""
    if indices.dtype not in (torch.int32, torch.uint32) or weight.dtype not in (torch.uint32, torch.int32):
        raise RuntimeError("quantized_embedding: indices and weight must use 32-bit integers")
    if scales.dtype != biases.dtype or scales.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("quantized_embedding: scales and biases must have the same 16-bit dtype")
    if group_size != 128 or bits != 4 or scales.shape != biases.shape:
        raise RuntimeError("quantized_embedding: expected 4-bit weights with group size 128")
    dim = weight.shape[1] * 8
    if scales.shape[0] != weight.shape[0] or scales.shape[1] != dim // group_size:
        raise RuntimeError("quantized_embedding: incompatible parameter shapes")
    _require_gpu("quantized_embedding", indices, scales, biases, weight)
    indices = indices.contiguous()
    out = torch.empty((*indices.shape, dim), dtype=scales.dtype, device=scales.device)
    _check(
        _lib.tl_quantized_embedding(
            _ptr(indices), int(indices.dtype == torch.uint32), _ptr(scales.contiguous()), _ptr(biases.contiguous()),
            _ptr(weight.contiguous()), _ptr(out), indices.numel(), dim, weight.shape[0], group_size, bits,
            _DTYPES[scales.dtype], _stream(),
        )
    )
    return out


def _require_float(x: torch.Tensor, name: str) -> None:
    if x.dtype not in _DTYPES:
        raise RuntimeError(f"{name}: expected float32, float16, or bfloat16")


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float, stream=None) -> torch.Tensor:
    """Fused RMSNorm over the last dim (reference week2_kernels.cpp:36-42)."""
    _require_float(x, "rms_norm")
    if x.dtype != weight.dtype or weight.dim() != 1 or weight.shape[0] != x.shape[-1]:
        raise RuntimeError("rms_norm: weight must match the input dtype and final dimension")
    _require_gpu("rms_norm", x, weight)
    x = x.contiguous()
    weight = weight.contiguous()
    out = torch.empty_like(x)
    dim = x.shape[-1]
    rows = x.numel() // dim if dim else 0
    _check(_lib.tl_rms_norm(_ptr(x), _ptr(weight), _ptr(out), rows, dim, float(eps), _DTYPES[x.dtype], _stream()))
    return out


def rope(
    x: torch.Tensor, offsets: torch.Tensor, dims: int, base: float, traditional: bool = False, stream=None
) -> torch.Tensor:
    """Fused RoPE, x=[B,L,H,D], one int32 offset per batch row (reference week2_kernels.cpp:44-55)."""
    _require_float(x, "rope")
    if x.dim() != 4 or offsets.dtype != torch.int32 or offsets.dim() != 1 or offsets.shape[0] != x.shape[0]:
        raise RuntimeError("rope: expected x=[B,L,H,D] and one int32 offset per batch row")
    if dims <= 0 or dims > x.shape[3] or dims % 2 != 0:
        raise RuntimeError("rope: dims must be positive, even, and no larger than the head dimension")
    _require_gpu("rope", x, offsets)
    x = x.contiguous()
    offsets = offsets.contiguous()
    out = torch.empty_like(x)
    B, L, H, D = x.shape
    _check(
        _lib.tl_rope(_ptr(x), _ptr(offsets), _ptr(out), B, L, H, D, int(dims), float(base), int(bool(traditional)),
                     _DTYPES[x.dtype], _stream())
    )
    return out


def swiglu(gate: torch.Tensor, up: torch.Tensor, stream=None) -> torch.Tensor:
    """``silu(gate) * up`` (reference week2_kernels.cpp:57-63)."""
    _require_float(gate, "swiglu")
    if gate.dtype != up.dtype or gate.shape != up.shape:
        raise RuntimeError("swiglu: gate and up must have the same shape and dtype")
    _require_gpu("swiglu", gate, up)
    gate = gate.contiguous()
    up = up.contiguous()
    out = torch.empty_like(gate)
    _check(_lib.tl_swiglu(_ptr(gate), _ptr(up), _ptr(out), gate.numel(), _DTYPES[gate.dtype], _stream()))
    return out


def decode_attention(
    query: torch.Tensor,
    key: torch.Tensor,
    value: torch.Tensor,
    mask: torch.Tensor,
    scale: float,
    is_causal: bool,
    has_mask: bool,
    num_heads: int,
    num_kv_heads: int,
    stream=None,
) -> torch.Tensor:
    """Dense-KV online-softmax GQA attention (reference week2_kernels.cpp:65-84)."""
    _require_float(query, "decode_attention")
    if query.dtype != key.dtype or query.dtype != value.dtype or mask.dtype != torch.float32:
        raise RuntimeError("decode_attention: q, k, and v dtypes must match; mask must be float32")
    if (
        query.dim() != 3 or key.dim() != 3 or value.dim() != 3 or query.shape[2] > 256
        or query.shape[2] != key.shape[2] or query.shape[2] != value.shape[2] or key.shape != value.shape
        or num_heads % num_kv_heads != 0
    ):
        raise RuntimeError("decode_attention: incompatible attention shapes")
    if has_mask and (
        mask.dim() != 3 or mask.shape[0] != query.shape[0] or mask.shape[1] != query.shape[1]
        or mask.shape[2] != key.shape[1]
    ):
        raise RuntimeError("decode_attention: mask must have shape [B*Hq,L,S]")
    _require_gpu("decode_attention", query, key, value)
    if has_mask:
        _require_gpu("decode_attention", mask)
    query, key, value, mask = query.contiguous(), key.contiguous(), value.contiguous(), mask.contiguous()
    out = torch.empty_like(query)
    q_rows, L, D = query.shape
    S = key.shape[1]
    _check(
        _lib.tl_decode_attention(
            _ptr(query), _ptr(key), _ptr(value), _ptr(mask) if has_mask else None, _ptr(out), q_rows, L, S, D,
            int(num_heads), int(num_kv_heads), float(scale), int(bool(is_causal)), int(bool(has_mask)),
            _DTYPES[query.dtype], _stream(),
        )
    )
    return out


def paged_cache_update(pages: torch.Tensor, values: torch.Tensor, page_id: int, start: int, stream=None) -> torch.Tensor:
    """In-place write of ``values[1,H,len,D]`` into ``pages[P,H,page,D]``; returns ``pages`` itself
    (the reference output aliases the input buffer, paged_attention.cpp:46-49)."""
    if pages.dtype not in (torch.float32, torch.bfloat16) or values.dtype != pages.dtype:
        raise RuntimeError("paged_cache_update: pages and values must have the same float32 or bfloat16 dtype")
    if pages.dim() != 4 or values.dim() != 4 or values.shape[0] != 1:
        raise RuntimeError("paged_cache_update: expected pages [P, H, page_size, D] and values [1, H, length, D]")
    if values.shape[1] != pages.shape[1] or values.shape[3] != pages.shape[3]:
        raise RuntimeError("paged_cache_update: values must match the page head count and head dimension")
    if page_id < 0 or page_id >= pages.shape[0] or start < 0 or start + values.shape[2] > pages.shape[2]:
        raise RuntimeError("paged_cache_update: destination slice is outside page storage")
    _require_gpu("paged_cache_update", pages, values)
    if not pages.is_contiguous() or not values.is_contiguous():
        raise RuntimeError("paged_cache_update: pages and values must be contiguous")
    P, H, page_size, D = pages.shape
    _check(
        _lib.tl_paged_cache_update(_ptr(pages), _ptr(values), P, H, page_size, D, values.shape[2], int(page_id),
                                   int(start), _DTYPES[pages.dtype], _stream())
    )
    return pages


def paged_attention(
    query: torch.Tensor,
    key_pages: torch.Tensor,
    value_pages: torch.Tensor,
    block_table: torch.Tensor,
    context_lens: torch.Tensor,
    scale: float = 1.0,
    is_causal: bool = False,
    *,
    num_kv_heads: int,
    num_heads: int,
    stream=None,
    max_context_hint: int = 0,
) -> torch.Tensor:
    """Block-table attention over paged K/V (reference paged_attention.cpp:77-122 for the checks).

    ``max_context_hint`` (extension of the reference signature) is a host-known upper
    bound of ``context_lens`` used only to size the context split of the decode kernel.
    """
    if query.dtype not in (torch.float32, torch.bfloat16) or key_pages.dtype != query.dtype \
            or value_pages.dtype != query.dtype:
        raise RuntimeError(
            "paged_attention: q, key_pages, and value_pages must have the same float32 or bfloat16 dtype")
    if block_table.dtype != torch.int32 or context_lens.dtype != torch.int32:
        raise RuntimeError("paged_attention: block_table and context_lens must be int32")
    if query.dim() != 3:
        raise RuntimeError("paged_attention: q must be 3D [B * H_q, L, D]")
    if key_pages.dim() != 4 or value_pages.dim() != 4:
        raise RuntimeError("paged_attention: page tensors must be 4D [P, H_kv, page_size, D]")
    if block_table.dim() != 2 or context_lens.dim() != 1:
        raise RuntimeError("paged_attention: block_table must be 2D and context_lens must be 1D")
    if num_heads % num_kv_heads != 0:
        raise RuntimeError("paged_attention: num_heads must be divisible by num_kv_heads")
    if query.shape[0] % num_heads != 0:
        raise RuntimeError("paged_attention: q.shape[0] must be divisible by num_heads")
    if key_pages.shape != value_pages.shape:
        raise RuntimeError("paged_attention: key_pages and value_pages must have the same shape")
    if key_pages.shape[1] != num_kv_heads:
        raise RuntimeError("paged_attention: page tensor head count must equal num_kv_heads")
    if query.shape[2] != key_pages.shape[3]:
        raise RuntimeError("paged_attention: q and page tensors must have the same head dimension")
    if block_table.shape[0] != context_lens.shape[0]:
        raise RuntimeError("paged_attention: block_table and context_lens batch sizes must match")
    if query.shape[0] // num_heads != block_table.shape[0]:
        raise RuntimeError("paged_attention: q batch size must match block_table batch size")
    _require_gpu("paged_attention", query, key_pages, value_pages, block_table, context_lens)
    N, L, D = query.shape
    P, _, page_size, _ = key_pages.shape
    max_pages = block_table.shape[1]
    if D > 128:
        raise RuntimeError("paged_attention: head dimension must be at most 128")
    if L > 8 and query.dtype == torch.bfloat16 and D != 128:
        raise RuntimeError("paged_attention: bfloat16 prefill requires head dimension 128")
    for name, t in (("q", query), ("key_pages", key_pages), ("value_pages", value_pages),
                    ("block_table", block_table), ("context_lens", context_lens)):
        if not t.is_contiguous():
            raise RuntimeError(f"paged_attention: {name} must be contiguous")
    out = torch.empty_like(query)
    ws_bytes = _lib.tl_paged_attention_workspace_bytes(N, L, D, page_size, max_pages, num_heads, num_kv_heads,
                                                       int(max_context_hint))
    ws = _workspace(ws_bytes, query.device)
    _check(
        _lib.tl_paged_attention(
            _ptr(query), _ptr(key_pages), _ptr(value_pages), _ptr(block_table), _ptr(context_lens), _ptr(out), N, L,
            D, P, page_size, max_pages, int(num_heads), int(num_kv_heads), float(scale), int(bool(is_causal)),
            int(max_context_hint), _DTYPES[query.dtype], _ptr(ws) if ws is not None else None,
            ws.numel() if ws is not None else 0, _stream(),
        )
    )
    return out


__all__ = [
    "load_library",
    "quantized_matmul",
    "gather_quantized_matvec",
    "quantized_embedding",
    "rms_norm",
    "rope",
    "swiglu",
    "decode_attention",
    "paged_cache_update",
    "paged_attention",
]


# ---- kernel-level entry points of the decode path (include/tinyllm_engine.h, last section) -------------------------
PRO_NONE, PRO_RMSNORM, PRO_ATTN_MERGE, PRO_RMS_WEIGHTED = 0, 1, 2, 3
EPI_STORE, EPI_RESIDUAL, EPI_SWIGLU = 0, 1, 2
ATTN_PARTIAL_ROW = 128 + 4  # floats per (head, split) row of decode-attention split partials: 128 value sums, max, sum, 2 pad
# values of tl_linear_info.kernel (what RAN); the `kernel` argument of decode_linear selects: 0 engine routing, 1 GEMV,
# 2 skinny matmul (grid by shape), 3 / 4 skinny matmul on its one-shot / persistent grid, 5 register-resident batched matmul
LINEAR_KERNELS = {1: "qmv3 (fused MFMA GEMV)", 2: "qmm3 (skinny MFMA matmul + slice reduction)",
                  3: "qmv (packed-dot GEMV fallback)", 4: "prefill GEMM path", 5: "qmm6 (register-resident batched matmul)"}


class TiledW4:
    """One W4 matrix re-packed into the decode engine's tiled layout (tl_tiled_w4).  Keeps the checkpoint tensors alive."""

    def __init__(self, weight: torch.Tensor, scales: torch.Tensor, biases: torch.Tensor):
        _require_gpu("TiledW4", weight, scales, biases)
        if weight.dtype not in (torch.int32, torch.uint32) or weight.dim() != 2:
            raise RuntimeError("TiledW4: weight must be a 2-D uint32 tensor [rows, cols/8]")
        if scales.dtype != torch.bfloat16 or biases.dtype != torch.bfloat16:
            raise RuntimeError("TiledW4: the decode path is bfloat16")
        self.weight, self.scales, self.biases = weight.contiguous(), scales.contiguous(), biases.contiguous()
        self.rows, self.cols = int(weight.shape[0]), int(weight.shape[1]) * 8
        if tuple(self.scales.shape) != (self.rows, self.cols // 128) or self.scales.shape != self.biases.shape:
            raise RuntimeError("TiledW4: scales / biases must be [rows, cols/128]")
        w4 = TlW4(_ptr(self.weight), _ptr(self.scales), _ptr(self.biases), self.rows, self.cols)
        handle = _c_void_p()
        _check(_lib.tl_tiled_w4_create(ctypes.byref(w4), _stream(), ctypes.byref(handle)))
        self._h = handle

    def close(self) -> None:
        if getattr(self, "_h", None):
            torch.cuda.synchronize()
            _lib.tl_tiled_w4_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_linear(w: TiledW4, a: torch.Tensor | None, *, prologue: int = PRO_NONE, epilogue: int = EPI_STORE,
                  norm_weight: torch.Tensor | None = None, residual: torch.Tensor | None = None, eps: float = 1e-6,
                  kernel: int = 0, merge_partials: torch.Tensor | None = None, ss_in: torch.Tensor | None = None,
                  want_ss_out: bool = False, norm_out: torch.Tensor | None = None, fragment_order: bool = False,
                  fragment_rows: int | None = None):
    """One projection of a decode step over ``a`` [M <= 64, cols] bf16 (tl_decode_linear).  Returns (out, info) where info
    names the kernel that ran and its launch parameters.

    The keyword arguments after ``kernel`` reach the routes of tl_decode_linear_ex (include/tinyllm_engine.h):
    ``merge_partials`` [cols / 128, n_splits, 132] fp32 with prologue PRO_ATTN_MERGE (``a`` is None, one row);
    ``ss_in`` [M, n] fp32 partial sums of squares for prologue PRO_RMSNORM / PRO_RMS_WEIGHTED; ``want_ss_out`` /
    ``norm_out`` [rows] with the residual epilogue -- info then carries "ss_out" [M, rows / 16] and "out_w" [M, rows].
    ``fragment_order``: the weighted rows on either side (``a`` with PRO_RMS_WEIGHTED through kernel 5: [ceil16(M), cols] in fragment
    order; "out_w": [ceil16(M), rows]) lie in the batched step's fragment order (fragment_order_of / rows_from_fragment_order)."""
    extended = merge_partials is not None or ss_in is not None or want_ss_out or norm_out is not None or fragment_order \
        or prologue in (PRO_ATTN_MERGE, PRO_RMS_WEIGHTED)
    if prologue == PRO_ATTN_MERGE:
        if merge_partials is None or a is not None:
            raise RuntimeError("decode_linear: the merging prologue takes merge_partials and no activation rows")
        _require_gpu("decode_linear", merge_partials)
        if merge_partials.dtype != torch.float32 or merge_partials.dim() != 3 or not merge_partials.is_contiguous() \
                or merge_partials.shape[0] * 128 != w.cols or merge_partials.shape[2] != ATTN_PARTIAL_ROW:
            raise RuntimeError("decode_linear: merge_partials must be contiguous float32 [cols / 128, n_splits, 132]")
        M, device = 1, merge_partials.device
    else:
        _require_gpu("decode_linear", a)
        if a.dtype != torch.bfloat16 or a.dim() != 2 or a.shape[1] != w.cols or not a.is_contiguous():
            raise RuntimeError("decode_linear: a must be a contiguous bfloat16 [M, cols] tensor")
        M, device = int(a.shape[0]), a.device
        if fragment_order and prologue == PRO_RMS_WEIGHTED:  # rows padded to 16: the caller says how many are real
            if M % 16 != 0 or fragment_rows is None or not (M - 16 < fragment_rows <= M):
                raise RuntimeError("decode_linear: rows in fragment order come as [ceil16(M), cols] with fragment_rows = M")
            M = int(fragment_rows)
    out_cols = w.rows // 2 if epilogue == EPI_SWIGLU else w.rows
    out = torch.empty((M, out_cols), dtype=torch.bfloat16, device=device)
    if residual is not None and (residual.dtype != torch.bfloat16 or tuple(residual.shape) != (M, w.rows)
                                 or not residual.is_contiguous()):
        raise RuntimeError("decode_linear: residual must be a contiguous bfloat16 [M, rows] tensor")
    if norm_weight is not None and (norm_weight.dtype != torch.bfloat16 or tuple(norm_weight.shape) != (w.cols,)):
        raise RuntimeError("decode_linear: norm_weight must be bfloat16 [cols]")
    ws_bytes = _lib.tl_decode_linear_workspace_bytes(M, w.rows, w.cols)
    ws = _workspace(ws_bytes, device)
    info = TlLinearInfo()
    common = (w._h, _ptr(a) if a is not None else None, _ptr(out), M, int(prologue), int(epilogue),
              _ptr(norm_weight) if norm_weight is not None else None, _ptr(residual) if residual is not None else None,
              float(eps), int(kernel), _ptr(ws) if ws is not None else None, ws.numel() if ws is not None else 0, _stream())
    extra = {}
    if not extended:
        _check(_lib.tl_decode_linear(*common, ctypes.byref(info)))
    else:
        ex = TlLinearEx()
        if merge_partials is not None:
            ex.merge_ws_dev, ex.n_splits = _ptr(merge_partials), int(merge_partials.shape[1])
        if ss_in is not None:
            if ss_in.dtype != torch.float32 or ss_in.dim() != 2 or ss_in.shape[0] != M or not ss_in.is_contiguous():
                raise RuntimeError("decode_linear: ss_in must be contiguous float32 [M, partials]")
            _require_gpu("decode_linear", ss_in)
            ex.ss_in_dev, ex.ss_in_n = _ptr(ss_in), int(ss_in.shape[1])
        if want_ss_out:
            extra["ss_out"] = torch.full((M, w.rows // 16), float("nan"), dtype=torch.float32, device=device)
            ex.ss_out_dev = _ptr(extra["ss_out"])
        if norm_out is not None:
            if norm_out.dtype != torch.bfloat16 or tuple(norm_out.shape) != (w.rows,):
                raise RuntimeError("decode_linear: norm_out must be bfloat16 [rows]")
            extra["out_w"] = torch.zeros(((M + 15) // 16 * 16 if fragment_order else M, w.rows), dtype=torch.bfloat16, device=device)
            ex.norm_out_dev, ex.out_w_dev = _ptr(norm_out), _ptr(extra["out_w"])
        ex.fragment_order = 1 if fragment_order else 0
        _check(_lib.tl_decode_linear_ex(*common, ctypes.byref(ex), ctypes.byref(info)))
    return out, {"kernel": info.kernel, "kernel_name": LINEAR_KERNELS.get(info.kernel, "?"), "launches": info.launches,
                 "rows_per_pass": info.rows_per_pass, "p": list(info.p), **extra}


def prefill_weights_bf16(weight: torch.Tensor, scales: torch.Tensor, biases: torch.Tensor) -> torch.Tensor:
    """W4 [rows, cols/8] -> bf16 [rows, cols] = bf16(q * scale + bias): the B operand of the reference's tile GEMM (tl_prefill_weights_bf16)."""
    _require_gpu("prefill_weights_bf16", weight, scales, biases)
    rows, cols = int(weight.shape[0]), int(weight.shape[1]) * 8
    w4 = TlW4(_ptr(weight.contiguous()), _ptr(scales.contiguous()), _ptr(biases.contiguous()), rows, cols)
    out = torch.empty((rows, cols), dtype=torch.bfloat16, device=weight.device)
    _check(_lib.tl_prefill_weights_bf16(ctypes.byref(w4), _ptr(out), _stream()))
    return out


def prefill_matmul_bf16(a: torch.Tensor, w_bf16: torch.Tensor, *, epilogue: int = 0, residual: torch.Tensor | None = None) -> torch.Tensor:
    """out = epilogue(a @ w_bf16^T): the plain bf16 GEMM large prefill chunks run on (tl_prefill_matmul_bf16, csrc/gemm8.h)."""
    _require_gpu("prefill_matmul_bf16", a, w_bf16)
    if a.dtype != torch.bfloat16 or w_bf16.dtype != torch.bfloat16 or a.dim() != 2 or w_bf16.dim() != 2 or a.shape[1] != w_bf16.shape[1]:
        raise RuntimeError("prefill_matmul_bf16: a [M, cols] and w_bf16 [rows, cols] must be bfloat16 with the same cols")
    M, rows, cols = int(a.shape[0]), int(w_bf16.shape[0]), int(a.shape[1])
    out = torch.empty((M, rows // 2 if epilogue == 2 else rows), dtype=torch.bfloat16, device=a.device)
    res = residual.contiguous() if residual is not None else None
    _check(_lib.tl_prefill_matmul_bf16(_ptr(a.contiguous()), _ptr(w_bf16.contiguous()), _ptr(out), M, rows, cols, int(epilogue),
                                       _ptr(res) if res is not None else None, _stream()))
    return out


def fragment_order_of(rows: torch.Tensor) -> torch.Tensor:
    """[M, cols] -> the batched step's fragment order, [ceil16(M), cols] (zero rows appended): [16-row block][128-column group]
    [k-step t][lane = r + 16 c][8 elements] = row 16 block + r, columns 128 g + 32 c + 8 t .. + 7 (csrc/qmm6.h, qmm6_frag_offset)."""
    M, cols = rows.shape
    pad = (M + 15) // 16 * 16
    x = torch.zeros((pad, cols), dtype=rows.dtype, device=rows.device)
    x[:M] = rows
    x = x.reshape(pad // 16, 16, cols // 128, 4, 4, 8)      # [block, r, g, c, t, e]
    return x.permute(0, 2, 4, 3, 1, 5).contiguous().reshape(pad, cols)  # [block, g, t, c, r, e]


def rows_from_fragment_order(frag: torch.Tensor, M: int) -> torch.Tensor:
    """Inverse of fragment_order_of: [ceil16(M), cols] in fragment order -> [M, cols] row-major."""
    pad, cols = frag.shape
    x = frag.reshape(pad // 16, cols // 128, 4, 4, 16, 8)   # [block, g, t, c, r, e]
    return x.permute(0, 4, 1, 3, 2, 5).contiguous().reshape(pad, cols)[:M]


def decode_attention_fused(qkv: torch.Tensor, q_norm: torch.Tensor, k_norm: torch.Tensor, key_pages: torch.Tensor,
                           value_pages: torch.Tensor, block_table: torch.Tensor, context_lens: torch.Tensor, *,
                           num_heads: int, num_kv_heads: int, rope_theta: float, eps: float,
                           max_context: int) -> tuple[torch.Tensor, dict]:
    """The attention launch of one decode layer (tl_decode_attention_fused): q/k-norm + RoPE + in-place KV append +
    paged GQA attention over ``context_lens + 1`` tokens.  qkv [B, (Hq + 2 Hkv) D]; pages [P, Hkv, page, D] (modified)."""
    _require_gpu("decode_attention_fused", qkv, q_norm, k_norm, key_pages, value_pages, block_table, context_lens)
    B = int(qkv.shape[0])
    P, Hkv, page, D = (int(x) for x in key_pages.shape)
    if Hkv != num_kv_heads or tuple(value_pages.shape) != tuple(key_pages.shape):
        raise RuntimeError("decode_attention_fused: page pools must be [P, num_kv_heads, page_size, D] and alike")
    if qkv.dim() != 2 or qkv.shape[1] != (num_heads + 2 * num_kv_heads) * D:
        raise RuntimeError("decode_attention_fused: qkv must be [batch, (Hq + 2 Hkv) * D]")
    for name, t in (("qkv", qkv), ("key_pages", key_pages), ("value_pages", value_pages), ("q_norm", q_norm),
                    ("k_norm", k_norm)):
        if t.dtype != torch.bfloat16 or not t.is_contiguous():
            raise RuntimeError(f"decode_attention_fused: {name} must be contiguous bfloat16")
    if block_table.dtype != torch.int32 or context_lens.dtype != torch.int32 or block_table.dim() != 2 \
            or block_table.shape[0] != B or tuple(context_lens.shape) != (B,):
        raise RuntimeError("decode_attention_fused: block_table [B, max_pages] and context_lens [B] must be int32")
    out = torch.empty((B, num_heads * D), dtype=torch.bfloat16, device=qkv.device)
    ws_bytes = _lib.tl_decode_attention_fused_workspace_bytes(B, num_heads, D)
    ws = _workspace(ws_bytes, qkv.device)
    info = TlAttentionInfo()
    _check(_lib.tl_decode_attention_fused(_ptr(qkv), _ptr(q_norm), _ptr(k_norm), _ptr(key_pages), _ptr(value_pages),
                                          _ptr(block_table.contiguous()), _ptr(context_lens), _ptr(out), B, num_heads,
                                          num_kv_heads, D, page, int(block_table.shape[1]), float(rope_theta), float(eps),
                                          int(max_context), _ptr(ws), ws.numel(), _stream(), ctypes.byref(info)))
    return out, {name: getattr(info, name) for name, _ in info._fields_}


# ---- FP8 (E4M3) KV pages: the quantised twins of paged_cache_update / paged_attention -------------------------------------
# (include/tinyllm_hip.h "FP8 KV pages"; SURVEY section 8f row 4 -- the reference has no quantised cache, README.md:134-135, so these
# are NOT part of the reference's interface and stay out of __all__)
KV_BF16, KV_FP8_E4M3 = 0, 1


def kv_fp8_quantize_rows(values: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """values [..., 128] bfloat16 -> (codes [..., 128] uint8, scales [...] float32): OCP E4M3 codes and one power-of-two scale per row."""
    _require_gpu("kv_fp8_quantize_rows", values)
    if values.dtype != torch.bfloat16 or values.shape[-1] != 128 or not values.is_contiguous():
        raise RuntimeError("kv_fp8_quantize_rows: contiguous bfloat16 rows of 128 values")
    rows = values.numel() // 128
    codes = torch.empty(values.shape, dtype=torch.uint8, device=values.device)
    scales = torch.empty(values.shape[:-1], dtype=torch.float32, device=values.device)
    _check(_lib.tl_kv_fp8_quantize_rows(_ptr(values), _ptr(codes), _ptr(scales), rows, 128, _stream()))
    return codes, scales


def kv_fp8_dequantize_rows(codes: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """(codes [..., 128] uint8, scales [...] float32) -> bfloat16 [..., 128] (exact)."""
    _require_gpu("kv_fp8_dequantize_rows", codes, scales)
    if codes.dtype != torch.uint8 or scales.dtype != torch.float32 or codes.shape[-1] != 128 or tuple(codes.shape[:-1]) != tuple(scales.shape) \
            or not codes.is_contiguous() or not scales.is_contiguous():
        raise RuntimeError("kv_fp8_dequantize_rows: contiguous uint8 codes [..., 128] and float32 scales [...]")
    out = torch.empty(codes.shape, dtype=torch.bfloat16, device=codes.device)
    _check(_lib.tl_kv_fp8_dequantize_rows(_ptr(codes), _ptr(scales), _ptr(out), codes.numel() // 128, 128, _stream()))
    return out


def paged_cache_update_fp8(pages: torch.Tensor, page_scales: torch.Tensor, values: torch.Tensor, page_id: int,
                           start: int) -> tuple[torch.Tensor, torch.Tensor]:
    """In-place quantising write of bfloat16 ``values[1,H,len,128]`` into ``pages[P,H,page,128]`` uint8 + ``page_scales[P,H,page]``."""
    if pages.dtype != torch.uint8 or page_scales.dtype != torch.float32 or values.dtype != torch.bfloat16:
        raise RuntimeError("paged_cache_update_fp8: uint8 pages, float32 page_scales, bfloat16 values")
    if pages.dim() != 4 or values.dim() != 4 or values.shape[0] != 1 or tuple(page_scales.shape) != tuple(pages.shape[:3]):
        raise RuntimeError("paged_cache_update_fp8: expected pages [P, H, page_size, 128], page_scales [P, H, page_size] and values [1, H, length, 128]")
    if values.shape[1] != pages.shape[1] or values.shape[3] != pages.shape[3]:
        raise RuntimeError("paged_cache_update_fp8: values must match the page head count and head dimension")
    if page_id < 0 or page_id >= pages.shape[0] or start < 0 or start + values.shape[2] > pages.shape[2]:
        raise RuntimeError("paged_cache_update_fp8: destination slice is outside page storage")
    _require_gpu("paged_cache_update_fp8", pages, page_scales, values)
    if not pages.is_contiguous() or not values.is_contiguous() or not page_scales.is_contiguous():
        raise RuntimeError("paged_cache_update_fp8: pages, page_scales and values must be contiguous")
    P, H, page_size, D = pages.shape
    _check(_lib.tl_paged_cache_update_fp8(_ptr(pages), _ptr(page_scales), _ptr(values), P, H, page_size, D, values.shape[2],
                                          int(page_id), int(start), _stream()))
    return pages, page_scales


def paged_attention_fp8(query: torch.Tensor, key_pages: torch.Tensor, key_scales: torch.Tensor, value_pages: torch.Tensor,
                        value_scales: torch.Tensor, block_table: torch.Tensor, context_lens: torch.Tensor, scale: float = 1.0,
                        is_causal: bool = False, *, num_kv_heads: int, num_heads: int, max_context_hint: int = 0) -> torch.Tensor:
    """paged_attention over FP8 pages: bfloat16 ``query`` [B * Hq, L, 128], uint8 pages [P, Hkv, page, 128], float32 scales [P, Hkv, page]."""
    if query.dtype != torch.bfloat16 or key_pages.dtype != torch.uint8 or value_pages.dtype != torch.uint8 \
            or key_scales.dtype != torch.float32 or value_scales.dtype != torch.float32:
        raise RuntimeError("paged_attention_fp8: bfloat16 q, uint8 pages, float32 scales")
    if block_table.dtype != torch.int32 or context_lens.dtype != torch.int32:
        raise RuntimeError("paged_attention_fp8: block_table and context_lens must be int32")
    if query.dim() != 3 or key_pages.dim() != 4 or key_pages.shape != value_pages.shape or block_table.dim() != 2 or context_lens.dim() != 1:
        raise RuntimeError("paged_attention_fp8: q [B * H_q, L, D], pages [P, H_kv, page_size, D], block_table [B, max_pages], context_lens [B]")
    if tuple(key_scales.shape) != tuple(key_pages.shape[:3]) or tuple(value_scales.shape) != tuple(value_pages.shape[:3]):
        raise RuntimeError("paged_attention_fp8: scales must be [P, H_kv, page_size]")
    if num_heads % num_kv_heads != 0 or query.shape[0] % num_heads != 0 or key_pages.shape[1] != num_kv_heads:
        raise RuntimeError("paged_attention_fp8: incompatible head counts")
    if query.shape[2] != 128 or key_pages.shape[3] != 128:
        raise RuntimeError("paged_attention_fp8: FP8 pages need head dimension 128")
    if query.shape[0] // num_heads != block_table.shape[0] or block_table.shape[0] != context_lens.shape[0]:
        raise RuntimeError("paged_attention_fp8: q batch size must match block_table and context_lens")
    tensors = (query, key_pages, key_scales, value_pages, value_scales, block_table, context_lens)
    _require_gpu("paged_attention_fp8", *tensors)
    if not all(t.is_contiguous() for t in tensors):
        raise RuntimeError("paged_attention_fp8: every tensor must be contiguous")
    N, L, D = query.shape
    P, _, page_size, _ = key_pages.shape
    max_pages = block_table.shape[1]
    out = torch.empty_like(query)
    ws_bytes = _lib.tl_paged_attention_workspace_bytes(N, L, D, page_size, max_pages, num_heads, num_kv_heads, int(max_context_hint))
    ws = _workspace(ws_bytes, query.device)
    _check(_lib.tl_paged_attention_fp8(
        _ptr(query), _ptr(key_pages), _ptr(key_scales), _ptr(value_pages), _ptr(value_scales), _ptr(block_table), _ptr(context_lens),
        _ptr(out), N, L, D, P, page_size, max_pages, int(num_heads), int(num_kv_heads), float(scale), int(bool(is_causal)),
        int(max_context_hint), _ptr(ws) if ws is not None else None, ws.numel() if ws is not None else 0, _stream()))
    return out


def decode_attention_fused_fp8(qkv: torch.Tensor, q_norm: torch.Tensor, k_norm: torch.Tensor, key_pages: torch.Tensor,
                               key_scales: torch.Tensor, value_pages: torch.Tensor, value_scales: torch.Tensor,
                               block_table: torch.Tensor, context_lens: torch.Tensor, *, num_heads: int, num_kv_heads: int,
                               rope_theta: float, eps: float, max_context: int) -> tuple[torch.Tensor, dict]:
    """decode_attention_fused over FP8 pages (tl_decode_attention_fused_fp8): the appended K / V rows are quantised in place."""
    _require_gpu("decode_attention_fused_fp8", qkv, q_norm, k_norm, key_pages, key_scales, value_pages, value_scales, block_table, context_lens)
    B = int(qkv.shape[0])
    P, Hkv, page, D = (int(x) for x in key_pages.shape)
    if D != 128 or Hkv != num_kv_heads or tuple(value_pages.shape) != tuple(key_pages.shape) \
            or tuple(key_scales.shape) != (P, Hkv, page) or tuple(value_scales.shape) != (P, Hkv, page):
        raise RuntimeError("decode_attention_fused_fp8: pages [P, num_kv_heads, page_size, 128] uint8 and scales [P, num_kv_heads, page_size]")
    if qkv.dim() != 2 or qkv.shape[1] != (num_heads + 2 * num_kv_heads) * D:
        raise RuntimeError("decode_attention_fused_fp8: qkv must be [batch, (Hq + 2 Hkv) * D]")
    for name, t, dt in (("qkv", qkv, torch.bfloat16), ("q_norm", q_norm, torch.bfloat16), ("k_norm", k_norm, torch.bfloat16),
                        ("key_pages", key_pages, torch.uint8), ("value_pages", value_pages, torch.uint8),
                        ("key_scales", key_scales, torch.float32), ("value_scales", value_scales, torch.float32)):
        if t.dtype != dt or not t.is_contiguous():
            raise RuntimeError(f"decode_attention_fused_fp8: {name} must be contiguous {dt}")
    if block_table.dtype != torch.int32 or context_lens.dtype != torch.int32 or block_table.dim() != 2 \
            or block_table.shape[0] != B or tuple(context_lens.shape) != (B,):
        raise RuntimeError("decode_attention_fused_fp8: block_table [B, max_pages] and context_lens [B] must be int32")
    out = torch.empty((B, num_heads * D), dtype=torch.bfloat16, device=qkv.device)
    ws = _workspace(_lib.tl_decode_attention_fused_workspace_bytes(B, num_heads, D), qkv.device)
    info = TlAttentionInfo()
    _check(_lib.tl_decode_attention_fused_fp8(_ptr(qkv), _ptr(q_norm), _ptr(k_norm), _ptr(key_pages), _ptr(key_scales), _ptr(value_pages),
                                              _ptr(value_scales), _ptr(block_table.contiguous()), _ptr(context_lens), _ptr(out), B,
                                              num_heads, num_kv_heads, D, page, int(block_table.shape[1]), float(rope_theta), float(eps),
                                              int(max_context), _ptr(ws), ws.numel(), _stream(), ctypes.byref(info)))
    return out, {name: getattr(info, name) for name, _ in info._fields_}


def paged_attention_waves(waves: int = 0) -> int:
    """Test / lab hook (tl_paged_attention_waves): 8 or 4 waves per workgroup in the bf16 FlashAttention prefill kernel; returns the previous value."""
    return int(_lib.tl_paged_attention_waves(int(waves)))
