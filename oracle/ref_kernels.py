"""ctypes binding of oracle/_ref/libref_metal_kernels.so: the REFERENCE'S OWN Metal kernel sources
(src/extensions_ref/src/{week2_kernels,quantized_matmul,paged_attention}.metal), compiled for the host from where they lie
against the Metal-on-CPU shim (oracle/metal_shim, recipe in oracle/Makefile).  TEST INFRASTRUCTURE: used to hold the numpy
oracle against the reference's kernel code; exists only where /root/reference was present at build time.

Arrays go in and out as numpy: float32 for 'f32', uint16 bit patterns for 'f16' / 'bf16' (helpers below convert)."""

from __future__ import annotations

import ctypes
from pathlib import Path

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "_ref" / "libref_metal_kernels.so"
DTYPE = {"f32": 0, "f16": 1, "bf16": 2}
_lib = None


def available() -> bool:
    return LIB_PATH.exists()


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(str(LIB_PATH))
    return _lib


def to_storage(a, dtype: str) -> np.ndarray:
    """float32 values (already representable in `dtype`) -> the kernel's storage array."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    if dtype == "f32":
        return a
    if dtype == "f16":
        return a.astype(np.float16).view(np.uint16)
    return (a.view(np.uint32) >> 16).astype(np.uint16)


def from_storage(a: np.ndarray, dtype: str) -> np.ndarray:
    if dtype == "f32":
        return np.asarray(a, dtype=np.float32)
    if dtype == "f16":
        return a.view(np.float16).astype(np.float32)
    return (a.astype(np.uint32) << 16).view(np.float32)


def _empty(shape, dtype: str) -> np.ndarray:
    return np.zeros(shape, dtype=np.float32 if dtype == "f32" else np.uint16)


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def rms_norm(x, weight, eps: float, dtype: str) -> np.ndarray:
    rows, dim = x.shape
    xs, ws, out = to_storage(x, dtype), to_storage(weight, dtype), _empty((rows, dim), dtype)
    lib().ref_week2_rms_norm(_p(xs), _p(ws), _p(out), rows, dim, ctypes.c_float(eps), DTYPE[dtype])
    return from_storage(out, dtype)


def rope(x, offsets, dims: int, base: float, traditional: bool, dtype: str) -> np.ndarray:
    B, L, H, D = x.shape
    xs, out = to_storage(x, dtype), _empty(x.shape, dtype)
    off = np.ascontiguousarray(offsets, dtype=np.int32)
    lib().ref_week2_rope(_p(xs), _p(off), _p(out), B, L, H, D, dims, ctypes.c_float(base), int(traditional), DTYPE[dtype])
    return from_storage(out, dtype)


def swiglu(gate, up, dtype: str) -> np.ndarray:
    gs, us, out = to_storage(gate, dtype), to_storage(up, dtype), _empty(gate.shape, dtype)
    lib().ref_week2_swiglu(_p(gs), _p(us), _p(out), int(gate.size), DTYPE[dtype])
    return from_storage(out, dtype)


def decode_attention(q, k, v, scale: float, num_heads: int, num_kv_heads: int, is_causal: bool, mask, dtype: str) -> np.ndarray:
    q_rows, length, dim = q.shape
    context = k.shape[1]
    qs, ks, vs, out = to_storage(q, dtype), to_storage(k, dtype), to_storage(v, dtype), _empty(q.shape, dtype)
    m = np.ascontiguousarray(mask, dtype=np.float32) if mask is not None else np.zeros((1,), dtype=np.float32)
    lib().ref_week2_decode_attention(_p(qs), _p(ks), _p(vs), _p(m), _p(out), q_rows, length, context, dim, num_heads, num_kv_heads,
                                     ctypes.c_float(scale), int(is_causal), int(mask is not None), DTYPE[dtype])
    return from_storage(out, dtype)


def _qmm(fn, scales, biases, a, b, dtype: str) -> np.ndarray:
    M, N = a.shape
    K = b.shape[0]
    ss, bs, xs, out = to_storage(scales, dtype), to_storage(biases, dtype), to_storage(a, dtype), _empty((M, K), dtype)
    packed = np.ascontiguousarray(b, dtype=np.uint32)
    fn(_p(ss), _p(bs), _p(xs), _p(packed), _p(out), M, N, K, DTYPE[dtype])
    return from_storage(out, dtype)


def quantized_matmul_vanilla(scales, biases, a, b, dtype: str) -> np.ndarray:
    return _qmm(lib().ref_quantized_matmul_vanilla, scales, biases, a, b, dtype)


def quantized_matvec_x4_fast(scales, biases, a, b, dtype: str) -> np.ndarray:
    return _qmm(lib().ref_quantized_matvec_x4_fast, scales, biases, a, b, dtype)


def splitk_reduce(partials, dtype: str) -> np.ndarray:
    split_k, elements = partials.shape[0], int(np.prod(partials.shape[1:]))
    ps, out = to_storage(partials, dtype), _empty(partials.shape[1:], dtype)
    lib().ref_quantized_matmul_splitk_reduce(_p(ps), _p(out), elements, split_k, DTYPE[dtype])
    return from_storage(out, dtype)


def quantized_embedding(indices, scales, biases, weights, dtype: str) -> np.ndarray:
    idx = np.ascontiguousarray(indices, dtype=np.int32)
    dim = weights.shape[1] * 8
    ss, bs, out = to_storage(scales, dtype), to_storage(biases, dtype), _empty((idx.size, dim), dtype)
    packed = np.ascontiguousarray(weights, dtype=np.uint32)
    lib().ref_quantized_embedding(_p(idx), _p(ss), _p(bs), _p(packed), _p(out), int(idx.size), dim, DTYPE[dtype])
    return from_storage(out, dtype).reshape(*idx.shape, dim)


def paged_cache_update(pages, values, page_id: int, start: int, dtype: str) -> np.ndarray:
    P, H, page_size, D = pages.shape
    length = values.shape[-2]
    ps, vs = to_storage(pages, dtype).copy(), to_storage(values, dtype)
    lib().ref_paged_cache_update(_p(vs), _p(ps), H, length, D, page_size, page_id, start, DTYPE[dtype])
    return from_storage(ps, dtype)


def paged_attention_decode(q, key_pages, value_pages, block_table, context_lens, scale: float, is_causal: bool, num_kv_heads: int,
                           num_heads: int, dtype: str, fixed_d128: bool = False) -> np.ndarray:
    N, L, D = q.shape
    P, Hkv, page_size, _ = key_pages.shape
    table = np.ascontiguousarray(block_table, dtype=np.int32)
    ctx = np.ascontiguousarray(context_lens, dtype=np.int32)
    qs, ks, vs, out = to_storage(q, dtype), to_storage(key_pages, dtype), to_storage(value_pages, dtype), _empty(q.shape, dtype)
    lib().ref_paged_attention_decode(_p(qs), _p(ks), _p(vs), _p(table), _p(ctx), _p(out), N, L, D, page_size, table.shape[1], int(is_causal),
                                     num_kv_heads, num_heads, ctypes.c_float(scale), DTYPE[dtype], int(fixed_d128))
    return from_storage(out, dtype)


def paged_attention_scalar_f32(q, key_pages, value_pages, block_table, context_lens, scale: float, is_causal: bool, num_kv_heads: int,
                               num_heads: int) -> np.ndarray:
    N, L, D = q.shape
    page_size = key_pages.shape[2]
    table = np.ascontiguousarray(block_table, dtype=np.int32)
    ctx = np.ascontiguousarray(context_lens, dtype=np.int32)
    qs, ks, vs = (np.ascontiguousarray(t, dtype=np.float32) for t in (q, key_pages, value_pages))
    out = np.zeros(q.shape, dtype=np.float32)
    lib().ref_paged_attention_scalar_f32(_p(qs), _p(ks), _p(vs), _p(table), _p(ctx), _p(out), N, L, D, page_size, table.shape[1], int(is_causal),
                                         num_kv_heads, num_heads, ctypes.c_float(scale))
    return out
