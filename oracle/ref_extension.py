"""The reference's extension API (`tiny_llm_ext_ref`: the eight primitives of src/extensions_ref/bindings.cpp:14-47) answered by
the REFERENCE'S OWN KERNELS run on the host (oracle/_ref, oracle/ref_kernels.py), with the kernel selection of the reference's
C++ primitives (quantized_matmul.cpp:137-207, paged_attention.cpp:168-224, week2_kernels.cpp).  Tensors are torch tensors on the
CPU, as the facade's `mlx.core` hands them around.  TEST INFRASTRUCTURE: it lets the reference's Python sources run on the
reference's kernel code in the build container (tests/golden/make_reference_stack_vectors.py).

Not available on the host: the 32x32 tile GEMM (more than 8 activation rows with use_simdgroup) and the MMA FlashAttention
(bf16, more than 8 query rows) -- they need MLX's steel headers; callers keep to at most 8 rows per call (the reference's
chunked prefill takes any chunk size).
"""

from __future__ import annotations

import numpy as np
import torch

from . import ref_kernels as K

_NAME = {torch.float32: "f32", torch.float16: "f16", torch.bfloat16: "bf16"}


def _np(t: torch.Tensor) -> np.ndarray:
    return t.detach().to(torch.float32).cpu().numpy()


def _t(a: np.ndarray, like: torch.Tensor) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(like.dtype)


def load_library(path) -> None:
    K.lib()


def quantized_matmul(scales, biases, group_size, bits, a, b, transpose_b=False, use_simdgroup=True, use_split_k=False, stream=None):
    if group_size != 128 or bits != 4 or not transpose_b:
        raise RuntimeError("quantized_matmul: only 4-bit groups of 128 with transpose_b=True")
    name = _NAME[a.dtype]
    packed = b.detach().cpu().numpy().view(np.uint32)
    M = a.shape[0]
    if use_simdgroup and M <= 8:
        out = K.quantized_matvec_x4_fast(_np(scales), _np(biases), _np(a), packed, name)
    elif not use_simdgroup:
        out = K.quantized_matmul_vanilla(_np(scales), _np(biases), _np(a), packed, name)
    else:
        raise RuntimeError("quantized_matmul: the tile GEMM (more than 8 rows) is not built for the host (MLX steel headers)")
    return _t(out, a)


def quantized_embedding(indices, scales, biases, weight, group_size, bits, stream=None):
    name = _NAME[scales.dtype]
    out = K.quantized_embedding(indices.detach().cpu().numpy().astype(np.int32), _np(scales), _np(biases), weight.detach().cpu().numpy().view(np.uint32), name)
    return _t(out, scales)


def rms_norm(x, weight, eps, stream=None):
    name = _NAME[x.dtype]
    out = K.rms_norm(_np(x).reshape(-1, x.shape[-1]), _np(weight), float(eps), name)
    return _t(out, x).reshape(x.shape)


def rope(x, offsets, dims, base, traditional=False, stream=None):
    return _t(K.rope(_np(x), offsets.detach().cpu().numpy().astype(np.int32), int(dims), float(base), bool(traditional), _NAME[x.dtype]), x)


def swiglu(gate, up, stream=None):
    return _t(K.swiglu(_np(gate), _np(up), _NAME[gate.dtype]), gate)


def decode_attention(query, key, value, mask, scale, is_causal, has_mask, num_heads, num_kv_heads, stream=None):
    m = _np(mask) if has_mask else None
    return _t(K.decode_attention(_np(query), _np(key), _np(value), float(scale), int(num_heads), int(num_kv_heads), bool(is_causal), m, _NAME[query.dtype]), query)


def paged_cache_update(pages, values, page_id, start, stream=None):
    out = K.paged_cache_update(_np(pages), _np(values), int(page_id), int(start), _NAME[pages.dtype])
    pages.copy_(_t(out, pages))  # the reference's output aliases its input buffer (paged_attention.cpp:46-49)
    return pages


def paged_attention(query, key_pages, value_pages, block_table, context_lens, scale=1.0, is_causal=False, num_kv_heads=0, num_heads=0, stream=None):
    name = _NAME[query.dtype]
    N, L, D = query.shape
    table = block_table.detach().cpu().numpy().astype(np.int32)
    ctx = context_lens.detach().cpu().numpy().astype(np.int32)
    if L <= 8:
        out = K.paged_attention_decode(_np(query), _np(key_pages), _np(value_pages), table, ctx, float(scale), bool(is_causal), int(num_kv_heads), int(num_heads),
                                       name, fixed_d128=(name == "bf16" and D == 128))
    elif name == "f32":
        out = K.paged_attention_scalar_f32(_np(query), _np(key_pages), _np(value_pages), table, ctx, float(scale), bool(is_causal), int(num_kv_heads), int(num_heads))
    else:
        raise RuntimeError("paged_attention: the bf16 MMA FlashAttention kernel is not built for the host (MLX steel headers)")
    return _t(out, query)
