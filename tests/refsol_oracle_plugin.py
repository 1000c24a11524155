"""pytest plugin (TEST INFRASTRUCTURE, CPU only): lets the reference's UNMODIFIED test files run in a container without a
GPU, through the import facade (tiny-llm_amd/compat) and the product's host mirror (tiny_llm_hip), by standing in for
libtinyllm_hip.so at the C ABI.

The product's extension binding (tiny_llm_ext_hip) keeps every Python-side check; only the two things that need a GPU are
replaced: ``_require_gpu`` (a no-op here) and ``_lib`` -- a ``FakeLib`` whose ``tl_*`` entry points take the same raw
pointers, sizes and dtype tags as include/tinyllm_hip.h declares, view them as numpy arrays in HOST memory and compute the
result with the numpy oracle (oracle/tiny_oracle.py).  What this validates: the facade, the operator API surface, the host
logic (caches, pools, scheduler, models) against the reference's own tests.  What it does NOT validate: the HIP kernels --
those are held against the same oracle by tests/test_*_gpu.py on the MI355X.  The product never loads this file.

Used by tests/test_refsol_facade_cpu.py:  pytest -p refsol_oracle_plugin <reference test files>.
"""

import ctypes
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for extra in (ROOT, ROOT / "tests", ROOT / "tiny-llm_amd" / "compat", ROOT / "tiny-llm_amd", ROOT / "tiny-llm_amd" / "extensions_hip"):
    if str(extra) not in sys.path:
        sys.path.insert(0, str(extra))

TL_F32, TL_F16, TL_BF16 = 0, 1, 2
_NAME = {TL_F32: "f32", TL_F16: "f16", TL_BF16: "bf16"}


def _raw(ptr, count, ctype):
    if not ptr or count <= 0:
        return np.zeros((0,), dtype=np.dtype(ctype))
    return np.ctypeslib.as_array((ctype * int(count)).from_address(int(ptr)))


def _load(ptr, count, dt):
    """Host memory of `count` elements of dtype tag `dt` -> float32 array (bf16 / f16 widened exactly)."""
    if dt == TL_F32:
        return _raw(ptr, count, ctypes.c_float).copy()
    bits = _raw(ptr, count, ctypes.c_uint16)
    if dt == TL_BF16:
        return (bits.astype(np.uint32) << 16).view(np.float32)
    return bits.view(np.float16).astype(np.float32)


def _store(ptr, values, dt):
    flat = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
    if dt == TL_F32:
        _raw(ptr, flat.size, ctypes.c_float)[:] = flat
    elif dt == TL_BF16:
        _raw(ptr, flat.size, ctypes.c_uint16)[:] = (flat.view(np.uint32) >> 16).astype(np.uint16)  # values are bf16-exact
    else:
        _raw(ptr, flat.size, ctypes.c_uint16)[:] = flat.astype(np.float16).view(np.uint16)


class FakeLib:
    """The C ABI of include/tinyllm_hip.h over host pointers, answered by the numpy oracle."""

    def __init__(self, real):
        from oracle import tiny_oracle as O

        self.O = O
        self.real = real  # host-only entry points (policies, sizes, error text) come from the real library
        self.calls = []

    def __getattr__(self, name):
        return getattr(self.real, name)  # tl_last_error, tl_abi_version, tl_quantized_matmul_split_k, ...workspace_bytes

    def tl_load_library(self, path):
        return 0

    def tl_quantized_matmul(self, scales, biases, a, b, out, M, N, K, group_size, bits, dt, use_simdgroup, use_split_k, ws,
                            ws_bytes, stream):
        O, name = self.O, _NAME[dt]
        s = _load(scales, K * (N // 128), dt).reshape(K, N // 128)
        z = _load(biases, K * (N // 128), dt).reshape(K, N // 128)
        x = _load(a, M * N, dt).reshape(M, N)
        w = _raw(b, K * (N // 8), ctypes.c_uint32).reshape(K, N // 8)
        self.calls.append(("quantized_matmul", M, N, K, int(use_simdgroup), int(use_split_k)))
        if not use_simdgroup or M <= 8:
            y = O.quantized_matmul(s, z, x, w, name)  # semantic definition / matvec (quantized_matmul.cpp:137-139)
        else:
            split = self.real.tl_quantized_matmul_split_k(M, N, K, 1, int(use_split_k)) if use_split_k else 1
            y = O.quantized_matmul_tile(s, z, x, w, name, split_k=split)
        _store(out, y, dt)
        return 0

    def tl_gather_quantized_matvec(self, scales, biases, a, b, expert_ids, out, M, N, K, num_experts, group_size, bits, dt,
                                   stream):
        E, name = num_experts, _NAME[dt]
        s = _load(scales, E * K * (N // 128), dt).reshape(E, K, N // 128)
        z = _load(biases, E * K * (N // 128), dt).reshape(E, K, N // 128)
        x = _load(a, M * N, dt).reshape(M, N)
        w = _raw(b, E * K * (N // 8), ctypes.c_uint32).reshape(E, K, N // 8)
        ids = np.clip(_raw(expert_ids, M, ctypes.c_int32), 0, E - 1)  # out-of-range ids are clamped (include/tinyllm_hip.h:104-110)
        _store(out, self.O.gather_quantized_matvec(s, z, x, w, ids, name), dt)
        return 0

    def tl_quantized_embedding(self, indices, indices_unsigned, scales, biases, weight, out, tokens, dim, vocab, group_size, bits,
                               dt, stream):
        idx = _raw(indices, tokens, ctypes.c_int32)
        s = _load(scales, vocab * (dim // 128), dt).reshape(vocab, dim // 128)
        z = _load(biases, vocab * (dim // 128), dt).reshape(vocab, dim // 128)
        w = _raw(weight, vocab * (dim // 8), ctypes.c_uint32).reshape(vocab, dim // 8)
        _store(out, self.O.quantized_embedding(idx, s, z, w, _NAME[dt]), dt)
        return 0

    def tl_rms_norm(self, x, weight, out, rows, dim, eps, dt, stream):
        xv = _load(x, rows * dim, dt).reshape(rows, dim)
        _store(out, self.O.rms_norm_fast(xv, _load(weight, dim, dt), eps, _NAME[dt]), dt)
        return 0

    def tl_rope(self, x, offsets, out, B, L, H, D, dims, base, traditional, dt, stream):
        xv = _load(x, B * L * H * D, dt).reshape(B, L, H, D)
        off = _raw(offsets, B, ctypes.c_int32).copy()
        _store(out, self.O.rope(xv, off, dims, base, bool(traditional), _NAME[dt]), dt)
        return 0

    def tl_swiglu(self, gate, up, out, size, dt, stream):
        _store(out, self.O.swiglu(_load(gate, size, dt), _load(up, size, dt), _NAME[dt]), dt)
        return 0

    def tl_decode_attention(self, q, k, v, mask, out, q_rows, L, S, D, num_heads, num_kv_heads, scale, is_causal, has_mask, dt,
                            stream):
        kv_rows = q_rows // num_heads * num_kv_heads
        qv = _load(q, q_rows * L * D, dt).reshape(q_rows, L, D)
        kv = _load(k, kv_rows * S * D, dt).reshape(kv_rows, S, D)
        vv = _load(v, kv_rows * S * D, dt).reshape(kv_rows, S, D)
        m = _raw(mask, q_rows * L * S, ctypes.c_float).reshape(q_rows, L, S).copy() if has_mask else None
        _store(out, self.O.decode_attention(qv, kv, vv, scale, num_heads, num_kv_heads, bool(is_causal), m, _NAME[dt]), dt)
        return 0

    def tl_paged_cache_update(self, pages, values, num_pages, heads, page_size, head_dim, length, page_id, start, dt, stream):
        item = 4 if dt == TL_F32 else 2
        ctype = ctypes.c_float if dt == TL_F32 else ctypes.c_uint16
        pv = _raw(pages, num_pages * heads * page_size * head_dim, ctype).reshape(num_pages, heads, page_size, head_dim)
        vv = _raw(values, heads * length * head_dim, ctype).reshape(heads, length, head_dim)
        pv[page_id, :, start:start + length, :] = vv  # raw element copy, in place (paged_attention.metal:82-106)
        del item
        return 0

    def tl_paged_attention(self, q, key_pages, value_pages, block_table, context_lens, out, N, L, D, num_pages, page_size,
                           max_pages, num_heads, num_kv_heads, scale, is_causal, hint, dt, ws, ws_bytes, stream):
        B = N // num_heads
        qv = _load(q, N * L * D, dt).reshape(N, L, D)
        kp = _load(key_pages, num_pages * num_kv_heads * page_size * D, dt).reshape(num_pages, num_kv_heads, page_size, D)
        vp = _load(value_pages, num_pages * num_kv_heads * page_size * D, dt).reshape(num_pages, num_kv_heads, page_size, D)
        table = _raw(block_table, B * max_pages, ctypes.c_int32).reshape(B, max_pages).copy()
        ctx = _raw(context_lens, B, ctypes.c_int32).copy()
        flash = L > 8 and dt == TL_BF16  # the MFMA FlashAttention branch rounds P to bf16 before PV (paged_attention.metal:439-444)
        y = self.O.paged_attention(qv, kp, vp, table, ctx, scale, bool(is_causal), num_kv_heads, num_heads, _NAME[dt], round_p=flash)
        _store(out, y, dt)
        return 0


def pytest_configure(config):
    import tiny_llm_ext_hip as ext

    ext._require_gpu = lambda op, *tensors: None
    ext._stream = lambda: 0
    ext._workspace = lambda nbytes, device: None
    ext._lib = FakeLib(ext._lib)
    ext.load_library = lambda path: None
    if os.environ.get("REFSOL_REFERENCE_SOURCES") == "1":
        # CONTROL RUN: the reference's OWN tiny_llm_ref sources (not the product's host mirror) on the same stand-ins -- the torch
        # facade for mlx, this oracle-backed binding for its Metal extension.  If the reference's solution passes the reference's
        # tests on these stand-ins, the stand-ins are faithful at the tests' tolerances (tests/test_refsol_facade_cpu.py).
        import types

        pkg = types.ModuleType("extensions_ref")
        pkg.__path__ = []
        pkg.tiny_llm_ext_ref = ext
        sys.modules["extensions_ref"] = pkg
        sys.modules["extensions_ref.tiny_llm_ext_ref"] = ext
        for name in [k for k in sys.modules if k == "tiny_llm_ref" or k.startswith("tiny_llm_ref.")]:
            del sys.modules[name]
        sys.path.insert(0, os.path.join(os.environ.get("TINYLLM_REFERENCE_ROOT", "/root/reference"), "src"))
    if os.environ.get("REFSOL_REFERENCE_BENCHES") == "1":
        # the reference's OWN benches/ package (a namespace package) must win over this repository's benches/ (a regular
        # package): the oracle is already imported, so the repository root can leave the module search path
        sys.modules.pop("benches", None)
        sys.path[:] = [p for p in sys.path if Path(p or ".").resolve() != ROOT]


try:  # the reference's benches/test_attention.py and test_quantized_matmul.py take pytest-benchmark's `benchmark` fixture;
    import pytest_benchmark  # noqa: F401  the plugin is not installed in this image: a one-call stand-in keeps their ASSERTIONS
except ImportError:
    import pytest

    @pytest.fixture
    def benchmark():
        def run_once(function, *args, **kwargs):
            return function(*args, **kwargs)

        return run_once
