"""CPU tier: the numpy oracle against the REFERENCE'S OWN KERNEL CODE.

oracle/_ref/libref_metal_kernels.so is the reference's Metal kernel sources (src/extensions_ref/src/week2_kernels.metal,
quantized_matmul.metal, paged_attention.metal) compiled for the host from where they lie (recipe: oracle/Makefile) against a
Metal-on-CPU shim (oracle/metal_shim: address spaces, half / bfloat storage types, libm for the `fast::` functions, and a fiber
scheduler that gives simd_sum / simd_max / simd_shuffle_xor / threadgroup_barrier their lock-step meaning), launched with the grid
geometry and threadgroup-memory sizes the reference's *.cpp set.  So the thing on the other side of every comparison below is
the reference's kernel arithmetic -- its fp32 accumulation chains, its rounding points, its masks and page walks -- not a
restatement of it.  What the shim cannot reproduce: Apple's approximate `fast::exp2 / sin / cos` (libm stands in) and the
hardware's combination order inside simd_sum (lane order here).  Not covered: the two kernels that need MLX's own "steel"
headers, which are not in the reference tree (32x32 tile GEMM, MMA FlashAttention).

16-bit results: bit equality is expected and asserted up to a tiny share of one-step differences (an fp32 sum landing on the other
side of a rounding boundary: the oracle accumulates in float64).  Built only where /root/reference exists; skipped otherwise.
"""

import numpy as np
import pytest

from oracle import ref_kernels as K
from oracle import tiny_oracle as O

pytestmark = pytest.mark.skipif(not K.available(), reason="oracle/_ref was not built (no /root/reference at build time)")
STEP = {"bf16": 2.0 ** -7, "f16": 2.0 ** -10}


def agree(got, want, dtype, what, exact_share=0.99, f32_tol=2e-6):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, what
    if dtype == "f32":
        np.testing.assert_allclose(got, want, rtol=f32_tol, atol=f32_tol, err_msg=what)
        return
    same = got == want
    step = STEP[dtype] * np.maximum(np.abs(want), 2.0 ** -10)
    assert np.all(np.abs(got - want) <= step), f"{what}: a value is more than one {dtype} step away (max {np.abs(got - want).max():.3e})"
    assert same.mean() >= exact_share, f"{what}: only {same.mean() * 100:.2f}% bit-identical"


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_pointwise_kernels(dtype):
    rng = np.random.default_rng(0)
    for rows, dim in ((3, 256), (2, 1000), (1, 96)):  # dims that are not multiples of the 256-thread stride too
        x = O.cast(rng.standard_normal((rows, dim)).astype(np.float32) * 1.7, dtype)
        w = O.cast(1 + 0.1 * rng.standard_normal(dim).astype(np.float32), dtype)
        agree(K.rms_norm(x, w, 1e-6, dtype), O.rms_norm_fast(x, w, 1e-6, dtype), dtype, f"week2_rms_norm {rows}x{dim}")
    g = O.cast(rng.standard_normal((7, 96)).astype(np.float32) * 3, dtype)
    u = O.cast(rng.standard_normal((7, 96)).astype(np.float32), dtype)
    agree(K.swiglu(g, u, dtype), O.swiglu(g, u, dtype), dtype, "week2_swiglu")
    h = O.cast(rng.standard_normal((2, 3, 6, 64)).astype(np.float32), dtype)  # 6 heads: a partial block of the 4-heads-per-thread loop
    for traditional in (False, True):
        for dims in (64, 32):
            agree(K.rope(h, [0, 117], dims, 1000000.0, traditional, dtype), O.rope(h, np.array([0, 117]), dims, 1000000.0, traditional, dtype),
                  dtype, f"week2_rope traditional={traditional} dims={dims}", exact_share=0.98, f32_tol=2e-5)  # fp32 angle of position 119


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_quantized_kernels(dtype):
    """The semantic definition (one thread per output, quantized_matmul.metal:8-56), the decode matvec the engine replaces
    (quantized_matvec_x4_fast, :441-538: SIMD-group reduction, scale / bias factored per group), the embedding gather and the
    split-K reduction -- each against the oracle function that restates it."""
    rng = np.random.default_rng(1)
    for K_rows, N in ((40, 256), (13, 384), (8, 128)):  # output rows that do not fill the 8-outputs-per-threadgroup tile
        w = O.cast(rng.standard_normal((K_rows, N)).astype(np.float32) * 0.05, dtype)
        packed, scales, biases = O.quantize_affine(w, dtype=dtype)
        for M in (1, 3):
            x = O.cast(rng.standard_normal((M, N)).astype(np.float32), dtype)
            want = O.quantized_matmul(scales, biases, x, packed, dtype)
            agree(K.quantized_matmul_vanilla(scales, biases, x, packed, dtype), want, dtype, f"vanilla matmul {M}x{N}->{K_rows}")
            agree(K.quantized_matvec_x4_fast(scales, biases, x, packed, dtype), want, dtype, f"matvec_x4_fast {M}x{N}->{K_rows}")
        idx = np.array([[3, K_rows - 1], [0, 7]])
        agree(K.quantized_embedding(idx, scales, biases, packed, dtype), O.quantized_embedding(idx, scales, biases, packed, dtype), dtype, "embedding",
              exact_share=1.0)
    parts = O.cast(rng.standard_normal((3, 4, 8)).astype(np.float32), dtype)
    agree(K.splitk_reduce(parts, dtype), O.cast(parts.astype(np.float64).sum(0).astype(np.float32), dtype), dtype, "split-K reduction")
    # and the oracle's split-K tile semantics END in exactly that reduction: partials rounded to T, then summed (metal:251-293)
    w = O.cast(rng.standard_normal((16, 256)).astype(np.float32) * 0.05, dtype)
    packed, scales, biases = O.quantize_affine(w, dtype=dtype)
    x = O.cast(rng.standard_normal((12, 256)).astype(np.float32), dtype)
    dense = O.dequantize_weights(packed, scales, biases, dtype=dtype).astype(np.float64)
    partials = np.stack([O.cast((x[:, s:s + 128].astype(np.float64) @ dense[:, s:s + 128].T).astype(np.float32), dtype) for s in (0, 128)])
    agree(K.splitk_reduce(partials, dtype), O.quantized_matmul_tile(scales, biases, x, packed, dtype, split_k=2), dtype, "split-K tile matmul")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_dense_decode_attention_kernel(dtype):
    rng = np.random.default_rng(2)
    Hq, Hkv, D, S, L = 4, 2, 64, 37, 2
    q = O.cast(rng.standard_normal((2 * Hq, L, D)).astype(np.float32), dtype)
    k = O.cast(rng.standard_normal((2 * Hkv, S, D)).astype(np.float32), dtype)
    v = O.cast(rng.standard_normal((2 * Hkv, S, D)).astype(np.float32), dtype)
    scale = D ** -0.5
    for causal, mask in ((True, None), (False, None), (False, np.where(rng.random((2 * Hq, L, S)) < 0.2, -2.0, 0.0).astype(np.float32))):
        agree(K.decode_attention(q, k, v, scale, Hq, Hkv, causal, mask, dtype), O.decode_attention(q, k, v, scale, Hq, Hkv, causal, mask, dtype), dtype,
              f"week2_decode_attention causal={causal} mask={'yes' if mask is not None else 'no'}")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_paged_kernels(dtype):
    rng = np.random.default_rng(3)
    Hq, Hkv, D, P, page = 4, 2, 64, 7, 8
    kp = O.cast(rng.standard_normal((P, Hkv, page, D)).astype(np.float32), dtype)
    vp = O.cast(rng.standard_normal((P, Hkv, page, D)).astype(np.float32), dtype)
    # scattered pages, a context that ends inside a page, an idle row (context 0, all -1) and L = 3 query positions
    table = np.array([[5, 1, 3, -1], [-1, -1, -1, -1], [0, 2, 6, 4]], dtype=np.int32)
    ctx = np.array([21, 0, 32], dtype=np.int32)
    for L, causal in ((1, True), (3, True), (2, False)):
        q = O.cast(rng.standard_normal((3 * Hq, L, D)).astype(np.float32), dtype)
        agree(K.paged_attention_decode(q, kp, vp, table, ctx, D ** -0.5, causal, Hkv, Hq, dtype), O.paged_attention(q, kp, vp, table, ctx, D ** -0.5, causal, Hkv, Hq, dtype),
              dtype, f"paged_attention_decode L={L} causal={causal}")
    values = O.cast(rng.standard_normal((1, Hkv, 3, D)).astype(np.float32), dtype)
    want = kp.copy()
    want[2, :, 4:7] = values[0]
    agree(K.paged_cache_update(kp, values, 2, 4, dtype), want, dtype, "paged_cache_update", exact_share=1.0)
    assert np.array_equal(want, O.paged_cache_update(kp.copy(), values, 2, 4))


def test_paged_decode_kernel_head_dim_128_and_the_fp32_prefill_kernel():
    rng = np.random.default_rng(4)
    Hq, Hkv, D, P, page = 2, 1, 128, 4, 16
    kp = O.bf16(rng.standard_normal((P, Hkv, page, D)).astype(np.float32))
    vp = O.bf16(rng.standard_normal((P, Hkv, page, D)).astype(np.float32))
    table = np.array([[2, 0, 3, -1]], dtype=np.int32)
    ctx = np.array([41], dtype=np.int32)
    q = O.bf16(rng.standard_normal((Hq, 1, D)).astype(np.float32))
    want = O.paged_attention(q, kp, vp, table, ctx, D ** -0.5, True, Hkv, Hq, "bf16")
    agree(K.paged_attention_decode(q, kp, vp, table, ctx, D ** -0.5, True, Hkv, Hq, "bf16", fixed_d128=True), want, "bf16", "paged decode, the _d128 instantiation")
    agree(K.paged_attention_decode(q, kp, vp, table, ctx, D ** -0.5, True, Hkv, Hq, "bf16", fixed_d128=False), want, "bf16", "paged decode, generic head dim")
    # L > 8 in fp32: the scalar tile kernel (paged_attention.metal:508-674), 20 query rows = one full and one partial 16-row block
    Hq, Hkv, D, page = 4, 2, 64, 8
    kp32 = rng.standard_normal((6, Hkv, page, D)).astype(np.float32)
    vp32 = rng.standard_normal((6, Hkv, page, D)).astype(np.float32)
    table = np.array([[5, 1, 3, -1], [0, 2, 4, -1]], dtype=np.int32)
    ctx = np.array([24, 20], dtype=np.int32)
    q32 = rng.standard_normal((2 * Hq, 20, D)).astype(np.float32)
    agree(K.paged_attention_scalar_f32(q32, kp32, vp32, table, ctx, D ** -0.5, True, Hkv, Hq), O.paged_attention(q32, kp32, vp32, table, ctx, D ** -0.5, True, Hkv, Hq, "f32"),
          "f32", "paged_attention_scalar_f32")
