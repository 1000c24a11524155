"""GPU parity: every HIP operator behind the C ABI vs the numpy oracle on the same seeded inputs.

Mirrors the reference's per-day kernel tests (tests_refsol/test_week_2_day_3..7.py, test_week_3_day_3..5.py):
boundary sweeps for shapes, dtypes, GQA ratios, masks, non-contiguous pages, and the validation errors.
Tolerances are PER ELEMENT (helpers.assert_rounded_close): one ulp of the oracle's rounded value (2^-8 relative for bf16,
2^-11 for f16) for the single rounding, plus a stated floor for fp32 accumulation in another order -- 2^-20 of the ABSOLUTE sum
of the terms of a dot product, 2^-18 |v|max for an attention average, 2^-9 |v|max where the reference rounds P to bf16.
"""

import numpy as np
import pytest
import torch

from helpers import assert_rounded_close, w4_abs_dot
from oracle import tiny_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"
TORCH = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
# (rtol, atol) for values of magnitude ~1 produced by a different fp32 accumulation order
TOL = {"bf16": (1.6e-2, 1e-2), "f16": (2e-3, 2e-3), "f32": (1e-5, 1e-5)}


@pytest.fixture(scope="module")
def ext():
    import tiny_llm_ext_hip

    tiny_llm_ext_hip.load_library(".")
    return tiny_llm_ext_hip


def dev(a: np.ndarray, dtype: str) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV, TORCH[dtype])


def host(t: torch.Tensor) -> np.ndarray:
    return t.float().cpu().numpy()


def packed_dev(p: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(p).view(np.int32)).to(DEV)


def make_w4(rng, K, N, dtype, sigma=0.05):
    w = O.cast(rng.standard_normal((K, N), dtype=np.float32) * sigma, dtype)
    return O.quantize_affine(w, dtype=dtype)


# fp32 accumulation in another order than the oracle's float64: 16 fp32 steps of the ABSOLUTE sum of the terms (helpers.py)
ACC_FLOOR = 2.0 ** -20


def close_matmul(got, want, dtype, a, packed, scales, biases, what, extra_floor=0.0):
    """One ulp of the oracle's rounded value per element + the fp32-accumulation floor derived from sum |a_k w_k|."""
    assert_rounded_close(got, want, dtype, ulps=1.0, floor=ACC_FLOOR * w4_abs_dot(a, packed, scales, biases, dtype) + extra_floor, what=what)


# ----------------------------------------------------------------------------- quantized matmul
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("M,K,N", [(1, 8, 128), (1, 512, 1024), (1, 1024, 2560), (1, 2560, 4096), (1, 515, 640),
                                   (2, 300, 256), (3, 64, 384), (5, 1024, 1024), (8, 37, 128), (8, 2048, 2560)])
def test_quantized_matvec(ext, dtype, M, K, N):
    """decode GEMV (use_simdgroup=True, M<=8) at ragged and Qwen-like shapes; reference test_week_2_day_3.py."""
    rng = np.random.default_rng(M * 1000 + K + N)
    packed, scales, biases = make_w4(rng, K, N, dtype)
    a = O.cast(rng.standard_normal((M, N), dtype=np.float32), dtype)
    want = O.quantized_matmul(scales, biases, a, packed, dtype)
    got = ext.quantized_matmul(dev(scales, dtype), dev(biases, dtype), 128, 4, dev(a, dtype), packed_dev(packed), True)
    assert got.shape == (M, K) and got.dtype == TORCH[dtype]
    close_matmul(host(got), want, dtype, a, packed, scales, biases, f"matvec {dtype} M={M} K={K} N={N}")


@pytest.mark.parametrize("M,K,N", [(1, 9728, 2560), (1, 2560, 9728), (4, 2560, 9728), (8, 2560, 9728)])
def test_quantized_matvec_qwen4b_shapes(ext, M, K, N):
    """The real Qwen3-4B MLP projection shapes, incl. the M=8/N=9728 case that splits M to fit LDS."""
    rng = np.random.default_rng(7)
    packed, scales, biases = make_w4(rng, K, N, "bf16", sigma=0.02)
    a = O.bf16(rng.standard_normal((M, N), dtype=np.float32))
    want = O.quantized_matmul(scales, biases, a, packed, "bf16")
    got = ext.quantized_matmul(dev(scales, "bf16"), dev(biases, "bf16"), 128, 4, dev(a, "bf16"), packed_dev(packed), True)
    close_matmul(host(got), want, "bf16", a, packed, scales, biases, f"matvec Qwen3-4B shape M={M} K={K} N={N}")


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("M,K,N", [(1, 64, 128), (7, 130, 256), (33, 96, 384)])
def test_quantized_matmul_vanilla(ext, dtype, M, K, N):
    """use_simdgroup=False: the one-thread-per-output semantic definition (quantized_matmul.metal:8-56)."""
    rng = np.random.default_rng(11 + M)
    packed, scales, biases = make_w4(rng, K, N, dtype)
    a = O.cast(rng.standard_normal((M, N), dtype=np.float32), dtype)
    want = O.quantized_matmul(scales, biases, a, packed, dtype)
    got = ext.quantized_matmul(dev(scales, dtype), dev(biases, dtype), 128, 4, dev(a, dtype), packed_dev(packed), True,
                               use_simdgroup=False)
    close_matmul(host(got), want, dtype, a, packed, scales, biases, f"vanilla matmul {dtype} M={M} K={K} N={N}")


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("M,K,N", [(9, 32, 128), (32, 128, 256), (33, 200, 384), (64, 1024, 2560), (130, 257, 1024),
                                   (256, 512, 512)])
def test_quantized_matmul_mfma(ext, dtype, M, K, N):
    """prefill tile GEMM on MFMA (weights rounded to T first); reference test_week_2_day_6.py:30-70."""
    rng = np.random.default_rng(M + K + N)
    packed, scales, biases = make_w4(rng, K, N, dtype)
    a = O.cast(rng.standard_normal((M, N), dtype=np.float32), dtype)
    want = O.quantized_matmul_tile(scales, biases, a, packed, dtype)
    got = ext.quantized_matmul(dev(scales, dtype), dev(biases, dtype), 128, 4, dev(a, dtype), packed_dev(packed), True,
                               use_simdgroup=True, use_split_k=False)
    # the tile kernel rounds the dequantised weights to T first (as the reference's does): the oracle restates that, so the same
    # one-ulp + accumulation-floor allowance applies
    close_matmul(host(got), want, dtype, a, packed, scales, biases, f"tile GEMM {dtype} M={M} K={K} N={N}")


@pytest.mark.parametrize("M,K,N", [(32, 128, 2048), (16, 1024, 4096), (64, 256, 9728)])
def test_quantized_matmul_split_k(ext, M, K, N):
    """split-K: partials stored in bf16 then reduced in fp32 (reference test_week_2_day_7.py:19-77)."""
    import ctypes

    rng = np.random.default_rng(3)
    packed, scales, biases = make_w4(rng, K, N, "bf16")
    a = O.bf16(rng.standard_normal((M, N), dtype=np.float32))
    split = ext.lib().tl_quantized_matmul_split_k(M, N, K, 1, 1)
    assert split > 1, "shape chosen so that the policy splits"
    want = O.quantized_matmul_tile(scales, biases, a, packed, "bf16", split_k=split)
    got = ext.quantized_matmul(dev(scales, "bf16"), dev(biases, "bf16"), 128, 4, dev(a, "bf16"), packed_dev(packed),
                               True, use_simdgroup=True, use_split_k=True)
    # every slice's partial is rounded to bf16 before the reduction (reference arithmetic): the kernel's fp32 partial may round
    # the other way than the oracle's float64 one -- half a bf16 step of each partial, bounded by 2^-8 of the absolute sum
    absdot = w4_abs_dot(a, packed, scales, biases, "bf16")
    close_matmul(host(got), want, "bf16", a, packed, scales, biases, f"split-K M={M} K={K} N={N} split={split}", extra_floor=2.0 ** -8 * absdot / 4)


def test_split_k_falls_back_bit_exact(ext):
    """When the policy yields split_k == 1 the result must be bit-identical to the unsplit kernel
    (reference test_week_2_day_7.py:80-109)."""
    rng = np.random.default_rng(5)
    M, K, N = 512, 2048, 128  # N/128 == 1 -> cannot split
    assert ext.lib().tl_quantized_matmul_split_k(M, N, K, 1, 1) == 1
    packed, scales, biases = make_w4(rng, K, N, "bf16")
    a = dev(O.bf16(rng.standard_normal((M, N), dtype=np.float32)), "bf16")
    args = (dev(scales, "bf16"), dev(biases, "bf16"), 128, 4, a, packed_dev(packed), True)
    plain = ext.quantized_matmul(*args, use_simdgroup=True, use_split_k=False)
    split = ext.quantized_matmul(*args, use_simdgroup=True, use_split_k=True)
    assert torch.equal(plain, split)


def test_quantized_matmul_validation(ext):
    """C++-level preconditions surface as RuntimeError (quantized_matmul.cpp:24-72)."""
    rng = np.random.default_rng(0)
    packed, scales, biases = make_w4(rng, 16, 128, "bf16")
    a = dev(O.bf16(rng.standard_normal((1, 128), dtype=np.float32)), "bf16")
    s, b, p = dev(scales, "bf16"), dev(biases, "bf16"), packed_dev(packed)
    with pytest.raises(RuntimeError, match="b must be transposed"):
        ext.quantized_matmul(s, b, 128, 4, a, p, False)
    with pytest.raises(RuntimeError, match="bits must be 4"):
        ext.quantized_matmul(s, b, 128, 8, a, p, True)
    with pytest.raises(RuntimeError, match="group_size must be 128"):
        ext.quantized_matmul(s, b, 64, 4, a, p, True)
    with pytest.raises(RuntimeError, match="same dtype as scales"):
        ext.quantized_matmul(s, b, 128, 4, a.to(torch.float16), p, True)
    with pytest.raises(RuntimeError, match="GPU-only"):
        ext.quantized_matmul(s.cpu(), b.cpu(), 128, 4, a.cpu(), p.cpu(), True)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_quantized_embedding(ext, dtype):
    rng = np.random.default_rng(2)
    V, dim = 300, 384
    packed, scales, biases = make_w4(rng, V, dim, dtype)
    idx = rng.integers(0, V, size=(2, 7)).astype(np.int32)
    want = O.quantized_embedding(idx, scales, biases, packed, dtype)
    got = ext.quantized_embedding(torch.from_numpy(idx).to(DEV), dev(scales, dtype), dev(biases, dtype),
                                  packed_dev(packed), 128, 4)
    assert got.shape == (2, 7, dim)
    np.testing.assert_array_equal(host(got), want)  # pure dequant: bit-exact


# ----------------------------------------------------------------------------- pointwise
@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("shape", [(1, 1, 2560), (3, 5, 128), (2, 4, 8, 128), (7, 100), (2, 3, 4104), (5, 64)])
def test_rms_norm(ext, dtype, shape):
    """single-rounding RMSNorm; reference test_week_2_day_4.py:29-34."""
    rng = np.random.default_rng(sum(shape))
    x = O.cast(rng.standard_normal(shape, dtype=np.float32) * 2, dtype)
    w = O.cast(1 + 0.1 * rng.standard_normal(shape[-1:], dtype=np.float32), dtype)
    want = O.rms_norm_fast(x, w, 1e-6, dtype)
    got = ext.rms_norm(dev(x, dtype), dev(w, dtype), 1e-6)
    assert_rounded_close(host(got), want, dtype, ulps=1.0 if dtype != "f32" else 8.0, what=f"rms_norm {dtype} {shape}")


@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("traditional", [False, True])
@pytest.mark.parametrize("B,L,H,D,dims", [(1, 1, 32, 128, 128), (2, 9, 5, 64, 64), (3, 4, 2, 96, 64), (1, 33, 8, 128, 128)])
def test_rope(ext, dtype, traditional, B, L, H, D, dims):
    """per-row offsets, partial rotary dims, both pairings; reference test_week_2_day_4.py / test_week_3_day_1.py:30-37."""
    rng = np.random.default_rng(B * 100 + L)
    x = O.cast(rng.standard_normal((B, L, H, D), dtype=np.float32), dtype)
    offsets = rng.integers(0, 3000, size=(B,)).astype(np.int32)
    want = O.rope(x, offsets, dims, 1000000.0, traditional, dtype)
    got = ext.rope(dev(x, dtype), torch.from_numpy(offsets).to(DEV), dims, 1000000.0, traditional)
    # Oracle and kernel evaluate the same fp32 expression; they may differ by 2 ulp in base^(-d/half) (exp2f vs numpy), half
    # an ulp in the product and one in the sincos argument reduction: <= 4 * 2^-24 relative on an angle of up to ~3e3 rad,
    # i.e. ~7e-4 rad, times |x| sqrt(2) on the rotated pair.  16-bit outputs add their own rounding (TOL).
    rtol, atol = TOL[dtype]
    angle_max = float(offsets.max() + L)
    angle_term = float(np.abs(x).max()) * np.sqrt(2.0) * angle_max * 4.0 * 2.0 ** -24
    np.testing.assert_allclose(host(got), want, rtol=rtol, atol=(atol if dtype != "f32" else 1e-6) + angle_term)


@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("shape", [(1, 1, 9728), (3, 7, 33), (2, 1024)])
def test_swiglu(ext, dtype, shape):
    rng = np.random.default_rng(1)
    g = O.cast(rng.standard_normal(shape, dtype=np.float32) * 3, dtype)
    u = O.cast(rng.standard_normal(shape, dtype=np.float32), dtype)
    want = O.swiglu(g, u, dtype)
    got = ext.swiglu(dev(g, dtype), dev(u, dtype))
    assert_rounded_close(host(got), want, dtype, ulps=1.0 if dtype != "f32" else 8.0, what=f"swiglu {dtype} {shape}")


# ----------------------------------------------------------------------------- dense decode attention
def sin_fixture(shape, phase):
    """deterministic sin ramp used by the reference sweep (test_week_2_day_5.py:127-143)."""
    n = int(np.prod(shape))
    return np.sin(np.arange(n, dtype=np.float32) * 0.017 + phase).reshape(shape)


@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("L", [1, 8])
@pytest.mark.parametrize("S", [1, 31, 32, 127, 128, 129, 255, 256])
@pytest.mark.parametrize("rep", [1, 4])
@pytest.mark.parametrize("mask_kind", ["none", "causal", "explicit"])
def test_decode_attention_sweep(ext, dtype, L, S, rep, mask_kind):
    if S < L:
        pytest.skip("context shorter than the query block")
    B, Hkv, D = 2, 2, 64
    Hq = Hkv * rep
    q = O.cast(sin_fixture((B * Hq, L, D), 0.1), dtype)
    k = O.cast(sin_fixture((B * Hkv, S, D), 0.7), dtype)
    v = O.cast(sin_fixture((B * Hkv, S, D), 1.3), dtype)
    mask = None
    if mask_kind == "explicit":
        mask = np.where(np.arange(B * Hq * L * S).reshape(B * Hq, L, S) % 5 == 0, -2.0, 0.0).astype(np.float32)
    want = O.decode_attention(q, k, v, D ** -0.5, Hq, Hkv, is_causal=(mask_kind == "causal"), mask=mask, dtype=dtype)
    m_arg = torch.from_numpy(mask).to(DEV) if mask is not None else torch.zeros(1, device=DEV)
    got = ext.decode_attention(dev(q, dtype), dev(k, dtype), dev(v, dtype), m_arg, D ** -0.5,
                               mask_kind == "causal", mask is not None, Hq, Hkv)
    assert_rounded_close(host(got), want, dtype, ulps=1.0 if dtype != "f32" else 16.0, floor=ACC_FLOOR * 4 * float(np.abs(v).max()), what=f"decode attention {dtype} L={L} S={S} rep={rep} mask={mask_kind}")


def test_decode_attention_d256(ext):
    rng = np.random.default_rng(9)
    q = O.bf16(rng.standard_normal((4, 2, 256), dtype=np.float32))
    k = O.bf16(rng.standard_normal((2, 70, 256), dtype=np.float32))
    v = O.bf16(rng.standard_normal((2, 70, 256), dtype=np.float32))
    want = O.decode_attention(q, k, v, 256 ** -0.5, 2, 1, is_causal=True)
    got = ext.decode_attention(dev(q, "bf16"), dev(k, "bf16"), dev(v, "bf16"), torch.zeros(1, device=DEV),
                               256 ** -0.5, True, False, 2, 1)
    assert_rounded_close(host(got), want, "bf16", ulps=1.0, floor=ACC_FLOOR * 4 * float(np.abs(v).max()), what="decode attention, head_dim 256")


# ----------------------------------------------------------------------------- paged KV
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_paged_cache_update_in_place(ext, dtype):
    """writes only the slice, returns the SAME storage (paged_attention.cpp:46-49)."""
    rng = np.random.default_rng(4)
    P, H, page, D = 5, 3, 8, 64
    pages = O.cast(rng.standard_normal((P, H, page, D), dtype=np.float32), dtype)
    vals = O.cast(rng.standard_normal((1, H, 3, D), dtype=np.float32), dtype)
    t_pages = dev(pages, dtype)
    ret = ext.paged_cache_update(t_pages, dev(vals, dtype), 2, 4)
    assert ret.data_ptr() == t_pages.data_ptr()
    want = O.paged_cache_update(pages.copy(), vals, 2, 4)
    np.testing.assert_array_equal(host(t_pages), want)
    with pytest.raises(RuntimeError, match="outside page storage"):
        ext.paged_cache_update(t_pages, dev(vals, dtype), 2, 6)
    with pytest.raises(RuntimeError, match="outside page storage"):
        ext.paged_cache_update(t_pages, dev(vals, dtype), 5, 0)


def scattered_pages(rng, B, Hkv, page, D, ctxs, dtype, holes=True):
    """Non-contiguous physical pages per request (a 'blocker' steals ids in between, test_week_3_day_5.py:25-37)."""
    need = [(c + page - 1) // page for c in ctxs]
    P = sum(need) + 3
    ids = list(rng.permutation(P))
    max_pages = max(max(need), 1) + (1 if holes else 0)
    table = -np.ones((B, max_pages), dtype=np.int32)
    for b in range(B):
        for j in range(need[b]):
            table[b, j] = ids.pop()
    kp = O.cast(rng.standard_normal((P, Hkv, page, D), dtype=np.float32), dtype)
    vp = O.cast(rng.standard_normal((P, Hkv, page, D), dtype=np.float32), dtype)
    return kp, vp, table, np.asarray(ctxs, dtype=np.int32)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("L", [1, 3, 8])
@pytest.mark.parametrize("D,page", [(128, 16), (64, 4), (32, 128), (128, 128)])
@pytest.mark.parametrize("rep", [1, 4])
def test_paged_attention_decode(ext, dtype, L, D, page, rep):
    """L<=8 split-context kernel, causal and not, batch with an idle (ctx=0) row; test_week_3_day_4.py:118-245."""
    rng = np.random.default_rng(L * 10 + D + rep)
    Hkv = 2
    Hq = Hkv * rep
    ctxs = [37, 0, 200 if page >= 16 else 21]
    B = len(ctxs)
    kp, vp, table, ctx = scattered_pages(rng, B, Hkv, page, D, ctxs, dtype)
    q = O.cast(rng.standard_normal((B * Hq, L, D), dtype=np.float32), dtype)
    for causal in (True, False):
        want = O.paged_attention(q, kp, vp, table, ctx, D ** -0.5, causal, Hkv, Hq, dtype)
        got = ext.paged_attention(dev(q, dtype), dev(kp, dtype), dev(vp, dtype), torch.from_numpy(table).to(DEV),
                                  torch.from_numpy(ctx).to(DEV), D ** -0.5, causal, num_kv_heads=Hkv, num_heads=Hq)
        assert_rounded_close(host(got), want, dtype, ulps=1.0 if dtype != "f32" else 16.0, floor=ACC_FLOOR * 4 * float(np.abs(vp).max()), what=f"paged attention {dtype} L={L} D={D} page={page}")
        assert not host(got)[Hq:2 * Hq].any(), "idle row (context 0) must produce zeros"


@pytest.mark.parametrize("ctx_len", [128, 1024, 5000])
def test_paged_attention_decode_long_context_splits(ext, ctx_len):
    """bench_week3_attention shape (B1,Hq32,Hkv8,L1,D128,page128): exercises n_splits>1 + merge kernel."""
    rng = np.random.default_rng(ctx_len)
    kp, vp, table, ctx = scattered_pages(rng, 1, 8, 128, 128, [ctx_len], "bf16")
    q = O.bf16(rng.standard_normal((32, 1, 128), dtype=np.float32))
    want = O.paged_attention(q, kp, vp, table, ctx, 128 ** -0.5, True, 8, 32)
    got = ext.paged_attention(dev(q, "bf16"), dev(kp, "bf16"), dev(vp, "bf16"), torch.from_numpy(table).to(DEV),
                              torch.from_numpy(ctx).to(DEV), 128 ** -0.5, True, num_kv_heads=8, num_heads=32,
                              max_context_hint=ctx_len)
    assert_rounded_close(host(got), want, "bf16", ulps=1.0, floor=ACC_FLOOR * 4 * float(np.abs(vp).max()), what=f"paged decode attention, context {ctx_len}")


@pytest.mark.parametrize("L,ctxs", [(9, [9, 40]), (65, [65, 130]), (33, [100, 33]), (128, [128, 300]), (200, [456, 200])])
@pytest.mark.parametrize("rep", [1, 2, 4])
@pytest.mark.parametrize("page", [16, 128])
def test_paged_attention_prefill_mfma(ext, L, ctxs, rep, page):
    """L>8 bf16 D=128: FlashAttention on MFMA, chunked-prefill contexts (ctx >= L), scattered pages;
    reference test_week_3_day_5.py:23-61 (L in {9, 65})."""
    rng = np.random.default_rng(L + rep)
    Hkv, D = 2, 128
    Hq = Hkv * rep
    B = len(ctxs)
    kp, vp, table, ctx = scattered_pages(rng, B, Hkv, page, D, ctxs, "bf16")
    q = O.bf16(rng.standard_normal((B * Hq, L, D), dtype=np.float32))
    for causal in (True, False):
        want = O.paged_attention(q, kp, vp, table, ctx, D ** -0.5, causal, Hkv, Hq, "bf16", round_p=True)
        got = ext.paged_attention(dev(q, "bf16"), dev(kp, "bf16"), dev(vp, "bf16"), torch.from_numpy(table).to(DEV),
                                  torch.from_numpy(ctx).to(DEV), D ** -0.5, causal, num_kv_heads=Hkv, num_heads=Hq)
        assert_rounded_close(host(got), want, "bf16", ulps=1.0, floor=2.0 ** -9 * float(np.abs(vp).max()), what="paged FlashAttention (P rounded to bf16: one weight step = 2^-9 |v|)")


@pytest.mark.parametrize("L,ctxs", [(64, [2048]), (40, [1500, 700]), (128, [4096])])
def test_paged_attention_prefill_context_splits(ext, L, ctxs):
    """A late chunk of a chunked prefill: few query rows against a long cached context.  The MFMA kernel then cuts the
    context into several workgroups per (head, query block) and paged_merge_kernel combines the partials
    (attention.hip: pick_fa_splits); causal and non-causal, two sequences of different length."""
    rng = np.random.default_rng(L + ctxs[0])
    Hkv, rep, D, page = 2, 4, 128, 128
    Hq = Hkv * rep
    B = len(ctxs)
    kp, vp, table, ctx = scattered_pages(rng, B, Hkv, page, D, ctxs, "bf16")
    q = O.bf16(rng.standard_normal((B * Hq, L, D), dtype=np.float32))
    for causal in (True, False):
        want = O.paged_attention(q, kp, vp, table, ctx, D ** -0.5, causal, Hkv, Hq, "bf16", round_p=True)
        got = ext.paged_attention(dev(q, "bf16"), dev(kp, "bf16"), dev(vp, "bf16"), torch.from_numpy(table).to(DEV),
                                  torch.from_numpy(ctx).to(DEV), D ** -0.5, causal, num_kv_heads=Hkv, num_heads=Hq,
                                  max_context_hint=max(ctxs))
        assert_rounded_close(host(got), want, "bf16", ulps=1.0, floor=2.0 ** -9 * float(np.abs(vp).max()), what="paged FlashAttention (P rounded to bf16: one weight step = 2^-9 |v|)")


def test_paged_attention_prefill_f32_fallback(ext):
    rng = np.random.default_rng(8)
    kp, vp, table, ctx = scattered_pages(rng, 1, 2, 8, 64, [30], "f32")
    q = rng.standard_normal((4, 12, 64), dtype=np.float32)
    want = O.paged_attention(q, kp, vp, table, ctx, 0.125, True, 2, 4, "f32")
    got = ext.paged_attention(dev(q, "f32"), dev(kp, "f32"), dev(vp, "f32"), torch.from_numpy(table).to(DEV),
                              torch.from_numpy(ctx).to(DEV), 0.125, True, num_kv_heads=2, num_heads=4)
    np.testing.assert_allclose(host(got), want, rtol=1e-4, atol=1e-5)


def test_paged_attention_validation(ext):
    rng = np.random.default_rng(0)
    kp, vp, table, ctx = scattered_pages(rng, 1, 2, 8, 64, [10], "bf16")
    q = dev(O.bf16(rng.standard_normal((4, 12, 64), dtype=np.float32)), "bf16")
    with pytest.raises(RuntimeError, match="prefill requires head dimension 128"):
        ext.paged_attention(q, dev(kp, "bf16"), dev(vp, "bf16"), torch.from_numpy(table).to(DEV),
                            torch.from_numpy(ctx).to(DEV), 1.0, True, num_kv_heads=2, num_heads=4)
    with pytest.raises(RuntimeError, match="must be int32"):
        ext.paged_attention(q[:, :1], dev(kp, "bf16"), dev(vp, "bf16"), torch.from_numpy(table).to(DEV).long(),
                            torch.from_numpy(ctx).to(DEV), 1.0, True, num_kv_heads=2, num_heads=4)


# ---------------------------------------------------------------------------------------------------------------------
# grouped-expert W4 product and the MoE block (reference moe.py:7-89, test_week_3_day_6.py)
# ---------------------------------------------------------------------------------------------------------------------
def _expert_stack(rng, E, K, N, dtype, sigma=0.05):
    packed, scales, biases = [], [], []
    for _ in range(E):
        p, s, b = O.quantize_affine(rng.standard_normal((K, N)).astype(np.float32) * sigma, dtype=dtype)
        packed.append(p), scales.append(s), biases.append(b)
    return np.stack(packed), np.stack(scales), np.stack(biases)


def _qw(triple, dtype):
    from tiny_llm_hip import QuantizedWeights

    packed, scales, biases = triple
    return QuantizedWeights(dev(scales, dtype), dev(biases, dtype), 128, 4,
                            torch.from_numpy(np.ascontiguousarray(packed).view(np.int32)).to(DEV))


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("M,E,K,N", [(1, 4, 64, 128), (5, 8, 200, 256), (16, 8, 256, 512), (40, 3, 96, 1024)])
def test_gather_quantized_matvec(ext, dtype, M, E, K, N):
    """One GEMV launch, an expert index per activation row (mx.gather_qmm as grouped_expert_linear uses it); repeated and
    out-of-range ids (clamped) included."""
    rng = np.random.default_rng(M * 100 + E)
    packed, scales, biases = _expert_stack(rng, E, K, N, dtype)
    a = O.cast(rng.standard_normal((M, N)).astype(np.float32), dtype)
    ids = rng.integers(0, E, size=M).astype(np.int32)
    want = O.gather_quantized_matvec(scales, biases, a, packed, ids, dtype)
    got = ext.gather_quantized_matvec(dev(scales, dtype), dev(biases, dtype), 128, 4, dev(a, dtype),
                                      torch.from_numpy(packed.view(np.int32)).to(DEV), torch.from_numpy(ids).to(DEV))
    floor = np.stack([ACC_FLOOR * w4_abs_dot(a[m:m + 1], packed[ids[m]], scales[ids[m]], biases[ids[m]], dtype)[0] for m in range(M)])
    assert_rounded_close(host(got), want, dtype, ulps=1.0, floor=floor, what=f"gather matvec {dtype} M={M} E={E}")
    wild = ids.copy()
    wild[0] = E + 5
    clamped = ext.gather_quantized_matvec(dev(scales, dtype), dev(biases, dtype), 128, 4, dev(a, dtype),
                                          torch.from_numpy(packed.view(np.int32)).to(DEV), torch.from_numpy(wild).to(DEV))
    want0 = O.gather_quantized_matvec(scales, biases, a[:1], packed, np.array([E - 1]), dtype)
    assert_rounded_close(host(clamped)[:1], want0, dtype, ulps=1.0, floor=ACC_FLOOR * w4_abs_dot(a[:1], packed[E - 1], scales[E - 1], biases[E - 1], dtype), what="gather matvec, clamped expert id")
    with pytest.raises(RuntimeError, match="one entry per row"):
        ext.gather_quantized_matvec(dev(scales, dtype), dev(biases, dtype), 128, 4, dev(a, dtype),
                                    torch.from_numpy(packed.view(np.int32)).to(DEV), torch.from_numpy(ids[:-1].copy()).to(DEV) if M > 1 else torch.zeros(2, dtype=torch.int32, device=DEV))


@pytest.mark.parametrize("norm_topk", [False, True])
def test_moe_block_matches_oracle(ext, norm_topk):
    """Router + top-k + SwiGLU experts + weighted sum (Qwen3-MoE MLP) against the numpy restatement; the oracle is given
    the selection the device made, so a tie between two experts' bf16 probabilities cannot fail the comparison."""
    from tiny_llm_hip import Moe, route_topk

    rng = np.random.default_rng(7 + norm_topk)
    E, D, H, k, B, L = 8, 256, 384, 2, 2, 3
    router = O.quantize_affine(rng.standard_normal((E, D)).astype(np.float32) * 0.3, dtype="bf16")
    gate, up = _expert_stack(rng, E, H, D, "bf16"), _expert_stack(rng, E, H, D, "bf16")
    down = _expert_stack(rng, E, D, H, "bf16")
    x = O.bf16(rng.standard_normal((B * L, D)).astype(np.float32))
    moe = Moe(_qw(router, "bf16"), _qw(gate, "bf16"), _qw(up, "bf16"), _qw(down, "bf16"), k, norm_topk_prob=norm_topk)
    xt = dev(x, "bf16").reshape(B, L, D)
    probs, ids, scores = route_topk(xt, moe.w_router, k, norm_topk)
    got = moe(xt)
    ids_np = ids.reshape(-1, k).cpu().numpy()
    want, _, want_scores = O.moe_block(x, router, gate, up, down, k, norm_topk, "bf16", ids=ids_np)
    # the selection is a valid top-k of the oracle's router probabilities
    logits = O.quantized_matmul(router[1], router[2], x, router[0])
    ref_probs = O.softmax(logits.astype(np.float32))
    kth = np.sort(ref_probs, axis=-1)[:, -k]
    assert np.all(np.take_along_axis(ref_probs, ids_np, axis=-1) >= kth[:, None] - 4e-3)
    np.testing.assert_allclose(host(scores).reshape(-1, k), want_scores, atol=8e-3, rtol=2e-2)
    np.testing.assert_allclose(host(got).reshape(-1, D), want, atol=3e-2, rtol=3e-2)
    assert abs(float(host(probs).reshape(-1, E).sum(axis=-1).mean()) - 1.0) < 2e-2
