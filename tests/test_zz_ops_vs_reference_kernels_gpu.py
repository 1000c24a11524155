"""GPU tier (collected last: written after the last GPU run of round 2): every HIP operator behind the C ABI against the
REFERENCE'S OWN KERNEL it replaces, executed on the box's host cores from oracle/_ref/libref_metal_kernels.so (the reference's
`.metal` sources compiled against the Metal-on-CPU shim, oracle/Makefile; the library is built in the build container and travels
with the repository).  No oracle in between: HIP kernel on the device vs the reference's kernel code on the same inputs.
Tolerances are per element, as in tests/test_ops_gpu.py (one step of the rounded result + a stated accumulation floor).
Skipped where the library is missing.
"""

import numpy as np
import pytest
import torch

from helpers import assert_rounded_close, w4_abs_dot
from oracle import ref_kernels as K
from oracle import tiny_oracle as O

# First device run: profiles/r02_labs/zz_gpu_tests_first_device_run.log (all passed).
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not K.available(), reason="oracle/_ref was not built")]

DEV = "cuda" if torch.cuda.is_available() else "cpu"  # "cpu" only in the build container's dry run (oracle behind the C ABI)
TORCH = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


@pytest.fixture(scope="module")
def ext():
    import tiny_llm_ext_hip

    tiny_llm_ext_hip.load_library(".")
    return tiny_llm_ext_hip


def dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV, TORCH[dtype])


def host(t):
    return t.float().cpu().numpy()


# Per-element tolerances (helpers.assert_rounded_close).  Both sides accumulate in fp32 here (the reference's kernel on the host cores,
# the HIP kernel on the device), in different orders, and round once: up to one step of the result apart, plus a floor for values
# that are small against the sums they were formed from.
ACC_FLOOR = 2.0 ** -19  # 2 x 16 fp32 steps of the absolute sum: two fp32 accumulations


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_w4_matvec_and_embedding(ext, dtype):
    rng = np.random.default_rng(1)
    for rows, cols, M in ((40, 256, 1), (24, 512, 3), (64, 1024, 8)):
        w = O.cast(rng.standard_normal((rows, cols)).astype(np.float32) * 0.05, dtype)
        packed, scales, biases = O.quantize_affine(w, dtype=dtype)
        a = O.cast(rng.standard_normal((M, cols)).astype(np.float32), dtype)
        want = K.quantized_matvec_x4_fast(scales, biases, a, packed, dtype)  # the reference's decode GEMV kernel
        p = torch.from_numpy(np.ascontiguousarray(packed).view(np.int32)).to(DEV)
        got = ext.quantized_matmul(dev(scales, dtype), dev(biases, dtype), 128, 4, dev(a, dtype), p, True)
        assert_rounded_close(host(got), want, dtype, ulps=1.0, floor=ACC_FLOOR * w4_abs_dot(a, packed, scales, biases, dtype),
                             what=f"W4 matvec vs the reference kernel, {dtype} {rows}x{cols} M={M}")
        idx = np.array([3, rows - 1, 0, 7], dtype=np.int32)
        want = K.quantized_embedding(idx, scales, biases, packed, dtype)
        got = ext.quantized_embedding(torch.from_numpy(idx).to(DEV), dev(scales, dtype), dev(biases, dtype), p, 128, 4)
        assert np.array_equal(host(got), want)  # one rounding of q * s + b: nothing to differ in


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_pointwise_kernels(ext, dtype):
    rng = np.random.default_rng(2)
    x = O.cast(rng.standard_normal((5, 1000)).astype(np.float32) * 1.7, dtype)
    w = O.cast(1 + 0.1 * rng.standard_normal(1000).astype(np.float32), dtype)
    pw = 1.0 if dtype != "f32" else 8.0  # 16-bit outputs: one step; fp32 outputs: a few fp32 steps of the two evaluation orders
    assert_rounded_close(host(ext.rms_norm(dev(x, dtype), dev(w, dtype), 1e-6)), K.rms_norm(x, w, 1e-6, dtype), dtype, ulps=pw, what=f"rms_norm {dtype}")
    g = O.cast(rng.standard_normal((7, 96)).astype(np.float32) * 3, dtype)
    u = O.cast(rng.standard_normal((7, 96)).astype(np.float32), dtype)
    assert_rounded_close(host(ext.swiglu(dev(g, dtype), dev(u, dtype))), K.swiglu(g, u, dtype), dtype, ulps=pw, what=f"swiglu {dtype}")
    h = O.cast(rng.standard_normal((2, 3, 6, 64)).astype(np.float32), dtype)
    offsets = np.array([0, 117], dtype=np.int32)
    for traditional in (False, True):
        for dims in (64, 32):
            got = ext.rope(dev(h, dtype), torch.from_numpy(offsets).to(DEV), dims, 1000000.0, traditional)
            # the rotation angle is formed in fp32 on both sides: at position 117 it may differ by an fp32 step, i.e. 117 * 2^-23 of |h|
            assert_rounded_close(host(got), K.rope(h, offsets, dims, 1000000.0, traditional, dtype), dtype, ulps=pw,
                                 floor=117 * 2.0 ** -22 * float(np.abs(h).max()), what=f"rope {dtype} dims={dims} traditional={traditional}")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_attention_kernels(ext, dtype):
    rng = np.random.default_rng(3)
    Hq, Hkv, D, S, L = 4, 2, 64, 37, 2
    q = O.cast(rng.standard_normal((2 * Hq, L, D)).astype(np.float32), dtype)
    k = O.cast(rng.standard_normal((2 * Hkv, S, D)).astype(np.float32), dtype)
    v = O.cast(rng.standard_normal((2 * Hkv, S, D)).astype(np.float32), dtype)
    mask = np.where(rng.random((2 * Hq, L, S)) < 0.2, -2.0, 0.0).astype(np.float32)
    none = torch.zeros((1,), dtype=torch.float32, device=DEV)
    for causal, m in ((True, None), (False, mask)):
        got = ext.decode_attention(dev(q, dtype), dev(k, dtype), dev(v, dtype), none if m is None else torch.from_numpy(m).to(DEV), D ** -0.5, causal,
                                   m is not None, Hq, Hkv)
        assert_rounded_close(host(got), K.decode_attention(q, k, v, D ** -0.5, Hq, Hkv, causal, m, dtype), dtype, ulps=1.0 if dtype != "f32" else 16.0,
                             floor=2.0 ** -18 * float(np.abs(v).max()), what=f"decode attention {dtype}")
    # paged: scattered pages, a context ending inside a page, an idle row, L = 1 and 3
    P, page = 7, 8
    kp = O.cast(rng.standard_normal((P, Hkv, page, D)).astype(np.float32), dtype)
    vp = O.cast(rng.standard_normal((P, Hkv, page, D)).astype(np.float32), dtype)
    table = np.array([[5, 1, 3, -1], [-1, -1, -1, -1], [0, 2, 6, 4]], dtype=np.int32)
    ctx = np.array([21, 0, 32], dtype=np.int32)
    for L in (1, 3):
        q = O.cast(rng.standard_normal((3 * Hq, L, D)).astype(np.float32), dtype)
        got = ext.paged_attention(dev(q, dtype), dev(kp, dtype), dev(vp, dtype), torch.from_numpy(table).to(DEV), torch.from_numpy(ctx).to(DEV), D ** -0.5,
                                  True, num_kv_heads=Hkv, num_heads=Hq)
        assert_rounded_close(host(got), K.paged_attention_decode(q, kp, vp, table, ctx, D ** -0.5, True, Hkv, Hq, dtype), dtype,
                             ulps=1.0 if dtype != "f32" else 16.0, floor=2.0 ** -18 * float(np.abs(vp).max()), what=f"paged decode attention {dtype} L={L}")
    values = O.cast(rng.standard_normal((1, Hkv, 3, D)).astype(np.float32), dtype)
    pages = dev(kp, dtype)
    ext.paged_cache_update(pages, dev(values, dtype), 2, 4)
    assert np.array_equal(host(pages), K.paged_cache_update(kp, values, 2, 4, dtype))


def test_fp32_paged_prefill_kernel(ext):
    rng = np.random.default_rng(4)
    Hq, Hkv, D, page = 4, 2, 64, 8
    kp = rng.standard_normal((6, Hkv, page, D)).astype(np.float32)
    vp = rng.standard_normal((6, Hkv, page, D)).astype(np.float32)
    table = np.array([[5, 1, 3, -1], [0, 2, 4, -1]], dtype=np.int32)
    ctx = np.array([24, 20], dtype=np.int32)
    q = rng.standard_normal((2 * Hq, 20, D)).astype(np.float32)
    got = ext.paged_attention(dev(q, "f32"), dev(kp, "f32"), dev(vp, "f32"), torch.from_numpy(table).to(DEV), torch.from_numpy(ctx).to(DEV), D ** -0.5, True,
                              num_kv_heads=Hkv, num_heads=Hq)
    assert_rounded_close(host(got), K.paged_attention_scalar_f32(q, kp, vp, table, ctx, D ** -0.5, True, Hkv, Hq), "f32", ulps=16.0,
                         floor=2.0 ** -18 * float(np.abs(vp).max()), what="fp32 paged prefill attention")
