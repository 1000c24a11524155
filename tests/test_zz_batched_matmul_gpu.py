"""GPU parity of the register-resident batched-decode matmul (csrc/qmm6.h, round 4): the kernel that takes every projection of a
5..64-row decode step -- rows in registers as MFMA A operands, the reduction dimension split across the four waves of a workgroup,
the epilogue (store / residual + sums of squares + weighted rows / SwiGLU) inside the launch, RMSNorm as `weighted rows in, 1 / rms on
the sums`.  Called through the kernel-level C entry (tl_decode_linear / tl_decode_linear_ex, kernel 5) at the real Qwen3-4B shapes and
held against the numpy oracle on the seeded matrices of tests/test_decode_kernels_gpu.py, the way the reference tests its matvec per
shape (tests_refsol/test_week_2_day_3.py:89-118; kernel semantics: src/extensions_ref/src/quantized_matmul.metal:441-538).

Tolerances: the plain allowance of test_decode_kernels_gpu.py (one bf16 step of the accumulated value, two for a residual); for
weighted rows additionally the 6-sigma term of the moved rounding point (x * w rounded instead of x * inv * w), exactly as the
weighted-row GEMV test there derives it.
"""

import numpy as np
import pytest
import torch

from oracle import tiny_oracle as O
from helpers import assert_within, bf16_ulp, log_parity
from test_decode_kernels_gpu import (DEV, EPS, EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, PRO_NONE, PRO_RMS_WEIGHTED, PROJECTIONS, _Projection,
                                     _bf16_host, _cache, _weighted_rows_case, ext)  # noqa: F401  (ext is a fixture)

pytestmark = pytest.mark.gpu

ROWS = [5, 8, 16, 17, 32, 33, 48, 64]
# groups per wave by projection (the row's 128-column groups over the four waves, rounded up to a compiled count), and the 16-row blocks per
# workgroup the planner may pick: the fragments of MB x GPW x 4 k-steps must fit the register file; since round 6 long rows against few tiles (wo)
# take the block count that is cheapest by the planner's arithmetic -- 16-row blocks at every row count -- and the others the largest
# (tests/test_decode_plans_cpu.py pins the plans at these shapes)
GPW = {"qkv": 5, "gate_up": 5, "lm_head": 5, "wo": 8, "down": 19}
LARGEST = {"qkv": {1: 1, 2: 2, 3: 4, 4: 4}, "gate_up": {1: 1, 2: 2, 3: 4, 4: 4}, "lm_head": {1: 1, 2: 2, 3: 4, 4: 4}, "wo": {1: 1, 2: 2, 3: 2, 4: 2},
           "down": {1: 1, 2: 1, 3: 1, 4: 1}}


def _proj(ext, name):
    if name not in _cache:
        _cache[name] = _Projection(ext, name)
    return _cache[name]


def _assert_plan(name, M, info, what):
    import ctypes

    import tiny_llm_ext_hip as e

    blocks = (M + 15) // 16
    assert info["kernel"] == 5 and info["launches"] == 1, f"{what}: {info}"
    mb, gpw, row_blocks = info["p"][0], info["p"][1], info["p"][3]
    assert gpw == GPW[name] and mb in (1, 2, 4) and mb * gpw * 16 <= 320 and mb <= LARGEST[name][blocks], f"{what}: planned (MB, GPW) {(mb, gpw)}"
    assert row_blocks == (blocks + mb - 1) // mb, f"{what}: row blocks {info['p']}"
    if name in ("qkv", "gate_up", "lm_head", "down"):
        assert mb == LARGEST[name][blocks], f"{what}: this projection keeps the largest block, planned {mb}"
    if name == "wo":
        assert mb == 1, f"{what}: wo runs in 16-row blocks, planned {mb}"
    # what ran is what the planner's own entry point says for this shape
    out = (ctypes.c_int * 6)()
    p = _cache[name]
    assert e._lib.tl_decode_batched_plan(M, p.K, p.N, out) == 1 and (out[0], out[1], out[3]) == (mb, gpw, row_blocks), f"{what}: tl_decode_batched_plan says {tuple(out)}"


@pytest.mark.parametrize("M", ROWS)
@pytest.mark.parametrize("name", ["qkv", "wo", "down", "lm_head"])
def test_plain_rows_store_and_residual(ext, name, M):
    """Plain bf16 rows: store (qkv / lm_head matrices) and residual add (wo / w_down) with the producer's hand-over -- per (row, 16-row
    tile) sums of squares of the stored values and the rows weighted by the consumer's RMSNorm weight."""
    p = _proj(ext, name)
    if p.epi == EPI_RESIDUAL:
        norm_out = (1.0 + 0.05 * torch.randn((p.K,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(6))).to(torch.bfloat16)
        got, info = ext.decode_linear(p.tiled, p.a[:M].contiguous(), prologue=PRO_NONE, epilogue=EPI_RESIDUAL, residual=p.residual[:M].contiguous(),
                                      eps=EPS, kernel=5, want_ss_out=True, norm_out=norm_out)
        what = f"qmm6 residual {name} M={M} {info['p']}"
        _assert_plan(name, M, info, what)
        assert_within(_bf16_host(got), p.want[(PRO_NONE, EPI_RESIDUAL)][:M], p.allowed[(PRO_NONE, EPI_RESIDUAL)][:M], what=what)
        g64 = got.double()
        assert torch.allclose(info["ss_out"].double(), (g64 * g64).reshape(M, p.K // 16, 16).sum(dim=2), rtol=1e-5, atol=1e-9), f"{what}: sums of squares"
        assert torch.equal(info["out_w"], (got.float() * norm_out.float()).to(torch.bfloat16)), f"{what}: weighted rows"
    else:
        got, info = ext.decode_linear(p.tiled, p.a[:M].contiguous(), prologue=PRO_NONE, epilogue=EPI_STORE, eps=EPS, kernel=5)
        what = f"qmm6 store {name} M={M} {info['p']}"
        _assert_plan(name, M, info, what)
        assert_within(_bf16_host(got), p.want[(PRO_NONE, EPI_STORE)][:M], p.allowed[(PRO_NONE, EPI_STORE)][:M], what=what)


@pytest.mark.parametrize("M", ROWS)
@pytest.mark.parametrize("name", ["qkv", "gate_up", "lm_head"])
def test_weighted_rows_against_the_reference_order(ext, name, M):
    """Rows weighted by their producer + its 160 partial sums of squares per row: bf16(inv * (bf16(x w) @ W^T)) against the reference's
    bf16(bf16(x inv w) @ W^T) (FastRMSNorm then the matvec).  One rounding per staged element on both sides, of different products:
    sigma^2 = 2 (u^2 / 12) sum_n (a_n W_kn)^2, u <= 2^-7; 6 sigma on top of the plain allowance."""
    p = _proj(ext, name)
    a_w, ss = _weighted_rows_case(p, M)
    got, info = ext.decode_linear(p.tiled, a_w, prologue=PRO_RMS_WEIGHTED, epilogue=p.epi, eps=EPS, kernel=5, ss_in=ss)
    what = f"qmm6 weighted rows {name} M={M} {info['p']}"
    _assert_plan(name, M, info, what)
    normed = O.rms_norm_fast(_bf16_host(p.a[:M]), _bf16_host(p.norm_w), EPS)
    pre = p.want[(p.pro, p.epi)][:M] if p.epi == EPI_STORE else O.quantized_matmul(p.scales_host, p.biases_host, normed, p.packed_host, "bf16")
    sigma = np.sqrt(2.0 / 12.0) * 2.0 ** -7 * np.sqrt(p.squared_dot(normed))
    floor = 2e-4 * max(1.0, p.scale)
    if p.epi == EPI_SWIGLU:
        g, u = pre[:, 0::2].astype(np.float64), pre[:, 1::2].astype(np.float64)
        dg, du = 6.0 * sigma[:, 0::2], 6.0 * sigma[:, 1::2]
        want = O.swiglu(pre[:, 0::2], pre[:, 1::2])
        allowed = 1.1 * (bf16_ulp(g) + floor + dg) * np.abs(u) + (bf16_ulp(u) + floor + du) * np.abs(g / (1 + np.exp(-g))) + bf16_ulp(want)
    else:
        want = pre
        allowed = bf16_ulp(pre) + floor + 6.0 * sigma
    assert_within(_bf16_host(got), want, allowed, what=what)
    err = np.abs(_bf16_host(got).astype(np.float64) - want)
    log_parity({"what": "qmm6_weighted_rows", "name": name, "M": M, "max_abs_err": float(err.max()),
                "share_of_allowance_max": float((err / allowed).max()), "p": info["p"]})


def test_partial_sums_of_squares_in_any_supported_count(ext):
    """8 partials per row (what the embedding kernels leave ahead of layer 0) give the same values as 160."""
    p = _proj(ext, "qkv")
    M = 24
    a_w, ss160 = _weighted_rows_case(p, M)
    ss8 = torch.zeros((M, 8), dtype=torch.float32, device=DEV)
    ss8[:, 0] = ss160.double().sum(dim=1).float()
    a, ia = ext.decode_linear(p.tiled, a_w, prologue=PRO_RMS_WEIGHTED, epilogue=EPI_STORE, eps=EPS, kernel=5, ss_in=ss160)
    b, ib = ext.decode_linear(p.tiled, a_w, prologue=PRO_RMS_WEIGHTED, epilogue=EPI_STORE, eps=EPS, kernel=5, ss_in=ss8)
    assert ia["kernel"] == ib["kernel"] == 5
    diff = (a.float() - b.float()).abs()
    assert float((diff > 0).float().mean()) < 0.01 and float(diff.max()) <= float(bf16_ulp(np.abs(_bf16_host(a)).max())), \
        "the two partial counts may differ by the last bit of 1 / rms only"


def test_refusals_name_the_cause(ext):
    p = _proj(ext, "qkv")
    with pytest.raises(RuntimeError, match="plain rows|weighted rows"):
        ext.decode_linear(p.tiled, p.a[:8].contiguous(), prologue=1, epilogue=EPI_STORE, norm_weight=p.norm_w, eps=EPS, kernel=5)
    with pytest.raises(RuntimeError, match="ss_in"):
        ext.decode_linear(p.tiled, p.a[:8].contiguous(), prologue=PRO_RMS_WEIGHTED, epilogue=EPI_STORE, eps=EPS, kernel=5)


@pytest.mark.parametrize("M", [5, 16, 33, 64])
def test_slice_reduction_of_w_down_leaves_the_weighted_rows(ext, M):
    """w_down stays on the K-sliced matmul at 5..64 rows (76 groups against 160 tiles: csrc/engine.hip); its slice reduction hands the
    next projection the rows weighted by that projection's RMSNorm weight, bf16(out * norm_out), next to out itself."""
    p = _proj(ext, "down")
    norm_out = (1.0 + 0.05 * torch.randn((p.K,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))).to(torch.bfloat16)
    got, info = ext.decode_linear(p.tiled, p.a[:M].contiguous(), prologue=PRO_NONE, epilogue=EPI_RESIDUAL, residual=p.residual[:M].contiguous(),
                                  eps=EPS, kernel=2, norm_out=norm_out)
    what = f"skinny matmul + reduction with weighted rows, w_down M={M} {info['p']}"
    assert info["kernel"] == 2 and info["launches"] == 2, what
    assert_within(_bf16_host(got), p.want[(PRO_NONE, EPI_RESIDUAL)][:M], p.allowed[(PRO_NONE, EPI_RESIDUAL)][:M], what=what)
    assert torch.equal(info["out_w"], (got.float() * norm_out.float()).to(torch.bfloat16)), f"{what}: weighted rows"


@pytest.mark.parametrize("M", [5, 16, 40, 64])
def test_weighted_rows_travel_in_fragment_order(ext, M):
    """The engine hands weighted rows from producer to consumer in FRAGMENT ORDER (csrc/qmm6.h: every load of the consumer is one
    contiguous 1 KiB, no LDS pass).  Producers: the register-resident kernel's residual epilogue (wo shape) and the slice reduction of
    the K-sliced matmul (w_down shape) must write exactly the row-major weighted rows, re-ordered; the consumer must give the same
    bits whether the entry point re-orders row-major rows itself or takes them in fragment order."""
    wo, down, qkv = _proj(ext, "wo"), _proj(ext, "down"), _proj(ext, "qkv")
    for p, kernel in ((wo, 5), (down, 2)):
        norm_out = (1.0 + 0.05 * torch.randn((p.K,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(11))).to(torch.bfloat16)
        kw = dict(prologue=PRO_NONE, epilogue=EPI_RESIDUAL, residual=p.residual[:M].contiguous(), eps=EPS, kernel=kernel, norm_out=norm_out)
        got_r, info_r = ext.decode_linear(p.tiled, p.a[:M].contiguous(), **kw)
        got_f, info_f = ext.decode_linear(p.tiled, p.a[:M].contiguous(), fragment_order=True, **kw)
        assert torch.equal(got_r, got_f), f"{p.name} M={M}: the output rows must not depend on the layout of the weighted copy"
        assert info_f["out_w"].shape[0] == (M + 15) // 16 * 16
        assert torch.equal(ext.rows_from_fragment_order(info_f["out_w"], M), info_r["out_w"]), f"{p.name} M={M} kernel {kernel}: fragment order of out_w"
    a_w, ss = _weighted_rows_case(qkv, M)
    a, ia = ext.decode_linear(qkv.tiled, a_w, prologue=PRO_RMS_WEIGHTED, epilogue=EPI_STORE, eps=EPS, kernel=5, ss_in=ss)
    b, ib = ext.decode_linear(qkv.tiled, ext.fragment_order_of(a_w), prologue=PRO_RMS_WEIGHTED, epilogue=EPI_STORE, eps=EPS, kernel=5, ss_in=ss,
                              fragment_order=True, fragment_rows=M)
    assert ia["kernel"] == ib["kernel"] == 5 and torch.equal(a, b), f"qkv M={M}: rows in fragment order"


@pytest.mark.parametrize("M", [33, 40, 48])
def test_fragment_order_rows_are_read_inside_ceil16_rows(ext, M):
    """Rows in fragment order are provided as ceil16(M) rows (include/tinyllm_engine.h).  At 33..48 rows the planner's 64-row workgroup
    (MB = 4) has a fourth 16-row block with no rows behind it: the kernel must re-read a real block there (csrc/qmm6.h, FRAG path), not
    the 16 x N x 2 bytes past the buffer.  The rows are handed over as the head of a larger allocation whose tail is NaN: outputs must be
    bit-identical to the call on an exact-size buffer -- and the call runs at the very end of a dedicated allocation in the last case
    (a read past it is a memory fault on a box whose next pages are unmapped)."""
    qkv = _proj(ext, "qkv")
    a_w, ss = _weighted_rows_case(qkv, M)
    frag = ext.fragment_order_of(a_w)
    rows16 = (M + 15) // 16 * 16
    assert frag.numel() == rows16 * qkv.N
    want, info = ext.decode_linear(qkv.tiled, frag, prologue=PRO_RMS_WEIGHTED, epilogue=EPI_STORE, eps=EPS, kernel=5, ss_in=ss, fragment_order=True, fragment_rows=M)
    assert info["kernel"] == 5 and info["p"][0] == 4
    big = torch.full((rows16 * qkv.N + 64 * qkv.N,), float("nan"), dtype=torch.bfloat16, device=DEV)
    big[: rows16 * qkv.N] = frag.reshape(-1)
    got, _ = ext.decode_linear(qkv.tiled, big[: rows16 * qkv.N].reshape(frag.shape), prologue=PRO_RMS_WEIGHTED, epilogue=EPI_STORE, eps=EPS, kernel=5, ss_in=ss,
                               fragment_order=True, fragment_rows=M)
    assert torch.equal(got, want) and torch.isfinite(got.float()).all()
    # the same rows as the tail of a dedicated 64-MiB allocation (torch's allocator serves requests of this size with their own hipMalloc)
    n = 32 * 1024 * 1024
    slab = torch.zeros((n,), dtype=torch.bfloat16, device=DEV)
    tail = slab[n - rows16 * qkv.N:]
    tail.copy_(frag.reshape(-1))
    got2, _ = ext.decode_linear(qkv.tiled, tail.reshape(frag.shape), prologue=PRO_RMS_WEIGHTED, epilogue=EPI_STORE, eps=EPS, kernel=5, ss_in=ss,
                                fragment_order=True, fragment_rows=M)
    torch.cuda.synchronize()
    assert torch.equal(got2, want)
