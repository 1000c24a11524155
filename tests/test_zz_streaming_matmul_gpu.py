"""GPU parity of the row-streaming batched-decode matmul (csrc/qmm7.h, round 6): gate|up and qkv of a 5..64-row decode step at the
Qwen3-4B shapes -- the wave walks its quantisation groups outside and the workgroup's tiles inside, the rows' columns of a group
arrive three groups ahead instead of all rows first, row blocks are exactly ceil(M / 16) (3 included).  Called through the
kernel-level C entry (tl_decode_linear_ex, kernel 6) and held

  * against the numpy oracle with the weighted-row allowance of tests/test_zz_batched_matmul_gpu.py (the reference's order is
    FastRMSNorm then the matvec: week2_kernels.metal:6-48, quantized_matmul.metal:441-538), and
  * BIT FOR BIT against the register-resident kernel (kernel 5) on the same inputs: the two spell the same arithmetic -- MFMA chains per
    (row block, group), fmaf(beta', sum a, fmaf(s, raw, acc)) in group order, the four waves' sums added in wave order, one rounding.
"""

import numpy as np
import pytest
import torch

from oracle import tiny_oracle as O
from helpers import assert_within, bf16_ulp, log_parity
from test_decode_kernels_gpu import (DEV, EPS, EPI_STORE, EPI_SWIGLU, PRO_NONE, PRO_RMS_WEIGHTED, _Projection, _bf16_host, _cache,
                                     _weighted_rows_case, ext)  # noqa: F401  (ext is a fixture)

pytestmark = pytest.mark.gpu

ROWS = [5, 8, 9, 16, 17, 31, 32, 33, 47, 48, 49, 64]
TILES = {"qkv": 2, "gate_up": 5}


def _proj(ext, name):
    if name not in _cache:
        _cache[name] = _Projection(ext, name)
    return _cache[name]


def _assert_plan(name, M, info, what):
    assert info["kernel"] == 6 and info["launches"] == 1, f"{what}: {info}"
    assert tuple(info["p"][:4]) == ((M + 15) // 16, 5, TILES[name], 1), f"{what}: (MB, GPW, T, row blocks) {info['p']}"


@pytest.mark.parametrize("M", ROWS)
@pytest.mark.parametrize("name", ["qkv", "gate_up"])
def test_weighted_rows_against_the_oracle_and_the_register_resident_kernel(ext, name, M):
    p = _proj(ext, name)
    a_w, ss = _weighted_rows_case(p, M)
    frag = ext.fragment_order_of(a_w)
    kw = dict(prologue=PRO_RMS_WEIGHTED, epilogue=p.epi, eps=EPS, ss_in=ss, fragment_order=True, fragment_rows=M)
    got, info = ext.decode_linear(p.tiled, frag, kernel=6, **kw)
    what = f"qmm7 weighted rows {name} M={M} {info['p']}"
    _assert_plan(name, M, info, what)
    twin, info5 = ext.decode_linear(p.tiled, frag, kernel=5, **kw)
    assert info5["kernel"] == 5
    assert torch.equal(got, twin), f"{what}: {int((got != twin).sum())} elements differ from the register-resident kernel"
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
    log_parity({"what": "qmm7_weighted_rows", "name": name, "M": M, "max_abs_err": float(err.max()),
                "share_of_allowance_max": float((err / allowed).max()), "p": info["p"]})


def test_row_major_rows_are_reordered_by_the_entry_point(ext):
    """Without fragment_order the entry point re-orders the caller's row-major weighted rows itself: same bits."""
    p = _proj(ext, "qkv")
    a_w, ss = _weighted_rows_case(p, 21)
    a, ia = ext.decode_linear(p.tiled, a_w, prologue=PRO_RMS_WEIGHTED, epilogue=EPI_STORE, eps=EPS, kernel=6, ss_in=ss)
    b, ib = ext.decode_linear(p.tiled, ext.fragment_order_of(a_w), prologue=PRO_RMS_WEIGHTED, epilogue=EPI_STORE, eps=EPS, kernel=6, ss_in=ss,
                              fragment_order=True, fragment_rows=21)
    assert ia["kernel"] == ib["kernel"] == 6 and torch.equal(a, b)


def test_partial_sums_of_squares_in_any_supported_count(ext):
    p = _proj(ext, "gate_up")
    M = 40
    a_w, ss160 = _weighted_rows_case(p, M)
    ss8 = torch.zeros((M, 8), dtype=torch.float32, device=DEV)
    ss8[:, 0] = ss160.double().sum(dim=1).float()
    a, _ = ext.decode_linear(p.tiled, a_w, prologue=PRO_RMS_WEIGHTED, epilogue=EPI_SWIGLU, eps=EPS, kernel=6, ss_in=ss160)
    b, _ = ext.decode_linear(p.tiled, a_w, prologue=PRO_RMS_WEIGHTED, epilogue=EPI_SWIGLU, eps=EPS, kernel=6, ss_in=ss8)
    diff = (a.float() - b.float()).abs()
    assert float((diff > 0).float().mean()) < 0.02 and float(diff.max()) <= 2 * float(bf16_ulp(np.abs(_bf16_host(a)).max()))


@pytest.mark.parametrize("M", [5, 17, 33, 40, 49])
def test_rows_are_read_inside_ceil16_rows_and_dead_rows_do_not_leak(ext, M):
    """The rows arrive as ceil16(M) rows in fragment order; the slots of the last block beyond M hold whatever the producer left.  NaN
    there must not reach a live output, and nothing past the buffer may be read (the rows as the tail of a dedicated allocation)."""
    qkv = _proj(ext, "qkv")
    a_w, ss = _weighted_rows_case(qkv, M)
    rows16 = (M + 15) // 16 * 16
    padded = torch.full((rows16, qkv.N), float("nan"), dtype=torch.bfloat16, device=DEV)
    padded[:M] = a_w
    want, _ = ext.decode_linear(qkv.tiled, ext.fragment_order_of(a_w), prologue=PRO_RMS_WEIGHTED, epilogue=EPI_STORE, eps=EPS, kernel=6, ss_in=ss,
                                fragment_order=True, fragment_rows=M)
    frag = ext.fragment_order_of(padded)
    got, info = ext.decode_linear(qkv.tiled, frag, prologue=PRO_RMS_WEIGHTED, epilogue=EPI_STORE, eps=EPS, kernel=6, ss_in=ss, fragment_order=True, fragment_rows=M)
    assert info["kernel"] == 6 and torch.equal(got, want) and torch.isfinite(got.float()).all()
    n = 32 * 1024 * 1024
    slab = torch.zeros((n,), dtype=torch.bfloat16, device=DEV)
    tail = slab[n - rows16 * qkv.N:]
    tail.copy_(frag.reshape(-1))
    got2, _ = ext.decode_linear(qkv.tiled, tail.reshape(frag.shape), prologue=PRO_RMS_WEIGHTED, epilogue=EPI_STORE, eps=EPS, kernel=6, ss_in=ss,
                                fragment_order=True, fragment_rows=M)
    torch.cuda.synchronize()
    assert torch.equal(got2, want)


def test_refusals_name_the_cause(ext):
    p = _proj(ext, "qkv")
    with pytest.raises(RuntimeError, match="weighted rows"):
        ext.decode_linear(p.tiled, p.a[:8].contiguous(), prologue=PRO_NONE, epilogue=EPI_STORE, eps=EPS, kernel=6)
    wo = _proj(ext, "wo")
    a_w, ss = _weighted_rows_case(p, 8)
    with pytest.raises(RuntimeError, match="does not cover this shape"):
        ext.decode_linear(wo.tiled, torch.zeros((8, wo.N), dtype=torch.bfloat16, device=DEV), prologue=PRO_RMS_WEIGHTED, epilogue=EPI_STORE, eps=EPS,
                          kernel=6, ss_in=torch.ones((8, 160), dtype=torch.float32, device=DEV))
