"""TEST INFRASTRUCTURE -- CPU oracle for FP8 (E4M3) quantised KV pages.  Never imported by the product path.

SURVEY section 8 row f4, second half ("quantized-KV").  The reference has NO behaviour for it: `/root/reference/README.md:134-135`
lists quantised KV caches as not covered, so nothing here restates reference code and there is no reference vector to match.
**Parity unpinned against the reference (nothing to pin to)**; what IS pinned:

* the number format is the published OCP 8-bit floating point E4M3 ("FN": bias 7, 3 mantissa bits, no infinities, 0x7f / 0xff =
  NaN, largest finite 448, subnormals in steps of 2^-9) -- the format gfx950's `v_cvt_pk_fp8_f32` / `v_cvt_scalef32_pk_bf16_fp8`
  implement.  `encode_e4m3` / `decode_e4m3` are held bit for bit against PyTorch's `torch.float8_e4m3fn` casts (round to nearest
  even) over every code and over ties / subnormals / the largest values (`tests/test_kv_fp8_oracle_cpu.py`);
* the semantics of a quantised cache are stated once, here, and the HIP path is held to them bit for bit (codes and scales) or to
  the bf16 path's own tolerances (attention over the dequantised pages):

  A K (or V) row = the `D` values of ONE token and ONE kv head, bf16 as the bf16 cache would hold them (K after k-norm + RoPE,
  `qwen3_week3.py:63-86`).  It is stored as `D` E4M3 codes and ONE float32 scale `s`, a power of two:
  `s = 2^ceil(log2(amax / 448))` (the smallest power of two with `amax / s <= 448`; exponent field clamped to [16, 250]),
  `code_i = E4M3_rne(x_i / s)`.  `x_i / s` is exact (power of two), never above 448 (no saturation case), and the dequantised
  value `decode(code_i) * s` is EXACTLY representable in bf16 (3 mantissa bits x a power of two) -- so "attention over a
  quantised cache" is, by definition, the bf16 cache's attention (`paged_attention.metal:108-506`, restated in
  `tiny_oracle.paged_attention`) over the dequantised rows, every token included (the token being decoded is quantised before
  it is attended to: what a later step reads from the page is what this step used).
"""
from __future__ import annotations

import numpy as np

E4M3_MAX = 448.0
SCALE_EXP_MIN, SCALE_EXP_MAX = 16, 250  # float32 exponent FIELD of a row scale


def decode_e4m3(codes) -> np.ndarray:
    """uint8 codes -> float32 (exact).  OCP FP8 E4M3: s eeee mmm, bias 7; e = 0 subnormal (m * 2^-9); 0x7f / 0xff NaN."""
    c = np.asarray(codes, dtype=np.uint8).astype(np.int32)
    sign = np.where(c & 0x80, -1.0, 1.0).astype(np.float32)
    e = (c >> 3) & 0xF
    m = c & 0x7
    mag = np.where(e == 0, m.astype(np.float32) * np.float32(2.0 ** -9),
                   (1.0 + m.astype(np.float32) / 8.0) * np.exp2((e - 7).astype(np.float32)))
    out = (sign * mag).astype(np.float32)
    return np.where((c & 0x7F) == 0x7F, np.float32(np.nan), out)


def encode_e4m3(x) -> np.ndarray:
    """float32 -> uint8 codes, round to nearest even, magnitudes above 448 saturate to 448 (the row-scale rule never produces one)."""
    x = np.asarray(x, dtype=np.float32)
    a = np.minimum(np.abs(x).astype(np.float64), E4M3_MAX)
    sub = a < 2.0 ** -6
    # subnormal range: multiples of 2^-9 (8 of them reach the smallest normal, code 0x08, which the formula below also yields)
    q_sub = np.rint(a * 2.0 ** 9).astype(np.int64)
    mant, exp = np.frexp(np.where(sub, 1.0, a))  # a = mant * 2^exp, mant in [0.5, 1)
    e = exp.astype(np.int64) - 1                 # a = (2 mant) * 2^e, 2 mant in [1, 2)
    q = np.rint((2.0 * mant - 1.0) * 8.0).astype(np.int64)  # 0 .. 8, ties to even
    carry = q == 8
    e = np.where(carry, e + 1, e)
    q = np.where(carry, 0, q)
    code = np.where(sub, q_sub, ((e + 7) << 3) | q)
    code = np.minimum(code, 0x7E)
    code = np.where(np.signbit(x), code | 0x80, code)
    return code.astype(np.uint8)


def row_scale(amax) -> np.ndarray:
    """The power-of-two scale of a row from its largest magnitude (float32), by exponent arithmetic (no rounding anywhere):
    amax = m * 2^e with m in [1, 2): amax / 448 = (m / 1.75) * 2^(e - 8), so s = 2^(e - 8) when m <= 1.75 and 2^(e - 7) otherwise."""
    bits = np.asarray(amax, dtype=np.float32).view(np.uint32).astype(np.int64)
    E = (bits >> 23) & 0xFF
    M = bits & 0x7FFFFF
    Es = np.clip(E - 8 + (M > 0x600000), SCALE_EXP_MIN, SCALE_EXP_MAX)
    return (Es.astype(np.uint32) << 23).view(np.float32)


def quantize_rows(x):
    """x [..., D] (bf16 values in float32 containers) -> (codes uint8 [..., D], scales float32 [...])."""
    x = np.asarray(x, dtype=np.float32)
    s = row_scale(np.max(np.abs(x), axis=-1))
    return encode_e4m3(x / s[..., None]), s


def dequantize_rows(codes, scales) -> np.ndarray:
    """(codes [..., D], scales [...]) -> float32 [..., D]; every value is exactly a bf16 value."""
    return decode_e4m3(codes) * np.asarray(scales, dtype=np.float32)[..., None]


def round_trip(x) -> np.ndarray:
    """What a quantised cache returns for rows x [..., D]."""
    return dequantize_rows(*quantize_rows(x))


def paged_cache_update(pages, page_scales, values, page_id: int, start: int):
    """The quantising twin of `tiny_oracle.paged_cache_update` (`paged_attention.metal:82-106`: values [1, H, len, D] into
    pages [P, H, page, D] at (page_id, start)); page_scales [P, H, page] float32.  In place, returns the two arrays."""
    v = np.asarray(values, dtype=np.float32)
    assert v.shape[0] == 1
    codes, s = quantize_rows(v[0])
    n = v.shape[2]
    pages[page_id, :, start:start + n, :] = codes
    page_scales[page_id, :, start:start + n] = s
    return pages, page_scales


def dequantize_pages(pages, page_scales) -> np.ndarray:
    return dequantize_rows(pages, page_scales)
