// FP8 (OCP E4M3) KV pages: the device side of the format oracle/kv_fp8.py states (the reference has no
// quantised cache, README.md:134-135 -- an extension, not a restatement).
//
// A K or V row (one token, one kv head, D = 128 bf16 values) is stored as D E4M3 codes + ONE float32 scale s = 2^e, the smallest
// power of two with amax / s <= 448: x / s is exact, nothing saturates, and code * s is exactly a bf16 value -- attention over a
// quantised page is the bf16 kernels' arithmetic over the dequantised rows.  Because s is a power of two it also commutes with
// every rounding of those kernels: a score is s_k * (q . codes), a value sum takes the weight p * s_v -- the walks fold the scales
// into one multiply per (token, head) instead of one per element.
//   pages  [P, Hkv, page, D] uint8   (the bf16 pool's layout, one byte per element)
//   scales [P, Hkv, page]    float32 (index = the row index of the page layout)
// gfx950 converts in hardware: v_cvt_pk_fp8_f32 (2 floats -> 2 codes, round to nearest even), v_cvt_pk_f32_fp8,
// v_cvt_scalef32_pk_bf16_fp8 (2 codes x a power-of-two scale -> 2 bf16).
#pragma once
#include "common.h"

namespace tl {

constexpr int KV8_SCALE_EXP_MIN = 16, KV8_SCALE_EXP_MAX = 250;  // float32 exponent field of a row scale (oracle/kv_fp8.py)

// row scale from the row's largest magnitude, by exponent arithmetic (oracle/kv_fp8.py row_scale): amax = m 2^e, m in [1, 2):
// amax / 448 = (m / 1.75) 2^(e - 8)  ->  s = 2^(e - 8) for m <= 1.75, else 2^(e - 7).  inv = 1 / s (exact).
__device__ __forceinline__ void kv8_row_scale(float amax, float &s, float &inv) {
    const uint32_t bits = __float_as_uint(amax);
    int es = (int)((bits >> 23) & 0xffu) - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
    es = min(max(es, KV8_SCALE_EXP_MIN), KV8_SCALE_EXP_MAX);
    s = __uint_as_float((uint32_t)es << 23);
    inv = __uint_as_float((uint32_t)(254 - es) << 23);
}

// 8 values (already divided by the row scale) -> 8 codes
__device__ __forceinline__ u32x2 kv8_pack8(const float (&x)[8]) {
    u32x2 r;
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], w, true);
    r[0] = (uint32_t)w;
    w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(x[4], x[5], w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(x[6], x[7], w, true);
    r[1] = (uint32_t)w;
    return r;
}

// 8 codes -> 8 floats, UNSCALED (the caller folds the row scale into its own arithmetic)
__device__ __forceinline__ void kv8_unpack8(u32x2 c, float (&f)[8]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)c[h], false);
        const f32x2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)c[h], true);
        f[4 * h + 0] = lo[0];
        f[4 * h + 1] = lo[1];
        f[4 * h + 2] = hi[0];
        f[4 * h + 3] = hi[1];
    }
}

// 8 codes x the row scale -> the 8 bf16 values a bf16 page would hold (exact)
__device__ __forceinline__ u32x4 kv8_to_bf16x8(u32x2 c, float s) {
    u32x4 r;
    r[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)c[0], s, false));
    r[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)c[0], s, true));
    r[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)c[1], s, false));
    r[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)c[1], s, true));
    return r;
}

// One row spread over an aligned group of 16 lanes, 8 values per lane (the decode kernels' prologue shape): quantise it and hand back
// what the page will hold -- codes, scale, and the dequantised values (what this step attends to for the token being decoded).
__device__ __forceinline__ void kv8_quantize_row16(const float (&x)[8], u32x2 &codes, float &s, float (&deq)[8]) {
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(x[i]));
    amax = group16_max(amax);
    float inv;
    kv8_row_scale(amax, s, inv);
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = x[i] * inv;
    codes = kv8_pack8(y);
    kv8_unpack8(codes, deq);
#pragma unroll
    for (int i = 0; i < 8; ++i) deq[i] *= s;
}

}  // namespace tl
