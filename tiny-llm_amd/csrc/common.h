// Shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <string>

#include "../../include/tinyllm_hip.h"

namespace tl {

// ---- error plumbing ---------------------------------------------------------
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

#define TL_REQUIRE(cond, msg)                                 \
    do {                                                      \
        if (!(cond)) return ::tl::fail(TL_ERR_INVALID, msg);  \
    } while (0)

#define TL_CHECK_LAUNCH(name)                                                               \
    do {                                                                                    \
        hipError_t e__ = hipGetLastError();                                                 \
        if (e__ != hipSuccess)                                                              \
            return ::tl::fail(TL_ERR_HIP, std::string(name) + ": " + hipGetErrorString(e__)); \
    } while (0)

constexpr int WAVE = 64;

// ---- scalar type traits -----------------------------------------------------
// Storage types are raw 16-bit words for f16/bf16 so that vector loads are
// plain integer loads; conversion is explicit.
struct BF16 {
    using storage = uint16_t;
    static constexpr tl_dtype tag = TL_BF16;
    __device__ __forceinline__ static float to_float(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
    // round-to-nearest-even, NaN preserved (matches static_cast<bfloat16_t>(float))
    // (lowers to v_cvt_pk_bf16_f32 on gfx950)
    __device__ __forceinline__ static uint16_t from_float(float f) {
        return __builtin_bit_cast(uint16_t, (__bf16)f);
    }
    __device__ __forceinline__ static uint32_t pack2(float lo, float hi) {
        typedef __bf16 v2 __attribute__((ext_vector_type(2)));
        v2 r;
        r[0] = (__bf16)lo;
        r[1] = (__bf16)hi;
        return __builtin_bit_cast(uint32_t, r);
    }
};
struct F16 {
    using storage = uint16_t;
    static constexpr tl_dtype tag = TL_F16;
    __device__ __forceinline__ static float to_float(uint16_t v) { return __half2float(__ushort_as_half(v)); }
    __device__ __forceinline__ static uint16_t from_float(float f) { return __half_as_ushort(__float2half_rn(f)); }
    __device__ __forceinline__ static uint32_t pack2(float lo, float hi) {
        return (uint32_t)from_float(lo) | ((uint32_t)from_float(hi) << 16);
    }
};
struct F32 {
    using storage = float;
    static constexpr tl_dtype tag = TL_F32;
    __device__ __forceinline__ static float to_float(float v) { return v; }
    __device__ __forceinline__ static float from_float(float f) { return f; }
};

__device__ __forceinline__ float bf16_round(float f) { return BF16::to_float(BF16::from_float(f)); }

// ---- wave helpers -----------------------------------------------------------
// Reductions by DPP row rotations + four v_readlane (round 3).  __shfl_xor compiles to ds_bpermute_b32: a trip through the LDS
// crossbar (~100 cycles) per step, six DEPENDENT steps for a wave-wide sum -- a quarter of a microsecond on the critical path of
// every fused RMSNorm.  A DPP step is one VALU instruction.
template <int N>
__device__ __forceinline__ float dpp_row_ror(float v) {  // lane i of a 16-lane row reads lane (i + N) mod 16 of the same row
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
// all-reduce over aligned groups of 16 lanes (every lane of a group gets the group's sum)
__device__ __forceinline__ float group16_sum(float v) {
    v += dpp_row_ror<8>(v);
    v += dpp_row_ror<4>(v);
    v += dpp_row_ror<2>(v);
    v += dpp_row_ror<1>(v);
    return v;
}
template <int N>
__device__ __forceinline__ float dpp_row_ror_self(float v) {  // a disabled source lane yields the reader's own value
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float group16_max(float v) {
    v = fmaxf(v, dpp_row_ror_self<8>(v));
    v = fmaxf(v, dpp_row_ror_self<4>(v));
    v = fmaxf(v, dpp_row_ror_self<2>(v));
    v = fmaxf(v, dpp_row_ror_self<1>(v));
    return v;
}
// value of the lane whose index differs in bit 4 / bit 5 (gfx950 row / half swaps: one VALU instruction instead of ds_bpermute_b32)
//   v_permlane16_swap a, b: a.row1 <-> b.row0, a.row3 <-> b.row2;  v_permlane32_swap a, b: a.lanes[32:63] <-> b.lanes[0:31]
__device__ __forceinline__ float lane_xor16(float v, int lane) {
    const int b = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);  // r[0] = {v0, v0, v2, v2}, r[1] = {v1, v1, v3, v3} by rows
    return __builtin_bit_cast(float, (lane & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor32(float v, int lane) {
    const int b = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);  // r[0] = {lo, lo}, r[1] = {hi, hi} by halves
    return __builtin_bit_cast(float, (lane & 32) ? r[0] : r[1]);
}
// lane i reads lane i ^ 1 (DPP quad_perm [1, 0, 3, 2])
__device__ __forceinline__ float lane_xor1(float v) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, 0xB1, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = group16_sum(v);
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
    v = group16_max(v);
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// Softmax exponentials: the bare v_exp_f32.  exp2f() wraps it in denormal-range handling (compare, scale, select, ldexp: four
// more VALU instructions per call); a weight below 2^-126 is zero at every precision the attention result is kept in.
__device__ __forceinline__ float exp2_hw(float x) { return __builtin_amdgcn_exp2f(x); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

// ---- activations between the launches of a decode step -------------------------------------------------------------------------
// A kernel reads what the launch before it wrote.  Behind an ordinary launch boundary the runtime writes every L2 back and invalidates
// every L2 and L1 (agent-scope release / acquire: 0.4-0.55 us per launch on this chip, and a cold L2 for whoever comes next).  The AQL
// replay route (aql.h) issues the launches INSIDE a step without that maintenance; what makes that correct is a pair of rules:
//   * every store of a value another launch of the step reads is WRITTEN THROUGH to memory (a relaxed device-scope atomic store: sc1)
//     -- in the route's device-only code objects, which are compiled with -DTL_COHERENT; relaxed atomics of 2 / 4 / 8 bytes order
//     nothing and wait for nothing (a `volatile` store, the back end's other spelling, is followed by s_waitcnt vmcnt(0));
//   * every such value lives at an address that is written ONCE per step and read only after it (csrc/engine.hip: the decode
//     activations of the route are per-layer buffers): no cache -- L1, or another XCD's L2 -- can hold an older copy of the line, because
//     the step's first packet invalidates them all and nobody touched the line since; the consumers' loads stay PLAIN (one fetch per
//     XCD, then L2 hits for the other workgroups of that XCD -- device-scope loads were tried first: every workgroup's copy of the row
//     crossed the fabric, 0.99 -> 1.03 ms per step).
// The fat binary inside the library is compiled without the macro: plain stores, for the routes that keep their launch boundaries.
template <typename T>
__device__ __forceinline__ T act_load(const T *ptr) {
    return *ptr;
}
template <typename T>
__device__ __forceinline__ void act_store(T *ptr, T v) {
#ifdef TL_COHERENT
    static_assert(sizeof(T) == 2 || sizeof(T) == 4 || sizeof(T) == 8, "act_store: 2, 4 or 8 bytes");
    if constexpr (sizeof(T) == 2) __hip_atomic_store(reinterpret_cast<uint16_t *>(ptr), __builtin_bit_cast(uint16_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (sizeof(T) == 4) __hip_atomic_store(reinterpret_cast<uint32_t *>(ptr), __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(reinterpret_cast<uint64_t *>(ptr), __builtin_bit_cast(uint64_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *ptr = v;
#endif
}

// the same rule for stores that go through a buffer resource (the batched-decode matmuls): the cache-policy operand of a raw buffer store,
// sc1 = write-through in the route's code objects
#ifdef TL_COHERENT
constexpr int ACT_STORE_AUX = 16;
#else
constexpr int ACT_STORE_AUX = 0;
#endif
// ... and for 16-byte stores: two 8-byte halves in the route's code objects (there is no 16-byte relaxed atomic), one store otherwise
__device__ __forceinline__ void act_store16(void *ptr, u32x4 v) {
#ifdef TL_COHERENT
    act_store(reinterpret_cast<u32x2 *>(ptr), u32x2{v[0], v[1]});
    act_store(reinterpret_cast<u32x2 *>(ptr) + 1, u32x2{v[2], v[3]});
#else
    *reinterpret_cast<u32x4 *>(ptr) = v;
#endif
}

// ---- optional in-kernel timing (engine profile step) ---------------------------------------------
// buf = nullptr in normal operation.  Otherwise buf[2*wg] receives the workgroup's first timestamp and
// buf[2*wg+1] the maximum end timestamp over its waves (constant-rate wall clock, hipDeviceAttributeWallClockRate).
// prof_begin only READS the clock: a store at the top of a kernel makes hipcc treat every later load as possibly
// clobbered, which turns the wave-uniform loads of the prologue (context length, page id) from s_load into vector
// loads.  Both stamps are written by prof_end.
typedef unsigned long long prof_t;
__device__ __forceinline__ unsigned prof_wg() {
    return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
}
__device__ __forceinline__ prof_t prof_begin(const prof_t *buf) { return buf ? (prof_t)wall_clock64() : (prof_t)0; }
__device__ __forceinline__ void prof_end(prof_t *buf, prof_t t0) {
    if (buf) {
        if (threadIdx.x == 0) buf[2 * (size_t)prof_wg()] = t0;
        if ((threadIdx.x & 63) == 0) atomicMax(&buf[2 * (size_t)prof_wg() + 1], (prof_t)wall_clock64());
    }
}

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace tl
