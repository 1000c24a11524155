// RMSNorm, RoPE, SwiGLU and quantized-embedding gather for gfx950.
// These are HBM/latency-trivial ops; they exist so the reference operator API
// (week2_kernels.cpp:36-63, quantized_matmul.cpp:82-101) is complete on HIP.
// Layout rules used throughout: 16-byte vector accesses when the row length
// allows (guide G13), fp32 math inside, one rounding at the op boundary.
#include "common.h"

namespace tl {

// ---------------------------------------------------------------------------
// RMSNorm.  A row is owned by a TEAM of lanes (16 / 64 / 256); the row is read
// twice (second pass is an L1/L2 hit).  reference kernel: week2_kernels.metal:6-48
// ---------------------------------------------------------------------------
template <typename TT, int TEAM, int VEC>
__global__ __launch_bounds__(256) void rms_norm_kernel(const typename TT::storage *__restrict__ x,
                                                       const typename TT::storage *__restrict__ w,
                                                       typename TT::storage *__restrict__ out, int rows, int dim,
                                                       float eps) {
    using S = typename TT::storage;
    constexpr int TEAMS_PER_BLOCK = 256 / TEAM;
    const int team = threadIdx.x / TEAM;
    const int t = threadIdx.x % TEAM;
    const long row = (long)blockIdx.x * TEAMS_PER_BLOCK + team;
    __shared__ float partial[4];
    const bool live = row < rows;
    const S *xr = x + row * (long)dim;
    S *orow = out + row * (long)dim;
    float sum = 0.f;
    if (live) {
        for (int c = t * VEC; c < dim; c += TEAM * VEC) {
            S v[VEC];
            if constexpr (VEC > 1) {
                *reinterpret_cast<uint4 *>(v) = *reinterpret_cast<const uint4 *>(xr + c);
            } else {
                v[0] = xr[c];
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float f = TT::to_float(v[i]);
                sum += f * f;
            }
        }
    }
    // team reduction
    if constexpr (TEAM == 16) {
        sum = group16_sum(sum);
    } else if constexpr (TEAM == 64) {
        sum = wave_sum(sum);
    } else {
        sum = wave_sum(sum);
        if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = sum;
        __syncthreads();
        sum = partial[0] + partial[1] + partial[2] + partial[3];
    }
    if (!live) return;
    const float inv = rsqrtf(sum / (float)dim + eps);
    for (int c = t * VEC; c < dim; c += TEAM * VEC) {
        S v[VEC], g[VEC], o[VEC];
        if constexpr (VEC > 1) {
            *reinterpret_cast<uint4 *>(v) = *reinterpret_cast<const uint4 *>(xr + c);
            *reinterpret_cast<uint4 *>(g) = *reinterpret_cast<const uint4 *>(w + c);
        } else {
            v[0] = xr[c];
            g[0] = w[c];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] = TT::from_float(TT::to_float(v[i]) * inv * TT::to_float(g[i]));
        if constexpr (VEC > 1) {
            *reinterpret_cast<uint4 *>(orow + c) = *reinterpret_cast<const uint4 *>(o);
        } else {
            orow[c] = o[0];
        }
    }
}

template <typename TT>
static int launch_rms_norm(const void *x, const void *w, void *out, int rows, int dim, float eps, hipStream_t st) {
    using S = typename TT::storage;
    constexpr int VECW = 16 / sizeof(S);
    const bool vec_ok = (dim % VECW == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) &&
                        ((uintptr_t)out % 16 == 0);
    auto xs = (const S *)x;
    auto ws = (const S *)w;
    auto os = (S *)out;
#define RN_LAUNCH(TEAM, VEC)                                                                                  \
    hipLaunchKernelGGL((rms_norm_kernel<TT, TEAM, VEC>), dim3(ceil_div(rows, 256 / TEAM)), dim3(256), 0, st, \
                       xs, ws, os, rows, dim, eps)
    if (vec_ok) {
        if (dim <= 16 * VECW * 2) {
            RN_LAUNCH(16, VECW);
        } else if (dim <= 64 * VECW * 4) {
            RN_LAUNCH(64, VECW);
        } else {
            RN_LAUNCH(256, VECW);
        }
    } else {
        if (dim <= 64) {
            RN_LAUNCH(16, 1);
        } else if (dim <= 1024) {
            RN_LAUNCH(64, 1);
        } else {
            RN_LAUNCH(256, 1);
        }
    }
#undef RN_LAUNCH
    return 0;
}

// ---------------------------------------------------------------------------
// RoPE.  One thread per (b, l, head-block of 4, item); item < dims/2 is a
// rotated pair, item >= dims/2 copies one tail element.  Trig is evaluated
// once per thread and reused by its heads.  reference: week2_kernels.metal:50-105
// ---------------------------------------------------------------------------
template <typename TT>
__global__ __launch_bounds__(256) void rope_kernel(const typename TT::storage *__restrict__ x,
                                                   const int32_t *__restrict__ offsets,
                                                   typename TT::storage *__restrict__ out, int B, int L, int H, int D,
                                                   int dims, float base, int traditional) {
    constexpr int HPT = 4;
    const int half = dims / 2;
    const int tail = D - dims;
    const int items = half + tail;
    const int hblocks = (H + HPT - 1) / HPT;
    const long total = (long)B * L * hblocks * items;
    const long index = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= total) return;
    const int item = (int)(index % items);
    const int hb = (int)((index / items) % hblocks);
    const int l = (int)((index / ((long)items * hblocks)) % L);
    const int b = (int)(index / ((long)items * hblocks * L));
    const int h0 = hb * HPT;
    const int h1 = min(h0 + HPT, H);
    const long row = ((long)b * L + l) * H * D;
    if (item >= half) {
        const int d = dims + item - half;
        for (int h = h0; h < h1; ++h) out[row + (long)h * D + d] = x[row + (long)h * D + d];
        return;
    }
    const float fp = -(float)item / (float)half;
    const float pos = (float)(offsets[b] + l);
    float angle;
    if constexpr (sizeof(typename TT::storage) == 4) {
        angle = pos * powf(base, fp);
    } else {
        angle = pos * exp2f(fp * log2f(base));
    }
    float s, c;
    sincosf(angle, &s, &c);
    for (int h = h0; h < h1; ++h) {
        const long hbase = row + (long)h * D;
        const long ri = traditional ? hbase + 2 * item : hbase + item;
        const long ii = traditional ? ri + 1 : ri + half;
        const float re = TT::to_float(x[ri]);
        const float im = TT::to_float(x[ii]);
        out[ri] = TT::from_float(re * c - im * s);
        out[ii] = TT::from_float(im * c + re * s);
    }
}

// ---------------------------------------------------------------------------
// SwiGLU: out = T(silu(gate) * up), fp32 inside.  week2_kernels.metal:107-117
// ---------------------------------------------------------------------------
template <typename TT, int VEC>
__global__ __launch_bounds__(256) void swiglu_kernel(const typename TT::storage *__restrict__ gate,
                                                     const typename TT::storage *__restrict__ up,
                                                     typename TT::storage *__restrict__ out, size_t size) {
    using S = typename TT::storage;
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    const size_t stride = (size_t)gridDim.x * blockDim.x * VEC;
    for (; i < size; i += stride) {
        S g[VEC], u[VEC], o[VEC];
        if constexpr (VEC > 1) {
            *reinterpret_cast<uint4 *>(g) = *reinterpret_cast<const uint4 *>(gate + i);
            *reinterpret_cast<uint4 *>(u) = *reinterpret_cast<const uint4 *>(up + i);
        } else {
            g[0] = gate[i];
            u[0] = up[i];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float gf = TT::to_float(g[k]);
            o[k] = TT::from_float((gf / (1.0f + expf(-gf))) * TT::to_float(u[k]));
        }
        if constexpr (VEC > 1) {
            *reinterpret_cast<uint4 *>(out + i) = *reinterpret_cast<const uint4 *>(o);
        } else {
            out[i] = o[0];
        }
    }
}

template <typename TT>
static int launch_swiglu(const void *gate, const void *up, void *out, size_t size, hipStream_t st) {
    using S = typename TT::storage;
    constexpr int VECW = 16 / sizeof(S);
    const bool vec_ok = (size % VECW == 0) && ((uintptr_t)gate % 16 == 0) && ((uintptr_t)up % 16 == 0) &&
                        ((uintptr_t)out % 16 == 0);
    if (vec_ok) {
        size_t work = size / VECW;
        int blocks = (int)std::min<size_t>((work + 255) / 256, 4096);
        hipLaunchKernelGGL((swiglu_kernel<TT, VECW>), dim3(blocks), dim3(256), 0, st, (const S *)gate, (const S *)up,
                           (S *)out, size);
    } else {
        int blocks = (int)std::min<size_t>((size + 255) / 256, 4096);
        hipLaunchKernelGGL((swiglu_kernel<TT, 1>), dim3(blocks), dim3(256), 0, st, (const S *)gate, (const S *)up,
                           (S *)out, size);
    }
    return 0;
}

// ---------------------------------------------------------------------------
// Quantized embedding gather: one thread per packed word (8 outputs, 16 B store).
// reference: quantized_matmul.metal:58-89
// ---------------------------------------------------------------------------
template <typename TT, typename IndexT>
__global__ __launch_bounds__(256) void qembed_kernel(const IndexT *__restrict__ indices,
                                                     const typename TT::storage *__restrict__ scales,
                                                     const typename TT::storage *__restrict__ biases,
                                                     const uint32_t *__restrict__ weight,
                                                     typename TT::storage *__restrict__ out, int tokens, int dim,
                                                     int vocab) {
    using S = typename TT::storage;
    const int words = dim / 8;
    const long index = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= (long)tokens * words) return;
    const int token = (int)(index / words);
    const int word = (int)(index - (long)token * words);
    long row = (long)indices[token];
    // Out-of-range ids are clamped instead of faulting; the host wrapper validates.
    row = row < 0 ? 0 : (row >= vocab ? vocab - 1 : row);
    const uint32_t packed = weight[row * words + word];
    const int groups = dim / 128;
    const float scale = TT::to_float(scales[row * groups + word / 16]);
    const float bias = TT::to_float(biases[row * groups + word / 16]);
    S o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = TT::from_float((float)((packed >> (4 * i)) & 0xfu) * scale + bias);
    *reinterpret_cast<uint4 *>(out + (long)token * dim + word * 8) = *reinterpret_cast<const uint4 *>(o);
}

}  // namespace tl

using namespace tl;

extern "C" int tl_rms_norm(const void *x, const void *weight, void *out, int rows, int dim, float eps, tl_dtype dtype,
                           void *stream) {
    TL_REQUIRE(x && weight && out, "rms_norm: null pointer");
    TL_REQUIRE(rows >= 0 && dim > 0, "rms_norm: weight must match the input dtype and final dimension");
    if (rows == 0) return TL_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case TL_F32: launch_rms_norm<F32>(x, weight, out, rows, dim, eps, st); break;
        case TL_F16: launch_rms_norm<F16>(x, weight, out, rows, dim, eps, st); break;
        case TL_BF16: launch_rms_norm<BF16>(x, weight, out, rows, dim, eps, st); break;
        default: return fail(TL_ERR_INVALID, "rms_norm: expected float32, float16, or bfloat16");
    }
    TL_CHECK_LAUNCH("rms_norm");
    return TL_OK;
}

extern "C" int tl_rope(const void *x, const int32_t *offsets, void *out, int B, int L, int H, int D, int dims,
                       float base, int traditional, tl_dtype dtype, void *stream) {
    TL_REQUIRE(x && offsets && out, "rope: null pointer");
    TL_REQUIRE(B >= 0 && L >= 0 && H >= 0 && D > 0, "rope: expected x=[B,L,H,D] and one int32 offset per batch row");
    TL_REQUIRE(dims > 0 && dims <= D && dims % 2 == 0,
               "rope: dims must be positive, even, and no larger than the head dimension");
    const int items = dims / 2 + (D - dims);
    const long total = (long)B * L * ((H + 3) / 4) * items;
    if (total == 0) return TL_OK;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = ceil_div(total, 256);
#define ROPE_LAUNCH(TT)                                                                                          \
    hipLaunchKernelGGL((rope_kernel<TT>), dim3(blocks), dim3(256), 0, st, (const typename TT::storage *)x, offsets, \
                       (typename TT::storage *)out, B, L, H, D, dims, base, traditional)
    switch (dtype) {
        case TL_F32: ROPE_LAUNCH(F32); break;
        case TL_F16: ROPE_LAUNCH(F16); break;
        case TL_BF16: ROPE_LAUNCH(BF16); break;
        default: return fail(TL_ERR_INVALID, "rope: expected float32, float16, or bfloat16");
    }
#undef ROPE_LAUNCH
    TL_CHECK_LAUNCH("rope");
    return TL_OK;
}

extern "C" int tl_swiglu(const void *gate, const void *up, void *out, size_t size, tl_dtype dtype, void *stream) {
    TL_REQUIRE(gate && up && out, "swiglu: null pointer");
    if (size == 0) return TL_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case TL_F32: launch_swiglu<F32>(gate, up, out, size, st); break;
        case TL_F16: launch_swiglu<F16>(gate, up, out, size, st); break;
        case TL_BF16: launch_swiglu<BF16>(gate, up, out, size, st); break;
        default: return fail(TL_ERR_INVALID, "swiglu: expected float32, float16, or bfloat16");
    }
    TL_CHECK_LAUNCH("swiglu");
    return TL_OK;
}

extern "C" int tl_quantized_embedding(const void *indices, int indices_unsigned, const void *scales,
                                      const void *biases, const uint32_t *weight, void *out, int tokens, int dim,
                                      int vocab, int group_size, int bits, tl_dtype dtype, void *stream) {
    TL_REQUIRE(indices && scales && biases && weight && out, "quantized_embedding: null pointer");
    TL_REQUIRE(group_size == 128 && bits == 4, "quantized_embedding: expected 4-bit weights with group size 128");
    TL_REQUIRE(dim > 0 && dim % 128 == 0 && vocab > 0, "quantized_embedding: incompatible parameter shapes");
    TL_REQUIRE(dtype == TL_F16 || dtype == TL_BF16,
               "quantized_embedding: scales and biases must have the same 16-bit dtype");
    if (tokens == 0) return TL_OK;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = ceil_div((long)tokens * (dim / 8), 256);
#define QE_LAUNCH(TT, IT)                                                                                     \
    hipLaunchKernelGGL((qembed_kernel<TT, IT>), dim3(blocks), dim3(256), 0, st, (const IT *)indices,          \
                       (const uint16_t *)scales, (const uint16_t *)biases, weight, (uint16_t *)out, tokens, dim, \
                       vocab)
    if (dtype == TL_F16) {
        if (indices_unsigned) QE_LAUNCH(F16, uint32_t); else QE_LAUNCH(F16, int32_t);
    } else {
        if (indices_unsigned) QE_LAUNCH(BF16, uint32_t); else QE_LAUNCH(BF16, int32_t);
    }
#undef QE_LAUNCH
    TL_CHECK_LAUNCH("quantized_embedding");
    return TL_OK;
}
