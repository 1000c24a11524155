// W4A16 (group 128) matmul operator: dispatch + the non-GEMV kernels.
//   reference dispatch: quantized_matmul.cpp:111-240
//   * use_simdgroup && M<=8        -> qmv_kernel (qmv.h)                       [decode]
//   * use_simdgroup (&& split-K)   -> qmm_mfma_kernel (bf16/f16 MFMA 32x32x16) [prefill]
//   * otherwise                    -> qmm_vanilla_kernel (semantic definition)
#include "common.h"
#include "qmv.h"

namespace tl {

// ---------------------------------------------------------------------------
// One thread per output element, the reference's order of operations
// (quantized_matmul.metal:8-56): sum += (q*scale + bias) * a, fp32, one cast.
// ---------------------------------------------------------------------------
template <typename TT>
__global__ __launch_bounds__(256) void qmm_vanilla_kernel(const uint16_t *__restrict__ scales,
                                                          const uint16_t *__restrict__ biases,
                                                          const uint16_t *__restrict__ a,
                                                          const uint32_t *__restrict__ b, uint16_t *__restrict__ out,
                                                          int M, int N, int K) {
    // x = output column k (fast, so weight rows differ per lane and activations broadcast)
    const int k = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= M || k >= K) return;
    const int G = N / 128;
    const uint32_t *bw = b + (size_t)k * (N / 8);
    const uint16_t *ar = a + (size_t)i * N;
    float sum = 0.f;
    for (int g = 0; g < G; ++g) {
        const float scale = TT::to_float(scales[(size_t)k * G + g]);
        const float bias = TT::to_float(biases[(size_t)k * G + g]);
        for (int w = 0; w < 16; ++w) {
            const uint32_t packed = bw[g * 16 + w];
            const uint4 av = *reinterpret_cast<const uint4 *>(ar + g * 128 + w * 8);
            const uint32_t aw[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float q = (float)((packed >> (4 * e)) & 0xfu);
                const uint16_t ae = (uint16_t)((aw[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
                sum += (q * scale + bias) * TT::to_float(ae);
            }
        }
    }
    out[(size_t)i * K + k] = TT::from_float(sum);
}

// ---------------------------------------------------------------------------
// Prefill GEMM on MFMA.  Semantics follow the reference tile kernel
// (quantized_matmul.metal:96-249): weights are dequantised to T (one rounding),
// multiplied on the matrix unit with fp32 accumulation, output cast to T.
//
// gfx950 mapping (v_mfma_f32_32x32x16_{bf16,f16}):
//   workgroup = 4 waves; tile = (32*MT activation rows) x (128 output features);
//   wave w owns output features [32w, 32w+32) for all MT row tiles.
//   B operand (W^T): a lane's fragment is 8 consecutive reduction elements of
//   ONE weight row = exactly one packed uint32, so weights go global -> VGPR ->
//   dequant -> MFMA with no LDS round trip.  A lane pulls 16 B (4 words) per
//   64-wide super-step; the reduction index inside a super-step is permuted
//   (n = 64j + 32h + 8s + i for lane-half h, step s) so those 4 words are
//   contiguous -- the activation fragment uses the same permutation.
//   A operand (activations): staged once per workgroup into a double-buffered,
//   XOR-swizzled LDS tile and read back with ds_read_b128.
// ---------------------------------------------------------------------------
template <typename TT>
struct Mfma;
template <>
struct Mfma<BF16> {
    __device__ __forceinline__ static f32x16 mma(const u32x4 &a, const u32x4 &b, const f32x16 &c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                       c, 0, 0, 0);
    }
};
template <>
struct Mfma<F16> {
    __device__ __forceinline__ static f32x16 mma(const u32x4 &a, const u32x4 &b, const f32x16 &c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b),
                                                      c, 0, 0, 0);
    }
};

template <typename TT>
__device__ __forceinline__ u32x4 dequant_word(uint32_t w, float s, float beta) {
    // even nibbles as the bytes of one word, odd nibbles as the bytes of another: each of the 8 conversions is then a single
    // v_cvt_f32_ubyteN (3 mask / shift instructions per word instead of 2 per nibble; the counters showed 11 VALU per MFMA)
    uint32_t even = w & 0x0f0f0f0fu, odd = (w >> 4) & 0x0f0f0f0fu;
    asm volatile("" : "+v"(even), "+v"(odd));  // or the optimiser folds the masks back into per-nibble shift + and
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lo = (float)((even >> (8 * e)) & 0xffu) * s + beta;
        const float hi = (float)((odd >> (8 * e)) & 0xffu) * s + beta;
        r[e] = TT::pack2(lo, hi);
    }
    return r;
}

#ifndef QMM_ABL
#define QMM_ABL 0  // tools/lab/gemm_lab only: 1 = no dequant arithmetic, 2 = no MFMA, 4 = no activation re-staging, 8 = the next
                   // step's requests outside the branch (clamped step index: the last step requests itself again)
#endif
#ifndef QMM_OCC
#define QMM_OCC 3  // waves per SIMD the register allocation aims for (136 registers fit 3; 4 costs 5 spilled VGPRs)
#endif
#ifndef QMM_OCC8
#define QMM_OCC8 4  // the same for the 8-wave (256-column) tile: two workgroups per CU
#endif
// NW = waves per workgroup: the tile is (32*MT) x (32*NW).  Every byte the kernel moves comes through the CU's memory pipe
// (~25-28 GB/s per CU when all 256 stream at once: r02 lab), and the activation tile re-read by every column tile is most of
// it: bytes per flop = 1/TN + 0.25/TM (bf16 activations, 4-bit weights) -- 128 x 128: 734 TFLOP/s at 28 GB/s per CU, close to
// what the kernel measures.  128 x 256 halves the activation traffic; measured it gains 10 % on gate|up and loses elsewhere
// (mfma_nw below).
// EPI (engine prefill only; the operator uses EPI_STORE): applied by the kernel when the reduction is not split, by the split-K
// reduction otherwise -- on the SAME bf16-rounded matmul result and with the same expressions as the separate residual /
// SwiGLU launches they replace (residual_add_kernel, swiglu_interleaved_kernel), so the results are bit-identical.
template <typename TT, int MT, int NW = 4, int EPI = EPI_STORE>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? QMM_OCC8 : QMM_OCC)) void qmm_mfma_kernel(const uint16_t *__restrict__ scales,
                                                       const uint16_t *__restrict__ biases,
                                                       const uint16_t *__restrict__ a, const uint32_t *__restrict__ b,
                                                       uint16_t *__restrict__ out, int M, int N, int K,
                                                       int partition_size, size_t partition_stride,
                                                       const uint16_t *__restrict__ residual = nullptr, int xcd_remap = 0) {
    constexpr int NT = 64 * NW;
    constexpr int AQ = 32 * MT * 8 / NT;  // 16-byte activation chunks per thread and 64-wide reduction step
    static_assert(AQ >= 1 && AQ * NT == 32 * MT * 8, "the activation tile must divide evenly over the threads");
    constexpr int BM = 32 * MT;
    __shared__ __attribute__((aligned(16))) uint16_t atile[2][BM * 64];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int l32 = lane & 31;
    const int h = lane >> 5;
    // XCD-aware tile order (xcd_remap): workgroups go to the 8 XCDs round-robin in launch order and every XCD has its own 4 MiB L2.
    // In launch order the column tiles of one row band are dealt over all 8 L2s and every L2 sees every activation row; remapped,
    // the workgroups an XCD receives form one contiguous range of a band-major order (bands of gridDim.y / 8 row tiles, column
    // by column inside a band): an XCD keeps a 256-row band of a 2,048-row chunk (1.3 MiB of activations) in its L2 and the weight
    // tile of a column is fetched once per band.  A bijection of the tiles for any grid; only locality relies on the deal order.
    int tx = blockIdx.x, ty = blockIdx.y;
    if (xcd_remap) {
        const int gx = gridDim.x, gy = gridDim.y, T = gx * gy;
        const int l = blockIdx.x + gx * blockIdx.y;  // launch order inside this reduction slice: every 8th workgroup shares an XCD
        const int c = l & 7;
        const int t = c * (T >> 3) + min(c, T & 7) + (l >> 3);  // class c takes one contiguous range of the band-major order
        const int R = max(1, gy >> 3);
        const int band = t / (R * gx), rem = t - band * (R * gx);
        const int rows = min(R, gy - band * R);
        tx = rem / rows;
        ty = band * R + (rem - tx * rows);
    }
    const int bn0 = tx * (32 * NW);
    const int bm0 = ty * BM;
    const int red0 = blockIdx.z * partition_size;
    const int j0 = red0 >> 6;
    const int j1 = (red0 + partition_size) >> 6;
    const int G = N >> 7;
    const int words = N >> 3;

    const int wrow = bn0 + wave * 32 + l32;  // weight row (= output feature) of this lane
    const bool wok = wrow < K;
    const uint32_t *wsrc = b + (size_t)(wok ? wrow : 0) * words;
    const uint16_t *ssrc = scales + (size_t)(wok ? wrow : 0) * G;
    const uint16_t *bsrc = biases + (size_t)(wok ? wrow : 0) * G;

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    u32x4 areg[AQ];
    auto load_a = [&](int j) {
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int c = tid + q * NT;
            const int r = c >> 3;
            const int ch = c & 7;
            // rows past M re-read the last row instead of being zeroed under a branch (their outputs are never stored): the
            // branch cost an exec-mask dance and four zero moves per chunk in every reduction step
            const int gr = min(bm0 + r, M - 1);
            areg[q] = *reinterpret_cast<const u32x4 *>(a + (size_t)gr * N + j * 64 + ch * 8);
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int c = tid + q * NT;
            const int r = c >> 3;
            const int ch = c & 7;
            const int sw = ch ^ ((r >> 1) & 7);
            *reinterpret_cast<u32x4 *>(&atile[buf][r * 64 + sw * 8]) = areg[q];
        }
    };

    u32x4 wcur = u32x4{0u, 0u, 0u, 0u}, wnext = wcur;
    float sc = 0.f, be = 0.f;
    // Scale / bias of the NEXT 64-wide step are requested together with its weights and consumed behind this step's MFMAs.
    // Read where they are used, right after the prefetch requests, they made every other step wait out its own prefetch
    // (loads return in issue order): r02 lab, 2,048 rows: qkv 753 -> 838, gate|up 762 -> 838 TFLOP/s.  Also tried there and
    // dropped: activation tiles two steps ahead (two register sets: slower), an explicit sub-step pipeline of fragment reads
    // and dequantisation (no gain), two 32-column blocks per wave (128 x 256 tile on four waves: no gain at two waves per SIMD),
    // eight row tiles per wave (256 x 128: the counters say 11 VALU instructions per MFMA, and a dequantised fragment then
    // feeds 8 MFMAs -- but at two waves per SIMD the layer got 5 % slower).  (32-bit holders: as
    // 16-bit values the compiler packs the pair into one register right behind the loads -- the same wait again.)
    uint32_t sc_next = 0, be_next = 0;
    if (wok) {
        sc = TT::to_float(ssrc[j0 >> 1]);
        be = TT::to_float(bsrc[j0 >> 1]);
    }
    load_a(j0);
    if (wok) wcur = *reinterpret_cast<const u32x4 *>(wsrc + j0 * 8 + h * 4);
    int buf = 0;
    for (int j = j0; j < j1; ++j) {
        if (!(QMM_ABL & 4) || j == j0) store_a(buf);
        __syncthreads();
        if ((QMM_ABL & 8) || j + 1 < j1) {
            const int jn = (QMM_ABL & 8) ? min(j + 1, j1 - 1) : j + 1;
            if (!(QMM_ABL & 4)) load_a(jn);
            // (columns past K read weight row 0 -- wsrc / ssrc / bsrc are clamped -- and are never stored)
            wnext = *reinterpret_cast<const u32x4 *>(wsrc + jn * 8 + h * 4);
            sc_next = ssrc[jn >> 1];
            be_next = bsrc[jn >> 1];
        }
        u32x4 bf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (QMM_ABL & 1) bf[s] = u32x4{wcur[s], wcur[s] ^ 0x3c003c00u, wcur[s] >> 1, __float_as_uint(sc + be)};
            else bf[s] = dequant_word<TT>(wcur[s], sc, be);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int r = mt * 32 + l32;
                const int ch = (4 * h + s) ^ ((r >> 1) & 7);
                const u32x4 af = *reinterpret_cast<const u32x4 *>(&atile[buf][r * 64 + ch * 8]);
                if constexpr (QMM_ABL & 2) acc[mt][s] += __uint_as_float((af[0] ^ bf[s][1]) & 0x3f800000u);
                else acc[mt] = Mfma<TT>::mma(af, bf[s], acc[mt]);
            }
        }
        buf ^= 1;
        __builtin_amdgcn_sched_barrier(0);
        wcur = wnext;
        sc = TT::to_float((uint16_t)sc_next);
        be = TT::to_float((uint16_t)be_next);
    }

    if constexpr (EPI == EPI_SWIGLU) {
        // weight rows interleaved (even = gate_i, odd = up_i): the partner column sits in the neighbouring lane
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = bm0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float mine = TT::to_float(TT::from_float(acc[mt][r]));
                const float other = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mine), 0xb1, 0xf, 0xf, false));  // quad_perm [1,0,3,2]
                if (wok && (l32 & 1) == 0 && m < M) out[(size_t)m * (K / 2) + (wrow >> 1)] = TT::from_float((mine / (1.0f + expf(-mine))) * other);
            }
        }
        return;
    }
    if (!wok) return;
    uint16_t *dst = out + (size_t)blockIdx.z * partition_stride;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = bm0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < M) {
                if constexpr (EPI == EPI_RESIDUAL)
                    dst[(size_t)m * K + wrow] = TT::from_float(TT::to_float(residual[(size_t)m * K + wrow]) + TT::to_float(TT::from_float(acc[mt][r])));
                else
                    dst[(size_t)m * K + wrow] = TT::from_float(acc[mt][r]);
            }
        }
    }
}

// partials [split_k][M*K] in T -> fp32 sum -> T   (quantized_matmul.metal:277-293)
// One thread = 8 consecutive elements (one 16-byte load per slice), the loads of 8 slices in flight together, the slices added in index
// order per element -- the same sums in the same order as one thread per element, so the results are unchanged; what changed is the time:
// the reduction behind a 128-row chunk cost 6.4-8.2 us with 2- / 4-byte loads in a loop of unknown length (round 5, profiles/r05_labs).
template <typename TT>
__device__ __forceinline__ void splitk_sum8(const uint16_t *__restrict__ partials, size_t elements, size_t i, int split_k, float (&sum)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) sum[e] = 0.f;
    for (int p0 = 0; p0 < split_k; p0 += 8) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const u32x4 *>(partials + (size_t)min(p0 + j, split_k - 1) * elements + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (p0 + j < split_k) {  // uniform; no load inside
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sum[2 * e] += TT::to_float((uint16_t)(v[j][e] & 0xffffu));
                    sum[2 * e + 1] += TT::to_float((uint16_t)(v[j][e] >> 16));
                }
            }
        }
    }
}
template <typename TT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const uint16_t *__restrict__ partials,
                                                            uint16_t *__restrict__ out, size_t elements, int split_k) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= elements) return;
    if (i + 8 <= elements && (elements & 7) == 0 && (((uintptr_t)out | (uintptr_t)partials) & 15) == 0) {  // (uniform but for the last thread)
        float sum[8];
        splitk_sum8<TT>(partials, elements, i, split_k, sum);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (uint32_t)TT::from_float(sum[2 * e]) | ((uint32_t)TT::from_float(sum[2 * e + 1]) << 16);
        *reinterpret_cast<u32x4 *>(out + i) = o;
        return;
    }
    for (size_t k = i; k < elements && k < i + 8; ++k) {  // a ragged element count: one element at a time
        float sum = 0.f;
        for (int p = 0; p < split_k; ++p) sum += TT::to_float(partials[(size_t)p * elements + k]);
        out[k] = TT::from_float(sum);
    }
}
// the same with the engine's epilogue on the rounded sum (elements % 8 == 0: the engine's shapes); adjacent elements are a gate / up pair under SwiGLU
template <typename TT, int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_epi_kernel(const uint16_t *__restrict__ partials, uint16_t *__restrict__ out,
                                                                size_t elements, int split_k, const uint16_t *__restrict__ residual) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= elements) return;
    u32x4 rv = u32x4{0u, 0u, 0u, 0u};
    if constexpr (EPI != EPI_SWIGLU) rv = *reinterpret_cast<const u32x4 *>(residual + i);  // goes out with the first slices
    float sum[8];
    splitk_sum8<TT>(partials, elements, i, split_k, sum);
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = TT::to_float(TT::from_float(sum[e]));
    if constexpr (EPI == EPI_SWIGLU) {
        u32x2 o;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint16_t a = TT::from_float((r[4 * e] / (1.0f + expf(-r[4 * e]))) * r[4 * e + 1]);
            const uint16_t b = TT::from_float((r[4 * e + 2] / (1.0f + expf(-r[4 * e + 2]))) * r[4 * e + 3]);
            o[e] = (uint32_t)a | ((uint32_t)b << 16);
        }
        *reinterpret_cast<u32x2 *>(out + (i >> 1)) = o;
    } else {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint16_t o0 = TT::from_float(TT::to_float((uint16_t)(rv[e] & 0xffffu)) + r[2 * e]);
            const uint16_t o1 = TT::from_float(TT::to_float((uint16_t)(rv[e] >> 16)) + r[2 * e + 1]);
            o[e] = (uint32_t)o0 | ((uint32_t)o1 << 16);
        }
        *reinterpret_cast<u32x4 *>(out + i) = o;
    }
}

// The residual reduction of a split-K projection AND the RMSNorm of the row it completes, one launch (round 6, late): at small prefill chunks the reduction
// pass (5 us) was followed by rms_norm_kernel (5 us) re-reading the rows it had just written -- 2 of a layer's 14 launches at a 128-token chunk.  One
// workgroup per row, one thread per 8-element chunk (dim / 8 threads: 2,049 .. 4,096 features, the range in which tl_rms_norm runs rms_norm_kernel<TT, 256, 8>):
// the row's elements are formed exactly as splitk_reduce_epi_kernel<EPI_RESIDUAL> forms them (slices added in index order, rounded, + residual, rounded),
// stored, and kept in registers.  The sum of squares follows rms_norm_kernel's ORDER: its thread u adds the squares of chunk u, then of chunk u + 256, to one
// accumulator -- so the threads of chunks 256 .. hand their values to thread u through LDS -- then wave_sum and the four wave partials in index order; the
// normalised row is its expression on the same values.  Both outputs are bit-identical to the two launches.  Slices are requested 16 at a time.
template <typename TT>
__global__ __launch_bounds__(512) void splitk_reduce_residual_norm_kernel(const uint16_t *__restrict__ partials, uint16_t *__restrict__ out, size_t elements,
                                                                          int split_k, const uint16_t *__restrict__ residual,
                                                                          const uint16_t *__restrict__ norm_w, uint16_t *__restrict__ out_norm, int dim, float eps) {
    __shared__ float hand[256][8];
    __shared__ float partial[4];
    const int t = threadIdx.x;
    const int chunks = dim >> 3;  // 257 .. 512
    const bool live = t < chunks;
    const size_t i = (size_t)blockIdx.x * dim + (size_t)(live ? t : 0) * 8;
    const u32x4 rv = *reinterpret_cast<const u32x4 *>(residual + i);
    const u32x4 g = *reinterpret_cast<const u32x4 *>(norm_w + (live ? t : 0) * 8);
    float sum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sum[e] = 0.f;
    for (int p0 = 0; p0 < split_k; p0 += 16) {
        u32x4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const u32x4 *>(partials + (size_t)min(p0 + j, split_k - 1) * elements + i);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (p0 + j < split_k) {  // uniform; no load inside
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sum[2 * e] += TT::to_float((uint16_t)(v[j][e] & 0xffffu));
                    sum[2 * e + 1] += TT::to_float((uint16_t)(v[j][e] >> 16));
                }
            }
        }
    }
    float f[8];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint16_t o0 = TT::from_float(TT::to_float((uint16_t)(rv[e] & 0xffffu)) + TT::to_float(TT::from_float(sum[2 * e])));
        const uint16_t o1 = TT::from_float(TT::to_float((uint16_t)(rv[e] >> 16)) + TT::to_float(TT::from_float(sum[2 * e + 1])));
        o[e] = (uint32_t)o0 | ((uint32_t)o1 << 16);
        f[2 * e] = TT::to_float(o0);
        f[2 * e + 1] = TT::to_float(o1);
    }
    if (live) *reinterpret_cast<u32x4 *>(out + i) = o;
    if (live && t >= 256) {
#pragma unroll
        for (int e = 0; e < 8; ++e) hand[t - 256][e] = f[e];
    }
    __syncthreads();
    float sum_sq = 0.f;
    if (t < 256) {
#pragma unroll
        for (int e = 0; e < 8; ++e) sum_sq += f[e] * f[e];
        if (t + 256 < chunks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float h = hand[t][e];
                sum_sq += h * h;
            }
        }
        sum_sq = wave_sum(sum_sq);
        if ((t & 63) == 0) partial[t >> 6] = sum_sq;
    }
    __syncthreads();
    sum_sq = partial[0] + partial[1] + partial[2] + partial[3];
    const float inv = rsqrtf(sum_sq / (float)dim + eps);
    if (live) {
        u32x4 n;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint16_t o0 = TT::from_float(f[2 * e] * inv * TT::to_float((uint16_t)(g[e] & 0xffffu)));
            const uint16_t o1 = TT::from_float(f[2 * e + 1] * inv * TT::to_float((uint16_t)(g[e] >> 16)));
            n[e] = (uint32_t)o0 | ((uint32_t)o1 << 16);
        }
        *reinterpret_cast<u32x4 *>(out_norm + i) = n;
    }
}

// XCD-aware tile order of qmm_mfma_kernel (round 3 A/B: 731 -> 752 TFLOP/s on a 2,048-row layer; the launch-order switch is gone)
static int qmm_xcd_remap() { return 1; }
static int mfma_mt(int M) { return M <= 32 ? 1 : (M <= 64 ? 2 : 4); }
// 8 waves (a 128 x 256 tile) for the widest projection at full chunks only -- r02 lab at 2,048 rows: gate|up 710 -> 781
// TFLOP/s, but qkv 767 -> 660 and the split-K projections (o, down) 531 / 596 -> 470 / 517, and everything slower at 512 rows
// (profiles/r02_labs/gemm_lab_r02_waves_per_workgroup.log): halving the activation re-reads is not the whole story, the
// 8-wave barrier per 64-wide reduction step costs about as much.
static int mfma_nw(int M, int K) { return (M >= 1024 && K >= 8192) ? 8 : 4; }

// Split-K policy.  Reference (quantized_matmul.cpp:138-151) targets 320
// threadgroups of 32x32 on an M4 Pro; here a tile is (32*MT)x128 and the
// target is two workgroups per CU on 256 CUs.
static int split_k_policy(int M, int N, int K) {
    const int mt = mfma_mt(M);
    const int tiles = ceil_div(M, 32 * mt) * ceil_div(K, 32 * mfma_nw(M, K));
    // three workgroups fit a CU (136 registers per lane): split until about 768 are in flight.  320 tiles (a 2048-row chunk
    // against the 2560-row o / down projections) measured 471 / 482 TFLOP/s unsplit against 750 for the wide projections.
    // (two of the 8-wave workgroups fit: 512)
    const int target = mfma_nw(M, K) == 8 ? 512 : 768;
    // (20 since round 5: the w_down reduction is 76 groups = 4 x 19 -- with 16 as the cap a 128- or 256-row chunk got 4 slices of 38 steps on 80
    // workgroups; 19 slices of 4 groups put 380 on the chip: 128-row chunks 24.8k -> see profiles/r05_labs/README.md section 8)
    constexpr int max_split = 20;
    // at least two quantisation groups per slice: a one-group slice is a K loop of 4 MFMA steps behind a full prologue and a
    // reduction pass (the reference's own fallback case -- 128 x 2560 over N = 256 -- must stay unsplit and bit-identical,
    // tests_refsol/test_week_2_day_7.py:80-109)
    int s = std::min(std::min(max_split, std::max(1, (target + tiles / 2) / std::max(tiles, 1))), N / 256);
    s = std::max(s, 1);
    while (s > 1 && N % (s * 128) != 0) --s;
    return std::max(s, 1);
}

template <typename TT>
static int launch_qmv(const QmvArgs &args, hipStream_t st) {
    const QmvPlan pl = qmv_plan(args.M, args.N, args.K);
    if (pl.lds > 150 * 1024) return -1;  // caller splits M
    const dim3 grid(pl.blocks), block(256);
#define QMV_CASE(MRv, WNv, RPLv)                                                                            \
    if (pl.MR == MRv && pl.WN == WNv && pl.RPL == RPLv) {                                                   \
        auto kern = qmv_kernel<TT, MRv, WNv, RPLv, PRO_NONE, EPI_STORE>;                                    \
        if (pl.lds > 64 * 1024)                                                                             \
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds); \
        hipLaunchKernelGGL(kern, grid, block, pl.lds, st, args);                                            \
        return 0;                                                                                           \
    }
#define QMV_MR(MRv) QMV_CASE(MRv, 1, 2) QMV_CASE(MRv, 1, 1) QMV_CASE(MRv, 2, 1) QMV_CASE(MRv, 4, 1)
    QMV_MR(1) QMV_MR(2) QMV_MR(4) QMV_MR(8)
#undef QMV_MR
#undef QMV_CASE
    return -2;
}

template <typename TT>
static int run_qmm(const void *scales, const void *biases, const void *a, const uint32_t *b, void *out, int M, int N,
                   int K, int use_simdgroup, int use_split_k, void *workspace, size_t workspace_bytes,
                   hipStream_t st) {
    auto S = (const uint16_t *)scales;
    auto Bi = (const uint16_t *)biases;
    auto A = (const uint16_t *)a;
    auto O = (uint16_t *)out;
    if (use_simdgroup && M <= 8) {
        // GEMV; if the activation tile would not fit in LDS, process the rows in halves.
        int m0 = 0;
        int step = M;
        while (qmv_plan(step, N, K).lds > 150 * 1024 && step > 1) step = (step + 1) / 2;
        for (; m0 < M; m0 += step) {
            QmvArgs args{};
            args.scales = S; args.biases = Bi; args.b = b;
            args.a = A + (size_t)m0 * N;
            args.out = O + (size_t)m0 * K;
            args.M = std::min(step, M - m0); args.N = N; args.K = K;
            const int rc = launch_qmv<TT>(args, st);
            if (rc != 0) return fail(TL_ERR_UNSUPPORTED, "quantized_matmul: no GEMV configuration for this shape");
        }
        return TL_OK;
    }
    if (use_simdgroup) {
        const int mt = mfma_mt(M);
        const int nw = mfma_nw(M, K);
        const int split = use_split_k ? split_k_policy(M, N, K) : 1;
        const dim3 grid(ceil_div(K, 32 * nw), ceil_div(M, 32 * mt), split), block(64 * nw);
        uint16_t *dst = O;
        if (split > 1) {
            const size_t need = (size_t)split * M * K * 2;
            if (!workspace || workspace_bytes < need)
                return fail(TL_ERR_INVALID, "quantized_matmul: split-K workspace is missing or too small");
            dst = (uint16_t *)workspace;
        }
        const int psize = N / split;
        const size_t pstride = (size_t)M * K;
        if (nw == 8) {
            hipLaunchKernelGGL((qmm_mfma_kernel<TT, 4, 8>), grid, block, 0, st, S, Bi, A, b, dst, M, N, K, psize, pstride, nullptr, qmm_xcd_remap());
        } else {
            switch (mt) {
                case 1: hipLaunchKernelGGL((qmm_mfma_kernel<TT, 1>), grid, block, 0, st, S, Bi, A, b, dst, M, N, K, psize, pstride, nullptr, qmm_xcd_remap()); break;
                case 2: hipLaunchKernelGGL((qmm_mfma_kernel<TT, 2>), grid, block, 0, st, S, Bi, A, b, dst, M, N, K, psize, pstride, nullptr, qmm_xcd_remap()); break;
                default: hipLaunchKernelGGL((qmm_mfma_kernel<TT, 4>), grid, block, 0, st, S, Bi, A, b, dst, M, N, K, psize, pstride, nullptr, qmm_xcd_remap()); break;
            }
        }
        if (split > 1) {
            const size_t elements = (size_t)M * K;
            hipLaunchKernelGGL((splitk_reduce_kernel<TT>), dim3(ceil_div(ceil_div(elements, 8), 256)), dim3(256), 0, st,
                               (const uint16_t *)workspace, O, elements, split);
        }
        return TL_OK;
    }
    hipLaunchKernelGGL((qmm_vanilla_kernel<TT>), dim3(ceil_div(K, 64), ceil_div(M, 4)), dim3(256), 0, st, S, Bi, A, b, O,
                       M, N, K);
    return TL_OK;
}

// The engine's prefill projection: the W4 MFMA GEMM of the operator (same kernels, same tile / split-K policy, therefore the same
// bits) with the residual add or the SwiGLU of the interleaved gate|up rows folded into the kernel's store (unsplit) or into the
// split-K reduction -- the separate elementwise launch and a round trip of the [M, K] intermediate through HBM are gone.
int qmm_bf16_epilogue(const void *scales, const void *biases, const uint16_t *A, const uint32_t *b, uint16_t *O, int M, int N, int K,
                      int epi, const uint16_t *residual, void *workspace, size_t workspace_bytes, hipStream_t st, const uint16_t *norm_w,
                      uint16_t *norm_out, float norm_eps, bool *norm_done) {
    if (norm_done) *norm_done = false;
    if (M <= 8 || epi == EPI_STORE) return fail(TL_ERR_INVALID, "qmm_bf16_epilogue: for more than 8 rows and a real epilogue");
    if (epi == EPI_RESIDUAL && !residual) return fail(TL_ERR_INVALID, "qmm_bf16_epilogue: residual rows missing");
    if (epi == EPI_SWIGLU && (K % 2) != 0) return fail(TL_ERR_INVALID, "qmm_bf16_epilogue: SwiGLU needs an even number of weight rows");
    auto S = (const uint16_t *)scales;
    auto Bi = (const uint16_t *)biases;
    const int mt = mfma_mt(M);
    const int nw = mfma_nw(M, K);
    const int split = split_k_policy(M, N, K);
    const dim3 grid(ceil_div(K, 32 * nw), ceil_div(M, 32 * mt), split), block(64 * nw);
    const int psize = N / split;
    const size_t pstride = (size_t)M * K;
    if (split > 1) {
        const size_t need = (size_t)split * M * K * 2;
        if (!workspace || workspace_bytes < need) return fail(TL_ERR_INVALID, "qmm_bf16_epilogue: split-K workspace is missing or too small");
        uint16_t *dst = (uint16_t *)workspace;
        if (nw == 8) hipLaunchKernelGGL((qmm_mfma_kernel<BF16, 4, 8>), grid, block, 0, st, S, Bi, A, b, dst, M, N, K, psize, pstride, nullptr, qmm_xcd_remap());
        else if (mt == 1) hipLaunchKernelGGL((qmm_mfma_kernel<BF16, 1>), grid, block, 0, st, S, Bi, A, b, dst, M, N, K, psize, pstride, nullptr, qmm_xcd_remap());
        else if (mt == 2) hipLaunchKernelGGL((qmm_mfma_kernel<BF16, 2>), grid, block, 0, st, S, Bi, A, b, dst, M, N, K, psize, pstride, nullptr, qmm_xcd_remap());
        else hipLaunchKernelGGL((qmm_mfma_kernel<BF16, 4>), grid, block, 0, st, S, Bi, A, b, dst, M, N, K, psize, pstride, nullptr, qmm_xcd_remap());
        const size_t elements = (size_t)M * K;
        if (elements % 8 != 0 || (((uintptr_t)O | (uintptr_t)dst | (uintptr_t)residual) & 15) != 0)
            return fail(TL_ERR_INVALID, "qmm_bf16_epilogue: rows x features must be a multiple of 8 and the rows 16-byte aligned");
        const dim3 rg(ceil_div(elements / 8, 256));
        if (epi == EPI_RESIDUAL && norm_w && norm_out && norm_done && K % 8 == 0 && K > 2048 && K <= 4096 && (((uintptr_t)norm_w | (uintptr_t)norm_out) & 15) == 0) {
            // the row's RMSNorm rides on its reduction (one workgroup per row): the caller skips its rms_norm launch
            hipLaunchKernelGGL((splitk_reduce_residual_norm_kernel<BF16>), dim3(M), dim3((K / 8 + 63) / 64 * 64), 0, st, dst, O, elements, split, residual, norm_w, norm_out, K, norm_eps);
            *norm_done = true;
            return TL_OK;
        }
        if (epi == EPI_SWIGLU) hipLaunchKernelGGL((splitk_reduce_epi_kernel<BF16, EPI_SWIGLU>), rg, dim3(256), 0, st, dst, O, elements, split, residual);
        else hipLaunchKernelGGL((splitk_reduce_epi_kernel<BF16, EPI_RESIDUAL>), rg, dim3(256), 0, st, dst, O, elements, split, residual);
        return TL_OK;
    }
#define QMM_EPI_LAUNCH(EPIv)                                                                                                     \
    if (nw == 8) hipLaunchKernelGGL((qmm_mfma_kernel<BF16, 4, 8, EPIv>), grid, block, 0, st, S, Bi, A, b, O, M, N, K, psize, pstride, residual, qmm_xcd_remap()); \
    else if (mt == 1) hipLaunchKernelGGL((qmm_mfma_kernel<BF16, 1, 4, EPIv>), grid, block, 0, st, S, Bi, A, b, O, M, N, K, psize, pstride, residual, qmm_xcd_remap()); \
    else if (mt == 2) hipLaunchKernelGGL((qmm_mfma_kernel<BF16, 2, 4, EPIv>), grid, block, 0, st, S, Bi, A, b, O, M, N, K, psize, pstride, residual, qmm_xcd_remap()); \
    else hipLaunchKernelGGL((qmm_mfma_kernel<BF16, 4, 4, EPIv>), grid, block, 0, st, S, Bi, A, b, O, M, N, K, psize, pstride, residual, qmm_xcd_remap());
    if (epi == EPI_SWIGLU) {
        QMM_EPI_LAUNCH(EPI_SWIGLU)
    } else {
        QMM_EPI_LAUNCH(EPI_RESIDUAL)
    }
#undef QMM_EPI_LAUNCH
    return TL_OK;
}

// one GEMV per activation row, each against the weights of its own expert (grid.y = rows)
template <typename TT>
static int run_gather_qmv(const void *scales, const void *biases, const void *a, const uint32_t *b, const int32_t *expert_ids,
                          void *out, int M, int N, int K, int num_experts, hipStream_t st, int a_rows_div = 1) {
    QmvArgs args{};
    args.scales = (const uint16_t *)scales;
    args.biases = (const uint16_t *)biases;
    args.a = (const uint16_t *)a;
    args.b = b;
    args.out = (uint16_t *)out;
    args.M = 1;
    args.N = N;
    args.K = K;
    args.expert_ids = expert_ids;
    args.num_experts = num_experts;
    args.a_rows_div = a_rows_div;
    const QmvPlan pl = qmv_plan(1, N, K);
    if (pl.lds > 150 * 1024) return fail(TL_ERR_UNSUPPORTED, "gather_quantized_matvec: reduction dimension too large");
    const dim3 grid(pl.blocks, M), block(256);
#define GQ_CASE(WNv, RPLv)                                                                                          \
    if (pl.WN == WNv && pl.RPL == RPLv) {                                                                           \
        auto kern = qmv_kernel<TT, 1, WNv, RPLv, PRO_NONE, EPI_STORE>;                                              \
        if (pl.lds > 64 * 1024)                                                                                     \
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds); \
        hipLaunchKernelGGL(kern, grid, block, pl.lds, st, args);                                                    \
        return TL_OK;                                                                                               \
    }
    GQ_CASE(1, 2) GQ_CASE(1, 1) GQ_CASE(2, 1) GQ_CASE(4, 1)
#undef GQ_CASE
    return fail(TL_ERR_UNSUPPORTED, "gather_quantized_matvec: no GEMV configuration for this shape");
}

int gather_qmv_bf16(const void *scales, const void *biases, const uint16_t *a, const uint32_t *b, const int32_t *expert_ids,
                    uint16_t *out, int M, int N, int K, int num_experts, int a_rows_div, hipStream_t st) {
    if (M > 65535) return fail(TL_ERR_INVALID, "grouped-expert matvec: at most 65535 rows per launch");
    if (M == 0 || K == 0) return TL_OK;
    return run_gather_qmv<BF16>(scales, biases, a, b, expert_ids, out, M, N, K, num_experts, st, a_rows_div);
}

}  // namespace tl

using namespace tl;

extern "C" int tl_gather_quantized_matvec(const void *scales, const void *biases, const void *a, const uint32_t *b,
                                          const int32_t *expert_ids, void *out, int M, int N, int K, int num_experts,
                                          int group_size, int bits, tl_dtype dtype, void *stream) {
    TL_REQUIRE(dtype == TL_F16 || dtype == TL_BF16, "gather_quantized_matvec: scales must be float16 or bfloat16");
    TL_REQUIRE(bits == 4, "gather_quantized_matvec: bits must be 4");
    TL_REQUIRE(group_size == 128, "gather_quantized_matvec: group_size must be 128");
    TL_REQUIRE(scales && biases && a && b && expert_ids && out, "gather_quantized_matvec: null pointer");
    TL_REQUIRE(M >= 0 && K >= 0 && N > 0 && num_experts > 0, "gather_quantized_matvec: bad shape");
    TL_REQUIRE(N % 128 == 0, "gather_quantized_matvec: N must be divisible by group_size");
    TL_REQUIRE(M <= 65535, "gather_quantized_matvec: at most 65535 rows per call");
    if (M == 0 || K == 0) return TL_OK;
    hipStream_t st = (hipStream_t)stream;
    const int rc = dtype == TL_F16 ? run_gather_qmv<F16>(scales, biases, a, b, expert_ids, out, M, N, K, num_experts, st)
                                   : run_gather_qmv<BF16>(scales, biases, a, b, expert_ids, out, M, N, K, num_experts, st);
    if (rc != TL_OK) return rc;
    TL_CHECK_LAUNCH("gather_quantized_matvec");
    return TL_OK;
}

extern "C" int tl_quantized_matmul_split_k(int M, int N, int K, int use_simdgroup, int use_split_k) {
    if (!use_simdgroup || !use_split_k || M <= 8 || N < 128) return 1;
    return split_k_policy(M, N, K);
}

extern "C" size_t tl_quantized_matmul_workspace_bytes(int M, int N, int K, tl_dtype dtype, int use_simdgroup,
                                                      int use_split_k) {
    (void)dtype;
    const int s = tl_quantized_matmul_split_k(M, N, K, use_simdgroup, use_split_k);
    return s > 1 ? (size_t)s * M * K * 2 : 0;
}

extern "C" int tl_quantized_matmul(const void *scales, const void *biases, const void *a, const uint32_t *b,
                                   void *out, int M, int N, int K, int group_size, int bits, tl_dtype dtype,
                                   int use_simdgroup, int use_split_k, void *workspace, size_t workspace_bytes,
                                   void *stream) {
    TL_REQUIRE(dtype == TL_F16 || dtype == TL_BF16, "quantized_matmul: scales must be float16 or bfloat16");
    TL_REQUIRE(bits == 4, "quantized_matmul: bits must be 4");
    TL_REQUIRE(group_size == 128, "quantized_matmul: group_size must be 128");
    TL_REQUIRE(scales && biases && a && b && out, "quantized_matmul: null pointer");
    TL_REQUIRE(M >= 0 && K >= 0 && N > 0, "quantized_matmul: a must be a 2D array");
    TL_REQUIRE(N % 128 == 0, "quantized_matmul: N must be divisible by group_size");
    TL_REQUIRE(((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0), "quantized_matmul: a must be contiguous");
    if (M == 0 || K == 0) return TL_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (dtype == TL_F16) {
        rc = run_qmm<F16>(scales, biases, a, b, out, M, N, K, use_simdgroup, use_split_k, workspace, workspace_bytes, st);
    } else {
        rc = run_qmm<BF16>(scales, biases, a, b, out, M, N, K, use_simdgroup, use_split_k, workspace, workspace_bytes, st);
    }
    if (rc != TL_OK) return rc;
    TL_CHECK_LAUNCH("quantized_matmul");
    return TL_OK;
}
