// W4A16 (group 128) decode GEMV for gfx950:  out[m,k] = sum_n a[m,n] * (q[k,n]*s[k,g] + beta[k,g])
//
// This is the north-star kernel (reference: quantized_matvec_x4_fast,
// quantized_matmul.metal:441-538; dispatch quantized_matmul.cpp:137,214-222).
// It is NOT a transliteration of the 32-lane Metal schedule:
//
//  * HBM stream.  A wave64 owns a 4-row x 512-column tile per load instruction:
//    lane = (row-in-tile rg = lane/16, column slot cl = lane%16), each lane
//    pulls 16 B (32 nibbles) with one global_load_dwordx4, so every row
//    contributes a 256 B contiguous segment per instruction.  All loads of a
//    batch (U chunks x RPL rows) are issued before anything waits on them, and
//    the next batch is issued before the current one is consumed.
//  * Activations live in LDS once per workgroup (bf16, <= 19 KB for N = 9728),
//    pre-permuted so that the packed pairs (a0,a4)(a1,a5)(a2,a6)(a3,a7) match
//    the nibble pairs that `(w >> 4i) & 0x000f000f` exposes.  Per-32-column
//    activation sums are kept next to them.
//  * Dequant without shifts-per-weight or int->float converts: OR-ing the
//    nibble pair with 0x4300_4300 makes two bf16 values (128+q), which feed
//    v_dot2c_f32_bf16 directly (f16: 0x6400_6400 = 1024+q, v_dot2_f32_f16).
//    acc += s * dot + (beta - OFF*s) * asum   (the algebraic form the reference
//    fast kernel uses, quantized_matmul.metal:510-521, with the offset folded
//    into the bias term).
//  * Reduction over the 16 column lanes is 4 xor-shuffles; when the reduction
//    dimension is split over the waves of a workgroup (WN > 1, used when K is
//    too small to fill 256 CUs) partials meet in LDS.
//  * Optional fused prologue (RMSNorm of the activation row, rounded to bf16 at
//    the reference op boundary) and epilogues (residual add, SwiGLU on
//    interleaved gate/up rows) serve the decode fast path; the plain variant
//    backs the public operator.
#pragma once
#include "common.h"

namespace tl {

enum { PRO_NONE = 0, PRO_RMSNORM = 1, PRO_ATTN_MERGE = 2, PRO_RMS_WEIGHTED = 3 };  // the last two: qmv3.h only (Qmv3Args::merge_ws / ::ss_in)
enum { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_SWIGLU = 2 };

struct QmvArgs {
    const uint16_t *scales;    // [K, N/128]
    const uint16_t *biases;    // [K, N/128]
    const uint16_t *a;         // [M, N]
    const uint32_t *b;         // [K, N/8]
    uint16_t *out;             // [M, K]  (EPI_SWIGLU: [M, K/2])
    const uint16_t *norm_w;    // [N]      PRO_RMSNORM
    const uint16_t *residual;  // [M, K]   EPI_RESIDUAL
    float eps;
    int M, N, K;
    prof_t *prof;   // nullptr except during an engine profile step
    prof_t *trace;  // lab-only phase stamps (QMV2_TRACE builds)
    // grouped-expert mode (tl_gather_quantized_matvec; reference mx.gather_qmm in moe.py:7-36): when expert_ids is set the
    // launch has gridDim.y = rows, and row m multiplies with the weights of expert expert_ids[m]: b [E, K, N/8],
    // scales / biases [E, K, N/128].  The kernel then runs as M = 1 on that row.
    const int32_t *expert_ids;
    int num_experts;
    int a_rows_div;  // grouped-expert mode: output row m reads activation row m / a_rows_div (0 or 1: its own row) -- the top_k
                     // expert rows of one token share the token's activation row, which is then not replicated (engine MoE layers)
};

template <typename TT>
struct Dot2;
template <>
struct Dot2<BF16> {
    static constexpr uint32_t MAGIC = 0x43004300u;
    static constexpr float OFFSET = 128.0f;
    __device__ __forceinline__ static uint32_t unbias(uint32_t w) { return w; }
    __device__ __forceinline__ static float dot(uint32_t w, uint32_t x, float acc) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, x), acc,
                                               false);
    }
};
template <>
struct Dot2<F16> {
    // 0x6400 | q = 1024 + q.  Left in place, the offset costs ten of the fp32 accumulator's bits (sum a (1024 + q) - 1024 sum a
    // cancels to a value 2^-10 of its operands): a few f16 steps of error in the result, found by the per-element tolerances of
    // round 3.  One packed f16 add takes the offset out exactly (1024 + q and q are both f16 values), so the dot product runs on q.
    static constexpr uint32_t MAGIC = 0x64006400u;
    static constexpr float OFFSET = 0.0f;
    __device__ __forceinline__ static uint32_t unbias(uint32_t w) {
        const f16x2_t off = {(_Float16)-1024.0f, (_Float16)-1024.0f};
        return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2_t, w) + off);
    }
    __device__ __forceinline__ static float dot(uint32_t w, uint32_t x, float acc) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, w), __builtin_bit_cast(f16x2_t, x), acc, false);
    }
};

constexpr int QMV_U = 4;  // chunks per lane per batch

template <int RPL>
struct QmvBatch {
    u32x4 w[QMV_U][RPL];
    uint16_t s[QMV_U][RPL];
    uint16_t b[QMV_U][RPL];
};

__host__ __device__ inline size_t qmv_lds_bytes(int MR, int N, int WN, int RPL) {
    size_t x = (size_t)MR * N * 2;
    size_t as = (size_t)MR * (N / 32) * 4;
    size_t red = WN > 1 ? (size_t)4 * RPL * 4 * MR * 4 : 0;  // [waves][RPL*4 rows][MR]
    return x + as + red + 64;
}

template <typename TT, int MR, int WN, int RPL, int PRO, int EPI>
__global__ __launch_bounds__(256) void qmv_kernel(const QmvArgs p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    QmvArgs p = p_in;
    if (p.expert_ids) {  // uniform: one activation row and one expert per blockIdx.y
        const int m = blockIdx.y;
        const long e = min(max(p.expert_ids[m], 0), p.num_experts - 1);
        p.b += e * (long)p.K * (p.N >> 3);
        p.scales += e * (long)p.K * (p.N >> 7);
        p.biases += e * (long)p.K * (p.N >> 7);
        p.a += (long)(p.a_rows_div > 1 ? m / p.a_rows_div : m) * p.N;
        p.out += (long)m * p.K;
        p.M = 1;
    }
    constexpr int WR = 4 / WN;
    constexpr int RB = 4 * WR * RPL;
    using D2 = Dot2<TT>;

    const prof_t prof_t0 = prof_begin(p.prof);
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int wr = wave / WN;
    const int wn = wave % WN;
    const int rg = lane >> 4;
    const int cl = lane & 15;
    const int N = p.N, K = p.K;
    const int G = N >> 7;
    const int words = N >> 3;
    const int c32 = N >> 5;  // 32-column chunks per row

    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);                        // [MR][N] permuted
    float *asum = reinterpret_cast<float *>(smem + (size_t)MR * N * 2);       // [MR][N/32]
    float *red = asum + (size_t)MR * c32;                                     // WN>1 partials

    int row[RPL];
    bool row_ok[RPL];
#pragma unroll
    for (int rp = 0; rp < RPL; ++rp) {
        row[rp] = blockIdx.x * RB + (wr * RPL + rp) * 4 + rg;
        row_ok[rp] = row[rp] < K;
    }
    const int J = (N + 511) >> 9;                 // 512-column chunks per row
    const int nch = J > wn ? (J - wn + WN - 1) / WN : 0;  // chunks owned by this wave

    auto load_batch = [&](QmvBatch<RPL> &B, int b0) {
#pragma unroll
        for (int u = 0; u < QMV_U; ++u) {
            const int j = wn + (b0 + u) * WN;
            const int col0 = (j * 16 + cl) * 32;
            const bool cok = (b0 + u) < nch && col0 < N;
#pragma unroll
            for (int rp = 0; rp < RPL; ++rp) {
                if (cok && row_ok[rp]) {
                    const size_t r = (size_t)row[rp];
                    B.w[u][rp] = *reinterpret_cast<const u32x4 *>(p.b + r * words + (col0 >> 3));
                    B.s[u][rp] = p.scales[r * G + (col0 >> 7)];
                    B.b[u][rp] = p.biases[r * G + (col0 >> 7)];
                } else {
                    B.w[u][rp] = u32x4{0u, 0u, 0u, 0u};
                    B.s[u][rp] = 0;
                    B.b[u][rp] = 0;
                }
            }
        }
    };

    // 1. Put the first weight batch in flight before touching the activations.
    QmvBatch<RPL> cur, nxt;
    load_batch(cur, 0);

    // 2. Stage activations: [optional RMSNorm] -> permuted bf16 in LDS + per-32 sums.
    float inv[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) inv[m] = 1.0f;
    if constexpr (PRO == PRO_RMSNORM) {
        float ss[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m) ss[m] = 0.f;
        for (int c = tid; c < MR * c32; c += 256) {
            const int m = c / c32;
            const int cc = c - m * c32;
            if (m < p.M) {
                const u32x4 *src = reinterpret_cast<const u32x4 *>(p.a + (size_t)m * N + cc * 32);
                float part = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u32x4 v = src[q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = TT::to_float((uint16_t)(v[e] & 0xffffu));
                        const float hi = TT::to_float((uint16_t)(v[e] >> 16));
                        part += lo * lo + hi * hi;
                    }
                }
#pragma unroll
                for (int mm = 0; mm < MR; ++mm) ss[mm] += (mm == m) ? part : 0.f;
            }
        }
        // block reduce through the asum area (not yet written; N >= 128 gives >= 4 slots per row)
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            float v = wave_sum(ss[m]);
            if (lane == 0) asum[m * 4 + wave] = v;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const float tot = asum[m * 4 + 0] + asum[m * 4 + 1] + asum[m * 4 + 2] + asum[m * 4 + 3];
            inv[m] = rsqrtf(tot / (float)N + p.eps);
        }
        __syncthreads();
    }
    for (int c = tid; c < MR * c32; c += 256) {
        const int m = c / c32;
        const int cc = c - m * c32;
        u32x4 o[4];
        float sum = 0.f;
        if (m < p.M) {
            const u32x4 *src = reinterpret_cast<const u32x4 *>(p.a + (size_t)m * N + cc * 32);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x4 v = src[q];
                if constexpr (PRO == PRO_RMSNORM) {
                    const u32x4 g = reinterpret_cast<const u32x4 *>(p.norm_w + cc * 32)[q];
                    float iv = 1.0f;
#pragma unroll
                    for (int mm = 0; mm < MR; ++mm) iv = (mm == m) ? inv[mm] : iv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = TT::to_float((uint16_t)(v[e] & 0xffffu)) * iv *
                                         TT::to_float((uint16_t)(g[e] & 0xffffu));
                        const float hi = TT::to_float((uint16_t)(v[e] >> 16)) * iv * TT::to_float((uint16_t)(g[e] >> 16));
                        v[e] = (uint32_t)TT::from_float(lo) | ((uint32_t)TT::from_float(hi) << 16);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sum += TT::to_float((uint16_t)(v[e] & 0xffffu)) + TT::to_float((uint16_t)(v[e] >> 16));
                }
                // v = {(a1,a0),(a3,a2),(a5,a4),(a7,a6)} -> {(a4,a0),(a5,a1),(a6,a2),(a7,a3)}
                o[q][0] = (v[0] & 0xffffu) | (v[2] << 16);
                o[q][1] = (v[0] >> 16) | (v[2] & 0xffff0000u);
                o[q][2] = (v[1] & 0xffffu) | (v[3] << 16);
                o[q][3] = (v[1] >> 16) | (v[3] & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = u32x4{0u, 0u, 0u, 0u};
        }
        u32x4 *dst = reinterpret_cast<u32x4 *>(xs + (size_t)m * N + cc * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = o[q];
        asum[m * c32 + cc] = sum;
    }
    __syncthreads();

    // 3. Stream the weights.
    float acc[RPL][MR];
#pragma unroll
    for (int rp = 0; rp < RPL; ++rp)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[rp][m] = 0.f;

    for (int b0 = 0; b0 < nch; b0 += QMV_U) {
        if (b0 + QMV_U < nch) load_batch(nxt, b0 + QMV_U);
#pragma unroll
        for (int u = 0; u < QMV_U; ++u) {
            const int j = wn + (b0 + u) * WN;
            int col0 = (j * 16 + cl) * 32;
            const bool cok = (b0 + u) < nch && col0 < N;
            col0 = cok ? col0 : 0;
            uint32_t pw[RPL][16];
            float sc[RPL], bo[RPL];
#pragma unroll
            for (int rp = 0; rp < RPL; ++rp) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t w = cur.w[u][rp][q];
                    pw[rp][q * 4 + 0] = D2::unbias((w & 0x000f000fu) | D2::MAGIC);
                    pw[rp][q * 4 + 1] = D2::unbias(((w >> 4) & 0x000f000fu) | D2::MAGIC);
                    pw[rp][q * 4 + 2] = D2::unbias(((w >> 8) & 0x000f000fu) | D2::MAGIC);
                    pw[rp][q * 4 + 3] = D2::unbias(((w >> 12) & 0x000f000fu) | D2::MAGIC);
                }
                sc[rp] = TT::to_float(cur.s[u][rp]);
                bo[rp] = TT::to_float(cur.b[u][rp]) - D2::OFFSET * sc[rp];
            }
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const u32x4 *xp = reinterpret_cast<const u32x4 *>(xs + (size_t)m * N + col0);
                const float as = asum[m * c32 + (col0 >> 5)];
                u32x4 xa[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) xa[q] = xp[q];
#pragma unroll
                for (int rp = 0; rp < RPL; ++rp) {
                    float d = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        d = D2::dot(pw[rp][q * 4 + 0], xa[q][0], d);
                        d = D2::dot(pw[rp][q * 4 + 1], xa[q][1], d);
                        d = D2::dot(pw[rp][q * 4 + 2], xa[q][2], d);
                        d = D2::dot(pw[rp][q * 4 + 3], xa[q][3], d);
                    }
                    acc[rp][m] += sc[rp] * d + bo[rp] * as;
                }
            }
        }
        cur = nxt;
    }

    // 4. Reduce over the 16 column lanes, then over the N-split waves.
#pragma unroll
    for (int rp = 0; rp < RPL; ++rp)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[rp][m] = group16_sum(acc[rp][m]);

    if constexpr (WN > 1) {
        if (cl == 0) {
#pragma unroll
            for (int rp = 0; rp < RPL; ++rp)
#pragma unroll
                for (int m = 0; m < MR; ++m) red[((wave * RPL + rp) * 4 + rg) * MR + m] = acc[rp][m];
        }
        __syncthreads();
        if (wn != 0) {
            prof_end(p.prof, prof_t0);
            return;
        }
#pragma unroll
        for (int rp = 0; rp < RPL; ++rp)
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < WN; ++k) v += red[(((wr * WN + k) * RPL + rp) * 4 + rg) * MR + m];
                acc[rp][m] = v;
            }
    }

    // 5. Epilogue.
#pragma unroll
    for (int rp = 0; rp < RPL; ++rp) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const float v = acc[rp][m];
            if constexpr (EPI == EPI_SWIGLU) {
                // rows are interleaved: even = gate_i, odd = up_i ; rg pairs (0,1) and (2,3)
                const float g = TT::to_float(TT::from_float(v));
                const float upv = __shfl(g, (lane + 16) & 63, 64);
                if (cl == 0 && (rg & 1) == 0 && row_ok[rp] && m < p.M) {
                    const float r = (g / (1.0f + expf(-g))) * upv;
                    p.out[(size_t)m * (K >> 1) + (row[rp] >> 1)] = TT::from_float(r);
                }
            } else if constexpr (EPI == EPI_RESIDUAL) {
                if (cl == 0 && row_ok[rp] && m < p.M) {
                    const size_t o = (size_t)m * K + row[rp];
                    const float r = TT::to_float(TT::from_float(v));
                    p.out[o] = TT::from_float(TT::to_float(p.residual[o]) + r);
                }
            } else {
                if (cl == 0 && row_ok[rp] && m < p.M) p.out[(size_t)m * K + row[rp]] = TT::from_float(v);
            }
        }
    }
    prof_end(p.prof, prof_t0);
}

// Host-side launch heuristic shared by the operator and the decode fast path.
struct QmvPlan {
    int MR, WN, RPL, RB, blocks;
    size_t lds;
};
inline QmvPlan qmv_plan(int M, int N, int K) {
    QmvPlan pl;
    pl.MR = M <= 1 ? 1 : (M <= 2 ? 2 : (M <= 4 ? 4 : 8));
    const int J = (N + 511) / 512;
    // rows per workgroup: largest tile that still yields >= 3 workgroups per CU
    // (256 CUs); small K falls back to splitting the reduction over the waves.
    int rb = 4;
    const int cands[3] = {32, 16, 8};
    for (int c : cands) {
        if ((K + c - 1) / c >= 768) {
            rb = c;
            break;
        }
    }
    if (rb == 32) {
        pl.WN = 1; pl.RPL = 2;
    } else if (rb == 16) {
        pl.WN = 1; pl.RPL = 1;
    } else if (rb == 8) {
        pl.WN = 2; pl.RPL = 1;
    } else {
        pl.WN = 4; pl.RPL = 1;
    }
    if (pl.WN > J) {  // not enough 512-column chunks to split: keep waves on rows
        pl.WN = J >= 2 ? 2 : 1;
        pl.RPL = 1;
    }
    pl.RB = 4 * (4 / pl.WN) * pl.RPL;
    pl.blocks = (K + pl.RB - 1) / pl.RB;
    pl.lds = qmv_lds_bytes(pl.MR, N, pl.WN, pl.RPL);
    return pl;
}

// Fused-variant launcher (bf16 only), defined in qmv_fused.hip.  pro/epi are PRO_* / EPI_*.
// Returns 0, or -1 when the activation tile does not fit in LDS, -2 for an unknown plan.
int launch_qmv_fused_bf16(const QmvArgs &args, int pro, int epi, hipStream_t st);

// qmm.hip: grouped-expert GEMV (bf16): out[m] = a[m / a_rows_div] @ dequant(b[expert_ids[m]])^T, one launch, grid.y = M rows
int gather_qmv_bf16(const void *scales, const void *biases, const uint16_t *a, const uint32_t *b, const int32_t *expert_ids,
                    uint16_t *out, int M, int N, int K, int num_experts, int a_rows_div, hipStream_t st);
// qmm.hip: the prefill W4 GEMM with the engine's epilogue folded in (rows > 8; epi = EPI_RESIDUAL / EPI_SWIGLU)
// norm_w / norm_out / norm_done (optional, EPI_RESIDUAL): when the projection is split and its reduction pass runs, that pass also writes the RMSNorm of the
// finished rows (weights norm_w) to norm_out and sets *norm_done -- bit-identical to tl_rms_norm on the output; otherwise *norm_done = false and nothing is written
int qmm_bf16_epilogue(const void *scales, const void *biases, const uint16_t *a, const uint32_t *b, uint16_t *out, int M, int N, int K,
                      int epi, const uint16_t *residual, void *workspace, size_t workspace_bytes, hipStream_t st, const uint16_t *norm_w = nullptr,
                      uint16_t *norm_out = nullptr, float norm_eps = 0.f, bool *norm_done = nullptr);
}  // namespace tl
