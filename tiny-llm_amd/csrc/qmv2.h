// W4A16 (group 128) decode GEMV on the matrix cores (gfx950):  out[m,k] = sum_n a[m,n] * (q[k,n]*s[k,g] + beta[k,g])
//
// Why MFMA for a matrix-VECTOR product: int4 weights carry 4x more multiply-adds per HBM byte than bf16 ones, and
// the packed-dot path (v_dot2c_f32_bf16, qmv.h) tops out at ~3.3 TB/s of weight stream on this chip — it is
// VALU-bound, not HBM-bound (tools/lab/gemv_lab.hip, profiles/).  One v_mfma_f32_16x16x32_bf16 retires
// 16 rows x 32 reduction elements per lane-word in ~16 cycles, so the dot products become almost free and the only
// per-weight VALU work left is the nibble unpack (7 ops per 8 weights).  The 16 columns of the B operand are the
// activation rows (decode batch M <= 8 here; columns >= M hold don't-care data), so batched decode costs no extra
// weight traffic.
//
// Data flow per wave (one 16-row tile x a slice of the reduction dimension):
//   * all of the wave's weight loads are issued up front: lane (r = lane&15, c = lane>>4) pulls 16 B = 32 nibbles
//     of row r for every quantisation group g of its slice (nontemporal: each byte is used once).  One load
//     instruction = one full group of 128 columns for 16 rows, i.e. exactly the A operands of 4 MFMAs.
//   * word t of that load is expanded with (w >> 4i) & 0x000f000f | 0x43004300 into four bf16 pairs (128+q); as an
//     MFMA A operand the element order is (q0,q4,q1,q5,q2,q6,q3,q7), so the activations are stored in LDS with the
//     same permutation inside every 8-element block.
//   * per group: D = sum_k (128+q) x  (4 chained MFMAs), then acc += s*D + (beta - 128 s) * sum_k x
//     (the algebraic form of the reference fast kernel, quantized_matmul.metal:510-521).  s and beta-128s of the
//     workgroup's rows are staged once in LDS as fp32 pairs, the per-group activation sums next to the activations.
//   * the reduction dimension is split over KS waves of the workgroup when a row tile alone would not keep the
//     memory system busy (K = 2560-row projections); partial sums meet in LDS.
//   * fused prologue / epilogues as in qmv.h (RMSNorm of the activation rows; residual add; SwiGLU on interleaved
//     gate/up rows — a lane owns 4 consecutive output rows, so a gate/up pair never leaves the lane).
// reference: quantized_matvec_x4_fast (quantized_matmul.metal:441-538), dispatch quantized_matmul.cpp:137,214-222.
#pragma once
#include "common.h"
#include "qmv.h"

namespace tl {

#ifdef QMV2_TRACE
// lab-only phase stamps: workgroup 0, wave 0, lane 0 writes wall-clock ticks to p.trace[slot]
#define Q2_STAMP(slot) do { __builtin_amdgcn_sched_barrier(0); if (p.trace && blockIdx.x == Q2_TRACE_WG && threadIdx.x == 0) p.trace[slot] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#ifndef Q2_TRACE_WG
#define Q2_TRACE_WG 0
#endif
#else
#define Q2_STAMP(slot) do {} while (0)
#endif

constexpr int Q2_LMAX = 10;   // weight loads (groups) a wave keeps in flight
constexpr int Q2_XU = 4;      // activation chunks a thread loads per staging round
constexpr int Q2_PAD = 8;     // bf16 elements of padding per activation row in LDS

struct Qmv2Lds {
    size_t xs, xsum, sb, nw, red, total;
};
__host__ __device__ inline Qmv2Lds qmv2_lds(int MR, int N, int KS, int WAVES, bool rms) {
    const int G = N / 128, WR = WAVES / KS;
    Qmv2Lds l;
    size_t off = 0;
    l.xs = off;   off += (size_t)MR * (N + Q2_PAD) * 2;
    off = (off + 15) & ~(size_t)15;
    l.xsum = off; off += (size_t)16 * G * 4;  // [g][16 activation rows]; rows >= MR are never read into a result
    off = (off + 15) & ~(size_t)15;
    l.sb = off;   off += (size_t)WR * G * 16 * 8;
    l.nw = off;   off += rms ? (size_t)N * 2 : 0;
    off = (off + 15) & ~(size_t)15;
    l.red = off;  off += KS > 1 ? (size_t)WAVES * MR * 16 * 4 : 0;
    l.total = off + 64;
    return l;
}

// All global loads below are issued UNCONDITIONALLY from clamped addresses and masked afterwards: hipcc puts a
// `s_waitcnt vmcnt(0)` at the join of every divergent branch that contains a load, which serialises the loads into
// one HBM round trip each (seen in the ISA of the first version of this kernel: 4 + 10 dependent round trips).
template <int MR, int KS, int WAVES, int PRO, int EPI>
__global__ __launch_bounds__(WAVES * 64) void qmv2_kernel(const QmvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = WAVES * 64;
    constexpr int WR = WAVES / KS;
    prof_begin(p.prof);
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int r = lane & 15;  // B (weights): row in tile | A (activations): activation row | D: weight row (column)
    const int c = lane >> 4;  // A, B: k-block                                              | D: activation rows 4c..4c+3
    const int wt = wave / KS;
    const int ks = wave - wt * KS;
    const int N = p.N, K = p.K, G = N >> 7, words = N >> 3;
    const int tiles = K >> 4;
    const int tile = blockIdx.x * WR + wt;
    const bool tile_ok = tile < tiles;
    const int row0 = tile_ok ? (tile << 4) : 0;
    const int Lper = (G + KS - 1) / KS;
    const int g0 = ks * Lper;
    const int g1 = tile_ok ? min(g0 + Lper, G) : g0;

    const Qmv2Lds L = qmv2_lds(MR, N, KS, WAVES, PRO == PRO_RMSNORM);
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem + L.xs);
    float *xsum = reinterpret_cast<float *>(smem + L.xsum);
    float2 *sb = reinterpret_cast<float2 *>(smem + L.sb);
    uint16_t *nws = reinterpret_cast<uint16_t *>(smem + L.nw);
    float *red = reinterpret_cast<float *>(smem + L.red);
    const int xstride = N + Q2_PAD;

    Q2_STAMP(0);
    // ---- 1. small loads first (vmcnt retires in order, so they come back first) -------------------------------
    const int cpr = N >> 3;  // 16-byte chunks per activation row
    const int xchunks = MR * cpr;
    u32x4 xv[Q2_XU];
#pragma unroll
    for (int u = 0; u < Q2_XU; ++u) {
        const int i = tid + u * T;
        const int m = i / cpr;
        const bool ok = i < xchunks && m < p.M;
        const size_t off = ok ? ((size_t)m * N + (size_t)(i - m * cpr) * 8) : 0;
        xv[u] = *reinterpret_cast<const u32x4 *>(p.a + off);
        if (!ok) xv[u] = u32x4{0u, 0u, 0u, 0u};
    }
    u32x4 nwv[Q2_XU];
    if constexpr (PRO == PRO_RMSNORM) {
#pragma unroll
        for (int u = 0; u < Q2_XU; ++u) {
            const int i = tid + u * T;
            nwv[u] = *reinterpret_cast<const u32x4 *>(p.norm_w + (size_t)(i < cpr ? i : 0) * 8);
        }
    }
    // scales / biases of the workgroup's rows: one contiguous [rows][G] region each
    const int sb_rows = min(WR * 16, K - blockIdx.x * WR * 16);
    const int sb_chunks = (sb_rows * G) >> 3;  // K % 16 == 0 -> a multiple of 8 elements
    const size_t sb_base = (size_t)blockIdx.x * WR * 16 * G;
    u32x4 sv, bv;
    {
        const size_t off = sb_base + (size_t)(tid < sb_chunks ? tid : 0) * 8;
        sv = *reinterpret_cast<const u32x4 *>(p.scales + off);
        bv = *reinterpret_cast<const u32x4 *>(p.biases + off);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the small loads ahead of the weight stream

    Q2_STAMP(1);
    // ---- 2. the wave's whole weight slice goes in flight ---------------------------------------------------------
    u32x4 wq[Q2_LMAX];
    const uint32_t *wrow = p.b + (size_t)(row0 + r) * words + 4 * c;
#pragma unroll
    for (int i = 0; i < Q2_LMAX; ++i) {
        const int g = min(g0 + i, G - 1);
        wq[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wrow + (size_t)g * 16));
    }

    __builtin_amdgcn_sched_barrier(0);

    Q2_STAMP(2);
    // ---- 3. scales / biases -> fp32 (s, beta - 128 s) in LDS, [tile][g][row] ---------------------------------------
    auto stage_sb = [&](int i, const u32x4 &s4, const u32x4 &b4) {
        int rl = (i * 8) / G;
        int g = i * 8 - rl * G;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t sw = s4[e >> 1], bw = b4[e >> 1];
            const float sc = BF16::to_float((uint16_t)((e & 1) ? (sw >> 16) : (sw & 0xffffu)));
            const float be = BF16::to_float((uint16_t)((e & 1) ? (bw >> 16) : (bw & 0xffffu)));
            sb[((size_t)(rl >> 4) * G + g) * 16 + (rl & 15)] = make_float2(sc, be - 128.0f * sc);
            if (++g == G) {
                g = 0;
                ++rl;
            }
        }
    };
    if (tid < sb_chunks) stage_sb(tid, sv, bv);
    for (int i = tid + T; i < sb_chunks; i += T)  // only for very long rows (16 * G > 8 * T)
        stage_sb(i, *reinterpret_cast<const u32x4 *>(p.scales + sb_base + (size_t)i * 8),
                 *reinterpret_cast<const u32x4 *>(p.biases + sb_base + (size_t)i * 8));

    Q2_STAMP(3);
    // ---- 4. activations: [RMSNorm ->] bf16, permuted inside 8-blocks, + per-group sums ---------------------------
    float inv[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) inv[m] = 1.0f;
    if constexpr (PRO == PRO_RMSNORM) {
        float ss[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m) ss[m] = 0.f;
#pragma unroll
        for (int u = 0; u < Q2_XU; ++u) {
            const int i = tid + u * T;
            const int m = i / cpr;
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = BF16::to_float((uint16_t)(xv[u][e] & 0xffffu));
                const float hi = BF16::to_float((uint16_t)(xv[u][e] >> 16));
                part += lo * lo + hi * hi;
            }
#pragma unroll
            for (int mm = 0; mm < MR; ++mm) ss[mm] += (mm == m) ? part : 0.f;
            if (i < cpr) *reinterpret_cast<u32x4 *>(nws + (size_t)i * 8) = nwv[u];
        }
        // more than Q2_XU * T chunks (large M * N): extra rounds straight from global memory
        for (int i = tid + Q2_XU * T; i < xchunks; i += T) {
            const int m = i / cpr;
            const bool ok = m < p.M;
            u32x4 v = *reinterpret_cast<const u32x4 *>(p.a + (ok ? ((size_t)m * N + (size_t)(i - m * cpr) * 8) : 0));
            if (!ok) v = u32x4{0u, 0u, 0u, 0u};
            float part = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = BF16::to_float((uint16_t)(v[e] & 0xffffu));
                const float hi = BF16::to_float((uint16_t)(v[e] >> 16));
                part += lo * lo + hi * hi;
            }
#pragma unroll
            for (int mm = 0; mm < MR; ++mm) ss[mm] += (mm == m) ? part : 0.f;
        }
        for (int i = tid + Q2_XU * T; i < cpr; i += T)
            *reinterpret_cast<u32x4 *>(nws + (size_t)i * 8) = *reinterpret_cast<const u32x4 *>(p.norm_w + (size_t)i * 8);
        float *scratch = xsum;  // not in use yet; MR * WAVES <= 16 * G floats
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const float v = wave_sum(ss[m]);
            if (lane == 0) scratch[m * WAVES + wave] = v;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) tot += scratch[m * WAVES + w];
            inv[m] = rsqrtf(tot / (float)N + p.eps);
        }
        __syncthreads();
    }
    auto stage_chunk = [&](int i, u32x4 v) {
        const int m = i / cpr;
        const int cc = i - m * cpr;
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[2 * e] = BF16::to_float((uint16_t)(v[e] & 0xffffu));
            f[2 * e + 1] = BF16::to_float((uint16_t)(v[e] >> 16));
        }
        if constexpr (PRO == PRO_RMSNORM) {
            const u32x4 g = *reinterpret_cast<const u32x4 *>(nws + (size_t)cc * 8);
            float iv = 1.0f;
#pragma unroll
            for (int mm = 0; mm < MR; ++mm) iv = (mm == m) ? inv[mm] : iv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[2 * e] = bf16_round(f[2 * e] * iv * BF16::to_float((uint16_t)(g[e] & 0xffffu)));
                f[2 * e + 1] = bf16_round(f[2 * e + 1] * iv * BF16::to_float((uint16_t)(g[e] >> 16)));
            }
        }
        float sum = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
        // element order (a0,a4,a1,a5,a2,a6,a3,a7): matches the nibble pairs of the unpacked weights
        u32x4 o;
        o[0] = BF16::pack2(f[0], f[4]);
        o[1] = BF16::pack2(f[1], f[5]);
        o[2] = BF16::pack2(f[2], f[6]);
        o[3] = BF16::pack2(f[3], f[7]);
        *reinterpret_cast<u32x4 *>(xs + (size_t)m * xstride + (size_t)cc * 8) = o;
        // 16 consecutive chunks = one group of 128 columns (T and cpr are multiples of 16)
        sum = group16_sum(sum);
        if ((cc & 15) == 0) xsum[(cc >> 4) * 16 + m] = sum;
    };
#pragma unroll
    for (int u = 0; u < Q2_XU; ++u) {
        const int i = tid + u * T;
        if (i < xchunks) stage_chunk(i, xv[u]);  // uniform per 16-lane group: xchunks % 16 == 0
    }
    for (int i = tid + Q2_XU * T; i < xchunks; i += T) {
        const int m = i / cpr;
        const bool ok = m < p.M;
        u32x4 v = *reinterpret_cast<const u32x4 *>(p.a + (ok ? ((size_t)m * N + (size_t)(i - m * cpr) * 8) : 0));
        if (!ok) v = u32x4{0u, 0u, 0u, 0u};
        stage_chunk(i, v);
    }
    __syncthreads();

    Q2_STAMP(4);
    // ---- 5. MFMA over the wave's groups: D[act row][weight row] += x[act row][k] * (128 + q[weight row][k]) ------------
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const uint16_t *xrow = xs + (size_t)(r % MR) * xstride + 32 * c;  // A rows >= MR repeat row r % MR (never stored)
    auto do_group = [&](const u32x4 &w4, int g) {
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t w = w4[t];
            u32x4 bq;
            bq[0] = (w & 0x000f000fu) | 0x43004300u;
            bq[1] = ((w >> 4) & 0x000f000fu) | 0x43004300u;
            bq[2] = ((w >> 8) & 0x000f000fu) | 0x43004300u;
            bq[3] = ((w >> 12) & 0x000f000fu) | 0x43004300u;
            const u32x4 ax = *reinterpret_cast<const u32x4 *>(xrow + (size_t)g * 128 + 8 * t);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ax), __builtin_bit_cast(bf16x8_t, bq), d,
                                                        0, 0, 0);
        }
        const float2 sc = sb[((size_t)wt * G + g) * 16 + r];
        const f32x4 xg = *reinterpret_cast<const f32x4 *>(&xsum[g * 16 + 4 * c]);
        acc[0] += sc.x * d[0] + sc.y * xg[0];
        acc[1] += sc.x * d[1] + sc.y * xg[1];
        acc[2] += sc.x * d[2] + sc.y * xg[2];
        acc[3] += sc.x * d[3] + sc.y * xg[3];
    };
#pragma unroll
    for (int i = 0; i < Q2_LMAX; ++i) {
        if (g0 + i < g1) do_group(wq[i], g0 + i);  // wave-uniform
    }
    // slices longer than Q2_LMAX groups (N > 1280 * KS): plain loop
    for (int g = g0 + Q2_LMAX; g < g1; ++g) {
        const u32x4 w4 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wrow + (size_t)g * 16));
        do_group(w4, g);
    }

    Q2_STAMP(5);
    // ---- 6. reduce over the KS waves of a tile ----------------------------------------------------------------------
    if constexpr (KS > 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int arow = 4 * c + i;
            if (arow < MR) red[((size_t)wave * MR + arow) * 16 + r] = acc[i];
        }
        __syncthreads();
        if (ks != 0) {
            prof_end(p.prof);
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int arow = 4 * c + i;
            if (arow < MR) {
#pragma unroll
                for (int k2 = 1; k2 < KS; ++k2) acc[i] += red[((size_t)(wave + k2) * MR + arow) * 16 + r];
            }
        }
    }

    Q2_STAMP(6);
    // ---- 7. epilogue: lane (weight row r, c) holds activation rows 4c .. 4c+3 -----------------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int arow = 4 * c + i;
        const bool live = tile_ok && arow < MR && arow < p.M;
        const int orow = row0 + r;
        if constexpr (EPI == EPI_SWIGLU) {
            // rows are interleaved: even = gate_i, odd = up_i
            const float gv = bf16_round(acc[i]);
            const float uv = __shfl_down(gv, 1, 64);
            if (live && (r & 1) == 0)
                p.out[(size_t)arow * (K >> 1) + (orow >> 1)] = BF16::from_float((gv / (1.0f + expf(-gv))) * uv);
        } else if constexpr (EPI == EPI_RESIDUAL) {
            if (live) {
                const size_t o = (size_t)arow * K + orow;
                p.out[o] = BF16::from_float(BF16::to_float(p.residual[o]) + bf16_round(acc[i]));
            }
        } else {
            if (live) p.out[(size_t)arow * K + orow] = BF16::from_float(acc[i]);
        }
    }
    Q2_STAMP(7);
    prof_end(p.prof);
}

// Host-side plan.  Applicable when K % 16 == 0 (whole row tiles) and the activation rows fit in LDS.
struct Qmv2Plan {
    int MR, KS, WAVES, blocks;
    size_t lds;
    bool ok;
};
inline Qmv2Plan qmv2_plan(int M, int N, int K, bool rms, int force_ks = 0) {
    Qmv2Plan pl{};
    pl.MR = M <= 1 ? 1 : (M <= 2 ? 2 : (M <= 4 ? 4 : 8));
    const int G = N / 128;
    const int tiles = K / 16;
    // smallest split that keeps a wave's slice within Q2_LMAX loads, then widen it while the grid is small
    int ks = 1;
    while (ks < 8 && (G + ks - 1) / ks > Q2_LMAX) ks *= 2;
    while (ks < 8 && (long)tiles * ks < 1024 && G / (ks * 2) >= 4) ks *= 2;
    if (force_ks > 0) ks = force_ks;
    pl.KS = ks;
    pl.WAVES = ks == 8 ? 8 : 4;
    const int wr = pl.WAVES / pl.KS;
    pl.blocks = (tiles + wr - 1) / wr;
    pl.lds = qmv2_lds(pl.MR, N, pl.KS, pl.WAVES, rms).total;
    pl.ok = (K % 16 == 0) && K > 0 && N % 128 == 0 && M >= 1 && M <= 8 && pl.lds <= 150 * 1024 && G >= pl.WAVES;
    return pl;
}

int launch_qmv2_bf16(const QmvArgs &args, int pro, int epi, hipStream_t st, int force_ks = 0);  // qmv2.hip; -1 = not applicable

}  // namespace tl
