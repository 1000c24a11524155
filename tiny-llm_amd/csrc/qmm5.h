// W4A16 (group 128) matmul over WHOLE activation rows for 5 .. 16 decode rows, persistent: one workgroup per CU, no reduction slices.
//
//   out[m,k] = epilogue(sum_n a[m,n] * (q[k,n]*s[k,g] + beta[k,g]))      (reference arithmetic: quantized_matvec_x4_fast,
//   quantized_matmul.metal:441-538 / :510-521; routing of <= 8 rows to the matvec: quantize.py:54-65)
//
// Why a third kernel next to qmv3.h (1-4 rows) and qmm3.h (5-64 rows, K sliced across workgroups): at 5-16 rows the K-sliced
// matmul pays a slice-reduction launch per projection -- ~4.7 us of kernel + a ~1.4 us boundary for a handful of KB, because a
// hand-off between workgroups costs that much by any mechanism on this chip (round 4 measured the in-launch variant: slower, and
// incoherent across XCDs).  Up to 16 rows of 2,560 columns are 80 KiB of bf16: they fit ONE CU's LDS.  So every workgroup stages
// the whole rows once (fused RMSNorm from the producer's sums of squares, as qmm3.h), walks its own range of 16-row weight tiles
// and finishes each tile itself: no slices, no partial planes in HBM, the epilogue (store / residual / SwiGLU) in the same launch.
// The GEMV did the same staging in EVERY one of its 1,216 workgroups (13.2 us at 8 rows); here 256 workgroups do it once each.
//
//   * work unit = 4 quantisation groups of one tile (4 KiB of weights per wave-load set, as qmm3p_kernel); a tile has G / 4 units;
//     the units of a chunk of 8 tiles are dealt round-robin to the 8 waves, so a workgroup with 4.75 tiles (gate|up on 256 CUs)
//     keeps all 8 waves busy instead of 4.75 of them;
//   * a unit's 16 x 16 fp32 sums go to their own LDS slot; after the chunk's barrier one wave per tile adds the tile's slots IN UNIT
//     ORDER (deterministic) and applies the epilogue.  Lane (r, c) owns activation rows 4c .. 4c + 3 of weight row r, as everywhere.
//   * weights: three register sets per wave in rotation -- two units (8 KiB per wave, 64 KiB per CU: what a CU keeps in flight) are on
//     their way while the current one runs.
//   * a workgroup owns at most 8 tiles (one finishing wave each): projections of up to 8 tiles per CU (qkv 1.5, gate|up 4.75).
#pragma once
#include "qmm3.h"

namespace tl {

constexpr int QM5_TILES = 8;  // tiles per chunk (= waves: one finishing wave per tile)

struct Qmm5Args {
    const uint32_t *wt;   // tiled packed weights [K/16][G][64][4]
    const uint32_t *sbt;  // tiled scale|bias<<16 [K/16][G][16]
    const uint16_t *a;    // [M, N] bf16 (PRO_RMSNORM: un-normalised)
    uint16_t *out;        // [M, K] bf16 (EPI_SWIGLU: [M, K/2])
    int M, N, K;
    int tiles_per_wg;     // contiguous tiles per workgroup (the last workgroup may own fewer)
    prof_t *prof;
    const uint16_t *norm_w;    // PRO_RMSNORM
    const float *ss;           // PRO_RMSNORM: [M][ss_n] partial sums of squares of the rows of a
    int ss_n;
    float eps;
    const uint16_t *residual;  // EPI_RESIDUAL [M, K]
    float *ss_out;             // EPI_RESIDUAL, optional: [M][K / 16] sums of squares of the stored bf16 values per (row, tile)
};

template <int G>
__host__ __device__ constexpr size_t qmm5_lds_bytes() {
    // staged rows [16][G * 128] bf16, group sums [G][16], 1 / rms [16], unit slots [QM5_TILES][G / 4][64 lanes][4] fp32
    return (size_t)16 * G * 128 * 2 + (size_t)G * 16 * 4 + 16 * 4 + (size_t)QM5_TILES * (G / 4) * 64 * 16;
}

// Staging of the whole rows in two steps, so that the activation loads are the FIRST thing in the wave's (in-order) load queue and the
// weight stream goes out right behind them -- qmm3_stage_slice issues its loads inside its batch loop, behind whatever the kernel
// fetched before, and hipcc waits for all of that (vmcnt(0)) at the loop's head: the rows would be requested only after 12 KiB of
// weights per wave had come back.  Thread t owns chunks t + 512 j (16 bytes = 8 columns of one row; 16 consecutive lanes = one
// quantisation group of one row); rows >= M are staged as zeros.  Same arithmetic and LDS layout as qmm3_stage_slice.
template <int G, int PRO>
struct Qmm5Staged {
    static constexpr int PER = 16 * G * 16 / (QM3_WAVES * 64);  // chunks per thread (G = 20: 10)
    u32x4 v[PER], gw[PRO == PRO_RMSNORM ? PER : 1];
};
template <int G, int PRO>
__device__ __forceinline__ void qmm5_stage_issue(const Qmm5Args &p, int tid, Qmm5Staged<G, PRO> &st) {
    constexpr int CPR = G * 16, T = QM3_WAVES * 64;
    static_assert(16 * CPR % T == 0, "whole chunks per thread");
#pragma unroll
    for (int j = 0; j < Qmm5Staged<G, PRO>::PER; ++j) {
        const int ch = tid + j * T;
        const int row = ch / CPR, cc = ch - row * CPR;
        const bool ok = row < p.M;
        st.v[j] = *reinterpret_cast<const u32x4 *>(p.a + (ok ? ((size_t)row * p.N + (size_t)cc * 8) : 0));
        if (!ok) st.v[j] = u32x4{0u, 0u, 0u, 0u};
    }
    if constexpr (PRO == PRO_RMSNORM) {
#pragma unroll
        for (int j = 0; j < Qmm5Staged<G, PRO>::PER; ++j) {
            const int ch = tid + j * T;
            const int cc = ch % CPR;
            st.gw[j] = *reinterpret_cast<const u32x4 *>(p.norm_w + (size_t)cc * 8);
        }
    }
}
template <int G, int PRO>
__device__ __forceinline__ void qmm5_stage_finish(const Qmm5Args &p, int tid, const Qmm5Staged<G, PRO> &st, uint16_t *xs, float *xsum,
                                                  const float *s_inv) {
    constexpr int CPR = G * 16, T = QM3_WAVES * 64, XS = G * 128, ROWS = 16;
#pragma unroll
    for (int j = 0; j < Qmm5Staged<G, PRO>::PER; ++j) {
        const int ch = tid + j * T;
        const int row = ch / CPR, cc = ch - row * CPR;
        u32x4 x = st.v[j];
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[2 * e] = BF16::to_float((uint16_t)(x[e] & 0xffffu));
            f[2 * e + 1] = BF16::to_float((uint16_t)(x[e] >> 16));
        }
        if constexpr (PRO == PRO_RMSNORM) {
            const float inv = s_inv[row];  // 1 / rms of the row from its producers' partial sums of squares
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[2 * e] = bf16_round(f[2 * e] * inv * BF16::to_float((uint16_t)(st.gw[j][e] & 0xffffu)));
                f[2 * e + 1] = bf16_round(f[2 * e + 1] * inv * BF16::to_float((uint16_t)(st.gw[j][e] >> 16)));
                x[e] = row < p.M ? BF16::pack2(f[2 * e], f[2 * e + 1]) : 0u;
            }
        }
        const int rr = row & 15;
        const int sw = rr ^ ((rr >= 4 && rr < 12) ? 4 : 0);
        *reinterpret_cast<u32x4 *>(xs + (size_t)row * XS + (size_t)((cc & ~15) | ((cc & 15) ^ sw)) * 8) = x;
        float sum = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
        sum = group16_sum(sum);
        if ((cc & 15) == 0) xsum[(cc >> 4) * ROWS + row] = sum;
    }
}

template <int G, int PRO, int EPI>
__global__ __launch_bounds__(QM3_WAVES * 64) void qmm5_kernel(const Qmm5Args p) {
    static_assert(G % 4 == 0 && G >= 4, "whole units of 4 groups");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int UPT = G / 4;        // units per tile
    constexpr int ROWS = 16;
    constexpr int XS = G * 128;       // staged row stride (elements)
    const prof_t prof_t0 = prof_begin(p.prof);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r = lane & 15, c = lane >> 4;
    const int K = p.K, tiles = K >> 4;
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);
    float *xsum = reinterpret_cast<float *>(smem + (size_t)ROWS * XS * 2);  // [G][ROWS]
    float *s_inv = xsum + G * ROWS;                                          // [ROWS]
    f32x4 *slots = reinterpret_cast<f32x4 *>(s_inv + ROWS);                  // [QM5_TILES * UPT][64]

    const int t_first = blockIdx.x * p.tiles_per_wg;
    const int nt = max(0, min(tiles, t_first + p.tiles_per_wg) - t_first);  // tiles of this workgroup (<= QM5_TILES), uniform
    const int nu = nt * UPT;                                                // its units; wave w takes units w, w + 8, ...
    const int n_mine = wave < nu ? (nu - wave + QM3_WAVES - 1) / QM3_WAVES : 0;  // <= UPT

    // the staging's 1 / rms partials first (they gate everything staged; vector loads return in issue order)
    Qmm3Args sa{};
    sa.a = p.a, sa.M = p.M, sa.N = p.N, sa.K = p.K, sa.norm_w = p.norm_w, sa.ss = p.ss, sa.ss_n = p.ss_n, sa.eps = p.eps;
    Qmm3RowSS<1> rss;
    if constexpr (PRO == PRO_RMSNORM) qmm3_row_ss_issue<1>(sa, tid, rss);
    Qmm5Staged<G, PRO> staged;
    qmm5_stage_issue<G, PRO>(p, tid, staged);  // the rows (and norm weights) go out BEFORE the weights: they are needed first
    __builtin_amdgcn_sched_barrier(0);

    // ---- weights: three register sets in ROTATION (A, B, C, A, B): a set is refilled right behind the unit that consumed it, so
    // two units are always on their way while one runs.  (Handing sets over by register copies does not work: a copy of a set whose
    // loads are still in flight waits for them -- the first version of this kernel did that and paid a memory latency per unit:
    // gate|up 14.1 us.)  An invalid position re-reads the wave's first block: every load is unconditional, from a clamped address.
    const uint32_t lane_w = (uint32_t)lane * 16u, lane_s = (uint32_t)r * 4u;
    auto fetch = [&](u32x4(&wq)[4], uint32_t(&sq)[4], int kk) {
        const int gl = kk < n_mine ? wave + QM3_WAVES * kk : 0;  // uniform
        const int tile = __builtin_amdgcn_readfirstlane(min(t_first + gl / UPT, tiles - 1));
        const int unit = gl % UPT;
        const char *wbase = reinterpret_cast<const char *>(p.wt) + ((size_t)tile * G + unit * 4) * 1024;
        const char *sbase = reinterpret_cast<const char *>(p.sbt) + ((size_t)tile * G + unit * 4) * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) sq[i] = *reinterpret_cast<const uint32_t *>(sbase + (lane_s + (uint32_t)i * 64u));
#pragma unroll
        for (int i = 0; i < 4; ++i) wq[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wbase + (lane_w + (uint32_t)i * 1024u)));
    };
    u32x4 wA[4], wB[4], wC[4];
    uint32_t sA[4], sB[4], sC[4];
    fetch(wA, sA, 0);
    fetch(wB, sB, 1);
    fetch(wC, sC, 2);
    __builtin_amdgcn_sched_barrier(0);

    // ---- whole activation rows -> LDS, once per workgroup ----------------------------------------------------------------------
    if constexpr (PRO == PRO_RMSNORM) {
        qmm3_row_ss_finish<1>(sa, tid, rss, s_inv);
        __syncthreads();  // s_inv complete
    }
    qmm5_stage_finish<G, PRO>(p, tid, staged, xs, xsum, s_inv);
    __syncthreads();

    const int swr = r ^ ((r >= 4 && r < 12) ? 4 : 0);
    int xoff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) xoff[t] = ((4 * c + t) ^ swr) * 8;
    const uint16_t *xrow = xs + r * XS;
    uint32_t nib_mask = 0x000f000fu;
    uint32_t magic = 0x43004300u;
    asm volatile("" : "+s"(nib_mask));
    asm volatile("" : "+v"(magic));

    // ---- one unit: 4 groups x 4 k-steps of MFMA against the staged rows; its 16 x 16 sums go to LDS slot `gl` ------------------------
    auto run = [&](const u32x4(&wq)[4], const uint32_t(&sq)[4], int kk) {
        const int gl = wave + QM3_WAVES * kk;
        const int unit = gl % UPT;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint16_t *xu = xrow + unit * 512;
        const float *xsl = xsum + (unit * 4) * ROWS + 4 * c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const u32x4 bq = unpack_w4_bf16(wq[i][t], nib_mask, magic);
                const u32x4 ax = *reinterpret_cast<const u32x4 *>(xu + i * 128 + xoff[t]);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ax), __builtin_bit_cast(bf16x8_t, bq), d, 0, 0, 0);
            }
            const float sc = __uint_as_float(sq[i] << 16);
            const float be = __uint_as_float(sq[i] & 0xffff0000u) - 128.0f * sc;
            const f32x4 xg = *reinterpret_cast<const f32x4 *>(xsl + i * ROWS);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += sc * d[j] + be * xg[j];
        }
        slots[gl * 64 + lane] = acc;
    };
    // positions 0 .. UPT - 1 of this wave (UPT <= 5), the sets in rotation; every branch is wave-uniform
    if (0 < n_mine) run(wA, sA, 0);
    if constexpr (UPT > 3) fetch(wA, sA, 3);
    if (1 < n_mine) run(wB, sB, 1);
    if constexpr (UPT > 4) fetch(wB, sB, 4);
    if (2 < n_mine) run(wC, sC, 2);
    if constexpr (UPT > 3) {
        if (3 < n_mine) run(wA, sA, 3);
    }
    if constexpr (UPT > 4) {
        if (4 < n_mine) run(wB, sB, 4);
    }
    static_assert(UPT <= 5, "five positions per wave are written out");
    __syncthreads();  // the workgroup's unit sums are in LDS
    if (wave < nt) {  // one wave finishes one tile: its UPT slots in unit order, then the epilogue
        const int tile = t_first + wave;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
            const f32x4 v = slots[(wave * UPT + u) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += v[j];
        }
        const int col = (tile << 4) + r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 4 * c + j;
            const bool live = row < p.M;
            if constexpr (EPI == EPI_SWIGLU) {  // rows interleaved: even column = gate_i, odd = up_i; the even lane stores
                const float gv = bf16_round(acc[j]);
                const float uv = lane_xor1(gv);
                if (live && (r & 1) == 0)
                    p.out[(size_t)row * (K >> 1) + (col >> 1)] = BF16::from_float((gv / (1.0f + expf(-gv))) * uv);
            } else if constexpr (EPI == EPI_RESIDUAL) {
                float sq = 0.f;
                if (live) {
                    const size_t o = (size_t)row * K + col;
                    const uint16_t ov = BF16::from_float(BF16::to_float(p.residual[o]) + bf16_round(acc[j]));
                    p.out[o] = ov;
                    sq = BF16::to_float(ov) * BF16::to_float(ov);
                }
                if (p.ss_out) {  // uniform
                    sq = group16_sum(sq);
                    if (r == 0 && live) p.ss_out[(size_t)row * tiles + tile] = sq;
                }
            } else {
                if (live) p.out[(size_t)row * K + col] = BF16::from_float(acc[j]);
            }
        }
    }
    prof_end(p.prof, prof_t0);
}

struct Qmm5Plan {
    int G, grid, tiles_per_wg;
    size_t lds;
    bool ok;
};
// Shapes the full-row kernel takes: up to 16 rows, a reduction of 8 / 16 / 20 groups (1,024 / 2,048 / 2,560 columns: rows + unit slots
// fit 160 KiB of LDS), whole 16-row tiles, at most 8 of them per CU.
inline Qmm5Plan qmm5_plan(int M, int N, int K) {
    Qmm5Plan pl{};
    pl.G = N / 128;
    pl.ok = M >= 1 && M <= 16 && N > 0 && N % 128 == 0 && K > 0 && K % 16 == 0 && (pl.G == 8 || pl.G == 16 || pl.G == 20);
    if (!pl.ok) return pl;
    const int tiles = K / 16, ncu = qmm3_num_cus();
    pl.tiles_per_wg = (tiles + ncu - 1) / ncu;
    pl.grid = (tiles + pl.tiles_per_wg - 1) / pl.tiles_per_wg;
    pl.ok = pl.tiles_per_wg <= QM5_TILES;  // one finishing wave per tile: up to 8 tiles per CU (a 151,936-row head has 37)
    pl.lds = pl.G == 8 ? qmm5_lds_bytes<8>() : (pl.G == 16 ? qmm5_lds_bytes<16>() : qmm5_lds_bytes<20>());
    return pl;
}

// qmm3.hip
int launch_qmm5_bf16(const Qmm5Args &args, int pro, int epi, hipStream_t st);  // -1: shape not taken, -2: no such variant

}  // namespace tl
