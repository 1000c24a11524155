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
//   * weights: three register sets per wave -- two units (8 KiB per wave, 64 KiB per CU: what a CU keeps in flight) are on their way
//     while the current one runs; 40 v_mov per unit hand them over.
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
    const int t_count = max(0, min(tiles, t_first + p.tiles_per_wg) - t_first);  // uniform
    const int n_chunks = (t_count + QM5_TILES - 1) / QM5_TILES;

    // the staging's 1 / rms partials first (they gate everything staged; vector loads return in issue order)
    Qmm3Args sa{};
    sa.a = p.a, sa.M = p.M, sa.N = p.N, sa.K = p.K, sa.norm_w = p.norm_w, sa.ss = p.ss, sa.ss_n = p.ss_n, sa.eps = p.eps;
    Qmm3RowSS<1> rss;
    if constexpr (PRO == PRO_RMSNORM) qmm3_row_ss_issue<1>(sa, tid, rss);

    // ---- this wave's units: chunk ch, position kk -> local unit wave + 8 kk of the chunk's (tiles x UPT) units --------------------
    // (a wave-uniform walk: next_unit advances to the wave's next unit or past the end)
    auto chunk_units = [&](int ch) { return min(QM5_TILES, t_count - ch * QM5_TILES) * UPT; };
    auto valid = [&](int ch, int kk) { return ch < n_chunks && wave + QM3_WAVES * kk < chunk_units(ch); };
    auto next_unit = [&](int &ch, int &kk) {
        ++kk;
        while (ch < n_chunks && !(wave + QM3_WAVES * kk < chunk_units(ch))) ++ch, kk = 0;
    };
    const uint32_t lane_w = (uint32_t)lane * 16u, lane_s = (uint32_t)r * 4u;
    auto fetch = [&](u32x4(&wq)[4], uint32_t(&sq)[4], int ch, int kk) {
        const bool ok = valid(ch, kk);  // uniform; an invalid unit re-reads the workgroup's first one (never used)
        const int gl = ok ? wave + QM3_WAVES * kk : 0;
        const int tile = __builtin_amdgcn_readfirstlane(min(t_first + (ok ? ch : 0) * QM5_TILES + gl / UPT, tiles - 1));
        const int unit = gl % UPT;
        const char *wbase = reinterpret_cast<const char *>(p.wt) + ((size_t)tile * G + unit * 4) * 1024;
        const char *sbase = reinterpret_cast<const char *>(p.sbt) + ((size_t)tile * G + unit * 4) * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) sq[i] = *reinterpret_cast<const uint32_t *>(sbase + (lane_s + (uint32_t)i * 64u));
#pragma unroll
        for (int i = 0; i < 4; ++i) wq[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wbase + (lane_w + (uint32_t)i * 1024u)));
    };
    u32x4 wcur[4], wnxt[4], wnx2[4];
    uint32_t scur[4], snxt[4], snx2[4];
    int f_ch = 0, f_kk = 0;  // the unit the NEXT fetch takes
    if (!valid(f_ch, f_kk)) next_unit(f_ch, f_kk);  // (only when the first chunk has fewer units than waves)
    fetch(wcur, scur, f_ch, f_kk);
    next_unit(f_ch, f_kk);
    fetch(wnxt, snxt, f_ch, f_kk);
    next_unit(f_ch, f_kk);
    fetch(wnx2, snx2, f_ch, f_kk);
    next_unit(f_ch, f_kk);
    __builtin_amdgcn_sched_barrier(0);

    // ---- whole activation rows -> LDS, once per workgroup ----------------------------------------------------------------------
    if constexpr (PRO == PRO_RMSNORM) qmm3_row_ss_finish<1>(sa, tid, rss, s_inv);
    qmm3_stage_slice<1, G, PRO, QMM3P_SB>(sa, 0, G, xs, xsum, tid, s_inv);
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

    for (int ch = 0; ch < n_chunks; ++ch) {
        const int nt = min(QM5_TILES, t_count - ch * QM5_TILES);  // tiles of this chunk
        const int nu = nt * UPT;
        for (int kk = 0; wave + QM3_WAVES * kk < nu; ++kk) {  // uniform per wave
            const int gl = wave + QM3_WAVES * kk;
            const int unit = gl % UPT;
            // ---- one unit: 4 groups x 4 k-steps of MFMA against the staged rows --------------------------------------------------
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            const uint16_t *xu = xrow + unit * 512;
            const float *xsl = xsum + (unit * 4) * ROWS + 4 * c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const u32x4 bq = unpack_w4_bf16(wcur[i][t], nib_mask, magic);
                    const u32x4 ax = *reinterpret_cast<const u32x4 *>(xu + i * 128 + xoff[t]);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ax), __builtin_bit_cast(bf16x8_t, bq), d, 0, 0, 0);
                }
                const float sc = __uint_as_float(scur[i] << 16);
                const float be = __uint_as_float(scur[i] & 0xffff0000u) - 128.0f * sc;
                const f32x4 xg = *reinterpret_cast<const f32x4 *>(xsl + i * ROWS);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += sc * d[j] + be * xg[j];
            }
            slots[gl * 64 + lane] = acc;
            // hand the prefetched sets over and put the unit behind them in flight
#pragma unroll
            for (int i = 0; i < 4; ++i) wcur[i] = wnxt[i], scur[i] = snxt[i], wnxt[i] = wnx2[i], snxt[i] = snx2[i];
            fetch(wnx2, snx2, f_ch, f_kk);
            next_unit(f_ch, f_kk);
        }
        __syncthreads();  // the chunk's unit sums are in LDS
        if (wave < nt) {  // one wave finishes one tile: its UPT slots in unit order, then the epilogue
            const int tile = t_first + ch * QM5_TILES + wave;
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
        __syncthreads();  // the slots are free for the next chunk
    }
    prof_end(p.prof, prof_t0);
}

struct Qmm5Plan {
    int G, grid, tiles_per_wg;
    size_t lds;
    bool ok;
};
// Shapes the full-row kernel takes: up to 16 rows, a reduction of 8 / 16 / 20 groups (1,024 / 2,048 / 2,560 columns: rows + unit slots
// fit 160 KiB of LDS), whole 16-row tiles.
inline Qmm5Plan qmm5_plan(int M, int N, int K) {
    Qmm5Plan pl{};
    pl.G = N / 128;
    pl.ok = M >= 1 && M <= 16 && N > 0 && N % 128 == 0 && K > 0 && K % 16 == 0 && (pl.G == 8 || pl.G == 16 || pl.G == 20);
    if (!pl.ok) return pl;
    const int tiles = K / 16, ncu = qmm3_num_cus();
    pl.tiles_per_wg = (tiles + ncu - 1) / ncu;
    pl.grid = (tiles + pl.tiles_per_wg - 1) / pl.tiles_per_wg;
    pl.lds = pl.G == 8 ? qmm5_lds_bytes<8>() : (pl.G == 16 ? qmm5_lds_bytes<16>() : qmm5_lds_bytes<20>());
    return pl;
}

// qmm3.hip
int launch_qmm5_bf16(const Qmm5Args &args, int pro, int epi, hipStream_t st);  // -1: shape not taken, -2: no such variant

}  // namespace tl
