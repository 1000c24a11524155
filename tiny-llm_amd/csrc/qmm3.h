// W4A16 (group 128) SKINNY matmul for batched decode (9 .. 64 activation rows) over the engine's tiled weight layout.
//
//   out[m,k] = sum_n a[m,n] * (q[k,n]*s[k,g] + beta[k,g]),  algebraic form sum_g (s_g sum a q + beta_g sum a) in fp32
//   (reference: quantized_matvec_x4_fast, quantized_matmul.metal:441-538 / :510-521 -- the decode GEMV semantics, so a
//   sequence decodes to the same values whether it runs alone or in a batch).
//
// Why a second kernel next to qmv3.h: the GEMV stages (and, fused, normalises) ALL activation rows in every workgroup.
// With M rows that prologue costs M times more while the weight stream stays the same; at M = 8 it already takes
// 60 % of the kernel (profiles/r01_labs).  Here the reduction dimension is SLICED ACROSS WORKGROUPS instead:
//
//   grid.x = groups of 8*TW weight tiles (16 rows each), grid.y = slices of LM quantisation groups (128 columns each).
//   A workgroup (8 waves) stages only its slice of the activations, [16 MB rows][LM*128] bf16 (<= 83 KiB of LDS), once;
//   each wave owns TW tiles, holds their LM 1-KiB weight blocks in registers (issued before the staging, as in qmv3),
//   and runs v_mfma_f32_16x16x32_bf16 against every 16-row block of the slice: the nibble unpack (7 VALU per 8 weights)
//   is amortised over MB row blocks, and with TW = 2 each LDS fragment feeds two MFMAs.
//   Partial sums go to an fp32 workspace [slices][M][K]; qmm3_reduce_kernel adds the slices in a fixed order and applies
//   the epilogue (store / residual add / SwiGLU over interleaved gate-up rows), so results do not depend on timing.
//
// RMSNorm: a slice cannot see the whole row, but the kernels that PRODUCE the row can -- the slice reduction of the previous
// projection and the embedding gather emit deterministic per-row partial sums of squares ([row][QM3_SS] floats, summed here
// in a fixed tree), and the staging pass normalises its slice with them (PRO_RMSNORM: bf16(x * inv * w), the reference's
// rounding point).  Where no such partials exist (the producer was a GEMV) the engine runs rms_norm as its own launch.
#pragma once
#include "common.h"
#include "qmv.h"
#include "qmv3.h"

namespace tl {

#ifndef QMM3_ABL
#define QMM3_ABL 0  // tools/lab/qmm3_lab only: 1 no MFMA, 2 no activation staging, 4 no partial stores, 8 no group-sum arithmetic,
                    // 64 per-wave phase stamps of the persistent kernel into args.prof
#endif
constexpr int QM3_WAVES = 8;
// The staged rows are NOT padded: a 128-element group is exactly one 64-bank row, and the 16-byte chunks inside it are
// XOR-swizzled by the row so that every ds_read_b128 service group ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS) hits 16
// different bank quads -- chunk j of row r sits at j ^ sw(r), sw(r) = (r & 15) ^ (4 if 4 <= (r & 15) < 12).  A padded row
// stride cannot do that for this lane -> (row, k-block) map (r02 lab: gate|up at 64 rows 29.1 -> 27.5 us).
constexpr int QM3_PAD = 0;  // bf16 elements of padding per staged activation row
constexpr int QM3_SS = 8;   // partial sums of squares kept per activation row (unused entries are zero)

struct Qmm3Args {
    const uint32_t *wt;   // tiled packed weights [K/16][G][64][4]
    const uint32_t *sbt;  // tiled scale|bias<<16 [K/16][G][16]
    const uint16_t *a;    // [M, N] bf16 (already normalised where the projection has a norm)
    float *partial;       // [slices][M][K] fp32
    int M, N, K;
    prof_t *prof;
    // PRO_RMSNORM only: a holds the UN-normalised rows; ss [M][ss_n] partial sums of squares of each row (ss_n a multiple of 4, at
    // most 256: 8 from the embedding kernels / the slice reduction, rows / 16 from a GEMV producer)
    const uint16_t *norm_w;
    const float *ss;
    float eps;
    int ss_n;
};

__host__ __device__ inline size_t qmm3_lds_bytes(int MB, int LM) {
    return (size_t)MB * 16 * (LM * 128 + QM3_PAD) * 2 + (size_t)LM * MB * 16 * 4 + (size_t)MB * 16 * 4;  // rows, group sums, 1 / rms per row
}

// Fused RMSNorm, step 1 of 2 (first thing a kernel does: these few loads gate the normalisation of everything staged, and vector
// loads return in issue order): 16 lanes per activation row fetch the row's partial sums of squares, lane l the float4s at
// 4 l + 64 k.  Rows go in passes of 32 (512 threads).
constexpr int QM3_SS_MAX = 256;
template <int MB>
struct Qmm3RowSS {
    static constexpr int PASSES = (MB * 16 + 31) / 32;
    f32x4 v[PASSES][QM3_SS_MAX / 64];
};
template <int MB>
__device__ __forceinline__ void qmm3_row_ss_issue(const Qmm3Args &p, int tid, Qmm3RowSS<MB> &r) {
#pragma unroll
    for (int ps = 0; ps < Qmm3RowSS<MB>::PASSES; ++ps) {
        const int row = ps * 32 + (tid >> 4);
        const bool ok = row < p.M;
        const float *src = p.ss + (size_t)(ok ? row : 0) * p.ss_n;
#pragma unroll
        for (int k = 0; k < QM3_SS_MAX / 64; ++k) {
            const int idx = 4 * (tid & 15) + 64 * k;
            const bool okk = ok && idx < p.ss_n;
            r.v[ps][k] = *reinterpret_cast<const f32x4 *>(src + (okk ? idx : 0));
            if (!okk) r.v[ps][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}
// step 2: fixed-order sums, 16 lanes meet by DPP rotations, 1 / rms of every staged row -> s_inv [MB * 16] in LDS (a barrier follows
// in the staging pass, behind its own loads)
template <int MB>
__device__ __forceinline__ void qmm3_row_ss_finish(const Qmm3Args &p, int tid, const Qmm3RowSS<MB> &r, float *s_inv) {
#pragma unroll
    for (int ps = 0; ps < Qmm3RowSS<MB>::PASSES; ++ps) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < QM3_SS_MAX / 64; ++k) tot += (r.v[ps][k][0] + r.v[ps][k][1]) + (r.v[ps][k][2] + r.v[ps][k][3]);
        tot = group16_sum(tot);
        const int row = ps * 32 + (tid >> 4);
        if ((tid & 15) == 0 && row < MB * 16) s_inv[row] = rsqrtf(tot / (float)p.N + p.eps);
    }
}

// Activation slice [MB*16 rows][LM groups] -> LDS (XOR-swizzled 16-byte chunks, see QM3_PAD) + per-(group,row) sums, by all
// QM3_WAVES*64 threads of the workgroup.  chunk = 8 consecutive elements; a row of the slice has LM*16 chunks; 16 consecutive
// lanes cover one group of one row.  The loads of SB chunks per thread are issued together (one dependent round trip per
// chunk cost ~1 us each).  PRO_RMSNORM: the rows are normalised on the way in (bf16(x * inv * w), the reference's rounding point).
template <int MB, int LM, int SBMAX>
struct Qmm3Stage {
    static constexpr int T = QM3_WAVES * 64, ROWS = MB * 16, CPR = LM * 16, CHUNKS = ROWS * CPR;
    static constexpr int ITER = (CHUNKS + T - 1) / T;
    static constexpr int SB = ITER < SBMAX ? ITER : SBMAX;
    u32x4 v[SB], gw[SB];  // one batch of a thread's chunks (and their norm weights) in flight
    bool ok[SB];
};
// the loads of the batch of chunks that starts at iteration it0 (unconditional, from clamped addresses)
template <int MB, int LM, int PRO, int SBMAX>
__device__ __forceinline__ void qmm3_stage_issue(const Qmm3Args &p, int g0, int gn, int tid, int it0, Qmm3Stage<MB, LM, SBMAX> &st) {
    using S = Qmm3Stage<MB, LM, SBMAX>;
#pragma unroll
    for (int j = 0; j < S::SB; ++j) {
        const int ch = min(tid + (it0 + j) * S::T, S::CHUNKS - 1);
        const int row = ch / S::CPR;
        const int cc = ch - row * S::CPR;
        st.ok[j] = tid + (it0 + j) * S::T < S::CHUNKS && row < p.M && (cc >> 4) < gn && !(QMM3_ABL & 2);
        st.v[j] = *reinterpret_cast<const u32x4 *>(p.a + (st.ok[j] ? ((size_t)row * p.N + (size_t)g0 * 128 + (size_t)cc * 8) : 0));
        if constexpr (PRO == PRO_RMSNORM)
            st.gw[j] = *reinterpret_cast<const u32x4 *>(p.norm_w + (st.ok[j] ? ((size_t)g0 * 128 + (size_t)cc * 8) : 0));
    }
}
// FIRST_OUT: the first batch's loads were issued by the caller (qmm3_stage_issue at it0 = 0) ahead of the weight stream -- vector
// loads return in issue order and the rows are needed first (round 4: staged rows used to arrive together with 64 KiB of weights)
template <int MB, int LM, int PRO, int SBMAX = 5, bool FIRST_OUT = false>
__device__ __forceinline__ void qmm3_stage_slice(const Qmm3Args &p, int g0, int gn, uint16_t *xs, float *xsum, int tid, const float *s_inv,
                                                 Qmm3Stage<MB, LM, SBMAX> &st) {
    using S = Qmm3Stage<MB, LM, SBMAX>;
    constexpr int T = S::T, ROWS = S::ROWS, CPR = S::CPR, CHUNKS = S::CHUNKS, ITER = S::ITER, SB = S::SB;
    constexpr int XS = LM * 128 + QM3_PAD;
    for (int it0 = 0; it0 < ITER; it0 += SB) {
        if (!(FIRST_OUT && it0 == 0)) qmm3_stage_issue<MB, LM, PRO, SBMAX>(p, g0, gn, tid, it0, st);
        u32x4(&v)[SB] = st.v;
        u32x4(&gw)[SB] = st.gw;
        bool(&ok)[SB] = st.ok;
        if constexpr (PRO == PRO_RMSNORM) {
            if (it0 == 0) __syncthreads();  // s_inv (qmm3_row_ss_finish) is complete; this batch's loads are already out
        }
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            const int chu = tid + (it0 + j) * T;
            if (chu >= CHUNKS) break;  // CHUNKS is a multiple of 64: whole waves drop out, a 16-lane group stays in one row/group
            const int row = chu / CPR;
            const int cc = chu - row * CPR;
            const int g = cc >> 4;
            u32x4 x = ok[j] ? v[j] : u32x4{0u, 0u, 0u, 0u};
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
                    f[2 * e] = bf16_round(f[2 * e] * inv * BF16::to_float((uint16_t)(gw[j][e] & 0xffffu)));
                    f[2 * e + 1] = bf16_round(f[2 * e + 1] * inv * BF16::to_float((uint16_t)(gw[j][e] >> 16)));
                    x[e] = ok[j] ? BF16::pack2(f[2 * e], f[2 * e + 1]) : 0u;
                }
            }
            {
                const int rr = row & 15;
                const int sw = rr ^ ((rr >= 4 && rr < 12) ? 4 : 0);
                *reinterpret_cast<u32x4 *>(xs + (size_t)row * XS + (size_t)((cc & ~15) | ((cc & 15) ^ sw)) * 8) = x;
            }
            if constexpr (QMM3_ABL & 8) {
                if ((cc & 15) == 0) xsum[g * ROWS + row] = f[0];
                continue;
            }
            float sum = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
            sum = group16_sum(sum);
            if ((cc & 15) == 0) xsum[g * ROWS + row] = sum;
        }
    }
}

// SF = the first batch of the staging's row loads goes out BEFORE the weights (the launcher sets it for MB == 1: qmm3.hip)
template <int MB, int TW, int LM, int PRO = PRO_NONE, bool SF = false>
__global__ __launch_bounds__(QM3_WAVES * 64) void qmm3_kernel(const Qmm3Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = MB * 16;
    constexpr int XS = LM * 128 + QM3_PAD;  // staged row stride (elements)
    const prof_t prof_t0 = prof_begin(p.prof);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r = lane & 15, c = lane >> 4;
    const int N = p.N, K = p.K, G = N >> 7;
    const int tiles = K >> 4;
    const int slice = blockIdx.y;
    const int g0 = slice * LM;
    const int gn = min(LM, G - g0);  // groups this slice really has (the last slice may be short)
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);
    float *xsum = reinterpret_cast<float *>(smem + (size_t)ROWS * XS * 2);  // [LM][ROWS]
    float *s_inv = xsum + LM * ROWS;                                         // [ROWS]
    Qmm3RowSS<MB> rss;
    if constexpr (PRO == PRO_RMSNORM) qmm3_row_ss_issue<MB>(p, tid, rss);  // first: they gate the staging, and loads return in order
    Qmm3Stage<MB, LM, 5> stage;
    if constexpr (SF) {
        qmm3_stage_issue<MB, LM, PRO, 5>(p, g0, gn, tid, 0, stage);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- 1. weights of this wave's tiles: everything in flight before the staging ----------------------------------
    u32x4 wq[TW][LM];
    uint32_t sq[TW][LM];
    int tile[TW];
#pragma unroll
    for (int tw = 0; tw < TW; ++tw) {
        tile[tw] = (blockIdx.x * QM3_WAVES + wave) * TW + tw;
        const int tc = min(tile[tw], tiles - 1);
        const uint32_t *sp = p.sbt + (size_t)tc * G * 16 + r;
        const u32x4 *wp = reinterpret_cast<const u32x4 *>(p.wt) + (size_t)tc * G * 64 + lane;
#pragma unroll
        for (int i = 0; i < LM; ++i) sq[tw][i] = sp[(size_t)min(g0 + i, G - 1) * 16];
#pragma unroll
        for (int i = 0; i < LM; ++i) wq[tw][i] = __builtin_nontemporal_load(wp + (size_t)min(g0 + i, G - 1) * 64);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- 2. activation slice -> LDS, per-(group,row) sums ------------------------------------------------------------
    if constexpr (PRO == PRO_RMSNORM) qmm3_row_ss_finish<MB>(p, tid, rss, s_inv);
    qmm3_stage_slice<MB, LM, PRO, 5, SF>(p, g0, gn, xs, xsum, tid, s_inv, stage);
    __syncthreads();

    // ---- 3. MFMA over the slice --------------------------------------------------------------------------------------
    f32x4 acc[TW][MB];
#pragma unroll
    for (int tw = 0; tw < TW; ++tw)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[tw][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int swr = r ^ ((r >= 4 && r < 12) ? 4 : 0);
    const uint16_t *xrow = xs + (size_t)r * XS;
    int xoff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) xoff[t] = ((4 * c + t) ^ swr) * 8;
    uint32_t nib_mask = 0x000f000fu;
    uint32_t magic = 0x43004300u;
    asm volatile("" : "+s"(nib_mask));  // opaque constants (qmv3.h unpack_w4_bf16): one v_and_or_b32 per unpacked pair
    asm volatile("" : "+v"(magic));
#pragma unroll
    for (int i = 0; i < LM; ++i) {
        f32x4 d[TW][MB];
#pragma unroll
        for (int tw = 0; tw < TW; ++tw)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) d[tw][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            u32x4 bq[TW];
#pragma unroll
            for (int tw = 0; tw < TW; ++tw) bq[tw] = unpack_w4_bf16(wq[tw][i][t], nib_mask, magic);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const u32x4 ax = *reinterpret_cast<const u32x4 *>(xrow + xoff[t] + (size_t)mb * 16 * XS + i * 128);
#pragma unroll
                for (int tw = 0; tw < TW; ++tw) {
                    if constexpr (QMM3_ABL & 1) d[tw][mb][t] += __uint_as_float((ax[0] ^ bq[tw][1]) & 0x3f800000u);
                    else
                    d[tw][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ax),
                                                                        __builtin_bit_cast(bf16x8_t, bq[tw]), d[tw][mb], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int tw = 0; tw < TW; ++tw) {
            const uint32_t sw = i < gn ? sq[tw][i] : 0u;  // groups past the end of the row contribute nothing
            const float sc = __uint_as_float(sw << 16);
            const float be = __uint_as_float(sw & 0xffff0000u) - 128.0f * sc;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const f32x4 xg = *reinterpret_cast<const f32x4 *>(xsum + i * ROWS + mb * 16 + 4 * c);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[tw][mb][j] += sc * d[tw][mb][j] + be * xg[j];
            }
        }
    }

    // ---- 4. partial sums: lane (weight row r, c) holds activation rows 16 mb + 4c + j --------------------------------
#pragma unroll
    for (int tw = 0; tw < TW; ++tw) {
        if (tile[tw] >= tiles) continue;
        const int ocol = (tile[tw] << 4) + r;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = mb * 16 + 4 * c + j;
                if (row < p.M && (!(QMM3_ABL & 4) || acc[tw][mb][j] == 123.f)) act_store(&p.partial[((size_t)slice * p.M + row) * K + ocol], acc[tw][mb][j]);
            }
    }
    prof_end(p.prof, prof_t0);
}

// Persistent variant: ONE workgroup per CU.  A workgroup stages its activation slice once and its 8 waves then walk the 16-row
// weight tiles of the workgroup's tile range (tile = first + wave + 8 j) -- the weight stream never stops for a staging pass or a
// workgroup hand-over (the one-shot grid above spends 2.4 rounds of load -> stage -> compute -> store per CU on gate|up at 64
// rows).  The weights move in UNITS of 4 quantisation groups of one tile (4 KiB per wave): two register sets, one in flight
// while the other runs through the MFMAs.  A slice is NU units wide: with NU = 2 a tile's accumulators live across its two
// units, so the slice is 8 groups (128 KiB of LDS at 64 rows) while the register budget stays that of 4 -- 3 slices instead of
// 5 on the 2,560-column projections, i.e. 40 % fewer fp32 partials to write here and to read in the slice reduction (at 64
// rows 5 slices of partials weigh as much as the weights themselves).  A short last slice (<= 4 groups left) runs the NU = 1
// body and gets proportionally fewer workgroups.  Same arithmetic, same partial layout, same slice reduction as qmm3_kernel.
#ifndef QMM3P_SB
#define QMM3P_SB 8  // staging chunks in flight per thread
#endif
template <int MB, int NU, int PRO, bool SF>
__device__ __forceinline__ void qmm3p_body(const Qmm3Args &p, char *smem, const int slice, const int g0, const int gn, const int wg,
                                           const int tiles_per_wg) {
    constexpr int LM = 4 * NU;
    constexpr int ROWS = MB * 16;
    constexpr int XS = LM * 128 + QM3_PAD;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r = lane & 15, c = lane >> 4;
    const int K = p.K, G = p.N >> 7;
    const int tiles = K >> 4;
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);
    float *xsum = reinterpret_cast<float *>(smem + (size_t)ROWS * XS * 2);  // [LM][ROWS]
    float *s_inv = xsum + LM * ROWS;                                         // [ROWS]
    Qmm3RowSS<MB> rss;
    if constexpr (PRO == PRO_RMSNORM) qmm3_row_ss_issue<MB>(p, tid, rss);  // first: they gate the staging, and loads return in order
    Qmm3Stage<MB, LM, QMM3P_SB> stage;
    if constexpr (SF) {
        qmm3_stage_issue<MB, LM, PRO, QMM3P_SB>(p, g0, gn, tid, 0, stage);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int first = wg * tiles_per_wg + wave;
    const int last = min(tiles, (wg + 1) * tiles_per_wg);
    const int n_tiles = first < last ? (last - first + QM3_WAVES - 1) / QM3_WAVES : 0;  // wave-uniform
#if QMM3_ABL & 64  // lab: per-wave phase stamps (start, loads issued, staged, after each unit ...) into p.prof[(wg*8+wave)*16 + k]
    unsigned long long stamps[16];
    int n_stamps = 0;
#define QM3_STAMP() do { if (n_stamps < 16) stamps[n_stamps++] = wall_clock64(); } while (0)
#else
#define QM3_STAMP() do { } while (0)
#endif
    QM3_STAMP();

    u32x4 wqa[4], wqb[4];
    uint32_t sqa[4], sqb[4];
    // uniform (SGPR) block base + per-lane 32-bit offset + group offset: one address register per stream instead of one per load
    const uint32_t lane_w = (uint32_t)lane * 16u, lane_s = (uint32_t)r * 4u;
    const uint32_t st_off = ((uint32_t)(4 * c) * (uint32_t)K + (uint32_t)r) * 4u;  // partial-store offset of this lane inside a row block
    const __amdgpu_buffer_rsrc_t prs =  // this slice's [M][K] partial plane: stores at or beyond M*K*4 bytes are dropped by the range check
        __builtin_amdgcn_make_buffer_rsrc(p.partial + (size_t)slice * p.M * K, 0, (int)((uint32_t)p.M * (uint32_t)K * 4u), 0x00020000);
    auto fetch = [&](u32x4(&wq)[4], uint32_t(&sq)[4], int tile, int unit) {
        const int tc = __builtin_amdgcn_readfirstlane(min(tile, tiles - 1));
        const char *wbase = reinterpret_cast<const char *>(p.wt) + ((size_t)tc * G + g0) * 1024;
        const char *sbase = reinterpret_cast<const char *>(p.sbt) + ((size_t)tc * G + g0) * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t gi = (uint32_t)min(unit * 4 + i, gn - 1);  // past the slice's last group: re-read it (scaled by zero later)
            sq[i] = *reinterpret_cast<const uint32_t *>(sbase + (lane_s + gi * 64u));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t gi = (uint32_t)min(unit * 4 + i, gn - 1);
            wq[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wbase + (lane_w + gi * 1024u)));
        }
    };
    // ---- 1. the first two units of this wave in flight before the staging ------------------------------------------
    fetch(wqa, sqa, first, 0);
    if constexpr (NU == 2) fetch(wqb, sqb, first, 1);
    else fetch(wqb, sqb, first + QM3_WAVES, 0);
    __builtin_amdgcn_sched_barrier(0);
    QM3_STAMP();

    // ---- 2. activation slice -> LDS, once per workgroup ----------------------------------------------------------------
    if constexpr (PRO == PRO_RMSNORM) qmm3_row_ss_finish<MB>(p, tid, rss, s_inv);
    qmm3_stage_slice<MB, LM, PRO, QMM3P_SB, SF>(p, g0, gn, xs, xsum, tid, s_inv, stage);
    QM3_STAMP();
    __syncthreads();
    QM3_STAMP();

    const int swr = r ^ ((r >= 4 && r < 12) ? 4 : 0);
    int xbase = r * XS;  // element offset of this lane's row; laundered per unit so the (tile-invariant) fragment reads stay in the loop
    int xoff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) xoff[t] = ((4 * c + t) ^ swr) * 8;
    uint32_t nib_mask = 0x000f000fu;
    uint32_t magic = 0x43004300u;
    asm volatile("" : "+s"(nib_mask));
    asm volatile("" : "+v"(magic));

    // ---- 3. one unit: 16 k-steps of MFMA over 4 groups of the slice; the tile's partial sums go out after its last unit ----
    // The k-steps run as an explicit pipeline (a scheduling fence per step pins it): fragment reads two steps ahead of the MFMAs
    // that use them (ring of three), the quantisation-group scaling of group i applied one step into group i+1 (two sets of raw
    // accumulators), so neither the LDS latency nor the MFMA latency sits on the issue path.
    f32x4 acc[MB];
    auto run_unit = [&](const u32x4(&wq)[4], const uint32_t(&sq)[4], int tile, auto unit_tag) {
        constexpr int UNIT = decltype(unit_tag)::value;
        asm volatile("" : "+v"(xbase));
        const uint16_t *xrow = xs + xbase + UNIT * 512;
        const float *xsl = xsum + ((xbase - r * XS) + 4 * c) + UNIT * 4 * ROWS;  // same laundering for the group sums
        constexpr int STEPS = 16;
        f32x4 d[2][MB], xg[MB];
        u32x4 ax[3][MB];
        if constexpr (UNIT == 0) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        auto read_step = [&](int st, u32x4(&dst)[MB]) {
            const int i = st >> 2, t = st & 3;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                dst[mb] = *reinterpret_cast<const u32x4 *>(xrow + xoff[t] + (size_t)mb * 16 * XS + i * 128);
        };
        auto apply_group = [&](int i) {
            const uint32_t sw = UNIT * 4 + i < gn ? sq[i] : 0u;  // groups past the end of the row contribute nothing
            const float sc = __uint_as_float(sw << 16);
            const float be = __uint_as_float(sw & 0xffff0000u) - 128.0f * sc;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mb][j] += sc * d[i & 1][mb][j] + be * xg[mb][j];
        };
        read_step(0, ax[0]);
        read_step(1, ax[1]);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int i = st >> 2, t = st & 3;
            if (st + 2 < STEPS) read_step(st + 2, ax[(st + 2) % 3]);
            if (t == 0) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) d[i & 1][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (t == 2) {  // after the previous group's scaling (t == 1) has consumed its sums
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) xg[mb] = *reinterpret_cast<const f32x4 *>(xsl + i * ROWS + mb * 16);
            }
            const u32x4 bq = unpack_w4_bf16(wq[i][t], nib_mask, magic);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                d[i & 1][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ax[st % 3][mb]),
                                                                       __builtin_bit_cast(bf16x8_t, bq), d[i & 1][mb], 0, 0, 0);
            if (t == 1 && i > 0) apply_group(i - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        apply_group(3);
        if constexpr (UNIT == NU - 1) {
            // lane (weight row r, c) holds activation rows 16 mb + 4c + j; rows >= M fall outside the slice's buffer and are dropped
            const uint32_t base = ((uint32_t)__builtin_amdgcn_readfirstlane(tile) << 6) + st_off;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (!(QMM3_ABL & 4) || acc[mb][j] == 123.f)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[mb][j]), prs, base + (uint32_t)(mb * 16 + j) * (uint32_t)K * 4u, 0, ACT_STORE_AUX);
        }
    };
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, NU - 1>;
    if constexpr (NU == 2) {
        for (int u = 0; u < n_tiles; ++u) {
            const int tile = first + u * QM3_WAVES;
            run_unit(wqa, sqa, tile, U0{});
            if (u + 1 < n_tiles) fetch(wqa, sqa, tile + QM3_WAVES, 0);
            __builtin_amdgcn_sched_barrier(0);
            QM3_STAMP();
            run_unit(wqb, sqb, tile, U1{});
            if (u + 1 < n_tiles) fetch(wqb, sqb, tile + QM3_WAVES, 1);
            __builtin_amdgcn_sched_barrier(0);
            QM3_STAMP();
        }
    } else {
        for (int u = 0; u < n_tiles; u += 2) {
            const int tile = first + u * QM3_WAVES;
            run_unit(wqa, sqa, tile, U0{});
            if (u + 2 < n_tiles) fetch(wqa, sqa, tile + 2 * QM3_WAVES, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (u + 1 < n_tiles) {
                run_unit(wqb, sqb, tile + QM3_WAVES, U0{});
                if (u + 3 < n_tiles) fetch(wqb, sqb, tile + 3 * QM3_WAVES, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#if QMM3_ABL & 64
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    QM3_STAMP();
    if (lane == 0 && p.prof) {
        unsigned long long *o = p.prof + ((size_t)blockIdx.x * QM3_WAVES + wave) * 16;
        for (int k = 0; k < 16; ++k) o[k] = k < n_stamps ? stamps[k] : 0ull;
    }
#endif
}

struct Qmm3pGrid {
    int full_slices, wgs_full, tpw_full;  // slices of 4*NU groups: workgroups and tiles per workgroup of each
    int last_groups, wgs_last, tpw_last;  // the short last slice (0 groups = none)
};
template <int MB, int NU, int PRO = PRO_NONE, bool SF = false>
__global__ __launch_bounds__(QM3_WAVES * 64) void qmm3p_kernel(const Qmm3Args p, const Qmm3pGrid gr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#if QMM3_ABL & 64
    const prof_t prof_t0 = 0;
#else
    const prof_t prof_t0 = prof_begin(p.prof);
#endif
    const int bid = blockIdx.x;
    const int nfull = gr.full_slices * gr.wgs_full;
    if (bid < nfull) {
        const int slice = bid / gr.wgs_full;
        qmm3p_body<MB, NU, PRO, SF>(p, smem, slice, slice * 4 * NU, 4 * NU, bid - slice * gr.wgs_full, gr.tpw_full);
    } else if (NU == 2 && gr.last_groups <= 4) {
        qmm3p_body<MB, 1, PRO, SF>(p, smem, gr.full_slices, gr.full_slices * 4 * NU, gr.last_groups, bid - nfull, gr.tpw_last);
    } else {
        qmm3p_body<MB, NU, PRO, SF>(p, smem, gr.full_slices, gr.full_slices * 4 * NU, gr.last_groups, bid - nfull, gr.tpw_last);
    }
#if !(QMM3_ABL & 64)
    prof_end(p.prof, prof_t0);
#endif
}

struct Qmm3Plan {
    int MB, TW, LM, slices, tile_groups;
    int grid_x;        // one-shot: tile groups (one workgroup per (tile group, slice)); persistent: workgroups per slice
    int persistent;    // 0: qmm3_kernel, 1: qmm3p_kernel (one workgroup per CU walking tiles_per_wg tiles)
    int tiles_per_wg;  // persistent grids only
    int NU;            // qmm3p_kernel only: units (4 groups) per slice
    Qmm3pGrid pgrid;   // qmm3p_kernel only
    size_t lds, partial_bytes;
    bool ok;
};
int qmm3_num_cus();       // qmm3.hip: CUs of the current device (256 when no device is visible)
int qmm3_default_mode();  // qmm3.hip: -1 = by shape (below); tl_decode_linear's kernel 3 / 4 pin a grid through `mode`
int qmm3_forced_lm();     // qmm3.hip: 0 = the planner's slice width

// mode 0: the one-shot grid, LM the largest of {10, 8, 5, 4} whose slice fits the LDS and that still yields about one
// workgroup per CU.  mode 1: the persistent grid.  mode -1: by shape, from the r02 lab (profiles/r02_labs/qmm3_lab_r02*.log,
// matmul + slice reduction): the persistent grid wins where a CU has several tiles to walk -- gate|up (1,216 tiles) at any
// row count, lm_head (9,496) from 17 rows, w_down (76 groups: 10 slices instead of 16-19) from 17 rows -- and loses on the
// small projections (qkv, wo: 3-4 tiles per workgroup, the staged slice is most of the kernel) and on lm_head at <= 16 rows
// (two 10-group slices fit the one-shot grid there).
inline bool qmm3_prefers_persistent(int MB, int G, int tiles) {
    if (MB >= 2) return tiles >= 1024 || G >= 64;
    return tiles >= 1024 && tiles < 4096;
}
inline Qmm3Plan qmm3_plan(int M, int N, int K, int mode = -1) {
    Qmm3Plan pl{};
    pl.ok = M >= 1 && M <= 64 && N > 0 && N % 128 == 0 && K > 0 && K % 16 == 0;
    if (!pl.ok) return pl;
    pl.MB = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
    const int G = N / 128, tiles = K / 16;
    if (mode < 0) mode = qmm3_default_mode();
    if (mode < 0) mode = qmm3_prefers_persistent(pl.MB, G, tiles) ? 1 : 0;
    if (mode == 1) {
        // 33-48 rows: THREE row blocks on the persistent grid (round 6: a 48-row slice is 96 KiB of LDS instead of 128, 12 MFMAs per k-step pair instead of
        // 16 -- w_down at 33-48 rows cost what 64 rows cost)
        if (M > 32 && M <= 48) pl.MB = 3;
        // slices of 8 groups (two units) where the LDS holds them, workgroups shared out in proportion to the groups of a slice
        const int ncu = qmm3_num_cus();
        int nu = qmm3_lds_bytes(pl.MB, 8) <= 150 * 1024 && G > 4 ? 2 : 1;
        if (qmm3_forced_lm() == 4) nu = 1;
        pl.NU = nu;
        pl.LM = 4 * nu;
        Qmm3pGrid &gr = pl.pgrid;
        gr.full_slices = G / pl.LM;
        gr.last_groups = G - gr.full_slices * pl.LM;
        pl.slices = gr.full_slices + (gr.last_groups ? 1 : 0);
        auto share = [&](int groups, int &wgs, int &tpw) {
            wgs = std::min(std::max(1, (int)((long)ncu * groups / G)), tiles);
            tpw = (tiles + wgs - 1) / wgs;
            wgs = (tiles + tpw - 1) / tpw;
        };
        gr.wgs_full = gr.tpw_full = gr.wgs_last = gr.tpw_last = 0;
        if (gr.full_slices) share(pl.LM, gr.wgs_full, gr.tpw_full);
        if (gr.last_groups) share(gr.last_groups, gr.wgs_last, gr.tpw_last);
        pl.grid_x = gr.full_slices * gr.wgs_full + gr.wgs_last;
        pl.tiles_per_wg = gr.full_slices ? gr.tpw_full : gr.tpw_last;
        pl.persistent = 1;
        pl.TW = 1;
        pl.tile_groups = (tiles + QM3_WAVES - 1) / QM3_WAVES;
    } else {
        pl.TW = (pl.MB == 4 && tiles >= 2048) ? 2 : 1;
        pl.tile_groups = (tiles + QM3_WAVES * pl.TW - 1) / (QM3_WAVES * pl.TW);
        const int cand[4] = {10, 8, 5, 4};
        pl.LM = 4;
        for (int lm : cand) {
            if (qmm3_lds_bytes(pl.MB, lm) > 100 * 1024) continue;
            if (pl.MB == 4 && pl.TW == 2 && lm == 5) continue;  // 64 rows x 2 tiles per wave x 5 groups spilled 14 VGPRs: removed in round 3
            const int slices = (G + lm - 1) / lm;
            pl.LM = lm;
            if ((long)slices * pl.tile_groups >= 192 || lm == 4) break;
        }
        pl.slices = (G + pl.LM - 1) / pl.LM;
        pl.grid_x = pl.tile_groups;
    }
    pl.lds = qmm3_lds_bytes(pl.MB, pl.LM);
    pl.partial_bytes = (size_t)pl.slices * M * K * 4;
    return pl;
}

// qmm3.hip
int launch_qmm3_bf16(const Qmm3Args &args, hipStream_t st, int pro = PRO_NONE, int mode = -1);
// out = epilogue(sum over slices); epi = EPI_STORE / EPI_RESIDUAL (residual [M,K]) / EPI_SWIGLU (out [M,K/2]).
// ss_out (optional, EPI_STORE / EPI_RESIDUAL with K <= QM3_SS * 1024): [M][QM3_SS] partial sums of squares of the bf16 output rows
// for the next projection's fused RMSNorm; returns the number of workgroups launched through *n_wg.
int launch_qmm3_reduce_bf16(const float *partial, int slices, int M, int K, int epi, const uint16_t *residual, uint16_t *out,
                            prof_t *prof, hipStream_t st, float *ss_out = nullptr, int *n_wg = nullptr, const uint16_t *norm_out = nullptr,
                            uint16_t *out_w = nullptr, int out_w_frag = 0);  // norm_out + out_w (EPI_RESIDUAL): also bf16(out * norm_out), the next
                                                                             // consumer's weighted rows (out_w_frag: in qmm6.h's fragment order)
inline bool qmm3_reduce_can_emit_ss(int epi, int K) { return epi != EPI_SWIGLU && K % 4 == 0 && (K / 4 + 255) / 256 <= QM3_SS; }
// the fused RMSNorm of the skinny matmul takes any ss_n the staging prologue can fetch in four 16-byte loads per lane
inline bool qmm3_takes_ss(int ss_n) { return ss_n > 0 && ss_n <= QM3_SS_MAX && ss_n % 4 == 0; }
}  // namespace tl
