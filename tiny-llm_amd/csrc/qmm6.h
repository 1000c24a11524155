// W4A16 (group 128) batched-decode matmul with REGISTER-RESIDENT activation rows (5 .. 64 rows), over the engine's tiled weights.
//
//   out[m,k] = sum_n a[m,n] * (q[k,n]*s[k,g] + beta[k,g])     (reference: quantized_matvec_x4_fast, quantized_matmul.metal:441-538;
//   algebraic form sum_g (s_g sum a q + beta_g sum a), :510-521 -- the decode GEMV's semantics, as qmv3.h / qmm3.h)
//
// Why a third kernel (round 4).  The K-sliced skinny matmul (qmm3.h) pays, per projection, a second launch that adds the slices
// (~4.7 us + a boundary: 108 of the 327 launches of a batched step), writes and re-reads fp32 planes that at 64 rows weigh as much
// as the weights, repeats the RMSNorm arithmetic of its slice in every workgroup and, in its main loop, reads one LDS fragment per
// MFMA.  Two structural alternatives were measured and dropped (profiles/r04_labs/README.md: an in-launch slice reduction, whole
// rows in LDS).  This kernel removes the four costs instead of moving them:
//
//  * ONE workgroup of 4 waves per CU (one wave per SIMD, the whole 512-register file of a lane).  The reduction dimension is split
//    ACROSS THE WAVES of the workgroup: wave w owns the quantisation groups [w GPW, (w+1) GPW) of every weight tile the workgroup
//    walks, and keeps ITS columns of all 16 MB activation rows in registers, already in MFMA A-operand order (MB x GPW x 4
//    fragments of 8 bf16 per lane: 320 registers at 64 rows x 5 groups).  They are fetched once per workgroup: no LDS fragment
//    read in the main loop at all.
//  * the fetch is a per-wave LDS transposer with no barrier: fragment-shaped global loads (16 rows x 64 bytes per instruction) cost
//    the address path twice (cdna_hip_programming.md), so each wave moves its OWN columns by LDS-DMA in full 256-byte row segments
//    (global_load_lds_dwordx4: 1 KiB per instruction, no registers; the XOR swizzle that makes the fragment reads conflict-free is
//    applied to the per-lane SOURCE address, the LDS image stays lane-linear) through a private ring of 4-KiB units and reads every
//    unit back once with four ds_read_b128.  Counted vmcnt / lgkmcnt waits keep RING - 2 units in flight per wave.
//  * the weights stream through two register sets (one tile ahead), one coalesced 1-KiB block per (tile, group) as everywhere else;
//  * a tile's four partial sums meet in LDS (4 x MB KiB, double-buffered, one barrier per tile) and are added in wave order, then
//    the epilogue runs IN the launch: store / residual add (+ the row's sums of squares and its weighted copy for the next
//    RMSNorm, as qmv3.h) / SwiGLU over interleaved gate-up rows.  No slices, no fp32 planes, no reduction launch.
//  * RMSNorm costs the consumer nothing: rows arrive WEIGHTED (x * w, written by the producer's epilogue) with the producer's
//    partial sums of squares, and the row's 1 / rms multiplies the finished sums (qmv3.h, PRO_RMS_WEIGHTED: one bf16 rounding per
//    staged element as in the reference, of x * w instead of x * inv * w).
//  * the per-(row, group) sums of the activations (the beta term) come out of the matrix pipe too: A x ones, once per workgroup.
#pragma once
#include <type_traits>

#include "common.h"
#include "qmv.h"
#include "qmv3.h"

namespace tl {

constexpr int QM6_WAVES = 4;
// 4-KiB units (16 rows x one group) of a wave's transposer ring, and how many units back a slot is re-used: 8 slots, 5 units (20 KiB per
// wave, 80 per CU) in flight.  (Two workgroups per CU with half the ring and half the register file were measured and lost at every
// row count but lm_head at 8-16 rows: profiles/r04_labs/README.md.)
constexpr int QM6_RING = 8;
constexpr int QM6_LAG = 2;
constexpr int QM6_SS_MAX = 256;  // partial sums of squares per row the prologue can fetch (16 lanes x 4 x 16 bytes)

struct Qmm6Args {
    const uint32_t *wt;        // tiled packed weights [K/16][G][64][4]
    const uint32_t *sbt;       // tiled scale|bias<<16 [K/16][G][16]
    const uint16_t *a;         // [M, N] bf16 rows; with `ss` they are WEIGHTED rows (x * norm weight) and out is scaled by 1 / rms.
                               // a_frag: the rows are stored in FRAGMENT ORDER (qmm6_frag_offset): every load of the kernel is one
                               // contiguous 1 KiB straight into the registers -- no LDS pass (the engine's weighted rows arrive so)
    uint16_t *out;             // [M, K]  (EPI_SWIGLU: [M, K/2])
    const uint16_t *residual;  // EPI_RESIDUAL [M, K]
    const uint16_t *norm_out;  // EPI_RESIDUAL, optional [K]: the consumer's RMSNorm weight
    uint16_t *out_w;           // EPI_RESIDUAL, optional [M, K]: out * norm_out (bf16), the consumer's weighted rows (out_w_frag: in fragment order)
    const float *ss;           // optional [M][ss_n]: partial sums of squares of the UNWEIGHTED rows (multiple of 4, <= 256)
    float *ss_out;             // EPI_RESIDUAL, optional [M][K/16]: per 16-row tile, the squares of the bf16 values stored
    float eps;
    int ss_n;
    int M, N, K;
    int tiles_per_wg;
    int a_frag, out_w_frag;
    prof_t *prof;
};

// Fragment order of a [rows, cols] bf16 matrix (cols % 128 == 0, rows padded to 16): [16-row block][group of 128 columns][k-step t]
// [lane = r + 16 c][8 elements] -- lane (r, c) of k-step t holds row 16 block + r, columns 128 g + 32 c + 8 t .. + 7, i.e. exactly the
// MFMA A fragment the tiled weight layout pairs with word t of its lane.  Element offset of (row, col):
__host__ __device__ inline size_t qmm6_frag_offset(int row, int col, int cols) {
    const int G = cols >> 7, k = col & 127;
    return (((((size_t)(row >> 4) * G + (col >> 7)) * 4 + ((k & 31) >> 3)) * 64) + (size_t)(16 * (k >> 5) + (row & 15))) * 8 + (k & 7);
}

#ifndef QMM6_ABL
#define QMM6_ABL 0  // tools/lab/qmm6_lab only: 1 no MFMA in the tile loop, 2 no nibble unpack, 4 no per-group scaling, 8 no bias pre-pass
#endif
#ifdef QMM6_TRACE  // tools/lab/qmm6_lab only: per-wave wall-clock stamps at the phase boundaries into args.prof [workgroup][wave][16]
#define QM6_STAMP() do { if (n_stamps < 14) stamps[n_stamps++] = wall_clock64(); } while (0)
#else
#define QM6_STAMP() do { } while (0)
#endif

// LDS: the four waves' transposer rings -- re-used, once every wave holds its fragments, for a tile's partial sums x 2 --, the
// per-(row, group) sums, 1 / rms per row
// (rows in fragment order need no rings: only the tiles' partial sums live there)
__host__ __device__ inline size_t qmm6_lds_ring_bytes(int MB, bool frag) { return frag ? (size_t)2 * QM6_WAVES * MB * 1024 : (size_t)QM6_WAVES * QM6_RING * 4096; }
__host__ __device__ inline size_t qmm6_lds_bytes(int MB, int GPW, bool frag = false) {
    return qmm6_lds_ring_bytes(MB, frag) + (size_t)QM6_WAVES * GPW * MB * 64 + (size_t)MB * 16 * 4;
}

template <int I, int E, typename F>
__device__ __forceinline__ void qmm6_static_for(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        qmm6_static_for<I + 1, E>(f);
    }
}

template <int MB, int GPW, int EPI, int NSETS, bool FRAG = false>
__global__ __launch_bounds__(QM6_WAVES * 64, 1) void qmm6_kernel(const Qmm6Args p) {
    static_assert(2 * QM6_WAVES * MB * 1024 <= QM6_WAVES * QM6_RING * 4096, "a tile's partial sums re-use the rings");
    static_assert(NSETS >= 1 && NSETS <= 4, "one to four weight sets");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int ROWS = MB * 16;
    constexpr int U = MB * GPW;                            // transposer units of a wave: (row block, group)
    constexpr int P = U < QM6_RING ? U : QM6_RING;         // units in flight before the first is read
    constexpr int W = (NSETS - 1) * 2 * GPW;               // weight-set loads issued behind the first P units
    static_assert(FRAG || 4 * (QM6_RING - 1) + W <= 63, "vmcnt holds 6 bits");  // the most that is ever counted behind a unit
#ifdef QMM6_TRACE
    unsigned long long stamps[16];
    int n_stamps = 0;
    const prof_t prof_t0 = 0;
    const unsigned long long sclk0 = (unsigned long long)clock64();
    QM6_STAMP();
#else
    const prof_t prof_t0 = prof_begin(p.prof);
#endif
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r = lane & 15, c = lane >> 4;  // A, B: row r, k-block c | D: weight row (column) r, activation rows 4c .. 4c+3
    const int N = p.N, K = p.K, G = N >> 7;
    const int tiles = K >> 4;
    const int row0 = blockIdx.y * ROWS;
    const int g0 = wave * GPW;
    char *ring = smem + (size_t)wave * QM6_RING * 4096;                          // this wave's units
    float *xg = reinterpret_cast<float *>(smem + qmm6_lds_ring_bytes(MB, FRAG)); // [wave][GPW][MB][16]: sum_k a of (row 4c+j, group)
    f32x4 *red = reinterpret_cast<f32x4 *>(smem);                                // [2][wave][MB][64 lanes], over the rings (after the barrier below)
    float *s_inv = xg + QM6_WAVES * GPW * MB * 16;                               // [ROWS]
    const int first = blockIdx.x * p.tiles_per_wg;
    const int n_tiles = min(p.tiles_per_wg, tiles - first);
    if (n_tiles <= 0) return;  // the grid is padded to a multiple of 8 workgroups per row block (uniform: before any barrier)

    // ---- 1. the rows' partial sums of squares: 16 lanes per row, 16 rows per pass.  Requested first and turned into 1 / rms (LDS) while
    // nothing else is live: 16 MB registers here, none later.
    const bool normed = p.ss != nullptr;  // uniform
    f32x4 ssv[MB][QM6_SS_MAX / 64];
#pragma unroll
    for (int ps = 0; ps < MB; ++ps) {
        const int row = row0 + ps * 16 + (tid >> 4);
        const bool ok = normed && row < p.M;
        const float *src = normed ? p.ss + (size_t)(ok ? row : 0) * p.ss_n : reinterpret_cast<const float *>(p.sbt);
#pragma unroll
        for (int k = 0; k < QM6_SS_MAX / 64; ++k) {
            const int idx = 4 * (tid & 15) + 64 * k;
            const bool okk = ok && idx < p.ss_n;
            ssv[ps][k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k == 0 || 64 * k < p.ss_n)  // uniform: every load is 16 cycles of the CU's one address path
                ssv[ps][k] = *reinterpret_cast<const f32x4 *>(src + (okk ? idx : 0));  // masked where it is used: nothing waits for it here
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- 2. this wave's columns of every activation row -> registers, in A-operand order ------------------------------------
    // unit (mb, gl) = rows 16 mb .. +15 x group g0 + gl: four LDS-DMA instructions of four rows each.  Lane l of instruction rq
    // fills slot j = l & 15 of row 4 rq + (l >> 4) with chunk j ^ sw(row) of that row's 256 bytes (sw as qmm3.h: every
    // ds_read_b128 service group then hits 16 different bank quads).  A buffer resource over the M rows: rows past M read zeros --
    // the row part of the address travels in the VECTOR offset (the range check does not see the scalar offset), the group in the scalar.
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.a), 0, (int)((uint32_t)p.M * (uint32_t)N * 2u), 0x00020000);
    int voff[4];
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
        const int row = 4 * rq + (lane >> 4);
        const int sw = row ^ ((row >= 4 && row < 12) ? 4 : 0);
        voff[rq] = (row0 + row) * N * 2 + (((lane & 15) ^ sw) << 4);
    }
    const int block_bytes = 16 * N * 2;
    auto issue_unit = [&](int u) __attribute__((always_inline)) {  // u = mb * GPW + gl (compile-time at every call site)
        const int mb = u / GPW, gl = u - mb * GPW;
        const int soff = min(g0 + gl, G - 1) * 256;  // uniform
        char *slot = ring + (size_t)(u % QM6_RING) * 4096;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void *)(slot + rq * 1024), 16, voff[rq] + mb * block_bytes, soff, 0, 0);
    };
    // The fragments are MFMA A operands and nothing else: the first 64 of them (256 registers) are pinned in the accumulation half of
    // the register file (ds_read_b128 / global_load write it, v_mfma reads it directly); left to itself hipcc parks them there as SPILLS
    // and copies every fragment back with four v_accvgpr_read before each MFMA (644 copies per two tiles at 64 rows).
    u32x4 av[MB][GPW][4];
    if constexpr (FRAG) {
        // rows stored in fragment order by their producer: MB x GPW x 4 contiguous 1-KiB loads per wave, straight into the registers
        // (the caller provides ceil16(M) rows: a row block past the last real one -- M = 33..48 on MB = 4, or the second workgroup row of
        // MB = 2 -- re-reads the last real block instead of running past the buffer; its rows lie at or beyond ceil16(M) and are never stored)
        const char *abase = reinterpret_cast<const char *>(p.a) + (size_t)lane * 16;
        const int last_block = ((p.M + 15) >> 4) - 1;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int gl = 0; gl < GPW; ++gl) {
                const char *gb = abase + ((size_t)min((row0 >> 4) + mb, last_block) * G + min(g0 + gl, G - 1)) * 4096;
#pragma unroll
                for (int t = 0; t < 4; ++t) av[mb][gl][t] = *reinterpret_cast<const u32x4 *>(gb + t * 1024);
            }
    } else {
        qmm6_static_for<0, P>([&](auto uc) __attribute__((always_inline)) { issue_unit(decltype(uc)::value); });
    }
    __builtin_amdgcn_sched_barrier(0);
    // the weights of the first NSETS - 1 tiles go out behind the first units (vector loads return in issue order; the rows come first)
    u32x4 wq[NSETS][GPW];
    uint32_t sq[NSETS][GPW];
    const uint32_t lane_w = (uint32_t)lane * 16u, lane_s = (uint32_t)r * 4u;
    const int last_tile = first + n_tiles - 1;
    auto fetch = [&](u32x4(&wqs)[GPW], uint32_t(&sqs)[GPW], int tile) __attribute__((always_inline)) {  // past the workgroup's last tile: that tile again (never used)
        const int tc = __builtin_amdgcn_readfirstlane(min(tile, last_tile));
        const char *wbase = reinterpret_cast<const char *>(p.wt) + (size_t)tc * G * 1024;
        const char *sbase = reinterpret_cast<const char *>(p.sbt) + (size_t)tc * G * 64;
#pragma unroll
        for (int gl = 0; gl < GPW; ++gl) {
            const uint32_t gi = (uint32_t)min(g0 + gl, G - 1);
            sqs[gl] = *reinterpret_cast<const uint32_t *>(sbase + (lane_s + gi * 64u));
        }
#pragma unroll
        for (int gl = 0; gl < GPW; ++gl) {
            const uint32_t gi = (uint32_t)min(g0 + gl, G - 1);
            wqs[gl] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wbase + (lane_w + gi * 1024u)));
        }
    };
#pragma unroll
    for (int st = 0; st < NSETS - 1; ++st) fetch(wq[st], sq[st], first + st);
    __builtin_amdgcn_sched_barrier(0);
    QM6_STAMP();  // 1: first units and weight sets requested

#define QM6_PIN_UNIT(uu)                                                              \
    if constexpr ((uu) * 4 + 3 < 64) {                                                \
        u32x4(&fr)[4] = av[(uu) / GPW][(uu) % GPW];                                   \
        asm volatile("" : "+a"(fr[0]), "+a"(fr[1]), "+a"(fr[2]), "+a"(fr[3]));        \
    }
    const int rsw = r ^ ((r >= 4 && r < 12) ? 4 : 0);
    const int frag_off = r * 256;
    // Per-(row, group) sums of the activations (the beta term): A x ones on the matrix pipe; D lane (r, c) = rows 4c .. 4c+3, the same
    // value in every column r -- all 16 lanes of a column group store it to the one address (no branch in the loop).
    const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    auto group_sum = [&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        constexpr int mb = u / GPW, gl = u % GPW;
        f32x4 sm = {0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+v"(sm));
#pragma unroll
        for (int t = 0; t < 4; ++t)
            sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[mb][gl][t]), __builtin_bit_cast(bf16x8_t, ones), sm, 0, 0, 0);
        *reinterpret_cast<f32x4 *>(xg + ((wave * GPW + gl) * MB + mb) * 16 + 4 * c) = sm;
    };
    // Unit u is read once its four pieces have landed; the slot of unit u - 2 is re-used (and its fragments pinned) once at most the
    // eight reads of units u - 1 and u pend -- LDS returns in order, so nothing here waits for a read it has just issued.
    constexpr int LAG = QM6_LAG;
    if constexpr (FRAG) {
        qmm6_static_for<0, U>([&](auto uc) __attribute__((always_inline)) {
            QM6_PIN_UNIT(decltype(uc)::value)
            group_sum(uc);
        });
    } else {
    qmm6_static_for<0, U>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        constexpr int mb = u / GPW, gl = u % GPW;
        // instructions issued behind unit u's at this point: the rest of the first P units, the weight sets (behind the first P), then
        // one unit per iteration from iteration LAG on
        constexpr int reach = u + QM6_RING - 1 - LAG > P - 1 ? u + QM6_RING - 1 - LAG : P - 1;
        constexpr int last = reach < U - 1 ? reach : U - 1;
        constexpr int behind = 4 * (last - u) + (u < P ? W : 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(behind) : "memory");
        const char *slot = ring + (size_t)(u % QM6_RING) * 4096 + frag_off;
#pragma unroll
        for (int t = 0; t < 4; ++t) av[mb][gl][t] = *reinterpret_cast<const u32x4 *>(slot + ((4 * c + t) ^ rsw) * 16);
        if constexpr (u >= LAG) {
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(4 * LAG) : "memory");
            QM6_PIN_UNIT(u - LAG)
            if constexpr (u - LAG + QM6_RING < U) issue_unit(u - LAG + QM6_RING);
            group_sum(std::integral_constant<int, u - LAG>{});  // the matrix pipe is idle while the address path moves the rows
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    // The units have landed; said with the builtin so that hipcc's own wait bookkeeping -- which cannot see the counted waits above --
    // stops treating the LDS-DMA as pending (it would otherwise drain vmcnt before EVERY tile's barrier, and with it the next tiles'
    // weights).  Few units (U <= RING): all of them went out ahead of the weight sets, which may stay in flight.
    if constexpr (U <= QM6_RING) __builtin_amdgcn_s_waitcnt(0x0070 | (W & 15) | ((W >> 4) << 14));  // vmcnt(W) lgkmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x0070);                                                          // vmcnt(0) lgkmcnt(0)
    qmm6_static_for<(U > LAG ? U - LAG : 0), U>([&](auto uc) __attribute__((always_inline)) {
        QM6_PIN_UNIT(decltype(uc)::value)
        group_sum(uc);
    });
    }
#undef QM6_PIN_UNIT
    // 1 / rms of the rows from the partial sums requested first (long landed): fixed-order sums, published by the first tile's barrier
#pragma unroll
    for (int ps = 0; ps < MB; ++ps) {
        const bool ok = normed && row0 + ps * 16 + (tid >> 4) < p.M;
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < QM6_SS_MAX / 64; ++k) {
            const float part = (ssv[ps][k][0] + ssv[ps][k][1]) + (ssv[ps][k][2] + ssv[ps][k][3]);
            tot += (ok && 4 * (tid & 15) + 64 * k < p.ss_n) ? part : 0.f;
        }
        tot = group16_sum(tot);
        if ((tid & 15) == 0) s_inv[ps * 16 + (tid >> 4)] = normed ? rsqrtf(tot / (float)N + p.eps) : 1.0f;
    }

    __syncthreads();  // every wave holds its fragments: the rings become the tiles' partial-sum buffers (and 1 / rms is published)
    QM6_STAMP();  // 2: fragments in registers, group sums in LDS
    uint32_t nib_mask = 0x000f000fu;
    uint32_t magic = 0x43004300u;
    asm volatile("" : "+s"(nib_mask));  // opaque constants (qmv3.h unpack_w4_bf16): one v_and_or_b32 per unpacked pair
    asm volatile("" : "+v"(magic));
    QM6_STAMP();  // 3: group sums done

    // ---- 4. one tile: GPW groups x 4 k-steps x MB row blocks of MFMA, the four waves' sums through LDS, the epilogue ------------
    // The tile loop is BRANCH-FREE: hipcc's wait bookkeeping loses the issue order of pending loads at every control-flow join (a
    // skipped fetch, an exec-masked store) and then drains vmcnt at the next use of a weight register -- including the set requested
    // a moment ago for a later tile.  So: fetches are unconditional (clamped to the workgroup's last tile), tiles past the end run and
    // store nothing, and every store is a buffer store whose offset lies outside the resource when the element is not live.
    const int out_cols = EPI == EPI_SWIGLU ? (K >> 1) : K;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((uint32_t)p.M * (uint32_t)out_cols * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(  // (fragment order pads the rows to 16)
        p.out_w ? p.out_w : p.out, 0, p.out_w ? (int)((uint32_t)(p.out_w_frag ? (p.M + 15) / 16 * 16 : p.M) * (uint32_t)K * 2u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t srs =
        __builtin_amdgcn_make_buffer_rsrc(p.ss_out ? p.ss_out : reinterpret_cast<float *>(p.out), 0, p.ss_out ? (int)((uint32_t)p.M * (uint32_t)tiles * 4u) : 0, 0x00020000);
    constexpr uint32_t DEAD = 0x7fffffffu;  // an offset no resource here reaches
    // what the epilogue of THIS wave's elements needs from memory (EPI_RESIDUAL); requested AHEAD of the tile's weight fetch: loads
    // return in issue order, and behind the fetch these values would make the epilogue wait for the next tile's weights
    auto residual_loads = [&](int tile, uint16_t(&resv)[MB], uint16_t &nwo) __attribute__((always_inline)) {
        if constexpr (EPI == EPI_RESIDUAL) {
            const int ocol = (tile << 4) + r;
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int e = wave * MB + i;
                const int row = row0 + (e >> 2) * 16 + 4 * c + (e & 3);
                resv[i] = p.residual[(size_t)min(row, p.M - 1) * K + ocol];
            }
            nwo = (p.out_w ? p.norm_out : p.residual)[ocol];
        }
    };
    auto run_tile = [&](const u32x4(&wqs)[GPW], const uint32_t(&sqs)[GPW], int tile, int u, const uint16_t(&resv)[MB], const uint16_t nwo) __attribute__((always_inline)) {
        const bool tile_live = u < n_tiles;  // uniform
        const int ocol = (tile << 4) + r;
        // Per group:  acc += s_g * (raw MFMA sums) + (beta_g - 128 s_g) * (sum_k a).  The scaling of group gl - 1 (VALU) runs under the
        // MFMAs of group gl: it is issued between that group's first and second k-step, ordered by DATA -- the second weight word
        // of the group and the running sums pass through one empty statement (a scheduling fence does not hold pure arithmetic:
        // hipcc ran all GPW groups' MFMAs first and kept their raw sums alive).  The group sums come from LDS two groups ahead, by
        // explicit ds_read_b128 statements (left to hipcc, each read sat directly in front of its use: an LDS round trip per group
        // on a wave that has no second wave to hide it).  With fewer than 4 row blocks a group's four k-steps run as CH independent
        // MFMA chains: one chain alone waits out every MFMA's latency.
        constexpr int CH = GPW > 8 ? 1 : 4 / MB;
        constexpr int XD = (MB == 4 || GPW > 8) ? 1 : 2;  // groups the LDS reads run ahead (a group of 16 MFMAs covers an LDS round trip; one of 4 does not; 19 groups: registers)
        f32x4 acc[MB], d[2][MB][CH], xs[XD][MB];
        const uint32_t xg_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)(reinterpret_cast<char *>(xg + (wave * GPW * MB) * 16 + 4 * c));
        // (macros, not generic lambdas: clang rejects inline-asm operands that name a variable of an enclosing lambda from inside a
        // generic one, and the read's offset must be an immediate)
#define QM6_XS_READ(gl, mb) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xs[(gl) % XD][(mb) < MB ? (mb) : 0]) : "v"(xg_lds), "n"((((gl) * MB + ((mb) < MB ? (mb) : 0)) * 64)) : "memory")
#define QM6_XS_ISSUE(gl)                                   \
    do {                                                   \
        QM6_XS_READ(gl, 0);                                \
        if constexpr (MB > 1) QM6_XS_READ(gl, 1);          \
        if constexpr (MB > 2) { QM6_XS_READ(gl, 2); QM6_XS_READ(gl, 3); } \
    } while (0)
        // every read issued so far has returned (the newest went out a whole group of MFMAs ago)
#define QM6_SCALE(gl)                                                                                                                         \
    do {                                                                                                                                      \
        if constexpr (MB == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xs[(gl) % XD][0]), "+v"(xs[(gl) % XD][MB > 1 ? 1 : 0]), "+v"(xs[(gl) % XD][MB > 2 ? 2 : 0]), "+v"(xs[(gl) % XD][MB > 3 ? 3 : 0])); \
        else if constexpr (MB == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xs[(gl) % XD][0]), "+v"(xs[(gl) % XD][MB > 1 ? 1 : 0]));         \
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xs[(gl) % XD][0]));                                                                    \
        const uint32_t sw = (g0 + (gl)) < G ? sqs[gl] : 0u; /* groups past the end of the row contribute nothing */                          \
        const float sc = __uint_as_float(sw << 16);                                                                                           \
        const float be = __uint_as_float(sw & 0xffff0000u) - 128.0f * sc;                                                                     \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) {                                                                                   \
            f32x4 raw = d[(gl) & 1][mb][0];                                                                                                   \
            _Pragma("unroll") for (int ch = 1; ch < CH; ++ch) raw += d[(gl) & 1][mb][ch];                                                     \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[mb][j] = fmaf(be, xs[(gl) % XD][mb][j], fmaf(sc, raw[j], acc[mb][j]));          \
        }                                                                                                                                     \
    } while (0)
#define QM6_GROUP(gl)                                                                                                                         \
    if constexpr ((gl) < GPW) {                                                                                                               \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                                                     \
            _Pragma("unroll") for (int ch = 0; ch < CH; ++ch) d[(gl) & 1][mb][ch] = f32x4{0.f, 0.f, 0.f, 0.f}; /* the MFMA's literal 0 */    \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                                                       \
            uint32_t wt = wqs[gl][t];                                                                                                         \
            if (t == 1) {                                                                                                                     \
                if constexpr ((gl) > 0 && !(QMM6_ABL & 4)) {                                                                                  \
                    QM6_SCALE(((gl) > 0 ? (gl) - 1 : 0));                                                                                     \
                    if constexpr ((gl) - 1 + XD < GPW) QM6_XS_ISSUE(((gl) - 1 + XD < GPW ? (gl) - 1 + XD : 0));                                \
                }                                                                                                                             \
                if constexpr (MB == 4) asm volatile("" : "+v"(wt), "+v"(acc[0]), "+v"(acc[MB > 1 ? 1 : 0]), "+v"(acc[MB > 2 ? 2 : 0]), "+v"(acc[MB > 3 ? 3 : 0])); \
                else if constexpr (MB == 2) asm volatile("" : "+v"(wt), "+v"(acc[0]), "+v"(acc[MB > 1 ? 1 : 0]));                             \
                else asm volatile("" : "+v"(wt), "+v"(acc[0]));                                                                               \
            }                                                                                                                                 \
            const u32x4 bq = (QMM6_ABL & 2) ? u32x4{wt, wt ^ magic, wt, wt ^ magic} : unpack_w4_bf16(wt, nib_mask, magic);                   \
            _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) {                                                                               \
                f32x4 &dd = d[(gl) & 1][mb][t % CH];                                                                                          \
                if constexpr (QMM6_ABL & 1) dd[t] += __uint_as_float((av[mb][gl][t][0] ^ bq[1]) & 0x3f800000u);                               \
                else dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[mb][gl][t]), __builtin_bit_cast(bf16x8_t, bq), dd, 0, 0, 0); \
            }                                                                                                                                 \
        }                                                                                                                                     \
    }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        QM6_XS_ISSUE(0);
        if constexpr (GPW > 1 && XD > 1) QM6_XS_ISSUE(1);
        QM6_GROUP(0) QM6_GROUP(1) QM6_GROUP(2) QM6_GROUP(3) QM6_GROUP(4) QM6_GROUP(5) QM6_GROUP(6) QM6_GROUP(7) QM6_GROUP(8) QM6_GROUP(9)
        QM6_GROUP(10) QM6_GROUP(11) QM6_GROUP(12) QM6_GROUP(13) QM6_GROUP(14) QM6_GROUP(15) QM6_GROUP(16) QM6_GROUP(17) QM6_GROUP(18)
        static_assert(GPW <= 19, "QM6_GROUP is expanded 19 times");
        QM6_SCALE(GPW - 1);
#undef QM6_GROUP
#undef QM6_SCALE
#undef QM6_XS_ISSUE
#undef QM6_XS_READ
        QM6_STAMP();  // tile: MFMA phase done
        f32x4 *rb = red + (u & 1) * QM6_WAVES * MB * 64;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) rb[(wave * MB + mb) * 64 + lane] = acc[mb];
        __syncthreads();
        QM6_STAMP();  // tile: the four waves met
        // wave w finishes elements e = w MB + i (e = 4 mb + j: activation row 16 mb + 4c + j, column r of the tile)
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int e = wave * MB + i;
            const int mb = e >> 2, j = e & 3;
            const int lrow = mb * 16 + 4 * c + j;
            const int row = row0 + lrow;
            float v = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < QM6_WAVES; ++s2) v += reinterpret_cast<const float *>(rb + (s2 * MB + mb) * 64 + lane)[j];
            v *= s_inv[lrow];
            const bool live = tile_live && row < p.M;
            if constexpr (EPI == EPI_SWIGLU) {
                const float gv = bf16_round(v);  // rows interleaved: even = gate_i, odd = up_i
                const float uv = lane_xor1(gv);  // the odd lane next door holds up_i (only even lanes store)
                const uint32_t off = (live && (r & 1) == 0) ? ((uint32_t)row * (uint32_t)(K >> 1) + (uint32_t)(ocol >> 1)) * 2u : DEAD;
                // silu by the hardware exponential and reciprocal (1 ulp each, far below the bf16 step of the result): the IEEE division
                // and expf of the other kernels are ~40 instructions per element on a wave that has nothing to overlap them with
                const float sig = __builtin_amdgcn_rcpf(1.0f + exp2_hw(-1.44269504f * gv));
                __builtin_amdgcn_raw_buffer_store_b16((short)BF16::from_float((gv * sig) * uv), ors, off, 0, ACT_STORE_AUX);
            } else if constexpr (EPI == EPI_RESIDUAL) {
                const uint32_t o = (uint32_t)row * (uint32_t)K + (uint32_t)ocol;
                const uint16_t ov = BF16::from_float(BF16::to_float(resv[i]) + bf16_round(v));
                __builtin_amdgcn_raw_buffer_store_b16((short)ov, ors, live ? o * 2u : DEAD, 0, ACT_STORE_AUX);
                const uint32_t ow = p.out_w_frag ? (uint32_t)qmm6_frag_offset(row, ocol, K) : o;
                __builtin_amdgcn_raw_buffer_store_b16((short)BF16::from_float(BF16::to_float(ov) * BF16::to_float(nwo)), wrs, live ? ow * 2u : DEAD, 0, ACT_STORE_AUX);
                // one partial per (activation row, 16-row tile): the squares of the stored bf16 values (an empty resource when not asked for)
                const float sq_v = group16_sum(live ? BF16::to_float(ov) * BF16::to_float(ov) : 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sq_v), srs, (live && r == 0) ? ((uint32_t)row * (uint32_t)tiles + (uint32_t)tile) * 4u : DEAD, 0, ACT_STORE_AUX);
            } else {
                __builtin_amdgcn_raw_buffer_store_b16((short)BF16::from_float(v), ors, live ? ((uint32_t)row * (uint32_t)K + (uint32_t)ocol) * 2u : DEAD, 0, ACT_STORE_AUX);
            }
        }
    };
    int u0 = 0;
    for (; u0 + NSETS <= n_tiles; u0 += NSETS) {  // whole rounds
        qmm6_static_for<0, NSETS>([&](auto sc) __attribute__((always_inline)) {
            constexpr int st = decltype(sc)::value;
            const int u = u0 + st;
            const int tile = first + u;
            uint16_t resv[MB] = {}, nwo = 0;
            residual_loads(tile, resv, nwo);
            __builtin_amdgcn_sched_barrier(0);
            // the set the previous tile has just released takes the tile NSETS - 1 ahead (with one set: this tile, not overlapped)
            fetch(wq[(st + NSETS - 1) % NSETS], sq[(st + NSETS - 1) % NSETS], first + u + NSETS - 1);
            __builtin_amdgcn_sched_barrier(0);
            run_tile(wq[st], sq[st], tile, u, resv, nwo);
            __builtin_amdgcn_sched_barrier(0);
            QM6_STAMP();  // tile: epilogue issued
        });
    }
    // the last, partial round: its tiles' sets are in flight already (nothing is requested any more, so hipcc's conservative waits
    // behind these branches cost nothing)
    if constexpr (NSETS > 1) {
        qmm6_static_for<0, NSETS - 1>([&](auto sc) __attribute__((always_inline)) {
            constexpr int st = decltype(sc)::value;
            const int u = u0 + st;
            if (u < n_tiles) {  // uniform
                uint16_t resv[MB] = {}, nwo = 0;
                residual_loads(first + u, resv, nwo);
                run_tile(wq[st], sq[st], first + u, u, resv, nwo);
                QM6_STAMP();
            }
        });
    }
#ifdef QMM6_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    QM6_STAMP();
    if (lane == 0 && p.prof) {
        unsigned long long *o = p.prof + ((size_t)prof_wg() * QM6_WAVES + wave) * 16;
        for (int k = 0; k < 14; ++k) o[k] = k < n_stamps ? stamps[k] : 0ull;
        o[14] = (unsigned long long)clock64() - sclk0;  // shader clocks over the kernel, against the wall clock of stamp 0 .. now
        o[15] = wall_clock64() - stamps[0];
    }
    (void)prof_t0;
#else
    prof_end(p.prof, prof_t0);
#endif
}

struct Qmm6Plan {
    int MB, GPW, NSETS, row_blocks, wgs, tiles_per_wg;
    size_t lds;
    bool ok;
};
int qmm3_num_cus();  // qmm3.hip
// The (MB, GPW) pairs qmm6.hip instantiates: the register file of one wave per SIMD holds MB x GPW x 16 fragment registers
inline bool qmm6_has_variant(int MB, int GPW) {
    if (MB != 1 && MB != 2 && MB != 4) return false;
    if (GPW != 2 && GPW != 4 && GPW != 5 && GPW != 8 && GPW != 19) return false;
    return MB * GPW * 16 <= 320;
}
// weight sets of a workgroup that walks several tiles: tiles in flight ahead of the one computed = sets - 1, as many as the registers
// next to the fragments hold (a tile of few rows computes in a fraction of the memory latency) and the 6-bit vmcnt counts behind the
// first ring of units (4 (RING - 1) + (sets - 1) 2 GPW <= 63)
constexpr int qmm6_sets(int MB, int GPW) { return GPW > 8 ? 1 : (MB * GPW <= 5 ? 4 : (MB * GPW <= 10 ? 3 : 2)); }
// sets for a workgroup of `tpw` tiles: whole rounds of `sets` tiles run in the branch-free loop, the rest behind it (qmm6_kernel)
inline int qmm6_pick_sets(int MB, int GPW, int tpw) { return std::min(tpw, qmm6_sets(MB, GPW)); }
inline int qmm6_round_gpw(int gpw) { return gpw <= 2 ? 2 : (gpw <= 4 ? 4 : (gpw <= 5 ? 5 : (gpw <= 8 ? 8 : 19))); }
inline Qmm6Plan qmm6_plan(int M, int N, int K, bool frag = false) {
    Qmm6Plan pl{};
    if (M < 1 || M > 64 || N <= 0 || N % 128 != 0 || K <= 0 || K % 16 != 0) return pl;
    const int G = N / 128, tiles = K / 16;
    if ((G + QM6_WAVES - 1) / QM6_WAVES > 19) return pl;
    pl.GPW = qmm6_round_gpw((G + QM6_WAVES - 1) / QM6_WAVES);
    // Row blocks per workgroup (MB = 1, 2 or 4 blocks of 16 rows; the rows beyond go to further workgroups over the same tiles).  Until round 6: the largest
    // block whose fragments fit the registers.  But what a workgroup costs is what its CU pulls -- MB x 16 rows of N columns + its tiles' weights, at ~13.5 ns
    // per KiB (the L2 -> CU path, profiles/r05_labs) -- plus its walk (a tile: the four waves' meeting and the stores, + MB x GPW x 4 MFMAs on one wave per
    // SIMD), and long rows against few tiles (wo: 4,096 columns, 160 tiles) are cheaper in MORE, SMALLER workgroups: 64 rows as 4 x 56 workgroups of 16 rows
    // x 3 tiles pull 227 KiB each where 2 x 80 of 32 rows x 2 tiles pull 326 (lab: 10.3 -> 9.1 us; 33-48 rows, three blocks instead of a padded four: 10.1
    // -> 7.2).  gate|up / lm_head (many tiles per workgroup) keep the largest block by the same arithmetic.
    const int blocks16 = (M + 15) / 16;
    const int ncu = qmm3_num_cus();
    auto shape_for = [&](int mb, Qmm6Plan &q) {
        q.MB = mb;
        q.row_blocks = (blocks16 + mb - 1) / mb;
        const int wgs = std::min(tiles, std::max(1, ncu / q.row_blocks));
        q.tiles_per_wg = (tiles + wgs - 1) / wgs;
        q.wgs = (tiles + q.tiles_per_wg - 1) / q.tiles_per_wg;
        // the workgroups of one tile range sit a multiple of 8 apart in the launch order: one XCD, one L2 for the weights they share
        if (q.row_blocks > 1) q.wgs = (q.wgs + 7) / 8 * 8;
    };
    const int mb_max = blocks16 >= 3 ? 4 : (blocks16 >= 2 ? 2 : 1);
    double best = 0.0;
    bool found = false;
    for (int mb = mb_max; mb >= 1; mb /= 2) {
        if (!qmm6_has_variant(mb, pl.GPW) || qmm6_lds_bytes(mb, pl.GPW, frag) > 150 * 1024) continue;
        Qmm6Plan q = pl;
        shape_for(mb, q);
        // (the arithmetic below decides for LONG rows against FEW tiles only -- 8 groups per wave and more, at most two tiles per workgroup on the largest
        // block: wo.  A projection whose workgroups walk more tiles keeps the largest block; so does qkv: the row-streaming kernel (qmm7.h) is its twin
        // bit for bit, and the order in which a group's four k-steps are added follows the row-block count -- CH below)
        if (found && (pl.tiles_per_wg > 2 || pl.GPW < 8)) break;
        if ((long)q.wgs * q.row_blocks > ncu + 7) continue;  // one workgroup per CU
        const double kib = (double)std::min(mb * 16, M) * N * 2 / 1024.0 + (double)q.tiles_per_wg * G;
        // (plain rows come through each wave's 8-slot transposer ring: a second and third pass of the ring -- MB x GPW units -- is a dependent round trip each)
        const double us = 0.0135 * kib + q.tiles_per_wg * (0.35 + 0.03 * mb * pl.GPW) + 1.3 * ((mb * pl.GPW - 1) / QM6_RING);
#ifdef QMM6_PLAN_LARGEST_BLOCK  // tools/lab only: the rule of rounds 4-5
        if (found) continue;
#endif
        if (!found || us < best - 1e-9) pl.MB = q.MB, pl.row_blocks = q.row_blocks, pl.tiles_per_wg = q.tiles_per_wg, pl.wgs = q.wgs, best = us, found = true;
    }
    if (!found) return pl;
    pl.NSETS = qmm6_pick_sets(pl.MB, pl.GPW, pl.tiles_per_wg);
    pl.lds = qmm6_lds_bytes(pl.MB, pl.GPW, frag);
    pl.ok = pl.lds <= 150 * 1024;
    return pl;
}

// qmm6.hip
bool qmm6_variant_in_table(int MB, int GPW);
int launch_qmm6_bf16(const Qmm6Args &args, int epi, hipStream_t st, int *n_wg = nullptr);  // -1: not applicable, nothing launched

}  // namespace tl
