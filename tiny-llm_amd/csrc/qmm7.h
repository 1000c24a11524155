// W4A16 (group 128) batched-decode matmul that STREAMS the activation rows (5 .. 64 rows), over the engine's tiled weights.
//
//   out[m,k] = sum_n a[m,n] * (q[k,n]*s[k,g] + beta[k,g])     (reference: quantized_matvec_x4_fast, quantized_matmul.metal:441-538;
//   algebraic form sum_g (s_g sum a q + beta_g sum a), :510-521 -- the decode GEMV's semantics, as qmv3.h / qmm3.h / qmm6.h)
//
// Why a fourth kernel (round 6).  The register-resident matmul (qmm6.h) first brings EVERY activation row into the registers of its
// workgroup (320 KB at 64 rows x 2,560 columns: 6-8 us through the L2 -> CU path, which sustains ~74 GB/s per CU whatever the order --
// profiles/r05_labs/README.md section 4) and only THEN walks its weight tiles, one tile's weights in flight ahead (20 KB per CU:
// a quarter of what the HBM latency needs).  gate|up at 64 rows: 8 us of rows + 5 tiles x 1.9 us against a 4.2-us weight stream; and a
// step of 17 rows pays for 32, one of 33 for 64 (MB is 1, 2 or 4).  The two phases want to overlap, and they can once the loop order is
// turned around:
//
//  * a workgroup still owns T consecutive 16-row weight tiles and its 4 waves still split the reduction dimension (wave w owns the
//    quantisation groups [w GPW, (w+1) GPW)) -- but the wave walks its GROUPS in the outer loop and the workgroup's T tiles in the
//    inner one, with the T x MB accumulator tiles of the whole workgroup range live in registers (80 at T = 5, 64 rows).  A group's
//    step needs that group's columns of the rows (MB x 4 fragments, 16 KB per wave at 64 rows) and that group's weights of the T
//    tiles (T KB): both are requested NB = 3 groups ahead into rotating register sets, so the MFMAs of group g run while the rows
//    and weights of g + 1 and g + 2 are on their way.  The rows are never resident: 64 fragment registers per set instead of 320.
//  * everything a wave will ever fetch is requested in the order it is used, from the first cycle on, ~40 KB per wave in flight
//    (vector loads return in issue order: rows of a group first -- the group sums only need them --, then scales, then weights);
//    every request stands outside any branch (clamped addresses), so hipcc's own wait counts are exact.
//  * row blocks are a template parameter 1 .. 4 INCLUDING 3: a step costs by its 16-row blocks, not by their power of two.
//  * rows arrive WEIGHTED and in FRAGMENT ORDER (qmm6.h: the producer's epilogue writes x * w that way), one contiguous 1-KiB load
//    per fragment; RMSNorm = 1 / rms on the finished sums; the per-(row, group) sums of the beta term are A x ones on the matrix
//    pipe, straight into the accumulator layout (no LDS round trip).
//  * at the end the four waves' T x MB sums meet in LDS (T x MB KiB per wave, ONE barrier per launch instead of one per tile) and the
//    epilogue (store / SwiGLU over interleaved gate-up rows) runs in the launch, element for element qmm6.h's: the arithmetic --
//    MFMA chains per group, fmaf(beta', xs, fmaf(s, raw, acc)) in group order, the waves' sums added in wave order -- is qmm6.h's,
//    so the two kernels agree bit for bit wherever both exist (tests/test_zz_batched_matmul_gpu.py holds them together).
//
// Shapes: the reduction dimension must give every wave at most GPW groups for an instantiated GPW, and the tile count at most T
// tiles per CU for an instantiated T (qmm7_plan); anything else -- lm_head's 37 tiles per CU, wo's 4,096 columns, w_down -- stays
// where it was (qmm6.h / qmm3.h).
#pragma once
#include "qmm6.h"

namespace tl {

constexpr int QM7_WAVES = 4;
#ifndef QM7_PIN_ABOVE
#define QM7_PIN_ABOVE 96
#endif

#ifdef QMM7_TRACE  // tools/lab/qmm6_lab only: per-wave wall-clock stamps at the phase boundaries into args.prof [workgroup][wave][16]
#define QM7_STAMP() do { if (n_stamps < 14) stamps[n_stamps++] = wall_clock64(); } while (0)
#else
#define QM7_STAMP() do { } while (0)
#endif

__host__ __device__ inline size_t qmm7_lds_bytes(int MB, int T) { return (size_t)QM7_WAVES * T * MB * 1024 + (size_t)MB * 16 * 4; }

#ifndef QM7_TIE2
#define QM7_TIE2 1
#endif
#ifndef QMM7_ABL
#define QMM7_ABL 0  // tools/lab/qmm6_lab only: 1 no MFMA in the walk, 2 no nibble unpack, 4 no per-tile scaling
#endif

// NBv: register sets of the ring (groups in flight + the one computed; 0 = 3).  OCC: workgroups per CU the register budget is cut for
// (2: two row-block halves of a step side by side on one CU, 256 registers per wave).  blockIdx.y = which run of MB row blocks.
template <int MB, int T, int GPW, int EPI, int NBv = 0, int OCC = 1>
__global__ __launch_bounds__(QM7_WAVES * 64, OCC) void qmm7_kernel(const Qmm6Args p) {
    static_assert(EPI == EPI_STORE || EPI == EPI_SWIGLU, "consumers of weighted rows: store, SwiGLU");
    static_assert(MB >= 1 && MB <= 4 && T >= 1 && GPW >= 1, "row blocks 1 .. 4");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int NB = NBv > 0 ? (NBv < GPW ? NBv : GPW) : (GPW < 3 ? GPW : 3);  // register sets (groups in flight + the one computed)
    static_assert(NBv > 0 || (NB - 1) * (MB * 4 + 2 * T) <= 63, "vmcnt holds 6 bits");  // what is ever counted behind the set a group waits for
#ifdef QMM7_TRACE
    unsigned long long stamps[16];
    int n_stamps = 0;
    const prof_t prof_t0 = 0;
    QM7_STAMP();
#else
    const prof_t prof_t0 = prof_begin(p.prof);
#endif
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r = lane & 15, c = lane >> 4;  // A, B: row r, k-block c | D: weight row (column) r, activation rows 4c .. 4c+3
    const int N = p.N, K = p.K, G = N >> 7;
    const int tiles = K >> 4;
    const int g0 = wave * GPW;
    const int first = blockIdx.x * T;
    const int row0 = blockIdx.y * (MB * 16);
    f32x4 *red = reinterpret_cast<f32x4 *>(smem);                                              // [wave][T][MB][64 lanes]
    float *s_inv = reinterpret_cast<float *>(smem + (size_t)QM7_WAVES * T * MB * 1024);        // [16 MB]

    // ---- 1. the rows' partial sums of squares (16 lanes per row, 16 rows per pass): requested FIRST (loads return in issue order) and turned
    // into 1 / rms right behind the ring's first requests -- while those are on their way, and before the walk needs the registers.  Every
    // load unconditional (a clamped index re-reads the row's first partials): no control flow ahead of the ring's requests.
    f32x4 ssv[MB][QM6_SS_MAX / 64];
#pragma unroll
    for (int ps = 0; ps < MB; ++ps) {
        const int row = row0 + ps * 16 + (tid >> 4);
        const float *src = p.ss + (size_t)(row < p.M ? row : 0) * p.ss_n;
#pragma unroll
        for (int k = 0; k < QM6_SS_MAX / 64; ++k) {
            const int idx = 4 * (tid & 15) + 64 * k;
            ssv[ps][k] = *reinterpret_cast<const f32x4 *>(src + (idx < p.ss_n ? idx : 0));  // masked where it is used
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- 2. the ring: set s holds one group's columns of all rows (MB x 4 fragments), its T scale words and its T weight blocks
    u32x4 av[NB][MB][4];
    u32x4 wq[NB][T];
    uint32_t sq[NB][T];
    const char *abase = reinterpret_cast<const char *>(p.a) + (size_t)lane * 16;
    const int last_block = ((p.M + 15) >> 4) - 1;  // the caller provides ceil16(M) rows; MB row blocks cover them exactly (qmm7_plan)
    const uint32_t lane_w = (uint32_t)lane * 16u, lane_s = (uint32_t)r * 4u;
    const int last_tile = tiles - 1;
    // part `part` (0 .. T - 1) of the requests of one set: the row blocks dealt to that part, then tile `part`'s scale word and weight block.
    // The walk issues the parts of the set it fills between its tiles (the address path takes 16 cycles per KiB and is shared by the four
    // waves: a whole set in one burst stalls the wave that issues it and leaves the path idle while the wave computes).
    auto issue_part = [&](auto sc, auto glc, int part) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value, gl = decltype(glc)::value;
        const uint32_t gi = (uint32_t)min(g0 + gl, G - 1);  // groups past the end of the row: the last one again, scaled by zero below
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            if (mb * T / MB != part) continue;  // folds: `part` is an unrolled loop index at every call site
            const char *gb = abase + ((size_t)min((row0 >> 4) + mb, last_block) * G + gi) * 4096;
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) av[s][mb][t4] = *reinterpret_cast<const u32x4 *>(gb + t4 * 1024);
        }
        const int tc = __builtin_amdgcn_readfirstlane(min(first + part, last_tile));  // past the last tile: that tile again (never stored)
        sq[s][part] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(p.sbt) + ((size_t)tc * G + gi) * 64 + lane_s);
        wq[s][part] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(p.wt) + ((size_t)tc * G + gi) * 1024 + lane_w));
    };
    auto issue = [&](auto sc, auto glc) __attribute__((always_inline)) {
#pragma unroll
        for (int part = 0; part < T; ++part) issue_part(sc, glc, part);
    };
    qmm6_static_for<0, (NB > 1 ? NB - 1 : 1)>([&](auto sc) __attribute__((always_inline)) { issue(sc, sc); });  // the last set is filled during group 0
    __builtin_amdgcn_sched_barrier(0);
    // 1 / rms of the rows (fixed-order sums), published by the one barrier of the launch (ahead of the epilogue)
#pragma unroll
    for (int ps = 0; ps < MB; ++ps) {
        const bool ok = row0 + ps * 16 + (tid >> 4) < p.M;
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < QM6_SS_MAX / 64; ++k) {
            const float part = (ssv[ps][k][0] + ssv[ps][k][1]) + (ssv[ps][k][2] + ssv[ps][k][3]);
            tot += (ok && 4 * (tid & 15) + 64 * k < p.ss_n) ? part : 0.f;
        }
        tot = group16_sum(tot);
        if ((tid & 15) == 0) s_inv[ps * 16 + (tid >> 4)] = rsqrtf(tot / (float)N + p.eps);
    }

    __builtin_amdgcn_sched_barrier(0);

    QM7_STAMP();  // 1: requests out, 1 / rms in LDS
    uint32_t nib_mask = 0x000f000fu;
    uint32_t magic = 0x43004300u;
    asm volatile("" : "+s"(nib_mask));  // opaque constants (qmv3.h unpack_w4_bf16): one v_and_or_b32 per unpacked pair
    asm volatile("" : "+v"(magic));
    const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};

    // ---- 3. the walk: groups outside, the workgroup's tiles inside; acc[t][mb] lane (r, c) = rows 16 mb + 4c .. + 3, column 16 (first + t) + r
    f32x4 acc[T][MB];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[t][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int CH = MB == 3 ? 1 : 4 / MB;  // independent MFMA chains per (row block, group): qmm6.h's, so that the sums agree bit for bit
    qmm6_static_for<0, GPW>([&](auto glc) __attribute__((always_inline)) {
        constexpr int gl = decltype(glc)::value;
        constexpr int s = gl % NB;
        if constexpr (MB * NB * 16 >= QM7_PIN_ABOVE) {  // the fragments are MFMA A operands and nothing else: the accumulation half of the register file
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) asm volatile("" : "+a"(av[s][mb][0]), "+a"(av[s][mb][1]), "+a"(av[s][mb][2]), "+a"(av[s][mb][3]));
        }
        const bool g_live = g0 + gl < G;  // uniform
        // per-(row, group) sums of the activations (the beta term): A x ones; D lane (r, c) = rows 4c .. 4c+3, the same value in every column r
        f32x4 xs[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            f32x4 sm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
                sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[s][mb][t4]), __builtin_bit_cast(bf16x8_t, ones), sm, 0, 0, 0);
            xs[mb] = sm;
        }
        // Per tile:  acc += s * (raw MFMA sums) + (beta - 128 s) * (sum_k a).  The scaling of tile t - 1 (VALU) runs under the MFMAs of tile t:
        // it is issued between that tile's first and second k-step, ordered by DATA -- the tile's second weight word and the sums it
        // completes pass through one empty statement (qmm6.h: a scheduling fence does not hold pure arithmetic; left alone hipcc runs a
        // whole group's MFMAs first and keeps every raw sum alive)
        f32x4 d[2][MB][CH];
        auto scale_tile = [&](int t, int buf) __attribute__((always_inline)) {
            const uint32_t sw = g_live ? sq[s][t] : 0u;
            const float scl = __uint_as_float(sw << 16);
            const float be = __uint_as_float(sw & 0xffff0000u) - 128.0f * scl;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                f32x4 raw = d[buf][mb][0];
#pragma unroll
                for (int ch = 1; ch < CH; ++ch) raw += d[buf][mb][ch];
                if constexpr (QMM7_ABL & 4) acc[t][mb] += raw;
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][mb][j] = fmaf(be, xs[mb][j], fmaf(scl, raw[j], acc[t][mb][j]));
                }
            }
        };
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int buf = t & 1;
            // the set the previous group released takes the group NB - 1 ahead, a part per tile
            if constexpr (NB > 1 && gl + NB - 1 < GPW) {
                __builtin_amdgcn_sched_barrier(0);
                issue_part(std::integral_constant<int, (gl + NB - 1) % NB>{}, std::integral_constant<int, gl + NB - 1>{}, t);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) d[buf][mb][ch] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                uint32_t wt = wq[s][t][t4];
                if (t4 == 1 && t > 0) {
                    const int tp = t > 0 ? t - 1 : 0;
                    // behind the tile's FIRST k-step: the sums the previous tile left and a fresh accumulator pass through one statement
                    // (left to itself hipcc scales ahead of the first MFMA -- the matrix pipe idles while the VALU waits for its results)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int ch = 0; ch < CH; ++ch) asm volatile("" : "+v"(d[buf][0][0]), "+v"(d[buf ^ 1][mb][ch]));
                    scale_tile(tp, buf ^ 1);
#if QM7_TIE2
                    if constexpr (MB == 4) asm volatile("" : "+v"(wt), "+v"(acc[tp][0]), "+v"(acc[tp][1]), "+v"(acc[tp][2]), "+v"(acc[tp][3]));
                    else if constexpr (MB == 3) asm volatile("" : "+v"(wt), "+v"(acc[tp][0]), "+v"(acc[tp][1]), "+v"(acc[tp][2]));
                    else if constexpr (MB == 2) asm volatile("" : "+v"(wt), "+v"(acc[tp][0]), "+v"(acc[tp][1]));
                    else asm volatile("" : "+v"(wt), "+v"(acc[tp][0]));
#endif
                }
                const u32x4 bq = (QMM7_ABL & 2) ? u32x4{wt, wt ^ magic, wt, wt ^ magic} : unpack_w4_bf16(wt, nib_mask, magic);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    f32x4 &dd = d[buf][mb][t4 % CH];
                    if constexpr (QMM7_ABL & 1) dd[t4] += __uint_as_float((av[s][mb][t4][0] ^ bq[1]) & 0x3f800000u);
                    else dd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av[s][mb][t4]), __builtin_bit_cast(bf16x8_t, bq), dd, 0, 0, 0);
                }
            }
        }
        scale_tile(T - 1, (T - 1) & 1);
        QM7_STAMP();  // group done
        __builtin_amdgcn_sched_barrier(0);
    });

    // ---- 4. the four waves' sums meet in LDS; wave w finishes elements e = w MB + i of every tile (e = 4 mb + j: row 16 mb + 4c + j, column r)
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) red[((wave * T + t) * MB + mb) * 64 + lane] = acc[t][mb];
    __syncthreads();
    QM7_STAMP();  // the four waves met
    const int out_cols = EPI == EPI_SWIGLU ? (K >> 1) : K;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((uint32_t)p.M * (uint32_t)out_cols * 2u), 0x00020000);
    constexpr uint32_t DEAD = 0x7fffffffu;  // an offset no resource here reaches
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int tile = first + t;
        const bool tile_live = tile < tiles;  // uniform
        const int ocol = (tile << 4) + r;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int e = wave * MB + i;
            const int mb = e >> 2, j = e & 3;
            const int lrow = mb * 16 + 4 * c + j;
            const int row = row0 + lrow;
            float v = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < QM7_WAVES; ++s2) v += reinterpret_cast<const float *>(red + ((s2 * T + t) * MB + mb) * 64 + lane)[j];
            v *= s_inv[lrow];
            const bool live = tile_live && row < p.M;
            if constexpr (EPI == EPI_SWIGLU) {
                const float gv = bf16_round(v);  // rows interleaved: even = gate_i, odd = up_i
                const float uv = lane_xor1(gv);  // the odd lane next door holds up_i (only even lanes store)
                const uint32_t off = (live && (r & 1) == 0) ? ((uint32_t)row * (uint32_t)(K >> 1) + (uint32_t)(ocol >> 1)) * 2u : DEAD;
                const float sig = __builtin_amdgcn_rcpf(1.0f + exp2_hw(-1.44269504f * gv));  // as qmm6.h
                __builtin_amdgcn_raw_buffer_store_b16((short)BF16::from_float((gv * sig) * uv), ors, off, 0, ACT_STORE_AUX);
            } else {
                __builtin_amdgcn_raw_buffer_store_b16((short)BF16::from_float(v), ors, live ? ((uint32_t)row * (uint32_t)K + (uint32_t)ocol) * 2u : DEAD, 0, ACT_STORE_AUX);
            }
        }
    }
#ifdef QMM7_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    QM7_STAMP();
    if (lane == 0 && p.prof) {
        unsigned long long *o = p.prof + ((size_t)prof_wg() * QM7_WAVES + wave) * 16;
        for (int k = 0; k < 14; ++k) o[k] = k < n_stamps ? stamps[k] : 0ull;
    }
    (void)prof_t0;
#else
    prof_end(p.prof, prof_t0);
#endif
}

struct Qmm7Plan {
    int MB, T, GPW, wgs, row_blocks;
    size_t lds;
    bool ok;
};
// the (T, GPW) pairs qmm7.hip instantiates, each for MB = 1 .. 4 and the two epilogues
inline bool qmm7_has_variant(int T, int GPW) { return GPW == 5 && (T == 2 || T == 5); }
inline Qmm7Plan qmm7_plan(int M, int N, int K) {
    Qmm7Plan pl{};
    if (M < 1 || M > 64 || N <= 0 || N % 128 != 0 || K <= 0 || K % 16 != 0) return pl;
    const int G = N / 128, tiles = K / 16;
    pl.MB = (M + 15) / 16;
    pl.row_blocks = 1;
    const int gpw = (G + QM7_WAVES - 1) / QM7_WAVES;
    pl.GPW = gpw <= 5 ? 5 : 0;
    if (gpw < 4) return pl;  // a wave that idles through most of a 5-group body: the register-resident kernel's smaller variants take the shape
    const int ncu = qmm3_num_cus();
    const int tpw = (tiles + ncu - 1) / ncu;
    pl.T = tpw <= 2 ? 2 : (tpw <= 5 ? 5 : 0);
    if (!qmm7_has_variant(pl.T, pl.GPW)) return pl;
    pl.wgs = (tiles + pl.T - 1) / pl.T;
    pl.lds = qmm7_lds_bytes(pl.MB, pl.T);
    pl.ok = pl.lds <= 150 * 1024;
    return pl;
}

// qmm7.hip
bool qmm7_variant_in_table(int T, int GPW);
// rows in fragment order with their sums of squares (a.a_frag, a.ss), epilogue store or SwiGLU.  -1: not applicable, nothing launched
int launch_qmm7_bf16(const Qmm6Args &args, int epi, hipStream_t st, int *n_wg = nullptr);

}  // namespace tl
