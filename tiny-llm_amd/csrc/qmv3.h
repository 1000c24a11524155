// W4A16 (group 128) decode GEMV on the matrix cores, over the engine's TILED weight layout (gfx950).
//
//   out[m,k] = sum_n a[m,n] * (q[k,n]*s[k,g] + beta[k,g])          (reference: quantized_matvec_x4_fast,
//   quantized_matmul.metal:441-538; algebraic form sum_g (s_g sum a q + beta_g sum a), :510-521)
//
// What the measurements said (tools/lab, profiles/): the packed-dot GEMV (qmv.h) is VALU-bound at ~3.3 TB/s because
// int4 carries 4x the multiply-adds per byte of bf16; a first MFMA version over the checkpoint layout spent its time
// ISSUING loads (a lane-load of "16 rows x 64 B" touches 16 half-used cache lines, ~1.2 us to issue 10 of them) and
// in a per-workgroup activation-staging prologue that sat in the same in-order vmcnt queue as the weight stream.
// This kernel removes all three:
//
//  * TILED WEIGHTS (tl_repack_w4_tiled, done once when the engine adopts a checkpoint): for every 16-row tile and
//    quantisation group the 16 x 64 B of packed nibbles are stored as one contiguous 1 KiB block in MFMA B-operand
//    lane order, [tile][g][lane = r + 16 c][4 words]; a wave-load is one fully coalesced 1 KiB burst.  Inside a
//    word the nibbles are reordered (n_i = q_2i, n_{i+4} = q_{2i+1}) so that (w >> 4i) & 0x000f000f | 0x43004300
//    yields the bf16 pair (128+q_2i, 128+q_2i+1): weights unpack straight into natural k order and the activations
//    need no permutation.  Scales and biases ride along as one dword (s | beta << 16) per (tile, g, row).
//  * MATRIX CORES: D[act row][weight row] += A[act rows x 32 k] * B[32 k x 16 weight rows] with
//    v_mfma_f32_16x16x32_bf16; the VALU only unpacks nibbles (7 ops per 8 weights) and applies the per-group scale /
//    bias term, acc += s * D + (beta - 128 s) * sum_k a, with sum_k a taken once per workgroup while the activations
//    are staged.  The 16 A rows are the decode batch (M <= 8; rows >= M repeat).
//  * COOPERATIVE STAGING: every wave first puts its whole weight slice in flight (LM 1-KiB blocks in registers), then
//    all waves together load the activation rows, apply the fused RMSNorm and write bf16 rows to LDS.  The
//    activation loads are issued BEFORE the weight loads (vmcnt retires in order and they are needed first); a
//    dedicated stager wave was measured 1.6-2.6x slower (profiles/README.md).
//  * every load is unconditional from a clamped address (a divergent branch around a load makes hipcc wait for it
//    at the join), and every compute wave runs a fixed LM-group body (surplus groups carry a zero scale).
#pragma once
#include "common.h"
#include "qmv.h"

namespace tl {

constexpr int Q3_LMAX = 10;  // most groups (1 KiB wave-loads) per compute wave; kernels are specialised on LM <= Q3_LMAX
constexpr int Q3_PAD = 8;    // bf16 elements of padding per activation row in LDS

struct Qmv3Args {
    const uint32_t *wt;   // tiled packed weights [K/16][G][64][4]
    const uint32_t *sbt;  // tiled scale|bias<<16 (bf16 pair) [K/16][G][16]
    const uint16_t *a;    // [M, N]
    uint16_t *out;        // [M, K]  (EPI_SWIGLU: [M, K/2])
    const uint16_t *norm_w;
    const uint16_t *residual;
    float eps;
    int M, N, K;
    prof_t *prof;
#ifdef QMV3_TRACE  // lab only (tools/lab/trace_lab): wall-clock stamps of wave 0 at the phase boundaries, p.trace [blocks][8]
#define Q3_STAMP(i) do { if (p.trace && threadIdx.x == 0) p.trace[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define Q3_STAMP(i) do { } while (0)
#endif
#ifdef QMV3_LAB
    int ablate;  // lab only: 1 = skip MFMA math, 2 = skip the staging arithmetic / LDS stores, 4 = skip the activation loads
#endif
#ifdef QMV3_TRACE
    unsigned long long *trace;  // lab only
#endif
    // PRO_ATTN_MERGE (the wo projection of a single decode row with 2 / 4 / 8 attention windows): `a` is not read; the activation row is
    // the merge of the decode-attention kernel's split partials merge_ws [head][NS][128 + 4] (value sums, running max, running
    // sum; head dimension 128, N = heads * 128), formed while the row is staged -- attn_merge_kernel's arithmetic, term for
    // term, so the staged bf16 row has the bits that kernel would have written; its launch is dropped.
    const float *merge_ws;
    // Producer-side sums of squares (round 3).  Every workgroup of a PRO_RMSNORM GEMV used to re-derive the row's sum of squares
    // (sum per thread, wave reduction, LDS, a barrier) before it could normalise: 0.44 us of the qkv GEMV and 1.1 us of the
    // 1,216-workgroup gate|up GEMV (tools/lab/abl_lab, bit 8).  The GEMV that WRITES the row (wo -> h, w_down -> x) now leaves,
    // per activation row, one partial per 16-row tile: the sum of the squares of the bf16 values it stored (ss_out [M][K / 16]);
    // the consumer adds ss_n partials in a fixed order (ss_in [M][ss_n]; 8 zero-padded partials from the embedding kernels).
    const float *ss_in;
    int ss_n;
    float *ss_out;
    // Rows weighted by their producer (round 3).  With the sums of squares handed over, what is left of the fused RMSNorm is work
    // every workgroup repeats on the whole row: fetch the norm weights, multiply, round.  RMSNorm is x * inv * w with ONE scalar
    // inv per row, so the GEMV is linear in it: out = inv * sum_n bf16(x_n w_n) W_kn.  An EPI_RESIDUAL GEMV therefore also
    // leaves the row its consumer stages, out_w = bf16(out * norm_out) (the next RMSNorm's weight; 2 bytes more per element
    // written once), and the consumer (PRO_RMS_WEIGHTED: `a` = out_w, ss_in required) stages it like a plain row and multiplies
    // its accumulators by inv at the end -- the partial sums are not even waited for until then.  One bf16 rounding per staged
    // element as in the reference (there of x * inv * w, here of x * w): same error bound, not the same bits.
    const uint16_t *norm_out;  // [K]      EPI_RESIDUAL, optional
    uint16_t *out_w;           // [M, K]   EPI_RESIDUAL, optional
    // EPI_STORE, optional (the lm_head projection of a decode step, round 4): per (activation row, 16-row tile) the largest STORED
    // bf16 value and the lowest output index that holds it, tile_max [M][K / 16] (value, index as float: indices stay below 2^24).
    // step_end_kernel then picks the greedy id from K / 16 pairs instead of re-reading all K logits (reference: mx.argmax over the
    // logits row, benches/bench.py:234-243; first maximum wins).
    f32x2 *tile_max;
};

// output stores (lab: -DQ3_STORE_MODE=1 nontemporal, 2 = system-scope atomic store, i.e. write-through)
#ifndef Q3_STORE_MODE
#define Q3_STORE_MODE 0
#endif
template <typename T> __device__ __forceinline__ void q3_store(T *ptr, T v) {
#if Q3_STORE_MODE == 1
    __builtin_nontemporal_store(v, ptr);
#elif Q3_STORE_MODE == 2
    __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    act_store(ptr, v);  // (write-through in the AQL route's code objects: common.h TL_COHERENT)
#endif
}
#ifdef QMV3_LAB
#define Q3_ABL(bit) (p.ablate & (bit))
#else
#define Q3_ABL(bit) 0
#endif

// LDS: activation rows [MR][N + Q3_PAD] bf16, then Q3_LMAX zero groups (a wave's fixed-length body may run past the
// end of the row: it then reads finite data it multiplies by a zero scale); per-group sums [G + Q3_LMAX][16] fp32;
// split-K partials.
__host__ __device__ inline size_t qmv3_lds_xsum_off(int MR, int N) {
    return (((size_t)MR * (N + Q3_PAD) + (size_t)Q3_LMAX * 128) * 2 + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t qmv3_lds_red_off(int MR, int N) {
    return qmv3_lds_xsum_off(MR, N) + (size_t)(N / 128 + Q3_LMAX) * 16 * 4;
}
__host__ __device__ inline size_t qmv3_lds_bytes(int MR, int N, int KS, int CW) {
    size_t red = KS > 1 ? (size_t)CW * MR * 16 * 4 : 0;
    return qmv3_lds_red_off(MR, N) + red + (size_t)MR * CW * 4 + 64;  // + RMSNorm partial sums of squares
}

// 8 packed nibbles (n_i = q_2i, n_{i+4} = q_{2i+1}) -> four bf16 pairs (128+q_2i, 128+q_2i+1): 3 shifts + 4 v_and_or_b32
// ((w >> 4i) & mask | magic).  With literal constants hipcc emits v_and + v_or (11 ops: VOP3 cannot take two literals), so
// the caller keeps the mask in an SGPR and the magic in a VGPR that are opaque to constant folding.
// This is plain C on purpose.  Round 1 wrote these seven instructions as ONE inline-asm statement; the hazard recogniser
// does not look inside asm, and in the <MR 2, KS 8, CW 8> instantiation the register allocator gave the statement's
// outputs the registers a still-executing MFMA was reading as its accumulator input (XDL read SrcC -> VALU write, a WAR
// hazard hipcc pads with s_nop for its own instructions): every output of that variant was garbage (found by
// tests/test_decode_kernels_gpu.py at the real w_down shape with two activation rows).
__device__ __forceinline__ u32x4 unpack_w4_bf16(uint32_t w, uint32_t mask_s, uint32_t magic_v) {
    return u32x4{(w & mask_s) | magic_v, ((w >> 4) & mask_s) | magic_v, ((w >> 8) & mask_s) | magic_v,
                 ((w >> 12) & mask_s) | magic_v};
}

// LM = groups per compute wave (the fixed-length body): the per-wave dependent chain (unpack + MFMA per group) is what
// bounds the small projections, so LM is the smallest listed value that covers ceil(G / KS)
template <int MR, int KS, int CW, int PRO, int EPI, int LM, int NS = 0>
__global__ __launch_bounds__(CW * 64) void qmv3_kernel(const Qmv3Args p) {
    static_assert(PRO != PRO_ATTN_MERGE || (MR == 1 && (NS == 2 || NS == 4 || NS == 8)), "PRO_ATTN_MERGE: one row, 2 / 4 / 8 splits");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = CW * 64;
    constexpr int WR = CW / KS;
    constexpr int ROWS = MR < 4 ? MR : 4;  // accumulator rows a lane actually needs (lanes c > 0 only when MR > 4)
    const prof_t prof_t0 = prof_begin(p.prof);
    Q3_STAMP(0);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps the slice arithmetic on the SALU
    const int lane = tid & 63;
    const int N = p.N, K = p.K, G = N >> 7;
    const int xstride = N + Q3_PAD;
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);
    float *xsum = reinterpret_cast<float *>(smem + qmv3_lds_xsum_off(MR, N));  // [G + Q3_LMAX][16 activation rows]
    float *red = reinterpret_cast<float *>(smem + qmv3_lds_red_off(MR, N));

    const int r = lane & 15;  // B: weight row in tile | A: activation row | D: weight row (column)
    const int c = lane >> 4;  // A, B: k-block         | D: activation rows 4c .. 4c+3
    const int wt = wave / KS;
    const int ks = wave - wt * KS;
    const int tiles = K >> 4;
    const int tile = blockIdx.x * WR + wt;
    const bool tile_ok = tile < tiles;
    const int tile_c = tile_ok ? tile : 0;
    const int Lper = (G + KS - 1) / KS;
    const int g0 = ks * Lper;
    const int g1 = min(g0 + Lper, G);

    // ---- 1. every load of the workgroup goes out first, smallest (and first needed) first: vmcnt retires in order ----
    const int cpr = N >> 3;  // 16-byte chunks per activation row (a multiple of 16: one group = 16 chunks)
    // Thread t owns column chunks cc = t + k*T (k < CCU) of EVERY activation row: the norm weights are fetched once per
    // column, the per-row sums of squares need no row selects, and a 16-lane group still covers one quantisation group.
    constexpr int CCU = MR <= 2 ? 3 : 2;
    const bool reg_path = cpr <= CCU * T;  // all activation chunks fit in registers: one global round trip
    u32x4 xv[CCU][MR], nwv[CCU];
    // PRO_ATTN_MERGE: chunk cc = 8 columns of head cc / 16; per split its 8 value sums and the head's (max, sum) pair
    constexpr int MS = PRO == PRO_ATTN_MERGE ? NS : 1;
    constexpr int MCCU = PRO == PRO_ATTN_MERGE ? 2 : 1;  // chunk sets a merging thread owns: the launcher admits N / 8 <= 2 T only
    constexpr int MWS = 128 + 4;  // floats per partial row in merge_ws (engine_kernels.h, ATTN_WS_PAD): rows start on 16 bytes
    f32x4 mval[MCCU][MS][2];
    f32x2 mml[MCCU][MS];
    if constexpr (PRO == PRO_ATTN_MERGE) {
#pragma unroll
        for (int k = 0; k < MCCU; ++k) {  // no branch around these loads: every thread reads from a clamped address
            const int cc = min(tid + k * T, cpr - 1);
            const float *hb = p.merge_ws + (size_t)(cc >> 4) * NS * MWS;
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) mml[k][s2] = act_load(reinterpret_cast<const f32x2 *>(hb + s2 * MWS + 128));
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
                for (int e = 0; e < 2; ++e) mval[k][s2][e] = act_load(reinterpret_cast<const f32x4 *>(hb + s2 * MWS + (cc & 15) * 8 + 4 * e));
        }
    }
    // EPI_RESIDUAL: the residual values and the consumer's norm weight of this lane's output element are fetched now, not by a
    // dependent load in the tail of the kernel
    const int orow = (tile_c << 4) + r;
    uint16_t resv[ROWS], nwo = 0;
    if constexpr (EPI == EPI_RESIDUAL) {
#pragma unroll
        for (int i2 = 0; i2 < ROWS; ++i2) {
            const int arow = 4 * c + i2;
            resv[i2] = act_load(p.residual + (size_t)((arow < MR && arow < p.M) ? arow : 0) * K + orow);
        }
        nwo = (p.out_w ? p.norm_out : p.residual)[orow];
    }
    // producer-side sums of squares: the few partial loads go out first (they gate the normalisation of everything staged)
    // (ONE 16-byte load per lane and row: up to 256 partials per row, i.e. hidden sizes up to 4,096)
    constexpr bool NORMED = PRO == PRO_RMSNORM || PRO == PRO_RMS_WEIGHTED;
    const bool ss_given = NORMED && p.ss_in != nullptr && reg_path && p.ss_n <= 256 && (p.ss_n & 3) == 0;  // PRO_RMS_WEIGHTED: the launcher insists
    f32x4 ssv[MR];
    if constexpr (NORMED) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {  // unconditional load from a clamped address (no branch around a load), masked after
            const bool ok = ss_given && m < p.M && 4 * lane < p.ss_n;
            const float *src = ss_given ? p.ss_in : reinterpret_cast<const float *>(p.norm_w);
            const f32x4 v = act_load(reinterpret_cast<const f32x4 *>(src + (ok ? (size_t)m * p.ss_n + 4 * lane : 0)));
            ssv[m] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int k = 0; k < CCU; ++k) {
        const int cc = tid + k * T;
        const bool okc = reg_path && cc < cpr;
        const size_t coff = (size_t)(okc ? cc : 0) * 8;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if constexpr (PRO == PRO_ATTN_MERGE) {
                xv[k][m] = u32x4{0u, 0u, 0u, 0u};  // filled from the partials below, once they have arrived
                continue;
            }
            const bool ok = okc && m < p.M;
            if (Q3_ABL(4)) {
                xv[k][m] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
                continue;
            }
            xv[k][m] = act_load(reinterpret_cast<const u32x4 *>(p.a + (ok ? (size_t)m * N + coff : 0)));
            if (!ok) xv[k][m] = u32x4{0u, 0u, 0u, 0u};
        }
        if constexpr (PRO == PRO_RMSNORM)
            nwv[k] = *reinterpret_cast<const u32x4 *>(p.norm_w + coff);
        else
            nwv[k] = u32x4{0u, 0u, 0u, 0u};
    }
    Q3_STAMP(1);  // activation / partial loads issued
    __builtin_amdgcn_sched_barrier(0);  // activations first: vmcnt retires in issue order, and they are needed first
    u32x4 wq[LM];
    uint32_t sq[LM];
    {
        const uint32_t *sp = p.sbt + (size_t)tile_c * G * 16 + r;
#pragma unroll
        for (int i = 0; i < LM; ++i) sq[i] = sp[(size_t)min(g0 + i, G - 1) * 16];
        const u32x4 *wp = reinterpret_cast<const u32x4 *>(p.wt) + (size_t)tile_c * G * 64 + lane;
#pragma unroll
        for (int i = 0; i < LM; ++i) wq[i] = __builtin_nontemporal_load(wp + (size_t)min(g0 + i, G - 1) * 64);
    }
    __builtin_amdgcn_sched_barrier(0);
    Q3_STAMP(2);  // weight loads issued

    // ---- 2. activation rows -> LDS (bf16, natural order) + per-group sums; fused RMSNorm ----------------------------
    {
        // zero tails (read by the fixed-length compute body, multiplied by a zero scale)
        u32x4 *zx = reinterpret_cast<u32x4 *>(xs + (size_t)MR * xstride);
        for (int i = tid; i < Q3_LMAX * 16; i += T) zx[i] = u32x4{0u, 0u, 0u, 0u};
        if (tid < MR) *reinterpret_cast<u32x4 *>(xs + (size_t)tid * xstride + N) = u32x4{0u, 0u, 0u, 0u};  // row padding (Q3_PAD = 8)
        f32x4 *zs = reinterpret_cast<f32x4 *>(xsum + (size_t)G * 16);
        for (int i = tid; i < Q3_LMAX * 4; i += T) zs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto chunk_sumsq = [](const u32x4 &v) {
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = BF16::to_float((uint16_t)(v[e] & 0xffffu));
            const float hi = BF16::to_float((uint16_t)(v[e] >> 16));
            part += lo * lo + hi * hi;
        }
        return part;
    };
    // normalise (optional), store chunk (m, cc), and contribute to its group's sum (16 consecutive lanes = one group)
    auto finish_chunk = [&](int m, int cc, u32x4 v, const u32x4 &g, float iv) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[2 * e] = BF16::to_float((uint16_t)(v[e] & 0xffffu));
            f[2 * e + 1] = BF16::to_float((uint16_t)(v[e] >> 16));
        }
        if constexpr (PRO == PRO_RMSNORM) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[2 * e] = bf16_round(f[2 * e] * iv * BF16::to_float((uint16_t)(g[e] & 0xffffu)));
                f[2 * e + 1] = bf16_round(f[2 * e + 1] * iv * BF16::to_float((uint16_t)(g[e] >> 16)));
                v[e] = BF16::pack2(f[2 * e], f[2 * e + 1]);
            }
        }
        *reinterpret_cast<u32x4 *>(xs + (size_t)m * xstride + (size_t)cc * 8) = v;
        float sum = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
        sum = group16_sum(sum);
        if ((cc & 15) == 0) xsum[(cc >> 4) * 16 + m] = sum;
    };
    if constexpr (PRO == PRO_ATTN_MERGE) {
        // attn_merge_kernel (engine_kernels.h), per output column: gm = max_s m_s; f_s = 2^(m_s - gm); l = sum_s l_s f_s;
        // acc = sum_s v_s f_s (both in split order); out = bf16(l == 0 ? 0 : acc / l)
#pragma unroll
        for (int k = 0; k < MCCU; ++k) {
            float gm = -1e30f;
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) gm = fmaxf(gm, mml[k][s2][0]);
            float gl = 0.f, f2[NS];
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                f2[s2] = exp2_hw(mml[k][s2][0] - gm);
                gl += mml[k][s2][1] * f2[s2];
            }
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float acc1 = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2) acc1 += mval[k][s2][e >> 2][e & 3] * f2[s2];
                o[e] = gl == 0.f ? 0.f : acc1 / gl;
            }
            const bool okc = reg_path && tid + k * T < cpr;
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[k][0][e] = okc ? BF16::pack2(o[2 * e], o[2 * e + 1]) : 0u;
        }
    }
    float inv[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) inv[m] = 1.0f;
    auto inv_from_partials = [&]() {  // uniform: every wave adds the row's partials in the same fixed order; no LDS, no barrier
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            float v = (ssv[m][0] + ssv[m][1]) + (ssv[m][2] + ssv[m][3]);  // partials 4 l .. 4 l + 3 of lane l
            v = wave_sum(v);
            inv[m] = rsqrtf(v / (float)N + p.eps);
        }
    };
    if constexpr (PRO == PRO_RMSNORM) {
      if (ss_given) {
        inv_from_partials();
      } else if (!Q3_ABL(8)) {  // lab only, bit 8: pretend the inverse RMS is known
        float ss[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m) ss[m] = 0.f;
        if (reg_path) {
#pragma unroll
            for (int k = 0; k < CCU; ++k)
#pragma unroll
                for (int m = 0; m < MR; ++m) ss[m] += chunk_sumsq(xv[k][m]);
        } else {
            // long rows: pass 1 parks the raw rows in LDS and accumulates the sums of squares
            for (int cc = tid; cc < cpr; cc += T) {
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const bool ok = m < p.M;
                    u32x4 v = act_load(reinterpret_cast<const u32x4 *>(p.a + (ok ? ((size_t)m * N + (size_t)cc * 8) : 0)));
                    if (!ok) v = u32x4{0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4 *>(xs + (size_t)m * xstride + (size_t)cc * 8) = v;
                    ss[m] += chunk_sumsq(v);
                }
            }
        }
        float *scratch = red + (KS > 1 ? (size_t)CW * MR * 16 : 0);  // sized in qmv3_lds_bytes
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const float v = wave_sum(ss[m]);
            if (lane == 0) scratch[m * CW + wave] = v;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < CW; ++w) tot += scratch[m * CW + w];
            inv[m] = rsqrtf(tot / (float)N + p.eps);
        }
      }
    }
    if (reg_path && !Q3_ABL(2)) {
#pragma unroll
        for (int k = 0; k < CCU; ++k) {
            const int cc = tid + k * T;
            if (cc < cpr) {  // uniform per 16 lanes (cpr % 16 == 0)
#pragma unroll
                for (int m = 0; m < MR; ++m) finish_chunk(m, cc, xv[k][m], nwv[k], inv[m]);
            }
        }
    } else if (!reg_path) {
        for (int cc = tid; cc < cpr; cc += T) {  // each thread revisits exactly the chunks it parked
            u32x4 g = u32x4{0u, 0u, 0u, 0u};
            if constexpr (PRO == PRO_RMSNORM) g = *reinterpret_cast<const u32x4 *>(p.norm_w + (size_t)cc * 8);
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                u32x4 v;
                if constexpr (PRO == PRO_RMSNORM) {
                    v = *reinterpret_cast<const u32x4 *>(xs + (size_t)m * xstride + (size_t)cc * 8);
                } else {
                    const bool ok = m < p.M;
                    v = act_load(reinterpret_cast<const u32x4 *>(p.a + (ok ? ((size_t)m * N + (size_t)cc * 8) : 0)));
                    if (!ok) v = u32x4{0u, 0u, 0u, 0u};
                }
                finish_chunk(m, cc, v, g, inv[m]);
            }
        }
    }
    __syncthreads();  // activations are staged
    Q3_STAMP(3);  // activations staged

    float acc[ROWS];
#pragma unroll
    for (int i2 = 0; i2 < ROWS; ++i2) acc[i2] = 0.f;
    // LDS addresses: one VGPR base + compile-time offsets (rows >= MR of the A operand repeat row r % MR; never stored)
    const uint16_t *xbase = xs + (size_t)(r % MR) * xstride + 32 * c + (size_t)g0 * 128;
    const float *sbase = xsum + (size_t)g0 * 16 + 4 * c;
    uint32_t nib_mask = 0x000f000fu;
    uint32_t magic = 0x43004300u;
    asm volatile("" : "+s"(nib_mask));  // opaque constants: the mask in an SGPR, the magic in a VGPR (VOP3 takes one scalar
    asm volatile("" : "+v"(magic));     // operand and no second literal), so that every unpack is one v_and_or_b32
#pragma unroll
    for (int i = 0; i < LM; ++i) {
        if (Q3_ABL(1)) {
            acc[0] += __uint_as_float((wq[i][0] ^ wq[i][1] ^ wq[i][2] ^ wq[i][3]) & 0x3fffffffu) + __uint_as_float(sq[i] & 0x3fffffffu);
            continue;
        }
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32x4 bq = unpack_w4_bf16(wq[i][t], nib_mask, magic);
            const u32x4 ax = *reinterpret_cast<const u32x4 *>(xbase + i * 128 + 8 * t);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ax), __builtin_bit_cast(bf16x8_t, bq), d,
                                                        0, 0, 0);
        }
        const uint32_t sw = (g0 + i) < g1 ? sq[i] : 0u;  // surplus groups of the fixed-length body contribute nothing
        const float sc = __uint_as_float(sw << 16);
        const float be = __uint_as_float(sw & 0xffff0000u) - 128.0f * sc;
        if constexpr (ROWS == 4) {
            const f32x4 xg = *reinterpret_cast<const f32x4 *>(sbase + i * 16);
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) acc[i2] += sc * d[i2] + be * xg[i2];
        } else {
#pragma unroll
            for (int i2 = 0; i2 < ROWS; ++i2) acc[i2] += sc * d[i2] + be * sbase[i * 16 + i2];
        }
    }

    Q3_STAMP(4);  // unpack + MFMA done (all weights landed)
    if constexpr (KS > 1) {
#pragma unroll
        for (int i2 = 0; i2 < ROWS; ++i2) {
            const int arow = 4 * c + i2;
            if (arow < MR) red[((size_t)wave * MR + arow) * 16 + r] = acc[i2];
        }
        __syncthreads();
        if (ks != 0) {
            prof_end(p.prof, prof_t0);
            return;
        }
#pragma unroll
        for (int i2 = 0; i2 < ROWS; ++i2) {
            const int arow = 4 * c + i2;
            if (arow < MR) {
#pragma unroll
                for (int k2 = 1; k2 < KS; ++k2) acc[i2] += red[((size_t)(wave + k2) * MR + arow) * 16 + r];
            }
        }
    }

    if constexpr (PRO == PRO_RMS_WEIGHTED) {  // the rows were staged as x * w: the row's 1 / rms multiplies the finished sums
        inv_from_partials();
#pragma unroll
        for (int i2 = 0; i2 < ROWS; ++i2) acc[i2] *= (MR > 4 && c == 1) ? inv[(4 + i2) % MR] : inv[i2 % MR];
    }
    // epilogue: lane (weight row r, c) holds activation rows 4c .. 4c + ROWS - 1
#pragma unroll
    for (int i2 = 0; i2 < ROWS; ++i2) {
        const int arow = 4 * c + i2;
        const bool live = tile_ok && arow < MR && arow < p.M;
        if constexpr (EPI == EPI_SWIGLU) {
            const float gv = bf16_round(acc[i2]);  // rows interleaved: even = gate_i, odd = up_i
            const float uv = lane_xor1(gv);  // the odd lane next door holds up_i (only even lanes store)
            if (live && (r & 1) == 0)
                q3_store(&p.out[(size_t)arow * (K >> 1) + (orow >> 1)], BF16::from_float((gv / (1.0f + expf(-gv))) * uv));
        } else if constexpr (EPI == EPI_RESIDUAL) {
            float sq_v = 0.f;
            if (live) {
                const size_t o = (size_t)arow * K + orow;
                const uint16_t ov = BF16::from_float(BF16::to_float(resv[i2]) + bf16_round(acc[i2]));
                q3_store(&p.out[o], ov);
                sq_v = BF16::to_float(ov) * BF16::to_float(ov);
                if (p.out_w) q3_store(&p.out_w[o], BF16::from_float(BF16::to_float(ov) * BF16::to_float(nwo)));
            }
            if (p.ss_out) {  // uniform.  One partial per (activation row, 16-row tile): the squares of the stored bf16 values
                sq_v = group16_sum(sq_v);
                if (r == 0 && tile_ok && arow < MR && arow < p.M) q3_store(&p.ss_out[(size_t)arow * tiles + tile_c], sq_v);
            }
        } else {
            const uint16_t ov = BF16::from_float(acc[i2]);
            if (live) q3_store(&p.out[(size_t)arow * K + orow], ov);
            if (p.tile_max) {  // uniform
                const float v = live ? BF16::to_float(ov) : -INFINITY;
                const float m = group16_max(v);
                const float cand = (v == m && live) ? (float)orow : 3.0e38f;  // lowest index among the lanes that hold the maximum
                const float lowest = -group16_max(-cand);
                if (r == 0 && tile_ok && arow < MR && arow < p.M) act_store(&p.tile_max[(size_t)arow * tiles + tile_c], f32x2{m, lowest});
            }
        }
    }
    Q3_STAMP(5);  // reduced and stored
    prof_end(p.prof, prof_t0);
}

struct Qmv3Plan {
    int MR, KS, CW, LM, blocks;
    size_t lds;
    bool ok;
};
// The (MR, KS, CW, LM) combinations qmv3.hip instantiates (its Q3_TABLE, restated) = the ones the planner below can return:
//   1 / 2 / 4 / 8 rows x {1, 2, 4 reduction splits on 4 waves} x {4, 5, 8, 10 groups per wave} (8 rows: 2 / 4 splits only beyond 10 groups,
//   i.e. 8 or 10 per wave -- the finer cut of short reductions is a rule for 1-4 rows); 8 splits on 8 waves with 8 or 10 groups per wave
//   (8 splits mean more than 40 groups; four rows over 65-80 groups go to 16 waves instead); 4 rows x 16 splits on 16 waves x {4, 5}.
inline bool qmv3_has_variant(int MR, int KS, int CW, int LM) {
    const bool mr = MR == 1 || MR == 2 || MR == 4 || MR == 8;
    const bool lm4 = LM == 4 || LM == 5 || LM == 8 || LM == 10, hi = LM == 8 || LM == 10;
    if (CW == 4 && KS == 1) return mr && lm4;
    if (CW == 4 && (KS == 2 || KS == 4)) return mr && (MR == 8 ? hi : lm4);
    if (CW == 8 && KS == 8) return mr && hi && !(MR == 4 && LM == 10);
    return MR == 4 && KS == 16 && CW == 16 && (LM == 4 || LM == 5);
}
inline int qmv3_round_lm(int lper) { return lper <= 4 ? 4 : (lper <= 5 ? 5 : (lper <= 8 ? 8 : Q3_LMAX)); }
inline Qmv3Plan qmv3_plan(int M, int N, int K, int force_ks = 0, int force_cw = 0) {
    Qmv3Plan pl{};
    pl.MR = M <= 1 ? 1 : (M <= 2 ? 2 : (M <= 4 ? 4 : 8));
    const int G = N / 128;
    const int tiles = K / 16;
    int ks = 1;
    while (ks < 8 && (G + ks - 1) / ks > Q3_LMAX) ks *= 2;
    {
        // a grid of 1 .. 4 workgroups per CU with a ragged last round (gate|up: 608 workgroups on 256 CUs = 3 rounds for
        // 2.4 rounds of work) runs faster cut twice as fine (measured 7.3 -> 6.9 us); smaller and larger grids do not.
        // Up to 4 rows since round 3 (tools/lab/plan_lab, gate|up with weighted rows: 2 rows 8.32 -> 7.47 us, 4 rows 10.05 -> 9.39;
        // qkv and wo do not meet the condition and would lose: profiles/r03_labs/gemv_plan_sweep_rows_1_2_4.log)
        const int blocks = (tiles + (4 / ks) - 1) / (4 / (ks > 4 ? 4 : ks));
        const double x = blocks / 256.0;
        const double rounds = (double)(int)(x + 0.999999);
        if (M <= 4 && ks < 4 && x >= 1.0 && x < 4.0 && rounds / x > 1.2 && (G + 2 * ks - 1) / (2 * ks) >= 4) ks *= 2;
    }
    // four rows over a long reduction (w_down: 76 groups): 16 waves with 5 groups each stage the 4 x 9,728 activations and run
    // their chains twice as fast as 8 waves with 10 (tools/lab/plan_lab: 9.50 -> 7.65 us; at 1 / 2 rows the 8-wave cut wins,
    // 5.66 / 6.27 against 5.86 / 6.58)
    // (only where a 16-wave kernel exists: 4 or 5 groups per wave, i.e. reductions of up to 80 groups; longer ones keep 8 waves)
    if (pl.MR == 4 && ks == 8 && (G + 15) / 16 >= 4 && (G + 15) / 16 <= 5) ks = 16;
    if (force_ks > 0) ks = force_ks;
    pl.KS = ks;
    pl.CW = ks >= 8 ? ks : 4;
    if (force_cw > 0 && force_cw >= ks) pl.CW = force_cw;
    pl.LM = qmv3_round_lm((G + ks - 1) / ks);
    const int wr = pl.CW / pl.KS;
    pl.blocks = (tiles + wr - 1) / wr;
    pl.lds = qmv3_lds_bytes(pl.MR, N, pl.KS, pl.CW);
    // ok = the shape fits AND a kernel is compiled for (MR, KS, CW, LM): everything else goes to the packed-dot GEMV (qmv.h)
    pl.ok = K > 0 && K % 16 == 0 && N % 128 == 0 && M >= 1 && M <= 8 && (G + ks - 1) / ks <= Q3_LMAX &&
            pl.lds <= 150 * 1024 && qmv3_has_variant(pl.MR, pl.KS, pl.CW, pl.LM);
    return pl;
}

// PRO_RMS_WEIGHTED: the row must sit in the staging registers of one pass and its sums of squares in one 16-byte load per lane
inline bool qmv3_takes_weighted_rows(const Qmv3Plan &pl, int N, int ss_n) {
    return pl.ok && N / 8 <= (pl.MR <= 2 ? 3 : 2) * pl.CW * 64 && ss_n > 0 && ss_n <= 256 && (ss_n & 3) == 0;
}

// qmv3.hip
bool qmv3_variant_in_table(int MR, int KS, int CW, int LM);  // the instantiation table itself (CPU test vs qmv3_has_variant)
int launch_qmv3_bf16(const Qmv3Args &args, int pro, int epi, hipStream_t st, int force_ks = 0, int force_cw = 0);  // -1: not applicable
int launch_qmv3_attn_merge_bf16(const Qmv3Args &args, int n_splits, hipStream_t st);  // -1: not applicable, nothing launched
// standard checkpoint layout -> tiled layout (device to device, stream ordered)
int repack_w4_tiled(const uint32_t *w, const uint16_t *scales, const uint16_t *biases, uint32_t *wt, uint32_t *sbt, int K,
                    int N, hipStream_t st);

}  // namespace tl
