// Launcher of the tiled-layout MFMA decode GEMV (qmv3.h) + the one-time checkpoint repack.
#include "qmv3.h"

namespace tl {

// standard [K][N/8] words (nibble i of word j = element 8j+i, reference quantize.py:113-115) + [K][G] bf16 scales/biases
//   -> wt  [K/16][G][64 lanes][4 words]  lane = r + 16c, word t = original word g*16 + 4c + t of row 16*tile + r,
//          nibbles reordered to (q0,q2,q4,q6,q1,q3,q5,q7)
//   -> sbt [K/16][G][16]  scale | bias << 16
__global__ __launch_bounds__(256) void repack_w4_tiled_kernel(const uint32_t *__restrict__ w,
                                                              const uint16_t *__restrict__ scales,
                                                              const uint16_t *__restrict__ biases,
                                                              uint32_t *__restrict__ wt, uint32_t *__restrict__ sbt, int K,
                                                              int N) {
    const int G = N >> 7, words = N >> 3;
    const size_t total = (size_t)K * words;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // output word index
    if (idx < total) {
        const int t = (int)(idx & 3);
        const int lane = (int)((idx >> 2) & 63);
        const size_t tg = idx >> 8;  // tile * G + g
        const int g = (int)(tg % G);
        const size_t tile = tg / G;
        const int r = lane & 15, c = lane >> 4;
        const uint32_t v = w[(tile * 16 + r) * words + g * 16 + 4 * c + t];
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o |= ((v >> (8 * i)) & 0xfu) << (4 * i);            // q_2i   -> nibble i
            o |= ((v >> (8 * i + 4)) & 0xfu) << (4 * i + 16);   // q_2i+1 -> nibble i + 4
        }
        wt[idx] = o;
    }
    const size_t stotal = (size_t)K * G;
    if (idx < stotal) {
        const int r = (int)(idx & 15);
        const size_t tg = idx >> 4;
        const int g = (int)(tg % G);
        const size_t tile = tg / G;
        const size_t src = (tile * 16 + r) * G + g;
        sbt[idx] = (uint32_t)scales[src] | ((uint32_t)biases[src] << 16);
    }
}

int repack_w4_tiled(const uint32_t *w, const uint16_t *scales, const uint16_t *biases, uint32_t *wt, uint32_t *sbt, int K,
                    int N, hipStream_t st) {
    const size_t total = (size_t)K * (N / 8);
    hipLaunchKernelGGL(repack_w4_tiled_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, scales, biases, wt,
                       sbt, K, N);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// The instantiation table, written once: Q3_TABLE(X) expands X(MR, KS, CW, LM) for every compiled combination -- exactly the
// combinations qmv3_plan can return (round 5: the 45 of 98 that no shape reaches -- 8-wave workgroups with 2 / 4 reduction splits,
// 8 splits with 4 / 5 groups per wave, 8 rows on the finer cut of 1-4 rows -- are gone: 231 kernels; tests/test_decode_plans_cpu.py
// sweeps the planner and fails on an entry nothing reaches).  Laboratories instantiate what they sweep themselves (tools/lab/plan_lab.hip).
#define Q3_LM4(X, MRv, KSv, CWv) X(MRv, KSv, CWv, 4) X(MRv, KSv, CWv, 5) X(MRv, KSv, CWv, 8) X(MRv, KSv, CWv, 10)
#define Q3_LMHI(X, MRv, KSv, CWv) X(MRv, KSv, CWv, 8) X(MRv, KSv, CWv, 10)
#define Q3_TABLE(X)                                                                                                     \
    Q3_LM4(X, 1, 1, 4) Q3_LM4(X, 1, 2, 4) Q3_LM4(X, 1, 4, 4) Q3_LMHI(X, 1, 8, 8)                                        \
    Q3_LM4(X, 2, 1, 4) Q3_LM4(X, 2, 2, 4) Q3_LM4(X, 2, 4, 4) Q3_LMHI(X, 2, 8, 8)                                        \
    Q3_LM4(X, 4, 1, 4) Q3_LM4(X, 4, 2, 4) Q3_LM4(X, 4, 4, 4) X(4, 8, 8, 8)                                              \
    X(4, 16, 16, 4) X(4, 16, 16, 5) /* qmv3_plan: four rows over a long reduction (8 x <= 10 groups cut 16 x <= 5) */   \
    Q3_LM4(X, 8, 1, 4) Q3_LMHI(X, 8, 2, 4) Q3_LMHI(X, 8, 4, 4) Q3_LMHI(X, 8, 8, 8)

// what the table holds, for the CPU test that keeps qmv3_has_variant (qmv3.h, used by the planner) in step with it
bool qmv3_variant_in_table(int MR, int KS, int CW, int LM) {
#define Q3_MEMBER(MRv, KSv, CWv, LMv) if (MR == MRv && KS == KSv && CW == CWv && LM == LMv) return true;
    Q3_TABLE(Q3_MEMBER)
#undef Q3_MEMBER
    return false;
}

template <int PRO, int EPI>
static int launch_variant3(const Qmv3Args &args, hipStream_t st, int force_ks, int force_cw) {
    const Qmv3Plan pl = qmv3_plan(args.M, args.N, args.K, force_ks, force_cw);
    if (!pl.ok) return -1;
    const dim3 grid(pl.blocks), block(pl.CW * 64);
#define Q3_CASE(MRv, KSv, CWv, LMv)                                                                                 \
    if (pl.MR == MRv && pl.KS == KSv && pl.CW == CWv && pl.LM == LMv) {                                             \
        auto kern = qmv3_kernel<MRv, KSv, CWv, PRO, EPI, LMv>;                                                      \
        if (pl.lds > 64 * 1024)                                                                                     \
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds); \
        hipLaunchKernelGGL(kern, grid, block, pl.lds, st, args);                                                    \
        return 0;                                                                                                   \
    }
    Q3_TABLE(Q3_CASE)
#undef Q3_CASE
    return -2;
}

// PRO_ATTN_MERGE + EPI_RESIDUAL, one row: the wo projection reading the decode-attention split partials (Qmv3Args::merge_ws).
// Instantiated for the plans a single row takes (4 waves with 2- or 4-way, 8 waves with 8-way reduction splits) and 2 / 4 / 8
// attention splits; every chunk of the row must sit in two register sets (N / 8 <= 2 * threads).  -1: not applicable (the caller keeps
// the merge launch).
template <int NS>
static int launch_merge_variant3(const Qmv3Args &args, hipStream_t st, const Qmv3Plan &pl) {
    const dim3 grid(pl.blocks), block(pl.CW * 64);
#define Q3_MCASE(KSv, CWv, LMv)                                                                                      \
    if (pl.KS == KSv && pl.CW == CWv && pl.LM == LMv) {                                                              \
        auto kern = qmv3_kernel<1, KSv, CWv, PRO_ATTN_MERGE, EPI_RESIDUAL, LMv, NS>;                                  \
        if (pl.lds > 64 * 1024)                                                                                      \
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);  \
        hipLaunchKernelGGL(kern, grid, block, pl.lds, st, args);                                                     \
        return 0;                                                                                                    \
    }
#define Q3_MLM(KSv, CWv) Q3_MCASE(KSv, CWv, 4) Q3_MCASE(KSv, CWv, 5) Q3_MCASE(KSv, CWv, 8) Q3_MCASE(KSv, CWv, 10)
    Q3_MLM(2, 4) Q3_MLM(4, 4) Q3_MCASE(8, 8, 8) Q3_MCASE(8, 8, 10)  // (8 splits mean more than 40 groups: 8 or 10 per wave)
#undef Q3_MLM
#undef Q3_MCASE
    return -1;
}
int launch_qmv3_attn_merge_bf16(const Qmv3Args &args, int n_splits, hipStream_t st) {
    if (args.M != 1 || !args.merge_ws || !args.residual || args.N % 128 != 0) return -1;
    const Qmv3Plan pl = qmv3_plan(1, args.N, args.K);
    if (!pl.ok || pl.MR != 1 || args.N / 8 > 2 * pl.CW * 64) return -1;  // two chunk sets per thread (qmv3.h, MCCU)
    if (n_splits == 2) return launch_merge_variant3<2>(args, st, pl);
    if (n_splits == 4) return launch_merge_variant3<4>(args, st, pl);
    if (n_splits == 8) return launch_merge_variant3<8>(args, st, pl);
    return -1;
}

int launch_qmv3_bf16(const Qmv3Args &args, int pro, int epi, hipStream_t st, int force_ks, int force_cw) {
    if (pro == PRO_NONE && epi == EPI_STORE) return launch_variant3<PRO_NONE, EPI_STORE>(args, st, force_ks, force_cw);
    if (pro == PRO_RMSNORM && epi == EPI_STORE) return launch_variant3<PRO_RMSNORM, EPI_STORE>(args, st, force_ks, force_cw);
    if (pro == PRO_NONE && epi == EPI_RESIDUAL) return launch_variant3<PRO_NONE, EPI_RESIDUAL>(args, st, force_ks, force_cw);
    if (pro == PRO_RMSNORM && epi == EPI_SWIGLU) return launch_variant3<PRO_RMSNORM, EPI_SWIGLU>(args, st, force_ks, force_cw);
    if (pro == PRO_RMS_WEIGHTED && epi == EPI_SWIGLU) {  // the one consumer of weighted rows there is: gate|up
        if (!args.ss_in || !qmv3_takes_weighted_rows(qmv3_plan(args.M, args.N, args.K, force_ks, force_cw), args.N, args.ss_n)) return -1;
        return launch_variant3<PRO_RMS_WEIGHTED, EPI_SWIGLU>(args, st, force_ks, force_cw);
    }
    return -2;
}

}  // namespace tl
