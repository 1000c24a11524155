// Instantiations of the decode GEMV with fused prologue/epilogue (decode engine fast path).
//   PRO_RMSNORM : the activation row is RMS-normalised (and rounded to bf16, the reference op
//                 boundary of FastRMSNorm, week2_kernels.py:10-19) while it is staged into LDS.
//   EPI_RESIDUAL: out = bf16(residual + bf16(acc))      (h = x + attn(x), qwen3_week3.py:206-207)
//   EPI_SWIGLU  : rows interleaved gate/up -> bf16(silu(bf16 gate) * bf16 up) (week2_kernels.metal:107-117)
#include "qmv.h"

namespace tl {

template <int PRO, int EPI>
static int launch_variant(const QmvArgs &args, hipStream_t st) {
    const QmvPlan pl = qmv_plan(args.M, args.N, args.K);
    if (pl.lds > 150 * 1024) return -1;
    const dim3 grid(pl.blocks), block(256);
#define QMV_CASE(MRv, WNv, RPLv)                                                                                    \
    if (pl.MR == MRv && pl.WN == WNv && pl.RPL == RPLv) {                                                           \
        auto kern = qmv_kernel<BF16, MRv, WNv, RPLv, PRO, EPI>;                                                     \
        if (pl.lds > 64 * 1024)                                                                                     \
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds); \
        hipLaunchKernelGGL(kern, grid, block, pl.lds, st, args);                                                    \
        return 0;                                                                                                   \
    }
#define QMV_MR(MRv) QMV_CASE(MRv, 1, 2) QMV_CASE(MRv, 1, 1) QMV_CASE(MRv, 2, 1) QMV_CASE(MRv, 4, 1)
    QMV_MR(1) QMV_MR(2) QMV_MR(4) QMV_MR(8)
#undef QMV_MR
#undef QMV_CASE
    return -2;
}

int launch_qmv_fused_bf16(const QmvArgs &args, int pro, int epi, hipStream_t st) {
    if (pro == PRO_NONE && epi == EPI_STORE) return launch_variant<PRO_NONE, EPI_STORE>(args, st);
    if (pro == PRO_RMSNORM && epi == EPI_STORE) return launch_variant<PRO_RMSNORM, EPI_STORE>(args, st);
    if (pro == PRO_NONE && epi == EPI_RESIDUAL) return launch_variant<PRO_NONE, EPI_RESIDUAL>(args, st);
    if (pro == PRO_RMSNORM && epi == EPI_SWIGLU) return launch_variant<PRO_RMSNORM, EPI_SWIGLU>(args, st);
    return -2;
}

}  // namespace tl
