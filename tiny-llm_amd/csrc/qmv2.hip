// Instantiations + launcher of the MFMA decode GEMV (qmv2.h).
#include "qmv2.h"

namespace tl {

template <int PRO, int EPI>
static int launch_variant2(const QmvArgs &args, hipStream_t st, int force_ks) {
    const Qmv2Plan pl = qmv2_plan(args.M, args.N, args.K, PRO == PRO_RMSNORM, force_ks);
    if (!pl.ok) return -1;
    const dim3 grid(pl.blocks), block(pl.WAVES * 64);
#define Q2_CASE(MRv, KSv, WAVESv)                                                                                   \
    if (pl.MR == MRv && pl.KS == KSv && pl.WAVES == WAVESv) {                                                       \
        auto kern = qmv2_kernel<MRv, KSv, WAVESv, PRO, EPI>;                                                        \
        if (pl.lds > 64 * 1024)                                                                                     \
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds); \
        hipLaunchKernelGGL(kern, grid, block, pl.lds, st, args);                                                    \
        return 0;                                                                                                   \
    }
#define Q2_MR(MRv) Q2_CASE(MRv, 1, 4) Q2_CASE(MRv, 2, 4) Q2_CASE(MRv, 4, 4) Q2_CASE(MRv, 8, 8)
    Q2_MR(1) Q2_MR(2) Q2_MR(4) Q2_MR(8)
#undef Q2_MR
#undef Q2_CASE
    return -2;
}

int launch_qmv2_bf16(const QmvArgs &args, int pro, int epi, hipStream_t st, int force_ks) {
    if (pro == PRO_NONE && epi == EPI_STORE) return launch_variant2<PRO_NONE, EPI_STORE>(args, st, force_ks);
    if (pro == PRO_RMSNORM && epi == EPI_STORE) return launch_variant2<PRO_RMSNORM, EPI_STORE>(args, st, force_ks);
    if (pro == PRO_NONE && epi == EPI_RESIDUAL) return launch_variant2<PRO_NONE, EPI_RESIDUAL>(args, st, force_ks);
    if (pro == PRO_RMSNORM && epi == EPI_SWIGLU) return launch_variant2<PRO_RMSNORM, EPI_SWIGLU>(args, st, force_ks);
    return -2;
}

}  // namespace tl
