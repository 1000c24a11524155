// Launcher of the register-resident batched-decode matmul (qmm6.h).
#include "qmm6.h"

namespace tl {

// The instantiation table, written once: QM6_TABLE(X) expands X(MB, GPW) for every compiled pair (qmm6_has_variant restates it for
// the planner; tests/test_decode_plans_cpu.py holds the two together).
#define QM6_TABLE(X) X(1, 2) X(1, 4) X(1, 5) X(1, 8) X(1, 19) X(2, 2) X(2, 4) X(2, 5) X(2, 8) X(4, 2) X(4, 4) X(4, 5)

bool qmm6_variant_in_table(int MB, int GPW) {
#define QM6_MEMBER(MBv, GPWv) if (MB == MBv && GPW == GPWv) return true;
    QM6_TABLE(QM6_MEMBER)
#undef QM6_MEMBER
    return false;
}

template <int MB, int GPW, int EPI, int NS>
static int launch_sets6(const Qmm6Args &a, const Qmm6Plan &pl, hipStream_t st) {
    if (pl.NSETS == NS) {
        const dim3 grid(pl.wgs, pl.row_blocks), block(QM6_WAVES * 64);
        // rows in fragment order: the consumers of weighted rows only (store / SwiGLU); a residual projection reads plain rows
        auto kern = qmm6_kernel<MB, GPW, EPI, NS, false>;
        if constexpr (EPI != EPI_RESIDUAL) {
            if (a.a_frag) kern = qmm6_kernel<MB, GPW, EPI, NS, true>;
        }
        if (pl.lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
        hipLaunchKernelGGL(kern, grid, block, pl.lds, st, a);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    if constexpr (NS > 1) return launch_sets6<MB, GPW, EPI, NS - 1>(a, pl, st);
    return -2;
}
template <int EPI>
static int launch_variant6(const Qmm6Args &a, const Qmm6Plan &pl, hipStream_t st) {
#define QM6_CASE(MBv, GPWv) \
    if (pl.MB == MBv && pl.GPW == GPWv) return launch_sets6<MBv, GPWv, EPI, qmm6_sets(MBv, GPWv)>(a, pl, st);
    QM6_TABLE(QM6_CASE)
#undef QM6_CASE
    return -2;
}

int launch_qmm6_bf16(const Qmm6Args &args, int epi, hipStream_t st, int *n_wg) {
    if (args.a_frag && (epi == EPI_RESIDUAL || !args.ss)) return -1;  // fragment order is the layout of WEIGHTED rows
    const Qmm6Plan pl = qmm6_plan(args.M, args.N, args.K, args.a_frag != 0);
    if (!pl.ok) return -1;
    if (args.ss && (args.ss_n <= 0 || args.ss_n > QM6_SS_MAX || args.ss_n % 4 != 0)) return -1;
    if (epi == EPI_RESIDUAL && !args.residual) return -1;
    if ((args.out_w != nullptr) != (args.norm_out != nullptr) || ((args.out_w || args.ss_out) && epi != EPI_RESIDUAL)) return -1;
    Qmm6Args a = args;
    a.tiles_per_wg = pl.tiles_per_wg;
    if (n_wg) *n_wg = pl.wgs * pl.row_blocks;
    if (epi == EPI_STORE) return launch_variant6<EPI_STORE>(a, pl, st);
    if (epi == EPI_RESIDUAL) return launch_variant6<EPI_RESIDUAL>(a, pl, st);
    if (epi == EPI_SWIGLU) return launch_variant6<EPI_SWIGLU>(a, pl, st);
    return -2;
}

}  // namespace tl
