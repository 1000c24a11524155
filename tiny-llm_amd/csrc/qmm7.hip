// Launcher of the row-streaming batched-decode matmul (qmm7.h).
#include "qmm7.h"

namespace tl {

// The instantiation table, written once: QM7_TABLE(X) expands X(T, GPW) for every compiled pair (qmm7_has_variant restates it for the
// planner; tests/test_decode_plans_cpu.py holds the two together).  Every pair exists for 1 .. 4 row blocks and both epilogues.
#define QM7_TABLE(X) X(2, 5) X(5, 5)

bool qmm7_variant_in_table(int T, int GPW) {
#define QM7_MEMBER(Tv, GPWv) if (T == Tv && GPW == GPWv) return true;
    QM7_TABLE(QM7_MEMBER)
#undef QM7_MEMBER
    return false;
}

#ifndef QM7_LAB_NB
#define QM7_LAB_NB 0
#endif
#ifndef QM7_LAB_OCC
#define QM7_LAB_OCC 1
#endif
template <int MB, int T, int GPW, int EPI>
static int launch_one7(const Qmm6Args &a, const Qmm7Plan &pl, hipStream_t st) {
    auto kern = qmm7_kernel<MB, T, GPW, EPI, QM7_LAB_NB, QM7_LAB_OCC>;
    if (pl.lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
    hipLaunchKernelGGL(kern, dim3(pl.wgs, pl.row_blocks), dim3(QM7_WAVES * 64), pl.lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <int T, int GPW, int EPI>
static int launch_rows7(const Qmm6Args &a, const Qmm7Plan &pl, hipStream_t st) {
    switch (pl.MB) {
        case 1: return launch_one7<1, T, GPW, EPI>(a, pl, st);
        case 2: return launch_one7<2, T, GPW, EPI>(a, pl, st);
        case 3: return launch_one7<3, T, GPW, EPI>(a, pl, st);
        case 4: return launch_one7<4, T, GPW, EPI>(a, pl, st);
    }
    return -2;
}
template <int EPI>
static int launch_variant7(const Qmm6Args &a, const Qmm7Plan &pl, hipStream_t st) {
#define QM7_CASE(Tv, GPWv) \
    if (pl.T == Tv && pl.GPW == GPWv) return launch_rows7<Tv, GPWv, EPI>(a, pl, st);
    QM7_TABLE(QM7_CASE)
#undef QM7_CASE
    return -2;
}

int launch_qmm7_bf16(const Qmm6Args &args, int epi, hipStream_t st, int *n_wg) {
    if (!args.a_frag || !args.ss) return -1;  // weighted rows in fragment order, with their sums of squares
    if (args.ss_n <= 0 || args.ss_n > QM6_SS_MAX || args.ss_n % 4 != 0) return -1;
    if (args.out_w || args.ss_out || args.residual) return -1;
    Qmm7Plan pl = qmm7_plan(args.M, args.N, args.K);
    if (!pl.ok) return -1;
#if QM7_LAB_OCC == 2  // lab: the step's row blocks in two halves, two workgroups per CU
    if (pl.MB == 4 || pl.MB == 2) pl.MB /= 2, pl.row_blocks = 2, pl.lds = qmm7_lds_bytes(pl.MB, pl.T);
#endif
    if (n_wg) *n_wg = pl.wgs;
    if (epi == EPI_STORE) return launch_variant7<EPI_STORE>(args, pl, st);
    if (epi == EPI_SWIGLU) return launch_variant7<EPI_SWIGLU>(args, pl, st);
    return -1;
}

}  // namespace tl
