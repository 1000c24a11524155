// Launchers of the skinny batched-decode matmul (qmm3.h) and its slice-reduction / epilogue kernel.
#include "qmm3.h"

namespace tl {

// partial [slices][M][K] fp32 -> out bf16.  One thread = 4 consecutive output columns of one row (EPI_SWIGLU: 4 columns of
// the K/2-wide output = 8 interleaved gate/up columns).  Slices are added in index order: the result is deterministic.
template <int EPI>
__global__ __launch_bounds__(256) void qmm3_reduce_kernel(const float *__restrict__ partial, int slices, int M, int K,
                                                          const uint16_t *__restrict__ residual,
                                                          uint16_t *__restrict__ out, prof_t *prof) {
    const prof_t prof_t0 = prof_begin(prof);
    constexpr int IN_PER = EPI == EPI_SWIGLU ? 8 : 4;
    const int per_row = K / IN_PER;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < (long)M * per_row) {
        const int m = (int)(idx / per_row);
        const int q = (int)(idx - (long)m * per_row);
        const size_t in0 = (size_t)m * K + (size_t)q * IN_PER;
        const size_t slice_stride = (size_t)M * K;
        float acc[IN_PER];
#pragma unroll
        for (int e = 0; e < IN_PER; ++e) acc[e] = 0.f;
        for (int s = 0; s < slices; ++s) {
#pragma unroll
            for (int v = 0; v < IN_PER / 4; ++v) {
                const f32x4 x = *reinterpret_cast<const f32x4 *>(partial + (size_t)s * slice_stride + in0 + 4 * v);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * v + e] += x[e];
            }
        }
        uint16_t o[4];
        if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {  // rows interleaved: even = gate_i, odd = up_i (the engine's fused gate/up weight)
                const float gv = bf16_round(acc[2 * e]);
                const float uv = bf16_round(acc[2 * e + 1]);
                o[e] = BF16::from_float((gv / (1.0f + expf(-gv))) * uv);
            }
            *reinterpret_cast<uint2 *>(out + (size_t)m * (K / 2) + (size_t)q * 4) = *reinterpret_cast<const uint2 *>(o);
        } else if constexpr (EPI == EPI_RESIDUAL) {
            const uint2 rv = *reinterpret_cast<const uint2 *>(residual + in0);
            const uint16_t *rr = reinterpret_cast<const uint16_t *>(&rv);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = BF16::from_float(BF16::to_float(rr[e]) + bf16_round(acc[e]));
            *reinterpret_cast<uint2 *>(out + in0) = *reinterpret_cast<const uint2 *>(o);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = BF16::from_float(acc[e]);
            *reinterpret_cast<uint2 *>(out + in0) = *reinterpret_cast<const uint2 *>(o);
        }
    }
    prof_end(prof, prof_t0);
}

int launch_qmm3_reduce_bf16(const float *partial, int slices, int M, int K, int epi, const uint16_t *residual, uint16_t *out,
                            prof_t *prof, hipStream_t st) {
    if (K % 8 != 0) return -1;
    const long items = (long)M * (K / (epi == EPI_SWIGLU ? 8 : 4));
    const dim3 grid((unsigned)((items + 255) / 256)), block(256);
    if (epi == EPI_SWIGLU) hipLaunchKernelGGL(qmm3_reduce_kernel<EPI_SWIGLU>, grid, block, 0, st, partial, slices, M, K, residual, out, prof);
    else if (epi == EPI_RESIDUAL) hipLaunchKernelGGL(qmm3_reduce_kernel<EPI_RESIDUAL>, grid, block, 0, st, partial, slices, M, K, residual, out, prof);
    else hipLaunchKernelGGL(qmm3_reduce_kernel<EPI_STORE>, grid, block, 0, st, partial, slices, M, K, residual, out, prof);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_qmm3_bf16(const Qmm3Args &args, hipStream_t st) {
    const Qmm3Plan pl = qmm3_plan(args.M, args.N, args.K);
    if (!pl.ok) return -1;
    const dim3 grid(pl.tile_groups, pl.slices), block(QM3_WAVES * 64);
#define QM3_CASE(MBv, TWv, LMv)                                                                                     \
    if (pl.MB == MBv && pl.TW == TWv && pl.LM == LMv) {                                                             \
        auto kern = qmm3_kernel<MBv, TWv, LMv>;                                                                     \
        if (pl.lds > 64 * 1024)                                                                                     \
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds); \
        hipLaunchKernelGGL(kern, grid, block, pl.lds, st, args);                                                    \
        return hipGetLastError() == hipSuccess ? 0 : -1;                                                            \
    }
#define QM3_LM(MBv, TWv) QM3_CASE(MBv, TWv, 4) QM3_CASE(MBv, TWv, 5) QM3_CASE(MBv, TWv, 8) QM3_CASE(MBv, TWv, 10)
    QM3_LM(1, 1) QM3_LM(2, 1) QM3_CASE(4, 1, 4) QM3_CASE(4, 1, 5) QM3_CASE(4, 2, 4) QM3_CASE(4, 2, 5)
#undef QM3_LM
#undef QM3_CASE
    return -2;
}

}  // namespace tl
