// Launchers of the skinny batched-decode matmul (qmm3.h) and its slice-reduction / epilogue kernel.
#include <type_traits>

#include "qmm3.h"
#include "qmm6.h"

namespace tl {

// partial [slices][M][K] fp32 -> out bf16.  grid = (chunks of 256 threads, M rows); one thread = 4 consecutive output columns
// of its row (EPI_SWIGLU: 4 columns of the K/2-wide output = 8 interleaved gate/up columns).  Slices are added in index order
// (deterministic); the loads of 8 slices are in flight together (a dependent load per slice cost ~1 us each).
// SS: the workgroup also writes the sum of squares of its (rounded) outputs to ss_out[row][blockIdx.x] -- partials of the
// next projection's RMSNorm, reduced in a fixed order (wave shuffles, then the 4 waves in index order).
template <int EPI, bool SS>
__global__ __launch_bounds__(256) void qmm3_reduce_kernel(const float *__restrict__ partial, int slices, int M, int K,
                                                          const uint16_t *__restrict__ residual,
                                                          uint16_t *__restrict__ out, float *__restrict__ ss_out, prof_t *prof,
                                                          const uint16_t *__restrict__ norm_out, uint16_t *__restrict__ out_w, int out_w_frag) {
    __shared__ float wave_ss[4];
    const prof_t prof_t0 = prof_begin(prof);
    constexpr int IN_PER = EPI == EPI_SWIGLU ? 8 : 4;
    const int per_row = K / IN_PER;
    const int m = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    float sumsq = 0.f;
    if (q < per_row) {
        const size_t in0 = (size_t)m * K + (size_t)q * IN_PER;
        const size_t slice_stride = (size_t)M * K;
        float acc[IN_PER];
#pragma unroll
        for (int e = 0; e < IN_PER; ++e) acc[e] = 0.f;
        uint2 rv = uint2{0u, 0u}, nv = uint2{0u, 0u};  // the residual values go out with the first slices, not behind the last (one round trip, not two)
        if constexpr (EPI == EPI_RESIDUAL) {
            rv = *reinterpret_cast<const uint2 *>(residual + in0);
            if (out_w) nv = *reinterpret_cast<const uint2 *>(norm_out + (size_t)q * 4);  // uniform
        }
        // B planes in flight per round trip: 8, or -- 9 to 16 planes (w_down: 10) -- all of them at once (round 6: the second, dependent batch of two
        // planes cost the launch ~1.5 us); the planes are added in index order either way
        auto add_planes = [&](auto bc) __attribute__((always_inline)) {
            constexpr int B = decltype(bc)::value;
            for (int s0 = 0; s0 < slices; s0 += B) {
                f32x4 x[B][IN_PER / 4];
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    const size_t sidx = (size_t)min(s0 + j, slices - 1);
#pragma unroll
                    for (int v = 0; v < IN_PER / 4; ++v)
                        x[j][v] = *reinterpret_cast<const f32x4 *>(partial + sidx * slice_stride + in0 + 4 * v);
                }
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    if (s0 + j < slices) {  // uniform; no load inside
#pragma unroll
                        for (int v = 0; v < IN_PER / 4; ++v)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[4 * v + e] += x[j][v][e];
                    }
                }
            }
        };
        if (slices <= 8 || slices > 16 || EPI == EPI_SWIGLU) add_planes(std::integral_constant<int, 8>{});  // uniform
        else if (slices <= 12) add_planes(std::integral_constant<int, 12>{});
        else add_planes(std::integral_constant<int, 16>{});
        uint16_t o[4];
        if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {  // rows interleaved: even = gate_i, odd = up_i (the engine's fused gate/up weight)
                const float gv = bf16_round(acc[2 * e]);
                const float uv = bf16_round(acc[2 * e + 1]);
                o[e] = BF16::from_float((gv / (1.0f + expf(-gv))) * uv);
            }
            act_store(reinterpret_cast<u32x2 *>(out + (size_t)m * (K / 2) + (size_t)q * 4), *reinterpret_cast<const u32x2 *>(o));
        } else if constexpr (EPI == EPI_RESIDUAL) {
            const uint16_t *rr = reinterpret_cast<const uint16_t *>(&rv);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = BF16::from_float(BF16::to_float(rr[e]) + bf16_round(acc[e]));
            act_store(reinterpret_cast<u32x2 *>(out + in0), *reinterpret_cast<const u32x2 *>(o));
            if (out_w) {  // uniform: the rows weighted for the next RMSNorm's consumer (qmm6.h), bf16(out * norm_out)
                const uint16_t *nn = reinterpret_cast<const uint16_t *>(&nv);
                uint16_t ow[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) ow[e] = BF16::from_float(BF16::to_float(o[e]) * BF16::to_float(nn[e]));
                // fragment order (qmm6.h): the four columns of a thread stay together inside an 8-column run
                const size_t wo = out_w_frag ? qmm6_frag_offset(m, q * 4, K) : in0;
                act_store(reinterpret_cast<u32x2 *>(out_w + wo), *reinterpret_cast<const u32x2 *>(ow));
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = BF16::from_float(acc[e]);
            act_store(reinterpret_cast<u32x2 *>(out + in0), *reinterpret_cast<const u32x2 *>(o));
        }
        if constexpr (SS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = BF16::to_float(o[e]);
                sumsq += v * v;
            }
        }
    }
    if constexpr (SS) {
        const float w = wave_sum(sumsq);
        if ((threadIdx.x & 63) == 0) wave_ss[threadIdx.x >> 6] = w;
        __syncthreads();
        if (threadIdx.x == 0) {
            act_store(&ss_out[(size_t)m * QM3_SS + blockIdx.x], (wave_ss[0] + wave_ss[1]) + (wave_ss[2] + wave_ss[3]));
            if (blockIdx.x == 0)
                for (int i = gridDim.x; i < QM3_SS; ++i) act_store(&ss_out[(size_t)m * QM3_SS + i], 0.f);
        }
    }
    prof_end(prof, prof_t0);
}

int launch_qmm3_reduce_bf16(const float *partial, int slices, int M, int K, int epi, const uint16_t *residual, uint16_t *out,
                            prof_t *prof, hipStream_t st, float *ss_out, int *n_wg, const uint16_t *norm_out, uint16_t *out_w, int out_w_frag) {
    if ((out_w != nullptr) != (norm_out != nullptr) || (out_w && epi != EPI_RESIDUAL) || (out_w_frag && K % 128 != 0)) return -1;
    if (K % 8 != 0 || M < 1 || M > 65535) return -1;
    const int per_row = K / (epi == EPI_SWIGLU ? 8 : 4);
    const dim3 grid((unsigned)((per_row + 255) / 256), (unsigned)M), block(256);
    if (ss_out && !qmm3_reduce_can_emit_ss(epi, K)) return -1;
    if (n_wg) *n_wg = (int)(grid.x * grid.y);
    if (epi == EPI_SWIGLU) hipLaunchKernelGGL((qmm3_reduce_kernel<EPI_SWIGLU, false>), grid, block, 0, st, partial, slices, M, K, residual, out, ss_out, prof, norm_out, out_w, out_w_frag);
    else if (epi == EPI_RESIDUAL && ss_out) hipLaunchKernelGGL((qmm3_reduce_kernel<EPI_RESIDUAL, true>), grid, block, 0, st, partial, slices, M, K, residual, out, ss_out, prof, norm_out, out_w, out_w_frag);
    else if (epi == EPI_RESIDUAL) hipLaunchKernelGGL((qmm3_reduce_kernel<EPI_RESIDUAL, false>), grid, block, 0, st, partial, slices, M, K, residual, out, ss_out, prof, norm_out, out_w, out_w_frag);
    else if (ss_out) hipLaunchKernelGGL((qmm3_reduce_kernel<EPI_STORE, true>), grid, block, 0, st, partial, slices, M, K, residual, out, ss_out, prof, norm_out, out_w, out_w_frag);
    else hipLaunchKernelGGL((qmm3_reduce_kernel<EPI_STORE, false>), grid, block, 0, st, partial, slices, M, K, residual, out, ss_out, prof, norm_out, out_w, out_w_frag);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int qmm3_num_cus() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            return 256;
        return cus;
    }();
    return n;
}
int qmm3_default_mode() { return -1; }  // the grid is chosen by shape (qmm3_prefers_persistent); callers that need one pass `mode`
int qmm3_forced_lm() { return 0; }

int launch_qmm3_bf16(const Qmm3Args &args, hipStream_t st, int pro, int mode) {
    const Qmm3Plan pl = qmm3_plan(args.M, args.N, args.K, mode);
    if (!pl.ok) return -1;
    if (pro == PRO_RMSNORM && (!args.ss || !args.norm_w || !qmm3_takes_ss(args.ss_n))) return -1;
    const dim3 grid(pl.grid_x, pl.slices), block(QM3_WAVES * 64);
    // The staging's row loads go out ahead of the weight stream up to 16 rows (qmm3.h, SF = MB == 1): the staged slice is small there
    // and the kernel is a latency chain -- rows first take 1.0-1.5 us off every projection (8 sequences 1.898 -> 1.767 ms per step,
    // 16: 2.067 -> 1.979, same-box A/B, profiles/r04_labs/skinny_matmul_rows_before_weights_ab*.jsonl); at 32 / 64 rows the slice is
    // 64-128 KiB per CU and delays the weight stream instead (2.87 -> 3.02, 4.675 -> 4.775 ms): weights first there.
    if (pl.persistent) {
        const dim3 pgrid(pl.grid_x);
#define QM3P_CASE(MBv, NUv)                                                                                          \
    if (pl.MB == MBv && pl.NU == NUv) {                                                                              \
        auto kern = pro == PRO_RMSNORM ? qmm3p_kernel<MBv, NUv, PRO_RMSNORM, MBv == 1> : qmm3p_kernel<MBv, NUv, PRO_NONE, MBv == 1>; \
        if (pl.lds > 64 * 1024)                                                                                      \
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);  \
        hipLaunchKernelGGL(kern, pgrid, block, pl.lds, st, args, pl.pgrid);                                          \
        return hipGetLastError() == hipSuccess ? 0 : -1;                                                             \
    }
        QM3P_CASE(1, 1) QM3P_CASE(1, 2) QM3P_CASE(2, 1) QM3P_CASE(2, 2) QM3P_CASE(3, 1) QM3P_CASE(3, 2) QM3P_CASE(4, 1) QM3P_CASE(4, 2)
#undef QM3P_CASE
        return -2;
    }
#define QM3_CASE(MBv, TWv, LMv)                                                                                     \
    if (pl.MB == MBv && pl.TW == TWv && pl.LM == LMv) {                                                             \
        auto kern = pro == PRO_RMSNORM ? qmm3_kernel<MBv, TWv, LMv, PRO_RMSNORM, MBv == 1> : qmm3_kernel<MBv, TWv, LMv, PRO_NONE, MBv == 1>; \
        if (pl.lds > 64 * 1024)                                                                                     \
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds); \
        hipLaunchKernelGGL(kern, grid, block, pl.lds, st, args);                                                    \
        return hipGetLastError() == hipSuccess ? 0 : -1;                                                            \
    }
#define QM3_LM(MBv, TWv) QM3_CASE(MBv, TWv, 4) QM3_CASE(MBv, TWv, 5) QM3_CASE(MBv, TWv, 8) QM3_CASE(MBv, TWv, 10)
    QM3_LM(1, 1) QM3_LM(2, 1) QM3_CASE(4, 1, 4) QM3_CASE(4, 1, 5) QM3_CASE(4, 2, 4)
#undef QM3_LM
#undef QM3_CASE
    return -2;
}

}  // namespace tl
