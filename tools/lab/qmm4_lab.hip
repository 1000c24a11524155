// Prototype (lab only, not linked into the product): a prefill W4 GEMM over the TILED weight layout in the algebraic form of the
// decode GEMV -- the MFMA multiplies activations by the raw codes (128 + q as bf16, one v_and_or per pair), a group's 32 x 32 block
// sums D are scaled once per group: acc += s * D + (beta - 128 s) * sum_k a -- on a 2 x 2 wave tiling of a 128 x 128 tile (each wave
// 64 rows x 64 columns: an activation fragment read from LDS feeds two MFMAs, a weight word two).  Measured against the product's
// prefill GEMM (tl_quantized_matmul, exact bf16 dequantisation, 4 x 1 waves).  Row-group sums of the activations come from a small
// kernel (in the engine they would come from the producer of the rows).
// build: see tools/lab/run_qmm4_lab.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "../../include/tinyllm_hip.h"
#include "../../tiny-llm_amd/csrc/qmv3.h"
using namespace tl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(64) void row_group_sums_kernel(const uint16_t *__restrict__ a, float *__restrict__ asum, int N) {
    const int row = blockIdx.x, G = N >> 7, lane = threadIdx.x;
    for (int g = 0; g < G; ++g) {
        const uint32_t v = *reinterpret_cast<const uint32_t *>(a + (size_t)row * N + g * 128 + 2 * lane);
        float s = BF16::to_float((uint16_t)(v & 0xffffu)) + BF16::to_float((uint16_t)(v >> 16));
        s = wave_sum(s);
        if (lane == 0) asum[(size_t)row * G + g] = s;
    }
}

typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));

// 128 x 128 tile, 4 waves as 2 (rows) x 2 (columns); 64-wide reduction stages, two per quantisation group
__global__ __launch_bounds__(256, 2) void qmm4_kernel(const uint32_t *__restrict__ wt, const uint32_t *__restrict__ sbt,
                                                      const uint16_t *__restrict__ a, const float *__restrict__ asum,
                                                      uint16_t *__restrict__ out, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t *atile = reinterpret_cast<uint16_t *>(smem);                 // [2][128 * 64]
    float *asl = reinterpret_cast<float *>(smem + 2 * 128 * 64 * 2);      // [G][128]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l32 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int G = N >> 7;
    const int bn0 = blockIdx.x * 128, bm0 = blockIdx.y * 128;

    for (int i = tid; i < 128 * G; i += 256) {  // asl[g][row]
        const int g = i >> 7, row = i & 127;
        asl[i] = asum[(size_t)min(bm0 + row, M - 1) * G + g];
    }

    // weights: column of this lane in each of its two 32-column blocks
    const uint32_t *wbase[2];
    const uint32_t *sbase[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int col = min(bn0 + wn * 64 + nb * 32 + l32, K - 1);
        const int tile = col >> 4, r = col & 15;
        wbase[nb] = wt + (size_t)tile * G * 256 + r * 4 + 2 * h;   // + g * 256 + c * 64
        sbase[nb] = sbt + (size_t)tile * G * 16 + r;               // + g * 16
    }

    f32x16 acc[2][2], D[2][2];
    f32x16 zero;
#pragma unroll
    for (int j = 0; j < 16; ++j) zero[j] = 0.f;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = zero, D[mb][nb] = zero;

    u32x4 areg[4];
    auto load_a = [&](int j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = tid + q * 256, r = c >> 3, ch = c & 7;
            areg[q] = *reinterpret_cast<const u32x4 *>(a + (size_t)min(bm0 + r, M - 1) * N + j * 64 + ch * 8);
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = tid + q * 256, r = c >> 3, ch = c & 7;
            *reinterpret_cast<u32x4 *>(&atile[buf * 8192 + r * 64 + ((ch ^ ((r >> 1) & 7)) * 8)]) = areg[q];
        }
    };
    u32x2v wcur[2][2], wnext[2][2];
    uint32_t sw[2] = {0u, 0u};
    auto load_w = [&](int j, u32x2v (&w)[2][2]) {
        const int g = j >> 1, half = j & 1;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int cl = 0; cl < 2; ++cl)
                w[nb][cl] = *reinterpret_cast<const u32x2v *>(wbase[nb] + (size_t)g * 256 + (2 * half + cl) * 64);
    };

    uint32_t nib_mask = 0x000f000fu, magic = 0x43004300u;
    asm volatile("" : "+s"(nib_mask));
    asm volatile("" : "+v"(magic));
    const int stages = 2 * G;
    load_a(0);
    load_w(0, wcur);
    int buf = 0;
    for (int j = 0; j < stages; ++j) {
        store_a(buf);
        __syncthreads();
        if (j + 1 < stages) {
            load_a(j + 1);
            load_w(j + 1, wnext);
        }
        if ((j & 1) == 1) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) sw[nb] = sbase[nb][(size_t)(j >> 1) * 16];
        }
#pragma unroll
        for (int cl = 0; cl < 2; ++cl) {
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const int kch = 4 * cl + 2 * h + sp;  // 16-byte chunk of the 64-wide stage this lane's k-octet lies in
                u32x4 af[2];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const int r = wm * 64 + mb * 32 + l32;
                    af[mb] = *reinterpret_cast<const u32x4 *>(&atile[buf * 8192 + r * 64 + ((kch ^ ((r >> 1) & 7)) * 8)]);
                }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const u32x4 bq = unpack_w4_bf16(wcur[nb][cl][sp], nib_mask, magic);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
                        D[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[mb]),
                                                                            __builtin_bit_cast(bf16x8_t, bq), D[mb][nb], 0, 0, 0);
                }
            }
        }
        if ((j & 1) == 1) {  // a quantisation group is complete: acc += s * D + (beta - 128 s) * sum_k a
            const int g = j >> 1;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                f32x4 as4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) as4[q] = *reinterpret_cast<const f32x4 *>(&asl[g * 128 + wm * 64 + mb * 32 + 8 * q + 4 * h]);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const float sc = __uint_as_float(sw[nb] << 16);
                    const float be = __uint_as_float(sw[nb] & 0xffff0000u) - 128.0f * sc;
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) acc[mb][nb][jj] += sc * D[mb][nb][jj] + be * as4[jj >> 2][jj & 3];
                    D[mb][nb] = zero;
                }
            }
        }
        buf ^= 1;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int cl = 0; cl < 2; ++cl) wcur[nb][cl] = wnext[nb][cl];
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int col = bn0 + wn * 64 + nb * 32 + l32;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int m = bm0 + wm * 64 + mb * 32 + (jj & 3) + 8 * (jj >> 2) + 4 * h;
                if (m < M && col < K) out[(size_t)m * K + col] = BF16::from_float(acc[mb][nb][jj]);
            }
        }
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 2048;
    struct Shape { const char *name; int K, N; } shapes[] = {{"qkv", 6144, 2560}, {"o", 2560, 4096}, {"gate_up", 19456, 2560}, {"down", 2560, 9728}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double tot_new = 0, tot_old = 0, tot_flop = 0;
    for (auto &sh : shapes) {
        const int K = sh.K, N = sh.N, G = N / 128;
        uint32_t *w, *wt4, *sb4; uint16_t *s, *b, *a, *out_ref, *out_new; float *asum; void *ws = nullptr;
        CK(hipMalloc(&w, (size_t)K * N / 2)); CK(hipMalloc(&s, (size_t)K * G * 2)); CK(hipMalloc(&b, (size_t)K * G * 2));
        CK(hipMalloc(&wt4, (size_t)K * N / 2 + 16384)); CK(hipMalloc(&sb4, (size_t)K * G * 4 + 1024));
        CK(hipMalloc(&a, (size_t)M * N * 2)); CK(hipMalloc(&out_ref, (size_t)M * K * 2)); CK(hipMalloc(&out_new, (size_t)M * K * 2));
        CK(hipMalloc(&asum, (size_t)M * G * 4));
        std::vector<uint32_t> hw((size_t)K * N / 8); for (auto &v : hw) v = (uint32_t)rand() * 2654435761u + (uint32_t)rand();
        std::vector<uint16_t> hs((size_t)K * G), hb((size_t)K * G), ha((size_t)M * N);
        for (auto &v : hs) v = (uint16_t)(0x3c00 + (rand() & 0x7f));            // scales ~ 0.0078 .. 0.0156
        for (auto &v : hb) v = (uint16_t)(0xbd80 + (rand() & 0x7f));            // biases ~ -0.0625 .. -0.125
        for (auto &v : ha) { const float f = ((rand() & 0xffff) / 32768.0f - 1.0f); uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
        CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(s, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
        if (repack_w4_tiled(w, s, b, wt4, sb4, K, N, 0) != 0) { printf("repack failed\n"); return 1; }
        const size_t wsb = tl_quantized_matmul_workspace_bytes(M, N, K, TL_BF16, 1, 1); if (wsb) CK(hipMalloc(&ws, wsb));
        auto run_old = [&]() { if (tl_quantized_matmul(s, b, a, w, out_ref, M, N, K, 128, 4, TL_BF16, 1, 1, ws, wsb, nullptr) != 0) { printf("matmul failed: %s\n", tl_last_error()); exit(1);} };
        const size_t lds = 2 * 128 * 64 * 2 + (size_t)G * 128 * 4;
        CK(hipFuncSetAttribute((const void *)qmm4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        auto run_new = [&]() {
            hipLaunchKernelGGL(row_group_sums_kernel, dim3(M), dim3(64), 0, 0, a, asum, N);
            hipLaunchKernelGGL(qmm4_kernel, dim3((K + 127) / 128, (M + 127) / 128), dim3(256), lds, 0, wt4, sb4, a, asum, out_new, M, N, K);
        };
        run_old(); run_new(); CK(hipDeviceSynchronize());
        std::vector<uint16_t> r1((size_t)M * K), r2((size_t)M * K);
        CK(hipMemcpy(r1.data(), out_ref, r1.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), out_new, r2.size() * 2, hipMemcpyDeviceToHost));
        double sd = 0, sr = 0, md = 0, mr = 0; size_t nan = 0;
        for (size_t i = 0; i < r1.size(); ++i) {
            uint32_t u1 = (uint32_t)r1[i] << 16, u2 = (uint32_t)r2[i] << 16; float f1, f2; memcpy(&f1, &u1, 4); memcpy(&f2, &u2, 4);
            if (!(f2 == f2)) { ++nan; continue; }
            sd += fabs(f1 - f2); sr += fabs(f1); md = fmax(md, fabs(f1 - f2)); mr = fmax(mr, fabs(f1));
        }
        float ms_old, ms_new; const int iters = 10;
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) run_old(); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms_old, e0, e1));
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) run_new(); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms_new, e0, e1));
        const double flop = 2.0 * M * K * N, uo = ms_old * 1e3 / iters, un = ms_new * 1e3 / iters;
        printf("  %-8s M=%d K=%d N=%d: product %7.1f us %6.1f TFLOP/s | prototype (+ row sums) %7.1f us %6.1f TFLOP/s | mean|diff|/mean|ref| %.4f  max|diff| %.3f (max|ref| %.1f) NaN %zu\n",
               sh.name, M, K, N, uo, flop / uo / 1e6, un, flop / un / 1e6, sd / fmax(sr, 1e-9), md, mr, nan);
        tot_new += un; tot_old += uo; tot_flop += flop;
        CK(hipFree(w)); CK(hipFree(s)); CK(hipFree(b)); CK(hipFree(wt4)); CK(hipFree(sb4)); CK(hipFree(a)); CK(hipFree(out_ref)); CK(hipFree(out_new)); CK(hipFree(asum)); if (ws) CK(hipFree(ws));
    }
    printf("  layer: product %7.1f us %6.1f TFLOP/s | prototype %7.1f us %6.1f TFLOP/s\n", tot_old, tot_flop / tot_old / 1e6, tot_new, tot_flop / tot_new / 1e6);
    return 0;
}
