// Kernel laboratory (not part of the product): the plain bf16 GEMM of csrc/gemm8.h -- C[M, N] = A[M, K] . W[N, K]^T, 256 x 256 tiles on 8 waves,
// both operand tiles by LDS-DMA -- at the Qwen3-4B prefill shapes, HIP events, checked against a naive fp32 kernel on sampled outputs.
// usage: gemm8_lab <rows> [epi: 0 store, 1 residual, 2 swiglu]     build + run: tools/lab/run_gemm8_lab.sh
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../tiny-llm_amd/csrc/gemm8.h"
using namespace tl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
namespace tl { int fail(int c, const std::string &) { return c; } void set_error(const std::string &) {} }

static uint32_t rng_state = 12345u;
static inline uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state; }
static inline uint16_t bf16_of(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static inline float f_of(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void ref_kernel(const uint16_t *A, const uint16_t *W, float *out, const int *rows, const int *cols, int n, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint16_t *a = A + (size_t)rows[i] * K, *w = W + (size_t)cols[i] * K;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += (double)__uint_as_float((uint32_t)a[k] << 16) * (double)__uint_as_float((uint32_t)w[k] << 16);
    out[i] = (float)s;
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 2048;
    const int EPI_T = argc > 2 ? atoi(argv[2]) : 0;  // the TIMED launches' epilogue (the check runs on EPI_STORE)
    struct Shape { const char *name; int N, K; } shapes[] = {{"qkv", 6144, 2560}, {"o", 2560, 4096}, {"gate_up", 19456, 2560}, {"down", 2560, 9728}, {"square", 8192, 8192}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double layer_us = 0, layer_flop = 0;
    int n_shapes = 5;
    if (argc > 4) { shapes[0] = Shape{"custom", atoi(argv[3]), atoi(argv[4])}; n_shapes = 1; }  // gemm8_lab <rows> <epi> <N> <K>
    for (int si = 0; si < n_shapes; ++si) {
        auto &sh = shapes[si];
        const int N = sh.N, K = sh.K, Mx = strcmp(sh.name, "square") == 0 ? 8192 : M;
        uint16_t *A, *W, *C;
        CK(hipMalloc(&A, (size_t)Mx * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)Mx * N * 2));
        {
            std::vector<uint16_t> ha((size_t)Mx * K), hw((size_t)N * K);
            for (auto &x : ha) x = bf16_of(((int)(rnd() >> 8) - 8388608) * (1.0f / 8388608.f));
            for (auto &x : hw) x = bf16_of(((int)(rnd() >> 8) - 8388608) * (0.05f / 8388608.f));
            CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        }
        CK(hipMemset(C, 0xff, (size_t)Mx * N * 2));
        Gemm8Args g{}; g.a = A; g.w = W; g.out = C; g.M = Mx; g.N = N; g.K = K;
        if (launch_gemm8_bf16(g, EPI_STORE, 0) != 0) { printf("%s: launch failed\n", sh.name); continue; }
        CK(hipDeviceSynchronize());
        // sampled check
        const int ns = 4096;
        std::vector<int> hr(ns), hc(ns);
        for (int i = 0; i < ns; ++i) { hr[i] = rnd() % Mx; hc[i] = rnd() % N; }
        hr[0] = 0, hc[0] = 0; hr[1] = Mx - 1, hc[1] = N - 1; hr[2] = 255, hc[2] = 256; hr[3] = 256, hc[3] = 255;
        int *dr, *dc; float *dref; CK(hipMalloc(&dr, ns * 4)); CK(hipMalloc(&dc, ns * 4)); CK(hipMalloc(&dref, ns * 4));
        CK(hipMemcpy(dr, hr.data(), ns * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, hc.data(), ns * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(ref_kernel, dim3((ns + 255) / 256), dim3(256), 0, 0, A, W, dref, dr, dc, ns, K);
        std::vector<float> href(ns); CK(hipMemcpy(href.data(), dref, ns * 4, hipMemcpyDeviceToHost));
        std::vector<uint16_t> hcout((size_t)Mx * N); CK(hipMemcpy(hcout.data(), C, hcout.size() * 2, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0; int bad = 0;
        for (int i = 0; i < ns; ++i) {
            const float got = f_of(hcout[(size_t)hr[i] * N + hc[i]]);
            const double err = fabs((double)got - href[i]), tol = fabs(href[i]) * (1.0 / 128.0) + 1e-3;
            worst = std::max(worst, err); scale = std::max(scale, (double)fabs(href[i]));
            if (!(err <= tol)) { if (bad < 4) printf("   mismatch at (%d, %d): got %g want %g\n", hr[i], hc[i], got, href[i]); ++bad; }
        }
        const int iters = 20;
        for (int i = 0; i < 3; ++i) launch_gemm8_bf16(g, EPI_STORE, 0);
        CK(hipDeviceSynchronize());
        float ms;
        uint16_t *R = nullptr;
        if (EPI_T == EPI_RESIDUAL) { CK(hipMalloc(&R, (size_t)Mx * N * 2)); CK(hipMemset(R, 0, (size_t)Mx * N * 2)); g.residual = R; }
        for (int i = 0; i < 3; ++i) launch_gemm8_bf16(g, EPI_T, 0);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) launch_gemm8_bf16(g, EPI_T, 0); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        if (R) CK(hipFree(R));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / iters, flop = 2.0 * Mx * N * K;
        { const Gemm8Plan pl = gemm8_plan(Mx, N, K); printf("[%dx%d, %d tiles] ", pl.BM, pl.BN, pl.tiles); }
        printf("%-8s M=%d N=%d K=%d: %8.1f us  %7.1f TFLOP/s   sampled outputs off: %d of %d (max err %.4g at scale %.3g)\n", sh.name, Mx, N, K, us, flop / us / 1e6, bad, ns, worst, scale);
        if (strcmp(sh.name, "square") != 0) layer_us += us, layer_flop += flop;
#ifdef G8_TRACE  // build gemm8.hip and this file with -DG8_TRACE=<workgroup> (per-step stamps of that workgroup) or -DG8_TRACE=-1 (start / end of every workgroup)
        {
            launch_gemm8_bf16(g, EPI_T, 0); CK(hipDeviceSynchronize());
            std::vector<unsigned long long> tr(2048); gemm8_trace_read(tr.data(), 2048);
            int rate = 0; CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));
            if ((G8_TRACE) < 0) {
                unsigned long long t0 = ~0ull; for (int b = 0; b < 256; ++b) t0 = std::min(t0, tr[2 * b]);
                printf("   start / end of every workgroup, us since the first start:\n");
                for (int b = 0; b < 256; ++b) printf("%s%d:%.1f-%.1f", b % 8 ? "  " : "\n   ", b, (double)(tr[2 * b] - t0) * 1000.0 / rate, (double)(tr[2 * b + 1] - t0) * 1000.0 / rate);
                printf("\n");
                continue;
            }
            const int n = (int)tr[0];
            printf("   trace of workgroup %d (%d stamps; us since the first; code 2 = behind a step's barrier, 3 / 4 = epilogue begins / ends):\n   ", (int)(G8_TRACE), n);
            unsigned long long t0 = tr[1] >> 8, prev = t0;
            for (int i = 0; i < n; ++i) { const unsigned long long t = tr[1 + i] >> 8; const int code = (int)(tr[1 + i] & 0xff);
                if (code != 2) printf("\n   [%d @%.2f] ", code, (double)(t - t0) * 1000.0 / rate); else printf("%.2f ", (double)(t - prev) * 1000.0 / rate); prev = t; }
            printf("\n");
        }
#endif
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(dr)); CK(hipFree(dc)); CK(hipFree(dref));
    }
    printf("layer (qkv + o + gate_up + down) at %d rows: %.1f us  %.1f TFLOP/s\n", M, layer_us, layer_flop / layer_us / 1e6);
    return 0;
}
