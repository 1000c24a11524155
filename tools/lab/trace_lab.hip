// Phase trace of the decode GEMV (lab only): wall-clock stamps of wave 0 of every workgroup at the kernel's phase boundaries
// (qmv3.h, QMV3_TRACE), for the four projections as the engine launches them at one row, rotating weight copies (cold HBM).
// Printed per projection: mean over workgroups and launches of the time since the FIRST workgroup's start.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 -DQMV3_TRACE tools/lab/trace_lab.hip -o tools/lab/trace_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../tiny-llm_amd/csrc/qmv3.h"
using namespace tl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
namespace tl { int fail(int c, const std::string &) { return c; } void set_error(const std::string &) {} }

template <int MR, int KS, int CW, int PRO, int EPI, int LM>
void run(const char *name, int K, int N, bool give_ss) {
    const int G = N / 128; const size_t wwords = (size_t)K * N / 8; const size_t wbytes = wwords * 4 + (size_t)K * G * 4;
    const int copies = (int)std::max<size_t>(2, std::min<size_t>(40, ((size_t)700 << 20) / wbytes + 1));
    uint32_t *w, *sb; uint16_t *x, *out, *nw, *res; float *ss, *sso; unsigned long long *trace;
    CK(hipMalloc(&w, wwords * 4 * copies)); CK(hipMalloc(&sb, (size_t)K * G * 4 * copies));
    CK(hipMalloc(&x, N * 2 * 8)); CK(hipMalloc(&out, (size_t)K * 2 * 8)); CK(hipMalloc(&nw, N * 2)); CK(hipMalloc(&res, (size_t)K * 2 * 8));
    CK(hipMalloc(&ss, 4096 * 4)); CK(hipMalloc(&sso, 4096 * 4 * 8));
    CK(hipMemset(w, 0x5a, wwords * 4 * copies)); CK(hipMemset(sb, 0x3c, (size_t)K * G * 4 * copies));
    CK(hipMemset(x, 0x3f, N * 16)); CK(hipMemset(nw, 0x3f, N * 2)); CK(hipMemset(res, 0, (size_t)K * 16)); CK(hipMemset(ss, 0x3f, 4096 * 4));
    const int WR = CW / KS; const int blocks = (K / 16 + WR - 1) / WR;
    const int iters = 60;
    CK(hipMalloc(&trace, (size_t)iters * blocks * 64)); CK(hipMemset(trace, 0, (size_t)iters * blocks * 64));
    int khz; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    const size_t lds = qmv3_lds_bytes(MR, N, KS, CW);
    for (int it = 0; it < iters; ++it) {
        Qmv3Args a{}; a.wt = w + (size_t)(it % copies) * wwords; a.sbt = sb + (size_t)(it % copies) * K * G; a.a = x; a.out = out; a.norm_w = nw; a.residual = res;
        a.eps = 1e-6f; a.M = 1; a.N = N; a.K = K; a.trace = trace + (size_t)it * blocks * 8;
        if (give_ss && (PRO == PRO_RMSNORM || PRO == PRO_RMS_WEIGHTED)) { a.ss_in = ss; a.ss_n = N / 16; }
        if (EPI == EPI_RESIDUAL && give_ss) { a.norm_out = nw; a.out_w = res; }
        if (EPI == EPI_RESIDUAL) a.ss_out = sso;
        hipLaunchKernelGGL((qmv3_kernel<MR, KS, CW, PRO, EPI, LM>), dim3(blocks), dim3(CW * 64), lds, 0, a);
    }
    CK(hipDeviceSynchronize());
    float ms_launch = 0.f;
    {   // the same launches without stamps, back to back: event time per launch (boundary included)
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        std::vector<float> all;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int it = 0; it < 200; ++it) {
                Qmv3Args a{}; a.wt = w + (size_t)(it % copies) * wwords; a.sbt = sb + (size_t)(it % copies) * K * G; a.a = x; a.out = out; a.norm_w = nw; a.residual = res;
                a.eps = 1e-6f; a.M = 1; a.N = N; a.K = K; a.trace = nullptr;
                if (give_ss && (PRO == PRO_RMSNORM || PRO == PRO_RMS_WEIGHTED)) { a.ss_in = ss; a.ss_n = N / 16; }
                if (EPI == EPI_RESIDUAL) a.ss_out = sso;
                if (EPI == EPI_RESIDUAL && give_ss) { a.norm_out = nw; a.out_w = res; }
                hipLaunchKernelGGL((qmv3_kernel<MR, KS, CW, PRO, EPI, LM>), dim3(blocks), dim3(CW * 64), lds, 0, a);
            }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float m; CK(hipEventElapsedTime(&m, e0, e1)); all.push_back(m * 1e3f / 200);
        }
        std::sort(all.begin(), all.end()); ms_launch = all[3];
    }
    std::vector<unsigned long long> h((size_t)iters * blocks * 8); CK(hipMemcpy(h.data(), trace, h.size() * 8, hipMemcpyDeviceToHost));
    double mean[6] = {0}, mx[6] = {0}; long n = 0;
    for (int it = 10; it < iters; ++it) {
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < blocks; ++b) t0 = std::min(t0, h[((size_t)it * blocks + b) * 8]);
        double m2[6] = {0};
        for (int b = 0; b < blocks; ++b) for (int i = 0; i < 6; ++i) { const double v = (double)(h[((size_t)it * blocks + b) * 8 + i] - t0) * 1e3 / khz; mean[i] += v; m2[i] = std::max(m2[i], v); }
        for (int i = 0; i < 6; ++i) mx[i] += m2[i];
        n += blocks;
    }
    printf("%-22s blocks %5d %6.2f MB | mean us since first start: start %.2f | x loads out %.2f | w loads out %.2f | staged %.2f | mfma done %.2f | end %.2f || slowest workgroup: staged %.2f mfma done %.2f end %.2f || %.2f us per launch back to back (no stamps)\n",
           name, blocks, wbytes / 1e6, mean[0] / n, mean[1] / n, mean[2] / n, mean[3] / n, mean[4] / n, mean[5] / n, mx[3] / (iters - 10), mx[4] / (iters - 10), mx[5] / (iters - 10), ms_launch);
    CK(hipFree(w)); CK(hipFree(sb)); CK(hipFree(x)); CK(hipFree(out)); CK(hipFree(nw)); CK(hipFree(res)); CK(hipFree(ss)); CK(hipFree(sso)); CK(hipFree(trace));
}
int main() {
    run<1, 2, 4, PRO_RMSNORM, EPI_STORE, 10>("qkv rms (ss given)", 6144, 2560, true);
    run<1, 2, 4, PRO_RMSNORM, EPI_STORE, 10>("qkv rms (own sum)", 6144, 2560, false);
    run<1, 2, 4, PRO_NONE, EPI_STORE, 10>("qkv plain", 6144, 2560, false);
    run<1, 2, 4, PRO_RMS_WEIGHTED, EPI_STORE, 10>("qkv weighted rows", 6144, 2560, true);
    run<1, 4, 4, PRO_NONE, EPI_RESIDUAL, 8>("wo resid", 2560, 4096, false);
    run<1, 4, 4, PRO_NONE, EPI_RESIDUAL, 8>("wo resid + out_w", 2560, 4096, true);
    run<1, 4, 4, PRO_RMSNORM, EPI_SWIGLU, 5>("gate_up rms (ss given)", 19456, 2560, true);
    run<1, 4, 4, PRO_RMS_WEIGHTED, EPI_SWIGLU, 5>("gate_up weighted rows", 19456, 2560, true);
    run<1, 4, 4, PRO_NONE, EPI_STORE, 5>("gate_up plain", 19456, 2560, false);
    run<1, 8, 8, PRO_NONE, EPI_RESIDUAL, 10>("down resid", 2560, 9728, false);
    run<1, 8, 8, PRO_NONE, EPI_RESIDUAL, 10>("down resid + out_w", 2560, 9728, true);
    run<1, 2, 4, PRO_RMSNORM, EPI_STORE, 10>("lm_head rms (ss given)", 151936, 2560, true);
    run<1, 2, 4, PRO_RMS_WEIGHTED, EPI_STORE, 10>("lm_head weighted rows", 151936, 2560, true);
    return 0;
}
