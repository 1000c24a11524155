// Plan sweep of the decode GEMV (lab only): the instantiated (reduction split KS, waves CW, groups per wave LM) choices for 1 / 2 / 4
// activation rows at the Qwen3-4B projections, rotating weight copies, time per launch back to back (HIP events).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 tools/lab/plan_lab.hip -o tools/lab/plan_lab
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
void run(const char *name, int K, int N) {
    const int G = N / 128; const size_t wwords = (size_t)K * N / 8; const size_t wbytes = wwords * 4 + (size_t)K * G * 4;
    const int copies = (int)std::max<size_t>(2, std::min<size_t>(40, ((size_t)700 << 20) / wbytes + 1));
    static uint32_t *w = nullptr, *sb = nullptr; static uint16_t *x, *out, *nw, *res; static float *ss, *sso;
    if (!w) {
        CK(hipMalloc(&w, (size_t)800 << 20)); CK(hipMalloc(&sb, (size_t)64 << 20)); CK(hipMalloc(&x, 9728 * 2 * 8)); CK(hipMalloc(&out, 19456 * 2 * 8));
        CK(hipMalloc(&nw, 9728 * 2)); CK(hipMalloc(&res, 19456 * 2 * 8)); CK(hipMalloc(&ss, 4096 * 4 * 8)); CK(hipMalloc(&sso, 4096 * 4 * 8));
        CK(hipMemset(w, 0x5a, (size_t)800 << 20)); CK(hipMemset(sb, 0x3c, (size_t)64 << 20)); CK(hipMemset(x, 0x3f, 9728 * 16)); CK(hipMemset(nw, 0x3f, 9728 * 2));
        CK(hipMemset(res, 0, 19456 * 16)); CK(hipMemset(ss, 0x3f, 4096 * 32));
    }
    const int WR = CW / KS; const int blocks = (K / 16 + WR - 1) / WR;
    const size_t lds = qmv3_lds_bytes(MR, N, KS, CW);
    auto kern = qmv3_kernel<MR, KS, CW, PRO, EPI, LM>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> all;
    for (int rep = 0; rep < 7; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int it = 0; it < 200; ++it) {
            Qmv3Args a{}; a.wt = w + (size_t)(it % copies) * wwords; a.sbt = sb + (size_t)(it % copies) * K * G; a.a = x; a.out = out; a.norm_w = nw; a.residual = res;
            a.eps = 1e-6f; a.M = MR; a.N = N; a.K = K;
            if (PRO == PRO_RMSNORM || PRO == PRO_RMS_WEIGHTED) { a.ss_in = ss; a.ss_n = N / 16; }
            if (EPI == EPI_RESIDUAL) { a.ss_out = sso; a.norm_out = nw; a.out_w = res + 19456 * 4; }
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(CW * 64), lds, 0, a);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float m; CK(hipEventElapsedTime(&m, e0, e1)); all.push_back(m * 1e3f / 200);
    }
    std::sort(all.begin(), all.end());
    printf("%-10s rows %d  KS%d CW%d LM%-2d blocks %5d lds %6zu: %6.2f us per launch\n", name, MR, KS, CW, LM, blocks, lds, all[3]);
}
template <int MR> void sweep() {
    run<MR, 2, 4, PRO_RMSNORM, EPI_STORE, 10>("qkv", 6144, 2560);
    run<MR, 2, 8, PRO_RMSNORM, EPI_STORE, 10>("qkv", 6144, 2560);
    run<MR, 4, 4, PRO_RMSNORM, EPI_STORE, 5>("qkv", 6144, 2560);
    run<MR, 4, 8, PRO_RMSNORM, EPI_STORE, 5>("qkv", 6144, 2560);
    run<MR, 8, 8, PRO_RMSNORM, EPI_STORE, 4>("qkv", 6144, 2560);
    run<MR, 4, 4, PRO_NONE, EPI_RESIDUAL, 8>("wo", 2560, 4096);
    run<MR, 4, 8, PRO_NONE, EPI_RESIDUAL, 8>("wo", 2560, 4096);
    run<MR, 8, 8, PRO_NONE, EPI_RESIDUAL, 4>("wo", 2560, 4096);
    run<MR, 2, 4, PRO_RMS_WEIGHTED, EPI_SWIGLU, 10>("gate_up w", 19456, 2560);
    run<MR, 2, 8, PRO_RMS_WEIGHTED, EPI_SWIGLU, 10>("gate_up w", 19456, 2560);
    run<MR, 4, 4, PRO_RMS_WEIGHTED, EPI_SWIGLU, 5>("gate_up w", 19456, 2560);
    run<MR, 4, 8, PRO_RMS_WEIGHTED, EPI_SWIGLU, 5>("gate_up w", 19456, 2560);
    run<MR, 8, 8, PRO_RMS_WEIGHTED, EPI_SWIGLU, 4>("gate_up w", 19456, 2560);
    run<MR, 4, 4, PRO_RMSNORM, EPI_SWIGLU, 5>("gate_up rms", 19456, 2560);
    run<MR, 2, 4, PRO_RMSNORM, EPI_SWIGLU, 10>("gate_up rms", 19456, 2560);
    run<MR, 8, 8, PRO_NONE, EPI_RESIDUAL, 10>("down", 2560, 9728);
}
int main(int argc, char **argv) {
    if (argc > 1 && atoi(argv[1]) == 8) {  // eight rows on the GEMV (the engine hands 5+ rows to the K-sliced skinny matmul instead)
        run<8, 2, 4, PRO_RMSNORM, EPI_STORE, 10>("qkv", 6144, 2560);
        run<8, 4, 4, PRO_RMSNORM, EPI_STORE, 5>("qkv", 6144, 2560);
        run<8, 2, 8, PRO_RMSNORM, EPI_STORE, 10>("qkv", 6144, 2560);
        run<8, 4, 8, PRO_RMSNORM, EPI_STORE, 5>("qkv", 6144, 2560);
        run<8, 4, 4, PRO_NONE, EPI_RESIDUAL, 8>("wo", 2560, 4096);
        run<8, 8, 8, PRO_NONE, EPI_RESIDUAL, 4>("wo", 2560, 4096);
        run<8, 4, 8, PRO_NONE, EPI_RESIDUAL, 8>("wo", 2560, 4096);
        run<8, 4, 4, PRO_RMS_WEIGHTED, EPI_SWIGLU, 5>("gate_up w", 19456, 2560);
        run<8, 2, 4, PRO_RMS_WEIGHTED, EPI_SWIGLU, 10>("gate_up w", 19456, 2560);
        run<8, 4, 8, PRO_RMS_WEIGHTED, EPI_SWIGLU, 5>("gate_up w", 19456, 2560);
        run<8, 2, 8, PRO_RMS_WEIGHTED, EPI_SWIGLU, 10>("gate_up w", 19456, 2560);
        run<8, 8, 8, PRO_RMS_WEIGHTED, EPI_SWIGLU, 4>("gate_up w", 19456, 2560);
        run<8, 8, 8, PRO_NONE, EPI_RESIDUAL, 10>("down", 2560, 9728);
        return 0;
    }
    if (argc > 1) {  // the 16-wave question: w_down cut 16 ways along the reduction (5 groups per wave) against the planner's 8 x 10
        run<1, 8, 8, PRO_NONE, EPI_RESIDUAL, 10>("down", 2560, 9728);
        run<1, 16, 16, PRO_NONE, EPI_RESIDUAL, 5>("down", 2560, 9728);
        run<2, 8, 8, PRO_NONE, EPI_RESIDUAL, 10>("down", 2560, 9728);
        run<2, 16, 16, PRO_NONE, EPI_RESIDUAL, 5>("down", 2560, 9728);
        run<4, 8, 8, PRO_NONE, EPI_RESIDUAL, 10>("down", 2560, 9728);
        run<4, 16, 16, PRO_NONE, EPI_RESIDUAL, 5>("down", 2560, 9728);
        run<1, 4, 4, PRO_NONE, EPI_RESIDUAL, 8>("wo", 2560, 4096);
        run<1, 8, 8, PRO_NONE, EPI_RESIDUAL, 4>("wo", 2560, 4096);
        run<1, 16, 16, PRO_NONE, EPI_RESIDUAL, 4>("wo", 2560, 4096);
        run<1, 8, 8, PRO_NONE, EPI_RESIDUAL, 10>("down", 2560, 9728);
        run<1, 16, 16, PRO_NONE, EPI_RESIDUAL, 5>("down", 2560, 9728);
        return 0;
    }
    sweep<1>(); sweep<2>(); sweep<4>(); return 0;
}
