// Ablation lab for qmv3 (QMV3_LAB): which phase costs what?
#define QMV3_LAB 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../tiny-llm_amd/csrc/qmv3.h"
#include "../../tiny-llm_amd/csrc/engine_kernels.h"
using namespace tl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
namespace tl { int fail(int c, const std::string &) { return c; } void set_error(const std::string &) {} }

template <int MR, int KS, int CW, int PRO, int EPI, int LM>
void run(const char *name, int K, int N, int M) {
    const int G = N / 128; const size_t wwords = (size_t)K * N / 8; const size_t wbytes = wwords * 4 + (size_t)K * G * 4;
    const int copies = (int)std::max<size_t>(2, std::min<size_t>(40, ((size_t)700 << 20) / wbytes + 1));
    uint32_t *w, *sb; uint16_t *x, *out, *nw, *res; prof_t *buf, *pairs;
    CK(hipMalloc(&w, wwords * 4 * copies)); CK(hipMalloc(&sb, (size_t)K * G * 4 * copies));
    CK(hipMalloc(&x, N * 2 * 8)); CK(hipMalloc(&out, (size_t)K * 2 * 8)); CK(hipMalloc(&nw, N * 2)); CK(hipMalloc(&res, (size_t)K * 2 * 8));
    CK(hipMemset(w, 0x5a, wwords * 4 * copies)); CK(hipMemset(sb, 0x3c, (size_t)K * G * 4 * copies));
    CK(hipMemset(x, 0x3f, N * 16)); CK(hipMemset(nw, 0x3f, N * 2)); CK(hipMemset(res, 0, (size_t)K * 16));
    const int WR = CW / KS; const int blocks = (K / 16 + WR - 1) / WR;
    CK(hipMalloc(&buf, (size_t)(blocks + 8) * 16)); CK(hipMemset(buf, 0, (size_t)(blocks + 8) * 16)); CK(hipMalloc(&pairs, 4096 * 16));
    int khz; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    const size_t lds = qmv3_lds_bytes(MR, N, KS, CW);
    printf("%-26s blocks %5d %.2f MB:", name, blocks, wbytes / 1e6);
    for (int abl : {0, 8, 1, 2, 4}) {
        const int iters = std::max(3 * copies, 60);
        for (int it = 0; it < iters; ++it) {
            Qmv3Args a{}; a.wt = w + (size_t)(it % copies) * wwords; a.sbt = sb + (size_t)(it % copies) * K * G; a.a = x; a.out = out; a.norm_w = nw; a.residual = res; a.eps = 1e-6f; a.M = M; a.N = N; a.K = K; a.prof = buf; a.ablate = abl;
            hipLaunchKernelGGL((qmv3_kernel<MR, KS, CW, PRO, EPI, LM>), dim3(blocks), dim3(CW * 64), lds, 0, a);
            hipLaunchKernelGGL(prof_reduce_kernel, dim3(1), dim3(1024), 0, 0, buf, blocks, pairs + 2 * (size_t)it);
        }
        CK(hipDeviceSynchronize());
        std::vector<prof_t> h(2 * iters); CK(hipMemcpy(h.data(), pairs, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> d; for (int i = 0; i < iters; ++i) d.push_back((double)(h[2 * i + 1] - h[2 * i]) * 1e3 / khz);
        std::sort(d.begin(), d.end());
        printf("  abl%d %.2f", abl, d[d.size() / 2]);
    }
    printf("  (us; 8=inverse RMS given 1=no MFMA math 2=no staging arithmetic 4=no activation loads)\n");
    CK(hipFree(w)); CK(hipFree(sb)); CK(hipFree(x)); CK(hipFree(out)); CK(hipFree(nw)); CK(hipFree(res)); CK(hipFree(buf)); CK(hipFree(pairs));
}
int main() {
    run<1, 4, 4, PRO_NONE, EPI_RESIDUAL, 8>("o KS4 resid", 2560, 4096, 1);
    run<1, 8, 8, PRO_NONE, EPI_RESIDUAL, 10>("down KS8 resid", 2560, 9728, 1);
    run<1, 2, 4, PRO_RMSNORM, EPI_STORE, 10>("qkv KS2 rms", 6144, 2560, 1);
    run<1, 2, 4, PRO_NONE, EPI_STORE, 10>("qkv KS2 plain", 6144, 2560, 1);
    run<1, 4, 4, PRO_RMSNORM, EPI_SWIGLU, 5>("gate_up KS4 rms+swiglu", 19456, 2560, 1);
    run<1, 4, 4, PRO_NONE, EPI_STORE, 5>("gate_up KS4 plain", 19456, 2560, 1);
    run<1, 2, 4, PRO_RMSNORM, EPI_SWIGLU, 10>("gate_up KS2 rms+swiglu", 19456, 2560, 1);
    run<1, 2, 4, PRO_RMSNORM, EPI_STORE, 10>("lm_head KS2 rms", 151936, 2560, 1);
    return 0;
}
