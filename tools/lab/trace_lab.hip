// Phase-trace lab for qmv2 (build with -DQMV2_TRACE): prints per-phase wall-clock deltas of one workgroup.
#define QMV2_TRACE 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../tiny-llm_amd/csrc/qmv2.h"
using namespace tl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
namespace tl { int fail(int c, const std::string &) { return c; } void set_error(const std::string &) {} }

template <int MR, int KS, int WAVES, int PRO, int EPI>
void run(const char *name, int K, int N, int M) {
    const int G = N / 128; const size_t wwords = (size_t)K * N / 8;
    const int copies = 8;
    uint32_t *w; uint16_t *s, *b, *x, *out, *nw, *res; prof_t *trace;
    CK(hipMalloc(&w, wwords * 4 * copies)); CK(hipMalloc(&s, (size_t)K * G * 2)); CK(hipMalloc(&b, (size_t)K * G * 2));
    CK(hipMalloc(&x, N * 2 * 8)); CK(hipMalloc(&out, (size_t)K * 2 * 8)); CK(hipMalloc(&nw, N * 2)); CK(hipMalloc(&res, (size_t)K * 2 * 8));
    CK(hipMalloc(&trace, 64 * 8)); CK(hipMemset(w, 0x5a, wwords * 4 * copies)); CK(hipMemset(s, 0x3c, (size_t)K * G * 2)); CK(hipMemset(b, 0x3c, (size_t)K * G * 2));
    CK(hipMemset(x, 0x3f, N * 16)); CK(hipMemset(nw, 0x3f, N * 2)); CK(hipMemset(res, 0, (size_t)K * 16));
    const Qmv2Lds L = qmv2_lds(MR, N, KS, WAVES, PRO == PRO_RMSNORM);
    const int WR = WAVES / KS; const int blocks = (K / 16 + WR - 1) / WR;
    int khz; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    std::vector<double> sum(8, 0.0); int reps = 0;
    for (int it = 0; it < 24; ++it) {
        CK(hipMemset(trace, 0, 64 * 8));
        QmvArgs a{}; a.scales = s; a.biases = b; a.b = w + (size_t)(it % copies) * wwords; a.a = x; a.out = out; a.norm_w = nw; a.residual = res; a.eps = 1e-6f; a.M = M; a.N = N; a.K = K; a.trace = trace;
        hipLaunchKernelGGL((qmv2_kernel<MR, KS, WAVES, PRO, EPI>), dim3(blocks), dim3(WAVES * 64), L.total, 0, a);
        CK(hipDeviceSynchronize());
        prof_t h[8]; CK(hipMemcpy(h, trace, 64, hipMemcpyDeviceToHost));
        if (it >= 8) { for (int i = 1; i < 8; ++i) sum[i] += (double)(h[i] - h[i - 1]) * 1e3 / khz; ++reps; }
    }
    printf("%-28s blocks %5d: issue-small %.2f | issue-w %.2f | stage-sb %.2f | stage-x(+wait x, syncs) %.2f | mfma(+wait w) %.2f | reduce %.2f | epilogue %.2f  = %.2f us\n",
           name, blocks, sum[1] / reps, sum[2] / reps, sum[3] / reps, sum[4] / reps, sum[5] / reps, sum[6] / reps, sum[7] / reps,
           (sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6] + sum[7]) / reps);
    CK(hipFree(w)); CK(hipFree(s)); CK(hipFree(b)); CK(hipFree(x)); CK(hipFree(out)); CK(hipFree(nw)); CK(hipFree(res)); CK(hipFree(trace));
}
int main() {
    run<1, 4, 4, PRO_NONE, EPI_RESIDUAL>("o KS4 resid", 2560, 4096, 1);
    run<1, 8, 8, PRO_NONE, EPI_RESIDUAL>("down KS8 resid", 2560, 9728, 1);
    run<1, 2, 4, PRO_RMSNORM, EPI_STORE>("qkv KS2 rms", 6144, 2560, 1);
    run<1, 2, 4, PRO_RMSNORM, EPI_SWIGLU>("gate_up KS2 rms+swiglu", 19456, 2560, 1);
    run<1, 2, 4, PRO_NONE, EPI_STORE>("gate_up KS2 plain", 19456, 2560, 1);
    run<1, 2, 4, PRO_RMSNORM, EPI_STORE>("lm_head KS2 rms", 151936, 2560, 1);
    return 0;
}
