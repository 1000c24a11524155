// Kernel laboratory (not part of the product): times GEMV variants with in-kernel clocks.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/gemv_lab.hip tiny-llm_amd/csrc/build/*.o -o /tmp/gemv_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <algorithm>
#include "../../tiny-llm_amd/csrc/qmv.h"
#include "../../tiny-llm_amd/csrc/qmv3.h"
#include "../../tiny-llm_amd/csrc/engine_kernels.h"

using namespace tl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ---- H3: pure streaming floor: every lane reads 16 B chunks, xor-reduces, one store per WG ---------
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4 *__restrict__ src, size_t n16, uint32_t *out, prof_t *prof) {
    const prof_t prof_t0 = prof_begin(prof);
    const size_t per_wg = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t begin = blockIdx.x * per_wg;
    const size_t end = min(begin + per_wg, n16);
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = begin + threadIdx.x; i < end; i += 256 * UNROLL) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const size_t j = i + (size_t)u * 256;
            if (j < end) {
                if constexpr (NT) v[u] = __builtin_nontemporal_load(src + j); else v[u] = src[j];
            } else v[u] = u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    }
    uint32_t r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (r == 0x12345678u) out[blockIdx.x] = r;  // practically never
    prof_end(prof, prof_t0);
}

struct Timer {
    prof_t *buf, *pairs; int cap, n = 0; double us_per_tick;
    Timer(int max_wg, int cap_) : cap(cap_) {
        CK(hipMalloc(&buf, (size_t)max_wg * 16)); CK(hipMemset(buf, 0, (size_t)max_wg * 16));
        CK(hipMalloc(&pairs, (size_t)cap * 16));
        int khz; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0)); us_per_tick = 1e3 / khz;
    }
    void after(int n_wg) { hipLaunchKernelGGL(prof_reduce_kernel, dim3(1), dim3(1024), 0, 0, buf, n_wg, pairs + 2 * (size_t)(n++ % cap)); }
    // returns median duration, resets
    double finish(double *mn = nullptr) {
        CK(hipDeviceSynchronize());
        int m = std::min(n, cap); std::vector<prof_t> h(2 * m);
        CK(hipMemcpy(h.data(), pairs, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> d; for (int i = 0; i < m; ++i) d.push_back((double)(h[2*i+1] - h[2*i]) * us_per_tick);
        std::sort(d.begin(), d.end()); n = 0; if (mn) *mn = d[0];
        return d[d.size() / 2];
    }
};

int main(int argc, char **argv) {
    const bool pmc_mode = argc > 1 && std::string(argv[1]) == "pmc";
    const int COPIES = getenv("LAB_COPIES") ? atoi(getenv("LAB_COPIES")) : 40;
    struct Shape { const char *name; int K, N; } shapes[] = {
        {"qkv", 6144, 2560}, {"o", 2560, 4096}, {"gate_up", 19456, 2560}, {"down", 2560, 9728}, {"lm_head", 151936, 2560}};
    Timer T(200000, 4096);
    uint32_t *dummy; CK(hipMalloc(&dummy, 1 << 20));
    for (auto &sh : shapes) {
        const int K = sh.K, N = sh.N, G = N / 128;
        const size_t wwords = (size_t)K * N / 8;
        const size_t wbytes = wwords * 4 + (size_t)K * G * 4;
        const int copies = (int)std::max<size_t>(1, std::min<size_t>(COPIES, ((size_t)700 << 20) / wbytes + 1));
        uint32_t *w; uint16_t *s, *b, *x, *out, *nw, *res;
        CK(hipMalloc(&w, wwords * 4 * copies)); CK(hipMalloc(&s, (size_t)K * G * 2 * copies)); CK(hipMalloc(&b, (size_t)K * G * 2 * copies));
        CK(hipMalloc(&x, N * 2 * 8)); CK(hipMalloc(&out, (size_t)K * 2 * 8)); CK(hipMalloc(&nw, N * 2)); CK(hipMalloc(&res, (size_t)K * 2 * 8));
        // random-ish contents
        std::vector<uint32_t> hw(wwords); for (size_t i = 0; i < wwords; ++i) hw[i] = (uint32_t)rand() * 2654435761u + (uint32_t)i;
        for (int c = 0; c < copies; ++c) CK(hipMemcpy(w + c * wwords, hw.data(), wwords * 4, hipMemcpyHostToDevice));
        std::vector<uint16_t> hs((size_t)K * G); for (auto &v : hs) v = (uint16_t)(0x3c00 + (rand() & 0x7f));
        for (int c = 0; c < copies; ++c) { CK(hipMemcpy(s + (size_t)c * K * G, hs.data(), hs.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(b + (size_t)c * K * G, hs.data(), hs.size() * 2, hipMemcpyHostToDevice)); }
        std::vector<uint16_t> hx(N * 8); for (auto &v : hx) v = (uint16_t)((rand() & 1 ? 0x3f00 : 0xbf00) + (rand() & 0xff)); CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(nw, hx.data(), N * 2, hipMemcpyHostToDevice));
        CK(hipMemset(res, 0, (size_t)K * 2 * 8));
        const int iters = std::max(3 * copies, 60);
        printf("== %s K=%d N=%d  %.2f MB  copies=%d\n", sh.name, K, N, wbytes / 1e6, copies);
        // floor
        for (int variant = 0; variant < 4; ++variant) {
            if (pmc_mode && variant != 2) continue;
            const int grids[2] = {512, 2048};
            for (int gi = 0; gi < 2; ++gi) {
                const int grid = grids[gi];
                for (int i = 0; i < iters; ++i) {
                    const u32x4 *src = (const u32x4 *)(w + (size_t)(i % copies) * wwords);
                    switch (variant) {
                        case 0: hipLaunchKernelGGL((stream_kernel<4, false>), dim3(grid), dim3(256), 0, 0, src, wwords / 4, dummy, T.buf); break;
                        case 1: hipLaunchKernelGGL((stream_kernel<8, false>), dim3(grid), dim3(256), 0, 0, src, wwords / 4, dummy, T.buf); break;
                        case 2: hipLaunchKernelGGL((stream_kernel<4, true>), dim3(grid), dim3(256), 0, 0, src, wwords / 4, dummy, T.buf); break;
                        case 3: hipLaunchKernelGGL((stream_kernel<8, true>), dim3(grid), dim3(256), 0, 0, src, wwords / 4, dummy, T.buf); break;
                    }
                    T.after(grid);
                }
                double mn; double med = T.finish(&mn);
                printf("   stream u=%d nt=%d grid=%4d : med %7.2f us  min %7.2f  -> %7.1f GB/s\n", variant & 1 ? 8 : 4, variant >> 1, grid, med, mn, wwords * 4 / med / 1e3);
            }
        }
        // current GEMV variants
        for (int M : {1, 4}) {
            if (pmc_mode) break;
            struct V { const char *n; int pro, epi; } vs[] = {{"plain", PRO_NONE, EPI_STORE}, {"rms", PRO_RMSNORM, EPI_STORE}, {"resid", PRO_NONE, EPI_RESIDUAL}, {"rms+swiglu", PRO_RMSNORM, EPI_SWIGLU}};
            for (auto &v : vs) {
                const QmvPlan pl = qmv_plan(M, N, K);
                for (int i = 0; i < iters; ++i) {
                    QmvArgs a{}; const int c = i % copies;
                    a.scales = s + (size_t)c * K * G; a.biases = b + (size_t)c * K * G; a.b = w + (size_t)c * wwords; a.a = x; a.out = out;
                    a.norm_w = nw; a.residual = res; a.eps = 1e-6f; a.M = M; a.N = N; a.K = K; a.prof = T.buf;
                    if (launch_qmv_fused_bf16(a, v.pro, v.epi, 0) != 0) { printf("launch failed\n"); break; }
                    T.after(pl.blocks);
                }
                double mn; double med = T.finish(&mn);
                printf("   qmv M=%d %-10s (MR%d WN%d RPL%d blocks %5d): med %7.2f us  min %7.2f -> %7.1f GB/s\n", M, v.n, pl.MR, pl.WN, pl.RPL, pl.blocks, med, mn, wbytes / med / 1e3);
            }
        }
        // qmv3 (tiled layout, MFMA)
        {
            uint32_t *wt3, *sb3; CK(hipMalloc(&wt3, wwords * 4 * copies)); CK(hipMalloc(&sb3, (size_t)K * G * 4 * copies));
            for (int c2 = 0; c2 < copies; ++c2) repack_w4_tiled(w + (size_t)c2 * wwords, s + (size_t)c2 * K * G, b + (size_t)c2 * K * G, wt3 + (size_t)c2 * wwords, sb3 + (size_t)c2 * K * G, K, N, 0);
            CK(hipDeviceSynchronize());
            for (int M : {1, 4, 8}) {
                struct V { const char *n; int pro, epi; } vs[] = {{"plain", PRO_NONE, EPI_STORE}, {"rms", PRO_RMSNORM, EPI_STORE}, {"resid", PRO_NONE, EPI_RESIDUAL}, {"rms+swiglu", PRO_RMSNORM, EPI_SWIGLU}};
                for (auto &v : vs) {
                    const int combos[][2] = {{0, 0}, {1, 4}, {2, 4}, {4, 4}, {2, 8}, {4, 8}, {8, 8}, {16, 16}};
                    for (auto &cb : combos) {
                        const int fks = cb[0], fcw = cb[1];
                        if (pmc_mode && (fks != 0 || M != 1)) continue;
                        if (fks && M != 1) continue;
                        const Qmv3Plan pl = qmv3_plan(M, N, K, fks, fcw);
                        if (fcw && pl.CW != fcw) continue;
                        if (!pl.ok) continue;
                        uint16_t *out2; CK(hipMalloc(&out2, (size_t)K * 2 * 8));
                        QmvArgs a{}; a.scales = s; a.biases = b; a.b = w; a.a = x; a.norm_w = nw; a.residual = res; a.eps = 1e-6f; a.M = M; a.N = N; a.K = K;
                        a.out = out; launch_qmv_fused_bf16(a, v.pro, v.epi, 0);
                        Qmv3Args a3{}; a3.wt = wt3; a3.sbt = sb3; a3.a = x; a3.out = out2; a3.norm_w = nw; a3.residual = res; a3.eps = 1e-6f; a3.M = M; a3.N = N; a3.K = K;
                        if (launch_qmv3_bf16(a3, v.pro, v.epi, 0, fks, fcw) != 0) { printf("qmv3 launch failed\n"); continue; }
                        const size_t ncmp = (size_t)M * (v.epi == EPI_SWIGLU ? K / 2 : K);
                        std::vector<uint16_t> h1(ncmp), h2(ncmp);
                        CK(hipDeviceSynchronize());
                        CK(hipMemcpy(h1.data(), out, ncmp * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), out2, ncmp * 2, hipMemcpyDeviceToHost));
                        double maxd = 0, maxv = 0; size_t bad = 0;
                        for (size_t i = 0; i < ncmp; ++i) { uint32_t u1 = (uint32_t)h1[i] << 16, u2 = (uint32_t)h2[i] << 16; float f1, f2; memcpy(&f1, &u1, 4); memcpy(&f2, &u2, 4);
                            double d = fabs((double)f1 - f2); if (!(d <= 0.02 * fabs(f1) + 1e-2)) ++bad; maxd = std::max(maxd, d); maxv = std::max(maxv, (double)fabs(f1)); }
                        for (int i = 0; i < iters; ++i) {
                            const int cidx = i % copies;
                            a3.wt = wt3 + (size_t)cidx * wwords; a3.sbt = sb3 + (size_t)cidx * K * G; a3.prof = T.buf;
                            launch_qmv3_bf16(a3, v.pro, v.epi, 0, fks, fcw);
                            T.after(pl.blocks);
                        }
                        double mn; double med = T.finish(&mn);
                        printf("   qmv3 M=%d %-10s (MR%d KS%d CW%d LM%2d blocks %5d lds %6zu): med %7.2f us  min %7.2f -> %7.1f GB/s | maxdiff %.4f (max|v| %.2f) bad %zu\n", M, v.n, pl.MR, pl.KS, pl.CW, pl.LM, pl.blocks, pl.lds, med, mn, wbytes / med / 1e3, maxd, maxv, bad);
                        CK(hipFree(out2));
                    }
                }
            }
            CK(hipFree(wt3)); CK(hipFree(sb3));
        }
        CK(hipFree(w)); CK(hipFree(s)); CK(hipFree(b)); CK(hipFree(x)); CK(hipFree(out)); CK(hipFree(nw)); CK(hipFree(res));
    }
    return 0;
}
