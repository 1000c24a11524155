// Kernel laboratory (not part of the product), round 5: what bounds "every CU reads the same rows"?
//
// The register-resident batched matmul (csrc/qmm6.h) has one workgroup per CU pull ALL activation rows (320 KB at 64 rows x 2,560
// columns) before its tile loop; round 4 measured 8 us for that at 64 rows -- 40 GB/s per CU, far below what a CU's L1 can take from
// its L2 (64 B / clk).  All 256 workgroups request the same lines in the same order at the same time: every request of a moment goes to
// ONE L2 channel.  This lab reads S bytes per workgroup (4 waves x 16-byte lane loads, all of a wave's loads in flight together, as
// qmm6's FRAG path) from one shared buffer
//   mode 0: every workgroup in the same order (qmm6 today)
//   mode 1: workgroup w starts at piece (w * stride) mod pieces and wraps (a rotation: at any moment the workgroups of an XCD are spread
//           over the buffer, i.e. over the L2 channels)
//   mode 2: each workgroup reads its OWN private copy (no sharing at all: the HBM / L2-capacity bound, for scale)
// and prints the time per launch (best of several, in-kernel wall clock of the slowest workgroup).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/l2bcast_lab.hip -o tools/lab/l2bcast_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int NL = 20;  // 1-KiB loads per wave and pass (20 KiB per wave in flight)

__global__ __launch_bounds__(256) void bcast_kernel(const u32x4 *__restrict__ src, size_t copy_units, int pieces, int mode, int stride, int passes,
                                                    uint32_t *sink, u64 *stamp) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const u64 t0 = wall_clock64();
    // a piece = 1 KiB (64 lanes x 16 bytes); the workgroup's 4 waves take pieces p, p + 1, p + 2, p + 3 of every group of four
    const u32x4 *base = src + (mode == 2 ? (size_t)blockIdx.x * copy_units : 0);
    const int rot = mode == 1 ? (int)(((size_t)blockIdx.x * (size_t)stride) % (size_t)pieces) : 0;
    uint32_t x = 0;
    for (int ps = 0; ps < passes; ++ps) {
        u32x4 v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            int piece = (ps * NL + i) * 4 + wave + rot;
            piece = piece >= pieces ? piece - pieces : piece;
            piece = piece >= pieces ? piece - pieces : piece;
            v[i] = base[(size_t)piece * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) x ^= v[i][0] ^ v[i][1] ^ v[i][2] ^ v[i][3];
    }
    if (x == 0x9e3779b9u) sink[0] = x;
    __syncthreads();
    if (tid == 0) stamp[2 * blockIdx.x] = t0, stamp[2 * blockIdx.x + 1] = wall_clock64();
}

int main() {
    int khz; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    const double us = 1e3 / khz;
    const int WG = 256;
    for (int kb : {80, 160, 320}) {
        const int pieces = kb;                      // 1-KiB pieces per workgroup
        const int passes = pieces / (4 * NL);       // 80 pieces per pass over the 4 waves
        if (passes < 1 || passes * 4 * NL != pieces) { if (kb != 40) continue; }
        const int use_passes = std::max(1, pieces / (4 * NL));
        const int use_pieces = use_passes * 4 * NL;
        const size_t copy_units = (size_t)use_pieces * 64;
        u32x4 *buf; CK(hipMalloc(&buf, copy_units * 16 * WG)); CK(hipMemset(buf, 1, copy_units * 16 * WG));
        uint32_t *sink; CK(hipMalloc(&sink, 64));
        u64 *stamp; CK(hipMalloc(&stamp, WG * 2 * 8));
        std::vector<u64> hs(WG * 2);
        for (int mode = 0; mode < 3; ++mode)
        for (int stride : {0, 3, 10, 37}) {
            if ((mode != 1) != (stride == 0)) continue;
            double best = 1e9;
            for (int rep = 0; rep < 12; ++rep) {
                // between repetitions something else streams through L2 so that the buffer is NOT resident from the last run? No: in the
                // product the rows were written a launch earlier and are L2 / MALL resident; keep them warm (first rep is the cold one)
                hipLaunchKernelGGL(bcast_kernel, dim3(WG), dim3(256), 0, 0, buf, copy_units, use_pieces, mode, stride, use_passes, sink, stamp);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(hs.data(), stamp, WG * 2 * 8, hipMemcpyDeviceToHost));
                u64 s0 = ~0ull, e9 = 0;
                for (int b = 0; b < WG; ++b) { s0 = std::min(s0, hs[2 * b]); e9 = std::max(e9, hs[2 * b + 1]); }
                if (rep > 0) best = std::min(best, (double)(e9 - s0) * us);
            }
            printf("%3d KiB per workgroup, %-28s: %6.2f us  = %6.1f GB/s per CU\n", use_pieces, mode == 0 ? "same order (shared rows)" : (mode == 2 ? "private copy per workgroup" : (stride == 3 ? "rotated by 3 KiB x wg" : (stride == 10 ? "rotated by 10 KiB x wg" : "rotated by 37 KiB x wg"))),
                   best, use_pieces * 1024.0 / best / 1e3);
        }
        CK(hipFree(buf)); CK(hipFree(sink)); CK(hipFree(stamp));
    }
    return 0;
}
