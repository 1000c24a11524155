// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds u16 element indices; every lane reads 8 bytes at lane * 8 (lane-linear) and at a
// few other address patterns; the 4 returned 16-bit values per lane tell which LDS elements the hardware hands to which lane.
// build: hipcc -O3 --offload-arch=gfx950 tools/lab/tr_probe.hip -o tools/lab/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(uint16_t *out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    int elem;  // element index each lane points at (4 contiguous elements = 8 bytes)
    if (mode == 0) elem = l * 4;                                         // lane-linear
    else if (mode == 1) elem = (l & 15) * 4 + (l >> 4) * 64;              // same as 0 (a 16-lane group = 64 contiguous elements)
    else if (mode == 2) elem = ((l & 15) >> 2) * 128 + (l & 3) * 4 + (l >> 4) * 16;  // rows 256 B apart: [4 rows][16 cols] blocks of a [.][128] image
    else elem = ((l & 15) >> 2) * 16 + (l & 3) * 4 + (l >> 4) * 512;      // [32][16] subtiles: rows 32 B apart, a group per subtile quarter
    const uint32_t addr = (uint32_t)(uintptr_t)(lds) + elem * 2;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16; out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}
int main() {
    uint16_t *d; CK(hipMalloc(&d, 64 * 4 * 2));
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode); CK(hipDeviceSynchronize());
        uint16_t h[256]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("  lane %2d: %4d %4d %4d %4d", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]); if ((l & 3) == 3) printf("\n"); }
    }
    return 0;
}
