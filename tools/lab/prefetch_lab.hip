// Kernel laboratory (not part of the product): does a RUN-AHEAD WEIGHT PREFETCHER shorten a chain of dependent decode GEMVs?
//
// The decode step is a chain of ~180 dependent phases; each costs ~3.3 us of latency (boundary, activation round trip, staging,
// reduce, store) during which HBM is idle, and only then streams its weights.  Weights do not depend on activations.  Here a
// SIDECAR kernel (one 2-wave workgroup per CU, on a forked graph branch) streams the weights of the phases AHEAD of the running
// one through the CU into its XCD's L2 by LDS-DMA (global_load_lds_dwordx4 into a 1 KiB dump slot: no VGPRs, nothing is read
// back), so that the consumer's one round of loads hits L2 instead of HBM.  The kernels of the chain stay ordinary launches
// (a kernel boundary is the cheapest all-to-all edge on this chip, MI355X_MICROARCH.md price list).
//
//   pacing   : every chain kernel publishes "stream offset reached" in one device word (block 0: start offset at entry, end
//              offset at exit); the sidecar stays at most LEAD bytes ahead of it (L2 is 4 MiB per XCD) and skips what the
//              consumer has already begun.  A poller wave refreshes the word into LDS, the loader wave never drains its queue.
//   placement: consumer block b runs on XCD b % 8 (observed, speed only); sidecar workgroup j prefetches the byte ranges of
//              consumer blocks b = j, j + 256, ...  `shift` moves that mapping by one XCD: the lines then land in the WRONG L2
//              (control: what the Infinity Cache alone gives).
// All spins are bounded by the wall clock.  build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/prefetch_lab.hip -o tools/lab/prefetch_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
constexpr uint32_t END_MARK = 0xffffffffu;

struct ChainArgs {
    int idx, n, wl, nt;          // n activation words; wl = 16-byte weight loads per thread (<= 10)
    int *err;
    const uint32_t *in;
    uint32_t *out;
    const u32x4 *w;
    u64 *stamp;                  // [kernels][1024 wgs][2]: start, end
    uint32_t *progress;          // null: no publishing
    uint32_t start64, end64;     // stream offsets of this kernel's weights in 64-byte units
};

__global__ __launch_bounds__(512) void chain_kernel(const ChainArgs p) {
    __shared__ uint32_t s_part[8];
    const int tid = threadIdx.x, T = blockDim.x;
    u64 *my = p.stamp + ((size_t)p.idx * 1024 + blockIdx.x) * 2;
    if (tid == 0) {
        my[0] = wall_clock64();
        if (p.progress && blockIdx.x == 0) __hip_atomic_store(p.progress, p.start64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // activations first (needed first; vmcnt retires in order), then the whole weight slice in one round -- as qmv3_kernel
    uint32_t s = 0;
    for (int i = tid; i < p.n; i += T) s += p.in[i];
    u32x4 wv[10];
    const u32x4 *wp = p.w + (size_t)blockIdx.x * T * p.wl + tid;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const u32x4 *q = wp + (size_t)(i < p.wl ? i : 0) * T;
        wv[i] = p.nt ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) s_part[tid >> 6] = s;
    __syncthreads();
    s = 0;
    for (int w = 0; w < T / 64; ++w) s += s_part[w];
    if (tid == 0 && s != (uint32_t)p.n * (uint32_t)p.idx) atomicAdd(&p.err[1], 1);
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) x ^= wv[i][0] ^ wv[i][1] ^ wv[i][2] ^ wv[i][3];
    const uint32_t v = s / (uint32_t)p.n + 1u + (x == 0x9e3779b9u ? 1u : 0u);
    const int per = (p.n + gridDim.x - 1) / gridDim.x;
    for (int j = blockIdx.x * per + tid; j < min(p.n, (int)(blockIdx.x + 1) * per); j += T) p.out[j] = v;
    __syncthreads();
    if (tid == 0) {
        if (p.progress && blockIdx.x == 0) __hip_atomic_store(p.progress, p.end64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        my[1] = wall_clock64();
    }
}

struct Seg {
    const char *base;
    uint32_t blocks, bpb;       // consumer blocks, bytes per block (a multiple of 1 KiB)
    uint32_t start64, end64;
};
struct SideArgs {
    const Seg *segs;
    int nseg;
    const uint32_t *progress;
    uint32_t lead64;
    int shift;                  // 1: map my ranges one XCD off (control)
    int *err;                   // err[2]: wall-clock give-ups, err[3]: segments skipped because the consumer had begun
    u64 *side_stamp;            // [nseg]: wall clock when workgroup 0 finished issuing segment s
    u64 give_up_ticks;
    int window;                 // most outstanding 1 KiB pieces / 8 (vmcnt), 2..6
};

__device__ __forceinline__ void dma_1k(const char *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// LDS word read that touches lgkmcnt only (a volatile __shared__ read compiles to flat_load + vmcnt(0): it would drain the DMA queue)
__device__ __forceinline__ uint32_t lds_peek(uint32_t addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}

__global__ __launch_bounds__(128) void sidecar_kernel(const SideArgs p) {
    __shared__ __attribute__((aligned(1024))) char dump[1024];
    __shared__ volatile uint32_t s_cons;
    __shared__ volatile uint32_t s_done;
    __shared__ Seg s_segs[160];  // the schedule, staged once: reading it from memory would drain the loader's queue (vmcnt) per segment
    for (int i = threadIdx.x; i < p.nseg && i < 160; i += blockDim.x) s_segs[i] = p.segs[i];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) { s_cons = 0; s_done = 0; }
    __syncthreads();
    const u64 t_begin = wall_clock64();
    if (wave == 1) {  // poller: one relaxed agent-scope load per round, result parked in LDS for the loader
        while (true) {
            const uint32_t v = __hip_atomic_load(p.progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) s_cons = v;
            if (v == END_MARK || s_done) break;
            if (wall_clock64() - t_begin > p.give_up_ticks) break;
            __builtin_amdgcn_s_sleep(6);
        }
        return;
    }
    const uint32_t lds_dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)dump);
    const uint32_t cons_addr = (uint32_t)(uintptr_t)&s_cons;
    const int nwg = gridDim.x;
    const int lid = ((int)blockIdx.x + (p.shift ? 1 : 0)) % nwg;  // whose consumer blocks I prefetch
    int in_batch = 0;
    bool gave_up = false;
    for (int s = 0; s < p.nseg && !gave_up; ++s) {
        const Seg sg = s_segs[s];
        const uint32_t bpb64 = sg.bpb >> 6;
        bool skipped = false;
        for (uint32_t b = lid; b < sg.blocks; b += nwg) {
            const uint32_t pos_end = sg.start64 + (b + 1) * bpb64;
            while (true) {
                const uint32_t cons = lds_peek(cons_addr);
                if (cons >= sg.start64 && !(s == 0 && cons == 0)) { skipped = true; break; }  // the consumer is already loading this segment
                if (pos_end <= cons + p.lead64) break;
                if (wall_clock64() - t_begin > p.give_up_ticks) { gave_up = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (skipped || gave_up) break;
            const char *src = sg.base + (size_t)b * sg.bpb + lane * 16;
            const int pieces = sg.bpb >> 10;
            for (int i = 0; i < pieces; ++i) {
                dma_1k(src + (size_t)i * 1024, lds_dst);
                if (++in_batch == 8) {
                    in_batch = 0;
                    switch (p.window) {
                        case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                        case 3: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                        case 4: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
                        case 5: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
                        default: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
                    }
                }
            }
        }
        if (blockIdx.x == 0 && lane == 0) {
            p.side_stamp[s] = wall_clock64();
            if (skipped) atomicAdd(&p.err[3], 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA must have landed before the workgroup gives its LDS back
    if (gave_up && lane == 0) atomicAdd(&p.err[2], 1);
    if (lane == 0) s_done = 1;
}

__global__ void init_kernel(uint32_t *progress, uint32_t *act, int n) {
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(progress, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) act[i] = 0;
}
__global__ void end_kernel(uint32_t *progress) {
    if (threadIdx.x == 0) __hip_atomic_store(progress, END_MARK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// flush: streams a large buffer so that neither L2 nor the Infinity Cache holds the chain's weights at the start of a replay
__global__ void flush_kernel(const u32x4 *buf, size_t n, uint32_t *sink) {
    uint32_t x = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = buf[i];
        x ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (x == 0x12345678u) *sink = x;
}

int main(int argc, char **argv) {
    const int LAYERS = 36, n = 1280;
    // Qwen3-4B decode projections as qmv3 launches them: blocks x threads x 16-byte loads per thread
    //   qkv 192 x 256 x 10 (40 KiB / block, 7.9 MB), wo 160 x 256 x 8 (32 KiB, 5.2 MB), gate|up 608 x 256 x 10 (24.9 MB),
    //   w_down 160 x 512 x 10 (80 KiB, 13.1 MB)
    struct Kind { int blocks, threads, wl; };
    const Kind kinds[4] = {{192, 256, 10}, {160, 256, 8}, {608, 256, 10}, {160, 512, 10}};
    const int kernels = LAYERS * 4;
    int khz; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    const double us_per_tick = 1e3 / khz;
    size_t layer_bytes = 0;
    for (auto &k : kinds) layer_bytes += (size_t)k.blocks * k.threads * k.wl * 16;
    const size_t wbytes = layer_bytes * LAYERS;  // every phase has its own weights: 1.84 GB, nothing re-read within a replay
    char *w; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 1, wbytes));
    uint32_t *act[2]; for (auto &a : act) CK(hipMalloc(&a, n * 4));
    int *err; CK(hipMalloc(&err, 16)); CK(hipMemset(err, 0, 16));
    uint32_t *progress; CK(hipMalloc(&progress, 256)); CK(hipMemset(progress, 0, 256));
    u64 *stamp; CK(hipMalloc(&stamp, (size_t)kernels * 1024 * 16));
    u64 *side_stamp; CK(hipMalloc(&side_stamp, (size_t)kernels * 8));
    uint32_t *sink; CK(hipMalloc(&sink, 4));
    hipStream_t st[2]; for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1, fork, join; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, wall clock %d kHz, chain %d kernels, %.1f MB per layer, %.2f GB per replay\n", prop.name,
           prop.multiProcessorCount, khz, kernels, layer_bytes / 1e6, wbytes / 1e9);

    // segments in consumption order
    std::vector<Seg> segs(kernels);
    std::vector<ChainArgs> cargs(kernels);
    {
        size_t off = 0;
        for (int k = 0; k < kernels; ++k) {
            const Kind &kd = kinds[k & 3];
            const size_t bytes = (size_t)kd.blocks * kd.threads * kd.wl * 16;
            segs[k] = Seg{w + off, (uint32_t)kd.blocks, (uint32_t)(kd.threads * kd.wl * 16), (uint32_t)(off >> 6), (uint32_t)((off + bytes) >> 6)};
            ChainArgs a{};
            a.idx = k; a.n = n; a.wl = kd.wl; a.err = err; a.in = act[k & 1]; a.out = act[(k + 1) & 1];
            a.w = reinterpret_cast<const u32x4 *>(w + off); a.stamp = stamp; a.start64 = segs[k].start64; a.end64 = segs[k].end64;
            cargs[k] = a;
            off += bytes;
        }
    }
    Seg *dsegs; CK(hipMalloc(&dsegs, sizeof(Seg) * kernels));
    CK(hipMemcpy(dsegs, segs.data(), sizeof(Seg) * kernels, hipMemcpyHostToDevice));

    struct Variant { const char *name; int side, lead_mb, shift, nt, window, publish, side_wgs; };
    std::vector<Variant> vs = {
        {"baseline (no sidecar, no publishing)", 0, 0, 0, 1, 6, 0, 256},
        {"publishing only", 0, 0, 0, 1, 6, 1, 256},
        {"sidecar lead 12 MB", 1, 12, 0, 1, 6, 1, 256},
        {"sidecar lead 20 MB", 1, 20, 0, 1, 6, 1, 256},
        {"sidecar lead 28 MB", 1, 28, 0, 1, 6, 1, 256},
        {"sidecar lead 48 MB", 1, 48, 0, 1, 6, 1, 256},
        {"sidecar lead 20 MB, wrong XCD (control)", 1, 20, 1, 1, 6, 1, 256},
        {"sidecar lead 20 MB, consumer plain loads", 1, 20, 0, 0, 6, 1, 256},
        {"baseline, consumer plain loads", 0, 0, 0, 0, 6, 0, 256},
        {"sidecar lead 20 MB, window 24", 1, 20, 0, 1, 3, 1, 256},
        {"sidecar lead 20 MB, 512 workgroups", 1, 20, 0, 1, 6, 1, 512},
        {"sidecar lead 28 MB, 512 workgroups", 1, 28, 0, 1, 6, 1, 512},
    };
    if (argc > 1) {  // custom: side lead_mb shift nt window publish side_wgs
        Variant v{"custom", 1, 20, 0, 1, 6, 1, 256};
        if (argc > 1) v.side = atoi(argv[1]);
        if (argc > 2) v.lead_mb = atoi(argv[2]);
        if (argc > 3) v.shift = atoi(argv[3]);
        if (argc > 4) v.nt = atoi(argv[4]);
        if (argc > 5) v.window = atoi(argv[5]);
        if (argc > 6) v.publish = atoi(argv[6]);
        if (argc > 7) v.side_wgs = atoi(argv[7]);
        vs = {v};
    }
    for (const Variant &v : vs) {
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeGlobal));
        hipLaunchKernelGGL(init_kernel, dim3(8), dim3(256), 0, st[0], progress, act[0], n);
        if (v.side) {
            CK(hipEventRecord(fork, st[0])); CK(hipStreamWaitEvent(st[1], fork, 0));
            SideArgs sa{};
            sa.segs = dsegs; sa.nseg = kernels; sa.progress = progress; sa.lead64 = (uint32_t)(((size_t)v.lead_mb << 20) >> 6);
            sa.shift = v.shift; sa.err = err; sa.side_stamp = side_stamp; sa.give_up_ticks = (u64)khz * 20;  // 20 ms
            sa.window = v.window;
            hipLaunchKernelGGL(sidecar_kernel, dim3(v.side_wgs), dim3(128), 0, st[1], sa);
        }
        for (int k = 0; k < kernels; ++k) {
            ChainArgs a = cargs[k];
            a.nt = v.nt;
            a.progress = v.publish ? progress : nullptr;
            const Kind &kd = kinds[k & 3];
            hipLaunchKernelGGL(chain_kernel, dim3(kd.blocks), dim3(kd.threads), 0, st[0], a);
        }
        hipLaunchKernelGGL(end_kernel, dim3(1), dim3(64), 0, st[0], progress);
        if (v.side) { CK(hipEventRecord(join, st[1])); CK(hipStreamWaitEvent(st[0], join, 0)); }
        CK(hipStreamEndCapture(st[0], &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, st[0]));
        CK(hipStreamSynchronize(st[0]));
        CK(hipMemset(err, 0, 16));
        const int reps = 20;
        CK(hipEventRecord(e0, st[0]));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(exec, st[0]));
        CK(hipEventRecord(e1, st[0]));
        CK(hipStreamSynchronize(st[0]));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<u64> hs((size_t)kernels * 1024 * 2); CK(hipMemcpy(hs.data(), stamp, hs.size() * 8, hipMemcpyDeviceToHost));
        std::vector<u64> ss(kernels); CK(hipMemcpy(ss.data(), side_stamp, ss.size() * 8, hipMemcpyDeviceToHost));
        std::vector<u64> lo(kernels), hi(kernels);
        for (int k = 0; k < kernels; ++k) {
            u64 l = ~0ull, h = 0;
            for (int b = 0; b < kinds[k & 3].blocks; ++b) { const u64 *q = &hs[((size_t)k * 1024 + b) * 2]; l = std::min(l, q[0]); h = std::max(h, q[1]); }
            lo[k] = l; hi[k] = h;
        }
        double body[4] = {0, 0, 0, 0}, gap = 0, ahead[4] = {0, 0, 0, 0};
        int late[4] = {0, 0, 0, 0};
        for (int k = 4; k < kernels; ++k) {
            body[k & 3] += (double)(hi[k] - lo[k]) * us_per_tick;
            gap += (double)((long long)(lo[k] - hi[k - 1])) * us_per_tick;
            if (v.side) {
                const double a = (double)((long long)(lo[k] - ss[k])) * us_per_tick;  // > 0: workgroup 0 had issued the segment before the consumer started
                ahead[k & 3] += a;
                if (a < 0) late[k & 3]++;
            }
        }
        const int L = LAYERS - 1;
        std::vector<uint32_t> ha(n); CK(hipMemcpy(ha.data(), act[kernels & 1], n * 4, hipMemcpyDeviceToHost));
        int herr[4]; CK(hipMemcpy(herr, err, 16, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < n; ++i) bad += ha[i] != (uint32_t)kernels;
        printf("%-44s: graph %7.1f us = %5.2f us/kernel | in-kernel qkv %.2f wo %.2f gate_up %.2f down %.2f, gap %.2f us", v.name,
               ms * 1e3 / reps, ms * 1e3 / reps / kernels, body[0] / L, body[1] / L, body[2] / L, body[3] / L, gap / (kernels - 4));
        if (v.side) printf(" | issued ahead of consumer start by %.1f %.1f %.1f %.1f us (late %d %d %d %d) give-ups %d skipped %d", ahead[0] / L,
                           ahead[1] / L, ahead[2] / L, ahead[3] / L, late[0], late[1], late[2], late[3], herr[2], herr[3]);
        printf(" | stale %d final-bad %d\n", herr[1], bad);
        fflush(stdout);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    return 0;
}
