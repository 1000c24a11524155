// Kernel laboratory (not part of the product): what does an all-gather cost INSIDE one launch when it is confined to ONE XCD?
//
// Qwen3-4B has 8 KV heads and the chip has 8 XCDs (32 CUs and one 4 MiB L2 each).  Split tensor-parallel BY XCD -- qkv and
// gate|up by output rows, wo and w_down by input columns -- three of a decoder layer's five all-to-all edges (qkv -> attention,
// attention -> wo, act -> w_down) connect only the 32 workgroups of one XCD, whose L2 is coherent among them; only the two
// residual-stream edges cross XCDs (8 fp32 partial vectors summed by the consumer) and those can stay kernel boundaries.
// This lab prices the XCD-local edge: R dependent rounds inside one launch of 256 workgroups (one per CU); per round every
// workgroup publishes its slice of a vector as 8-byte {value, tag} granules and gathers the whole vector of its group.
//   group  = the 32 workgroups that report the same XCC_ID (rank by ticket), or all 256 (the chip-wide edge, for comparison)
//   stores = plain (stay in the XCD's L2) or sc1 (write-through: the placement-independent form)
//   loads  = sc1 (bypass L1)
//   stream = every round each workgroup also pulls 40 KiB of fresh weights from HBM, issued BEFORE the gather (weights do not
//            depend on activations); a dedicated gather wave polls so that no poll queues behind a weight load (vmcnt is in order)
// Every spin is bounded by the wall clock; a workgroup that gives up stops polling for good (the launch always ends).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/xcd_lab.hip -o tools/lab/xcd_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

struct LabArgs {
    int rounds, n;            // n values per group vector (a multiple of the group size)
    int group_all;            // 0: group = XCD, 1: all workgroups
    int sc1_store, stream;
    u64 *vec;                 // [groups][2 buffers][n] granules {value | tag << 32}
    unsigned *ticket;         // [8] per-XCC ranks (reset per launch by the host)
    int *err;                 // [0] give-ups, [1] wrong sums, [2] XCC_ID != blockIdx % 8, [3] rank overflow
    const u32x4 *w;
    size_t w_units;           // u32x4 units in w
    u64 *stamp;               // [rounds + 1] wall clock of workgroup 0
    u64 give_up_ticks;
    int epoch;                // tags are epoch * 65536 + round + 1: no reset of vec between launches
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

constexpr int CW = 4;  // compute waves; wave CW is the gather wave
__global__ __launch_bounds__((CW + 1) * 64) void xcd_kernel(const LabArgs p) {
    extern __shared__ uint32_t s_vec[];  // [n]
    __shared__ unsigned s_rank, s_group;
    __shared__ volatile unsigned s_gave_up;
    __shared__ uint32_t s_part[CW];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    if (tid == 0) {
        const unsigned x = xcc_id();
        if (x != (blockIdx.x & 7u)) atomicAdd(&p.err[2], 1);
        unsigned rank, group, gsize;
        if (p.group_all) { rank = blockIdx.x; group = 0; gsize = gridDim.x; }
        else {
            rank = atomicAdd(&p.ticket[x], 1u);
            group = x; gsize = gridDim.x / 8;
            if (rank >= gsize) { atomicAdd(&p.err[3], 1); rank = rank % gsize; }
        }
        s_rank = rank; s_group = group; s_gave_up = 0;
    }
    __syncthreads();
    const unsigned rank = s_rank, group = s_group;
    const int gsize = p.group_all ? (int)gridDim.x : (int)gridDim.x / 8;
    const int per = p.n / gsize;  // values this workgroup publishes per round
    const u64 t_begin = wall_clock64();
    uint32_t carry = 0;  // every value of round r is r (+ carry, which stays 0 when all sums are right)
    for (int r = 0; r < p.rounds; ++r) {
        if (blockIdx.x == 0 && tid == 0) p.stamp[r] = wall_clock64();
        u64 *buf = p.vec + ((size_t)group * 2 + (r & 1)) * p.n;
        const uint32_t tag = (uint32_t)p.epoch * 65536u + (uint32_t)r + 1u;
        u32x4 wv[10];
        if (wave < CW) {
            // publish my slice of round r (depends on the gathered vector of round r - 1 through `carry`)
            if (tid < per) {
                const u64 g = (u64)((uint32_t)r + carry) | ((u64)tag << 32);
                u64 *dst = buf + (size_t)rank * per + tid;
                if (p.sc1_store) __hip_atomic_store(dst, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *reinterpret_cast<volatile u64 *>(dst) = g;
            }
            // this round's weights: independent of the activations, issued before the gather completes
            if (p.stream) {
                const size_t unit = ((size_t)r * gridDim.x + blockIdx.x) * (CW * 64 * 10) % (p.w_units - CW * 64 * 10);
                const u32x4 *wp = p.w + unit + tid;
#pragma unroll
                for (int i = 0; i < 10; ++i) wv[i] = __builtin_nontemporal_load(wp + (size_t)i * CW * 64);
            }
        } else {
            // gather wave: sweep the group's vector until every granule carries this round's tag.  All of a lane's granules are
            // in flight together (a sweep is one round trip, not n / 64 of them); a sweep that finds a stale tag is repeated.
            constexpr int NVMAX = 40;
            const int nv = p.n >> 6;
            u64 g[NVMAX];
            while (true) {
#pragma unroll
                for (int k = 0; k < NVMAX; ++k)
                    g[k] = __hip_atomic_load(buf + lane + 64 * (k < nv ? k : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool ok = true;
#pragma unroll
                for (int k = 0; k < NVMAX; ++k) ok = ok && (uint32_t)(g[k] >> 32) == tag;
                if (__all(ok)) break;
                if (s_gave_up) break;
                if (wall_clock64() - t_begin > p.give_up_ticks) { s_gave_up = 1; if (lane == 0) atomicAdd(&p.err[0], 1); break; }
            }
#pragma unroll
            for (int k = 0; k < NVMAX; ++k)
                if (k < nv) s_vec[lane + 64 * k] = (uint32_t)g[k];
        }
        __syncthreads();
        uint32_t s = 0;
        if (wave < CW) {
            for (int i = tid; i < p.n; i += CW * 64) s += s_vec[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == 0) s_part[wave] = s;
        }
        __syncthreads();
        s = 0;
        for (int w2 = 0; w2 < CW; ++w2) s += s_part[w2];
        uint32_t x = 0;
        if (wave < CW && p.stream) {
#pragma unroll
            for (int i = 0; i < 10; ++i) x ^= wv[i][0] ^ wv[i][1] ^ wv[i][2] ^ wv[i][3];
        }
        const bool right = s == (uint32_t)p.n * (uint32_t)r;
        if (tid == 0 && !right && !s_gave_up) atomicAdd(&p.err[1], 1);
        carry = (right ? 0u : 1u) + (x == 0x9e3779b9u ? 1u : 0u);
        carry = s_gave_up ? 0u : carry;
    }
    if (blockIdx.x == 0 && tid == 0) p.stamp[p.rounds] = wall_clock64();
}

// the same dependent rounds as separate launches (one kernel boundary per round): every workgroup reads the whole vector of
// the previous launch, pulls its weights, writes its slice
__global__ __launch_bounds__(CW * 64) void round_kernel(const uint32_t *in, uint32_t *out, int n, int r, const u32x4 *w, size_t w_units,
                                                         int stream, int *err) {
    __shared__ uint32_t s_part[CW];
    const int tid = threadIdx.x;
    uint32_t s = 0;
    for (int i = tid; i < n; i += CW * 64) s += in[i];
    u32x4 wv[10];
    if (stream) {
        const size_t unit = ((size_t)r * gridDim.x + blockIdx.x) * (CW * 64 * 10) % (w_units - CW * 64 * 10);
        const u32x4 *wp = w + unit + tid;
#pragma unroll
        for (int i = 0; i < 10; ++i) wv[i] = __builtin_nontemporal_load(wp + (size_t)i * CW * 64);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) s_part[tid >> 6] = s;
    __syncthreads();
    s = 0;
    for (int w2 = 0; w2 < CW; ++w2) s += s_part[w2];
    uint32_t x = 0;
    if (stream) {
#pragma unroll
        for (int i = 0; i < 10; ++i) x ^= wv[i][0] ^ wv[i][1] ^ wv[i][2] ^ wv[i][3];
    }
    if (tid == 0 && s != (uint32_t)n * (uint32_t)r) atomicAdd(&err[1], 1);
    const int per = n / gridDim.x;
    if (tid < per) out[blockIdx.x * per + tid] = (uint32_t)r + 1u + (x == 0x9e3779b9u ? 1u : 0u);
}

int main() {
    int khz; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    const double us = 1e3 / khz;
    const int WG = 256, R = 200;
    const size_t wbytes = (size_t)2 << 30;
    u32x4 *w; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 1, wbytes));
    u64 *vec; CK(hipMalloc(&vec, (size_t)8 * 2 * 8192 * 8)); CK(hipMemset(vec, 0, (size_t)8 * 2 * 8192 * 8));
    unsigned *ticket; CK(hipMalloc(&ticket, 64));
    int *err; CK(hipMalloc(&err, 16));
    u64 *stamp; CK(hipMalloc(&stamp, (R + 1) * 8));
    uint32_t *act[2]; for (auto &a : act) CK(hipMalloc(&a, 8192 * 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int epoch = 1;
    printf("xcd_lab: %d workgroups x %d threads, %d dependent rounds per launch\n", WG, (CW + 1) * 64, R);
    for (int stream = 0; stream < 2; ++stream) {
        // reference: one launch per round (graph-captured), the whole vector re-read by every workgroup after the boundary
        for (int n : {1024, 2560}) {
            hipGraph_t graph; hipGraphExec_t exec;
            CK(hipMemset(act[0], 0, 8192 * 4)); CK(hipMemset(err, 0, 16));
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int r = 0; r < R; ++r)
                hipLaunchKernelGGL(round_kernel, dim3(WG), dim3(CW * 64), 0, st, act[r & 1], act[(r + 1) & 1], n, r, w, wbytes / 16, stream, err);
            CK(hipStreamEndCapture(st, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemsetAsync(act[0], 0, 8192 * 4, st));
                CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(exec, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
            }
            int herr[4]; CK(hipMemcpy(herr, err, 16, hipMemcpyDeviceToHost));
            printf("stream %d  launches (one boundary per round), n %4d               : %6.2f us/round | wrong sums %d\n", stream, n, best * 1e3 / R, herr[1]);
            CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
        }
        for (int group_all = 0; group_all < 2; ++group_all)
        for (int sc1_store = 0; sc1_store < 2; ++sc1_store)
        for (int n : {1024, 2560}) {
            if (group_all && !sc1_store) continue;  // plain stores are not a cross-XCD protocol
            LabArgs a{};
            a.rounds = R; a.n = n; a.group_all = group_all; a.sc1_store = sc1_store; a.stream = stream;
            a.vec = vec; a.ticket = ticket; a.err = err; a.w = w; a.w_units = wbytes / 16; a.stamp = stamp;
            a.give_up_ticks = (u64)khz * 30;  // 30 ms for the whole launch
            float best = 1e9f;
            int herr[4] = {0, 0, 0, 0};
            std::vector<u64> hst(R + 1);
            std::vector<double> per_round;
            for (int rep = 0; rep < 5; ++rep) {
                a.epoch = epoch++;
                CK(hipMemsetAsync(ticket, 0, 64, st)); CK(hipMemsetAsync(err, 0, 16, st));
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(xcd_kernel, dim3(WG), dim3((CW + 1) * 64), n * 4, st, a);
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                int h2[4]; CK(hipMemcpy(h2, err, 16, hipMemcpyDeviceToHost));
                for (int i = 0; i < 4; ++i) herr[i] += h2[i];
                if (ms < best) {
                    best = ms;
                    CK(hipMemcpy(hst.data(), stamp, (R + 1) * 8, hipMemcpyDeviceToHost));
                    per_round.clear();
                    for (int r = 8; r < R; ++r) per_round.push_back((double)(hst[r + 1] - hst[r]) * us);
                    std::sort(per_round.begin(), per_round.end());
                }
            }
            printf("stream %d  in-launch, group %-3s, %-5s stores, sc1 loads, n %4d : %6.2f us/round (median %.2f, p90 %.2f) | give-ups %d wrong sums %d xcc!=b%%8 %d rank overflow %d\n",
                   stream, group_all ? "all" : "XCD", sc1_store ? "sc1" : "plain", n, best * 1e3 / R, per_round[per_round.size() / 2],
                   per_round[per_round.size() * 9 / 10], herr[0], herr[1], herr[2], herr[3]);
            fflush(stdout);
        }
    }
    return 0;
}
