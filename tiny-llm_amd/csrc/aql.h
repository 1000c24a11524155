// AQL replay: a captured decode step as hand-written dispatch packets on the engine's own HSA queue (gfx950, ROCr).
//
// Why (round 5, profiles/r05_labs/README.md).  A single-sequence decode step is 182 dependent launches of 3-7 us; what is left
// between the kernels' own floors is the boundary -- 1.1-1.3 us per launch in a replayed hipGraph.  Measured with hand-written
// packets (tools/lab/overlap_lab.hip): the barrier bit alone is 0.7-0.8 us of it, the agent-scope acquire / release fences HIP puts on
// every kernel (an L2 write-back and an L2 invalidate across the eight XCDs) the other 0.4-0.55 us -- and the invalidate is also what
// makes every launch start on a cold L2.  Packets of one queue cannot overlap on this chip (a successor starts per XCD when its
// predecessor has drained there), so the boundary itself stays; but a step whose kernels exchange their activations through
// device-scope (L2-bypassing) loads and write-through stores needs NO cache maintenance between its launches, and then nothing
// invalidates what a prefetcher on a second queue has brought into an XCD's L2 ahead of the launch that reads it.
//
// This file is the plumbing: the ROCr objects (agent, queues, one executable per device-only code object built next to the library,
// kernel descriptors + argument layouts from tl_kernels.meta), a program = the kernel nodes of a captured hipGraph turned into packet
// templates + one argument buffer, and submission (templates copied into the ring, headers stored last, one doorbell per step).
// HIP stays the owner of everything else (memory, streams, the captured graph the program is built FROM -- and the route the
// engine falls back to whenever a program cannot be built).
#pragma once
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace tl {

struct AqlKernelInfo {
    uint64_t object = 0;          // kernel descriptor address
    uint32_t kernarg_bytes = 0;   // whole segment (explicit + implicit)
    uint32_t group_static = 0, private_bytes = 0;
    int hidden_base = -1;         // offset of hidden_block_count_x (code object v5), -1: the kernel reads no implicit argument
    std::vector<std::pair<uint32_t, uint32_t>> args;  // explicit arguments: (offset, size)
};

// One per HIP DEVICE (process-wide table, created on first use under a lock): the HSA agent of that device and the code objects
// loaded on it.  An engine keeps the runtime of the device it was created on; queues, kernel objects and programs of one runtime
// are only ever used with buffers of that device (round 5's single instance bound itself to whichever device was current at the
// first engine creation: a second engine on another GPU of the same process would have dispatched on the wrong agent).
class AqlRuntime {
public:
    static AqlRuntime &for_device(int hip_device);
    // loads every tl_kernels_*.hsaco + tl_kernels.meta of `dir` on this runtime's device, once; false (and why()) when anything is
    // missing.  Thread-safe.
    bool ensure_loaded(const std::string &dir);
    int device() const { return device_; }
    const AqlKernelInfo *find(const std::string &mangled) const;
    hsa_agent_t agent() const { return agent_; }
    const std::string &why() const { return why_; }
    bool ok() const { return ok_; }

private:
    explicit AqlRuntime(int dev) : device_(dev) {}
    int device_ = 0;
    bool ok_ = false, tried_ = false;
    std::string why_;
    hsa_agent_t agent_{};
    std::vector<hsa_executable_t> exes_;
    std::vector<std::vector<char>> blobs_;
    std::map<std::string, AqlKernelInfo> kernels_;
    bool fail(const std::string &msg) {
        why_ = msg;
        return false;
    }
};

// one captured step: packet templates (header field left zero) + their kernel arguments in ONE device buffer
struct AqlProgram {
    std::vector<hsa_kernel_dispatch_packet_t> packets;
    std::vector<std::string> names;  // kernel of every packet (diagnostics, tests)
    char *kernarg_dev = nullptr;     // hipMalloc'ed; the packets point into it
    size_t kernarg_bytes = 0;
    ~AqlProgram();
    AqlProgram() = default;
    AqlProgram(const AqlProgram &) = delete;
    AqlProgram &operator=(const AqlProgram &) = delete;
};

// Builds a program from the kernel nodes of a captured graph (a linear chain: anything else is refused).  Returns 0, or a negative
// code with `why` set: the caller keeps the hipGraph route.
int aql_program_from_graph(const AqlRuntime &rt, hipGraph_t graph, hipStream_t stream, AqlProgram &out, std::string &why);

struct AqlFences {
    // fence scopes (hsa_fence_scope_t) of the packets INSIDE a step; the first packet of a submission always acquires at system scope
    // and the last one releases at system scope
    int inner_acquire = HSA_FENCE_SCOPE_AGENT, inner_release = HSA_FENCE_SCOPE_AGENT;
    // ... and of the first / last packet of every step but the outermost ones
    int step_acquire = HSA_FENCE_SCOPE_AGENT, step_release = HSA_FENCE_SCOPE_AGENT;
};

class AqlQueue {
public:
    ~AqlQueue();
    bool create(const AqlRuntime &rt, std::string &why, uint32_t packets = 8192);
    // copies the program's packets into the ring `times` times (steps back to back); `first` / `last` mark the outermost packets of the
    // whole submission (system-scope fences, completion signal on the very last packet)
    bool submit(const AqlProgram &p, const AqlFences &f, bool first, bool last, std::string &why);
    // for everything submitted so far: the completion signal of the last submit(..., last = true), or -- when steps went out without one
    // (a drain in the middle of a call: a page id changes, a plan leaves the route) -- of a barrier packet put behind them first
    bool wait(double seconds, std::string &why);
    bool busy() const { return pending_ || unsignalled_; }

private:
    bool submit_barrier(std::string &why);
    hsa_queue_t *q_ = nullptr;
    hsa_signal_t done_{};
    bool pending_ = false;      // a packet carrying the completion signal is in the ring
    bool unsignalled_ = false;  // packets went out after the last one that carries the signal
};

}  // namespace tl
