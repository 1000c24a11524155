// AQL replay plumbing (aql.h): ROCr agent / executables / kernel descriptors, program building from a captured hipGraph, submission.
#include "aql.h"

#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <sstream>

#include <dirent.h>

namespace tl {

namespace {

std::string hsa_err(hsa_status_t s) {
    const char *m = nullptr;
    hsa_status_string(s, &m);
    return m ? std::string(m) : std::string("HSA status ") + std::to_string((int)s);
}

struct AgentPick {
    std::vector<hsa_agent_t> gpus;
};
hsa_status_t collect_gpu(hsa_agent_t a, void *data) {
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_GPU) ((AgentPick *)data)->gpus.push_back(a);
    return HSA_STATUS_SUCCESS;
}

}  // namespace

namespace {
std::mutex g_runtime_lock;  // the table of runtimes and every ensure_loaded
}

AqlRuntime &AqlRuntime::for_device(int hip_device) {
    static std::map<int, std::unique_ptr<AqlRuntime>> table;
    std::lock_guard<std::mutex> guard(g_runtime_lock);
    auto &slot = table[hip_device];
    if (!slot) slot.reset(new AqlRuntime(hip_device));
    return *slot;
}

bool AqlRuntime::ensure_loaded(const std::string &dir) {
    std::lock_guard<std::mutex> guard(g_runtime_lock);
    if (tried_) return ok_;
    tried_ = true;
    hsa_status_t s = hsa_init();  // reference-counted: HIP holds the runtime already
    if (s != HSA_STATUS_SUCCESS) return fail("hsa_init: " + hsa_err(s));
    AgentPick pick;
    hsa_iterate_agents(collect_gpu, &pick);
    if (pick.gpus.empty()) return fail("no HSA GPU agent");
    // the agent of THIS runtime's HIP device: by PCI bus / device / function, else by ordinal
    const int dev = device_;
    char bus_id[64] = {0};
    unsigned want_bdf = ~0u;
    if (hipDeviceGetPCIBusId(bus_id, sizeof(bus_id), dev) == hipSuccess) {
        unsigned dom = 0, bus = 0, d = 0, fn = 0;
        if (sscanf(bus_id, "%x:%x:%x.%x", &dom, &bus, &d, &fn) == 4) want_bdf = (bus << 8) | (d << 3) | fn;
    }
    bool found = false;
    for (hsa_agent_t a : pick.gpus) {
        uint32_t bdf = 0;
        if (want_bdf != ~0u && hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) == HSA_STATUS_SUCCESS && bdf == want_bdf) {
            agent_ = a;
            found = true;
            break;
        }
    }
    if (!found) {
        if (dev < 0 || dev >= (int)pick.gpus.size()) return fail("cannot match HIP device " + std::to_string(dev) + " to an HSA agent");
        agent_ = pick.gpus[dev];
    }
    char agent_name[64] = {0};
    hsa_agent_get_info(agent_, HSA_AGENT_INFO_NAME, agent_name);
    if (strncmp(agent_name, "gfx950", 6) != 0) return fail(std::string("the code objects are built for gfx950, the agent is ") + agent_name);

    // argument layouts
    std::ifstream meta(dir + "/tl_kernels.meta");
    if (!meta) return fail("missing " + dir + "/tl_kernels.meta (make -C tiny-llm_amd/csrc builds it)");
    std::map<std::string, AqlKernelInfo> layouts;
    std::string line;
    while (std::getline(meta, line)) {
        std::istringstream is(line);
        std::string name;
        AqlKernelInfo k;
        int n_args = 0;
        if (!(is >> name >> k.kernarg_bytes >> k.hidden_base >> n_args)) continue;
        for (int i = 0; i < n_args; ++i) {
            std::string tok;
            if (!(is >> tok)) return fail("tl_kernels.meta: truncated line for " + name);
            const size_t c = tok.find(':');
            if (c == std::string::npos) return fail("tl_kernels.meta: bad argument token for " + name);
            k.args.emplace_back((uint32_t)std::stoul(tok.substr(0, c)), (uint32_t)std::stoul(tok.substr(c + 1)));
        }
        layouts[name] = k;
    }
    if (layouts.empty()) return fail("tl_kernels.meta holds no kernel");

    // code objects: one executable each (kernels of a shared header exist in more than one of them under the same name)
    std::vector<std::string> files;
    if (DIR *d = opendir(dir.c_str())) {
        while (dirent *e = readdir(d)) {
            const std::string f = e->d_name;
            if (f.rfind("tl_kernels_", 0) == 0 && f.size() > 6 && f.substr(f.size() - 6) == ".hsaco") files.push_back(dir + "/" + f);
        }
        closedir(d);
    }
    std::sort(files.begin(), files.end());
    if (files.empty()) return fail("no tl_kernels_*.hsaco in " + dir);
    for (const std::string &path : files) {
        std::ifstream f(path, std::ios::binary);
        std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (blob.empty()) return fail("cannot read " + path);
        blobs_.push_back(std::move(blob));
        hsa_code_object_reader_t reader;
        if ((s = hsa_code_object_reader_create_from_memory(blobs_.back().data(), blobs_.back().size(), &reader)) != HSA_STATUS_SUCCESS)
            return fail(path + ": code object reader: " + hsa_err(s));
        hsa_executable_t exe;
        if ((s = hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe)) != HSA_STATUS_SUCCESS)
            return fail("hsa_executable_create_alt: " + hsa_err(s));
        if ((s = hsa_executable_load_agent_code_object(exe, agent_, reader, nullptr, nullptr)) != HSA_STATUS_SUCCESS)
            return fail(path + ": load: " + hsa_err(s));
        if ((s = hsa_executable_freeze(exe, nullptr)) != HSA_STATUS_SUCCESS) return fail(path + ": freeze: " + hsa_err(s));
        exes_.push_back(exe);
    }
    for (auto &kv : layouts) {
        const std::string sym_name = kv.first + ".kd";
        for (hsa_executable_t exe : exes_) {
            hsa_executable_symbol_t sym;
            if (hsa_executable_get_symbol_by_name(exe, sym_name.c_str(), &agent_, &sym) != HSA_STATUS_SUCCESS) continue;
            AqlKernelInfo k = kv.second;
            uint32_t ka = 0;
            if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object) != HSA_STATUS_SUCCESS) continue;
            hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group_static);
            hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.private_bytes);
            hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &ka);
            if (ka != k.kernarg_bytes) return fail(kv.first + ": argument segment of " + std::to_string(ka) + " bytes, tl_kernels.meta says " + std::to_string(k.kernarg_bytes));
            kernels_[kv.first] = k;
            break;
        }
    }
    if (kernels_.empty()) return fail("none of the kernels of tl_kernels.meta was found in the code objects");
    ok_ = true;
    return true;
}

const AqlKernelInfo *AqlRuntime::find(const std::string &mangled) const {
    auto it = kernels_.find(mangled);
    return it == kernels_.end() ? nullptr : &it->second;
}

AqlProgram::~AqlProgram() {
    if (kernarg_dev) (void)hipFree(kernarg_dev);
}

int aql_program_from_graph(const AqlRuntime &rt, hipGraph_t graph, hipStream_t stream, AqlProgram &out, std::string &why) {
    if (!rt.ok()) {
        why = "AQL runtime not loaded: " + rt.why();
        return -1;
    }
    size_t n = 0;
    if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess || n == 0) {
        why = "hipGraphGetNodes failed or the graph is empty";
        return -1;
    }
    size_t n_roots = 0;
    if (hipGraphGetRootNodes(graph, nullptr, &n_roots) != hipSuccess || n_roots != 1) {
        why = "the captured step is not a single chain (" + std::to_string(n_roots) + " root nodes)";
        return -2;
    }
    hipGraphNode_t node = nullptr;
    n_roots = 1;
    if (hipGraphGetRootNodes(graph, &node, &n_roots) != hipSuccess || !node) {
        why = "hipGraphGetRootNodes failed";
        return -1;
    }
    constexpr size_t SLOT = 256;
    std::vector<char> host;
    std::vector<size_t> offsets;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType type;
        if (hipGraphNodeGetType(node, &type) != hipSuccess || type != hipGraphNodeTypeKernel) {
            why = "node " + std::to_string(i) + " of the captured step is not a kernel";
            return -2;
        }
        hipKernelNodeParams kp{};
        if (hipGraphKernelNodeGetParams(node, &kp) != hipSuccess || !kp.func) {
            why = "hipGraphKernelNodeGetParams failed";
            return -1;
        }
        const char *nm = hipKernelNameRefByPtr(kp.func, stream);
        if (!nm) {
            why = "no kernel name for node " + std::to_string(i);
            return -1;
        }
        const AqlKernelInfo *k = rt.find(nm);
        if (!k) {
            why = std::string("kernel not in the AQL code objects: ") + nm;
            return -2;
        }
        if (k->private_bytes != 0) {
            why = std::string("kernel uses scratch: ") + nm;
            return -2;
        }
        if (!kp.kernelParams && !k->args.empty()) {
            why = std::string("node without kernelParams: ") + nm;
            return -2;
        }
        const size_t off = host.size();
        host.resize(off + (k->kernarg_bytes + SLOT - 1) / SLOT * SLOT, 0);
        char *ka = host.data() + off;
        for (size_t a = 0; a < k->args.size(); ++a) memcpy(ka + k->args[a].first, kp.kernelParams[a], k->args[a].second);
        if (k->hidden_base >= 0) {  // code object v5 implicit arguments (tools/kernel_meta.py checks the offsets at build time)
            char *h = ka + k->hidden_base;
            const uint32_t bc[3] = {kp.gridDim.x, kp.gridDim.y, kp.gridDim.z};
            const uint16_t gs[3] = {(uint16_t)kp.blockDim.x, (uint16_t)kp.blockDim.y, (uint16_t)kp.blockDim.z};
            memcpy(h + 0, bc, 12);
            memcpy(h + 12, gs, 6);
            const uint16_t dims = 3;
            memcpy(h + 64, &dims, 2);  // remainders (+18) and global offsets (+40) stay zero
        }
        hsa_kernel_dispatch_packet_t p{};
        p.setup = 3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        p.workgroup_size_x = (uint16_t)kp.blockDim.x;
        p.workgroup_size_y = (uint16_t)kp.blockDim.y;
        p.workgroup_size_z = (uint16_t)kp.blockDim.z;
        p.grid_size_x = kp.gridDim.x * kp.blockDim.x;
        p.grid_size_y = kp.gridDim.y * kp.blockDim.y;
        p.grid_size_z = kp.gridDim.z * kp.blockDim.z;
        p.private_segment_size = 0;
        p.group_segment_size = k->group_static + kp.sharedMemBytes;
        p.kernel_object = k->object;
        out.packets.push_back(p);
        out.names.push_back(nm);
        offsets.push_back(off);
        if (i + 1 < n) {
            size_t n_dep = 0;
            if (hipGraphNodeGetDependentNodes(node, nullptr, &n_dep) != hipSuccess || n_dep != 1) {
                why = "the captured step is not a single chain (node " + std::to_string(i) + " has " + std::to_string(n_dep) + " dependents)";
                return -2;
            }
            hipGraphNode_t next = nullptr;
            if (hipGraphNodeGetDependentNodes(node, &next, &n_dep) != hipSuccess || !next) {
                why = "hipGraphNodeGetDependentNodes failed";
                return -1;
            }
            node = next;
        }
    }
    if (hipMalloc((void **)&out.kernarg_dev, host.size()) != hipSuccess) {
        why = "hipMalloc of the argument buffer failed";
        return -1;
    }
    out.kernarg_bytes = host.size();
    if (hipMemcpy(out.kernarg_dev, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) {
        why = "copy of the argument buffer failed";
        return -1;
    }
    for (size_t i = 0; i < out.packets.size(); ++i) out.packets[i].kernarg_address = out.kernarg_dev + offsets[i];
    return 0;
}

AqlQueue::~AqlQueue() {
    if (q_) hsa_queue_destroy(q_);
    if (done_.handle) hsa_signal_destroy(done_);
}

bool AqlQueue::create(const AqlRuntime &rt, std::string &why, uint32_t packets) {
    if (!rt.ok()) {
        why = "AQL runtime not loaded: " + rt.why();
        return false;
    }
    hsa_status_t s = hsa_queue_create(rt.agent(), packets, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q_);
    if (s != HSA_STATUS_SUCCESS) {
        why = "hsa_queue_create: " + hsa_err(s);
        q_ = nullptr;
        return false;
    }
    if ((s = hsa_signal_create(0, 0, nullptr, &done_)) != HSA_STATUS_SUCCESS) {
        why = "hsa_signal_create: " + hsa_err(s);
        return false;
    }
    return true;
}

bool AqlQueue::submit(const AqlProgram &p, const AqlFences &f, bool first, bool last, std::string &why) {
    if (!q_ || p.packets.empty()) {
        why = "AQL queue or program missing";
        return false;
    }
    const size_t n = p.packets.size();
    if (n > q_->size) {
        why = "program larger than the queue";
        return false;
    }
    if (last && pending_) {  // one signal: two carriers in flight would let the first completion pass for the second
        why = "AQL submission with a completion signal while the previous one is pending (wait() first)";
        return false;
    }
    const uint32_t mask = q_->size - 1;
    // (one producer: room is awaited BEFORE the write index moves, so that a refused submission leaves no reserved, never-written slots
    // -- invalid headers the packet processor would wait on for ever -- behind)
    const uint64_t at = hsa_queue_load_write_index_relaxed(q_);
    auto *ring = (hsa_kernel_dispatch_packet_t *)q_->base_address;
    const auto t0 = std::chrono::steady_clock::now();
    while (at + n - hsa_queue_load_read_index_scacquire(q_) > q_->size) {  // ring full: the device is several steps behind
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0) {
            why = "AQL queue stayed full for 10 s";
            return false;
        }
    }
    const uint64_t idx = hsa_queue_add_write_index_relaxed(q_, n);
    if (idx != at) {
        why = "AQL queue has a second producer";
        return false;
    }
    if (last) hsa_signal_store_relaxed(done_, 1);
    for (size_t i = 0; i < n; ++i) {
        hsa_kernel_dispatch_packet_t d = p.packets[i];
        const bool head = i == 0, tail = i + 1 == n;
        d.completion_signal.handle = (last && tail) ? done_.handle : 0;
        const int acq = head ? (first ? HSA_FENCE_SCOPE_SYSTEM : f.step_acquire) : f.inner_acquire;
        const int rel = tail ? (last ? HSA_FENCE_SCOPE_SYSTEM : f.step_release) : f.inner_release;
        const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                           (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        hsa_kernel_dispatch_packet_t *slot = ring + ((idx + i) & mask);
        memcpy((char *)slot + 4, (const char *)&d + 4, sizeof(d) - 4);
        __atomic_store_n((uint32_t *)slot, (uint32_t)header | ((uint32_t)d.setup << 16), __ATOMIC_RELEASE);
    }
    hsa_signal_store_screlease(q_->doorbell_signal, (hsa_signal_value_t)(idx + n - 1));
    if (last) {
        pending_ = true;
        unsignalled_ = false;  // the queue runs its packets in order (barrier bit): the signal of this one covers all before it
    } else {
        unsignalled_ = true;
    }
    return true;
}

// An empty barrier-AND packet (no dependencies, barrier bit set) behind everything in the ring, carrying the completion signal.
bool AqlQueue::submit_barrier(std::string &why) {
    if (!q_) {
        why = "AQL queue missing";
        return false;
    }
    const uint32_t mask = q_->size - 1;
    const uint64_t at = hsa_queue_load_write_index_relaxed(q_);
    const auto t0 = std::chrono::steady_clock::now();
    while (at + 1 - hsa_queue_load_read_index_scacquire(q_) > q_->size) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0) {
            why = "AQL queue stayed full for 10 s";
            return false;
        }
    }
    const uint64_t idx = hsa_queue_add_write_index_relaxed(q_, 1);
    if (idx != at) {
        why = "AQL queue has a second producer";
        return false;
    }
    hsa_signal_store_relaxed(done_, 1);
    hsa_barrier_and_packet_t b;
    memset(&b, 0, sizeof(b));
    b.completion_signal = done_;
    const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                       (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                       (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
    auto *slot = (hsa_barrier_and_packet_t *)q_->base_address + (idx & mask);
    static_assert(sizeof(hsa_barrier_and_packet_t) == sizeof(hsa_kernel_dispatch_packet_t), "one ring slot");
    memcpy((char *)slot + 4, (const char *)&b + 4, sizeof(b) - 4);
    __atomic_store_n((uint32_t *)slot, (uint32_t)header, __ATOMIC_RELEASE);
    hsa_signal_store_screlease(q_->doorbell_signal, (hsa_signal_value_t)idx);
    pending_ = true;
    unsignalled_ = false;
    return true;
}

bool AqlQueue::wait(double seconds, std::string &why) {
    const auto t0 = std::chrono::steady_clock::now();
    auto wait_signal = [&]() {
        // spin for the first 100 ms (a timed run of decode steps ends inside it: no wake-up latency on the metric's path), after that the
        // host thread sleeps on the signal's interrupt instead of burning a core per engine for the rest of a long call
        for (;;) {
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const hsa_wait_state_t how = waited < 0.1 ? HSA_WAIT_STATE_ACTIVE : HSA_WAIT_STATE_BLOCKED;
            if (hsa_signal_wait_scacquire(done_, HSA_SIGNAL_CONDITION_LT, 1, 2000000, how) < 1) break;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) {
                why = "AQL submission did not complete within " + std::to_string(seconds) + " s";
                return false;
            }
        }
        pending_ = false;
        return true;
    };
    // (one signal: an earlier carrier is waited out before the barrier packet arms it again)
    if (pending_ && !wait_signal()) return false;
    if (unsignalled_) {
        if (!submit_barrier(why)) return false;
        if (!wait_signal()) return false;
    }
    return true;
}

}  // namespace tl
