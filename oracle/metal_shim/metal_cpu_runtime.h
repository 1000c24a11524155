// TEST INFRASTRUCTURE (oracle/_ref): a CPU stand-in for the Metal execution model, just large enough to run the reference's
// `.metal` kernel sources (src/extensions_ref/src/*.metal) unmodified, as C++, on the host -- so that the numpy oracle can be
// held against the reference's OWN kernel code.  Nothing here is derived from the reference; it re-creates the pieces of the
// Metal Shading Language those files use:
//
//   * address-space keywords (`device`, `constant`, `thread`) are erased; `threadgroup` is handled by the build recipe
//     (kernel-scope `threadgroup T name[N]` arrays become function-static: threadgroups run one at a time, so one static
//     instance IS the threadgroup's memory);
//   * `half`, `bfloat16_t` / `bfloat`: 16-bit storage types with round-to-nearest-even conversion from float;
//   * `uint3` & co., `fast::` / `precise::` math (mapped to libm: Apple's fast-math approximations are NOT reproduced);
//   * the SIMT part: every thread of a threadgroup is a cooperative FIBER (ucontext) on one OS thread.  `simd_sum`, `simd_max`,
//     `simd_shuffle_xor` and `threadgroup_barrier` deposit the lane's value, yield to the scheduler until all live lanes of the
//     SIMD group (32 lanes) / threadgroup have arrived, then read the result -- the lock-step semantics the kernels rely on,
//     deterministic and race-free by construction.  Lanes that have returned no longer take part.
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <vector>

namespace metal_cpu {

constexpr int SIMD_WIDTH = 32;

struct Lane {
    ucontext_t ctx;
    std::vector<char> stack;
    bool done = false;
    int index = 0;
};

struct Rendezvous {  // one per SIMD group, one for the whole threadgroup
    int live = 0;     // lanes that have not returned
    int arrived = 0;
    unsigned generation = 0;
    double slot[2][1024];  // values of the current / previous generation by lane-in-scope (double holds float / int exactly)
    bool present[2][1024] = {};  // which lanes deposited in that generation (lanes that returned earlier do not take part)
    double result[2];
};

struct Threadgroup {
    std::vector<Lane> lanes;
    std::vector<Rendezvous> simd;
    Rendezvous all;
    ucontext_t scheduler;
    int current = -1;
    std::function<void(int)> body;
};

inline Threadgroup *&active() {
    static Threadgroup *tg = nullptr;
    return tg;
}

inline void yield_lane() {
    Threadgroup *tg = active();
    swapcontext(&tg->lanes[tg->current].ctx, &tg->scheduler);
}

inline void lane_entry() {
    Threadgroup *tg = active();
    const int me = tg->current;
    tg->body(me);
    tg->lanes[me].done = true;
    // a lane that returns leaves every rendezvous scope: waiters must not wait for it
    Rendezvous &s = tg->simd[me / SIMD_WIDTH];
    s.live--;
    tg->all.live--;
    swapcontext(&tg->lanes[me].ctx, &tg->scheduler);
}

// Runs `threads` lanes of ONE threadgroup to completion; body(lane_index) is the kernel call for that lane.
inline void run_threadgroup(int threads, const std::function<void(int)> &body, size_t stack_bytes = 256 * 1024) {
    Threadgroup tg;
    tg.body = body;
    tg.lanes.resize(threads);
    tg.simd.resize((threads + SIMD_WIDTH - 1) / SIMD_WIDTH);
    for (int i = 0; i < threads; ++i) tg.simd[i / SIMD_WIDTH].live++;
    tg.all.live = threads;
    Threadgroup *previous = active();
    active() = &tg;
    for (int i = 0; i < threads; ++i) {
        Lane &l = tg.lanes[i];
        l.index = i;
        l.stack.resize(stack_bytes);
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = l.stack.data();
        l.ctx.uc_stack.ss_size = l.stack.size();
        l.ctx.uc_link = &tg.scheduler;
        makecontext(&l.ctx, (void (*)())lane_entry, 0);
    }
    int remaining = threads;
    long idle_rounds = 0;
    while (remaining > 0) {
        int progressed = 0;
        for (int i = 0; i < threads; ++i) {
            if (tg.lanes[i].done) continue;
            tg.current = i;
            swapcontext(&tg.scheduler, &tg.lanes[i].ctx);
            if (tg.lanes[i].done) {
                remaining--;
                progressed++;
            }
        }
        // a round in which no lane finished is normal (lanes waiting at a rendezvous); a kernel whose lanes wait for one that
        // never arrives would spin forever: give up loudly instead
        idle_rounds = progressed ? 0 : idle_rounds + 1;
        if (idle_rounds > 50'000'000) {
            std::abort();
        }
    }
    active() = previous;
}

enum class Reduce { Sum, Max, Shuffle, Barrier };

// The lane deposits `value`, waits until every live lane of the scope has arrived, and gets the scope's result.
inline double rendezvous(Rendezvous &r, int lane_in_scope, double value, Reduce op, int scope_lanes, int partner = 0) {
    const unsigned gen = r.generation;
    const int buf = gen & 1;
    r.slot[buf][lane_in_scope] = value;
    r.present[buf][lane_in_scope] = true;
    r.arrived++;
    while (r.generation == gen) {
        if (r.arrived >= r.live) {  // last to arrive (or the others have returned meanwhile): publish and release
            double acc = 0.0;
            if (op == Reduce::Sum || op == Reduce::Max) {
                // the lanes' floats, combined in lane order in float (the hardware's combination order is not specified)
                float facc = op == Reduce::Max ? -std::numeric_limits<float>::infinity() : 0.0f;
                for (int i = 0; i < scope_lanes; ++i) {
                    if (!r.present[buf][i]) continue;
                    const float v = (float)r.slot[buf][i];
                    facc = op == Reduce::Max ? (v > facc ? v : facc) : facc + v;
                }
                acc = facc;
            }
            for (int i = 0; i < scope_lanes; ++i) r.present[buf ^ 1][i] = false;  // the other buffer is free again
            r.result[buf] = acc;
            r.arrived = 0;
            r.generation = gen + 1;
            break;
        }
        yield_lane();
    }
    if (op == Reduce::Shuffle) return r.slot[buf][partner];
    return r.result[buf];
}

inline int current_lane() { return active()->current; }

}  // namespace metal_cpu
