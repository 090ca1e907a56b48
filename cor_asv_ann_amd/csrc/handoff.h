// Hand-offs between the workgroups of ONE persistent launch (persist.hip, train_persist.hip): the waiting and the signalling side.
// MI355X: 8 XCDs with private L2s, per-CU L1s -- see the header of persist.hip for the protocol these implement.
#pragma once
#include "common.h"

namespace casv {

// A wait gives up after this many ticks of the 100 MHz wall clock (50 ms: a whole decode takes milliseconds; a hand-off
// microseconds).  Lost residency -- a compiler that changed the register count, a partitioned or shared GPU -- then costs
// one such wait per call, after which the host stops choosing the persistent path for a while (engine.hip).
constexpr unsigned long long PERSIST_WAIT_TICKS = 5ull * 1000ull * 1000ull;

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Dep { const unsigned* c; unsigned target; };

// One lane waits until every counter has reached its target.  No cache invalidation follows: the payload is read with loads
// that go past the L2 (load_sc1) -- an agent-scope acquire would empty this XCD's L2 of the weights as well, and the K loops
// would wait for memory instead of the L2.  A wait that gives up does not return (see below).
__device__ __forceinline__ bool wait_deps(const Dep d0, const Dep d1, const Dep d2, unsigned* abort_w, int* s_ok) {
    if (threadIdx.x == 0) {
        int good = 1;
        unsigned spins = 0;
        unsigned long long t_begin = 0;
        for (;;) {
            const bool ready = (!d0.c || ld_agent(d0.c) >= d0.target) && (!d1.c || ld_agent(d1.c) >= d1.target) &&
                               (!d2.c || ld_agent(d2.c) >= d2.target);
            if (ready) break;
            ++spins;
            if ((spins & 255u) == 0) {
                if (ld_agent(abort_w)) { good = 0; break; }
                const unsigned long long now = wall_clock64();
                if (!t_begin) t_begin = now;
                else if (now - t_begin > PERSIST_WAIT_TICKS) {
                    __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    good = 0;
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(1);
        }
        *s_ok = good;
    }
    __syncthreads();
    const int ok = __builtin_amdgcn_readfirstlane(*s_ok);      // (one word for the whole workgroup: a scalar, the branch on it uniform)
    __syncthreads();
    // A wait that gave up ends the wave here and now -- every wave of the workgroup takes this turn together, with nothing left in
    // flight.  The function therefore never returns false (the callers' `if (!wait_deps(..)) return;` folds away): a `return` out
    // of their loops would be laid out as a structured exit that shares blocks with the loop body, i.e. code paths on which
    // registers with hidden loads in flight meet instructions that use them -- never walked, but neither check_asm_loads.py nor a
    // reader could tell.
    if (!ok) asm volatile("s_waitcnt vmcnt(0)\n\ts_endpgm" ::: "memory");
    return true;
}

// Every wave has stored its share write-through; one lane signals for the workgroup.
__device__ __forceinline__ void publish(unsigned* counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace casv
