// doorbell.hpp -- completion of a SHORT call without hipStreamSynchronize (round 6).
//
// The everyday regime of the reference (tens of observations: demos/Branin-Hoo.ipynb; one evaluation of
// the hyper-parameter objective, turbo/modules/surrogates.py:313-318 -> _gpr.py:296-337, or one batched
// acquisition value + gradient of the gradient stage, turbo/modules/auxiliary_optimisers.py:69-112) is
// bound by what surrounds a 10-30 us kernel, not by the kernel: event records, a stream synchronisation,
// pageable copies.  A polled call has none of them.  Its last kernel ends by storing the call's sequence
// number into ONE word of coherent, device-mapped host memory -- after a system-scope fence in every
// thread that wrote results (which also live in mapped host memory) -- and the host spins on that word.
// PCIe keeps posted writes in order, so a host that sees the number sees the results.
//
//   single-workgroup kernels: fence, barrier, thread 0 rings;
//   multi-workgroup kernels:  the same per workgroup, then a ticket (device memory, agent scope); the
//                             workgroup that draws the last ticket resets the counter and rings.
// No workgroup ever WAITS for another one: nothing here can hang.
//
// The host side (tgp_api.hip, bell_wait) gives up after TGP_POLL_US and synchronises the stream instead, so
// a kernel that never rings costs time, not correctness; TGP_POLL_US=0 switches the polling off (the
// kernels then get a null word and the call ends in hipStreamSynchronize as before) -- the A/B switch.
// The kernels also leave wall_clock64() (constant 100 MHz) at their start and end beside the word: the
// call's device time without two event records.
#pragma once
#include <hip/hip_runtime.h>

namespace tgp {

struct Bell {
    unsigned long long *word;   // device view of [sequence number, start tick, end tick, -], or null: no bell
    unsigned long long seq;
    unsigned *ticket;           // device memory, zero between launches (multi-workgroup kernels only)
};

// first thing in the kernel, every thread may call it
__device__ __forceinline__ void bell_start(const Bell &b) {
    if (b.word && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) b.word[1] = wall_clock64();
}

// last thing in the kernel, EVERY thread of EVERY workgroup calls it (uniformly); total = workgroups in the launch
__device__ __forceinline__ void bell_ring(const Bell &b, unsigned total) {
    if (!b.word) return;
    __threadfence_system();     // this thread's result stores are out (device memory and mapped host memory)
    __syncthreads();
    if (threadIdx.x == 0) {
        bool last = true;
        if (total > 1) {
            const unsigned t = __hip_atomic_fetch_add(b.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            last = t == total - 1;
            if (last) __hip_atomic_store(b.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (last) {
            b.word[2] = wall_clock64();
            __hip_atomic_store(b.word, b.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace tgp
