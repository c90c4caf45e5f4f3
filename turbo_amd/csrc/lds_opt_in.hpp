// lds_opt_in.hpp -- per-device opt-in to more than 64 KB of dynamic LDS for one kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace tgp {
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: remember
// per (call site = kernel instantiation, device) so that one process may hold handles on several
// GPUs.  Lock-free; a lost race only repeats the (idempotent) call.
struct LdsOptIn {
    std::atomic<uint64_t> mask[4];   // devices 0..255
    hipError_t ensure(const void *fn, int device, size_t bytes) {
        const unsigned d = (unsigned)device & 255u;
        const uint64_t bit = 1ull << (d & 63u);
        if (mask[d >> 6].load(std::memory_order_acquire) & bit) return hipSuccess;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) mask[d >> 6].fetch_or(bit, std::memory_order_release);
        return e;
    }
};

}  // namespace tgp
