// zbatch.hpp -- lock-step batches of fits (round 6): the starts of a hyper-parameter fit evaluated by ONE chain of
// launches instead of a chain per start (turbo/modules/surrogates.py:313-324 -> sklearn _gpr.py:296-337: the restarts
// of the optimiser are independent; each evaluation is a fit + LML gradient, _gpr.py:584-650).
//
// The members of a batch keep their device buffers in equal SLOTS of one arena and their mapped staging in equal slots
// of one host block, so that member b's copy of ANY buffer of the lead (member 0 of the arena) is a fixed number of
// bytes away: a kernel of the fit takes the lead's pointers plus this table and shifts every pointer by dev[b] (pin[b]
// for pointers into the staging block), b = blockIdx.z (blockIdx.y in the GEMM kernels, whose z is taken).  A launch
// that is not a batch passes a zeroed table: b = 0, every shift 0, the by-value scalars as before.
#pragma once

namespace tgp {

constexpr int ZMAX = 4;   // members of a batch (= the worker pool's size, tgp_api.hip MAX_WORKERS)

struct ZBatch {
    long dev[ZMAX];        // bytes from the lead's device buffers to member b's
    long pin[ZMAX];        // ... from the lead's mapped staging to member b's
    double hp[ZMAX][4];    // member b's constant, noise, jitter, pivot threshold (the by-value scalars of a single fit)
    int n;                 // members in this launch; 0 = not a batch
};

#ifdef __HIPCC__
template <typename P>
__device__ __forceinline__ P *zshift(P *p, long bytes) {
    return reinterpret_cast<P *>(reinterpret_cast<unsigned long long>(p) + (unsigned long long)bytes);
}
#endif

}  // namespace tgp
