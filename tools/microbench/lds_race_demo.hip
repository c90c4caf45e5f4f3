// lds_race_demo.hip -- the demonstration object of tools/check_lds_dma_barriers.py (never linked into the product).
// Instantiates the f64 GEMM's k-loop as it was before round 4's fix (DBG = 5: the s_barrier reached with this trip's
// ds_reads still pending -- the compiler sinks their wait below the raw barrier) beside the shipped loop (DBG = 0),
// compiled with the flags of fit_kernels.hip.  The checker must flag the first and pass the second: a checker that
// cannot see the old bug in the old code is blind.
//   hipcc --offload-arch=gfx950 --cuda-device-only -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DTGP_DEBUG_KERNELS \
//         -I turbo_amd/csrc -c tools/microbench/lds_race_demo.hip -o lds_race_demo.co
#include "gemm64_glds.hpp"

namespace tgp {
template __global__ void gemm64_glds_kernel<KR_FULL, TM_LOWER, 3, 5>(GemmArgs);
template __global__ void gemm64_glds_kernel<KR_FULL, TM_LOWER, 3, 0>(GemmArgs);
}  // namespace tgp
