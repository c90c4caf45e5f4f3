// cu_mask_probe.hip -- which physical CUs (HW_REG_XCC_ID, HW_REG_HW_ID) a CU-masked HIP stream reaches, and where
// workgroup 0 of a launch lands (round 6: why no CU mask keeps the pivot workgroup off the background stream's CUs;
// csrc/fit_kernels.hip, profiles/r06_pivot_cu_stamps.txt).
//   hipcc --offload-arch=gfx950 -O2 -o tools/microbench/cu_mask_probe tools/microbench/cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <set>
#include <vector>
#include <tuple>

__global__ void probe(unsigned long long *out) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        out[blockIdx.x] = ((unsigned long long)xcc << 32) | hw | (1ull << 63);
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int run(hipStream_t st, int wgs, std::vector<unsigned long long> &h, unsigned long long *d) {
    h.assign((size_t)wgs, 0);
    CK(hipMemset(d, 0, wgs * sizeof(unsigned long long)));
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(64), 0, st, d);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}
static void summary(const char *name, const std::vector<unsigned long long> &h) {
    int per[16] = {0};
    std::set<std::tuple<int, int, int>> cus[16];
    for (unsigned long long v : h) {
        if (!(v >> 63)) continue;
        const int x = (int)((v >> 32) & 0xf), hw = (int)(v & 0xffffffff);
        cus[x].insert(std::make_tuple((hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15));
        per[x]++;
    }
    printf("%-34s first wg on xcc %d | CUs per xcc:", name, (int)((h[0] >> 32) & 0xf));
    for (int x = 0; x < 8; ++x) printf(" %2zu", cus[x].size());
    printf("\n");
}

int main() {
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs: %d\n", ncu);
    unsigned long long *d = nullptr;
    CK(hipMalloc((void **)&d, 16384 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h;
    hipStream_t plain;
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    for (int wgs : {1, 1, 1, 2, 48, 277, 330, 16384}) {
        if (run(plain, wgs, h, d)) return 1;
        char nm[64];
        snprintf(nm, sizeof nm, "unmasked stream, %d wgs", wgs);
        summary(nm, h);
    }
    const int nw = (ncu + 31) / 32;
    auto masked = [&](const char *name, std::vector<uint32_t> mask) -> int {
        hipStream_t st = nullptr;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
        if (e != hipSuccess) { printf("%-34s hipExtStreamCreateWithCUMask: %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); return 0; }
        if (run(st, 16384, h, d)) return 1;
        summary(name, h);
        if (run(st, 1, h, d)) return 1;
        summary("   ... the same stream, 1 wg", h);
        (void)hipStreamDestroy(st);
        return 0;
    };
    {   // round 5's mask: bits [0, 192)
        std::vector<uint32_t> m((size_t)nw, 0u);
        for (int i = 0; i < (ncu * 3) / 4; ++i) m[i >> 5] |= 1u << (i & 31);
        if (masked("bits [0, 3/4 ncu)", m)) return 1;
    }
    for (int r = 0; r < 8; ++r) {   // every bit but those with i mod 8 == r, 28 per remaining class
        std::vector<uint32_t> m((size_t)nw, 0u);
        for (int i = 0; i < ncu; ++i)
            if (i % 8 != r && i / 8 < 28) m[i >> 5] |= 1u << (i & 31);
        char nm[64];
        snprintf(nm, sizeof nm, "all but i mod 8 == %d, 28 each", r);
        if (masked(nm, m)) return 1;
    }
    {   // one 32-bit word cleared
        std::vector<uint32_t> m((size_t)nw, 0xffffffffu);
        m[0] = 0;
        if (masked("word 0 cleared", m)) return 1;
    }
    return 0;
}
