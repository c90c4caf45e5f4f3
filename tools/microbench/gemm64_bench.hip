// gemm64_bench -- the fit's f64 NT products alone on the chip (no second stream): the register-staged 64 x 64
// template (mfma_gemm.hpp) against gemm64_glds.hpp and the 128 x 128 direct-to-LDS kernel (gemm_nt_glds.hpp), on
// the shapes of the blocked Cholesky's rank-512 trailing updates (lower-triangular tile sets) and of the
// inverse's products.  Prints microseconds and TFLOP/s (nominal f64 MFMA peak 78.6, attainable 66.8) and the
// largest deviation between the kernels' results; then the ablations of gemm64_glds (no MFMA / no DMA / every
// tile fetching the same operands).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../turbo_amd/csrc gemm64_bench.hip -o gemm64_bench && ./gemm64_bench [N]
// (profiles/r04_gemm64_ksplit_experiment.txt is this program at the commit that still had the k-split inside
// the workgroup: KS = 2 / 4 columns.)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gemm64_glds.hpp"
#include "gemm_nt_glds.hpp"
#include "mfma_gemm.hpp"

using namespace tgp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int KR, int TMAP>
static hipError_t launch_old(hipStream_t s, const GemmArgs &g, int nblocks, int batch) {
    auto kern = mfma_gemm_kernel<double, 64, 64, 16, true, KR, TMAP, EP_STORE>;
    constexpr size_t lds = gemm_lds_bytes<double, 64, 64, 16>();
    static LdsOptIn opt_in;
    if (hipError_t e = opt_in.ensure(reinterpret_cast<const void *>(kern), 0, lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nblocks, 1, batch), dim3(256), lds, s, g);
    return hipGetLastError();
}

template <typename F>
static float time_us(F f, int reps, double *C, const double *C0, size_t bytes) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemcpy(C, C0, bytes, hipMemcpyDeviceToDevice));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        CK(f());
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best * 1e3f;
}

int main(int argc, char **argv) {
    const int Np = argc > 1 ? atoi(argv[1]) : 4096, OB = 512;
    const long NN = (long)Np * Np;
    std::vector<double> h((size_t)NN);
    srand(1);
    for (long i = 0; i < NN; ++i) h[(size_t)i] = (rand() / (double)RAND_MAX - 0.5) * 0.1;
    double *K0, *K1, *K2;
    CK(hipMalloc(&K0, NN * 8)); CK(hipMalloc(&K1, NN * 8)); CK(hipMalloc(&K2, NN * 8));
    CK(hipMemcpy(K0, h.data(), NN * 8, hipMemcpyHostToDevice));
    std::vector<double> r1((size_t)NN), r2((size_t)NN);
    auto maxdiff = [&](double *a, double *b) {
        CK(hipMemcpy(r1.data(), a, NN * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(r2.data(), b, NN * 8, hipMemcpyDeviceToHost));
        double d = 0.0;
        for (long i = 0; i < NN; ++i) d = fmax(d, fabs(r1[(size_t)i] - r2[(size_t)i]));
        return d;
    };
    printf("trailing updates of N = %d, OB = %d (TM_LOWER, K = %d): 64-tiles | template us (TF) | gemm64_glds NBUF=3 | NBUF=4 | NBUF=2 | 128-tile glds | max|glds - template|\n", Np, OB, OB);
    double tot[5] = {0, 0, 0, 0, 0};
    for (int O = 0; O + OB < Np; O += OB) {
        const int R = Np - O - OB, nt = R / 64, nb = nt * (nt + 1) / 2;
        auto args = [&](double *K) {
            GemmArgs g{};
            g.A = K + (long)(O + OB) * Np + O; g.lda = Np; g.B = g.A; g.ldb = Np;
            g.C = K + (long)(O + OB) * Np + (O + OB); g.ldc = Np;
            g.ntm = g.ntn = nt; g.K = OB; g.alpha = -1.0; g.beta = 1.0;
            return g;
        };
        const double flops = (double)nb * 64 * 64 * OB * 2;
        float t[5];
        t[0] = time_us([&] { return launch_old<KR_FULL, TM_LOWER>(0, args(K1), nb, 1); }, 5, K1, K0, NN * 8);
        t[1] = time_us([&] { return launch_gemm64_glds<KR_FULL, TM_LOWER, 3>(0, 0, args(K2), nb, 1); }, 5, K2, K0, NN * 8);
        const double d = maxdiff(K1, K2);
        t[2] = time_us([&] { return launch_gemm64_glds<KR_FULL, TM_LOWER, 4>(0, 0, args(K2), nb, 1); }, 5, K2, K0, NN * 8);
        t[3] = time_us([&] { return launch_gemm64_glds<KR_FULL, TM_LOWER, 2>(0, 0, args(K2), nb, 1); }, 5, K2, K0, NN * 8);
        t[4] = 0.f;
        if (R % 128 == 0) {
            const int n128 = R / 128;
            t[4] = time_us([&] {
                GemmNtArgs g{};
                g.A = K2 + (long)(O + OB) * Np + O; g.lda = Np; g.B = g.A; g.ldb = Np;
                g.C = K2 + (long)(O + OB) * Np + (O + OB); g.ldc = Np; g.Ct = nullptr;
                g.ntm = g.ntn = n128; g.K = OB; g.alpha = -1.0; g.beta = 1.0;
                return launch_gemm_nt_glds<double, KN_FULL, TM_LOWER>(0, 0, g, n128 * (n128 + 1) / 2, 1);
            }, 5, K2, K0, NN * 8);
        }
        printf("  O=%5d tiles %5d ", O, nb);
        for (int i = 0; i < 5; ++i) { printf("  %7.1f (%5.1f)", t[i], t[i] > 0 ? flops / t[i] * 1e-6 : 0.0); tot[i] += t[i]; }
        printf("   %.2e\n", d);
    }
    printf("  sum                ");
    for (int i = 0; i < 5; ++i) printf("  %7.1f        ", tot[i]);
    printf("\n");
    printf("ablations of gemm64_glds<NBUF=3> on the first trailing update: all tiles fetch tile (0,0)'s operands | no MFMAs | no DMA\n");
    {
        const int O = 0, R = Np - O - OB, nt = R / 64, nb = nt * (nt + 1) / 2;
        auto args = [&](double *K) {
            GemmArgs g{};
            g.A = K + (long)(O + OB) * Np + O; g.lda = Np; g.B = g.A; g.ldb = Np;
            g.C = K + (long)(O + OB) * Np + (O + OB); g.ldc = Np;
            g.ntm = g.ntn = nt; g.K = OB; g.alpha = -1.0; g.beta = 1.0;
            return g;
        };
        float a1 = time_us([&] { return launch_gemm64_glds<KR_FULL, TM_LOWER, 3, 1>(0, 0, args(K2), nb, 1); }, 5, K2, K0, NN * 8);
        float a2 = time_us([&] { return launch_gemm64_glds<KR_FULL, TM_LOWER, 3, 2>(0, 0, args(K2), nb, 1); }, 5, K2, K0, NN * 8);
        float a3 = time_us([&] { return launch_gemm64_glds<KR_FULL, TM_LOWER, 3, 3>(0, 0, args(K2), nb, 1); }, 5, K2, K0, NN * 8);
        printf("  %7.1f | %7.1f | %7.1f\n", a1, a2, a3);
    }
    printf("small trailing updates (the launch's floor): tiles | K = 0 (launch + epilogue only) | full | no MFMAs | no DMA | K = 256 | K = 128\n");
    for (int O : {Np - 2 * OB, Np - 3 * OB, Np - 4 * OB}) {
        if (O < 0) continue;
        const int R = Np - O - OB, nt = R / 64, nb = nt * (nt + 1) / 2;
        auto args = [&](double *K, int kk) {
            GemmArgs g{};
            g.A = K + (long)(O + OB) * Np + O; g.lda = Np; g.B = g.A; g.ldb = Np;
            g.C = K + (long)(O + OB) * Np + (O + OB); g.ldc = Np;
            g.ntm = g.ntn = nt; g.K = kk; g.alpha = -1.0; g.beta = 1.0;
            return g;
        };
        float a0 = time_us([&] { return launch_gemm64_glds<KR_FULL, TM_LOWER, 3>(0, 0, args(K2, 0), nb, 1); }, 7, K2, K0, NN * 8);
        float a1 = time_us([&] { return launch_gemm64_glds<KR_FULL, TM_LOWER, 3>(0, 0, args(K2, OB), nb, 1); }, 7, K2, K0, NN * 8);
        float a2 = time_us([&] { return launch_gemm64_glds<KR_FULL, TM_LOWER, 3, 2>(0, 0, args(K2, OB), nb, 1); }, 7, K2, K0, NN * 8);
        float a3 = time_us([&] { return launch_gemm64_glds<KR_FULL, TM_LOWER, 3, 3>(0, 0, args(K2, OB), nb, 1); }, 7, K2, K0, NN * 8);
        float a4 = time_us([&] { return launch_gemm64_glds<KR_FULL, TM_LOWER, 3>(0, 0, args(K2, 256), nb, 1); }, 7, K2, K0, NN * 8);
        float a5 = time_us([&] { return launch_gemm64_glds<KR_FULL, TM_LOWER, 3>(0, 0, args(K2, 128), nb, 1); }, 7, K2, K0, NN * 8);
        printf("  tiles %5d   %7.1f | %7.1f | %7.1f | %7.1f | %7.1f | %7.1f\n", nb, a0, a1, a2, a3, a4, a5);
    }
    // the inverse's products: T^T = U11 * L21^T (KR_UPPER_A, K = a), the shape of merge_t in fit_kernels.hip
    printf("inverse merges (KR_UPPER_A, TM_FULL): a x b | 64-tiles | template us (TF) | gemm64_glds | max|diff|\n");
    const int shapes[][2] = {{128, 128}, {256, 256}, {512, 512}, {1024, 1024}, {2048, 2048}, {512, 3584}, {512, 7680}, {2048, 512}, {3584, 512}};
    for (auto &sh : shapes) {
        const int a = sh[0], b = sh[1];
        if (a + b > Np) continue;
        auto targs = [&](double *K) {
            GemmArgs g{};
            g.A = K0; g.lda = Np; g.B = K0 + (long)a * Np; g.ldb = Np;
            g.C = K + a; g.ldc = Np; g.ntm = a / 64; g.ntn = b / 64; g.K = a; g.alpha = 1.0; g.beta = 0.0;
            return g;
        };
        const int nb = (a / 64) * (b / 64);
        const double flops = (double)b * a * a;   // upper-triangular A: half of 2 a a b
        float t_old = time_us([&] { return launch_old<KR_UPPER_A, TM_FULL>(0, targs(K1), nb, 1); }, 5, K1, K0, NN * 8);
        float t1 = time_us([&] { return launch_gemm64_glds<KR_UPPER_A, TM_FULL, 3>(0, 0, targs(K2), nb, 1); }, 5, K2, K0, NN * 8);
        printf("  %4d x %4d tiles %5d   %7.1f (%5.1f)  %7.1f (%5.1f)   %.2e\n", a, b, nb, t_old, flops / t_old * 1e-6, t1,
               flops / t1 * 1e-6, maxdiff(K1, K2));
    }
    return 0;
}
