// gemm64_vs_rocblas -- an OUTSIDE yardstick for the fit's f64 GEMM (tools only: rocBLAS is never linked into the
// product).  north_star allows "rocBLAS trailing-update GEMM on MFMA"; the library hand-writes gemm64_glds.hpp instead.
// This program runs both on exactly the shapes the blocked Cholesky of N = 4096 (outer block 512) and N = 8192 (outer
// block 1024) issues -- the rank-OB trailing updates  C[lower] -= A A^T  (7260 / 1596 / 528 / 36 ... 64 x 64 tiles) --
// and on the inverse's merge products  T^T = U11 L21^T  (U11 upper triangular), alone on the chip, and prints
// microseconds and TFLOP/s of the ALGORITHMIC flops (the lower tile set; the triangular half) for each:
//     gemm64_glds   the product's kernel
//     dsyrk         rocblas_dsyrk on the same operands in place (column-major view: uplo = upper, trans = T)
//     dgemm         rocblas_dgemm computing the FULL square (twice the flops; priced at the algorithmic half)
//     dtrmm         rocblas_dtrmm for the merge (in place on a copy of L21^T)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../turbo_amd/csrc gemm64_vs_rocblas.hip -o gemm64_vs_rocblas -lrocblas
//   ./gemm64_vs_rocblas > profiles/r05_gemm64_vs_rocblas.txt
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gemm64_glds.hpp"
#include "gemm_nt_glds.hpp"

using namespace tgp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define RB(x) do { rocblas_status s_ = (x); if (s_ != rocblas_status_success) { printf("rocBLAS status %d at %s:%d\n", (int)s_, __FILE__, __LINE__); exit(1); } } while (0)

template <typename F>
static float time_us(F f, int reps, double *C, const double *C0, size_t bytes) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipMemcpy(C, C0, bytes, hipMemcpyDeviceToDevice));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        f();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best * 1e3f;
}

static void run(rocblas_handle hb, int Np, int OB) {
    const long NN = (long)Np * Np;
    std::vector<double> h((size_t)NN);
    srand(1);
    for (long i = 0; i < NN; ++i) h[(size_t)i] = (rand() / (double)RAND_MAX - 0.5) * 0.1;
    double *K0, *K1, *K2;
    CK(hipMalloc(&K0, NN * 8)); CK(hipMalloc(&K1, NN * 8)); CK(hipMalloc(&K2, NN * 8));
    CK(hipMemcpy(K0, h.data(), NN * 8, hipMemcpyHostToDevice));
    std::vector<double> r1((size_t)NN), r2((size_t)NN);
    printf("== N = %d, outer block %d: trailing updates  C[lower] -= A A^T  (rank %d)\n", Np, OB, OB);
    printf("   rows  64-tiles | gemm64_glds us (TF) | 128-tile glds us (TF) | rocblas_dsyrk us (TF) | rocblas_dgemm full square us (TF of the lower half) | max |dsyrk - gemm64_glds| on the lower triangle\n");
    double tot[4] = {0, 0, 0, 0};
    for (int O = 0; O + OB < Np; O += OB) {
        const int R = Np - O - OB, nt = R / 64, nb = nt * (nt + 1) / 2;
        const double flops = (double)nb * 64 * 64 * OB * 2;
        auto Aof = [&](double *K) { return K + (long)(O + OB) * Np + O; };
        auto Cof = [&](double *K) { return K + (long)(O + OB) * Np + (O + OB); };
        float t[4] = {0, 0, 0, 0};
        t[0] = time_us([&] {
            GemmArgs g{};
            g.A = Aof(K1); g.lda = Np; g.B = g.A; g.ldb = Np; g.C = Cof(K1); g.ldc = Np;
            g.ntm = g.ntn = nt; g.K = OB; g.alpha = -1.0; g.beta = 1.0;
            CK((launch_gemm64_glds<KR_FULL, TM_LOWER, 3>(0, 0, g, nb, 1)));
        }, 7, K1, K0, NN * 8);
        CK(hipMemcpy(r1.data(), K1, NN * 8, hipMemcpyDeviceToHost));
        if (R % 128 == 0) {
            const int n128 = R / 128;
            t[1] = time_us([&] {
                GemmNtArgs g{};
                g.A = Aof(K2); g.lda = Np; g.B = g.A; g.ldb = Np; g.C = Cof(K2); g.ldc = Np; g.Ct = nullptr;
                g.ntm = g.ntn = n128; g.K = OB; g.alpha = -1.0; g.beta = 1.0;
                CK((launch_gemm_nt_glds<double, KN_FULL, TM_LOWER>(0, 0, g, n128 * (n128 + 1) / 2, 1)));
            }, 7, K2, K0, NN * 8);
        }
        const double m1 = -1.0, p1 = 1.0;
        // row-major A (R x OB, lda Np) is the column-major OB x R matrix: C = -A^T A on its upper triangle = the row-major lower one
        t[2] = time_us([&] { RB(rocblas_dsyrk(hb, rocblas_fill_upper, rocblas_operation_transpose, R, OB, &m1, Aof(K2), Np, &p1, Cof(K2), Np)); },
                       7, K2, K0, NN * 8);
        CK(hipMemcpy(r2.data(), K2, NN * 8, hipMemcpyDeviceToHost));
        double d = 0.0;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j <= i; ++j) {
                const size_t idx = (size_t)(O + OB + i) * Np + (O + OB + j);
                d = fmax(d, fabs(r1[idx] - r2[idx]));
            }
        t[3] = time_us([&] { RB(rocblas_dgemm(hb, rocblas_operation_transpose, rocblas_operation_none, R, R, OB, &m1, Aof(K2), Np, Aof(K2), Np, &p1, Cof(K2), Np)); },
                       7, K2, K0, NN * 8);
        printf("  %5d  %5d  ", R, nb);
        for (int i = 0; i < 4; ++i) { printf("  %8.1f (%5.1f)", t[i], t[i] > 0 ? flops / t[i] * 1e-6 : 0.0); tot[i] += t[i]; }
        printf("   %.2e\n", d);
    }
    printf("  sum           ");
    for (int i = 0; i < 4; ++i) printf("  %8.1f        ", tot[i]);
    printf("\n");
    // the inverse's merge: T^T (a x b) = U11 (a x a, upper triangular) * L21^T  with L21 (b x a) row-major -- merge_t of fit_kernels.hip
    printf("== inverse merges  T^T = U11 L21^T  (U11 upper triangular a x a, L21 b x a): a x b | 64-tiles | gemm64_glds us (TF) | rocblas_dtrmm us (TF) | rocblas_dgemm (dense U11) us (TF of the triangular half)\n");
    const int shapes[][2] = {{256, 256}, {512, 512}, {1024, 1024}, {2048, 2048}, {512, 3584}, {1024, 7168}, {2048, 512}, {3584, 512}};
    for (auto &sh : shapes) {
        const int a = sh[0], b = sh[1];
        if (a + b > Np) continue;
        const int nb = (a / 64) * (b / 64);
        const double flops = (double)b * a * a;   // triangular A: half of 2 a a b
        const double one = 1.0, zero = 0.0;
        float t0 = time_us([&] {
            GemmArgs g{};
            g.A = K0; g.lda = Np; g.B = K0 + (long)a * Np; g.ldb = Np;
            g.C = K1 + a; g.ldc = Np; g.ntm = a / 64; g.ntn = b / 64; g.K = a; g.alpha = 1.0; g.beta = 0.0;
            CK((launch_gemm64_glds<KR_UPPER_A, TM_FULL, 3>(0, 0, g, nb, 1)));
        }, 7, K1, K0, NN * 8);
        // dtrmm in place: B := B * op(U11) in the column-major view.  Row-major T^T (a x b) = U11 L21^T; column-major view of
        // row-major L21 (b x a, ld Np) is L21^T (a x b): B_cm = L21^T, and T^T_rowmajor = (T)_cm ... we time the equivalent
        // product  B_cm := U11_cm-view applied from the left on the a x b block: side = left, the row-major upper U11 is
        // the column-major LOWER U11^T, op = transpose.
        float t1 = time_us([&] { RB(rocblas_dtrmm(hb, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_non_unit,
                                                  a, b, &one, K0, Np, K2 + (long)a * Np, Np, K2 + (long)a * Np, Np)); },
                         7, K2, K0, NN * 8);
        float t2 = time_us([&] { RB(rocblas_dgemm(hb, rocblas_operation_transpose, rocblas_operation_none, a, b, a, &one, K0, Np, K0 + (long)a * Np, Np,
                                                  &zero, K2 + a, Np)); },
                         7, K2, K0, NN * 8);
        printf("  %4d x %4d  %5d    %8.1f (%5.1f)   %8.1f (%5.1f)   %8.1f (%5.1f)\n", a, b, nb, t0, flops / t0 * 1e-6, t1, flops / t1 * 1e-6, t2,
               flops / t2 * 1e-6);
    }
    CK(hipFree(K0)); CK(hipFree(K1)); CK(hipFree(K2));
}

int main() {
    rocblas_handle hb;
    RB(rocblas_create_handle(&hb));
    RB(rocblas_set_stream(hb, 0));
    printf("f64 MFMA: nominal peak 78.6 TFLOP/s, attainable (register-only microbenchmark) 66.8\n");
    run(hb, 4096, 512);
    run(hb, 8192, 1024);
    RB(rocblas_destroy_handle(hb));
    return 0;
}
