// Driver of tests/test_host_backend.py::test_host_backend_under_sanitizers: the host backend
// (turbo_amd/csrc/host_backend.cpp) compiled with -fsanitize=address,undefined and run over small and
// ragged sizes, threads forced on (GPU sanitizers are not available on this pool: CPU build only).
#include <cstdio>
#include <vector>
#include <cmath>
#include "host_backend.hpp"
int main() {
    for (int N : {1, 5, 63, 64, 65, 130, 517}) {
        int D = 1 + N % 7, M = 37 + N % 50;
        std::vector<double> X(N * D), y(N), Xc(M * D), ls(D, 0.7);
        unsigned s = 123u + N;
        auto rnd = [&] { s = s * 1664525u + 1013904223u; return (s >> 8) / 16777216.0; };
        for (auto &v : X) v = rnd();
        for (int i = 0; i < N; ++i) y[i] = std::sin(3 * X[i * D]) + 0.01 * rnd();
        for (auto &v : Xc) v = rnd();
        tgp_host::HostGP g;
        double lml, ym, ys;
        int rc = g.fit(X.data(), N, D, y.data(), N % 4, 1.2, ls.data(), D, 1e-3, 1e-10, 1, &lml, &ym, &ys);
        std::vector<double> mu(M), sg(M), aq(M);
        double bv; int64_t bi, nc;
        rc |= g.set_candidates(Xc.data(), M);
        rc |= g.sweep(3, -1.0, -0.5, 0.01, mu.data(), sg.data(), aq.data(), &bv, &bi, &nc);
        int64_t need = 0;
        rc |= g.export_state(nullptr, 0, &need);
        std::vector<char> blob(need);
        rc |= g.export_state(blob.data(), need, &need);
        double lml2;
        rc |= g.import_state(blob.data(), need, &lml2);
        printf("N=%d rc=%d lml=%.6f lml2==lml %d best %lld %.4g\n", N, rc, lml, lml2 == lml, (long long)bi, bv);
    }
    return 0;
}
