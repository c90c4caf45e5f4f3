// tuning.hpp -- EVERY environment switch of the library, in one table (round 5).
//
// The defaults are the fastest measured settings; the switches exist for A/B measurements and so that the tests
// can run kernel variants the size-based defaults would only pick at sizes the suite never uses
// (tests/test_gpu_parity.py::test_alternate_kernel_paths).  The table is read ONCE, at the first use in the
// process (two rows are marked "per call": they are looked up again at every call because tests flip them inside
// one process).  tgp_tuning() (include/turbogp.h) prints the table with the values in force, so documentation
// cannot drift from the code: DESIGN.md section 5 is generated from that print.
//
// Plain C++ (no HIP): host_backend.cpp and the ROCm-less libturbogp_host.so include it too.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

namespace tgp {

// X(kind, field, "ENV_NAME", default, "what it selects")      kind: INT | DBL | STR
#define TGP_TUNING_TABLE(X)                                                                                                      \
    /* ---- sweep (sweep_kernels.hip, tgp_api.hip) ---- */                                                                      \
    X(INT, chunk, "TGP_CHUNK", 0, "candidates per slab group (multiple of 1024; 0 = about 256 MiB of slab) [per call]")          \
    X(DBL, slab_gb, "TGP_SLAB_GB", 1.0, "GiB of cross-kernel slab per launch pair (groups of TGP_CHUNK inside it)")              \
    X(STR, tile, "TGP_TILE", "", "contraction tile: 128 | 256x128 | 256x256 (default: by launch size)")                         \
    X(INT, nbuf, "TGP_NBUF", 0, "LDS k-tile buffers of the 256x128 contraction: 2 | 3 (default: 3 for f64, 2 for f32)")          \
    X(STR, trmm, "TGP_TRMM", "", "reg = the register-staged contraction template instead of the direct-to-LDS kernels (A/B)")    \
    X(INT, bk, "TGP_BK", 0, "k-tile of the contraction (f32: 16 | 32 | 64, f64: 8 | 16 | 32); other than 128 bytes = template")  \
    X(INT, mfma16, "TGP_MFMA16", 0, "1 = v_mfma_f32_16x16x4 fragments in the 128-tile f32 contraction")                         \
    X(INT, ks_js, "TGP_KS_JS", 0, "splits of the training points over the cross-kernel grid (0 = up to 8)")                       \
    X(INT, sweep_zc, "TGP_SWEEP_ZC", 1, "0 = the sweep's result record by D2H copies + memset instead of mapped host memory")    \
    X(INT, mean_in_trmm, "TGP_MEAN", 1, "1 = posterior mean K*.alpha accumulated inside the contraction's full-k row tiles (f64 / f32); 0 = in the cross-kernel (round 4)") \
    X(INT, overlap, "TGP_OVERLAP", 2, "sweep started inside tgp_fit for the resident batch: 0 = never, 1 = scaling + first cross-kernel, 2 = + contraction row tiles whose rows of Linv are final") \
    X(INT, pre_tiles, "TGP_PRE_TILES", -1, "128-row tiles of the first launch pair contracted inside the fit in mode 2 (-1 = a quarter of the rows)") \
    X(INT, pre_cus, "TGP_PRE_CUS", -1, "CUs of the third stream, which carries the sweep's front inside a fit (-1 = three quarters of the device, 0 = unmasked)") \
    X(INT, pre_cu0, "TGP_PRE_CU0", 0, "first CU of that mask")                                                                 \
    X(INT, pre_lds_kb, "TGP_PRE_LDS_KB", 81, "KiB of LDS requested per workgroup of the early contraction (64 = two per CU, 81 = one)") \
    X(INT, mid, "TGP_MID", 1, "0 = 128 < N <= 256 down the general sweep instead of the one-launch kernel")                      \
    X(INT, mid_maxm, "TGP_MID_MAXM", 0, "largest batch that takes the one-launch sweep for 256 < N <= 512 (0 = never) [per call]") \
    X(INT, small, "TGP_SMALL", 1, "0 = N <= 128 down the blocked path instead of the one-workgroup kernels")                     \
    /* ---- fit (fit_kernels.hip, grad_kernels.hip) ---- */                                                                     \
    X(INT, ob, "TGP_OB", 0, "outer block of the Cholesky (multiple of 256; 0 = by size: all of Np <= 1024, else 256 / 512 / 1024)") \
    X(INT, panel, "TGP_PANEL", 5, "diagonal-block factorisation: 5 = variant D (LDS block, MFMA), 3 / 38 = C, 4 / 8 = B, 0 = round 1") \
    X(INT, panel_la, "TGP_PANEL_LA", 1, "0 = pivot factored in the panel launch instead of beside the previous update")          \
    X(INT, panel_fuse, "TGP_PANEL_FUSE", 1, "0 = two launches per panel instead of fused_panel_kernel")                          \
    X(INT, panel_fuse_tiles, "TGP_PANEL_FUSE_TILES", 384, "fused panel launches for outer blocks whose first update has at most this many tiles") \
    X(STR, inner, "TGP_INNER", "", "gemm64 = the in-block rank-64 update on the generic template")                               \
    X(STR, gemm64, "TGP_GEMM64", "", "reg = the f64 64x64 NT products on the register-staged template (A/B, bit-identical)")     \
    X(INT, trail64, "TGP_TRAIL64", 8192, "trailing updates of up to this many rows on 64x64 tiles (beyond: 128-tile kernel)")    \
    X(INT, merge64, "TGP_MERGE64", 2048, "inverse merges with a leading block up to this on 64x64 tiles")                        \
    X(INT, kinv64, "TGP_KINV64", 4096, "K^-1 = U U^T of the LML gradient on 64x64 tiles up to this Np")                          \
    X(INT, bginv, "TGP_BGINV", 1, "0 = the inverse level by level after the factorisation instead of behind the panel chain")   \
    X(INT, level64_fused, "TGP_LEVEL64_FUSED", 1, "0 = the 64 -> 128 level of the inverse as three launches (two products + transpose) instead of one") \
    X(INT, bginv_max, "TGP_BGINV_MAX", 9216, "largest Np whose inverse runs behind the chain")                                   \
    X(INT, bg_lease, "TGP_BG_LEASE", 1, "0 = a fit on a private stream never borrows the background stream (its inverse always in line)") \
    X(INT, bg_cus, "TGP_BG_CUS", -1, "CUs of the background stream (-1 = three quarters of the device, 0 = unmasked)")           \
    X(INT, bg_probe, "TGP_BG_PROBE", 1, "0 = skip the probe that the stream pair really overlaps")                               \
    X(INT, linv_zero, "TGP_LINV_ZERO", 0, "1 = every fit zero-fills Linv (default: only when it may hold stale rows)")           \
    X(STR, stamp_file, "TGP_STAMP_FILE", "", "debug: the panel chain's in-kernel time stamps are dumped to this file")           \
    /* ---- hyper-parameter fit ---- */                                                                                         \
    X(INT, hyper_wgs, "TGP_HYPER_WGS", 0, "1 = one workgroup per start in the one-launch hyper-parameter fit (default: 3 for 64 < N <= 128)") \
    X(INT, hyper_threads, "TGP_HYPER_THREADS", 0, "host threads of tgp_fit_lbfgsb (0 = by size: 4 to N = 1536, 3 to 8192, 1 beyond)") \
    /* ---- latency of the short calls ---- */                                                                                  \
    X(INT, poll_us, "TGP_POLL_US", 50000, "microseconds a short call (small fit, fit + gradient, acquisition gradient) spins on its doorbell before it synchronises the stream instead (0 = never poll: events + hipStreamSynchronize as in round 5)") \
    X(INT, small_live, "TGP_SMALL_LIVE", 1, "0 = the N <= 128 fit factors the identity padding of its 64-blocks too and fetches the targets a second time (round 5's body; same bytes)") \
    X(INT, small_fused, "TGP_SMALL_FUSED", 1, "0 = N <= 128 fit + LML gradient as two launches (round 5) instead of one")          \
    X(INT, small_query, "TGP_SMALL_QUERY", 1, "0 = tgp_acq_grad for N <= 128 down the general kernels instead of one workgroup per point") \
    X(INT, query_mfma, "TGP_QUERY_MFMA", 1, "0 = the two triangular products of tgp_acq_grad above N = 128 on the wave-per-row / split-column kernels instead of v_mfma_f64_16x16x4 (A/B; same values to rounding)") \
    /* ---- host backend ---- */                                                                                                \
    X(INT, host_threads, "TGP_HOST_THREADS", 0, "worker threads of the host backend (0 = hardware concurrency)")

struct Tuning {
#define TGP_TF_INT(f, d) int f = d;
#define TGP_TF_DBL(f, d) double f = d;
#define TGP_TF_STR(f, d) std::string f = d;
#define TGP_TF(kind, f, env, d, doc) TGP_TF_##kind(f, d)
    TGP_TUNING_TABLE(TGP_TF)
#undef TGP_TF
#undef TGP_TF_INT
#undef TGP_TF_DBL
#undef TGP_TF_STR
    // derived
    bool trmm_reg = false, gemm64_reg = false, inner_generic = false;

    Tuning() {
#define TGP_TP_INT(f, env) if (const char *v = getenv(env)) f = atoi(v);
#define TGP_TP_DBL(f, env) if (const char *v = getenv(env)) f = atof(v);
#define TGP_TP_STR(f, env) if (const char *v = getenv(env)) f = v;
#define TGP_TP(kind, f, env, d, doc) TGP_TP_##kind(f, env)
        TGP_TUNING_TABLE(TGP_TP)
#undef TGP_TP
#undef TGP_TP_INT
#undef TGP_TP_DBL
#undef TGP_TP_STR
        trmm_reg = trmm == "reg";
        gemm64_reg = gemm64 == "reg";
        inner_generic = inner == "gemm64";
    }

    // "NAME=value    # what it selects" per line, the values in force in this process
    std::string dump() const {
        std::string out;
        char buf[512];
#define TGP_TD_INT(f) snprintf(val, sizeof val, "%d", f);
#define TGP_TD_DBL(f) snprintf(val, sizeof val, "%g", f);
#define TGP_TD_STR(f) snprintf(val, sizeof val, "%s", f.c_str());
#define TGP_TD(kind, f, env, d, doc)                                    \
    {                                                                   \
        char val[256];                                                  \
        TGP_TD_##kind(f)                                                \
        snprintf(buf, sizeof buf, "%s=%s\t# %s\n", env, val, doc);      \
        out += buf;                                                     \
    }
        TGP_TUNING_TABLE(TGP_TD)
#undef TGP_TD
#undef TGP_TD_INT
#undef TGP_TD_DBL
#undef TGP_TD_STR
        return out;
    }
};

// the process's table, read at first use
inline const Tuning &tuning() {
    static const Tuning t;
    return t;
}

// NOT a switch of this library: the HIP runtime's own variable, reported by tgp_stream_status (0 = unset)
inline int runtime_hw_queues_env() { const char *v = getenv("GPU_MAX_HW_QUEUES"); return v ? atoi(v) : 0; }

// the two rows tests flip inside one process: looked up at every call
inline int tuning_chunk_now() { const char *v = getenv("TGP_CHUNK"); return v ? atoi(v) : 0; }
inline int tuning_mid_maxm_now() { const char *v = getenv("TGP_MID_MAXM"); return v ? atoi(v) : 0; }

}  // namespace tgp
