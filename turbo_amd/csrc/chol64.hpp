// chol64.hpp -- Cholesky of one 64 x 64 diagonal block AND the inverse of its factor, by one
// 256-thread workgroup with the block in registers (sklearn _gpr.py:349 -> LAPACK dpotrf, here the
// unblocked kernel under the blocked factorisation of fit_kernels.hip and the whole factorisation
// of the small-problem path).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "mfma_gemm.hpp"
#include "tgp_internal.hpp"

namespace tgp {

#ifndef TGP_STAMP
#define TGP_STAMP(slot)
#endif

// Diagonal block: Cholesky of a 64x64 block and the inverse of its factor, one workgroup.
// Thread (tc = tid>>4, tr = tid&15) keeps the 4x4 sub-block rows 4tr.., cols 4tc.. in registers.
// Right-looking, one barrier per column: the 16 lanes that own column c publish it through a
// double-buffered LDS vector, every thread derives 1/sqrt(pivot) itself (v_rsq_f64 + two
// Newton steps; the pivot chain, not arithmetic, bounds this kernel) and applies the rank-1
// update to its registers.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double rsqrt_newton(double x) {
    // v_rsq_f64 is good to ~5e-8 (measured); one third-order step y (1 + e/2 + 3e^2/8),
    // e = 1 - x y^2, takes it to < 2e-16 with a 5-op dependent chain (two Newton steps need 6)
    const double y = __builtin_amdgcn_rsq(x);
    const double t = x * y;
    const double e = fma(-t, y, 1.0);
    const double p = fma(0.375, e, 0.5);
    return fma(y, e * p, y);
}

// ---- diagonal-block factorisation, variant A: FOUR columns per barrier -----------------------
// For column group g (columns c..c+3, c = 4g) the 16 lanes that own those columns publish them
// (unscaled) and the 16 lanes that own rows c..c+3 of the running inverse publish those rows;
// after ONE barrier every thread factors the 4x4 diagonal block itself, solves its own 4 rows /
// 4 columns against it and applies a rank-4 update to its registers.  The inverse X = L_kk^-1
// rides along (outer-product forward substitution on the identity): rows c..c+3 of X become
// final, the rows below get the same rank-4 update with the same columns of L.
__device__ __forceinline__ void factor64_steps4(double (&a)[4][4], double (&x)[4][4], double *panel_lds,
                                                int o, int *__restrict__ flag, double tiny) {
    double (*colbuf)[4][NB] = reinterpret_cast<double (*)[4][NB]>(panel_lds);             // [slot][column in group][row]
    double (*xbuf)[4][NB] = reinterpret_cast<double (*)[4][NB]>(panel_lds + 2 * 4 * NB);   // [slot][row in group][column]
    const int tid = threadIdx.x;
    const int tc = tid >> 4, tr = tid & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // holds tc = 4*wave .. 4*wave+3
#pragma unroll 1
    for (int g = 0; g < NB / 4; ++g) {
        const int pb = g & 1, c = 4 * g;
        if (tc == g) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) colbuf[pb][m][4 * tr + i] = a[i][m];
        }
        if (tr == g) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) xbuf[pb][k][4 * tc + j] = x[k][j];
        }
        __syncthreads();
        // ---- 4x4 diagonal block: d[k][m] = A[c+k][c+m], k >= m ----
        double d[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k][m] = colbuf[pb][m][c + k];
        double rs[4], L[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            double piv = d[m][m];
#pragma unroll
            for (int q = 0; q < m; ++q) piv = fma(-L[m][q], L[m][q], piv);
            // LAPACK dpotrf stops at a pivot <= 0 (scipy.linalg.cholesky -> LinAlgError,
            // _gpr.py:348-358).  A pivot that has lost every significant digit (< 8 eps of the
            // diagonal) is reported the same way: its sign is rounding noise.
            if (!(piv > tiny) || !isfinite(piv)) {
                if (tid == 0 && *flag == 0) *flag = o + c + m + 1;
                piv = 1.0;
            }
            rs[m] = rsqrt_newton(piv);
            L[m][m] = piv * rs[m];
#pragma unroll
            for (int k = m + 1; k < 4; ++k) {
                double v = d[k][m];
#pragma unroll
                for (int q = 0; q < m; ++q) v = fma(-L[k][q], L[m][q], v);
                L[k][m] = v * rs[m];
            }
        }
        // ---- own rows and own columns against the block: y = v * L_dd^-T ----
        double lrow[4][4], lcol[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                double v = colbuf[pb][m][4 * tr + i];
                double w = colbuf[pb][m][4 * tc + i];
#pragma unroll
                for (int q = 0; q < m; ++q) {
                    v = fma(-lrow[i][q], L[m][q], v);
                    w = fma(-lcol[i][q], L[m][q], w);
                }
                lrow[i][m] = v * rs[m];
                lcol[i][m] = w * rs[m];
            }
        }
        // ---- rows c..c+3 of the inverse: xr[k][j] for this thread's columns ----
        double xr[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double v = xbuf[pb][k][4 * tc + j];
#pragma unroll
                for (int q = 0; q < k; ++q) v = fma(-L[k][q], xr[q][j], v);
                xr[k][j] = v * rs[k];
            }
        }
        // ---- finalise the owners' entries ----
        const bool below = tr > g;   // rows below the block
        if (tc == g) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const double fin = (tr == g) ? ((i >= m) ? L[i][m] : a[i][m]) : lrow[i][m];
                    a[i][m] = (tr >= g) ? fin : a[i][m];
                }
        }
        if (tr == g) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) x[k][j] = xr[k][j];
        }
        // ---- rank-4 updates of the rows below the block ----
        // A: only column groups right of g still change; X: only column groups up to g are non-zero.
        // Both tests are wave-uniform on the wave's four column groups.
        const double rmask = below ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m) lrow[i][m] *= rmask;
        if (4 * wave + 3 > g) {
            const double cmask = (tc > g) ? 1.0 : 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < 4; ++m) lcol[j][m] *= cmask;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < 4; ++m) a[i][j] = fma(-lrow[i][m], lcol[j][m], a[i][j]);
        }
        if (4 * wave <= g) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) x[i][j] = fma(-lrow[i][k], xr[k][j], x[i][j]);
        }
    }
}

// ---- diagonal-block factorisation, variant B: GW = 4 or 8 columns per barrier, no thread does
// the A-side and the X-side work both -----------------------------------------------------------
// Column group g = columns c..c+GW-1 (c = GW g) = the thread columns tc with tc / TPG == g
// (TPG = GW / 4).  Their owners publish the raw columns, the owners of rows c..c+GW-1 of the
// running inverse X publish those rows; after ONE barrier every thread factors the GW x GW pivot
// block itself and runs TWO forward substitutions against it, both of the form  L_dd R = B:
//     rowsT[m][i]  from B = raw columns at this thread's own ROWS  -> L[4tr+i][c+m]
//     rhs[m][j]    from B = raw columns at this thread's own COLUMN indices (tc right of the
//                  group: the solved panel entries L[4tc+j][c+m] the rank-GW update of A needs)
//                  or  B = rows c.. of X (tc up to the group: the finished rows of the inverse,
//                  which the rank-GW update of X needs)
// A thread updates A or X, never both: A only changes right of the group, X below the block is
// non-zero only up to the group.  So one solve + one rank-GW update per thread serve both.
// The kernel is bound by VALU issue from ONE wave per SIMD (about 8 cycles per f64 operation),
// i.e. by the operation count per thread: 4800 in the round-1 form (variant A), 5200 with
// GW = 8, 3200 with GW = 4 -- measured 30.5 / 28.3 / ... us per panel launch at N = 4096.
template <int GW>
__device__ __forceinline__ void factor64_steps(double (&a)[4][4], double (&x)[4][4], double *panel_lds,
                                               int o, int *__restrict__ flag, double tiny) {
    constexpr int TPG = GW / 4;                               // thread columns (rows) per group
    double (*colbuf)[GW][NB] = reinterpret_cast<double (*)[GW][NB]>(panel_lds);                // [slot][column in group][row]
    double (*xbuf)[GW][NB] = reinterpret_cast<double (*)[GW][NB]>(panel_lds + 2 * GW * NB);     // [slot][row in group][column]
    const int tid = threadIdx.x;
    const int tc = tid >> 4, tr = tid & 15;
#pragma unroll 1
    for (int g = 0; g < NB / GW; ++g) {
        const int pb = g & 1, c = GW * g;
        TGP_STAMP(0);
        if ((tc / TPG) == g) {
            const int cm0 = (tc % TPG) * 4;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                d2_t v0, v1;
                v0[0] = a[0][m]; v0[1] = a[1][m]; v1[0] = a[2][m]; v1[1] = a[3][m];
                *reinterpret_cast<d2_t *>(&colbuf[pb][cm0 + m][4 * tr]) = v0;
                *reinterpret_cast<d2_t *>(&colbuf[pb][cm0 + m][4 * tr + 2]) = v1;
            }
        }
        if ((tr / TPG) == g) {
            const int k0 = (tr % TPG) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d2_t v0, v1;
                v0[0] = x[k][0]; v0[1] = x[k][1]; v1[0] = x[k][2]; v1[1] = x[k][3];
                *reinterpret_cast<d2_t *>(&xbuf[pb][k0 + k][4 * tc]) = v0;
                *reinterpret_cast<d2_t *>(&xbuf[pb][k0 + k][4 * tc + 2]) = v1;
            }
        }
        TGP_STAMP(1);
        __syncthreads();
        TGP_STAMP(2);
        // ---- GW x GW pivot block, lower part: L[k][m] (k > m) and 1 / L[m][m] ----
        double L[GW][GW], rs[GW];
#pragma unroll
        for (int m = 0; m < GW; ++m) {
            // rows c+m .. of raw column m, read as aligned pairs
            double dcol[GW];
#pragma unroll
            for (int k2 = (m & ~1); k2 < GW; k2 += 2) {
                const d2_t v = *reinterpret_cast<const d2_t *>(&colbuf[pb][m][c + k2]);
                dcol[k2] = v[0]; dcol[k2 + 1] = v[1];
            }
            double piv = dcol[m];
#pragma unroll
            for (int q = 0; q < m; ++q) piv = fma(-L[m][q], L[m][q], piv);
            // LAPACK dpotrf stops at a pivot <= 0 (scipy.linalg.cholesky -> LinAlgError,
            // _gpr.py:348-358).  A pivot that has lost every significant digit (< 8 eps of the
            // diagonal) is reported the same way: its sign is rounding noise.
            if (!(piv > tiny) || !isfinite(piv)) {
                if (tid == 0 && *flag == 0) *flag = o + c + m + 1;
                piv = 1.0;
            }
            rs[m] = rsqrt_newton(piv);
#pragma unroll
            for (int k = m + 1; k < GW; ++k) {
                double v = dcol[k];
#pragma unroll
                for (int q = 0; q < m; ++q) v = fma(-L[k][q], L[m][q], v);
                L[k][m] = v * rs[m];
            }
        }
        TGP_STAMP(3);
        // ---- forward substitutions L_dd R = B ----
        const bool isA = tc >= TPG * (g + 1);                // this thread updates A (else X)
        const double *rsrc = isA ? &colbuf[pb][0][0] : &xbuf[pb][0][0];
        double rowsT[GW][4], rhs[GW][4];
#pragma unroll
        for (int m = 0; m < GW; ++m) {
            const d2_t r0 = *reinterpret_cast<const d2_t *>(&colbuf[pb][m][4 * tr]);
            const d2_t r1 = *reinterpret_cast<const d2_t *>(&colbuf[pb][m][4 * tr + 2]);
            const d2_t s0 = *reinterpret_cast<const d2_t *>(rsrc + m * NB + 4 * tc);
            const d2_t s1 = *reinterpret_cast<const d2_t *>(rsrc + m * NB + 4 * tc + 2);
            double v[4] = {r0[0], r0[1], r1[0], r1[1]};
            double w[4] = {s0[0], s0[1], s1[0], s1[1]};
#pragma unroll
            for (int q = 0; q < m; ++q) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = fma(-rowsT[q][i], L[m][q], v[i]);
                    w[i] = fma(-rhs[q][i], L[m][q], w[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { rowsT[m][i] = v[i] * rs[m]; rhs[m][i] = w[i] * rs[m]; }
        }
        TGP_STAMP(4);
        // ---- finalise the owners' entries ----
        // columns of the group, rows from the block down: L (for the block's own rows the same
        // substitution yields L_dd itself in the lower part; the upper part is never stored)
        if ((tc / TPG) == g && tr >= TPG * g) {
            const bool hi = (tc % TPG) != 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if (TPG == 1) {
                        a[i][m] = rowsT[m][i];
                    } else {
                        // (opaque to the optimiser: it would otherwise turn the half-select into a
                        // dynamically indexed load and park the whole array in scratch memory)
                        double lo_v = rowsT[m][i], hi_v = rowsT[GW - 4 + m][i];
                        asm volatile("" : "+v"(lo_v), "+v"(hi_v));
                        a[i][m] = hi ? hi_v : lo_v;
                    }
                }
        }
        // rows of the block in X (non-zero only up to the group's columns)
        if ((tr / TPG) == g && !isA) {
            const bool hi = (tr % TPG) != 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (TPG == 1) {
                        x[k][j] = rhs[k][j];
                    } else {
                        double lo_v = rhs[k][j], hi_v = rhs[GW - 4 + k][j];
                        asm volatile("" : "+v"(lo_v), "+v"(hi_v));
                        x[k][j] = hi ? hi_v : lo_v;
                    }
                }
        }
        TGP_STAMP(5);
        // ---- rank-GW update of the rows below the block: A right of the group, X up to it ----
        if (g + 1 < NB / GW) {
            const bool below = tr >= TPG * (g + 1);
            double t[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) t[i][j] = isA ? a[i][j] : x[i][j];
#pragma unroll
            for (int m = 0; m < GW; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[i][j] = fma(-rowsT[m][i], rhs[m][j], t[i][j]);
            const bool updA = below && isA, updX = below && !isA;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[i][j] = updA ? t[i][j] : a[i][j];
                    x[i][j] = updX ? t[i][j] : x[i][j];
                }
        }
        TGP_STAMP(6);
    }
}

// ---- diagonal-block factorisation, variant C (default): four columns per barrier, ONE live
// tile per thread, no selects on the hot path ---------------------------------------------------
// In-kernel stamps (tools/microbench/chol64_stamp.hip) showed variant B's 3100 cycles per step to
// be ~270 f64 operations (1400 cycles of issue) plus ~1000 cycles of v_cndmask selects between the
// A and X tiles, and a 4-pivot chain of ~220 cycles per pivot that carried the not-PD test
// (compare, branch, select) on it.  Here
//   * a thread keeps ONE 4x4 tile `act`: its A tile while its column group lies right of the
//     current group (g < tc), its X tile afterwards.  At the step g == tc the finished L tile is
//     stored straight to its destination (no second register tile) and `act` restarts as the X
//     tile (final rows of X for the diagonal tile, zero elsewhere);
//   * rows that must not change (at / above the pivot block) are masked by multiplying the solved
//     rows with 0 / 1 (16 multiplies) instead of selecting 32 + 32 results;
//   * the not-PD test only records the first failing pivot; the pivot itself flows on unchanged
//     (a failing fit is discarded as a whole), so compare / select leave the dependency chain;
//   * waves without a lane in the group skip the publish / switch code by wave-uniform branches.
// Ldst: where this workgroup's L tile goes (leading dimension ldL), or null to skip the store.
// On return x holds the X = L^-1 tile (zeros above the diagonal); the reciprocal pivots
// 1 / L[j][j] are in lds[CHOL64_RS_OFF + j] for the caller's log-sum.
constexpr int CHOL64_RS_OFF = 4 * 8 * NB;   // doubles: after colbuf[2][GW][64] and xbuf[2][GW][64], GW <= 8

template <int GW>
__device__ __forceinline__ void factor64_v3(double (&act)[4][4], double *panel_lds, int o,
                                            double *__restrict__ Ldst, long ldL,
                                            int *__restrict__ flag, double tiny) {
    constexpr int TPG = GW / 4;                                  // thread columns (rows) per group
    double (*colbuf)[GW][NB] = reinterpret_cast<double (*)[GW][NB]>(panel_lds);              // [slot][column in group][row]
    double (*xbuf)[GW][NB] = reinterpret_cast<double (*)[GW][NB]>(panel_lds + 2 * GW * NB);   // [slot][row in group][column]
    double *rsbuf = panel_lds + CHOL64_RS_OFF;
    const int tid = threadIdx.x;
    const int tc = tid >> 4, tr = tid & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // holds tc = 4*wave .. 4*wave+3
    int bad = 0;                                                 // first failing pivot + 1 (same in every thread)
#pragma unroll 1
    for (int g = 0; g < NB / GW; ++g) {
        const int pb = g & 1, c = GW * g;
        const int gw = (TPG * g) >> 2;                           // the wave that holds the group's thread columns
        TGP_STAMP(0);
        if (wave == gw && (tc / TPG) == g) {                     // raw columns of the group, all 64 rows
            const int cm0 = (tc % TPG) * 4;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                d2_t v0, v1;
                v0[0] = act[0][m]; v0[1] = act[1][m]; v1[0] = act[2][m]; v1[1] = act[3][m];
                *reinterpret_cast<d2_t *>(&colbuf[pb][cm0 + m][4 * tr]) = v0;
                *reinterpret_cast<d2_t *>(&colbuf[pb][cm0 + m][4 * tr + 2]) = v1;
            }
        }
        if (4 * wave < TPG * (g + 1) && (tr / TPG) == g && tc < TPG * (g + 1)) {
            // rows of the running inverse that belong to the group, columns up to the group's.
            // Inside the group's own diagonal block X is still the identity.
            const bool inblk = (tc / TPG) == g;
            const double one = (tr == tc) ? 1.0 : 0.0;
            const int k0 = (tr % TPG) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d2_t v0, v1;
                v0[0] = inblk ? (k == 0 ? one : 0.0) : act[k][0];
                v0[1] = inblk ? (k == 1 ? one : 0.0) : act[k][1];
                v1[0] = inblk ? (k == 2 ? one : 0.0) : act[k][2];
                v1[1] = inblk ? (k == 3 ? one : 0.0) : act[k][3];
                *reinterpret_cast<d2_t *>(&xbuf[pb][k0 + k][4 * tc]) = v0;
                *reinterpret_cast<d2_t *>(&xbuf[pb][k0 + k][4 * tc + 2]) = v1;
            }
        }
        TGP_STAMP(1);
        __syncthreads();
        TGP_STAMP(2);
        // ---- GW x GW pivot block: L[k][m] (k > m) and rs[m] = 1 / L[m][m] ----
        double L[GW][GW], rs[GW];
#pragma unroll
        for (int m = 0; m < GW; ++m) {
            double dcol[GW];
#pragma unroll
            for (int k2 = (m & ~1); k2 < GW; k2 += 2) {
                const d2_t v = *reinterpret_cast<const d2_t *>(&colbuf[pb][m][c + k2]);
                dcol[k2] = v[0]; dcol[k2 + 1] = v[1];
            }
            double piv = dcol[m];
#pragma unroll
            for (int q = 0; q < m; ++q) piv = fma(-L[m][q], L[m][q], piv);
            // LAPACK dpotrf stops at a pivot <= 0 (scipy.linalg.cholesky -> LinAlgError,
            // _gpr.py:348-358).  A pivot that has lost every significant digit (< 8 eps of the
            // diagonal) is reported the same way: its sign is rounding noise.  Recorded only: the
            // (NaN) results of a failing factorisation are never used.
            const bool ok = (piv > tiny) && (piv <= 1.7976931348623157e308);
            bad = (bad == 0 && !ok) ? (o + c + m + 1) : bad;
            rs[m] = rsqrt_newton(piv);
#pragma unroll
            for (int k = m + 1; k < GW; ++k) {
                double v = dcol[k];
#pragma unroll
                for (int q = 0; q < m; ++q) v = fma(-L[k][q], L[m][q], v);
                L[k][m] = v * rs[m];
            }
        }
#pragma unroll
        for (int m = 0; m < GW; ++m)
            if (tid == m) rsbuf[c + m] = rs[m];
        TGP_STAMP(3);
        // ---- forward substitutions L_dd R = B: own rows, and own columns (A side) or the rows of X ----
        const bool isA = tc >= TPG * (g + 1);
        const double *rsrc = isA ? &colbuf[pb][0][0] : &xbuf[pb][0][0];
        double rowsT[GW][4], rhs[GW][4];
#pragma unroll
        for (int m = 0; m < GW; ++m) {
            const d2_t r0 = *reinterpret_cast<const d2_t *>(&colbuf[pb][m][4 * tr]);
            const d2_t r1 = *reinterpret_cast<const d2_t *>(&colbuf[pb][m][4 * tr + 2]);
            const d2_t s0 = *reinterpret_cast<const d2_t *>(rsrc + m * NB + 4 * tc);
            const d2_t s1 = *reinterpret_cast<const d2_t *>(rsrc + m * NB + 4 * tc + 2);
            double v[4] = {r0[0], r0[1], r1[0], r1[1]};
            double w[4] = {s0[0], s0[1], s1[0], s1[1]};
#pragma unroll
            for (int q = 0; q < m; ++q) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = fma(-rowsT[q][i], L[m][q], v[i]);
                    w[i] = fma(-rhs[q][i], L[m][q], w[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { rowsT[m][i] = v[i] * rs[m]; rhs[m][i] = w[i] * rs[m]; }
        }
        TGP_STAMP(4);
        // ---- the group's own thread columns: L tile out, `act` restarts as the X tile ----
        if (wave == gw) {
            if ((tc / TPG) == g) {
                if (Ldst != nullptr) {
                    // the lower triangle holds L, everything above the diagonal is zero
                    const bool hi = (tc % TPG) != 0;
                    double *dst = Ldst + (long)(4 * tr) * ldL + 4 * tc;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        double lv[4];
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            double v;
                            if (TPG == 1) {
                                v = rowsT[m][i];
                            } else {
                                // (opaque to the optimiser: it would otherwise turn the half-select
                                // into a dynamically indexed load and park the array in scratch)
                                double lo_v = rowsT[m][i], hi_v = rowsT[GW - 4 + m][i];
                                asm volatile("" : "+v"(lo_v), "+v"(hi_v));
                                v = hi ? hi_v : lo_v;
                            }
                            lv[m] = ((4 * tc + m) <= (4 * tr + i)) ? v : 0.0;
                        }
                        d2_t v0, v1;
                        v0[0] = lv[0]; v0[1] = lv[1]; v1[0] = lv[2]; v1[1] = lv[3];
                        *reinterpret_cast<d2_t *>(dst + (long)i * ldL) = v0;
                        *reinterpret_cast<d2_t *>(dst + (long)i * ldL + 2) = v1;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) act[i][j] = 0.0;
            }
        }
        if (4 * wave < TPG * (g + 1)) {
            if ((tr / TPG) == g && tc < TPG * (g + 1)) {          // the group's rows of X are final
                const bool hi = (tr % TPG) != 0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (TPG == 1) {
                            act[k][j] = rhs[k][j];
                        } else {
                            double lo_v = rhs[k][j], hi_v = rhs[GW - 4 + k][j];
                            asm volatile("" : "+v"(lo_v), "+v"(hi_v));
                            act[k][j] = hi ? hi_v : lo_v;
                        }
                    }
            }
        }
        TGP_STAMP(5);
        // ---- rank-GW update of the rows below the block (A right of the group, X up to it) ----
        if (g + 1 < NB / GW) {
            const double rmask = (tr >= TPG * (g + 1)) ? 1.0 : 0.0;
#pragma unroll
            for (int m = 0; m < GW; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) rowsT[m][i] *= rmask;
#pragma unroll
            for (int m = 0; m < GW; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) act[i][j] = fma(-rowsT[m][i], rhs[m][j], act[i][j]);
        }
        TGP_STAMP(6);
    }
    if (bad != 0 && tid == 0 && *flag == 0) *flag = bad;
}

// ---- diagonal-block factorisation, variant D: the block lives in LDS, 16-column steps ----------
// Variants A-C keep the block in registers (4 x 4 per thread) and pay ~450 instructions per thread
// and 4-column step (publish / barrier / redundant pivot-block factorisation / two forward
// substitutions / update): about 690 cycles per COLUMN, instruction-issue bound (in-kernel stamps,
// tools/microbench/chol64_stamp.hip).  Here
//   * a 16 x 16 pivot block is factored AND inverted by ONE wave without any barrier: lane i < 16
//     holds row i of the block, lane 16 + j the (accumulating) column j of the inverse; each step
//     broadcasts the pivot and the scaled column through v_readlane (SGPR operands), and the same
//     fma stream  r[idx] -= S(idx) * own  updates the trailing rows of A in one lane group and the
//     forward substitution of the inverse in the other -- about 275 cycles per column (stamped);
//   * the panel below the block (L = A X_dd^T) and the rank-16 update of the trailing block run on
//     MFMA (v_mfma_f64_16x16x4), one 16 x 16 tile per wave and turn, operands read from LDS;
//   * wave 0 updates the next pivot tile first and goes straight on to factor it while the other
//     waves finish the update: two barriers per 16 columns;
//   * X = L^-1 is assembled at the end from the four inverted pivot blocks by two levels of
//     [[A,0],[C,B]]^-1 = [[Ai,0],[-Bi C Ai,Bi]] merges on MFMA.
// In: At = the 64 x 64 block (row-major, leading dimension CH_LD), Xt = zeros.  Out: At lower
// triangle = L (upper garbage), Xt = L^-1 (zeros above the diagonal); scratch: 64 doubles.
constexpr int CH_LD = NB + 2;

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// one wave: Cholesky factor and its inverse of the 16 x 16 block At[c.., c..]
//   lanes 0..15 (group A): r[k] = row i of the block, turning into row i of L;
//   lanes 16..31 (group X): r[i] = acc(i, j) = sum_k L[i][k] X[k][j] for column j = lane & 15 of
//   X = L^-1, turning into X[i][j] = -acc / L[i][i]; the diagonal starts at -1 so that the same
//   formula yields X[j][j] = 1 / L[j][j], and everything above it stays an exact zero.
// Step k: t = r[k] / L[k][k] in EVERY lane (group A: L[i][k]; group X: -X[k][j]), then
// r[idx] -= L[idx][k] * t for idx > k is the Cholesky update in group A and the forward
// substitution of the inverse in group X -- one instruction stream, no selects.  L[idx][k] comes
// back from a 16-double LDS vector as broadcast reads, except L[k+1][k], which the next pivot
// waits for and which travels by v_readlane.
__device__ __forceinline__ int chol16_inv_wave(double (*At)[CH_LD], double (*Xt)[CH_LD], double *scratch,
                                               int c, int o, double tiny, int bad) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15;
    const bool isX = (lane >> 4) & 1;
    const double sg = isX ? -1.0 : 1.0;
    double *colv = scratch + 16 * (lane >> 4);            // each 16-lane group writes its own copy; group A's is read
    double r[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; k2 += 2) {
        const d2_t v = *reinterpret_cast<const d2_t *>(&At[c + i][c + k2]);
        r[k2] = isX ? ((k2 == i) ? -1.0 : 0.0) : v[0];
        r[k2 + 1] = isX ? ((k2 + 1 == i) ? -1.0 : 0.0) : v[1];
    }
    double tprev = 0.0, Sprev[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double piv = readlane_f64(r[k], k);
        const double rs = rsqrt_newton(piv);                  // (a failing pivot is found after the loop)
        const double t = r[k] * rs;                           // group A: L[i][k];  group X: -X[k][j]
        r[k] = sg * t;                                        // final L[i][k] / X[k][j]
        // the next two pivots wait for rows k + 1 and k + 2: those travel by v_readlane ...
        if (k < 15) r[k + 1] = fma(-readlane_f64(t, k + 1), t, r[k + 1]);
        if (k < 14) r[k + 2] = fma(-readlane_f64(t, k + 2), t, r[k + 2]);
        // ... the rest of column k - 1 arrived from LDS during this step (issued a step ago)
        if (k >= 1) {
#pragma unroll
            for (int idx = k + 2; idx < 16; ++idx) r[idx] = fma(-Sprev[idx], tprev, r[idx]);   // Sprev[idx] = L[idx][k-1]
        }
        if (k < 13) {
            // publish column k, fetch its rows k + 3 .. for the NEXT step.  One wave: DS operations
            // execute in issue order and the compiler keeps may-alias stores and loads in program
            // order, so no fence and no wait sit between the write and the reads.
            colv[i] = t;
#pragma unroll
            for (int p2 = ((k + 3) & ~1); p2 < 16; p2 += 2) {                     // aligned pairs, same address in every lane
                const d2_t v = *reinterpret_cast<const d2_t *>(scratch + p2);
                Sprev[p2] = v[0]; Sprev[p2 + 1] = v[1];
            }
            tprev = t;
        }
    }
    // LAPACK dpotrf stops at a pivot <= 0 (scipy.linalg.cholesky -> LinAlgError, _gpr.py:348-358);
    // a pivot without a significant digit left (< 8 eps of the diagonal) is reported the same way.
    // Checked here instead of on the pivot chain: a bad pivot leaves L[i][i] = sqrt(pivot) small or
    // NaN, and NaN spreads to every later pivot, so the FIRST failing row is the lowest lane that
    // fails.  (The results of a failing factorisation are discarded as a whole.)
    double dg = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) dg = (k == i) ? r[k] : dg;   // L[i][i] in group A
    const bool ok = isX || ((dg * dg > tiny) && (dg <= 1.3407807929942596e154));
    int first = ok ? 64 : i;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int f2 = __shfl_xor(first, off, 64); first = f2 < first ? f2 : first; }
    bad = (bad == 0 && first < 16) ? (o + c + first + 1) : bad;
    if (lane < 32) {
        if (!isX) {
            // row i of L_dd (what lies right of the diagonal is never used)
#pragma unroll
            for (int k2 = 0; k2 < 16; k2 += 2) {
                d2_t v; v[0] = r[k2]; v[1] = r[k2 + 1];
                *reinterpret_cast<d2_t *>(&At[c + i][c + k2]) = v;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) Xt[c + k][c + i] = r[k];                    // column i of X_dd, exact zeros above the diagonal
        }
    }
    return bad;
}

// C (16 x 16, accumulators) (+)= A[ra.., ka..ka+K) * B[rb.., kb..kb+K)^T, operands in LDS tiles;
// TRANSB: B is read transposed (B[n][k] = Bt[kb + k][rb + n])
template <bool TRANSB>
__device__ __forceinline__ void mma16(const double (*A)[CH_LD], int ra, int ka, const double (*B)[CH_LD], int rb,
                                      int kb, int K, d4_t &acc, double sign) {
    const int lane = threadIdx.x & 63;
    const int m = lane & 15, kg = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const double a = sign * A[ra + m][ka + k0 + kg];
        const double b = TRANSB ? B[kb + k0 + kg][rb + m] : B[rb + m][kb + k0 + kg];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
}
// accumulator element r of this lane sits at (row (lane >> 4) + 4 r, column lane & 15)
__device__ __forceinline__ void acc16_load(const double (*T)[CH_LD], int r0, int c0, d4_t &acc) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = T[r0 + (lane >> 4) + 4 * r][c0 + (lane & 15)];
}
__device__ __forceinline__ void acc16_store(double (*T)[CH_LD], int r0, int c0, const d4_t &acc, double scale) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < 4; ++r) T[r0 + (lane >> 4) + 4 * r][c0 + (lane & 15)] = scale * acc[r];
}

// nlive (round 6, the small-problem path): columns from nlive on are identity padding (unit diagonal, zero
// elsewhere, as kmat_tile builds them).  A 16-column pivot block of pure padding factors to itself, its panel
// and its update are zero: the serial pivot chain -- 2 us of one wave per block, most of this function --
// is only walked for the live blocks, and the padding blocks of X get their unit diagonal directly.  What the
// live blocks see is unchanged (they never read a padding block), so L and X are the bytes of nlive = 64.
// The loop keeps its three trips and every barrier (the panel and the update of a padding block are a few MFMAs
// on zeros): with nlive = NB, the default of every caller on the blocked path, the code is round 5's.
// (A first form that ended the loop early -- trip count in an SGPR from readfirstlane -- returned wrong factors
// from the one-launch hyper-parameter fit, where this function sits in a non-inlined callee inside the optimiser's
// loop, and only there; profiles/r06_device_optimiser_bisect.txt.  tests/test_gpu_round6.py holds every short call
// to the bytes of round 5's body, TGP_SMALL_LIVE=0, and the one-launch optimiser to the same answer twice.)
__device__ __forceinline__ void factor64_v4(double (*At)[CH_LD], double (*Xt)[CH_LD], double (*Tb)[CH_LD],
                                            double *scratch, int o, int *__restrict__ flag, double tiny,
                                            int nlive = NB) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bad = 0;
    const d4_t zero4 = {0.0, 0.0, 0.0, 0.0};
    const int live16 = __builtin_amdgcn_readfirstlane(nlive >= NB ? 4 : (nlive + 15) / 16);   // live 16-column pivot blocks
    if (tid >= 16 * live16 && tid < NB) Xt[tid][tid] = 1.0;                      // (Xt came in as zeros; published by the barriers below)
    if (wave == 0) bad = chol16_inv_wave(At, Xt, scratch, 0, o, tiny, bad);
#pragma unroll 1
    for (int s = 0; s < 3; ++s) {
        const int c = 16 * s, nb = 3 - s;                      // nb row tiles below the pivot block
        __syncthreads();                                        // L_dd, X_dd of this step are in LDS
        // ---- panel: L_t = A_t X_dd^T, tile t on wave t + 1 ----
        if (wave >= 1 && wave <= nb) {
            const int r0 = c + 16 * wave;
            d4_t acc = zero4;
            mma16<false>(At, r0, c, Xt, c, c, 16, acc, 1.0);
            acc16_store(At, r0, c, acc, 1.0);
        }
        __syncthreads();
        // ---- rank-16 update of the trailing block, lower tiles; wave 0 takes the next pivot tile
        // first and goes on to factor it while the others finish ----
        int t = 0;
        for (int ti = 0; ti < nb; ++ti)
            for (int tj = 0; tj <= ti; ++tj, ++t) {
                if ((t & 3) != wave) continue;
                const int ri = c + 16 * (ti + 1), rj = c + 16 * (tj + 1);
                d4_t acc;
                acc16_load(At, ri, rj, acc);
                mma16<false>(At, ri, c, At, rj, c, 16, acc, -1.0);
                acc16_store(At, ri, rj, acc, 1.0);
            }
        if (wave == 0 && s + 1 < live16) bad = chol16_inv_wave(At, Xt, scratch, c + 16, o, tiny, bad);
    }
    __syncthreads();
    // ---- X = L^-1 from the four inverted pivot blocks: [[A,0],[C,B]]^-1 = [[Ai,0],[-Bi C Ai,Bi]] ----
    // level 32: blocks (0,1) on wave 0, (2,3) on wave 1
    if (wave < 2) {
        const int c0 = 32 * wave, c1 = c0 + 16;
        d4_t acc = zero4;
        mma16<true>(At, c1, c0, Xt, c0, c0, 16, acc, 1.0);     // T = L21 X11
        acc16_store(Tb, c1, c0, acc, 1.0);
        // (same wave wrote and reads Tb: the LDS operations of one wave complete in order)
        acc = zero4;
        mma16<true>(Xt, c1, c1, Tb, c0, c1, 16, acc, 1.0);     // X22 T
        acc16_store(Xt, c1, c0, acc, -1.0);
    }
    __syncthreads();
    // level 64: T = L21 X11 (32 x 32, one 16 x 16 tile per wave), then X21 = -X22 T
    {
        const int ti = wave >> 1, tj = wave & 1;
        d4_t acc = zero4;
        mma16<true>(At, 32 + 16 * ti, 0, Xt, 16 * tj, 0, 32, acc, 1.0);
        acc16_store(Tb, 32 + 16 * ti, 16 * tj, acc, 1.0);
        __syncthreads();
        acc = zero4;
        mma16<true>(Xt, 32 + 16 * ti, 32, Tb, 16 * tj, 32, 32, acc, 1.0);
        __syncthreads();                                        // every wave has read Xt's 32.. rows before any writes X21
        acc16_store(Xt, 32 + 16 * ti, 16 * tj, acc, -1.0);
    }
    __syncthreads();
    if (bad != 0 && (tid & 63) == 0 && wave == 0 && *flag == 0) *flag = bad;
}

// ---- whole 64 x 64 LDS tiles on MFMA (shared by the fit's panel kernels and the small-problem path) ----
// C (+)= A * B^T for 64 x 64 LDS tiles, both K-contiguous; 4 waves, each a 32 x 32 quadrant of
// 2 x 2 v_mfma_f64_16x16x4 fragments (the panel solve's arrangement)
__device__ __forceinline__ void tile_mma64(const double (*As)[CH_LD], const double (*Bs)[CH_LD], d4_t (&acc)[2][2]) {
    using MF = Mfma<double>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    const int fidx = MF::ab_idx(lane), fkg = MF::ab_kg(lane) * 2;
#pragma unroll
    for (int ks = 0; ks < NB; ks += 8) {
        d2_t av[2], bv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const d2_t *>(&As[wm0 + 16 * i + fidx][ks + fkg]);
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const d2_t *>(&Bs[wn0 + 16 * j + fidx][ks + fkg]);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = MF::mma(av[i][e], bv[j][e], acc[i][j]);
    }
}

__device__ __forceinline__ void acc_zero(d4_t (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
}

// visit the (row, col) of every accumulator element of this lane
template <typename F>
__device__ __forceinline__ void acc_foreach(const d4_t (&acc)[2][2], F f) {
    using MF = Mfma<double>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                f(wm0 + 16 * i + MF::c_row(lane, r), wn0 + 16 * j + MF::c_col(lane), acc[i][j][r]);
}


}  // namespace tgp
