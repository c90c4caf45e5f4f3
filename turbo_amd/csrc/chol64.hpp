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

}  // namespace tgp
