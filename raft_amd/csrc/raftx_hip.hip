// raftx_hip.hip -- MI355X (gfx950 / CDNA4) implementation of include/raftx.h.
//
// Hot path of WISDEM/RAFT: Morison strip sweep + stochastic drag linearisation
// fixed point + per-frequency 6x6 complex solve (raft/raft_model.py:994-1236,
// raft/raft_fowt.py:1732-1957, raft/raft_member.py:1899-2152), written
// directly for CDNA4: fp64 VALU, wave64, one workgroup per (design, sea state),
// NB frequency bins per lane.
//
// Mapping:
//   * frequency is the contiguous axis of every reference array, so lane <-> w
//     makes every global load/store of a [6,nw] / [6,6,nw] slab coalesced;
//   * strip records (256 B each) are wave-uniform: the kernels read them with
//     scalar loads (constant cache -> SGPRs) and use them as the scalar operand
//     of v_fma_f64, amortised over the NB bins a lane owns;
//   * the only cross-frequency couplings -- the per-strip vRMS sums
//     (raft_member.py:2084-2090, helpers.py:684) and the convergence test
//     (raft_model.py:1104) -- go through per-wave LDS transposition tiles
//     inside one workgroup; nothing crosses workgroups or devices;
//   * the 6x6 complex impedance is factorised per lane in registers with
//     LAPACK-style partial pivoting (pivot on |re|+|im|, as izamax).
//
// No fallback paths: every entry point either runs on the GPU or fails.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>        // types only: the library is dlopen'ed on the first raftx_comm_* call
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <map>
#include <functional>
#include <mutex>
#include <type_traits>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/raftx.h"

// roctx ranges around the phases of the host side: named spans for `rocprofv3 --marker-trace`.
// librocprofiler-sdk-roctx is bound at run time on first use; without it (or outside a profiler) the ranges cost a branch.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        if (getenv("RAFTX_NO_ROCTX")) return;
        for (const char *n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr;
                pop = nullptr;
            }
        }
    }
};
static Roctx &roctx() {
    static Roctx r;
    return r;
}
struct RangeScope {
    bool on;
    explicit RangeScope(const char *name) : on(roctx().push != nullptr) {
        if (on) roctx().push(name);
    }
    ~RangeScope() {
        if (on) roctx().pop();
    }
};

#include "raftx_kernels.h"
#include "raftx_qtf.h"
#include "raftx_geom.h"
#include "raftx_fusedgen.h"
#include "raftx_dense.h"
#include "raftx_flex.h"

// Coupled array solve (raft_model.py:1164-1236): Xi = Z_sys^-1 F for every (system, bin).  One wavefront per
// (system, bin), NBIN (1, 2 or 4: what fits LDS) consecutive bins per workgroup so that the loads of one matrix entry
// for the bins of a workgroup are one contiguous segment of the [.., nw] slabs.  The augmented matrix [Z_sys | F] of a
// wave lives in LDS; Gaussian elimination with partial pivoting (pivot on |re| + |im|, first largest, as izamax):
// pivot search as a wave reduction, multipliers with a lane per row, the rank-1 update with the lanes over the
// (row, column) entries, back substitution column by column with a lane per (row, right-hand side).  A wave's LDS
// traffic needs no s_barrier (wave_lds_fence).
template <bool RESIDENT>
__global__ void __launch_bounds__(256) k_solve_system(int nSys, int nUnit, int nRhs, int nw, int nCase,
                                                      const double *__restrict__ w, const cplx *__restrict__ Zblk,
                                                      const double *__restrict__ Mc, const double *__restrict__ Bc,
                                                      const double *__restrict__ Cc, const cplx *__restrict__ F,
                                                      cplx *__restrict__ Xi) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n = 6 * nUnit, ld = n + nRhs, nel = n * ld;
    const int nbin = blockDim.x >> 6, ngrp = (nw + nbin - 1) / nbin;
    const int s = blockIdx.x / ngrp, iw0 = (blockIdx.x % ngrp) * nbin;
    cplx *Aall = reinterpret_cast<cplx *>(smem);           // [nbin][n][ld]
    // host layout: Zblk [nSys,nUnit,36,nw], F [nSys,nRhs,n,nw], coupling per system.
    // resident layout (results of k_solve_dynamics): unit u of system (g, c) is pair (g*nUnit+u)*nCase + c;
    // Z [pair,36,nw], F_wave [pair,nRhs,6,nw], coupling per group g.
    const int g = RESIDENT ? s / nCase : s, ic = RESIDENT ? s % nCase : 0;
    for (int t = threadIdx.x; t < nel * nbin; t += blockDim.x) {
        const int e = t / nbin, b = t % nbin;
        const int iw = min(iw0 + b, nw - 1);
        const double ww = w[iw];
        const int r = e / ld, c = e % ld;
        cplx v = {0.0, 0.0};
        if (c < n) {
            if (r / 6 == c / 6) {
                const size_t pair = RESIDENT ? ((size_t)g * nUnit + r / 6) * nCase + ic : (size_t)s * nUnit + r / 6;
                v = Zblk[((pair * 6 + r % 6) * 6 + c % 6) * nw + iw];
            }
            size_t o = (size_t)g * n * n + (size_t)r * n + c;
            double m = Mc ? Mc[o] : 0.0, bb = Bc ? Bc[o] : 0.0, kk = Cc ? Cc[o] : 0.0;
            if (Mc || Bc || Cc) {
                v.re += -(ww * ww) * m + kk;
                v.im += ww * bb;
            }
        } else if (RESIDENT) {
            const size_t pair = ((size_t)g * nUnit + r / 6) * nCase + ic;
            v = F[((pair * nRhs + (c - n)) * 6 + r % 6) * nw + iw];
        } else {
            v = F[(((size_t)s * nRhs + (c - n)) * n + r) * nw + iw];
        }
        Aall[(size_t)b * nel + e] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    cplx *A = Aall + (size_t)wv * nel;
    for (int k = 0; k < n; k++) {
        // pivot row: first row of the largest |re| + |im| in column k
        double best = -1.0;
        int p = n;
        for (int r = k + lane; r < n; r += 64) {
            const double v = fabs(A[r * ld + k].re) + fabs(A[r * ld + k].im);
            if (v > best) {
                best = v;
                p = r;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) argmax_step(best, p, off);
        if (p >= n) p = k;                                    // a column of NaNs: no row compares larger
        if (p != k)
            for (int c = lane; c < ld; c += 64) {
                const cplx t = A[k * ld + c];
                A[k * ld + c] = A[p * ld + c];
                A[p * ld + c] = t;
            }
        wave_lds_fence();
        const cplx pv = A[k * ld + k];
        const double dd = pv.re * pv.re + pv.im * pv.im;
        const cplx inv = {pv.re / dd, -pv.im / dd};
        for (int r = k + 1 + lane; r < n; r += 64) A[r * ld + k] = cmul(A[r * ld + k], inv);
        wave_lds_fence();
        // rank-1 update: the lanes as an 8 x 8 grid over (row, column), strided by 8 -- no division in the loop
        for (int r = k + 1 + (lane >> 3); r < n; r += 8) {
            const cplx l = A[r * ld + k];
            for (int c = k + 1 + (lane & 7); c < ld; c += 8) A[r * ld + c] = csub(A[r * ld + c], cmul(l, A[k * ld + c]));
        }
        wave_lds_fence();
    }
    // back substitution, column by column: x_k = b_k / u_kk, then b_r -= u_rk x_k for the rows above
    for (int k = n - 1; k >= 0; k--) {
        const cplx pv = A[k * ld + k];
        const double dd = pv.re * pv.re + pv.im * pv.im;
        for (int j = lane; j < nRhs; j += 64) {
            const cplx sum = A[k * ld + n + j];
            A[k * ld + n + j] = cplx{(sum.re * pv.re + sum.im * pv.im) / dd, (sum.im * pv.re - sum.re * pv.im) / dd};
        }
        wave_lds_fence();
        if (nRhs == 1) {
            const cplx xk = A[k * ld + n];
            for (int r = lane; r < k; r += 64) A[r * ld + n] = csub(A[r * ld + n], cmul(A[r * ld + k], xk));
        } else {
            for (int e = lane; e < k * nRhs; e += 64) {
                const int r = e / nRhs, j = e % nRhs;
                A[r * ld + n + j] = csub(A[r * ld + n + j], cmul(A[r * ld + k], A[k * ld + n + j]));
            }
        }
        wave_lds_fence();
    }
    __syncthreads();
    // responses out: the bins of the workgroup side by side
    for (int t = threadIdx.x; t < n * nRhs * nbin; t += blockDim.x) {
        const int e = t / nbin, b = t % nbin;
        const int k = e % n, j = e / n;
        if (iw0 + b < nw) Xi[(((size_t)s * nRhs + j) * n + k) * nw + iw0 + b] = Aall[(size_t)b * nel + k * ld + n + j];
    }
}
// Register-resident variant for arrays of 2 .. 5 units (n = 12 .. 30 rows): a LANE PER ROW, two systems per wavefront (lanes
// 0-31 and 32-63), the whole row -- n complex entries plus up to four right-hand sides -- in the lane's registers.  The
// elimination is unrolled over the column k, so every register index is static; partial pivoting is implicit (the pivot
// row stays where it is and is marked done: the same pivot rows, multipliers and updates as zgetrf's interchanges,
// raft_model.py:1191), the pivot search is a two-phase 32-bit DPP argmax within the half-wave, the pivot row reaches the other
// rows through a row buffer in LDS (its owner lane stores it, everyone reads it back as a broadcast).  ~3.7 k VALU
// instructions per PAIR of systems against ~10 k per system of the LDS-resident kernel above (whose update loop spends its
// time on index arithmetic, and whose assembly through LDS is a chain of dependent loads).  What bounds it -- the LDS itself:
// a published pivot-row entry costs 8 LDS-array cycles whatever the store's width or active lanes (ds_write_b128 is eight
// 8-lane groups; 4 x b32 is four times two 32-lane groups) plus 4 to read it back, ~2 100 LDS cycles per system on the ONE
// LDS of a CU = 37 ms per 10^7 systems before anything else; 24 of the 60 ms go with the stores compiled out
// (profiles/r05_lu_store_experiment.json); a software-pipelined publication measured 16 % slower.  Round 6 built the
// variant with TWO LANES PER ROW (half the registers, four waves per SIMD instead of two, both owner lanes storing in one
// instruction): bit-identical and NOT faster (61.0 against 60.2 ms, profiles/r06_experiments/lu_two_lanes_per_row.txt) --
// more waves cannot help a kernel bound by LDS-array cycles per system, and the one-lane form already shares every store
// between its two systems.  Removed again.  What would cut the published volume is a blocked factorisation whose trailing
// update runs on the matrix pipe (v_mfma_f64_16x16x4), at the price of the pivots' bit-identity: DESIGN.md 3.3.
#define SYSROWS_MAXRHS 4
// ASM (resident form only): the units' 6 x 6 impedance blocks are ASSEMBLED here, Z = -w^2 M0 + i w (B0 + B_drag) + C0 with the
// unit kernel's own expression (assemble_and_solve: fma(-w^2, M, C), w * (B0 + Bd) -- the same bits), from 36 x 4 doubles
// per unit instead of 36 x nw complex per pair: the fixed points then need not export Z at all (a farm sweep of 200 000
// pairs writes and re-reads 23 GB of it) and run as the LEAN kernel with the excitation export (KF_OUTF).
template <int NU, int NR, bool RESIDENT, bool ASM = false>          // NR: right-hand sides compiled in (nRhs <= NR; the rest are zero columns)
#ifndef RAFTX_SYSROWS_WAVES
#define RAFTX_SYSROWS_WAVES 2     // waves per SIMD the register allocation aims at (tuning builds: 3, 4)
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RAFTX_SYSROWS_WAVES, RAFTX_SYSROWS_WAVES))) k_solve_system_rows(int nSys, int nRhs, int nw, int nCase, const double *__restrict__ w,
                                                          const cplx *__restrict__ Zblk, const double *__restrict__ Mc,
                                                          const double *__restrict__ Bc, const double *__restrict__ Cc,
                                                          const cplx *__restrict__ F, cplx *__restrict__ Xi,
                                                          const double *__restrict__ uM = nullptr, const double *__restrict__ uB = nullptr,
                                                          const double *__restrict__ uC = nullptr, const double *__restrict__ uBd = nullptr) {
    constexpr int N = 6 * NU;
    const int ngrp = (nw + 1) / 2;                       // two bins (systems) per wavefront: lanes 0-31 and 32-63
    const int s = blockIdx.x / ngrp, lane = threadIdx.x, half = lane >> 5, r = lane & 31, hbase = lane & 32;
    const int iwr = (blockIdx.x % ngrp) * 2 + half;
    const bool live = iwr < nw;
    const int iw = live ? iwr : nw - 1;
    const bool row = r < N;
    const int rr = row ? r : 0;
    const int g = RESIDENT ? s / nCase : s, ic = RESIDENT ? s % nCase : 0;
    // The lane's row of [Z_sys | F] (raft_model.py:1164-1191), straight into registers: the unit's own 6 x 6 block from the
    // per-unit impedances, the coupling terms from the group's matrices (rows are contiguous), everything else zero.  All
    // loads are independent: one exposed round trip.
    cplx a[N], bR[NR];
    const double ww = w[iw];
    const int u = rr / 6, q = rr % 6;
    const size_t pair = RESIDENT ? ((size_t)g * NU + u) * nCase + ic : (size_t)s * NU + u;
    cplx zb[6];
    if constexpr (ASM) {
        const size_t dsg = ((size_t)g * NU + u) * 36 + q * 6, pr = pair * 36 + q * 6;
        const double w2 = ww * ww;
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const double Bq = uB[dsg + c] + uBd[pr + c];
            zb[c] = {fma(-w2, uM[dsg + c], uC[dsg + c]), ww * Bq};
        }
    } else {
#pragma unroll
        for (int c = 0; c < 6; c++) zb[c] = Zblk[((pair * 6 + q) * 6 + c) * nw + iw];
    }
#pragma unroll
    for (int j = 0; j < NR; j++)
        bR[j] = j < nRhs ? (RESIDENT ? F[((pair * nRhs + j) * 6 + q) * nw + iw] : F[(((size_t)s * nRhs + j) * N + rr) * nw + iw])
                         : cplx{0.0, 0.0};
    const size_t o = (size_t)g * N * N + (size_t)rr * N;
#pragma unroll
    for (int c = 0; c < N; c++) {
        const double m = Mc ? Mc[o + c] : 0.0, bb = Bc ? Bc[o + c] : 0.0, kk = Cc ? Cc[o + c] : 0.0;
        a[c] = {fma(-(ww * ww), m, kk), ww * bb};
    }
#pragma unroll
    for (int c = 0; c < N; c++) {                        // the diagonal block of the lane's unit (u is per lane: selects)
        const int cu = c / 6;
        const cplx z = zb[c % 6];
        a[c].re += cu == u ? z.re : 0.0;
        a[c].im += cu == u ? z.im : 0.0;
    }
    // the pivot row of a step travels through a row buffer in LDS: written by the one lane of each half that owns it, read
    // back by all (same address within a half: a broadcast read)
    __shared__ __attribute__((aligned(16))) double rowbuf_[2 * 2 * (N + NR)];
    cplx *rowbuf = reinterpret_cast<cplx *>(rowbuf_) + half * (N + NR);
    bool todo = row;                                     // this row has not been a pivot row yet
    int mystep = N;                                      // elimination step at which it was
    const unsigned rkey = 31u - (unsigned)r;             // ties go to the lower row (izamax takes the first largest)
#pragma unroll
    for (int k = 0; k < N; k++) {
        // pivot: the row of the largest |re| + |im| in column k among the rows still to do.  Magnitude and row travel as ONE
        // key -- the row index replaces the five lowest mantissa bits (a choice between candidates equal to 7e-15 is
        // arbitrary anyway) -- so that the argmax is a max: four row_shr steps inside the 16-lane rows, one row_bcast
        // across the two rows of a half, all DPP (no LDS round trips).
        double best = todo ? fabs(a[k].re) + fabs(a[k].im) : -1.0;
        if (todo && !(best >= 0.0)) best = 0.0;          // NaN: comparable, so that a pivot is always found
        // The (magnitude, row) key is compared in two 32-bit phases -- `v_max_f64` takes no DPP operand (round 4's form cost
        // seven VALU instructions per reduction step: two copies, two DPP moves, a canonicalising max, the max), while
        // `v_max_u32` with a zero-filling row_shr is ONE: first the high words (+1, so that 0 means "no candidate": rows
        // that are done), then the low words -- five lowest mantissa bits replaced by the row -- of the lanes that tie on
        // the high word.  Lexicographic (hi, lo) on non-negative doubles is the order of the doubles: the same pivots as
        // before, bit for bit.  The two 16-lane rows of a half meet in scalar registers (readlane + s_max).
        const unsigned khi = todo ? (unsigned)__double2hiint(best) + 1u : 0u;
        const unsigned klo = ((unsigned)__double2loint(best) & ~31u) | rkey;
#define ROW_UMAX_(x)                                                                                   \
        x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));           \
        x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));           \
        x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));           \
        x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));
        unsigned mh = khi;
        ROW_UMAX_(mh)                                    // row_shr:1,2,4,8 -> lane 15 of each 16-lane row
        const unsigned h0 = max((unsigned)__builtin_amdgcn_readlane((int)mh, 15), (unsigned)__builtin_amdgcn_readlane((int)mh, 31));
        const unsigned h1 = max((unsigned)__builtin_amdgcn_readlane((int)mh, 47), (unsigned)__builtin_amdgcn_readlane((int)mh, 63));
        unsigned ml = (todo && khi == (half ? h1 : h0)) ? klo : 0u;
        ROW_UMAX_(ml)
#undef ROW_UMAX_
        const unsigned l0 = max((unsigned)__builtin_amdgcn_readlane((int)ml, 15), (unsigned)__builtin_amdgcn_readlane((int)ml, 31));
        const unsigned l1 = max((unsigned)__builtin_amdgcn_readlane((int)ml, 47), (unsigned)__builtin_amdgcn_readlane((int)ml, 63));
        const int p = 31 - (int)((half ? l1 : l0) & 31u);
        const bool mine = todo && r == p;
        const bool upd = todo && r != p;
#ifndef RAFTX_SYSROWS_EXP
#define RAFTX_SYSROWS_EXP 0       // timing experiments (wrong results): 1 = no pivot-row stores, 2 = neither stores nor loads
#endif
        if (mine && RAFTX_SYSROWS_EXP == 0) {
#pragma unroll
            for (int c = k; c < N; c++) rowbuf[c] = a[c];
#pragma unroll
            for (int j = 0; j < NR; j++)
                rowbuf[N + j] = bR[j];
        }
        wave_lds_fence();
#if RAFTX_SYSROWS_EXP == 2
#define ROWBUF_(c) (cplx{a[(c) < N ? (c) : 0].im + 1.0, a[(c) < N ? (c) : 0].re})
#else
#define ROWBUF_(c) rowbuf[c]
#endif
        const cplx pv = ROWBUF_(k);
        // 1 / |pivot|^2: v_rcp_f64 + two Newton steps (5 instructions; the IEEE division sequence is ~15).  A zero pivot still
        // ends in NaN.
        const double pp = pv.re * pv.re + pv.im * pv.im;
        double dinv = __builtin_amdgcn_rcp(pp);
        dinv = fma(fma(-pp, dinv, 1.0), dinv, dinv);
        dinv = fma(fma(-pp, dinv, 1.0), dinv, dinv);
        const cplx inv = {pv.re * dinv, -pv.im * dinv};
        // rows that are done (or beyond the system) take a zero multiplier: the update itself stays straight-line code
        const cplx lk = cmul(a[k], inv);
        const cplx l = {upd ? lk.re : 0.0, upd ? lk.im : 0.0};
#pragma unroll
        for (int c = k + 1; c < N; c++) {
            const cplx u = ROWBUF_(c);
            a[c] = cfnma(a[c], l, u);                     // four FMAs (as a difference of a product: six instructions)
        }
#pragma unroll
        for (int j = 0; j < NR; j++)
        {
            const cplx u = ROWBUF_(N + j);
            bR[j] = cfnma(bR[j], l, u);
        }
#undef ROWBUF_
        if (mine) {
            todo = false;
            mystep = k;
            a[k] = inv;                                  // the reciprocal pivot, for the back substitution
        }
        wave_lds_fence();                                // the buffer is rewritten in the next step
    }
    // back substitution in pivot order: x_k from the row that was the pivot of step k, then out of the rows of earlier steps
#pragma unroll
    for (int k = N - 1; k >= 0; k--) {
        if (mystep == k) {
#pragma unroll
            for (int j = 0; j < NR; j++) {
                bR[j] = cmul(bR[j], a[k]);
                rowbuf[N + j] = bR[j];
            }
        }
        wave_lds_fence();
        const cplx f = {mystep < k ? a[k].re : 0.0, mystep < k ? a[k].im : 0.0};
#pragma unroll
        for (int j = 0; j < NR; j++)
        {
            const cplx xk = rowbuf[N + j];
            bR[j] = cfnma(bR[j], f, xk);
        }
        wave_lds_fence();
    }
    // responses out: unknown k sits in the row that was the pivot of step k
    if (row && live && mystep < N) {
#pragma unroll
        for (int j = 0; j < NR; j++)
            if (j < nRhs) Xi[(((size_t)s * nRhs + j) * N + mystep) * nw + iw] = bR[j];
    }
}
static bool solve_system_rows_ok(int nUnit, int nRhs) {
    static const char *off = getenv("RAFTX_SYSTEM_LDS");                 // tuning / tests: keep the LDS-resident kernel
    return !(off && atoi(off)) && nUnit >= 2 && nUnit <= 5 && nRhs >= 1 && nRhs <= SYSROWS_MAXRHS;
}
// launches the register-resident kernel; returns false if this shape has none
template <bool RESIDENT, bool ASM = false>
static bool launch_solve_system_rows(hipStream_t st, int nSys, int nUnit, int nRhs, int nw, int nCase, const double *w, const cplx *Z,
                                     const double *Mc, const double *Bc, const double *Cc, const cplx *F, cplx *X,
                                     const double *uM = nullptr, const double *uB = nullptr, const double *uC = nullptr,
                                     const double *uBd = nullptr) {
    if (!solve_system_rows_ok(nUnit, nRhs)) return false;
    const dim3 grid((unsigned)((size_t)nSys * ((nw + 1) / 2)));
    const int nr = nRhs == 1 ? 1 : (nRhs == 2 ? 2 : 4);
#define ROWS_CASE(NU_, NR_)                                                                                             \
    if (nUnit == NU_ && nr == NR_) {                                                                                    \
        hipLaunchKernelGGL((k_solve_system_rows<NU_, NR_, RESIDENT, ASM>), grid, dim3(64), 0, st, nSys, nRhs, nw, nCase, w, Z, Mc, Bc, Cc, F, X, \
                           uM, uB, uC, uBd);                                                                           \
        return true;                                                                                                    \
    }
    ROWS_CASE(2, 1) ROWS_CASE(2, 2) ROWS_CASE(2, 4) ROWS_CASE(3, 1) ROWS_CASE(3, 2) ROWS_CASE(3, 4)
    ROWS_CASE(4, 1) ROWS_CASE(4, 2) ROWS_CASE(4, 4) ROWS_CASE(5, 1) ROWS_CASE(5, 2) ROWS_CASE(5, 4)
#undef ROWS_CASE
    return false;
}

// The same assembly as a kernel of its own, for the shapes the register-resident solver does not take (more than five
// units / four right-hand sides): Z [pair,36,nw] into a scratch slab, then the LDS-resident solver as before.
__global__ void __launch_bounds__(256) k_assemble_unit_z(int npair, int nCase, int nw, const double *__restrict__ w,
                                                         const double *__restrict__ uM, const double *__restrict__ uB,
                                                         const double *__restrict__ uC, const double *__restrict__ uBd,
                                                         cplx *__restrict__ Z) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)npair * 36 * nw) return;
    const int iw = (int)(t % nw), e = (int)((t / nw) % 36);
    const size_t pair = t / ((size_t)36 * nw), d = pair / nCase;
    const double ww = w[iw], Bq = uB[d * 36 + e] + uBd[pair * 36 + e];
    Z[t] = {fma(-(ww * ww), uM[d * 36 + e], uC[d * 36 + e]), ww * Bq};
}

// bins per workgroup and dynamic LDS of k_solve_system
static int solve_system_shape(int n, int nRhs, size_t *lds) {
    const size_t per = sizeof(cplx) * (size_t)n * (n + nRhs);
    const int nbin = per * 4 <= 64 * 1024 ? 4 : (per * 2 <= 64 * 1024 ? 2 : 1);
    *lds = per * nbin;
    return nbin;
}

// Motion statistics of the resident responses (raft_fowt.py:2310-2357; helpers.py:678-700): one
// workgroup per (design, case), lanes stride the frequency axis (coalesced 16 B/lane reads of the
// Xi slab: a pure HBM stream, 19.2 KB in -> 48 B out per pair at C3).
__global__ void __launch_bounds__(256) k_motion_stats(int npair, int nHead, int nw, double inv_dw,
                                                      const cplx *__restrict__ Xi, double *__restrict__ sd,
                                                      double *__restrict__ psd,
                                                      const int *__restrict__ niter = nullptr, const int *__restrict__ flags = nullptr,
                                                      int *__restrict__ ints_out = nullptr) {
    __shared__ double part[4][6];
    const int p = blockIdx.x;
    if (p >= npair) return;
    const double r2d = 180.0 / 3.14159265358979323846;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < nw; i += blockDim.x) {
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const double scale = j >= 3 ? r2d : 1.0;
            double a2 = 0.0;
            for (int ih = 0; ih < nHead; ih++) {
                const cplx x = Xi[(((size_t)p * nHead + ih) * 6 + j) * nw + i];
                const double xr = x.re * scale, xi = x.im * scale;
                a2 += xr * xr + xi * xi;
            }
            acc[j] += a2;
            if (psd) psd[((size_t)p * 6 + j) * nw + i] = 0.5 * a2 * inv_dw;
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double a = acc[j];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) part[wv][j] = a;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double a = 0.0;
        for (int q = 0; q < (int)(blockDim.x >> 6); q++) a += part[q][threadIdx.x];
        sd[(size_t)p * 6 + threadIdx.x] = sqrt(0.5 * a);
    }
    if (ints_out && threadIdx.x == 6) {                   // sweep crossings: iteration count and flags of the pair ride along
        ints_out[p] = niter[p];                           // (out[0..n) = niter, out[n..2n) = flags, page-locked host memory)
        ints_out[npair + p] = flags[p];
    }
}

// Statistics of linear output channels y_c = w^p sum_j L[c,j] Xi_j (hub accelerations, tension Jacobian rows ...):
// one workgroup per (design, case); the Xi slab is streamed once per channel (L2-resident after the first).
__global__ void __launch_bounds__(256) k_channel_stats(int nCase, int nHead, int nw, int nChan, double inv_dw,
                                                       const double *__restrict__ w, const cplx *__restrict__ Xi,
                                                       const double *__restrict__ L, const int *__restrict__ pw,
                                                       double *__restrict__ sd, double *__restrict__ psd) {
    __shared__ double part[4];
    const int p = blockIdx.x, d = p / nCase;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int ch = 0; ch < nChan; ch++) {
        const double *row = L + ((size_t)d * nChan + ch) * 6;
        const double l0 = row[0], l1 = row[1], l2 = row[2], l3 = row[3], l4 = row[4], l5 = row[5];
        const int pe = pw[ch];
        double acc = 0.0;
        for (int i = threadIdx.x; i < nw; i += blockDim.x) {
            double ws = 1.0;
            for (int e = 0; e < pe; e++) ws *= w[i];
            double a2 = 0.0;
            for (int ih = 0; ih < nHead; ih++) {
                const cplx *x = Xi + (((size_t)p * nHead + ih) * 6) * nw + i;
                const cplx x0 = x[0], x1 = x[(size_t)nw], x2 = x[(size_t)2 * nw], x3 = x[(size_t)3 * nw], x4 = x[(size_t)4 * nw],
                           x5 = x[(size_t)5 * nw];
                const double yr = ws * (l0 * x0.re + l1 * x1.re + l2 * x2.re + l3 * x3.re + l4 * x4.re + l5 * x5.re);
                const double yi = ws * (l0 * x0.im + l1 * x1.im + l2 * x2.im + l3 * x3.im + l4 * x4.im + l5 * x5.im);
                a2 += yr * yr + yi * yi;
            }
            acc += a2;
            if (psd) psd[((size_t)p * nChan + ch) * nw + i] = 0.5 * a2 * inv_dw;
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        __syncthreads();
        if (lane == 0) part[wv] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = 0.0;
            for (int q = 0; q < (int)(blockDim.x >> 6); q++) a += part[q];
            sd[(size_t)p * nChan + ch] = sqrt(0.5 * a);
        }
    }
}

// BEM excitation with heading interpolation (raft_fowt.py:1796-1849): one workgroup per (pair, heading), lanes over w
__global__ void __launch_bounds__(256) k_bem_excitation(DevTables T, int nHB, const double *__restrict__ heads,
                                                        const cplx *__restrict__ X, const double *__restrict__ hadj,
                                                        const double *__restrict__ xy, const cplx *__restrict__ Fadd,
                                                        cplx *__restrict__ F) {
    const int ih = blockIdx.x % T.nHead, p = blockIdx.x / T.nHead;
    const int d = p / T.nCase, ic = p % T.nCase;
    const double beta = T.beta[(size_t)ic * T.nHead + ih];
    double bdeg = fmod(beta * (180.0 / M_PI) - (hadj ? hadj[d] : 0.0), 360.0);        // Python's % : result in [0, 360)
    if (bdeg < 0.0) bdeg += 360.0;
    int i1 = nHB - 1, i2 = 0;
    double f2;
    if (bdeg <= heads[0]) {
        const double hlast = heads[nHB - 1] - 360.0;
        f2 = (bdeg - hlast) / (heads[0] - hlast);
    } else if (bdeg >= heads[nHB - 1]) {
        const double hfirst = heads[0] + 360.0;
        f2 = (bdeg - heads[nHB - 1]) / (hfirst - heads[nHB - 1]);
    } else {
        f2 = 0.0;
        for (int i = 0; i < nHB - 1; i++)
            if (heads[i + 1] > bdeg) {
                i1 = i;
                i2 = i + 1;
                f2 = (bdeg - heads[i]) / (heads[i + 1] - heads[i]);
                break;
            }
    }
    const double f1 = 1.0 - f2, sb = sin(beta), cb = cos(beta);
    const double xr = xy ? xy[2 * d] : 0.0, yr = xy ? xy[2 * d + 1] : 0.0;
    const cplx *X1 = X + ((size_t)d * nHB + i1) * 6 * T.nw, *X2 = X + ((size_t)d * nHB + i2) * 6 * T.nw;
    const size_t base = ((size_t)p * T.nHead + ih) * 6 * T.nw;
    for (int i = threadIdx.x; i < T.nw; i += blockDim.x) {
        cplx xp[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const cplx a = X1[(size_t)j * T.nw + i], b = X2[(size_t)j * T.nw + i];
            xp[j] = {a.re * f1 + b.re * f2, a.im * f1 + b.im * f2};
        }
        cplx g[6];
        g[0] = {xp[0].re * cb - xp[1].re * sb, xp[0].im * cb - xp[1].im * sb};
        g[1] = {xp[0].re * sb + xp[1].re * cb, xp[0].im * sb + xp[1].im * cb};
        g[2] = xp[2];
        g[3] = {xp[3].re * cb - xp[4].re * sb, xp[3].im * cb - xp[4].im * sb};
        g[4] = {xp[3].re * sb + xp[4].re * cb, xp[3].im * sb + xp[4].im * cb};
        g[5] = xp[5];
        double ps, pc;
        sincos(-(T.k[i] * (xr * cos(beta) + yr * sin(beta))), &ps, &pc);
        const double z = T.zeta[((size_t)ic * T.nHead + ih) * T.nw + i];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const double gr = g[j].re * z, gi = g[j].im * z;
            cplx o = {gr * pc - gi * ps, gr * ps + gi * pc};
            if (Fadd) {
                const cplx a = Fadd[base + (size_t)j * T.nw + i];
                o.re += a.re;
                o.im += a.im;
            }
            F[base + (size_t)j * T.nw + i] = o;
        }
    }
}

// y_c = sum_p (i w)^p sum_j L[c,p,j] Xi_j + sum_j Gw[c,j,w] Xi_j  (tower-base moment: weight p=0, inertial reaction
// p=2, complex aero reaction through Gw; raft_fowt.py:2500-2537)
__global__ void __launch_bounds__(256) k_channel_stats_poly(int nCase, int nHead, int nw, int nChan, double inv_dw,
                                                            const double *__restrict__ w, const cplx *__restrict__ Xi,
                                                            const double *__restrict__ L, const cplx *__restrict__ Gw,
                                                            double *__restrict__ sd, double *__restrict__ psd) {
    __shared__ double part[4];
    const int p = blockIdx.x, d = p / nCase;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int ch = 0; ch < nChan; ch++) {
        const double *row = L + ((size_t)d * nChan + ch) * 18;
        const cplx *g = Gw ? Gw + ((size_t)d * nChan + ch) * 6 * nw : nullptr;
        double acc = 0.0;
        for (int i = threadIdx.x; i < nw; i += blockDim.x) {
            const double wi = w[i], w2 = wi * wi;
            double a2 = 0.0;
            for (int ih = 0; ih < nHead; ih++) {
                const cplx *x = Xi + (((size_t)p * nHead + ih) * 6) * nw + i;
                double yr = 0.0, yi = 0.0;
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const cplx xj = x[(size_t)j * nw];
                    double cr = row[j] - w2 * row[12 + j], ci = wi * row[6 + j];     // L0 + (i w) L1 + (i w)^2 L2
                    if (g) {
                        const cplx gj = g[(size_t)j * nw + i];
                        cr += gj.re;
                        ci += gj.im;
                    }
                    yr += cr * xj.re - ci * xj.im;
                    yi += cr * xj.im + ci * xj.re;
                }
                a2 += yr * yr + yi * yi;
            }
            acc += a2;
            if (psd) psd[((size_t)p * nChan + ch) * nw + i] = 0.5 * a2 * inv_dw;
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        __syncthreads();
        if (lane == 0) part[wv] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = 0.0;
            for (int q = 0; q < (int)(blockDim.x >> 6); q++) a += part[q];
            sd[(size_t)p * nChan + ch] = sqrt(0.5 * a);
        }
    }
}

// The same for a response with any number of DOFs handed over by the caller (raftx_response_stats): one workgroup per channel,
// lanes over the bins, the DOF loop inside (coalesced reads of the [.,nw] slabs)
__global__ void __launch_bounds__(256) k_response_stats(int nDof, int nResp, int nw, double inv_dw, const double *__restrict__ w,
                                                        const cplx *__restrict__ Xi, const double *__restrict__ L,
                                                        const cplx *__restrict__ Gw, double *__restrict__ sd, double *__restrict__ psd) {
    __shared__ double part[4];
    const int ch = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double *row = L + (size_t)ch * 3 * nDof;
    const cplx *g = Gw ? Gw + (size_t)ch * nDof * nw : nullptr;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nw; i += blockDim.x) {
        const double wi = w[i], w2 = wi * wi;
        double a2 = 0.0;
        for (int ih = 0; ih < nResp; ih++) {
            const cplx *x = Xi + ((size_t)ih * nDof) * nw + i;
            double yr = 0.0, yi = 0.0;
            for (int j = 0; j < nDof; j++) {
                const cplx xj = x[(size_t)j * nw];
                double cr = row[j] - w2 * row[2 * nDof + j], ci = wi * row[nDof + j];     // L0 + (i w) L1 + (i w)^2 L2
                if (g) {
                    const cplx gj = g[(size_t)j * nw + i];
                    cr += gj.re;
                    ci += gj.im;
                }
                yr += cr * xj.re - ci * xj.im;
                yi += cr * xj.im + ci * xj.re;
            }
            a2 += yr * yr + yi * yi;
        }
        acc += a2;
        if (psd) psd[(size_t)ch * nw + i] = 0.5 * a2 * inv_dw;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) part[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int q = 0; q < (int)(blockDim.x >> 6); q++) a += part[q];
        sd[ch] = sqrt(0.5 * a);
    }
}

// ------------------------------------------------------------------ host side
#define RAFTX_NSLOT RAFTX_SWEEP_SLOTS   // crossings in flight per context: one downloading its responses, one solving, one generated / queued, one uploading
// Per-context device-memory pool: every call of the C-ABI needs a handful of device buffers (tables, results, scratch);
// hipMalloc / hipFree cost 0.1-1 ms each and hipFree synchronises the whole device, which serialises contexts that
// otherwise overlap copies and kernels on their own streams.  Blocks are rounded up (powers of two below 1 MiB, 1 MiB
// multiples above), cached on release and handed out again; everything goes back to the driver in raftx_ctx_destroy.
// Callers return blocks only after their stream has drained.
struct DevPool {
    std::multimap<size_t, void *> free_;
    std::unordered_map<void *, size_t> size_;
    static size_t round_up(size_t b) {
        if (b < 256) b = 256;
        if (b <= ((size_t)1 << 20)) {
            size_t p = 256;
            while (p < b) p <<= 1;
            return p;
        }
        return (b + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    }
    hipError_t get(size_t bytes, void **out) {
        const size_t cap = round_up(bytes);
        auto it = free_.lower_bound(cap);
        if (it != free_.end() && it->first <= cap + cap / 2 + ((size_t)1 << 20)) {
            *out = it->second;
            free_.erase(it);
            return hipSuccess;
        }
        static const bool dbg = getenv("RAFTX_POOL_DEBUG") != nullptr;   // tuning: pool misses (they cost a hipMalloc) on stderr
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(out, cap);
        if (dbg)
            fprintf(stderr, "[raftx pool] miss: hipMalloc(%zu) %.3f ms (%zu blocks free)\n", cap,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), free_.size());
        if (e != hipSuccess) {                      // make room and retry once
            (void)hipGetLastError();
            trim();
            e = hipMalloc(out, cap);
        }
        if (e == hipSuccess) size_[*out] = cap;
        return e;
    }
    void put(void *p) {
        if (!p) return;
        auto it = size_.find(p);
        if (it == size_.end()) {
            (void)hipFree(p);
            return;
        }
        free_.emplace(it->second, p);
    }
    void trim() {
        for (auto &kv : free_) {
            size_.erase(kv.second);
            (void)hipFree(kv.second);
        }
        free_.clear();
    }
};

// A raftx_build_designs in flight: phase 1 (descriptor H2D, member pass, scans, totals to the host) has been enqueued,
// phase 2 (strip tables, statics) follows once the totals are known.  The sweep crossing keeps several of them in flight.
struct BuildJob {
    GeomArgs A;
    std::vector<void *> tmp;            // descriptor uploads and per-member scratch: back to the pool when the job retires
    int nDesign = 0, nw = 0, add_mask = 0;
    int64_t nMember = 0;
    int64_t maxMem = 0, maxSta = 0;     // members / stations of the largest design (host loop of phase 1)
    double *M0d = nullptr, *C0d = nullptr;
    const double *B0d = nullptr, *MBwd = nullptr;
    bool active = false;
    bool reduce_late = false;
    hipStream_t reduce_stream = nullptr; // side stream the member -> platform reductions were launched on (evRed), or null
    // phase 2 left the tables of the designs to the fused kernel that solves them (raftx_fusedgen.h): nothing has been
    // written yet; solve_enqueue launches the generating form, or k_geom_design + k_geom_addup if it cannot
    bool gen_deferred = false;
    size_t gen_lds = 0;                  // dynamic LDS of geom_design_block for this batch
};

// One set of sea-state tables resident for the sweep crossings.  A crossing pins the set it was PREPARED with until it has
// been waited for (or cancelled), so that a later prepare with other sea states on another slot can neither free nor
// replace what an earlier batch still reads (its member pass reads k; its fused kernel everything).
struct CaseSet {
    std::vector<double> key;             // nCase, nHead, nw, depth, rho, g, w, k, zeta, beta as given by the caller
    DevTables T;                         // only the sea-state fields are used
    std::vector<void *> allocs;
    int users = 0;                       // crossings prepared or in flight on this set
    unsigned long long stamp = 0;        // last use (the idle set used longest ago is replaced)
};

// Edit program of parametric variants (raftx_variant_program): the base unit's descriptors and the affine edits, resident
// on the device; the uniform offset arrays of a batch of nDesign variants on the host (cached per batch size).
struct VariantProg {
    int nM = 0, nSt = 0, nCap = 0, nP = 0;
    bool has_caps = false;
    double *gm = nullptr, *gs = nullptr, *gc = nullptr;                  // base descriptors
    int *stMember = nullptr, *capMember = nullptr;                       // member of every station / cap row
    double *stFrac = nullptr, *fillFrac = nullptr, *capFrac = nullptr;   // positions as fractions of the member length
    double *endCoef = nullptr, *headCS = nullptr, *diaCoef = nullptr;
    int *endEdit = nullptr, *diaEdit = nullptr;
    std::vector<int64_t> hS, hC;                                         // base stationOff / capOff (host)
    std::vector<void *> allocs;
    int cachedN = -1;
    std::vector<int64_t> memberOff, stationOff, capOff;                  // uniform offsets of cachedN variants
};
// where a block's descriptors come from when they are not the caller's arrays
struct VariantSrc {
    const VariantProg *prog;
    const double *params;                                                // host, [nDesign of the batch, nP]
};

static void expand_args(const VariantProg &P, int n, double *gm_out, double *gs_out, double *gc_out, ExpandArgs &E) {
    E.n = n; E.nM = P.nM; E.nSt = P.nSt; E.nCap = P.nCap; E.nP = P.nP;
    E.gm = P.gm; E.gs = P.gs; E.gc = P.gc;
    E.stMember = P.stMember; E.capMember = P.capMember;
    E.stFrac = P.stFrac; E.fillFrac = P.fillFrac; E.capFrac = P.capFrac;
    E.endCoef = P.endCoef; E.headCS = P.headCS; E.diaCoef = P.diaCoef;
    E.endEdit = P.endEdit; E.diaEdit = P.diaEdit;
    E.gm_out = gm_out; E.gs_out = gs_out; E.gc_out = gc_out;
}

// resident matrices of the dense solves (raftx_dense_resident): device pointers owned by `allocs`
struct DenseResident {
    int nSet = 0, n = 0, nw = 0, freq_mask = 0;
    double *w = nullptr, *M = nullptr, *B = nullptr, *C = nullptr;
    std::vector<void *> allocs;
};
struct raftx_ctx {
    int device;
    hipStream_t stream;
    bool owns_stream;                    // false for the block contexts of raftx_sweep_stats (they run on the parent's stream)
    hipEvent_t ev0, ev1;
    hipEvent_t evUp, evTot, evG0, evG1, evG2, evG3, evS0, evS1, evDone;   // build phases, statistics, block finished
    hipEvent_t evZ;                      // the first kernel of the member pass has run (phase 1)
    hipEvent_t evMem, evRed;             // member pass done (preparation stream) / per-design reduction done (side stream)
    hipStream_t sAux;                    // side stream of the parent ctx: the reductions of the member pass run beside its scans
    BuildJob job;
    long long *pin;                      // page-locked landing area of the build totals and offsets [8 + nDesign + 1]
    size_t pin_n;
    double *pinRes;                      // page-locked landing area of a block's statistics (sweep crossing)
    size_t pinRes_n;
    hipStream_t sCopy, sPrep, sD2H, sGen; // internal streams of raftx_sweep_stats (created on first use)
    hipStream_t sSlab[2] = {nullptr, nullptr}; // with sGen: the streams the slabs of a crossing with responses out go to (SlabPlan)
    hipStream_t sD2Hlow = nullptr;        // bulk download of the responses: a stream of its own priority class, created when first needed
    hipEvent_t evEpoch = nullptr;         // zero of raftx_sweep_solve_span: recorded when the first crossing of the ctx is launched
    hipStream_t sMainB = nullptr;         // second compute stream of the sweep crossings (odd slots), created when first needed
    hipStream_t sExp = nullptr;           // k_geom_expand of a block (variants): a HIGH-priority stream of its own, created when first needed
    hipEvent_t evExp = nullptr;
    CaseSet csets[RAFTX_NSLOT + 1];      // sea-state tables of the sweep crossings: one per crossing in flight + one being replaced
    unsigned long long cset_clock = 0;
    char err[512];
    DevTables T;
    DevPool pool;
    std::vector<void *> design_allocs, case_allocs, result_allocs;
    // resident results of the last raftx_solve_dynamics_device
    cplx *rXi, *rFw, *rZ, *rFe;
    double *rB, *rXl;
    cplx *flexXl0;                       // raftx_flex_start: the linearisation point of the next raftx_flex_solve (device copy)
    size_t flexXl0_n, flexXl0_cap;       // entries set (0: none) / allocated
    unsigned long long *rXlSlots;
    size_t rXl_n;
    unsigned *kpCtr = nullptr;           // claim counters of the persistent fused launches: a ring of KP_RING sets of 8 (one per XCD slab, 128 B apart)
    unsigned kpNext = 0;
    int nCU = 0;                         // compute units of the device (the persistent grid is what they hold at once)
    int last_flags = -1, last_minb = 0, last_rc = 0;       // specialisation of the last fused-kernel launch (raftx_last_solve_kernel)
    cplx *rQtf;                          // QTFs of the last raftx_qtf_slender call, kept for raftx_qtf_force
    size_t rQtf_n;
    int rQtf_sets, rQtf_nw2;
    cplx *rXl0, *rXlOut;                 // optional restart point / exported linearisation point [npair,6,nw]
    size_t rXlio_n;
    bool have_xl0, want_xlout;
    int *rNi, *rFl;
    size_t r_npair, r_nx, r_nz;
    int r_mask;
    int r_last_mask = 0;                 // RAFTX_WANT_* outputs the LAST fused solve wrote (r_mask: what the buffers could hold)
    bool r_fe;
    int maxS;
    std::vector<int> hS;                 // submerged strips of every design (host copy: LDS classes of the fused kernel)
    int *pairList;                       // device pair lists of a launch split into LDS classes
    size_t pairList_n;
    DenseResident dense;                 // matrices kept for raftx_solve_dense_resident
    int *identList = nullptr;            // 0, 1, 2, ..: the pair list of a launch cut into slabs (SlabPlan)
    size_t identList_n = 0;
    std::vector<hipEvent_t> evSlab;      // completion markers of the slabs (no timing), created as needed, kept
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    unsigned long long *dbg;
    double last_ms;
    bool have_designs, have_cases;
    int nw_designs;
    cplx *rKay;                          // Kim & Yue table of raftx_qtf_kay, consumed by the next raftx_qtf_slender call
    size_t rKay_n;
    int rKay_sets, rKay_nw2;
    bool kay_ready;
    cplx *bemF;                          // resident BEM (+ added) excitation of raftx_bem_excitation [npair,nHead,6,nw]
    size_t bemF_n;
    bool bem_ready;
    // results of the last raftx_build_designs (device pointers owned by design_allocs)
    int g_n;
    size_t g_nStrips, g_nRows;
    double *g_abi, *g_A, *g_Ch, *g_Wh, *g_props, *g_Ms, *g_Cs, *g_Ws;
    hipStream_t sStat = nullptr;         // RAFTX_STATS_STREAM=1: the statistics kernels of sweep crossings
    bool last_gen_fused = false;         // the last fused launch generated its designs' tables itself (raftx_fusedgen.h)
    cplx *g_cm;
    void *comm;                          // ncclComm_t of raftx_comm_init (RCCL), or null
    int comm_rank, comm_world;
    int *commFlag = nullptr;             // one int in HBM: the status word the ranks agree on before an exchange step (comm_agree)
    std::vector<raftx_ctx *> workers[RAFTX_NSLOT]; // block contexts of the sweep crossings (device buffers, pool, events), per slot, kept for reuse
    struct SweepSlot *slots;             // [RAFTX_NSLOT] crossings in flight (raftx_sweep_prepare / _launch / _wait)
    VariantProg vprog;                   // raftx_variant_program
};
// Offset arrays of a whole batch, resident on the device (uploaded once by the sweep crossing, shared by its blocks).
struct DevOffsets {
    const int64_t *memberOff, *stationOff, *capOff;      // device copies of the caller's arrays, absolute values
};
// One sweep crossing in flight: everything raftx_sweep_wait needs to finish it.
struct SweepSlot {
    bool busy = false;                   // launched (phase 2 enqueued), not yet waited for
    int cset = -1;                       // the CaseSet of the parent this crossing was prepared with (pinned until retired)
    bool prepared = false;               // phase 1 enqueued (descriptor upload, member pass), not yet launched
    int nIter = 0;
    double tol = 0, XiStart = 0, dw = 0;
    std::vector<raftx_ctx *> blk;
    std::vector<int> bnd;
    int nCase = 0, nHead = 0, nw = 0;
    double *sd = nullptr;
    int32_t *niter = nullptr, *flags = nullptr;
    raftx_c128 *Xi = nullptr;
    int64_t *stripOffsets = nullptr;
    std::vector<void *> allocs;          // the batch's offset arrays on the device, shared by its blocks
    hipEvent_t evXi = nullptr;           // download of the responses finished (sD2H)
    // phase 1 (descriptor upload + member pass) of the blocks behind the second one is enqueued by raftx_sweep_launch, one
    // block ahead of the block being launched: the caller's arrays (alive until the batch has been waited for) and the
    // batch's device offsets are kept for that
    struct {
        const int64_t *memberOff, *stationOff, *capOff;
        const double *members, *stations, *caps, *pose, *M0, *B0, *C0, *Fz_moor, *k;
        double rho, g;
        int add_mask;
        DevOffsets dOff;
        VariantSrc var;                  // prog == nullptr: the caller's descriptor arrays
    } p1;
    size_t next_p1 = 0;                  // first block whose phase 1 has not been enqueued yet
    std::chrono::steady_clock::time_point t0;
    double tl[4] = {0, 0, 0, 0};
    double span[2] = {0, 0};             // fused launches of the crossing last waited for: start / end, ms since the ctx epoch
    size_t nSlabEv = 0;
    bool slab = false;                   // this crossing's fused launches were cut into slabs (SlabPlan)
    std::vector<double> tlb;             // RAFTX_SWEEP_DEBUG: host time per block of raftx_sweep_launch (upload enqueued | totals seen | enqueued)
};

#define MAX_NW 2048
#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                              \
            return -2;                                                                                 \
        }                                                                                              \
    } while (0)
#define FAIL(ctx, ...)                                        \
    do {                                                      \
        snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__); \
        return -1;                                            \
    } while (0)

extern "C" int raftx_version(void) { return RAFTX_VERSION; }
extern "C" int raftx_is_device(void) { return 1; }
extern "C" int raftx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int raftx_ctx_create(int device_id, raftx_ctx **out) {
    if (!out) return -1;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return -3;   // no GPU: fail loudly, no fallback
    if (device_id < 0 || device_id >= ndev) return -4;
    if (hipSetDevice(device_id) != hipSuccess) return -5;
    raftx_ctx *c = new raftx_ctx();
    memset(&c->T, 0, sizeof(c->T));
    c->device = device_id;
    c->err[0] = 0;
    c->maxS = 0;
    c->pairList = nullptr;
    c->pairList_n = 0;
    c->dbg = nullptr;
    c->last_ms = 0.0;
    c->have_designs = c->have_cases = false;
    c->nw_designs = 0;
    c->g_n = 0;
    c->bemF = nullptr;
    c->bemF_n = 0;
    c->bem_ready = false;
    c->rKay = nullptr;
    c->rKay_n = 0;
    c->rKay_sets = c->rKay_nw2 = 0;
    c->kay_ready = false;
    c->rXi = c->rFw = c->rZ = c->rFe = nullptr;
    c->rB = c->rXl = nullptr;
    c->rXlSlots = nullptr;
    c->flexXl0 = nullptr;
    c->flexXl0_n = c->flexXl0_cap = 0;
    c->rXl_n = 0;
    c->rQtf = nullptr;
    c->rQtf_n = 0;
    c->rQtf_sets = c->rQtf_nw2 = 0;
    c->rXl0 = c->rXlOut = nullptr;
    c->rXlio_n = 0;
    c->have_xl0 = c->want_xlout = false;
    c->rNi = c->rFl = nullptr;
    c->r_npair = c->r_nx = c->r_nz = 0;
    c->r_mask = 0;
    c->r_fe = false;
    c->comm = nullptr;
    c->comm_rank = 0;
    c->comm_world = 1;
    c->owns_stream = true;
    c->slots = new SweepSlot[RAFTX_NSLOT];
    c->pin = nullptr;
    c->pin_n = 0;
    c->pinRes = nullptr;
    c->pinRes_n = 0;
    c->sCopy = c->sPrep = c->sD2H = c->sGen = nullptr;
    // RAFTX_CTX_PRIORITY=high (tuning): the ctx stream -- the fused fixed points, table generation, statistics -- in the
    // highest priority class, so that the dispatcher serves its grids before the preparation kernels of the batches behind.
    // Measured and NOT the default (gpurun_out/r05_ctxprio, same box, K = 40, two repeats): 3.27 against 3.07 ms per step with
    // three batches in flight, 3.21 against 3.13 with two, 3.29 against 3.08 with host-made descriptors -- the member pass of
    // the next batch then cannot use the running kernel's drain and lands behind it.
    bool ok;
    {
        static const char *prio = getenv("RAFTX_CTX_PRIORITY");
        int least = 0, greatest = 0;
        if (prio && !strcmp(prio, "high") && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
            ok = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest) == hipSuccess;
        else
            ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    }
    c->sAux = nullptr;
    for (hipEvent_t *e : {&c->evZ, &c->ev0, &c->ev1, &c->evUp, &c->evTot, &c->evG0, &c->evG1, &c->evG2, &c->evG3, &c->evS0,
                          &c->evS1, &c->evDone, &c->evMem, &c->evRed})
        ok = ok && hipEventCreate(e) == hipSuccess;
    if (!ok) {
        delete[] c->slots;
        delete c;
        return -6;
    }
    *out = c;
    return 0;
}

static void free_list(raftx_ctx *c, std::vector<void *> &v) {
    for (void *p : v) c->pool.put(p);
    v.clear();
}

extern "C" void raftx_ctx_destroy(raftx_ctx *c) {
    if (!c) return;
#ifdef GEOM_PHASE_TIMING
    if (c->owns_stream && c->slots) {
        unsigned long long ph[10] = {0};
        (void)hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(ph, HIP_SYMBOL(geom_phase_cycles), sizeof(ph)) == hipSuccess && ph[0]) {
            unsigned long long tot = 0;
            for (int i = 0; i < 7; i++) tot += ph[i];
            fprintf(stderr, "[k_geom_design phases, %% of wave time] counts %.1f | generate %.1f | abi out %.1f | runs %.1f | ds out %.1f | g-vectors %.1f | morison+matrices %.1f\n",
                    100.0 * ph[0] / tot, 100.0 * ph[1] / tot, 100.0 * ph[2] / tot, 100.0 * ph[3] / tot, 100.0 * ph[4] / tot,
                    100.0 * ph[5] / tot, 100.0 * ph[6] / tot);
        }
    }
#endif
    (void)raftx_comm_destroy(c);
    if (c->owns_stream) {                             // a crossing prepared and never launched still has work on sCopy / sPrep
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
    }
    for (int sl = 0; sl < RAFTX_NSLOT; sl++) {
        for (raftx_ctx *w : c->workers[sl]) raftx_ctx_destroy(w);
        c->workers[sl].clear();
    }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_list(c, c->design_allocs);
    free_list(c, c->case_allocs);
    free_list(c, c->result_allocs);
    for (CaseSet &cs : c->csets) free_list(c, cs.allocs);
    c->pool.trim();
    if (c->rXl) (void)hipFree(c->rXl);
    if (c->rXlSlots) (void)hipFree(c->rXlSlots);
    if (c->kpCtr) (void)hipFree(c->kpCtr);
    if (c->flexXl0) (void)hipFree(c->flexXl0);
    if (c->rXl0) (void)hipFree(c->rXl0);
    if (c->rXlOut) (void)hipFree(c->rXlOut);
    if (c->rQtf) (void)hipFree(c->rQtf);
    if (c->bemF) (void)hipFree(c->bemF);
    if (c->rKay) (void)hipFree(c->rKay);
    if (c->pairList) (void)hipFree(c->pairList);
    free_list(c, c->dense.allocs);
    free_list(c, c->vprog.allocs);
    if (c->identList) (void)hipFree(c->identList);
    for (hipEvent_t e : c->evSlab) (void)hipEventDestroy(e);
    if (c->evFork) (void)hipEventDestroy(c->evFork);
    if (c->evJoin) (void)hipEventDestroy(c->evJoin);
    free_list(c, c->job.tmp);
    for (int sl = 0; sl < RAFTX_NSLOT; sl++) {
        free_list(c, c->slots[sl].allocs);
        if (c->slots[sl].evXi) (void)hipEventDestroy(c->slots[sl].evXi);
    }
    delete[] c->slots;
    c->pool.trim();
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->pinRes) (void)hipHostFree(c->pinRes);
    for (hipEvent_t e : {c->evZ, c->ev0, c->ev1, c->evUp, c->evTot, c->evG0, c->evG1, c->evG2, c->evG3, c->evS0, c->evS1, c->evDone,
                         c->evMem, c->evRed})
        (void)hipEventDestroy(e);
    if (c->sAux) (void)hipStreamDestroy(c->sAux);
    if (c->evExp) (void)hipEventDestroy(c->evExp);
    if (c->evEpoch) (void)hipEventDestroy(c->evEpoch);
    for (hipStream_t st : {c->sCopy, c->sPrep, c->sD2H, c->sGen, c->sD2Hlow, c->sSlab[0], c->sSlab[1], c->sExp, c->sMainB, c->sStat})
        if (st) (void)hipStreamDestroy(st);
    if (c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int raftx_host_alloc(raftx_ctx *c, size_t bytes, void **out) {
    if (!c || !out) return -1;
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return 0;
}
extern "C" int raftx_host_free(raftx_ctx *c, void *ptr) {
    if (!c) return -1;
    if (ptr) HIPCHK(c, hipHostFree(ptr));
    return 0;
}

// Where the GPU sits on the host: PCI address and NUMA node (sysfs), for placing the feeding thread and its page-locked
// buffers on the socket the GPU hangs off.  No ctx: callers ask before they create one.
extern "C" int raftx_device_locality(int device, char *pci_bus_id, int len, int *numa_node) {
    if (numa_node) *numa_node = -1;
    char id[64] = {0};
    if (hipDeviceGetPCIBusId(id, (int)sizeof(id), device) != hipSuccess) return -2;
    if (pci_bus_id && len > 0) snprintf(pci_bus_id, (size_t)len, "%s", id);
    for (char *p = id; *p; p++) *p = (char)tolower(*p);
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", id);
    if (FILE *f = fopen(path, "r")) {
        int node = -1;
        if (fscanf(f, "%d", &node) == 1 && numa_node) *numa_node = node;
        fclose(f);
    }
    return 0;
}

extern "C" const char *raftx_last_error(raftx_ctx *c) { return c ? c->err : "null ctx"; }
extern "C" double raftx_last_kernel_ms(raftx_ctx *c) { return c ? c->last_ms : 0.0; }

template <typename Tp, typename Dp>
static int upload(raftx_ctx *c, std::vector<void *> &bag, const Tp *host, size_t n, Dp *dev) {
    *dev = nullptr;
    if (!host || n == 0) return 0;
    void *p = nullptr;
    HIPCHK(c, c->pool.get(n * sizeof(Tp), &p));
    bag.push_back(p);
    HIPCHK(c, hipMemcpyAsync(p, host, n * sizeof(Tp), hipMemcpyHostToDevice, c->stream));
    *dev = reinterpret_cast<const Tp *>(p);
    return 0;
}

extern "C" int raftx_upload_designs(raftx_ctx *c, int nDesign, const int64_t *stripOffsets, const double *strips,
                                    int nStripFields, const double *M0, const double *B0, const double *C0, int nw,
                                    const double *MBw, const int64_t *cmOffsets, const raftx_c128 *CmMCF) {
    RangeScope range_("raftx_upload_designs: run detection + H2D");
    if (!c) return -1;
    if (nStripFields != NF) FAIL(c, "nStripFields=%d, expected %d", nStripFields, NF);
    if (nDesign < 0 || !stripOffsets || !M0 || !B0 || !C0) FAIL(c, "upload_designs: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_list(c, c->design_allocs);
    c->have_designs = false;
    c->bem_ready = false;
    int maxS = 0;
    c->hS.assign((size_t)nDesign, 0);
    for (int d = 0; d < nDesign; d++) {
        int64_t S = stripOffsets[d + 1] - stripOffsets[d];
        if (S < 0) FAIL(c, "strip offsets not monotone at design %d", d);
        if (S > maxS) maxS = (int)S;
        c->hS[(size_t)d] = (int)S;
    }
    // Device strip table.  Straight runs of equally spaced strips (members) are detected here,
    // from the absolute positions alone, so that the kernels can advance the wave kinematics
    // along a run with rotors instead of re-evaluating sincos/exp per strip.  The optional
    // RAFTX_F_STEP / RAFTX_F_UNIT hints of the ABI record are not trusted (nor needed).
    std::vector<double> dsv((size_t)stripOffsets[nDesign] * DS_N, 0.0);
    std::vector<int> dsf((size_t)stripOffsets[nDesign], 0);
    for (int d = 0; d < nDesign; d++)
        derive_design_tables(strips, stripOffsets[d], stripOffsets[d + 1], dsv.data(), dsf.data());
    DevTables &T = c->T;
    T.nDesign = nDesign;
    int rc = 0;
    rc |= upload(c, c->design_allocs, stripOffsets, (size_t)nDesign + 1, &T.off);
    rc |= upload(c, c->design_allocs, dsv.data(), dsv.size(), &T.ds);
    rc |= upload(c, c->design_allocs, dsf.data(), dsf.size(), &T.dsi);
    rc |= upload(c, c->design_allocs, M0, (size_t)nDesign * 36, &T.M0);
    rc |= upload(c, c->design_allocs, B0, (size_t)nDesign * 36, &T.B0);
    rc |= upload(c, c->design_allocs, C0, (size_t)nDesign * 36, &T.C0);
    rc |= upload(c, c->design_allocs, MBw, MBw ? (size_t)nDesign * 72 * nw : 0, &T.MBw);
    T.cmoff = nullptr;
    T.cm = nullptr;
    if (cmOffsets && CmMCF) {
        rc |= upload(c, c->design_allocs, cmOffsets, (size_t)nDesign + 1, &T.cmoff);
        rc |= upload(c, c->design_allocs, reinterpret_cast<const cplx *>(CmMCF), (size_t)cmOffsets[nDesign] * 2 * nw,
                     &T.cm);
    }
    if (rc) return -2;
    HIPCHK(c, hipStreamSynchronize(c->stream));   // host buffers may be released after return
    c->maxS = maxS;
    c->nw_designs = nw;
    c->have_designs = true;
    c->g_n = 0;
    return 0;
}

// ---- geometry -> strip tables + statics on the device (raftx_geom.h)
template <typename Tp>
static int dev_alloc(raftx_ctx *c, std::vector<void *> &bag, size_t n, Tp **out, bool zero = false) {
    *out = nullptr;
    void *p = nullptr;
    HIPCHK(c, c->pool.get((n ? n : 1) * sizeof(Tp), &p));
    bag.push_back(p);
    if (zero) HIPCHK(c, hipMemsetAsync(p, 0, (n ? n : 1) * sizeof(Tp), c->stream));
    *out = reinterpret_cast<Tp *>(p);
    return 0;
}

// ---- raftx_build_designs in two phases, so that the sweep crossing can keep several design blocks in flight.
// Phase 1 enqueues the descriptor H2D on sCopy and, behind it on sPrep, the member pass and the scans; the totals
// (wet strips, MacCamy-Fuchs rows, strips of the largest design, error flags) and the design offsets land in page-locked
// memory and evTot marks them.  Nothing here waits for the device.
static int pin_reserve(raftx_ctx *c, size_t n) {
    if (c->pin && c->pin_n >= n) return 0;
    if (c->pin) HIPCHK(c, hipHostFree(c->pin));
    c->pin = nullptr;
    void *p_ = nullptr;
    HIPCHK(c, hipHostMalloc(&p_, n * sizeof(long long), hipHostMallocDefault));
    c->pin = reinterpret_cast<long long *>(p_);
    c->pin_n = n;
    return 0;
}
template <typename Tp, typename Dp>
static int upload_on(raftx_ctx *c, hipStream_t st, std::vector<void *> &bag, const Tp *host, size_t n, Dp *dev) {
    *dev = nullptr;
    if (!host || n == 0) return 0;
    void *p = nullptr;
    HIPCHK(c, c->pool.get(n * sizeof(Tp), &p));
    bag.push_back(p);
    HIPCHK(c, hipMemcpyAsync(p, host, n * sizeof(Tp), hipMemcpyHostToDevice, st));
    *dev = reinterpret_cast<const Tp *>(p);
    return 0;
}
// Designs [lo, lo + nDesign) of the caller's batch: memberOff / stationOff / capOff are the batch's own (absolute) host
// arrays, the descriptor arrays are sliced here.  shared == NULL: the offsets of the slice are uploaded by this call.
static void launch_scan(hipStream_t st, const GeomArgs &A) {
    static const int scan_t = getenv("RAFTX_SCAN_T") ? atoi(getenv("RAFTX_SCAN_T")) : 1024;     // tuning: 256 | 1024 threads
    if (scan_t == 256) hipLaunchKernelGGL(k_geom_scan_t<256>, dim3(1), dim3(256), 0, st, A);
    else hipLaunchKernelGGL(k_geom_scan_t<1024>, dim3(1), dim3(1024), 0, st, A);
}
static int build_phase1(raftx_ctx *c, hipStream_t sCopy, hipStream_t sPrep, int lo, int nDesign, const int64_t *memberOff,
                        const double *members, const int64_t *stationOff, const double *stations, const int64_t *capOff,
                        const double *caps, const double *pose, double rho, double g, int nw, const double *k, int add_mask,
                        const double *M0, const double *B0, const double *C0, const double *MBw, const double *Fz_moor,
                        const DevOffsets *shared, const double *k_dev = nullptr, const VariantSrc *var = nullptr) {
    RangeScope range_("build phase 1: descriptor H2D, member pass, scans (enqueue)");
    BuildJob &J = c->job;
    if (var && !var->prog) var = nullptr;
    if (nDesign < 0 || lo < 0 || !memberOff || (!members && !var) || !stationOff || (!stations && !var) || !M0 || !B0 || !C0)
        FAIL(c, "build_designs: bad arguments");
    if (!var && (capOff == nullptr) != (caps == nullptr)) FAIL(c, "build_designs: capOff and caps must be given together");
    if (nw < 1 || nw > MAX_NW) FAIL(c, "build_designs: nw=%d outside 1..%d", nw, MAX_NW);
    const int64_t m0 = memberOff[lo], m1 = memberOff[lo + nDesign], nMember = m1 - m0;
    if (nMember < 0 || m0 < 0) FAIL(c, "build_designs: member offsets not monotone");
    const int64_t s0 = stationOff[m0], s1 = stationOff[m1];
    if (s1 < s0 || s0 < 0) FAIL(c, "build_designs: station offsets not monotone");
    const int64_t c0 = capOff ? capOff[m0] : 0, c1 = capOff ? capOff[m1] : 0;
    if (c1 < c0 || c0 < 0) FAIL(c, "build_designs: cap offsets not monotone");
    HIPCHK(c, hipSetDevice(c->device));
    free_list(c, c->design_allocs);               // callers guarantee that nothing in flight reads the previous tables
    free_list(c, J.tmp);
    c->have_designs = false;
    c->bem_ready = false;
    c->g_n = 0;
    if (pin_reserve(c, (size_t)nDesign + 9)) return -2;
    std::vector<void *> &tmp = J.tmp;
    GeomArgs &A = J.A;
    memset(&A, 0, sizeof(A));
    A.nDesign = nDesign;
    A.nMember = nMember;
    A.rho = rho; A.g = g; A.nw = nw; A.add_mask = add_mask;
    A.mbase = m0; A.sbase = s0; A.cbase = c0;
    A.hostOut = c->pin;
    memset(c->pin, 0, 9 * sizeof(long long));
    J.nDesign = nDesign; J.nw = nw; J.add_mask = add_mask; J.nMember = nMember;
    int rc = 0;
    if (shared) {
        A.memberOff = shared->memberOff + lo;
        A.stationOff = shared->stationOff + m0;
        A.capOff = capOff ? shared->capOff + m0 : nullptr;
    } else {
        rc |= upload_on(c, sCopy, tmp, memberOff + lo, (size_t)nDesign + 1, &A.memberOff);
        rc |= upload_on(c, sCopy, tmp, stationOff + m0, (size_t)nMember + 1, &A.stationOff);
        if (capOff) rc |= upload_on(c, sCopy, tmp, capOff + m0, (size_t)nMember + 1, &A.capOff);
    }
    ExpandArgs E;
    memset(&E, 0, sizeof(E));
    if (var) {
        // variants of one base unit: only their parameters cross the bus (nP doubles per design); the descriptors are
        // written in HBM by k_geom_expand, enqueued below on sPrep ahead of the member pass
        const VariantProg &P = *var->prog;
        if (nMember != (int64_t)nDesign * P.nM || s1 - s0 != (int64_t)nDesign * P.nSt || c1 - c0 != (int64_t)nDesign * P.nCap)
            FAIL(c, "build_designs: offsets do not describe %d variants of the program's base unit", nDesign);
        void *pg = nullptr, *ps = nullptr, *pc = nullptr;
        HIPCHK(c, c->pool.get(std::max<size_t>((size_t)nMember * RAFTX_GM_N, 1) * sizeof(double), &pg));
        tmp.push_back(pg);
        HIPCHK(c, c->pool.get(std::max<size_t>((size_t)(s1 - s0) * RAFTX_GS_N, 1) * sizeof(double), &ps));
        tmp.push_back(ps);
        HIPCHK(c, c->pool.get(std::max<size_t>((size_t)(c1 - c0) * RAFTX_GC_N, 1) * sizeof(double), &pc));
        tmp.push_back(pc);
        rc |= upload_on(c, sCopy, tmp, var->params + (size_t)lo * P.nP, (size_t)nDesign * P.nP, &E.params);
        expand_args(P, nDesign, reinterpret_cast<double *>(pg), reinterpret_cast<double *>(ps), reinterpret_cast<double *>(pc), E);
        A.gm = E.gm_out;
        A.gs = E.gs_out;
        if (capOff) A.caps = E.gc_out;
    } else {
        rc |= upload_on(c, sCopy, tmp, members + (size_t)m0 * RAFTX_GM_N, (size_t)nMember * RAFTX_GM_N, &A.gm);
        rc |= upload_on(c, sCopy, tmp, stations + (size_t)s0 * RAFTX_GS_N, (size_t)(s1 - s0) * RAFTX_GS_N, &A.gs);
    }
    rc |= upload_on(c, sCopy, tmp, pose ? pose + (size_t)lo * 6 : nullptr, pose ? (size_t)nDesign * 6 : 0, &A.pose);
    if (capOff && !var) {
        if (c1 > c0) rc |= upload_on(c, sCopy, tmp, caps + (size_t)c0 * RAFTX_GC_N, (size_t)(c1 - c0) * RAFTX_GC_N, &A.caps);
        else {                                        // no caps in this slice: a valid, never-read address
            void *p_ = nullptr;
            HIPCHK(c, c->pool.get(RAFTX_GC_N * sizeof(double), &p_));
            tmp.push_back(p_);
            A.caps = reinterpret_cast<const double *>(p_);
        }
    }
    if (k_dev) A.k = k_dev;                       // wave numbers already resident (the sweep crossing's sea-state tables)
    else rc |= upload_on(c, sCopy, c->design_allocs, k, k ? (size_t)nw : 0, &A.k);
    const double *M0c = nullptr, *C0c = nullptr;
    rc |= upload_on(c, sCopy, c->design_allocs, M0 + (size_t)lo * 36, (size_t)nDesign * 36, &M0c);
    rc |= upload_on(c, sCopy, c->design_allocs, C0 + (size_t)lo * 36, (size_t)nDesign * 36, &C0c);
    rc |= upload_on(c, sCopy, c->design_allocs, B0 + (size_t)lo * 36, (size_t)nDesign * 36, &J.B0d);
    rc |= upload_on(c, sCopy, c->design_allocs, MBw ? MBw + (size_t)lo * 72 * nw : nullptr, MBw ? (size_t)nDesign * 72 * nw : 0, &J.MBwd);
    if (Fz_moor) rc |= upload_on(c, sCopy, tmp, Fz_moor + lo, (size_t)nDesign, &A.Fz);
    if (rc) return -2;
    J.M0d = const_cast<double *>(M0c);
    J.C0d = const_cast<double *>(C0c);
    A.M0 = J.M0d;
    A.C0 = J.C0d;
    HIPCHK(c, hipEventRecord(c->evUp, sCopy));
    HIPCHK(c, hipStreamWaitEvent(sPrep, c->evUp, 0));
    if (var && nDesign > 0) {
        // The expansion is 0.09 ms of chip time.  On the preparation stream, beside the fused kernel of the batch before --
        // whose waves own every register of every CU -- its workgroups are handed out a few at a time over that whole
        // kernel.  RAFTX_EXPAND_STREAM=1 puts it on a HIGH-priority stream of its own (the dispatcher then takes its
        // workgroups first as slots free up; the member pass waits for it by event).  Same box, K = 40
        // (gpurun_out/r05_prio): own stream 3.12 ms per step with the fused kernel at 2.81 ms, preparation stream 3.14 /
        // 2.785, host-made descriptors uploaded by DMA 3.075 / 2.76 -- the 0.09 ms of stores land inside the running fused
        // kernel either way; the default keeps that kernel least disturbed.
        static const bool own_stream = getenv("RAFTX_EXPAND_STREAM") && atoi(getenv("RAFTX_EXPAND_STREAM"));
        if (own_stream && sPrep != c->stream) {
            if (!c->sExp) {
                int least = 0, greatest = 0;
                if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
                    HIPCHK(c, hipStreamCreateWithPriority(&c->sExp, hipStreamNonBlocking, greatest));
                else
                    HIPCHK(c, hipStreamCreateWithFlags(&c->sExp, hipStreamNonBlocking));
                HIPCHK(c, hipEventCreateWithFlags(&c->evExp, hipEventDisableTiming));
            }
            HIPCHK(c, hipStreamWaitEvent(c->sExp, c->evUp, 0));
            launch_expand(E, c->sExp);
            HIPCHK(c, hipEventRecord(c->evExp, c->sExp));
            HIPCHK(c, hipStreamWaitEvent(sPrep, c->evExp, 0));
        } else {
            launch_expand(E, sPrep);
        }
    }
    // device-side scratch; on a pooled block a memset on sPrep is ordered before the kernels that use it
    int *errd = nullptr;
    {
        std::vector<void *> &tb = tmp;
        auto alloc = [&](size_t bytes, void **out, std::vector<void *> &bag) -> int {
            HIPCHK(c, c->pool.get(bytes ? bytes : 8, out));
            bag.push_back(*out);
            return 0;
        };
        void *p_[13] = {nullptr};
        if (alloc((size_t)nMember * sizeof(int), &p_[0], tb) || alloc((size_t)nMember * sizeof(int), &p_[1], tb) ||
            alloc((size_t)nMember * MP_N * sizeof(double), &p_[4], tb) || alloc((size_t)nMember * MH_N * sizeof(double), &p_[5], tb) ||
            alloc((size_t)nMember * MI_N * sizeof(double), &p_[6], tb) || alloc(4 * sizeof(int), &p_[7], tb) ||
            alloc((size_t)nMember * sizeof(int), &p_[12], tb) ||
            alloc((size_t)nDesign * sizeof(double), &p_[8], tb) || alloc(5 * sizeof(long long), &p_[9], tb) ||
            alloc(((size_t)nDesign + 1) * sizeof(int64_t), &p_[10], c->design_allocs) ||
            alloc(((size_t)nDesign + 1) * sizeof(int64_t), &p_[11], c->design_allocs))
            return -2;
        A.cnt = (int *)p_[0]; A.cntm = (int *)p_[1];
        A.mpose = (double *)p_[4]; A.mhyd = (double *)p_[5]; A.minert = (double *)p_[6];
        errd = (int *)p_[7]; A.err = errd;
        A.drho = (double *)p_[8];
        A.tot = (long long *)p_[9];
        A.off = (int64_t *)p_[10]; A.cmoff = (int64_t *)p_[11];
        A.mdesign_w = (int *)p_[12]; A.mdesign = A.mdesign_w;
    }
    // the member offsets of every design, checked here (the kernels walk them): monotone, inside the slice; and the
    // largest design's members / stations (they size the LDS of k_geom_design)
    J.maxMem = J.maxSta = 0;
    for (int d = 0; d < nDesign; d++) {
        const int64_t a = memberOff[lo + d], b = memberOff[lo + d + 1];
        if (b < a || a < m0 || b > m1) FAIL(c, "member offsets not monotone at design %d", lo + d);
        if (stationOff[b] < stationOff[a]) FAIL(c, "build_designs: station offsets not monotone");
        J.maxMem = std::max(J.maxMem, b - a);
        J.maxSta = std::max(J.maxSta, stationOff[b] - stationOff[a]);
    }
    hipLaunchKernelGGL(k_geom_zero, dim3((unsigned)(nDesign / 256 + 1)), dim3(256), 0, sPrep, A);
    HIPCHK(c, hipEventRecord(c->evZ, sPrep));
    HIPCHK(c, hipEventRecord(c->evG2, sPrep));
    A.mgrid = 0;
    if (nMember > 0) {
        // member kernels on a (member position, design) grid when that wastes few threads: wavefronts of like members
        const int64_t maxMem = J.maxMem;
        static const bool flat_members = getenv("RAFTX_GEOM_FLAT_MEMBERS") != nullptr;
        A.mgrid = (!flat_members && nDesign >= 64 && maxMem * nDesign <= nMember + nMember / 4) ? (int)maxMem : 0;
        const int64_t nThread = A.mgrid > 0 ? (int64_t)A.mgrid * nDesign : nMember;
        hipLaunchKernelGGL(k_geom_member, dim3((unsigned)((nThread + 127) / 128)), dim3(128), 0, sPrep, A);
        if (add_mask & RAFTX_TRIM_BALLAST) {              // heave trim: density correction, then the inertia again
            hipLaunchKernelGGL(k_geom_trim, dim3((unsigned)(nDesign / 128 + 1)), dim3(128), 0, sPrep, A);
            hipLaunchKernelGGL(k_geom_reinertia, dim3((unsigned)((nThread + 127) / 128)), dim3(128), 0, sPrep, A);
        }
    }
    // the member -> platform reductions need the member pass only: they run on a side stream beside the scans, off the
    // stream the fused kernel waits on (phase 2 orders the design kernel behind them)
    if (dev_alloc(c, c->design_allocs, (size_t)nDesign * 36, &A.Ch) || dev_alloc(c, c->design_allocs, (size_t)nDesign * 6, &A.Wh) ||
        dev_alloc(c, c->design_allocs, (size_t)nDesign * 36, &A.Ms) || dev_alloc(c, c->design_allocs, (size_t)nDesign * 36, &A.Cs) ||
        dev_alloc(c, c->design_allocs, (size_t)nDesign * 6, &A.Ws) || dev_alloc(c, c->design_allocs, (size_t)nDesign * RAFTX_SP_N, &A.props))
        return -2;
    J.reduce_stream = nullptr;
    static const bool reduce_late = getenv("RAFTX_REDUCE_PHASE2") != nullptr;      // tuning: on the ctx stream, before the design kernel
    J.reduce_late = reduce_late;
    const bool side = nDesign > 0 && !reduce_late && sPrep != c->stream;          // crossings: a stream of its own per block context
    if (side) HIPCHK(c, hipEventRecord(c->evMem, sPrep));
    // the scan first: streams share hardware queues, and a reduction submitted ahead of it on the same queue would sit on
    // the path to the totals (the host waits for them before it can size and launch the generation)
    if (nDesign > 0 && side) launch_scan(sPrep, A);
    if (side) {                                           // ... and its markers, for the same reason
        HIPCHK(c, hipEventRecord(c->evG3, sPrep));
        HIPCHK(c, hipEventRecord(c->evTot, sPrep));
    }
    if (nDesign > 0 && !reduce_late) {
        hipStream_t sRed = sPrep;
        if (side) {
            if (!c->sAux) HIPCHK(c, hipStreamCreateWithFlags(&c->sAux, hipStreamNonBlocking));
            sRed = c->sAux;
            HIPCHK(c, hipStreamWaitEvent(sRed, c->evMem, 0));
        }
        hipLaunchKernelGGL(k_geom_reduce, dim3((unsigned)(((size_t)nDesign * 3 + 63) / 64)), dim3(64), 0, sRed, A);
        if (sRed != sPrep) {
            HIPCHK(c, hipEventRecord(c->evRed, sRed));
            J.reduce_stream = sRed;
        }
    }
    // totals, error flags and design offsets reach the host through the kernels' own stores into page-locked memory
    // (A.hostOut): a D2H copy of them would queue on the DMA engine behind a bulk download of the previous batch
    if (!side) {
        if (nDesign > 0) launch_scan(sPrep, A);
        HIPCHK(c, hipEventRecord(c->evG3, sPrep));
        HIPCHK(c, hipEventRecord(c->evTot, sPrep));
    }
    J.active = true;
    return 0;
}

// Phase 2: waits (host) for the totals of phase 1, sizes the strip tables, and enqueues strip generation, the MacCamy-
// Fuchs table and the per-design reduction on the ctx stream, ordered behind phase 1 by evTot.  Does not wait for them.
// sGen: the stream the generation kernels go to (null: the ctx stream).  The sweep crossing gives the preparation stream
// for every block but the first, so that a block's tables are generated WHILE the fused kernel of the block before it
// runs (they fill the CUs its last residency round leaves idle); the ctx stream is ordered behind them by evG1.
static void launch_design(raftx_ctx *c, hipStream_t st) {
    BuildJob &J = c->job;
    hipLaunchKernelGGL(k_geom_design, dim3((unsigned)J.nDesign), dim3(GD_T), J.gen_lds, st, J.A);
}
// crossing: 0 = raftx_build_designs (tables + ABI copy); bits: 1 = a sweep crossing (no ABI copy), 2 = its tables may be left
// to the fused kernel (RAFTX_FUSED_GEN=1), 4 = other crossings are in flight (its member pass ran a step ago: the generation
// adds the design matrices up itself)
static int build_phase2(raftx_ctx *c, int64_t *stripOffsets, hipStream_t sGen = nullptr, int crossing = 0) {
    RangeScope range_("build phase 2: wait for totals, strip tables + statics (enqueue)");
    BuildJob &J = c->job;
    if (!J.active) FAIL(c, "build_designs: phase 2 without phase 1");
    GeomArgs &A = J.A;
    const int nDesign = J.nDesign, nw = J.nw;
    for (;;) {                                        // spin: the blocking wait costs ~0.2 ms of wake-up latency per block
        const hipError_t q = hipEventQuery(c->evTot);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) HIPCHK(c, q);
    }
    HIPCHK(c, hipGetLastError());
    const int *bad = reinterpret_cast<const int *>(c->pin + 3);
    if (bad[3] > 0) FAIL(c, "member %d: needs 2..%d stations, dlsMax > 0 and length > 0", bad[3] - 1, GEOM_MAX_STATIONS);
    if (bad[3] < 0) FAIL(c, "build_designs: member %d is MacCamy-Fuchs but no wave numbers were given", -bad[3] - 1);
    if (bad[0]) FAIL(c, "member %d: cap/bulkhead layout not supported (the reference raises here too)", bad[0] - 1);
    if (bad[1]) FAIL(c, "design %d: ballast trim needs some ballast volume", bad[1] - 1);
    const size_t nStrips = (size_t)c->pin[0], nRows = (size_t)c->pin[1];
    const int maxS = (int)c->pin[2];
    if (stripOffsets) memcpy(stripOffsets, c->pin + 8, ((size_t)nDesign + 1) * sizeof(int64_t));
    c->hS.resize((size_t)nDesign);
    for (int d = 0; d < nDesign; d++) c->hS[(size_t)d] = (int)(c->pin[8 + d + 1] - c->pin[8 + d]);
    std::vector<void *> &tmp = J.tmp;
    // RAFTX_FUSED_GEN=1: sweep crossings build their tables inside the fused kernel (raftx_fusedgen.h).  Measured and NOT
    // the default (profiles/r06_experiments/fused_generation_ab.txt): bit-identical, the gap between two fused kernels
    // shrinks from 0.30 to 0.09 ms, but the kernel grows by 0.50 ms -- the ~50 us of dependent loads per design are not
    // hidden by the seven other waves of the CU (each is bound by its own dependency chains, not by issue slots).
    const char *fg_ = getenv("RAFTX_FUSED_GEN");          // (read per call: tests switch it inside one process)
    const bool fused_gen = fg_ && atoi(fg_);
    const bool defer = (crossing & 2) && fused_gen && nDesign > 0 && nRows == 0;
    J.gen_deferred = false;
    // the ABI copy of the strip records (raftx_fetch_strips) is for raftx_build_designs; a sweep crossing never
    // fetches it: 137 MB of stores per 10 000 designs less between two fused kernels (k_geom_design checks the pointer)
    A.abi = nullptr;
    if ((!crossing && dev_alloc(c, c->design_allocs, nStrips * NF, &A.abi)) || dev_alloc(c, c->design_allocs, nStrips * DS_N, &A.ds) ||
        dev_alloc(c, c->design_allocs, nStrips, &A.dsi) || dev_alloc(c, c->design_allocs, nRows * 3, &A.mcfaux) ||
        dev_alloc(c, c->design_allocs, nRows * 2 * (size_t)nw, &A.cm) ||
        dev_alloc(c, c->design_allocs, (size_t)nDesign * 36, &A.A))
        return -2;
    (void)tmp;
    const size_t gd_lds = geom_design_lds(maxS, (int)J.maxSta, (int)J.maxMem);
    if (gd_lds > 160 * 1024)
        FAIL(c, "build_designs: a design has %d submerged strips (at most %d supported)", maxS, (int)((160 * 1024 - 16) / (8 * (GD_ROW + 2) + 12)));
    if (gd_lds > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_geom_design), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gd_lds));
    J.gen_lds = gd_lds;
    if (!sGen || defer) sGen = c->stream;
    HIPCHK(c, hipStreamWaitEvent(sGen, c->evTot, 0));
    HIPCHK(c, hipEventRecord(c->evG0, sGen));
    if (defer) {
        // the ctx stream (where the fused kernel goes) behind everything the generation reads: scans, reductions
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->evTot, 0));
        if (J.reduce_stream) HIPCHK(c, hipStreamWaitEvent(c->stream, c->evRed, 0));
        if (J.reduce_late) hipLaunchKernelGGL(k_geom_reduce, dim3((unsigned)(((size_t)nDesign * 3 + 63) / 64)), dim3(64), 0, c->stream, A);
        J.gen_deferred = true;
    } else if (nDesign > 0) {
        if (J.reduce_late) hipLaunchKernelGGL(k_geom_reduce, dim3((unsigned)(((size_t)nDesign * 3 + 63) / 64)), dim3(64), 0, sGen, A);
        // A sweep crossing in flight behind others (its member pass and reductions ran a step ago): the generation adds its
        // design's matrices up itself -- one kernel and one launch gap less on the path between two fused kernels (same
        // additions in the same order: bit-identical).  RAFTX_ADDUP_KERNEL=1 keeps k_geom_addup.
        static const bool addup_kernel = getenv("RAFTX_ADDUP_KERNEL") && atoi(getenv("RAFTX_ADDUP_KERNEL"));
        A.addup_in_design = ((crossing & 4) && !addup_kernel && nRows == 0 && !A.abi) ? 1 : 0;
        if (A.addup_in_design && J.reduce_stream) HIPCHK(c, hipStreamWaitEvent(sGen, c->evRed, 0));
        launch_design(c, sGen);
        if (!A.addup_in_design) {
            // the reductions of phase 1 (side stream) are not waited for until their results are added up: streams share
            // hardware queues, and a reduction that ended up behind the scan would otherwise hold the generation back
            if (J.reduce_stream) HIPCHK(c, hipStreamWaitEvent(sGen, c->evRed, 0));
            hipLaunchKernelGGL(k_geom_addup, dim3((unsigned)(((size_t)nDesign * 36 + 255) / 256)), dim3(256), 0, sGen, A);
        }
    }
    if (nRows > 0)                                        // after k_geom_design: it leaves (R, Ca) of the MacCamy-Fuchs strips
        hipLaunchKernelGGL(k_geom_mcf, dim3((unsigned)nRows, (unsigned)((nw + 63) / 64)), dim3(64), 0, sGen, A, (int64_t)nRows);
    HIPCHK(c, hipEventRecord(c->evG1, sGen));
    if (sGen != c->stream) HIPCHK(c, hipStreamWaitEvent(c->stream, c->evG1, 0));
    DevTables &T = c->T;
    T.nDesign = nDesign;
    T.off = A.off;
    T.ds = A.ds;
    T.dsi = A.dsi;
    T.M0 = J.M0d;
    T.C0 = J.C0d;
    T.B0 = J.B0d;
    T.MBw = J.MBwd;
    T.cmoff = nRows ? A.cmoff : nullptr;
    T.cm = nRows ? A.cm : nullptr;
    c->maxS = maxS;
    c->nw_designs = nw;
    c->have_designs = true;
    c->g_n = nDesign;
    c->g_nStrips = nStrips;
    c->g_nRows = nRows;
    c->g_abi = A.abi;
    c->g_cm = A.cm;
    c->g_A = A.A; c->g_Ch = A.Ch; c->g_Wh = A.Wh; c->g_props = A.props;
    c->g_Ms = A.Ms; c->g_Cs = A.Cs; c->g_Ws = A.Ws;
    return 0;
}
// kernel time of a finished build (both phases), and its scratch back to the pool
static int build_retire(raftx_ctx *c, double *ms_out) {
    BuildJob &J = c->job;
    float a = 0.f, b = 0.f;
    if (J.active) {
        HIPCHK(c, hipEventElapsedTime(&a, c->evG2, c->evG3));
        HIPCHK(c, hipEventElapsedTime(&b, c->evG0, c->evG1));
    }
    if (ms_out) *ms_out = (double)a + (double)b;
    free_list(c, J.tmp);
    J.active = false;
    return 0;
}

extern "C" int raftx_build_designs(raftx_ctx *c, int nDesign, const int64_t *memberOff, const double *members,
                                   const int64_t *stationOff, const double *stations, const int64_t *capOff,
                                   const double *caps, const double *pose, double rho, double g, int nw, const double *k,
                                   int add_mask, const double *M0, const double *B0, const double *C0, const double *MBw,
                                   const double *Fz_moor, int64_t *stripOffsets) {
    if (!c) return -1;
    if (!stripOffsets) FAIL(c, "build_designs: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int rc = build_phase1(c, c->stream, c->stream, 0, nDesign, memberOff, members, stationOff, stations, capOff, caps, pose, rho,
                          g, nw, k, add_mask, M0, B0, C0, MBw, Fz_moor, nullptr);
    if (!rc) rc = build_phase2(c, stripOffsets);
    const hipError_t e = hipStreamSynchronize(c->stream);     // host buffers may be released after return
    if (rc) {
        free_list(c, c->job.tmp);
        c->job.active = false;
        c->have_designs = false;
        return rc;
    }
    HIPCHK(c, e);
    HIPCHK(c, hipGetLastError());
    double ms = 0.0;
    if (build_retire(c, &ms)) return -2;
    c->last_ms = ms;                                 // the kernels; allocations and descriptor H2D are outside
    return 0;
}

extern "C" int raftx_fetch_strips(raftx_ctx *c, double *strips, raftx_c128 *cm) {
    if (!c) return -1;
    if (!c->g_n || !c->have_designs) FAIL(c, "fetch_strips: no raftx_build_designs call on this ctx");
    HIPCHK(c, hipSetDevice(c->device));
    if (strips && c->g_nStrips && !c->g_abi)
        FAIL(c, "fetch_strips: the tables of this batch were generated inside the fused kernel of a sweep crossing (no ABI copy of the strip records)");
    if (strips && c->g_nStrips)
        HIPCHK(c, hipMemcpyAsync(strips, c->g_abi, c->g_nStrips * NF * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (cm && c->g_nRows)
        HIPCHK(c, hipMemcpyAsync(cm, c->g_cm, c->g_nRows * 2 * (size_t)c->nw_designs * sizeof(cplx), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int raftx_fetch_statics(raftx_ctx *c, double *A_morison, double *C_hydro, double *W_hydro, double *M_struc,
                                   double *C_struc, double *W_struc, double *props) {
    if (!c) return -1;
    if (!c->g_n || !c->have_designs) FAIL(c, "fetch_statics: no raftx_build_designs call on this ctx");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->g_n;
    if (A_morison) HIPCHK(c, hipMemcpyAsync(A_morison, c->g_A, n * 36 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (C_hydro) HIPCHK(c, hipMemcpyAsync(C_hydro, c->g_Ch, n * 36 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (W_hydro) HIPCHK(c, hipMemcpyAsync(W_hydro, c->g_Wh, n * 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (props) HIPCHK(c, hipMemcpyAsync(props, c->g_props, n * RAFTX_SP_N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (M_struc) HIPCHK(c, hipMemcpyAsync(M_struc, c->g_Ms, n * 36 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (C_struc) HIPCHK(c, hipMemcpyAsync(C_struc, c->g_Cs, n * 36 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (W_struc) HIPCHK(c, hipMemcpyAsync(W_struc, c->g_Ws, n * 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// The sea-state tables of one set of cases into device memory of ctx `c` (allocations into `bag`, fields into `T`), copied
// on `st`, which has drained when this returns (the depth constants are host temporaries).
static int upload_case_tables(raftx_ctx *c, hipStream_t st, std::vector<void *> &bag, DevTables &T, int nCase, int nHead, int nw,
                              const double *w, const double *k, double depth, double rho, double g, const double *zeta,
                              const double *beta) {
    // per-bin depth constants, computed once on the host in full libm precision.  The kernels derive
    // the depth regime (k == 0 / deep / finite) from k themselves, with the same rule.
    std::vector<double> csh(nw), cch(nw);
    for (int i = 0; i < nw; i++) {
        double kh = k[i] * depth;
        if (k[i] == 0.0 || kh > 89.4) {   // helpers.py:211-218: Sh = 1, Ch = Cc = 99999  /  Sh = Ch = e^{kz}, Cc = e^{kz} + e^{-k(z+2h)}
            csh[i] = cch[i] = 1.0;
        } else {                          // helpers.py:219-222 written with decaying exponentials only
            double e2kh = exp(-2.0 * kh);
            csh[i] = 1.0 / (-expm1(-2.0 * kh));
            cch[i] = 1.0 / (1.0 + e2kh);
        }
    }
    T.nCase = nCase;
    T.nHead = nHead;
    T.nw = nw;
    T.depth = depth;
    T.rho = rho;
    T.g = g;
    int rc = 0;
    rc |= upload_on(c, st, bag, w, (size_t)nw, &T.w);
    rc |= upload_on(c, st, bag, k, (size_t)nw, &T.k);
    rc |= upload_on(c, st, bag, csh.data(), (size_t)nw, &T.csh);
    rc |= upload_on(c, st, bag, cch.data(), (size_t)nw, &T.cch);
    rc |= upload_on(c, st, bag, zeta, (size_t)nCase * nHead * nw, &T.zeta);
    rc |= upload_on(c, st, bag, beta, (size_t)nCase * nHead, &T.beta);
    const hipError_t e = hipStreamSynchronize(st);        // also on failure: csh / cch go out of scope
    if (rc) return -2;
    HIPCHK(c, e);
    return 0;
}

extern "C" int raftx_upload_cases(raftx_ctx *c, int nCase, int nHead, int nw, const double *w, const double *k,
                                  double depth, double rho, double g, const double *zeta, const double *beta) {
    RangeScope range_("raftx_upload_cases: H2D");
    if (!c) return -1;
    if (nCase < 0 || nHead < 1 || nw < 1 || !w || !k || !zeta || !beta) FAIL(c, "upload_cases: bad arguments");
    if (nw > MAX_NW) FAIL(c, "nw=%d exceeds the %d bins per workgroup supported by this build", nw, MAX_NW);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_list(c, c->case_allocs);
    c->have_cases = false;
    c->bem_ready = false;
    if (int rc = upload_case_tables(c, c->stream, c->case_allocs, c->T, nCase, nHead, nw, w, k, depth, rho, g, zeta, beta)) return rc;
    c->have_cases = true;
    return 0;
}

#define LDS_LIMIT (160 * 1024)

// Launch shape: NB bins per lane x threads per workgroup (one workgroup per pair).  Defaults
// (measured on MI355X, profiles/): two waves per SIMD with 2 bins per lane beat both the
// 4-bins-per-lane / 1-wave-per-SIMD and the 1-bin-per-lane shapes at nw = 200.
//   nw <=  64 : (1,  64)    one wave per pair, no s_barrier anywhere in the kernel
//   nw <= 128 : (2,  64)
//   nw <= 256 : (2, 128)
//   nw <= 512 : (2, 256)
//   larger    : (2..4, 512)
// RAFTX_SHAPE="nb,threads" overrides (tuning only; must be one of the instantiated shapes).
struct Shape {
    int nb, threads;
};
#ifndef RAFTX_MINB128
#define RAFTX_MINB128 2          // waves per SIMD the default 200-bin shape is compiled for (tuning builds override)
#endif
#define SHAPES(X) X(1, 64, 1) X(2, 64, 1) X(4, 64, 1) X(2, 128, RAFTX_MINB128) X(1, 256, 2) X(2, 256, 2) X(2, 512, 1) X(3, 512, 1) X(4, 512, 1)
// MAXT template value of the kernel instantiated for a shape
static int shape_maxt(Shape sh) {
#define X(NB_, MT_, MB_) if (sh.nb == NB_ && sh.threads == MT_) return MT_;
    SHAPES(X)
#undef X
    return 0;
}
// waves per SIMD the kernel of a shape is compiled for
static int shape_minb(Shape sh) {
#define X(NB_, MT_, MB_) if (sh.nb == NB_ && sh.threads == MT_) return MB_;
    SHAPES(X)
#undef X
    return 1;
}
static bool shape_ok(Shape sh, int nw) {
#define X(NB_, MT_, MB_) if (sh.nb == NB_ && sh.threads == MT_) return (long)NB_ * MT_ >= nw;
    SHAPES(X)
#undef X
    return false;
}
static Shape pick_shape(int nw) {
    static const char *env = getenv("RAFTX_SHAPE");
    if (env) {
        Shape sh = {0, 0};
        if (sscanf(env, "%d,%d", &sh.nb, &sh.threads) == 2 && shape_ok(sh, nw)) return sh;
    }
    if (nw <= 64) return {1, 64};
    if (nw <= 128) return {2, 64};
    if (nw <= 256) return {2, 128};
    if (nw <= 512) return {2, 256};
    int nb = (nw + 511) / 512;
    return {nb < 2 ? 2 : nb, 512};
}

template <typename K>
static int prep_lds(raftx_ctx *c, K kernel, size_t bytes) {
    if (bytes > LDS_LIMIT)
        FAIL(c, "a design has %d submerged strips / nw=%d: %zu B of LDS needed, 160 KiB available", c->maxS, c->T.nw, bytes);
    if (bytes > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)bytes));
    return 0;
}

// expands BODY(NB, MAXT, MINB) for the shape sh
#define DISPATCH_SHAPE(sh, BODY)                                                     \
    do {                                                                             \
        bool hit_ = false;                                                           \
        SHAPES(DISPATCH_ONE_)                                                        \
        if (!hit_) FAIL(c, "no kernel for shape %d x %d", (sh).nb, (sh).threads);    \
    } while (0)

static int check_ready(raftx_ctx *c) {
    if (!c) return -1;
    if (!c->have_designs) FAIL(c, "no designs uploaded");
    if (!c->have_cases) FAIL(c, "no cases uploaded");
    if ((c->T.MBw || c->T.cm) && c->nw_designs != c->T.nw)
        FAIL(c, "nw mismatch between designs (%d) and cases (%d)", c->nw_designs, c->T.nw);
    return 0;
}

// device scratch that lives for one call
struct Scratch {
    raftx_ctx *c;
    std::vector<void *> bag;
    explicit Scratch(raftx_ctx *c_) : c(c_) {}
    ~Scratch() {
        (void)hipStreamSynchronize(c->stream);      // blocks go back to the pool only when nothing in flight uses them
        free_list(c, bag);
    }
    template <typename Tp>
    Tp *alloc(size_t n) {
        void *p = nullptr;
        if (n == 0) return nullptr;
        if (c->pool.get(n * sizeof(Tp), &p) != hipSuccess) return nullptr;
        bag.push_back(p);
        return reinterpret_cast<Tp *>(p);
    }
};

#define D2H(c, dst, src, bytes) HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (c)->stream))
#define H2D(c, dst, src, bytes) HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (c)->stream))

static int finish_timed(raftx_ctx *c) {
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_ms = ms;
    return 0;
}

extern "C" int raftx_excitation(raftx_ctx *c, raftx_c128 *F_iner) {
    if (check_ready(c)) return -1;
    if (!F_iner) FAIL(c, "excitation: F_iner is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    const DevTables &T = c->T;
    size_t npair = (size_t)T.nDesign * T.nCase;
    size_t n = npair * T.nHead * 6 * T.nw;
    Scratch sc(c);
    cplx *dF = sc.alloc<cplx>(n);
    if (n && !dF) FAIL(c, "excitation: device allocation failed");
    const Shape sh = pick_shape(T.nw);
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
#define DISPATCH_ONE_(NB_, MT_, MB_)                                                                                  \
    if (!hit_ && sh.nb == NB_ && sh.threads == MT_) {                                                                 \
        hit_ = true;                                                                                                  \
        hipLaunchKernelGGL((k_excitation<NB_, MT_, MB_>), dim3((unsigned)(npair * T.nHead)), dim3(sh.threads), 0,     \
                           c->stream, T, dF);                                                                         \
    }
    if (npair) DISPATCH_SHAPE(sh, _);
#undef DISPATCH_ONE_
    if (finish_timed(c)) return -2;
    if (n) D2H(c, F_iner, dF, n * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// one drag linearisation on device arrays (enqueued on the ctx stream between ev0 and the caller's ev1)
static int linearize_enqueue(raftx_ctx *c, const cplx *dXi, double *dB, cplx *dF, bool timed = true) {
    const DevTables &T = c->T;
    const size_t npair = (size_t)T.nDesign * T.nCase;
    const Shape sh = pick_shape(T.nw);
    const size_t lds = lds_bytes(c->maxS, 0, sh.threads / 64, stage_policy(sh.nb, shape_maxt(sh)));
#define DISPATCH_ONE_(NB_, MT_, MB_)                                                                                  \
    if (!hit_ && sh.nb == NB_ && sh.threads == MT_) {                                                                 \
        hit_ = true;                                                                                                  \
        if (prep_lds(c, k_linearize<NB_, MT_, MB_>, lds)) return -1;                                                  \
        if (timed) HIPCHK(c, hipEventRecord(c->ev0, c->stream));                                                      \
        if (npair)                                                                                                    \
            hipLaunchKernelGGL((k_linearize<NB_, MT_, MB_>), dim3(grid_for_pairs(npair)), dim3(sh.threads), lds,      \
                               c->stream, T, dXi, dB, dF);                                                            \
    }
    DISPATCH_SHAPE(sh, _);
#undef DISPATCH_ONE_
    return 0;
}
extern "C" int raftx_linearize(raftx_ctx *c, const raftx_c128 *Xi, double *B_drag, raftx_c128 *F_drag) {
    if (check_ready(c)) return -1;
    if (!Xi) FAIL(c, "linearize: Xi is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    const DevTables &T = c->T;
    size_t npair = (size_t)T.nDesign * T.nCase;
    Scratch sc(c);
    cplx *dXi = sc.alloc<cplx>(npair * 6 * T.nw);
    double *dB = B_drag ? sc.alloc<double>(npair * 36) : nullptr;
    cplx *dF = F_drag ? sc.alloc<cplx>(npair * T.nHead * 6 * T.nw) : nullptr;
    if (npair && (!dXi || (B_drag && !dB) || (F_drag && !dF))) FAIL(c, "linearize: device allocation failed");
    if (npair) H2D(c, dXi, Xi, npair * 6 * T.nw * sizeof(cplx));
    if (linearize_enqueue(c, dXi, dB, dF)) return -1;
    if (finish_timed(c)) return -2;
    if (dB) D2H(c, B_drag, dB, npair * 36 * sizeof(double));
    if (dF) D2H(c, F_drag, dF, npair * T.nHead * 6 * T.nw * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// the resident strips of one design (host copy of its offsets)
static int strip_count(raftx_ctx *c, int design, int icase, const char *what, int *S) {
    if (check_ready(c)) return -1;
    const DevTables &T = c->T;
    if (design < 0 || design >= T.nDesign || icase < 0 || icase >= T.nCase) FAIL(c, "%s: design / case outside the resident set", what);
    HIPCHK(c, hipSetDevice(c->device));
    int64_t o[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(o, T.off + design, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *S = (int)(o[1] - o[0]);
    return 0;
}
extern "C" int raftx_strip_kinematics(raftx_ctx *c, int design, int icase, raftx_c128 *u, raftx_c128 *ud, raftx_c128 *pDyn) {
    if (!c) return -1;
    int S = 0;
    if (int rc = strip_count(c, design, icase, "strip_kinematics", &S)) return rc;
    const DevTables &T = c->T;
    const size_t n = (size_t)T.nHead * S * T.nw;
    if (!n) return 0;
    Scratch sc(c);
    cplx *du = u ? sc.alloc<cplx>(n * 3) : nullptr, *dud = ud ? sc.alloc<cplx>(n * 3) : nullptr, *dp = pDyn ? sc.alloc<cplx>(n) : nullptr;
    if ((u && !du) || (ud && !dud) || (pDyn && !dp)) FAIL(c, "strip_kinematics: device allocation failed");
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    hipLaunchKernelGGL(k_strip_kinematics, dim3((unsigned)(S * T.nHead)), dim3(256), 0, c->stream, T, design, icase, du, dud, dp);
    if (finish_timed(c)) return -2;
    if (du) D2H(c, u, du, n * 3 * sizeof(cplx));
    if (dud) D2H(c, ud, dud, n * 3 * sizeof(cplx));
    if (dp) D2H(c, pDyn, dp, n * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int raftx_strip_drag(raftx_ctx *c, int design, int icase, const raftx_c128 *Xi, int ih, double *Bmat,
                                raftx_c128 *F_exc_drag) {
    if (!c) return -1;
    int S = 0;
    if (int rc = strip_count(c, design, icase, "strip_drag", &S)) return rc;
    const DevTables &T = c->T;
    if (!Xi) FAIL(c, "strip_drag: Xi is NULL");
    if (ih < 0 || ih >= T.nHead) FAIL(c, "strip_drag: heading outside the resident sea state");
    if (!S) return 0;
    Scratch sc(c);
    cplx *dXi = sc.alloc<cplx>((size_t)6 * T.nw), *dF = F_exc_drag ? sc.alloc<cplx>((size_t)S * 3 * T.nw) : nullptr;
    double *dB = Bmat ? sc.alloc<double>((size_t)S * 9) : nullptr;
    if (!dXi || (F_exc_drag && !dF) || (Bmat && !dB)) FAIL(c, "strip_drag: device allocation failed");
    H2D(c, dXi, Xi, (size_t)6 * T.nw * sizeof(cplx));
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    hipLaunchKernelGGL(k_strip_drag, dim3((unsigned)S), dim3(256), 0, c->stream, T, design, icase, ih, dXi, dB, dF);
    if (finish_timed(c)) return -2;
    if (dB) D2H(c, Bmat, dB, (size_t)S * 9 * sizeof(double));
    if (dF) D2H(c, F_exc_drag, dF, (size_t)S * 3 * T.nw * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

template <typename Tp>
static Tp *dev_alloc(raftx_ctx *c, size_t n) {
    void *p = nullptr;
    if (n == 0) n = 1;
    if (c->pool.get(n * sizeof(Tp), &p) != hipSuccess) return nullptr;
    c->result_allocs.push_back(p);
    return reinterpret_cast<Tp *>(p);
}

// (re)size the ctx-owned result buffers for the current designs x cases
static int ensure_results(raftx_ctx *c, int want_mask, bool need_fe) {
    const DevTables &T = c->T;
    size_t npair = (size_t)T.nDesign * T.nCase;
    size_t nx = npair * T.nHead * 6 * T.nw, nz = npair * 36 * T.nw;
    bool ok = c->rXi && c->r_npair == npair && c->r_nx == nx && c->r_nz == nz &&
              (c->r_mask & want_mask) == want_mask && (!need_fe || c->r_fe);
    if (ok) return 0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_list(c, c->result_allocs);
    c->rXi = c->rFw = c->rZ = c->rFe = nullptr;
    c->rB = nullptr;
    c->rNi = c->rFl = nullptr;
    c->rXi = dev_alloc<cplx>(c, nx);
    c->rNi = dev_alloc<int>(c, npair);
    c->rFl = dev_alloc<int>(c, npair);
    if (want_mask & RAFTX_WANT_BDRAG) c->rB = dev_alloc<double>(c, npair * 36);
    if (want_mask & RAFTX_WANT_FWAVE) c->rFw = dev_alloc<cplx>(c, nx);
    if (want_mask & RAFTX_WANT_Z) c->rZ = dev_alloc<cplx>(c, nz);
    if (need_fe) c->rFe = dev_alloc<cplx>(c, nx);
    if (!c->rXi || !c->rNi || !c->rFl || ((want_mask & RAFTX_WANT_BDRAG) && !c->rB) ||
        ((want_mask & RAFTX_WANT_FWAVE) && !c->rFw) || ((want_mask & RAFTX_WANT_Z) && !c->rZ) || (need_fe && !c->rFe)) {
        free_list(c, c->result_allocs);
        c->rXi = nullptr;
        FAIL(c, "solve_dynamics: device allocation of result buffers failed");
    }
    c->r_npair = npair;
    c->r_nx = nx;
    c->r_nz = nz;
    c->r_mask = want_mask;
    c->r_fe = need_fe;
    return 0;
}

// enqueues the fused fixed point on the ctx stream between ev0 and ev1; does not wait for it
// A fused-kernel launch cut into SLABS of the pair list (a crossing that downloads its responses: slab k's download runs
// while the slabs behind it solve).  The slabs go round-robin to the streams `alts`, none to the ctx stream: pairs are
// independent, and a grid queued on another stream is handed out as soon as the grids before it are exhausted -- the
// workgroups of the next slabs fill the CUs a slab's last residency round leaves idle, so a cut costs no tail (slabs as
// consecutive launches of ONE stream each wait for the slowest pair of the slab before: +0.1-0.15 ms per cut,
// profiles/r02_crossing_splits.txt), and the ctx stream stays free for what the NEXT block needs (its table generation
// runs beside these slabs instead of behind them).  `after(p0, p1, s)` enqueues what follows the pairs [p0, p1) on the
// stream s they were launched on.  The caller joins: `slab_join` orders the ctx stream behind every stream in `alts`.
// A launch that is already split into LDS classes is not cut again: it runs on the ctx stream, one call of `after`.
struct SlabPlan {
    std::vector<size_t> bnd;                                             // 0 = bnd[0] < .. < bnd.back() = pairs of the ctx
    std::vector<hipStream_t> alts;
    size_t next = 0;                                                     // round-robin position, carried from block to block
    std::function<int(size_t, size_t, hipStream_t)> after;
};
static int slab_join(raftx_ctx *c, hipStream_t into, const std::vector<hipStream_t> &alts) {
    for (hipStream_t s : alts) {
        HIPCHK(c, hipEventRecord(c->evJoin, s));                         // a wait captures the record that precedes it: one event serves
        HIPCHK(c, hipStreamWaitEvent(into, c->evJoin, 0));
    }
    return 0;
}
__global__ void k_iota(int n, int *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

// ---- persistent fused launches (raftx_kernels.h: raftx_kp_f<FLAGS>)
typedef void (*kp_fn)(PersistArgs);
static kp_fn kp_kernel(int flags) {
#ifdef RAFTX_KP_ONLY
    return flags == 0 ? raftx_kp_f0 : nullptr;
#else
#define X(F) if (flags == F) return raftx_kp_f##F;
    RAFTX_PERSIST128(X)
#undef X
    return nullptr;
#endif
}
#define KP_RING 64
#define KP_MAX_GRID 2048                 // workgroups of a persistent grid at most (8 per CU x 256 CUs)
static int launch_persistent(raftx_ctx *c, kp_fn kernel, const DevTables &T, const SolveArgs &A, size_t npairs, int threads, size_t lds,
                             int wg_per_cu, const GeomArgs *gen = nullptr) {
    if (!c->nCU) {
        hipDeviceProp_t pr;
        HIPCHK(c, hipGetDeviceProperties(&pr, c->device));
        c->nCU = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }
    if (!c->kpCtr) {
        void *p_ = nullptr;
        const size_t bytes = (size_t)KP_RING * 9 * KP_CTR_STRIDE * sizeof(unsigned);
        HIPCHK(c, hipMalloc(&p_, bytes));
        c->kpCtr = reinterpret_cast<unsigned *>(p_);
        HIPCHK(c, hipMemsetAsync(c->kpCtr, 0, bytes, c->stream));         // once: every launch leaves its set zeroed (kp_leave)
    }
    if (lds > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute(gen ? reinterpret_cast<const void *>(raftx_kpg_f0) : reinterpret_cast<const void *>(kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PersistArgs P;
    P.T = T;
    P.A = A;
    P.ctr = c->kpCtr + (size_t)(c->kpNext++ % KP_RING) * 9 * KP_CTR_STRIDE;   // (a set is reused KP_RING launches of this ctx later)
    P.xl_base = XL_SLOTS;
    static const int grid_env = getenv("RAFTX_KP_GRID") ? atoi(getenv("RAFTX_KP_GRID")) : 0;      // tuning: workgroups of the grid
    size_t grid = (size_t)(grid_env > 0 ? grid_env : wg_per_cu * c->nCU);
    grid = std::min<size_t>(std::min<size_t>(grid, KP_MAX_GRID), grid_for_pairs(npairs));
    if (gen) {                                            // the generating form: the workgroups build their designs' tables first
        PersistGenArgs PG;
        PG.P = P;
        PG.G = *gen;
        hipLaunchKernelGGL(raftx_kpg_f0, dim3((unsigned)grid), dim3(threads), lds, c->stream, PG);
        return 0;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(threads), lds, c->stream, P);
    return 0;
}

static int solve_enqueue(raftx_ctx *c, int nIter, double tol, double XiStart, const raftx_c128 *F_extra, int want_mask,
                         SlabPlan *plan = nullptr) {
    if (check_ready(c)) return -1;
    if (nIter < 0) FAIL(c, "solve_dynamics: nIter < 0");
    HIPCHK(c, hipSetDevice(c->device));
    const DevTables &T = c->T;
    if (ensure_results(c, want_mask, F_extra != nullptr)) return -1;
    SolveArgs A;
    A.nIter = nIter + 1;
    A.tol = tol;
    A.XiStart = XiStart;
    A.F_extra = F_extra ? c->rFe : (c->bem_ready ? c->bemF : nullptr);
    A.Xi = c->rXi;
    A.niter = c->rNi;
    A.flags = c->rFl;
    A.B_drag = (want_mask & RAFTX_WANT_BDRAG) ? c->rB : nullptr;
    A.F_wave = (want_mask & RAFTX_WANT_FWAVE) ? c->rFw : nullptr;
    A.Z = (want_mask & RAFTX_WANT_Z) ? c->rZ : nullptr;
    c->r_last_mask = want_mask;          // buffers of a wider earlier request stay allocated (ensure_results) but are NOT rewritten
    A.dbg = nullptr;
#ifdef RAFTX_PHASE_TIMING
    if (!c->dbg) {
        void *p_ = nullptr;
        HIPCHK(c, hipMalloc(&p_, (10 + 2 * PT_CLOCK_BUCKETS) * sizeof(unsigned long long)));
        c->dbg = reinterpret_cast<unsigned long long *>(p_);
    }
    HIPCHK(c, hipMemsetAsync(c->dbg, 0, (10 + 2 * PT_CLOCK_BUCKETS) * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(c->dbg + 8, 0xFF, sizeof(unsigned long long), c->stream));      // earliest start: a minimum
    A.dbg = c->dbg;
#endif
    if (F_extra && c->r_nx) H2D(c, c->rFe, F_extra, c->r_nx * sizeof(cplx));
    // the lean specialisation (no optional inputs / outputs) is the sweep path
    int need = (T.MBw ? KF_FDEP : 0) | (A.Z ? KF_OUTZ : 0) | (A.F_wave ? KF_OUTF : 0) | (A.F_extra ? KF_EXTRA : 0) |
                     (T.cm ? KF_MCF : 0) | (T.nHead > 1 ? KF_MULTI : 0);
    const Shape sh = pick_shape(T.nw);
    const bool xlg = xl_global(sh.nb, shape_maxt(sh));       // XiLast in a global scratch slab (raftx_kernels.h XlStore)
    // Workgroups a CU can hold by registers (the shape's waves per SIMD); LDS beyond what that residency needs goes to
    // the run-start cache (raftx_kernels.h Kin): as many 16-byte-per-bin slots as fit without costing a resident pair.
    const bool rc_shape = shape_maxt(sh) == 128 && sh.nb == 2;          // RC of k_solve_dynamics
    // Which specialisation runs.  The 200-bin shape has lean kernels (two waves per SIMD, no Z / F_wave / restart I/O)
    // for the feature sets a sweep meets -- several headings, MacCamy-Fuchs columns, frequency-dependent M / B (turbine
    // aerodynamics, potential-flow coefficients), a resident extra excitation (BEM, second-order) -- and takes the
    // smallest one that covers what this call needs; everything else (the drop-in's optional outputs) is the
    // full-featured kernel at one wave per SIMD.
#define RAFTX_LEAN128(X) X(0) X(KF_FDEP) X(KF_MCF) X(KF_MULTI) X(KF_FDEP | KF_EXTRA) X(KF_FDEP | KF_MCF) X(KF_FDEP | KF_MULTI) \
    X(KF_MCF | KF_MULTI) X(KF_FDEP | KF_EXTRA | KF_MULTI) X(KF_FDEP | KF_MCF | KF_MULTI) X(KF_OUTF) X(KF_OUTF | KF_MULTI)
    if (c->have_xl0 || c->want_xlout) need |= KF_XLIO;
    int lean = -1;
    if (rc_shape) {
#define X(F) if (lean < 0 && (need & ~(F)) == 0) lean = (F);
        RAFTX_LEAN128(X)
#undef X
    } else if (need == 0) {
        lean = 0;
    }
    static const char *force_all = getenv("RAFTX_FORCE_ALL");           // tuning / tests: always the full-featured kernel
    if (force_all && atoi(force_all)) lean = -1;
    const int minb_used = lean >= 0 ? shape_minb(sh) : 1;
    c->last_flags = lean >= 0 ? lean : KF_ALL;
    c->last_minb = minb_used;
    static const int wgcu_env = getenv("RAFTX_WG_PER_CU") ? atoi(getenv("RAFTX_WG_PER_CU")) : 0;     // tuning: pairs per CU the LDS is budgeted for
    const int wg_per_cu = wgcu_env > 0 ? wgcu_env : std::max(1, minb_used * 4 / (sh.threads / 64));
    // the persistent form (raftx_kernels.h k_solve_dynamics_p / raftx_kp_f*): lean 200-bin launches of one LDS class that
    // are not cut into slabs; RAFTX_PERSIST=0 keeps the one-workgroup-per-pair launches (A/B, tuning)
    static const bool persist_env = !(getenv("RAFTX_PERSIST") && !atoi(getenv("RAFTX_PERSIST")));
    const bool persist = persist_env && rc_shape && lean >= 0 && (xlg || RAFTX_XL_LDS);
    auto rc_slots = [&](int S_) {
        if (!(shape_maxt(sh) == 128 && sh.nb == 2)) return 0;                // (= rc_shape below)
        static const char *env = getenv("RAFTX_RC_SLOTS");                 // tuning: cap (0 = no cache)
        const int cap = env ? atoi(env) : 24;
        const size_t base = lds_bytes(S_, xlg ? 0 : T.nw, sh.threads / 64, stage_policy(sh.nb, shape_maxt(sh)),
                                      park_policy(sh.nb, shape_maxt(sh)), 0, T.nw);
        const size_t budget = LDS_LIMIT / (size_t)wg_per_cu - (persist ? KP_STASH * sizeof(double) : 0);
        if (budget <= base) return 0;
        return (int)std::min<size_t>((size_t)cap, (budget - base) / (16 * (size_t)xl_row(T.nw)));
    };
    A.rc_n = rc_slots(c->maxS);
    c->last_rc = A.rc_n;
    const size_t lds = lds_bytes(c->maxS, xlg ? 0 : T.nw, sh.threads / 64, stage_policy(sh.nb, shape_maxt(sh)),
                                 park_policy(sh.nb, shape_maxt(sh)), A.rc_n, rc_shape ? T.nw : 0);
#ifdef RAFTX_XL_PER_PAIR
    const size_t xl_regions = std::max<size_t>(XL_SLOTS, c->r_npair);
#else
    const size_t xl_regions = XL_SLOTS + KP_MAX_GRID;      // the slot pool + one region per workgroup of a persistent grid
#endif
    if (xlg && (!c->rXl || c->rXl_n < xl_regions * 12 * (size_t)T.nw)) {
        // XiLast scratch: a slot per RUNNING workgroup (xl_slot_acquire), not per pair; the slot bits are cleared once --
        // every workgroup returns its slot
        HIPCHK(c, hipDeviceSynchronize());                 // fused kernels of the other streams may hold slots of the old slab
        if (c->rXl) { (void)hipFree(c->rXl); }
        c->rXl = nullptr;
        c->rXl_n = 0;
        void *p_ = nullptr;
        HIPCHK(c, hipMalloc(&p_, xl_regions * 12 * (size_t)T.nw * sizeof(double)));
        c->rXl = reinterpret_cast<double *>(p_);
        c->rXl_n = xl_regions * 12 * (size_t)T.nw;
    }
    if (xlg && !c->rXlSlots) {                             // (on its own: a failed allocation above leaves no half-made pair)
        void *p_ = nullptr;
        HIPCHK(c, hipMalloc(&p_, XL_POOLS * XL_POOL_WORDS * sizeof(unsigned long long)));
        c->rXlSlots = reinterpret_cast<unsigned long long *>(p_);
        HIPCHK(c, hipMemsetAsync(c->rXlSlots, 0, XL_POOLS * XL_POOL_WORDS * sizeof(unsigned long long), c->stream));   // ahead of this ctx's launches, in stream order
    }
    A.Xl = c->rXl;
    A.slots = c->rXlSlots;
    // restart / export of the linearisation point (the re-entry of raft_model.py:1108-1131)
    const size_t nxl = c->r_npair * 6 * (size_t)T.nw;
    A.Xl0 = nullptr;
    A.XlOut = nullptr;
    if (c->have_xl0 || c->want_xlout) {
        if (c->rXlio_n != nxl || !c->rXlOut) {
            if (c->have_xl0) FAIL(c, "solve_dynamics: the linearisation point was set for a different batch shape");
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->rXl0) (void)hipFree(c->rXl0);
            if (c->rXlOut) (void)hipFree(c->rXlOut);
            c->rXl0 = c->rXlOut = nullptr;
            void *p0 = nullptr, *p1 = nullptr;
            HIPCHK(c, hipMalloc(&p0, (nxl ? nxl : 1) * sizeof(cplx)));
            HIPCHK(c, hipMalloc(&p1, (nxl ? nxl : 1) * sizeof(cplx)));
            c->rXl0 = reinterpret_cast<cplx *>(p0);
            c->rXlOut = reinterpret_cast<cplx *>(p1);
            c->rXlio_n = nxl;
        }
        if (c->have_xl0) A.Xl0 = c->rXl0;
        A.XlOut = c->rXlOut;
        need |= KF_XLIO;
        c->have_xl0 = false;                  // one-shot
    }
    // LDS classes: the workgroups of a launch all get the LDS of its largest design, and the number of pairs a CU holds
    // (4 at C3) falls with it -- one 140-strip candidate would cost a whole 10 k-design sweep a quarter of its
    // residency.  Designs are therefore grouped by how many of their pairs fit a CU, one launch per group (largest
    // residency first), through a pair list; a batch of one class (the usual case) is one launch without a list.
    std::vector<std::vector<int>> cls;
    std::vector<int> clsS;
    {
        auto lds_of = [&](int S_) {
            return lds_bytes(S_, xlg ? 0 : T.nw, sh.threads / 64, stage_policy(sh.nb, shape_maxt(sh)), park_policy(sh.nb, shape_maxt(sh)),
                             0, rc_shape ? T.nw : 0);
        };
        auto fit = [&](int S_) { return std::min(wg_per_cu, (int)(LDS_LIMIT / lds_of(S_))); };   // pairs a CU holds
        const int kmax = fit(0);
        bool mixed = false;
        if ((int)c->hS.size() == T.nDesign && T.nDesign > 0) {
            const int k0 = fit(c->hS[0]);
            for (int d = 1; d < T.nDesign && !mixed; d++) mixed = fit(c->hS[(size_t)d]) != k0;
        }
        if (mixed) {
            cls.assign((size_t)kmax + 1, {});
            clsS.assign((size_t)kmax + 1, 0);
            for (int d = 0; d < T.nDesign; d++) {
                const int S_ = c->hS[(size_t)d], kk = fit(S_);
                for (int ic = 0; ic < T.nCase; ic++) cls[(size_t)kk].push_back(d * T.nCase + ic);
                if (S_ > clsS[(size_t)kk]) clsS[(size_t)kk] = S_;
            }
            if (c->pairList_n < c->r_npair) {
                HIPCHK(c, hipStreamSynchronize(c->stream));
                if (c->pairList) (void)hipFree(c->pairList);
                c->pairList = nullptr;
                void *p_ = nullptr;
                HIPCHK(c, hipMalloc(&p_, c->r_npair * sizeof(int)));
                c->pairList = reinterpret_cast<int *>(p_);
                c->pairList_n = c->r_npair;
            }
            size_t at = 0;
            for (int kk = kmax; kk >= 0; kk--)
                if (!cls[(size_t)kk].empty()) {
                    H2D(c, c->pairList + at, cls[(size_t)kk].data(), cls[(size_t)kk].size() * sizeof(int));
                    at += cls[(size_t)kk].size();
                }
        }
    }
    const bool slabbed = plan && cls.empty() && !plan->alts.empty() && plan->bnd.size() >= 2 && plan->bnd.back() == c->r_npair;
    if (slabbed) {
        if (c->identList_n < c->r_npair) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->identList) (void)hipFree(c->identList);
            c->identList = nullptr;
            void *p_ = nullptr;
            HIPCHK(c, hipMalloc(&p_, c->r_npair * sizeof(int)));
            c->identList = reinterpret_cast<int *>(p_);
            c->identList_n = c->r_npair;
            hipLaunchKernelGGL(k_iota, dim3((unsigned)((c->r_npair + 255) / 256)), dim3(256), 0, c->stream, (int)c->r_npair, c->identList);
        }
        if (!c->evFork) HIPCHK(c, hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
        if (!c->evJoin) HIPCHK(c, hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming));
    }
    // A generation deferred to this launch (build_phase2): the generating form of the plain persistent kernel builds every
    // design's tables in the workgroup that solves it -- if this launch is one (every design claimed exactly once);
    // otherwise the generation kernels go first, here.
    BuildJob &J = c->job;
    bool gen_fused = false;
    if (J.gen_deferred) {
        J.gen_deferred = false;
        gen_fused = persist && lean == 0 && T.nCase == 1 && cls.empty() && !slabbed && c->r_npair == (size_t)T.nDesign && c->r_npair > 0 &&
                    J.nDesign == T.nDesign && J.gen_lds + KP_STASH * sizeof(double) <= LDS_LIMIT / (size_t)wg_per_cu;
        if (!gen_fused) {
            J.A.addup_in_design = 0;
            launch_design(c, c->stream);
            hipLaunchKernelGGL(k_geom_addup, dim3((unsigned)(((size_t)J.nDesign * 36 + 255) / 256)), dim3(256), 0, c->stream, J.A);
        }
    }
    c->last_gen_fused = gen_fused;
    int rc_after = 0;
#define LAUNCH_SOLVE(NB_, MT_, MB_, FL)                                                                              \
    do {                                                                                                             \
        if (prep_lds(c, k_solve_dynamics<NB_, FL, MT_, MB_>, lds)) return -1;                                        \
        HIPCHK(c, hipEventRecord(c->ev0, c->stream));                                                                \
        A.pairs = nullptr;                                                                                           \
        A.npairs = 0;                                                                                                \
        if (c->r_npair && cls.empty() && !slabbed) {                                                                 \
            kp_fn kp_ = (persist && NB_ == 2 && MT_ == 128 && MB_ == RAFTX_KP_MINB) ? kp_kernel(FL) : nullptr;       \
            if (kp_) {                                                                                               \
                if (launch_persistent(c, kp_, T, A, c->r_npair, sh.threads,                                          \
                                      std::max(lds, gen_fused ? J.gen_lds : (size_t)0) + KP_STASH * sizeof(double), wg_per_cu, \
                                      (gen_fused && FL == 0) ? &J.A : nullptr))                                      \
                    return -1;                                                                                       \
            } else {                                                                                                 \
                hipLaunchKernelGGL((k_solve_dynamics<NB_, FL, MT_, MB_>), dim3(grid_for_pairs(c->r_npair)),          \
                                   dim3(sh.threads), lds, c->stream, T, A);                                          \
            }                                                                                                        \
        }                                                                                                            \
        if (slabbed) {                                            /* the tables and the iota list are on the ctx stream */ \
            HIPCHK(c, hipEventRecord(c->evFork, c->stream));                                                         \
            for (size_t k_ = 0; k_ < plan->alts.size() && k_ + 1 < plan->bnd.size(); k_++)                           \
                HIPCHK(c, hipStreamWaitEvent(plan->alts[(plan->next + k_) % plan->alts.size()], c->evFork, 0));      \
            for (size_t k_ = 0; k_ + 1 < plan->bnd.size() && !rc_after; k_++) {                                      \
                hipStream_t s_ = plan->alts[plan->next++ % plan->alts.size()];                                       \
                A.pairs = c->identList + plan->bnd[k_];                                                              \
                A.npairs = (int)(plan->bnd[k_ + 1] - plan->bnd[k_]);                                                 \
                hipLaunchKernelGGL((k_solve_dynamics<NB_, FL, MT_, MB_>), dim3(grid_for_pairs((size_t)A.npairs)),    \
                                   dim3(sh.threads), lds, s_, T, A);                                                 \
                if (plan->after) rc_after = plan->after(plan->bnd[k_], plan->bnd[k_ + 1], s_);                       \
            }                                                                                                        \
        }                                                                                                            \
        size_t at_ = 0;                                                                                              \
        for (int kk = (int)cls.size() - 1; kk >= 0; kk--) {                                                          \
            const size_t n_ = cls[(size_t)kk].size();                                                                \
            if (!n_) continue;                                                                                       \
            A.pairs = c->pairList + at_;                                                                             \
            A.npairs = (int)n_;                                                                                      \
            at_ += n_;                                                                                               \
            A.rc_n = rc_slots(clsS[(size_t)kk]);                                                                     \
            const size_t l_ = lds_bytes(clsS[(size_t)kk], xlg ? 0 : T.nw, sh.threads / 64, stage_policy(sh.nb, shape_maxt(sh)), \
                                        park_policy(sh.nb, shape_maxt(sh)), A.rc_n, rc_shape ? T.nw : 0);            \
            hipLaunchKernelGGL((k_solve_dynamics<NB_, FL, MT_, MB_>), dim3(grid_for_pairs(n_)), dim3(sh.threads),    \
                               l_, c->stream, T, A);                                                                 \
        }                                                                                                            \
        if (plan && plan->after && !slabbed) rc_after = plan->after(0, c->r_npair, c->stream);                       \
    } while (0)
#define TRY_LEAN_(F) if (lean == (F)) LAUNCH_SOLVE(2, 128, RAFTX_MINB128, (F));
#define DISPATCH_ONE_(NB_, MT_, MB_)                                                                                 \
    if (!hit_ && sh.nb == NB_ && sh.threads == MT_) {                                                                \
        hit_ = true;                                                                                                 \
        if (lean < 0) LAUNCH_SOLVE(NB_, MT_, 1, KF_ALL);   /* full-featured variant: trade occupancy for registers */ \
        else if constexpr (NB_ == 2 && MT_ == 128) { RAFTX_LEAN128(TRY_LEAN_) }                                      \
        else LAUNCH_SOLVE(NB_, MT_, MB_, 0);                                                                         \
    }
    DISPATCH_SHAPE(sh, _);
#undef DISPATCH_ONE_
#undef LAUNCH_SOLVE
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipGetLastError());
    if (rc_after) FAIL(c, "solve_dynamics: what follows a slab of the launch could not be enqueued");
    return 0;
}
static int finish_enqueued(raftx_ctx *c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_ms = ms;
    return 0;
}
extern "C" int raftx_solve_dynamics_device(raftx_ctx *c, int nIter, double tol, double XiStart,
                                           const raftx_c128 *F_extra, int want_mask) {
    RangeScope range_("raftx_solve_dynamics_device: fused fixed point");
    const int rc = solve_enqueue(c, nIter, tol, XiStart, F_extra, want_mask);
    if (rc) return rc;
    return finish_enqueued(c);
}

extern "C" int raftx_set_linearisation_point(raftx_ctx *c, const raftx_c128 *XiLast0, int keep_last) {
    if (!c) return -1;
    if (check_ready(c)) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    c->want_xlout = keep_last != 0;
    c->have_xl0 = false;
    if (XiLast0) {
        const DevTables &T = c->T;
        const size_t nxl = (size_t)T.nDesign * T.nCase * 6 * T.nw;
        if (c->rXlio_n != nxl || !c->rXl0) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->rXl0) (void)hipFree(c->rXl0);
            if (c->rXlOut) (void)hipFree(c->rXlOut);
            c->rXl0 = c->rXlOut = nullptr;
            void *p0 = nullptr, *p1 = nullptr;
            HIPCHK(c, hipMalloc(&p0, (nxl ? nxl : 1) * sizeof(cplx)));
            HIPCHK(c, hipMalloc(&p1, (nxl ? nxl : 1) * sizeof(cplx)));
            c->rXl0 = reinterpret_cast<cplx *>(p0);
            c->rXlOut = reinterpret_cast<cplx *>(p1);
            c->rXlio_n = nxl;
        }
        if (nxl) H2D(c, c->rXl0, XiLast0, nxl * sizeof(cplx));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->have_xl0 = true;
    }
    return 0;
}

extern "C" int raftx_fetch_linearisation_point(raftx_ctx *c, raftx_c128 *XiLast) {
    if (!c) return -1;
    if (!c->rXlOut || !XiLast) FAIL(c, "fetch_linearisation_point: nothing kept (call raftx_set_linearisation_point(ctx, ., 1) first)");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->rXlio_n) D2H(c, XiLast, c->rXlOut, c->rXlio_n * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int raftx_fetch_results(raftx_ctx *c, raftx_c128 *Xi, int32_t *niter, int32_t *flags, double *B_drag,
                                   raftx_c128 *F_wave, raftx_c128 *Z) {
    RangeScope range_("raftx_fetch_results: D2H");
    if (!c) return -1;
    if (!c->rXi) FAIL(c, "fetch_results: no resident results");
    HIPCHK(c, hipSetDevice(c->device));
    // (kept = written by the LAST solve: a buffer left over from a wider earlier request holds that request's results)
    if (B_drag && !(c->rB && (c->r_last_mask & RAFTX_WANT_BDRAG))) FAIL(c, "fetch_results: B_drag was not kept");
    if (F_wave && !(c->rFw && (c->r_last_mask & RAFTX_WANT_FWAVE))) FAIL(c, "fetch_results: F_wave was not kept");
    if (Z && !(c->rZ && (c->r_last_mask & RAFTX_WANT_Z))) FAIL(c, "fetch_results: Z was not kept");
    if (Xi && c->r_nx) D2H(c, Xi, c->rXi, c->r_nx * sizeof(cplx));
    if (niter && c->r_npair) D2H(c, niter, c->rNi, c->r_npair * sizeof(int));
    if (flags && c->r_npair) D2H(c, flags, c->rFl, c->r_npair * sizeof(int));
    if (B_drag && c->r_npair) D2H(c, B_drag, c->rB, c->r_npair * 36 * sizeof(double));
    if (F_wave && c->r_nx) D2H(c, F_wave, c->rFw, c->r_nx * sizeof(cplx));
    if (Z && c->r_nz) D2H(c, Z, c->rZ, c->r_nz * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int raftx_solve_dynamics(raftx_ctx *c, int nIter, double tol, double XiStart, const raftx_c128 *F_extra,
                                    raftx_c128 *Xi, int32_t *niter, int32_t *flags, double *B_drag,
                                    raftx_c128 *F_wave, raftx_c128 *Z) {
    int mask = (B_drag ? RAFTX_WANT_BDRAG : 0) | (F_wave ? RAFTX_WANT_FWAVE : 0) | (Z ? RAFTX_WANT_Z : 0);
    int rc = raftx_solve_dynamics_device(c, nIter, tol, XiStart, F_extra, mask);
    if (rc) return rc;
    return raftx_fetch_results(c, Xi, niter, flags, B_drag, F_wave, Z);
}

extern "C" int raftx_bem_excitation(raftx_ctx *c, int nHeadBEM, const double *headings_deg, const raftx_c128 *X_BEM,
                                    const double *heading_adjust, const double *xy_ref, const raftx_c128 *F_add,
                                    raftx_c128 *F_out) {
    if (check_ready(c)) return -1;
    if (nHeadBEM < 1 || !headings_deg || !X_BEM) FAIL(c, "bem_excitation: bad arguments");
    for (int i = 0; i < nHeadBEM; i++)
        if (!(headings_deg[i] >= 0.0 && headings_deg[i] < 360.0) || (i && !(headings_deg[i] > headings_deg[i - 1])))
            FAIL(c, "bem_excitation: headings must be ascending in [0, 360)");
    HIPCHK(c, hipSetDevice(c->device));
    const DevTables &T = c->T;
    const size_t npair = (size_t)T.nDesign * T.nCase, nx = npair * T.nHead * 6 * T.nw;
    const size_t nX = (size_t)T.nDesign * nHeadBEM * 6 * T.nw;
    c->bem_ready = false;
    if (c->bemF_n < nx || !c->bemF) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->bemF) (void)hipFree(c->bemF);
        c->bemF = nullptr;
        void *p_ = nullptr;
        HIPCHK(c, hipMalloc(&p_, (nx ? nx : 1) * sizeof(cplx)));
        c->bemF = reinterpret_cast<cplx *>(p_);
        c->bemF_n = nx;
    }
    Scratch sc(c);
    double *dH = sc.alloc<double>(nHeadBEM), *dA = heading_adjust ? sc.alloc<double>(T.nDesign) : nullptr,
           *dXY = xy_ref ? sc.alloc<double>((size_t)T.nDesign * 2) : nullptr;
    cplx *dX = sc.alloc<cplx>(nX), *dAdd = F_add ? sc.alloc<cplx>(nx) : nullptr;
    if (nx && (!dH || !dX || (heading_adjust && !dA) || (xy_ref && !dXY) || (F_add && !dAdd)))
        FAIL(c, "bem_excitation: device allocation failed");
    if (nx) {
        H2D(c, dH, headings_deg, nHeadBEM * sizeof(double));
        H2D(c, dX, X_BEM, nX * sizeof(cplx));
        if (dA) H2D(c, dA, heading_adjust, T.nDesign * sizeof(double));
        if (dXY) H2D(c, dXY, xy_ref, (size_t)T.nDesign * 2 * sizeof(double));
        if (dAdd) H2D(c, dAdd, F_add, nx * sizeof(cplx));
    }
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (nx)
        hipLaunchKernelGGL(k_bem_excitation, dim3((unsigned)(npair * T.nHead)), dim3(T.nw > 128 ? 256 : (T.nw > 64 ? 128 : 64)), 0,
                           c->stream, T, nHeadBEM, dH, dX, dA, dXY, dAdd, c->bemF);
    if (finish_timed(c)) return -2;
    if (nx && F_out) D2H(c, F_out, c->bemF, nx * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->bem_ready = true;
    return 0;
}

extern "C" int raftx_motion_stats(raftx_ctx *c, double dw, double *sd, double *psd) {
    if (!c) return -1;
    if (!c->rXi) FAIL(c, "motion_stats: no resident results");
    if (!sd) FAIL(c, "motion_stats: std is NULL");
    if (!(dw > 0.0)) FAIL(c, "motion_stats: dw must be positive");
    HIPCHK(c, hipSetDevice(c->device));
    const DevTables &T = c->T;
    const size_t npair = c->r_npair;
    Scratch sc(c);
    double *dS = sc.alloc<double>(npair * 6);
    double *dP = psd ? sc.alloc<double>(npair * 6 * T.nw) : nullptr;
    if (npair && (!dS || (psd && !dP))) FAIL(c, "motion_stats: device allocation failed");
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (npair)
        hipLaunchKernelGGL(k_motion_stats, dim3((unsigned)npair), dim3(T.nw > 128 ? 256 : (T.nw > 64 ? 128 : 64)), 0,
                           c->stream, (int)npair, T.nHead, T.nw, 1.0 / dw, c->rXi, dS, dP);
    if (finish_timed(c)) return -2;
    if (npair) D2H(c, sd, dS, npair * 6 * sizeof(double));
    if (npair && psd) D2H(c, psd, dP, npair * 6 * T.nw * sizeof(double));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int raftx_channel_stats(raftx_ctx *c, int nChan, const double *L, const int32_t *pw, double dw, double *sd,
                                   double *psd) {
    if (!c) return -1;
    if (!c->rXi) FAIL(c, "channel_stats: no resident results");
    if (nChan < 0 || (nChan && (!L || !pw)) || !sd) FAIL(c, "channel_stats: bad arguments");
    if (!(dw > 0.0)) FAIL(c, "channel_stats: dw must be positive");
    for (int i = 0; i < nChan; i++)
        if (pw[i] < 0 || pw[i] > 4) FAIL(c, "channel_stats: pow[%d]=%d outside 0..4", i, pw[i]);
    HIPCHK(c, hipSetDevice(c->device));
    const DevTables &T = c->T;
    const size_t npair = c->r_npair;
    Scratch sc(c);
    double *dL = sc.alloc<double>((size_t)T.nDesign * nChan * 6), *dS = sc.alloc<double>(npair * nChan);
    int *dP = sc.alloc<int>(nChan);
    double *dPsd = psd ? sc.alloc<double>(npair * nChan * T.nw) : nullptr;
    if (npair && nChan && (!dL || !dS || !dP || (psd && !dPsd))) FAIL(c, "channel_stats: device allocation failed");
    if (npair && nChan) {
        H2D(c, dL, L, (size_t)T.nDesign * nChan * 6 * sizeof(double));
        H2D(c, dP, pw, nChan * sizeof(int));
    }
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (npair && nChan)
        hipLaunchKernelGGL(k_channel_stats, dim3((unsigned)npair), dim3(T.nw > 128 ? 256 : (T.nw > 64 ? 128 : 64)), 0,
                           c->stream, T.nCase, T.nHead, T.nw, nChan, 1.0 / dw, T.w, c->rXi, dL, dP, dS, dPsd);
    if (finish_timed(c)) return -2;
    if (npair && nChan) D2H(c, sd, dS, npair * nChan * sizeof(double));
    if (npair && nChan && psd) D2H(c, psd, dPsd, npair * nChan * T.nw * sizeof(double));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int raftx_channel_stats_poly(raftx_ctx *c, int nChan, const double *L, const raftx_c128 *Gw, double dw, double *sd,
                                        double *psd) {
    if (!c) return -1;
    if (!c->rXi) FAIL(c, "channel_stats_poly: no resident results");
    if (nChan < 0 || (nChan && !L) || !sd) FAIL(c, "channel_stats_poly: bad arguments");
    if (!(dw > 0.0)) FAIL(c, "channel_stats_poly: dw must be positive");
    HIPCHK(c, hipSetDevice(c->device));
    const DevTables &T = c->T;
    const size_t npair = c->r_npair;
    Scratch sc(c);
    const size_t nL = (size_t)T.nDesign * nChan * 18, nG = (size_t)T.nDesign * nChan * 6 * T.nw;
    double *dL = sc.alloc<double>(nL), *dS = sc.alloc<double>(npair * nChan);
    cplx *dG = Gw ? sc.alloc<cplx>(nG) : nullptr;
    double *dPsd = psd ? sc.alloc<double>(npair * nChan * T.nw) : nullptr;
    if (npair && nChan && (!dL || !dS || (Gw && !dG) || (psd && !dPsd))) FAIL(c, "channel_stats_poly: device allocation failed");
    if (npair && nChan) {
        H2D(c, dL, L, nL * sizeof(double));
        if (dG) H2D(c, dG, Gw, nG * sizeof(cplx));
    }
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (npair && nChan)
        hipLaunchKernelGGL(k_channel_stats_poly, dim3((unsigned)npair), dim3(T.nw > 128 ? 256 : (T.nw > 64 ? 128 : 64)), 0,
                           c->stream, T.nCase, T.nHead, T.nw, nChan, 1.0 / dw, T.w, c->rXi, dL, dG, dS, dPsd);
    if (finish_timed(c)) return -2;
    if (npair && nChan) D2H(c, sd, dS, npair * nChan * sizeof(double));
    if (npair && nChan && psd) D2H(c, psd, dPsd, npair * nChan * T.nw * sizeof(double));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int raftx_response_stats(raftx_ctx *c, int nChan, int nDof, int nResp, int nw, const double *w, const double *L,
                                    const raftx_c128 *Gw, const raftx_c128 *Xi, double dw, double *sd, double *psd) {
    RangeScope range_("raftx_response_stats: linear channels of a caller-held response");
    if (!c) return -1;
    if (nChan < 0 || nDof < 1 || nResp < 1 || nw < 1 || !w || (nChan && !L) || !Xi || !sd) FAIL(c, "response_stats: bad arguments");
    if (!nChan) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    Scratch sc(c);
    const size_t nl = (size_t)nChan * 3 * nDof, ng = (size_t)nChan * nDof * nw, nx = (size_t)nResp * nDof * nw;
    double *dw_ = sc.alloc<double>(nw), *dL = sc.alloc<double>(nl), *dS = sc.alloc<double>(nChan);
    double *dP = psd ? sc.alloc<double>((size_t)nChan * nw) : nullptr;
    cplx *dG = Gw ? sc.alloc<cplx>(ng) : nullptr, *dX = sc.alloc<cplx>(nx);
    if (!dw_ || !dL || !dS || (psd && !dP) || (Gw && !dG) || !dX) FAIL(c, "response_stats: device allocation failed");
    H2D(c, dw_, w, nw * sizeof(double));
    H2D(c, dL, L, nl * sizeof(double));
    if (dG) H2D(c, dG, Gw, ng * sizeof(cplx));
    H2D(c, dX, Xi, nx * sizeof(cplx));
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    hipLaunchKernelGGL(k_response_stats, dim3((unsigned)nChan), dim3(nw > 128 ? 256 : (nw > 64 ? 128 : 64)), 0, c->stream, nDof, nResp, nw,
                       1.0 / dw, dw_, dX, dL, dG, dS, dP);
    if (finish_timed(c)) return -2;
    D2H(c, sd, dS, nChan * sizeof(double));
    if (psd) D2H(c, psd, dP, (size_t)nChan * nw * sizeof(double));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int raftx_solve_system(raftx_ctx *c, int nSys, int nUnit, int nRhs, int nw, const double *w,
                                  const raftx_c128 *Zblk, const double *Mc, const double *Bc, const double *Cc,
                                  const raftx_c128 *F, raftx_c128 *Xi) {
    if (!c) return -1;
    if (nSys < 0 || nUnit < 1 || nRhs < 1 || nw < 1 || !w || !Zblk || !F || !Xi) FAIL(c, "solve_system: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const int n = 6 * nUnit;
    size_t lds = 0;
    const int nbin = solve_system_shape(n, nRhs, &lds);
    if (lds > 150 * 1024) FAIL(c, "solve_system: %d DOFs x %d rhs does not fit the LDS-resident solver", n, nRhs);
    Scratch sc(c);
    size_t nz = (size_t)nSys * nUnit * 36 * nw, nf = (size_t)nSys * nRhs * n * nw, nc = (size_t)nSys * n * n;
    double *dw = sc.alloc<double>(nw);
    cplx *dZ = sc.alloc<cplx>(nz), *dF = sc.alloc<cplx>(nf), *dX = sc.alloc<cplx>(nf);
    double *dM = Mc ? sc.alloc<double>(nc) : nullptr, *dB = Bc ? sc.alloc<double>(nc) : nullptr,
           *dC = Cc ? sc.alloc<double>(nc) : nullptr;
    if (nSys && (!dw || !dZ || !dF || !dX || (Mc && !dM) || (Bc && !dB) || (Cc && !dC)))
        FAIL(c, "solve_system: device allocation failed");
    if (nSys) {
        H2D(c, dw, w, nw * sizeof(double));
        H2D(c, dZ, Zblk, nz * sizeof(cplx));
        H2D(c, dF, F, nf * sizeof(cplx));
        if (dM) H2D(c, dM, Mc, nc * sizeof(double));
        if (dB) H2D(c, dB, Bc, nc * sizeof(double));
        if (dC) H2D(c, dC, Cc, nc * sizeof(double));
    }
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (nSys && !launch_solve_system_rows<false>(c->stream, nSys, nUnit, nRhs, nw, 1, dw, dZ, dM, dB, dC, dF, dX)) {
        if (lds > 64 * 1024)
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve_system<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_solve_system<false>, dim3((unsigned)((size_t)nSys * ((nw + nbin - 1) / nbin))), dim3(64 * nbin), lds,
                           c->stream, nSys, nUnit, nRhs, nw, 1, dw, dZ, dM, dB, dC, dF, dX);
    }
    if (finish_timed(c)) return -2;
    if (nSys) D2H(c, Xi, dX, nf * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// launch of the dense solves of nSys systems x nw bins on device arrays: the register-resident kernel where the augmented
// matrix fits its grid (the flexible deck: 150 + 1), the L2-workspace kernel otherwise.  Matrix set of system s: s / mdiv.
static bool dense_reg_shape(int n, int nRhs) {
    static const bool dense_l2 = getenv("RAFTX_DENSE_L2") && atoi(getenv("RAFTX_DENSE_L2"));     // tuning / tests: the L2-workspace kernel
    return !dense_l2 && n + nRhs <= 160;
}
static int dense_launch(raftx_ctx *c, int nSys, int mdiv, int n, int nRhs, int nw, const double *dw, const double *dM, const double *dB,
                        const double *dC, int freq_mask, const double *dBadd, const cplx *dF, cplx *dX, cplx *dZ, cplx *dA,
                        const int *dActive = nullptr) {
    const dim3 grid((unsigned)nw, (unsigned)nSys);
    if (dense_reg_shape(n, nRhs)) {
#define DENSE_REG_(RB_, CB_)                                                                                            \
        hipLaunchKernelGGL((k_solve_dense_reg2<RB_, CB_, 16>), grid, dim3(512), 0, c->stream, n, nRhs, nw, dw, dM, dB, dC, freq_mask, \
                           mdiv, dBadd, dActive, dF, dX, dZ)
        // entries per thread: the smallest 32 RB x 16 CB grid that holds [Z | F] (measured at 60 DOFs: 0.076 ms with 3 x 6
        // against 0.136 with 5 x 10 and 0.23 for the L2-workspace kernel)
        if (n + nRhs <= 32) DENSE_REG_(1, 2);
        else if (n + nRhs <= 64) DENSE_REG_(2, 4);
        else if (n + nRhs <= 96) DENSE_REG_(3, 6);
        else if (n + nRhs <= 128) DENSE_REG_(4, 8);
        else DENSE_REG_(5, 10);
#undef DENSE_REG_
    } else {
        hipLaunchKernelGGL(k_solve_dense, grid, dim3(256), 0, c->stream, n, nRhs, nw, dw, dM, dB, dC, freq_mask, mdiv, dBadd, dActive, dF, dA, dX, dZ);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}
static int dense_check(raftx_ctx *c, const char *who, int nSys, int n, int nRhs, int nw) {
    if (nSys < 0 || n < 1 || nRhs < 1 || nw < 1) FAIL(c, "%s: bad arguments", who);
    if (n + nRhs > DENSE_MAX_LD) FAIL(c, "%s: %d DOFs + %d right-hand sides exceed %d", who, n, nRhs, DENSE_MAX_LD);
    if (nSys > 65535) FAIL(c, "%s: at most 65 535 systems per call", who);
    return 0;
}

static int solve_dense_impl(raftx_ctx *c, int nSys, int n, int nRhs, int nw, const double *w, const double *M, const double *B,
                            const double *C, int freq_mask, const raftx_c128 *F, raftx_c128 *Xi, raftx_c128 *Z) {
    if (!c) return -1;
    if (!w || !M || !B || !C || !F || !Xi) FAIL(c, "solve_dense: bad arguments");
    if (dense_check(c, "solve_dense", nSys, n, nRhs, nw)) return -1;
    if (nSys == 0) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    Scratch sc(c);
    const bool reg = dense_reg_shape(n, nRhs);
    const size_t nn = (size_t)n * n, nf = (size_t)nSys * nRhs * n * nw;
    const size_t nM = (size_t)nSys * nn * ((freq_mask & 1) ? nw : 1), nB = (size_t)nSys * nn * ((freq_mask & 2) ? nw : 1);
    double *dw = sc.alloc<double>(nw), *dM = sc.alloc<double>(nM), *dB = sc.alloc<double>(nB), *dC = sc.alloc<double>((size_t)nSys * nn);
    cplx *dF = sc.alloc<cplx>(nf), *dX = sc.alloc<cplx>(nf);
    cplx *dA = reg ? nullptr : sc.alloc<cplx>((size_t)nSys * nw * n * (n + nRhs));
    cplx *dZ = Z ? sc.alloc<cplx>((size_t)nSys * nn * nw) : nullptr;
    if (!dw || !dM || !dB || !dC || !dF || !dX || (!reg && !dA) || (Z && !dZ)) FAIL(c, "solve_dense: device allocation failed");
    H2D(c, dw, w, nw * sizeof(double));
    H2D(c, dM, M, nM * sizeof(double));
    H2D(c, dB, B, nB * sizeof(double));
    H2D(c, dC, C, (size_t)nSys * nn * sizeof(double));
    H2D(c, dF, F, nf * sizeof(cplx));
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (dense_launch(c, nSys, 1, n, nRhs, nw, dw, dM, dB, dC, freq_mask, nullptr, dF, dX, dZ, dA)) return -1;
    if (finish_timed(c)) return -2;
    D2H(c, Xi, dX, nf * sizeof(cplx));
    if (Z) D2H(c, Z, dZ, (size_t)nSys * nn * nw * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int raftx_solve_dense(raftx_ctx *c, int n, int nRhs, int nw, const double *w, const double *M, const double *B,
                                 const double *C, int freq_mask, const raftx_c128 *F, raftx_c128 *Xi, raftx_c128 *Z) {
    return solve_dense_impl(c, 1, n, nRhs, nw, w, M, B, C, freq_mask, F, Xi, Z);
}
extern "C" int raftx_solve_dense_batch(raftx_ctx *c, int nSys, int n, int nRhs, int nw, const double *w, const double *M,
                                       const double *B, const double *C, int freq_mask, const raftx_c128 *F, raftx_c128 *Xi,
                                       raftx_c128 *Z) {
    return solve_dense_impl(c, nSys, n, nRhs, nw, w, M, B, C, freq_mask, F, Xi, Z);
}

// The matrices of a fixed point stay on the device: M, B, C of nSet units are uploaded once (raftx_dense_resident); every
// iteration then sends what CHANGES -- the drag linearisation Badd [n,n] of each (unit, sea state) and the right-hand sides --
// and gets the responses back (raftx_solve_dense_resident).  With frequency-dependent rotor matrices the flexible deck's M and
// B are 7.2 MB each: re-uploaded by every raftx_solve_dense call they were most of the call.
extern "C" int raftx_dense_resident(raftx_ctx *c, int nSet, int n, int nw, const double *w, const double *M, const double *B,
                                    const double *C, int freq_mask) {
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    DenseResident &R = c->dense;
    free_list(c, R.allocs);
    R = DenseResident();
    if (nSet == 0) return 0;                                              // (release)
    if (!w || !M || !B || !C) FAIL(c, "dense_resident: bad arguments");
    if (dense_check(c, "dense_resident", nSet, n, 1, nw)) return -1;
    const size_t nn = (size_t)n * n, nM = (size_t)nSet * nn * ((freq_mask & 1) ? nw : 1), nB = (size_t)nSet * nn * ((freq_mask & 2) ? nw : 1);
    if (dev_alloc(c, R.allocs, (size_t)nw, &R.w) || dev_alloc(c, R.allocs, nM, &R.M) || dev_alloc(c, R.allocs, nB, &R.B) ||
        dev_alloc(c, R.allocs, (size_t)nSet * nn, &R.C))
        return -2;
    H2D(c, R.w, w, nw * sizeof(double));
    H2D(c, R.M, M, nM * sizeof(double));
    H2D(c, R.B, B, nB * sizeof(double));
    H2D(c, R.C, C, (size_t)nSet * nn * sizeof(double));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    R.nSet = nSet; R.n = n; R.nw = nw; R.freq_mask = freq_mask;
    return 0;
}
extern "C" int raftx_solve_dense_resident(raftx_ctx *c, int nPer, const double *Badd, int nRhs, const raftx_c128 *F, raftx_c128 *Xi,
                                          raftx_c128 *Z) {
    if (!c) return -1;
    const DenseResident &R = c->dense;
    if (R.nSet < 1) FAIL(c, "solve_dense_resident: no resident matrices (raftx_dense_resident first)");
    if (nPer < 1 || !F || !Xi) FAIL(c, "solve_dense_resident: bad arguments");
    const int nSys = R.nSet * nPer, n = R.n, nw = R.nw;
    if (dense_check(c, "solve_dense_resident", nSys, n, nRhs, nw)) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    Scratch sc(c);
    const bool reg = dense_reg_shape(n, nRhs);
    const size_t nn = (size_t)n * n, nf = (size_t)nSys * nRhs * n * nw;
    double *dBadd = Badd ? sc.alloc<double>((size_t)nSys * nn) : nullptr;
    cplx *dF = sc.alloc<cplx>(nf), *dX = sc.alloc<cplx>(nf);
    cplx *dA = reg ? nullptr : sc.alloc<cplx>((size_t)nSys * nw * n * (n + nRhs));
    cplx *dZ = Z ? sc.alloc<cplx>((size_t)nSys * nn * nw) : nullptr;
    if ((Badd && !dBadd) || !dF || !dX || (!reg && !dA) || (Z && !dZ)) FAIL(c, "solve_dense_resident: device allocation failed");
    if (Badd) H2D(c, dBadd, Badd, (size_t)nSys * nn * sizeof(double));
    H2D(c, dF, F, nf * sizeof(cplx));
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (dense_launch(c, nSys, nPer, n, nRhs, nw, R.w, R.M, R.B, R.C, R.freq_mask, dBadd, dF, dX, dZ, dA)) return -1;
    if (finish_timed(c)) return -2;
    D2H(c, Xi, dX, nf * sizeof(cplx));
    if (Z) D2H(c, Z, dZ, (size_t)nSys * nn * nw * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// include/raftx.h raftx_flex_start: the next raftx_flex_solve starts from this iterate (one-shot)
extern "C" int raftx_flex_start(raftx_ctx *c, int nUnit, int n, const raftx_c128 *XiLast0) {
    if (!c) return -1;
    c->flexXl0_n = 0;
    if (!XiLast0) return 0;
    if (check_ready(c)) return -1;
    const DevTables &T = c->T;
    if (nUnit < 1 || n < 6) FAIL(c, "flex_start: bad arguments (nUnit=%d, n=%d)", nUnit, n);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t need = (size_t)nUnit * T.nCase * n * T.nw;
    if (need > c->flexXl0_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->flexXl0) (void)hipFree(c->flexXl0);
        c->flexXl0 = nullptr;
        c->flexXl0_cap = 0;
        void *p_ = nullptr;
        HIPCHK(c, hipMalloc(&p_, need * sizeof(cplx)));
        c->flexXl0 = reinterpret_cast<cplx *>(p_);
        c->flexXl0_cap = need;
    }
    H2D(c, c->flexXl0, XiLast0, need * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));           // the caller's array is free again
    c->flexXl0_n = need;
    return 0;
}

// The fixed point of flexible units on the device (raftx_flex.h; include/raftx.h raftx_flex_solve).
extern "C" int raftx_flex_solve(raftx_ctx *c, int nUnit, const int64_t *nodeOff, int n, const double *Tn, const double *M,
                                const double *B, const double *C, int freq_mask, const raftx_c128 *F_lin, int nIter, double tol,
                                double XiStart, raftx_c128 *Xi, int32_t *niter, int32_t *flags, double *B_drag, raftx_c128 *F_drag,
                                raftx_c128 *Z) {
    RangeScope range_("raftx_flex_solve: fixed point of units with flexible members");
    if (check_ready(c)) return -1;
    if (nUnit < 1 || !nodeOff || !Tn || !M || !B || !C || !F_lin || !Xi || !niter || !flags || nIter < 0)
        FAIL(c, "flex_solve: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const DevTables &T = c->T;
    const int nCase = T.nCase, nHead = T.nHead, nw = T.nw, nNode = T.nDesign;
    if (nodeOff[0] != 0 || nodeOff[nUnit] != nNode) FAIL(c, "flex_solve: nodeOff must run from 0 to the %d resident node tables", nNode);
    for (int u = 0; u < nUnit; u++)
        if (nodeOff[u + 1] < nodeOff[u]) FAIL(c, "flex_solve: node offsets not monotone");
    const int nSys = nUnit * nCase;
    if (dense_check(c, "flex_solve", nSys, n, nHead, nw)) return -1;
    if (n < 6) FAIL(c, "flex_solve: n = %d reduced DOFs", n);
    Scratch sc(c);
    const size_t nn = (size_t)n * n, nxs = (size_t)n * nw, npn = (size_t)nNode * nCase;
    const size_t nM = (size_t)nUnit * nn * ((freq_mask & 1) ? nw : 1), nB = (size_t)nUnit * nn * ((freq_mask & 2) ? nw : 1);
    const bool reg1 = dense_reg_shape(n, 1), regH = dense_reg_shape(n, nHead);
    std::vector<int> hNodeUnit((size_t)nNode);
    for (int u = 0; u < nUnit; u++)
        for (int64_t i = nodeOff[u]; i < nodeOff[u + 1]; i++) hNodeUnit[(size_t)i] = u;
    double *dM = sc.alloc<double>(nM), *dB = sc.alloc<double>(nB), *dC = sc.alloc<double>((size_t)nUnit * nn);
    double *dTn = sc.alloc<double>((size_t)nNode * 6 * n), *dW = sc.alloc<double>((size_t)nCase * nNode * 6 * n);
    double *dBn = sc.alloc<double>(npn * 36), *dBd = sc.alloc<double>((size_t)nSys * nn);
    int64_t *dOff = sc.alloc<int64_t>((size_t)nUnit + 1);
    int *dNodeUnit = sc.alloc<int>((size_t)nNode), *dAct = sc.alloc<int>((size_t)nSys), *dNi = sc.alloc<int>((size_t)nSys),
        *dFl = sc.alloc<int>((size_t)nSys), *dCount = sc.alloc<int>(1);
    cplx *dXiN = sc.alloc<cplx>(npn * 6 * nw), *dFn = sc.alloc<cplx>(npn * nHead * 6 * nw);
    cplx *dFlin = sc.alloc<cplx>((size_t)nSys * nHead * nxs), *dFw = sc.alloc<cplx>((size_t)nSys * nHead * nxs);
    cplx *dFd = sc.alloc<cplx>((size_t)nSys * nHead * nxs), *dRhs = sc.alloc<cplx>((size_t)nSys * nxs);
    cplx *dXnew = sc.alloc<cplx>((size_t)nSys * nxs), *dXi = sc.alloc<cplx>((size_t)nSys * nxs), *dXl = sc.alloc<cplx>((size_t)nSys * nxs);
    cplx *dXh = sc.alloc<cplx>((size_t)nSys * nHead * nxs);
    cplx *dA = (reg1 && regH) ? nullptr : sc.alloc<cplx>((size_t)nSys * nw * n * (n + nHead));
    cplx *dZ = Z ? sc.alloc<cplx>((size_t)nSys * nn * nw) : nullptr;
    if (!dM || !dB || !dC || !dTn || !dW || !dBn || !dBd || !dOff || !dNodeUnit || !dAct || !dNi || !dFl || !dCount || !dXiN || !dFn ||
        !dFlin || !dFw || !dFd || !dRhs || !dXnew || !dXi || !dXl || !dXh || (!(reg1 && regH) && !dA) || (Z && !dZ))
        FAIL(c, "flex_solve: device allocation failed");
    if (pin_reserve(c, 16)) return -2;
    H2D(c, dM, M, nM * sizeof(double));
    H2D(c, dB, B, nB * sizeof(double));
    H2D(c, dC, C, (size_t)nUnit * nn * sizeof(double));
    H2D(c, dTn, Tn, (size_t)nNode * 6 * n * sizeof(double));
    H2D(c, dOff, nodeOff, ((size_t)nUnit + 1) * sizeof(int64_t));
    H2D(c, dNodeUnit, hNodeUnit.data(), (size_t)nNode * sizeof(int));
    H2D(c, dFlin, F_lin, (size_t)nSys * nHead * nxs * sizeof(cplx));
    HIPCHK(c, hipMemsetAsync(dNi, 0, (size_t)nSys * sizeof(int), c->stream));
    HIPCHK(c, hipMemsetAsync(dFl, 0, (size_t)nSys * sizeof(int), c->stream));
    HIPCHK(c, hipMemsetAsync(dBd, 0, (size_t)nSys * nn * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(dXi, 0, (size_t)nSys * nxs * sizeof(cplx), c->stream));
    HIPCHK(c, hipMemsetAsync(dFd, 0, (size_t)nSys * nHead * nxs * sizeof(cplx), c->stream));
    {
        std::vector<int> ones((size_t)nSys, 1);
        H2D(c, dAct, ones.data(), (size_t)nSys * sizeof(int));
        HIPCHK(c, hipStreamSynchronize(c->stream));                       // (the host vectors above go out of scope)
    }
    const size_t nxl = (size_t)nSys * nxs;
    if (c->flexXl0_n) {                                                   // raftx_flex_start: an explicit iterate, one-shot
        const size_t had = c->flexXl0_n;
        c->flexXl0_n = 0;
        if (had != nxl) FAIL(c, "flex_solve: the linearisation point of raftx_flex_start has %zu entries, this call %zu", had, nxl);
        HIPCHK(c, hipMemcpyAsync(dXl, c->flexXl0, nxl * sizeof(cplx), hipMemcpyDeviceToDevice, c->stream));
    } else {
        hipLaunchKernelGGL(k_flex_fill, dim3((unsigned)((nxl + 255) / 256)), dim3(256), 0, c->stream, nxl, cplx{XiStart, 0.0}, dXl);   // :999
    }
    hipEvent_t e0 = c->evG2, e1 = c->evG3;                                // (free here: the span of the whole fixed point)
    HIPCHK(c, hipEventRecord(e0, c->stream));
    const int nt = (n + 15) / 16;
    const dim3 gridB((unsigned)((nt * nt + 3) / 4), (unsigned)nSys), gridF((unsigned)((nxs + 255) / 256), (unsigned)(nSys * nHead));
    volatile int *hCount = reinterpret_cast<volatile int *>(c->pin);
    int *dIter = sc.alloc<int>(1);
    if (!dIter) FAIL(c, "flex_solve: device allocation failed");
    HIPCHK(c, hipMemsetAsync(dIter, 0, sizeof(int), c->stream));
    // one iteration: node motions -> linearisation of every (node, sea state) -> projections with the units' T -> dense solves ->
    // convergence test / relaxation -> the count of pairs still iterating to the host
    auto enqueue_iteration = [&]() -> int {
        hipLaunchKernelGGL(k_flex_node_motion, dim3((unsigned)npn), dim3(256), 0, c->stream, nCase, n, nw, dNodeUnit, dTn, dXl, dXiN);
        if (linearize_enqueue(c, dXiN, dBn, dFn, false)) return -1;
        hipLaunchKernelGGL(k_flex_w, dim3((unsigned)npn), dim3(256), 0, c->stream, nNode, nCase, n, dTn, dBn, dW);
        hipLaunchKernelGGL(k_flex_gemm_B, gridB, dim3(256), 0, c->stream, nNode, nCase, n, dOff, dTn, dW, dAct, dBd);
        hipLaunchKernelGGL(k_flex_project_F, gridF, dim3(256), 0, c->stream, nCase, nHead, n, nw, dOff, dTn, dFn, dFlin, dAct, dFw, dFd, dRhs);
        if (dense_launch(c, nSys, nCase, n, 1, nw, T.w, dM, dB, dC, freq_mask, dBd, dRhs, dXnew, nullptr, dA, dAct)) return -1;
        HIPCHK(c, hipMemsetAsync(dCount, 0, sizeof(int), c->stream));
        hipLaunchKernelGGL(k_flex_converge, dim3((unsigned)nSys), dim3(256), 0, c->stream, n, nw, tol, dIter, dXnew, dXi, dXl, dAct, dNi, dFl,
                           dCount);
        hipLaunchKernelGGL(k_flex_tick, dim3(1), dim3(1), 0, c->stream, dIter);
        HIPCHK(c, hipMemcpyAsync(const_cast<int *>(hCount), dCount, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        return 0;
    };
    // The launches of an iteration are the same every time (the iteration number lives in device memory), so they can be
    // captured ONCE into a hipGraph and replayed: RAFTX_FLEX_GRAPH=1.  Measured (scripts/bench_flex.py, prof_flex_dropin.py): the
    // capture + instantiation of every call costs more than the four or five replays save -- 13.5 against 13.1 ms for a batch
    // of 16 units x 3 sea states, 2.5-2.9 against 2.3 ms for a single unit -- so plain launches are the default (a graph kept
    // across calls of one shape would pay; not built).  The first iteration always runs plainly (it sets the kernels' LDS
    // attributes, which a capture must not contain).
    static const bool use_graph = getenv("RAFTX_FLEX_GRAPH") && atoi(getenv("RAFTX_FLEX_GRAPH"));
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    int rc_it = 0;
    for (int it = 0; it <= nIter && !rc_it; it++) {                       // :977, 1052
        if (it == 1 && use_graph && nIter > 1) {
            if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int rcq = enqueue_iteration();
                const hipError_t ee = hipStreamEndCapture(c->stream, &graph);
                if (rcq || ee != hipSuccess || !graph || hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0) != hipSuccess) gexec = nullptr;
                (void)hipGetLastError();
            }
        }
        if (gexec && it >= 1) {
            if (hipGraphLaunch(gexec, c->stream) != hipSuccess) rc_it = -2;
        } else {
            rc_it = enqueue_iteration();
        }
        if (!rc_it && hipStreamSynchronize(c->stream) != hipSuccess) rc_it = -2;
        if (!rc_it && *hCount == 0) break;
    }
    if (gexec) (void)hipGraphExecDestroy(gexec);
    if (graph) (void)hipGraphDestroy(graph);
    if (rc_it) FAIL(c, "flex_solve: an iteration could not be enqueued (%s)", hipGetErrorString(hipGetLastError()));
    // every heading with the impedance of the pair's last iteration (:1155, 1191, 1212-1216)
    if (dense_launch(c, nSys, nCase, n, nHead, nw, T.w, dM, dB, dC, freq_mask, dBd, dFw, dXh, dZ, dA)) return -1;
    HIPCHK(c, hipEventRecord(e1, c->stream));
    D2H(c, Xi, dXh, (size_t)nSys * nHead * nxs * sizeof(cplx));
    D2H(c, niter, dNi, (size_t)nSys * sizeof(int));
    D2H(c, flags, dFl, (size_t)nSys * sizeof(int));
    if (B_drag) D2H(c, B_drag, dBd, (size_t)nSys * nn * sizeof(double));
    if (F_drag) D2H(c, F_drag, dFd, (size_t)nSys * nHead * nxs * sizeof(cplx));
    if (Z) D2H(c, Z, dZ, (size_t)nSys * nn * nw * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    c->last_ms = ms;
    return 0;
}

extern "C" int raftx_solve_system_resident(raftx_ctx *c, int nUnit, const double *Mc, const double *Bc, const double *Cc,
                                           raftx_c128 *Xi) {
    if (!c) return -1;
    const DevTables &T = c->T;
    // what the coupled solve is fed from: the exported impedances (RAFTX_WANT_Z), or -- no Z kept -- the units' constant
    // matrices + the exported B_drag (RAFTX_WANT_BDRAG), assembled on the fly; frequency-dependent M(w), B(w) need the export
    // -- decided on what the LAST solve wrote, not on which buffers exist: a same-shape solve that asked for less keeps the
    // wider buffers of the call before it, with that call's contents
    const int lm = c->r_last_mask;
    const bool haveZ = c->rZ && (lm & RAFTX_WANT_Z), haveF = c->rFw && (lm & RAFTX_WANT_FWAVE);
    const bool assemble = c->rXi && !haveZ && c->rB && (lm & RAFTX_WANT_BDRAG) && haveF && !T.MBw;
    if (!c->rXi || !haveF || (!haveZ && !assemble))
        FAIL(c, "solve_system_resident: needs resident F_wave and either Z or (constant matrices) B_drag "
                "(raftx_solve_dynamics_device with RAFTX_WANT_FWAVE | RAFTX_WANT_Z, or RAFTX_WANT_FWAVE | RAFTX_WANT_BDRAG)");
    if (nUnit < 1 || T.nDesign % nUnit != 0 || !Xi) FAIL(c, "solve_system_resident: bad arguments (nDesign=%d, nUnit=%d)", T.nDesign, nUnit);
    HIPCHK(c, hipSetDevice(c->device));
    const int nGroup = T.nDesign / nUnit, nSys = nGroup * T.nCase, n = 6 * nUnit, nRhs = T.nHead, nw = T.nw;
    size_t lds = 0;
    const int nbin = solve_system_shape(n, nRhs, &lds);
    if (lds > 150 * 1024) FAIL(c, "solve_system: %d DOFs x %d rhs does not fit the LDS-resident solver", n, nRhs);
    Scratch sc(c);
    size_t nf = (size_t)nSys * nRhs * n * nw, nc = (size_t)nGroup * n * n;
    cplx *dX = sc.alloc<cplx>(nf);
    double *dM = Mc ? sc.alloc<double>(nc) : nullptr, *dB = Bc ? sc.alloc<double>(nc) : nullptr,
           *dC = Cc ? sc.alloc<double>(nc) : nullptr;
    if (nSys && (!dX || (Mc && !dM) || (Bc && !dB) || (Cc && !dC))) FAIL(c, "solve_system_resident: device allocation failed");
    if (dM) H2D(c, dM, Mc, nc * sizeof(double));
    if (dB) H2D(c, dB, Bc, nc * sizeof(double));
    if (dC) H2D(c, dC, Cc, nc * sizeof(double));
    const cplx *Zsrc = c->rZ;
    const bool rows = solve_system_rows_ok(nUnit, nRhs);
    if (assemble && !rows && nSys) {                     // no register-resident solver for this shape: materialise Z once
        const size_t nz = (size_t)T.nDesign * T.nCase * 36 * nw;
        cplx *dZ = sc.alloc<cplx>(nz);
        if (!dZ) FAIL(c, "solve_system_resident: device allocation failed");
        hipLaunchKernelGGL(k_assemble_unit_z, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, c->stream, T.nDesign * T.nCase, T.nCase, nw,
                           T.w, T.M0, T.B0, T.C0, c->rB, dZ);
        Zsrc = dZ;
    }
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    bool done = !nSys;
    if (!done && rows)
        done = assemble ? launch_solve_system_rows<true, true>(c->stream, nSys, nUnit, nRhs, nw, T.nCase, T.w, nullptr, dM, dB, dC, c->rFw, dX,
                                                               T.M0, T.B0, T.C0, c->rB)
                        : launch_solve_system_rows<true>(c->stream, nSys, nUnit, nRhs, nw, T.nCase, T.w, c->rZ, dM, dB, dC, c->rFw, dX);
    if (!done) {
        if (lds > 64 * 1024)
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve_system<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_solve_system<true>, dim3((unsigned)((size_t)nSys * ((nw + nbin - 1) / nbin))), dim3(64 * nbin), lds,
                           c->stream, nSys, nUnit, nRhs, nw, T.nCase, T.w, Zsrc, dM, dB, dC, c->rFw, dX);
    }
    if (finish_timed(c)) return -2;
    if (nSys) D2H(c, Xi, dX, nf * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int raftx_qtf_kay(raftx_ctx *c, int nSet, int nw2, const double *w2, const double *k2, double depth, double rho,
                             double g, const int64_t *itemOff, const double *items, const double *beta, int Nm,
                             raftx_c128 *kay_out) {
    if (!c) return -1;
    if (nSet < 0 || nw2 < 1 || !w2 || !k2 || !itemOff || !beta) FAIL(c, "qtf_kay: bad arguments");
    if (Nm < 0 || Nm + 2 > KAY_MAXN) FAIL(c, "qtf_kay: Nm=%d outside 0..%d", Nm, KAY_MAXN - 2);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nItem = (size_t)itemOff[nSet], nq = (size_t)nSet * nw2 * nw2 * 6;
    if (nItem && !items) FAIL(c, "qtf_kay: missing items");
    c->kay_ready = false;
    if (c->rKay_n < nq || !c->rKay) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->rKay) (void)hipFree(c->rKay);
        c->rKay = nullptr;
        void *p_ = nullptr;
        HIPCHK(c, hipMalloc(&p_, (nq ? nq : 1) * sizeof(cplx)));
        c->rKay = reinterpret_cast<cplx *>(p_);
        c->rKay_n = nq;
    }
    Scratch sc(c);
    double *dw = sc.alloc<double>(nw2), *dk = sc.alloc<double>(nw2), *dI = sc.alloc<double>(nItem * QK_N),
           *dB = sc.alloc<double>(nSet);
    int64_t *dio = sc.alloc<int64_t>(nSet + 1);
    cplx *dH = sc.alloc<cplx>(nItem * KAY_ROWS * nw2);
    double *dq = sc.alloc<double>(nw2);
    if (nSet && (!dw || !dk || !dB || !dio || !dq || (nItem && (!dI || !dH)))) FAIL(c, "qtf_kay: device allocation failed");
    if (nSet) {
        H2D(c, dw, w2, nw2 * sizeof(double));
        H2D(c, dk, k2, nw2 * sizeof(double));
        H2D(c, dB, beta, nSet * sizeof(double));
        H2D(c, dio, itemOff, (nSet + 1) * sizeof(int64_t));
        if (nItem) H2D(c, dI, items, nItem * QK_N * sizeof(double));
        HIPCHK(c, hipMemsetAsync(c->rKay, 0, nq * sizeof(cplx), c->stream));       // lower triangle stays zero
    }
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (nSet) {
        if (nItem) hipLaunchKernelGGL(k_kay_tables, dim3((unsigned)nItem), dim3(128), 0, c->stream, nw2, Nm + 2, nSet, depth, dk, dio, dI, dB, dH, dq);
        if (nItem) hipLaunchKernelGGL(k_kay_pairs, dim3((unsigned)((size_t)nSet * nw2)), dim3(nw2 > 64 ? 128 : 64), 0, c->stream, nw2, Nm,
                                      depth, rho, g, dw, dk, dio, dI, dH, dq, c->rKay);
    }
    if (finish_timed(c)) return -2;
    if (nSet && kay_out) D2H(c, kay_out, c->rKay, nq * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->rKay_sets = nSet;
    c->rKay_nw2 = nw2;
    c->kay_ready = true;
    return 0;
}

static int qtf_slender_impl(raftx_ctx *c, int nSet, int nw2, const double *w2, const double *k2, double depth,
                            double rho, double g, const int64_t *stripOff, const double *strips,
                            const int64_t *memOff, const double *members, const raftx_c128 *Xi,
                            const double *beta, const double *Mstruc, const raftx_c128 *kay, int row_off, int row_stride,
                            raftx_c128 *qtf) {
    if (!c) return -1;
    if (nSet < 0 || nw2 < 1 || !w2 || !k2 || !stripOff || !memOff || !beta || !Mstruc)
        FAIL(c, "qtf_slender: bad arguments");
    if (!Xi && (!c->rXi || !c->have_cases || c->r_npair != (size_t)nSet))
        FAIL(c, "qtf_slender: Xi == NULL asks for the RAOs of the resident responses, but %s",
             !c->rXi ? "nothing is resident" : "the number of sets differs from the resident (design, case) pairs");
    if (row_stride < 1 || row_off < 0 || row_off >= row_stride) FAIL(c, "qtf_slender: bad row partition %d/%d", row_off, row_stride);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nStrip = (size_t)stripOff[nSet], nMem = (size_t)memOff[nSet];
    if ((nStrip && !strips) || (nMem && !members)) FAIL(c, "qtf_slender: missing tables");
    std::vector<int> sset(nStrip), mset(nMem);
    for (int s = 0; s < nSet; s++) {
        if (stripOff[s + 1] < stripOff[s] || memOff[s + 1] < memOff[s]) FAIL(c, "qtf_slender: offsets not monotone");
        for (int64_t i = stripOff[s]; i < stripOff[s + 1]; i++) sset[(size_t)i] = s;
        for (int64_t i = memOff[s]; i < memOff[s + 1]; i++) mset[(size_t)i] = s;
    }
    Scratch sc(c);
    const size_t nq = (size_t)nSet * nw2 * nw2 * 6;
    QtfArgs A;
    A.nSet = nSet;
    A.nw = nw2;
    A.depth = depth;
    A.rho = rho;
    A.g = g;
    double *dw = sc.alloc<double>(nw2), *dk = sc.alloc<double>(nw2), *dS = sc.alloc<double>(nStrip * QS_N),
           *dM = sc.alloc<double>(nMem * QM_N), *dB = sc.alloc<double>(nSet), *dMs = sc.alloc<double>((size_t)nSet * 36);
    int64_t *dso = sc.alloc<int64_t>(nSet + 1), *dmo = sc.alloc<int64_t>(nSet + 1);
    int *dss = sc.alloc<int>(nStrip), *dms = sc.alloc<int>(nMem);
    const bool use_resident_kay = !kay && c->kay_ready && c->rKay_sets == nSet && c->rKay_nw2 == nw2;
    c->kay_ready = false;                                // one-shot: a table never outlives the call it was made for
    cplx *dXi = sc.alloc<cplx>((size_t)nSet * 6 * nw2), *dK = kay ? sc.alloc<cplx>(nq) : (use_resident_kay ? c->rKay : nullptr);
    cplx *dT = sc.alloc<cplx>(nStrip * QT_N * nw2), *dTM = sc.alloc<cplx>(nMem * QTM_N * nw2),
         *dTS = sc.alloc<cplx>((size_t)nSet * QTS_N * nw2);
    double *dD = sc.alloc<double>(nStrip * QD_N);
    cplx *dTA = sc.alloc<cplx>(nStrip * QT_N * nw2);
    if (c->rQtf_n < nq || !c->rQtf) {                    // the result stays resident (raftx_qtf_force can reuse it)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->rQtf) (void)hipFree(c->rQtf);
        c->rQtf = nullptr;
        void *p_ = nullptr;
        HIPCHK(c, hipMalloc(&p_, (nq ? nq : 1) * sizeof(cplx)));
        c->rQtf = reinterpret_cast<cplx *>(p_);
        c->rQtf_n = nq;
    }
    c->rQtf_sets = nSet;
    c->rQtf_nw2 = nw2;
    cplx *dQ = c->rQtf;
    if (nSet && (!dw || !dk || (nStrip && (!dS || !dss || !dT || !dD || !dTA)) || (nMem && (!dM || !dms || !dTM)) || !dB || !dMs || !dso ||
                 !dmo || !dXi || (kay && !dK) || !dTS || !dQ))
        FAIL(c, "qtf_slender: device allocation failed");
    if (nSet) {
        H2D(c, dw, w2, nw2 * sizeof(double));
        H2D(c, dk, k2, nw2 * sizeof(double));
        if (nStrip) H2D(c, dS, strips, nStrip * QS_N * sizeof(double));
        if (nMem) H2D(c, dM, members, nMem * QM_N * sizeof(double));
        H2D(c, dB, beta, nSet * sizeof(double));
        H2D(c, dMs, Mstruc, (size_t)nSet * 36 * sizeof(double));
        H2D(c, dso, stripOff, (nSet + 1) * sizeof(int64_t));
        H2D(c, dmo, memOff, (nSet + 1) * sizeof(int64_t));
        if (nStrip) H2D(c, dss, sset.data(), nStrip * sizeof(int));
        if (nMem) H2D(c, dms, mset.data(), nMem * sizeof(int));
        if (Xi) H2D(c, dXi, Xi, (size_t)nSet * 6 * nw2 * sizeof(cplx));
        if (kay) H2D(c, dK, kay, nq * sizeof(cplx));
    }
    A.w = dw; A.k = dk; A.soff = dso; A.strips = dS; A.moff = dmo; A.members = dM; A.sset = dss; A.mset = dms;
    A.Xi = dXi; A.beta = dB; A.Ms = dMs; A.kay = dK; A.T = dT; A.TA = dTA; A.D = dD; A.TM = dTM; A.TS = dTS; A.qtf = dQ;
    A.row_off = row_off;
    A.row_stride = row_stride;
    A.nrow = (nw2 - row_off + row_stride - 1) / row_stride;           // rows w1 = row_off + m*row_stride of this call
    if (A.nrow < 0) A.nrow = 0;
    if (nSet && row_stride > 1) HIPCHK(c, hipMemsetAsync(dQ, 0, nq * sizeof(cplx), c->stream));   // other ranks' rows stay 0
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (nSet && !Xi)                                     // motion RAOs straight from the resident first-order responses
        hipLaunchKernelGGL(k_rao_to_grid, dim3((unsigned)((size_t)nSet * 6)), dim3(128), 0, c->stream, c->T.nCase, c->T.nHead,
                           c->T.nw, nw2, c->T.w, c->T.zeta, dw, c->rXi, dXi);
    if (nSet) {
        hipLaunchKernelGGL(k_qtf_tables, dim3((unsigned)(nStrip + nMem + nSet)), dim3(nw2 > 128 ? 256 : 128), 0, c->stream, A,
                           (int)nStrip, (int)nMem);
        if (A.nrow) {
            static const int blk_env = getenv("RAFTX_QTF_BLOCK") ? atoi(getenv("RAFTX_QTF_BLOCK")) : 0;      // tuning: 64 | 128
            const int blk = (blk_env == 64 || blk_env == 128) ? blk_env : (nw2 > 64 ? 128 : 64);
            const size_t per_xcd = ((size_t)nSet * A.nrow + 7) / 8;                 // see k_qtf_pairs: a slab of the (set, row) list per XCD
            hipLaunchKernelGGL(k_qtf_pairs, dim3((unsigned)(per_xcd * 8)), dim3(blk), 0, c->stream, A);
        }
    }
    if (finish_timed(c)) return -2;
    if (nSet && qtf) D2H(c, qtf, dQ, nq * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int raftx_qtf_slender(raftx_ctx *c, int nSet, int nw2, const double *w2, const double *k2, double depth,
                                 double rho, double g, const int64_t *stripOff, const double *strips,
                                 const int64_t *memOff, const double *members, const raftx_c128 *Xi,
                                 const double *beta, const double *Mstruc, const raftx_c128 *kay, raftx_c128 *qtf) {
    return qtf_slender_impl(c, nSet, nw2, w2, k2, depth, rho, g, stripOff, strips, memOff, members, Xi, beta, Mstruc, kay, 0, 1, qtf);
}

extern "C" int raftx_qtf_slender_rows(raftx_ctx *c, int nSet, int nw2, const double *w2, const double *k2, double depth,
                                      double rho, double g, const int64_t *stripOff, const double *strips,
                                      const int64_t *memOff, const double *members, const raftx_c128 *Xi,
                                      const double *beta, const double *Mstruc, const raftx_c128 *kay, int row_off,
                                      int row_stride, raftx_c128 *qtf) {
    return qtf_slender_impl(c, nSet, nw2, w2, k2, depth, rho, g, stripOff, strips, memOff, members, Xi, beta, Mstruc, kay,
                            row_off, row_stride, qtf);
}

extern "C" int raftx_qtf_force(raftx_ctx *c, int nSet, int nw2, const double *w2, const raftx_c128 *qtf, int nw,
                               const double *w, double dw, const double *S0, double *f_mean, double *f) {
    if (!c) return -1;
    if (nSet < 0 || nw2 < 2 || nw < 1 || !w2 || !w || !S0 || !f_mean || !f) FAIL(c, "qtf_force: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nq = (size_t)nSet * nw2 * nw2 * 6;
    Scratch sc(c);
    const cplx *dQ = nullptr;
    if (qtf) {
        cplx *t = sc.alloc<cplx>(nq);
        if (nq && !t) FAIL(c, "qtf_force: device allocation failed");
        if (nq) H2D(c, t, qtf, nq * sizeof(cplx));
        dQ = t;
    } else {
        if (!c->rQtf || c->rQtf_sets != nSet || c->rQtf_nw2 != nw2)
            FAIL(c, "qtf_force: no resident QTFs of this shape (run raftx_qtf_slender first or pass qtf)");
        dQ = c->rQtf;
    }
    double *dw2 = sc.alloc<double>(nw2), *dwv = sc.alloc<double>(nw), *dS = sc.alloc<double>((size_t)nSet * nw),
           *dfm = sc.alloc<double>((size_t)nSet * 6), *df = sc.alloc<double>((size_t)nSet * 6 * nw);
    if (nSet && (!dw2 || !dwv || !dS || !dfm || !df)) FAIL(c, "qtf_force: device allocation failed");
    if (nSet) {
        H2D(c, dw2, w2, nw2 * sizeof(double));
        H2D(c, dwv, w, nw * sizeof(double));
        H2D(c, dS, S0, (size_t)nSet * nw * sizeof(double));
    }
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (nSet) {
        const size_t lds = sizeof(double) * ((size_t)nw + (nw + 1) / 2 + 8);
        hipLaunchKernelGGL(k_qtf_force, dim3((unsigned)(nSet * 6)), dim3(256), lds, c->stream, nSet, nw2, dw2, dQ, nw, dwv, dw,
                           dS, dfm, df);
    }
    if (finish_timed(c)) return -2;
    if (nSet) {
        D2H(c, f_mean, dfm, (size_t)nSet * 6 * sizeof(double));
        D2H(c, f, df, (size_t)nSet * 6 * nw * sizeof(double));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------ one-call sweep crossing
// raftx_sweep_stats: the designs are cut into a few blocks, every block owns a block context (device tables, result
// buffers, memory pool, events -- created on first use, kept by the parent ctx) and ONE host thread drives four streams:
//   sCopy   descriptor H2D of every block, back to back (the first block is small, so its tables are ready early)
//   sPrep   member pass + scans of a block as soon as its descriptors have landed, totals to page-locked memory
//   stream  (the ctx stream; every kernel that matters) strip generation, per-design reduction, the fused fixed point
//           and the statistics of block 0, 1, 2, ... strictly one after the other -- the HIP events around each
//           k_solve_dynamics launch therefore time that launch alone
//   sD2H    full responses of a finished block (only when the caller asks for Xi)
// The host waits only for the few bytes of totals that size a block's strip table (they are ready long before the
// compute stream reaches the block) and, at the end, for the streams to drain.
static int block_ctx(raftx_ctx *c, int slot, size_t i, raftx_ctx **out) {
    std::vector<raftx_ctx *> &W = c->workers[slot];
    while (W.size() <= i) {
        raftx_ctx *sub = nullptr;
        const int rc = raftx_ctx_create(c->device, &sub);
        if (rc) FAIL(c, "sweep_stats: cannot create a block context (rc=%d)", rc);
        (void)hipStreamDestroy(sub->stream);          // block contexts run on the parent's stream
        sub->stream = c->stream;
        sub->owns_stream = false;
        W.push_back(sub);
    }
    *out = W[i];
    return 0;
}
// block sizes: RAFTX_SWEEP_SPLIT="f0,f1,..." (fractions, tuning) | nChunk equal blocks | default: ONE block when the other slot
// has a crossing in flight (streamed batches: that crossing's kernels hide this one's upload), otherwise a small first block
// whose kernels hide the descriptor upload of the rest (every further block costs a partial last residency round of the
// fused kernel plus the fixed latencies of the generation kernels: two blocks measured best)
static std::vector<int> sweep_bounds(int nDesign, long pairs, int nChunk, bool pipelined, bool with_xi = false, int nCase = 1) {
    // RAFTX_XI_SLABS=1: round 4's first form for a crossing that downloads its responses with nothing else in flight -- four
    // BLOCKS, 20 / 30 / 30 / 20 %, each block's download under the next block's kernels (6.2-6.4 ms against 7.5 for two
    // blocks downloaded whole).  Superseded by slabs of the fused launch inside the two blocks (SlabPlan, raftx_sweep_launch:
    // 5.7 ms on the box where this form took 6.95); kept for A/B runs.
    static const bool xi_slabs = getenv("RAFTX_XI_SLABS") && atoi(getenv("RAFTX_XI_SLABS"));     // round 4's first form: blocks as slabs
    (void)nCase;
    std::vector<double> fr;
    static const char *env = getenv("RAFTX_SWEEP_SPLIT");
    if (env && nChunk <= 0) {
        const char *p = env;
        while (*p) {
            char *e = nullptr;
            const double v = strtod(p, &e);
            if (e == p) break;
            if (v > 0) fr.push_back(v);
            p = (*e == ',') ? e + 1 : e;
            if (*e != ',') break;
        }
    }
    if (fr.empty()) {
        if (nChunk > 0) fr.assign((size_t)nChunk, 1.0);
        else if (pipelined) fr = {1.0};              // the crossing in the other slot hides this one's upload: one launch, no extra tail
        else if (with_xi && xi_slabs && pairs >= 4096) fr = {0.2, 0.3, 0.3, 0.2};
        else if (pairs >= 3072) fr = {0.2, 0.8};     // measured on MI355X at 10 k pairs (profiles/r02_crossing_splits.txt)
        else fr = {1.0};
    }
    double tot = 0;
    for (double v : fr) tot += v;
    std::vector<int> b{0};
    double acc = 0;
    for (size_t i = 0; i < fr.size(); i++) {
        acc += fr[i];
        int hi = (i + 1 == fr.size()) ? nDesign : (int)llround(acc / tot * nDesign);
        if (hi > nDesign) hi = nDesign;
        if (hi > b.back()) b.push_back(hi);
    }
    if (b.back() != nDesign) b.push_back(nDesign);
    return b;
}

// is a crossing of another slot prepared or solving?
static bool others_in_flight(raftx_ctx *c, int slot) {
    for (int sl = 0; sl < RAFTX_NSLOT; sl++)
        if (sl != slot && (c->slots[sl].busy || c->slots[sl].prepared)) return true;
    return false;
}

// the crossing of slot S no longer reads its sea-state set
static void slot_release_cases(raftx_ctx *c, SweepSlot &S) {
    if (S.cset >= 0 && c->csets[S.cset].users > 0) c->csets[S.cset].users--;
    S.cset = -1;
}

static int sweep_prepare_impl(raftx_ctx *c, int slot, int nDesign, const int64_t *memberOff, const double *members,
                              const int64_t *stationOff, const double *stations, const int64_t *capOff,
                              const double *caps, const double *pose, double rho, double g, int add_mask,
                              const double *M0, const double *B0, const double *C0, const double *Fz_moor, int nCase,
                              int nHead, int nw, const double *w, const double *k, double depth, double rho_wave,
                              double g_wave, const double *zeta, const double *beta, int nIter, double tol,
                              double XiStart, int nChunk, double *sd, int32_t *niter, int32_t *flags,
                              raftx_c128 *Xi, int64_t *stripOffsets, const VariantSrc *var) {
    RangeScope range_("raftx_sweep_prepare: descriptor H2D + member pass (enqueue)");
    if (!c) return -1;
    if (slot < 0 || slot >= RAFTX_NSLOT) FAIL(c, "sweep_prepare: slot must be 0 .. %d", RAFTX_NSLOT - 1);
    SweepSlot &S = c->slots[slot];
    if (S.busy || S.prepared) FAIL(c, "sweep_prepare: slot %d is still in flight (call raftx_sweep_wait first)", slot);
    if (nDesign < 0 || !memberOff || (!members && !var) || !stationOff || (!stations && !var) || !M0 || !B0 || !C0)
        FAIL(c, "sweep_stats: bad design arguments");
    if (nCase < 1 || nHead < 1 || nw < 1 || !w || !k || !zeta || !beta) FAIL(c, "sweep_stats: bad sea-state arguments");
    if (!sd || !niter || !flags) FAIL(c, "sweep_stats: std, niter and flags are required");
    if (!var && (capOff == nullptr) != (caps == nullptr)) FAIL(c, "sweep_stats: capOff and caps must be given together");
    if (nIter < 0) FAIL(c, "sweep_stats: nIter < 0");
    if (nChunk > 64) nChunk = 64;
    HIPCHK(c, hipSetDevice(c->device));
    S.t0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - S.t0).count(); };
    if (!c->sCopy) {
        // All four are ordinary streams.  Measured (scripts/ubench/queue_probe2.hip, scripts/gpu_prep.sh): a HIGH-priority
        // preparation stream whose hardware queue happens to sit apart from the busy one does get the next batch's member
        // pass onto the chip early -- and the step gets SLOWER (4.21 vs 4.02 ms): its 184-VGPR waves break up the
        // 2 x 256-VGPR residency of the fused kernel's workgroups, and the earlier generation / fused kernel of the next
        // batch then only share the chip with the current one.  As it is, the member pass runs beside the fused kernel's
        // last residency round, which costs nothing.
        HIPCHK(c, hipStreamCreateWithFlags(&c->sCopy, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->sPrep, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->sD2H, hipStreamNonBlocking));
        {
            // RAFTX_GEN_PRIORITY=high (tuning): the generation stream in the highest priority class -- the tables of batch
            // i+1 are what the next fused kernel waits for, the member pass of batch i+2 beside them is not.  Measured and NOT
            // the default: the generation then runs INSIDE the running fused kernel and lengthens it (2.93 against 2.71 ms;
            // step 3.13-3.15 against 3.05-3.07, profiles/r06_experiments/gap_design_staging_priority_ab.txt)
            static const char *gp = getenv("RAFTX_GEN_PRIORITY");
            int least = 0, greatest = 0;
            if (gp && !strcmp(gp, "high") && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
                HIPCHK(c, hipStreamCreateWithPriority(&c->sGen, hipStreamNonBlocking, greatest));
            else
                HIPCHK(c, hipStreamCreateWithFlags(&c->sGen, hipStreamNonBlocking));
        }
    }
    if (!S.evXi) HIPCHK(c, hipEventCreate(&S.evXi));
    // sea-state tables: sets resident on the parent, shared by the blocks of every slot that was prepared with the same
    // tables; identical tables are not uploaded again.  The slot pins its set until it is retired (wait / cancel / failure),
    // so other sea states prepared on another slot meanwhile get a set of their own and leave this one alone.
    if (nw > MAX_NW) FAIL(c, "nw=%d exceeds the %d bins per workgroup supported by this build", nw, MAX_NW);
    {
        std::vector<double> key;
        key.reserve((size_t)nw * 2 + (size_t)nCase * nHead * (nw + 1) + 6);
        key.push_back(nCase); key.push_back(nHead); key.push_back(nw); key.push_back(depth); key.push_back(rho_wave); key.push_back(g_wave);
        key.insert(key.end(), w, w + nw);
        key.insert(key.end(), k, k + nw);
        key.insert(key.end(), zeta, zeta + (size_t)nCase * nHead * nw);
        key.insert(key.end(), beta, beta + (size_t)nCase * nHead);
        int hit = -1, idle = -1;
        for (int i = 0; i <= RAFTX_NSLOT; i++) {
            CaseSet &cs = c->csets[i];
            if (!cs.key.empty() && cs.key.size() == key.size() && memcmp(key.data(), cs.key.data(), key.size() * sizeof(double)) == 0) hit = i;
            else if (cs.users == 0 && (idle < 0 || cs.stamp < c->csets[idle].stamp)) idle = i;
        }
        if (hit < 0) {
            if (idle < 0) FAIL(c, "sweep_prepare: no free sea-state set (every set is pinned by a crossing in flight)");   // cannot happen: NSLOT + 1 sets
            CaseSet &cs = c->csets[idle];
            cs.key.clear();
            free_list(c, cs.allocs);                      // users == 0: every crossing that read it has been waited for
            if (int rc = upload_case_tables(c, c->sCopy, cs.allocs, cs.T, nCase, nHead, nw, w, k, depth, rho_wave, g_wave, zeta, beta)) {
                free_list(c, cs.allocs);
                return rc;
            }
            cs.key.swap(key);
            hit = idle;
        }
        c->csets[hit].users++;
        c->csets[hit].stamp = ++c->cset_clock;
        S.cset = hit;
    }
    const DevTables &CT = c->csets[S.cset].T;
    S.bnd = sweep_bounds(nDesign, (long)nDesign * nCase, nChunk, others_in_flight(c, slot), Xi != nullptr, nCase);
    const std::vector<int> &bnd = S.bnd;
    const size_t nB = bnd.size() - 1;
    S.dw = nw > 1 ? w[1] - w[0] : w[0];
    S.nIter = nIter; S.tol = tol; S.XiStart = XiStart;
    S.blk.assign(nB, nullptr);
    std::vector<raftx_ctx *> &blk = S.blk;
    S.nCase = nCase; S.nHead = nHead; S.nw = nw;
    S.sd = sd; S.niter = niter; S.flags = flags; S.Xi = Xi; S.stripOffsets = stripOffsets;
    S.tl[0] = since();
    // the slot's previous crossing has been waited for: its tables and offset arrays are free
    free_list(c, S.allocs);
    auto fail_drain = [&](int rc) {
        (void)hipDeviceSynchronize();
        for (raftx_ctx *sub : blk)
            if (sub) {
                free_list(sub, sub->job.tmp);
                sub->job.active = false;
            }
        slot_release_cases(c, S);
        return rc;
    };
    // ---- the batch's offset arrays: one upload, shared by the blocks
    DevOffsets dOff{nullptr, nullptr, nullptr};
    {
        const int64_t nMemberAll = memberOff[nDesign];
        if (nMemberAll < 0) {
            slot_release_cases(c, S);
            FAIL(c, "sweep_stats: member offsets not monotone");
        }
        int rc = upload_on(c, c->sCopy, S.allocs, memberOff, (size_t)nDesign + 1, &dOff.memberOff);
        rc |= upload_on(c, c->sCopy, S.allocs, stationOff, (size_t)nMemberAll + 1, &dOff.stationOff);
        if (capOff) rc |= upload_on(c, c->sCopy, S.allocs, capOff, (size_t)nMemberAll + 1, &dOff.capOff);
        if (rc) return fail_drain(-2);
    }
    // ---- phase 1: H2D on sCopy, member pass + scans on sPrep.  Every block here -- except for an isolated crossing cut into
    // slabs (responses wanted, nothing else in flight): there only the first two; raftx_sweep_launch enqueues the others one
    // block ahead of the block it launches.  (The ordinary streams share hardware queues: with every block's copies queued
    // first, the first slab's generation sat behind 1.7 ms of uploads -- profiles/r04_iso_timeline.txt.)
    S.p1 = {memberOff, stationOff, capOff, members, stations, caps, pose, M0, B0, C0, Fz_moor, k, rho, g, add_mask, dOff,
            var ? *var : VariantSrc{nullptr, nullptr}};
    const size_t nFirst = (Xi && nB > 2 && !others_in_flight(c, slot)) ? 2 : nB;
    for (size_t b = 0; b < nB; b++)
        if (block_ctx(c, slot, b, &blk[b])) return fail_drain(-1);
    for (size_t b = 0; b < nFirst; b++) {
        raftx_ctx *sub = blk[b];
        const int lo = bnd[b], n = bnd[b + 1] - lo;
        const int rc = build_phase1(sub, c->sCopy, c->sPrep, lo, n, memberOff, members, stationOff, stations, capOff, caps, pose,
                                    rho, g, nw, k, add_mask, M0, B0, C0, nullptr, Fz_moor, &S.p1.dOff, CT.k, var);
        if (rc) {
            snprintf(c->err, sizeof(c->err), "sweep_stats (block %zu): %s", b, sub->err);
            return fail_drain(rc);
        }
    }
    S.next_p1 = nFirst;
    S.tl[1] = since();
    S.prepared = true;
    return 0;
}

extern "C" int raftx_sweep_prepare(raftx_ctx *c, int slot, int nDesign, const int64_t *memberOff, const double *members,
                                  const int64_t *stationOff, const double *stations, const int64_t *capOff,
                                  const double *caps, const double *pose, double rho, double g, int add_mask,
                                  const double *M0, const double *B0, const double *C0, const double *Fz_moor, int nCase,
                                  int nHead, int nw, const double *w, const double *k, double depth, double rho_wave,
                                  double g_wave, const double *zeta, const double *beta, int nIter, double tol,
                                  double XiStart, int nChunk, double *sd, int32_t *niter, int32_t *flags,
                                  raftx_c128 *Xi, int64_t *stripOffsets) {
    return sweep_prepare_impl(c, slot, nDesign, memberOff, members, stationOff, stations, capOff, caps, pose, rho, g, add_mask, M0, B0,
                              C0, Fz_moor, nCase, nHead, nw, w, k, depth, rho_wave, g_wave, zeta, beta, nIter, tol, XiStart, nChunk, sd,
                              niter, flags, Xi, stripOffsets, nullptr);
}

// ---- parametric variants of one base unit (include/raftx.h; raft/parametersweep.py:39-87)
extern "C" int raftx_variant_program(raftx_ctx *c, int nMember, const double *members, const int64_t *stationOff, const double *stations,
                                     const int64_t *capOff, const double *caps, int nParam, const double *endCoef,
                                     const int32_t *endEdit, const double *headCS, const double *diaCoef, const int32_t *diaEdit) {
    if (!c) return -1;
    for (int sl = 0; sl < RAFTX_NSLOT; sl++)
        if (c->slots && (c->slots[sl].busy || c->slots[sl].prepared) && c->slots[sl].p1.var.prog)
            FAIL(c, "variant_program: a batch of variants is still in flight (raftx_sweep_wait first)");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    VariantProg &P = c->vprog;
    free_list(c, P.allocs);
    P = VariantProg();
    if (nMember == 0) return 0;
    if (nMember < 0 || nParam < 0 || nParam > 64 || !members || !stationOff || !stations || (capOff == nullptr) != (caps == nullptr))
        FAIL(c, "variant_program: bad arguments");
    if (!endEdit || !endCoef || !headCS || !diaEdit || !diaCoef) FAIL(c, "variant_program: the edit tables are required (all-zero flags = no edit)");
    const int nM = nMember, nP = nParam;
    if (stationOff[0] != 0 || (capOff && capOff[0] != 0)) FAIL(c, "variant_program: offsets must start at 0");
    for (int m = 0; m < nM; m++)
        if (stationOff[m + 1] < stationOff[m] || (capOff && capOff[m + 1] < capOff[m])) FAIL(c, "variant_program: offsets not monotone");
    const int nSt = (int)stationOff[nM], nCap = capOff ? (int)capOff[nM] : 0;
    std::vector<int> stM((size_t)nSt), cpM((size_t)nCap);
    std::vector<double> stF((size_t)nSt), flF((size_t)nSt), cpF((size_t)nCap);
    for (int m = 0; m < nM; m++) {
        const double L = members[(size_t)m * RAFTX_GM_N + RAFTX_GM_L];
        if (endEdit[m] && !(L > 0.0)) FAIL(c, "variant_program: member %d has no length", m);
        for (int64_t i = stationOff[m]; i < stationOff[m + 1]; i++) {
            stM[(size_t)i] = m;
            stF[(size_t)i] = stations[(size_t)i * RAFTX_GS_N + RAFTX_GS_S] / L;          // fractions of the length: what the deck's
            flF[(size_t)i] = stations[(size_t)i * RAFTX_GS_N + RAFTX_GS_LFILL] / L;      // arbitrary station units mean
        }
        if (capOff)
            for (int64_t i = capOff[m]; i < capOff[m + 1]; i++) {
                cpM[(size_t)i] = m;
                cpF[(size_t)i] = caps[(size_t)i * RAFTX_GC_N + RAFTX_GC_S] / L;
            }
    }
    P.nM = nM; P.nSt = nSt; P.nCap = nCap; P.nP = nP; P.has_caps = capOff != nullptr;
    P.hS.assign(stationOff, stationOff + nM + 1);
    if (capOff) P.hC.assign(capOff, capOff + nM + 1);
    else P.hC.assign((size_t)nM + 1, 0);
    auto up = [&](const void *h, size_t bytes, void **d) -> int {
        *d = nullptr;
        if (!bytes) bytes = 8, h = nullptr;
        HIPCHK(c, hipMalloc(d, bytes));
        P.allocs.push_back(*d);
        if (h) HIPCHK(c, hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
        return 0;
    };
    int rc = 0;
    rc |= up(members, (size_t)nM * RAFTX_GM_N * 8, (void **)&P.gm);
    rc |= up(stations, (size_t)nSt * RAFTX_GS_N * 8, (void **)&P.gs);
    rc |= up(nCap ? caps : nullptr, (size_t)nCap * RAFTX_GC_N * 8, (void **)&P.gc);
    rc |= up(stM.data(), (size_t)nSt * 4, (void **)&P.stMember);
    rc |= up(nCap ? cpM.data() : nullptr, (size_t)nCap * 4, (void **)&P.capMember);
    rc |= up(stF.data(), (size_t)nSt * 8, (void **)&P.stFrac);
    rc |= up(flF.data(), (size_t)nSt * 8, (void **)&P.fillFrac);
    rc |= up(nCap ? cpF.data() : nullptr, (size_t)nCap * 8, (void **)&P.capFrac);
    rc |= up(endCoef, (size_t)nM * 6 * (nP + 1) * 8, (void **)&P.endCoef);
    rc |= up(headCS, (size_t)nM * 2 * 8, (void **)&P.headCS);
    rc |= up(diaCoef, (size_t)nSt * 2 * (nP + 1) * 8, (void **)&P.diaCoef);
    rc |= up(endEdit, (size_t)nM * 4, (void **)&P.endEdit);
    rc |= up(diaEdit, (size_t)nSt * 4, (void **)&P.diaEdit);
    if (rc) {
        free_list(c, P.allocs);
        P = VariantProg();
        return rc;
    }
    return 0;
}
// uniform offsets of n variants, cached on the program
static void variant_offsets(VariantProg &P, int n) {
    if (P.cachedN == n) return;
    P.memberOff.resize((size_t)n + 1);
    P.stationOff.resize((size_t)n * P.nM + 1);
    P.capOff.resize((size_t)n * P.nM + 1);
    for (int d = 0; d <= n; d++) P.memberOff[(size_t)d] = (int64_t)d * P.nM;
    for (int d = 0; d < n; d++)
        for (int m = 0; m < P.nM; m++) {
            P.stationOff[(size_t)d * P.nM + m] = (int64_t)d * P.nSt + P.hS[(size_t)m];
            P.capOff[(size_t)d * P.nM + m] = (int64_t)d * P.nCap + P.hC[(size_t)m];
        }
    P.stationOff[(size_t)n * P.nM] = (int64_t)n * P.nSt;
    P.capOff[(size_t)n * P.nM] = (int64_t)n * P.nCap;
    P.cachedN = n;
}
extern "C" int raftx_expand_variants(raftx_ctx *c, int nDesign, const double *params, double *members, double *stations, double *caps) {
    if (!c) return -1;
    const VariantProg &P = c->vprog;
    if (!P.nM) FAIL(c, "expand_variants: no program (raftx_variant_program first)");
    if (nDesign < 0 || (!params && nDesign && P.nP) || !members || !stations) FAIL(c, "expand_variants: bad arguments");
    if (!nDesign) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    Scratch sc(c);
    double *dp = sc.alloc<double>((size_t)nDesign * std::max(P.nP, 1)), *dg = sc.alloc<double>((size_t)nDesign * P.nM * RAFTX_GM_N),
           *ds_ = sc.alloc<double>((size_t)nDesign * std::max(P.nSt, 1) * RAFTX_GS_N),
           *dc = sc.alloc<double>((size_t)nDesign * std::max(P.nCap, 1) * RAFTX_GC_N);
    if (!dp || !dg || !ds_ || !dc) FAIL(c, "expand_variants: device allocation failed");
    if (P.nP) H2D(c, dp, params, (size_t)nDesign * P.nP * sizeof(double));
    ExpandArgs E;
    memset(&E, 0, sizeof(E));
    E.params = dp;
    expand_args(P, nDesign, dg, ds_, dc, E);
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    launch_expand(E, c->stream);
    if (finish_timed(c)) return -2;
    D2H(c, members, dg, (size_t)nDesign * P.nM * RAFTX_GM_N * sizeof(double));
    if (P.nSt) D2H(c, stations, ds_, (size_t)nDesign * P.nSt * RAFTX_GS_N * sizeof(double));
    if (caps && P.nCap) D2H(c, caps, dc, (size_t)nDesign * P.nCap * RAFTX_GC_N * sizeof(double));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int raftx_sweep_prepare_variants(raftx_ctx *c, int slot, int nDesign, const double *params, const double *pose, double rho,
                                           double g, int add_mask, const double *M0, const double *B0, const double *C0,
                                           const double *Fz_moor, int nCase, int nHead, int nw, const double *w, const double *k,
                                           double depth, double rho_wave, double g_wave, const double *zeta, const double *beta,
                                           int nIter, double tol, double XiStart, int nChunk, double *sd, int32_t *niter,
                                           int32_t *flags, raftx_c128 *Xi, int64_t *stripOffsets) {
    if (!c) return -1;
    VariantProg &P = c->vprog;
    if (!P.nM) FAIL(c, "sweep_prepare_variants: no program (raftx_variant_program first)");
    if (nDesign < 0 || (!params && nDesign && P.nP)) FAIL(c, "sweep_prepare_variants: bad arguments");
    if (slot < 0 || slot >= RAFTX_NSLOT) FAIL(c, "sweep_prepare_variants: slot must be 0 .. %d", RAFTX_NSLOT - 1);
    if (c->slots[slot].busy || c->slots[slot].prepared)   // before the cached offset arrays are touched: the batch on this slot reads them
        FAIL(c, "sweep_prepare_variants: slot %d is still in flight (call raftx_sweep_wait first)", slot);
    for (int sl = 0; sl < RAFTX_NSLOT; sl++)        // the offset arrays of batches in flight are the cached ones: one batch size at a time
        if (sl != slot && (c->slots[sl].busy || c->slots[sl].prepared) && c->slots[sl].p1.var.prog && P.cachedN != nDesign)
            FAIL(c, "sweep_prepare_variants: batches of variants in flight together must have the same size (%d in flight, %d asked)",
                 P.cachedN, nDesign);
    variant_offsets(P, nDesign);
    const VariantSrc var{&P, params};
    return sweep_prepare_impl(c, slot, nDesign, P.memberOff.data(), nullptr, P.stationOff.data(), nullptr,
                              P.has_caps ? P.capOff.data() : nullptr, nullptr, pose, rho, g, add_mask, M0, B0, C0, Fz_moor, nCase, nHead,
                              nw, w, k, depth, rho_wave, g_wave, zeta, beta, nIter, tol, XiStart, nChunk, sd, niter, flags, Xi,
                              stripOffsets, &var);
}

// The compute stream of a crossing: the ctx stream -- or, RAFTX_SWEEP_STREAMS=2, one of two that crossings of consecutive slots
// alternate between, so that their persistent grids can be on the chip together.  Measured in round 6 and NOT the default
// (profiles/r06_experiments/): the fused grids of consecutive 10 000-design batches do not overlap in practice, because what
// lies between them is the chain member pass -> host -> table generation of the NEXT batch, whose kernels cannot get onto a
// chip that a persistent grid fills and so run in its drain whatever the streams (3.06-3.09 ms per step on one stream or
// two, 4 or 12 hardware queues, pipeline depth 3; depth 4: 3.13-3.3); leaving 32-128 workgroup places of the grid free for
// them does not help either -- a 256-thread block needs room on all four SIMDs of a CU, and a CU with three of its four
// pairs still has two SIMDs full (3.14-3.35 ms).  The 1 250-design shard gains 8 % of kernel time and nothing per step.
static hipStream_t slot_stream(raftx_ctx *c, int slot) {
    static const int n_streams = getenv("RAFTX_SWEEP_STREAMS") ? std::max(1, std::min(2, atoi(getenv("RAFTX_SWEEP_STREAMS")))) : 1;
    if (n_streams > 1 && !c->sMainB) (void)hipStreamCreateWithFlags(&c->sMainB, hipStreamNonBlocking);
    return (n_streams > 1 && c->sMainB && (slot & 1)) ? c->sMainB : c->stream;
}

extern "C" int raftx_sweep_launch(raftx_ctx *c, int slot) {
    RangeScope range_("raftx_sweep_launch: generation + fused fixed point + statistics (enqueue)");
    if (!c) return -1;
    if (slot < 0 || slot >= RAFTX_NSLOT) FAIL(c, "sweep_launch: slot must be 0 .. %d", RAFTX_NSLOT - 1);
    SweepSlot &S = c->slots[slot];
    if (!S.prepared) FAIL(c, "sweep_launch: nothing prepared on slot %d (raftx_sweep_prepare first)", slot);
    if (S.cset < 0 || c->csets[S.cset].T.nCase != S.nCase || c->csets[S.cset].T.nHead != S.nHead || c->csets[S.cset].T.nw != S.nw)
        FAIL(c, "sweep_launch: slot %d has lost its sea-state tables (internal error)", slot);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t sM = slot_stream(c, slot);
    const bool two_streams = c->sMainB != nullptr;
    for (raftx_ctx *sub : S.blk)
        if (sub) sub->stream = sM;
    if (!c->evEpoch) {
        HIPCHK(c, hipEventCreate(&c->evEpoch));
        HIPCHK(c, hipEventRecord(c->evEpoch, sM));
    }
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - S.t0).count(); };
    const std::vector<int> &bnd = S.bnd;
    const size_t nB = bnd.size() - 1;
    std::vector<raftx_ctx *> &blk = S.blk;
    const int nCase = S.nCase, nHead = S.nHead, nw = S.nw, nIter = S.nIter;
    const double tol = S.tol, XiStart = S.XiStart, dw = S.dw;
    raftx_c128 *Xi = S.Xi;
    S.prepared = false;
    auto fail_drain = [&](int rc) {
        (void)hipDeviceSynchronize();
        for (raftx_ctx *sub : blk)
            if (sub) {
                free_list(sub, sub->job.tmp);
                sub->job.active = false;
            }
        slot_release_cases(c, S);
        return rc;
    };
    // ---- full responses, if asked for: every block's download is enqueued right behind the block's kernels, on a stream of
    // its own (enqueued after ALL blocks -- as until round 4 -- the first download of an isolated call could not start before
    // the host had seen the member pass of the LAST block, i.e. before every descriptor had been uploaded: 1.6 ms late)
    hipStream_t sDown = nullptr;
    if (Xi) {
        // The bulk download goes to a stream of its own PRIORITY CLASS, created when first needed: priority classes have
        // hardware queues of their own, whereas the ordinary streams of this library share four, and a hardware queue is in
        // order -- the generation and the fused kernel of batch i+1 used to queue behind the 3.4 ms copy of batch i whenever
        // the two streams landed on one queue (round 3: 7.0 ms per step instead of 4.6-5.0).  Created late, it does not move the
        // other streams' queues (the plain step is sensitive to those: +5 % with the generation stream one queue further).
        // Which class: round 3 took the LOWEST, and the copy / kernel timeline of round 4 (profiles/r04_xi_timeline.txt)
        // shows what that costs -- the command processor does not look at a low-priority queue while a 10 000-workgroup grid
        // of the ordinary class is being handed out, so the download of batch i only STARTED 0.26 ms before the end of batch
        // i+1's fused kernel, a whole step late, and then ran beside nothing.  The HIGHEST class is served at once: the copy
        // is a barrier packet and an SDMA transfer, no compute, and starts when the batch's statistics kernel has finished.
        // RAFTX_D2H_PRIORITY = high (default) | low | 0 (the ordinary download stream).
        static const char *d2h_env = getenv("RAFTX_D2H_PRIORITY");
        static const bool d2h_own = !(d2h_env && !strcmp(d2h_env, "0"));
        static const bool d2h_low = d2h_own && d2h_env && !strcmp(d2h_env, "low");
        if (d2h_own && !c->sD2Hlow) {
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
                (void)hipStreamCreateWithPriority(&c->sD2Hlow, hipStreamNonBlocking, d2h_low ? least : greatest);
        }
        sDown = (d2h_own && c->sD2Hlow) ? c->sD2Hlow : c->sD2H;
    }
    // ---- phase 2 + fixed point + statistics of every block, in order, on the ctx stream
    int rc_all = 0;
    static const long slab_pairs = getenv("RAFTX_XI_SLAB_PAIRS") ? atol(getenv("RAFTX_XI_SLAB_PAIRS")) : 1024;
    static const int slab_streams = getenv("RAFTX_XI_SLAB_STREAMS") ? std::min(3, std::max(1, atoi(getenv("RAFTX_XI_SLAB_STREAMS")))) : 1;
    const bool slab_mode = Xi && sDown && slab_pairs > 0 && !others_in_flight(c, slot);
    SlabPlan plan;
    S.slab = slab_mode;
    if (slab_mode) {
        hipStream_t *ss[3] = {&c->sGen, &c->sSlab[0], &c->sSlab[1]};
        for (int i = 0; i < slab_streams; i++) {
            if (!*ss[i]) HIPCHK(c, hipStreamCreateWithFlags(ss[i], hipStreamNonBlocking));
            plan.alts.push_back(*ss[i]);
        }
    }
    // statistics, iteration counts and flags of a block go straight into its page-locked landing area (the kernels store
    // there: no small D2H copy that could queue on the DMA engine behind a bulk download); behind them, on the download
    // stream, the block's responses unless they leave slab by slab
    // RAFTX_STATS_STREAM=1: the statistics of a batch on a stream of their own behind its fused kernel, so that the next
    // batch's fused kernel (same main stream) does not wait for them
    const char *ss_ = getenv("RAFTX_STATS_STREAM");
    const bool stats_side = ss_ && atoi(ss_);
    if (stats_side && !c->sStat) HIPCHK(c, hipStreamCreateWithFlags(&c->sStat, hipStreamNonBlocking));
    auto enqueue_stats = [&](size_t b) -> int {
        raftx_ctx *sub = blk[b];
        const int lo = bnd[b];
        const size_t npair = (size_t)(bnd[b + 1] - lo) * nCase;
        hipStream_t sS = sM;
        hipError_t e = hipSuccess;
        if (stats_side && c->sStat && !slab_mode) {
            sS = c->sStat;
            e = hipStreamWaitEvent(sS, sub->ev1, 0);                   // the end of the block's fused launch (solve_enqueue)
        }
        if (e == hipSuccess) e = hipEventRecord(sub->evS0, sS);
        if (npair) {
            hipLaunchKernelGGL(k_motion_stats, dim3((unsigned)npair), dim3(nw > 128 ? 256 : (nw > 64 ? 128 : 64)), 0, sS,
                               (int)npair, nHead, nw, 1.0 / dw, sub->rXi, sub->pinRes, (double *)nullptr, (const int *)sub->rNi,
                               (const int *)sub->rFl, reinterpret_cast<int *>(sub->pinRes + npair * 6));
        }
        if (e == hipSuccess) e = hipEventRecord(sub->evS1, sS);
        if (e == hipSuccess) e = hipEventRecord(sub->evDone, sS);
        if (e == hipSuccess && Xi && !slab_mode) {
            const size_t p0 = (size_t)lo * nCase;
            e = hipStreamWaitEvent(sDown, sub->evDone, 0);
            if (e == hipSuccess && sub->r_nx)
                e = hipMemcpyAsync(Xi + p0 * nHead * 6 * nw, sub->rXi, sub->r_nx * sizeof(cplx), hipMemcpyDeviceToHost, sDown);
        }
        if (e != hipSuccess) {
            snprintf(sub->err, sizeof(sub->err), "statistics / download of the block: %s", hipGetErrorString(e));
            return -2;
        }
        return 0;
    };
    for (size_t b = 0; b < nB && !rc_all; b++) {
        raftx_ctx *sub = blk[b];
        const int lo = bnd[b], n = bnd[b + 1] - lo;
        const size_t npair = (size_t)n * nCase;
        // measured (profiles/r02_crossing_splits.txt): generating a block's tables beside the fused kernel of the block
        // before it gains nothing (the block's descriptor upload is what it waits for) and inflates the kernel's timed
        // duration, so the default keeps everything on the ctx stream; RAFTX_SWEEP_GEN_OVERLAP=1 turns the overlap on
        // A crossing launched while another one is solving generates its tables on a stream of its own: its member pass
        // ran a step earlier (raftx_sweep_prepare), so the tables can be built in the drain of the running fused kernel
        // and this crossing's fused kernel follows it without a gap.  RAFTX_SWEEP_GEN_OVERLAP=0 keeps the generation on
        // the ctx stream.
        static const bool gen_overlap = !(getenv("RAFTX_SWEEP_GEN_OVERLAP") && !atoi(getenv("RAFTX_SWEEP_GEN_OVERLAP")));
        const bool pipelined = others_in_flight(c, slot);
        int rc = 0;
        if (b == 0) S.tlb.clear();
        if (S.next_p1 < nB && S.next_p1 <= b + 1) {                     // deferred phase 1: keep one block's upload ahead
            const size_t bn = S.next_p1++;
            const int rc1 = build_phase1(blk[bn], c->sCopy, c->sPrep, bnd[bn], bnd[bn + 1] - bnd[bn], S.p1.memberOff, S.p1.members,
                                         S.p1.stationOff, S.p1.stations, S.p1.capOff, S.p1.caps, S.p1.pose, S.p1.rho, S.p1.g, nw, S.p1.k,
                                         S.p1.add_mask, S.p1.M0, S.p1.B0, S.p1.C0, nullptr, S.p1.Fz_moor, &S.p1.dOff,
                                         c->csets[S.cset].T.k, &S.p1.var);
            if (rc1) {
                snprintf(sub->err, sizeof(sub->err), "%s", blk[bn]->err);
                rc = rc1;
            }
        }
        // RAFTX_GEN_EARLY=1 (tuning): no such wait -- behind a PERSISTENT grid the generation cannot start before the first
        // workgroups of that grid leave anyway
        static const bool gen_early = getenv("RAFTX_GEN_EARLY") && atoi(getenv("RAFTX_GEN_EARLY"));
        if (b == 0 && pipelined && gen_overlap && !two_streams && !gen_early) {
            // When does the generation run?  Enqueued now, beside a fused kernel that has only just started, it would be
            // dispatched at once and take LDS from that kernel for its whole run (measured: +0.25 ms on the kernel).  A
            // small kernel queued BEHIND a running big grid is dispatched when that grid has been handed out -- which is
            // when the member pass of the batch prepared last gets onto the chip: the generation waits for that batch's
            // first kernel and so runs in the drain, beside that member pass.
            for (int sl = 0; sl < RAFTX_NSLOT && !rc; sl++)
                if (sl != slot && c->slots[sl].prepared && !c->slots[sl].blk.empty() && c->slots[sl].blk[0])
                    if (hipStreamWaitEvent(c->sGen, c->slots[sl].blk[0]->evZ, 0) != hipSuccess) rc = -2;
        }
        // slabs of a crossing that downloads its responses (sweep_bounds): RAFTX_XI_GEN_OVERLAP=1 generates the tables of slab
        // b + 1 on the side stream while slab b solves -- measured with five slabs: 6.71 against 6.74 ms, so it stays off (the
        // fused kernel's HIP-event time then is that launch alone)
        static const bool xi_gen_overlap = getenv("RAFTX_XI_GEN_OVERLAP") && atoi(getenv("RAFTX_XI_GEN_OVERLAP"));
        const bool gen_side = (pipelined && gen_overlap) || (b > 0 && Xi != nullptr && nB > 2 && xi_gen_overlap);
        S.tlb.push_back(since());
        // (a crossing: no ABI copy of the strip records; with RAFTX_FUSED_GEN=1 the tables are left to the fused kernel itself,
        // raftx_fusedgen.h -- build_phase2 / solve_enqueue decide)
        if (!rc) rc = build_phase2(sub, nullptr, gen_side ? c->sGen : nullptr, 1 | ((nCase == 1 && !slab_mode) ? 2 : 0) | (pipelined ? 4 : 0));
        S.tlb.push_back(since());
        // (with the generation inside the fused kernel nothing of this batch runs in the drain any more: the fused kernel
        // follows the one before it at once, and the next batch's member pass takes the places the drain frees beside it;
        // RAFTX_FUSED_WAIT_MEMBER=1 keeps the wait)
        const char *fw_ = getenv("RAFTX_FUSED_WAIT_MEMBER");
        const bool fused_wait_member = fw_ && atoi(fw_);
        // Round 6: that wait is OFF by default (RAFTX_MEMBER_WAIT=1 restores it).  With the persistent grid the member pass
        // queued behind this batch gets onto the chip in the drain of the fused kernel before it either way, and what has
        // not finished then runs beside this kernel's first workgroups; same box, alternating, three batches in flight:
        // 3.033-3.043 ms per step without the wait against 3.045-3.062 with it, no difference with two batches in flight
        // (profiles/r06_experiments/gap_design_staging_priority_ab.txt).
        static const bool no_member_wait = !(getenv("RAFTX_MEMBER_WAIT") && atoi(getenv("RAFTX_MEMBER_WAIT")));
        if (!rc && b == 0 && pipelined && gen_overlap && !two_streams && !no_member_wait && (!sub->job.gen_deferred || fused_wait_member)) {
            // Small kernels are not dispatched while a big grid is being handed out: whatever of the NEXT batch's member pass
            // has not finished when this batch's fused kernel starts would wait for the whole kernel and stall that batch's
            // launch a step later.  So this fused kernel starts only when the member passes already queued (the batches
            // prepared but not yet launched) are done as well -- they run beside this batch's table generation, in the
            // drain of the fused kernel before.
            for (int sl = 0; sl < RAFTX_NSLOT && !rc; sl++)
                if (sl != slot && c->slots[sl].prepared)
                    for (raftx_ctx *o : c->slots[sl].blk)
                        if (o && hipStreamWaitEvent(sM, o->evTot, 0) != hipSuccess) rc = -2;
        }
        if (!rc) {                                                      // the sea states this crossing was prepared with
            DevTables &T = sub->T;
            const DevTables &P = c->csets[S.cset].T;
            T.nCase = P.nCase; T.nHead = P.nHead; T.nw = P.nw;
            T.w = P.w; T.k = P.k; T.csh = P.csh; T.cch = P.cch; T.zeta = P.zeta; T.beta = P.beta;
            T.depth = P.depth; T.rho = P.rho; T.g = P.g;
            sub->have_cases = true;
            // Responses wanted and nothing else in flight: the block's launch is cut into slabs of the pair list on the slab
            // stream(s), each followed by its own download (SlabPlan).  The download (3.5 ms for 192 MB) is longer than the
            // solve, so the call ends one slab's download after the last slab when the first download starts early and the
            // copy engine is then never left waiting.  Measured (scripts/gpu_r4_slab.sh, 10 000 pairs, one box, median of six
            // calls): whole blocks 7.7 ms; four blocks as slabs (round 4's first form) 6.95; slabs of 512 / 768 / 1024 / 1536
            // pairs on one slab stream 7.9 / 6.2 / 5.7 / 6.1 -- one residency round per slab (256 CUs x 4 pairs) -- and no
            // better on two or three streams (6.0-6.3).  RAFTX_XI_SLAB_PAIRS: pairs per slab (0: whole blocks);
            // RAFTX_XI_SLAB_STREAMS: 1 .. 3.
            if (slab_mode) {
                plan.bnd.clear();
                for (size_t p = 0; p < npair; p += (size_t)slab_pairs) plan.bnd.push_back(p);
                if (plan.bnd.size() > 1 && npair - plan.bnd.back() < (size_t)slab_pairs / 2) plan.bnd.pop_back();   // no sliver at the end
                plan.bnd.push_back(npair);
                size_t *kslab = &S.nSlabEv;
                S.nSlabEv = 0;
                plan.after = [=](size_t q0, size_t q1, hipStream_t s) -> int {
                    if (q1 <= q0) return 0;
                    while (sub->evSlab.size() <= *kslab) {
                        hipEvent_t e = nullptr;
                        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -2;
                        sub->evSlab.push_back(e);
                    }
                    hipEvent_t e = sub->evSlab[(*kslab)++];
                    const size_t per = (size_t)nHead * 6 * nw;
                    if (hipEventRecord(e, s) != hipSuccess || hipStreamWaitEvent(sDown, e, 0) != hipSuccess) return -2;
                    return hipMemcpyAsync(Xi + ((size_t)lo * nCase + q0) * per, sub->rXi + q0 * per, (q1 - q0) * per * sizeof(cplx),
                                          hipMemcpyDeviceToHost, sDown) == hipSuccess ? 0 : -2;
                };
            }
            if (!rc) rc = solve_enqueue(sub, nIter, tol, XiStart, nullptr, 0, slab_mode ? &plan : nullptr);
        }
        if (!rc) {
            const size_t need = npair * 7 + 2;                          // std [npair,6] | niter, flags [npair] int32 each
            if (!sub->pinRes || sub->pinRes_n < need) {
                if (sub->pinRes) (void)hipHostFree(sub->pinRes);
                sub->pinRes = nullptr;
                void *p_ = nullptr;
                if (hipHostMalloc(&p_, need * sizeof(double), hipHostMallocDefault) != hipSuccess) rc = -2;
                sub->pinRes = reinterpret_cast<double *>(p_);
                sub->pinRes_n = need;
            }
        }
        if (!rc && !slab_mode) rc = enqueue_stats(b);                   // (slab mode: after the join, below)
        S.tlb.push_back(since());
        if (rc) {
            snprintf(c->err, sizeof(c->err), "sweep_stats (block %zu): %s", b, sub->err);
            rc_all = rc;
        }
    }
    if (!rc_all && slab_mode) {                                         // the statistics read what the slabs wrote: behind all of them
        if (!blk[0]->evJoin && hipEventCreateWithFlags(&blk[0]->evJoin, hipEventDisableTiming) != hipSuccess) rc_all = -2;
        if (!rc_all) rc_all = slab_join(blk[0], sM, plan.alts);
        if (!rc_all && hipEventRecord(blk[nB - 1]->ev1, sM) != hipSuccess) rc_all = -2;
        for (size_t b = 0; b < nB && !rc_all; b++) {
            rc_all = enqueue_stats(b);
            if (rc_all) snprintf(c->err, sizeof(c->err), "sweep_stats (block %zu): %s", b, blk[b]->err);
        }
    }
    if (!rc_all && Xi && hipEventRecord(S.evXi, sDown) != hipSuccess) rc_all = -2;
    if (rc_all) return fail_drain(rc_all);
    S.tl[2] = since();
    S.busy = true;
    return 0;
}

// raftx_sweep_submit = prepare + launch (the two-call form the earlier rounds had)
extern "C" int raftx_sweep_submit(raftx_ctx *c, int slot, int nDesign, const int64_t *memberOff, const double *members,
                                  const int64_t *stationOff, const double *stations, const int64_t *capOff,
                                  const double *caps, const double *pose, double rho, double g, int add_mask,
                                  const double *M0, const double *B0, const double *C0, const double *Fz_moor, int nCase,
                                  int nHead, int nw, const double *w, const double *k, double depth, double rho_wave,
                                  double g_wave, const double *zeta, const double *beta, int nIter, double tol,
                                  double XiStart, int nChunk, double *sd, int32_t *niter, int32_t *flags,
                                  raftx_c128 *Xi, int64_t *stripOffsets) {
    const int rc = raftx_sweep_prepare(c, slot, nDesign, memberOff, members, stationOff, stations, capOff, caps, pose, rho, g, add_mask,
                                       M0, B0, C0, Fz_moor, nCase, nHead, nw, w, k, depth, rho_wave, g_wave, zeta, beta, nIter, tol,
                                       XiStart, nChunk, sd, niter, flags, Xi, stripOffsets);
    if (rc) return rc;
    return raftx_sweep_launch(c, slot);
}

// Retires a crossing that was prepared and will not be launched: its uploads and member pass are drained, its scratch and
// sea-state set released; the output arrays given at prepare time are not touched.
extern "C" int raftx_sweep_cancel(raftx_ctx *c, int slot) {
    if (!c) return -1;
    if (slot < 0 || slot >= RAFTX_NSLOT) FAIL(c, "sweep_cancel: slot must be 0 .. %d", RAFTX_NSLOT - 1);
    SweepSlot &S = c->slots[slot];
    if (S.busy) FAIL(c, "sweep_cancel: slot %d has been launched (raftx_sweep_wait collects it)", slot);
    if (!S.prepared) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    hipError_t e = hipSuccess;
    for (hipStream_t st : {c->sCopy, c->sPrep})
        if (st && e == hipSuccess) e = hipStreamSynchronize(st);
    for (raftx_ctx *sub : S.blk)
        if (sub) {
            if (sub->sAux && e == hipSuccess) e = hipStreamSynchronize(sub->sAux);
            free_list(sub, sub->job.tmp);
            sub->job.active = false;
        }
    S.prepared = false;
    slot_release_cases(c, S);
    HIPCHK(c, e);
    return 0;
}

extern "C" int raftx_sweep_wait(raftx_ctx *c, int slot, double *timing_ms) {
    RangeScope range_("raftx_sweep_wait: drain + outputs");
    if (!c) return -1;
    if (slot < 0 || slot >= RAFTX_NSLOT) FAIL(c, "sweep_wait: slot must be 0 .. %d", RAFTX_NSLOT - 1);
    SweepSlot &S = c->slots[slot];
    if (!S.busy) FAIL(c, "sweep_wait: nothing submitted on slot %d", slot);
    HIPCHK(c, hipSetDevice(c->device));
    static const bool dbg_host = getenv("RAFTX_SWEEP_DEBUG") != nullptr;
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - S.t0).count(); };
    const size_t nB = S.blk.size();
    S.busy = false;
    hipError_t es = nB ? hipEventSynchronize(S.blk[nB - 1]->evDone) : hipSuccess;     // the ctx stream is in order: all blocks are done
    S.tl[3] = since();
    if (es == hipSuccess && S.Xi) es = hipEventSynchronize(S.evXi);
    if (es == hipSuccess) es = hipGetLastError();
    if (es == hipSuccess) slot_release_cases(c, S);
    if (es != hipSuccess) {
        (void)hipDeviceSynchronize();
        slot_release_cases(c, S);
        for (raftx_ctx *sub : S.blk) {
            free_list(sub, sub->job.tmp);
            sub->job.active = false;
        }
        HIPCHK(c, es);
    }
    double tb = 0, ts = 0, tst = 0;
    S.span[0] = S.span[1] = 0;
    if (nB && c->evEpoch) {
        float a = 0.f, b_ = 0.f;
        HIPCHK(c, hipEventElapsedTime(&a, c->evEpoch, S.blk[0]->ev0));
        HIPCHK(c, hipEventElapsedTime(&b_, c->evEpoch, S.blk[nB - 1]->ev1));
        S.span[0] = a;
        S.span[1] = b_;
    }
    if (S.stripOffsets) S.stripOffsets[0] = 0;
    for (size_t b = 0; b < nB; b++) {
        raftx_ctx *sub = S.blk[b];
        const int lo = S.bnd[b], n = S.bnd[b + 1] - lo;
        const size_t npair = (size_t)n * S.nCase, p0 = (size_t)lo * S.nCase;
        float ms = 0.f;
        // slabs on their own streams: the span from the first block's first launch to the last slab's end (the generation
        // of the later blocks runs inside it)
        if (!S.slab) HIPCHK(c, hipEventElapsedTime(&ms, sub->ev0, sub->ev1));
        else if (b + 1 == nB) HIPCHK(c, hipEventElapsedTime(&ms, S.blk[0]->ev0, sub->ev1));
        ts += ms;
        HIPCHK(c, hipEventElapsedTime(&ms, sub->evS0, sub->evS1));
        tst += ms;
        double g_ms = 0.0;
        if (S.stripOffsets)                                             // block-relative offsets of phase 1 -> batch offsets
            for (int i = 0; i < n; i++) S.stripOffsets[lo + i + 1] = S.stripOffsets[lo] + sub->pin[8 + i + 1];
        if (build_retire(sub, &g_ms)) return -2;
        tb += g_ms;
        memcpy(S.sd + p0 * 6, sub->pinRes, npair * 6 * sizeof(double));
        memcpy(S.niter + p0, sub->pinRes + npair * 6, npair * sizeof(int));
        memcpy(S.flags + p0, reinterpret_cast<int *>(sub->pinRes + npair * 6) + npair, npair * sizeof(int));
    }
    const double wall = since();
    if (dbg_host)
        fprintf(stderr, "[raftx_sweep slot %d] host ms since submit: pre %.3f | phase-1 enqueued %.3f | phase-2 enqueued %.3f | ctx stream "
                "reached the end %.3f | done %.3f\n", slot, S.tl[0], S.tl[1], S.tl[2], S.tl[3], wall);
    if (dbg_host && S.tlb.size() > 3) {
        fprintf(stderr, "[raftx_sweep slot %d] launch loop, host ms per block (next upload enqueued | totals seen | block enqueued):", slot);
        for (size_t i = 0; i + 2 < S.tlb.size(); i += 3) fprintf(stderr, "  %.3f %.3f %.3f", S.tlb[i], S.tlb[i + 1], S.tlb[i + 2]);
        fprintf(stderr, "\n");
    }
    if (timing_ms) { timing_ms[0] = wall; timing_ms[1] = tb; timing_ms[2] = ts; timing_ms[3] = tst; }
    c->last_ms = ts;
    return 0;
}

extern "C" int raftx_sweep_solve_span(raftx_ctx *c, int slot, double *start_ms, double *end_ms) {
    if (!c) return -1;
    if (slot < 0 || slot >= RAFTX_NSLOT) FAIL(c, "sweep_solve_span: slot must be 0 .. %d", RAFTX_NSLOT - 1);
    if (c->slots[slot].busy) FAIL(c, "sweep_solve_span: slot %d has not been waited for", slot);
    if (start_ms) *start_ms = c->slots[slot].span[0];
    if (end_ms) *end_ms = c->slots[slot].span[1];
    return 0;
}

extern "C" int raftx_sweep_generation(raftx_ctx *c, int slot, int *blocks_fused, int *blocks) {
    if (!c) return -1;
    if (slot < 0 || slot >= RAFTX_NSLOT) FAIL(c, "sweep_generation: slot must be 0 .. %d", RAFTX_NSLOT - 1);
    int nf = 0, nb = 0;
    for (raftx_ctx *sub : c->slots[slot].blk)
        if (sub) {
            nb++;
            nf += sub->last_gen_fused ? 1 : 0;
        }
    if (blocks_fused) *blocks_fused = nf;
    if (blocks) *blocks = nb;
    return 0;
}

extern "C" int raftx_sweep_stats(raftx_ctx *c, int nDesign, const int64_t *memberOff, const double *members,
                                 const int64_t *stationOff, const double *stations, const int64_t *capOff,
                                 const double *caps, const double *pose, double rho, double g, int add_mask,
                                 const double *M0, const double *B0, const double *C0, const double *Fz_moor, int nCase,
                                 int nHead, int nw, const double *w, const double *k, double depth, double rho_wave,
                                 double g_wave, const double *zeta, const double *beta, int nIter, double tol,
                                 double XiStart, int nChunk, int nWorker, double *sd, int32_t *niter, int32_t *flags,
                                 raftx_c128 *Xi, int64_t *stripOffsets, double *timing_ms) {
    if (!c) return -1;
    (void)nWorker;                                     // reserved (earlier versions drove the blocks from several host threads)
    int slot = 0;                                      // a blocking crossing beside streamed ones takes a free slot
    while (slot < RAFTX_NSLOT - 1 && (c->slots[slot].busy || c->slots[slot].prepared)) slot++;
    const int rc = raftx_sweep_submit(c, slot, nDesign, memberOff, members, stationOff, stations, capOff, caps, pose, rho, g, add_mask,
                                      M0, B0, C0, Fz_moor, nCase, nHead, nw, w, k, depth, rho_wave, g_wave, zeta, beta, nIter, tol,
                                      XiStart, nChunk, sd, niter, flags, Xi, stripOffsets);
    if (rc) return rc;
    return raftx_sweep_wait(c, slot, timing_ms);
}

// ------------------------------------------------------------------ RCCL exchange steps
// librccl is bound at run time (dlopen on the first raftx_comm_* call): single-GPU users never load it.
struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};
static RcclApi *rccl_api(raftx_ctx *c) {
    static RcclApi api;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (api.h) return &api;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        snprintf(c->err, sizeof(c->err), "cannot load librccl: %s", dlerror());
        return nullptr;
    }
#define RCCL_SYM(field, name)                                                             \
    *reinterpret_cast<void **>(&api.field) = dlsym(h, name);                              \
    if (!api.field) {                                                                     \
        snprintf(c->err, sizeof(c->err), "librccl does not export %s", name);             \
        return nullptr;                                                                   \
    }
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    RCCL_SYM(CommInitRank, "ncclCommInitRank")
    RCCL_SYM(CommDestroy, "ncclCommDestroy")
    RCCL_SYM(GetErrorString, "ncclGetErrorString")
    RCCL_SYM(Broadcast, "ncclBroadcast")
    RCCL_SYM(Reduce, "ncclReduce")
    RCCL_SYM(AllReduce, "ncclAllReduce")
    RCCL_SYM(Send, "ncclSend")
    RCCL_SYM(Recv, "ncclRecv")
    RCCL_SYM(GroupStart, "ncclGroupStart")
    RCCL_SYM(GroupEnd, "ncclGroupEnd")
#undef RCCL_SYM
    api.h = h;
    return &api;
}
#define NCCLCHK(ctx, api, call)                                                                               \
    do {                                                                                                      \
        ncclResult_t r_ = (call);                                                                             \
        if (r_ != ncclSuccess) {                                                                              \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call, (api)->GetErrorString(r_), \
                     __FILE__, __LINE__);                                                                     \
            return -7;                                                                                        \
        }                                                                                                     \
    } while (0)

extern "C" int raftx_comm_unique_id(raftx_ctx *c, char *id128) {
    if (!c || !id128) return -1;
    RcclApi *R = rccl_api(c);
    if (!R) return -7;
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId id;
    NCCLCHK(c, R, R->GetUniqueId(&id));
    static_assert(sizeof(id) == RAFTX_COMM_ID_BYTES, "RCCL unique id size");
    memcpy(id128, &id, sizeof(id));
    return 0;
}

extern "C" int raftx_comm_destroy(raftx_ctx *c) {
    if (!c) return -1;
    if (c->comm) {
        RcclApi *R = rccl_api(c);
        if (R) {
            (void)hipSetDevice(c->device);
            (void)hipStreamSynchronize(c->stream);
            (void)R->CommDestroy(reinterpret_cast<ncclComm_t>(c->comm));
        }
        c->comm = nullptr;
    }
    if (c->commFlag) {
        (void)hipFree(c->commFlag);
        c->commFlag = nullptr;
    }
    c->comm_rank = 0;
    c->comm_world = 1;
    return 0;
}

extern "C" int raftx_comm_init(raftx_ctx *c, int rank, int world, const char *id128) {
    if (!c || !id128) return -1;
    if (world < 1 || rank < 0 || rank >= world) FAIL(c, "comm_init: bad rank/world %d/%d", rank, world);
    if (c->comm) FAIL(c, "comm_init: this ctx already has a communicator");
    RcclApi *R = rccl_api(c);
    if (!R) return -7;
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    if (!c->commFlag) {                                   // the status word of comm_agree: allocated HERE so that no exchange
        void *p = nullptr;                                // step can fail on it later, between its peers' collectives
        HIPCHK(c, hipMalloc(&p, sizeof(int)));
        c->commFlag = reinterpret_cast<int *>(p);
    }
    NCCLCHK(c, R, R->CommInitRank(&comm, world, id, rank));
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_world = world;
    return 0;
}

static int comm_ready(raftx_ctx *c, RcclApi **R, int root) {
    if (!c) return -1;
    if (!c->comm) FAIL(c, "no communicator on this ctx (raftx_comm_init)");
    if (root < 0 || root >= c->comm_world) FAIL(c, "root %d outside the communicator (%d ranks)", root, c->comm_world);
    *R = rccl_api(c);
    return *R ? 0 : -7;
}

// Every exchange step starts with the ranks agreeing that all of them can go through with it: each rank validates its
// arguments and allocates its buffers FIRST, then the worst local status is MAX-reduced over the communicator (one int).
// A rank that failed locally (bad argument, allocation) therefore fails the call on EVERY rank, with an error, instead of
// leaving its peers blocked inside a send / receive it never posts.  `local_rc` 0 = ready.
static int comm_agree(raftx_ctx *c, RcclApi *R, int local_rc) {
    // the status word exists since raftx_comm_init.  A local HIP failure on the way INTO the collective is folded into
    // this rank's vote, never returned early: the AllReduce below is always posted, so the peers are not left waiting.
    // What cannot be recovered (documented in include/raftx.h): a rank whose AllReduce itself fails, or that dies.
    int mine = local_rc ? 1 : 0, worst = 0;
    hipError_t e = hipMemcpyAsync(c->commFlag, &mine, sizeof(int), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) {                                // vote "failed" through a device-side fill instead (no host source)
        mine = 1;
        if (!local_rc) {
            snprintf(c->err, sizeof(c->err), "comm: status upload failed: %s", hipGetErrorString(e));
            local_rc = -2;
        }
        (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->commFlag), 1, 1, c->stream);
    }
    NCCLCHK(c, R, R->AllReduce(c->commFlag, c->commFlag, 1, ncclInt, ncclMax, reinterpret_cast<ncclComm_t>(c->comm), c->stream));
    e = hipMemcpyAsync(&worst, c->commFlag, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (local_rc) return local_rc;                    // c->err already says why
    if (e != hipSuccess) FAIL(c, "comm: status download failed: %s", hipGetErrorString(e));
    if (worst) FAIL(c, "comm: another rank could not take part in this exchange step (see its error)");
    return 0;
}
// local preparation of an exchange step: returns its status instead of leaving the function (see comm_agree)
#define COMM_LOCAL(rc, cond, ...)                                  \
    do {                                                           \
        if (!(rc) && (cond)) {                                     \
            snprintf(c->err, sizeof(c->err), __VA_ARGS__);         \
            (rc) = -1;                                             \
        }                                                          \
    } while (0)

extern "C" int raftx_comm_broadcast(raftx_ctx *c, void *buf, size_t bytes, int root) {
    RcclApi *R = nullptr;
    if (int rc = comm_ready(c, &R, root)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    Scratch sc(c);
    int rc = 0;
    char *d = nullptr;
    COMM_LOCAL(rc, !buf && bytes, "comm_broadcast: buf is NULL");
    if (!rc && bytes) {
        d = sc.alloc<char>(bytes);
        COMM_LOCAL(rc, !d, "comm_broadcast: device allocation failed");
    }
    if (int a = comm_agree(c, R, rc)) return a;
    if (!bytes) return 0;
    if (c->comm_rank == root) H2D(c, d, buf, bytes);
    NCCLCHK(c, R, R->Broadcast(d, d, bytes, ncclChar, root, reinterpret_cast<ncclComm_t>(c->comm), c->stream));
    if (c->comm_rank != root) D2H(c, buf, d, bytes);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// rows of every rank onto root, point to point; dsend: this rank's rows already in HBM.  `rc`: status of the caller's own
// preparation.  Everything that can fail locally (counts, root's receive buffer) is done BEFORE the ranks agree to go on;
// inside the ncclGroupStart / ncclGroupEnd pair nothing returns early.
static int gather_rows_dev(raftx_ctx *c, RcclApi *R, const void *dsend, const int64_t *counts, size_t row_bytes, void *recv_host,
                           int root, Scratch &sc, int rc) {
    const int me = c->comm_rank, world = c->comm_world;
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(c->comm);
    size_t total = 0;
    for (int r = 0; r < world && !rc; r++) {
        COMM_LOCAL(rc, counts[r] < 0, "comm_gather: negative count for rank %d", r);
        if (!rc) total += (size_t)counts[r];
    }
    char *dall = nullptr;
    if (!rc && me == root) {
        COMM_LOCAL(rc, !recv_host && total, "comm_gather: recv is NULL on root");
        if (!rc && total) {
            dall = sc.alloc<char>(total * row_bytes);
            COMM_LOCAL(rc, !dall, "comm_gather: device allocation failed");
        }
    }
    if (int a = comm_agree(c, R, rc)) return a;
    ncclResult_t nr = R->GroupStart();
    hipError_t he = hipSuccess;
    if (nr == ncclSuccess) {
        if (me == root) {
            size_t o = 0;
            for (int r = 0; r < world; r++) {
                const size_t nb = (size_t)counts[r] * row_bytes;
                if (nb && nr == ncclSuccess && he == hipSuccess) {
                    if (r == me) he = hipMemcpyAsync(dall + o, dsend, nb, hipMemcpyDeviceToDevice, c->stream);
                    else nr = R->Recv(dall + o, nb, ncclChar, r, comm, c->stream);
                }
                o += nb;
            }
        } else {
            const size_t nb = (size_t)counts[me] * row_bytes;
            if (nb) nr = R->Send(dsend, nb, ncclChar, root, comm, c->stream);
        }
        const ncclResult_t ne = R->GroupEnd();        // always closed, whatever happened inside
        if (nr == ncclSuccess) nr = ne;
    }
    if (nr != ncclSuccess) {
        (void)hipStreamSynchronize(c->stream);
        FAIL(c, "comm_gather: RCCL send / receive failed: %s", R->GetErrorString(nr));
    }
    HIPCHK(c, he);
    if (me == root && total) D2H(c, recv_host, dall, total * row_bytes);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int raftx_comm_gather_rows(raftx_ctx *c, const void *send, const int64_t *counts, size_t row_bytes, void *recv,
                                      int root) {
    RcclApi *R = nullptr;
    if (int rc = comm_ready(c, &R, root)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    Scratch sc(c);
    int rc = 0;
    COMM_LOCAL(rc, !counts || !row_bytes, "comm_gather_rows: bad arguments");
    const size_t nb = rc ? 0 : (size_t)(counts[c->comm_rank] > 0 ? counts[c->comm_rank] : 0) * row_bytes;
    char *d = nullptr;
    if (!rc && nb) {
        COMM_LOCAL(rc, !send, "comm_gather_rows: send is NULL");
        if (!rc) {
            d = sc.alloc<char>(nb);
            COMM_LOCAL(rc, !d, "comm_gather_rows: device allocation failed");
        }
        if (!rc && hipMemcpyAsync(d, send, nb, hipMemcpyHostToDevice, c->stream) != hipSuccess)
            COMM_LOCAL(rc, true, "comm_gather_rows: H2D of this rank's rows failed");
    }
    if (rc && !counts) return comm_agree(c, R, rc);   // nothing to walk: only tell the peers
    return gather_rows_dev(c, R, d, counts, row_bytes, recv, root, sc, rc);
}

extern "C" int raftx_comm_gather_xi(raftx_ctx *c, const int64_t *counts, raftx_c128 *Xi_all, int root) {
    RcclApi *R = nullptr;
    if (int rc = comm_ready(c, &R, root)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    Scratch sc(c);
    int rc = 0;
    COMM_LOCAL(rc, !counts, "comm_gather_xi: counts is NULL");
    COMM_LOCAL(rc, !c->rXi, "comm_gather_xi: no resident results");
    COMM_LOCAL(rc, (size_t)counts[c->comm_rank] != c->r_npair, "comm_gather_xi: counts[%d] = %lld but %zu (design, case) pairs are resident",
               c->comm_rank, (long long)counts[c->comm_rank], c->r_npair);
    if (rc && !counts) return comm_agree(c, R, rc);
    const size_t row_bytes = (size_t)c->T.nHead * 6 * c->T.nw * sizeof(cplx);
    return gather_rows_dev(c, R, c->rXi, counts, row_bytes, Xi_all, root, sc, rc);
}

extern "C" int raftx_comm_reduce_sum(raftx_ctx *c, double *buf, size_t n, int root) {
    RcclApi *R = nullptr;
    if (int rc = comm_ready(c, &R, root)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    Scratch sc(c);
    int rc = 0;
    double *d = nullptr;
    COMM_LOCAL(rc, !buf && n, "comm_reduce_sum: buf is NULL");
    if (!rc && n) {
        d = sc.alloc<double>(n);
        COMM_LOCAL(rc, !d, "comm_reduce_sum: device allocation failed");
    }
    if (int a = comm_agree(c, R, rc)) return a;
    if (!n) return 0;
    H2D(c, d, buf, n * sizeof(double));
    NCCLCHK(c, R, R->Reduce(d, d, n, ncclDouble, ncclSum, root, reinterpret_cast<ncclComm_t>(c->comm), c->stream));
    if (c->comm_rank == root) D2H(c, buf, d, n * sizeof(double));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------ diagnostics
#ifdef RAFTX_PHASE_TIMING
// Tuning builds only (csrc/build.py --timing): cycles of the last raftx_solve_dynamics_device,
// summed over the first lane of every workgroup, per phase (set-up+inertial, pass A, strip
// phase, pass B, solve, tail).
extern "C" int raftx_debug_phase_cycles(raftx_ctx *c, unsigned long long *out8) {
    if (!c || !out8 || !c->dbg) return -1;
    HIPCHK(c, hipMemcpy(out8, c->dbg, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}
// timing builds: the clock trace of the last fused launch -- out[2 b], out[2 b + 1] = shader cycles, 100 MHz ticks of the
// workgroups that started in the b-th 125 us of the launch (PT_FLUSH); n_bucket <= 64
extern "C" int raftx_debug_clock_trace(raftx_ctx *c, unsigned long long *out, int n_bucket) {
    if (!c || !out || !c->dbg || n_bucket < 1 || n_bucket > PT_CLOCK_BUCKETS) return -1;
    HIPCHK(c, hipMemcpy(out, c->dbg + 10, 2 * (size_t)n_bucket * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}
#endif

__global__ void k_debug_math(int n, const double *__restrict__ x, double *__restrict__ s, double *__restrict__ c,
                             double *__restrict__ e, int table) {
    __shared__ __attribute__((aligned(16))) double tab[2 * RAFTX_SC_N];
    stage_sincos_table((ldptr)tab);
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        if (table) tab_sincos(x[i], (ldptr)tab, s[i], c[i]);
        else fast_sincos(x[i], s[i], c[i]);
        e[i] = fast_exp(x[i]);
    }
}

extern "C" int raftx_device_synchronize(raftx_ctx *c) {
    if (!c) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    return 0;
}
extern "C" int raftx_last_solve_kernel(raftx_ctx *c, int *flags, int *waves_per_simd, int *cache_slots) {
    if (!c || c->last_flags < 0) return -1;
    if (flags) *flags = c->last_flags;
    if (waves_per_simd) *waves_per_simd = c->last_minb;
    if (cache_slots) *cache_slots = c->last_rc;
    return 0;
}
static int debug_math(raftx_ctx *c, int n, const double *x, double *sin_out, double *cos_out, double *exp_out, int table);
extern "C" int raftx_debug_math(raftx_ctx *c, int n, const double *x, double *sin_out, double *cos_out, double *exp_out) {
    return debug_math(c, n, x, sin_out, cos_out, exp_out, 0);
}
extern "C" int raftx_debug_flex_gemm(raftx_ctx *c, int K, int n, const double *A, const double *W, double *out) {
    if (!c) return -1;
    if (K < 1 || K % 6 || n < 1 || !A || !W || !out) FAIL(c, "debug_flex_gemm: K must be a multiple of 6");
    HIPCHK(c, hipSetDevice(c->device));
    Scratch sc(c);
    double *dA = sc.alloc<double>((size_t)K * n), *dW = sc.alloc<double>((size_t)K * n), *dO = sc.alloc<double>((size_t)n * n);
    int64_t *dOff = sc.alloc<int64_t>(2);
    if (!dA || !dW || !dO || !dOff) FAIL(c, "debug_flex_gemm: device allocation failed");
    const int64_t off[2] = {0, K / 6};
    H2D(c, dA, A, (size_t)K * n * sizeof(double));
    H2D(c, dW, W, (size_t)K * n * sizeof(double));
    H2D(c, dOff, off, sizeof(off));
    const int nt = (n + 15) / 16;
    hipLaunchKernelGGL(k_flex_gemm_B, dim3((unsigned)((nt * nt + 3) / 4), 1), dim3(256), 0, c->stream, K / 6, 1, n, dOff, dA, dW,
                       (const int *)nullptr, dO);
    D2H(c, out, dO, (size_t)n * n * sizeof(double));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int raftx_debug_math_table(raftx_ctx *c, int n, const double *x, double *sin_out, double *cos_out, double *exp_out) {
    return debug_math(c, n, x, sin_out, cos_out, exp_out, 1);
}
static int debug_math(raftx_ctx *c, int n, const double *x, double *sin_out, double *cos_out, double *exp_out, int table) {
    if (!c || n < 0 || !x || !sin_out || !cos_out || !exp_out) return -1;
    HIPCHK(c, hipSetDevice(c->device));
    Scratch sc(c);
    double *dx = sc.alloc<double>(n), *ds = sc.alloc<double>(n), *dc = sc.alloc<double>(n), *de = sc.alloc<double>(n);
    if (n && (!dx || !ds || !dc || !de)) FAIL(c, "debug_math: device allocation failed");
    if (n) {
        H2D(c, dx, x, n * sizeof(double));
        hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, dx, ds, dc, de, table);
        D2H(c, sin_out, ds, n * sizeof(double));
        D2H(c, cos_out, dc, n * sizeof(double));
        D2H(c, exp_out, de, n * sizeof(double));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
