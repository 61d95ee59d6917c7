// raftx_kernels.h -- gfx950 device code of the RAFT hot path, kernel generation 4
// (included by raftx_hip.hip).
//
// Work decomposition:
//   workgroup <-> one (design, sea state) pair; 64*NWV threads (NWV waves, run time)
//   lane      <-> NB frequency bins (compile time, 1..4): bin(j) = j*blockDim.x + tid,
//                 so every global access of a [.,nw] slab is coalesced
//   strip loop <-> sequential; the strip record index is wave-uniform, so the static
//                 strip constants are read with SCALAR loads (s_load_dwordx8/16 through
//                 the constant cache) straight into SGPRs and feed v_fma_f64 as the
//                 scalar operand: no LDS bandwidth, no VGPRs, amortised over the NB
//                 bins of the lane.  Only data produced inside the workgroup (linearised
//                 drag vectors U,V, XiLast, reduction tiles) lives in LDS.
//
// Maths (reference lines in the function comments): per strip s and bin w the Airy
// kinematics reduce to two phasors
//   t1 = w zeta e^{-i k xi} cosh k(z+h)/sinh kh,   t2 = i w zeta e^{-i k xi} sinh k(z+h)/sinh kh
// (helpers.py:201-228), u = (cb t1, sb t1, t2).  They are advanced strip-to-strip along
// straight members by rotors (e^{-i k du}, e^{+-k dz}); run starts are evaluated exactly.
// Runs are detected by the library at upload from the absolute strip positions.
#pragma once
#include <type_traits>
#include "raftx_tables.h"

#define NF RAFTX_NFIELD

struct cplx {
    double re, im;
};
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cscale(cplx a, double s) { return {a.re * s, a.im * s}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }

// Strip tables are read-only for the whole launch.  Reading them through the CONSTANT address
// space lets the compiler use scalar loads (s_load_dwordx8/16 -> SGPRs) even after the kernel
// has stored to global memory (plain global pointers lose that once any store may alias).
#define CONST_AS __attribute__((address_space(4)))
typedef const CONST_AS double *cdptr;
typedef const CONST_AS int *ciptr;
// LDS pointers carry their address space: 32-bit addresses, always ds_* instructions
#define LDS_AS __attribute__((address_space(3)))
typedef LDS_AS double *ldptr;
typedef LDS_AS int *liptr;
__device__ __forceinline__ cdptr as_const(const double *p) { return (cdptr)p; }
__device__ __forceinline__ ciptr as_const(const int *p) { return (ciptr)p; }

// ------------------------------------------------------------------ fp64 elementary functions
// Straight-line sincos / exp for the moderate arguments of this problem
// (|k xi| < 1e5, |k z| < 700): Cody-Waite reduction + Taylor polynomials whose
// truncation error is < 2^-55 on the reduced interval.  No tables, no branches.
// The run-start evaluations are compiled WITHOUT floating-point contraction (explicit fma calls only): a quantity read
// back from the run-start cache was computed by the sweep that filled it, a quantity that found no slot by the sweep
// that needs it -- the two must be the same bits, or a batch's results would depend on how many cache slots its
// launch happened to get (tests/test_geometry.py: chunked crossings are bit-identical to the plain sequence).
#pragma clang fp contract(off)
__device__ __forceinline__ void fast_sincos(double x, double &s, double &c) {
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00;    // first 33 bits of pi/2
    const double pio2_2 = 6.07710050630396597660e-11;    // second 33 bits
    const double pio2_2t = 2.02226624879595063154e-21;   // pi/2 - (pio2_1 + pio2_2)
    double fn = rint(x * two_over_pi);
    double r = fma(-fn, pio2_1, x);
    r = fma(-fn, pio2_2, r);
    r = fma(-fn, pio2_2t, r);
    int n = (int)fn;
    double z = r * r;
    double ps = -1.0 / 1307674368000.0;                  // -1/15!
    ps = fma(ps, z, 1.0 / 6227020800.0);                 //  1/13!
    ps = fma(ps, z, -1.0 / 39916800.0);                  // -1/11!
    ps = fma(ps, z, 1.0 / 362880.0);                     //  1/9!
    ps = fma(ps, z, -1.0 / 5040.0);                      // -1/7!
    ps = fma(ps, z, 1.0 / 120.0);                        //  1/5!
    ps = fma(ps, z, -1.0 / 6.0);                         // -1/3!
    double sr = fma(r * z, ps, r);
    double pc = 1.0 / 20922789888000.0;                  //  1/16!
    pc = fma(pc, z, -1.0 / 87178291200.0);               // -1/14!
    pc = fma(pc, z, 1.0 / 479001600.0);                  //  1/12!
    pc = fma(pc, z, -1.0 / 3628800.0);                   // -1/10!
    pc = fma(pc, z, 1.0 / 40320.0);                      //  1/8!
    pc = fma(pc, z, -1.0 / 720.0);                       // -1/6!
    pc = fma(pc, z, 1.0 / 24.0);                         //  1/4!
    double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    double ss = (n & 1) ? cr : sr;
    double cc = (n & 1) ? sr : cr;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}

// Table-driven variant for the kernels that have LDS to spare (the fused fixed point at the two-waves-per-SIMD shape):
// x = n pi/32 + r, |r| <= pi/64, (cos, sin)(n pi/32) from a 64-entry table of correctly rounded values (raftx_tables.h,
// staged in LDS), short polynomials for sin r and cos r - 1 (truncation < 2^-56), angle addition with the table value
// as the addend.  23 VALU instructions instead of 45; same accuracy (tests/test_hip_parity.py).
__device__ __forceinline__ void tab_sincos(double x, ldptr tab, double &s, double &c) {
    const double fn = rint(x * RAFTX_32OPI);
    double r = fma(-fn, RAFTX_PIO32_1, x);
    r = fma(-fn, RAFTX_PIO32_2, r);
    r = fma(-fn, RAFTX_PIO32_3, r);
    const int n = (int)fn;
    ldptr e = tab + ((n & (RAFTX_SC_N - 1)) << 1);
    const double c0 = e[0], s0 = e[1];
    const double z = r * r;
    double ps = fma(z, 1.0 / 362880.0, -1.0 / 5040.0);
    ps = fma(ps, z, 1.0 / 120.0);
    ps = fma(ps, z, -1.0 / 6.0);
    const double sr = fma(r * z, ps, r);                 // sin r
    double pc = fma(z, 1.0 / 40320.0, -1.0 / 720.0);
    pc = fma(pc, z, 1.0 / 24.0);
    pc = fma(pc, z, -0.5);
    const double cm = pc * z;                            // cos r - 1
    c = fma(c0, cm, fma(-s0, sr, c0));
    s = fma(s0, cm, fma(c0, sr, s0));
}
__device__ __forceinline__ void stage_sincos_table(ldptr tab) {
    for (int i = threadIdx.x; i < 2 * RAFTX_SC_N; i += blockDim.x) tab[i] = RAFTX_SC_TAB[i];
}

__device__ __forceinline__ double fast_exp(double x) {
    const double log2e = 1.44269504088896338700e+00;
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    x = fmin(fmax(x, -740.0), 700.0);
    double fn = rint(x * log2e);
    double r = fma(-fn, ln2_hi, x);
    r = fma(-fn, ln2_lo, r);
    double p = 1.0 / 6227020800.0;                       // 1/13!
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)fn);
}

#pragma clang fp contract(fast)
// ------------------------------------------------------------------ device tables
// Device strip record (doubles), built at upload from the 32-double ABI record.
#define DS_N 32
#define DS_MCF 2     // -1 or row of the complex Cm table
#define DS_X 4       // x, y, z (absolute)
#define DS_U 7       // unit step vector of the run (unit * q)
#define DS_A 10      // arm about the reduced-DOF reference point
#define DS_Q 13
#define DS_P1 16
#define DS_P2 19
#define DS_IQ 22     // Iq, Ip1, Ip2, a_i, rhoV
#define DS_DQ 27     // dq, dp1, dp2, dend
#define DSI_M 3
#define DSI_CIRC 4
#define DSI_AXAL 8       // upright cross-section: p1 = (0, 0, +-1), p2 horizontal, up to rounding dust (< 1e-15)
#define DSI_VAX 16       // vertical axis: q = (0, 0, +-1), p1 and p2 horizontal, up to rounding dust (< 1e-15)

struct DevTables {
    int nDesign;
    const int64_t *__restrict__ off;     // [nDesign+1]
    const double *__restrict__ ds;       // [nStrips,DS_N] device strip records
    const int *__restrict__ dsi;         // [nStrips] flags: bits 0-1 rotor steps from the previous strip (0 = run
                                         // start, evaluated exactly; 1, 2 = unit steps), bit 2 circular section
    const double *__restrict__ M0, *__restrict__ B0, *__restrict__ C0;   // [nDesign,36]
    const double *__restrict__ MBw;      // [nDesign,2,36,nw] or null
    const int64_t *__restrict__ cmoff;   // [nDesign+1] or null
    const cplx *__restrict__ cm;         // [nRows,2,nw] or null
    int nCase, nHead, nw;
    const double *__restrict__ w, *__restrict__ k;    // [nw]
    const double *__restrict__ csh, *__restrict__ cch;   // per-bin 1/(1 - e^{-2kh}), 1/(1 + e^{-2kh}) (host, libm);
                                                         // 1 in the deep-water and k == 0 branches (helpers.py:211-222)
    const double *__restrict__ zeta;     // [nCase,nHead,nw]
    const double *__restrict__ beta;     // [nCase,nHead]
    double depth, rho, g;
};

// XCD-aware pair mapping: workgroups are dealt round-robin to the 8 XCDs, so give each
// XCD a contiguous slab of pairs (the sea states of one design then share one L2).
__device__ __forceinline__ int pair_of_block(int b, int npair) {
    const int per = (npair + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}
static inline unsigned grid_for_pairs(size_t npair) { return (unsigned)(((npair + 7) / 8) * 8); }

// ------------------------------------------------------------------ LDS layout
#ifndef RAFTX_RUN_LOOPS
#define RAFTX_RUN_LOOPS 1    // sweeps iterate over runs with run-type-specialised inner loops (0: one flag-switched loop over strips)
#endif
// ... for the shapes with up to two bins per lane; the larger ones (more than 1024 bins) keep the single loop, whose
// register footprint is smaller there
template <int NB>
constexpr bool RUN_LOOPS = RAFTX_RUN_LOOPS && NB <= 2;
#define RA_N 18              // doubles per staged record: arm, q, p1, p2 (pass A) + x, y, z, unit step (run starts)
#define TR_ROWS 6            // rows of a reduction tile (pass A: 2 strips x 3 sums per batch)
#define TR_STRIDE 72         // doubles per row (8 segments of 9: conflict-free)
#define SB 2                 // strips per reduction batch of pass A (3 rows each)

// Per-pair LDS (bytes at S = 53, nw = 200, 2 waves): xl 19.2 K + uv 5.1 K + vsq 2.5 K + tiles 8.7 K
// + 1.5 K  ~= 37 K  ->  four pairs per CU (160 KiB).  The one-wave-per-SIMD shapes additionally
// stage the hot strip constants (ra, +7.6 K).
struct Lds {
    ldptr sct;     // [RAFTX_SC_N][2] (cos, sin) table of tab_sincos (shapes with a run-start cache)
    ldptr rc;      // [rc_n][nw_rc][2] run-start cache: rc_n slots of two doubles per bin (see Kin), rc_stride doubles apart
    int rc_n, rc_stride;
    ldptr xl;      // [12][nxl]    XiLast (re/im rows), nxl = nw rounded up to even
    ldptr ra;      // [S][stage_n] hot strip constants: all RA_N, the first 6 (arm, q), or none -- see stage_policy
    ldptr uv;      // [S][12]      linearised drag vectors of the current heading
    ldptr vsq;     // [S][3] (+ [NWV-1][S][3] when NWV > 5)  sums over wave 0's bins of |v_q|^2, |v_p1|^2 (|v_perp|^2),
                     //              |v_p2|^2; overwritten by the live coefficients b_c (strip_phase).  The other waves' sums
                     //              lie in uv (dead between pass B and the strip phase): wave i at uv[s][3 (i-1) ..], read
                     //              by the lane of strip s before it writes that record -- see vsq_of
    ldptr tile;    // [NWV][TR_ROWS][TR_STRIDE]
    ldptr park;    // start of the span (vsq rows >= 1 | uv | tile) the solve phase may reuse, see park_policy
    ldptr bdw;     // [NWV][24]    per-wave partials of the 21 unique B_drag entries
    ldptr Bd;      // [36]
    ldptr mat;     // [108]        M0, B0, C0
    ldptr Bb;      // [36]         B0 + B_drag of the live linearisation (what the assembly of Z adds)
    liptr fl;         // [S]          strip flags (STAGE shapes only)
    int nxl;
};
static __host__ __device__ inline int xl_row(int nw) { return (nw + 1) & ~1; }
// rows of dedicated storage for the per-wave velocity sums: up to five waves, only wave 0's (the others lie in uv)
static __host__ __device__ inline int vsq_rows(int nwv) { return nwv <= 5 ? 1 : nwv; }
// park_n: doubles of the [uv | vsq | tile] span that the solve phase reuses as a per-lane parking column (0 = none);
// the span is padded up to that size for designs with few strips
static __host__ __device__ constexpr int park_policy(int nb, int maxt) { return (nb == 2 && maxt == 128) ? 12 * 128 : 0; }
// nw_xl: bins of the XiLast rows kept in LDS (0: XiLast lives in a global slab); rc_n slots of the run-start cache over
// nw_rc bins
__device__ __forceinline__ Lds carve(double *base_, int S, int nw_xl, int nwv, int stage_n, int park_n = 0, int rc_n = 0,
                                     int nw_rc = 0) {
    Lds l;
    ldptr base = (ldptr)base_;
    l.sct = base;                                 // first: 16-byte aligned entries (ds_read_b128 / ds_write_b128)
    if (nw_rc) base += 2 * RAFTX_SC_N;
    l.rc = base;
    l.rc_n = rc_n;
    l.rc_stride = 2 * xl_row(nw_rc);
    base += (size_t)rc_n * l.rc_stride;
    l.nxl = xl_row(nw_xl);
    l.xl = base;
    l.ra = l.xl + (size_t)12 * l.nxl;
    l.vsq = l.ra + (size_t)S * stage_n;
    l.uv = l.vsq + (size_t)vsq_rows(nwv) * S * 3;
    l.tile = l.uv + (size_t)S * 12;
    l.bdw = l.tile + (size_t)nwv * TR_ROWS * TR_STRIDE;
    l.park = l.vsq + (size_t)S * 3;        // row 0 of vsq keeps the drag coefficients b_c for the other headings
    if (park_n && l.bdw < l.park + park_n) l.bdw = l.park + park_n;
    l.Bd = l.bdw + (size_t)nwv * 24;
    l.mat = l.Bd + 36;
    l.Bb = l.mat + 108;
    l.fl = (liptr)(l.Bb + 36);
    return l;
}
static size_t lds_bytes(int S, int nw, int nwv, int stage_n, int park_n = 0, int rc_n = 0, int nw_rc = 0) {
    size_t span = (size_t)S * (12 + 3 * vsq_rows(nwv)) + (size_t)nwv * TR_ROWS * TR_STRIDE;          // vsq | uv | tile
    if (park_n && span < (size_t)S * 3 + (size_t)park_n) span = (size_t)S * 3 + (size_t)park_n;
    return sizeof(double) * ((size_t)(nw_rc ? 2 * RAFTX_SC_N : 0) + (size_t)rc_n * 2 * xl_row(nw_rc) + (size_t)12 * xl_row(nw) + (size_t)S * stage_n + span +
                             (size_t)nwv * 24 + 36 + 108 + 36 + 2) +
           sizeof(int) * (size_t)(stage_n == RA_N ? S + 2 : 2);
}
// where wave wv's velocity sums of strip s start, as base + s * stride
__device__ __forceinline__ ldptr vsq_of(const Lds &l, int wv, int nwv, int S, int &stride) {
    if (wv == 0 || nwv > 5) {
        stride = 3;
        return l.vsq + wv * S * 3;
    }
    stride = 12;
    return l.uv + (wv - 1) * 3;
}
// shapes that keep XiLast in global scratch (SolveArgs::Xl, a region per RUNNING workgroup: xl_slot_acquire) instead of LDS: the largest ones (no room), and the
// two-waves-per-SIMD 200-bin shape, whose LDS goes to the run-start cache instead (XiLast is touched twice per
// iteration, the run starts fourteen times)
#ifndef RAFTX_XL_LDS
#define RAFTX_XL_LDS 0       // tuning build (profiles/r06_xilast_ab.json): 1 = the 200-bin shape keeps XiLast in LDS too (three pairs per CU)
#endif
static __host__ __device__ constexpr bool xl_global(int nb, int maxt) { return (maxt == 512 && nb >= 3) || (!RAFTX_XL_LDS && maxt == 128 && nb == 2); }

// LDS traffic between lanes of ONE wave needs no s_barrier (a wave's LDS instructions
// execute in order); it only needs the compiler to keep the order.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void wg_sync(bool multi) {
    if (multi) __syncthreads();
    else wave_lds_fence();
}

// DPP moves on doubles (pure VALU, no LDS crossbar): quad_perm butterflies and row_shl:4
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Reduction tile of one wave: TR_ROWS (<= 8) rows; lane L writes its value of row r at
// tile[r*TR_STRIDE + tile_pos(L)].  tile_reduce returns, on lanes with (L & 7) == 0, the sum
// over the 64 lanes of row L >> 3 (other lanes: partial sums).  Reader lane (row, seg) adds
// the 8 entries of its segment, then three DPP steps fold the 8 segments.
__device__ __forceinline__ int tile_pos(int lane) { return (lane >> 3) * 9 + (lane & 7); }
__device__ __forceinline__ double tile_reduce(ldptr tile, int lane) {
    ldptr p = tile + min(lane >> 3, TR_ROWS - 1) * TR_STRIDE + (lane & 7) * 9;     // lanes 48..63 redo row 5 (unused)
    double a0 = p[0] + p[4], a1 = p[1] + p[5], a2 = p[2] + p[6], a3 = p[3] + p[7];
    double a = (a0 + a1) + (a2 + a3);
    a += dpp_mov<0xB1>(a);      // quad_perm [1,0,3,2]
    a += dpp_mov<0x4E>(a);      // quad_perm [2,3,0,1]
    a += dpp_mov<0x104>(a);     // row_shl:4
    return a;
}

// A wave-uniform double computed with vector instructions (e.g. cos of the heading) moved into
// SGPRs: frees two VGPRs for the whole kernel and feeds v_fma_f64 as its scalar operand.
__device__ __forceinline__ double to_sgpr(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Hides a value's provenance from the optimiser: address arithmetic based on it is redone at
// the use instead of being hoisted out of the fixed-point loop and spilled.
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// ------------------------------------------------------------------ per-bin data
template <int NB>
struct Bins {
    double k[NB];
    double depth;         // water depth (uniform)
    double w[NB];         // angular frequency (0 for inactive bins)
    double c1[NB];        // w * zeta0 * csh of the current heading (0 for inactive bins)
    bool act[NB];
    int iw[NB];           // clamped bin index (valid address even when inactive)
};

template <int NB>
__device__ __forceinline__ void load_bins(const DevTables &T, Bins<NB> &b, int tid) {
    b.depth = T.depth;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const int i = j * blockDim.x + tid;
        b.act[j] = i < T.nw;
        b.iw[j] = b.act[j] ? i : 0;
        const double kk = T.k[b.iw[j]], ww = T.w[b.iw[j]];       // unconditional loads, then selects
        b.k[j] = b.act[j] ? kk : 0.0;
        b.w[j] = b.act[j] ? ww : 0.0;
        b.c1[j] = 0.0;
    }
}
template <int NB>
__device__ __forceinline__ void set_heading_amp(const DevTables &T, Bins<NB> &b, int ic, int ih) {
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const double z0 = T.zeta[((size_t)ic * T.nHead + ih) * T.nw + b.iw[j]];
        b.c1[j] = b.w[j] * z0 * T.csh[b.iw[j]];
    }
}

// depth regime of a bin, derived from k exactly as the host derives DevTables::mode
// (helpers.py:211-218): 2 = k == 0, 1 = deep water (k h > 89.4), 0 = finite depth
__device__ __forceinline__ int depth_mode(double k, double depth) { return k == 0.0 ? 2 : (k * depth > 89.4 ? 1 : 0); }

// RAFTX_UNIT_ROTORS: keep the rotors of ONE unit step only and apply them twice for a two-unit step -- no more
// instructions than choosing between two rotor sets per step with selects, and 16 VGPRs less at two bins per lane
#ifndef RAFTX_UNIT_ROTORS
#define RAFTX_UNIT_ROTORS 1
#endif
// ------------------------------------------------------------------ wave kinematics along a run
// State per bin: a = amp * e^{-i k xi_s};  P = e^{k z_s};  Q = e^{-k (z_s + 2h)}  (Q = 0 in the
// deep-water branch, helpers.py:215-218, unless KEEPQ);  rotors of one and of two unit steps.
template <int NB>
struct Kin {
    double ar[NB], ai[NB], P[NB], Q[NB];
    double r1r[NB], r1i[NB], r1p[NB], r1q[NB];
#if !RAFTX_UNIT_ROTORS
    double r2r[NB], r2i[NB], r2p[NB], r2q[NB];
#endif
    // memo of the previous run start (wave-uniform keys, per-bin values): members that start at
    // the same depth share P, Q; members with the same step vector share the rotors
    double P0[NB], Q0[NB];
    double mz, mux, muy, muz;
    bool rot, dec;        // of the current step vector (wave-uniform): the phase rotor / the depth-decay rotors differ from 1
    bool vert;            // the step vector has no horizontal part (a vertical member, or a run of one strip)
    // run-start cache cursor: every sweep of a pair meets the same run starts in the same order with the same memo
    // hits, so the n-th transcendental evaluation of a sweep is the same quantity in every sweep -- the first sweep
    // (the inertial excitation) stores the first cn of them per bin in LDS, the later sweeps read them back
    ldptr rc, sct;
    int ce, cn, cstride;
};
struct RunCache {
    ldptr p;
    int n, stride;
    ldptr sct;
};
__device__ __forceinline__ RunCache run_cache_of(const Lds &l) { return {l.rc, l.rc_n, l.rc_stride, l.sct}; }

// Run start: exact evaluation from the absolute position (helpers.py:201,211-222).
//   amp[j] multiplies the phase factor (c1 for the velocity sweeps, 1 for the pressure sweep).
struct RunStart {
    double x, y, z, ux, uy, uz;
};
__device__ __forceinline__ RunStart run_start_of(cdptr rec) {
    return {rec[DS_X], rec[DS_X + 1], rec[DS_X + 2], rec[DS_U], rec[DS_U + 1], rec[DS_U + 2]};
}
__device__ __forceinline__ RunStart run_start_of(ldptr r) {              // staged LDS record
    return {r[12], r[13], r[14], r[15], r[16], r[17]};
}
#pragma clang fp contract(off)          // (see fast_sincos)
// One cacheable evaluation: two doubles per bin in slot `slot` of the run-start cache.  CM 1 (first sweep of a pair):
// a fresh quantity is computed and stored while slots last; CM 2: read back what was stored, compute the rest.  A
// quantity met again in the same sweep (fresh == false: a later run with the same step vector) is read back in both
// modes.  Every lane reads only what it wrote itself.
template <int CM, int NB, typename Fn>
__device__ __forceinline__ void kin_cached(Kin<NB> &K, const Bins<NB> &b, int slot, bool fresh, double (&v0)[NB],
                                           double (&v1)[NB], Fn compute) {
    static_assert(CM == 1 || CM == 2, "cache mode");
    const bool hit = slot < K.cn;                             // wave-uniform
    ldptr sl = K.rc + slot * K.cstride;
    if (hit && !(CM == 1 && fresh)) {
#pragma unroll
        for (int j = 0; j < NB; j++) {
            ldptr e = sl + 2 * b.iw[j];
            v0[j] = e[0];
            v1[j] = e[1];
        }
    } else {
#pragma unroll
        for (int j = 0; j < NB; j++) compute(j, v0[j], v1[j]);
        if (CM == 1 && hit) {
#pragma unroll
            for (int j = 0; j < NB; j++)
                if (b.act[j]) {
                    ldptr e = sl + 2 * b.iw[j];
                    e[0] = v0[j];
                    e[1] = v1[j];
                }
        }
    }
}
// CM 0: no cache.  The rotors and P, Q of the previous run start stay in registers for members with the same step
// vector / start depth in every mode.
template <int NB, bool KEEPQ, int CM = 0>
__device__ __forceinline__ void kin_start(Kin<NB> &K, const RunStart rs, const Bins<NB> &b,
                                          const double (&amp)[NB], double cb, double sb) {
    const double x = rs.x, y = rs.y, z = rs.z, ux = rs.ux, uy = rs.uy, uz = rs.uz;
    const double xi = cb * x + sb * y;
    const double du = cb * ux + sb * uy;
    const bool same_z = (z == K.mz);                      // wave-uniform
    const bool same_u = (ux == K.mux) && (uy == K.muy) && (uz == K.muz);
    auto f_phasor = [&](int j, double &c_, double &s_) {
        if constexpr (CM != 0) tab_sincos(-(b.k[j] * xi), K.sct, s_, c_);
        else fast_sincos(-(b.k[j] * xi), s_, c_);
    };
    auto f_pq = [&](int j, double &P_, double &Q_) {
        const double kz = b.k[j] * z;
        const double Pe = fast_exp(kz);
        const double Qe = fast_exp(-(b.k[j] * (z + 2.0 * b.depth)));       // e^{-k (z + 2h)}
        // k == 0 (helpers.py:211-214): Sh = 1, Ch = Cc = 99999  <=>  P + Q = 99999, P - Q = 1 with csh = cch = 1
        const bool k0 = b.k[j] == 0.0;
        P_ = k0 ? 50000.0 : Pe;
        Q_ = k0 ? 49999.0 : Qe;
    };
    auto f_rot = [&](int j, double &c_, double &s_) {
        if constexpr (CM != 0) tab_sincos(-(b.k[j] * du), K.sct, s_, c_);
        else fast_sincos(-(b.k[j] * du), s_, c_);
    };
    auto f_dec = [&](int j, double &p_, double &q_) {
        p_ = fast_exp(b.k[j] * uz);
        q_ = fast_exp(-(b.k[j] * uz));
    };
    {
        double c[NB], s[NB];
        if constexpr (CM == 0) {
#pragma unroll
            for (int j = 0; j < NB; j++) f_phasor(j, c[j], s[j]);
        } else {
            kin_cached<CM>(K, b, K.ce++, true, c, s, f_phasor);
        }
#pragma unroll
        for (int j = 0; j < NB; j++) {
            K.ar[j] = amp[j] * c[j];
            K.ai[j] = amp[j] * s[j];
        }
    }
    if (!same_z) {
        double P[NB], Q[NB];                              // cached in the KEEPQ form: the reader masks Q of the deep-water branch
        if constexpr (CM == 0) {
#pragma unroll
            for (int j = 0; j < NB; j++) f_pq(j, P[j], Q[j]);
        } else {
            kin_cached<CM>(K, b, K.ce++, true, P, Q, f_pq);
        }
#pragma unroll
        for (int j = 0; j < NB; j++) {
            K.P0[j] = P[j];
            K.Q0[j] = (!KEEPQ && depth_mode(b.k[j], b.depth) == 1) ? 0.0 : Q[j];
        }
    }
#pragma unroll
    for (int j = 0; j < NB; j++) {
        K.P[j] = K.P0[j];
        K.Q[j] = K.Q0[j];
    }
    K.mz = z;
    if (!same_u) {
        K.rot = du != 0.0;                            // wave-uniform: vertical members skip the phase rotor,
        K.dec = uz != 0.0;                            // horizontal ones the depth-decay rotors
        K.vert = (ux == 0.0) && (uy == 0.0);
        K.mux = ux;
        K.muy = uy;
        K.muz = uz;
        double c[NB], s[NB], pp[NB], qq[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) {
            c[j] = 1.0; s[j] = 0.0; pp[j] = 1.0; qq[j] = 1.0;
        }
        if constexpr (CM == 0) {
#pragma unroll
            for (int j = 0; j < NB; j++) {
                if (K.rot) f_rot(j, c[j], s[j]);
                if (K.dec) f_dec(j, pp[j], qq[j]);
            }
        } else {
            if (K.rot) kin_cached<CM>(K, b, K.ce++, true, c, s, f_rot);
            if (K.dec) kin_cached<CM>(K, b, K.ce++, true, pp, qq, f_dec);
        }
#pragma unroll
        for (int j = 0; j < NB; j++) {
            K.r1r[j] = c[j];
            K.r1i[j] = s[j];
            K.r1p[j] = pp[j];
            K.r1q[j] = qq[j];
#if !RAFTX_UNIT_ROTORS
            K.r2r[j] = c[j] * c[j] - s[j] * s[j];
            K.r2i[j] = 2.0 * c[j] * s[j];
            K.r2p[j] = pp[j] * pp[j];
            K.r2q[j] = qq[j] * qq[j];
#endif
        }
    }
}

#pragma clang fp contract(fast)
template <int NB>
__device__ __forceinline__ void kin_step1(Kin<NB> &K) {
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const double t = K.ar[j] * K.r1r[j] - K.ai[j] * K.r1i[j];
        K.ai[j] = K.ar[j] * K.r1i[j] + K.ai[j] * K.r1r[j];
        K.ar[j] = t;
        K.P[j] *= K.r1p[j];
        K.Q[j] *= K.r1q[j];
    }
}
template <int NB>
__device__ __forceinline__ void kin_step2(Kin<NB> &K) {
#if RAFTX_UNIT_ROTORS
    kin_step1(K);
    kin_step1(K);
#else
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const double t = K.ar[j] * K.r2r[j] - K.ai[j] * K.r2i[j];
        K.ai[j] = K.ar[j] * K.r2i[j] + K.ai[j] * K.r2r[j];
        K.ar[j] = t;
        K.P[j] *= K.r2p[j];
        K.Q[j] *= K.r2q[j];
    }
#endif
}
// advance to the strip described by (fl, rec) -- wave-uniform control flow; rec is either the
// global device record (scalar loads) or the staged LDS record
template <int NB, bool KEEPQ, int CM = 0, typename RecPtr>
__device__ __forceinline__ void kin_advance(Kin<NB> &K, int fl, RecPtr rec, const Bins<NB> &b,
                                            const double (&amp)[NB], double cb, double sb) {
    const int m = fl & DSI_M;
    if (m == 0) {
        kin_start<NB, KEEPQ, CM>(K, run_start_of(rec), b, amp, cb, sb);
    } else if (m == 1) {
        kin_step1(K);
    } else {
        kin_step2(K);
    }
}
template <int NB>
__device__ __forceinline__ void kin_reset(Kin<NB> &K) {
#pragma unroll
    for (int j = 0; j < NB; j++) {
        K.ar[j] = 0.0; K.ai[j] = 0.0; K.P[j] = 1.0; K.Q[j] = 0.0;
        K.r1r[j] = K.r1p[j] = K.r1q[j] = 1.0;
        K.r1i[j] = 0.0;
#if !RAFTX_UNIT_ROTORS
        K.r2r[j] = K.r2p[j] = K.r2q[j] = 1.0;
        K.r2i[j] = 0.0;
#endif
        K.P0[j] = 1.0;
        K.Q0[j] = 0.0;
    }
    K.mz = K.mux = K.muy = K.muz = __builtin_nan("");      // never equal: the first run start computes everything
    K.rot = K.dec = false;
    K.vert = true;
    K.rc = nullptr;
    K.sct = nullptr;
    K.ce = K.cn = K.cstride = 0;
}
template <int NB>
__device__ __forceinline__ void kin_reset(Kin<NB> &K, const RunCache &rc) {
    kin_reset(K);
    K.rc = rc.p;
    K.cn = rc.n;
    K.cstride = rc.stride;
    K.sct = rc.sct;
}

// ------------------------------------------------------------------ strip sweeps
#ifdef RAFTX_PHASE_TIMING
#define PT_DECL unsigned long long pt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_t = __builtin_readcyclecounter(), pt_c0 = pt_t, pt_w0 = wall_clock64()
// clock trace (timing builds): dbg[8] = earliest workgroup start of the launch (100 MHz ticks), dbg[10 + 2 b], dbg[11 + 2 b] =
// shader cycles / 100 MHz ticks summed over the workgroups that STARTED in the b-th 125 us of the launch: cycles / ticks x 100 MHz
// is the shader clock those workgroups lived at
#define PT_CLOCK_BUCKETS 64
#define PT_BEGIN(A)                                                                    \
    do {                                                                               \
        if ((A).dbg && threadIdx.x == 0) atomicMin((A).dbg + 8, pt_w0);                \
    } while (0)
#define PT_MARK(i)                                              \
    do {                                                        \
        unsigned long long n_ = __builtin_readcyclecounter();   \
        pt_[i] += n_ - pt_t;                                    \
        pt_t = n_;                                              \
    } while (0)
#define PT_FLUSH(A)                                                                    \
    do {                                                                               \
        if ((A).dbg && threadIdx.x == 0) {                                             \
            for (int i_ = 0; i_ < 8; i_++) atomicAdd((A).dbg + i_, pt_[i_]);          \
            const unsigned long long c1_ = __builtin_readcyclecounter(), w1_ = wall_clock64();                       \
            const unsigned long long t0_ = atomicAdd((A).dbg + 8, 0ull);              \
            unsigned long long b_ = (pt_w0 - t0_) / 12500ull;                          \
            if (b_ >= PT_CLOCK_BUCKETS) b_ = PT_CLOCK_BUCKETS - 1;                     \
            atomicAdd((A).dbg + 10 + 2 * b_, c1_ - pt_c0);                             \
            atomicAdd((A).dbg + 11 + 2 * b_, w1_ - pt_w0);                             \
        }                                                                              \
    } while (0)
#define PT_ARG , unsigned long long *pt_, unsigned long long &pt_t
#define PT_PASS , pt_, pt_t
#else
#define PT_ARG
#define PT_PASS
#define PT_DECL
#define PT_BEGIN(A)
#define PT_MARK(i)
#define PT_FLUSH(A)
#endif

// Inertial excitation of one heading (raft_member.py:1965-1991), ACCUMULATED into F:
//   f3 = Imat ud + pDyn a_i q,  ud = i w u,  F += [f3 ; a x f3]     (helpers.py:468-483)
// Imat = Iq qq^T + Ip1 p1p1^T + Ip2 p2p2^T (raft_member.py:1423-1448), or rhoV Cm(w) for
// MacCamy-Fuchs strips (complex, per bin; raft_member.py:1415-1420).
template <int NB, bool MCF, bool RC = false>
__device__ __forceinline__ void inertial_excitation(const DevTables &T, cdptr ds,
                                                    ciptr dsi, int S, const cplx *__restrict__ cm,
                                                    const Bins<NB> &b, int ic, int ih, double cb, double sb,
                                                    cplx (&F)[NB][6], const RunCache rcache = {nullptr, 0, 0, nullptr}) {
    double one[NB], w[NB], s1[NB], sp[NB], qm[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) {
        one[j] = 1.0;
        w[j] = b.w[j];
        s1[j] = b.c1[j];                                                   // w zeta0 csh
        const double z0r = T.zeta[((size_t)ic * T.nHead + ih) * T.nw + b.iw[j]];
        const double z0 = b.act[j] ? z0r : 0.0;
        sp[j] = T.rho * T.g * z0 * T.cch[b.iw[j]];                         // rho g zeta0 / cosh kh scaling (helpers.py:231)
        qm[j] = (depth_mode(b.k[j], b.depth) == 1) ? 0.0 : 1.0;                              // deep water: Sh = Ch = e^{kz}
    }
    Kin<NB> K;
    kin_reset(K, rcache);                          // this sweep fills the run-start cache (if any)
    if (S <= 0) return;
    int fn = dsi[0];
#pragma unroll 1
    for (int s = 0; s < S; s++) {
        cdptr rec = ds + (size_t)s * DS_N;
        // fetch the whole record (one burst of scalar loads) before the branchy kinematics update
        const double ax = rec[DS_A], ay = rec[DS_A + 1], az = rec[DS_A + 2];
        const double ai_ = rec[DS_IQ + 3], rhoV = rec[DS_IQ + 4];
        const int mcf = MCF ? (int)rec[DS_MCF] : -1;
        double al[3], ga[3], I[3], n[3][3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            n[c][0] = rec[DS_Q + 3 * c];
            n[c][1] = rec[DS_Q + 3 * c + 1];
            n[c][2] = rec[DS_Q + 3 * c + 2];
            I[c] = rec[DS_IQ + c];
        }
        const int fl = fn;
        fn = dsi[min(s + 1, S - 1)];
        kin_advance<NB, true, RC ? 1 : 0>(K, fl, rec, b, one, cb, sb);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            al[c] = n[c][0] * cb + n[c][1] * sb;
            ga[c] = n[c][2];
        }
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const double pq = K.P[j] + K.Q[j];
            const double Qk = qm[j] * K.Q[j];
            const double hs = s1[j] * (K.P[j] + Qk), hd = s1[j] * (K.P[j] - Qk);
            const cplx t1 = {hs * K.ar[j], hs * K.ai[j]};
            const cplx t2 = {-hd * K.ai[j], hd * K.ar[j]};
            const double pp = sp[j] * pq;
            const cplx pd = {pp * K.ar[j], pp * K.ai[j]};                  // rho g zeta_s Cc
            cplx f3[3] = {{0, 0}, {0, 0}, {0, 0}};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const cplx G = {al[c] * t1.re + ga[c] * t2.re, al[c] * t1.im + ga[c] * t2.im};   // n_c . u
                const cplx a = {-w[j] * G.im, w[j] * G.re};                                       // n_c . ud
                cplx g;
                if (c == 0) {
                    g = {I[0] * a.re + pd.re * ai_, I[0] * a.im + pd.im * ai_};
                } else if (MCF && mcf >= 0) {
                    const cplx m = cm[((size_t)mcf * 2 + (c - 1)) * T.nw + b.iw[j]];
                    g = cmul(cscale(m, rhoV), a);
                } else {
                    g = cscale(a, I[c]);
                }
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    f3[q].re = fma(g.re, n[c][q], f3[q].re);
                    f3[q].im = fma(g.im, n[c][q], f3[q].im);
                }
            }
            F[j][0] = cadd(F[j][0], f3[0]);
            F[j][1] = cadd(F[j][1], f3[1]);
            F[j][2] = cadd(F[j][2], f3[2]);
            F[j][3].re += ay * f3[2].re - az * f3[1].re;
            F[j][3].im += ay * f3[2].im - az * f3[1].im;
            F[j][4].re += az * f3[0].re - ax * f3[2].re;
            F[j][4].im += az * f3[0].im - ax * f3[2].im;
            F[j][5].re += ax * f3[1].re - ay * f3[0].re;
            F[j][5].im += ax * f3[1].im - ay * f3[0].im;
        }
    }
}

// How many doubles of each strip record a shape stages in LDS (host and device must agree):
//   RA_N (all 18)  one-wave-per-SIMD shapes: nothing else hides latency there
//   6 (arm, q)     the 2-wave shape: what every strip of pass A needs first; p1/p2 (rectangular strips only) and the
//                  run-start fields keep coming through scalar loads, whose latency the kinematics update covers.
//                  Four pairs per CU leave no LDS for more.
//   0              larger shapes: scalar loads only
#ifndef RAFTX_STAGE256
#define RAFTX_STAGE256 0
#endif
#ifndef RAFTX_STAGE128
#define RAFTX_STAGE128 6     // (0, all through scalar loads, measured at the round-3 kernel: 3.63 ms against 3.33)
#endif
static __host__ __device__ constexpr int stage_policy(int nb, int maxt) {
    return maxt == 64 ? RA_N : (maxt == 128 ? RAFTX_STAGE128 : ((RAFTX_STAGE256 && maxt == 256 && nb == 1) ? RA_N : 0));
}

// Hot per-strip constants of pass A.  Two sources (template STAGE):
//  * one-wave-per-SIMD shapes stage them once per workgroup into LDS and read them back as
//    wave-wide broadcasts (~100 cycles, nothing else hides latency there);
//  * the two-waves-per-SIMD shapes read them with scalar loads straight into SGPRs (no LDS
//    space, no VGPRs; the sibling wave covers the constant-cache miss latency).
struct RecA {
    double ax, ay, az, qx, qy, qz, p1x, p1y, p1z, p2x, p2y, p2z;
};
__device__ __forceinline__ void stage_recA(cdptr ds, ciptr dsi, int S, const Lds &l, int n) {
    for (int i = threadIdx.x; i < S * n; i += blockDim.x) {
        const int s = i / n, f = i % n;
        // DS_A .. DS_P2+2 are contiguous (12), then DS_X .. DS_U+2 (6)
        l.ra[i] = ds[(size_t)s * DS_N + (f < 12 ? DS_A + f : DS_X + (f - 12))];
    }
    if (n == RA_N)
        for (int i = threadIdx.x; i < S; i += blockDim.x) l.fl[i] = dsi[i];
}
__device__ __forceinline__ RecA load_recA(ldptr r) {              // staged LDS record
    return {r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11]};
}
__device__ __forceinline__ RecA load_recA(cdptr rec) {             // global device record (scalar loads)
    return {rec[DS_A], rec[DS_A + 1], rec[DS_A + 2], rec[DS_Q], rec[DS_Q + 1], rec[DS_Q + 2],
            rec[DS_P1], rec[DS_P1 + 1], rec[DS_P1 + 2], rec[DS_P2], rec[DS_P2 + 1], rec[DS_P2 + 2]};
}
// Per-strip source of flags and records for the sweeps (STAGE = stage_policy value)
template <int STAGE>
struct StripSrc;
template <>
struct StripSrc<RA_N> {
    const Lds &l;
    int fnv;                                   // flag word of the next strip (VGPR, fetched one strip ahead)
    __device__ __forceinline__ StripSrc(const Lds &l_, cdptr, ciptr) : l(l_), fnv(l_.fl[0]) {}
    __device__ __forceinline__ ldptr rec(int s) const { return l.ra + s * RA_N; }
    __device__ __forceinline__ int flags(int s_next) {
        const int f = __builtin_amdgcn_readfirstlane(fnv);
        fnv = l.fl[s_next];
        return f;
    }
};
template <>
struct StripSrc<0> {
    cdptr ds;
    ciptr dsi;
    int fn;                                    // flag word of the next strip (SGPR, fetched one strip ahead)
    __device__ __forceinline__ StripSrc(const Lds &, cdptr ds_, ciptr dsi_) : ds(ds_), dsi(dsi_), fn(dsi_[0]) {}
    __device__ __forceinline__ cdptr rec(int s) const { return ds + (size_t)s * DS_N; }
    __device__ __forceinline__ int flags(int s_next) {
        const int f = fn;
        fn = dsi[s_next];
        return f;
    }
};
// arm and q from LDS, the rest from the global record
struct RecSplit {
    ldptr lr;
    cdptr gr;
};
__device__ __forceinline__ RecA load_recA(RecSplit r) {
    return {r.lr[0], r.lr[1], r.lr[2], r.lr[3], r.lr[4], r.lr[5],
            r.gr[DS_P1], r.gr[DS_P1 + 1], r.gr[DS_P1 + 2], r.gr[DS_P2], r.gr[DS_P2 + 1], r.gr[DS_P2 + 2]};
}
__device__ __forceinline__ RunStart run_start_of(RecSplit r) { return run_start_of(r.gr); }
template <>
struct StripSrc<6> {
    const Lds &l;
    cdptr ds;
    ciptr dsi;
    int fn;
    __device__ __forceinline__ StripSrc(const Lds &l_, cdptr ds_, ciptr dsi_) : l(l_), ds(ds_), dsi(dsi_), fn(dsi_[0]) {}
    __device__ __forceinline__ RecSplit rec(int s) const { return {l.ra + s * 6, ds + (size_t)s * DS_N}; }
    __device__ __forceinline__ int flags(int s_next) {
        const int f = fn;
        fn = dsi[s_next];
        return f;
    }
};

__device__ __forceinline__ void load_arm(cdptr rec, RecA &r) { r.ax = rec[DS_A]; r.ay = rec[DS_A + 1]; r.az = rec[DS_A + 2]; }
__device__ __forceinline__ void load_arm(ldptr p, RecA &r) { r.ax = p[0]; r.ay = p[1]; r.az = p[2]; }
__device__ __forceinline__ void load_arm(RecSplit p, RecA &r) { r.ax = p.lr[0]; r.ay = p.lr[1]; r.az = p.lr[2]; }
__device__ __forceinline__ double src_qz(cdptr rec) { return rec[DS_Q + 2]; }
__device__ __forceinline__ double src_qz(ldptr r) { return r[5]; }
__device__ __forceinline__ double src_qz(RecSplit r) { return r.lr[5]; }
// one strip of pass A at the kinematic state (ar, ai, ps = P + Q, pd = P - Q): the three sums over this lane's bins
// Body-velocity terms that do not change along a run (the strips of a run share the arm components across its axis):
//   RT 1 (vertical run: arm x, y fixed)   x: X0 - X5 ay,  y: X1 + X5 ax,  z: X2 + X3 ay - X4 ax   (all of z)
//   RT 2 (horizontal run: arm z fixed)    x: X0 + X4 az,  y: X1 - X3 az
// in the rotated form of pass A (re <- im, im <- -re of i w V).
template <int NB>
struct BodyHoist {
    double xr[NB], xi[NB], yr[NB], yi[NB], zr[NB], zi[NB];
};
template <int NB, int RT>
__device__ __forceinline__ void body_hoist(BodyHoist<NB> &H, const cplx (&X)[NB][6], const RecA &r0) {
    const double ax = r0.ax, ay = r0.ay, az = r0.az;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        if (RT == 3) {
            // Upright pontoon, in the member's own frame (q and p2 horizontal): with R(X) = -i X (the rotated form of pass A)
            //   v_q  = al_q t1 + R(q.(X0,X1) + az (qx X4 - qy X3) + (qy ax - qx ay) X5)
            //   v_p2 = al_2 t1 + R(p2.(X0,X1) + az (p2x X4 - p2y X3)) + (p2y ax - p2x ay) R(X5)
            // and (qy ax - qx ay) = -(arm x q)_z does not change along the run (the arm moves along q), so all of v_q's body
            // part is a run constant (xr, xi); v_p2 keeps one strip-dependent term, w2 R(X5); v_p1 = +-v_z.
            const double wq = r0.qy * ax - r0.qx * ay;
            const double cr = fma(wq, X[j][5].re, fma(az, r0.qx * X[j][4].re - r0.qy * X[j][3].re, fma(r0.qy, X[j][1].re, r0.qx * X[j][0].re)));
            const double ci = fma(wq, X[j][5].im, fma(az, r0.qx * X[j][4].im - r0.qy * X[j][3].im, fma(r0.qy, X[j][1].im, r0.qx * X[j][0].im)));
            H.xr[j] = ci;
            H.xi[j] = -cr;
            const double dr = fma(az, r0.p2x * X[j][4].re - r0.p2y * X[j][3].re, fma(r0.p2y, X[j][1].re, r0.p2x * X[j][0].re));
            const double di = fma(az, r0.p2x * X[j][4].im - r0.p2y * X[j][3].im, fma(r0.p2y, X[j][1].im, r0.p2x * X[j][0].im));
            H.yr[j] = di;
            H.yi[j] = -dr;
            continue;
        }
        if (RT == 1) {
            H.xr[j] = fma(-X[j][5].im, ay, X[j][0].im);
            H.xi[j] = fma(X[j][5].re, ay, -X[j][0].re);
            H.yr[j] = fma(X[j][5].im, ax, X[j][1].im);
            H.yi[j] = fma(-X[j][5].re, ax, -X[j][1].re);
            H.zr[j] = fma(-X[j][4].im, ax, fma(X[j][3].im, ay, X[j][2].im));
            H.zi[j] = fma(X[j][4].re, ax, fma(-X[j][3].re, ay, -X[j][2].re));
        } else if (RT == 2) {
            H.xr[j] = fma(X[j][4].im, az, X[j][0].im);
            H.xi[j] = fma(-X[j][4].re, az, -X[j][0].re);
            H.yr[j] = fma(-X[j][3].im, az, X[j][1].im);
            H.yi[j] = fma(X[j][3].re, az, -X[j][1].re);
        }
    }
}
template <int NB, int RT = 0>
__device__ __forceinline__ void passA_core(const double (&ar)[NB], const double (&ai)[NB], const double (&psv)[NB],
                                           const double (&pdv)[NB], const RecA &r, bool circ, double cb, double sb,
                                           const cplx (&X)[NB][6], const BodyHoist<NB> &H, double &v0, double &v1, double &v2) {
    v0 = 0.0; v1 = 0.0; v2 = 0.0;
    if constexpr (RT == 3) {            // upright pontoon in its own frame (see body_hoist)
        const double alq = r.qx * cb + r.qy * sb, al2 = r.p2x * cb + r.p2y * sb;     // wave-uniform
        const double w2 = r.p2y * r.ax - r.p2x * r.ay;
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const double ps = psv[j], pd = pdv[j];
            const double t1r = ar[j] * ps, t1i = ai[j] * ps, t2r = -ai[j] * pd, t2i = ar[j] * pd;
            const double vqr = fma(alq, t1r, H.xr[j]), vqi = fma(alq, t1i, H.xi[j]);
            const double v2r = fma(w2, X[j][5].im, fma(al2, t1r, H.yr[j])), v2i = fma(-w2, X[j][5].re, fma(al2, t1i, H.yi[j]));
            const double rzr = fma(-X[j][4].im, r.ax, fma(X[j][3].im, r.ay, t2r + X[j][2].im));
            const double rzi = fma(X[j][4].re, r.ax, fma(-X[j][3].re, r.ay, t2i - X[j][2].re));
            v0 = fma(vqr, vqr, fma(vqi, vqi, v0));
            v1 = fma(rzr, rzr, fma(rzi, rzi, v1));
            v2 = fma(v2r, v2r, fma(v2i, v2i, v2));
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const double ps = psv[j], pd = pdv[j];
        const double t1r = ar[j] * ps, t1i = ai[j] * ps, t2r = -ai[j] * pd, t2i = ar[j] * pd;
        double rxr, rxi, ryr, ryi, rzr, rzi;
        if (RT == 1) {
            rxr = fma(X[j][4].im, r.az, fma(cb, t1r, H.xr[j]));
            rxi = fma(-X[j][4].re, r.az, fma(cb, t1i, H.xi[j]));
            ryr = fma(-X[j][3].im, r.az, fma(sb, t1r, H.yr[j]));
            ryi = fma(X[j][3].re, r.az, fma(sb, t1i, H.yi[j]));
            rzr = t2r + H.zr[j];
            rzi = t2i + H.zi[j];
        } else if (RT == 2 || RT == 3) {
            rxr = fma(-X[j][5].im, r.ay, fma(cb, t1r, H.xr[j]));
            rxi = fma(X[j][5].re, r.ay, fma(cb, t1i, H.xi[j]));
            ryr = fma(X[j][5].im, r.ax, fma(sb, t1r, H.yr[j]));
            ryi = fma(-X[j][5].re, r.ax, fma(sb, t1i, H.yi[j]));
            rzr = fma(-X[j][4].im, r.ax, fma(X[j][3].im, r.ay, t2r + X[j][2].im));
            rzi = fma(X[j][4].re, r.ax, fma(-X[j][3].re, r.ay, t2i - X[j][2].re));
        } else {
            rxr = fma(cb, t1r, X[j][0].im); rxi = fma(cb, t1i, -X[j][0].re);
            ryr = fma(sb, t1r, X[j][1].im); ryi = fma(sb, t1i, -X[j][1].re);
            rzr = t2r + X[j][2].im; rzi = t2i - X[j][2].re;
            rxr = fma(X[j][4].im, r.az, rxr); rxr = fma(-X[j][5].im, r.ay, rxr);
            rxi = fma(-X[j][4].re, r.az, rxi); rxi = fma(X[j][5].re, r.ay, rxi);
            ryr = fma(X[j][5].im, r.ax, ryr); ryr = fma(-X[j][3].im, r.az, ryr);
            ryi = fma(-X[j][5].re, r.ax, ryi); ryi = fma(X[j][3].re, r.az, ryi);
            rzr = fma(X[j][3].im, r.ay, rzr); rzr = fma(-X[j][4].im, r.ax, rzr);
            rzi = fma(-X[j][3].re, r.ay, rzi); rzi = fma(X[j][4].re, r.ax, rzi);
        }
        // the axis of a run of two or more strips is its step direction: q = (0, 0, +-1) on a vertical run, q_z = 0 on a
        // horizontal one (the step vector is unit * q with unit > 0, so these are exact zeros and so are the dropped products)
        const double vqr = RT == 1 ? r.qz * rzr : ((RT == 2 || RT == 3) ? fma(r.qy, ryr, r.qx * rxr) : fma(r.qz, rzr, fma(r.qy, ryr, r.qx * rxr)));
        const double vqi = RT == 1 ? r.qz * rzi : ((RT == 2 || RT == 3) ? fma(r.qy, ryi, r.qx * rxi) : fma(r.qz, rzi, fma(r.qy, ryi, r.qx * rxi)));
        if (RT == 1 && circ) {      // q = (0, 0, +-1): v_q is the z component, v_perp the horizontal part
            v0 = fma(rzr, rzr, fma(rzi, rzi, v0));
            v1 = fma(rxr, rxr, fma(rxi, rxi, v1));
            v1 = fma(ryr, ryr, fma(ryi, ryi, v1));
        } else if (circ) {          // |v_perp|^2 = |v|^2 - |v_q|^2   (raft_member.py:2084-2087)
            const double q2 = fma(vqi, vqi, vqr * vqr);
            double n2 = fma(rxi, rxi, rxr * rxr);
            n2 = fma(ryr, ryr, n2); n2 = fma(ryi, ryi, n2);
            n2 = fma(rzr, rzr, n2); n2 = fma(rzi, rzi, n2);
            v0 += q2;
            v1 += n2 - q2;
        } else if (RT == 3) {       // horizontal member with an upright cross-section (pontoons): p1 = (0, 0, +-1), p2z = 0 up to
                                    // 1e-15 -- the dust products are left out of the squares (DSI_AXAL)
            const double v1r = r.p1z * rzr, v1i = r.p1z * rzi;
            const double v2r = fma(r.p2y, ryr, r.p2x * rxr), v2i = fma(r.p2y, ryi, r.p2x * rxi);
            v0 = fma(vqr, vqr, fma(vqi, vqi, v0));
            v1 = fma(v1r, v1r, fma(v1i, v1i, v1));
            v2 = fma(v2r, v2r, fma(v2i, v2i, v2));
        } else {
            const double v1r = fma(r.p1z, rzr, fma(r.p1y, ryr, r.p1x * rxr));
            const double v1i = fma(r.p1z, rzi, fma(r.p1y, ryi, r.p1x * rxi));
            const double v2r = fma(r.p2z, rzr, fma(r.p2y, ryr, r.p2x * rxr));
            const double v2i = fma(r.p2z, rzi, fma(r.p2y, ryi, r.p2x * rxi));
            v0 = fma(vqr, vqr, fma(vqi, vqi, v0));
            v1 = fma(v1r, v1r, fma(v1i, v1i, v1));
            v2 = fma(v2r, v2r, fma(v2i, v2i, v2));
        }
    }
}
// one strip of pass A: advances K, returns the three sums over this lane's bins
template <int NB, int CM, typename RecPtr>
__device__ __forceinline__ void passA_strip(Kin<NB> &K, const RecA &r, int fl, RecPtr rec, const Bins<NB> &b,
                                            double cb, double sb, const cplx (&X)[NB][6], double &v0, double &v1,
                                            double &v2) {
    kin_advance<NB, false, CM>(K, fl, rec, b, b.c1, cb, sb);
    double ps[NB], pd[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) {
        ps[j] = K.P[j] + K.Q[j];
        pd[j] = K.P[j] - K.Q[j];
    }
    BodyHoist<NB> H;      // unused by the general form
    passA_core<NB, 0>(K.ar, K.ai, ps, pd, r, (fl & DSI_CIRC) != 0, cb, sb, X, H, v0, v1, v2);
}

// Run-type-specialised step of the kinematic state (RT: 0 inclined, 1 vertical = no phase rotation, 2 horizontal = no
// depth decay); m = 1 or 2 unit steps (wave-uniform)
// The step count m (1 or 2 units) is wave-uniform: a BRANCH between the two rotor sets, not a select per value (the
// compiler if-converts a plain `m == 1 ? r1 : r2` into 2 v_cndmask per double and computes the stepped state into
// temporaries that are copied back -- 12 of 45 VALU instructions per strip of the vertical pass-B loop).  The state is
// updated IN PLACE by tied-operand instructions, so that the two arms leave it in the same registers (a plain C++
// update gives each arm its own result registers and a copy per value where the arms meet).
__device__ __forceinline__ void mul_inplace(double &x, double r) { asm("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(r)); }
// (ar, ai) <- (ar, ai) * (rr, ri)
__device__ __forceinline__ void rot_inplace(double &ar, double &ai, double rr, double ri) {
    const double t1 = ai * ri, t2 = ar * ri;
    asm("v_fma_f64 %0, %0, %1, -%2" : "+v"(ar) : "v"(rr), "v"(t1));
    asm("v_fma_f64 %0, %0, %1, %2" : "+v"(ai) : "v"(rr), "v"(t2));
}
template <int NB>
__device__ __forceinline__ void kin_rotate1(Kin<NB> &K) {
#pragma unroll
    for (int j = 0; j < NB; j++) rot_inplace(K.ar[j], K.ai[j], K.r1r[j], K.r1i[j]);
}
template <int NB>
__device__ __forceinline__ void kin_decay1(Kin<NB> &K) {
#pragma unroll
    for (int j = 0; j < NB; j++) {
        mul_inplace(K.P[j], K.r1p[j]);
        mul_inplace(K.Q[j], K.r1q[j]);
    }
}
#if RAFTX_UNIT_ROTORS
template <int NB>
__device__ __forceinline__ void kin_rotate2(Kin<NB> &K) {
    kin_rotate1(K);
    kin_rotate1(K);
}
template <int NB>
__device__ __forceinline__ void kin_decay2(Kin<NB> &K) {
    kin_decay1(K);
    kin_decay1(K);
}
// m = 1 or 2 (wave-uniform): the second application sits in a branch; the tied operands keep the state where it is
template <int NB>
__device__ __forceinline__ void kin_rotate(Kin<NB> &K, int m) {
    kin_rotate1(K);
    if (m == 2) kin_rotate1(K);
}
template <int NB>
__device__ __forceinline__ void kin_decay(Kin<NB> &K, int m) {
    kin_decay1(K);
    if (m == 2) kin_decay1(K);
}
#else
template <int NB>
__device__ __forceinline__ void kin_rotate2(Kin<NB> &K) {
#pragma unroll
    for (int j = 0; j < NB; j++) rot_inplace(K.ar[j], K.ai[j], K.r2r[j], K.r2i[j]);
}
template <int NB>
__device__ __forceinline__ void kin_decay2(Kin<NB> &K) {
#pragma unroll
    for (int j = 0; j < NB; j++) {
        mul_inplace(K.P[j], K.r2p[j]);
        mul_inplace(K.Q[j], K.r2q[j]);
    }
}
template <int NB>
__device__ __forceinline__ void kin_rotate(Kin<NB> &K, int m) {
    if (m == 1) kin_rotate1(K);
    else kin_rotate2(K);
}
template <int NB>
__device__ __forceinline__ void kin_decay(Kin<NB> &K, int m) {
    if (m == 1) kin_decay1(K);
    else kin_decay2(K);
}
#endif
template <int NB, int RT>
__device__ __forceinline__ void kin_step_rt(Kin<NB> &K, int m) {      // RT 3 = RT 2 with an upright cross-section
    if (RT != 1 && RT != 4) kin_rotate(K, m);
    if (RT != 2 && RT != 3) kin_decay(K, m);
}

// Two-unit rotors of ONE run type, held by the loop that uses them (the interior strips of a member step by two units; Kin keeps
// the one-unit rotors only): squares of the one-unit rotors, formed at the run start.
template <int NB>
struct Rot2 {
    double r[NB], i[NB];
};
template <int NB>
__device__ __forceinline__ Rot2<NB> rot2_of(const Kin<NB> &K) {
    Rot2<NB> R;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        R.r[j] = fma(K.r1r[j], K.r1r[j], -(K.r1i[j] * K.r1i[j]));
        R.i[j] = 2.0 * (K.r1r[j] * K.r1i[j]);
    }
    return R;
}
template <int NB>
__device__ __forceinline__ void kin_rotate_m(Kin<NB> &K, const Rot2<NB> &R, int m) {
    // the two-unit rotor always, the one-unit rotor's inverse (its conjugate) on top for a one-unit step: ONE conditional arm --
    // with a branch between two updates the compiler merges the arms' results through register copies
#pragma unroll
    for (int j = 0; j < NB; j++) rot_inplace(K.ar[j], K.ai[j], R.r[j], R.i[j]);
    if (m != 2) {
#pragma unroll
        for (int j = 0; j < NB; j++) rot_inplace(K.ar[j], K.ai[j], K.r1r[j], -K.r1i[j]);
    }
}
template <int NB>
struct Dec2 {
    double p[NB], q[NB];
};
template <int NB>
__device__ __forceinline__ Dec2<NB> dec2_of(const Kin<NB> &K) {
    Dec2<NB> D;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        D.p[j] = K.r1p[j] * K.r1p[j];
        D.q[j] = K.r1q[j] * K.r1q[j];
    }
    return D;
}
template <int NB>
__device__ __forceinline__ void kin_decay_m(Kin<NB> &K, const Dec2<NB> &D, int m) {
#pragma unroll
    for (int j = 0; j < NB; j++) {
        mul_inplace(K.P[j], D.p[j]);
        mul_inplace(K.Q[j], D.q[j]);
    }
    if (m != 2) {                                   // one-unit step: back by one unit (the rotors of P and Q are each other's inverses)
#pragma unroll
        for (int j = 0; j < NB; j++) {
            mul_inplace(K.P[j], K.r1q[j]);
            mul_inplace(K.Q[j], K.r1p[j]);
        }
    }
}

// Pass A of one linearisation (raft_member.py:2039-2090, helpers.py:149-184,684): per strip,
// the sums over ALL bins of |v_rel . q|^2 and of the transverse squares, v_rel = u - i w (Xi_t + theta x a).
// X[j][.] = w * XiLast (re/im), so that i w V = i (X_t + X_theta x a).
// Cross-lane sums go through
// this wave's LDS transposition tile (no barrier, no shuffles); per-wave results land in
// vsq[wave][s][3].
template <int NB, int STAGE, bool RC = false>      // RC: the launch has a run-start cache (Kin), filled by the inertial sweep
__device__ __forceinline__ void linearize_passA(cdptr ds, ciptr dsi, int S,
                                                const Lds &l, const Bins<NB> &b, double cb, double sb,
                                                const cplx (&X)[NB][6] PT_ARG) {
    constexpr int CM = RC ? 2 : 0;
    const int tid = opaque((int)threadIdx.x);
    const int lane = tid & 63, wv = tid >> 6;
    ldptr tile = l.tile + wv * TR_ROWS * TR_STRIDE;
    ldptr wr = tile + tile_pos(lane);
    int vstride;
    ldptr vout = vsq_of(l, wv, blockDim.x >> 6, S, vstride);
    vout += ((lane >> 3) >= 3 ? vstride - 3 : 0) + (lane >> 3);          // this reader lane's row: strip s0 or s0 + 1, component row % 3
    Kin<NB> K;
    kin_reset(K, run_cache_of(l));
    if (S <= 0) return;
    StripSrc<STAGE> src(l, ds, dsi);
    // Two strips per batch (6 tile rows).  The cross-lane reduction of batch k is software-pipelined into batch
    // k+1: its 8 tile reads are issued before the first strip of batch k+1 and consumed after it, so neither the
    // write->read turnaround nor the read latency of the tile is exposed.
    ldptr rp = tile + min(lane >> 3, TR_ROWS - 1) * TR_STRIDE + (lane & 7) * 9;
    const int row = lane >> 3;
    const bool writer = (lane & 7) == 0;
    int prev_s0 = -1, prev_rows = 0;                // the batch waiting in the tile: its first strip, its valid rows ...
    bool prev_two = false;                          // ... and whether a strip owns two rows (circular strips: v_q, v_perp) or three
    ldptr vout2 = vsq_of(l, wv, blockDim.x >> 6, S, vstride) + (lane >> 4) * vstride + ((lane >> 3) & 1);
    if constexpr (RUN_LOOPS<NB>) {
    // Loop over RUNS; batches of up to two strips of the same run form the inner loop, specialised by run type (see
    // drag_excitation): a vertical run steps P, Q only, a horizontal run the phasor only and keeps P + Q, P - Q out of
    // the loop.  The run-start evaluation (sincos / exp) is outside the batch loop's body.
    int s = 0;
    int fl = src.flags(min(1, S - 1));              // flags of strip 0; strip 1's are on their way
    // The strip flags come as one vector load per run -- lane i holds those of strip s + 1 + i, fetched beside the run-start
    // evaluation -- and reach the scalar side by v_readlane: no scalar load (and no wait for one) per strip.
    int fa = 0;
    auto run = [&](auto rt_tag, const int n, const int m0) {
        constexpr int RT = decltype(rt_tag)::value;
        if (m0) kin_step_rt<NB, RT>(K, m0);         // continuation of a run longer than 64 strips
        double psh[NB], pdh[NB];                    // RT == 2: P + Q, P - Q of the whole run
#pragma unroll
        for (int j = 0; j < NB; j++) {
            psh[j] = K.P[j] + K.Q[j];
            pdh[j] = K.P[j] - K.Q[j];
        }
        BodyHoist<NB> H;
        RecA r = load_recA(src.rec(s));             // a run has one unit triad (derive_design_tables); the arm is per strip
        body_hoist<NB, RT == 4 ? 1 : RT>(H, X, r);
        // RT 4, a vertical run of circular strips: the phasor a is the same for every strip, so with t1 = a (P+Q), t2 = i a (P-Q)
        //   |v_z|^2               = |a|^2 pd^2 + 2 pd Re(conj(i a) Hz) + |Hz|^2
        //   |v_x|^2 + |v_y|^2     = (cb^2 + sb^2) |a|^2 ps^2 + 2 ps Re(conj(a) (Mh + az Mg)) + |Hx + az Gx|^2 + |Hy + az Gy|^2
        // (H: body terms fixed along the run, G = R(X4), R(-X3): the part that grows with the arm's z; Mh = cb Hx + sb Hy, Mg
        // likewise).  Everything but ps, pd and az is a per-bin constant of the run: 5 FMAs per strip and bin instead of
        // 4 multiplies + 16 FMAs, and the last two terms are a polynomial in az summed over this lane's bins once.
        double A2[NB], A2p[NB], Cz2[NB], C1p[NB], C2p[NB], hz2s = 0.0, D0s = 0.0, D1s = 0.0, D2s = 0.0;
        if constexpr (RT == 4) {
            const double hh = cb * cb + sb * sb;
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const double ar = K.ar[j], ai = K.ai[j];
                const double gxr = X[j][4].im, gxi = -X[j][4].re, gyr = -X[j][3].im, gyi = X[j][3].re;
                A2[j] = fma(ar, ar, ai * ai);
                A2p[j] = hh * A2[j];
                Cz2[j] = 2.0 * fma(ar, H.zi[j], -(ai * H.zr[j]));
                const double mhr = fma(cb, H.xr[j], sb * H.yr[j]), mhi = fma(cb, H.xi[j], sb * H.yi[j]);
                const double mgr = fma(cb, gxr, sb * gyr), mgi = fma(cb, gxi, sb * gyi);
                C1p[j] = 2.0 * fma(ar, mhr, ai * mhi);
                C2p[j] = 2.0 * fma(ar, mgr, ai * mgi);
                hz2s += fma(H.zr[j], H.zr[j], H.zi[j] * H.zi[j]);
                D0s += fma(H.xr[j], H.xr[j], fma(H.xi[j], H.xi[j], fma(H.yr[j], H.yr[j], H.yi[j] * H.yi[j])));
                D1s += 2.0 * fma(H.xr[j], gxr, fma(H.xi[j], gxi, fma(H.yr[j], gyr, H.yi[j] * gyi)));
                D2s += fma(gxr, gxr, fma(gxi, gxi, fma(gyr, gyr, gyi * gyi)));
            }
        }
        // RT 3, an upright pontoon: P + Q, P - Q do not change along the run and the rotation keeps |a|; the arm moves along q,
        // (ax, ay) = (ax0, ay0) + lam (qx, qy).  With R(X) = -i X and
        //   Zc = R(X2) + ay0 R(X3) - ax0 R(X4),  Zl = qy R(X3) - qx R(X4);   Yc = Hy + w2_0 R(X5),  Yl = (p2y qx - p2x qy) R(X5)
        //   |v_q|^2  = alq^2 |a|^2 ps^2 + |Hx|^2                     + 2 alq ps Re(conj(a) Hx)
        //   |v_z|^2  = |a|^2 pd^2 + |Zc + lam Zl|^2                  + 2 pd Re(conj(i a) (Zc + lam Zl))
        //   |v_p2|^2 = al2^2 |a|^2 ps^2 + |Yc + lam Yl|^2            + 2 al2 ps Re(conj(a) (Yc + lam Yl))
        // the first two columns are polynomials in lam summed over this lane's bins once per run; the cross terms cost
        // 10 FMAs per strip and bin (22 in the direct form).
        double hxr[NB], hxi[NB], zcr[NB], zci[NB], zlr[NB], zli[NB], ycr[NB], yci[NB], ylr[NB], yli[NB];
        double E0 = 0.0, e10 = 0.0, e11 = 0.0, e12 = 0.0, e20 = 0.0, e21 = 0.0, e22 = 0.0, ax0 = 0.0, ay0 = 0.0;
        if constexpr (RT == 3) {
            const double alq = r.qx * cb + r.qy * sb, al2 = r.p2x * cb + r.p2y * sb;
            ax0 = r.ax;
            ay0 = r.ay;
            const double w20 = r.p2y * ax0 - r.p2x * ay0, kap2 = r.p2y * r.qx - r.p2x * r.qy;
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const double A2 = fma(K.ar[j], K.ar[j], K.ai[j] * K.ai[j]), ps = psh[j], pd = pdh[j];
                const double zr = fma(-ax0, X[j][4].im, fma(ay0, X[j][3].im, X[j][2].im));
                const double zi = fma(ax0, X[j][4].re, fma(-ay0, X[j][3].re, -X[j][2].re));
                const double lr = fma(-r.qx, X[j][4].im, r.qy * X[j][3].im), li = fma(r.qx, X[j][4].re, -(r.qy * X[j][3].re));
                const double yr = fma(w20, X[j][5].im, H.yr[j]), yi = fma(-w20, X[j][5].re, H.yi[j]);
                const double mr = kap2 * X[j][5].im, mi = -(kap2 * X[j][5].re);
                const double sx = 2.0 * alq * ps, sz = 2.0 * pd, sy = 2.0 * al2 * ps;
                hxr[j] = sx * H.xr[j]; hxi[j] = sx * H.xi[j];
                zcr[j] = sz * zr; zci[j] = sz * zi; zlr[j] = sz * lr; zli[j] = sz * li;
                ycr[j] = sy * yr; yci[j] = sy * yi; ylr[j] = sy * mr; yli[j] = sy * mi;
                const double aps = A2 * ps * ps;
                E0 += fma(alq * alq, aps, fma(H.xr[j], H.xr[j], H.xi[j] * H.xi[j]));
                e10 += fma(A2 * pd, pd, fma(zr, zr, zi * zi));
                e11 += 2.0 * fma(zr, lr, zi * li);
                e12 += fma(lr, lr, li * li);
                e20 += fma(al2 * al2, aps, fma(yr, yr, yi * yi));
                e21 += 2.0 * fma(yr, mr, yi * mi);
                e22 += fma(mr, mr, mi * mi);
            }
        }
        auto strip = [&](double (&v)[3]) {          // strip s at the state K
            if constexpr (RT == 3) {
                const double lam = fma(r.ax - ax0, r.qx, (r.ay - ay0) * r.qy);
                double v0 = E0, v1 = fma(lam, fma(lam, e12, e11), e10), v2 = fma(lam, fma(lam, e22, e21), e20);
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    const double ar = K.ar[j], ai = K.ai[j];
                    v0 = fma(ar, hxr[j], fma(ai, hxi[j], v0));
                    const double wr = fma(lam, zlr[j], zcr[j]), wi = fma(lam, zli[j], zci[j]);
                    v1 = fma(ar, wi, fma(-ai, wr, v1));
                    const double yr = fma(lam, ylr[j], ycr[j]), yi = fma(lam, yli[j], yci[j]);
                    v2 = fma(ar, yr, fma(ai, yi, v2));
                }
                v[0] = v0;
                v[1] = v1;
                v[2] = v2;
            } else if constexpr (RT == 4) {
                const double az = r.az;
                double v0 = hz2s, v1 = fma(az, fma(az, D2s, D1s), D0s);
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    const double ps = K.P[j] + K.Q[j], pd = K.P[j] - K.Q[j];
                    v0 = fma(pd, fma(A2[j], pd, Cz2[j]), v0);
                    v1 = fma(ps, fma(A2p[j], ps, fma(az, C2p[j], C1p[j])), v1);
                }
                v[0] = v0;
                v[1] = v1;
                v[2] = 0.0;
            } else if (RT == 2) {
                passA_core<NB, RT>(K.ar, K.ai, psh, pdh, r, (fl & DSI_CIRC) != 0, cb, sb, X, H, v[0], v[1], v[2]);
            } else {
                double ps[NB], pd[NB];
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    ps[j] = K.P[j] + K.Q[j];
                    pd[j] = K.P[j] - K.Q[j];
                }
                passA_core<NB, RT>(K.ar, K.ai, ps, pd, r, (fl & DSI_CIRC) != 0, cb, sb, X, H, v[0], v[1], v[2]);
            }
        };
        // Batches of two strips (tile rows 0-2: strip A, rows 3-5: strip B).  The cross-lane reduction of a batch is software-
        // pipelined into strip A of the next one: its 8 tile reads are issued before that strip and consumed after it.
        // The run's length n is known, so the strips are a peeled A followed by a COUNTED loop of (step, B, step, A) with the
        // steps unconditional and in place -- exits from the middle of a batch made the compiler carry the stepped state in
        // temporaries and copy it back (22 v_mov_b64 per strip).
        auto stripA = [&]() {
            double pend[8];
            const bool red = prev_s0 >= 0;              // false on the very first strip of the pass only
            if (red) {
#pragma unroll
                for (int e = 0; e < 8; e++) pend[e] = rp[e];
            }
            double va[3];
            strip(va);
            if (red) {
                double a = ((pend[0] + pend[4]) + (pend[1] + pend[5])) + ((pend[2] + pend[6]) + (pend[3] + pend[7]));
                a += dpp_mov<0xB1>(a);
                a += dpp_mov<0x4E>(a);
                a += dpp_mov<0x104>(a);
                if (writer && row < prev_rows) (prev_two ? vout2 : vout)[prev_s0 * vstride] = a;
            }
#pragma unroll
            for (int c = 0; c < (RT == 4 ? 2 : 3); c++) wr[c * TR_STRIDE] = va[c];
            PT_MARK(6);   // strips of pass A
        };
        auto stripB = [&]() {
            double vb[3];
            strip(vb);
#pragma unroll
            for (int c = 0; c < 3; c++) wr[(3 + c) * TR_STRIDE] = vb[c];
            wave_lds_fence();
            prev_s0 = s - 1;
            prev_rows = 6;
            prev_two = false;
            PT_MARK(6);
        };
        // RT 4: circular strips have two sums each (v_q, v_perp), so THREE strips share the six tile rows and one reduction
        auto stripB4 = [&]() {
            double vb[3];
            strip(vb);
            wr[2 * TR_STRIDE] = vb[0];
            wr[3 * TR_STRIDE] = vb[1];
            PT_MARK(6);
        };
        auto stripC4 = [&]() {
            double vc[3];
            strip(vc);
            wr[4 * TR_STRIDE] = vc[0];
            wr[5 * TR_STRIDE] = vc[1];
            PT_MARK(6);
        };
        auto close4 = [&](int nstrip) {            // the batch of the last nstrip strips (ending at s) is complete
            wave_lds_fence();
            prev_s0 = s - (nstrip - 1);
            prev_rows = 2 * nstrip;
            prev_two = true;
        };
        // two-unit rotors of this run type (the general run keeps applying the one-unit ones twice)
        Rot2<NB> R2;
        Dec2<NB> D2;
        if constexpr (RT == 2 || RT == 3) R2 = rot2_of(K);
        if constexpr (RT == 1 || RT == 4) D2 = dec2_of(K);
        const int s_run = s;
        auto advance = [&]() {                      // to the next strip of this run: flags (lane i of fa: strip s_run + 1 + i), arm, state
            ++s;
            fl = __builtin_amdgcn_readlane(fa, s - s_run - 1);
            load_arm(src.rec(s), r);
            if constexpr (RT == 2 || RT == 3) kin_rotate_m(K, R2, fl & DSI_M);
            else if constexpr (RT == 1 || RT == 4) kin_decay_m(K, D2, fl & DSI_M);
            else kin_step_rt<NB, RT>(K, fl & DSI_M);
        };
        stripA();
        int rem = n - 1;
        if constexpr (RT == 4) {
#pragma unroll 1
            for (; rem >= 3; rem -= 3) {
                advance();
                stripB4();
                advance();
                stripC4();
                close4(3);
                advance();
                stripA();
            }
            if (rem == 2) {
                advance();
                stripB4();
                advance();
                stripC4();
                close4(3);
            } else if (rem == 1) {
                advance();
                stripB4();
                close4(2);
            } else {
                close4(1);
            }
        } else {
#pragma unroll 1
        for (; rem >= 2; rem -= 2) {
            advance();
            stripB();
            advance();
            stripA();
        }
        if (rem) {
            advance();
            stripB();
        } else {                                    // an odd strip count: the last batch has one strip
            wave_lds_fence();
            prev_s0 = s;
            prev_rows = 3;
            prev_two = false;
        }
        }
        ++s;
        fl = __builtin_amdgcn_readlane(fa, s - s_run - 1);         // flags of the strip after the run (the last strip's again at the end of the table)
    };
    int rt = 0;
#pragma unroll 1
    while (s < S) {
        // strips of this run (1..64; a longer one continues through another pass of this loop, without a start): the first
        // strip that does not continue it ends the run
        const int ahead = s + 1 + lane;
        fa = dsi[min(ahead, S - 1)];                                // covered by the run start
        const int m0 = fl & DSI_M;
        if (m0 == 0) {
            kin_start<NB, false, CM>(K, run_start_of(src.rec(s)), b, b.c1, cb, sb);
            // a run of one strip has a zero step vector and says nothing about the member's axis: general form
            if (K.vert && K.dec && fabs(src_qz(src.rec(s))) == 1.0) rt = (fl & DSI_CIRC) ? 4 : 1;     // vertical (implies no phase rotation)
            else if (!K.vert && !K.dec) rt = (fl & (DSI_AXAL | DSI_CIRC)) == DSI_AXAL ? 3 : 2;       // horizontal; 3: rectangular with an upright cross-section (a run has one triad and shape)
            else rt = 0;
        }
        const unsigned long long stop = __ballot(ahead >= S || (fa & DSI_M) == 0);
        const int n = stop ? 1 + (int)__builtin_ctzll(stop) : 64;
        if (rt == 4) run(std::integral_constant<int, 4>{}, n, m0);
        else if (rt == 1) run(std::integral_constant<int, 1>{}, n, m0);
        else if (rt == 3) run(std::integral_constant<int, 3>{}, n, m0);
        else if (rt == 2) run(std::integral_constant<int, 2>{}, n, m0);
        else run(std::integral_constant<int, 0>{}, n, m0);
    }
    } else {
#pragma unroll 1
    for (int s0 = 0; s0 < S; s0 += SB) {
        const int nb = min(SB, S - s0);
        double pend[8];
        if (prev_s0 >= 0) {
#pragma unroll
            for (int e = 0; e < 8; e++) pend[e] = rp[e];
        }
        double va[3], vb[3] = {0.0, 0.0, 0.0};
        {
            const auto rec = src.rec(s0);
            const RecA r = load_recA(rec);
            const int fl = src.flags(min(s0 + 1, S - 1));
            passA_strip<NB, CM>(K, r, fl, rec, b, cb, sb, X, va[0], va[1], va[2]);
        }
        if (prev_s0 >= 0) {
            double a = ((pend[0] + pend[4]) + (pend[1] + pend[5])) + ((pend[2] + pend[6]) + (pend[3] + pend[7]));
            a += dpp_mov<0xB1>(a);
            a += dpp_mov<0x4E>(a);
            a += dpp_mov<0x104>(a);
            if (writer && row < prev_rows) vout[prev_s0 * vstride] = a;
        }
        if (nb > 1) {
            const auto rec = src.rec(s0 + 1);
            const RecA r = load_recA(rec);
            const int fl = src.flags(min(s0 + 2, S - 1));
            passA_strip<NB, CM>(K, r, fl, rec, b, cb, sb, X, vb[0], vb[1], vb[2]);
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            wr[c * TR_STRIDE] = va[c];
            wr[(3 + c) * TR_STRIDE] = vb[c];
        }
        wave_lds_fence();
        prev_s0 = s0;
        prev_rows = 3 * nb;
        PT_MARK(6);   // strips of pass A
    }
    }
    {   // drain: the last batch
        const double a = tile_reduce(tile, lane);
        if (writer && row < prev_rows) (prev_two ? vout2 : vout)[prev_s0 * vstride] = a;
    }
    wave_lds_fence();
}
// Strip-lane phase of a linearisation: one lane per strip turns the velocity sums into the
// linearised coefficients (raft_member.py:2093-2110), the heading-projected drag vectors
//   U = sum_c b_c al_c W_c,  V = sum_c b_c ga_c W_c,  W_c = [n_c ; a x n_c]
// (so that translate(Bmat u) = t1 U + t2 V; raft_member.py:2122-2124, helpers.py:468-483) and its
// share of B_drag = sum_{s,c} b_c W_c W_c^T (raft_member.py:2117-2118, helpers.py:537-560).
// FRESH: recompute b_c from vsq (and park them in vsq row 0); otherwise reuse the parked b_c (other
// headings of the same linearisation).
template <bool FRESH>
__device__ __forceinline__ void strip_phase(cdptr ds, ciptr dsi, int S,
                                            const Lds &l, double cb, double sb, bool multi) {
    const int tid = opaque((int)threadIdx.x);       // per-lane LDS addresses of this phase are formed here, not hoisted
    const int lane = tid & 63, wv = tid >> 6, nwv = blockDim.x >> 6;
    double b6[21];
#pragma unroll
    for (int e = 0; e < 21; e++) b6[e] = 0.0;
    for (int s = tid; s < S; s += blockDim.x) {
        cdptr rec = ds + (size_t)s * DS_N;
        double bc[3];
        if (FRESH) {
            double a = 0, c1 = 0, c2 = 0;
            for (int i = 0; i < nwv; i++) {
                int st;
                ldptr r = vsq_of(l, i, nwv, S, st) + s * st;
                a += r[0];
                c1 += r[1];
                c2 += r[2];
            }
            const double vRq = sqrt(0.5 * a);
            double vR1, vR2;
            if (dsi[s] & DSI_CIRC) {
                vR1 = vR2 = sqrt(0.5 * fmax(c1, 0.0));
            } else {
                vR1 = sqrt(0.5 * c1);
                vR2 = sqrt(0.5 * c2);
            }
            bc[0] = rec[DS_DQ] * vRq + rec[DS_DQ + 3] * vRq;     // Bprime_q + Bprime_End (:2093,:2110)
            bc[1] = rec[DS_DQ + 1] * vR1;
            bc[2] = rec[DS_DQ + 2] * vR2;
            l.vsq[s * 3 + 0] = bc[0];
            l.vsq[s * 3 + 1] = bc[1];
            l.vsq[s * 3 + 2] = bc[2];
        } else {
            bc[0] = l.vsq[s * 3 + 0];
            bc[1] = l.vsq[s * 3 + 1];
            bc[2] = l.vsq[s * 3 + 2];
        }
        const double ax = rec[DS_A], ay = rec[DS_A + 1], az = rec[DS_A + 2];
        double U[6] = {0, 0, 0, 0, 0, 0}, V[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double W[6];
            W[0] = rec[DS_Q + 3 * c];
            W[1] = rec[DS_Q + 3 * c + 1];
            W[2] = rec[DS_Q + 3 * c + 2];
            W[3] = ay * W[2] - az * W[1];
            W[4] = az * W[0] - ax * W[2];
            W[5] = ax * W[1] - ay * W[0];
            const double al = bc[c] * (W[0] * cb + W[1] * sb), ga = bc[c] * W[2];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                U[j] += al * W[j];
                V[j] += ga * W[j];
            }
            if (FRESH) {
                int e = 0;
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int j = i; j < 6; j++) b6[e++] += bc[c] * (W[i] * W[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 6; j++) {
            l.uv[s * 12 + j] = U[j];
            l.uv[s * 12 + 6 + j] = V[j];
        }
    }
    if (FRESH) {
        // B_drag: reduce the 21 unique entries over the lanes of each wave (four tile rounds), then over waves; only the
        // waves that own strips (S <= 64: the first alone) take part
        const int nwv_s = min(nwv, (S + 63) >> 6);
        ldptr tile = l.tile + wv * TR_ROWS * TR_STRIDE;
        ldptr wr = tile + tile_pos(lane);
        if (wv < nwv_s)
#pragma unroll
        for (int r0 = 0; r0 < 21; r0 += TR_ROWS) {
            wave_lds_fence();
#pragma unroll
            for (int e = 0; e < TR_ROWS; e++)
                if (r0 + e < 21) wr[e * TR_STRIDE] = b6[r0 + e];
            wave_lds_fence();
            const double a = tile_reduce(tile, lane);
            if ((lane & 7) == 0 && r0 + (lane >> 3) < 21) l.bdw[wv * 24 + r0 + (lane >> 3)] = a;
        }
        wg_sync(multi);
        if (tid < 36) {
            const int i = tid / 6, j = tid % 6;
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            const int e = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
            double acc = 0.0;
            for (int q = 0; q < nwv_s; q++) acc += l.bdw[q * 24 + e];
            l.Bd[tid] = acc;
            l.Bb[tid] = l.mat[36 + tid] + acc;
        }
    }
    wg_sync(multi);
}

// Pass B: drag excitation of one heading with the live coefficients (raft_member.py:2122-2124,
// :2146-2151), ACCUMULATED into F: F += sum_s t1 U_s + t2 V_s.  U,V of the next strip are
// fetched from LDS one strip ahead.
template <int NB>
__device__ __forceinline__ void passB_core(const Kin<NB> &K, const double (&U)[6], const double (&V)[6], cplx (&F)[NB][6]) {
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const double ps = K.P[j] + K.Q[j], pd = K.P[j] - K.Q[j];
        const double t1r = K.ar[j] * ps, t1i = K.ai[j] * ps, t2r = -K.ai[j] * pd, t2i = K.ar[j] * pd;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            F[j][q].re = fma(t1r, U[q], fma(t2r, V[q], F[j][q].re));
            F[j][q].im = fma(t1i, U[q], fma(t2i, V[q], F[j][q].im));
        }
    }
}
template <int NB, int CM, typename RecPtr>
__device__ __forceinline__ void passB_strip(Kin<NB> &K, int fl, RecPtr rec, const Bins<NB> &b,
                                            double cb, double sb, const double (&U)[6], const double (&V)[6],
                                            cplx (&F)[NB][6]) {
    kin_advance<NB, false, CM>(K, fl, rec, b, b.c1, cb, sb);
    passB_core<NB>(K, U, V, F);
}
__device__ __forceinline__ void load_uv(ldptr uv, double (&U)[6], double (&V)[6]) {
#pragma unroll
    for (int q = 0; q < 6; q++) {
        U[q] = uv[q];
        V[q] = uv[6 + q];
    }
}
template <int NB, int STAGE, bool RC = false>
__device__ __forceinline__ void drag_excitation(cdptr ds, ciptr dsi, int S, const Lds &l, const Bins<NB> &b, double cb,
                                                double sb, cplx (&F)[NB][6]) {
    constexpr int CM = RC ? 2 : 0;
    Kin<NB> K;
    kin_reset(K, run_cache_of(l));
    if (S <= 0) return;
    StripSrc<STAGE> src(l, ds, dsi);
    if constexpr (RUN_LOOPS<NB>) {
    // Loop over RUNS; the strips of a run form the inner loop, specialised by what changes along the run:
    //  * vertical member (no phase rotation: a = e^{-i k xi} is the same for every strip): the sums G = sum (P+Q) U_s,
    //    H = sum (P-Q) V_s are REAL and, of their twelve components, five are independent (5 FMAs per strip and bin instead
    //    of 24 + 4); they meet the phasor once per run, F += a G + i a H; the step is two multiplies.  Other runs
    //    without phase rotation (a member across the wave direction) take the loops below with the identity rotor;
    //  * horizontal run (no depth decay: P, Q are the same for every strip): P+Q, P-Q leave the loop, the step is the
    //    phase rotation alone;
    //  * inclined run: the general form.
    // Every strip loop has the shape  body(first strip); while (next strip continues the run) { step; body; }  -- the
    // step feeds the body of the same trip, so the state is advanced in place (with the body ahead of the exit tests the
    // compiler keeps the stepped state in temporaries beside the live one and copies it back on every trip).
    int s = 0;
    int fl = src.flags(min(1, S - 1));              // flags of strip 0; strip 1's are on their way
    // The interior of a member steps by two units, its end strips by one: the two-unit steps get a loop of their own, so that
    // no trip chooses between rotor sets (as selects, or as two arms whose results the compiler then copies together).
#ifndef RAFTX_SPLIT_MASK
#define RAFTX_SPLIT_MASK 0      // which pass-B run loops get the separate two-unit loop: 1 vertical, 2 upright pontoon, 4 horizontal, 8 inclined
#endif
#define RUN_LOOP(WHICH, STEP1, STEP2, STEPM, PRE)      \
    do {                                               \
        body();                                        \
        int m_ = next_step();                          \
        if constexpr ((RAFTX_SPLIT_MASK & (WHICH)) != 0) { \
        _Pragma("unroll 1") while (m_ != 0) {          \
            if (m_ == 2) {                             \
                _Pragma("unroll 1") do {               \
                    PRE;                               \
                    STEP2;                             \
                    body();                            \
                    m_ = next_step();                  \
                } while (m_ == 2);                     \
            } else {                                   \
                PRE;                                   \
                STEP1;                                 \
                body();                                \
                m_ = next_step();                      \
            }                                          \
        }                                              \
        } else {                                       \
        _Pragma("unroll 1") while (m_ != 0) {          \
            PRE;                                       \
            STEPM;                                     \
            body();                                    \
            m_ = next_step();                          \
        }                                              \
        }                                              \
    } while (0)
    // As in pass A: the flags of the strips after s come as one vector load per run (lane i: strip s + 1 + i), a ballot gives
    // the run's length, and the strips are a COUNTED loop whose flags arrive by v_readlane -- no scalar load per strip.
#define RUN_COUNTED(STEPM, PRE)                                    \
    do {                                                           \
        const int s_run_ = s;                                      \
        body();                                                    \
        _Pragma("unroll 1") for (int i_ = 1; i_ < n; i_++) {      \
            ++s;                                                   \
            fl = __builtin_amdgcn_readlane(fa, s - s_run_ - 1);    \
            const int m_ = fl & DSI_M;                             \
            PRE;                                                   \
            STEPM;                                                 \
            body();                                                \
        }                                                          \
        ++s;                                                       \
        fl = __builtin_amdgcn_readlane(fa, s - s_run_ - 1);        \
    } while (0)
#pragma unroll 1
    while (s < S) {
        const int ahead = s + 1 + ((int)threadIdx.x & 63);
        const int fa = dsi[min(ahead, S - 1)];                      // covered by the run start
        const int m0 = fl & DSI_M;
        if (m0 == 0) {
            kin_start<NB, false, CM>(K, run_start_of(src.rec(s)), b, b.c1, cb, sb);
        } else {                                                    // continuation of a run longer than 64 strips
            kin_rotate(K, m0);
            kin_decay(K, m0);
        }
        const unsigned long long stop = __ballot(ahead >= S || (fa & DSI_M) == 0);
        const int n = stop ? 1 + (int)__builtin_ctzll(stop) : 64;
        if (!K.rot && (fl & DSI_VAX)) {
            // Vertical member (q = +-z, p1 and p2 horizontal; arm x, y fixed along the run): with h_s = sum_c b_c al_c n_c (horizontal)
            //   U_s = [h_x, h_y, 0, -a_z h_y, a_z h_x, a_x h_y - a_y h_x],   V_s = b_q [0, 0, 1, a_y, -a_x, 0]
            // so FIVE real sums per bin carry the run (U_0, U_1, U_3, U_4 and V_2 of the rows the strip phase wrote):
            // 5 FMAs per strip and bin instead of 12, and the rest follows from the run's arm when the sums meet the phasor.
            double G0[NB], G1[NB], G3[NB], G4[NB], H2[NB];
#pragma unroll
            for (int j = 0; j < NB; j++) G0[j] = G1[j] = G3[j] = G4[j] = H2[j] = 0.0;
            RecA r;
            load_arm(src.rec(s), r);
            auto body = [&]() {
                ldptr uv = l.uv + s * 12;
                const double u0 = uv[0], u1 = uv[1], u3 = uv[3], u4 = uv[4], v2 = uv[8];
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    const double ps = K.P[j] + K.Q[j], pd = K.P[j] - K.Q[j];
                    G0[j] = fma(ps, u0, G0[j]);
                    G1[j] = fma(ps, u1, G1[j]);
                    G3[j] = fma(ps, u3, G3[j]);
                    G4[j] = fma(ps, u4, G4[j]);
                    H2[j] = fma(pd, v2, H2[j]);
                }
            };
            const Dec2<NB> D2 = dec2_of(K);
            RUN_COUNTED(kin_decay_m(K, D2, m_), );
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const double g5 = r.ax * G1[j] - r.ay * G0[j], h3 = r.ay * H2[j], h4 = -r.ax * H2[j];
                F[j][0].re = fma(K.ar[j], G0[j], F[j][0].re);
                F[j][0].im = fma(K.ai[j], G0[j], F[j][0].im);
                F[j][1].re = fma(K.ar[j], G1[j], F[j][1].re);
                F[j][1].im = fma(K.ai[j], G1[j], F[j][1].im);
                F[j][2].re = fma(-K.ai[j], H2[j], F[j][2].re);
                F[j][2].im = fma(K.ar[j], H2[j], F[j][2].im);
                F[j][3].re = fma(K.ar[j], G3[j], fma(-K.ai[j], h3, F[j][3].re));
                F[j][3].im = fma(K.ai[j], G3[j], fma(K.ar[j], h3, F[j][3].im));
                F[j][4].re = fma(K.ar[j], G4[j], fma(-K.ai[j], h4, F[j][4].re));
                F[j][4].im = fma(K.ai[j], G4[j], fma(K.ar[j], h4, F[j][4].im));
                F[j][5].re = fma(K.ar[j], g5, F[j][5].re);
                F[j][5].im = fma(K.ai[j], g5, F[j][5].im);
            }
        } else if (!K.dec && !K.vert && (fl & (DSI_AXAL | DSI_CIRC)) == DSI_AXAL) {
            // Upright pontoon (see linearize_passA): q, p2 horizontal, p1 = +-z.  With W_c = [n_c ; arm x n_c] and the arm moving
            // along q, W_q is the same for every strip of the run, W_p2 changes in its last component only
            // (w2 = (arm x p2)_z) and V_s = b_1 [0, 0, 1, a_y, -a_x, 0].  The arm moves along q, (a_x, a_y) = (a_x0, a_y0) + lam (q_x, q_y),
            // so w2, a_x, a_y are linear in lam and the strips add to FIVE complex scalars per bin
            //   Sq = sum b_q al_q a,  S2 = sum b_2 al_2 a,  S2l = sum lam b_2 al_2 a,  S1 = sum b_1 a,  S1l = sum lam b_1 a
            // (10 FMAs per strip and bin instead of 24, from the drag coefficients b_c the strip phase left in vsq row 0 --
            // the U, V rows are not read); they meet P+Q, P-Q (no depth decay along the run) and the constant vectors once per run.
            RecA r = load_recA(src.rec(s));
            const double alq = r.qx * cb + r.qy * sb, al2 = r.p2x * cb + r.p2y * sb;
            const double ax0 = r.ax, ay0 = r.ay, az = r.az;
            const double wq = ax0 * r.qy - ay0 * r.qx, w20 = r.p2y * ax0 - r.p2x * ay0, kap2 = r.p2y * r.qx - r.p2x * r.qy;
            double Sq[NB][2], S2[NB][2], S2l[NB][2], S1[NB][2], S1l[NB][2];
#pragma unroll
            for (int j = 0; j < NB; j++)
                Sq[j][0] = Sq[j][1] = S2[j][0] = S2[j][1] = S2l[j][0] = S2l[j][1] = S1[j][0] = S1[j][1] = S1l[j][0] = S1l[j][1] = 0.0;
            auto body = [&]() {
                ldptr bc = l.vsq + s * 3;
                const double lam = fma(r.ax - ax0, r.qx, (r.ay - ay0) * r.qy);
                const double bq = bc[0] * alq, b1 = bc[1], b2 = bc[2] * al2;
                const double b2l = b2 * lam, b1l = b1 * lam;
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    const double ar = K.ar[j], ai = K.ai[j];
                    Sq[j][0] = fma(bq, ar, Sq[j][0]);    Sq[j][1] = fma(bq, ai, Sq[j][1]);
                    S2[j][0] = fma(b2, ar, S2[j][0]);    S2[j][1] = fma(b2, ai, S2[j][1]);
                    S2l[j][0] = fma(b2l, ar, S2l[j][0]); S2l[j][1] = fma(b2l, ai, S2l[j][1]);
                    S1[j][0] = fma(b1, ar, S1[j][0]);    S1[j][1] = fma(b1, ai, S1[j][1]);
                    S1l[j][0] = fma(b1l, ar, S1l[j][0]); S1l[j][1] = fma(b1l, ai, S1l[j][1]);
                }
            };
            const Rot2<NB> R2 = rot2_of(K);
            RUN_COUNTED(kin_rotate_m(K, R2, m_), load_arm(src.rec(s), r));
            double S26[NB][2], S1y[NB][2], S1x[NB][2];
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const double ps = K.P[j] + K.Q[j], pd = K.P[j] - K.Q[j];
                Sq[j][0] *= ps;  Sq[j][1] *= ps;
                S2[j][0] *= ps;  S2[j][1] *= ps;
                S2l[j][0] *= ps; S2l[j][1] *= ps;
                double t;                                             // times i pd
                t = S1[j][0];  S1[j][0] = -S1[j][1] * pd;   S1[j][1] = t * pd;
                t = S1l[j][0]; S1l[j][0] = -S1l[j][1] * pd; S1l[j][1] = t * pd;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    S26[j][e] = fma(kap2, S2l[j][e], w20 * S2[j][e]);
                    S1y[j][e] = fma(r.qy, S1l[j][e], ay0 * S1[j][e]);
                    S1x[j][e] = fma(r.qx, S1l[j][e], ax0 * S1[j][e]);
                }
            }
            const double c3q = -az * r.qy, c3p = -az * r.p2y, c4q = az * r.qx, c4p = az * r.p2x;
#pragma unroll
            for (int j = 0; j < NB; j++) {
                F[j][0].re = fma(r.qx, Sq[j][0], fma(r.p2x, S2[j][0], F[j][0].re));
                F[j][0].im = fma(r.qx, Sq[j][1], fma(r.p2x, S2[j][1], F[j][0].im));
                F[j][1].re = fma(r.qy, Sq[j][0], fma(r.p2y, S2[j][0], F[j][1].re));
                F[j][1].im = fma(r.qy, Sq[j][1], fma(r.p2y, S2[j][1], F[j][1].im));
                F[j][2].re += S1[j][0];
                F[j][2].im += S1[j][1];
                F[j][3].re = fma(c3q, Sq[j][0], fma(c3p, S2[j][0], F[j][3].re + S1y[j][0]));
                F[j][3].im = fma(c3q, Sq[j][1], fma(c3p, S2[j][1], F[j][3].im + S1y[j][1]));
                F[j][4].re = fma(c4q, Sq[j][0], fma(c4p, S2[j][0], F[j][4].re - S1x[j][0]));
                F[j][4].im = fma(c4q, Sq[j][1], fma(c4p, S2[j][1], F[j][4].im - S1x[j][1]));
                F[j][5].re = fma(wq, Sq[j][0], F[j][5].re + S26[j][0]);
                F[j][5].im = fma(wq, Sq[j][1], F[j][5].im + S26[j][1]);
            }
        } else if (!K.dec) {
            double ps[NB], pd[NB];
#pragma unroll
            for (int j = 0; j < NB; j++) {
                ps[j] = K.P[j] + K.Q[j];
                pd[j] = K.P[j] - K.Q[j];
            }
            auto body = [&]() {
                double U[6], V[6];
                load_uv(l.uv + s * 12, U, V);
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    const double t1r = K.ar[j] * ps[j], t1i = K.ai[j] * ps[j], t2r = -K.ai[j] * pd[j], t2i = K.ar[j] * pd[j];
#pragma unroll
                    for (int q = 0; q < 6; q++) {
                        F[j][q].re = fma(t1r, U[q], fma(t2r, V[q], F[j][q].re));
                        F[j][q].im = fma(t1i, U[q], fma(t2i, V[q], F[j][q].im));
                    }
                }
            };
            RUN_COUNTED(kin_rotate(K, m_), );
        } else {
            auto body = [&]() {
                double U[6], V[6];
                load_uv(l.uv + s * 12, U, V);
                passB_core<NB>(K, U, V, F);
            };
            RUN_COUNTED((kin_rotate(K, m_), kin_decay(K, m_)), );
        }
    }
    } else {
#pragma unroll 1
    for (int s = 0; s < S; s++) {
        double U[6], V[6];
        load_uv(l.uv + s * 12, U, V);               // issued before the (branchy) kinematics update
        const int fl = src.flags(min(s + 1, S - 1));
        passB_strip<NB, CM>(K, fl, src.rec(s), b, cb, sb, U, V, F);
    }
    }
}

// Inertial excitation for real (frequency-independent) inertia coefficients, in the same factored form as pass B:
//   Imat ud = sum_c I_c n_c (n_c . ud),  n_c . ud = i w (al_c t1 + ga_c t2)   (al_c = n_c . (cb, sb, 0), ga_c = n_cz)
//   => F += i w (t1 U' + t2 V') + pDyn A',   U' = sum_c I_c al_c W_c,  V' = sum_c I_c ga_c W_c,  A' = a_i W_q,
//   W_c = [n_c ; a x n_c]   (raft_member.py:1984-1991, helpers.py:468-483).
// U', V' are built once per strip by one lane each (into the uv rows, free before the first linearisation); the bin
// sweep then costs 36 FMAs per strip and bin instead of the ~60 of the direct form.  MacCamy-Fuchs strips (complex
// per-bin Cm) cannot be factored this way: in kernels with KF_MCF their runs take the direct form (mcf_strip), the other
// runs of the design the factored loops.
template <int NB, bool RC = false, bool MCF = false>
__device__ __forceinline__ void inertial_excitation_uv(const DevTables &T, cdptr ds, ciptr dsi, int S, const Lds &l,
                                                       const Bins<NB> &b, int ic, int ih, double cb, double sb,
                                                       cplx (&F)[NB][6], bool multi, const cplx *__restrict__ cm = nullptr) {
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        cdptr rec = ds + (size_t)s * DS_N;
        const double ax = rec[DS_A], ay = rec[DS_A + 1], az = rec[DS_A + 2];
        double U[6] = {0, 0, 0, 0, 0, 0}, V[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double W[6];
            W[0] = rec[DS_Q + 3 * c];
            W[1] = rec[DS_Q + 3 * c + 1];
            W[2] = rec[DS_Q + 3 * c + 2];
            W[3] = ay * W[2] - az * W[1];
            W[4] = az * W[0] - ax * W[2];
            W[5] = ax * W[1] - ay * W[0];
            const double I = rec[DS_IQ + c];
            const double al = I * (W[0] * cb + W[1] * sb), ga = I * W[2];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                U[j] += al * W[j];
                V[j] += ga * W[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 6; j++) {
            l.uv[s * 12 + j] = U[j];
            l.uv[s * 12 + 6 + j] = V[j];
        }
    }
    wg_sync(multi);
    double ws[NB], wd[NB], sp[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const double z0r = T.zeta[((size_t)ic * T.nHead + ih) * T.nw + b.iw[j]];
        const double z0 = b.act[j] ? z0r : 0.0;
        sp[j] = T.rho * T.g * z0 * T.cch[b.iw[j]];                         // rho g zeta0 / cosh kh scaling (helpers.py:231)
        const double qm = (depth_mode(b.k[j], b.depth) == 1) ? 0.0 : 1.0;   // deep water: Sh = Ch = e^{kz}
        ws[j] = b.w[j] * b.c1[j];                                           // w . (w zeta0 csh)
        wd[j] = ws[j] * qm;
    }
    double one[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) one[j] = 1.0;
    Kin<NB> K;
    kin_reset(K, run_cache_of(l));                 // this sweep fills the run-start cache
    // one strip in the general form: F += i w (t1 U' + t2 V') + pDyn A'
    auto general_strip = [&](int s) {
        cdptr rec = ds + (size_t)s * DS_N;
        double U[6], V[6];
        load_uv(l.uv + s * 12, U, V);
        const double ax = rec[DS_A], ay = rec[DS_A + 1], az = rec[DS_A + 2];
        const double qx = rec[DS_Q], qy = rec[DS_Q + 1], qz = rec[DS_Q + 2];
        const double ai_ = rec[DS_IQ + 3];
        double Aq[6];
        Aq[0] = ai_ * qx;
        Aq[1] = ai_ * qy;
        Aq[2] = ai_ * qz;
        Aq[3] = ai_ * (ay * qz - az * qy);
        Aq[4] = ai_ * (az * qx - ax * qz);
        Aq[5] = ai_ * (ax * qy - ay * qx);
#pragma unroll
        for (int j = 0; j < NB; j++) {
            // i w t1 = i w s1 (P + Qk) a,  i w t2 = -w s1 (P - Qk) a   (a = e^{-i k xi}; Qk = 0 in deep water)
            const double hs = fma(wd[j], K.Q[j], ws[j] * K.P[j]), hd = fma(-wd[j], K.Q[j], ws[j] * K.P[j]);
            const double pp = sp[j] * (K.P[j] + K.Q[j]);                  // rho g zeta_s Cc
            const double t1r = -hs * K.ai[j], t1i = hs * K.ar[j];
            const double t2r = -hd * K.ar[j], t2i = -hd * K.ai[j];
            const double pr = pp * K.ar[j], pi = pp * K.ai[j];
#pragma unroll
            for (int q = 0; q < 6; q++) {
                F[j][q].re = fma(t1r, U[q], fma(t2r, V[q], fma(pr, Aq[q], F[j][q].re)));
                F[j][q].im = fma(t1i, U[q], fma(t2i, V[q], fma(pi, Aq[q], F[j][q].im)));
            }
        }
    };
    // one MacCamy-Fuchs strip (KF_MCF kernels): the transverse inertia coefficients are rhoV Cm(w), complex and per bin
    // (raft_member.py:1415-1420), which no sum over a run factors -- the direct form of inertial_excitation<.., true>
    auto mcf_strip = [&](int s) {
        cdptr rec = ds + (size_t)s * DS_N;
        const double ax = rec[DS_A], ay = rec[DS_A + 1], az = rec[DS_A + 2];
        const double ai_ = rec[DS_IQ + 3], rhoV = rec[DS_IQ + 4], I0 = rec[DS_IQ];
        const int mcf = (int)rec[DS_MCF];
        double al[3], ga[3], n[3][3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            n[c][0] = rec[DS_Q + 3 * c];
            n[c][1] = rec[DS_Q + 3 * c + 1];
            n[c][2] = rec[DS_Q + 3 * c + 2];
            al[c] = n[c][0] * cb + n[c][1] * sb;
            ga[c] = n[c][2];
        }
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const double hs = fma(wd[j], K.Q[j], ws[j] * K.P[j]), hd = fma(-wd[j], K.Q[j], ws[j] * K.P[j]);
            const double pp = sp[j] * (K.P[j] + K.Q[j]);
            // n_c . ud = i w (al_c t1 + ga_c t2) with i w t1 = i hs a, i w t2 = -hd a
            const cplx it1 = {-hs * K.ai[j], hs * K.ar[j]}, it2 = {-hd * K.ar[j], -hd * K.ai[j]};
            const cplx pd = {pp * K.ar[j], pp * K.ai[j]};
            cplx f3[3] = {{0, 0}, {0, 0}, {0, 0}};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const cplx a = {al[c] * it1.re + ga[c] * it2.re, al[c] * it1.im + ga[c] * it2.im};
                cplx g;
                if (c == 0) {
                    g = {I0 * a.re + pd.re * ai_, I0 * a.im + pd.im * ai_};
                } else {
                    const cplx m = cm[((size_t)mcf * 2 + (c - 1)) * T.nw + b.iw[j]];
                    g = cmul(cscale(m, rhoV), a);
                }
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    f3[q].re = fma(g.re, n[c][q], f3[q].re);
                    f3[q].im = fma(g.im, n[c][q], f3[q].im);
                }
            }
            F[j][0] = cadd(F[j][0], f3[0]);
            F[j][1] = cadd(F[j][1], f3[1]);
            F[j][2] = cadd(F[j][2], f3[2]);
            F[j][3].re += ay * f3[2].re - az * f3[1].re;
            F[j][3].im += ay * f3[2].im - az * f3[1].im;
            F[j][4].re += az * f3[0].re - ax * f3[2].re;
            F[j][4].im += az * f3[0].im - ax * f3[2].im;
            F[j][5].re += ax * f3[1].re - ay * f3[0].re;
            F[j][5].im += ax * f3[1].im - ay * f3[0].im;
        }
    };
    if constexpr (RUN_LOOPS<NB>) {
    // Loop over runs, specialised as in drag_excitation: with c1 = i hs a, c2 = -hd a, c3 = pp a the three terms are real
    // multiples of the phasor (times i for the first), so
    //  * vertical member (no phase rotation): REAL sums over the run's strips of hs U'_{0,1,3,4} and of pp a_i q_z - hd V'_2
    //    (6 FMAs per strip and bin instead of 36 + 12), which meet the phasor once per run;
    //  * upright pontoon (no depth decay; W_q the same for every strip, V'_s = I_1 [0, 0, 1, a_y, -a_x, 0]): seven complex sums
    //    of the rotating phasor with real per-strip weights, which meet hs, hd, pp and the constant vectors once per run;
    //  * anything else: the general strip.
    if (S > 0) {
        int s = 0;
        int fl = dsi[0], fn = dsi[min(1, S - 1)];
        auto next_step = [&]() -> int {
            if (++s >= S) return 0;
            fl = fn;
            fn = dsi[min(s + 1, S - 1)];
            return fl & DSI_M;
        };
#pragma unroll 1
        while (s < S) {
            cdptr rec0 = ds + (size_t)s * DS_N;
            kin_start<NB, true, RC ? 1 : 0>(K, run_start_of(rec0), b, one, cb, sb);
            if (MCF && rec0[DS_MCF] >= 0.0) {                       // a run is one member: MacCamy-Fuchs or not as a whole
                auto body = [&]() { mcf_strip(s); };
                RUN_LOOP(8, (kin_rotate1(K), kin_decay1(K)), (kin_rotate2(K), kin_decay2(K)), (kin_rotate(K, m_), kin_decay(K, m_)), );
            } else if (!K.rot && (fl & DSI_VAX)) {
                double G0[NB], G1[NB], G3[NB], G4[NB], Z2[NB];
#pragma unroll
                for (int j = 0; j < NB; j++) G0[j] = G1[j] = G3[j] = G4[j] = Z2[j] = 0.0;
                const double ax = rec0[DS_A], ay = rec0[DS_A + 1];
                auto body = [&]() {
                    ldptr uv = l.uv + s * 12;
                    const double u0 = uv[0], u1 = uv[1], u3 = uv[3], u4 = uv[4], v2 = uv[8];
                    cdptr rec = ds + (size_t)s * DS_N;
                    const double aq = rec[DS_IQ + 3] * rec[DS_Q + 2];
#pragma unroll
                    for (int j = 0; j < NB; j++) {
                        const double wp = ws[j] * K.P[j];
                        const double hs = fma(wd[j], K.Q[j], wp), hd = fma(-wd[j], K.Q[j], wp);
                        const double pp = sp[j] * (K.P[j] + K.Q[j]);
                        G0[j] = fma(hs, u0, G0[j]);
                        G1[j] = fma(hs, u1, G1[j]);
                        G3[j] = fma(hs, u3, G3[j]);
                        G4[j] = fma(hs, u4, G4[j]);
                        Z2[j] = fma(pp, aq, fma(-hd, v2, Z2[j]));
                    }
                };
                const Dec2<NB> D2 = dec2_of(K);
                RUN_LOOP(1, kin_decay1(K), kin_decay2(K), kin_decay_m(K, D2, m_), );
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    const double ar = K.ar[j], ai = K.ai[j];
                    const double g5 = ax * G1[j] - ay * G0[j], z3 = ay * Z2[j], z4 = -ax * Z2[j];
                    F[j][0].re = fma(-ai, G0[j], F[j][0].re);
                    F[j][0].im = fma(ar, G0[j], F[j][0].im);
                    F[j][1].re = fma(-ai, G1[j], F[j][1].re);
                    F[j][1].im = fma(ar, G1[j], F[j][1].im);
                    F[j][2].re = fma(ar, Z2[j], F[j][2].re);
                    F[j][2].im = fma(ai, Z2[j], F[j][2].im);
                    F[j][3].re = fma(-ai, G3[j], fma(ar, z3, F[j][3].re));
                    F[j][3].im = fma(ar, G3[j], fma(ai, z3, F[j][3].im));
                    F[j][4].re = fma(-ai, G4[j], fma(ar, z4, F[j][4].re));
                    F[j][4].im = fma(ar, G4[j], fma(ai, z4, F[j][4].im));
                    F[j][5].re = fma(-ai, g5, F[j][5].re);
                    F[j][5].im = fma(ar, g5, F[j][5].im);
                }
            } else if (!K.dec && !K.vert && (fl & (DSI_AXAL | DSI_CIRC)) == DSI_AXAL) {
                const double qx = rec0[DS_Q], qy = rec0[DS_Q + 1], p2x = rec0[DS_P2], p2y = rec0[DS_P2 + 1];
                const double alq = qx * cb + qy * sb, al2 = p2x * cb + p2y * sb;
                const double wq = rec0[DS_A] * qy - rec0[DS_A + 1] * qx, az = rec0[DS_A + 2];
                double Sq[NB][2], S2[NB][2], S26[NB][2], S1[NB][2], S1y[NB][2], S1x[NB][2], SA[NB][2];
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    Sq[j][0] = Sq[j][1] = S2[j][0] = S2[j][1] = S26[j][0] = S26[j][1] = SA[j][0] = SA[j][1] = 0.0;
                    S1[j][0] = S1[j][1] = S1y[j][0] = S1y[j][1] = S1x[j][0] = S1x[j][1] = 0.0;
                }
                auto body = [&]() {
                    cdptr rec = ds + (size_t)s * DS_N;
                    const double ax = rec[DS_A], ay = rec[DS_A + 1];
                    const double bq = rec[DS_IQ] * alq, b1 = rec[DS_IQ + 1], b2 = rec[DS_IQ + 2] * al2, ba = rec[DS_IQ + 3];
                    const double b2w = b2 * (p2y * ax - p2x * ay), b1y = b1 * ay, b1x = b1 * ax;
#pragma unroll
                    for (int j = 0; j < NB; j++) {
                        const double ar = K.ar[j], ai = K.ai[j];
                        Sq[j][0] = fma(bq, ar, Sq[j][0]);    Sq[j][1] = fma(bq, ai, Sq[j][1]);
                        S2[j][0] = fma(b2, ar, S2[j][0]);    S2[j][1] = fma(b2, ai, S2[j][1]);
                        S26[j][0] = fma(b2w, ar, S26[j][0]); S26[j][1] = fma(b2w, ai, S26[j][1]);
                        S1[j][0] = fma(b1, ar, S1[j][0]);    S1[j][1] = fma(b1, ai, S1[j][1]);
                        S1y[j][0] = fma(b1y, ar, S1y[j][0]); S1y[j][1] = fma(b1y, ai, S1y[j][1]);
                        S1x[j][0] = fma(b1x, ar, S1x[j][0]); S1x[j][1] = fma(b1x, ai, S1x[j][1]);
                        SA[j][0] = fma(ba, ar, SA[j][0]);    SA[j][1] = fma(ba, ai, SA[j][1]);
                    }
                };
                const Rot2<NB> R2 = rot2_of(K);
                RUN_LOOP(2, kin_rotate1(K), kin_rotate2(K), kin_rotate_m(K, R2, m_), );
                const double c3q = -az * qy, c3p = -az * p2y, c4q = az * qx, c4p = az * p2x;
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    const double wp = ws[j] * K.P[j];
                    const double hs = fma(wd[j], K.Q[j], wp), hd = fma(-wd[j], K.Q[j], wp);
                    const double pp = sp[j] * (K.P[j] + K.Q[j]);
                    // c1 = i hs a:  (re, im) -> (-hs im, hs re);  c2 = -hd a;  c3 = pp a (shares W_q with the axial term)
                    const double qr = fma(pp, SA[j][0], -hs * Sq[j][1]), qi = fma(pp, SA[j][1], hs * Sq[j][0]);
                    const double s2r = -hs * S2[j][1], s2i = hs * S2[j][0];
                    const double s26r = -hs * S26[j][1], s26i = hs * S26[j][0];
                    const double s1r = -hd * S1[j][0], s1i = -hd * S1[j][1];
                    const double s1yr = -hd * S1y[j][0], s1yi = -hd * S1y[j][1];
                    const double s1xr = -hd * S1x[j][0], s1xi = -hd * S1x[j][1];
                    F[j][0].re = fma(qx, qr, fma(p2x, s2r, F[j][0].re));
                    F[j][0].im = fma(qx, qi, fma(p2x, s2i, F[j][0].im));
                    F[j][1].re = fma(qy, qr, fma(p2y, s2r, F[j][1].re));
                    F[j][1].im = fma(qy, qi, fma(p2y, s2i, F[j][1].im));
                    F[j][2].re += s1r;
                    F[j][2].im += s1i;
                    F[j][3].re = fma(c3q, qr, fma(c3p, s2r, F[j][3].re + s1yr));
                    F[j][3].im = fma(c3q, qi, fma(c3p, s2i, F[j][3].im + s1yi));
                    F[j][4].re = fma(c4q, qr, fma(c4p, s2r, F[j][4].re - s1xr));
                    F[j][4].im = fma(c4q, qi, fma(c4p, s2i, F[j][4].im - s1xi));
                    F[j][5].re = fma(wq, qr, F[j][5].re + s26r);
                    F[j][5].im = fma(wq, qi, F[j][5].im + s26i);
                }
            } else {
                auto body = [&]() { general_strip(s); };
                RUN_LOOP(8, (kin_rotate1(K), kin_decay1(K)), (kin_rotate2(K), kin_decay2(K)), (kin_rotate(K, m_), kin_decay(K, m_)), );
            }
        }
    }
    } else {
    if (S > 0) {
        int fn = dsi[0];
#pragma unroll 1
        for (int s = 0; s < S; s++) {
            const int fl = fn;
            fn = dsi[min(s + 1, S - 1)];
            kin_advance<NB, true, RC ? 1 : 0>(K, fl, ds + (size_t)s * DS_N, b, one, cb, sb);
            if (MCF && ds[(size_t)s * DS_N + DS_MCF] >= 0.0) mcf_strip(s);
            else general_strip(s);
        }
    }
    }
    // (the caller's barrier before the first linearisation also orders these uv reads before the strip phase's writes)
}

// ------------------------------------------------------------------ 6x6 complex solve in registers
struct Lu6 {
    double ar[6][6], ai[6][6];
};

// x <- A^-1 x by Gaussian elimination with partial pivoting on the augmented system [A | x]:
// the pivot rule of LAPACK zgetrf/zgesv (np.linalg.solve, raft_model.py:1089): largest
// |re|+|im| in the column (izamax), full row interchange.  Everything stays in registers:
// row swaps are predicated selects (wave-uniformly skipped when no lane needs one).
__device__ __forceinline__ void solve6(Lu6 &A, cplx x[6]) {
#pragma unroll
    for (int k = 0; k < 6; k++) {
        int p = k;
        double best = fabs(A.ar[k][k]) + fabs(A.ai[k][k]);
#pragma unroll
        for (int r = k + 1; r < 6; r++) {
            double v = fabs(A.ar[r][k]) + fabs(A.ai[r][k]);
            if (v > best) {
                best = v;
                p = r;
            }
        }
        if (__any(p != k)) {
#pragma unroll
            for (int r = k + 1; r < 6; r++) {
                const bool sw = (p == r);
                if (!__any(sw)) continue;           // neighbouring bins mostly agree on the pivot row: one or two candidates per wave
#pragma unroll
                for (int c = k; c < 6; c++) {
                    double tr = A.ar[k][c], ti = A.ai[k][c];
                    A.ar[k][c] = sw ? A.ar[r][c] : tr;
                    A.ai[k][c] = sw ? A.ai[r][c] : ti;
                    A.ar[r][c] = sw ? tr : A.ar[r][c];
                    A.ai[r][c] = sw ? ti : A.ai[r][c];
                }
                double tr = x[k].re, ti = x[k].im;
                x[k].re = sw ? x[r].re : tr;
                x[k].im = sw ? x[r].im : ti;
                x[r].re = sw ? tr : x[r].re;
                x[r].im = sw ? ti : x[r].im;
            }
        }
        double pr = A.ar[k][k], pi = A.ai[k][k];
        // 1 / |pivot|^2: v_rcp_f64 + two Newton steps (5 instructions; the IEEE division sequence is 12).  A zero pivot
        // still gives inf -> NaN and is flagged; |pivot|^2 of any physical system is far from the exponent range's ends.
        const double y = pr * pr + pi * pi;
        double d = __builtin_amdgcn_rcp(y);
        d = fma(fma(-y, d, 1.0), d, d);
        d = fma(fma(-y, d, 1.0), d, d);
        double ir = pr * d, ii = -pi * d;
        A.ar[k][k] = ir;            // keep the reciprocal pivot for the back substitution
        A.ai[k][k] = ii;
#pragma unroll
        for (int r = k + 1; r < 6; r++) {
            double lr = A.ar[r][k] * ir - A.ai[r][k] * ii;
            double li = A.ar[r][k] * ii + A.ai[r][k] * ir;
#pragma unroll
            for (int c = k + 1; c < 6; c++) {      // two FMAs per part (a -= l u written as a difference costs three)
                const double ur = A.ar[k][c], ui = A.ai[k][c];
                A.ar[r][c] = fma(-lr, ur, fma(li, ui, A.ar[r][c]));
                A.ai[r][c] = fma(-lr, ui, fma(-li, ur, A.ai[r][c]));
            }
            const double xr = x[k].re, xi = x[k].im;
            x[r].re = fma(-lr, xr, fma(li, xi, x[r].re));
            x[r].im = fma(-lr, xi, fma(-li, xr, x[r].im));
        }
    }
#pragma unroll
    for (int k = 5; k >= 0; k--) {
        cplx s = x[k];
#pragma unroll
        for (int c = k + 1; c < 6; c++) {
            s.re = fma(-A.ar[k][c], x[c].re, fma(A.ai[k][c], x[c].im, s.re));
            s.im = fma(-A.ar[k][c], x[c].im, fma(-A.ai[k][c], x[c].re, s.im));
        }
        x[k] = {s.re * A.ar[k][k] - s.im * A.ai[k][k], s.re * A.ai[k][k] + s.im * A.ar[k][k]};
    }
}

// kernel specialisation flags
#define KF_FDEP 1    // frequency-dependent M(w), B(w)
#define KF_OUTZ 2    // export Z
#define KF_OUTF 4    // export F_wave
#define KF_EXTRA 8   // F_extra input
#define KF_MCF 16    // MacCamy-Fuchs complex Cm table
#define KF_MULTI 32  // more than one wave heading
#define KF_XLIO 64   // linearisation point given (restart) / exported (raft_model.py:1108-1131 re-entry)
#define KF_ALL 127

#ifndef RAFTX_EQ_ORDER
#define RAFTX_EQ_ORDER 4, 3, 2, 1, 0, 5
#endif
__device__ constexpr int EQ_ORDER[6] = {RAFTX_EQ_ORDER};

// Assemble and solve one bin's 6x6 system: x <- Z^-1 x  (raft_model.py:1086-1089)
// Frequency-dependent added mass / damping of a design, [2][36][nw] doubles, as a raw buffer: entry e of bin iw is one
// buffer_load_dwordx2 with the lane's byte offset in ONE VGPR and the entry's row offset in an SGPR (a 64-bit global
// pointer per entry costs a VALU add and two VGPRs each, and the compiler keeps all 72 of them).  No table: a buffer of
// zero records, whose loads return 0.
typedef __amdgpu_buffer_rsrc_t MbRsrc;
__device__ __forceinline__ MbRsrc mb_rsrc(const double *MBw_design, int nw) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)MBw_design, 0, MBw_design ? 72 * nw * (int)sizeof(double) : 0, 0x00020000);
}
__device__ __forceinline__ double mb_load(MbRsrc r, int lane_bytes, int row_bytes) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, lane_bytes, row_bytes, 0));
}
template <int FLAGS>
__device__ __forceinline__ void assemble_and_solve(const Lds &l, MbRsrc mb, int nw, int iw, double w, cplx x[6],
                                                   cplx *__restrict__ Zout, bool active) {
    Lu6 lu;
    const double w2 = w * w;
    if constexpr ((FLAGS & (KF_FDEP | KF_OUTZ)) != 0) iw = opaque(iw);   // per-entry addresses are formed here, not hoisted
    const int row = nw * (int)sizeof(double), lane_bytes = iw * (int)sizeof(double);
    // The equations enter the elimination in the order EQ_ORDER (register row i holds equation EQ_ORDER[i]): a
    // compile-time renaming, free.  Partial pivoting chooses rows by magnitude, so the same pivot rows, multipliers and
    // updates follow whatever the starting order (it only decides exact ties); what changes is how often a row
    // interchange is needed.  For a floating body referred to a point near the waterline the surge-pitch / sway-roll
    // inertia coupling m z_g w^2 exceeds m w^2, and LAPACK's first two interchanges are 0<->4 and 1<->3 in ~94 % of
    // the bins of the VolturnUS-S sweep; starting from that arrangement makes the predicated swap blocks of the two
    // largest steps wave-uniformly skippable.
    if constexpr ((FLAGS & KF_FDEP) != 0) {
        // All 72 loads of the bin's M(w), B(w) column are issued back to back and land in the registers of the system they
        // become part of -- one exposed round trip per solve instead of one per row, and no staging registers.
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const int e = EQ_ORDER[r] * 6 + c;
                lu.ar[r][c] = mb_load(mb, lane_bytes, e * row);
                lu.ai[r][c] = mb_load(mb, lane_bytes, (36 + e) * row);
            }
    }
#pragma unroll
    for (int r = 0; r < 6; r++) {
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const int e = EQ_ORDER[r] * 6 + c;
            double M = l.mat[e], B = l.Bb[e];
            if constexpr ((FLAGS & KF_FDEP) != 0) {
                M += lu.ar[r][c];
                B += lu.ai[r][c];
            }
            lu.ar[r][c] = fma(-w2, M, l.mat[72 + e]);     // Z = -w^2 M + i w B + C  (:1086)
            lu.ai[r][c] = w * B;
        }
        __builtin_amdgcn_sched_barrier(0);               // assemble row by row: bounds the LDS loads in flight
    }
    if constexpr ((FLAGS & KF_OUTZ) != 0) {
        if (active && Zout) {      // last iterate wins (fowt.Z, :1155)
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = 0; c < 6; c++) (Zout + (size_t)(EQ_ORDER[r] * 6 + c) * nw)[iw] = cplx{lu.ar[r][c], lu.ai[r][c]};
        }
    }
    cplx y[6];
#pragma unroll
    for (int r = 0; r < 6; r++) y[r] = x[EQ_ORDER[r]];
    solve6(lu, y);
#pragma unroll
    for (int r = 0; r < 6; r++) x[r] = y[r];          // unknowns keep their order
}

// XiLast storage: LDS rows [12][nxl] for every shape whose bins fit there; the largest shapes
// (more than 1024 bins) keep it in a per-pair global scratch slab instead (coalesced rows).
template <bool XLG>
struct XlStore;
template <>
struct XlStore<false> {
    ldptr p;
    int n;
    __device__ __forceinline__ double get(int row, int iw) const { return p[row * n + iw]; }
    __device__ __forceinline__ void put(int row, int iw, double v) const { p[row * n + iw] = v; }
};
template <>
struct XlStore<true> {
    double *p;
    int n;
    __device__ __forceinline__ double get(int row, int iw) const { return p[(size_t)row * n + iw]; }
    __device__ __forceinline__ void put(int row, int iw, double v) const { p[(size_t)row * n + iw] = v; }
};
// The global XiLast scratch is indexed by a SLOT the workgroup holds while it runs, not by its pair: 2 048 regions of 12 nw
// doubles (39 MB at 200 bins) whatever the batch size, instead of 19.2 KB per pair (192 MB for 10 000 pairs, 3.8 GB for the
// 200 000-pair launches).  A slot is re-used by the workgroups that follow each other on the same XCD -- 128 resident ones x
// 19.2 KB = 2.4 MB of the XCD's 4 MB L2.  Measured (round 5, same box, alternating: scripts/gpu_r5_slots_ab.sh): kernel 2.791-
// 2.800 ms against 2.800-2.809 with the per-pair slab; the L2 <-> fabric traffic does NOT change (FETCH_SIZE 1.323 GB,
// WRITE_SIZE 1.451 GB per launch either way, gpurun_out/r05_slots): the stores of a kernel are written through to the fabric
// whether or not the line stays in the L2 -- 1.451 GB is exactly the XiLast (113 KB), F_lin (19 KB) and Xi (19 KB) stores of
// 10 000 pairs -- so that figure is a property of keeping XiLast off-chip, not of evictions.
// Pools are PER XCD (HW_REG_XCC_ID: the L2s of different XCDs are not coherent with each other, so a slot never changes
// XCD; placement is used for speed only -- any workgroup may run on any XCD), 256 slots each = the most workgroups 32 CUs
// can hold; a full pool (which cannot happen) would be waited for.  Protocol: lane 0 claims a free bit with an atomic OR at
// the L2 (start position hashed from blockIdx: the workgroups of a launch's first round do not collide), the slot travels
// to the other waves through the first LDS word (before anything is staged there); the new owner overwrites every entry
// before it reads it; release = all waves' stores acknowledged (vmcnt(0)), barrier, atomic AND.
#define XL_POOLS 8
#define XL_POOL_WORDS 4
#define XL_SLOTS (XL_POOLS * XL_POOL_WORDS * 64)
__device__ __forceinline__ unsigned xl_slot_acquire(unsigned long long *pools, double *smem0) {
    volatile int *mail = reinterpret_cast<volatile int *>(smem0);
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= (unsigned)(XL_POOLS - 1);
        unsigned long long *pool = pools + xcc * XL_POOL_WORDS;
        const unsigned h = blockIdx.x >> 3;
        unsigned w = (h >> 6) % XL_POOL_WORDS;
        const unsigned rot = h & 63u;
        int slot = -1;
        while (slot < 0) {
            const unsigned long long cur = __hip_atomic_fetch_or(pool + w, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long fr = ~cur;
            const unsigned long long frr = rot ? ((fr >> rot) | (fr << (64u - rot))) : fr;
            if (frr) {
                const unsigned bit = ((unsigned)__builtin_ctzll(frr) + rot) & 63u;
                const unsigned long long m = 1ull << bit;
                const unsigned long long old = __hip_atomic_fetch_or(pool + w, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!(old & m)) slot = (int)((xcc * XL_POOL_WORDS + w) * 64u + bit);
            } else {
                w = (w + 1) % XL_POOL_WORDS;
            }
        }
        *mail = slot;
    }
    __syncthreads();
    const int slot = *mail;
    __syncthreads();
    return (unsigned)__builtin_amdgcn_readfirstlane(slot);
}
__device__ __forceinline__ void xl_slot_release(unsigned long long *pools, unsigned slot) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        (void)__hip_atomic_fetch_and(pools + (slot >> 6), ~(1ull << (slot & 63u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Convergence test of one response entry (raft_model.py:1103): |d| / (|x| + tol) < tol.
// Two IEEE square roots and a division per entry are ~50 instructions; almost every entry is far from the
// threshold, so the test is first decided on |d|^2 against tol^2 (|x| (1 -+ 1e-6) + tol)^2 with the hardware
// approximation of sqrt (v_sqrt_f64, relative error ~2^-23): inside the band between the two bounds (and for
// NaNs) the reference expression itself decides, so every decision is the one that expression makes.
__device__ __forceinline__ bool conv_test(double dr, double di, double xr, double xi, double tol) {
    const double a = fma(dr, dr, di * di), bb = fma(xr, xr, xi * xi);
    const double s = __builtin_amdgcn_sqrt(bb);
    const double lo = tol * fma(s, 1.0 - 1e-6, tol), hi = tol * fma(s, 1.0 + 1e-6, tol);
    const bool sure_ok = a < lo * lo, sure_not = a >= hi * hi;
    bool ok = sure_ok;
    if (!(sure_ok || sure_not)) ok = sqrt(dr * dr + di * di) / (sqrt(xr * xr + xi * xi) + tol) < tol;
    return ok;
}

// ------------------------------------------------------------------ kernels
// workgroup-wide OR / AND of a per-thread predicate
__device__ __forceinline__ int wg_or(int v, bool multi) {
    if (!multi) return __any(v) ? 1 : 0;
    return __syncthreads_or(v);
}
__device__ __forceinline__ int wg_and(int v, bool multi) {
    if (!multi) return __all(v) ? 1 : 0;
    return __syncthreads_and(v);
}

struct PairCtx {
    int pair, d, ic, S;
    cdptr ds;
    ciptr dsi;
    const cplx *__restrict__ cm;
};
__device__ __forceinline__ bool pair_ctx(const DevTables &T, PairCtx &p, int pair) {
    const int npair = T.nDesign * T.nCase;
    p.pair = pair;
    if (p.pair >= npair) return false;
    p.d = p.pair / T.nCase;
    p.ic = p.pair % T.nCase;
    // read-only for the launch: through the constant address space these stay SCALAR loads (and S, the LDS layout and
    // the table pointers scalar registers) even behind the atomics and LDS traffic of a persistent workgroup's claim
    const CONST_AS int64_t *off = (const CONST_AS int64_t *)T.off;
    const int64_t o0 = off[p.d];
    p.S = (int)(off[p.d + 1] - o0);
    p.ds = as_const(T.ds + (size_t)o0 * DS_N);
    p.dsi = as_const(T.dsi + (size_t)o0);
    p.cm = T.cm ? T.cm + (size_t)((const CONST_AS int64_t *)T.cmoff)[p.d] * 2 * T.nw : nullptr;
    return true;
}

template <int NB>
__device__ __forceinline__ void zero6(cplx (&F)[NB][6]) {
#pragma unroll
    for (int j = 0; j < NB; j++)
#pragma unroll
        for (int q = 0; q < 6; q++) F[j][q] = {0.0, 0.0};
}
template <int NB>
__device__ __forceinline__ void store6(cplx *__restrict__ base, int nw, const Bins<NB> &b, const cplx (&F)[NB][6]) {
#pragma unroll
    for (int j = 0; j < NB; j++)
        if (b.act[j]) {
#pragma unroll
            for (int q = 0; q < 6; q++) base[(size_t)q * nw + b.iw[j]] = F[j][q];
        }
}

// F_iner [nDesign,nCase,nHead,6,nw]   (raft_fowt.py:1854-1857,1888); one workgroup per (pair, heading)
template <int NB, int MAXT, int MINB>
__global__ void __launch_bounds__(MAXT, MINB) k_excitation(DevTables T, cplx *__restrict__ F_iner) {
    PairCtx p;
    const int ih = blockIdx.x % T.nHead;
    if (!pair_ctx(T, p, blockIdx.x / T.nHead)) return;
    Bins<NB> b;
    load_bins(T, b, threadIdx.x);
    set_heading_amp(T, b, p.ic, ih);
    const double beta = T.beta[(size_t)p.ic * T.nHead + ih];
    const double cb = to_sgpr(cos(beta)), sb = to_sgpr(sin(beta));
    cplx F[NB][6];
    zero6(F);
    inertial_excitation<NB, true>(T, p.ds, p.dsi, p.S, p.cm, b, p.ic, ih, cb, sb, F);
    store6(F_iner + (((size_t)p.pair * T.nHead + ih) * 6) * T.nw, T.nw, b, F);
}

// One linearisation about a given Xi (raft_fowt.py:1891-1957).
template <int NB, int MAXT, int MINB>
__global__ void __launch_bounds__(MAXT, MINB) k_linearize(DevTables T, const cplx *__restrict__ Xi_in,
                                                           double *__restrict__ B_drag, cplx *__restrict__ F_drag) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    PairCtx p;
    if (!pair_ctx(T, p, pair_of_block(blockIdx.x, T.nDesign * T.nCase))) return;
    const bool multi = blockDim.x > 64;
    constexpr int STAGE = stage_policy(NB, MAXT);
    Lds l = carve(smem, p.S, 0, blockDim.x >> 6, STAGE);        // no XiLast storage in this kernel
    if (STAGE) stage_recA(p.ds, p.dsi, p.S, l, STAGE);
    Bins<NB> b;
    load_bins(T, b, threadIdx.x);
    wg_sync(multi);
    {
        set_heading_amp(T, b, p.ic, 0);
        const double beta = T.beta[(size_t)p.ic * T.nHead];
        const double cb = to_sgpr(cos(beta)), sb = to_sgpr(sin(beta));
        cplx X[NB][6];
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const double w = b.w[j];
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const cplx xi = Xi_in[((size_t)p.pair * 6 + q) * T.nw + b.iw[j]];
                X[j][q] = {w * xi.re, w * xi.im};
            }
        }
#ifdef RAFTX_PHASE_TIMING
        unsigned long long pt_[8], pt_t = 0;
#endif
        linearize_passA<NB, STAGE>(p.ds, p.dsi, p.S, l, b, cb, sb, X PT_PASS);
        wg_sync(multi);
        strip_phase<true>(p.ds, p.dsi, p.S, l, cb, sb, multi);
    }
    if (B_drag && threadIdx.x < 36) B_drag[(size_t)p.pair * 36 + threadIdx.x] = l.Bd[threadIdx.x];
    if (F_drag) {
        for (int ih = 0; ih < T.nHead; ih++) {
            const double beta = T.beta[(size_t)p.ic * T.nHead + ih];
            const double cb = to_sgpr(cos(beta)), sb = to_sgpr(sin(beta));
            load_bins(T, b, opaque((int)threadIdx.x));    // re-derived (L1 / L2 hits): nothing of b stays live -- and spilled -- through pass A
            set_heading_amp(T, b, p.ic, ih);
            if (ih > 0) {
                wg_sync(multi);
                strip_phase<false>(p.ds, p.dsi, p.S, l, cb, sb, multi);
            }
            cplx F[NB][6];
            zero6(F);
            drag_excitation<NB, STAGE>(p.ds, p.dsi, p.S, l, b, cb, sb, F);
            {   // bin indices re-derived for the stores: no per-lane address lives through the sweep
                cplx *__restrict__ base = F_drag + (((size_t)p.pair * T.nHead + ih) * 6) * T.nw;
                const int t_ = opaque((int)threadIdx.x);
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    const int iw = j * (int)blockDim.x + t_;
                    if (iw < T.nw) {
#pragma unroll
                        for (int q = 0; q < 6; q++) base[(size_t)q * T.nw + iw] = F[j][q];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ per-strip by-products on request
// What the reference keeps on its Member objects and the fused kernels never materialise: wave kinematics per strip
// (raft_member.py:1927-1937, helpers.py:188-236) and the linearised drag matrix / local drag excitation per strip
// (raft_member.py:2075-2123), for ONE design and sea state -- raftx_strip_kinematics / raftx_strip_drag.  One workgroup
// per strip, lanes over the bins, every bin evaluated directly (no rotors: this is an export, 112 B per strip, heading
// and bin of stores).
__device__ inline void strip_wave_kin(const DevTables &T, double x, double y, double z, double cb, double sb, int ic, int ih,
                                      int iw, cplx (&u)[3], cplx &p) {
    const double k = T.k[iw], w = T.w[iw];
    const double z0 = T.zeta[((size_t)ic * T.nHead + ih) * T.nw + iw];
    double s_, c_;
    fast_sincos(-(k * (cb * x + sb * y)), s_, c_);                          // e^{-i k xi} (helpers.py:201)
    const bool k0 = k == 0.0;                                               // :211-214  Sh = 1, Ch = Cc = 99999
    const double P = k0 ? 50000.0 : fast_exp(k * z), Qk = k0 ? 49999.0 : fast_exp(-(k * (z + 2.0 * T.depth)));
    const double Qv = depth_mode(k, T.depth) == 1 ? 0.0 : Qk;              // :215-218 deep water: Sh = Ch = e^{kz}
    const double c1 = w * z0 * T.csh[iw];
    const double ar = c1 * c_, ai = c1 * s_, ps = P + Qv, pd = P - Qv;
    u[0] = {cb * (ar * ps), cb * (ai * ps)};                                // :225-227
    u[1] = {sb * (ar * ps), sb * (ai * ps)};
    u[2] = {-(ai * pd), ar * pd};
    const double sp = T.rho * T.g * z0 * T.cch[iw] * (P + Qk);              // :231 (the deep-water branch keeps both exponentials)
    p = {sp * c_, sp * s_};
}
__global__ void __launch_bounds__(256) k_strip_kinematics(DevTables T, int d, int ic, cplx *__restrict__ u_out,
                                                          cplx *__restrict__ ud_out, cplx *__restrict__ p_out) {
    const int S = (int)(T.off[d + 1] - T.off[d]);
    const int s = blockIdx.x % S, ih = blockIdx.x / S, nw = T.nw;
    const double *rec = T.ds + ((size_t)T.off[d] + s) * DS_N;
    const double beta = T.beta[(size_t)ic * T.nHead + ih], cb = cos(beta), sb = sin(beta);
    const double x = rec[DS_X], y = rec[DS_X + 1], z = rec[DS_X + 2];
    for (int iw = threadIdx.x; iw < nw; iw += blockDim.x) {
        cplx u[3], p;
        strip_wave_kin(T, x, y, z, cb, sb, ic, ih, iw, u, p);
        const double w = T.w[iw];
        const size_t o = ((size_t)ih * S + s) * 3;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (u_out) u_out[(o + j) * nw + iw] = u[j];
            if (ud_out) ud_out[(o + j) * nw + iw] = {-(w * u[j].im), w * u[j].re};      // ud = i w u (:229)
        }
        if (p_out) p_out[((size_t)ih * S + s) * nw + iw] = p;
    }
}
__global__ void __launch_bounds__(256) k_strip_drag(DevTables T, int d, int ic, int ih, const cplx *__restrict__ Xi,
                                                    double *__restrict__ Bmat, cplx *__restrict__ Fexc) {
    __shared__ double red[3][256];
    __shared__ double Bm[9];
    const int s = blockIdx.x, nw = T.nw, tid = threadIdx.x;
    const size_t row = (size_t)T.off[d] + s;
    const double *rec = T.ds + row * DS_N;
    const bool circ = (T.dsi[row] & DSI_CIRC) != 0;
    const double x = rec[DS_X], y = rec[DS_X + 1], z = rec[DS_X + 2];
    const double ax = rec[DS_A], ay = rec[DS_A + 1], az = rec[DS_A + 2];
    double n[3][3];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int j = 0; j < 3; j++) n[c][j] = rec[DS_Q + 3 * c + j];
    {   // sums of the squared relative-velocity components over the bins, heading 0 (raft_fowt.py:1910)
        const double beta = T.beta[(size_t)ic * T.nHead], cb = cos(beta), sb = sin(beta);
        double a = 0, c1 = 0, c2 = 0;
        for (int iw = tid; iw < nw; iw += blockDim.x) {
            cplx u[3], p;
            strip_wave_kin(T, x, y, z, cb, sb, ic, 0, iw, u, p);
            const double w = T.w[iw];
            cplx X[6];
#pragma unroll
            for (int q = 0; q < 6; q++) X[q] = Xi[(size_t)q * nw + iw];
            // node displacement Xi_t + theta x arm (helpers.py:178-181,396-402), velocity i w (.)
            const cplx dr[3] = {{X[0].re + (-X[5].re * ay + X[4].re * az), X[0].im + (-X[5].im * ay + X[4].im * az)},
                                {X[1].re + (X[5].re * ax - X[3].re * az), X[1].im + (X[5].im * ax - X[3].im * az)},
                                {X[2].re + (-X[4].re * ax + X[3].re * ay), X[2].im + (-X[4].im * ax + X[3].im * ay)}};
            cplx v[3];
#pragma unroll
            for (int j = 0; j < 3; j++) v[j] = {u[j].re + w * dr[j].im, u[j].im - w * dr[j].re};     // u - i w dr (:2075)
            double sq[3], vv = 0;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double pr = v[0].re * n[c][0] + v[1].re * n[c][1] + v[2].re * n[c][2];
                const double pi = v[0].im * n[c][0] + v[1].im * n[c][1] + v[2].im * n[c][2];
                sq[c] = pr * pr + pi * pi;
            }
#pragma unroll
            for (int j = 0; j < 3; j++) vv += v[j].re * v[j].re + v[j].im * v[j].im;
            a += sq[0];
            c1 += circ ? vv - sq[0] : sq[1];                                 // :2084-2090
            c2 += sq[2];
        }
        red[0][tid] = a; red[1][tid] = c1; red[2][tid] = c2;
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) {
            if (tid < h)
#pragma unroll
                for (int r = 0; r < 3; r++) red[r][tid] += red[r][tid + h];
            __syncthreads();
        }
        if (tid < 9) {
            const double vRq = sqrt(0.5 * red[0][0]);
            const double vR1 = circ ? sqrt(0.5 * fmax(red[1][0], 0.0)) : sqrt(0.5 * red[1][0]);
            const double vR2 = circ ? vR1 : sqrt(0.5 * red[2][0]);
            const double bc[3] = {rec[DS_DQ] * vRq + rec[DS_DQ + 3] * vRq, rec[DS_DQ + 1] * vR1, rec[DS_DQ + 2] * vR2};   // :2093-2110
            const int i = tid / 3, j = tid % 3;
            const double m = bc[0] * n[0][i] * n[0][j] + bc[1] * n[1][i] * n[1][j] + bc[2] * n[2][i] * n[2][j];
            Bm[tid] = m;
            if (Bmat) Bmat[(size_t)s * 9 + tid] = m;
        }
        __syncthreads();
    }
    if (!Fexc) return;
    const double beta = T.beta[(size_t)ic * T.nHead + ih], cb = cos(beta), sb = sin(beta);
    for (int iw = tid; iw < nw; iw += blockDim.x) {
        cplx u[3], p;
        strip_wave_kin(T, x, y, z, cb, sb, ic, ih, iw, u, p);
#pragma unroll
        for (int a = 0; a < 3; a++)
            Fexc[((size_t)s * 3 + a) * nw + iw] = {Bm[a * 3] * u[0].re + Bm[a * 3 + 1] * u[1].re + Bm[a * 3 + 2] * u[2].re,
                                                   Bm[a * 3] * u[0].im + Bm[a * 3 + 1] * u[1].im + Bm[a * 3 + 2] * u[2].im};
    }
}

struct SolveArgs {
    int nIter;          // loop bound = YAML nIter + 1 (raft_model.py:977)
    double tol, XiStart;
    const cplx *__restrict__ F_extra;    // [pair,nHead,6,nw] or null
    cplx *Xi;               // [pair,nHead,6,nw]
    int *__restrict__ niter, *__restrict__ flags;     // [pair]
    double *__restrict__ B_drag;         // [pair,36] or null
    cplx *__restrict__ F_wave;           // [pair,nHead,6,nw] or null
    cplx *__restrict__ Z;                // [pair,36,nw] or null
    double *Xl;                          // [XL_SLOTS,12,nw] XiLast scratch of the shapes that keep it in global memory (xl_global)
    unsigned long long *slots;           // [XL_POOLS * XL_POOL_WORDS] bit per slot of Xl: held by a running workgroup
    const cplx *__restrict__ Xl0;        // [pair,6,nw] or null: initial linearisation point instead of XiStart
    cplx *__restrict__ XlOut;            // [pair,6,nw] or null: linearisation point of the LAST iteration
    unsigned long long *dbg;             // RAFTX_PHASE_TIMING builds: [8] accumulated wave-0 cycles per phase
    const int *__restrict__ pairs;       // [npairs] the pairs of this launch, or null: all pairs 0 .. nDesign * nCase - 1
    int npairs;                          // (batches with very unequal strip counts are launched per LDS class)
    int rc_n;                            // slots of the LDS run-start cache (0: none)
};

// The fused fixed point (raft_model.py:1052-1142) + per-heading response (:1189-1236).
// Storage plan: XiLast lives in LDS (read once per pass A, once per convergence test) -- or, for the shapes of xl_global(), in a
// global scratch region the workgroup holds while it runs (read and written once per iteration).
// F_lin (raft_model.py:1048) is parked in the pair's own heading-0 slab of the Xi OUTPUT
// buffer until the final iterate overwrites it -- each lane re-reads only what it wrote,
// so no extra HBM footprint and no synchronisation is needed.  The 6x6 systems of a lane's
// NB bins are factorised one after the other.
#define KP_STASH 4        // doubles at the start of the dynamic LDS that a persistent workgroup keeps for itself (see RAFTX_KP_REENTER)
// solve_pair: one (design, sea state) pair by the calling workgroup.  `idx` = position in the launch's pair list (or
// the pair itself without one); PERSIST: the workgroup belongs to a persistent grid (k_solve_dynamics_p) and keeps the
// XiLast region `xslot` for its whole life instead of claiming one per pair.
template <int NB, int FLAGS, int MAXT, bool PERSIST>
__device__ __forceinline__ void solve_pair(const DevTables &T, const SolveArgs &A, int idx, unsigned xslot) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr bool FDEP = (FLAGS & KF_FDEP) != 0, OUTZ = (FLAGS & KF_OUTZ) != 0, OUTF = (FLAGS & KF_OUTF) != 0;
    constexpr bool EXTRA = (FLAGS & KF_EXTRA) != 0, MCF = (FLAGS & KF_MCF) != 0, MULTI = (FLAGS & KF_MULTI) != 0;
    constexpr bool XLIO = (FLAGS & KF_XLIO) != 0;
    PairCtx p;
    {
        const int nl = A.pairs ? A.npairs : T.nDesign * T.nCase;
        if (idx >= nl) return;
        if (!pair_ctx(T, p, A.pairs ? A.pairs[idx] : idx)) return;
    }
    PT_DECL;
    PT_BEGIN(A);
    const bool multi = blockDim.x > 64;
    const int nw = T.nw, nHs = T.nHead, nH = MULTI ? T.nHead : 1;
    const int pair = p.pair, S = p.S;
    const cplx *cm = (MCF && T.cm) ? p.cm : nullptr;
    constexpr int STAGE = stage_policy(NB, MAXT);
    constexpr bool XLG = xl_global(NB, MAXT);
    constexpr int PARK = park_policy(NB, MAXT);
    constexpr bool RC = PARK != 0 && (XLG || RAFTX_XL_LDS);   // the shape whose spare LDS is a run-start cache (A.rc_n slots)
    Lds l = carve(smem + (PERSIST ? KP_STASH : 0), S, XLG ? 0 : nw, blockDim.x >> 6, STAGE, PARK, RC ? A.rc_n : 0, RC ? nw : 0);
    XlStore<XLG> xl;
    if constexpr (XLG) {
#ifdef RAFTX_XL_PER_PAIR                                   // tuning build: round 1-4's form, a region per pair
        xl.p = A.Xl + (size_t)pair * 12 * nw;
#else
        if constexpr (PERSIST) xl.p = A.Xl + (size_t)xslot * 12 * nw;
        else xl.p = A.Xl + (size_t)xl_slot_acquire(A.slots, smem) * 12 * nw;  // before anything is staged in LDS
#endif
        xl.n = nw;
    }
    if (STAGE) stage_recA(p.ds, p.dsi, S, l, STAGE);
    if constexpr (RC) stage_sincos_table(l.sct);
    if constexpr (XLG) {
    } else {
        xl.p = l.xl;
        xl.n = l.nxl;
    }
    Bins<NB> b;
    load_bins(T, b, threadIdx.x);
    set_heading_amp(T, b, p.ic, 0);
    if (threadIdx.x < 36) {                               // (three plain loads: a pointer chosen by index would put T in memory)
        const size_t e = (size_t)p.d * 36 + threadIdx.x;
        l.mat[threadIdx.x] = T.M0[e];
        l.mat[36 + threadIdx.x] = T.B0[e];
        l.mat[72 + threadIdx.x] = T.C0[e];
    }
    const double beta0 = T.beta[(size_t)p.ic * nHs];
    const double cb0 = to_sgpr(cos(beta0)), sb0 = to_sgpr(sin(beta0));
    cplx *xio = A.Xi + ((size_t)pair * nHs) * 6 * nw;      // heading-0 slab: F_lin until the end, then Xi

    {   // F_lin = F_extra[0] + F_iner[0]   (raft_model.py:1048)
        cplx Flin[NB][6];
        zero6(Flin);
        if constexpr (EXTRA) {
            if (A.F_extra) {
#pragma unroll
                for (int j = 0; j < NB; j++)
#pragma unroll
                    for (int q = 0; q < 6; q++)
                        Flin[j][q] = A.F_extra[(((size_t)pair * nHs) * 6 + q) * nw + b.iw[j]];
            }
        }
        if constexpr (MCF)
            inertial_excitation_uv<NB, RC, true>(T, p.ds, p.dsi, S, l, b, p.ic, 0, cb0, sb0, Flin, multi, cm);
        else
            inertial_excitation_uv<NB, RC>(T, p.ds, p.dsi, S, l, b, p.ic, 0, cb0, sb0, Flin, multi);
        store6(xio, nw, b, Flin);
    }
    // XiLast <- XiStart (:999), kept as xl[2q][bin] = re, xl[2q+1][bin] = im
#pragma unroll
    for (int j = 0; j < NB; j++)
        if (b.act[j]) {
#pragma unroll
            for (int q = 0; q < 6; q++) {
                cplx x0 = {A.XiStart, 0.0};
                if constexpr (XLIO) {
                    if (A.Xl0) x0 = A.Xl0[((size_t)pair * 6 + q) * nw + b.iw[j]];
                }
                xl.put(2 * q, b.iw[j], x0.re);
                xl.put(2 * q + 1, b.iw[j], x0.im);
            }
        }
    const MbRsrc mb = mb_rsrc((FDEP && T.MBw) ? T.MBw + (size_t)p.d * 72 * nw : nullptr, nw);
    cplx *Zout = (OUTZ && A.Z) ? A.Z + (size_t)pair * 36 * nw : nullptr;
    wg_sync(multi);
    PT_MARK(0);   // set-up + inertial excitation

    int iiter = 0, done = 0, converged = 0, nan = 0;
    // DEFER (the parked two-bin shape with XiLast in its global slab): convergence test and relaxation of BOTH bins
    // follow the second solve -- one round trip to the slab per iteration, its 24 loads in flight together while
    // the registers of the 6x6 systems are dead -- and w * XiLast of the next linearisation stays in registers (Xc)
    // instead of being fetched again at the top of the loop.
    constexpr bool DEFER = PARK != 0 && (XLG || RAFTX_XL_LDS);
    cplx Xc[DEFER ? NB : 1][6];
    if constexpr (DEFER) {
#pragma unroll
        for (int j = 0; j < NB; j++)
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const double vr = xl.get(2 * q, b.iw[j]), vi = xl.get(2 * q + 1, b.iw[j]);      // what this lane just stored
                Xc[j][q].re = b.act[j] ? b.w[j] * vr : 0.0;
                Xc[j][q].im = b.act[j] ? b.w[j] * vi : 0.0;
            }
    }
#pragma unroll 1
    while (true) {
        // The per-bin constants are re-read (L1/L2 hits) at the top of every iteration from an
        // opaque thread id, so that they do not stay live -- and get spilled -- across the solve.
        load_bins(T, b, opaque((int)threadIdx.x));
        set_heading_amp(T, b, p.ic, 0);
        if constexpr (DEFER) {
            linearize_passA<NB, STAGE, RC>(p.ds, p.dsi, S, l, b, cb0, sb0, Xc PT_PASS);         // :1063
        } else {
            cplx X[NB][6];
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const double w = b.w[j];
                const int iw = opaque(b.iw[j]);
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const double vr = xl.get(2 * q, iw), vi = xl.get(2 * q + 1, iw);
                    X[j][q].re = b.act[j] ? w * vr : 0.0;          // unconditional loads (clamped bin), then select
                    X[j][q].im = b.act[j] ? w * vi : 0.0;
                }
            }
            PT_MARK(7);   // XiLast fetch
            linearize_passA<NB, STAGE, RC>(p.ds, p.dsi, S, l, b, cb0, sb0, X PT_PASS);      // :1063
        }
        wg_sync(multi);
        PT_MARK(1);   // pass A
        strip_phase<true>(p.ds, p.dsi, S, l, cb0, sb0, multi);
        PT_MARK(2);   // strip-lane phase
        cplx x[NB][6];
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int iw = opaque(b.iw[j]);
#pragma unroll
            for (int q = 0; q < 6; q++) x[j][q] = xio[(size_t)q * nw + iw];   // F_lin (clamped bin when inactive)
        }
        drag_excitation<NB, STAGE, RC>(p.ds, p.dsi, S, l, b, cb0, sb0, x);           // + F_drag (:1064,:1081)
        PT_MARK(3);   // pass B
        int bad = 0, ok = 1;
        const bool last_chance = iiter + 1 >= A.nIter;
        const int tid_s = opaque((int)threadIdx.x);                  // bin bookkeeping re-derived: nothing of b stays live here
        // one bin: total excitation out, Z assembly + 6x6 solve (:1086-1089), NaN check (:1098), convergence
        // (:1103-1104) and relaxation (:1133)
        auto solve_bin = [&](int j, cplx (&xx)[6]) {
            const int ib = j * blockDim.x + tid_s;
            const bool act = ib < nw;
            const int iw = act ? ib : 0;
            const double wl = T.w[iw];
            const double w = act ? wl : 0.0;
            if constexpr (OUTF) {                                     // total excitation, heading 0 (:1212)
                if (act && A.F_wave) {
#pragma unroll
                    for (int q = 0; q < 6; q++) A.F_wave[(((size_t)pair * nHs) * 6 + q) * nw + iw] = xx[q];
                }
            }
            assemble_and_solve<FLAGS>(l, mb, nw, iw, w, xx, Zout, act);
            if constexpr (DEFER) return;
            if (act) {
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const double lr = xl.get(2 * q, iw), li = xl.get(2 * q + 1, iw);
                    if constexpr (XLIO) {
                        if (A.XlOut) A.XlOut[((size_t)pair * 6 + q) * nw + iw] = cplx{lr, li};
                    }
                    if (isnan(xx[q].re) || isnan(xx[q].im)) bad = 1;
                    const double dr = xx[q].re - lr, di = xx[q].im - li;
                    if (!conv_test(dr, di, xx[q].re, xx[q].im, A.tol)) ok = 0;
                    xl.put(2 * q, iw, 0.2 * lr + 0.8 * xx[q].re);
                    xl.put(2 * q + 1, iw, 0.2 * li + 0.8 * xx[q].im);
                }
            }
        };
        if constexpr (PARK != 0) {
            // Two bins per lane: while one bin's 6x6 complex system (72 doubles) is being factorised, the other
            // bin's right-hand side / solution waits in a per-lane LDS column carved out of the sweep buffers
            // (vsq rows >= 1 | uv | tile are dead between pass B and the next strip phase; vsq row 0 holds b_c) instead of being spilled to scratch.
            wg_sync(multi);                                           // the other wave may still read uv / vsq in pass B
            ldptr park = l.park + tid_s;
            const int nt = blockDim.x;
#pragma unroll
            for (int q = 0; q < 6; q++) {
                park[(2 * q) * nt] = x[1][q].re;
                park[(2 * q + 1) * nt] = x[1][q].im;
            }
            solve_bin(0, x[0]);
            ldptr park2 = l.park + opaque(tid_s);                       // opaque: no store-to-load forwarding back into VGPRs
            cplx r1[6];
#pragma unroll
            for (int q = 0; q < 6; q++) {
                r1[q] = cplx{park2[(2 * q) * nt], park2[(2 * q + 1) * nt]};
                park2[(2 * q) * nt] = x[0][q].re;
                park2[(2 * q + 1) * nt] = x[0][q].im;
            }
            solve_bin(1, r1);
            if constexpr (DEFER) {
                // NaN check (:1098), convergence (:1103-1104) and relaxation (:1133) of both bins; bin 0's solution comes
                // back from its parking column
                ldptr pk = l.park + opaque(tid_s);
                double lr[NB][6], li[NB][6];
                int iwj[NB];
                bool actj[NB];
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    const int ib = j * nt + tid_s;
                    actj[j] = ib < nw;
                    iwj[j] = actj[j] ? ib : 0;
#pragma unroll
                    for (int q = 0; q < 6; q++) {
                        lr[j][q] = xl.get(2 * q, iwj[j]);
                        li[j][q] = xl.get(2 * q + 1, iwj[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    const double wl = T.w[iwj[j]];
#pragma unroll
                    for (int q = 0; q < 6; q++) {
                        const cplx xx = j == 0 ? cplx{pk[(2 * q) * nt], pk[(2 * q + 1) * nt]} : r1[q];
                        if constexpr (XLIO) {
                            if (actj[j] && A.XlOut) A.XlOut[((size_t)pair * 6 + q) * nw + iwj[j]] = cplx{lr[j][q], li[j][q]};
                        }
                        if (actj[j] && (isnan(xx.re) || isnan(xx.im))) bad = 1;
                        // the vote is an AND over the workgroup: once a lane of this wave has failed, the remaining
                        // tests of the wave cannot change it (every iteration but the last one of a pair ends here)
                        if (__all(ok)) {
                            const double dr = xx.re - lr[j][q], di = xx.im - li[j][q];
                            if (actj[j] && !conv_test(dr, di, xx.re, xx.im, A.tol)) ok = 0;
                        }
                        const double nr = 0.2 * lr[j][q] + 0.8 * xx.re, ni = 0.2 * li[j][q] + 0.8 * xx.im;
                        if (actj[j]) {
                            xl.put(2 * q, iwj[j], nr);
                            xl.put(2 * q + 1, iwj[j], ni);
                        }
                        Xc[j][q].re = actj[j] ? wl * nr : 0.0;
                        Xc[j][q].im = actj[j] ? wl * ni : 0.0;
                    }
                }
            }
            PT_MARK(4);   // assemble + solve + convergence
            done = iiter + 1;
            nan = wg_or(bad, multi);
            converged = nan ? 0 : wg_and(ok, multi);
            if (nan || converged || last_chance) {
                // Heading 0 of the final response, Zinv (F_lin + F_drag(0)), is exactly this solve (:1216)
                ldptr park3 = l.park + opaque(tid_s);
                if (tid_s < nw) {
#pragma unroll
                    for (int q = 0; q < 6; q++)
                        xio[(size_t)q * nw + tid_s] = nan ? cplx{NAN, NAN} : cplx{park3[(2 * q) * nt], park3[(2 * q + 1) * nt]};
                }
                if (nt + tid_s < nw) {
#pragma unroll
                    for (int q = 0; q < 6; q++) xio[(size_t)q * nw + nt + tid_s] = nan ? cplx{NAN, NAN} : r1[q];
                }
                break;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NB; j++) solve_bin(j, x[j]);
            PT_MARK(4);   // assemble + solve + convergence
            done = iiter + 1;
            nan = wg_or(bad, multi);
            converged = nan ? 0 : wg_and(ok, multi);
            if (nan || converged || last_chance) {
                // Heading 0 of the final response, Zinv (F_lin + F_drag(0)), is exactly this solve (:1216)
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    const int ib = j * blockDim.x + tid_s;
                    if (ib < nw) {
#pragma unroll
                        for (int q = 0; q < 6; q++) xio[(size_t)q * nw + ib] = nan ? cplx{NAN, NAN} : x[j][q];
                    }
                }
                break;
            }
        }
        iiter++;
        wg_sync(multi);
    }
#ifndef RAFTX_XL_PER_PAIR
    if constexpr (XLG && !PERSIST) xl_slot_release(A.slots, (unsigned)((size_t)(xl.p - A.Xl) / ((size_t)12 * nw)));    // XiLast is dead from here on
#endif

    // remaining headings: same impedance, same linearised coefficients (:1200-1236)
    if constexpr (MULTI) {
#pragma unroll 1
        for (int ih = 1; ih < nH; ih++) {
            const double beta = T.beta[(size_t)p.ic * nHs + ih];
            const double cb = to_sgpr(cos(beta)), sb = to_sgpr(sin(beta));
            load_bins(T, b, opaque((int)threadIdx.x));    // re-derived, as at the top of every iteration: nothing of b lives through the fixed point
            set_heading_amp(T, b, p.ic, ih);
            cplx x[NB][6];
            zero6(x);
            if constexpr (EXTRA) {
                if (A.F_extra) {
#pragma unroll
                    for (int j = 0; j < NB; j++)
#pragma unroll
                        for (int q = 0; q < 6; q++)
                            x[j][q] = A.F_extra[(((size_t)pair * nHs + ih) * 6 + q) * nw + b.iw[j]];
                }
            }
            wg_sync(multi);                               // the uv rows are free (previous heading's pass B is over)
            // the inertial sweep refills the run-start cache for this heading
            if constexpr (MCF)
                inertial_excitation_uv<NB, RC, true>(T, p.ds, p.dsi, S, l, b, p.ic, ih, cb, sb, x, multi, cm);
            else
                inertial_excitation_uv<NB, RC>(T, p.ds, p.dsi, S, l, b, p.ic, ih, cb, sb, x, multi);      // U', V' through the uv rows
            wg_sync(multi);
            strip_phase<false>(p.ds, p.dsi, S, l, cb, sb, multi);
            drag_excitation<NB, STAGE, RC>(p.ds, p.dsi, S, l, b, cb, sb, x);           // :1209,:1212
            cplx *xo = A.Xi + ((size_t)pair * nHs + ih) * 6 * nw;
            if constexpr (OUTF) {
#pragma unroll
                for (int j = 0; j < NB; j++)
                    if (b.act[j] && A.F_wave) {
#pragma unroll
                        for (int q = 0; q < 6; q++) A.F_wave[(((size_t)pair * nHs + ih) * 6 + q) * nw + b.iw[j]] = x[j][q];
                    }
            }
            // While one bin's system is being factorised the right-hand sides of the later bins wait in their own
            // places of the output slab (written and read back by the same lane), not in registers.
#pragma unroll
            for (int j = 1; j < NB; j++)
                if (b.act[j]) {
#pragma unroll
                    for (int q = 0; q < 6; q++) xo[(size_t)q * nw + b.iw[j]] = x[j][q];
                }
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const int ib = j * blockDim.x + opaque((int)threadIdx.x);
                const bool act = ib < nw;
                const int iw = act ? ib : 0;
                const double wl = T.w[iw];
                cplx y[6];
#pragma unroll
                for (int q = 0; q < 6; q++) y[q] = j == 0 ? x[0][q] : xo[(size_t)q * nw + iw];
                assemble_and_solve<(FLAGS & ~KF_OUTZ)>(l, mb, nw, iw, act ? wl : 0.0, y, nullptr, act);   // Zinv @ F_wave (:1216)
                if (act) {
#pragma unroll
                    for (int q = 0; q < 6; q++) xo[(size_t)q * nw + iw] = nan ? cplx{NAN, NAN} : y[q];
                }
            }
        }
    }
    PT_MARK(5);       // vote, final stores, other headings
    PT_FLUSH(A);
    if (threadIdx.x < 36 && A.B_drag) A.B_drag[(size_t)pair * 36 + threadIdx.x] = l.Bd[threadIdx.x];
    if (threadIdx.x == 0) {
        if (A.niter) A.niter[pair] = done;
        if (A.flags) A.flags[pair] = (converged ? RAFTX_FLAG_CONVERGED : 0) | (nan ? RAFTX_FLAG_NAN : 0);
    }
}

template <int NB, int FLAGS, int MAXT, int MINB>
__global__ void __launch_bounds__(MAXT, MINB) k_solve_dynamics(DevTables T, SolveArgs A) {
    const int nl = A.pairs ? A.npairs : T.nDesign * T.nCase;
    solve_pair<NB, FLAGS, MAXT, false>(T, A, pair_of_block(blockIdx.x, nl), 0u);
}

// ------------------------------------------------------------------ persistent form of the fused fixed point
// The grid is what the chip holds at once (host: workgroups per CU x CUs); a workgroup CLAIMS positions of the launch's
// pair list until none is left, instead of being dispatched once per pair: no dispatch gap between two pairs of a
// slot (~10-15 us of 257 at C3), the XiLast region is the workgroup's own for the launch (no slot pool traffic), and a
// launch on another queue starts on the slots this one's workgroups leave when they run dry (the host alternates two
// queues: the drain of batch i is the ramp of batch i+1).
// Claiming: eight counters (one per XCD slab of the list, as pair_of_block deals them), a workgroup takes from the slab
// of the XCD it runs on and, when that one is dry, from the others in turn -- placement is for L2 locality only.
// Registers: the kernel sits at the 256-VGPR wall with ~120 spilled SGPRs; kernel arguments that stay live around the
// loop tip it into scratch.  They are therefore RE-READ from the kernarg segment for every pair behind an opaque copy
// of the segment pointer (scalar loads from the constant cache, exactly what the prologue of the one-pair kernel does).
#define KP_CTR_STRIDE 32                 // unsigned: each slab's counter on a 128-byte line of its own
struct PersistArgs {
    DevTables T;
    SolveArgs A;
    unsigned *ctr;          // [9][KP_CTR_STRIDE] claimed positions per XCD slab + workgroups that have left; all zero when the
                            // launch starts, zeroed again by the last workgroup to leave
    unsigned xl_base;       // first XiLast region of this launch (launches in flight at once use disjoint ranges)
};
// a persistent workgroup has found the pair list dry
__device__ __forceinline__ void kp_leave(const PersistArgs &P) {
    if (threadIdx.x == 0) {
        unsigned *left = P.ctr + 8 * KP_CTR_STRIDE;
        const unsigned n = __hip_atomic_fetch_add(left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n + 1 == gridDim.x) {                        // the last one out: every claim has been made, the set is clean for its next launch
            for (int x = 0; x < 9; x++) __hip_atomic_store(P.ctr + x * KP_CTR_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__device__ __forceinline__ int claim_pair(unsigned *ctr, int nl, LDS_AS int *mail) {
    __syncthreads();                                    // the previous pair's last LDS reads are over
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        const int per = (nl + 7) >> 3;
        int idx = -1;
        for (unsigned t = 0; t < 8u && idx < 0; t++) {
            const unsigned x = (xcc + t) & 7u;
            const int lim = min(per, nl - (int)x * per);                  // positions of slab x
            if (lim <= 0) continue;
            unsigned *cx = ctr + x * KP_CTR_STRIDE;
            // a dry slab costs one plain load, not an atomic that keeps counting
            if ((int)__hip_atomic_load(cx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= lim) continue;
            const int i = (int)__hip_atomic_fetch_add(cx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i < lim) idx = (int)x * per + i;
        }
        *mail = idx;
    }
    __syncthreads();
    const int idx = *mail;
    __syncthreads();                                    // the word is staged over by the pair's set-up
    return __builtin_amdgcn_readfirstlane(idx);
}
// The pair loop is NOT a loop for the compiler (as one, every lane-index-derived address of the 16 000-instruction body is
// hoisted in front of it and spilled: 168 B of scratch).  A workgroup that has finished a pair RE-ENTERS the kernel at its
// first instruction with the registers a fresh dispatch would find: nothing is live across pairs, the code is the
// one-pair kernel's.  What a dispatch hands over besides the lane ids -- queue pointer, kernarg segment pointer,
// workgroup id -- is stashed in the first 32 bytes of the dynamic LDS at entry (re-stashed, unchanged, on every
// re-entry) and read back from there at the end, so that it occupies no register in between (the kernel sits at 256
// VGPRs with ~120 SGPRs spilled to lanes; eight more live scalars cost 72 B of scratch).
// Entry state of these kernels with this compiler: user SGPRs s[0:1] = queue pointer, s[2:3] = kernarg segment,
// system SGPRs s4..s6 = workgroup id x, y, z (y = z = 0: the host launches one-dimensional grids), v0 = item ids
// (x | y << 10 | z << 20 = x for one-dimensional blocks), EXEC = all lanes, no private segment.
// tests/test_code_object.py reads exactly that back from the kernel descriptors of the built library, and that no
// persistent specialisation has scratch.  The kernels carry C names (the branch target is spelled in the assembly):
// raftx_kp_f<FLAGS> for the lean 2 x 128 specialisations, FLAGS as a decimal literal.
// Stash layout (dwords from the dynamic LDS base): 0 claim mail, 1 workgroup id x, 2-3 queue pointer, 4-5 kernarg pointer.
#define RAFTX_PERSIST128(X) X(0) X(1) X(16) X(32) X(9) X(17) X(33) X(48) X(41) X(49) X(4) X(36)
#define RAFTX_KP_REENTER(NAME)                                                                                          \
    do {                                                                                                                \
        asm volatile("s_mov_b64 exec, -1\n\t"                                                                           \
                     "v_mov_b32 v1, %1\n\t"                                                                             \
                     "v_mov_b32 v0, %0\n\t"                                                                             \
                     "ds_read_b128 v[2:5], v1\n\t"                                                                      \
                     "ds_read_b64 v[6:7], v1 offset:16\n\t"                                                             \
                     "s_waitcnt lgkmcnt(0)\n\t"                                                                         \
                     "v_readfirstlane_b32 s4, v3\n\t"                                                                   \
                     "v_readfirstlane_b32 s0, v4\n\t"                                                                   \
                     "v_readfirstlane_b32 s1, v5\n\t"                                                                   \
                     "v_readfirstlane_b32 s2, v6\n\t"                                                                   \
                     "v_readfirstlane_b32 s3, v7\n\t"                                                                   \
                     "s_mov_b32 s5, 0\n\t"                                                                              \
                     "s_mov_b32 s6, 0\n\t"                                                                              \
                     "s_getpc_b64 s[8:9]\n\t"                                                                           \
                     "s_add_u32 s8, s8, " #NAME "@rel32@lo+4\n\t"                                                       \
                     "s_addc_u32 s9, s9, " #NAME "@rel32@hi+12\n\t"                                                     \
                     "s_setpc_b64 s[8:9]"                                                                               \
                     :                                                                                                  \
                     : "v"(threadIdx.x), "v"((unsigned)(size_t)(LDS_AS double *)smem)                                   \
                     : "memory", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "s0", "s1", "s2", "s3", "s4", "s5",    \
                       "s6", "s8", "s9");                                                                               \
        __builtin_unreachable();                                                                                        \
    } while (0)
#define RAFTX_KP_DEFINE(NAME, NB, FLAGS, MAXT, MINB)                                                                    \
    extern "C" __global__ void __launch_bounds__(MAXT, MINB) NAME(PersistArgs P) {                                      \
        extern __shared__ __attribute__((aligned(16))) double smem[];                                                   \
        if (threadIdx.x == 0) {                                                                                         \
            LDS_AS unsigned *st_ = (LDS_AS unsigned *)smem;                                                             \
            const size_t qp_ = (size_t)__builtin_amdgcn_queue_ptr(), ka_ = (size_t)__builtin_amdgcn_kernarg_segment_ptr(); \
            st_[1] = blockIdx.x + 0u * (blockIdx.y + blockIdx.z + threadIdx.y + threadIdx.z);                           \
            st_[2] = (unsigned)qp_;                                                                                     \
            st_[3] = (unsigned)(qp_ >> 32);                                                                             \
            st_[4] = (unsigned)ka_;                                                                                     \
            st_[5] = (unsigned)(ka_ >> 32);                                                                             \
        }                                                                                                               \
        const int nl = P.A.pairs ? P.A.npairs : P.T.nDesign * P.T.nCase;                                                \
        const int idx = claim_pair(P.ctr, nl, (LDS_AS int *)smem);                                                      \
        if (idx < 0) {                                                                                                  \
            kp_leave(P);                                                                                                \
            return;                                                                                                     \
        }                                                                                                               \
        solve_pair<NB, FLAGS, MAXT, true>(P.T, P.A, idx, P.xl_base + blockIdx.x);                                       \
        RAFTX_KP_REENTER(NAME);                                                                                         \
    }
#ifndef RAFTX_KP_MINB
#define RAFTX_KP_MINB 2
#endif
#define X(F) RAFTX_KP_DEFINE(raftx_kp_f##F, 2, F, 128, RAFTX_KP_MINB)
#ifdef RAFTX_KP_ONLY          // screening builds: one specialisation
X(0)
#else
RAFTX_PERSIST128(X)
#endif
#undef X
