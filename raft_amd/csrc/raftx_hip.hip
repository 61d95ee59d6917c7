// raftx_hip.hip -- MI355X (gfx950 / CDNA4) implementation of include/raftx.h.
//
// Hot path of WISDEM/RAFT: Morison strip sweep + stochastic drag linearisation
// fixed point + per-frequency 6x6 complex solve (raft/raft_model.py:994-1236,
// raft/raft_fowt.py:1732-1957, raft/raft_member.py:1899-2152), written
// directly for CDNA4: fp64 VALU, wave64, one 256-thread workgroup per
// (design, sea state), one lane per frequency bin.
//
// Mapping:
//   * frequency is the contiguous axis of every reference array, so lane <-> w
//     makes every global load/store of a [6,nw] / [6,6,nw] slab coalesced;
//   * strip records (256 B each) are wave-uniform: they are read through the
//     scalar cache / LDS, never per lane;
//   * the only cross-frequency couplings -- the per-strip vRMS sums
//     (raft_member.py:2084-2090, helpers.py:684) and the convergence test
//     (raft_model.py:1104) -- are wave shuffles + a few LDS words inside one
//     workgroup; nothing crosses workgroups or devices;
//   * the 6x6 complex impedance is factorised per lane in registers with
//     LAPACK-style partial pivoting (pivot on |re|+|im|, as izamax).
//
// No fallback paths: every entry point either runs on the GPU or fails.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/raftx.h"

#define NF RAFTX_NFIELD
#define BLOCK 256
#define NWAVE (BLOCK / 64)

struct cplx {
    double re, im;
};
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cplx cscale(cplx a, double s) { return {a.re * s, a.im * s}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ double cabs2(cplx a) { return a.re * a.re + a.im * a.im; }

// ------------------------------------------------------------------ device tables
struct DevTables {
    // designs
    int nDesign;
    const int64_t *off;     // [nDesign+1]
    const double *strips;   // [nStrips,32]
    const double *M0, *B0, *C0;   // [nDesign,36]
    const double *MBw;      // [nDesign,2,36,nw] or null
    const int64_t *cmoff;   // [nDesign+1] or null
    const cplx *cm;         // [nRows,2,nw] or null
    // cases
    int nCase, nHead, nw;
    const double *w, *k;    // [nw]
    const double *csh, *cch, *e2kh;   // per-bin depth constants (host-computed)
    const int *mode;        // 0 finite depth, 1 deep (k h > 89.4), 2 k==0   (helpers.py:211-222)
    const double *zeta;     // [nCase,nHead,nw]
    const double *beta;     // [nCase,nHead]
    double depth, rho, g;
};

// per-lane (per frequency bin) wave constants
struct Bin {
    double w, k, csh, cch, e2kh;
    int mode;
};

// Depth-decay ratios of helpers.py:208-223, evaluated in the overflow-free
// exponential form: with E = e^{kz}, Q = e^{-k(z+2h)} = e^{-2kh}/E
//   sinh k(z+h)/sinh kh = (E-Q)/(1-e^{-2kh}),  cosh k(z+h)/sinh kh = (E+Q)/(1-e^{-2kh}),
//   cosh k(z+h)/cosh kh = (E+Q)/(1+e^{-2kh}).
__device__ __forceinline__ void depth_ratios(const Bin &b, double z, double depth, double &Sh, double &Ch, double &Cc) {
    if (b.mode == 0) {
        double E = exp(b.k * z);
        double Q = b.e2kh / E;
        Sh = (E - Q) * b.csh;
        Ch = (E + Q) * b.csh;
        Cc = (E + Q) * b.cch;
    } else if (b.mode == 1) {
        double E = exp(b.k * z);
        Sh = E;
        Ch = E;
        Cc = E + exp(-b.k * (z + 2.0 * depth));
    } else {
        Sh = 1.0;
        Ch = 99999.0;
        Cc = 99999.0;
    }
}

// Local wave elevation phasor zeta*exp(-i k (x cos b + y sin b))  (helpers.py:201)
__device__ __forceinline__ cplx local_elevation(double zeta0, double k, double xi) {
    double s, c;
    sincos(-(k * xi), &s, &c);
    return {zeta0 * c, zeta0 * s};
}

// wave-level sum of one double over 64 lanes (result in every lane)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ------------------------------------------------------------------ shared layout
// LDS carve (doubles).  W[S][18] = [n_c ; a x n_c] for c = q,p1,p2 (geometry only);
// bc[S][4] = (Bq, Bp1, Bp2, -) linearised coefficients of the live iteration;
// uv[S][12] = heading-projected drag excitation vectors; red[...] reduction scratch.
struct Lds {
    double *W;      // S*18
    double *bc;     // S*4
    double *uv;     // S*12
    double *red;    // S*NWAVE*3 per-wave partial sums of pass A
    double *Bd;     // 36
};

__device__ __forceinline__ Lds carve(double *base, int S) {
    Lds l;
    l.W = base;
    l.bc = l.W + (size_t)S * 18;
    l.uv = l.bc + (size_t)S * 4;
    l.red = l.uv + (size_t)S * 12;
    l.Bd = l.red + (size_t)S * NWAVE * 3;
    return l;
}
static size_t lds_bytes(int S) { return sizeof(double) * ((size_t)S * (18 + 4 + 12 + NWAVE * 3) + 36 + 8); }

// W_{s,c} = [n_c ; a_s x n_c]  -- the 6-vector that both projects the body
// velocity on direction c (helpers.py:178-181,396-402 folded with raft_member.py:2078-2081)
// and translates a force along n_c to the reference point (helpers.py:468-483).
__device__ __forceinline__ void build_W(const double *__restrict__ strips, int S, const Lds &l) {
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const double *rec = strips + (size_t)s * NF;
        double ax = rec[RAFTX_F_AX], ay = rec[RAFTX_F_AX + 1], az = rec[RAFTX_F_AX + 2];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double *n = rec + RAFTX_F_Q + 3 * c;
            double *W = l.W + (size_t)s * 18 + c * 6;
            W[0] = n[0];
            W[1] = n[1];
            W[2] = n[2];
            W[3] = ay * n[2] - az * n[1];
            W[4] = az * n[0] - ax * n[2];
            W[5] = ax * n[1] - ay * n[0];
        }
    }
}

// uv[s] for heading (cb,sb): U = sum_c b_c alpha_c W_c, V = sum_c b_c gamma_c W_c,
// alpha_c = n_c.x cb + n_c.y sb, gamma_c = n_c.z.  Then the strip's drag
// excitation (raft_member.py:2122-2124 / :2146-2151) is  t1*U + t2*V  with
// t1 = w zeta_s Ch, t2 = i w zeta_s Sh.
__device__ __forceinline__ void build_uv(int S, const Lds &l, double cb, double sb) {
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        double U[6] = {0, 0, 0, 0, 0, 0}, V[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double *W = l.W + (size_t)s * 18 + c * 6;
            double b = l.bc[(size_t)s * 4 + c];
            double al = b * (W[0] * cb + W[1] * sb);
            double ga = b * W[2];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                U[j] += al * W[j];
                V[j] += ga * W[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 6; j++) {
            l.uv[(size_t)s * 12 + j] = U[j];
            l.uv[(size_t)s * 12 + 6 + j] = V[j];
        }
    }
}

// B_drag[6][6] = sum_{s,c} b_{s,c} W W^T  == sum_s translateMatrix3to6DOF(Bmat_s, a_s)
// (raft_member.py:2117-2118, helpers.py:537-560).  36 lanes, one entry each.
__device__ __forceinline__ void build_Bdrag(int S, const Lds &l) {
    int e = threadIdx.x;
    if (e < 36) {
        int i = e / 6, j = e % 6;
        double acc = 0.0;
        for (int s = 0; s < S; s++) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double *W = l.W + (size_t)s * 18 + c * 6;
                acc += l.bc[(size_t)s * 4 + c] * (W[i] * W[j]);
            }
        }
        l.Bd[e] = acc;
    }
}

// Inertial excitation of one heading, accumulated over the strips
// (raft_member.py:1965-1991; helpers.py:188-236).  F[6] per lane.
__device__ __forceinline__ void inertial_excitation(const DevTables &T, const double *__restrict__ strips, int S,
                                                    const cplx *__restrict__ cm, const Bin &b, bool active, int iw,
                                                    double zeta0, double cb, double sb, cplx F[6]) {
#pragma unroll
    for (int j = 0; j < 6; j++) F[j] = {0.0, 0.0};
    if (!active) return;
    for (int s = 0; s < S; s++) {
        const double *__restrict__ rec = strips + (size_t)s * NF;
        double x = rec[RAFTX_F_X], y = rec[RAFTX_F_X + 1], z = rec[RAFTX_F_X + 2];
        cplx zs = local_elevation(zeta0, b.k, cb * x + sb * y);
        double Sh, Ch, Cc;
        depth_ratios(b, z, T.depth, Sh, Ch, Cc);
        // u = (w zs Ch cb, w zs Ch sb, i w zs Sh);  ud = i w u   (helpers.py:225-231)
        cplx wz = cscale(zs, b.w);
        cplx uh = cscale(wz, Ch);
        cplx u0 = cscale(uh, cb), u1 = cscale(uh, sb);
        cplx u2 = {-wz.im * Sh, wz.re * Sh};
        cplx ud0 = {-b.w * u0.im, b.w * u0.re};
        cplx ud1 = {-b.w * u1.im, b.w * u1.re};
        cplx ud2 = {-b.w * u2.im, b.w * u2.re};
        cplx pd = cscale(zs, T.rho * T.g);
        pd = cscale(pd, Cc);
        const double *q = rec + RAFTX_F_Q, *p1 = rec + RAFTX_F_P1, *p2 = rec + RAFTX_F_P2;
        // projections of ud on q, p1, p2
        cplx aq = cadd(cadd(cscale(ud0, q[0]), cscale(ud1, q[1])), cscale(ud2, q[2]));
        cplx a1 = cadd(cadd(cscale(ud0, p1[0]), cscale(ud1, p1[1])), cscale(ud2, p1[2]));
        cplx a2 = cadd(cadd(cscale(ud0, p2[0]), cscale(ud1, p2[1])), cscale(ud2, p2[2]));
        cplx c1, c2;
        int mcf = (int)rec[RAFTX_F_MCF];
        if (mcf >= 0) {   // MacCamy-Fuchs: complex per-bin Cm (raft_member.py:1415-1420)
            double rv = rec[RAFTX_F_RHOV];
            cplx m1 = cm[((size_t)mcf * 2 + 0) * T.nw + iw];
            cplx m2 = cm[((size_t)mcf * 2 + 1) * T.nw + iw];
            c1 = cmul(cscale(m1, rv), a1);
            c2 = cmul(cscale(m2, rv), a2);
        } else {
            c1 = cscale(a1, rec[RAFTX_F_IP1]);
            c2 = cscale(a2, rec[RAFTX_F_IP2]);
        }
        cplx cq = cadd(cscale(aq, rec[RAFTX_F_IQ]), cscale(pd, rec[RAFTX_F_AI]));   // + pDyn*a_i along q (:1988)
        cplx f0 = cadd(cadd(cscale(cq, q[0]), cscale(c1, p1[0])), cscale(c2, p2[0]));
        cplx f1 = cadd(cadd(cscale(cq, q[1]), cscale(c1, p1[1])), cscale(c2, p2[1]));
        cplx f2 = cadd(cadd(cscale(cq, q[2]), cscale(c1, p1[2])), cscale(c2, p2[2]));
        double ax = rec[RAFTX_F_AX], ay = rec[RAFTX_F_AX + 1], az = rec[RAFTX_F_AX + 2];
        F[0] = cadd(F[0], f0);
        F[1] = cadd(F[1], f1);
        F[2] = cadd(F[2], f2);
        F[3] = cadd(F[3], csub(cscale(f2, ay), cscale(f1, az)));    // a x f  (helpers.py:481)
        F[4] = cadd(F[4], csub(cscale(f0, az), cscale(f2, ax)));
        F[5] = cadd(F[5], csub(cscale(f1, ax), cscale(f0, ay)));
    }
}

// Pass A of one linearisation: per strip, RMS over all bins of the relative
// velocity components (raft_member.py:2075-2090, helpers.py:684) -> bc[S][3].
// Xi[6] is this lane's response amplitude (zero contribution for inactive lanes).
__device__ __forceinline__ void linearize_passA(const DevTables &T, const double *__restrict__ strips, int S,
                                                const Lds &l, const Bin &b, bool active, double zeta0,
                                                double cb, double sb, const cplx Xi[6]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int s = 0; s < S; s++) {
        const double *__restrict__ rec = strips + (size_t)s * NF;
        double vq2 = 0.0, v12 = 0.0, v22 = 0.0;
        if (active) {
            double x = rec[RAFTX_F_X], y = rec[RAFTX_F_X + 1], z = rec[RAFTX_F_X + 2];
            cplx zs = local_elevation(zeta0, b.k, cb * x + sb * y);
            double Sh, Ch, Cc;
            depth_ratios(b, z, T.depth, Sh, Ch, Cc);
            cplx wz = cscale(zs, b.w);
            cplx t1 = cscale(wz, Ch);                 // horizontal velocity phasor
            cplx t2 = {-wz.im * Sh, wz.re * Sh};      // vertical   velocity phasor (i w zs Sh)
            const double *W = l.W + (size_t)s * 18;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double *Wc = W + c * 6;
                double al = Wc[0] * cb + Wc[1] * sb, ga = Wc[2];
                cplx G = cadd(cscale(t1, al), cscale(t2, ga));           // n_c . u
                cplx P = {0.0, 0.0};                                       // W_c . Xi  (body displacement along c)
#pragma unroll
                for (int j = 0; j < 6; j++) P = cadd(P, cscale(Xi[j], Wc[j]));
                cplx v = {G.re + b.w * P.im, G.im - b.w * P.re};           // G - i w P
                double m = cabs2(v);
                if (c == 0) vq2 = m;
                else if (c == 1) v12 = m;
                else v22 = m;
            }
        }
        vq2 = wave_sum(vq2);
        v12 = wave_sum(v12);
        v22 = wave_sum(v22);
        if (lane == 0) {
            double *r = l.red + ((size_t)s * NWAVE + wv) * 3;
            r[0] = vq2;
            r[1] = v12;
            r[2] = v22;
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const double *__restrict__ rec = strips + (size_t)s * NF;
        double a = 0, c1 = 0, c2 = 0;
        for (int i = 0; i < NWAVE; i++) {
            const double *r = l.red + ((size_t)s * NWAVE + i) * 3;
            a += r[0];
            c1 += r[1];
            c2 += r[2];
        }
        double vRq = sqrt(0.5 * a), vR1, vR2;
        if (rec[RAFTX_F_CIRC] != 0.0) {        // circular: total transverse velocity (:2085-2087)
            vR1 = sqrt(0.5 * (c1 + c2));
            vR2 = vR1;
        } else {
            vR1 = sqrt(0.5 * c1);
            vR2 = sqrt(0.5 * c2);
        }
        l.bc[(size_t)s * 4 + 0] = rec[RAFTX_F_DQ] * vRq + rec[RAFTX_F_DEND] * vRq;   // Bprime_q + Bprime_End (:2093,:2110)
        l.bc[(size_t)s * 4 + 1] = rec[RAFTX_F_DP1] * vR1;
        l.bc[(size_t)s * 4 + 2] = rec[RAFTX_F_DP2] * vR2;
    }
    __syncthreads();
}

// Pass B: drag excitation of one heading with the live coefficients (uv built
// for that heading): F[6] per lane.  raft_member.py:2122-2124, :2146-2151.
__device__ __forceinline__ void drag_excitation(const DevTables &T, const double *__restrict__ strips, int S,
                                                const Lds &l, const Bin &b, bool active, double zeta0,
                                                double cb, double sb, cplx F[6]) {
#pragma unroll
    for (int j = 0; j < 6; j++) F[j] = {0.0, 0.0};
    if (!active) return;
    for (int s = 0; s < S; s++) {
        const double *__restrict__ rec = strips + (size_t)s * NF;
        double x = rec[RAFTX_F_X], y = rec[RAFTX_F_X + 1], z = rec[RAFTX_F_X + 2];
        cplx zs = local_elevation(zeta0, b.k, cb * x + sb * y);
        double Sh, Ch, Cc;
        depth_ratios(b, z, T.depth, Sh, Ch, Cc);
        cplx wz = cscale(zs, b.w);
        cplx t1 = cscale(wz, Ch);
        cplx t2 = {-wz.im * Sh, wz.re * Sh};
        const double *uv = l.uv + (size_t)s * 12;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            F[j].re += t1.re * uv[j] + t2.re * uv[6 + j];
            F[j].im += t1.im * uv[j] + t2.im * uv[6 + j];
        }
    }
}

// ------------------------------------------------------------------ 6x6 complex LU in registers
struct Lu6 {
    double ar[6][6], ai[6][6];
    int piv[6];
};

// zgetrf-style: partial pivoting on |re|+|im| (izamax), full row interchanges.
__device__ __forceinline__ void lu6_factor(Lu6 &A) {
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
        A.piv[k] = p;
        if (__any(p != k)) {
#pragma unroll
            for (int r = k + 1; r < 6; r++) {
                bool sw = (p == r);
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double tr = A.ar[k][c], ti = A.ai[k][c];
                    A.ar[k][c] = sw ? A.ar[r][c] : tr;
                    A.ai[k][c] = sw ? A.ai[r][c] : ti;
                    A.ar[r][c] = sw ? tr : A.ar[r][c];
                    A.ai[r][c] = sw ? ti : A.ai[r][c];
                }
            }
        }
        // reciprocal of the pivot
        double pr = A.ar[k][k], pi = A.ai[k][k];
        double d = pr * pr + pi * pi;
        double ir = pr / d, ii = -pi / d;
#pragma unroll
        for (int r = k + 1; r < 6; r++) {
            double lr = A.ar[r][k] * ir - A.ai[r][k] * ii;
            double li = A.ar[r][k] * ii + A.ai[r][k] * ir;
            A.ar[r][k] = lr;
            A.ai[r][k] = li;
#pragma unroll
            for (int c = k + 1; c < 6; c++) {
                A.ar[r][c] -= lr * A.ar[k][c] - li * A.ai[k][c];
                A.ai[r][c] -= lr * A.ai[k][c] + li * A.ar[k][c];
            }
        }
    }
}

__device__ __forceinline__ void lu6_solve(const Lu6 &A, cplx b[6]) {
#pragma unroll
    for (int k = 0; k < 6; k++) {   // zlaswp
        int p = A.piv[k];
#pragma unroll
        for (int r = k + 1; r < 6; r++) {
            bool sw = (p == r);
            cplx t = b[k];
            b[k] = sw ? b[r] : t;
            b[r] = sw ? t : b[r];
        }
    }
#pragma unroll
    for (int k = 0; k < 6; k++)
#pragma unroll
        for (int r = k + 1; r < 6; r++) {
            double lr = A.ar[r][k], li = A.ai[r][k];
            b[r].re -= lr * b[k].re - li * b[k].im;
            b[r].im -= lr * b[k].im + li * b[k].re;
        }
#pragma unroll
    for (int k = 5; k >= 0; k--) {
        cplx s = b[k];
#pragma unroll
        for (int c = k + 1; c < 6; c++) {
            s.re -= A.ar[k][c] * b[c].re - A.ai[k][c] * b[c].im;
            s.im -= A.ar[k][c] * b[c].im + A.ai[k][c] * b[c].re;
        }
        double pr = A.ar[k][k], pi = A.ai[k][k];
        double d = pr * pr + pi * pi;
        b[k] = {(s.re * pr + s.im * pi) / d, (s.im * pr - s.re * pi) / d};
    }
}

// ------------------------------------------------------------------ kernels
__device__ __forceinline__ Bin load_bin(const DevTables &T, int iw, bool active) {
    Bin b;
    int i = active ? iw : 0;
    b.w = T.w[i];
    b.k = T.k[i];
    b.csh = T.csh[i];
    b.cch = T.cch[i];
    b.e2kh = T.e2kh[i];
    b.mode = T.mode[i];
    return b;
}

// F_iner [nDesign,nCase,nHead,6,nw]   (raft_fowt.py:1854-1857,1888)
__global__ void __launch_bounds__(BLOCK) k_excitation(DevTables T, cplx *__restrict__ F_iner) {
    const int pair = blockIdx.x / T.nHead, ih = blockIdx.x % T.nHead;
    const int d = pair / T.nCase, ic = pair % T.nCase;
    const int S = (int)(T.off[d + 1] - T.off[d]);
    const double *strips = T.strips + (size_t)T.off[d] * NF;
    const cplx *cm = T.cm ? T.cm + (size_t)T.cmoff[d] * 2 * T.nw : nullptr;
    const int iw = threadIdx.x;
    const bool active = iw < T.nw;
    Bin b = load_bin(T, iw, active);
    const double beta = T.beta[(size_t)ic * T.nHead + ih];
    const double cb = cos(beta), sb = sin(beta);
    const double zeta0 = active ? T.zeta[((size_t)ic * T.nHead + ih) * T.nw + iw] : 0.0;
    cplx F[6];
    inertial_excitation(T, strips, S, cm, b, active, iw, zeta0, cb, sb, F);
    if (active) {
        cplx *out = F_iner + (((size_t)pair * T.nHead + ih) * 6) * T.nw + iw;
#pragma unroll
        for (int j = 0; j < 6; j++) out[(size_t)j * T.nw] = F[j];
    }
}

// One linearisation about a given Xi (raft_fowt.py:1891-1957).
__global__ void __launch_bounds__(BLOCK) k_linearize(DevTables T, const cplx *__restrict__ Xi_in,
                                                     double *__restrict__ B_drag, cplx *__restrict__ F_drag) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int pair = blockIdx.x;
    const int d = pair / T.nCase, ic = pair % T.nCase;
    const int S = (int)(T.off[d + 1] - T.off[d]);
    const double *strips = T.strips + (size_t)T.off[d] * NF;
    Lds l = carve(smem, S);
    const int iw = threadIdx.x;
    const bool active = iw < T.nw;
    Bin b = load_bin(T, iw, active);
    build_W(strips, S, l);
    __syncthreads();
    cplx Xi[6];
#pragma unroll
    for (int j = 0; j < 6; j++) Xi[j] = active ? Xi_in[((size_t)pair * 6 + j) * T.nw + iw] : cplx{0.0, 0.0};
    {
        const double beta = T.beta[(size_t)ic * T.nHead + 0];
        const double zeta0 = active ? T.zeta[((size_t)ic * T.nHead + 0) * T.nw + iw] : 0.0;
        linearize_passA(T, strips, S, l, b, active, zeta0, cos(beta), sin(beta), Xi);
    }
    if (B_drag) {
        build_Bdrag(S, l);
        __syncthreads();
        if (threadIdx.x < 36) B_drag[(size_t)pair * 36 + threadIdx.x] = l.Bd[threadIdx.x];
    }
    if (F_drag) {
        for (int ih = 0; ih < T.nHead; ih++) {
            const double beta = T.beta[(size_t)ic * T.nHead + ih];
            const double cb = cos(beta), sb = sin(beta);
            const double zeta0 = active ? T.zeta[((size_t)ic * T.nHead + ih) * T.nw + iw] : 0.0;
            __syncthreads();
            build_uv(S, l, cb, sb);
            __syncthreads();
            cplx F[6];
            drag_excitation(T, strips, S, l, b, active, zeta0, cb, sb, F);
            if (active) {
                cplx *out = F_drag + (((size_t)pair * T.nHead + ih) * 6) * T.nw + iw;
#pragma unroll
                for (int j = 0; j < 6; j++) out[(size_t)j * T.nw] = F[j];
            }
        }
    }
}

struct SolveArgs {
    int nIter;          // loop bound = YAML nIter + 1 (raft_model.py:977)
    double tol, XiStart;
    const cplx *F_extra;    // [pair,nHead,6,nw] or null
    cplx *Xi;               // [pair,nHead,6,nw] or null
    int *niter, *flags;     // [pair]
    double *B_drag;         // [pair,36] or null
    cplx *F_wave;           // [pair,nHead,6,nw] or null
    cplx *Z;                // [pair,36,nw] or null
};

// The fused fixed point (raft_model.py:1052-1142) + per-heading response (:1189-1236).
__global__ void __launch_bounds__(BLOCK) k_solve_dynamics(DevTables T, SolveArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int pair = blockIdx.x;
    const int d = pair / T.nCase, ic = pair % T.nCase;
    const int S = (int)(T.off[d + 1] - T.off[d]);
    const double *strips = T.strips + (size_t)T.off[d] * NF;
    const cplx *cm = T.cm ? T.cm + (size_t)T.cmoff[d] * 2 * T.nw : nullptr;
    Lds l = carve(smem, S);
    const int iw = threadIdx.x;
    const bool active = iw < T.nw;
    const int nw = T.nw, nH = T.nHead;
    Bin b = load_bin(T, iw, active);
    build_W(strips, S, l);

    const double beta0 = T.beta[(size_t)ic * nH];
    const double cb0 = cos(beta0), sb0 = sin(beta0);
    const double zeta00 = active ? T.zeta[((size_t)ic * nH) * nw + iw] : 0.0;

    // F_lin = F_extra[0] + F_iner[0]   (raft_model.py:1048)
    cplx Flin[6];
    inertial_excitation(T, strips, S, cm, b, active, iw, zeta00, cb0, sb0, Flin);
    if (A.F_extra && active) {
#pragma unroll
        for (int j = 0; j < 6; j++) {
            cplx fe = A.F_extra[(((size_t)pair * nH) * 6 + j) * nw + iw];
            Flin[j] = cadd(fe, Flin[j]);
        }
    }
    cplx XiLast[6], Xi[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        XiLast[j] = active ? cplx{A.XiStart, 0.0} : cplx{0.0, 0.0};   // :999
        Xi[j] = {0.0, 0.0};
    }
    // frequency-dependent + constant system matrices of this lane's bin (:1045-1047)
    const double *M0 = T.M0 + (size_t)d * 36, *B0 = T.B0 + (size_t)d * 36, *C0 = T.C0 + (size_t)d * 36;
    const double *Mw = T.MBw ? T.MBw + ((size_t)d * 2 + 0) * 36 * nw : nullptr;
    const double *Bw = T.MBw ? T.MBw + ((size_t)d * 2 + 1) * 36 * nw : nullptr;
    __syncthreads();

    Lu6 lu;
    int iiter = 0, done = 0, converged = 0, nan = 0;
    while (iiter < A.nIter) {
        linearize_passA(T, strips, S, l, b, active, zeta00, cb0, sb0, XiLast);   // :1063
        build_Bdrag(S, l);
        build_uv(S, l, cb0, sb0);
        __syncthreads();
        cplx Fd[6];
        drag_excitation(T, strips, S, l, b, active, zeta00, cb0, sb0, Fd);      // :1064
        // Z = -w^2 M + i w B + C   (:1086)
        const int iwc = active ? iw : 0;
        const double w = b.w, w2 = b.w * b.w;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 6; c++) {
                int e = r * 6 + c;
                double M = M0[e], B = B0[e];
                if (Mw) {
                    M += Mw[(size_t)e * nw + iwc];
                    B += Bw[(size_t)e * nw + iwc];
                }
                B += l.Bd[e];
                lu.ar[r][c] = -w2 * M + C0[e];
                lu.ai[r][c] = w * B;
            }
        if (A.Z && active) {      // last iterate wins (fowt.Z, :1155)
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = 0; c < 6; c++)
                    A.Z[((size_t)pair * 36 + r * 6 + c) * nw + iw] = cplx{lu.ar[r][c], lu.ai[r][c]};
        }
        lu6_factor(lu);
#pragma unroll
        for (int j = 0; j < 6; j++) Xi[j] = cadd(Flin[j], Fd[j]);               // :1081
        lu6_solve(lu, Xi);                                                       // :1089
        done = iiter + 1;
        // NaN check (:1098) and convergence (:1103-1104)
        int bad = 0, ok = 1;
        if (active) {
#pragma unroll
            for (int j = 0; j < 6; j++) {
                if (isnan(Xi[j].re) || isnan(Xi[j].im)) bad = 1;
                double dr = Xi[j].re - XiLast[j].re, di = Xi[j].im - XiLast[j].im;
                double tc = hypot(dr, di) / (hypot(Xi[j].re, Xi[j].im) + A.tol);
                if (!(tc < A.tol)) ok = 0;
            }
        }
        nan = __syncthreads_or(bad);
        if (nan) break;
        converged = __syncthreads_and(ok);
        if (converged) break;
#pragma unroll
        for (int j = 0; j < 6; j++) {                                            // :1133
            XiLast[j].re = 0.2 * XiLast[j].re + 0.8 * Xi[j].re;
            XiLast[j].im = 0.2 * XiLast[j].im + 0.8 * Xi[j].im;
        }
        iiter++;
    }

    // per-heading response with the last impedance and the last coefficients (:1200-1236)
    for (int ih = 0; ih < nH; ih++) {
        const double beta = T.beta[(size_t)ic * nH + ih];
        const double cb = cos(beta), sb = sin(beta);
        const double zeta0 = active ? T.zeta[((size_t)ic * nH + ih) * nw + iw] : 0.0;
        cplx Fi[6], Fd[6];
        if (ih == 0) {
#pragma unroll
            for (int j = 0; j < 6; j++) Fi[j] = Flin[j];
        } else {
            inertial_excitation(T, strips, S, cm, b, active, iw, zeta0, cb, sb, Fi);
            if (A.F_extra && active) {
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    cplx fe = A.F_extra[(((size_t)pair * nH + ih) * 6 + j) * nw + iw];
                    Fi[j] = cadd(fe, Fi[j]);
                }
            }
        }
        __syncthreads();
        build_uv(S, l, cb, sb);
        __syncthreads();
        drag_excitation(T, strips, S, l, b, active, zeta0, cb, sb, Fd);          // :1209
        cplx f[6];
#pragma unroll
        for (int j = 0; j < 6; j++) f[j] = cadd(Fi[j], Fd[j]);                   // :1212
        if (active) {
            if (A.F_wave) {
#pragma unroll
                for (int j = 0; j < 6; j++) A.F_wave[(((size_t)pair * nH + ih) * 6 + j) * nw + iw] = f[j];
            }
            if (A.Xi) {
                lu6_solve(lu, f);                                                // Zinv @ F_wave (:1216)
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    cplx v = nan ? cplx{NAN, NAN} : f[j];
                    A.Xi[(((size_t)pair * nH + ih) * 6 + j) * nw + iw] = v;
                }
            }
        }
    }
    if (threadIdx.x < 36 && A.B_drag) A.B_drag[(size_t)pair * 36 + threadIdx.x] = l.Bd[threadIdx.x];
    if (threadIdx.x == 0) {
        if (A.niter) A.niter[pair] = done;
        if (A.flags) A.flags[pair] = (converged ? RAFTX_FLAG_CONVERGED : 0) | (nan ? RAFTX_FLAG_NAN : 0);
    }
}

// Coupled array solve (raft_model.py:1164-1236): one 64-lane workgroup per
// (system, bin); the augmented matrix [Z_sys | F] lives in LDS, lane r owns
// row r during the elimination.
__global__ void __launch_bounds__(64) k_solve_system(int nSys, int nUnit, int nRhs, int nw,
                                                     const double *__restrict__ w, const cplx *__restrict__ Zblk,
                                                     const double *__restrict__ Mc, const double *__restrict__ Bc,
                                                     const double *__restrict__ Cc, const cplx *__restrict__ F,
                                                     cplx *__restrict__ Xi) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n = 6 * nUnit, ld = n + nRhs;
    cplx *A = reinterpret_cast<cplx *>(smem);              // [n][ld]
    __shared__ int s_p;
    const int s = blockIdx.x / nw, iw = blockIdx.x % nw;
    const double ww = w[iw];
    for (int e = threadIdx.x; e < n * ld; e += 64) {
        int r = e / ld, c = e % ld;
        cplx v = {0.0, 0.0};
        if (c < n) {
            if (r / 6 == c / 6) v = Zblk[((((size_t)s * nUnit + r / 6) * 6 + r % 6) * 6 + c % 6) * nw + iw];
            size_t o = (size_t)s * n * n + (size_t)r * n + c;
            double m = Mc ? Mc[o] : 0.0, bb = Bc ? Bc[o] : 0.0, kk = Cc ? Cc[o] : 0.0;
            if (Mc || Bc || Cc) {
                v.re += -(ww * ww) * m + kk;
                v.im += ww * bb;
            }
        } else {
            v = F[(((size_t)s * nRhs + (c - n)) * n + r) * nw + iw];
        }
        A[e] = v;
    }
    __syncthreads();
    for (int k = 0; k < n; k++) {
        if (threadIdx.x == 0) {
            int p = k;
            double best = fabs(A[k * ld + k].re) + fabs(A[k * ld + k].im);
            for (int r = k + 1; r < n; r++) {
                double v = fabs(A[r * ld + k].re) + fabs(A[r * ld + k].im);
                if (v > best) {
                    best = v;
                    p = r;
                }
            }
            s_p = p;
        }
        __syncthreads();
        const int p = s_p;
        if (p != k)
            for (int c = threadIdx.x; c < ld; c += 64) {
                cplx t = A[k * ld + c];
                A[k * ld + c] = A[p * ld + c];
                A[p * ld + c] = t;
            }
        __syncthreads();
        const cplx pv = A[k * ld + k];
        const double dd = pv.re * pv.re + pv.im * pv.im;
        const cplx inv = {pv.re / dd, -pv.im / dd};
        for (int r = k + 1 + threadIdx.x; r < n; r += 64) {
            cplx lf = cmul(A[r * ld + k], inv);
            for (int c = k + 1; c < ld; c++) A[r * ld + c] = csub(A[r * ld + c], cmul(lf, A[k * ld + c]));
            A[r * ld + k] = lf;
        }
        __syncthreads();
    }
    // back substitution: lane per right-hand side
    for (int r = threadIdx.x; r < nRhs; r += 64) {
        for (int k = n - 1; k >= 0; k--) {
            cplx sum = A[k * ld + n + r];
            for (int c = k + 1; c < n; c++) sum = csub(sum, cmul(A[k * ld + c], A[c * ld + n + r]));
            cplx pv = A[k * ld + k];
            double dd = pv.re * pv.re + pv.im * pv.im;
            cplx x = {(sum.re * pv.re + sum.im * pv.im) / dd, (sum.im * pv.re - sum.re * pv.im) / dd};
            A[k * ld + n + r] = x;
        }
        for (int k = 0; k < n; k++) Xi[(((size_t)s * nRhs + r) * n + k) * nw + iw] = A[k * ld + n + r];
    }
}

// ------------------------------------------------------------------ host side
struct raftx_ctx {
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    char err[512];
    DevTables T;
    std::vector<void *> design_allocs, case_allocs, result_allocs;
    // resident results of the last raftx_solve_dynamics_device
    cplx *rXi, *rFw, *rZ, *rFe;
    double *rB;
    int *rNi, *rFl;
    size_t r_npair, r_nx, r_nz;
    int r_mask;
    bool r_fe;
    int maxS;
    double last_ms;
    bool have_designs, have_cases;
    int nw_designs;
};

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
    c->last_ms = 0.0;
    c->have_designs = c->have_cases = false;
    c->nw_designs = 0;
    c->rXi = c->rFw = c->rZ = c->rFe = nullptr;
    c->rB = nullptr;
    c->rNi = c->rFl = nullptr;
    c->r_npair = c->r_nx = c->r_nz = 0;
    c->r_mask = 0;
    c->r_fe = false;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return -6;
    }
    *out = c;
    return 0;
}

static void free_list(std::vector<void *> &v) {
    for (void *p : v)
        if (p) (void)hipFree(p);
    v.clear();
}

extern "C" void raftx_ctx_destroy(raftx_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_list(c->design_allocs);
    free_list(c->case_allocs);
    free_list(c->result_allocs);
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char *raftx_last_error(raftx_ctx *c) { return c ? c->err : "null ctx"; }
extern "C" double raftx_last_kernel_ms(raftx_ctx *c) { return c ? c->last_ms : 0.0; }

template <typename Tp>
static int upload(raftx_ctx *c, std::vector<void *> &bag, const Tp *host, size_t n, const Tp **dev) {
    *dev = nullptr;
    if (!host || n == 0) return 0;
    void *p = nullptr;
    HIPCHK(c, hipMalloc(&p, n * sizeof(Tp)));
    bag.push_back(p);
    HIPCHK(c, hipMemcpyAsync(p, host, n * sizeof(Tp), hipMemcpyHostToDevice, c->stream));
    *dev = reinterpret_cast<const Tp *>(p);
    return 0;
}

extern "C" int raftx_upload_designs(raftx_ctx *c, int nDesign, const int64_t *stripOffsets, const double *strips,
                                    int nStripFields, const double *M0, const double *B0, const double *C0, int nw,
                                    const double *MBw, const int64_t *cmOffsets, const raftx_c128 *CmMCF) {
    if (!c) return -1;
    if (nStripFields != NF) FAIL(c, "nStripFields=%d, expected %d", nStripFields, NF);
    if (nDesign < 0 || !stripOffsets || !M0 || !B0 || !C0) FAIL(c, "upload_designs: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_list(c->design_allocs);
    c->have_designs = false;
    int maxS = 0;
    for (int d = 0; d < nDesign; d++) {
        int64_t S = stripOffsets[d + 1] - stripOffsets[d];
        if (S < 0) FAIL(c, "strip offsets not monotone at design %d", d);
        if (S > maxS) maxS = (int)S;
    }
    DevTables &T = c->T;
    T.nDesign = nDesign;
    int rc = 0;
    rc |= upload(c, c->design_allocs, stripOffsets, (size_t)nDesign + 1, &T.off);
    rc |= upload(c, c->design_allocs, strips, (size_t)stripOffsets[nDesign] * NF, &T.strips);
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
    return 0;
}

extern "C" int raftx_upload_cases(raftx_ctx *c, int nCase, int nHead, int nw, const double *w, const double *k,
                                  double depth, double rho, double g, const double *zeta, const double *beta) {
    if (!c) return -1;
    if (nCase < 0 || nHead < 1 || nw < 1 || !w || !k || !zeta || !beta) FAIL(c, "upload_cases: bad arguments");
    if (nw > BLOCK) FAIL(c, "nw=%d exceeds the %d bins per workgroup supported by this build", nw, BLOCK);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_list(c->case_allocs);
    c->have_cases = false;
    // per-bin depth constants, computed once on the host in full libm precision
    std::vector<double> csh(nw), cch(nw), e2kh(nw);
    std::vector<int> mode(nw);
    for (int i = 0; i < nw; i++) {
        double kh = k[i] * depth;
        if (k[i] == 0.0) {
            mode[i] = 2;
            csh[i] = cch[i] = e2kh[i] = 0.0;
        } else if (kh > 89.4) {
            mode[i] = 1;
            csh[i] = cch[i] = 1.0;
            e2kh[i] = 0.0;
        } else {
            mode[i] = 0;
            e2kh[i] = exp(-2.0 * kh);
            csh[i] = 1.0 / (-expm1(-2.0 * kh));
            cch[i] = 1.0 / (1.0 + e2kh[i]);
        }
    }
    DevTables &T = c->T;
    T.nCase = nCase;
    T.nHead = nHead;
    T.nw = nw;
    T.depth = depth;
    T.rho = rho;
    T.g = g;
    int rc = 0;
    rc |= upload(c, c->case_allocs, w, (size_t)nw, &T.w);
    rc |= upload(c, c->case_allocs, k, (size_t)nw, &T.k);
    rc |= upload(c, c->case_allocs, csh.data(), (size_t)nw, &T.csh);
    rc |= upload(c, c->case_allocs, cch.data(), (size_t)nw, &T.cch);
    rc |= upload(c, c->case_allocs, e2kh.data(), (size_t)nw, &T.e2kh);
    rc |= upload(c, c->case_allocs, mode.data(), (size_t)nw, &T.mode);
    rc |= upload(c, c->case_allocs, zeta, (size_t)nCase * nHead * nw, &T.zeta);
    rc |= upload(c, c->case_allocs, beta, (size_t)nCase * nHead, &T.beta);
    if (rc) return -2;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_cases = true;
    return 0;
}

#define LDS_LIMIT (160 * 1024)

template <typename K>
static int prep_lds(raftx_ctx *c, K kernel, size_t bytes) {
    if (bytes > LDS_LIMIT) FAIL(c, "a design has %d submerged strips: exceeds the LDS-resident strip budget", c->maxS);
    if (bytes > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)bytes));
    return 0;
}

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
    ~Scratch() { free_list(bag); }
    template <typename Tp>
    Tp *alloc(size_t n) {
        void *p = nullptr;
        if (n == 0) return nullptr;
        if (hipMalloc(&p, n * sizeof(Tp)) != hipSuccess) return nullptr;
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
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (npair) hipLaunchKernelGGL(k_excitation, dim3((unsigned)(npair * T.nHead)), dim3(BLOCK), 0, c->stream, T, dF);
    if (finish_timed(c)) return -2;
    if (n) D2H(c, F_iner, dF, n * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
    if (prep_lds(c, k_linearize, lds_bytes(c->maxS))) return -1;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (npair)
        hipLaunchKernelGGL(k_linearize, dim3((unsigned)npair), dim3(BLOCK), lds_bytes(c->maxS), c->stream, T, dXi, dB,
                           dF);
    if (finish_timed(c)) return -2;
    if (dB) D2H(c, B_drag, dB, npair * 36 * sizeof(double));
    if (dF) D2H(c, F_drag, dF, npair * T.nHead * 6 * T.nw * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

template <typename Tp>
static Tp *dev_alloc(raftx_ctx *c, size_t n) {
    void *p = nullptr;
    if (n == 0) n = 1;
    if (hipMalloc(&p, n * sizeof(Tp)) != hipSuccess) return nullptr;
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
    free_list(c->result_allocs);
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
        free_list(c->result_allocs);
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

extern "C" int raftx_solve_dynamics_device(raftx_ctx *c, int nIter, double tol, double XiStart,
                                           const raftx_c128 *F_extra, int want_mask) {
    if (check_ready(c)) return -1;
    if (nIter < 0) FAIL(c, "solve_dynamics: nIter < 0");
    HIPCHK(c, hipSetDevice(c->device));
    const DevTables &T = c->T;
    if (ensure_results(c, want_mask, F_extra != nullptr)) return -1;
    SolveArgs A;
    A.nIter = nIter + 1;
    A.tol = tol;
    A.XiStart = XiStart;
    A.F_extra = F_extra ? c->rFe : nullptr;
    A.Xi = c->rXi;
    A.niter = c->rNi;
    A.flags = c->rFl;
    A.B_drag = (want_mask & RAFTX_WANT_BDRAG) ? c->rB : nullptr;
    A.F_wave = (want_mask & RAFTX_WANT_FWAVE) ? c->rFw : nullptr;
    A.Z = (want_mask & RAFTX_WANT_Z) ? c->rZ : nullptr;
    if (F_extra && c->r_nx) H2D(c, c->rFe, F_extra, c->r_nx * sizeof(cplx));
    if (prep_lds(c, k_solve_dynamics, lds_bytes(c->maxS))) return -1;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    if (c->r_npair)
        hipLaunchKernelGGL(k_solve_dynamics, dim3((unsigned)c->r_npair), dim3(BLOCK), lds_bytes(c->maxS), c->stream, T,
                           A);
    return finish_timed(c);
}

extern "C" int raftx_fetch_results(raftx_ctx *c, raftx_c128 *Xi, int32_t *niter, int32_t *flags, double *B_drag,
                                   raftx_c128 *F_wave, raftx_c128 *Z) {
    if (!c) return -1;
    if (!c->rXi) FAIL(c, "fetch_results: no resident results");
    HIPCHK(c, hipSetDevice(c->device));
    if (B_drag && !c->rB) FAIL(c, "fetch_results: B_drag was not kept");
    if (F_wave && !c->rFw) FAIL(c, "fetch_results: F_wave was not kept");
    if (Z && !c->rZ) FAIL(c, "fetch_results: Z was not kept");
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

extern "C" int raftx_solve_system(raftx_ctx *c, int nSys, int nUnit, int nRhs, int nw, const double *w,
                                  const raftx_c128 *Zblk, const double *Mc, const double *Bc, const double *Cc,
                                  const raftx_c128 *F, raftx_c128 *Xi) {
    if (!c) return -1;
    if (nSys < 0 || nUnit < 1 || nRhs < 1 || nw < 1 || !w || !Zblk || !F || !Xi) FAIL(c, "solve_system: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const int n = 6 * nUnit;
    size_t lds = sizeof(cplx) * (size_t)n * (n + nRhs);
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
    if (nSys) {
        if (lds > 64 * 1024)
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve_system),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_solve_system, dim3((unsigned)((size_t)nSys * nw)), dim3(64), lds, c->stream, nSys, nUnit,
                           nRhs, nw, dw, dZ, dM, dB, dC, dF, dX);
    }
    if (finish_timed(c)) return -2;
    if (nSys) D2H(c, Xi, dX, nf * sizeof(cplx));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
