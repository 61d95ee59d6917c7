// raftx_device.h -- gfx950 device code of the RAFT hot path (included by raftx_hip.hip).
//
// Work decomposition:
//   workgroup  <-> one (design, sea state) item; 256 threads = 4 wave64
//   lane       <-> one frequency bin w (the contiguous axis of every reference array)
//   strip loop <-> sequential inside each lane; strip constants are staged ONCE per
//                  workgroup into LDS ("LDS-staged member geometry") and read as
//                  wave-uniform broadcasts.
//
// Everything a strip contributes is expressed through three 6-vectors
//   W_c = [n_c ; a x n_c],  c = q, p1, p2      (n_c unit vectors, a = arm to the reference point)
// because  n_c . (body velocity at the strip) = i w W_c . Xi           (helpers.py:178-181,396-402)
//          translate(f n_c, a)               = f W_c                   (helpers.py:468-483)
//          translateMatrix3to6DOF(b n_c n_c^T, a) = b W_c W_c^T        (helpers.py:537-560)
// and through two complex phasors per (strip, bin)
//   t1 = w zeta e^{-i k xi} cosh k(z+h)/sinh kh      (horizontal particle velocity)
//   t2 = i w zeta e^{-i k xi} sinh k(z+h)/sinh kh    (vertical particle velocity)      (helpers.py:201-228)
// so that  n_c . u = al_c t1 + ga_c t2  with al_c = n_cx cos(beta) + n_cy sin(beta), ga_c = n_cz.
//
// Wave kinematics along a member are advanced strip-to-strip by complex/real
// rotors (e^{-i k du}, e^{+-k dz}) instead of re-evaluating sincos/exp per strip;
// the packer marks run starts ("step 0": exact evaluation) and the host checks
// every hint against the absolute strip positions at upload.
#pragma once

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

// ------------------------------------------------------------------ fp64 elementary functions
// Straight-line sincos / exp for the moderate arguments of this problem
// (|k xi| < 1e5, |k z| < 700): Cody-Waite reduction + Taylor polynomials whose
// truncation error is < 2^-55 on the reduced interval.  No tables, no branches.
__device__ __forceinline__ void fast_sincos(double x, double &s, double &c) {
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00;    // first 33 bits of pi/2
    const double pio2_1t = 6.07710050650619224932e-11;   // pi/2 - pio2_1
    const double pio2_2 = 6.07710050630396597660e-11;    // second 33 bits
    const double pio2_2t = 2.02226624879595063154e-21;   // pi/2 - (pio2_1 + pio2_2)
    double fn = rint(x * two_over_pi);
    double r = fma(-fn, pio2_1, x);
    r = fma(-fn, pio2_2, r);
    r = fma(-fn, pio2_2t, r);
    (void)pio2_1t;
    int n = (int)fn;
    double z = r * r;
    // sin r = r + r^3 (-1/3! + z/5! - z^2/7! + ... - z^6/15!) ; cos r = 1 - z/2 + z^2 (1/4! - z/6! + ... + z^6/16!)
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
    // quadrant
    double ss = (n & 1) ? cr : sr;
    double cc = (n & 1) ? sr : cr;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
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
    const double *csh, *cch, *e2kh;   // per-bin depth constants (host-computed, full libm precision)
    const int *mode;        // 0 finite depth / deep (same formula), 2 k==0   (helpers.py:211-222)
    const double *zeta;     // [nCase,nHead,nw]
    const double *beta;     // [nCase,nHead]
    double depth, rho, g;
};

// per-lane (per frequency bin) wave constants
struct Bin {
    double w, k, csh, cch, e2kh;
    int mode;
};

__device__ __forceinline__ Bin load_bin(const DevTables &T, int iw) {
    Bin b;
    b.w = T.w[iw];
    b.k = T.k[iw];
    b.csh = T.csh[iw];
    b.cch = T.cch[iw];
    b.e2kh = T.e2kh[iw];
    b.mode = T.mode[iw];
    return b;
}

// ------------------------------------------------------------------ LDS layout
// Per-strip staged record (doubles).  SR_W holds, for c = q,p1,p2: W_c[6], al_c, ga_c.
#define SR_X 0      // x, y, z
#define SR_STEP 3   // rotor multiplier m (0 = exact evaluation), unit step length
#define SR_UNIT 4
#define SR_Q 5      // qx, qy, qz (direction of the run)
#define SR_W 8      // 3 x 8
#define SR_IQ 32    // Iq, Ip1, Ip2, a_i, rhoV, mcf
#define SR_DQ 38    // dq, dp1, dp2, dend, circ
#define SR_N 44

#define SB 4                 // strips per reduction chunk of pass A
#define TROW 66              // padded row of the per-wave transposition tile (stride == 2 mod 32)

struct Lds {
    double *sr;     // S*SR_N   staged strips
    double *bc;     // S*4      live linearised coefficients (Bq+Bend, Bp1, Bp2)
    double *uv;     // S*12     heading-projected drag excitation vectors
    double *accw;   // NWAVE*S*3 per-wave partial sums of pass A
    double *tile;   // NWAVE*SB*3*TROW transposition tiles
    double *Bd;     // 36
    double *mat;    // 108: M0, B0, C0 of this design
    double *flin;   // 12*nw  per-lane F_lin (lives across iterations)
    // per-lane parking of XiLast while the 6x6 system is factorised; aliases the
    // transposition tiles, which are idle outside pass A
    double *park0;
    int nlane;      // row length of flin / park0 (= nw)
};

__device__ __forceinline__ Lds carve(double *base, int S, int nw) {
    Lds l;
    l.nlane = nw;
    l.sr = base;
    l.bc = l.sr + (size_t)S * SR_N;
    l.uv = l.bc + (size_t)S * 4;
    l.accw = l.uv + (size_t)S * 12;
    l.tile = l.accw + (size_t)NWAVE * S * 3;
    l.Bd = l.tile + (size_t)NWAVE * SB * 3 * TROW;
    l.mat = l.Bd + 36;
    l.flin = l.mat + 108;
    l.park0 = l.tile;
    return l;
}
static_assert(NWAVE * SB * 3 * TROW >= 12 * BLOCK, "the parking area must fit inside the tile region");
static size_t lds_bytes(int S, int nw) {
    return sizeof(double) * ((size_t)S * (SR_N + 4 + 12 + NWAVE * 3) + (size_t)NWAVE * SB * 3 * TROW + 36 + 108 +
                             12 * (size_t)nw + 8);
}

// Stage the design's strip records into LDS (geometry only; al/ga filled by set_heading).
__device__ __forceinline__ void stage_strips(const double *__restrict__ strips, int S, const Lds &l) {
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const double *rec = strips + (size_t)s * NF;
        double *o = l.sr + (size_t)s * SR_N;
        double ax = rec[RAFTX_F_AX], ay = rec[RAFTX_F_AX + 1], az = rec[RAFTX_F_AX + 2];
        o[SR_X] = rec[RAFTX_F_X];
        o[SR_X + 1] = rec[RAFTX_F_X + 1];
        o[SR_X + 2] = rec[RAFTX_F_X + 2];
        o[SR_STEP] = rec[RAFTX_F_STEP];
        o[SR_UNIT] = rec[RAFTX_F_UNIT];
        o[SR_Q] = rec[RAFTX_F_Q];
        o[SR_Q + 1] = rec[RAFTX_F_Q + 1];
        o[SR_Q + 2] = rec[RAFTX_F_Q + 2];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double *n = rec + RAFTX_F_Q + 3 * c;
            double *W = o + SR_W + c * 8;
            W[0] = n[0];
            W[1] = n[1];
            W[2] = n[2];
            W[3] = ay * n[2] - az * n[1];
            W[4] = az * n[0] - ax * n[2];
            W[5] = ax * n[1] - ay * n[0];
            W[6] = 0.0;
            W[7] = n[2];
        }
        o[SR_IQ] = rec[RAFTX_F_IQ];
        o[SR_IQ + 1] = rec[RAFTX_F_IP1];
        o[SR_IQ + 2] = rec[RAFTX_F_IP2];
        o[SR_IQ + 3] = rec[RAFTX_F_AI];
        o[SR_IQ + 4] = rec[RAFTX_F_RHOV];
        o[SR_IQ + 5] = rec[RAFTX_F_MCF];
        o[SR_DQ] = rec[RAFTX_F_DQ];
        o[SR_DQ + 1] = rec[RAFTX_F_DP1];
        o[SR_DQ + 2] = rec[RAFTX_F_DP2];
        o[SR_DQ + 3] = rec[RAFTX_F_DEND];
        o[SR_DQ + 4] = rec[RAFTX_F_CIRC];
    }
}

// al_c = n_cx cos(beta) + n_cy sin(beta) for the heading being processed
__device__ __forceinline__ void set_heading(int S, const Lds &l, double cb, double sb) {
    for (int i = threadIdx.x; i < S * 3; i += blockDim.x) {
        double *W = l.sr + (size_t)(i / 3) * SR_N + SR_W + (i % 3) * 8;
        W[6] = W[0] * cb + W[1] * sb;
    }
}

// ------------------------------------------------------------------ wave kinematics along a run
struct Kin {
    double Er, Ei;   // e^{-i k xi_s}
    double P, Q;     // e^{k z_s},  e^{-k (z_s + 2h)}
    double Rr, Ri;   // rotor e^{-i k du}
    double Rp, Rq;   // rotors e^{k dz}, e^{-k dz}
};

// Advance the per-lane kinematic state to strip `sr`; returns the velocity phasors
// t1 = A0 Ch E, t2 = i A0 Sh E (A0 = w zeta0) and the pressure ratio Cc.
__device__ __forceinline__ void kin_advance(Kin &K, const double *__restrict__ sr, const Bin &b, double cb, double sb,
                                            double A0, cplx &t1, cplx &t2, double &Cc) {
    const int m = (int)sr[SR_STEP];
    if (m == 0) {                       // run start: exact evaluation (helpers.py:201,216-222)
        double xi = cb * sr[SR_X] + sb * sr[SR_X + 1];
        fast_sincos(-(b.k * xi), K.Ei, K.Er);
        double kz = b.k * sr[SR_X + 2];
        K.P = fast_exp(kz);
        K.Q = b.e2kh * fast_exp(-kz);
        double unit = sr[SR_UNIT];
        if (unit != 0.0) {              // rotors of this run
            double du = unit * (cb * sr[SR_Q] + sb * sr[SR_Q + 1]);
            double dz = unit * sr[SR_Q + 2];
            fast_sincos(-(b.k * du), K.Ri, K.Rr);
            K.Rp = fast_exp(b.k * dz);
            K.Rq = fast_exp(-(b.k * dz));
        }
    } else {
        for (int i = 0; i < m; i++) {
            double er = K.Er * K.Rr - K.Ei * K.Ri;
            K.Ei = K.Er * K.Ri + K.Ei * K.Rr;
            K.Er = er;
            K.P *= K.Rp;
            K.Q *= K.Rq;
        }
    }
    double Sh = (K.P - K.Q) * b.csh;    // sinh k(z+h) / sinh kh
    double Ch = (K.P + K.Q) * b.csh;    // cosh k(z+h) / sinh kh
    Cc = (K.P + K.Q) * b.cch;           // cosh k(z+h) / cosh kh
    if (b.mode == 2) {                  // k == 0 (helpers.py:211-214)
        Sh = 1.0;
        Ch = 99999.0;
        Cc = 99999.0;
    }
    double ar = A0 * K.Er, ai = A0 * K.Ei;
    t1 = {ar * Ch, ai * Ch};
    t2 = {-ai * Sh, ar * Sh};
}

// ------------------------------------------------------------------ strip sweeps
// Inertial excitation of one heading (raft_member.py:1965-1991), F[6] per lane.
//   f3 = Imat ud + pDyn a_i q,  ud = i w u   ->   F += sum_c g_c W_c,
//   g_c = coef_c (i w)(al_c t1 + ga_c t2)  (+ pDyn a_i for c = q)
template <bool MCF>
__device__ __forceinline__ void inertial_excitation(const DevTables &T, int S, const Lds &l, const cplx *__restrict__ cm,
                                                    const Bin &b, int iw, double zeta0, double cb, double sb, cplx F[6]) {
#pragma unroll
    for (int j = 0; j < 6; j++) F[j] = {0.0, 0.0};
    Kin K = {1, 0, 1, 0, 1, 0, 1, 1};
    const double A0 = b.w * zeta0;
    const double rg = T.rho * T.g * zeta0;
    for (int s = 0; s < S; s++) {
        const double *__restrict__ sr = l.sr + (size_t)s * SR_N;
        cplx t1, t2;
        double Cc;
        kin_advance(K, sr, b, cb, sb, A0, t1, t2, Cc);
        cplx pd = {rg * K.Er * Cc, rg * K.Ei * Cc};            // rho g zeta_s Cc (helpers.py:231)
        const int mcf = MCF ? (int)sr[SR_IQ + 5] : -1;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double *W = sr + SR_W + c * 8;
            cplx G = {W[6] * t1.re + W[7] * t2.re, W[6] * t1.im + W[7] * t2.im};   // n_c . u
            cplx a = {-b.w * G.im, b.w * G.re};                                       // n_c . ud
            cplx g;
            if (c == 0) {
                double ai = sr[SR_IQ + 3];
                g = {sr[SR_IQ] * a.re + pd.re * ai, sr[SR_IQ] * a.im + pd.im * ai};
            } else if (mcf >= 0) {      // MacCamy-Fuchs: complex per-bin Cm (raft_member.py:1415-1420)
                cplx m = cm[((size_t)mcf * 2 + (c - 1)) * T.nw + iw];
                g = cmul(cscale(m, sr[SR_IQ + 4]), a);
            } else {
                g = cscale(a, sr[SR_IQ + c]);
            }
#pragma unroll
            for (int j = 0; j < 6; j++) {
                F[j].re = fma(g.re, W[j], F[j].re);
                F[j].im = fma(g.im, W[j], F[j].im);
            }
        }
    }
}

// Pass A of one linearisation: per strip, sums over all bins of |relative velocity|^2
// along q, p1, p2 (raft_member.py:2075-2090, helpers.py:684) -> bc[S][3].
// Cross-lane sums go through a per-wave LDS transposition tile: lanes write their
// squares column-wise, 48 lanes then add rows (no shuffles, no barriers inside the
// wave), and the 4 per-wave partials are combined once per pass.
__device__ __forceinline__ void linearize_passA(int S, const Lds &l, const Bin &b, bool active, double zeta0, double cb,
                                                double sb, const cplx Xi[6]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *tile = l.tile + (size_t)wv * SB * 3 * TROW;
    const int pos = lane + (lane >> 5);                 // skip the pad slot at 32
    const int rr = lane >> 1, rh = lane & 1;            // reducer lane -> (row, half)
    Kin K = {1, 0, 1, 0, 1, 0, 1, 1};
    const double A0 = active ? b.w * zeta0 : 0.0;
    const double wm = active ? b.w : 0.0;
    for (int s0 = 0; s0 < S; s0 += SB) {
        const int nb = min(SB, S - s0);
        for (int j = 0; j < nb; j++) {
            const double *__restrict__ sr = l.sr + (size_t)(s0 + j) * SR_N;
            cplx t1, t2;
            double Cc;
            kin_advance(K, sr, b, cb, sb, A0, t1, t2, Cc);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double *W = sr + SR_W + c * 8;
                double Gr = W[6] * t1.re + W[7] * t2.re, Gi = W[6] * t1.im + W[7] * t2.im;   // n_c . u
                double Pr = 0.0, Pi = 0.0;                                                    // W_c . Xi
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    Pr = fma(W[q], Xi[q].re, Pr);
                    Pi = fma(W[q], Xi[q].im, Pi);
                }
                double vr = fma(wm, Pi, Gr), vi = fma(-wm, Pr, Gi);                          // G - i w P
                tile[(j * 3 + c) * TROW + pos] = vr * vr + vi * vi;
            }
        }
        __syncthreads();
        if (rr < nb * 3) {
            const double *row = tile + rr * TROW + rh * 33;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int e = 0; e < 32; e += 4) {
                a0 += row[e];
                a1 += row[e + 1];
                a2 += row[e + 2];
                a3 += row[e + 3];
            }
            double a = (a0 + a1) + (a2 + a3);
            a += __shfl_xor(a, 1, 64);
            if (rh == 0) l.accw[((size_t)wv * S + s0 + rr / 3) * 3 + rr % 3] = a;
        }
        __syncthreads();
    }
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const double *sr = l.sr + (size_t)s * SR_N;
        double a = 0, c1 = 0, c2 = 0;
#pragma unroll
        for (int i = 0; i < NWAVE; i++) {
            const double *r = l.accw + ((size_t)i * S + s) * 3;
            a += r[0];
            c1 += r[1];
            c2 += r[2];
        }
        double vRq = sqrt(0.5 * a), vR1, vR2;
        if (sr[SR_DQ + 4] != 0.0) {            // circular: total transverse velocity (:2085-2087)
            vR1 = sqrt(0.5 * (c1 + c2));
            vR2 = vR1;
        } else {
            vR1 = sqrt(0.5 * c1);
            vR2 = sqrt(0.5 * c2);
        }
        l.bc[(size_t)s * 4 + 0] = sr[SR_DQ] * vRq + sr[SR_DQ + 3] * vRq;   // Bprime_q + Bprime_End (:2093,:2110)
        l.bc[(size_t)s * 4 + 1] = sr[SR_DQ + 1] * vR1;
        l.bc[(size_t)s * 4 + 2] = sr[SR_DQ + 2] * vR2;
    }
    __syncthreads();
}

// uv[s]: U = sum_c b_c al_c W_c, V = sum_c b_c ga_c W_c (current heading's al).
__device__ __forceinline__ void build_uv(int S, const Lds &l) {
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        double U[6] = {0, 0, 0, 0, 0, 0}, V[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double *W = l.sr + (size_t)s * SR_N + SR_W + c * 8;
            double bq = l.bc[(size_t)s * 4 + c];
            double al = bq * W[6], ga = bq * W[7];
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

// B_drag = sum_{s,c} b_{s,c} W_c W_c^T  (raft_member.py:2117-2118).  36 lanes, one entry each.
__device__ __forceinline__ void build_Bdrag(int S, const Lds &l) {
    int e = threadIdx.x;
    if (e < 36) {
        int i = e / 6, j = e % 6;
        double acc = 0.0;
        for (int s = 0; s < S; s++) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double *W = l.sr + (size_t)s * SR_N + SR_W + c * 8;
                acc += l.bc[(size_t)s * 4 + c] * (W[i] * W[j]);
            }
        }
        l.Bd[e] = acc;
    }
}

// Pass B: drag excitation of one heading with the live coefficients: F = sum_s t1 U_s + t2 V_s
// (raft_member.py:2122-2124, :2146-2151).
__device__ __forceinline__ void drag_excitation(int S, const Lds &l, const Bin &b, double zeta0, double cb, double sb,
                                                cplx F[6]) {
#pragma unroll
    for (int j = 0; j < 6; j++) F[j] = {0.0, 0.0};
    Kin K = {1, 0, 1, 0, 1, 0, 1, 1};
    const double A0 = b.w * zeta0;
    for (int s = 0; s < S; s++) {
        const double *__restrict__ sr = l.sr + (size_t)s * SR_N;
        cplx t1, t2;
        double Cc;
        kin_advance(K, sr, b, cb, sb, A0, t1, t2, Cc);
        const double *uv = l.uv + (size_t)s * 12;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            F[j].re = fma(t1.re, uv[j], fma(t2.re, uv[6 + j], F[j].re));
            F[j].im = fma(t1.im, uv[j], fma(t2.im, uv[6 + j], F[j].im));
        }
    }
}

// ------------------------------------------------------------------ 6x6 complex solve in registers
struct Lu6 {
    double ar[6][6], ai[6][6];
};

// x <- A^-1 x by Gaussian elimination with partial pivoting on the augmented system
// [A | x]: the pivot rule of LAPACK zgetrf/zgesv (np.linalg.solve, raft_model.py:1089):
// largest |re|+|im| in the column (izamax), full row interchange.  Everything stays in
// registers: row swaps are predicated selects (wave-uniformly skipped when no lane
// needs one), the right-hand side is eliminated together with the matrix.
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
        double d = 1.0 / (pr * pr + pi * pi);
        double ir = pr * d, ii = -pi * d;
        A.ar[k][k] = ir;            // keep the reciprocal pivot for the back substitution
        A.ai[k][k] = ii;
#pragma unroll
        for (int r = k + 1; r < 6; r++) {
            double lr = A.ar[r][k] * ir - A.ai[r][k] * ii;
            double li = A.ar[r][k] * ii + A.ai[r][k] * ir;
#pragma unroll
            for (int c = k + 1; c < 6; c++) {
                A.ar[r][c] -= lr * A.ar[k][c] - li * A.ai[k][c];
                A.ai[r][c] -= lr * A.ai[k][c] + li * A.ar[k][c];
            }
            x[r].re -= lr * x[k].re - li * x[k].im;
            x[r].im -= lr * x[k].im + li * x[k].re;
        }
    }
#pragma unroll
    for (int k = 5; k >= 0; k--) {
        cplx s = x[k];
#pragma unroll
        for (int c = k + 1; c < 6; c++) {
            s.re -= A.ar[k][c] * x[c].re - A.ai[k][c] * x[c].im;
            s.im -= A.ar[k][c] * x[c].im + A.ai[k][c] * x[c].re;
        }
        x[k] = {s.re * A.ar[k][k] - s.im * A.ai[k][k], s.re * A.ai[k][k] + s.im * A.ar[k][k]};
    }
}

// ------------------------------------------------------------------ kernels
// F_iner [nDesign,nCase,nHead,6,nw]   (raft_fowt.py:1854-1857,1888)
__global__ void __launch_bounds__(BLOCK) k_excitation(DevTables T, cplx *__restrict__ F_iner) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int pair = blockIdx.x / T.nHead, ih = blockIdx.x % T.nHead;
    const int d = pair / T.nCase, ic = pair % T.nCase;
    const int S = (int)(T.off[d + 1] - T.off[d]);
    const double *strips = T.strips + (size_t)T.off[d] * NF;
    const cplx *cm = T.cm ? T.cm + (size_t)T.cmoff[d] * 2 * T.nw : nullptr;
    Lds l = carve(smem, S, T.nw);
    const bool active = threadIdx.x < T.nw;
    const int iw = active ? threadIdx.x : 0;
    Bin b = load_bin(T, iw);
    const double beta = T.beta[(size_t)ic * T.nHead + ih];
    const double cb = cos(beta), sb = sin(beta);
    stage_strips(strips, S, l);
    __syncthreads();
    set_heading(S, l, cb, sb);
    __syncthreads();
    const double zeta0 = T.zeta[((size_t)ic * T.nHead + ih) * T.nw + iw];
    cplx F[6];
    inertial_excitation<true>(T, S, l, cm, b, iw, zeta0, cb, sb, F);
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
    Lds l = carve(smem, S, T.nw);
    const bool active = threadIdx.x < T.nw;
    const int iw = active ? threadIdx.x : 0;
    Bin b = load_bin(T, iw);
    stage_strips(strips, S, l);
    cplx Xi[6];
#pragma unroll
    for (int j = 0; j < 6; j++) Xi[j] = active ? Xi_in[((size_t)pair * 6 + j) * T.nw + iw] : cplx{0.0, 0.0};
    {
        const double beta = T.beta[(size_t)ic * T.nHead + 0];
        const double cb = cos(beta), sb = sin(beta);
        const double zeta0 = T.zeta[((size_t)ic * T.nHead + 0) * T.nw + iw];
        __syncthreads();
        set_heading(S, l, cb, sb);
        __syncthreads();
        linearize_passA(S, l, b, active, zeta0, cb, sb, Xi);
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
            const double zeta0 = T.zeta[((size_t)ic * T.nHead + ih) * T.nw + iw];
            __syncthreads();
            set_heading(S, l, cb, sb);
            __syncthreads();
            build_uv(S, l);
            __syncthreads();
            cplx F[6];
            drag_excitation(S, l, b, zeta0, cb, sb, F);
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

// per-lane parking of a 6-vector of complex numbers in LDS ([12][nw], conflict-free)
__device__ __forceinline__ void park_store(double *area, int n, int iw, bool active, const cplx v[6]) {
    if (active) {
#pragma unroll
        for (int j = 0; j < 6; j++) {
            area[(2 * j) * n + iw] = v[j].re;
            area[(2 * j + 1) * n + iw] = v[j].im;
        }
    }
}
__device__ __forceinline__ void park_load(const double *area, int n, int iw, cplx v[6]) {
#pragma unroll
    for (int j = 0; j < 6; j++) {
        v[j].re = area[(2 * j) * n + iw];
        v[j].im = area[(2 * j + 1) * n + iw];
    }
}

// kernel specialisation flags: optional inputs/outputs cost registers in the 6x6 solve,
// so the sweep path (none of them) gets its own lean instantiation
#define KF_FDEP 1    // frequency-dependent M(w), B(w)
#define KF_OUTZ 2    // export Z
#define KF_OUTF 4    // export F_wave
#define KF_EXTRA 8   // F_extra input
#define KF_MCF 16    // MacCamy-Fuchs complex Cm table
#define KF_MULTI 32  // more than one wave heading
#define KF_ALL 63
#define KF_WIDE 64   // tuning: allow 512 registers (1 wave/SIMD) instead of 256 (2 waves/SIMD)

// Assemble and solve this lane's 6x6 system: x <- Z^-1 x  (raft_model.py:1086-1089)
template <int FLAGS>
__device__ __forceinline__ void assemble_and_solve(const Lds &l, const double *__restrict__ Mw, const double *__restrict__ Bw,
                                                   int nw, int iw, double w, cplx x[6], cplx *__restrict__ Zout,
                                                   bool active) {
    Lu6 lu;
    const double w2 = w * w;
#pragma unroll
    for (int r = 0; r < 6; r++) {
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const int e = r * 6 + c;
            double M = l.mat[e], B = l.mat[36 + e];
            if constexpr ((FLAGS & KF_FDEP) != 0) {
                if (Mw) {
                    M += Mw[(size_t)e * nw + iw];
                    B += Bw[(size_t)e * nw + iw];
                }
            }
            B += l.Bd[e];
            lu.ar[r][c] = fma(-w2, M, l.mat[72 + e]);     // Z = -w^2 M + i w B + C  (:1086)
            lu.ai[r][c] = w * B;
        }
        __builtin_amdgcn_sched_barrier(0);               // assemble row by row: bounds the loads in flight
    }
    if constexpr ((FLAGS & KF_OUTZ) != 0) {
        if (active && Zout) {      // last iterate wins (fowt.Z, :1155)
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = 0; c < 6; c++) Zout[(size_t)(r * 6 + c) * nw + iw] = cplx{lu.ar[r][c], lu.ai[r][c]};
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    solve6(lu, x);
}

// The fused fixed point (raft_model.py:1052-1142) + per-heading response (:1189-1236).
// Register discipline: only XiLast (12 doubles) is carried in registers through the strip
// sweeps; F_lin lives in LDS, XiLast is parked there while the 6x6 system (72 doubles of
// matrix) is factorised; the solved Xi goes straight to HBM when the iteration ends.
template <int FLAGS>
__global__ void __launch_bounds__(BLOCK, ((FLAGS & ~KF_WIDE) == 0 && !(FLAGS & KF_WIDE) ? 2 : 1)) k_solve_dynamics(DevTables T, SolveArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr bool FDEP = (FLAGS & KF_FDEP) != 0, OUTZ = (FLAGS & KF_OUTZ) != 0, OUTF = (FLAGS & KF_OUTF) != 0;
    constexpr bool EXTRA = (FLAGS & KF_EXTRA) != 0, MCF = (FLAGS & KF_MCF) != 0, MULTI = (FLAGS & KF_MULTI) != 0;
    const int pair = blockIdx.x;
    const int d = pair / T.nCase, ic = pair % T.nCase;
    const int S = (int)(T.off[d + 1] - T.off[d]);
    const double *strips = T.strips + (size_t)T.off[d] * NF;
    // (a specialisation may be a superset of what the call needs: flags enable code, pointers decide)
    const cplx *cm = (MCF && T.cm) ? T.cm + (size_t)T.cmoff[d] * 2 * T.nw : nullptr;
    Lds l = carve(smem, S, T.nw);
    const int nw = T.nw, nH = MULTI ? T.nHead : 1;
    const int nHs = T.nHead;                       // stride of the heading axis in the arrays
    const bool active = threadIdx.x < nw;
    const int iw = active ? threadIdx.x : 0;
    const Bin b = load_bin(T, iw);
    stage_strips(strips, S, l);
    if (threadIdx.x < 108) {
        const int e = threadIdx.x % 36, wh = threadIdx.x / 36;
        const double *src = wh == 0 ? T.M0 : (wh == 1 ? T.B0 : T.C0);
        l.mat[threadIdx.x] = src[(size_t)d * 36 + e];
    }
    const double beta0 = T.beta[(size_t)ic * nHs];
    const double cb0 = cos(beta0), sb0 = sin(beta0);
    const double zeta00 = T.zeta[((size_t)ic * nHs) * nw + iw];
    __syncthreads();
    set_heading(S, l, cb0, sb0);
    __syncthreads();

    {   // F_lin = F_extra[0] + F_iner[0]   (raft_model.py:1048)
        cplx Flin[6];
        inertial_excitation<MCF>(T, S, l, cm, b, iw, zeta00, cb0, sb0, Flin);
        if constexpr (EXTRA) {
            if (A.F_extra) {
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    cplx fe = A.F_extra[(((size_t)pair * nHs) * 6 + j) * nw + iw];
                    Flin[j] = cadd(fe, Flin[j]);
                }
            }
        }
        park_store(l.flin, nw, iw, active, Flin);
    }
    cplx XiLast[6];
#pragma unroll
    for (int j = 0; j < 6; j++) XiLast[j] = {A.XiStart, 0.0};   // :999
    const double *Mw = (FDEP && T.MBw) ? T.MBw + ((size_t)d * 2 + 0) * 36 * nw : nullptr;
    const double *Bw = (FDEP && T.MBw) ? T.MBw + ((size_t)d * 2 + 1) * 36 * nw : nullptr;
    cplx *Zout = (OUTZ && A.Z) ? A.Z + (size_t)pair * 36 * nw : nullptr;

    int iiter = 0, done = 0, converged = 0, nan = 0;
#pragma unroll 1
    while (true) {
        linearize_passA(S, l, b, active, zeta00, cb0, sb0, XiLast);   // :1063 (ends with a barrier)
        park_store(l.park0, nw, iw, active, XiLast);
        build_Bdrag(S, l);
        build_uv(S, l);
        __syncthreads();
        cplx x[6];
        drag_excitation(S, l, b, zeta00, cb0, sb0, x);               // :1064
        {
            cplx Flin[6];
            park_load(l.flin, nw, iw, Flin);
#pragma unroll
            for (int j = 0; j < 6; j++) x[j] = cadd(Flin[j], x[j]);   // :1081
            if constexpr (OUTF) {                                     // total excitation, heading 0 (:1212)
                if (active && A.F_wave) {
#pragma unroll
                    for (int j = 0; j < 6; j++) A.F_wave[(((size_t)pair * nHs) * 6 + j) * nw + iw] = x[j];
                }
            }
        }
        assemble_and_solve<FLAGS>(l, Mw, Bw, nw, iw, b.w, x, Zout, active);   // :1086-1089
        park_load(l.park0, nw, iw, XiLast);
        done = iiter + 1;
        // NaN check (:1098) and convergence (:1103-1104)
        int bad = 0, ok = 1;
        if (active) {
#pragma unroll
            for (int j = 0; j < 6; j++) {
                if (isnan(x[j].re) || isnan(x[j].im)) bad = 1;
                double dr = x[j].re - XiLast[j].re, di = x[j].im - XiLast[j].im;
                double tc = sqrt(dr * dr + di * di) / (sqrt(x[j].re * x[j].re + x[j].im * x[j].im) + A.tol);
                if (!(tc < A.tol)) ok = 0;
            }
        }
        nan = __syncthreads_or(bad);
        converged = nan ? 0 : __syncthreads_and(ok);
        if (nan || converged || iiter + 1 >= A.nIter) {
            // Heading 0 of the final response, Zinv (F_lin + F_drag(0)), is exactly this solve (:1216)
            if (active && A.Xi) {
#pragma unroll
                for (int j = 0; j < 6; j++)
                    A.Xi[(((size_t)pair * nHs) * 6 + j) * nw + iw] = nan ? cplx{NAN, NAN} : x[j];
            }
            break;
        }
#pragma unroll
        for (int j = 0; j < 6; j++) {                                    // :1133
            XiLast[j].re = 0.2 * XiLast[j].re + 0.8 * x[j].re;
            XiLast[j].im = 0.2 * XiLast[j].im + 0.8 * x[j].im;
        }
        iiter++;
        __syncthreads();     // the parking area aliases the tiles pass A is about to overwrite
    }

    // remaining headings: same impedance, same linearised coefficients (:1200-1236)
    if constexpr (MULTI) {
#pragma unroll 1
        for (int ih = 1; ih < nH; ih++) {
            const double beta = T.beta[(size_t)ic * nHs + ih];
            const double cb = cos(beta), sb = sin(beta);
            const double zeta0 = T.zeta[((size_t)ic * nHs + ih) * nw + iw];
            __syncthreads();
            set_heading(S, l, cb, sb);
            __syncthreads();
            build_uv(S, l);
            __syncthreads();
            cplx x[6], f[6];
            inertial_excitation<MCF>(T, S, l, cm, b, iw, zeta0, cb, sb, x);
            if constexpr (EXTRA) {
                if (A.F_extra) {
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        cplx fe = A.F_extra[(((size_t)pair * nHs + ih) * 6 + j) * nw + iw];
                        x[j] = cadd(fe, x[j]);
                    }
                }
            }
            drag_excitation(S, l, b, zeta0, cb, sb, f);                  // :1209
#pragma unroll
            for (int j = 0; j < 6; j++) x[j] = cadd(x[j], f[j]);         // :1212
            if constexpr (OUTF) {
                if (active && A.F_wave) {
#pragma unroll
                    for (int j = 0; j < 6; j++) A.F_wave[(((size_t)pair * nHs + ih) * 6 + j) * nw + iw] = x[j];
                }
            }
            if (A.Xi) {
                assemble_and_solve<(FLAGS & ~KF_OUTZ)>(l, Mw, Bw, nw, iw, b.w, x, nullptr, active);   // Zinv @ F_wave (:1216)
                if (active) {
#pragma unroll
                    for (int j = 0; j < 6; j++)
                        A.Xi[(((size_t)pair * nHs + ih) * 6 + j) * nw + iw] = nan ? cplx{NAN, NAN} : x[j];
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
