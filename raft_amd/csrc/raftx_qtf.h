// raftx_qtf.h -- gfx950 kernels of the second-order slender-body QTF (included by raftx_hip.hip).
//
// Reference: raft/raft_member.py:1488-1674 (Member.calcQTF_slenderBody), raft/raft_fowt.py:2044-2070
// (Pinkster IV term, member loop, Hermitian fill), raft/helpers.py:239-373 (gradient / second-order
// potential helpers), raft/helpers.py:149-236 (getKinematics, getWaveKin).
//
// Two launches per batch of "sets" (one set = one strip table + motion RAOs + heading):
//   k_qtf_tables : per (strip | member | set, frequency) the first-order quantities every pair needs
//                  (wave velocity, its gradient, body displacement/velocity, pressure gradient ...),
//                  evaluated ONCE per frequency with full-precision device libm; 24 complex per strip-bin.
//   k_qtf_pairs  : one thread per (w1, w2 >= w1) pair, one workgroup row per w1; loops the strips, reads
//                  the two table columns (the w1 column is wave-uniform, the w2 column coalesced) and
//                  evaluates the bilinear Rainey / Pinkster terms, the second-order-potential term
//                  (two exp per strip-pair) and the waterline term; writes Q[i1,i2] and conj to Q[i2,i1].
// The reference's quirks are kept (see oracle/qtf_oracle.py, which documents each one).
#pragma once

#define QS_N 24        // doubles per QTF strip record (raft_amd/qtf.py)
#define QM_N 16        // doubles per QTF member record
#define QT_N 26        // complex table entries per (strip, frequency)
#define QTM_N 12       // complex table entries per (member, frequency)
#define QTS_N 12       // complex table entries per (set, frequency)
#define QD_N 32        // real per-strip constants derived by the table kernel

// Strip table fields.  Every vector / matrix is stored in the STRIP'S OWN FRAME E = [p1 p2 q] (v' = E^T v,
// M' = E^T M E): the projections of the Rainey terms are then component selections -- P_Ca v = (Ca1 v_1, Ca2 v_2, 0),
// "remove the axial part" = drop component 3, q . v = v_3 -- and the pair kernel is left with the bilinear products
// alone (tests/qtf_device_model.py is the numpy statement of the formulation, checked against the term-by-term
// restatement of the reference in oracle/qtf_oracle.py).
#define QT_U 0         // u[3]
#define QT_DR 3        // dr[3]
#define QT_X 6         // (u - nodeV_t)[1..2]  (nodeV_t = node velocity with its axial part removed; component 3 of the
                       // difference never enters: every use is behind a transverse projection.  The transverse relative
                       // velocity of axdivAcc, (u - (q.u) q) - nodeV_t, has the same two components and a zero third)
#define QT_GP 8        // grad_pres1st[3]
#define QT_NAR 11      // nodeV_axial_rel
#define QT_DWDZ 12     // q . grad_u q
#define QT_E2 13       // exp(-i k (cos(deg2rad b) x + sin(deg2rad b) y)): phase of the 2nd-order potential
#define QT_G 14        // E^T grad_u E, row-major 3 x 3 (grad_u with the reference's twisted third row, helpers.py:274)
#define QT_S 23        // E^T [i w theta]x E: entries 01, 02, 12 of the antisymmetric matrix
// per-strip constants (k_qtf_tables, thread of frequency 0)
#define QD_CA 0        // Ca_p1, Ca_p2
#define QD_RV 2        // rho * v_side
#define QD_RE 3        // rho * v_end * Ca_End
#define QD_AI 4        // a_i
#define QD_Z 5
#define QD_HX 6        // cos(deg2rad b) E[0][i] + sin(deg2rad b) E[1][i], i = 0..2: horizontal direction of the 2nd-order potential
#define QD_HZ 9        // E[2][i]
#define QD_W 12        // [3][6]: [e_i ; r x e_i], the strip-frame force components as 6-DOF loads about the origin
// member table fields
#define QTM_UD 0       // ud_wl[3]
#define QTM_ETAR 3     // eta_r
#define QTM_AB 4       // body acceleration at the waterline [3]
#define QTM_GE 7       // g_e1[3]
// set table fields
#define QTS_TH 0       // theta[3]
#define QTS_F1 3       // F1st[6]

struct c3 {
    cplx x, y, z;
};
__device__ __forceinline__ cplx cconj(cplx a) { return {a.re, -a.im}; }
__device__ __forceinline__ cplx cmuli(cplx a) { return {-a.im, a.re}; }                 // i a
__device__ __forceinline__ cplx cfma(cplx a, cplx b, cplx c) { return {fma(a.re, b.re, fma(-a.im, b.im, c.re)), fma(a.re, b.im, fma(a.im, b.re, c.im))}; }
__device__ __forceinline__ c3 c3add(c3 a, c3 b) { return {cadd(a.x, b.x), cadd(a.y, b.y), cadd(a.z, b.z)}; }
__device__ __forceinline__ c3 c3sub(c3 a, c3 b) { return {csub(a.x, b.x), csub(a.y, b.y), csub(a.z, b.z)}; }
__device__ __forceinline__ c3 c3scale(c3 a, double s) { return {cscale(a.x, s), cscale(a.y, s), cscale(a.z, s)}; }
__device__ __forceinline__ c3 c3cmul(c3 a, cplx s) { return {cmul(a.x, s), cmul(a.y, s), cmul(a.z, s)}; }
__device__ __forceinline__ c3 c3conj(c3 a) { return {cconj(a.x), cconj(a.y), cconj(a.z)}; }
__device__ __forceinline__ cplx rdot(const double *n, c3 v) {                          // real n . complex v
    return {n[0] * v.x.re + n[1] * v.y.re + n[2] * v.z.re, n[0] * v.x.im + n[1] * v.y.im + n[2] * v.z.im};
}
__device__ __forceinline__ cplx cdot(c3 a, c3 b) {                                     // sum a_i b_i (no conjugation)
    return cadd(cadd(cmul(a.x, b.x), cmul(a.y, b.y)), cmul(a.z, b.z));
}
__device__ __forceinline__ c3 rvec(const double *n, cplx s) { return {cscale(s, n[0]), cscale(s, n[1]), cscale(s, n[2])}; }
__device__ __forceinline__ c3 ccross(c3 a, c3 b) {
    return {csub(cmul(a.y, b.z), cmul(a.z, b.y)), csub(cmul(a.z, b.x), cmul(a.x, b.z)), csub(cmul(a.x, b.y), cmul(a.y, b.x))};
}
// symmetric-with-a-twist gradient matrix of helpers.py:239-277: [[g00,g01,g02],[g01,g11,g12],[g02,g01,g22]]
struct g6 {
    cplx g00, g01, g02, g11, g12, g22;
};
__device__ __forceinline__ c3 gmul(const g6 &G, c3 v) {
    return {cadd(cadd(cmul(G.g00, v.x), cmul(G.g01, v.y)), cmul(G.g02, v.z)),
            cadd(cadd(cmul(G.g01, v.x), cmul(G.g11, v.y)), cmul(G.g12, v.z)),
            cadd(cadd(cmul(G.g02, v.x), cmul(G.g01, v.y)), cmul(G.g22, v.z))};
}
__device__ __forceinline__ g6 gconj(const g6 &G) { return {cconj(G.g00), cconj(G.g01), cconj(G.g02), cconj(G.g11), cconj(G.g12), cconj(G.g22)}; }
__device__ __forceinline__ g6 gscale(const g6 &G, cplx s) { return {cmul(G.g00, s), cmul(G.g01, s), cmul(G.g02, s), cmul(G.g11, s), cmul(G.g12, s), cmul(G.g22, s)}; }
// P v = a1 (p1.v) p1 + a2 (p2.v) p2
__device__ __forceinline__ c3 proj2(const double *p1, const double *p2, double a1, double a2, c3 v) {
    cplx s1 = cscale(rdot(p1, v), a1), s2 = cscale(rdot(p2, v), a2);
    return c3add(rvec(p1, s1), rvec(p2, s2));
}

struct QtfArgs {
    int nSet, nw;
    const double *__restrict__ w, *__restrict__ k;
    double depth, rho, g;
    const int64_t *__restrict__ soff;      // [nSet+1]
    const double *__restrict__ strips;     // [nStrip,QS_N]
    const int64_t *__restrict__ moff;      // [nSet+1]
    const double *__restrict__ members;    // [nMem,QM_N]
    const int *__restrict__ sset;          // [nStrip] set of each strip
    const int *__restrict__ mset;          // [nMem]   set of each member
    const cplx *__restrict__ Xi;           // [nSet,6,nw]
    const double *__restrict__ beta;       // [nSet]
    const double *__restrict__ Ms;         // [nSet,36]
    const cplx *__restrict__ kay;          // [nSet,nw,nw,6] or null
    cplx *T;                               // [nStrip,QT_N,nw]   field-major: the w2 column of a field is one coalesced load
    cplx *TA;                              // [nStrip,nw,QT_N]   the same values frequency-major: the w1 row of a strip is 416
                                           //                    contiguous bytes for the scalar loads of the pair kernel
    double *D;                             // [nStrip,QD_N]
    cplx *TM;                              // [nMem,QTM_N,nw]
    cplx *TS;                              // [nSet,QTS_N,nw]
    cplx *qtf;                             // [nSet,nw,nw,6]
    int row_off, row_stride, nrow;         // this call computes rows w1 = row_off + m*row_stride (m < nrow): the interleaved
                                           // row partition of one QTF over ranks (row cost ~ nw - i1: triangular)
};

// getWaveKin with zeta0 = 1 (helpers.py:188-236): u, and Cc for the pressure/elevation
__device__ __forceinline__ void qtf_wavekin(double w, double k, double h, double beta, const double *r, c3 &u, cplx &zCc) {
    double s, c;
    sincos(-(k * (cos(beta) * r[0] + sin(beta) * r[1])), &s, &c);
    const cplx zeta = {c, s};
    const double z = r[2];
    u = {{0, 0}, {0, 0}, {0, 0}};
    zCc = {0, 0};
    if (z <= 0) {
        double Sh, Ch, Cc;
        if (k == 0.0) {
            Sh = 1.0; Ch = 99999.0; Cc = 99999.0;
        } else if (k * h > 89.4) {
            Sh = exp(k * z); Ch = Sh; Cc = exp(k * z) + exp(-k * (z + 2.0 * h));
        } else {
            Sh = sinh(k * (z + h)) / sinh(k * h);
            Ch = cosh(k * (z + h)) / sinh(k * h);
            Cc = cosh(k * (z + h)) / cosh(k * h);
        }
        u.x = cscale(zeta, w * Ch * cos(beta));
        u.y = cscale(zeta, w * Ch * sin(beta));
        u.z = cmuli(cscale(zeta, w * Sh));
        zCc = cscale(zeta, Cc);
    }
}

__device__ __forceinline__ c3 load3(const cplx *T, int f, int nw, int i) { return {T[(size_t)f * nw + i], T[(size_t)(f + 1) * nw + i], T[(size_t)(f + 2) * nw + i]}; }
__device__ __forceinline__ void store3(cplx *T, int f, int nw, int i, c3 v) {
    T[(size_t)f * nw + i] = v.x;
    T[(size_t)(f + 1) * nw + i] = v.y;
    T[(size_t)(f + 2) * nw + i] = v.z;
}

// ---- first-order tables: grid.x = nStrip + nMem + nSet, threads over frequency
__global__ void __launch_bounds__(256) k_qtf_tables(QtfArgs A, int nStrip, int nMem) {
    const int b = blockIdx.x, nw = A.nw;
    for (int i = threadIdx.x; i < nw; i += blockDim.x) {
        const double w = A.w[i], k = A.k[i], h = A.depth;
        if (b < nStrip) {
            const double *rec = A.strips + (size_t)b * QS_N;
            const int set = A.sset[b];
            const double beta = A.beta[set];
            const double r[3] = {rec[0], rec[1], rec[2]}, q[3] = {rec[3], rec[4], rec[5]};
            const cplx *X = A.Xi + (size_t)set * 6 * nw;
            c3 xt = {X[0 * nw + i], X[1 * nw + i], X[2 * nw + i]}, th = {X[3 * nw + i], X[4 * nw + i], X[5 * nw + i]};
            // dr = Xi_t + th x r (helpers.py:178, 396-402); nodeV = i w dr
            c3 rr = {{r[0], 0}, {r[1], 0}, {r[2], 0}};
            c3 dr = c3add(xt, ccross(th, rr));
            c3 nodeV = c3cmul(dr, cplx{0.0, w});
            c3 u;
            cplx zc;
            qtf_wavekin(w, k, h, beta, r, u, zc);
            // grad_u1 (helpers.py:239-277)
            g6 G = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
            c3 gp = {{0, 0}, {0, 0}, {0, 0}};
            cplx e2 = {1.0, 0.0};
            const double cB = cos(beta * (M_PI / 180.0)), sB = sin(beta * (M_PI / 180.0));        // (sic) deg2rad of a radian heading
            if (r[2] <= 0 && k > 0) {
                double xy, zz, pxy, pzz;
                if (k * h >= 10) {
                    xy = zz = pxy = pzz = exp(k * r[2]);
                } else {
                    xy = cosh(k * (r[2] + h)) / sinh(k * h);
                    zz = sinh(k * (r[2] + h)) / sinh(k * h);
                    pxy = cosh(k * (r[2] + h)) / cosh(k * h);
                    pzz = sinh(k * (r[2] + h)) / cosh(k * h);
                }
                double s, c;
                sincos(-(k * (cos(beta) * r[0] + sin(beta) * r[1])), &s, &c);
                const cplx ph = {c, s};
                cplx aux = cscale(ph, w * cB);
                G.g00 = cscale(cmuli(aux), -xy * k * cB);
                G.g01 = cscale(cmuli(aux), -xy * k * sB);
                G.g02 = cscale(aux, k * zz);
                aux = cscale(ph, w * sB);
                G.g11 = cscale(cmuli(aux), -xy * k * sB);
                G.g12 = cscale(aux, k * zz);
                aux = cmuli(cscale(ph, w));
                G.g22 = cscale(aux, k * xy);
                // grad_pres1st (helpers.py:283-308): deg2rad'ed heading also in the phase
                sincos(-(k * (cB * r[0] + sB * r[1])), &s, &c);
                const cplx ph2 = {c, s};
                const double rg = A.rho * A.g;
                gp.x = cscale(cmuli(ph2), -rg * pxy * k * cB);
                gp.y = cscale(cmuli(ph2), -rg * pxy * k * sB);
                gp.z = cscale(ph2, rg * pzz * k);
            }
            {
                double s, c;
                sincos(-(k * (cB * r[0] + sB * r[1])), &s, &c);                                   // helpers.py:343-361 phase factor
                e2 = {c, s};
            }
            const cplx nar = rdot(q, c3sub(u, nodeV));
            const c3 nodeVt = c3sub(nodeV, rvec(q, rdot(q, nodeV)));
            const cplx dwdz = rdot(q, gmul(G, c3{{q[0], 0}, {q[1], 0}, {q[2], 0}}));
            const c3 ua = c3sub(u, nodeVt);
            // into the strip's frame
            const double *e[3] = {rec + 6, rec + 9, q};                       // p1, p2, q
            cplx *T = A.T + (size_t)b * QT_N * nw;
            cplx *TA = A.TA + ((size_t)b * nw + i) * QT_N;
            auto put = [&](int f, cplx v) {
                T[(size_t)f * nw + i] = v;
                TA[f] = v;
            };
            const c3 Om = c3cmul(th, cplx{0.0, w});                           // i w theta: OMEGA v = Om x v (:1588-1589)
            for (int a = 0; a < 3; a++) {
                put(QT_U + a, rdot(e[a], u));
                put(QT_DR + a, rdot(e[a], dr));
                put(QT_GP + a, rdot(e[a], gp));
                if (a < 2) put(QT_X + a, rdot(e[a], ua));
                for (int c = 0; c < 3; c++) {
                    const c3 ec = {{e[c][0], 0}, {e[c][1], 0}, {e[c][2], 0}};
                    put(QT_G + 3 * a + c, rdot(e[a], gmul(G, ec)));
                    if (c > a) put(QT_S + (a == 0 ? c - 1 : 2), rdot(e[a], ccross(Om, ec)));
                }
            }
            put(QT_NAR, nar);
            put(QT_DWDZ, dwdz);
            put(QT_E2, e2);
            if (i == 0) {
                double *D = A.D + (size_t)b * QD_N;
                D[QD_CA] = rec[12]; D[QD_CA + 1] = rec[13];
                D[QD_RV] = A.rho * rec[15];
                D[QD_RE] = A.rho * rec[16] * rec[14];
                D[QD_AI] = rec[17];
                D[QD_Z] = r[2];
                for (int a = 0; a < 3; a++) {
                    D[QD_HX + a] = cB * e[a][0] + sB * e[a][1];
                    D[QD_HZ + a] = e[a][2];
                    double *W = D + QD_W + 6 * a;
                    W[0] = e[a][0]; W[1] = e[a][1]; W[2] = e[a][2];
                    W[3] = r[1] * e[a][2] - r[2] * e[a][1];
                    W[4] = r[2] * e[a][0] - r[0] * e[a][2];
                    W[5] = r[0] * e[a][1] - r[1] * e[a][0];
                }
            }
        } else if (b < nStrip + nMem) {
            const int m = b - nStrip;
            const double *rec = A.members + (size_t)m * QM_N;
            const int set = A.mset[m];
            cplx *T = A.TM + (size_t)m * QTM_N * nw;
            if (rec[0] != 0.0) {
                const double beta = A.beta[set];
                const double r[3] = {rec[1], rec[2], rec[3]};
                const double *p1 = rec + 7, *p2 = rec + 10;
                const cplx *X = A.Xi + (size_t)set * 6 * nw;
                c3 xt = {X[0 * nw + i], X[1 * nw + i], X[2 * nw + i]}, th = {X[3 * nw + i], X[4 * nw + i], X[5 * nw + i]};
                c3 rr = {{r[0], 0}, {r[1], 0}, {r[2], 0}};
                c3 dr = c3add(xt, ccross(th, rr));
                c3 u;
                cplx eta;
                qtf_wavekin(w, k, h, beta, r, u, eta);                     // rho = g = 1: pDyn is the elevation (:1525)
                c3 ud = c3cmul(u, cplx{0.0, w});
                c3 ab = c3scale(dr, -w * w);
                // g_e1 = -g ( (th x p1)_z p1 + (th x p2)_z p2 )   (:1531-1532)
                cplx c1 = csub(cscale(th.x, p1[1]), cscale(th.y, p1[0])), c2 = csub(cscale(th.x, p2[1]), cscale(th.y, p2[0]));
                c3 ge = c3scale(c3add(rvec(p1, c1), rvec(p2, c2)), -A.g);
                store3(T, QTM_UD, nw, i, ud);
                T[(size_t)QTM_ETAR * nw + i] = csub(eta, dr.z);
                store3(T, QTM_AB, nw, i, ab);
                store3(T, QTM_GE, nw, i, ge);
            } else {
                for (int f = 0; f < QTM_N; f++) T[(size_t)f * nw + i] = {0, 0};
            }
        } else {
            const int set = b - nStrip - nMem;
            const cplx *X = A.Xi + (size_t)set * 6 * nw;
            cplx *T = A.TS + (size_t)set * QTS_N * nw;
            const double *M = A.Ms + (size_t)set * 36;
            for (int j = 0; j < 3; j++) T[(size_t)(QTS_TH + j) * nw + i] = X[(3 + j) * nw + i];
            for (int r = 0; r < 6; r++) {                                  // F1st = M_struc (-w^2 Xi)  (raft_fowt.py:2045)
                cplx a = {0, 0};
                for (int c = 0; c < 6; c++) a = cadd(a, cscale(X[c * nw + i], -w * w * M[r * 6 + c]));
                T[(size_t)(QTS_F1 + r) * nw + i] = a;
            }
        }
    }
}

// ---- pair kernel: grid (nSet * nw) rows of w1; threads stride w2 >= w1
__global__ void __launch_bounds__(128) k_qtf_pairs(QtfArgs A) {
    const int nw = A.nw;
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MiB L2; a set's table is 4.4 MB at C5.  Every XCD
    // therefore gets a CONTIGUOUS slab of the (set, row) list -- whole sets when there are at least 8 of them -- instead of
    // rows of every set: the w2 columns it streams then stay in its L2 (measured: the set-interleaved order ran at the
    // fabric's bandwidth, not the VALU's).  Inside a set the rows alternate between the long and the short end of the
    // triangle, so that any part of a slab carries the same work.
    const int total = A.nSet * A.nrow, per = (total + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (item >= total || (int)(blockIdx.x >> 3) >= per) return;
    const int set = item / A.nrow, m_ = item % A.nrow;
    const int i1 = A.row_off + ((m_ & 1) ? A.nrow - 1 - (m_ >> 1) : (m_ >> 1)) * A.row_stride;
    const double h = A.depth, rho = A.rho, g = A.g;
    const double w1 = A.w[i1], k1 = A.k[i1];
    const double beta = A.beta[set];
    const double cB = cos(beta * (M_PI / 180.0)), sB = sin(beta * (M_PI / 180.0));
    const cplx *TS = A.TS + (size_t)set * QTS_N * nw;
    const c3 th1 = load3(TS, QTS_TH, nw, i1);
    for (int i2 = i1 + threadIdx.x; i2 < nw; i2 += blockDim.x) {
        const double w2 = A.w[i2], k2 = A.k[i2];
        const c3 th2 = load3(TS, QTS_TH, nw, i2);
        cplx F[6] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
        // Pinkster IV (raft_fowt.py:2053-2062)
        {
            c3 fa1 = load3(TS, QTS_F1, nw, i1), fb1 = load3(TS, QTS_F1 + 3, nw, i1);
            c3 fa2 = load3(TS, QTS_F1, nw, i2), fb2 = load3(TS, QTS_F1 + 3, nw, i2);
            c3 t = c3scale(c3add(ccross(th1, c3conj(fa2)), ccross(c3conj(th2), fa1)), 0.25);
            c3 m = c3scale(c3add(ccross(th1, c3conj(fb2)), ccross(c3conj(th2), fb1)), 0.25);
            F[0] = t.x; F[1] = t.y; F[2] = t.z; F[3] = m.x; F[4] = m.y; F[5] = m.z;
        }
        // second-order potential coefficients of this pair (helpers.py:337-358)
        const bool pot = (w1 != w2) && (k1 > 0) && (k2 > 0);
        double nrm = 0.0, kx = 0.0, ky = 0.0;
        cplx paux = {0, 0};
        if (pot) {
            kx = (k1 - k2) * cB;
            ky = (k1 - k2) * sB;
            nrm = sqrt(kx * kx + ky * ky);
            const double t1 = tanh(k1 * h), t2 = tanh(k2 * h);
            const double den = (w1 - w2) * (w1 - w2) / g - nrm * tanh(nrm * h);
            // gamma = (-i g / (2 w)) * num / den  ->  purely imaginary
            const double g12 = (-g / (2 * w1)) * ((k1 * k1) * (1 - t1 * t1) - 2 * k1 * k2 * (1 + t1 * t2)) / den;
            const double g21 = (-g / (2 * w2)) * ((k2 * k2) * (1 - t2 * t2) - 2 * k2 * k1 * (1 + t2 * t1)) / den;
            paux = {0.0, 0.5 * (g21 - g12)};                               // 0.5 (gamma_21 + conj(gamma_12))
        }
        // ---- strip terms (raft_member.py:1541-1633) in each strip's own frame, see the field list above and
        // tests/qtf_device_model.py.  The row frequency's fields (index 1) are wave-uniform: scalar loads, scalar operands.
        double zprev = __builtin_nan(""), xdk = 0.0, znr = 0.0, xr = 0.0;     // depth-dependent factors of the 2nd-order potential
        const double dwp = w1 - w2, dk = k1 - k2;
        double rden = 0.0;
        if (pot) rden = 1.0 / (1.0 + exp(-2.0 * (nrm * h)));
        for (int64_t s = A.soff[set]; s < A.soff[set + 1]; s++) {
            cdptr D = as_const(A.D + (size_t)s * QD_N);
            const cplx *T2 = A.T + (size_t)s * QT_N * nw + i2;
            cdptr T1 = as_const(reinterpret_cast<const double *>(A.TA + ((size_t)s * nw + i1) * QT_N));
#define U1(f) (cplx{T1[2 * (f)], T1[2 * (f) + 1]})
#define L2(f) (T2[(size_t)(f) * nw])
            const double Ca0 = D[QD_CA], Ca1 = D[QD_CA + 1], rv = D[QD_RV], a_i = D[QD_AI], z = D[QD_Z];
            // second-order potential: acceleration and pressure (helpers.py:360-372); cosh / sinh of nrm (z + h) over
            // cosh(nrm h) in decaying exponentials, re-evaluated only when the depth changes (wave-uniform branch)
            if (z != zprev) {
                zprev = z;
                xdk = znr = xr = 0.0;
                if (pot && z <= 0) {
                    const double P = exp(nrm * z), Q = exp(-(nrm * (z + 2.0 * h)));
                    const double xy = (P + Q) * rden, zz = (P - Q) * rden;
                    xdk = xy * dk;
                    znr = zz * nrm;
                    xr = xy * rho;
                }
            }
            cplx A3[3], p2nd;
            {
                const cplx e1 = U1(QT_E2), e2c = cconj(L2(QT_E2));
                const cplx ph = cmul(e1, e2c);                                            // e^{-i (k1-k2)(cB x + sB y)}
                const cplx bw = {-(paux.im * ph.im) * dwp, (paux.im * ph.re) * dwp};      // paux (purely imaginary) * ph * (w1 - w2)
#pragma unroll
                for (int a_ = 0; a_ < 3; a_++) {
                    const double cr = xdk * D[QD_HX + a_], ci = znr * D[QD_HZ + a_];
                    A3[a_] = {bw.re * cr - bw.im * ci, bw.re * ci + bw.im * cr};
                }
                p2nd = {bw.im * xr, -(bw.re * xr)};                                       // -i * base * xy * rho * (w1 - w2)
            }
            // convective acceleration (:1575) + body motion in the wave field (:1582) through one product per side:
            // G1 conj(u2 - i w1 dr2) + conj(G2) (u1 - i w2 dr1)
            {
                cplx av[3], bv[3];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const cplx u2 = L2(QT_U + j), d2 = L2(QT_DR + j), u1 = U1(QT_U + j), d1 = U1(QT_DR + j);
                    av[j] = {fma(w1, d2.im, u2.re), fma(w1, d2.re, -u2.im)};
                    bv[j] = {fma(w2, d1.im, u1.re), fma(-w2, d1.re, u1.im)};
                }
#pragma unroll
                for (int a_ = 0; a_ < 3; a_++) {
                    cplx acc = {0, 0};
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        acc = cfma(U1(QT_G + 3 * a_ + j), av[j], acc);
                        acc = cfma(cconj(L2(QT_G + 3 * a_ + j)), bv[j], acc);
                    }
                    A3[a_] = {fma(0.25, acc.re, A3[a_].re), fma(0.25, acc.im, A3[a_].im)};
                }
            }
            const cplx nar1 = U1(QT_NAR), nar2 = L2(QT_NAR), dz1 = U1(QT_DWDZ), dz2c = cconj(L2(QT_DWDZ));
            const cplx x1[2] = {U1(QT_X), U1(QT_X + 1)}, x2[2] = {L2(QT_X), L2(QT_X + 1)};
            const cplx s01_1 = U1(QT_S), s01_2 = L2(QT_S);
            // off-diagonal entries of M = G + S (the Rainey terms aux - aux2 leave only (Ca_other - Ca_i) M_i,other x_other)
            const cplx m1[2] = {cadd(U1(QT_G + 1), s01_1), csub(U1(QT_G + 3), s01_1)};                 // M1_01, M1_10
            const cplx m2c[2] = {cconj(cadd(L2(QT_G + 1), s01_2)), cconj(csub(L2(QT_G + 3), s01_2))};  // conj(M2_01), conj(M2_10)
            const double Ca[2] = {Ca0, Ca1};
            cplx f[3];
#pragma unroll
            for (int a_ = 0; a_ < 2; a_++) {
                const int o = 1 - a_;
                const cplx x2c = cconj(x2[a_]), x2oc = cconj(x2[o]);
                const cplx t = cfma(m1[a_], x2oc, cmul(m2c[a_], x1[o]));                              // M1_io conj(x2_o) + conj(M2_io) x1_o
                const cplx ax0 = cfma(x2c, dz1, cmul(x1[a_], dz2c));                                  // axial-divergence (helpers.py:311-335)
                const cplx rs = cfma(U1(QT_S + 1 + a_), cconj(nar2), cmul(cconj(L2(QT_S + 1 + a_)), nar1));   // Rainey rotation (:1587-1590)
                const cplx en = cfma(cconj(x1[a_]), nar2, cmul(x2[a_], cconj(nar1)));                 // end term of :1626
                const double c_t = 0.25 * (Ca[o] - Ca[a_]), c_ax = 0.25 * Ca[a_], c_rs = -0.5 * Ca[a_], c_A = 1.0 + Ca[a_];
                cplx g = {c_A * A3[a_].re, c_A * A3[a_].im};
                g = {fma(c_ax, ax0.re, g.re), fma(c_ax, ax0.im, g.im)};
                g = {fma(c_rs, rs.re, g.re), fma(c_rs, rs.im, g.im)};
                g = {fma(c_t, t.re, g.re), fma(c_t, t.im, g.im)};
                const double c_en = 0.25 * a_i * rho * Ca[a_];
                f[a_] = {fma(c_en, en.re, rv * g.re), fma(c_en, en.im, rv * g.im)};
            }
            {
                // end effects (:1611-1627): axial component
                cplx pn = {0, 0};
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    pn = cfma(U1(QT_GP + j), cconj(L2(QT_DR + j)), pn);
                    pn = cfma(cconj(L2(QT_GP + j)), U1(QT_DR + j), pn);
                }
                cplx pd = cscale(cmul(x1[0], cconj(x2[0])), Ca0);
                pd = cadd(pd, cscale(cmul(x1[1], cconj(x2[1])), Ca1));
                const double re_ = D[QD_RE], c_pd = -0.25 * rho;
                cplx qs = {p2nd.re + 0.25 * pn.re + c_pd * pd.re, p2nd.im + 0.25 * pn.im + c_pd * pd.im};
                f[2] = {fma(re_, A3[2].re, a_i * qs.re), fma(re_, A3[2].im, a_i * qs.im)};
            }
#pragma unroll
            for (int a_ = 0; a_ < 3; a_++)
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const double wj = D[QD_W + 6 * a_ + j];
                    F[j] = {fma(wj, f[a_].re, F[j].re), fma(wj, f[a_].im, F[j].im)};
                }
#undef U1
#undef L2
        }
        // waterline term (:1635-1668)
        for (int64_t m = A.moff[set]; m < A.moff[set + 1]; m++) {
            const double *rec = A.members + (size_t)m * QM_N;
            if (rec[0] == 0.0) continue;
            const double r[3] = {rec[1], rec[2], rec[3]};
            const double a_wl = rec[4], Ca1 = rec[5], Ca2 = rec[6];
            const double *p1 = rec + 7, *p2 = rec + 10;
            const cplx *T = A.TM + (size_t)m * QTM_N * nw;
            const c3 ud1 = load3(T, QTM_UD, nw, i1), ud2c = c3conj(load3(T, QTM_UD, nw, i2));
            const cplx e1 = T[(size_t)QTM_ETAR * nw + i1], e2c = cconj(T[(size_t)QTM_ETAR * nw + i2]);
            const c3 ab1 = load3(T, QTM_AB, nw, i1), ab2c = c3conj(load3(T, QTM_AB, nw, i2));
            const c3 ge1 = load3(T, QTM_GE, nw, i1), ge2c = c3conj(load3(T, QTM_GE, nw, i2));
            c3 f = proj2(p1, p2, 1.0 + Ca1, 1.0 + Ca2, c3scale(c3add(c3cmul(ud1, e2c), c3cmul(ud2c, e1)), 0.25));
            f = c3sub(f, proj2(p1, p2, Ca1, Ca2, c3scale(c3add(c3cmul(ab1, e2c), c3cmul(ab2c, e1)), 0.25)));
            f = c3sub(f, c3scale(c3add(c3cmul(ge1, e2c), c3cmul(ge2c, e1)), 0.25));
            f = c3scale(f, rho * a_wl);
            F[0] = cadd(F[0], f.x); F[1] = cadd(F[1], f.y); F[2] = cadd(F[2], f.z);
            F[3] = cadd(F[3], csub(cscale(f.z, r[1]), cscale(f.y, r[2])));
            F[4] = cadd(F[4], csub(cscale(f.x, r[2]), cscale(f.z, r[0])));
            F[5] = cadd(F[5], csub(cscale(f.y, r[0]), cscale(f.x, r[1])));
        }
        // Kim & Yue table, then Hermitian completion (raft_fowt.py:2069-2070)
        cplx *out = A.qtf + (size_t)set * nw * nw * 6;
        for (int j = 0; j < 6; j++) {
            cplx v = F[j];
            if (A.kay) v = cadd(v, A.kay[(((size_t)set * nw + i1) * nw + i2) * 6 + j]);
            out[((size_t)i1 * nw + i2) * 6 + j] = v;
            if (i2 != i1) out[((size_t)i2 * nw + i1) * 6 + j] = cconj(v);
        }
    }
}


// ---- second-order force amplitudes from the QTF (raft_fowt.py:2209-2245): grid (nSet * 6), threads stride the
// difference-frequency index mu.  Interpolation indices / weights of the first-order bins on the second-order grid
// are built once per workgroup in LDS (np.searchsorted - 1 clipped to [0, n-2], RegularGridInterpolator rules).
__global__ void __launch_bounds__(256) k_qtf_force(int nSet, int nw2, const double *__restrict__ w2,
                                                   const cplx *__restrict__ qtf, int nw, const double *__restrict__ w,
                                                   double dw, const double *__restrict__ S0, double *__restrict__ f_mean,
                                                   double *__restrict__ f) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *tw = smem;                                    // [nw] norm. distance, < 0 marks "outside the grid"
    int *ix = reinterpret_cast<int *>(smem + nw);        // [nw]
    double *red = smem + nw + (nw + 1) / 2;               // [4]
    const int set = blockIdx.x / 6, j = blockIdx.x % 6;
    for (int i = threadIdx.x; i < nw; i += blockDim.x) {
        const double x = w[i];
        int lo = 0, hi = nw2;                             // searchsorted(w2, x, side='left')
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (w2[mid] < x) lo = mid + 1; else hi = mid;
        }
        int k = lo - 1;
        k = k < 0 ? 0 : (k > nw2 - 2 ? nw2 - 2 : k);
        ix[i] = k;
        tw[i] = (x < w2[0] || x > w2[nw2 - 1]) ? -1.0 : (x - w2[k]) / (w2[k + 1] - w2[k]);
    }
    __syncthreads();
    const cplx *Q = qtf + (size_t)set * nw2 * nw2 * 6 + j;
    const double *S = S0 + (size_t)set * nw;
    auto interp = [&](int a, int b) -> cplx {             // Q_j(w_a, w_b)
        const double ta = tw[a], tb = tw[b];
        if (ta < 0.0 || tb < 0.0) return cplx{0.0, 0.0};
        const int ia = ix[a], ib = ix[b];
        const cplx q00 = Q[((size_t)ia * nw2 + ib) * 6], q01 = Q[((size_t)ia * nw2 + ib + 1) * 6];
        const cplx q10 = Q[((size_t)(ia + 1) * nw2 + ib) * 6], q11 = Q[((size_t)(ia + 1) * nw2 + ib + 1) * 6];
        const double w00 = (1.0 - ta) * (1.0 - tb), w01 = (1.0 - ta) * tb, w10 = ta * (1.0 - tb), w11 = ta * tb;
        return cplx{w00 * q00.re + w01 * q01.re + w10 * q10.re + w11 * q11.re,
                    w00 * q00.im + w01 * q01.im + w10 * q10.im + w11 * q11.im};
    };
    double *fo = f + ((size_t)set * 6 + j) * nw;
    for (int mu = 1 + threadIdx.x; mu < nw; mu += blockDim.x) {
        double acc = 0.0;
        for (int i = 0; i + mu < nw; i++) {
            const cplx q = interp(i, i + mu);
            acc += S[i] * S[i + mu] * (q.re * q.re + q.im * q.im);
        }
        fo[mu - 1] = 4.0 * sqrt(acc) * dw;                // written one bin lower: the shift of :2241-2245
    }
    if (threadIdx.x == 0) fo[nw - 1] = 0.0;
    double m = 0.0;
    for (int i = threadIdx.x; i < nw; i += blockDim.x) m += S[i] * interp(i, i).re;
    for (int o = 32; o > 0; o >>= 1) m += __shfl_xor(m, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int q = 0; q < (int)(blockDim.x >> 6); q++) a += red[q];
        f_mean[(size_t)set * 6 + j] = 2.0 * a * dw;
    }
}


// ---------------------------------------------------------------------------------------------------------
// Kim & Yue correction (Member.correction_KAY, raft_member.py:1676-1791) as a device table.
//   k_kay_tables : per (item, frequency) the Hankel-derivative values H'_n(k R), n = 0 .. Nm+1
//   k_kay_pairs  : per (set, w1 row) threads over w2 >= w1: the n-sums for every item of the set
#define QK_N 12
#define KAY_MAXN 12          // Nm + 2 <= 12 stored orders (Nm = 10 upstream)
#define KAY_ROWS (KAY_MAXN + 3)
// Everything of the correction that depends on ONE frequency is tabulated per (item, frequency), so that the pair loop has
// no transcendental function and no division (it used to evaluate four sinh, two cosh, two tanh, two sqrt, a sincos and 22
// IEEE divisions per item and pair -- on the C5 deck the correction cost more than the QTF it corrects):
//   rows 0 .. Nm+1   1 / H'_n(k R)                                  (the pair needs Im of products of these)
//   row KAY_MAXN     (e^{k (z1 + h)}, e^{-k (z1 + h)})              sinh((k1 +- k2)(z + h)) = (E1 E2^{+-1} - 1 / ...) / 2 as
//   row KAY_MAXN+1   (e^{k (z2 + h)}, e^{-k (z2 + h)})              products of these: no cancellation between large terms
//   row KAY_MAXN+2   e^{-i k xi},  xi = cos(beta) x + sin(beta) y   (phase factor of the pair = product with a conjugate)
// and per frequency alone  kq = k h / (sqrt(k h tanh k h) cosh k h)  (raft_member.py:1757-1760: "pre / cosh / cosh").
__global__ void __launch_bounds__(128) k_kay_tables(int nw, int nOrd, int nSet, double h, const double *__restrict__ k,
                                                    const int64_t *__restrict__ ioff, const double *__restrict__ items,
                                                    const double *__restrict__ beta, cplx *__restrict__ HK,
                                                    double *__restrict__ kq) {
    const int item = blockIdx.x;
    const double *rec = items + (size_t)item * QK_N;
    const double R = rec[0];
    int set = 0;
    while (set + 1 < nSet && ioff[set + 1] <= item) set++;
    const double xi = cos(beta[set]) * rec[10] + sin(beta[set]) * rec[11];
    cplx *H = HK + (size_t)item * KAY_ROWS * nw;
    for (int i = threadIdx.x; i < nw; i += blockDim.x) {
        const double x = k[i] * R;
        // H_n = J_n + i Y_n for n = -1 .. nOrd; H_{-1} = -H_1
        double jm = -j1(x), ym = -y1(x);              // n - 1
        double jc = j0(x), yc = y0(x);                // n
        for (int n = 0; n < nOrd; n++) {
            const double jp = jn(n + 1, x), yp = yn(n + 1, x);
            const double hr = 0.5 * (jm - jp), hi = 0.5 * (ym - yp), d = hr * hr + hi * hi;     // H'_n
            H[(size_t)n * nw + i] = cplx{hr / d, -hi / d};
            jm = jc; ym = yc;
            jc = jp; yc = yp;
        }
        const double a1 = k[i] * (rec[2] + h), a2 = k[i] * (rec[3] + h);
        H[(size_t)KAY_MAXN * nw + i] = cplx{exp(a1), exp(-a1)};
        H[(size_t)(KAY_MAXN + 1) * nw + i] = cplx{exp(a2), exp(-a2)};
        double ps, pc;
        sincos(-(k[i] * xi), &ps, &pc);
        H[(size_t)(KAY_MAXN + 2) * nw + i] = cplx{pc, ps};
        if (item == 0) {
            const double kh = k[i] * h;
            kq[i] = kh / (sqrt(kh * tanh(kh)) * cosh(kh));
        }
    }
}

__global__ void __launch_bounds__(128) k_kay_pairs(int nw, int Nm, double h, double rho, double g,
                                                   const double *__restrict__ w, const double *__restrict__ k,
                                                   const int64_t *__restrict__ ioff, const double *__restrict__ items,
                                                   const cplx *__restrict__ HK, const double *__restrict__ kq,
                                                   cplx *__restrict__ kay) {
    const int set = blockIdx.x / nw, i1 = blockIdx.x % nw;
    const double w1 = w[i1], k1 = k[i1], q1 = kq[i1];
    for (int i2 = i1 + threadIdx.x; i2 < nw; i2 += blockDim.x) {
        const double w2 = w[i2], k2 = k[i2];
        const double k1h = k1 * h, k2h = k2 * h, rsum = 1.0 / (k1h + k2h), rdif = (w1 == w2) ? 0.0 : 1.0 / (k1h - k2h);
        const double qq = q1 * kq[i2], rk = 1.0 / (k1 * k2);
        double Fr[6] = {0, 0, 0, 0, 0, 0}, Fi[6] = {0, 0, 0, 0, 0, 0};
        for (int64_t it = ioff[set]; it < ioff[set + 1]; it++) {
            const double *rec = items + (size_t)it * QK_N;
            const double R = rec[0];
            const bool seg = rec[1] != 0.0;
            const cplx *H1 = HK + (size_t)it * KAY_ROWS * nw + i1, *H2 = HK + (size_t)it * KAY_ROWS * nw + i2;
            double wgtA = 1.0, wgtB = 0.0;             // term_n = omega_n * (wgtA + wgtB n(n+1))
            if (seg) {
                // sinh((k1 + k2)(z + h)) / (k1h + k2h) and sinh((k1 - k2)(z + h)) / (k1h - k2h) at z1, z2 (:1752-1756)
                const cplx e1a = H1[(size_t)KAY_MAXN * nw], e2a = H2[(size_t)KAY_MAXN * nw];
                const cplx e1b = H1[(size_t)(KAY_MAXN + 1) * nw], e2b = H2[(size_t)(KAY_MAXN + 1) * nw];
                const double sp1 = 0.5 * (e1a.re * e2a.re - e1a.im * e2a.im) * rsum, sp2 = 0.5 * (e1b.re * e2b.re - e1b.im * e2b.im) * rsum;
                const double sm1 = (w1 == w2) ? (rec[2] + h) / h : 0.5 * (e1a.re * e2a.im - e1a.im * e2a.re) * rdif;
                const double sm2 = (w1 == w2) ? (rec[3] + h) / h : 0.5 * (e1b.re * e2b.im - e1b.im * e2b.re) * rdif;
                const double Im = 0.5 * (sp2 - sm2 - sp1 + sm1), Ip = 0.5 * (sp2 + sm2 - sp1 - sm1);
                wgtA = qq * Im;
                wgtB = qq * Ip * rk / (R * R);
            }
            // sum_n (+-) rho g R 2i/pi/(k1R k2R) omega_n weight_n ; only the real part is kept (:1747,1783):
            // Re(2i sum) = -2 Im(sum),  omega_n = 1 / (H'_{n+1}(k1R) conj H'_n(k2R)) - 1 / (H'_n(k1R) conj H'_{n+1}(k2R))
            double si = 0.0;
            cplx a0 = H1[0], b0 = H2[0];                                                  // reciprocals of H'_n
            for (int n = 0; n <= Nm; n++) {
                const cplx a1 = H1[(size_t)(n + 1) * nw], b1 = H2[(size_t)(n + 1) * nw];
                // Im(a1 conj b0) - Im(a0 conj b1)
                const double oii = (a1.im * b0.re - a1.re * b0.im) - (a0.im * b1.re - a0.re * b1.im);
                si = fma(oii, wgtA + wgtB * (double)(n * (n + 1)), si);
                a0 = a1;
                b0 = b1;
            }
            double Fs = rho * g / M_PI * rk / R * (-2.0 * si);
            if (!seg) Fs = -Fs;                                                          // waterline term carries the minus sign (:1745)
            const cplx p1 = H1[(size_t)(KAY_MAXN + 2) * nw], p2 = H2[(size_t)(KAY_MAXN + 2) * nw];
            const double pc = p1.re * p2.re + p1.im * p2.im, ps = p1.im * p2.re - p1.re * p2.im;     // e^{-i (k1 - k2) xi}
            const double fr = Fs * pc, fi = Fs * ps;
            const double px = rec[7], py = rec[8], pz = rec[9], ax = rec[4], ay = rec[5], az = rec[6];
            const double g6[6] = {px, py, pz, ay * pz - az * py, az * px - ax * pz, ax * py - ay * px};
#pragma unroll
            for (int j = 0; j < 6; j++) { Fr[j] += fr * g6[j]; Fi[j] += fi * g6[j]; }
        }
        const bool cj = k1 < k2;                                                          // :1787-1788
        cplx *o = kay + (((size_t)set * nw + i1) * nw + i2) * 6;
#pragma unroll
        for (int j = 0; j < 6; j++) o[j] = cplx{Fr[j], cj ? -Fi[j] : Fi[j]};
    }
}


// Motion RAOs of the resident first-order responses on the second-order grid: RAO = Xi / zeta where |zeta| > 1e-6
// (helpers.py:762-784), then np.interp(w2, w, RAO, left=0, right=0) per DOF (raft_fowt.py:2022-2024).  Set = (design,
// case) pair, heading 0.
__global__ void __launch_bounds__(128) k_rao_to_grid(int nCase, int nHead, int nw, int nw2, const double *__restrict__ w,
                                                     const double *__restrict__ zeta, const double *__restrict__ w2,
                                                     const cplx *__restrict__ Xi, cplx *__restrict__ out) {
    const int p = blockIdx.x / 6, j = blockIdx.x % 6, ic = p % nCase;
    const cplx *x = Xi + (((size_t)p * nHead + 0) * 6 + j) * nw;
    const double *z = zeta + ((size_t)ic * nHead + 0) * nw;
    for (int i2 = threadIdx.x; i2 < nw2; i2 += blockDim.x) {
        const double t = w2[i2];
        cplx r = {0.0, 0.0};
        if (t >= w[0] && t <= w[nw - 1]) {
            int lo = 0, hi = nw - 1;                        // largest lo with w[lo] <= t
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (w[mid] <= t) lo = mid; else hi = mid;
            }
            if (w[hi] <= t) lo = hi;
            auto rao = [&](int i) -> cplx {
                const double zz = z[i];
                return fabs(zz) > 1e-6 ? cplx{x[i].re / zz, x[i].im / zz} : cplx{0.0, 0.0};
            };
            const cplx a = rao(lo);
            if (lo == nw - 1 || w[lo] == t) {
                r = a;
            } else {
                const cplx b = rao(lo + 1);
                const double dx = w[lo + 1] - w[lo], f = t - w[lo];
                r = cplx{(b.re - a.re) / dx * f + a.re, (b.im - a.im) / dx * f + a.im};      // numpy: slope*(x - xp) + fp
            }
        }
        out[((size_t)p * 6 + j) * nw2 + i2] = r;
    }
}
