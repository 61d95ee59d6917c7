/* raftx_port_simd.h -- TEST / BENCH INFRASTRUCTURE ONLY (included by raftx_oracle.c).
 *
 * The oracle's fixed point (solve_core above: raft/raft_model.py:1052-1142 with raft/raft_member.py:2039-2152 inside)
 * restated ONCE MORE for the CPU's vector units, as the honest CPU datapoint of bench.py's cpu_baseline leg
 * (kind "port-simd"): the same loops in the same order -- kinematics per strip, inertial excitation, then per iteration
 * the relative-velocity sums of every strip, the strip's drag matrix, the drag excitation, one pivoted 6 x 6 complex
 * solve per frequency, the convergence test, the relaxation -- but
 *   * frequency is the innermost, unit-stride loop everywhere and complex numbers are split into re / im arrays, so that
 *     gcc vectorises the sweeps (#pragma omp simd, AVX2 at -march=x86-64-v3);
 *   * |z|^2 is re^2 + im^2 instead of cabs()^2 (libm's hypot was a third of the plain oracle's time);
 *   * the per-component squares of a projection, sum_j |(v.q) q_j|^2, are |v.q|^2 sum_j q_j^2;
 *   * heading 0's response is the last iteration's solve (the plain oracle forms the explicit inverse, as the reference
 *     does for the general multi-heading case).
 * It covers what the bench's plain sweep needs and nothing else: one wave heading, real inertia coefficients (no
 * MacCamy-Fuchs rows), frequency-independent M / B, no extra excitation, no restart point.  The checker stays
 * solve_core(); this function is compared with it (<= 1e-12, equal iteration counts) by tests/test_oracle_golden.py and
 * inside bench.py before its time is reported.  It never enters a parity claim by itself. */

static inline void *simd_alloc(size_t n_doubles) {
    void *p = NULL;
    if (posix_memalign(&p, 64, sizeof(double) * (n_doubles ? n_doubles : 8))) return NULL;
    return p;
}

/* 6 x 6 complex solve, zgetrf / zgetrs order, on split re / im storage; returns 1 if singular */
static inline int simd_solve6(double *Ar, double *Ai, double *br, double *bi) {
    for (int k = 0; k < 6; k++) {
        int p = k;
        double best = fabs(Ar[k * 6 + k]) + fabs(Ai[k * 6 + k]);
        for (int i = k + 1; i < 6; i++) {
            const double v = fabs(Ar[i * 6 + k]) + fabs(Ai[i * 6 + k]);
            if (v > best) { best = v; p = i; }
        }
        if (best == 0.0) return 1;
        if (p != k) {
            for (int j = 0; j < 6; j++) {
                double t = Ar[k * 6 + j]; Ar[k * 6 + j] = Ar[p * 6 + j]; Ar[p * 6 + j] = t;
                t = Ai[k * 6 + j]; Ai[k * 6 + j] = Ai[p * 6 + j]; Ai[p * 6 + j] = t;
            }
            double t = br[k]; br[k] = br[p]; br[p] = t;
            t = bi[k]; bi[k] = bi[p]; bi[p] = t;
        }
        const double pr = Ar[k * 6 + k], pi = Ai[k * 6 + k], d = 1.0 / (pr * pr + pi * pi);
        const double ir = pr * d, ii = -pi * d;                    /* 1 / pivot */
        for (int i = k + 1; i < 6; i++) {
            const double ar = Ar[i * 6 + k], ai = Ai[i * 6 + k];
            const double lr = ar * ir - ai * ii, li = ar * ii + ai * ir;
            for (int j = k + 1; j < 6; j++) {
                Ar[i * 6 + j] -= lr * Ar[k * 6 + j] - li * Ai[k * 6 + j];
                Ai[i * 6 + j] -= lr * Ai[k * 6 + j] + li * Ar[k * 6 + j];
            }
            br[i] -= lr * br[k] - li * bi[k];
            bi[i] -= lr * bi[k] + li * br[k];
        }
    }
    for (int k = 5; k >= 0; k--) {
        double sr = br[k], si = bi[k];
        for (int j = k + 1; j < 6; j++) {
            sr -= Ar[k * 6 + j] * br[j] - Ai[k * 6 + j] * bi[j];
            si -= Ar[k * 6 + j] * bi[j] + Ai[k * 6 + j] * br[j];
        }
        const double pr = Ar[k * 6 + k], pi = Ai[k * 6 + k], d = 1.0 / (pr * pr + pi * pi);
        br[k] = (sr * pr + si * pi) * d;
        bi[k] = (si * pr - sr * pi) * d;
    }
    return 0;
}

/* (the library is built with -ffp-contract=off so that the checker's arithmetic is the reference's; this one function may
 * fuse multiply-adds, as any production CPU build would) */
__attribute__((optimize("-ffp-contract=fast")))
int raftx_oracle_solve_simd(raftx_ctx *c, int nIter_in, double tol, double XiStart) {
    if (check_ready(c)) return -1;
    if (c->nHead != 1 || c->cm || c->MBw || c->bem_ready || c->have_xl0 || c->want_xlout)
        FAIL(c, "solve_simd covers the plain sweep only (one heading, no MacCamy-Fuchs rows, no M(w) / B(w), no extra excitation)");
    const int nw = c->nw, nIter = nIter_in + 1;
    const long npair = (long)c->nDesign * c->nCase;
    const size_t np_ = (size_t)npair, nx = np_ * 6 * nw;
    free_results(c);
    c->rmask = 0;
    c->rXi = (c128 *)malloc(sizeof(c128) * (nx ? nx : 1));
    c->rNi = (int32_t *)malloc(sizeof(int32_t) * (np_ ? np_ : 1));
    c->rFl = (int32_t *)malloc(sizeof(int32_t) * (np_ ? np_ : 1));
    int maxS = 0;
    for (int d = 0; d < c->nDesign; d++) {
        const int S = (int)(c->off[d + 1] - c->off[d]);
        if (S > maxS) maxS = S;
    }
    const double t0 = now_ms();
    int bad_alloc = 0;
#pragma omp parallel
    {
        /* per-thread work arrays, frequency innermost: u [S][3] re / im; F_lin, F_drag, XiLast, Xi [6] re / im */
        double *ur = (double *)simd_alloc((size_t)maxS * 3 * nw), *ui = (double *)simd_alloc((size_t)maxS * 3 * nw);
        double *W = (double *)simd_alloc((size_t)8 * 6 * nw);
        double *Bm = (double *)simd_alloc((size_t)(maxS > 0 ? maxS : 1) * 9);
        if (!ur || !ui || !W || !Bm) {
#pragma omp atomic write
            bad_alloc = 1;
        }
        double *Flr = W, *Fli = W + 6 * nw, *Fdr = W + 12 * nw, *Fdi = W + 18 * nw, *Xlr = W + 24 * nw, *Xli = W + 30 * nw,
               *Xr = W + 36 * nw, *Xim = W + 42 * nw;
#pragma omp for schedule(dynamic)
        for (long p = 0; p < npair; p++) {
            if (bad_alloc) continue;
            const int d = (int)(p / c->nCase), ic = (int)(p % c->nCase);
            const int S = (int)(c->off[d + 1] - c->off[d]);
            const double *strips = c->strips + (size_t)c->off[d] * RAFTX_NFIELD;
            const double *M0 = c->M0 + (size_t)d * 36, *B0 = c->B0 + (size_t)d * 36, *C0 = c->C0 + (size_t)d * 36;
            const double *zeta = c->zeta + (size_t)ic * nw;
            const double beta = c->beta[ic], cb = cos(beta), sb = sin(beta), h = c->depth;
            const double *w = c->w, *k = c->k;
            for (int a = 0; a < 6 * nw; a++) { Flr[a] = 0.0; Fli[a] = 0.0; Xlr[a] = XiStart; Xli[a] = 0.0; }
            /* ---- kinematics (helpers.py:188-236) and inertial excitation (raft_member.py:1965-1991), strip by strip */
            for (int s = 0; s < S; s++) {
                const double *rec = strips + (size_t)s * RAFTX_NFIELD;
                const double *r = rec + RAFTX_F_X, *q = rec + RAFTX_F_Q, *p1 = rec + RAFTX_F_P1, *p2 = rec + RAFTX_F_P2, *arm = rec + RAFTX_F_AX;
                double *u0r = ur + ((size_t)s * 3 + 0) * nw, *u1r = u0r + nw, *u2r = u1r + nw;
                double *u0i = ui + ((size_t)s * 3 + 0) * nw, *u1i = u0i + nw, *u2i = u1i + nw;
                const double z = r[2], xi = cb * r[0] + sb * r[1];
                double Im[3][3] = {{0}};
                vvt(q, rec[RAFTX_F_IQ], Im);
                vvt(p1, rec[RAFTX_F_IP1], Im);
                vvt(p2, rec[RAFTX_F_IP2], Im);
                const double ai_ = rec[RAFTX_F_AI];
                for (int i = 0; i < nw; i++) {
                    double Sh = 0.0, Ch = 0.0, Cc = 0.0;
                    const double kk = k[i];
                    if (z <= 0) {
                        if (kk == 0.0) { Sh = 1.0; Ch = 99999.0; Cc = 99999.0; }
                        else if (kk * h > 89.4) { Sh = exp(kk * z); Ch = Sh; Cc = Sh + exp(-kk * (z + 2.0 * h)); }
                        else { const double sh = sinh(kk * h); Sh = sinh(kk * (z + h)) / sh; Ch = cosh(kk * (z + h)) / sh; Cc = cosh(kk * (z + h)) / cosh(kk * h); }
                    }
                    const double ph = -(kk * xi), zr = zeta[i] * cos(ph), zi = zeta[i] * sin(ph);
                    const double wi = w[i];
                    u0r[i] = wi * zr * Ch * cb; u0i[i] = wi * zi * Ch * cb;
                    u1r[i] = wi * zr * Ch * sb; u1i[i] = wi * zi * Ch * sb;
                    u2r[i] = -(wi * zi * Sh);   u2i[i] = wi * zr * Sh;          /* i w zeta Sh */
                    /* pDyn = rho g zeta Cc;  ud = i w u;  F3 = Imat ud + pDyn a_i q */
                    const double pr = c->rho * c->g * zr * Cc, pi = c->rho * c->g * zi * Cc;
                    const double dr_[3] = {-wi * u0i[i], -wi * u1i[i], -wi * u2i[i]}, di_[3] = {wi * u0r[i], wi * u1r[i], wi * u2r[i]};
                    double f3r[3], f3i[3];
                    for (int a = 0; a < 3; a++) {
                        f3r[a] = Im[a][0] * dr_[0] + Im[a][1] * dr_[1] + Im[a][2] * dr_[2] + pr * ai_ * q[a];
                        f3i[a] = Im[a][0] * di_[0] + Im[a][1] * di_[1] + Im[a][2] * di_[2] + pi * ai_ * q[a];
                    }
                    Flr[0 * nw + i] += f3r[0]; Fli[0 * nw + i] += f3i[0];
                    Flr[1 * nw + i] += f3r[1]; Fli[1 * nw + i] += f3i[1];
                    Flr[2 * nw + i] += f3r[2]; Fli[2 * nw + i] += f3i[2];
                    Flr[3 * nw + i] += arm[1] * f3r[2] - arm[2] * f3r[1]; Fli[3 * nw + i] += arm[1] * f3i[2] - arm[2] * f3i[1];
                    Flr[4 * nw + i] += arm[2] * f3r[0] - arm[0] * f3r[2]; Fli[4 * nw + i] += arm[2] * f3i[0] - arm[0] * f3i[2];
                    Flr[5 * nw + i] += arm[0] * f3r[1] - arm[1] * f3r[0]; Fli[5 * nw + i] += arm[0] * f3i[1] - arm[1] * f3i[0];
                }
            }
            /* ---- the fixed point (raft_model.py:1052-1142) */
            int iiter = 0, converged = 0, nan = 0, done_iters = 0;
            double Bd[36];
            while (iiter < nIter) {
                memset(Bd, 0, sizeof(Bd));
                for (int a = 0; a < 6 * nw; a++) { Fdr[a] = 0.0; Fdi[a] = 0.0; }
                for (int s = 0; s < S; s++) {
                    const double *rec = strips + (size_t)s * RAFTX_NFIELD;
                    const double *q = rec + RAFTX_F_Q, *p1 = rec + RAFTX_F_P1, *p2 = rec + RAFTX_F_P2, *r = rec + RAFTX_F_AX;
                    const double *u0r = ur + ((size_t)s * 3 + 0) * nw, *u1r = u0r + nw, *u2r = u1r + nw;
                    const double *u0i = ui + ((size_t)s * 3 + 0) * nw, *u1i = u0i + nw, *u2i = u1i + nw;
                    const int circ = rec[RAFTX_F_CIRC] != 0.0;
                    const double qq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], p1p1 = p1[0] * p1[0] + p1[1] * p1[1] + p1[2] * p1[2],
                                 p2p2 = p2[0] * p2[0] + p2[1] * p2[1] + p2[2] * p2[2];
                    double sq = 0.0, sp = 0.0, sp1 = 0.0, sp2 = 0.0;
                    const double r0 = r[0], r1 = r[1], r2 = r[2], q0 = q[0], q1 = q[1], q2 = q[2];
                    const double a0 = p1[0], a1 = p1[1], a2 = p1[2], b0 = p2[0], b1 = p2[1], b2 = p2[2];
                    /* relative velocity u - i w (Xi_t + theta x r) and its projections (raft_member.py:2075-2087) */
#pragma omp simd reduction(+ : sq, sp, sp1, sp2)
                    for (int i = 0; i < nw; i++) {
                        const double wi = w[i];
                        const double t0r = Xlr[3 * nw + i], t0i = Xli[3 * nw + i], t1r = Xlr[4 * nw + i], t1i = Xli[4 * nw + i],
                                     t2r = Xlr[5 * nw + i], t2i = Xli[5 * nw + i];
                        const double d0r = Xlr[0 * nw + i] + (t1r * r2 - t2r * r1), d0i = Xli[0 * nw + i] + (t1i * r2 - t2i * r1);
                        const double d1r = Xlr[1 * nw + i] + (t2r * r0 - t0r * r2), d1i = Xli[1 * nw + i] + (t2i * r0 - t0i * r2);
                        const double d2r = Xlr[2 * nw + i] + (t0r * r1 - t1r * r0), d2i = Xli[2 * nw + i] + (t0i * r1 - t1i * r0);
                        const double v0r = u0r[i] + wi * d0i, v0i = u0i[i] - wi * d0r;       /* u - i w dr */
                        const double v1r = u1r[i] + wi * d1i, v1i = u1i[i] - wi * d1r;
                        const double v2r = u2r[i] + wi * d2i, v2i = u2i[i] - wi * d2r;
                        const double vqr = v0r * q0 + v1r * q1 + v2r * q2, vqi = v0i * q0 + v1i * q1 + v2i * q2;
                        sq += (vqr * vqr + vqi * vqi) * qq;
                        if (circ) {
                            const double e0r = v0r - vqr * q0, e0i = v0i - vqi * q0, e1r = v1r - vqr * q1, e1i = v1i - vqi * q1,
                                         e2r = v2r - vqr * q2, e2i = v2i - vqi * q2;
                            sp += e0r * e0r + e0i * e0i + e1r * e1r + e1i * e1i + e2r * e2r + e2i * e2i;
                        } else {
                            const double ar = v0r * a0 + v1r * a1 + v2r * a2, ai = v0i * a0 + v1i * a1 + v2i * a2;
                            const double br = v0r * b0 + v1r * b1 + v2r * b2, bi = v0i * b0 + v1i * b1 + v2i * b2;
                            sp1 += (ar * ar + ai * ai) * p1p1;
                            sp2 += (br * br + bi * bi) * p2p2;
                        }
                    }
                    const double vq_ = sqrt(0.5 * sq), vp1_ = circ ? sqrt(0.5 * sp) : sqrt(0.5 * sp1), vp2_ = circ ? vp1_ : sqrt(0.5 * sp2);
                    double B3[3][3] = {{0}};
                    vvt(q, rec[RAFTX_F_DQ] * vq_, B3);                           /* raft_member.py:2093-2098 */
                    vvt(p1, rec[RAFTX_F_DP1] * vp1_, B3);
                    vvt(p2, rec[RAFTX_F_DP2] * vp2_, B3);
                    vvt(q, rec[RAFTX_F_DEND] * vq_, B3);                         /* :2110-2113 */
                    memcpy(Bm + (size_t)s * 9, B3, sizeof(B3));
                    translate_matrix_add(B3, r, Bd);                             /* :2118 */
                    /* drag excitation of this strip (raft_member.py:2146-2151) */
                    const double m00 = B3[0][0], m01 = B3[0][1], m02 = B3[0][2], m10 = B3[1][0], m11 = B3[1][1], m12 = B3[1][2],
                                 m20 = B3[2][0], m21 = B3[2][1], m22 = B3[2][2];
#pragma omp simd
                    for (int i = 0; i < nw; i++) {
                        const double f0r = m00 * u0r[i] + m01 * u1r[i] + m02 * u2r[i], f0i = m00 * u0i[i] + m01 * u1i[i] + m02 * u2i[i];
                        const double f1r = m10 * u0r[i] + m11 * u1r[i] + m12 * u2r[i], f1i = m10 * u0i[i] + m11 * u1i[i] + m12 * u2i[i];
                        const double f2r = m20 * u0r[i] + m21 * u1r[i] + m22 * u2r[i], f2i = m20 * u0i[i] + m21 * u1i[i] + m22 * u2i[i];
                        Fdr[0 * nw + i] += f0r; Fdi[0 * nw + i] += f0i;
                        Fdr[1 * nw + i] += f1r; Fdi[1 * nw + i] += f1i;
                        Fdr[2 * nw + i] += f2r; Fdi[2 * nw + i] += f2i;
                        Fdr[3 * nw + i] += r1 * f2r - r2 * f1r; Fdi[3 * nw + i] += r1 * f2i - r2 * f1i;
                        Fdr[4 * nw + i] += r2 * f0r - r0 * f2r; Fdi[4 * nw + i] += r2 * f0i - r0 * f2i;
                        Fdr[5 * nw + i] += r0 * f1r - r1 * f0r; Fdi[5 * nw + i] += r0 * f1i - r1 * f0i;
                    }
                }
                /* Z(w) = -w^2 M + i w (B + B_drag) + C, one pivoted solve per frequency (raft_model.py:1084-1089) */
                for (int i = 0; i < nw; i++) {
                    double Ar[36], Ai[36], br[6], bi[6];
                    const double wi = w[i], w2 = wi * wi;
                    for (int a = 0; a < 36; a++) {
                        Ar[a] = C0[a] - w2 * M0[a];
                        Ai[a] = wi * (B0[a] + Bd[a]);
                    }
                    for (int j = 0; j < 6; j++) { br[j] = Flr[j * nw + i] + Fdr[j * nw + i]; bi[j] = Fli[j * nw + i] + Fdi[j * nw + i]; }
                    if (simd_solve6(Ar, Ai, br, bi)) for (int j = 0; j < 6; j++) br[j] = bi[j] = NAN;
                    for (int j = 0; j < 6; j++) { Xr[j * nw + i] = br[j]; Xim[j * nw + i] = bi[j]; }
                }
                done_iters = iiter + 1;
                int ok = 1;
                for (int a = 0; a < 6 * nw; a++) {
                    if (isnan(Xr[a]) || isnan(Xim[a])) nan = 1;
                    const double er = Xr[a] - Xlr[a], ei = Xim[a] - Xli[a];
                    const double tc = sqrt(er * er + ei * ei) / (sqrt(Xr[a] * Xr[a] + Xim[a] * Xim[a]) + tol);      /* :1103-1104 */
                    if (!(tc < tol)) ok = 0;
                }
                if (nan) break;
                if (ok) { converged = 1; break; }
#pragma omp simd
                for (int a = 0; a < 6 * nw; a++) { Xlr[a] = 0.2 * Xlr[a] + 0.8 * Xr[a]; Xli[a] = 0.2 * Xli[a] + 0.8 * Xim[a]; }   /* :1133 */
                iiter++;
            }
            c128 *out = c->rXi + (size_t)p * 6 * nw;
            for (int a = 0; a < 6 * nw; a++) out[a] = nan ? (c128)NAN : Xr[a] + I * Xim[a];
            c->rNi[p] = done_iters;
            c->rFl[p] = (converged ? RAFTX_FLAG_CONVERGED : 0) | (nan ? RAFTX_FLAG_NAN : 0);
        }
        free(ur); free(ui); free(W); free(Bm);
    }
    c->last_ms = now_ms() - t0;
    if (bad_alloc) FAIL(c, "solve_simd: out of memory");
    return 0;
}
