// raftx_geom.h -- geometry -> strip tables + statics on the device (raftx_build_designs, include/raftx.h).
//
// What the reference does per design in Python objects -- Member.__init__ strip discretisation
// (raft/raft_member.py:190-271), Member.setPosition (:312-377), Member.calcHydroConstants / calcImat / getCmSides
// (:1261-1486), the drag areas of Member.calcHydroLinearization (:2061-2110), Member.getHydrostatics (:838-1010),
// FOWT.calcHydroConstants (raft/raft_fowt.py:1589-1625) and the hydrostatic part of FOWT.calcStatics
// (:811-1201) -- runs here as five small kernels over ALL designs of a sweep at once:
//
//   k_geom_member   one thread per member : pose (q, p1, p2, R, end A), wet-strip count, member hydrostatics
//   k_geom_scan     one workgroup         : exclusive scans of the wet / MacCamy-Fuchs strip counts
//   k_geom_fill     one wavefront / member: lanes = strips; compacted with ballots into the design's strip table
//   k_geom_mcf      (row, bin)            : MacCamy-Fuchs complex Cm table (Hankel functions)
//   k_geom_design   one thread per design : run detection for the rotor recurrences (the same routine the host
//                                           upload path uses), Morison added mass, hydrostatic reduction
//
// The work is tiny next to the solve (a 10k-design sweep has ~110k members / ~530k strips); it exists to remove
// the host packing and the 256 B/strip upload from the sweep's critical path, not to reach a roofline.
// All arithmetic is fp64 with contraction OFF, so that wet/dry decisions and strip constants agree with the
// reference's NumPy arithmetic to the last bits wherever libm agrees.
#pragma once

#define MP_N 24      // member pose scratch: rA0(3) rA(3) q(3) p1(3) p2(3) R00 R01 R10 R11 L
#define MH_N 48      // member hydrostatics: Cmat(36) Fvec(6) V rcV(3) AWP  (about the member's own node)

// ---------------------------------------------------------------------------------------------------------
// Run detection shared by raftx_upload_designs (host) and raftx_build_designs (device): turns the ABI strip
// records [i0,i1) of one design into the device records (DS_*) and flags (DSI_*).  Straight runs of equally
// spaced strips are found from the absolute positions alone, so that the kernels can advance the wave
// kinematics along a run with rotors instead of re-evaluating sincos/exp per strip.
__host__ __device__ inline void derive_design_tables(const double *strips, int64_t i0, int64_t i1, double *dsv,
                                                     int *dsf) {
    int64_t s = i0;
    while (s < i1) {
        // maximal collinear sequence [s, e): same q, displacement along +q; at most 64 strips per run
        int64_t e = s + 1;
        double proj[64];
        int np = 0;
        while (e < i1 && (e - s) < 64) {
            const double *pr = strips + (size_t)(e - 1) * NF, *cr = strips + (size_t)e * NF;
            bool same = true;
            double dv[3], pj = 0.0;
            for (int j = 0; j < 3; j++) {
                same = same && (pr[RAFTX_F_Q + j] == cr[RAFTX_F_Q + j]);
                dv[j] = cr[RAFTX_F_X + j] - pr[RAFTX_F_X + j];
                pj += dv[j] * cr[RAFTX_F_Q + j];
            }
            if (!same || !(pj > 0.0) || !(fabs(pj) <= 1.797e308)) break;
            double perp2 = 0.0, scale = 1.0;
            for (int j = 0; j < 3; j++) {
                double t = dv[j] - pj * cr[RAFTX_F_Q + j];
                perp2 += t * t;
                scale += fabs(cr[RAFTX_F_X + j]);
            }
            if (sqrt(perp2) > 1e-10 * scale) break;
            proj[np++] = pj;
            e++;
        }
        double unit = 0.0;
        for (int i = 0; i < np; i++) unit = (unit == 0.0 || proj[i] < unit) ? proj[i] : unit;
        // emit records; break the run wherever a step is not 1 or 2 units (re-anchored exactly)
        for (int64_t i = s; i < e; i++) {
            const double *rec = strips + (size_t)i * NF;
            double *o = dsv + (size_t)i * DS_N;
            int m = 0;
            if (i > s && unit > 0.0) {
                double ratio = proj[i - s - 1] / unit;
                int mi = (int)floor(ratio + 0.5);
                if (mi >= 1 && mi <= 2 && fabs(ratio - mi) < 1e-9) {
                    const double *pr = strips + (size_t)(i - 1) * NF;     // verify the prediction from the previous strip
                    bool ok = true;
                    for (int j = 0; j < 3; j++) {
                        double pred = pr[RAFTX_F_X + j] + (double)mi * unit * rec[RAFTX_F_Q + j];
                        if (fabs(pred - rec[RAFTX_F_X + j]) > 1e-10 * (1.0 + fabs(rec[RAFTX_F_X + j]))) ok = false;
                    }
                    if (ok) m = mi;
                }
            }
            dsf[(size_t)i] = m | (rec[RAFTX_F_CIRC] != 0.0 ? DSI_CIRC : 0);
            for (int j = 0; j < DS_N; j++) o[j] = 0.0;
            o[DS_MCF] = rec[RAFTX_F_MCF];
            for (int j = 0; j < 3; j++) {
                o[DS_X + j] = rec[RAFTX_F_X + j];
                o[DS_U + j] = unit * rec[RAFTX_F_Q + j];
                o[DS_A + j] = rec[RAFTX_F_AX + j];
                o[DS_Q + j] = rec[RAFTX_F_Q + j];
                o[DS_P1 + j] = rec[RAFTX_F_P1 + j];
                o[DS_P2 + j] = rec[RAFTX_F_P2 + j];
            }
            o[DS_IQ] = rec[RAFTX_F_IQ];
            o[DS_IQ + 1] = rec[RAFTX_F_IP1];
            o[DS_IQ + 2] = rec[RAFTX_F_IP2];
            o[DS_IQ + 3] = rec[RAFTX_F_AI];
            o[DS_IQ + 4] = rec[RAFTX_F_RHOV];
            o[DS_DQ] = rec[RAFTX_F_DQ];
            o[DS_DQ + 1] = rec[RAFTX_F_DP1];
            o[DS_DQ + 2] = rec[RAFTX_F_DP2];
            o[DS_DQ + 3] = rec[RAFTX_F_DEND];
        }
        s = e;
    }
}

// ---------------------------------------------------------------------------------------------------------
struct GeomArgs {
    int nDesign;
    int64_t nMember;
    const int64_t *memberOff;    // [nDesign+1]
    const double *gm;            // [nMember,RAFTX_GM_N]
    const int64_t *stationOff;   // [nMember+1]
    const double *gs;            // [nStation,RAFTX_GS_N]
    const double *pose;          // [nDesign,6] or null
    const int *mdesign;          // [nMember] design of each member
    double rho, g;
    int nw;
    const double *k;             // [nw] or null
    int *cnt, *cntm;             // [nMember] wet strips / MacCamy-Fuchs rows per member
    int64_t *soff, *cmsoff;      // [nMember+1] exclusive scans of the above
    double *mpose, *mhyd;        // [nMember,MP_N], [nMember,MH_N]
    double *abi;                 // [nStrips,NF] generated ABI records
    double *mcfaux;              // [nRows,3] R, Ca_p1, Ca_p2 of every MacCamy-Fuchs strip
    cplx *cm;                    // [nRows,2,nw]
    int64_t *off, *cmoff;        // [nDesign+1] per-design strip / cm-row offsets
    double *ds;                  // [nStrips,DS_N]
    int *dsi;                    // [nStrips]
    double *A, *Ch, *Wh, *props; // [nDesign,36] [nDesign,36] [nDesign,6] [nDesign,RAFTX_SP_N]
    double *M0, *C0;             // [nDesign,36] design matrices, updated in place per add_mask
    int add_mask;
};

#define GEOM_NOFMA _Pragma("clang fp contract(off)")

// numpy.interp for one abscissa over the station table (clamped; an exact hit on a repeated station returns the
// LAST of the repeats, as numpy's binary search does) -- raft_member.py:1315-1318, 2061-2064
__device__ inline double geom_interp(double x, const double *gs, int n, int f) {
    GEOM_NOFMA
    if (x < gs[RAFTX_GS_S]) return gs[f];
    if (x > gs[(size_t)(n - 1) * RAFTX_GS_N + RAFTX_GS_S]) return gs[(size_t)(n - 1) * RAFTX_GS_N + f];
    int j = 0;
    for (int i = 0; i < n; i++)
        if (gs[(size_t)i * RAFTX_GS_N + RAFTX_GS_S] <= x) j = i;
    const double fj = gs[(size_t)j * RAFTX_GS_N + f];
    if (j == n - 1) return fj;
    const double xj = gs[(size_t)j * RAFTX_GS_N + RAFTX_GS_S];
    if (xj == x) return fj;
    const double slope = (gs[(size_t)(j + 1) * RAFTX_GS_N + f] - fj) / (gs[(size_t)(j + 1) * RAFTX_GS_N + RAFTX_GS_S] - xj);
    return slope * (x - xj) + fj;
}

// strips of a station interval: 0 for a decreasing one, 1 for a flat transition, ceil(l / dlsMax) otherwise
__device__ inline int geom_interval_strips(double lstrip, double dlsMax) {
    GEOM_NOFMA
    if (lstrip > 0.0) return (int)ceil(lstrip / dlsMax);
    return lstrip == 0.0 ? 1 : 0;
}

struct GStrip {
    double ls, dls, ds0, ds1, drs0, drs1;
};
// group g of a member with n stations: 0 = end A, 1..n-1 = the station intervals, n = end B (raft_member.py:205-262)
__device__ inline GStrip geom_strip(const double *gs, int n, int g, int j, int nsub, bool circ) {
    GEOM_NOFMA
    GStrip s;
    const int c1 = circ ? 0 : 1;
    if (g == 0) {
        s.ls = 0.0; s.dls = 0.0;
        s.ds0 = 0.5 * gs[RAFTX_GS_D]; s.ds1 = 0.5 * gs[RAFTX_GS_D + c1];
        s.drs0 = s.ds0; s.drs1 = s.ds1;
    } else if (g == n) {
        const double *b = gs + (size_t)(n - 1) * RAFTX_GS_N;
        s.ls = b[RAFTX_GS_S]; s.dls = 0.0;
        s.ds0 = 0.5 * b[RAFTX_GS_D]; s.ds1 = 0.5 * b[RAFTX_GS_D + c1];
        s.drs0 = -0.5 * b[RAFTX_GS_D]; s.drs1 = -0.5 * b[RAFTX_GS_D + c1];
    } else {
        const double *a = gs + (size_t)(g - 1) * RAFTX_GS_N, *b = gs + (size_t)g * RAFTX_GS_N;
        const double lstrip = b[RAFTX_GS_S] - a[RAFTX_GS_S];
        if (lstrip > 0.0) {
            const double dl = lstrip / nsub;
            const double m0 = 0.5 * (b[RAFTX_GS_D] - a[RAFTX_GS_D]) / lstrip;
            const double m1 = 0.5 * (b[RAFTX_GS_D + c1] - a[RAFTX_GS_D + c1]) / lstrip;
            s.ls = a[RAFTX_GS_S] + dl * (0.5 + j);
            s.dls = dl;
            s.ds0 = a[RAFTX_GS_D] + dl * 2 * m0 * (0.5 + j);
            s.ds1 = a[RAFTX_GS_D + c1] + dl * 2 * m1 * (0.5 + j);
            s.drs0 = dl * m0;
            s.drs1 = dl * m1;
        } else {
            s.ls = a[RAFTX_GS_S]; s.dls = 0.0;
            s.ds0 = 0.5 * (a[RAFTX_GS_D] + b[RAFTX_GS_D]); s.ds1 = 0.5 * (a[RAFTX_GS_D + c1] + b[RAFTX_GS_D + c1]);
            s.drs0 = 0.5 * (b[RAFTX_GS_D] - a[RAFTX_GS_D]); s.drs1 = 0.5 * (b[RAFTX_GS_D + c1] - a[RAFTX_GS_D + c1]);
        }
    }
    return s;
}
// node position of a strip along the member (raft_member.py:362)
__device__ inline double geom_along(double rA, double rB, double ls, double L) {
    GEOM_NOFMA
    return rA + (ls / L) * (rB - rA);
}

// FrustumVCV (helpers.py:36-63): volume and centre of volume of a (circular | rectangular) frustum
__device__ inline void geom_frustum(double a0, double a1, double b0, double b1, bool circ, double H, double &V, double &hc) {
    GEOM_NOFMA
    const double sA = circ ? a0 : a0 + a1, sB = circ ? b0 : b0 + b1;
    if (sA == 0.0 && sB == 0.0) { V = 0.0; hc = 0.0; return; }
    double A1, A2, Am;
    if (circ) {
        A1 = (M_PI / 4) * a0 * a0; A2 = (M_PI / 4) * b0 * b0; Am = (M_PI / 4) * a0 * b0;
    } else {
        A1 = a0 * a1; A2 = b0 * b1; Am = sqrt(A1 * A2);
    }
    V = (A1 + A2 + Am) * H / 3;
    hc = ((A1 + 2 * Am + 3 * A2) / (A1 + Am + A2)) * H / 4;
}

// one thread per member: pose, wet-strip count, hydrostatics about the member's own node
__global__ __launch_bounds__(128) void k_geom_member(GeomArgs A) {
    GEOM_NOFMA
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= A.nMember) return;
    const double *gm = A.gm + (size_t)m * RAFTX_GM_N;
    const double *gs = A.gs + (size_t)A.stationOff[m] * RAFTX_GS_N;
    const int n = (int)(A.stationOff[m + 1] - A.stationOff[m]);
    const int d = A.mdesign[m];
    double ps[6] = {0, 0, 0, 0, 0, 0};
    if (A.pose)
        for (int i = 0; i < 6; i++) ps[i] = A.pose[(size_t)d * 6 + i];
    const bool circ = gm[RAFTX_GM_SHAPE] != 0.0;
    const int flags = (int)gm[RAFTX_GM_FLAGS];
    // ---- Member.setPosition (raft_member.py:324-372) for a rigid member of a rigid unit
    double rA0[3], q0[3], nrm = 0.0;
    for (int i = 0; i < 3; i++) {
        rA0[i] = gm[RAFTX_GM_RA + i];
        q0[i] = gm[RAFTX_GM_RB + i] - gm[RAFTX_GM_RA + i];
        nrm += q0[i] * q0[i];
    }
    nrm = sqrt(nrm);
    for (int i = 0; i < 3; i++) q0[i] /= nrm;
    const double beta = atan2(q0[1], q0[0]);
    const double phi = atan2(sqrt(q0[0] * q0[0] + q0[1] * q0[1]), q0[2]);
    const double gam = gm[RAFTX_GM_GAMMA] * (M_PI / 180.0);
    const double s1 = sin(beta), c1 = cos(beta), s2 = sin(phi), c2 = cos(phi), s3 = sin(gam), c3 = cos(gam);
    const double R0[3][3] = {{c1 * c2 * c3 - s1 * s3, -c3 * s1 - c1 * c2 * s3, c1 * s2},
                             {c1 * s3 + c2 * c3 * s1, c1 * c3 - c2 * s1 * s3, s1 * s2},
                             {-c3 * s2, s2 * s3, c2}};
    const double p10[3] = {R0[0][0], R0[1][0], R0[2][0]};
    const double p20[3] = {q0[1] * p10[2] - q0[2] * p10[1], q0[2] * p10[0] - q0[0] * p10[2], q0[0] * p10[1] - q0[1] * p10[0]};
    // platform rotation, helpers.py:439-466 with (x3, x2, x1) = (roll, pitch, yaw)
    const double sy = sin(ps[5]), cy = cos(ps[5]), sp = sin(ps[4]), cp = cos(ps[4]), sr = sin(ps[3]), cr = cos(ps[3]);
    const double Rp[3][3] = {{cy * cp, cy * sp * sr - cr * sy, sy * sr + cy * cr * sp},
                             {cp * sy, cy * cr + sy * sp * sr, cr * sy * sp - cy * sr},
                             {-sp, cp * sr, cp * cr}};
    double q[3], p1[3], p2[3], rA[3], rB[3], R[2][2];
    const double L = gm[RAFTX_GM_L];
    for (int i = 0; i < 3; i++) {
        q[i] = Rp[i][0] * q0[0] + Rp[i][1] * q0[1] + Rp[i][2] * q0[2];
        p1[i] = Rp[i][0] * p10[0] + Rp[i][1] * p10[1] + Rp[i][2] * p10[2];
        p2[i] = Rp[i][0] * p20[0] + Rp[i][1] * p20[1] + Rp[i][2] * p20[2];
        // node displacement = unit displacement + (R_platform - I) rA0 (raft_fowt.py:706-718)
        const double dsp = (Rp[i][0] - (i == 0 ? 1.0 : 0.0)) * rA0[0] + (Rp[i][1] - (i == 1 ? 1.0 : 0.0)) * rA0[1] +
                           (Rp[i][2] - (i == 2 ? 1.0 : 0.0)) * rA0[2];
        rA[i] = rA0[i] + (ps[i] + dsp);
    }
    for (int i = 0; i < 3; i++) rB[i] = rA[i] + L * q[i];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) R[i][j] = Rp[i][0] * R0[0][j] + Rp[i][1] * R0[1][j] + Rp[i][2] * R0[2][j];
    double *mp = A.mpose + (size_t)m * MP_N;
    for (int i = 0; i < 3; i++) {
        mp[i] = rA0[i]; mp[3 + i] = rA[i]; mp[6 + i] = q[i]; mp[9 + i] = p1[i]; mp[12 + i] = p2[i];
    }
    mp[15] = R[0][0]; mp[16] = R[0][1]; mp[17] = R[1][0]; mp[18] = R[1][1]; mp[19] = L;
    // ---- wet strips (raft_member.py:1310: r[il,2] < 0)
    int wet = 0;
    for (int g = 0; g <= n; g++) {
        int cntg = 1, nsub = 1;
        if (g > 0 && g < n) {
            cntg = geom_interval_strips(gs[(size_t)g * RAFTX_GS_N + RAFTX_GS_S] - gs[(size_t)(g - 1) * RAFTX_GS_N + RAFTX_GS_S],
                                        gm[RAFTX_GM_DLSMAX]);
            nsub = cntg;
        }
        for (int j = 0; j < cntg; j++) {
            const GStrip s = geom_strip(gs, n, g, j, nsub, circ);
            if (geom_along(rA[2], rB[2], s.ls, L) < 0) wet++;
        }
    }
    A.cnt[m] = wet;
    A.cntm[m] = ((flags & RAFTX_GM_FLAG_MCF) && circ && !(flags & RAFTX_GM_FLAG_POTMOD)) ? wet : 0;
    // ---- Member.getHydrostatics, rigid branch, about the member's node (raft_member.py:838-1010)
    double C[36], F[6], Vt = 0.0, rcV[3] = {0, 0, 0}, AWPm = 0.0;
    for (int i = 0; i < 36; i++) C[i] = 0.0;
    for (int i = 0; i < 6; i++) F[i] = 0.0;
    const double beta2 = atan2(q[1], q[0]), phi2 = atan2(sqrt(q[0] * q[0] + q[1] * q[1]), q[2]);
    const double cosPhi = cos(phi2), sinPhi = sin(phi2), tanPhi = tan(phi2), cosBeta = cos(beta2), sinBeta = sin(beta2);
    const double rg = A.rho * A.g;
    const int c1i = circ ? 0 : 1;
    for (int i = 1; i < n; i++) {
        const double *a = gs + (size_t)(i - 1) * RAFTX_GS_N, *b = gs + (size_t)i * RAFTX_GS_N;
        double ra[3], rb[3];
        for (int c = 0; c < 3; c++) {
            ra[c] = rA[c] + q[c] * a[RAFTX_GS_S];
            rb[c] = rA[c] + q[c] * b[RAFTX_GS_S];
        }
        if (ra[2] * rb[2] <= 0) {                                   // crosses (or touches) the waterplane
            const double t = (0 - ra[2]);
            double xWP = ra[0] + t * (rb[0] - ra[0]) / (rb[2] - ra[2]);
            double yWP = ra[1] + t * (rb[1] - ra[1]) / (rb[2] - ra[2]);
            // (sic) interpolated from the UPPER station value at end A's elevation, raft_member.py:899,905
            const double w0 = b[RAFTX_GS_D] + t * (a[RAFTX_GS_D] - b[RAFTX_GS_D]) / (rb[2] - ra[2]);
            const double w1 = b[RAFTX_GS_D + c1i] + t * (a[RAFTX_GS_D + c1i] - b[RAFTX_GS_D + c1i]) / (rb[2] - ra[2]);
            double AWP, IxWP, IyWP;
            if (circ) {
                AWP = (M_PI / 4) * w0 * w0;
                IxWP = IyWP = (M_PI / 64) * w0 * w0 * w0 * w0;
            } else {
                AWP = w0 * w1;
                const double Ix = (1.0 / 12) * w0 * w1 * w1 * w1, Iy = (1.0 / 12) * w0 * w0 * w0 * w1;
                IxWP = R[0][0] * Ix * R[0][0] + R[0][1] * Iy * R[0][1];      // (R diag(Ix,Iy,0) R^T)[0,0], :909-913
                IyWP = R[1][0] * Ix * R[1][0] + R[1][1] * Iy * R[1][1];
            }
            const double LWP = fabs(ra[2] / cosPhi);
            double V, hc;
            geom_frustum(a[RAFTX_GS_D], a[RAFTX_GS_D + c1i], w0, w1, circ, LWP, V, hc);
            double M = 0.0;
            if (circ) M = -rg * M_PI * (w0 * w0 / 32 * (2.0 + tanPhi * tanPhi) + 0.5 * (ra[2] / cosPhi) * (ra[2] / cosPhi)) * sinPhi;
            const double Fz = rg * V;
            const double ex = ra[0] - rA[0], ey = ra[1] - rA[1];
            F[2] += Fz;
            F[3] += ey * Fz;                     // translateForce3to6DOF of (0,0,Fz) at rA_seg - node
            F[4] += -ex * Fz;
            F[3] += M * (-sinBeta);
            F[4] += M * cosBeta;
            xWP -= rA[0];
            yWP -= rA[1];
            C[2 * 6 + 2] += rg * AWP / cosPhi;
            C[2 * 6 + 3] += rg * (-AWP * yWP);
            C[2 * 6 + 4] += rg * (AWP * xWP);
            C[3 * 6 + 2] += rg * (-AWP * yWP);
            C[3 * 6 + 3] += rg * (IxWP + AWP * yWP * yWP);
            C[3 * 6 + 4] += rg * (AWP * xWP * yWP);
            C[4 * 6 + 2] += rg * (AWP * xWP);
            C[4 * 6 + 3] += rg * (AWP * xWP * yWP);
            C[4 * 6 + 4] += rg * (IyWP + AWP * xWP * xWP);
            double rc[3];
            for (int c = 0; c < 3; c++) rc[c] = ra[c] + q[c] * hc;
            C[3 * 6 + 3] += rg * V * (rc[2] - rA[2]);
            C[4 * 6 + 4] += rg * V * (rc[2] - rA[2]);
            C[3 * 6 + 5] += -rg * V * (rc[0] - rA[0]);
            C[4 * 6 + 5] += -rg * V * (rc[1] - rA[1]);
            Vt += V;
            for (int c = 0; c < 3; c++) rcV[c] += rc[c] * V;
            AWPm = AWP;
        } else if (ra[2] <= 0 && rb[2] <= 0) {                     // fully submerged
            double V, hc;
            geom_frustum(a[RAFTX_GS_D], a[RAFTX_GS_D + c1i], b[RAFTX_GS_D], b[RAFTX_GS_D + c1i], circ,
                         b[RAFTX_GS_S] - a[RAFTX_GS_S], V, hc);
            double rc[3], rr[3];
            for (int c = 0; c < 3; c++) { rc[c] = ra[c] + q[c] * hc; rr[c] = rc[c] - rA[c]; }
            const double Fz = rg * V;
            F[2] += Fz;
            F[3] += rr[1] * Fz;
            F[4] += -rr[0] * Fz;
            C[3 * 6 + 3] += rg * V * rr[2];
            C[4 * 6 + 4] += rg * V * rr[2];
            C[3 * 6 + 5] += -rg * V * rr[0];
            C[4 * 6 + 5] += -rg * V * rr[1];
            Vt += V;
            for (int c = 0; c < 3; c++) rcV[c] += rc[c] * V;
        }
    }
    double *mh = A.mhyd + (size_t)m * MH_N;
    for (int i = 0; i < 36; i++) mh[i] = C[i];
    for (int i = 0; i < 6; i++) mh[36 + i] = F[i];
    mh[42] = Vt; mh[43] = rcV[0]; mh[44] = rcV[1]; mh[45] = rcV[2]; mh[46] = AWPm;
}

// exclusive scans of the per-member counts (one workgroup; nMember is ~1e5 for a 10k-design sweep)
__global__ __launch_bounds__(1024) void k_geom_scan(GeomArgs A) {
    __shared__ long long part[2][1024];
    const int t = threadIdx.x, T = blockDim.x;
    const int64_t n = A.nMember;
    const int64_t per = (n + T - 1) / T, lo = (int64_t)t * per, hi = (lo + per < n) ? lo + per : n;
    long long a = 0, b = 0;
    for (int64_t i = lo; i < hi; i++) { a += A.cnt[i]; b += A.cntm[i]; }
    part[0][t] = a; part[1][t] = b;
    __syncthreads();
    if (t == 0) {
        long long sa = 0, sb = 0;
        for (int i = 0; i < T; i++) {
            long long x = part[0][i], y = part[1][i];
            part[0][i] = sa; part[1][i] = sb;
            sa += x; sb += y;
        }
        A.soff[n] = sa;
        A.cmsoff[n] = sb;
    }
    __syncthreads();
    a = part[0][t]; b = part[1][t];
    for (int64_t i = lo; i < hi; i++) {
        A.soff[i] = a; A.cmsoff[i] = b;
        a += A.cnt[i]; b += A.cntm[i];
    }
}
// per-design offsets from the per-member scans
__global__ void k_geom_offsets(GeomArgs A) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d > A.nDesign) return;
    const int64_t m = A.memberOff[d];
    A.off[d] = A.soff[m];
    A.cmoff[d] = A.cmsoff[m];
}

// one wavefront per member: lanes = strips of a group; wet strips are compacted with ballots
__global__ __launch_bounds__(64) void k_geom_fill(GeomArgs A) {
    GEOM_NOFMA
    const int64_t m = blockIdx.x;
    const int lane = threadIdx.x;
    const double *gm = A.gm + (size_t)m * RAFTX_GM_N;
    const double *gs = A.gs + (size_t)A.stationOff[m] * RAFTX_GS_N;
    const int n = (int)(A.stationOff[m + 1] - A.stationOff[m]);
    const int d = A.mdesign[m];
    const double *mp = A.mpose + (size_t)m * MP_N;
    const bool circ = gm[RAFTX_GM_SHAPE] != 0.0;
    const int flags = (int)gm[RAFTX_GM_FLAGS];
    const bool potMod = flags & RAFTX_GM_FLAG_POTMOD;
    const bool mcf = (flags & RAFTX_GM_FLAG_MCF) && circ && !potMod;
    double rA[3], rB[3], q[3], p1[3], p2[3], rP[3] = {0, 0, 0};
    const double L = mp[19];
    if (A.pose)
        for (int i = 0; i < 3; i++) rP[i] = A.pose[(size_t)d * 6 + i];
    for (int i = 0; i < 3; i++) {
        rA[i] = mp[3 + i]; q[i] = mp[6 + i]; p1[i] = mp[9 + i]; p2[i] = mp[12 + i];
        rB[i] = rA[i] + L * q[i];
    }
    const double armN[3] = {rA[0] - rP[0], rA[1] - rP[1], rA[2] - rP[2]};
    const int64_t out0 = A.soff[m];
    const int64_t cmrow0 = A.cmsoff[m], cmbase = A.cmsoff[A.memberOff[d]];
    const int64_t mlocal = m - A.memberOff[d];
    const double rho = A.rho, cdrag = sqrt(8 / M_PI);
    int nwet = 0, il0 = 0;
    for (int g = 0; g <= n; g++) {
        int cntg = 1, nsub = 1;
        if (g > 0 && g < n) {
            cntg = geom_interval_strips(gs[(size_t)g * RAFTX_GS_N + RAFTX_GS_S] - gs[(size_t)(g - 1) * RAFTX_GS_N + RAFTX_GS_S],
                                        gm[RAFTX_GM_DLSMAX]);
            nsub = cntg;
        }
        for (int j0 = 0; j0 < cntg; j0 += 64) {
            const int j = j0 + lane;
            const bool act = j < cntg;
            GStrip s = geom_strip(gs, n, g, act ? j : 0, nsub, circ);
            double r[3];
            for (int c = 0; c < 3; c++) r[c] = geom_along(rA[c], rB[c], s.ls, L);
            const bool wet = act && (r[2] < 0);
            const unsigned long long mask = __ballot(wet);
            const int pos = nwet + __popcll(mask & ((1ull << lane) - 1ull));
            nwet += __popcll(mask);
            if (!wet) continue;
            double *rec = A.abi + (size_t)(out0 + pos) * NF;
            for (int c = 0; c < NF; c++) rec[c] = 0.0;
            for (int c = 0; c < 3; c++) {
                rec[RAFTX_F_X + c] = r[c];
                rec[RAFTX_F_AX + c] = (r[c] - rA[c]) + armN[c];
                rec[RAFTX_F_Q + c] = q[c];
                rec[RAFTX_F_P1 + c] = p1[c];
                rec[RAFTX_F_P2 + c] = p2[c];
            }
            rec[RAFTX_F_CIRC] = circ ? 1.0 : 0.0;
            rec[RAFTX_F_MCF] = -1.0;
            rec[26] = (double)mlocal;
            rec[27] = (double)(il0 + j);
            const double ds0 = s.ds0, ds1 = s.ds1, dr0 = s.drs0, dr1 = s.drs1, dls = s.dls;
            if (!potMod) {
                double v_i, v_end, a_i;
                if (circ) {
                    v_i = 0.25 * M_PI * ds0 * ds0 * dls;
                    const double a3 = ds0 + dr0, b3 = ds0 - dr0;
                    v_end = M_PI / 12.0 * fabs(a3 * a3 * a3 - b3 * b3 * b3);
                    a_i = M_PI * ds0 * dr0;
                } else {
                    v_i = ds0 * ds1 * dls;
                    const double ma = 0.5 * ((ds0 + dr0) + (ds1 + dr1)), mb = 0.5 * ((ds0 - dr0) + (ds1 - dr1));
                    v_end = M_PI / 12.0 * (ma * ma * ma - mb * mb * mb);
                    a_i = (ds0 + dr0) * (ds1 + dr1) - (ds0 - dr0) * (ds1 - dr1);
                }
                if (r[2] + 0.5 * dls > 0) v_i = v_i * (0.5 * dls - r[2]) / dls;      // pierces the waterline, :1328-1330
                const double Ca1 = geom_interp(s.ls, gs, n, RAFTX_GS_CA + 1), Ca2 = geom_interp(s.ls, gs, n, RAFTX_GS_CA + 2);
                const double CaE = geom_interp(s.ls, gs, n, RAFTX_GS_CA + 3);
                rec[RAFTX_F_IQ] = rho * v_end * CaE;
                rec[RAFTX_F_AI] = a_i;
                rec[RAFTX_F_RHOV] = rho * v_i;
                rec[RAFTX_F_AP1] = rho * v_i * Ca1;
                rec[RAFTX_F_AP2] = rho * v_i * Ca2;
                if (mcf) {
                    rec[RAFTX_F_MCF] = (double)((cmrow0 - cmbase) + pos);
                    double *ax = A.mcfaux + (size_t)(cmrow0 + pos) * 3;
                    ax[0] = ds0 / 2; ax[1] = Ca1; ax[2] = Ca2;
                } else {
                    rec[RAFTX_F_IP1] = rho * v_i * (1.0 + Ca1);
                    rec[RAFTX_F_IP2] = rho * v_i * (1.0 + Ca2);
                }
            }
            double a_q, a_p1, a_p2, a_end;                                             // :2066-2110
            if (circ) {
                a_q = M_PI * ds0 * dls; a_p1 = ds0 * dls; a_p2 = ds0 * dls; a_end = fabs(M_PI * ds0 * dr0);
            } else {
                a_q = 2 * (ds0 + ds0) * dls; a_p1 = ds0 * dls; a_p2 = ds1 * dls;      // (sic) :2070
                a_end = fabs((ds0 + dr0) * (ds1 + dr1) - (ds0 - dr0) * (ds1 - dr1));
            }
            rec[RAFTX_F_DQ] = cdrag * 0.5 * rho * a_q * geom_interp(s.ls, gs, n, RAFTX_GS_CD + 0);
            rec[RAFTX_F_DP1] = cdrag * 0.5 * rho * a_p1 * geom_interp(s.ls, gs, n, RAFTX_GS_CD + 1);
            rec[RAFTX_F_DP2] = cdrag * 0.5 * rho * a_p2 * geom_interp(s.ls, gs, n, RAFTX_GS_CD + 2);
            rec[RAFTX_F_DEND] = cdrag * 0.5 * rho * a_end * geom_interp(s.ls, gs, n, RAFTX_GS_CD + 3);
        }
        il0 += cntg;
    }
}

// MacCamy-Fuchs (Cm_p1, Cm_p2)(k) with its cosine ramp, raft_member.py:1459-1484:
//   Cm = 4i / (pi (kR)^2 H1'(kR)),  H1' = (H0 - H2)/2  (Hankel functions of the first kind)
__global__ __launch_bounds__(64) void k_geom_mcf(GeomArgs A, int64_t nRows) {
    GEOM_NOFMA
    const int64_t row = blockIdx.x;
    const int iw = blockIdx.y * blockDim.x + threadIdx.x;
    if (row >= nRows || iw >= A.nw) return;
    const double R = A.mcfaux[row * 3], Ca1 = A.mcfaux[row * 3 + 1], Ca2 = A.mcfaux[row * 3 + 2];
    const double k = A.k[iw];
    const double Tr = M_PI / 5 / R;
    double ramp;
    if (k <= 0.0) ramp = 0.0;
    else if (k < Tr) ramp = 0.5 * (1 - cos(M_PI * k / Tr));
    else ramp = 1.0;
    double cr = 0.0, ci = 0.0;
    if (ramp != 0.0) {
        const double x = k * R;
        const double hr = 0.5 * (j0(x) - jn(2, x)), hi = 0.5 * (y0(x) - yn(2, x));
        const double den = M_PI * (x * x);
        // 4i / (den (hr + i hi)) = 4 (hi + i hr) / (den (hr^2 + hi^2))
        const double mag = den * (hr * hr + hi * hi);
        cr = 4.0 * hi / mag;
        ci = 4.0 * hr / mag;
    }
    A.cm[((size_t)row * 2 + 0) * A.nw + iw] = cplx{cr * ramp + (1.0 + Ca1) * (1 - ramp), ci * ramp};
    A.cm[((size_t)row * 2 + 1) * A.nw + iw] = cplx{cr * ramp + (1.0 + Ca2) * (1 - ramp), ci * ramp};
}

// one thread per design: device strip records + run flags, Morison added mass, hydrostatic reduction
__global__ __launch_bounds__(64) void k_geom_design(GeomArgs A) {
    GEOM_NOFMA
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= A.nDesign) return;
    const int64_t i0 = A.off[d], i1 = A.off[d + 1];
    derive_design_tables(A.abi, i0, i1, A.ds, A.dsi);
    // ---- A_hydro_morison (raft_member.py:1333-1361 + helpers.py:537-560, raft_fowt.py:1625): every strip adds
    // c_n g g^T with g = [n ; arm x n] for its three directions n = p1, p2, q
    double Am[36];
    for (int i = 0; i < 36; i++) Am[i] = 0.0;
    for (int64_t s = i0; s < i1; s++) {
        const double *rec = A.abi + (size_t)s * NF;
        const double ax = rec[RAFTX_F_AX], ay = rec[RAFTX_F_AX + 1], az = rec[RAFTX_F_AX + 2];
        const double cs[3] = {rec[RAFTX_F_AP1], rec[RAFTX_F_AP2], rec[RAFTX_F_IQ]};
        const int fs[3] = {RAFTX_F_P1, RAFTX_F_P2, RAFTX_F_Q};
        for (int t = 0; t < 3; t++) {
            const double nx = rec[fs[t]], ny = rec[fs[t] + 1], nz = rec[fs[t] + 2];
            const double g6[6] = {nx, ny, nz, ay * nz - az * ny, az * nx - ax * nz, ax * ny - ay * nx};
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) Am[i * 6 + j] += cs[t] * g6[i] * g6[j];
        }
    }
    // ---- hydrostatics: T^T C T + the geometric stiffness of the varying T, symmetrised (raft_fowt.py:1122,1181-1199)
    double Ch[36], Wh[6] = {0, 0, 0, 0, 0, 0}, Vt = 0.0, AWPt = 0.0, sVr[3] = {0, 0, 0};
    for (int i = 0; i < 36; i++) Ch[i] = 0.0;
    double th[3] = {0, 0, 0}, rP[3] = {0, 0, 0};
    if (A.pose)
        for (int i = 0; i < 3; i++) { rP[i] = A.pose[(size_t)d * 6 + i]; th[i] = A.pose[(size_t)d * 6 + 3 + i]; }
    for (int64_t m = A.memberOff[d]; m < A.memberOff[d + 1]; m++) {
        const double *mp = A.mpose + (size_t)m * MP_N, *mh = A.mhyd + (size_t)m * MH_N;
        const double a[3] = {mp[3] - rP[0], mp[4] - rP[1], mp[5] - rP[2]};
        const double H[3][3] = {{0, a[2], -a[1]}, {-a[2], 0, a[0]}, {a[1], -a[0], 0}};          // helpers.py:428-437
        // blocks of the member matrix C = [[C11, C12],[C21, C22]]
        double t12[3][3], t21[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s12 = mh[i * 6 + 3 + j], s21 = mh[(3 + i) * 6 + j];
                for (int l = 0; l < 3; l++) {
                    s12 += mh[i * 6 + l] * H[l][j];                 // C11 H + C12
                    s21 += H[l][i] * mh[l * 6 + j];                 // H^T C11 + C21
                }
                t12[i][j] = s12;
                t21[i][j] = s21;
            }
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                Ch[i * 6 + j] += mh[i * 6 + j];
                Ch[i * 6 + 3 + j] += t12[i][j];
                Ch[(3 + i) * 6 + j] += t21[i][j];
                double s = mh[(3 + i) * 6 + 3 + j];
                for (int l = 0; l < 3; l++) s += H[l][i] * t12[l][j] + mh[(3 + i) * 6 + l] * H[l][j];   // H^T (C11 H + C12) + C21 H
                Ch[(3 + i) * 6 + 3 + j] += s;
            }
        const double *F = mh + 36;
        Wh[0] += F[0]; Wh[1] += F[1]; Wh[2] += F[2];
        Wh[3] += F[3] + (a[1] * F[2] - a[2] * F[1]);
        Wh[4] += F[4] + (a[2] * F[0] - a[0] * F[2]);
        Wh[5] += F[5] + (a[0] * F[1] - a[1] * F[0]);
        // dT of a unit rotation applied LINEARLY from the undisplaced arm (raft_fowt.py:640-666): column j moves the
        // arm to arm0 + (theta + e_j) x arm
        for (int j = 0; j < 3; j++) {
            double tj[3] = {th[0], th[1], th[2]};
            tj[j] += 1.0;
            const double dA[3] = {mp[0] + (tj[1] * a[2] - tj[2] * a[1]) - a[0], mp[1] + (tj[2] * a[0] - tj[0] * a[2]) - a[1],
                                  mp[2] + (tj[0] * a[1] - tj[1] * a[0]) - a[2]};
            // -(H(dA)^T F)_i = -(dA x F)_i
            Ch[3 * 6 + 3 + j] -= dA[1] * F[2] - dA[2] * F[1];
            Ch[4 * 6 + 3 + j] -= dA[2] * F[0] - dA[0] * F[2];
            Ch[5 * 6 + 3 + j] -= dA[0] * F[1] - dA[1] * F[0];
        }
        const double V = mh[42];
        Vt += V;
        AWPt += mh[46];
        if (V > 0)
            for (int c = 0; c < 3; c++) sVr[c] += ((mh[43 + c] / V - mp[3 + c]) + mp[c]) * V;   // raft_member.py:1006, raft_fowt.py:938
    }
    for (int i = 0; i < 6; i++)
        for (int j = i + 1; j < 6; j++) {
            const double s = (Ch[i * 6 + j] + Ch[j * 6 + i]) / 2;
            Ch[i * 6 + j] = Ch[j * 6 + i] = s;
        }
    for (int i = 0; i < 36; i++) {
        A.A[(size_t)d * 36 + i] = Am[i];
        A.Ch[(size_t)d * 36 + i] = Ch[i];
        if (A.add_mask & RAFTX_ADD_MORISON) A.M0[(size_t)d * 36 + i] += Am[i];
        if (A.add_mask & RAFTX_ADD_HYDROSTATIC) A.C0[(size_t)d * 36 + i] += Ch[i];
    }
    for (int i = 0; i < 6; i++) A.Wh[(size_t)d * 6 + i] = Wh[i];
    double *pr = A.props + (size_t)d * RAFTX_SP_N;
    for (int i = 0; i < RAFTX_SP_N; i++) pr[i] = 0.0;
    pr[RAFTX_SP_V] = Vt;
    pr[RAFTX_SP_AWP] = AWPt;
    for (int c = 0; c < 3; c++) pr[RAFTX_SP_RCB + c] = Vt != 0.0 ? sVr[c] / Vt : 0.0;
}
