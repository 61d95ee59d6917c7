// raftx_geom.h -- geometry -> strip tables + statics on the device (raftx_build_designs, include/raftx.h).
//
// What the reference does per design in Python objects -- Member.__init__ strip discretisation
// (raft/raft_member.py:190-271), Member.setPosition (:312-377), Member.calcHydroConstants / calcImat / getCmSides
// (:1261-1486), the drag areas of Member.calcHydroLinearization (:2061-2110), Member.getHydrostatics (:838-1010),
// FOWT.calcHydroConstants (raft/raft_fowt.py:1589-1625) and the hydrostatic part of FOWT.calcStatics
// (:811-1201) -- runs here as a handful of small kernels over ALL designs of a sweep at once:
//
//   k_geom_member   one thread per member : pose (q, p1, p2, R, end A), wet-strip count, member hydrostatics
//   k_geom_scan          : exclusive scan over the designs of the wet / MacCamy-Fuchs strip totals (added up by the member pass)
//                   (per-design totals, one-workgroup scan over designs, member offsets inside each design)
//   k_geom_reduce   thread per (design, role): member -> platform reduction of hydrostatics / weight stiffness / inertia
//   k_geom_design   one wavefront / design: the design's strip records generated straight into LDS (lanes = candidate
//                                           strips of all its members, wet ones compacted with ballots), run detection
//                                           for the rotor recurrences (the rules of the routine the host upload path
//                                           uses), device strip records, Morison added mass
//   k_geom_mcf      (row, bin)            : MacCamy-Fuchs complex Cm table (Hankel functions)
//
// The work is tiny next to the solve (a 10k-design sweep has ~110k members / ~530k strips); it exists to remove
// the host packing and the 256 B/strip upload from the sweep's critical path, not to reach a roofline.
// All arithmetic is fp64 with contraction OFF, so that wet/dry decisions and strip constants agree with the
// reference's NumPy arithmetic to the last bits wherever libm agrees.
#pragma once

#define MP_N 24      // member pose scratch: rA0(3) rA(3) q(3) p1(3) p2(3) R00 R01 R10 R11 L
#define MH_N 48      // member hydrostatics: Cmat(36) Fvec(6) V rcV(3) AWP  (about the member's own node)
#define MI_N 48      // member inertia: M(36) mass centre(3) W(6) c33  (about the member's own node)

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
            if (m != 0) {     // a run keeps ONE unit triad and one kind of cross-section: collinear members that differ start a new run
                const double *pr = strips + (size_t)(i - 1) * NF;
                for (int j = 0; j < 3; j++)
                    if (pr[RAFTX_F_P1 + j] != rec[RAFTX_F_P1 + j] || pr[RAFTX_F_P2 + j] != rec[RAFTX_F_P2 + j]) m = 0;
                if ((pr[RAFTX_F_CIRC] != 0.0) != (rec[RAFTX_F_CIRC] != 0.0)) m = 0;      // ... and one kind of cross-section
                if ((pr[RAFTX_F_MCF] >= 0.0) != (rec[RAFTX_F_MCF] >= 0.0)) m = 0;        // ... MacCamy-Fuchs or not as a whole (the inertial sweep picks its loop per run)
                // ... and the arm moves with the strip (the run-type loops keep arm components that cannot change along the
                // run's axis out of the strip loop): a table whose arms do not follow the positions gets no runs
                for (int j = 0; j < 3; j++) {
                    const double da = rec[RAFTX_F_AX + j] - pr[RAFTX_F_AX + j], dx = rec[RAFTX_F_X + j] - pr[RAFTX_F_X + j];
                    if (!(fabs(da - dx) <= 1e-9 * (1.0 + fabs(rec[RAFTX_F_X + j]) + fabs(rec[RAFTX_F_AX + j])))) m = 0;
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
            // Upright cross-section: p1 = (0, 0, +-1) and p2 horizontal up to the rounding dust of the member's rotation
            // matrix (cos(pi/2) = 6e-17 and the like, raft_member.py:355-372).  Pass A may then leave the dust products
            // out of its velocity SQUARES (1e-17 of a positive sum); the records themselves keep the reference's values.
            if (fabs(o[DS_P1]) < 1e-15 && fabs(o[DS_P1 + 1]) < 1e-15 && fabs(o[DS_P2 + 2]) < 1e-15) dsf[(size_t)i] |= DSI_AXAL;
            // Vertical axis: q = (0, 0, +-1) and p1, p2 horizontal up to the same dust; pass B leaves the dust components out
            if (fabs(o[DS_Q]) < 1e-15 && fabs(o[DS_Q + 1]) < 1e-15 && fabs(o[DS_P1 + 2]) < 1e-15 && fabs(o[DS_P2 + 2]) < 1e-15)
                dsf[(size_t)i] |= DSI_VAX;
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
    const int64_t *capOff;       // [nMember+1] or null
    const double *caps;          // [nCap,RAFTX_GC_N]
    double *minert;              // [nMember,MI_N]
    int *err;                    // [4] first unsupported cap layout (member+1) | ballast trim without ballast (design+1) |
                                 //     bad member offsets (design+1) | bad member description (+-(member+1)); 0 = none
    double *Ms, *Cs, *Ws;        // [nDesign,36] [nDesign,36] [nDesign,6]
    const double *Fz;            // [nDesign] vertical mooring force for the ballast trim, or null
    double *drho;                // [nDesign] ballast density correction (RAFTX_TRIM_BALLAST)
    double rho, g;
    int nw;
    const double *k;             // [nw] or null
    int *cnt, *cntm;             // [nMember] wet strips / MacCamy-Fuchs rows per member
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
    int addup_in_design;         // k_geom_design also does its design's share of k_geom_addup (the member -> platform reductions
                                 // have finished when it is launched): sweep crossings, one kernel and one launch gap less
    long long *tot;              // [3] wet strips, MacCamy-Fuchs rows, strips of the largest design (zeroed before the scans)
    // A block of a larger batch reads the batch's own offset arrays (uploaded once, shared by its blocks): the pointers
    // are shifted to the block's first design / member and the bases make the values block-relative.
    long long *hostOut;          // page-locked host memory the device writes directly (no DMA copy to queue behind a bulk
                                 // download): [0..2] = tot, [3..4] = err as 4 ints, [8 .. 8 + nDesign] = off
    int64_t mbase, sbase, cbase;
    int *mdesign_w;              // mdesign, writable (filled on the device by k_geom_zero)
    int mgrid;                   // > 0: member kernels run on a (member position < mgrid, design) grid, designs fastest
    __device__ int64_t mo(int d) const { return memberOff[d] - mbase; }
    __device__ int64_t so(int64_t m) const { return stationOff[m] - sbase; }
    __device__ int64_t co(int64_t m) const { return capOff[m] - cbase; }
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

// the same interpolation with the search done once for several fields of one abscissa (same operations per field)
struct GLocate {
    int idx;        // >= 0: the value is the field of station idx (clamped ends, exact hit, last station); -1: interpolate
    int j;
    double xj, dx;
};
__device__ inline GLocate geom_locate(double x, const double *gs, int n) {
    GEOM_NOFMA
    GLocate L{0, 0, 0.0, 0.0};
    if (x < gs[RAFTX_GS_S]) return L;
    if (x > gs[(size_t)(n - 1) * RAFTX_GS_N + RAFTX_GS_S]) { L.idx = n - 1; return L; }
    int j = 0;
    for (int i0 = 0; i0 < n; i0 += 8) {                   // eight independent loads in flight instead of one per step
        double sv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) sv[u] = gs[(size_t)min(i0 + u, n - 1) * RAFTX_GS_N + RAFTX_GS_S];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i0 + u < n && sv[u] <= x) j = i0 + u;
    }
    L.j = j;
    L.idx = j;
    if (j == n - 1) return L;
    L.xj = gs[(size_t)j * RAFTX_GS_N + RAFTX_GS_S];
    if (L.xj == x) return L;
    L.dx = gs[(size_t)(j + 1) * RAFTX_GS_N + RAFTX_GS_S] - L.xj;
    L.idx = -1;
    return L;
}
__device__ inline double geom_interp_at(const GLocate &L, double x, const double *gs, int f) {
    GEOM_NOFMA
    if (L.idx >= 0) return gs[(size_t)L.idx * RAFTX_GS_N + f];
    const double fj = gs[(size_t)L.j * RAFTX_GS_N + f];
    const double slope = (gs[(size_t)(L.j + 1) * RAFTX_GS_N + f] - fj) / L.dx;
    return slope * (x - L.xj) + fj;
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


// r^5 to (almost always) the correctly rounded double, via double-double products.  FrustumMOI's tapered branch
// divides (r2^5 - r1^5) by (r2 - r1) (helpers.py:81-82); when a cap's hole is "tapered" by one rounding error
// (dAi = dA*(dBi/dB), raft_member.py:691) the last bit of r**5 decides the quotient, and CPython's r**5 is libm pow.
__device__ inline double geom_pow5(double r) {
    GEOM_NOFMA                     // the error-free transformations below must not be re-contracted
    const double h2 = r * r, l2 = fma(r, r, -h2);
    double h4 = h2 * h2, l4 = fma(h2, h2, -h4) + 2.0 * h2 * l2;
    const double s = h4 + l4;
    l4 = l4 - (s - h4);
    h4 = s;
    const double h5 = h4 * r, l5 = fma(h4, r, -h5) + l4 * r;
    return h5 + l5;
}
// FrustumMOI (helpers.py:65-84): radial (about the end) and axial moments of inertia of a circular frustum
__device__ inline void geom_frustum_moi(double dA, double dB, double H, double p, double &Irad, double &Iax) {
    GEOM_NOFMA
    if (H == 0) { Irad = 0; Iax = 0; return; }
    const double r1 = dA / 2, r2 = dB / 2;
    if (dA == dB) {
        Irad = (1.0 / 12) * (p * H * M_PI * r1 * r1) * (3 * r1 * r1 + 4 * H * H);
        Iax = (1.0 / 2) * p * M_PI * H * r1 * r1 * r1 * r1;
    } else {
        const double q5 = (geom_pow5(r2) - geom_pow5(r1)) / (r2 - r1);
        Irad = (1.0 / 20) * p * M_PI * H * q5 + (1.0 / 30) * p * M_PI * H * H * H * (r1 * r1 + 3 * r1 * r2 + 6 * r2 * r2);
        Iax = (1.0 / 10) * p * M_PI * H * q5;
    }
}
// RectangularFrustumMOI (helpers.py:86-148)
__device__ inline void geom_rect_moi(double La, double Wa, double Lb, double Wb, double H, double p, double (&Iv)[3]) {
    GEOM_NOFMA
    if (H == 0) { Iv[0] = Iv[1] = Iv[2] = 0; return; }
    double x2, y2, z2;
    if (La == Lb && Wa == Wb) {
        const double M = p * La * Wa * H;
        Iv[0] = (1.0 / 12) * M * (Wa * Wa + 4 * H * H);
        Iv[1] = (1.0 / 12) * M * (La * La + 4 * H * H);
        Iv[2] = (1.0 / 12) * M * (La * La + Wa * Wa);
        return;
    } else if (La != Lb && Wa != Wb) {
        const double dL = Lb - La, dW = Wb - Wa;
        x2 = (1.0 / 12) * p * (dL * dL * dL * H * (Wb / 5 + Wa / 20) + dL * dL * La * H * (3 * Wb / 4 + Wa / 4) +
                               dL * La * La * H * (Wb + Wa / 2) + La * La * La * H * (Wb / 2 + Wa / 2));
        y2 = (1.0 / 12) * p * (dW * dW * dW * H * (Lb / 5 + La / 20) + dW * dW * Wa * H * (3 * Lb / 4 + La / 4) +
                               dW * Wa * Wa * H * (Lb + La / 2) + Wa * Wa * Wa * H * (Lb / 2 + La / 2));
        z2 = p * (Wb * Lb / 5 + Wa * Lb / 20 + La * Wb / 20 + Wa * La * (1.0 / 30)) * H * H * H;
    } else if (La == Lb) {
        x2 = (1.0 / 24) * p * (La * La * La) * H * (Wb + Wa);
        y2 = (1.0 / 48) * p * La * H * (Wb * Wb * Wb + Wa * Wb * Wb + Wa * Wa * Wb + Wa * Wa * Wa);
        z2 = (1.0 / 12) * p * La * (H * H * H) * (3 * Wb + Wa);
    } else {
        x2 = (1.0 / 48) * p * Wa * H * (Lb * Lb * Lb + La * Lb * Lb + La * La * Lb + La * La * La);
        y2 = (1.0 / 24) * p * (Wa * Wa * Wa) * H * (Lb + La);
        z2 = (1.0 / 12) * p * Wa * (H * H * H) * (3 * Lb + La);
    }
    Iv[0] = y2 + z2; Iv[1] = x2 + z2; Iv[2] = x2 + y2;
}
// inner diameter / side length (d - 2t) along the member, np.interp semantics (raft_member.py:667,727)
__device__ inline double geom_interp_inner(double x, const double *gs, int n, int c) {
    GEOM_NOFMA
#define DIN_(j) (gs[(size_t)(j) * RAFTX_GS_N + RAFTX_GS_D + c] - 2 * gs[(size_t)(j) * RAFTX_GS_N + RAFTX_GS_T])
    if (x < gs[RAFTX_GS_S]) return DIN_(0);
    if (x > gs[(size_t)(n - 1) * RAFTX_GS_N + RAFTX_GS_S]) return DIN_(n - 1);
    int j = 0;
    for (int i = 0; i < n; i++)
        if (gs[(size_t)i * RAFTX_GS_N + RAFTX_GS_S] <= x) j = i;
    const double fj = DIN_(j);
    if (j == n - 1) return fj;
    const double xj = gs[(size_t)j * RAFTX_GS_N + RAFTX_GS_S];
    if (xj == x) return fj;
    const double slope = (DIN_(j + 1) - fj) / (gs[(size_t)(j + 1) * RAFTX_GS_N + RAFTX_GS_S] - xj);
    return slope * (x - xj) + fj;
#undef DIN_
}
// M += translateMatrix6to6DOF(diag(m,m,m) (+) I_rot, r) with I_rot = Ix p1p1^T + Iy p2p2^T + Iz qq^T
// (helpers.py:563-585, raft_member.py:507-516): [[m I, m H],[m H^T, m H H^T + I_rot]], H = getH(r)
__device__ inline void geom_add_submember(double *M, double mass, const double (&Iv)[3], const double *p1, const double *p2,
                                          const double *q, const double (&r)[3]) {
    GEOM_NOFMA
    const double H[3][3] = {{0, r[2], -r[1]}, {-r[2], 0, r[0]}, {r[1], -r[0], 0}};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double hh = 0.0;
            for (int l = 0; l < 3; l++) hh += H[i][l] * mass * H[j][l];
            M[i * 6 + j] += (i == j) ? mass : 0.0;
            M[i * 6 + 3 + j] += mass * H[i][j];
            M[(3 + i) * 6 + j] += mass * H[j][i];
            M[(3 + i) * 6 + 3 + j] += hh + (p1[i] * Iv[0] * p1[j] + p2[i] * Iv[1] * p2[j] + q[i] * Iv[2] * q[j]);
        }
}

// Member.getInertia + getWeight, rigid branch, about the member's own node (raft_member.py:380-836, 1179-1181).
// Zero-length sections re-add the local inertia tensor of the previous section with zero mass, as the reference
// does (Ixx, Iyy, Izz are not reset between sections, :420-513).  Returns a non-zero code for cap layouts the
// reference itself cannot handle.
// The running sums of Member.getInertia over the sections and caps of one member.  In pieces, so that the member pass can feed
// the sections from station rows it holds in registers (one pass over the stations for wet strips, hydrostatics and
// inertia) while k_geom_reinertia walks the table: the arithmetic is the same statements in the same order either way.
struct GInertia {
    double M[36], mc[3], I3[3], vfill;
};
__device__ inline void geom_inertia_init(GInertia &J) {
    for (int i = 0; i < 36; i++) J.M[i] = 0.0;
    for (int c = 0; c < 3; c++) { J.mc[c] = 0.0; J.I3[c] = 0.0; }
    J.vfill = 0.0;
}
// the section between stations a and b (rows of RAFTX_GS_N doubles: registers or memory)
__device__ inline void geom_inertia_section(GInertia &J, const double *a, const double *b, bool circ, double rho_shell, bool trim,
                                            double drho, const double *rA, const double *q, const double *p1, const double *p2) {
    GEOM_NOFMA
    double (&M)[36] = J.M;
    double (&mc)[3] = J.mc;
    double (&I3)[3] = J.I3;
    double &vfill = J.vfill;
    {
        const double l = b[RAFTX_GS_S] - a[RAFTX_GS_S];
        double mass = 0.0, center[3] = {0, 0, 0};
        if (l > 0) {
            // ballast trim (Model.adjustBallastDensity, raft_model.py:1780-1787,1810-1817): zero-density sections lose
            // their fill, every ballasted section gets the design's density correction
            const double l_fill = (trim && a[RAFTX_GS_RHOFILL] == 0.0) ? 0.0 : a[RAFTX_GS_LFILL];
            const double rho_fill = a[RAFTX_GS_RHOFILL] + (l_fill > 0.0 ? drho : 0.0);
            const double dA0 = a[RAFTX_GS_D], dA1 = (circ ? a[RAFTX_GS_D] : a[RAFTX_GS_D + 1]), dB0 = b[RAFTX_GS_D], dB1 = (circ ? b[RAFTX_GS_D] : b[RAFTX_GS_D + 1]);
            const double iA0 = dA0 - 2 * a[RAFTX_GS_T], iA1 = dA1 - 2 * a[RAFTX_GS_T];
            const double iB0 = dB0 - 2 * b[RAFTX_GS_T], iB1 = dB1 - 2 * b[RAFTX_GS_T];
            const double f0 = (iB0 - iA0) * (l_fill / l) + iA0, f1 = (iB1 - iA1) * (l_fill / l) + iA1;
            double Vo, hco, Vi, hci, vf, hcf;
            geom_frustum(dA0, dA1, dB0, dB1, circ, l, Vo, hco);
            geom_frustum(iA0, iA1, iB0, iB1, circ, l, Vi, hci);
            const double m_shell = (Vo - Vi) * rho_shell;
            const double hc_shell = (Vo - Vi != 0) ? ((hco * Vo) - (hci * Vi)) / (Vo - Vi) : 0.0;
            geom_frustum(iA0, iA1, f0, f1, circ, l_fill, vf, hcf);
            vfill += vf;
            const double m_fill = vf * rho_fill;
            mass = m_shell + m_fill;
            const double hc = (mass != 0) ? ((hcf * m_fill) + (hc_shell * m_shell)) / mass : 0.0;
            if (circ) {
                double Iro, Iao, Iri, Iai, Irf, Iaf;
                geom_frustum_moi(dA0, dB0, l, rho_shell, Iro, Iao);
                geom_frustum_moi(iA0, iB0, l, rho_shell, Iri, Iai);
                geom_frustum_moi(iA0, f0, l_fill, rho_fill, Irf, Iaf);
                I3[0] = I3[1] = ((Iro - Iri) + Irf) - mass * hc * hc;
                I3[2] = (Iao - Iai) + Iaf;
            } else {
                double Io[3], Ii[3], If[3];
                geom_rect_moi(dA0, dA1, dB0, dB1, l, rho_shell, Io);
                geom_rect_moi(iA0, iA1, iB0, iB1, l, rho_shell, Ii);
                geom_rect_moi(iA0, iA1, f0, f1, l_fill, rho_fill, If);
                I3[0] = ((Io[0] - Ii[0]) + If[0]) - mass * hc * hc;
                I3[1] = ((Io[1] - Ii[1]) + If[1]) - mass * hc * hc;
                I3[2] = (Io[2] - Ii[2]) + If[2];
            }
            for (int c = 0; c < 3; c++) center[c] = rA[c] + q[c] * (a[RAFTX_GS_S] + hc);
        }
        double rel[3];
        for (int c = 0; c < 3; c++) { mc[c] += mass * center[c]; rel[c] = center[c] - rA[c]; }
        geom_add_submember(M, mass, I3, p1, p2, q, rel);
    }
}
// the caps and bulkheads of the member (station table in memory: the interpolations look stations up by position), then the
// totals into out[MI_N].  Non-zero: a cap layout the reference itself cannot handle.
__device__ inline int geom_inertia_caps_finish(GInertia &J, const double *gs, int n, const double *gc, int ncap, bool circ,
                                               double rho_shell, const double *rA, const double *q, const double *p1,
                                               const double *p2, double g, double *out) {
    GEOM_NOFMA
    const int c1 = circ ? 0 : 1;
    double (&M)[36] = J.M;
    double (&mc)[3] = J.mc;
    const double vfill = J.vfill;
    const double s0 = gs[RAFTX_GS_S], s1 = gs[(size_t)(n - 1) * RAFTX_GS_N + RAFTX_GS_S];
    for (int i = 0; i < ncap; i++) {
        const double *cp = gc + (size_t)i * RAFTX_GC_N;
        const double Lc = cp[RAFTX_GC_S], h = cp[RAFTX_GC_T];
        double dA[2] = {0, 0}, dB[2] = {0, 0}, dAi[2] = {0, 0}, dBi[2] = {0, 0};
        int where;                                       // 0 bottom, 1 top, 2 middle
        if (Lc == s0) {
            where = 0;
            for (int c = 0; c <= c1; c++) {
                dA[c] = gs[RAFTX_GS_D + c] - 2 * gs[RAFTX_GS_T];
                dB[c] = geom_interp_inner(Lc + h, gs, n, c);
                dAi[c] = cp[RAFTX_GC_DIN + c];
                dBi[c] = dB[c] * (dAi[c] / dA[c]);
            }
        } else if (Lc == s1) {
            where = 1;
            if (!circ) return 2;                         // the reference fails here (slBi used before assignment, :731-735)
            dA[0] = geom_interp_inner(Lc - h, gs, n, 0);
            dB[0] = gs[(size_t)(n - 1) * RAFTX_GS_N + RAFTX_GS_D] - 2 * gs[(size_t)(n - 1) * RAFTX_GS_N + RAFTX_GS_T];
            dBi[0] = cp[RAFTX_GC_DIN];
            dAi[0] = dA[0] * (dBi[0] / dB[0]);
        } else if ((Lc > s0 && Lc < s0 + h) || (Lc < s1 && Lc > s1 - h)) {
            return 3;                                    // ValueError('This setup cannot be handled by getIneria yet')
        } else {
            where = 2;
            if (!circ) return 2;                         // np.interp on a 2-D table raises in the reference (:741-755)
            if (i < ncap - 1 && Lc == gc[(size_t)(i + 1) * RAFTX_GC_N + RAFTX_GC_S]) {         // discontinuity: cap going down
                if (i >= n) return 4;
                dA[0] = geom_interp_inner(Lc - h, gs, n, 0);
                dB[0] = gs[(size_t)i * RAFTX_GS_N + RAFTX_GS_D] - 2 * gs[(size_t)i * RAFTX_GS_N + RAFTX_GS_T];   // (sic) d[i] with the CAP index, :690
                dBi[0] = cp[RAFTX_GC_DIN];
                dAi[0] = dA[0] * (dBi[0] / dB[0]);
            } else if (i > 0 && Lc == gc[(size_t)(i - 1) * RAFTX_GC_N + RAFTX_GC_S]) {       // ... and one going up
                if (i >= n) return 4;
                dA[0] = gs[(size_t)i * RAFTX_GS_N + RAFTX_GS_D] - 2 * gs[(size_t)i * RAFTX_GS_N + RAFTX_GS_T];
                dB[0] = geom_interp_inner(Lc + h, gs, n, 0);
                dAi[0] = cp[RAFTX_GC_DIN];
                dBi[0] = dB[0] * (dAi[0] / dA[0]);
            } else {
                dA[0] = geom_interp_inner(Lc - h / 2, gs, n, 0);
                dB[0] = geom_interp_inner(Lc + h / 2, gs, n, 0);
                const double dM = geom_interp_inner(Lc, gs, n, 0);
                dAi[0] = dA[0] * (cp[RAFTX_GC_DIN] / dM);
                dBi[0] = dB[0] * (cp[RAFTX_GC_DIN] / dM);
            }
        }
        double Vo, hco, Vi, hci;
        geom_frustum(dA[0], dA[1], dB[0], dB[1], circ, h, Vo, hco);
        geom_frustum(dAi[0], dAi[1], dBi[0], dBi[1], circ, h, Vi, hci);
        const double m_cap = (Vo - Vi) * rho_shell;
        const double hc_cap = (Vo - Vi != 0) ? ((hco * Vo) - (hci * Vi)) / (Vo - Vi) : 0.0;
        double Ic[3];
        if (circ) {
            double Iro, Iao, Iri, Iai;
            geom_frustum_moi(dA[0], dB[0], h, rho_shell, Iro, Iao);
            geom_frustum_moi(dAi[0], dBi[0], h, rho_shell, Iri, Iai);
            Ic[0] = Ic[1] = (Iro - Iri) - m_cap * hc_cap * hc_cap;
            Ic[2] = Iao - Iai;
        } else {
            double Io[3], Ii[3];
            geom_rect_moi(dA[0], dA[1], dB[0], dB[1], h, rho_shell, Io);
            geom_rect_moi(dAi[0], dAi[1], dBi[0], dBi[1], h, rho_shell, Ii);
            Ic[0] = (Io[0] - Ii[0]) - m_cap * hc_cap * hc_cap;
            Ic[1] = (Io[1] - Ii[1]) - m_cap * hc_cap * hc_cap;
            Ic[2] = Io[2] - Ii[2];
        }
        double rel[3];
        for (int c = 0; c < 3; c++) {
            const double pos = rA[c] + q[c] * Lc;                                       // :772-778
            const double cc = (where == 0) ? pos + q[c] * hc_cap : (where == 1 ? pos - q[c] * (h - hc_cap) : pos - q[c] * ((h / 2) - hc_cap));
            rel[c] = cc - rA[c];
            mc[c] += m_cap * cc;
        }
        geom_add_submember(M, m_cap, Ic, p1, p2, q, rel);
    }
    const double mass = M[0];
    for (int i = 0; i < 36; i++) out[i] = M[i];
    out[36] = mass;
    double dR[3];
    for (int c = 0; c < 3; c++) {
        const double cen = (mass != 0) ? mc[c] / mass : 0.0;
        out[37 + c] = cen;
        dR[c] = cen - rA[c];
    }
    // getWeightOfPointMass(mass, rCoG - node) (helpers.py:1060-1082)
    const double Fz = -g * mass;
    out[40] = 0.0; out[41] = 0.0; out[42] = Fz;
    out[43] = dR[1] * Fz; out[44] = -dR[0] * Fz; out[45] = 0.0;
    out[46] = -mass * g * dR[2];
    out[47] = vfill;                                                                 // member.vfill summed (raft_member.py:506)
    return 0;
}
__device__ inline int geom_member_inertia(const double *gm, const double *gs, int n, const double *gc, int ncap,
                                          const double *rA, const double *q, const double *p1, const double *p2, double g,
                                          bool trim, double drho, double *out) {
    const bool circ = gm[RAFTX_GM_SHAPE] != 0.0;
    const double rho_shell = gm[RAFTX_GM_RHOSHELL];
    GInertia J;
    geom_inertia_init(J);
    for (int i = 1; i < n; i++)
        geom_inertia_section(J, gs + (size_t)(i - 1) * RAFTX_GS_N, gs + (size_t)i * RAFTX_GS_N, circ, rho_shell, trim, drho, rA, q, p1, p2);
    return geom_inertia_caps_finish(J, gs, n, gc, ncap, circ, rho_shell, rA, q, p1, p2, g, out);
}

// one thread per member: pose, wet-strip count, hydrostatics and inertia about the member's own node
// clears what the pass accumulates into (one launch instead of five fills: every launch of the preparation stream has
// to find a free wave slot beside the fused kernel of the batch before)
__global__ void k_geom_zero(GeomArgs A) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < A.nDesign) {
        A.drho[d] = 0.0;
        A.off[d + 1] = 0;                                 // per-design totals: the member pass adds into them (k_geom_scan
        A.cmoff[d + 1] = 0;                               // turns them into offsets)
    }
    if (d < 4) A.err[d] = 0;
    if (d < 5) A.tot[d] = 0;
    if (d == 0) { A.off[0] = 0; A.cmoff[0] = 0; }
    // design index of every member (the host has checked that the member offsets are monotone and in range): this used
    // to be a kernel of its own between this one and the member pass, i.e. on the path between two batches
    if (d < A.nDesign)
        for (int64_t m = A.mo(d); m < A.mo(d + 1); m++) A.mdesign_w[m] = d;
}
#define GEOM_MAX_STATIONS 1024
// Member of this thread.  With A.mgrid > 0 the threads are laid out (member position, design) with the design running
// fastest: a wavefront then holds the SAME member of 64 designs -- in a sweep the designs share a topology, so its lanes
// take the same branches (the straight member order puts columns, pontoons and braces of a few designs side by side in
// one wave, which then walks every path in turn).  -1: no member for this thread.
__device__ inline int64_t geom_member_of_thread(const GeomArgs &A) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (A.mgrid <= 0) return t < A.nMember ? t : -1;
    const int64_t k = t / A.nDesign;
    if (k >= A.mgrid) return -1;
    const int d = (int)(t % A.nDesign);
    const int64_t m = A.mo(d) + k;
    return m < A.mo(d + 1) ? m : -1;
}
// a station row into registers: eight 16-byte loads, all in flight together (rows are 128-byte records of 16-byte-aligned
// tables)
__device__ __forceinline__ void geom_load_row(const double *gs, int i, double (&r)[RAFTX_GS_N]) {
    const double2 *p = reinterpret_cast<const double2 *>(gs + (size_t)i * RAFTX_GS_N);
#pragma unroll
    for (int k = 0; k < RAFTX_GS_N / 2; k++) {
        const double2 v = p[k];
        r[2 * k] = v.x;
        r[2 * k + 1] = v.y;
    }
}
// Memory-wise the pass is a handful of round trips per member: the member's row and offsets, then ONE pass over its
// stations -- each row read once into registers (geom_load_row) and used for the wet-strip count, the hydrostatics and the
// inertia of the section it closes -- then the caps, and every result stored at the end.  (Up to round 5 the three
// computations walked the station table one 8-byte load at a time, each waited for: ~130 dependent round trips, 149 us
// for 1 700 wavefronts that issue VALU in 11 % of their cycles.)
__global__ __launch_bounds__(128) void k_geom_member(GeomArgs A) {
    GEOM_NOFMA
    if (A.mgrid > 0 && A.err[2]) return;                  // member offsets rejected: the grid cannot be walked
    const int64_t m = geom_member_of_thread(A);
    if (m < 0) return;
    if (A.err[2]) {                                       // member offsets rejected: mdesign is not valid
        A.cnt[m] = 0;
        A.cntm[m] = 0;
        return;
    }
    // ---- first round trip: everything that hangs on m alone
    double gm[RAFTX_GM_N];
    {
        const double2 *p = reinterpret_cast<const double2 *>(A.gm + (size_t)m * RAFTX_GM_N);
#pragma unroll
        for (int k = 0; k < RAFTX_GM_N / 2; k++) {
            const double2 v = p[k];
            gm[2 * k] = v.x;
            gm[2 * k + 1] = v.y;
        }
    }
    const int64_t so0 = A.so(m), so1 = A.so(m + 1);
    const int64_t co0 = A.capOff ? A.co(m) : 0, co1 = A.capOff ? A.co(m + 1) : 0;
    const int d = A.mdesign[m];
    const double *gs = A.gs + (size_t)so0 * RAFTX_GS_N;
    const int n = (int)(so1 - so0);
    if (n < 2 || n > GEOM_MAX_STATIONS || !(gm[RAFTX_GM_DLSMAX] > 0.0) || !(gm[RAFTX_GM_L] > 0.0)) {
        A.cnt[m] = 0;
        A.cntm[m] = 0;
        atomicCAS(A.err + 3, 0, (int)(m + 1));            // bad station count / dlsMax / length: the member is skipped
        return;
    }
    if (((int)gm[RAFTX_GM_FLAGS] & RAFTX_GM_FLAG_MCF) && gm[RAFTX_GM_SHAPE] != 0.0 && !A.k) {
        A.cnt[m] = 0;
        A.cntm[m] = 0;
        atomicCAS(A.err + 3, 0, -(int)(m + 1));           // MacCamy-Fuchs member without wave numbers
        return;
    }
    // ---- second round trip: the unit's pose and the first station row
    double ps[6] = {0, 0, 0, 0, 0, 0};
    if (A.pose)
        for (int i = 0; i < 6; i++) ps[i] = A.pose[(size_t)d * 6 + i];
    double a[RAFTX_GS_N], b[RAFTX_GS_N];
    geom_load_row(gs, 0, b);
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
    // ---- one pass over the stations.  Wet strips (raft_member.py:1310: r[il,2] < 0): only the position ls of a candidate
    // matters here -- geom_strip's expressions for it, group by group: end A (ls = 0), the station intervals, end B
    int wet = 0;
    if (geom_along(rA[2], rB[2], 0.0, L) < 0) wet++;                                     // group 0: end A
    // Member.getHydrostatics, rigid branch, about the member's node (raft_member.py:838-1010)
    double C[36], F[6], Vt = 0.0, rcV[3] = {0, 0, 0}, AWPm = 0.0;
    for (int i = 0; i < 36; i++) C[i] = 0.0;
    for (int i = 0; i < 6; i++) F[i] = 0.0;
    const double beta2 = atan2(q[1], q[0]), phi2 = atan2(sqrt(q[0] * q[0] + q[1] * q[1]), q[2]);
    const double cosPhi = cos(phi2), sinPhi = sin(phi2), tanPhi = tan(phi2), cosBeta = cos(beta2), sinBeta = sin(beta2);
    const double rg = A.rho * A.g;
    const bool nostatic = (flags & RAFTX_GM_FLAG_NOSTATIC) != 0;          // nacelle members stay out of the statics (raft_fowt.py:876)
    const double rho_shell = gm[RAFTX_GM_RHOSHELL];
    const bool trim = (A.add_mask & RAFTX_TRIM_BALLAST) != 0;
    for (int i = 1; i < n; i++) {
#pragma unroll
        for (int f = 0; f < RAFTX_GS_N; f++) a[f] = b[f];
        geom_load_row(gs, i, b);
        {   // wet strips of the interval (group i)
            const double lstrip = b[RAFTX_GS_S] - a[RAFTX_GS_S];
            const int cntg = geom_interval_strips(lstrip, gm[RAFTX_GM_DLSMAX]);
            for (int j = 0; j < cntg; j++) {
                double ls;
                if (lstrip > 0.0) {
                    const double dl = lstrip / cntg;
                    ls = a[RAFTX_GS_S] + dl * (0.5 + j);
                } else {
                    ls = a[RAFTX_GS_S];
                }
                if (geom_along(rA[2], rB[2], ls, L) < 0) wet++;
            }
        }
        if (nostatic) continue;
        {   // hydrostatics of the section
            double ra[3], rb[3];
            for (int c = 0; c < 3; c++) {
                ra[c] = rA[c] + q[c] * a[RAFTX_GS_S];
                rb[c] = rA[c] + q[c] * b[RAFTX_GS_S];
            }
            const double aD1 = circ ? a[RAFTX_GS_D] : a[RAFTX_GS_D + 1], bD1 = circ ? b[RAFTX_GS_D] : b[RAFTX_GS_D + 1];
            if (ra[2] * rb[2] <= 0) {                                   // crosses (or touches) the waterplane
                const double t = (0 - ra[2]);
                double xWP = ra[0] + t * (rb[0] - ra[0]) / (rb[2] - ra[2]);
                double yWP = ra[1] + t * (rb[1] - ra[1]) / (rb[2] - ra[2]);
                // (sic) interpolated from the UPPER station value at end A's elevation, raft_member.py:899,905
                const double w0 = b[RAFTX_GS_D] + t * (a[RAFTX_GS_D] - b[RAFTX_GS_D]) / (rb[2] - ra[2]);
                const double w1 = bD1 + t * (aD1 - bD1) / (rb[2] - ra[2]);
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
                geom_frustum(a[RAFTX_GS_D], aD1, w0, w1, circ, LWP, V, hc);
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
                geom_frustum(a[RAFTX_GS_D], aD1, b[RAFTX_GS_D], bD1, circ, b[RAFTX_GS_S] - a[RAFTX_GS_S], V, hc);
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
    }
    if (geom_along(rA[2], rB[2], b[RAFTX_GS_S], L) < 0) wet++;                           // group n: end B (b = the last station)
    double *mh = A.mhyd + (size_t)m * MH_N, *mi = A.minert + (size_t)m * MI_N;
    if (!nostatic) {
        for (int i = 0; i < 36; i++) mh[i] = C[i];
        for (int i = 0; i < 6; i++) mh[36 + i] = F[i];
        mh[42] = Vt; mh[43] = rcV[0]; mh[44] = rcV[1]; mh[45] = rcV[2]; mh[46] = AWPm; mh[47] = 0.0;
    }
    // ---- Member.getInertia: a second pass over the station rows (its 6 x 6 running sums beside the hydrostatic ones would
    // cost the pass its two waves per SIMD), then caps and bulkheads and the totals (the station table by position: memory)
    int code = 0;
    if (!nostatic) {
        GInertia J;
        geom_inertia_init(J);
        geom_load_row(gs, 0, b);
        for (int i = 1; i < n; i++) {
#pragma unroll
            for (int f = 0; f < RAFTX_GS_N; f++) a[f] = b[f];
            geom_load_row(gs, i, b);
            geom_inertia_section(J, a, b, circ, rho_shell, trim, 0.0, rA, q, p1, p2);
        }
        const int ncap = A.capOff ? (int)(co1 - co0) : 0;
        const double *gc = A.capOff ? A.caps + (size_t)co0 * RAFTX_GC_N : nullptr;
        code = geom_inertia_caps_finish(J, gs, n, gc, ncap, circ, rho_shell, rA, q, p1, p2, A.g, mi);
    }
    // ---- everything out
    double *mp = A.mpose + (size_t)m * MP_N;
    for (int i = 0; i < 3; i++) {
        mp[i] = rA0[i]; mp[3 + i] = rA[i]; mp[6 + i] = q[i]; mp[9 + i] = p1[i]; mp[12 + i] = p2[i];
    }
    mp[15] = R[0][0]; mp[16] = R[0][1]; mp[17] = R[1][0]; mp[18] = R[1][1]; mp[19] = L;
    const int wetm = ((flags & RAFTX_GM_FLAG_MCF) && circ && !(flags & RAFTX_GM_FLAG_POTMOD)) ? wet : 0;
    A.cnt[m] = wet;
    A.cntm[m] = wetm;
    // per-design totals (integer sums: the order of the additions does not matter)
    if (wet) atomicAdd(reinterpret_cast<unsigned long long *>(A.off + d + 1), (unsigned long long)wet);
    if (wetm) atomicAdd(reinterpret_cast<unsigned long long *>(A.cmoff + d + 1), (unsigned long long)wetm);
    if (nostatic) {
        for (int i = 0; i < MH_N; i++) mh[i] = 0.0;
        for (int i = 0; i < MI_N; i++) mi[i] = 0.0;
        return;
    }
    if (code) {
        for (int i = 0; i < MI_N; i++) mi[i] = 0.0;
        atomicCAS(A.err, 0, (int)(m + 1));
    }
}

// Model.adjustBallastDensity (raft_model.py:1789-1805), one thread per design: heave imbalance of the untrimmed unit ->
// density correction of all ballasted sections
__global__ void k_geom_trim(GeomArgs A) {
    GEOM_NOFMA
    if (A.err[2] | A.err[3]) return;                      // rejected descriptors: phase 2 reports them
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= A.nDesign) return;
    double mass = A.M0[(size_t)d * 36], V = 0.0, vol = 0.0;
    for (int64_t m = A.mo(d); m < A.mo(d + 1); m++) {
        mass += A.minert[(size_t)m * MI_N + 36];
        vol += A.minert[(size_t)m * MI_N + 47];
        V += A.mhyd[(size_t)m * MH_N + 42];
    }
    if (!(vol > 0.0)) {
        A.drho[d] = 0.0;
        atomicCAS(A.err + 1, 0, d + 1);                       // "can only be used for platforms that have some ballast volume"
        return;
    }
    const double sumFz = -mass * A.g + V * A.rho * A.g + (A.Fz ? A.Fz[d] : 0.0);
    A.drho[d] = sumFz / A.g / vol;
}
// ... and the statics recomputed with the corrected densities (:1820), one thread per member
__global__ __launch_bounds__(128) void k_geom_reinertia(GeomArgs A) {
    GEOM_NOFMA
    if (A.err[2] | A.err[3]) return;                      // rejected descriptors: phase 2 reports them
    const int64_t m = geom_member_of_thread(A);
    if (m < 0) return;
    const double *gm = A.gm + (size_t)m * RAFTX_GM_N;
    if ((int)gm[RAFTX_GM_FLAGS] & RAFTX_GM_FLAG_NOSTATIC) return;
    const double *gs = A.gs + (size_t)A.so(m) * RAFTX_GS_N;
    const int n = (int)(A.so(m + 1) - A.so(m));
    const double *mp = A.mpose + (size_t)m * MP_N;
    const int ncap = A.capOff ? (int)(A.co(m + 1) - A.co(m)) : 0;
    const double *gc = A.capOff ? A.caps + (size_t)A.co(m) * RAFTX_GC_N : nullptr;
    double *mi = A.minert + (size_t)m * MI_N;
    if (geom_member_inertia(gm, gs, n, gc, ncap, mp + 3, mp + 6, mp + 9, mp + 12, A.g, true, A.drho[A.mdesign[m]], mi))
        for (int i = 0; i < MI_N; i++) mi[i] = 0.0;
}

// Exclusive scan over the DESIGNS of the wet-strip / MacCamy-Fuchs-row totals the member pass left in off[d + 1] /
// cmoff[d + 1] (one workgroup; 10^4 entries); the member offsets inside a design are formed where they are used
// (k_geom_design).  Also the largest design (LDS of k_geom_design), and the error flags: this is the last kernel of
// phase 1.
#define GSCAN_CH 4096        // designs per pass of k_geom_scan (two 32-bit LDS rows)
// T threads: 1024 for the shortest pass on an empty chip; 256 fit beside the resident workgroups of a fused kernel (a
// 16-wave block needs a whole CU to itself)
template <int T>
__global__ __launch_bounds__(T) void k_geom_scan_t(GeomArgs A) {
    __shared__ long long part[2][17];
    __shared__ unsigned buf[2][GSCAN_CH];
    const int t = threadIdx.x;
    const int n = A.nDesign;
    constexpr int PER = GSCAN_CH / T;
    // Chunks of GSCAN_CH designs go through LDS, so that global memory (and the page-locked copy of the offsets for the
    // host) is read and written by consecutive lanes; a thread then scans PER consecutive entries of the chunk.  (A
    // thread walking its own stretch of the global arrays took 50 us for 10^4 designs, on the path between two batches.)
    long long carrya = 0, carryb = 0, mx = 0;
    for (int c0 = 0; c0 < n; c0 += GSCAN_CH) {
        const int cn = (n - c0 < GSCAN_CH) ? n - c0 : GSCAN_CH;
        for (int i = t; i < GSCAN_CH; i += T) {
            long long va = 0, vb = 0;
            if (i < cn) { va = A.off[c0 + i + 1]; vb = A.cmoff[c0 + i + 1]; }
            buf[0][i] = (unsigned)va;
            buf[1][i] = (unsigned)vb;
            mx = va > mx ? va : mx;
        }
        __syncthreads();
        long long a = 0, b = 0;
        for (int i = 0; i < PER; i++) { a += buf[0][t * PER + i]; b += buf[1][t * PER + i]; }
        // exclusive scan of the T partial sums: inside a wave by shuffles, across the T / 64 waves through LDS
        const int lane = t & 63, wv = t >> 6;
        long long ia = a, ib = b;
        for (int o = 1; o < 64; o <<= 1) {
            const long long ua = __shfl_up(ia, o, 64), ub = __shfl_up(ib, o, 64);
            if (lane >= o) { ia += ua; ib += ub; }
        }
        if (lane == 63) { part[0][wv] = ia; part[1][wv] = ib; }
        __syncthreads();
        long long ra = carrya, rb = carryb, sa = 0, sb = 0;
        for (int i = 0; i < T / 64; i++) {
            if (i < wv) { ra += part[0][i]; rb += part[1][i]; }
            sa += part[0][i]; sb += part[1][i];
        }
        ra += ia - a;                                     // offset of this thread's first design
        rb += ib - b;
        for (int i = 0; i < PER; i++) {                   // inclusive running totals -> offsets of design i + 1 (below 2^32: the
            ra += buf[0][t * PER + i];                    // strip tables of that many strips would not fit any memory)
            rb += buf[1][t * PER + i];
            buf[0][t * PER + i] = (unsigned)ra;
            buf[1][t * PER + i] = (unsigned)rb;
        }
        __syncthreads();
        for (int i = t; i < cn; i += T) {
            const long long va = buf[0][i], vb = buf[1][i];
            A.off[c0 + i + 1] = va;
            A.cmoff[c0 + i + 1] = vb;
            if (A.hostOut) A.hostOut[8 + c0 + i + 1] = va;
        }
        carrya += sa;
        carryb += sb;
        __syncthreads();
    }
    if (mx) atomicMax(reinterpret_cast<unsigned long long *>(A.tot + 2), (unsigned long long)mx);
    __syncthreads();
    if (t == 0) {
        A.off[0] = 0;
        A.cmoff[0] = 0;
        A.tot[0] = carrya;
        A.tot[1] = carryb;
        if (A.hostOut) {
            A.hostOut[0] = carrya;
            A.hostOut[1] = carryb;
            A.hostOut[2] = A.tot[2];                     // final: every thread's maximum is in (a barrier ago)
            int *e = reinterpret_cast<int *>(A.hostOut + 3);        // every error flag is final: phase 1 ends here
            e[0] = A.err[0]; e[1] = A.err[1]; e[2] = A.err[2]; e[3] = A.err[3];
            A.hostOut[8] = 0;
        }
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


// out += T^T C T for the rigid map T = [[I, H(a)],[0, I]] of a member node (raft_fowt.py:1120-1123; H: helpers.py:428-437)
__device__ inline void geom_reduce_matrix(const double *C, const double (&a)[3], double *out) {
    GEOM_NOFMA
    const double H[3][3] = {{0, a[2], -a[1]}, {-a[2], 0, a[0]}, {a[1], -a[0], 0}};
    double t12[3][3], t21[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s12 = C[i * 6 + 3 + j], s21 = C[(3 + i) * 6 + j];
            for (int l = 0; l < 3; l++) {
                s12 += C[i * 6 + l] * H[l][j];                 // C11 H + C12
                s21 += H[l][i] * C[l * 6 + j];                 // H^T C11 + C21
            }
            t12[i][j] = s12;
            t21[i][j] = s21;
        }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            out[i * 6 + j] += C[i * 6 + j];
            out[i * 6 + 3 + j] += t12[i][j];
            out[(3 + i) * 6 + j] += t21[i][j];
            double s = C[(3 + i) * 6 + 3 + j];
            for (int l = 0; l < 3; l++) s += H[l][i] * t12[l][j] + C[(3 + i) * 6 + l] * H[l][j];   // H^T (C11 H + C12) + C21 H
            out[(3 + i) * 6 + 3 + j] += s;
        }
}
// Wout += T^T W, and the geometric stiffness of the varying T added to Cout (raft_fowt.py:1181-1192): dT of a unit
// rotation applied LINEARLY from the undisplaced arm (:640-666) moves the arm to arm0 + (theta + e_j) x arm, and
// C[3+i,3+j] -= (dArm_j x F)_i for the node force F = W[0:3]
__device__ inline void geom_reduce_vector(const double *W, const double (&a)[3], const double *arm0, const double (&th)[3],
                                          double *Wout, double *Cout) {
    GEOM_NOFMA
    Wout[0] += W[0]; Wout[1] += W[1]; Wout[2] += W[2];
    Wout[3] += W[3] + (a[1] * W[2] - a[2] * W[1]);
    Wout[4] += W[4] + (a[2] * W[0] - a[0] * W[2]);
    Wout[5] += W[5] + (a[0] * W[1] - a[1] * W[0]);
    for (int j = 0; j < 3; j++) {
        double tj[3] = {th[0], th[1], th[2]};
        tj[j] += 1.0;
        const double dA[3] = {arm0[0] + (tj[1] * a[2] - tj[2] * a[1]) - a[0], arm0[1] + (tj[2] * a[0] - tj[0] * a[2]) - a[1],
                              arm0[2] + (tj[0] * a[1] - tj[1] * a[0]) - a[2]};
        Cout[3 * 6 + 3 + j] -= dA[1] * W[2] - dA[2] * W[1];
        Cout[4 * 6 + 3 + j] -= dA[2] * W[0] - dA[0] * W[2];
        Cout[5 * 6 + 3 + j] -= dA[0] * W[1] - dA[1] * W[0];
    }
}

// The per-design step after the strip records exist (it used to be one thread per design: 0.6-0.9 ms of latency for
// 10 k designs).  Every sum keeps the order of that serial form (and of the oracle), so the tables and statics are
// bit-identical to it.
//   k_geom_reduce   thread per (design, role): member -> platform reduction of hydrostatics | weight stiffness | inertia
//   k_geom_design   one wavefront per design; the design's ABI strip records are staged in LDS (coalesced reads), then
//                   run flags (lane per strip), device strip records (lanes over (strip, field): coalesced 256 B rows),
//                   Morison added mass (lane per matrix entry, strips in table order, LDS broadcasts) and the updates
//                   of the design matrices M0 / C0 (add_mask)
// Field j of a device strip record comes from ABI field DS_SRC[j] (-1: zero, -2 - c: unit step x q_c)
__device__ constexpr int DS_SRC[DS_N] = {
    -1, -1, RAFTX_F_MCF, -1,
    RAFTX_F_X, RAFTX_F_X + 1, RAFTX_F_X + 2,
    -2, -3, -4,
    RAFTX_F_AX, RAFTX_F_AX + 1, RAFTX_F_AX + 2,
    RAFTX_F_Q, RAFTX_F_Q + 1, RAFTX_F_Q + 2,
    RAFTX_F_P1, RAFTX_F_P1 + 1, RAFTX_F_P1 + 2,
    RAFTX_F_P2, RAFTX_F_P2 + 1, RAFTX_F_P2 + 2,
    RAFTX_F_IQ, RAFTX_F_IP1, RAFTX_F_IP2, RAFTX_F_AI, RAFTX_F_RHOV,
    RAFTX_F_DQ, RAFTX_F_DP1, RAFTX_F_DP2, RAFTX_F_DEND, -1};
static_assert(DS_MCF == 2 && DS_X == 4 && DS_U == 7 && DS_A == 10 && DS_Q == 13 && DS_P1 == 16 && DS_P2 == 19 &&
              DS_IQ == 22 && DS_DQ == 27 && DS_N == 32, "DS_SRC follows the device record layout");
#define GD_ROW (NF + 1)      // LDS row stride of a staged record (odd: lane-per-strip column reads are conflict-free)
#define GD_GM_N 3            // staged columns of a member row: dlsMax, shape, flags
#define GD_MP_N 13           // staged columns of a member pose: rA, q, p1, p2 (MP_N 3..14), L (19)
#ifdef GEOM_PHASE_TIMING     // tuning builds: cycles per phase of k_geom_design, summed over its waves (printed at ctx destroy)
__device__ unsigned long long geom_phase_cycles[10];
#define GEOM_PHASE(i)                                                                                   \
    do {                                                                                                \
        __syncthreads();                                                                                \
        const unsigned long long now_ = wall_clock64();                                                 \
        if (threadIdx.x == 0) atomicAdd(&geom_phase_cycles[i], now_ - tphase_);                         \
        tphase_ = now_;                                                                                 \
    } while (0)
#define GEOM_PHASE_BEGIN unsigned long long tphase_ = wall_clock64();
#else
#define GEOM_PHASE(i)
#define GEOM_PHASE_BEGIN
#endif
#ifndef GD_T
#define GD_T 128            // threads per design: two wavefronts share the candidate strips and the output loops
#endif
static_assert(GD_T % 64 == 0 && GD_T % DS_N == 0 && GD_T <= 256, "k_geom_design: whole wavefronts, a lane keeps its record field");
// The tables of design d, by the GD_T threads of the calling workgroup: the body of k_geom_design, and the prologue of the
// fused fixed point's generating form (raftx_fusedgen.h), where the workgroup that has claimed a pair builds its design's
// tables itself.  gd_lds: geom_design_lds() bytes of LDS; wcnt: GD_T / 64 ints of LDS.  ABI: also the public-layout copy
// of the strip records (raftx_fetch_strips).  ADDUP: the design's share of k_geom_addup in the same pass (the member ->
// platform reductions must have finished).
template <bool ABI, bool ADDUP>
__device__ __forceinline__ void geom_design_block(const GeomArgs &A, const int d, double *gd_lds, int *wcnt) {
    GEOM_NOFMA
    const int lane = threadIdx.x;                         // lane: thread of the design's workgroup (0 .. GD_T-1)
    const int64_t i0 = A.off[d], i1 = A.off[d + 1];
    const int S = (int)(i1 - i0);
    double *rec = gd_lds;                                 // [S][GD_ROW]
    double *pjv = rec + (size_t)S * GD_ROW, *unv = pjv + S;
    int *okv = reinterpret_cast<int *>(unv + S), *rsv = okv + S, *mfl = rsv + S;
    GEOM_PHASE_BEGIN
    // ---- the strip records of the design (Member.__init__ strip discretisation + calcHydroConstants / drag areas,
    // raft_member.py:190-271, 1261-1368, 2061-2110) are generated HERE, straight into the LDS rows the rest of the kernel
    // works on: lanes take the design's candidate strips (member by member, end plates and interval sub-strips in
    // order) 64 at a time, wet ones are compacted with ballots -- the order of the one-wave-per-member pass this
    // replaces, whose 256-B records went through HBM once more.  The ABI copy (raftx_fetch_strips) is written from LDS.
    {
        const int64_t m0 = A.mo(d), m1 = A.mo(d + 1);
        const int nMem = (int)(m1 - m0);
        const int64_t s0 = A.so(m0);
        const int nSta = (int)(A.so(m1) - s0);
        int *mcum = mfl + S;                              // per member: n + 2 running strip counts of its groups
        int *mbase = mcum + nSta + 2 * nMem;              // [nMem + 1] candidates before each member
        // The design's descriptors -- member rows, station rows, the member pass's poses, first stations -- are staged in
        // LDS by consecutive lanes, once: the candidate loop below then looks everything up on chip instead of walking
        // chains of dependent global loads (member -> first station -> station rows -> interpolation neighbours), which
        // was most of a workgroup's ~20 us on the path between two fused kernels.
        int *ssta = mbase + nMem + 1;                     // [nMem + 1] first station of each member, relative to the design's
        // (okv is 8-byte aligned: an even number of ints from it is too)
        // Only the columns the generation reads are staged -- of a member row: dlsMax, shape, flags (GD_GM_N); of a pose:
        // rA, q, p1, p2, L (GD_MP_N) -- which is what lets SIX workgroups share a CU's LDS at the C3 shape instead of five.
        double *sgm = reinterpret_cast<double *>(okv + (((int)(ssta - okv) + nMem + 2) & ~1));      // [nMem][GD_GM_N]
        double *sgs = sgm + (size_t)nMem * GD_GM_N;                               // [nSta][RAFTX_GS_N]
        double *smp = sgs + (size_t)nSta * RAFTX_GS_N;                            // [nMem][GD_MP_N]
        {
            // (the three tables are one contiguous stretch of LDS: a lane's loads of a round -- ten, enough for a design of 16
            // members and 64 stations -- and its first-station offset are all in flight before the first of them is waited for)
            const double *g0 = A.gm + (size_t)m0 * RAFTX_GM_N, *g1 = A.gs + (size_t)s0 * RAFTX_GS_N, *g2 = A.mpose + (size_t)m0 * MP_N;
            const int n0 = nMem * GD_GM_N, n01 = n0 + nSta * RAFTX_GS_N, total = n01 + nMem * GD_MP_N;
            constexpr int SU = 10;
            const int64_t so_lane = lane <= nMem ? A.so(m0 + lane) : s0;
            for (int base = 0; base < total; base += SU * GD_T) {
                double v[SU];
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    const int idx = base + u * GD_T + lane;
                    const double *src;
                    if (idx < n0) {
                        const int row = idx / GD_GM_N, col = idx % GD_GM_N;
                        src = g0 + (size_t)row * RAFTX_GM_N + (col == 0 ? RAFTX_GM_DLSMAX : (col == 1 ? RAFTX_GM_SHAPE : RAFTX_GM_FLAGS));
                    } else if (idx < n01) {
                        src = g1 + (idx - n0);
                    } else {
                        const int k = idx - n01, row = k / GD_MP_N, col = k % GD_MP_N;
                        src = g2 + (size_t)row * MP_N + (col < 12 ? 3 + col : 19);
                    }
                    v[u] = idx < total ? *src : 0.0;
                }
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    const int idx = base + u * GD_T + lane;
                    if (idx < total) sgm[idx] = v[u];
                }
            }
            // groups of all members flattened over the lanes: member mi owns the n + 1 groups [sta0(mi) + mi, sta0(mi + 1) + mi + 1)
            // (mbase doubles as the members' first-station table until the counts are in)
            if (lane <= nMem) mbase[lane] = ssta[lane] = (int)(so_lane - s0);
            for (int mi = lane + GD_T; mi <= nMem; mi += GD_T) mbase[mi] = ssta[mi] = (int)(A.so(m0 + mi) - s0);
        }
        __syncthreads();
        for (int u = lane; u < nSta + nMem; u += GD_T) {
            int mi = 0;
            while (mi + 1 < nMem && mbase[mi + 1] + mi + 1 <= u) mi++;
            const int sta0 = mbase[mi], n = mbase[mi + 1] - sta0, g = u - (sta0 + mi);
            int cntg = 1;
            if (g > 0 && g < n) {
                const double *gs = sgs + (size_t)sta0 * RAFTX_GS_N;
                cntg = geom_interval_strips(gs[(size_t)g * RAFTX_GS_N + RAFTX_GS_S] - gs[(size_t)(g - 1) * RAFTX_GS_N + RAFTX_GS_S],
                                            sgm[(size_t)mi * GD_GM_N]);
            }
            mcum[sta0 + 2 * mi + g + 1] = cntg;
        }
        __syncthreads();
        for (int mi = lane; mi < nMem; mi += GD_T) {
            const int n = ssta[mi + 1] - ssta[mi];
            int *cum = mcum + ssta[mi] + 2 * mi;
            int a = 0;
            cum[0] = 0;
            for (int g = 0; g <= n; g++) { a += cum[g + 1]; cum[g + 1] = a; }
            mbase[mi + 1] = a;
        }
        __syncthreads();
        if (lane == 0) {
            int a = 0;
            mbase[0] = 0;
            for (int mi = 0; mi < nMem; mi++) { a += mbase[mi + 1]; mbase[mi + 1] = a; }
        }
        __syncthreads();
        const int total = mbase[nMem];
        GEOM_PHASE(0);
        const double rho = A.rho, cdrag = sqrt(8 / M_PI);
        double rP[3] = {0, 0, 0};
        if (A.pose)
            for (int i = 0; i < 3; i++) rP[i] = A.pose[(size_t)d * 6 + i];
        int nwet = 0;
        for (int t0 = 0; t0 < total; t0 += GD_T) {
            const int t = t0 + lane;
            const bool act = t < total;
            const int tq = act ? t : 0;
            int mi = 0;
            while (mi + 1 < nMem && mbase[mi + 1] <= tq) mi++;
            const int64_t m = m0 + mi;
            const int tt = tq - mbase[mi];
            const int n = ssta[mi + 1] - ssta[mi];
            const int *cum = mcum + ssta[mi] + 2 * mi;
            const double *gm = sgm + (size_t)mi * GD_GM_N;                     // dlsMax, shape, flags
            const double *gs = sgs + (size_t)ssta[mi] * RAFTX_GS_N;
            const double *mp = smp + (size_t)mi * GD_MP_N;                     // rA, q, p1, p2, L
            const bool circ = gm[1] != 0.0;
            const int flags = (int)gm[2];
            const bool potMod = flags & RAFTX_GM_FLAG_POTMOD;
            const bool mcf = (flags & RAFTX_GM_FLAG_MCF) && circ && !potMod;
            const double L = mp[12];
            double rA[3], rB[3], q[3], p1[3], p2[3], armN[3];
            for (int i = 0; i < 3; i++) {
                rA[i] = mp[i]; q[i] = mp[3 + i]; p1[i] = mp[6 + i]; p2[i] = mp[9 + i];
                rB[i] = rA[i] + L * q[i];
                armN[i] = rA[i] - rP[i];
            }
            int g = 0;
            while (g < n && cum[g + 1] <= tt) g++;
            const int j = tt - cum[g];
            const int nsub = (g > 0 && g < n) ? cum[g + 1] - cum[g] : 1;
            GStrip st = geom_strip(gs, n, g, j, nsub, circ);
            double r[3];
            for (int c = 0; c < 3; c++) r[c] = geom_along(rA[c], rB[c], st.ls, L);
            const bool wet = act && (r[2] < 0);
            const GLocate at = geom_locate(st.ls, gs, n);     // one search for the seven coefficient interpolations
            // compaction over the two wavefronts: ballots inside each, the first wave's count through LDS
            const unsigned long long mask = __ballot(wet);
            const int wv = lane >> 6, ln = lane & 63;
            if (ln == 0) wcnt[wv] = __popcll(mask);
            __syncthreads();
            int before = 0, all = 0;
            for (int w = 0; w < GD_T / 64; w++) {
                before += w < wv ? wcnt[w] : 0;
                all += wcnt[w];
            }
            const int pos = nwet + before + __popcll(mask & ((1ull << ln) - 1ull));
            nwet += all;
            __syncthreads();
            if (!wet || pos >= S) continue;
            double *row = rec + (size_t)pos * GD_ROW;
            for (int c = 0; c < NF; c++) row[c] = 0.0;
            for (int c = 0; c < 3; c++) {
                row[RAFTX_F_X + c] = r[c];
                row[RAFTX_F_AX + c] = (r[c] - rA[c]) + armN[c];
                row[RAFTX_F_Q + c] = q[c];
                row[RAFTX_F_P1 + c] = p1[c];
                row[RAFTX_F_P2 + c] = p2[c];
            }
            row[RAFTX_F_CIRC] = circ ? 1.0 : 0.0;
            row[RAFTX_F_MCF] = -1.0;
            row[26] = (double)mi;
            row[27] = (double)tt;
            const double ds0 = st.ds0, ds1 = st.ds1, dr0 = st.drs0, dr1 = st.drs1, dls = st.dls;
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
                const double Ca1 = geom_interp_at(at, st.ls, gs, RAFTX_GS_CA + 1), Ca2 = geom_interp_at(at, st.ls, gs, RAFTX_GS_CA + 2);
                const double CaE = geom_interp_at(at, st.ls, gs, RAFTX_GS_CA + 3);
                row[RAFTX_F_IQ] = rho * v_end * CaE;
                row[RAFTX_F_AI] = a_i;
                row[RAFTX_F_RHOV] = rho * v_i;
                row[RAFTX_F_AP1] = rho * v_i * Ca1;
                row[RAFTX_F_AP2] = rho * v_i * Ca2;
                if (mcf) {
                    // rows of this member in the MacCamy-Fuchs table / its first wet strip in the design: sums of the
                    // member pass's counts over the members before it (a design has about ten)
                    int64_t cmbase = A.cmoff[d], cmrow0 = cmbase;
                    int wet0 = 0;
                    for (int64_t mm = m0; mm < m; mm++) { cmrow0 += A.cntm[mm]; wet0 += A.cnt[mm]; }
                    const int posm = pos - wet0;
                    row[RAFTX_F_MCF] = (double)((cmrow0 - cmbase) + posm);
                    double *ax = A.mcfaux + (size_t)(cmrow0 + posm) * 3;
                    ax[0] = ds0 / 2; ax[1] = Ca1; ax[2] = Ca2;
                } else {
                    row[RAFTX_F_IP1] = rho * v_i * (1.0 + Ca1);
                    row[RAFTX_F_IP2] = rho * v_i * (1.0 + Ca2);
                }
            }
            double a_q, a_p1, a_p2, a_end;                                             // :2066-2110
            if (circ) {
                a_q = M_PI * ds0 * dls; a_p1 = ds0 * dls; a_p2 = ds0 * dls; a_end = fabs(M_PI * ds0 * dr0);
            } else {
                a_q = 2 * (ds0 + ds0) * dls; a_p1 = ds0 * dls; a_p2 = ds1 * dls;      // (sic) :2070
                a_end = fabs((ds0 + dr0) * (ds1 + dr1) - (ds0 - dr0) * (ds1 - dr1));
            }
            row[RAFTX_F_DQ] = cdrag * 0.5 * rho * a_q * geom_interp_at(at, st.ls, gs, RAFTX_GS_CD + 0);
            row[RAFTX_F_DP1] = cdrag * 0.5 * rho * a_p1 * geom_interp_at(at, st.ls, gs, RAFTX_GS_CD + 1);
            row[RAFTX_F_DP2] = cdrag * 0.5 * rho * a_p2 * geom_interp_at(at, st.ls, gs, RAFTX_GS_CD + 2);
            row[RAFTX_F_DEND] = cdrag * 0.5 * rho * a_end * geom_interp_at(at, st.ls, gs, RAFTX_GS_CD + 3);
        }
        __syncthreads();
        GEOM_PHASE(1);
        if constexpr (ABI) {
            double *abi = A.abi + (size_t)i0 * NF;
            for (int t = lane; t < S * NF; t += GD_T) abi[t] = rec[(t / NF) * GD_ROW + (t % NF)];
        }
    }
    __syncthreads();
    GEOM_PHASE(2);
    // ---- run detection (derive_design_tables, same expressions and order; see there for the rules)
    // (both rows' fields are read up front, unconditionally -- strip 0 reads itself as its predecessor -- so that the LDS reads
    // of a strip are in flight together instead of one per branch)
    for (int t = lane; t < S; t += GD_T) {
        bool pass = false;
        double pj = 0.0;
        const double *pr = rec + (size_t)(t > 0 ? t - 1 : 0) * GD_ROW, *cr = rec + (size_t)t * GD_ROW;
        double pq[3], cq[3], px[3], cx[3];
        for (int j = 0; j < 3; j++) {
            pq[j] = pr[RAFTX_F_Q + j]; cq[j] = cr[RAFTX_F_Q + j]; px[j] = pr[RAFTX_F_X + j]; cx[j] = cr[RAFTX_F_X + j];
        }
        if (t > 0) {
            bool same = true;
            double dv[3];
            for (int j = 0; j < 3; j++) {
                same = same && (pq[j] == cq[j]);
                dv[j] = cx[j] - px[j];
                pj += dv[j] * cq[j];
            }
            if (same && (pj > 0.0) && (fabs(pj) <= 1.797e308)) {
                double perp2 = 0.0, scale = 1.0;
                for (int j = 0; j < 3; j++) {
                    double tt = dv[j] - pj * cq[j];
                    perp2 += tt * tt;
                    scale += fabs(cx[j]);
                }
                pass = !(sqrt(perp2) > 1e-10 * scale);
            }
        }
        okv[t] = pass ? 1 : 0;
        pjv[t] = pj;
    }
    __syncthreads();
    // pair tests -> runs (run start of every strip, smallest step of the run; a run is cut after 64 strips).  With one strip
    // per lane: the start of strip t is the last strip <= t that failed its pair test (ballots, the first wave's last start
    // through LDS), the smallest step an integer minimum over the run's strips (positive doubles order like their bit
    // patterns).  Designs with more strips than lanes, or with a run that hits the cap, take the serial pass.
    bool serial = S > GD_T;
    if (!serial) {
        const int wv = lane >> 6, ln = lane & 63;
        const bool st = lane < S && (lane == 0 || !okv[lane]);
        const unsigned long long mask = __ballot(st);
        if (ln == 0) wcnt[wv] = mask ? wv * 64 + 63 - __builtin_clzll(mask) : -1;
        if (st) reinterpret_cast<unsigned long long *>(unv)[lane] = 0x7ff0000000000000ull;
        __syncthreads();
        const unsigned long long below = mask & (ln == 63 ? ~0ull : ((2ull << ln) - 1ull));
        int s = below ? wv * 64 + 63 - __builtin_clzll(below) : -1;
        for (int w = wv - 1; w >= 0 && s < 0; w--) s = wcnt[w];
        if (lane < S) {
            rsv[lane] = s;
            if (!st) atomicMin(reinterpret_cast<unsigned long long *>(unv) + s, (unsigned long long)__double_as_longlong(pjv[lane]));
        }
        serial = __syncthreads_or(lane < S && lane - s >= 64) != 0;
        if (!serial && st && reinterpret_cast<unsigned long long *>(unv)[lane] == 0x7ff0000000000000ull) unv[lane] = 0.0;
    }
    if (serial && lane == 0) {
        int s = 0;
        double unit = 0.0;
        for (int t = 0; t < S; t++) {
            if (t == 0 || !okv[t] || (t - s) >= 64) {
                if (t > 0) unv[s] = unit;
                s = t;
                unit = 0.0;
            } else {
                const double pj = pjv[t];
                unit = (unit == 0.0 || pj < unit) ? pj : unit;
            }
            rsv[t] = s;
        }
        if (S > 0) unv[s] = unit;
    }
    __syncthreads();
    for (int t = lane; t < S; t += GD_T) {
        const int s = rsv[t];
        const double unit = unv[s];
        const double *cr = rec + (size_t)t * GD_ROW, *pr = rec + (size_t)(t > 0 ? t - 1 : 0) * GD_ROW;
        // (the same: the two rows' fields up front)
        double cX[3], cQ[3], cP1[3], cP2[3], cA[3], pX[3], pP1[3], pP2[3], pA[3];
        for (int j = 0; j < 3; j++) {
            cX[j] = cr[RAFTX_F_X + j]; cQ[j] = cr[RAFTX_F_Q + j]; cP1[j] = cr[RAFTX_F_P1 + j]; cP2[j] = cr[RAFTX_F_P2 + j];
            cA[j] = cr[RAFTX_F_AX + j];
            pX[j] = pr[RAFTX_F_X + j]; pP1[j] = pr[RAFTX_F_P1 + j]; pP2[j] = pr[RAFTX_F_P2 + j]; pA[j] = pr[RAFTX_F_AX + j];
        }
        const double cCirc = cr[RAFTX_F_CIRC], cMcf = cr[RAFTX_F_MCF], pCirc = pr[RAFTX_F_CIRC], pMcf = pr[RAFTX_F_MCF];
        const double pjt = pjv[t];
        int m = 0;
        if (t > s && unit > 0.0) {
            double ratio = pjt / unit;
            int mi = (int)floor(ratio + 0.5);
            if (mi >= 1 && mi <= 2 && fabs(ratio - mi) < 1e-9) {
                bool ok = true;
                for (int j = 0; j < 3; j++) {
                    double pred = pX[j] + (double)mi * unit * cQ[j];
                    if (fabs(pred - cX[j]) > 1e-10 * (1.0 + fabs(cX[j]))) ok = false;
                }
                if (ok) m = mi;
            }
        }
        if (m != 0) {
            for (int j = 0; j < 3; j++)
                if (pP1[j] != cP1[j] || pP2[j] != cP2[j]) m = 0;
            if ((pCirc != 0.0) != (cCirc != 0.0)) m = 0;
            if ((pMcf >= 0.0) != (cMcf >= 0.0)) m = 0;
            for (int j = 0; j < 3; j++) {
                const double da = cA[j] - pA[j], dx = cX[j] - pX[j];
                if (!(fabs(da - dx) <= 1e-9 * (1.0 + fabs(cX[j]) + fabs(cA[j])))) m = 0;
            }
        }
        int fl = m | (cCirc != 0.0 ? DSI_CIRC : 0);
        if (fabs(cP1[0]) < 1e-15 && fabs(cP1[1]) < 1e-15 && fabs(cP2[2]) < 1e-15) fl |= DSI_AXAL;
        if (fabs(cQ[0]) < 1e-15 && fabs(cQ[1]) < 1e-15 && fabs(cP1[2]) < 1e-15 && fabs(cP2[2]) < 1e-15)
            fl |= DSI_VAX;
        A.dsi[(size_t)i0 + t] = fl;
    }
    GEOM_PHASE(3);
    // ---- device strip records, one 256 B row per strip, written by consecutive lanes (a lane keeps its field)
    {
        double *dso = A.ds + (size_t)i0 * DS_N;
        // (four rows of the table per trip: the LDS reads of a trip are in flight together, then its stores -- one read, one
        // wait, one store per trip was a fifth of the kernel)
        const int j = lane % DS_N, f = DS_SRC[j], nT = S * DS_N;
        constexpr int DU = 4;
        for (int t0 = lane; t0 < nT; t0 += DU * GD_T) {
          double vv[DU];
#pragma unroll
          for (int u = 0; u < DU; u++) {
            const int t = t0 + u * GD_T;
            const int i = t < nT ? t / DS_N : 0;              // (past the end: a valid row, read and dropped)
            const double *cr = rec + (size_t)i * GD_ROW;
            double v = 0.0;
            if (f >= 0) v = cr[f];
            else if (f <= -2) v = unv[rsv[i]] * cr[RAFTX_F_Q + (-2 - f)];
            if (t < nT && j == DS_MCF && v >= 0.0) {
                // MacCamy-Fuchs row of the DEVICE record: the first row of this design with the same (R, Ca_p1, Ca_p2) --
                // the strips of a column share them, so the solver's reads of the Cm table fall on a few rows per design
                // (lines it has just touched) instead of one row per strip, 32 B per strip and bin out of HBM.  The ABI
                // record (raftx_fetch_strips) keeps its own row, as the reference's table has one.
                const double *ax0 = A.mcfaux + (size_t)A.cmoff[d] * 3, *axm = ax0 + (size_t)v * 3;
                const int mrow = (int)v;
                for (int r = 0; r < mrow; r++)
                    if (ax0[r * 3] == axm[0] && ax0[r * 3 + 1] == axm[1] && ax0[r * 3 + 2] == axm[2]) {
                        v = (double)r;
                        break;
                    }
            }
            vv[u] = v;
          }
#pragma unroll
          for (int u = 0; u < DU; u++) {
            const int t = t0 + u * GD_T;
            if (t < nT) dso[t] = vv[u];
          }
        }
    }
    GEOM_PHASE(4);
    // ---- A_hydro_morison (raft_member.py:1333-1361 + helpers.py:537-560, raft_fowt.py:1625): every strip adds
    // c_n g g^T with g = [n ; arm x n] for its three directions n = p1, p2, q.  The 3 x (c, g) of every strip are formed
    // once (lane per strip) and parked over the staged records; lane (i, j) then sums c g_i g_j over the strips in order.
    constexpr int GV = 21;                                // doubles per strip: 3 x (c, g[6])
    for (int t0 = 0; t0 < S; t0 += GD_T) {
        const int t = t0 + lane;
        double gv[GV];
        if (t < S) {
            const double *cr = rec + (size_t)t * GD_ROW;
            const double ax = cr[RAFTX_F_AX], ay = cr[RAFTX_F_AX + 1], az = cr[RAFTX_F_AX + 2];
            const double cs[3] = {cr[RAFTX_F_AP1], cr[RAFTX_F_AP2], cr[RAFTX_F_IQ]};
            const int fs[3] = {RAFTX_F_P1, RAFTX_F_P2, RAFTX_F_Q};
            for (int tt = 0; tt < 3; tt++) {
                const double nx = cr[fs[tt]], ny = cr[fs[tt] + 1], nz = cr[fs[tt] + 2];
                gv[tt * 7] = cs[tt];
                gv[tt * 7 + 1] = nx; gv[tt * 7 + 2] = ny; gv[tt * 7 + 3] = nz;
                gv[tt * 7 + 4] = ay * nz - az * ny;
                gv[tt * 7 + 5] = az * nx - ax * nz;
                gv[tt * 7 + 6] = ax * ny - ay * nx;
            }
        }
        __syncthreads();                                  // every record of this round is in registers (and the emission is done)
        if (t < S)
            for (int e = 0; e < GV; e++) rec[(size_t)t * GV + e] = gv[e];
    }
    __syncthreads();
    GEOM_PHASE(5);
    // lane (k, i, j): the sum over the k-th third of the strips; the three partial sums meet in lane (0, i, j)
    {
        const int k3 = lane / 36, e = lane % 36, i = e / 6, j = e % 6;
        double am = 0.0;
        if (k3 < 3) {
            const int lo = (int)(((long long)S * k3) / 3), hi = (int)(((long long)S * (k3 + 1)) / 3);
            for (int s = lo; s < hi; s++) {
                const double *g = rec + (size_t)s * GV;
                for (int tt = 0; tt < 3; tt++) am += g[tt * 7] * g[tt * 7 + 1 + i] * g[tt * 7 + 1 + j];
            }
        }
        __syncthreads();                                  // the partial sums go where the g-vectors were
        if (k3 < 3) rec[lane] = am;
        __syncthreads();
        if (lane < 36) {
            am = (rec[lane] + rec[36 + lane]) + rec[72 + lane];
            const size_t o = (size_t)d * 36 + lane;
            A.A[o] = am;                                  // k_geom_addup adds it (and the platform reductions) to M0 / C0
            if constexpr (ADDUP) {                        // ... or this lane does, in k_geom_addup's order
                double m0 = A.M0[o], c0 = A.C0[o];
                if (A.add_mask & RAFTX_ADD_MORISON) m0 += am;
                if (A.add_mask & RAFTX_ADD_HYDROSTATIC) c0 += A.Ch[o];
                if (A.add_mask & RAFTX_ADD_INERTIA) { m0 += A.Ms[o]; c0 += A.Cs[o]; }
                A.M0[o] = m0;
                A.C0[o] = c0;
            }
        }
    }
    GEOM_PHASE(6);
    (void)mfl;
}
__global__ __launch_bounds__(GD_T) void k_geom_design(GeomArgs A) {
    extern __shared__ double gd_lds[];
    __shared__ int wcnt[GD_T / 64];
    if ((int)blockIdx.x >= A.nDesign) return;
    if (A.abi) geom_design_block<true, false>(A, (int)blockIdx.x, gd_lds, wcnt);
    else if (A.addup_in_design) geom_design_block<false, true>(A, (int)blockIdx.x, gd_lds, wcnt);       // sweep crossings
    else geom_design_block<false, false>(A, (int)blockIdx.x, gd_lds, wcnt);
}
// What the device adds to the caller's matrices (add_mask), in a fixed order: Morison added mass (k_geom_design), then the
// member -> platform reductions (k_geom_reduce, which runs beside the generation on its own stream).
__global__ void k_geom_addup(GeomArgs A) {
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (size_t)A.nDesign * 36) return;
    if (A.add_mask & RAFTX_ADD_MORISON) A.M0[o] += A.A[o];
    if (A.add_mask & RAFTX_ADD_HYDROSTATIC) A.C0[o] += A.Ch[o];
    if (A.add_mask & RAFTX_ADD_INERTIA) { A.M0[o] += A.Ms[o]; A.C0[o] += A.Cs[o]; }
}
// dynamic LDS of k_geom_design for designs of up to maxS strips
static size_t geom_design_lds(int maxS, int maxSta, int maxMem) {
    const size_t S = (size_t)(maxS > 0 ? maxS : 1);
    return sizeof(double) * (S * GD_ROW + 2 * S) + sizeof(int) * (3 * S + (size_t)maxSta + 3 * (size_t)maxMem + 1) + 16 +
           sizeof(int) * ((size_t)maxMem + 4) +                                                           // first stations
           sizeof(double) * ((size_t)maxMem * (GD_GM_N + GD_MP_N) + (size_t)maxSta * RAFTX_GS_N);        // staged descriptors
}

__global__ __launch_bounds__(64) void k_geom_reduce(GeomArgs A) {
    GEOM_NOFMA
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int d = t / 3, role = t % 3;        // 0: C_hydro, W_hydro, V, AWP, rCB;  1: C_struc, W_struc;  2: M_struc, mass, rCG
    if (d >= A.nDesign) return;
    // ---- hydrostatics: T^T C T + the geometric stiffness of the varying T, symmetrised (raft_fowt.py:1122,1181-1199);
    // weight stiffness and inertia of the members the same way (raft_fowt.py:876-900)
    double th[3] = {0, 0, 0}, rP[3] = {0, 0, 0};
    if (A.pose)
        for (int i = 0; i < 3; i++) { rP[i] = A.pose[(size_t)d * 6 + i]; th[i] = A.pose[(size_t)d * 6 + 3 + i]; }
    double out[36], Wv[6] = {0, 0, 0, 0, 0, 0}, sc0 = 0.0, sc1 = 0.0, sr[3] = {0, 0, 0};
    for (int i = 0; i < 36; i++) out[i] = 0.0;
    for (int64_t m = A.mo(d); m < A.mo(d + 1); m++) {
        const double *mp = A.mpose + (size_t)m * MP_N, *mh = A.mhyd + (size_t)m * MH_N, *mi = A.minert + (size_t)m * MI_N;
        const double a[3] = {mp[3] - rP[0], mp[4] - rP[1], mp[5] - rP[2]};
        double C[36];
        if (role == 1) {                      // weight stiffness of the member about its node: C[3,3] = C[4,4] = -m g dR_z (helpers.py:1076-1078)
            for (int i = 0; i < 36; i++) C[i] = 0.0;
            C[3 * 6 + 3] = C[4 * 6 + 4] = mi[46];
        } else {
            const double *src = role == 0 ? mh : mi;
            for (int i = 0; i < 36; i++) C[i] = src[i];
        }
        geom_reduce_matrix(C, a, out);
        if (role < 2) {
            const double *W = role == 0 ? mh + 36 : mi + 40;
            geom_reduce_vector(W, a, mp, th, Wv, out);
        }
        if (role == 2) {
            for (int c = 0; c < 3; c++) sr[c] += ((mi[37 + c] - mp[3 + c]) + mp[c]) * mi[36];        // raft_fowt.py:902-903
            sc0 += mi[47];                                                                              // ballast volume
        } else if (role == 0) {
            const double V = mh[42];
            sc0 += V;
            sc1 += mh[46];
            if (V > 0)
                for (int c = 0; c < 3; c++) sr[c] += ((mh[43 + c] / V - mp[3 + c]) + mp[c]) * V;      // raft_member.py:1006, raft_fowt.py:938
        }
    }
    for (int i = 0; i < 6; i++)
        for (int j = i + 1; j < 6; j++) {
            const double s = (out[i * 6 + j] + out[j * 6 + i]) / 2;
            out[i * 6 + j] = out[j * 6 + i] = s;
        }
    double *dst = (role == 0 ? A.Ch : role == 1 ? A.Cs : A.Ms) + (size_t)d * 36;
    for (int i = 0; i < 36; i++) dst[i] = out[i];
    double *pr = A.props + (size_t)d * RAFTX_SP_N;
    if (role == 0) {
        for (int i = 0; i < 6; i++) A.Wh[(size_t)d * 6 + i] = Wv[i];
        pr[RAFTX_SP_V] = sc0;
        pr[RAFTX_SP_AWP] = sc1;
        for (int c = 0; c < 3; c++) pr[RAFTX_SP_RCB + c] = sc0 != 0.0 ? sr[c] / sc0 : 0.0;
    } else if (role == 1) {
        for (int i = 0; i < 6; i++) A.Ws[(size_t)d * 6 + i] = Wv[i];
    } else {
        pr[RAFTX_SP_VFILL] = sc0;
        pr[RAFTX_SP_DRHO] = (A.add_mask & RAFTX_TRIM_BALLAST) ? A.drho[d] : 0.0;
        pr[RAFTX_SP_MASS] = out[0];                                                               // raft_fowt.py:1206-1207
        for (int c = 0; c < 3; c++) pr[RAFTX_SP_RCG + c] = out[0] != 0.0 ? sr[c] / out[0] : 0.0;
        pr[11] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Parametric variants of one base unit, written in HBM (raftx_variant_program / raftx_sweep_prepare_variants;
// raft/parametersweep.py:39-87 edits a handful of parameters per candidate and the dependent geometry follows).
// One thread per descriptor ROW of a variant -- member, station or cap -- builds it from the base row and the edits:
// the member's end points as affine functions of the parameters, its length (raft_member.py:72), the heading rotation
// (:75-77, helpers.py:587-602), station / ballast / cap positions as fractions of the length (:99,143,173), diameters
// as affine functions.  Every expression is evaluated like the host's NumPy form -- left to right, no fused
// multiply-add -- so that the rows are the bits raft_amd/geometry.py SweepTables produces (tests/test_geometry.py).
// 6.6 KB per VolturnUS-S variant: a 10^4-design batch is 66 MB of stores (16-double rows, one full line per thread)
// instead of 66 MB over PCIe.
struct ExpandArgs {
    int n, nM, nSt, nCap, nP;
    const double *params;                                    // [n,nP]
    const double *gm, *gs, *gc;                              // base rows
    const int *stMember, *capMember;
    const double *stFrac, *fillFrac, *capFrac;
    const double *endCoef, *headCS, *diaCoef;                // [nM,6,nP+1] [nM,2] [nSt,2,nP+1]
    const int *endEdit, *diaEdit;
    double *gm_out, *gs_out, *gc_out;                        // [n*nM,GM_N] [n*nSt,GS_N] [n*nCap,GC_N]
};
__device__ inline double affine_eval(const double *coef, const double *p, int nP) {
    GEOM_NOFMA
    double v = coef[0];
    for (int q = 0; q < nP; q++) v = v + coef[1 + q] * p[q];
    return v;
}
// Kind-major grid: the first blocks take member rows, the next station rows, the last cap rows, GE_T consecutive rows of
// ONE output array each.  A thread computes its row into an LDS tile (row stride 17 doubles: conflict-free for the
// per-thread writes and for the read-back), then the block copies the tile out with 16-byte stores over consecutive
// addresses -- as one thread per row storing its own 128 bytes, every store instruction touched 64 different lines (80 us
// for 66 MB; beside a running fused kernel that is 80 us of its time).
#define GE_T 256
__global__ void __launch_bounds__(GE_T) k_geom_expand(ExpandArgs E, unsigned nbM, unsigned nbS) {
    GEOM_NOFMA
    __shared__ double tile[GE_T * 17];
    const int kind = blockIdx.x < nbM ? 0 : (blockIdx.x < nbM + nbS ? 1 : 2);
    const int nX = kind == 0 ? E.nM : (kind == 1 ? E.nSt : E.nCap);
    const int W = kind == 2 ? RAFTX_GC_N : RAFTX_GM_N;                    // doubles per row (RAFTX_GS_N == RAFTX_GM_N)
    const size_t total = (size_t)E.n * nX;
    const size_t row0 = (size_t)(blockIdx.x - (kind == 0 ? 0u : (kind == 1 ? nbM : nbM + nbS))) * GE_T;
    const int nrows = (int)(total - row0 < (size_t)GE_T ? total - row0 : (size_t)GE_T);
    const int t = threadIdx.x;
    if (t < nrows) {
        const size_t R = row0 + t;
        const int d = (int)(R / nX), idx = (int)(R % nX);
        const double *p = E.params + (size_t)d * E.nP;
        const int m = kind == 0 ? idx : (kind == 1 ? E.stMember[idx] : E.capMember[idx]);
        const double *bm = E.gm + (size_t)m * RAFTX_GM_N;
        const bool ed = E.endEdit[m] != 0;
        double e[6] = {0, 0, 0, 0, 0, 0}, L = bm[RAFTX_GM_L];
        if (ed) {
            for (int i = 0; i < 6; i++) e[i] = affine_eval(E.endCoef + ((size_t)m * 6 + i) * (E.nP + 1), p, E.nP);
            const double dx = e[3] - e[0], dy = e[4] - e[1], dz = e[5] - e[2];
            L = sqrt((dx * dx + dy * dy) + dz * dz);         // |rB - rA| before the heading rotation (raft_member.py:72)
        }
        double *o = tile + t * 17;
        if (kind == 0) {
            for (int i = 0; i < RAFTX_GM_N; i++) o[i] = bm[i];
            if (ed) {
                const double c = E.headCS[2 * m], s = E.headCS[2 * m + 1];
                o[RAFTX_GM_RA + 0] = c * e[0] + (-s) * e[1];
                o[RAFTX_GM_RA + 1] = s * e[0] + c * e[1];
                o[RAFTX_GM_RA + 2] = e[2];
                o[RAFTX_GM_RB + 0] = c * e[3] + (-s) * e[4];
                o[RAFTX_GM_RB + 1] = s * e[3] + c * e[4];
                o[RAFTX_GM_RB + 2] = e[5];
                o[RAFTX_GM_L] = L;
            }
        } else if (kind == 1) {
            const double *b = E.gs + (size_t)idx * RAFTX_GS_N;
            for (int i = 0; i < RAFTX_GS_N; i++) o[i] = b[i];
            if (ed) {
                o[RAFTX_GS_S] = E.stFrac[idx] * L;
                o[RAFTX_GS_LFILL] = E.fillFrac[idx] * L;
            }
            if (E.diaEdit[idx]) {
                o[RAFTX_GS_D] = affine_eval(E.diaCoef + ((size_t)idx * 2 + 0) * (E.nP + 1), p, E.nP);
                o[RAFTX_GS_D + 1] = affine_eval(E.diaCoef + ((size_t)idx * 2 + 1) * (E.nP + 1), p, E.nP);
            }
        } else {
            const double *b = E.gc + (size_t)idx * RAFTX_GC_N;
            for (int i = 0; i < RAFTX_GC_N; i++) o[i] = b[i];
            if (ed) o[RAFTX_GC_S] = E.capFrac[idx] * L;
        }
    }
    __syncthreads();
    double *out = (kind == 0 ? E.gm_out : (kind == 1 ? E.gs_out : E.gc_out)) + row0 * W;
    const int npiece = nrows * W / 2;                                     // 16-byte pieces of the block's rows, consecutive in the output
    for (int i = t; i < npiece; i += GE_T) {
        const int r = (2 * i) / W, f = (2 * i) % W;
        *reinterpret_cast<double2 *>(out + 2 * (size_t)i) = make_double2(tile[r * 17 + f], tile[r * 17 + f + 1]);
    }
}
static inline void launch_expand(const ExpandArgs &E, hipStream_t st) {
    const unsigned nbM = (unsigned)(((size_t)E.n * E.nM + GE_T - 1) / GE_T), nbS = (unsigned)(((size_t)E.n * E.nSt + GE_T - 1) / GE_T),
                   nbC = (unsigned)(((size_t)E.n * E.nCap + GE_T - 1) / GE_T);
    if (nbM + nbS + nbC) hipLaunchKernelGGL(k_geom_expand, dim3(nbM + nbS + nbC), dim3(GE_T), 0, st, E, nbM, nbS);
}
