/* raftx.h -- C-ABI of the MI355X-native RAFT hot path (libraftx_hip.so).
 *
 * The reference (WISDEM/RAFT, pure Python) has no FFI; the boundary a
 * maintainer binds is the Python method surface
 *     raft/raft_model.py:966   Model.solveDynamics(case, tol=0.01, ...)
 *     raft/raft_fowt.py:1732   FOWT.calcHydroExcitation(case, memberList)
 *     raft/raft_fowt.py:1891   FOWT.calcHydroLinearization(Xi)
 *     raft/raft_fowt.py:1940   FOWT.calcDragExcitation(ih)
 * Each entry point below names the reference lines it replaces.  Plain C,
 * plain pointers and sizes; no C++ / torch types cross this boundary.
 *
 * Conventions
 *   - return 0 = ok, <0 = error (text via raftx_last_error()).
 *   - all pointers are caller-owned C-contiguous host buffers (NumPy
 *     arrays); raftx_upload_* copy to HBM, nothing is retained past a call.
 *   - complex numbers are interleaved (re, im) doubles == numpy complex128
 *     == C99 double _Complex.
 *   - frequency is the last (contiguous) axis of every array, as in the
 *     reference.
 *   - calls on one ctx are stream-ordered; different ctxs are independent;
 *     there is no global state.  One ctx per process per GPU.
 *   - all arithmetic is IEEE fp64.
 *
 * The same header is implemented twice: libraftx_hip.so (gfx950, the
 * product) and oracle/libraftx_oracle.so (plain C restatement of the
 * reference algorithm; TEST INFRASTRUCTURE ONLY).
 */
#ifndef RAFTX_H
#define RAFTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAFTX_VERSION 100          /* 0.1.0 */
#define RAFTX_NFIELD  32           /* doubles per strip record (256 B) */

/* strip record field offsets (doubles); packer: raft_amd/strips.py */
#define RAFTX_F_X     0   /* absolute position x,y,z      (raft_member.py:362 mem.r[il]) */
#define RAFTX_F_AX    3   /* arm about reduced-DOF ref pt  (raft_fowt.py:1919-1929 folded) */
#define RAFTX_F_Q     6   /* axial unit vector             (raft_member.py:370) */
#define RAFTX_F_P1    9   /* transverse unit vector 1      (raft_member.py:371) */
#define RAFTX_F_P2    12  /* transverse unit vector 2      (raft_member.py:372) */
#define RAFTX_F_IQ    15  /* rho*v_end*Ca_End              (raft_member.py:1442) */
#define RAFTX_F_IP1   16  /* rho*v_side*Cm_p1              (raft_member.py:1423) */
#define RAFTX_F_IP2   17  /* rho*v_side*Cm_p2 */
#define RAFTX_F_AI    18  /* signed end area a_i           (raft_member.py:1343,1347) */
#define RAFTX_F_DQ    19  /* sqrt(8/pi)*rho/2*a_q*Cd_q     (raft_member.py:2070,2093) */
#define RAFTX_F_DP1   20  /* ... a_p1*Cd_p1                (raft_member.py:2071,2094) */
#define RAFTX_F_DP2   21  /* ... a_p2*Cd_p2                (raft_member.py:2072,2095) */
#define RAFTX_F_DEND  22  /* ... |a_end|*Cd_End            (raft_member.py:2105-2110) */
#define RAFTX_F_CIRC  23  /* 1 circular, 0 rectangular     (raft_member.py:2085-2090) */
#define RAFTX_F_MCF   24  /* -1, or row in the complex Cm table (raft_member.py:1984-1985) */
#define RAFTX_F_RHOV  25  /* rho*v_side (times complex Cm when MCF) */
/* 26,27: member / strip indices (diagnostic).  28,29: optional run hints for the device
 * kinematics: strip = previous strip + STEP*UNIT*q (STEP in 1..4; 0 = start of a run,
 * evaluated exactly).  Verified against x,y,z at upload; the oracle ignores them. */
#define RAFTX_F_STEP  28
#define RAFTX_F_UNIT  29
/* 30,31: Morison added-mass scalars rho*v_side*Ca_p1, rho*v_side*Ca_p2 (raft_member.py:1333); written by
 * raftx_build_designs, optional (0) in hand-packed tables -- only raftx_fetch_statics reads them. */
#define RAFTX_F_AP1   30
#define RAFTX_F_AP2   31

/* flags[] bits written by raftx_solve_dynamics */
#define RAFTX_FLAG_CONVERGED 1    /* raft_model.py:1104 test passed */
#define RAFTX_FLAG_NAN       2    /* raft_model.py:1098-1099 would have raised */

typedef struct raftx_ctx raftx_ctx;
typedef struct { double re, im; } raftx_c128;

int         raftx_version(void);
/* 1 if this library computes on a GPU (libraftx_hip), 0 for the CPU oracle */
int         raftx_is_device(void);
/* GPUs this process can open (hipGetDeviceCount; 0 when there is none or the runtime cannot start).  bench.py's
 * rank launcher checks it against --gpus before it starts one process per GPU.  The oracle answers 0. */
int         raftx_device_count(void);
int         raftx_ctx_create(int device_id, raftx_ctx **out);
void        raftx_ctx_destroy(raftx_ctx *ctx);
const char *raftx_last_error(raftx_ctx *ctx);

/* Designs = independent floating units (sweep candidates, or the N units of
 * a farm).  stripOffsets[nDesign+1] indexes rows of strips[.,32].
 * M0/B0/C0 [nDesign,6,6] are the frequency-independent sums of
 * raft_model.py:1045-1047 (M_struc+A_hydro_morison(+moor), B_struc+sum B_gyro,
 * C_struc+C_hydro+C_moor+C_elast).  MBw (optional, may be NULL)
 * [nDesign,2,6,6,nw] carries the frequency-dependent parts (M_turb+A_BEM,
 * B_turb+B_BEM).  cmOffsets/CmMCF (optional) are the MacCamy-Fuchs complex
 * (Cm_p1,Cm_p2) rows [nRows,2,nw] (raft_member.py:1415-1420,1467-1484). */
int raftx_upload_designs(raftx_ctx *ctx, int nDesign, const int64_t *stripOffsets,
                         const double *strips, int nStripFields,
                         const double *M0, const double *B0, const double *C0,
                         int nw, const double *MBw,
                         const int64_t *cmOffsets, const raftx_c128 *CmMCF);

/* Sea states.  w,k [nw] come from the host (the reference's own dispersion
 * solve, helpers.py:377-392, tolerance 1e-3 -- never re-solved on device).
 * zeta [nCase,nHead,nw] = sqrt(2 S dw) (raft_fowt.py:1763-1769), beta
 * [nCase,nHead] in rad.  rho,g feed the dynamic pressure (helpers.py:231). */
int raftx_upload_cases(raftx_ctx *ctx, int nCase, int nHead, int nw,
                       const double *w, const double *k,
                       double depth, double rho, double g,
                       const double *zeta, const double *beta);

/* Strip-theory inertial excitation for every (design, case, heading):
 * raft_fowt.py:1854-1857,1888 + raft_member.py:1940-1992 + helpers.py:188-236.
 * F_iner [nDesign,nCase,nHead,6,nw]. */
int raftx_excitation(raftx_ctx *ctx, raftx_c128 *F_iner);

/* One drag linearisation about a given response: raft_fowt.py:1891-1957 +
 * raft_member.py:1995-2152.  Xi [nDesign,nCase,6,nw] ->
 * B_drag [nDesign,nCase,6,6], F_drag [nDesign,nCase,nHead,6,nw] (heading 0 is
 * what calcHydroLinearization stores; the others are calcDragExcitation(ih)).
 * Either output may be NULL. */
int raftx_linearize(raftx_ctx *ctx, const raftx_c128 *Xi, double *B_drag, raftx_c128 *F_drag);

/* Per-strip by-products the reference leaves on its Member objects, for callers that read them after the drop-in ran
 * (SURVEY.md 8b "side effects"): an un-replaced Member.calcDragExcitation (raft_member.py:2128-2152) reads mem.Bmat and
 * mem.u; post-processing reads mem.u / ud / pDyn.  The fused kernels never materialise these (979 KB per heading at C2);
 * these two calls evaluate them on request for ONE resident design and sea state, strips in table order:
 *   raftx_strip_kinematics  u, ud [nHead,S,3,nw], pDyn [nHead,S,nw]   raft_member.py:1927-1937 + helpers.py:188-236
 *   raftx_strip_drag        Bmat [S,3,3] of the linearisation about Xi [6,nw] (heading 0, raft_fowt.py:1910;
 *                           raft_member.py:2075-2117) and F_exc_drag [S,3,nw] = Bmat u[ih] (:2122, 2146)
 * S = strips of `design` in the resident set; any output may be NULL. */
int raftx_strip_kinematics(raftx_ctx *ctx, int design, int icase, raftx_c128 *u, raftx_c128 *ud, raftx_c128 *pDyn);
int raftx_strip_drag(raftx_ctx *ctx, int design, int icase, const raftx_c128 *Xi, int ih, double *Bmat, raftx_c128 *F_exc_drag);

/* The fused fixed-point solve, raft_model.py:994-1155 per unit plus the
 * per-heading response of :1189-1236 for an uncoupled unit:
 *   XiLast <- XiStart; repeat <= nIter+1 times { linearise about XiLast;
 *   Z = -w^2 M + i w (B + B_drag) + C; Xi = Z^-1 (F_lin + F_drag); converged if
 *   |Xi-XiLast|/(|Xi|+tol) < tol everywhere, else XiLast <- 0.2 XiLast + 0.8 Xi }
 * then for each heading ih  Xi[ih] = Z^-1 (F_extra[ih] + F_iner[ih] + F_drag(ih)).
 * nIter is the YAML setting (the loop runs nIter+1 times, raft_model.py:977).
 * F_extra (optional) [nDesign,nCase,nHead,6,nw] = F_BEM + Fhydro_2nd.
 * Outputs (any may be NULL): Xi [nDesign,nCase,nHead,6,nw]; niter, flags
 * [nDesign,nCase]; B_drag [nDesign,nCase,6,6]; F_wave [nDesign,nCase,nHead,6,nw]
 * (total excitation per heading, raft_model.py:1212); Z [nDesign,nCase,6,6,nw]. */
int raftx_solve_dynamics(raftx_ctx *ctx, int nIter, double tol, double XiStart,
                         const raftx_c128 *F_extra,
                         raftx_c128 *Xi, int32_t *niter, int32_t *flags,
                         double *B_drag, raftx_c128 *F_wave, raftx_c128 *Z);

/* Device-resident form of raftx_solve_dynamics for sweeps: identical maths,
 * but every output stays in ctx-owned HBM buffers (no D2H inside the call), so
 * a 10k-design batch is one kernel launch.  want_mask selects the optional
 * outputs to keep (RAFTX_WANT_*; Xi/niter/flags are always kept).  F_extra, if
 * given, is a host buffer uploaded before the launch.  Results are copied out
 * on demand with raftx_fetch_results (any pointer may be NULL). */
#define RAFTX_WANT_BDRAG 1
#define RAFTX_WANT_FWAVE 2
#define RAFTX_WANT_Z     4
int raftx_solve_dynamics_device(raftx_ctx *ctx, int nIter, double tol, double XiStart,
                                const raftx_c128 *F_extra, int want_mask);
int raftx_fetch_results(raftx_ctx *ctx, raftx_c128 *Xi, int32_t *niter, int32_t *flags,
                        double *B_drag, raftx_c128 *F_wave, raftx_c128 *Z);

/* Potential-flow (BEM) wave excitation with heading interpolation, the first block of FOWT.calcHydroExcitation
 * (raft/raft_fowt.py:1796-1849), for every (design, case, heading) of the uploaded designs and sea states:
 *   beta' = (deg(beta) - heading_adjust[d]) mod 360;  X' = X_BEM[d,i1] f1 + X_BEM[d,i2] f2  (neighbouring BEM headings,
 *   wrapping around 360);  rotate surge/sway and roll/pitch back by beta (:1838-1844);
 *   F_BEM = X zeta exp(-i k (x_ref cos beta + y_ref sin beta))                        (:1800-1801, 1847)
 * X_BEM [nDesign,nHeadBEM,6,nw]: excitation coefficients per unit wave amplitude in the wave-heading frame, as
 * FOWT.readHydro leaves them (:1485-1501; host feeder raft_amd/bem.py); headings_deg [nHeadBEM] ascending in [0,360);
 * heading_adjust [nDesign] deg and xy_ref [nDesign,2] m may be NULL (zeros).  F_add (optional host buffer
 * [nDesign,nCase,nHead,6,nw], e.g. second-order forces) is added.  The result stays RESIDENT and is the F_extra of
 * every following raftx_solve_dynamics[_device] call that passes F_extra == NULL (until designs or cases are
 * re-uploaded); F_out (optional) receives a host copy. */
int raftx_bem_excitation(raftx_ctx *ctx, int nHeadBEM, const double *headings_deg, const raftx_c128 *X_BEM,
                         const double *heading_adjust, const double *xy_ref, const raftx_c128 *F_add,
                         raftx_c128 *F_out);

/* Response statistics of the resident results of the last raftx_solve_dynamics_device -- the motion
 * block of FOWT.saveTurbineOutputs (raft/raft_fowt.py:2310-2357) with getRMS / getPSD
 * (raft/helpers.py:678-700), for rigid units whose reduced DOFs are the platform reference point:
 *   std[d,c,j]   = sqrt(0.5 * sum_{ih,w} |Xi[d,c,ih,j,w]|^2)           (j = 3..5 in degrees, :2332-2354)
 *   psd[d,c,j,w] = sum_ih 0.5 |Xi[d,c,ih,j,w]|^2 / dw                   (optional, may be NULL)
 * so that a sweep can return ~48 B per (design, case) instead of 19 KB.  std [nDesign,nCase,6],
 * psd [nDesign,nCase,6,nw]; dw = w[1]-w[0] (raft_fowt.py:169). */
int raftx_motion_stats(raftx_ctx *ctx, double dw, double *std, double *psd);

/* Restart / export of the fixed point's linearisation point, for the re-entry of raft_model.py:1108-1131
 * (internal QTFs: converge once, add the second-order force, iterate again FROM THE SAME Xi_last).
 * XiLast0 [nDesign,nCase,6,nw] (or NULL): the next raftx_solve_dynamics[_device] call starts from it
 * instead of XiStart (one-shot).  keep_last != 0: solve calls also keep the linearisation point of their
 * last iteration (the Xi_last the reference holds when its loop exits), readable with
 * raftx_fetch_linearisation_point (XiLast [nDesign,nCase,6,nw]). */
int raftx_set_linearisation_point(raftx_ctx *ctx, const raftx_c128 *XiLast0, int keep_last);
int raftx_fetch_linearisation_point(raftx_ctx *ctx, raftx_c128 *XiLast);

/* Statistics of LINEAR OUTPUT CHANNELS of the resident responses: channel c of design d is
 *   y_c(w) = w^pow[c] * sum_j L[d,c,j] Xi[d,case,ih,j,w]
 * (real L: a rigid-body transfer to another point, a tension Jacobian row, a unit conversion ...), and
 *   std[d,case,c] = sqrt(0.5 sum_{ih,w} |y_c|^2),   psd[d,case,c,w] = sum_ih 0.5 |y_c|^2 / dw   (optional).
 * This is the getRMS/getPSD pattern of FOWT.saveTurbineOutputs for the nacelle accelerations
 * (raft/raft_fowt.py:2422-2444: hub rows of T, pow = 2) and the quasi-static mooring tensions
 * (:2367-2373: rows of J_moor, pow = 0); raftx_motion_stats is the special case L = diag(1,1,1,deg,deg,deg).
 * L [nDesign,nChan,6], pow [nChan] (0..4), std [nDesign,nCase,nChan], psd [nDesign,nCase,nChan,nw] or NULL. */
int raftx_channel_stats(raftx_ctx *ctx, int nChan, const double *L, const int32_t *pow, double dw,
                        double *std, double *psd);

/* The general form: channels that mix displacement, velocity and acceleration terms and (optionally) a complex,
 * frequency-dependent transfer,
 *   y_c(w) = sum_{p=0..2} (i w)^p sum_j L[d,c,p,j] Xi[d,case,ih,j,w]  +  sum_j Gw[d,c,j,w] Xi[d,case,ih,j,w],
 * with the same std / psd definitions as raftx_channel_stats.  This is what the tower-base fore-aft bending moment
 * of FOWT.saveTurbineOutputs needs for a rigid tower (raft/raft_fowt.py:2500-2537): weight moment m g h Xi_pitch
 * (p = 0), inertial reaction -m a_CG h - I_CG (-w^2 Xi_pitch) (p = 2) and the aero reaction
 * -(-w^2 A_aero(w) + i w B_aero(w)) z^2 Xi_pitch (Gw); host feeder: raft_amd/dropin.py tower_base_rows.
 * L [nDesign,nChan,3,6] real; Gw [nDesign,nChan,6,nw] complex or NULL; std [nDesign,nCase,nChan];
 * psd [nDesign,nCase,nChan,nw] or NULL. */
int raftx_channel_stats_poly(raftx_ctx *ctx, int nChan, const double *L, const raftx_c128 *Gw, double dw,
                             double *std, double *psd);

/* The same statistics for a response the CALLER holds, with any number of degrees of freedom: units with flexible members
 * (raft_fowt.py's T reduction: 150 reduced DOFs for the flexible VolturnUS-S) are solved by raftx_flex_solve /
 * raftx_solve_dense, whose responses are returned, not kept.  Channel c:
 *   y_c(ih,w) = sum_{p=0..2} (i w)^p sum_j L[c,p,j] Xi[ih,j,w]  +  sum_j Gw[c,j,w] Xi[ih,j,w],
 *   std[c] = sqrt(0.5 sum_{ih,w} |y_c|^2),   psd[c,w] = sum_ih 0.5 |y_c|^2 / dw.
 * What FOWT.saveTurbineOutputs computes for such a unit: platform motions at the PRP (raft/raft_fowt.py:2299-2355), hub
 * accelerations (:2422-2444) and -- for a FLEXIBLE tower -- the internal loads at the tower base from the finite-element
 * stiffness, Fi_base = -(Kf Xi_internal)[base node] (:2540-2601), all linear in the reduced response through the rows of T
 * (host feeder: raft_amd/dropin.py general_output_rows).
 * w [nw]; L [nChan,3,nDof] real; Gw [nChan,nDof,nw] complex or NULL; Xi [nResp,nDof,nw] (nResp = wave headings + 1);
 * std [nChan]; psd [nChan,nw] or NULL.  Independent of the upload_* state of the ctx. */
int raftx_response_stats(raftx_ctx *ctx, int nChan, int nDof, int nResp, int nw, const double *w, const double *L,
                         const raftx_c128 *Gw, const raftx_c128 *Xi, double dw, double *std, double *psd);

/* Coupled array solve, raft_model.py:1164-1236: for each system s and bin w
 *   Z_sys = blockdiag_i(Zblk[s,i]) + (-w^2 Mc[s] + i w Bc[s] + Cc[s]);
 *   Xi[s,r] = Z_sys^-1 F[s,r].
 * Zblk [nSys,nUnit,6,6,nw]; Mc,Bc,Cc [nSys,6nUnit,6nUnit] or NULL;
 * F, Xi [nSys,nRhs,6nUnit,nw]. */
int raftx_solve_system(raftx_ctx *ctx, int nSys, int nUnit, int nRhs, int nw,
                       const double *w, const raftx_c128 *Zblk,
                       const double *Mc, const double *Bc, const double *Cc,
                       const raftx_c128 *F, raftx_c128 *Xi);

/* Impedance solve of ONE unit with n reduced DOFs (n > 6: flexible members, raft_model.py:1081-1088 with the
 * nDOF x nDOF matrices of raft_fowt.py's T reduction -- 150 for the reference's VolturnUS-S-flexible deck): per bin
 *   Z = -w^2 M + i w B + C,  Xi[r] = Z^-1 F[r].
 * M, B [n,n], or [n,n,nw] where bit 0 (M) / bit 1 (B) of freq_mask is set; C [n,n]; F, Xi [nRhs,n,nw];
 * Z [n,n,nw] out or NULL.  The drag linearisation of such a unit runs node by node through raftx_linearize (every
 * structural node with wet strips is one "design" about its own position, raft_member.py:2046-2056); the projections
 * with the unit's T matrix between the two calls are host glue (raft_amd/dropin.py Engine._solve_general). */
int raftx_solve_dense(raftx_ctx *ctx, int n, int nRhs, int nw, const double *w, const double *M, const double *B,
                      const double *C, int freq_mask, const raftx_c128 *F, raftx_c128 *Xi, raftx_c128 *Z);
/* The same for nSys systems of one size in ONE launch (a sweep of flexible units x sea states, raft_amd/flex.py): every
 * array gains a leading system axis -- M, B [nSys,n,n(,nw)], C [nSys,n,n], F, Xi [nSys,nRhs,n,nw], Z [nSys,n,n,nw] or
 * NULL; w [nw] and freq_mask are shared. */
int raftx_solve_dense_batch(raftx_ctx *ctx, int nSys, int n, int nRhs, int nw, const double *w, const double *M,
                            const double *B, const double *C, int freq_mask, const raftx_c128 *F, raftx_c128 *Xi,
                            raftx_c128 *Z);
/* The fixed point of such units re-solves with the SAME M, C and frequency-dependent part of B in every iteration
 * (raft_model.py:1045-1047, 1081-1088): only the drag linearisation B_drag and the drag excitation change (:1063-1064).
 * raftx_dense_resident keeps the matrices of nSet units on the device (shapes as raftx_solve_dense_batch; nSet = 0 releases
 * them); raftx_solve_dense_resident then solves nSet * nPer systems -- system s with the matrices of unit s / nPer (the
 * sea states of one unit share them) and B + Badd[s], Badd [nSet*nPer,n,n] or NULL -- F, Xi [nSet*nPer,nRhs,n,nw],
 * Z [nSet*nPer,n,n,nw] or NULL.  With the rotor's frequency-dependent matrices the flexible deck's M and B are 7.2 MB
 * each: an iteration then uploads 0.2 MB instead of 14.6. */
int raftx_dense_resident(raftx_ctx *ctx, int nSet, int n, int nw, const double *w, const double *M, const double *B,
                         const double *C, int freq_mask);
int raftx_solve_dense_resident(raftx_ctx *ctx, int nPer, const double *Badd, int nRhs, const raftx_c128 *F, raftx_c128 *Xi,
                               raftx_c128 *Z);
/* The whole fixed point of such units on the device, every (unit, sea state) of a batch at once: raft_model.py:1052-1155
 * with the projections of raft_fowt.py:1886-1888, 1912-1936 --
 *   XiLast <- XiStart; repeat <= nIter+1 times { node motions T_node XiLast; drag linearisation of every (node, sea state)
 *   (the strip kernels of raftx_linearize); B_drag = sum_nodes T_node^T B_node T_node, F_drag = sum_nodes T_node^T F_node;
 *   Xi = (-w^2 M + i w (B + B_drag) + C)^-1 (F_lin + F_drag)[heading 0]; converged (|Xi - XiLast| / (|Xi| + tol) < tol for all
 *   entries)? else XiLast <- 0.2 XiLast + 0.8 Xi }, then every heading with the impedance of the last iteration.
 * A pair that has converged keeps the results of ITS last iteration (as if it had been solved alone).
 * The RESIDENT designs are the wet structural nodes, unit after unit (raftx_upload_designs with the strip tables of
 * raft_amd/strips.py pack_fowt_nodes: node i of unit u is design nodeOff[u] + i, nodeOff [nUnit+1]); the resident sea
 * states (raftx_upload_cases) are shared by the units.  Tn [nNode,6,n]: the six rows of each node in its unit's T;
 * M, B [nUnit,n,n] or [nUnit,n,n,nw] (freq_mask bits 0, 1), C [nUnit,n,n]: raft_model.py:1045-1047; F_lin
 * [nUnit,nCase,nHead,n,nw]: every excitation but the drag's (:1048).
 * Out: Xi [nUnit,nCase,nHead,n,nw]; niter, flags [nUnit,nCase] (bit 0 converged, bit 1 NaN); B_drag [nUnit,nCase,n,n],
 * F_drag [nUnit,nCase,nHead,n,nw], Z [nUnit,nCase,n,n,nw] or NULL each. */
int raftx_flex_solve(raftx_ctx *ctx, int nUnit, const int64_t *nodeOff, int n, const double *Tn, const double *M,
                     const double *B, const double *C, int freq_mask, const raftx_c128 *F_lin, int nIter, double tol,
                     double XiStart, raftx_c128 *Xi, int32_t *niter, int32_t *flags, double *B_drag, raftx_c128 *F_drag,
                     raftx_c128 *Z);

/* The linearisation point the NEXT raftx_flex_solve starts from, instead of the constant XiStart (raft_model.py:999):
 * XiLast0 [nUnit,nCase,n,nw], copied to the device; one-shot (consumed by that call, which fails if its shape differs).
 * With nIter = 0 a call is then ONE pass of the loop body of raft_model.py:1058-1138 about a given iterate -- what a host
 * step between iterations needs: a unit's own lumped-mass mooring (moorMod == 2) has its line damping re-linearised by the
 * mooring model about every iterate (:1069-1072), also when the unit has more than 6 reduced DOFs (the matrices are lumped
 * at its first six, :1019-1030).  The caller relaxes (XiLast <- 0.2 XiLast + 0.8 Xi[heading 0], :1133) and stops on flag
 * bit 0.  NULL clears a point that was set and not used. */
int raftx_flex_start(raftx_ctx *ctx, int nUnit, int n, const raftx_c128 *XiLast0);

/* The same coupled solve fed from the RESIDENT results of raftx_solve_dynamics_device (which must have
 * kept Z and F_wave): consecutive groups of nUnit designs are the units of one array; for group g,
 * case c: Z_sys = blockdiag_u(Z[g*nUnit+u, c]) + (-w^2 Mc[g] + i w Bc[g] + Cc[g]),
 * Xi[g,c,ih] = Z_sys^-1 F_wave[., c, ih].  Mc,Bc,Cc [nGroup,6nUnit,6nUnit] or NULL;
 * Xi [nGroup,nCase,nHead,6nUnit,nw].  A 4-unit x 50-sea-state x 200-bin batch is two launches. */
int raftx_solve_system_resident(raftx_ctx *ctx, int nUnit, const double *Mc, const double *Bc, const double *Cc,
                                raftx_c128 *Xi);

/* Second-order (difference-frequency) slender-body QTF -- raft/raft_fowt.py:1988-2078
 * (FOWT.calcQTF_slenderBody: Pinkster IV term, member loop, Hermitian fill) and
 * raft/raft_member.py:1488-1674 (Member.calcQTF_slenderBody) with the helpers of raft/helpers.py:239-373 --
 * for nSet independent (strip table, motion RAOs, heading) sets on a common second-order grid w2,k2 [nw2].
 * strips [stripOff[nSet],24] / members [memOff[nSet],16]: records of raft_amd/qtf.py (QS_N, QM_N);
 * Xi [nSet,6,nw2] motion RAOs on that grid (zeros = fixed body), or NULL: the RAOs are then taken on the device from
 * the RESIDENT first-order responses of the last raftx_solve_dynamics_device (set s = (design, case) pair s, heading
 * 0): RAO = Xi / zeta where |zeta| > 1e-6 (raft/helpers.py:762-784), interpolated to w2 and zero outside the
 * first-order grid (raft_fowt.py:2022-2024) -- the internal-QTF flow of raft_model.py:1108-1131 without a round trip;
 * beta [nSet] rad; Mstruc [nSet,6,6];
 * kay [nSet,nw2,nw2,6] optional Kim & Yue table (raft_member.py:1676-1791; upper triangle, host feeder);
 * qtf [nSet,nw2,nw2,6] out, Hermitian-completed (may be NULL: the result then only stays resident in HBM
 * for raftx_qtf_force).  Independent of the upload_* state of the ctx. */
int raftx_qtf_slender(raftx_ctx *ctx, int nSet, int nw2, const double *w2, const double *k2,
                      double depth, double rho, double g,
                      const int64_t *stripOff, const double *strips, const int64_t *memOff, const double *members,
                      const raftx_c128 *Xi, const double *beta, const double *Mstruc,
                      const raftx_c128 *kay, raftx_c128 *qtf);

/* Kim & Yue second-order diffraction correction of the MacCamy-Fuchs members on the device --
 * Member.correction_KAY, raft/raft_member.py:1676-1791 -- for nSet sets on the grid w2,k2 [nw2]:
 *   waterline term  F = Re( sum_{n=0..Nm} -rho g R 2i/(pi k1R k2R) Omega_n ) e^{-i (k1-k2)(x cosB + y sinB)},
 *   segment terms   the same sum weighted by the depth integrals I-, I+ of :1757-1779,
 *   Omega_n = 1/(H'_{n+1}(k1R) conj H'_n(k2R)) - 1/(H'_n(k1R) conj H'_{n+1}(k2R)),  H'_n = (H_{n-1} - H_{n+1})/2,
 * each applied along the member's unit force direction at its moment arm, conjugated where k1 < k2 (:1787-1788);
 * upper triangle (w2 >= w1) only, like the host table it replaces.
 * items [itemOff[nSet], RAFTX_QK_N]: records of raft_amd/qtf.py kay_items (R, kind, z1, z2, arm, pforce, phase x y);
 * beta [nSet] rad.  The table [nSet,nw2,nw2,6] stays resident and is CONSUMED by the next raftx_qtf_slender[_rows]
 * call on this ctx that passes kay == NULL with the same nSet, nw2 (one-shot); kay_out (optional) receives a copy. */
#define RAFTX_QK_N 12
int raftx_qtf_kay(raftx_ctx *ctx, int nSet, int nw2, const double *w2, const double *k2, double depth, double rho,
                  double g, const int64_t *itemOff, const double *items, const double *beta, int Nm,
                  raftx_c128 *kay_out);

/* One QTF shared by several ranks: the same call restricted to the rows w1 = w2[row_off + m*row_stride] (and their
 * Hermitian mirrors); every other entry of qtf is returned as 0, so the partial results of row_stride ranks SUM to the
 * full matrix.  Rows are interleaved because their cost is triangular (row i1 holds nw2 - i1 pairs). */
int raftx_qtf_slender_rows(raftx_ctx *ctx, int nSet, int nw2, const double *w2, const double *k2,
                           double depth, double rho, double g,
                           const int64_t *stripOff, const double *strips, const int64_t *memOff, const double *members,
                           const raftx_c128 *Xi, const double *beta, const double *Mstruc,
                           const raftx_c128 *kay, int row_off, int row_stride, raftx_c128 *qtf);

/* Second-order difference-frequency force amplitudes from QTFs -- FOWT.calcHydroForce_2ndOrd with
 * interpMode='qtf' (raft/raft_fowt.py:2209-2245): for every set, bilinear interpolation of the QTF from the
 * second-order grid w2 onto the first-order grid w (outside the grid: 0, like RegularGridInterpolator with
 * fill_value=0), then  f[j,mu] = 4 sqrt(sum_i S0[i] S0[i+mu] |Q_j(w_i, w_{i+mu})|^2) dw,
 * f_mean[j] = 2 sum_i S0[i] Re Q_j(w_i,w_i) dw, and the one-bin shift of :2241-2245.
 * qtf [nSet,nw2,nw2,6] host buffer, or NULL to use the QTFs left resident by the last raftx_qtf_slender
 * call on this ctx (same nSet, nw2).  S0 [nSet,nw]; f_mean [nSet,6]; f [nSet,6,nw] (real amplitudes). */
int raftx_qtf_force(raftx_ctx *ctx, int nSet, int nw2, const double *w2, const raftx_c128 *qtf,
                    int nw, const double *w, double dw, const double *S0, double *f_mean, double *f);

/* ------------------------------------------------------------------------------------------------
 * Geometry -> strip tables + statics on the device (the step BEFORE the hot path).
 *
 * A design is a rigid 6-DOF floating unit described by its members, exactly as the reference's YAML does
 * (docs/usage.rst "platform: members"): end points, stations, diameters / side pairs, coefficients.
 * raftx_build_designs discretises every member into Morison strips, places them at the unit's mean pose,
 * evaluates the per-strip hydrodynamic constants and leaves the result resident as the ctx's design set --
 * the equivalent of Member.__init__ (strip discretisation, raft/raft_member.py:190-271), Member.setPosition
 * (:312-377), Member.calcHydroConstants / calcImat / getCmSides (:1261-1486), the drag areas of
 * Member.calcHydroLinearization (:2061-2110) and FOWT.calcHydroConstants (raft/raft_fowt.py:1589-1625),
 * followed by raftx_upload_designs -- without the 256 B/strip host packing and upload.
 *
 * Member descriptor (RAFTX_GM_N doubles), host feeder raft_amd/geometry.py: */
#define RAFTX_GM_N        16
#define RAFTX_GM_RA       0   /* end A relative to the unit's reference point, member heading applied (raft_member.py:41,75-77) */
#define RAFTX_GM_RB       3   /* end B */
#define RAFTX_GM_GAMMA    6   /* twist about the axis [deg]; 0 for circular (raft_member.py:70,79-80,106) */
#define RAFTX_GM_SHAPE    7   /* 1 circular, 0 rectangular (raft_member.py:102-114) */
#define RAFTX_GM_DLSMAX   8   /* maximum strip length (raft_member.py:202) */
#define RAFTX_GM_FLAGS    9   /* bit 0: potMod (no strip-theory inertia/added mass), bit 1: MacCamy-Fuchs */
#define RAFTX_GM_L        10  /* member length |rB - rA| (raft_member.py:72) */
#define RAFTX_GM_RHOSHELL 11  /* shell density (raft_member.py:124) */
#define RAFTX_GM_FLAG_POTMOD   1
#define RAFTX_GM_FLAG_MCF      2
#define RAFTX_GM_FLAG_NOSTATIC 4   /* nacelle members: strips only, left out of the statics (raft_fowt.py:876) */
/* Station record (RAFTX_GS_N doubles); stations of member m are rows stationOff[m]..stationOff[m+1]: */
#define RAFTX_GS_N        16
#define RAFTX_GS_S        0   /* position along the axis from end A [m] (raft_member.py:99) */
#define RAFTX_GS_D        1   /* diameter, or the two side lengths (1,2) (raft_member.py:104,111) */
#define RAFTX_GS_T        3   /* shell thickness (raft_member.py:123) */
#define RAFTX_GS_CD       4   /* Cd_q, Cd_p1, Cd_p2, Cd_End (raft_member.py:178-181) */
#define RAFTX_GS_CA       8   /* Ca_q, Ca_p1, Ca_p2, Ca_End (raft_member.py:184-187) */
#define RAFTX_GS_LFILL    12  /* ballast fill length of the section that STARTS at this station [m] (:143) */
#define RAFTX_GS_RHOFILL  13  /* ballast density of that section (:146-155) */
/* End caps / bulkheads (RAFTX_GC_N doubles; caps of member m are rows capOff[m]..capOff[m+1]; raft_member.py:162-175,
 * 659-810): position along the axis [m], thickness, inner (hole) diameter or side pair. */
#define RAFTX_GC_N        4
#define RAFTX_GC_S        0
#define RAFTX_GC_T        1
#define RAFTX_GC_DIN      2
/* add_mask: what the call adds to the caller's M0 / C0 before installing them (0 = M0/B0/C0 used as given,
 * exactly like raftx_upload_designs) */
#define RAFTX_ADD_MORISON     1   /* M0 += A_hydro_morison (raft_fowt.py:1625) */
#define RAFTX_ADD_HYDROSTATIC 2   /* C0 += C_hydro         (raft_fowt.py:1214-1256) */
#define RAFTX_ADD_INERTIA     4   /* M0 += M_struc, C0 += C_struc of the described members (raft_fowt.py:876-900,1120-1199) */
/* Ballast trim in heave BEFORE the statics are taken, Model.adjustBallastDensity (raft/raft_model.py:1772-1827, the
 * ballast == 2 option of analyzeUnloaded :247-248): sections with zero fill density lose their fill length; with
 *   sumFz = -(M_struc[0,0] + M0[0,0]) g + V rho g + Fz_moor,   drho = sumFz / g / (total ballast volume),
 * every ballasted section's density is raised by drho and the inertia / weight terms are recomputed (M0[0,0] is the
 * mass that is not geometry: rotor-nacelle assembly, point masses).  props[RAFTX_SP_DRHO] returns drho. */
#define RAFTX_TRIM_BALLAST    8
/* memberOff [nDesign+1] rows of members[.,RAFTX_GM_N]; stationOff [nMember+1] rows of stations[.,RAFTX_GS_N];
 * pose [nDesign,6] = mean position of the reduced DOFs (x,y,z,roll,pitch,yaw; FOWT.setPosition's argument,
 * raft_fowt.py:754) or NULL for zeros; rho, g: water density and gravity (raft_fowt.py:172-173);
 * k [nw] wave numbers for the MacCamy-Fuchs Cm table (required iff a member carries RAFTX_GM_FLAG_MCF);
 * M0,B0,C0,MBw as in raftx_upload_designs; Fz_moor [nDesign] vertical mooring force at the undisplaced position for
 * RAFTX_TRIM_BALLAST (NULL = 0).  stripOffsets [nDesign+1] (out): submerged strips per design.
 * Only strips below the mean waterline are kept (raft_member.py:1310,1979,2058). */
int raftx_build_designs(raftx_ctx *ctx, int nDesign, const int64_t *memberOff, const double *members,
                        const int64_t *stationOff, const double *stations,
                        const int64_t *capOff, const double *caps, const double *pose,
                        double rho, double g, int nw, const double *k, int add_mask,
                        const double *M0, const double *B0, const double *C0, const double *MBw,
                        const double *Fz_moor, int64_t *stripOffsets);
/* The strip records (ABI layout, [nStrips,RAFTX_NFIELD]) and MacCamy-Fuchs rows ([nRows,2,nw], may be NULL)
 * that raftx_build_designs generated -- what raft_amd/strips.py would have packed on the host. */
int raftx_fetch_strips(raftx_ctx *ctx, double *strips, raftx_c128 *cm);
/* Per-design statics of the last raftx_build_designs call (any pointer may be NULL):
 *   A_morison [nDesign,6,6]  FOWT.A_hydro_morison (raft_fowt.py:1589-1625), about the unit's reference point
 *   C_hydro   [nDesign,6,6]  hydrostatic stiffness (raft_member.py:838-1010, raft_fowt.py:1214-1256)
 *   W_hydro   [nDesign,6]    buoyancy force/moment vector (same lines)
 *   M_struc   [nDesign,6,6]  mass/inertia of the described members: shells, ballast, caps (raft_member.py:380-836)
 *   C_struc   [nDesign,6,6]  weight part of the stiffness (raft_member.py:1179-1181, raft_fowt.py:1123,1191,1200)
 *   W_struc   [nDesign,6]    weight force/moment vector (helpers.py:1060-1082)
 *   props     [nDesign,12]   V (displaced volume), AWP, rCB x,y,z, mass, rCG x,y,z (RAFTX_SP_*)
 * Rotor-nacelle assemblies, point inertias and moorings are not geometry: the caller adds them through M0/C0. */
#define RAFTX_SP_N     12
#define RAFTX_SP_V     0
#define RAFTX_SP_AWP   1
#define RAFTX_SP_RCB   2   /* x,y,z of the centre of buoyancy (raft_fowt.py:1245) */
#define RAFTX_SP_MASS  5
#define RAFTX_SP_RCG   6   /* x,y,z of the centre of mass (raft_fowt.py:1210) */
#define RAFTX_SP_DRHO  9   /* ballast density change of RAFTX_TRIM_BALLAST [kg/m^3] */
#define RAFTX_SP_VFILL 10  /* total ballast volume [m^3] */
int raftx_fetch_statics(raftx_ctx *ctx, double *A_morison, double *C_hydro, double *W_hydro,
                        double *M_struc, double *C_struc, double *W_struc, double *props);

/* One whole SWEEP CROSSING in one call ("H2D of the tables + kernels + D2H"): member descriptions of
 * nDesign candidates in, response statistics (and optionally the responses) out -- raftx_build_designs +
 * raftx_upload_cases + raftx_solve_dynamics_device + raftx_motion_stats + raftx_fetch_results, with the designs cut
 * into a few contiguous blocks (device buffers and memory pool per block, kept for the life of ctx) that one host thread
 * drives over internal streams: the descriptor H2D of every block back to back on a copy stream, the member pass and
 * the scans of a block on a preparation stream as soon as its descriptors have landed, and strip generation, the fused
 * fixed point and the statistics of block 0, 1, 2, ... one after the other on the ctx stream (so the HIP events around
 * each fixed-point launch time that launch alone); the responses, if asked for, follow on a download stream -- block by
 * block, or, when no other crossing is in flight, in slabs of one residency round of the fused kernel (each block's launch
 * is then cut into slabs of the pair list on a slab stream, every slab followed by its own download; timing_ms[2] is then
 * the span from the first slab's launch to the last slab's end).  What the
 * reference does per candidate in raft/parametersweep.py:39-100 / raft/omdao_raft.py:746-792 (build a Model,
 * analyzeCases, read the statistics).
 * Descriptor arguments as raftx_build_designs (MBw not supported here; k doubles as the wave numbers of the
 * MacCamy-Fuchs table); sea-state arguments as raftx_upload_cases (rho_wave, g_wave scale the dynamic pressure:
 * the reference hard-wires 1025 / 9.81 there, raft_fowt.py:1857); nIter, tol, XiStart as raftx_solve_dynamics.
 * nChunk > 0: that many equal blocks; nChunk <= 0: library default (an isolated call: a small first block whose kernels hide
 * the descriptor upload of the rest; a crossing submitted while the other slot is in flight: one block).  nWorker is reserved (ignored).  Page-locked descriptor and M0/B0/C0 arrays
 * (raftx_host_alloc) copy at full PCIe rate and without blocking the host; a page-locked Xi likewise.
 * Outputs: std [nDesign,nCase,6] (raftx_motion_stats; required), niter / flags [nDesign,nCase] (required),
 * Xi [nDesign,nCase,nHead,6,nw] or NULL, stripOffsets [nDesign+1] or NULL, timing_ms [4] or NULL
 * (wall, sum of generation kernels, sum of solve kernels, sum of statistics kernels).
 * Results are bit-identical to the unchunked sequence of calls: designs do not interact. */
int raftx_sweep_stats(raftx_ctx *ctx, int nDesign, const int64_t *memberOff, const double *members,
                      const int64_t *stationOff, const double *stations, const int64_t *capOff, const double *caps,
                      const double *pose, double rho, double g, int add_mask,
                      const double *M0, const double *B0, const double *C0, const double *Fz_moor,
                      int nCase, int nHead, int nw, const double *w, const double *k, double depth,
                      double rho_wave, double g_wave, const double *zeta, const double *beta,
                      int nIter, double tol, double XiStart, int nChunk, int nWorker,
                      double *std, int32_t *niter, int32_t *flags, raftx_c128 *Xi, int64_t *stripOffsets,
                      double *timing_ms);

/* The same crossing in stages, for back-to-back batches of a long sweep (a 10^6-candidate sweep is a stream of 10^4-design
 * batches), on slots 0 .. RAFTX_SWEEP_SLOTS - 1:
 *   raftx_sweep_prepare  enqueues the descriptor upload and the member pass of a batch (returns at once; arguments as
 *                        raftx_sweep_stats, all arrays must stay alive and untouched until the batch has been waited for);
 *   raftx_sweep_launch   enqueues table generation, the fused fixed point and the statistics of a prepared batch (the host
 *                        waits only for the few bytes of the member pass that size the strip tables -- long there when
 *                        the batch was prepared a step ahead);
 *   raftx_sweep_wait     blocks until that batch has finished and fills the output arrays given at prepare time
 *                        (timing_ms as raftx_sweep_stats, [0] = host time from prepare to the end of wait).
 * raftx_sweep_submit = prepare + launch.  With launch(i+1), prepare(i+2), wait(i) per step, three batches are in flight:
 * batch i solving; batch i+1 with its member pass done a step earlier, so that its tables are generated in the drain of
 * batch i's fused kernel and its own fused kernel follows without a gap; batch i+2 uploading.  When the
 * responses are downloaded (192 MB per 10 k-design batch: 3.5 ms of PCIe, longer than the batch's kernels) a fourth slot
 * keeps the download off the critical path: prepare(i+3), launch(i+2), wait(i) -- batch i downloading, i+1 solving, i+2
 * queued behind it, i+3 uploading.
 * raftx_sweep_stats is submit + wait on a free slot.  Results are bit-identical whatever the staging. */
#define RAFTX_SWEEP_SLOTS 4
int raftx_sweep_prepare(raftx_ctx *ctx, int slot, int nDesign, const int64_t *memberOff, const double *members,
                        const int64_t *stationOff, const double *stations, const int64_t *capOff, const double *caps,
                        const double *pose, double rho, double g, int add_mask,
                        const double *M0, const double *B0, const double *C0, const double *Fz_moor,
                        int nCase, int nHead, int nw, const double *w, const double *k, double depth,
                        double rho_wave, double g_wave, const double *zeta, const double *beta,
                        int nIter, double tol, double XiStart, int nChunk,
                        double *std, int32_t *niter, int32_t *flags, raftx_c128 *Xi, int64_t *stripOffsets);
int raftx_sweep_launch(raftx_ctx *ctx, int slot);
int raftx_sweep_submit(raftx_ctx *ctx, int slot, int nDesign, const int64_t *memberOff, const double *members,
                       const int64_t *stationOff, const double *stations, const int64_t *capOff, const double *caps,
                       const double *pose, double rho, double g, int add_mask,
                       const double *M0, const double *B0, const double *C0, const double *Fz_moor,
                       int nCase, int nHead, int nw, const double *w, const double *k, double depth,
                       double rho_wave, double g_wave, const double *zeta, const double *beta,
                       int nIter, double tol, double XiStart, int nChunk,
                       double *std, int32_t *niter, int32_t *flags, raftx_c128 *Xi, int64_t *stripOffsets);
int raftx_sweep_wait(raftx_ctx *ctx, int slot, double *timing_ms);
/* Where on the device's clock the fused fixed point of the crossing LAST WAITED FOR on `slot` ran: start of its first
 * launch and end of its last one, in ms since the first crossing of this ctx was launched.  Crossings of consecutive
 * slots run on alternating streams and their fused kernels overlap (the drain of batch i is the ramp of batch i+1), so
 * the busy time of k_solve_dynamics over a stream of batches is the UNION of these spans, not the sum of the per-batch
 * durations of timing_ms[2] (bench.py: roofline.kernel_ms_per_step).  The CPU oracle returns zeros.
 * (the loop the kernel fuses: raft/raft_model.py:1052-1142) */
int raftx_sweep_solve_span(raftx_ctx *ctx, int slot, double *start_ms, double *end_ms);
/* Where the strip tables of the crossing LAST LAUNCHED on `slot` were generated: *blocks_fused = the number of its blocks
 * whose tables were built INSIDE the fused fixed point, by the workgroup that solves the design (raftx_kpg_f0,
 * raft_amd/csrc/raftx_fusedgen.h: streamed crossings with one sea state per design and no MacCamy-Fuchs rows),
 * *blocks = all its blocks; the others went through k_geom_design, a kernel of its own.  Either way the tables are what
 * Member.__init__ / calcHydroConstants produce (raft/raft_member.py:190-271, 1261-1368, 2061-2110), bit for bit.
 * The fused form is opt-in (RAFTX_FUSED_GEN=1 in the environment): it closes the gap between consecutive fused kernels but
 * costs the kernel more than it saves (profiles/r06_experiments/fused_generation_ab.txt).  The CPU oracle reports 0. */
int raftx_sweep_generation(raftx_ctx *ctx, int slot, int *blocks_fused, int *blocks);
/* Retires a batch that was prepared and will not be launched (its uploads and member pass are drained, its scratch is
 * released, its outputs are left untouched).  No-op on an idle slot; an error on a launched one (raftx_sweep_wait
 * collects that).  raftx_ctx_destroy retires whatever is still prepared.
 * Sea states and slots: a batch is solved with the sea-state tables it was PREPARED with.  The library keeps one resident
 * set of tables per distinct (w, k, zeta, beta, depth, rho_wave, g_wave) among the batches in flight (identical tables are
 * uploaded once and shared), and a set is released only when the last batch prepared with it has been waited for or
 * cancelled -- so consecutive batches of a stream may change sea states freely, at the cost of one small upload. */
int raftx_sweep_cancel(raftx_ctx *ctx, int slot);

/* ------------------------------------------------------------------------------------------------
 * Parametric VARIANTS of one base unit, expanded on the device (raft/parametersweep.py:39-87: a sweep edits a handful of
 * parameters per candidate -- column diameters, draft, column radius, pontoon height -- and the dependent geometry
 * follows).  The host describes the base unit once and, per candidate, sends only its nParam parameter values; the
 * library writes each candidate's member / station / cap descriptors in HBM (k_geom_expand) and goes on as
 * raftx_sweep_prepare does.  Per batch of 10^4 candidates: 0.4 MB of parameters instead of 66 MB of descriptors, and no
 * host work per candidate.
 *
 * raftx_variant_program: the base unit's descriptors (as raftx_build_designs takes them, ONE design: nMember members,
 *   stationOff / capOff [nMember+1], capOff / caps may be NULL) and the edit program.  Edits are AFFINE in the parameters:
 *     endEdit [nMember] != 0: the member's end points BEFORE its heading rotation are
 *         (rA, rB)[i] = endCoef[m][i][0] + sum_p endCoef[m][i][1+p] * param[p]      endCoef [nMember,6,nParam+1]
 *       evaluated left to right without fused multiply-adds; the length |rB - rA| follows (raft_member.py:72), the ends
 *       are rotated by the member's heading, headCS [nMember,2] = (cos, sin) of it (raft_member.py:75-77,
 *       helpers.py:587-602), and the positions of the member's stations, ballast fills and caps keep their FRACTION of
 *       the length (raft_member.py:99,143,173: the deck's station units are arbitrary).
 *     diaEdit [nStation] != 0: the station's diameter / side pair is (d, d2)[j] = diaCoef[s][j][0] + sum_p ... [nStation,2,nParam+1].
 *   Everything else of a variant is the base unit's.  The program stays on ctx until replaced (nMember = 0 clears it).
 * raftx_expand_variants: the descriptors of nDesign variants back on the host (members [nDesign*nMember,RAFTX_GM_N],
 *   stations [nDesign*nStation,RAFTX_GS_N], caps [nDesign*nCap,RAFTX_GC_N] or NULL; offsets are uniform:
 *   memberOff[d] = d*nMember, stationOff[d*nMember+m] = d*nStation + base stationOff[m]) -- for checking and for feeding
 *   another library; the sweep path never downloads them.
 * raftx_sweep_prepare_variants: raftx_sweep_prepare with params [nDesign,nParam] in place of the six descriptor
 *   arguments; raftx_sweep_launch / _wait / _cancel as usual (params must stay alive until the batch has been waited for). */
int raftx_variant_program(raftx_ctx *ctx, int nMember, const double *members, const int64_t *stationOff, const double *stations,
                          const int64_t *capOff, const double *caps, int nParam,
                          const double *endCoef, const int32_t *endEdit, const double *headCS,
                          const double *diaCoef, const int32_t *diaEdit);
int raftx_expand_variants(raftx_ctx *ctx, int nDesign, const double *params, double *members, double *stations, double *caps);
int raftx_sweep_prepare_variants(raftx_ctx *ctx, int slot, int nDesign, const double *params,
                                 const double *pose, double rho, double g, int add_mask,
                                 const double *M0, const double *B0, const double *C0, const double *Fz_moor,
                                 int nCase, int nHead, int nw, const double *w, const double *k, double depth,
                                 double rho_wave, double g_wave, const double *zeta, const double *beta,
                                 int nIter, double tol, double XiStart, int nChunk,
                                 double *std, int32_t *niter, int32_t *flags, raftx_c128 *Xi, int64_t *stripOffsets);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU exchange steps: one process per GPU, one RCCL communicator per ctx, xGMI underneath.
 * The path shards with NO collective while kernels run; these calls are the once-per-batch exchanges around it:
 * the shared sea-state tables out (broadcast), the responses / statistics / QTF partials back (gather, reduce).
 * Every rank of the communicator must make the same call.  All calls are enqueued on the ctx's stream and return
 * after it has drained.  Each exchange step first validates its arguments and allocates its buffers locally, then the
 * ranks MAX-reduce one status word: if any rank cannot take part (bad argument, allocation failure) the call fails on
 * every rank with an error instead of leaving the others blocked in a send / receive that is never posted.  The status
 * word lives on the device since raftx_comm_init and a local HIP error on the way into the vote is folded into the vote,
 * so the vote itself is always posted.  NOT recoverable (the peers stay in the collective until RCCL's own watchdog /
 * the caller's deadline, raft_amd/comm.py `timeout`): a rank whose process dies, a rank whose status AllReduce itself
 * fails, or a copy that fails on the root in the MIDDLE of the grouped sends / receives after a successful vote.
 * (The CPU oracle does not implement these calls: its tests use the host transport of raft_amd/comm.py.)
 *
 * raftx_comm_unique_id: rank 0 creates the 128-byte RCCL unique id and hands it to the other ranks by any host
 *   channel (raft_amd/comm.py: a TCP rendezvous on MASTER_ADDR);  raftx_comm_init: collective, binds ctx to
 *   (rank, world);  raftx_comm_destroy: releases the communicator (also done by raftx_ctx_destroy). */
#define RAFTX_COMM_ID_BYTES 128
int raftx_comm_unique_id(raftx_ctx *ctx, char *id128);
int raftx_comm_init(raftx_ctx *ctx, int rank, int world, const char *id128);
int raftx_comm_destroy(raftx_ctx *ctx);
/* Host buffers, staged through HBM inside the call: buf [bytes] is sent by root and overwritten everywhere else
 * (ncclBroadcast) -- the shared case tables w, k, zeta, beta of a sweep (a few KB). */
int raftx_comm_broadcast(raftx_ctx *ctx, void *buf, size_t bytes, int root);
/* Gather-to-root of ragged row blocks: rank r contributes counts[r] rows of row_bytes bytes (send, host);
 * root receives them back to back in rank order (recv, host, sum(counts) rows; ignored elsewhere).  Point-to-point
 * ncclSend/ncclRecv inside one group: only root's links carry data (no all-gather of 24 MB per rank to everyone). */
int raftx_comm_gather_rows(raftx_ctx *ctx, const void *send, const int64_t *counts, size_t row_bytes, void *recv,
                           int root);
/* The same for the RESIDENT responses of the last raftx_solve_dynamics_device: rank r's Xi
 * [counts[r] (= its nDesign*nCase), nHead,6,nw] goes straight from its HBM buffer to root's HBM and from there to
 * root's host array Xi_all (page-locked for full PCIe rate); no host bounce on the sending ranks. */
int raftx_comm_gather_xi(raftx_ctx *ctx, const int64_t *counts, raftx_c128 *Xi_all, int root);
/* Element-wise SUM of n doubles onto root (ncclReduce; buf host, overwritten on root only): the partial QTFs of the
 * interleaved row partition (raftx_qtf_slender_rows), 3.8 MB for the 200-point grid. */
int raftx_comm_reduce_sum(raftx_ctx *ctx, double *buf, size_t n, int root);

/* Page-locked host buffers for the bulk outputs (the 19 KB per design-case of raftx_fetch_results): copies into
 * them run at full PCIe rate and asynchronously to other streams, which pageable NumPy memory does not.  The caller
 * wraps the pointer in an array (raft_amd/_abi.py Context.pinned_empty) and must return it with raftx_host_free before
 * the ctx is destroyed.  The oracle hands out ordinary malloc memory. */
int raftx_host_alloc(raftx_ctx *ctx, size_t bytes, void **out);
int raftx_host_free(raftx_ctx *ctx, void *ptr);

/* Where device `device` sits on the host: its PCI address (into pci_bus_id, at most len bytes incl. the terminator)
 * and the NUMA node sysfs reports for it (-1 when unknown).  One process per GPU feeds 16 GB/s of descriptors from
 * page-locked memory: raft_amd/locality.py pins the process to that node's cores BEFORE the ctx and its buffers are
 * created, so that on a two-socket host no rank's stream crosses the socket link.  The oracle reports ("", -1). */
int raftx_device_locality(int device, char *pci_bus_id, int len, int *numa_node);

/* Duration (ms) of the device work of the last raftx_excitation /
 * raftx_linearize / raftx_solve_dynamics / raftx_solve_system call on this
 * ctx, measured with HIP events on the ctx's own stream (excludes H2D/D2H).
 * The oracle returns host wall time of the compute loop. */
double raftx_last_kernel_ms(raftx_ctx *ctx);

/* Diagnostics: evaluates the device's own fp64 sincos/exp on n host values (the
 * oracle answers with libm), so the elementary functions are testable alone. */
int raftx_debug_math(raftx_ctx *ctx, int n, const double *x, double *sin_out, double *cos_out, double *exp_out);
/* Blocks until every stream of the context's DEVICE is idle (hipDeviceSynchronize): the device-side half of a timed
 * region's barrier, for callers that bracket library calls themselves (bench.py).  The oracle returns at once. */
int raftx_device_synchronize(raftx_ctx *ctx);

/* Diagnostics: which specialisation of the fused fixed point the last solve on this ctx launched -- the feature bits
 * compiled in (1 frequency-dependent M/B, 2 Z out, 4 F_wave out, 8 extra excitation, 16 MacCamy-Fuchs, 32 several
 * headings, 64 linearisation-point I/O; 127 = the full-featured kernel), the waves per SIMD it is compiled for and the
 * slots of its LDS run-start cache.  The oracle answers 0, 0, 0.  Returns -1 before the first solve. */
int raftx_last_solve_kernel(raftx_ctx *ctx, int *flags, int *waves_per_simd, int *cache_slots);
/* The same with the table-driven sincos the fused fixed point uses at its run starts (64-entry table in LDS). */
int raftx_debug_math_table(raftx_ctx *ctx, int n, const double *x, double *sin_out, double *cos_out, double *exp_out);
/* Test hook: out [n,n] = A^T W for A, W [K,n] through the projection kernel of raftx_flex_solve (MFMA tiles): checked
 * against NumPy with an ASYMMETRIC product (the projections themselves are symmetric and would hide a transposed tile). */
int raftx_debug_flex_gemm(raftx_ctx *ctx, int K, int n, const double *A, const double *W, double *out);

#ifdef __cplusplus
}
#endif
#endif /* RAFTX_H */
