// raftx_dense.h -- dense complex impedance solves for units with more than 6 reduced DOFs (flexible members):
// k_solve_dense (any size, L2-resident workspace), k_solve_dense_reg (1024 threads, registers + an LDS tail),
// k_solve_dense_reg2 (512 threads, all registers).  Included by raftx_hip.hip (after raftx_kernels.h: cplx, cmul, csub) and by
// scripts/ubench/dense_probe.hip, the stand-alone timing harness of these kernels.
#pragma once
#include <type_traits>

__device__ __forceinline__ void argmax_step(double &v, int &r, int off) {
    const double ov = __shfl_xor(v, off, 64);
    const int orr = __shfl_xor(r, off, 64);
    if (ov > v || (ov == v && orr < r)) {
        v = ov;
        r = orr;
    }
}

// General dense impedance solve for units with more than 6 reduced DOFs (flexible members: raft_model.py:1081-1088 with
// nDOF x nDOF matrices, 150 for the reference's flexible VolturnUS-S).  One workgroup per frequency bin; the augmented
// matrix [Z | F] of the bin lives in a global workspace (n = 150: 360 KB, beyond LDS, resident in L2), the pivot row and
// the multiplier column of every elimination step are staged in LDS.  Partial pivoting by |re| + |im|, first largest
// (zgetrf's izamax).  M and B are [n,n] or, with the bit of freq_mask set, [n,n,nw].
#define DENSE_MAX_LD 1536
__global__ void __launch_bounds__(256) k_solve_dense(int n, int nRhs, int nw, const double *__restrict__ w,
                                                     const double *__restrict__ M, const double *__restrict__ B,
                                                     const double *__restrict__ C, int freq_mask, int mdiv,
                                                     const double *__restrict__ Badd, const int *__restrict__ active,
                                                     const cplx *__restrict__ F, cplx *__restrict__ work, cplx *__restrict__ Xi,
                                                     cplx *__restrict__ Zout) {
    if (active && !active[blockIdx.y]) return;            // a system whose fixed point has ended keeps its last response
    __shared__ cplx rowk[DENSE_MAX_LD], colk[DENSE_MAX_LD];
    __shared__ double rbest[4];
    __shared__ int rrow[4];
    const int iw = blockIdx.x, ld = n + nRhs, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool mw = freq_mask & 1, bw = freq_mask & 2;
    {                                                     // blockIdx.y: the system of a batch; its matrices: set y / mdiv
        const size_t sy = blockIdx.y, sm = sy / (size_t)mdiv, nn = (size_t)n * n;
        M += sm * nn * (mw ? nw : 1);
        B += sm * nn * (bw ? nw : 1);
        C += sm * nn;
        if (Badd) Badd += sy * nn;
        F += sy * nRhs * n * nw;
        Xi += sy * nRhs * n * nw;
        if (Zout) Zout += sy * nn * nw;
        work += sy * nw * n * ld;
    }
    cplx *A = work + (size_t)iw * n * ld;
    const double ww = w[iw];
    for (int e = tid; e < n * ld; e += 256) {
        const int r = e / ld, c = e % ld;
        cplx v;
        if (c < n) {
            const size_t o = (size_t)r * n + c;
            const double m = mw ? M[o * nw + iw] : M[o];
            double b = bw ? B[o * nw + iw] : B[o];
            if (Badd) b += Badd[o];
            v = cplx{-(ww * ww) * m + C[o], ww * b};
            if (Zout) Zout[o * nw + iw] = v;
        } else {
            v = F[((size_t)(c - n) * n + r) * nw + iw];
        }
        A[e] = v;
    }
    __syncthreads();
    for (int k = 0; k < n; k++) {
        double best = -1.0;
        int p = n;
        for (int r = k + tid; r < n; r += 256) {
            const cplx a = A[(size_t)r * ld + k];
            const double v = fabs(a.re) + fabs(a.im);
            if (v > best) {
                best = v;
                p = r;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) argmax_step(best, p, off);
        if (lane == 0) {
            rbest[wv] = best;
            rrow[wv] = p;
        }
        __syncthreads();
        best = rbest[0];
        p = rrow[0];
        for (int i = 1; i < 4; i++)
            if (rbest[i] > best || (rbest[i] == best && rrow[i] < p)) {
                best = rbest[i];
                p = rrow[i];
            }
        if (p >= n) p = k;                                    // a column of NaNs: no row compares larger
        for (int c = k + tid; c < ld; c += 256) {            // swap rows k and p; the pivot row goes to LDS
            const cplx t = A[(size_t)k * ld + c], u = A[(size_t)p * ld + c];
            A[(size_t)p * ld + c] = t;
            A[(size_t)k * ld + c] = u;
            rowk[c] = u;
        }
        __syncthreads();
        const cplx pv = rowk[k];
        const double dd = pv.re * pv.re + pv.im * pv.im;
        const cplx inv = {pv.re / dd, -pv.im / dd};
        for (int r = k + 1 + tid; r < n; r += 256) {
            const cplx l = cmul(A[(size_t)r * ld + k], inv);
            A[(size_t)r * ld + k] = l;
            colk[r] = l;
        }
        __syncthreads();
        for (int r = k + 1 + wv; r < n; r += 4) {            // a wave per row, lanes along the row
            const cplx l = colk[r];
            cplx *row = A + (size_t)r * ld;
            for (int c = k + 1 + lane; c < ld; c += 64) row[c] = csub(row[c], cmul(l, rowk[c]));
        }
        __syncthreads();
    }
    for (int k = n - 1; k >= 0; k--) {                        // back substitution on the right-hand columns
        const cplx pv = A[(size_t)k * ld + k];
        const double dd = pv.re * pv.re + pv.im * pv.im;
        for (int j = tid; j < nRhs; j += 256) {
            const cplx sum = A[(size_t)k * ld + n + j];
            const cplx x = {(sum.re * pv.re + sum.im * pv.im) / dd, (sum.im * pv.re - sum.re * pv.im) / dd};
            A[(size_t)k * ld + n + j] = x;
            rowk[j] = x;
        }
        __syncthreads();
        for (int e = tid; e < k * nRhs; e += 256) {
            const int r = e / nRhs, j = e % nRhs;
            A[(size_t)r * ld + n + j] = csub(A[(size_t)r * ld + n + j], cmul(A[(size_t)r * ld + k], rowk[j]));
        }
        __syncthreads();
    }
    for (int e = tid; e < n * nRhs; e += 256) {
        const int r = e % n, j = e / n;
        Xi[((size_t)j * n + r) * nw + iw] = A[(size_t)r * ld + n + j];
    }
}

// The same solve for systems of up to 32 RB - nRhs .. unknowns with the whole augmented matrix in REGISTERS (the reference's
// flexible deck: 150): 32 x TJ threads (512 at TJ = 16: two waves per SIMD, 256 registers per lane), thread (ti, tj) owns the
// entries (ti + 32 a, tj + TJ b), a < RB, b < CB -- 50 complex numbers at RB = 5, CB = 10 -- so that a step of the
// elimination touches LDS for the pivot row and the column of multipliers only.  k_solve_dense above keeps the matrix in an
// L2-resident workspace and pays three round trips through L2 per step (150 steps: 1.34 ms per launch).  Gauss-Jordan with
// partial pivoting (izamax order: first largest |re| + |im| among the rows that have not been pivots yet) and IMPLICIT row
// interchanges -- a pivot row stays where it is, every register index is static (the step loop is unrolled over the column
// blocks).  Rows above the pivot are eliminated too (the lanes that own them would idle otherwise), so there is no back
// substitution: unknown k is the pivot row's right-hand side over its pivot.  A step:
//   pivot search by the half-wave that owns column k -- magnitude and row packed into ONE key (255 - row replaces the eight
//   lowest mantissa bits: a choice between candidates equal to 6e-14 is arbitrary anyway, ties go to the lower row as izamax
//   takes the first largest), so the argmax is a max over four row_shr DPP steps and one row_bcast (as __shfl_xor steps it was
//   fifteen dependent ds_bpermute round trips: 0.36 -> 0.29 ms);  BARRIER;  the thread row of the pivot row publishes that row
//   (only that row -- publishing all RB rows of the thread row costs 0.50 against 0.36 ms);  BARRIER;  the owners of column k
//   form the multipliers (1 / pivot by v_rcp_f64 + two Newton steps);  BARRIER;  update, a - l u in four FMAs.
// History of the form (scripts/ubench/dense_probe.hip, n = 150, 40 bins): 1024 threads with the last block row in LDS 0.59 ms;
// 512 threads all in registers 0.36; DPP argmax 0.29; rcp + four-FMA update 0.275; the owners' work (search, reciprocal,
// multipliers) inside their branch so that the other seven waves skip it 0.267.  Measured and not kept: every thread forming
// its own multipliers from the raw column (one barrier less: 0.37 against 0.36), the next column's pivot search by its owners
// right after they have updated it (look-ahead: no gain), the owners taking the pivot by v_readlane inside their half-wave and
// forming the multipliers while the pivot row is being published (two barriers per step: 0.269 against 0.273), 768 threads x 35
// entries (three waves per SIMD, 196 B scratch: 0.40) and 1024 x 25 (0.53).  Where a step's 1.8 us go, by elimination on the probe
// (variants that skip a part; wrong results, timing only): the update 0.10 of the 0.275 ms, the pivot search 0.04, the second
// and third barrier 0.012 each; the remaining 0.13 ms is the chain publish -> read pivot -> reciprocal -> multipliers -> read
// them back, with two dozen branches per step around the divergent parts.
// Badd: [n,n] per system added to B (the drag linearisation of the iteration; B itself may stay resident); mdiv: systems
// sy share the matrices of set sy / mdiv (the sea states of one unit).
// a - l u in four FMAs (a - (l u) written as a difference costs six instructions)
__device__ __forceinline__ cplx cfnma(cplx a, cplx l, cplx u) {
    return {fma(-l.re, u.re, fma(l.im, u.im, a.re)), fma(-l.re, u.im, fma(-l.im, u.re, a.im))};
}
template <int RB, int CB, int TJ>
__global__ void __launch_bounds__(32 * TJ) k_solve_dense_reg2(int n, int nRhs, int nw, const double *__restrict__ w,
                                                               const double *__restrict__ M, const double *__restrict__ B,
                                                               const double *__restrict__ C, int freq_mask, int mdiv,
                                                               const double *__restrict__ Badd, const int *__restrict__ active,
                                                               const cplx *__restrict__ F, cplx *__restrict__ Xi,
                                                               cplx *__restrict__ Zout) {
    if (active && !active[blockIdx.y]) return;            // a system whose fixed point has ended keeps its last response
    static_assert(RB * 32 <= 256 && RB <= 8, "k_solve_dense_reg2: the pivot key carries the row in eight bits");
    constexpr int NC = CB * TJ;                           // columns of the augmented matrix the grid covers
    __shared__ cplx stage[NC];                            // the pivot row of the step
    __shared__ cplx colk[RB * 32];                        // multipliers of the step, by row (0 for the pivot row)
    __shared__ cplx invp[RB * 32];                        // 1 / pivot of the step in which the row was the pivot row
    __shared__ int ord[RB * 32];                          // ... and that step
    __shared__ int psel;
    const int iw = blockIdx.x, ld = n + nRhs, tid = threadIdx.x, ti = tid & 31, tj = tid >> 5;
    const double ww = w[iw];
    const bool mw = freq_mask & 1, bw = freq_mask & 2;
    {
        const size_t sy = blockIdx.y, sm = sy / (size_t)mdiv, nn = (size_t)n * n;
        M += sm * nn * (mw ? nw : 1);
        B += sm * nn * (bw ? nw : 1);
        C += sm * nn;
        if (Badd) Badd += sy * nn;
        F += sy * nRhs * n * nw;
        Xi += sy * nRhs * n * nw;
        if (Zout) Zout += sy * nn * nw;
    }
    cplx A[RB][CB];
    unsigned done = 0;                                    // bit a: row ti + 32 a has been a pivot row (or does not exist)
#pragma unroll
    for (int a = 0; a < RB; a++) {
        const int r = ti + 32 * a;
        if (r >= n) done |= 1u << a;
#pragma unroll
        for (int b = 0; b < CB; b++) {
            const int c = tj + TJ * b;
            cplx v = {0.0, 0.0};
            if (r < n && c < n) {
                const size_t o = (size_t)r * n + c;
                const double m = mw ? M[o * nw + iw] : M[o];
                double bb = bw ? B[o * nw + iw] : B[o];
                if (Badd) bb += Badd[o];
                v = cplx{-(ww * ww) * m + C[o], ww * bb};
                if (Zout) Zout[o * nw + iw] = v;
            } else if (r < n && c < ld) {
                v = F[((size_t)(c - n) * n + r) * nw + iw];
            }
            A[a][b] = v;
        }
    }
#pragma unroll
    for (int kb = 0; kb < CB; kb++) {
        for (int kk = 0; kk < TJ; kk++) {
            const int k = kb * TJ + kk;
            if (k >= n) break;                            // (uniform)
            // ---- pivot search down column k: its owners are the 32 threads with tj == kk, one half-wave
            // magnitude and row as ONE key -- 255 - row replaces the eight lowest mantissa bits (a choice between candidates
            // equal to 6e-14 is arbitrary anyway; ties go to the lower row as izamax takes the first largest) -- so that
            // the argmax is a max: four row_shr steps inside the 16-lane rows, one row_bcast across the two rows of the
            // half-wave, all DPP (the shuffle form is fifteen dependent ds_bpermute round trips)
            // (everything the owners alone need sits INSIDE their branch: the other seven waves skip it -- the chain of a step
            // is issue-bound, two waves per SIMD each walking through whatever is not branched around)
            if (tj == kk) {
                double key = -1.0;
#pragma unroll
                for (int a = 0; a < RB; a++) {
                    double v = fabs(A[a][kb].re) + fabs(A[a][kb].im);
                    if (!(v <= 1.7e308)) v = 1.7e308;
                    const double kv = __hiloint2double(__double2hiint(v), (int)(((unsigned)__double2loint(v) & ~255u) | (unsigned)(255 - (ti + 32 * a))));
                    key = fmax(key, (done >> a & 1u) ? -1.0 : kv);
                }
                // the (magnitude, row) key compared in two 32-bit phases (as in k_solve_system_rows, raftx_hip.hip): v_max_f64
                // takes no DPP operand -- seven dependent VALU instructions per reduction step -- while v_max_u32 with a
                // zero-filling row_shr is one.  High words (+1: 0 = no candidate) first, then the low words of the lanes that
                // tie; the two 16-lane rows of the owners' half-wave meet in scalar registers.  Same pivots, bit for bit.
                const unsigned khi = key >= 0.0 ? (unsigned)__double2hiint(key) + 1u : 0u, klo = (unsigned)__double2loint(key);
#define ROW_UMAX_(x)                                                                                   \
                x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));   \
                x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));   \
                x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));   \
                x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));
                const bool upper = (kk & 1) != 0;         // tid = tj * 32 + ti: the owners (tj == kk) are one half of their wave
                unsigned mh = khi;
                ROW_UMAX_(mh)
                const unsigned hmax = upper ? max((unsigned)__builtin_amdgcn_readlane((int)mh, 47), (unsigned)__builtin_amdgcn_readlane((int)mh, 63))
                                            : max((unsigned)__builtin_amdgcn_readlane((int)mh, 15), (unsigned)__builtin_amdgcn_readlane((int)mh, 31));
                unsigned ml = (khi != 0u && khi == hmax) ? klo : 0u;
                ROW_UMAX_(ml)
#undef ROW_UMAX_
                const unsigned lmax = upper ? max((unsigned)__builtin_amdgcn_readlane((int)ml, 47), (unsigned)__builtin_amdgcn_readlane((int)ml, 63))
                                            : max((unsigned)__builtin_amdgcn_readlane((int)ml, 15), (unsigned)__builtin_amdgcn_readlane((int)ml, 31));
                if (ti == 31) psel = 255 - (int)(lmax & 255u);
            }
            __syncthreads();
            const int p = __builtin_amdgcn_readfirstlane(psel);   // (uniform: scalar compares below)
            const int ap = p >> 5;
            if (ti == (p & 31)) {                         // the thread row of the pivot row publishes it; the row is done
#pragma unroll
                for (int a = 0; a < RB; a++)
                    if (a == ap) {                        // (workgroup-uniform)
#pragma unroll
                        for (int b = 0; b < CB; b++)
                            if (b >= kb) stage[tj + TJ * b] = A[a][b];
                    }
                done |= 1u << ap;
            }
            __syncthreads();
            if (tj == kk) {
                const cplx pv = stage[k];
                const double dd = pv.re * pv.re + pv.im * pv.im;
                double d = __builtin_amdgcn_rcp(dd);       // 1 / |pivot|^2: v_rcp_f64 + two Newton steps (a zero pivot still ends in NaN)
                d = fma(fma(-dd, d, 1.0), d, d);
                d = fma(fma(-dd, d, 1.0), d, d);
                const cplx inv = {pv.re * d, -pv.im * d};
#pragma unroll
                for (int a = 0; a < RB; a++) {
                    const int r = ti + 32 * a;
                    colk[r] = (r == p) ? cplx{0.0, 0.0} : cmul(A[a][kb], inv);
                }
                if (ti == 0) {
                    invp[p] = inv;
                    ord[p] = k;
                }
            }
            __syncthreads();
            cplx l[RB];
#pragma unroll
            for (int a = 0; a < RB; a++) l[a] = colk[ti + 32 * a];
#pragma unroll
            for (int b = 0; b < CB; b++)
                if (b > kb || (b == kb && tj > kk)) {
                    const cplx rkb = stage[tj + TJ * b];
#pragma unroll
                    for (int a = 0; a < RB; a++) A[a][b] = cfnma(A[a][b], l[a], rkb);
                }
        }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < RB; a++) {
        const int r = ti + 32 * a;
#pragma unroll
        for (int b = 0; b < CB; b++) {
            const int c = tj + TJ * b;
            if (r < n && c >= n && c < ld) Xi[((size_t)(c - n) * n + ord[r]) * nw + iw] = cmul(A[a][b], invp[r]);
        }
    }
}
