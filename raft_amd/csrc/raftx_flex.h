// raftx_flex.h -- the fixed point of units with MORE than 6 reduced DOFs (flexible members) on the device: what
// raft_model.py:1052-1155 does per unit and load case with the nDOF x nDOF matrices of raft_fowt.py's T reduction, for every
// (unit, sea state) of a batch at once.  The strip sweeps are the ones of the rigid path, node by node (every structural node
// with wet strips is a "design" about its own position, raft_member.py:1969-1976, 2046-2056: k_linearize); this file holds what
// sits between them and the dense solves (raftx_dense.h):
//   k_flex_node_motion   Xi_node = T_node XiLast                              (raft_fowt.py:1912-1929)
//   k_flex_w, k_flex_gemm_B   B_drag = sum_nodes T_node^T B_node T_node        (raft_fowt.py:1931-1934) -- a plain GEMM
//                        (n x 6 N_node) (6 N_node x n) per (unit, sea state): v_mfma_f64_16x16x4_f64 tiles
//   k_flex_project_F     F_wave = F_lin + sum_nodes T_node^T F_node           (raft_fowt.py:1936, raft_model.py:1048, 1212)
//   k_flex_converge      convergence test and relaxation per (unit, sea state) (raft_model.py:1098-1133)
// A pair that has converged is FROZEN: its response, B_drag and drag excitation stay those of its last iteration, exactly as
// if it had been solved alone (the launches still cover it; their results for it are dropped).
// Included by raftx_hip.hip after raftx_kernels.h (cplx).
#pragma once

// Xi_node[node, c, a, iw] = sum_j Tn[node, a, j] XiLast[(u, c), j, iw]; grid (nNode * nCase), lanes over (a, iw)
__global__ void __launch_bounds__(256) k_flex_node_motion(int nCase, int n, int nw, const int *__restrict__ nodeUnit,
                                                          const double *__restrict__ Tn, const cplx *__restrict__ XiLast,
                                                          cplx *__restrict__ XiN) {
    const int node = blockIdx.x / nCase, c = blockIdx.x % nCase, u = nodeUnit[node];
    const cplx *X = XiLast + ((size_t)u * nCase + c) * n * nw;
    const double *T = Tn + (size_t)node * 6 * n;
    cplx *out = XiN + ((size_t)node * nCase + c) * 6 * nw;
    for (int e = threadIdx.x; e < 6 * nw; e += blockDim.x) {
        const int a = e / nw, iw = e % nw;
        double re = 0.0, im = 0.0;
        for (int j = 0; j < n; j++) {
            const double t = T[a * n + j];
            const cplx x = X[(size_t)j * nw + iw];
            re = fma(t, x.re, re);
            im = fma(t, x.im, im);
        }
        out[e] = cplx{re, im};
    }
}

// W[c][node * 6 + a][j] = sum_b B_node[node, c, a, b] Tn[node, b, j]; grid (nNode * nCase), lanes over (a, j)
__global__ void __launch_bounds__(256) k_flex_w(int nNode, int nCase, int n, const double *__restrict__ Tn,
                                                const double *__restrict__ Bn, double *__restrict__ W) {
    const int node = blockIdx.x / nCase, c = blockIdx.x % nCase;
    __shared__ double b[36];
    if (threadIdx.x < 36) b[threadIdx.x] = Bn[((size_t)node * nCase + c) * 36 + threadIdx.x];
    __syncthreads();
    const double *T = Tn + (size_t)node * 6 * n;
    double *out = W + ((size_t)c * nNode + node) * 6 * n;
    for (int e = threadIdx.x; e < 6 * n; e += blockDim.x) {
        const int a = e / n, j = e % n;
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 6; q++) s = fma(b[a * 6 + q], T[q * n + j], s);
        out[e] = s;
    }
}

// B_drag[s] (n x n) = T2_u^T W[c]_u over the unit's 6 N_u rows: one wavefront per 16 x 16 tile, v_mfma_f64_16x16x4_f64
// (A fragment: lane l holds A[i = l & 15][k = l >> 4]; B fragment: B[k = l >> 4][j = l & 15]; the four results of lane l are
// D[i = (l >> 4) + 4 r][j = l & 15], r < 4).  grid (tiles, nSys) with four tiles per 256-thread workgroup.
typedef double flex_v4d __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_flex_gemm_B(int nNode, int nCase, int n, const int64_t *__restrict__ nodeOff,
                                                     const double *__restrict__ Tn, const double *__restrict__ W,
                                                     const int *__restrict__ active, double *__restrict__ Bd) {
    const int s = blockIdx.y;
    if (active && !active[s]) return;
    const int u = s / nCase, c = s % nCase, nt = (n + 15) >> 4;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (tile >= nt * nt) return;
    const int i0 = (tile / nt) * 16, j0 = (tile % nt) * 16;
    const int64_t k0 = nodeOff[u] * 6, k1 = nodeOff[u + 1] * 6;
    const double *A = Tn, *B = W + (size_t)c * nNode * 6 * n;           // both [row k][column], n columns
    const int li = l & 15, lk = l >> 4;
    const bool ai = i0 + li < n, bj = j0 + li < n;
    flex_v4d acc = {0.0, 0.0, 0.0, 0.0};
    for (int64_t k = k0; k < k1; k += 4) {
        const int64_t kr = k + lk;
        const double a = (ai && kr < k1) ? A[kr * n + i0 + li] : 0.0;
        const double b = (bj && kr < k1) ? B[kr * n + j0 + li] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    double *out = Bd + (size_t)s * n * n;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int i = i0 + lk + 4 * r, j = j0 + li;
        if (i < n && j < n) out[(size_t)i * n + j] = acc[r];
    }
}

// Fw[s, h, i, iw] = F_lin[s, h, i, iw] + sum_k T2[k, i] F_node[node(k), c, h, a(k), iw]  (and the sum alone into Fd, and heading 0
// into rhs1 [nSys, n, nw], the right-hand side of the iteration); grid (ceil(n nw / 256), nSys * nHead)
__global__ void __launch_bounds__(256) k_flex_project_F(int nCase, int nHead, int n, int nw, const int64_t *__restrict__ nodeOff,
                                                        const double *__restrict__ Tn, const cplx *__restrict__ Fn,
                                                        const cplx *__restrict__ Flin, const int *__restrict__ active,
                                                        cplx *__restrict__ Fw, cplx *__restrict__ Fd, cplx *__restrict__ rhs1) {
    const int s = blockIdx.y / nHead, h = blockIdx.y % nHead;
    if (active && !active[s]) return;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * nw) return;
    const int i = e / nw, iw = e % nw, u = s / nCase, c = s % nCase;
    double re = 0.0, im = 0.0;
    for (int64_t node = nodeOff[u]; node < nodeOff[u + 1]; node++) {
        const double *T = Tn + (size_t)node * 6 * n + i;
        const cplx *F = Fn + ((((size_t)node * nCase + c) * nHead + h) * 6) * nw + iw;
#pragma unroll
        for (int a = 0; a < 6; a++) {
            const double t = T[(size_t)a * n];
            const cplx f = F[(size_t)a * nw];
            re = fma(t, f.re, re);
            im = fma(t, f.im, im);
        }
    }
    const size_t o = ((size_t)s * nHead + h) * n * nw + e;
    const cplx fl = Flin[o];
    if (Fd) Fd[o] = cplx{re, im};
    const cplx sum = {fl.re + re, fl.im + im};
    Fw[o] = sum;
    if (h == 0) rhs1[(size_t)s * n * nw + e] = sum;
}

// Convergence test and relaxation of every (unit, sea state) still iterating (raft_model.py:1098-1133): its response of this
// iteration is kept; NaN -> flagged 2 and dropped; all entries within tol -> flagged 1 (converged) and frozen; otherwise
// XiLast <- 0.2 XiLast + 0.8 Xi.  The number of pairs still iterating is added up in *nActive; the iteration number comes from
// device memory (*iterp, advanced by k_flex_tick) so that the launches of an iteration are the same every time: one hipGraph.
__global__ void __launch_bounds__(256) k_flex_converge(int n, int nw, double tol, const int *__restrict__ iterp, const cplx *__restrict__ Xnew,
                                                       cplx *__restrict__ Xi, cplx *__restrict__ XiLast, int *__restrict__ active,
                                                       int *__restrict__ niter, int *__restrict__ flags, int *__restrict__ nActive) {
    const int s = blockIdx.x;
    if (!active[s]) return;
    __shared__ int bad[2];
    if (threadIdx.x < 2) bad[threadIdx.x] = 0;
    __syncthreads();
    const size_t o = (size_t)s * n * nw;
    int nan_ = 0, far_ = 0;
    for (int e = threadIdx.x; e < n * nw; e += blockDim.x) {
        const cplx x = Xnew[o + e], xl = XiLast[o + e];
        Xi[o + e] = x;
        if (x.re != x.re || x.im != x.im) nan_ = 1;
        const double dr = x.re - xl.re, di = x.im - xl.im;
        const double check = sqrt(dr * dr + di * di) / (sqrt(x.re * x.re + x.im * x.im) + tol);          // :1103
        if (!(check < tol)) far_ = 1;
    }
    if (nan_) atomicOr(&bad[0], 1);
    if (far_) atomicOr(&bad[1], 1);
    __syncthreads();
    const bool isnan_ = bad[0] != 0, conv = !isnan_ && bad[1] == 0;
    if (!isnan_ && !conv)
        for (int e = threadIdx.x; e < n * nw; e += blockDim.x) {
            const cplx x = Xnew[o + e], xl = XiLast[o + e];
            XiLast[o + e] = cplx{0.2 * xl.re + 0.8 * x.re, 0.2 * xl.im + 0.8 * x.im};                    // :1133
        }
    if (threadIdx.x == 0) {
        niter[s] = *iterp + 1;
        if (isnan_) flags[s] |= 2;
        if (conv) flags[s] |= 1;
        if (isnan_ || conv) active[s] = 0;
        else atomicAdd(nActive, 1);
    }
}

__global__ void k_flex_tick(int *iterp) { *iterp += 1; }

__global__ void k_flex_fill(size_t n, cplx v, cplx *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}
