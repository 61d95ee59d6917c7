// Stand-alone timing harness of the dense impedance solves (raft_amd/csrc/raftx_dense.h): random 150 x 150 systems, one
// right-hand side, nw bins x nSys systems; HIP-event time per launch and the worst residual |Z x - f| / |f|.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DDENSE_VARIANT=..] -o dense_probe scripts/ubench/dense_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/raftx.h"
#include "../../raft_amd/csrc/raftx_kernels.h"
#include "../../raft_amd/csrc/raftx_dense.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 150, nw = argc > 2 ? atoi(argv[2]) : 40, nSys = argc > 3 ? atoi(argv[3]) : 1;
    const int nRhs = 1;
    std::vector<double> w(nw), M((size_t)nSys * n * n), B((size_t)nSys * n * n), C((size_t)nSys * n * n);
    std::vector<cplx> F((size_t)nSys * n * nw), X((size_t)nSys * n * nw);
    srand(3);
    auto rnd = []() { return rand() / (double)RAND_MAX - 0.5; };
    for (int i = 0; i < nw; i++) w[i] = 0.1 + 1.9 * i / (nw > 1 ? nw - 1 : 1);
    for (int s = 0; s < nSys; s++)
        for (int r = 0; r < n; r++)
            for (int c = 0; c < n; c++) {
                const size_t o = ((size_t)s * n + r) * n + c;
                M[o] = rnd() + (r == c ? 3.0 : 0.0);       // not diagonally dominant: pivoting has work to do
                B[o] = rnd();
                C[o] = rnd() + (r == c ? 1.0 : 0.0);
            }
    for (auto &f : F) f = cplx{rnd(), rnd()};
    double *dw, *dM, *dB, *dC;
    cplx *dF, *dX;
    CK(hipMalloc(&dw, nw * 8)); CK(hipMalloc(&dM, M.size() * 8)); CK(hipMalloc(&dB, B.size() * 8)); CK(hipMalloc(&dC, C.size() * 8));
    CK(hipMalloc(&dF, F.size() * 16)); CK(hipMalloc(&dX, X.size() * 16));
    CK(hipMemcpy(dw, w.data(), nw * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dF, F.data(), F.size() * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid(nw, nSys);
    const char *names[3] = {"k_solve_dense (L2 workspace, 256 threads)", "k_solve_dense_reg2<5,10,16>", "k_solve_dense_reg2<3,6,16>"};
    cplx *dA = nullptr;
    CK(hipMalloc(&dA, (size_t)nSys * nw * n * (n + nRhs) * 16));
    auto run = [&](int which) {
        if (which == 0)
            hipLaunchKernelGGL(k_solve_dense, grid, dim3(256), 0, 0, n, nRhs, nw, dw, dM, dB, dC, 0, 1, (const double *)nullptr, (const int *)nullptr, dF, dA, dX, (cplx *)nullptr);
        else if (which == 1)
            hipLaunchKernelGGL((k_solve_dense_reg2<5, 10, 16>), grid, dim3(512), 0, 0, n, nRhs, nw, dw, dM, dB, dC, 0, 1, (const double *)nullptr, (const int *)nullptr, dF, dX,
                               (cplx *)nullptr);
        else
            hipLaunchKernelGGL((k_solve_dense_reg2<3, 6, 16>), grid, dim3(512), 0, 0, n, nRhs, nw, dw, dM, dB, dC, 0, 1, (const double *)nullptr, (const int *)nullptr, dF, dX,
                               (cplx *)nullptr);
    };
    for (int which = 0; which < 3; which++) {
        if ((which == 1 && n + nRhs > 160) || (which == 2 && n + nRhs > 96)) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            CK(hipMemset(dX, 0, X.size() * 16));
            CK(hipEventRecord(e0, 0));
            run(which);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2 && ms < best) best = ms;
        }
        CK(hipMemcpy(X.data(), dX, X.size() * 16, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int s = 0; s < nSys; s += (nSys > 4 ? nSys / 4 : 1))
            for (int i = 0; i < nw; i += (nw > 8 ? nw / 8 : 1)) {
                double fmax = 0, rmax = 0;
                for (int r = 0; r < n; r++) {
                    double re = 0, im = 0;
                    for (int c = 0; c < n; c++) {
                        const size_t o = ((size_t)s * n + r) * n + c;
                        const double zr = -w[i] * w[i] * M[o] + C[o], zi = w[i] * B[o];
                        const cplx x = X[((size_t)s * n + c) * nw + i];
                        re += zr * x.re - zi * x.im;
                        im += zr * x.im + zi * x.re;
                    }
                    const cplx f = F[((size_t)s * n + r) * nw + i];
                    rmax = fmax > 0 ? rmax : rmax;
                    const double e = std::hypot(re - f.re, im - f.im);
                    if (e > rmax) rmax = e;
                    if (std::hypot(f.re, f.im) > fmax) fmax = std::hypot(f.re, f.im);
                }
                if (rmax / fmax > worst) worst = rmax / fmax;
            }
        printf("%s  n=%d nw=%d nSys=%d  %.3f ms per launch  residual %.1e\n", names[which],
               n, nw, nSys, best, worst);
    }
    return 0;
}
