/* TEST INFRASTRUCTURE ONLY.  Drives the CPU oracle (oracle/raftx_oracle.c, compiled into this executable with
 * -fsanitize=address,undefined) through the C-ABI on a problem dumped by tests/test_sanitizers.py:
 *   upload_designs -> upload_cases -> excitation -> linearize -> solve_dynamics (all optional outputs) -> motion_stats,
 * so that out-of-bounds accesses, leaks of the per-call scratch and undefined behaviour in the restatement show up in the
 * CPU suite.  Input: a flat binary file, see the reader below.  Prints "OK" and a checksum. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/raftx.h"

static void *rd(FILE *f, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (bytes && fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int64_t hdr[6];                                   /* nDesign, nStrips, nw, nCase, nHead, nIter */
    if (fread(hdr, sizeof(int64_t), 6, f) != 6) return 2;
    const int nD = (int)hdr[0], nS = (int)hdr[1], nw = (int)hdr[2], nC = (int)hdr[3], nH = (int)hdr[4], nIter = (int)hdr[5];
    int64_t *off = rd(f, sizeof(int64_t) * (nD + 1));
    double *strips = rd(f, sizeof(double) * (size_t)nS * RAFTX_NFIELD);
    double *M0 = rd(f, sizeof(double) * nD * 36), *B0 = rd(f, sizeof(double) * nD * 36), *C0 = rd(f, sizeof(double) * nD * 36);
    double *w = rd(f, sizeof(double) * nw), *k = rd(f, sizeof(double) * nw);
    double scal[2];                                   /* depth, XiStart */
    if (fread(scal, sizeof(double), 2, f) != 2) return 2;
    double *zeta = rd(f, sizeof(double) * (size_t)nC * nH * nw), *beta = rd(f, sizeof(double) * nC * nH);
    fclose(f);
    raftx_ctx *c = NULL;
    if (raftx_ctx_create(0, &c)) return 3;
#define CK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, raftx_last_error(c)); return 4; } } while (0)
    CK(raftx_upload_designs(c, nD, off, strips, RAFTX_NFIELD, M0, B0, C0, nw, NULL, NULL, NULL));
    CK(raftx_upload_cases(c, nC, nH, nw, w, k, scal[0], 1025.0, 9.81, zeta, beta));
    const size_t npair = (size_t)nD * nC, nx = npair * nH * 6 * nw;
    raftx_c128 *F = malloc(sizeof(raftx_c128) * nx), *Xi = malloc(sizeof(raftx_c128) * nx), *Fw = malloc(sizeof(raftx_c128) * nx);
    raftx_c128 *Z = malloc(sizeof(raftx_c128) * npair * 36 * nw), *Xl = calloc(npair * 6 * nw, sizeof(raftx_c128));
    double *B = malloc(sizeof(double) * npair * 36), *sd = malloc(sizeof(double) * npair * 6), *psd = malloc(sizeof(double) * npair * 6 * nw);
    int32_t *ni = malloc(sizeof(int32_t) * npair), *fl = malloc(sizeof(int32_t) * npair);
    CK(raftx_excitation(c, F));
    CK(raftx_linearize(c, Xl, B, Fw));
    CK(raftx_solve_dynamics(c, nIter, 0.01, scal[1], NULL, Xi, ni, fl, B, Fw, Z));
    CK(raftx_motion_stats(c, nw > 1 ? w[1] - w[0] : w[0], sd, psd));
    double cs = 0.0;
    for (size_t i = 0; i < nx; i++) cs += Xi[i].re + Xi[i].im;
    for (size_t i = 0; i < npair; i++) cs += ni[i] + sd[i * 6];
    raftx_ctx_destroy(c);
    free(off); free(strips); free(M0); free(B0); free(C0); free(w); free(k); free(zeta); free(beta);
    free(F); free(Xi); free(Fw); free(Z); free(Xl); free(B); free(sd); free(psd); free(ni); free(fl);
    printf("OK %.17g\n", cs);
    return 0;
}
