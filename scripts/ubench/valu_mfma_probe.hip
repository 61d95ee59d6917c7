// valu_mfma_probe.hip -- what the fp64 pipes of gfx950 (MI355X) can do, measured:
//   fma64        v_fma_f64 chains, all 64 lanes                    -> the fp64 vector roof
//   fma16, fma8  the same under EXEC = low 16 / low 8 lanes        -> does the SIMD skip idle quarter-waves?  (the
//                200-bin shape of k_solve_dynamics runs one of its four bin slots with 8 active lanes)
//   mfma16       v_mfma_f64_16x16x4_f64 chains                     -> the fp64 matrix roof
//   mfma4        v_mfma_f64_4x4x4_4b_f64 chains
//   mix          mfma16 and v_fma_f64 interleaved in one wave      -> do the two pipes run side by side?
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_mfma_probe valu_mfma_probe.hip ; run on the GPU box, prints JSON lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
#define NACC 8

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(double *out, int iters, double seed, long long *clk) {
    const int lane = threadIdx.x & 63;
    const long long c0 = clock64(), w0 = wall_clock64();     // shader cycles / 100 MHz reference ticks
    double a[NACC], x = seed + lane * 1e-9, y = 1.0 - 1e-9 * lane;
    v4d m[NACC];
    double m1[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) {
        a[i] = 1.0 + i;
        m[i] = v4d{0.0, 0.0, 0.0, 0.0};
        m1[i] = 0.0;
    }
    const bool active = MODE == 1 ? lane < 16 : (MODE == 2 ? lane < 8 : true);
    if (active) {
#pragma unroll 8
        for (int it = 0; it < iters; it++) {                 // unrolled: the loop branch is not what is measured
            if (MODE <= 2 || MODE == 5) {
#pragma unroll
                for (int i = 0; i < NACC; i++) a[i] = __builtin_fma(a[i], y, x);
            }
            if (MODE == 3 || MODE == 5) {
#pragma unroll
                for (int i = 0; i < NACC; i++) m[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, m[i], 0, 0, 0);
            }
            if (MODE == 4) {
#pragma unroll
                for (int i = 0; i < NACC; i++) m1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, m1[i], 0, 0, 0);
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += a[i] + m[i][0] + m[i][1] + m[i][2] + m[i][3] + m1[i];
    if (s == 12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;      // keep the work alive
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = clock64() - c0;
        clk[1] = wall_clock64() - w0;
    }
}

template <int MODE>
static double run(const char *name, int blocks, int iters, double flop_per_wave_iter, double *dout) {
    static long long *dclk = nullptr;
    if (!dclk) (void)hipMalloc(&dclk, 2 * sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, dout, iters / 10, 0.5, (long long *)nullptr);     // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, dout, iters, 0.5, dclk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    long long hclk[2] = {0, 0};
    (void)hipMemcpy(hclk, dclk, sizeof(hclk), hipMemcpyDeviceToHost);
    const double shader_mhz = hclk[1] > 0 ? (double)hclk[0] / ((double)hclk[1] / 100.0) : 0.0;     // wall_clock64 ticks at 100 MHz
    const double waves = (double)blocks * 4;
    const double tf = flop_per_wave_iter * waves * iters / (ms * 1e-3) / 1e12;
    printf("{\"probe\": \"%s\", \"ms\": %.4f, \"waves\": %.0f, \"iters\": %d, \"useful_TFLOPs\": %.2f, \"shader_clock_MHz_during\": %.0f, "
           "\"cycles_per_wave_instr_per_simd\": %.2f}\n", name, ms, waves, iters, tf, shader_mhz,
           shader_mhz * 1e3 * ms / ((double)iters * NACC * 2 * (MODE == 5 ? 2 : 1)));
    fflush(stdout);
    return ms;
}

int main(int argc, char **argv) {
    int dev = 0;
    hipSetDevice(dev);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, dev);
    const int cus = p.multiProcessorCount;
    const int blocks = cus * 2;                      // 8 waves per CU = 2 per SIMD, as k_solve_dynamics runs
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    double *dout = nullptr;
    hipMalloc(&dout, sizeof(double) * (size_t)blocks * 256);
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_MHz\": %d, \"waves_per_simd\": 2}\n", p.gcnArchName, cus, p.clockRate / 1000);
    // useful flops per wave and iteration: NACC fmas x active lanes x 2;  mfma 16x16x4: 16*16*4*2;  4x4x4 4 blocks: 4*4*4*4*2
    const double t64 = run<0>("fma64", blocks, iters, NACC * 64 * 2.0, dout);
    const double t16 = run<1>("fma16_exec_low16", blocks, iters, NACC * 16 * 2.0, dout);
    const double t8 = run<2>("fma8_exec_low8", blocks, iters, NACC * 8 * 2.0, dout);
    const double tm = run<3>("mfma_f64_16x16x4", blocks, iters, NACC * 2048.0, dout);
    const double t4 = run<4>("mfma_f64_4x4x4_4b", blocks, iters, NACC * 512.0, dout);
    const double tx = run<5>("mix_fma64+mfma16", blocks, iters, NACC * (128.0 + 2048.0), dout);
    printf("{\"summary\": {\"exec16_over_exec64_time\": %.3f, \"exec8_over_exec64_time\": %.3f, \"mix_over_sum\": %.3f, "
           "\"mix_over_max\": %.3f, \"mfma4_over_mfma16_time\": %.3f}}\n",
           t16 / t64, t8 / t64, tx / (t64 + tm), tx / (t64 > tm ? t64 : tm), t4 / tm);
    hipFree(dout);
    return 0;
}
