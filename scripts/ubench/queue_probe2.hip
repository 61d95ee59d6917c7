// Which second streams get a small kernel onto the chip WHILE a chip-filling kernel runs on stream A?  Creates 18
// candidate streams up front (plain / high / low priority in turn) and tests each against the same long kernel.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_long(double *out, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.0000001;
    for (int i = 0; i < iters; i++) a = a * b + 1e-9;
    if (a == 12345.678) out[0] = a;
}
__global__ void k_small(int *flag) { if (threadIdx.x == 0 && blockIdx.x == 0) *flag = 1; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int nA = argc > 1 ? atoi(argv[1]) : 1;      // streams created before A
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    double *out; int *flag;
    CK(hipMalloc(&out, 8));
    CK(hipHostMalloc(&flag, sizeof(int), hipHostMallocDefault));
    std::vector<hipStream_t> pre(nA);
    for (auto &s : pre) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipStream_t sA = pre.back();
    const int prios[3] = {0, greatest, least};
    const char *names[3] = {"plain", "high", "low"};
    std::vector<hipStream_t> cand(18);
    for (int i = 0; i < 18; i++) CK(hipStreamCreateWithPriority(&cand[i], hipStreamNonBlocking, prios[i % 3]));
    for (int i = 0; i < 18; i++) {
        double tS = -1, tL = -1;
        for (int rep = 0; rep < 2; rep++) {
            *flag = 0; tS = tL = -1;
            hipEvent_t eA; CK(hipEventCreate(&eA));
            const double t0 = now_ms();
            hipLaunchKernelGGL(k_long, dim3(20480), dim3(128), 0, sA, out, 12000);       // ~5 rounds, ~3 ms
            CK(hipEventRecord(eA, sA));
            hipLaunchKernelGGL(k_small, dim3(40), dim3(256), 0, cand[i], flag);
            while (tS < 0 || tL < 0) {
                if (tS < 0 && *(volatile int *)flag) tS = now_ms() - t0;
                if (tL < 0 && hipEventQuery(eA) == hipSuccess) tL = now_ms() - t0;
            }
            CK(hipEventDestroy(eA)); CK(hipDeviceSynchronize());
        }
        printf("A = stream #%d, candidate %2d (%-5s): small done %.3f ms, long %.3f ms %s\n", nA, i, names[i % 3], tS, tL, tS < 0.3 ? "<== immediate" : "");
    }
    return 0;
}
