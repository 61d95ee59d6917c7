// Where do the two waves of a 128-thread workgroup land?  (HW_ID: wave slot, SIMD, CU, SE, XCC) -- for workgroups shaped like
// the fused kernel's (256 VGPRs -> two waves per SIMD, ~40 KB LDS -> four workgroups per CU).
// build: hipcc --offload-arch=gfx950 -O3 -o hwid_probe hwid_probe.hip ; run: ./hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void __launch_bounds__(128, 2) k(unsigned *out, int spin) {
    extern __shared__ double sm[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // burn registers and time so that several rounds of workgroups overlap
    double acc[100];
    for (int i = 0; i < 100; i++) acc[i] = threadIdx.x * 0.5 + i;
    for (int it = 0; it < spin; it++)
        for (int i = 0; i < 100; i++) acc[i] = fma(acc[i], 1.0000001, 0.5);
    double s = 0;
    for (int i = 0; i < 100; i++) s += acc[i];
    sm[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc | (sm[threadIdx.x] == 1.234 ? 1u << 31 : 0);
    }
}
int main() {
    const int nb = 10000;
    unsigned *d;
    hipMalloc(&d, nb * 4 * sizeof(unsigned));
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 40000);
    hipLaunchKernelGGL(k, dim3(nb), dim3(128), 40000, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 4);
    hipMemcpy(h.data(), d, nb * 4 * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<std::pair<int, int>, int> simd_pairs, slot_pairs;
    int maxslot = 0;
    for (int b = 0; b < nb; b++) {
        unsigned a = h[b * 4], c = h[b * 4 + 2];
        int slot0 = a & 15, simd0 = (a >> 4) & 3, slot1 = c & 15, simd1 = (c >> 4) & 3;
        simd_pairs[{simd0, simd1}]++;
        slot_pairs[{slot0, slot1}]++;
        if (slot0 > maxslot) maxslot = slot0;
        if (slot1 > maxslot) maxslot = slot1;
    }
    printf("max wave slot id %d\n", maxslot);
    for (auto &kv : simd_pairs) printf("SIMD (w0, w1) = (%d, %d): %d workgroups\n", kv.first.first, kv.first.second, kv.second);
    for (auto &kv : slot_pairs) printf("wave slot (w0, w1) = (%d, %d): %d workgroups\n", kv.first.first, kv.first.second, kv.second);
    printf("first workgroups: ");
    for (int b = 0; b < 6; b++) printf("[hw %08x %08x xcc %x] ", h[b * 4], h[b * 4 + 2], h[b * 4 + 1] & 0xff);
    printf("\n");
    return 0;
}
