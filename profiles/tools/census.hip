// Workgroup placement census: which XCD / SE / CU does each 256-thread block land on when the
// grid is k blocks per CU?  (diagnostic only; not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void census(unsigned *out, int spin) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hwid; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
    for (int per_cu : {3, 4, 5, 6}) {
        int nb = per_cu * 256;
        unsigned *d; hipMalloc(&d, nb * 8);
        hipLaunchKernelGGL(census, dim3(nb), dim3(256), 0, 0, d, 200000);
        hipDeviceSynchronize();
        std::vector<unsigned> h(2 * nb);
        hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
        std::map<unsigned, int> cnt;
        for (int b = 0; b < nb; ++b) {
            unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            cnt[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
        }
        std::map<int, int> hist;
        for (auto &kv : cnt) hist[kv.second]++;
        printf("blocks/CU requested %d: %zu distinct CUs used; blocks-per-CU histogram:", per_cu, cnt.size());
        for (auto &kv : hist) printf("  %d blocks x %d CUs", kv.first, kv.second);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
