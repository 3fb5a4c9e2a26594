// valu_probe.hip -- issue cost of the float64 vector instructions the forward kernel is made of (round 4).
//   hipcc --offload-arch=gfx950 -O3 -o profiles/tools/valu_probe profiles/tools/valu_probe.hip && profiles/tools/valu_probe
// One workgroup of 64 x W threads per CU (W waves per SIMD = W / 4 ... see `waves`), each wave runs `iters` x 32 instructions of ONE
// kind, either as one dependent chain (dep) or as 8 independent chains (ind); cycles by s_memtime around the loop of wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)
// KIND: 0 v_fma_f64, 1 v_add_f64, 2 v_floor_f64, 3 v_cvt_u32_f64, 4 v_fma_f32, 5 v_lshl_add_u32, 6 v_mul_f64
template <int KIND, int DEP>
__global__ void k_probe(int iters, double *out, unsigned long long *clk) {
    double a0 = threadIdx.x * 1e-9 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 0.999999, c = 1e-7;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    unsigned long long c0, c1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0 && DEP) { REP32(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));) }
        if (KIND == 0 && !DEP) {
            REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        }
        if (KIND == 1 && DEP) { REP32(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a0) : "v"(c));) }
        if (KIND == 1 && !DEP) {
            REP8(asm volatile("v_add_f64 %0, %0, %8\n\tv_add_f64 %1, %1, %8\n\tv_add_f64 %2, %2, %8\n\tv_add_f64 %3, %3, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        }
        if (KIND == 2 && DEP) { REP32(asm volatile("v_floor_f64 %0, %0" : "+v"(a0));) }
        if (KIND == 2 && !DEP) {
            REP8(asm volatile("v_floor_f64 %0, %0\n\tv_floor_f64 %1, %1\n\tv_floor_f64 %2, %2\n\tv_floor_f64 %3, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        }
        if (KIND == 3) { REP8(asm volatile("v_cvt_u32_f64 %0, %4\n\tv_cvt_u32_f64 %1, %5\n\tv_cvt_u32_f64 %2, %6\n\tv_cvt_u32_f64 %3, %7"
                                            : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        if (KIND == 4 && DEP) { REP32(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"((float)m), "v"((float)c));) }
        if (KIND == 4 && !DEP) {
            REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"((float)m), "v"((float)c));)
        }
        if (KIND == 5) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 3, %4\n\tv_lshl_add_u32 %1, %1, 3, %4\n\tv_lshl_add_u32 %2, %2, 3, %4\n\tv_lshl_add_u32 %3, %3, 3, %4"
                                            : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u7));) }
        if (KIND == 6 && DEP) { REP32(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(m));) }
        if (KIND == 6 && !DEP) {
            REP8(asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));)
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) clk[2 * (threadIdx.x >> 6)] = c0, clk[2 * (threadIdx.x >> 6) + 1] = c1;
}
template <int KIND, int DEP>
double run(int cus, int waves_per_simd, int iters, double *out, unsigned long long *clk) {
    hipLaunchKernelGGL((k_probe<KIND, DEP>), dim3(cus), dim3(256 * waves_per_simd), 0, 0, iters, out, clk);
    hipLaunchKernelGGL((k_probe<KIND, DEP>), dim3(cus), dim3(256 * waves_per_simd), 0, 0, iters, out, clk);
    CK(hipDeviceSynchronize());
    unsigned long long h[32], lo = ~0ull, hi = 0;
    CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    for (int w = 0; w < 4 * waves_per_simd; ++w) lo = h[2 * w] < lo ? h[2 * w] : lo, hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi;
    // SIMD cycles per wave-instruction: the span of the workgroup's waves over the instructions ONE SIMD executed in it
    return (double)(hi - lo) / ((double)iters * 32.0 * waves_per_simd);
}
int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double *out;
    unsigned long long *clk;
    CK(hipMalloc((void **)&out, sizeof(double) * cus * 1024));
    CK(hipMalloc((void **)&clk, 512));
    const int iters = 2000;
    printf("{\"unit\": \"shader cycles per wave-instruction, as seen by one wave; w = waves per SIMD\"");
#define ROW(NAME, K)                                                                                                          \
    printf(", \"%s\": {\"dep_w1\": %.2f, \"ind_w1\": %.2f, \"dep_w2\": %.2f, \"ind_w2\": %.2f, \"dep_w4\": %.2f, \"ind_w4\": %.2f}", NAME,      \
           run<K, 1>(cus, 1, iters, out, clk), run<K, 0>(cus, 1, iters, out, clk), run<K, 1>(cus, 2, iters, out, clk),          \
           run<K, 0>(cus, 2, iters, out, clk), run<K, 1>(cus, 4, iters, out, clk), run<K, 0>(cus, 4, iters, out, clk));
    ROW("v_fma_f64", 0)
    ROW("v_add_f64", 1)
    ROW("v_floor_f64", 2)
    ROW("v_mul_f64", 6)
    ROW("v_fma_f32", 4)
    printf(", \"v_cvt_u32_f64\": {\"ind_w1\": %.2f, \"ind_w4\": %.2f}", run<3, 0>(cus, 1, iters, out, clk), run<3, 0>(cus, 4, iters, out, clk));
    printf(", \"v_lshl_add_u32\": {\"ind_w1\": %.2f, \"ind_w4\": %.2f}", run<5, 0>(cus, 1, iters, out, clk), run<5, 0>(cus, 4, iters, out, clk));
    printf("}\n");
    return 0;
}
