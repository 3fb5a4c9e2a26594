// A plain float64 FMA kernel (no memory traffic, ~100 us per launch) launched N times back to back after idle: is the per-launch
// duration ramp of profiles/r05_clock_ramp.json a property of the device or of k_forward_bundle?
//   hipcc --offload-arch=gfx950 -O3 -o profiles/tools/ramp_probe profiles/tools/ramp_probe.hip
//   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/ramp2 -- $REPO/profiles/tools/ramp_probe 5000
//   python profiles/tools/clock_ramp.py gpurun_out/ramp2 k_ramp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k_ramp_probe(double *out, int iters, double a, double b) {
    double x0 = threadIdx.x * 1e-3, x1 = x0 + 1.0, x2 = x0 + 2.0, x3 = x0 + 3.0;
    for (int i = 0; i < iters; ++i) {
        x0 = fma(x0, a, b), x1 = fma(x1, a, b), x2 = fma(x2, a, b), x3 = fma(x3, a, b);
        x0 = fma(x0, a, b), x1 = fma(x1, a, b), x2 = fma(x2, a, b), x3 = fma(x3, a, b);
    }
    if (x0 + x1 + x2 + x3 == 12345.678) out[blockIdx.x * 256 + threadIdx.x] = x0;
}
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 5000, iters = argc > 2 ? atoi(argv[2]) : 2500;
    double *out;
    if (hipMalloc((void **)&out, 4096 * 256 * sizeof(double)) != hipSuccess) return 1;
    hipDeviceSynchronize();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_ramp_probe, dim3(4096), dim3(256), 0, 0, out, iters, 0.999999, 1e-6);
    hipDeviceSynchronize();
    printf("%d launches\n", n);
    return 0;
}
