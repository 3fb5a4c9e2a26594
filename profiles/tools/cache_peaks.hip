// cache_peaks.hip -- measured on-chip denominators for bench.py's roofline.
//
//   hipcc --offload-arch=gfx950 -O3 -o profiles/tools/cache_peaks profiles/tools/cache_peaks.hip
//   profiles/tools/cache_peaks            -> one JSON line
//
// Every lane issues global_load_dwordx4 (16 B) in an unrolled loop of independent loads and xors the
// result into a register (so nothing is optimised away).  What differs is the window the addresses fall in:
//   vl1d  : a 2 KiB window per workgroup, 8 workgroups per CU (16 KiB per CU: resident in its 32 KiB vector L1)
//   l2    : a 2 MiB window shared by all workgroups (misses every L1, hits every XCD's 4 MiB L2)
//   mall  : a 128 MiB window (misses L2, hits the 256 MiB Infinity Cache)
//   hbm   : a 2 GiB window streamed once per pass
// and the shape of one wave-load (the texture-address path prices a wave-load by the lines it touches):
//   dense : 64 lanes x 16 B contiguous = 1 KiB = 8 lines
//   same  : all 64 lanes read the same 16 B
//   l4    : 4 groups of 16 lanes, each group one 16-B piece of a different line (the lanes = rays forward kernel)
//   l20   : 64 lanes spread over 20 lines, 8-B-aligned 16-B pieces in runs (the lanes = samples forward kernel)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// shape: 0 dense, 1 same, 2 l4, 3 l20.  window_bytes is a power of two; wg_private: each workgroup has its own window.
template <int SHAPE>
__global__ __launch_bounds__(256) void k_read(const char *__restrict__ base, size_t window_bytes, int wg_private, int iters,
                                              unsigned *__restrict__ sink) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const char *w = base + (wg_private ? (size_t)blockIdx.x * window_bytes : 0);
    const size_t mask = window_bytes - 1;
    size_t lane_off;
    if (SHAPE == 0) lane_off = (size_t)lane * 16;
    else if (SHAPE == 1) lane_off = 0;
    else if (SHAPE == 2) lane_off = (size_t)(lane >> 4) * 128 * 3 + 16;            // 4 lines, 384 B apart
    else lane_off = (size_t)(lane / 3) * 136 + (size_t)(lane % 3) * 8;             // ~22 lines, runs of 3 lanes 8 B apart
    // every wave starts somewhere else in the window and strides through it
    const size_t stride = SHAPE == 0 ? 1024 : 2048;
    size_t pos = ((size_t)blockIdx.x * 4 + wid) * (size_t)iters * 8 * stride;       // contiguous partition of the sweep, wrapped
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const u32x4 v = *(const u32x4 *)(w + ((pos + lane_off) & mask & ~(size_t)7));
            acc ^= v;
            pos += stride;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;      // never true for the zero-filled buffer + keeps the loads
}

// LDS read rates, one instruction form per kernel (inline asm: the compiler picks its own mix otherwise).  KIND 0: ds_read2_b64 on
// 8-B-aligned (not 16-B-aligned) pairs; 1: ds_read_b64; 2: ds_read_b128 on 16-B-aligned addresses; 3: ds_read_b128 on addresses
// that are only 8-B aligned (what a (level, level + 1) pair of float64 nodes is).  Lanes read consecutive 16-B (8-B) words.
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k_lds(int iters, unsigned *__restrict__ sink) {
    __shared__ __attribute__((aligned(16))) double buf[4096 + 64];
    for (int i = threadIdx.x; i < 4096 + 64; i += 256) buf[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned pos = wid * 512;
    const unsigned base = (unsigned)(size_t)buf;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        dbl2 v[8];
        double w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned a16 = base + (((pos + 2 * lane) & 4094) << 3), a8 = base + (((pos + lane) & 4095) << 3);
            if (KIND == 0) asm volatile("ds_read2_b64 %0, %1 offset0:1 offset1:2" : "=v"(v[u]) : "v"(a8));        // (lanes 8 B apart: no bank conflict)
            else if (KIND == 1) asm volatile("ds_read_b64 %0, %1" : "=v"(w[u]) : "v"(a8));
            else if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(a16));
            else asm volatile("ds_read_b128 %0, %1 offset:8" : "=v"(v[u]) : "v"(a16));
            pos += 130;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 1) {
                asm volatile("" : "+v"(w[u]));
                acc += w[u];
            } else {
                asm volatile("" : "+v"(v[u]));
                acc += v[u].x + v[u].y;
            }
        }
    }
    if (acc == 1.2345) sink[0] = 1;
}
template <int KIND>
double run_lds(int blocks, int iters, unsigned *sink, hipStream_t s) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_lds<KIND>), dim3(blocks), dim3(256), 0, s, iters, sink);
    CK(hipEventRecord(a, s));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_lds<KIND>), dim3(blocks), dim3(256), 0, s, iters, sink);
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1e-3;
}

template <int SHAPE>
double run(const char *buf, size_t window, int wg_private, int blocks, int iters, unsigned *sink, hipStream_t s) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_read<SHAPE>), dim3(blocks), dim3(256), 0, s, buf, window, wg_private, iters, sink);
    CK(hipEventRecord(a, s));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_read<SHAPE>), dim3(blocks), dim3(256), 0, s, buf, window, wg_private, iters, sink);
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms / reps * 1e-3;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const size_t big = (size_t)2 << 30;
    char *buf;
    unsigned *sink;
    CK(hipMalloc((void **)&buf, big + 4096));
    CK(hipMemset(buf, 0, big + 4096));
    CK(hipMalloc((void **)&sink, 64));
    CK(hipMemset(sink, 0, 64));
    const int blocks = cus * 8;                 // 8 workgroups of 4 waves per CU
    struct Case { const char *name; size_t window; int priv; int iters; };
    const Case cases[] = {{"vl1d", 2 << 10, 1, 512}, {"l2", 2 << 20, 0, 256}, {"mall", 128 << 20, 0, 64}, {"hbm", big, 0, 32}};
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"wave_load\": \"global_load_dwordx4, 8 independent loads per loop\"", prop.name, cus,
           prop.clockRate / 1000);
    for (const Case &c : cases) {
        // wave-loads per launch: blocks * 4 waves * iters * 8
        const double wl = (double)blocks * 4 * c.iters * 8;
        const double t0 = run<0>(buf, c.window, c.priv, blocks, c.iters, sink, s);
        const double t1 = run<1>(buf, c.window, c.priv, blocks, c.iters, sink, s);
        const double t2 = run<2>(buf, c.window, c.priv, blocks, c.iters, sink, s);
        const double t3 = run<3>(buf, c.window, c.priv, blocks, c.iters, sink, s);
        printf(", \"%s\": {\"dense_gbs\": %.1f, \"dense_ns_per_wave_load_per_cu\": %.3f, \"same_ns\": %.3f, \"l4_ns\": %.3f, \"l20_ns\": %.3f, "
               "\"l20_lane_gbs\": %.1f}",
               c.name, wl * 1024 / t0 / 1e9, t0 / (wl / cus) * 1e9, t1 / (wl / cus) * 1e9, t2 / (wl / cus) * 1e9, t3 / (wl / cus) * 1e9,
               wl * 1024 / t3 / 1e9);
    }
    {
        const int iters = 2048;
        const double lanes = (double)blocks * 256 * iters * 8;
        const double t0 = run_lds<0>(blocks, iters, sink, s), t1 = run_lds<1>(blocks, iters, sink, s), t2 = run_lds<2>(blocks, iters, sink, s),
                     t3 = run_lds<3>(blocks, iters, sink, s);
        printf(", \"lds\": {\"read2_b64_gbs\": %.1f, \"read_b64_gbs\": %.1f, \"read_b128_gbs\": %.1f, \"read_b128_8B_aligned_gbs\": %.1f}",
               lanes * 16 / t0 / 1e9, lanes * 8 / t1 / 1e9, lanes * 16 / t2 / 1e9, lanes * 16 / t3 / 1e9);
    }
    printf("}\n");
    return 0;
}
