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
// that are only 8-B aligned (what a (level, level + 1) pair of float64 nodes is).  Lanes read consecutive 16-B (8-B) words:
// conflict-free in every lane group of MI355X_MICROARCH.md's LDS table.
// Round 4: the round-3 loop drained lgkmcnt(0) and issued 8 v_add_f64 after every 8 reads on 4 waves per
// SIMD -- issue-bound, 64 % of the array's rate.  Now: 16 reads per loop body in ONE asm block, every one with its own
// immediate offset (no address arithmetic between them), NO s_waitcnt and no vector instruction inside the loop (LDS data
// returns in order, so re-using the destination registers while older reads are in flight is safe for a rate measurement;
// issue stalls only when the 4-bit lgkmcnt is full = 15 reads in flight), 8 workgroups of 4 waves per CU = 8 waves per SIMD
// (18 KB of LDS per workgroup).  The registers are consumed once, after the loop, by integer XORs.  s_memtime (shader clock)
// and s_memrealtime (100 MHz) around the loop give the clock the CU actually held: B/clk/CU = bytes / (cycles x CUs).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define R16(OP, T, O0, ST)                                                                                                   \
    asm volatile(OP " %0, %16 offset:%17\n\t" OP " %1, %16 offset:%18\n\t" OP " %2, %16 offset:%19\n\t" OP " %3, %16 offset:%20\n\t" \
                 OP " %4, %16 offset:%21\n\t" OP " %5, %16 offset:%22\n\t" OP " %6, %16 offset:%23\n\t" OP " %7, %16 offset:%24\n\t" \
                 OP " %8, %16 offset:%25\n\t" OP " %9, %16 offset:%26\n\t" OP " %10, %16 offset:%27\n\t" OP " %11, %16 offset:%28\n\t" \
                 OP " %12, %16 offset:%29\n\t" OP " %13, %16 offset:%30\n\t" OP " %14, %16 offset:%31\n\t" OP " %15, %16 offset:%32" \
                 : "=&v"(T[0]), "=&v"(T[1]), "=&v"(T[2]), "=&v"(T[3]), "=&v"(T[4]), "=&v"(T[5]), "=&v"(T[6]), "=&v"(T[7]),      \
                   "=&v"(T[8]), "=&v"(T[9]), "=&v"(T[10]), "=&v"(T[11]), "=&v"(T[12]), "=&v"(T[13]), "=&v"(T[14]), "=&v"(T[15]) \
                 : "v"(addr), "n"(O0), "n"(O0 + ST), "n"(O0 + 2 * ST), "n"(O0 + 3 * ST), "n"(O0 + 4 * ST), "n"(O0 + 5 * ST),       \
                   "n"(O0 + 6 * ST), "n"(O0 + 7 * ST), "n"(O0 + 8 * ST), "n"(O0 + 9 * ST), "n"(O0 + 10 * ST), "n"(O0 + 11 * ST),    \
                   "n"(O0 + 12 * ST), "n"(O0 + 13 * ST), "n"(O0 + 14 * ST), "n"(O0 + 15 * ST)                                       \
                 : "memory")
template <int KIND>
__global__ __launch_bounds__(256) void k_lds(int iters, unsigned *__restrict__ sink, unsigned long long *__restrict__ clk) {
    __shared__ __attribute__((aligned(16))) double buf[2304];                      // 18 KB: 8 workgroups per CU
    for (int i = threadIdx.x; i < 2304; i += 256) buf[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned addr = (unsigned)(size_t)buf + (KIND <= 1 ? lane * 8 : lane * 16);      // (ds_read2_b64: pairs 8 B apart)
    u32x2 w[16];
    u32x4 v[16];
    unsigned long long c0 = 0, c1 = 0, r0 = 0, r1 = 0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (KIND == 1) R16("ds_read_b64", w, 0, 520);                              // 15 * 520 + 512 = 8 312 B
        else if (KIND == 2) R16("ds_read_b128", v, 0, 1040);                       // 15 * 1040 + 1024 = 16 624 B
        else if (KIND == 3) R16("ds_read_b128", v, 8, 1040);
        else {                                                                     // ds_read2_b64: two 8-B words at +8 and +16
            asm volatile("ds_read2_b64 %0, %8 offset0:1 offset1:2\n\tds_read2_b64 %1, %8 offset0:3 offset1:4\n\t"
                         "ds_read2_b64 %2, %8 offset0:5 offset1:6\n\tds_read2_b64 %3, %8 offset0:7 offset1:8\n\t"
                         "ds_read2_b64 %4, %8 offset0:9 offset1:10\n\tds_read2_b64 %5, %8 offset0:11 offset1:12\n\t"
                         "ds_read2_b64 %6, %8 offset0:13 offset1:14\n\tds_read2_b64 %7, %8 offset0:15 offset1:16"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                         : "v"(addr)
                         : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    unsigned acc = 0;
    const int nreg = KIND == 0 ? 8 : 16;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        if (u >= nreg) break;
        if (KIND == 1) {
            asm volatile("" : "+v"(w[u]));                  // (orders the XORs behind the final wait)
            acc ^= w[u].x ^ w[u].y;
        } else {
            asm volatile("" : "+v"(v[u]));
            acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0, clk[1] = r1 - r0;
}
struct LdsRate { double seconds, shader_mhz; };
template <int KIND>
LdsRate run_lds(int blocks, int iters, unsigned *sink, unsigned long long *clk, hipStream_t s) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_lds<KIND>), dim3(blocks), dim3(256), 0, s, iters, sink, clk);
    CK(hipEventRecord(a, s));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_lds<KIND>), dim3(blocks), dim3(256), 0, s, iters, sink, clk);
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long h[2];
    CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    return {ms / reps * 1e-3, h[1] ? (double)h[0] / ((double)h[1] / 100.0) : 0.0};      // s_memrealtime ticks at 100 MHz
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
        unsigned long long *clk;
        CK(hipMalloc((void **)&clk, 16));
        CK(hipMemset(clk, 0, 16));
        const int iters = 4096;
        const double reads = (double)blocks * 256 * iters * 16;          // lane-reads per launch (ds_read2_b64: 8 instructions of 2 words)
        const LdsRate t0 = run_lds<0>(blocks, iters, sink, clk, s), t1 = run_lds<1>(blocks, iters, sink, clk, s),
                      t2 = run_lds<2>(blocks, iters, sink, clk, s), t3 = run_lds<3>(blocks, iters, sink, clk, s);
        const double g0 = reads * 8 / t0.seconds / 1e9, g1 = reads * 8 / t1.seconds / 1e9, g2 = reads * 16 / t2.seconds / 1e9,
                     g3 = reads * 16 / t3.seconds / 1e9;
        printf(", \"lds\": {\"read2_b64_gbs\": %.1f, \"read_b64_gbs\": %.1f, \"read_b128_gbs\": %.1f, \"read_b128_8B_aligned_gbs\": %.1f, "
               "\"read_b64_shader_mhz\": %.0f, \"read_b128_shader_mhz\": %.0f, \"read_b64_bytes_per_clk_per_cu\": %.1f, "
               "\"read_b128_bytes_per_clk_per_cu\": %.1f, \"waves_per_simd\": 8, \"reads_in_flight_per_wave\": 15, "
               "\"method\": \"16 reads per loop body, immediate offsets, no s_waitcnt / VALU in the loop; clock = s_memtime / s_memrealtime\"}",
               g0, g1, g2, g3, t1.shader_mhz, t2.shader_mhz, t1.shader_mhz > 0 ? g1 * 1e3 / (t1.shader_mhz * cus) : 0.0,
               t2.shader_mhz > 0 ? g2 * 1e3 / (t2.shader_mhz * cus) : 0.0);
    }
    printf("}\n");
    return 0;
}
