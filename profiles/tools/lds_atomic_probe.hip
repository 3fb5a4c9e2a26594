// lds_atomic_probe.hip -- what a ds_add_f64 wave-instruction costs the LDS (round 4).
//   hipcc --offload-arch=gfx950 -O3 -o profiles/tools/lds_atomic_probe profiles/tools/lds_atomic_probe.hip
// 5 workgroups of 4 waves per CU (the back-projection's occupancy), every wave issues `iters` x 8 ds_add_f64 (no return) on a
// 30-KB image; address patterns:
//   0 linear    : lane l -> word l (conflict-free for any lane grouping)
//   1 seg16     : 4 segments of 16 consecutive words at pseudo-random column bases (stride 17 words per column: the kernel's image)
//   2 seg16_q   : as 1, but only the LAST lane of every segment + every 6th lane active (the kernel's "upper" atomics: ~14 lanes)
//   3 same      : all 64 lanes the same word
//   4 seg16_s19 : as 1 with a column stride of 19 words
//   5 linear_h  : as 0 with only lanes 0..31 active
//   6-9         : as 1, lanes 8..15 of every segment one column further (j + 1 / i + 1), column stride 17 / 16 words
// cycles = LDS-array cycles per wave-instruction = elapsed shader cycles x CUs-share: reported as ns per instruction per CU and as
// cycles at the measured clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int PAT, int OP = 0>
__global__ __launch_bounds__(256) void k_atom(int iters, double *out, unsigned long long *clk) {
    __shared__ double tile[15 * 15 * 19 + 512];
    for (int i = threadIdx.x; i < 15 * 15 * 19 + 512; i += 256) tile[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int seg = lane >> 4, sub = lane & 15;
    unsigned h = (blockIdx.x * 4 + wid) * 2654435761u + seg * 40503u;
    unsigned long long c0, c1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        const int stride = PAT == 4 ? 19 : (PAT == 7 || PAT == 9) ? 16 : 17;
        const int col = (h >> 8) % (14 * 15);
        int w;
        bool act = true;
        if (PAT == 0 || PAT == 5) w = lane + (it & 7) * 64;
        else if (PAT == 3) w = 100;
        else w = col * stride + sub;
        if (PAT == 6 || PAT == 7) w += sub >= 8 ? stride : 0;                 // a column change (j + 1) in the middle of every segment
        if (PAT == 8 || PAT == 9) w += sub >= 8 ? 15 * stride : 0;            // ... (i + 1)
        if (PAT == 2) act = sub == 15 || (lane % 6) == 0;
        if (PAT == 5) act = lane < 32;
        double *t = tile + w;
        const int s17 = stride, s15 = 15 * stride;
        if (act) {
            const double v = 1.0;
            // the kernel's 8 atomics: 4 columns x 2 levels
#define ATOM2(INS, A, V) asm volatile(INS " %0, %1\n\t" INS " %0, %1 offset:8" ::"v"((unsigned)(size_t)(A)), "v"(V) : "memory")
            const float vf = 1.0f;
            if (OP == 0) { ATOM2("ds_add_f64", t, v); ATOM2("ds_add_f64", t + s17, v); ATOM2("ds_add_f64", t + s15, v); ATOM2("ds_add_f64", t + s15 + s17, v); }
            if (OP == 1) { ATOM2("ds_add_u64", t, v); ATOM2("ds_add_u64", t + s17, v); ATOM2("ds_add_u64", t + s15, v); ATOM2("ds_add_u64", t + s15 + s17, v); }
            if (OP == 2) { ATOM2("ds_add_f32", t, vf); ATOM2("ds_add_f32", t + s17, vf); ATOM2("ds_add_f32", t + s15, vf); ATOM2("ds_add_f32", t + s15 + s17, vf); }
            if (OP == 3) { ATOM2("ds_add_u32", t, vf); ATOM2("ds_add_u32", t + s17, vf); ATOM2("ds_add_u32", t + s15, vf); ATOM2("ds_add_u32", t + s15 + s17, vf); }
            if (OP == 4) { ATOM2("ds_write_b64", t, v); ATOM2("ds_write_b64", t + s17, v); ATOM2("ds_write_b64", t + s15, v); ATOM2("ds_write_b64", t + s15 + s17, v); }
            if (OP == 5) { ATOM2("ds_max_f64", t, v); ATOM2("ds_max_f64", t + s17, v); ATOM2("ds_max_f64", t + s15, v); ATOM2("ds_max_f64", t + s15 + s17, v); }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tile[100];
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0, clk[1] = r1 - r0;
}
template <int PAT, int OP = 0>
void run(const char *name, int cus, int iters, double *out, unsigned long long *clk) {
    const int blocks = cus * 5;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_atom<PAT, OP>), dim3(blocks), dim3(256), 0, 0, iters, out, clk);
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((k_atom<PAT, OP>), dim3(blocks), dim3(256), 0, 0, iters, out, clk);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long h[2];
    CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double mhz = h[1] ? (double)h[0] / ((double)h[1] / 100.0) : 0.0;
    const double instr_per_cu = 5.0 * 4 * iters * 8;
    const double ns = ms * 1e6 / instr_per_cu;
    printf(", \"%s\": {\"ns_per_instruction_per_cu\": %.3f, \"shader_mhz\": %.0f, \"lds_cycles_per_instruction\": %.2f}", name, ns, mhz, ns * mhz * 1e-3);
}
int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double *out;
    unsigned long long *clk;
    CK(hipMalloc((void **)&out, sizeof(double) * cus * 8));
    CK(hipMalloc((void **)&clk, 64));
    printf("{\"what\": \"ds_add_f64 (no return), 20 waves per CU, 8 per loop body\"");
    run<0>("linear", cus, 4000, out, clk);
    run<5>("linear_half_lanes", cus, 4000, out, clk);
    run<1>("seg16_stride17", cus, 4000, out, clk);
    run<4>("seg16_stride19", cus, 4000, out, clk);
    run<2>("seg16_quarter_lanes", cus, 4000, out, clk);
    run<6>("seg16_stride17_column_change_j", cus, 4000, out, clk);
    run<7>("seg16_stride16_column_change_j", cus, 4000, out, clk);
    run<8>("seg16_stride17_column_change_i", cus, 4000, out, clk);
    run<9>("seg16_stride16_column_change_i", cus, 4000, out, clk);
    run<3>("same_word", cus, 500, out, clk);
    run<7, 1>("u64_seg16_stride16_column_change", cus, 4000, out, clk);
    run<0, 1>("u64_linear", cus, 4000, out, clk);
    run<0, 2>("f32_linear", cus, 4000, out, clk);
    run<0, 3>("u32_linear", cus, 4000, out, clk);
    run<0, 4>("write_b64_linear", cus, 4000, out, clk);
    run<0, 5>("max_f64_linear", cus, 4000, out, clk);
    printf("}\n");
    return 0;
}
