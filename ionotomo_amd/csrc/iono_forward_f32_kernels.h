// float32 "fast mode" of the bundle-stationary forward: float32 window images, packed-float32 interpolation, float64 sums
#ifndef IONO_FORWARD_F32_KERNELS_H
#define IONO_FORWARD_F32_KERNELS_H

namespace {

// ---- bytes_per_ray_integral carries sizeof(T_grid) for a float32 grid ------------------------------
// The float64 bundle kernel (k_forward_bundle) sits on vector issue: 28 float64 instructions + 8 ds_read_b64 per sample.  With float32
// STORAGE the same structure pays 8 conversions more per sample and wins nothing; what float32 buys is the ARITHMETIC:
//   * the window image holds 4-byte values: a column is 12 levels = 48 B = three 16-byte pieces (float64: 10 levels = 80 B = five),
//     so one wave-load stages 21 columns instead of 12 and a window is 2-3 loads, all through registers;
//   * a sample's eight corners are FOUR ds_read2_b32 -- each returns (column j, column j + 1) of one level in a register pair, so the
//     z- and x-lerps run on both columns at once with v_pk_add_f32 / v_pk_fma_f32 (2 + 2 + 2 packed instructions) and only the final
//     y-lerp is scalar: 8 vector instructions for the interpolation against 14;
//   * positions are float32 RELATIVE TO THE WINDOW ORIGIN (< 32, so one unit in the last place is 2e-6 of a cell), re-based from
//     the float64 ray every 8 samples; the LDS address falls out of three float32 fmas and one conversion;
//   * a chunk's 8 terms are summed in float32 with float32 Simpson weights, the chunk sums in float64.
// Error budget (measured: tests/test_gpu_parity.py, bench.py extra.f32_*): node values rounded once (6e-8), weights 2e-6 of a cell x
// the relative gradient per cell, three lerp roundings per sample, random over 257 samples: TEC ~1e-7 relative -- inside north_star's
// 1e-6 and SURVEY 8(d)'s gate (dTEC <= 1e-6 max|TEC|), outside the 2e-7 of float32 storage with float64 arithmetic, which is why
// this is a separate mode (storage="f32" + a forward plan), never the float64 headline.
// Chunks whose window does not fit, bundles whose rays were edited in place, and the rays outside the served bundles take float64
// arithmetic on the float32 values (direct loads), as the unplanned float32 kernels do.
#define F_KC 8                 // samples per chunk
#define F_PPC 3                // 16-byte pieces per staged column
#define F_LEV 12               // levels per column: 7 dfz + 2 + up to 3 (a window starts on a multiple of 4 levels)
#define F_COLB 48              // bytes per column
#ifndef F_CAPCOLS
#define F_CAPCOLS 133          // columns per wave image: 6 384 B, five workgroups per CU (200 columns / four workgroups: 0.083 against 0.081 ms)
#endif
#ifndef F_WPE
#define F_WPE 4, 5             // waves per SIMD the kernel is compiled for
#endif
#ifndef F_LEVEL_MAJOR
#define F_LEVEL_MAJOR 1        // image layout: 1 = [level][column] (rays of a bundle sit on the same level: their columns fall into
#endif                         //   different LDS banks), 0 = [column][level] as staged (31 % of the LDS cycles were bank conflicts)
#define F_WAVE_LDS (F_CAPCOLS * F_COLB)      // 9 600 B per wave, as the float64 kernel: four workgroups per CU
#define F_SLOTS 21             // columns per wave-load (63 lanes x 16 B)
#define F_NPF 6                // wave-loads per window, all prefetched through registers

typedef float f32x2 __attribute__((ext_vector_type(2)));

// window of chunk c of bundle b for the float32 kernel: {element offset of the window origin (imin, jmin, kz0) in the values array,
// imin | jmin << 16, kz0 | wx << 16 | wy << 24, fits | rpl << 8 | nl << 16}; rpl = whole rows per wave-load, nl = wave-loads.
// One wave per bundle; same extents as k_bundle_windows (the samples of a chunk are reached from its first one: eps margin).
__global__ __launch_bounds__(64) void k_bundle_windows_f32(GridView g, const double *__restrict__ origins, const double *__restrict__ dirs,
                                                           const int *__restrict__ order, const int *__restrict__ bstart, int nb, double tmax,
                                                           int Ns, int nchunks, uint4 *__restrict__ win, unsigned long long *__restrict__ fit_count) {
    const int b = blockIdx.x;
    if (b >= nb) return;
    const BundleRays B = load_bundle(g, origins, dirs, order, bstart, b, tmax, Ns);
    const double eps = 1e-5;       // (float32 positions inside a chunk: a wider margin than the float64 kernel's 1e-9)
    int nfit = 0;
    for (int c = 0; c < nchunks; ++c) {
        uint4 w = make_uint4(0, 0, 0, 0);
        if (B.any) {
            const int k0 = c * F_KC, ke = min(k0 + F_KC, Ns);
            const double kd0 = (double)k0, kd1 = (double)(ke - 1);
            const double fx = fma(kd0, B.dfx, B.fx0), fy = fma(kd0, B.dfy, B.fy0), fz = fma(kd0, B.dfz, B.fz0);
            const double fxe = fma(kd1, B.dfx, B.fx0), fye = fma(kd1, B.dfy, B.fy0), fze = fma(kd1, B.dfz, B.fz0);
            const int imin = wave_minmax_i32<false>((int)fmax(fmin(fx, fxe) - eps, 0.0)), imax = wave_minmax_i32<true>((int)(fmax(fx, fxe) + eps));
            const int jmin = wave_minmax_i32<false>((int)fmax(fmin(fy, fye) - eps, 0.0)), jmax = wave_minmax_i32<true>((int)(fmax(fy, fye) + eps));
            const int kmin = wave_minmax_i32<false>((int)fmax(fmin(fz, fze) - eps, 0.0)), kmax = wave_minmax_i32<true>((int)(fmax(fz, fze) + eps));
            const int kz0 = kmin & ~3;
            const int wx = imax - imin + 2, wy = jmax - jmin + 2, nlev = kmax + 2 - kz0;
            const int rpl = wy <= F_SLOTS ? F_SLOTS / wy : 1, nl = (wx + rpl - 1) / rpl;
            // (+ the buffer descriptor of a window spans wx planes: below 2^31 bytes)
            const bool fits = wy <= F_SLOTS && nlev <= F_LEV && nl <= F_NPF && nl * rpl * wy <= F_CAPCOLS && wx < 256 && imin < 65536 && jmin < 65536 &&
                              kz0 < 65536 && (unsigned long long)wx * (unsigned long long)g.ny * (unsigned long long)g.nz * 4ull < (1ull << 31);
            w = make_uint4(((unsigned)imin * (unsigned)g.ny + (unsigned)jmin) * (unsigned)g.nz + (unsigned)kz0, (unsigned)imin | ((unsigned)jmin << 16),
                           (unsigned)kz0 | ((unsigned)min(wx, 255) << 16) | ((unsigned)min(wy, 255) << 24),
                           (fits ? 1u : 0u) | ((unsigned)rpl << 8) | ((unsigned)min(nl, 255) << 16));
            nfit += fits;
        }
        if ((threadIdx.x & 63) == 0) win[(size_t)b * nchunks + c] = w;
    }
    if (fit_count && (threadIdx.x & 63) == 0) {
        if (nfit) atomicAdd(fit_count, (unsigned long long)nfit);
        if (B.any) atomicAdd(fit_count + 1, (unsigned long long)nchunks);
    }
}

struct FWin {                  // a float32 window record, decoded (all wave-uniform)
    unsigned woff;
    int imin, jmin, kz0, wx, wy, fits, rpl, nl;
};
__device__ __forceinline__ FWin fwin_decode(uint4 w) {
    FWin W;
    W.woff = (unsigned)__builtin_amdgcn_readfirstlane((int)w.x);
    const int a = __builtin_amdgcn_readfirstlane((int)w.y), b = __builtin_amdgcn_readfirstlane((int)w.z), f = __builtin_amdgcn_readfirstlane((int)w.w);
    W.imin = a & 0xffff, W.jmin = (a >> 16) & 0xffff, W.kz0 = b & 0xffff, W.wx = (b >> 16) & 255, W.wy = (b >> 24) & 255;
    W.fits = f & 1, W.rpl = (f >> 8) & 255, W.nl = (f >> 16) & 255;
    return W;
}

// the eight corners of a sample: (column j, column j + 1) of level k and of level k + 1, rows i (address a) and i + 1 (address a2)
__device__ __forceinline__ void lds_read8_f32(f32x2 &p0, f32x2 &p1, f32x2 &q0, f32x2 &q1, unsigned a, unsigned a2) {
#if F_LEVEL_MAJOR
    // image[level][column], F_CAPCOLS words per level: the next column is the next word, the next level F_CAPCOLS words on
    asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %4 offset0:%6 offset1:%7\n\t"
                 "ds_read2_b32 %2, %5 offset1:1\n\tds_read2_b32 %3, %5 offset0:%6 offset1:%7"
                 : "=&v"(p0), "=&v"(p1), "=&v"(q0), "=&v"(q1)
                 : "v"(a), "v"(a2), "n"(F_CAPCOLS), "n"(F_CAPCOLS + 1)
                 : "memory");
#else
    asm volatile("ds_read2_b32 %0, %4 offset1:12\n\tds_read2_b32 %1, %4 offset0:1 offset1:13\n\t"
                 "ds_read2_b32 %2, %5 offset1:12\n\tds_read2_b32 %3, %5 offset0:1 offset1:13"
                 : "=&v"(p0), "=&v"(p1), "=&v"(q0), "=&v"(q1)
                 : "v"(a), "v"(a2)
                 : "memory");
#endif
}
static_assert(F_COLB == 48, "ds_read2_b32 offset1:12 is the next column");
static_assert(!F_LEVEL_MAJOR || F_CAPCOLS + 1 <= 255, "ds_read2_b32 offsets are 8 bits");
// (value of row i, value of row i + 1) -> interpolated: z on both columns at once, x on both columns at once, y last
__device__ __forceinline__ float lerp8_f32(f32x2 p0, f32x2 p1, f32x2 q0, f32x2 q1, float tx, float ty, float tz) {
    const f32x2 tz2 = {tz, tz}, tx2 = {tx, tx};
    const f32x2 r0 = __builtin_elementwise_fma(tz2, p1 - p0, p0);      // row i    : (column j, column j + 1) at the sample's z
    const f32x2 r1 = __builtin_elementwise_fma(tz2, q1 - q0, q0);      // row i + 1
    const f32x2 s = __builtin_elementwise_fma(tx2, r1 - r0, r0);       // at the sample's x
    return __builtin_fmaf(ty, s.y - s.x, s.x);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(F_WPE))) void k_forward_bundle_f32(
    GridView g, const double *__restrict__ origins, const double *__restrict__ dirs, const BundleRec *__restrict__ brec,
    const uint2 *__restrict__ bhash, const uint4 *__restrict__ win, int nb, int nchunks, double tmax, int Ns,
    const float *__restrict__ unitw32, const double *__restrict__ unitw, double *__restrict__ tec, int *flags) {
    extern __shared__ __attribute__((aligned(16))) char blds[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);     // XCD-major
    if (b >= nb) return;
    const float *M = (const float *)g.M;
    const float *b00 = M, *b01 = b00 + g.nz, *b10 = b00 + (size_t)g.ny * g.nz, *b11 = b10 + g.nz;
    BundleRays B;
    {
        const double2 *rp = (const double2 *)(brec + (size_t)b * 64 + lane);
        const double2 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
        B.fx0 = q0.x, B.fy0 = q0.y, B.fz0 = q1.x, B.dfx = q1.y, B.dfy = q2.x, B.dfz = q2.y, B.h = q3.x;
        const unsigned long long tag = (unsigned long long)__double_as_longlong(q3.y);
        B.r = (int64_t)(tag & BREC_RMASK), B.mine = (tag & BREC_MINE) != 0, B.valid = (tag & BREC_VALID) != 0;
        B.any = __builtin_amdgcn_readfirstlane((int)(tag >> 60) & 1) != 0;
        B.stale = false;
    }
    char *img = blds + wid * F_WAVE_LDS;
    double *part = (double *)(blds + B_SPLIT * F_WAVE_LDS);
    int *sflag = (int *)(part + B_SPLIT * 64);
    const int cs = (int)(((unsigned)lane * 43691u) >> 17);            // lane / 3: column slot of a wave-load
    const int pc = lane - F_PPC * cs;                                 // piece: levels kz0 + 4 pc .. + 3
    const float csf = (float)cs + 0.5f;
    const uint4 *wb = win + (size_t)b * nchunks;
    const int c0 = nchunks * wid / B_SPLIT, c1 = nchunks * (wid + 1) / B_SPLIT;
    const unsigned plane4 = (unsigned)g.ny * (unsigned)g.nz * 4u;     // bytes between two rows (i, i + 1) of a window in memory
    const f32x2 dxy = {(float)B.dfx, (float)B.dfy};
    const float dz = (float)B.dfz;
    u32x4 pre[F_NPF];
#pragma unroll
    for (int n = 0; n < F_NPF; ++n) pre[n] = u32x4{0u, 0u, 0u, 0u};
    bool act = false;
    // the wave-loads of a window: lane = (row r, column dj, piece pc) of a load of rpl rows; rows beyond the window's last one fall
    // outside the buffer descriptor (num_records = wx planes) and read zeros -- no per-load predicate
    auto issue = [&](const FWin &W) {
        const int r = (int)(csf * __builtin_amdgcn_rcpf((float)W.wy));                 // cs / wy (cs <= 21: exact)
        const int dj = cs - r * W.wy;
        const unsigned loff = (unsigned)r * plane4 + ((unsigned)dj * (unsigned)g.nz + 4u * (unsigned)pc) * 4u;
        act = cs < W.rpl * W.wy;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)M + (size_t)W.woff * 4), (short)0,
                                                                            (int)((unsigned)W.wx * plane4), (int)0x00020000);
        const unsigned gs32 = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)W.rpl * plane4));
        if (act) {
#pragma unroll
            for (int n = 0; n < F_NPF; ++n)
                if (n < W.nl) pre[n] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)loff, (int)((unsigned)n * gs32), 0);
        }
    };
    double acc = 0.0;
    FWin Wc = fwin_decode(B.any && c0 < c1 ? wb[c0] : make_uint4(0, 0, 0, 0));
    if (Wc.fits) issue(Wc);
    if (wid == 0) {           // do the arrays still hold the rays this plan was made for?  (overlaps with the first window's loads)
        bool differs = false;
        if (B.mine) {
            const uint2 want = bhash[(size_t)b * 64 + lane], have = ray_hash(origins, dirs, B.r);
            differs = (want.x != have.x) | (want.y != have.y);
        }
        B.stale = __any(differs);
        const bool leaves = __any(B.mine && !B.valid);
        if (lane == 0) {
            sflag[0] = B.stale ? 1 : 0;
            if (B.stale) atomicOr(flags + 2, 1);
            else if (leaves) atomicOr(flags, 1);
        }
    }
    for (int c = c0; c < c1 && B.any; ++c) {
        const int k0 = c * F_KC, ke = min(k0 + F_KC, Ns);
        const double kd0 = (double)k0;
        const double fx = fma(kd0, B.dfx, B.fx0), fy = fma(kd0, B.dfy, B.fy0), fz = fma(kd0, B.dfz, B.fz0);
        // this chunk's eight float32 weights and the next chunk's window record: two scalar loads (unitw32 is padded with zeros)
        typedef float f32x8 __attribute__((ext_vector_type(8)));
        f32x8 wq;
        u32x4 wn;
        asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx4 %1, %3, 0x0" : "=&s"(wq), "=&s"(wn) : "s"(unitw32 + k0), "s"(wb + min(c + 1, c1 - 1)) : "memory");
        __builtin_amdgcn_s_waitcnt(0x0f70);                     // vmcnt(0): this chunk's window rows have landed
        if (Wc.fits && act) {
#if F_LEVEL_MAJOR
            // transposed on the way in: the lane's piece = 4 levels of one column -> 4 words F_CAPCOLS apart
            const unsigned lstep = (unsigned)(Wc.rpl * Wc.wy) * 4u;
            char *dst = img + ((unsigned)(4 * pc) * F_CAPCOLS + (unsigned)cs) * 4u;
#pragma unroll
            for (int n = 0; n < F_NPF; ++n)
                if (n < Wc.nl) {
                    unsigned *q = (unsigned *)(dst + n * lstep);
                    q[0] = pre[n].x, q[F_CAPCOLS] = pre[n].y, q[2 * F_CAPCOLS] = pre[n].z, q[3 * F_CAPCOLS] = pre[n].w;
                }
#else
            const unsigned lstep = (unsigned)(Wc.rpl * Wc.wy) * F_COLB;
            char *dst = img + lane * 16;
#pragma unroll
            for (int n = 0; n < F_NPF; ++n)
                if (n < Wc.nl) *(u32x4 *)(dst + n * lstep) = pre[n];
#endif
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wq), "+s"(wn)::"memory");
        const FWin Wn = fwin_decode(make_uint4(wn.x, wn.y, wn.z, c + 1 < c1 ? wn.w : 0u));
        const FWin Wuse = Wc;
        if (Wn.fits) issue(Wn);                                 // the next chunk's window: in flight during this chunk's samples
        Wc = Wn;
        if (Wuse.fits) {
            // positions relative to the window origin, float32; LDS byte address = image + ((ri wy + rj) + rk F_CAPCOLS) 4, formed in float32
            const f32x2 rxy = {(float)(fx - (double)Wuse.imin), (float)(fy - (double)Wuse.jmin)};
            const float rz = (float)(fz - (double)Wuse.kz0);
#if F_LEVEL_MAJOR
            const float rowb = (float)(Wuse.wy * 4), imgf = (float)(unsigned)(size_t)img, colb = 4.0f, levb = (float)(F_CAPCOLS * 4);
            const unsigned row = (unsigned)(Wuse.wy * 4);
#else
            const float rowb = (float)(Wuse.wy * F_COLB), imgf = (float)(unsigned)(size_t)img, colb = (float)F_COLB, levb = 4.0f;
            const unsigned row = (unsigned)(Wuse.wy * F_COLB);
#endif
            float acc32 = 0.0f;
            auto sample = [&](float uf, f32x2 &p0, f32x2 &p1, f32x2 &q0, f32x2 &q1, float &tx, float &ty, float &tz) {
                const f32x2 u2 = {uf, uf};
                const f32x2 pxy = __builtin_elementwise_fma(u2, dxy, rxy);
                const float pz = __builtin_fmaf(uf, dz, rz);
                const float fi = __builtin_floorf(__builtin_fabsf(pxy.x)), fj = __builtin_floorf(__builtin_fabsf(pxy.y)), fk = __builtin_floorf(__builtin_fabsf(pz));
                tx = pxy.x - fi, ty = pxy.y - fj, tz = pz - fk;
                const unsigned a = (unsigned)__builtin_fmaf(fi, rowb, __builtin_fmaf(fj, colb, __builtin_fmaf(fk, levb, imgf)));
                lds_read8_f32(p0, p1, q0, q1, a, a + row);
            };
            if (ke - k0 == F_KC) {
                // software pipeline: the reads of sample u + 1 are in flight while sample u is interpolated (LDS returns in order)
                f32x2 p0[2], p1[2], q0[2], q1[2];
                float tx[2], ty[2], tz[2];
                sample(0.0f, p0[0], p1[0], q0[0], q1[0], tx[0], ty[0], tz[0]);
#pragma unroll
                for (int u = 0; u < F_KC; ++u) {
                    const int s = u & 1, t = s ^ 1;
                    if (u + 1 < F_KC) {
                        sample((float)(u + 1), p0[t], p1[t], q0[t], q1[t], tx[t], ty[t], tz[t]);
                        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    asm volatile("" : "+v"(p0[s]), "+v"(p1[s]), "+v"(q0[s]), "+v"(q1[s]));
                    acc32 = __builtin_fmaf(wq[u], lerp8_f32(p0[s], p1[s], q0[s], q1[s], tx[s], ty[s], tz[s]), acc32);
                }
            } else {
                for (int k = k0; k < ke; ++k) {                 // (the last chunk of a ray: Ns = 32 x 8 + 1)
                    f32x2 p0, p1, q0, q1;
                    float tx, ty, tz;
                    sample((float)(k - k0), p0, p1, q0, q1, tx, ty, tz);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(q0), "+v"(q1));
                    acc32 = __builtin_fmaf(unitw32[k], lerp8_f32(p0, p1, q0, q1, tx, ty, tz), acc32);
                }
            }
            acc += (double)acc32;
        } else {
            double px = fx, py = fy, pz = fz;
            for (int k = k0; k < ke; ++k) {                     // window too large for the image: direct loads, float64 arithmetic
                acc = fma(unitw[k], trilinear_u<float>(b00, b01, b10, b11, g.ny, g.nz, px, py, pz), acc);
                px += B.dfx, py += B.dfy, pz += B.dfz;
            }
        }
    }
    part[wid * 64 + lane] = acc;
    __syncthreads();
    double hscale = B.h;
    bool valid = B.valid;
    if (sflag[0]) {
        // the arrays do not hold the planned rays: the whole bundle again FROM THE ARRAYS with direct loads (what an unplanned launch gives)
        URay u = {};
        if (B.mine) u = load_uray(g, origins, dirs, B.r, tmax, Ns);
        valid = u.valid, hscale = u.h;
        acc = 0.0;
        if (B.mine && u.valid) {
            for (int c = c0; c < c1; ++c) {
                const int k0 = c * F_KC, ke = min(k0 + F_KC, Ns);
                const double kd0 = (double)k0;
                double px = fma(kd0, u.dfx, u.fx0), py = fma(kd0, u.dfy, u.fy0), pz = fma(kd0, u.dfz, u.fz0);
                for (int k = k0; k < ke; ++k) {
                    acc = fma(unitw[k], trilinear_u<float>(b00, b01, b10, b11, g.ny, g.nz, px, py, pz), acc);
                    px += u.dfx, py += u.dfy, pz += u.dfz;
                }
            }
        }
        if (wid == 0 && __any(B.mine && !u.valid) && lane == 0) atomicOr(flags, 1);
        __syncthreads();
        part[wid * 64 + lane] = acc;
        __syncthreads();
    }
    if (wid == 0 && B.mine) {
        const double *pl = part + lane;
        const double tot = ((pl[0] + pl[64]) + pl[128]) + pl[192];
        tec[B.r] = valid ? tot * hscale : nan("");
    }
}

}  // namespace

#endif
