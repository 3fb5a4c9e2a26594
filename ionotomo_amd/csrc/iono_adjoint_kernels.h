// adjoint (back-projection) kernels: plain atomics and the LDS-privatised tile kernel
#ifndef IONO_ADJOINT_KERNELS_H
#define IONO_ADJOINT_KERNELS_H

namespace {

// ------------------------------------------------------------------------------------------------
// adjoint kernels: exact transpose of trilinear + quadrature (SURVEY section 8a, A7')
// ------------------------------------------------------------------------------------------------
template <typename AT>
__device__ __forceinline__ void scatter_trilinear(const GridView &g, const Axes &ax, AT *__restrict__ G, double x, double y,
                                                  double z, double c) {
    const int i = find_cell(ax.x, ax.nx, x, g.inv_h[0], g.uniform[0]);
    const int j = find_cell(ax.y, ax.ny, y, g.inv_h[1], g.uniform[1]);
    const int k = find_cell(ax.z, ax.nz, z, g.inv_h[2], g.uniform[2]);
    const double tx = (x - ax.x[i]) / (ax.x[i + 1] - ax.x[i]);
    const double ty = (y - ax.y[j]) / (ax.y[j + 1] - ax.y[j]);
    const double tz = (z - ax.z[k]) / (ax.z[k + 1] - ax.z[k]);
    AT *p = G + ((size_t)i * g.ny + j) * g.nz + k;
    const size_t sj = g.nz, si = (size_t)g.ny * g.nz;
    const double w0 = c * (1 - tx), w1 = c * tx;
    const double w00 = w0 * (1 - ty), w01 = w0 * ty, w10 = w1 * (1 - ty), w11 = w1 * ty;
    atomicAdd(p, (AT)(w00 * (1 - tz)));
    atomicAdd(p + 1, (AT)(w00 * tz));
    atomicAdd(p + sj, (AT)(w01 * (1 - tz)));
    atomicAdd(p + sj + 1, (AT)(w01 * tz));
    atomicAdd(p + si, (AT)(w10 * (1 - tz)));
    atomicAdd(p + si + 1, (AT)(w10 * tz));
    atomicAdd(p + si + sj, (AT)(w11 * (1 - tz)));
    atomicAdd(p + si + sj + 1, (AT)(w11 * tz));
}

// MODE 0: weights given (w[R]);  MODE 1: fused residual -> differential weights for layout
// [Na][NtNd]: dd = (tec - tec[i0] - dobs)/(CdCt + 1e-15) (inversion/gradient.py:77-81),
// w = dd - [a == i0] sum_a' dd[a']  (transpose of "tec - tec[i0]", forward_equation.py:50)
template <typename AT>
__device__ __forceinline__ void scatter_tricubic(const GridView &g, const Axes &ax, AT *__restrict__ G, double x, double y,
                                                 double z, double c);
template <int MODE>
__device__ __forceinline__ double dd_of(const double *__restrict__ tec, const double *__restrict__ dobs,
                                        const double *__restrict__ cdct, double tref, int64_t r);
template <typename AT, int MODE, int KIND = IONO_INTERP_TRILINEAR>
__global__ __launch_bounds__(256) void k_adjoint_straight(GridView g, const double *__restrict__ origins,
                                                          const double *__restrict__ dirs, const double *__restrict__ wray,
                                                          const double *__restrict__ tec, const double *__restrict__ dobs,
                                                          const double *__restrict__ cdct, int Na, int64_t NtNd, int i0,
                                                          int64_t R, double tmax, int Ns, const double *__restrict__ unitw,
                                                          AT *__restrict__ G, int *oob_flag) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const Axes ax = stage_axes(g, lds);
    const int lane = threadIdx.x & 63;
    bool oob = false;
    for (RayWalk w = ray_walk(R); w.r < w.end; w.r += w.stride) {
        double wr;
        if (MODE == 0) {
            wr = wray[w.r];
        } else {
            const int a = (int)(w.r / NtNd);
            const int64_t p = w.r % NtNd;
            const double tref = MODE == 2 ? 0.0 : tec[(int64_t)i0 * NtNd + p];
            wr = dd_of<MODE>(tec, dobs, cdct, tref, w.r);
            if (a == i0) {
                double s = 0.0;
                for (int a2 = lane; a2 < Na; a2 += 64) s += dd_of<MODE>(tec, dobs, cdct, tref, (int64_t)a2 * NtNd + p);
                wr -= wave_sum(s);
            }
        }
        if (wr == 0.0) continue;
        const StraightRay q = load_straight(origins, dirs, w.r, tmax, Ns);
        const double scale = wr * q.h;
        for (int k = lane; k < Ns; k += 64) {
            double x, y, z;
            straight_point(q, k, Ns, x, y, z);
            if (sample_outside<KIND>(ax, x, y, z)) {
                oob = true;
                continue;
            }
            if (KIND == IONO_INTERP_TRILINEAR) scatter_trilinear<AT>(g, ax, G, x, y, z, scale * unitw[k]);
            else scatter_tricubic<AT>(g, ax, G, x, y, z, scale * unitw[k]);
        }
    }
    if (__any(oob) && lane == 0) atomicOr(oob_flag, 1);
}

// ---- privatised adjoint (ideal-uniform grids) --------------------------------------------------------
// Plain atomics run at ~0.3 TB/s here: every ray of a station crosses the same low-altitude cells,
// and rays of neighbouring stations / consecutive timesteps nearly coincide all the way up, so the
// same addresses are hit thousands of times.  This kernel pre-reduces in LDS.  A workgroup takes a
// BUNDLE of 64 consecutive rays of the walk (callers order the walk so that consecutive rays are
// neighbours in space).  Per slab of 64 samples it keeps a SHEARED tile in LDS: for each of T_TK z
// levels an 8 x 8 window of nodes whose origin follows the bundle's reference ray (its first valid
// ray) at that level.  Contributions falling inside the tile are LDS float atomics (lanes =
// consecutive z levels -> consecutive LDS words, conflict-free); anything outside goes straight to
// global atomics, so the result never depends on how good the ordering is.  After the slab the
// tile's non-zero nodes are flushed with ONE global atomic each.
#define T_WIN 8
#define T_TK 72
#define T_TKP 73
// Timing ablations (drop out-of-window contributions / tile flushes: WRONG results) exist only in the
// -DIONO_ABLATION build that profiles/tools uses; the shipped library has no switch that changes results.
#ifdef IONO_ABLATION
#define ADJ_ABLATE(dbg, bit) ((dbg) & (bit))
#else
#define ADJ_ABLATE(dbg, bit) false
#endif

static_assert(T_WIN == 8 && T_TKP == 73, "tile_offset spells out these strides as shifts");
// (a * T_WIN + b) * T_TKP + m without an integer multiply (v_mul_lo_u32 is quarter rate)
__device__ __forceinline__ int tile_offset(int a, int b, int m) {
    const int cell = (a << 3) + b;
    return (cell << 6) + (cell << 3) + cell + m;
}
template <typename AT>
__device__ __forceinline__ void tile_add4(AT *t, double w00, double w01, double w10, double w11) {
    atomicAdd(t, (AT)w00);
    atomicAdd(t + T_TKP, (AT)w01);
    atomicAdd(t + T_WIN * T_TKP, (AT)w10);
    atomicAdd(t + (T_WIN + 1) * T_TKP, (AT)w11);
}
template <typename AT>
__device__ __forceinline__ void global_add4(AT *__restrict__ G, int i, int j, int kk, int ny, int nz, double w00, double w01,
                                            double w10, double w11) {
    AT *p = G + ((size_t)i * ny + j) * nz + kk;
    atomicAdd(p, (AT)w00);
    atomicAdd(p + nz, (AT)w01);
    atomicAdd(p + (size_t)ny * nz, (AT)w10);
    atomicAdd(p + (size_t)ny * nz + nz, (AT)w11);
}

// One sample's 8 trilinear contributions: the four (i..i+1, j..j+1) nodes of z levels k and k+1 (tile levels m, m+1).
// A 2x2 patch goes to the tile when it lies inside that level's window, to global memory otherwise.  The common case
// -- every lane of the wave inside on both levels, which the bundle selection guarantees for bundles that fit -- is
// a branch-free straight line.
// `field` < 0: trilinear weights (1 - t, t).  `field` = p + 2 q + 4 r >= 0: channel (p, q, r) of the tricubic transpose
// (iono_cubic_kernels.h): per axis the cubic Hermite VALUE weights of the two nodes (bit 0) or their SLOPE weights (1).
__device__ __forceinline__ void axis_pair(double t, int field_bit, bool cubic, double &w0, double &w1) {
    if (!cubic) {
        w0 = 1 - t, w1 = t;
        return;
    }
    const double t2 = t * t, t3 = t2 * t;
    if (field_bit) {
        w0 = t3 - 2.0 * t2 + t, w1 = t3 - t2;
    } else {
        w1 = 3.0 * t2 - 2.0 * t3, w0 = 1.0 - w1;
    }
}
template <typename AT, bool CUBIC>
__device__ __forceinline__ void scatter_sample_tiled(const GridView &g, double *tile, AT *__restrict__ G, const int *I0, const int *J0,
                                                     int kz0, double fx, double fy, double fz, double c, int dbg = 0, int field = -1) {
    // floor(|f|): see load_corners (a validated ray may graze a low face at f = -1e-14; never cell -1)
    const double fi = fmin(__builtin_floor(__builtin_fabs(fx)), (double)(g.nx - 2)),
                 fj = fmin(__builtin_floor(__builtin_fabs(fy)), (double)(g.ny - 2));
    const double fk = fmin(__builtin_floor(__builtin_fabs(fz)), (double)(g.nz - 2));
    const int i = (int)fi, j = (int)fj, k = (int)fk;
    double ax0, ax1, ay0, ay1, uz, tz;
    axis_pair(fx - fi, field & 1, CUBIC, ax0, ax1);
    axis_pair(fy - fj, field & 2, CUBIC, ay0, ay1);
    axis_pair(fz - fk, field & 4, CUBIC, uz, tz);
    const double w0 = c * ax0, w1 = c * ax1;
    const double w00 = w0 * ay0, w01 = w0 * ay1, w10 = w1 * ay0, w11 = w1 * ay1;
    const int m = k - kz0;
    const int m0 = min(max(m, 0), T_TK - 1), m1 = min(max(m + 1, 0), T_TK - 1);      // clamped for the window look-up only
    const unsigned a0 = (unsigned)(i - I0[m0]), b0 = (unsigned)(j - J0[m0]);
    const unsigned a1 = (unsigned)(i - I0[m1]), b1 = (unsigned)(j - J0[m1]);
    const bool in0 = ((unsigned)m < (unsigned)T_TK) & (a0 < (unsigned)(T_WIN - 1)) & (b0 < (unsigned)(T_WIN - 1));
    const bool in1 = ((unsigned)(m + 1) < (unsigned)T_TK) & (a1 < (unsigned)(T_WIN - 1)) & (b1 < (unsigned)(T_WIN - 1));
    if (__all(in0 & in1)) {
        tile_add4<double>(tile + tile_offset((int)a0, (int)b0, m), w00 * uz, w01 * uz, w10 * uz, w11 * uz);
        tile_add4<double>(tile + tile_offset((int)a1, (int)b1, m + 1), w00 * tz, w01 * tz, w10 * tz, w11 * tz);
        return;
    }
    if (in0) tile_add4<double>(tile + tile_offset((int)a0, (int)b0, m), w00 * uz, w01 * uz, w10 * uz, w11 * uz);
    else if (!ADJ_ABLATE(dbg, 4)) global_add4<AT>(G, i, j, k, g.ny, g.nz, w00 * uz, w01 * uz, w10 * uz, w11 * uz);
    if (in1) tile_add4<double>(tile + tile_offset((int)a1, (int)b1, m + 1), w00 * tz, w01 * tz, w10 * tz, w11 * tz);
    else if (!ADJ_ABLATE(dbg, 4)) global_add4<AT>(G, i, j, k + 1, g.ny, g.nz, w00 * tz, w01 * tz, w10 * tz, w11 * tz);
}

// differential weight of ray r = (a, p) in layout [Na][NtNd]:  w = dd(r) - [a == i0] sum_a' dd(a', p)  (the transpose of
// "tec - tec[i0]", inversion/forward_equation.py:50), with
//   MODE 1: dd = (tec - tec[i0] - dobs) / (cdct + 1e-15)     (fused residual, inversion/gradient.py:77-81)
//   MODE 2: dd = v * scale   (v passed as `tec`, scale as `cdct`, null = 1): A^T (scale o v) for the linear solvers
template <int MODE>
__device__ __forceinline__ double dd_of(const double *__restrict__ tec, const double *__restrict__ dobs,
                                        const double *__restrict__ cdct, double tref, int64_t r) {
    if (MODE == 2) return cdct ? tec[r] * cdct[r] : tec[r];
    return (tec[r] - tref - dobs[r]) / (cdct[r] + 1e-15);
}
template <int MODE>
__device__ __forceinline__ double residual_weight(const double *__restrict__ tec, const double *__restrict__ dobs,
                                                  const double *__restrict__ cdct, int Na, int64_t NtNd, int i0, int64_t r) {
    const int a = (int)(r / NtNd);
    const int64_t p = r % NtNd;
    const double tref = MODE == 2 ? 0.0 : tec[(int64_t)i0 * NtNd + p];
    double wr = dd_of<MODE>(tec, dobs, cdct, tref, r);
    if (a == i0) {
        double s = 0.0;
        for (int a2 = 0; a2 < Na; ++a2) s += dd_of<MODE>(tec, dobs, cdct, tref, (int64_t)a2 * NtNd + p);
        wr -= s;
    }
    return wr;
}

// min / max over the 64 lanes (wave-uniform result), same DPP ladder as wave_sum_dpp
template <int CTRL, int ROW_MASK, bool IS_MAX>
__device__ __forceinline__ double dpp_minmax(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const double o = __hiloint2double(hi2, lo2);
    return IS_MAX ? fmax(v, o) : fmin(v, o);
}
template <bool IS_MAX>
__device__ __forceinline__ double wave_minmax_dpp(double v) {
    v = dpp_minmax<0x111, 0xf, IS_MAX>(v);
    v = dpp_minmax<0x112, 0xf, IS_MAX>(v);
    v = dpp_minmax<0x114, 0xf, IS_MAX>(v);
    v = dpp_minmax<0x118, 0xf, IS_MAX>(v);
    v = dpp_minmax<0x142, 0xa, IS_MAX>(v);
    v = dpp_minmax<0x143, 0xc, IS_MAX>(v);
    return bcast_lane(v, 63);
}

// min / max over the wave's first 8, 16, 32 and 64 lanes from ONE ladder: after row_shr 1, 2, 4 lane 7 holds lanes
// 0..7, after row_shr 8 lane 15 of each row holds its row (lanes without a source keep their own value)
template <bool IS_MAX>
__device__ __forceinline__ void wave_prefix_minmax(double v, double &p8, double &p16, double &p32, double &p64) {
    v = dpp_minmax<0x111, 0xf, IS_MAX>(v);
    v = dpp_minmax<0x112, 0xf, IS_MAX>(v);
    v = dpp_minmax<0x114, 0xf, IS_MAX>(v);
    p8 = bcast_lane(v, 7);
    v = dpp_minmax<0x118, 0xf, IS_MAX>(v);
    p16 = bcast_lane(v, 15);
    const double r1 = bcast_lane(v, 31), r2 = bcast_lane(v, 47), r3 = bcast_lane(v, 63);
    p32 = IS_MAX ? fmax(p16, r1) : fmin(p16, r1);
    p64 = IS_MAX ? fmax(fmax(p32, r2), r3) : fmin(fmin(p32, r2), r3);
}
constexpr int ADJ_REF = 16;     // doubles per wave in the bundle scratch: [4,5,7] sums, [8..15] box of its 64 lanes
constexpr int ADJ_SUB = 24;     // + wave 0's boxes of its first 32 / 16 / 8 lanes (kept small: 4 workgroups per CU)

__device__ __forceinline__ double phase_factor(double ne, const double (&wl)[8], const PhaseFreqs &pf);
struct AdjRay {
    URay u;
    double scale;
    double pw[8];       // PHASE: per-frequency weights of the ray (wrf[r][l])
};
// lane-parallel load of `q` rays per wave starting at walk position qw (lanes >= cnt idle)
__device__ __forceinline__ URay load_uray_cubic(const GridView &g, const double *origins, const double *dirs, int64_t r,
                                                double tmax, int Ns);
template <int MODE, bool CUBIC = false, bool PHASE = false>
__device__ __forceinline__ AdjRay load_adj_ray(const GridView &g, const double *origins, const double *dirs, const int *order,
                                               const double *wray, const double *tec, const double *dobs, const double *cdct,
                                               int Na, int64_t NtNd, int i0, int64_t q, bool active, double tmax, int Ns,
                                               bool &oob, int nf = 0, int ldw = 0) {
    AdjRay a;
    a.u = URay{};
    a.scale = 0.0;
#pragma unroll
    for (int l = 0; l < 8; ++l) a.pw[l] = 0.0;
    if (active) {
        const int64_t r = order ? (int64_t)order[q] : q;
        a.u = CUBIC ? load_uray_cubic(g, origins, dirs, r, tmax, Ns) : load_uray(g, origins, dirs, r, tmax, Ns);
        double wr;
        if (PHASE) {                 // `wray` is wrf[R][ldw]; the per-sample factor is applied at the scatter
            bool any = false;
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                if (l < nf) a.pw[l] = wray[(size_t)r * ldw + l];
                any |= a.pw[l] != 0.0;
            }
            wr = any ? 1.0 : 0.0;
        } else {
            wr = MODE == 0 ? wray[r] : residual_weight<MODE>(tec, dobs, cdct, Na, NtNd, i0, r);
        }
        if (a.u.valid) a.scale = wr * a.u.h; else oob = true;
    }
    return a;
}

// `dbg`: bits 4 / 8 -- drop out-of-window contributions / tile flushes -- are timing ablations compiled in only with
// -DIONO_ABLATION.  (The bundle-size A/B switches of rounds 1-5, env IONOTOMO_ADJ_BUNDLE, are closed: smallest bundle 8, largest 64 NW.)
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits until every global
// atomic the wave has in flight is acknowledged (2-3 k cycles under load) -- but everything the tile kernel's barriers
// protect lives in LDS (windows, tile, bundle scratch), and its global atomics are fire-and-forget: they may complete
// in the background while the next slab is scattered.  The hardware drains them at s_endpgm.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <typename AT, int MODE, int NW, bool CUBIC = false, bool PHASE = false, typename GT = double>
__global__ __launch_bounds__(64 * NW) void k_adjoint_straight_tile(GridView g, const double *__restrict__ origins,
                                                               const double *__restrict__ dirs, const int *__restrict__ order,
                                                               const double *__restrict__ wray, const double *__restrict__ tec,
                                                               const double *__restrict__ dobs, const double *__restrict__ cdct,
                                                               int Na, int64_t NtNd, int i0, int64_t R, double tmax, int Ns,
                                                               int dbg, const double *__restrict__ unitw, AT *__restrict__ G,
                                                               int *oob_flag, const int64_t *__restrict__ part, int n_chunks,
                                                               unsigned int *__restrict__ chunk_counter,
                                                               unsigned long long *__restrict__ blk_cycles, int field = -1,
                                                               PhaseFreqs pf = PhaseFreqs{}, int ldw = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *wlds = (double *)smem;                                   // [Ns] quadrature weights
    double *ref = wlds + ((Ns + 1) & ~1);                            // [NW waves][ADJ_REF] per-wave sums and bounding boxes
    double *sub = ref + ADJ_REF * NW;                                // [3 levels][8] wave 0's nested boxes
    // float64 tile whatever the accumulation type of the result (ds_add_f32 measured several times slower than ds_add_f64
    // on gfx950: iono_binned_kernels.h); a float32 result is rounded at the flush
    double *tile = sub + ADJ_SUB;                                         // [T_WIN*T_WIN][T_TKP]
    int *I0 = (int *)(tile + T_WIN * T_WIN * T_TKP);                 // [T_TK] window origins per z level
    int *J0 = I0 + T_TK;
    for (int t = threadIdx.x; t < Ns; t += blockDim.x) wlds[t] = unitw[t];
    for (int t = threadIdx.x; t < T_WIN * T_WIN * T_TKP; t += blockDim.x) tile[t] = 0.0;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nfull = Ns >> 6, ntail0 = nfull << 6;
    const bool tail_by_lane = (Ns - ntail0) <= 8;
    const int nslab = tail_by_lane ? nfull : nfull + 1;
    const double klast = (double)(Ns - 1);
    // PHASE: corner columns of the stored values (the per-sample factor needs the interpolated ne)
    const GT *pb00 = (const GT *)g.M, *pb01 = pb00 + g.nz, *pb10 = pb00 + (size_t)g.ny * g.nz, *pb11 = pb10 + g.nz;
    // Contiguous chunks of the walk per workgroup.  Without `part`: one chunk of equal ray count each (XCD-major).
    // With `part` (n_chunks + 1 boundaries from iono_walk_partition_set, cost-balanced from measured cycles):
    // chunk b goes to workgroup b, and the remaining -- progressively smaller -- chunks are handed out through an
    // atomic counter as workgroups finish (guided self-scheduling: the tail is made of small chunks).
    int64_t bidx = blockIdx.x;
    if ((gridDim.x & 7) == 0) bidx = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int64_t base = R / gridDim.x, rem = R % gridDim.x;
    long long *next_chunk = (long long *)(J0 + T_TK);
    const double BIG = 1e300;
    bool oob = false;
    lds_barrier();
  for (int64_t chunk = bidx;;) {
    const int64_t lo = part ? part[chunk] : chunk * base + min(chunk, rem);
    const int64_t hi = part ? part[chunk + 1] : lo + base + (chunk < rem ? 1 : 0);
    const unsigned long long t_start = __builtin_readcyclecounter();
    int cw = 64 * NW;                                                                          // candidate width of the next bundle
    for (int64_t q0 = lo; q0 < hi;) {
        // ---- candidate bundle: up to 64 NW rays, wave w lanes 0..63 own walk positions q0 + 64 w + l ----------
        // (only the first `cw` walk positions are examined: cw follows the size of the previous bundle, so a sparse
        // stretch of the walk does not load 256 candidates for every 8 rays it consumes)
        int q = 64;                                     // rays per wave
        int64_t qw = q0 + (int64_t)q * wid;
        int cnt = (int)max((int64_t)0, min((int64_t)q, min(hi, q0 + (int64_t)cw) - qw));
        AdjRay a = load_adj_ray<MODE, CUBIC, PHASE>(g, origins, dirs, order, wray, tec, dobs, cdct, Na, NtNd, i0, qw + lane, lane < cnt,
                                             tmax, Ns, oob, pf.nf, ldw);
        int c = 64 * NW;
        for (int round = 0; round < 2; ++round) {
            // per-wave sums (for the mean ray) and bounding boxes at the bottom / top of the rays, for the wave's
            // first 64 / 32 / 16 / 8 rays (levels 0..3: the nested candidates for the bundle size)
            const bool lv = a.scale != 0.0;
            const double live = lv ? 1.0 : 0.0;
            const double xe = fma(klast, a.u.dfx, a.u.fx0), ye = fma(klast, a.u.dfy, a.u.fy0);
            const double s4 = wave_sum_dpp(live * a.u.fz0), s5 = wave_sum_dpp(live * a.u.dfz), s7 = wave_sum_dpp(live);
            double bb[8][4];        // [quantity][level 3..0 = first 8, 16, 32, 64 lanes]
            wave_prefix_minmax<false>(lv ? a.u.fx0 : BIG, bb[0][3], bb[0][2], bb[0][1], bb[0][0]);
            wave_prefix_minmax<true>(lv ? a.u.fx0 : -BIG, bb[1][3], bb[1][2], bb[1][1], bb[1][0]);
            wave_prefix_minmax<false>(lv ? a.u.fy0 : BIG, bb[2][3], bb[2][2], bb[2][1], bb[2][0]);
            wave_prefix_minmax<true>(lv ? a.u.fy0 : -BIG, bb[3][3], bb[3][2], bb[3][1], bb[3][0]);
            wave_prefix_minmax<false>(lv ? xe : BIG, bb[4][3], bb[4][2], bb[4][1], bb[4][0]);
            wave_prefix_minmax<true>(lv ? xe : -BIG, bb[5][3], bb[5][2], bb[5][1], bb[5][0]);
            wave_prefix_minmax<false>(lv ? ye : BIG, bb[6][3], bb[6][2], bb[6][1], bb[6][0]);
            wave_prefix_minmax<true>(lv ? ye : -BIG, bb[7][3], bb[7][2], bb[7][1], bb[7][0]);
            if (lane == 0) {
                double *rp = ref + ADJ_REF * wid;
                rp[4] = s4, rp[5] = s5, rp[7] = s7;
#pragma unroll
                for (int qi = 0; qi < 8; ++qi) rp[8 + qi] = bb[qi][0];
                if (wid == 0) {
#pragma unroll
                    for (int lev = 1; lev < 4; ++lev)
#pragma unroll
                        for (int qi = 0; qi < 8; ++qi) sub[8 * (lev - 1) + qi] = bb[qi][lev];
                }
            }
            lds_barrier();
            if (round == 1) break;
            // largest nested candidate whose rays stay within the tile window at both ends (block-uniform): all NW
            // waves' rays, the first NW/2 waves', ..., wave 0's 64, then wave 0's first 32 / 16 / 8
            const double lim = (double)(T_WIN - 3);
            double m0 = BIG, M0 = -BIG, m1 = BIG, M1 = -BIG, m2 = BIG, M2 = -BIG, m3 = BIG, M3 = -BIG;
            int fit = 0;
            for (int w2 = 0; w2 < NW; ++w2) {
                const double *rp = ref + ADJ_REF * w2 + 8;
                m0 = fmin(m0, rp[0]), M0 = fmax(M0, rp[1]), m1 = fmin(m1, rp[2]), M1 = fmax(M1, rp[3]);
                m2 = fmin(m2, rp[4]), M2 = fmax(M2, rp[5]), m3 = fmin(m3, rp[6]), M3 = fmax(M3, rp[7]);
                const bool ok = (M0 - m0 <= lim) & (M1 - m1 <= lim) & (M2 - m2 <= lim) & (M3 - m3 <= lim);
                if (ok && ((w2 + 1) & w2) == 0) fit = 64 * (w2 + 1);      // 1, 2, 4 (, 8) waves' worth of rays
            }
            if (fit == 0) {
                for (int lev = 1; lev < 4 && fit == 0; ++lev) {
                    const double *rp = sub + 8 * (lev - 1);                // wave 0's first 64 >> lev rays
                    const bool ok = (rp[1] - rp[0] <= lim) & (rp[3] - rp[2] <= lim) & (rp[5] - rp[4] <= lim) &
                                    (rp[7] - rp[6] <= lim);
                    if (ok) fit = 64 >> lev;
                }
            }
            const int cmin = 8, cmax = 64 * NW;
            if (fit < cmin) fit = 0;
            c = min(min(fit ? fit : cmin, cw), cmax);    // nothing fits: cmin rays, their out-of-window parts go straight to global memory
            cw = min(cmax, max(16, 2 * c));
            if (c == 64 * NW) break;
            // re-deal the chosen rays evenly over the waves
            lds_barrier();
            q = c / NW;
            qw = q0 + (int64_t)q * wid;
            cnt = (int)max((int64_t)0, min((int64_t)q, hi - qw));
            a = load_adj_ray<MODE, CUBIC, PHASE>(g, origins, dirs, order, wray, tec, dobs, cdct, Na, NtNd, i0, qw + lane, lane < cnt, tmax,
                                                 Ns, oob, pf.nf, ldw);
        }
        q0 += c;
        if (a.scale != 0.0 && tail_by_lane && nslab == 0) {  // fewer than 9 samples in all: straight to global memory
            for (int k = ntail0; k < Ns; ++k) {
                const double kd = (double)k;
                const double sfx = fma(kd, a.u.dfx, a.u.fx0), sfy = fma(kd, a.u.dfy, a.u.fy0), sfz = fma(kd, a.u.dfz, a.u.fz0);
                double cw = a.scale * wlds[k];
                if (PHASE) cw *= phase_factor(trilinear_u<GT>(pb00, pb01, pb10, pb11, g.ny, g.nz, sfx, sfy, sfz), a.pw, pf);
                scatter_sample_tiled<AT, CUBIC>(g, tile, G, I0, J0, -(1 << 28), sfx, sfy, sfz, cw, dbg, field);
            }
        }
        double nlive = 0.0, sz0 = 0.0, sdz = 0.0;
        for (int w2 = 0; w2 < NW; ++w2) nlive += ref[ADJ_REF * w2 + 7], sz0 += ref[ADJ_REF * w2 + 4], sdz += ref[ADJ_REF * w2 + 5];
        if (nlive == 0.0) {            // nothing to do in this bundle (block-uniform)
            lds_barrier();
            continue;
        }
        // reference line of the bundle: through the centres of its bounding boxes at the bottom and at the
        // top (a bundle that passed the spread test then lies entirely inside the windows); z from the mean
        const double inl = 1.0 / nlive;
        double bx0 = BIG, bx1 = -BIG, by0 = BIG, by1 = -BIG, tx0 = BIG, tx1 = -BIG, ty0 = BIG, ty1 = -BIG;
        for (int w2 = 0; w2 < NW; ++w2) {
            const double *rp = ref + ADJ_REF * w2;
            bx0 = fmin(bx0, rp[8]), bx1 = fmax(bx1, rp[9]), by0 = fmin(by0, rp[10]), by1 = fmax(by1, rp[11]);
            tx0 = fmin(tx0, rp[12]), tx1 = fmax(tx1, rp[13]), ty0 = fmin(ty0, rp[14]), ty1 = fmax(ty1, rp[15]);
        }
        const double rfx0 = 0.5 * (bx0 + bx1), rdfx = (0.5 * (tx0 + tx1) - rfx0) / klast;
        const double rfy0 = 0.5 * (by0 + by1), rdfy = (0.5 * (ty0 + ty1) - rfy0) / klast;
        const double rfz0 = sz0 * inl, rdfz = sdz * inl;
        for (int it = 0; it < nslab; ++it) {
            const int k0 = it << 6;
            const int kz0 = max((int)fma((double)k0, rdfz, rfz0) - 1, 0);
            if (threadIdx.x < T_TK) {     // window origin per z level: follow the reference ray
                const double kk = ((double)(kz0 + (int)threadIdx.x) - rfz0) / rdfz;      // (real) sample index at that level
                I0[threadIdx.x] = (int)floor(fma(kk, rdfx, rfx0)) - (T_WIN / 2 - 1);
                J0[threadIdx.x] = (int)floor(fma(kk, rdfy, rfy0)) - (T_WIN / 2 - 1);
            }
            lds_barrier();
            for (int gi = 0; gi < cnt; ++gi) {
                const double sc = bcast_lane(a.scale, gi);
                if (sc == 0.0) continue;
                const int k = k0 + lane;
                if (k < Ns && (tail_by_lane ? k < ntail0 : true)) {
                    const double kd = (double)k;
                    const double sfx = fma(kd, bcast_lane(a.u.dfx, gi), bcast_lane(a.u.fx0, gi)),
                                 sfy = fma(kd, bcast_lane(a.u.dfy, gi), bcast_lane(a.u.fy0, gi)),
                                 sfz = fma(kd, bcast_lane(a.u.dfz, gi), bcast_lane(a.u.fz0, gi));
                    double cw = sc * wlds[k];
                    if (PHASE) {
                        double wl[8];
#pragma unroll
                        for (int l = 0; l < 8; ++l) wl[l] = bcast_lane(a.pw[l], gi);
                        cw *= phase_factor(trilinear_u<GT>(pb00, pb01, pb10, pb11, g.ny, g.nz, sfx, sfy, sfz), wl, pf);
                    }
                    scatter_sample_tiled<AT, CUBIC>(g, tile, G, I0, J0, kz0, sfx, sfy, sfz, cw, dbg, field);
                }
            }
            if (tail_by_lane && it == nslab - 1 && a.scale != 0.0) {
                // the <= 8 samples after the last full slab, lanes = rays: the tile spans 72 levels for exactly this
                // (anything that still falls outside goes to global memory as everywhere else)
                for (int k = ntail0; k < Ns; ++k) {
                    const double kd = (double)k;
                    const double sfx = fma(kd, a.u.dfx, a.u.fx0), sfy = fma(kd, a.u.dfy, a.u.fy0), sfz = fma(kd, a.u.dfz, a.u.fz0);
                    double cw = a.scale * wlds[k];
                    if (PHASE) cw *= phase_factor(trilinear_u<GT>(pb00, pb01, pb10, pb11, g.ny, g.nz, sfx, sfy, sfz), a.pw, pf);
                    scatter_sample_tiled<AT, CUBIC>(g, tile, G, I0, J0, kz0, sfx, sfy, sfz, cw, dbg, field);
                }
            }
            lds_barrier();
            // ---- flush + re-zero: one global atomic per touched node -------------------------------------
            for (int e = threadIdx.x; e < T_WIN * T_WIN * T_TKP; e += blockDim.x) {
                const double v = tile[e];
                if (v != 0.0) {
                    tile[e] = 0.0;
                    const int cell = e / T_TKP, m = e - cell * T_TKP;
                    const int gi_ = I0[m] + cell / T_WIN, gj_ = J0[m] + cell % T_WIN, gk_ = kz0 + m;
                    if (m < T_TK && gi_ >= 0 && gi_ < g.nx && gj_ >= 0 && gj_ < g.ny && gk_ < g.nz && !ADJ_ABLATE(dbg, 8))
                        atomicAdd(G + ((size_t)gi_ * g.ny + gj_) * g.nz + gk_, (AT)v);
                }
            }
            lds_barrier();
        }
    }
    if (blk_cycles && threadIdx.x == 0) blk_cycles[chunk] = __builtin_readcyclecounter() - t_start;   // per chunk, walk order
    if (!part || n_chunks <= (int)gridDim.x) break;
    lds_barrier();
    if (threadIdx.x == 0) *next_chunk = (long long)gridDim.x + (long long)atomicAdd(chunk_counter, 1u);
    lds_barrier();
    chunk = *next_chunk;
    if (chunk >= n_chunks) break;
  }
    if (__any(oob) && lane == 0) atomicOr(oob_flag, 1);
}

// ---- adjoint of the phase observable (inversion/iterative_newton.py:86-127) w.r.t. the node values ne ------------------
// phi_{r,l} = sum_k c_k f_l(ne_k),  f_l(ne) = 1 - sqrt(1 - ne / n_p,l),  ne_k = sum_v W_kv ne_v   =>
//   d/d ne_v  sum_{r,l} wrf_{r,l} phi_{r,l} = sum_r sum_k c_k q_{r,k} W_kv,   q_{r,k} = sum_l wrf_{r,l} / (2 n_p,l sqrt(1 - ne_k / n_p,l))
// i.e. the trilinear scatter with a per-SAMPLE factor that needs the interpolated value: forward gather and transpose
// scatter in one traversal.  General tier (any grid, plain hardware atomics).
__device__ __forceinline__ double phase_factor(double ne, const double (&wl)[8], const PhaseFreqs &pf) {
    double q = 0.0;
#pragma unroll
    for (int l = 0; l < 8; ++l) q += wl[l] * (0.5 * pf.inv_np[l]) * rsqrt(1.0 - ne * pf.inv_np[l]);
    return q;
}
template <typename GT, typename AT>
__global__ __launch_bounds__(256) void k_adjoint_phase_straight(GridView g, const double *__restrict__ origins,
                                                                const double *__restrict__ dirs, const double *__restrict__ wrf,
                                                                int ldw, PhaseFreqs pf, int64_t R, double tmax, int Ns,
                                                                const double *__restrict__ unitw, AT *__restrict__ G, int *oob_flag) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const Axes ax = stage_axes(g, lds);
    const int lane = threadIdx.x & 63;
    bool oob = false;
    for (RayWalk w = ray_walk(R); w.r < w.end; w.r += w.stride) {
        double wl[8];
        bool any = false;
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            wl[l] = l < pf.nf ? wrf[(size_t)w.r * ldw + l] : 0.0;
            any |= wl[l] != 0.0;
        }
        if (!any) continue;
        const StraightRay q = load_straight(origins, dirs, w.r, tmax, Ns);
        for (int k = lane; k < Ns; k += 64) {
            double x, y, z;
            straight_point(q, k, Ns, x, y, z);
            if (sample_outside<IONO_INTERP_TRILINEAR>(ax, x, y, z)) {
                oob = true;
                continue;
            }
            const double ne = trilinear_at<GT>(g, ax, x, y, z);
            scatter_trilinear<AT>(g, ax, G, x, y, z, q.h * unitw[k] * phase_factor(ne, wl, pf));
        }
    }
    if (__any(oob) && lane == 0) atomicOr(oob_flag, 1);
}

// ---- the reference's SHIPPED gradient discretisation: voxel chord lengths -------------------------------
// geometry/ray_dirac.py:5-34 + inversion/gradient.py:15-20:  grad[v] = sum_rays dd[ray] M[v] chord(ray, v), where chord is
// the length of the straight first-sample -> last-sample line inside the voxel-centred box of node v
// (geometry/slab_method.py:19-58, incl. its `t_enter > 0` rule and NaN -> 0), evaluated for the 27 nodes around every
// sample and ASSIGNED (not accumulated) per ray.  Here: one wave per ray, lanes = samples; a node counts for the first
// sample whose 3 x 3 x 3 neighbourhood contains it (cell indices are monotone along a ray, so "first" = "not in the previous
// sample's neighbourhood") -- one atomic per (ray, node), the reference's assignment semantics without a per-ray volume.
__device__ __forceinline__ int bisection_cell(const double *g, int n, double v) {       // geometry/tri_cubic.py:105-132
    if (v < g[0]) return -1;
    if (v > g[n - 1]) return n;
    if (v == g[n - 1]) return n - 1;
    int lo = 0, hi = n - 1;                                                           // searchsorted(side='right') - 1, clipped
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (g[mid] <= v) lo = mid; else hi = mid;
    }
    return min(max(lo, 0), n - 2);
}
__device__ __forceinline__ void slab_axis(double lo, double hi, double r0, double inv_n, double &tmin, double &tmax) {
    double t1 = (lo - r0) * inv_n, t2 = (hi - r0) * inv_n;
    if (t1 != t1) t1 = 0.0;
    if (t2 != t2) t2 = 0.0;
    tmin = fmin(t1, t2), tmax = fmax(t1, t2);
}
template <typename GT, typename AT>
__global__ __launch_bounds__(256) void k_gradient_chords(GridView g, const double *__restrict__ rays, const double *__restrict__ dd,
                                                         int64_t R, int Ns, AT *__restrict__ G) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const Axes ax = stage_axes(g, lds);
    const int lane = threadIdx.x & 63;
    const double hx = 0.5 * (ax.x[1] - ax.x[0]), hy = 0.5 * (ax.y[1] - ax.y[0]), hz = 0.5 * (ax.z[1] - ax.z[0]);
    const GT *M = (const GT *)g.M;
    for (RayWalk w = ray_walk(R); w.r < w.end; w.r += w.stride) {
        const double *rx = rays + (size_t)w.r * 4 * Ns, *ry = rx + Ns, *rz = ry + Ns;
        const double wr = dd[w.r];
        if (wr == 0.0) continue;
        const double ox = rx[0], oy = ry[0], oz = rz[0];
        double nx = rx[Ns - 1] - ox, ny = ry[Ns - 1] - oy, nz = rz[Ns - 1] - oz;
        const double nrm = sqrt(nx * nx + ny * ny + nz * nz);
        nx /= nrm, ny /= nrm, nz /= nrm;
        const double ix = 1.0 / nx, iy = 1.0 / ny, iz = 1.0 / nz;
        for (int s = lane; s < Ns; s += 64) {
            const int ci = bisection_cell(ax.x, ax.nx, rx[s]), cj = bisection_cell(ax.y, ax.ny, ry[s]), ck = bisection_cell(ax.z, ax.nz, rz[s]);
            int pi = -100, pj = -100, pk = -100;
            if (s > 0) pi = bisection_cell(ax.x, ax.nx, rx[s - 1]), pj = bisection_cell(ax.y, ax.ny, ry[s - 1]), pk = bisection_cell(ax.z, ax.nz, rz[s - 1]);
            for (int xi = max(0, ci - 1); xi < min(ax.nx, ci + 2); ++xi)
                for (int yi = max(0, cj - 1); yi < min(ax.ny, cj + 2); ++yi)
                    for (int zi = max(0, ck - 1); zi < min(ax.nz, ck + 2); ++zi) {
                        if (abs(xi - pi) <= 1 && abs(yi - pj) <= 1 && abs(zi - pk) <= 1) continue;       // the previous sample had it
                        double a0, a1, b0, b1, c0, c1;
                        slab_axis(ax.x[xi] - hx, ax.x[xi] + hx, ox, ix, a0, a1);
                        slab_axis(ax.y[yi] - hy, ax.y[yi] + hy, oy, iy, b0, b1);
                        slab_axis(ax.z[zi] - hz, ax.z[zi] + hz, oz, iz, c0, c1);
                        const double t_in = fmax(fmax(a0, b0), c0), t_out = fmin(fmin(a1, b1), c1);
                        if (t_in < t_out && t_in > 0.0) {
                            const size_t v = ((size_t)xi * ax.ny + yi) * ax.nz + zi;
                            atomicAdd(G + v, (AT)(wr * (double)M[v] * (t_out - t_in)));
                        }
                    }
        }
    }
}

template <typename AT, int KIND = IONO_INTERP_TRILINEAR>
__global__ __launch_bounds__(256) void k_adjoint_rays(GridView g, const double *__restrict__ rays,
                                                      const double *__restrict__ wray, int64_t R, int Ns, int rule,
                                                      AT *__restrict__ G, int *oob_flag) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const Axes ax = stage_axes(g, lds);
    const int lane = threadIdx.x & 63;
    bool oob = false;
    for (RayWalk w = ray_walk(R); w.r < w.end; w.r += w.stride) {
        const double *rx = rays + (size_t)w.r * 4 * Ns, *ry = rx + Ns, *rz = ry + Ns, *rs = rz + Ns;
        const double wr = wray[w.r];
        if (wr == 0.0) continue;
        for (int k = lane; k < Ns; k += 64) {
            const double x = rx[k], y = ry[k], z = rz[k];
            if (sample_outside<KIND>(ax, x, y, z)) {
                oob = true;
                continue;
            }
            if (KIND == IONO_INTERP_TRILINEAR) scatter_trilinear<AT>(g, ax, G, x, y, z, wr * quad_weight(rs, Ns, k, rule));
            else scatter_tricubic<AT>(g, ax, G, x, y, z, wr * quad_weight(rs, Ns, k, rule));
        }
    }
    if (__any(oob) && lane == 0) atomicOr(oob_flag, 1);
}

}  // namespace

#endif
