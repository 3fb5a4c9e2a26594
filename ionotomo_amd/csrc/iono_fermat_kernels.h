// fused curved-ray kernels: the Fermat ray ODE and the line integral (or its transpose) in ONE traversal, rays[R,4,Ns] never exists
#ifndef IONO_FERMAT_KERNELS_H
#define IONO_FERMAT_KERNELS_H

namespace {

// The reference's curved path is cast_ray -> Fermat.integrate_ray -> forward_equation on the returned samples
// (geometry/calc_rays.py:61-96, inversion/fermat.py:58-72,150-174, inversion/forward_equation.py:27-28): the samples of every ray are
// stored (x, y, z, s: 4 Ns doubles per ray = 5.1 GB at config 4's ray count) and then integrated with Simpson's rule on the
// NON-UNIFORM abscissae s.  Here the RK4 stepper of k_trace_fermat (same right-hand side, same order of operations: the samples
// are bit-identical to the traced ones) feeds a streaming form of that quadrature, so a ray is traced and integrated -- or traced
// and back-projected -- in registers.
//
// Streaming quadrature.  Every rule the explicit-sample kernels know (quad_weight, iono_device_common.h: composite Simpson for odd
// N; for even N the reference-era even='avg' rule or scipy >= 1.11's end correction; trapezoid) is a sum of contributions of
// intervals (k-1, k) and of triples (k-2, k-1, k), so the weight of sample k-2 is final once sample k has been seen: a window of
// three abscissae / weights, one finished weight per step, the last two flushed at the end.
struct StreamQuad {
    double s0, s1, w0, w1;       // abscissae and weights-so-far of samples k-2 and k-1
    int N, rule;
    __device__ __forceinline__ void init(int N_, int rule_, double s_first) {
        N = N_, rule = rule_, s0 = 0.0, w0 = 0.0, s1 = s_first, w1 = 0.0;
    }
    // sample k >= 1 at abscissa s2: returns the FINAL weight of sample k-2 (meaningful for k >= 2) and shifts the window
    __device__ __forceinline__ double feed(int k, double s2) {
        double w2 = 0.0;
        const double h1 = s2 - s1;
        if (rule == IONO_QUAD_TRAPEZOID || N == 2) {
            w1 += 0.5 * h1, w2 += 0.5 * h1;
        } else {
            const bool even = !(N & 1), avg = even && rule == IONO_QUAD_SIMPSON_AVG;
            if (avg && (k == 1 || k == N - 1)) w1 += 0.25 * h1, w2 += 0.25 * h1;      // the two trapezoid ends, halved by the average
            if (k >= 2) {
                const double h0 = s1 - s0, hs = h0 + h1;
                // Simpson pairs: odd N and the scipy rule pair (0,1,2), (2,3,4), ...; 'avg' also pairs (1,2,3), (3,4,5), ..., each at 1/2
                const double f = (k & 1) ? (avg && k >= 3 ? 0.5 : 0.0) : (avg ? 0.5 : 1.0);
                if (f != 0.0) {
                    const double c = f * hs / 6.0;
                    w0 += c * (2.0 - h1 / h0), w1 += c * (hs * hs / (h0 * h1)), w2 += c * (2.0 - h0 / h1);
                }
                if (even && !avg && k == N - 1) {                                       // scipy >= 1.11: last interval from the last three points
                    w2 += (2.0 * h1 * h1 + 3.0 * h0 * h1) / (6.0 * hs), w1 += (h1 * h1 + 3.0 * h0 * h1) / (6.0 * h0);
                    w0 -= h1 * h1 * h1 / (6.0 * h0 * hs);
                }
            }
        }
        const double out = w0;
        s0 = s1, s1 = s2, w0 = w1, w1 = w2;
        return out;
    }
};

// LDS-privatised scatter of the transpose (trilinear integrand).  The 64 rays of a wave are neighbours of the walk order and climb
// together, so the nodes they hit at one step lie in a window of a few columns and levels -- and the next step hits mostly the same
// columns one level up.  The wave keeps a FW x FW x FWZ-node window of the gradient in LDS (ds_add_f64), moves it when a sample
// falls outside (flush: consecutive lanes = consecutive levels of a column, so the global atomics leave as contiguous runs, zeros
// are skipped) and flushes it at the end.  A sample that does not fit even the re-centred window (rays of the wave far apart)
// goes straight to global atomics: exactness never depends on the rays being neighbours.  Plain hardware atomics per corner
// (64 lanes in 64 different rows) measured 131 ms at 620 000 rays x 257 samples; see profiles/ for the windowed number.
// Window size (round 5, 620 000 rays x 257 samples through a trilinear index, ms per transpose; FW x FW x FWZ): 12 x 12 x 16 (rounds
// 3-4) 13.5 | 12 x 12 x 8: 10.5 | 12 x 12 x 4: 10.0 | 16 x 16 x 4: 9.0 | 14 x 14 x 4: 8.3 | 16 x 16 x 3: 8.8 | 18 x 18 x 4: 9.5 | 8 x 8 x 8: 12.4.  The rays
// of a wave are at (nearly) the same height at the same step, so a few levels suffice, and every KB of window costs resident waves
// (18 KB: six waves per CU where the registers allow twelve).
#ifndef FW
#define FW 14
#endif
#ifndef FWZ
#define FWZ 4
#endif
struct ScatterWindow {
    double *win;            // [FW][FW][FWZ] in LDS, this wave's
    int i0, j0, k0;         // first node of the window
    __device__ __forceinline__ void init(double *w) {
        win = w, i0 = j0 = k0 = 0;
        for (int t = threadIdx.x & 63; t < FW * FW * FWZ; t += 64) win[t] = 0.0;
    }
    __device__ __forceinline__ void flush(const GridView &g, double *__restrict__ G) {
        for (int t = threadIdx.x & 63; t < FW * FW * FWZ; t += 64) {
            const double v = win[t];
            if (v != 0.0) {
                const int m = t % FWZ, ab = t / FWZ, b = ab % FW, a = ab / FW;
                atomicAdd(G + ((size_t)(i0 + a) * g.ny + (j0 + b)) * g.nz + (k0 + m), v);
                win[t] = 0.0;
            }
        }
    }
    // every lane of the wave calls this together; `active` lanes add c x (trilinear weights) at cell (i, j, k), weights (tx, ty, tz)
    __device__ __forceinline__ void add(const GridView &g, double *__restrict__ G, bool active, int i, int j, int k, double tx, double ty,
                                        double tz, double c) {
        auto inside = [&]() {
            return !active || (((unsigned)(i - i0) <= (unsigned)(FW - 2)) & ((unsigned)(j - j0) <= (unsigned)(FW - 2)) &
                               ((unsigned)(k - k0) <= (unsigned)(FWZ - 2)));
        };
        if (!__all(inside())) {                              // (wave-uniform) move the window under the wave's current samples
            flush(g, G);
            const int big = 0x3fffffff;
            i0 = max(0, wave_minmax_i32<false>(active ? i : big) - 1);
            j0 = max(0, wave_minmax_i32<false>(active ? j : big) - 1);
            k0 = max(0, wave_minmax_i32<false>(active ? k : big));
            i0 = min(i0, max(0, g.nx - FW)), j0 = min(j0, max(0, g.ny - FW)), k0 = min(k0, max(0, g.nz - FWZ));
        }
        if (!active) return;
        const double w0 = c * (1 - tx), w1 = c * tx;
        const double w00 = w0 * (1 - ty), w01 = w0 * ty, w10 = w1 * (1 - ty), w11 = w1 * ty;
        if (inside()) {
            double *p = win + ((i - i0) * FW + (j - j0)) * FWZ + (k - k0);
            atomicAdd(p, w00 * (1 - tz)), atomicAdd(p + 1, w00 * tz);
            atomicAdd(p + FWZ, w01 * (1 - tz)), atomicAdd(p + FWZ + 1, w01 * tz);
            atomicAdd(p + FW * FWZ, w10 * (1 - tz)), atomicAdd(p + FW * FWZ + 1, w10 * tz);
            atomicAdd(p + (FW + 1) * FWZ, w11 * (1 - tz)), atomicAdd(p + (FW + 1) * FWZ + 1, w11 * tz);
        } else {
            double *p = G + ((size_t)i * g.ny + j) * g.nz + k;
            const size_t sj = g.nz, si = (size_t)g.ny * g.nz;
            atomicAdd(p, w00 * (1 - tz)), atomicAdd(p + 1, w00 * tz);
            atomicAdd(p + sj, w01 * (1 - tz)), atomicAdd(p + sj + 1, w01 * tz);
            atomicAdd(p + si, w10 * (1 - tz)), atomicAdd(p + si + 1, w10 * tz);
            atomicAdd(p + si + sj, w11 * (1 - tz)), atomicAdd(p + si + sj + 1, w11 * tz);
        }
    }
};

// KN: interpolant of the refractive index (the tracer's right-hand side); kne (run-time): interpolant of the integrand.
// ADJ = false: tec[r] = ne_scale * sum_k c_k ne(x_k);  ADJ = true: G += ne_scale * w[r] * c_k * (interpolation weights at x_k).
// Lanes = rays; samples that leave the grid are skipped and flagged, as in k_forward_rays / k_adjoint_rays.
// (the trilinear-index FORWARD is built for four waves per SIMD -- 127 VGPRs, 0 B of scratch: 620 000 rays 6.1 -> 5.6 ms, half of its
//  wave cycles wait on memory; the same cap costs the transpose 44 B of scratch in its scatter: 8.2 -> 10.0 ms, left alone)
#define FT_WPE_FOR(KN, ADJ) ((KN) == 0 && !(ADJ) ? 4 : 1), ((KN) == 0 && !(ADJ) ? 4 : 8)
template <int KN, bool BEND, bool ADJ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FT_WPE_FOR(KN, ADJ)))) void k_fermat_tec(GridView g, const double *__restrict__ nM, const double *__restrict__ origins,
                                                   const double *__restrict__ dirs, int64_t R, double tmax, int Ns, int substeps, int rule,
                                                   int stype, int kne, double ne_scale, const double *__restrict__ wray,
                                                   double *__restrict__ tec, double *__restrict__ G, int *oob_flag, double ztop) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const Axes ax = stage_axes(g, lds);
    const bool windowed = ADJ && kne == IONO_INTERP_TRILINEAR && g.nx >= FW && g.ny >= FW && g.nz >= FWZ;     // (wave-uniform)
    ScatterWindow sw;
    if (windowed) sw.init(lds + ((g.nx + g.ny + g.nz + 1) & ~1));
    const int64_t r_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // the windowed transpose is a wave operation: lanes beyond R / with zero weight walk ray min(r, R-1) and contribute nothing
    const bool lane_on = r_raw < R && (!ADJ || wray[r_raw] != 0.0);
    if (!windowed && !lane_on) return;
    const int64_t r = r_raw < R ? r_raw : R - 1;
    const double wr = ADJ ? wray[r] * ne_scale : 0.0;
    const double dx = dirs[3 * r], dy = dirs[3 * r + 1], dz = dirs[3 * r + 2];
    const double nrm = sqrt(dx * dx + dy * dy + dz * dz);
    FState u;
    u.px = dx / nrm, u.py = dy / nrm, u.pz = dz / nrm;
    u.x = origins[3 * r], u.y = origins[3 * r + 1], u.z = origins[3 * r + 2];
    u.s = 0.0;
    const double h = fermat_step(tmax, u.z, Ns, substeps, stype);
    // (ztop = last level + 1e-9 |tmax|, from the host: a kernel argument stays in scalar registers; computed here it was a vector
    //  register pair the four-waves build spilled)
    bool oob = false;
    // value of the integrand (forward) or "inside" marker (adjoint) of the samples in the window
    const bool ideal_lin = g.ideal && kne == IONO_INTERP_TRILINEAR;       // (wave-uniform) no axis tables for the integrand either
    auto inside = [&](const FState &p) {
        if (ideal_lin)
            return p.x >= g.g0[0] && p.x <= g.glast[0] && p.y >= g.g0[1] && p.y <= g.glast[1] && p.z >= g.g0[2] && p.z <= g.glast[2];
        return !(kne == IONO_INTERP_TRILINEAR ? sample_outside<IONO_INTERP_TRILINEAR>(ax, p.x, p.y, p.z)
                                               : sample_outside<IONO_INTERP_TRICUBIC>(ax, p.x, p.y, p.z));
    };
    auto value = [&](const FState &p) {
        if (ideal_lin) return trilinear_ideal(g, (const double *)g.M, p.x, p.y, p.z);
        return kne == IONO_INTERP_TRILINEAR ? sample_at<double, IONO_INTERP_TRILINEAR>(g, ax, p.x, p.y, p.z)
                                            : sample_at<double, IONO_INTERP_TRICUBIC>(g, ax, p.x, p.y, p.z);
    };
    // (`on`: this lane has something to add; in the windowed form every lane of the wave makes the call)
    auto scatter = [&](const FState &p, double wgt, bool on) {
        if (windowed) {
            int i = 0, j = 0, k = 0;
            double tx = 0, ty = 0, tz = 0;
            if (on && g.ideal) {          // cell and weights as trilinear_ideal forms them (the transpose of that very interpolation)
                const double ux = (p.x - g.g0[0]) * g.inv_h[0], uy = (p.y - g.g0[1]) * g.inv_h[1], uz = (p.z - g.g0[2]) * g.inv_h[2];
                const double fi = fmin(__builtin_floor(__builtin_fabs(ux)), (double)(g.nx - 2)), fj = fmin(__builtin_floor(__builtin_fabs(uy)), (double)(g.ny - 2)),
                             fk = fmin(__builtin_floor(__builtin_fabs(uz)), (double)(g.nz - 2));
                i = (int)fi, j = (int)fj, k = (int)fk;
                tx = ux - fi, ty = uy - fj, tz = uz - fk;
            } else if (on) {
                i = find_cell(ax.x, ax.nx, p.x, g.inv_h[0], g.uniform[0]), j = find_cell(ax.y, ax.ny, p.y, g.inv_h[1], g.uniform[1]);
                k = find_cell(ax.z, ax.nz, p.z, g.inv_h[2], g.uniform[2]);
                tx = (p.x - ax.x[i]) / (ax.x[i + 1] - ax.x[i]), ty = (p.y - ax.y[j]) / (ax.y[j + 1] - ax.y[j]);
                tz = (p.z - ax.z[k]) / (ax.z[k + 1] - ax.z[k]);
            }
            sw.add(g, G, on, i, j, k, tx, ty, tz, wgt);
        } else if (on) {
            if (kne == IONO_INTERP_TRILINEAR) scatter_trilinear<double>(g, ax, G, p.x, p.y, p.z, wgt);
            else scatter_tricubic<double>(g, ax, G, p.x, p.y, p.z, wgt);
        }
    };
    StreamQuad q;
    q.init(Ns, rule, 0.0);
    FState p0 = u, p1 = u;                 // positions of samples k-2, k-1 (adjoint)
    double y0 = 0.0, y1 = 0.0;             // integrand at samples k-2, k-1 (forward)
    bool in0 = false, in1 = inside(u);
    if (!in1) oob = true;
    if (!ADJ && in1) y1 = value(u);
    double acc = 0.0;
    for (int k = 1; k < Ns; ++k) {
        for (int sub = 0; sub < substeps; ++sub) {
            FState kprev = {}, sum = {};
            if (KN == IONO_INTERP_TRILINEAR) {
                // the four stages spelled out (same operations in the same order as the loop below: the loop only exists because
                // four inlined tricubic evaluations do not fit the register file)
                kprev = fermat_rhs<KN, BEND>(g, nM, axpy(u, 0.0, kprev), stype);
                sum = axpy(sum, 1.0, kprev);
                kprev = fermat_rhs<KN, BEND>(g, nM, axpy(u, 0.5 * h, kprev), stype);
                sum = axpy(sum, 2.0, kprev);
                kprev = fermat_rhs<KN, BEND>(g, nM, axpy(u, 0.5 * h, kprev), stype);
                sum = axpy(sum, 2.0, kprev);
                kprev = fermat_rhs<KN, BEND>(g, nM, axpy(u, h, kprev), stype);
                sum = axpy(sum, 1.0, kprev);
            } else {
#pragma unroll 1
                for (int st = 0; st < 4; ++st) {
                    const double ca = st == 0 ? 0.0 : (st == 3 ? h : 0.5 * h);
                    kprev = fermat_rhs<KN, BEND>(g, nM, axpy(u, ca, kprev), stype);
                    sum = axpy(sum, (st == 1 || st == 2) ? 2.0 : 1.0, kprev);
                }
            }
            u = axpy(u, h / 6.0, sum);
        }
        oob |= outside(ax.x, ax.nx, u.x) || outside(ax.y, ax.ny, u.y) || !(u.z >= ax.z[0] && u.z <= ztop);
        const bool in2 = inside(u);
        if (!in2) oob = true;
        const double y2 = (!ADJ && in2) ? value(u) : 0.0;
        const double wk = q.feed(k, u.s);                       // final weight of sample k-2
        if (ADJ) {
            if (k >= 2) scatter(p0, wr * wk, in0 && lane_on);
        } else if (k >= 2 && in0) {
            acc = fma(wk, y0, acc);
        }
        p0 = p1, p1 = u, y0 = y1, y1 = y2, in0 = in1, in1 = in2;
    }
    // the last two samples (the window after the final shift: k-2 = Ns-2, k-1 = Ns-1)
    if (ADJ) {
        if (Ns >= 2) scatter(p0, wr * q.w0, in0 && lane_on);
        scatter(p1, wr * q.w1, in1 && lane_on);
        if (windowed) sw.flush(g, G);
    } else {
        if (Ns >= 2 && in0) acc = fma(q.w0, y0, acc);
        if (in1) acc = fma(q.w1, y1, acc);
        // (the ray index again from the lane id -- one wave per workgroup -- instead of a register pair held, or spilled, for the whole walk)
        tec[(int64_t)blockIdx.x * 64 + __lane_id()] = acc * ne_scale;
    }
    if (oob && lane_on) atomicOr(oob_flag, 1);
}

// ---- fused curved-ray TEC through a TRICUBIC refractive index on ideal-uniform grids: 8 lanes per ray ---------------------------------
// k_fermat_tec<tricubic> steps lanes = rays with 216 taps per right-hand side out of line: 31 ms at config 3 against 1.13 ms for
// k_trace_fermat_lm + k_forward_rays, which is why curved rays through a tricubic index used to be traced into rays[R][4][Ns] (5.1 GB at
// 620 000 rays) and integrated in a second launch.  Here the record-per-lane stepper of k_trace_fermat_lm (same right-hand side, same
// order of operations: the samples are bit-identical to the traced ones) feeds the streaming quadrature directly.  The integrand at
// a sample is shared by the ray's eight lanes: trilinear -- every lane one corner; tricubic -- every lane a 3 x 3 x 3 block of the
// 6 x 6 x 6 taps (a, b, c = the lane's bits) -- summed with three DPP steps (sum8).  No ray tensor, any batch size.
// LPR (round 5): lanes per ray -- 8 as above, or fewer lanes that own 8 / LPR nodes of the cell each (fermat_rhs_lmn, iono_aux_kernels.h):
// the launch is bound by vector-instruction issue, and most of what the eight lanes of a ray issue is the same work eight times.
// ADJ (round 5): the transpose on the same stepper -- G += ne_scale * w[r] * c_k * (interpolation weights at x_k), the samples
// bit-identical to the forward's.  Trilinear integrand: the first lane of a ray adds its eight corners into the wave's ScatterWindow
// (all lanes make the call); tricubic integrand: every lane adds its 3 x 3 x 3 blocks of the 6 x 6 x 6 taps with global atomics.
// Launched with every lane group live (rays_per_wave = 64 / LPR): the window is a wave operation.
template <bool BEND, int LPR = 8, bool ADJ = false>
__global__ __launch_bounds__(64) void k_fermat_tec_lm(GridView g, const double *__restrict__ F8, const double *__restrict__ origins,
                                                      const double *__restrict__ dirs, int64_t R, double tmax, int Ns, int substeps, int rule,
                                                      int stype, int kne, double ne_scale, double *__restrict__ tec, int *oob_flag,
                                                      int rays_per_wave, const double *__restrict__ wray, double *__restrict__ G) {
    if (!ADJ && (int)(threadIdx.x / LPR) >= rays_per_wave) return;
    constexpr int NPL = 8 / LPR;                // nodes of a cell per lane: node (a, b, c) = the bits of sub * NPL + n
    const int sub = threadIdx.x & (LPR - 1);
    int64_t r = (int64_t)blockIdx.x * rays_per_wave + (threadIdx.x / LPR);
    const bool live = r < R;
    if (!live) r = R - 1;                      // idle groups shadow the last ray (DPP needs all lanes running)
    const double dx = dirs[3 * r], dy = dirs[3 * r + 1], dz = dirs[3 * r + 2];
    const double nrm = sqrt(dx * dx + dy * dy + dz * dz);
    FState u;
    u.px = dx / nrm, u.py = dy / nrm, u.pz = dz / nrm;
    u.x = origins[3 * r], u.y = origins[3 * r + 1], u.z = origins[3 * r + 2];
    u.s = 0.0;
    const double h = fermat_step(tmax, u.z, Ns, substeps, stype);
    const double *M = (const double *)g.M;
    const size_t sj = (size_t)g.nz, si = (size_t)g.ny * g.nz;
    const bool cubic = kne == IONO_INTERP_TRICUBIC;                    // (wave-uniform)
    // a sample is integrated if it lies inside the integrand's domain (the whole grid / the tricubic domain g[2] .. g[n-3]), as
    // k_forward_rays decides it; samples outside are skipped and flagged
    auto inside = [&](const FState &p) {
        if (cubic) return p.x >= g.c0[0] && p.x <= g.clast[0] && p.y >= g.c0[1] && p.y <= g.clast[1] && p.z >= g.c0[2] && p.z <= g.clast[2];
        return p.x >= g.g0[0] && p.x <= g.glast[0] && p.y >= g.g0[1] && p.y <= g.glast[1] && p.z >= g.g0[2] && p.z <= g.glast[2];
    };
    auto value = [&](const FState &p) {          // called by all the lanes of a ray with the same p
        const double ux = (p.x - g.g0[0]) * g.inv_h[0], uy = (p.y - g.g0[1]) * g.inv_h[1], uz = (p.z - g.g0[2]) * g.inv_h[2];
        if (!cubic) {
            const double fi = fmin(__builtin_floor(__builtin_fabs(ux)), (double)(g.nx - 2)), fj = fmin(__builtin_floor(__builtin_fabs(uy)), (double)(g.ny - 2)),
                         fk = fmin(__builtin_floor(__builtin_fabs(uz)), (double)(g.nz - 2));
            const double tx = ux - fi, ty = uy - fj, tz = uz - fk;
            double f = 0.0;
#pragma unroll
            for (int n = 0; n < NPL; ++n) {
                const int nn = sub * NPL + n, la = nn >> 2, lb = (nn >> 1) & 1, lc = nn & 1;
                const double v = M[((size_t)((int)fi + la) * g.ny + (size_t)((int)fj + lb)) * g.nz + (size_t)((int)fk + lc)];
                f += v * (la ? tx : 1.0 - tx) * (lb ? ty : 1.0 - ty) * (lc ? tz : 1.0 - tz);
            }
            return sum_lanes<LPR>(f);
        }
        const double fi = fmin(fmax(__builtin_floor(ux), 2.0), (double)(g.nx - 4)), fj = fmin(fmax(__builtin_floor(uy), 2.0), (double)(g.ny - 4)),
                     fk = fmin(fmax(__builtin_floor(uz), 2.0), (double)(g.nz - 4));
        double wx[6], wy[6], wz[6];
        cubic_taps_ideal(ux - fi, wx);
        cubic_taps_ideal(uy - fj, wy);
        cubic_taps_ideal(uz - fk, wz);
        double f = 0.0;
#pragma unroll
        for (int n = 0; n < NPL; ++n) {
            const int nn = sub * NPL + n, la = nn >> 2, lb = (nn >> 1) & 1, lc = nn & 1;
            // this node's half of each axis' six taps: a 3 x 3 x 3 block of the 6 x 6 x 6 (selects, not indexed arrays: no scratch)
            const double x3[3] = {la ? wx[3] : wx[0], la ? wx[4] : wx[1], la ? wx[5] : wx[2]};
            const double y3[3] = {lb ? wy[3] : wy[0], lb ? wy[4] : wy[1], lb ? wy[5] : wy[2]};
            const double z3[3] = {lc ? wz[3] : wz[0], lc ? wz[4] : wz[1], lc ? wz[5] : wz[2]};
            const double *base = M + ((size_t)((int)fi - 2 + 3 * la) * g.ny + (size_t)((int)fj - 2 + 3 * lb)) * g.nz + (size_t)((int)fk - 2 + 3 * lc);
            double fn = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                double fa = 0.0;
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const double *q = base + (size_t)a * si + (size_t)b * sj;
                    fa += (q[0] * z3[0] + q[1] * z3[1] + q[2] * z3[2]) * y3[b];
                }
                fn += fa * x3[a];
            }
            f += fn;
        }
        return sum_lanes<LPR>(f);
    };
    bool oob = false;
    int ci = -1, cj = -1, ck = -1;
    double rec[NPL * 8] = {};
    const double ztop = g.glast[2] + 1e-9 * fabs(tmax);
    StreamQuad q;
    q.init(Ns, rule, 0.0);
    auto clamped = [&](const FState &p, bool in) {       // (a sample that is skipped is still EVALUATED, at a position clamped into the grid)
        FState uc = p;
        if (!in) uc.x = fmin(fmax(p.x, g.c0[0]), g.clast[0]), uc.y = fmin(fmax(p.y, g.c0[1]), g.clast[1]), uc.z = fmin(fmax(p.z, g.c0[2]), g.clast[2]);
        return uc;
    };
    auto step = [&]() {
        for (int s2 = 0; s2 < substeps; ++s2) {
            FState kprev = {}, sum = {};
#pragma unroll 1
            for (int st = 0; st < 4; ++st) {
                const double ca = st == 0 ? 0.0 : (st == 3 ? h : 0.5 * h);
                kprev = fermat_rhs_lm_any<BEND, LPR>(g, F8, axpy(u, ca, kprev), sub, ci, cj, ck, rec, stype);
                sum = axpy(sum, (st == 1 || st == 2) ? 2.0 : 1.0, kprev);
            }
            u = axpy(u, h / 6.0, sum);
        }
        oob |= !(u.x >= g.g0[0] && u.x <= g.glast[0] && u.y >= g.g0[1] && u.y <= g.glast[1] && u.z >= g.g0[2] && u.z <= ztop);
    };
    bool in0 = false, in1 = inside(u);
    if (!in1) oob = true;
    if constexpr (ADJ) {
        __shared__ __attribute__((aligned(16))) double window[FW * FW * FWZ];
        const bool windowed = !cubic && g.nx >= FW && g.ny >= FW && g.nz >= FWZ;             // (wave-uniform)
        ScatterWindow sw;
        if (windowed) sw.init(window);
        const double wr = wray[r] * ne_scale;
        const bool ray_on = live && wr != 0.0;
        auto scatter = [&](const FState &p, double wgt, bool on) {         // every lane of the wave makes the call
            const double ux = (p.x - g.g0[0]) * g.inv_h[0], uy = (p.y - g.g0[1]) * g.inv_h[1], uz = (p.z - g.g0[2]) * g.inv_h[2];
            if (!cubic) {                // cell and weights as value() forms them; the ray's first lane adds the eight corners
                const double fi = fmin(__builtin_floor(__builtin_fabs(ux)), (double)(g.nx - 2)), fj = fmin(__builtin_floor(__builtin_fabs(uy)), (double)(g.ny - 2)),
                             fk = fmin(__builtin_floor(__builtin_fabs(uz)), (double)(g.nz - 2));
                const bool mine = on && sub == 0;
                if (windowed) {
                    sw.add(g, G, mine, (int)fi, (int)fj, (int)fk, ux - fi, uy - fj, uz - fk, wgt);
                } else if (mine) {
                    const double tx = ux - fi, ty = uy - fj, tz = uz - fk;
                    double *pp = G + ((size_t)(int)fi * g.ny + (size_t)(int)fj) * g.nz + (size_t)(int)fk;
#pragma unroll
                    for (int nn = 0; nn < 8; ++nn) {
                        const int la = nn >> 2, lb = (nn >> 1) & 1, lc = nn & 1;
                        atomicAdd(pp + la * si + lb * sj + lc, wgt * (la ? tx : 1.0 - tx) * (lb ? ty : 1.0 - ty) * (lc ? tz : 1.0 - tz));
                    }
                }
                return;
            }
            if (!on) return;
            const double fi = fmin(fmax(__builtin_floor(ux), 2.0), (double)(g.nx - 4)), fj = fmin(fmax(__builtin_floor(uy), 2.0), (double)(g.ny - 4)),
                         fk = fmin(fmax(__builtin_floor(uz), 2.0), (double)(g.nz - 4));
            double wx[6], wy[6], wz[6];
            cubic_taps_ideal(ux - fi, wx);
            cubic_taps_ideal(uy - fj, wy);
            cubic_taps_ideal(uz - fk, wz);
#pragma unroll
            for (int n = 0; n < NPL; ++n) {
                const int nn = sub * NPL + n, la = nn >> 2, lb = (nn >> 1) & 1, lc = nn & 1;
                const double x3[3] = {la ? wx[3] : wx[0], la ? wx[4] : wx[1], la ? wx[5] : wx[2]};
                const double y3[3] = {lb ? wy[3] : wy[0], lb ? wy[4] : wy[1], lb ? wy[5] : wy[2]};
                const double z3[3] = {lc ? wz[3] : wz[0], lc ? wz[4] : wz[1], lc ? wz[5] : wz[2]};
                double *base = G + ((size_t)((int)fi - 2 + 3 * la) * g.ny + (size_t)((int)fj - 2 + 3 * lb)) * g.nz + (size_t)((int)fk - 2 + 3 * lc);
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        double *qq = base + (size_t)a * si + (size_t)b * sj;
                        const double wab = wgt * x3[a] * y3[b];
                        atomicAdd(qq, wab * z3[0]), atomicAdd(qq + 1, wab * z3[1]), atomicAdd(qq + 2, wab * z3[2]);
                    }
            }
        };
        FState p0 = u, p1 = u;                 // positions of samples k-2, k-1
        for (int k = 1; k < Ns; ++k) {
            step();
            const bool in2 = inside(u);
            if (!in2) oob = true;
            const double wk = q.feed(k, u.s);                       // final weight of sample k-2
            if (k >= 2) scatter(p0, wr * wk, in0 && ray_on);
            p0 = p1, p1 = u, in0 = in1, in1 = in2;
        }
        if (Ns >= 2) scatter(p0, wr * q.w0, in0 && ray_on);
        scatter(p1, wr * q.w1, in1 && ray_on);
        if (windowed) sw.flush(g, G);
        if (live && sub == 0 && oob) atomicOr(oob_flag, 1);
    } else {
        double y0 = 0.0, y1 = 0.0;             // integrand at samples k-2, k-1
        {       // (all lanes run the DPP sums)
            const double yv = value(clamped(u, in1));
            y1 = in1 ? yv : 0.0;
        }
        double acc = 0.0;
        for (int k = 1; k < Ns; ++k) {
            step();
            const bool in2 = inside(u);
            if (!in2) oob = true;
            const double yv = value(clamped(u, in2));
            const double y2 = in2 ? yv : 0.0;
            const double wk = q.feed(k, u.s);                       // final weight of sample k-2
            if (k >= 2 && in0) acc = fma(wk, y0, acc);
            y0 = y1, y1 = y2, in0 = in1, in1 = in2;
        }
        if (Ns >= 2 && in0) acc = fma(q.w0, y0, acc);
        if (in1) acc = fma(q.w1, y1, acc);
        if (live && sub == 0) {
            tec[r] = acc * ne_scale;
            if (oob) atomicOr(oob_flag, 1);
        }
    }
}

}  // namespace

#endif
