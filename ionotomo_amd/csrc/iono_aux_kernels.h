// elementwise kernels, C_m smoothing, point interpolation, straight / Fermat ray tracers
#ifndef IONO_AUX_KERNELS_H
#define IONO_AUX_KERNELS_H

namespace {

// ------------------------------------------------------------------------------------------------
// small elementwise / geometry kernels
// ------------------------------------------------------------------------------------------------
template <typename GT>
struct SetValues {                 // dst = (GT)src, or (GT)(exp(src) * scale); flags NaN / Inf
    const double *__restrict__ src;
    GT *__restrict__ dst;
    int do_exp;
    double scale;
    int *nonfinite;
    __device__ __forceinline__ void operator()(int64_t i) const {
        double v = src[i];
        if (do_exp) v = exp(v) * scale;
        if (!isfinite(v)) atomicOr(nonfinite, 1);
        dst[i] = (GT)v;
    }
};

template <typename GT>
struct GetValues {
    const GT *__restrict__ src;
    double *__restrict__ dst;
    __device__ __forceinline__ void operator()(int64_t i) const { dst[i] = (double)src[i]; }
};

// n = sqrt(1 - 8.980^2 ne / nu^2) at the nodes (inversion/fermat.py:36-46)
template <typename GT>
struct NeToN {
    const GT *__restrict__ ne;
    double *__restrict__ nM;
    double freq;
    __device__ __forceinline__ void operator()(int64_t i) const { nM[i] = sqrt(1.0 + (double)ne[i] * (-PLASMA_A / (freq * freq))); }
};

template <typename AT, typename GT>
struct ScaleByGrid {
    AT *__restrict__ G;
    const GT *__restrict__ M;
    __device__ __forceinline__ void operator()(int64_t i) const { G[i] = (AT)((double)G[i] * (double)M[i]); }
};

struct SubtractReference {         // over Na * NtNd: rows other than i0 (they read row i0); row i0 is zeroed by a memset afterwards
    double *__restrict__ tec;
    int64_t NtNd;
    int i0;
    __device__ __forceinline__ void operator()(int64_t idx) const {
        const int a = idx / NtNd;
        if (a != i0) tec[idx] -= tec[(int64_t)i0 * NtNd + idx % NtNd];
    }
};
// y = a x + b y with a = sa * a_num / a_den, b = b_num / b_den read from DEVICE scalars (null pointer = 1): the
// solvers' step lengths are ratios of all-reduced dot products that never visit the host.  One pass, 16 B per lane.
__global__ __launch_bounds__(256) void k_axpby(double *__restrict__ y, const double *__restrict__ x, int64_t n,
                                               const double *a_num, const double *a_den, double sa, const double *b_num,
                                               const double *b_den) {
    const double a = sa * (a_num ? *a_num : 1.0) / (a_den ? *a_den : 1.0);
    const double b = (b_num ? *b_num : 1.0) / (b_den ? *b_den : 1.0);
    const int64_t n2 = n >> 1, stride = (int64_t)gridDim.x * blockDim.x;
    double2 *y2 = (double2 *)y;
    const double2 *x2 = (const double2 *)x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const double2 xv = x2[i];
        double2 yv = y2[i];
        yv.x = fma(a, xv.x, b * yv.x);
        yv.y = fma(a, xv.y, b * yv.y);
        y2[i] = yv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = fma(a, x[n - 1], b * y[n - 1]);
}

// C_m smoothing: Covariance.smooth = scipy.ndimage.convolve(phi, c_stencil, mode='nearest')
// (ionosphere/covariance.py:46-63,383-385).  The reference's stencil is the product of three 1-D
// exponential kernels, so the (2h+1)^3 convolution is three 1-D passes with edge replication.
// Lanes run along z (contiguous) in every pass; taps along x / y are whole coalesced rows.
#define CONV_T 32          // outputs along the filtered axis per workgroup (x / y passes)
// x / y pass: a workgroup owns 64 consecutive z (lanes) x CONV_T outputs along the axis for one value
// of the third index; the CONV_T + 2h input rows (edge rows replicated) are staged in LDS once, then
// every thread produces CONV_T/4 outputs from LDS (2h+1 taps each).  Global loads per output:
// (CONV_T + 2h) / CONV_T, all coalesced 512-B rows.
template <int AXIS>
__global__ __launch_bounds__(256) void k_conv_xy(const double *__restrict__ in, double *__restrict__ out, int nx, int ny, int nz,
                                                 const double *__restrict__ w, int h) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *wl = sm;                      // [2h+1]
    double *tile = sm + ((2 * h + 2) & ~1);    // [CONV_T + 2h][64]
    const int m = 2 * h + 1, rows = CONV_T + 2 * h;
    for (int t = threadIdx.x; t < m; t += blockDim.x) wl[t] = w[t];
    const int na = AXIS == 0 ? nx : ny;               // filtered axis
    const int no = AXIS == 0 ? ny : nx;               // the other non-z axis
    const int64_t sa = AXIS == 0 ? (int64_t)ny * nz : nz, so = AXIS == 0 ? nz : (int64_t)ny * nz;
    const int zt = (nz + 63) / 64, at = (na + CONV_T - 1) / CONV_T;
    const int64_t ntile = (int64_t)zt * at * no;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    for (int64_t tid = blockIdx.x; tid < ntile; tid += gridDim.x) {
        const int z0 = (int)(tid % zt) * 64;
        const int a0 = (int)((tid / zt) % at) * CONV_T;
        const int o = (int)(tid / ((int64_t)zt * at));
        const int k = z0 + lane;
        __syncthreads();
        if (k < nz) {
            const double *base = in + (int64_t)o * so + k;
            for (int r = grp; r < rows; r += 4) {
                const int a = min(max(a0 + r - h, 0), na - 1);
                tile[r * 64 + lane] = base[(int64_t)a * sa];
            }
        }
        __syncthreads();
        if (k < nz) {
            for (int q = grp; q < CONV_T; q += 4) {
                if (a0 + q >= na) break;
                double acc = 0.0;
                const double *tp = tile + q * 64 + lane;
                for (int t = 0; t < m; ++t) acc += wl[m - 1 - t] * tp[t * 64];
                out[(int64_t)o * so + (int64_t)(a0 + q) * sa + k] = acc;
            }
        }
    }
}

// z pass: each wave owns 64 consecutive z of one (i, j) row; the 64 + 2h inputs go through LDS
__global__ __launch_bounds__(256) void k_conv_z(const double *__restrict__ in, double *__restrict__ out, int nx, int ny, int nz,
                                                const double *__restrict__ w, int h) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *wl = sm;
    const int m = 2 * h + 1, span = 64 + 2 * h;
    double *tile = sm + ((2 * h + 2) & ~1) + (threadIdx.x >> 6) * span;     // per wave
    for (int t = threadIdx.x; t < m; t += blockDim.x) wl[t] = w[t];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int zt = (nz + 63) / 64;
    const int64_t nseg = (int64_t)nx * ny * zt;
    for (int64_t sid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); sid < nseg; sid += (int64_t)gridDim.x * 4) {
        const int z0 = (int)(sid % zt) * 64;
        const double *row = in + (sid / zt) * nz;
        for (int t = lane; t < span; t += 64) tile[t] = row[min(max(z0 + t - h, 0), nz - 1)];
        // same-wave LDS write -> read: the wave executes in lockstep, a waitcnt is all that is needed
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const int k = z0 + lane;
        if (k < nz) {
            double acc = 0.0;
            for (int t = 0; t < m; ++t) acc += wl[m - 1 - t] * tile[lane + t];
            out[(sid / zt) * nz + k] = acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename GT, int KIND, bool EXTRAP>
__global__ void k_interp_points(GridView g, const double *__restrict__ x, const double *__restrict__ y,
                                const double *__restrict__ z, int64_t n, double *__restrict__ out, int *oob_flag) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const Axes ax = stage_axes(g, lds);
    bool oob = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double px = x[i], py = y[i], pz = z[i];
        if (!EXTRAP && sample_outside<KIND>(ax, px, py, pz)) {
            oob = true;
            out[i] = nan("");
            continue;
        }
        out[i] = sample_at<GT, KIND>(g, ax, px, py, pz);
    }
    if (oob) atomicOr(oob_flag, 1);
}

// stype = 1: arc length is the independent variable (Fermat type='s', inversion/fermat.py:74-82,165-166):
// s = linspace(0, tmax, Ns), position = origin + p s
struct TraceStraight {             // over R * Ns samples
    const double *__restrict__ origins;
    const double *__restrict__ dirs;
    double tmax;
    int Ns, stype;
    double *__restrict__ rays;
    __device__ __forceinline__ void operator()(int64_t idx) const {
        const int64_t r = idx / Ns;
        const int k = idx % Ns;
        double *o = rays + (size_t)r * 4 * Ns;
        if (stype) {
            const double dx = dirs[3 * r], dy = dirs[3 * r + 1], dz = dirs[3 * r + 2];
            const double nrm = sqrt(dx * dx + dy * dy + dz * dz);
            const double sv = (k == Ns - 1) ? tmax : tmax * ((double)k * (1.0 / (double)(Ns - 1)));
            o[k] = origins[3 * r] + dx / nrm * sv;
            o[Ns + k] = origins[3 * r + 1] + dy / nrm * sv;
            o[2 * Ns + k] = origins[3 * r + 2] + dz / nrm * sv;
            o[3 * Ns + k] = sv;
            return;
        }
        const StraightRay q = load_straight(origins, dirs, r, tmax, Ns);
        double x, y, z;
        straight_point(q, k, Ns, x, y, z);
        o[k] = x;
        o[Ns + k] = y;
        o[2 * Ns + k] = z;
        const double frac = (k == Ns - 1) ? 1.0 : (double)k * q.step;
        o[3 * Ns + k] = q.L * frac / q.pz;     // s = (z - z0)/pz
    }
};

// Fermat ray ODE in z (inversion/fermat.py:64-72; notebooks/FermatClass.ipynb c0:76-84):
//   s' = n/pz, p' = grad(n) n/pz, x' = px/pz, y' = py/pz, z' = 1.   Lanes = rays, RK4.
struct FState {
    double px, py, pz, x, y, z, s;
};
// stype (wave-uniform): 0 = d/dz (type='z'), 1 = d/ds (type='s': s' = 1, p' = grad n, x' = p / n; fermat.py:74-82)
__device__ __forceinline__ FState fermat_rates(double n, double nx, double ny, double nz, const FState &u, int stype) {
    FState d;
    if (stype) {
        const double rn = 1.0 / n;
        d.px = nx, d.py = ny, d.pz = nz;
        d.x = u.px * rn, d.y = u.py * rn, d.z = u.pz * rn;
        d.s = 1.0;
        return d;
    }
    const double ipz = 1.0 / u.pz;              // one reciprocal instead of three divisions per stage
    const double f = n * ipz;
    d.px = nx * f, d.py = ny * f, d.pz = nz * f;
    d.x = u.px * ipz, d.y = u.py * ipz, d.z = 1.0;
    d.s = f;
    return d;
}
__device__ __forceinline__ double fermat_step(double tmax, double z0, int Ns, int substeps, int stype) {
    return (stype ? tmax : tmax - z0) / (double)((Ns - 1) * substeps);
}
// (out of line: inlined into the lanes = rays kernels -- four RK4 stages, in k_fermat_tec beside the integrand's own evaluation --
//  the 216-tap value-and-gradient evaluation pushed them to 512 VGPRs + 1 kB of scratch per lane.  Everything crosses the call BY
//  VALUE, in registers: with `const GridView &` in and four `double &` out -- round 4 -- the caller had to keep the 192-byte view and
//  the four results in scratch memory, 224 bytes per lane of every k_trace_fermat<1, *> / k_fermat_tec<1, *, *>)
struct NGrad {
    double n, nx, ny, nz;
};
__device__ __attribute__((noinline)) NGrad tricubic_n_and_gradient(const double *nM, const double *axes, int nx, int ny, int nz, double ih0,
                                                                   double ih1, double ih2, int u0, int u1, int u2, double x, double y, double z) {
    GridView gn;
    gn.axes = axes, gn.M = nM, gn.nx = nx, gn.ny = ny, gn.nz = nz;
    gn.inv_h[0] = ih0, gn.inv_h[1] = ih1, gn.inv_h[2] = ih2, gn.uniform[0] = u0, gn.uniform[1] = u1, gn.uniform[2] = u2;
    NGrad o;
    tricubic_eval<double, true>(gn, axes, axes + nx, axes + nx + ny, x, y, z, o.n, o.nx, o.ny, o.nz);
    return o;
}
template <int KIND, bool BEND>
__device__ __forceinline__ FState fermat_rhs(const GridView &g, const double *nM, const FState &u, int stype) {
    double n, nx, ny, nz;
    if (KIND == IONO_INTERP_TRILINEAR) {
        // ideal-uniform grid, point not on a cell face: no axis tables, no divisions (iono_device_common.h)
        if (!(g.ideal && trilinear_grad_ideal(g, nM, u.x, u.y, u.z, n, nx, ny, nz))) trilinear_grad_at(g, nM, u.x, u.y, u.z, n, nx, ny, nz);
    } else {
        const NGrad q = tricubic_n_and_gradient(nM, g.axes, g.nx, g.ny, g.nz, g.inv_h[0], g.inv_h[1], g.inv_h[2], g.uniform[0], g.uniform[1],
                                                g.uniform[2], u.x, u.y, u.z);
        n = q.n, nx = q.nx, ny = q.ny, nz = q.nz;
    }
    if (!BEND) nx = ny = nz = 0.0;
    return fermat_rates(n, nx, ny, nz, u, stype);
}
__device__ __forceinline__ FState axpy(const FState &u, double a, const FState &d) {
    FState r;
    r.px = u.px + a * d.px;
    r.py = u.py + a * d.py;
    r.pz = u.pz + a * d.pz;
    r.x = u.x + a * d.x;
    r.y = u.y + a * d.y;
    r.z = u.z + a * d.z;
    r.s = u.s + a * d.s;
    return r;
}
template <int KIND, bool BEND>
__global__ __launch_bounds__(64) void k_trace_fermat(GridView g, const double *__restrict__ nM,
                                                     const double *__restrict__ origins, const double *__restrict__ dirs,
                                                     int64_t R, double tmax, int Ns, int substeps, double *__restrict__ rays,
                                                     int *oob_flag, int stype) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const double dx = dirs[3 * r], dy = dirs[3 * r + 1], dz = dirs[3 * r + 2];
    const double nrm = sqrt(dx * dx + dy * dy + dz * dz);
    FState u;
    u.px = dx / nrm;
    u.py = dy / nrm;
    u.pz = dz / nrm;
    u.x = origins[3 * r];
    u.y = origins[3 * r + 1];
    u.z = origins[3 * r + 2];
    u.s = 0.0;
    const double h = fermat_step(tmax, u.z, Ns, substeps, stype);
    double *o = rays + (size_t)r * 4 * Ns;
    o[0] = u.x;
    o[Ns] = u.y;
    o[2 * Ns] = u.z;
    o[3 * Ns] = u.s;
    const double *gx = g.axes, *gy = g.axes + g.nx, *gz = g.axes + g.nx + g.ny;
    bool oob = false;
    for (int k = 1; k < Ns; ++k) {
        for (int sub = 0; sub < substeps; ++sub) {
            // classic RK4 with the four stages as a loop (one copy of the right-hand side: four inlined
            // tricubic evaluations need > 512 VGPRs and spill): sum = k1 + 2 k2 + 2 k3 + k4
            FState kprev = {}, sum = {};
#pragma unroll 1
            for (int st = 0; st < 4; ++st) {
                const double ca = st == 0 ? 0.0 : (st == 3 ? h : 0.5 * h);
                kprev = fermat_rhs<KIND, BEND>(g, nM, axpy(u, ca, kprev), stype);
                sum = axpy(sum, (st == 1 || st == 2) ? 2.0 : 1.0, kprev);
            }
            u = axpy(u, h / 6.0, sum);
        }
        oob |= outside(gx, g.nx, u.x) || outside(gy, g.ny, u.y) || !(u.z >= gz[0] && u.z <= gz[g.nz - 1] + 1e-9 * fabs(tmax));
        o[k] = u.x;
        o[Ns + k] = u.y;
        o[2 * Ns + k] = u.z;
        o[3 * Ns + k] = u.s;
    }
    if (oob) atomicOr(oob_flag, 1);
}

// ---- cooperative tricubic tracer: 8 lanes per ray -------------------------------------------------------
// With lanes = rays a 2,604-ray config is 41 waves, each lane serially gathering 216 nodes per RK4
// stage.  Here a ray is shared by 8 consecutive lanes: lane `sub` (< 6) owns x-tap `a = sub` and
// contracts its 6 x 6 (y, z) plane (z taps are 6 contiguous doubles per load group); the four partial
// results (n, nx, ny, nz) are summed over the 8 lanes with three DPP steps (quad_perm xor 1, xor 2,
// row_half_mirror).  Every lane keeps the full ray state, so no broadcast is needed.
// The add must see the ROUNDED value of v on both sides of the exchange: with FP contraction on, `v + other` where
// v = a * b in the caller became fma(a, b, other) -- the lane's own term unrounded, its partner's rounded -- so the
// two lanes of a pair got sums that differ in the last bit.  Every lane of a group integrates its own copy of the
// ray state, so the copies drifted apart (random walk, ~1e-12 km after 2,000 stages), the x-tap weights of the
// lanes then belonged to slightly different positions and sum_a dw_a no longer cancelled: a spurious gradient
// of ~1e-12 / km along the lane-split axis, 2e-7 km of bending error at config-3 size.  Contraction is off here.
template <int CTRL>
__device__ __forceinline__ double dpp_xadd(double v) {
#pragma clang fp contract(off)
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double sum8(double v) {
    v = dpp_xadd<0xB1>(v);      // quad_perm:[1,0,3,2]
    v = dpp_xadd<0x4E>(v);      // quad_perm:[2,3,0,1]
    return dpp_xadd<0x141>(v);  // row_half_mirror: lane i <-> 7 - i within each group of 8
}
__device__ __forceinline__ void pick_tap(const double w[6], const double dw[6], int a, double &wa, double &da) {
    wa = a == 0 ? w[0] : a == 1 ? w[1] : a == 2 ? w[2] : a == 3 ? w[3] : a == 4 ? w[4] : a == 5 ? w[5] : 0.0;
    da = a == 0 ? dw[0] : a == 1 ? dw[1] : a == 2 ? dw[2] : a == 3 ? dw[3] : a == 4 ? dw[4] : a == 5 ? dw[5] : 0.0;
}
// Per-axis constants of the current cell, kept while the ray stays inside it: the seven divisions per axis of
// cubic_axis (t, the two slope scales, the four derivative scalings) become multiplications by cached
// reciprocals (ulp-level differences), and the cell search is skipped.
struct CubicAxisCache {
    int ic, i;                  // found cell (ic = -1: nothing cached) and stencil centre i = clamp(ic, 2, n - 4)
    double lo, hi;              // g[ic] < x <= g[ic + 1]
    double gi, rh, c0, c1;      // g[i], 1 / (g[i+1] - g[i]), h / (6 (g[i+1] - g[i-1])), h / (6 (g[i+2] - g[i]))
};
__device__ __forceinline__ int cubic_axis_cached(const double *g, int n, double x, double inv_h, int uniform,
                                                 CubicAxisCache &cc, double w[6], double dw[6]) {
    if (cc.ic < 0 || !((x > cc.lo || cc.ic == 0) && x <= cc.hi)) {
        const int ic = find_cell(g, n, x, inv_h, uniform);
        const int i = min(max(ic, 2), n - 4);
        const double h = g[i + 1] - g[i];
        cc.ic = ic, cc.i = i, cc.lo = g[ic], cc.hi = g[ic + 1];
        cc.gi = g[i], cc.rh = 1.0 / h;
        cc.c0 = h / (6.0 * (g[i + 1] - g[i - 1]));
        cc.c1 = h / (6.0 * (g[i + 2] - g[i]));
    }
    const double t = (x - cc.gi) * cc.rh, c0 = cc.c0, c1 = cc.c1;
    const double t2 = t * t, t3 = t2 * t;
    const double b0 = 2 * t3 - 3 * t2 + 1, b1 = -2 * t3 + 3 * t2, b2 = t3 - 2 * t2 + t, b3 = t3 - t2;
    w[0] = b2 * c0;
    w[1] = -8.0 * b2 * c0 + b3 * c1;
    w[2] = b0 - 8.0 * b3 * c1;
    w[3] = b1 + 8.0 * b2 * c0;
    w[4] = -b2 * c0 + 8.0 * b3 * c1;
    w[5] = -b3 * c1;
    const double d0 = (6 * t2 - 6 * t) * cc.rh, d1 = -d0;
    const double d2 = (3 * t2 - 4 * t + 1) * cc.rh, d3 = (3 * t2 - 2 * t) * cc.rh;
    dw[0] = d2 * c0;
    dw[1] = -8.0 * d2 * c0 + d3 * c1;
    dw[2] = d0 - 8.0 * d3 * c1;
    dw[3] = d1 + 8.0 * d2 * c0;
    dw[4] = -d2 * c0 + 8.0 * d3 * c1;
    dw[5] = -d3 * c1;
    return cc.i;
}
struct StencilPlane {
    int i, j, k;                // stencil centre the values belong to (i = -1: nothing cached)
    double v[6][6];             // nM at (i - 2 + a, j - 2 + b, k - 2 + c) for this lane's x tap a
};
template <bool BEND>
__device__ __forceinline__ FState fermat_rhs_coop(const GridView &g, const double *gx, const double *gy, const double *gz,
                                                  const double *__restrict__ nM, const FState &u, int sub,
                                                  CubicAxisCache (&cc)[3], StencilPlane &nc, int stype) {
    double wx[6], wy[6], wz[6], dx[6], dy[6], dz[6];
    const int i = cubic_axis_cached(gx, g.nx, u.x, g.inv_h[0], g.uniform[0], cc[0], wx, dx);
    const int j = cubic_axis_cached(gy, g.ny, u.y, g.inv_h[1], g.uniform[1], cc[1], wy, dy);
    const int k = cubic_axis_cached(gz, g.nz, u.z, g.inv_h[2], g.uniform[2], cc[2], wz, dz);
    double wxa, dxa;
    pick_tap(wx, dx, sub, wxa, dxa);
    const int a = min(sub, 5);
    if (i != nc.i || j != nc.j || k != nc.k) {      // this lane's 6 x 6 (y, z) plane of the stencil: reload on a cell change
        const double *base = nM + ((size_t)(i - 2 + a) * g.ny + (j - 2)) * g.nz + (k - 2);
#pragma unroll
        for (int b = 0; b < 6; ++b)
#pragma unroll
            for (int c = 0; c < 6; ++c) nc.v[b][c] = base[(size_t)b * g.nz + c];
        nc.i = i, nc.j = j, nc.k = k;
    }
    double fa = 0.0, fya = 0.0, fza = 0.0;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        double sv = 0.0, sz = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double v = nc.v[b][c];
            sv += v * wz[c];
            sz += v * dz[c];
        }
        fa += sv * wy[b];
        fya += sv * dy[b];
        fza += sz * wy[b];
    }
    const double n = sum8(fa * wxa);
    double nx = sum8(fa * dxa), ny = sum8(fya * wxa), nz = sum8(fza * wxa);
    if (!BEND) nx = ny = nz = 0.0;
    return fermat_rates(n, nx, ny, nz, u, stype);
}
template <bool BEND>
__global__ __launch_bounds__(64) void k_trace_fermat_coop(GridView g, const double *__restrict__ nM,
                                                          const double *__restrict__ origins, const double *__restrict__ dirs,
                                                          int64_t R, double tmax, int Ns, int substeps, double *__restrict__ rays,
                                                          int *oob_flag, int axes_in_lds, int rays_per_wave, int stype) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const double *gx = g.axes, *gy = g.axes + g.nx, *gz = g.axes + g.nx + g.ny;
    if (axes_in_lds) {              // the cell search and the slope weights read 7 axis values per axis per stage
        const Axes ax = stage_axes(g, (double *)smem);
        gx = ax.x, gy = ax.y, gz = ax.z;
    }
    // fewer than 8 rays per wave leaves lanes idle on purpose (small batches): a wave reloads whenever ANY of its
    // rays changes cell
    if ((int)(threadIdx.x >> 3) >= rays_per_wave) return;
    const int sub = threadIdx.x & 7;
    int64_t r = (int64_t)blockIdx.x * rays_per_wave + (threadIdx.x >> 3);
    const bool live = r < R;
    if (!live) r = R - 1;                      // idle groups shadow the last ray (DPP needs all lanes running)
    const double dx = dirs[3 * r], dy = dirs[3 * r + 1], dz = dirs[3 * r + 2];
    const double nrm = sqrt(dx * dx + dy * dy + dz * dz);
    FState u;
    u.px = dx / nrm;
    u.py = dy / nrm;
    u.pz = dz / nrm;
    u.x = origins[3 * r];
    u.y = origins[3 * r + 1];
    u.z = origins[3 * r + 2];
    u.s = 0.0;
    const double h = fermat_step(tmax, u.z, Ns, substeps, stype);
    double *o = rays + (size_t)r * 4 * Ns;
    const bool writer = live && sub == 0;
    if (writer) {
        o[0] = u.x;
        o[Ns] = u.y;
        o[2 * Ns] = u.z;
        o[3 * Ns] = u.s;
    }
    bool oob = false;
    CubicAxisCache cc[3] = {};
    cc[0].ic = cc[1].ic = cc[2].ic = -1;
    StencilPlane nc = {};
    nc.i = nc.j = nc.k = -1;
    for (int k = 1; k < Ns; ++k) {
        for (int s2 = 0; s2 < substeps; ++s2) {
            FState kprev = {}, sum = {};
#pragma unroll 1
            for (int st = 0; st < 4; ++st) {
                const double ca = st == 0 ? 0.0 : (st == 3 ? h : 0.5 * h);
                kprev = fermat_rhs_coop<BEND>(g, gx, gy, gz, nM, axpy(u, ca, kprev), sub, cc, nc, stype);
                sum = axpy(sum, (st == 1 || st == 2) ? 2.0 : 1.0, kprev);
            }
            u = axpy(u, h / 6.0, sum);
        }
        oob |= outside(gx, g.nx, u.x) || outside(gy, g.ny, u.y) || !(u.z >= gz[0] && u.z <= gz[g.nz - 1] + 1e-9 * fabs(tmax));
        if (writer) {
            o[k] = u.x;
            o[Ns + k] = u.y;
            o[2 * Ns + k] = u.z;
            o[3 * Ns + k] = u.s;
        }
    }
    if (oob && writer) atomicOr(oob_flag, 1);
}

// ---- tricubic tracer on ideal-uniform grids: 8 lanes per ray, ONE NODE of the cell per lane (Lekien-Marsden records) ---------------
// k_trace_fermat_coop gives a lane one x-tap plane of the 6 x 6 x 6 stencil: 36 values, 90 multiply-adds and the three 6-tap weight
// sets per stage (250 vector instructions, which IS its time: a lone wave issues one instruction per ~10 cycles).  With the
// derivative records of the refractive index (F8[node][p + 2 q + 4 r] = Dx^p Dy^q Dz^r n: the forward kernels' layout, built from
// the n nodes) a cell is eight 64-byte records, one per lane (a, b, c) = bits of the lane's index in its group of eight:
//     n = sum_lanes sum_pqr Hx[a][p](tx) Hy[b][q](ty) Hz[c][r](tz) F[pqr],   dn/dx = the same with Hx' / h, ...
// -- 40 multiply-adds per lane and stage, the record reloaded when the cell changes.  Same interpolant as the 216-tap form to
// rounding (iono_cubic_kernels.h); it is C1, so the cell rule on faces is immaterial: cell = floor, clamped to the tricubic domain.
template <bool BEND>
__device__ __forceinline__ FState fermat_rhs_lm(const GridView &g, const double *__restrict__ F8, const FState &u, int sub, int &ci, int &cj,
                                                int &ck, double (&rec)[8], int stype) {
    const double ux = (u.x - g.g0[0]) * g.inv_h[0], uy = (u.y - g.g0[1]) * g.inv_h[1], uz = (u.z - g.g0[2]) * g.inv_h[2];
    const double fi = fmin(fmax(__builtin_floor(ux), 2.0), (double)(g.nx - 4)), fj = fmin(fmax(__builtin_floor(uy), 2.0), (double)(g.ny - 4)),
                 fk = fmin(fmax(__builtin_floor(uz), 2.0), (double)(g.nz - 4));
    const int i = (int)fi, j = (int)fj, k = (int)fk;
    const int a = sub >> 2, b = (sub >> 1) & 1, c = sub & 1;
    if ((i != ci) | (j != cj) | (k != ck)) {          // (the eight lanes of a ray hold the same state: they reload together)
        const double2 *p = (const double2 *)(F8 + ((size_t)(i + a) * LM_SI(g.ny, g.nz) + (size_t)(j + b) * LM_NZP(g.nz) + (size_t)(k + c)) * LM_NF);
        const double2 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
        rec[0] = r0.x, rec[1] = r0.y, rec[2] = r1.x, rec[3] = r1.y, rec[4] = r2.x, rec[5] = r2.y, rec[6] = r3.x, rec[7] = r3.y;
        ci = i, cj = j, ck = k;
    }
    // Hermite value / slope weights of THIS lane's node along each axis, and their t-derivatives
    auto axis = [](double t, int hi, double &w0, double &w1, double &d0, double &d1) {
        const double t2 = t * t, t3 = t2 * t;
        if (hi) {
            w0 = 3.0 * t2 - 2.0 * t3, w1 = t3 - t2;                 // node 1: value weight h1, slope weight s1
            d0 = 6.0 * t - 6.0 * t2, d1 = 3.0 * t2 - 2.0 * t;
        } else {
            w0 = 1.0 - (3.0 * t2 - 2.0 * t3), w1 = t3 - 2.0 * t2 + t;      // node 0: h0, s0
            d0 = 6.0 * t2 - 6.0 * t, d1 = 3.0 * t2 - 4.0 * t + 1.0;
        }
    };
    double X0, X1, dX0, dX1, Y0, Y1, dY0, dY1, Z0, Z1, dZ0, dZ1;
    axis(ux - fi, a, X0, X1, dX0, dX1);
    axis(uy - fj, b, Y0, Y1, dY0, dY1);
    axis(uz - fk, c, Z0, Z1, dZ0, dZ1);
    // contract p (x), then q (y), then r (z): rec[p + 2 q + 4 r]
    const double g00 = X0 * rec[0] + X1 * rec[1], g10 = X0 * rec[2] + X1 * rec[3], g01 = X0 * rec[4] + X1 * rec[5], g11 = X0 * rec[6] + X1 * rec[7];
    const double x00 = dX0 * rec[0] + dX1 * rec[1], x10 = dX0 * rec[2] + dX1 * rec[3], x01 = dX0 * rec[4] + dX1 * rec[5],
                 x11 = dX0 * rec[6] + dX1 * rec[7];
    const double h0 = Y0 * g00 + Y1 * g10, h1 = Y0 * g01 + Y1 * g11;             // r = 0, 1
    const double hx0 = Y0 * x00 + Y1 * x10, hx1 = Y0 * x01 + Y1 * x11;
    const double hy0 = dY0 * g00 + dY1 * g10, hy1 = dY0 * g01 + dY1 * g11;
    const double n = sum8(Z0 * h0 + Z1 * h1);
    double nx = sum8(Z0 * hx0 + Z1 * hx1) * g.inv_h[0], ny = sum8(Z0 * hy0 + Z1 * hy1) * g.inv_h[1], nz = sum8(dZ0 * h0 + dZ1 * h1) * g.inv_h[2];
    if (!BEND) nx = ny = nz = 0.0;
    return fermat_rates(n, nx, ny, nz, u, stype);
}
// The same right-hand side with FEWER lanes per ray (round 5).  Counters of the 8-lane form at 620 000 rays (profiles/r05_pmc_summary.json,
// leg fermat_cubic): vector issue 1.0 busy, 38.1 G wave-instructions = 62 ms of issue against 58 ms measured -- and most of them are work
// every one of a ray's eight lanes repeats (position, cell, Hermite sets, the ray equations, the RK4 bookkeeping: ~165 of ~200 per stage;
// only the 36 of a record's contraction differ).  With LPR lanes per ray a lane owns 8 / LPR nodes of the cell -- node (a, b, c) = the
// bits of sub * NPL + n -- and the shared part is done LPR times instead of eight.  LPR = 8 keeps the function above (small batches:
// 2 604 rays are 326 waves even so); large batches take LPR = 2 (four records = 64 VGPRs per lane, two waves per SIMD).
template <int LPR>
__device__ __forceinline__ double sum_lanes(double v) {
    if (LPR >= 2) v = dpp_xadd<0xB1>(v);      // quad_perm:[1,0,3,2]
    if (LPR >= 4) v = dpp_xadd<0x4E>(v);      // quad_perm:[2,3,0,1]
    if (LPR >= 8) v = dpp_xadd<0x141>(v);     // row_half_mirror
    return v;
}
template <bool BEND, int LPR>
__device__ __forceinline__ FState fermat_rhs_lmn(const GridView &g, const double *__restrict__ F8, const FState &u, int sub, int &ci, int &cj,
                                                 int &ck, double (&rec)[(8 / LPR) * 8], int stype) {
    constexpr int NPL = 8 / LPR;
    const double ux = (u.x - g.g0[0]) * g.inv_h[0], uy = (u.y - g.g0[1]) * g.inv_h[1], uz = (u.z - g.g0[2]) * g.inv_h[2];
    const double fi = fmin(fmax(__builtin_floor(ux), 2.0), (double)(g.nx - 4)), fj = fmin(fmax(__builtin_floor(uy), 2.0), (double)(g.ny - 4)),
                 fk = fmin(fmax(__builtin_floor(uz), 2.0), (double)(g.nz - 4));
    const int i = (int)fi, j = (int)fj, k = (int)fk;
    if ((i != ci) | (j != cj) | (k != ck)) {          // (the lanes of a ray hold the same state: they reload together)
#pragma unroll
        for (int n = 0; n < NPL; ++n) {
            const int nn = sub * NPL + n, a = nn >> 2, b = (nn >> 1) & 1, c = nn & 1;
            const double2 *p = (const double2 *)(F8 + ((size_t)(i + a) * LM_SI(g.ny, g.nz) + (size_t)(j + b) * LM_NZP(g.nz) + (size_t)(k + c)) * LM_NF);
            const double2 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
            double *q = rec + n * 8;
            q[0] = r0.x, q[1] = r0.y, q[2] = r1.x, q[3] = r1.y, q[4] = r2.x, q[5] = r2.y, q[6] = r3.x, q[7] = r3.y;
        }
        ci = i, cj = j, ck = k;
    }
    // Hermite value / slope weights of BOTH nodes along each axis, and their t-derivatives: w[hi][0 = value, 1 = slope]
    struct AxisW {
        double w[2][2], d[2][2];
    };
    auto axis = [](double t) {
        const double t2 = t * t, t3 = t2 * t;
        AxisW A;
        A.w[1][0] = 3.0 * t2 - 2.0 * t3, A.w[1][1] = t3 - t2, A.d[1][0] = 6.0 * t - 6.0 * t2, A.d[1][1] = 3.0 * t2 - 2.0 * t;
        A.w[0][0] = 1.0 - (3.0 * t2 - 2.0 * t3), A.w[0][1] = t3 - 2.0 * t2 + t, A.d[0][0] = 6.0 * t2 - 6.0 * t, A.d[0][1] = 3.0 * t2 - 4.0 * t + 1.0;
        return A;
    };
    const AxisW AX = axis(ux - fi), AY = axis(uy - fj), AZ = axis(uz - fk);
    double sn = 0.0, snx = 0.0, sny = 0.0, snz = 0.0;
#pragma unroll
    for (int n = 0; n < NPL; ++n) {
        const int nn = sub * NPL + n, a = nn >> 2, b = (nn >> 1) & 1, c = nn & 1;      // (b, c compile-time for LPR <= 2; a too for LPR = 1)
        const double X0 = a ? AX.w[1][0] : AX.w[0][0], X1 = a ? AX.w[1][1] : AX.w[0][1], dX0 = a ? AX.d[1][0] : AX.d[0][0], dX1 = a ? AX.d[1][1] : AX.d[0][1];
        const double Y0 = b ? AY.w[1][0] : AY.w[0][0], Y1 = b ? AY.w[1][1] : AY.w[0][1], dY0 = b ? AY.d[1][0] : AY.d[0][0], dY1 = b ? AY.d[1][1] : AY.d[0][1];
        const double Z0 = c ? AZ.w[1][0] : AZ.w[0][0], Z1 = c ? AZ.w[1][1] : AZ.w[0][1], dZ0 = c ? AZ.d[1][0] : AZ.d[0][0], dZ1 = c ? AZ.d[1][1] : AZ.d[0][1];
        const double *q = rec + n * 8;
        // contract p (x), then q (y), then r (z): q[p + 2 q + 4 r] (fermat_rhs_lm's order)
        const double g00 = X0 * q[0] + X1 * q[1], g10 = X0 * q[2] + X1 * q[3], g01 = X0 * q[4] + X1 * q[5], g11 = X0 * q[6] + X1 * q[7];
        const double x00 = dX0 * q[0] + dX1 * q[1], x10 = dX0 * q[2] + dX1 * q[3], x01 = dX0 * q[4] + dX1 * q[5], x11 = dX0 * q[6] + dX1 * q[7];
        const double h0 = Y0 * g00 + Y1 * g10, h1 = Y0 * g01 + Y1 * g11;
        const double hx0 = Y0 * x00 + Y1 * x10, hx1 = Y0 * x01 + Y1 * x11;
        const double hy0 = dY0 * g00 + dY1 * g10, hy1 = dY0 * g01 + dY1 * g11;
        sn += Z0 * h0 + Z1 * h1, snx += Z0 * hx0 + Z1 * hx1, sny += Z0 * hy0 + Z1 * hy1, snz += dZ0 * h0 + dZ1 * h1;
    }
    const double nv = sum_lanes<LPR>(sn);
    double nx = sum_lanes<LPR>(snx) * g.inv_h[0], ny = sum_lanes<LPR>(sny) * g.inv_h[1], nz = sum_lanes<LPR>(snz) * g.inv_h[2];
    if (!BEND) nx = ny = nz = 0.0;
    return fermat_rates(nv, nx, ny, nz, u, stype);
}
// (LPR = 8: the function the round-4 kernels were validated with, instruction for instruction)
template <bool BEND, int LPR>
__device__ __forceinline__ FState fermat_rhs_lm_any(const GridView &g, const double *__restrict__ F8, const FState &u, int sub, int &ci, int &cj,
                                                    int &ck, double (&rec)[(8 / LPR) * 8], int stype) {
    if constexpr (LPR == 8) return fermat_rhs_lm<BEND>(g, F8, u, sub, ci, cj, ck, rec, stype);
    else return fermat_rhs_lmn<BEND, LPR>(g, F8, u, sub, ci, cj, ck, rec, stype);
}
template <bool BEND, int LPR = 8>
__global__ __launch_bounds__(64) void k_trace_fermat_lm(GridView g, const double *__restrict__ F8, const double *__restrict__ origins,
                                                        const double *__restrict__ dirs, int64_t R, double tmax, int Ns, int substeps,
                                                        double *__restrict__ rays, int *oob_flag, int rays_per_wave, int stype) {
    if ((int)(threadIdx.x / LPR) >= rays_per_wave) return;
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
    double *o = rays + (size_t)r * 4 * Ns;
    const bool writer = live && sub == 0;
    if (writer) o[0] = u.x, o[Ns] = u.y, o[2 * Ns] = u.z, o[3 * Ns] = u.s;
    bool oob = false;
    int ci = -1, cj = -1, ck = -1;
    double rec[(8 / LPR) * 8] = {};
    const double ztop = g.glast[2] + 1e-9 * fabs(tmax);
    for (int k = 1; k < Ns; ++k) {
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
        if (writer) o[k] = u.x, o[Ns + k] = u.y, o[2 * Ns + k] = u.z, o[3 * Ns + k] = u.s;
    }
    if (oob && writer) atomicOr(oob_flag, 1);
}

// ---- trilinear tracer for small batches: 4 lanes per ray, axes in LDS, cell-cached corners ---------------
// With lanes = rays every RK4 stage is a chain of dependent global loads (axis look-ups for the cell, then 8
// corners spread over 4 cache lines per lane: 64 lanes x 4 lines = the whole 32 KB L1), ~1.4 us per stage and
// 2 064 stages per ray.  Here a ray is shared by 4 consecutive lanes: lane (a, b) = (sub >> 1, sub & 1) owns the
// corner column (i + a, j + b) and KEEPS a run of 8 of its z nodes in registers, so global memory is touched
// only when the ray changes column or runs off the run (every ~7 cells = ~100 stages at 4 sub-steps per
// sample); the cell bounds are cached too, so a stage inside the same cell needs no axis look-up; the axis
// tables sit in LDS and the four partial results are summed with two DPP steps.  Same cell rule as find_cell,
// same cell polynomial as trilinear_grad_at (reciprocal widths instead of divisions: ulp-level differences).
__device__ __forceinline__ double sum4(double v) {
    v = dpp_xadd<0xB1>(v);      // quad_perm:[1,0,3,2]
    return dpp_xadd<0x4E>(v);   // quad_perm:[2,3,0,1]
}
constexpr int L4_RUN = 8;       // z-run of nodes cached per lane: cells kb .. kb + L4_RUN - 2 need no reload
struct CornerPair {
    int i, j, k, kb;            // cached cell (i, j, k) and first node of the cached z-run (-1: nothing cached)
    double run[L4_RUN];         // nM at (i + a, j + b, kb .. kb + L4_RUN - 1)
    double v0, v1;              // the pair of the current cell: run[k - kb], run[k - kb + 1]
    double x0, x1, y0, y1, z0, z1, rhx, rhy, rhz;   // cell bounds (g[i] < x <= g[i+1]) and reciprocal widths
};
__device__ __forceinline__ double pick_run(const double (&v)[L4_RUN], int t) {
    const double a0 = (t & 1) ? v[1] : v[0], a1 = (t & 1) ? v[3] : v[2], a2 = (t & 1) ? v[5] : v[4], a3 = (t & 1) ? v[7] : v[6];
    const double b0 = (t & 2) ? a1 : a0, b1 = (t & 2) ? a3 : a2;
    return (t & 4) ? b1 : b0;
}
template <bool BEND>
__device__ __forceinline__ FState fermat_rhs_lin4(const GridView &g, const Axes &ax, const double *__restrict__ nM,
                                                  const FState &u, int a, int b, CornerPair &cc, int stype) {
    // fast path: still inside the cached cell (same rule as find_cell: g[i] < x <= g[i+1]; the first cell also
    // owns its lower face).  The 4 lanes of a ray hold the same state, so they take the same branch.
    const bool in_x = (u.x > cc.x0 || cc.i == 0) && u.x <= cc.x1, in_y = (u.y > cc.y0 || cc.j == 0) && u.y <= cc.y1;
    const bool in_z = (u.z > cc.z0 || cc.k == 0) && u.z <= cc.z1;
    if (!(in_x && in_y && in_z) || cc.kb < 0) {
        const int i = find_cell(ax.x, ax.nx, u.x, g.inv_h[0], g.uniform[0]);
        const int j = find_cell(ax.y, ax.ny, u.y, g.inv_h[1], g.uniform[1]);
        const int k = find_cell(ax.z, ax.nz, u.z, g.inv_h[2], g.uniform[2]);
        if (i != cc.i || j != cc.j || cc.kb < 0 || k < cc.kb || k > cc.kb + L4_RUN - 2) {
            // (re)load the z-run of this lane's corner column; the run is clamped to the axis, so a run that
            // would stick out at the top starts lower instead
            const int kb = max(0, min(k, g.nz - L4_RUN));
            const double *p = nM + ((size_t)(i + a) * g.ny + (j + b)) * g.nz + kb;
#pragma unroll
            for (int t = 0; t < L4_RUN; ++t) cc.run[t] = (kb + t < g.nz) ? p[t] : 0.0;
            cc.kb = kb;
        }
        cc.i = i, cc.j = j, cc.k = k;
        cc.v0 = pick_run(cc.run, k - cc.kb);
        cc.v1 = pick_run(cc.run, k - cc.kb + 1);
        cc.x0 = ax.x[i], cc.x1 = ax.x[i + 1], cc.y0 = ax.y[j], cc.y1 = ax.y[j + 1], cc.z0 = ax.z[k], cc.z1 = ax.z[k + 1];
        cc.rhx = 1.0 / (cc.x1 - cc.x0), cc.rhy = 1.0 / (cc.y1 - cc.y0), cc.rhz = 1.0 / (cc.z1 - cc.z0);
    }
    const double tx = (u.x - cc.x0) * cc.rhx, ty = (u.y - cc.y0) * cc.rhy, tz = (u.z - cc.z0) * cc.rhz;
    const double wxa = a ? tx : 1.0 - tx, wyb = b ? ty : 1.0 - ty;
    const double sa = a ? 1.0 : -1.0, sb = b ? 1.0 : -1.0;
    const double dvz = cc.v1 - cc.v0, vz = cc.v0 + tz * dvz;
    const double n = sum4(vz * wxa * wyb);
    double nx = sum4(vz * sa * wyb) * cc.rhx, ny = sum4(vz * wxa * sb) * cc.rhy, nz = sum4(dvz * wxa * wyb) * cc.rhz;
    if (!BEND) nx = ny = nz = 0.0;
    return fermat_rates(n, nx, ny, nz, u, stype);
}
template <bool BEND>
__global__ __launch_bounds__(64) void k_trace_fermat_lin4(GridView g, const double *__restrict__ nM,
                                                          const double *__restrict__ origins, const double *__restrict__ dirs,
                                                          int64_t R, double tmax, int Ns, int substeps, double *__restrict__ rays,
                                                          int *oob_flag, int rays_per_wave, int stype) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Axes ax = stage_axes(g, (double *)smem);
    __syncthreads();
    // rays_per_wave < 16 leaves lanes idle on purpose: the wave takes the slow path whenever ANY of its rays changes
    // cell, so while the batch does not fill the chip anyway, fewer rays per wave means fewer slow stages per ray
    if ((int)(threadIdx.x >> 2) >= rays_per_wave) return;
    const int sub = threadIdx.x & 3, a = sub >> 1, b = sub & 1;
    int64_t r = (int64_t)blockIdx.x * rays_per_wave + (threadIdx.x >> 2);
    const bool live = r < R;
    if (!live) r = R - 1;                      // idle groups shadow the last ray (DPP needs all lanes running)
    const double dx = dirs[3 * r], dy = dirs[3 * r + 1], dz = dirs[3 * r + 2];
    const double nrm = sqrt(dx * dx + dy * dy + dz * dz);
    FState u;
    u.px = dx / nrm;
    u.py = dy / nrm;
    u.pz = dz / nrm;
    u.x = origins[3 * r];
    u.y = origins[3 * r + 1];
    u.z = origins[3 * r + 2];
    u.s = 0.0;
    const double h = fermat_step(tmax, u.z, Ns, substeps, stype);
    double *o = rays + (size_t)r * 4 * Ns;
    const bool writer = live && sub == 0;
    if (writer) {
        o[0] = u.x;
        o[Ns] = u.y;
        o[2 * Ns] = u.z;
        o[3 * Ns] = u.s;
    }
    CornerPair cc = {};
    cc.i = cc.j = cc.k = cc.kb = -1;
    bool oob = false;
    for (int k = 1; k < Ns; ++k) {
        for (int s2 = 0; s2 < substeps; ++s2) {
            FState kprev = {}, sum = {};
#pragma unroll 1
            for (int st = 0; st < 4; ++st) {
                const double ca = st == 0 ? 0.0 : (st == 3 ? h : 0.5 * h);
                kprev = fermat_rhs_lin4<BEND>(g, ax, nM, axpy(u, ca, kprev), a, b, cc, stype);
                sum = axpy(sum, (st == 1 || st == 2) ? 2.0 : 1.0, kprev);
            }
            u = axpy(u, h / 6.0, sum);
        }
        oob |= outside(ax.x, ax.nx, u.x) || outside(ax.y, ax.ny, u.y) ||
               !(u.z >= ax.z[0] && u.z <= ax.z[ax.nz - 1] + 1e-9 * fabs(tmax));
        if (writer) {
            o[k] = u.x;
            o[Ns + k] = u.y;
            o[2 * Ns + k] = u.z;
            o[3 * Ns + k] = u.s;
        }
    }
    if (oob && writer) atomicOr(oob_flag, 1);
}

// ---- small batches on ideal-uniform grids: one ray per lane, few lanes per wave, the cell's polynomial in registers -------------------
// A single coherence window (config 3: 2 604 rays) cannot fill the chip whatever the mapping: the time is the dependent chain
// of one ray -- 128 samples x 4 substeps x 4 stages -- and the 4-lanes-per-ray kernel spends it on DPP sums, cell-bound compares and
// a reciprocal per stage (510 ns per evaluation).  Inside a cell the trilinear index is the polynomial
//     n = k0 + k1 tx + k2 ty + k3 tz + k4 tx ty + k5 tx tz + k6 ty tz + k7 tx ty tz        (t = local coordinates in the cell)
// whose value and gradient are 12 fused multiply-adds three deep; the eight coefficients are formed once per cell (a ray stays
// ~16 stages in one) from four 16-byte loads.  The cell is floor((x - g0) / h) except within 1e-9 of a face, where the axis
// tables decide as scipy does (find_cell: the gradient jumps there) -- the same rule, the same cell polynomial as
// trilinear_grad_at, to rounding.  rays_per_wave lanes of a wave are used (16 measured best: a lone wave issues one instruction
// per ~10 cycles whatever its lane count, 88 vector + 24 scalar instructions per stage -- the time IS that chain; more rays per
// wave only add cell changes that stall their neighbours, fewer waste issue slots).
struct CellPoly {
    int i, j, k;          // cached cell (-1: none)
    double k0, k1, k2, k3, k4, k5, k6, k7;
};
__device__ __forceinline__ void cell_poly_load(const GridView &g, const double *__restrict__ nM, int i, int j, int k, CellPoly &cp) {
    const size_t sj = (size_t)g.nz, si = (size_t)g.ny * g.nz;
    const double *p = nM + ((size_t)i * g.ny + j) * g.nz + k;
    const double c000 = p[0], c001 = p[1], c010 = p[sj], c011 = p[sj + 1];
    const double c100 = p[si], c101 = p[si + 1], c110 = p[si + sj], c111 = p[si + sj + 1];
    cp.i = i, cp.j = j, cp.k = k;
    cp.k0 = c000;
    cp.k1 = c100 - c000, cp.k2 = c010 - c000, cp.k3 = c001 - c000;
    cp.k4 = (c110 - c100) - (c010 - c000);
    cp.k5 = (c101 - c100) - (c001 - c000);
    cp.k6 = (c011 - c010) - (c001 - c000);
    cp.k7 = ((c111 - c110) - (c101 - c100)) - ((c011 - c010) - (c001 - c000));
}
template <bool BEND>
__device__ __forceinline__ FState fermat_rhs_poly(const GridView &g, const double *gx, const double *gy, const double *gz,
                                                  const double *__restrict__ nM, const FState &u, CellPoly &cp, int stype) {
    const double ux = (u.x - g.g0[0]) * g.inv_h[0], uy = (u.y - g.g0[1]) * g.inv_h[1], uz = (u.z - g.g0[2]) * g.inv_h[2];
    double fi = __builtin_floor(ux), fj = __builtin_floor(uy), fk = __builtin_floor(uz);
    const double eps = 1e-9;
    {
        const double tx = ux - fi, ty = uy - fj, tz = uz - fk;
        // within eps of a face (or outside the grid): the axis tables name the cell, as scipy would
        if (!((tx > eps) & (tx < 1.0 - eps) & (fi >= 0.0) & (fi <= (double)(g.nx - 2)))) fi = (double)find_cell(gx, g.nx, u.x, g.inv_h[0], g.uniform[0]);
        if (!((ty > eps) & (ty < 1.0 - eps) & (fj >= 0.0) & (fj <= (double)(g.ny - 2)))) fj = (double)find_cell(gy, g.ny, u.y, g.inv_h[1], g.uniform[1]);
        if (!((tz > eps) & (tz < 1.0 - eps) & (fk >= 0.0) & (fk <= (double)(g.nz - 2)))) fk = (double)find_cell(gz, g.nz, u.z, g.inv_h[2], g.uniform[2]);
    }
    const int i = (int)fi, j = (int)fj, k = (int)fk;
    if ((i != cp.i) | (j != cp.j) | (k != cp.k)) cell_poly_load(g, nM, i, j, k, cp);
    const double tx = ux - fi, ty = uy - fj, tz = uz - fk;
    const double a = __builtin_fma(cp.k7, tz, cp.k4);              // d2n / dtx dty
    const double nxt = __builtin_fma(ty, a, __builtin_fma(cp.k5, tz, cp.k1));
    const double b = __builtin_fma(cp.k6, tz, cp.k2);
    const double nyt = __builtin_fma(tx, a, b);
    const double nzt = __builtin_fma(ty, __builtin_fma(cp.k7, tx, cp.k6), __builtin_fma(cp.k5, tx, cp.k3));
    const double n = __builtin_fma(tx, nxt, __builtin_fma(ty, b, __builtin_fma(cp.k3, tz, cp.k0)));
    double nx = nxt * g.inv_h[0], ny = nyt * g.inv_h[1], nz = nzt * g.inv_h[2];
    if (!BEND) nx = ny = nz = 0.0;
    return fermat_rates(n, nx, ny, nz, u, stype);
}
template <bool BEND>
__global__ __launch_bounds__(64) void k_trace_fermat_poly(GridView g, const double *__restrict__ nM, const double *__restrict__ origins,
                                                          const double *__restrict__ dirs, int64_t R, double tmax, int Ns, int substeps,
                                                          double *__restrict__ rays, int *oob_flag, int rays_per_wave, int stype) {
    if ((int)threadIdx.x >= rays_per_wave) return;
    const int64_t r = (int64_t)blockIdx.x * rays_per_wave + threadIdx.x;
    if (r >= R) return;
    const double *gx = g.axes, *gy = g.axes + g.nx, *gz = g.axes + g.nx + g.ny;
    const double dx = dirs[3 * r], dy = dirs[3 * r + 1], dz = dirs[3 * r + 2];
    const double nrm = sqrt(dx * dx + dy * dy + dz * dz);
    FState u;
    u.px = dx / nrm, u.py = dy / nrm, u.pz = dz / nrm;
    u.x = origins[3 * r], u.y = origins[3 * r + 1], u.z = origins[3 * r + 2];
    u.s = 0.0;
    const double h = fermat_step(tmax, u.z, Ns, substeps, stype);
    double *o = rays + (size_t)r * 4 * Ns;
    o[0] = u.x, o[Ns] = u.y, o[2 * Ns] = u.z, o[3 * Ns] = u.s;
    CellPoly cp;
    cp.i = cp.j = cp.k = -1;
    const double ztop = g.glast[2] + 1e-9 * fabs(tmax);
    bool oob = false;
    for (int k = 1; k < Ns; ++k) {
        for (int sub = 0; sub < substeps; ++sub) {
            // the four stages spelled out (same operations in the same order as the stage loop of k_trace_fermat)
            FState kprev = {}, sum = {};
            kprev = fermat_rhs_poly<BEND>(g, gx, gy, gz, nM, axpy(u, 0.0, kprev), cp, stype);
            sum = axpy(sum, 1.0, kprev);
            kprev = fermat_rhs_poly<BEND>(g, gx, gy, gz, nM, axpy(u, 0.5 * h, kprev), cp, stype);
            sum = axpy(sum, 2.0, kprev);
            kprev = fermat_rhs_poly<BEND>(g, gx, gy, gz, nM, axpy(u, 0.5 * h, kprev), cp, stype);
            sum = axpy(sum, 2.0, kprev);
            kprev = fermat_rhs_poly<BEND>(g, gx, gy, gz, nM, axpy(u, h, kprev), cp, stype);
            sum = axpy(sum, 1.0, kprev);
            u = axpy(u, h / 6.0, sum);
        }
        oob |= !(u.x >= g.g0[0] && u.x <= g.glast[0] && u.y >= g.g0[1] && u.y <= g.glast[1] && u.z >= g.g0[2] && u.z <= ztop);
        o[k] = u.x, o[Ns + k] = u.y, o[2 * Ns + k] = u.z, o[3 * Ns + k] = u.s;
    }
    if (oob) atomicOr(oob_flag, 1);
}


}  // namespace

#endif
