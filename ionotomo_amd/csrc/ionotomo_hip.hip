// ionotomo_hip.hip -- MI355X (gfx950 / CDNA4) ray-integral engine behind include/ionotomo_hip.h.
//
// Hot path of Joshuaalbert/IonoTomo, rebuilt for 64-wide wavefronts:
//   forward : one wavefront per ray, lanes = samples along the ray.  Rays are z-parametrised and
//             the grid is C-ordered with z fastest, so the 64 lanes of a wave read z-contiguous
//             runs of the four (i,j) corner columns -> coalesced HBM/L2 reads; the per-ray
//             quadrature (Simpson) is a weighted wave-level reduction (DPP shuffles).
//   adjoint : the same traversal scattering  w_r c_k W_kv  with hardware float atomics; lanes
//             of a wave hit z-contiguous addresses (well-shaped atomic wave-instructions).
//   tracer  : the Fermat ODE is sequential in z, so there lanes = rays (fixed-step RK4).
// The path is gather/bandwidth bound (about 1 flop per byte): no MFMA anywhere.
//
// Reference citations (file:line) are relative to /root/reference/src/ionotomo/.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/ionotomo_hip.h"

#define IONO_VERSION 100
#define PLASMA_A (8.980 * 8.980)             // inversion/fermat.py:42
#define SPEED_OF_LIGHT 299792458.0           // inversion/iterative_newton.py:15

namespace {

thread_local std::string g_last_error;

// ------------------------------------------------------------------------------------------------
// device-side grid description
// ------------------------------------------------------------------------------------------------
struct GridView {
    const double *axes;   // xvec | yvec | zvec concatenated (device)
    const void *M;        // nx*ny*nz values, float64 or float32
    int nx, ny, nz;
    double inv_h[3];      // 1/(mean spacing) per axis: first guess of the cell index
    int uniform[3];       // axis is (numerically) uniform -> guess + fix-up; else binary search
    double g0[3], glast[3];   // first / last node per axis (host copies)
};

struct Axes {             // axis tables staged in LDS
    const double *x, *y, *z;
    int nx, ny, nz;
};

__device__ __forceinline__ Axes stage_axes(const GridView &g, double *lds) {
    const int n = g.nx + g.ny + g.nz;
    for (int t = threadIdx.x; t < n; t += blockDim.x) lds[t] = g.axes[t];
    __syncthreads();
    Axes a;
    a.x = lds;
    a.y = lds + g.nx;
    a.z = lds + g.nx + g.ny;
    a.nx = g.nx;
    a.ny = g.ny;
    a.nz = g.nz;
    return a;
}

// scipy RegularGridInterpolator._find_indices: i = clip(searchsorted(g, x) - 1, 0, n-2), i.e.
// g[i] < x <= g[i+1] inside the grid (tomography/interpolation.py:166-196 spells it out).
__device__ __forceinline__ int find_cell(const double *g, int n, double x, double inv_h, int uniform) {
    int i;
    if (uniform) {
        double f = (x - g[0]) * inv_h;
        f = fmin(fmax(f, 0.0), (double)(n - 2));
        i = (int)f;
    } else {
        int lo = 0, hi = n - 1;
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (g[mid] < x) lo = mid; else hi = mid;
        }
        i = lo;
    }
    while (i > 0 && !(g[i] < x)) --i;
    while (i < n - 2 && g[i + 1] < x) ++i;
    return i;
}

__device__ __forceinline__ bool outside(const double *g, int n, double x) {
    return !(x >= g[0] && x <= g[n - 1]);     // NaN is outside, like scipy
}

// ---- trilinear (geometry/tri_cubic.py:69-70 -> scipy RGI 'linear') -----------------------------
template <typename GT>
__device__ __forceinline__ double trilinear_at(const GridView &g, const Axes &ax, double x, double y, double z) {
    const int i = find_cell(ax.x, ax.nx, x, g.inv_h[0], g.uniform[0]);
    const int j = find_cell(ax.y, ax.ny, y, g.inv_h[1], g.uniform[1]);
    const int k = find_cell(ax.z, ax.nz, z, g.inv_h[2], g.uniform[2]);
    const double tx = (x - ax.x[i]) / (ax.x[i + 1] - ax.x[i]);
    const double ty = (y - ax.y[j]) / (ax.y[j + 1] - ax.y[j]);
    const double tz = (z - ax.z[k]) / (ax.z[k + 1] - ax.z[k]);
    const GT *p = (const GT *)g.M + ((size_t)i * g.ny + j) * g.nz + k;
    const size_t sj = g.nz, si = (size_t)g.ny * g.nz;
    const double c000 = p[0], c001 = p[1];
    const double c010 = p[sj], c011 = p[sj + 1];
    const double c100 = p[si], c101 = p[si + 1];
    const double c110 = p[si + sj], c111 = p[si + sj + 1];
    const double c00 = c000 + tz * (c001 - c000);
    const double c01 = c010 + tz * (c011 - c010);
    const double c10 = c100 + tz * (c101 - c100);
    const double c11 = c110 + tz * (c111 - c110);
    const double c0 = c00 + ty * (c01 - c00);
    const double c1 = c10 + ty * (c11 - c10);
    return c0 + tx * (c1 - c0);
}

// value and analytic gradient of the trilinear cell polynomial (double grid only; tracer)
__device__ __forceinline__ void trilinear_grad_at(const GridView &g, const double *M, double x, double y, double z,
                                                  double &f, double &fx, double &fy, double &fz) {
    const double *gx = g.axes, *gy = g.axes + g.nx, *gz = g.axes + g.nx + g.ny;
    const int i = find_cell(gx, g.nx, x, g.inv_h[0], g.uniform[0]);
    const int j = find_cell(gy, g.ny, y, g.inv_h[1], g.uniform[1]);
    const int k = find_cell(gz, g.nz, z, g.inv_h[2], g.uniform[2]);
    const double hx = gx[i + 1] - gx[i], hy = gy[j + 1] - gy[j], hz = gz[k + 1] - gz[k];
    const double tx = (x - gx[i]) / hx, ty = (y - gy[j]) / hy, tz = (z - gz[k]) / hz;
    const double *p = M + ((size_t)i * g.ny + j) * g.nz + k;
    const size_t sj = g.nz, si = (size_t)g.ny * g.nz;
    const double wx[2] = {1 - tx, tx}, wy[2] = {1 - ty, ty}, wz[2] = {1 - tz, tz};
    const double sg[2] = {-1.0, 1.0};
    f = fx = fy = fz = 0.0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double v = p[a * si + b * sj + c];
                f += v * wx[a] * wy[b] * wz[c];
                fx += v * sg[a] * wy[b] * wz[c];
                fy += v * wx[a] * sg[b] * wz[c];
                fz += v * wx[a] * wy[b] * sg[c];
            }
    fx /= hx;
    fy /= hy;
    fz /= hz;
}

// ---- tricubic: Lekien-Marsden with 4th-order central-difference derivative data --------------
// (notebooks/TricubicInterpolation.ipynb c0:138-1257).  With finite-difference slopes (mixed
// ones formed by the same 1-D stencil along each axis) the interpolant is the tensor product of
// 1-D cubic Hermite splines whose slopes are (f[i-2] - 8 f[i-1] + 8 f[i+1] - f[i+2]) /
// (6 (x[i+1] - x[i-1])): 6 taps per axis, support i-2 .. i+3.  Slopes are scaled to cell units
// (df/du = h df/dx), which Lekien-Marsden requires; see oracle.tricubic_axis_weights.
__device__ __forceinline__ int cubic_axis(const double *g, int n, double x, double inv_h, int uniform,
                                          double w[6], double dw[6], bool want_d) {
    int i = find_cell(g, n, x, inv_h, uniform);
    i = min(max(i, 2), n - 4);
    const double h = g[i + 1] - g[i];
    const double t = (x - g[i]) / h;
    const double t2 = t * t, t3 = t2 * t;
    const double b0 = 2 * t3 - 3 * t2 + 1, b1 = -2 * t3 + 3 * t2, b2 = t3 - 2 * t2 + t, b3 = t3 - t2;
    const double c0 = h / (6.0 * (g[i + 1] - g[i - 1]));
    const double c1 = h / (6.0 * (g[i + 2] - g[i]));
    w[0] = b2 * c0;
    w[1] = -8.0 * b2 * c0 + b3 * c1;
    w[2] = b0 - 8.0 * b3 * c1;
    w[3] = b1 + 8.0 * b2 * c0;
    w[4] = -b2 * c0 + 8.0 * b3 * c1;
    w[5] = -b3 * c1;
    if (want_d) {
        const double d0 = (6 * t2 - 6 * t) / h, d1 = (-6 * t2 + 6 * t) / h;
        const double d2 = (3 * t2 - 4 * t + 1) / h, d3 = (3 * t2 - 2 * t) / h;
        dw[0] = d2 * c0;
        dw[1] = -8.0 * d2 * c0 + d3 * c1;
        dw[2] = d0 - 8.0 * d3 * c1;
        dw[3] = d1 + 8.0 * d2 * c0;
        dw[4] = -d2 * c0 + 8.0 * d3 * c1;
        dw[5] = -d3 * c1;
    }
    return i;
}

template <typename GT, bool GRAD>
__device__ __forceinline__ void tricubic_eval(const GridView &g, const double *gx, const double *gy, const double *gz,
                                              double x, double y, double z, double &f, double &fx, double &fy, double &fz) {
    double wx[6], wy[6], wz[6], dx[6], dy[6], dz[6];
    const int i = cubic_axis(gx, g.nx, x, g.inv_h[0], g.uniform[0], wx, dx, GRAD);
    const int j = cubic_axis(gy, g.ny, y, g.inv_h[1], g.uniform[1], wy, dy, GRAD);
    const int k = cubic_axis(gz, g.nz, z, g.inv_h[2], g.uniform[2], wz, dz, GRAD);
    const GT *base = (const GT *)g.M + ((size_t)(i - 2) * g.ny + (j - 2)) * g.nz + (k - 2);
    f = fx = fy = fz = 0.0;
    for (int a = 0; a < 6; ++a) {
        double fa = 0.0, fya = 0.0, fza = 0.0;
        for (int b = 0; b < 6; ++b) {
            const GT *p = base + ((size_t)a * g.ny + b) * g.nz;
            double s = 0.0, sz = 0.0;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const double v = p[c];
                s += v * wz[c];
                if (GRAD) sz += v * dz[c];
            }
            fa += s * wy[b];
            if (GRAD) {
                fya += s * dy[b];
                fza += sz * wy[b];
            }
        }
        f += fa * wx[a];
        if (GRAD) {
            fx += fa * dx[a];
            fy += fya * wx[a];
            fz += fza * wx[a];
        }
    }
}

template <typename GT, int KIND>
__device__ __forceinline__ double sample_at(const GridView &g, const Axes &ax, double x, double y, double z) {
    if (KIND == IONO_INTERP_TRILINEAR) return trilinear_at<GT>(g, ax, x, y, z);
    double f, fx, fy, fz;
    tricubic_eval<GT, false>(g, ax.x, ax.y, ax.z, x, y, z, f, fx, fy, fz);
    return f;
}

template <int KIND>
__device__ __forceinline__ bool sample_outside(const Axes &ax, double x, double y, double z) {
    if (KIND == IONO_INTERP_TRILINEAR)
        return outside(ax.x, ax.nx, x) || outside(ax.y, ax.ny, y) || outside(ax.z, ax.nz, z);
    // tricubic needs the 6-node stencil: valid for g[2] <= x <= g[n-3]
    return !(x >= ax.x[2] && x <= ax.x[ax.nx - 3]) || !(y >= ax.y[2] && y <= ax.y[ax.ny - 3]) ||
           !(z >= ax.z[2] && z <= ax.z[ax.nz - 3]);
}

// ------------------------------------------------------------------------------------------------
// quadrature weights on explicit abscissae s[0..N) (tomography/integrate.py:50-74,130-153;
// scipy.integrate.simpson for the Cartwright even-N rule)
// ------------------------------------------------------------------------------------------------
// composite Simpson weight of sample k within the odd-length sub-range [a, b]
__device__ __forceinline__ double basic_simpson_weight(const double *s, int a, int b, int k) {
    if (k < a || k > b || b - a < 2) return 0.0;
    const int p = k - a;
    double w = 0.0;
    if (p & 1) {
        const double h0 = s[k] - s[k - 1], h1 = s[k + 1] - s[k], hs = h0 + h1;
        w = hs / 6.0 * (hs * hs / (h0 * h1));
    } else {
        if (k > a) {
            const double h0 = s[k - 1] - s[k - 2], h1 = s[k] - s[k - 1], hs = h0 + h1;
            w += hs / 6.0 * (2.0 - h0 / h1);
        }
        if (k < b) {
            const double h0 = s[k + 1] - s[k], h1 = s[k + 2] - s[k + 1], hs = h0 + h1;
            w += hs / 6.0 * (2.0 - h1 / h0);
        }
    }
    return w;
}

__device__ __forceinline__ double quad_weight(const double *s, int N, int k, int rule) {
    if (rule == IONO_QUAD_TRAPEZOID || N == 2) {
        double w = 0.0;
        if (k > 0) w += 0.5 * (s[k] - s[k - 1]);
        if (k < N - 1) w += 0.5 * (s[k + 1] - s[k]);
        return w;
    }
    if (N & 1) return basic_simpson_weight(s, 0, N - 1, k);
    if (rule == IONO_QUAD_SIMPSON_AVG) {
        double wa = basic_simpson_weight(s, 0, N - 2, k);
        if (k >= N - 2) wa += 0.5 * (s[N - 1] - s[N - 2]);
        double wb = basic_simpson_weight(s, 1, N - 1, k);
        if (k <= 1) wb += 0.5 * (s[1] - s[0]);
        return 0.5 * (wa + wb);
    }
    double w = basic_simpson_weight(s, 0, N - 2, k);
    const double h0 = s[N - 2] - s[N - 3], h1 = s[N - 1] - s[N - 2];
    if (k == N - 1) w += (2 * h1 * h1 + 3 * h0 * h1) / (6 * (h0 + h1));
    if (k == N - 2) w += (h1 * h1 + 3 * h0 * h1) / (6 * h0);
    if (k == N - 3) w -= h1 * h1 * h1 / (6 * h0 * (h0 + h1));
    return w;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// rays are dealt so that each XCD (blocks b, b+8, ... share one) walks a contiguous range of
// rays: neighbouring rays (same antenna / neighbouring directions) share grid columns, which then
// stay in that XCD's private L2.  Pure speed heuristic; correctness never depends on placement.
struct RayWalk {
    int64_t r, end, stride;
};
__device__ __forceinline__ RayWalk ray_walk(int64_t R) {
    const int wpb = blockDim.x >> 6, wid = threadIdx.x >> 6;
    RayWalk w;
    if ((gridDim.x & 7) == 0 && R >= 64 * 8) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
        const int64_t per = (R + 7) / 8;
        const int64_t lo = per * xcd;
        w.end = min(R, lo + per);
        w.r = lo + (int64_t)slot * wpb + wid;
        w.stride = (int64_t)nslot * wpb;
    } else {
        w.r = (int64_t)blockIdx.x * wpb + wid;
        w.end = R;
        w.stride = (int64_t)gridDim.x * wpb;
    }
    return w;
}

struct StraightRay {
    double ox, oy, oz, sx, sy, L, h, step, pz;
};
// straight z-parametrised ray: z = linspace(z0, tmax, N), x = x0 + px/pz (z - z0), s = (z - z0)/pz
// (inversion/fermat.py:64-72,150-174 with n = 1; == tomography/model.py:27-35)
__device__ __forceinline__ StraightRay load_straight(const double *origins, const double *dirs, int64_t r,
                                                     double tmax, int Ns) {
    StraightRay q;
    q.ox = origins[3 * r];
    q.oy = origins[3 * r + 1];
    q.oz = origins[3 * r + 2];
    const double dx = dirs[3 * r], dy = dirs[3 * r + 1], dz = dirs[3 * r + 2];
    const double nrm = sqrt(dx * dx + dy * dy + dz * dz);
    const double px = dx / nrm, py = dy / nrm, pz = dz / nrm;
    q.sx = px / pz;
    q.sy = py / pz;
    q.L = tmax - q.oz;
    q.step = 1.0 / (double)(Ns - 1);
    q.h = q.L * q.step / pz;      // uniform spacing of s
    q.pz = pz;
    return q;
}
__device__ __forceinline__ void straight_point(const StraightRay &q, int k, int Ns, double &x, double &y, double &z) {
    const double frac = (k == Ns - 1) ? 1.0 : (double)k * q.step;
    const double dz = q.L * frac;
    x = q.ox + q.sx * dz;
    y = q.oy + q.sy * dz;
    z = q.oz + dz;
}

// ------------------------------------------------------------------------------------------------
// forward kernels
// ------------------------------------------------------------------------------------------------
template <typename GT, int KIND>
__global__ __launch_bounds__(256) void k_forward_straight(GridView g, const double *__restrict__ origins,
                                                          const double *__restrict__ dirs, int64_t R, double tmax, int Ns,
                                                          const double *__restrict__ unitw, double *__restrict__ tec,
                                                          int *oob_flag) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const Axes ax = stage_axes(g, lds);
    const int lane = threadIdx.x & 63;
    bool oob = false;
    for (RayWalk w = ray_walk(R); w.r < w.end; w.r += w.stride) {
        const StraightRay q = load_straight(origins, dirs, w.r, tmax, Ns);
        double acc = 0.0;
        for (int k = lane; k < Ns; k += 64) {
            double x, y, z;
            straight_point(q, k, Ns, x, y, z);
            if (sample_outside<KIND>(ax, x, y, z)) {
                oob = true;
                continue;
            }
            acc += unitw[k] * sample_at<GT, KIND>(g, ax, x, y, z);
        }
        acc = wave_sum(acc);
        if (lane == 0) tec[w.r] = acc * q.h;
    }
    if (__any(oob) && lane == 0) atomicOr(oob_flag, 1);
}

// ---- fast path (trilinear, numerically uniform axes, grid < 4 GB): the instruction diet ---------
// The general kernel above is issue-bound, not memory-bound (float32 storage buys nothing): three
// f64 divisions, per-sample bounds tests and looped cell fix-ups dominate.  Here: reciprocal cell
// widths are tabulated in LDS beside the axes (t = (x - g[i]) * inv[i]), the cell guess
// floor((x - g0)/h) is verified against the table with one compare pair (the exact searchsorted
// rule runs only for lanes whose guess is off, i.e. samples within rounding of a node), the
// bounds test is done once per ray on its two end points (a straight segment in a convex box),
// and addressing is 32-bit.
struct FastAxes {
    const double *g[3];
    const double *inv[3];
    double g0[3];
};

__device__ __forceinline__ FastAxes stage_axes_fast(const GridView &g, double *lds) {
    const int n = g.nx + g.ny + g.nz;
    for (int t = threadIdx.x; t < n; t += blockDim.x) lds[t] = g.axes[t];
    __syncthreads();
    for (int t = threadIdx.x; t < n - 1; t += blockDim.x) lds[n + t] = 1.0 / (lds[t + 1] - lds[t]);
    __syncthreads();
    FastAxes a;
    a.g[0] = lds;
    a.g[1] = lds + g.nx;
    a.g[2] = lds + g.nx + g.ny;
    a.inv[0] = lds + n;
    a.inv[1] = lds + n + g.nx;
    a.inv[2] = lds + n + g.nx + g.ny;
    for (int d = 0; d < 3; ++d) a.g0[d] = a.g[d][0];
    return a;
}

__device__ __forceinline__ void cell_fast(const double *g, const double *inv, int n, double g0, double ih, double x, int &i,
                                          double &t) {
    double f = (x - g0) * ih;
    f = fmin(fmax(f, 0.0), (double)(n - 2));
    i = (int)f;
    double a = g[i];
    const double b = g[i + 1];
    if (__builtin_expect(!((a < x) & (x <= b)), 0)) {      // guess off by one, or x on the clipped edge
        while (i > 0 && !(g[i] < x)) --i;
        while (i < n - 2 && g[i + 1] < x) ++i;
        a = g[i];
    }
    t = (x - a) * inv[i];
}

template <typename GT>
__device__ __forceinline__ double trilinear_fast(const GridView &g, const FastAxes &ax, double x, double y, double z) {
    int i, j, k;
    double tx, ty, tz;
    cell_fast(ax.g[0], ax.inv[0], g.nx, ax.g0[0], g.inv_h[0], x, i, tx);
    cell_fast(ax.g[1], ax.inv[1], g.ny, ax.g0[1], g.inv_h[1], y, j, ty);
    cell_fast(ax.g[2], ax.inv[2], g.nz, ax.g0[2], g.inv_h[2], z, k, tz);
    const unsigned sj = (unsigned)g.nz, si = (unsigned)g.ny * (unsigned)g.nz;
    const unsigned off = ((unsigned)i * (unsigned)g.ny + (unsigned)j) * sj + (unsigned)k;
    const GT *p = (const GT *)g.M + off;
    const double c000 = p[0], c001 = p[1];
    const double c010 = p[sj], c011 = p[sj + 1];
    const double c100 = p[si], c101 = p[si + 1];
    const double c110 = p[si + sj], c111 = p[si + sj + 1];
    const double c00 = c000 + tz * (c001 - c000);
    const double c01 = c010 + tz * (c011 - c010);
    const double c10 = c100 + tz * (c101 - c100);
    const double c11 = c110 + tz * (c111 - c110);
    const double c0 = c00 + ty * (c01 - c00);
    const double c1 = c10 + ty * (c11 - c10);
    return c0 + tx * (c1 - c0);
}

__device__ __forceinline__ bool ray_leaves_grid(const FastAxes &ax, const GridView &g, const StraightRay &q) {
    const double xe = q.ox + q.sx * q.L, ye = q.oy + q.sy * q.L, ze = q.oz + q.L;
    return outside(ax.g[0], g.nx, q.ox) || outside(ax.g[0], g.nx, xe) || outside(ax.g[1], g.ny, q.oy) ||
           outside(ax.g[1], g.ny, ye) || outside(ax.g[2], g.nz, q.oz) || outside(ax.g[2], g.nz, ze);
}

template <typename GT>
__global__ __launch_bounds__(256) void k_forward_straight_fast(GridView g, const double *__restrict__ origins,
                                                               const double *__restrict__ dirs, int64_t R, double tmax, int Ns,
                                                               const double *__restrict__ unitw, double *__restrict__ tec,
                                                               int *oob_flag) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const FastAxes ax = stage_axes_fast(g, lds);
    const int lane = threadIdx.x & 63;
    bool oob = false;
    for (RayWalk w = ray_walk(R); w.r < w.end; w.r += w.stride) {
        const StraightRay q = load_straight(origins, dirs, w.r, tmax, Ns);
        if (ray_leaves_grid(ax, g, q)) {
            oob = true;
            if (lane == 0) tec[w.r] = nan("");
            continue;
        }
        double acc = 0.0;
        for (int k = lane; k < Ns; k += 64) {
            double x, y, z;
            straight_point(q, k, Ns, x, y, z);
            acc += unitw[k] * trilinear_fast<GT>(g, ax, x, y, z);
        }
        acc = wave_sum(acc);
        if (lane == 0) tec[w.r] = acc * q.h;
    }
    if (__any(oob) && lane == 0) atomicOr(oob_flag, 1);
}

// ---- v2 fast path: "ideal uniform" grid coordinates ------------------------------------------------
// Taken when every axis equals g0 + i*h to within 2.5e-13 h (what np.linspace produces; checked on
// the host), so a sample's grid coordinate is ONE fma per axis, f = f0 + k*df, its cell is
// (int)f and its weight fract(f): no axis tables, no divisions in the loop.  Per-ray work that is
// wave-uniform in the kernels above (normalisation, slopes, bounds test on the two end points,
// the <= 8 tail samples when Ns is not a multiple of 64) is done LANE-PARALLEL for a group of up
// to 16 rays (lane = ray) and broadcast with v_readlane; the Simpson sum is a DPP row_shr /
// row_bcast reduction (no LDS round trips); the weight table lives in LDS.  `order` (optional)
// is a permutation of the rays giving the walk order (it matters for the adjoint's LDS
// pre-reduction; for this kernel it measured neutral).
#define U_MAXG 16

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}
// sum over the 64 lanes; the total is returned wave-uniform (read from lane 63)
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v = dpp_add<0x111, 0xf>(v);    // row_shr:1
    v = dpp_add<0x112, 0xf>(v);    // row_shr:2
    v = dpp_add<0x114, 0xf>(v);    // row_shr:4
    v = dpp_add<0x118, 0xf>(v);    // row_shr:8   -> lane 15 of each row holds the row total
    v = dpp_add<0x142, 0xa>(v);    // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v);    // row_bcast:31 into rows 2,3 -> lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast_lane(double v, int src) {     // src must be wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

struct URay {            // a straight ray in ideal grid coordinates: f(k) = f0 + k * df per axis
    double fx0, dfx, fy0, dfy, fz0, dfz, h;
    bool valid;
};
__device__ __forceinline__ URay load_uray(const GridView &g, const double *origins, const double *dirs, int64_t r, double tmax,
                                          int Ns) {
    const double ox = origins[3 * r], oy = origins[3 * r + 1], oz = origins[3 * r + 2];
    const double dx = dirs[3 * r], dy = dirs[3 * r + 1], dz = dirs[3 * r + 2];
    const double nrm = sqrt(dx * dx + dy * dy + dz * dz);
    const double px = dx / nrm, py = dy / nrm, pz = dz / nrm;
    const double sx = px / pz, sy = py / pz;
    const double L = tmax - oz;
    const double Lstep = L * (1.0 / (double)(Ns - 1));
    URay u;
    u.h = Lstep / pz;
    u.fx0 = (ox - g.g0[0]) * g.inv_h[0];
    u.fy0 = (oy - g.g0[1]) * g.inv_h[1];
    u.fz0 = (oz - g.g0[2]) * g.inv_h[2];
    u.dfx = sx * Lstep * g.inv_h[0];
    u.dfy = sy * Lstep * g.inv_h[1];
    u.dfz = Lstep * g.inv_h[2];
    const double xe = ox + sx * L, ye = oy + sy * L, ze = oz + L;
    u.valid = (ox >= g.g0[0]) & (ox <= g.glast[0]) & (xe >= g.g0[0]) & (xe <= g.glast[0]) & (oy >= g.g0[1]) &
              (oy <= g.glast[1]) & (ye >= g.g0[1]) & (ye <= g.glast[1]) & (oz >= g.g0[2]) & (oz <= g.glast[2]) &
              (ze >= g.g0[2]) & (ze <= g.glast[2]);
    return u;
}

// The grid allocation is padded by one plane + one row + 2 zero elements (iono_grid_set), so a
// sample sitting exactly on the top face of an axis (cell index n-1, weight 0 on the far corner)
// may read the far corner without a clamp: it is multiplied by 0.
template <typename GT>
struct Corners {
    GT c000, c001, c010, c011, c100, c101, c110, c111;
    double tx, ty, tz;
};
template <typename GT>
__device__ __forceinline__ Corners<GT> load_corners(const GT *__restrict__ b00, const GT *__restrict__ b01,
                                                    const GT *__restrict__ b10, const GT *__restrict__ b11, int ny, int nz,
                                                    double fx, double fy, double fz) {
    const int i = (int)fx, j = (int)fy, k = (int)fz;
    Corners<GT> c;
    c.tx = fx - (double)i;
    c.ty = fy - (double)j;
    c.tz = fz - (double)k;
    const unsigned boff = (((unsigned)i * (unsigned)ny + (unsigned)j) * (unsigned)nz + (unsigned)k) * (unsigned)sizeof(GT);
    const GT *p00 = (const GT *)((const char *)b00 + boff), *p01 = (const GT *)((const char *)b01 + boff);
    const GT *p10 = (const GT *)((const char *)b10 + boff), *p11 = (const GT *)((const char *)b11 + boff);
    c.c000 = p00[0];
    c.c001 = p00[1];
    c.c010 = p01[0];
    c.c011 = p01[1];
    c.c100 = p10[0];
    c.c101 = p10[1];
    c.c110 = p11[0];
    c.c111 = p11[1];
    return c;
}
template <typename GT>
__device__ __forceinline__ double lerp_corners(const Corners<GT> &c) {
    const double c000 = c.c000, c010 = c.c010, c100 = c.c100, c110 = c.c110;
    const double c00 = c000 + c.tz * ((double)c.c001 - c000);
    const double c01 = c010 + c.tz * ((double)c.c011 - c010);
    const double c10 = c100 + c.tz * ((double)c.c101 - c100);
    const double c11 = c110 + c.tz * ((double)c.c111 - c110);
    const double c0 = c00 + c.ty * (c01 - c00);
    const double c1 = c10 + c.ty * (c11 - c10);
    return c0 + c.tx * (c1 - c0);
}
template <typename GT>
__device__ __forceinline__ double trilinear_u(const GT *__restrict__ b00, const GT *__restrict__ b01,
                                              const GT *__restrict__ b10, const GT *__restrict__ b11, int ny, int nz, double fx,
                                              double fy, double fz) {
    return lerp_corners<GT>(load_corners<GT>(b00, b01, b10, b11, ny, nz, fx, fy, fz));
}

// Every wave owns one contiguous, balanced chunk of the walk (floor or ceil of R / #waves rays) and
// goes through it in groups of up to U_MAXG rays; waves are numbered XCD-major (blocks b, b+8, ...
// share an XCD), so each XCD's L2 sees one contiguous eighth of the rays.  The grid is sized to
// what is resident at once, so there is no second, under-occupied round of workgroups.
struct Chunk {
    int64_t lo, hi, stride;     // walk positions lo, lo+stride, ... < hi
};
// mode 0 (default): one contiguous chunk per wave.  mode bit 0: one contiguous chunk per WORKGROUP, its
// 4 waves interleaved (wave w takes lo+w, lo+w+4, ...).  mode bit 2: plain block order instead of
// XCD-major.  Both alternatives measured slower or equal on the bench workload; kept for A/B runs
// (env IONOTOMO_WALK).
__device__ __forceinline__ Chunk wave_chunk(int64_t R, int mode) {
    const int wpb = blockDim.x >> 6, wid = threadIdx.x >> 6;
    int64_t bidx;
    if ((gridDim.x & 7) == 0 && !(mode & 4)) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
        bidx = (int64_t)xcd * nslot + slot;
    } else {
        bidx = blockIdx.x;
    }
    Chunk c;
    if (mode & 1) {
        const int64_t nb = gridDim.x, base = R / nb, rem = R % nb;
        const int64_t lo = bidx * base + min(bidx, rem);
        c.hi = lo + base + (bidx < rem ? 1 : 0);
        c.lo = lo + wid;
        c.stride = wpb;
    } else {
        const int64_t widx = bidx * wpb + wid, nw = (int64_t)gridDim.x * wpb;
        const int64_t base = R / nw, rem = R % nw;
        c.lo = widx * base + min(widx, rem);
        c.hi = c.lo + base + (widx < rem ? 1 : 0);
        c.stride = 1;
    }
    return c;
}

template <typename GT>
__global__ __launch_bounds__(256) void k_forward_straight_u(GridView g, const double *__restrict__ origins,
                                                            const double *__restrict__ dirs, const int *__restrict__ order,
                                                            int64_t R, double tmax, int Ns, int walk_mode,
                                                            const double *__restrict__ unitw, double *__restrict__ tec,
                                                            int *oob_flag) {
    extern __shared__ __attribute__((aligned(16))) double wlds[];
    for (int t = threadIdx.x; t < Ns; t += blockDim.x) wlds[t] = unitw[t];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int nfull = Ns >> 6, ntail0 = nfull << 6;        // samples [ntail0, Ns) are the tail
    const bool tail_by_lane = (Ns - ntail0) <= 8;          // else: one more (masked) wave iteration
    const GT *b00 = (const GT *)g.M, *b01 = b00 + g.nz, *b10 = b00 + (size_t)g.ny * g.nz, *b11 = b10 + g.nz;
    const Chunk ch = wave_chunk(R, walk_mode);
    const double dlane = (double)lane;
    const double *wp = wlds + lane;
    bool oob = false;
    for (int64_t q0 = ch.lo; q0 < ch.hi; q0 += U_MAXG * ch.stride) {
        const int cnt = (int)min((int64_t)U_MAXG, (ch.hi - q0 + ch.stride - 1) / ch.stride);
        // ---- lane-parallel set-up: lane l owns ray q0 + l ------------------------------------------
        URay u = {};
        int64_t r = 0;
        double tail = 0.0;
        if (lane < cnt) {
            const int64_t q = q0 + lane * ch.stride;
            r = order ? (int64_t)order[q] : q;
            u = load_uray(g, origins, dirs, r, tmax, Ns);
            if (u.valid && tail_by_lane) {
                for (int k = ntail0; k < Ns; ++k) {
                    const double kd = (double)k;
                    tail += wlds[k] * trilinear_u<GT>(b00, b01, b10, b11, g.ny, g.nz, fma(kd, u.dfx, u.fx0),
                                                       fma(kd, u.dfy, u.fy0), fma(kd, u.dfz, u.fz0));
                }
            }
            if (!u.valid) oob = true;
        }
        // ---- one ray at a time, lanes = samples ----------------------------------------------------
        double res = 0.0;
        for (int gi = 0; gi < cnt; ++gi) {
            const int ok = __builtin_amdgcn_readlane((int)u.valid, gi);
            if (!ok) continue;
            const double dfx = bcast_lane(u.dfx, gi), dfy = bcast_lane(u.dfy, gi), dfz = bcast_lane(u.dfz, gi);
            double fx = fma(dlane, dfx, bcast_lane(u.fx0, gi));
            double fy = fma(dlane, dfy, bcast_lane(u.fy0, gi));
            double fz = fma(dlane, dfz, bcast_lane(u.fz0, gi));
            const double sx64 = 64.0 * dfx, sy64 = 64.0 * dfy, sz64 = 64.0 * dfz;
            double acc = 0.0;
            // (a software-pipelined version of this loop -- next iteration's loads in flight during the
            //  interpolation -- measured 15 % SLOWER: +26 VGPRs cost more occupancy than the overlap won)
            for (int it = 0; it < nfull; ++it) {
                acc = fma(wp[it << 6], trilinear_u<GT>(b00, b01, b10, b11, g.ny, g.nz, fx, fy, fz), acc);
                fx += sx64;
                fy += sy64;
                fz += sz64;
            }
            if (!tail_by_lane && lane + ntail0 < Ns)
                acc = fma(wp[ntail0], trilinear_u<GT>(b00, b01, b10, b11, g.ny, g.nz, fx, fy, fz), acc);
            const double total = wave_sum_dpp(acc);
            if (lane == gi) res = total;
        }
        if (lane < cnt) tec[r] = u.valid ? (res + tail) * u.h : nan("");
    }
    if (__any(oob) && lane == 0) atomicOr(oob_flag, 1);
}

template <typename GT, int KIND>
__global__ __launch_bounds__(256) void k_forward_rays(GridView g, const double *__restrict__ rays, int64_t R, int Ns,
                                                      int rule, double *__restrict__ tec, int *oob_flag) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const Axes ax = stage_axes(g, lds);
    const int lane = threadIdx.x & 63;
    bool oob = false;
    for (RayWalk w = ray_walk(R); w.r < w.end; w.r += w.stride) {
        const double *rx = rays + (size_t)w.r * 4 * Ns, *ry = rx + Ns, *rz = ry + Ns, *rs = rz + Ns;
        double acc = 0.0;
        for (int k = lane; k < Ns; k += 64) {
            const double x = rx[k], y = ry[k], z = rz[k];
            if (sample_outside<KIND>(ax, x, y, z)) {
                oob = true;
                continue;
            }
            acc += quad_weight(rs, Ns, k, rule) * sample_at<GT, KIND>(g, ax, x, y, z);
        }
        acc = wave_sum(acc);
        if (lane == 0) tec[w.r] = acc;
    }
    if (__any(oob) && lane == 0) atomicOr(oob_flag, 1);
}

// phase observable, per-frequency integrals of 1 - sqrt(1 - ne/n_p) (inversion/iterative_newton.py:108-119)
template <typename GT, int MAXF>
__global__ __launch_bounds__(256) void k_forward_phase_rays(GridView g, const double *__restrict__ rays, int64_t R, int Ns,
                                                            int rule, const double *__restrict__ inv_np, int nf, int ldf,
                                                            double *__restrict__ phi, int *oob_flag) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const Axes ax = stage_axes(g, lds);
    const int lane = threadIdx.x & 63;
    bool oob = false;
    double inp[MAXF];
#pragma unroll
    for (int l = 0; l < MAXF; ++l) inp[l] = l < nf ? inv_np[l] : 0.0;
    for (RayWalk w = ray_walk(R); w.r < w.end; w.r += w.stride) {
        const double *rx = rays + (size_t)w.r * 4 * Ns, *ry = rx + Ns, *rz = ry + Ns, *rs = rz + Ns;
        double acc[MAXF];
#pragma unroll
        for (int l = 0; l < MAXF; ++l) acc[l] = 0.0;
        for (int k = lane; k < Ns; k += 64) {
            const double x = rx[k], y = ry[k], z = rz[k];
            if (sample_outside<IONO_INTERP_TRILINEAR>(ax, x, y, z)) {
                oob = true;
                continue;
            }
            const double ne = trilinear_at<GT>(g, ax, x, y, z);
            const double c = quad_weight(rs, Ns, k, rule);
#pragma unroll
            for (int l = 0; l < MAXF; ++l) acc[l] += c * (1.0 - sqrt(1.0 - ne * inp[l]));
        }
#pragma unroll
        for (int l = 0; l < MAXF; ++l) {
            const double v = wave_sum(acc[l]);
            if (lane == 0 && l < nf) phi[(size_t)w.r * ldf + l] = v;
        }
    }
    if (__any(oob) && lane == 0) atomicOr(oob_flag, 1);
}

// g = const_i + 2 pi nu clock_ij - (phi - phi[i0]) 2 pi nu / c   (inversion/iterative_newton.py:107-123)
__global__ void k_phase_finish(const double *__restrict__ phi, const double *__restrict__ freqs,
                               const double *__restrict__ clock, const double *__restrict__ cst, int Na, int Nt, int Nd,
                               int Nf, int i0, double *__restrict__ gout) {
    const int64_t n = (int64_t)Na * Nt * Nd * Nf;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int l = idx % Nf;
        const int64_t r = idx / Nf;
        const int64_t td = r % ((int64_t)Nt * Nd);
        const int a = r / ((int64_t)Nt * Nd);
        const int t = td / Nd;
        const double a_ = 2.0 * M_PI * freqs[l];
        const double ph = (phi[r * Nf + l] - phi[((int64_t)i0 * Nt * Nd + td) * Nf + l]) * (a_ / SPEED_OF_LIGHT);
        gout[idx] = cst[a] + a_ * clock[(int64_t)a * Nt + t] - ph;
    }
}

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
template <typename AT, int MODE>
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
            const double tref = tec[(int64_t)i0 * NtNd + p];
            wr = (tec[w.r] - tref - dobs[w.r]) / (cdct[w.r] + 1e-15);
            if (a == i0) {
                double s = 0.0;
                for (int a2 = lane; a2 < Na; a2 += 64) {
                    const int64_t r2 = (int64_t)a2 * NtNd + p;
                    s += (tec[r2] - tref - dobs[r2]) / (cdct[r2] + 1e-15);
                }
                wr -= wave_sum(s);
            }
        }
        if (wr == 0.0) continue;
        const StraightRay q = load_straight(origins, dirs, w.r, tmax, Ns);
        const double scale = wr * q.h;
        for (int k = lane; k < Ns; k += 64) {
            double x, y, z;
            straight_point(q, k, Ns, x, y, z);
            if (sample_outside<IONO_INTERP_TRILINEAR>(ax, x, y, z)) {
                oob = true;
                continue;
            }
            scatter_trilinear<AT>(g, ax, G, x, y, z, scale * unitw[k]);
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

template <typename AT>
__device__ __forceinline__ void tile_or_global_add(AT *tile, AT *__restrict__ G, const int *I0, const int *J0, int m, int i, int j,
                                                   int kk, int ny, int nz, double w00, double w01, double w10, double w11,
                                                   int dbg = 0) {
    // the four (i..i+1, j..j+1) nodes of z level kk (tile level m); tile if the 2x2 patch is inside the window
    bool in = (m >= 0) & (m < T_TK);
    int a = 0, b = 0;
    if (in) {
        a = i - I0[m];
        b = j - J0[m];
        in = (a >= 0) & (a + 1 < T_WIN) & (b >= 0) & (b + 1 < T_WIN);
    }
    if (in) {
        AT *t = tile + (a * T_WIN + b) * T_TKP + m;
        atomicAdd(t, (AT)w00);
        atomicAdd(t + T_TKP, (AT)w01);
        atomicAdd(t + T_WIN * T_TKP, (AT)w10);
        atomicAdd(t + (T_WIN + 1) * T_TKP, (AT)w11);
    } else if (!(dbg & 4)) {
        AT *p = G + ((size_t)i * ny + j) * nz + kk;
        atomicAdd(p, (AT)w00);
        atomicAdd(p + nz, (AT)w01);
        atomicAdd(p + (size_t)ny * nz, (AT)w10);
        atomicAdd(p + (size_t)ny * nz + nz, (AT)w11);
    }
}

template <typename AT>
__device__ __forceinline__ void scatter_sample_tiled(const GridView &g, AT *tile, AT *__restrict__ G, const int *I0, const int *J0,
                                                     int kz0, double fx, double fy, double fz, double c, int dbg = 0) {
    const int i = min((int)fx, g.nx - 2), j = min((int)fy, g.ny - 2), k = min((int)fz, g.nz - 2);
    const double tx = fx - (double)i, ty = fy - (double)j, tz = fz - (double)k;
    const double w0 = c * (1 - tx), w1 = c * tx;
    const double w00 = w0 * (1 - ty), w01 = w0 * ty, w10 = w1 * (1 - ty), w11 = w1 * ty;
    const int m = k - kz0;
    tile_or_global_add<AT>(tile, G, I0, J0, m, i, j, k, g.ny, g.nz, w00 * (1 - tz), w01 * (1 - tz), w10 * (1 - tz), w11 * (1 - tz), dbg);
    tile_or_global_add<AT>(tile, G, I0, J0, m + 1, i, j, k + 1, g.ny, g.nz, w00 * tz, w01 * tz, w10 * tz, w11 * tz, dbg);
}

// residual -> differential weight of ray r = (a, p) in layout [Na][NtNd] (see k_adjoint_straight MODE 1)
__device__ __forceinline__ double residual_weight(const double *__restrict__ tec, const double *__restrict__ dobs,
                                                  const double *__restrict__ cdct, int Na, int64_t NtNd, int i0, int64_t r) {
    const int a = (int)(r / NtNd);
    const int64_t p = r % NtNd;
    const double tref = tec[(int64_t)i0 * NtNd + p];
    double wr = (tec[r] - tref - dobs[r]) / (cdct[r] + 1e-15);
    if (a == i0) {
        double s = 0.0;
        for (int a2 = 0; a2 < Na; ++a2) {
            const int64_t r2 = (int64_t)a2 * NtNd + p;
            s += (tec[r2] - tref - dobs[r2]) / (cdct[r2] + 1e-15);
        }
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

struct AdjRay {
    URay u;
    double scale;
};
// lane-parallel load of `q` rays per wave starting at walk position qw (lanes >= cnt idle)
template <int MODE>
__device__ __forceinline__ AdjRay load_adj_ray(const GridView &g, const double *origins, const double *dirs, const int *order,
                                               const double *wray, const double *tec, const double *dobs, const double *cdct,
                                               int Na, int64_t NtNd, int i0, int64_t q, bool active, double tmax, int Ns,
                                               bool &oob) {
    AdjRay a;
    a.u = URay{};
    a.scale = 0.0;
    if (active) {
        const int64_t r = order ? (int64_t)order[q] : q;
        a.u = load_uray(g, origins, dirs, r, tmax, Ns);
        const double wr = MODE == 0 ? wray[r] : residual_weight(tec, dobs, cdct, Na, NtNd, i0, r);
        if (a.u.valid) a.scale = wr * a.u.h; else oob = true;
    }
    return a;
}

template <typename AT, int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void k_adjoint_straight_tile(GridView g, const double *__restrict__ origins,
                                                               const double *__restrict__ dirs, const int *__restrict__ order,
                                                               const double *__restrict__ wray, const double *__restrict__ tec,
                                                               const double *__restrict__ dobs, const double *__restrict__ cdct,
                                                               int Na, int64_t NtNd, int i0, int64_t R, double tmax, int Ns,
                                                               int dbg, const double *__restrict__ unitw, AT *__restrict__ G,
                                                               int *oob_flag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *wlds = (double *)smem;                                   // [Ns] quadrature weights
    double *ref = wlds + ((Ns + 1) & ~1);                            // [NW waves][16] per-wave sums and bounding boxes
    AT *tile = (AT *)(ref + 16 * NW);                                     // [T_WIN*T_WIN][T_TKP]
    int *I0 = (int *)(tile + T_WIN * T_WIN * T_TKP);                 // [T_TK] window origins per z level
    int *J0 = I0 + T_TK;
    for (int t = threadIdx.x; t < Ns; t += blockDim.x) wlds[t] = unitw[t];
    for (int t = threadIdx.x; t < T_WIN * T_WIN * T_TKP; t += blockDim.x) tile[t] = (AT)0;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nfull = Ns >> 6, ntail0 = nfull << 6;
    const bool tail_by_lane = (Ns - ntail0) <= 8;
    const int nslab = tail_by_lane ? nfull : nfull + 1;
    const double klast = (double)(Ns - 1);
    // contiguous balanced range of the walk per workgroup (XCD-major)
    int64_t bidx = blockIdx.x;
    if ((gridDim.x & 7) == 0) bidx = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int64_t base = R / gridDim.x, rem = R % gridDim.x;
    const int64_t lo = bidx * base + min(bidx, rem), hi = lo + base + (bidx < rem ? 1 : 0);
    const double BIG = 1e300;
    bool oob = false;
    __syncthreads();
    for (int64_t q0 = lo; q0 < hi;) {
        // ---- candidate bundle: up to 64 rays, wave w lanes 0..15 own walk positions q0 + 16 w + l ----------
        int q = 16;                                     // rays per wave
        int64_t qw = q0 + (int64_t)q * wid;
        int cnt = (int)max((int64_t)0, min((int64_t)q, hi - qw));
        AdjRay a = load_adj_ray<MODE>(g, origins, dirs, order, wray, tec, dobs, cdct, Na, NtNd, i0, qw + lane, lane < cnt, tmax,
                                      Ns, oob);
        int c = 16 * NW;
        for (int round = 0; round < 2; ++round) {
            // per-wave sums (for the mean ray) and bounding boxes at the bottom / top of the rays
            const bool lv = a.scale != 0.0;
            const double live = lv ? 1.0 : 0.0;
            const double xe = fma(klast, a.u.dfx, a.u.fx0), ye = fma(klast, a.u.dfy, a.u.fy0);
            const double s4 = wave_sum_dpp(live * a.u.fz0), s5 = wave_sum_dpp(live * a.u.dfz), s7 = wave_sum_dpp(live);
            double b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0;
            {
                b0 = wave_minmax_dpp<false>(lv ? a.u.fx0 : BIG);
                b1 = wave_minmax_dpp<true>(lv ? a.u.fx0 : -BIG);
                b2 = wave_minmax_dpp<false>(lv ? a.u.fy0 : BIG);
                b3 = wave_minmax_dpp<true>(lv ? a.u.fy0 : -BIG);
                b4 = wave_minmax_dpp<false>(lv ? xe : BIG);
                b5 = wave_minmax_dpp<true>(lv ? xe : -BIG);
                b6 = wave_minmax_dpp<false>(lv ? ye : BIG);
                b7 = wave_minmax_dpp<true>(lv ? ye : -BIG);
            }
            if (lane == 0) {
                double *rp = ref + 16 * wid;
                rp[4] = s4, rp[5] = s5, rp[7] = s7;
                rp[8] = b0, rp[9] = b1, rp[10] = b2, rp[11] = b3, rp[12] = b4, rp[13] = b5, rp[14] = b6, rp[15] = b7;
            }
            __syncthreads();
            if (round == 1) break;
            // largest c in {64, 32, 16} whose rays stay within the tile window at both ends (block-uniform)
            const double lim = (double)(T_WIN - 3);
            double m0 = BIG, M0 = -BIG, m1 = BIG, M1 = -BIG, m2 = BIG, M2 = -BIG, m3 = BIG, M3 = -BIG;
            int fit = 0;
            for (int w2 = 0; w2 < NW; ++w2) {
                const double *rp = ref + 16 * w2;
                m0 = fmin(m0, rp[8]), M0 = fmax(M0, rp[9]), m1 = fmin(m1, rp[10]), M1 = fmax(M1, rp[11]);
                m2 = fmin(m2, rp[12]), M2 = fmax(M2, rp[13]), m3 = fmin(m3, rp[14]), M3 = fmax(M3, rp[15]);
                const bool ok = (M0 - m0 <= lim) & (M1 - m1 <= lim) & (M2 - m2 <= lim) & (M3 - m3 <= lim);
                if (ok && ((w2 + 1) & w2) == 0) fit = w2 + 1;           // 1, 2, 4 (, 8) waves' worth of rays
            }
            c = 16 * max(fit, 1);
            if (c == 16 * NW) break;
            // spread too wide: shrink the bundle and re-deal its rays evenly over the four waves
            __syncthreads();
            q = c / NW;
            qw = q0 + (int64_t)q * wid;
            cnt = (int)max((int64_t)0, min((int64_t)q, hi - qw));
            a = load_adj_ray<MODE>(g, origins, dirs, order, wray, tec, dobs, cdct, Na, NtNd, i0, qw + lane, lane < cnt, tmax, Ns,
                                   oob);
        }
        q0 += c;
        if (a.scale != 0.0 && tail_by_lane) {               // the <= 8 tail samples: straight to global memory
            for (int k = ntail0; k < Ns; ++k) {
                const double kd = (double)k;
                scatter_sample_tiled<AT>(g, tile, G, I0, J0, -(1 << 28), fma(kd, a.u.dfx, a.u.fx0), fma(kd, a.u.dfy, a.u.fy0),
                                         fma(kd, a.u.dfz, a.u.fz0), a.scale * wlds[k]);
            }
        }
        double nlive = 0.0, sz0 = 0.0, sdz = 0.0;
        for (int w2 = 0; w2 < NW; ++w2) nlive += ref[16 * w2 + 7], sz0 += ref[16 * w2 + 4], sdz += ref[16 * w2 + 5];
        if (nlive == 0.0) {            // nothing to do in this bundle (block-uniform)
            __syncthreads();
            continue;
        }
        // reference line of the bundle: through the centres of its bounding boxes at the bottom and at the
        // top (a bundle that passed the spread test then lies entirely inside the windows); z from the mean
        const double inl = 1.0 / nlive;
        double bx0 = BIG, bx1 = -BIG, by0 = BIG, by1 = -BIG, tx0 = BIG, tx1 = -BIG, ty0 = BIG, ty1 = -BIG;
        for (int w2 = 0; w2 < NW; ++w2) {
            const double *rp = ref + 16 * w2;
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
            __syncthreads();
            for (int gi = 0; gi < cnt; ++gi) {
                const double sc = bcast_lane(a.scale, gi);
                if (sc == 0.0) continue;
                const int k = k0 + lane;
                if (k < Ns && (tail_by_lane ? k < ntail0 : true)) {
                    const double kd = (double)k;
                    scatter_sample_tiled<AT>(g, tile, G, I0, J0, kz0, fma(kd, bcast_lane(a.u.dfx, gi), bcast_lane(a.u.fx0, gi)),
                                             fma(kd, bcast_lane(a.u.dfy, gi), bcast_lane(a.u.fy0, gi)),
                                             fma(kd, bcast_lane(a.u.dfz, gi), bcast_lane(a.u.fz0, gi)), sc * wlds[k], dbg);
                }
            }
            __syncthreads();
            // ---- flush + re-zero: one global atomic per touched node -------------------------------------
            for (int e = threadIdx.x; e < T_WIN * T_WIN * T_TKP; e += blockDim.x) {
                const AT v = tile[e];
                if (v != (AT)0) {
                    tile[e] = (AT)0;
                    const int cell = e / T_TKP, m = e - cell * T_TKP;
                    const int gi_ = I0[m] + cell / T_WIN, gj_ = J0[m] + cell % T_WIN, gk_ = kz0 + m;
                    if (m < T_TK && gi_ >= 0 && gi_ < g.nx && gj_ >= 0 && gj_ < g.ny && gk_ < g.nz && !(dbg & 8))
                        atomicAdd(G + ((size_t)gi_ * g.ny + gj_) * g.nz + gk_, v);
                }
            }
            __syncthreads();
        }
    }
    if (__any(oob) && lane == 0) atomicOr(oob_flag, 1);
}

template <typename AT>
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
            if (sample_outside<IONO_INTERP_TRILINEAR>(ax, x, y, z)) {
                oob = true;
                continue;
            }
            scatter_trilinear<AT>(g, ax, G, x, y, z, wr * quad_weight(rs, Ns, k, rule));
        }
    }
    if (__any(oob) && lane == 0) atomicOr(oob_flag, 1);
}

// ------------------------------------------------------------------------------------------------
// small elementwise / geometry kernels
// ------------------------------------------------------------------------------------------------
template <typename GT>
__global__ void k_set_values(const double *__restrict__ src, GT *__restrict__ dst, int64_t n, int do_exp, double scale,
                             int *nonfinite) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double v = src[i];
        if (do_exp) v = exp(v) * scale;
        if (!isfinite(v)) bad = true;
        dst[i] = (GT)v;
    }
    if (bad) atomicOr(nonfinite, 1);
}

template <typename GT>
__global__ void k_get_values(const GT *__restrict__ src, double *__restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (double)src[i];
}

// n = sqrt(1 - 8.980^2 ne / nu^2) at the nodes (inversion/fermat.py:36-46)
template <typename GT>
__global__ void k_ne_to_n(const GT *__restrict__ ne, double *__restrict__ nM, int64_t n, double freq) {
    const double A = -PLASMA_A / (freq * freq);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        nM[i] = sqrt(1.0 + (double)ne[i] * A);
}

template <typename AT, typename GT>
__global__ void k_scale_by_grid(AT *__restrict__ G, const GT *__restrict__ M, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        G[i] = (AT)((double)G[i] * (double)M[i]);
}

__global__ void k_subtract_reference(double *__restrict__ tec, int Na, int64_t NtNd, int i0) {
    // rows other than i0 first (they read row i0), row i0 is zeroed by a second launch
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (int64_t)Na * NtNd;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int a = idx / NtNd;
        if (a != i0) tec[idx] -= tec[(int64_t)i0 * NtNd + idx % NtNd];
    }
}
__global__ void k_zero(double *__restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.0;
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

__global__ void k_trace_straight(const double *__restrict__ origins, const double *__restrict__ dirs, int64_t R, double tmax,
                                 int Ns, double *__restrict__ rays) {
    const int64_t n = R * Ns;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / Ns;
        const int k = idx % Ns;
        const StraightRay q = load_straight(origins, dirs, r, tmax, Ns);
        double x, y, z;
        straight_point(q, k, Ns, x, y, z);
        double *o = rays + (size_t)r * 4 * Ns;
        o[k] = x;
        o[Ns + k] = y;
        o[2 * Ns + k] = z;
        const double frac = (k == Ns - 1) ? 1.0 : (double)k * q.step;
        o[3 * Ns + k] = q.L * frac / q.pz;     // s = (z - z0)/pz
    }
}

// Fermat ray ODE in z (inversion/fermat.py:64-72; notebooks/FermatClass.ipynb c0:76-84):
//   s' = n/pz, p' = grad(n) n/pz, x' = px/pz, y' = py/pz, z' = 1.   Lanes = rays, RK4.
struct FState {
    double px, py, pz, x, y, z, s;
};
template <int KIND, bool BEND>
__device__ __forceinline__ FState fermat_rhs(const GridView &g, const double *nM, const FState &u) {
    double n, nx, ny, nz;
    if (KIND == IONO_INTERP_TRILINEAR) {
        trilinear_grad_at(g, nM, u.x, u.y, u.z, n, nx, ny, nz);
    } else {
        GridView gn = g;
        gn.M = nM;
        tricubic_eval<double, true>(gn, g.axes, g.axes + g.nx, g.axes + g.nx + g.ny, u.x, u.y, u.z, n, nx, ny, nz);
    }
    if (!BEND) nx = ny = nz = 0.0;
    const double f = n / u.pz;
    FState d;
    d.px = nx * f;
    d.py = ny * f;
    d.pz = nz * f;
    d.x = u.px / u.pz;
    d.y = u.py / u.pz;
    d.z = 1.0;
    d.s = f;
    return d;
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
                                                     int *oob_flag) {
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
    const double h = (tmax - u.z) / (double)((Ns - 1) * substeps);
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
                kprev = fermat_rhs<KIND, BEND>(g, nM, axpy(u, ca, kprev));
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
template <int CTRL>
__device__ __forceinline__ double dpp_xadd(double v) {
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
template <bool BEND>
__device__ __forceinline__ FState fermat_rhs_coop(const GridView &g, const double *__restrict__ nM, const FState &u, int sub) {
    const double *gx = g.axes, *gy = g.axes + g.nx, *gz = g.axes + g.nx + g.ny;
    double wx[6], wy[6], wz[6], dx[6], dy[6], dz[6];
    const int i = cubic_axis(gx, g.nx, u.x, g.inv_h[0], g.uniform[0], wx, dx, true);
    const int j = cubic_axis(gy, g.ny, u.y, g.inv_h[1], g.uniform[1], wy, dy, true);
    const int k = cubic_axis(gz, g.nz, u.z, g.inv_h[2], g.uniform[2], wz, dz, true);
    double wxa, dxa;
    pick_tap(wx, dx, sub, wxa, dxa);
    const int a = min(sub, 5);
    const double *base = nM + ((size_t)(i - 2 + a) * g.ny + (j - 2)) * g.nz + (k - 2);
    double fa = 0.0, fya = 0.0, fza = 0.0;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const double *p = base + (size_t)b * g.nz;
        double sv = 0.0, sz = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double v = p[c];
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
    const double f = n / u.pz;
    FState d;
    d.px = nx * f;
    d.py = ny * f;
    d.pz = nz * f;
    d.x = u.px / u.pz;
    d.y = u.py / u.pz;
    d.z = 1.0;
    d.s = f;
    return d;
}
template <bool BEND>
__global__ __launch_bounds__(64) void k_trace_fermat_coop(GridView g, const double *__restrict__ nM,
                                                          const double *__restrict__ origins, const double *__restrict__ dirs,
                                                          int64_t R, double tmax, int Ns, int substeps, double *__restrict__ rays,
                                                          int *oob_flag) {
    const int sub = threadIdx.x & 7;
    int64_t r = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 3);
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
    const double h = (tmax - u.z) / (double)((Ns - 1) * substeps);
    double *o = rays + (size_t)r * 4 * Ns;
    const bool writer = live && sub == 0;
    if (writer) {
        o[0] = u.x;
        o[Ns] = u.y;
        o[2 * Ns] = u.z;
        o[3 * Ns] = u.s;
    }
    const double *gx = g.axes, *gy = g.axes + g.nx, *gz = g.axes + g.nx + g.ny;
    bool oob = false;
    for (int k = 1; k < Ns; ++k) {
        for (int s2 = 0; s2 < substeps; ++s2) {
            FState kprev = {}, sum = {};
#pragma unroll 1
            for (int st = 0; st < 4; ++st) {
                const double ca = st == 0 ? 0.0 : (st == 3 ? h : 0.5 * h);
                kprev = fermat_rhs_coop<BEND>(g, nM, axpy(u, ca, kprev), sub);
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

}  // namespace

// ================================================================================================
// host side
// ================================================================================================
struct iono_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    int nx = 0, ny = 0, nz = 0, storage = IONO_F64;
    double *d_axes = nullptr;
    void *d_M = nullptr;
    double inv_h[3] = {0, 0, 0};
    int uniform[3] = {0, 0, 0};
    int *d_flags = nullptr;          // [0] out-of-bounds, [1] non-finite
    double *d_unitw = nullptr;       // cached unit-spacing quadrature weights
    int unitw_n = 0, unitw_rule = -1;
    std::string err;
    int num_cus = 256;
    int force_general = 0;           // testing/ablation: 1 = general kernels only, 2 = no "ideal uniform" kernels
    void *d_work = nullptr;          // workspace of the host-pointer entry points (grow-only)
    size_t work_cap = 0;
    double *d_kern = nullptr;        // 3 x (2h+1) smoothing kernels
    int kern_cap = 0;
    double *d_nM = nullptr;          // refractive-index nodes for the tracer (device-pointer entry), lazily built
    double nM_freq = -1.0;           // frequency d_nM was built for; < 0 = stale
    int variant = 0;                 // kernel variant for A/B runs (env IONOTOMO_VARIANT)
    int blocks_per_cu_override = 0;  // env IONOTOMO_BLOCKS_PER_CU
    int walk_mode = 0;               // env IONOTOMO_WALK (see wave_chunk)
    int ideal = 0;                   // every axis is g0 + i*h to within 2.5e-13 h (np.linspace)
    double g0[3] = {0, 0, 0}, glast[3] = {0, 0, 0};
};

namespace {

int fail(iono_ctx *c, int code, const std::string &msg) {
    g_last_error = msg;
    if (c) c->err = msg;
    return code;
}
#define HIP_TRY(c, expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(c, IONO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

// Device workspace of the host-pointer entry points: one grow-only buffer per ctx instead of a
// hipMalloc/hipFree pair per call (those cost more than the kernels at config-2 sizes).  Host entry
// points are synchronous and a ctx is single-threaded, so the buffer is free again when they return.
struct DevBuf {
    iono_ctx *c;
    void *p = nullptr;
    explicit DevBuf(iono_ctx *ctx) : c(ctx) {}
    hipError_t alloc(size_t bytes);
    template <typename T> T *as() { return (T *)p; }
};

GridView view(const iono_ctx *c) {
    GridView g;
    g.axes = c->d_axes;
    g.M = c->d_M;
    g.nx = c->nx;
    g.ny = c->ny;
    g.nz = c->nz;
    for (int a = 0; a < 3; ++a) {
        g.inv_h[a] = c->inv_h[a];
        g.uniform[a] = c->uniform[a];
        g.g0[a] = c->g0[a];
        g.glast[a] = c->glast[a];
    }
    return g;
}
hipError_t DevBuf::alloc(size_t bytes) {
    if (bytes > c->work_cap) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return e;
        if (c->d_work) (void)hipFree(c->d_work);
        c->d_work = nullptr;
        c->work_cap = 0;
        const size_t cap = bytes + bytes / 4 + 4096;
        e = hipMalloc(&c->d_work, cap);
        if (e != hipSuccess) return e;
        c->work_cap = cap;
    }
    p = c->d_work;
    return hipSuccess;
}

size_t lds_bytes(const iono_ctx *c) { return sizeof(double) * (size_t)(c->nx + c->ny + c->nz); }
int64_t ncells(const iono_ctx *c) { return (int64_t)c->nx * c->ny * c->nz; }
// fast kernels: all three axes uniform (cell guess off by at most one), 32-bit element offsets
bool fast_path_ok(const iono_ctx *c) {
    // 32-bit BYTE offsets (SGPR base + VGPR offset addressing), padded far-corner reads included
    const uint64_t bytes = ((uint64_t)ncells(c) + (uint64_t)c->ny * c->nz + c->nz + 2) * (c->storage == IONO_F64 ? 8 : 4);
    return c->uniform[0] && c->uniform[1] && c->uniform[2] && bytes < ((uint64_t)1 << 32) && c->force_general != 1;
}
bool ideal_path_ok(const iono_ctx *c) { return fast_path_ok(c) && c->ideal && c->force_general == 0; }
// the v2 kernels keep the Ns quadrature weights in LDS
bool ideal_path_ok(const iono_ctx *c, int Ns) { return ideal_path_ok(c) && Ns <= 4096; }
// v2 kernels: grid = what is resident at once (blocks per CU from the occupancy query, cached per
// kernel), but no more waves than rays
template <typename K>
int resident_blocks(iono_ctx *c, K kernel, size_t lds) {
    // the occupancy query costs a few microseconds: remember the answer per (kernel, LDS size)
    static thread_local std::vector<std::pair<std::pair<const void *, size_t>, int>> cache;
    const std::pair<const void *, size_t> key((const void *)kernel, lds);
    int per_cu = 0;
    for (auto &e : cache)
        if (e.first == key) per_cu = e.second;
    if (per_cu == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess || per_cu < 1) per_cu = 4;
        cache.push_back({key, per_cu});
    }
    if (per_cu > 8) per_cu = 8;
    if (c->blocks_per_cu_override > 0) per_cu = c->blocks_per_cu_override;
    return per_cu * c->num_cus;
}
int chunk_grid_blocks(int resident, int64_t R) {
    int64_t b = (R + 3) / 4;                 // at least one ray per wave
    if (b > resident) b = resident;
    if (b >= 8) b = b / 8 * 8;               // multiple of 8: XCD-major wave numbering
    return (int)(b < 1 ? 1 : b);
}

int need_grid(iono_ctx *c) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    if (!c->d_M) return fail(c, IONO_ERR_ARG, "no grid set (call iono_grid_set first)");
    // every allocation / launch below belongs to the ctx's GPU (one process per GPU is the model, but a
    // caller whose current device differs must not end up allocating on the wrong card)
    HIP_TRY(c, hipSetDevice(c->device));
    return IONO_OK;
}

int ray_grid_blocks(const iono_ctx *c, int64_t R) {
    // one wave per ray, 4 waves per block; cap at 8 blocks per CU and grid-stride the rest
    int64_t b = (R + 3) / 4;
    const int64_t cap = (int64_t)c->num_cus * 8;
    if (b > cap) b = cap;
    if (b >= 8) b = (b + 7) / 8 * 8;      // multiple of 8 so the XCD-aware walk applies
    return (int)(b < 1 ? 1 : b);
}
int ew_blocks(const iono_ctx *c, int64_t n) {
    int64_t b = (n + 255) / 256;
    const int64_t cap = (int64_t)c->num_cus * 8;
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

// unit-spacing quadrature weights (h = 1): straight rays sample s uniformly, so the integral is
// h * sum(unitw * y).  Same formulas as quad_weight() above, evaluated once on the host.
void host_unit_weights(int N, int rule, std::vector<double> &w) {
    w.assign(N, 0.0);
    auto basic = [&](int a, int b, double f) {
        for (int k = a; k + 2 <= b; k += 2) {
            w[k] += f / 3.0;
            w[k + 1] += f * 4.0 / 3.0;
            w[k + 2] += f / 3.0;
        }
    };
    if (rule == IONO_QUAD_TRAPEZOID || N == 2) {
        for (int k = 0; k + 1 < N; ++k) {
            w[k] += 0.5;
            w[k + 1] += 0.5;
        }
    } else if (N & 1) {
        basic(0, N - 1, 1.0);
    } else if (rule == IONO_QUAD_SIMPSON_AVG) {
        basic(0, N - 2, 0.5);
        w[N - 1] += 0.25;
        w[N - 2] += 0.25;
        basic(1, N - 1, 0.5);
        w[0] += 0.25;
        w[1] += 0.25;
    } else {
        basic(0, N - 2, 1.0);
        w[N - 1] += 5.0 / 12.0;
        w[N - 2] += 4.0 / 6.0;
        w[N - 3] -= 1.0 / 12.0;
    }
}

int ensure_unitw(iono_ctx *c, int Ns, int rule) {
    if (c->d_unitw && c->unitw_n == Ns && c->unitw_rule == rule) return IONO_OK;
    std::vector<double> w;
    host_unit_weights(Ns, rule, w);
    if (c->d_unitw) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipFree(c->d_unitw));
        c->d_unitw = nullptr;
    }
    HIP_TRY(c, hipMalloc((void **)&c->d_unitw, sizeof(double) * Ns));
    HIP_TRY(c, hipMemcpy(c->d_unitw, w.data(), sizeof(double) * Ns, hipMemcpyHostToDevice));
    c->unitw_n = Ns;
    c->unitw_rule = rule;
    return IONO_OK;
}

int check_common(iono_ctx *c, int64_t R, int Ns, int kind, int rule) {
    int rc = need_grid(c);
    if (rc) return rc;
    if (R < 0 || Ns < 2) return fail(c, IONO_ERR_SHAPE, "need R >= 0 and Ns >= 2");
    if (kind != IONO_INTERP_TRILINEAR && kind != IONO_INTERP_TRICUBIC) return fail(c, IONO_ERR_ARG, "bad interp_kind");
    if (rule < 0 || rule > 2) return fail(c, IONO_ERR_ARG, "bad quad_rule");
    if (kind == IONO_INTERP_TRICUBIC && (c->nx < 6 || c->ny < 6 || c->nz < 6))
        return fail(c, IONO_ERR_SHAPE, "tricubic needs at least 6 nodes per axis");
    return IONO_OK;
}

int read_flag(iono_ctx *c, int which, int *out) {
    int v[2] = {0, 0};
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(v, c->d_flags, sizeof(v), hipMemcpyDeviceToHost));
    *out = v[which];
    if (v[which]) {
        v[which] = 0;
        HIP_TRY(c, hipMemcpy(c->d_flags, v, sizeof(v), hipMemcpyHostToDevice));
    }
    return IONO_OK;
}

int finish_host_call(iono_ctx *c, const char *what) {
    int oob = 0;
    int rc = read_flag(c, 0, &oob);
    if (rc) return rc;
    if (oob) return fail(c, IONO_ERR_OOB, std::string(what) + ": One of the requested xi is out of bounds");
    return IONO_OK;
}

template <typename F> int dispatch_storage(iono_ctx *c, F f) {
    return c->storage == IONO_F64 ? f((double *)nullptr) : f((float *)nullptr);
}

}  // namespace

extern "C" {

int iono_version(void) { return IONO_VERSION; }

const char *iono_last_error(iono_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int iono_ctx_create(int device_id, iono_ctx **out) {
    if (!out) return fail(nullptr, IONO_ERR_ARG, "null out pointer");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, IONO_ERR_HIP, "no HIP device visible: libionotomo_hip has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, IONO_ERR_ARG, "device_id out of range");
    iono_ctx *c = new iono_ctx();
    c->device = device_id;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_flags, 2 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(c->d_flags, 0, 2 * sizeof(int));
    if (e != hipSuccess) {
        std::string m = std::string("iono_ctx_create: ") + hipGetErrorString(e);
        delete c;
        return fail(nullptr, IONO_ERR_HIP, m);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
        c->num_cus = prop.multiProcessorCount;
    c->stream = c->own_stream;
    if (const char *e = getenv("IONOTOMO_FORCE_GENERAL")) c->force_general = atoi(e);
    if (const char *e = getenv("IONOTOMO_VARIANT")) c->variant = atoi(e);
    if (const char *e = getenv("IONOTOMO_BLOCKS_PER_CU")) c->blocks_per_cu_override = atoi(e);
    if (const char *e = getenv("IONOTOMO_WALK")) c->walk_mode = atoi(e);
    *out = c;
    return IONO_OK;
}

int iono_ctx_destroy(iono_ctx *c) {
    if (!c) return IONO_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->d_axes) (void)hipFree(c->d_axes);
    if (c->d_M) (void)hipFree(c->d_M);
    if (c->d_flags) (void)hipFree(c->d_flags);
    if (c->d_unitw) (void)hipFree(c->d_unitw);
    if (c->d_nM) (void)hipFree(c->d_nM);
    if (c->d_kern) (void)hipFree(c->d_kern);
    if (c->d_work) (void)hipFree(c->d_work);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return IONO_OK;
}

int iono_ctx_set_stream(iono_ctx *c, void *s) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = (hipStream_t)s;
    return IONO_OK;
}

int iono_ctx_use_own_stream(iono_ctx *c) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = c->own_stream;
    return IONO_OK;
}

int iono_ctx_synchronize(iono_ctx *c) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return IONO_OK;
}

// ---- grid ------------------------------------------------------------------------------------
int iono_grid_set(iono_ctx *c, const double *xv, int nx, const double *yv, int ny, const double *zv, int nz,
                  const double *M, int storage) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    if (!xv || !yv || !zv) return fail(c, IONO_ERR_ARG, "null axis");
    if (nx < 2 || ny < 2 || nz < 2) return fail(c, IONO_ERR_SHAPE, "every axis needs at least 2 nodes");
    if (storage != IONO_F64 && storage != IONO_F32) return fail(c, IONO_ERR_ARG, "bad storage type");
    if ((size_t)(nx + ny + nz) * sizeof(double) > 96 * 1024) return fail(c, IONO_ERR_SHAPE, "axes do not fit the LDS budget");
    const double *ax[3] = {xv, yv, zv};
    const int n[3] = {nx, ny, nz};
    bool ideal = true;
    for (int a = 0; a < 3; ++a) {
        bool uni = true;
        const double h = (ax[a][n[a] - 1] - ax[a][0]) / (n[a] - 1);
        for (int i = 0; i + 1 < n[a]; ++i) {
            const double d = ax[a][i + 1] - ax[a][i];
            if (!(d > 0) || !std::isfinite(d)) return fail(c, IONO_ERR_ARG, "axes must be strictly increasing and finite");
            if (std::fabs(d - h) > 1e-6 * h) uni = false;
        }
        c->uniform[a] = uni ? 1 : 0;
        c->inv_h[a] = 1.0 / h;
        c->g0[a] = ax[a][0];
        c->glast[a] = ax[a][n[a] - 1];
        for (int i = 0; i < n[a]; ++i)
            if (std::fabs(ax[a][i] - (ax[a][0] + i * h)) > 2.5e-13 * h) ideal = false;
    }
    c->ideal = ideal ? 1 : 0;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_axes) HIP_TRY(c, hipFree(c->d_axes));
    if (c->d_M) HIP_TRY(c, hipFree(c->d_M));
    if (c->d_nM) HIP_TRY(c, hipFree(c->d_nM));
    c->d_axes = nullptr;
    c->d_M = nullptr;
    c->d_nM = nullptr;
    c->nM_freq = -1.0;
    c->nx = nx;
    c->ny = ny;
    c->nz = nz;
    c->storage = storage;
    std::vector<double> cat;
    cat.insert(cat.end(), xv, xv + nx);
    cat.insert(cat.end(), yv, yv + ny);
    cat.insert(cat.end(), zv, zv + nz);
    HIP_TRY(c, hipMalloc((void **)&c->d_axes, cat.size() * sizeof(double)));
    HIP_TRY(c, hipMemcpy(c->d_axes, cat.data(), cat.size() * sizeof(double), hipMemcpyHostToDevice));
    const size_t esz = storage == IONO_F64 ? 8 : 4;
    // + one plane + one row + 2 zero elements: see trilinear_u (unclamped far-corner reads, weight 0)
    const size_t padded = (size_t)ncells(c) + (size_t)ny * nz + nz + 2;
    HIP_TRY(c, hipMalloc(&c->d_M, padded * esz));
    HIP_TRY(c, hipMemset(c->d_M, 0, padded * esz));
    if (M) return iono_grid_set_values(c, M);
    return IONO_OK;
}

static int set_values_dev_impl(iono_ctx *c, const double *src_dev, int do_exp, double scale) {
    const int64_t n = ncells(c);
    c->nM_freq = -1.0;
    int rc = dispatch_storage(c, [&](auto *tag) {
        using GT = std::remove_pointer_t<decltype(tag)>;
        hipLaunchKernelGGL((k_set_values<GT>), dim3(ew_blocks(c, n)), dim3(256), 0, c->stream, src_dev, (GT *)c->d_M, n,
                           do_exp, scale, c->d_flags + 1);
        return IONO_OK;
    });
    HIP_TRY(c, hipGetLastError());
    return rc;
}

static int set_values_host_impl(iono_ctx *c, const double *M, int do_exp, double scale) {
    int rc = need_grid(c);
    if (rc) return rc;
    if (!M) return fail(c, IONO_ERR_ARG, "null values");
    DevBuf tmp(c);
    HIP_TRY(c, tmp.alloc((size_t)ncells(c) * 8));
    HIP_TRY(c, hipMemcpyAsync(tmp.p, M, (size_t)ncells(c) * 8, hipMemcpyHostToDevice, c->stream));
    rc = set_values_dev_impl(c, tmp.as<double>(), do_exp, scale);
    if (rc) return rc;
    int bad = 0;
    rc = read_flag(c, 1, &bad);
    if (rc) return rc;
    if (bad) return fail(c, IONO_ERR_NONFINITE, "grid values contain NaN or Inf");
    return IONO_OK;
}

int iono_grid_set_values(iono_ctx *c, const double *M) { return set_values_host_impl(c, M, 0, 1.0); }
int iono_grid_set_exp(iono_ctx *c, const double *m, double scale) { return set_values_host_impl(c, m, 1, scale); }
int iono_grid_set_values_dev(iono_ctx *c, const double *M_dev) {
    int rc = need_grid(c);
    return rc ? rc : set_values_dev_impl(c, M_dev, 0, 1.0);
}
int iono_grid_set_exp_dev(iono_ctx *c, const double *m_dev, double scale) {
    int rc = need_grid(c);
    return rc ? rc : set_values_dev_impl(c, m_dev, 1, scale);
}
void *iono_grid_values_ptr(iono_ctx *c) { return c ? c->d_M : nullptr; }

int iono_grid_get_values(iono_ctx *c, double *out) {
    int rc = need_grid(c);
    if (rc) return rc;
    const int64_t n = ncells(c);
    DevBuf tmp(c);
    HIP_TRY(c, tmp.alloc((size_t)n * 8));
    dispatch_storage(c, [&](auto *tag) {
        using GT = std::remove_pointer_t<decltype(tag)>;
        hipLaunchKernelGGL((k_get_values<GT>), dim3(ew_blocks(c, n)), dim3(256), 0, c->stream, (const GT *)c->d_M,
                           tmp.as<double>(), n);
        return IONO_OK;
    });
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, tmp.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return IONO_OK;
}

// ---- interp at points --------------------------------------------------------------------------
int iono_interp(iono_ctx *c, const double *x, const double *y, const double *z, int64_t n, int kind, int extrapolate,
                double *out) {
    int rc = check_common(c, n, 2, kind, 0);
    if (rc) return rc;
    if (n == 0) return IONO_OK;
    DevBuf b(c);
    HIP_TRY(c, b.alloc((size_t)n * 8 * 4));
    double *dx = b.as<double>(), *dy = dx + n, *dz = dy + n, *dout = dz + n;
    HIP_TRY(c, hipMemcpyAsync(dx, x, n * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dy, y, n * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dz, z, n * 8, hipMemcpyHostToDevice, c->stream));
    const GridView g = view(c);
    const dim3 grid(ew_blocks(c, n)), block(256);
    const size_t lds = lds_bytes(c);
    dispatch_storage(c, [&](auto *tag) {
        using GT = std::remove_pointer_t<decltype(tag)>;
#define LAUNCH_INTERP(K, E) \
    hipLaunchKernelGGL((k_interp_points<GT, K, E>), grid, block, lds, c->stream, g, dx, dy, dz, n, dout, c->d_flags)
        if (kind == IONO_INTERP_TRILINEAR) {
            if (extrapolate) LAUNCH_INTERP(IONO_INTERP_TRILINEAR, true); else LAUNCH_INTERP(IONO_INTERP_TRILINEAR, false);
        } else {
            if (extrapolate) LAUNCH_INTERP(IONO_INTERP_TRICUBIC, true); else LAUNCH_INTERP(IONO_INTERP_TRICUBIC, false);
        }
#undef LAUNCH_INTERP
        return IONO_OK;
    });
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, dout, n * 8, hipMemcpyDeviceToHost, c->stream));
    return finish_host_call(c, "iono_interp");
}

// ---- forward (device pointers) ---------------------------------------------------------------
int iono_forward_tec_straight_dev(iono_ctx *c, const double *o, const double *d, const int *order, int64_t R, double tmax,
                                  int Ns, int kind, int rule, double *tec) {
    int rc = check_common(c, R, Ns, kind, rule);
    if (rc) return rc;
    if (R == 0) return IONO_OK;
    rc = ensure_unitw(c, Ns, rule);
    if (rc) return rc;
    const GridView g = view(c);
    const dim3 grid(ray_grid_blocks(c, R)), block(256);
    const size_t lds = lds_bytes(c);
    dispatch_storage(c, [&](auto *tag) {
        using GT = std::remove_pointer_t<decltype(tag)>;
        if (kind == IONO_INTERP_TRILINEAR && ideal_path_ok(c, Ns)) {
            const size_t wl = sizeof(double) * Ns;
            const int nb = chunk_grid_blocks(resident_blocks(c, k_forward_straight_u<GT>, wl), R);
            hipLaunchKernelGGL((k_forward_straight_u<GT>), dim3(nb), block, wl, c->stream, g, o, d, order, R, tmax, Ns,
                               c->walk_mode, c->d_unitw, tec, c->d_flags);
        } else if (kind == IONO_INTERP_TRILINEAR && fast_path_ok(c))      // (`order` is a speed hint: ignored here)
            hipLaunchKernelGGL((k_forward_straight_fast<GT>), grid, block, 2 * lds, c->stream, g, o, d, R, tmax, Ns,
                               c->d_unitw, tec, c->d_flags);
        else if (kind == IONO_INTERP_TRILINEAR)
            hipLaunchKernelGGL((k_forward_straight<GT, IONO_INTERP_TRILINEAR>), grid, block, lds, c->stream, g, o, d, R, tmax,
                               Ns, c->d_unitw, tec, c->d_flags);
        else
            hipLaunchKernelGGL((k_forward_straight<GT, IONO_INTERP_TRICUBIC>), grid, block, lds, c->stream, g, o, d, R, tmax,
                               Ns, c->d_unitw, tec, c->d_flags);
        return IONO_OK;
    });
    HIP_TRY(c, hipGetLastError());
    return rc;
}

int iono_forward_tec_rays_dev(iono_ctx *c, const double *rays, int64_t R, int Ns, int kind, int rule, double *tec) {
    int rc = check_common(c, R, Ns, kind, rule);
    if (rc) return rc;
    if (R == 0) return IONO_OK;
    const GridView g = view(c);
    const dim3 grid(ray_grid_blocks(c, R)), block(256);
    const size_t lds = lds_bytes(c);
    dispatch_storage(c, [&](auto *tag) {
        using GT = std::remove_pointer_t<decltype(tag)>;
        if (kind == IONO_INTERP_TRILINEAR)
            hipLaunchKernelGGL((k_forward_rays<GT, IONO_INTERP_TRILINEAR>), grid, block, lds, c->stream, g, rays, R, Ns, rule,
                               tec, c->d_flags);
        else
            hipLaunchKernelGGL((k_forward_rays<GT, IONO_INTERP_TRICUBIC>), grid, block, lds, c->stream, g, rays, R, Ns, rule,
                               tec, c->d_flags);
        return IONO_OK;
    });
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

int iono_subtract_reference_dev(iono_ctx *c, double *tec, int Na, int64_t NtNd, int i0) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    if (i0 < 0 || i0 >= Na) return fail(c, IONO_ERR_ARG, "reference antenna index out of range");
    hipLaunchKernelGGL(k_subtract_reference, dim3(ew_blocks(c, (int64_t)Na * NtNd)), dim3(256), 0, c->stream, tec, Na, NtNd, i0);
    hipLaunchKernelGGL(k_zero, dim3(ew_blocks(c, NtNd)), dim3(256), 0, c->stream, tec + (int64_t)i0 * NtNd, NtNd);
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

// ---- adjoint (device pointers) ----------------------------------------------------------------
static int adjoint_straight_launch(iono_ctx *c, int mode, const double *o, const double *d, const int *order, const double *w,
                                   const double *tec, const double *dobs, const double *cdct, int Na, int64_t NtNd, int i0,
                                   int64_t R, double tmax, int Ns, int rule, void *grad, int accum) {
    int rc = check_common(c, R, Ns, IONO_INTERP_TRILINEAR, rule);
    if (rc) return rc;
    if (accum != IONO_F64 && accum != IONO_F32) return fail(c, IONO_ERR_ARG, "bad accum_dtype");
    if (R == 0) return IONO_OK;
    rc = ensure_unitw(c, Ns, rule);
    if (rc) return rc;
    const GridView g = view(c);
    const dim3 block(256);
    if (ideal_path_ok(c, Ns) && c->variant != 2) {
        const size_t esz = accum == IONO_F64 ? 8 : 4;
        constexpr int NWv = 4;    // waves per workgroup (8 waves sharing one tile, bundles of 128: measured 8 % slower)
        const size_t tl = sizeof(double) * (((size_t)Ns + 1) & ~(size_t)1) + 16 * NWv * sizeof(double) +
                          esz * T_WIN * T_WIN * T_TKP + 2 * T_TK * sizeof(int) + 16;
#define LAUNCH_ADJT(AT, MODE, NW)                                                                                          \
    do {                                                                                                                   \
        int per_cu = 0;                                                                                                    \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_adjoint_straight_tile<AT, MODE, NW>, 64 * NW, tl) !=   \
                hipSuccess || per_cu < 1)                                                                                  \
            per_cu = 1;                                                                                                    \
        int nb = per_cu * c->num_cus;                                                                                      \
        const int64_t nbund = (R + 16 * NW - 1) / (16 * NW);                                                               \
        if (nb > nbund) nb = (int)nbund;                                                                                   \
        if (nb >= 8) nb = nb / 8 * 8;                                                                                      \
        hipLaunchKernelGGL((k_adjoint_straight_tile<AT, MODE, NW>), dim3(nb), dim3(64 * NW), tl, c->stream, g, o, d,      \
                           order, w, tec, dobs, cdct, Na, NtNd, i0, R, tmax, Ns, c->walk_mode, c->d_unitw, (AT *)grad,     \
                           c->d_flags);                                                                                    \
    } while (0)
#define LAUNCH_ADJT_NW(AT, MODE) LAUNCH_ADJT(AT, MODE, NWv)
        if (accum == IONO_F64) {
            if (mode == 0) LAUNCH_ADJT_NW(double, 0); else LAUNCH_ADJT_NW(double, 1);
        } else {
            if (mode == 0) LAUNCH_ADJT_NW(float, 0); else LAUNCH_ADJT_NW(float, 1);
        }
#undef LAUNCH_ADJT_NW
#undef LAUNCH_ADJT
        HIP_TRY(c, hipGetLastError());
        return IONO_OK;
    }
    const dim3 grid(ray_grid_blocks(c, R));
    const size_t lds = lds_bytes(c);
#define LAUNCH_ADJ(AT, MODE)                                                                                          \
    hipLaunchKernelGGL((k_adjoint_straight<AT, MODE>), grid, block, lds, c->stream, g, o, d, w, tec, dobs, cdct, Na, \
                       NtNd, i0, R, tmax, Ns, c->d_unitw, (AT *)grad, c->d_flags)
    if (accum == IONO_F64) {
        if (mode == 0) LAUNCH_ADJ(double, 0); else LAUNCH_ADJ(double, 1);
    } else {
        if (mode == 0) LAUNCH_ADJ(float, 0); else LAUNCH_ADJ(float, 1);
    }
#undef LAUNCH_ADJ
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

int iono_adjoint_straight_dev(iono_ctx *c, const double *o, const double *d, const int *order, const double *w, int64_t R,
                              double tmax, int Ns, int rule, void *grad, int accum) {
    return adjoint_straight_launch(c, 0, o, d, order, w, nullptr, nullptr, nullptr, 1, R, 0, R, tmax, Ns, rule, grad, accum);
}

int iono_adjoint_residual_straight_dev(iono_ctx *c, const double *o, const double *d, const int *order, const double *tec,
                                       const double *dobs, const double *cdct, int Na, int64_t NtNd, int i0, double tmax, int Ns,
                                       int rule, void *grad, int accum) {
    if (Na < 1 || NtNd < 0 || i0 < 0 || i0 >= Na) return fail(c, IONO_ERR_ARG, "bad [Na][NtNd]/i0");
    return adjoint_straight_launch(c, 1, o, d, order, nullptr, tec, dobs, cdct, Na, NtNd, i0, (int64_t)Na * NtNd, tmax, Ns,
                                   rule, grad, accum);
}

int iono_adjoint_rays_dev(iono_ctx *c, const double *rays, const double *w, int64_t R, int Ns, int rule, void *grad,
                          int accum) {
    int rc = check_common(c, R, Ns, IONO_INTERP_TRILINEAR, rule);
    if (rc) return rc;
    if (accum != IONO_F64 && accum != IONO_F32) return fail(c, IONO_ERR_ARG, "bad accum_dtype");
    if (R == 0) return IONO_OK;
    const GridView g = view(c);
    const dim3 grid(ray_grid_blocks(c, R)), block(256);
    const size_t lds = lds_bytes(c);
    if (accum == IONO_F64)
        hipLaunchKernelGGL((k_adjoint_rays<double>), grid, block, lds, c->stream, g, rays, w, R, Ns, rule, (double *)grad,
                           c->d_flags);
    else
        hipLaunchKernelGGL((k_adjoint_rays<float>), grid, block, lds, c->stream, g, rays, w, R, Ns, rule, (float *)grad,
                           c->d_flags);
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

int iono_check_oob(iono_ctx *c, int *oob) {
    if (!c || !oob) return fail(c, IONO_ERR_ARG, "null argument");
    return read_flag(c, 0, oob);
}

// ---- host-pointer wrappers --------------------------------------------------------------------
int iono_forward_tec_straight(iono_ctx *c, const double *o, const double *d, int64_t R, double tmax, int Ns, int kind, int rule,
                              double *tec) {
    int rc = check_common(c, R, Ns, kind, rule);
    if (rc) return rc;
    if (R == 0) return IONO_OK;
    DevBuf b(c);
    HIP_TRY(c, b.alloc((size_t)R * 8 * 7));
    double *dO = b.as<double>(), *dD = dO + 3 * R, *dT = dD + 3 * R;
    HIP_TRY(c, hipMemcpyAsync(dO, o, R * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dD, d, R * 24, hipMemcpyHostToDevice, c->stream));
    rc = iono_forward_tec_straight_dev(c, dO, dD, nullptr, R, tmax, Ns, kind, rule, dT);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(tec, dT, R * 8, hipMemcpyDeviceToHost, c->stream));
    return finish_host_call(c, "iono_forward_tec_straight");
}

int iono_forward_tec_rays(iono_ctx *c, const double *rays, int64_t R, int Ns, int kind, int rule, double *tec) {
    int rc = check_common(c, R, Ns, kind, rule);
    if (rc) return rc;
    if (R == 0) return IONO_OK;
    DevBuf b(c);
    HIP_TRY(c, b.alloc((size_t)R * 8 * (4 * (size_t)Ns + 1)));
    double *dR = b.as<double>(), *dT = dR + (size_t)R * 4 * Ns;
    HIP_TRY(c, hipMemcpyAsync(dR, rays, (size_t)R * 4 * Ns * 8, hipMemcpyHostToDevice, c->stream));
    rc = iono_forward_tec_rays_dev(c, dR, R, Ns, kind, rule, dT);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(tec, dT, R * 8, hipMemcpyDeviceToHost, c->stream));
    return finish_host_call(c, "iono_forward_tec_rays");
}

int iono_subtract_reference(iono_ctx *c, double *tec, int Na, int64_t NtNd, int i0) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    const int64_t n = (int64_t)Na * NtNd;
    DevBuf b(c);
    HIP_TRY(c, b.alloc((size_t)n * 8));
    HIP_TRY(c, hipMemcpyAsync(b.p, tec, n * 8, hipMemcpyHostToDevice, c->stream));
    int rc = iono_subtract_reference_dev(c, b.as<double>(), Na, NtNd, i0);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(tec, b.p, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return IONO_OK;
}

int iono_forward_phase_rays(iono_ctx *c, const double *rays, int Na, int Nt, int Nd, int Ns, const double *freqs, int Nf,
                            const double *clock, const double *cst, int i0, int rule, double *gout) {
    const int64_t R = (int64_t)Na * Nt * Nd;
    int rc = check_common(c, R, Ns, IONO_INTERP_TRILINEAR, rule);
    if (rc) return rc;
    if (Nf < 1 || i0 < 0 || i0 >= Na) return fail(c, IONO_ERR_ARG, "bad Nf / i0");
    if (R == 0) return IONO_OK;
    constexpr int MAXF = 8;
    DevBuf b(c);
    const size_t nr = (size_t)R * 4 * Ns, nphi = (size_t)R * Nf;
    HIP_TRY(c, b.alloc(8 * (nr + 2 * nphi + 2 * (size_t)Nf + (size_t)Na * Nt + Na)));
    double *dR = b.as<double>(), *dPhi = dR + nr, *dG = dPhi + nphi, *dF = dG + nphi, *dInv = dF + Nf, *dClock = dInv + Nf,
           *dConst = dClock + (size_t)Na * Nt;
    std::vector<double> inv(Nf);
    for (int l = 0; l < Nf; ++l) inv[l] = 1.0 / (1.2404e-2 * freqs[l] * freqs[l]);     // iterative_newton.py:112
    HIP_TRY(c, hipMemcpyAsync(dR, rays, nr * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dF, freqs, Nf * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dInv, inv.data(), Nf * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dClock, clock, (size_t)Na * Nt * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dConst, cst, (size_t)Na * 8, hipMemcpyHostToDevice, c->stream));
    const GridView g = view(c);
    const dim3 grid(ray_grid_blocks(c, R)), block(256);
    for (int f0 = 0; f0 < Nf; f0 += MAXF) {
        const int nf = std::min(MAXF, Nf - f0);
        dispatch_storage(c, [&](auto *tag) {
            using GT = std::remove_pointer_t<decltype(tag)>;
            hipLaunchKernelGGL((k_forward_phase_rays<GT, MAXF>), grid, block, lds_bytes(c), c->stream, g, dR, R, Ns, rule,
                               dInv + f0, nf, Nf, dPhi + f0, c->d_flags);
            return IONO_OK;
        });
    }
    hipLaunchKernelGGL(k_phase_finish, dim3(ew_blocks(c, (int64_t)nphi)), dim3(256), 0, c->stream, dPhi, dF, dClock, dConst, Na,
                       Nt, Nd, Nf, i0, dG);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(gout, dG, nphi * 8, hipMemcpyDeviceToHost, c->stream));
    return finish_host_call(c, "iono_forward_phase_rays");
}

static int adjoint_host_finish(iono_ctx *c, double *dG, int scale_by_grid, double *grad_out, const char *what) {
    const int64_t n = ncells(c);
    if (scale_by_grid)
        dispatch_storage(c, [&](auto *tag) {
            using GT = std::remove_pointer_t<decltype(tag)>;
            hipLaunchKernelGGL((k_scale_by_grid<double, GT>), dim3(ew_blocks(c, n)), dim3(256), 0, c->stream, dG,
                               (const GT *)c->d_M, n);
            return IONO_OK;
        });
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(grad_out, dG, n * 8, hipMemcpyDeviceToHost, c->stream));
    return finish_host_call(c, what);
}

int iono_adjoint_straight(iono_ctx *c, const double *o, const double *d, const double *w, int64_t R, double tmax, int Ns,
                          int rule, int scale_by_grid, double *grad_out) {
    int rc = check_common(c, R, Ns, IONO_INTERP_TRILINEAR, rule);
    if (rc) return rc;
    const int64_t n = ncells(c);
    DevBuf b(c);
    HIP_TRY(c, b.alloc(8 * ((size_t)R * 7 + (size_t)n)));
    double *dO = b.as<double>(), *dD = dO + 3 * R, *dW = dD + 3 * R, *dG = dW + R;
    HIP_TRY(c, hipMemcpyAsync(dO, o, R * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dD, d, R * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dW, w, R * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(dG, 0, n * 8, c->stream));
    rc = iono_adjoint_straight_dev(c, dO, dD, nullptr, dW, R, tmax, Ns, rule, dG, IONO_F64);
    if (rc) return rc;
    return adjoint_host_finish(c, dG, scale_by_grid, grad_out, "iono_adjoint_straight");
}

int iono_adjoint_rays(iono_ctx *c, const double *rays, const double *w, int64_t R, int Ns, int rule, int scale_by_grid,
                      double *grad_out) {
    int rc = check_common(c, R, Ns, IONO_INTERP_TRILINEAR, rule);
    if (rc) return rc;
    const int64_t n = ncells(c);
    DevBuf b(c);
    const size_t nr = (size_t)R * 4 * Ns;
    HIP_TRY(c, b.alloc(8 * (nr + (size_t)R + (size_t)n)));
    double *dR = b.as<double>(), *dW = dR + nr, *dG = dW + R;
    HIP_TRY(c, hipMemcpyAsync(dR, rays, nr * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dW, w, R * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(dG, 0, n * 8, c->stream));
    rc = iono_adjoint_rays_dev(c, dR, dW, R, Ns, rule, dG, IONO_F64);
    if (rc) return rc;
    return adjoint_host_finish(c, dG, scale_by_grid, grad_out, "iono_adjoint_rays");
}

// ---- C_m smoothing ------------------------------------------------------------------------------------
int iono_smooth_separable_dev(iono_ctx *c, const double *in_dev, double *out_dev, double *work_dev, const double *kx,
                              const double *ky, const double *kz, int h) {
    int rc = need_grid(c);
    if (rc) return rc;
    if (h < 0 || h > 512 || !kx || !ky || !kz) return fail(c, IONO_ERR_ARG, "bad smoothing kernel");
    if (in_dev == out_dev || in_dev == work_dev || out_dev == work_dev) return fail(c, IONO_ERR_ARG, "in/out/work must differ");
    const int m = 2 * h + 1;
    if (c->kern_cap < 3 * m) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->d_kern) HIP_TRY(c, hipFree(c->d_kern));
        c->d_kern = nullptr;
        HIP_TRY(c, hipMalloc((void **)&c->d_kern, sizeof(double) * 3 * m));
        c->kern_cap = 3 * m;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_kern, kx, sizeof(double) * m, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_kern + m, ky, sizeof(double) * m, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_kern + 2 * m, kz, sizeof(double) * m, hipMemcpyHostToDevice, c->stream));
    const dim3 block(256);
    const size_t wpad = sizeof(double) * ((2 * h + 2) & ~1);
    const size_t lds_xy = wpad + sizeof(double) * 64 * (CONV_T + 2 * h), lds_z = wpad + sizeof(double) * 4 * (64 + 2 * h);
    if (lds_xy > 150 * 1024) return fail(c, IONO_ERR_ARG, "smoothing stencil too wide for the LDS tile");
    const int zt = (c->nz + 63) / 64;
    auto tiles = [&](int na, int no) { return (int64_t)zt * ((na + CONV_T - 1) / CONV_T) * no; };
    auto nblk = [&](int64_t work) { return dim3((unsigned)std::min<int64_t>(work, (int64_t)c->num_cus * 16)); };
    hipLaunchKernelGGL((k_conv_xy<0>), nblk(tiles(c->nx, c->ny)), block, lds_xy, c->stream, in_dev, out_dev, c->nx, c->ny, c->nz,
                       c->d_kern, h);
    hipLaunchKernelGGL((k_conv_xy<1>), nblk(tiles(c->ny, c->nx)), block, lds_xy, c->stream, out_dev, work_dev, c->nx, c->ny, c->nz,
                       c->d_kern + m, h);
    hipLaunchKernelGGL(k_conv_z, nblk(((int64_t)c->nx * c->ny * zt + 3) / 4), block, lds_z, c->stream, work_dev, out_dev, c->nx,
                       c->ny, c->nz, c->d_kern + 2 * m, h);
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

int iono_smooth_separable(iono_ctx *c, const double *in, double *out, const double *kx, const double *ky, const double *kz,
                          int h) {
    int rc = need_grid(c);
    if (rc) return rc;
    const int64_t n = ncells(c);
    DevBuf b(c);
    HIP_TRY(c, b.alloc((size_t)n * 8 * 3));
    double *dI = b.as<double>(), *dO = dI + n, *dW = dO + n;
    HIP_TRY(c, hipMemcpyAsync(dI, in, n * 8, hipMemcpyHostToDevice, c->stream));
    rc = iono_smooth_separable_dev(c, dI, dO, dW, kx, ky, kz, h);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, dO, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return IONO_OK;
}

// ---- ray geometry ------------------------------------------------------------------------------
int iono_trace_straight(iono_ctx *c, const double *o, const double *d, int64_t R, double tmax, int Ns, double *rays_out) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    if (R < 0 || Ns < 2) return fail(c, IONO_ERR_SHAPE, "need R >= 0 and Ns >= 2");
    if (R == 0) return IONO_OK;
    DevBuf b(c);
    const size_t nr = (size_t)R * 4 * Ns;
    HIP_TRY(c, b.alloc(8 * (nr + 6 * (size_t)R)));
    double *dR = b.as<double>(), *dO = dR + nr, *dD = dO + 3 * R;
    HIP_TRY(c, hipMemcpyAsync(dO, o, R * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dD, d, R * 24, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_trace_straight, dim3(ew_blocks(c, R * Ns)), dim3(256), 0, c->stream, dO, dD, R, tmax, Ns, dR);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(rays_out, dR, nr * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return IONO_OK;
}

int iono_trace_fermat_dev(iono_ctx *c, const double *dO, const double *dD, int64_t R, double tmax, int Ns, double frequency,
                          int bend, int kind, int substeps, double *dR) {
    int rc = check_common(c, R, Ns, kind, 0);
    if (rc) return rc;
    if (substeps < 1 || !(frequency > 0)) return fail(c, IONO_ERR_ARG, "need substeps >= 1 and frequency > 0");
    if (R == 0) return IONO_OK;
    const int64_t n = ncells(c);
    if (!c->d_nM) HIP_TRY(c, hipMalloc((void **)&c->d_nM, (size_t)n * 8));
    if (c->nM_freq != frequency) {       // n = sqrt(1 - 8.98^2 ne / nu^2) at the nodes, rebuilt when ne or nu changed
        dispatch_storage(c, [&](auto *tag) {
            using GT = std::remove_pointer_t<decltype(tag)>;
            hipLaunchKernelGGL((k_ne_to_n<GT>), dim3(ew_blocks(c, n)), dim3(256), 0, c->stream, (const GT *)c->d_M, c->d_nM, n,
                               frequency);
            return IONO_OK;
        });
        c->nM_freq = frequency;
    }
    const GridView g = view(c);
    const dim3 grid((unsigned)((R + 63) / 64)), block(64);
    double *dN = c->d_nM;
#define LAUNCH_F(K, B) \
    hipLaunchKernelGGL((k_trace_fermat<K, B>), grid, block, 0, c->stream, g, dN, dO, dD, R, tmax, Ns, substeps, dR, c->d_flags)
    if (kind == IONO_INTERP_TRILINEAR) {
        if (bend) LAUNCH_F(IONO_INTERP_TRILINEAR, true); else LAUNCH_F(IONO_INTERP_TRILINEAR, false);
    } else if (c->variant == 3) {           // lanes = rays (kept for A/B)
        if (bend) LAUNCH_F(IONO_INTERP_TRICUBIC, true); else LAUNCH_F(IONO_INTERP_TRICUBIC, false);
    } else {                                // 8 lanes per ray
        const dim3 cgrid((unsigned)((R + 7) / 8));
        if (bend)
            hipLaunchKernelGGL((k_trace_fermat_coop<true>), cgrid, block, 0, c->stream, g, dN, dO, dD, R, tmax, Ns, substeps, dR,
                               c->d_flags);
        else
            hipLaunchKernelGGL((k_trace_fermat_coop<false>), cgrid, block, 0, c->stream, g, dN, dO, dD, R, tmax, Ns, substeps, dR,
                               c->d_flags);
    }
#undef LAUNCH_F
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

int iono_trace_fermat(iono_ctx *c, const double *o, const double *d, int64_t R, double tmax, int Ns, double frequency, int bend,
                      int kind, int substeps, double *rays_out) {
    int rc = check_common(c, R, Ns, kind, 0);
    if (rc) return rc;
    if (R == 0) return IONO_OK;
    DevBuf b(c);
    const size_t nr = (size_t)R * 4 * Ns;
    HIP_TRY(c, b.alloc(8 * (nr + 6 * (size_t)R)));
    double *dR = b.as<double>(), *dO = dR + nr, *dD = dO + 3 * R;
    HIP_TRY(c, hipMemcpyAsync(dO, o, R * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dD, d, R * 24, hipMemcpyHostToDevice, c->stream));
    rc = iono_trace_fermat_dev(c, dO, dD, R, tmax, Ns, frequency, bend, kind, substeps, dR);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(rays_out, dR, nr * 8, hipMemcpyDeviceToHost, c->stream));
    return finish_host_call(c, "iono_trace_fermat");
}

}  // extern "C"
