// device-side building blocks: grid view, cell lookup, trilinear / tricubic interpolation, quadrature weights, wave reductions, ray walks (included by ionotomo_hip.hip)
#ifndef IONO_DEVICE_COMMON_H
#define IONO_DEVICE_COMMON_H

namespace {


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
    double c0[3], clast[3];   // tricubic domain: nodes 2 and n-3 per axis (host copies)
    int ideal;            // every axis is g0 + i h to within 2.5e-13 h (np.linspace) and the general tiers are not forced
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

// Value and gradient on an IDEAL-uniform grid without the axis tables: grid coordinates as the straight-ray kernels form them
// (load_uray), cell = floor, slopes scaled by the reciprocal mean spacing.  Returns false -- nothing computed -- when the point lies
// within 1e-9 of a cell face or outside the grid: there trilinear_grad_at decides the cell by comparing with the axis values themselves
// (g[i] < x <= g[i+1], scipy's rule), which matters because the gradient of a trilinear field jumps across faces; everywhere else the
// two rules name the same cell (an ideal axis deviates from g0 + i h by 2.5e-13 h at most) and the results agree to rounding.
// The tracer's right-hand side spent 300 vector instructions per evaluation on the general form: three cell searches against axis
// tables in memory (a chain of dependent loads before the first corner load can issue) and nine float64 divisions.
__device__ __forceinline__ bool trilinear_grad_ideal(const GridView &g, const double *__restrict__ M, double x, double y, double z, double &f,
                                                     double &fx, double &fy, double &fz) {
    const double ux = (x - g.g0[0]) * g.inv_h[0], uy = (y - g.g0[1]) * g.inv_h[1], uz = (z - g.g0[2]) * g.inv_h[2];
    const double fi = __builtin_floor(ux), fj = __builtin_floor(uy), fk = __builtin_floor(uz);
    const double tx = ux - fi, ty = uy - fj, tz = uz - fk;
    const double eps = 1e-9;
    // (one predicate, no short-circuit branches; `ideal` implies fewer than 2^32 nodes: a 32-bit node index)
    const bool interior = (tx > eps) & (tx < 1.0 - eps) & (ty > eps) & (ty < 1.0 - eps) & (tz > eps) & (tz < 1.0 - eps) & (fi >= 0.0) &
                          (fi <= (double)(g.nx - 2)) & (fj >= 0.0) & (fj <= (double)(g.ny - 2)) & (fk >= 0.0) & (fk <= (double)(g.nz - 2));
    if (!interior) return false;
    const size_t sj = (size_t)g.nz, si = (size_t)g.ny * g.nz;
    const double *p = M + (unsigned)__builtin_fma(fi, (double)si, __builtin_fma(fj, (double)sj, fk));
    const double c000 = p[0], c001 = p[1], c010 = p[sj], c011 = p[sj + 1];
    const double c100 = p[si], c101 = p[si + 1], c110 = p[si + sj], c111 = p[si + sj + 1];
    const double z00 = c001 - c000, z01 = c011 - c010, z10 = c101 - c100, z11 = c111 - c110;        // d/dz along the four columns
    const double v00 = __builtin_fma(tz, z00, c000), v01 = __builtin_fma(tz, z01, c010);
    const double v10 = __builtin_fma(tz, z10, c100), v11 = __builtin_fma(tz, z11, c110);
    const double y0 = v01 - v00, y1 = v11 - v10;                                                       // d/dy in the two x planes
    const double w0 = __builtin_fma(ty, y0, v00), w1 = __builtin_fma(ty, y1, v10);
    const double zz0 = __builtin_fma(ty, z01 - z00, z00), zz1 = __builtin_fma(ty, z11 - z10, z10);
    f = __builtin_fma(tx, w1 - w0, w0);
    fx = (w1 - w0) * g.inv_h[0];
    fy = __builtin_fma(tx, y1 - y0, y0) * g.inv_h[1];
    fz = __builtin_fma(tx, zz1 - zz0, zz0) * g.inv_h[2];
    return true;
}

// value only (continuous across faces: no face rule needed); the caller has checked g0 <= x <= glast on every axis
__device__ __forceinline__ double trilinear_ideal(const GridView &g, const double *__restrict__ M, double x, double y, double z) {
    const double ux = (x - g.g0[0]) * g.inv_h[0], uy = (y - g.g0[1]) * g.inv_h[1], uz = (z - g.g0[2]) * g.inv_h[2];
    const double fi = fmin(__builtin_floor(__builtin_fabs(ux)), (double)(g.nx - 2)), fj = fmin(__builtin_floor(__builtin_fabs(uy)), (double)(g.ny - 2)),
                 fk = fmin(__builtin_floor(__builtin_fabs(uz)), (double)(g.nz - 2));
    const double tx = ux - fi, ty = uy - fj, tz = uz - fk;
    const size_t sj = (size_t)g.nz, si = (size_t)g.ny * g.nz;
    const double *p = M + (unsigned)__builtin_fma(fi, (double)si, __builtin_fma(fj, (double)sj, fk));
    const double v00 = __builtin_fma(tz, p[1] - p[0], p[0]), v01 = __builtin_fma(tz, p[sj + 1] - p[sj], p[sj]);
    const double v10 = __builtin_fma(tz, p[si + 1] - p[si], p[si]), v11 = __builtin_fma(tz, p[si + sj + 1] - p[si + sj], p[si + sj]);
    const double w0 = __builtin_fma(ty, v01 - v00, v00), w1 = __builtin_fma(ty, v11 - v10, v10);
    return __builtin_fma(tx, w1 - w0, w0);
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

__device__ __forceinline__ void cubic_taps_ideal(double t, double (&w)[6]) {        // cubic_axis on a uniform axis: c0 = c1 = 1 / 12
    const double t2 = t * t, t3 = t2 * t;
    const double b0 = 2 * t3 - 3 * t2 + 1, b1 = -2 * t3 + 3 * t2, b2 = (t3 - 2 * t2 + t) * (1.0 / 12.0), b3 = (t3 - t2) * (1.0 / 12.0);
    w[0] = b2, w[1] = -8.0 * b2 + b3, w[2] = b0 - 8.0 * b3, w[3] = b1 + 8.0 * b2, w[4] = -b2 + 8.0 * b3, w[5] = -b3;
}
// The interpolant at GRID coordinates (u = (x - g0) / h per axis) of an ideal-uniform grid straight from the node values: 216 taps.
// What the planned tricubic forward falls back to for rays edited in place when its derivative fields were rebuilt only where the
// PLANNED rays read them (k_forward_bundle_lm): slow, exact, independent of any derived array.
// (inlined with the x loop ROLLED -- its taps rotate through named values, as in tricubic_eval below -- so that it needs ~60 VGPRs and
//  no call: fully unrolled it cost k_forward_bundle_lm a wave per SIMD, out of line it left that kernel with a scratch frame)
__device__ __forceinline__ double tricubic_from_nodes(const double *__restrict__ M, int nx, int ny, int nz, double ux, double uy, double uz) {
    const double fi = fmin(fmax(__builtin_floor(ux), 2.0), (double)(nx - 4)), fj = fmin(fmax(__builtin_floor(uy), 2.0), (double)(ny - 4)),
                 fk = fmin(fmax(__builtin_floor(uz), 2.0), (double)(nz - 4));
    double wx[6], wy[6], wz[6];
    cubic_taps_ideal(ux - fi, wx);
    cubic_taps_ideal(uy - fj, wy);
    cubic_taps_ideal(uz - fk, wz);
    const size_t sj = (size_t)nz, si = (size_t)ny * nz;
    const double *base = M + ((size_t)((int)fi - 2) * ny + (size_t)((int)fj - 2)) * nz + (size_t)((int)fk - 2);
    double x0 = wx[0], x1 = wx[1], x2 = wx[2], x3 = wx[3], x4 = wx[4], x5 = wx[5];
    double f = 0.0;
#pragma unroll 1
    for (int a = 0; a < 6; ++a) {
        double fa = 0.0;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const double *q = base + (size_t)b * sj;
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 6; ++c) s += q[c] * wz[c];
            fa += s * wy[b];
        }
        f += fa * x0;
        x0 = x1, x1 = x2, x2 = x3, x3 = x4, x4 = x5;
        base += si;
    }
    return f;
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
    // (the x taps ROTATE through six named values instead of being indexed by the loop counter: where the compiler keeps this loop
    //  rolled -- the Fermat kernels, which inline four of these per step -- a run-time index put wx, dx, wy, dy into scratch memory,
    //  224 bytes per lane in the round-4 build; a select chain on the counter is turned back into an indexed load.  Where it unrolls
    //  the loop the moves vanish.  Same operations in the same order, bit for bit)
    double x0 = wx[0], x1 = wx[1], x2 = wx[2], x3 = wx[3], x4 = wx[4], x5 = wx[5];
    double e0 = GRAD ? dx[0] : 0.0, e1 = GRAD ? dx[1] : 0.0, e2 = GRAD ? dx[2] : 0.0, e3 = GRAD ? dx[3] : 0.0, e4 = GRAD ? dx[4] : 0.0,
           e5 = GRAD ? dx[5] : 0.0;
    for (int a = 0; a < 6; ++a) {
        double fa = 0.0, fya = 0.0, fza = 0.0;
#pragma unroll
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
        f += fa * x0;
        if (GRAD) {
            fx += fa * e0;
            fy += fya * x0;
            fz += fza * x0;
        }
        x0 = x1, x1 = x2, x2 = x3, x3 = x4, x4 = x5;
        e0 = e1, e1 = e2, e2 = e3, e3 = e4, e4 = e5;
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

// ---- one kernel for every element-wise helper: f(i) for i in [0, n), grid-stride.  F is a small struct of the operands with
// `__device__ void operator()(int64_t i) const` and, where a per-thread set-up pays (device scalars), `__device__ void begin()`.
template <class F>
__device__ __forceinline__ auto map_begin(F &f, int) -> decltype(f.begin(), void()) {
    f.begin();
}
template <class F>
__device__ __forceinline__ void map_begin(F &, long) {}
template <class F>
__global__ __launch_bounds__(256) void k_map(int64_t n, F f) {
    map_begin(f, 0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) f(i);
}

}  // namespace

#endif
