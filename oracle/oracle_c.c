/*
 * oracle_c.c -- plain-C restatement of the straight-ray dTEC forward model and its exact adjoint.
 *
 * TEST INFRASTRUCTURE ONLY: the multi-core CPU baseline that bench.py times beside the GPU
 * kernels ("cpu_baseline", kind "port") and a second, independent parity checker.  Nothing under
 * ionotomo_amd/ links or loads this.  It is itself checked against oracle.py (which is pinned to
 * the reference's golden vectors) in tests/test_oracle_c.py.
 *
 * Algorithm (citations relative to /root/reference/src/ionotomo/):
 *   rays   : z = linspace(z0, tmax, N), x = x0 + px/pz (z - z0), s = (z - z0)/pz
 *            (inversion/fermat.py:64-72,150-174 with n = 1; geometry/calc_rays.py:61-96)
 *   interp : scipy RegularGridInterpolator 'linear' -- i = clip(searchsorted(g, x) - 1, 0, n-2)
 *            (geometry/tri_cubic.py:69-70; tomography/interpolation.py:145-196)
 *   simps  : composite Simpson, non-uniform weights hs/6 (2 - h1/h0), hs^3/(6 h0 h1),
 *            hs/6 (2 - h0/h1); even N: 'avg' rule (tomography/integrate.py:50-74,130-153)
 *   tec    : inversion/forward_equation.py:13-33
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int find_cell(const double *g, int n, double x) {
    int lo = 0, hi = n;                 /* searchsorted(g, x, side='left') */
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (g[mid] < x) lo = mid + 1; else hi = mid;
    }
    int i = lo - 1;
    if (i < 0) i = 0;
    if (i > n - 2) i = n - 2;
    return i;
}

static void basic_simpson(const double *s, int n, double f, double *w) {
    for (int k = 0; k + 2 < n; k += 2) {
        double h0 = s[k + 1] - s[k], h1 = s[k + 2] - s[k + 1], hs = h0 + h1;
        w[k] += f * hs / 6.0 * (2.0 - h1 / h0);
        w[k + 1] += f * hs / 6.0 * (hs * hs / (h0 * h1));
        w[k + 2] += f * hs / 6.0 * (2.0 - h0 / h1);
    }
}

static void simpson_weights_avg(const double *s, int n, double *w) {
    memset(w, 0, sizeof(double) * n);
    if (n == 2) { w[0] = w[1] = 0.5 * (s[1] - s[0]); return; }
    if (n & 1) { basic_simpson(s, n, 1.0, w); return; }
    basic_simpson(s, n - 1, 0.5, w);
    w[n - 1] += 0.25 * (s[n - 1] - s[n - 2]);
    w[n - 2] += 0.25 * (s[n - 1] - s[n - 2]);
    basic_simpson(s + 1, n - 1, 0.5, w + 1);
    w[0] += 0.25 * (s[1] - s[0]);
    w[1] += 0.25 * (s[1] - s[0]);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* tec[r] for R straight rays; returns the number of out-of-grid samples (reference raises) */
int64_t oracle_forward_tec_straight(const double *xv, int nx, const double *yv, int ny, const double *zv, int nz,
                                    const double *M, const double *origins, const double *dirs, int64_t R,
                                    double tmax, int Ns, double *tec, int nthreads) {
    int64_t oob = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel reduction(+ : oob)
    {
        double *buf = (double *)malloc(sizeof(double) * 3 * (size_t)Ns);
        double *s = buf, *f = buf + Ns, *w = buf + 2 * Ns;
#pragma omp for schedule(dynamic, 64)
        for (int64_t r = 0; r < R; ++r) {
            const double *o = origins + 3 * r, *d = dirs + 3 * r;
            const double nrm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            const double px = d[0] / nrm, py = d[1] / nrm, pz = d[2] / nrm;
            const double L = tmax - o[2], step = 1.0 / (double)(Ns - 1);
            for (int k = 0; k < Ns; ++k) {
                const double frac = (k == Ns - 1) ? 1.0 : (double)k * step;
                const double dz = L * frac;
                const double x = o[0] + (px / pz) * dz, y = o[1] + (py / pz) * dz, z = o[2] + dz;
                s[k] = dz / pz;
                if (!(x >= xv[0] && x <= xv[nx - 1] && y >= yv[0] && y <= yv[ny - 1] && z >= zv[0] && z <= zv[nz - 1])) {
                    ++oob;
                    f[k] = 0.0;
                    continue;
                }
                const int i = find_cell(xv, nx, x), j = find_cell(yv, ny, y), kk = find_cell(zv, nz, z);
                const double tx = (x - xv[i]) / (xv[i + 1] - xv[i]);
                const double ty = (y - yv[j]) / (yv[j + 1] - yv[j]);
                const double tz = (z - zv[kk]) / (zv[kk + 1] - zv[kk]);
                const double *p = M + ((size_t)i * ny + j) * nz + kk;
                const size_t sj = nz, si = (size_t)ny * nz;
                double acc = 0.0;
                acc += p[0] * ((1 - tx) * (1 - ty) * (1 - tz));
                acc += p[1] * ((1 - tx) * (1 - ty) * tz);
                acc += p[sj] * ((1 - tx) * ty * (1 - tz));
                acc += p[sj + 1] * ((1 - tx) * ty * tz);
                acc += p[si] * (tx * (1 - ty) * (1 - tz));
                acc += p[si + 1] * (tx * (1 - ty) * tz);
                acc += p[si + sj] * (tx * ty * (1 - tz));
                acc += p[si + sj + 1] * (tx * ty * tz);
                f[k] = acc;
            }
            simpson_weights_avg(s, Ns, w);
            double t = 0.0;
            for (int k = 0; k < Ns; ++k) t += w[k] * f[k];
            tec[r] = t;
        }
        free(buf);
    }
    return oob;
}

/* grad[v] += sum_r w_r sum_k c_k W_kv (serial scatter; exact transpose of the above) */
void oracle_adjoint_straight(const double *xv, int nx, const double *yv, int ny, const double *zv, int nz,
                             const double *origins, const double *dirs, const double *wray, int64_t R, double tmax,
                             int Ns, double *grad) {
    double *buf = (double *)malloc(sizeof(double) * 2 * (size_t)Ns);
    double *s = buf, *w = buf + Ns;
    for (int64_t r = 0; r < R; ++r) {
        const double *o = origins + 3 * r, *d = dirs + 3 * r;
        const double nrm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const double px = d[0] / nrm, py = d[1] / nrm, pz = d[2] / nrm;
        const double L = tmax - o[2], step = 1.0 / (double)(Ns - 1);
        for (int k = 0; k < Ns; ++k) s[k] = L * ((k == Ns - 1) ? 1.0 : (double)k * step) / pz;
        simpson_weights_avg(s, Ns, w);
        for (int k = 0; k < Ns; ++k) {
            const double dz = L * ((k == Ns - 1) ? 1.0 : (double)k * step);
            const double x = o[0] + (px / pz) * dz, y = o[1] + (py / pz) * dz, z = o[2] + dz;
            const int i = find_cell(xv, nx, x), j = find_cell(yv, ny, y), kk = find_cell(zv, nz, z);
            const double tx = (x - xv[i]) / (xv[i + 1] - xv[i]);
            const double ty = (y - yv[j]) / (yv[j + 1] - yv[j]);
            const double tz = (z - zv[kk]) / (zv[kk + 1] - zv[kk]);
            double *p = grad + ((size_t)i * ny + j) * nz + kk;
            const size_t sj = nz, si = (size_t)ny * nz;
            const double c = wray[r] * w[k];
            p[0] += c * ((1 - tx) * (1 - ty) * (1 - tz));
            p[1] += c * ((1 - tx) * (1 - ty) * tz);
            p[sj] += c * ((1 - tx) * ty * (1 - tz));
            p[sj + 1] += c * ((1 - tx) * ty * tz);
            p[si] += c * (tx * (1 - ty) * (1 - tz));
            p[si + 1] += c * (tx * (1 - ty) * tz);
            p[si + sj] += c * (tx * ty * (1 - tz));
            p[si + sj + 1] += c * (tx * ty * tz);
        }
    }
    free(buf);
}
