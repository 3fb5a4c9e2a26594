// fused vector passes of the inversion loop: ray-sized residual / dot kernels and grid updates on the active node set
#ifndef IONO_SOLVER_KERNELS_H
#define IONO_SOLVER_KERNELS_H

namespace {

// ------------------------------------------------------------------------------------------------
// A CGLS / SIRT iteration is the two ray kernels plus a handful of vector updates whose coefficients are ratios of
// dot products (iterative_newton.py:542-554; geometry/oct_trees/Inversion.py:533,559,564).  Round 1 formed them with
// ~15 separate torch kernels over full 256^3 vectors (0.27 ms of a 1.34 ms iteration).  Here:
//   * dot products never leave the device and never use atomics: a producing kernel writes one partial per workgroup
//     into a fixed array of IONO_NPART doubles; a consuming kernel sums that array in a fixed order (every workgroup
//     gets the same bits, so replicas of the model on different ranks cannot drift apart);
//   * the ray geometry is fixed for a whole inversion and the rays reach only a fraction of the box (21 % at the bench
//     shape), so grid-sized vectors are kept COMPACT over the active node set (sorted int32 indices); the full grid
//     the forward kernel reads is updated by scattering through the index, the back-projected update is gathered
//     (and its nodes re-zeroed) through it.
// ------------------------------------------------------------------------------------------------
#define IONO_NPART 512      // workgroups (and partials) of every dot-producing launch

// sum over the 256 threads of a workgroup, returned to every thread (fixed order: deterministic)
__device__ __forceinline__ double block_sum_bcast(double v) {
    __shared__ double red[4];
    const double w = wave_sum_dpp(v);
    __syncthreads();                       // red[] may still be read from a previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}
// value of a device scalar given as (pointer, count): null = 1, count 1 = plain scalar, otherwise a partials array
__device__ __forceinline__ double read_scalar(const double *__restrict__ p, int n) {
    if (!p) return 1.0;
    if (n == 1) return p[0];
    double v = 0.0;
    for (int t = threadIdx.x; t < n; t += blockDim.x) v += p[t];
    return block_sum_bcast(v);
}

// out[r] = s1[r] * (a * (tec[r] - tec[i0, p]) + b * dobs[r]);  partial[blk] = sum out^2 * s2[r]
// (s1, s2, dobs, partial nullable).  CGLS: q = W^1/2 (A p);  r0 = W^1/2 (d - A x).  SIRT: r = d - A x, S = 1/2 sum r^2 W.
__global__ __launch_bounds__(256) void k_rays_combine(const double *__restrict__ tec, const double *__restrict__ dobs,
                                                      const double *__restrict__ s1, const double *__restrict__ s2, int Na,
                                                      int64_t NtNd, int i0, double a, double b, double *__restrict__ out,
                                                      double *__restrict__ partial) {
    const int64_t n = (int64_t)Na * NtNd;
    double acc = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = r % NtNd;
        double v = a * (tec[r] - tec[(int64_t)i0 * NtNd + p]);
        if (dobs) v += b * dobs[r];
        if (s1) v *= s1[r];
        out[r] = v;
        acc += s2 ? v * v * s2[r] : v * v;
    }
    if (partial) {
        const double t = block_sum_bcast(acc);
        if (threadIdx.x == 0) partial[blockIdx.x] = t;
    }
}

// y = alpha x + beta y,  alpha = sa * an / ad,  beta = bn / bd (device scalars);  partial[blk] = sum y^2
__global__ __launch_bounds__(256) void k_axpby_dot(double *__restrict__ y, const double *__restrict__ x, int64_t n,
                                                   const double *an, int ann, const double *ad, int adn, double sa,
                                                   const double *bn, int bnn, const double *bd, int bdn,
                                                   double *__restrict__ partial) {
    const double alpha = sa * read_scalar(an, ann) / read_scalar(ad, adn);
    const double beta = read_scalar(bn, bnn) / read_scalar(bd, bdn);
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = fma(alpha, x[i], beta * y[i]);
        y[i] = v;
        acc += v * v;
    }
    if (partial) {
        const double t = block_sum_bcast(acc);
        if (threadIdx.x == 0) partial[blockIdx.x] = t;
    }
}

// ---- one coherence window (<= IONO_SMALL_RAYS rays): the ray-sized passes of an iteration in ONE launch -----------------------------
// At that size every pass is a few microseconds of launch and the dot products between them are what forces separate kernels; a
// single workgroup holds all the rays, so the dots are workgroup reductions and the three passes become phases of one kernel:
//   CG   (MODE 0): q = Wh (tec - tec[i0]), qq = <q, q>;  r -= (gamma / qq) q, rr = <r, r>;  w = differential weights of (r Wh)
//   SIRT (MODE 1): r = dobs - (tec - tec[i0]), S2 = sum r^2 Wt;                             w = differential weights of (r L)
// with the arithmetic of k_rays_combine, k_axpby_dot and k_ray_weights<2> element for element.  The weights feed the back-projection directly
// (iono_adjoint_straight_dev), which then needs no k_ray_weights launch either: three launches fewer per CG iteration.
#define IONO_SMALL_RAYS 32768
__device__ __forceinline__ double block_sum_bcast_1024(double v) {
    __shared__ double red16[16];
    const double w = wave_sum_dpp(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red16[threadIdx.x >> 6] = w;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red16[i];
    return t;
}
template <int MODE>
__global__ __launch_bounds__(1024) void k_small_ray_pass(const double *__restrict__ tec, const double *__restrict__ dobs,
                                                         const double *__restrict__ s1, const double *__restrict__ s2, double *__restrict__ r,
                                                         double *__restrict__ q, int Na, int64_t NtNd, int i0, const double *gam, int gamn,
                                                         double *__restrict__ dot1, double *__restrict__ dot2, double *__restrict__ w) {
    const int64_t n = (int64_t)Na * NtNd;
    if (MODE == 0) {
        double acc = 0.0;
        for (int64_t t = threadIdx.x; t < n; t += blockDim.x) {
            const int64_t p = t % NtNd;
            double v = 1.0 * (tec[t] - tec[(int64_t)i0 * NtNd + p]);
            v *= s1[t];
            q[t] = v;
            acc += v * v;
        }
        const double qq = block_sum_bcast_1024(acc);
        if (threadIdx.x == 0) dot1[0] = qq;
        // gamma: a plain scalar or a partials array of at most IONO_NPART entries, summed in the order read_scalar uses
        double gs = 0.0;
        if (gamn == 1) gs = gam[0];
        else {
            for (int t = threadIdx.x; t < gamn; t += blockDim.x) gs += gam[t];
            gs = block_sum_bcast_1024(gs);
        }
        const double alpha = -1.0 * gs / qq;
        acc = 0.0;
        for (int64_t t = threadIdx.x; t < n; t += blockDim.x) {
            const double v = fma(alpha, q[t], 1.0 * r[t]);
            r[t] = v;
            acc += v * v;
        }
        const double rr = block_sum_bcast_1024(acc);
        if (threadIdx.x == 0) dot2[0] = rr;
    } else {
        double acc = 0.0;
        for (int64_t t = threadIdx.x; t < n; t += blockDim.x) {
            const int64_t p = t % NtNd;
            double v = -1.0 * (tec[t] - tec[(int64_t)i0 * NtNd + p]);
            v += 1.0 * dobs[t];
            r[t] = v;
            acc += v * v * s2[t];
        }
        const double S2 = block_sum_bcast_1024(acc);
        if (threadIdx.x == 0) dot1[0] = S2;
    }
    __syncthreads();      // r is complete (written by this workgroup: visible after the barrier)
    // differential weights of v = r * scale, as k_ray_weights<2> forms them: one wave per (time, direction) pair, lanes = antennas
    const int lane = threadIdx.x & 63;
    for (int64_t p = threadIdx.x >> 6; p < NtNd; p += blockDim.x >> 6) {
        double s = 0.0;
        for (int a = lane; a < Na; a += 64) s += r[(int64_t)a * NtNd + p] * s1[(int64_t)a * NtNd + p];
        s = wave_sum_dpp(s);
        for (int a = lane; a < Na; a += 64) {
            const int64_t t = (int64_t)a * NtNd + p;
            const double v = r[t] * s1[t];
            w[t] = a == i0 ? v - s : v;
        }
    }
}

// out[t] = full[idx[t]] (then full[idx[t]] = 0 if `zero`);  partial[blk] = sum out^2
__global__ __launch_bounds__(256) void k_compact_gather(double *__restrict__ full, const int *__restrict__ idx, int64_t n,
                                                        double *__restrict__ out, int zero, double *__restrict__ partial) {
    double acc = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int j = idx[t];
        const double v = full[j];
        if (zero) full[j] = 0.0;
        out[t] = v;
        acc += v * v;
    }
    if (partial) {
        const double s = block_sum_bcast(acc);
        if (threadIdx.x == 0) partial[blockIdx.x] = s;
    }
}

struct CompactScatter {
    double *__restrict__ full;
    const int *__restrict__ idx;
    const double *__restrict__ src;
    __device__ __forceinline__ void operator()(int64_t t) const { full[idx[t]] = src[t]; }
};

// CGLS tail in one pass:  x += alpha p;  p = s + beta p;  full_p[idx] = p      (alpha = an / ad, beta = bn / bd)
__global__ __launch_bounds__(256) void k_compact_cg_update(double *__restrict__ x, double *__restrict__ p,
                                                           const double *__restrict__ s, const int *__restrict__ idx, int64_t n,
                                                           double *__restrict__ full_p, const double *an, int ann,
                                                           const double *ad, int adn, const double *bn, int bnn, const double *bd,
                                                           int bdn) {
    const double alpha = read_scalar(an, ann) / read_scalar(ad, adn);
    const double beta = read_scalar(bn, bnn) / read_scalar(bd, bdn);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const double pv = p[t];
        x[t] = fma(alpha, pv, x[t]);
        const double pn = fma(beta, pv, s[t]);
        p[t] = pn;
        full_p[idx[t]] = pn;
    }
}

// SIRT tail in one pass:  u = relax C s (s = full_s[idx], re-zeroed);  x += u (clamped at 0 if nonneg);  full_x[idx] = x;
// partial[blk] = max |x_new - x_old| (for the reference's stopping rule)
__global__ __launch_bounds__(256) void k_compact_sirt_update(double *__restrict__ x, const double *__restrict__ C,
                                                             double *__restrict__ full_s, const int *__restrict__ idx, int64_t n,
                                                             double *__restrict__ full_x, double relax, int nonneg,
                                                             double *__restrict__ partial_max) {
    double mx = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int j = idx[t];
        const double s = full_s[j];
        full_s[j] = 0.0;
        const double xo = x[t];
        double xn = fma(relax * C[t], s, xo);
        if (nonneg) xn = fmax(xn, 0.0);
        x[t] = xn;
        full_x[j] = xn;
        mx = fmax(mx, fabs(xn - xo));
    }
    if (partial_max) {
        __shared__ double red[4];
        const double w = wave_minmax_dpp<true>(mx);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
        __syncthreads();
        if (threadIdx.x == 0) partial_max[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    }
}

}  // namespace

#endif
