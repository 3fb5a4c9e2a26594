// node-stationary back-projection: ray segments binned by grid box once per geometry, each box reduced in LDS and flushed once
#ifndef IONO_BINNED_KERNELS_H
#define IONO_BINNED_KERNELS_H

namespace {

// ------------------------------------------------------------------------------------------------
// The ray-stationary tile kernel (iono_adjoint_kernels.h) walks bundles of neighbouring rays and flushes an LDS tile
// per bundle and 64-sample slab: the same grid nodes are flushed by hundreds of bundles (all rays of a station share
// its low-altitude cells; 100 timesteps of a direction nearly coincide), 12.3 M 64-B atomic requests per launch at the
// bench shape against 0.44 M distinct 64-B segments under the fan -- the kernel sits on the memory-side atomic rate.
// The ray geometry is fixed for a whole inversion, so the transpose can be organised the other way round, ONCE:
//   * the grid is cut into boxes of BIN_SX x BIN_SY x BIN_SZ cells (+ a halo of BIN_H cells in x and y);
//   * every ray is cut into segments of <= 16 consecutive samples that lie in one z-layer of boxes; a segment goes to
//     the box around the middle of its (x, y) extent (host side, iono_adjoint_plan_dev);
//   * segments are sorted by box; a workgroup takes a unit (a box and up to BIN_UNIT of its segments), accumulates them
//     in an LDS image of the box (16 lanes per segment, lanes = consecutive samples = consecutive z words: conflict-free
//     LDS atomics) and flushes the box ONCE.
// Every sample of every ray is in exactly one segment by construction, and a segment is cut where its (x, y) extent would leave
// the halo (round 5: a steep ray gets shorter segments, `outside` is 0 for every geometry); the kernels still send a contribution
// that falls outside the box image straight to global atomics, so the result never depends on the quality of the binning.
// ------------------------------------------------------------------------------------------------
#define BIN_SX 8
#define BIN_SY 8
#ifndef BIN_SZ
#define BIN_SZ 15                            // cells per z-layer of boxes (A/B builds: -DBIN_SZ=7 -DBIN_BZP=8, profiles/r05_backprojection_occupancy.json)
#endif
#ifndef BIN_H
#define BIN_H 3
#endif
#define BIN_BX (BIN_SX + 2 * BIN_H + 1)      // nodes of the box image along x (15)
#define BIN_BY (BIN_SY + 2 * BIN_H + 1)
#define BIN_BZ (BIN_SZ + 1)                  // 16 z nodes
#ifndef BIN_BZP
#define BIN_BZP 16                           // z stride of the image in words.  NOT padded: ds_add_f64 is served in four groups of 16 lanes
#endif                                       // with bank = word mod 16, a segment = one group = 16 consecutive levels, and a segment that
                                             // changes column half-way keeps distinct banks only if the column strides are multiples of 16
                                             // words (17: every such segment met a 2-way conflict: 16.2 against 8.2 cycles per instruction,
                                             // profiles/r04_lds_atomic_probe.json; nine segments in ten change column at the bench geometry)
#define BIN_SEG 16                           // most samples per segment = lanes per segment (a plan may use 4 or 8: `segl`)
#define BIN_ENTRY_PAD 1024                   // zero entries behind the list: the kernels prefetch three passes (of <= 1024 / 4 segments) ahead
#ifndef BIN_UNIT
#define BIN_UNIT 1024                        // segments per work unit (round 4, bench geometry, ms per back-projection: 256: 0.401, 512: 0.337, 1024: 0.322,
                                             // 2048: 0.325, 4096: 0.324: zeroing + flushing a 29-KB image per unit against load balance)
#endif
#define BIN_TILE (BIN_BX * BIN_BY * BIN_BZP)
#ifndef BIN_THREADS
#define BIN_THREADS 256                      // threads per workgroup of k_adjoint_binned (one box image per workgroup)
#endif

struct BinUnit {
    int x0, y0, z0;      // first node of the box image
    int e_lo, e_hi;      // its segments: entries [e_lo, e_hi)
};

// ray parameters in ideal grid coordinates, computed ONCE with the kernels' own arithmetic (load_uray / load_uray_cubic)
// so that validity and sample positions are those of the forward kernels: uray[r] = fx0, dfx, fy0, dfy, fz0, dfz, h, valid
template <bool CUBIC>
struct PlanUrays {                 // over the R rays
    GridView g;
    const double *__restrict__ origins;
    const double *__restrict__ dirs;
    double tmax;
    int Ns;
    double *__restrict__ uray;
    uint2 *__restrict__ hash;
    __device__ __forceinline__ void operator()(int64_t r) const {
        const URay u = CUBIC ? load_uray_cubic(g, origins, dirs, r, tmax, Ns) : load_uray(g, origins, dirs, r, tmax, Ns);
        hash[r] = ray_hash(origins, dirs, r);
        double *o = uray + r * 8;
        o[0] = u.fx0, o[1] = u.dfx, o[2] = u.fy0, o[3] = u.dfy, o[4] = u.fz0, o[5] = u.dfz, o[6] = u.h, o[7] = u.valid ? 1.0 : 0.0;
    }
};

// The plan itself, on the device (one thread per ray; a host loop over R x Ns samples took 0.26 s at the bench shape, eight
// times a 50-iteration inversion).  Segments: <= segl consecutive samples of a ray inside one z-layer of boxes, filed under
// the (x, y) box around the middle of their extent; every sample of every valid ray lands in exactly one segment.
// segl (4, 8 or 16) = the lanes a segment occupies in the back-projection: an LDS float atomic costs the same ~14 cycles
// whatever its lane mask, so the plan picks the width that leaves the fewest lanes empty (64 samples through 256 cells
// put 3.8 samples into a 15-cell layer: 16-lane segments ran the atomics 24 % full).
//   pass 1 (EMIT = false): nseg[r] (-1: the ray leaves the grid) and the number of segments per box;
//   pass 2 (EMIT = true, after an exclusive scan of the box counts on the host): entries[box_start + slot] = (ray, first
//           sample | count << 16 | ordinal << 24), slot from an atomic counter of the box (order inside a box is immaterial:
//           every segment of a unit goes to the same LDS image).
// Cells use the kernels' own expression floor(|fma(k, df, f0)|), so a segment never disagrees with the samples it holds.
__device__ __forceinline__ int plan_cell(double f0, double df, int k, int n) {
    return (int)fmin(__builtin_floor(__builtin_fabs(__builtin_fma((double)k, df, f0))), (double)(n - 2));
}
// The tiles (LMT_X x LMT_Y x LMT_Z nodes, iono_cubic_kernels.h) the samples of the planned rays add into, with a margin of one cell
// either side (the corner nodes c .. c + 1 of a sample's cell c: tiles of c - 1 .. c + 2): touch[tile] = 1.  Thread per ray;
// a store only when the tile range changed since the previous sample.
__global__ __launch_bounds__(256) void k_plan_touch(const double *__restrict__ uray, int64_t R, int Ns, int nx, int ny, int nz, int nty,
                                                    int ntz, unsigned char *__restrict__ touch) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
        const double *u = uray + r * 8;
        if (u[7] == 0.0) continue;
        const double fx0 = u[0], dfx = u[1], fy0 = u[2], dfy = u[3], fz0 = u[4], dfz = u[5];
        int px = -1, py = -1, pz = -1;
        for (int k = 0; k < Ns; ++k) {
            const int cx = plan_cell(fx0, dfx, k, nx), cy = plan_cell(fy0, dfy, k, ny), cz = plan_cell(fz0, dfz, k, nz);
            const int xa = max(cx - 1, 0) / LMT_X, xb = min(cx + 2, nx - 1) / LMT_X, ya = max(cy - 1, 0) / LMT_Y, yb = min(cy + 2, ny - 1) / LMT_Y;
            const int za = max(cz - 1, 0) / LMT_Z, zb = min(cz + 2, nz - 1) / LMT_Z;
            const int kx = xa | (xb << 16), ky = ya | (yb << 16), kz = za | (zb << 16);
            if (kx == px && ky == py && kz == pz) continue;
            px = kx, py = ky, pz = kz;
            for (int a = xa; a <= xb; ++a)
                for (int b = ya; b <= yb; ++b)
                    for (int cc = za; cc <= zb; ++cc) touch[((int64_t)a * nty + b) * ntz + cc] = 1;
        }
    }
}
// Wave-aggregated counter update: the 64 rays of a wave are neighbours and file their k-th segments under a handful of boxes,
// so one lane per distinct box adds the number of lanes that share it (per-segment atomics on a few hot addresses made the two
// passes 8.6 + 9.5 ms; aggregated: see profiles/tools/time_plan.py).  Must be called by the whole wave; returns this lane's slot
// (old counter value + rank among the lanes with the same key) when WANT_SLOT.
template <bool WANT_SLOT>
__device__ __forceinline__ int wave_counter_add(int *__restrict__ ctr, int key, bool active) {
    const int lane = threadIdx.x & 63;
    int slot = 0;
    unsigned long long todo = __ballot(active);
    while (todo) {                                           // wave-uniform
        const int leader = __ffsll((long long)todo) - 1;
        const int k0 = __builtin_amdgcn_readlane(key, leader);
        const bool mine = active && key == k0;
        const unsigned long long same = __ballot(mine);
        int base = 0;
        if (lane == leader) {
            if (WANT_SLOT) base = atomicAdd(ctr + k0, __popcll(same));
            else atomicAdd(ctr + k0, __popcll(same));
        }
        if (WANT_SLOT) {
            base = __builtin_amdgcn_readlane(base, leader);
            if (mine) slot = base + __popcll(same & ((1ull << lane) - 1ull));
        }
        todo &= ~same;
    }
    return slot;
}
template <bool EMIT>
__global__ __launch_bounds__(256) void k_plan_segments(const double *__restrict__ uray, int64_t R, int Ns, int nx, int ny, int nz,
                                                       int nbx, int nby, int nbz, int segl, int *__restrict__ nseg,
                                                       int *__restrict__ box_count, const int *__restrict__ box_start,
                                                       int *__restrict__ box_fill, uint2 *__restrict__ entries,
                                                       unsigned long long *__restrict__ outside, const int *__restrict__ order) {
    // whole waves walk the ray range together (the counter update is a wave operation): the loop bound is rounded up to a
    // multiple of the grid's thread count and lanes beyond R idle
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r0 = (int64_t)blockIdx.x * blockDim.x; r0 < R; r0 += nthreads) {
        // `order` (the forward plan's walk, when it exists for these rays): the 64 rays of a wave then nearly coincide and file their
        // k-th segments under one to three boxes instead of twenty to forty -- the counter update below loops over the DISTINCT boxes of
        // the wave and waits for an atomic's return in every turn (0.7 + 0.8 ms for the two passes in ray order)
        const int64_t q = r0 + threadIdx.x;
        const int64_t r = q < R ? (order ? (int64_t)order[q] : q) : R;
        bool valid = r < R;
        double fx0 = 0, dfx = 0, fy0 = 0, dfy = 0, fz0 = 0, dfz = 0;
        if (valid) {
            const double *u = uray + r * 8;
            valid = u[7] != 0.0;
            fx0 = u[0], dfx = u[1], fy0 = u[2], dfy = u[3], fz0 = u[4], dfz = u[5];
            if (!valid && !EMIT) nseg[r] = -1;
        }
        int k = valid ? 0 : Ns, j = 0, out = 0;
        while (__any(k < Ns)) {
            const bool active = k < Ns;
            int box = 0, ke = k;
            if (active) {
                const int zb = plan_cell(fz0, dfz, k, nz) / BIN_SZ;
                const int xa = plan_cell(fx0, dfx, k, nx), ya = plan_cell(fy0, dfy, k, ny);
                ke = k + 1;
                // (... and whose (x, y) extent fits the halo whatever its middle box: a steep ray is cut into shorter segments instead of
                //  sending samples past the image to global atomics -- `outside` stays 0 for every geometry)
                int xb = xa, yb = ya;
                while (ke < Ns && ke - k < segl && plan_cell(fz0, dfz, ke, nz) / BIN_SZ == zb) {
                    const int xn = plan_cell(fx0, dfx, ke, nx), yn = plan_cell(fy0, dfy, ke, ny);
                    if (abs(xn - xa) > 2 * BIN_H - 1 || abs(yn - ya) > 2 * BIN_H - 1) break;
                    xb = xn, yb = yn, ++ke;
                }
                const int bi = min(nbx - 1, (xa + xb + 1) / (2 * BIN_SX)), bj = min(nby - 1, (ya + yb + 1) / (2 * BIN_SY));
                box = (bi * nby + bj) * nbz + zb;
                if (!EMIT) {
                    const int x0 = bi * BIN_SX - BIN_H, y0 = bj * BIN_SY - BIN_H;
                    out += (min(xa, xb) < x0) | (max(xa, xb) > x0 + BIN_BX - 2) | (min(ya, yb) < y0) | (max(ya, yb) > y0 + BIN_BY - 2);
                }
            }
            if (EMIT) {
                int slot = wave_counter_add<true>(box_fill, box, active);
                if (active && order) {
                    // walked in the forward plan's order, neighbouring slots of a box are nearly coincident rays: the four segments of a
                    // back-projection wave would add into the SAME LDS words at once (+4.5 % measured).  A multiplicative
                    // permutation of the box's slots spreads them (a bijection: 7919 is prime and does not divide the count).
                    const unsigned cnt = (unsigned)box_count[box];
                    if (cnt % 7919u) slot = (int)(((unsigned long long)(unsigned)slot * 7919ull) % cnt);
                }
                if (active)
                    entries[box_start[box] + slot] = make_uint2((unsigned)r, (unsigned)k | ((unsigned)(ke - k) << 16) | ((unsigned)(j & 255) << 24));
            } else {
                wave_counter_add<false>(box_count, box, active);
            }
            if (active) k = ke, ++j;
        }
        if (!EMIT && valid) {
            nseg[r] = j;
            if (out) atomicAdd(outside, (unsigned long long)out);
        }
    }
}

// The back-projection works from the plan's OWN ray records (uray), so a planned array edited in place would silently give the
// old rays' answer.  Every planned launch therefore re-hashes the rays it is handed (12.5 MB read at the bench shape, 3 us) against
// the hashes the plan recorded: a ray that differs has its record poisoned (h = NaN: every node it touches comes out NaN, never a
// plausible number) and raises flags[2] (iono_plan_stale -> the host layer turns it into IONO_ERR_ARG); the poison lasts until the rays
// are planned again (iono_adjoint_plan_dev), and so does the flag: every later launch on this plan raises it again.
__device__ __forceinline__ void plan_verify_ray(const double *__restrict__ origins, const double *__restrict__ dirs, int64_t r,
                                                const uint2 *__restrict__ hash, double *__restrict__ uray, int *__restrict__ flags) {
    const uint2 want = hash[r], have = ray_hash(origins, dirs, r);
    // (a record poisoned by an EARLIER launch stays poisoned even if the caller has since restored the ray in place -- the hashes
    //  then match again -- so it raises the flag on every launch until the rays are planned anew: the flag is sticky only until it
    //  is read once, the NaN is permanent, and the two must not come apart: ADVICE r4)
    const double2 hv = *(const double2 *)(uray + r * 8 + 6);                     // (h, valid)
    const bool poisoned = hv.y != 0.0 && hv.x != hv.x;
    if ((want.x != have.x) | (want.y != have.y) | poisoned) {
        uray[r * 8 + 6] = nan("");
        atomicOr(flags + 2, 1);
    }
}
__global__ __launch_bounds__(256) void k_plan_verify(const double *__restrict__ origins, const double *__restrict__ dirs, int64_t R,
                                                     const uint2 *__restrict__ hash, double *__restrict__ uray, int *__restrict__ flags) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x)
        plan_verify_ray(origins, dirs, r, hash, uray, flags);
}

// per-ray weights of the fused modes (residual / differential), one value per ray: the binned kernel visits a ray once
// per segment, so the reference-antenna sums are formed here, once.  Lanes = RSTEP_PT consecutive (time, direction) pairs p (ray
// index a NtNd + p: every load of a wave is a few contiguous 128-byte runs), the 64 / RSTEP_PT x RSTEP_WAVES lane groups of a workgroup
// = groups of antennas; the column sum over antennas goes through LDS in a fixed order.  (Round 5: lanes = antennas -- every lane of
// a load in a different row of the arrays, the ray check included -- took 21 us of a 0.28 ms back-projection.)
// (+ the plan's ray check of the launch that follows, in the same pass over the rays: `origins` non-null)
#define RSTEP_WAVES 16      // waves of a workgroup of k_ray_weights / k_rays_step
#define RSTEP_PT 16         // consecutive (time, direction) pairs per workgroup (a 128-byte run per load); 64 / RSTEP_PT antenna groups per wave
template <int MODE>
__global__ __launch_bounds__(64 * RSTEP_WAVES) void k_ray_weights(const double *__restrict__ tec, const double *__restrict__ dobs,
                                                                  const double *__restrict__ cdct, int Na, int64_t NtNd, int i0,
                                                                  double *__restrict__ w, const double *__restrict__ origins = nullptr,
                                                                  const double *__restrict__ dirs = nullptr,
                                                                  const uint2 *__restrict__ hash = nullptr, double *__restrict__ uray = nullptr,
                                                                  int *__restrict__ flags = nullptr) {
    constexpr int AS = 64 / RSTEP_PT, NG = RSTEP_WAVES * AS;      // antenna groups per wave / per workgroup
    __shared__ double ssum[NG][RSTEP_PT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pl = lane & (RSTEP_PT - 1), gidx = wv * AS + lane / RSTEP_PT;
    const int a_lo = Na * gidx / NG, a_hi = Na * (gidx + 1) / NG;
    for (int64_t p0 = (int64_t)blockIdx.x * RSTEP_PT; p0 < NtNd; p0 += (int64_t)gridDim.x * RSTEP_PT) {
        const int64_t p = p0 + pl;
        const bool on = p < NtNd;
        const double tref = (MODE != 2 && on) ? tec[(int64_t)i0 * NtNd + p] : 0.0;
        double s = 0.0;
        if (on) {
            for (int a = a_lo; a < a_hi; ++a) {
                const int64_t r = (int64_t)a * NtNd + p;
                if (origins) plan_verify_ray(origins, dirs, r, hash, uray, flags);
                const double v = dd_of<MODE>(tec, dobs, cdct, tref, r);
                w[r] = v;                                    // (the reference-antenna row is corrected below)
                s += v;
            }
        }
        ssum[gidx][pl] = s;
        __syncthreads();
        if (on && i0 >= a_lo && i0 < a_hi) {                 // the thread that wrote w[i0, p] (its own store)
            double tot = 0.0;
#pragma unroll 8
            for (int t = 0; t < NG; ++t) tot += ssum[t][pl];
            const int64_t r = (int64_t)i0 * NtNd + p;
            w[r] = w[r] - tot;
        }
        __syncthreads();
    }
}

// ---- one pass over the rays where an iteration had two or three (round 4: launches per iteration) -----------------------------------
//  MODE 0 (CGLS): r -= (an / ad) q, partial = sum r^2;                        w = differential weights of r * scale
//                 (= k_axpby_dot + k_ray_weights<2>: 7 -> 6 launches per iteration)
//  MODE 1 (SIRT): v = dobs - (tec - tec[i0]), partial = sum v^2 * weight;    w = differential weights of v * scale; r (nullable) = v
//                 (= k_rays_combine + k_ray_weights<2>: 5 -> 4)
// + the plan's ray check of the back-projection that follows (origins non-null).  One wave per (time, direction) pair, lanes =
// antennas; one partial per workgroup, summed in a fixed order by the consumer (no atomics: identical bits on every rank).
template <int MODE>
__global__ __launch_bounds__(64 * RSTEP_WAVES) void k_rays_step(const double *__restrict__ tq, const double *__restrict__ dobs,
                                                                const double *__restrict__ scale, const double *__restrict__ weight,
                                                                double *__restrict__ r, const double *an, int ann, const double *ad, int adn,
                                                                int Na, int64_t NtNd, int i0, double *__restrict__ w,
                                                                double *__restrict__ partial, int npartial,
                                                                const double *__restrict__ origins, const double *__restrict__ dirs,
                                                                const uint2 *__restrict__ hash, double *__restrict__ uray,
                                                                int *__restrict__ flags) {
    // lanes = consecutive pairs p (ray index a NtNd + p: every load of a wave is one contiguous run), waves = groups of antennas; the
    // sum over antennas of a pair is a sum over the waves of a workgroup, through LDS, in a fixed order
    constexpr int AS = 64 / RSTEP_PT, NG = RSTEP_WAVES * AS;      // antenna groups per wave / per workgroup
    __shared__ double ssum[NG][RSTEP_PT];
    __shared__ double sacc[RSTEP_WAVES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double alpha = 0.0;
    if (MODE == 0) {
        // an / ad exactly as read_scalar() forms them in the other consumers of these scalars (k_axpby_dot, k_compact_cg_update): the
        // first 256 threads take the entries t, t + 256, ..., four wave sums, added in the same order -- the SAME bits, so x += alpha p
        // and r -= alpha q use one alpha
        __shared__ double red[4];
        auto rd = [&](const double *__restrict__ pp, int n) {
            if (!pp) return 1.0;
            if (n == 1) return pp[0];
            double v = 0.0;
            if (threadIdx.x < 256)
                for (int t = threadIdx.x; t < n; t += 256) v += pp[t];
            const double ws = wave_sum_dpp(v);
            __syncthreads();
            if (wv < 4 && lane == 0) red[wv] = ws;
            __syncthreads();
            return ((red[0] + red[1]) + red[2]) + red[3];
        };
        const double va = rd(an, ann), vd = rd(ad, adn);
        alpha = -1.0 * va / vd;
    }
    const int pl = lane & (RSTEP_PT - 1), gidx = wv * AS + lane / RSTEP_PT;
    const int a_lo = Na * gidx / NG, a_hi = Na * (gidx + 1) / NG;
    double acc = 0.0;
    for (int64_t p0 = (int64_t)blockIdx.x * RSTEP_PT; p0 < NtNd; p0 += (int64_t)gridDim.x * RSTEP_PT) {
        const int64_t p = p0 + pl;
        const bool on = p < NtNd;
        const double tref = (MODE == 1 && on) ? tq[(int64_t)i0 * NtNd + p] : 0.0;
        double s = 0.0;
        if (on) {
            for (int a = a_lo; a < a_hi; ++a) {
                const int64_t i = (int64_t)a * NtNd + p;
                if (origins) plan_verify_ray(origins, dirs, i, hash, uray, flags);
                double v;
                if (MODE == 0) {
                    v = fma(alpha, tq[i], 1.0 * r[i]);
                    r[i] = v;
                    acc += v * v;
                } else {
                    v = -1.0 * (tq[i] - tref) + 1.0 * dobs[i];
                    if (r) r[i] = v;
                    acc += weight ? v * v * weight[i] : v * v;
                }
                const double vs = scale ? v * scale[i] : v;
                w[i] = vs;                                   // (the reference-antenna row is corrected below)
                s += vs;
            }
        }
        ssum[gidx][pl] = s;
        __syncthreads();
        if (on && i0 >= a_lo && i0 < a_hi) {                 // the thread that wrote w[i0, p] (its own store)
            double tot = 0.0;
#pragma unroll 8
            for (int t = 0; t < NG; ++t) tot += ssum[t][pl];
            const int64_t i = (int64_t)i0 * NtNd + p;
            w[i] = w[i] - tot;
        }
        __syncthreads();
    }
    if (partial) {
        const double t = wave_sum_dpp(acc);
        if (lane == 0) sacc[wv] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int t2 = 0; t2 < RSTEP_WAVES; ++t2) tot += sacc[t2];
            partial[blockIdx.x] = tot;
        }
        // entries beyond the grid of this launch read as zero by the consumers (which sum all `npartial` of them)
        if (blockIdx.x == 0)
            for (int t2 = (int)gridDim.x + (int)threadIdx.x; t2 < npartial; t2 += (int)blockDim.x) partial[t2] = 0.0;
    }
}

// ---- deterministic back-projection (FIX): the box image and the grid accumulate 64-bit FIXED-POINT integers ---------------------------
// Integer addition is associative: whatever order the LDS and memory atomics are served in, the sums are the same bits run after run
// (float atomics differ in the last bits from launch to launch, which CG amplifies once it has converged to the noise).
// Scale = 2^e with  (largest contribution M) * 2^e <= 2^(62 - fixbits),  fixbits >= 12 bounding the contributions a node can receive
// (the plan knows: 8 boxes' segments x lanes), so a sum never leaves int64 and a single value stays below 2^50: the conversion is ONE
// fma against 1.5 * 2^52 (the integer sits in the mantissa) + a 64-bit subtraction.  M = max_r |w_r h_r| * 4/3 (Simpson), found by
// k_fix_absmax for the weights of THIS launch; the grid of integers is turned into float64 and re-zeroed by FixConvert (k_map).
// A NaN weight (or a ray record poisoned by plan_verify_ray) gives M = inf -> scale NaN -> NaN at every node the launch reaches.
#define FIX_MAGIC 6755399441055744.0       // 1.5 * 2^52
__device__ __forceinline__ double fix_scale(unsigned long long maxbits, int fixbits) {
    const double M = __longlong_as_double((long long)maxbits) * (4.0 / 3.0) * (1.0 + 0x1p-40);
    if (M == 0.0) return 1.0;
    if (!(M < 1.7e308)) return __builtin_nan("");
    return ldexp(1.0, 62 - fixbits - (ilogb(M) + 1));
}
__global__ __launch_bounds__(256) void k_fix_absmax(const double *__restrict__ w, const double *__restrict__ uray, int64_t R,
                                                    unsigned long long *__restrict__ out) {
    double m = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
        if (uray[r * 8 + 7] == 0.0) continue;            // rays outside the grid have no segments
        double v = __builtin_fabs(w[r] * uray[r * 8 + 6]);
        if (!(v == v)) v = __builtin_inf();
        m = fmax(m, v);
    }
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    // one atomic per workgroup (4 000 waves on one address made this pass 50 us)
    __shared__ double wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmax(fmax(wm[0], wm[1]), fmax(wm[2], wm[3]));
        if (m > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(m));      // (positive doubles order as integers)
    }
}
// The largest number of terms any node's sum receives from the planned rays (once per plan, when the deterministic mode first needs
// it): the fixed-point kernel itself in COUNTING mode (fixbits < 0: every contribution is the integer 1) leaves that count per node in
// the integer grid; this pass takes the maximum and re-zeroes the grid.  It bounds the sums far below the plan's segment-count bound,
// so the scale keeps that many more bits (2^-48 instead of 2^-38 of the largest contribution at the bench shape).
__global__ __launch_bounds__(256) void k_fix_nodemax(unsigned long long *__restrict__ F, int64_t n, unsigned long long *__restrict__ out) {
    unsigned long long m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long v = F[i];
        if (v) {
            m = max(m, v);
            F[i] = 0ull;                                      // (the integer grid is zero between launches)
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)m, off), hi = (unsigned)__shfl_xor((int)(unsigned)(m >> 32), off);
        m = max(m, ((unsigned long long)hi << 32) | lo);
    }
    __shared__ unsigned long long wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, max(max(wm[0], wm[1]), max(wm[2], wm[3])));
}
template <typename AT>
struct FixConvert {                // grad += F / scale; F = 0 (the integer grid is zero between launches)
    unsigned long long *__restrict__ F;
    AT *__restrict__ grad;
    const unsigned long long *__restrict__ fixmax;
    int fixbits;
    double inv;
    __device__ __forceinline__ void begin() { inv = 1.0 / fix_scale(*fixmax, fixbits); }      // a power of two (or NaN)
    __device__ __forceinline__ void operator()(int64_t i) const {
        const long long q = (long long)F[i];
        if (q != 0) {
            grad[i] = (AT)((double)grad[i] + (double)q * inv);
            F[i] = 0ull;
        }
    }
};
// the same over the tiles the planned rays reach (iono_cubic_kernels.h: LmTile; i = tile ordinal * LMT_NODES + node of the tile): a third
// of the bench grid instead of all of it
template <typename AT>
struct FixConvertTiles {
    unsigned long long *__restrict__ F;
    AT *__restrict__ grad;
    const unsigned long long *__restrict__ fixmax;
    int fixbits;
    const LmTile *__restrict__ tiles;
    LmTileGeom tg;
    double inv;
    __device__ __forceinline__ void begin() { inv = 1.0 / fix_scale(*fixmax, fixbits); }
    __device__ __forceinline__ void operator()(int64_t it) const {
        int i, j, k, a, b, cc;
        if (!lm_tile_node(tg, tiles[it / LMT_NODES].id, (int)(it % LMT_NODES), i, j, k, a, b, cc)) return;
        const int64_t idx = ((int64_t)i * tg.ny + j) * tg.nz + k;
        const long long q = (long long)F[idx];
        if (q != 0) {
            grad[idx] = (AT)((double)grad[idx] + (double)q * inv);
            F[idx] = 0ull;
        }
    }
};

__device__ __forceinline__ double dpp_shr1(double v) {       // value of the previous lane of the 16-lane row (0 for its first lane)
    // (bound_ctrl: the row's first lane reads 0 without an initialised destination -- eight v_mov fewer per pass)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// PNF > 0: transpose of the PHASE observable (inversion/iterative_newton.py:86-127) for PNF frequencies per pass: the ray weight
// becomes a per-sample factor  sum_l wrf[r][l] / (2 n_p,l sqrt(1 - ne_k / n_p,l))  of the electron density ne_k interpolated at
// the sample (gathered from the grid exactly as the forward kernel does), wray = wrf with row stride ldw.
// FIX: deterministic fixed-point accumulation (above); G is then the grid of 64-bit integers.
template <typename AT, int PNF = 0, typename GT = double, int SEGL = BIN_SEG, bool FIX = false>
__global__ __launch_bounds__(BIN_THREADS) void k_adjoint_binned(GridView g, const double *__restrict__ uray, const uint2 *__restrict__ entries,
                                                        const BinUnit *__restrict__ units, const double *__restrict__ wray,
                                                        int Ns, const double *__restrict__ unitw, AT *__restrict__ G,
                                                        PhaseFreqs pf = PhaseFreqs{}, int ldw = 0,
                                                        const unsigned long long *__restrict__ fixmax = nullptr, int fixbits = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *wlds = (double *)smem;                                   // [Ns] quadrature weights
    // the box image is float64 whatever the accumulation type of the result: ds_add_f32 measured FIVE times slower than
    // ds_add_f64 on gfx950 (1.71 against 0.34 ms for this kernel); a float32 result is rounded once per box at the flush
    double *tile = wlds + ((Ns + 1) & ~1);                           // [BIN_BX * BIN_BY][BIN_BZP]
    for (int t = threadIdx.x; t < Ns; t += blockDim.x) wlds[t] = unitw[t];
    for (int t = threadIdx.x; t < BIN_TILE; t += blockDim.x) tile[t] = 0.0;
    const BinUnit un = units[blockIdx.x];
    lds_barrier();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    constexpr int PASS = BIN_THREADS / SEGL;                                     // segments per pass of the workgroup
    const int sub = lane & (SEGL - 1), grp = wid * (64 / SEGL) + lane / SEGL;
    // The segment list and the ray records are dependent gathers (entry -> ray id -> ray record): software-pipelined two
    // deep so that a pass computes while the next pass's ray records and the one after's entries are in flight.
    struct RayRec {
        double2 ux, uy, uz;          // (f0, df) per axis in grid coordinates
        double uh;                   // arc length per sample (the record's "valid" is not needed: only valid rays have segments;
                                     //  8 bytes instead of 16 through the texture path, which returns 64 B per clock to a wave)
        double w[PNF > 0 ? PNF : 1]; // ray weight (PNF: one per frequency)
    };
    const GT *b00 = (const GT *)g.M, *b01 = b00 + g.nz, *b10 = b00 + (size_t)g.ny * g.nz, *b11 = b10 + g.nz;
    // (the entry array is padded by BIN_ENTRY_PAD zero entries: unconditional loads; a pass beyond the unit is masked by its count)
    auto load_entry = [&](int e) { return entries[e]; };
    auto load_ray = [&](const uint2 en) {
        const double2 *up = (const double2 *)(uray + (size_t)en.x * 8);
        RayRec r;
        r.ux = up[0], r.uy = up[1], r.uz = up[2], r.uh = ((const double *)up)[6];
        if (PNF > 0) {
#pragma unroll
            for (int l = 0; l < (PNF > 0 ? PNF : 1); ++l) r.w[l] = wray[(size_t)en.x * ldw + l];
        } else {
            r.w[0] = wray[en.x];
        }
        return r;
    };
    int e = un.e_lo + grp;
    uint2 en0 = load_entry(e), en1 = load_entry(e + PASS);
    RayRec r0 = load_ray(en0);
    // (round 4: the instruction diet of this loop -- 105 -> ~85 vector instructions per pass: clamps without the NaN canonicalisation
    //  fmin() implies, node and image indices by 24-bit multiply-adds instead of 64-bit ones (nx ny <= 2^24 is a condition of the plan),
    //  (1 - t) weights as differences, two passes per loop body so that the software pipeline's registers rename instead of moving)
    const double lim_x = (double)(g.nx - 2), lim_y = (double)(g.ny - 2), lim_z = (double)(g.nz - 2);
    // a * b + c on 24-bit operands, ONE full-rate instruction (the compiler turns small-integer index arithmetic into v_mad_u64_u32 +
    // v_mul_lo_u32: quarter rate); b is wave-uniform
    auto mad24 = [](unsigned a, unsigned b, unsigned c) {
        unsigned o;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(o) : "v"(a), "s"(b), "v"(c));
        return o;
    };
    // the box image's four columns at one level, as LDS atomics the compiler cannot merge with the global ones of the other branch
    // (it formed ONE flat atomic out of the two branches' last atomics: flat atomics are slower than either kind)
    auto tile_add4 = [](unsigned a, double v00, double v01, double v10, double v11) {
        asm volatile("ds_add_f64 %0, %1\n\tds_add_f64 %0, %2 offset:%5\n\tds_add_f64 %0, %3 offset:%6\n\tds_add_f64 %0, %4 offset:%7"
                     ::"v"(a), "v"(v00), "v"(v01), "v"(v10), "v"(v11), "n"(BIN_BZP * 8), "n"(BIN_BY * BIN_BZP * 8), "n"((BIN_BY + 1) * BIN_BZP * 8) : "memory");
    };
    auto tile_add4_up = [](unsigned a, double v00, double v01, double v10, double v11) {
        asm volatile("ds_add_f64 %0, %1 offset:8\n\tds_add_f64 %0, %2 offset:%5\n\tds_add_f64 %0, %3 offset:%6\n\tds_add_f64 %0, %4 offset:%7"
                     ::"v"(a), "v"(v00), "v"(v01), "v"(v10), "v"(v11), "n"(BIN_BZP * 8 + 8), "n"(BIN_BY * BIN_BZP * 8 + 8), "n"((BIN_BY + 1) * BIN_BZP * 8 + 8) : "memory");
    };
    // FIX: the same eight words as integers (the values are int64 bit patterns carried in double registers)
    auto tile_add4_fix = [](unsigned a, double v00, double v01, double v10, double v11) {
        asm volatile("ds_add_u64 %0, %1\n\tds_add_u64 %0, %2 offset:%5\n\tds_add_u64 %0, %3 offset:%6\n\tds_add_u64 %0, %4 offset:%7"
                     ::"v"(a), "v"(v00), "v"(v01), "v"(v10), "v"(v11), "n"(BIN_BZP * 8), "n"(BIN_BY * BIN_BZP * 8), "n"((BIN_BY + 1) * BIN_BZP * 8) : "memory");
    };
    auto tile_add4_up_fix = [](unsigned a, double v00, double v01, double v10, double v11) {
        asm volatile("ds_add_u64 %0, %1 offset:8\n\tds_add_u64 %0, %2 offset:%5\n\tds_add_u64 %0, %3 offset:%6\n\tds_add_u64 %0, %4 offset:%7"
                     ::"v"(a), "v"(v00), "v"(v01), "v"(v10), "v"(v11), "n"(BIN_BZP * 8 + 8), "n"(BIN_BY * BIN_BZP * 8 + 8), "n"((BIN_BY + 1) * BIN_BZP * 8 + 8) : "memory");
    };
    const bool counting = FIX && fixbits < 0;                       // every contribution counts 1: the terms per node (k_fix_nodemax)
    const double fscale = FIX && !counting ? fix_scale(*fixmax, fixbits) : 1.0;
    auto qf = [&](double v) {
        const double t = fma(v, fscale, FIX_MAGIC);
        return __longlong_as_double(__double_as_longlong(t) - __double_as_longlong(FIX_MAGIC));
    };
    auto global_add4_fix = [&](int i, int j, int kk, double v00, double v01, double v10, double v11) {
        unsigned long long *p = (unsigned long long *)G + ((size_t)i * g.ny + j) * g.nz + kk;
        atomicAdd(p, (unsigned long long)__double_as_longlong(v00));
        atomicAdd(p + g.nz, (unsigned long long)__double_as_longlong(v01));
        atomicAdd(p + (size_t)g.ny * g.nz, (unsigned long long)__double_as_longlong(v10));
        atomicAdd(p + (size_t)g.ny * g.nz + g.nz, (unsigned long long)__double_as_longlong(v11));
    };
    const unsigned tile_base = (unsigned)(size_t)tile;
    auto clampf = [](double v, double lim) {
        double o;
        asm("v_min_f64 %0, %1, %2" : "=v"(o) : "v"(v), "v"(lim));
        return o;
    };
    // one pass: the 256 / SEGL segments en0 (ray records r0) of this workgroup, entry index e
    auto pass = [&](const uint2 en0, const RayRec &r0, const int e) {
        const int cnt = e < un.e_hi ? (int)((en0.y >> 16) & 0xffu) : 0, k = min((int)(en0.y & 0xffffu) + sub, Ns - 1);
        // every lane computes (no divergence before the lane exchange below); inactive lanes carry zero weights
        const double kd = (double)k;
        const double fx = fma(kd, r0.ux.y, r0.ux.x), fy = fma(kd, r0.uy.y, r0.uy.x), fz = fma(kd, r0.uz.y, r0.uz.x);
        double wr = r0.w[0];
        if (PNF > 0 && sub < cnt) {  // (only real samples gather: a padding entry may belong to a ray outside the grid)
            const double ne = trilinear_u<GT>(b00, b01, b10, b11, g.ny, g.nz, fx, fy, fz);
            wr = 0.0;
#pragma unroll
            for (int l = 0; l < (PNF > 0 ? PNF : 1); ++l) wr += r0.w[l] * (0.5 * pf.inv_np[l]) * rsqrt(1.0 - ne * pf.inv_np[l]);
        }
        const double c = sub < cnt ? (FIX && counting ? 1.0 : wr * r0.uh * wlds[k]) : 0.0;
        const bool active = c != 0.0;
        const double fi = clampf(__builtin_floor(__builtin_fabs(fx)), lim_x), fj = clampf(__builtin_floor(__builtin_fabs(fy)), lim_y),
                     fk = clampf(__builtin_floor(__builtin_fabs(fz)), lim_z);
        double w00, w01, w10, w11, l00, l01, l10, l11, u00, u01, u10, u11;
        {
            // trilinear: weights (1 - t, t) per axis as x (1 - t) = x - x t: one multiply and one subtraction per split
            // (the tricubic transpose has its own kernel, four derivative channels per traversal: k_adjoint_binned_lm4)
            const double tx = fx - fi, ty = fy - fj, tz = fz - fk;
            const double w1 = c * tx, w0 = c - w1;
            w01 = w0 * ty, w00 = w0 - w01, w11 = w1 * ty, w10 = w1 - w11;
            u00 = w00 * tz, u01 = w01 * tz, u10 = w10 * tz, u11 = w11 * tz;
            l00 = w00 - u00, l01 = w01 - u01, l10 = w10 - u10, l11 = w11 - u11;
        }
        if (FIX && counting) l00 = l01 = l10 = l11 = u00 = u01 = u10 = u11 = c;      // (a handed-over pair then counts 2)
        const int i = (int)fi, j = (int)fj, kz = (int)fk;
        // Consecutive samples of a ray mostly sit in consecutive z cells of the same (i, j) column: the upper-level
        // contributions of lane s then hit the very nodes of lane s + 1's lower level.  Pass them one lane up inside the
        // 16-lane DPP row (row_shr:1) and let the receiver add them to its own before its LDS atomics: 4 + (rarely 4)
        // instead of 8 LDS atomics per sample -- the kernel is bound by LDS atomic throughput.  (The test is on the NODE, not
        // on the ray: with narrower segments lane s and lane s + 1 may belong to different rays of the unit; a contribution
        // that meets the cell below its own is merged all the same, into the same LDS word it would have gone to.)
        const int lin = active ? (int)mad24(mad24((unsigned)i, (unsigned)g.ny, (unsigned)j), (unsigned)g.nz, (unsigned)kz) : -7;
        const int prev = __builtin_amdgcn_update_dpp(-9, lin, 0x111, 0xf, 0xf, false);          // row_shr:1 (row lane 0 keeps -9)
        const bool accept = active && prev + 1 == lin;
        const double p00 = dpp_shr1(u00), p01 = dpp_shr1(u01), p10 = dpp_shr1(u10), p11 = dpp_shr1(u11);
        if (accept) l00 += p00, l01 += p01, l10 += p10, l11 += p11;
        const int taken = __builtin_amdgcn_update_dpp(0, (int)accept, 0x101, 0xf, 0xf, false);  // row_shl:1: did lane s + 1 take mine?
        const bool upper = active && !taken;
        if (FIX) {       // (after the lane exchange: what is handed over and merged is decided by the plan, not by the launch)
            l00 = qf(l00), l01 = qf(l01), l10 = qf(l10), l11 = qf(l11);
            u00 = qf(u00), u01 = qf(u01), u10 = qf(u10), u11 = qf(u11);
        }
        if (active) {
            const unsigned a = (unsigned)(i - un.x0), b = (unsigned)(j - un.y0), m = (unsigned)(kz - un.z0);
            if ((a < (unsigned)(BIN_BX - 1)) & (b < (unsigned)(BIN_BY - 1)) & (m < (unsigned)(BIN_BZ - 1))) {
                const unsigned t = tile_base + mad24(mad24(a, BIN_BY, b), BIN_BZP, m) * 8u;
                if (FIX) {
                    tile_add4_fix(t, l00, l01, l10, l11);
                    if (upper) tile_add4_up_fix(t, u00, u01, u10, u11);
                } else {
                    tile_add4(t, l00, l01, l10, l11);
                    if (upper) tile_add4_up(t, u00, u01, u10, u11);
                }
            } else if (FIX) {
                global_add4_fix(i, j, kz, l00, l01, l10, l11);
                if (upper) global_add4_fix(i, j, kz + 1, u00, u01, u10, u11);
            } else {
                global_add4<AT>(G, i, j, kz, g.ny, g.nz, l00, l01, l10, l11);
                if (upper) global_add4<AT>(G, i, j, kz + 1, g.ny, g.nz, u00, u01, u10, u11);
            }
        }
    };
    // two passes per loop body (the DPP exchange is a convergent operation: the compiler does not unroll such a loop by itself), so
    // that the registers of the software pipeline -- entries three passes ahead, ray records one -- rename instead of moving
    for (; e < un.e_hi; e += 2 * PASS) {
        const uint2 en2 = load_entry(e + 2 * PASS);
        const RayRec r1 = load_ray(en1);
        pass(en0, r0, e);
        const uint2 en3 = load_entry(e + 3 * PASS);
        r0 = load_ray(en2);
        pass(en1, r1, e + PASS);
        en0 = en2, en1 = en3;
    }
    lds_barrier();
    // ---- flush the box once: 16 lanes per (x, y) column, consecutive z -> one 128-B run of global atomics per column ----
    static_assert((BIN_BZP & (BIN_BZP - 1)) == 0 && BIN_BZP >= BIN_BZ, "the image keeps a power-of-two number of words per column");
    const int m = threadIdx.x & (BIN_BZP - 1);
    for (int col = threadIdx.x / BIN_BZP; col < BIN_BX * BIN_BY; col += BIN_THREADS / BIN_BZP) {
        const double v = tile[col * BIN_BZP + m];
        if (FIX ? __double_as_longlong(v) != 0 : v != 0.0) {
            const int a = col / BIN_BY, b = col - a * BIN_BY;
            const int gi = un.x0 + a, gj = un.y0 + b, gk = un.z0 + m;
            if (gi >= 0 && gi < g.nx && gj >= 0 && gj < g.ny && gk >= 0 && gk < g.nz) {
                if (FIX) atomicAdd((unsigned long long *)G + ((size_t)gi * g.ny + gj) * g.nz + gk, (unsigned long long)__double_as_longlong(v));
                else atomicAdd(G + ((size_t)gi * g.ny + gj) * g.nz + gk, (AT)v);
            }
        }
    }
}

// ---- tricubic transpose: FOUR derivative channels per traversal ---------------------------------------------------------------------
// The Lekien-Marsden transpose scatters, per sample, Hermite weights X_p[cx] Y_q[cy] Z_r[cz] into channel (p, q, r) of the 8 corner
// nodes (iono_cubic_kernels.h).  One k_adjoint_binned<.., CUBIC> pass per channel repeats everything that is NOT an LDS atomic eight
// times -- entry and ray-record gathers through the texture path (0.6 busy), positions, clamps, the unit's zeroing and flush -- and
// the LDS atomics are only half of a pass (profiles/r04_adjoint_pmc.json).  Here a workgroup of LM4_THREADS threads keeps the box images of
// the four channels (p, q) in {0, 1}^2 of ONE z kind r (4 x 28.8 KB) and every lane scatters its sample into all four: positions,
// records and the x / y Hermite sets once per sample, the z pair once, 4 x (4 + 4) LDS atomics with the same DPP hand-over of the
// upper level.  Two launches (r = 0, 1) replace eight.
#ifndef LM4_THREADS
#define LM4_THREADS 1024
#endif
#ifndef LM_NCH
#define LM_NCH 4      // channels per traversal.  A/B build -DLM_NCH=8 -DBIN_SZ=7 -DBIN_BZP=8: all eight in ONE traversal over half-height boxes
#endif                //   (eight 14.4 KB images): measured slower, profiles/r06_tricubic_transpose_one_traversal_rejected.json
#define LM4_LDS_BYTES(Ns) (sizeof(double) * ((((size_t)(Ns) + 1) & ~(size_t)1) + LM_NCH * (size_t)BIN_TILE))
// FIX: deterministic fixed-point accumulation (k_adjoint_binned<.., FIX>): the images and G8 hold 64-bit integers, which the z fold
// turns into float64 as it reads them (k_lm_fold_z_tiles<FIX>).
template <int SEGL, bool FIX = false>
__global__ __launch_bounds__(LM4_THREADS) void k_adjoint_binned_lm4(GridView g, const double *__restrict__ uray, const uint2 *__restrict__ entries,
                                                                    const BinUnit *__restrict__ units, const double *__restrict__ wray, int Ns,
                                                                    const double *__restrict__ unitw, double *__restrict__ G8, int64_t nstride,
                                                                    int rbit, const unsigned long long *__restrict__ fixmax = nullptr,
                                                                    int fixbits = 0, int n_units = 0, int *__restrict__ next_unit = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *wlds = (double *)smem;                                   // [Ns] quadrature weights
    double *tile = wlds + ((Ns + 1) & ~1);                           // [4][BIN_BX * BIN_BY][BIN_BZP]: channel c = p + 2 q
    for (int t = threadIdx.x; t < Ns; t += LM4_THREADS) wlds[t] = unitw[t];
    for (int t = threadIdx.x; t < LM_NCH * BIN_TILE; t += LM4_THREADS) tile[t] = 0.0;
    // PERSISTENT workgroups (the four images leave room for one per CU): a workgroup starts on unit blockIdx.x and then takes the next
    // unit nobody has (`next_unit`, zeroed by the host: the list is sorted largest first, so this is longest-processing-time-first);
    // the images are re-zeroed word by word as they are flushed, the weights are staged once -- a unit no longer pays a workgroup
    // launch, a 115-KB zeroing pass and a weights load, and the tail of the launch is one small unit per CU.
    __shared__ int ui_next;
    BinUnit un = units[blockIdx.x];
    lds_barrier();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    constexpr int PASS = LM4_THREADS / SEGL;
    const int sub = lane & (SEGL - 1), grp = wid * (64 / SEGL) + lane / SEGL;
    struct RayRec {
        double2 ux, uy, uz;
        double uh, w;
    };
    auto load_ray = [&](const uint2 en) {
        const double2 *up = (const double2 *)(uray + (size_t)en.x * 8);
        RayRec r;
        r.ux = up[0], r.uy = up[1], r.uz = up[2], r.uh = ((const double *)up)[6], r.w = wray[en.x];
        return r;
    };
    const double lim_x = (double)(g.nx - 2), lim_y = (double)(g.ny - 2), lim_z = (double)(g.nz - 2);
    auto clampf = [](double v, double lim) {
        double o;
        asm("v_min_f64 %0, %1, %2" : "=v"(o) : "v"(v), "v"(lim));
        return o;
    };
    auto mad24 = [](unsigned a, unsigned b, unsigned c) {
        unsigned o;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(o) : "v"(a), "s"(b), "v"(c));
        return o;
    };
    auto tile_add4 = [](unsigned a, double v00, double v01, double v10, double v11) {
        asm volatile("ds_add_f64 %0, %1\n\tds_add_f64 %0, %2 offset:%5\n\tds_add_f64 %0, %3 offset:%6\n\tds_add_f64 %0, %4 offset:%7"
                     ::"v"(a), "v"(v00), "v"(v01), "v"(v10), "v"(v11), "n"(BIN_BZP * 8), "n"(BIN_BY * BIN_BZP * 8), "n"((BIN_BY + 1) * BIN_BZP * 8) : "memory");
    };
    auto tile_add4_up = [](unsigned a, double v00, double v01, double v10, double v11) {
        asm volatile("ds_add_f64 %0, %1 offset:8\n\tds_add_f64 %0, %2 offset:%5\n\tds_add_f64 %0, %3 offset:%6\n\tds_add_f64 %0, %4 offset:%7"
                     ::"v"(a), "v"(v00), "v"(v01), "v"(v10), "v"(v11), "n"(BIN_BZP * 8 + 8), "n"(BIN_BY * BIN_BZP * 8 + 8), "n"((BIN_BY + 1) * BIN_BZP * 8 + 8) : "memory");
    };
    auto tile_add4_fix = [](unsigned a, double v00, double v01, double v10, double v11) {
        asm volatile("ds_add_u64 %0, %1\n\tds_add_u64 %0, %2 offset:%5\n\tds_add_u64 %0, %3 offset:%6\n\tds_add_u64 %0, %4 offset:%7"
                     ::"v"(a), "v"(v00), "v"(v01), "v"(v10), "v"(v11), "n"(BIN_BZP * 8), "n"(BIN_BY * BIN_BZP * 8), "n"((BIN_BY + 1) * BIN_BZP * 8) : "memory");
    };
    auto tile_add4_up_fix = [](unsigned a, double v00, double v01, double v10, double v11) {
        asm volatile("ds_add_u64 %0, %1 offset:8\n\tds_add_u64 %0, %2 offset:%5\n\tds_add_u64 %0, %3 offset:%6\n\tds_add_u64 %0, %4 offset:%7"
                     ::"v"(a), "v"(v00), "v"(v01), "v"(v10), "v"(v11), "n"(BIN_BZP * 8 + 8), "n"(BIN_BY * BIN_BZP * 8 + 8), "n"((BIN_BY + 1) * BIN_BZP * 8 + 8) : "memory");
    };
    const double fscale = FIX ? fix_scale(*fixmax, fixbits) : 1.0;
    auto qf = [&](double v) {
        const double t = fma(v, fscale, FIX_MAGIC);
        return __longlong_as_double(__double_as_longlong(t) - __double_as_longlong(FIX_MAGIC));
    };
    auto global_add4_fix = [&](double *G, int i, int j, int kk, double v00, double v01, double v10, double v11) {
        unsigned long long *p = (unsigned long long *)G + ((size_t)i * g.ny + j) * g.nz + kk;
        atomicAdd(p, (unsigned long long)__double_as_longlong(v00));
        atomicAdd(p + g.nz, (unsigned long long)__double_as_longlong(v01));
        atomicAdd(p + (size_t)g.ny * g.nz, (unsigned long long)__double_as_longlong(v10));
        atomicAdd(p + (size_t)g.ny * g.nz + g.nz, (unsigned long long)__double_as_longlong(v11));
    };
    const unsigned tile_base = (unsigned)(size_t)tile;
    auto pass = [&](const uint2 en0, const RayRec &r0, const int e) {
        const int cnt = e < un.e_hi ? (int)((en0.y >> 16) & 0xffu) : 0, k = min((int)(en0.y & 0xffffu) + sub, Ns - 1);
        const double kd = (double)k;
        const double fx = fma(kd, r0.ux.y, r0.ux.x), fy = fma(kd, r0.uy.y, r0.uy.x), fz = fma(kd, r0.uz.y, r0.uz.x);
        const double c = sub < cnt ? r0.w * r0.uh * wlds[k] : 0.0;
        const bool active = c != 0.0;
        const double fi = clampf(__builtin_floor(__builtin_fabs(fx)), lim_x), fj = clampf(__builtin_floor(__builtin_fabs(fy)), lim_y),
                     fk = clampf(__builtin_floor(__builtin_fabs(fz)), lim_z);
        // Hermite value (bit 0) and slope (bit 1) weights of the two nodes per axis: axis_pair (iono_adjoint_kernels.h)
        double xv0, xv1, xs0, xs1, yv0, yv1, ys0, ys1, z0, z1, zb0 = 0.0, zb1 = 0.0;
        axis_pair(fx - fi, 0, true, xv0, xv1);
        axis_pair(fx - fi, 1, true, xs0, xs1);
        axis_pair(fy - fj, 0, true, yv0, yv1);
        axis_pair(fy - fj, 1, true, ys0, ys1);
        axis_pair(fz - fk, LM_NCH == 8 ? 0 : rbit, true, z0, z1);
        if (LM_NCH == 8) axis_pair(fz - fk, 1, true, zb0, zb1);      // (one traversal: both z kinds)
        const int i = (int)fi, j = (int)fj, kz = (int)fk;
        const int lin = active ? (int)mad24(mad24((unsigned)i, (unsigned)g.ny, (unsigned)j), (unsigned)g.nz, (unsigned)kz) : -7;
        const int prev = __builtin_amdgcn_update_dpp(-9, lin, 0x111, 0xf, 0xf, false);          // row_shr:1 (row lane 0 keeps -9)
        const bool accept = active && prev + 1 == lin;
        const int taken = __builtin_amdgcn_update_dpp(0, (int)accept, 0x101, 0xf, 0xf, false);  // row_shl:1: did lane s + 1 take mine?
        const bool upper = active && !taken;
        const unsigned a = (unsigned)(i - un.x0), b = (unsigned)(j - un.y0), m = (unsigned)(kz - un.z0);
        const bool inside = (a < (unsigned)(BIN_BX - 1)) & (b < (unsigned)(BIN_BY - 1)) & (m < (unsigned)(BIN_BZ - 1));
        const unsigned t0 = tile_base + mad24(mad24(a, BIN_BY, b), BIN_BZP, m) * 8u;
        const double cza0 = c * z0, cza1 = c * z1, czb0 = c * zb0, czb1 = c * zb1;
#pragma unroll
        for (int ch = 0; ch < LM_NCH; ++ch) {
            const double cz0 = (ch & 4) ? czb0 : cza0, cz1 = (ch & 4) ? czb1 : cza1;
            const double x0 = (ch & 1) ? xs0 : xv0, x1 = (ch & 1) ? xs1 : xv1, y0 = (ch & 2) ? ys0 : yv0, y1 = (ch & 2) ? ys1 : yv1;
            const double w00 = x0 * y0, w01 = x0 * y1, w10 = x1 * y0, w11 = x1 * y1;
            double l00 = w00 * cz0, l01 = w01 * cz0, l10 = w10 * cz0, l11 = w11 * cz0;
            double u00 = w00 * cz1, u01 = w01 * cz1, u10 = w10 * cz1, u11 = w11 * cz1;
            const double p00 = dpp_shr1(u00), p01 = dpp_shr1(u01), p10 = dpp_shr1(u10), p11 = dpp_shr1(u11);
            if (accept) l00 += p00, l01 += p01, l10 += p10, l11 += p11;
            if (FIX) {
                l00 = qf(l00), l01 = qf(l01), l10 = qf(l10), l11 = qf(l11);
                u00 = qf(u00), u01 = qf(u01), u10 = qf(u10), u11 = qf(u11);
            }
            if (active) {
                if (inside) {
                    const unsigned t = t0 + (unsigned)ch * (BIN_TILE * 8u);
                    if (FIX) {
                        tile_add4_fix(t, l00, l01, l10, l11);
                        if (upper) tile_add4_up_fix(t, u00, u01, u10, u11);
                    } else {
                        tile_add4(t, l00, l01, l10, l11);
                        if (upper) tile_add4_up(t, u00, u01, u10, u11);
                    }
                } else {
                    double *G = G8 + (size_t)(LM_NCH == 8 ? ch : 4 * rbit + ch) * nstride;
                    if (FIX) {
                        global_add4_fix(G, i, j, kz, l00, l01, l10, l11);
                        if (upper) global_add4_fix(G, i, j, kz + 1, u00, u01, u10, u11);
                    } else {
                        global_add4<double>(G, i, j, kz, g.ny, g.nz, l00, l01, l10, l11);
                        if (upper) global_add4<double>(G, i, j, kz + 1, g.ny, g.nz, u00, u01, u10, u11);
                    }
                }
            }
        }
    };
    const int mz = threadIdx.x & (BIN_BZP - 1);
    int e = un.e_lo + grp;
    uint2 en0 = entries[e], en1 = entries[e + PASS];
    RayRec r0 = load_ray(en0);
    for (int ui = blockIdx.x; ui < n_units;) {
        if (threadIdx.x == 0) ui_next = (int)gridDim.x + atomicAdd(next_unit, 1);
        for (; e < un.e_hi; e += 2 * PASS) {
            const uint2 en2 = entries[e + 2 * PASS];
            const RayRec r1 = load_ray(en1);
            pass(en0, r0, e);
            const uint2 en3 = entries[e + 3 * PASS];
            r0 = load_ray(en2);
            pass(en1, r1, e + PASS);
            en0 = en2, en1 = en3;
        }
        lds_barrier();
        // the next unit's record, first entries and first ray records are in flight while this unit's images are flushed
        const int x0 = un.x0, y0 = un.y0, z0 = un.z0;
        ui = ui_next;
        if (ui < n_units) {
            un = units[ui];
            e = un.e_lo + grp;
            en0 = entries[e], en1 = entries[e + PASS];
            r0 = load_ray(en0);
        }
        // ---- flush the four images: 16 lanes per (x, y) column, consecutive z -> one 128-B run of global atomics per column ----
        for (int cc = threadIdx.x / BIN_BZP; cc < LM_NCH * BIN_BX * BIN_BY; cc += LM4_THREADS / BIN_BZP) {
            const double v = tile[cc * BIN_BZP + mz];
            if (FIX ? __double_as_longlong(v) != 0 : v != 0.0) {
                tile[cc * BIN_BZP + mz] = 0.0;                       // (the next unit of this workgroup starts from a clean image)
                const int ch = cc / (BIN_BX * BIN_BY), col = cc - ch * (BIN_BX * BIN_BY);
                const int a = col / BIN_BY, b = col - a * BIN_BY;
                const int gi = x0 + a, gj = y0 + b, gk = z0 + mz;
                if (gi >= 0 && gi < g.nx && gj >= 0 && gj < g.ny && gk >= 0 && gk < g.nz) {
                    double *dst = G8 + (size_t)(LM_NCH == 8 ? ch : 4 * rbit + ch) * nstride + ((size_t)gi * g.ny + gj) * g.nz + gk;
                    if (FIX) atomicAdd((unsigned long long *)dst, (unsigned long long)__double_as_longlong(v));
                    else atomicAdd(dst, v);
                }
            }
        }
        lds_barrier();
    }
}

// (round 5: the node-stationary FORWARD on this plan -- k_forward_binned + k_forward_binned_finish, an A/B of north_star's wording measured
//  at 0.42 ms against the bundle kernel's 0.11 -- profiles/r02_ab_forward_binned.json -- is no longer built)

}  // namespace

#endif
