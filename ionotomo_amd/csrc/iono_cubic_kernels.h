// tricubic on the hot path (ideal-uniform grids): Lekien-Marsden derivative fields, fast forward, channel scatter + fold
#ifndef IONO_CUBIC_KERNELS_H
#define IONO_CUBIC_KERNELS_H

namespace {

// ------------------------------------------------------------------------------------------------
// The tricubic of this build (notebooks/TricubicInterpolation.ipynb c0:138-1257: Lekien-Marsden with 4th-order
// central-difference derivative data) is the tensor product of 1-D cubic Hermite splines whose node slopes are
//   D f_i = (f[i-2] - 8 f[i-1] + 8 f[i+1] - f[i+2]) / 12      (cell units, uniform axis; 2 <= i <= n-3)
// (iono_device_common.h:cubic_axis spells out the equivalent 6-tap form).  Evaluating the 6 x 6 x 6 taps per sample
// costs 216 gathered values and 258 FMAs.  Lekien-Marsden's own formulation keeps, per NODE, the 8 derivative data
//   F[p + 2 q + 4 r] = Dx^p Dy^q Dz^r f,   p, q, r in {0, 1}
// (what the notebook calls bVec and caches per cell, c0:165-299) and evaluates in the cell's 8 corners:
//   f(x, y, z) = sum_{a,b,c in {0,1}} sum_{p,q,r} Hx[a][p](tx) Hy[b][q](ty) Hz[c][r](tz) F[p,q,r](i+a, j+b, k+c),
//   H[0][0] = 2t^3 - 3t^2 + 1,  H[1][0] = -2t^3 + 3t^2,  H[0][1] = t^3 - 2t^2 + t,  H[1][1] = t^3 - t^2
// -- 64 values (4 corner columns x one contiguous 128-B run of two nodes) and 84 FMAs per sample, the same
// arithmetic to rounding.  The fields are rebuilt (one kernel) whenever the grid values change; 8 doubles per node:
// 1 GiB at 256^3, which is what 288 GB of HBM is for.
//
// Transpose: each of the 8 channels is a trilinear-shaped scatter with the Hermite weights in place of (1-t, t), so
// the LDS-tiled back-projection kernel is reused per channel (field-major buffer G8[8][nx ny nz]); one fold kernel
// then applies the transposed difference stencils:  grad = sum_{pqr} (Dx^p Dy^q Dz^r)^T G8[pqr].
// ------------------------------------------------------------------------------------------------
#define LM_NF 8

// forward difference coefficient of tap offset d (-2..2), x 1/12
__device__ __forceinline__ double fd_coef(int d) {
    return d == -2 ? 1.0 / 12.0 : d == -1 ? -8.0 / 12.0 : d == 1 ? 8.0 / 12.0 : d == 2 ? -1.0 / 12.0 : 0.0;
}

// F8[node][8] from the stored values, axis by axis (the stencils are separable): Z = (f, Dz f), then Dy, then Dx, lanes along z,
// instead of the 125 loads of a direct 5 x 5 x 5 evaluation.  (Round 5: Z is formed on the fly inside the y / x kernel -- its five z
// taps are the neighbouring lanes' own loads, L1 hits -- instead of by a pass of its own through a 16-byte-per-node array.)  Nodes within 2 of a face have no slope along that axis (no valid sample
// ever weighs them: tricubic samples live in g[2] <= x <= g[n-3]): 0.
// `xrange` (round 5): the node lines the current forward plan's windows hold (k_lm_touch_lines) -- Z is then formed only where the
// restricted y / x pass reads it: on the lines within two of such a line in y, two planes beyond its range in x.
#define LM_XSEG 8
// z stride of F8 in nodes.  (Padding it to nz + 1 -- at 256^3 the column and plane strides, 16 KB and 4 MB, are powers of
// two and TCP_READ_TAGCONFLICT_STALL is 15 % of the forward's cycles -- measured 8 % SLOWER: 2.07 vs 1.92 ms.)
#ifndef LM_PAD_J
#define LM_PAD_J 0
#define LM_PAD_I 0
#endif
#define LM_NZP(nz) ((nz) + LM_PAD_J)
#define LM_SI(ny, nz) ((int64_t)(ny) * LM_NZP(nz) + LM_PAD_I)     // x-plane stride in nodes
// passes y and x in one kernel: a thread owns a (j, k) line of nodes and marches along i through one of LM_XSEG stretches, keeping
// Y4 = (f, Dy f, Dz f, Dy Dz f) of the five planes i - 2 .. i + 2 in registers (each formed from the five j-neighbours of Z = (f, Dz f):
// rows 4 KB apart, L1 / L2 hits), so that every Z is read from memory once and no Y4 array exists.  (Three separate passes took
// 0.09 + 0.26 + 0.68 ms at 256^3 -- the x pass re-read five planes 2 MB apart, beyond the L2.)  Lanes run along k: every load and
// store of a wave is one contiguous run.  Output: F8[node][p + 2 (q + 2 r)] node-major (k_forward_straight_lm), or -- PAIRS -- the
// same four (value, Dx value) pairs PAIR-major, FP[t = q + 2 r][node] (double2, node = the grid's own linear index, npad nodes per
// pair array): the layout the bundle-stationary forward stages from (k_forward_bundle_lm).
// `xrange` (round 5, PAIRS only): per (j, k) line the planes [lo, hi] the current forward plan's windows hold (k_lm_touch_lines) -- only
// those nodes are written: an inversion iteration rebuilds the fields its rays read (a third of the bench grid), not 1 GiB.
template <bool PAIRS, typename GT>
__global__ __launch_bounds__(256) void k_lm_fields_yx(const GT *__restrict__ M, double *__restrict__ F8, int nx, int ny, int nz, int64_t npad,
                                                      const int2 *__restrict__ xrange) {
    const int64_t sx = (int64_t)ny * nz, lines = sx * LM_XSEG;
    const int seg_len = (nx + LM_XSEG - 1) / LM_XSEG;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < lines; t += (int64_t)gridDim.x * blockDim.x) {
        const int seg = (int)(t / sx);
        const int64_t jk = t - (int64_t)seg * sx;
        const int j = (int)(jk / nz);
        int i0 = seg * seg_len, i1 = min(i0 + seg_len, nx);
        if (xrange) {
            const int2 xr = xrange[jk];
            if (xr.y < 0) continue;                                // no window of the plan holds this line
            i0 = max(i0, xr.x), i1 = min(i1, xr.y + 1);
            if (i0 >= i1) continue;
        }
        const bool yslope = j >= 2 && j <= ny - 3;
        const int kk = (int)(jk - (int64_t)j * nz);
        const bool zslope = kk >= 2 && kk <= nz - 3;
        auto zpair = [&](int64_t node) {                 // Z = (f, Dz f) of a node: 0 slope within two of a z face
            const GT *row = M + node;
            double dz = 0.0;
            if (zslope) dz = fd_coef(-2) * (double)row[-2] + fd_coef(-1) * (double)row[-1] + fd_coef(1) * (double)row[1] + fd_coef(2) * (double)row[2];
            return make_double2((double)row[0], dz);
        };
        double2 a[5], b[5];                      // a = (f, Dy f), b = (Dz f, Dy Dz f) of planes io - 2 .. io + 2 (slot 4 = the newest)
#pragma unroll
        for (int q = 0; q < 5; ++q) a[q] = b[q] = make_double2(0.0, 0.0);
        for (int ii = i0 - 2; ii < i1 + 2; ++ii) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = a[q + 1], b[q] = b[q + 1];
            a[4] = b[4] = make_double2(0.0, 0.0);
            if (ii >= 0 && ii < nx) {
                const int64_t node = (int64_t)ii * sx + jk;
                const double2 z0 = zpair(node);
                double dy0 = 0.0, dy1 = 0.0;
                if (yslope) {
#pragma unroll
                    for (int db = -2; db <= 2; ++db) {
                        if (db == 0) continue;
                        const double2 z = zpair(node + (int64_t)db * nz);
                        dy0 += fd_coef(db) * z.x, dy1 += fd_coef(db) * z.y;
                    }
                }
                a[4] = make_double2(z0.x, dy0), b[4] = make_double2(z0.y, dy1);
            }
            const int io = ii - 2;
            if (io < i0) continue;
            double2 d0 = make_double2(0.0, 0.0), d1 = d0;
            if (io >= 2 && io <= nx - 3) {
#pragma unroll
                for (int da = -2; da <= 2; ++da) {
                    if (da == 0) continue;
                    const double cf = fd_coef(da);
                    d0.x += cf * a[2 + da].x, d0.y += cf * a[2 + da].y, d1.x += cf * b[2 + da].x, d1.y += cf * b[2 + da].y;
                }
            }
            const double2 a0 = a[2], a1 = b[2];
            if (PAIRS) {
                double2 *o = (double2 *)F8 + (int64_t)io * sx + jk;
                o[0] = make_double2(a0.x, d0.x);
                o[npad] = make_double2(a0.y, d0.y);
                o[2 * npad] = make_double2(a1.x, d1.x);
                o[3 * npad] = make_double2(a1.y, d1.y);
            } else {
                // index p + 2 (q + 2 r): (value, Dx value) for (q, r) = 00, 10, 01, 11; padded z stride in the output
                double2 *o = (double2 *)(F8 + (io * LM_SI(ny, nz) + (int64_t)j * LM_NZP(nz) + (jk - (int64_t)j * nz)) * LM_NF);
                o[0] = make_double2(a0.x, d0.x);
                o[1] = make_double2(a0.y, d0.y);
                o[2] = make_double2(a1.x, d1.x);
                o[3] = make_double2(a1.y, d1.y);
            }
        }
    }
}

// The node lines a forward plan's tricubic windows hold: xrange[j nz + k] = (first plane, last plane) over every window that contains
// line (j, k) -- window w of chunk c holds nodes [imin, imin + wx) x [jmin, jmin + wy) x [kz0, kz0 + BL_LEV) when it fits (the plan
// restricts a rebuild only if EVERY window fits: a chunk that does not reads nodes the record does not bound).  One thread per
// (window, line of the window); xrange starts as 0xff bytes: (-1, -1) = no plane.
__global__ __launch_bounds__(256) void k_lm_touch_lines(const uint4 *__restrict__ win, int64_t nwin, int ny, int nz, int lev, int maxlines,
                                                        int2 *__restrict__ xrange) {
    const int64_t total = nwin * maxlines;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t wi = t / maxlines;
        const int li = (int)(t - wi * maxlines);
        const uint4 w = win[wi];
        const int wx = (int)(w.w & 255u), wy = (int)((w.w >> 8) & 255u);
        if (wx == 0) continue;                                   // (a bundle without a valid ray has empty records)
        const int dj = li / lev, l = li - dj * lev;
        if (dj >= wy) continue;
        const int j = (int)w.y + dj, k = (int)w.z + l;
        if (j >= ny || k >= nz) continue;
        int *xr = (int *)(xrange + (int64_t)j * nz + k);
        atomicMin((unsigned *)xr, w.x);                          // (all bits set = no plane yet: the array starts as 0xff bytes)
        atomicMax(xr + 1, (int)w.x + wx - 1);
    }
}

// grad[m] += sum_{pqr} sum_{offsets} ct_x^p ct_y^q ct_z^r G8[pqr][m + offset]: the transposed stencils, pass by pass:
// (z) H[node][p + 2 q] = G8[pq0] + Dz^T G8[pq1];  (y) K[node][p] = H[p, q=0] + Dy^T H[p, q=1];  (x) grad += K[0] + Dx^T K[1].
// The source node of a slope must itself be a valid slope node (2 <= i <= n-3 along that axis); transposed coefficient =
// fd_coef(-d).
// Layouts of the intermediates: H as two arrays of (p = 0, p = 1) pairs -- H0 = q 0, H1 = q 1 -- and K as two plain arrays K0 = p 0,
// K1 = p 1, so that the four NEIGHBOUR reads of the y and x passes (which want q = 1 / p = 1 only) are dense runs along z
// (interleaved records made every neighbour load half empty: 0.37 ms for the y pass, 2.2 TB/s).
__global__ __launch_bounds__(256) void k_lm_fold_z(const double *__restrict__ G8, double2 *__restrict__ H0, double2 *__restrict__ H1, int nx,
                                                   int ny, int nz) {
    const int64_t n = (int64_t)nx * ny * nz;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % nz);
        double h[4];
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) {
            const double *r0 = G8 + (int64_t)pq * n + idx, *r1 = G8 + (int64_t)(pq + 4) * n + idx;
            double s = r0[0];
#pragma unroll
            for (int dc = -2; dc <= 2; ++dc) {
                if (dc == 0) continue;
                const int sk = k + dc;
                if (sk >= 2 && sk <= nz - 3) s += fd_coef(-dc) * r1[dc];
            }
            h[pq] = s;
        }
        H0[idx] = make_double2(h[0], h[1]);
        H1[idx] = make_double2(h[2], h[3]);
    }
}
__global__ __launch_bounds__(256) void k_lm_fold_y(const double2 *__restrict__ H0, const double2 *__restrict__ H1, double *__restrict__ K0,
                                                   double *__restrict__ K1, int nx, int ny, int nz) {
    const int64_t n = (int64_t)nx * ny * nz;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)((idx / nz) % ny);
        double2 k2 = H0[idx];                                // (p = 0, p = 1) of q = 0
#pragma unroll
        for (int db = -2; db <= 2; ++db) {
            if (db == 0) continue;
            const int sj = j + db;
            if (sj >= 2 && sj <= ny - 3) {
                const double2 t = H1[idx + (int64_t)db * nz];      // q = 1
                k2.x += fd_coef(-db) * t.x, k2.y += fd_coef(-db) * t.y;
            }
        }
        K0[idx] = k2.x, K1[idx] = k2.y;
    }
}
template <typename AT>
__global__ __launch_bounds__(256) void k_lm_fold_x(const double *__restrict__ K0, const double *__restrict__ K1, AT *__restrict__ grad, int nx,
                                                   int ny, int nz) {
    const int64_t n = (int64_t)nx * ny * nz, sx = (int64_t)ny * nz;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx / sx);
        double acc = K0[idx];
#pragma unroll
        for (int da = -2; da <= 2; ++da) {
            if (da == 0) continue;
            const int si = i + da;
            if (si >= 2 && si <= nx - 3) acc += fd_coef(-da) * K1[idx + da * sx];
        }
        grad[idx] = (AT)((double)grad[idx] + acc);
    }
}

// ---- the same three passes over the TILES a back-projection plan's rays reach (round 4) ------------------------------------------------
// A plan knows which LMT_X x LMT_Y x LMT_Z-node tiles its samples add into (k_plan_touch: T).  The z pass can only give non-zero values on
// A1 = T dilated by one tile along z, the y pass on A2 = A1 dilated along y, the x pass on A3 = A2 dilated along x: each pass runs
// over the list of its OUTPUT tiles and reads its input only where the previous pass wrote it (flags: centre / lower / upper
// neighbour tile of the fold axis belong to the input set; G8 is zeroed on A1 only).  Everything outside is never written and
// never read, so the channel buffers need no full-grid memset and the folds stream the reached fraction of the grid.
// (tile shape, ms per planned tricubic back-projection at the bench shape, same box: 8 x 8 x 16 (round 4) 2.010 | 4 x 4 x 16 1.973 | 8 x 4 x 16 1.979 |
//  4 x 2 x 16 1.992 | 2 x 4 x 16 1.999 | 4 x 4 x 32 2.002 | 4 x 4 x 8 2.002 | 4 x 8 x 32 2.049 | 4 x 4 x 64 2.088 | 2 x 2 x 16 2.090: the finer the tiles
//  across the rays, the tighter the set the zeroing and the three folds stream; longer z runs bought nothing)
#ifndef LMT_X
#define LMT_X 4
#define LMT_Y 4
#define LMT_Z 16
#endif
#define LMT_NODES (LMT_X * LMT_Y * LMT_Z)
struct LmTile {
    int id;         // (ti * nty + tj) * ntz + tk
    int flags;      // bit 0: this tile, bit 1: the tile below it along the fold axis, bit 2: the tile above it -- belong to the INPUT set
};
struct LmTileGeom {
    int nx, ny, nz, nty, ntz;
};
// node (i, j, k) number q of tile t for thread-contiguous z runs: q = (a * LMT_Y + b) * LMT_Z + c
__device__ __forceinline__ bool lm_tile_node(const LmTileGeom &tg, int id, int q, int &i, int &j, int &k, int &a, int &b, int &cc) {
    const int tk = id % tg.ntz, tj = (id / tg.ntz) % tg.nty, ti = id / (tg.ntz * tg.nty);
    cc = q % LMT_Z, b = (q / LMT_Z) % LMT_Y, a = q / (LMT_Z * LMT_Y);
    i = ti * LMT_X + a, j = tj * LMT_Y + b, k = tk * LMT_Z + cc;
    return i < tg.nx && j < tg.ny && k < tg.nz;
}
// is the neighbour at offset dd (-2..2, != 0) along an axis, local coordinate l of extent L, inside the input set?
__device__ __forceinline__ bool lm_tile_has(int flags, int l, int dd, int L) {
    const int m = l + dd;
    return m < 0 ? (flags & 2) != 0 : (m >= L ? (flags & 4) != 0 : (flags & 1) != 0);
}
__global__ __launch_bounds__(256) void k_lm_zero_tiles(double *__restrict__ G8, const LmTile *__restrict__ tiles, LmTileGeom tg) {
    const int64_t n = (int64_t)tg.nx * tg.ny * tg.nz;
    const int id = tiles[blockIdx.x].id;
    for (int q = threadIdx.x; q < LMT_NODES / 2; q += blockDim.x) {       // (a thread = two consecutive levels: 16-byte stores)
        int i, j, k, a, b, cc;
        if (!lm_tile_node(tg, id, 2 * q, i, j, k, a, b, cc)) continue;
        const int64_t idx = ((int64_t)i * tg.ny + j) * tg.nz + k;
        const bool two = k + 1 < tg.nz;
#pragma unroll
        for (int f = 0; f < LM_NF; ++f) {
            if (two) *(double2 *)(G8 + (int64_t)f * n + idx) = make_double2(0.0, 0.0);
            else G8[(int64_t)f * n + idx] = 0.0;
        }
    }
}
// FIX (deterministic mode): G8 holds the 64-bit fixed-point integers of k_adjoint_binned_lm4<.., FIX>, scale fix_scale(*fixmax, fixbits)
// (iono_binned_kernels.h).
__device__ __forceinline__ double fix_scale(unsigned long long maxbits, int fixbits);
// (round 5: a thread folds TWO consecutive levels -- eight 16-byte loads per node instead of twenty 8-byte ones: the six source levels
//  k - 2 .. k + 3 of a channel are three aligned pairs; same sums in the same order as the one-level form: bit-identical)
template <bool FIX>
__global__ __launch_bounds__(256) void k_lm_fold_z_tiles(const double *__restrict__ G8, double2 *__restrict__ H0, double2 *__restrict__ H1,
                                                         const LmTile *__restrict__ tiles, LmTileGeom tg,
                                                         const unsigned long long *__restrict__ fixmax, int fixbits) {
    static_assert(LMT_Z % 2 == 0, "a thread of the z fold owns an even / odd pair of levels of its tile");
    const int64_t n = (int64_t)tg.nx * tg.ny * tg.nz;
    const LmTile t = tiles[blockIdx.x];
    double inv = 1.0;
    if (FIX) inv = 1.0 / fix_scale(*fixmax, fixbits);
    auto cv = [&](double x) { return FIX ? (double)__double_as_longlong(x) * inv : x; };
    for (int q = threadIdx.x; q < LMT_NODES / 2; q += blockDim.x) {
        int i, j, k, a, b, cc;
        if (!lm_tile_node(tg, t.id, 2 * q, i, j, k, a, b, cc)) continue;            // (cc even; the level k exists)
        const bool two = k + 1 < tg.nz;
        const int64_t idx = ((int64_t)i * tg.ny + j) * tg.nz + k;
        // the source levels k - 2 .. k + 3 that count: inside 2 .. nz - 3 and in a tile of the input set
        bool ok[6];
#pragma unroll
        for (int sl = 0; sl < 6; ++sl) {
            const int sk = k + sl - 2;
            ok[sl] = sk >= 2 && sk <= tg.nz - 3 && lm_tile_has(t.flags, cc, sl - 2, LMT_Z);
        }
        double h0[4], h1[4];
#pragma unroll
        for (int pq = 0; pq < 4; ++pq) {
            const double *r0 = G8 + (int64_t)pq * n + idx, *r1 = G8 + (int64_t)(pq + 4) * n + idx;
            double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, c0, c1 = 0.0;
            if (two) {
                const double2 p = *(const double2 *)r0;
                c0 = p.x, c1 = p.y;
            } else {
                c0 = r0[0];
            }
            if (ok[0] | ok[1]) {
                const double2 p = *(const double2 *)(r1 - 2);
                v[0] = p.x, v[1] = p.y;
            }
            if (two) {
                if (ok[2] | ok[3]) {
                    const double2 p = *(const double2 *)r1;
                    v[2] = p.x, v[3] = p.y;
                }
            } else if (ok[2]) {
                v[2] = r1[0];
            }
            if (ok[4] | ok[5]) {
                const double2 p = *(const double2 *)(r1 + 2);
                v[4] = p.x, v[5] = p.y;
            }
            // out[k] = r0[k] + sum_dc fd_coef(-dc) r1[k + dc], dc = -2, -1, 1, 2 in this order (the one-level form's)
            double s0 = cv(c0), s1 = cv(c1);
            if (ok[0]) s0 += fd_coef(2) * cv(v[0]);
            if (ok[1]) s0 += fd_coef(1) * cv(v[1]);
            if (ok[3]) s0 += fd_coef(-1) * cv(v[3]);
            if (ok[4]) s0 += fd_coef(-2) * cv(v[4]);
            if (ok[1]) s1 += fd_coef(2) * cv(v[1]);
            if (ok[2]) s1 += fd_coef(1) * cv(v[2]);
            if (ok[4]) s1 += fd_coef(-1) * cv(v[4]);
            if (ok[5]) s1 += fd_coef(-2) * cv(v[5]);
            h0[pq] = s0, h1[pq] = s1;
        }
        H0[idx] = make_double2(h0[0], h0[1]);
        H1[idx] = make_double2(h0[2], h0[3]);
        if (two) {
            H0[idx + 1] = make_double2(h1[0], h1[1]);
            H1[idx + 1] = make_double2(h1[2], h1[3]);
        }
    }
}
__global__ __launch_bounds__(256) void k_lm_fold_y_tiles(const double2 *__restrict__ H0, const double2 *__restrict__ H1, double *__restrict__ K0,
                                                         double *__restrict__ K1, const LmTile *__restrict__ tiles, LmTileGeom tg) {
    // (one level per thread: its loads are 16 bytes already; pairing levels for 16-byte stores measured 4 us slower)
    const LmTile t = tiles[blockIdx.x];
#pragma unroll
    for (int q = threadIdx.x; q < LMT_NODES; q += 256) {
        int i, j, k, a, b, cc;
        if (!lm_tile_node(tg, t.id, q, i, j, k, a, b, cc)) continue;
        const int64_t idx = ((int64_t)i * tg.ny + j) * tg.nz + k;
        double2 k2 = (t.flags & 1) ? H0[idx] : make_double2(0.0, 0.0);
#pragma unroll
        for (int db = -2; db <= 2; ++db) {
            if (db == 0) continue;
            const int sj = j + db;
            if (sj >= 2 && sj <= tg.ny - 3 && lm_tile_has(t.flags, b, db, LMT_Y)) {
                const double2 v = H1[idx + (int64_t)db * tg.nz];
                k2.x += fd_coef(-db) * v.x, k2.y += fd_coef(-db) * v.y;
            }
        }
        K0[idx] = k2.x, K1[idx] = k2.y;
    }
}
template <typename AT>
__global__ __launch_bounds__(256) void k_lm_fold_x_tiles(const double *__restrict__ K0, const double *__restrict__ K1, AT *__restrict__ grad,
                                                         const LmTile *__restrict__ tiles, LmTileGeom tg) {
    const int64_t sx = (int64_t)tg.ny * tg.nz;
    const LmTile t = tiles[blockIdx.x];
    for (int q = threadIdx.x; q < LMT_NODES / 2; q += blockDim.x) {       // (a thread = two consecutive levels: 16-byte loads)
        int i, j, k, a, b, cc;
        if (!lm_tile_node(tg, t.id, 2 * q, i, j, k, a, b, cc)) continue;
        const int64_t idx = ((int64_t)i * tg.ny + j) * tg.nz + k;
        const bool two = k + 1 < tg.nz;
        auto ld2 = [&](const double *p) { return two ? *(const double2 *)p : make_double2(p[0], 0.0); };
        double2 acc = make_double2(0.0, 0.0);
        if (t.flags & 1) acc = ld2(K0 + idx);
#pragma unroll
        for (int da = -2; da <= 2; ++da) {
            if (da == 0) continue;
            const int si = i + da;
            if (si >= 2 && si <= tg.nx - 3 && lm_tile_has(t.flags, a, da, LMT_X)) {
                const double2 v = ld2(K1 + idx + da * sx);
                acc.x += fd_coef(-da) * v.x, acc.y += fd_coef(-da) * v.y;
            }
        }
        grad[idx] = (AT)((double)grad[idx] + acc.x);
        if (two) grad[idx + 1] = (AT)((double)grad[idx + 1] + acc.y);
    }
}

struct Herm {
    double h0, h1, s0, s1;        // value weights of nodes 0 / 1, slope weights of nodes 0 / 1
};
__device__ __forceinline__ Herm hermite(double t) {
    const double t2 = t * t, t3 = t2 * t;
    Herm h;
    h.h1 = 3.0 * t2 - 2.0 * t3;
    h.h0 = 1.0 - h.h1;
    h.s0 = t3 - 2.0 * t2 + t;
    h.s1 = t3 - t2;
    return h;
}

// one sample: 4 corner columns x (2 nodes x 8 fields = 128 contiguous bytes)
__device__ __forceinline__ double tricubic_lm(const double *__restrict__ F8, int ny, int nz, double fx, double fy, double fz) {
    const double fi = __builtin_floor(__builtin_fabs(fx)), fj = __builtin_floor(__builtin_fabs(fy)),
                 fk = __builtin_floor(__builtin_fabs(fz));
    const Herm hx = hermite(fx - fi), hy = hermite(fy - fj), hz = hermite(fz - fk);
    const double nzp = (double)LM_NZP(nz);
    const double lin = __builtin_fma(fi, (double)LM_SI(ny, nz), __builtin_fma(fj, nzp, fk));
    const double2 *base = (const double2 *)(F8 + (size_t)lin * LM_NF);
    const size_t sj = (size_t)LM_NZP(nz) * (LM_NF / 2), si = (size_t)LM_SI(ny, nz) * (LM_NF / 2);      // strides in double2
    double ux[2][2];                          // [a][p]: after the z and y contractions
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        ux[a][0] = ux[a][1] = 0.0;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const double2 *col = base + a * si + b * sj;
            double2 n0[4], n1[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) n0[t] = col[t], n1[t] = col[4 + t];
#ifdef IONO_LM_ABL     // timing-only builds (WRONG results): 1 = no loads, 2 = every wave reads the same lines (no fills)
#if IONO_LM_ABL == 1
#pragma unroll
            for (int t = 0; t < 4; ++t) n0[t] = n1[t] = make_double2(lin, fx);
#elif IONO_LM_ABL == 2
            {
                const double2 *cc = (const double2 *)F8 + (threadIdx.x & 63) * 4 + (a * 2 + b) * 512;
                asm volatile("" : "+v"(cc) : "v"(lin));
#pragma unroll
                for (int t = 0; t < 4; ++t) n0[t] = cc[t], n1[t] = cc[4 + t];
            }
#endif
#endif
            // z: value fields (r = 0) sit in n?[0..1], their z-slopes (r = 1) in n?[2..3]; component .x = p 0, .y = p 1
            const double v00 = hz.h0 * n0[0].x + hz.h1 * n1[0].x + hz.s0 * n0[2].x + hz.s1 * n1[2].x;     // p 0, q 0
            const double v10 = hz.h0 * n0[0].y + hz.h1 * n1[0].y + hz.s0 * n0[2].y + hz.s1 * n1[2].y;     // p 1, q 0
            const double v01 = hz.h0 * n0[1].x + hz.h1 * n1[1].x + hz.s0 * n0[3].x + hz.s1 * n1[3].x;     // p 0, q 1
            const double v11 = hz.h0 * n0[1].y + hz.h1 * n1[1].y + hz.s0 * n0[3].y + hz.s1 * n1[3].y;     // p 1, q 1
            const double wy = b ? hy.h1 : hy.h0, wys = b ? hy.s1 : hy.s0;
            ux[a][0] += wy * v00 + wys * v01;
            ux[a][1] += wy * v10 + wys * v11;
        }
    }
    return hx.h0 * ux[0][0] + hx.s0 * ux[0][1] + hx.h1 * ux[1][0] + hx.s1 * ux[1][1];
}

// The same sample evaluated by a FULL wave whose lanes are consecutive samples of one ray: the upper node (k + 1) of lane l
// is the lower node (k) of lane l + 1 whenever the two samples sit in the same column one cell apart (80 % of the lane
// pairs at the bench geometry), so a lane loads its lower node only (64 B) and takes the upper one from its neighbour
// with 16 wave_shl DPP moves; the lanes without such a neighbour load theirs under an execution mask.  Each 16-B load
// instruction of this kernel costs one L1 look-up PER LANE (the lanes are 64 B apart), which is what binds it: this halves
// the full-wave instructions.  Same values, same arithmetic: bit-identical to tricubic_lm.  Every lane must be active.
__device__ __forceinline__ double2 dpp_next_lane(double2 v) {
    // (bound_ctrl: lane 63, which has no next lane, reads 0 -- it always loads its own upper node -- and the moves need no
    //  initialised destination)
    const int q[4] = {__double2loint(v.x), __double2hiint(v.x), __double2loint(v.y), __double2hiint(v.y)};
    int o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) o[t] = __builtin_amdgcn_update_dpp(0, q[t], 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
    return make_double2(__hiloint2double(o[1], o[0]), __hiloint2double(o[3], o[2]));
}
struct LmCols {
    const char *c[2][2];      // F8 + record offset of the corner columns (a, b): uniform, kept in SGPRs
};
__device__ __forceinline__ LmCols lm_cols(const double *F8, int ny, int nz) {
    LmCols C;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) C.c[a][b] = (const char *)(F8 + ((size_t)a * LM_SI(ny, nz) + (size_t)b * LM_NZP(nz)) * LM_NF);
    return C;
}
__device__ __forceinline__ double tricubic_lm_wave(const LmCols &C, int ny, int nz, double fx, double fy, double fz) {
    const double fi = __builtin_floor(__builtin_fabs(fx)), fj = __builtin_floor(__builtin_fabs(fy)),
                 fk = __builtin_floor(__builtin_fabs(fz));
    const Herm hx = hermite(fx - fi), hy = hermite(fy - fj), hz = hermite(fz - fk);
    const double lin = __builtin_fma(fi, (double)LM_SI(ny, nz), __builtin_fma(fj, (double)LM_NZP(nz), fk));
    const unsigned node = (unsigned)lin;      // (the fast tier is limited to field arrays below 4 GiB on the host)
    const unsigned next = (unsigned)__builtin_amdgcn_update_dpp(-2, (int)node, 0x130, 0xf, 0xf, false);
    const bool own = next != node + 1;        // no neighbour holding my upper node (lane 63 keeps -2)
    const unsigned boff = node * (unsigned)(LM_NF * sizeof(double));
    double res = 0.0;
#pragma unroll
    for (int a = 0; a < 2; ++a) {             // one x-plane = two columns per batch: 8 full + 8 masked loads in flight
        double2 n0[2][4], n1[2][4];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int t = 0; t < 4; ++t) n0[b][t] = ((const double2 *)(C.c[a][b] + boff))[t];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int t = 0; t < 4; ++t) n1[b][t] = dpp_next_lane(n0[b][t]);
        if (own) {      // (the masked loads land in the registers the DPP moves wrote)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int t = 0; t < 4; ++t) n1[b][t] = ((const double2 *)(C.c[a][b] + boff))[4 + t];
        }
        double u0 = 0.0, u1 = 0.0;            // p 0 / p 1 after the z and y contractions
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            // z: value fields (r = 0) sit in n?[0..1], their z-slopes (r = 1) in n?[2..3]; component .x = p 0, .y = p 1
            const double v00 = hz.h0 * n0[b][0].x + hz.h1 * n1[b][0].x + hz.s0 * n0[b][2].x + hz.s1 * n1[b][2].x;     // p 0, q 0
            const double v10 = hz.h0 * n0[b][0].y + hz.h1 * n1[b][0].y + hz.s0 * n0[b][2].y + hz.s1 * n1[b][2].y;     // p 1, q 0
            const double v01 = hz.h0 * n0[b][1].x + hz.h1 * n1[b][1].x + hz.s0 * n0[b][3].x + hz.s1 * n1[b][3].x;     // p 0, q 1
            const double v11 = hz.h0 * n0[b][1].y + hz.h1 * n1[b][1].y + hz.s0 * n0[b][3].y + hz.s1 * n1[b][3].y;     // p 1, q 1
            const double wy = b ? hy.h1 : hy.h0, wys = b ? hy.s1 : hy.s0;
            u0 += wy * v00 + wys * v01;
            u1 += wy * v10 + wys * v11;
        }
        res += a ? hx.h1 * u0 + hx.s1 * u1 : hx.h0 * u0 + hx.s0 * u1;
    }
    return res;
}

// straight ray in ideal grid coordinates, valid when both end points lie in the tricubic domain g[2] .. g[n-3]
__device__ __forceinline__ URay load_uray_cubic(const GridView &g, const double *origins, const double *dirs, int64_t r,
                                                double tmax, int Ns) {
    URay u = load_uray(g, origins, dirs, r, tmax, Ns);
    const double ox = origins[3 * r], oy = origins[3 * r + 1], oz = origins[3 * r + 2];
    const double dx = dirs[3 * r], dy = dirs[3 * r + 1], dz = dirs[3 * r + 2];
    const double nrm = sqrt(dx * dx + dy * dy + dz * dz);
    const double pz = dz / nrm, sx = dx / nrm / pz, sy = dy / nrm / pz, L = tmax - oz;
    const double xe = ox + sx * L, ye = oy + sy * L, ze = oz + L;
    u.valid = (ox >= g.c0[0]) & (ox <= g.clast[0]) & (xe >= g.c0[0]) & (xe <= g.clast[0]) & (oy >= g.c0[1]) &
              (oy <= g.clast[1]) & (ye >= g.c0[1]) & (ye <= g.clast[1]) & (oz >= g.c0[2]) & (oz <= g.clast[2]) &
              (ze >= g.c0[2]) & (ze <= g.clast[2]);
    return u;
}

// Same wave / chunk structure as k_forward_straight_u (iono_forward_kernels.h): one wave per ray, lanes = samples,
// lane-parallel ray set-up for groups of 16 rays, DPP Simpson reduction, quadrature weights in LDS.  The per-ray state of a
// group is parked in LDS (LM_RS doubles per ray, read back as wave-uniform broadcasts): the sample loop keeps 64 VGPRs of
// field data in flight and has none to spare for 16 rays' worth of lane-private state.
// (2, 3 or 4 workgroups per CU measured the same to 4 %: the kernel is bound by L1 look-ups, not by latency; 2 leaves the
//  compiler 256 VGPRs and no spills)
#ifndef LM_WG
#define LM_WG 2
#endif
#define LM_RS 10      // fx0 fy0 fz0 dfx dfy dfz h tail ray-index (as double) pad
__global__ __launch_bounds__(256, LM_WG) void k_forward_straight_lm(GridView g, const double *__restrict__ F8,
                                                                 const double *__restrict__ origins,
                                                                 const double *__restrict__ dirs, const int *__restrict__ order,
                                                                 int64_t R, double tmax, int Ns, int walk_mode,
                                                                 const double *__restrict__ unitw, double *__restrict__ tec,
                                                                 int *oob_flag) {
    extern __shared__ __attribute__((aligned(16))) double wlds[];
    for (int t = threadIdx.x; t < Ns; t += blockDim.x) wlds[t] = unitw[t];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    double *rs = wlds + ((Ns + 1) & ~1) + (threadIdx.x >> 6) * (U_MAXG * LM_RS);      // this wave's ray-state block
    const int nfull = Ns >> 6, ntail0 = nfull << 6;
    const bool tail_by_lane = (Ns - ntail0) <= 8;
    const Chunk ch = wave_chunk(R, walk_mode, nullptr);
    const LmCols C = lm_cols(F8, g.ny, g.nz);
    const double dlane = (double)lane;
    const double *wp = wlds + lane;
    bool oob = false;
    for (int64_t q0 = ch.lo; q0 < ch.hi; q0 += U_MAXG * ch.stride) {
        const int cnt = (int)min((int64_t)U_MAXG, (ch.hi - q0 + ch.stride - 1) / ch.stride);
        bool valid = false;
        if (lane < cnt) {
            const int64_t q = q0 + lane * ch.stride;
            const int64_t r = order ? (int64_t)order[q] : q;
            const URay u = load_uray_cubic(g, origins, dirs, r, tmax, Ns);
            double tail = 0.0;
            if (u.valid && tail_by_lane) {
                for (int k = ntail0; k < Ns; ++k) {
                    const double kd = (double)k;
                    tail += wlds[k] * tricubic_lm(F8, g.ny, g.nz, fma(kd, u.dfx, u.fx0), fma(kd, u.dfy, u.fy0), fma(kd, u.dfz, u.fz0));
                }
            }
            valid = u.valid;
            if (!valid) {
                oob = true;
                tec[r] = nan("");
            }
            double *w = rs + lane * LM_RS;
            w[0] = u.fx0, w[1] = u.fy0, w[2] = u.fz0, w[3] = u.dfx, w[4] = u.dfy, w[5] = u.dfz, w[6] = u.h, w[7] = tail, w[8] = (double)r;
        }
        const unsigned long long vmask = __ballot(valid);
        // (LDS operations of one wave execute in order: the broadcasts below see the writes above; this only stops the compiler)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int gi = 0; gi < cnt; ++gi) {
            if (!((vmask >> gi) & 1)) continue;
            const double *p = rs + gi * LM_RS;               // wave-uniform address: one broadcast per read
            const double dfx = p[3], dfy = p[4], dfz = p[5];
            double fx = fma(dlane, dfx, p[0]);
            double fy = fma(dlane, dfy, p[1]);
            double fz = fma(dlane, dfz, p[2]);
            const double sx64 = 64.0 * dfx, sy64 = 64.0 * dfy, sz64 = 64.0 * dfz;
            double acc = 0.0;
            for (int it = 0; it < nfull; ++it) {
                acc = fma(wp[it << 6], tricubic_lm_wave(C, g.ny, g.nz, fx, fy, fz), acc);
                fx += sx64;
                fy += sy64;
                fz += sz64;
            }
            if (!tail_by_lane && lane + ntail0 < Ns) acc = fma(wp[ntail0], tricubic_lm(F8, g.ny, g.nz, fx, fy, fz), acc);
            const double total = wave_sum_dpp(acc);
            if (lane == 0) tec[(int64_t)p[8]] = (total + p[7]) * p[6];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // the next group overwrites the ray-state block
        __builtin_amdgcn_wave_barrier();
    }
    if (__any(oob) && lane == 0) atomicOr(oob_flag, 1);
}

// ---- tricubic forward on the bundle plan (iono_forward_kernels.h: bundle-stationary forward) ---------------------------------------
// k_forward_straight_lm moves 512 B per sample through 16-B loads whose lanes sit 64 B apart: one L1 tag look-up per lane and
// load, 1.44 G per launch, which is what binds it.  The bundle plan of the trilinear forward -- <= 64 neighbouring rays
// per workgroup, lane = ray, per B_KC samples a window of <= B_CAPCOLS columns x B_LEV levels that holds every cell the bundle
// touches -- serves the Lekien-Marsden form: a sample still needs the 2 x 2 x 2 nodes of its cell, only 64 B of each.
// The interpolant is a SUM over the four (q, r) field pairs of independent 8-corner contractions
//     f = sum_{q,r} sum_{a,b,c} Hy[b][q](ty) Hz[c][r](tz) ( Hx[a][0](tx) F[0,q,r] + Hx[a][1](tx) F[1,q,r] )(i+a, j+b, k+c),
// so the four waves of the workgroup take one pair each (the PAIR-major arrays FP[t = q + 2 r][node] = (value, Dx value): 16 B
// per node, contiguous along z), stage ITS window with LDS-DMA (one node per lane, BL_CPL columns per wave-load) and walk ALL the
// samples; the four partial integrals are added in a fixed order.  Per ray the arithmetic never depends on the bundling, and a
// chunk whose window does not fit reads the same 16-B nodes from memory: results do not depend on the plan.
// Chunks are BL_KC = 4 samples here (their own windows in the plan: BL_LEV = 6 levels >= 3 dfz + 2 + the spread of the lanes).
// Round 5: the kernel lives on the waves that hide its L1 fills (two workgroups per CU instead of three: 0.94 against 0.76 ms,
// profiles/r05_cubic_forward_occupancy.json), so it is built for FOUR workgroups = sixteen waves per CU: an image of BL_CAPCOLS = 100
// columns = 9.6 KB per wave (4 x 40.4 KB of a CU's 160 KB; 120 columns = 11.5 KB gave three) and at most 128 VGPRs (the sample loops keep
// ONE set of eight nodes in flight instead of two: 166 before) -- 0.80 -> 0.735 ms on the box that measured both.
typedef double lm_d2 __attribute__((ext_vector_type(2)));
#define BL_NODE 16
#define BL_KC 4                                   // samples per chunk
#define BL_LEV 6                                  // levels per staged column
#define BL_COL (BL_LEV * BL_NODE)                 // 96 B per staged column
#define BL_CPL (64 / BL_LEV)                      // columns per staging wave-load (10: 60 lanes, one node each)
#define BL_WAVE_LDS (BL_CAPCOLS * BL_COL)         // 9 600 B per wave
#define BL_LDS_BYTES (B_SPLIT * BL_WAVE_LDS + B_SPLIT * 64 * (int)sizeof(double))

struct LmPairW {          // weights of one sample for the wave's pair (q, r): x needs both kinds, y and z one kind each
    double xh0, xh1, xs0, xs1, w00, w01, w10, w11;      // w[b][c] = Hy[b][q] Hz[c][r]
};
template <int Q, int RZ>
__device__ __forceinline__ LmPairW lm_pair_weights(double tx, double ty, double tz) {
    const Herm hx = hermite(tx), hy = hermite(ty), hz = hermite(tz);
    const double y0 = Q ? hy.s0 : hy.h0, y1 = Q ? hy.s1 : hy.h1, z0 = RZ ? hz.s0 : hz.h0, z1 = RZ ? hz.s1 : hz.h1;
    LmPairW w;
    w.xh0 = hx.h0, w.xh1 = hx.h1, w.xs0 = hx.s0, w.xs1 = hx.s1;
    w.w00 = y0 * z0, w.w01 = y0 * z1, w.w10 = y1 * z0, w.w11 = y1 * z1;
    return w;
}
__device__ __forceinline__ double lm_pair_value(const LmPairW &w, const lm_d2 (&n)[8]) {      // n[a * 4 + b * 2 + c]
    const lm_d2 s0 = w.w00 * n[0] + w.w01 * n[1] + w.w10 * n[2] + w.w11 * n[3];
    const lm_d2 s1 = w.w00 * n[4] + w.w01 * n[5] + w.w10 * n[6] + w.w11 * n[7];
    return (w.xh0 * s0.x + w.xs0 * s0.y) + (w.xh1 * s1.x + w.xs1 * s1.y);
}

// Two ways through a chunk, chosen by the size of its window (wave-uniform, from the plan):
//  * at most BL_ALL = 25 columns (the median window has 20): ONE wave -- wave c mod 4 -- copies the windows of all four pairs into
//    the four quarters of its image and evaluates the whole interpolant for its lanes: position, floor and the three Hermite
//    sets are then computed once per sample instead of once per sample and wave (31 of the 67 vector instructions);
//  * larger: every wave copies the window of ITS pair and adds its quarter of the interpolant, as above.
// One wave-load moves 60 nodes = BL_CPL columns x BL_LEV levels: floor(BL_CPL / wy) whole rows of a window wy <= BL_CPL wide, or one
// of the two column groups of a wider row.
#define BL_ALL (BL_CAPCOLS / 4)
struct LmWindow {
    int imin, jmin, kz0, wx, wy, rpl, fits;      // rpl: rows per staging wave-load (bits 20..23 of the plan's window word)
};
__device__ __forceinline__ LmWindow lm_window(const uint4 *__restrict__ wb, int c) {
    const uint4 w = wb[c];
    LmWindow W;
    W.imin = __builtin_amdgcn_readfirstlane((int)w.x), W.jmin = __builtin_amdgcn_readfirstlane((int)w.y);
    W.kz0 = __builtin_amdgcn_readfirstlane((int)w.z);
    const int wxy = __builtin_amdgcn_readfirstlane((int)w.w);
    W.wx = wxy & 255, W.wy = (wxy >> 8) & 255, W.fits = (wxy >> 16) & 1, W.rpl = (wxy >> 20) & 15;
    return W;
}
// copy window W of the pair array into the lane-linear image at `dst`: column (di, dj) at byte (di * wy + dj) * 96, level at + 16 l.
// Lane = (staged column col = lane / 6, level lane % 6); a window at most BL_CPL columns wide puts rpl whole rows into one load
// (col = dr * wy + dj), a wider one takes two loads per row.
__device__ __forceinline__ void lm_stage(const GridView &g, const lm_d2 *__restrict__ FPt, const LmWindow &W, char *dst) {
#if defined(IONO_BL_ABL) && IONO_BL_ABL == 1      // timing-only build (WRONG results): no staging
    return;
#endif
    const int lane = threadIdx.x & 63;
    const int col = (int)(((unsigned)lane * 10923u) >> 16), lev = lane - BL_LEV * col;       // lane / 6, lane % 6
    const unsigned si = (unsigned)g.ny * (unsigned)g.nz;       // (field arrays below 4 GiB per pair: cubic_fast_ok)
    const char *base = (const char *)FPt + ((size_t)((size_t)W.imin * g.ny + W.jmin) * g.nz + W.kz0) * BL_NODE;
    if (W.wy <= BL_CPL) {
        // dr = col / wy without an integer division: floor((col + 1/2) / wy) in float is exact for these small integers
        const int dr = (int)(((float)col + 0.5f) * __builtin_amdgcn_rcpf((float)W.wy));
        const int dj = col - dr * W.wy;
        const unsigned off = ((unsigned)dr * si + (unsigned)dj * (unsigned)g.nz + (unsigned)lev) * BL_NODE;
        const int lim = dr < W.rpl && lane < BL_CPL * BL_LEV ? W.wx - dr : 0;      // this lane copies rows di + dr for di < lim
        const unsigned step = (unsigned)(W.rpl * W.wy) * BL_COL;
        const size_t bstep = (size_t)W.rpl * si * BL_NODE;
        for (int di = 0; di < W.wx; di += W.rpl) {
            if (di < lim)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + off),
                                                 (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            base += bstep;
            dst += step;
        }
    } else {
        const unsigned off = ((unsigned)col * (unsigned)g.nz + (unsigned)lev) * BL_NODE;
        const unsigned rstride = (unsigned)W.wy * BL_COL;
        for (int di = 0; di < W.wx; ++di) {
            if (col < BL_CPL)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + off),
                                                 (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            if (col < BL_CPL && col + BL_CPL < W.wy)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + off + (size_t)BL_CPL * g.nz * BL_NODE),
                                                 (__attribute__((address_space(3))) void *)(dst + BL_CPL * BL_COL), 16, 0, 0);
            base += (size_t)si * BL_NODE;
            dst += rstride;
        }
    }
}

typedef const __attribute__((address_space(3))) lm_d2 *lm_lds_node;
__device__ __forceinline__ void lm_read8(lm_d2 (&n)[8], unsigned a, unsigned row2) {
    const __attribute__((address_space(3))) char *p0 = (const __attribute__((address_space(3))) char *)(size_t)a;
    const __attribute__((address_space(3))) char *p1 = (const __attribute__((address_space(3))) char *)(size_t)(a + row2);
    n[0] = *(lm_lds_node)(p0), n[1] = *(lm_lds_node)(p0 + BL_NODE), n[2] = *(lm_lds_node)(p0 + BL_COL), n[3] = *(lm_lds_node)(p0 + BL_COL + BL_NODE);
    n[4] = *(lm_lds_node)(p1), n[5] = *(lm_lds_node)(p1 + BL_NODE), n[6] = *(lm_lds_node)(p1 + BL_COL), n[7] = *(lm_lds_node)(p1 + BL_COL + BL_NODE);
}
template <int Q, int RZ>
__device__ __forceinline__ double lm_pair_of(const Herm &hx, const Herm &hy, const Herm &hz, const lm_d2 (&n)[8]) {
    const double y0 = Q ? hy.s0 : hy.h0, y1 = Q ? hy.s1 : hy.h1, z0 = RZ ? hz.s0 : hz.h0, z1 = RZ ? hz.s1 : hz.h1;
    LmPairW w;
    w.xh0 = hx.h0, w.xh1 = hx.h1, w.xs0 = hx.s0, w.xs1 = hx.s1;
    w.w00 = y0 * z0, w.w01 = y0 * z1, w.w10 = y1 * z0, w.w11 = y1 * z1;
    return lm_pair_value(w, n);
}

template <int Q, int RZ>
__device__ __forceinline__ double bundle_lm_walk(const GridView &g, const lm_d2 *__restrict__ FP, int64_t npad, const BundleRays &B,
                                                 const uint4 *__restrict__ wb, int nchunks, int Ns, const double *__restrict__ unitw, char *img,
                                                 int wid) {
    const lm_d2 *FPt = FP + (size_t)wid * npad;          // this wave's pair (q, r) = (Q, RZ)
    const size_t sj = (size_t)g.nz, si = (size_t)g.ny * g.nz;
    double acc = 0.0;
    for (int c = 0; c < nchunks; ++c) {
        const int k0 = c * BL_KC;
        int ke = min(k0 + BL_KC, Ns);
        LmWindow W = lm_window(wb, c);
        if (B.stale) W.fits = 0;          // rays edited since the plan was made: every chunk reads its nodes from memory (exact for any bundling)
        const bool all = W.fits && W.wx * W.wy <= BL_ALL;
        if (all && (c & (B_SPLIT - 1)) != wid) continue;          // another wave takes the whole chunk
        const double kd0 = (double)k0;
        double fx = fma(kd0, B.dfx, B.fx0), fy = fma(kd0, B.dfy, B.fy0), fz = fma(kd0, B.dfz, B.fz0);
        // the chunk's quadrature weights in one scalar load, BEFORE the LDS reads (scalar loads share their counter: waiting for
        // one inside the loop would wait for every read in flight); weights beyond the last sample are zeros
        const double w0 = unitw[k0], w1 = unitw[k0 + 1], w2 = unitw[k0 + 2], w3 = unitw[k0 + 3];
        static_assert(BL_KC == 4, "four weights per chunk");
        const double cw = (double)(W.wy * BL_LEV), cj = (double)BL_LEV;
        const unsigned ibase = (unsigned)(size_t)img - (((unsigned)W.imin * (unsigned)W.wy + (unsigned)W.jmin) * BL_LEV + (unsigned)W.kz0) * BL_NODE;
        const unsigned row2 = (unsigned)W.wy * BL_COL;
        if (all) {
            // ---- all four pairs by this wave: pair t in quarter t of the image -------------------------------------------------------
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the previous chunk's LDS reads have returned: the image may be overwritten
#pragma unroll
            for (int t = 0; t < 4; ++t) lm_stage(g, FP + (size_t)t * npad, W, img + t * (BL_ALL * BL_COL));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            constexpr unsigned PS = BL_ALL * BL_COL;
#if defined(IONO_BL_ABL) && IONO_BL_ABL == 2      // timing-only build (WRONG results): staging only
            ke = k0 + 1;
#endif
            for (int k = k0; k < ke; ++k) {
                const double fi = __builtin_floor(__builtin_fabs(fx)), fj = __builtin_floor(__builtin_fabs(fy)), fk = __builtin_floor(__builtin_fabs(fz));
                const Herm hx = hermite(fx - fi), hy = hermite(fy - fj), hz = hermite(fz - fk);
                const unsigned a = (unsigned)__builtin_fma(fi, cw, __builtin_fma(fj, cj, fk)) * BL_NODE + ibase;
                lm_d2 na[8];
                lm_read8(na, a, row2);
                double v = lm_pair_of<0, 0>(hx, hy, hz, na);
                lm_read8(na, a + PS, row2);
                v += lm_pair_of<1, 0>(hx, hy, hz, na);
                lm_read8(na, a + 2 * PS, row2);
                v += lm_pair_of<0, 1>(hx, hy, hz, na);
                lm_read8(na, a + 3 * PS, row2);
                v += lm_pair_of<1, 1>(hx, hy, hz, na);
                const int dk = k - k0;
                acc = fma(dk == 0 ? w0 : dk == 1 ? w1 : dk == 2 ? w2 : w3, v, acc);
                fx += B.dfx, fy += B.dfy, fz += B.dfz;
            }
        } else if (W.fits) {
            // ---- this wave's pair --------------------------------------------------------------------------------------------------------
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lm_stage(g, FPt, W, img);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // one sample: weights, the LDS address of its cell's lower corner, eight ds_read_b128 (issued, not yet waited for)
            auto fetch = [&](LmPairW &pw, lm_d2 (&n)[8]) {
                const double fi = __builtin_floor(__builtin_fabs(fx)), fj = __builtin_floor(__builtin_fabs(fy)), fk = __builtin_floor(__builtin_fabs(fz));
                pw = lm_pair_weights<Q, RZ>(fx - fi, fy - fj, fz - fk);
                lm_read8(n, (unsigned)__builtin_fma(fi, cw, __builtin_fma(fj, cj, fk)) * BL_NODE + ibase, row2);
                fx += B.dfx, fy += B.dfy, fz += B.dfz;
            };
            LmPairW pa;
            lm_d2 na[8];
#if defined(IONO_BL_ABL) && IONO_BL_ABL == 2      // timing-only build (WRONG results): staging only
            fetch(pa, na);
            acc = fma(w0, lm_pair_value(pa, na), acc);
#else
            for (int k = k0; k < ke; ++k) {
                fetch(pa, na);
                const int dk = k - k0;
                acc = fma(dk == 0 ? w0 : dk == 1 ? w1 : dk == 2 ? w2 : w3, lm_pair_value(pa, na), acc);
            }
#endif
        } else {
            for (int k = k0; k < ke; ++k) {
                const double fi = __builtin_floor(__builtin_fabs(fx)), fj = __builtin_floor(__builtin_fabs(fy)), fk = __builtin_floor(__builtin_fabs(fz));
                const LmPairW pw = lm_pair_weights<Q, RZ>(fx - fi, fy - fj, fz - fk);
                const lm_d2 *q0 = FPt + (size_t)__builtin_fma(fi, (double)si, __builtin_fma(fj, (double)sj, fk)), *q1 = q0 + si;
                lm_d2 n[8];
                n[0] = q0[0], n[1] = q0[1], n[2] = q0[sj], n[3] = q0[sj + 1], n[4] = q1[0], n[5] = q1[1], n[6] = q1[sj], n[7] = q1[sj + 1];
                acc = fma(unitw[k], lm_pair_value(pw, n), acc);
                fx += B.dfx, fy += B.dfy, fz += B.dfz;
            }
        }
    }
    return acc;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_forward_bundle_lm(GridView g, const double *__restrict__ FP, int64_t npad, const double *__restrict__ origins,
                                                           const double *__restrict__ dirs, const int *__restrict__ order,
                                                           const int *__restrict__ bstart, const uint4 *__restrict__ win, const uint2 *__restrict__ rhash,
                                                           int nb, int nchunks, double tmax, int Ns, const double *__restrict__ unitw,
                                                           double *__restrict__ tec, int *oob_flag, int restricted) {
    extern __shared__ __attribute__((aligned(16))) char blds[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);     // XCD-major
    if (b >= nb) return;
    const BundleRays B = load_bundle<true>(g, origins, dirs, order, bstart, b, tmax, Ns, rhash);
    if (wid == 0 && __any(B.mine && !B.valid) && lane == 0) atomicOr(oob_flag, 1);
    if (wid == 0 && B.stale && lane == 0) atomicOr(oob_flag + 2, 1);
    if (!B.any) {
        if (wid == 0 && B.mine) tec[B.r] = nan("");
        return;
    }
    char *img = blds + wid * BL_WAVE_LDS;
    double *part = (double *)(blds + B_SPLIT * BL_WAVE_LDS);
    const lm_d2 *FPd = (const lm_d2 *)FP;
    const uint4 *wb = win + (size_t)b * nchunks;
    double acc;
    if (B.stale && restricted) {
        // The rays of this bundle were edited in place since the plan was made AND the pair arrays were rebuilt only where the PLANNED
        // rays read them (ensure_lm_fields): direct loads along the NEW rays could meet nodes that were not rebuilt.  The node VALUES are
        // always current: 216 taps per sample straight from them, wave w the samples k = w mod 4 -- slow, exact, and only for the
        // bundles concerned (B holds the rays as the arrays have them now: load_bundle)
        acc = 0.0;
        if (B.valid)
            for (int k = wid; k < Ns; k += B_SPLIT) {
                const double kd = (double)k;
                acc = fma(unitw[k], tricubic_from_nodes((const double *)g.M, g.nx, g.ny, g.nz, fma(kd, B.dfx, B.fx0), fma(kd, B.dfy, B.fy0), fma(kd, B.dfz, B.fz0)), acc);
            }
    } else if (wid == 0) acc = bundle_lm_walk<0, 0>(g, FPd, npad, B, wb, nchunks, Ns, unitw, img, 0);
    else if (wid == 1) acc = bundle_lm_walk<1, 0>(g, FPd, npad, B, wb, nchunks, Ns, unitw, img, 1);
    else if (wid == 2) acc = bundle_lm_walk<0, 1>(g, FPd, npad, B, wb, nchunks, Ns, unitw, img, 2);
    else acc = bundle_lm_walk<1, 1>(g, FPd, npad, B, wb, nchunks, Ns, unitw, img, 3);
    part[wid * 64 + lane] = acc;
    __syncthreads();
    if (wid == 0 && B.mine) {
        const double *pl = part + lane;
        tec[B.r] = B.valid ? (((pl[0] + pl[64]) + pl[128]) + pl[192]) * B.h : nan("");
    }
}

// ---- general tier of the tricubic transpose: 216 hardware atomics per sample (any grid; explicit or straight rays) ----
template <typename AT>
__device__ __forceinline__ void scatter_tricubic(const GridView &g, const Axes &ax, AT *__restrict__ G, double x, double y,
                                                 double z, double c) {
    double wx[6], wy[6], wz[6], dd[6];
    const int i = cubic_axis(ax.x, g.nx, x, g.inv_h[0], g.uniform[0], wx, dd, false);
    const int j = cubic_axis(ax.y, g.ny, y, g.inv_h[1], g.uniform[1], wy, dd, false);
    const int k = cubic_axis(ax.z, g.nz, z, g.inv_h[2], g.uniform[2], wz, dd, false);
    AT *base = G + ((size_t)(i - 2) * g.ny + (j - 2)) * g.nz + (k - 2);
    for (int a = 0; a < 6; ++a) {
        const double wa = c * wx[a];
        for (int b = 0; b < 6; ++b) {
            const double wab = wa * wy[b];
            AT *p = base + ((size_t)a * g.ny + b) * g.nz;
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) atomicAdd(p + cc, (AT)(wab * wz[cc]));
        }
    }
}

}  // namespace

#endif
