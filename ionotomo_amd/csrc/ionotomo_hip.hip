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
#include <dlfcn.h>
#include <rccl/rccl.h>      // declarations only: the library is dlopen-ed by iono_comm_* (no link-time dependency)

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>      // device sort of the bundle plan's Morton keys

#include "../../include/ionotomo_hip.h"

#define IONO_VERSION 200
#define PLASMA_A (8.980 * 8.980)             // inversion/fermat.py:42
#define SPEED_OF_LIGHT 299792458.0           // inversion/iterative_newton.py:15

#include "iono_device_common.h"
#include "iono_forward_kernels.h"
#include "iono_forward_f32_kernels.h"
#include "iono_adjoint_kernels.h"
#include "iono_cubic_kernels.h"
#include "iono_solver_kernels.h"
#include "iono_binned_kernels.h"
#include "iono_aux_kernels.h"
#include "iono_fermat_kernels.h"

namespace {
thread_local std::string g_last_error;
}

// ================================================================================================
// host side
// ================================================================================================
struct iono_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    int nx = 0, ny = 0, nz = 0, storage = IONO_F64;
    double *d_axes = nullptr;
    void *d_M = nullptr;             // library-owned node values (storage type)
    double *d_M_ext = nullptr;       // caller-owned float64 values bound with iono_grid_bind_values_dev (takes precedence)
    double inv_h[3] = {0, 0, 0};
    int uniform[3] = {0, 0, 0};
    int *d_flags = nullptr;          // [0] out-of-bounds, [1] non-finite, [2] a planned launch met rays its plan was not made for, [3] spare
    double *d_unitw = nullptr;       // cached unit-spacing quadrature weights
    float *d_unitw32 = nullptr;      // ... and their float32 copy, inside the same allocation
    int unitw_n = 0, unitw_rule = -1;
    std::string err;
    int num_cus = 256;
    int force_general = 0;           // testing/ablation: 1 = general kernels only, 2 = no "ideal uniform" kernels
    void *d_work = nullptr;          // workspace of the host-pointer entry points (grow-only)
    // measured load balance (iono_walk_cycles / iono_walk_partition_set): [0] forward (one chunk per wave),
    // [1] LDS-tiled adjoint (one chunk per workgroup + dynamically handed-out extras)
    struct WalkPart {
        int64_t *d_starts = nullptr;
        int n = 0;                       // chunks in d_starts (0 = none)
        int64_t R = -1;                  // ray count it was built for
        unsigned long long *d_cyc = nullptr;
        int cyc_cap = 0, last_n = 0, last_units = 0;     // cycles of the last launch: entries, resident waves / workgroups
    } walk[2];
    unsigned int *d_chunk_counter = nullptr;
    size_t work_cap = 0;
    double *d_kern = nullptr;        // 3 x (2h+1) smoothing kernels
    int kern_cap = 0;
    double *d_nM = nullptr;          // refractive-index nodes for the tracer (device-pointer entry), lazily built
    double nM_freq = -1.0;           // frequency d_nM was built for; < 0 = stale
    double *d_nF8 = nullptr;         // Lekien-Marsden records of the refractive index (tricubic tracer on ideal grids), lazily built
    double nF8_freq = -1.0;          // frequency they were built for; < 0 = stale
    int variant = 0;                 // kernel variant for A/B runs (env IONOTOMO_VARIANT)
    int hybrid_min = 0;              // forward plan: bundles of fewer rays are left to the lanes = samples kernel.  0 (default): chosen per
                                     //   plan by the cost model in iono_forward_plan_dev; env IONOTOMO_HYBRID_MIN=1..65 forces it (1: every
                                     //   bundle to the bundle kernels; 65: none -- the plan then only orders the walk)
    int blocks_per_cu_override = 0;  // env IONOTOMO_BLOCKS_PER_CU
    ncclComm_t comm = nullptr;       // iono_comm_init
    int comm_ranks = 0;
    int seg_lanes = 0;               // env IONOTOMO_SEG_LANES=4|8|16: lanes per segment of the back-projection plan (0: chosen per geometry)
    int adj_mode = 0;                // -DIONO_ABLATION builds only: timing ablations of the tiled adjoint (env IONOTOMO_ADJ_ABLATE)
    int fermat_lm_lanes = 0;            // record tracer / fused TEC through a tricubic index: lanes per ray (8 or 2); 0 = by batch size
    int lm4_groups = 0;                 // env IONOTOMO_LM4_GROUPS: persistent workgroups of the planned tricubic transpose (A/B; default: one per CU)
                                        // (env IONOTOMO_FERMAT_LM_LANES): 8 below fermat_lm_few_min rays, 2 from there on
    int64_t fermat_lm_few_min = 32768;  // (env IONOTOMO_FERMAT_LM_FEW_MIN)
    int64_t fermat_coop_max = INT64_MAX;   // tricubic tracer: 8 lanes per ray (faster than lanes = rays at every batch size
                                           // measured since it caches its stencil; env IONOTOMO_FERMAT_COOP_MAX for A/B)
    int fermat_coop_rpw = 0;               // rays per wave of that kernel, 1..8: 0 = default (env IONOTOMO_FERMAT_COOP_RPW)
    int64_t fermat_poly_max = 4096;     // trilinear tracer on ideal grids: cell-polynomial kernel (few rays per wave) up to this many rays
                                        // (env IONOTOMO_FERMAT_POLY_MAX, _RPW; grids without ideal-uniform axes -- IONOTOMO_FORCE_GENERAL=2 -- get the 4-lanes-per-ray kernel)
    int fermat_poly_rpw = 0;
    int64_t fermat_lin4_max = 4096;     // trilinear tracer: 4 lanes per ray up to this many rays, lanes = rays beyond (crossover
                                        // ~5k rays since the lanes = rays right-hand side dropped the axis tables on ideal grids:
                                        // 2 604 rays 1.05 vs 1.16 ms, 10 416: 1.33 vs 1.21, 78 120: 3.08 vs 1.87; it was ~150k rays
                                        // before.  env IONOTOMO_FERMAT_LIN4_MAX)
    int fermat_lin4_rpw = 0;            // rays per wave of that kernel: 0 = by batch size (env IONOTOMO_FERMAT_LIN4_RPW)
    int ideal = 0;                   // every axis is g0 + i*h to within 2.5e-13 h (np.linspace)
    double g0[3] = {0, 0, 0}, glast[3] = {0, 0, 0};
    double c0[3] = {0, 0, 0}, clast[3] = {0, 0, 0};     // tricubic domain g[2] .. g[n-3] (n >= 6)
    double *d_F8 = nullptr;          // Lekien-Marsden derivative fields [node][8] of the current values (lazily built)
    bool F8_valid = false;
    double *d_FP = nullptr;          // the same fields PAIR-major [4][padded nodes][2] for the bundle-stationary tricubic forward (lazily built)
    bool FP_valid = false;
    int64_t FP_plan_serial = -1;     // ... or valid only on the node lines forward plan number `serial` reads (FwdPlan::d_xrange)
    int64_t fplan_counter = 0;
    double *d_G8 = nullptr;          // channel buffers [8][nodes] of the tricubic transpose (lazily allocated)
    bool deterministic = false;      // iono_set_deterministic / env IONOTOMO_DETERMINISTIC=1: fixed-point back-projection (k_adjoint_binned<.., FIX>)
    unsigned long long *d_fixgrid = nullptr;      // its grid of 64-bit integers [nodes] + the launch's largest |w h| behind it; all zero between launches
    double *d_LMw = nullptr;         // [nodes][6] scratch of the tricubic transpose's folds (H0 | H1 | K0 | K1)
    float4 *d_Q4 = nullptr;          // float32 storage: 2 x 2 (y, z) corner blocks for the 2-loads-per-sample forward (lazily built)
    bool Q4_valid = false;
    // node-stationary back-projection plan (iono_adjoint_plan_dev; iono_binned_kernels.h): geometry only, library-owned
    struct AdjPlan {
        const void *o_key = nullptr, *d_key = nullptr;     // the ray arrays it was built for (caller keeps them unchanged)
        int64_t R = -1, n_entries = 0;
        int Ns = 0, kind = -1, n_units = 0;
        int segl = BIN_SEG;                               // lanes per segment of this plan (4, 8 or 16)
        double tmax = 0;
        double *d_uray = nullptr;
        uint2 *d_hash = nullptr;                          // ray_hash of every ray as planned, by ray index (checked by every planned launch)
        uint2 *d_entries = nullptr;
        BinUnit *d_units = nullptr;
        int nslab = 1;                                    // units are ordered by z-slab (then largest first): slab s = units [slab_unit[s], slab_unit[s + 1])
        int slab_unit[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // ... and owns the node levels [slab_z[s], slab_z[s + 1])
        int slab_z[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        double outside_fraction = 0;                      // segments whose (x, y) extent exceeds the box image
        int fix_bits = 12;                                // deterministic mode: log2 bound of the contributions one node can receive (+ 1)
        bool fix_counted = false;                         // ... tightened by a count of the terms per node (the kernel in counting mode + k_fix_nodemax, first deterministic launch)
        LmTile *d_tiles = nullptr;                        // tricubic plans: output tiles of the z | y | x fold passes (k_lm_fold_*_tiles)
        int tile_n[3] = {0, 0, 0}, tile_off[3] = {0, 0, 0};
        size_t cap_tiles = 0;
        int64_t n_invalid = 0;                            // rays that leave the grid (skipped; every launch raises the flag)
        size_t cap_uray = 0, cap_hash = 0, cap_entries = 0, cap_units = 0;      // bytes (grow-only: a new geometry reuses them)
    } plan;
    // bundle plan of the forward (iono_forward_plan_dev; k_forward_bundle): geometry only, library-owned
    struct FwdPlan {
        const void *o_key = nullptr, *d_key = nullptr;
        int64_t R = -1;
        int Ns = 0, nb = 0, nchunks = 0;      // nb: the bundles the bundle kernels serve (those of >= hybrid_min rays, first in the walk)
        int nb_all = 0, split_min = 0;         // bundles the cut produced; the threshold this plan was split at
        int64_t n_planned = 0, n_rest = 0;     // rays in the nb served bundles | the rest: walk positions [n_planned, R), served by the
                                               //   lanes = samples kernels in the SAME call (hybrid dispatch, round 6)
        int64_t hist[65] = {0};                // bundles by ray count, as cut (before the split)
        double model_us[3] = {0, 0, 0};        // the cost model's estimates: every bundle served | the chosen split | lanes = samples only
        bool forced = false;                   // the threshold was forced (env), not chosen
        double tmax = 0;
        int *d_order = nullptr, *d_bstart = nullptr;
        uint4 *d_win = nullptr;
        uint2 *d_rhash = nullptr;        // 64-bit checksum of every ray's six doubles, in walk order: checked by every planned launch (ray_hash)
        size_t cap_rhash = 0;
        BundleRec *d_brec = nullptr;     // the rays of every bundle as the trilinear kernel wants them, 64 B per (bundle, lane) (k_bundle_records)
        uint2 *d_bhash = nullptr;        // ... and their checksums in the same layout
        size_t cap_brec = 0, cap_bhash = 0;
        uint4 *d_win_lm = nullptr;       // windows of the tricubic kernel's shorter chunks (BL_KC samples, BL_LEV levels of 16-byte nodes)
        int nchunks_lm = 0;
        int2 *d_xrange = nullptr;        // per (j, k) node line along x: first / last plane any window of the tricubic kernel holds (lo > hi: none),
        size_t cap_xrange = 0;           //   what a field rebuild for THIS plan needs to cover (k_lm_touch_lines; only when every window fits)
        bool lm_all_fit = false;
        int64_t serial = 0;              // plan number (FP_plan_serial)
        size_t cap_order = 0, cap_bstart = 0, cap_win = 0, cap_win_lm = 0;
        double fit_fraction = 0;         // chunks whose window fits the LDS image
    } fplan;
    char *h_pinned = nullptr;        // pinned staging of small downloads (iono_dev_download): result + flags in one round trip
    size_t pinned_cap = 0;
    char *h_plan = nullptr;          // pinned staging of the plan builders' host round trips (ray summaries, walk order: 8 MB at the
    size_t plan_pinned_cap = 0;      // bench shape; from pageable memory those copies were half of the 3 ms a forward plan took)
    int plan_slabs = 1;              // z-slabs the next back-projection plan groups its work units by (iono_adjoint_plan_slabs)
    int unit_lo = -1, unit_hi = -1;  // work units of the NEXT planned trilinear back-projection (iono_adjoint_unit_range; -1: all)
    bool plan_verified = false;      // the ray pass in front of a planned back-projection already checked the rays (k_rays_step)
    bool lm4_attr[3] = {false, false, false};      // k_adjoint_binned_lm4<SEGL>: > 64 KB of dynamic LDS allowed (hipFuncSetAttribute, once per context)
    double *d_rayw = nullptr;        // per-ray weights of the fused modes for the binned kernel
    int64_t rayw_cap = 0;
    double *d_freqs = nullptr;       // frequencies of the phase observable on the device (cached copy of h_freqs)
    std::vector<double> h_freqs;
};

namespace {

void plan_free(iono_ctx *c) {
    if (c->plan.d_uray) (void)hipFree(c->plan.d_uray);
    if (c->plan.d_hash) (void)hipFree(c->plan.d_hash);
    if (c->plan.d_entries) (void)hipFree(c->plan.d_entries);
    if (c->plan.d_units) (void)hipFree(c->plan.d_units);
    if (c->plan.d_tiles) (void)hipFree(c->plan.d_tiles);
    c->plan = iono_ctx::AdjPlan();
}
// forget the plan but keep its buffers: the next geometry (a new timestep's directions) reuses them -- hipMalloc / hipFree
// pairs were half of the 19 ms a re-plan cost
void plan_reset(iono_ctx *c) {
    iono_ctx::AdjPlan &p = c->plan, fresh;
    fresh.d_uray = p.d_uray, fresh.d_entries = p.d_entries, fresh.d_units = p.d_units;
    fresh.cap_uray = p.cap_uray, fresh.cap_entries = p.cap_entries, fresh.cap_units = p.cap_units;
    fresh.d_hash = p.d_hash, fresh.cap_hash = p.cap_hash;
    fresh.d_tiles = p.d_tiles, fresh.cap_tiles = p.cap_tiles;
    p = fresh;
}
template <typename T>
hipError_t plan_reserve(T *&ptr, size_t &cap, size_t bytes) {
    if (bytes <= cap && ptr) return hipSuccess;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr, cap = 0;
    const hipError_t e = hipMalloc((void **)&ptr, bytes + bytes / 8 + 256);
    if (e == hipSuccess) cap = bytes + bytes / 8 + 256;
    return e;
}

void fplan_free(iono_ctx *c) {
    if (c->fplan.d_order) (void)hipFree(c->fplan.d_order);
    if (c->fplan.d_bstart) (void)hipFree(c->fplan.d_bstart);
    if (c->fplan.d_win) (void)hipFree(c->fplan.d_win);
    if (c->fplan.d_rhash) (void)hipFree(c->fplan.d_rhash);
    if (c->fplan.d_brec) (void)hipFree(c->fplan.d_brec);
    if (c->fplan.d_bhash) (void)hipFree(c->fplan.d_bhash);
    if (c->fplan.d_win_lm) (void)hipFree(c->fplan.d_win_lm);
    if (c->fplan.d_xrange) (void)hipFree(c->fplan.d_xrange);
    c->fplan = iono_ctx::FwdPlan();
}

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

void *cur_values(const iono_ctx *c) { return c->d_M_ext ? (void *)c->d_M_ext : c->d_M; }

GridView view(const iono_ctx *c) {
    GridView g;
    g.axes = c->d_axes;
    g.M = cur_values(c);
    g.nx = c->nx;
    g.ny = c->ny;
    g.nz = c->nz;
    for (int a = 0; a < 3; ++a) {
        g.inv_h[a] = c->inv_h[a];
        g.uniform[a] = c->uniform[a];
        g.g0[a] = c->g0[a];
        g.glast[a] = c->glast[a];
        g.c0[a] = c->c0[a];
        g.clast[a] = c->clast[a];
    }
    g.ideal = c->ideal && c->force_general == 0 && (uint64_t)c->nx * c->ny * c->nz < ((uint64_t)1 << 32);      // (IONOTOMO_FORCE_GENERAL=2: the tracer's general right-hand side, A/B)
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

// elements of a values array as the kernels address it: the grid + one plane + one row + 2 (unclamped far-corner reads of a
// sample on a top face, weight 0: trilinear_u) + 16 (the LDS windows of k_forward_bundle are staged ten levels at a time)
int64_t padded_count(const iono_ctx *c) { return (int64_t)c->nx * c->ny * c->nz + (int64_t)c->ny * c->nz + c->nz + 2 + 16; }
size_t lds_bytes(const iono_ctx *c) { return sizeof(double) * (size_t)(c->nx + c->ny + c->nz); }
int64_t ncells(const iono_ctx *c) { return (int64_t)c->nx * c->ny * c->nz; }
// fast kernels: all three axes uniform (cell guess off by at most one), 32-bit element offsets
bool fast_path_ok(const iono_ctx *c) {
    // 32-bit BYTE offsets (SGPR base + VGPR offset addressing), padded far-corner reads included
    const uint64_t bytes = (uint64_t)padded_count(c) * (c->storage == IONO_F64 ? 8 : 4);
    return c->uniform[0] && c->uniform[1] && c->uniform[2] && bytes < ((uint64_t)1 << 32) && c->force_general != 1;
}
bool ideal_path_ok(const iono_ctx *c) { return fast_path_ok(c) && c->ideal && c->force_general == 0; }
// the v2 kernels keep the Ns quadrature weights in LDS
bool ideal_path_ok(const iono_ctx *c, int Ns) { return ideal_path_ok(c) && Ns <= 4096; }
// v2 kernels: grid = what is resident at once (blocks per CU from the occupancy query, cached per
// kernel), but no more waves than rays
// the occupancy query costs a few microseconds: remember the answer per (kernel, block size, LDS size)
template <typename K>
int blocks_per_cu(K kernel, int block, size_t lds, int fallback) {
    static thread_local std::vector<std::pair<std::pair<const void *, size_t>, int>> cache;
    const std::pair<const void *, size_t> key((const void *)kernel, lds * 2048 + (size_t)block);
    for (auto &e : cache)
        if (e.first == key) return e.second;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds) != hipSuccess || per_cu < 1) per_cu = fallback;
    cache.push_back({key, per_cu});
    return per_cu;
}
template <typename K>
int resident_blocks(iono_ctx *c, K kernel, size_t lds) {
    int per_cu = blocks_per_cu(kernel, 256, lds, 4);
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

// entry points that need no grid still allocate / launch on the ctx's GPU
int need_ctx(iono_ctx *c) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
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

// the element-wise helpers are functors of ONE kernel (iono_device_common.h:k_map): F{operands...} by value, 256 threads per workgroup
template <class F, class... A>
void launch_map(iono_ctx *c, int blocks, int64_t n, A... a) {
    hipLaunchKernelGGL((k_map<F>), dim3((unsigned)blocks), dim3(256), 0, c->stream, n, F{a...});
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
    w.resize((size_t)Ns + 8, 0.0);      // (zero weights beyond the last sample: kernels may read a whole chunk's weights at once)
    // ... followed by the same weights in float32 (the float32 fast mode's chunk sums: k_forward_bundle_f32), 32-byte aligned
    const size_t off32 = (sizeof(double) * w.size() + 31) & ~(size_t)31;
    std::vector<float> w32(w.size() + 8, 0.0f);
    for (size_t k = 0; k < w.size(); ++k) w32[k] = (float)w[k];
    HIP_TRY(c, hipMalloc((void **)&c->d_unitw, off32 + sizeof(float) * w32.size()));
    HIP_TRY(c, hipMemcpy(c->d_unitw, w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice));
    c->d_unitw32 = (float *)((char *)c->d_unitw + off32);
    HIP_TRY(c, hipMemcpy(c->d_unitw32, w32.data(), sizeof(float) * w32.size(), hipMemcpyHostToDevice));
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
    int v[4] = {0, 0, 0, 0};
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

// Walk order for the LDS-tiled adjoint (speed only): 4-D Morton code of each ray's foot point and far end in grid
// cells (quantised at 3/4 of a cell), so that consecutive rays of the walk nearly coincide all the way up and any
// bundle of them fits the kernel's 8 x 8 tile window.  Host-side integer work + one sort, O(R log R).
void morton_walk_order(const iono_ctx *c, const double *o, const double *d, int64_t R, double tmax, int *order) {
    std::vector<std::pair<uint64_t, int>> key((size_t)R);
    const double q = 4.0 / 3.0;
    for (int64_t r = 0; r < R; ++r) {
        const double L = (tmax - o[3 * r + 2]) / d[3 * r + 2];
        const double v[4] = {(o[3 * r] - c->g0[0]) * c->inv_h[0], (o[3 * r + 1] - c->g0[1]) * c->inv_h[1],
                             (o[3 * r] + d[3 * r] * L - c->g0[0]) * c->inv_h[0],
                             (o[3 * r + 1] + d[3 * r + 1] * L - c->g0[1]) * c->inv_h[1]};
        uint64_t code = 0;
        for (int dim = 0; dim < 4; ++dim) {
            double t = v[dim] * q;
            if (!(t > 0)) t = 0;                      // also NaN
            if (t > 32767.0) t = 32767.0;
            const uint64_t u = (uint64_t)t;
            for (int b = 0; b < 15; ++b) code |= ((u >> b) & 1ull) << (4 * b + dim);
        }
        key[(size_t)r] = {code, (int)r};
    }
    std::stable_sort(key.begin(), key.end(), [](const std::pair<uint64_t, int> &a, const std::pair<uint64_t, int> &b) {
        return a.first < b.first;
    });
    for (int64_t r = 0; r < R; ++r) order[r] = key[(size_t)r].second;
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
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_flags, 4 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(c->d_flags, 0, 4 * sizeof(int));
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
    if (const char *e = getenv("IONOTOMO_HYBRID_MIN")) c->hybrid_min = std::min(65, std::max(0, atoi(e)));
    if (const char *e = getenv("IONOTOMO_SEG_LANES")) c->seg_lanes = atoi(e);
    if (const char *e = getenv("IONOTOMO_DETERMINISTIC")) c->deterministic = atoi(e) != 0;
#ifdef IONO_ABLATION
    if (const char *e = getenv("IONOTOMO_ADJ_ABLATE")) c->adj_mode |= atoi(e) & (4 | 8);     // timing only: WRONG results
#endif
    if (const char *e = getenv("IONOTOMO_FERMAT_COOP_MAX")) c->fermat_coop_max = atoll(e);
    if (const char *e = getenv("IONOTOMO_FERMAT_LM_LANES")) c->fermat_lm_lanes = atoi(e);
    if (const char *e = getenv("IONOTOMO_LM4_GROUPS")) c->lm4_groups = atoi(e);
    if (const char *e = getenv("IONOTOMO_FERMAT_LM_FEW_MIN")) c->fermat_lm_few_min = atoll(e);
    if (const char *e = getenv("IONOTOMO_FERMAT_COOP_RPW")) c->fermat_coop_rpw = std::min(8, std::max(1, atoi(e)));
    if (const char *e = getenv("IONOTOMO_FERMAT_LIN4_MAX")) c->fermat_lin4_max = atoll(e);
    if (const char *e = getenv("IONOTOMO_FERMAT_POLY_MAX")) c->fermat_poly_max = atoll(e);
    if (const char *e = getenv("IONOTOMO_FERMAT_POLY_RPW")) c->fermat_poly_rpw = atoi(e);
    if (const char *e = getenv("IONOTOMO_FERMAT_LIN4_RPW")) c->fermat_lin4_rpw = std::min(16, std::max(1, atoi(e)));
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
    if (c->d_F8) (void)hipFree(c->d_F8);
    if (c->d_FP) (void)hipFree(c->d_FP);
    if (c->d_nF8) (void)hipFree(c->d_nF8);
    if (c->d_G8) (void)hipFree(c->d_G8);
    if (c->d_fixgrid) (void)hipFree(c->d_fixgrid);
    if (c->d_LMw) (void)hipFree(c->d_LMw);
    if (c->d_Q4) (void)hipFree(c->d_Q4);
    if (c->d_freqs) (void)hipFree(c->d_freqs);
    if (c->d_rayw) (void)hipFree(c->d_rayw);
    plan_free(c);
    fplan_free(c);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->h_plan) (void)hipHostFree(c->h_plan);
    if (c->d_kern) (void)hipFree(c->d_kern);
    if (c->d_work) (void)hipFree(c->d_work);
    for (auto &wp : c->walk) {
        if (wp.d_starts) (void)hipFree(wp.d_starts);
        if (wp.d_cyc) (void)hipFree(wp.d_cyc);
    }
    if (c->d_chunk_counter) (void)hipFree(c->d_chunk_counter);
    (void)iono_comm_destroy(c);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return IONO_OK;
}

// The outgoing stream is drained so that work queued on it cannot race with launches on the new one; when it is a
// caller-owned handle that has since been destroyed the synchronisation fails, which must not wedge the ctx.
int iono_ctx_set_stream(iono_ctx *c, void *s) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    (void)hipSetDevice(c->device);
    if (hipStreamSynchronize(c->stream) != hipSuccess) (void)hipGetLastError();
    c->stream = (hipStream_t)s;
    return IONO_OK;
}

int iono_ctx_use_own_stream(iono_ctx *c) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    (void)hipSetDevice(c->device);
    if (hipStreamSynchronize(c->stream) != hipSuccess) (void)hipGetLastError();
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
    // validate every axis into locals first: a call that fails leaves the ctx exactly as it was
    bool ideal = true;
    int uniform[3];
    double inv_h[3], g0[3], glast[3];
    for (int a = 0; a < 3; ++a) {
        bool uni = true;
        const double h = (ax[a][n[a] - 1] - ax[a][0]) / (n[a] - 1);
        for (int i = 0; i + 1 < n[a]; ++i) {
            const double d = ax[a][i + 1] - ax[a][i];
            if (!(d > 0) || !std::isfinite(d)) return fail(c, IONO_ERR_ARG, "axes must be strictly increasing and finite");
            if (std::fabs(d - h) > 1e-6 * h) uni = false;
        }
        uniform[a] = uni ? 1 : 0;
        inv_h[a] = 1.0 / h;
        g0[a] = ax[a][0];
        glast[a] = ax[a][n[a] - 1];
        for (int i = 0; i < n[a]; ++i)
            if (std::fabs(ax[a][i] - (ax[a][0] + i * h)) > 2.5e-13 * h) ideal = false;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_axes) HIP_TRY(c, hipFree(c->d_axes));
    if (c->d_M) HIP_TRY(c, hipFree(c->d_M));
    if (c->d_nM) HIP_TRY(c, hipFree(c->d_nM));
    if (c->d_F8) HIP_TRY(c, hipFree(c->d_F8));
    if (c->d_FP) HIP_TRY(c, hipFree(c->d_FP));
    if (c->d_nF8) HIP_TRY(c, hipFree(c->d_nF8));
    if (c->d_G8) HIP_TRY(c, hipFree(c->d_G8));
    if (c->d_fixgrid) HIP_TRY(c, hipFree(c->d_fixgrid));
    c->d_fixgrid = nullptr;
    if (c->d_LMw) HIP_TRY(c, hipFree(c->d_LMw));
    if (c->d_Q4) HIP_TRY(c, hipFree(c->d_Q4));
    c->d_LMw = nullptr;
    c->d_Q4 = nullptr, c->Q4_valid = false;
    c->d_axes = nullptr;
    c->d_M = nullptr;
    c->d_nM = nullptr;
    c->d_F8 = c->d_G8 = c->d_FP = c->d_nF8 = nullptr;
    c->d_M_ext = nullptr;
    plan_free(c);
    fplan_free(c);
    c->F8_valid = c->FP_valid = false, c->FP_plan_serial = -1;
    c->nM_freq = c->nF8_freq = -1.0;
    c->nx = nx;
    c->ny = ny;
    c->nz = nz;
    c->storage = storage;
    c->ideal = ideal ? 1 : 0;
    for (int a = 0; a < 3; ++a) {
        c->uniform[a] = uniform[a], c->inv_h[a] = inv_h[a], c->g0[a] = g0[a], c->glast[a] = glast[a];
        c->c0[a] = n[a] >= 6 ? ax[a][2] : g0[a];
        c->clast[a] = n[a] >= 6 ? ax[a][n[a] - 3] : glast[a];
    }
    std::vector<double> cat;
    cat.insert(cat.end(), xv, xv + nx);
    cat.insert(cat.end(), yv, yv + ny);
    cat.insert(cat.end(), zv, zv + nz);
    HIP_TRY(c, hipMalloc((void **)&c->d_axes, cat.size() * sizeof(double)));
    HIP_TRY(c, hipMemcpy(c->d_axes, cat.data(), cat.size() * sizeof(double), hipMemcpyHostToDevice));
    const size_t esz = storage == IONO_F64 ? 8 : 4;
    // + one plane + one row + 2 zero elements: see trilinear_u (unclamped far-corner reads, weight 0)
    const size_t padded = (size_t)padded_count(c);
    HIP_TRY(c, hipMalloc(&c->d_M, padded * esz));
    HIP_TRY(c, hipMemset(c->d_M, 0, padded * esz));
    if (M) return iono_grid_set_values(c, M);
    return IONO_OK;
}

static int set_values_dev_impl(iono_ctx *c, const double *src_dev, int do_exp, double scale) {
    const int64_t n = ncells(c);
    c->nM_freq = c->nF8_freq = -1.0;
    c->F8_valid = c->FP_valid = false, c->FP_plan_serial = -1;
    c->Q4_valid = false;
    int rc = dispatch_storage(c, [&](auto *tag) {
        using GT = std::remove_pointer_t<decltype(tag)>;
        launch_map<SetValues<GT>>(c, ew_blocks(c, n), n, src_dev, (GT *)cur_values(c), do_exp, scale, c->d_flags + 1);
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
void *iono_grid_values_ptr(iono_ctx *c) { return c ? cur_values(c) : nullptr; }

int iono_grid_get_values(iono_ctx *c, double *out) {
    int rc = need_grid(c);
    if (rc) return rc;
    const int64_t n = ncells(c);
    DevBuf tmp(c);
    HIP_TRY(c, tmp.alloc((size_t)n * 8));
    dispatch_storage(c, [&](auto *tag) {
        using GT = std::remove_pointer_t<decltype(tag)>;
        launch_map<GetValues<GT>>(c, ew_blocks(c, n), n, (const GT *)cur_values(c), tmp.as<double>());
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

// room for the per-chunk cycle counts of the launch about to be made
static int walk_cycles_reserve(iono_ctx *c, iono_ctx::WalkPart &wp, int n_chunks, int units) {
    if (n_chunks > wp.cyc_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (wp.d_cyc) (void)hipFree(wp.d_cyc);
        wp.d_cyc = nullptr, wp.cyc_cap = 0;
        HIP_TRY(c, hipMalloc((void **)&wp.d_cyc, (size_t)n_chunks * 8));
        wp.cyc_cap = n_chunks;
    }
    wp.last_n = n_chunks, wp.last_units = units;
    return IONO_OK;
}

// Lekien-Marsden derivative fields of the current grid values (iono_cubic_kernels.h): rebuilt after every change.  Two layouts, each
// built when a kernel first asks for it: node-major records F8 (k_forward_straight_lm) and pair-major arrays FP (k_forward_bundle_lm).
// `for_plan` (pairs only): the launch that follows is the bundle kernel on the current forward plan -- rebuild only the node lines its
// windows hold (FwdPlan::d_xrange: a third of the bench grid; the pair arrays are then valid for THAT plan only, FP_plan_serial)
static int ensure_lm_fields(iono_ctx *c, bool pairs = false, bool for_plan = false) {
    const int64_t n = ncells(c), npad = padded_count(c);
    const bool restricted = pairs && for_plan && c->fplan.lm_all_fit && c->fplan.d_xrange;
    if (pairs && !c->FP_valid && restricted && c->d_FP && c->FP_plan_serial == c->fplan.serial) return IONO_OK;
    if (!pairs && !c->d_F8) {
        const size_t fb = (size_t)c->nx * LM_SI(c->ny, c->nz) * LM_NF * sizeof(double);
        HIP_TRY(c, hipMalloc((void **)&c->d_F8, fb));
        HIP_TRY(c, hipMemsetAsync(c->d_F8, 0, fb, c->stream));      // (pad nodes: never read by a valid sample)
    }
    if (pairs && !c->d_FP) {
        const size_t fb = (size_t)npad * 4 * 2 * sizeof(double);
        HIP_TRY(c, hipMalloc((void **)&c->d_FP, fb));
        HIP_TRY(c, hipMemsetAsync(c->d_FP, 0, fb, c->stream));      // (pad nodes: staged with a window's last levels, never weighed)
    }
    if (!(pairs ? c->FP_valid : c->F8_valid)) {
        const int64_t lines = (int64_t)c->ny * c->nz * LM_XSEG;
        dispatch_storage(c, [&](auto *tag) {
            using GT = std::remove_pointer_t<decltype(tag)>;
            if (pairs)
                hipLaunchKernelGGL((k_lm_fields_yx<true, GT>), dim3(ew_blocks(c, lines)), dim3(256), 0, c->stream, (const GT *)cur_values(c), c->d_FP,
                                   c->nx, c->ny, c->nz, npad, restricted ? (const int2 *)c->fplan.d_xrange : (const int2 *)nullptr);
            else
                hipLaunchKernelGGL((k_lm_fields_yx<false, GT>), dim3(ew_blocks(c, lines)), dim3(256), 0, c->stream, (const GT *)cur_values(c), c->d_F8,
                                   c->nx, c->ny, c->nz, npad, (const int2 *)nullptr);
            return IONO_OK;
        });
        HIP_TRY(c, hipGetLastError());
        if (!pairs) c->F8_valid = true;
        else if (restricted) c->FP_plan_serial = c->fplan.serial;
        else c->FP_valid = true;
    }
    return IONO_OK;
}
// the same records of the refractive-index nodes d_nM (built for `frequency`): what the tricubic tracer on ideal grids reads
static int ensure_n_fields(iono_ctx *c, double frequency) {
    const int64_t n = ncells(c);
    if (!c->d_nF8) {
        const size_t fb = (size_t)c->nx * LM_SI(c->ny, c->nz) * LM_NF * sizeof(double);
        HIP_TRY(c, hipMalloc((void **)&c->d_nF8, fb));
        HIP_TRY(c, hipMemsetAsync(c->d_nF8, 0, fb, c->stream));
        c->nF8_freq = -1.0;
    }
    if (c->nF8_freq != frequency) {
        const int64_t lines = (int64_t)c->ny * c->nz * LM_XSEG;
        hipLaunchKernelGGL((k_lm_fields_yx<false, double>), dim3(ew_blocks(c, lines)), dim3(256), 0, c->stream, (const double *)c->d_nM, c->d_nF8,
                           c->nx, c->ny, c->nz, padded_count(c), (const int2 *)nullptr);
        HIP_TRY(c, hipGetLastError());
        c->nF8_freq = frequency;
    }
    return IONO_OK;
}
// Walk of the trilinear forward kernels when IONOTOMO_WALK does not say: one contiguous chunk of rays per wave while the array
// they read fits the 256 MiB Infinity Cache (256^3 f64: 0.244 ms against 0.252 interleaved); beyond it, all the waves of an XCD
// interleaved in that XCD's eighth of the rays, so that what is in flight on an XCD shares lines in ITS L2 (512^3 f64, 1 GiB:
// 0.68 against 0.72 ms; float32 block layout at 512^3, 2 GiB: 0.53 against 0.68 ms).
// A caller-supplied walk order means "neighbours in this order are nearly the same ray": interleaved too, so that they run at
// the same time on neighbouring waves and share lines in the L1 (RayEngine.coherent_order: 0.212 against 0.224 ms).
static int forward_walk_mode(const iono_ctx *c, uint64_t array_bytes, const int *order) {
    (void)c;
    return order || array_bytes > ((uint64_t)256 << 20) ? 2 : 0;
}
// Forward mapping on ideal-uniform grids without a bundle plan: lanes = samples of one ray (k_forward_straight_u).
// IONOTOMO_HYBRID_MIN=65 forces lanes = samples even on a planned geometry (A/B; results agree to rounding).
// the bundle plan serves this launch: same ray arrays, R, tmax, Ns, and it fills the chip / is good (see the forward dispatch)
static bool fplan_serves(const iono_ctx *c, const double *o, const double *d, int64_t R, double tmax, int Ns, int storage = IONO_F64) {
    const iono_ctx::FwdPlan &fp = c->fplan;
    // (WHICH bundles are worth a workgroup -- all, those of >= T rays, none -- was decided when the plan was made: iono_forward_plan_dev;
    //  IONOTOMO_HYBRID_MIN=65: none, A/B)
    return c->storage == storage && fp.R == R && fp.o_key == o && fp.d_key == d && fp.Ns == Ns && fp.tmax == tmax && ideal_path_ok(c) &&
           fp.nb > 0 && (fp.fit_fraction >= 0.5 || fp.forced);
}
// fast tricubic tier: ideal-uniform axes, weights in LDS, 32-bit-safe field array (IONOTOMO_VARIANT=4 forces the general tier)
static bool cubic_fast_ok(const iono_ctx *c, int Ns) {
    // (field records of 64 B per node, addressed with 32-bit byte offsets from the column bases)
    return ideal_path_ok(c, Ns) && c->variant != 4 && c->nx >= 6 && c->ny >= 6 && c->nz >= 6 &&
           (uint64_t)c->nx * LM_SI(c->ny, c->nz) * LM_NF * sizeof(double) < ((uint64_t)1 << 32);
}

// ================================================================================================
// ONE dispatch table (include/ionotomo_hip.h: iono_dispatch_facts, iono_dispatch_name): every launcher below fills the facts of its
// launch (facts_of) and launches what the pick_* function of its operation answers; iono_dispatch_name turns the same answer into text.
// ================================================================================================
enum FwdKernel { FK_BUNDLE_F32, FK_Q4, FK_BUNDLE, FK_U, FK_BUNDLE_LM, FK_LM, FK_FAST, FK_GENERAL };
enum AdjKernel { AK_BINNED, AK_BINNED_FIX, AK_BINNED_LM4, AK_TILE, AK_TILE_CUBIC, AK_GENERAL, AK_REFUSED };
enum TraceKernel { TK_POLY, TK_LIN4, TK_RAYS, TK_LM8, TK_LM2, TK_COOP };
enum FermatKernel { FT_LM8, FT_LM2, FT_RAYS };
enum PhaseFwdKernel { PF_BUNDLE, PF_U, PF_GENERAL };
enum PhaseAdjKernel { PA_BINNED, PA_TILE, PA_GENERAL };

static FwdKernel pick_forward(const iono_dispatch_facts &f) {
    const bool lin = f.interp_kind == IONO_INTERP_TRILINEAR, f32 = f.storage == IONO_F32, ideal = f.tier == 2;
    if (lin && f32 && f.fwd_bundles > 0) return FK_BUNDLE_F32;      // float32 fast mode: served bundles (+ the tail: Q4 or lanes = samples)
    if (lin && f32 && ideal && f.q4_ok) return FK_Q4;               // float32 storage, no plan: 2 x 2 corner blocks
    if (lin && !f32 && f.fwd_bundles > 0) return FK_BUNDLE;         // the headline: windows in LDS (+ the tail: lanes = samples)
    if (lin && ideal) return FK_U;                                  // lanes = samples on ideal-uniform axes
    if (!lin && f.cubic_fast && f.fwd_bundles > 0) return FK_BUNDLE_LM;
    if (!lin && f.cubic_fast) return FK_LM;                         // lanes = samples on Lekien-Marsden records
    if (lin && f.tier >= 1) return FK_FAST;                         // table-uniform axes
    return FK_GENERAL;                                              // any axes: binary search per sample (trilinear / 216-tap tricubic)
}
static AdjKernel pick_adjoint(const iono_dispatch_facts &f) {
    const bool lin = f.interp_kind == IONO_INTERP_TRILINEAR, planned = f.adj_planned && f.variant != 2;
    if (planned && lin) return f.deterministic ? AK_BINNED_FIX : AK_BINNED;
    const bool lm4 = planned && !lin && f.cubic_fast && f.adj_tiles;
    if (f.deterministic && !lm4) return AK_REFUSED;                 // fixed point serves the planned back-projections only
    if (lin && f.tier == 2 && f.variant != 2) return AK_TILE;       // ray-stationary LDS tiles
    if (!lin && f.cubic_fast && f.variant != 2) return lm4 ? AK_BINNED_LM4 : AK_TILE_CUBIC;
    return AK_GENERAL;
}
static int fermat_lanes_of(const iono_dispatch_facts &f) {
    if (f.fermat_lm_lanes == 8 || f.fermat_lm_lanes == 2) return f.fermat_lm_lanes;      // (4 and 1 lanes were measured too: profiles/r05_ab_fermat_lanes.json)
    return f.R >= f.fermat_lm_few_min ? 2 : 8;
}
static TraceKernel pick_tracer(const iono_dispatch_facts &f) {
    const bool lin = f.interp_kind == IONO_INTERP_TRILINEAR;
    if (lin && f.ideal_axes && f.variant != 3 && f.R <= f.fermat_poly_max) return TK_POLY;
    if (lin && f.variant != 3 && f.R <= f.fermat_lin4_max && f.axes_bytes <= 48 * 1024) return TK_LIN4;
    if (lin) return TK_RAYS;
    if (f.variant == 3 || f.R > f.fermat_coop_max) return TK_RAYS;      // lanes = rays: enough rays to fill the chip without splitting them
    if (f.ideal_axes && f.cubic_records) return fermat_lanes_of(f) == 8 ? TK_LM8 : TK_LM2;
    return TK_COOP;
}
// the fused forward through a tricubic index runs on k_fermat_tec_lm (ideal-uniform axes, 32-bit-safe record array, float64); its
// transpose too for bending rays at two lanes per ray -- the large batches, where the alternative is a ray tensor of 32 R Ns bytes and
// a back-projection with one hardware atomic per corner; small batches keep k_trace_fermat_lm + k_adjoint_rays (engine.py)
static FermatKernel pick_fermat(const iono_dispatch_facts &f, bool transpose) {
    const int lanes = fermat_lanes_of(f);
    const bool lm = f.interp_kind == IONO_INTERP_TRICUBIC && f.storage == IONO_F64 && f.ideal_axes && f.cubic_records && f.variant != 3 &&
                    f.R <= f.fermat_coop_max && (!transpose || (f.bend && lanes == 2));
    return lm ? (lanes == 8 ? FT_LM8 : FT_LM2) : FT_RAYS;
}
static PhaseFwdKernel pick_phase_forward(const iono_dispatch_facts &f) {
    return f.fwd_bundles > 0 && f.storage == IONO_F64 ? PF_BUNDLE : f.tier == 2 ? PF_U : PF_GENERAL;
}
static PhaseAdjKernel pick_phase_adjoint(const iono_dispatch_facts &f) {
    return f.adj_planned && f.variant != 2 ? PA_BINNED : f.tier == 2 && f.variant != 2 ? PA_TILE : PA_GENERAL;
}

// the facts of a launch on this context (o, d null: no plan matches)
static iono_dispatch_facts facts_of(const iono_ctx *c, int op, const double *o, const double *d, int64_t R, double tmax, int Ns, int kind,
                                    int kind_ne, int bend) {
    iono_dispatch_facts f;
    memset(&f, 0, sizeof(f));
    f.storage = c->storage, f.variant = c->variant, f.deterministic = c->deterministic ? 1 : 0;
    f.tier = ideal_path_ok(c, Ns) ? 2 : fast_path_ok(c) ? 1 : 0;
    f.cubic_fast = cubic_fast_ok(c, Ns) ? 1 : 0, f.cubic_records = cubic_fast_ok(c, 2) ? 1 : 0;
    f.ideal_axes = view(c).ideal ? 1 : 0;
    f.q4_ok = (uint64_t)padded_count(c) * 16 < ((uint64_t)1 << 32) ? 1 : 0;
    f.interp_kind = kind, f.ne_kind = kind_ne, f.bend = bend, f.Ns = Ns, f.R = R;
    const bool straight = op == IONO_OP_FORWARD || op == IONO_OP_PHASE_FORWARD;
    // (the tricubic bundle kernel and the phase observable read float64 fields: their plans serve float64 storage only)
    const int plan_storage = op == IONO_OP_FORWARD && kind == IONO_INTERP_TRILINEAR ? c->storage : IONO_F64;
    if (straight && o && d && fplan_serves(c, o, d, R, tmax, Ns, plan_storage)) f.fwd_bundles = c->fplan.nb, f.fwd_tail = c->fplan.n_rest;
    const iono_ctx::AdjPlan &pl = c->plan;
    const int adj_kind = op == IONO_OP_PHASE_ADJOINT ? IONO_INTERP_TRILINEAR : kind;
    if ((op == IONO_OP_ADJOINT || op == IONO_OP_PHASE_ADJOINT) && o && pl.R == R && pl.o_key == o && pl.d_key == d && pl.Ns == Ns && pl.tmax == tmax &&
        pl.kind == adj_kind)
        f.adj_planned = 1, f.adj_tiles = pl.tile_n[2] > 0 ? 1 : 0, f.adj_seg_lanes = pl.segl;
    f.axes_bytes = (c->nx + c->ny + c->nz) * 8;
    f.fermat_lm_lanes = c->fermat_lm_lanes, f.fermat_lm_few_min = c->fermat_lm_few_min, f.fermat_poly_max = c->fermat_poly_max;
    f.fermat_coop_max = c->fermat_coop_max;
    // (a grid that is not ideal-uniform keeps the general right-hand side in the lanes = rays kernel: crossover ~150k rays as before)
    f.fermat_lin4_max = f.ideal_axes || c->fermat_lin4_max != 4096 ? c->fermat_lin4_max : 131072;
    return f;
}

static std::string dispatch_text(const iono_dispatch_facts &f, int op) {
    const char *T = f.storage == IONO_F32 ? "float" : "double";
    const std::string lin_or_cubic = f.interp_kind == IONO_INTERP_TRILINEAR ? "IONO_INTERP_TRILINEAR" : "IONO_INTERP_TRICUBIC";
    const std::string segl = std::to_string(f.adj_seg_lanes ? f.adj_seg_lanes : 16);
    auto tail = [&](const std::string &k) { return f.fwd_tail > 0 ? " + " + k : std::string(); };
    switch (op) {
    case IONO_OP_FORWARD:
        switch (pick_forward(f)) {
        case FK_BUNDLE_F32: return "k_forward_bundle_f32" + tail(f.q4_ok ? "k_forward_straight_q4" : "k_forward_straight_u<float>");
        case FK_Q4: return "k_forward_straight_q4";
        case FK_BUNDLE: return "k_forward_bundle<0>" + tail("k_forward_straight_u<double>");
        case FK_U: return std::string("k_forward_straight_u<") + T + ">";
        case FK_BUNDLE_LM: return "k_forward_bundle_lm" + tail("k_forward_straight_lm");
        case FK_LM: return "k_forward_straight_lm";
        case FK_FAST: return std::string("k_forward_straight_fast<") + T + ">";
        default: return std::string("k_forward_straight<") + T + ", " + lin_or_cubic + ">";
        }
    case IONO_OP_ADJOINT:
        switch (pick_adjoint(f)) {
        case AK_BINNED: return "k_adjoint_binned<AT, 0, double, " + segl + ", false>";
        case AK_BINNED_FIX: return "k_adjoint_binned<double, 0, double, " + segl + ", true> + FixConvert";
        case AK_BINNED_LM4: return "2 x k_adjoint_binned_lm4<" + segl + ", true> + k_lm_fold_{z,y,x}_tiles";
        case AK_TILE: return "k_adjoint_straight_tile<AT, MODE, 4, false>";
        case AK_TILE_CUBIC: return "8 x k_adjoint_straight_tile<double, MODE, 4, true> + k_lm_fold_{z,y,x}";
        case AK_REFUSED: return "refused: deterministic mode serves the planned back-projections only";
        default: return "k_adjoint_straight<AT, MODE, " + lin_or_cubic + ">";
        }
    case IONO_OP_TRACE:
        switch (pick_tracer(f)) {
        case TK_POLY: return std::string("k_trace_fermat_poly<") + (f.bend ? "true" : "false") + ">";
        case TK_LIN4: return std::string("k_trace_fermat_lin4<") + (f.bend ? "true" : "false") + ">";
        case TK_RAYS: return "k_trace_fermat<" + lin_or_cubic + ", " + (f.bend ? "true" : "false") + ">";
        case TK_LM8: return std::string("k_trace_fermat_lm<") + (f.bend ? "true" : "false") + ", 8>";
        case TK_LM2: return std::string("k_trace_fermat_lm<") + (f.bend ? "true" : "false") + ", 2>";
        default: return std::string("k_trace_fermat_coop<") + (f.bend ? "true" : "false") + ">";
        }
    case IONO_OP_FERMAT_FORWARD:
    case IONO_OP_FERMAT_ADJOINT: {
        const bool adj = op == IONO_OP_FERMAT_ADJOINT;
        const FermatKernel k = pick_fermat(f, adj);
        const std::string b = f.bend ? "true" : "false";
        if (k == FT_RAYS) return "k_fermat_tec<" + lin_or_cubic + ", " + b + ", " + (adj ? "true" : "false") + ">";
        return "k_fermat_tec_lm<" + b + ", " + (k == FT_LM8 ? "8" : "2") + (adj ? ", true>" : ", false>");
    }
    case IONO_OP_PHASE_FORWARD:
        switch (pick_phase_forward(f)) {
        case PF_BUNDLE: return "k_forward_bundle<NF>" + tail(std::string("k_forward_phase_u<") + T + ", NF>");
        case PF_U: return std::string("k_forward_phase_u<") + T + ", NF>";
        default: return std::string("k_forward_phase_straight<") + T + ", false>";
        }
    case IONO_OP_PHASE_ADJOINT:
        switch (pick_phase_adjoint(f)) {
        case PA_BINNED: return std::string("k_adjoint_binned<double, NF, ") + T + ", " + segl + ">";
        case PA_TILE: return std::string("k_adjoint_straight_tile<double, 0, 4, false, true, ") + T + ">";
        default: return std::string("k_adjoint_phase_straight<") + T + ", double>";
        }
    }
    return "?";
}

// ---- bundle plan of the forward (k_forward_bundle) ---------------------------------------------------------------------------
// Geometry only, once per inversion: (1) the rays sorted along a 4-D Morton curve of foot and end point in grid cells (neighbours
// in the walk nearly coincide all the way up); (2) the walk cut greedily into bundles of <= 64 rays whose window -- the columns
// the bundle touches in any B_KC consecutive samples, bounded through the extents at the two ends (positions are linear in the
// sample index, so the extent of a bundle is convex in it) + the drift of its steepest ray -- fits the LDS image of one wave;
// (3) the exact window of every (bundle, chunk), by the device (k_bundle_windows).  Keys, sort (rocPRIM radix sort) and windows run
// on the device; the host only cuts the sorted walk (one sequential pass over R 32-byte ray summaries).  No plan (non-uniform
// axes, float32 storage): the other forward kernels serve the launch.  Any parity of nz (round 5: with an odd nz the 16-byte
// window loads start on 8-byte boundaries, as the lanes = samples kernel's always have).
int iono_forward_plan_clear(iono_ctx *c) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    fplan_free(c);
    return IONO_OK;
}

static int plan_pinned(iono_ctx *c, size_t bytes, char **out) {
    if (c->plan_pinned_cap < bytes) {
        if (c->h_plan) (void)hipHostFree(c->h_plan);
        c->h_plan = nullptr, c->plan_pinned_cap = 0;
        const size_t cap = bytes + bytes / 4 + 4096;
        HIP_TRY(c, hipHostMalloc((void **)&c->h_plan, cap, hipHostMallocDefault));
        c->plan_pinned_cap = cap;
    }
    *out = c->h_plan;
    return IONO_OK;
}

// window set of the tricubic bundle kernel (chunks of BL_KC samples, BL_LEV levels of 16-byte nodes): computed on its first launch,
// so that a trilinear inversion does not pay for it (0.1 ms of kernel + 4.8 MB at the bench shape)
static int ensure_lm_windows(iono_ctx *c) {
    iono_ctx::FwdPlan &fp = c->fplan;
    if (fp.R < 0 || fp.nchunks_lm > 0) return IONO_OK;
    const int nchunks_lm = (fp.Ns + BL_KC - 1) / BL_KC;
    HIP_TRY(c, plan_reserve(fp.d_win_lm, fp.cap_win_lm, (size_t)fp.nb * nchunks_lm * sizeof(uint4)));
    // (+ the number of windows that fit and the node lines the windows hold: what a field rebuild for this plan must cover)
    const int64_t nlines = (int64_t)c->ny * c->nz;
    HIP_TRY(c, plan_reserve(fp.d_xrange, fp.cap_xrange, (size_t)nlines * sizeof(int2) + 32));
    unsigned long long *d_fits = (unsigned long long *)(fp.d_xrange + nlines);
    HIP_TRY(c, hipMemsetAsync(d_fits, 0, 3 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL((k_bundle_windows<BL_KC, BL_LEV, 2 * BL_CPL, false>), dim3(fp.nb), dim3(64), 0, c->stream, view(c),
                       (const double *)fp.o_key, (const double *)fp.d_key, fp.d_order, fp.d_bstart, fp.nb, fp.tmax, fp.Ns, nchunks_lm,
                       fp.d_win_lm, d_fits);
    HIP_TRY(c, hipMemsetAsync(fp.d_xrange, 0xff, (size_t)nlines * sizeof(int2), c->stream));
    const int64_t nwin = (int64_t)fp.nb * nchunks_lm;
    hipLaunchKernelGGL(k_lm_touch_lines, dim3(ew_blocks(c, nwin * 2 * BL_CPL * BL_LEV)), dim3(256), 0, c->stream, (const uint4 *)fp.d_win_lm, nwin, c->ny,
                       c->nz, BL_LEV, 2 * BL_CPL * BL_LEV, fp.d_xrange);
    HIP_TRY(c, hipGetLastError());
    unsigned long long h_fits[3] = {0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(h_fits, d_fits, sizeof(h_fits), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // every (bundle, chunk) of a bundle with a valid ray has a window record; bundles without one have none and read nothing.  A rebuild
    // may be restricted to the recorded lines when every record BOUNDS its chunk's reads (a window too wide for the image still does:
    // its chunk then reads the same nodes from memory; one that needs more levels than the record can say does not)
    fp.lm_all_fit = h_fits[1] > 0 && h_fits[2] == h_fits[1];
    fp.nchunks_lm = nchunks_lm;
    return IONO_OK;
}

int iono_forward_plan_dev(iono_ctx *c, const double *o, const double *d, int64_t R, double tmax, int Ns) {
    int rc = check_common(c, R, Ns, IONO_INTERP_TRILINEAR, 0);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    iono_ctx::FwdPlan &fp = c->fplan;
    fp.R = -1, fp.nb = 0, fp.o_key = fp.d_key = nullptr;
    if (R == 0 || R > (int64_t)INT32_MAX / 2 || !ideal_path_ok(c) || !o || !d ||
        (int64_t)c->ny * c->nz * 8 > (int64_t)B_MAX_PLANE)      // (the kernel's 32-bit offsets inside a window: iono_forward_kernels.h)
        return IONO_OK;
    // keys and ray summaries on the device, device radix sort; only the sorted 32-byte summaries travel to the host for the cut
    DevBuf scratch(c);
    const size_t off_k0 = 0, off_k1 = off_k0 + (size_t)R * 8, off_i0 = off_k1 + (size_t)R * 8, off_i1 = off_i0 + (size_t)R * 4,
                 off_r0 = (off_i1 + (size_t)R * 4 + 31) & ~(size_t)31, off_r1 = off_r0 + (size_t)R * sizeof(BundleSummary),
                 off_h = off_r1 + (size_t)R * sizeof(BundleSummary), off_tmp = off_h + (size_t)R * sizeof(uint2);
    size_t tmp_bytes = 0;
    HIP_TRY(c, rocprim::radix_sort_pairs(nullptr, tmp_bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (int *)nullptr,
                                         (int *)nullptr, (size_t)R, 0, 64, c->stream));
    HIP_TRY(c, scratch.alloc(off_tmp + tmp_bytes + 256));
    char *sb = scratch.as<char>();
    unsigned long long *k0 = (unsigned long long *)(sb + off_k0), *k1 = (unsigned long long *)(sb + off_k1);
    int *i0 = (int *)(sb + off_i0), *i1 = (int *)(sb + off_i1);
    BundleSummary *r0 = (BundleSummary *)(sb + off_r0), *r1 = (BundleSummary *)(sb + off_r1);
    uint2 *hash_by_ray = (uint2 *)(sb + off_h);
    hipLaunchKernelGGL(k_bundle_keys, dim3(ew_blocks(c, R)), dim3(256), 0, c->stream, view(c), o, d, R, tmax, Ns, k0, i0, r0, hash_by_ray);
    HIP_TRY(c, rocprim::radix_sort_pairs(sb + off_tmp, tmp_bytes, k0, k1, i0, i1, (size_t)R, 0, 64, c->stream));
    launch_map<BundleGather>(c, ew_blocks(c, R), R, r0, i1, r1);
    HIP_TRY(c, hipGetLastError());
    // (pinned: [R summaries | R walk positions])
    char *hp = nullptr;
    rc = plan_pinned(c, (size_t)R * (sizeof(BundleSummary) + 2 * sizeof(int)) + 64, &hp);
    if (rc) return rc;
    const BundleSummary *hr = (const BundleSummary *)hp;
    int *perm = (int *)(hp + (size_t)R * sizeof(BundleSummary));
    HIP_TRY(c, hipMemcpyAsync(hp, r1, (size_t)R * sizeof(BundleSummary), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, plan_reserve(fp.d_order, fp.cap_order, (size_t)R * sizeof(int)));
    HIP_TRY(c, plan_reserve(fp.d_rhash, fp.cap_rhash, (size_t)R * sizeof(uint2)));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // The cut: walk the sorted rays; a bundle takes the next rays that keep its window within the image, looking up to
    // B_LOOKAHEAD rejected rays ahead (the Morton curve jumps: a ray that does not fit now often belongs to a later bundle, while
    // the ones behind it still fit this one: 4 838 -> ~4 560 bundles at the bench shape).  perm = walk positions, bundle by bundle.
    // The walk is cut in up to 16 stretches by as many host threads (a bundle never crosses a stretch: 15 bundles in 4 600 end early);
    // the sequential cut was 1.5 of the 2.8 ms a plan took.
    const int nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(16, (int64_t)std::thread::hardware_concurrency()), R / 8192));
    std::vector<std::vector<int>> starts((size_t)nthreads);
    std::vector<unsigned char> used((size_t)R, 0);
    auto cut = [&](int t) {
        const int64_t lo = R * t / nthreads, hi = R * (t + 1) / nthreads;
        std::vector<int> &bs = starts[(size_t)t];
        bs.reserve((size_t)(hi - lo) / 32 + 2);
        int64_t np = lo;              // walk positions filed so far (a stretch permutes its own positions)
        for (int64_t i = lo; i < hi; ++i) {
            if (used[(size_t)i]) continue;
            const BundleSummary &h0 = hr[(size_t)i];
            const bool valid = h0.adx >= 0.0f;
            float x0lo = h0.fx0, x0hi = h0.fx0, y0lo = h0.fy0, y0hi = h0.fy0, xelo = h0.fxe, xehi = h0.fxe, yelo = h0.fye, yehi = h0.fye, zlo = h0.fz0,
                  zhi = h0.fz0, ax = h0.adx, ay = h0.ady, az = h0.dz;
            bs.push_back((int)np);      // (a single ray whose own window does not fit still gets a bundle: its chunks take the direct loads)
            perm[np++] = (int)i;
            used[(size_t)i] = 1;
            int cnt = 1, miss = 0;
            for (int64_t j = i + 1; j < hi && cnt < 64 && miss <= B_LOOKAHEAD; ++j) {
                if (used[(size_t)j]) continue;
                const BundleSummary &h = hr[(size_t)j];
                if ((h.adx >= 0.0f) != valid) break;               // rays that leave the grid sit at the end of the walk, in bundles of their own
                bool ok = true;
                float nx0lo = x0lo, nx0hi = x0hi, ny0lo = y0lo, ny0hi = y0hi, nxelo = xelo, nxehi = xehi, nyelo = yelo, nyehi = yehi, nzlo = zlo, nzhi = zhi,
                      nax = ax, nay = ay, naz = az;
                if (valid) {
                    nx0lo = std::min(x0lo, h.fx0), nx0hi = std::max(x0hi, h.fx0), ny0lo = std::min(y0lo, h.fy0), ny0hi = std::max(y0hi, h.fy0);
                    nxelo = std::min(xelo, h.fxe), nxehi = std::max(xehi, h.fxe), nyelo = std::min(yelo, h.fye), nyehi = std::max(yehi, h.fye);
                    nzlo = std::min(zlo, h.fz0), nzhi = std::max(zhi, h.fz0), nax = std::max(ax, h.adx), nay = std::max(ay, h.ady), naz = std::max(az, h.dz);
                    const int wxb = (int)std::floor(std::max(nx0hi - nx0lo, nxehi - nxelo) + nax * (B_KC - 1) + 1e-3f) + 3;
                    const int wyb = (int)std::floor(std::max(ny0hi - ny0lo, nyehi - nyelo) + nay * (B_KC - 1) + 1e-3f) + 3;
                    const int nlb = (int)std::floor((nzhi - nzlo) + naz * (B_KC - 1) + 1e-3f) + 4;
                    ok = wxb * wyb <= B_CAPCOLS && wyb <= B_MAXWY && nlb <= B_LEV;
                }
                if (!ok) {
                    ++miss;
                    continue;
                }
                x0lo = nx0lo, x0hi = nx0hi, y0lo = ny0lo, y0hi = ny0hi, xelo = nxelo, xehi = nxehi, yelo = nyelo, yehi = nyehi;
                zlo = nzlo, zhi = nzhi, ax = nax, ay = nay, az = naz;
                perm[np++] = (int)j;
                used[(size_t)j] = 1;
                ++cnt;
            }
        }
    };
    if (nthreads == 1) {
        cut(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t) pool.emplace_back(cut, t);
        for (std::thread &th : pool) th.join();
    }
    std::vector<int> bstart;
    for (const std::vector<int> &bs : starts) bstart.insert(bstart.end(), bs.begin(), bs.end());
    bstart.push_back((int)R);
    // Hybrid dispatch (round 6): a bundle costs its workgroup the same whether it holds 64 rays or 3 (a lane per ray), while the
    // lanes = samples kernel costs per RAY -- the two meet at ~25 rays per bundle (21 ns per bundle against 0.8 ns per ray at the bench
    // shape).  So the choice is made per BUNDLE, not per launch: bundles of >= hybrid_min rays move to the front of the walk and are
    // what the bundle kernels serve; the rays of the others follow, still in Morton order, and every planned launch hands them to the
    // lanes = samples kernel of the same interpolant (its `order` argument = that tail of the walk).  A batch of a few timesteps --
    // what the reference's pipeline forms per coherence window -- thus keeps its well-filled bundles instead of losing the plan.
    const int nb_all = (int)bstart.size() - 1;
    for (int64_t &h : fp.hist) h = 0;
    for (int b = 0; b < nb_all; ++b) ++fp.hist[std::min(bstart[(size_t)b + 1] - bstart[(size_t)b], 64)];
    // The threshold T (serve bundles of >= T rays; 1: all, 65: none): forced (IONOTOMO_HYBRID_MIN), else the T
    // with the smallest modelled time.  Model, fitted to the round-6 sweeps on MI355X (profiles/r06_coherence_sweep_*_ab.json; 256^3 float64,
    // Ns = 257, scaled by Ns): a bundle launch of n workgroups takes max(22, 8 + 0.019 n) us (one workgroup lives ~20 us; 570 bundles: 22.5 us; 4 597 bundles:
    // 95 us), a lanes = samples launch of r rays 5.5 + 0.00078 r us (2 604 rays: 7.7 us; 260 400: 208 us).  Checked against the
    // measurements: 42 directions x 1 / 4 / 16 / 100 timesteps of 62 stations -> none / none / all / all, as measured fastest; half the
    // bench rays + as many scattered ones -> T = 24: 0.20 ms against 0.57 (all bundles) and 0.34 (none).
    fp.forced = c->hybrid_min > 0;
    int hmin = c->hybrid_min;
    {
        const double su = (double)Ns / 257.0;
        auto t_bundles = [&](int64_t n) { return n ? std::max(22.0 * (0.25 + 0.75 * su), 8.0 + 0.019 * su * (double)n) : 0.0; };
        // (a lanes = samples launch BEHIND a bundle launch costs ~10 us before its first ray, not 5.5: the bundle kernel's last
        //  workgroups drain first -- 1 772 tail rays: +10.5 us, 9 299: +15.6 us at the bench shape)
        auto t_rays = [&](int64_t r, bool second) { return r ? (second ? 10.0 : 5.5) + 0.00078 * su * (double)r : 0.0; };
        const int cand[10] = {1, 2, 4, 8, 12, 16, 24, 32, 48, 65};
        double best = -1.0;
        for (int T : cand) {
            int64_t nbT = 0, restT = 0;
            for (int n = 1; n <= 64; ++n) {
                if (n >= T) nbT += fp.hist[n];
                else restT += (int64_t)n * fp.hist[n];
            }
            const double t = t_bundles(nbT) + t_rays(restT, nbT > 0);
            if (T == 1) fp.model_us[0] = t;
            if (T == 65) fp.model_us[2] = t;
            if (best < 0 || t < best) {
                best = t;
                if (!fp.forced) hmin = T;
            }
            if (T == hmin) fp.model_us[1] = t;
        }
    }
    int nb = 0;
    int64_t n_planned = 0;
    for (int b = 0; b < nb_all; ++b) {
        const int cnt = bstart[(size_t)b + 1] - bstart[(size_t)b];
        if (cnt >= hmin) ++nb, n_planned += cnt;
    }
    if (nb != nb_all) {
        int *perm2 = perm + R;
        std::vector<int> bs2;
        bs2.reserve((size_t)nb + 1);
        int64_t np = 0;
        for (int pass = 0; pass < 2; ++pass) {
            for (int b = 0; b < nb_all; ++b) {
                const int lo = bstart[(size_t)b], cnt = bstart[(size_t)b + 1] - lo;
                if ((cnt >= hmin) != (pass == 0)) continue;
                if (pass == 0) bs2.push_back((int)np);
                std::copy(perm + lo, perm + lo + cnt, perm2 + np);
                np += cnt;
            }
            if (pass == 0) bs2.push_back((int)np);
        }
        perm = perm2;
        bstart.swap(bs2);
    }
    // order[q] = sorted index at walk position perm[q]
    {
        int *d_perm = i0;                                         // (scratch: the unsorted index array is no longer needed)
        HIP_TRY(c, hipMemcpyAsync(d_perm, perm, (size_t)R * sizeof(int), hipMemcpyHostToDevice, c->stream));
        launch_map<BundlePermute>(c, ew_blocks(c, R), R, i1, d_perm, fp.d_order, hash_by_ray, fp.d_rhash);
        HIP_TRY(c, hipGetLastError());
    }
    const int nchunks = (Ns + B_KC - 1) / B_KC;
    fp.nb_all = nb_all, fp.split_min = hmin, fp.n_planned = n_planned, fp.n_rest = R - n_planned;
    if (nb == 0) {      // no bundle worth a workgroup: the walk order alone is kept (the back-projection plan walks the rays in it)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        fp.o_key = o, fp.d_key = d, fp.R = R, fp.Ns = Ns, fp.tmax = tmax, fp.nb = 0, fp.nchunks = nchunks, fp.nchunks_lm = 0;
        fp.fit_fraction = 0.0, fp.serial = ++c->fplan_counter, fp.lm_all_fit = false;
        return IONO_OK;
    }
    HIP_TRY(c, plan_reserve(fp.d_bstart, fp.cap_bstart, bstart.size() * sizeof(int)));
    HIP_TRY(c, plan_reserve(fp.d_win, fp.cap_win, (size_t)nb * nchunks * sizeof(uint4)));
    HIP_TRY(c, hipMemcpyAsync(fp.d_bstart, bstart.data(), bstart.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    unsigned long long *d_fits = (unsigned long long *)k0;        // (scratch: the key arrays are no longer needed)
    HIP_TRY(c, hipMemsetAsync(d_fits, 0, 3 * sizeof(unsigned long long), c->stream));
    if (c->storage == IONO_F32)          // float32 storage: the fast mode's window records (12 levels of 4-byte values per column)
        hipLaunchKernelGGL(k_bundle_windows_f32, dim3(nb), dim3(64), 0, c->stream, view(c), o, d, fp.d_order, fp.d_bstart, nb, tmax, Ns, nchunks,
                           fp.d_win, d_fits);
    else
        hipLaunchKernelGGL((k_bundle_windows<B_KC, B_LEV, B_MAXWY, true, true>), dim3(nb), dim3(64), 0, c->stream, view(c), o, d, fp.d_order, fp.d_bstart,
                           nb, tmax, Ns, nchunks, fp.d_win, d_fits);
    HIP_TRY(c, plan_reserve(fp.d_brec, fp.cap_brec, (size_t)nb * 64 * sizeof(BundleRec)));
    HIP_TRY(c, plan_reserve(fp.d_bhash, fp.cap_bhash, (size_t)nb * 64 * sizeof(uint2)));
    hipLaunchKernelGGL((k_bundle_records<false>), dim3(nb), dim3(64), 0, c->stream, view(c), o, d, fp.d_order, fp.d_bstart, nb, tmax, Ns,
                       fp.d_brec, fp.d_bhash);
    fp.nchunks_lm = 0;      // the tricubic kernel's window set is computed by its first launch (ensure_lm_windows)
    HIP_TRY(c, hipGetLastError());
    if (getenv("IONOTOMO_PLAN_STATS") && c->storage == IONO_F64) {      // columns per window, both chunk lengths (stderr; tuning aid)
        fp.o_key = o, fp.d_key = d, fp.R = R, fp.Ns = Ns, fp.tmax = tmax, fp.nb = nb, fp.nchunks = nchunks;
        rc = ensure_lm_windows(c);
        if (rc) return rc;
        const int nchunks_lm = fp.nchunks_lm;
        for (int which = 0; which < 2; ++which) {
            std::vector<uint4> hv((size_t)nb * (which ? nchunks_lm : nchunks));
            HIP_TRY(c, hipMemcpy(hv.data(), which ? fp.d_win_lm : fp.d_win, hv.size() * sizeof(uint4), hipMemcpyDeviceToHost));
            std::vector<int> cols, wys;
            size_t nfit = 0;
            for (const uint4 &w : hv) cols.push_back((int)((w.w & 255u) * ((w.w >> 8) & 255u))), wys.push_back((int)((w.w >> 8) & 255u)), nfit += (w.w >> 16) & 1u;
            std::sort(cols.begin(), cols.end());
            std::sort(wys.begin(), wys.end());
            auto q = [&](const std::vector<int> &v, double f) { return v.empty() ? 0 : v[std::min(v.size() - 1, (size_t)(f * v.size()))]; };
            if (!which) {      // wave-loads per window at rpl rows per load (the trilinear kernel's copy)
                std::vector<int> nl, wxs;
                for (const uint4 &w : hv) {
                    const int wx = (int)(w.w & 255u), rpl = std::max(1, (int)((w.w >> 20) & 15u));
                    nl.push_back((wx + rpl - 1) / rpl), wxs.push_back(wx);
                }
                std::sort(nl.begin(), nl.end());
                std::sort(wxs.begin(), wxs.end());
                double mean = 0;
                for (int v : nl) mean += v;
                fprintf(stderr, "[plan] wave-loads per window: mean %.2f p50 %d p75 %d p90 %d p95 %d p99 %d max %d; wx p50 %d p90 %d p99 %d max %d\n", mean / nl.size(),
                        q(nl, 0.5), q(nl, 0.75), q(nl, 0.9), q(nl, 0.95), q(nl, 0.99), nl.back(), q(wxs, 0.5), q(wxs, 0.9), q(wxs, 0.99), wxs.back());
            }
            fprintf(stderr, "[plan] %s windows: %zu, fit %.4f, columns p10 %d p50 %d p90 %d p99 %d max %d; wy p50 %d p99 %d max %d\n",
                    which ? "4-sample" : "8-sample", hv.size(), (double)nfit / hv.size(), q(cols, 0.1), q(cols, 0.5), q(cols, 0.9), q(cols, 0.99),
                    cols.back(), q(wys, 0.5), q(wys, 0.99), wys.back());
        }
    }
    unsigned long long *h_fits = (unsigned long long *)hp;        // (the summaries are no longer needed)
    HIP_TRY(c, hipMemcpyAsync(h_fits, d_fits, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));                  // (bstart is a host vector; the walk positions sit in pinned memory)
    fp.fit_fraction = nb > 0 ? (double)*h_fits / ((double)nb * nchunks) : 0.0;
    fp.o_key = o, fp.d_key = d, fp.R = R, fp.Ns = Ns, fp.tmax = tmax, fp.nb = nb, fp.nchunks = nchunks;
    fp.serial = ++c->fplan_counter, fp.lm_all_fit = false;
    return IONO_OK;
}

int iono_dispatch_name(const iono_dispatch_facts *facts, int op, char *out, int cap) {
    if (!facts || !out || cap < 1 || op < IONO_OP_FORWARD || op > IONO_OP_PHASE_ADJOINT) return fail(nullptr, IONO_ERR_ARG, "iono_dispatch_name: bad argument");
    const std::string t = dispatch_text(*facts, op);
    snprintf(out, (size_t)cap, "%s", t.c_str());
    return IONO_OK;
}
int iono_dispatch_describe(iono_ctx *c, int op, const double *o, const double *d, int64_t R, double tmax, int Ns, int kind, int kind_ne, int bend,
                           iono_dispatch_facts *facts_out, char *out, int cap) {
    { const int rc = need_grid(c); if (rc) return rc; }
    if (op < IONO_OP_FORWARD || op > IONO_OP_PHASE_ADJOINT || Ns < 2) return fail(c, IONO_ERR_ARG, "iono_dispatch_describe: bad argument");
    const iono_dispatch_facts f = facts_of(c, op, o, d, R, tmax, Ns, kind, kind_ne, bend);
    if (facts_out) *facts_out = f;
    return out ? iono_dispatch_name(&f, op, out, cap) : IONO_OK;
}

int iono_forward_plan_split(iono_ctx *c, int64_t *bundles_cut, int64_t *bundles_served, int64_t *rays_served, int64_t *rays_rest, int *min_rays,
                            int64_t *hist65, double *model_us3) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    const iono_ctx::FwdPlan &fp = c->fplan;
    const bool have = fp.R >= 0;
    if (bundles_cut) *bundles_cut = have ? fp.nb_all : 0;
    if (bundles_served) *bundles_served = have ? fp.nb : 0;
    if (rays_served) *rays_served = have ? fp.n_planned : 0;
    if (rays_rest) *rays_rest = have ? fp.n_rest : 0;
    if (min_rays) *min_rays = have ? fp.split_min : 0;
    if (hist65)
        for (int i = 0; i < 65; ++i) hist65[i] = have ? fp.hist[i] : 0;
    if (model_us3)
        for (int i = 0; i < 3; ++i) model_us3[i] = have ? fp.model_us[i] : 0.0;
    return IONO_OK;
}

int iono_forward_plan_info(iono_ctx *c, int64_t *n_bundles, int *n_chunks, double *fit_fraction) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    if (n_bundles) *n_bundles = c->fplan.R >= 0 ? c->fplan.nb : 0;
    if (n_chunks) *n_chunks = c->fplan.R >= 0 ? c->fplan.nchunks : 0;
    if (fit_fraction) *fit_fraction = c->fplan.R >= 0 ? c->fplan.fit_fraction : 0.0;
    return IONO_OK;
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
    rc = dispatch_storage(c, [&](auto *tag) -> int {
        using GT = std::remove_pointer_t<decltype(tag)>;
        // lanes = samples on ideal-uniform grids (k_forward_straight_u) for the rays ord[0 .. Rn) (ord null: all, in memory order)
        auto launch_u = [&](const int *ord, int64_t Rn, hipStream_t st) -> int {
            const size_t wl = sizeof(double) * Ns;
            const int nb = chunk_grid_blocks(resident_blocks(c, k_forward_straight_u<GT>, wl), Rn);
            iono_ctx::WalkPart &wp = c->walk[0];
            const int nw = nb * 4;                              // one chunk per wave
            const int wm = forward_walk_mode(c, (uint64_t)ncells(c) * sizeof(GT), ord);
            const bool use_part = wp.n == nw && wp.R == Rn && !(wm & 3) && ord == nullptr;
            const int rc2 = walk_cycles_reserve(c, wp, nw, nw);
            if (rc2) return rc2;
            hipLaunchKernelGGL((k_forward_straight_u<GT>), dim3(nb), block, wl, st, g, o, d, ord, Rn, tmax, Ns,
                               wm, c->d_unitw, tec, c->d_flags, use_part ? wp.d_starts : nullptr, wp.d_cyc);
            return IONO_OK;
        };
        // the tricubic counterpart on the node-major Lekien-Marsden records (k_forward_straight_lm; ensure_lm_fields first)
        auto launch_lm = [&](const int *ord, int64_t Rn, hipStream_t st) {
            const size_t wl = sizeof(double) * ((((size_t)Ns + 1) & ~(size_t)1) + 4 * U_MAXG * LM_RS);      // weights + ray state of 4 waves
            const int nb = chunk_grid_blocks(resident_blocks(c, k_forward_straight_lm, wl), Rn);
            // default walk for this kernel: all the waves of an XCD interleaved in that XCD's eighth of the walk (what is in
            // flight on an XCD is then one short stretch of neighbouring rays whose field records stay in its L2):
            // 1.49 ms against 1.60 with one contiguous chunk per wave
            const int wm = 2;
            hipLaunchKernelGGL(k_forward_straight_lm, dim3(nb), block, wl, st, g, c->d_F8, o, d, ord, Rn, tmax, Ns, wm,
                               c->d_unitw, tec, c->d_flags);
        };
        // float32 storage, unplanned: 2 x 2 corner blocks, two 16-B loads per sample (k_forward_straight_q4), for the rays ord[0 .. Rn)
        auto launch_q4 = [&](const int *ord, int64_t Rn) -> int {
            const int64_t n = ncells(c), padded = padded_count(c);
            if (!c->d_Q4) {
                HIP_TRY(c, hipMalloc((void **)&c->d_Q4, (size_t)padded * sizeof(float4)));
                HIP_TRY(c, hipMemsetAsync(c->d_Q4, 0, (size_t)padded * sizeof(float4), c->stream));
            }
            if (!c->Q4_valid) {
                launch_map<BlockPairs>(c, ew_blocks(c, n), n, (const float *)cur_values(c), c->d_Q4, c->nz);
                c->Q4_valid = true;
            }
            const size_t wl = sizeof(double) * Ns;
            const int nb = chunk_grid_blocks(resident_blocks(c, k_forward_straight_q4, wl), Rn);
            hipLaunchKernelGGL(k_forward_straight_q4, dim3(nb), block, wl, c->stream, g, c->d_Q4, o, d, ord, Rn, tmax, Ns,
                               forward_walk_mode(c, (uint64_t)padded * sizeof(float4), ord), c->d_unitw, tec, c->d_flags);
            return IONO_OK;
        };
        const iono_dispatch_facts facts = facts_of(c, IONO_OP_FORWARD, o, d, R, tmax, Ns, kind, kind, 0);
        const FwdKernel fk = pick_forward(facts);            // (the dispatch table: pick_forward)
        const bool q4_ok = facts.q4_ok != 0;
        if (fk == FK_BUNDLE_F32) {
            // float32 FAST MODE (iono_forward_f32_kernels.h): the served bundles with float32 window images and packed-float32
            // interpolation (TEC to ~1e-7), the rays outside them with the unplanned float32 kernel (float64 arithmetic)
            const iono_ctx::FwdPlan &fp = c->fplan;
            hipLaunchKernelGGL(k_forward_bundle_f32, dim3((unsigned)((fp.nb + 7) / 8 * 8)), block, B_SPLIT * F_WAVE_LDS + B_SPLIT * 64 * sizeof(double) + 16,
                               c->stream, g, o, d, fp.d_brec, fp.d_bhash, fp.d_win, fp.nb, fp.nchunks, tmax, Ns, c->d_unitw32, c->d_unitw, tec, c->d_flags);
            if (fp.n_rest > 0) {
                const int rc2 = q4_ok ? launch_q4(fp.d_order + fp.n_planned, fp.n_rest) : launch_u(fp.d_order + fp.n_planned, fp.n_rest, c->stream);
                if (rc2) return rc2;
            }
        } else if (fk == FK_Q4) {
            const int rc2 = launch_q4(order, R);
            if (rc2) return rc2;
        } else if (fk == FK_BUNDLE) {
            // (a workgroup per bundle: below two bundles per CU -- a single timestep is 214 -- the lanes = samples kernel, one wave
            //  per ray, fills the chip better: 7.5 against 11 us at config 2; likewise when the windows mostly do not fit the LDS
            //  image.  IONOTOMO_HYBRID_MIN=1 forces the bundle kernel for every bundle)
            // bundle-stationary: one workgroup per SERVED bundle, windows staged in LDS (iono_forward_plan_dev); the rays of the other
            // bundles, if the plan left any: lanes = samples, right behind it on the same stream (a second stream between fork / join
            // events measured SLOWER at every batch size: 0.106 against 0.103 ms at the bench shape, 0.040 against 0.032 at 41 664 rays)
            const iono_ctx::FwdPlan &fp = c->fplan;
            hipLaunchKernelGGL((k_forward_bundle<0>), dim3((unsigned)((fp.nb + 7) / 8 * 8)), block, B_SPLIT * B_WAVE_LDS + B_SPLIT * 64 * sizeof(double) + 16,
                               c->stream, g, o, d, fp.d_brec, fp.d_bhash, fp.d_win, fp.nb, fp.nchunks, tmax, Ns, c->d_unitw, tec,
                               c->d_flags, PhaseFreqs{}, 0);
            if (fp.n_rest > 0) {
                const int rc2 = launch_u(fp.d_order + fp.n_planned, fp.n_rest, c->stream);
                if (rc2) return rc2;
            }
        } else if (fk == FK_U) {
            const int rc2 = launch_u(order, R, c->stream);
            if (rc2) return rc2;
        } else if (fk == FK_BUNDLE_LM) {
            // bundles of neighbouring rays, one field pair per wave, windows staged in LDS (iono_cubic_kernels.h:k_forward_bundle_lm)
            int rc2 = ensure_lm_windows(c);          // (first: the windows say which node lines a rebuild for this plan must cover)
            if (rc2) return rc2;
            rc2 = ensure_lm_fields(c, true, true);
            if (rc2) return rc2;
            const int restricted = !c->FP_valid ? 1 : 0;      // the pair arrays hold this plan's lines only
            const iono_ctx::FwdPlan &fp = c->fplan;
            if (fp.n_rest > 0) {      // hybrid: the rays outside the served bundles on the node-major records, lanes = samples
                rc2 = ensure_lm_fields(c);
                if (rc2) return rc2;
            }
            static_assert(BL_LDS_BYTES <= 64 * 1024, "dynamic LDS beyond 64 KB would need hipFuncSetAttribute per device");
            hipLaunchKernelGGL(k_forward_bundle_lm, dim3((unsigned)((fp.nb + 7) / 8 * 8)), block, BL_LDS_BYTES, c->stream, g, c->d_FP,
                               padded_count(c), o, d, fp.d_order, fp.d_bstart, fp.d_win_lm, fp.d_rhash, fp.nb, fp.nchunks_lm, tmax, Ns, c->d_unitw, tec,
                               c->d_flags, restricted);
            if (fp.n_rest > 0) launch_lm(fp.d_order + fp.n_planned, fp.n_rest, c->stream);
        } else if (fk == FK_LM) {
            const int rc2 = ensure_lm_fields(c);
            if (rc2) return rc2;
            launch_lm(order, R, c->stream);
        } else if (fk == FK_FAST)      // (`order` is a speed hint: ignored here)
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
    { const int rc = need_ctx(c); if (rc) return rc; }
    if (i0 < 0 || i0 >= Na) return fail(c, IONO_ERR_ARG, "reference antenna index out of range");
    launch_map<SubtractReference>(c, ew_blocks(c, (int64_t)Na * NtNd), (int64_t)Na * NtNd, tec, NtNd, i0);
    HIP_TRY(c, hipMemsetAsync(tec + (int64_t)i0 * NtNd, 0, (size_t)NtNd * sizeof(double), c->stream));      // (row i0 - row i0, exactly)
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

int iono_vec_axpby_dev(iono_ctx *c, double *y, const double *x, int64_t n, const double *a_num, const double *a_den,
                       double a_sign, const double *b_num, const double *b_den) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    if (n < 0 || (n > 0 && (!y || !x))) return fail(c, IONO_ERR_ARG, "iono_vec_axpby_dev: null vector");
    if (n == 0) return IONO_OK;
    if ((((uintptr_t)y) | ((uintptr_t)x)) & 15) return fail(c, IONO_ERR_ARG, "iono_vec_axpby_dev: vectors must be 16-byte aligned");
    HIP_TRY(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_axpby, dim3(ew_blocks(c, (n + 1) / 2)), dim3(256), 0, c->stream, y, x, n, a_num, a_den, a_sign, b_num,
                       b_den);
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

// ---- caller-owned grid values + fused solver passes (iono_solver_kernels.h) ---------------------------------------
int iono_grid_padded_size(iono_ctx *c, int64_t *count) {
    int rc = need_grid(c);
    if (rc) return rc;
    if (!count) return fail(c, IONO_ERR_ARG, "null count");
    *count = padded_count(c);
    return IONO_OK;
}

int iono_grid_bind_values_dev(iono_ctx *c, double *values_dev) {
    int rc = need_grid(c);
    if (rc) return rc;
    if (values_dev && c->storage != IONO_F64) return fail(c, IONO_ERR_ARG, "caller-owned values need float64 storage");
    if (values_dev && (((uintptr_t)values_dev) & 15)) return fail(c, IONO_ERR_ARG, "values must be 16-byte aligned");
    c->d_M_ext = values_dev;
    c->nM_freq = c->nF8_freq = -1.0;
    c->F8_valid = c->FP_valid = false, c->FP_plan_serial = -1;
    return IONO_OK;
}

int iono_grid_values_changed(iono_ctx *c) {
    int rc = need_grid(c);
    if (rc) return rc;
    c->nM_freq = c->nF8_freq = -1.0;
    c->F8_valid = c->FP_valid = false, c->FP_plan_serial = -1;
    return IONO_OK;
}

int iono_rays_combine_dev(iono_ctx *c, const double *tec, const double *dobs, const double *s1, const double *s2, int Na,
                          int64_t NtNd, int i0, double a, double b, double *out, double *partial) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    if (Na < 1 || NtNd < 0 || i0 < 0 || i0 >= Na || !tec || !out) return fail(c, IONO_ERR_ARG, "iono_rays_combine_dev: bad argument");
    hipLaunchKernelGGL(k_rays_combine, dim3(IONO_NPART), dim3(256), 0, c->stream, tec, dobs, s1, s2, Na, NtNd, i0, a, b, out, partial);
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

int iono_vec_axpby_dot_dev(iono_ctx *c, double *y, const double *x, int64_t n, const double *an, int ann, const double *ad, int adn,
                           double a_sign, const double *bn, int bnn, const double *bd, int bdn, double *partial) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    if (n < 0 || (n > 0 && (!y || !x))) return fail(c, IONO_ERR_ARG, "iono_vec_axpby_dot_dev: null vector");
    if (ann > IONO_NPART || adn > IONO_NPART || bnn > IONO_NPART || bdn > IONO_NPART) return fail(c, IONO_ERR_ARG, "scalar count too large");
    hipLaunchKernelGGL(k_axpby_dot, dim3(IONO_NPART), dim3(256), 0, c->stream, y, x, n, an, ann, ad, adn, a_sign, bn, bnn, bd, bdn,
                       partial);
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

int iono_small_ray_pass_dev(iono_ctx *c, int mode, const double *tec, const double *dobs, const double *scale, const double *weight, double *r,
                            double *q, int Na, int64_t NtNd, int i0, const double *gamma, int gamma_count, double *dot1, double *dot2,
                            double *w) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    if (Na < 1 || NtNd < 1 || i0 < 0 || i0 >= Na || (int64_t)Na * NtNd > IONO_SMALL_RAYS) return fail(c, IONO_ERR_ARG, "iono_small_ray_pass_dev: need 1 <= Na * NtNd <= 32768");
    if (!tec || !scale || !r || !dot1 || !w || gamma_count > IONO_NPART) return fail(c, IONO_ERR_ARG, "iono_small_ray_pass_dev: bad argument");
    if (mode == 0) {
        if (!q || !gamma || gamma_count < 1 || !dot2) return fail(c, IONO_ERR_ARG, "iono_small_ray_pass_dev: CG needs q, gamma, dot2");
        hipLaunchKernelGGL((k_small_ray_pass<0>), dim3(1), dim3(1024), 0, c->stream, tec, dobs, scale, weight, r, q, Na, NtNd, i0, gamma,
                           gamma_count, dot1, dot2, w);
    } else if (mode == 1) {
        if (!dobs || !weight) return fail(c, IONO_ERR_ARG, "iono_small_ray_pass_dev: SIRT needs dobs, weight");
        hipLaunchKernelGGL((k_small_ray_pass<1>), dim3(1), dim3(1024), 0, c->stream, tec, dobs, scale, weight, r, q, Na, NtNd, i0, gamma,
                           gamma_count, dot1, dot2, w);
    } else {
        return fail(c, IONO_ERR_ARG, "iono_small_ray_pass_dev: mode 0 (CG) or 1 (SIRT)");
    }
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

static int compact_args_ok(iono_ctx *c, const int *idx, int64_t n) {
    if (n < 0 || (n > 0 && !idx)) return fail(c, IONO_ERR_ARG, "compact op: bad index");
    return IONO_OK;
}

int iono_compact_gather_dev(iono_ctx *c, double *full, const int *idx, int64_t n, double *out, int zero, double *partial) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    { const int rc = compact_args_ok(c, idx, n); if (rc) return rc; }
    hipLaunchKernelGGL(k_compact_gather, dim3(IONO_NPART), dim3(256), 0, c->stream, full, idx, n, out, zero, partial);
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

int iono_compact_scatter_dev(iono_ctx *c, double *full, const int *idx, int64_t n, const double *src) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    { const int rc = compact_args_ok(c, idx, n); if (rc) return rc; }
    launch_map<CompactScatter>(c, IONO_NPART, n, full, idx, src);
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

int iono_compact_cg_update_dev(iono_ctx *c, double *x, double *p, const double *s, const int *idx, int64_t n, double *full_p,
                               const double *an, int ann, const double *ad, int adn, const double *bn, int bnn, const double *bd,
                               int bdn) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    { const int rc = compact_args_ok(c, idx, n); if (rc) return rc; }
    hipLaunchKernelGGL(k_compact_cg_update, dim3(IONO_NPART), dim3(256), 0, c->stream, x, p, s, idx, n, full_p, an, ann, ad, adn, bn,
                       bnn, bd, bdn);
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

int iono_compact_sirt_update_dev(iono_ctx *c, double *x, const double *C, double *full_s, const int *idx, int64_t n, double *full_x,
                                 double relax, int nonneg, double *partial_max) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    { const int rc = compact_args_ok(c, idx, n); if (rc) return rc; }
    hipLaunchKernelGGL(k_compact_sirt_update, dim3(IONO_NPART), dim3(256), 0, c->stream, x, C, full_s, idx, n, full_x, relax, nonneg,
                       partial_max);
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

// ---- measured load balance of the chunked kernels -----------------------------------------------
int iono_walk_cycles(iono_ctx *c, int which, uint64_t *out, int cap, int *n_chunks, int *n_units) {
    if (!c || !n_chunks || which < 0 || which > 1) return fail(c, IONO_ERR_ARG, "iono_walk_cycles: bad argument");
    const iono_ctx::WalkPart &wp = c->walk[which];
    *n_chunks = wp.last_n;
    if (n_units) *n_units = wp.last_units;
    if (wp.last_n == 0 || !out || cap <= 0) return IONO_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, wp.d_cyc, (size_t)std::min(cap, wp.last_n) * 8, hipMemcpyDeviceToHost));
    return IONO_OK;
}

int iono_walk_partition_set(iono_ctx *c, int which, const int64_t *starts, int n_chunks, int64_t R) {
    if (!c || which < 0 || which > 1) return fail(c, IONO_ERR_ARG, "iono_walk_partition_set: bad argument");
    iono_ctx::WalkPart &wp = c->walk[which];
    if (!starts || n_chunks <= 0) {          // clear: back to equal ray counts
        wp.n = 0, wp.R = -1;
        return IONO_OK;
    }
    if (starts[0] != 0 || starts[n_chunks] != R) return fail(c, IONO_ERR_ARG, "partition must start at 0 and end at R");
    for (int b = 0; b < n_chunks; ++b)
        if (starts[b + 1] < starts[b]) return fail(c, IONO_ERR_ARG, "partition boundaries must not decrease");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));           // a launch still reading the old boundaries
    if (wp.d_starts) (void)hipFree(wp.d_starts);
    wp.d_starts = nullptr, wp.n = 0, wp.R = -1;
    HIP_TRY(c, hipMalloc((void **)&wp.d_starts, (size_t)(n_chunks + 1) * 8));
    HIP_TRY(c, hipMemcpy(wp.d_starts, starts, (size_t)(n_chunks + 1) * 8, hipMemcpyHostToDevice));
    wp.n = n_chunks, wp.R = R;
    return IONO_OK;
}

// ---- adjoint (device pointers) ----------------------------------------------------------------
// ---- node-stationary back-projection plan (iono_binned_kernels.h) ---------------------------------------------------------
int iono_adjoint_plan_clear(iono_ctx *c) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    plan_free(c);
    return IONO_OK;
}

// Tricubic plans: the tiles the fold passes have to visit (iono_cubic_kernels.h: k_lm_fold_*_tiles).  T = tiles the samples reach
// (k_plan_touch); A1 = T dilated along z, A2 = A1 along y, A3 = A2 along x; per pass the list of output tiles with the flags of its
// input set along the fold axis.
static int plan_fold_tiles(iono_ctx *c, int64_t R, int Ns) {
    iono_ctx::AdjPlan &pl = c->plan;
    pl.tile_n[0] = pl.tile_n[1] = pl.tile_n[2] = 0;
    const int ntx = (c->nx + LMT_X - 1) / LMT_X, nty = (c->ny + LMT_Y - 1) / LMT_Y, ntz = (c->nz + LMT_Z - 1) / LMT_Z;
    const int64_t nt = (int64_t)ntx * nty * ntz;
    if (nt > ((int64_t)1 << 28)) return IONO_OK;
    DevBuf tb(c);
    HIP_TRY(c, tb.alloc((size_t)nt));
    unsigned char *d_touch = tb.as<unsigned char>();
    HIP_TRY(c, hipMemsetAsync(d_touch, 0, (size_t)nt, c->stream));
    hipLaunchKernelGGL(k_plan_touch, dim3(ew_blocks(c, R)), dim3(256), 0, c->stream, pl.d_uray, R, Ns, c->nx, c->ny, c->nz, nty, ntz, d_touch);
    HIP_TRY(c, hipGetLastError());
    std::vector<unsigned char> in((size_t)nt), out((size_t)nt);
    HIP_TRY(c, hipMemcpyAsync(in.data(), d_touch, (size_t)nt, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const long long reached = (long long)std::count(in.begin(), in.end(), (unsigned char)1);
    const int64_t stride[3] = {1, ntz, (int64_t)ntz * nty};      // fold order: z, y, x
    const int extent[3] = {ntz, nty, ntx};
    std::vector<LmTile> lists[3];
    auto coord = [&](int64_t t, int axis) { return axis == 0 ? (int)(t % ntz) : axis == 1 ? (int)((t / ntz) % nty) : (int)(t / ((int64_t)ntz * nty)); };
    for (int axis = 0; axis < 3; ++axis) {
        const int64_t st = stride[axis];
        for (int64_t t = 0; t < nt; ++t) {
            const int l = coord(t, axis);
            const bool lo = l > 0 && in[(size_t)(t - st)], hi = l + 1 < extent[axis] && in[(size_t)(t + st)];
            out[(size_t)t] = in[(size_t)t] || lo || hi;
            if (out[(size_t)t]) lists[axis].push_back(LmTile{(int)t, (in[(size_t)t] ? 1 : 0) | (lo ? 2 : 0) | (hi ? 4 : 0)});
        }
        if (axis == 0)       // G8 is zeroed on A1 and the z pass reads it there: its input set is A1 itself
            for (LmTile &e : lists[0]) {
                const int l = coord(e.id, 0);
                e.flags = 1 | (l > 0 && out[(size_t)(e.id - 1)] ? 2 : 0) | (l + 1 < ntz && out[(size_t)(e.id + 1)] ? 4 : 0);
            }
        in.swap(out);
    }
    const size_t total = lists[0].size() + lists[1].size() + lists[2].size();
    if (total == 0) return IONO_OK;
    HIP_TRY(c, plan_reserve(pl.d_tiles, pl.cap_tiles, total * sizeof(LmTile)));
    size_t off = 0;
    for (int axis = 0; axis < 3; ++axis) {
        HIP_TRY(c, hipMemcpy(pl.d_tiles + off, lists[axis].data(), lists[axis].size() * sizeof(LmTile), hipMemcpyHostToDevice));
        pl.tile_off[axis] = (int)off, pl.tile_n[axis] = (int)lists[axis].size();
        off += lists[axis].size();
    }
    if (getenv("IONOTOMO_PLAN_STATS"))
        fprintf(stderr, "[ionotomo] fold tiles: %lld of %lld reached; passes over %d / %d / %d\n",
                reached, (long long)nt, pl.tile_n[0], pl.tile_n[1], pl.tile_n[2]);
    return IONO_OK;
}

int iono_adjoint_plan_dev(iono_ctx *c, const double *o, const double *d, int64_t R, double tmax, int Ns, int kind) {
    int rc = check_common(c, R, Ns, kind, 0);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    plan_reset(c);
    const bool cubic = kind == IONO_INTERP_TRICUBIC;
    if (R == 0 || R > (int64_t)UINT32_MAX || Ns > 65535 || !(cubic ? cubic_fast_ok(c, Ns) : ideal_path_ok(c, Ns)) ||
        (int64_t)c->nx * c->ny > ((int64_t)1 << 24) || c->nz >= (1 << 24))      // (k_adjoint_binned forms node indices with 24-bit multiply-adds)
        return IONO_OK;                                  // no plan: the ray-stationary kernels serve this case
    iono_ctx::AdjPlan &pl = c->plan;
    HIP_TRY(c, plan_reserve(pl.d_uray, pl.cap_uray, (size_t)R * 8 * sizeof(double)));
    HIP_TRY(c, plan_reserve(pl.d_hash, pl.cap_hash, (size_t)R * sizeof(uint2)));
    const GridView g = view(c);
    if (cubic)
        launch_map<PlanUrays<true>>(c, ew_blocks(c, R), R, g, o, d, tmax, Ns, pl.d_uray, pl.d_hash);
    else
        launch_map<PlanUrays<false>>(c, ew_blocks(c, R), R, g, o, d, tmax, Ns, pl.d_uray, pl.d_hash);
    HIP_TRY(c, hipGetLastError());
    const int nbx = (c->nx - 1 + BIN_SX - 1) / BIN_SX, nby = (c->ny - 1 + BIN_SY - 1) / BIN_SY, nbz = (c->nz - 1 + BIN_SZ - 1) / BIN_SZ;
    const int64_t nbox = (int64_t)nbx * nby * nbz;
    if (nbox > (int64_t)INT32_MAX / 4) {
        plan_reset(c);
        return IONO_OK;
    }
    // pass 1 on the device: segments per ray and per box (iono_binned_kernels.h:k_plan_segments), walking the rays in the forward
    // plan's order when one exists for these very arrays (neighbours in it share boxes)
    const int *walk = c->fplan.R == R && c->fplan.o_key == (const void *)o && c->fplan.d_key == (const void *)d ? c->fplan.d_order : nullptr;
    DevBuf scratch(c);
    const size_t off_cnt = ((size_t)R * sizeof(int) + 15) & ~(size_t)15, off_start = off_cnt + (size_t)nbox * sizeof(int),
                 off_fill = off_start + (size_t)nbox * sizeof(int), off_out = (off_fill + (size_t)nbox * sizeof(int) + 7) & ~(size_t)7;
    HIP_TRY(c, scratch.alloc(off_out + 16));
    char *sb = scratch.as<char>();
    int *d_nseg32 = (int *)sb, *d_cnt = (int *)(sb + off_cnt), *d_start = (int *)(sb + off_start), *d_fill = (int *)(sb + off_fill);
    unsigned long long *d_out = (unsigned long long *)(sb + off_out);
    std::vector<int> h_nseg((size_t)R), h_cnt((size_t)nbox);
    unsigned long long outside = 0;
    // The lanes a segment occupies (segl): the widest first; while the segments come out less than half full, count again with
    // half the width; keep the width with the fewest lanes in total (an entry costs about two lanes' worth of gathers).
    auto count_segments = [&](int width, int64_t *n_seg) -> int {
        HIP_TRY(c, hipMemsetAsync(sb + off_cnt, 0, off_out + 16 - off_cnt, c->stream));
        hipLaunchKernelGGL((k_plan_segments<false>), dim3(ew_blocks(c, R)), dim3(256), 0, c->stream, pl.d_uray, R, Ns, c->nx, c->ny, c->nz,
                           nbx, nby, nbz, width, d_nseg32, d_cnt, (const int *)nullptr, (int *)nullptr, (uint2 *)nullptr, d_out, walk);
        HIP_TRY(c, hipGetLastError());
        if (n_seg) {
            HIP_TRY(c, hipMemcpyAsync(h_cnt.data(), d_cnt, (size_t)nbox * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            *n_seg = 0;
            for (int64_t q = 0; q < nbox; ++q) *n_seg += h_cnt[(size_t)q];
        }
        return IONO_OK;
    };
    int segl = BIN_SEG;
    if (c->seg_lanes == 4 || c->seg_lanes == 8 || c->seg_lanes == 16) {      // forced width (A/B, tests)
        segl = c->seg_lanes;
        rc = count_segments(segl, nullptr);
        if (rc) return rc;
    } else {
        double best = -1;
        int counted = 0;
        for (int width = BIN_SEG; width >= 4; width /= 2) {
            int64_t n_seg = 0;
            rc = count_segments(width, &n_seg);
            if (rc) return rc;
            counted = width;
            const double cost = (double)n_seg * (width + 2);
            if (best < 0 || cost < best) best = cost, segl = width;
            if (n_seg == 0 || (double)R * Ns > 0.5 * (double)n_seg * width) break;      // at least half full: narrower cannot pay
        }
        if (counted != segl) {
            rc = count_segments(segl, nullptr);
            if (rc) return rc;
        }
    }
    HIP_TRY(c, hipMemcpyAsync(h_nseg.data(), d_nseg32, (size_t)R * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(h_cnt.data(), d_cnt, (size_t)nbox * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&outside, d_out, sizeof(outside), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // Work-unit size: BIN_UNIT 16-lane segments' worth of lane passes whatever the width -- but a problem too small to give every
    // workgroup slot (5 per CU) a unit of that size is cut finer, down to two passes per unit: the launch then lasts as long as its
    // LONGEST unit (2 604 rays through 128^3: 26 040 segments, the hot boxes under the station core held 512 of them each).
    int unit_segs = BIN_UNIT * (BIN_SEG / segl);
    {
        int64_t total = 0;
        for (int64_t q = 0; q < nbox; ++q) total += h_cnt[(size_t)q];
        const int64_t slots = (int64_t)c->num_cus * 5, pass = 256 / segl;
        const int64_t even = (total + slots - 1) / slots;
        if (even < unit_segs) {
            const int full = unit_segs;
            unit_segs = (int)std::max<int64_t>(2 * pass, (even + pass - 1) / pass * pass);
            // ... but no finer than one round of workgroups: boxes do not fill their last unit, so the count is taken, not estimated
            auto n_units = [&](int64_t size) {
                int64_t u = 0;
                for (int64_t q = 0; q < nbox; ++q) u += (h_cnt[(size_t)q] + size - 1) / size;
                return u;
            };
            while (unit_segs < full && n_units(unit_segs) > slots) unit_segs += (int)pass;
        }
    }
    for (int64_t r = 0; r < R; ++r) pl.n_invalid += h_nseg[(size_t)r] < 0;
    // exclusive scan of the box counts; work units of <= BIN_UNIT segments, largest first
    std::vector<int> start((size_t)nbox + 1, 0);
    int64_t ne = 0;
    for (int64_t b = 0; b < nbox; ++b) {
        start[(size_t)b] = (int)ne;
        ne += h_cnt[(size_t)b];
        if (ne > (int64_t)INT32_MAX) break;
    }
    if (ne == 0 || ne > (int64_t)INT32_MAX) {
        plan_reset(c);
        return IONO_OK;
    }
    start[(size_t)nbox] = (int)ne;
    const int64_t n_invalid = pl.n_invalid;
    std::vector<BinUnit> units;
    for (int64_t b = 0; b < nbox; ++b) {
        const int64_t lo = start[(size_t)b], hi = start[(size_t)b + 1];
        if (lo == hi) continue;
        const int zb = (int)(b % nbz), bj = (int)((b / nbz) % nby), bi = (int)(b / ((int64_t)nbz * nby));
        for (int64_t q = lo; q < hi; q += unit_segs)
            units.push_back(BinUnit{bi * BIN_SX - BIN_H, bj * BIN_SY - BIN_H, zb * BIN_SZ, (int)q, (int)std::min(hi, q + unit_segs)});
    }
    // z-slabs (iono_adjoint_plan_slabs; 1 by default): slab s = box layers [nbz s / nslab, nbz (s + 1) / nslab).  A unit writes the 16 node
    // levels of its layer only, so once the units of slab s have run, the node levels below the first level of slab s + 1 are final:
    // a multi-GPU solver exchanges them while the next slab is back-projected (ionotomo_amd/parallel.py).  Units: by slab, then largest first.
    const int nslab = std::max(1, std::min(std::min(c->plan_slabs, 8), nbz));
    auto slab_of = [&](const BinUnit &u) { return std::min(nslab - 1, (u.z0 / BIN_SZ) * nslab / nbz); };
    std::stable_sort(units.begin(), units.end(), [&](const BinUnit &a, const BinUnit &b) {
        const int sa = slab_of(a), sb = slab_of(b);
        return sa != sb ? sa < sb : a.e_hi - a.e_lo > b.e_hi - b.e_lo;
    });
    pl.nslab = nslab;
    for (int sidx = 0; sidx <= nslab; ++sidx) {
        // first box layer of slab sidx: the smallest layer L with L * nslab / nbz >= sidx
        int L = 0;
        while (L < nbz && std::min(nslab - 1, L * nslab / nbz) < sidx) ++L;
        pl.slab_z[sidx] = sidx == 0 ? 0 : (sidx == nslab ? c->nz : L * BIN_SZ);
        pl.slab_unit[sidx] = (int)(std::lower_bound(units.begin(), units.end(), sidx, [&](const BinUnit &u, int v) { return slab_of(u) < v; }) - units.begin());
    }
    // pass 2 on the device: the segments into their boxes (+ BIN_ENTRY_PAD zero entries: the kernel prefetches two passes ahead)
    HIP_TRY(c, plan_reserve(pl.d_entries, pl.cap_entries, ((size_t)ne + BIN_ENTRY_PAD) * sizeof(uint2)));
    HIP_TRY(c, hipMemsetAsync(pl.d_entries + ne, 0, BIN_ENTRY_PAD * sizeof(uint2), c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_start, start.data(), (size_t)nbox * sizeof(int), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL((k_plan_segments<true>), dim3(ew_blocks(c, R)), dim3(256), 0, c->stream, pl.d_uray, R, Ns, c->nx, c->ny, c->nz, nbx,
                       nby, nbz, segl, (int *)nullptr, d_cnt, (const int *)d_start, d_fill, pl.d_entries, (unsigned long long *)nullptr,
                       walk);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, plan_reserve(pl.d_units, pl.cap_units, units.size() * sizeof(BinUnit)));
    HIP_TRY(c, hipMemcpyAsync(pl.d_units, units.data(), units.size() * sizeof(BinUnit), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));          // (start / units are host vectors; scratch goes back to the pool)
    if (cubic) {
        rc = plan_fold_tiles(c, R, Ns);
        if (rc) return rc;
    }
    pl.o_key = o, pl.d_key = d, pl.R = R, pl.Ns = Ns, pl.tmax = tmax, pl.kind = kind;
    pl.n_entries = ne, pl.n_units = (int)units.size(), pl.n_invalid = n_invalid, pl.segl = segl;
    pl.outside_fraction = (double)outside / (double)ne;
    {   // a node lies in the images of at most 8 boxes; a segment holds at most segl samples
        int64_t maxcnt = 0;
        for (int64_t b = 0; b < nbox; ++b) maxcnt = std::max<int64_t>(maxcnt, h_cnt[(size_t)b]);
        const double bound = std::min((double)R * Ns, 8.0 * (double)maxcnt * segl);
        pl.fix_bits = std::max(12, (int)std::ceil(std::log2(std::max(bound, 2.0))) + 1);
    }
    return IONO_OK;
}

int iono_adjoint_plan_info(iono_ctx *c, int64_t *n_entries, int *n_units, double *outside_fraction) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    if (n_entries) *n_entries = c->plan.R >= 0 ? c->plan.n_entries : 0;
    if (n_units) *n_units = c->plan.R >= 0 ? c->plan.n_units : 0;
    if (outside_fraction) *outside_fraction = c->plan.outside_fraction;
    return IONO_OK;
}

int iono_adjoint_plan_segment_lanes(iono_ctx *c, int *lanes) {
    if (!c || !lanes) return fail(c, IONO_ERR_ARG, "null argument");
    *lanes = c->plan.R >= 0 ? c->plan.segl : 0;
    return IONO_OK;
}

}  // extern "C"  (templates below need C++ linkage)

// the node-stationary kernel instantiated for the plan's segment width: BY_SEGL(pl.segl, launch using SL)
#define BY_SEGL(segl, ...)                                                                                                          \
    do {                                                                                                                           \
        if ((segl) == 4) {                                                                                                         \
            constexpr int SL = 4;                                                                                                  \
            __VA_ARGS__;                                                                                                           \
        } else if ((segl) == 8) {                                                                                                  \
            constexpr int SL = 8;                                                                                                  \
            __VA_ARGS__;                                                                                                           \
        } else {                                                                                                                   \
            constexpr int SL = 16;                                                                                                 \
            __VA_ARGS__;                                                                                                           \
        }                                                                                                                          \
    } while (0)

// One launch of the LDS-tiled back-projection (ideal-uniform grids).  CUBIC: channel `field` of the tricubic transpose.
template <typename AT, int MODE, bool CUBIC, bool PHASE = false, typename GT = double>
static int launch_adjoint_tile(iono_ctx *c, const GridView &g, const double *o, const double *d, const int *order, const double *w,
                               const double *tec, const double *dobs, const double *cdct, int Na, int64_t NtNd, int i0, int64_t R,
                               double tmax, int Ns, AT *grad, int field, PhaseFreqs pf = PhaseFreqs{}, int ldw = 0) {
    constexpr int NW = 4;    // waves per workgroup (8 waves sharing one tile, bundles of 128: measured 8 % slower)
    const size_t tl = sizeof(double) * (((size_t)Ns + 1) & ~(size_t)1) + (ADJ_REF * NW + ADJ_SUB) * sizeof(double) +
                      sizeof(double) * T_WIN * T_WIN * T_TKP + 2 * T_TK * sizeof(int) + 16;
    const int per_cu = blocks_per_cu(k_adjoint_straight_tile<AT, MODE, NW, CUBIC, PHASE, GT>, 64 * NW, tl, 1);
    int nb = per_cu * c->num_cus;
    const int64_t nbund = (R + 16 * NW - 1) / (16 * NW);   // at least ~64 rays per workgroup
    if (nb > nbund) nb = (int)nbund;
    if (nb >= 8) nb = nb / 8 * 8;
    iono_ctx::WalkPart &wp = c->walk[1];
    const bool use_part = wp.n >= nb && wp.R == R;
    const int nchunks = use_part ? wp.n : nb;
    const int rc = walk_cycles_reserve(c, wp, nchunks, nb);
    if (rc) return rc;
    if (!c->d_chunk_counter) HIP_TRY(c, hipMalloc((void **)&c->d_chunk_counter, 4));
    if (use_part) HIP_TRY(c, hipMemsetAsync(c->d_chunk_counter, 0, 4, c->stream));
    hipLaunchKernelGGL((k_adjoint_straight_tile<AT, MODE, NW, CUBIC, PHASE, GT>), dim3(nb), dim3(64 * NW), tl, c->stream, g, o, d,
                       order, w, tec, dobs, cdct, Na, NtNd, i0, R, tmax, Ns, c->adj_mode, c->d_unitw, grad, c->d_flags,
                       use_part ? wp.d_starts : nullptr, nchunks, c->d_chunk_counter, wp.d_cyc, field, pf, ldw);
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

// Deterministic mode, before a fixed-point launch: the integer grid (+ one word behind it for the launch's largest |w h|), the plan's
// bound on the terms of a node's sum (counted once per plan by the fixed-point kernel itself), and that largest |w h| itself.
static int fix_prepare(iono_ctx *c, const double *wr, int64_t R, int Ns) {
    iono_ctx::AdjPlan &pl = c->plan;
    const int64_t n = ncells(c);
    if (!c->d_fixgrid) {
        // [n] integer grid | the launch's largest |w h| | the two unit counters of a tricubic transpose (one memset clears both words)
        HIP_TRY(c, hipMalloc((void **)&c->d_fixgrid, ((size_t)n + 2) * sizeof(unsigned long long)));
        HIP_TRY(c, hipMemsetAsync(c->d_fixgrid, 0, ((size_t)n + 2) * sizeof(unsigned long long), c->stream));
    }
    unsigned long long *fixmax = c->d_fixgrid + n;
    if (!pl.fix_counted) {
        // the terms of the fullest node's sum: the fixed-point kernel in counting mode over the whole plan, then the maximum
        HIP_TRY(c, hipMemsetAsync(fixmax, 0, sizeof(unsigned long long), c->stream));
        const GridView g = view(c);
        const size_t bin_lds = sizeof(double) * ((((size_t)Ns + 1) & ~(size_t)1) + BIN_TILE);
        if (pl.n_units > 0)
            BY_SEGL(pl.segl, hipLaunchKernelGGL((k_adjoint_binned<double, 0, double, SL, true>), dim3(pl.n_units), dim3(BIN_THREADS), bin_lds,
                                                c->stream, g, pl.d_uray, pl.d_entries, pl.d_units, wr, Ns, c->d_unitw, (double *)c->d_fixgrid,
                                                PhaseFreqs{}, 0, fixmax, -1));
        hipLaunchKernelGGL(k_fix_nodemax, dim3((unsigned)std::min<int64_t>(1024, (n + 255) / 256)), dim3(256), 0, c->stream, c->d_fixgrid, n, fixmax);
        unsigned long long nodemax = 0;
        HIP_TRY(c, hipMemcpyAsync(&nodemax, fixmax, sizeof(nodemax), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        const int bits = std::max(12, (int)std::ceil(std::log2(std::max((double)nodemax, 2.0))) + 1);
        // (the count is authoritative: it ran this very kernel, overhang samples that go by global atomics included, which the a-priori
        //  bound of the plan -- 8 boxes x the fullest box x the segment length -- does not cover: ADVICE r4)
        pl.fix_bits = bits;
        pl.fix_counted = true;
        // a trilinear plan: list the tiles its rays reach, so that the integer grid is converted (and re-zeroed) only there
        if (pl.kind == IONO_INTERP_TRILINEAR && pl.tile_n[0] == 0) {
            const int rct = plan_fold_tiles(c, R, Ns);
            if (rct) return rct;
        }
    }
    HIP_TRY(c, hipMemsetAsync(fixmax, 0, 2 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_fix_absmax, dim3((unsigned)std::min<int64_t>(256, (R + 255) / 256)), dim3(256), 0, c->stream, wr, pl.d_uray, R, fixmax);
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

// mode 0: weights w[R];  1: fused residual (tec, dobs, cdct);  2: differential weights of v[R] * scale[R] (tec = v,
// cdct = scale or null).  kind: trilinear or tricubic transpose.
template <typename AT, int MODE>
static int adjoint_straight_typed(iono_ctx *c, const GridView &g, const double *o, const double *d, const int *order,
                                  const double *w, const double *tec, const double *dobs, const double *cdct, int Na,
                                  int64_t NtNd, int i0, int64_t R, double tmax, int Ns, int kind, AT *grad) {
    const iono_ctx::AdjPlan &pl = c->plan;
    const iono_dispatch_facts facts = facts_of(c, IONO_OP_ADJOINT, o, d, R, tmax, Ns, kind, kind, 0);
    const AdjKernel ak = pick_adjoint(facts);                // (the dispatch table: pick_adjoint)
    const bool planned = facts.adj_planned && c->variant != 2;
    const double *wr = w;
    if (planned && pl.n_invalid > 0) HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)c->d_flags, 1, 1, c->stream));   // out-of-grid rays
    if (planned && MODE != 0) {          // the reference-antenna sums of the fused modes, once per ray
        if (c->rayw_cap < R) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (c->d_rayw) (void)hipFree(c->d_rayw);
            c->d_rayw = nullptr, c->rayw_cap = 0;
            HIP_TRY(c, hipMalloc((void **)&c->d_rayw, (size_t)R * sizeof(double)));
            c->rayw_cap = R;
        }
        // (+ the check that the rays are still the planned ones, in the same pass: plan_verify_ray)
        hipLaunchKernelGGL((k_ray_weights<MODE == 0 ? 1 : MODE>), dim3((unsigned)std::min<int64_t>(IONO_NPART, (NtNd + RSTEP_PT - 1) / RSTEP_PT)),
                           dim3(64 * RSTEP_WAVES), 0, c->stream, tec, dobs,
                           cdct, Na, NtNd, i0, c->d_rayw, o, d, (const uint2 *)pl.d_hash, pl.d_uray, c->d_flags);
        wr = c->d_rayw;
    } else if (planned && !c->plan_verified) {
        hipLaunchKernelGGL(k_plan_verify, dim3(ew_blocks(c, R)), dim3(256), 0, c->stream, o, d, R, (const uint2 *)pl.d_hash, pl.d_uray, c->d_flags);
    }
    c->plan_verified = false;
    const size_t bin_lds = sizeof(double) * ((((size_t)Ns + 1) & ~(size_t)1) + BIN_TILE);      // float64 box image for either AT
    if (ak == AK_BINNED || ak == AK_BINNED_FIX) {
        int u_lo = 0, u_hi = pl.n_units;
        if (c->unit_lo >= 0) u_lo = std::min(c->unit_lo, pl.n_units), u_hi = std::max(u_lo, std::min(c->unit_hi, pl.n_units));
        c->unit_lo = c->unit_hi = -1;
        if (ak == AK_BINNED_FIX) {
            // fixed-point accumulation: the largest |w h| of this launch -> scale; integers in the box images and in d_fixgrid; converted
            // into `grad` (and re-zeroed) by FixConvert (iono_binned_kernels.h)
            const int64_t n = ncells(c);
            { const int rcf = fix_prepare(c, wr, R, Ns); if (rcf) return rcf; }
            unsigned long long *fixmax = c->d_fixgrid + n;      // (allocated by fix_prepare on first use)
            if (u_hi > u_lo)
                BY_SEGL(pl.segl, hipLaunchKernelGGL((k_adjoint_binned<double, 0, double, SL, true>), dim3(u_hi - u_lo), dim3(BIN_THREADS), bin_lds,
                                                    c->stream, g, pl.d_uray, pl.d_entries, pl.d_units + u_lo, wr, Ns, c->d_unitw,
                                                    (double *)c->d_fixgrid, PhaseFreqs{}, 0, fixmax, pl.fix_bits));
            if (pl.tile_n[0] > 0) {          // (the tiles the planned rays reach, listed when the mode was first used: fix_prepare)
                const LmTileGeom tg{c->nx, c->ny, c->nz, (c->ny + LMT_Y - 1) / LMT_Y, (c->nz + LMT_Z - 1) / LMT_Z};
                const int64_t nn = (int64_t)pl.tile_n[0] * LMT_NODES;
                launch_map<FixConvertTiles<AT>>(c, ew_blocks(c, nn), nn, c->d_fixgrid, grad, fixmax, pl.fix_bits,
                                                (const LmTile *)(pl.d_tiles + pl.tile_off[0]), tg, 0.0);
            } else {
                launch_map<FixConvert<AT>>(c, ew_blocks(c, n), n, c->d_fixgrid, grad, fixmax, pl.fix_bits, 0.0);
            }
            HIP_TRY(c, hipGetLastError());
            return IONO_OK;
        }
        if (u_hi > u_lo)
            BY_SEGL(pl.segl, hipLaunchKernelGGL((k_adjoint_binned<AT, 0, double, SL>), dim3(u_hi - u_lo), dim3(BIN_THREADS), bin_lds, c->stream, g,
                                                pl.d_uray, pl.d_entries, pl.d_units + u_lo, wr, Ns, c->d_unitw, grad, PhaseFreqs{}, 0));
        HIP_TRY(c, hipGetLastError());
        return IONO_OK;
    }
    if (c->unit_lo >= 0) {
        // a unit range is honoured by the planned trilinear back-projection only: any other launch would add the WHOLE back-projection
        // once per slab: refuse instead of discarding the range
        c->unit_lo = c->unit_hi = -1;
        return fail(c, IONO_ERR_ARG, "a work-unit range (iono_adjoint_unit_range) is pending, but this launch is not the planned trilinear "
                                     "back-projection of these rays (no plan, a replaced plan, tricubic, or IONOTOMO_VARIANT=2): nothing launched");
    }
    // The planned tricubic transpose accumulates 64-bit fixed point BY DEFAULT (round 5): the integer LDS atomic is the cheaper
    // instruction (6.4 against 8.2 LDS cycles) and the z fold reads the integers directly -- 2.15 against 2.55 ms at the bench shape,
    // run-to-run identical bits, 1.3e-12 of the largest value away from the float sum (profiles/r05_ab_binned.json).
    // (The float-atomic form is no longer built: round 6.)  iono_set_deterministic(1) additionally switches the TRILINEAR back-projection.
    const bool fix_cubic = ak == AK_BINNED_LM4;
    if (ak == AK_REFUSED)
        return fail(c, IONO_ERR_ARG, "deterministic mode serves the planned trilinear and tricubic back-projections only (iono_adjoint_plan_dev for these rays first)");
    if (ak == AK_TILE)
        return launch_adjoint_tile<AT, MODE, false>(c, g, o, d, order, w, tec, dobs, cdct, Na, NtNd, i0, R, tmax, Ns, grad, -1);
    if (ak == AK_BINNED_LM4 || ak == AK_TILE_CUBIC) {
        // 8 channel scatters (cubic Hermite value / slope weights per axis) into G8[8][nodes], then the transposed
        // difference stencils fold them into the node gradient (iono_cubic_kernels.h)
        const int64_t n = ncells(c);
        if (!c->d_G8) HIP_TRY(c, hipMalloc((void **)&c->d_G8, (size_t)n * LM_NF * sizeof(double)));
        // planned: only the tiles the plan's rays reach are zeroed, scattered into and folded
        const bool tiled = fix_cubic;
        const LmTileGeom tg{c->nx, c->ny, c->nz, (c->ny + LMT_Y - 1) / LMT_Y, (c->nz + LMT_Z - 1) / LMT_Z};
        if (tiled)
            hipLaunchKernelGGL(k_lm_zero_tiles, dim3(pl.tile_n[0]), dim3(LMT_NODES / 2), 0, c->stream, c->d_G8, pl.d_tiles + pl.tile_off[0], tg);
        else
            HIP_TRY(c, hipMemsetAsync(c->d_G8, 0, (size_t)n * LM_NF * sizeof(double), c->stream));
        const unsigned long long *fixmax = nullptr;
        if (fix_cubic) {
            const int rcf = fix_prepare(c, wr, R, Ns);
            if (rcf) return rcf;
            fixmax = c->d_fixgrid + n;
        }
        if (fix_cubic) {
            // four channels (one z kind) per traversal: two launches instead of eight (k_adjoint_binned_lm4), fixed-point LDS atomics
            const size_t l4 = LM4_LDS_BYTES(Ns);
            BY_SEGL(pl.segl, {
                if (!c->lm4_attr[SL == 4 ? 0 : SL == 8 ? 1 : 2]) {
                    HIP_TRY(c, hipFuncSetAttribute((const void *)k_adjoint_binned_lm4<SL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
                    c->lm4_attr[SL == 4 ? 0 : SL == 8 ? 1 : 2] = true;
                }
                // (persistent workgroups, one per CU -- the four images are all the LDS a CU has -- pulling units off a counter that
                //  fix_prepare has just cleared behind its scale word)
                const dim3 pgrid((unsigned)std::min(pl.n_units, c->lm4_groups > 0 ? c->lm4_groups : c->num_cus));
                for (int rb = 0; rb < (LM_NCH == 8 ? 1 : 2); ++rb)      // (-DLM_NCH=8: all eight channels in one traversal, A/B)
                    hipLaunchKernelGGL((k_adjoint_binned_lm4<SL, true>), pgrid, dim3(LM4_THREADS), l4, c->stream, g, pl.d_uray,
                                       pl.d_entries, pl.d_units, wr, Ns, c->d_unitw, c->d_G8, n, rb, fixmax, pl.fix_bits, pl.n_units,
                                       (int *)(c->d_fixgrid + n + 1) + rb);
            });
            HIP_TRY(c, hipGetLastError());
        }
        for (int f = 0; f < LM_NF && !fix_cubic; ++f) {
            const int rc = launch_adjoint_tile<double, MODE, true>(c, g, o, d, order, w, tec, dobs, cdct, Na, NtNd, i0, R, tmax, Ns,
                                                                   c->d_G8 + (size_t)f * n, f);
            if (rc) return rc;
        }
        if (!c->d_LMw) HIP_TRY(c, hipMalloc((void **)&c->d_LMw, (size_t)n * 6 * sizeof(double)));
        // scratch [6 n]: H0 | H1 (double2 each) | K0 | K1
        double2 *H0 = (double2 *)c->d_LMw, *H1 = H0 + n;
        double *K0 = c->d_LMw + 4 * n, *K1 = K0 + n;
        if (tiled) {
            hipLaunchKernelGGL((k_lm_fold_z_tiles<true>), dim3(pl.tile_n[0]), dim3(LMT_NODES / 2), 0, c->stream, c->d_G8, H0, H1, pl.d_tiles + pl.tile_off[0], tg,
                               fixmax, pl.fix_bits);
            hipLaunchKernelGGL(k_lm_fold_y_tiles, dim3(pl.tile_n[1]), dim3(256), 0, c->stream, (const double2 *)H0, (const double2 *)H1, K0, K1,
                               pl.d_tiles + pl.tile_off[1], tg);
            hipLaunchKernelGGL((k_lm_fold_x_tiles<AT>), dim3(pl.tile_n[2]), dim3(LMT_NODES / 2), 0, c->stream, (const double *)K0, (const double *)K1, grad,
                               pl.d_tiles + pl.tile_off[2], tg);
            HIP_TRY(c, hipGetLastError());
            return IONO_OK;
        }
        hipLaunchKernelGGL(k_lm_fold_z, dim3(ew_blocks(c, n)), dim3(256), 0, c->stream, c->d_G8, H0, H1, c->nx, c->ny, c->nz);
        hipLaunchKernelGGL(k_lm_fold_y, dim3(ew_blocks(c, n)), dim3(256), 0, c->stream, (const double2 *)H0, (const double2 *)H1, K0, K1, c->nx,
                           c->ny, c->nz);
        hipLaunchKernelGGL((k_lm_fold_x<AT>), dim3(ew_blocks(c, n)), dim3(256), 0, c->stream, (const double *)K0, (const double *)K1, grad, c->nx,
                           c->ny, c->nz);
        HIP_TRY(c, hipGetLastError());
        return IONO_OK;
    }
    const dim3 grid(ray_grid_blocks(c, R)), block(256);
    const size_t lds = lds_bytes(c);
    if (kind == IONO_INTERP_TRILINEAR)
        hipLaunchKernelGGL((k_adjoint_straight<AT, MODE, IONO_INTERP_TRILINEAR>), grid, block, lds, c->stream, g, o, d, w, tec, dobs,
                           cdct, Na, NtNd, i0, R, tmax, Ns, c->d_unitw, grad, c->d_flags);
    else
        hipLaunchKernelGGL((k_adjoint_straight<AT, MODE, IONO_INTERP_TRICUBIC>), grid, block, lds, c->stream, g, o, d, w, tec, dobs,
                           cdct, Na, NtNd, i0, R, tmax, Ns, c->d_unitw, grad, c->d_flags);
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

extern "C" {

static int adjoint_straight_launch(iono_ctx *c, int mode, const double *o, const double *d, const int *order, const double *w,
                                   const double *tec, const double *dobs, const double *cdct, int Na, int64_t NtNd, int i0,
                                   int64_t R, double tmax, int Ns, int kind, int rule, void *grad, int accum) {
    int rc = check_common(c, R, Ns, kind, rule);
    if (rc) return rc;
    if (accum != IONO_F64 && accum != IONO_F32) return fail(c, IONO_ERR_ARG, "bad accum_dtype");
    if (R == 0) {
        c->unit_lo = c->unit_hi = -1;      // (a range never outlives the launch it was set for)
        return IONO_OK;
    }
    rc = ensure_unitw(c, Ns, rule);
    if (rc) return rc;
    const GridView g = view(c);
#define ADJ_CALL(AT, MODE) \
    adjoint_straight_typed<AT, MODE>(c, g, o, d, order, w, tec, dobs, cdct, Na, NtNd, i0, R, tmax, Ns, kind, (AT *)grad)
    if (accum == IONO_F64) return mode == 0 ? ADJ_CALL(double, 0) : mode == 1 ? ADJ_CALL(double, 1) : ADJ_CALL(double, 2);
    return mode == 0 ? ADJ_CALL(float, 0) : mode == 1 ? ADJ_CALL(float, 1) : ADJ_CALL(float, 2);
#undef ADJ_CALL
}

int iono_adjoint_straight_dev(iono_ctx *c, const double *o, const double *d, const int *order, const double *w, int64_t R,
                              double tmax, int Ns, int kind, int rule, void *grad, int accum) {
    return adjoint_straight_launch(c, 0, o, d, order, w, nullptr, nullptr, nullptr, 1, R, 0, R, tmax, Ns, kind, rule, grad, accum);
}

int iono_adjoint_residual_straight_dev(iono_ctx *c, const double *o, const double *d, const int *order, const double *tec,
                                       const double *dobs, const double *cdct, int Na, int64_t NtNd, int i0, double tmax, int Ns,
                                       int kind, int rule, void *grad, int accum) {
    if (Na < 1 || NtNd < 0 || i0 < 0 || i0 >= Na) return fail(c, IONO_ERR_ARG, "bad [Na][NtNd]/i0");
    return adjoint_straight_launch(c, 1, o, d, order, nullptr, tec, dobs, cdct, Na, NtNd, i0, (int64_t)Na * NtNd, tmax, Ns,
                                   kind, rule, grad, accum);
}

int iono_adjoint_differential_straight_dev(iono_ctx *c, const double *o, const double *d, const int *order, const double *v,
                                           const double *scale, int Na, int64_t NtNd, int i0, double tmax, int Ns, int kind,
                                           int rule, void *grad, int accum) {
    if (Na < 1 || NtNd < 0 || i0 < 0 || i0 >= Na || !v) return fail(c, IONO_ERR_ARG, "bad [Na][NtNd]/i0/v");
    return adjoint_straight_launch(c, 2, o, d, order, nullptr, v, nullptr, scale, Na, NtNd, i0, (int64_t)Na * NtNd, tmax, Ns,
                                   kind, rule, grad, accum);
}

// ---- a solver iteration's ray pass + back-projection: one small launch + the back-projection (include/ionotomo_hip.h) ----------
static int rays_step_then_adjoint(iono_ctx *c, int mode, const double *o, const double *d, const int *order, const double *tq,
                                  const double *dobs, const double *scale, const double *weight, double *r, const double *an, int ann,
                                  const double *ad, int adn, int Na, int64_t NtNd, int i0, double tmax, int Ns, int kind, int rule,
                                  double *partial, void *grad, int accum) {
    const int64_t R = (int64_t)Na * NtNd;
    int rc = check_common(c, R, Ns, kind, rule);
    if (rc) return rc;
    if (Na < 1 || NtNd < 0 || i0 < 0 || i0 >= Na || !tq) return fail(c, IONO_ERR_ARG, "bad [Na][NtNd] / i0 / vector");
    if (ann > IONO_NPART || adn > IONO_NPART) return fail(c, IONO_ERR_ARG, "scalar count too large");
    if (R == 0) return IONO_OK;
    if (c->rayw_cap < R) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->d_rayw) (void)hipFree(c->d_rayw);
        c->d_rayw = nullptr, c->rayw_cap = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_rayw, (size_t)R * sizeof(double)));
        c->rayw_cap = R;
    }
    const iono_ctx::AdjPlan &pl = c->plan;
    const bool planned = facts_of(c, IONO_OP_ADJOINT, o, d, R, tmax, Ns, kind, kind, 0).adj_planned && c->variant != 2;
    const double *vo = planned ? o : nullptr;
    const unsigned nblk = (unsigned)std::min<int64_t>(IONO_NPART, (NtNd + RSTEP_PT - 1) / RSTEP_PT);
    if (mode == 0)
        hipLaunchKernelGGL((k_rays_step<0>), dim3(nblk), dim3(64 * RSTEP_WAVES), 0, c->stream, tq, dobs, scale, weight, r, an, ann, ad, adn, Na,
                           NtNd, i0, c->d_rayw, partial, IONO_NPART, vo, d, (const uint2 *)pl.d_hash, pl.d_uray, c->d_flags);
    else
        hipLaunchKernelGGL((k_rays_step<1>), dim3(nblk), dim3(64 * RSTEP_WAVES), 0, c->stream, tq, dobs, scale, weight, r, an, ann, ad, adn, Na,
                           NtNd, i0, c->d_rayw, partial, IONO_NPART, vo, d, (const uint2 *)pl.d_hash, pl.d_uray, c->d_flags);
    HIP_TRY(c, hipGetLastError());
    c->plan_verified = planned;
    if (!grad) return IONO_OK;      // (the ray pass alone: iono_adjoint_planned_weights_dev back-projects its weights, slab by slab)
    return adjoint_straight_launch(c, 0, o, d, order, c->d_rayw, nullptr, nullptr, nullptr, 1, R, 0, R, tmax, Ns, kind, rule, grad, accum);
}
int iono_adjoint_planned_weights_dev(iono_ctx *c, const double *o, const double *d, const int *order, int64_t R, double tmax, int Ns, int kind,
                                     int rule, void *grad, int accum) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    if (!grad || !c->d_rayw || c->rayw_cap < R) return fail(c, IONO_ERR_ARG, "iono_adjoint_planned_weights_dev: no weights of a *_step call for these rays");
    c->plan_verified = true;        // (checked by the step that formed the weights)
    return adjoint_straight_launch(c, 0, o, d, order, c->d_rayw, nullptr, nullptr, nullptr, 1, R, 0, R, tmax, Ns, kind, rule, grad, accum);
}
int iono_set_deterministic(iono_ctx *c, int on) {
    if (!c) return fail(c, IONO_ERR_ARG, "null context");
    c->deterministic = on != 0;
    return IONO_OK;
}
int iono_adjoint_plan_slabs(iono_ctx *c, int nslab) {
    if (!c || nslab < 1 || nslab > 8) return fail(c, IONO_ERR_ARG, "iono_adjoint_plan_slabs: 1 <= nslab <= 8");
    c->plan_slabs = nslab;
    return IONO_OK;
}
int iono_adjoint_plan_slab_info(iono_ctx *c, int *nslab, int *unit_lo, int *z_lo) {
    if (!c || !nslab) return fail(c, IONO_ERR_ARG, "null argument");
    const iono_ctx::AdjPlan &pl = c->plan;
    *nslab = pl.R >= 0 ? pl.nslab : 0;
    for (int s = 0; s <= *nslab; ++s) {
        if (unit_lo) unit_lo[s] = pl.slab_unit[s];
        if (z_lo) z_lo[s] = pl.slab_z[s];
    }
    return IONO_OK;
}
int iono_adjoint_unit_range(iono_ctx *c, int lo, int hi) {
    if (!c || lo < 0 || hi < lo) return fail(c, IONO_ERR_ARG, "iono_adjoint_unit_range: 0 <= lo <= hi");
    c->unit_lo = lo, c->unit_hi = hi;
    return IONO_OK;
}
int iono_adjoint_cg_step_dev(iono_ctx *c, const double *o, const double *d, const int *order, double *r, const double *q,
                             const double *an, int ann, const double *ad, int adn, const double *scale, int Na, int64_t NtNd, int i0,
                             double tmax, int Ns, int kind, int rule, double *partial, void *grad, int accum) {
    if (!r || !an || !ad) return fail(c, IONO_ERR_ARG, "iono_adjoint_cg_step_dev: r, an, ad are required");
    return rays_step_then_adjoint(c, 0, o, d, order, q, nullptr, scale, nullptr, r, an, ann, ad, adn, Na, NtNd, i0, tmax, Ns, kind, rule,
                                  partial, grad, accum);
}
int iono_adjoint_sirt_step_dev(iono_ctx *c, const double *o, const double *d, const int *order, const double *tec, const double *dobs,
                               const double *scale, const double *weight, int Na, int64_t NtNd, int i0, double tmax, int Ns, int kind,
                               int rule, double *r_out, double *partial, void *grad, int accum) {
    if (!dobs) return fail(c, IONO_ERR_ARG, "iono_adjoint_sirt_step_dev: dobs is required");
    return rays_step_then_adjoint(c, 1, o, d, order, tec, dobs, scale, weight, r_out, nullptr, 0, nullptr, 0, Na, NtNd, i0, tmax, Ns, kind,
                                  rule, partial, grad, accum);
}

int iono_adjoint_rays_dev(iono_ctx *c, const double *rays, const double *w, int64_t R, int Ns, int kind, int rule, void *grad,
                          int accum) {
    int rc = check_common(c, R, Ns, kind, rule);
    if (rc) return rc;
    if (accum != IONO_F64 && accum != IONO_F32) return fail(c, IONO_ERR_ARG, "bad accum_dtype");
    if (c->deterministic) return fail(c, IONO_ERR_ARG, "deterministic mode serves the planned straight-ray back-projections (trilinear, tricubic) only: not this entry point");
    if (R == 0) return IONO_OK;
    const GridView g = view(c);
    const dim3 grid(ray_grid_blocks(c, R)), block(256);
    const size_t lds = lds_bytes(c);
#define LAUNCH_AR(AT, K) \
    hipLaunchKernelGGL((k_adjoint_rays<AT, K>), grid, block, lds, c->stream, g, rays, w, R, Ns, rule, (AT *)grad, c->d_flags)
    if (accum == IONO_F64) {
        if (kind == IONO_INTERP_TRILINEAR) LAUNCH_AR(double, IONO_INTERP_TRILINEAR); else LAUNCH_AR(double, IONO_INTERP_TRICUBIC);
    } else {
        if (kind == IONO_INTERP_TRILINEAR) LAUNCH_AR(float, IONO_INTERP_TRILINEAR); else LAUNCH_AR(float, IONO_INTERP_TRICUBIC);
    }
#undef LAUNCH_AR
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

// ---- phase observable on the device (inversion/iterative_newton.py:86-127) and its adjoint -------------------------------
static int phase_freqs_dev(iono_ctx *c, const double *freqs, int Nf) {
    if (!freqs || Nf < 1 || Nf > 4096) return fail(c, IONO_ERR_ARG, "bad frequency list");
    for (int l = 0; l < Nf; ++l)
        if (!(freqs[l] > 0)) return fail(c, IONO_ERR_ARG, "frequencies must be positive");
    if (c->d_freqs && (int)c->h_freqs.size() == Nf && std::equal(freqs, freqs + Nf, c->h_freqs.begin())) return IONO_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_freqs) (void)hipFree(c->d_freqs);
    c->d_freqs = nullptr;
    HIP_TRY(c, hipMalloc((void **)&c->d_freqs, sizeof(double) * Nf));
    HIP_TRY(c, hipMemcpy(c->d_freqs, freqs, sizeof(double) * Nf, hipMemcpyHostToDevice));
    c->h_freqs.assign(freqs, freqs + Nf);
    return IONO_OK;
}
static PhaseFreqs phase_chunk(const double *freqs, int f0, int Nf) {
    PhaseFreqs pf;
    pf.nf = std::min(8, Nf - f0);
    for (int l = 0; l < 8; ++l) pf.inv_np[l] = l < pf.nf ? 1.0 / (1.2404e-2 * freqs[f0 + l] * freqs[f0 + l]) : 0.0;     // iterative_newton.py:112
    return pf;
}

int iono_forward_phase_straight_dev(iono_ctx *c, const double *o, const double *d, int Na, int Nt, int Nd, double tmax, int Ns,
                                    const double *freqs, int Nf, const double *clock, const double *cst, int i0, int rule,
                                    double *phi_work, double *gout) {
    const int64_t R = (int64_t)Na * Nt * Nd;
    int rc = check_common(c, R, Ns, IONO_INTERP_TRILINEAR, rule);
    if (rc) return rc;
    if (i0 < 0 || i0 >= Na || !clock || !cst || !phi_work || !gout) return fail(c, IONO_ERR_ARG, "iono_forward_phase_straight_dev: bad argument");
    rc = phase_freqs_dev(c, freqs, Nf);
    if (rc) return rc;
    if (R == 0) return IONO_OK;
    rc = ensure_unitw(c, Ns, rule);
    if (rc) return rc;
    const GridView g = view(c);
    const dim3 grid(ray_grid_blocks(c, R)), block(256);
    const PhaseFwdKernel pfk = pick_phase_forward(facts_of(c, IONO_OP_PHASE_FORWARD, o, d, R, tmax, Ns, IONO_INTERP_TRILINEAR, IONO_INTERP_TRILINEAR, 0));
    for (int f0 = 0; f0 < Nf; f0 += 8) {
        const PhaseFreqs pf = phase_chunk(freqs, f0, Nf);
        // lanes = samples on ideal-uniform grids for the rays ord[0 .. Rn) (null: all of them)
        auto phase_u = [&](const int *ord, int64_t Rn, hipStream_t st) {
            dispatch_storage(c, [&](auto *tag) {
                using GT = std::remove_pointer_t<decltype(tag)>;
                const size_t wl = sizeof(double) * Ns;
#define PHASE_U(NF)                                                                                                                  \
    hipLaunchKernelGGL((k_forward_phase_u<GT, NF>), dim3(chunk_grid_blocks(resident_blocks(c, k_forward_phase_u<GT, NF>, wl), Rn)), block, \
                       wl, st, g, o, d, ord, Rn, tmax, Ns, c->d_unitw, pf, Nf, phi_work + f0, c->d_flags)
                if (pf.nf == 1) PHASE_U(1);
                else if (pf.nf <= 4) PHASE_U(4);          // (two frequencies: the 4-slot series kernel beats two plain roots)
                else PHASE_U(8);
#undef PHASE_U
                return IONO_OK;
            });
        };
        if (pfk == PF_BUNDLE) {      // bundle-stationary (k_forward_bundle<NF>: windows in LDS), as the TEC forward
            const iono_ctx::FwdPlan &fp = c->fplan;
#define PHASE_B(NF)                                                                                                                        \
    hipLaunchKernelGGL((k_forward_bundle<NF>), dim3((unsigned)((fp.nb + 7) / 8 * 8)), block,                                               \
                       B_SPLIT * B_WAVE_LDS + NF * B_SPLIT * 64 * sizeof(double) + 16, c->stream, g, o, d, fp.d_brec, fp.d_bhash, fp.d_win, \
                       fp.nb, fp.nchunks, tmax, Ns, c->d_unitw, phi_work + f0, c->d_flags, pf, Nf)
            if (pf.nf == 1) PHASE_B(1);
            else if (pf.nf <= 4) PHASE_B(4);
            else PHASE_B(8);
#undef PHASE_B
            if (fp.n_rest > 0) phase_u(fp.d_order + fp.n_planned, fp.n_rest, c->stream);      // (hybrid: the rays outside the served bundles)
            continue;
        }
        if (pfk == PF_U) {
            phase_u(nullptr, R, c->stream);
            continue;
        }
        dispatch_storage(c, [&](auto *tag) {
            using GT = std::remove_pointer_t<decltype(tag)>;
            hipLaunchKernelGGL((k_forward_phase_straight<GT, false>), grid, block, lds_bytes(c), c->stream, g, o, d, R, tmax, Ns,
                               c->d_unitw, pf, Nf, phi_work + f0, c->d_flags);
            return IONO_OK;
        });
    }
    launch_map<PhaseFinish>(c, ew_blocks(c, R * Nf), (int64_t)Na * Nt * Nd * Nf, phi_work, c->d_freqs, clock, cst, Nt, Nd, Nf, i0, gout);
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

int iono_adjoint_phase_straight_dev(iono_ctx *c, const double *o, const double *d, const int *order, const double *y, int Na,
                                    int64_t NtNd, double tmax, int Ns, const double *freqs, int Nf, int i0, int rule,
                                    double *wrf_work, int wrt_log_model, double *grad) {
    const int64_t R = (int64_t)Na * NtNd;
    int rc = check_common(c, R, Ns, IONO_INTERP_TRILINEAR, rule);
    if (rc) return rc;
    if (i0 < 0 || i0 >= Na || !y || !wrf_work || !grad) return fail(c, IONO_ERR_ARG, "iono_adjoint_phase_straight_dev: bad argument");
    if (c->deterministic) return fail(c, IONO_ERR_ARG, "deterministic mode serves the planned straight-ray back-projections (trilinear, tricubic) only: not this entry point");
    rc = phase_freqs_dev(c, freqs, Nf);
    if (rc) return rc;
    if (R == 0) return IONO_OK;
    rc = ensure_unitw(c, Ns, rule);
    if (rc) return rc;
    const GridView g = view(c);
    launch_map<PhaseWeights>(c, ew_blocks(c, R * Nf), (int64_t)Na * NtNd * Nf, y, c->d_freqs, Na, NtNd, Nf, i0, wrf_work);
    const PhaseAdjKernel pak = pick_phase_adjoint(facts_of(c, IONO_OP_PHASE_ADJOINT, o, d, R, tmax, Ns, IONO_INTERP_TRILINEAR, IONO_INTERP_TRILINEAR, 0));
    const bool tiled = pak == PA_TILE, planned = pak == PA_BINNED;
    const iono_ctx::AdjPlan &pl = c->plan;
    if (planned && pl.n_invalid > 0) HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)c->d_flags, 1, 1, c->stream));   // out-of-grid rays
    if (planned)      // the rays handed over are still the planned ones? (plan_verify_ray)
        hipLaunchKernelGGL(k_plan_verify, dim3(ew_blocks(c, R)), dim3(256), 0, c->stream, o, d, R, (const uint2 *)pl.d_hash, pl.d_uray, c->d_flags);
    for (int f0 = 0; f0 < Nf; f0 += 8) {
        const PhaseFreqs pf = phase_chunk(freqs, f0, Nf);
        rc = dispatch_storage(c, [&](auto *tag) -> int {
            using GT = std::remove_pointer_t<decltype(tag)>;
            if (planned) {      // node-stationary: box images in LDS, ne gathered per sample (iono_binned_kernels.h)
                const size_t bl = sizeof(double) * ((((size_t)Ns + 1) & ~(size_t)1) + BIN_TILE);
#define PHASE_BIN(NF)                                                                                                              \
    BY_SEGL(pl.segl, hipLaunchKernelGGL((k_adjoint_binned<double, NF, GT, SL>), dim3(pl.n_units), dim3(BIN_THREADS), bl, c->stream, g, \
                                        pl.d_uray, pl.d_entries, pl.d_units, wrf_work + f0, Ns, c->d_unitw, grad, pf, Nf))
                if (pf.nf == 1) PHASE_BIN(1);
                else if (pf.nf == 2) PHASE_BIN(2);
                else if (pf.nf <= 4) PHASE_BIN(4);
                else PHASE_BIN(8);
#undef PHASE_BIN
                return IONO_OK;
            }
            if (tiled)
                return launch_adjoint_tile<double, 0, false, true, GT>(c, g, o, d, order, wrf_work + f0, nullptr, nullptr, nullptr, Na,
                                                                       NtNd, i0, R, tmax, Ns, grad, -1, pf, Nf);
            hipLaunchKernelGGL((k_adjoint_phase_straight<GT, double>), dim3(ray_grid_blocks(c, R)), dim3(256), lds_bytes(c), c->stream,
                               g, o, d, wrf_work + f0, Nf, pf, R, tmax, Ns, c->d_unitw, grad, c->d_flags);
            return IONO_OK;
        });
        if (rc) return rc;
    }
    if (wrt_log_model) {
        const int64_t n = ncells(c);
        dispatch_storage(c, [&](auto *tag) {
            using GT = std::remove_pointer_t<decltype(tag)>;
            launch_map<ScaleByGrid<double, GT>>(c, ew_blocks(c, n), n, grad, (const GT *)cur_values(c));
            return IONO_OK;
        });
    }
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

// ---- the reference's shipped chord-length gradient (A7): do_gradient of inversion/gradient.py:15-20 -----------------------
int iono_gradient_chords_dev(iono_ctx *c, const double *rays, const double *dd, int64_t R, int Ns, double *grad) {
    int rc = check_common(c, R, Ns, IONO_INTERP_TRILINEAR, 0);
    if (rc) return rc;
    if (!rays || !dd || !grad) return fail(c, IONO_ERR_ARG, "iono_gradient_chords_dev: null argument");
    if (R == 0) return IONO_OK;
    const GridView g = view(c);
    dispatch_storage(c, [&](auto *tag) {
        using GT = std::remove_pointer_t<decltype(tag)>;
        hipLaunchKernelGGL((k_gradient_chords<GT, double>), dim3(ray_grid_blocks(c, R)), dim3(256), lds_bytes(c), c->stream, g, rays, dd, R,
                           Ns, grad);
        return IONO_OK;
    });
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

int iono_gradient_chords(iono_ctx *c, const double *rays, const double *dd, int64_t R, int Ns, double *grad_out) {
    int rc = check_common(c, R, Ns, IONO_INTERP_TRILINEAR, 0);
    if (rc) return rc;
    const int64_t n = ncells(c);
    DevBuf b(c);
    const size_t nr = (size_t)R * 4 * Ns;
    HIP_TRY(c, b.alloc(8 * (nr + (size_t)R + (size_t)n)));
    double *dR = b.as<double>(), *dW = dR + nr, *dG = dW + R;
    HIP_TRY(c, hipMemcpyAsync(dR, rays, nr * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dW, dd, R * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(dG, 0, n * 8, c->stream));
    rc = iono_gradient_chords_dev(c, dR, dW, R, Ns, dG);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(grad_out, dG, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return IONO_OK;
}

int iono_check_oob(iono_ctx *c, int *oob) {
    if (!c || !oob) return fail(c, IONO_ERR_ARG, "null argument");
    return read_flag(c, 0, oob);
}
// 1 (and the flag is cleared) if a planned launch since the last call met rays that are not the ones its plan was made for -- the
// caller edited a planned array in place.  The forward then took the direct loads (its results are exact whatever the bundling); a
// back-projection poisoned the edited rays' weights with NaN.  Waits for the ctx stream.
#ifdef IONO_B_STAMP      // timing-only build: the forward bundle kernel's in-kernel stamps (8 x uint64 per wave, 4 waves per workgroup)
extern "C" int iono_debug_bundle_stamps(iono_ctx *c, unsigned long long *out, size_t count) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bstamp), count * sizeof(unsigned long long)));
    return IONO_OK;
}
#endif
int iono_plan_stale(iono_ctx *c, int *stale) {
    if (!c || !stale) return fail(c, IONO_ERR_ARG, "null argument");
    return read_flag(c, 2, stale);
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
    { const int rc = need_ctx(c); if (rc) return rc; }
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
    launch_map<PhaseFinish>(c, ew_blocks(c, (int64_t)nphi), (int64_t)Na * Nt * Nd * Nf, dPhi, dF, dClock, dConst, Nt, Nd, Nf, i0, dG);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(gout, dG, nphi * 8, hipMemcpyDeviceToHost, c->stream));
    return finish_host_call(c, "iono_forward_phase_rays");
}

static int adjoint_host_finish(iono_ctx *c, double *dG, int scale_by_grid, double *grad_out, const char *what) {
    const int64_t n = ncells(c);
    if (scale_by_grid)
        dispatch_storage(c, [&](auto *tag) {
            using GT = std::remove_pointer_t<decltype(tag)>;
            launch_map<ScaleByGrid<double, GT>>(c, ew_blocks(c, n), n, dG, (const GT *)cur_values(c));
            return IONO_OK;
        });
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(grad_out, dG, n * 8, hipMemcpyDeviceToHost, c->stream));
    return finish_host_call(c, what);
}

// ---- device memory for hosts that keep operands resident between calls without torch (the reference-signature facade) ---------
int iono_dev_alloc(iono_ctx *c, size_t bytes, void **out) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    if (!out) return fail(c, IONO_ERR_ARG, "null out pointer");
    *out = nullptr;
    HIP_TRY(c, hipMalloc(out, bytes ? bytes : 8));
    return IONO_OK;
}
int iono_dev_free(iono_ctx *c, void *p) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    if (!p) return IONO_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipFree(p));
    return IONO_OK;
}
int iono_dev_upload(iono_ctx *c, void *dst_dev, const void *src_host, size_t bytes) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    if (bytes && (!dst_dev || !src_host)) return fail(c, IONO_ERR_ARG, "iono_dev_upload: null pointer");
    HIP_TRY(c, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return IONO_OK;
}
int iono_dev_zero(iono_ctx *c, void *p_dev, size_t bytes) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    if (bytes && !p_dev) return fail(c, IONO_ERR_ARG, "iono_dev_zero: null pointer");
    HIP_TRY(c, hipMemsetAsync(p_dev, 0, bytes, c->stream));
    return IONO_OK;
}
// also the end of a chain of _dev launches: waits for the ctx stream and turns an out-of-grid flag into IONO_ERR_OOB
int iono_dev_download(iono_ctx *c, void *dst_host, const void *src_dev, size_t bytes) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    if (bytes && (!dst_host || !src_dev)) return fail(c, IONO_ERR_ARG, "iono_dev_download: null pointer");
    if (bytes <= ((size_t)1 << 20)) {
        // small results (a [Na,Nt,Nd] dTEC is 20 kB): payload and flags through pinned memory, ONE wait on the stream
        if (c->pinned_cap < bytes + 16) {
            if (c->h_pinned) (void)hipHostFree(c->h_pinned);
            c->h_pinned = nullptr, c->pinned_cap = 0;
            HIP_TRY(c, hipHostMalloc((void **)&c->h_pinned, ((size_t)1 << 20) + 16, hipHostMallocDefault));
            c->pinned_cap = ((size_t)1 << 20) + 16;
        }
        int *hf = (int *)(c->h_pinned + ((bytes + 15) & ~(size_t)15));
        HIP_TRY(c, hipMemcpyAsync(c->h_pinned, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(hf, c->d_flags, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        memcpy(dst_host, c->h_pinned, bytes);
        if (hf[0]) {
            hf[0] = 0;
            HIP_TRY(c, hipMemcpy(c->d_flags, hf, 2 * sizeof(int), hipMemcpyHostToDevice));
            return fail(c, IONO_ERR_OOB, "iono_dev_download: One of the requested xi is out of bounds");
        }
        return IONO_OK;
    }
    HIP_TRY(c, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    return finish_host_call(c, "iono_dev_download");
}
// grad *= grid values: d/dm = ne d/dne for the log-model (what the host entry points' scale_by_grid does)
int iono_scale_by_grid_dev(iono_ctx *c, double *grad_dev) {
    int rc = need_grid(c);
    if (rc) return rc;
    if (!grad_dev) return fail(c, IONO_ERR_ARG, "iono_scale_by_grid_dev: null pointer");
    const int64_t n = ncells(c);
    dispatch_storage(c, [&](auto *tag) {
        using GT = std::remove_pointer_t<decltype(tag)>;
        launch_map<ScaleByGrid<double, GT>>(c, ew_blocks(c, n), n, grad_dev, (const GT *)cur_values(c));
        return IONO_OK;
    });
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

int iono_walk_order(iono_ctx *c, const double *o, const double *d, int64_t R, double tmax, int *order_out) {
    int rc = need_grid(c);
    if (rc) return rc;
    if (R < 0 || R > INT32_MAX || (R > 0 && (!o || !d || !order_out))) return fail(c, IONO_ERR_ARG, "iono_walk_order: bad argument");
    morton_walk_order(c, o, d, R, tmax, order_out);
    return IONO_OK;
}

int iono_adjoint_straight(iono_ctx *c, const double *o, const double *d, const double *w, int64_t R, double tmax, int Ns,
                          int kind, int rule, int scale_by_grid, double *grad_out) {
    int rc = check_common(c, R, Ns, kind, rule);
    if (rc) return rc;
    const int64_t n = ncells(c);
    DevBuf b(c);
    HIP_TRY(c, b.alloc(8 * ((size_t)R * 8 + (size_t)n)));
    double *dO = b.as<double>(), *dD = dO + 3 * R, *dW = dD + 3 * R, *dG = dW + R;
    int *dOrder = nullptr;
    std::vector<int> order;
    if (ideal_path_ok(c, Ns) && R >= 1024 && R <= INT32_MAX) {
        // the tiled kernel pre-reduces bundles of consecutive rays of the walk: without a locality order a
        // 260k-ray batch takes 8 ms instead of 0.8 ms
        order.resize((size_t)R);
        morton_walk_order(c, o, d, R, tmax, order.data());
        dOrder = (int *)(dG + n);
        HIP_TRY(c, hipMemcpyAsync(dOrder, order.data(), R * 4, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipMemcpyAsync(dO, o, R * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dD, d, R * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dW, w, R * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(dG, 0, n * 8, c->stream));
    rc = iono_adjoint_straight_dev(c, dO, dD, dOrder, dW, R, tmax, Ns, kind, rule, dG, IONO_F64);
    if (rc) return rc;
    return adjoint_host_finish(c, dG, scale_by_grid, grad_out, "iono_adjoint_straight");
}

int iono_adjoint_rays(iono_ctx *c, const double *rays, const double *w, int64_t R, int Ns, int kind, int rule, int scale_by_grid,
                      double *grad_out) {
    int rc = check_common(c, R, Ns, kind, rule);
    if (rc) return rc;
    const int64_t n = ncells(c);
    DevBuf b(c);
    const size_t nr = (size_t)R * 4 * Ns;
    HIP_TRY(c, b.alloc(8 * (nr + (size_t)R + (size_t)n)));
    double *dR = b.as<double>(), *dW = dR + nr, *dG = dW + R;
    HIP_TRY(c, hipMemcpyAsync(dR, rays, nr * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dW, w, R * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(dG, 0, n * 8, c->stream));
    rc = iono_adjoint_rays_dev(c, dR, dW, R, Ns, kind, rule, dG, IONO_F64);
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
int iono_trace_straight(iono_ctx *c, const double *o, const double *d, int64_t R, double tmax, int Ns, int independent,
                        double *rays_out) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    if (R < 0 || Ns < 2) return fail(c, IONO_ERR_SHAPE, "need R >= 0 and Ns >= 2");
    if (independent != IONO_RAY_Z && independent != IONO_RAY_S) return fail(c, IONO_ERR_ARG, "bad independent variable");
    if (R == 0) return IONO_OK;
    DevBuf b(c);
    const size_t nr = (size_t)R * 4 * Ns;
    HIP_TRY(c, b.alloc(8 * (nr + 6 * (size_t)R)));
    double *dR = b.as<double>(), *dO = dR + nr, *dD = dO + 3 * R;
    HIP_TRY(c, hipMemcpyAsync(dO, o, R * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dD, d, R * 24, hipMemcpyHostToDevice, c->stream));
    launch_map<TraceStraight>(c, ew_blocks(c, R * Ns), R * Ns, dO, dD, tmax, Ns, independent, dR);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(rays_out, dR, nr * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return IONO_OK;
}

int iono_trace_fermat_dev(iono_ctx *c, const double *dO, const double *dD, int64_t R, double tmax, int Ns, double frequency,
                          int bend, int kind, int substeps, int independent, double *dR) {
    int rc = check_common(c, R, Ns, kind, 0);
    if (rc) return rc;
    if (independent != IONO_RAY_Z && independent != IONO_RAY_S) return fail(c, IONO_ERR_ARG, "bad independent variable");
    const int stype = independent;
    if (substeps < 1 || !(frequency > 0)) return fail(c, IONO_ERR_ARG, "need substeps >= 1 and frequency > 0");
    if (R == 0) return IONO_OK;
    const int64_t n = ncells(c);
    if (!c->d_nM) HIP_TRY(c, hipMalloc((void **)&c->d_nM, (size_t)n * 8));
    if (c->nM_freq != frequency) {       // n = sqrt(1 - 8.98^2 ne / nu^2) at the nodes, rebuilt when ne or nu changed
        dispatch_storage(c, [&](auto *tag) {
            using GT = std::remove_pointer_t<decltype(tag)>;
            launch_map<NeToN<GT>>(c, ew_blocks(c, n), n, (const GT *)cur_values(c), c->d_nM, frequency);
            return IONO_OK;
        });
        c->nM_freq = frequency;
        c->nF8_freq = -1.0;
    }
    const GridView g = view(c);
    const dim3 grid((unsigned)((R + 63) / 64)), block(64);
    double *dN = c->d_nM;
#define LAUNCH_F(K, B) \
    hipLaunchKernelGGL((k_trace_fermat<K, B>), grid, block, 0, c->stream, g, dN, dO, dD, R, tmax, Ns, substeps, dR, c->d_flags, stype)
    const size_t axes_bytes = (size_t)(c->nx + c->ny + c->nz) * 8;
    const iono_dispatch_facts facts = facts_of(c, IONO_OP_TRACE, nullptr, nullptr, R, tmax, Ns, kind, kind, bend);
    const TraceKernel tk = pick_tracer(facts);               // (the dispatch table: pick_tracer)
    if (tk == TK_POLY) {
        // small batch on an ideal-uniform grid: the cell's polynomial in registers, no DPP sums (iono_aux_kernels.h).  16 rays per
        // wave measured best from 600 to 4 096 rays (0.91-0.94 ms at config 3's 129 samples x 4 substeps; 8: 0.89-1.14, 32: 0.93-0.96,
        // 1: 3.0 -- a lone lane still pays the whole wave's issue slots; the 4-lanes-per-ray kernel: 1.08-1.18)
        const int rpw = c->fermat_poly_rpw > 0 ? c->fermat_poly_rpw : 16;
        const dim3 pgrid((unsigned)((R + rpw - 1) / rpw));
        if (bend)
            hipLaunchKernelGGL((k_trace_fermat_poly<true>), pgrid, block, 0, c->stream, g, dN, dO, dD, R, tmax, Ns, substeps, dR, c->d_flags,
                               rpw, stype);
        else
            hipLaunchKernelGGL((k_trace_fermat_poly<false>), pgrid, block, 0, c->stream, g, dN, dO, dD, R, tmax, Ns, substeps, dR, c->d_flags,
                               rpw, stype);
    } else if (tk == TK_LIN4) {
        // small batch: 4 lanes per ray, axes in LDS, corners cached per cell (latency-bound regime)
        const int rpw = c->fermat_lin4_rpw > 0 ? c->fermat_lin4_rpw : (R <= 4096 ? 4 : 16);   // measured
        const dim3 qgrid((unsigned)((R + rpw - 1) / rpw));
        if (bend)
            hipLaunchKernelGGL((k_trace_fermat_lin4<true>), qgrid, block, axes_bytes, c->stream, g, dN, dO, dD, R, tmax, Ns,
                               substeps, dR, c->d_flags, rpw, stype);
        else
            hipLaunchKernelGGL((k_trace_fermat_lin4<false>), qgrid, block, axes_bytes, c->stream, g, dN, dO, dD, R, tmax, Ns,
                               substeps, dR, c->d_flags, rpw, stype);
    } else if (tk == TK_RAYS && kind == IONO_INTERP_TRILINEAR) {
        if (bend) LAUNCH_F(IONO_INTERP_TRILINEAR, true); else LAUNCH_F(IONO_INTERP_TRILINEAR, false);
    } else if (tk == TK_RAYS) {   // lanes = rays: enough rays to fill the chip without splitting them
        if (bend) LAUNCH_F(IONO_INTERP_TRICUBIC, true); else LAUNCH_F(IONO_INTERP_TRICUBIC, false);
    } else if (tk == TK_LM8 || tk == TK_LM2) {
        // ideal-uniform grid: 8 lanes per ray, one Lekien-Marsden record of n per lane (iono_aux_kernels.h:k_trace_fermat_lm;
        // the 216-tap kernel below serves the other grids: IONOTOMO_FORCE_GENERAL=2 for A/B)
        const int rcf = ensure_n_fields(c, frequency);
        if (rcf) return rcf;
        const int lpr = tk == TK_LM8 ? 8 : 2;
        const int rpw = lpr == 8 && c->fermat_coop_rpw > 0 ? std::min(c->fermat_coop_rpw, 8) : 64 / lpr;
        const dim3 cgrid((unsigned)((R + rpw - 1) / rpw));
#define LAUNCH_TLM(B, L)                                                                                                                     \
    hipLaunchKernelGGL((k_trace_fermat_lm<B, L>), cgrid, block, 0, c->stream, g, (const double *)c->d_nF8, dO, dD, R, tmax, Ns, substeps, dR, \
                       c->d_flags, rpw, stype)
        if (bend) { if (lpr == 8) LAUNCH_TLM(true, 8); else LAUNCH_TLM(true, 2); }
        else { if (lpr == 8) LAUNCH_TLM(false, 8); else LAUNCH_TLM(false, 2); }
#undef LAUNCH_TLM
    } else {                                // 8 lanes per ray: the 6x6x6 stencil of one ray spread over 8 lanes
        const int rpw = c->fermat_coop_rpw > 0 ? c->fermat_coop_rpw : 8;
        const dim3 cgrid((unsigned)((R + rpw - 1) / rpw));
        const int in_lds = axes_bytes <= 48 * 1024;         // axis tables staged in LDS when they fit
        if (bend)
            hipLaunchKernelGGL((k_trace_fermat_coop<true>), cgrid, block, in_lds ? axes_bytes : 0, c->stream, g, dN, dO, dD, R,
                               tmax, Ns, substeps, dR, c->d_flags, in_lds, rpw, stype);
        else
            hipLaunchKernelGGL((k_trace_fermat_coop<false>), cgrid, block, in_lds ? axes_bytes : 0, c->stream, g, dN, dO, dD, R,
                               tmax, Ns, substeps, dR, c->d_flags, in_lds, rpw, stype);
    }
#undef LAUNCH_F
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

// ---- fused curved-ray forward / transpose (iono_fermat_kernels.h): trace and integrate in one traversal ---------------------------
// (which launches run on the Lekien-Marsden record stepper k_fermat_tec_lm: pick_fermat in the dispatch table)
static bool fermat_lm_ok(const iono_ctx *c, int kind_n, int kind_ne, int64_t R, bool transpose = false, int bend = 1) {
    const iono_dispatch_facts f = facts_of(c, transpose ? IONO_OP_FERMAT_ADJOINT : IONO_OP_FERMAT_FORWARD, nullptr, nullptr, R, 0.0, 2, kind_n, kind_ne, bend);
    return pick_fermat(f, transpose) != FT_RAYS;
}
int iono_fermat_lm_ok(iono_ctx *c, int kind_n, int kind_ne, int64_t R, int transpose, int bend, int *ok) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    if (!ok) return fail(c, IONO_ERR_ARG, "null argument");
    *ok = fermat_lm_ok(c, kind_n, kind_ne, R, transpose != 0, bend) ? 1 : 0;
    return IONO_OK;
}
static int fermat_tec_launch(iono_ctx *c, bool adjoint, const double *dO, const double *dD, const double *dW, int64_t R, double tmax, int Ns,
                             double frequency, int bend, int kind_n, int substeps, int independent, int kind_ne, int rule, double ne_scale,
                             double *tec, double *grad) {
    int rc = check_common(c, R, Ns, kind_n, rule);
    if (rc) return rc;
    rc = check_common(c, R, Ns, kind_ne, rule);
    if (rc) return rc;
    if (independent != IONO_RAY_Z && independent != IONO_RAY_S) return fail(c, IONO_ERR_ARG, "bad independent variable");
    if (substeps < 1 || !(frequency > 0)) return fail(c, IONO_ERR_ARG, "need substeps >= 1 and frequency > 0");
    if (c->storage != IONO_F64) return fail(c, IONO_ERR_ARG, "the fused curved-ray kernels need float64 grid storage");
    if (!dO || !dD || (adjoint ? (!dW || !grad) : !tec)) return fail(c, IONO_ERR_ARG, "null pointer");
    if (R == 0) return IONO_OK;
    const int64_t n = ncells(c);
    if (!c->d_nM) HIP_TRY(c, hipMalloc((void **)&c->d_nM, (size_t)n * 8));
    if (c->nM_freq != frequency) {       // n = sqrt(1 - 8.98^2 ne / nu^2) at the nodes, rebuilt when ne or nu changed
        launch_map<NeToN<double>>(c, ew_blocks(c, n), n, (const double *)cur_values(c), c->d_nM, frequency);
        c->nM_freq = frequency;
        c->nF8_freq = -1.0;
    }
    const GridView g = view(c);
    const dim3 grid((unsigned)((R + 63) / 64)), block(64);
    const FermatKernel ftk = pick_fermat(facts_of(c, adjoint ? IONO_OP_FERMAT_ADJOINT : IONO_OP_FERMAT_FORWARD, nullptr, nullptr, R, tmax, Ns, kind_n,
                                                   kind_ne, bend), adjoint);      // (the dispatch table: pick_fermat)
    if (ftk != FT_RAYS) {
        // tricubic index on an ideal-uniform grid: 8 lanes per ray, one Lekien-Marsden record of n per lane, streaming quadrature
        // (iono_fermat_kernels.h:k_fermat_tec_lm; IONOTOMO_VARIANT=3: the lanes = rays kernel below, A/B)
        const int rcf = ensure_n_fields(c, frequency);
        if (rcf) return rcf;
        const int lpr = ftk == FT_LM8 ? 8 : 2;
        const int rpw = lpr == 8 && c->fermat_coop_rpw > 0 ? std::min(c->fermat_coop_rpw, 8) : 64 / lpr;
        const dim3 cgrid((unsigned)((R + rpw - 1) / rpw));
#define LAUNCH_FLM(B, L)                                                                                                                       \
    hipLaunchKernelGGL((k_fermat_tec_lm<B, L>), cgrid, block, 0, c->stream, g, (const double *)c->d_nF8, dO, dD, R, tmax, Ns, substeps, rule,   \
                       independent, kind_ne, ne_scale, tec, c->d_flags, rpw, (const double *)nullptr, (double *)nullptr)
        if (adjoint)      // (fermat_lm_ok: bending rays, two lanes per ray)
            hipLaunchKernelGGL((k_fermat_tec_lm<true, 2, true>), cgrid, block, 0, c->stream, g, (const double *)c->d_nF8, dO, dD, R, tmax, Ns, substeps,
                               rule, independent, kind_ne, ne_scale, (double *)nullptr, c->d_flags, rpw, dW, grad);
        else if (bend) { if (lpr == 8) LAUNCH_FLM(true, 8); else LAUNCH_FLM(true, 2); }
        else { if (lpr == 8) LAUNCH_FLM(false, 8); else LAUNCH_FLM(false, 2); }
#undef LAUNCH_FLM
        HIP_TRY(c, hipGetLastError());
        return IONO_OK;
    }
    // axes (+ the wave's scatter window of the transpose: iono_fermat_kernels.h)
    const size_t lds = ((lds_bytes(c) + 15) & ~(size_t)15) + (adjoint ? sizeof(double) * FW * FW * FWZ : 0);
#define LAUNCH_FT(K, B, A)                                                                                                              \
    hipLaunchKernelGGL((k_fermat_tec<K, B, A>), grid, block, lds, c->stream, g, c->d_nM, dO, dD, R, tmax, Ns, substeps, rule, independent,   \
                       kind_ne, ne_scale, dW, tec, grad, c->d_flags, g.glast[2] + 1e-9 * std::fabs(tmax))
    if (kind_n == IONO_INTERP_TRILINEAR) {
        if (adjoint) { if (bend) LAUNCH_FT(IONO_INTERP_TRILINEAR, true, true); else LAUNCH_FT(IONO_INTERP_TRILINEAR, false, true); }
        else { if (bend) LAUNCH_FT(IONO_INTERP_TRILINEAR, true, false); else LAUNCH_FT(IONO_INTERP_TRILINEAR, false, false); }
    } else {
        if (adjoint) { if (bend) LAUNCH_FT(IONO_INTERP_TRICUBIC, true, true); else LAUNCH_FT(IONO_INTERP_TRICUBIC, false, true); }
        else { if (bend) LAUNCH_FT(IONO_INTERP_TRICUBIC, true, false); else LAUNCH_FT(IONO_INTERP_TRICUBIC, false, false); }
    }
#undef LAUNCH_FT
    HIP_TRY(c, hipGetLastError());
    return IONO_OK;
}

int iono_forward_tec_fermat_dev(iono_ctx *c, const double *o, const double *d, int64_t R, double tmax, int Ns, double frequency, int bend,
                                int kind_n, int substeps, int independent, int kind_ne, int rule, double ne_scale, double *tec) {
    return fermat_tec_launch(c, false, o, d, nullptr, R, tmax, Ns, frequency, bend, kind_n, substeps, independent, kind_ne, rule, ne_scale, tec,
                             nullptr);
}

int iono_adjoint_fermat_dev(iono_ctx *c, const double *o, const double *d, const double *w, int64_t R, double tmax, int Ns, double frequency,
                            int bend, int kind_n, int substeps, int independent, int kind_ne, int rule, double ne_scale, double *grad) {
    if (c && c->deterministic) return fail(c, IONO_ERR_ARG, "deterministic mode serves the planned straight-ray back-projections (trilinear, tricubic) only: not this entry point");
    return fermat_tec_launch(c, true, o, d, w, R, tmax, Ns, frequency, bend, kind_n, substeps, independent, kind_ne, rule, ne_scale, nullptr, grad);
}

int iono_trace_fermat(iono_ctx *c, const double *o, const double *d, int64_t R, double tmax, int Ns, double frequency, int bend,
                      int kind, int substeps, int independent, double *rays_out) {
    int rc = check_common(c, R, Ns, kind, 0);
    if (rc) return rc;
    if (R == 0) return IONO_OK;
    DevBuf b(c);
    const size_t nr = (size_t)R * 4 * Ns;
    HIP_TRY(c, b.alloc(8 * (nr + 6 * (size_t)R)));
    double *dR = b.as<double>(), *dO = dR + nr, *dD = dO + 3 * R;
    HIP_TRY(c, hipMemcpyAsync(dO, o, R * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dD, d, R * 24, hipMemcpyHostToDevice, c->stream));
    rc = iono_trace_fermat_dev(c, dO, dD, R, tmax, Ns, frequency, bend, kind, substeps, independent, dR);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(rays_out, dR, nr * 8, hipMemcpyDeviceToHost, c->stream));
    return finish_host_call(c, "iono_trace_fermat");
}

// ---- multi-GPU collective (RCCL over xGMI), for hosts that do not bring their own (include/ionotomo_hip.h) -------------
namespace {
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) get_id = nullptr;
    decltype(&ncclCommInitRank) init_rank = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclCommDestroy) destroy = nullptr;
    decltype(&ncclGetErrorString) err_str = nullptr;
};
Rccl *rccl(iono_ctx *c) {          // loaded on first use, kept for the life of the process
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (r.lib) {
            r.get_id = (decltype(r.get_id))dlsym(r.lib, "ncclGetUniqueId");
            r.init_rank = (decltype(r.init_rank))dlsym(r.lib, "ncclCommInitRank");
            r.all_reduce = (decltype(r.all_reduce))dlsym(r.lib, "ncclAllReduce");
            r.destroy = (decltype(r.destroy))dlsym(r.lib, "ncclCommDestroy");
            r.err_str = (decltype(r.err_str))dlsym(r.lib, "ncclGetErrorString");
        }
    }
    if (!r.lib || !r.get_id || !r.init_rank || !r.all_reduce || !r.destroy || !r.err_str) {
        fail(c, IONO_ERR_HIP, "RCCL (librccl.so) could not be loaded");
        return nullptr;
    }
    return &r;
}
int rccl_fail(iono_ctx *c, Rccl *r, ncclResult_t e, const char *what) {
    return fail(c, IONO_ERR_HIP, std::string(what) + ": " + r->err_str(e));
}
}  // namespace

int iono_comm_unique_id(iono_ctx *c, char id_out[IONO_COMM_ID_BYTES]) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    static_assert(sizeof(ncclUniqueId) == IONO_COMM_ID_BYTES, "id size");
    if (!id_out) return fail(c, IONO_ERR_ARG, "null id buffer");
    Rccl *r = rccl(c);
    if (!r) return IONO_ERR_HIP;
    ncclUniqueId id;
    const ncclResult_t e = r->get_id(&id);
    if (e != ncclSuccess) return rccl_fail(c, r, e, "ncclGetUniqueId");
    memcpy(id_out, id.internal, IONO_COMM_ID_BYTES);
    return IONO_OK;
}

int iono_comm_init(iono_ctx *c, const char id_in[IONO_COMM_ID_BYTES], int rank, int nranks) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    if (!id_in || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, IONO_ERR_ARG, "need an id and 0 <= rank < nranks");
    if (c->comm) return fail(c, IONO_ERR_ARG, "this context already has a communicator (iono_comm_destroy first)");
    Rccl *r = rccl(c);
    if (!r) return IONO_ERR_HIP;
    ncclUniqueId id;
    memcpy(id.internal, id_in, IONO_COMM_ID_BYTES);
    const ncclResult_t e = r->init_rank(&c->comm, nranks, id, rank);
    if (e != ncclSuccess) {
        c->comm = nullptr;
        return rccl_fail(c, r, e, "ncclCommInitRank");
    }
    c->comm_ranks = nranks;
    return IONO_OK;
}

int iono_comm_allreduce_dev(iono_ctx *c, void *buf, int64_t count, int dtype) {
    { const int rc = need_ctx(c); if (rc) return rc; }
    if (!c->comm) return fail(c, IONO_ERR_ARG, "no communicator: call iono_comm_init on every rank first");
    if (count < 0 || (count > 0 && !buf)) return fail(c, IONO_ERR_ARG, "bad buffer");
    if (dtype != IONO_F64 && dtype != IONO_F32) return fail(c, IONO_ERR_ARG, "dtype must be IONO_F64 or IONO_F32");
    if (count == 0) return IONO_OK;
    Rccl *r = rccl(c);
    if (!r) return IONO_ERR_HIP;
    const ncclResult_t e = r->all_reduce(buf, buf, (size_t)count, dtype == IONO_F64 ? ncclFloat64 : ncclFloat32, ncclSum, c->comm, c->stream);
    if (e != ncclSuccess) return rccl_fail(c, r, e, "ncclAllReduce");
    return IONO_OK;
}

int iono_comm_destroy(iono_ctx *c) {
    if (!c || !c->comm) return IONO_OK;
    Rccl *r = rccl(c);
    if (r) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)r->destroy(c->comm);
    }
    c->comm = nullptr;
    c->comm_ranks = 0;
    return IONO_OK;
}

}  // extern "C"
