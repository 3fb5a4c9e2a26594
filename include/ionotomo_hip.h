/*
 * ionotomo_hip.h -- C-ABI of libionotomo_hip.so: the MI355X (gfx950) ray-integral engine.
 *
 * The reference (Joshuaalbert/IonoTomo) is pure Python and has no FFI; the boundary of its
 * hot path is a set of Python call signatures.  Each entry point below names the reference
 * function (file:line, relative to /root/reference/src/ionotomo/) whose numeric work it
 * replaces; ionotomo_amd/ re-creates those Python signatures on top of this ABI via ctypes
 * (INTEGRATION.md shows the binding a reference maintainer would add).
 *
 * Conventions
 *   - plain C: opaque context, plain pointers and sizes, int return code (0 = OK, < 0 error;
 *     iono_last_error() gives the text).  Nothing throws or aborts across the ABI.
 *   - one iono_ctx per GPU per process, created AFTER fork; a ctx is not thread-safe,
 *     independent ctxs are.
 *   - "host" entry points take host pointers, run synchronously and return results in
 *     caller-allocated host arrays.  "_dev" entry points take DEVICE pointers, enqueue on the
 *     ctx stream (iono_ctx_set_stream lets the caller supply e.g. torch's current stream) and
 *     return immediately; the caller owns all buffers.  No library-allocated memory is ever
 *     returned.
 *   - all real data is float64 except the grid values, which may be STORED as float32
 *     (IONO_F32) -- arithmetic and accumulation stay float64.
 *   - grid values are C-ordered M[i][j][k] at flat index k + nz*(j + ny*i)
 *     (notebooks/TricubicInterpolation.ipynb c0:113-115).
 *   - rays are R independent rays; the Python layer flattens [Na,Nt,Nd] -> R.
 */
#ifndef IONOTOMO_HIP_H
#define IONOTOMO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct iono_ctx iono_ctx;

enum {
    IONO_OK = 0,
    IONO_ERR_OOB = -1,       /* a sample lies outside the grid  -> Python ValueError (scipy bounds_error=True, geometry/tri_cubic.py:22,59) */
    IONO_ERR_NONFINITE = -2, /* NaN/Inf in grid values          -> Python AssertionError (geometry/tri_cubic.py:51) */
    IONO_ERR_SHAPE = -3,
    IONO_ERR_HIP = -4,       /* HIP runtime failure             -> Python RuntimeError */
    IONO_ERR_ARG = -5
};

enum { IONO_F64 = 0, IONO_F32 = 1 };                     /* grid storage type */
enum { IONO_INTERP_TRILINEAR = 0,                        /* what TriCubic.interp ships: scipy RGI 'linear' (geometry/tri_cubic.py:69-70) */
       IONO_INTERP_TRICUBIC = 1 };                       /* Lekien-Marsden, 4th-order FD slopes (notebooks/TricubicInterpolation.ipynb c0:138-1257) */
enum { IONO_QUAD_SIMPSON_AVG = 0,                        /* odd N composite Simpson; even N reference-era simps(even='avg') (tomography/integrate.py:130-153) */
       IONO_QUAD_SIMPSON_SCIPY = 1,                      /* even N: scipy>=1.11 simpson (Cartwright correction) */
       IONO_QUAD_TRAPEZOID = 2 };

/* ---- context ------------------------------------------------------------------------------ */
int iono_ctx_create(int device_id, iono_ctx **out);
int iono_ctx_destroy(iono_ctx *ctx);
const char *iono_last_error(iono_ctx *ctx);              /* ctx may be NULL: last global error */
/* every launch goes to the handle given, used as is: NULL is HIP's null (legacy default) stream,
 * which is what torch.cuda.current_stream() is unless the caller switched streams */
int iono_ctx_set_stream(iono_ctx *ctx, void *hip_stream);
int iono_ctx_use_own_stream(iono_ctx *ctx);              /* back to the ctx's private non-blocking stream (the default) */
int iono_ctx_synchronize(iono_ctx *ctx);
int iono_version(void);

/* ---- grid: TriCubic(xvec,yvec,zvec,M) container (geometry/tri_cubic.py:13-59) ---------------- */
/* axes are strictly increasing, may be non-uniform; M (host, float64, nx*ny*nz) may be NULL to
 * allocate only.  storage = IONO_F64 | IONO_F32. */
int iono_grid_set(iono_ctx *ctx, const double *xvec, int nx, const double *yvec, int ny,
                  const double *zvec, int nz, const double *M, int storage);
int iono_grid_set_values(iono_ctx *ctx, const double *M);           /* TriCubic.M setter (:49-59); NaN/Inf -> IONO_ERR_NONFINITE */
int iono_grid_get_values(iono_ctx *ctx, double *M_out);
int iono_grid_set_values_dev(iono_ctx *ctx, const double *M_dev);   /* float64 device source, converted to the storage type */
/* M = scale * exp(m) at the NODES: ne = K_ne exp(m)/TECU (inversion/forward_equation.py:41-43),
 * ne = K exp(mu) (inversion/iterative_newton.py:104-106) */
int iono_grid_set_exp(iono_ctx *ctx, const double *m_host, double scale);
int iono_grid_set_exp_dev(iono_ctx *ctx, const double *m_dev, double scale);
void *iono_grid_values_ptr(iono_ctx *ctx);                          /* device pointer of the stored values */

/* ---- TriCubic.interp / .extrapolate (geometry/tri_cubic.py:69-75) --------------------------- */
int iono_interp(iono_ctx *ctx, const double *x, const double *y, const double *z, int64_t n,
                int interp_kind, int extrapolate, double *out);

/* ---- ray geometry: cast_ray / Fermat.integrate_ray (geometry/calc_rays.py:61-96,
 *      inversion/fermat.py:150-174).  rays_out[R][4][Ns] = x,y,z,s ----------------------------- */
/* independent: IONO_RAY_Z = Fermat(type='z') -- z = linspace(z0, tmax, Ns), the mode every reference call site uses;
 * IONO_RAY_S = Fermat(type='s') -- arc length s = linspace(0, tmax, Ns) (inversion/fermat.py:74-82,165-166). */
enum { IONO_RAY_Z = 0, IONO_RAY_S = 1 };
int iono_trace_straight(iono_ctx *ctx, const double *origins, const double *directions, int64_t R,
                        double tmax, int Ns, int independent, double *rays_out);
/* grid must hold ne [m^-3].  bend = 0 reproduces the shipped "curved" mode (grad n forced to 0,
 * fermat.py:54-55: straight x,y,z and s = int n/pz dz); bend = 1 integrates the true equations
 * (notebooks/FermatClass.ipynb c0:60-96) with fixed-step RK4, `substeps` steps per output sample. */
int iono_trace_fermat(iono_ctx *ctx, const double *origins, const double *directions, int64_t R,
                      double tmax, int Ns, double frequency, int bend, int interp_kind, int substeps,
                      int independent, double *rays_out);

/* ---- forward: tec[r] = simps(interp(M; x,y,z), s) (inversion/forward_equation.py:13-33) ------ */
/* samples generated in-kernel on straight z-parametrised rays (never materialises rays[R,4,Ns]) */
int iono_forward_tec_straight(iono_ctx *ctx, const double *origins, const double *directions,
                              int64_t R, double tmax, int Ns, int interp_kind, int quad_rule,
                              double *tec_out);
/* explicit samples rays[R][4][Ns], exactly the reference's argument */
int iono_forward_tec_rays(iono_ctx *ctx, const double *rays, int64_t R, int Ns, int interp_kind,
                          int quad_rule, double *tec_out);
/* dtec = tec - tec[i0] over [Na][Nt*Nd] (inversion/forward_equation.py:50) */
int iono_subtract_reference(iono_ctx *ctx, double *tec_inout, int Na, int64_t NtNd, int i0);
/* phase observable g[Na][Nt][Nd][Nf] (inversion/iterative_newton.py:86-127); grid holds ne = K exp(mu) */
int iono_forward_phase_rays(iono_ctx *ctx, const double *rays, int Na, int Nt, int Nd, int Ns,
                            const double *freqs, int Nf, const double *clock /*[Na][Nt]*/,
                            const double *const_ /*[Na]*/, int i0, int quad_rule, double *g_out);

/* ---- adjoint (exact transpose of the forward; SURVEY.md section 8a row A7') ------------------- */
/* grad[v] (+)= sum_r w[r] sum_k c_{r,k} W_{k,v}; if scale_by_grid, grad[v] *= M[v] afterwards
 * (gradient w.r.t. the log-model, cf. inversion/gradient.py:19).  grad_out: float64[nx*ny*nz]. */
/* interp_kind selects WHICH forward is transposed: IONO_INTERP_TRILINEAR (8 weights per sample) or
 * IONO_INTERP_TRICUBIC (the 216-tap tensor-product form; notebooks/TricubicInterpolation.ipynb c0:165-299). */
int iono_adjoint_straight(iono_ctx *ctx, const double *origins, const double *directions,
                          const double *w, int64_t R, double tmax, int Ns, int interp_kind, int quad_rule,
                          int scale_by_grid, double *grad_out);
int iono_adjoint_rays(iono_ctx *ctx, const double *rays, const double *w, int64_t R, int Ns,
                      int interp_kind, int quad_rule, int scale_by_grid, double *grad_out);

/* The reference's SHIPPED gradient discretisation, for comparison with it: do_gradient of
 * inversion/gradient.py:15-20 = einsum(dirac, M, dd) with dirac = voxel chord lengths of the straight first -> last
 * sample line (geometry/ray_dirac.py:5-34, geometry/slab_method.py:19-58).  rays[R][4][Ns], dd[R]; the grid holds M
 * (uniform axes, as the reference assumes: it uses x[1] - x[0]).  grad_out float64[nx ny nz].  NOT the transpose of the
 * forward model -- optimisers should use iono_adjoint_*. */
int iono_gradient_chords(iono_ctx *ctx, const double *rays, const double *dd, int64_t R, int Ns, double *grad_out);
int iono_gradient_chords_dev(iono_ctx *ctx, const double *rays_dev, const double *dd_dev, int64_t R, int Ns,
                             double *grad_dev /* accumulated into */);

/* ---- device-pointer (asynchronous) variants used by the inversion loop, bench.py and the
 *      multi-GPU driver.  Out-of-grid samples set a sticky device flag read by iono_check_oob. ---- */
/* order_dev (nullable): int32 permutation of 0..R-1 giving the order in which rays are WALKED
 * (results still land in tec_dev[ray]).  Given an order, the ideal-uniform forward kernels (trilinear,
 * float32 blocks, tricubic) interleave all the waves of an XCD in it, so neighbours in the order run
 * at the same time on neighbouring waves: make them NEARLY IDENTICAL rays (one antenna, one line of
 * sight a few seconds apart -- RayEngine.coherent_order) and the lines they read meet in the L1.
 * Speed only; the kernels for non-uniform grids ignore it. */
int iono_forward_tec_straight_dev(iono_ctx *ctx, const double *origins_dev, const double *directions_dev,
                                  const int *order_dev, int64_t R, double tmax, int Ns, int interp_kind,
                                  int quad_rule, double *tec_dev);
int iono_forward_tec_rays_dev(iono_ctx *ctx, const double *rays_dev, int64_t R, int Ns,
                              int interp_kind, int quad_rule, double *tec_dev);
/* accumulates INTO grad_dev (caller zeroes it); accum_dtype IONO_F64 | IONO_F32 selects the
 * element type of grad_dev and of the atomics.  order_dev (nullable) as for the forward: on uniform
 * grids the adjoint pre-reduces bundles of 64 consecutive rays of the walk in LDS, so an order that
 * puts spatially neighbouring rays next to each other cuts global atomics by an order of magnitude. */
int iono_adjoint_straight_dev(iono_ctx *ctx, const double *origins_dev, const double *directions_dev,
                              const int *order_dev, const double *w_dev, int64_t R, double tmax, int Ns,
                              int interp_kind, int quad_rule, void *grad_dev, int accum_dtype);
int iono_adjoint_rays_dev(iono_ctx *ctx, const double *rays_dev, const double *w_dev, int64_t R,
                          int Ns, int interp_kind, int quad_rule, void *grad_dev, int accum_dtype);
/* fused residual -> differential weights -> back-projection for layout [Na][Nt*Nd]:
 *   dd = (tec[a,p] - tec[i0,p] - dobs[a,p]) / (CdCt[a,p] + 1e-15)      (inversion/gradient.py:77-81)
 *   w[a,p] = dd[a,p] - [a == i0] sum_a' dd[a',p]                        (transpose of the i0 differencing)
 * then the adjoint of the straight-ray forward, in ONE launch. */
int iono_adjoint_residual_straight_dev(iono_ctx *ctx, const double *origins_dev, const double *directions_dev,
                                       const int *order_dev, const double *tec_dev, const double *dobs_dev, const double *cdct_dev,
                                       int Na, int64_t NtNd, int i0, double tmax, int Ns, int interp_kind, int quad_rule,
                                       void *grad_dev, int accum_dtype);
/* the same fused launch for the LINEAR solvers: back-projection of the differential weights of v * scale,
 *   w[a,p] = y[a,p] - [a == i0] sum_a' y[a',p],  y = v * scale  (scale_dev nullable = 1)
 * i.e. A^T (scale o v) for the differenced operator A x = G x - (G x)[i0] -- CGLS needs A^T (W^1/2 r), SIRT
 * A^T (L r) (row-sum normalisation, geometry/oct_trees/Inversion.py:559) -- without forming y or w in memory. */
int iono_adjoint_differential_straight_dev(iono_ctx *ctx, const double *origins_dev, const double *directions_dev,
                                           const int *order_dev, const double *v_dev, const double *scale_dev,
                                           int Na, int64_t NtNd, int i0, double tmax, int Ns, int interp_kind,
                                           int quad_rule, void *grad_dev, int accum_dtype);
/* A solver iteration's ray-sized pass FUSED with the differential-weights pass of the back-projection that follows it: one small
 * launch + the back-projection instead of two or three small launches + the back-projection (round 4; the loop these replace:
 * inversion/iterative_newton.py:993-1000 -- forward, residual, gradient).  All vectors in ray layout [Na][NtNd]; `partial`
 * (nullable): IONO_NPART per-workgroup partial sums, to be summed in order by the consumer (iono_vec_axpby_dot_dev et al.).
 *   cg_step  : r -= (sum an / sum ad) q in place, partial = sum r^2; grad += A^T (scale o r)          (A = differenced operator)
 *   sirt_step: v = dobs - (tec - tec[i0]); partial = sum v^2 weight; grad += A^T (scale o v); r_out (nullable) = v
 * an / ad: device scalars given as (pointer, count) like iono_vec_axpby_dot_dev's. */
int iono_adjoint_cg_step_dev(iono_ctx *ctx, const double *origins_dev, const double *directions_dev, const int *order_dev,
                             double *r_dev, const double *q_dev, const double *an_dev, int an_count, const double *ad_dev, int ad_count,
                             const double *scale_dev, int Na, int64_t NtNd, int i0, double tmax, int Ns, int interp_kind,
                             int quad_rule, double *partial_dev, void *grad_dev, int accum_dtype);
int iono_adjoint_sirt_step_dev(iono_ctx *ctx, const double *origins_dev, const double *directions_dev, const int *order_dev,
                               const double *tec_dev, const double *dobs_dev, const double *scale_dev, const double *weight_dev, int Na,
                               int64_t NtNd, int i0, double tmax, int Ns, int interp_kind, int quad_rule, double *r_out_dev,
                               double *partial_dev, void *grad_dev, int accum_dtype);
/* ---- z-slabs of the back-projection plan: what lets a multi-GPU solver exchange a finished part of the update while the rest is still
 *      being back-projected (replaces the reference's sum over per-direction gradients, inversion/gradient.py:52-54).
 * iono_adjoint_plan_slabs(n): the NEXT iono_adjoint_plan_dev orders its work units by z-slab (n <= 8 slabs of whole box layers).
 * iono_adjoint_plan_slab_info: nslab, unit_lo[nslab + 1] (slab s = units [unit_lo[s], unit_lo[s+1])) and z_lo[nslab + 1] (slab s owns
 *   the node levels [z_lo[s], z_lo[s+1]): once the units of slabs 0 .. s have run, those levels are final -- segments never leave
 *   their z-layer of boxes, also where samples overhang the box image in x or y and go by global atomics).
 * iono_adjoint_unit_range(lo, hi): the next planned trilinear back-projection runs units [lo, hi) only (one-shot).  Any other
 * back-projection launched while a range is pending (no plan / a replaced plan / tricubic) returns IONO_ERR_ARG and launches nothing.
 * iono_adjoint_cg_step_dev / _sirt_step_dev with grad_dev = NULL: the ray pass alone; iono_adjoint_planned_weights_dev then
 *   back-projects the weights it left in the library (slab by slab with iono_adjoint_unit_range). */
int iono_adjoint_plan_slabs(iono_ctx *ctx, int nslab);
/* Deterministic back-projection (also env IONOTOMO_DETERMINISTIC=1): the planned trilinear and tricubic transposes accumulate 64-bit
 * fixed-point integers (box images in LDS and the grid / the derivative channels), so the result does not depend on the order the atomics are served in: two launches on the
 * same inputs return the same bits, and so does every solver iterate built on them.  Resolution: 2^-(62 - b) of the launch's largest
 * contribution, b = log2 bound of the terms one node's sum can receive (>= 12; counted at the first such launch of a plan by running
 * the kernel with every contribution = 1: 2^-48 at the bench shape, 9.4e-13 of the largest value from the float sum).  Costs one small
 * reduction and one grid-sized conversion pass per launch (trilinear: 0.32 against 0.29 ms; the tricubic transpose, whose z fold reads the
 * integers directly, is FASTER in this mode -- 2.21 against 2.52 ms -- because the integer LDS atomic is the cheaper instruction).  A
 * back-projection the fixed-point kernels do not serve (no plan for these rays; the explicit-sample, phase and curved-ray transposes)
 * returns IONO_ERR_ARG while the mode is on.  No counterpart in the reference (numpy sums in a fixed order). */
int iono_set_deterministic(iono_ctx *ctx, int on);
int iono_adjoint_plan_slab_info(iono_ctx *ctx, int *nslab_out, int *unit_lo_out, int *z_lo_out);
int iono_adjoint_unit_range(iono_ctx *ctx, int unit_lo, int unit_hi);
int iono_adjoint_planned_weights_dev(iono_ctx *ctx, const double *origins_dev, const double *directions_dev, const int *order_dev,
                                     int64_t R, double tmax, int Ns, int interp_kind, int quad_rule, void *grad_dev, int accum_dtype);
int iono_subtract_reference_dev(iono_ctx *ctx, double *tec_dev, int Na, int64_t NtNd, int i0);
/* Measured load balance of the two chunked kernels.  The straight-ray forward gives every resident wave one
 * contiguous chunk of the ray walk, the LDS-tiled adjoint every resident workgroup; the cost per ray varies with where
 * the ray runs (cache locality for the forward; for the adjoint several-fold with the local ray density: dense bundles
 * share one tile flush, sparse fans do not), so chunks of equal ray COUNT leave much of the chip idle while the slowest
 * chunk finishes.  The ray geometry is fixed for a whole inversion (it is what the reference's per-direction dask tasks
 * re-derive on every call, inversion/forward_equation.py:60-67, inversion/gradient.py:22-54), so the balance is tuned
 * once.  `which`: IONO_WALK_FORWARD or IONO_WALK_ADJOINT.
 *   iono_walk_cycles         -> cycles each chunk of the LAST such launch took, in walk order; n_chunks entries;
 *                               n_units = resident waves (forward) / workgroups (adjoint) of that launch
 *                               (cycles_out may be NULL to query the two counts);
 *   iono_walk_partition_set  -> n_chunks + 1 non-decreasing chunk boundaries (host), starts[0] = 0,
 *                               starts[n_chunks] = R.  Forward: n_chunks must equal n_units (used only for launches
 *                               without an `order`).  Adjoint: n_chunks >= n_units; chunk b < n_units is taken by
 *                               workgroup b, the rest are handed out through an atomic counter as workgroups finish, so
 *                               make them progressively smaller (guided self-scheduling).  Used by later launches with
 *                               the same R; any other launch falls back to equal counts.  NULL clears it.  Never
 *                               affects results, only which wave / workgroup handles which rays. */
enum { IONO_WALK_FORWARD = 0, IONO_WALK_ADJOINT = 1 };
/* Walk order for `order_dev` above (host arrays in, int32[R] host permutation out): 4-D Morton order of the rays'
 * foot points and far ends in grid cells, so that consecutive rays of the walk nearly coincide.  Geometry only --
 * compute once per ray set (the reference re-derives its per-direction task split on every call,
 * inversion/gradient.py:22-54); the host-pointer iono_adjoint_straight applies it internally. */
int iono_walk_order(iono_ctx *ctx, const double *origins, const double *directions, int64_t R, double tmax, int *order_out);
int iono_walk_cycles(iono_ctx *ctx, int which, uint64_t *cycles_out, int capacity, int *n_chunks, int *n_units);
int iono_walk_partition_set(iono_ctx *ctx, int which, const int64_t *starts, int n_chunks, int64_t R);
/* Node-stationary back-projection (ionotomo_amd/csrc/iono_binned_kernels.h).  The ray geometry is fixed for a whole
 * inversion (the reference re-derives its per-direction task split on every call, inversion/gradient.py:22-54), so the
 * transpose can be organised around the GRID once: rays are cut into segments of <= 16 samples, the segments are binned by
 * grid box, and every later iono_adjoint_*_straight_dev call with the SAME origins_dev / directions_dev pointers, R, tmax,
 * Ns and interp_kind reduces each box in LDS and flushes it once (5 x fewer memory-side atomics than the ray-stationary
 * kernel on the bench geometry).  The caller must not modify the two arrays while the plan is in use; results never depend
 * on the plan (samples outside their box image go straight to global atomics).  Uniform (np.linspace) grids only: on any
 * other grid no plan is built and the calls keep using the ray-stationary kernels.  A new grid clears the plan. */
int iono_adjoint_plan_dev(iono_ctx *ctx, const double *origins_dev, const double *directions_dev, int64_t R, double tmax,
                          int Ns, int interp_kind);
int iono_adjoint_plan_clear(iono_ctx *ctx);
int iono_adjoint_plan_info(iono_ctx *ctx, int64_t *n_segments, int *n_units, double *outside_fraction);
/* Lanes per segment the plan chose for this geometry (4, 8 or 16; 0 without a plan): the width that leaves the fewest lanes of
 * the LDS atomics empty.  Geometry only; never changes which samples are summed (env IONOTOMO_SEG_LANES forces a width). */
int iono_adjoint_plan_segment_lanes(iono_ctx *ctx, int *lanes);
/* Device memory for hosts that keep operands resident between calls but have no torch (the reference-signature facade:
 * a line search calls forward_equation(rays, K_ne, m_tci, i0) again and again with the SAME rays,
 * inversion/line_search.py:56,71,83).  iono_dev_download is also the end of a chain of *_dev launches: it waits for the
 * ctx stream and returns IONO_ERR_OOB if a sample left the grid.  iono_scale_by_grid_dev: grad *= grid values
 * (d/dm = ne d/dne for the log-model, what scale_by_grid does in the host entry points). */
int iono_dev_alloc(iono_ctx *ctx, size_t bytes, void **out_dev);
int iono_dev_free(iono_ctx *ctx, void *p_dev);
int iono_dev_upload(iono_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int iono_dev_zero(iono_ctx *ctx, void *p_dev, size_t bytes);
int iono_dev_download(iono_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int iono_scale_by_grid_dev(iono_ctx *ctx, double *grad_dev);
/* Bundle-stationary forward (ionotomo_amd/csrc/iono_forward_kernels.h: k_forward_bundle).  The same observation as for the
 * back-projection plan: the rays of forward_equation (inversion/forward_equation.py:13-33) are fixed for a whole inversion,
 * so they can be organised once: sorted along a 4-D Morton curve of foot and end point, cut into bundles of <= 64 nearly
 * coincident rays, and every later iono_forward_tec_straight_dev call with the SAME origins_dev / directions_dev pointers,
 * R, tmax and Ns (trilinear, float64 storage, np.linspace axes) gives a workgroup one bundle: the voxel neighbourhood the
 * bundle needs for 8 consecutive samples is copied to LDS once (LDS-DMA) and every ray interpolates from there.  The
 * caller must not modify the two arrays while the plan is in use.  TEC never depends on the plan: a chunk whose
 * neighbourhood does not fit the LDS image takes direct loads with bit-identical arithmetic, and a ray's partial sums are
 * added in a fixed order.  No plan is built (and the other forward kernels serve the call) on other grids / storage.
 * interp_kind = IONO_INTERP_TRICUBIC launches use the same plan (k_forward_bundle_lm, iono_cubic_kernels.h: the Lekien-Marsden
 * derivative fields kept as four pair-major arrays, one pair per wave, windows of 4-sample chunks), so does the phase forward. */
int iono_forward_plan_dev(iono_ctx *ctx, const double *origins_dev, const double *directions_dev, int64_t R, double tmax, int Ns);
int iono_forward_plan_clear(iono_ctx *ctx);
int iono_forward_plan_info(iono_ctx *ctx, int64_t *n_bundles, int *n_chunks, double *fit_fraction);
/* Hybrid dispatch (round 6): which bundles are worth a workgroup is decided per BUNDLE when the plan is made, not per launch.  A
 * bundle costs its workgroup the same whether it holds 64 rays or 3, the lanes = samples kernel costs per ray, and a second launch
 * costs ~5 us: the plan picks the threshold T (serve bundles of >= T rays; 1 = all, 65 = none) with the smallest modelled time
 * (env IONOTOMO_HYBRID_MIN=1..65 forces T).  The rays of unserved bundles form the tail of the plan's walk and every planned launch
 * (TEC, tricubic TEC, phase) hands that tail to the lanes = samples kernel of the same interpolant in the same call.  Reports: bundles
 * the cut produced, bundles served, rays in served bundles, rays in the tail, T, (hist65, optional, 65 entries) the number of bundles
 * by ray count as cut, (model_us3, optional) the model's microseconds for: all bundles served | the chosen split | lanes = samples only.
 * iono_forward_plan_info's n_bundles counts the SERVED bundles (0: the plan only orders the walk).  A batch that mixes well-filled
 * bundles with thousands of singletons -- a few timesteps of a sparse array, cf. the coherence windows of
 * inversion/inversion_pipeline.py:41-50 -- keeps its bundles this way instead of losing the plan.  Results never depend on T. */
int iono_forward_plan_split(iono_ctx *ctx, int64_t *bundles_cut, int64_t *bundles_served, int64_t *rays_served, int64_t *rays_rest,
                            int *min_rays, int64_t *hist65, double *model_us3);
/* Solver vector update  y = a x + b y  on device vectors (16-byte aligned), one pass.  The coefficients are ratios
 * of DEVICE scalars, a = a_sign * a_num[0] / a_den[0], b = b_num[0] / b_den[0] (a null pointer stands for 1), so the
 * step lengths of the iteration -- eps = sum(Gdm dd/Cd) / sum(Gdm^2/Cd), inversion/iterative_newton.py:542-554; the
 * model update m <- m - eps (...), geometry/oct_trees/Inversion.py:533 -- never visit the host. */
int iono_vec_axpby_dev(iono_ctx *ctx, double *y_dev, const double *x_dev, int64_t n, const double *a_num_dev,
                       const double *a_den_dev, double a_sign, const double *b_num_dev, const double *b_den_dev);
/* ---- the phase observable on the device (inversion/iterative_newton.py:86-127) and its adjoint ------------------------
 * g[a,t,d,l] = const[a] + 2 pi nu_l clock[a,t] - (2 pi nu_l / c) (phi[a,t,d,l] - phi[i0,t,d,l]),
 * phi = int (1 - sqrt(1 - ne / n_p)) ds along straight rays whose samples are generated in-kernel (the reference's
 * rays[Na,Nt,Nd,4,Ns] never exists); grid holds ne = K exp(mu).  freqs is a HOST array (Nf), everything else device.
 * phi_work_dev: R x Nf doubles of scratch. */
int iono_forward_phase_straight_dev(iono_ctx *ctx, const double *origins_dev, const double *directions_dev, int Na, int Nt,
                                    int Nd, double tmax, int Ns, const double *freqs_host, int Nf, const double *clock_dev,
                                    const double *const_dev, int i0, int quad_rule, double *phi_work_dev, double *g_dev);
/* grad (+)= d/d ne [ sum y[r,l] g[r,l] ]  for y = dS/dg (e.g. (g - dobs)/CdCt, iterative_newton.py:32-38): the trilinear
 * transpose with the per-sample factor sum_l w_{r,l} / (2 n_p,l sqrt(1 - ne_k / n_p,l)), forward gather and scatter in ONE
 * traversal; w = -(2 pi nu / c) x differential weights of y (reference-antenna differencing).  wrt_log_model: multiply by
 * ne at the nodes afterwards (d/d mu; grad_dev must then have been zero).  grad_dev float64[nx ny nz];
 * wrf_work_dev: R x Nf doubles of scratch.  After iono_adjoint_plan_dev on the same origins / directions / tmax / Ns the
 * node-stationary (box-binned) kernel runs instead of the ray-stationary one (0.6 against 2.1 ms at the bench shape). */
int iono_adjoint_phase_straight_dev(iono_ctx *ctx, const double *origins_dev, const double *directions_dev,
                                    const int *order_dev, const double *y_dev, int Na, int64_t NtNd, double tmax, int Ns,
                                    const double *freqs_host, int Nf, int i0, int quad_rule, double *wrf_work_dev,
                                    int wrt_log_model, double *grad_dev);

/* ---- fused vector passes of the inversion loop (ionotomo_amd/csrc/iono_solver_kernels.h).  A CGLS / SIRT iteration
 *      (objective / step lengths: inversion/iterative_newton.py:32-38,542-554; normalisations:
 *      geometry/oct_trees/Inversion.py:533,559,564) is the two ray kernels + these.  Device scalars are passed as
 *      (pointer, count): NULL = 1.0; count 1 = a plain scalar (e.g. after an all-reduce); count IONO_NPART = the
 *      per-workgroup partial sums a dot-producing pass wrote (summed in a fixed order by the consumer: no atomics,
 *      identical bits in every workgroup and on every rank). ------------------------------------------------------- */
enum { IONO_NPART = 512 };
/* Grid values owned by the caller: a float64 device buffer of iono_grid_padded_size() elements (the tail beyond
 * nx*ny*nz must stay zero: unclamped far-corner reads) that the kernels read IN PLACE -- the solver updates it through
 * the compact index instead of copying a model vector into the library every iteration.  NULL unbinds.  Tell the
 * library with iono_grid_values_changed() when the buffer was modified (cached refractive-index nodes / tricubic
 * derivative fields are then rebuilt on next use). */
int iono_grid_padded_size(iono_ctx *ctx, int64_t *count_out);
int iono_grid_bind_values_dev(iono_ctx *ctx, double *values_dev);
int iono_grid_values_changed(iono_ctx *ctx);
/* out[a,p] = s1[a,p] * (a_coef * (tec[a,p] - tec[i0,p]) + b_coef * dobs[a,p]);  partial[blk] = sum out^2 * s2
 * (dobs, s1, s2, partial nullable): the differenced forward (inversion/forward_equation.py:50), the residual and its
 * weighted norm in ONE pass over the rays. */
int iono_rays_combine_dev(iono_ctx *ctx, const double *tec_dev, const double *dobs_dev, const double *s1_dev,
                          const double *s2_dev, int Na, int64_t NtNd, int i0, double a_coef, double b_coef,
                          double *out_dev, double *partial_dev);
/* One coherence window (Na * NtNd <= 32768 rays): the ray-sized passes of a CG (mode 0) or SIRT (mode 1) iteration in one launch of one
 * workgroup -- iono_rays_combine_dev, iono_vec_axpby_dot_dev and the reference-antenna sums of the differential back-projection,
 * element for element (inversion/iterative_newton.py:542-554; geometry/oct_trees/Inversion.py:533,559,564):
 *   mode 0: q = scale (tec - tec[i0]), dot1 = <q, q>;  r -= (gamma / dot1) q, dot2 = <r, r>;  w = differential weights of (r scale)
 *   mode 1: r = dobs - (tec - tec[i0]), dot1 = sum r^2 weight;                              w = differential weights of (r scale)
 * gamma: device scalar as (pointer, count) like iono_vec_axpby_dot_dev.  w feeds iono_adjoint_straight_dev directly. */
int iono_small_ray_pass_dev(iono_ctx *ctx, int mode, const double *tec_dev, const double *dobs_dev, const double *scale_dev,
                            const double *weight_dev, double *r_dev, double *q_dev, int Na, int64_t NtNd, int i0, const double *gamma,
                            int gamma_count, double *dot1_dev, double *dot2_dev, double *w_dev);
/* y = (a_sign an / ad) x + (bn / bd) y;  partial[blk] = sum y^2 */
int iono_vec_axpby_dot_dev(iono_ctx *ctx, double *y_dev, const double *x_dev, int64_t n, const double *an, int an_count,
                           const double *ad, int ad_count, double a_sign, const double *bn, int bn_count,
                           const double *bd, int bd_count, double *partial_dev);
/* active-set (compact) grid vectors: idx_dev = sorted int32 node indices the rays reach, n of them */
int iono_compact_gather_dev(iono_ctx *ctx, double *full_dev, const int *idx_dev, int64_t n, double *out_dev,
                            int zero_source, double *partial_dev);          /* out = full[idx] (re-zeroed); sum out^2 */
int iono_compact_scatter_dev(iono_ctx *ctx, double *full_dev, const int *idx_dev, int64_t n, const double *src_dev);
/* x += (an/ad) p;  p = s + (bn/bd) p;  full_p[idx] = p   (the tail of a CGLS iteration in one pass) */
int iono_compact_cg_update_dev(iono_ctx *ctx, double *x_dev, double *p_dev, const double *s_dev, const int *idx_dev,
                               int64_t n, double *full_p_dev, const double *an, int an_count, const double *ad,
                               int ad_count, const double *bn, int bn_count, const double *bd, int bd_count);
/* s = full_s[idx] (re-zeroed);  x += relax C s (>= 0 if nonneg);  full_x[idx] = x;  partial_max[blk] = max |dx| */
int iono_compact_sirt_update_dev(iono_ctx *ctx, double *x_dev, const double *C_dev, double *full_s_dev,
                                 const int *idx_dev, int64_t n, double *full_x_dev, double relax, int nonneg,
                                 double *partial_max_dev);
/* Fermat tracer with device buffers (rays_dev[R][4][Ns]); the refractive-index nodes are cached in the
 * ctx and rebuilt when the grid values or the frequency change.  Feed rays_dev to
 * iono_forward_tec_rays_dev for the curved-ray TEC (BASELINE config 3) without leaving the GPU. */
int iono_trace_fermat_dev(iono_ctx *ctx, const double *origins_dev, const double *directions_dev, int64_t R,
                          double tmax, int Ns, double frequency, int bend, int interp_kind, int substeps,
                          int independent, double *rays_dev);
/* Curved-ray TEC and its transpose WITHOUT the rays: the reference's curved path is cast_ray -> Fermat.integrate_ray ->
 * forward_equation on the returned samples (geometry/calc_rays.py:61-96, inversion/fermat.py:58-72,150-174,
 * inversion/forward_equation.py:27-28); rays[R][4][Ns] is 5.1 GB at config 4's ray count.  These two entry points run the same
 * RK4 stepper as iono_trace_fermat_dev (identical samples) and accumulate the non-uniform Simpson rule on the fly
 * (k_fermat_tec, iono_fermat_kernels.h):
 *   tec[r]  = ne_scale * sum_k c_k(s) ne(x_k)                     (== iono_trace_fermat_dev + iono_forward_tec_rays_dev)
 *   grad   += ne_scale * sum_r w[r] sum_k c_k(s) W(x_k)           (== ... + iono_adjoint_rays_dev; the path is held fixed)
 * The grid must hold ne [m^-3] (the refractive index is derived from it at `frequency`); interp_kind_n interpolates n
 * for the ray equation, interp_kind_ne the integrand.  float64 storage only. */
int iono_forward_tec_fermat_dev(iono_ctx *ctx, const double *origins_dev, const double *directions_dev, int64_t R, double tmax,
                                int Ns, double frequency, int bend, int interp_kind_n, int substeps, int independent,
                                int interp_kind_ne, int quad_rule, double ne_scale, double *tec_dev);
int iono_adjoint_fermat_dev(iono_ctx *ctx, const double *origins_dev, const double *directions_dev, const double *w_dev, int64_t R,
                            double tmax, int Ns, double frequency, int bend, int interp_kind_n, int substeps, int independent,
                            int interp_kind_ne, int quad_rule, double ne_scale, double *grad_dev);
/* Would iono_forward_tec_fermat_dev (transpose = 0) / iono_adjoint_fermat_dev (transpose = 1) serve (interp_kind_n, interp_kind_ne, R,
 * bend) with the fused tricubic-index kernel (k_fermat_tec_lm: a few lanes per ray on node records, 1.2 ms at config 3) -- 1 -- or with
 * the lanes = rays kernel (31 ms there) -- 0?  The library's own dispatch predicate, so that a host layer choosing between the fused
 * call and trace + integrate never re-implements it.  (The transpose runs on it for bending rays in batches large enough for two
 * lanes per ray: where the alternative is a ray tensor of 32 R Ns bytes.) */
int iono_fermat_lm_ok(iono_ctx *ctx, int interp_kind_n, int interp_kind_ne, int64_t R, int transpose, int bend, int *ok_out);
int iono_check_oob(iono_ctx *ctx, int *oob_out);          /* synchronises; reads and clears the flag */
/* Plans (iono_forward_plan_dev, iono_adjoint_plan_dev) are keyed on device pointers, but every planned launch re-hashes the rays
 * it is handed (64 bits per ray) against the hashes recorded when the plan was built.  If a planned array was edited in place:
 * the forward falls back to direct loads for the bundles concerned (its TEC is exact for ANY bundling, so the result is still
 * correct), the back-projection -- which works from the plan's own ray records -- poisons the edited rays (every node they touch
 * comes out NaN), and both raise a sticky flag.  iono_plan_stale synchronises, reads and clears it; the Python layer raises
 * ValueError (IONO_ERR_ARG semantics) and asks for a new plan.  Replaces no reference code (the reference has no plans). */
int iono_plan_stale(iono_ctx *ctx, int *stale_out);

/* ---- model-covariance smoothing C_m: Covariance.smooth =
 *      scipy.ndimage.convolve(phi, c_stencil, mode='nearest') (ionosphere/covariance.py:46-63,383-385) with the
 *      reference's separable stencil c[a][b][c] = kx[a] ky[b] kz[c], each of length 2h+1 (host arrays).  Arrays
 *      have the grid's shape; in/out/work (device variant) must be distinct buffers. ---------------------------- */
int iono_smooth_separable(iono_ctx *ctx, const double *in, double *out, const double *kx, const double *ky,
                          const double *kz, int h);
int iono_smooth_separable_dev(iono_ctx *ctx, const double *in_dev, double *out_dev, double *work_dev,
                              const double *kx, const double *ky, const double *kz, int h);

/* ---- multi-GPU: the sum over ranks of the back-projected update, one process per GPU (SURVEY.md 8b `allreduce_grid`, 8e).
 *      Replaces `da.sum(da.stack(gradient, axis=-1), axis=-1)` over the per-direction dask tasks
 *      (inversion/gradient.py:52-54).  RCCL over xGMI, loaded lazily (dlopen "librccl.so": hosts that do the collective
 *      themselves -- the Python layer uses torch.distributed -- never load it).  Rank 0 obtains the 128-byte id and ships
 *      it to the other ranks over the host's own channel (dask scheduler, MPI, a file); every rank then calls
 *      iono_comm_init (collective).  iono_comm_allreduce_dev sums `count` elements IN PLACE over all ranks, enqueued on
 *      the ctx stream (ordered after the adjoint launch that produced them, no host synchronisation); dtype IONO_F64 or
 *      IONO_F32.  IONO_ERR_HIP (with the RCCL message in iono_last_error) on any RCCL failure. ------------------------ */
enum { IONO_COMM_ID_BYTES = 128 };
int iono_comm_unique_id(iono_ctx *ctx, char id_out[IONO_COMM_ID_BYTES]);
int iono_comm_init(iono_ctx *ctx, const char id[IONO_COMM_ID_BYTES], int rank, int nranks);
int iono_comm_allreduce_dev(iono_ctx *ctx, void *buf_dev, int64_t count, int dtype);
int iono_comm_destroy(iono_ctx *ctx);

/* ---- ONE dispatch table: which kernel a launch gets --------------------------------------------------------------------------
 * Every launcher in the library asks the same pure function of a handful of FACTS -- the grid's tier, its storage, the interpolant,
 * the batch size, whether a plan serves the launch -- and launches what it answers; iono_dispatch_name answers without a GPU (the
 * table is pinned by tests/test_cabi.py), iono_dispatch_describe fills the facts from a context and the launch's own arguments
 * (what bench.py prints as config.forward_kernel).  The reference has no counterpart: its path is one numpy code
 * (inversion/forward_equation.py:13-51, inversion/gradient.py:66-102, inversion/fermat.py:150-174) -- this is the map from those three
 * entry points to the kernels that serve them here. */
enum { IONO_OP_FORWARD = 0,          /* iono_forward_tec_straight*            <- inversion/forward_equation.py:13-51 */
       IONO_OP_ADJOINT = 1,          /* iono_adjoint_*straight*               <- inversion/gradient.py:66-102 (exact transpose) */
       IONO_OP_TRACE = 2,            /* iono_trace_fermat*                    <- inversion/fermat.py:150-174 */
       IONO_OP_FERMAT_FORWARD = 3,   /* iono_forward_tec_fermat_dev           (trace + integrate fused) */
       IONO_OP_FERMAT_ADJOINT = 4,   /* iono_adjoint_fermat_dev */
       IONO_OP_PHASE_FORWARD = 5,    /* iono_forward_phase_straight_dev       <- inversion/iterative_newton.py:86-127 */
       IONO_OP_PHASE_ADJOINT = 6 };  /* iono_adjoint_phase_straight_dev */
typedef struct iono_dispatch_facts {
    int storage;              /* IONO_F64 | IONO_F32 */
    int tier;                 /* axes: 0 general (binary search), 1 table-uniform, 2 ideal-uniform (np.linspace) with Ns <= 4096 */
    int cubic_fast;           /* the Lekien-Marsden record tier serves the grid (tier 2, >= 6 nodes per axis, 32-bit-safe records, IONOTOMO_VARIANT != 4) */
    int cubic_records;        /* ... whatever Ns (what the tracers ask) */
    int ideal_axes;           /* every axis is g0 + i h to 2.5e-13 h and the general tiers are not forced (the tracers' fast right-hand side) */
    int q4_ok;                /* float32 2 x 2 corner blocks addressable with 32-bit offsets */
    int variant;              /* IONOTOMO_VARIANT: 0 | 2 (general back-projection) | 3 (lanes = rays tracers) | 4 (216-tap tricubic) */
    int deterministic;        /* iono_set_deterministic */
    int interp_kind;          /* of the integrand / the transposed forward | of the refractive index (tracer ops) */
    int ne_kind;              /* fused Fermat ops: the integrand's interpolant */
    int bend;                 /* tracer ops */
    int Ns;
    int64_t R;
    int64_t fwd_bundles;      /* served bundles of a forward plan that serves THIS launch (0: none) */
    int64_t fwd_tail;         /* ... and the rays it leaves to the lanes = samples kernel */
    int adj_planned;          /* a back-projection plan of this interpolant serves this launch */
    int adj_tiles;            /* ... and lists its fold tiles (tricubic plans) */
    int adj_seg_lanes;        /* ... its lanes per segment (4 | 8 | 16) */
    int axes_bytes;           /* 8 (nx + ny + nz) */
    int fermat_lm_lanes;      /* forced lanes per ray of the record tracer (8 | 2; 0: by batch size) */
    int64_t fermat_lm_few_min, fermat_poly_max, fermat_lin4_max, fermat_coop_max;      /* batch-size thresholds of the tracers */
} iono_dispatch_facts;
/* The kernel(s) `op` gets under `facts`, as text ("k_forward_bundle<0> + k_forward_straight_u<double>"), into out[cap].  Pure. */
int iono_dispatch_name(const iono_dispatch_facts *facts, int op, char *out, int cap);
/* The facts of a launch on `ctx` with these arguments (device pointers may be null: then no plan matches), and its kernel name. */
int iono_dispatch_describe(iono_ctx *ctx, int op, const double *origins_dev, const double *directions_dev, int64_t R, double tmax,
                           int Ns, int interp_kind, int ne_kind, int bend, iono_dispatch_facts *facts_out, char *out, int cap);

#ifdef __cplusplus
}
#endif
#endif
