"""Device-resident driver of the ray-integral kernels: everything stays in HBM.

The reference API (``forward_equation(rays, K_ne, m_tci, i0)`` ...) hands over host numpy arrays,
so every facade call pays PCIe for the grid and the rays.  The inversion loop, bench.py and the
multi-GPU driver instead keep origins / directions / model / data as torch CUDA tensors and call
the ``*_dev`` C-ABI entry points with raw device pointers on torch's current stream.  torch is
used for device memory, streams and torch.distributed only -- all numerics are the HIP kernels.
"""
import numpy as np
import torch

from . import _lib

TECU = 1e13


def _ptr(t):
    return _lib._V(t.data_ptr())


def ctypes_int64():
    import ctypes
    return ctypes.c_int64(0)


def _byref(v):
    import ctypes
    return ctypes.byref(v)


class RayEngine(object):
    """One GPU, one grid, straight z-parametrised rays generated in-kernel (rays[R,4,Ns] is never
    materialised: at config 4 it would be 5.1 GB)."""

    def __init__(self, device=0, storage="f64", interp="linear", quad="avg"):
        if not torch.cuda.is_available():
            raise RuntimeError("RayEngine needs a GPU (no CPU fallback)")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.ctx = _lib.Context(device)
        self.storage = storage
        self.kind = _lib.interp_kind(interp)
        self.trilinear = self.kind == _lib.interp_kind("linear")
        self.rule = _lib.quad_rule(quad)
        self.shape = None
        self.deterministic = False
        self._bound_stream = -1
        self._values_serial = 0        # bumped whenever the node values may have changed: invalidates the cached Fermat step choice
        self._fermat_steps = {}

    def _sync_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        if s != self._bound_stream:
            self.ctx.set_stream(s)
            self._bound_stream = s

    def tensor(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

    # -- grid ------------------------------------------------------------------------------------
    def set_grid(self, xvec, yvec, zvec, M=None):
        self.ctx.set_grid(xvec, yvec, zvec, M, storage=self.storage)
        self._values_serial += 1
        self.shape = self.ctx.grid_shape
        self.ncells = int(np.prod(self.shape))

    def set_values(self, M_t):
        """grid values <- float64 device tensor"""
        self._sync_stream()
        assert M_t.is_cuda and M_t.dtype == torch.float64 and M_t.numel() == self.ncells and M_t.is_contiguous()
        self.ctx.call("iono_grid_set_values_dev", _ptr(M_t))
        self._values_serial += 1

    def set_log_model(self, m_t, scale):
        """grid values <- scale * exp(m) at the nodes (inversion/forward_equation.py:41-43)"""
        self._sync_stream()
        assert m_t.is_cuda and m_t.dtype == torch.float64 and m_t.numel() == self.ncells and m_t.is_contiguous()
        self.ctx.call("iono_grid_set_exp_dev", _ptr(m_t), float(scale))
        self._values_serial += 1

    # -- hot path ----------------------------------------------------------------------------------
    def forward(self, origins_t, dirs_t, tmax, Ns, out=None, order=None):
        """tec[R] for straight rays; origins/dirs are [R,3] float64 device tensors.  ``order``: optional
        int32 device permutation -- the order in which rays are walked (see ``locality_order``)."""
        self._sync_stream()
        R = origins_t.shape[0]
        if out is None:
            out = torch.empty(R, dtype=torch.float64, device=self.device)
        op = _lib._V(0) if order is None else _ptr(order)
        self.ctx.call("iono_forward_tec_straight_dev", _ptr(origins_t), _ptr(dirs_t), op, R, float(tmax), int(Ns),
                      self.kind, self.rule, _ptr(out))
        return out

    @staticmethod
    def coherent_order(origins_t, dirs_t, bits=16):
        """Permutation for the FORWARD kernels: rays grouped by origin (antenna), and inside a group sorted along a 2-D
        Morton curve of their direction, so that neighbours in the walk are NEARLY IDENTICAL rays (the same line of sight a
        few seconds apart).  Given an order, the forward kernel interleaves all the waves of an XCD in it, so those rays
        run on neighbouring waves at the same time and share the lines they read in the L1 (0.212 instead of 0.224 ms at the
        bench shape; a 4-D Morton order of foot and end points -- ``locality_order``, made for the back-projection --
        measured slower than no order here).  Host-side plumbing, results do not depend on it."""
        o, d = origins_t, dirs_t
        _, ant = torch.unique(o, dim=0, return_inverse=True)
        s = d[:, :2] / d[:, 2:3]
        lo = s.min(dim=0).values
        span = (s.max(dim=0).values - lo).clamp_min(1e-300)
        q = ((s - lo) / span * ((1 << bits) - 1)).to(torch.int64).clamp_(0, (1 << bits) - 1)
        code = ant.to(torch.int64) << (2 * bits)
        for b in range(bits):
            code |= ((q[:, 0] >> b) & 1) << (2 * b)
            code |= ((q[:, 1] >> b) & 1) << (2 * b + 1)
        return torch.argsort(code, stable=True).to(torch.int32).contiguous()

    def forward_launcher(self, origins_t, dirs_t, tmax, Ns, out, order=None):
        """A zero-argument callable that enqueues ``forward`` with every argument converted once (an inversion calls the same
        launch thousands of times: the per-call Python work -- stream look-up, pointer conversions -- is ~5 us of a
        250 us kernel).  The tensors must stay alive and the current stream unchanged."""
        self._sync_stream()
        fn = getattr(self.ctx._lib, "iono_forward_tec_straight_dev")
        args = (self.ctx._h, _ptr(origins_t), _ptr(dirs_t), _lib._V(0) if order is None else _ptr(order), origins_t.shape[0],
                float(tmax), int(Ns), self.kind, self.rule, _ptr(out))
        check = self.ctx._check

        def launch():
            check(fn(*args))
        launch.keep = (origins_t, dirs_t, out, order)
        return launch

    @staticmethod
    def locality_order(origins_t, dirs_t, tmax, cell=0.5, bits=15):
        """Permutation that walks rays whose paths nearly coincide one after another: 4-D Morton
        (Z-order) code of the quantised origin (x, y) and far-end (x, y) of each ray, so that any 64
        consecutive rays of the walk form a compact bundle all the way up.  The adjoint pre-reduces
        such bundles in LDS, which cuts its global atomics by an order of magnitude.  Host-side
        plumbing (torch integer ops + sort), not part of the numerics: any permutation gives the
        same results (up to atomic summation order)."""
        o, d = origins_t, dirs_t
        L = (tmax - o[:, 2]) / d[:, 2]
        end = o[:, :2] + d[:, :2] * L[:, None]
        coords = torch.stack([o[:, 0], o[:, 1], end[:, 0], end[:, 1]], dim=1)
        q = torch.floor((coords - coords.min(dim=0).values) / cell).to(torch.int64).clamp_(0, (1 << bits) - 1)
        code = torch.zeros(q.shape[0], dtype=torch.int64, device=q.device)
        for b in range(bits):
            for dim in range(4):
                code |= ((q[:, dim] >> b) & 1) << (4 * b + dim)
        return torch.argsort(code, stable=True).to(torch.int32).contiguous()

    def _cached_locality_order(self, origins_t, dirs_t, tmax):
        """``locality_order`` (some 250 small launches + a sort: 2 ms for 620 000 rays) remembered for the last pair of ray tensors,
        keyed on their storage and shape.  The order is a speed hint only -- every permutation gives the same results -- so a
        tensor rewritten in place at the same address merely walks in a stale order."""
        key = (origins_t.data_ptr(), dirs_t.data_ptr(), tuple(origins_t.shape), float(tmax))
        if getattr(self, "_loc_key", None) != key:
            self._loc_key, self._loc_order = key, self.locality_order(origins_t, dirs_t, tmax).long()
        return self._loc_order

    def adjoint(self, origins_t, dirs_t, w_t, tmax, Ns, out=None, accum=torch.float64, order=None):
        """out[nx,ny,nz] += G^T w  (out is zeroed when allocated here)."""
        self._sync_stream()
        R = origins_t.shape[0]
        if out is None:
            out = torch.zeros(self.shape, dtype=accum, device=self.device)
        op = _lib._V(0) if order is None else _ptr(order)
        self.ctx.call("iono_adjoint_straight_dev", _ptr(origins_t), _ptr(dirs_t), op, _ptr(w_t), R, float(tmax), int(Ns),
                      self.kind, self.rule, _ptr(out), _lib.F64 if out.dtype == torch.float64 else _lib.F32)
        return out

    def plan_adjoint(self, origins_t, dirs_t, tmax, Ns, slabs=1):
        """Bin the rays' segments by grid box ONCE (geometry only): later ``adjoint*`` calls with these same two tensors
        reduce every box in LDS and flush it once (include/ionotomo_hip.h:iono_adjoint_plan_dev).  Returns
        (segments, work units, fraction of segments not fully inside their box image) -- (0, 0, 0.0) when the grid is not
        uniform and the ray-stationary kernels stay in charge.  A planned launch checks a checksum per ray: ``plan_stale``.
        ``slabs`` > 1: the work units are ordered by z-slab, so that a back-projection can run slab by slab (``plan_slabs``,
        ``adjoint_planned_weights``): what a multi-GPU solver overlaps its exchange with."""
        import ctypes
        self._sync_stream()
        self.ctx.call("iono_adjoint_plan_slabs", int(slabs))
        self.ctx.call("iono_adjoint_plan_dev", _ptr(origins_t), _ptr(dirs_t), origins_t.shape[0], float(tmax), int(Ns), self.kind)
        self._planned = (origins_t, dirs_t)
        n, u, f = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_double(0)
        self.ctx.call("iono_adjoint_plan_info", ctypes.byref(n), ctypes.byref(u), ctypes.byref(f))
        return n.value, u.value, f.value

    def set_deterministic(self, on=True):
        """Order-independent (fixed-point) accumulation in the planned trilinear AND tricubic back-projections: run-to-run identical bits, also
        for every solver iterate built on it (include/ionotomo_hip.h: iono_set_deterministic).  Back-projections the fixed-point
        kernel does not serve raise while the mode is on."""
        self.ctx.call("iono_set_deterministic", 1 if on else 0)
        self.deterministic = bool(on)

    def plan_slabs(self):
        """(unit_lo[nslab + 1], z_lo[nslab + 1]) of the current back-projection plan: slab s = work units [unit_lo[s], unit_lo[s+1])
        and owns the node levels [z_lo[s], z_lo[s+1]) (final once slabs 0 .. s have run)."""
        import ctypes
        n = ctypes.c_int(0)
        ul, zl = (ctypes.c_int * 9)(), (ctypes.c_int * 9)()
        self.ctx.call("iono_adjoint_plan_slab_info", ctypes.byref(n), ul, zl)
        return list(ul[:n.value + 1]), list(zl[:n.value + 1])

    def adjoint_planned_weights(self, origins_t, dirs_t, tmax, Ns, out, unit_range=None, order=None):
        """out += G^T w for the ray weights the last ``adjoint_cg_step`` / ``adjoint_sirt_step`` (called with ``out=None``) left in the
        library; ``unit_range`` = (lo, hi): only those work units of the plan (one z-slab)."""
        self._sync_stream()
        if unit_range is not None:
            self.ctx.call("iono_adjoint_unit_range", int(unit_range[0]), int(unit_range[1]))
        op = _lib._V(0) if order is None else _ptr(order)
        self.ctx.call("iono_adjoint_planned_weights_dev", _ptr(origins_t), _ptr(dirs_t), op, origins_t.shape[0], float(tmax), int(Ns),
                      self.kind, self.rule, _ptr(out), _lib.F64 if out.dtype == torch.float64 else _lib.F32)
        return out

    def plan_segment_lanes(self):
        """Lanes per segment of the current back-projection plan (4, 8 or 16; 0: no plan)."""
        import ctypes
        v = ctypes.c_int(0)
        self.ctx.call("iono_adjoint_plan_segment_lanes", ctypes.byref(v))
        return v.value

    def plan_forward(self, origins_t, dirs_t, tmax, Ns):
        """Bundle the rays ONCE (geometry only): later ``forward`` calls with these same two tensors give every workgroup a
        bundle of <= 64 nearly coincident rays whose voxel neighbourhood is staged in LDS
        (include/ionotomo_hip.h:iono_forward_plan_dev).  Returns (bundles, chunks per ray, fraction of chunks served from LDS)
        -- (0, 0, 0.0) when no plan applies (non-uniform grid, float32 storage).  Keep the tensors alive; a planned launch checks a
        64-bit checksum per ray and falls back to direct loads (exact) for bundles whose rays were edited in place: ``plan_stale``."""
        import ctypes
        self._sync_stream()
        self.ctx.call("iono_forward_plan_dev", _ptr(origins_t), _ptr(dirs_t), origins_t.shape[0], float(tmax), int(Ns))
        self._fplanned = (origins_t, dirs_t)
        n, k, f = ctypes.c_int64(0), ctypes.c_int(0), ctypes.c_double(0)
        self.ctx.call("iono_forward_plan_info", ctypes.byref(n), ctypes.byref(k), ctypes.byref(f))
        return n.value, k.value, f.value

    def describe(self, op, origins_t=None, dirs_t=None, tmax=0.0, Ns=2, R=None, kind=None, ne_kind=None, bend=True):
        """Which kernel(s) the launch ``op`` ("forward", "adjoint", "trace", "fermat_forward", "fermat_adjoint", "phase_forward",
        "phase_adjoint") gets on this engine with these ray tensors -- the library's one dispatch table; returns (name, facts)."""
        op_, dp = (0, 0) if origins_t is None else (origins_t.data_ptr(), dirs_t.data_ptr())
        R = (0 if origins_t is None else origins_t.shape[0]) if R is None else R
        return self.ctx.dispatch_describe(op, op_, dp, R, tmax, Ns, self.kind if kind is None else kind, ne_kind, bend)

    def forward_plan_split(self, histogram=False):
        """How the current forward plan divides its rays (hybrid dispatch, include/ionotomo_hip.h:iono_forward_plan_split): bundles as
        cut / served by the bundle kernels, rays in served bundles / in the lanes = samples tail, the rays-per-bundle threshold the
        plan chose (1: every bundle served, 65: none) and its cost model's estimates."""
        import ctypes
        v = [ctypes.c_int64(0) for _ in range(4)]
        m = ctypes.c_int(0)
        h = (ctypes.c_int64 * 65)()
        us = (ctypes.c_double * 3)()
        self.ctx.call("iono_forward_plan_split", *[ctypes.byref(x) for x in v], ctypes.byref(m), h, us)
        out = {"bundles_cut": v[0].value, "bundles_served": v[1].value, "rays_served": v[2].value, "rays_tail": v[3].value,
               "min_rays_per_served_bundle": m.value,
               "model_us": {"all_bundles": us[0], "chosen": us[1], "lanes_samples_only": us[2]}}
        if histogram:
            out["bundles_by_ray_count"] = list(h)
        return out

    def clear_forward_plan(self):
        self.ctx.call("iono_forward_plan_clear")
        self._fplanned = None

    def clear_adjoint_plan(self):
        self.ctx.call("iono_adjoint_plan_clear")
        self._planned = None

    def adjoint_differential(self, origins_t, dirs_t, v_t, scale_t, Na, i0, tmax, Ns, out=None, accum=torch.float64,
                             order=None):
        """One launch: out += A^T (scale o v) for the differenced operator A x = G x - (G x)[i0], ray layout [Na][NtNd]
        (``scale_t`` may be None).  What CGLS (scale = W^1/2) and SIRT (scale = row normalisation) back-project."""
        self._sync_stream()
        R = origins_t.shape[0]
        assert R % Na == 0 and v_t.numel() == R and (scale_t is None or scale_t.numel() == R)
        if out is None:
            out = torch.zeros(self.shape, dtype=accum, device=self.device)
        op = _lib._V(0) if order is None else _ptr(order)
        sp = _lib._V(0) if scale_t is None else _ptr(scale_t)
        self.ctx.call("iono_adjoint_differential_straight_dev", _ptr(origins_t), _ptr(dirs_t), op, _ptr(v_t), sp, int(Na),
                      R // Na, int(i0), float(tmax), int(Ns), self.kind, self.rule, _ptr(out),
                      _lib.F64 if out.dtype == torch.float64 else _lib.F32)
        return out

    def adjoint_cg_step(self, origins_t, dirs_t, r_t, q_t, an, ad, scale_t, Na, i0, tmax, Ns, out, order=None, want_dot=True):
        """CGLS: r -= (an / ad) q in place, partials of <r, r>, out += A^T (scale o r) -- the residual update fused with the
        back-projection's differential-weights pass (include/ionotomo_hip.h:iono_adjoint_cg_step_dev).  Returns the partials."""
        self._sync_stream()
        R = origins_t.shape[0]
        assert R % Na == 0 and r_t.numel() == R and q_t.numel() == R
        part = self._partial(want_dot)
        (anp, ann), (adp, adn) = self._sc(an), self._sc(ad)
        opt = lambda t: _lib._V(0) if t is None else _ptr(t)
        self.ctx.call("iono_adjoint_cg_step_dev", _ptr(origins_t), _ptr(dirs_t), opt(order), _ptr(r_t), _ptr(q_t), anp, ann, adp, adn,
                      opt(scale_t), int(Na), R // Na, int(i0), float(tmax), int(Ns), self.kind, self.rule, opt(part), opt(out),
                      _lib.F64 if out is None or out.dtype == torch.float64 else _lib.F32)
        return part

    def adjoint_sirt_step(self, origins_t, dirs_t, tec_t, dobs_t, scale_t, weight_t, Na, i0, tmax, Ns, out, order=None, r_out=None,
                          want_dot=True):
        """SIRT: v = dobs - (tec - tec[i0]), partials of sum v^2 weight, out += A^T (scale o v) -- residual, objective and the
        back-projection's differential-weights pass in one launch (iono_adjoint_sirt_step_dev).  Returns the partials."""
        self._sync_stream()
        R = origins_t.shape[0]
        assert R % Na == 0 and tec_t.numel() == R and dobs_t.numel() == R
        part = self._partial(want_dot)
        opt = lambda t: _lib._V(0) if t is None else _ptr(t)
        self.ctx.call("iono_adjoint_sirt_step_dev", _ptr(origins_t), _ptr(dirs_t), opt(order), _ptr(tec_t), _ptr(dobs_t), opt(scale_t),
                      opt(weight_t), int(Na), R // Na, int(i0), float(tmax), int(Ns), self.kind, self.rule, opt(r_out), opt(part),
                      opt(out), _lib.F64 if out is None or out.dtype == torch.float64 else _lib.F32)
        return part

    def adjoint_residual(self, origins_t, dirs_t, tec_t, dobs_t, cdct_t, Na, i0, tmax, Ns, out=None,
                         accum=torch.float64, order=None):
        """One launch: dd = (tec - tec[i0] - dobs)/(CdCt + 1e-15) -> differential weights -> G^T.
        Ray layout [Na][NtNd]."""
        self._sync_stream()
        R = origins_t.shape[0]
        assert R % Na == 0
        if out is None:
            out = torch.zeros(self.shape, dtype=accum, device=self.device)
        op = _lib._V(0) if order is None else _ptr(order)
        self.ctx.call("iono_adjoint_residual_straight_dev", _ptr(origins_t), _ptr(dirs_t), op, _ptr(tec_t), _ptr(dobs_t),
                      _ptr(cdct_t), int(Na), R // Na, int(i0), float(tmax), int(Ns), self.kind, self.rule, _ptr(out),
                      _lib.F64 if out.dtype == torch.float64 else _lib.F32)
        return out

    def forward_phase(self, origins_t, dirs_t, Na, Nt, Nd, tmax, Ns, freqs, clock_t, const_t, i0, out=None):
        """g[Na,Nt,Nd,Nf] of inversion/iterative_newton.py:86-127 for straight rays (samples in-kernel); the grid must hold
        ne = K exp(mu).  ``freqs``: host array; ``clock_t`` [Na,Nt], ``const_t`` [Na]: device tensors."""
        self._sync_stream()
        freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        R = origins_t.shape[0]
        assert R == Na * Nt * Nd
        if out is None:
            out = torch.empty((Na, Nt, Nd, freqs.size), dtype=torch.float64, device=self.device)
        work = torch.empty(R * freqs.size, dtype=torch.float64, device=self.device)
        self.ctx.call("iono_forward_phase_straight_dev", _ptr(origins_t), _ptr(dirs_t), int(Na), int(Nt), int(Nd), float(tmax),
                      int(Ns), _lib._dp(freqs), freqs.size, _ptr(clock_t), _ptr(const_t), int(i0), self.rule, _ptr(work), _ptr(out))
        return out

    def adjoint_phase(self, origins_t, dirs_t, y_t, Na, tmax, Ns, freqs, i0, order=None, wrt_log_model=True, out=None):
        """d/d mu (``wrt_log_model``) or d/d ne of sum y g for the phase observable: one traversal per 8 frequencies that
        gathers ne at every sample and scatters the transpose.  ``y_t``: [Na, Nt*Nd, Nf] (= dS/dg)."""
        self._sync_stream()
        freqs = np.ascontiguousarray(freqs, dtype=np.float64)
        R = origins_t.shape[0]
        assert y_t.numel() == R * freqs.size and y_t.is_contiguous()
        if out is None:
            out = torch.zeros(self.shape, dtype=torch.float64, device=self.device)
        work = torch.empty(R * freqs.size, dtype=torch.float64, device=self.device)
        op = _lib._V(0) if order is None else _ptr(order)
        self.ctx.call("iono_adjoint_phase_straight_dev", _ptr(origins_t), _ptr(dirs_t), op, _ptr(y_t), int(Na), R // Na, float(tmax),
                      int(Ns), _lib._dp(freqs), freqs.size, int(i0), self.rule, _ptr(work), int(bool(wrt_log_model)), _ptr(out))
        return out

    def trace_fermat(self, origins_t, dirs_t, tmax, Ns, frequency, bend=False, kind="linear", substeps=4, out=None, type="z"):
        """rays[R,4,Ns] (x,y,z,s) of the Fermat ray ODE, on the device; the grid must hold ne [m^-3].  ``substeps``: RK4 steps per
        output sample, or "auto" / ("auto", rtol): chosen by step doubling (``choose_fermat_substeps``)."""
        self._sync_stream()
        substeps = self._fermat_substeps(substeps, origins_t, dirs_t, tmax, Ns, frequency, bend, kind, type)
        R = origins_t.shape[0]
        if out is None:
            out = torch.empty((R, 4, int(Ns)), dtype=torch.float64, device=self.device)
        self.ctx.call("iono_trace_fermat_dev", _ptr(origins_t), _ptr(dirs_t), R, float(tmax), int(Ns), float(frequency),
                      int(bool(bend)), _lib.interp_kind(kind), int(substeps), _lib.ray_type(type), _ptr(out))
        return out

    # a tricubic refractive index is traced 13 x faster by the 8-lanes-per-ray tracer with its cached stencil than by the lanes = rays
    # stepper of the fused kernel (config 3: 2.3 against 31 ms): below this many bytes of rays[R,4,Ns] the two-step path serves it
    FUSED_CUBIC_ABOVE_BYTES = 8 << 30

    def fermat_lm_ok(self, kind, ne_kind, R, transpose=False, bend=True):
        """Would ``forward_fermat`` (``transpose``: ``adjoint_fermat``) run the fused tricubic-index kernel (k_fermat_tec_lm) for this
        launch?  Asked of the library (iono_fermat_lm_ok), never re-derived here."""
        import ctypes
        ok = ctypes.c_int(0)
        as_kind = lambda k: k if isinstance(k, int) else _lib.interp_kind(k)
        self.ctx.call("iono_fermat_lm_ok", as_kind(kind), as_kind(ne_kind), int(R), int(bool(transpose)), int(bool(bend)), ctypes.byref(ok))
        return bool(ok.value)

    def _two_step_fermat(self, R, Ns, kind, fused, adjoint=False, ne_kind=None, bend=True):
        """True: trace into a TEMPORARY rays[R,4,Ns] tensor (32 R Ns bytes of device memory) and integrate along it.  Round 4: the
        FORWARD through a tricubic index on ideal-uniform axes is fused too (k_fermat_tec_lm: a few lanes per ray on node records,
        streaming quadrature); round 5: so is its TRANSPOSE for large batches of bending rays -- the two-step route remains the
        default for small transposes and for non-uniform axes."""
        if fused is not None:
            return not fused
        need = R * 4 * int(Ns) * 8
        if _lib.interp_kind(kind) != _lib.interp_kind("cubic") or need > self.FUSED_CUBIC_ABOVE_BYTES:
            return False
        if self.fermat_lm_ok(kind, self.kind if ne_kind is None else ne_kind, R, transpose=adjoint, bend=bend):
            return False                       # the library's own dispatch predicate: its fused tricubic-index kernel serves this launch
        try:                                   # never a hidden allocation beyond half of what the device has free
            free = torch.cuda.mem_get_info(self.device)[0]
        except Exception:
            free = 0
        return need <= free // 2

    def choose_fermat_substeps(self, origins_t, dirs_t, tmax, Ns, frequency, bend=True, kind="linear", type="z", rtol=None, atol=None,
                               max_substeps=32, fraction=0.01):
        """Error control for the fixed-step RK4 tracer (the reference integrates with adaptive LSODA, ``odeint`` default
        rtol = atol = 1.49e-8: inversion/fermat.py:163-167): a strided sample of <= 1 % of the rays (>= 208) is traced at s and 2 s
        steps per output sample, s = 1, 2, 4, ...; returns (the smallest s whose step-doubling difference on x, y, s meets
        rtol |y| + atol, report).  One small launch per level; call again when the node values have changed enough to matter.  The
        report lists, per level, the largest position differences (km) and the relative change of the TEC along the sample rays.
        MEASURED (profiles/r06_fermat_steps.json): on this repo's turbulent test ionosphere a TRILINEAR index converges first order
        (its gradient jumps at cell faces) and no affordable step count reaches 1.49e-8; a TRICUBIC index converges second order."""
        from .inversion import fermat as F
        rtol = F.ODEINT_RTOL if rtol is None else float(rtol)
        idx = torch.as_tensor(F.sample_indices(origins_t.shape[0], fraction), device=self.device)
        o_s, d_s = origins_t.index_select(0, idx).contiguous(), dirs_t.index_select(0, idx).contiguous()

        tec = {}

        def trace(sub):
            rays = self.trace_fermat(o_s, d_s, tmax, Ns, frequency, bend=bend, kind=kind, substeps=sub, type=type)
            tec[sub] = self.forward_rays(rays).cpu().numpy()              # the observable along these rays (the engine's interpolant)
            return rays.cpu().numpy()
        return F.choose_substeps(trace, rtol, atol, max_substeps=max_substeps, observe=lambda sub: tec[sub])

    def _fermat_substeps(self, substeps, origins_t, dirs_t, tmax, Ns, frequency, bend, kind, type):
        """``substeps``: an int, "auto" (step doubling at the reference's odeint tolerance) or ("auto", rtol[, atol]).  The choice is
        remembered per (ray tensors, sampling, frequency, interpolant, tolerance) until the node values change; ``fermat_step_report``
        keeps what it was based on."""
        if not (substeps == "auto" or (isinstance(substeps, tuple) and substeps and substeps[0] == "auto")):
            return int(substeps)
        if not bend:
            return 1 if type == "z" else 4
        tol = tuple(substeps[1:]) if isinstance(substeps, tuple) else ()
        key = (origins_t.data_ptr(), dirs_t.data_ptr(), tuple(origins_t.shape), float(tmax), int(Ns), float(frequency), str(kind), str(type), tol)
        hit = self._fermat_steps.get(key)
        if hit is None or hit[0] != self._values_serial:
            sub, rep = self.choose_fermat_substeps(origins_t, dirs_t, tmax, Ns, frequency, bend=bend, kind=kind, type=type,
                                                   rtol=tol[0] if len(tol) > 0 else None, atol=tol[1] if len(tol) > 1 else None)
            self._fermat_steps = {key: (self._values_serial, sub, rep)}
            hit = self._fermat_steps[key]
        self.fermat_step_report = hit[2]
        return hit[1]

    def forward_fermat(self, origins_t, dirs_t, tmax, Ns, frequency, bend=True, kind="linear", substeps=4, type="z", ne_kind=None,
                       ne_scale=1.0, out=None, fused=None):
        """tec[R] along the Fermat rays WITHOUT materialising them: RK4 stepper + streaming non-uniform Simpson in one kernel
        (== ``forward_rays(trace_fermat(...))`` to rounding; include/ionotomo_hip.h:iono_forward_tec_fermat_dev).  The grid must
        hold ne [m^-3]; ``kind`` interpolates the refractive index, ``ne_kind`` (default: the engine's) the integrand.
        ``fused``: None = the fused kernel, except for a tricubic index while the ray tensor fits (see above); True / False force."""
        self._sync_stream()
        R = origins_t.shape[0]
        if out is None:
            out = torch.empty(R, dtype=torch.float64, device=self.device)
        substeps = self._fermat_substeps(substeps, origins_t, dirs_t, tmax, Ns, frequency, bend, kind, type)
        if self._two_step_fermat(R, Ns, kind, fused, ne_kind=ne_kind, bend=bend):
            rays = self.trace_fermat(origins_t, dirs_t, tmax, Ns, frequency, bend=bend, kind=kind, substeps=substeps, type=type)
            self.forward_rays(rays, out=out, kind=ne_kind)
            if ne_scale != 1.0:
                out.mul_(float(ne_scale))
            return out
        nk = self.kind if ne_kind is None else _lib.interp_kind(ne_kind)
        order, res = None, out
        if R >= 4096:            # lanes = rays: neighbouring lanes on nearly coincident rays share the lines they load (27 -> 15 ms at 620 000 rays)
            order = self._cached_locality_order(origins_t, dirs_t, tmax)
            origins_t, dirs_t, res = origins_t[order].contiguous(), dirs_t[order].contiguous(), torch.empty_like(out)
        self.ctx.call("iono_forward_tec_fermat_dev", _ptr(origins_t), _ptr(dirs_t), R, float(tmax), int(Ns), float(frequency),
                      int(bool(bend)), _lib.interp_kind(kind), int(substeps), _lib.ray_type(type), nk, self.rule, float(ne_scale), _ptr(res))
        if order is not None:
            out.view(-1)[order] = res.view(-1)
        return out

    def adjoint_fermat(self, origins_t, dirs_t, w_t, tmax, Ns, frequency, bend=True, kind="linear", substeps=4, type="z",
                       ne_kind=None, ne_scale=1.0, out=None, fused=None):
        """out[nx,ny,nz] += transpose of ``forward_fermat`` applied to w (the ray paths held fixed): re-trace and scatter.
        Large batches are walked in ``locality_order`` (neighbouring lanes = nearly coincident rays, so the hardware atomics of a
        wave-instruction fall into a few lines instead of 64: a sum over rays does not care about their order).  ``fused`` as in
        ``forward_fermat``."""
        self._sync_stream()
        R = origins_t.shape[0]
        substeps = self._fermat_substeps(substeps, origins_t, dirs_t, tmax, Ns, frequency, bend, kind, type)
        if self._two_step_fermat(R, Ns, kind, fused, adjoint=True, ne_kind=ne_kind, bend=bend):
            if out is None:
                out = torch.zeros(self.shape, dtype=torch.float64, device=self.device)
            rays = self.trace_fermat(origins_t, dirs_t, tmax, Ns, frequency, bend=bend, kind=kind, substeps=substeps, type=type)
            w = w_t.reshape(-1) if ne_scale == 1.0 else w_t.reshape(-1) * float(ne_scale)
            nk = self.kind if ne_kind is None else _lib.interp_kind(ne_kind)
            self.ctx.call("iono_adjoint_rays_dev", _ptr(rays), _ptr(w.contiguous()), R, int(Ns), nk, self.rule, _ptr(out), _lib.F64)
            return out
        if R >= 4096:
            order = self._cached_locality_order(origins_t, dirs_t, tmax)
            origins_t, dirs_t, w_t = origins_t[order].contiguous(), dirs_t[order].contiguous(), w_t.reshape(-1)[order].contiguous()
        if out is None:
            out = torch.zeros(self.shape, dtype=torch.float64, device=self.device)
        nk = self.kind if ne_kind is None else _lib.interp_kind(ne_kind)
        self.ctx.call("iono_adjoint_fermat_dev", _ptr(origins_t), _ptr(dirs_t), _ptr(w_t), R, float(tmax), int(Ns), float(frequency),
                      int(bool(bend)), _lib.interp_kind(kind), int(substeps), _lib.ray_type(type), nk, self.rule, float(ne_scale), _ptr(out))
        return out

    def forward_rays(self, rays_t, out=None, kind=None):
        """tec[R] along explicit samples rays[R,4,Ns] (device), non-uniform Simpson in-kernel."""
        self._sync_stream()
        R, _, Ns = rays_t.shape
        if out is None:
            out = torch.empty(R, dtype=torch.float64, device=self.device)
        self.ctx.call("iono_forward_tec_rays_dev", _ptr(rays_t), R, int(Ns), self.kind if kind is None else _lib.interp_kind(kind),
                      self.rule, _ptr(out))
        return out

    def smooth(self, phi_t, kx, ky, kz, out=None, work=None):
        """C_m phi with the separable stencil kx x ky x kz (host arrays, odd length), on the device."""
        self._sync_stream()
        kx, ky, kz, h = _lib._smooth_args(kx, ky, kz)
        out = torch.empty_like(phi_t) if out is None else out
        work = torch.empty_like(phi_t) if work is None else work
        self.ctx.call("iono_smooth_separable_dev", _ptr(phi_t), _ptr(out), _ptr(work), _lib._dp(kx), _lib._dp(ky), _lib._dp(kz), h)
        return out

    def subtract_reference(self, tec_t, Na, i0):
        self._sync_stream()
        self.ctx.call("iono_subtract_reference_dev", _ptr(tec_t), int(Na), tec_t.numel() // Na, int(i0))
        return tec_t

    def tune_partition(self, which, launch, R, fractions=None, refine=2):
        """Balance a chunked kernel's work over the chip by MEASURED cost.  ``which``: ``_lib.WALK_FORWARD`` (one
        chunk of the ray walk per resident wave) or ``_lib.WALK_ADJOINT`` (LDS-tiled back-projection, one chunk per
        resident workgroup).  ``launch()`` runs the kernel to be tuned (same rays, same walk order).  The kernel
        reports the cycles each contiguous chunk of the walk took; from them a cost-per-ray profile is built and the
        walk is re-cut into equal-COST chunks.  For the adjoint the walk is cut into ``len(fractions)`` levels, one
        chunk per workgroup and level, level l holding ``fractions[l]`` of the total cost: the first level is
        assigned statically, the later -- smaller -- chunks are handed out as workgroups finish (guided
        self-scheduling).  ``refine`` further rounds re-measure with the new chunks.  The fastest partition (timed
        with events) is kept in the context and used by every later launch with the same ray count.  Geometry only,
        like ``locality_order``: results never depend on it.  Returns a dict of timings, or None when the launch did
        not go through a chunked kernel."""
        ctx = self.ctx
        R = int(R)
        if fractions is None:
            fractions = (0.75, 0.25) if which == _lib.WALK_ADJOINT else (1.0,)
        if which == _lib.WALK_FORWARD:
            fractions = (1.0,)

        def timed(n=3):
            best = float("inf")
            for _ in range(n):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                launch()
                b.record()
                b.synchronize()
                best = min(best, a.elapsed_time(b))
            return best

        def density(cyc, starts):
            lens = np.diff(starts)
            return np.repeat(cyc / np.maximum(lens, 1), lens)

        def guided(dens, units):
            cum = np.concatenate([[0.0], np.cumsum(dens)])
            edges = np.concatenate([[0.0], np.cumsum(fractions) / np.sum(fractions)]) * cum[-1]
            targets = np.concatenate([np.linspace(edges[l], edges[l + 1], units, endpoint=False)
                                      for l in range(len(fractions))] + [[cum[-1]]])
            starts = np.maximum.accumulate(np.searchsorted(cum, targets, side="left").astype(np.int64))
            starts[0], starts[-1] = 0, R
            return np.minimum(starts, R)

        ctx.walk_partition_set(which, None, R)
        launch()                                             # warm-up, and tells us the kernel's geometry
        cyc, units = ctx.walk_cycles(which)
        if units < 2 or cyc.size != units or not np.all(cyc > 0):
            return None
        t_equal = timed()
        launch()
        cyc, _ = ctx.walk_cycles(which)
        base, rem = divmod(R, units)
        starts = np.array([b * base + min(b, rem) for b in range(units + 1)], dtype=np.int64)
        dens = density(cyc.astype(np.float64), starts)
        best_t, best_starts, tried = t_equal, None, []
        for it in range(refine + 1):
            starts = guided(dens, units)
            ctx.walk_partition_set(which, starts, R)
            t = timed()
            tried.append(t)
            if t < best_t:
                best_t, best_starts = t, starts
            if it < refine:
                launch()
                cyc, _ = ctx.walk_cycles(which)
                dens = 0.5 * dens + 0.5 * density(cyc.astype(np.float64), starts)
        ctx.walk_partition_set(which, best_starts, R)
        return {"equal_count_ms": t_equal, "tuned_ms": best_t, "tried_ms": tried, "units": int(units),
                "chunks": int(0 if best_starts is None else best_starts.size - 1)}

    def tune_adjoint_partition(self, launch, R, fractions=(0.75, 0.25), refine=2):
        return self.tune_partition(_lib.WALK_ADJOINT, launch, R, fractions, refine)

    def tune_forward_partition(self, launch, R, refine=3):
        return self.tune_partition(_lib.WALK_FORWARD, launch, R, None, refine)

    def axpby_(self, y, x, a_num=None, a_den=None, a_sign=1.0, b_num=None, b_den=None):
        """y = (a_sign a_num / a_den) x + (b_num / b_den) y in one pass; the coefficients are 0-dim DEVICE tensors
        (None = 1), so step lengths built from all-reduced dot products never synchronise with the host."""
        self._sync_stream()
        assert y.is_contiguous() and x.is_contiguous() and y.numel() == x.numel() and y.dtype == x.dtype == torch.float64
        sc = [None if t is None else t.reshape(1) for t in (a_num, a_den, b_num, b_den)]     # keep alive over the call
        self.ctx.call("iono_vec_axpby_dev", _ptr(y), _ptr(x), y.numel(), *[0 if t is None else _ptr(t) for t in sc[:2]],
                      float(a_sign), *[0 if t is None else _ptr(t) for t in sc[2:]])
        return y

    # -- RCCL behind the C-ABI (the route for hosts without torch.distributed; parallel.py uses torch.distributed) ----
    def comm_unique_id(self):
        return self.ctx.comm_unique_id()

    def comm_init(self, comm_id, rank, nranks):
        self._sync_stream()
        self.ctx.comm_init(comm_id, rank, nranks)

    def comm_allreduce_(self, t):
        """Sum the float64 / float32 device tensor ``t`` over the ranks of the communicator, in place, on the current stream."""
        assert t.is_contiguous() and t.dtype in (torch.float64, torch.float32)
        self._sync_stream()
        self.ctx.comm_allreduce_dev(t.data_ptr(), t.numel(), _lib.F64 if t.dtype == torch.float64 else _lib.F32)
        return t

    def comm_destroy(self):
        self.ctx.comm_destroy()

    # -- fused solver passes (csrc/iono_solver_kernels.h): dot products as per-workgroup partials, compact grid vectors ----
    def new_grid_buffer(self):
        """(padded, view): a zeroed float64 buffer the kernels can read IN PLACE as grid values (``bind_values``) and its
        [nx,ny,nz] view; the tail beyond the grid stays zero (unclamped far-corner reads)."""
        n = ctypes_int64()
        self.ctx.call("iono_grid_padded_size", _byref(n))
        padded = torch.zeros(n.value, dtype=torch.float64, device=self.device)
        return padded, padded[:self.ncells].view(self.shape)

    def bind_values(self, padded):
        """Kernels read ``padded`` (from ``new_grid_buffer``) as the grid values from now on; None: back to the
        library's own storage.  Call ``values_changed`` after modifying the buffer."""
        self._sync_stream()
        self.ctx.call("iono_grid_bind_values_dev", _lib._V(0) if padded is None else _ptr(padded))
        self._bound = padded                                  # keep it alive
        self._values_serial += 1

    def values_changed(self):
        self.ctx.call("iono_grid_values_changed")
        self._values_serial += 1

    def _partial(self, want):
        return torch.empty(_lib.NPART, dtype=torch.float64, device=self.device) if want else None

    @staticmethod
    def _sc(t):
        """device scalar -> (pointer, count): None = 1.0, 1 element, or NPART partial sums"""
        return (_lib._V(0), 0) if t is None else (_ptr(t), int(t.numel()))

    def rays_combine(self, tec, Na, i0, a, b, dobs=None, s1=None, s2=None, out=None, want_dot=True):
        """out = s1 * (a * (tec - tec[i0]) + b * dobs) over rays [Na][NtNd]; partials of sum out^2 * s2."""
        self._sync_stream()
        out = torch.empty_like(tec) if out is None else out
        part = self._partial(want_dot)
        opt = lambda t: _lib._V(0) if t is None else _ptr(t)
        self.ctx.call("iono_rays_combine_dev", _ptr(tec), opt(dobs), opt(s1), opt(s2), int(Na), tec.numel() // Na, int(i0),
                      float(a), float(b), _ptr(out), opt(part))
        return out, part

    SMALL_RAYS = 32768

    def small_ray_pass(self, mode, tec, scale, r, Na, i0, q=None, gamma=None, dobs=None, weight=None, w=None):
        """The ray-sized passes of one CG (``mode`` 0) or SIRT (1) iteration for at most ``SMALL_RAYS`` rays in ONE launch
        (include/ionotomo_hip.h:iono_small_ray_pass_dev): CG: q = scale (tec - tec[i0]), <q, q>; r -= (gamma / <q, q>) q, <r, r>;
        SIRT: r = dobs - (tec - tec[i0]), sum r^2 weight; both: w = the differential back-projection's ray weights of r * scale
        (feed ``adjoint(origins, dirs, w, ...)``).  Returns (dot1, dot2, w): one-element device tensors + the weights."""
        self._sync_stream()
        R = tec.numel()
        w = torch.empty(R, dtype=torch.float64, device=self.device) if w is None else w
        dots = torch.empty(2, dtype=torch.float64, device=self.device)
        gp, gn = self._sc(gamma)
        opt = lambda t: _lib._V(0) if t is None else _ptr(t)
        self.ctx.call("iono_small_ray_pass_dev", int(mode), _ptr(tec), opt(dobs), _ptr(scale), opt(weight), _ptr(r), opt(q), int(Na),
                      R // int(Na), int(i0), gp, gn, _ptr(dots), _lib._V(dots.data_ptr() + 8), _ptr(w))
        return dots[0:1], dots[1:2], w

    def axpby_dot_(self, y, x, an=None, ad=None, a_sign=1.0, bn=None, bd=None, want_dot=True):
        """y = (a_sign an / ad) x + (bn / bd) y in place; returns the partials of sum y^2 (or None)."""
        self._sync_stream()
        part = self._partial(want_dot)
        (anp, ann), (adp, adn), (bnp, bnn), (bdp, bdn) = (self._sc(t) for t in (an, ad, bn, bd))
        self.ctx.call("iono_vec_axpby_dot_dev", _ptr(y), _ptr(x), y.numel(), anp, ann, adp, adn, float(a_sign), bnp, bnn, bdp, bdn,
                      _lib._V(0) if part is None else _ptr(part))
        return part

    def compact_gather(self, full, idx, out=None, zero=True, want_dot=True):
        self._sync_stream()
        out = torch.empty(idx.numel(), dtype=torch.float64, device=self.device) if out is None else out
        part = self._partial(want_dot)
        self.ctx.call("iono_compact_gather_dev", _ptr(full), _ptr(idx), idx.numel(), _ptr(out), int(bool(zero)),
                      _lib._V(0) if part is None else _ptr(part))
        return out, part

    def compact_scatter(self, full, idx, src):
        self._sync_stream()
        self.ctx.call("iono_compact_scatter_dev", _ptr(full), _ptr(idx), idx.numel(), _ptr(src))

    def compact_cg_update(self, x, p, s, idx, full_p, an, ad, bn, bd):
        self._sync_stream()
        (anp, ann), (adp, adn), (bnp, bnn), (bdp, bdn) = (self._sc(t) for t in (an, ad, bn, bd))
        self.ctx.call("iono_compact_cg_update_dev", _ptr(x), _ptr(p), _ptr(s), _ptr(idx), idx.numel(), _ptr(full_p), anp, ann, adp,
                      adn, bnp, bnn, bdp, bdn)

    def compact_sirt_update(self, x, C, full_s, idx, full_x, relax=1.0, nonneg=False, want_max=False):
        self._sync_stream()
        part = self._partial(want_max)
        self.ctx.call("iono_compact_sirt_update_dev", _ptr(x), _ptr(C), _ptr(full_s), _ptr(idx), idx.numel(), _ptr(full_x),
                      float(relax), int(bool(nonneg)), _lib._V(0) if part is None else _ptr(part))
        return part

    def check_oob(self):
        return self.ctx.check_oob()

    def plan_stale(self):
        """True (flag cleared) if a planned launch since the last call was handed rays that are not the ones its plan was made
        for -- a planned tensor was edited in place (``o_t.copy_(new)``).  The forward took the direct loads for the bundles
        concerned (its TEC is exact whatever the bundling: the result is still correct, just slower); a back-projection poisoned
        the edited rays, so its result holds NaN.  Synchronises the stream (include/ionotomo_hip.h:iono_plan_stale)."""
        return self.ctx.plan_stale()

    def check_plans(self):
        """Raise ``ValueError`` (IONO_ERR_ARG semantics) if ``plan_stale()``: re-plan (``plan_forward`` / ``plan_adjoint``) after
        changing planned ray tensors in place."""
        if self.plan_stale():
            raise ValueError("a planned ray tensor was modified in place after plan_forward / plan_adjoint: forward results were "
                             "computed without the plan, back-projections of the modified rays are NaN; build the plans again")
