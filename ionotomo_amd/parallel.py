"""Ray sharding over GPUs: one process per GPU, rays split by (time, direction) pair, grid replicated.

The reference's only "reduction across workers" is ``da.sum(da.stack([...]))`` over per-direction
dask tasks that each return a full [nx,ny,nz] gradient (inversion/gradient.py:52-54); its forward
splits rays by antenna or direction with no exchange (inversion/forward_equation.py:60-67).
Here:

* rays are laid out [Na][P] with P = Nt*Nd (time,direction) pairs; rank r owns a contiguous block
  of pairs and ALL Na antennas of each pair, so the reference-antenna differencing
  ``tec - tec[i0]`` (forward_equation.py:50) never leaves the GPU;
* forward: no collective;
* adjoint: every rank back-projects its rays into a full-size partial gradient, then ONE
  ``all_reduce(sum)`` (RCCL over xGMI with the nccl backend; gloo in the CPU tests).  xGMI is
  point-to-point, so the ring all-reduce is bound by one link (~150 GB/s peak): a dense 256^3 f64
  gradient (128 MiB) costs 2 (N-1)/N x 128 MiB / link rate = milliseconds per iteration, more than
  the kernels.  The ray fan only touches a fraction of the box (~20 % in the bench geometry), and the
  geometry is fixed for the whole inversion, so ``exchange="compact"`` finds the union of touched
  nodes once (one unit-weight back-projection + a MAX all-reduce of the mask) and afterwards
  all-reduces only those nodes, gathered into a contiguous buffer (optionally in float32);
* scalars (objective, step lengths): all_reduce of a few doubles.

``engine`` is any object with ``forward(origins, dirs, tmax, Ns) -> tec`` and
``adjoint_residual(...)/adjoint(...)`` over torch tensors: ``ionotomo_amd.engine.RayEngine`` in
production; the tests drive the same code with a CPU stand-in over gloo.
"""
import torch
import torch.distributed as dist


def pair_block(n_pairs, world, rank):
    """Contiguous, balanced block [lo, hi) of the P (time,direction) pairs for ``rank``."""
    base, rem = divmod(n_pairs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# Every rank solves ITS OWN problems (independent solves shared over the ranks: inversion/parallel_solves.py:solve_share) although a
# process group exists: the sharded code paths then see one rank -- no pair blocks, no collective, nothing to wait for on a rank
# that holds other solves.  Set it before building the ``ShardedRays`` and leave it set while they are in use.
INDEPENDENT_RANKS = False


def world_info():
    if dist.is_available() and dist.is_initialized() and not INDEPENDENT_RANKS:
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


# Test / measurement switch (tests/test_gpu_engine.py, profiles/tools/nccl_1rank.py): with a process group of ONE rank initialised,
# run every collective of the multi-rank code paths anyway (a one-rank sum is the identity, so results must not change by a bit).
# This is how the torch-nccl (= RCCL) launch path, the compact gather / scatter around it and the overlap pipeline are exercised
# and timed on a 1-GPU box; production never sets it.
FORCE_COLLECTIVES = False


def multi_rank():
    """True when the collectives of the sharded code paths must run: more than one rank (or the switch above on a 1-rank group)."""
    if INDEPENDENT_RANKS or not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or FORCE_COLLECTIVES


def all_reduce_sum_(t):
    """In-place sum over ranks (no-op on one rank)."""
    if multi_rank():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


class GradientExchange(object):
    """Sum of the ranks' partial gradients (module docstring).  ``mode``: "dense" (all-reduce the whole
    grid), "compact" (only the nodes any rank's rays touch) or "auto" (compact when that set is below
    ``dense_above`` of the grid).  ``reduce_dtype=torch.float32`` halves the bytes on the links; the sum is
    then exact to ~1e-7 relative (the solvers' iterates change at that level)."""

    def __init__(self, mode="auto", reduce_dtype=None, dense_above=0.6):
        assert mode in ("dense", "compact", "auto", "sharded", "overlap")
        self.sharded = mode == "sharded"        # fused solvers: reduce-scatter + sharded update + all-gather (below)
        self.overlap = mode == "overlap"        # fused solvers: the compact sum leaves slab by slab while the back-projection still runs
        if self.sharded or self.overlap:
            mode = "compact"
        self.mode, self.reduce_dtype, self.dense_above = mode, reduce_dtype, float(dense_above)
        self.index = None          # int64 [n_active] flat node indices, identical on every rank
        self.fraction = 1.0

    def plan(self, touched):
        """``touched``: this rank's gradient of unit ray weights (> 0 exactly at the nodes its rays reach)."""
        if not multi_rank() or self.mode == "dense":
            return self
        mask = (touched.reshape(-1) != 0).to(torch.int32)
        dist.all_reduce(mask, op=dist.ReduceOp.MAX)
        index = mask.nonzero().reshape(-1)
        self.fraction = index.numel() / max(mask.numel(), 1)
        if self.mode == "compact" or self.fraction < self.dense_above:
            self.index = index
        return self

    def sum_(self, g):
        """In-place sum of ``g`` over ranks; nodes outside the plan are zero on every rank and stay so."""
        if not multi_rank():
            return g
        if self.index is None:
            if self.reduce_dtype is not None and self.reduce_dtype != g.dtype:
                buf = g.to(self.reduce_dtype)
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
                g.copy_(buf)
                return g
            return all_reduce_sum_(g)
        flat = g.view(-1)
        buf = flat.index_select(0, self.index)
        if self.reduce_dtype is not None:
            buf = buf.to(self.reduce_dtype)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        flat.index_copy_(0, self.index, buf.to(g.dtype))
        return g


class ShardedRays(object):
    """This rank's slice of a [Na][P] ray bundle (+ optional data), resident on the engine's device."""

    def __init__(self, engine, origins, directions, tmax, Ns, dobs=None, cdct=None, i0=0, exchange="auto",
                 reduce_dtype=None, tune=True, plan=True):
        """origins/directions: [Na,P,3] (numpy or tensor, FULL problem); dobs/cdct: [Na,P].
        ``exchange`` / ``reduce_dtype``: see ``GradientExchange``; ``tune``: balance the back-projection's work
        partition by measurement (a few extra launches, once)."""
        self.engine = engine
        self.world, self.rank = world_info()
        self.multi = multi_rank()           # collectives run (world > 1, or the FORCE_COLLECTIVES switch on a 1-rank group)
        o = torch.as_tensor(origins, dtype=torch.float64)
        d = torch.as_tensor(directions, dtype=torch.float64)
        self.Na, self.P = o.shape[0], o.shape[1]
        self.lo, self.hi = pair_block(self.P, self.world, self.rank)
        dev = engine.device
        self.origins = o[:, self.lo:self.hi].reshape(-1, 3).contiguous().to(dev)
        self.dirs = d[:, self.lo:self.hi].reshape(-1, 3).contiguous().to(dev)
        self.tmax, self.Ns, self.i0 = float(tmax), int(Ns), int(i0)
        self.P_local = self.hi - self.lo
        self.R_local = self.Na * self.P_local
        self.dobs = None if dobs is None else self.slice(dobs)
        self.cdct = None if cdct is None else self.slice(cdct)
        # walk orders for the unplanned kernels (speed only; ``order`` / ``forward_order``): computed on first use -- some 250 small
        # launches and a sort, 2-3 ms at the bench shape, which a planned geometry never needs
        self._order = self._forward_order = None
        # bundle plan of the forward (speed only, once per geometry; engine.plan_forward): every later forward of THESE two
        # tensors gives a workgroup <= 64 nearly coincident rays and stages their voxel neighbourhood in LDS.  First: the
        # back-projection plan below is built along its walk (2.2 -> 1 ms)
        self.forward_plan = None
        if plan and hasattr(engine, "plan_forward") and self.R_local > 0:
            self.forward_plan = engine.plan_forward(self.origins, self.dirs, self.tmax, self.Ns)
        # node-stationary back-projection plan (speed only, once per geometry; engine.plan_adjoint): when the grid is
        # uniform every later adjoint of THESE two tensors reduces each grid box in LDS and flushes it once
        self.plan = None
        self.slabs = None               # exchange="overlap": (unit_lo[], z_lo[]) of the back-projection plan's z-slabs
        if plan and hasattr(engine, "plan_adjoint") and self.R_local > 0:
            # (slab by slab only the planned TRILINEAR back-projection can run -- a tricubic fold's stencil crosses slab boundaries:
            #  a cubic engine keeps one slab and exchanges compactly)
            if exchange == "overlap" and self.multi and hasattr(engine, "plan_slabs") and getattr(engine, "trilinear", True):
                self.plan = engine.plan_adjoint(self.origins, self.dirs, self.tmax, self.Ns, slabs=self.OVERLAP_SLABS)
                # (a segment never leaves its z-layer of boxes -- the plan cuts rays at layer boundaries -- so a slab's node levels are
                #  final once its units have run even where samples overhang their box image in x or y and go by global atomics)
                if self.plan[0]:
                    self.slabs = engine.plan_slabs()
            else:
                self.plan = engine.plan_adjoint(self.origins, self.dirs, self.tmax, self.Ns)
        # measured load balance of the ray-stationary back-projection (used when no plan could be built)
        self.partition = None
        if tune and not (self.plan and self.plan[0]) and hasattr(engine, "tune_adjoint_partition") and self.R_local > 0:
            ones = torch.ones(self.R_local, dtype=torch.float64, device=dev)
            scratch = torch.zeros(engine.shape, dtype=torch.float64, device=dev)

            def launch():
                scratch.zero_()
                engine.adjoint(self.origins, self.dirs, ones, self.tmax, self.Ns, out=scratch, order=self.order)
            self.partition = engine.tune_adjoint_partition(launch, self.R_local)
            del scratch
        self.exchange = GradientExchange(exchange, reduce_dtype)
        self._active = None
        self.slab_ranges = None
        if self.multi and exchange != "dense":
            self.exchange.plan(self._touched())
        if self.exchange.overlap and self.multi:
            # every rank must have a slab plan with the same node-level boundaries (same grid -> same box layers), else nobody overlaps
            zl = self.slabs[1] if self.slabs else []
            flag = torch.tensor([len(zl)] + (zl + [0] * 9)[:9], dtype=torch.int64, device=dev)
            lo_, hi_ = flag.clone(), flag.clone()
            dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
            if not (self.slabs and bool((lo_ == hi_).all()) and len(zl) > 2):
                self.slabs = None

    @property
    def order(self):
        """Walk order of the ray-stationary back-projection: spatial neighbours next to each other (``engine.locality_order``)."""
        if self._order is None and hasattr(self.engine, "locality_order") and self.R_local > 0:
            self._order = self.engine.locality_order(self.origins, self.dirs, self.tmax)
        return self._order

    @property
    def forward_order(self):
        """Walk order of the unplanned forward: nearly identical rays (one line of sight a few seconds apart) next to each other."""
        if self._forward_order is None and hasattr(self.engine, "coherent_order") and self.R_local > 0:
            self._forward_order = self.engine.coherent_order(self.origins, self.dirs)
        return self._forward_order

    def _adjoint_order(self):
        return None if (self.plan and self.plan[0]) else self.order          # the node-stationary kernel has its own order

    def _fwd_order(self):
        return None if (self.forward_plan and self.forward_plan[0]) else self.forward_order

    def _touched(self):
        """this rank's back-projection of unit ray weights: non-zero exactly at the nodes its rays reach"""
        ones = torch.ones(self.R_local, dtype=torch.float64, device=self.engine.device)
        return self.engine.adjoint(self.origins, self.dirs, ones, self.tmax, self.Ns, order=self._adjoint_order())

    OVERLAP_SLABS = 4

    def active_index(self):
        """int32 indices of the grid nodes ANY rank's rays reach (identical on every rank; computed once per geometry), sorted --
        with ``exchange="overlap"`` sorted by z-slab of the back-projection plan first (``slab_ranges``: the compact range of every
        slab).  The solvers keep their grid-sized vectors compact over this set."""
        if self._active is None:
            mask = (self._touched().reshape(-1) != 0).to(torch.int32)
            if self.multi:
                dist.all_reduce(mask, op=dist.ReduceOp.MAX)
            idx = mask.nonzero().reshape(-1)
            if self.slabs is not None:
                zl = torch.tensor(self.slabs[1][1:-1], dtype=torch.int64, device=idx.device)
                slab = torch.bucketize(idx % self.engine.shape[2], zl, right=True)       # node level z -> slab (a boundary level: the upper one)
                idx = idx[torch.sort(slab, stable=True)[1]]
                cnt = torch.bincount(slab, minlength=len(self.slabs[1]) - 1).cpu().tolist()
                lo, self.slab_ranges = 0, []
                for c in cnt:
                    self.slab_ranges.append((lo, lo + c))
                    lo += c
            self._active = idx.to(torch.int32).contiguous()
        return self._active

    def overlapped(self):
        """exchange="overlap" is in force: the plan has z-slabs every rank agrees on and the engine back-projects slab by slab."""
        if not (self.multi and self.slabs is not None and self.fused_steps()):
            return False
        # ... and the engine's (single) back-projection plan is still the one made here for THESE tensors: a later plan_adjoint on
        # the same engine (another ShardedRays, say) replaces it, and a slab's unit range would mean nothing to the new plan
        planned = getattr(self.engine, "_planned", None)
        return planned is not None and planned[0] is self.origins and planned[1] is self.dirs

    def backproject_exchange_overlapped(self, ray_step, s_full, s_c, idx):
        """The summed back-projected update in the compact vector ``s_c`` with the exchange HIDDEN behind the back-projection
        (replaces the reference's da.sum(da.stack(...)), inversion/gradient.py:52-54): ``ray_step()`` leaves the ray weights
        in the library; then, slab by slab, back-project the slab's work units, gather (and re-zero) its finished node levels and
        start their all-reduce -- asynchronously, so the next slab's kernel runs while the previous slab's sum is on the links."""
        out = ray_step()
        ul = self.slabs[0]
        pending = []
        rd = self.exchange.reduce_dtype
        for sidx, (lo, hi) in enumerate(self.slab_ranges):
            self.engine.adjoint_planned_weights(self.origins, self.dirs, self.tmax, self.Ns, s_full, unit_range=(ul[sidx], ul[sidx + 1]))
            if hi <= lo:
                continue
            view = s_c[lo:hi]
            self.engine.compact_gather(s_full, idx[lo:hi], out=view, zero=True, want_dot=False)
            buf = view if rd is None or rd == view.dtype else view.to(rd)
            pending.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True), buf, view))
        for work, buf, view in pending:
            work.wait()
            if buf is not view:
                view.copy_(buf)
        return out

    def scalar(self, partial):
        """A device scalar from the per-workgroup partial sums of a fused pass: used as is on one rank (the consuming
        kernel sums it in a fixed order), summed and all-reduced to ONE value when rays are sharded."""
        if not self.multi:
            return partial
        t = partial.sum().reshape(1)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def backproject_differential(self, v, scale, out_full):
        """out_full += (local rays) A^T (scale o v): fused differential weights + back-projection, no exchange."""
        return self.engine.adjoint_differential(self.origins, self.dirs, v, scale, self.Na, self.i0, self.tmax, self.Ns,
                                                out=out_full, order=self._adjoint_order())

    def fused_steps(self):
        """The engine fuses a solver's ray-sized pass with the back-projection's differential-weights pass (RayEngine does; the
        CPU test engine does not)."""
        return hasattr(self.engine, "adjoint_cg_step")

    def backproject_cg_step(self, r, q, an, ad, scale, out_full):
        """r -= (an / ad) q in place; out_full += (local rays) A^T (scale o r); returns the partials of <r, r>."""
        return self.engine.adjoint_cg_step(self.origins, self.dirs, r, q, an, ad, scale, self.Na, self.i0, self.tmax, self.Ns,
                                           out_full, order=self._adjoint_order())

    def backproject_sirt_step(self, tec, scale, weight, out_full, r_out=None):
        """v = dobs - (tec - tec[i0]); out_full += (local rays) A^T (scale o v); returns the partials of sum v^2 weight."""
        return self.engine.adjoint_sirt_step(self.origins, self.dirs, tec, self.dobs, scale, weight, self.Na, self.i0, self.tmax,
                                             self.Ns, out_full, order=self._adjoint_order(), r_out=r_out)

    # -- sharded model update ("reduce-scatter by slab and keep the model update sharded") ------------------------
    def shard_len(self, n):
        """per-rank chunk length of a compact vector of n entries (even, so that chunk starts stay 16-byte aligned)"""
        per = -(-n // self.world)
        return per + (per & 1)

    def reduce_scatter_compact(self, s_c_padded):
        """sum over ranks of a compact vector padded to world * shard_len; returns THIS rank's chunk of the sum"""
        per = s_c_padded.numel() // self.world
        out = torch.empty(per, dtype=s_c_padded.dtype, device=s_c_padded.device)
        rd = self.exchange.reduce_dtype
        if rd is not None and rd != s_c_padded.dtype:
            o2 = torch.empty(per, dtype=rd, device=s_c_padded.device)
            dist.reduce_scatter_tensor(o2, s_c_padded.to(rd), op=dist.ReduceOp.SUM)
            out.copy_(o2)
        else:
            dist.reduce_scatter_tensor(out, s_c_padded, op=dist.ReduceOp.SUM)
        return out

    def all_gather_compact(self, full_padded, chunk):
        dist.all_gather_into_tensor(full_padded, chunk)
        return full_padded

    def reduce_compact_(self, s_c):
        """In-place sum over ranks of a compact (active-set) vector."""
        if self.multi:
            rd = self.exchange.reduce_dtype
            if rd is not None and rd != s_c.dtype:
                buf = s_c.to(rd)
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
                s_c.copy_(buf)
            else:
                dist.all_reduce(s_c, op=dist.ReduceOp.SUM)
        return s_c

    def slice(self, full):
        """[Na,P] (full problem) -> this rank's [Na*P_local] device vector."""
        t = torch.as_tensor(full, dtype=torch.float64)
        return t[:, self.lo:self.hi].reshape(-1).contiguous().to(self.engine.device)

    # -- operators ---------------------------------------------------------------------------------
    def forward_tec(self):
        # walked in the coherent order (engine.coherent_order): the waves of an XCD then work on nearly identical rays at the
        # same time and share lines in the L1 (trilinear 0.224 -> 0.210 ms, float32 blocks 0.172 -> 0.142 ms, tricubic 1.62 ->
        # 1.45 ms at the bench shape); results do not depend on it
        return self.engine.forward(self.origins, self.dirs, self.tmax, self.Ns, order=self._fwd_order())

    def forward(self):
        """differential TEC of the current grid values, local rays: A x = G x - (G x)[i0]."""
        tec = self.forward_tec().view(self.Na, self.P_local)
        return (tec - tec[self.i0:self.i0 + 1]).reshape(-1)

    def adjoint(self, y):
        """A^T y summed over all ranks: differential weights, back-projection, all-reduce."""
        w = y.view(self.Na, self.P_local).clone()
        w[self.i0] -= y.view(self.Na, self.P_local).sum(dim=0)
        g = self.engine.adjoint(self.origins, self.dirs, w.reshape(-1), self.tmax, self.Ns, order=self._adjoint_order())
        return self.exchange.sum_(g)

    def gradient_from_tec(self, tec):
        """Fused residual -> weights -> back-projection (one launch) + all-reduce:
        G^T diff((tec - tec[i0] - dobs)/(CdCt + 1e-15))."""
        g = self.engine.adjoint_residual(self.origins, self.dirs, tec, self.dobs, self.cdct, self.Na, self.i0,
                                         self.tmax, self.Ns, order=self._adjoint_order())
        return self.exchange.sum_(g)

    def dot_rays(self, a, b):
        """<a, b> over ALL rays (local dot + scalar all-reduce)."""
        return float(self.dot_rays_t(a, b))

    def dot_rays_t(self, a, b):
        """Same, as a 0-dim tensor that stays on the device (no host synchronisation: the solvers keep
        their step lengths on the GPU so the CPU can queue launches ahead of the kernels)."""
        return all_reduce_sum_(torch.dot(a, b).reshape(1))[0]

    def gather_rays(self, local):
        """[Na*P_local] on every rank -> [Na,P] on every rank (host), for reporting/tests."""
        loc = local.view(self.Na, self.P_local).cpu()
        if self.world == 1:
            return loc
        parts = [None] * self.world
        dist.all_gather_object(parts, loc)
        return torch.cat(parts, dim=1)
