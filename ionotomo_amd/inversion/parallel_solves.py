"""Independent solves side by side in ONE set of launches: the counterpart of ``num_parallel_solves`` of the reference's pipeline.

The reference solves one time step per task (``inversion/inversion_pipeline.py:131-216``: for every ``time_idx`` its own rays,
``calc_rays(antennas, patches, times[time_idx:time_idx+1], ...)`` = Na x 1 x Nd rays, its own model ``ne_0`` and its own
``iterative_newton_solve``) and runs ``num_parallel_solves`` of them at once as dask threads (``:41-50``).  On the GPU a solve of
2 604 rays is launch-bound (6.9 us per forward whatever its size), and host threads cannot overlap the launches of
several solves (the Python launch path holds the GIL).  Data parallelism does what the dask threads did: the B models are stacked
along x into ONE grid of B nx x ny x nz nodes, solve b's rays are moved into slab b of it, and every launch of the hot path
(forward, dTEC, back-projection, SIRT) then serves all B solves at the rate of a B-times larger batch.

Why this is exact.  Solve b owns the nodes ``[b nx, (b + 1) nx)`` along x; a ray of solve b stays inside its own slab (checked
here on the host: both ends of every ray inside the solve's own x range -- the stacked grid would otherwise hand a stray ray the
NEIGHBOUR's nodes instead of the reference's ``bounds_error``), so no sample ever reads or writes another solve's nodes: forward
and back-projection are block diagonal.  The cell between the last node of slab b and the first of slab b + 1 is never entered
(at most touched at x == last node, with weight 0 on the neighbour's side).  The reference antenna is subtracted per (time,
direction) pair, and pairs are concatenated, not mixed.  SIRT's row and column sums are per ray and per node: B stacked SIRT solves
ARE B separate ones (the one shared number is the cut-off 1e-9 max(col) below which a column counts as empty).  CGLS is different:
its step lengths are global scalars, so ``solvers.cgls`` on the stacked problem is ONE conjugate-gradient solve of the block-diagonal
system -- it converges to the same B solutions, but its iterates are not those of B separate runs; ``StackedSolves.cgls`` keeps one
alpha and one beta per solve and IS B separate runs.

``interp="cubic"``: the tricubic takes its node derivatives from differences across +-2 nodes (csrc/iono_cubic_kernels.h), one-sided
next to a face.  In the stacked grid the nodes next to a seam see the neighbour's values instead, so cells 0, 1 and nx - 3, nx - 2
of a slab interpolate differently from the separate solve; every other cell uses own nodes and the same central formula.  The
check on the rays is therefore two cells stricter along x (x_2 <= x <= x_{nx-3}); then forward and transpose are again those of the
separate solves.  Smoothing with ``C_m`` (``solvers.smooth_grid``) would blur across the seams: not offered.
Straight rays only (a bent ray is not confined by its end points).  Host-side helper: no kernel knows about it.
"""
import numpy as np
import torch

from ..engine import RayEngine


def _is_axis(v):
    try:
        return np.asarray(v, dtype=np.float64).ndim == 1
    except (ValueError, TypeError):
        return False


def solve_share(n_solves, world=None, rank=None):
    """The solves rank ``rank`` of ``world`` stacks (default: this process's torch.distributed rank): contiguous blocks, sizes
    differing by at most one.  Solves are independent, so several GPUs need NO exchange: every rank builds a ``StackedSolves`` of its
    own share (``inversion_pipeline.py:131-216``: the tasks of different time steps only meet in the final list of solutions) and
    sets ``ionotomo_amd.parallel.INDEPENDENT_RANKS = True`` first, so that ``ShardedRays`` / the solvers treat the rank's problem as
    whole instead of as one shard of a problem all ranks share.  (The share is taken from the process group whatever that flag says.)"""
    if world is None or rank is None:
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized()
        world, rank = (dist.get_world_size(), dist.get_rank()) if on else (1, 0)
    if not 0 <= rank < world:
        raise ValueError("solve_share: rank %d of %d" % (rank, world))
    per, extra = divmod(int(n_solves), int(world))
    lo = rank * per + min(rank, extra)
    return range(lo, lo + per + (1 if rank < extra else 0))


class StackedSolves(object):
    """B solves on grids of the same shape and spacing (their origins may differ: every solve's rays are moved by its own offset).

    ``grids``: list of ``(xvec, yvec, zvec)``, one per solve, or ONE tuple + ``count``.
    ``engine``: the ``RayEngine`` holding the stacked grid (``storage``, ``device``, ``interp`` as for any engine).
    ``split_grid`` / ``split_rays`` return VIEWS: block b of a stacked result is solve b's result.
    The stacked grid must stay below 4 GiB (32 solves of 256^3 float64 nodes, 256 of 128^3): see ``allow_general``."""

    def __init__(self, grids, count=None, device=0, storage="f64", interp="linear", rtol=1e-9, allow_general=False):
        if interp not in ("linear", "cubic"):
            raise ValueError("StackedSolves: interp is 'linear' or 'cubic'")
        self.interp = interp
        self.margin = 2 if interp == "cubic" else 0         # cells a ray keeps clear of its slab's x faces (module docstring)
        if len(grids) == 3 and all(_is_axis(v) for v in grids):              # ONE (xvec, yvec, zvec): `count` solves on it
            grids = [tuple(grids)] * int(1 if count is None else count)
        elif count is not None:
            raise ValueError("StackedSolves: count goes with ONE (xvec, yvec, zvec)")
        if not grids:
            raise ValueError("StackedSolves: no grid")
        ax = [[np.asarray(v, dtype=np.float64) for v in g] for g in grids]
        x0, y0, z0 = ax[0]
        self.shape1 = (len(x0), len(y0), len(z0))
        if min(self.shape1) < 2:
            raise ValueError("StackedSolves: every axis needs two nodes")
        self.spacing = tuple(float((v[-1] - v[0]) / (len(v) - 1)) for v in ax[0])
        for b, g in enumerate(ax):
            if tuple(len(v) for v in g) != self.shape1:
                raise ValueError("StackedSolves: grid %d has shape %s, grid 0 %s" % (b, tuple(len(v) for v in g), self.shape1))
            for v, h in zip(g, self.spacing):
                # uniform, and the same spacing as grid 0 (the stacked axis is ONE uniform axis; y and z axes are shared)
                if not np.allclose(np.diff(v), h, rtol=rtol, atol=abs(h) * rtol):
                    raise ValueError("StackedSolves: grid %d is not uniform with the spacing of grid 0" % b)
        self.B = len(ax)
        self.nx = self.shape1[0]
        self.origins = np.array([[g[0][0], g[1][0], g[2][0]] for g in ax])             # first node of every solve
        self.x_hi = np.array([g[0][-1] for g in ax])
        self.base = self.origins[0].copy()
        hx = self.spacing[0]
        # slab b's first node sits at base_x + b nx hx: the move applied to solve b's rays
        self.shift = self.base[None, :] - self.origins
        self.shift[:, 0] += np.arange(self.B) * self.nx * hx
        self.xvec = self.base[0] + hx * np.arange(self.B * self.nx)
        self.xvec[:self.nx] = x0                # (a stack of ONE solve is that solve's grid, node for node)
        self.yvec, self.zvec = y0.copy(), z0.copy()
        # the library's planned / lanes = samples kernels address the grid with 32-bit BYTE offsets (csrc/ionotomo_hip.hip:
        # fast_path_ok); a larger grid falls to the general kernels, several times slower -- several stacks serve it better
        item = 4 if str(storage) in ("f32", "float32") else 8
        if self.B * int(np.prod(self.shape1)) * item >= 2 ** 32 and not allow_general:
            per = 2 ** 32 // (int(np.prod(self.shape1)) * item)
            raise ValueError("StackedSolves: %d solves of %s nodes exceed the 4 GiB the fast kernels address; stack at most %d of them "
                             "(allow_general=True runs the general kernels instead)" % (self.B, self.shape1, max(per - 1, 1)))
        if self.nx < 2 * self.margin + 2:
            raise ValueError("StackedSolves: %d nodes along x leave no cell two cells away from both faces" % self.nx)
        self._device, self._storage, self._engine = device, storage, None
        self.pairs, self._pair_ids = None, {}   # pairs per solve of the last rays() call

    @property
    def engine(self):
        """The ``RayEngine`` on the stacked grid (created on first use: it needs the GPU, the geometry above does not)."""
        if self._engine is None:
            self._engine = RayEngine(self._device, storage=self._storage, interp=self.interp)
            self._engine.set_grid(self.xvec, self.yvec, self.zvec)
        return self._engine

    # ---- geometry -------------------------------------------------------------------------------------------------------------
    def rays(self, origins, directions, tmax):
        """Per-solve ``origins[b]``, ``directions[b]`` of shape [Na, P_b, 3] (P_b = Nt_b x Nd_b pairs) -> the stacked
        ``[Na, sum P_b, 3]`` arrays (float64 numpy).  Raises ``ValueError`` if a ray leaves its solve's x range before ``tmax``
        (``interp="cubic"``: or comes within two cells of its x faces; y and z are checked by the kernels themselves, as for any grid)."""
        if len(origins) != self.B or len(directions) != self.B:
            raise ValueError("StackedSolves.rays: %d solves, got %d / %d ray sets" % (self.B, len(origins), len(directions)))
        oo, dd, self.pairs, self._pair_ids = [], [], [], {}
        Na = None
        for b in range(self.B):
            o = np.asarray(origins[b], dtype=np.float64)
            d = np.asarray(directions[b], dtype=np.float64)
            if o.ndim != 3 or o.shape[2] != 3 or d.shape != o.shape:
                raise ValueError("StackedSolves.rays: solve %d needs [Na, P, 3] origins and directions" % b)
            if Na is None:
                Na = o.shape[0]
            if o.shape[0] != Na:
                raise ValueError("StackedSolves.rays: solve %d has %d antennas, solve 0 %d" % (b, o.shape[0], Na))
            lo, hi = self.origins[b, 0] + self.margin * self.spacing[0], self.x_hi[b] - self.margin * self.spacing[0]
            end = o[..., 0] + float(tmax) * d[..., 0]
            bad = (o[..., 0] < lo) | (o[..., 0] > hi) | (end < lo) | (end > hi)
            if bad.any():
                raise ValueError("StackedSolves.rays: %d ray(s) of solve %d leave its grid along x%s (a value in x_new is out of "
                                 "bounds)" % (int(bad.sum()), b, " by less than two cells" if self.margin else ""))
            oo.append(o + self.shift[b][None, None, :])
            dd.append(d)
            self.pairs.append(o.shape[1])
        return np.concatenate(oo, axis=1), np.concatenate(dd, axis=1)

    # ---- node fields ------------------------------------------------------------------------------------------------------------
    def stack_grids(self, models):
        """B node arrays of shape (nx, ny, nz) (numpy or torch) -> one device tensor (B nx, ny, nz), float64."""
        if len(models) != self.B:
            raise ValueError("StackedSolves.stack_grids: %d solves, got %d models" % (self.B, len(models)))
        eng = self.engine
        out = torch.empty((self.B * self.nx,) + self.shape1[1:], dtype=torch.float64, device=eng.device)
        for b, m in enumerate(models):
            t = m if torch.is_tensor(m) else torch.as_tensor(np.asarray(m, dtype=np.float64))
            if tuple(t.shape) != self.shape1:
                raise ValueError("StackedSolves.stack_grids: model %d has shape %s, not %s" % (b, tuple(t.shape), self.shape1))
            out[b * self.nx:(b + 1) * self.nx].copy_(t)
        return out

    def split_grid(self, g):
        """Views of a stacked node tensor, one (nx, ny, nz) block per solve."""
        g = g.reshape((self.B * self.nx,) + self.shape1[1:])
        return [g[b * self.nx:(b + 1) * self.nx] for b in range(self.B)]

    def split_rays(self, v, Na):
        """Views of a stacked per-ray tensor ([Na x sum P_b] or [Na, sum P_b]) -> B tensors [Na, P_b]."""
        if self.pairs is None:
            raise ValueError("StackedSolves.split_rays: call rays() first")
        v = v.reshape(Na, -1)
        out, lo = [], 0
        for p in self.pairs:
            out.append(v[:, lo:lo + p])
            lo += p
        return out

    def per_solve_sum(self, v, Na):
        """Sum of a stacked per-ray tensor over the rays of every solve -> tensor [B] on ``v``'s device (e.g. the solves' own
        objectives 1/2 sum r^2 / CdCt from the stacked residual: the reference's stop rule is per solve, iterative_newton.py:542-554)."""
        if self.pairs is None:
            raise ValueError("StackedSolves.per_solve_sum: call rays() first")
        per_pair = v.reshape(Na, -1).sum(dim=0)
        return torch.zeros(self.B, dtype=per_pair.dtype, device=v.device).index_add_(0, self.pair_solve(v.device), per_pair)

    def pair_solve(self, device):
        """int64 tensor [sum P_b]: the solve every (time, direction) pair of the stacked layout belongs to (built once per device:
        a host-to-device copy in the middle of a loop would wait for every queued launch)."""
        key = str(device)
        if key not in self._pair_ids:
            ids = np.repeat(np.arange(self.B, dtype=np.int64), self.pairs)
            self._pair_ids[key] = torch.as_tensor(ids).to(device)
        return self._pair_ids[key]

    def cgls(self, problem, x0, n_iter=50, damp=0.0):
        """B SEPARATE CGLS solves in one loop: the arithmetic of ``solvers.cgls`` (min 1/2 ||W^(1/2) (A x - d)||^2 + damp/2 ||x||^2 per
        solve) with one step length alpha_b and one beta_b PER SOLVE, so block b's iterates are those of solve b alone
        (``solvers.cgls`` on the stacked problem shares the two scalars: one conjugate-gradient solve of the block system).
        ``problem``: ``ShardedRays`` on ``self.engine`` and the rays of ``self.rays`` (one rank).  Per iteration ONE forward and ONE
        back-projection launch for all solves; the grid-sized vectors are kept compact over the nodes the rays reach
        (``problem.active_index()``, as in ``solvers.cgls``) and updated by torch passes with the per-solve scalars looked up per
        node / per ray; the search direction is scattered into a buffer the kernels read in place (``bind_values``).
        ``damp`` acts on the reached nodes (the others never move).
        Returns (x, history [n_iter][B] of the solves' objectives 1/2 sum r^2 / CdCt)."""
        if self.pairs is None:
            raise ValueError("StackedSolves.cgls: call rays() first")
        if getattr(problem, "multi", False):
            raise NotImplementedError("StackedSolves.cgls: one rank per stack (share the SOLVES over ranks: solve_share)")
        eng, Na, B, i0 = problem.engine, problem.Na, self.B, problem.i0
        dev = eng.device
        idx = problem.active_index()
        il = idx.long()
        nid = il // int(np.prod(self.shape1))                               # active node -> solve
        rid = self.pair_solve(dev).unsqueeze(0).expand(Na, -1).reshape(-1)  # ray -> solve ([Na, sum P_b] layout)

        def ray_dot(a, b):
            return self.per_solve_sum(a * b, Na)

        # (the active index is sorted: a solve's reached nodes are one contiguous run of the compact vectors -- segment sums, not B
        #  bins hammered by millions of atomic adds)
        runs = torch.bincount(nid, minlength=B)
        if nid.numel() > 1 and not bool((nid[1:] >= nid[:-1]).all()):
            raise ValueError("StackedSolves.cgls: the problem's active index is not sorted by node (exchange=\"overlap\" is for several ranks)")

        def node_dot(a, b):
            return torch.segment_reduce(a * b, "sum", lengths=runs, unsafe=True)

        Wh = torch.rsqrt(problem.cdct.reshape(-1) + 1e-15)
        s_full = torch.zeros(eng.shape, dtype=torch.float64, device=dev)
        x_c = x0.reshape(-1).index_select(0, il).contiguous()

        def normal_residual(r_):                # compact A^T W^(1/2) r - damp x; leaves s_full zero again
            y = (Wh * r_).view(Na, -1)
            w = y.clone()
            w[i0] -= y.sum(dim=0)
            eng.adjoint(problem.origins, problem.dirs, w.reshape(-1), problem.tmax, problem.Ns, out=s_full, order=problem._adjoint_order())
            s_c, _ = eng.compact_gather(s_full, idx, zero=True, want_dot=False)
            return s_c.sub_(x_c, alpha=damp) if damp != 0.0 else s_c

        eng.bind_values(None)
        eng.set_values(x0.reshape(-1).contiguous())
        r = Wh * (problem.dobs.reshape(-1) - problem.forward())
        s = normal_residual(r)
        p = s.clone()
        gamma = node_dot(s, s)
        p_pad, p_full = eng.new_grid_buffer()
        hist = []
        try:
            eng.bind_values(p_pad)
            for _ in range(n_iter):
                hist.append(0.5 * ray_dot(r, r))
                eng.compact_scatter(p_full, idx, p)
                eng.values_changed()
                q = Wh * problem.forward()
                den = ray_dot(q, q)
                if damp != 0.0:
                    den = den + damp * node_dot(p, p)
                # a solve whose residual is already zero keeps alpha = beta = 0 (0 / 0 otherwise)
                alpha = torch.where(den > 0, gamma / den, torch.zeros_like(den))
                x_c.addcmul_(alpha[nid], p)
                r = r - alpha[rid] * q
                s = normal_residual(r)
                gnew = node_dot(s, s)
                beta = torch.where(gamma > 0, gnew / gamma, torch.zeros_like(gamma))
                p = s + beta[nid] * p
                gamma = gnew
        finally:
            eng.bind_values(None)
        x = x0.clone()
        x.view(-1)[il] = x_c
        eng.set_values(x.reshape(-1).contiguous())
        h = torch.stack(hist).cpu().numpy() if hist else np.zeros((0, B))
        return x, h

    def stack_rays(self, per_solve):
        """B per-ray arrays [Na, P_b] (e.g. ``dobs``, ``CdCt``) -> [Na, sum P_b] numpy."""
        return np.concatenate([np.asarray(a, dtype=np.float64) for a in per_solve], axis=1)
