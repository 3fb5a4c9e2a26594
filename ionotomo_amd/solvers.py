"""Tomographic inversion drivers on top of the two hot kernels (forward, adjoint).

The reference has no solver literally called SIRT or CG; the pieces it does
have, and which are reused here, are
  * the objective  S = 1/2 sum (g - dobs)^2 / CdCt           (inversion/iterative_newton.py:32-38,
                                                              inversion/line_search.py:48-49: + 1e-15)
  * the stopping rule: at least 5 iterations; stop when the relative decrease <= factr*eps
    (factr = 1e7), or max|dm| <= pgtol = 1e-2, or iter >= 20   (iterative_newton.py:959-962,993)
  * the linearised exact line search eps = sum(Gdm dd/Cd) / sum(Gdm^2/Cd)   (:542-554)
  * SIRT's row/column-sum normalisation L = diag(sum_v G), C = diag(sum_r G)
                                                              (geometry/oct_trees/Inversion.py:559,564)
  * steepest descent  m <- m - eps (C_m G^T C_d^-1 r + m - m_prior)       (Inversion.py:533)
Every iteration is one forward launch + one adjoint launch (+ one all-reduce of the
back-projected update when rays are sharded over GPUs) + grid-sized vector updates.

``problem`` is a ``parallel.ShardedRays``; grid-sized vectors are torch tensors on its device and
are identical on every rank (they only ever change by all-reduced quantities).
"""
import torch

FACTR, PGTOL, EPS = 1e7, 1e-2, 2.220446049250313e-16


def reference_stop(S_prev, S, max_step, it, max_iter=20, min_iter=5, pgtol=PGTOL):
    """The reference's loop condition (inversion/iterative_newton.py:959-962,993), negated:
    ``while ((S_n - S_np1)/max(|S_n|, |S_np1|, 1) > factr*eps and max|m_n - m_np1| > pgtol and iter < max_iter)
    or iter < 5`` with factr = 1e7, pgtol = 1e-2, max_iter = 20.  ``it`` = updates done so far, ``S_prev`` / ``S`` the
    objective before / after the last update, ``max_step`` the largest change of a model value in it."""
    if it < min_iter:
        return False
    going = ((S_prev - S) / max(abs(S_prev), abs(S), 1.0) > FACTR * EPS) and (max_step > pgtol) and (it < max_iter)
    return not going


def smooth_grid(eng, v, covariance):
    """C_m v on the engine's device (engines without a smoothing kernel -- the CPU test stand-in -- may
    provide ``smooth`` too)."""
    return eng.smooth(v.contiguous(), covariance.kx, covariance.ky, covariance.kz)


def objective(problem, resid):
    """S = 1/2 sum r^2/(CdCt + 1e-15) over all rays."""
    return 0.5 * problem.dot_rays(resid, resid / (problem.cdct + 1e-15))


def _set_x(problem, x):
    problem.engine.set_values(x.reshape(-1))


def _sirt_dense(problem, x0, n_iter=50, relax=1.0, nonneg=False, callback=None, stop=None, pgtol=PGTOL):
    """x_{k+1} = x_k + relax * C A^T L (d - A x_k),  A = differenced ray operator.
    L, C from row / column sums of |A| bounded by the un-differenced sums (keeps rho <= 1).
    ``stop="reference"``: the reference's stopping rule (``reference_stop``; ``n_iter`` is its max_iter) on the
    objective S = 1/2 sum r^2/CdCt and the largest model change per update -- one host read-back per iteration;
    ``stop=None`` runs exactly ``n_iter`` updates without ever synchronising with the host."""
    eng = problem.engine
    x = x0.clone()
    ones = torch.ones(eng.shape, dtype=torch.float64, device=eng.device)
    _set_x(problem, ones)
    rows = problem.forward_tec().view(problem.Na, problem.P_local)          # path lengths
    L = 1.0 / (rows + rows[problem.i0:problem.i0 + 1]).reshape(-1)
    wcol = torch.ones(problem.Na, problem.P_local, dtype=torch.float64, device=eng.device)
    wcol[problem.i0] += problem.Na
    col = parallel_adjoint_raw(problem, wcol.reshape(-1))
    # nodes with (numerically) no ray coverage are left alone: 1/col would turn rounding noise into updates
    C = torch.where(col > 1e-9 * col.max(), 1.0 / col, torch.zeros_like(col))
    hist = []
    Wt = 1.0 / (problem.cdct + 1e-15)
    max_step = None
    for k in range(n_iter + (1 if stop else 0)):
        _set_x(problem, x)
        r = problem.dobs - problem.forward()
        hist.append(0.5 * problem.dot_rays_t(r, r * Wt))        # stays on the device (see cgls)
        if callback:
            callback(k, x, float(hist[-1]))
        if stop and k > 0 and (k >= n_iter or _stop_now(hist, max_step, k, n_iter, pgtol)):
            break
        upd = problem.adjoint(L * r).mul_(C)
        if nonneg:
            xn = torch.clamp(torch.add(x, upd, alpha=relax), min=0)
            upd, x = xn - x, xn
            max_step = upd.abs().max() if stop else None
        else:
            x.add_(upd, alpha=relax)
            max_step = relax * upd.abs().max() if stop else None
    return x, [float(h) for h in torch.stack(hist).cpu()] if hist else []


def _stop_now(hist, max_step, k, max_iter, pgtol):
    """``reference_stop`` on device scalars (one read-back)."""
    vals = torch.stack([hist[-2], hist[-1], max_step.reshape(())]).cpu()
    S_prev, S, step = (float(v) for v in vals)
    return reference_stop(S_prev, S, step, k, max_iter, pgtol=pgtol)


def parallel_adjoint_raw(problem, w):
    """G^T w (no differencing), summed over ranks."""
    from .parallel import all_reduce_sum_
    return all_reduce_sum_(problem.engine.adjoint(problem.origins, problem.dirs, w, problem.tmax, problem.Ns,
                                                  order=problem._adjoint_order()))


def _cgls_dense(problem, x0, n_iter=50, damp=0.0, callback=None, stop=None, pgtol=PGTOL):
    """CGLS on  min 1/2 || W^(1/2) (A x - d) ||^2 + damp/2 ||x||^2,  W = 1/(CdCt + 1e-15).
    All scalars (alpha, beta, objective) stay on the device; the history is read back once at the
    end, so an iteration never waits for the host (``callback`` forces a read-back per iteration).
    ``stop="reference"``: the reference's stopping rule as in ``sirt`` (one read-back per iteration)."""
    eng = problem.engine
    x = x0.clone()
    Wh = torch.rsqrt(problem.cdct + 1e-15)

    def normal_residual(r_):                    # A^T W^(1/2) r - damp x   (grid-sized; one fused pass when damp = 0)
        s_ = problem.adjoint(Wh * r_)
        return s_.sub_(x, alpha=damp) if damp != 0.0 else s_

    _set_x(problem, x)
    r = Wh * (problem.dobs - problem.forward())
    s = normal_residual(r)
    p = s.clone()
    gamma = torch.dot(s.reshape(-1), s.reshape(-1))
    hist = []
    max_step = None
    for k in range(n_iter + (1 if stop else 0)):
        hist.append(0.5 * problem.dot_rays_t(r, r))
        if callback:
            callback(k, x, float(hist[-1]))
        if stop and k > 0 and (k >= n_iter or _stop_now(hist, max_step, k, n_iter, pgtol)):
            break
        _set_x(problem, p)
        q = Wh * problem.forward()
        qq = problem.dot_rays_t(q, q)
        den = qq + damp * torch.dot(p.reshape(-1), p.reshape(-1)) if damp != 0.0 else qq
        # alpha = gamma / den and beta = gnew / gamma are 0-dim device tensors: one fused pass per update
        if stop:
            max_step = p.abs().max() * (gamma / den).abs()            # max |alpha p|
        eng.axpby_(x, p, a_num=gamma, a_den=den)                      # x += alpha p
        eng.axpby_(r, q, a_num=gamma, a_den=den, a_sign=-1.0)         # r -= alpha q
        s = normal_residual(r)
        gnew = torch.dot(s.reshape(-1), s.reshape(-1))
        eng.axpby_(p, s, b_num=gnew, b_den=gamma)                     # p = s + beta p
        gamma = gnew
    return x, [float(h) for h in torch.stack(hist).cpu()] if hist else []


def _history(hist, reduce_over=None):
    """list of device scalars (1 element or per-workgroup partials) -> list of floats, one read-back (and, for local
    sums whose all-reduce was deferred, ONE stacked all-reduce over ``reduce_over``'s ranks)"""
    if not hist:
        return []
    t = torch.stack([h.sum() for h in hist])
    if reduce_over is not None and reduce_over.multi:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
    return [0.5 * float(v) for v in t.cpu()]


def _fused_ok(problem, callback):
    eng = problem.engine
    return (callback is None and hasattr(eng, "compact_gather") and problem.dobs is not None and
            getattr(eng, "storage", "f64") in ("f64", "float64"))


def _run_iterations(n, body, keep):
    """One iteration captured into a hipGraph (through torch's stream capture: the library launches on the stream torch is
    capturing) and replayed n - 1 times.  ``body()`` must be self-similar -- every buffer it reads or writes at a fixed
    address -- and returns the device partial sums whose history is wanted; ``keep(k, partials)`` files them (one small
    copy per iteration, outside the graph).  Iteration 0 runs eagerly, so every lazily created library buffer exists before
    the capture.  Same kernels, same order.  MEASURED, and off by default because it does not pay on this stack
    (profiles/tools/time_graph_solvers.py, 200 iterations): CGLS 63.5 against 68 us per iteration at a single-timestep
    size (2 604 rays, 128^3) but SIRT 54.6 against 43.8 us, and 0.596 against 0.582 ms (CGLS) at the bench shape -- the
    eager loop never waits for the host (launches are queued ahead of the GPU), so a graph has no host time to remove;
    what bounds the small case is the ~8 us dispatch latency between dependent kernels, which graph nodes pay too."""
    keep(0, body())
    if n < 2:
        return None
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        part = body()
    for k in range(1, n):
        g.replay()
        keep(k, part)
    return g              # owns the memory of every tensor created during capture: keep it until the results are copied out


def _small(problem, small_pass):
    """One coherence window on one rank: the ray-sized passes of an iteration run as ONE launch (engine.small_ray_pass)."""
    eng = problem.engine
    return bool(small_pass) and not problem.multi and hasattr(eng, "small_ray_pass") and 0 < problem.R_local <= eng.SMALL_RAYS


def _backproject_weights(problem, w, s_full):
    eng = problem.engine
    return eng.adjoint(problem.origins, problem.dirs, w, problem.tmax, problem.Ns, out=s_full, order=problem._adjoint_order())


def _check_plans(eng):
    """A solve ends with a read-back anyway: raise if any planned launch in it was handed rays its plan was not made for (a planned
    tensor edited in place: engine.RayEngine.check_plans; engines without plans -- the CPU test engine -- have nothing to check)."""
    chk = getattr(eng, "check_plans", None)
    if chk is not None:
        chk()


def sirt(problem, x0, n_iter=50, relax=1.0, nonneg=False, callback=None, stop=None, pgtol=PGTOL, graph=False, small_pass=False):
    """x_{k+1} = x_k + relax * C A^T L (d - A x_k),  A = differenced ray operator; L, C from row / column sums of |A|
    bounded by the un-differenced sums (keeps rho <= 1; geometry/oct_trees/Inversion.py:559,564).
    One iteration = forward launch, ONE pass over the rays (residual + objective), fused differential back-projection,
    ONE pass over the active nodes (update + re-zero + refresh of the grid the forward reads): 4 launches -- forward, ray pass
    (k_rays_step: residual, objective, differential weights), back-projection, update
    (``_sirt_dense`` is the same arithmetic with full-grid torch vectors; used when a ``callback`` wants the iterate).
    ``stop="reference"``: the reference's stopping rule (``reference_stop``; ``n_iter`` is its max_iter) on the
    objective S = 1/2 sum r^2/CdCt and the largest model change per update -- one host read-back per iteration;
    ``stop=None`` runs exactly ``n_iter`` updates without ever synchronising with the host.
    The row / column sums are those of A itself: a bound on the iteration only while its weights are non-negative (trilinear).
    The tricubic Hermite basis has negative lobes, so with ``interp="cubic"`` SIRT carries no contraction guarantee (it can
    diverge after a few sweeps): use ``relax`` < 1/3 (|weights| sum to < 1.44 per axis) or CGLS there.
    ``graph=True`` (one rank, no ``stop``): iterations 1 .. n-1 replayed from one hipGraph (``_run_iterations``).
    ``small_pass`` (off by default): at most 32 768 rays on one rank -- residual, objective and the back-projection's ray weights in
    one launch of one workgroup (3 launches per iteration instead of 5).  Measured at config 2: 44.9 against 42.9 us per iteration --
    the single workgroup's three dependent phases take as long as the two small kernels and the boundary they replace (for CGLS,
    where it replaces three kernels and two dot-product hand-overs, it pays: ``cgls``)."""
    if not _fused_ok(problem, callback):
        return _sirt_dense(problem, x0, n_iter, relax, nonneg, callback, stop, pgtol)
    eng = problem.engine
    idx = problem.active_index()
    il = idx.long()
    ones = torch.ones(eng.shape, dtype=torch.float64, device=eng.device)
    eng.bind_values(None)
    _set_x(problem, ones)
    rows = problem.forward_tec().view(problem.Na, problem.P_local)          # path lengths
    L = (1.0 / (rows + rows[problem.i0:problem.i0 + 1])).reshape(-1).contiguous()
    wcol = torch.ones(problem.Na, problem.P_local, dtype=torch.float64, device=eng.device)
    wcol[problem.i0] += problem.Na
    col = parallel_adjoint_raw(problem, wcol.reshape(-1)).reshape(-1)
    C = torch.where(col > 1e-9 * col.max(), 1.0 / col, torch.zeros_like(col))
    C_c = C.index_select(0, il).contiguous()
    del ones, col, C
    Wt = (1.0 / (problem.cdct + 1e-15)).contiguous()
    x_pad, x_full = eng.new_grid_buffer()
    x_full.copy_(x0)
    x_c = x0.reshape(-1).index_select(0, il).contiguous()
    s_full = torch.zeros(eng.shape, dtype=torch.float64, device=eng.device)
    multi, sharded = problem.multi, problem.multi and problem.exchange.sharded
    n = idx.numel()
    if sharded:
        # reduce-scatter by slab, sharded update, all-gather: every rank sums and updates only its contiguous
        # chunk of the active nodes; the collectives move the same bytes as the all-reduce, the update work is 1 / world
        per = problem.shard_len(n)
        lo = problem.rank * per
        s_c = torch.zeros(per * problem.world, dtype=torch.float64, device=eng.device)
        xg = torch.zeros_like(s_c)
        xg[:n] = x_c
        C_loc = torch.zeros(per, dtype=torch.float64, device=eng.device)
        m = max(0, min(per, n - lo))
        C_loc[:m] = C_c[lo:lo + m]
        x_loc = xg[lo:lo + per].clone()
    else:
        s_c = torch.empty_like(x_c) if multi else None
    eng.bind_values(x_pad)
    hist, r, step = [], None, None
    small = _small(problem, small_pass)
    # residual + objective + the back-projection's ray weights in ONE ray pass (4 launches per iteration instead of 5); with a
    # stopping rule the objective is read back BEFORE the back-projection is started, so the passes stay apart there
    fused = not small and not stop and problem.fused_steps()
    overlap = fused and multi and not sharded and problem.overlapped()       # exchange="overlap": the sum hidden behind the back-projection
    wbuf = torch.empty(problem.R_local, dtype=torch.float64, device=eng.device) if small else None
    defer = multi and not stop                  # objective history: ONE stacked all-reduce at the end instead of one per iteration
    held = None
    try:
        if graph and not multi and not stop and n_iter > 0:
            r = torch.empty(problem.R_local, dtype=torch.float64, device=eng.device)
            hbuf = []

            def body():
                eng.values_changed()
                if small:
                    S2, _, _ = eng.small_ray_pass(1, problem.forward_tec(), L, r, problem.Na, problem.i0, dobs=problem.dobs, weight=Wt, w=wbuf)
                    _backproject_weights(problem, wbuf, s_full)
                elif fused:
                    S2 = problem.backproject_sirt_step(problem.forward_tec(), L, Wt, s_full)
                else:
                    _, S2 = eng.rays_combine(problem.forward_tec(), problem.Na, problem.i0, -1.0, 1.0, dobs=problem.dobs, s2=Wt, out=r)
                    problem.backproject_differential(r, L, s_full)
                eng.compact_sirt_update(x_c, C_c, s_full, idx, x_full, relax, nonneg, want_max=False)
                return S2
            held = _run_iterations(n_iter, body, lambda k, part: hbuf.append(part.clone()))
            hist = hbuf
            n_iter = 0
        for k in range(n_iter + (1 if stop else 0)):
            eng.values_changed()
            tec = problem.forward_tec()
            if small:
                if r is None:
                    r = torch.empty(problem.R_local, dtype=torch.float64, device=eng.device)
                S2, _, _ = eng.small_ray_pass(1, tec, L, r, problem.Na, problem.i0, dobs=problem.dobs, weight=Wt, w=wbuf)   # r = d - A x, weights
            elif overlap:
                # ray pass, then back-projection, gather and all-reduce slab by slab: the exchange runs behind the next slab's kernel
                S2 = problem.backproject_exchange_overlapped(lambda: problem.backproject_sirt_step(tec, L, Wt, None), s_full, s_c, idx)
            elif fused:
                S2 = problem.backproject_sirt_step(tec, L, Wt, s_full)          # r = d - A x, S, s_full += A^T (L r): one ray pass
            else:
                r, S2 = eng.rays_combine(tec, problem.Na, problem.i0, -1.0, 1.0, dobs=problem.dobs, s2=Wt, out=r)   # r = d - A x
            hist.append(S2.sum().reshape(1) if defer else problem.scalar(S2))
            if stop and k > 0 and (k >= n_iter or _stop_fused(hist, step, k, n_iter, pgtol)):
                break
            if small:
                _backproject_weights(problem, wbuf, s_full)
            elif not fused:
                problem.backproject_differential(r, L, s_full)
            if sharded:
                eng.compact_gather(s_full, idx, out=s_c[:n], zero=True, want_dot=False)
                s_loc = problem.reduce_scatter_compact(s_c)
                x_new = torch.addcmul(x_loc, C_loc, s_loc, value=relax)
                if nonneg:
                    x_new.clamp_(min=0)
                if stop:
                    step = (x_new - x_loc).abs().max().reshape(1)
                    torch.distributed.all_reduce(step, op=torch.distributed.ReduceOp.MAX)
                x_loc = x_new
                problem.all_gather_compact(xg, x_loc)
                eng.compact_scatter(x_full, idx, xg[:n])
                continue
            if multi and not overlap:            # sum the partial updates over ranks on the active nodes only
                eng.compact_gather(s_full, idx, out=s_c, zero=True, want_dot=False)
                problem.reduce_compact_(s_c)
                eng.compact_scatter(s_full, idx, s_c)
            elif multi:                          # (summed slab by slab behind the back-projection, above)
                eng.compact_scatter(s_full, idx, s_c)
            step = eng.compact_sirt_update(x_c, C_c, s_full, idx, x_full, relax, nonneg, want_max=bool(stop))
    finally:
        eng.bind_values(None)
    out = x_full.clone(), _history(hist, problem if defer else None)
    del held
    _check_plans(eng)
    return out


def _stop_fused(hist, step_partial, k, max_iter, pgtol):
    vals = torch.stack([hist[-2].sum() * 0.5, hist[-1].sum() * 0.5, step_partial.max()]).cpu()
    return reference_stop(float(vals[0]), float(vals[1]), float(vals[2]), k, max_iter, pgtol=pgtol)


def cgls(problem, x0, n_iter=50, damp=0.0, callback=None, stop=None, pgtol=PGTOL, graph=False, small_pass=True):
    """CGLS on  min 1/2 || W^(1/2) (A x - d) ||^2,  W = 1/(CdCt + 1e-15).
    One iteration = forward launch (reads the search direction IN PLACE), one pass over the rays (q = W^1/2 A p and
    <q, q>), one (r -= alpha q and <r, r>), the fused differential back-projection, one gather over the active nodes
    (s and <s, s>, re-zeroing the back-projection buffer) and one update (x += alpha p, p = s + beta p, refresh of the
    grid the forward reads): 6 launches (the second ray pass also forms the back-projection's differential weights: k_rays_step),
    all scalars on the device as per-workgroup partial sums.
    ``damp`` > 0 (Tikhonov term: the gradient is then non-zero off the ray fan) or a ``callback`` use the dense-vector
    form ``_cgls_dense``.  ``stop="reference"``: the reference's stopping rule as in ``sirt``.  ``graph=True`` (one rank): iterations
    1 .. n-1 replayed from one hipGraph (``_run_iterations``).  ``small_pass``: at most 32 768 rays on one rank -- the three ray-sized
    passes (and the back-projection's ray weights) in one launch: 4 launches per iteration instead of 7 (config 2: 61 -> 52 us)."""
    if damp != 0.0 or stop or not _fused_ok(problem, callback):
        return _cgls_dense(problem, x0, n_iter, damp, callback, stop, pgtol)
    eng = problem.engine
    idx = problem.active_index()
    il = idx.long()
    Wh = torch.rsqrt(problem.cdct + 1e-15).contiguous()
    multi = problem.multi
    eng.bind_values(None)
    _set_x(problem, x0)
    r, rr = eng.rays_combine(problem.forward_tec(), problem.Na, problem.i0, -1.0, 1.0, dobs=problem.dobs, s1=Wh)   # W^1/2 (d - A x0)
    s_full = torch.zeros(eng.shape, dtype=torch.float64, device=eng.device)
    problem.backproject_differential(r, Wh, s_full)
    s_c, gamma = eng.compact_gather(s_full, idx, zero=True, want_dot=not multi)
    if multi:
        problem.reduce_compact_(s_c)
        gamma = eng.axpby_dot_(s_c, s_c, a_sign=0.0)                    # y = 0 x + 1 y: just the dot
    p_c = s_c.clone()
    x_c = x0.reshape(-1).index_select(0, il).contiguous()
    p_pad, p_full = eng.new_grid_buffer()
    eng.compact_scatter(p_full, idx, p_c)
    eng.bind_values(p_pad)
    n = idx.numel()
    sharded = multi and problem.exchange.sharded
    if sharded:                                   # see sirt: reduce-scatter, update of this rank's chunk, all-gather
        per = problem.shard_len(n)
        lo = problem.rank * per
        sg = torch.zeros(per * problem.world, dtype=torch.float64, device=eng.device)
        pg = torch.zeros_like(sg)
        pg[:n] = p_c
        xg = torch.zeros_like(sg)
        xg[:n] = x_c
        x_loc, p_loc = xg[lo:lo + per].clone(), pg[lo:lo + per].clone()
    hist, q = [rr.sum().reshape(1) if multi else rr], None
    small = _small(problem, small_pass)
    fused = not small and problem.fused_steps()      # r -= alpha q, <r, r> and the back-projection's ray weights in ONE ray pass (6 launches, not 7)
    overlap = fused and multi and not (multi and problem.exchange.sharded) and problem.overlapped()
    wbuf = torch.empty(problem.R_local, dtype=torch.float64, device=eng.device) if small else None
    held = None
    try:
        if graph and not multi and n_iter > 0:
            q = torch.empty(problem.R_local, dtype=torch.float64, device=eng.device)
            gam = gamma.clone()                                   # <s, s> of the previous iteration, at a fixed address
            hbuf = []

            def body():
                eng.values_changed()
                if small:
                    qq, rr, _ = eng.small_ray_pass(0, problem.forward_tec(), Wh, r, problem.Na, problem.i0, q=q, gamma=gam, w=wbuf)
                    _backproject_weights(problem, wbuf, s_full)
                elif fused:
                    _, qq = eng.rays_combine(problem.forward_tec(), problem.Na, problem.i0, 1.0, 0.0, s1=Wh, out=q)
                    rr = problem.backproject_cg_step(r, q, gam, qq, Wh, s_full)
                else:
                    _, qq = eng.rays_combine(problem.forward_tec(), problem.Na, problem.i0, 1.0, 0.0, s1=Wh, out=q)
                    rr = eng.axpby_dot_(r, q, an=gam, ad=qq, a_sign=-1.0)
                    problem.backproject_differential(r, Wh, s_full)
                _, gnew = eng.compact_gather(s_full, idx, out=s_c, zero=True, want_dot=True)
                eng.compact_cg_update(x_c, p_c, s_c, idx, p_full, gam, qq, gnew, gam)
                gam.copy_(gnew)
                return rr
            held = _run_iterations(n_iter, body, lambda k, part: hbuf.append(part.clone()))
            hist += hbuf[:n_iter - 1]                             # (the eager loop does not record the last residual either)
            n_iter = 0
        for k in range(n_iter):
            eng.values_changed()
            if small:          # W^1/2 A p, r -= alpha q and the back-projection's ray weights in one launch
                if q is None:
                    q = torch.empty(problem.R_local, dtype=torch.float64, device=eng.device)
                qq, rr, _ = eng.small_ray_pass(0, problem.forward_tec(), Wh, r, problem.Na, problem.i0, q=q, gamma=gamma, w=wbuf)
            else:
                q, qq = eng.rays_combine(problem.forward_tec(), problem.Na, problem.i0, 1.0, 0.0, s1=Wh, out=q)      # W^1/2 A p
                qq = problem.scalar(qq)
                if overlap:
                    rr = problem.backproject_exchange_overlapped(lambda: problem.backproject_cg_step(r, q, gamma, qq, Wh, None), s_full, s_c, idx)
                elif fused:
                    rr = problem.backproject_cg_step(r, q, gamma, qq, Wh, s_full)   # r -= alpha q, <r, r>, s_full += A^T (W^1/2 r)
                else:
                    rr = eng.axpby_dot_(r, q, an=gamma, ad=qq, a_sign=-1.0)         # r -= alpha q
            if k + 1 < n_iter:
                hist.append(rr.sum().reshape(1) if multi else rr)           # sharded rays: summed over ranks once, at the end
            if small:
                _backproject_weights(problem, wbuf, s_full)
            elif not fused:
                problem.backproject_differential(r, Wh, s_full)
            if sharded:
                eng.compact_gather(s_full, idx, out=sg[:n], zero=True, want_dot=False)
                s_loc = problem.reduce_scatter_compact(sg)
                gnew = torch.dot(s_loc, s_loc).reshape(1)
                torch.distributed.all_reduce(gnew, op=torch.distributed.ReduceOp.SUM)
                x_loc.add_(p_loc * (gamma.sum() / qq.sum()))
                p_loc = torch.add(s_loc, p_loc * (gnew.sum() / gamma.sum()))
                problem.all_gather_compact(pg, p_loc)
                eng.compact_scatter(p_full, idx, pg[:n])
                gamma = gnew
                continue
            if overlap:                          # (s_c was gathered and summed slab by slab behind the back-projection)
                gnew = eng.axpby_dot_(s_c, s_c, a_sign=0.0)
            else:
                _, gnew = eng.compact_gather(s_full, idx, out=s_c, zero=True, want_dot=not multi)
                if multi:
                    problem.reduce_compact_(s_c)
                    gnew = eng.axpby_dot_(s_c, s_c, a_sign=0.0)
            eng.compact_cg_update(x_c, p_c, s_c, idx, p_full, gamma, qq, gnew, gamma)
            gamma = gnew
        if sharded:
            problem.all_gather_compact(xg, x_loc)
            x_c = xg[:n]
    finally:
        eng.bind_values(None)
    x = x0.clone()
    x.view(-1).index_copy_(0, il, x_c)
    out = x, _history(hist, problem if multi else None)
    del held
    _check_plans(eng)
    return out


def steepest_descent_log_model(problem, m0, K_scale, m_prior=None, prior_weight=0.0, max_iter=20, min_iter=5,
                               callback=None, covariance=None):
    """Nonlinear inversion in the log-model m (ne = K_scale exp(m) at the nodes) with the
    reference's objective, update, line search and stopping rule (module docstring).
    ``covariance``: an ``ionosphere.covariance.Covariance``; the data gradient is then pre-multiplied
    by C_m (its separable smoothing stencil), giving the reference's regularised direction
    C_m G^T C_d^-1 r (+ m - m_prior)."""
    eng = problem.engine
    m = m0.clone()
    hist = []
    Wt = 1.0 / (problem.cdct + 1e-15)
    for k in range(max_iter):
        eng.set_log_model(m.reshape(-1), K_scale)
        tec = problem.forward_tec()
        t2 = tec.view(problem.Na, problem.P_local)
        resid = (t2 - t2[problem.i0:problem.i0 + 1]).reshape(-1) - problem.dobs
        rw = resid * Wt
        S = 0.5 * problem.dot_rays(resid, rw)                        # host read-back 1 of 2: the stopping rule needs it
        hist.append(S)
        if callback:
            callback(k, m, S)
        if k >= min_iter and len(hist) > 1 and (hist[-2] - S) <= FACTR * EPS * max(abs(hist[-2]), abs(S), 1.0):
            break
        ne = torch.exp(m).mul_(K_scale)
        dm = problem.gradient_from_tec(tec).mul_(ne)                 # d S / d m  (exp at nodes => node-wise product)
        if covariance is not None:
            dm = smooth_grid(eng, dm, covariance)
        if m_prior is not None and prior_weight > 0:
            dm = dm + prior_weight * (m - m_prior)
        # linearised exact line search along -dm: d(A ne)/d eps = -A (ne * dm); the step length stays on the device
        _set_x(problem, ne.mul_(dm))
        Gdm = problem.forward()
        eps = problem.dot_rays_t(Gdm, rw) / problem.dot_rays_t(Gdm, Gdm * Wt).clamp_min(1e-300)
        m.addcmul_(dm, eps, value=-1.0)                              # m -= eps dm
        if k >= min_iter and float(eps.abs() * torch.linalg.vector_norm(dm, ord=float("inf"))) <= PGTOL:   # read-back 2
            break
    _check_plans(problem.engine)
    return m, hist


def steepest_descent_phase(eng, origins, dirs, Na, Nt, Nd, tmax, Ns, freqs, clock, const, dobs, CdCt, mu0, K=1e11, i0=0,
                           max_iter=20, min_iter=5, covariance=None, order=None, callback=None):
    """Inversion of the reference's REAL observable -- the phase g[Na,Nt,Nd,Nf] of inversion/iterative_newton.py:86-127 --
    for the log-model mu (ne = K exp(mu)) on one GPU, without ever materialising rays[Na,Nt,Nd,4,Ns]:
      S = 1/2 sum (g - dobs)^2 / CdCt            (iterative_newton.py:32-38)
      dm = [C_m] dS/dmu                          (adjoint: one gather + scatter traversal per 8 frequencies)
      linearised exact line search eps = <J dm, dd/Cd> / <J dm, J dm/Cd>  (:542-554), J dm by a forward difference of g
      stopping rule of the reference              (:959-962,993).
    All arguments are device tensors except ``freqs`` (host).  Returns (mu, [S_0, S_1, ...])."""
    mu = mu0.clone()
    hist = []
    W = 1.0 / CdCt
    fd = 1e-4
    if hasattr(eng, "plan_adjoint"):           # node-stationary transpose for every iteration (geometry only, speed only)
        eng.plan_adjoint(origins, dirs, tmax, Ns)
    if hasattr(eng, "plan_forward"):           # ... and the bundle plan of the forward (windows in LDS): same two tensors
        eng.plan_forward(origins, dirs, tmax, Ns)
    for k in range(max_iter + 1):
        eng.set_log_model(mu.reshape(-1), K)
        g = eng.forward_phase(origins, dirs, Na, Nt, Nd, tmax, Ns, freqs, clock, const, i0)
        dd = g - dobs
        S = 0.5 * float((dd * dd * W).sum())
        hist.append(S)
        if callback:
            callback(k, mu, S)
        if k > 0 and reference_stop(hist[-2], S, step_max, k, max_iter, min_iter):
            break
        if k == max_iter:
            break
        dm = eng.adjoint_phase(origins, dirs, (dd * W).reshape(Na, Nt * Nd, -1).contiguous(), Na, tmax, Ns, freqs, i0, order=order)
        if covariance is not None:
            dm = smooth_grid(eng, dm, covariance)
        scale = float(dm.abs().max())
        if scale == 0.0:
            break
        # J dm: directional derivative of g along -dm (forward difference with a step that changes mu by <= fd)
        t = fd / scale
        eng.set_log_model((mu - t * dm).reshape(-1), K)
        Jdm = (eng.forward_phase(origins, dirs, Na, Nt, Nd, tmax, Ns, freqs, clock, const, i0) - g) / t      # = -J dm
        eps = -float((Jdm * dd * W).sum()) / max(float((Jdm * Jdm * W).sum()), 1e-300)
        step = eps * dm
        step_max = float(step.abs().max())
        mu -= step
    _check_plans(eng)
    return mu, hist
