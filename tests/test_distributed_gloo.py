"""world_size-2, 3 and 8 gloo runs of the sharded path on CPU: partition by (time,direction)
pair, forward without collective, adjoint + all-reduce, CGLS/SIRT iterates identical to one rank.  The world-8 cases shard
config 4's layout (Na antennas x (Nt x Nd) pairs, pair blocks per rank) the way the 8-GPU SCALE run will, with every exchange
mode."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, exchange="dense", engine="oracle", size=None, interp="linear", deterministic=False):
    """Executed by every rank (and with world == 1 in the parent for the reference result).
    ``engine="oracle"``: the CPU stand-in (this file's tests); ``engine="hip"``: the product's RayEngine on GPU 0
    (tests/test_gpu_configs.py runs the same function in two fresh processes sharing the card)."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    from ionotomo_amd import parallel, solvers
    from problems import small_problem
    pb = small_problem(**(size or dict(na=4, nd=5, nt=2, n=12, Ns=13)))
    w = pb["w"]
    rng = np.random.default_rng(1)
    d = rng.normal(size=(pb["na"], pb["P"])) * 0.01
    cd = np.full((pb["na"], pb["P"]), 1e-4)
    if engine == "hip":
        from ionotomo_amd.engine import RayEngine
        eng = RayEngine(0, interp=interp)
        eng.set_grid(w["xvec"], w["yvec"], w["zvec"])
        if deterministic:            # fixed-point back-projection: run-to-run identical bits, so that two runs can be compared exactly
            eng.set_deterministic(True)
    else:
        from cpu_engine import OracleEngine
        eng = OracleEngine(w["xvec"], w["yvec"], w["zvec"])
    dev = eng.device
    host = lambda t: t.detach().cpu().numpy()
    prob = parallel.ShardedRays(eng, pb["o"], pb["d"], pb["tmax"], pb["Ns"], dobs=d, cdct=cd, i0=pb["i0"], exchange=exchange)
    x = torch.from_numpy(pb["x_true"].copy()).to(dev)
    eng.set_values(x)
    fwd = prob.gather_rays(prob.forward()).numpy()
    y_full = torch.from_numpy(rng.normal(size=(pb["na"], pb["P"])))
    adj = host(prob.adjoint(prob.slice(y_full)))
    xc, hc = solvers.cgls(prob, torch.from_numpy(pb["x0"].copy()).to(dev), n_iter=4)
    xs, hs = solvers.sirt(prob, torch.from_numpy(pb["x0"].copy()).to(dev), n_iter=3)
    overlapped = bool(getattr(prob, "overlapped", lambda: False)())      # (while the engine's plan is still prob's)
    # float32 on the links (compact plan shared with the float64 exchange above)
    p32 = parallel.ShardedRays(eng, pb["o"], pb["d"], pb["tmax"], pb["Ns"], i0=pb["i0"], exchange=exchange,
                               reduce_dtype=torch.float32)
    adj32 = host(p32.adjoint(p32.slice(y_full)))
    # p32 has replaced the engine's single back-projection plan: prob must notice (no slab pipeline on a foreign plan: ADVICE r4)
    # and give the same iterates through the compact exchange (the fixed-point mode serves planned back-projections only: skipped there)
    xs2 = xs if deterministic else solvers.sirt(prob, torch.from_numpy(pb["x0"].copy()).to(dev), n_iter=3)[0]
    return dict(fwd=fwd, adj=adj, xc=host(xc), hc=np.array(hc), xs=host(xs), hs=np.array(hs),
                block=(prob.lo, prob.hi), adj32=adj32, active=prob.exchange.fraction,
                compact=prob.exchange.index is not None, P=pb["P"],
                overlapped=overlapped, overlapped_after_replan=bool(getattr(prob, "overlapped", lambda: False)()),
                xs2=host(xs2), nslab=len(prob.slab_ranges or []))


def get_or_fail(q, procs, timeout):
    """The next result from the workers' queue -- or an assertion as soon as a worker has died without sending one (a plain
    ``q.get(timeout=...)`` sits out the whole timeout, which a GPU box takes for a hang)."""
    import queue
    import time
    t0 = time.time()
    while True:
        try:
            return q.get(timeout=2.0)
        except queue.Empty:
            dead = [p.exitcode for p in procs if not p.is_alive() and p.exitcode not in (0, None)]
            assert not dead, "a worker exited with %s before sending its result" % dead
            assert time.time() - t0 < timeout, "no result from the workers within %d s" % timeout


def _worker(rank, world, port, q, exchange, engine="oracle", size=None, interp="linear", backend="gloo", force=False, deterministic=False):
    """``backend="nccl"`` + ``force``: ONE rank on torch's nccl backend (= RCCL) with parallel.FORCE_COLLECTIVES -- every collective of
    the sharded paths is issued although a one-rank sum is the identity (tests/test_gpu_configs.py)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    if force:
        sys.path.insert(0, os.path.dirname(HERE))
        from ionotomo_amd import parallel
        parallel.FORCE_COLLECTIVES = True
    out = _run(world, exchange, engine, size, interp, deterministic)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


SIZE8 = dict(na=5, nd=6, nt=4, n=12, Ns=13)          # 24 (time, direction) pairs: three per rank at world 8


@pytest.mark.parametrize("world,exchange,size", [(2, "dense", None), (2, "compact", None), (3, "auto", None), (2, "sharded", None),
                                                 (3, "sharded", None), (8, "dense", SIZE8), (8, "compact", SIZE8),
                                                 (8, "sharded", SIZE8), (8, "auto", SIZE8)])
def test_sharded_path_matches_single_rank(world, exchange, size):
    """``exchange``: the gradient all-reduce over the whole grid, or only over the nodes some ray touches."""
    ref = _run(1, size=size)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, exchange, "oracle", size)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(get_or_fail(q, procs, 400) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    blocks = [res[r]["block"] for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == ref["P"] and all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
    assert all(b[1] > b[0] for b in blocks)                           # every rank holds pairs
    for r in range(world):
        assert np.allclose(res[r]["fwd"], ref["fwd"], rtol=1e-13, atol=1e-15)
        assert np.allclose(res[r]["adj"], ref["adj"], rtol=1e-11, atol=1e-14)
        assert np.allclose(res[r]["hc"], ref["hc"], rtol=1e-9)
        assert np.allclose(res[r]["xc"], ref["xc"], rtol=1e-8, atol=1e-12)
        assert np.allclose(res[r]["hs"], ref["hs"], rtol=1e-10)
        assert np.allclose(res[r]["xs"], ref["xs"], rtol=1e-10, atol=1e-14)
        assert np.array_equal(res[r]["xc"], res[0]["xc"])        # replicas stay bit-identical across ranks
        assert np.allclose(res[r]["adj32"], ref["adj"], rtol=0, atol=3e-7 * np.abs(ref["adj"]).max())
        assert not np.array_equal(res[r]["adj32"], ref["adj"])
        assert res[r]["compact"] == (exchange in ("compact", "sharded") or (exchange == "auto" and res[r]["active"] < 0.6))
        if exchange != "dense":
            assert 0.0 < res[r]["active"] < 1.0 and res[r]["active"] == res[0]["active"]


def _independent_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, HERE)
        sys.path.insert(0, os.path.dirname(HERE))
        from ionotomo_amd import parallel, solvers
        from ionotomo_amd.inversion.parallel_solves import solve_share
        from problems import small_problem
        from cpu_engine import OracleEngine
        parallel.INDEPENDENT_RANKS = True
        mine = list(solve_share(5, None, None))              # (the share still comes from the process group)
        res = {}
        for t in mine:                                       # every rank: ITS solves, different from rank to rank, different in number
            pb = small_problem(na=3, nd=4, nt=1 + t % 2, n=10, Ns=11)
            w = pb["w"]
            eng = OracleEngine(w["xvec"], w["yvec"], w["zvec"])
            rng = np.random.default_rng(10 + t)
            d = rng.normal(size=(pb["na"], pb["P"])) * 0.01
            prob = parallel.ShardedRays(eng, pb["o"], pb["d"], pb["tmax"], pb["Ns"], dobs=d, cdct=np.full(d.shape, 1e-4), i0=pb["i0"])
            assert (prob.world, prob.rank, prob.multi, prob.lo, prob.hi) == (1, 0, False, 0, pb["P"])
            x, h = solvers.sirt(prob, torch.from_numpy(pb["x0"].copy()), n_iter=3)
            xc, hc = solvers.cgls(prob, torch.from_numpy(pb["x0"].copy()), n_iter=3)
            res[t] = (x.numpy(), np.array(h), xc.numpy(), np.array(hc))
        q.put((rank, mine, res))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_independent_ranks_solve_their_own_problems_without_any_collective():
    """parallel.INDEPENDENT_RANKS: with a process group of two ranks, every rank's ShardedRays is a whole problem (solves shared
    over ranks, inversion/parallel_solves.py:solve_share): 3 + 2 solves of different sizes, no collective (a mismatch would hang or
    abort gloo), results equal to the same solves without any group."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_independent_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    shares = {}
    for _ in range(world):
        rank, mine, res = get_or_fail(q, procs, 240)
        shares[rank] = mine
        got.update(res)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert shares == {0: [0, 1, 2], 1: [3, 4]} and sorted(got) == [0, 1, 2, 3, 4]
    sys.path.insert(0, HERE)
    from ionotomo_amd import parallel, solvers
    from problems import small_problem
    from cpu_engine import OracleEngine
    assert parallel.INDEPENDENT_RANKS is False
    for t in range(5):
        pb = small_problem(na=3, nd=4, nt=1 + t % 2, n=10, Ns=11)
        w = pb["w"]
        eng = OracleEngine(w["xvec"], w["yvec"], w["zvec"])
        d = np.random.default_rng(10 + t).normal(size=(pb["na"], pb["P"])) * 0.01
        prob = parallel.ShardedRays(eng, pb["o"], pb["d"], pb["tmax"], pb["Ns"], dobs=d, cdct=np.full(d.shape, 1e-4), i0=pb["i0"])
        x, h = solvers.sirt(prob, torch.from_numpy(pb["x0"].copy()), n_iter=3)
        xc, hc = solvers.cgls(prob, torch.from_numpy(pb["x0"].copy()), n_iter=3)
        np.testing.assert_array_equal(got[t][0], x.numpy())
        np.testing.assert_array_equal(got[t][1], np.array(h))
        np.testing.assert_array_equal(got[t][2], xc.numpy())
        np.testing.assert_array_equal(got[t][3], np.array(hc))
