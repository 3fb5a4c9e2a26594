"""Host logic of the stacked solves (ionotomo_amd/inversion/parallel_solves.py): where a solve's rays and nodes go in the stacked
problem, what is refused.  No GPU: the engine is only created when a launch needs it."""
import numpy as np
import pytest
import torch

from ionotomo_amd.inversion.parallel_solves import StackedSolves, solve_share


def grids():
    x, y, z = np.linspace(-10, 10, 11), np.linspace(-5, 5, 6), np.linspace(0, 100, 21)
    return [(x, y, z), (x + 3, y + 1, z - 2), (x - 7, y, z)]


def test_rays_land_in_their_own_slab_at_the_same_grid_coordinates():
    gs = grids()
    st = StackedSolves(gs)
    assert st.B == 3 and len(st.xvec) == 33 and st.xvec[0] == -10 and np.allclose(np.diff(st.xvec), 2.0)
    rng = np.random.default_rng(0)
    oo, dd = [], []
    for x, y, z in gs:
        o = np.stack([rng.uniform(x[2], x[-3], (4, 5)), rng.uniform(y[1], y[-2], (4, 5)), np.full((4, 5), z[0] + 1.0)], -1)
        d = np.stack([rng.normal(size=(4, 5)) * 0.02, rng.normal(size=(4, 5)) * 0.02, np.ones((4, 5))], -1)
        oo.append(o), dd.append(d / np.linalg.norm(d, axis=-1, keepdims=True))
    O, D = st.rays(oo, dd, 90.0)
    assert O.shape == (4, 15, 3) and st.pairs == [5, 5, 5]
    for b, (x, y, z) in enumerate(gs):
        blk = O[:, 5 * b:5 * b + 5]
        # grid coordinates of a moved origin in the stacked grid = b nx + its coordinates in its own grid (y, z: unchanged)
        fx = (blk[..., 0] - st.xvec[0]) / 2.0
        np.testing.assert_allclose(fx, b * 11 + (oo[b][..., 0] - x[0]) / 2.0, atol=1e-12)
        np.testing.assert_allclose((blk[..., 1] - st.yvec[0]) / 2.0, (oo[b][..., 1] - y[0]) / 2.0, atol=1e-12)
        np.testing.assert_allclose((blk[..., 2] - st.zvec[0]) / 5.0, (oo[b][..., 2] - z[0]) / 5.0, atol=1e-12)
        assert np.array_equal(D[:, 5 * b:5 * b + 5], dd[b])
        assert fx.min() >= b * 11 and fx.max() <= b * 11 + 10
    # per-ray and per-node results come back as views, block by block
    v = torch.arange(4 * 15, dtype=torch.float64)
    parts = st.split_rays(v, 4)
    assert [tuple(p.shape) for p in parts] == [(4, 5)] * 3 and float(parts[1][0, 0]) == 5.0
    g = torch.arange(33 * 6 * 21, dtype=torch.float64)
    blocks = st.split_grid(g)
    assert [tuple(b_.shape) for b_ in blocks] == [(11, 6, 21)] * 3 and float(blocks[2][0, 0, 0]) == 22 * 6 * 21
    assert st.stack_rays([np.ones((4, 5)) * b for b in range(3)]).shape == (4, 15)
    sums = st.per_solve_sum(torch.as_tensor(st.stack_rays([np.ones((4, 5)) * (b + 1) for b in range(3)])), 4)
    assert sums.tolist() == [20.0, 40.0, 60.0]


def test_what_is_refused():
    gs = grids()
    with pytest.raises(ValueError, match="shape"):
        StackedSolves([gs[0], (gs[1][0][:-1], gs[1][1], gs[1][2])])
    with pytest.raises(ValueError, match="spacing"):
        StackedSolves([gs[0], (gs[1][0] * 1.5, gs[1][1], gs[1][2])])
    with pytest.raises(ValueError, match="no grid"):
        StackedSolves([])
    assert StackedSolves(gs[0]).B == 1 and StackedSolves(gs[0], count=3).B == 3 and StackedSolves([gs[0]] * 3).B == 3
    with pytest.raises(ValueError, match="count"):
        StackedSolves(gs, count=2)
    big = (np.linspace(0, 1, 256), np.linspace(0, 1, 256), np.linspace(0, 1, 256))
    with pytest.raises(ValueError, match="4 GiB"):
        StackedSolves(big, count=32)
    assert StackedSolves(big, count=31).B == 31 and StackedSolves(big, count=32, storage="f32").B == 32
    assert StackedSolves(big, count=32, allow_general=True).B == 32
    st = StackedSolves(gs[0], count=2)
    o = np.zeros((2, 3, 3))
    d = np.tile(np.array([0.0, 0.0, 1.0]), (2, 3, 1))
    st.rays([o, o], [d, d], 50.0)
    # a ray that would walk into the neighbouring slab is the reference's bounds error (geometry/tri_cubic.py:86-103), not a
    # silent read of the neighbour's model
    d2 = d.copy()
    d2[1, 2] = [0.6, 0.0, 0.8]
    with pytest.raises(ValueError, match="out of bounds"):
        st.rays([o, o], [d, d2], 50.0)
    with pytest.raises(ValueError, match="solves"):
        st.rays([o], [d], 50.0)
    with pytest.raises(ValueError, match="antennas"):
        st.rays([o, o[:1]], [d, d[:1]], 50.0)


def test_solves_are_shared_over_ranks_without_overlap():
    for n, world in ((10, 4), (3, 8), (64, 8), (0, 2), (7, 1)):
        got = [list(solve_share(n, world, r)) for r in range(world)]
        assert sorted(sum(got, [])) == list(range(n))
        assert max(len(g) for g in got) - min(len(g) for g in got) <= 1
    assert list(solve_share(5)) == [0, 1, 2, 3, 4]               # no process group: one rank
    with pytest.raises(ValueError):
        solve_share(4, 2, 2)
