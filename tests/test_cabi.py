"""CPU-only checks of the C-ABI library and host logic (no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from ionotomo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "ionotomo_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(iono_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "missing export %s" % name
    assert sorted(declared) == sorted(_lib.EXPORTED), set(declared) ^ set(_lib.EXPORTED)
    assert lib.iono_version() >= 100


def test_header_constants_match_the_binding():
    """Enumerators of include/ionotomo_hip.h against the Python constants that mirror them."""
    hdr = open(os.path.join(ROOT, "include", "ionotomo_hip.h")).read()
    enums = {}
    for body in re.findall(r"enum\s*\{([^}]*)\}", hdr):
        for name, val in re.findall(r"(IONO_[A-Z0-9_]+)\s*=\s*(-?\d+)", body):
            enums[name] = int(val)
    for name, py in (("IONO_OK", _lib.OK), ("IONO_ERR_OOB", _lib.ERR_OOB), ("IONO_ERR_NONFINITE", _lib.ERR_NONFINITE),
                     ("IONO_ERR_SHAPE", _lib.ERR_SHAPE), ("IONO_ERR_HIP", _lib.ERR_HIP), ("IONO_ERR_ARG", _lib.ERR_ARG),
                     ("IONO_F64", _lib.F64), ("IONO_F32", _lib.F32), ("IONO_WALK_FORWARD", _lib.WALK_FORWARD),
                     ("IONO_WALK_ADJOINT", _lib.WALK_ADJOINT), ("IONO_INTERP_TRILINEAR", _lib.interp_kind("linear")),
                     ("IONO_INTERP_TRICUBIC", _lib.interp_kind("cubic"))):
        assert enums.get(name) == py, (name, enums.get(name), py)


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        _lib.Context(0)
    import ionotomo_amd as it
    tci = it.TriCubic(np.linspace(0, 1, 4), np.linspace(0, 1, 4), np.linspace(0, 1, 4), np.zeros((4, 4, 4)))
    with pytest.raises(RuntimeError):
        tci.interp(np.array([0.5]), np.array([0.5]), np.array([0.5]))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ionotomo_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_tricubic_container_semantics():
    import ionotomo_amd as it
    xv, yv, zv = np.linspace(-0.1, 1.1, 10), np.linspace(-0.1, 1.1, 11), np.linspace(-0.1, 1.11, 12)
    x, y, z = np.meshgrid(xv, yv, zv, indexing='ij')
    M = x * y * z + x - y - 2 * z + x ** 2
    tci = it.TriCubic(xv, yv, zv, M)
    assert (tci.nx, tci.ny, tci.nz) == (10, 11, 12)
    t2 = tci.copy()
    assert np.all(t2.M == tci.M) and t2.M is not tci.M
    t2.M = M.ravel()                                   # flat M is reshaped (geometry/tri_cubic.py:52-54)
    assert t2.M.shape == (10, 11, 12)
    with pytest.raises(AssertionError):
        t2.M = np.full((10, 11, 12), np.nan)
    with pytest.raises(AssertionError):
        t2.M = np.zeros((3, 3, 3))
    X, Y, Z = tci.get_model_coordinates()
    assert X.shape == (10 * 11 * 12,) and X[12 * 11] == xv[1]
    from scipy.integrate import simpson
    ref = simpson(simpson(simpson(M * M, x=zv, axis=2), x=yv, axis=1), x=xv, axis=0)
    odd = it.TriCubic(xv[:9], yv, zv[:11], M[:9, :, :11])
    ref_odd = simpson(simpson(simpson(odd.M * odd.M, x=zv[:11], axis=2), x=yv, axis=1), x=xv[:9], axis=0)
    assert abs(odd.inner(odd.M) - ref_odd) < 1e-12 * abs(ref_odd)
    assert abs(tci.inner(M) - ref) < 1e-2 * abs(ref)      # even axes: 'avg' rule vs scipy-1.15 rule
    t0 = it.clock()                                    # the reference exports its wall timer at top level (__init__.py:26)
    assert isinstance(t0, float) and it.clock() >= t0
    assert it.bisection(xv, xv[3] + 1e-3) == 3 and it.bisection(xv, -5) == -1 and it.bisection(xv, 5) == 10


def test_radio_array_lofar():
    import ionotomo_amd as it
    ra = it.RadioArray(array_file=it.RadioArray.lofar_array)
    assert ra.Nantenna == 62 and ra.get_antenna_locs().shape == (62, 3)
    assert ra.get_antenna_labels()[0] == "CS001HBA0"
    assert np.allclose(ra.get_center(), ra.get_antenna_locs().mean(0))
    enu = ra.enu_km()
    assert enu.shape == (62, 3) and abs(enu.mean(0)).max() < 1e-6 and np.abs(enu[:, 2]).max() < 1.0
    assert ra.get_antenna_idx("CS002HBA1") == 3
    assert it.RadioArray(array_file=it.RadioArray.gmrt_array).Nantenna == 32
    assert it.RadioArray(array_file=it.RadioArray.lofar_cycle0_array).Nantenna == 47
    ex = it.generate_example_radio_array(Nant=7, seed=1)
    assert ex.Nantenna == 7


def test_dispatch_table_is_pinned():
    """The library's ONE dispatch table (include/ionotomo_hip.h: iono_dispatch_name -- pure host code, asked here without a GPU):
    which kernel a launch gets from the facts of the launch.  Every launcher asks the same pick_* functions."""
    n = _lib.dispatch_name
    ideal = dict(tier=2, cubic_fast=1, cubic_records=1, ideal_axes=1, q4_ok=1, Ns=257, fermat_lm_few_min=32768, fermat_poly_max=4096,
                 fermat_lin4_max=4096, fermat_coop_max=2 ** 62, axes_bytes=6144)
    # forward TEC: float64 / float32 storage x plan / no plan x interpolant x axis tier
    assert n("forward", R=260400, **ideal) == "k_forward_straight_u<double>"
    assert n("forward", R=260400, fwd_bundles=4597, **ideal) == "k_forward_bundle<0>"
    assert n("forward", R=260400, fwd_bundles=2484, fwd_tail=126600, **ideal) == "k_forward_bundle<0> + k_forward_straight_u<double>"
    assert n("forward", R=260400, storage=1, **ideal) == "k_forward_straight_q4"
    assert n("forward", R=260400, storage=1, fwd_bundles=4597, **ideal) == "k_forward_bundle_f32"
    assert n("forward", R=260400, storage=1, fwd_bundles=10, fwd_tail=5, **ideal) == "k_forward_bundle_f32 + k_forward_straight_q4"
    assert n("forward", R=260400, storage=1, **dict(ideal, q4_ok=0)) == "k_forward_straight_u<float>"
    assert n("forward", R=2604, interp_kind=1, **ideal) == "k_forward_straight_lm"
    assert n("forward", R=260400, interp_kind=1, fwd_bundles=4597, **ideal) == "k_forward_bundle_lm"
    assert n("forward", R=260400, interp_kind=1, fwd_bundles=9, fwd_tail=3, **ideal) == "k_forward_bundle_lm + k_forward_straight_lm"
    assert n("forward", R=100, tier=1, Ns=65) == "k_forward_straight_fast<double>"
    assert n("forward", R=100, tier=0, Ns=65) == "k_forward_straight<double, IONO_INTERP_TRILINEAR>"
    assert n("forward", R=100, tier=1, interp_kind=1, Ns=65) == "k_forward_straight<double, IONO_INTERP_TRICUBIC>"
    assert n("forward", R=100, interp_kind=1, **dict(ideal, cubic_fast=0)) == "k_forward_straight<double, IONO_INTERP_TRICUBIC>"      # IONOTOMO_VARIANT=4
    # back-projection
    assert n("adjoint", R=260400, **ideal) == "k_adjoint_straight_tile<AT, MODE, 4, false>"
    assert n("adjoint", R=260400, adj_planned=1, adj_seg_lanes=16, **ideal) == "k_adjoint_binned<AT, 0, double, 16, false>"
    assert n("adjoint", R=260400, adj_planned=1, adj_seg_lanes=8, deterministic=1, **ideal) == "k_adjoint_binned<double, 0, double, 8, true> + FixConvert"
    assert n("adjoint", R=260400, adj_planned=1, adj_tiles=1, adj_seg_lanes=16, interp_kind=1, **ideal) == "2 x k_adjoint_binned_lm4<16, true> + k_lm_fold_{z,y,x}_tiles"
    assert n("adjoint", R=260400, interp_kind=1, **ideal) == "8 x k_adjoint_straight_tile<double, MODE, 4, true> + k_lm_fold_{z,y,x}"
    assert n("adjoint", R=260400, deterministic=1, **ideal).startswith("refused")
    assert n("adjoint", R=260400, adj_planned=1, variant=2, **ideal) == "k_adjoint_straight<AT, MODE, IONO_INTERP_TRILINEAR>"
    assert n("adjoint", R=100, tier=1, interp_kind=1, Ns=65) == "k_adjoint_straight<AT, MODE, IONO_INTERP_TRICUBIC>"
    # Fermat tracer: batch size decides the mapping
    assert n("trace", R=2604, bend=1, **ideal) == "k_trace_fermat_poly<true>"
    assert n("trace", R=2604, bend=0, **dict(ideal, ideal_axes=0, fermat_lin4_max=131072)) == "k_trace_fermat_lin4<false>"
    assert n("trace", R=620000, bend=1, **ideal) == "k_trace_fermat<IONO_INTERP_TRILINEAR, true>"
    assert n("trace", R=2604, bend=1, interp_kind=1, **ideal) == "k_trace_fermat_lm<true, 8>"
    assert n("trace", R=620000, bend=1, interp_kind=1, **ideal) == "k_trace_fermat_lm<true, 2>"
    assert n("trace", R=2604, bend=1, interp_kind=1, **dict(ideal, ideal_axes=0)) == "k_trace_fermat_coop<true>"
    assert n("trace", R=2604, bend=1, interp_kind=1, variant=3, **ideal) == "k_trace_fermat<IONO_INTERP_TRICUBIC, true>"
    # fused trace + integrate, and its transpose
    assert n("fermat_forward", R=620000, bend=1, interp_kind=1, **ideal) == "k_fermat_tec_lm<true, 2, false>"
    assert n("fermat_forward", R=2604, bend=1, interp_kind=1, **ideal) == "k_fermat_tec_lm<true, 8, false>"
    assert n("fermat_forward", R=620000, bend=1, **ideal) == "k_fermat_tec<IONO_INTERP_TRILINEAR, true, false>"
    assert n("fermat_adjoint", R=620000, bend=1, interp_kind=1, **ideal) == "k_fermat_tec_lm<true, 2, true>"
    assert n("fermat_adjoint", R=2604, bend=1, interp_kind=1, **ideal) == "k_fermat_tec<IONO_INTERP_TRICUBIC, true, true>"      # (engine.py: two-step there)
    # phase observable
    assert n("phase_forward", R=260400, fwd_bundles=4597, **ideal) == "k_forward_bundle<NF>"
    assert n("phase_forward", R=260400, **ideal) == "k_forward_phase_u<double, NF>"
    assert n("phase_forward", R=100, tier=0, Ns=65) == "k_forward_phase_straight<double, false>"
    assert n("phase_adjoint", R=260400, adj_planned=1, adj_seg_lanes=16, **ideal) == "k_adjoint_binned<double, NF, double, 16>"
    assert n("phase_adjoint", R=260400, **ideal) == "k_adjoint_straight_tile<double, 0, 4, false, true, double>"
    with pytest.raises(ValueError):
        n(9)
    with pytest.raises(KeyError):
        n("forward", no_such_fact=1)
