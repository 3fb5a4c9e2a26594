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
