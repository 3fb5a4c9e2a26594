"""ionotomo_amd/utils/hdf5_lite.py -- the HDF5 subset of the reference's on-disk containers -- against REAL HDF5:
files the reference's own TriCubic.save / DataPack.save wrote through h5py 3.3 on libhdf5 1.10.6 in the build container
(tests/golden/*_reference_h5py.hdf5, made by oracle/make_golden_conda.py) are read here; files written here are read back
here and, where the build image's second interpreter (h5py) or h5dump exist, by real libhdf5.  CPU only."""
import json
import os
import subprocess

import numpy as np
import pytest

from ionotomo_amd import TriCubic
from ionotomo_amd.astro.real_data import DataPack, generate_example_datapack
from ionotomo_amd.utils import hdf5_lite

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONDA_PY = "/opt/conda/bin/python3.9"
H5DUMP = "/opt/conda/bin/h5dump"


def test_reads_the_tci_file_the_reference_wrote():
    exp = np.load(os.path.join(GOLDEN, "tci_reference_h5py_expected.npz"))
    tree = hdf5_lite.read(os.path.join(GOLDEN, "tci_reference_h5py.hdf5"))
    assert set(tree) == {"TCI"} and set(tree["TCI"]) == {"xvec", "yvec", "zvec", "M", "@attrs"}
    for k in ("xvec", "yvec", "zvec", "M"):
        assert tree["TCI"][k].dtype == np.float64 and np.array_equal(tree["TCI"][k], exp[k])
    a = tree["TCI"]["@attrs"]
    assert a["obstime"] == float(exp["obstime"]) and a["fixtime"] == float(exp["fixtime"])
    assert np.array_equal(a["location"], exp["location"]) and np.array_equal(a["phase"], exp["phase"])
    # ... through the product's container class (geometry/tri_cubic.py:81-88)
    tci = TriCubic(filename=os.path.join(GOLDEN, "tci_reference_h5py.hdf5"))
    assert np.array_equal(tci.M, exp["M"]) and np.array_equal(tci.zvec, exp["zvec"]) and tci.frame_attrs["obstime"] == float(exp["obstime"])


def test_reads_the_datapack_file_the_reference_wrote():
    exp = np.load(os.path.join(GOLDEN, "datapack_reference_h5py_expected.npz"))
    dp = DataPack(filename=os.path.join(GOLDEN, "datapack_reference_h5py.hdf5"))
    assert list(dp.antenna_labels) == list(exp["labels"]) and list(dp.patch_names) == list(exp["patch_names"])
    assert list(dp.timestamps) == list(exp["timestamps"])
    assert np.array_equal(dp.antennas, exp["locs"]) and np.array_equal(dp.freqs, exp["freqs"])
    assert np.allclose(np.rad2deg(dp.directions), np.stack([exp["ra"], exp["dec"]], -1), rtol=0, atol=1e-12)
    assert np.array_equal(dp.variance, exp["variance"]) and dp.radio_array.frequency == float(exp["frequency"])
    assert dp.ref_ant == str(exp["ref_ant"])
    # load() re-references the phases to the stored reference antenna (astro/real_data.py:116), as the reference does
    i0 = list(exp["labels"]).index(str(exp["ref_ant"]))
    assert np.allclose(dp.phase, exp["phase"] - exp["phase"][i0], rtol=0, atol=1e-15)
    raw = hdf5_lite.read(os.path.join(GOLDEN, "datapack_reference_h5py.hdf5"))["datapack"]
    assert np.array_equal(raw["phase"], exp["phase"]) and np.array_equal(raw["clock"], exp["clock"]) and np.array_equal(raw["const"], exp["const"])
    assert np.array_equal(raw["times"]["gps"], exp["gps"]) and raw["phase@attrs"]["ref_ant"] == str(exp["ref_ant"])


def _example(tmp_path):
    dp = generate_example_datapack(Nant=5, Ntime=3, Ndir=4, time=1.5e9, seed=4)
    dp.set_variance(np.random.default_rng(0).uniform(size=dp.phase.shape), -1, -1, -1, -1)
    f = str(tmp_path / "dp.hdf5")
    dp.save(f)
    rng = np.random.default_rng(1)
    tci = TriCubic(np.linspace(0, 1, 4), np.linspace(-2, 2, 5), np.linspace(0, 50, 6), rng.normal(size=(4, 5, 6)))
    g = str(tmp_path / "tci.hdf5")
    tci.save(g, frame_attrs={"obstime": 1.2e9, "fixtime": 1.2e9 + 32, "location": np.array([3826.5, 461.0, 5064.9]), "phase": [210.0, 35.0]})
    return dp, f, tci, g


def test_round_trip_of_both_containers(tmp_path):
    dp, f, tci, g = _example(tmp_path)
    back = DataPack(filename=f)
    assert back.ref_ant == dp.ref_ant and repr(back) == repr(dp)
    for k in ("antennas", "freqs", "phase", "variance", "clock", "const"):
        assert np.array_equal(getattr(back, k), getattr(dp, k)), k
    assert np.allclose(back.times, dp.times, rtol=0, atol=1e-6) and np.allclose(back.directions, dp.directions, rtol=0, atol=1e-15)
    for k in ("antenna_labels", "patch_names", "timestamps"):
        assert list(getattr(back, k)) == list(getattr(dp, k))
    t2 = TriCubic(filename=g)
    assert np.array_equal(t2.M, tci.M) and np.array_equal(t2.xvec, tci.xvec) and t2.frame_attrs["fixtime"] == 1.2e9 + 32
    assert np.array_equal(t2.frame_attrs["location"], [3826.5, 461.0, 5064.9])
    # unsupported structures are refused, not misread
    with open(str(tmp_path / "junk.hdf5"), "wb") as fh:
        fh.write(b"not an hdf5 file at all")
    with pytest.raises(ValueError):
        hdf5_lite.read(str(tmp_path / "junk.hdf5"))
    # groups of many links, empty groups, unicode strings, 0-d and empty arrays
    tree = {"g": {("d%02d" % i): np.arange(i, dtype=float) for i in range(30)}, "empty": {}, "s": np.array(["été", "", "x" * 300], dtype=object)}
    hdf5_lite.write(str(tmp_path / "many.hdf5"), tree)
    back = hdf5_lite.read(str(tmp_path / "many.hdf5"))
    assert list(back["s"]) == ["été", "", "x" * 300] and back["empty"] == {} and len(back["g"]) == 30
    assert all(np.array_equal(back["g"]["d%02d" % i], np.arange(i, dtype=float)) for i in range(30))


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="the image's second interpreter (h5py) is not there")
def test_files_written_here_are_read_by_real_h5py(tmp_path):
    dp, f, tci, g = _example(tmp_path)
    code = r'''
import sys, json, warnings
warnings.filterwarnings("ignore")
import h5py, numpy as np
out = {}
with h5py.File(sys.argv[1], "r") as f:
    out["labels"] = [s.decode() if isinstance(s, bytes) else s for s in f["datapack/antennas/labels"][:]]
    out["timestamps"] = [s.decode() if isinstance(s, bytes) else s for s in f["datapack/times/timestamps"][:]]
    out["frequency"] = float(f["datapack/antennas"].attrs["frequency"])
    r = f["datapack/phase"].attrs["ref_ant"]
    out["ref_ant"] = r.decode() if isinstance(r, bytes) else str(r)
    out["phase_sum"] = float(f["datapack/phase"][...].sum())
    out["phase_shape"] = list(f["datapack/phase"].shape)
    out["locs"] = f["datapack/antennas/locs"][...].tolist()
    out["keys"] = sorted(f["datapack"].keys())
with h5py.File(sys.argv[2], "r") as f:
    out["M"] = f["TCI/M"][...].tolist()
    out["attrs"] = {k: np.asarray(v).tolist() for k, v in f["TCI"].attrs.items()}
print(json.dumps(out))
'''
    res = subprocess.run([CONDA_PY, "-c", code, f, g], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["labels"] == list(dp.antenna_labels) and out["timestamps"] == list(dp.timestamps)
    assert out["frequency"] == dp.radio_array.frequency and out["ref_ant"] == str(dp.ref_ant)
    assert out["phase_shape"] == list(dp.phase.shape) and abs(out["phase_sum"] - dp.phase.sum()) < 1e-9
    assert np.array_equal(np.array(out["locs"]), dp.antennas)
    assert out["keys"] == ["antennas", "clock", "const", "directions", "freqs", "phase", "times", "variance"]
    assert np.array_equal(np.array(out["M"]), tci.M)
    assert out["attrs"]["obstime"] == 1.2e9 and out["attrs"]["phase"] == [210.0, 35.0]


@pytest.mark.skipif(not os.path.exists(H5DUMP), reason="h5dump is not there")
def test_h5dump_accepts_the_files(tmp_path):
    dp, f, tci, g = _example(tmp_path)
    for path, needle in ((f, 'DATASET "patchnames"'), (g, 'ATTRIBUTE "location"')):
        res = subprocess.run([H5DUMP, path], capture_output=True, text=True, timeout=120)
        assert res.returncode == 0 and "error" not in res.stderr.lower(), res.stderr[-2000:]
        assert needle in res.stdout and "H5T_IEEE_F64LE" in res.stdout
