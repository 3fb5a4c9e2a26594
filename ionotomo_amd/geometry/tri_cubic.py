"""``TriCubic`` -- drop-in for ``ionotomo.geometry.tri_cubic.TriCubic`` (geometry/tri_cubic.py:13-103).

Same constructor, attributes and methods; ``interp`` / ``extrapolate`` run on the GPU through
libionotomo_hip.  As in the reference the default interpolant is TRILINEAR (the reference wraps
scipy's RegularGridInterpolator with method='linear', :22,59,69-70); ``kind='cubic'`` selects the
Lekien-Marsden tricubic the reference's notebooks specify.

``M`` is a host numpy array the caller may mutate in place (the reference does:
``np.exp(ne_tci.M, out=ne_tci.M)``, inversion/forward_equation.py:42), so every numeric call
re-uploads it; device-resident workflows use ``ionotomo_amd.engine.RayEngine`` instead.
"""
import numpy as np

from .. import _lib


def simpson_axis_weights(v):
    """1-D composite-Simpson weights on the (possibly non-uniform) nodes ``v`` with the
    reference-era even='avg' rule (host helper for ``inner``; tomography/integrate.py:50-153)."""
    v = np.asarray(v, dtype=np.float64)
    n = v.size

    def basic(s):
        w = np.zeros(s.size)
        if s.size < 3:
            return w
        h = np.diff(s)
        h0, h1 = h[0:s.size - 2:2], h[1:s.size - 1:2]
        hs = h0 + h1
        w[0:s.size - 2:2] += hs / 6.0 * (2.0 - h1 / h0)
        w[1:s.size - 1:2] += hs / 6.0 * (hs * hs / (h0 * h1))
        w[2:s.size:2] += hs / 6.0 * (2.0 - h0 / h1)
        return w
    if n == 2:
        return np.full(2, 0.5 * (v[1] - v[0]))
    if n % 2 == 1:
        return basic(v)
    a = np.zeros(n)
    a[:-1] = basic(v[:-1])
    a[-2:] += 0.5 * (v[-1] - v[-2])
    b = np.zeros(n)
    b[1:] = basic(v[1:])
    b[:2] += 0.5 * (v[1] - v[0])
    return 0.5 * (a + b)


class TriCubic(object):
    def __init__(self, xvec=None, yvec=None, zvec=None, M=None, filename=None, kind="linear", storage="f64"):
        self.kind = kind
        self.storage = storage
        if filename is not None:
            self.load(filename)
        else:
            self.xvec = xvec
            self.yvec = yvec
            self.zvec = zvec
            self.M = M

    # axis / value properties: geometry/tri_cubic.py:24-59
    @property
    def xvec(self):
        return self._xvec

    @xvec.setter
    def xvec(self, val):
        self._xvec = np.array(val, dtype=np.float64)
        self.nx = int(np.size(self._xvec))

    @property
    def yvec(self):
        return self._yvec

    @yvec.setter
    def yvec(self, val):
        self._yvec = np.array(val, dtype=np.float64)
        self.ny = int(np.size(self._yvec))

    @property
    def zvec(self):
        return self._zvec

    @zvec.setter
    def zvec(self, val):
        self._zvec = np.array(val, dtype=np.float64)
        self.nz = int(np.size(self._zvec))

    @property
    def M(self):
        return self._M

    @M.setter
    def M(self, val):
        val = np.asarray(val)
        assert not np.any(np.isnan(val)) and not np.any(np.isinf(val))
        if val.ndim == 1:
            val = val.reshape((self.nx, self.ny, self.nz))
        assert val.shape[0] == self.nx
        assert val.shape[1] == self.ny
        assert val.shape[2] == self.nz
        self._M = val

    # -- device plumbing --------------------------------------------------------------------------
    def bind(self, ctx=None):
        """Upload axes + current values to ``ctx`` (default: the process context); returns it."""
        ctx = ctx or _lib.default_context()
        ctx.set_grid(self._xvec, self._yvec, self._zvec, self._M, storage=self.storage)
        return ctx

    # -- numerics ---------------------------------------------------------------------------------
    def interp(self, x, y, z):
        """Values at points (any equal shapes).  Out-of-grid points raise ``ValueError`` like the
        reference's ``bounds_error=True`` interpolator."""
        return self.bind().interp(x, y, z, kind=self.kind, extrapolate=False)

    def extrapolate(self, x, y, z):
        """Linear extension outside the grid (geometry/tri_cubic.py:71-75)."""
        return self.bind().interp(x, y, z, kind=self.kind, extrapolate=True)

    def inner(self, M, inplace=False):
        """Simpson^3 inner product of ``self.M`` with ``M`` (geometry/tri_cubic.py:61-67)."""
        if not inplace:
            M = M * self.M
        else:
            M *= self.M
        wx, wy, wz = (simpson_axis_weights(v) for v in (self._xvec, self._yvec, self._zvec))
        return float(np.einsum("ijk,i,j,k->", M, wx, wy, wz))

    def copy(self, **kwargs):
        kw = dict(kind=self.kind, storage=self.storage)
        kw.update(kwargs)
        return TriCubic(self.xvec.copy(), self.yvec.copy(), self.zvec.copy(), self.M.copy(), **kw)

    def get_model_coordinates(self):
        X, Y, Z = np.meshgrid(self.xvec, self.yvec, self.zvec, indexing='ij')
        return X.flatten(order='C'), Y.flatten(order='C'), Z.flatten(order='C')

    # -- storage: HDF5 "TCI/{xvec,yvec,zvec,M}" (geometry/tri_cubic.py:81-99; + the frame attributes inversion/solution.py:26-47
    #    adds to "TCI") through the pure-numpy reader / writer of utils/hdf5_lite.py, or .npz with the same four names ----------
    def load(self, filename, **kwargs):
        self.frame_attrs = {}
        if str(filename).endswith(".npz"):
            with np.load(filename, allow_pickle=False) as z:
                xvec, yvec, zvec, M = z["xvec"], z["yvec"], z["zvec"], z["M"]
        else:
            from ..utils import hdf5_lite
            t = hdf5_lite.read(filename)["TCI"]
            xvec, yvec, zvec, M = t["xvec"], t["yvec"], t["zvec"], t["M"]
            self.frame_attrs = dict(t.get("@attrs", {}))          # obstime, fixtime [gps s], location [km], phase [deg] when present
        self.xvec, self.yvec, self.zvec = xvec, yvec, zvec
        self.M = M

    def save(self, filename, frame_attrs=None):
        arrays = {k: np.asarray(v, dtype=np.double) for k, v in
                  (("xvec", self.xvec), ("yvec", self.yvec), ("zvec", self.zvec), ("M", self.M))}
        if str(filename).endswith(".npz"):
            with open(filename, "wb") as f:
                np.savez(f, **arrays)
            return
        from ..utils import hdf5_lite
        attrs = frame_attrs if frame_attrs is not None else getattr(self, "frame_attrs", None)
        if attrs:
            arrays["@attrs"] = dict(attrs)
        hdf5_lite.write(filename, {"TCI": arrays})


def bisection(array, value):
    """Index j with array[j] <= value < array[j+1]; -1 / len(array) when out of range
    (geometry/tri_cubic.py:105-132)."""
    n = len(array)
    if value < array[0]:
        return -1
    if value > array[n - 1]:
        return n
    if value == array[n - 1]:
        return n - 1
    return int(min(max(np.searchsorted(array, value, side='right') - 1, 0), n - 2))
