"""CPU restatement (numpy, float64) of the reference's ray-integral hot path.

THIS IS TEST INFRASTRUCTURE -- the parity oracle and the timed CPU baseline.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
it; the product (ionotomo_amd/) never does and fails loudly without its HIP
library.

Pinned against the reference itself: tests/test_oracle_golden.py checks every
function below against tests/golden/*.npz, which oracle/make_golden.py produced
by running the reference's own functions in the build container -- including
the reference-era even-N ``simps(even='avg')`` rule, pinned since round 3 by
running the UNMODIFIED inversion/forward_equation.py on scipy 1.7.1
(oracle/make_golden_conda.py -> forward_tec_even_simps_unmodified.npz; the
rule itself is restated from tomography/integrate.py:50-74,130-153).  ONE
piece has no runnable reference and stays "parity unpinned" (stated in
DESIGN.md): the truly bending Fermat tracer (the shipped one zeroes its
gradients, inversion/fermat.py:54-55; spec: notebooks/FermatClass.ipynb
c0:60-96), which is cross-checked against scipy's LSODA on the same right-hand
side instead.

All ``file:line`` citations are relative to /root/reference/src/ionotomo/.
"""
import numpy as np

TECU = 1e13                      # inversion/forward_equation.py:12 (TEC unit per km)
SPEED_OF_LIGHT = 299792458.0     # inversion/iterative_newton.py:15
PLASMA_CONST = 8.980             # inversion/fermat.py:42  n = sqrt(1 - 8.98^2 ne / nu^2)

QUAD_SIMPSON_AVG = 0      # odd N: composite Simpson; even N: reference-era simps(even='avg')
QUAD_SIMPSON_SCIPY = 1    # odd N: composite Simpson; even N: scipy>=1.11 simpson (Cartwright)
QUAD_TRAPEZOID = 2

INTERP_TRILINEAR = 0      # what geometry/tri_cubic.py ships (scipy RGI, method='linear')
INTERP_TRICUBIC = 1       # notebook Lekien-Marsden, 4th-order central differences


# --------------------------------------------------------------------------- A2a trilinear
def find_cell(grid, x):
    """``i = clip(searchsorted(grid, x) - 1, 0, n-2)`` and ``t = (x-g[i])/(g[i+1]-g[i])``.
    scipy RegularGridInterpolator._find_indices; same rule spelled out in the
    reference's TF port, tomography/interpolation.py:166-196."""
    grid = np.asarray(grid)
    i = np.searchsorted(grid, x) - 1
    i = np.clip(i, 0, grid.size - 2)
    t = (x - grid[i]) / (grid[i + 1] - grid[i])
    return i, t


def out_of_bounds(xvec, yvec, zvec, x, y, z):
    """True where scipy's ``bounds_error=True`` would raise (geometry/tri_cubic.py:22,59)."""
    bad = np.zeros(np.shape(x), dtype=bool)
    for g, p in ((xvec, x), (yvec, y), (zvec, z)):
        p = np.asarray(p)
        bad |= ~((p >= g[0]) & (p <= g[-1]))     # NaN counts as out of bounds
    return bad


def trilinear(xvec, yvec, zvec, M, x, y, z, bounds_error=True):
    """TriCubic.interp (geometry/tri_cubic.py:69-70) == scipy RGI 'linear':
    sum over the 8 cell corners of the product of 1-D weights
    (tomography/interpolation.py:145-158).  ``bounds_error=False`` is
    TriCubic.extrapolate (:71-75): the same formula with t outside [0,1]."""
    x, y, z = np.asarray(x, float), np.asarray(y, float), np.asarray(z, float)
    if bounds_error and np.any(out_of_bounds(xvec, yvec, zvec, x, y, z)):
        raise ValueError("One of the requested xi is out of bounds")
    i, tx = find_cell(xvec, x)
    j, ty = find_cell(yvec, y)
    k, tz = find_cell(zvec, z)
    out = np.zeros(np.shape(x))
    for di, wx in ((0, 1 - tx), (1, tx)):
        for dj, wy in ((0, 1 - ty), (1, ty)):
            for dk, wz in ((0, 1 - tz), (1, tz)):
                out = out + M[i + di, j + dj, k + dk] * (wx * wy * wz)
    return out


# --------------------------------------------------------------------------- A2b tricubic
def fd4_slopes_matrix(g):
    """Rows of the 1-D derivative operator the notebook uses at node i (2 <= i <= n-3):
    (f[i-2] - 8 f[i-1] + 8 f[i+1] - f[i+2]) / (6 (x[i+1] - x[i-1]))
    (notebooks/TricubicInterpolation.ipynb c0:313-987, e.g. c0:326-327)."""
    n = len(g)
    D = np.zeros((n, n))
    for i in range(2, n - 2):
        c = 1.0 / (6.0 * (g[i + 1] - g[i - 1]))
        D[i, i - 2], D[i, i - 1], D[i, i + 1], D[i, i + 2] = c, -8 * c, 8 * c, -c
    return D


def hermite_basis(t):
    """cubic Hermite basis on [0,1]: value-at-0, value-at-1, slope-at-0, slope-at-1, and d/dt."""
    t2, t3 = t * t, t * t * t
    h = np.stack([2 * t3 - 3 * t2 + 1, -2 * t3 + 3 * t2, t3 - 2 * t2 + t, t3 - t2])
    dh = np.stack([6 * t2 - 6 * t, -6 * t2 + 6 * t, 3 * t2 - 4 * t + 1, 3 * t2 - 2 * t])
    return h, dh


def tricubic_axis_weights(g, x, deriv=False, cell_units=True):
    """Per-axis 6-tap weights of the C1 tricubic: cubic Hermite between nodes i, i+1 with
    4th-order central-difference slopes => support i-2 .. i+3.  Returns (i, w[6,...])
    (and dw/dx).  The Lekien-Marsden interpolant with finite-difference derivative data
    (including the mixed ones, which the notebook forms by applying the same 1-D stencil
    along each axis in turn -- verified against its coefficients) is exactly this
    tensor-product form.

    ``cell_units``: Lekien-Marsden works in unit-cell coordinates u = (x - x_i)/h, so the
    derivative data must be df/du = h df/dx.  The notebook feeds the PHYSICAL finite
    differences unscaled (ratio 1, not h, measured on its coefficients), which is only right
    for unit spacing -- its own test output (c0 outputs: f = 0.0634 vs -0.0051 ...) shows the
    resulting garbage on a 0.0157-spaced grid, and is why its ``interp`` short-circuits to
    nearest-voxel (c0:164).  ``cell_units=False`` reproduces that, to pin the stencils on the
    non-uniform golden coefficients; the product uses the correct ``True``."""
    g = np.asarray(g)
    i = np.clip(np.searchsorted(g, x, side='right') - 1, 2, g.size - 4)
    hcell = g[i + 1] - g[i]
    t = (x - g[i]) / hcell
    hb, dhb = hermite_basis(t)
    sc = hcell if cell_units else 1.0
    c0 = sc / (6.0 * (g[i + 1] - g[i - 1]))      # slope at node i   (in cell units)
    c1 = sc / (6.0 * (g[i + 2] - g[i]))          # slope at node i+1 (in cell units)

    def taps(b):
        w = np.zeros((6,) + np.shape(x))
        w[2] += b[0]
        w[3] += b[1]
        for tap, coef in ((0, 1.0), (1, -8.0), (3, 8.0), (4, -1.0)):
            w[tap] += b[2] * c0 * coef
        for tap, coef in ((1, 1.0), (2, -8.0), (4, 8.0), (5, -1.0)):
            w[tap] += b[3] * c1 * coef
        return w
    if deriv:
        return i, taps(hb), taps(dhb) / hcell
    return i, taps(hb)


def tricubic(xvec, yvec, zvec, M, x, y, z, grad=False, cell_units=True):
    """Tricubic value (and gradient) at points; needs 2 <= cell <= n-4 on every axis
    (notebooks/TricubicInterpolation.ipynb c0:107-109)."""
    x, y, z = np.asarray(x, float), np.asarray(y, float), np.asarray(z, float)
    i, wx, dwx = tricubic_axis_weights(xvec, x, True, cell_units)
    j, wy, dwy = tricubic_axis_weights(yvec, y, True, cell_units)
    k, wz, dwz = tricubic_axis_weights(zvec, z, True, cell_units)
    f = np.zeros(np.shape(x))
    fx, fy, fz = np.zeros_like(f), np.zeros_like(f), np.zeros_like(f)
    for a in range(6):
        for b in range(6):
            for c in range(6):
                v = M[i + a - 2, j + b - 2, k + c - 2]
                f = f + v * (wx[a] * wy[b] * wz[c])
                if grad:
                    fx = fx + v * (dwx[a] * wy[b] * wz[c])
                    fy = fy + v * (wx[a] * dwy[b] * wz[c])
                    fz = fz + v * (wx[a] * wy[b] * dwz[c])
    return (f, fx, fy, fz) if grad else f


def lm_polynomial(coeffs, u, v, w):
    """Evaluate the notebook's 64-coefficient cell polynomial: index 16 i + 4 j + k multiplies
    u^i v^j w^k (notebooks/TricubicInterpolation.ipynb c0:286)."""
    A = np.asarray(coeffs).reshape(4, 4, 4)
    pu, pv, pw = u ** np.arange(4), v ** np.arange(4), w ** np.arange(4)
    return np.einsum("ijk,i,j,k->", A, pu, pv, pw)


def interpolate(kind, xvec, yvec, zvec, M, x, y, z):
    if kind == INTERP_TRILINEAR:
        return trilinear(xvec, yvec, zvec, M, x, y, z)
    return tricubic(xvec, yvec, zvec, M, x, y, z)


# --------------------------------------------------------------------------- A5' Simpson
def _basic_simpson_weights(s):
    """Composite Simpson weights on an ODD number of (possibly non-uniform) abscissae,
    last axis.  Per triple with h0, h1:  hs/6 (2 - h1/h0),  hs^3 / (6 h0 h1),  hs/6 (2 - h0/h1)
    (tomography/integrate.py:50-74; scipy _basic_simpson)."""
    n = s.shape[-1]
    w = np.zeros(s.shape)
    if n < 3:
        return w
    h = np.diff(s, axis=-1)
    h0, h1 = h[..., 0:n - 2:2], h[..., 1:n - 1:2]
    hs = h0 + h1
    w[..., 0:n - 2:2] += hs / 6.0 * (2.0 - h1 / h0)
    w[..., 1:n - 1:2] += hs / 6.0 * (hs * hs / (h0 * h1))
    w[..., 2:n:2] += hs / 6.0 * (2.0 - h0 / h1)
    return w


def quadrature_weights(s, rule=QUAD_SIMPSON_AVG):
    """w such that  integral = sum(w * y, -1)  for samples y at abscissae s (last axis)."""
    s = np.asarray(s, float)
    n = s.shape[-1]
    w = np.zeros(s.shape)
    if rule == QUAD_TRAPEZOID or n == 2:
        h = np.diff(s, axis=-1)
        w[..., :-1] += 0.5 * h
        w[..., 1:] += 0.5 * h
        return w
    if n % 2 == 1:
        return _basic_simpson_weights(s)
    if rule == QUAD_SIMPSON_AVG:
        # tomography/integrate.py:130-153: mean of {Simpson on first N-1 + trapezoid on last
        # interval} and {trapezoid on first interval + Simpson on last N-1}
        a = np.zeros(s.shape)
        a[..., :-1] = _basic_simpson_weights(s[..., :-1])
        hl = s[..., -1] - s[..., -2]
        a[..., -1] += 0.5 * hl
        a[..., -2] += 0.5 * hl
        b = np.zeros(s.shape)
        b[..., 1:] = _basic_simpson_weights(s[..., 1:])
        hf = s[..., 1] - s[..., 0]
        b[..., 0] += 0.5 * hf
        b[..., 1] += 0.5 * hf
        return 0.5 * (a + b)
    if rule == QUAD_SIMPSON_SCIPY:
        # scipy >= 1.11 simpson(): Simpson on the first N-1 points + Cartwright's 3-point
        # correction for the last interval
        w[..., :-1] = _basic_simpson_weights(s[..., :-1])
        h0 = s[..., -2] - s[..., -3]
        h1 = s[..., -1] - s[..., -2]
        w[..., -1] += (2 * h1 ** 2 + 3 * h0 * h1) / (6 * (h0 + h1))
        w[..., -2] += (h1 ** 2 + 3 * h0 * h1) / (6 * h0)
        w[..., -3] -= h1 ** 3 / (6 * h0 * (h0 + h1))
        return w
    raise ValueError("unknown quadrature rule")


def simps(y, s, rule=QUAD_SIMPSON_AVG):
    return np.sum(quadrature_weights(s, rule) * y, axis=-1)


def unit_weights(n, rule=QUAD_SIMPSON_AVG):
    """Weights for unit-spaced abscissae 0..n-1 (straight rays sample s uniformly, so
    integral = h * sum(unit_weights * y))."""
    return quadrature_weights(np.arange(n, dtype=float), rule)


# --------------------------------------------------------------------------- A3/A3' ray geometry
def straight_rays(origins, directions, tmax, N):
    """rays[...,4,N] = x,y,z,s for straight rays parametrised by z:
    z = linspace(z0, tmax, N); x = x0 + px/pz (z - z0); s = (z - z0)/pz with p the unit
    direction (inversion/fermat.py:64-72,150-174 with n = 1, grad n = 0; packed as
    geometry/calc_rays.py:78-96).  Identical to tomography/model.py:27-35."""
    o = np.asarray(origins, float)
    d = np.asarray(directions, float)
    p = d / np.linalg.norm(d, axis=-1, keepdims=True)
    frac = np.linspace(0.0, 1.0, N)
    dz = (tmax - o[..., 2])[..., None] * frac                  # z - z0
    rays = np.empty(o.shape[:-1] + (4, N))
    rays[..., 2, :] = o[..., 2, None] + dz
    rays[..., 0, :] = o[..., 0, None] + (p[..., 0] / p[..., 2])[..., None] * dz
    rays[..., 1, :] = o[..., 1, None] + (p[..., 1] / p[..., 2])[..., None] * dz
    rays[..., 3, :] = dz / p[..., 2, None]
    return rays


def straight_rays_s(origins, directions, smax, N):
    """type='s' (arc length as the independent variable, inversion/fermat.py:74-82,165-166) with
    n = 1: s = linspace(0, smax, N), position = origin + p s."""
    o = np.asarray(origins, float)
    d = np.asarray(directions, float)
    p = d / np.linalg.norm(d, axis=-1, keepdims=True)
    s = np.linspace(0.0, smax, N)
    rays = np.empty(o.shape[:-1] + (4, N))
    for a in range(3):
        rays[..., a, :] = o[..., a, None] + p[..., a, None] * s
    rays[..., 3, :] = s
    return rays


# --------------------------------------------------------------------------- A5 forward dTEC
def ne_from_log_model(m, K_ne):
    """ne = K_ne exp(m) / TECU at the NODES (inversion/forward_equation.py:41-43)."""
    return np.exp(m) * (K_ne / TECU)


def forward_tec(rays, xvec, yvec, zvec, ne, rule=QUAD_SIMPSON_AVG, kind=INTERP_TRILINEAR):
    """tec[...] = simps(interp(ne; x,y,z), s) per ray (inversion/forward_equation.py:13-33)."""
    vals = interpolate(kind, xvec, yvec, zvec, ne, rays[..., 0, :], rays[..., 1, :], rays[..., 2, :])
    return simps(vals, rays[..., 3, :], rule)


def forward_equation(rays, K_ne, xvec, yvec, zvec, m, i0, rule=QUAD_SIMPSON_AVG, kind=INTERP_TRILINEAR):
    """dtec[Na,Nt,Nd] = tec - tec[i0] (inversion/forward_equation.py:36-51)."""
    tec = forward_tec(rays, xvec, yvec, zvec, ne_from_log_model(m, K_ne), rule, kind)
    return tec - tec[i0]


def forward_tec_loop(rays, xvec, yvec, zvec, ne, rule=QUAD_SIMPSON_AVG):
    """Per-ray loop form (one interp + one Simpson per ray), the shape of
    inversion/forward_equation.py:13-33 -- used as the single-thread CPU baseline."""
    flat = rays.reshape(-1, 4, rays.shape[-1])
    out = np.empty(flat.shape[0])
    for r in range(flat.shape[0]):
        v = trilinear(xvec, yvec, zvec, ne, flat[r, 0], flat[r, 1], flat[r, 2])
        out[r] = simps(v, flat[r, 3], rule)
    return out.reshape(rays.shape[:-2])


# --------------------------------------------------------------------------- A6 phase forward
def forward_phase(mu, clock, const, xvec, yvec, zvec, rays, freqs, K=1e11, i0=0, rule=QUAD_SIMPSON_AVG,
                  emulate_reference_reshape=False):
    """g[Na,Nt,Nd,Nf] (inversion/iterative_newton.py:86-127):
    ne = K exp(mu) at nodes; per frequency  phi = simps(1 - sqrt(1 - ne/n_p), s),
    n_p = 1.2404e-2 nu^2;  phi -= phi[i0];  phi *= 2 pi nu / c;
    g = const_i + 2 pi nu clock_ij - phi.

    ``emulate_reference_reshape``: as shipped, TriCubic.interp (geometry/tri_cubic.py:70) does
    ``np.reshape(rgi(np.array([x,y,z]).T), np.shape(x))`` -- for the 4-D x this path passes
    (iterative_newton.py:108) the values come back in TRANSPOSED order and are reshaped, not
    transposed back, so every sample lands on the wrong ray.  (1-D x, the only shape
    inversion/forward_equation.py uses, is unaffected.)  The flag reproduces that permutation so
    the golden vector -- the reference's actual output -- pins every other term of this
    function; the product implements the evidently intended, un-permuted semantics."""
    ne = np.exp(mu) * K
    ne_rays = trilinear(xvec, yvec, zvec, ne, rays[..., 0, :], rays[..., 1, :], rays[..., 2, :])
    if emulate_reference_reshape:
        ne_rays = np.reshape(ne_rays.transpose(3, 2, 1, 0), ne_rays.shape)
    w = quadrature_weights(rays[..., 3, :], rule)
    Na, Nt, Nd = rays.shape[:3]
    g = np.empty((Na, Nt, Nd, len(freqs)))
    for l, nu in enumerate(freqs):
        a_ = 2 * np.pi * nu
        n_p = 1.2404e-2 * nu ** 2
        phi = np.sum(w * (1.0 - np.sqrt(1.0 - ne_rays / n_p)), axis=-1)
        phi = (phi - phi[i0]) * (a_ / SPEED_OF_LIGHT)
        g[..., l] = const[:, None, None] + a_ * clock[:, :, None] - phi
    return g


def neg_log_like(g, dobs, CdCt):
    """S = 1/2 sum (dobs - g)^2 / CdCt  (inversion/iterative_newton.py:32-38, full=False)."""
    return 0.5 * np.sum((dobs - g) ** 2 / CdCt)


# --------------------------------------------------------------------------- A7' exact adjoint
def adjoint_tec(rays, xvec, yvec, zvec, w_ray, rule=QUAD_SIMPSON_AVG, kind=INTERP_TRILINEAR):
    """(G^T w)[v] = sum_r w_r sum_k c_{r,k} W_{k,v}: the exact transpose of forward_tec
    (interpolation weights x quadrature weights).  Not in the reference (its gradient.py:15-20 uses
    voxel chord lengths instead -- a different discretisation); defined by A2a / A2b + A5'.
    ``kind``: trilinear (8 weights per sample) or the tricubic's 6 x 6 x 6 tensor-product taps."""
    nx, ny, nz = len(xvec), len(yvec), len(zvec)
    c = quadrature_weights(rays[..., 3, :], rule) * np.asarray(w_ray)[..., None]
    out = np.zeros(nx * ny * nz)
    if kind == INTERP_TRICUBIC:
        i, wx = tricubic_axis_weights(xvec, rays[..., 0, :])
        j, wy = tricubic_axis_weights(yvec, rays[..., 1, :])
        k, wz = tricubic_axis_weights(zvec, rays[..., 2, :])
        for a in range(6):
            for b in range(6):
                wab = c * wx[a] * wy[b]
                for cc in range(6):
                    idx = (k + cc - 2) + nz * ((j + b - 2) + ny * (i + a - 2))
                    out += np.bincount(idx.ravel(), weights=(wab * wz[cc]).ravel(), minlength=out.size)
        return out.reshape(nx, ny, nz)
    i, tx = find_cell(xvec, rays[..., 0, :])
    j, ty = find_cell(yvec, rays[..., 1, :])
    k, tz = find_cell(zvec, rays[..., 2, :])
    for di, wx in ((0, 1 - tx), (1, tx)):
        for dj, wy in ((0, 1 - ty), (1, ty)):
            for dk, wz in ((0, 1 - tz), (1, tz)):
                idx = (k + dk) + nz * ((j + dj) + ny * (i + di))
                out += np.bincount(idx.ravel(), weights=(c * wx * wy * wz).ravel(), minlength=out.size)
    return out.reshape(nx, ny, nz)


def scatter_trilinear(rays, xvec, yvec, zvec, c):
    """sum_{r,k} c[r,k] W_{k,v}: the trilinear transpose with arbitrary per-SAMPLE weights c[..., Ns]."""
    nx, ny, nz = len(xvec), len(yvec), len(zvec)
    i, tx = find_cell(xvec, rays[..., 0, :])
    j, ty = find_cell(yvec, rays[..., 1, :])
    k, tz = find_cell(zvec, rays[..., 2, :])
    out = np.zeros(nx * ny * nz)
    for di, wx in ((0, 1 - tx), (1, tx)):
        for dj, wy in ((0, 1 - ty), (1, ty)):
            for dk, wz in ((0, 1 - tz), (1, tz)):
                idx = (k + dk) + nz * ((j + dj) + ny * (i + di))
                out += np.bincount(idx.ravel(), weights=(c * wx * wy * wz).ravel(), minlength=out.size)
    return out.reshape(nx, ny, nz)


def gradient_phase(mu, xvec, yvec, zvec, rays, freqs, y, K=1e11, i0=0, rule=QUAD_SIMPSON_AVG, wrt_log_model=True):
    """d/d mu (or d/d ne) of sum y[a,t,d,l] g[a,t,d,l] for the phase observable (forward_phase;
    inversion/iterative_newton.py:86-127), y = dS/dg:
      g = const + a_l clock - (a_l / c) (phi - phi[i0]),  phi_l = sum_k c_k (1 - sqrt(1 - ne_k / n_p,l))
      d phi_l / d ne_v = sum_k c_k W_kv / (2 n_p,l sqrt(1 - ne_k / n_p,l))."""
    ne = np.exp(mu) * K
    ne_rays = trilinear(xvec, yvec, zvec, ne, rays[..., 0, :], rays[..., 1, :], rays[..., 2, :])
    cq = quadrature_weights(rays[..., 3, :], rule)
    q = np.zeros_like(ne_rays)
    for l, nu in enumerate(freqs):
        n_p = 1.2404e-2 * nu ** 2
        wl = -(2 * np.pi * nu / SPEED_OF_LIGHT) * differential_weights(y[..., l], i0)
        q += wl[..., None] * (0.5 / n_p) / np.sqrt(1.0 - ne_rays / n_p)
    g = scatter_trilinear(rays, xvec, yvec, zvec, cq * q)
    return g * ne if wrt_log_model else g


def differential_weights(w, i0):
    """Transpose of ``tec - tec[i0]``:  w_r -> w_r - [a(r) == i0] sum_a w[a,t,d]."""
    w = np.array(w, float)
    w[i0] -= w.sum(axis=0)
    return w


def gradient_log_model(rays, xvec, yvec, zvec, m, K_ne, i0, g, dobs, CdCt, rule=QUAD_SIMPSON_AVG):
    """dS/dm for S = 1/2 sum (g - dobs)^2 / (CdCt + 1e-15), g = forward_equation(...):
    G^T (differential dd) * ne   (exp applied at nodes => chain rule is a node-wise product;
    cf. inversion/gradient.py:19,77-81 and geometry/oct_trees/Inversion.py:736)."""
    dd = (g - dobs) / (CdCt + 1e-15)
    gt = adjoint_tec(rays, xvec, yvec, zvec, differential_weights(dd, i0), rule)
    return gt * ne_from_log_model(m, K_ne)


# --------------------------------------------------------------------------- A7 as shipped (chords)
def slab_chord(r0, n_unit, lo, hi):
    """Chord length of the ray r0 + t n through the box [lo, hi]
    (geometry/slab_method.py:19-58, incl. its ``tmax > 0`` rule)."""
    with np.errstate(divide='ignore', invalid='ignore'):
        t1 = (lo - r0) / n_unit
        t2 = (hi - r0) / n_unit
    t1 = np.where(np.isnan(t1), 0.0, t1)
    t2 = np.where(np.isnan(t2), 0.0, t2)
    t_enter = np.max(np.minimum(t1, t2))
    t_exit = np.min(np.maximum(t1, t2))
    if t_enter < t_exit and t_enter > 0:
        return np.linalg.norm(n_unit * (t_enter - t_exit))
    return 0.0


def bisection(array, value):
    """geometry/tri_cubic.py:105-132."""
    n = len(array)
    if value < array[0]:
        return -1
    if value > array[n - 1]:
        return n
    if value == array[n - 1]:
        return n - 1
    return int(np.clip(np.searchsorted(array, value, side='right') - 1, 0, n - 2))


def ray_dirac(rays, xvec, yvec, zvec):
    """geometry/ray_dirac.py:5-34: chord of the straight first->last segment through the 27
    voxel-centred boxes around every sample."""
    N1, N2, _, Ns = rays.shape
    dx, dy, dz = xvec[1] - xvec[0], yvec[1] - yvec[0], zvec[1] - zvec[0]
    half = np.array([dx, dy, dz]) / 2.0
    dirac = np.zeros((N1, N2, len(xvec), len(yvec), len(zvec)))
    for a in range(N1):
        for b in range(N2):
            r0 = rays[a, b, 0:3, 0]
            n = rays[a, b, 0:3, -1] - r0
            n = n / np.linalg.norm(n)
            for s in range(Ns):
                ci = bisection(xvec, rays[a, b, 0, s])
                cj = bisection(yvec, rays[a, b, 1, s])
                ck = bisection(zvec, rays[a, b, 2, s])
                for xi in range(max(0, ci - 1), min(len(xvec), ci + 2)):
                    for yi in range(max(0, cj - 1), min(len(yvec), cj + 2)):
                        for zi in range(max(0, ck - 1), min(len(zvec), ck + 2)):
                            c = np.array([xvec[xi], yvec[yi], zvec[zi]])
                            dirac[a, b, xi, yi, zi] = slab_chord(r0, n, c - half, c + half)
    return dirac


def gradient_chords(rays, xvec, yvec, zvec, M, dd):
    """inversion/gradient.py:15-20: einsum('ijklm,klm,ij->klm', dirac, M, dd)."""
    return np.einsum("ijklm,klm,ij->klm", ray_dirac(rays, xvec, yvec, zvec), M, dd)


# --------------------------------------------------------------------------- A4 Fermat
def ne_to_n(ne, frequency):
    """n = sqrt(1 - 8.980^2 ne / nu^2) at the nodes (inversion/fermat.py:36-46)."""
    return np.sqrt(1.0 + ne * (-PLASMA_CONST ** 2 / frequency ** 2))


def fermat_rhs(state, field, bend, type='z'):
    """d/dz of (px,py,pz,x,y,s) for type='z' (inversion/fermat.py:64-72;
    notebooks/FermatClass.ipynb c0:76-84): s' = n/pz, p' = grad(n) n/pz, x' = px/pz, y' = py/pz;
    d/ds for type='s' (fermat.py:74-82): s' = 1, p' = grad(n), (x,y,z)' = p/n.
    ``field(x,y,z) -> n, nx, ny, nz``; with bend=False the gradient is dropped, which is what
    the shipped code does (fermat.py:54-55)."""
    px, py, pz, x, y, z, s = state
    n, nx, ny, nz = field(x, y, z)
    if not bend:
        nx = ny = nz = np.zeros_like(n)
    if type == 's':
        rn = 1.0 / n
        return np.stack([nx, ny, nz, px * rn, py * rn, pz * rn, np.ones_like(pz)])
    f = n / pz
    return np.stack([nx * f, ny * f, nz * f, px / pz, py / pz, np.ones_like(pz), f])


def fermat_trace(origins, directions, tmax, N, field, bend=True, substeps=4, type='z'):
    """Fixed-step RK4 in z from z0 to tmax (type='z') or in arc length from 0 to tmax (type='s'),
    N output samples, ``substeps`` RK4 steps between outputs; vectorised over rays.  Returns
    rays[...,4,N] (x,y,z,s).  The GPU kernels perform the same arithmetic in the same order."""
    o = np.asarray(origins, float)
    d = np.asarray(directions, float)
    shp = o.shape[:-1]
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    p = d / np.linalg.norm(d, axis=-1, keepdims=True)
    st = np.stack([p[:, 0], p[:, 1], p[:, 2], o[:, 0], o[:, 1], o[:, 2], np.zeros(len(o))])
    h = (tmax - o[:, 2]) / ((N - 1) * substeps) if type == 'z' else np.full(len(o), tmax / ((N - 1) * substeps))
    rays = np.empty((len(o), 4, N))
    rays[:, 0, 0], rays[:, 1, 0], rays[:, 2, 0], rays[:, 3, 0] = st[3], st[4], st[5], st[6]
    for kout in range(1, N):
        for _ in range(substeps):
            k1 = fermat_rhs(st, field, bend, type)
            k2 = fermat_rhs(st + 0.5 * h * k1, field, bend, type)
            k3 = fermat_rhs(st + 0.5 * h * k2, field, bend, type)
            k4 = fermat_rhs(st + h * k3, field, bend, type)
            st = st + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
        rays[:, 0, kout], rays[:, 1, kout], rays[:, 2, kout], rays[:, 3, kout] = st[3], st[4], st[5], st[6]
    return rays.reshape(shp + (4, N))


def n_field_trilinear(xvec, yvec, zvec, nM):
    """n from the trilinear interpolant of the node values; gradient = the analytic gradient of
    that trilinear cell polynomial."""
    def field(x, y, z):
        i, tx = find_cell(xvec, x)
        j, ty = find_cell(yvec, y)
        k, tz = find_cell(zvec, z)
        hx, hy, hz = xvec[i + 1] - xvec[i], yvec[j + 1] - yvec[j], zvec[k + 1] - zvec[k]
        c = [[[nM[i + a, j + b, k + cc] for cc in (0, 1)] for b in (0, 1)] for a in (0, 1)]
        wx, wy, wz = (1 - tx, tx), (1 - ty, ty), (1 - tz, tz)
        sx, sy, sz = (-1.0, 1.0), (-1.0, 1.0), (-1.0, 1.0)
        n = gx = gy = gz = 0.0
        for a in (0, 1):
            for b in (0, 1):
                for cc in (0, 1):
                    v = c[a][b][cc]
                    n = n + v * wx[a] * wy[b] * wz[cc]
                    gx = gx + v * sx[a] * wy[b] * wz[cc]
                    gy = gy + v * wx[a] * sy[b] * wz[cc]
                    gz = gz + v * wx[a] * wy[b] * sz[cc]
        return n, gx / hx, gy / hy, gz / hz
    return field


def n_field_tricubic(xvec, yvec, zvec, nM):
    def field(x, y, z):
        return tricubic(xvec, yvec, zvec, nM, x, y, z, grad=True)
    return field


# --------------------------------------------------------------------------- C_m smoothing (section 8f #3)
def covariance_stencil_half_width(dx, dy, dz, l=20.0):
    """Half width h of the (2h+1)^3 stencil ionosphere/covariance.py:46-63 settles on: start at m = 5
    and grow by 2 while the corner value / centre value of the kernel exceeds 0.05.  The default kernel
    is the product of three exponential (Matern p = 0) factors exp(-|r_axis| / l)
    (ionosphere/covariance.py:22; utils/gaussian_process.py:440-467 with p = 0, sigma = 1)."""
    h = 2
    while np.exp(-h * (dx + dy + dz) / l) > 0.05:
        h += 1
    return h


def exp_kernel_1d(d, h, l=20.0):
    return np.exp(-np.abs(np.arange(-h, h + 1) * d) / l)


def smooth(phi, dx, dy, dz, l=20.0):
    """Covariance.smooth (ionosphere/covariance.py:383-385): scipy.ndimage.convolve(phi, c_stencil,
    mode='nearest').  The stencil is separable, so this is three 1-D convolutions with edge
    replication."""
    from scipy.ndimage import convolve1d
    h = covariance_stencil_half_width(dx, dy, dz, l)
    out = convolve1d(phi, exp_kernel_1d(dx, h, l), axis=0, mode='nearest')
    out = convolve1d(out, exp_kernel_1d(dy, h, l), axis=1, mode='nearest')
    return convolve1d(out, exp_kernel_1d(dz, h, l), axis=2, mode='nearest')
