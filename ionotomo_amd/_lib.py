"""ctypes binding of libionotomo_hip.so (include/ionotomo_hip.h).

There is NO CPU fallback: if the shared library is missing or no GPU is visible every
compute entry point raises.  (``oracle/`` holds a CPU restatement, but that is test
infrastructure and is never imported from here.)
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# IONOTOMO_LIB: an alternative build of the same sources (A/B builds with different -D tuning constants, the
# -DIONO_ABLATION timing build of profiles/tools) -- never a different implementation
LIB_PATH = os.environ.get("IONOTOMO_LIB") or os.path.join(_HERE, "libionotomo_hip.so")

OK, ERR_OOB, ERR_NONFINITE, ERR_SHAPE, ERR_HIP, ERR_ARG = 0, -1, -2, -3, -4, -5
WALK_FORWARD, WALK_ADJOINT = 0, 1          # iono_walk_cycles / iono_walk_partition_set
NPART = 512                                # IONO_NPART: partial sums per dot-producing pass
F64, F32 = 0, 1
COMM_ID_BYTES = 128                        # IONO_COMM_ID_BYTES
INTERP_TRILINEAR, INTERP_TRICUBIC = 0, 1
RAY_Z, RAY_S = 0, 1                        # independent variable of a ray: Fermat(type='z' | 's')
QUAD_SIMPSON_AVG, QUAD_SIMPSON_SCIPY, QUAD_TRAPEZOID = 0, 1, 2

_INTERP = {"linear": 0, "trilinear": 0, 0: 0, "cubic": 1, "tricubic": 1, 1: 1}
_QUAD = {"avg": 0, "simpson": 0, "simps": 0, 0: 0, "scipy": 1, "cartwright": 1, 1: 1, "trapz": 2, "trapezoid": 2, 2: 2}
_STORAGE = {"f64": 0, "float64": 0, np.float64: 0, 0: 0, "f32": 1, "float32": 1, np.float32: 1, 1: 1}


def interp_kind(k):
    return _INTERP[k]


def ray_type(t):
    return {"z": RAY_Z, "s": RAY_S, RAY_Z: RAY_Z, RAY_S: RAY_S}[t]


def quad_rule(q):
    return _QUAD[q]


def storage_code(s):
    return _STORAGE[s]


# the operations of the dispatch table (include/ionotomo_hip.h: IONO_OP_*)
OP_FORWARD, OP_ADJOINT, OP_TRACE, OP_FERMAT_FORWARD, OP_FERMAT_ADJOINT, OP_PHASE_FORWARD, OP_PHASE_ADJOINT = range(7)
_OPS = {"forward": 0, "adjoint": 1, "trace": 2, "fermat_forward": 3, "fermat_adjoint": 4, "phase_forward": 5, "phase_adjoint": 6}


class DispatchFacts(ctypes.Structure):
    """iono_dispatch_facts: what decides which kernel a launch gets (the header documents every field)."""
    _fields_ = [("storage", ctypes.c_int), ("tier", ctypes.c_int), ("cubic_fast", ctypes.c_int), ("cubic_records", ctypes.c_int),
                ("ideal_axes", ctypes.c_int), ("q4_ok", ctypes.c_int), ("variant", ctypes.c_int), ("deterministic", ctypes.c_int),
                ("interp_kind", ctypes.c_int), ("ne_kind", ctypes.c_int), ("bend", ctypes.c_int), ("Ns", ctypes.c_int),
                ("R", ctypes.c_int64), ("fwd_bundles", ctypes.c_int64), ("fwd_tail", ctypes.c_int64), ("adj_planned", ctypes.c_int),
                ("adj_tiles", ctypes.c_int), ("adj_seg_lanes", ctypes.c_int), ("axes_bytes", ctypes.c_int), ("fermat_lm_lanes", ctypes.c_int),
                ("fermat_lm_few_min", ctypes.c_int64), ("fermat_poly_max", ctypes.c_int64), ("fermat_lin4_max", ctypes.c_int64),
                ("fermat_coop_max", ctypes.c_int64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def dispatch_name(op, **facts):
    """The kernel(s) operation ``op`` ("forward", "adjoint", "trace", "fermat_forward", "fermat_adjoint", "phase_forward",
    "phase_adjoint") gets under ``facts`` (fields of ``DispatchFacts``; unnamed ones are 0): the library's ONE dispatch table, asked
    without a GPU (iono_dispatch_name is pure host code)."""
    f = DispatchFacts()
    for k, v in facts.items():
        if k not in dict(DispatchFacts._fields_):
            raise KeyError(k)
        setattr(f, k, int(v))
    buf = ctypes.create_string_buffer(256)
    rc = load().iono_dispatch_name(ctypes.byref(f), _OPS.get(op, op), buf, 256)
    if rc != OK:
        raise ValueError("iono_dispatch_name: bad operation %r" % (op,))
    return buf.value.decode()


c_double_p = ctypes.POINTER(ctypes.c_double)
_P, _I, _L, _D, _V = c_double_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p

# name -> argument types after the leading ctx pointer (None = no ctx argument)
_SIGNATURES = {
    "iono_ctx_destroy": [],
    "iono_ctx_set_stream": [_V],
    "iono_ctx_use_own_stream": [],
    "iono_ctx_synchronize": [],
    "iono_grid_set": [_P, _I, _P, _I, _P, _I, _P, _I],
    "iono_grid_set_values": [_P],
    "iono_grid_get_values": [_P],
    "iono_grid_set_values_dev": [_V],
    "iono_grid_set_exp": [_P, _D],
    "iono_grid_set_exp_dev": [_V, _D],
    "iono_interp": [_P, _P, _P, _L, _I, _I, _P],
    "iono_trace_straight": [_P, _P, _L, _D, _I, _I, _P],
    "iono_trace_fermat": [_P, _P, _L, _D, _I, _D, _I, _I, _I, _I, _P],
    "iono_forward_tec_straight": [_P, _P, _L, _D, _I, _I, _I, _P],
    "iono_forward_tec_rays": [_P, _L, _I, _I, _I, _P],
    "iono_subtract_reference": [_P, _I, _L, _I],
    "iono_forward_phase_rays": [_P, _I, _I, _I, _I, _P, _I, _P, _P, _I, _I, _P],
    "iono_adjoint_straight": [_P, _P, _P, _L, _D, _I, _I, _I, _I, _P],
    "iono_adjoint_rays": [_P, _P, _L, _I, _I, _I, _I, _P],
    "iono_gradient_chords": [_P, _P, _L, _I, _P],
    "iono_gradient_chords_dev": [_V, _V, _L, _I, _V],
    "iono_forward_tec_straight_dev": [_V, _V, _V, _L, _D, _I, _I, _I, _V],
    "iono_forward_tec_rays_dev": [_V, _L, _I, _I, _I, _V],
    "iono_adjoint_straight_dev": [_V, _V, _V, _V, _L, _D, _I, _I, _I, _V, _I],
    "iono_adjoint_rays_dev": [_V, _V, _L, _I, _I, _I, _V, _I],
    "iono_adjoint_residual_straight_dev": [_V, _V, _V, _V, _V, _V, _I, _L, _I, _D, _I, _I, _I, _V, _I],
    "iono_adjoint_differential_straight_dev": [_V, _V, _V, _V, _V, _I, _L, _I, _D, _I, _I, _I, _V, _I],
    "iono_adjoint_cg_step_dev": [_V, _V, _V, _V, _V, _V, _I, _V, _I, _V, _I, _L, _I, _D, _I, _I, _I, _V, _V, _I],
    "iono_adjoint_sirt_step_dev": [_V, _V, _V, _V, _V, _V, _V, _I, _L, _I, _D, _I, _I, _I, _V, _V, _V, _I],
    "iono_adjoint_plan_slabs": [_I],
    "iono_set_deterministic": [_I],
    "iono_adjoint_plan_slab_info": [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)],
    "iono_adjoint_unit_range": [_I, _I],
    "iono_adjoint_planned_weights_dev": [_V, _V, _V, _L, _D, _I, _I, _I, _V, _I],
    "iono_subtract_reference_dev": [_V, _I, _L, _I],
    "iono_vec_axpby_dev": [_V, _V, _L, _V, _V, _D, _V, _V],
    "iono_forward_phase_straight_dev": [_V, _V, _I, _I, _I, _D, _I, _P, _I, _V, _V, _I, _I, _V, _V],
    "iono_adjoint_phase_straight_dev": [_V, _V, _V, _V, _I, _L, _D, _I, _P, _I, _I, _I, _V, _I, _V],
    "iono_grid_padded_size": [ctypes.POINTER(ctypes.c_int64)],
    "iono_grid_bind_values_dev": [_V],
    "iono_grid_values_changed": [],
    "iono_rays_combine_dev": [_V, _V, _V, _V, _I, _L, _I, _D, _D, _V, _V],
    "iono_vec_axpby_dot_dev": [_V, _V, _L, _V, _I, _V, _I, _D, _V, _I, _V, _I, _V],
    "iono_small_ray_pass_dev": [_I, _V, _V, _V, _V, _V, _V, _I, _L, _I, _V, _I, _V, _V, _V],
    "iono_compact_gather_dev": [_V, _V, _L, _V, _I, _V],
    "iono_compact_scatter_dev": [_V, _V, _L, _V],
    "iono_compact_cg_update_dev": [_V, _V, _V, _V, _L, _V, _V, _I, _V, _I, _V, _I, _V, _I],
    "iono_compact_sirt_update_dev": [_V, _V, _V, _V, _L, _V, _D, _I, _V],
    "iono_walk_cycles": [_I, _V, _I, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)],
    "iono_walk_partition_set": [_I, _V, _I, _L],
    "iono_adjoint_plan_dev": [_V, _V, _L, _D, _I, _I],
    "iono_adjoint_plan_clear": [],
    "iono_adjoint_plan_info": [ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)],
    "iono_adjoint_plan_segment_lanes": [ctypes.POINTER(ctypes.c_int)],
    "iono_dev_alloc": [ctypes.c_size_t, ctypes.POINTER(_V)],
    "iono_dev_free": [_V],
    "iono_dev_upload": [_V, _V, ctypes.c_size_t],
    "iono_dev_zero": [_V, ctypes.c_size_t],
    "iono_dev_download": [_V, _V, ctypes.c_size_t],
    "iono_scale_by_grid_dev": [_V],
    "iono_forward_plan_dev": [_V, _V, _L, _D, _I],
    "iono_forward_plan_clear": [],
    "iono_forward_plan_info": [ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)],
    "iono_forward_plan_split": [ctypes.POINTER(ctypes.c_int64)] * 4 + [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64),
                                ctypes.POINTER(ctypes.c_double)],
    "iono_dispatch_describe": [_I, _V, _V, _L, _D, _I, _I, _I, _I, ctypes.POINTER(DispatchFacts), ctypes.c_char_p, _I],
    "iono_walk_order": [_P, _P, _L, _D, ctypes.POINTER(ctypes.c_int)],
    "iono_trace_fermat_dev": [_V, _V, _L, _D, _I, _D, _I, _I, _I, _I, _V],
    "iono_forward_tec_fermat_dev": [_V, _V, _L, _D, _I, _D, _I, _I, _I, _I, _I, _I, _D, _V],
    "iono_adjoint_fermat_dev": [_V, _V, _V, _L, _D, _I, _D, _I, _I, _I, _I, _I, _I, _D, _V],
    "iono_fermat_lm_ok": [_I, _I, _L, _I, _I, ctypes.POINTER(ctypes.c_int)],
    "iono_check_oob": [ctypes.POINTER(ctypes.c_int)],
    "iono_plan_stale": [ctypes.POINTER(ctypes.c_int)],
    "iono_smooth_separable": [_P, _P, _P, _P, _P, _I],
    "iono_smooth_separable_dev": [_V, _V, _V, _P, _P, _P, _I],
    "iono_comm_unique_id": [ctypes.c_char_p],
    "iono_comm_init": [ctypes.c_char_p, _I, _I],
    "iono_comm_allreduce_dev": [_V, _L, _I],
    "iono_comm_destroy": [],
}
EXPORTED = sorted(list(_SIGNATURES) + ["iono_ctx_create", "iono_last_error", "iono_version", "iono_grid_values_ptr", "iono_dispatch_name"])

_lib = None


def load():
    """dlopen the library (once).  Raises if it has not been built -- by design."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "ionotomo_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # PyTorch's ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 with the SAME sonames as
    # /opt/rocm's, so whichever is loaded first serves every later user in the process.  If ours pulled in
    # the system runtime first, a later `import torch` would find the GPU "unavailable"; loading torch
    # first gives one runtime for both (kernels then run on torch-allocated memory without surprises).
    try:
        import torch  # noqa: F401
    except Exception:                                   # torch is plumbing only: the facade works without it
        pass
    lib = ctypes.CDLL(LIB_PATH)
    lib.iono_ctx_create.argtypes = [_I, ctypes.POINTER(_V)]
    lib.iono_ctx_create.restype = _I
    lib.iono_last_error.argtypes = [_V]
    lib.iono_last_error.restype = ctypes.c_char_p
    lib.iono_version.argtypes = []
    lib.iono_version.restype = _I
    lib.iono_grid_values_ptr.argtypes = [_V]
    lib.iono_grid_values_ptr.restype = _V
    lib.iono_dispatch_name.argtypes = [ctypes.POINTER(DispatchFacts), _I, ctypes.c_char_p, _I]
    lib.iono_dispatch_name.restype = _I
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = [_V] + args
        fn.restype = _I
    _lib = lib
    return lib


def _dp(a):
    """float64 C-contiguous host array -> double*"""
    return a.ctypes.data_as(c_double_p)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Context(object):
    """One GPU context (``iono_ctx``).  Create after fork; not thread-safe."""

    def __init__(self, device=0):
        self._lib = load()
        h = _V()
        rc = self._lib.iono_ctx_create(int(device), ctypes.byref(h))
        if rc != OK:
            raise RuntimeError("iono_ctx_create failed: %s" % self._lib.iono_last_error(None).decode())
        self._h = h
        self.device = int(device)
        self.pid = os.getpid()
        self.grid_shape = None
        self.storage = None
        self._axes = (None, None, None)
        self._dev_arrays = []          # facade cache: [(weakref to a host array, shape, device pointer, bytes)], most recent first
        self._values_key = None        # facade cache: what the device grid values were computed from
        self._scratch = {}

    def close(self):
        if getattr(self, "_h", None) is not None and self.pid == os.getpid():
            for _, _, ptr, _ in getattr(self, "_dev_arrays", []):
                self._lib.iono_dev_free(self._h, ptr)
            for ptr, _ in getattr(self, "_scratch", {}).values():
                self._lib.iono_dev_free(self._h, ptr)
            self._dev_arrays, self._scratch = [], {}
            self._lib.iono_ctx_destroy(self._h)
        self._h = None

    # -- facade cache (the reference API hands over host arrays; a line search hands over the SAME ones again and again) ----
    def dev_alloc(self, nbytes):
        p = _V()
        self.call("iono_dev_alloc", int(nbytes), ctypes.byref(p))
        return _V(p.value)

    def staged(self, name, a):
        """Device pointer of a FRESH copy of the float64 C-contiguous host array ``a`` in the named grow-only buffer: uploaded on
        every call, so whatever the caller did to ``a`` since the last call (in place or not) is seen.  This is what the
        reference-signature facade does by default."""
        ptr = self.scratch(name, a.nbytes)
        self.call("iono_dev_upload", ptr, _V(a.ctypes.data), a.nbytes)
        return ptr

    def resident(self, a, keep=3, max_bytes=4 << 30):
        """OPT-IN (``assume_unchanged=True`` of the facade functions): device pointer of a copy of the float64 host array ``a``,
        uploaded once per OBJECT.  The key is the array's identity (a weak reference) and shape ONLY -- no content check of any
        kind: an in-place edit of ``a`` after the first call is NOT seen, the caller promises there is none (or calls
        ``forget``).  The ``keep`` most recently used arrays stay resident."""
        import weakref
        for i, (ref, shape, ptr, nb) in enumerate(self._dev_arrays):
            if ref() is a and shape == a.shape:
                if i:
                    self._dev_arrays.insert(0, self._dev_arrays.pop(i))
                return ptr
        if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]) or a.nbytes > max_bytes:
            return None
        ptr = self.dev_alloc(a.nbytes)
        self.call("iono_dev_upload", ptr, _V(a.ctypes.data), a.nbytes)
        self._dev_arrays.insert(0, (weakref.ref(a), a.shape, ptr, a.nbytes))
        live = []
        for j, ent in enumerate(self._dev_arrays):
            if j < keep and ent[0]() is not None:
                live.append(ent)
            else:
                self.call("iono_dev_free", ent[2])
        self._dev_arrays = live
        return ptr

    def forget(self, a=None):
        """Drop the resident copy of ``a`` (all of them with no argument) and the record of what the grid values came from."""
        keep = []
        for ent in self._dev_arrays:
            if a is None or ent[0]() is a or ent[0]() is None:
                self.call("iono_dev_free", ent[2])
            else:
                keep.append(ent)
        self._dev_arrays = keep
        self._values_key = None

    def scratch(self, name, nbytes):
        """A named grow-only device buffer (outputs of the resident facade path)."""
        ptr, cap = self._scratch.get(name, (None, 0))
        if cap < nbytes:
            if ptr is not None:
                self.call("iono_dev_free", ptr)
            ptr, cap = self.dev_alloc(nbytes + nbytes // 4 + 64), nbytes + nbytes // 4 + 64
            self._scratch[name] = (ptr, cap)
        return ptr

    @staticmethod
    def _fingerprint(M):
        """SAMPLED content check of a grid-sized array, used ONLY on the opt-in path (``assume_unchanged=True``) as a safety net
        for whole-array in-place updates (``m_tci.M += step``): three contiguous runs of 64 values (start, middle, end) + 128
        evenly spaced ones, ~3 us.  An in-place change confined to nodes between the probes (one perturbed node of a
        finite-difference loop) is NOT seen: that is why the default path never consults it."""
        f = M.reshape(-1)
        n = f.size
        mid = n >> 1
        return (float(f[:64].sum()), float(f[mid:mid + 64].sum()), float(f[-64:].sum()), float(f[::max(1, n // 128)].sum()), n)

    def set_values_exp_cached(self, M, scale):
        """OPT-IN path: grid values <- scale * exp(M) unless they already are (the same LIVE object, same sampled fingerprint,
        same scale).  Not exact against single-node in-place edits (see ``_fingerprint``); the default facade path calls
        ``set_values_exp`` every time."""
        import weakref
        src = M
        M = as_f64(M)
        # identity = a WEAK REFERENCE to the caller's array: id() and the data address are reused by the allocator as soon as an array
        # dies (a finite difference builds m + e, drops it, builds m - e at the same address: one node apart, between the probes)
        key = (M.shape, float(scale), self._fingerprint(M), self.grid_shape, self.storage)
        if self._values_key is not None and self._values_key[0]() is src and M is src and self._values_key[1] == key:
            return
        self.set_values_exp(M, scale)
        try:
            self._values_key = (weakref.ref(src), key) if M is src else None       # (a converted copy is never the same object twice)
        except TypeError:
            self._values_key = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- error mapping: mirrors what the reference raises --------------------------
    def _check(self, rc):
        if rc == OK:
            return
        msg = self._lib.iono_last_error(self._h).decode()
        if rc == ERR_OOB:
            raise ValueError(msg)                 # scipy RGI bounds_error=True
        if rc == ERR_NONFINITE:
            raise AssertionError(msg)             # geometry/tri_cubic.py:51
        if rc in (ERR_SHAPE, ERR_ARG):
            raise ValueError(msg)
        raise RuntimeError(msg)

    def call(self, name, *args):
        self._check(getattr(self._lib, name)(self._h, *args))

    def dispatch_describe(self, op, origins_ptr=0, dirs_ptr=0, R=0, tmax=0.0, Ns=2, kind="linear", ne_kind=None, bend=True):
        """(kernel name, facts) of the launch ``op`` would be on this context with these arguments (device pointers 0: no plan
        matches): the library's dispatch table (include/ionotomo_hip.h: iono_dispatch_describe)."""
        f = DispatchFacts()
        buf = ctypes.create_string_buffer(256)
        self.call("iono_dispatch_describe", _OPS.get(op, op), _V(origins_ptr), _V(dirs_ptr), int(R), float(tmax), int(Ns), interp_kind(kind),
                  interp_kind(kind if ne_kind is None else ne_kind), int(bool(bend)), ctypes.byref(f), buf, 256)
        return buf.value.decode(), f.as_dict()

    # -- grid ------------------------------------------------------------------------------------
    def set_grid(self, xvec, yvec, zvec, M=None, storage="f64"):
        xv, yv, zv = as_f64(xvec).ravel(), as_f64(yvec).ravel(), as_f64(zvec).ravel()
        Mp = None
        if M is not None:
            M = as_f64(M)
            if M.size != xv.size * yv.size * zv.size:
                raise ValueError("M has %d values, grid has %d nodes" % (M.size, xv.size * yv.size * zv.size))
            Mp = _dp(M)
        same = (self.grid_shape == (xv.size, yv.size, zv.size) and self.storage == storage_code(storage)
                and all(np.array_equal(a, b) for a, b in zip(self._axes, (xv, yv, zv))))
        if same:                      # same axes: keep the device allocation, refresh the values only
            if Mp is not None:
                self._values_key = None
                self.call("iono_grid_set_values", Mp)
            return
        self._values_key = None
        self.call("iono_grid_set", _dp(xv), xv.size, _dp(yv), yv.size, _dp(zv), zv.size, Mp, storage_code(storage))
        self.grid_shape = (xv.size, yv.size, zv.size)
        self.storage = storage_code(storage)
        self._axes = (xv.copy(), yv.copy(), zv.copy())

    def set_values(self, M):
        M = as_f64(M)
        self._need(M.size)
        self._values_key = None
        self.call("iono_grid_set_values", _dp(M))

    def set_values_exp(self, m, scale):
        m = as_f64(m)
        self._need(m.size)
        self._values_key = None
        self.call("iono_grid_set_exp", _dp(m), float(scale))

    def get_values(self):
        out = np.empty(self.grid_shape, dtype=np.float64)
        self.call("iono_grid_get_values", _dp(out))
        return out

    def _need(self, size):
        if self.grid_shape is None:
            raise ValueError("no grid set")
        if size != int(np.prod(self.grid_shape)):
            raise ValueError("expected %d grid values, got %d" % (int(np.prod(self.grid_shape)), size))

    def values_ptr(self):
        return self._lib.iono_grid_values_ptr(self._h)

    def set_stream(self, stream_ptr):
        self.call("iono_ctx_set_stream", _V(stream_ptr))

    def synchronize(self):
        self.call("iono_ctx_synchronize")

    def check_oob(self):
        v = ctypes.c_int(0)
        self.call("iono_check_oob", ctypes.byref(v))
        return bool(v.value)

    def plan_stale(self):
        """True (and the flag is cleared) if a planned launch since the last call met rays its plan was not made for: a planned
        array was edited in place (include/ionotomo_hip.h:iono_plan_stale).  Synchronises."""
        v = ctypes.c_int(0)
        self.call("iono_plan_stale", ctypes.byref(v))
        return bool(v.value)

    # -- measured load balance of the chunked kernels (include/ionotomo_hip.h) -----------------------------
    def walk_cycles(self, which):
        """(cycles per chunk in walk order, resident waves / workgroups) of the last forward (WALK_FORWARD) or
        tiled-adjoint (WALK_ADJOINT) launch; empty if there was none."""
        n, units = ctypes.c_int(0), ctypes.c_int(0)
        self.call("iono_walk_cycles", int(which), None, 0, ctypes.byref(n), ctypes.byref(units))
        out = np.zeros(n.value, dtype=np.uint64)
        if n.value:
            self.call("iono_walk_cycles", int(which), out.ctypes.data_as(ctypes.c_void_p), n.value, ctypes.byref(n),
                      ctypes.byref(units))
        return out, units.value

    def walk_order(self, origins, directions, tmax):
        """int32 permutation: Morton walk order of the rays (speed only; include/ionotomo_hip.h:iono_walk_order)."""
        o, d, R = _rays_in(origins, directions)
        out = np.empty(R, dtype=np.int32)
        self.call("iono_walk_order", _dp(o), _dp(d), R, float(tmax), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        return out

    def walk_partition_set(self, which, starts, R):
        if starts is None:
            self.call("iono_walk_partition_set", int(which), None, 0, int(R))
            return
        st = np.ascontiguousarray(starts, dtype=np.int64)
        self.call("iono_walk_partition_set", int(which), st.ctypes.data_as(ctypes.c_void_p), st.size - 1, int(R))

    # -- host-pointer numerics -----------------------------------------------------------------------
    def interp(self, x, y, z, kind="linear", extrapolate=False):
        shp = np.shape(x)
        x, y, z = as_f64(x).ravel(), as_f64(y).ravel(), as_f64(z).ravel()
        if not (x.size == y.size == z.size):
            raise ValueError("x, y, z must have equal shapes")
        out = np.empty(x.size, dtype=np.float64)
        self.call("iono_interp", _dp(x), _dp(y), _dp(z), x.size, interp_kind(kind), int(bool(extrapolate)), _dp(out))
        return out.reshape(shp)

    def trace_straight(self, origins, directions, tmax, Ns, type="z"):
        o, d, R = _rays_in(origins, directions)
        out = np.empty((R, 4, int(Ns)), dtype=np.float64)
        self.call("iono_trace_straight", _dp(o), _dp(d), R, float(tmax), int(Ns), ray_type(type), _dp(out))
        return out

    def trace_fermat(self, origins, directions, tmax, Ns, frequency, bend=False, kind="linear", substeps=4, type="z"):
        o, d, R = _rays_in(origins, directions)
        out = np.empty((R, 4, int(Ns)), dtype=np.float64)
        self.call("iono_trace_fermat", _dp(o), _dp(d), R, float(tmax), int(Ns), float(frequency), int(bool(bend)),
                  interp_kind(kind), int(substeps), ray_type(type), _dp(out))
        return out

    def forward_tec_straight(self, origins, directions, tmax, Ns, kind="linear", rule="avg"):
        o, d, R = _rays_in(origins, directions)
        out = np.empty(R, dtype=np.float64)
        self.call("iono_forward_tec_straight", _dp(o), _dp(d), R, float(tmax), int(Ns), interp_kind(kind),
                  quad_rule(rule), _dp(out))
        return out

    def forward_tec_rays(self, rays, kind="linear", rule="avg"):
        rays = as_f64(rays)
        Ns = rays.shape[-1]
        if rays.ndim < 2 or rays.shape[-2] != 4:
            raise ValueError("rays must have shape [..., 4, Ns]")
        R = int(np.prod(rays.shape[:-2], dtype=np.int64))
        out = np.empty(R, dtype=np.float64)
        self.call("iono_forward_tec_rays", _dp(rays), R, int(Ns), interp_kind(kind), quad_rule(rule), _dp(out))
        return out.reshape(rays.shape[:-2])

    def forward_phase_rays(self, rays, freqs, clock, const, i0, rule="avg"):
        rays = as_f64(rays)
        Na, Nt, Nd, four, Ns = rays.shape
        freqs, clock, const = as_f64(freqs), as_f64(clock), as_f64(const)
        if clock.shape != (Na, Nt) or const.shape != (Na,):
            raise ValueError("clock must be [Na,Nt] and const [Na]")
        out = np.empty((Na, Nt, Nd, freqs.size), dtype=np.float64)
        self.call("iono_forward_phase_rays", _dp(rays), Na, Nt, Nd, Ns, _dp(freqs), freqs.size, _dp(clock), _dp(const),
                  int(i0), quad_rule(rule), _dp(out))
        return out

    def adjoint_straight(self, origins, directions, w, tmax, Ns, rule="avg", scale_by_grid=False, kind="linear"):
        o, d, R = _rays_in(origins, directions)
        w = as_f64(w).ravel()
        if w.size != R:
            raise ValueError("one weight per ray expected")
        out = np.empty(self.grid_shape, dtype=np.float64)
        self.call("iono_adjoint_straight", _dp(o), _dp(d), _dp(w), R, float(tmax), int(Ns), interp_kind(kind), quad_rule(rule),
                  int(bool(scale_by_grid)), _dp(out))
        return out

    def smooth_separable(self, phi, kx, ky, kz):
        phi = as_f64(phi)
        self._need(phi.size)
        kx, ky, kz, h = _smooth_args(kx, ky, kz)
        out = np.empty(self.grid_shape, dtype=np.float64)
        self.call("iono_smooth_separable", _dp(phi), _dp(out), _dp(kx), _dp(ky), _dp(kz), h)
        return out

    # ---- RCCL collective behind the C-ABI (for hosts without torch.distributed; include/ionotomo_hip.h) ----
    def comm_unique_id(self):
        """128 opaque bytes: rank 0 creates them and ships them to every other rank over the host's own channel."""
        buf = ctypes.create_string_buffer(COMM_ID_BYTES)
        self.call("iono_comm_unique_id", buf)
        return buf.raw

    def comm_init(self, comm_id, rank, nranks):
        assert len(comm_id) == COMM_ID_BYTES
        self.call("iono_comm_init", ctypes.create_string_buffer(bytes(comm_id), COMM_ID_BYTES), int(rank), int(nranks))

    def comm_allreduce_dev(self, ptr, count, dtype=F64):
        """In-place sum over ranks of `count` elements at device pointer `ptr`, enqueued on the ctx stream."""
        self.call("iono_comm_allreduce_dev", ctypes.c_void_p(int(ptr)), int(count), int(dtype))

    def comm_destroy(self):
        self.call("iono_comm_destroy")

    def gradient_chords(self, rays, dd):
        """The reference's shipped chord-length gradient einsum(dirac, M, dd) (inversion/gradient.py:15-20) for
        rays[..., 4, Ns] and dd[...]: NOT the transpose of the forward (include/ionotomo_hip.h)."""
        rays = as_f64(rays)
        Ns = rays.shape[-1]
        R = int(np.prod(rays.shape[:-2], dtype=np.int64))
        dd = as_f64(dd).ravel()
        if dd.size != R:
            raise ValueError("one weight per ray expected")
        out = np.empty(self.grid_shape, dtype=np.float64)
        self.call("iono_gradient_chords", _dp(rays), _dp(dd), R, int(Ns), _dp(out))
        return out

    def adjoint_rays(self, rays, w, rule="avg", scale_by_grid=False, kind="linear"):
        rays = as_f64(rays)
        Ns = rays.shape[-1]
        R = int(np.prod(rays.shape[:-2], dtype=np.int64))
        w = as_f64(w).ravel()
        if w.size != R:
            raise ValueError("one weight per ray expected")
        out = np.empty(self.grid_shape, dtype=np.float64)
        self.call("iono_adjoint_rays", _dp(rays), _dp(w), R, int(Ns), interp_kind(kind), quad_rule(rule),
                  int(bool(scale_by_grid)), _dp(out))
        return out


def _smooth_args(kx, ky, kz):
    kx, ky, kz = as_f64(kx).ravel(), as_f64(ky).ravel(), as_f64(kz).ravel()
    if not (kx.size == ky.size == kz.size) or kx.size % 2 != 1:
        raise ValueError("the three 1-D kernels must have the same odd length")
    return kx, ky, kz, kx.size // 2


def _rays_in(origins, directions):
    o, d = as_f64(origins), as_f64(directions)
    if o.shape != d.shape or o.shape[-1] != 3:
        raise ValueError("origins and directions must both have shape [..., 3]")
    return o, d, int(np.prod(o.shape[:-1], dtype=np.int64))


_default = None


def default_context():
    """Process-wide context on device LOCAL_RANK (or 0); re-created after fork."""
    global _default
    if _default is None or _default.pid != os.getpid():
        _default = Context(int(os.environ.get("IONOTOMO_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    return _default
