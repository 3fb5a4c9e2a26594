"""ctypes wrapper of oracle_c.c (TEST INFRASTRUCTURE: multi-core CPU baseline + second checker)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle_c.so")            # portable build (__graft_entry__.build), travels with the snapshot


def _native_lib():
    """A -march=native build for THIS host, compiled on first use (the timed CPU baseline must be tuned for the box it
    runs on, not for the build container); falls back to the portable library if no compiler is there."""
    import hashlib
    import subprocess
    try:
        model = [l for l in open("/proc/cpuinfo") if l.startswith(("model name", "flags"))][:2]
    except OSError:
        model = []
    tag = hashlib.md5("".join(model).encode()).hexdigest()[:10]
    d = os.path.join(_HERE, "_native")
    path = os.path.join(d, "liboracle_c.%s.so" % tag)
    src = os.path.join(_HERE, "oracle_c.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        try:
            os.makedirs(d, exist_ok=True)
            subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-o", path + ".tmp%d" % os.getpid(),
                                   src, "-lm"])
            os.replace(path + ".tmp%d" % os.getpid(), path)
        except Exception:
            return None
    return path
_P = ctypes.POINTER(ctypes.c_double)
_lib = None


def load():
    global _lib
    if _lib is None:
        path = _native_lib()
        if path is None:
            if not os.path.exists(LIB):
                import subprocess
                subprocess.check_call(["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-o", LIB, os.path.join(_HERE, "oracle_c.c"), "-lm"])
            path = LIB
        lib = ctypes.CDLL(path)
        lib.oracle_forward_tec_straight.restype = ctypes.c_int64
        lib.oracle_forward_tec_straight.argtypes = [_P, ctypes.c_int, _P, ctypes.c_int, _P, ctypes.c_int, _P, _P, _P,
                                                    ctypes.c_int64, ctypes.c_double, ctypes.c_int, _P, ctypes.c_int]
        lib.oracle_adjoint_straight.restype = None
        lib.oracle_adjoint_straight.argtypes = [_P, ctypes.c_int, _P, ctypes.c_int, _P, ctypes.c_int, _P, _P, _P,
                                                ctypes.c_int64, ctypes.c_double, ctypes.c_int, _P]
        lib.oracle_num_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(_P)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def num_threads():
    return load().oracle_num_threads()


def forward_tec_straight(xvec, yvec, zvec, M, origins, directions, tmax, Ns, nthreads=0):
    lib = load()
    xv, yv, zv, M, o, d = _c(xvec), _c(yvec), _c(zvec), _c(M), _c(origins), _c(directions)
    R = o.size // 3
    tec = np.empty(R)
    oob = lib.oracle_forward_tec_straight(_p(xv), xv.size, _p(yv), yv.size, _p(zv), zv.size, _p(M), _p(o), _p(d), R,
                                          float(tmax), int(Ns), _p(tec), int(nthreads))
    if oob:
        raise ValueError("One of the requested xi is out of bounds")
    return tec.reshape(o.shape[:-1])


def adjoint_straight(xvec, yvec, zvec, origins, directions, w, tmax, Ns):
    lib = load()
    xv, yv, zv, o, d, w = _c(xvec), _c(yvec), _c(zvec), _c(origins), _c(directions), _c(w)
    grad = np.zeros((xv.size, yv.size, zv.size))
    lib.oracle_adjoint_straight(_p(xv), xv.size, _p(yv), yv.size, _p(zv), zv.size, _p(o), _p(d), _p(w), o.size // 3,
                                float(tmax), int(Ns), _p(grad))
    return grad
