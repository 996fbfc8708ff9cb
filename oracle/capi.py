"""ORACLE (test infrastructure): ctypes loader for oracle/_build/liboracle.so (plain C, fp64).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess
from ctypes import c_double, c_int, c_long, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, "c", f) for f in os.listdir(os.path.join(_HERE, "c"))]
    stale = force or not os.path.exists(LIB) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(c_void_p)


def mlpg(features, variances, dim, col0=0):
    """MLPG.generation(features[:, col0:col0+3*dim], diag=variances, dim) -> [T, dim] f64."""
    f = np.ascontiguousarray(features, dtype=np.float64)
    v = np.ascontiguousarray(variances, dtype=np.float64)
    assert v.shape == (3 * dim,)
    T = f.shape[0]
    out = np.zeros((T, dim))
    fn = lib().orc_mlpg
    fn.restype = c_int
    fn.argtypes = [c_void_p, c_long, c_long, c_int, c_int, c_void_p, c_void_p, c_long, c_int]
    rc = fn(_p(f), T, f.shape[1], col0, dim, _p(v), _p(out), dim, 0)
    if rc != 0:
        raise RuntimeError("orc_mlpg failed: {}".format(rc))
    return out
