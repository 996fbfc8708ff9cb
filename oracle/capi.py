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


# ---------------------------------------------------------------------------------- WORLD / SPTK
def _fn(name, restype, argtypes):
    f = getattr(lib(), name)
    f.restype = restype
    f.argtypes = argtypes
    return f


def num_frames(n, fs, frame_period=5.0):
    return int(1000.0 * n / fs / frame_period) + 1


def cheaptrick_fft_size(fs, f0_floor=71.0):
    import math
    return 2 ** (1 + int(math.log2(3.0 * fs / f0_floor + 1)))


def dio(x, fs, frame_period=5.0):
    x = np.ascontiguousarray(x, dtype=np.float64)
    T = num_frames(len(x), fs, frame_period)
    f0 = np.zeros(T)
    tp = np.zeros(T)
    fn = _fn("orc_dio", c_int, [c_void_p, c_int, c_int, c_double, c_double, c_double, c_double,
                                c_double, c_void_p, c_void_p])
    rc = fn(_p(x), len(x), fs, frame_period, 71.0, 800.0, 2.0, 0.1, _p(f0), _p(tp))
    assert rc == 0
    return f0, tp


def harvest_num_frames(n, fs, frame_period=5.0):
    return int(1000.0 * n / fs / frame_period) + 1


def harvest(x, fs, frame_period=5.0, f0_floor=71.0, f0_ceil=800.0, debug=False, mirror_write=True):
    """pyworld.harvest(x, fs, f0_floor, f0_ceil, frame_period) -> (f0, t); with `debug` also the
    per-stage arrays (1 ms grid).  mirror_write=False drops the spectrum bins that
    GetFilteredSignal overwrites while it multiplies (test decomposition only)."""
    _fn("orc_harvest_set_mirror_write", None, [c_int])(1 if mirror_write else 0)
    x = np.ascontiguousarray(x, dtype=np.float64)
    T = harvest_num_frames(len(x), fs, frame_period)
    f0 = np.zeros(T)
    tp = np.zeros(T)
    fn = _fn("orc_harvest_debug", c_int, [c_void_p, c_int, c_int, c_double, c_double, c_double,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p])
    if not debug:
        assert fn(_p(x), len(x), fs, frame_period, f0_floor, f0_ceil, _p(f0), _p(tp), None, None,
                  None, None, None) == 0
        return f0, tp
    import math
    nch = 1 + int(math.log(f0_ceil * 1.1 / (f0_floor * 0.9)) / math.log(2.0) * 40.0)
    T1 = harvest_num_frames(len(x), fs, 1.0)
    maxc = int(nch / 10.0 + 0.5) * 7
    raw = np.zeros((nch, T1))
    cand = np.zeros((T1, maxc))
    score = np.zeros((T1, maxc))
    best = np.zeros(T1)
    dims = np.zeros(8, dtype=np.int32)
    assert fn(_p(x), len(x), fs, frame_period, f0_floor, f0_ceil, _p(f0), _p(tp), _p(raw),
              _p(cand), _p(score), _p(best), _p(dims)) == 0
    assert dims[0] == nch and dims[1] == T1 and dims[2] == maxc
    return f0, tp, dict(raw=raw, cand=cand, score=score, best=best, n_cand=int(dims[4]),
                        y_length=int(dims[3]))


def harvest_waveform(x, fs):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(len(x) + 8)
    fn = _fn("orc_harvest_waveform", c_int, [c_void_p, c_int, c_int, c_void_p])
    n = fn(_p(x), len(x), fs, _p(y))
    return y[:n].copy()


def stonemask(x, fs, tp, f0):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros(len(f0))
    fn = _fn("orc_stonemask", c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p])
    assert fn(_p(x), len(x), fs, _p(np.ascontiguousarray(tp)), _p(np.ascontiguousarray(f0)),
              len(f0), _p(out)) == 0
    return out


def cheaptrick(x, fs, tp, f0, fft_size=None, q1=-0.15):
    x = np.ascontiguousarray(x, dtype=np.float64)
    fft_size = fft_size or cheaptrick_fft_size(fs)
    sp = np.zeros((len(f0), fft_size // 2 + 1))
    fn = _fn("orc_cheaptrick", c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                       c_double, c_void_p])
    assert fn(_p(x), len(x), fs, _p(np.ascontiguousarray(tp)), _p(np.ascontiguousarray(f0)),
              len(f0), fft_size, q1, _p(sp)) == 0
    return sp


def d4c(x, fs, tp, f0, fft_size=None, threshold=0.85):
    x = np.ascontiguousarray(x, dtype=np.float64)
    fft_size = fft_size or cheaptrick_fft_size(fs)
    ap = np.zeros((len(f0), fft_size // 2 + 1))
    fn = _fn("orc_d4c", c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                c_double, c_void_p])
    assert fn(_p(x), len(x), fs, _p(np.ascontiguousarray(tp)), _p(np.ascontiguousarray(f0)),
              len(f0), fft_size, threshold, _p(ap)) == 0
    return ap


def wav2world(x, fs, fft_size=None, frame_period=5.0):
    """pyworld.wav2world -> (f0 [T], sp [T,K] power, ap [T,K])."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    fft_size = fft_size or cheaptrick_fft_size(fs)
    T = num_frames(len(x), fs, frame_period)
    K = fft_size // 2 + 1
    f0 = np.zeros(T)
    sp = np.zeros((T, K))
    ap = np.zeros((T, K))
    fn = _fn("orc_wav2world", c_int, [c_void_p, c_int, c_int, c_double, c_int, c_void_p,
                                      c_void_p, c_void_p])
    assert fn(_p(x), len(x), fs, frame_period, fft_size, _p(f0), _p(sp), _p(ap)) == 0
    return f0, sp, ap


def code_aperiodicity(ap, fs):
    ap = np.ascontiguousarray(ap, dtype=np.float64)
    T, K = ap.shape
    nap = int(min(15000.0, fs / 2.0 - 3000.0) / 3000.0)
    bap = np.zeros((T, nap))
    fn = _fn("orc_code_aperiodicity", c_int, [c_void_p, c_int, c_int, c_int, c_void_p])
    assert fn(_p(ap), T, (K - 1) * 2, fs, _p(bap)) == 0
    return bap


def decode_aperiodicity(bap, fs, fft_size):
    bap = np.ascontiguousarray(bap, dtype=np.float64)
    T = bap.shape[0]
    ap = np.zeros((T, fft_size // 2 + 1))
    fn = _fn("orc_decode_aperiodicity", c_int, [c_void_p, c_int, c_int, c_int, c_void_p])
    assert fn(_p(bap), T, fs, fft_size, _p(ap)) == 0
    return ap


def mcep(amp_sp, order, alpha, eps=1e-8, miniter=2, maxiter=30, threshold=1e-3,
         return_iters=False):
    """pysptk.mcep(amp_sp, order, alpha, eps=eps, etype=1, itype=3) on [T, K] f64."""
    a = np.ascontiguousarray(amp_sp, dtype=np.float64)
    T, K = a.shape
    out = np.zeros((T, order + 1))
    iters = np.zeros(T, dtype=np.int32)
    fn = _fn("orc_mcep", c_int, [c_void_p, c_int, c_int, c_int, c_double, c_double, c_int, c_int,
                                 c_double, c_void_p, c_void_p])
    rc = fn(_p(a), T, K, order, alpha, eps, miniter, maxiter, threshold, _p(out), _p(iters))
    assert rc == 0, rc
    return (out, iters) if return_iters else out


def mgc2sp_logamp(mc, alpha, fftlen):
    """pysptk.mgc2sp(mc, alpha, 0.0, fftlen).real (log amplitude) [T, fftlen/2+1] f64."""
    m = np.ascontiguousarray(mc, dtype=np.float64)
    T, M1 = m.shape
    out = np.zeros((T, fftlen // 2 + 1))
    fn = _fn("orc_mgc2sp_logamp", c_int, [c_void_p, c_int, c_int, c_double, c_int, c_void_p])
    assert fn(_p(m), T, M1 - 1, alpha, fftlen, _p(out)) == 0
    return out


def mgcep(amp_sp, order, alpha, gamma, eps=1e-8, miniter=2, maxiter=30, threshold=1e-3,
          return_iters=False):
    """pysptk.mgcep(amp_sp, order, alpha, gamma, eps=eps, min_det=0, etype=1, itype=3) on [T, K]."""
    a = np.ascontiguousarray(amp_sp, dtype=np.float64)
    T, K = a.shape
    out = np.zeros((T, order + 1))
    iters = np.zeros(T, dtype=np.int32)
    fn = _fn("orc_mgcep", c_int, [c_void_p, c_int, c_int, c_int, c_double, c_double, c_double,
                                  c_int, c_int, c_double, c_void_p, c_void_p])
    rc = fn(_p(a), T, K, order, alpha, gamma, eps, miniter, maxiter, threshold, _p(out), _p(iters))
    assert rc == 0, rc
    return (out, iters) if return_iters else out


def mgc2sp_gamma_logamp(mgc, alpha, gamma, fftlen):
    """pysptk.mgc2sp(mgc, alpha, gamma, fftlen).real (log amplitude) [T, fftlen/2+1] f64."""
    m = np.ascontiguousarray(mgc, dtype=np.float64)
    T, M1 = m.shape
    out = np.zeros((T, fftlen // 2 + 1))
    fn = _fn("orc_mgc2sp_gamma", c_int, [c_void_p, c_int, c_int, c_double, c_double, c_int,
                                         c_void_p])
    assert fn(_p(m), T, M1 - 1, alpha, gamma, fftlen, _p(out)) == 0
    return out


def synthesize(f0, sp, ap, fs, frame_period=5.0):
    f0 = np.ascontiguousarray(f0, dtype=np.float64)
    sp = np.ascontiguousarray(sp, dtype=np.float64)
    ap = np.ascontiguousarray(ap, dtype=np.float64)
    T, K = sp.shape
    yl = int(T * frame_period * fs / 1000)
    y = np.zeros(yl)
    fn = _fn("orc_synthesize", c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_double,
                                       c_int, c_int, c_void_p])
    assert fn(_p(f0), T, _p(sp), _p(ap), (K - 1) * 2, frame_period, fs, yl, _p(y)) == 0
    return y
