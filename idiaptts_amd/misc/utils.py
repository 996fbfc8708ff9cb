"""Frame utilities with the reference's exact semantics (idiaptts/misc/utils.py:40-105).

`interpolate_lin` is sequential index logic on a few thousand frames and stays on the host (as
in the reference); it reproduces the reference's quirks bit for bit:
the output aliases the scanned array, the interpolation target is reached one frame early, and a
gap whose next voiced frame is the LAST frame is filled with the last voiced value (overwriting
that frame).  `compute_deltas` = np.gradient in float32; on the device it is part of
idiaptts_amd.ops.assemble_cmp (itts_assemble_cmp_f32: static, delta and delta-delta columns at once).
"""
import numpy as np


def interpolate_lin(data):
    """Continuous f0/lf0 + V/UV from a discontinuous contour (reference: misc/utils.py:40-86).

    :return: (interpolated [T,1] in the input dtype, vuv [T,1] float64)
    """
    data = np.reshape(np.copy(data), (data.size, 1))
    n = data.size
    vuv_vector = np.zeros((n, 1))
    vuv_vector[data > 0.0] = 1.0

    flat = data[:, 0]  # view: writes go to `data`, which is also what the scan reads (aliasing)
    last_value = 0.0
    i = 0
    while i < n:
        if flat[i] <= 0.0:
            # first voiced frame after i; if none: last index (python's for-else free `j`)
            j = i + 1
            if j < n:
                nz = np.nonzero(flat[j:] > 0.0)[0]
                j = j + int(nz[0]) if nz.size else n - 1
            if j < n - 1:
                if last_value > 0.0:
                    step = (flat[j] - flat[i - 1]) / float(j - i)
                    ks = np.arange(1, j - i + 1).astype(flat.dtype)
                    flat[i:j] = flat[i - 1] + step * ks
                else:
                    flat[i:j] = flat[j]
                # the filled frames are re-scanned by the reference; positive fills just update
                # last_value, non-positive fills would be handled again (cannot happen for j<n-1
                # because flat[j] > 0 and flat[i-1] > 0), so continue scanning from i.
                if flat[i] > 0.0:
                    last_value = flat[j - 1]
                    i = j
                    continue
                i += 1
                continue
            else:
                flat[i:] = last_value
                if last_value > 0.0:
                    break          # tail is positive: every remaining frame keeps last_value
                i += 1             # tail filled with 0: rescanned frame by frame, same result
                continue
        else:
            last_value = flat[i]
        i += 1
    return data, vuv_vector


def compute_deltas(labels):
    """np.gradient along time, float32 (reference: misc/utils.py:103-105)."""
    return np.gradient(labels, axis=0).astype(dtype=np.float32)
