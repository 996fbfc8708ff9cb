"""Frame-feature kernels (csrc/features.hip) against the reference's own host logic:
interpolate_lin bit for bit against vectors captured from the imported reference
(tests/golden/host_logic.npz) and against the host restatement on long / degenerate contours;
deltas and stream layout bit for bit against np.gradient and the delta columns of the reference's
golden `.cmp` files; normalisation sums against numpy float64."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_il(gpu, arrs):
    from idiaptts_amd import ops
    off = np.concatenate([[0], np.cumsum([len(a) for a in arrs])]).tolist()
    x = torch.from_numpy(np.concatenate(arrs).astype(np.float32)).to(gpu)
    ip, vuv = ops.interpolate_lin_f32(x, off)
    ip, vuv = ip.cpu().numpy(), vuv.cpu().numpy()
    return [(ip[off[i]:off[i + 1]], vuv[off[i]:off[i + 1]]) for i in range(len(arrs))]


def test_interpolate_lin_matches_reference_vectors_bit_for_bit(gpu, golden_dir):
    g = np.load(os.path.join(golden_dir, "host_logic.npz"))
    idx = [i for i in range(int(g["il_count"])) if g["il_in_%d" % i].dtype == np.float32]
    assert len(idx) >= 30
    res = _run_il(gpu, [g["il_in_%d" % i].reshape(-1) for i in idx])
    for (ip, vuv), i in zip(res, idx):
        assert np.array_equal(ip, g["il_ip_%d" % i].reshape(-1)), i
        assert np.array_equal(vuv, g["il_vuv_%d" % i].reshape(-1).astype(np.float32)), i


def test_interpolate_lin_long_and_degenerate_contours(gpu):
    """More than one 256-frame chunk per utterance, gaps across chunk borders, all voiced, all
    unvoiced, one and two frames, the next voiced frame being the last one -- against the host
    restatement (itself pinned to the reference on the CPU, tests/test_host_logic.py)."""
    from idiaptts_amd.misc.utils import interpolate_lin
    rng = np.random.default_rng(3)
    arrs = []
    for T, pv in [(5000, 0.5), (777, 0.02), (1300, 0.97), (256, 0.5), (257, 0.5), (513, 0.3),
                  (4096, 0.001)]:
        x = rng.uniform(3.5, 6.0, size=T) * (rng.uniform(size=T) < pv)
        arrs.append(x.astype(np.float32))
    arrs += [np.zeros(700, np.float32), np.full(300, 5.0, np.float32), np.zeros(1, np.float32),
             np.array([4.5], np.float32), np.array([0, 5], np.float32),
             np.array([5, 0], np.float32), np.array([4, 0, 0, 0, 6], np.float32),
             np.array([0, 0, 0, 6], np.float32), np.array([4, 0, 0, 6, 0, 0, 5], np.float32)]
    long_tail = rng.uniform(3.5, 6.0, size=1000).astype(np.float32)
    long_tail[400:999] = 0          # the gap ends at the LAST frame: tail fill overwrites it
    arrs.append(long_tail)
    res = _run_il(gpu, arrs)
    for (ip, vuv), a in zip(res, arrs):
        want_ip, want_vuv = interpolate_lin(a)
        assert np.array_equal(ip, want_ip.reshape(-1)), len(a)
        assert np.array_equal(vuv, want_vuv.reshape(-1).astype(np.float32))


def test_lf0_vuv_matches_host_formula(gpu):
    """WorldFeatLabelGen.py:798-802 on f0 contours: V/UV exact; lf0 within 2 float32 ulp of
    numpy's float32 log (numpy's SIMD log is not correctly rounded and differs between CPUs; the
    kernel rounds the float64 log of the float32 f0)."""
    import math
    from idiaptts_amd import ops
    from idiaptts_amd.misc.utils import interpolate_lin
    rng = np.random.default_rng(4)
    f0s = []
    for T in (900, 1, 300, 2000):
        f0 = rng.uniform(71.0, 400.0, size=T) * (rng.uniform(size=T) < 0.6)
        f0[rng.integers(0, T, size=max(1, T // 50))] = rng.uniform(1.0, 29.9)   # below threshold
        f0s.append(f0)
    off = np.concatenate([[0], np.cumsum([len(f) for f in f0s])]).tolist()
    lf0, vuv = ops.lf0_vuv(torch.from_numpy(np.concatenate(f0s)).to(gpu), off, 30, 0)
    lf0, vuv = lf0.cpu().numpy(), vuv.cpu().numpy()
    for u, f0 in enumerate(f0s):
        ref = np.log(f0.clip(min=1e-10), dtype=np.float32)
        ref[ref <= math.log(30)] = 0
        want, want_vuv = interpolate_lin(ref)
        got = lf0[off[u]:off[u + 1]]
        assert np.array_equal(vuv[off[u]:off[u + 1]], want_vuv.reshape(-1).astype(np.float32))
        assert np.abs(got - want.reshape(-1)).max() <= 2 * np.spacing(np.float32(6.0))


def _np_cmp(sp, lf0, vuv, bap, deltas):
    from idiaptts_amd.misc.utils import compute_deltas
    parts = []
    for f in (sp, lf0[:, None], None, bap):
        if f is None:
            parts.append(vuv[:, None])
        elif deltas:
            d = compute_deltas(f)
            parts += [f, d, compute_deltas(d)]
        else:
            parts.append(f)
    return np.concatenate(parts, axis=1)


@pytest.mark.parametrize("deltas", [True, False])
@pytest.mark.parametrize("n_sp,n_bap", [(60, 1), (20, 1), (60, 5)])
def test_assemble_cmp_equals_numpy_gradient_bit_for_bit(gpu, deltas, n_sp, n_bap):
    from idiaptts_amd import ops
    rng = np.random.default_rng(n_sp + n_bap)
    lens = [2, 57, 3, 1300, 4, 300]
    off = np.concatenate([[0], np.cumsum(lens)]).tolist()
    N = off[-1]
    sp = rng.normal(size=(N, n_sp)).astype(np.float32)
    lf0 = rng.uniform(4, 6, size=N).astype(np.float32)
    vuv = (rng.uniform(size=N) < 0.5).astype(np.float32)
    bap = rng.normal(size=(N, n_bap)).astype(np.float32)
    out = ops.assemble_cmp(*(torch.from_numpy(a).to(gpu) for a in (sp, lf0, vuv, bap)), off,
                           add_deltas=deltas).cpu().numpy()
    assert out.shape == (N, (3 if deltas else 1) * (n_sp + 1 + n_bap) + 1)
    for u in range(len(lens)):
        a, b = off[u], off[u + 1]
        assert np.array_equal(out[a:b], _np_cmp(sp[a:b], lf0[a:b], vuv[a:b], bap[a:b], deltas)), u


def test_delta_columns_of_the_reference_cmp_files(gpu, golden_dir):
    """The reference's golden `.cmp` (mcep20 + deltas, lf0 + deltas, vuv, bap + deltas): feeding
    its static columns through the kernel reproduces every delta / delta-delta column exactly."""
    from idiaptts_amd import ops
    for name in ("LJ001-0002", "LJ001-0008"):
        cmp_ = np.fromfile(os.path.join(golden_dir, name + ".cmp"), dtype=np.float32).reshape(-1, 67)
        sp, lf0, vuv, bap = cmp_[:, :20], cmp_[:, 60], cmp_[:, 63], cmp_[:, 64:65]
        out = ops.assemble_cmp(*(torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
                                 for a in (sp, lf0, vuv, bap)), [0, len(cmp_)]).cpu().numpy()
        assert np.array_equal(out, cmp_), name


@pytest.mark.parametrize("width,col0", [(180, 0), (3, 180), (15, 5), (17, 1)])
def test_feature_stats_match_numpy_float64(gpu, width, col0):
    from idiaptts_amd import ops
    rng = np.random.default_rng(width)
    for n in (1, 3, 1000, 70001):
        x = (rng.normal(size=(n, col0 + width + 2)) * 3 + 1).astype(np.float32)
        xd = torch.from_numpy(x).to(gpu)
        xs = x[:, col0:col0 + width].astype(np.float64)
        s, c = ops.feature_stats(xd, col0, width, True)
        assert np.allclose(s.cpu().numpy(), xs.sum(0), rtol=1e-12, atol=1e-9)
        assert np.allclose(c.cpu().numpy(), xs.T @ xs, rtol=1e-12, atol=1e-9)
        ops.feature_stats(xd, col0, width, True, sums=s, second=c)          # accumulate
        assert np.allclose(c.cpu().numpy(), 2 * (xs.T @ xs), rtol=1e-12, atol=1e-9)
        s, q = ops.feature_stats(xd, col0, width, False)
        assert np.allclose(s.cpu().numpy(), xs.sum(0), rtol=1e-12, atol=1e-9)
        assert np.allclose(q.cpu().numpy(), (xs ** 2).sum(0), rtol=1e-12, atol=1e-9)


def test_empty_batches_are_no_ops(gpu):
    from idiaptts_amd import ops
    e = torch.empty(0, dtype=torch.float32, device=gpu)
    ip, vuv = ops.interpolate_lin_f32(e, [0])
    assert ip.numel() == 0
    ip, vuv = ops.interpolate_lin_f32(e, [0, 0, 0])
    assert vuv.numel() == 0
    out = ops.assemble_cmp(torch.empty((0, 60), dtype=torch.float32, device=gpu), e, e,
                           torch.empty((0, 1), dtype=torch.float32, device=gpu), [0, 0])
    assert out.shape == (0, 187)
    s, c = ops.feature_stats(torch.empty((0, 8), dtype=torch.float32, device=gpu), 0, 8, True)
    assert float(s.abs().sum()) == 0 and float(c.abs().sum()) == 0


def test_device_square_root_is_numpys(gpu):
    """ops.sqrt_inplace (WorldFeatLabelGen.py:795, amp_sp = np.sqrt(sp) taken before the envelope leaves the device):
    IEEE square roots -- the same bits as numpy, over magnitudes from denormal to huge"""
    from idiaptts_amd import ops
    rng = np.random.default_rng(8)
    x = np.concatenate([rng.uniform(0, 1, 4097) * 10.0 ** rng.integers(-300, 300, 4097),
                        [0.0, 5e-324, 2.2250738585072014e-308, 1.0, 2.0, 1e-16, np.inf]])
    got = ops.sqrt_inplace(torch.from_numpy(x.copy()).to(gpu)).cpu().numpy()
    assert np.array_equal(got.view(np.uint64), np.sqrt(x).view(np.uint64))


def test_device_square_is_numpys(gpu):
    """ops.square_inplace (WorldFeatLabelGen.py:925, np.square(amp_sp, dtype=float64) taken after the upload): the same
    bits, also for a float32 envelope widened first"""
    from idiaptts_amd import ops
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, 5000) * 10.0 ** rng.integers(-160, 150, 5000)
    got = ops.square_inplace(torch.from_numpy(x.copy()).to(gpu)).cpu().numpy()
    assert np.array_equal(got.view(np.uint64), np.square(x).view(np.uint64))
    x32 = rng.uniform(0, 3, 4096).astype(np.float32)
    got = ops.square_inplace(torch.from_numpy(x32).to(gpu).double()).cpu().numpy()
    assert np.array_equal(got.view(np.uint64), np.square(x32, dtype=np.float64).view(np.uint64))
