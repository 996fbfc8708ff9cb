"""GPU parity: HIP MLPG (through the C ABI) vs the C oracle (oracle/c/mlpg.c)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "ring", "stream"])
def solve(request, monkeypatch):
    """The library picks between its solves by batch shape (sequential sweeps under 194 frames; above that the
    one-pass ring kernel from 128 (utterance, 64-dimension) units, reduce -> scan -> solve below): these tests run
    under its own choice and with each of the two long-utterance forms forced (csrc/mlpg.hip reads the variables
    at every call)."""
    monkeypatch.delenv("ITTS_MLPG_RING", raising=False)
    monkeypatch.delenv("ITTS_MLPG_STREAM", raising=False)
    if request.param == "ring":
        monkeypatch.setenv("ITTS_MLPG_RING", "1")
    elif request.param == "stream":
        monkeypatch.setenv("ITTS_MLPG_STREAM", "1")
    return request.param


def _case(rng, lengths, dim, extra_cols=0, col0=0):
    T = int(sum(lengths))
    feat = rng.normal(size=(T, col0 + 3 * dim + extra_cols))
    var = rng.uniform(0.01, 1.0, size=3 * dim)
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    return feat, var, offsets


@pytest.mark.parametrize("lengths,dim,col0", [
    ([1], 3, 0), ([2], 3, 0), ([3, 1, 2], 5, 0), ([4, 7, 300], 60, 0), ([1931], 20, 0),
    ([50, 0, 75], 1, 2), ([640, 1300], 62, 1),
    # time-parallel path: utterance lengths around the 64-frame chunk boundaries, mixed with short ones
    ([194, 195, 257, 258, 259, 130, 66, 3, 1, 322], 4, 0), ([193, 2], 2, 0), ([4098], 1, 0),
    # ring kernel: around its 24-frame segments, the five segments under the staged factor rows, the 288-frame ring
    ([200, 23, 24, 25, 47, 48, 49, 119, 120, 121, 143, 144, 145], 62, 0), ([287, 288, 289, 311, 312, 313, 575, 576, 577], 7, 0),
    ([2000, 600, 601], 64, 1),
])
def test_mlpg_matches_oracle(gpu, solve, lengths, dim, col0):
    """The solves the library holds -- the sequential sweeps for batches whose longest utterance is
    under 194 frames, the one-pass ring kernel or reduce -> scan -> solve otherwise -- against the C oracle."""
    from idiaptts_amd import ops
    from oracle import capi
    rng = np.random.default_rng(7)
    feat, var, offsets = _case(rng, lengths, dim, extra_cols=2, col0=col0)
    out = ops.mlpg_generation(torch.from_numpy(feat).to(gpu), torch.from_numpy(var).to(gpu), dim,
                              offsets.tolist(), col0=col0).cpu().numpy()
    for u in range(len(lengths)):
        a, b = offsets[u], offsets[u + 1]
        if b == a:
            continue
        ref = capi.mlpg(feat[a:b], var, dim, col0=col0)
        err = np.abs(out[a:b] - ref).max()
        rmse = np.sqrt(np.mean((out[a:b] - ref) ** 2))
        assert rmse <= 1e-10 and err <= 1e-9, (u, err, rmse)  # north-star bar: 1e-4 RMSE


def test_mlpg_full_size_property(gpu, solve):
    """BASELINE config 4 shape (187-dim cmp, 256 utterances): P x = b must hold to fp64
    round-off for the solution returned, checked through the normal equations on a sample."""
    from idiaptts_amd import ops
    rng = np.random.default_rng(11)
    lengths = rng.integers(400, 2000, size=256)
    dim = 60
    feat, var, offsets = _case(rng, lengths, dim)
    x = ops.mlpg_generation(torch.from_numpy(feat).to(gpu), torch.from_numpy(var).to(gpu), dim,
                            offsets.tolist()).cpu().numpy()
    assert np.isfinite(x).all()
    for u in (0, 100, 255):
        a, b = offsets[u], offsets[u + 1]
        T = b - a
        for d in (0, 31, 59):
            tau = np.zeros((T, 3))
            for w in range(3):
                tau[:, w] = 1.0 / var[w * dim + d]
            tau[0, 1:] = tau[-1, 1:] = 1e-11
            m = feat[a:b, [d, dim + d, 2 * dim + d]]
            xs = x[a:b, d]
            # gradient of the quadratic form: sum_w W_w^T tau_w (W_w x - m_w) = 0
            xp = np.concatenate([[0.0], xs, [0.0]])
            w1x = 0.5 * (xp[2:] - xp[:-2])
            w2x = xp[2:] - 2 * xp[1:-1] + xp[:-2]
            r0 = tau[:, 0] * (xs - m[:, 0])
            r1 = tau[:, 1] * (w1x - m[:, 1])
            r2 = tau[:, 2] * (w2x - m[:, 2])
            r1p = np.concatenate([[0.0], r1, [0.0]])
            r2p = np.concatenate([[0.0], r2, [0.0]])
            g = r0 + 0.5 * (r1p[:-2] - r1p[2:]) + (r2p[:-2] - 2 * r2p[1:-1] + r2p[2:])
            scale = np.abs(tau[:, 0] * m[:, 0]).max() + 1.0
            assert np.abs(g).max() / scale < 1e-9


@pytest.mark.parametrize("lengths,ratio", [([700, 90, 333], 1e-6), ([700, 90, 333], 1e-3), ([3000, 260], 1e-7)])
def test_mlpg_slowly_settling_factor(gpu, solve, lengths, ratio):
    """Delta variances up to 1e7 times smaller than the static ones: the Cholesky factor needs hundreds
    of frames to become stationary, so most chunks of an utterance carry their own matrices (the scan
    kernel's single-chunk segments, and -- at 3 000 frames -- its sequential road, taken when there
    are more such chunks than the workgroup has waves)."""
    from idiaptts_amd import ops
    from oracle import capi
    rng = np.random.default_rng(3)
    dim = 3
    feat, var, offsets = _case(rng, lengths, dim)
    var[dim:] *= ratio
    out = ops.mlpg_generation(torch.from_numpy(feat).to(gpu), torch.from_numpy(var).to(gpu), dim,
                              offsets.tolist()).cpu().numpy()
    for u in range(len(lengths)):
        a, b = offsets[u], offsets[u + 1]
        ref = capi.mlpg(feat[a:b], var, dim)
        scale = max(1.0, np.abs(ref).max())
        assert np.abs(out[a:b] - ref).max() <= 1e-9 * scale, (u, np.abs(out[a:b] - ref).max())


def test_mlpg_more_than_64_dimensions_and_an_output_slice(gpu, solve):
    """Two 64-dimension blocks (dim = 70) and a result written into columns 3 .. 72 of a wider
    array whose other columns must stay untouched (the solve also parks b in those rows)."""
    from idiaptts_amd import ops
    from oracle import capi
    rng = np.random.default_rng(21)
    lengths, dim = [300, 17, 500], 70
    feat, var, offsets = _case(rng, lengths, dim)
    out = torch.full((int(offsets[-1]), dim + 5), -7.0, dtype=torch.float64, device=gpu)
    ops.mlpg_generation(torch.from_numpy(feat).to(gpu), torch.from_numpy(var).to(gpu), dim, offsets.tolist(),
                        out=out, ocol0=3)
    got = out.cpu().numpy()
    assert (got[:, :3] == -7.0).all() and (got[:, 3 + dim:] == -7.0).all()
    for u in range(len(lengths)):
        a, b = offsets[u], offsets[u + 1]
        ref = capi.mlpg(feat[a:b], var, dim)
        assert np.abs(got[a:b, 3:3 + dim] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("lengths,dim", [([300, 40, 1000], 62), ([250] * 130, 5), ([120, 60], 3)])
def test_mlpg_float32_rows(gpu, solve, lengths, dim):
    """float32 input rows (the acoustic model's output type; mlpg.py:119-121 assigns them into float64 arrays): the
    result is that of the widened rows, whichever solve takes the batch -- the one-pass kernel converts in its loads,
    the others read a widened copy."""
    from idiaptts_amd import ops
    from oracle import capi
    rng = np.random.default_rng(17)
    feat, var, offsets = _case(rng, lengths, dim, extra_cols=3, col0=2)
    feat32 = feat.astype(np.float32)
    wide = feat32.astype(np.float64)
    got = ops.mlpg_generation(torch.from_numpy(feat32).to(gpu), torch.from_numpy(var).to(gpu), dim,
                              offsets.tolist(), col0=2).cpu().numpy()
    same = ops.mlpg_generation(torch.from_numpy(wide).to(gpu), torch.from_numpy(var).to(gpu), dim,
                               offsets.tolist(), col0=2).cpu().numpy()
    assert np.array_equal(got, same)
    for u in (0, len(lengths) - 1):
        a, b = offsets[u], offsets[u + 1]
        ref = capi.mlpg(wide[a:b], var, dim, col0=2)
        assert np.abs(got[a:b] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_mlpg_wide_helpers(gpu, monkeypatch, dtype):
    """The one-pass kernel's helpers with two dimensions a lane and two rows a memory instruction (the library takes
    them for float32 rows from 1 024 units; forced here): the same values as with one dimension a lane, bit for bit,
    on lengths around the segment / ring boundaries, 62 and 70 dimensions (one and two blocks), column offsets that
    leave the rows 8- but not 16-byte aligned."""
    from idiaptts_amd import ops
    rng = np.random.default_rng(29)
    for lengths, dim, col0 in (([700, 25, 24, 289, 1, 2, 600, 313], 62, 1), ([400, 333], 70, 0), ([2100], 2, 3)):
        feat, var, offsets = _case(rng, lengths, dim, extra_cols=1, col0=col0)
        feat = feat.astype(dtype)
        f, v = torch.from_numpy(feat).to(gpu), torch.from_numpy(var).to(gpu)
        monkeypatch.setenv("ITTS_MLPG_RING", "1")
        monkeypatch.setenv("ITTS_MLPG_NARROW", "1")
        narrow = ops.mlpg_generation(f, v, dim, offsets.tolist(), col0=col0).cpu().numpy()
        monkeypatch.delenv("ITTS_MLPG_NARROW")
        monkeypatch.setenv("ITTS_MLPG_WIDE", "1")
        wide = ops.mlpg_generation(f, v, dim, offsets.tolist(), col0=col0).cpu().numpy()
        monkeypatch.delenv("ITTS_MLPG_WIDE")
        assert np.array_equal(narrow, wide), (lengths, dim)


def test_planned_calls_equal_plain_calls(gpu):
    """ops.MlpgPlan (itts_mlpg_plan_create / itts_mlpg_generation_planned): the offsets' share of a call prepared
    once -- the same trajectories bit for bit, for the one-pass kernel and the small-batch forms, float64 and float32
    rows, several streams on one plan (misc/mlpg.py:94-127 is called once per stream)."""
    from idiaptts_amd import ops
    g = torch.Generator().manual_seed(4)
    for n_utts, dim in ((3, 5), (300, 62), (1, 1)):
        lens = torch.randint(40, 400, (n_utts,), generator=g).tolist()
        off = [0]
        for n in lens:
            off.append(off[-1] + n)
        feat = torch.randn(off[-1], 3 * dim + 4, dtype=torch.float64, generator=g).to(gpu)
        var = (torch.rand(3 * dim, dtype=torch.float64, generator=g) + 0.05).to(gpu)
        plan = ops.MlpgPlan(off)
        for rows in (feat, feat.float()):
            for col0 in (0, 4):
                want = ops.mlpg_generation(rows, var, dim, off, col0=col0)
                got = ops.mlpg_generation(rows, var, dim, off, col0=col0, plan=plan)
                assert torch.equal(want, got)
        plan.close()
    with pytest.raises(Exception):
        ops.MlpgPlan([0, 5, 3])
