"""Random-shape sweeps as tests: MLPG (whichever solve the library picks for the batch) against the C
oracle utterance by utterance, the fp32 GEMM entry points against torch in float64
(scripts/gemm_fuzz.py, a child process)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("solve", ["auto", "ring"])
def test_mlpg_random_shapes(gpu, solve, monkeypatch):
    """(under the library's own choice of solve, and with the one-pass ring kernel forced for every batch whose
    longest utterance has 194 frames or more)
    80 random batches: empty, one-frame and chunk-boundary lengths, 1 .. 129 dimensions (one to three
    64-dimension blocks), input / output column offsets, slowly settling factors; untouched columns
    of the output array must stay untouched (misc/mlpg.py:94-127 per utterance is the reference)."""
    from idiaptts_amd import ops
    from oracle import capi
    monkeypatch.delenv("ITTS_MLPG_STREAM", raising=False)
    if solve == "ring":
        monkeypatch.setenv("ITTS_MLPG_RING", "1")
    else:
        monkeypatch.delenv("ITTS_MLPG_RING", raising=False)
    rng = np.random.default_rng(5)
    worst = 0.0
    for case in range(80):
        n_utts = int(rng.integers(1, 24))
        kind = rng.integers(0, 4)
        hi = (40, 400, 2500, 5000)[kind]
        lengths = rng.integers(0 if kind == 0 else 1, hi, size=n_utts)
        if rng.random() < 0.3:
            lengths[rng.integers(0, n_utts)] = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, 64, 65]))
        if lengths.sum() == 0:
            lengths[0] = 5
        dim = int(rng.choice([1, 2, 3, 7, 20, 60, 62, 64, 65, 70, 129]))
        col0, extra, ocol0, oextra = (int(rng.integers(0, 4)) for _ in range(4))
        off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        T = int(off[-1])
        feat = rng.normal(size=(T, col0 + 3 * dim + extra))
        var = rng.uniform(0.01, 1.0, size=3 * dim)
        if rng.random() < 0.25:
            var[dim:] *= 10.0 ** rng.uniform(-6, 0)          # slowly settling factor
        out = torch.full((T, ocol0 + dim + oextra), 3.5, dtype=torch.float64, device=gpu)
        ops.mlpg_generation(torch.from_numpy(feat).to(gpu), torch.from_numpy(var).to(gpu), dim, off.tolist(),
                            col0=col0, out=out, ocol0=ocol0)
        got = out.cpu().numpy()
        assert (got[:, :ocol0] == 3.5).all() and (got[:, ocol0 + dim:] == 3.5).all(), (case, "columns touched")
        for u in range(n_utts):
            a, b = off[u], off[u + 1]
            if b == a:
                continue
            ref = capi.mlpg(feat[a:b], var, dim, col0=col0)
            err = np.abs(got[a:b, ocol0:ocol0 + dim] - ref).max() / max(1.0, np.abs(ref).max())
            assert np.isfinite(err) and err < 1e-9, (case, u, err, lengths.tolist()[:8], dim, col0, ocol0)
            worst = max(worst, err)
    print("worst relative difference to the oracle: %.1e" % worst)


def test_gemm_random_shapes(gpu):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gemm_fuzz.py"), "60", "9"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:]
    assert "cases 60" in res.stdout


def test_grouped_tile_order_random_shapes(gpu):
    """Forward products with more than ~4 MB of weights walk groups of column tiles (nn.hip ring_group):
    25 random shapes, every output element against torch fp64 (scripts/grouped_order_fuzz.py)."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "grouped_order_fuzz.py"), "25"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
    assert "cases 25" in res.stdout


def test_valid_rows_and_cached_batches_random_shapes(gpu):
    """Random batch shapes through the round-6 batch kernels (csrc/batch_rows.hip): Linear groups on valid rows against
    the padded computation (outputs bit for bit, gradients to rounding), and the cached loader against prepare_batch --
    batches of one, no padding at all, a single frame, widths that are and are not multiples of four."""
    from functools import partial
    from torch.nn.utils.rnn import pad_sequence
    from torch.utils.data import DataLoader
    import types
    from idiaptts_amd.nn.functional import padding_rows_identical
    from idiaptts_amd.src.data_preparation.DeviceBatchCache import CachedBatchLoader
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import ModularModelHandlerPyTorch as H
    from idiaptts_amd.src.neural_networks.pytorch.models import rnn_dyn
    rng = np.random.default_rng(2026)
    for case in range(14):
        B = int(rng.choice([1, 2, 3, 9, 33]))
        t_hi = int(rng.choice([1, 2, 17, 130]))
        lens = torch.from_numpy(rng.integers(1, t_hi + 1, size=B))
        if case % 5 == 0:
            lens[:] = int(lens.max())                               # no padding at all
        batch_first = bool(case % 2)
        d_in, d_h, d_out = int(rng.choice([3, 8, 13])), int(rng.choice([16, 20])), int(rng.choice([1, 4, 7]))
        act = ["TANH", "RELU"][case % 2]
        torch.manual_seed(case)
        hp = types.SimpleNamespace(model_type="RNNDYN-2_{}_{}-1_FC_{}".format(act, d_h, d_out), batch_first=batch_first,
                                   dropout=0.0)
        model = rnn_dyn.convert_legacy_to_config((d_in,), hp).create_model().to(gpu)
        model.layer_groups[0].min_padding_share = 0.0               # (pack whenever there is any padding)
        model.layer_groups[1].min_padding_share = 0.0
        g = torch.Generator().manual_seed(case)
        seqs = [torch.randn(int(n), d_in, generator=g) for n in lens]
        x = pad_sequence(seqs, batch_first=batch_first).to(gpu)
        T = int(lens.max())
        outs = []
        for packed in (False, True):
            model.zero_grad()
            with padding_rows_identical(packed):
                y, _ = model(x, seq_lengths_input=lens, max_length_inputs=T)
            (y ** 2).sum().backward()
            outs.append((y.detach().clone(), [p.grad.clone() for p in model.parameters()]))
        assert torch.equal(outs[0][0], outs[1][0]), case
        for a, b in zip(outs[0][1], outs[1][1]):
            assert float((a - b).abs().max()) <= 3e-5 * (float(a.abs().max()) + 1e-30), case

        # the cached loader over the same utterances (two streams, one masked)
        class R(object):
            min_frames = other_pad_dims = max_frames = None
            pad_mode = "constant"

            def __init__(self, name, mask):
                self.name, self.output_names, self.requires_seq_mask = name, [name], mask

        class D(torch.utils.data.Dataset):
            datareaders = [R("x", False), R("y", True)]

            def get_datareader_by_output_name(self, name):
                return next(r for r in self.datareaders if r.name == name)

            def __len__(self):
                return B

            def __getitem__(self, i):
                return {"x": seqs[i].numpy(), "_id_list": str(i), "y": seqs[i].numpy()[:, :1] * 2}, self

        ds = D()
        bs = int(rng.integers(1, B + 1))
        torch.manual_seed(100 + case)
        ref = [[b for b in DataLoader(ds, batch_size=bs, shuffle=True, num_workers=0,
                                      collate_fn=partial(H.prepare_batch, batch_first=batch_first))] for _ in range(2)]
        torch.manual_seed(100 + case)
        loader = CachedBatchLoader(ds, bs, True, gpu, batch_first, threads=int(case % 3), host_collate=H.prepare_batch)
        got = [[b for b in loader] for _ in range(2)]
        for e0, e1 in zip(ref, got):
            assert len(e0) == len(e1)
            for (d0, l0), (d1, l1) in zip(e0, e1):
                assert list(d0) == list(d1) and d0["_id_list"] == d1["_id_list"]
                for k in ("x", "y", "y_mask"):
                    assert torch.equal(d0[k], d1[k].cpu()), (case, k)
                assert torch.equal(l0["x"], l1["x"])
