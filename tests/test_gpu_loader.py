"""The module path's batch loading on a GPU box: padding straight into page-locked memory, and the handler's
thread-backed loader against torch's DataLoader (reference: DataLoader(..., num_workers, pin_memory),
model_trainers/ModularTrainer.py:831-841)."""
import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pad_sequence

pytestmark = pytest.mark.gpu


def test_pad_into_pinned_memory_equals_pad_sequence(gpu):
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import ModularModelHandlerPyTorch as H
    rng = np.random.default_rng(4)
    seqs = [rng.standard_normal((n, 7)).astype(np.float32) for n in (5, 1, 12, 3)]
    for batch_first in (False, True):
        got = H.unsorted_pad_sequence(seqs, batch_first, pinned=True)
        want = pad_sequence([torch.from_numpy(s) for s in seqs], batch_first)
        assert got.is_pinned() and torch.equal(got, want)
    ids = [torch.arange(n) for n in (4, 2)]
    assert torch.equal(H.unsorted_pad_sequence(ids, True, pinned=True), pad_sequence(ids, True))


class _Dicts(torch.utils.data.Dataset):
    datareaders = []                       # (no reader draws random numbers: the handler may use threads)

    def __init__(self, n):
        rng = np.random.default_rng(1)
        self.items = [{"x": rng.standard_normal((3 + (5 * i) % 11, 4)).astype(np.float32),
                       "y": rng.standard_normal((3 + (5 * i) % 11, 2)).astype(np.float32)} for i in range(n)]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


@pytest.mark.parametrize("shuffle", [False, True])
def test_handler_thread_loader_equals_dataloader(gpu, shuffle):
    from idiaptts_amd.src.data_preparation.ThreadedBatchLoader import ThreadedBatchLoader
    from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import ModularModelHandlerPyTorch as H
    handler = H.__new__(H)
    ds = _Dicts(37)

    def batches(worker_kind, workers):
        torch.manual_seed(5)
        loader = handler._get_dataloader(batch_size=8, dataset=ds, batch_first=False, num_workers=workers,
                                         pin_memory=True, shuffle=shuffle, worker_kind=worker_kind)
        return loader, [(d["x"].clone(), d["y"].clone(), l["x"].clone()) for d, l in loader]

    ref_loader, ref = batches("process", 0)
    thr_loader, got = batches("thread", 3)
    assert isinstance(thr_loader, ThreadedBatchLoader) and not isinstance(ref_loader, ThreadedBatchLoader)
    assert len(ref) == len(got) == 5
    for (x0, y0, l0), (x1, y1, l1) in zip(ref, got):
        assert torch.equal(x0, x1) and torch.equal(y0, y1) and torch.equal(l0, l1)
    assert all(d["x"].is_pinned() for d, _ in thr_loader)


def test_synthesis_with_spectra_on_the_side_stream_equals_the_sequential_calls(gpu):
    """world.synthesise_features (mgc2sp / decode_aperiodicity on the side stream, itts_world_synthesize_after waiting
    for their event right before the pulse kernel) against the three calls in a row on one stream: the same samples
    (reference call sites: WorldFeatLabelGen.py:925, 940-945)."""
    from idiaptts_amd import lib, ops, world
    from idiaptts_amd.bench_support import make_audio_batch
    L = lib.load()
    fs, hop = 16000, 5.0
    raws = make_audio_batch(6, fs, seed=3)
    n_fft = L.itts_cheaptrick_fft_size(fs, 71.0)
    alpha = L.itts_mcep_alpha(fs)
    x_off = world.offsets([len(r) for r in raws])
    f_off = world.offsets([world.num_frames(len(r), fs, hop) for r in raws])
    x = torch.from_numpy(np.concatenate(raws)).to(gpu)
    f0 = ops.stonemask(x, x_off, ops.dio(x, x_off, f_off, fs, hop), f_off, fs, hop)
    _, bap = ops.d4c(x, x_off, f0, f_off, fs, hop, n_fft, want_ap=False, want_bap=torch.float32)
    _, mc, _ = ops.cheaptrick_mcep(x, x_off, f0, f_off, fs, hop, n_fft, want_sp=False, order=24, alpha=alpha)
    mc64, bap64 = mc.double(), bap.double()
    pw = ops.mgc2sp(mc64, alpha, n_fft, want_pow=True)
    apd = ops.decode_aperiodicity(bap64, fs, n_fft)
    want, off0 = ops.world_synthesize(f0, pw, apd, f_off, fs, hop)
    for _ in range(3):
        got, off1 = world.synthesise_features(f0, f_off, fs, n_fft, mc=mc64, alpha=alpha, bap=bap64, hop_ms=hop)
        assert off0 == off1 and torch.equal(want, got)


def test_voiced_only_aperiodicity_decode_writes_the_rows_a_voiced_pulse_reads(gpu):
    """ops.decode_aperiodicity(..., voiced_f0=f0): frames with f0 > 0 and their two neighbours carry exactly what the
    full decode gives; the synthesis never reads the others (the equality test above is the proof of that)."""
    from idiaptts_amd import ops
    g = torch.Generator().manual_seed(11)
    T, fs, n_fft = 1237, 16000, 1024
    f0 = torch.zeros(T, dtype=torch.float64)
    for a, b in ((0, 3), (40, 41), (100, 350), (777, 778), (1230, 1237)):
        f0[a:b] = 120.0
    bap = -torch.rand(T, 1, dtype=torch.float64, generator=g) * 30.0
    bap[5::7] = -0.2          # frames the codec takes for unvoiced (mean above -0.5 dB)
    full = ops.decode_aperiodicity(bap.to(gpu), fs, n_fft).cpu()
    part = ops.decode_aperiodicity(bap.to(gpu), fs, n_fft, voiced_f0=f0.to(gpu)).cpu()
    v = f0 > 0
    need = v.clone()
    need[1:] |= v[:-1]
    need[:-1] |= v[1:]
    assert 0 < int(need.sum()) < T
    assert torch.equal(full[need], part[need])
