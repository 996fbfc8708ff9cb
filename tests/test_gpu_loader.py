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
