"""Host logic of the device batch cache (data_preparation/DeviceBatchCache.py) without a GPU: the loader's
bookkeeping -- index draws, slots, budget, scratch rows, key order, lengths, masks -- against the reference's own
road, `DataLoader(dataset, collate_fn=prepare_batch)` (ModularModelHandlerPyTorch.py:388-465, :500-548).  The arena
here is a stand-in on host tensors (the product's arena is HBM + itts_batch_pad_gather_f32; tests/test_gpu_loader.py
runs the same comparison through it)."""
from functools import partial

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from idiaptts_amd.src.data_preparation.DataReaders import ReaderBase
from idiaptts_amd.src.data_preparation.DeviceBatchCache import CachedBatchLoader
from idiaptts_amd.src.data_preparation.PyTorchDatareadersDataset import PyTorchDatareadersDataset
from idiaptts_amd.src.neural_networks.pytorch.ModularModelHandlerPyTorch import ModularModelHandlerPyTorch as H


class HostArena(object):
    """The arena's contract restated with torch on host tensors (test side only)."""

    def __init__(self, width, device, capacity_rows):
        self.width = width
        self.rows = torch.full((max(capacity_rows, 1), width), float("nan"))
        self.used = 0
        self.grown = 0

    def ensure(self, rows_needed):
        if rows_needed > self.rows.shape[0]:
            grown = torch.full((max(rows_needed, self.rows.shape[0] * 3 // 2), self.width), float("nan"))
            grown[:self.used] = self.rows[:self.used]          # (like the device arena: scratch rows are NOT carried over)
            self.rows = grown
            self.grown += 1

    def write(self, start, staged, keep):
        n = staged.shape[0]
        self.ensure(start + n)
        self.rows[start:start + n] = staged
        if keep:
            assert start == self.used
            self.used += n

    def pad_gather(self, table, n_utts, t_max, batch_first, want_mask):
        out = torch.zeros((n_utts, t_max, self.width))
        mask = torch.zeros((n_utts, t_max, 1))
        for b in range(n_utts):
            s, n = int(table[0, b]), int(table[1, b])
            out[b, :n] = self.rows[s:s + n]
            mask[b, :n] = 1.0
        if not batch_first:
            out, mask = out.transpose(0, 1).contiguous(), mask.transpose(0, 1).contiguous()
        return out, (mask if want_mask else None)


class MemReader(ReaderBase):
    def __init__(self, name, store, **kw):
        self._configure(name, **kw)
        self.store = store
        self.loads = 0

    def load(self, id_name):
        self.loads += 1
        return self.store[id_name]

    def preprocess_sample(self, sample):
        return sample


def _dataset(n=23, min_frames=None, with_ragged=False, with_int=False):
    rng = np.random.default_rng(3)
    ids = ["u%02d" % i for i in range(n)]
    lens = {i: 2 + (7 * k) % 9 for k, i in enumerate(ids)}
    x = {i: rng.standard_normal((lens[i] + 1, 5)).astype(np.float32) for i in ids}     # one frame longer: trimmed
    y = {i: rng.standard_normal((lens[i], 3)).astype(np.float32) for i in ids}
    readers = [MemReader("x", x, match_length="y", min_frames=min_frames),
               MemReader("y", y, requires_seq_mask=True, min_frames=min_frames)]
    if with_int:
        readers.append(MemReader("dur", {i: np.arange(lens[i], dtype=np.int64)[:, None] for i in ids}))
    if with_ragged:
        readers.append(MemReader("att", {i: rng.standard_normal((lens[i], 1 + k % 4)).astype(np.float32)
                                         for k, i in enumerate(ids)}, other_pad_dims=[1]))
    return PyTorchDatareadersDataset(ids, readers), readers


def _reference_epochs(ds, batch_size, shuffle, batch_first, epochs, seed, common_divisor=1, shard=None):
    torch.manual_seed(seed)
    extra = {"shard": shard} if shard else {}
    loader = DataLoader(ds, batch_size=batch_size, shuffle=shuffle, num_workers=0,
                        collate_fn=partial(H.prepare_batch, common_divisor=common_divisor, batch_first=batch_first,
                                           **extra))
    return [[b for b in loader] for _ in range(epochs)], torch.rand(2)


def _cached_epochs(ds, batch_size, shuffle, batch_first, epochs, seed, **kw):
    torch.manual_seed(seed)
    loader = CachedBatchLoader(ds, batch_size, shuffle, "cpu", batch_first, arena_factory=HostArena,
                               host_collate=H.prepare_batch, **kw)
    return [[b for b in loader] for _ in range(epochs)], torch.rand(2), loader


def _assert_same(ref, got):
    assert len(ref) == len(got)
    for e_ref, e_got in zip(ref, got):
        assert len(e_ref) == len(e_got)
        for (d0, l0), (d1, l1) in zip(e_ref, e_got):
            assert list(d0.keys()) == list(d1.keys())
            assert list(l0.keys()) == list(l1.keys())
            for k in d0:
                if torch.is_tensor(d0[k]):
                    assert d0[k].dtype == d1[k].dtype and d0[k].shape == d1[k].shape, k
                    assert torch.equal(d0[k], d1[k]), k
                else:
                    assert d0[k] == d1[k], k
            for k in l0:
                assert torch.equal(torch.as_tensor(l0[k]), torch.as_tensor(l1[k])), k


@pytest.mark.parametrize("batch_first", [False, True])
@pytest.mark.parametrize("shuffle", [False, True])
@pytest.mark.parametrize("threads", [0, 3])
def test_cached_batches_equal_prepare_batch_over_epochs(batch_first, shuffle, threads):
    ds, readers = _dataset()
    ref, ref_next = _reference_epochs(ds, 4, shuffle, batch_first, 3, seed=11)
    loads_ref = [r.loads for r in readers]
    ds2, readers2 = _dataset()
    got, got_next, loader = _cached_epochs(ds2, 4, shuffle, batch_first, 3, seed=11, threads=threads)
    _assert_same(ref, got)
    assert torch.equal(ref_next, got_next)              # the same draws from the global generator
    # every utterance was read once, not once per epoch
    assert [r.loads for r in readers2] == [l // 3 for l in loads_ref]
    assert loader.stats == {"hits": 2 * len(ds), "misses": len(ds), "passed_through": 0, "host_tier": 0}
    if shuffle:
        assert not all(torch.equal(a[0]["x"], b[0]["x"]) for a, b in zip(ref[0], ref[1]) if a[0]["x"].shape == b[0]["x"].shape)


def test_min_frames_other_dtypes_and_ragged_streams():
    """`min_frames` pads the batch up (constant mode: zeros); an int64 stream and a stream padded in a second
    dimension stay on the host road (prepare_batch itself) inside the same batches"""
    ds, _ = _dataset(min_frames=14, with_ragged=True, with_int=True)
    ref, _ = _reference_epochs(ds, 5, True, False, 2, seed=2)
    ds2, _ = _dataset(min_frames=14, with_ragged=True, with_int=True)
    got, _, loader = _cached_epochs(ds2, 5, True, False, 2, seed=2)
    _assert_same(ref, got)
    assert loader._kind == {"x": "arena", "_id_list": "ids", "y": "arena", "dur": "host", "att": "host"}
    assert got[0][0][0]["y"].shape[0] == 14


def test_budget_passes_the_rest_through_scratch_rows():
    ds, _ = _dataset()
    ref, _ = _reference_epochs(ds, 4, True, True, 3, seed=5)
    ds2, readers2 = _dataset()
    per_frame = 4 * (5 + 3)
    budget = per_frame * 40                                  # room for a handful of utterances
    got, _, loader = _cached_epochs(ds2, 4, True, True, 3, seed=5, byte_budget=budget, host_byte_budget=0)
    _assert_same(ref, got)
    assert 0 < loader._cached.sum() < len(ds2)
    assert loader.cached_bytes() <= budget
    assert loader.stats["passed_through"] > 0 and loader.stats["host_tier"] == 0
    assert readers2[0].loads > len(ds2)                      # the passed-through ones are read again
    # kept rows never moved: an arena never grows beyond what the budget pays for plus one batch of scratch
    assert all(a.used * a.width * 4 <= budget for a in loader._arenas.values())


def test_page_locked_second_tier_between_device_and_files():
    """device budget for a few utterances, host budget for some more, the rest passed through: every batch equal to
    prepare_batch's; an utterance of the first two tiers is read once, only the last tier again"""
    ds, _ = _dataset()
    ref, _ = _reference_epochs(ds, 4, True, False, 3, seed=9)
    ds2, readers2 = _dataset()
    per_frame = 4 * (5 + 3)
    got, _, loader = _cached_epochs(ds2, 4, True, False, 3, seed=9, byte_budget=per_frame * 30,
                                    host_byte_budget=per_frame * 45, threads=2)
    _assert_same(ref, got)
    n_dev = int((loader._cached & ~loader._on_host).sum())
    n_host = int(loader._on_host.sum())
    assert n_dev > 0 and n_host > 0 and n_dev + n_host < len(ds2)
    assert loader.host_cached_bytes() <= per_frame * 45 and loader.cached_bytes() <= per_frame * 30
    assert loader.stats["host_tier"] >= 2 * n_host - 1           # used from page-locked memory in epochs 2 and 3
    assert loader.stats["passed_through"] == 3 * (len(ds2) - n_dev - n_host)
    # all of it in page-locked memory: nothing read twice
    ds3, readers3 = _dataset()
    got, _, loader = _cached_epochs(ds3, 4, True, False, 3, seed=9, byte_budget=0, host_byte_budget=1 << 40)
    _assert_same(ref, got)
    assert loader._on_host.all() and loader.stats["passed_through"] == 0
    assert readers3[1].loads <= 2 * len(ds3)                 # (one load per utterance, plus get_length's first look)


def test_data_parallel_shards_and_remainder():
    """prepare_batch's remainder drop and `shard` selection (:392-395, the handler's one-process-per-GPU split): two
    ranks over the same draws hold complementary halves of every global batch"""
    for rank in range(2):
        ds, _ = _dataset(n=22)
        ref, _ = _reference_epochs(ds, 5, True, False, 2, seed=7, common_divisor=2, shard=(rank, 2))
        ds2, _ = _dataset(n=22)
        got, _, loader = _cached_epochs(ds2, 5, True, False, 2, seed=7, common_divisor=2, shard=(rank, 2))
        _assert_same(ref, got)
        assert [len(d["_id_list"]) for d, _ in got[0]] == [2, 2, 2, 2, 1]


def test_arena_growth_keeps_rows():
    """a first batch of short utterances under-estimates the arena: it grows, kept rows stay"""
    ds, _ = _dataset(n=40)
    order = np.argsort([len(ds.datareaders[1].store[i]) for i in ds.id_list])
    ds.id_list = [ds.id_list[i] for i in order]              # shortest first
    ref, _ = _reference_epochs(ds, 4, False, True, 2, seed=1)
    got, _, loader = _cached_epochs(ds, 4, False, True, 2, seed=1)
    _assert_same(ref, got)
    assert any(a.grown for a in loader._arenas.values())


def test_early_exit_and_reuse():
    ds, _ = _dataset()
    loader = CachedBatchLoader(ds, 4, False, "cpu", True, arena_factory=HostArena, host_collate=H.prepare_batch,
                               threads=2)
    assert len(loader) == 6
    for k, _ in enumerate(loader):
        if k == 1:
            break
    assert sum(1 for _ in loader) == 6
    assert loader._cached.all()
