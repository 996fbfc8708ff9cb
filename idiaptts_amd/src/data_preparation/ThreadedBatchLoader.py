"""Batches prepared by a pool of THREADS a few batches ahead of the consumer -- what the reference gets from
`DataLoader(num_workers=4)` (model_trainers/ModularTrainer.py:831-841, ExtendedHParams.py:191), without the worker
processes: on this stack forked workers have to ship every batch (3 MB per utterance) back through shared memory and
are re-forked from a process with an initialised GPU every epoch (measured, 1 024 utterances: 6.5 s an epoch with four
workers against 2.5 s with none).  The readers' work -- file reads, numpy normalisation, torch padding -- releases the
interpreter lock, so threads overlap it with each other and with the training steps.

The batches, their order and the random numbers drawn from torch's global generator are exactly those of
`DataLoader(dataset, batch_size, shuffle, generator=generator, num_workers=0)`: the indices come from such a loader over
`range(len(dataset))`, only `dataset[i]` and the collate function run in the pool.  Datasets whose items draw random
numbers themselves (a reader with `max_frames` and `random_select`) must not come here -- the draws would interleave
in thread order; ModularModelHandlerPyTorch._get_dataloader keeps those on the DataLoader."""
import collections
from concurrent.futures import ThreadPoolExecutor

import torch
from torch.utils.data import DataLoader, Dataset


class _Indices(Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


class ThreadedBatchLoader(object):

    def __init__(self, dataset, batch_size, shuffle, collate_fn, threads, generator=None, pin_memory=False,
                 depth=None):
        self.dataset = dataset
        self.batch_size = batch_size
        self.collate_fn = collate_fn
        self.threads = max(1, int(threads))
        self.depth = int(depth) if depth else 2 * self.threads      # batches under way or waiting
        self.pin_memory = bool(pin_memory) and torch.cuda.is_available()
        self._index_loader = DataLoader(_Indices(len(dataset)), batch_size=batch_size, shuffle=shuffle,
                                        generator=generator, collate_fn=list, num_workers=0)
        self._pool = None

    def __len__(self):
        return len(self._index_loader)

    def _build(self, indices):
        batch = self.collate_fn([self.dataset[i] for i in indices])
        if self.pin_memory:
            from torch.utils.data._utils.pin_memory import pin_memory
            batch = pin_memory(batch)
        return batch

    def __iter__(self):
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=self.threads, thread_name_prefix="itts_batch")
        pending = collections.deque()
        index_iter = iter(self._index_loader)      # (draws what a DataLoader draws, when a DataLoader draws it)
        try:
            for indices in index_iter:
                pending.append(self._pool.submit(self._build, indices))
                if len(pending) >= self.depth:
                    yield pending.popleft().result()
            while pending:
                yield pending.popleft().result()
        finally:
            for f in pending:
                f.cancel()

    def __del__(self):
        pool, self._pool = self._pool, None
        if pool is not None:
            pool.shutdown(wait=False)
