"""Packed, pre-normalised frame shards resident in HBM -- the input pipeline SURVEY.md section 8(f)
row 1 asks for.  The reference loads one `.npz` / raw file per utterance and stream in every
__getitem__ (NpzDataReader.load :312-345), normalises it on the host (:347-371) and pads the
mini-batch (ModularModelHandlerPyTorch.prepare_batch :388-465); at 29 M frames/s on the device
that host loop is the bottleneck by three orders of magnitude.  A shard holds what those calls
produce for a whole id list, once:

    x [N, in_pitch]  float32   inputs after the readers' preprocess_sample and length matching,
                               utterances back to back, row pitch padded to a multiple of 4 floats
    y [N, out_pitch] float32   targets, same rows
    offsets [U + 1]  int64     first row of every utterance;  ids [U]

LJSpeech-size data (17 M frames x (428 + 188) floats = 42 GB) fits the 288 GB of one MI355X several
times over, so a shard is uploaded once and every epoch gathers its mini-batches (packed valid
frames: feed-forward layers are frame independent, padding is never materialised) on the device.

File layout (`save` / `load`): b"ITTSHRD1", u64 header length, UTF-8 JSON header, zero padding to a
4096-byte boundary, then x, y, offsets as raw little-endian arrays (np.memmap-able)."""
import json
import os

import numpy as np
import torch

_MAGIC = b"ITTSHRD1"
_ALIGN = 4096


def _pad4(n):
    return (n + 3) // 4 * 4


class FrameShard(object):

    def __init__(self, x, y, offsets, ids, in_dim, out_dim, meta=None):
        self.x, self.y = x, y                       # [N, pitch] float32 (numpy or torch)
        self.offsets = np.asarray(offsets, dtype=np.int64)
        self.ids = list(ids)
        self.in_dim, self.out_dim = int(in_dim), int(out_dim)
        self.meta = dict(meta or {})

    # ------------------------------------------------------------------------------ building
    @staticmethod
    def from_dataset(dataset, input_name, target_name, meta=None, threads=0):
        """Runs the data readers once per id (load, normalise, symmetric length matching: exactly
        what a training step would have seen, PyTorchDatareadersDataset.get_id_name) and packs the
        results.  threads > 0: the items are read by that many threads (order kept; only for datasets whose
        items draw no random numbers -- the caller's business, as in ModularModelHandlerPyTorch._get_dataloader)."""
        xs, ys = [], []
        if threads > 0 and len(dataset) > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=int(threads), thread_name_prefix="itts_shard") as pool:
                items = list(pool.map(dataset.__getitem__, range(len(dataset))))
        else:
            items = None
        for i in range(len(dataset)):
            item, _ = items[i] if items is not None else dataset[i]
            xi, yi = item[input_name], item[target_name]
            if len(xi) != len(yi):
                raise ValueError("{}: {} and {} differ in length ({} vs {}); give the readers a "
                                 "match_length".format(dataset.id_list[i], input_name, target_name,
                                                       len(xi), len(yi)))
            xs.append(np.asarray(xi, dtype=np.float32))
            ys.append(np.asarray(yi, dtype=np.float32))
        return FrameShard.from_arrays(xs, ys, dataset.id_list, meta)

    @staticmethod
    def from_arrays(xs, ys, ids, meta=None):
        lengths = [len(a) for a in xs]
        offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        in_dim, out_dim = xs[0].shape[1], ys[0].shape[1]
        x = np.zeros((offsets[-1], _pad4(in_dim)), dtype=np.float32)
        y = np.zeros((offsets[-1], _pad4(out_dim)), dtype=np.float32)
        for a, b, o in zip(xs, ys, offsets[:-1]):
            x[o:o + len(a), :in_dim] = a
            y[o:o + len(b), :out_dim] = b
        return FrameShard(x, y, offsets, ids, in_dim, out_dim, meta)

    # ------------------------------------------------------------------------------- file I/O
    def save(self, path):
        x = self.x.cpu().numpy() if torch.is_tensor(self.x) else self.x
        y = self.y.cpu().numpy() if torch.is_tensor(self.y) else self.y
        header = json.dumps({"n_frames": int(x.shape[0]), "in_dim": self.in_dim,
                             "in_pitch": int(x.shape[1]), "out_dim": self.out_dim,
                             "out_pitch": int(y.shape[1]), "ids": self.ids,
                             "meta": self.meta}).encode("utf-8")
        with open(path, "wb") as f:
            f.write(_MAGIC)
            f.write(np.uint64(len(header)).tobytes())
            f.write(header)
            f.write(b"\0" * (-f.tell() % _ALIGN))
            for a in (x, y, self.offsets):
                f.write(np.ascontiguousarray(a).tobytes())
        return path

    @staticmethod
    def load(path, device=None, mmap=True):
        """Reads a shard; with `device` the frame matrices are uploaded (one copy each)."""
        with open(path, "rb") as f:
            if f.read(8) != _MAGIC:
                raise ValueError("{} is not a frame shard".format(path))
            hlen = int(np.frombuffer(f.read(8), dtype=np.uint64)[0])
            h = json.loads(f.read(hlen).decode("utf-8"))
            start = f.tell() + (-f.tell() % _ALIGN)
        n, ip, op, u = h["n_frames"], h["in_pitch"], h["out_pitch"], len(h["ids"])
        opener = np.memmap if mmap else (lambda p, dtype, mode, offset, shape:
                                         np.fromfile(p, dtype=dtype, offset=offset,
                                                     count=int(np.prod(shape))).reshape(shape))
        x = opener(path, dtype=np.float32, mode="r", offset=start, shape=(n, ip))
        y = opener(path, dtype=np.float32, mode="r", offset=start + n * ip * 4, shape=(n, op))
        offsets = np.array(opener(path, dtype=np.int64, mode="r",
                                  offset=start + n * (ip + op) * 4, shape=(u + 1,)))
        shard = FrameShard(x, y, offsets, h["ids"], h["in_dim"], h["out_dim"], h.get("meta"))
        return shard.to(device) if device is not None else shard

    def to(self, device, chunk_rows=1 << 20):
        """Uploads the frame matrices (in chunks: a memory-mapped shard never has to fit in host
        memory a second time)."""
        def up(a):
            if torch.is_tensor(a):
                return a.to(device)
            out = torch.empty(a.shape, dtype=torch.float32, device=device)
            for i in range(0, a.shape[0], chunk_rows):
                out[i:i + chunk_rows].copy_(torch.from_numpy(np.array(a[i:i + chunk_rows])))
            return out
        return FrameShard(up(self.x), up(self.y), self.offsets, self.ids, self.in_dim,
                          self.out_dim, self.meta)

    # ------------------------------------------------------------------------------ batching
    def __len__(self):
        return len(self.ids)

    @property
    def lengths(self):
        return np.diff(self.offsets)

    def gather(self, utt_indices):
        """Packed valid frames of the given utterances: (x [M, in_pitch], y [M, out_dim] view of a
        [M, out_pitch] buffer, lengths).  On the device: one upload of the batch's (start, start in the batch, length)
        table and one `itts_batch_concat_rows_f32` launch per matrix (until round 6: a row index of M entries built
        with numpy per step, uploaded, two index_select)."""
        utt_indices = np.asarray(utt_indices, dtype=np.int64)
        lens = self.offsets[utt_indices + 1] - self.offsets[utt_indices]
        if torch.is_tensor(self.x) and self.x.is_cuda and len(utt_indices) > 0:
            from idiaptts_amd import ops
            table = torch.empty((3, len(utt_indices)), dtype=torch.int64, pin_memory=True)
            tab = table.numpy()
            tab[0] = self.offsets[utt_indices]
            tab[1, 0] = 0
            np.cumsum(lens[:-1], out=tab[1, 1:])
            tab[2] = lens
            dev_table = table.to(self.x.device, non_blocking=True)
            m, t_max = int(lens.sum()), int(lens.max())
            x = ops.batch_concat_rows(self.x, dev_table, m, t_max)
            y = ops.batch_concat_rows(self.y, dev_table, m, t_max)
            return x, y[:, :self.out_dim], lens
        starts = np.repeat(self.offsets[utt_indices] - np.concatenate([[0], np.cumsum(lens)[:-1]]),
                           lens)
        rows = starts + np.arange(int(lens.sum()), dtype=np.int64)
        if torch.is_tensor(self.x):
            idx = torch.from_numpy(rows).to(self.x.device)
            return self.x.index_select(0, idx), self.y.index_select(0, idx)[:, :self.out_dim], lens
        return self.x[rows], self.y[rows][:, :self.out_dim], lens
