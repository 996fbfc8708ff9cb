"""Model handler: batching, train / eval step, checkpoints -- the subset of the reference's
ModularModelHandlerPyTorch (neural_networks/pytorch/ModularModelHandlerPyTorch.py) that the
acoustic-model hot path exercises: prepare_batch / sequence_mask / unsorted_pad_sequence
(:388-499), the body of process_dataloader for one mini-batch (:745-831), save_checkpoint /
load_checkpoint with the reference's file layout (:71-262: `params_<suffix>`,
`optimiser_<suffix>`, suffix in {e<N>, s<N>, best, last}; each `torch.save({'params':
state_dict, 'epoch', 'step'})`).  The optimiser is a fused HIP Adam (one launch per tensor).
"""
import os
import re

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

from idiaptts_amd import ops


class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no amsgrad) through itts_adam_step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                state['step'] += 1
                ops.adam_step(p.data.view(-1), p.grad.contiguous().view(-1),
                              state['exp_avg'].view(-1), state['exp_avg_sq'].view(-1),
                              state['step'], lr=group['lr'], betas=group['betas'],
                              eps=group['eps'], weight_decay=group['weight_decay'])


class ModularModelHandlerPyTorch(object):

    def __init__(self):
        self.model = None
        self.optimiser = None
        self.losses = []
        self.model_config = None

    # ----------------------------------------------------------------------------- batching
    @staticmethod
    def sequence_mask(sequence_length, max_len=None, batch_first=False):
        """[B, T, 1] / [T, B, 1] float mask, 1 where t < length (reference :467-491)."""
        sequence_length = torch.as_tensor(sequence_length)
        if max_len is None:
            max_len = int(sequence_length.max())
        t = torch.arange(0, int(max_len), dtype=torch.long)
        length = sequence_length.to(torch.long)
        m = (t.unsqueeze(0) < length.unsqueeze(1)) if batch_first else \
            (t.unsqueeze(1) < length.unsqueeze(0))
        return m.unsqueeze(-1).contiguous().float()

    @staticmethod
    def unsorted_pad_sequence(sequence, batch_first):
        sequence = [torch.from_numpy(s) if isinstance(s, np.ndarray) else s for s in sequence]
        return pad_sequence(sequence, batch_first)

    @staticmethod
    def prepare_batch(batch, common_divisor=1, batch_first=False, mask_keys=()):
        """List of {name: array [T_i, D]} dicts -> (data, temporal lengths). The remainder that
        is not divisible by `common_divisor` (# GPUs) is dropped first (reference :392-395);
        every key in `mask_keys` also gets `<key>_mask`."""
        assert len(batch) >= common_divisor
        remainder = len(batch) % common_divisor
        if remainder > 0:
            batch = batch[:-remainder]
        data, lengths = dict(), dict()
        for key in batch[0].keys():
            values = [b[key] for b in batch if key in b]
            if key == "_id_list":
                data[key] = list(values)
                continue
            lengths[key] = torch.tensor([x.shape[0] for x in values], dtype=torch.long)
            if key in mask_keys:
                data[key + "_mask"] = ModularModelHandlerPyTorch.sequence_mask(
                    lengths[key], int(lengths[key].max()), batch_first=batch_first)
                lengths[key + "_mask"] = lengths[key]
            data[key] = ModularModelHandlerPyTorch.unsorted_pad_sequence(values, batch_first)
        return data, lengths

    # -------------------------------------------------------------------------------- model
    def create_model(self, config, use_gpu=True):
        self.model_config = config
        self.model = config.create_model()
        if use_gpu:
            self.model = self.model.cuda()
        return self.model

    def set_optimiser(self, optimiser_type="Adam", **optimiser_args):
        if optimiser_type != "Adam":
            raise NotImplementedError("Only Adam has a fused HIP kernel so far.")
        self.optimiser = HipAdam(self.model.parameters(), **optimiser_args)

    def set_losses(self, loss_configs):
        self.losses = [c.create_loss() for c in loss_configs]

    def _to_device(self, data, device):
        return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v)
                for k, v in data.items()}

    def process_batch(self, data, lengths, step, training=True, grad_clip_norm=None):
        """One iteration of process_dataloader (:745-831): forward, losses, backward, clip, step."""
        device = next(self.model.parameters()).device
        data = self._to_device(data, device)
        max_lengths = {k: int(v.max()) for k, v in lengths.items()}
        batch_dim = 0 if self.model.batch_first else 1
        B = data[self.model.input_names[0]].shape[batch_dim]
        self.model.init_hidden(B)
        ctx = torch.enable_grad() if training else torch.no_grad()
        with ctx:
            self.model(data, lengths, max_lengths)
            loss_dict = {}
            for loss_fn in self.losses:
                loss_dict.update(loss_fn(data, lengths, step))
            total = sum(loss_dict.values())
            if torch.isnan(total):
                raise ValueError("Found NaN in loss.")       # reference :778-781
            if training:
                self.optimiser.zero_grad()
                total.backward()
                if grad_clip_norm is not None:
                    torch.nn.utils.clip_grad_norm_(self.model.parameters(), grad_clip_norm)
                self.optimiser.step()
        return {k: float(v.detach()) for k, v in loss_dict.items()}, data

    # -------------------------------------------------------------------------- checkpoints
    @staticmethod
    def _suffix(epoch=None, step=None, best=False, last=False):
        if best:
            return "best"
        if last:
            return "last"
        return "e{}".format(epoch) if epoch is not None else "s{}".format(step)

    def save_checkpoint(self, model_path, epoch=None, step=None, best=False, last=False,
                        save_optimiser=True):
        os.makedirs(model_path, exist_ok=True)
        sfx = self._suffix(epoch, step, best, last)
        torch.save({"params": self.model.state_dict(), "epoch": epoch, "step": step},
                   os.path.join(model_path, "params_" + sfx))
        if save_optimiser and self.optimiser is not None:
            torch.save({"params": self.optimiser.state_dict(), "epoch": epoch, "step": step},
                       os.path.join(model_path, "optimiser_" + sfx))

    def load_checkpoint(self, model_path, epoch=None, step=None, best=False, last=False,
                        ignore_layers=None, load_optimiser=False):
        sfx = self._suffix(epoch, step, best, last)
        ckpt = torch.load(os.path.join(model_path, "params_" + sfx), map_location="cpu",
                          weights_only=False)
        params = ckpt["params"]
        if ignore_layers:  # regex list (reference :285-309)
            pats = [re.compile(p) for p in ignore_layers]
            own = self.model.state_dict()
            params = {k: (own[k] if any(p.fullmatch(k) or p.match(k) for p in pats) else v)
                      for k, v in params.items()}
        missing, unexpected = self.model.load_state_dict(params, strict=False)
        if load_optimiser and self.optimiser is not None:
            path = os.path.join(model_path, "optimiser_" + sfx)
            if os.path.isfile(path):
                self.optimiser.load_state_dict(torch.load(path, map_location="cpu",
                                                          weights_only=False)["params"])
        return ckpt.get("epoch"), ckpt.get("step"), missing, unexpected
