"""MI355X-native training step of the feed-forward acoustic model (BASELINE configs 1/2).

What the reference does per mini-batch in ModularModelHandlerPyTorch.process_dataloader
(:745-820) for an `RNNDYN-2_TANH_512-1_FC_187` model -- FFWrapper forward (FFWrapper.py:63-73),
NamedLoss masked MSE 'mean_per_frame' (NamedLoss.py:70-117), backward, Adam -- is done here on
*packed valid frames* (FF layers are frame independent, so padding is never computed) with all
parameters, gradients and Adam moments in ONE flat fp32 buffer each:

  fwd   3 x fp32-MFMA GEMM with fused bias+tanh
  loss  fused masked-MSE + dLoss/dPred
  bwd   dX GEMMs with the previous layer's tanh' fused in the epilogue, dW GEMMs with
        deterministic split-K slabs, bias column sums
  DP    one all-reduce(sum) of the flat gradient buffer over RCCL (torch.distributed 'nccl');
        local gradients are already divided by the GLOBAL valid-frame count, so the result equals
        the single-GPU step on the concatenated batch
  opt   one fused Adam launch over the flat buffers
"""
import math

import torch

from . import ops


def _pad4(n):
    return (n + 3) // 4 * 4


class FlatFFModel:
    """Dense stack dims[0] -> dims[1] -> ... with activations `acts` (one per layer)."""

    def __init__(self, dims=(425, 512, 512, 187), acts=("tanh", "tanh", None), device="cuda",
                 seed=0, state_dict=None):
        assert len(acts) == len(dims) - 1
        self.dims = tuple(int(d) for d in dims)
        self.acts = [ops.ACT_BY_NAME[a] for a in acts]
        self.device = torch.device(device)
        # Row pitch of every weight matrix (and of the packed input) is padded to a multiple of
        # 4 floats so all GEMM operands take the 16-byte load path; pad columns stay exactly zero
        # (their gradient is dz^T * 0 = 0, Adam leaves a zero parameter with zero moments at zero).
        self.in_pitch = _pad4(self.dims[0])
        self.layout = []  # (w_off, b_off, N, K, Kpitch)
        off = 0
        for li, (K, N) in enumerate(zip(self.dims[:-1], self.dims[1:])):
            kp = _pad4(K) if li == 0 else K
            w_off = off
            off += _pad4(N * kp)
            b_off = off
            off += _pad4(N)
            self.layout.append((w_off, b_off, N, K, kp))
        self.numel = off
        self.params = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.step_count = 0
        self._buffers = {}
        if state_dict is None:
            state_dict = self.reference_init(self.dims, seed)
        self.load_layers(state_dict)

    # torch.nn.Linear default init in the order the reference constructs the layers
    @staticmethod
    def reference_init(dims, seed):
        g = torch.Generator().manual_seed(seed)
        layers = []
        for K, N in zip(dims[:-1], dims[1:]):
            bound = 1.0 / math.sqrt(K)
            w = (torch.rand(N, K, generator=g) * 2 - 1) * bound
            b = (torch.rand(N, generator=g) * 2 - 1) * bound
            layers.append((w, b))
        return layers

    @staticmethod
    def from_module(model, device=None):
        """Mirror of a drop-in module stack (NamedForwardWrapper / RNNDyn) that consists of Linear
        groups only: same weights, flat buffers.  Returns None when the model has anything else
        (recurrent groups, dropout) -- those train through the module path."""
        from .nn.modules import LinearAct
        inner = getattr(model, "model", model)
        layers, acts = [], []
        for group in getattr(inner, "layer_groups", []):
            seq = getattr(group, "module", None)
            if not isinstance(seq, torch.nn.Sequential):
                return None
            for m in seq:
                if isinstance(m, LinearAct):
                    layers.append((m.weight.detach(), m.bias.detach()))
                    acts.append({ops.ACT_NONE: None, ops.ACT_TANH: "tanh", ops.ACT_RELU: "relu"}[m.act])
                elif isinstance(m, torch.nn.Dropout):
                    return None
                elif type(m).__name__ != "FusedActivation":
                    return None
        if not layers:
            return None
        dims = [layers[0][0].shape[1]] + [w.shape[0] for w, _ in layers]
        device = device if device is not None else layers[0][0].device
        return FlatFFModel(dims, acts, device=device, state_dict=layers)

    def store_to_module(self, model, buf=None):
        """Writes the flat parameters (or another buffer of the same layout, e.g. the EMA shadow)
        back into the module stack (checkpoints, inference)."""
        from .nn.modules import LinearAct
        inner = getattr(model, "model", model)
        mods = [m for g in inner.layer_groups for m in g.module if isinstance(m, LinearAct)]
        with torch.no_grad():
            for i, m in enumerate(mods):
                m.weight.copy_(self.weight(i, buf))
                m.bias.copy_(self.bias(i, buf))
        return mods

    def load_layers(self, layers):
        for i, (w, b) in enumerate(layers):
            self.weight(i).copy_(w.to(self.device))
            self.bias(i).copy_(b.to(self.device))

    def weight_padded(self, i, buf=None):
        """[N, Kpitch] contiguous storage of layer i (pad columns are zero)."""
        w_off, _, N, K, kp = self.layout[i]
        return (self.params if buf is None else buf)[w_off:w_off + N * kp].view(N, kp)

    def weight(self, i, buf=None):
        """[N, K] view with the reference's state-dict shape (strided when K is padded)."""
        K = self.layout[i][3]
        return self.weight_padded(i, buf)[:, :K]

    def bias(self, i, buf=None):
        _, b_off, N, _, _ = self.layout[i]
        return (self.params if buf is None else buf)[b_off:b_off + N]

    def pack_input(self, x):
        """[M, dims[0]] -> [M, in_pitch] with zero pad columns (done once per batch at collate
        time; a loader can write into the padded buffer directly)."""
        if x.shape[1] == self.in_pitch:
            return x
        xp = torch.zeros((x.shape[0], self.in_pitch), dtype=torch.float32, device=x.device)
        xp[:, :x.shape[1]] = x
        return xp

    def layers(self):
        return [(self.weight(i).clone(), self.bias(i).clone()) for i in range(len(self.layout))]

    # ------------------------------------------------------------------------------------
    def _rows_buffer(self, name, M, width):
        """[M, width] view of a persistent zero-initialised buffer whose row pitch is padded to a
        multiple of 4 floats (16-byte GEMM loads); the pad columns are never written, so they
        stay zero.  One buffer per name, grown to the largest M seen."""
        pitch = _pad4(width)
        buf = self._buffers.get(name)
        if buf is None or buf.shape[0] < M:
            buf = torch.zeros((M, pitch), dtype=torch.float32, device=self.device)
            self._buffers[name] = buf
        return buf[:M, :width]

    def forward(self, x, n_layers=None):
        """x [M, dims[0]] fp32 packed frames -> list of layer outputs (of the first n_layers)."""
        acts = []
        h = self.pack_input(x)
        M = h.shape[0]
        for i in range(len(self.layout) if n_layers is None else n_layers):
            h = ops.linear_fwd(h, self.weight_padded(i), self.bias(i), self.acts[i],
                               out=self._rows_buffer("h%d" % i, M, self.layout[i][2]))
            acts.append(h)
        return acts

    def loss_and_backward(self, x, target, row_valid, n_valid_global, reduce_group=None, reduce=False):
        """Fills self.grads with d(loss)/d(params) of this rank's frames; returns loss tensor
        (this rank's contribution, already divided by the global frame count).  reduce=True (data
        parallel): the sum all-reduce of a layer's gradient segment is issued as soon as its weight
        gradient is queued -- it runs on the collective's own stream under the remaining GEMMs --
        and the handles are left in self._pending for train_step to wait on before Adam."""
        self._pending = []
        x = self.pack_input(x)
        if reduce:
            return self._loss_and_backward(x, target, row_valid, n_valid_global, reduce_group, reduce)
        # one process: the loss sum and the layers' split-K slab reductions are queued and run as ONE
        # launch at the end (five launches less per step); with data parallelism every layer's
        # gradient is reduced at once, because its all-reduce starts right behind it
        with ops.deferred_reductions():
            return self._loss_and_backward(x, target, row_valid, n_valid_global, reduce_group, False)

    def _loss_and_backward(self, x, target, row_valid, n_valid_global, reduce_group, reduce):
        M = x.shape[0]
        n = len(self.layout)
        if self.acts[-1] in (None, ops.ACT_NONE) and n > 1:
            # the output layer never materialises: its GEMM epilogue forms the masked difference
            # to the target, the loss partial sums and d loss / d output
            hs = self.forward(x, n_layers=n - 1)
            loss, dz = ops.linear_fwd_mse(hs[-1], self.weight_padded(n - 1), self.bias(n - 1), target,
                                          row_valid, n_valid_global,
                                          grad=self._rows_buffer("dz_out", M, self.dims[-1]))
            hs.append(None)
        else:
            hs = self.forward(x)
            loss, dz = ops.masked_mse(hs[-1], target, row_valid, n_valid_global,
                                      grad=self._rows_buffer("dz_out", M, self.dims[-1]))
        for i in range(n - 1, -1, -1):
            inp = hs[i - 1] if i > 0 else x
            dz_in = None
            if i > 0:
                # dW, db and the input gradient of the layer in one call (one launch)
                dz_in = self._rows_buffer("dz%d" % (i & 1), M, self.layout[i][3])
                ops.linear_bwd(dz, inp, self.weight_padded(i), self.weight_padded(i, self.grads),
                               self.bias(i, self.grads), dz_in, yprev=hs[i - 1], act_prev=self.acts[i - 1])
            else:
                ops.linear_bwd_weight(dz, inp, dw=self.weight_padded(i, self.grads),
                                      db=self.bias(i, self.grads))
            if reduce:
                from .parallel import allreduce_flat_
                w_off, b_off, N, _, _ = self.layout[i]
                work = allreduce_flat_(self.grads[w_off:b_off + _pad4(N)], reduce_group, async_op=True)
                if work is not None:
                    self._pending.append(work)
            if i > 0:
                dz = dz_in
        return loss

    def train_step(self, x, target, row_valid, n_valid_global, lr=1e-3, betas=(0.9, 0.999),
                   eps=1e-8, weight_decay=0.0, process_group=None, world_size=1,
                   clip_norm_kind=None, clip_max_norm=0.0, clip_value=None, ema_shadow=None,
                   ema_decay=0.0):
        """forward + masked MSE + backward + [all-reduce] + optimiser tail.  clip_norm_kind 2 / 0
        (infinity) clips the gradient norm to clip_max_norm, clip_value clamps the elements,
        ema_shadow (flat, this model's layout) is updated with ema_decay -- all inside the one
        Adam pass (itts_adam_step_fused)."""
        from . import parallel
        dist_on = world_size > 1 or parallel._active(process_group)
        loss = self.loss_and_backward(x, target, row_valid, n_valid_global, reduce_group=process_group,
                                      reduce=dist_on)
        for work in self._pending:   # the optimiser's stream waits for the collectives
            work.wait()
        self._pending = []
        self.step_count += 1
        if clip_norm_kind is None and not clip_value and ema_shadow is None:
            ops.adam_step(self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.step_count,
                          lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
            return loss
        accum = None
        if clip_norm_kind is not None:
            if getattr(self, "_norm_accum", None) is None:
                self._norm_accum = torch.zeros(1, dtype=torch.float32, device=self.device)
            accum = ops.grad_norm_accum(self.grads, self._norm_accum, clip_norm_kind)
        ops.adam_step_fused(self.params, self.grads, self.exp_avg, self.exp_avg_sq,
                            self.step_count, lr=lr, betas=betas, eps=eps,
                            weight_decay=weight_decay, norm_accum=accum,
                            norm_kind=clip_norm_kind if clip_norm_kind is not None else 2,
                            clip_max_norm=clip_max_norm, clip_value=clip_value or 0.0,
                            ema_shadow=ema_shadow, ema_decay=ema_decay)
        return loss


def flops_per_frame(dims, skip_first_dgrad=True):
    """fwd + bwd FLOPs per valid frame: 2*N*K fwd, 2*N*K dW, 2*N*K dX (not for layer 0)."""
    f = 0
    for i, (K, N) in enumerate(zip(dims[:-1], dims[1:])):
        f += 2 * N * K * (2 if (i == 0 and skip_first_dgrad) else 3)
    return f
