"""NamedLoss with the reference's interface (loss/NamedLoss.py:16-131).  The loss on the hot path --
MSELoss(reduction='none') * seq_mask, reduced 'mean_per_frame' (sum over batch and time / total
frames, mean over features) -- is one fused HIP kernel producing the loss and its gradient.  The
other element-wise types (MSELoss, L1Loss) and reductions ('mean_per_sample', 'mean', 'sum',
'none', with or without a sequence mask; reference :113-131) run a second kernel in which the mask
and the reduction are folded into one weight per frame."""
import torch
from torch import nn

from idiaptts_amd import ops
from idiaptts_amd.nn.functional import is_unit_gradient


def _total(lengths):
    """float(sum(lengths)) -- on a tensor without walking it element by element through Python."""
    import torch
    return float(lengths.sum()) if torch.is_tensor(lengths) else float(sum(lengths))


class MaskedMSEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, row_valid, n_valid):
        D = pred.shape[-1]
        p2 = pred.reshape(-1, D)
        t2 = target.reshape(-1, D)
        if p2.stride(-1) != 1:
            p2 = p2.contiguous()
        if t2.stride(-1) != 1:
            t2 = t2.contiguous()
        loss, grad = ops.masked_mse(p2, t2, row_valid, n_valid, want_grad=True)
        ctx.save_for_backward(grad)
        ctx.shape = pred.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        (grad,) = ctx.saved_tensors
        if is_unit_gradient(dloss):                 # the loop's own start of backward: the factor is 1
            return grad.reshape(ctx.shape), None, None, None
        return (grad * dloss).reshape(ctx.shape), None, None, None


class WeightedLossFunction(torch.autograd.Function):
    """sum_r w[r] sum_c e(pred - target) (e: squared / absolute error) or, with `elementwise`, the
    weighted element-wise values themselves (reduction 'none')."""

    @staticmethod
    def forward(ctx, pred, target, row_weight, kind, elementwise):
        D = pred.shape[-1]
        p2 = pred.reshape(-1, D)
        t2 = target.reshape(-1, D)
        if p2.stride(-1) != 1:
            p2 = p2.contiguous()
        if t2.stride(-1) != 1:
            t2 = t2.contiguous()
        loss, grad, elem = ops.weighted_loss(p2, t2, row_weight, kind, want_grad=True,
                                             want_elem=elementwise)
        ctx.save_for_backward(grad)
        ctx.shape = pred.shape
        ctx.elementwise = elementwise
        return elem.reshape(pred.shape) if elementwise else loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        (grad,) = ctx.saved_tensors
        if ctx.elementwise:
            return grad.reshape(ctx.shape) * dloss, None, None, None, None
        if is_unit_gradient(dloss):
            return grad.reshape(ctx.shape), None, None, None, None
        return (grad * dloss).reshape(ctx.shape), None, None, None, None


class NamedLoss(nn.Module):

    class Config:
        def __init__(self, name, type_, input_names, seq_mask=None, batch_first=False,
                     reduction='mean_per_frame', loss_weight=1.0, start_step=0, **kwargs):
            self.name = name
            self.type = type_
            self.input_names = input_names
            self.seq_mask = seq_mask
            self.batch_first = batch_first
            self.reduction = reduction
            self.loss_weight = loss_weight
            self.start_step = start_step
            self.kwargs = kwargs

        def create_loss(self):
            return NamedLoss(self)

    KINDS = {"MSELoss": 0, "L1Loss": 1}
    REDUCTIONS = ("mean_per_frame", "mean_per_sample", "mean", "sum", "none")

    def __init__(self, config):
        super().__init__()
        if config.type not in self.KINDS:
            raise NotImplementedError("Loss type {} has no kernel here (element-wise MSELoss and "
                                      "L1Loss do).".format(config.type))
        if config.reduction not in self.REDUCTIONS:
            raise NotImplementedError("Unknown reduction type {}.".format(config.reduction))
        if config.seq_mask is None and config.reduction in ("mean_per_frame", "mean_per_sample"):
            raise ValueError("Reduction '{}' divides by the sequence lengths: it needs a seq_mask "
                             "(reference NamedLoss.py:114-121).".format(config.reduction))
        self.kind = self.KINDS[config.type]
        self.reduction = config.reduction
        self.name = config.name
        self.input_names = config.input_names
        self.seq_mask = config.seq_mask
        self.batch_first = config.batch_first
        self.loss_weight = config.loss_weight
        self.start_step = config.start_step

    def _row_weight(self, a, mask, length_dict):
        """One weight per (time, batch) position: sequence mask x what the reduction divides by."""
        D = a.shape[-1]
        lead = tuple(a.shape[:-1])                            # [T, B] or [B, T]
        n_rows = 1
        for n in lead:
            n_rows *= int(n)
        if mask is not None:
            w = data_mask = mask.reshape(lead).to(torch.float32)
        else:
            w = data_mask = torch.ones(lead, dtype=torch.float32, device=a.device)
        if self.reduction == "mean_per_frame":
            w = data_mask / (_total(length_dict[self.seq_mask]) * D)
        elif self.reduction == "mean_per_sample":
            lens = torch.as_tensor(length_dict[self.seq_mask], dtype=torch.float32, device=a.device)
            batch_dim = 0 if self.batch_first else 1
            shape = [1] * len(lead)
            shape[batch_dim] = lens.numel()
            w = data_mask / (lens.reshape(shape) * (lens.numel() * D))
        elif self.reduction == "mean":
            w = data_mask / float(n_rows * D)
        return w.reshape(-1).contiguous()

    def forward(self, data, length_dict, step):
        a, b = (data[n] for n in self.input_names)       # (target, prediction)
        weight = 0. if step < self.start_step else self.loss_weight
        if self.kind == 0 and self.reduction == "mean_per_frame":
            mask = data[self.seq_mask]                   # [.., .., 1] float, 1 inside the sequence
            row_valid = (mask.reshape(-1) > 0).view(torch.uint8)      # (bool and uint8 are both one byte of 0 / 1)
            total_num_frames = _total(length_dict[self.seq_mask])
            # MSE is symmetric: differentiate through whichever input needs it
            if b.requires_grad or not a.requires_grad:
                loss = MaskedMSEFunction.apply(b, a.detach(), row_valid, total_num_frames)
            else:
                loss = MaskedMSEFunction.apply(a, b.detach(), row_valid, total_num_frames)
        else:
            w = self._row_weight(a, data[self.seq_mask] if self.seq_mask is not None else None,
                                 length_dict)
            # both element-wise losses are symmetric in their arguments
            if b.requires_grad or not a.requires_grad:
                loss = WeightedLossFunction.apply(b, a.detach(), w, self.kind,
                                                  self.reduction == "none")
            else:
                loss = WeightedLossFunction.apply(a, b.detach(), w, self.kind,
                                                  self.reduction == "none")
        # (x * 1.0 is x: the launch and its twin in backward are left out on the usual path)
        out = {self.name: loss if weight == 1.0 else loss * weight}
        data.update(out)
        return out
