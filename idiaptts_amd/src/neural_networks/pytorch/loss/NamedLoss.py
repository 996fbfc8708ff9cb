"""NamedLoss with the reference's interface (loss/NamedLoss.py:16-131) for the loss on the hot
path: MSELoss(reduction='none') * seq_mask, reduced 'mean_per_frame' (sum over batch and time /
total frames, mean over features) -- one fused HIP kernel producing the loss and its gradient."""
import torch
from torch import nn

from idiaptts_amd import ops


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
        return (grad * dloss).reshape(ctx.shape), None, None, None


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

    def __init__(self, config):
        super().__init__()
        if config.type != "MSELoss" or config.reduction != 'mean_per_frame' \
                or config.seq_mask is None:
            raise NotImplementedError("Accelerated: MSELoss with seq_mask and 'mean_per_frame' "
                                      "(AcousticModelTrainer.py:179-185).")
        self.name = config.name
        self.input_names = config.input_names
        self.seq_mask = config.seq_mask
        self.batch_first = config.batch_first
        self.loss_weight = config.loss_weight
        self.start_step = config.start_step

    def forward(self, data, length_dict, step):
        a, b = (data[n] for n in self.input_names)       # (target, prediction)
        mask = data[self.seq_mask]                       # [.., .., 1] float, 1 inside the sequence
        row_valid = (mask.reshape(-1) > 0).to(torch.uint8)
        total_num_frames = float(sum(length_dict[self.seq_mask]))
        # MSE is symmetric: differentiate through whichever input needs it
        if b.requires_grad or not a.requires_grad:
            loss = MaskedMSEFunction.apply(b, a.detach(), row_valid, total_num_frames)
        else:
            loss = MaskedMSEFunction.apply(a, b.detach(), row_valid, total_num_frames)
        weight = 0. if step < self.start_step else self.loss_weight
        out = {self.name: loss * weight}
        data.update(out)
        return out
