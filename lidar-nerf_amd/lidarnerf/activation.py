"""trunc_exp — exp in fp32 with a clamped backward (mirrors lidarnerf/activation.py:6-20 of the reference)."""
import torch
from torch.autograd import Function


class _TruncExp(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.float()  # custom_fwd(cast_inputs=float32) of the reference
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply
