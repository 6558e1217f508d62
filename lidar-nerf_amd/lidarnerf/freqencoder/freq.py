"""FreqEncoder — [x | sin(2^f x) | cos(2^f x)]_f positional encoding (API of lidarnerf/freqencoder/freq.py:55-77)."""
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _hip


class _FreqEncode(Function):
    @staticmethod
    def forward(ctx, inputs, degree, output_dim):
        inputs = inputs.contiguous().float()
        _hip.require_cuda(inputs)
        B, D = inputs.shape
        out = torch.empty((B, output_dim), dtype=torch.float32, device=inputs.device)
        _hip.call("lnh_freq_encode_forward", inputs.data_ptr(), B, D, degree, output_dim, out.data_ptr())
        ctx.save_for_backward(out)
        ctx.dims = (B, D, degree, output_dim)
        return out

    @staticmethod
    def backward(ctx, grad):
        (out,) = ctx.saved_tensors
        B, D, degree, C = ctx.dims
        grad = grad.contiguous().float()
        gi = torch.empty((B, D), dtype=torch.float32, device=grad.device)
        _hip.call("lnh_freq_encode_backward", grad.data_ptr(), out.data_ptr(), B, D, degree, C, gi.data_ptr())
        return gi, None, None


freq_encode = _FreqEncode.apply


class FreqEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree = input_dim, degree
        self.output_dim = input_dim + input_dim * 2 * degree

    def __repr__(self):
        return f"FreqEncoder: input_dim={self.input_dim} degree={self.degree} output_dim={self.output_dim}"

    def forward(self, inputs, **kwargs):
        lead = list(inputs.shape[:-1])
        out = freq_encode(inputs.reshape(-1, self.input_dim), self.degree, self.output_dim)
        return out.reshape(lead + [self.output_dim])
