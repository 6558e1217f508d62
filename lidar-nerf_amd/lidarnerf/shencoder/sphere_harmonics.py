"""SHEncoder — real spherical-harmonics direction encoding (API of lidarnerf/shencoder/sphere_harmonics.py:62-90)."""
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _hip


class _SHEncode(Function):
    @staticmethod
    def forward(ctx, inputs, degree, calc_grad_inputs):
        inputs = inputs.contiguous().float()
        _hip.require_cuda(inputs)
        B, D = inputs.shape
        out = torch.empty((B, degree * degree), dtype=torch.float32, device=inputs.device)
        dy_dx = torch.empty((B, D * degree * degree), dtype=torch.float32, device=inputs.device) \
            if calc_grad_inputs else None
        _hip.call("lnh_sh_encode_forward", inputs.data_ptr(), out.data_ptr(), B, D, degree, _hip.ptr(dy_dx))
        ctx.save_for_backward(inputs, dy_dx)
        ctx.dims = (B, D, degree)
        return out

    @staticmethod
    def backward(ctx, grad):
        inputs, dy_dx = ctx.saved_tensors
        if dy_dx is None:
            return None, None, None
        B, D, degree = ctx.dims
        grad = grad.contiguous().float()
        gi = torch.zeros_like(inputs)
        _hip.call("lnh_sh_encode_backward", grad.data_ptr(), inputs.data_ptr(), B, D, degree, dy_dx.data_ptr(),
                  gi.data_ptr())
        return gi, None, None


sh_encode = _SHEncode.apply


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree = input_dim, degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert 0 < self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        inputs = inputs / size
        lead = list(inputs.shape[:-1])
        flat = inputs.reshape(-1, self.input_dim)
        return sh_encode(flat, self.degree, flat.requires_grad).reshape(lead + [self.output_dim])
