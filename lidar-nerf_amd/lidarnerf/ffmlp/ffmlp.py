"""FFMLP — fully fused tiny MLP on MFMA (API of lidarnerf/ffmlp/ffmlp.py:187-283).

`FFMLP(input_dim, output_dim, hidden_dim, num_layers, activation)` keeps the reference's flat `weights` Parameter
([hidden*in | hidden*hidden*(num_layers-1) | out_pad16*hidden], each row-major [out,in]) and its seed-42
U(+-sqrt(3/hidden)) initialisation.  `fused_mlp(x, mats, act)` is the functional form used by the networks: it takes
the individual bias-free Linear weight matrices (so nn.Linear state-dict keys stay intact) and runs them as ONE kernel.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from .. import _hip

_ACT = {"relu": 0, "exponential": 1, "sine": 2, "sigmoid": 3, "squareplus": 4, "softplus": 5}


def convert_activation(act):
    return _ACT.get(act, 6)


def mlp_dtype():
    """Element type of the fused MLP kernels: bf16 under torch.autocast(dtype=torch.bfloat16) (BASELINE config 5: "fp16
    hash features + bf16 MFMA MLP"), fp16 otherwise (the reference's ffmlp is fp16 only, ffmlp.py:14-60)."""
    if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        return torch.bfloat16
    return torch.half


def _forward_raw(x16, w16, in_dim, hidden, nhm, act, out_act, save_hidden=False):
    B = x16.shape[0]
    y = torch.empty((B, 16), dtype=x16.dtype, device=x16.device)
    fb = torch.empty((nhm + 1, B, hidden), dtype=x16.dtype, device=x16.device) if save_hidden else None
    _hip.call("lnh_mlp_forward" + _hip.mlp_suffix(x16.dtype), x16.data_ptr(), w16.data_ptr(), B, in_dim, 16, hidden, nhm, act, out_act,
              _hip.ptr(fb), y.data_ptr())
    return y, fb


def _backward_raw(gy16, x16, w16, in_dim, hidden, nhm, act, need_dx):
    B = x16.shape[0]
    gx = torch.empty((B, in_dim), dtype=x16.dtype, device=x16.device) if need_dx else None
    gw = torch.zeros(w16.numel(), dtype=torch.float32, device=x16.device)
    _hip.call("lnh_mlp_backward" + _hip.mlp_suffix(x16.dtype), gy16.data_ptr(), x16.data_ptr(), w16.data_ptr(), B, in_dim, 16, hidden, nhm, act, 6,
              _hip.ptr(gx), gw.data_ptr())
    return gx, gw


class _FusedMLP(Function):
    """x [B,in_pad] (any float dtype), flat weights (any float dtype) -> y [B,16] fp16 (bf16 under bf16 autocast)."""

    @staticmethod
    def forward(ctx, x, w, in_dim, hidden, nhm, act, out_act, inference):
        dt = mlp_dtype()
        x16 = x.contiguous().to(dt)
        w16 = w.contiguous().to(dt)
        _hip.require_cuda(x16, w16)
        y, _ = _forward_raw(x16, w16, in_dim, hidden, nhm, act, out_act)
        if not inference:
            ctx.save_for_backward(x16, w16)
            ctx.meta = (in_dim, hidden, nhm, act, out_act, x.dtype, w.dtype, x.requires_grad)
        return y

    @staticmethod
    def backward(ctx, gy):
        x16, w16 = ctx.saved_tensors
        in_dim, hidden, nhm, act, out_act, xdt, wdt, need_dx = ctx.meta
        if out_act != 6:
            raise RuntimeError("fused MLP: backward through an output activation is not supported (ffmlp.py:196)")
        gx, gw = _backward_raw(gy.contiguous().to(x16.dtype), x16, w16, in_dim, hidden, nhm, act, need_dx)
        return (gx.to(xdt) if gx is not None else None), gw.to(wdt), None, None, None, None, None, None


def _act_torch(a, x):
    """lidarnerf/ffmlp/src/utils.h:479-531 on tensors (K = 10 for squareplus / softplus)."""
    if a == 0:
        return torch.relu(x)
    if a == 1:
        return torch.exp(x)
    if a == 2:
        return torch.sin(x)
    if a == 3:
        return torch.sigmoid(x)
    if a == 4:
        y = x * 10.0
        return 0.5 * (y + torch.sqrt(y * y + 4.0)) / 10.0
    if a == 5:
        return torch.log(torch.exp(x * 10.0) + 1.0) / 10.0
    return x


def kernel_supported(in_dim, hidden, nhm):
    """Shapes the register-resident MFMA kernels (csrc/mlp.hip) are instantiated for."""
    return hidden in (32, 64) and nhm <= 2 and in_dim <= 128 and in_dim % 16 == 0


def gemm_mlp(x, w, in_dim, hidden, nhm, act, out_act):
    """The reference's remaining hidden widths (128, 256: ffmlp.py:202-209) and deeper nets — FFMLP(gemm_chain=True) only — as a chain of plain library GEMMs
    (rocBLAS / hipBLASLt through torch.matmul, fp32 accumulation) with the same storage model as the fused kernels: every
    layer's activations are stored in the 16-bit element type.  At 128 / 256 the layers ARE library-sized GEMMs (weights
    no longer fit a wave's registers, the point of the fused kernel); autograd differentiates the chain."""
    dt = mlp_dtype()
    with torch.autocast("cuda", enabled=False):
        h = x.to(dt)
        w = w.to(dt)
        o = hidden * in_dim
        h = _act_torch(act, h @ w[:o].view(hidden, in_dim).t())
        for _ in range(nhm):
            h = _act_torch(act, h @ w[o:o + hidden * hidden].view(hidden, hidden).t())
            o += hidden * hidden
        return _act_torch(out_act, h @ w[o:o + 16 * hidden].view(16, hidden).t())


def fused_mlp(x, mats, activation=0, inference=False):
    """Bias-free Linear stack as one kernel.  mats: list of weight tensors [out_k, in_k] (>= 2); hidden width 32 or 64;
    last out <= 16; first in <= 128.  Returns [B, out_last] fp16."""
    hidden = mats[0].shape[0]
    in_dim = mats[0].shape[1]
    out_dim = mats[-1].shape[0]
    in_pad = (in_dim + 15) // 16 * 16
    w0 = mats[0] if in_pad == in_dim else torch.nn.functional.pad(mats[0], (0, in_pad - in_dim))
    wl = mats[-1] if out_dim == 16 else torch.nn.functional.pad(mats[-1], (0, 0, 0, 16 - out_dim))
    flat = torch.cat([w0.reshape(-1)] + [m.reshape(-1) for m in mats[1:-1]] + [wl.reshape(-1)])
    if x.shape[1] != in_pad:
        x = torch.nn.functional.pad(x, (0, in_pad - x.shape[1]))
    y = _FusedMLP.apply(x, flat, in_pad, hidden, len(mats) - 2, activation, 6, inference)
    return y[:, :out_dim] if out_dim != 16 else y


class FFMLP(nn.Module):
    """Shapes with a fused MFMA kernel in this build: hidden 16 (zero-padded onto the hidden-32 kernels) / 32 / 64, at most
    3 hidden layers (2 hidden->hidden matrices), input <= 128.  The reference's other widths (128, 256; ffmlp.cu:756-800)
    and deeper nets have NONE: the constructor refuses them exactly as the C ABI does (lnh_mlp_forward ->
    LNH_ERR_UNSUPPORTED) instead of quietly being a different kind of implementation behind the same class.  A caller who
    wants them anyway says so — `gemm_chain=True` runs the layers as library GEMMs (rocBLAS / hipBLASLt through
    torch.matmul, the same 16-bit storage model, autograd for the backward), which at 128 / 256 is what the layers are."""

    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation="relu", gemm_chain=False):
        super().__init__()
        self.input_dim, self.output_dim, self.hidden_dim, self.num_layers = input_dim, output_dim, hidden_dim, num_layers
        self.activation = convert_activation(activation)
        self.output_activation = convert_activation("none")
        self.tensorcore_width = 16
        assert hidden_dim in [16, 32, 64, 128, 256], \
            f"FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got {hidden_dim}"
        assert input_dim > 0 and input_dim % 16 == 0, f"FFMLP input_dim should be 16 * m (m  > 0), but got {input_dim}"
        assert output_dim <= 16, f"FFMLP current only supports output dim <= 16, but got {output_dim}"
        assert num_layers >= 2, f"FFMLP num_layers should be larger than 2 (3 matmuls), but got {num_layers}"
        self.gemm_chain = bool(gemm_chain)
        fused = kernel_supported(input_dim, 32 if hidden_dim == 16 else hidden_dim, num_layers - 1)
        if not fused and not self.gemm_chain:
            raise RuntimeError(
                f"FFMLP(input_dim={input_dim}, hidden_dim={hidden_dim}, num_layers={num_layers}): no fused MFMA kernel for "
                "this shape in this build (kernels: hidden_dim 16 / 32 / 64, num_layers <= 3, input_dim <= 128 — the C ABI "
                "refuses it the same way: lnh_mlp_forward returns LNH_ERR_UNSUPPORTED).  Pass gemm_chain=True to run it as a "
                "chain of library GEMMs instead.")
        self.padded_output_dim = int(math.ceil(output_dim / 16)) * 16
        self.num_parameters = hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + self.padded_output_dim)
        self.weights = nn.Parameter(torch.zeros(self.num_parameters))
        self.reset_parameters()

    def cleanup(self):  # the reference frees its split-K streams here; nothing to free
        pass

    def __repr__(self):
        return (f"FFMLP: input_dim={self.input_dim} output_dim={self.output_dim} hidden_dim={self.hidden_dim} "
                f"num_layers={self.num_layers} activation={self.activation}")

    def reset_parameters(self):
        torch.manual_seed(42)
        std = math.sqrt(3 / self.hidden_dim)
        self.weights.data.uniform_(-std, std)

    def _weights_padded_to_32(self):
        """hidden_dim = 16 on the hidden-32 MFMA kernels: every matrix zero-padded to 32 hidden units (a padded unit has
        zero weights in AND out: it contributes nothing forward, receives a zero gradient, and autograd drops the padding's
        gradient on the way back to `weights`).  Same flat layout, [32*in | 32*32*(layers-1) | 16*32]."""
        h, i, n = self.hidden_dim, self.input_dim, self.num_layers - 1
        w = self.weights
        parts = [F.pad(w[:h * i].view(h, i), (0, 0, 0, 32 - h)).reshape(-1)]
        o = h * i
        for _ in range(n):
            parts.append(F.pad(w[o:o + h * h].view(h, h), (0, 32 - h, 0, 32 - h)).reshape(-1))
            o += h * h
        parts.append(F.pad(w[o:o + self.padded_output_dim * h].view(self.padded_output_dim, h), (0, 32 - h)).reshape(-1))
        return torch.cat(parts)

    def forward(self, inputs):
        if self.hidden_dim == 16 and kernel_supported(self.input_dim, 32, self.num_layers - 1):
            y = _FusedMLP.apply(inputs, self._weights_padded_to_32(), self.input_dim, 32, self.num_layers - 1,
                                self.activation, self.output_activation, not self.training)
        elif kernel_supported(self.input_dim, self.hidden_dim, self.num_layers - 1):
            y = _FusedMLP.apply(inputs, self.weights, self.input_dim, self.hidden_dim, self.num_layers - 1,
                                self.activation, self.output_activation, not self.training)
        else:  # hidden 128 / 256, deeper or wider-input nets, asked for with gemm_chain=True: library GEMM chain, same semantics
            if not inputs.is_cuda:
                raise RuntimeError("lidarnerf_hip: tensor must live on the GPU (no CPU path in this library)")
            y = gemm_mlp(inputs, self.weights, self.input_dim, self.hidden_dim, self.num_layers - 1, self.activation,
                         self.output_activation)
        return y[:, :self.output_dim] if self.output_dim != self.padded_output_dim else y
