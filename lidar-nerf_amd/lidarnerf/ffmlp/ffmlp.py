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
              _hip.ptr(gx), gw.data_ptr(), *_hip.wgrad_ws(x16.device))
    return gx, gw


def one_kernel_backward(hidden, nhm):
    """Shapes whose backward is ONE kernel (weights and all weight-gradient tiles in a wave's registers)."""
    return hidden in (32, 64) and nhm <= 2


def _split(w, in_dim, hidden, nhm):
    """The matrices of a flat weight vector: W0 [hidden, in], Wh_m [hidden, hidden], Wo [16, hidden]."""
    o = hidden * in_dim
    mats = [w[:o].view(hidden, in_dim)]
    for _ in range(nhm):
        mats.append(w[o:o + hidden * hidden].view(hidden, hidden))
        o += hidden * hidden
    mats.append(w[o:o + 16 * hidden].view(16, hidden))
    return mats


def _wgrad(g16, a16, out):
    """out [M, N] fp32 (a slice of the flat gradient vector) += g16^T a16, the contraction over the batch: lnh_mlp_wgrad
    (csrc/mlp_wgrad.hip; rounds 4-5: torch.bmm over 4096-row slices).  The first matrix of an MLP whose input width is not a
    multiple of 16 cannot occur here: FFMLP pads its input to 16 (ffmlp.py:221-224 of the reference)."""
    B, M = g16.shape
    N = a16.shape[1]
    _hip.call("lnh_mlp_wgrad" + _hip.mlp_suffix(g16.dtype), g16.data_ptr(), a16.data_ptr(), B, M, N, out.data_ptr(),
              *_hip.wgrad_ws(g16.device))


def _backward_wide(gy16, x16, w16, fb, in_dim, hidden, nhm, act, need_dx):
    """Backward of the shapes without a one-kernel backward (hidden 128 / 256, more than two hidden matrices), structured
    as the reference's own (ffmlp.cu:578-733 + 1107-1263): lnh_mlp_backward_data — one fused kernel for the activation and
    input gradients — then the weight gradients dW_l = G_l^T A_(l-1) from the two buffers, one lnh_mlp_wgrad launch per
    matrix (a fixed-order sum over the batch)."""
    B = x16.shape[0]
    wt = torch.cat([m.t().contiguous().reshape(-1) for m in _split(w16, in_dim, hidden, nhm)])
    gb = torch.empty((nhm + 1, B, hidden), dtype=x16.dtype, device=x16.device)
    gx = torch.empty((B, in_dim), dtype=x16.dtype, device=x16.device) if need_dx else None
    _hip.call("lnh_mlp_backward_data" + _hip.mlp_suffix(x16.dtype), gy16.data_ptr(), fb.data_ptr(), wt.data_ptr(), B, in_dim,
              16, hidden, nhm, act, gb.data_ptr(), _hip.ptr(gx))
    gw = torch.zeros(w16.numel(), dtype=torch.float32, device=x16.device)
    o = hidden * in_dim
    _wgrad(gb[0], x16, gw[:o])
    for m in range(nhm):
        _wgrad(gb[m + 1], fb[m], gw[o:o + hidden * hidden])
        o += hidden * hidden
    _wgrad(gy16, fb[nhm], gw[o:o + 16 * hidden])
    return gx, gw


class _FusedMLP(Function):
    """x [B,in_pad] (any float dtype), flat weights (any float dtype) -> y [B,16] fp16 (bf16 under bf16 autocast)."""

    @staticmethod
    def forward(ctx, x, w, in_dim, hidden, nhm, act, out_act, inference):
        dt = mlp_dtype()
        x16 = x.contiguous().to(dt)
        w16 = w.contiguous().to(dt)
        _hip.require_cuda(x16, w16)
        # (the one-kernel backward recomputes the hidden activations; the wide shapes' backward reads them back)
        keep = not inference and not one_kernel_backward(hidden, nhm)
        y, fb = _forward_raw(x16, w16, in_dim, hidden, nhm, act, out_act, save_hidden=keep)
        if not inference:
            ctx.save_for_backward(x16, w16, *([fb] if keep else []))
            ctx.meta = (in_dim, hidden, nhm, act, out_act, x.dtype, w.dtype, x.requires_grad)
        return y

    @staticmethod
    def backward(ctx, gy):
        x16, w16 = ctx.saved_tensors[:2]
        in_dim, hidden, nhm, act, out_act, xdt, wdt, need_dx = ctx.meta
        if out_act != 6:
            raise RuntimeError("fused MLP: backward through an output activation is not supported (ffmlp.py:196)")
        gy16 = gy.contiguous().to(x16.dtype)
        if one_kernel_backward(hidden, nhm):
            gx, gw = _backward_raw(gy16, x16, w16, in_dim, hidden, nhm, act, need_dx)
        else:
            gx, gw = _backward_wide(gy16, x16, w16, ctx.saved_tensors[2], in_dim, hidden, nhm, act, need_dx)
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
    """Shapes with fused MFMA kernels: hidden 32 / 64 with <= 2 hidden matrices in the register-resident kernels
    (csrc/mlp.hip), hidden 128 / 256 and up to 14 hidden matrices at any width in the kernels that load a weight fragment
    where it is used (csrc/mlp_wide.hip)."""
    return hidden in (32, 64, 128, 256) and nhm <= 14 and in_dim <= 128 and in_dim % 16 == 0


def gemm_mlp(x, w, in_dim, hidden, nhm, act, out_act):
    """Shapes without a fused kernel (input_dim > 128, more than 15 hidden layers) — FFMLP(gemm_chain=True) only — as a chain of plain library GEMMs
    (rocBLAS / hipBLASLt through torch.matmul, fp32 accumulation) with the same storage model as the fused kernels: every
    layer's activations are stored in the 16-bit element type; autograd differentiates the chain."""
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
    """Every width of the reference (16 .. 256, ffmlp.py:202-209) runs on fused MFMA kernels: hidden 16 zero-padded onto the
    hidden-32 kernels, 32 / 64 with at most 3 hidden layers with weights AND weight gradients in registers (forward and
    backward one kernel each), 128 / 256 and deeper nets with weight fragments loaded where they are used and the backward
    split as the reference splits its own — a fused kernel for the activation / input gradients, library GEMMs for the
    weight gradients (ffmlp.cu:578-733, 1107-1263).  What has no kernel — input_dim > 128, more than 15 hidden layers — the
    reference's constructor accepts (any input_dim % 16 == 0, ffmlp.py:202-216), so it is accepted here too and runs every
    layer as a library GEMM (torch.matmul, the same 16-bit storage model, autograd) with a one-time warning;
    `gemm_chain=True` asks for that chain explicitly (no warning), `strict_fused=True` refuses such shapes the way the C
    ABI does (LNH_ERR_UNSUPPORTED)."""

    _warned_gemm_chain = False

    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation="relu", gemm_chain=False,
                 strict_fused=False):
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
            msg = (f"FFMLP(input_dim={input_dim}, hidden_dim={hidden_dim}, num_layers={num_layers}): no fused MFMA kernel for "
                   "this shape in this build (kernels: hidden_dim 16 .. 256, num_layers <= 15, input_dim <= 128; the C ABI "
                   "returns LNH_ERR_UNSUPPORTED for it)")
            if strict_fused:
                raise RuntimeError(msg + ".  Drop strict_fused to run it as a chain of library GEMMs instead.")
            if not FFMLP._warned_gemm_chain:
                import warnings
                warnings.warn(msg + ": running it as a chain of library GEMMs (torch.matmul), same storage model.")
                FFMLP._warned_gemm_chain = True
            self.gemm_chain = True
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
        else:  # wider-input or very deep nets, asked for with gemm_chain=True: library GEMM chain, same semantics
            if not inputs.is_cuda:
                raise RuntimeError("lidarnerf_hip: tensor must live on the GPU (no CPU path in this library)")
            y = gemm_mlp(inputs, self.weights, self.input_dim, self.hidden_dim, self.num_layers - 1, self.activation,
                         self.output_activation)
        return y[:, :self.output_dim] if self.output_dim != self.padded_output_dim else y
