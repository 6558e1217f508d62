"""tiny-cuda-nn shaped modules on top of the HIP kernels, so that the reference's `network_tcnn.NeRFNetwork`
(lidarnerf/nerf/network_tcnn.py:46-132) — the backend `main_lidarnerf.py -L` really selects — runs without
tinycudann.  Only the two entry points that file uses exist: `Encoding(n_input_dims, encoding_config)` and
`Network(n_input_dims, n_output_dims, network_config)`, with tcnn's conventions: inputs in [0,1], a flat fp32 `params`
Parameter per module (state-dict key `<name>.params`), `n_output_dims`, fp16 outputs under autocast.

NOT CHECKPOINT-COMPATIBLE with real tiny-cuda-nn: `_HashGrid` keeps torch-ngp's level geometry ((res+1)^D rows and
stride res+1 on the dense levels, row counts rounded up to 8) whereas tiny-cuda-nn uses res^D rows / stride res there, so
the flat `params` vector of a tcnn-trained HashGrid has a different length and index function (strict load fails with a
size mismatch).  The modules train, render and resume their OWN checkpoints; see INTEGRATION.md §A.

PARITY UNPINNED: tiny-cuda-nn is an external, unversioned dependency of the reference (readme.md:74-76) whose source is
not in the tree.  HashGrid / SphericalHarmonics reuse the in-tree encoders' arithmetic (torch-ngp's gridencoder and
shencoder are restatements of tcnn's); Frequency follows tcnn's published layout (72 = 3 * 12 * 2 outputs,
out[dim*2K + 2k + p] = sin(2^k * pi * x[dim] + p * pi/2)); FullyFusedMLP uses tcnn's weight order (first matrix
[n_neurons, pad16(n_in)], hidden matrices, last matrix [pad16(n_out), n_neurons], all row-major, bias-free).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from .ffmlp import fused_mlp
from .gridencoder import GridEncoder
from .shencoder import SHEncoder


def _pad16(n):
    return (n + 15) // 16 * 16


class _HashGrid(GridEncoder):
    """GridEncoder whose table is exposed as tcnn's flat `params` vector."""

    def __init__(self, cfg, n_input_dims):
        super().__init__(input_dim=n_input_dims, num_levels=int(cfg.get("n_levels", 16)),
                         level_dim=int(cfg.get("n_features_per_level", 2)),
                         per_level_scale=float(cfg.get("per_level_scale", 2.0)),
                         base_resolution=int(cfg.get("base_resolution", 16)),
                         log2_hashmap_size=int(cfg.get("log2_hashmap_size", 19)), desired_resolution=None,
                         gridtype="hash", align_corners=False,
                         interpolation="smoothstep" if str(cfg.get("interpolation", "Linear")).lower() == "smoothstep"
                         else "linear")
        flat = self.embeddings.data.reshape(-1).clone()
        del self.embeddings
        self.params = nn.Parameter(flat)
        offs = self.offsets  # tcnn checkpoints hold nothing but `params`: keep the level offsets out of the state dict
        del self.offsets
        self.register_buffer("offsets", offs, persistent=False)

    @property
    def embeddings(self):  # [rows, level_dim] view; gradients flow into `params`
        return self.params.view(-1, self.level_dim)

    def reset_parameters(self):
        if "params" in self._parameters:
            self.params.data.uniform_(-1e-4, 1e-4)
        else:
            super().reset_parameters()


def _tcnn_hashgrid_param_count(cfg, n_input_dims):
    """Length of the flat `params` vector REAL tiny-cuda-nn allocates for this HashGrid config (its published layout,
    GridEncoding's constructor: per level resolution = ceil(exp2(l * log2(per_level_scale)) * base - 1) + 1, rows =
    resolution^D rounded up to 8, capped at 2^log2_hashmap_size) — used only to recognise such a checkpoint."""
    L, F = int(cfg.get("n_levels", 16)), int(cfg.get("n_features_per_level", 2))
    pls, base = float(cfg.get("per_level_scale", 2.0)), int(cfg.get("base_resolution", 16))
    cap = 1 << int(cfg.get("log2_hashmap_size", 19))
    total = 0
    for l in range(L):
        scale = np.float32(np.exp2(np.float32(l) * np.float32(np.log2(pls)))) * np.float32(base) - np.float32(1.0)
        res = int(np.ceil(scale)) + 1
        total += min(-(-res ** n_input_dims // 8) * 8, cap)
    return total * F


def _tcnn_levels(cfg, n_input_dims):
    """Per level of a tiny-cuda-nn HashGrid (its published GridEncoding constructor / grid_index): (resolution, rows, hashed)."""
    L = int(cfg.get("n_levels", 16))
    pls, base = float(cfg.get("per_level_scale", 2.0)), int(cfg.get("base_resolution", 16))
    cap = 1 << int(cfg.get("log2_hashmap_size", 19))
    out = []
    for l in range(L):
        scale = np.float32(np.exp2(np.float32(l) * np.float32(np.log2(pls)))) * np.float32(base) - np.float32(1.0)
        res = int(np.ceil(scale)) + 1
        dense_rows = -(-res ** n_input_dims // 8) * 8
        out.append((res, min(dense_rows, cap), res ** n_input_dims > min(dense_rows, cap)))
    return out


def convert_tcnn_hashgrid_params(params, cfg, n_input_dims=3):
    """The flat `params` vector of a HashGrid trained by REAL tiny-cuda-nn -> this package's layout (a new fp32 vector).

    PARITY UNPINNED (tiny-cuda-nn is not in the tree; restated from its published source): tiny-cuda-nn indexes vertex
    (x, y, z) of a DENSE level at x + y res + z res^2 (mod the level's rows: the vertices x = res alias the next row), this
    package at x + y (res + 1) + z (res + 1)^2 — every vertex 0 .. res gets the value tiny-cuda-nn would have read for it, so the
    encoder's OUTPUT is the same at load time (the aliased vertices are separate parameters afterwards).  Hashed levels use
    the same hash and the same row count: copied.  A level that tiny-cuda-nn keeps dense and this package hashes (res^D <=
    2^log2_hashmap_size < (res + 1)^D) cannot be converted and raises.  Opt-in: `Encoding.load_state_dict` refuses such a
    checkpoint by name unless LNH_TCNN_CONVERT=1."""
    from .gridencoder.grid import level_offsets
    F_ = int(cfg.get("n_features_per_level", 2))
    D = int(n_input_dims)
    theirs = _tcnn_levels(cfg, D)
    mine = level_offsets(D, int(cfg.get("n_levels", 16)), float(cfg.get("per_level_scale", 2.0)),
                         int(cfg.get("base_resolution", 16)), int(cfg.get("log2_hashmap_size", 19)), False)
    mine = [int(v) for v in mine]
    src = torch.as_tensor(params).detach().float().cpu().reshape(-1, F_)
    if src.shape[0] != sum(r for _, r, _ in theirs):
        raise RuntimeError(f"convert_tcnn_hashgrid_params: {src.shape[0]} rows, tiny-cuda-nn holds {sum(r for _, r, _ in theirs)} for this config")
    out = torch.zeros((mine[-1], F_), dtype=torch.float32)
    o_t = 0
    for l, (res, rows_t, hashed_t) in enumerate(theirs):
        rows_m = mine[l + 1] - mine[l]
        hashed_m = (res + 1) ** D > rows_m
        lvl = src[o_t:o_t + rows_t]
        if hashed_t and hashed_m:
            if rows_t != rows_m:
                raise RuntimeError(f"convert_tcnn_hashgrid_params: level {l}: hashed with {rows_t} rows there, {rows_m} here")
            out[mine[l]:mine[l + 1]] = lvl
        elif not hashed_t and not hashed_m:
            g = torch.arange(res + 1)
            idx_t = torch.zeros((res + 1,) * D, dtype=torch.long)
            idx_m = torch.zeros((res + 1,) * D, dtype=torch.long)
            for d in range(D):  # dimension 0 is the fastest in both index functions
                shape = [1] * D
                shape[D - 1 - d] = res + 1
                idx_t += g.view(shape) * res ** d
                idx_m += g.view(shape) * (res + 1) ** d
            out[mine[l] + idx_m.reshape(-1)] = lvl[idx_t.reshape(-1) % rows_t]
        else:
            raise RuntimeError(f"convert_tcnn_hashgrid_params: level {l} (resolution {res}) is dense in tiny-cuda-nn and hashed "
                               "here: no exact conversion")
        o_t += rows_t
    return out.reshape(-1)


class Encoding(nn.Module):
    """tcnn.Encoding(n_input_dims, encoding_config): otype HashGrid | Frequency | SphericalHarmonics | Identity."""

    def __init__(self, n_input_dims, encoding_config, dtype=None, seed=1337):
        super().__init__()
        self.n_input_dims = int(n_input_dims)
        self.encoding_config = dict(encoding_config)
        otype = str(encoding_config.get("otype", "")).lower()
        self.otype = otype
        if otype in ("hashgrid", "grid"):
            self.impl = _HashGrid(encoding_config, self.n_input_dims)
            self.n_output_dims = self.impl.output_dim
        elif otype == "frequency":
            self.degree = int(encoding_config.get("n_frequencies", encoding_config.get("degree", 12)))
            self.impl = None
            self.n_output_dims = self.n_input_dims * self.degree * 2
        elif otype == "sphericalharmonics":
            if self.n_input_dims != 3:
                raise RuntimeError("SphericalHarmonics encoding needs 3 input dims")
            self.impl = SHEncoder(input_dim=3, degree=int(encoding_config.get("degree", 4)))
            self.n_output_dims = self.impl.output_dim
        elif otype == "identity":
            self.impl = None
            self.n_output_dims = self.n_input_dims
        else:
            raise RuntimeError(f"tcnn_compat.Encoding: unsupported otype {encoding_config.get('otype')!r}")

    # tcnn modules own ONE flat parameter called `params` (empty for parameter-free encodings)
    @property
    def params(self):
        if self.otype in ("hashgrid", "grid"):
            return self.impl.params
        return torch.zeros(0)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = super().state_dict(*args, destination=destination, prefix=prefix, keep_vars=keep_vars)
        key = prefix + "impl.params"
        if key in sd:  # tcnn checkpoints call it `<module>.params`
            sd[prefix + "params"] = sd.pop(key)
        elif prefix + "params" not in sd:  # parameter-free encodings still own an (empty) `params` in tcnn
            sd[prefix + "params"] = torch.zeros(0)
        return sd

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        if prefix + "params" in state_dict and self.otype in ("hashgrid", "grid"):
            got, mine = state_dict[prefix + "params"].numel(), self.impl.params.numel()
            if got != mine:
                theirs = _tcnn_hashgrid_param_count(self.encoding_config, self.n_input_dims)
                if got == theirs and os.environ.get("LNH_TCNN_CONVERT") == "1":
                    import warnings
                    warnings.warn("tcnn_compat.Encoding: converting a tiny-cuda-nn HashGrid checkpoint to this package's level "
                                  "layout (convert_tcnn_hashgrid_params: same encoder output at load time; parity unpinned)")
                    state_dict[prefix + "params"] = convert_tcnn_hashgrid_params(
                        state_dict[prefix + "params"], self.encoding_config, self.n_input_dims).to(self.impl.params.device)
                elif got == theirs:
                    raise RuntimeError(
                        f"tcnn_compat.Encoding: `{prefix}params` holds {got} values, this HashGrid has {mine} — {got} is what REAL "
                        "tiny-cuda-nn allocates for this config: the checkpoint was written by the reference running on tinycudann.  "
                        "Such checkpoints do not load into this package: tiny-cuda-nn's dense levels have res^D rows with stride "
                        "res, this package keeps torch-ngp's (res+1)^D rows with stride res+1 (INTEGRATION.md §A), so the flat "
                        "vector has another length AND another index function.  Re-train with this package, resume one of its own "
                        "checkpoints, or set LNH_TCNN_CONVERT=1 to convert the table at load (tcnn_compat."
                        "convert_tcnn_hashgrid_params: same encoder output at load time, restated from tiny-cuda-nn's published "
                        "index function — unpinned).")
                else:
                    raise RuntimeError(
                    f"tcnn_compat.Encoding: `{prefix}params` holds {got} values, this HashGrid has {mine} (real tiny-cuda-nn "
                    f"would hold {theirs} for this config): the checkpoint belongs to another encoding config.")
            state_dict[prefix + "impl.params"] = state_dict.pop(prefix + "params")
        elif prefix + "params" in state_dict:
            state_dict.pop(prefix + "params")  # parameter-free encodings store an empty tensor
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def frequency(self, x):
        """[M, D] in [0,1] -> [M, D*2K] fp32: out[d*2K + 2k + p] = sin(2^k * pi * x_d + p*pi/2)."""
        k = torch.arange(self.degree, device=x.device, dtype=torch.float32)
        arg = x.float().unsqueeze(-1) * (torch.exp2(k) * math.pi)  # [M, D, K]
        phase = torch.tensor([0.0, 0.5 * math.pi], device=x.device)
        return torch.sin(arg.unsqueeze(-1) + phase).reshape(x.shape[0], -1)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("tcnn_compat.Encoding: input must be a CUDA (ROCm) tensor")
        if self.otype in ("hashgrid", "grid"):
            return self.impl(x * 2.0 - 1.0, bound=1)  # GridEncoder maps [-b, b] -> [0, 1] itself
        if self.otype == "frequency":
            return self.frequency(x.reshape(-1, self.n_input_dims)).view(*x.shape[:-1], self.n_output_dims)
        if self.otype == "sphericalharmonics":
            return self.impl(x * 2.0 - 1.0)  # tcnn's SH takes [0,1] and maps it back to [-1,1]
        return x


_ACT = {"relu": 0, "exponential": 1, "sine": 2, "sigmoid": 3, "squareplus": 4, "softplus": 5, "none": 6}


class Network(nn.Module):
    """tcnn.Network(n_input_dims, n_output_dims, network_config) with otype FullyFusedMLP / CutlassMLP."""

    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        super().__init__()
        self.n_input_dims, self.n_output_dims = int(n_input_dims), int(n_output_dims)
        self.network_config = dict(network_config)
        self.n_neurons = int(network_config.get("n_neurons", 64))
        self.n_hidden_layers = int(network_config.get("n_hidden_layers", 1))
        act = str(network_config.get("activation", "ReLU")).lower()
        out_act = str(network_config.get("output_activation", "None")).lower()
        if act not in _ACT or out_act != "none":
            raise RuntimeError("tcnn_compat.Network: activation must be one of %s and output_activation None"
                               % sorted(_ACT))
        if self.n_hidden_layers < 1:
            raise RuntimeError("tcnn_compat.Network: n_hidden_layers >= 1")
        self.activation = _ACT[act]
        self.in_pad, self.out_pad = _pad16(self.n_input_dims), _pad16(self.n_output_dims)
        n = self.n_neurons
        self._shapes = [(n, self.in_pad)] + [(n, n)] * (self.n_hidden_layers - 1) + [(self.out_pad, n)]
        total = sum(a * b for a, b in self._shapes)
        g = torch.Generator().manual_seed(seed)
        flat = torch.empty(total)
        off = 0
        for (o, i) in self._shapes:  # xavier-uniform per matrix, as tcnn initialises
            lim = math.sqrt(6.0 / (o + i))
            flat[off:off + o * i] = (torch.rand(o * i, generator=g) * 2 - 1) * lim
            off += o * i
        self.params = nn.Parameter(flat)

    def matrices(self):
        """Views [out, in] of the flat parameter vector (padded shapes)."""
        mats, off = [], 0
        for (o, i) in self._shapes:
            mats.append(self.params[off:off + o * i].view(o, i))
            off += o * i
        return mats

    def forward(self, x):
        if not x.is_cuda:  # like tinycudann itself: GPU tensors only, no host fallback
            raise RuntimeError("tcnn_compat.Network: input must be a CUDA (ROCm) tensor")
        lead = x.shape[:-1]
        x = x.reshape(-1, self.n_input_dims)
        mats = self.matrices()
        if self.n_neurons in (32, 64) and self.out_pad == 16 and self.in_pad <= 128 and len(mats) <= 4:
            xp = torch.nn.functional.pad(x, (0, self.in_pad - self.n_input_dims)) if self.in_pad != self.n_input_dims else x
            y = fused_mlp(xp, mats, activation=self.activation, inference=not torch.is_grad_enabled())
        else:
            if self.activation != 0:
                raise RuntimeError("tcnn_compat.Network: only ReLU outside the fused 64-wide configuration")
            h = torch.nn.functional.pad(x, (0, self.in_pad - self.n_input_dims)).to(self.params.dtype)
            for k, m in enumerate(mats):
                h = h @ m.t()
                if k != len(mats) - 1:
                    h = torch.relu(h)
            y = h
        return y[:, :self.n_output_dims].view(*lead, self.n_output_dims)


def per_level_scale(desired_resolution, bound, base_resolution=16, n_levels=16):
    """network_tcnn.py:39-41"""
    return float(np.exp2(np.log2(desired_resolution * bound / base_resolution) / (n_levels - 1)))
