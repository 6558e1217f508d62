"""NumPy restatement of the fully-fused tiny MLP (forward / backward) in float64.

TEST INFRASTRUCTURE ONLY.  Follows /root/reference:
  lidarnerf/ffmlp/ffmlp.py:187-283            FFMLP module: flat weight vector, padding, constraints
  lidarnerf/ffmlp/src/ffmlp.cu:460-576        kernel_mlp_fused           (layer chain, per-layer activation)
  lidarnerf/ffmlp/src/ffmlp.cu:578-733        kernel_mlp_fused_backward  (dgrad with activation transfer)
  lidarnerf/ffmlp/src/ffmlp.cu:1107-1263      weight gradients dW = dY^T X summed over the batch
  lidarnerf/ffmlp/src/utils.h:479-664         activations (ReLU/Exponential/Sine/Sigmoid/Squareplus/Softplus/None)
  lidarnerf/nerf/network.py:45-59,83-99       bias-free Linear stacks (same algebra, n matrices >= 2)

Numerics model (`half=True`; `half="bf16"` models the bf16-operand build the same way with bfloat16 roundings):
inputs, weights and inter-layer activations are fp16 values (as in the reference, which stores them
as __half); every dot product is evaluated exactly (float64) and rounded ONCE to fp16 when it is stored as an
activation.  The reference accumulates in fp16 WMMA fragments, the HIP kernel in fp32 MFMA accumulators; both are
approximations of this value (tolerance stated in tests/test_mlp_gpu.py and DESIGN.md).
"""
import math

import numpy as np

ACT_RELU, ACT_EXP, ACT_SINE, ACT_SIGMOID, ACT_SQUAREPLUS, ACT_SOFTPLUS, ACT_NONE = range(7)
K_ACT = 10.0  # utils.h: squareplus / softplus sharpness


def ffmlp_num_params(input_dim, output_dim, hidden_dim, num_layers):
    padded_out = int(math.ceil(output_dim / 16)) * 16
    return hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + padded_out)


def ffmlp_split_weights(w, input_dim, output_dim, hidden_dim, num_layers):
    """ffmlp.py:222-226 / ffmlp.cu:861-864: [hidden*in | hidden*hidden*(layers-1) | out_pad16*hidden], each [out,in]."""
    padded_out = int(math.ceil(output_dim / 16)) * 16
    mats, o = [], 0
    mats.append(w[o:o + hidden_dim * input_dim].reshape(hidden_dim, input_dim)); o += hidden_dim * input_dim
    for _ in range(num_layers - 1):
        mats.append(w[o:o + hidden_dim * hidden_dim].reshape(hidden_dim, hidden_dim)); o += hidden_dim * hidden_dim
    mats.append(w[o:o + padded_out * hidden_dim].reshape(padded_out, hidden_dim))
    return mats


def act_forward(a, x):
    if a == ACT_RELU:
        return np.where(x > 0, x, 0.0)
    if a == ACT_EXP:
        return np.exp(x)
    if a == ACT_SINE:
        return np.sin(x)
    if a == ACT_SIGMOID:
        return 1.0 / (1.0 + np.exp(-x))
    if a == ACT_SQUAREPLUS:
        y = x * K_ACT
        return 0.5 * (y + np.sqrt(y * y + 4)) / K_ACT
    if a == ACT_SOFTPLUS:
        return np.log(np.exp(x * K_ACT) + 1.0) / K_ACT
    return x


def act_backward_from_post(a, g, post):
    """utils.h:609-664 (transfer using the stored POST-activation)."""
    if a == ACT_RELU:
        return g * (post > 0)
    if a == ACT_EXP:
        return g * post
    if a == ACT_SIGMOID:
        return g * post * (1 - post)
    if a == ACT_SQUAREPLUS:
        y = post * K_ACT
        return g * (y * y / (y * y + 1))
    if a == ACT_SOFTPLUS:
        return g * (1.0 - np.exp(-post * K_ACT))
    if a == ACT_SINE:
        raise ValueError("Sine backward needs pre-activations (unsupported in the reference as well, utils.h:626-630)")
    return g


def round_bf16(x):
    """float -> nearest bfloat16 (ties to even), returned as float32.  bf16 = the upper 16 bits of an IEEE float32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def _h(x, half):
    """Storage rounding of the MLP element type: True / "f16" -> fp16 (the reference), "bf16" -> bfloat16 (the bf16-operand
    build of the kernels, BASELINE config 5), False -> none (float64 throughout)."""
    if half == "bf16":
        return round_bf16(x).astype(np.float64)
    return x.astype(np.float16).astype(np.float64) if half else x.astype(np.float64)


def mlp_forward(x, mats, act=ACT_RELU, out_act=ACT_NONE, half=True):
    """x [B, in]; mats list of [out_k, in_k].  Returns (out [B, out_last] float64, list of stored post-activations)."""
    h = _h(np.asarray(x), half)
    saved = []
    for k, W in enumerate(mats):
        Wk = _h(np.asarray(W), half)
        z = h @ Wk.T
        if k < len(mats) - 1:
            h = _h(act_forward(act, z), half)
            saved.append(h)
        else:
            out = act_forward(out_act, z)
    return out, saved


def mlp_backward(x, mats, grad_out, act=ACT_RELU, half=True):
    """Gradients for out_act == None.  Returns (grad_x [B,in], [dW_k]) in float64.
    Inter-layer gradients are fp16 in the reference (backward_buffer is half); we model that with `half`."""
    x64 = _h(np.asarray(x), half)
    _, saved = mlp_forward(x, mats, act, ACT_NONE, half)
    acts_in = [x64] + saved  # input of every matrix
    g = _h(np.asarray(grad_out), half)
    dWs = [None] * len(mats)
    for k in range(len(mats) - 1, -1, -1):
        Wk = _h(np.asarray(mats[k]), half)
        dWs[k] = g.T @ acts_in[k]
        g = g @ Wk
        if k > 0:
            g = _h(act_backward_from_post(act, g, saved[k - 1]), half)
    return g, dWs
