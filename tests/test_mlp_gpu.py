"""Fused MFMA MLP (lnh_mlp_forward/backward) vs the float64 oracle.

Tolerance: inputs/weights/activations are fp16 in both; the kernel accumulates in fp32 and rounds each stored
activation once to fp16 (the reference accumulates in fp16).  fp16 has 11 significant bits -> one rounding is
<= 2^-11 relative; a 64-term dot product of O(1) values lands within ~3e-3 absolute of the exact value.
"""
import numpy as np
import pytest
import torch

from oracle import mlp_ref

pytestmark = pytest.mark.gpu


def _run_fwd(x, w, in_dim, hidden, nhm, act, out_act=6, fb=None):
    from gpu_util import call, dev, host
    B = x.shape[0]
    y = torch.empty((B, 16), dtype=torch.float16, device="cuda")
    call("lnh_mlp_forward", dev(x), dev(w), B, in_dim, 16, hidden, nhm, act, out_act, fb, y)
    return host(y)


@pytest.mark.parametrize("in_dim,nhm", [(32, 0), (32, 1), (96, 1), (16, 0), (48, 2), (128, 1), (64, 2)])
@pytest.mark.parametrize("B", [64, 1000])
def test_forward_relu(in_dim, nhm, B):
    r = np.random.default_rng(in_dim + nhm)
    x = r.standard_normal((B, in_dim)).astype(np.float16)
    n = mlp_ref.ffmlp_num_params(in_dim, 16, 64, nhm + 1)
    w = (r.uniform(-1, 1, n) * np.sqrt(3 / 64)).astype(np.float16)
    mats = mlp_ref.ffmlp_split_weights(w, in_dim, 16, 64, nhm + 1)
    want, _ = mlp_ref.mlp_forward(x, mats)
    got = _run_fwd(x, w, in_dim, 64, nhm, mlp_ref.ACT_RELU)
    np.testing.assert_allclose(got.astype(np.float64), want, rtol=2e-3, atol=4e-3)


def test_forward_is_not_transposed():
    """Asymmetric one-hot weights: output o must read hidden unit o (catches row/col or k-enumeration mix-ups)."""
    in_dim, B = 32, 48
    x = np.zeros((B, in_dim), np.float16)
    for b in range(B):
        x[b, b % in_dim] = 1 + b
    W0 = np.zeros((64, in_dim), np.float16)
    for h in range(64):
        W0[h, (h * 5 + 3) % in_dim] = 1 + 0.01 * h
    Wo = np.zeros((16, 64), np.float16)
    for o in range(16):
        Wo[o, (o * 7 + 2) % 64] = 1 + 0.1 * o
    w = np.concatenate([W0.ravel(), Wo.ravel()])
    want, _ = mlp_ref.mlp_forward(x, [W0, Wo])
    got = _run_fwd(x, w, in_dim, 64, 0, mlp_ref.ACT_RELU)
    np.testing.assert_allclose(got.astype(np.float64), want, rtol=2e-3, atol=1e-3)


@pytest.mark.parametrize("act", [1, 2, 3, 4, 5, 6])
def test_forward_activations(act):
    r = np.random.default_rng(act)
    x = (r.standard_normal((256, 32)) * 0.5).astype(np.float16)
    w = (r.uniform(-1, 1, mlp_ref.ffmlp_num_params(32, 16, 64, 2)) * 0.15).astype(np.float16)
    mats = mlp_ref.ffmlp_split_weights(w, 32, 16, 64, 2)
    want, _ = mlp_ref.mlp_forward(x, mats, act=act)
    got = _run_fwd(x, w, 32, 64, 1, act)
    np.testing.assert_allclose(got.astype(np.float64), want, rtol=4e-3, atol=4e-3)


def test_forward_buffer_matches_hidden_activations():
    from gpu_util import host
    r = np.random.default_rng(9)
    B = 130
    x = r.standard_normal((B, 32)).astype(np.float16)
    w = (r.uniform(-1, 1, mlp_ref.ffmlp_num_params(32, 16, 64, 2)) * 0.2).astype(np.float16)
    mats = mlp_ref.ffmlp_split_weights(w, 32, 16, 64, 2)
    _, saved = mlp_ref.mlp_forward(x, mats)
    fb = torch.zeros((2, B, 64), dtype=torch.float16, device="cuda")
    _run_fwd(x, w, 32, 64, 1, 0, fb=fb)
    got = host(fb).astype(np.float64)
    np.testing.assert_allclose(got[0], saved[0], rtol=2e-3, atol=3e-3)
    np.testing.assert_allclose(got[1], saved[1], rtol=2e-3, atol=3e-3)


@pytest.mark.parametrize("in_dim,nhm", [(32, 0), (96, 1), (32, 1), (64, 2), (16, 0)])
@pytest.mark.parametrize("B", [128, 1000])
def test_backward(in_dim, nhm, B):
    from gpu_util import call, dev, host, wgrad
    r = np.random.default_rng(in_dim * 3 + nhm)
    x = r.standard_normal((B, in_dim)).astype(np.float16)
    n = mlp_ref.ffmlp_num_params(in_dim, 16, 64, nhm + 1)
    w = (r.uniform(-1, 1, n) * np.sqrt(3 / 64)).astype(np.float16)
    gy = (r.standard_normal((B, 16)) * 0.1).astype(np.float16)
    mats = mlp_ref.ffmlp_split_weights(w, in_dim, 16, 64, nhm + 1)
    gx_want, dws = mlp_ref.mlp_backward(x, mats, gy)
    dw_want = np.concatenate([d.ravel() for d in dws])
    gx = torch.zeros((B, in_dim), dtype=torch.float16, device="cuda")
    dw = torch.zeros(n, dtype=torch.float32, device="cuda")
    call("lnh_mlp_backward", dev(gy), dev(x), dev(w), B, in_dim, 16, 64, nhm, 0, 6, gx, dw, *wgrad())
    np.testing.assert_allclose(host(gx).astype(np.float64), gx_want, rtol=5e-3, atol=2e-3)
    scale = np.abs(dw_want).max()
    np.testing.assert_allclose(host(dw).astype(np.float64), dw_want, rtol=5e-3, atol=2e-3 * scale)
    # weights-only variant (grad_inputs == NULL)
    dw2 = torch.zeros(n, dtype=torch.float32, device="cuda")
    call("lnh_mlp_backward", dev(gy), dev(x), dev(w), B, in_dim, 16, 64, nhm, 0, 6, None, dw2, *wgrad())
    np.testing.assert_allclose(host(dw2), host(dw), rtol=1e-4, atol=1e-4 * scale)


@pytest.mark.parametrize("in_dim,nhm", [(32, 0), (16, 1), (96, 2), (128, 1)])
def test_hidden_32_forward_backward(in_dim, nhm):
    """hidden = 32 (two 16-row tiles per layer) — same kernels, HT = 2."""
    from gpu_util import call, dev, host, wgrad
    B, H = 777, 32
    r = np.random.default_rng(in_dim * 5 + nhm)
    x = r.standard_normal((B, in_dim)).astype(np.float16)
    n = mlp_ref.ffmlp_num_params(in_dim, 16, H, nhm + 1)
    w = (r.uniform(-1, 1, n) * np.sqrt(3 / H)).astype(np.float16)
    mats = mlp_ref.ffmlp_split_weights(w, in_dim, 16, H, nhm + 1)
    want, _ = mlp_ref.mlp_forward(x, mats)
    got = _run_fwd(x, w, in_dim, H, nhm, mlp_ref.ACT_RELU)
    np.testing.assert_allclose(got.astype(np.float64), want, rtol=2e-3, atol=4e-3)
    gy = (r.standard_normal((B, 16)) * 0.1).astype(np.float16)
    gx_want, dws = mlp_ref.mlp_backward(x, mats, gy)
    dw_want = np.concatenate([d.ravel() for d in dws])
    gx = torch.zeros((B, in_dim), dtype=torch.float16, device="cuda")
    dw = torch.zeros(n, dtype=torch.float32, device="cuda")
    call("lnh_mlp_backward", dev(gy), dev(x), dev(w), B, in_dim, 16, H, nhm, 0, 6, gx, dw, *wgrad())
    np.testing.assert_allclose(host(gx).astype(np.float64), gx_want, rtol=5e-3, atol=2e-3)
    np.testing.assert_allclose(host(dw).astype(np.float64), dw_want, rtol=5e-3, atol=2e-3 * np.abs(dw_want).max())


# ------------------------------------------------------------------------------------------------ bf16 operands
def _bf(a):
    """numpy float array -> CUDA bfloat16 tensor (values are rounded to bf16 by the oracle's own model first)."""
    return torch.from_numpy(mlp_ref.round_bf16(a)).cuda().to(torch.bfloat16)


@pytest.mark.parametrize("in_dim,nhm,H", [(32, 0, 64), (96, 1, 64), (48, 2, 64), (32, 1, 32)])
def test_bf16_forward_backward(in_dim, nhm, H):
    """lnh_mlp_forward_bf16 / lnh_mlp_backward_bf16 (v_mfma_f32_16x16x32_bf16 operands, BASELINE config 5) vs the
    oracle with bfloat16 storage roundings.  bf16 keeps 8 significant bits: one rounding is <= 2^-9 relative, so the
    bounds are 8x those of the fp16 tests (2^-8 vs 2^-11)."""
    from gpu_util import call, host, wgrad
    B = 1000
    r = np.random.default_rng(in_dim * 7 + nhm)
    x = mlp_ref.round_bf16(r.standard_normal((B, in_dim)))
    n = mlp_ref.ffmlp_num_params(in_dim, 16, H, nhm + 1)
    w = mlp_ref.round_bf16(r.uniform(-1, 1, n) * np.sqrt(3 / H))
    gy = mlp_ref.round_bf16(r.standard_normal((B, 16)) * 0.1)
    mats = mlp_ref.ffmlp_split_weights(w, in_dim, 16, H, nhm + 1)
    want, _ = mlp_ref.mlp_forward(x, mats, half="bf16")
    y = torch.empty((B, 16), dtype=torch.bfloat16, device="cuda")
    call("lnh_mlp_forward_bf16", _bf(x), _bf(w), B, in_dim, 16, H, nhm, 0, 6, None, y)
    np.testing.assert_allclose(host(y.float()).astype(np.float64), want, rtol=1.6e-2, atol=3e-2)
    gx_want, dws = mlp_ref.mlp_backward(x, mats, gy, half="bf16")
    dw_want = np.concatenate([d.ravel() for d in dws])
    gx = torch.zeros((B, in_dim), dtype=torch.bfloat16, device="cuda")
    dw = torch.zeros(n, dtype=torch.float32, device="cuda")
    call("lnh_mlp_backward_bf16", _bf(gy), _bf(x), _bf(w), B, in_dim, 16, H, nhm, 0, 6, gx, dw, *wgrad())
    np.testing.assert_allclose(host(gx.float()).astype(np.float64), gx_want, rtol=4e-2, atol=1.6e-2)
    np.testing.assert_allclose(host(dw).astype(np.float64), dw_want, rtol=4e-2, atol=1.6e-2 * np.abs(dw_want).max())


def test_bf16_ffmlp_module_under_bf16_autocast():
    from lidarnerf.ffmlp import FFMLP
    m = FFMLP(32, 3, 64, 2).cuda()
    x = torch.randn(500, 32, device="cuda", requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
    assert y.dtype == torch.bfloat16 and y.shape == (500, 3)
    with torch.autocast("cuda", dtype=torch.float16):
        y16 = m(x)
    torch.testing.assert_close(y.float(), y16.float(), rtol=3e-2, atol=3e-2)  # same function, 8 vs 11 significant bits
    y.float().square().sum().backward()
    assert torch.isfinite(m.weights.grad).all() and float(m.weights.grad.abs().max()) > 0


@pytest.mark.parametrize("sfx", ["", "_bf16"])
@pytest.mark.parametrize("hidden,nhm,in_dim,act", [(128, 1, 48, 0), (256, 2, 64, 0), (64, 4, 32, 0), (32, 3, 16, 0),
                                                    (128, 0, 128, 0), (256, 1, 32, 3), (128, 2, 96, 5)])
def test_wide_kernels_vs_oracle(hidden, nhm, in_dim, act, sfx):
    """csrc/mlp_wide.hip through the C ABI: lnh_mlp_forward at hidden 128 / 256 and with more than two hidden matrices
    (output + every saved hidden activation), lnh_mlp_backward_data (the activation gradients of every layer + the input
    gradient) against the oracle's MLP (16-bit storage of every layer), a batch that is not a multiple of anything, ReLU /
    sigmoid / softplus; lnh_mlp_backward refuses these shapes and names the entry point that serves them."""
    from gpu_util import call, wgrad
    dt = torch.bfloat16 if sfx else torch.float16
    npd = np.float32
    r = np.random.default_rng(hidden + nhm)
    B = 1000
    acts = {0: "relu", 3: "sigmoid", 5: "softplus"}
    x = torch.from_numpy((r.standard_normal((B, in_dim)) * 0.5).astype(npd)).to(dt)
    std = np.sqrt(3 / hidden)
    mats = [r.uniform(-std, std, (hidden, in_dim))] + [r.uniform(-std, std, (hidden, hidden)) for _ in range(nhm)] + \
        [r.uniform(-std, std, (16, hidden))]
    mats = [torch.from_numpy(m.astype(npd)).to(dt) for m in mats]
    w = torch.cat([m.reshape(-1) for m in mats]).cuda()
    y = torch.empty((B, 16), dtype=dt, device="cuda")
    fb = torch.empty((nhm + 1, B, hidden), dtype=dt, device="cuda")
    call("lnh_mlp_forward" + sfx, x.cuda(), w, B, in_dim, 16, hidden, nhm, act, 6, fb, y)
    # reference chain in fp64 on the 16-bit values, every layer's activation rounded to the element type
    f = {0: torch.relu, 3: torch.sigmoid, 5: lambda v: torch.log1p(torch.exp(v * 10.0)) / 10.0}[act]
    hs, h = [], x.double()
    for m in mats[:-1]:
        h = f(h @ m.double().t()).to(dt).double()
        hs.append(h)
    want_y = (h @ mats[-1].double().t())
    tol = 4e-3 if not sfx else 3e-2
    for l in range(nhm + 1):
        torch.testing.assert_close(fb[l].cpu().double(), hs[l], rtol=tol, atol=tol)
    torch.testing.assert_close(y.cpu().double(), want_y, rtol=tol, atol=tol * max(1.0, float(want_y.abs().max())))
    # backward (data): from the kernel's own saved activations
    gy = torch.zeros((B, 16), dtype=dt)
    gy[:, :5] = torch.from_numpy((r.standard_normal((B, 5)) * 0.1).astype(npd)).to(dt)
    wt = torch.cat([m.t().contiguous().reshape(-1) for m in mats]).cuda()
    gb = torch.empty((nhm + 1, B, hidden), dtype=dt, device="cuda")
    gx = torch.empty((B, in_dim), dtype=dt, device="cuda")
    call("lnh_mlp_backward_data" + sfx, gy.cuda(), fb, wt, B, in_dim, 16, hidden, nhm, act, gb, gx)
    fbh = fb.cpu().double()
    dact = {0: lambda p: (p > 0).double(), 3: lambda p: p * (1 - p), 5: lambda p: 1 - torch.exp(-p * 10.0)}[act]
    g = (gy.double() @ mats[-1].double()) * dact(fbh[nhm])
    want_gb = [None] * (nhm + 1)
    want_gb[nhm] = g.to(dt).double()
    for m in range(nhm - 1, -1, -1):
        g = (want_gb[m + 1] @ mats[m + 1].double()) * dact(fbh[m])
        want_gb[m] = g.to(dt).double()
    want_gx = want_gb[0] @ mats[0].double()
    for l in range(nhm + 1):
        scale = float(want_gb[l].abs().max())
        torch.testing.assert_close(gb[l].cpu().double(), want_gb[l], rtol=tol, atol=tol * scale)
    torch.testing.assert_close(gx.cpu().double(), want_gx, rtol=tol, atol=tol * float(want_gx.abs().max()))
    # the one-kernel backward does not exist for these shapes: refused, and the message names the split
    gw = torch.zeros(w.numel(), device="cuda")
    if hidden >= 128 or nhm > 2:
        with pytest.raises(RuntimeError, match="lnh_mlp_backward_data"):
            call("lnh_mlp_backward" + sfx, gy.cuda(), x.cuda(), w, B, in_dim, 16, hidden, nhm, act, 6, gx, gw, *wgrad())


@pytest.mark.parametrize("sfx", ["", "_bf16"])
@pytest.mark.parametrize("M,N,B", [(16, 16, 1000), (16, 256, 4099), (256, 16, 777), (128, 128, 100_000), (256, 256, 65_537),
                                   (64, 48, 33), (128, 112, 31), (32, 256, 300_000)])
def test_wide_weight_gradient_kernel(M, N, B, sfx):
    """lnh_mlp_wgrad (csrc/mlp_wgrad.hip): grad_weights[M, N] += grad^T acts, the batch contraction of the wide MLPs' backward
    (the reference: split-K CUTLASS GEMMs, ffmlp.cu:1107-1263) — against fp64 on the same 16-bit values (products of 16-bit
    values are exact in fp32; what differs is the order of the fp32 sums), ADDED to what the gradient holds, and the same
    bits on every launch (fixed-order sum, csrc/wgrad.h)."""
    from gpu_util import call, wgrad
    dt = torch.bfloat16 if sfx else torch.float16
    g = torch.Generator().manual_seed(M * 1000 + N + B)
    G = (torch.randn(B, M, generator=g) * 0.1).to(dt).cuda()
    A = (torch.randn(B, N, generator=g) * 0.5).to(dt).cuda()
    want = G.double().t() @ A.double()
    outs = []
    for _ in range(3):
        gw = torch.full((M, N), 2.0, device="cuda")
        call("lnh_mlp_wgrad" + sfx, G, A, B, M, N, gw, *wgrad())
        outs.append(gw)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    scale = float(want.abs().max())
    torch.testing.assert_close(outs[0].double() - 2.0, want, rtol=1e-5, atol=2e-6 * scale + 1e-6 * np.sqrt(B))
    # refused by name: widths outside 16 .. 256 / not multiples of 16, a missing workspace
    with pytest.raises(RuntimeError, match="multiples of 16"):
        call("lnh_mlp_wgrad" + sfx, G, A, B, M, N + 8, outs[0], *wgrad())
    with pytest.raises(RuntimeError, match="lnh_wgrad_workspace_bytes"):
        call("lnh_mlp_wgrad" + sfx, G, A, B, M, N, outs[0], None, 0)


@pytest.mark.parametrize("hidden,layers,in_dim", [(16, 2, 32), (128, 3, 48), (256, 2, 64), (64, 5, 32), (256, 4, 32)])
def test_ffmlp_module_all_reference_widths(hidden, layers, in_dim):
    """Every width the reference accepts (ffmlp.py:202-209: 16 .. 256) and deeper nets, all on fused kernels (hidden 16 on
    the hidden-32 kernels; 128 / 256 and more than 3 hidden layers on csrc/mlp_wide.hip with the weight gradients from
    csrc/mlp_wgrad.hip): forward and gradients vs the oracle, and the launches that ran."""
    from lidarnerf.ffmlp import FFMLP
    from lidarnerf import _hip
    m = FFMLP(in_dim, 5, hidden, layers).cuda()
    with torch.no_grad():
        m.weights.copy_(m.weights.half().float())
    r = np.random.default_rng(hidden)
    x = r.standard_normal((300, in_dim)).astype(np.float16)
    xt = torch.from_numpy(x.astype(np.float32)).cuda().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        y = m(xt)
    assert y.shape == (300, 5) and y.dtype == torch.float16
    mats = mlp_ref.ffmlp_split_weights(m.weights.detach().cpu().numpy(), in_dim, 5, hidden, layers)
    want, _ = mlp_ref.mlp_forward(x, mats)
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), want[:, :5], rtol=4e-3, atol=6e-3)
    gy = np.zeros((300, 16), np.float16)
    gy[:, :5] = (r.standard_normal((300, 5)) * 0.1).astype(np.float16)
    y.backward(torch.from_numpy(gy[:, :5]).cuda())
    gx_want, dws = mlp_ref.mlp_backward(x, mats, gy)
    dw_want = np.concatenate([d.ravel() for d in dws])
    np.testing.assert_allclose(xt.grad.cpu().numpy(), gx_want, rtol=1e-2, atol=4e-3)
    np.testing.assert_allclose(m.weights.grad.cpu().numpy(), dw_want, rtol=1e-2, atol=4e-3 * np.abs(dw_want).max())
    # which kernels ran: one forward launch; one-kernel backward for the narrow shapes, the data kernel for the wide ones
    names = ["lnh_mlp_forward", "lnh_mlp_backward", "lnh_mlp_backward_data", "lnh_mlp_wgrad"]
    _hip.enable_timers(names)
    with torch.autocast("cuda", dtype=torch.float16):
        m(xt.detach().requires_grad_(True)).float().sum().backward()
    calls = _hip.disable_timers()
    n = {k: len(calls.get(k, [])) for k in names}
    wide = hidden >= 128 or layers - 1 > 2
    assert n == {"lnh_mlp_forward": 1, "lnh_mlp_backward": 0 if wide else 1, "lnh_mlp_backward_data": 1 if wide else 0,
                 "lnh_mlp_wgrad": layers + 1 if wide else 0}, n  # (one weight-gradient launch per matrix — `layers` hidden-side matrices + the output matrix: no library GEMM)
    # what still has no kernel runs as the library-GEMM chain (the reference's constructor accepts it); strict_fused refuses
    with pytest.raises(RuntimeError, match="no fused MFMA kernel"):
        FFMLP(256, 5, hidden, layers, strict_fused=True)
    big = FFMLP(256, 5, hidden, layers).cuda()
    assert big.gemm_chain
    with torch.autocast("cuda", dtype=torch.float16):
        assert big(torch.randn(64, 256, device="cuda")).shape == (64, 5)


def test_ffmlp_module_hidden_32():
    from lidarnerf.ffmlp import FFMLP
    m = FFMLP(32, 3, 32, 3).cuda()
    x = torch.randn(500, 32, device="cuda", requires_grad=True)
    with torch.autocast("cuda", dtype=torch.float16):
        y = m(x)
    y.float().square().sum().backward()
    assert y.shape == (500, 3) and torch.isfinite(m.weights.grad).all() and float(m.weights.grad.abs().max()) > 0


def test_unsupported_shapes_fail_loudly():
    from lidarnerf import _hip
    t = torch.zeros(16, device="cuda")
    with pytest.raises(RuntimeError, match="input_dim should be 16"):
        _hip.call("lnh_mlp_forward", t.data_ptr(), t.data_ptr(), 16, 20, 16, 64, 0, 0, 6, None, t.data_ptr())
    with pytest.raises(RuntimeError, match="hidden_dim"):
        _hip.call("lnh_mlp_forward", t.data_ptr(), t.data_ptr(), 16, 32, 16, 96, 0, 0, 6, None, t.data_ptr())
