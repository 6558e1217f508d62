"""Unit parity of the fused LiDAR-field kernels (lidar_field.hip, DensityIO variants of the MLP kernels) against
plain PyTorch fp32/fp64 restatements of the same reference arithmetic (renderer.py:164-167, 217-243;
network.py:162-237; activation.py)."""
import numpy as np
import pytest
import torch

from oracle import mlp_ref, render_ref

pytestmark = pytest.mark.gpu


def test_sample_points():
    from gpu_util import call
    g = torch.Generator().manual_seed(0)
    N, T = 37, 100
    o = (torch.rand(N, 3, generator=g) - 0.5).cuda()
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).cuda()
    z = (torch.rand(N, T, generator=g) * 2.5).cuda()  # some samples leave the box -> clipped
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1]).cuda()
    x01 = torch.empty((N * T, 3), device="cuda")
    call("lnh_lidar_sample_points", o, d, z, aabb, 1.0, N, T, T, 0, x01)
    p = o[:, None, :] + d[:, None, :] * z[..., None]
    want = (torch.min(torch.max(p, aabb[:3]), aabb[3:]) + 1.0) / 2.0
    torch.testing.assert_close(x01.view(N, T, 3), want, rtol=0, atol=1e-7)
    assert float(x01.min()) >= 0.0 and float(x01.max()) <= 1.0


def test_merge_weights():
    from gpu_util import call
    g = torch.Generator().manual_seed(1)
    N, T = 29, 832
    z = torch.sort(torch.rand(N, T, generator=g) * 0.8 + 0.01, dim=1)[0]
    sigma_pt = torch.rand(N, T, generator=g) * 30
    perm = torch.stack([torch.randperm(T, generator=g) for _ in range(N)]).int()
    sd = torch.full((N,), 1e-3)
    sigma_m = torch.empty((N, T), device="cuda")
    w = torch.empty((N, T), device="cuda")
    call("lnh_lidar_merge_weights", z.cuda(), sigma_pt.cuda(), perm.cuda(), sd.cuda(), N, T, 1.0, sigma_m, w)
    want_sigma = torch.gather(sigma_pt, 1, perm.long())
    assert torch.equal(sigma_m.cpu(), want_sigma)
    want_w, _ = render_ref.weights_from_sigma(z, want_sigma, sd[:, None])
    torch.testing.assert_close(w.cpu(), want_w, rtol=2e-5, atol=1e-7)


def _sigma_net(seed):
    r = np.random.default_rng(seed)
    w0 = (r.uniform(-1, 1, (64, 32)) * 0.4).astype(np.float16)
    w1 = (r.uniform(-1, 1, (16, 64)) * 0.3).astype(np.float16)
    return w0, w1


# (420, 768, ...): 322 560 points > one pass of the persistent grids (forward 262 144, backward 65 536 per pass)
@pytest.mark.parametrize("N,Tc,Ttot,off", [(13, 64, 80, 16), (5, 768, 832, 0), (7, 16, 16, 0), (420, 768, 832, 64)])
def test_density_mlp_forward_backward(N, Tc, Ttot, off):
    from gpu_util import call, dev, host, wgrad
    B = N * Tc
    r = np.random.default_rng(2)
    feat = r.standard_normal((16, B, 2)).astype(np.float16)  # level-major, as the encoder writes it
    w0, w1 = _sigma_net(3)
    wflat = np.concatenate([w0.ravel(), w1.ravel()])
    x_rows = feat.transpose(1, 0, 2).reshape(B, 32)
    want, _ = mlp_ref.mlp_forward(x_rows, [w0, w1])
    h16 = torch.full((N * Ttot, 16), float("nan"), dtype=torch.float16, device="cuda")
    sigma = torch.full((N * Ttot,), float("nan"), device="cuda")
    call("lnh_density_mlp_forward", dev(feat), dev(wflat), B, Tc, Ttot, off, 0, h16, sigma)
    rows = (np.arange(B) // Tc) * Ttot + off + (np.arange(B) % Tc)
    got = host(h16).astype(np.float64)
    np.testing.assert_allclose(got[rows], want, rtol=2e-3, atol=4e-3)
    untouched = np.setdiff1d(np.arange(N * Ttot), rows)
    assert np.isnan(got[untouched]).all()  # strided destination: other slots are not written
    # sigma = exp(fp16-rounded pre-activation) exactly (trunc_exp forward)
    np.testing.assert_allclose(host(sigma)[rows], np.exp(got[rows, 0]).astype(np.float32), rtol=2e-6)
    # backward
    gy = (r.standard_normal((N * Ttot, 16)) * 0.1).astype(np.float16)
    gx_want, dws = mlp_ref.mlp_backward(x_rows, [w0, w1], gy[rows])
    gfeat = torch.empty((16, B, 2), dtype=torch.float16, device="cuda")
    gw = torch.zeros(wflat.size, dtype=torch.float32, device="cuda")
    call("lnh_density_mlp_backward", dev(gy), dev(feat), dev(wflat), B, Tc, Ttot, off, gfeat, gw, *wgrad())
    got_gx = host(gfeat).astype(np.float64).transpose(1, 0, 2).reshape(B, 32)
    # fp16 storage of the hidden layer: a pre-activation within an fp16 ulp of 0 flips its ReLU mask, which moves a few
    # elements in ten million beyond the bulk tolerance
    bad = np.abs(got_gx - gx_want) > 2e-3 + 5e-3 * np.abs(gx_want)
    assert bad.mean() < 1e-5, bad.sum()
    np.testing.assert_allclose(got_gx, gx_want, rtol=5e-3, atol=1e-2)
    dw_want = np.concatenate([d.ravel() for d in dws])
    np.testing.assert_allclose(host(gw), dw_want, rtol=5e-3, atol=2e-3 * np.abs(dw_want).max())


def _color_reference(h16, perm, weights, cdir, W0g, W1, W2, g_rgb=None, g_sigma=None):
    """fp64 autograd restatement of the colour head on merged samples (+ trunc_exp backward column)."""
    N, T = perm.shape
    x = h16.view(N, T, 16).double().requires_grad_(True)
    xg = torch.gather(x, 1, perm.long().unsqueeze(-1).expand(-1, -1, 16))
    params = [p.half().double().requires_grad_(True) for p in (W0g, W1, W2)]
    cd = cdir.double().requires_grad_(True)
    rh = lambda t: t + (t.half().double() - t).detach()  # fp16 storage rounding, straight-through gradient
    h0 = rh(torch.relu(xg @ params[0].t() + cd[:, None, :]))
    h1 = rh(torch.relu(h0 @ params[1].t()))
    o = rh(h1 @ params[2].t())
    mask = (weights > 1e-4)[..., None]
    rgb = torch.sigmoid(o) * mask
    if g_rgb is None:
        return rgb.detach()
    (rgb * g_rgb.double()).sum().backward()
    gx = x.grad.clone()
    pre = x.detach()[..., 0].clamp(-15, 15)
    g_sig_pt = torch.zeros(N, T, dtype=torch.float64)
    g_sig_pt.scatter_(1, perm.long(), g_sigma.double())  # merged -> point order
    gx[..., 0] = g_sig_pt * torch.exp(pre)
    return rgb.detach(), gx, [p.grad for p in params], cd.grad


# (2100, 64): more rays than resident waves (256 workgroups x 4), so waves walk SEVERAL rays — the flattened, software-
# pipelined (ray, step) loop crosses ray boundaries (direction-term reload, per-ray S write)
# (5, 20): a ray shorter than one span; (3, 2200): three groups per ray, three busy waves
# (6, 1100): a ray spans two 1024-sample groups of the backward's span iterator, the second one partial
# "front": only a leading stretch of every ray is active (what a LiDAR ray looks like once trained) — most 32-sample spans
# are transparent and take the batched row store of the iterator; single active samples in otherwise transparent rays
@pytest.mark.parametrize("N,T,pattern", [(8, 64, "random"), (5, 832, "random"), (3, 48, "random"), (2100, 64, "random"),
                                         (6, 1100, "random"), (9, 832, "front"), (7, 1100, "front"), (1300, 96, "front"),
                                         (5, 20, "random"), (3, 2200, "front")])
def test_color_head_forward_backward(N, T, pattern):
    from gpu_util import call, wgrad
    g = torch.Generator().manual_seed(N + T)
    h16 = (torch.randn(N * T, 16, generator=g) * 0.5).half()
    perm = torch.stack([torch.randperm(T, generator=g) for _ in range(N)]).int()
    weights = torch.rand(N, T, generator=g) * 2.5e-4  # ~60 % of the samples above the 1e-4 mask threshold
    if pattern == "front":
        n_act = torch.randint(0, T // 2, (N,), generator=g)
        weights = torch.where(torch.arange(T)[None, :] < n_act[:, None], weights, torch.full_like(weights, 1e-6))
        weights[1] = 1e-6
        weights[1, T - 1] = 1e-2   # one active sample, in the last span of the ray
        weights[2] = 1e-6
        weights[2, 33] = 1e-2      # ... in the second span
    weights[0] = 0.0  # a fully masked ray
    cdir = torch.randn(N, 64, generator=g)
    W0g = torch.cat([torch.zeros(64, 1), torch.randn(64, 15, generator=g) * 0.3], 1)
    W1 = torch.randn(64, 64, generator=g) * 0.2
    W2 = torch.randn(2, 64, generator=g) * 0.2
    w16 = torch.cat([W0g.reshape(-1), W1.reshape(-1), torch.nn.functional.pad(W2, (0, 0, 0, 14)).reshape(-1)]).half()
    rgb = torch.empty((N, T, 2), device="cuda")
    call("lnh_lidar_color_forward", h16.cuda(), perm.cuda(), weights.cuda(), cdir.cuda(), w16.cuda(), N, T, rgb)
    g_rgb = torch.randn(N, T, 2, generator=g)
    g_sigma = torch.randn(N, T, generator=g)
    want_rgb, want_gx, want_gw, want_gcd = _color_reference(h16, perm, weights, cdir, W0g, W1, W2, g_rgb, g_sigma)
    torch.testing.assert_close(rgb.cpu().double(), want_rgb, rtol=2e-3, atol=2e-3)
    assert float(rgb[0].abs().max()) == 0.0
    g_h16 = torch.full((N * T, 16), float("nan"), dtype=torch.float16, device="cuda")
    g_w = torch.zeros(w16.numel(), device="cuda")
    S = torch.empty((N, 64), device="cuda")
    call("lnh_lidar_color_backward", g_rgb.cuda(), g_sigma.cuda(), h16.cuda(), perm.cuda(), weights.cuda(), cdir.cuda(),
         w16.cuda(), N, T, g_h16, g_w, S, *wgrad())
    got_gx = g_h16.cpu().double().view(N, T, 16)
    assert torch.isfinite(got_gx).all()  # every point row written exactly once
    scale = want_gx.abs().max().item()
    torch.testing.assert_close(got_gx, want_gx, rtol=1e-2, atol=4e-3 * scale)
    gw = g_w.cpu().double()
    for got, want in ((gw[:1024].view(64, 16), want_gw[0]), (gw[1024:1024 + 4096].view(64, 64), want_gw[1]),
                      (gw[1024 + 4096:].view(16, 64)[:2], want_gw[2])):
        torch.testing.assert_close(got, want, rtol=1e-2, atol=5e-3 * want.abs().max().item())
    assert float(gw[1024 + 4096:].view(16, 64)[2:].abs().max()) == 0.0  # padded output rows get no gradient
    # per-ray sum of d(hidden0) == gradient w.r.t. the per-ray direction bias
    torch.testing.assert_close(S.cpu().double(), want_gcd, rtol=1e-2, atol=5e-3 * want_gcd.abs().max().item())


@pytest.mark.parametrize("sfx", ["", "_bf16"])
def test_color_backward_from_image_gradient_equals_explicit_rgb_gradient(sfx):
    """lnh_lidar_color_backward_image(grad_image) == lnh_lidar_color_backward(grad_rgb = weights (x) grad_image), bit for bit
    (the product the compositing backward would have written is formed in the kernel); fp16 and bf16 builds."""
    from gpu_util import call, wgrad
    dt16 = torch.bfloat16 if sfx else torch.float16
    N, T = 37, 832
    g = torch.Generator().manual_seed(99)
    h16 = (torch.randn(N * T, 16, generator=g) * 0.5).to(dt16).cuda()
    perm = torch.stack([torch.randperm(T, generator=g) for _ in range(N)]).int().cuda()
    weights = (torch.rand(N, T, generator=g) * 2.5e-4).cuda()
    cdir = torch.randn(N, 64, generator=g).cuda()
    w16 = (torch.randn(64 * 16 + 64 * 64 + 16 * 64, generator=g) * 0.2).to(dt16).cuda()
    g_image = torch.randn(N, 2, generator=g).cuda()
    g_sigma = torch.randn(N, T, generator=g).cuda()
    g_rgb = (weights[..., None] * g_image[:, None, :]).contiguous()
    outs = []
    for mode in ("rgb", "image"):
        g_h16 = torch.full((N * T, 16), float("nan"), dtype=dt16, device="cuda")
        g_w = torch.zeros(w16.numel(), device="cuda")
        S = torch.empty((N, 64), device="cuda")
        if mode == "rgb":
            call("lnh_lidar_color_backward" + sfx, g_rgb, g_sigma, h16, perm, weights, cdir, w16, N, T, g_h16, g_w, S, *wgrad())
        else:
            call("lnh_lidar_color_backward_image" + sfx, g_image, g_sigma, h16, perm, weights, cdir, w16, N, T, g_h16, g_w, S, *wgrad())
        outs.append((g_h16, S, g_w))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # (the weight gradient is flushed with float atomics from several workgroups: same values, order-dependent last bits)
    torch.testing.assert_close(outs[0][2], outs[1][2], rtol=1e-5, atol=1e-6 * float(outs[0][2].abs().max()))


@pytest.mark.parametrize("N,T,pattern", [(9, 832, "random"), (7, 100, "front"), (1100, 96, "front"), (3, 2048, "random")])
def test_color_composite_forward_equals_the_three_separate_entry_points(N, T, pattern):
    """lnh_lidar_color_composite_forward (one wave per ray) against lnh_lidar_merge_weights + lnh_lidar_color_forward +
    lnh_lidar_composite_forward on the same inputs: sigma_m, weights, weights_sum, depth and rgb bit for bit, image to
    fp32 rounding (it is summed by other lanes)."""
    from gpu_util import call
    g = torch.Generator().manual_seed(3 * N + T)
    z = torch.sort(torch.rand(N, T, generator=g) * 0.8 + 0.01, dim=1)[0].cuda()
    perm = torch.stack([torch.randperm(T, generator=g) for _ in range(N)]).int().cuda()
    sigma_pt = (torch.rand(N, T, generator=g) * (60.0 if pattern == "front" else 3.0)).cuda()
    if pattern == "front":
        sigma_pt[0] = 0.0  # a ray without any weight above the threshold
    sd = torch.full((N,), 0.8 / T).cuda()
    h16 = (torch.randn(N * T, 16, generator=g) * 0.5).half().cuda()
    cdir = torch.randn(N, 64, generator=g).cuda()
    w16 = (torch.randn(64 * 16 + 64 * 64 + 16 * 64, generator=g) * 0.2).half().cuda()
    ds = 1.0
    sig_a, wts_a = torch.empty(N, T, device="cuda"), torch.empty(N, T, device="cuda")
    rgb_a = torch.full((N, T, 2), float("nan"), device="cuda")
    ws_a, dp_a, im_a = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, 2, device="cuda")
    call("lnh_lidar_merge_weights", z, sigma_pt, perm, sd, N, T, ds, sig_a, wts_a)
    call("lnh_lidar_color_forward", h16, perm, wts_a, cdir, w16, N, T, rgb_a)
    call("lnh_lidar_composite_forward", z, sig_a, rgb_a, sd, N, T, 2, ds, None, ws_a, dp_a, im_a)
    sig_b, wts_b = torch.empty(N, T, device="cuda"), torch.empty(N, T, device="cuda")
    rgb_b = torch.full((N, T, 2), float("nan"), device="cuda")
    ws_b, dp_b, im_b = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, 2, device="cuda")
    call("lnh_lidar_color_composite_forward", z, sigma_pt, perm, sd, h16, cdir, w16, N, T, ds, sig_b, wts_b, rgb_b, ws_b,
         dp_b, im_b)
    assert float((wts_a > 1e-4).float().mean()) > 0.02  # the colour head does run
    for a, b in ((sig_a, sig_b), (wts_a, wts_b), (rgb_a, rgb_b), (ws_a, ws_b), (dp_a, dp_b)):
        assert torch.equal(a, b)
    torch.testing.assert_close(im_b, im_a, rtol=2e-6, atol=1e-7)
    with pytest.raises(RuntimeError, match="LDS"):
        call("lnh_lidar_color_composite_forward", z, sigma_pt, perm, sd, h16, cdir, w16, 1, 2049, ds, sig_b, wts_b, rgb_b,
             ws_b, dp_b, im_b)


# ---------------------------------------------------------------------------------------------- glue kernels
def test_coarse_samples_match_torch_linspace_form():
    from lidarnerf import _hip
    N, T = 37, 768
    near = float(np.float32(0.0107848535))
    far = float(np.float32(near) * np.float32(81.0))
    u = torch.rand(N, T, device="cuda")
    z = torch.empty(N, T, device="cuda")
    for noise in (None, u):
        _hip.call("lnh_lidar_coarse_samples", None if noise is None else noise.data_ptr(), N, T, near, far, z.data_ptr())
        nears = torch.full((N, 1), near, device="cuda")
        fars = torch.full((N, 1), far, device="cuda")
        want = nears + (fars - nears) * torch.linspace(0.0, 1.0, T, device="cuda").unsqueeze(0)
        if noise is not None:
            want = want + (noise - 0.5) * ((fars - nears) / T)
        torch.testing.assert_close(z, want, rtol=0, atol=2.5e-7)  # same fp32 expression; linspace itself within an ulp or two


def test_coarse_sample_points_equals_the_two_separate_entry_points():
    """lnh_lidar_coarse_sample_points == lnh_lidar_coarse_samples followed by lnh_lidar_sample_points, bit for bit, and only
    the coarse slots of the [N, T+t] coordinate buffer are written."""
    from gpu_util import call
    N, T, Ttot = 29, 768, 832
    g = torch.Generator().manual_seed(21)
    o = ((torch.rand(N, 3, generator=g) - 0.5) * 0.05).cuda()
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).cuda()
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32).cuda()
    near = float(np.float32(0.0107848535))
    far = float(np.float32(near) * np.float32(81.0))
    u = torch.rand(N * T, generator=g).cuda()
    for noise in (None, u):
        z_a = torch.empty(N, T, device="cuda")
        x_a = torch.full((N * Ttot, 3), float("nan"), device="cuda")
        call("lnh_lidar_coarse_samples", noise, N, T, near, far, z_a)
        call("lnh_lidar_sample_points", o, d, z_a, aabb, 1.0, N, T, Ttot, 0, x_a)
        z_b = torch.empty(N, T, device="cuda")
        x_b = torch.full((N * Ttot, 3), float("nan"), device="cuda")
        call("lnh_lidar_coarse_sample_points", noise, o, d, aabb, 1.0, N, T, Ttot, near, far, z_b, x_b)
        assert torch.equal(z_a, z_b)
        assert torch.equal(torch.nan_to_num(x_a, nan=-7.0), torch.nan_to_num(x_b, nan=-7.0))
        assert bool(torch.isnan(x_b.view(N, Ttot, 3)[:, T:]).all())


def test_dir_term_and_weight_packing():
    from lidarnerf import _hip
    torch.manual_seed(0)
    N, K = 1001, 75
    enc = torch.randn(N, K, device="cuda")
    big = torch.randn(64, 96, device="cuda") * 0.2          # strided view, like the tcnn-shaped flat parameter
    wc0 = big[:, :K + 15]
    enc16 = torch.empty_like(enc)
    cdir = torch.empty(N, 64, device="cuda")
    _hip.call("lnh_lidar_dir_term", enc.data_ptr(), wc0.data_ptr(), wc0.stride(0), N, K, enc16.data_ptr(), cdir.data_ptr())
    want16 = enc.half().float()
    assert torch.equal(enc16, want16)
    want = want16.double() @ wc0[:, :K].half().double().t()
    torch.testing.assert_close(cdir.double(), want, rtol=1e-5, atol=1e-5)
    ws0, ws1 = torch.randn(64, 32, device="cuda"), torch.randn(16, 64, device="cuda")
    wc1, wc2 = torch.randn(64, 64, device="cuda"), torch.randn(2, 64, device="cuda")
    wsig = torch.empty(64 * 32 + 16 * 64, dtype=torch.half, device="cuda")
    wcol = torch.empty(64 * 16 + 64 * 64 + 16 * 64, dtype=torch.half, device="cuda")
    _hip.call("lnh_lidar_pack_weights", ws0.data_ptr(), 32, ws1.data_ptr(), 64, wc0.data_ptr(), wc0.stride(0), K,
              wc1.data_ptr(), 64, wc2.data_ptr(), 64, wsig.data_ptr(), wcol.data_ptr())
    assert torch.equal(wsig, torch.cat([ws0.reshape(-1), ws1.reshape(-1)]).half())
    w0g = torch.cat([torch.zeros(64, 1, device="cuda"), wc0[:, K:K + 15]], dim=1)
    w2p = torch.nn.functional.pad(wc2, (0, 0, 0, 14))
    assert torch.equal(wcol, torch.cat([w0g.reshape(-1), wc1.reshape(-1), w2p.reshape(-1)]).half())


@pytest.mark.parametrize("sfx", ["", "_bf16"])
def test_dir_term_with_the_frequency_encoder_folded_in(sfx):
    """lnh_lidar_dir_term_freq(dirs) == lnh_freq_encode_forward followed by lnh_lidar_dir_term, bit for bit."""
    from gpu_util import call
    N, deg = 1001, 12
    K = 3 + 6 * deg
    g = torch.Generator().manual_seed(5)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).cuda()
    W0 = (torch.randn(64, K + 15, generator=g) * 0.3).cuda()
    enc = torch.empty(N, K, device="cuda")
    call("lnh_freq_encode_forward", d, N, 3, deg, K, enc)
    e_a, c_a = torch.empty(N, K, device="cuda"), torch.empty(N, 64, device="cuda")
    call("lnh_lidar_dir_term" + sfx, enc, W0, K + 15, N, K, e_a, c_a)
    e_b, c_b = torch.empty(N, K, device="cuda"), torch.empty(N, 64, device="cuda")
    call("lnh_lidar_dir_term_freq" + sfx, d, deg, W0, K + 15, N, e_b, c_b)
    assert torch.equal(e_a, e_b) and torch.equal(c_a, c_b)


@pytest.mark.parametrize("sfx,dt", [("", torch.half), ("_bf16", torch.bfloat16)])
def test_step_prologue_equals_its_three_entry_points(sfx, dt):
    """lnh_lidar_step_prologue == lnh_lidar_pack_weights + lnh_lidar_dir_term_freq + lnh_lidar_coarse_sample_points, bit for
    bit (one launch instead of three), with and without the stratified noise, N not a multiple of 4."""
    from gpu_util import call
    N, T, Ttot, deg = 1001, 96, 128, 12
    K = 3 + 6 * deg
    g = torch.Generator().manual_seed(8)
    o = ((torch.rand(N, 3, generator=g) - 0.5) * 0.05).cuda()
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).cuda()
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32).cuda()
    near = float(np.float32(0.0107848535))
    far = float(np.float32(near) * np.float32(81.0))
    u = torch.rand(N * T, generator=g).cuda()
    ws0, ws1 = torch.randn(64, 32, generator=g).cuda(), torch.randn(16, 64, generator=g).cuda()
    wc0 = (torch.randn(64, 96, generator=g) * 0.3).cuda()[:, :K + 15]      # strided rows, like the tcnn-shaped flat parameter
    wc1, wc2 = torch.randn(64, 64, generator=g).cuda(), torch.randn(2, 64, generator=g).cuda()
    new = lambda *shape, dtype=torch.float32: torch.full(shape, float("nan"), dtype=dtype, device="cuda")
    for noise in (None, u):
        a = dict(wsig=new(64 * 32 + 16 * 64, dtype=dt), wcol=new(64 * 16 + 64 * 64 + 16 * 64, dtype=dt), e=new(N, K), c=new(N, 64),
                 z=new(N, T), x=new(N * Ttot, 3))
        b = {k: v.clone() for k, v in a.items()}
        call("lnh_lidar_pack_weights" + sfx, ws0, 32, ws1, 64, wc0, wc0.stride(0), K, wc1, 64, wc2, 64, a["wsig"], a["wcol"])
        call("lnh_lidar_dir_term_freq" + sfx, d, deg, wc0, wc0.stride(0), N, a["e"], a["c"])
        call("lnh_lidar_coarse_sample_points", noise, o, d, aabb, 1.0, N, T, Ttot, near, far, a["z"], a["x"])
        call("lnh_lidar_step_prologue" + sfx, ws0, 32, ws1, 64, wc0, wc0.stride(0), deg, wc1, 64, wc2, 64, b["wsig"], b["wcol"],
             noise, o, d, aabb, 1.0, N, T, Ttot, near, far, b["z"], b["x"], b["e"], b["c"])
        for k in a:
            fa, fb = torch.nan_to_num(a[k].float(), nan=-7.0), torch.nan_to_num(b[k].float(), nan=-7.0)
            assert torch.equal(fa, fb), k
        assert bool(torch.isnan(b["x"].view(N, Ttot, 3)[:, T:]).all()) and not bool(torch.isnan(b["x"].view(N, Ttot, 3)[:, :T]).any())


def test_fused_lidar_loss_matches_train_step_loss():
    from lidarnerf.nerf.train_step import fused_lidar_loss, lidar_loss
    torch.manual_seed(1)
    N = 4099
    depth = torch.rand(1, N, device="cuda", requires_grad=True)
    image = torch.rand(1, N, 2, device="cuda", requires_grad=True)
    gt = torch.rand(1, N, 3, device="cuda")
    gt[..., 0] = (gt[..., 0] > 0.3).float()
    with torch.no_grad():
        depth[0, :5] = gt[0, :5, 2]  # exact ties: sign(0) = 0
    want, _, _ = lidar_loss({"depth_lidar": depth, "image_lidar": image}, gt)
    (want * 3.0).backward()
    gd, gi = depth.grad.clone(), image.grad.clone()
    depth.grad = image.grad = None
    got = fused_lidar_loss({"depth_lidar": depth, "image_lidar": image}, gt)
    (got * 3.0).backward()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(depth.grad, gd, rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(image.grad, gi, rtol=1e-5, atol=1e-9)


def test_fused_patch_loss_matches_the_restatement():
    """lnh_lidar_loss_patch (per-ray LiDAR loss + the structural-gradient term of the patch epochs, one launch) against the
    CPU restatement of nerf/utils.py:712-746 + 760-876 (oracle/render_ref.py lidar_loss + patch_grad_loss, autograd for the
    gradients): 256 patches of 2 x 8 rays, ground truth smooth inside a patch (so the 0.01 m gate passes), dropped rays,
    exact ties.  The restatement itself is pinned by G8 — the reference's own Trainer.train_step run on fixed inputs
    (tests/test_oracle_golden.py::test_g8_train_step_loss_restatements) — and the kernel meets G8 directly in
    tests/test_g8_train_step_gpu.py; this test adds the shapes G8 does not hold (4096 rays, a rejected patch, |dx| = 0)."""
    from lidarnerf.nerf.train_step import fused_lidar_loss
    from oracle import render_ref
    g = torch.Generator().manual_seed(3)
    scale = 0.010784853507573345
    n = 256 * 16
    base = (torch.rand(256, 1, generator=g) * 0.6).expand(256, 16).reshape(n)
    gt_depth = base + 0.003 * scale * torch.randn(n, generator=g)
    gt_depth[16:32] = base[16:32] + 0.05 * scale * torch.randn(16, generator=g)       # a patch the gate rejects
    depth0 = gt_depth + 0.05 * scale * torch.randn(n, generator=g)
    depth0[40] = depth0[41]                                                              # |dx| = 0: sign(0) = 0
    gt = torch.stack([(torch.rand(n, generator=g) > 0.15).float(), torch.rand(n, generator=g), gt_depth], -1)
    image0 = torch.rand(n, 2, generator=g)
    dc, ic = depth0.clone().requires_grad_(True), image0.clone().requires_grad_(True)
    want = render_ref.lidar_loss(dc, ic, gt) + render_ref.patch_grad_loss(dc, gt, 2, 8, scale)
    (want * 5.0).backward()
    dg, ig = depth0.cuda().requires_grad_(True), image0.cuda().requires_grad_(True)
    got = fused_lidar_loss({"depth_lidar": dg[None], "image_lidar": ig[None]}, gt.cuda()[None], patch=(2, 8, scale, 100.0))
    (got * 5.0).backward()
    torch.testing.assert_close(got.cpu(), want.detach(), rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(dg.grad.cpu(), dc.grad, rtol=2e-5, atol=1e-6 * float(dc.grad.abs().max()))
    torch.testing.assert_close(ig.grad.cpu(), ic.grad, rtol=2e-5, atol=1e-9)
    assert float(render_ref.patch_grad_loss(dc.detach(), gt, 2, 8, scale)) > 1.0                 # the term is active in this test


def test_dir_term_backward():
    from lidarnerf import _hip
    torch.manual_seed(2)
    for N, K in ((4096, 75), (1001, 72), (3, 1)):
        S = torch.randn(N, 64, device="cuda")
        E = torch.randn(N, K, device="cuda")
        g = torch.zeros(64, K + 15, device="cuda")
        _hip.call("lnh_lidar_dir_term_backward", S.data_ptr(), E.data_ptr(), N, K, None, g.data_ptr(), K + 15, *_hip.wgrad_ws("cuda"))
        want = S.double().t() @ E.double()
        torch.testing.assert_close(g[:, :K].double(), want, rtol=1e-4, atol=1e-3)
        assert float(g[:, K:].abs().max()) == 0.0
        # with the packed geo-feature block: its columns 1..15 land behind the K direction columns, in the same launch
        g2 = torch.zeros(64, K + 15, device="cuda")
        w0g = torch.randn(64, 16, device="cuda")
        _hip.call("lnh_lidar_dir_term_backward", S.data_ptr(), E.data_ptr(), N, K, w0g.data_ptr(), g2.data_ptr(), K + 15,
                  *_hip.wgrad_ws("cuda"))
        torch.testing.assert_close(g2[:, :K].double(), want, rtol=1e-4, atol=1e-3)
        assert torch.equal(g2[:, K:], w0g[:, 1:])
