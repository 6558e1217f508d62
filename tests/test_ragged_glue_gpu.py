"""The element-wise stages of the occupancy-grid render chain (csrc/lidar_ragged.hip, lnh_ragged_*), each against the tensor
expression of the reference's network code it replaces (network.py:162-237 on flat sample lists; activation.py:17-19;
gridencoder/grid.py:213), fp16 and bf16 builds, called through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import encoders_ref

pytestmark = pytest.mark.gpu
M = 3001  # not a multiple of anything


def _dt(sfx):
    return torch.bfloat16 if sfx else torch.float16


def test_ragged_points():
    from gpu_util import call
    xyz = (torch.rand(M, 3, device="cuda") * 2 - 1) * 2.0
    out = torch.empty_like(xyz)
    call("lnh_ragged_points", xyz, 2.0, M, out)
    assert torch.equal(out, (xyz + 2.0) / (2 * 2.0))


@pytest.mark.parametrize("sfx", ["", "_bf16"])
def test_ragged_pack_weights(sfx):
    from gpu_util import call
    dt = _dt(sfx)
    g = torch.Generator().manual_seed(1)
    ws0, ws1 = torch.randn(64, 32, generator=g).cuda(), torch.randn(16, 64, generator=g).cuda()
    big = torch.randn(64, 100, generator=g).cuda()
    wc0 = big[:, 5:95]                                         # a strided view: leading dimension 100, 90 used columns
    wc1, wc2 = torch.randn(64, 64, generator=g).cuda(), torch.randn(2, 64, generator=g).cuda()
    wsig = torch.empty(64 * 32 + 16 * 64, dtype=dt, device="cuda")
    wcol = torch.empty(64 * 96 + 64 * 64 + 16 * 64, dtype=dt, device="cuda")
    call("lnh_ragged_pack_weights" + sfx, ws0, 32, ws1, 64, wc0, 100, 90, wc1, 64, wc2, 64, wsig, wcol)
    assert torch.equal(wsig, torch.cat([ws0.reshape(-1), ws1.reshape(-1)]).to(dt))
    w0 = torch.zeros(64, 96, device="cuda")
    w0[:, :90] = wc0
    w2 = torch.zeros(16, 64, device="cuda")
    w2[:2] = wc2
    assert torch.equal(wcol, torch.cat([w0.reshape(-1), wc1.reshape(-1), w2.reshape(-1)]).to(dt))


@pytest.mark.parametrize("sfx", ["", "_bf16"])
def test_ragged_color_input(sfx):
    from gpu_util import call
    dt = _dt(sfx)
    g = torch.Generator().manual_seed(2)
    d = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1)
    h16 = torch.randn(M, 16, generator=g).to(dt)
    cin = torch.full((M, 96), 7.0, dtype=dt, device="cuda")
    call("lnh_ragged_color_input" + sfx, d.cuda(), h16.cuda(), M, 12, cin)
    got = cin.float().cpu()
    want = torch.from_numpy(encoders_ref.freq_forward(d.numpy(), 12)).to(dt).float()   # freqencoder.cu:34-63 layout
    # frequency features: the kernel's sinf against NumPy's, then one rounding to the element type
    np.testing.assert_allclose(got[:, :75].numpy(), want.numpy(), rtol=0, atol=2e-3 if sfx == "" else 1.6e-2)
    assert torch.equal(got[:, 75:90], h16[:, 1:].float()) and float(got[:, 90:].abs().max()) == 0.0
    # bit-identical to the product's own frequency encoder followed by the cast
    enc = torch.empty(M, 75, device="cuda")
    call("lnh_freq_encode_forward", d.cuda(), M, 3, 12, 75, enc)
    assert torch.equal(cin[:, :75], enc.to(dt))


@pytest.mark.parametrize("sfx", ["", "_bf16"])
def test_ragged_color_input_ray_by_ray(sfx):
    """lnh_ragged_color_input_rays on a marcher-shaped layout (rays own disjoint runs of rows in shuffled order, one ray
    overflows the buffer, one has no sample, the tail is unused): bit-identical to the per-sample kernel on the rows rays
    own; exact zeros on every other row (whatever the buffer held before)."""
    from gpu_util import call
    dt = _dt(sfx)
    g = torch.Generator().manual_seed(4)
    N = 37
    counts = torch.randint(1, 140, (N,), generator=g)
    counts[5] = 0
    order = torch.randperm(N, generator=g)                     # the marcher hands out offsets in arrival order
    offs = torch.zeros(N, dtype=torch.long)
    acc = 0
    for n in order.tolist():
        offs[n] = acc
        acc += int(counts[n])
    last = int(order[-1])                                      # the ray that arrived last does not fit
    Mbuf = acc - int(counts[last]) + int(counts[last]) // 2 + 200 if counts[last] > 1 else acc + 200
    fits = (offs + counts <= Mbuf) & (counts > 0)
    rays = torch.stack([torch.arange(N), offs, counts], -1).int()
    dray = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    dirs, deltas = torch.zeros(Mbuf, 3), torch.zeros(Mbuf, 2)
    owned = torch.zeros(Mbuf, dtype=torch.bool)
    for n in range(N):
        if fits[n]:
            a, b = int(offs[n]), int(offs[n] + counts[n])
            dirs[a:b], deltas[a:b, 0], owned[a:b] = dray[n], 0.003, True
    assert 0 < int(owned.sum()) < Mbuf and not bool(fits.all())
    h16 = torch.randn(Mbuf, 16, generator=g).to(dt)
    want = torch.empty((Mbuf, 96), dtype=dt, device="cuda")
    call("lnh_ragged_color_input" + sfx, dirs.cuda(), h16.cuda(), Mbuf, 12, want)
    got = torch.full((Mbuf, 96), float("nan"), dtype=dt, device="cuda")
    call("lnh_ragged_color_input_rays" + sfx, dirs.cuda(), h16.cuda(), rays.cuda(), deltas.cuda(), N, Mbuf, 12, got)
    assert torch.equal(got[owned.cuda()], want[owned.cuda()])
    assert float(got[~owned.cuda()].float().abs().max()) == 0.0


@pytest.mark.parametrize("sfx", ["", "_bf16"])
def test_ragged_color_output_and_backward(sfx):
    from gpu_util import call
    dt = _dt(sfx)
    g = torch.Generator().manual_seed(3)
    y = (torch.randn(M, 16, generator=g) * 3).to(dt).cuda()
    rgb = torch.empty(M, 2, device="cuda")
    call("lnh_ragged_color_output" + sfx, y, M, rgb)
    torch.testing.assert_close(rgb, torch.sigmoid(y[:, :2].float()), rtol=2e-6, atol=1e-7)
    grad = torch.randn(M, 2, generator=g).cuda()
    gy = torch.full((M, 16), 9.0, dtype=dt, device="cuda")
    call("lnh_ragged_color_output_backward" + sfx, grad, rgb, M, gy)
    want = torch.zeros(M, 16, device="cuda")
    want[:, :2] = grad * rgb * (1 - rgb)
    assert torch.equal(gy, want.to(dt))


@pytest.mark.parametrize("sfx", ["", "_bf16"])
def test_ragged_grad_rows(sfx):
    from gpu_util import call
    from lidarnerf.activation import trunc_exp
    dt = _dt(sfx)
    g = torch.Generator().manual_seed(4)
    h16 = (torch.randn(M, 16, generator=g) * 8).to(dt).cuda()     # |h0| beyond 15 on some rows: the clamp of trunc_exp's backward
    h16[:4, 0] = torch.tensor([20.0, -20.0, 15.0, -15.0]).to(dt)
    gs = torch.randn(M, generator=g).cuda()
    gx = torch.randn(M, 96, generator=g).to(dt).cuda()
    out = torch.empty(M, 16, dtype=dt, device="cuda")
    call("lnh_ragged_grad_rows" + sfx, gs, 1.5, h16, gx, 12, M, out)
    # column 0 through the product's own trunc_exp autograd (activation.py:6-20): d sigma / d h0 = exp(clamp(h0, -15, 15))
    h0 = h16[:, 0].float().clone().requires_grad_(True)
    (trunc_exp(h0) * 1.5 * gs).sum().backward()
    torch.testing.assert_close(out[:, 0].float(), h0.grad.to(dt).float(), rtol=2e-3 if sfx == "" else 1.6e-2, atol=0)
    assert torch.equal(out[:, 1:], gx[:, 75:90])


@pytest.mark.parametrize("sfx", ["", "_bf16"])
def test_ragged_colour_head_per_ray_equals_the_per_sample_chain(sfx):
    """lnh_ragged_color_forward / _backward (round 5: the colour head ray by ray — direction term once per ray, the sample's
    sigma-net row as the 16-wide input) against the per-sample chain they replace in the occupancy-grid render step:
    lnh_ragged_color_input_rays -> lnh_mlp_forward (96 -> 64 -> 64 -> 16) -> lnh_ragged_color_output, and backward
    lnh_ragged_color_output_backward -> lnh_mlp_backward -> lnh_ragged_grad_rows.  Ray table with an empty ray, a dropped ray
    (samples beyond M), rays in arrival order != ray index, rows nobody owns."""
    from gpu_util import call, wgrad
    dt = _dt(sfx)
    g = torch.Generator().manual_seed(12)
    N, deg = 37, 12
    kd = 3 + 6 * deg
    counts = torch.randint(1, 150, (N,), generator=g)
    counts[5] = 0
    order = torch.randperm(N, generator=g)                                 # table slot -> ray index
    offs, tot = [], 0
    for r in range(N):
        offs.append(tot)
        tot += int(counts[order[r]])
    Mr = tot + 23                                                          # 23 rows nobody owns
    rays = torch.stack([order.int(), torch.tensor(offs, dtype=torch.int32), counts[order].int()], dim=1).contiguous()
    rays[7, 1], rays[7, 2] = Mr - 3, 10                                    # a dropped ray: its samples do not fit
    dropped_rows = torch.arange(offs[7], offs[7] + int(counts[order[7]]))  # (its original rows: nobody's now)
    owned = torch.zeros(Mr, dtype=torch.bool)
    for r in range(N):
        if r != 7:
            owned[offs[r]:offs[r] + int(counts[order[r]])] = True
    rays_d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    dirs = torch.zeros(Mr, 3)
    deltas = torch.zeros(Mr, 2)
    for r in range(N):
        if r != 7:
            sl = slice(offs[r], offs[r] + int(counts[order[r]]))
            dirs[sl] = rays_d[order[r]]
            deltas[sl] = 0.01
    h16 = (torch.randn(Mr, 16, generator=g) * 0.5).to(dt)
    ws0, ws1 = torch.randn(64, 32, generator=g) * 0.2, torch.randn(16, 64, generator=g) * 0.2
    wc0, wc1, wc2 = torch.randn(64, kd + 15, generator=g) * 0.15, torch.randn(64, 64, generator=g) * 0.15, torch.randn(2, 64, generator=g) * 0.3
    C = lambda t: t.cuda().contiguous()
    rays_c, dirs_c, deltas_c, h16_c, rd_c = C(rays), C(dirs), C(deltas), C(h16), C(rays_d)
    ws0, ws1, wc0, wc1, wc2 = C(ws0), C(ws1), C(wc0), C(wc1), C(wc2)
    # ---- per-sample chain (rounds 2-4)
    wsig = torch.empty(64 * 32 + 16 * 64, dtype=dt, device="cuda")
    wcol96 = torch.empty(64 * 96 + 64 * 64 + 16 * 64, dtype=dt, device="cuda")
    call("lnh_ragged_pack_weights" + sfx, ws0, 32, ws1, 64, wc0, kd + 15, kd + 15, wc1, 64, wc2, 64, wsig, wcol96)
    cin = torch.empty(Mr, 96, dtype=dt, device="cuda")
    call("lnh_ragged_color_input_rays" + sfx, dirs_c, h16_c, rays_c, deltas_c, N, Mr, deg, cin)
    y = torch.empty(Mr, 16, dtype=dt, device="cuda")
    call("lnh_mlp_forward" + sfx, cin, wcol96, Mr, 96, 16, 64, 1, 0, 6, None, y)
    rgb_a = torch.empty(Mr, 2, device="cuda")
    call("lnh_ragged_color_output" + sfx, y, Mr, rgb_a)
    gf = torch.randn(Mr, 2, generator=g) * 1e-2
    gf[~owned] = 0
    gs = torch.randn(Mr, generator=g) * 1e-2
    gs[~owned] = 0
    gf_c, gs_c = C(gf), C(gs)
    gy = torch.empty(Mr, 16, dtype=dt, device="cuda")
    call("lnh_ragged_color_output_backward" + sfx, gf_c, rgb_a, Mr, gy)
    gx = torch.empty(Mr, 96, dtype=dt, device="cuda")
    gw96 = torch.zeros(wcol96.numel(), device="cuda")
    call("lnh_mlp_backward" + sfx, gy, cin, wcol96, Mr, 96, 16, 64, 1, 0, 6, gx, gw96, *wgrad())
    gh_a = torch.empty(Mr, 16, dtype=dt, device="cuda")
    call("lnh_ragged_grad_rows" + sfx, gs_c, 1.7, h16_c, gx, deg, Mr, gh_a)
    # ---- ray by ray (round 5)
    wcol16 = torch.empty(64 * 16 + 64 * 64 + 16 * 64, dtype=dt, device="cuda")
    call("lnh_lidar_pack_weights" + sfx, ws0, 32, ws1, 64, wc0, kd + 15, kd, wc1, 64, wc2, 64, wsig, wcol16)
    enc16, cdir = torch.empty(N, kd, device="cuda"), torch.empty(N, 64, device="cuda")
    call("lnh_lidar_dir_term_freq" + sfx, rd_c, deg, wc0, kd + 15, N, enc16, cdir)
    rgb_b = torch.zeros(Mr, 2, device="cuda")
    call("lnh_ragged_color_forward" + sfx, h16_c, rays_c, cdir, wcol16, N, Mr, rgb_b)
    gh_b = torch.zeros(Mr, 16, dtype=dt, device="cuda")
    gw16 = torch.zeros(wcol16.numel(), device="cuda")
    S = torch.zeros(N, 64, device="cuda")
    call("lnh_ragged_color_backward" + sfx, gf_c, gs_c, 1.7, h16_c, rays_c, cdir, wcol16, N, Mr, gh_b, gw16, S, *wgrad())
    g_wc0 = torch.zeros(64, kd + 15, device="cuda")
    call("lnh_lidar_dir_term_backward", S, enc16, N, kd, gw16[:1024], g_wc0, kd + 15, *wgrad())
    torch.cuda.synchronize()
    ow = owned.cuda()
    tol = 4e-3 if sfx == "" else 3e-2                                     # (two different summation orders of the first layer)
    torch.testing.assert_close(rgb_b[ow], rgb_a[ow], rtol=tol, atol=tol)
    assert float(rgb_b[~ow].abs().max()) == 0.0 and float(gh_b[~ow].float().abs().max()) == 0.0
    assert float(rgb_b[dropped_rows.cuda()].abs().max()) == 0.0            # the dropped ray was not evaluated
    ga, gb = gh_a[ow].float(), gh_b[ow].float()
    assert float((ga - gb).norm()) <= 2.5 * tol * float(ga.norm()), (float((ga - gb).norm()), float(ga.norm()))
    want_w0 = gw96[:64 * 96].view(64, 96)[:, :kd + 15]
    for name, a_, b_ in (("wc0", want_w0, g_wc0), ("wc1", gw96[64 * 96:64 * 96 + 4096], gw16[1024:1024 + 4096]),
                         ("wc2", gw96[64 * 96 + 4096:].view(16, 64)[:2], gw16[1024 + 4096:].view(16, 64)[:2])):
        assert float((a_ - b_.view_as(a_)).norm()) <= 2.5 * tol * float(a_.norm()), (name, float((a_ - b_.view_as(a_)).norm()), float(a_.norm()))
    assert float(S[5].abs().max()) == 0.0 and float(S[order[7]].abs().max()) == 0.0   # the empty ray (index 5), the dropped one
