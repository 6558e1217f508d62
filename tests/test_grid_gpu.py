"""HIP hash-grid encoder vs the CPU oracle, through the C ABI (lnh_grid_*)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle, grid_ref

pytestmark = pytest.mark.gpu

H, L, CH = 16, 16, 2
PLS = grid_ref.per_level_scale(32768, H, L)
S = float(np.log2(PLS))
OFF = grid_ref.make_offsets(3, L, PLS, H, 19)


def _points(B, seed, D=3):
    r = np.random.default_rng(seed)
    x = r.random((B, D), dtype=np.float32)
    x[0] = 0
    x[1] = 1
    x[2, 0] = 1.0000001
    x[3, 1] = -1e-7
    x[4] = 0.5
    return x


def _ray_points(n_rays, T, seed):
    """Consecutive samples along rays (exercises the wave run-merge in backward)."""
    r = np.random.default_rng(seed)
    o = r.random((n_rays, 1, 3), dtype=np.float32) * 0.2 + 0.4
    d = r.standard_normal((n_rays, 1, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = np.linspace(0.005, 0.45, T, dtype=np.float32)[None, :, None]
    return np.clip(o + d * t, 0, 1).reshape(-1, 3).astype(np.float32)


@pytest.mark.parametrize("gridtype,align,D", [(0, False, 3), (1, False, 3), (0, True, 3), (0, False, 2), (0, False, 4)])
def test_corner_indices_bit_exact(gridtype, align, D):
    from gpu_util import call, dev, host
    x = _points(4099, 1, D)
    off = grid_ref.make_offsets(D, L, PLS, H, 19, align_corners=align)
    want = c_oracle.grid_indices(x, off, CH, S, H, gridtype, align)
    xd = dev(x)
    out = torch.empty((L, x.shape[0], 1 << D), dtype=torch.int32, device="cuda")
    offh = torch.from_numpy(off)
    call("lnh_grid_corner_indices", xd, offh, out, x.shape[0], D, CH, L, S, H, gridtype, int(align))
    np.testing.assert_array_equal(host(out).view(np.uint32), want)


def test_corner_indices_full_size_properties():
    """BASELINE size (4096 rays x 832 samples): every index in range, dense levels match the closed form."""
    from gpu_util import call, dev, host
    B = 4096 * 832
    x = torch.rand((B, 3), device="cuda")
    out = torch.empty((L, B, 8), dtype=torch.int32, device="cuda")
    call("lnh_grid_corner_indices", x, torch.from_numpy(OFF), out, B, 3, CH, L, S, H, 0, 0)
    rows = np.diff(OFF)
    for l in range(L):
        o = out[l].to(torch.int64) & 0xFFFFFFFF
        assert int(o.max()) < rows[l] * CH and int(o.min()) >= 0
    # level 0 is dense: row = x + y*R + z*R^2 with R = 17, corner 0
    sc, res = c_oracle.grid_level(0, S, H)
    pg = torch.floor(x * float(sc) + 0.5).to(torch.int64)
    want = (pg[:, 0] + pg[:, 1] * (res + 1) + pg[:, 2] * (res + 1) ** 2) * CH
    # x*scale+0.5 is one fma in the kernel; the torch expression above rounds twice -> allow the rare floor flip
    mism = (out[0, :, 0].to(torch.int64) != want).float().mean().item()
    assert mism < 1e-4
    # sample check against the oracle
    sel = torch.randint(0, B, (2048,), device="cuda")
    want_s = c_oracle.grid_indices(host(x[sel]), OFF, CH, S, H)
    np.testing.assert_array_equal(host(out[:, sel]).view(np.uint32), want_s)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
@pytest.mark.parametrize("C", [1, 2, 4, 8])
def test_forward(dt, C):
    from gpu_util import call, dev, host
    if dt == torch.float16 and C == 1:
        pytest.skip("reference forces fp32 for odd C (grid.py:54-57)")
    x = _points(3001, 2)
    r = np.random.default_rng(0)
    nd = np.float32 if dt == torch.float32 else np.float16
    emb = (r.random((int(OFF[-1]), C), dtype=np.float32) * 2 - 1).astype(nd)
    want, _ = c_oracle.grid_forward(x, emb, OFF, S, H)
    out = torch.empty((L, x.shape[0], C), dtype=dt, device="cuda")
    call("lnh_grid_encode_forward", dev(x), dev(emb), torch.from_numpy(OFF), out, x.shape[0], 3, C, L, S, H, None, 0, 0,
         0, 0 if dt == torch.float32 else 1)
    got = host(out)
    if dt == torch.float32:
        # same fma chain in the same corner order: bit exact
        np.testing.assert_array_equal(got, want)
    else:
        np.testing.assert_array_equal(got.view(np.uint16), want.view(np.uint16))


@pytest.mark.parametrize("interp,gridtype,align", [(1, 0, False), (0, 1, False), (0, 0, True)])
def test_forward_variants_and_dydx(interp, gridtype, align):
    from gpu_util import call, dev, host
    x = _points(1500, 5)
    off = grid_ref.make_offsets(3, L, PLS, H, 19, align_corners=align)
    emb = np.random.default_rng(3).standard_normal((int(off[-1]), CH)).astype(np.float32)
    want, wdy = c_oracle.grid_forward(x, emb, off, S, H, gridtype, align, interp, calc_dy_dx=True)
    out = torch.empty((L, x.shape[0], CH), device="cuda")
    dy = torch.empty((x.shape[0], L, 3, CH), device="cuda")
    call("lnh_grid_encode_forward", dev(x), dev(emb), torch.from_numpy(off), out, x.shape[0], 3, CH, L, S, H, dy,
         gridtype, int(align), interp, 0)
    np.testing.assert_array_equal(host(out), want)
    np.testing.assert_allclose(host(dy), wdy, rtol=1e-5, atol=1e-4)
    # input backward
    g = np.random.default_rng(4).standard_normal((L, x.shape[0], CH)).astype(np.float32)
    ge = torch.zeros((int(off[-1]), CH), device="cuda")
    gi = torch.zeros((x.shape[0], 3), device="cuda")
    call("lnh_grid_encode_backward", dev(g), dev(x), None, torch.from_numpy(off), ge, x.shape[0], 3, CH, L, S, H, dy, gi,
         gridtype, int(align), interp, 0)
    np.testing.assert_allclose(host(gi), c_oracle.grid_input_backward(g, wdy), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
@pytest.mark.parametrize("kind", ["random", "rays"])
def test_backward(dt, kind):
    from gpu_util import call, dev, host
    x = _points(5000, 7) if kind == "random" else _ray_points(24, 256, 7)
    B = x.shape[0]
    nd = np.float32 if dt == torch.float32 else np.float16
    g = (np.random.default_rng(8).standard_normal((L, B, CH)) * 0.1).astype(nd)
    rows = int(OFF[-1])
    want = c_oracle.grid_backward(g, x, OFF, rows, S, H)
    ge = torch.zeros((rows, CH), dtype=dt, device="cuda")
    call("lnh_grid_encode_backward", dev(g), dev(x), None, torch.from_numpy(OFF), ge, B, 3, CH, L, S, H, None, None, 0, 0,
         0, 0 if dt == torch.float32 else 1)
    got = host(ge).astype(np.float64)
    if dt == torch.float32:
        # atomics: order-dependent fp32 sums vs the order-free float64 oracle
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
    else:
        # fp16 accumulation in the table (packed fp16 atomics): error grows with the number of addends per cell
        np.testing.assert_allclose(got, want, rtol=2e-2, atol=2e-2)
    # untouched cells stay exactly zero
    assert np.all(got[want == 0] == 0)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
@pytest.mark.parametrize("kind", ["random", "rays", "same_cell", "tiny"])
def test_backward_bucketed(dt, kind):
    """lnh_grid_encode_backward_ws (no HBM atomics) must give the same table as the oracle's order-free sum."""
    from gpu_util import call, dev, host
    from lidarnerf import _hip
    if kind == "random":
        x = _points(5000, 7)
    elif kind == "rays":
        x = _ray_points(24, 256, 7)
    elif kind == "tiny":  # fewer points than a wave, one of them outside the grid
        x = np.random.default_rng(9).random((3, 3), dtype=np.float32)
        x[1] = [1.5, 0.2, 0.3]
    else:  # adversarial: every point in one cell -> one bucket overflows its pool, excess goes through atomics
        x = (np.random.default_rng(1).random((20000, 3), dtype=np.float32) * 1e-6 + 0.3).astype(np.float32)
        x[::2] += np.float32(0.11)  # break the runs so the wave merge cannot collapse everything
    B = x.shape[0]
    nd = np.float32 if dt == torch.float32 else np.float16
    g = (np.random.default_rng(8).standard_normal((L, B, CH)) * 0.1).astype(nd)
    rows = int(OFF[-1])
    want = c_oracle.grid_backward(g, x, OFF, rows, S, H)
    code = 0 if dt == torch.float32 else 1
    offh = torch.from_numpy(OFF)
    need = _hip.lib().lnh_grid_backward_workspace_size(offh.data_ptr(), B, 3, CH, L, S, H, 0, 0, code)
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    ge = torch.zeros((rows, CH), dtype=dt, device="cuda")
    call("lnh_grid_encode_backward_ws", dev(g), dev(x), offh, ge, B, 3, CH, L, S, H, 0, 0, 0, code, ws, need)
    got = host(ge).astype(np.float64)
    if dt == torch.float32:
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
    elif kind == "same_cell":
        # pool overflow -> the excess takes the reference's route (packed fp16 atomics into the table): thousands of
        # fp16 additions per row, each rounding at the running sum's ulp (2^-10 relative)
        np.testing.assert_allclose(got, want, rtol=0.1, atol=0.05 * np.abs(want).max())
    else:  # fp32 LDS accumulation, ONE rounding to fp16 per touched row
        np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-3 * max(1.0, np.abs(want).max() / 10))
    assert np.all(got[want == 0] == 0)
    # too-small workspace is an error, not silent corruption
    with pytest.raises(RuntimeError, match="workspace too small"):
        call("lnh_grid_encode_backward_ws", dev(g), dev(x), offh, ge, B, 3, CH, L, S, H, 0, 0, 0, code, ws, 1024)


@pytest.mark.parametrize("dt", [torch.float32, torch.float16])
def test_backward_bucketed_level_windows(dt):
    """lnh_grid_encode_backward_ws_levels over consecutive windows == the one-shot call, bit for bit (the bucketed sum
    is order-independent), and a window leaves the rows of the other levels alone."""
    from gpu_util import call, dev
    from lidarnerf import _hip
    x = _ray_points(40, 256, 3)
    B = x.shape[0]
    nd = np.float32 if dt == torch.float32 else np.float16
    g = (np.random.default_rng(4).standard_normal((L, B, CH)) * 0.1).astype(nd)
    rows = int(OFF[-1])
    code = 0 if dt == torch.float32 else 1
    offh = torch.from_numpy(OFF)
    need = _hip.lib().lnh_grid_backward_workspace_size(offh.data_ptr(), B, 3, CH, L, S, H, 0, 0, code)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    full = torch.zeros((rows, CH), dtype=dt, device="cuda")
    call("lnh_grid_encode_backward_ws", dev(g), dev(x), offh, full, B, 3, CH, L, S, H, 0, 0, 0, code, ws, need)
    part = torch.zeros((rows, CH), dtype=dt, device="cuda")
    gd, xd = dev(g), dev(x)
    windows = [(0, 7), (7, 10), (10, 13), (13, L)] if L == 16 else [(0, L // 2), (L // 2, L)]
    for k, (l0, l1) in enumerate(windows):
        call("lnh_grid_encode_backward_ws_levels", gd, xd, offh, part, B, 3, CH, L, S, H, 0, 0, 0, code, ws, need, l0, l1)
        done = int(OFF[l1])
        assert torch.equal(part[:done], full[:done])
        assert float(part[done:].abs().max()) == 0.0 if done < rows else True
    assert torch.equal(part, full)
    call("lnh_grid_encode_backward_ws_levels", gd, xd, offh, part, B, 3, CH, L, S, H, 0, 0, 0, code, ws, need, 5, 5)  # empty
    assert torch.equal(part, full)
    with pytest.raises(RuntimeError, match="level_begin"):
        call("lnh_grid_encode_backward_ws_levels", gd, xd, offh, part, B, 3, CH, L, S, H, 0, 0, 0, code, ws, need, 3, L + 1)


def test_backward_full_size_checksum():
    """Full BASELINE size: sum of the gradient table == sum of upstream grads (weights of a cell sum to 1)."""
    from gpu_util import call
    n_rays, T = 4096, 832
    x = torch.from_numpy(_ray_points(64, T, 3)).cuda().repeat(n_rays // 64, 1)
    x = (x + torch.rand_like(x) * 1e-3).clamp(0, 1)
    B = x.shape[0]
    g = torch.randn((L, B, CH), device="cuda") * 0.01
    rows = int(OFF[-1])
    ge = torch.zeros((rows, CH), device="cuda")
    call("lnh_grid_encode_backward", g, x, None, torch.from_numpy(OFF), ge, B, 3, CH, L, S, H, None, None, 0, 0, 0, 0)
    torch.cuda.synchronize()
    offs = torch.from_numpy(OFF.astype(np.int64))
    for l in range(L):
        s_tab = ge[offs[l]:offs[l + 1]].double().sum(0)
        s_g = g[l].double().sum(0)
        assert torch.allclose(s_tab, s_g, rtol=1e-3, atol=1e-2), (l, s_tab, s_g)


def test_forward_full_size_properties():
    """Full BASELINE size (4096 rays x 832 samples), fp16 table, size-independent properties of the interpolation:
    (1) a table that is constant per level reproduces that constant (the 8 weights of a cell sum to 1) wherever the
    point is inside the grid, and 0 outside; (2) the bucketed backward of the same batch is the adjoint of the forward:
    <forward(table), g> == <table, backward(g)>; (3) the row-mapped variant writes the same values into its slots."""
    from gpu_util import call
    from lidarnerf import _hip
    n_rays, T = 4096, 832
    x = torch.from_numpy(_ray_points(64, T, 5)).cuda().repeat(n_rays // 64, 1)
    x = (x + torch.rand_like(x) * 1e-3)
    x[::1000] = 1.5                                   # some points outside the grid
    B = x.shape[0]
    rows = int(OFF[-1])
    offh = torch.from_numpy(OFF)
    offs = OFF.astype(np.int64)
    consts = torch.linspace(0.25, 4.0, L)
    tab = torch.empty((rows, CH), dtype=torch.half, device="cuda")
    for l in range(L):
        tab[offs[l]:offs[l + 1]] = consts[l]
    out = torch.empty((L, B, CH), dtype=torch.half, device="cuda")
    call("lnh_grid_encode_forward", x, tab, offh, out, B, 3, CH, L, S, H, None, 0, 0, 0, 1)
    inside = ((x >= 0) & (x <= 1)).all(1)
    for l in range(L):
        v = out[l].float()
        assert float((v[inside] - consts[l]).abs().max()) <= 4e-3 * float(consts[l])   # 8 fp16 roundings
        assert float(v[~inside].abs().max()) == 0.0
    # (2) adjointness on a random POSITIVE table / gradient (so that the two inner products are large sums without
    #     cancellation and a relative tolerance means something; fp32 accumulation on both sides, fp16 storage)
    tab = (torch.rand((rows, CH), device="cuda") * 0.5 + 0.1).half()
    call("lnh_grid_encode_forward", x, tab, offh, out, B, 3, CH, L, S, H, None, 0, 0, 0, 1)
    g = (torch.rand((L, B, CH), device="cuda") * 1e-3 + 1e-4).half()  # row sums of ~50 of these stay far below 65504
    need = _hip.lib().lnh_grid_backward_workspace_size(offh.data_ptr(), B, 3, CH, L, S, H, 0, 0, 1)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    gt = torch.zeros((rows, CH), dtype=torch.half, device="cuda")
    call("lnh_grid_encode_backward_ws", g, x, offh, gt, B, 3, CH, L, S, H, 0, 0, 0, 1, ws, need)
    lhs = float((out.double() * g.double()).sum())
    rhs = float((tab.double() * gt.double()).sum())
    assert lhs > 1e3 and abs(lhs - rhs) <= 2e-3 * abs(lhs), (lhs, rhs)
    # (3) row map: T_cur = 768 of T_tot = 832 slots per ray, offset 0 -> rows r*832 + j
    Tc = 768
    Bc = n_rays * Tc
    xs = torch.zeros((B, 3), device="cuda")
    ray = torch.arange(Bc, device="cuda") // Tc
    dst = ray * T + torch.arange(Bc, device="cuda") % Tc
    xs[dst] = x[:Bc]
    mapped = torch.zeros((L, B, CH), dtype=torch.half, device="cuda")
    call("lnh_grid_encode_forward_mapped", xs, tab, offh, mapped, Bc, Tc, T, 0, B, CH, L, S, H, 1)
    plain = torch.empty((L, Bc, CH), dtype=torch.half, device="cuda")
    call("lnh_grid_encode_forward", x[:Bc].contiguous(), tab, offh, plain, Bc, 3, CH, L, S, H, None, 0, 0, 0, 1)
    assert torch.equal(mapped[:, dst], plain)
    untouched = torch.ones(B, dtype=torch.bool, device="cuda")
    untouched[dst] = False
    assert float(mapped[:, untouched].abs().max()) == 0.0


def test_error_paths():
    from lidarnerf import _hip
    x = torch.rand((8, 3), device="cuda")
    out = torch.empty((L, 8, 3), device="cuda")
    emb = torch.zeros((int(OFF[-1]), 3), device="cuda")
    with pytest.raises(RuntimeError, match="C must be 1, 2, 4, or 8"):
        _hip.call("lnh_grid_encode_forward", x.data_ptr(), emb.data_ptr(), torch.from_numpy(OFF).data_ptr(),
                  out.data_ptr(), 8, 3, 3, L, S, H, None, 0, 0, 0, 0)
    with pytest.raises(RuntimeError, match="D must be"):
        _hip.call("lnh_grid_encode_forward", x.data_ptr(), emb.data_ptr(), torch.from_numpy(OFF).data_ptr(),
                  out.data_ptr(), 8, 7, 2, L, S, H, None, 0, 0, 0, 0)
    # empty batch is a no-op
    _hip.call("lnh_grid_encode_forward", x.data_ptr(), emb.data_ptr(), torch.from_numpy(OFF).data_ptr(),
              out.data_ptr(), 0, 3, 2, L, S, H, None, 0, 0, 0, 0)
