"""Self-consistency and size-independent property tests (no oracle on the other side).  The file name sorts after every
oracle-parity file on purpose: under `pytest -x` a failure here cannot hide a parity test."""
import numpy as np
import pytest
import torch

from oracle import grid_ref

pytestmark = pytest.mark.gpu

H, L, CH = 16, 16, 2
PLS = grid_ref.per_level_scale(32768, H, L)
S = float(np.log2(PLS))
OFF = grid_ref.make_offsets(3, L, PLS, H, 19)


def _ray_points(n_rays, T, seed):
    """Consecutive samples along rays (exercises the wave run-merge in backward)."""
    r = np.random.default_rng(seed)
    o = r.random((n_rays, 1, 3), dtype=np.float32) * 0.2 + 0.4
    d = r.standard_normal((n_rays, 1, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = np.linspace(0.005, 0.45, T, dtype=np.float32)[None, :, None]
    return np.clip(o + d * t, 0, 1).reshape(-1, 3).astype(np.float32)


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
    # begin + finish per window (what the data-parallel backward calls: ONE scatter pass, the reduce pass per window)
    split = torch.zeros_like(full)
    ws.random_(0, 255)
    call("lnh_grid_encode_backward_ws_begin", gd, xd, offh, split, B, 3, CH, L, S, H, 0, 0, 0, code, ws, need)
    for l0, l1 in ((0, 7), (7, 10), (10, 13), (13, L)):
        before = split.clone()
        call("lnh_grid_encode_backward_ws_finish", gd, xd, offh, split, B, 3, CH, L, S, H, 0, 0, 0, code, ws, need, l0, l1)
        changed = (split != before).any(1).nonzero()
        assert changed.numel() and int(changed.min()) >= int(OFF[l0]) and int(changed.max()) < int(OFF[l1])
    assert torch.equal(split, full)
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


def test_backward_bucketed_chunked_matches_atomic_path():
    """5 M points (> 4 M: the bucketed backward walks the batch in two chunks) — fp32 table gradient of the bucketed path
    vs the independent atomic kernel (lnh_grid_encode_backward), plus the per-level checksum sum(table) == sum(grad)
    (the 8 weights of a cell sum to 1): a point dropped or counted twice at the chunk boundary would show in both."""
    from gpu_util import call
    from lidarnerf import _hip
    n_rays, T = 6016, 832
    x = torch.from_numpy(_ray_points(64, T, 11)).cuda().repeat(n_rays // 64, 1)
    x = (x + torch.rand_like(x) * 1e-3).clamp(0, 1).contiguous()
    B = x.shape[0]
    assert B > 4 * 1024 * 1024
    g = torch.randn((L, B, CH), device="cuda") * 0.01
    rows = int(OFF[-1])
    offh = torch.from_numpy(OFF)
    need = _hip.lib().lnh_grid_backward_workspace_size(offh.data_ptr(), B, 3, CH, L, S, H, 0, 0, 0)
    need_small = _hip.lib().lnh_grid_backward_workspace_size(offh.data_ptr(), (B + 1) // 2, 3, CH, L, S, H, 0, 0, 0)
    assert need <= need_small * 1.01                          # the workspace serves one chunk, not the whole batch
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    a = torch.zeros((rows, CH), device="cuda")
    call("lnh_grid_encode_backward_ws", g, x, offh, a, B, 3, CH, L, S, H, 0, 0, 0, 0, ws, need)
    b = torch.zeros((rows, CH), device="cuda")
    call("lnh_grid_encode_backward", g, x, None, offh, b, B, 3, CH, L, S, H, None, None, 0, 0, 0, 0)
    torch.cuda.synchronize()
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) <= 2e-4 * scale
    # the data-parallel form of the same call on a chunked batch: begin (all chunks but the last in full + the last chunk's
    # scatter pass), then the last chunk's reduce pass window by window — bit-identical to the one-shot call
    c = torch.zeros((rows, CH), device="cuda")
    call("lnh_grid_encode_backward_ws_begin", g, x, offh, c, B, 3, CH, L, S, H, 0, 0, 0, 0, ws, need)
    for l0, l1 in ((0, 10), (10, L)):
        call("lnh_grid_encode_backward_ws_finish", g, x, offh, c, B, 3, CH, L, S, H, 0, 0, 0, 0, ws, need, l0, l1)
    assert torch.equal(c, a)
    offs = torch.from_numpy(OFF.astype(np.int64))
    for l in range(L):
        s_tab = a[offs[l]:offs[l + 1]].double().sum(0)
        s_g = g[l].double().sum(0)
        assert torch.allclose(s_tab, s_g, rtol=1e-3, atol=1e-2), (l, s_tab, s_g)
    a2 = torch.zeros((rows, CH), device="cuda")
    call("lnh_grid_encode_backward_ws", g, x, offh, a2, B, 3, CH, L, S, H, 0, 0, 0, 0, ws, need)
    assert torch.equal(a, a2)                                 # chunked sums are bit-reproducible too


def test_fused_table_trainer_checkpoint_resume(tmp_path):
    """Fused table optimizer: save after 3 steps, load into a fresh trainer, and the 4th step of both leaves bit-identical
    tables (moments, device-side step count, loss scale and the fp16 shadow all survive the reference-format checkpoint);
    the optimizer entry is loadable by the reference's stock Adam over model.get_params()."""
    import os
    import bench
    from lidarnerf.nerf.train_step import LidarTrainer
    dev = torch.device("cuda")
    kw = dict(num_steps=128, upsample_steps=32)

    def fresh():
        torch.manual_seed(0)
        from lidarnerf.nerf.network import NeRFNetwork
        m = NeRFNetwork(encoding="hashgrid", desired_resolution=32768, bound=1, min_near=bench.SCALE,
                        min_near_lidar=bench.SCALE).to(dev).train()
        return m, LidarTrainer(m, fp16=True, scale=bench.SCALE, render_kwargs=kw)
    poses = bench.synthetic_frames(4, dev)
    batches = [bench.make_batch(poses, s, 256, 0, dev) for s in range(4)]
    m1, t1 = fresh()
    assert t1.table is not None
    for s in range(3):
        torch.manual_seed(100 + s)
        t1.step(*batches[s])
    path = t1.save_checkpoint(os.path.join(tmp_path, "ck.pth"))
    m2, t2 = fresh()
    with torch.no_grad():
        m2.encoder.embeddings.add_(0.25)       # make sure the load really overwrites (and refreshes the fp16 shadow)
    t2.load_checkpoint(path)
    for t in (t1, t2):
        torch.manual_seed(103)
        t.step(*batches[3])
    assert torch.equal(m1.encoder.embeddings, m2.encoder.embeddings)
    assert torch.equal(t1.t_m, t2.t_m) and torch.equal(t1.t_v, t2.t_v) and float(t1.loss_scale) == float(t2.loss_scale)
    ref_opt = torch.optim.Adam(m1.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    ref_opt.load_state_dict(torch.load(path, weights_only=False)["optimizer"])
    assert float(ref_opt.state[m1.encoder.embeddings]["step"]) == 3


def test_empty_batches_are_no_ops_on_the_fused_step_entry_points():
    """N = 0 rays / B = 0 points: every entry point of the fused render step returns success without touching a byte of
    its outputs (and without launching anything: a zero-sized grid is a launch error on HIP)."""
    from gpu_util import call, wgrad
    from lidarnerf.gridencoder.grid import level_offsets
    canary = 123456.0
    d = torch.full((4096,), canary, device="cuda")              # stands in for every fp32 / fp16 / int32 pointer
    pls = float(np.exp2(np.log2(32768 / 16) / 15))
    off = torch.from_numpy(level_offsets(3, 16, pls, 16, 19, False))  # host-side int32 offsets, as the encoder keeps them
    S = float(np.log2(pls))
    T, t = 768, 64
    calls = [
        ("lnh_lidar_coarse_samples", d, 0, T, 0.01, 0.8, d),
        ("lnh_lidar_sample_points", d, d, d, d, 1.0, 0, T, T + t, 0, d),
        ("lnh_grid_encode_forward_mapped", d, d, off, d, 0, T, T + t, 0, 0, 2, 16, S, 16, 1),
        ("lnh_density_mlp_forward", d, d, 0, T, T + t, 0, 0, d, d),
        ("lnh_lidar_resample_strided", d, d, T + t, d, d, 0, T, t, 1.0, 1, d, d, d),
        ("lnh_lidar_merge_weights", d, d, d, d, 0, T + t, 1.0, d, d),
        ("lnh_lidar_dir_term", d, d, 90, 0, 75, d, d),
        ("lnh_lidar_color_forward", d, d, d, d, d, 0, T + t, d),
        ("lnh_lidar_color_composite_forward", d, d, d, d, d, d, d, 0, T + t, 1.0, d, d, d, d, d, d),
        ("lnh_lidar_composite_forward", d, d, d, d, 0, T + t, 2, 1.0, None, d, d, d),
        ("lnh_lidar_composite_backward", d, d, d, d, d, d, d, 0, T + t, 2, 1.0, d, None),
        ("lnh_lidar_color_backward", d, d, d, d, d, d, d, 0, T + t, d, d, d, *wgrad()),
        ("lnh_lidar_color_backward_image", d, d, d, d, d, d, d, 0, T + t, d, d, d, *wgrad()),
        ("lnh_lidar_dir_term_backward", d, d, 0, 75, None, d, 90, *wgrad()),
        ("lnh_density_mlp_backward", d, d, d, 0, T + t, T + t, 0, d, d, *wgrad()),
    ]
    for c in calls:
        call(*c)
    torch.cuda.synchronize()
    assert bool((d == canary).all())


def test_batches_beyond_32_bit_sample_indices_are_refused():
    """N * T >= 2^32: the per-sample indices of the fused-step kernels are 32 bits wide; the entry points say so instead of
    wrapping around (nothing is launched, the dummy pointers are never dereferenced)."""
    from gpu_util import call, wgrad
    d = torch.zeros(64, device="cuda")
    N, T = 1 << 22, 1 << 10
    for c in (("lnh_lidar_sample_points", d, d, d, d, 1.0, N, T, T, 0, d),
              ("lnh_lidar_color_forward", d, d, d, d, d, N, T, d),
              ("lnh_lidar_color_composite_forward", d, d, d, d, d, d, d, N, T, 1.0, d, d, d, d, d, d),
              ("lnh_lidar_color_backward", d, d, d, d, d, d, d, N, T, d, d, d, *wgrad()),
              ("lnh_lidar_color_backward_image", d, d, d, d, d, d, d, N, T, d, d, d, *wgrad())):
        with pytest.raises(RuntimeError, match="32 bits"):
            call(*c)
