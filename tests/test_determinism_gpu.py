"""Bit-reproducibility of the training path.  Every sum over samples on it has a fixed order: the hash-table gradient is an
integer sum per row (csrc/grid.hip, bucketed backward), the weight gradients of the MLP family are per-workgroup partials
added up in index order (csrc/wgrad.h — rounds 1-5 used fp32 device atomics there, and two runs of the same seed drifted
apart within a few hundred steps), the loss is a slot-ordered sum.  So: the same entry point twice gives the same bits, and
two training runs from the same seed give the same table and the same weights — launch by launch and as a captured step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SCALE = 0.010784853507573345


def _repeat(fn, n=4):
    outs = [fn() for _ in range(n)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)
    return outs[0]


def test_mlp_backward_weight_gradients_same_bits_every_launch():
    """lnh_mlp_backward in its three kernel families (wave-independent one-hidden-layer, workgroup-cooperative with 1 and 2
    hidden matrices, hidden 32) on batches that occupy every workgroup of the launch."""
    from gpu_util import call, wgrad
    g = torch.Generator().manual_seed(3)
    for in_dim, hidden, nhm, B in ((32, 64, 0, 300_000), (64, 64, 0, 200_000), (96, 64, 1, 150_000), (32, 64, 2, 200_000),
                                   (48, 32, 1, 100_000)):
        nw = hidden * in_dim + nhm * hidden * hidden + 16 * hidden
        x = (torch.randn(B, in_dim, generator=g) * 0.5).half().cuda()
        gy = (torch.randn(B, 16, generator=g) * 0.1).half().cuda()
        w = (torch.randn(nw, generator=g) * 0.2).half().cuda()

        def run():
            gw = torch.zeros(nw, device="cuda")
            gx = torch.empty((B, in_dim), dtype=torch.half, device="cuda")
            call("lnh_mlp_backward", gy, x, w, B, in_dim, 16, hidden, nhm, 0, 6, gx, gw, *wgrad())
            return gw, gx
        gw, _ = _repeat(run)
        assert float(gw.abs().sum()) > 0
        # ... and the sum is ADDED to what grad_weights holds (two calls = twice the gradient, exactly: x + x)
        gw2 = torch.zeros(nw, device="cuda")
        for _ in range(2):
            call("lnh_mlp_backward", gy, x, w, B, in_dim, 16, hidden, nhm, 0, 6, None, gw2, *wgrad())
        assert torch.equal(gw2, gw + gw)


def test_color_and_density_backward_same_bits_every_launch():
    from gpu_util import call, wgrad
    g = torch.Generator().manual_seed(4)
    N, T = 1500, 832
    h16 = (torch.randn(N * T, 16, generator=g) * 0.5).half().cuda()
    perm = torch.stack([torch.randperm(T, generator=g) for _ in range(N)]).int().cuda()
    weights = (torch.rand(N, T, generator=g) * (torch.rand(N, T, generator=g) > 0.6)).cuda() * 1e-2
    cdir = torch.randn(N, 64, generator=g).cuda()
    w16 = (torch.randn(64 * 16 + 64 * 64 + 16 * 64, generator=g) * 0.2).half().cuda()
    g_image = torch.randn(N, 2, generator=g).cuda()
    g_sigma = (torch.randn(N, T, generator=g) * 1e-3).cuda()

    def color():
        g_h16 = torch.empty((N * T, 16), dtype=torch.half, device="cuda")
        g_w = torch.zeros(w16.numel(), device="cuda")
        S = torch.empty((N, 64), device="cuda")
        call("lnh_lidar_color_backward_image", g_image, g_sigma, h16, perm, weights, cdir, w16, N, T, g_h16, g_w, S, *wgrad())
        return g_w, S, g_h16
    g_w, S, g_h16 = _repeat(color)
    assert float(g_w.abs().sum()) > 0

    feat = (torch.randn(16, N * T, 2, generator=g) * 0.3).half().cuda()
    wsig = (torch.randn(64 * 32 + 16 * 64, generator=g) * 0.2).half().cuda()

    def density():
        g_feat = torch.empty((16, N * T, 2), dtype=torch.half, device="cuda")
        gw = torch.zeros(wsig.numel(), device="cuda")
        call("lnh_density_mlp_backward", g_h16, feat, wsig, N * T, T, T, 0, g_feat, gw, *wgrad())
        return gw, g_feat
    gw, _ = _repeat(density)
    assert float(gw.abs().sum()) > 0

    E = torch.randn(N, 75, generator=g).cuda()

    def dir_term():
        gW = torch.zeros(64, 90, device="cuda")
        call("lnh_lidar_dir_term_backward", S, E, N, 75, g_w[:1024], gW, 90, *wgrad())
        return (gW,)
    (gW,) = _repeat(dir_term)
    torch.testing.assert_close(gW[:, :75].double(), S.double().t() @ E.double(), rtol=1e-4, atol=1e-3)


def test_the_workspace_needs_no_preparation_and_is_checked():
    """The weight-gradient workspace is scratch: whatever it holds when a launch starts, the gradient comes out the same (no
    counters to keep zeroed); a workspace that is missing or too small is refused by name."""
    from lidarnerf import _hip
    from gpu_util import call, wgrad
    ptr, nbytes = wgrad()
    g = torch.Generator().manual_seed(5)
    B = 50_000
    x, gy = (torch.randn(B, 32, generator=g)).half().cuda(), (torch.randn(B, 16, generator=g)).half().cuda()
    w = (torch.randn(64 * 32 + 16 * 64, generator=g) * 0.2).half().cuda()
    ws = _hip._WGRAD_WS[torch.cuda.current_device()]
    assert nbytes == ws.numel()
    outs = []
    for fill in (0, 0xFF, 0x7F):
        ws.fill_(fill)  # 0xFF..: NaN patterns, 0x7F7F7F7F: 3.4e38
        gw = torch.zeros(w.numel(), device="cuda")
        call("lnh_mlp_backward", gy, x, w, B, 32, 16, 64, 0, 0, 6, None, gw, *wgrad())
        outs.append(gw)
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # a workspace that is missing or too small is refused by name
    with pytest.raises(RuntimeError, match="lnh_wgrad_workspace_bytes"):
        call("lnh_mlp_backward", gy, x, w, B, 32, 16, 64, 0, 0, 6, None, gw, None, 0)
    with pytest.raises(RuntimeError, match="lnh_wgrad_workspace_bytes"):
        call("lnh_mlp_backward", gy, x, w, B, 32, 16, 64, 0, 0, 6, None, gw, ptr, 4096)


def _train(patch, steps, graph):
    import bench
    from lidarnerf.nerf.train_step import LidarTrainer
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = bench.build_model(dev)
    tr = LidarTrainer(model, lr=1e-2, iters=30000, fp16=True, scale=bench.SCALE, graph=graph,
                      render_kwargs=dict(num_steps=768, upsample_steps=64))
    poses = bench.synthetic_frames(8, dev)
    batches = [bench.make_batch(poses, s, 4096, 0, dev, patch, "analytic") for s in range(8)]
    torch.manual_seed(11)
    losses = []
    for s in range(steps):
        losses.append(tr.step(*batches[s % 8], **({} if patch == (1, 1) else {"patch": patch})).detach().clone())
    torch.cuda.synchronize()
    state = [tr.table.detach().clone(), tr.table._lnh_table16.clone(), tr.t_m.clone(), tr.t_v.clone(), tr.opt_state.clone()]
    state += [p.detach().clone() for p in tr.small] + [torch.stack(losses)]
    return state


@pytest.mark.parametrize("patch", [(1, 1), (2, 8)])
def test_two_training_runs_from_one_seed_are_bit_identical(patch):
    """100 optimizer steps at the benchmark's shape (4096 rays x 832 samples), twice: the fp32 master table, its fp16 copy,
    both Adam moments, every MLP matrix, the optimizer's scalars (loss scale, step counts) and all 100 losses agree bit for
    bit — and a third run through the captured step (hipGraph replay) gives the same bits again."""
    a = _train(patch, 100, graph=False)
    b = _train(patch, 100, graph=False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    c = _train(patch, 100, graph=True)
    for x, y in zip(a, c):
        assert torch.equal(x, y)
    assert np.isfinite(a[-1].float().cpu().numpy()).all() and float(a[4][0]) > 0  # (loss scale alive)
