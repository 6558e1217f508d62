"""Occupancy-grid training path (BASELINE config 4): ragged LiDAR compositing (K = 2, absolute depth with gradient) vs
the oracle, the occupancy render vs a recomposition of its own marched samples, the grid update, and a short training
run on an object-centric synthetic scene.  The indexing underneath (Morton codes, packbits, cell lookup, march_rays_train)
is pinned bit-exactly in tests/test_raymarch_gpu.py."""
import numpy as np
import pytest
import torch

from oracle import c_oracle, render_ref

pytestmark = pytest.mark.gpu
SCALE = 0.005  # configs/nerf_mvl.txt


def _ragged(N, seed, K=2):
    r = np.random.default_rng(seed)
    counts = r.integers(0, 40, N)
    counts[0] = 0                       # a ray without samples
    counts[1] = 1
    order = r.permutation(N)            # (id, offset, count) rows come in allocation order, not ray order
    offs = np.zeros(N, np.int64)
    off = 0
    rays = np.zeros((N, 3), np.int32)
    for row, ray in enumerate(order):
        rays[row] = (ray, off, counts[ray])
        off += counts[ray]
    M = int(off) + 5                    # a few unused tail slots
    sig = (r.random(M) * 30).astype(np.float32)
    sig[r.random(M) < 0.1] = 2000.0     # opaque samples: early termination inside a ray
    feats = r.random((M, K)).astype(np.float32)
    deltas = np.stack([r.random(M) * 0.02 + 0.002, r.random(M) * 0.03], -1).astype(np.float32)
    o = (r.random((N, 3)) - 0.5).astype(np.float32) * 0.2
    d = r.standard_normal((N, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    xyz = np.zeros((M, 3), np.float32)
    for row in range(N):
        ray, o0, c = rays[row]
        t = 0.05 + np.cumsum(r.random(c) * 0.03)
        xyz[o0:o0 + c] = o[ray] + d[ray] * t[:, None]
    return sig, feats, deltas, xyz, o, d, rays


@pytest.mark.parametrize("K", [1, 2, 3])
def test_ragged_lidar_composite_vs_oracle(K):
    from lidarnerf.raymarching import composite_rays_train_lidar
    sig, feats, deltas, xyz, o, d, rays = _ragged(57, 3 + K, K)
    t = lambda a: torch.from_numpy(a)
    sg, ft = t(sig).double().requires_grad_(True), t(feats).double().requires_grad_(True)
    ws, dep, img = render_ref.composite_ragged(sg, ft, t(deltas).double(), t(xyz).double(), t(o).double(), t(d).double(),
                                               t(rays))
    g = torch.Generator().manual_seed(1)
    gws, gdp, gim = torch.randn(57, generator=g), torch.randn(57, generator=g) * 5, torch.randn(57, K, generator=g)
    ((ws * gws.double()).sum() + (dep * gdp.double()).sum() + (img * gim.double()).sum()).backward()
    sc, fc = t(sig).cuda().requires_grad_(True), t(feats).cuda().requires_grad_(True)
    ws2, dep2, img2 = composite_rays_train_lidar(sc, fc, t(deltas).cuda(), t(xyz).cuda(), t(o).cuda(), t(d).cuda(),
                                                 t(rays).cuda(), 1e-4)
    torch.testing.assert_close(ws2.cpu().double(), ws, rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(dep2.cpu().double(), dep, rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(img2.cpu().double(), img, rtol=2e-5, atol=1e-6)
    ((ws2 * gws.cuda()).sum() + (dep2 * gdp.cuda()).sum() + (img2 * gim.cuda()).sum()).backward()
    scale = sg.grad.abs().max().item()
    torch.testing.assert_close(sc.grad.cpu().double(), sg.grad, rtol=1e-3, atol=2e-5 * scale)
    torch.testing.assert_close(fc.grad.cpu().double(), ft.grad, rtol=1e-4, atol=1e-7)
    # samples behind the termination point and the unused tail receive exactly zero gradient
    assert float(sc.grad[-5:].abs().max()) == 0.0


def _net(cuda_ray=True, seed=0):
    from lidarnerf.nerf.network import NeRFNetwork
    torch.manual_seed(seed)
    net = NeRFNetwork(encoding="hashgrid", desired_resolution=2048, bound=1, min_near=SCALE, min_near_lidar=SCALE,
                      density_thresh=10, cuda_ray=cuda_ray)
    with torch.no_grad():
        net.encoder.embeddings.uniform_(-0.5, 0.5)
    return net.cuda()


def _object_rays(n, seed):
    """Sensor on a circle of radius 0.6 around an object at the origin, rays towards a jittered point of the object."""
    g = torch.Generator().manual_seed(seed)
    ang = torch.rand(n, generator=g) * 2 * np.pi
    o = torch.stack([0.6 * torch.cos(ang), 0.6 * torch.sin(ang), torch.zeros(n)], -1)
    target = (torch.rand(n, 3, generator=g) - 0.5) * 0.3
    d = torch.nn.functional.normalize(target - o, dim=-1)
    return o, d


def test_run_cuda_matches_recomposition_of_its_samples():
    from lidarnerf import raymarching
    net = _net().eval()
    net.density_bitfield.fill_(255)                      # everything occupied: the marcher walks the whole ray
    o, d = _object_rays(64, 2)
    with torch.no_grad():
        out = net.render(o.cuda()[None], d.cuda()[None], cal_lidar_color=True, staged=False, perturb=False)
        nears = torch.full((64,), SCALE, device="cuda")
        xyzs, dirs, deltas, rays = raymarching.march_rays_train(o.cuda(), d.cuda(), 1, net.density_bitfield, 1, 128,
                                                                nears, nears * 81.0, None, -1, False, 128, True, 0, 1024)
        dens = net.density(xyzs)
        feats = net.color(xyzs, dirs, cal_lidar_color=True, mask=None, geo_feat=dens["geo_feat"])
    ws, dep, img = render_ref.composite_ragged(dens["sigma"].cpu().double(), feats.cpu().double(), deltas.cpu().double(),
                                               xyzs.cpu().double(), o.double(), d.double(), rays.cpu())
    torch.testing.assert_close(out["depth_lidar"][0].cpu().double(), dep, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out["image_lidar"][0].cpu().double(), img, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out["weights_sum_lidar"].cpu().double(), ws, rtol=1e-4, atol=1e-6)
    # far beyond the box nothing is sampled: 81 * scale = 0.405 < 0.6 + box, rays end inside; count is plausible
    assert 0 < xyzs.shape[0] <= 64 * 1024


def test_update_extra_state_grid_and_bitfield():
    net = _net().train()
    net.update_extra_state()
    grid = net.density_grid.clone()
    assert net.iter_density == 1 and float(grid.min()) >= 0 and float(grid.max()) > 0
    thresh = min(net.mean_density, net.density_thresh)
    want = c_oracle.packbits(grid.cpu().numpy().reshape(-1), thresh)
    np.testing.assert_array_equal(net.density_bitfield.cpu().numpy(), want)
    net.update_extra_state()                              # EMA: never below decay * previous
    assert bool((net.density_grid >= grid * 0.95 - 1e-6).all())
    net.reset_extra_state()
    assert float(net.density_grid.abs().max()) == 0.0 and net.iter_density == 0


def test_occupancy_training_prunes_empty_space():
    """Object-centric scene (a sphere of radius 0.15 seen from a ring of sensor positions): 96 training steps with the
    occupancy path must bring the loss down while marching far fewer samples per ray than the 768 + 64 of the dense path
    (at scale 0.005 the 1 m .. 81 m ray segment is 0.4 scene units = at most 118 marcher steps of 2 sqrt(3) / 1024)."""
    from lidarnerf.nerf.train_step import LidarTrainer
    net = _net(seed=1).train()
    with torch.no_grad():
        net.encoder.embeddings.uniform_(-1e-4, 1e-4)
    tr = LidarTrainer(net, lr=1e-2, fp16=True, scale=SCALE, render_kwargs={})
    assert tr.occupancy and tr.table is not None  # fused ragged chain: the table is stepped by the fused optimizer
    R = 0.15
    losses, counts = [], []
    for step in range(96):
        o, d = _object_rays(2048, 100 + step)
        b = (o * d).sum(-1)
        disc = b * b - ((o * o).sum(-1) - R * R)
        hit = disc > 0
        depth = torch.where(hit, -b - torch.sqrt(disc.clamp(min=0)), torch.zeros_like(b))
        gt = torch.stack([hit.float(), torch.full_like(b, 0.5), depth], -1)[None].cuda()
        loss = tr.step(o.cuda()[None], d.cuda()[None], gt)
        losses.append(float(loss.detach()))
        counts.append(int(net.step_counter[(net.local_step - 1) % 16, 0]) / 2048)
    assert np.mean(losses[-8:]) < 0.35 * np.mean(losses[:8]), (losses[:8], losses[-8:])
    assert 0 < np.mean(counts[-8:]) < 0.25 * 832, (counts[:8], counts[-8:])
    assert all(np.isfinite(losses))


def test_fused_ragged_chain_vs_oracle_and_modular_path():
    """The occupancy-grid render as ONE autograd node (nerf/fused.py FusedLidarRagged: encode -> sigma net -> colour head ->
    ragged compositing) against (a) the CPU restatement evaluated on the very samples the marcher produced — RefLidarField
    (network.py:162-237) + composite_ragged (raymarching.cu:577-655 weights, renderer.py:268-271 outputs), parameters
    representable in fp16 — outputs and every gradient; (b) the modular path (separate autograd nodes per module)."""
    from lidarnerf import raymarching
    from lidarnerf.nerf.network import NeRFNetwork
    from lidarnerf.nerf.train_step import lidar_loss
    torch.manual_seed(5)
    ref = render_ref.RefLidarField(desired_resolution=2048)
    with torch.no_grad():
        ref.embeddings.uniform_(-0.4, 0.4)
        for p in ref.parameters():
            p.copy_(p.half().float())

    def product(fused_lidar):
        net = NeRFNetwork(encoding="hashgrid", desired_resolution=2048, bound=1, min_near=SCALE, min_near_lidar=SCALE,
                          density_thresh=10, cuda_ray=True, fused_lidar=fused_lidar)
        with torch.no_grad():
            net.encoder.embeddings.copy_(ref.embeddings)
            for a, b in list(zip(net.sigma_net, ref.sigma_net)) + list(zip(net.lidar_color_net, ref.lidar_color_net)):
                a.weight.copy_(b.weight)
        net = net.cuda().train()
        net.density_bitfield.fill_(255)
        return net

    N = 48
    o, d = _object_rays(N, 7)
    g = torch.Generator().manual_seed(8)
    gt = torch.rand(N, 3, generator=g)
    gt[:, 0] = (gt[:, 0] > 0.2).float()
    gt[:, 2] *= 0.3
    scale = 64.0
    res = {}
    for name, flag in (("fused", True), ("modular", False)):
        net = product(flag)
        with torch.autocast("cuda", dtype=torch.float16):
            out = net.render(o.cuda()[None], d.cuda()[None], cal_lidar_color=True, staged=False, perturb=False,
                             force_all_rays=True)
            loss, _, _ = lidar_loss(out, gt.cuda()[None])
        (loss * scale).backward()
        res[name] = (out, float(loss.detach()), net)
    # the marcher's samples (deterministic without perturbation) for the CPU side
    nears = torch.full((N,), SCALE, device="cuda")
    _, far_box = raymarching.near_far_from_aabb(o.cuda(), d.cuda(), res["fused"][2].aabb_train, SCALE)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(o.cuda(), d.cuda(), 1, res["fused"][2].density_bitfield, 1, 128,
                                                            nears, torch.minimum(nears * 81.0, far_box), None, -1, False, 128,
                                                            True, 0, 1024)
    M = xyzs.shape[0]
    assert 1000 < M < N * 1024
    ref.storage = torch.float16  # the CPU side models WHERE 16-bit values are stored (oracle/render_ref.py RefLidarField)
    sigma, geo = ref.density(xyzs.cpu())
    feats = ref.color(xyzs.cpu(), dirs.cpu(), torch.ones(M, dtype=torch.bool), geo)
    ws, dep, img = render_ref.composite_ragged(sigma, feats, deltas.cpu(), xyzs.cpu(), o, d, rays.cpu())
    lw = render_ref.lidar_loss(dep, img, gt)
    (lw * scale).backward()  # the same loss scale: 16-bit gradient rows round at the same magnitudes
    out, loss, net = res["fused"]
    # measured on MI355X against the storage model: outputs 1.5e-6 .. 3.5e-6 of the largest entry, hash-table gradient
    # 2.4e-4, MLP weight gradients <= 3.1e-5 (against the fp32 restatement without the model: 2e-4 / 1e-2 / 5e-3)
    for got, want in ((out["depth_lidar"][0], dep), (out["image_lidar"][0], img), (out["weights_sum_lidar"], ws)):
        err = (got.detach().float().cpu() - want.detach().float()).abs().max().item() / (want.detach().abs().max().item() + 1e-12)
        assert err < 2e-5, err
    assert abs(loss - float(lw.detach())) <= 2e-5 * abs(float(lw.detach()))
    ge = net.encoder.embeddings.grad.detach().float().cpu().double() / scale
    gr = ref.embeddings.grad.double() / scale
    assert ((ge - gr).norm() / gr.norm()).item() < 8e-4
    for a, b in list(zip(net.sigma_net, ref.sigma_net)) + list(zip(net.lidar_color_net, ref.lidar_color_net)):
        ga, gb = a.weight.grad.detach().float().cpu().double() / scale, b.weight.grad.double() / scale
        assert ((ga - gb).norm() / gb.norm()).item() < 2e-4
    # (b) the modular path computes the same step through other kernels / autograd nodes
    outm, lossm, netm = res["modular"]
    for k in ("depth_lidar", "image_lidar", "weights_sum_lidar"):
        torch.testing.assert_close(out[k].float(), outm[k].float(), rtol=3e-3, atol=3e-4)
    gm = netm.encoder.embeddings.grad.detach().float().cpu().double() / scale
    assert ((ge - gm).norm() / gm.norm()).item() < 2e-2


def test_training_step_with_no_marched_sample():
    """Every ray misses the occupied cells (empty bitfield): the fused ragged node runs on ZERO samples — outputs are zeros
    connected to the graph, the step goes through with zero gradients (the table and the MLP weights do not move: Adam on a
    zero gradient from zero moments), and nothing raises.  (Before: run_cuda returned graph-less zeros and the step failed
    in backward, or with 'produced no fp16 table gradient'; under data parallel that rank skipped the table all-reduce.)"""
    from lidarnerf.nerf.train_step import LidarTrainer
    net = _net(seed=3).train()
    tr = LidarTrainer(net, lr=1e-2, fp16=True, scale=SCALE, render_kwargs={})
    assert tr.occupancy and tr.table is not None
    tr.update_extra_interval = 10 ** 9            # keep the (emptied) occupancy grid as it is
    tr.global_step = 1
    net.density_bitfield.zero_()
    before = net.encoder.embeddings.detach().clone()
    w_before = net.sigma_net[0].weight.detach().clone()
    o, d = _object_rays(256, 5)
    gt = torch.rand(1, 256, 3, device="cuda")
    loss = tr.step(o.cuda()[None], d.cuda()[None], gt)
    assert torch.isfinite(loss) and float(loss) > 0                     # |0 - gt| terms
    assert torch.equal(net.encoder.embeddings.detach(), before)
    assert torch.equal(net.sigma_net[0].weight.detach(), w_before)
    g16 = tr.table._lnh_grad16
    assert g16 is not None and float(g16.abs().max()) == 0.0
    # and the next step, with samples again, trains normally
    net.density_bitfield.fill_(255)
    loss2 = tr.step(o.cuda()[None], d.cuda()[None], gt)
    assert torch.isfinite(loss2) and not torch.equal(net.encoder.embeddings.detach(), before)


def _sphere_batch(n, seed, R=0.15):
    o, d = _object_rays(n, seed)
    b = (o * d).sum(-1)
    disc = b * b - ((o * o).sum(-1) - R * R)
    hit = disc > 0
    depth = torch.where(hit, -b - torch.sqrt(disc.clamp(min=0)), torch.zeros_like(b))
    gt = torch.stack([hit.float(), torch.full_like(b, 0.5), depth], -1)[None].cuda()
    return o.cuda()[None], d.cuda()[None], gt


def test_captured_step_trains_like_the_eager_step():
    """LidarTrainer(graph=True): the occupancy-grid step (march .. both optimizers .. loss-scale update) captured in a
    hipGraph per sample capacity and replayed.  Against an eager trainer on the same batches from the same initial state:
    the same loss trajectory (not bit-equal: the jitter comes from the generator's graph-safe Philox stream, the marcher's
    capacity is rounded up to a ladder of sizes, and the MLP weight gradients are float atomics), the learning rate
    follows the schedule through the device scalar, the step counters of both optimizers advance once per replay, the
    marcher's counters reach the ring update_extra_state reads, and a checkpoint taken in graph mode loads into an eager
    trainer."""
    import os
    import tempfile
    from lidarnerf.nerf.train_step import LidarTrainer
    nets = [_net(seed=1).train(), _net(seed=1).train()]
    for net in nets:
        with torch.no_grad():
            net.encoder.embeddings.uniform_(-1e-4, 1e-4)
    nets[1].load_state_dict(nets[0].state_dict())
    eager = LidarTrainer(nets[0], lr=1e-2, iters=200, fp16=True, scale=SCALE, render_kwargs={})
    graph = LidarTrainer(nets[1], lr=1e-2, iters=200, fp16=True, scale=SCALE, render_kwargs={}, graph=True)
    assert graph.graph and not eager.graph
    le, lg = [], []
    n_steps = 96
    for step in range(n_steps):
        batch = _sphere_batch(2048, 100 + step)
        le.append(float(eager.step(*batch).detach()))
        lg.append(float(graph.step(*batch).detach()))
    assert len(graph._graphs) >= 1                        # captured (the first 16 steps ran launch by launch)
    assert len(graph._graphs) <= 8, list(graph._graphs)   # ... and the capacity ladder keeps the number of graphs small
    caps = sorted(k[-1] for k in graph._graphs)
    assert all(c % 1024 == 0 for c in caps) and caps[-1] >= nets[1].mean_count
    assert all(np.isfinite(lg))
    # same training: start equal, both fall by the same factor, and the smoothed trajectories stay close
    np.testing.assert_allclose(lg[0], le[0], rtol=1e-3)
    se, sg = np.convolve(le, np.ones(8) / 8, "valid"), np.convolve(lg, np.ones(8) / 8, "valid")
    assert sg[-1] < 0.35 * sg[0]
    assert np.abs(sg - se).max() < 0.25 * se[0], (se[::8], sg[::8])
    assert abs(sg[-1] - se[-1]) < 0.5 * se[-1] + 0.02 * se[0], (se[-1], sg[-1])
    # learning rate: the schedule's value — on the host (scheduler bookkeeping) and in the device scalar the optimizer
    # kernel formed from its own step counter for the LAST step (n_steps - 1 scheduler steps before it)
    from lidarnerf import _hip
    np.testing.assert_allclose(float(graph.optimizer.param_groups[0]["lr"]), 1e-2 * 0.1 ** (n_steps / 200), rtol=1e-5)
    np.testing.assert_allclose(float(graph.opt_state[_hip.TS_LR]), 1e-2 * 0.1 ** ((n_steps - 1) / 200), rtol=1e-5)
    assert float(graph.opt_state[_hip.TS_IT_NEXT]) == n_steps == graph.scheduler.last_epoch
    # step counter: one device-side counter for every parameter, moved once per step that was not skipped
    skipped = n_steps - int(graph.t_steps[graph.t_flip])
    assert 0 <= skipped <= 3                              # (the dynamic loss scale may back off a couple of times early on)
    assert graph.steps_taken() == n_steps - skipped
    # the marcher's counters reach the ring: update_extra_state derived a plausible mean from them
    assert 0 < nets[1].mean_count < 2048 * 832 * 0.25
    assert abs(nets[1].mean_count - nets[0].mean_count) < 0.3 * nets[0].mean_count
    # a checkpoint written in graph mode carries plain numbers and loads into an eager trainer, which continues
    with tempfile.TemporaryDirectory() as tmp:
        path = graph.save_checkpoint(os.path.join(tmp, "g.pth"))
        ck = torch.load(path, weights_only=False)
        assert isinstance(ck["optimizer"]["param_groups"][0]["lr"], float)
        third = LidarTrainer(_net(seed=7).train(), lr=1e-2, iters=200, fp16=True, scale=SCALE, render_kwargs={})
        third.load_checkpoint(path)
        l3 = float(third.step(*_sphere_batch(2048, 500)).detach())
        # ... and back into a graph-mode trainer: the captured steps are dropped, the device counters follow the file
        graph.load_checkpoint(path)
        assert not graph._graphs and float(graph.opt_state[_hip.TS_IT_NEXT]) == graph.scheduler.last_epoch == n_steps
        l4 = float(graph.step(*_sphere_batch(2048, 500)).detach())
        l5 = float(graph.step(*_sphere_batch(2048, 501)).detach())   # (recaptured)
    assert np.isfinite([l3, l4, l5]).all() and abs(l3 - l4) < 0.3 * max(l3, l4) + 1e-3
