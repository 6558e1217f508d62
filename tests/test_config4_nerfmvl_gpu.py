"""BASELINE config 4 at its shape: NeRF-MVL object scene — 256 x 1800 range image, intrinsics (fov_up, fov) = (15, 40), scale
0.005, the rays of a frame restricted to the object's bounding sphere and thinned to 4096 per step (nerfmvl_dataset.py:116-168),
hash grid up to resolution 32768, OCCUPANCY-GRID ray sampling (lnh_march_rays_train on an occupancy grid that marks the
object's neighbourhood), the fused ragged chain and the fused table optimizer.  What `python bench.py --workload nerfmvl`
times, as a parity test:

  * 32 rays of the 4096-ray batch against the CPU restatement (oracle/render_ref.py RefLidarField + composite_ragged) evaluated
    on the very samples the marcher produced for them (rays are independent);
  * properties at full size: finite, weights_sum in [0, 1], depth inside [near, far] x weights_sum, samples only inside
    occupied cells and far fewer than the dense path's 832 per ray; the step is repeatable (the marcher's atomic offset
    counter permutes the rays in the sample list, so the table gradient agrees to the last fp16 bit on all but a few rows),
    and linear over rays (gradient of the batch = sum over its quarters).
"""
import numpy as np
import pytest
import torch

from oracle import render_ref

pytestmark = pytest.mark.gpu
SCALE, H, W, INTR = 0.005, 256, 1800, (15.0, 40.0)
R, RING = 2.0 * SCALE, 6.0 * SCALE


def _field(seed):
    from lidarnerf.nerf.network import NeRFNetwork
    torch.manual_seed(seed)
    ref = render_ref.RefLidarField(desired_resolution=32768)
    with torch.no_grad():
        ref.embeddings.uniform_(-0.4, 0.4)
        for p in ref.parameters():
            p.copy_(p.half().float())
    net = NeRFNetwork(encoding="hashgrid", desired_resolution=32768, log2_hashmap_size=19, bound=1, min_near=SCALE,
                      min_near_lidar=SCALE, density_thresh=10, cuda_ray=True)
    with torch.no_grad():
        net.encoder.embeddings.copy_(ref.embeddings)
        for a, b in list(zip(net.sigma_net, ref.sigma_net)) + list(zip(net.lidar_color_net, ref.lidar_color_net)):
            a.weight.copy_(b.weight)
    return net.cuda().train(), ref.train()


def _occupy_object(net):
    """Occupancy grid of a trained object scene: the cells touching the ball of 1.3 R around the origin (Morton order,
    lnh_packbits)."""
    from lidarnerf import raymarching
    G = net.grid_size
    ar = torch.arange(G, dtype=torch.int32, device="cuda")
    xx, yy, zz = torch.meshgrid(ar, ar, ar, indexing="ij")
    coords = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1).contiguous()
    idx = raymarching.morton3D(coords).long()
    centre = (2 * coords.float() / (G - 1) - 1) * (net.bound - net.bound / G)
    # at scale 0.005 the 2 m object is smaller than a grid cell (2 / 128 scene units): every cell that touches the ball
    occ = centre.norm(dim=-1) < 1.3 * R + np.sqrt(3) / G
    net.density_grid.zero_()
    net.density_grid[0, idx] = occ.float()
    raymarching.packbits(net.density_grid, 0.5, net.density_bitfield)
    return float(occ.float().mean())


def _batch(n_rays):
    from lidarnerf.dataset.rays import get_lidar_rays
    th = 0.7
    pose = torch.eye(4)
    pose[:3, :3] = torch.tensor([[-np.cos(th), np.sin(th), 0], [-np.sin(th), -np.cos(th), 0], [0, 0, 1.0]])
    pose[:3, 3] = torch.tensor([RING * np.cos(th), RING * np.sin(th), 0.0])
    r = get_lidar_rays(pose[None], INTR, H, W, -1)
    o, d = r["rays_o"][0], r["rays_d"][0]
    assert o.shape[0] == H * W
    b = (o * d).sum(-1)
    keep = ((b * b - ((o * o).sum(-1) - (1.2 * R) ** 2)) > 0) & (b < 0)           # rays through the object's bounding sphere
    o, d, b = o[keep], d[keep], b[keep]
    assert o.shape[0] > n_rays
    g = torch.Generator().manual_seed(4)
    sel = torch.randperm(o.shape[0], generator=g)[:n_rays]
    o, d, b = o[sel].contiguous(), d[sel].contiguous(), b[sel]
    disc = b * b - ((o * o).sum(-1) - R * R)
    hit = disc > 0
    depth = torch.where(hit, -b - torch.sqrt(disc.clamp(min=0)), torch.zeros_like(b))
    gt = torch.stack([hit.float(), torch.rand(n_rays, generator=g), depth], -1)
    return o, d, gt


def _step(net, o, d, gt, scale):
    from lidarnerf.nerf.train_step import lidar_loss
    net.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        out = net.render(o.cuda()[None], d.cuda()[None], cal_lidar_color=True, staged=False, perturb=False,
                         force_all_rays=True)
        loss, _, _ = lidar_loss(out, gt.cuda()[None])
    (loss * (o.shape[0] * scale)).backward()     # a SUM loss: additive over rays
    grads = [net.encoder.embeddings.grad.detach().clone()] + \
        [m.weight.grad.detach().clone() for m in list(net.sigma_net) + list(net.lidar_color_net)]
    return {k: v.detach().float().clone() for k, v in out.items()}, float(loss.detach()), grads


def test_config4_nerfmvl_shape_occupancy_path():
    from lidarnerf import raymarching
    from lidarnerf.nerf import fused
    N = 4096
    net, ref = _field(41)
    assert fused.ragged_supported(net) and net.fused_lidar
    occupied = _occupy_object(net)
    assert 1e-6 < occupied < 1e-3                                   # the object's neighbourhood: a tiny part of the box
    o, d, gt = _batch(N)
    scale = 2.0 ** -4
    out, loss, grads = _step(net, o, d, gt, scale)
    torch.cuda.synchronize()
    ws, depth, image = out["weights_sum_lidar"].reshape(-1), out["depth_lidar"].reshape(-1), out["image_lidar"].reshape(-1, 2)
    assert torch.isfinite(ws).all() and torch.isfinite(depth).all() and torch.isfinite(image).all()
    assert all(torch.isfinite(g).all() for g in grads) and float(grads[0].abs().sum()) > 0
    assert float(ws.min()) >= 0 and float(ws.max()) <= 1 + 1e-5
    assert (depth >= SCALE * ws * (1 - 1e-3) - 1e-7).all() and (depth <= 81 * SCALE * ws + 1e-6).all()
    assert (image >= 0).all() and (image <= ws[:, None] + 1e-5).all()
    # the marcher's samples of the batch: inside occupied cells only, far fewer than the dense path's 832 per ray
    nears = torch.full((N,), SCALE, device="cuda")
    _, far_box = raymarching.near_far_from_aabb(o.cuda(), d.cuda(), net.aabb_train, SCALE)
    fars = torch.minimum(nears * 81.0, far_box)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(o.cuda(), d.cuda(), 1, net.density_bitfield, 1, 128, nears, fars,
                                                            None, -1, False, 128, True, 0, 1024)
    cnt = rays[:, 2].long()
    M = int(cnt.sum())
    assert 0 < M < 0.25 * 832 * N and int((cnt > 0).sum()) > 0.9 * N
    assert float(xyzs[:M].norm(dim=-1).max()) < 1.3 * R + 6 * np.sqrt(3) / 128     # within the occupied cells around the ball

    # ---- the same step again.  The marcher hands out sample offsets with an atomic counter (raymarching.cu:331-534, as the
    #      reference does), so the ORDER of the rays in the flat sample list differs run to run: per-ray outputs are
    #      identical, the hash-table gradient sums the same contributions in another grouping (the wave run-merge pairs
    #      other neighbours) — equal to the last fp16 bit on all but a handful of rows
    out2, loss2, grads2 = _step(net, o, d, gt, scale)
    assert abs(loss2 - loss) <= 1e-6 * abs(loss)
    for k in out:
        torch.testing.assert_close(out[k], out2[k], rtol=1e-6, atol=1e-7)
    dg = (grads[0].double() - grads2[0].double())
    assert (dg.norm() / grads[0].double().norm()).item() < 5e-5 and int((dg.abs().sum(1) > 0).sum()) < 1000

    # ---- 32 rays of the batch against the CPU restatement on the marcher's own samples for those rays
    sel = torch.arange(0, N, N // 32)[:32]
    xs, ds_, dl, ry = raymarching.march_rays_train(o[sel].cuda(), d[sel].cuda(), 1, net.density_bitfield, 1, 128, nears[sel],
                                                   fars[sel], None, -1, False, 128, True, 0, 1024)
    m = xs.shape[0]
    ref.storage = torch.float16
    sigma, geo = ref.density(xs.cpu())
    feats = ref.color(xs.cpu(), ds_.cpu(), torch.ones(m, dtype=torch.bool), geo)
    ws_r, dep_r, img_r = render_ref.composite_ragged(sigma, feats, dl.cpu(), xs.cpu(), o[sel], d[sel], ry.cpu())
    for got, want in ((depth.cpu()[sel], dep_r), (image.cpu()[sel], img_r), (ws.cpu()[sel], ws_r)):
        err = (got - want.detach()).abs().max().item() / (want.detach().abs().max().item() + 1e-12)
        assert err < 5e-5, err

    # ---- linearity over rays: gradient of the batch = sum of the gradients of its four quarters
    acc = [torch.zeros_like(g_, dtype=torch.float64) for g_ in grads]
    for q in range(4):
        s = slice(q * N // 4, (q + 1) * N // 4)
        _, _, gq = _step(net, o[s], d[s], gt[s], scale)
        for a, g_ in zip(acc, gq):
            a += g_.double()
    rel = ((grads[0].double() - acc[0]).norm() / acc[0].norm()).item()
    assert rel < 2e-3, rel                                          # fp16 rows are rounded once per call on both sides
    for i in range(1, len(grads)):
        rel = ((grads[i].double() - acc[i]).norm() / acc[i].norm()).item()
        assert rel < 1e-4, (i, rel)


def test_config4_nerfmvl_shape_captured_training():
    """The benchmarked form of config 4 at its shape: LidarTrainer(graph=True) — 4096 rays per step through the 32768-resolution
    grid, the step captured in a hipGraph per sample capacity and replayed.  64 steps on the synthetic object: the loss falls,
    every step after the first grid-update block is a replay (a handful of captures at most), the marched samples stay far
    below the dense path's 832 per ray, and the replayed steps advance optimizer state exactly once each."""
    from lidarnerf.nerf.network import NeRFNetwork
    from lidarnerf.nerf.train_step import LidarTrainer
    torch.manual_seed(5)
    net = NeRFNetwork(encoding="hashgrid", desired_resolution=32768, log2_hashmap_size=19, bound=1, min_near=SCALE,
                      min_near_lidar=SCALE, density_thresh=10, cuda_ray=True).cuda().train()
    tr = LidarTrainer(net, lr=1e-2, iters=30000, fp16=True, scale=SCALE, graph=True)
    assert tr.graph and tr.occupancy and tr.table is not None
    o, d, gt = _batch(4096)
    o, d, gt = o.cuda()[None], d.cuda()[None], gt.cuda()[None]
    losses, replays = [], 0
    for step in range(64):
        n_before = len(tr._graphs)
        warm = bool(tr._graph_warm) and tr._graph_capacity() > 0
        losses.append(float(tr.step(o, d, gt)))
        replays += int(warm and len(tr._graphs) == n_before)
    assert np.isfinite(losses).all() and np.mean(losses[-8:]) < 0.5 * losses[0], (losses[:8], losses[-8:])
    assert 1 <= len(tr._graphs) <= 4 and replays >= 64 - 17 - len(tr._graphs), (len(tr._graphs), replays)
    skipped = 64 - int(tr.t_steps[tr.t_flip])
    assert 0 <= skipped <= 4                                   # (loss-scale back-offs)
    assert tr.steps_taken() == 64 - skipped and float(tr.small_m.abs().sum()) > 0   # (the MLP weights were stepped by the same kernel)
    assert 0 < net.mean_count < 0.25 * 832 * 4096
