"""BASELINE config 5 at its batch size: 16 384 rays x (768 + 64) samples through the fused chain with fp16 hash features
and bf16 MFMA MLPs (torch.autocast(dtype=bfloat16)) — 13.6 M sample points, the table-gradient backward walked in 4 chunks
of <= 4 M points.  At this size the CPU oracle cannot follow, so:

  * a 32-ray subset of THE SAME batch (same rays, same random draws) is rendered by oracle/render_ref.py and must agree with
    those rays' outputs of the big batch (rays are independent: renderer.py:99-298 has no cross-ray term);
  * size-independent properties: the step is bit-reproducible (outputs and hash-table gradient: the bucketed
    scatter-reduce has one integer sum per row and chunk; MLP weight gradients to fp32 summation order); the gradient of a SUM loss over the batch equals the sum
    of the gradients of its four 4096-ray quarters rendered on their own (linearity over rays — catches a chunk that is
    dropped, doubled or mis-addressed); weights_sum in [0, 1], depth inside [near, far] x weights_sum, everything finite.
"""
import numpy as np
import pytest
import torch

from oracle import render_ref

pytestmark = pytest.mark.gpu
SCALE = 0.010784853507573345
N, T, t = 16384, 768, 64


def _field(seed):
    from lidarnerf.nerf.network import NeRFNetwork
    torch.manual_seed(seed)
    ref = render_ref.RefLidarField(desired_resolution=32768)
    with torch.no_grad():
        ref.embeddings.uniform_(-0.3, 0.3)
        for p in ref.parameters():
            p.copy_(p.bfloat16().float())  # representable in bf16 and (|v| < 2) in fp16
    net = NeRFNetwork(encoding="hashgrid", desired_resolution=32768, bound=1, min_near=SCALE, min_near_lidar=SCALE)
    with torch.no_grad():
        net.encoder.embeddings.copy_(ref.embeddings)
        for a, b in list(zip(net.sigma_net, ref.sigma_net)) + list(zip(net.lidar_color_net, ref.lidar_color_net)):
            a.weight.copy_(b.weight)
    return net.cuda().train(), ref.train()


def _batch():
    from lidarnerf.dataset.rays import get_lidar_rays
    torch.manual_seed(5)
    pose = torch.eye(4)[None]
    pose[0, :3, 3] = torch.tensor([0.02, -0.01, 0.005])
    r = get_lidar_rays(pose, (2.0, 26.9), 66, 1030, N, patch_size=1)
    g = torch.Generator().manual_seed(6)
    rd = (torch.rand(N, generator=g) < 0.85).float()
    gt = torch.stack([rd, torch.rand(N, generator=g), SCALE * (2 + 78 * torch.rand(N, generator=g)) * rd], -1)
    noise, u = torch.rand(N, T, generator=g), torch.rand(N, t, generator=g)
    return r["rays_o"][0].contiguous(), r["rays_d"][0].contiguous(), gt, noise, u


def _step(net, o, d, gt, noise, u, scale):
    """Sum loss (not mean: additive over rays) x loss scale through the fused bf16 chain; returns outputs + gradients."""
    from lidarnerf.nerf.train_step import lidar_loss
    net.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net.render(o.cuda()[None], d.cuda()[None], cal_lidar_color=True, staged=False, perturb=True, num_steps=T,
                         upsample_steps=t, noise=noise, u=u)
        loss, _, _ = lidar_loss(out, gt.cuda()[None])
    (loss * (o.shape[0] * scale)).backward()
    grads = [net.encoder.embeddings.grad.detach().clone()] + \
        [m.weight.grad.detach().clone() for m in list(net.sigma_net) + list(net.lidar_color_net)]
    return {k: v.detach().float().clone() for k, v in out.items()}, float(loss.detach()), grads


def test_config5_batch_bf16_fused_chain():
    from lidarnerf.nerf import fused
    net, ref = _field(51)
    o, d, gt, noise, u = _batch()
    assert fused.supported(net, True, T, t)
    scale = 2.0 ** -6  # sum loss over 16 384 rays: keep the fp16 table gradient inside the fp16 range
    out, loss, grads = _step(net, o, d, gt, noise, u, scale)
    torch.cuda.synchronize()

    # ---- properties
    ws, depth, image = out["weights_sum_lidar"].reshape(-1), out["depth_lidar"].reshape(-1), out["image_lidar"].reshape(-1, 2)
    assert torch.isfinite(ws).all() and torch.isfinite(depth).all() and torch.isfinite(image).all()
    assert all(torch.isfinite(g).all() for g in grads)
    assert float(ws.min()) >= 0 and float(ws.max()) <= 1 + 1e-5
    near, far = SCALE, 81 * SCALE
    assert (depth >= near * ws * (1 - 1e-3) - 1e-7).all() and (depth <= (far + (far - near) / T) * ws + 1e-7).all()
    assert (image >= 0).all() and (image <= ws[:, None] + 1e-5).all()
    assert float(grads[0].abs().sum()) > 0

    # ---- the same step again: bit-identical (4-chunk bucketed backward included)
    out2, loss2, grads2 = _step(net, o, d, gt, noise, u, scale)
    assert loss2 == loss
    for k in out:
        assert torch.equal(out[k], out2[k]), k
    assert torch.equal(grads[0], grads2[0])  # hash table: one integer sum per row and chunk, order-free
    for a, b in zip(grads[1:], grads2[1:]):   # MLP weights: fp32 partial sums of the workgroups meet in device atomics
        assert (a - b).norm() <= 1e-5 * a.norm()

    # ---- 32 rays of the batch against the CPU restatement (same rays, same draws)
    sel = torch.arange(0, N, N // 32)[:32]
    want = render_ref.run_lidar(o[sel], d[sel], ref.density, ref.color, torch.tensor([-1.0, -1, -1, 1, 1, 1]), SCALE, T, t,
                                perturb=True, training=True, noise=noise[sel], u=u[sel])
    for k, tol in (("depth_lidar", 1e-4), ("weights_sum_lidar", 1e-4), ("image_lidar", 1e-3)):  # as the 32-ray bf16 test
        a = out[k].cpu().reshape(N, -1)[sel].reshape(-1)
        b = want[k].detach().reshape(-1)
        err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
        assert err < tol, (k, err)

    # ---- linearity over rays: gradient of the batch = sum of the gradients of its quarters (each quarter is the
    #      4096-ray single-chunk configuration).  16-bit storage on both sides: hash-table rows are rounded to fp16 once
    #      per chunk, so the two sides round different partial sums — agreement in norm, per level
    acc = [torch.zeros_like(g, dtype=torch.float64) for g in grads]
    for q in range(4):
        s = slice(q * N // 4, (q + 1) * N // 4)
        _, _, gq = _step(net, o[s], d[s], gt[s], noise[s], u[s], scale)
        for a, g_ in zip(acc, gq):
            a += g_.double()
    offs = ref.offsets
    for l in range(16):
        a, b = grads[0][offs[l]:offs[l + 1]].double(), acc[0][offs[l]:offs[l + 1]]
        rel = ((a - b).norm() / b.norm()).item()
        assert rel < 2e-3, (l, rel)
    for i in range(1, len(grads)):
        rel = ((grads[i].double() - acc[i]).norm() / acc[i].norm()).item()
        assert rel < 1e-4, (i, rel)  # fp32 accumulation; only the order of the sums differs
