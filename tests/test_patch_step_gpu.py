"""One PATCH-mode training step (the reference's 2 x 8 patch epochs: nerf/utils.py:760-876, 1057-1065; patch rays of
dataset/base_dataset.py:50-70) through the benchmarked chain — LidarTrainer._forward_backward: fused render, the one-launch
patch loss (lnh_lidar_loss_patch) with the loss scale folded into its gradients, backward — against the oracle:
oracle/render_ref.run_lidar + lidar_loss + patch_grad_loss (both pinned to the imported reference: G2, G8).  Compared: the
loss, the hash-table gradient per level, all five weight gradients.  Deterministic (fixed seeds, fixed-order sums
everywhere): the step that tests/test_zzz_learning_gpu.py repeats a few hundred times is checked here once, exactly.

The sample depths the HIP chain drew are handed to the CPU side (fused.CAPTURE), as in
tests/test_parity_fused_gpu.py::test_fused_chain_gradients_with_injected_samples, so every level has to agree tightly."""
import numpy as np
import pytest
import torch

from oracle import render_ref

pytestmark = pytest.mark.gpu
SCALE = 0.010784853507573345


def _patch_batch(N, seed):
    """N rays in patches of 2 x 8; ground truth smooth inside a patch row (neighbours differ by millimetres: below the 0.01 m
    gate of utils.py:789-797, so the structural-gradient term is active on most pairs), a few dropped returns."""
    g = torch.Generator().manual_seed(seed)
    o = (torch.rand(N, 3, generator=g) - 0.5) * 0.1
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    base = (0.1 + 0.6 * torch.rand(N // 16, 1, generator=g)).expand(N // 16, 16).reshape(N)
    depth = base + 0.003 * SCALE * torch.randn(N, generator=g)
    gt = torch.stack([(torch.rand(N, generator=g) > 0.15).float(), torch.rand(N, generator=g), depth], -1)
    return o, d, gt


@pytest.mark.parametrize("N", [512, 4096])
def test_patch_mode_step_vs_oracle(N):
    from test_parity_fused_gpu import _pair_fp16
    from lidarnerf.nerf import fused
    from lidarnerf.nerf.train_step import LidarTrainer
    T, t, px, py = 768, 64, 2, 8
    net, ref = _pair_fp16(seed=51)
    ref.storage = torch.float16
    net.train()
    ref.train()
    o, d, gt = _patch_batch(N, seed=52)
    tr = LidarTrainer(net, lr=1e-2, iters=30000, fp16=True, scale=SCALE, render_kwargs=dict(num_steps=T, upsample_steps=t))
    assert tr.table is not None
    scale = 128.0
    tr.loss_scale.fill_(scale)
    torch.manual_seed(1234)
    fused.CAPTURE = {}
    try:
        loss = tr._forward_backward(o.cuda()[None], d.cuda()[None], gt.cuda()[None], (px, py))
        cap = {k: v.detach().cpu() for k, v in fused.CAPTURE.items()}
    finally:
        fused.CAPTURE = None
    g16 = tr.table._lnh_grad16
    assert g16 is not None and bool(torch.isfinite(g16.float()).all())
    want = render_ref.run_lidar(o, d, ref.density, ref.color, torch.tensor([-1.0, -1, -1, 1, 1, 1]), SCALE, T, t,
                                perturb=True, training=True, z_override=cap["z"], new_z_override=cap["new_z"])
    assert torch.equal(want["z_vals"], cap["z_all"])
    lw = render_ref.lidar_loss(want["depth_lidar"], want["image_lidar"], gt) + \
        render_ref.patch_grad_loss(want["depth_lidar"], gt, px, py, SCALE)
    (lw * scale).backward()
    rel_loss = abs(float(loss.detach()) - float(lw.detach())) / abs(float(lw.detach()))
    offs = ref.offsets
    ge = g16.float().cpu().double().reshape(-1, 2)
    gr = ref.embeddings.grad.double().reshape(-1, 2)
    lvl = [((ge[offs[l]:offs[l + 1]] - gr[offs[l]:offs[l + 1]]).norm() / gr[offs[l]:offs[l + 1]].norm()).item()
           for l in range(16)]
    wts = []
    for a, b in list(zip(net.sigma_net, ref.sigma_net)) + list(zip(net.lidar_color_net, ref.lidar_color_net)):
        ga, gb = a.weight.grad.detach().float().cpu().double(), b.weight.grad.double()
        assert ga.shape == gb.shape
        wts.append(((ga - gb).norm() / gb.norm()).item())
    report = {"loss": rel_loss, "levels": [round(v, 5) for v in lvl], "weights": [round(v, 5) for v in wts]}
    # bounds: those of the injected-sample test of the 1 x 1 step (tests/test_parity_fused_gpu.py) — the patch term adds
    # sign-type gradients of the rendered depths, nothing that conditions the comparison differently
    assert rel_loss < 1e-5, report
    assert max(lvl) < 8e-4, report
    assert max(wts) < 4e-4, report
    # the structural-gradient term is part of what was compared: it is active, and its share of the loss (1.8 % at N = 512 on a
    # freshly initialised field, where the x1000 depth term dominates) is more than ten times the gradient tolerances above —
    # a step without it, or with a wrong one, does not pass them
    lp = float(render_ref.patch_grad_loss(want["depth_lidar"].detach(), gt, px, py, SCALE))
    assert lp > 0.01 * float(lw.detach()), (lp, float(lw.detach()))


def test_patch_step_is_bit_reproducible():
    """The same patch step twice from the same state: the same bits in every gradient (no float atomics anywhere on the
    path: csrc/wgrad.h for the MLP matrices, integer accumulation for the table)."""
    from test_parity_fused_gpu import _pair_fp16
    from lidarnerf.nerf.train_step import LidarTrainer
    net, _ = _pair_fp16(seed=53)
    net.train()
    o, d, gt = _patch_batch(1024, seed=54)
    tr = LidarTrainer(net, lr=1e-2, iters=30000, fp16=True, scale=SCALE, render_kwargs=dict(num_steps=768, upsample_steps=64))
    tr.loss_scale.fill_(256.0)
    runs = []
    for _ in range(3):
        torch.manual_seed(7)
        loss = tr._forward_backward(o.cuda()[None], d.cuda()[None], gt.cuda()[None], (2, 8))
        runs.append([loss.detach().clone(), tr.table._lnh_grad16.clone()] + [p.grad.clone() for p in tr.small if p.grad is not None])
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert torch.equal(a, b)
    assert float(runs[0][1].float().abs().sum()) > 0 and all(float(g.abs().sum()) > 0 for g in runs[0][2:])
    assert len(runs[0]) >= 2 + 5  # loss, table, the five matrices of the LiDAR branch (the camera branch gets no gradient here)
