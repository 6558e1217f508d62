"""End-to-end learning checks of the benchmarked path (fused fp16 chain + fused optimizer + loss scaling) on the analytic scene
of bench.py (`--table trained`: sensor inside a 32 m sphere over a ground plane, consistent between frames): the depth error
on a HELD-OUT ray set must fall from ~11 m at initialisation to well below a metre — with 1x1 rays and with the reference's
2x8 patch epochs (structural-gradient term, nerf/utils.py:760-876, 1057-1065; patch rays of dataset/base_dataset.py:50-70).
A path that computes plausible numbers but wrong gradients does not pass this.

These are the LAST tests of the suite (file name), behind every deterministic parity test.  What they train with is the
reference's recipe (Adam, fp16 gradients behind a dynamic loss scale with torch's growth interval of 2000 steps) at HALF its
learning rate: 5e-3.  Measured in round 6 (profiles/r06_learning_distribution.md, ~300 runs): at the reference's 1e-2 this
synthetic scene sits at the edge of stability — every run is at 0.59 .. 0.69 m after 100 steps, then the loss spikes, the
loss scale halves at every overflow, and a third to a half of the runs (whatever the seed, the optimizer — torch.optim.Adam +
GradScaler as well — or the tree, round 4's included) wander at 0.5 .. 5 m for hundreds of steps; round 5's gate went red on
such a run (3.89 m against a bound of 2.5).  At 5e-3 all 22 measured runs (12 seeds 1 x 1, 10 seeds 2 x 8) lie within
0.15 .. 0.19 m at step 400 with the loss scale untouched at 4096.  Since round 6 the training path has no float atomics left
(tests/test_determinism_gpu.py): a seed gives ONE trajectory on every box, so the figures asserted below are those of seed 0,
with a factor two of room to the whole measured band."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _train(patch, steps, every):
    import bench
    from lidarnerf.nerf.train_step import LidarTrainer
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = bench.build_model(dev)
    tr = LidarTrainer(model, lr=5e-3, iters=30000, fp16=True, scale=bench.SCALE,
                      render_kwargs=dict(num_steps=768, upsample_steps=64))
    poses = bench.synthetic_frames(60, dev)
    batches = [bench.make_batch(poses, s, 4096, 0, dev, patch, "analytic") for s in range(60)]
    held = bench.make_batch(poses, 30, 4096, 1, dev, (1, 1), "analytic")
    torch.manual_seed(0)

    def depth_error_m():
        model.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            out = model.render(held[0], held[1], cal_lidar_color=True, staged=False, perturb=False, num_steps=768,
                               upsample_steps=64)
        model.train()
        return float(((out["depth_lidar"][0].float() - held[2][0, :, 2]).abs() / bench.SCALE).median())

    errs = {0: depth_error_m()}
    losses = []
    for s in range(steps):
        losses.append(tr.step(*batches[s % 60], **({} if patch == (1, 1) else {"patch": patch})).detach())
        if (s + 1) % every == 0:
            errs[s + 1] = depth_error_m()
    return errs, torch.stack(losses).float().cpu().numpy(), tr


def test_dense_path_learns_the_analytic_scene():
    errs, losses, tr = _train((1, 1), 400, 100)
    assert np.isfinite(losses).all()
    # measured on MI355X, 12 seeds: 11.6 m -> 0.47 .. 0.49 m after 100 steps, 0.15 .. 0.17 m after 400
    assert errs[0] > 5.0 and errs[100] < 1.0 and errs[400] < 0.4, errs
    # the dynamic loss scale stays where the fp16 table gradient fits (measured: 256 .. 4096 after 800 steps)
    assert float(tr.loss_scale) >= 32.0


def test_patch_mode_step_learns_too():
    errs, losses, tr = _train((2, 8), 400, 100)
    assert np.isfinite(losses).all()
    # 256 patches of 16 neighbouring rays per step + the structural-gradient term; 10 seeds: 0.35 .. 0.46 m after 100 steps,
    # 0.15 .. 0.19 m after 400 (not monotonic in between: the 60 frames cycle, 0.82 .. 0.95 m at step 300)
    assert errs[0] > 5.0 and errs[100] < 1.0 and errs[400] < 0.4, errs
    assert float(tr.loss_scale) >= 32.0


def test_patch_gradient_term_matches_the_restatement():
    """The GPU-side torch ops of the patch term (train_step.patch_gradient_loss: the fallback for tensors the one-launch
    kernel does not take; the step above goes through lnh_lidar_loss_patch, tests/test_lidar_field_gpu.py,
    tests/test_g8_train_step_gpu.py and tests/test_patch_step_gpu.py) against oracle/render_ref.patch_grad_loss on the same
    depths; both are pinned to the reference's own Trainer.train_step by G8 (tests/test_oracle_golden.py)."""
    from lidarnerf.nerf.train_step import patch_gradient_loss
    from oracle import render_ref
    g = torch.Generator().manual_seed(1)
    scale = 0.010784853507573345
    n = 256 * 16
    # ground truth: smooth inside a patch (neighbours differ by millimetres: below the 0.01 m gate of utils.py:789-797),
    # prediction: the same surface + 5 cm of noise
    base = (torch.rand(256, 1, generator=g) * 0.6).expand(256, 16).reshape(n)
    gt_depth = base + 0.003 * scale * torch.randn(n, generator=g)
    depth = gt_depth + 0.05 * scale * torch.randn(n, generator=g)
    gt = torch.stack([(torch.rand(n, generator=g) > 0.15).float(), torch.rand(n, generator=g), gt_depth], -1)
    rd = gt[:, 0]
    want = render_ref.patch_grad_loss(depth, gt, 2, 8, scale)
    got = patch_gradient_loss((depth * rd).cuda(), (gt[:, 2] * rd).cuda(), rd.cuda(), 2, 8, scale)
    assert abs(float(got) - float(want)) <= 1e-5 * abs(float(want)) and float(want) > 0
