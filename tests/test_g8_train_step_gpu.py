"""G8 on the GPU: the one-launch loss kernels (lnh_lidar_loss, lnh_lidar_loss_patch — what LidarTrainer.train_step runs) against
the loss and the gradients the reference's OWN Trainer.train_step produced (lidarnerf/nerf/utils.py:697-884, called unbound
on a stub self in the build container: tests/golden/make_golden.py g8 -> tests/golden/g8_train_step.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,patch", [("p1", None), ("p2x8", (2, 8)), ("p4x4", (4, 4))])
def test_loss_kernels_reproduce_the_reference_train_step(golden_dir, tag, patch):
    from lidarnerf.nerf.train_step import fused_lidar_loss
    g = np.load(os.path.join(golden_dir, "g8_train_step.npz"))
    ad, ar, ai, ag = (float(v) for v in g["alphas"])
    scale = float(g["scale"])
    gt = torch.from_numpy(g["gt"]).cuda()
    depth = torch.from_numpy(g["depth"]).cuda()[None].requires_grad_(True)
    image = torch.from_numpy(g["image"]).cuda()[None].requires_grad_(True)
    loss = fused_lidar_loss({"depth_lidar": depth, "image_lidar": image}, gt, ad, ar, ai,
                            patch=None if patch is None else (patch[0], patch[1], scale, ag))
    (loss * 4.0).backward()                      # an upstream factor (the loss scale of the training step)
    want = float(g[f"{tag}_loss"])
    # fp32: the kernel sums 512 per-ray terms in another order than torch's mean
    assert abs(float(loss) - want) <= 3e-6 * abs(want), (float(loss), want)
    gd, gi = g[f"{tag}_grad_depth"], g[f"{tag}_grad_image"]
    np.testing.assert_allclose(depth.grad[0].cpu().numpy() / 4.0, gd, rtol=2e-5, atol=2e-6 * np.abs(gd).max())
    np.testing.assert_allclose(image.grad[0].cpu().numpy() / 4.0, gi, rtol=2e-5, atol=1e-9)
    # the trainer's form: the kernel multiplies the gradients by a device scalar (the loss scale) and backward() starts
    # from ONE — the same numbers, the loss itself unscaled, twice in a row (the kernel leaves no accumulator behind)
    for rep in range(2):
        depth.grad = image.grad = None
        sc = torch.full((), 4.0, device="cuda")
        loss2 = fused_lidar_loss({"depth_lidar": depth, "image_lidar": image}, gt, ad, ar, ai,
                                 patch=None if patch is None else (patch[0], patch[1], scale, ag), grad_scale=sc)
        loss2.backward(gradient=torch.ones((), device="cuda"))
        assert float(loss2) == float(loss)
        np.testing.assert_allclose(depth.grad[0].cpu().numpy() / 4.0, gd, rtol=2e-5, atol=2e-6 * np.abs(gd).max())
        np.testing.assert_allclose(image.grad[0].cpu().numpy() / 4.0, gi, rtol=2e-5, atol=1e-9)


def test_trainer_train_step_loss_is_the_reference_loss(golden_dir):
    """LidarTrainer.loss_of (the dispatch between the dense step and the patch epochs) on the G8 inputs."""
    from lidarnerf.nerf import train_step
    g = np.load(os.path.join(golden_dir, "g8_train_step.npz"))
    gt = torch.from_numpy(g["gt"]).cuda()
    out = {"depth_lidar": torch.from_numpy(g["depth"]).cuda()[None], "image_lidar": torch.from_numpy(g["image"]).cuda()[None]}
    for tag, patch in (("p1", None), ("p2x8", (2, 8, float(g["scale"]), 100.0))):
        loss = train_step.fused_lidar_loss(out, gt, 1000.0, 1.0, 10.0, patch=patch)
        assert abs(float(loss) - float(g[f"{tag}_loss"])) <= 3e-6 * float(g[f"{tag}_loss"])
