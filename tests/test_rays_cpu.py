"""G3: restated get_lidar_rays vs the imported reference's outputs (pure torch, runs on CPU)."""
import os

import numpy as np
import torch

from lidarnerf.dataset.rays import get_lidar_rays


def test_full_grid_and_patches(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_lidar_rays.npz"))
    pose = torch.from_numpy(g["pose"])
    full = get_lidar_rays(pose, (2.0, 26.9), 66, 1030, -1)
    np.testing.assert_allclose(full["rays_d"][0, g["sel"]].numpy(), g["rays_d"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(full["rays_o"][0, g["sel"]].numpy(), g["rays_o"])
    for tag, ps in (("p1", 1), ("p28", [2, 8])):
        torch.manual_seed(1234)
        r = get_lidar_rays(pose, (2.0, 26.9), 66, 1030, 4096, patch_size=ps)
        np.testing.assert_array_equal(r["inds"][0].numpy(), g[f"inds_{tag}"])
        np.testing.assert_allclose(r["rays_d"][0, :64].numpy(), g[f"rays_d_{tag}"], rtol=0, atol=1e-6)
    # patch size 1 never samples the last row / column
    assert g["inds_p1"].max() < 65 * 1030 and (g["inds_p1"] % 1030).max() < 1029
    mv = get_lidar_rays(pose, (15.0, 40.0), 256, 1800, -1)
    np.testing.assert_allclose(mv["rays_d"][0, g["mvl_sel"]].numpy(), g["mvl_rays_d"], rtol=0, atol=1e-6)
