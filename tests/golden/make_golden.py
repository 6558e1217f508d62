#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference's pure-PyTorch pieces (CPU).

Runs only in the build container (needs /root/reference); the reference's Python never ships to the GPU box —
only the .npz data written next to this script does.  Re-run:  python tests/golden/make_golden.py

Vectors:
  G1 sample_pdf        lidarnerf/nerf/renderer.py:10-46      det=True and det=False (u recovered by re-seeding)
  G2 NeRFRenderer.run  lidarnerf/nerf/renderer.py:99-298     LiDAR mode, analytic stub density/colour,
                       eval (det, no perturb) and train (perturb, random u) incl. gradients w.r.t. stub params
  G3 get_lidar_rays    lidarnerf/dataset/base_dataset.py:16-105  full 66x1030 grid (subsampled) + patch indices
  G4 FreqEncoder       lidarnerf/encoding.py:6-47            pure-torch fwd/bwd, degree 12
  G5 trunc_exp         lidarnerf/activation.py:6-20          fwd/bwd incl. the clamp region
  G6 convert           lidarnerf/convert.py:99-160, 194-237  lidar_to_pano_with_intensities (per-point loop) on a
                       synthetic 20 k-point sweep incl. ties, out-of-range and beyond-max-depth points, and
                       pano_to_lidar_with_intensities of the result
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault("trimesh", types.ModuleType("trimesh"))  # visualisation-only import in the reference
sys.path.insert(0, REF)

from lidarnerf.activation import trunc_exp  # noqa: E402
from lidarnerf.dataset.base_dataset import get_lidar_rays  # noqa: E402
from lidarnerf.encoding import FreqEncoder  # noqa: E402
from lidarnerf.nerf.renderer import NeRFRenderer, sample_pdf  # noqa: E402

torch.set_num_threads(4)
SCALE = 0.010784853507573345  # configs/kitti360_1908.txt:12


def g1():
    torch.manual_seed(11)
    B, T, n = 37, 767, 64
    bins = torch.sort(torch.rand(B, T) * 0.8 + 0.01, dim=-1)[0]
    w = torch.rand(B, T - 1) ** 6  # peaky
    w[3] = 0.0  # all-zero weights row (exercises the +1e-5 / denom<1e-5 branches)
    w[5, :] = 0
    w[5, 100] = 1.0
    det = sample_pdf(bins, w, n, det=True)
    torch.manual_seed(123)
    rnd = sample_pdf(bins, w, n, det=False)
    torch.manual_seed(123)
    u = torch.rand(B, n)
    np.savez_compressed(os.path.join(OUT, "g1_sample_pdf.npz"), bins=bins.numpy(), weights=w.numpy(),
                        det=det.numpy(), rnd=rnd.numpy(), u=u.numpy())


class StubField(NeRFRenderer):
    """Analytic density/colour so any implementation can recompute the inputs of the renderer."""

    def __init__(self, **kw):
        super().__init__(**kw)
        g = torch.Generator().manual_seed(5)
        self.a0 = torch.nn.Parameter(torch.tensor(2.5))
        self.a = torch.nn.Parameter(torch.tensor([3.0, -2.0, 1.0]))
        self.bump = torch.nn.Parameter(torch.tensor([0.35, 0.03, 400.0]))  # radius, width, height
        self.M = torch.nn.Parameter(torch.randn(15, 3, generator=g))
        self.Wc = torch.nn.Parameter(torch.randn(2, 18, generator=g) * 0.7)
        self.out_color_dim = 3
        self.out_lidar_color_dim = 2

    def density(self, x):
        r = x.norm(dim=-1)
        sigma = torch.exp(self.a0 + (x * self.a).sum(-1)) + self.bump[2] * torch.exp(
            -((r - self.bump[0]) / self.bump[1]) ** 2)
        return {"sigma": sigma, "geo_feat": torch.tanh(x @ self.M.t() * 3.0)}

    def color(self, x, d, cal_lidar_color=False, mask=None, geo_feat=None, **kw):
        rgbs = torch.zeros(mask.shape[0], self.out_dim, dtype=x.dtype)
        if not mask.any():
            return rgbs
        h = torch.sigmoid(torch.cat([d[mask], geo_feat[mask]], -1) @ self.Wc.t())
        rgbs[mask] = h
        return rgbs


def make_rays(N, seed):
    g = torch.Generator().manual_seed(seed)
    pose = torch.eye(4).unsqueeze(0)
    pose[0, :3, 3] = torch.tensor([0.05, -0.02, 0.01])
    torch.manual_seed(seed)
    r = get_lidar_rays(pose, (2.0, 26.9), 66, 1030, N, patch_size=1)
    return r["rays_o"].contiguous(), r["rays_d"].contiguous()


def g2():
    out = {}
    for tag, train in (("eval", False), ("train", True)):
        m = StubField(bound=1, min_near=SCALE, min_near_lidar=SCALE)
        m.train(train)
        N, T, t = 48, 768, 64
        rays_o, rays_d = make_rays(N, 7)
        torch.manual_seed(99)
        res = m.render(rays_o, rays_d, cal_lidar_color=True, staged=False, perturb=train, num_steps=T,
                       upsample_steps=t)
        cd, ci, cw = torch.linspace(0.5, 1.5, N), torch.linspace(-1, 1, N * 2).view(1, N, 2), torch.linspace(1, 0.2, N)
        loss = (res["depth_lidar"] * cd).sum() + (res["image_lidar"] * ci).sum() + (res["weights_sum_lidar"] * cw).sum()
        loss.backward()
        out.update({f"{tag}_depth": res["depth_lidar"].detach().numpy(),
                    f"{tag}_image": res["image_lidar"].detach().numpy(),
                    f"{tag}_ws": res["weights_sum_lidar"].detach().numpy(),
                    f"{tag}_loss": loss.detach().numpy()})
        for name in ("a0", "a", "bump", "M", "Wc"):
            out[f"{tag}_grad_{name}"] = getattr(m, name).grad.numpy().copy()
        if train:  # replay the two torch.rand draws made inside run(): perturb noise, then sample_pdf's u
            torch.manual_seed(99)
            out["train_noise"] = torch.rand(N, T).numpy()
            out["train_u"] = torch.rand(N, t).numpy()
    out["rays_o"], out["rays_d"] = rays_o.numpy(), rays_d.numpy()
    for name in ("a0", "a", "bump", "M", "Wc"):
        out[f"param_{name}"] = getattr(m, name).detach().numpy()
    out["cd"], out["ci"], out["cw"] = cd.numpy(), ci.numpy(), cw.numpy()
    np.savez_compressed(os.path.join(OUT, "g2_renderer_run.npz"), **out)


def g3():
    pose = torch.eye(4).unsqueeze(0)
    th = np.deg2rad(12.5)
    pose[0, :3, :3] = torch.tensor([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    pose[0, :3, 3] = torch.tensor([0.11, -0.07, 0.02])
    full = get_lidar_rays(pose, (2.0, 26.9), 66, 1030, -1)
    sel = np.unique(np.concatenate([np.arange(0, 66 * 1030, 37), [0, 1029, 65 * 1030, 66 * 1030 - 1]]))
    out = {"pose": pose.numpy(), "sel": sel, "rays_o": full["rays_o"][0, sel].numpy(),
           "rays_d": full["rays_d"][0, sel].numpy()}
    for tag, ps in (("p1", 1), ("p28", [2, 8])):
        torch.manual_seed(1234)
        r = get_lidar_rays(pose, (2.0, 26.9), 66, 1030, 4096, patch_size=ps)
        out[f"inds_{tag}"] = r["inds"][0].numpy()
        out[f"rays_d_{tag}"] = r["rays_d"][0, :64].numpy()
    # NeRF-MVL shaped variant (configs/nerf_mvl.txt; generate_train_rangeview.py:166-168)
    mv = get_lidar_rays(pose, (15.0, 40.0), 256, 1800, -1)
    sel2 = np.arange(0, 256 * 1800, 997)
    out["mvl_sel"], out["mvl_rays_d"] = sel2, mv["rays_d"][0, sel2].numpy()
    np.savez_compressed(os.path.join(OUT, "g3_lidar_rays.npz"), **out)


def g4():
    torch.manual_seed(3)
    d = torch.nn.functional.normalize(torch.randn(257, 3), dim=-1)
    d[0] = torch.tensor([1.0, 0.0, 0.0])
    d[1] = torch.tensor([0.0, -1.0, 0.0])
    d.requires_grad_(True)
    enc = FreqEncoder(input_dim=3, max_freq_log2=11, N_freqs=12, log_sampling=True)
    y = enc(d)
    g = torch.randn(y.shape)
    y.backward(g)
    np.savez_compressed(os.path.join(OUT, "g4_freq_encoder.npz"), d=d.detach().numpy(), y=y.detach().numpy(),
                        g=g.numpy(), gd=d.grad.numpy())


def g5():
    x = torch.tensor([-30.0, -15.5, -15.0, -3.0, 0.0, 0.7, 5.0, 14.9, 15.0, 15.5, 20.0, 40.0], requires_grad=True)
    y = trunc_exp(x)
    g = torch.linspace(0.5, 2.0, x.numel())
    y.backward(g)
    np.savez_compressed(os.path.join(OUT, "g5_trunc_exp.npz"), x=x.detach().numpy(), y=y.detach().numpy(),
                        g=g.numpy(), gx=x.grad.numpy())


def g6():
    from lidarnerf.convert import lidar_to_pano_with_intensities, pano_to_lidar_with_intensities
    rng = np.random.default_rng(7)
    H, W, K = 66, 1030, (2.0, 26.9)
    n = 20000
    az = rng.uniform(-np.pi, np.pi, n)
    el = np.deg2rad(rng.uniform(-30.0, 6.0, n))          # some rays outside the 26.9 deg field of view
    d = rng.uniform(0.5, 95.0, n)                          # some beyond max_depth = 80
    pts = np.stack([d * np.cos(el) * np.cos(az), d * np.cos(el) * np.sin(az), d * np.sin(el),
                    rng.uniform(0, 1, n)], 1).astype(np.float32)
    pts[100:200] = pts[0:100]                              # exact duplicates: the first one must win
    pts[100:200, 3] += 1.0
    pts[300:400, :3] = pts[200:300, :3] * np.float32(0.5)  # same pixel, nearer: the later one must win
    pano, inten = lidar_to_pano_with_intensities(pts, H, W, K, max_depth=80)
    back = pano_to_lidar_with_intensities(pano.astype(np.float32), inten.astype(np.float32), K)
    np.savez_compressed(os.path.join(OUT, "g6_convert.npz"), pts=pts, H=H, W=W, K=np.array(K), pano=pano,
                        intensities=inten, back=back)


if __name__ == "__main__":
    g1(); g2(); g3(); g4(); g5(); g6()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
