#!/usr/bin/env python3
"""Micro-benchmark of the colour-head and sigma-net entry points on FIXED synthetic inputs (4096 rays x 832 merged samples;
the bench.py numbers move with the mask fraction of the model being trained, these do not).

    python tools/bench_mlp.py [--rays 4096] [--samples 832] [--active 0.6] [--reps 7]

`--active` = fraction of every ray (its front part) whose compositing weight lies above the colour mask threshold."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lidar-nerf_amd")]
from lidarnerf import _hip  # noqa: E402


def timed(fn, reps):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return min(ts), float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=832)
    ap.add_argument("--active", type=float, default=0.6)
    ap.add_argument("--reps", type=int, default=7)
    a = ap.parse_args()
    _hip.lib()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    N, T = a.rays, a.samples
    h16 = (torch.randn(N * T, 16, device=dev, generator=g) * 0.5).half()
    perm = torch.stack([torch.randperm(T, device=dev, generator=g) for _ in range(64)])[
        torch.randint(0, 64, (N,), device=dev, generator=g)].int().contiguous()
    n_act = int(round(a.active * T))
    weights = torch.full((N, T), 1e-6, device=dev)
    weights[:, :n_act] = 1e-2
    cdir = torch.randn(N, 64, device=dev, generator=g) * 0.3
    wcol = (torch.randn(64 * 16 + 64 * 64 + 16 * 64, device=dev, generator=g) * 0.15).half()
    rgb = torch.empty(N, T, 2, device=dev)
    g_rgb = torch.randn(N, T, 2, device=dev, generator=g) * 1e-3
    g_sigma = torch.randn(N, T, device=dev, generator=g) * 1e-3
    g_h16 = torch.empty(N * T, 16, device=dev, dtype=torch.half)
    g_w = torch.zeros(wcol.numel(), device=dev)
    ray_sum = torch.empty(N, 64, device=dev)

    def fwd():
        _hip.call("lnh_lidar_color_forward", h16.data_ptr(), perm.data_ptr(), weights.data_ptr(), cdir.data_ptr(),
                  wcol.data_ptr(), N, T, rgb.data_ptr())

    def bwd():
        _hip.call("lnh_lidar_color_backward", g_rgb.data_ptr(), g_sigma.data_ptr(), h16.data_ptr(), perm.data_ptr(),
                  weights.data_ptr(), cdir.data_ptr(), wcol.data_ptr(), N, T, g_h16.data_ptr(), g_w.data_ptr(),
                  ray_sum.data_ptr(), *_hip.wgrad_ws(dev))

    for name, fn in (("lnh_lidar_color_forward", fwd), ("lnh_lidar_color_backward", bwd)):
        fn()
        torch.cuda.synchronize()
        lo, med = timed(fn, a.reps)
        print(f"{name:28s} active {a.active:.2f}  min {lo:8.1f} us  median {med:8.1f} us")
    print("checksum", float(g_h16.float().abs().sum()), float(g_w.abs().sum()), float(ray_sum.abs().sum()))

    # fused forward tail (merge + weights + colour + compositing, one wave per ray): densities chosen so that the first
    # `active` fraction of every ray carries weights above the mask threshold and a wall absorbs the rest
    z = (torch.arange(T, device=dev, dtype=torch.float32) + 0.5)[None, :].expand(N, T).contiguous() * (0.8 / T) + 0.01
    delta = 0.8 / T
    sig_m = torch.full((N, T), -np.log(1 - 2e-3) / delta, device=dev)
    if n_act < T:
        sig_m[:, n_act:] = 1e9
    if n_act == 0:
        sig_m[:] = 0.0
    sigma_pt = torch.empty_like(sig_m)
    sigma_pt.scatter_(1, perm.long(), sig_m)  # point order: sigma_pt[perm[i]] = merged[i]
    sd = torch.full((N,), delta, device=dev)
    o_sig, o_w = torch.empty(N, T, device=dev), torch.empty(N, T, device=dev)
    o_ws, o_dp, o_im = torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, 2, device=dev)

    def tail():
        _hip.call("lnh_lidar_color_composite_forward", z.data_ptr(), sigma_pt.data_ptr(), perm.data_ptr(), sd.data_ptr(),
                  h16.data_ptr(), cdir.data_ptr(), wcol.data_ptr(), N, T, 1.0, o_sig.data_ptr(), o_w.data_ptr(),
                  rgb.data_ptr(), o_ws.data_ptr(), o_dp.data_ptr(), o_im.data_ptr())

    tail()
    torch.cuda.synchronize()
    lo, med = timed(tail, a.reps)
    print(f"{'lnh_lidar_color_composite_forward':28s} active {float((o_w > 1e-4).float().mean()):.2f}  min {lo:8.1f} us  median {med:8.1f} us")

    # sigma net on the coarse pass of the same batch: [16, N*T, 2] level-major features -> 16-wide rows
    Tc, B_all = T - 64, N * T
    B = N * Tc
    feat = (torch.randn(16, B_all, 2, device=dev, generator=g) * 0.3).half()
    wsig = (torch.randn(64 * 32 + 16 * 64, device=dev, generator=g) * 0.2).half()
    sigma = torch.empty(N * T, device=dev)
    g_feat = torch.empty(16, B_all, 2, device=dev, dtype=torch.half)
    g_ws = torch.zeros(wsig.numel(), device=dev)

    def dfwd():
        _hip.call("lnh_density_mlp_forward", feat.data_ptr(), wsig.data_ptr(), B, Tc, T, 0, B_all, h16.data_ptr(),
                  sigma.data_ptr())

    def dbwd():
        _hip.call("lnh_density_mlp_backward", g_h16.data_ptr(), feat.data_ptr(), wsig.data_ptr(), B_all, T, T, 0,
                  g_feat.data_ptr(), g_ws.data_ptr(), *_hip.wgrad_ws(dev))

    for name, fn, pts in (("lnh_density_mlp_forward", dfwd, B), ("lnh_density_mlp_backward", dbwd, B_all)):
        fn()
        torch.cuda.synchronize()
        lo, med = timed(fn, a.reps)
        print(f"{name:28s} {pts / 1e6:.2f} M points  min {lo:8.1f} us  median {med:8.1f} us")


if __name__ == "__main__":
    main()
