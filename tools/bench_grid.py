#!/usr/bin/env python3
"""Micro-benchmark of the hash-grid entry points at the BASELINE shape (4096 rays x 832 samples, fp16 tables, LiDAR ray
geometry: rays leave one sensor position, samples 1 m .. 81 m in scene units, stratified jitter).

    python tools/bench_grid.py [--rays 4096] [--per-level] [--reps 5]

Prints HIP-event times (min over reps) of the whole forward / backward and, with --per-level, of every level on its own
(backward: lnh_grid_encode_backward_ws_levels(l, l+1); forward: a one-level call on the level's table slice).  Run it
under `rocprofv3 --kernel-trace` to split the backward into its scatter / reduce / finalize kernels."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lidar-nerf_amd")]
from lidarnerf import _hip  # noqa: E402
from lidarnerf.gridencoder.grid import level_offsets  # noqa: E402

SCALE = 0.010784853507573345


def lidar_points(n_rays, T, device="cuda", seed=0):
    """[n_rays*T, 3] in [0,1]^3: KITTI-360-shaped rays (fov_up 2 deg, fov 26.9 deg, full azimuth) from near the origin."""
    g = torch.Generator(device=device).manual_seed(seed)
    o = (torch.rand(n_rays, 1, 3, device=device, generator=g) - 0.5) * 0.02
    beta = (torch.rand(n_rays, device=device, generator=g) - 0.5) * 2 * np.pi
    alpha = (2.0 - torch.rand(n_rays, device=device, generator=g) * 26.9) / 180 * np.pi
    d = torch.stack([alpha.cos() * beta.cos(), alpha.cos() * beta.sin(), alpha.sin()], -1)[:, None, :]
    z = (torch.linspace(0, 1, T, device=device) * 80 * SCALE + SCALE)[None, :, None]
    z = z + (torch.rand(n_rays, T, 1, device=device, generator=g) - 0.5) * (80 * SCALE / T)
    x = ((o + d * z).clamp(-1, 1) + 1) / 2
    return x.reshape(-1, 3).contiguous()


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
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--per-level", action="store_true")
    ap.add_argument("--skip-fwd", action="store_true")
    ap.add_argument("--skip-bwd", action="store_true")
    ap.add_argument("--zero-sprinkle", type=float, default=0.0,
                    help="fraction of the samples (drawn independently) whose upstream gradient is exactly zero on every level")
    ap.add_argument("--zero-tail", type=float, default=0.0,
                    help="fraction of every ray (its far end: samples behind a surface) whose upstream gradient is exactly zero")
    a = ap.parse_args()
    lib = _hip.lib()
    x = lidar_points(a.rays, a.samples)
    B = x.shape[0]
    Lv = 16
    pls = np.exp2(np.log2(32768 / 16) / 15)
    S = float(np.log2(pls))
    off = torch.from_numpy(level_offsets(3, Lv, pls, 16, 19, False))
    rows = int(off[-1])
    tab = ((torch.rand(rows, 2, device="cuda") - 0.5) * 2e-4).half()
    out = torch.empty(Lv, B, 2, dtype=torch.half, device="cuda")
    g = (torch.randn(Lv, B, 2, device="cuda") * 0.01).half()
    if a.zero_sprinkle > 0 or a.zero_tail > 0:  # what a trained field hands the backward (profiles/r05_reduce_drift.txt)
        gen = torch.Generator(device="cuda").manual_seed(7)
        keep = torch.rand(a.rays, a.samples, device="cuda", generator=gen) >= a.zero_sprinkle
        keep[:, a.samples - int(round(a.zero_tail * a.samples)):] = False
        g = g * keep.reshape(1, B, 1).half()
        print(f"zero-gradient samples: {1 - float(keep.float().mean()):.3f}")
    ge = torch.zeros(rows, 2, dtype=torch.half, device="cuda")
    need = lib.lnh_grid_backward_workspace_size(off.data_ptr(), B, 3, 2, Lv, S, 16, 0, 0, 1)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    print(f"B = {B} points, workspace {need / 1e9:.2f} GB")

    def fwd():
        _hip.call("lnh_grid_encode_forward", x.data_ptr(), tab.data_ptr(), off.data_ptr(), out.data_ptr(), B, 3, 2, Lv, S, 16,
                  None, 0, 0, 0, 1)

    def bwd(l0=0, l1=Lv):
        _hip.call("lnh_grid_encode_backward_ws_levels", g.data_ptr(), x.data_ptr(), off.data_ptr(), ge.data_ptr(), B, 3, 2,
                  Lv, S, 16, 0, 0, 0, 1, ws.data_ptr(), need, l0, l1)

    if not a.skip_fwd:
        fwd()
        mn, md = timed(fwd, a.reps)
        print(f"forward  all levels: min {mn:8.1f} us  median {md:8.1f} us   {588 * B / mn / 1e3:7.1f} GB/s algorithmic")
    if not a.skip_bwd:
        bwd()
        mn, md = timed(bwd, a.reps)
        print(f"backward all levels: min {mn:8.1f} us  median {md:8.1f} us   {1100 * B / mn / 1e3:7.1f} GB/s algorithmic")
    if a.per_level:
        for l in range(Lv):
            line = f"level {l:2d}:"
            if not a.skip_fwd:
                # one-level forward on the level's table slice: H' = round(scale + 1) reproduces the level's resolution
                scale = float(np.exp2(l * S) * 16 - 1)
                sub = torch.tensor([0, int(off[l + 1] - off[l])], dtype=torch.int32)
                t_l = tab[int(off[l]):int(off[l + 1])]

                def f1():
                    _hip.call("lnh_grid_encode_forward", x.data_ptr(), t_l.data_ptr(), sub.data_ptr(), out.data_ptr(), B, 3, 2,
                              1, 0.0, int(round(scale + 1)), None, 0, 0, 0, 1)
                f1()
                line += f"  fwd {timed(f1, a.reps)[0]:7.1f} us"
            if not a.skip_bwd:
                bwd(l, l + 1)
                line += f"  bwd {timed(lambda: bwd(l, l + 1), a.reps)[0]:7.1f} us"
            print(line)


if __name__ == "__main__":
    main()
