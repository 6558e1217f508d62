#!/usr/bin/env python3
"""bench.py — train rays/s of the LiDAR-NeRF hot path (encode + MLP + composite + backward + optimizer step).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): KITTI-360 seq-1908-shaped synthetic data — 66x1030 range image,
(fov_up, fov) = (2.0, 26.9) deg, 4096 rays per step per GPU, 768 coarse + 64 importance samples per ray,
hash grid L=16 F=2 (2^19 rows/level, finest 32768) + 64-wide fused MLPs, fp16 tables / fp16 MFMA MLPs with fp32
accumulation (the reference's documented `--fp16` mode), Adam + GradScaler.  One step = one pass of the hot path over
one batch of rays: render forward, LiDAR loss, backward, (DP: gradient all-reduce), optimizer step.

Prints ONE JSON line on rank 0 (see the contract in the task description): metric/value/unit, roofline of the
dominant kernel (algorithmic bytes / HIP-event duration measured inside the timed region), and a CPU baseline (the
oracle restatement of the same pipeline, timed on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lidar-nerf_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SCALE = 0.010784853507573345  # configs/kitti360_1908.txt:12
H_IMG, W_IMG, INTRINSICS = 66, 1030, (2.0, 26.9)
NUM_STEPS, UPSAMPLE = 768, 64  # configs/kitti360_1908.txt:9-10
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)

# algorithmic bytes per sample point, fp16 tables, L=16, F=2, D=3
GRID_FWD_BYTES = 12 + 16 * 8 * 2 * 2 + 16 * 2 * 2  # 588
GRID_BWD_BYTES = 12 + 16 * 2 * 2 + 2 * (16 * 8 * 2 * 2)  # 1100


def synthetic_frames(n_frames, device):
    """60 poses on a straight ~1 m/frame trajectory with a small yaw."""
    poses = torch.eye(4).repeat(n_frames, 1, 1)
    for k in range(n_frames):
        th = np.deg2rad(0.5 * k)
        poses[k, :3, :3] = torch.tensor([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
        poses[k, :3, 3] = torch.tensor([(k - n_frames / 2) * SCALE, 0.0, 0.0])
    return poses.to(device)


def make_batch(poses, step, n_rays, rank, device):
    """One frame, n_rays pixels (patch size 1), synthetic ground truth (raydrop, intensity, depth*scale)."""
    from lidarnerf.dataset.rays import get_lidar_rays
    g = torch.Generator(device="cpu").manual_seed(1234 + step * 131 + rank * 7919)
    torch.manual_seed(1234 + step * 131 + rank * 7919)
    pose = poses[step % poses.shape[0]][None]
    r = get_lidar_rays(pose, INTRINSICS, H_IMG, W_IMG, n_rays, patch_size=1)
    raydrop = (torch.rand(n_rays, generator=g) < 0.85).float()
    intensity = torch.rand(n_rays, generator=g)
    depth = SCALE * (2 + 78 * torch.rand(n_rays, generator=g)) * raydrop
    gt = torch.stack([raydrop, intensity, depth], -1)[None].to(device)
    return r["rays_o"].contiguous(), r["rays_d"].contiguous(), gt


def build_model(device):
    from lidarnerf.nerf.network import NeRFNetwork
    torch.manual_seed(0)
    model = NeRFNetwork(encoding="hashgrid", desired_resolution=32768, log2_hashmap_size=19, num_layers=2,
                        hidden_dim=64, geo_feat_dim=15, bound=1, density_scale=1, min_near=SCALE,
                        min_near_lidar=SCALE, density_thresh=10, bg_radius=-1)
    return model.to(device).train()


def cpu_baseline(budget_s=12.0):
    """Oracle restatement of the same pipeline (hash grid via the scalar C oracle, MLPs/compositing in torch fp32)
    on the host cores; bounded sample: batches of 64 rays x 832 samples until ~budget_s of CPU work."""
    from oracle import render_ref
    threads = torch.get_num_threads()
    torch.manual_seed(0)
    ref = render_ref.RefLidarField(desired_resolution=32768).train()
    n = 64
    g = torch.Generator().manual_seed(1)
    o = (torch.rand(n, 3, generator=g) - 0.5) * 0.02
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    gt = torch.rand(n, 3, generator=g)
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    steps, t0 = 0, time.perf_counter()
    first = None
    while True:
        ref.zero_grad(set_to_none=True)
        res = render_ref.run_lidar(o, d, ref.density, ref.color, aabb, SCALE, NUM_STEPS, UPSAMPLE, perturb=True,
                                   training=True)
        loss = render_ref.lidar_loss(res["depth_lidar"], res["image_lidar"], gt)
        loss.backward()
        steps += 1
        if first is None:  # first step = warm-up (page-in of the 55 MB table, thread pools)
            first = time.perf_counter()
            steps = 0
            t0 = first
        if steps >= 2 and time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n * steps / dt, 2), "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"{steps} steps x {n} rays x {NUM_STEPS + UPSAMPLE} samples, fwd+bwd, fp32, oracle/render_ref.py "
                      f"(hash grid in scalar C, MLP/compositing torch CPU with {threads} threads); "
                      f"{os.cpu_count()} host cores present"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=4096, help="rays per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eval", action="store_true", help="skip the secondary full-frame evaluation measurement")
    ap.add_argument("--kernel-timers", action="store_true", help="HIP-event timing of every C-ABI call (adds ~4 %)")
    args = ap.parse_args()

    from lidarnerf import _hip, parallel
    from lidarnerf.nerf.train_step import LidarTrainer

    rank, local, world = parallel.init_from_env()
    if world != args.gpus and rank == 0:
        print(f"[bench] WORLD_SIZE={world} differs from --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP extension is the product path (no CPU fallback)")
    local = local % torch.cuda.device_count()  # (functional tests run several ranks on one GPU)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    _hip.lib()  # fail loudly now if the extension is missing

    model = build_model(device)
    parallel.broadcast_parameters(model)
    trainer = LidarTrainer(model, lr=1e-2, iters=30000, fp16=True, scale=SCALE, world_size=world,
                           render_kwargs=dict(num_steps=NUM_STEPS, upsample_steps=UPSAMPLE))
    poses = synthetic_frames(60, device)
    n_steps_total = args.warmup + args.steps
    batches = [make_batch(poses, s, args.rays, rank, device) for s in range(min(n_steps_total, 60))]

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for s in range(args.warmup):
        trainer.step(*batches[s % len(batches)])
    sync()
    # HIP events around the encoder entry points only (the roofline candidates): every timed call costs two event
    # records on the stream, and timing all ~25 calls of a step inflates the step by ~4 % (--kernel-timers for all)
    grid_calls = ["lnh_grid_encode_forward", "lnh_grid_encode_forward_mapped", "lnh_grid_encode_backward",
                  "lnh_grid_encode_backward_ws"]
    all_calls = grid_calls + ["lnh_mlp_forward", "lnh_mlp_backward", "lnh_lidar_composite_forward",
                              "lnh_lidar_composite_backward", "lnh_lidar_resample", "lnh_lidar_weights",
                              "lnh_freq_encode_forward", "lnh_density_mlp_forward", "lnh_density_mlp_backward",
                              "lnh_lidar_color_forward", "lnh_lidar_color_backward", "lnh_lidar_merge_weights",
                              "lnh_lidar_sample_points", "lnh_adam_table_step"]
    _hip.enable_timers(all_calls if args.kernel_timers else grid_calls)
    t0 = time.perf_counter()
    for s in range(args.steps):
        loss = trainer.step(*batches[(args.warmup + s) % len(batches)])
    sync()
    elapsed = time.perf_counter() - t0
    timers = _hip.disable_timers()
    elapsed = parallel.max_over_ranks(elapsed, device)
    loss_val = float(loss.detach().float().item())

    if rank != 0:
        return
    rays_total = args.rays * world * args.steps
    # ---- per-kernel HIP-event durations inside the timed region
    kernels = {}
    for name, evs in timers.items():
        ms = [a.elapsed_time(b) for a, b, _ in evs]
        units = [t for _, _, t in evs if t]
        kernels[name] = {"calls": len(ms), "total_ms": round(sum(ms), 3), "avg_us": round(1e3 * sum(ms) / len(ms), 2),
                         "points": int(sum(units)) if units else None}
    dom = max(("lnh_grid_encode_forward", "lnh_grid_encode_forward_mapped", "lnh_grid_encode_backward",
               "lnh_grid_encode_backward_ws"),
              key=lambda k: kernels.get(k, {}).get("total_ms", 0))
    per_pt = GRID_FWD_BYTES if "forward" in dom else GRID_BWD_BYTES
    k = kernels[dom]
    avg_points = k["points"] / k["calls"]
    achieved = per_pt * k["points"] / (k["total_ms"] * 1e-3) / 1e9  # GB/s over all launches of that kernel
    # HBM bytes per launch from the committed PMC pass of this same command (profiles/collect.sh); bench.py itself
    # cannot run rocprofv3, so the number is only reported when that file exists and names the dominant kernels
    traffic, traffic_src = None, None
    pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc.json")
    if os.path.exists(pmc_path) and args.rays == 4096:
        pk = json.load(open(pmc_path)).get("kernels", {})
        names = {"lnh_grid_encode_backward_ws": ("k_grid_bwd_scatter", "k_grid_bwd_reduce"),
                 "lnh_grid_encode_forward_mapped": ("k_grid_forward",), "lnh_grid_encode_forward": ("k_grid_forward",)}
        parts = [pk.get(n, {}).get("hbm_bytes_per_launch") for n in names.get(dom, ())]
        if parts and all(p is not None for p in parts):
            traffic = int(sum(parts))
            traffic_src = "profiles/r01_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, " + " + ".join(names[dom]) + ")"
    result = {
        "metric": "train rays/sec (encode+MLP+composite+bwd), KITTI-360 66x1030",
        "value": round(rays_total / elapsed, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f16 (fp16 hash tables + fp16 MFMA MLP, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": "KITTI-360 seq 1908 shaped: hash-grid L=16 F=2 (2^19 rows, res 16..32768) + 64-wide "
                               "fused MLPs, 66x1030 range image", "rays_per_gpu_per_step": args.rays,
                   "samples_per_ray": NUM_STEPS + UPSAMPLE, "parallelism": f"dp{world}",
                   "optimizer": "Adam + dynamic loss scaling, in the timed region (hash table: fused lnh_adam_table_step; MLPs: torch fused Adam)", "final_loss": round(loss_val, 5)},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": traffic_src, "algorithmic_bytes_per_launch": int(per_pt * avg_points),
                     "bytes_per_point": per_pt, "points_per_launch": int(avg_points),
                     "avg_launch_us": k["avg_us"]},
        "kernels": kernels,
    }
    # secondary number of SURVEY §8(d): full-frame evaluation (67 980 rays of a 66 x 1030 range image, staged in
    # chunks of 4096, no perturbation, no gradient) — reported beside the headline metric, never instead of it
    if world == 1 and not args.no_eval:
        model.eval()
        frame = make_batch(poses, 0, 66 * 1030, rank, device)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            for _ in range(2):
                model.render(frame[0], frame[1], cal_lidar_color=True, staged=True, max_ray_batch=4096, perturb=False,
                             num_steps=NUM_STEPS, upsample_steps=UPSAMPLE)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                model.render(frame[0], frame[1], cal_lidar_color=True, staged=True, max_ray_batch=4096, perturb=False,
                             num_steps=NUM_STEPS, upsample_steps=UPSAMPLE)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
        result["eval"] = {"metric": "full-frame eval rays/sec (66x1030, staged 4096)", "value": round(66 * 1030 / dt, 1),
                          "unit": "rays/s", "ms_per_frame": round(1e3 * dt, 3)}
        model.train()
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
    print(json.dumps(result))


if __name__ == "__main__":
    main()
