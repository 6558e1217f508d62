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
dominant kernel (algorithmic bytes / HIP-event duration measured inside the timed region) with the forward encode and
the MFMA utilisation of the MLP kernels beside it, and a CPU baseline (BASELINE config 1 — the reference's CPU-runnable
path: pure-torch frequency encoder + nn.Linear stacks — restated in oracle/render_ref.py, timed on the host cores, N=1
only).  `--gpus N` without torchrun's environment re-launches itself under torch.distributed.run with N ranks.
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
MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak
# SURVEY.md §8(d) / BASELINE.md §4: MLP flops per sample point, forward; backward (dgrad + wgrad) counts twice that
SIGMA_FLOPS = 2 * (32 * 64 + 64 * 16)                 # 6 144
COLOR_FLOPS = 2 * (96 * 64 + 64 * 64 + 64 * 16)       # 22 528 (90 -> 64 -> 64 -> 2 padded to MFMA tiles), masked points
# what the colour kernels execute per masked point: the 75 direction columns of layer 0 are folded into a per-ray
# term, leaving K = 16 per sample
COLOR_FLOPS_EXECUTED = 2 * (16 * 64 + 64 * 64 + 64 * 16)

# algorithmic bytes per sample point, fp16 tables, L=16, F=2, D=3
GRID_FWD_BYTES = 12 + 16 * 8 * 2 * 2 + 16 * 2 * 2  # 588
GRID_BWD_BYTES = 12 + 16 * 2 * 2 + 2 * (16 * 8 * 2 * 2)  # 1100
COMPOSITE_BYTES = 16 + 28  # compositing forward 16 B + backward 28 B per sample (K = 2 channels)


def synthetic_frames(n_frames, device):
    """60 poses on a straight ~1 m/frame trajectory with a small yaw."""
    poses = torch.eye(4).repeat(n_frames, 1, 1)
    for k in range(n_frames):
        th = np.deg2rad(0.5 * k)
        poses[k, :3, :3] = torch.tensor([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
        poses[k, :3, 3] = torch.tensor([(k - n_frames / 2) * SCALE, 0.0, 0.0])
    return poses.to(device)


def analytic_scene(o, d):
    """A learnable synthetic scene (SURVEY.md §8(d) "trained-like" table): the sensor drives inside a sphere of 32 m radius
    over a ground plane 1.9 m below it — depth = first hit along the ray (scene units), intensity = a smooth pattern on the
    hit point, ray-drop = 1 where the hit lies within the 80 m range.  o, d [..., 3] -> gt [..., 3] = (raydrop, intensity,
    depth), consistent between frames, so a field can actually learn it (the default ground truth is random per ray)."""
    R, ground = 32.0 * SCALE, -1.9 * SCALE
    b = (o * d).sum(-1)
    t_sphere = -b + torch.sqrt((b * b - ((o * o).sum(-1) - R * R)).clamp(min=0))
    dz = d[..., 2]
    t_ground = torch.where(dz < -1e-6, (ground - o[..., 2]) / dz.clamp(max=-1e-6), torch.full_like(dz, float("inf")))
    t = torch.minimum(t_sphere, t_ground)
    hit = o + d * t.unsqueeze(-1)
    raydrop = (t < 80.0 * SCALE).float()
    intensity = 0.5 + 0.25 * torch.sin(hit[..., 0] / SCALE * 0.7) + 0.25 * torch.cos(hit[..., 1] / SCALE * 0.45)
    return torch.stack([raydrop, intensity * raydrop, t * raydrop], -1)


def make_batch(poses, step, n_rays, rank, device, patch=(1, 1), scene="random"):
    """One frame, n_rays pixels drawn as patches of `patch` pixels (base_dataset.py:50-70: 1x1, or 2x8 on the reference's
    patch epochs), synthetic ground truth (raydrop, intensity, depth*scale): random per ray or the analytic
    scene above."""
    from lidarnerf.dataset.rays import get_lidar_rays
    g = torch.Generator(device="cpu").manual_seed(1234 + step * 131 + rank * 7919)
    torch.manual_seed(1234 + step * 131 + rank * 7919)
    pose = poses[step % poses.shape[0]][None]
    r = get_lidar_rays(pose, INTRINSICS, H_IMG, W_IMG, n_rays, patch_size=1 if tuple(patch) == (1, 1) else list(patch))
    n = r["rays_o"].shape[1]
    if scene == "analytic":
        gt = analytic_scene(r["rays_o"][0].float(), r["rays_d"][0].float())[None].to(device)
    else:
        raydrop = (torch.rand(n, generator=g) < 0.85).float()
        intensity = torch.rand(n, generator=g)
        depth = SCALE * (2 + 78 * torch.rand(n, generator=g)) * raydrop
        gt = torch.stack([raydrop, intensity, depth], -1)[None].to(device)
    return r["rays_o"].contiguous(), r["rays_d"].contiguous(), gt


def build_model(device):
    from lidarnerf.nerf.network import NeRFNetwork
    torch.manual_seed(0)
    model = NeRFNetwork(encoding="hashgrid", desired_resolution=32768, log2_hashmap_size=19, num_layers=2,
                        hidden_dim=64, geo_feat_dim=15, bound=1, density_scale=1, min_near=SCALE,
                        min_near_lidar=SCALE, density_thresh=10, bg_radius=-1)
    return model.to(device).train()


def usable_cores():
    """Host cores this process may actually use: os.cpu_count() capped by the scheduler affinity and the cgroup CPU quota
    (the GPU boxes show 256 logical CPUs behind a 16-CPU quota: 256 torch threads then run 45 s per step, 16 threads 1 s)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def _cpu_config1(n_rays, threads, warmup, steps, budget_s):
    """Step times of BASELINE config 1 on `threads` host threads: RefFreqField (pure-torch positional encoder + bias-free
    nn.Linear stacks, fp32) through run_lidar (NeRFRenderer.run restated), forward + LiDAR loss + backward, n_rays x 832
    samples of KITTI-360-shaped synthetic rays.  `warmup` untimed steps, then up to `steps` timed ones, stopping early once
    `budget_s` of wall time is spent (at least 3 timed steps).  Returns the list of timed step durations."""
    from oracle import render_ref
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    ref = render_ref.RefFreqField().train()
    poses = synthetic_frames(60, "cpu")
    o, d, gt = make_batch(poses, 0, n_rays, 0, "cpu")
    o, d, gt = o[0], d[0], gt[0]
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    times, t_begin = [], time.perf_counter()
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        ref.zero_grad(set_to_none=True)
        res = render_ref.run_lidar(o, d, ref.density, ref.color, aabb, SCALE, NUM_STEPS, UPSAMPLE, perturb=True,
                                   training=True)
        render_ref.lidar_loss(res["depth_lidar"], res["image_lidar"], gt).backward()
        if it >= warmup:
            times.append(time.perf_counter() - t0)
        if len(times) >= 3 and time.perf_counter() - t_begin > budget_s:
            break
    return times


def cpu_baseline(full=False):
    """config 1, N = 1024 rays x 832 samples, forward + loss + backward, fp32, median step time on k = 1 and
    k = ALL usable host cores (see usable_cores(): os.cpu_count() capped by affinity and cgroup quota — both stated).
    `full` (the default of bench.py) runs the protocol as written there (10 warm-up + 30 timed steps per leg, ~2 min of CPU
    time); --cpu-baseline-quick takes a bounded sample of it (1 + up to 10 steps per leg inside ~12 s each, at least 3
    timed).  `value` / `cores` = the FASTER leg; both legs are listed.  Validated against the imported reference in the build
    container (tests/test_config1_cpu.py, golden vector G7) and pinned on the GPU box by golden vectors G2 / G4 / G7."""
    import platform
    k_all, quota = usable_cores()
    saved = torch.get_num_threads()
    legs = []
    try:
        for k in sorted({1, k_all}):
            ts = _cpu_config1(1024, k, 10 if full else 1, 30 if full else 10, 1e9 if full else 12.0)
            med = float(np.median(ts))
            legs.append({"cores": k, "value": round(1024 / med, 2), "unit": "rays/s", "steps": len(ts),
                         "warmup": 10 if full else 1, "s_per_step_median": round(med, 3),
                         "s_per_step_min_max": [round(min(ts), 3), round(max(ts), 3)]})
    finally:
        torch.set_num_threads(saved)
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = platform.processor()
    top = max(legs, key=lambda l: l["value"])  # the FASTEST leg is the baseline (on some hosts k = 1 beats k = all)
    return {"value": top["value"], "unit": "rays/s", "cores": top["cores"], "kind": "port",
            "sample": f"BASELINE config 1 (pure-torch freq encoder + nn.Linear 39->64->16 / 90->64->64->2, fp32), 1024 rays x "
                      f"{NUM_STEPS + UPSAMPLE} samples, fwd+loss+bwd; median of {top['steps']} timed steps after "
                      f"{top['warmup']} warm-up, torch.set_num_threads({top['cores']}) = the faster of k = 1 and k = all usable cores "
                      f"({k_all}); " + ("BASELINE.md §3's protocol (10 warm-up + 30 timed steps per leg)" if full else
                                         "bounded sample of BASELINE.md §3's 10 + 30 protocol (--cpu-baseline-quick)")
                      + "; oracle/render_ref.py RefFreqField",
            "legs": legs,
            "host": {"cpu": model, "os_cpu_count": os.cpu_count(), "cgroup_cpu_quota": quota, "usable_cores": k_all,
                     "torch": torch.__version__}}


def _relaunch_distributed(n):
    """`python bench.py --gpus N` outside torchrun: start N ranks (one per GPU) and hand them the same command line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def _flush_c_stdout():
    """Flush the C runtime's stdout buffer (libraries of this process that printf — RCCL's banner — would otherwise empty
    it at exit, after the JSON line)."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass


def lib_sha16():
    """First 16 hex digits of the SHA-256 of the loaded liblidarnerf_hip.so: ties a PMC file to the library it profiled."""
    import hashlib
    from lidarnerf import _hip
    try:
        return hashlib.sha256(open(_hip.lib_path(), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def _latest_pmc():
    """HBM bytes per launch of the grid kernels from the newest committed PMC pass (profiles/r*_pmc.json, written by
    profiles/collect.sh from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command), and the hash of
    the library that pass profiled (None for files written before round 4)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            d = json.load(open(path))
            return os.path.relpath(path, ROOT), d.get("kernels", {}), d.get("lib_sha16")
        except Exception:
            continue
    return None, {}, None


def run_nerfmvl(args):
    """BASELINE config 4 (secondary workload, `--workload nerfmvl`): NeRF-MVL-shaped object scene — 256 x 1800 range image,
    intrinsics (fov_up, fov) = (15, 40), scale 0.005 (configs/nerf_mvl.txt; preprocess/generate_train_rangeview.py:166-168),
    rays of a frame restricted to the object's bounding box and thinned to <= 4096 per step as the reference's collate
    does (nerfmvl_dataset.py:116-168) — trained with OCCUPANCY-GRID ray sampling: density-grid update every 16 steps,
    lnh_march_rays_train, ragged LiDAR compositing.  Synthetic object: a sphere of 2 m radius seen from a 6 m ring."""
    from lidarnerf import _hip
    from lidarnerf.dataset.rays import get_lidar_rays
    from lidarnerf.nerf.network import NeRFNetwork
    from lidarnerf.nerf.train_step import LidarTrainer
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP extension is the product path (no CPU fallback)")
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    _hip.lib()
    scale, Himg, Wimg, intr = 0.005, 256, 1800, (15.0, 40.0)
    R, ring = 2.0 * scale, 6.0 * scale
    torch.manual_seed(0)
    model = NeRFNetwork(encoding="hashgrid", desired_resolution=32768, log2_hashmap_size=19, num_layers=2, hidden_dim=64,
                        geo_feat_dim=15, bound=1, density_scale=1, min_near=scale, min_near_lidar=scale,
                        density_thresh=10, bg_radius=-1, cuda_ray=True).to(device).train()
    # the whole step captured in a hipGraph and replayed (LidarTrainer graph mode); --no-graph = one launch at a time
    use_graph = not args.no_graph
    trainer = LidarTrainer(model, lr=1e-2, iters=30000, fp16=True, scale=scale, graph=use_graph)

    def frame(k):
        th = 2 * np.pi * k / 60
        pose = torch.eye(4)
        # sensor x axis points at the object (beta = 0 is +x in the sensor frame: base_dataset.py:72-87)
        pose[:3, :3] = torch.tensor([[-np.cos(th), np.sin(th), 0], [-np.sin(th), -np.cos(th), 0], [0, 0, 1.0]])
        pose[:3, 3] = torch.tensor([ring * np.cos(th), ring * np.sin(th), 0.0])
        r = get_lidar_rays(pose[None].to(device), intr, Himg, Wimg, -1)
        o, d = r["rays_o"][0], r["rays_d"][0]
        b = (o * d).sum(-1)
        disc = b * b - ((o * o).sum(-1) - (1.2 * R) ** 2)          # inside the object's bounding sphere (bbox mask)
        keep = (disc > 0) & (b < 0)
        o, d, b = o[keep], d[keep], b[keep]
        disc_s = b * b - ((o * o).sum(-1) - R * R)
        hit = disc_s > 0
        depth = torch.where(hit, -b - torch.sqrt(disc_s.clamp(min=0)), torch.zeros_like(b))
        gt = torch.stack([hit.float(), torch.full_like(b, 0.5), depth], -1)
        return o, d, gt

    frames = [frame(k) for k in range(60)]

    def batch(step):
        o, d, gt = frames[step % 60]
        g = torch.Generator(device="cpu").manual_seed(99 + step)
        sel = torch.randperm(o.shape[0], generator=g)[:args.rays].to(device)
        return o[sel][None].contiguous(), d[sel][None].contiguous(), gt[sel][None].contiguous()

    n_pre = 320  # let the occupancy grid settle before anything is timed: 20 grid updates — the first 16 are full 128^3 sweeps
                 # (renderer.py update_extra_state: ~100 ms each), the steady state updates a quarter of the cells
    n_prof = min(args.steps, 5)
    n_regions = 3  # three timed regions of K steps each, `value` = the median one (see the note in the line printed below)
    batches = [batch(s) for s in range(n_pre + args.warmup + n_regions * args.steps + n_prof)]
    for s in range(n_pre + args.warmup):
        trainer.step(*batches[s])
    torch.cuda.synchronize()
    use_graph = use_graph and trainer.graph  # (False if a capture did not go through: launch by launch from there on)
    grid_calls = ["lnh_grid_encode_forward", "lnh_grid_encode_backward_ws", "lnh_grid_encode_backward"]

    def region(r):
        """K steps bracketed by synchronisation; host time per step split into plain steps / steps that begin with a grid
        update (which reads the sample counts back, i.e. WAITS for the device to drain its queue)."""
        first = n_pre + args.warmup + r * args.steps
        if not use_graph:  # (a replayed graph makes no library calls to put events around: the eager pass below times them)
            _hip.enable_timers(grid_calls)
        graphs0 = len(trainer._graphs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        counts, covered, host_step, host_update = [], 0, [], []
        # (covered: the marches the ring reads below cover — the first block may reach a few steps back before the region)
        for s in range(args.steps):
            upd = trainer.global_step % trainer.update_extra_interval == 0
            if upd and model.local_step:
                # the ring of the last 16 marches, before the grid update resets it: ONE tiny kernel per 16 steps
                counts.append(model.step_counter[:model.local_step, 0].sum())
                covered += model.local_step
            t1 = time.perf_counter()
            loss = trainer.step(*batches[first + s])
            (host_update if upd else host_step).append((time.perf_counter() - t1) * 1e3)
        counts.append(model.step_counter[:model.local_step, 0].sum())
        covered += model.local_step
        host_ms = (time.perf_counter() - t0) * 1e3 / args.steps
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        n_rays = sum(b[0].shape[1] for b in batches[first:first + args.steps])
        samples = float(torch.stack(counts).float().sum()) / (covered * (n_rays / args.steps))  # marched samples per ray
        return {"elapsed": elapsed, "host_ms": host_ms, "host_step": host_step, "host_update": host_update, "loss": loss,
                "n_rays": n_rays, "samples": samples, "captures": len(trainer._graphs) - graphs0,
                "timers": _hip.disable_timers() if not use_graph else {}}

    regions = [region(r) for r in range(n_regions)]
    mid = sorted(regions, key=lambda g: g["elapsed"])[n_regions // 2]
    elapsed, host_ms, host_step, host_update, loss = (mid[k] for k in ("elapsed", "host_ms", "host_step", "host_update", "loss"))
    n_rays, samples, timers = mid["n_rays"], mid["samples"], mid["timers"]
    samples_total = samples * n_rays
    occ = float((model.density_grid > min(model.mean_density, model.density_thresh)).float().mean())
    graphs_captured = len(trainer._graphs)

    def event_table(tm):
        out = {}
        for name, evs in tm.items():
            ms = [a.elapsed_time(b) for a, b, _ in evs]
            out[name] = {"calls": len(ms), "total_ms": round(sum(ms), 3), "avg_us": round(1e3 * sum(ms) / len(ms), 2),
                         "points": int(sum(t for _, _, t in evs if t))}
        return out
    kernels = event_table(timers)
    # every entry point of a few more steps (outside the timed region: the events cost ~4 %)
    _hip.enable_timers(None)
    trainer.graph = False  # launch by launch (same kernels, same device-side learning rate)
    for s in range(n_prof):
        trainer.step(*batches[n_pre + args.warmup + n_regions * args.steps + s])
    torch.cuda.synchronize()
    trainer.graph = use_graph
    eager_tab = event_table(_hip.disable_timers())
    per_call = {k: v["avg_us"] for k, v in eager_tab.items()}
    if use_graph:
        kernels = {k: v for k, v in eager_tab.items() if k in grid_calls}

    def roof(name, per_pt):
        k = kernels.get(name)
        if not k or not k["points"]:
            return None
        gbs = per_pt * k["points"] / (k["total_ms"] * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": name, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "bytes_per_point": per_pt,
                "points_per_launch": int(k["points"] / k["calls"]), "avg_launch_us": k["avg_us"],
                "algorithmic_bytes_per_launch": int(per_pt * k["points"] / k["calls"]),
                "note": "marched samples per launch are two orders of magnitude below the dense workload's 3.4 M: the launch "
                        "is dominated by per-launch costs (bucket images, cursors), not by bytes per point"}
    bwd_name = "lnh_grid_encode_backward_ws" if "lnh_grid_encode_backward_ws" in kernels else "lnh_grid_encode_backward"
    print(json.dumps({
        "metric": "train rays/sec (occupancy-grid sampling + encode + MLP + ragged composite + bwd), NeRF-MVL-shaped 256x1800",
        "value": round(n_rays / elapsed, 1), "unit": "rays/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16 (fp16 hash tables + fp16 MFMA MLP, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": "NeRF-MVL shaped (BASELINE configs[3]): 256x1800 range image, intrinsics (15, 40), scale "
                               "0.005, hash-grid L=16 F=2 + 64-wide MLPs, occupancy-grid ray sampling (128^3, update every "
                               "16 steps), synthetic 2 m sphere seen from a 6 m ring",
                   "rays_per_gpu_per_step": int(n_rays / args.steps), "mean_samples_per_ray": round(samples, 2),
                   "dense_samples_per_ray": NUM_STEPS + UPSAMPLE, "occupied_cell_fraction": round(occ, 5),
                   "pretrain_steps": n_pre, "final_loss": round(float(loss.detach()), 5),
                   "render_path": "fused ragged chain (nerf/fused.py FusedLidarRagged) + fused table optimizer"
                   if trainer.table is not None else "modular density()/color() path",
                   "launch": (f"hipGraph replay of the whole step (march .. optimizers; {graphs_captured} graph(s) captured, "
                              "one per sample capacity of the ladder; a step at a capacity not seen before is captured then and there)")
                   if use_graph else "launch by launch from Python"},
        "ms_per_step_repeats": [round(1e3 * g["elapsed"] / args.steps, 3) for g in regions],
        "ms_per_step_first_region": round(1e3 * regions[0]["elapsed"] / args.steps, 3),
        "captures_in_region": [g["captures"] for g in regions],
        "capture_host_ms": list(getattr(trainer, "capture_ms", [])),
        "repeats_note": "three timed regions of K steps each, one after the other; `value`, `ms_per_step`, the sample and host "
                        "figures are those of the MEDIAN region.  The occupancy grid keeps changing while it trains, and a "
                        "step whose sample capacity reaches a rung of the ladder it has not seen is captured anew (a few "
                        "milliseconds, once per rung for the whole run — captures_in_region says which regions held one): a "
                        "region of 128 steps that holds a capture is not the steady state the metric names.",
        "samples_per_s": round(samples_total / elapsed, 1), "host_enqueue_ms_per_step": round(host_ms, 3),
        "host_ms": {"per_plain_step_median": round(float(np.median(host_step)), 4) if host_step else None,
                    "per_update_step_mean": round(float(np.mean(host_update)), 3) if host_update else None,
                    "update_steps": len(host_update),
                    "update_steps_ms": [round(v, 2) for v in host_update],
                    "plain_step_max": round(float(np.max(host_step)), 3) if host_step else None,
                    "note": "host_enqueue_ms_per_step is the wall time of the loop / steps: it CONTAINS the waits of the grid "
                            "updates (every 16th step reads the marched-sample counts back and so waits for the queued steps to "
                            "finish) — with a replayed graph the host runs ~15 steps ahead and spends that wait there.  "
                            "per_plain_step_median is what the host needs to issue one step."},
        "roofline": roof(bwd_name, GRID_BWD_BYTES), "roofline_fwd": roof("lnh_grid_encode_forward", GRID_FWD_BYTES),
        "kernels": kernels, "entry_points_avg_us": per_call}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=4096, help="rays per step per GPU")
    ap.add_argument("--mlp-dtype", choices=("fp16", "bf16"), default="fp16",
                    help="MFMA operand type of the MLP kernels (bf16 = BASELINE config 5; hash features stay fp16)")
    ap.add_argument("--no-graph", action="store_true",
                    help="issue the training step launch by launch instead of replaying the captured hipGraph (one GPU)")
    ap.add_argument("--workload", choices=("kitti360", "nerfmvl"), default="kitti360",
                    help="kitti360 = the headline benchmark (BASELINE configs[1]); nerfmvl = configs[3], occupancy-grid path")
    ap.add_argument("--table", choices=("init", "trained"), default="init",
                    help="init = fresh-init field, random ground truth (SURVEY 8(d) default: flat weights, the colour head sees "
                         "most samples); trained = the 'trained-like' variant: the field is first trained --pretrain-steps steps on "
                         "the analytic scene (sphere + ground plane), so the weights concentrate on surfaces, then timed on it")
    ap.add_argument("--pretrain-steps", type=int, default=600)
    ap.add_argument("--patch", default="1x1", help="rays drawn as PXxPY pixel patches; 2x8 = the reference's patch epochs "
                                                   "(structural-gradient loss term, utils.py:760-876)")
    ap.add_argument("--dp-windows", action="store_true",
                    help="1 GPU: run the table-gradient backward the way data parallel does (one scatter pass + the reduce pass per level window, the exchange a "
                         "no-op) to price the compute side of the DP pipeline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="(default since round 4) BASELINE.md 3 protocol in full: 10 warm-up + 30 timed steps per leg")
    ap.add_argument("--cpu-baseline-quick", action="store_true", help="bounded sample of that protocol (1 + up to 10 steps per leg, ~12 s each)")
    ap.add_argument("--no-mfma-states", action="store_true", help="skip the two fixed-state MFMA measurements (fresh table with a frozen optimizer; 600-step trained table)")
    ap.add_argument("--no-eval", action="store_true", help="skip the secondary full-frame evaluation measurement")
    ap.add_argument("--kernel-timers", action="store_true", help="HIP-event timing of every C-ABI call (adds ~4 %%)")
    args = ap.parse_args()

    from lidarnerf import _hip, parallel
    from lidarnerf.nerf.train_step import LidarTrainer

    if args.workload == "nerfmvl":
        return run_nerfmvl(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _relaunch_distributed(args.gpus)
    # (checked BEFORE the rendezvous: a rank that is going to refuse must not leave the others waiting for it)
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world} ranks were launched")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP extension is the product path (no CPU fallback)")
    if env_world > torch.cuda.device_count() and os.environ.get("LNH_DIST_BACKEND") != "gloo":
        raise SystemExit(f"bench.py: {env_world} ranks but only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU; "
                         "LNH_DIST_BACKEND=gloo allows functional runs with several ranks on one GPU)")
    rank, local, world = parallel.init_from_env()
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    _hip.lib()  # fail loudly now if the extension is missing

    model = build_model(device)
    parallel.broadcast_parameters(model)
    # One GPU (default; --no-graph turns it off): the step replayed as a captured hipGraph (LidarTrainer graph mode) in the timed
    # region — the same kernels in the same order, 0.05 ms of host time per step instead of 0.8 .. 0.9: on the hosts of this
    # build the step is GPU-bound either way (2.124 against 2.136 ms), on a host 2.4 x slower (such boxes exist in the pool:
    # the config-4 step took 2.0 instead of 0.85 ms there) it would not be.  The per-entry-point timings (rooflines, MFMA)
    # come from a launch-by-launch region of the same length right after it (HIP events need calls to bracket).
    # Data parallel (round 5): under RCCL the step is captured WITH its collectives (LidarTrainer graph mode), so that N ranks
    # replay N identical graphs instead of issuing ~60 launches per 2 ms step from N Python processes sharing the box's CPU
    # quota; gloo (functional runs with several ranks on one GPU) cannot be captured and runs launch by launch.
    use_graph = bool(not args.no_graph and not args.dp_windows and (world == 1 or parallel.backend() == "nccl"))
    trainer = LidarTrainer(model, lr=1e-2, iters=30000, fp16=True, scale=SCALE, world_size=world,
                           render_kwargs=dict(num_steps=NUM_STEPS, upsample_steps=UPSAMPLE),
                           mlp_dtype=torch.bfloat16 if args.mlp_dtype == "bf16" else torch.float16, graph=use_graph)
    poses = synthetic_frames(60, device)
    n_steps_total = args.warmup + args.steps
    patch = tuple(int(v) for v in args.patch.lower().split("x"))
    scene = "analytic" if args.table == "trained" else "random"
    batches = [make_batch(poses, s, args.rays, rank, device, patch, scene) for s in range(60 if args.table == "trained"
                                                                                          else min(n_steps_total, 60))]
    step_kw = {} if patch == (1, 1) else {"patch": patch}
    if args.dp_windows:
        if world != 1:
            raise SystemExit("bench.py: --dp-windows prices the data-parallel backward on ONE GPU")
        from lidarnerf.nerf import fused as _fused
        _fused.FORCE_DP_WINDOWS = True
    pretrain = None
    if args.table == "trained":  # let the field learn the scene (untimed): weights concentrate, the colour mask thins out
        first = last = None
        for s in range(args.pretrain_steps):
            l = trainer.step(*batches[s % len(batches)], **step_kw)
            if s == 0:
                first = l
            last = l
        pretrain = {"steps": args.pretrain_steps, "first_loss": round(float(first.detach()), 4),
                    "last_loss": round(float(last.detach()), 4)}

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for s in range(args.warmup):
        trainer.step(*batches[s % len(batches)], **step_kw)
    if world > 1:
        # First contact of this step with N ranks (the collectives, and in graph mode their replay from a hipGraph, have met
        # one-rank groups only on the boxes this was developed on): wait for the warm-up with a deadline and leave with a
        # message instead of sitting in a collective that never completes.
        ev = torch.cuda.Event()
        ev.record()
        t_dead = time.perf_counter() + 180.0
        while not ev.query():
            if time.perf_counter() > t_dead:
                sys.stderr.write(json.dumps({"error": "the data-parallel warm-up steps did not complete within 180 s",
                                             "rank": rank, "world": world, "graph": bool(trainer.graph),
                                             "graph_error": trainer.graph_error,
                                             "hint": "python bench.py --gpus N --no-graph runs the step launch by launch"}) + "\n")
                sys.stderr.flush()
                os._exit(19)
            time.sleep(0.005)
    sync()
    # HIP events around the encoder entry points only (the roofline candidates): every timed call costs two event
    # records on the stream, and timing all ~25 calls of a step inflates the step by ~4 % (--kernel-timers for all)
    grid_calls = ["lnh_grid_encode_forward", "lnh_grid_encode_forward_mapped", "lnh_grid_encode_backward",
                  "lnh_grid_encode_backward_ws", "lnh_grid_encode_backward_ws_levels", "lnh_grid_encode_backward_ws_begin",
                  "lnh_grid_encode_backward_ws_finish"]
    sfx = "_bf16" if args.mlp_dtype == "bf16" else ""
    # roofline_path (encode forward + backward + compositing): the compositing backward and the fused forward tail (merged
    # weights + colour head + compositing sums of a ray in ONE kernel) are timed beside the encoder entry points
    path_calls = grid_calls + ["lnh_lidar_composite_backward", "lnh_lidar_color_composite_forward" + sfx]
    # (the colour head's forward runs inside the fused forward tail — merged weights + colour head + compositing sums of a
    #  ray in one kernel — whose whole time is charged to the MLPs here)
    mlp_calls = [n + sfx for n in ("lnh_density_mlp_forward", "lnh_density_mlp_backward", "lnh_lidar_color_forward",
                                   "lnh_lidar_color_composite_forward", "lnh_lidar_color_backward",
                                   "lnh_lidar_color_backward_image")]
    all_calls = grid_calls + mlp_calls + ["lnh_mlp_forward", "lnh_mlp_backward", "lnh_lidar_composite_forward",
                                          "lnh_lidar_composite_backward", "lnh_lidar_resample", "lnh_lidar_weights",
                                          "lnh_freq_encode_forward", "lnh_lidar_merge_weights",
                                          "lnh_lidar_sample_points", "lnh_adam_table_step"]
    if use_graph:  # (a graph exists after two steps at a batch shape: never capture inside the timed region)
        while trainer.graph and not trainer._graphs:
            trainer.step(*batches[0], **step_kw)
        sync()
        use_graph = trainer.graph  # (False if the capture did not go through: launch by launch, reason in the line)
    _hip.enable_timers(all_calls if args.kernel_timers else path_calls)
    t0 = time.perf_counter()
    for s in range(args.steps):
        loss = trainer.step(*batches[(args.warmup + s) % len(batches)], **step_kw)
    host_ms = (time.perf_counter() - t0) * 1e3 / args.steps  # time the host needs to ENQUEUE a step (no device wait)
    sync()
    elapsed = time.perf_counter() - t0
    timers = _hip.disable_timers()
    elapsed = parallel.max_over_ranks(elapsed, device)
    loss_val = float(loss.detach().float().item())
    graph_info = None
    if use_graph:
        # the same number of steps launch by launch: entry-point timings for the rooflines, and the eager step time beside it
        graph_info = {"graphs_captured": len(trainer._graphs), "host_enqueue_ms_per_step_graph": round(host_ms, 4)}
        trainer.graph = False
        _hip.enable_timers(all_calls if args.kernel_timers else path_calls)
        t1 = time.perf_counter()
        for s in range(args.steps):
            trainer.step(*batches[(args.warmup + args.steps + s) % len(batches)], **step_kw)
        graph_info["host_enqueue_ms_per_step_eager"] = round((time.perf_counter() - t1) * 1e3 / args.steps, 4)
        sync()
        graph_info["ms_per_step_eager"] = round((time.perf_counter() - t1) * 1e3 / args.steps, 3)
        timers = _hip.disable_timers()

    def event_table(tm):
        out = {}
        for name, evs in tm.items():
            ms = [a.elapsed_time(b) for a, b, _ in evs]
            units = [t for _, _, t in evs if t]
            out[name] = {"calls": len(ms), "total_ms": round(sum(ms), 3), "avg_us": round(1e3 * sum(ms) / len(ms), 2),
                         "points": int(sum(units)) if units else None}
        return out

    # ---- MFMA pass (outside the timed region, every rank): a few more steps with HIP events on the four MLP entry
    #      points and the fraction of samples the colour head really evaluates (weights > 1e-4) recorded per step
    from lidarnerf.nerf import fused
    n_prof = min(args.steps, 5)
    fused.MASK_STATS = []
    _hip.enable_timers(mlp_calls)
    for s in range(n_prof):
        trainer.step(*batches[(args.warmup + args.steps + s) % len(batches)], **step_kw)
    sync()
    mlp_timers = event_table(_hip.disable_timers())
    # (the mask fraction belongs to the steps the MLP kernels were timed in: close the collection before anything else runs)
    mask_frac = float(torch.stack(fused.MASK_STATS).mean().item()) if fused.MASK_STATS else 1.0
    fused.MASK_STATS = None
    # ---- the same MFMA measurement at two FIXED states (the fraction of samples above the colour mask moves with the
    #      training trajectory, so the as-run figure above depends on --steps / --warmup): (a) a fresh table with the
    #      optimizer frozen (lr = 0: every step sees the initial field, mask ~ 1.0), (b) the 'trained-like' table (the field
    #      first learns the analytic scene for --pretrain-steps untimed steps, mask ~ 0.15).  One GPU only.
    def mfma_pass(tr, bats, n):
        fused.MASK_STATS = []
        _hip.enable_timers(mlp_calls + path_calls)
        for s in range(n):
            tr.step(*bats[s % len(bats)], **step_kw)
        sync()
        tm = event_table(_hip.disable_timers())
        mf = float(torch.stack(fused.MASK_STATS).mean().item()) if fused.MASK_STATS else 1.0
        fused.MASK_STATS = None
        return tm, mf, n

    mfma_states = {}
    if world == 1 and not args.no_mfma_states:
        mk = dict(iters=30000, fp16=True, scale=SCALE, world_size=1,
                  render_kwargs=dict(num_steps=NUM_STEPS, upsample_steps=UPSAMPLE),
                  mlp_dtype=torch.bfloat16 if args.mlp_dtype == "bf16" else torch.float16)
        torch.manual_seed(0)
        m2 = build_model(device)
        t2 = LidarTrainer(m2, lr=0.0, **mk)
        bats_r = [make_batch(poses, s, args.rays, rank, device, patch, "random") for s in range(8)]
        for s in range(2):
            t2.step(*bats_r[s], **step_kw)
        mfma_states["fresh_frozen"] = mfma_pass(t2, bats_r, 5)
        del m2, t2
        if args.table == "trained":
            mfma_states["trained"] = (mlp_timers, mask_frac, n_prof)  # the run itself IS that state
        else:
            torch.manual_seed(0)
            m3 = build_model(device)
            t3 = LidarTrainer(m3, lr=1e-2, **mk)
            bats_a = [make_batch(poses, s, args.rays, rank, device, patch, "analytic") for s in range(60)]
            for s in range(args.pretrain_steps):
                t3.step(*bats_a[s % 60], **step_kw)
            mfma_states["trained"] = mfma_pass(t3, bats_a, 5)
            del m3, t3
        torch.cuda.empty_cache()
    # ---- spread: the timed region is short (K x ~2.3 ms); repeat it twice more (outside the reported number) and list all
    spread = [round(1e3 * elapsed / args.steps, 3)]
    for rep in range(2):
        sync()
        t_r = time.perf_counter()
        for s in range(args.steps):
            trainer.step(*batches[(args.warmup + s + 7 * (rep + 1)) % len(batches)], **step_kw)
        sync()
        spread.append(round(1e3 * parallel.max_over_ranks(time.perf_counter() - t_r, device) / args.steps, 3))
    # ---- SURVEY 8(d)'s protocol asks for >= 100 timed steps: when the driver asks for fewer (--steps 20), one more region of
    #      100 steps is timed the same way (barrier + synchronize on both sides, MAX over ranks) and reported next to the K-step
    #      figure as `value_100` / `ms_per_step_100` — `value` stays the K-step number the driver's own clock brackets
    ms_100 = None
    if args.steps < 100:
        sync()
        t_r = time.perf_counter()
        for s in range(100):
            trainer.step(*batches[(args.warmup + s + 21) % len(batches)], **step_kw)
        sync()
        ms_100 = round(1e3 * parallel.max_over_ranks(time.perf_counter() - t_r, device) / 100, 4)
    # the roofline below divides by the time of ALL MLP kernels of a step: a renamed entry point must not drop out silently
    for role in ("density_mlp_forward", "density_mlp_backward", "color_backward", "color_"):
        fw = [k for k in mlp_timers if role in k and (role != "color_" or "forward" in k)]
        if not fw:
            raise RuntimeError(f"bench: no '{role}' entry point was timed in the MFMA pass (got {sorted(mlp_timers)})")

    # ---- data parallel: the same step with the gradient exchange switched off (every rank keeps its own gradient;
    #      timing only, the replicas are not used afterwards) -> all-reduce cost = inclusive - exclusive
    comm = None
    if world > 1:
        saved_ws, saved_dp, trainer.world, trainer.dp = parallel.world_size, parallel.dp_active, 1, False
        parallel.world_size = lambda: 1
        parallel.dp_active = lambda: False
        n_nc = min(args.steps, 10)
        for s in range(2):
            trainer.step(*batches[s % len(batches)], **step_kw)
        sync()
        t0 = time.perf_counter()
        for s in range(n_nc):
            trainer.step(*batches[(2 + s) % len(batches)], **step_kw)
        sync()
        t_nc = parallel_max = time.perf_counter() - t0
        parallel.world_size, parallel.dp_active, trainer.world, trainer.dp = saved_ws, saved_dp, world, True
        t_nc = parallel.max_over_ranks(parallel_max, device) / n_nc
        # the exchange in numbers: bytes per rank and step, and the level windows they travel in (fused._DP_LEVEL_WINDOWS)
        from lidarnerf.nerf.fused import _DP_LEVEL_WINDOWS
        enc = model.encoder
        off = enc._offsets_host
        wins = _DP_LEVEL_WINDOWS if enc.num_levels == 16 else ((0, enc.num_levels),)
        plan = [{"levels": [int(l0), int(l1)], "rows": int(off[l1] - off[l0]), "bytes_fp16": int(off[l1] - off[l0]) * 2 * 2,
                 "overlaps": "the reduce pass of the following window(s)" if i + 1 < len(wins) else
                             "the MLP gradients' all-reduce and the loss-scale bookkeeping (waited for at the table optimizer)"}
                for i, (l0, l1) in enumerate(wins)]
        mlp_bytes = sum(p.numel() for p in trainer.params) * 4
        comm = {"ms_per_step_inclusive": round(1e3 * elapsed / args.steps, 3),
                "ms_per_step_without_allreduce": round(1e3 * t_nc, 3),
                "allreduce_exposed_ms": round(1e3 * (elapsed / args.steps - t_nc), 3),
                "payload_bytes": {"table_gradient_fp16": sum(w["bytes_fp16"] for w in plan), "mlp_gradients_fp32": mlp_bytes,
                                  "per_rank_on_the_wire_ring_allreduce": int(2 * (world - 1) / world *
                                                                           (sum(w["bytes_fp16"] for w in plan) + mlp_bytes))},
                "window_plan": plan,
                "backend": "nccl (RCCL)" if parallel.backend() == "nccl" else str(parallel.backend()),
                "payload": "hash-table gradient fp16, SUM over ranks, one all-reduce per level window (the scatter pass runs once, "
                           "the reduce pass per window) + the MLP gradients fp32 in one flat buffer"}

    if rank != 0:
        return
    # SURVEY 8(d) asks for >= 100 steps and the median: with fewer steps the timed region is tens of milliseconds and one
    # region moves by ~2 % against the next, so the reported step time is the MEDIAN of the three regions of K steps each
    # (ms_per_step_repeats lists all three, the first being the region bracketed right after the warm-up)
    elapsed_first = elapsed
    if args.steps < 100:
        elapsed = sorted(spread)[1] * args.steps / 1e3
    rays_total = args.rays * world * args.steps
    kernels = event_table(timers)
    if "lnh_grid_encode_backward_ws_begin" in kernels:  # DP: one scatter (begin) + the window reduces (finish) = one logical launch
        kb, kf = kernels.pop("lnh_grid_encode_backward_ws_begin"), kernels.pop("lnh_grid_encode_backward_ws_finish")
        tot = kb["total_ms"] + kf["total_ms"]
        kernels["lnh_grid_encode_backward_ws"] = {"calls": args.steps, "total_ms": round(tot, 3),
                                                  "avg_us": round(1e3 * tot / args.steps, 2),
                                                  "points": args.rays * (NUM_STEPS + UPSAMPLE) * args.steps,
                                                  "note": "begin (scatter pass) + one finish call (reduce pass) per level window, per step",
                                                  "begin_avg_us": kb["avg_us"], "finish_avg_us": kf["avg_us"]}
    fwd_names = ("lnh_grid_encode_forward", "lnh_grid_encode_forward_mapped")
    bwd_names = ("lnh_grid_encode_backward", "lnh_grid_encode_backward_ws")
    dom = max(fwd_names + bwd_names, key=lambda k: kernels.get(k, {}).get("total_ms", 0))
    pmc_file, pmc, pmc_lib = _latest_pmc()
    this_lib = lib_sha16()
    # (round 4: the plain level classes have their own scatter kernel; older PMC files know k_grid_bwd_scatter only)
    scatter_name = "k_grid_bwd_scatter_plain" if "k_grid_bwd_scatter_plain" in pmc else "k_grid_bwd_scatter"
    pmc_names = {"lnh_grid_encode_backward_ws": (scatter_name, "k_grid_bwd_reduce"),
                 "lnh_grid_encode_forward_mapped": ("k_grid_forward",), "lnh_grid_encode_forward": ("k_grid_forward",)}

    def hbm_roofline(name):
        k = kernels[name]
        per_pt = GRID_FWD_BYTES if "forward" in name else GRID_BWD_BYTES
        avg_points = k["points"] / k["calls"]
        achieved = per_pt * k["points"] / (k["total_ms"] * 1e-3) / 1e9  # GB/s over all launches of that entry point
        # HBM bytes per launch from the newest committed PMC pass of this same command (profiles/collect.sh);
        # bench.py itself cannot run rocprofv3.  Scaled to this call's points (the PMC pass profiles whole launches).
        traffic, src = None, None
        parts = [pmc.get(n, {}).get("hbm_bytes_per_launch") for n in pmc_names.get(name, ())]
        if not (pmc_file and parts and all(p is not None for p in parts)):
            src = "no committed PMC pass (profiles/r*_pmc.json) covers this entry point: traffic not reported"
        elif args.rays != 4096 or args.table != "init" or args.patch.lower() != "1x1" or args.dp_windows:
            # the PMC passes profile the default command (4096 rays, fresh table): bytes per launch of another workload
            # are not that number scaled — say so instead of quoting it
            src = f"{pmc_file} was collected on the default workload (4096 rays, fresh table, 1x1 rays): not quoted for this one"
        else:
            traffic, src = int(sum(parts)), f"{pmc_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: " + \
                " + ".join(pmc_names[name]) + ")"
            if pmc_lib is None:
                src += "; that file does not record which library it profiled"
            elif pmc_lib != this_lib:
                src += f"; COLLECTED ON ANOTHER BUILD of the library (sha256 {pmc_lib} there, {this_lib} running now): " \
                       "re-run profiles/collect.sh"
            else:
                src += f"; same library as the one running now (sha256 {this_lib})"
        return {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src,
                "traffic_lib_matches": (pmc_lib == this_lib) if (traffic is not None and pmc_lib is not None) else None,
                "algorithmic_bytes_per_launch": int(per_pt * avg_points), "bytes_per_point": per_pt,
                "points_per_launch": int(avg_points), "avg_launch_us": k["avg_us"]}

    # MFMA utilisation of the MLP kernels: BASELINE.md §4 flops (6 144 sigma + 22 528 * mask fraction colour per sample
    # forward, backward = 2x) over the summed HIP-event time of the four MLP entry points, against the dense fp16 peak
    pts = args.rays * (NUM_STEPS + UPSAMPLE)

    def mfma_entry(tm, mf, n):
        tm = {k: v for k, v in tm.items() if k in mlp_calls}  # (the fixed-state passes time the encoder calls as well)
        ms = sum(v["total_ms"] for v in tm.values()) / max(n, 1)
        fl = 3 * pts * (SIGMA_FLOPS + COLOR_FLOPS * mf)
        fl_exec = 3 * pts * (SIGMA_FLOPS + COLOR_FLOPS_EXECUTED * mf)
        tf_ = fl / (ms * 1e-3) / 1e12 if ms else 0.0
        tf_exec = fl_exec / (ms * 1e-3) / 1e12 if ms else 0.0
        return {"achieved": round(tf_, 1), "frac": round(tf_ / MFMA_PEAK_TFLOPS, 4), "flops_per_step": int(fl),
                "mask_fraction": round(mf, 4), "mlp_kernel_ms_per_step": round(ms, 4),
                "executed_tflops": round(tf_exec, 1), "executed_frac": round(tf_exec / MFMA_PEAK_TFLOPS, 4),
                "per_kernel_us": {k: v["avg_us"] for k, v in tm.items()}}

    # the headline MFMA figure is the one of THIS run (round 4 headlined the favourable fixed state; the fixed states stay
    # beside it so that the number can be compared across rounds whatever --steps / --warmup the caller chose)
    as_run = mfma_entry(mlp_timers, mask_frac, n_prof)
    roofline_mfma = dict({"bound": "mfma", "kernel": "+".join(sorted(mlp_timers)), "peak": MFMA_PEAK_TFLOPS,
                          "unit": "TFLOP/s", "state": "as_run"}, **as_run)
    roofline_mfma["states"] = dict({"as_run": dict(as_run, note=f"the {n_prof} steps after the timed region of THIS run "
                                                                   f"(--warmup {args.warmup} --steps {args.steps}): the mask "
                                                                   "fraction depends on how far training got")},
                                   **{k: mfma_entry(*v) for k, v in mfma_states.items()})
    roofline_mfma["note"] = ("flops per SURVEY 8(d): 3 x points x (6144 + 22528 x mask fraction), 'frac' = that nominal count over "
                             "the dense fp16 peak; 'executed_frac' counts the colour head at the K = 16 per sample the kernels "
                             "really multiply (the 75 direction columns are folded into a per-ray term) and is the hardware "
                             "figure; the headline is the state 'as_run' (the steps right after the timed region of this run); "
                             "'fresh_frozen' = fresh table with the optimizer frozen at lr = 0 (every step sees the initial "
                             "field, mask ~ 1.0), 'trained' = after --pretrain-steps untimed steps on the analytic scene (mask "
                             "~ 0.1); the MLP kernels also read the encoder output and write activations: at 61 flop/B the sigma "
                             "net is HBM-bound by construction (DESIGN.md)")

    # ---- path-level HBM roofline (north_star states its 60 % bar on encode + composite together): algorithmic bytes of the
    #      encode forward (588 B), the encode backward (1100 B) and the compositing (44 B) per sample point of a step over
    #      the summed HIP-event time of the kernels that do that work.  The forward compositing sums are formed inside the
    #      fused forward tail (merged weights + colour head + sums, one kernel per ray): 'frac' charges that kernel's WHOLE
    #      time to the path (conservative), 'frac_encode_only' is encode forward + backward alone.
    def path_entry(tm, n, note):
        def ms_of(names):
            return sum(tm[k]["total_ms"] for k in names if k in tm) / max(n, 1)
        t_fwd = ms_of(fwd_names)
        t_bwd = ms_of(bwd_names + ("lnh_grid_encode_backward_ws_begin", "lnh_grid_encode_backward_ws_finish"))
        t_cb, t_tail = ms_of(("lnh_lidar_composite_backward",)), ms_of(("lnh_lidar_color_composite_forward" + sfx,))
        b_enc = (GRID_FWD_BYTES + GRID_BWD_BYTES) * pts
        b_all = b_enc + COMPOSITE_BYTES * pts
        t_all = t_fwd + t_bwd + t_cb + t_tail
        if not (t_fwd and t_bwd):
            return None
        return {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "achieved": round(b_all / (t_all * 1e-3) / 1e9, 1), "frac": round(b_all / (t_all * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "frac_encode_only": round(b_enc / ((t_fwd + t_bwd) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_step": int(b_all), "bytes_per_point": GRID_FWD_BYTES + GRID_BWD_BYTES + COMPOSITE_BYTES,
                "kernel_ms_per_step": {"encode_forward": round(t_fwd, 4), "encode_backward": round(t_bwd, 4),
                                       "composite_backward": round(t_cb, 4), "fused_forward_tail": round(t_tail, 4)},
                "note": note}

    roofline_path = path_entry(kernels, args.steps, "as run: the launch-by-launch region of this run (the timed workload)")
    if roofline_path is not None:
        roofline_path["state"] = "as_run"
        roofline_path["states"] = {k: path_entry(v[0], v[2], "fixed state, see roofline_mfma.note") for k, v in mfma_states.items()
                                   if path_entry(v[0], v[2], "") is not None}
    result = {
        "metric": "train rays/sec (encode+MLP+composite+bwd), KITTI-360 66x1030",
        "value": round(rays_total / elapsed, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "host_enqueue_ms_per_step": round(host_ms, 3),  # < ms_per_step: the launch queue runs ahead of the device
        "dtype": f"{'bf16' if args.mlp_dtype == 'bf16' else 'f16'} (fp16 hash tables + {args.mlp_dtype} MFMA MLP, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": "KITTI-360 seq 1908 shaped: hash-grid L=16 F=2 (2^19 rows, res 16..32768) + 64-wide "
                               "fused MLPs, 66x1030 range image", "rays_per_gpu_per_step": args.rays,
                   "samples_per_ray": NUM_STEPS + UPSAMPLE, "parallelism": f"dp{world}",
                   "table": "fresh init, random per-ray ground truth" if args.table == "init" else
                            f"trained-like: {args.pretrain_steps} untimed steps on the analytic scene (sphere + ground plane)",
                   "patch": args.patch, "dp_windows_forced": bool(args.dp_windows),
                   "optimizer": "Adam + dynamic loss scaling + lr schedule, in the timed region (lnh_train_check + lnh_train_step: the hash table and the MLP tensors in two launches)", "final_loss": round(loss_val, 5)},
        "ms_per_step_repeats": spread,  # three timed regions of K steps each; `value` / `ms_per_step` = their median when K < 100
        "ms_per_step_100": ms_100 if ms_100 is not None else round(1e3 * elapsed / args.steps, 4),
        "value_100": round(args.rays * world / ((ms_100 if ms_100 is not None else 1e3 * elapsed / args.steps) * 1e-3), 1),
        "value_100_note": "SURVEY 8(d) protocol figure: one timed region of 100 steps (the K-step region itself when K >= 100), "
                          "same bracketing as `value`",
        "ms_per_step_first_region": round(1e3 * elapsed_first / args.steps, 3),
        "roofline": hbm_roofline(dom),
        "roofline_fwd": hbm_roofline(max(fwd_names, key=lambda k: kernels.get(k, {}).get("total_ms", 0))),
        "roofline_mfma": roofline_mfma,
        "roofline_path": roofline_path,
        "kernels": kernels,
    }
    if getattr(trainer, "graph_error", None):
        result["graph"] = {"error": trainer.graph_error, "note": "capture failed: every step was issued launch by launch"}
    if graph_info is not None:
        result["graph"] = graph_info
        result["config"]["launch"] = "hipGraph replay of the whole step in the timed region (LidarTrainer graph mode)" + (
            f"; data parallel: the step's collectives ({parallel.backend()}: table-gradient windows, MLP gradients) are "
            "captured inside the graph, every rank replays its own copy" if trainer.dp else "")
    elif world > 1:
        result["config"]["launch"] = ("launch by launch on every rank (" + ("--no-graph" if args.no_graph else
                                      f"backend {parallel.backend()}: only RCCL collectives can be captured") + ")")
    if comm is not None:
        result["comm"] = comm
    if pretrain is not None:
        result["pretrain"] = pretrain
    # secondary number of SURVEY §8(d): full-frame evaluation (67 980 rays of a 66 x 1030 range image, staged in
    # chunks of 4096, no perturbation, no gradient) — reported beside the headline metric, never instead of it
    if world == 1 and not args.no_eval:
        model.eval()
        frame = make_batch(poses, 0, 66 * 1030, rank, device)
        with torch.no_grad(), torch.autocast("cuda", dtype=trainer.amp_dtype):
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
        result["cpu_baseline"] = cpu_baseline(full=not args.cpu_baseline_quick)
    _flush_c_stdout()  # (RCCL prints its version banner through C stdio: it must not land BEHIND the line)
    print(json.dumps(result), flush=True)


def _shutdown():
    """Leave the process group in an orderly way (RCCL's teardown at interpreter exit without it warns, and a crash there
    would turn a finished run into a non-zero exit code)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            torch.cuda.synchronize()
            dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


if __name__ == "__main__":
    try:
        main()
    finally:
        _flush_c_stdout()
        sys.stdout.flush()
        _shutdown()
