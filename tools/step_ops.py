#!/usr/bin/env python3
"""Which framework ops launch the small kernels of a training step: torch.profiler over 3 steps of the bench workload,
printed as (op -> kernels) in launch order for ONE step.   python tools/step_ops.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lidar-nerf_amd")]
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    from lidarnerf.nerf.train_step import LidarTrainer
    tr = LidarTrainer(model, lr=1e-2, iters=30000, fp16=True, scale=bench.SCALE)
    poses = bench.synthetic_frames(60, dev)
    for s in range(4):
        o, d, gt = bench.make_batch(poses, s, 4096, 0, dev)
        tr.step(o, d, gt)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    o, d, gt = bench.make_batch(poses, 5, 4096, 0, dev)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        tr.step(o, d, gt)
        torch.cuda.synchronize()
    # launch order: every top-level-ish CPU op that owns device time, with its shapes and innermost python frame
    evs = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.device_time_total > 0
                  and not any(c.device_time_total > 0 and c.device_type == torch.autograd.DeviceType.CPU
                              for c in e.cpu_children)),
                 key=lambda e: e.time_range.start)
    tot = 0.0
    for e in evs:
        if e.name.startswith("lnh_") or "Fused" in e.name:
            continue
        fr = [f for f in (e.stack or []) if "lidarnerf" in f or "bench.py" in f]
        tot += e.device_time_total
        print(f"{e.device_time_total:7.1f} us  {e.name[:40]:40s} {str(e.input_shapes)[:60]:60s} {fr[0][-70:] if fr else ''}")
    print("torch-op device time per step: %.1f us" % tot)

    # the same step under a dispatch mode: every aten op that reaches the device, with the innermost repository frame
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = str(func)
            if not any(k in name for k in ("view", "empty", "as_strided", "detach", "slice", "select", "expand", "t.default",
                                           "alias", "_unsafe_view", "reshape", "unsqueeze", "squeeze", "permute", "transpose")):
                fr = [f for f in traceback.extract_stack() if ("lidarnerf" in f.filename or "bench.py" in f.filename)]
                shp = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]
                loc = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "(autograd / optimizer)"
                print(f"  {name[:44]:44s} {str(shp)[:50]:50s} {loc}")
            return out

    with Log():
        tr.step(o, d, gt)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
