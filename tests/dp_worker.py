"""Worker of tests/test_dp_gpu.py (not a test module): run under torch.distributed.run with 2 ranks (gloo, both on GPU 0).
The windowed, overlapped all-reduce of the table gradient must reproduce the single-process gradient when every rank
sees the same rays, and data-parallel training steps must leave identical tables on all ranks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lidar-nerf_amd"))
os.environ.setdefault("LNH_DIST_BACKEND", "gloo")
import torch, bench
from lidarnerf import parallel
from lidarnerf.nerf.train_step import LidarTrainer
rank, local, world = parallel.init_from_env()
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
torch.manual_seed(0)
model = bench.build_model(device)
parallel.broadcast_parameters(model)
poses = bench.synthetic_frames(60, device)
batch = bench.make_batch(poses, 0, 512, 0, device)   # same rays on every rank
tp = model.encoder.embeddings
def grads(dp):
    model.zero_grad(set_to_none=True)
    saved = parallel.world_size
    if not dp: parallel.world_size = lambda: 1
    try:
        torch.manual_seed(7)
        with torch.autocast("cuda", dtype=torch.float16):
            out = model.render(batch[0], batch[1], cal_lidar_color=True, staged=False, perturb=True, num_steps=768, upsample_steps=64)
            loss = out["depth_lidar"].sum() * 64 + out["image_lidar"].sum()
        loss.backward()
    finally:
        parallel.world_size = saved
    return tp.grad.detach().clone()
g_single = grads(False)
g_dp = grads(True)
d = (g_dp - g_single).abs().max().item()
rel = d / (g_single.abs().max().item() + 1e-12)
print(f"rank {rank}: max abs diff {d:.3e} rel {rel:.3e} nonzero rows {(g_single.abs().sum(1) > 0).sum().item()}")
assert rel < 2e-3
tr = LidarTrainer(model, fp16=True, scale=bench.SCALE, world_size=world, render_kwargs=dict(num_steps=768, upsample_steps=64))
for s in range(3): l = tr.step(*batch)
chk = tp.detach().double().sum()
all_chk = [torch.zeros_like(chk) for _ in range(world)]
torch.distributed.all_gather(all_chk, chk)
same = all(float(c) == float(all_chk[0]) for c in all_chk)
print(f"rank {rank}: table checksum after 3 DP steps {chk.item():.9f} identical across ranks: {same}")
assert same
print(f"rank {rank}: DP-OK")
