"""The weight-gradient contraction over the batch (dW = G^T A, B = 1 M rows): the HIP kernel lnh_mlp_wgrad (csrc/mlp_wgrad.hip)
next to the library forms it replaced — one GEMM, and the batched GEMM over 4096-row slices summed afterwards.
    python tools/bench_wgrad.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-nerf_amd"))
import torch
from lidarnerf import _hip
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return min(ts)
B = 1 << 20
for M, N in ((128, 128), (256, 256), (64, 64), (256, 32), (16, 256), (128, 48)):
    a = (torch.randn(B, M, device="cuda") * 0.1).half(); b = torch.randn(B, N, device="cuda").half()
    ref = (a.t().double() @ b.double())
    res = {}
    out = torch.zeros(M, N, device="cuda")
    def hip():
        _hip.call("lnh_mlp_wgrad", a.data_ptr(), b.data_ptr(), B, M, N, out.data_ptr(), *_hip.wgrad_ws("cuda"))
    t = timed(hip)
    out.zero_(); hip(); torch.cuda.synchronize()
    res["lnh_mlp_wgrad us"] = (round(t, 1), float((out.double() - ref).abs().max() / ref.abs().max()),
                               f"{(B * (M + N) * 2) / t / 1e6:.2f} TB/s", f"{2 * B * M * N / t / 1e6:.0f} TFLOP/s")
    def mm(): return torch.mm(a.t(), b, out_dtype=torch.float32)
    try: res["mm out f32"] = (round(timed(mm), 1), float((mm().double() - ref).abs().max() / ref.abs().max()))
    except Exception as e: res["mm out f32"] = str(e)[:60]
    rows = 4096; S = B // rows
    def bmm(): return torch.bmm(a.view(S, rows, M).transpose(1, 2), b.view(S, rows, N), out_dtype=torch.float32).sum(0)
    try: res[f"bmm f32 {rows}"] = (round(timed(bmm), 1), float((bmm().double() - ref).abs().max() / ref.abs().max()))
    except Exception as e: res[f"bmm f32 {rows}"] = str(e)[:60]
    print(M, N, res, flush=True)
