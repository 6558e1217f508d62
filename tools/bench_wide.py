"""Fused FFMLP kernels against the library-GEMM chain at 1 M points (profiles/r04_ffmlp_wide.txt): python tools/bench_wide.py"""
import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lidar-nerf_amd")]
from lidarnerf.ffmlp import FFMLP
from lidarnerf.ffmlp.ffmlp import gemm_mlp
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return min(ts)
B = 1 << 20
for hidden, layers, in_dim in ((64, 2, 32), (128, 2, 32), (128, 3, 64), (256, 2, 32), (256, 3, 64), (64, 5, 32)):
    m = FFMLP(in_dim, 16, hidden, layers).cuda()
    x = torch.randn(B, in_dim, device="cuda").half()
    nhm = layers - 1
    flops = 2 * B * (in_dim * hidden + nhm * hidden * hidden + hidden * 16)
    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            m.eval(); return m(x)
    def fb():
        m.train()
        xx = x.detach().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16):
            y = m(xx)
        y.backward(torch.ones_like(y))
    def chain_f():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return gemm_mlp(x, m.weights, in_dim, hidden, nhm, 0, 6)
    def chain_fb():
        xx = x.detach().requires_grad_(True); w = m.weights.detach().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16):
            y = gemm_mlp(xx, w, in_dim, hidden, nhm, 0, 6)
        y.backward(torch.ones_like(y))
    tf, tfb, cf, cfb = timed(fwd), timed(fb), timed(chain_f), timed(chain_fb)
    from lidarnerf import _hip
    _hip.enable_timers(["lnh_mlp_forward", "lnh_mlp_backward", "lnh_mlp_backward_data"])
    for _ in range(5): fb()
    torch.cuda.synchronize()
    ev = {k: min(a.elapsed_time(b) for a, b, _ in v) * 1e3 for k, v in _hip.disable_timers().items()}
    print("   entry points (us):", {k: round(v, 1) for k, v in ev.items()})
    print(f"hidden {hidden:3d} layers {layers} in {in_dim:3d}: fused fwd {tf:8.1f} us ({flops / tf / 1e6:6.1f} TFLOP/s)  fwd+bwd {tfb:8.1f} us | GEMM chain fwd {cf:8.1f} us  fwd+bwd {cfb:8.1f} us   [{B} points]")
