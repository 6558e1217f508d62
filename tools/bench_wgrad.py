"""The weight-gradient GEMM (contraction over the batch) as one GEMM and as a batched split-K GEMM: python tools/bench_wgrad.py"""
import torch, time
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    return min(ts)
B = 1 << 20
for M, N in ((128, 128), (256, 256), (64, 64), (256, 32), (16, 256)):
    a = torch.randn(B, M, device="cuda").half(); b = torch.randn(B, N, device="cuda").half()
    ref = (a.t().float() @ b.float())
    def mm(): return torch.mm(a.t(), b, out_dtype=torch.float32)
    res = {}
    try: res["mm out f32"] = (timed(mm), float((mm() - ref).abs().max() / ref.abs().max()))
    except Exception as e: res["mm out f32"] = str(e)[:60]
    for rows in (2048, 4096, 8192, 16384):
        S = B // rows
        def bmm(): return torch.bmm(a.view(S, rows, M).transpose(1, 2), b.view(S, rows, N), out_dtype=torch.float32).sum(0)
        def bmm16(): return torch.bmm(a.view(S, rows, M).transpose(1, 2), b.view(S, rows, N)).float().sum(0)
        try: res[f"bmm f32 {rows}"] = (round(timed(bmm), 1), float((bmm() - ref).abs().max() / ref.abs().max()))
        except Exception as e: res[f"bmm f32 {rows}"] = str(e)[:60]
        res[f"bmm f16 {rows}"] = (round(timed(bmm16), 1), float((bmm16() - ref).abs().max() / ref.abs().max()))
    print(M, N, res)
