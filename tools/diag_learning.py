#!/usr/bin/env python3
"""Distribution of the learning tests' held-out depth error (tests/test_zzz_learning_gpu.py) over repeated runs.

    python tools/diag_learning.py --patch 2x8 --runs 6 --steps 800 --every 100 [--variant fused|nozero|torchopt|graph]

Per run and checkpoint: median held-out depth error (m), dynamic loss scale, optimizer steps taken (skipped steps do not
count).  Variants: `fused` = the product path (fused table optimizer, LNH_BWD_TABLE_ZERO); `nozero` = the reduce pass
reads-adds-stores (flag off); `torchopt` = torch.optim.Adam + GradScaler on an fp32 .grad; `graph` = captured step.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-nerf_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def run(patch, steps, every, variant, seed, lr):
    import bench
    from lidarnerf import _hip
    from lidarnerf.nerf.train_step import LidarTrainer
    dev = torch.device("cuda", 0)
    torch.manual_seed(seed)
    model = bench.build_model(dev)
    if variant == "nozero":
        _hip.LNH_BWD_TABLE_ZERO = 0
    probe = None
    if variant == "probe":
        # every table backward twice: as the step runs it (caller-cleared buffers, LNH_BWD_TABLE_ZERO) and again into a second
        # zeroed table without flags; rows that differ + the fill of the pools / spill lists are recorded
        from lidarnerf.nerf import fused
        import ctypes as C
        probe = dict(mismatch_steps=0, mismatch_rows=0, max_fill=0.0, max_spill=0, spill_cap=0, calls=0)
        orig = fused._grid_bwd

        def both(g_feat, x01, g_table16, enc, B, ws=None, flags=0):
            orig(g_feat, x01, g_table16, enc, B, ws, flags)
            L = enc.num_levels
            off = enc._offsets_host
            out4 = (C.c_uint32 * 4)()
            nb, cap = [], []
            for l in range(L):
                _hip.lib().lnh_grid_backward_plan_info(off.data_ptr(), B, 3, 2, L, enc.log2_scale, enc.base_resolution, 0, 0,
                                                       _hip.LNH_F16, l, out4)
                nb.append(out4[0]); cap.append(out4[1])
            nbt = sum(nb)
            wsbuf, _ = fused._grid_bwd_workspace(g_feat.device, enc, B)
            head = wsbuf[: (2 * nbt + L) * 4].view(torch.int32).clone()
            cur, sp = head[:nbt].cpu().numpy().astype("int64"), head[nbt:nbt + L].cpu().numpy().astype("int64")
            b0 = 0
            for l in range(L):
                probe["max_fill"] = max(probe["max_fill"], float(cur[b0:b0 + nb[l]].max()) / cap[l])
                b0 += nb[l]
            probe["max_spill"] = max(probe["max_spill"], int(sp.max()))
            probe["spill_cap"] = max(B * 8 // 16, 1 << 20)
            ref = torch.zeros_like(g_table16)
            orig(g_feat, x01, ref, enc, B, None, 0)
            bad = int((ref.view(torch.int16) != g_table16.view(torch.int16)).any(-1).sum())
            probe["calls"] += 1
            if bad:
                probe["mismatch_steps"] += 1
                probe["mismatch_rows"] += bad
        fused._grid_bwd = both
    tr = LidarTrainer(model, lr=lr, iters=30000, fp16=True, scale=bench.SCALE,
                      render_kwargs=dict(num_steps=768, upsample_steps=64),
                      fused_table_optimizer=variant != "torchopt", graph=variant == "graph")
    poses = bench.synthetic_frames(60, dev)
    batches = [bench.make_batch(poses, s, 4096, 0, dev, patch, "analytic") for s in range(60)]
    held = bench.make_batch(poses, 30, 4096, 1, dev, (1, 1), "analytic")
    torch.manual_seed(seed)

    def depth_error_m():
        model.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            out = model.render(held[0], held[1], cal_lidar_color=True, staged=False, perturb=False, num_steps=768,
                               upsample_steps=64)
        model.train()
        err = (out["depth_lidar"][0].float() - held[2][0, :, 2]).abs() / bench.SCALE
        return float(err.median()), float(err.mean())

    rows = [dict(step=0, e_med=depth_error_m()[0])]
    kw = {} if patch == (1, 1) else {"patch": patch}
    last = None
    trace = []  # (step, loss scale) whenever the scale has moved; fused optimizer only
    prev_scale = None
    for s in range(steps):
        last = tr.step(*batches[s % 60], **kw)
        if variant == "sync":
            torch.cuda.synchronize()
        if variant == "sync" and tr.table is not None:
            sc = float(tr.loss_scale)
            if sc != prev_scale:
                trace.append((s + 1, sc, round(float(last), 2)))
                prev_scale = sc
        if variant == "trace":  # per step, without a sync: loss, loss scale, largest |gradient| of the table (scaled) and of the MLPs
            g16 = getattr(tr.table, "_lnh_grad16", None)
            arena = getattr(tr.table, "_lnh_small_arena", None)
            trace.append(torch.stack([last.detach().float(), tr.loss_scale.clone(),
                                      g16.float().abs().max() if g16 is not None else last.detach() * 0,
                                      arena.abs().max() if arena is not None else last.detach() * 0]))
        if (s + 1) % every == 0:
            med, mean = depth_error_m()
            row = dict(step=s + 1, e_med=round(med, 3), e_mean=round(mean, 3), loss=round(float(last), 3))
            if tr.table is not None:
                row.update(scale=float(tr.loss_scale), taken=tr.steps_taken())
            else:
                row.update(scale=float(tr.scaler.get_scale()))
            rows.append(row)
    if probe is not None:
        rows[-1]["probe"] = probe
    if trace and variant == "trace":
        t = torch.stack(trace).cpu().numpy()
        rows[-1]["probe"] = [[i + 1] + [float(f"{x:.4g}") for x in t[i]] for i in range(len(t))]
    elif trace:
        rows[-1]["probe"] = trace
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--patch", default="2x8")
    ap.add_argument("--runs", type=int, default=4)
    ap.add_argument("--steps", type=int, default=800)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--variant", default="fused")
    ap.add_argument("--lr", type=float, default=1e-2)
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--same-seed", action="store_true", help="every run with seed0 (what differs is the hardware's order)")
    a = ap.parse_args()
    patch = tuple(int(v) for v in a.patch.split("x"))
    for r in range(a.runs):
        seed = a.seed0 if a.same_seed else a.seed0 + r
        rows = run(patch, a.steps, a.every, a.variant, seed, a.lr)
        print(json.dumps(dict(variant=a.variant, patch=a.patch, run=r, seed=seed,
                              e=[(x["step"], x["e_med"]) for x in rows],
                              scale=[x.get("scale") for x in rows[1:]], taken=[x.get("taken") for x in rows[1:]],
                              loss=[x.get("loss") for x in rows[1:]], probe=rows[-1].get("probe"))), flush=True)


if __name__ == "__main__":
    main()
