"""Checkpoint format of the reference Trainer (lidarnerf/nerf/utils.py:1449-1568): keys, the optimizer state in the
layout of Adam(model.get_params(lr)) — a stock optimizer built the way main_lidarnerf.py:389-391 builds it must load it —
and a save -> load round trip.  CPU (no render step involved); the fused-table variant is covered on the GPU."""
import os

import torch


def _model():
    from lidarnerf.nerf.network import NeRFNetwork
    torch.manual_seed(0)
    return NeRFNetwork(encoding="hashgrid", desired_resolution=256, log2_hashmap_size=12, bound=1, min_near=0.01,
                       min_near_lidar=0.01)


def test_checkpoint_roundtrip_and_reference_layout(tmp_path):
    from lidarnerf.nerf.train_step import LidarTrainer
    m = _model()
    tr = LidarTrainer(m, lr=1e-2, fp16=False)
    # give every parameter a gradient and take two optimizer steps so that there is state to save
    for _ in range(2):
        tr.optimizer.zero_grad()
        for i, p in enumerate(tr.params):
            p.grad = torch.full_like(p, 1e-3 * (i + 1))
        tr.optimizer.step()
        tr.scheduler.step()
    tr.epoch, tr.global_step = 3, 180
    path = tr.save_checkpoint(os.path.join(tmp_path, "ngp_ep0003.pth"))
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"epoch", "global_step", "stats", "optimizer", "lr_scheduler", "scaler", "model"}
    assert set(ck["model"]) == set(m.state_dict()) and "encoder.embeddings" in ck["model"]
    # the reference's optimizer loads it as is
    ref_opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    ref_opt.load_state_dict(ck["optimizer"])
    table_state = ref_opt.state[m.encoder.embeddings]
    assert float(table_state["step"]) == 2 and float(table_state["exp_avg"].abs().max()) > 0
    # round trip into a fresh trainer
    m2 = _model()
    with torch.no_grad():
        for p in m2.parameters():
            p.add_(1.0)
    tr2 = LidarTrainer(m2, lr=1e-2, fp16=False)
    missing, unexpected = tr2.load_checkpoint(path)
    assert not missing and not unexpected
    assert tr2.epoch == 3 and tr2.global_step == 180
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    sa, sb = tr.optimizer.state_dict()["state"], tr2.optimizer.state_dict()["state"]
    assert set(sa) == set(sb)
    for i in sa:
        assert torch.equal(sa[i]["exp_avg"], sb[i]["exp_avg"]) and float(sa[i]["step"]) == float(sb[i]["step"])
    assert tr2.scheduler.state_dict()["last_epoch"] == tr.scheduler.state_dict()["last_epoch"]
    # a bare model state dict is accepted too (utils.py:1524-1527)
    torch.save(m.state_dict(), os.path.join(tmp_path, "bare.pth"))
    tr2.load_checkpoint(os.path.join(tmp_path, "bare.pth"))


def test_lr_scheduler_cross_resume_both_directions(tmp_path):
    """The `lr_scheduler` entry is in the reference layout too: the reference builds LambdaLR over
    Adam(model.get_params(lr)) — 6 parameter groups (main_lidarnerf.py:389-410) — while LidarTrainer steps one merged group.
    A stock scheduler must load our file and step; our trainer must load a reference-shaped file and step."""
    from lidarnerf.nerf.train_step import LidarTrainer
    iters = 1000
    lam = lambda it: 0.1 ** min(it / iters, 1)  # noqa: E731  (main_lidarnerf.py:408-410)
    m = _model()
    tr = LidarTrainer(m, lr=1e-2, iters=iters, fp16=False)
    for _ in range(7):
        tr.optimizer.zero_grad()
        for p in tr.params:
            p.grad = torch.full_like(p, 1e-3)
        tr.optimizer.step()
        tr.scheduler.step()
    path = tr.save_checkpoint(os.path.join(tmp_path, "a.pth"))
    ck = torch.load(path, weights_only=False)
    n_ref_groups = len(m.get_params(1e-2))
    assert len(ck["lr_scheduler"]["base_lrs"]) == n_ref_groups == 6
    assert len(ck["lr_scheduler"]["_last_lr"]) == 6 and len(ck["lr_scheduler"]["lr_lambdas"]) == 6
    # -> reference side: stock optimizer + LambdaLR, as Trainer.load_checkpoint (utils.py:1549-1561) restores them
    ref_opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    ref_sched = torch.optim.lr_scheduler.LambdaLR(ref_opt, lam)
    ref_opt.load_state_dict(ck["optimizer"])
    ref_sched.load_state_dict(ck["lr_scheduler"])
    for p in m.parameters():
        p.grad = torch.full_like(p, 1e-3)
    ref_opt.step()
    ref_sched.step()  # (raised "zip() argument 2 is shorter than argument 1" with a 1-group state)
    assert ref_sched.last_epoch == 8
    want = 1e-2 * lam(8)
    assert all(abs(g["lr"] - want) < 1e-12 for g in ref_opt.param_groups)
    # <- our side: a reference-shaped checkpoint (6 groups in optimizer and scheduler)
    ref_ck = {"epoch": 1, "global_step": 8, "stats": tr.stats, "model": m.state_dict(),
              "optimizer": ref_opt.state_dict(), "lr_scheduler": ref_sched.state_dict(), "scaler": {}}
    torch.save(ref_ck, os.path.join(tmp_path, "ref.pth"))
    tr2 = LidarTrainer(_model(), lr=1e-2, iters=iters, fp16=False)
    tr2.load_checkpoint(os.path.join(tmp_path, "ref.pth"))
    assert tr2.scheduler.last_epoch == 8 and len(tr2.scheduler.base_lrs) == len(tr2.optimizer.param_groups)
    for p in tr2.params:
        p.grad = torch.full_like(p, 1e-3)
    tr2.optimizer.step()
    tr2.scheduler.step()
    assert abs(tr2.optimizer.param_groups[0]["lr"] - 1e-2 * lam(9)) < 1e-12


def test_occupancy_bookkeeping_in_checkpoint(tmp_path):
    """cuda_ray runs: mean_count / mean_density / iter_density / local_step travel with the checkpoint (torch-ngp, the
    origin of the occupancy path, saves the first two), so a resumed run neither re-sweeps the full grid 16 times nor
    allocates N x 1024 sample buffers."""
    from lidarnerf.nerf.network import NeRFNetwork
    from lidarnerf.nerf.train_step import LidarTrainer
    torch.manual_seed(0)
    kw = dict(encoding="hashgrid", desired_resolution=256, log2_hashmap_size=12, bound=1, min_near=0.01, min_near_lidar=0.01,
              cuda_ray=True)
    m = NeRFNetwork(**kw)
    m.mean_count, m.mean_density, m.iter_density, m.local_step = 345678, 0.0123, 40, 7
    tr = LidarTrainer(m, lr=1e-2, fp16=False)
    path = tr.save_checkpoint(os.path.join(tmp_path, "occ.pth"))
    m2 = NeRFNetwork(**kw)
    tr2 = LidarTrainer(m2, lr=1e-2, fp16=False)
    tr2.load_checkpoint(path)
    assert (m2.mean_count, m2.mean_density, m2.iter_density, m2.local_step) == (345678, 0.0123, 40, 7)


def test_sharded_table_guards(tmp_path):
    """Sharded table optimizer: between steps the fp32 master table of a rank is current on its own rows only.  Everything
    that would read (or persist) the parameter itself refuses until gather_table_state() has completed it, and a trainer
    told about more ranks than the process group has says so at construction."""
    import pytest
    from lidarnerf.nerf import fused
    from lidarnerf.nerf.train_step import LidarTrainer
    m = _model()
    with pytest.raises(RuntimeError, match="process group"):
        LidarTrainer(m, lr=1e-2, fp16=False, world_size=2)
    emb = m.encoder.embeddings
    emb._lnh_master_stale = True
    with pytest.raises(RuntimeError, match="gather_table_state"):
        m.encoder(torch.zeros(4, 3))
    emb.grad = torch.zeros_like(emb)
    with pytest.raises(RuntimeError, match="gather_table_state"):
        m.encoder.grad_total_variation()
    # the fp16 shadow of a sharded table is never re-cast from a stale master
    emb._lnh_table16 = emb.detach().half().reshape(-1, 2).contiguous()
    emb._lnh_shard_optimizer, emb._lnh_table16_version = True, emb._version
    assert fused.table16_of(emb) is emb._lnh_table16
    with torch.no_grad():
        emb.mul_(1.0)                                  # an in-place write through torch: the version counter moves
    with pytest.raises(RuntimeError, match="sharded table optimizer"):
        fused.table16_of(emb)
    emb._lnh_master_stale = False                      # gather_table_state() / load_checkpoint: the master is whole again
    assert fused.table16_of(emb) is emb._lnh_table16
    # save_checkpoint(gather=False) on a rank whose master is stale must not write a file
    tr = LidarTrainer(m, lr=1e-2, fp16=False)
    tr.table, tr.sharded = emb, True
    emb._lnh_master_stale = True
    with pytest.raises(RuntimeError, match="gather_table_state"):
        tr.save_checkpoint(os.path.join(tmp_path, "x.pth"), gather=False)
    assert not os.path.exists(os.path.join(tmp_path, "x.pth"))
