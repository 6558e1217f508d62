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
