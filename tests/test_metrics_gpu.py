"""lidarnerf.metrics (HIP chamfer kernel + device-side meters) vs the CPU restatement."""
import numpy as np
import pytest
import torch

from oracle import c_oracle, convert_ref, metrics_ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m", [(1, 1), (257, 1025), (3000, 2100)])
def test_chamfer_nn_bit_exact(n, m):
    from lidarnerf import _hip
    rng = np.random.default_rng(n + m)
    a = rng.normal(size=(n, 3)).astype(np.float32)
    b = rng.normal(size=(m, 3)).astype(np.float32)
    if m > 10:
        b[7] = b[3]            # duplicate target: the first index must win
        a[0] = b[3]            # exact hit
    want_d, want_i = c_oracle.chamfer_nn(a, b)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    d = torch.empty(n, device="cuda")
    i = torch.empty(n, dtype=torch.int32, device="cuda")
    _hip.call("lnh_chamfer_nn", ta.data_ptr(), n, tb.data_ptr(), m, d.data_ptr(), i.data_ptr())
    assert np.array_equal(i.cpu().numpy(), want_i)
    assert np.array_equal(d.cpu().numpy(), want_d)


def test_chamfer_module_and_fscore():
    from lidarnerf.metrics import chamfer_3DDist, fscore
    rng = np.random.default_rng(5)
    a = rng.uniform(-1, 1, (2, 700, 3)).astype(np.float32)
    b = a[:, ::-1].copy() + rng.normal(scale=0.05, size=a.shape).astype(np.float32)
    d1, d2, i1, i2 = chamfer_3DDist()(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    assert d1.shape == (2, 700) and i1.dtype == torch.int32
    for k in range(2):
        wd1, wi1 = c_oracle.chamfer_nn(a[k], b[k])
        wd2, wi2 = c_oracle.chamfer_nn(b[k], a[k])
        assert np.array_equal(d1[k].cpu().numpy(), wd1) and np.array_equal(i2[k].cpu().numpy(), wi2)
        f, p, r = fscore(d1[k:k + 1], d2[k:k + 1], 0.01)
        assert abs(float(f[0]) - metrics_ref.fscore(wd1, wd2, 0.01)) < 1e-6
    with pytest.raises(RuntimeError):
        chamfer_3DDist()(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3))
    # symmetric, zero on identical clouds, size-independent: full 66x1030-sized clouds
    big = torch.rand(1, 60000, 3, device="cuda")
    d1, d2, i1, _ = chamfer_3DDist()(big, big)
    assert float(d1.max()) == 0.0 and float(d2.max()) == 0.0 and torch.equal(i1[0].long(), torch.arange(60000, device="cuda"))


def test_meters_match_restatement():
    from lidarnerf.metrics import DepthMeter, MAEMeter, PointsMeter, RMSEMeter
    rng = np.random.default_rng(9)
    H, W, K, scale = 66, 1030, (2.0, 26.9), 0.0107848535
    gt = (rng.uniform(2.0, 70.0, (1, H, W)) * (rng.uniform(size=(1, H, W)) > 0.2)).astype(np.float32)
    pred = (gt * rng.normal(1.0, 0.02, gt.shape) + (gt == 0) * (rng.uniform(size=gt.shape) > 0.97) * 5.0).astype(np.float32)
    r, m = RMSEMeter(), MAEMeter(intensity_inv_scale=2.0)
    r.update(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda())
    m.update(pred, gt)
    assert abs(r.measure() - metrics_ref.rmse(pred, gt)) < 1e-4 * metrics_ref.rmse(pred, gt)
    assert abs(m.measure() - metrics_ref.mae(pred, gt, 2.0)) < 1e-4 * metrics_ref.mae(pred, gt, 2.0)
    dm = DepthMeter(scale)
    dm.update(torch.from_numpy(pred * scale).cuda(), torch.from_numpy(gt * scale).cuda())
    want = metrics_ref.depth_errors((gt * np.float32(scale)) / np.float32(scale), (pred * np.float32(scale)) / np.float32(scale))
    np.testing.assert_allclose(dm.measure(), np.array(want), rtol=2e-4, atol=1e-6)
    pm = PointsMeter(scale, K)
    pm.update(torch.from_numpy(pred * scale).cuda(), torch.from_numpy(gt * scale).cuda())
    small = slice(0, 16)  # restate on a 16-row crop (the O(n*m) CPU loop is slow) and compare that crop
    pm2 = PointsMeter(1.0, K)
    pm2.update(torch.from_numpy(pred[:, small]).cuda(), torch.from_numpy(gt[:, small]).cuda())
    pl = convert_ref.pano_to_lidar_with_intensities(pred[0, small], np.zeros_like(pred[0, small]), K)[:, :3]
    gl = convert_ref.pano_to_lidar_with_intensities(gt[0, small], np.zeros_like(gt[0, small]), K)[:, :3]
    d1, _ = c_oracle.chamfer_nn(pl, gl)
    d2, _ = c_oracle.chamfer_nn(gl, pl)
    cd, f = pm2.measure()
    assert abs(cd - (d1.mean() + d2.mean())) < 1e-3 * (d1.mean() + d2.mean())
    assert abs(f - metrics_ref.fscore(d1, d2, 0.05)) < 2e-3
    assert pm.measure().shape == (2,) and "CD f-score" in pm.report()


def test_meters_match_the_reference_meters_g11():
    """The HIP meters against what the reference's own RMSEMeter / MAEMeter / DepthMeter and extern/fscore.py computed on the
    same frames (tests/golden/g11_metrics.npz, made by tests/golden/make_g11_metrics.py from the imported reference)."""
    import os
    from lidarnerf.metrics import DepthMeter, MAEMeter, RMSEMeter, fscore
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g11_metrics.npz"))
    scale = float(G["scale"])
    r, m, d = RMSEMeter(), MAEMeter(intensity_inv_scale=2.0), DepthMeter(scale)
    for k in range(3):
        p, g = torch.from_numpy(G[f"pred{k}"]).cuda(), torch.from_numpy(G[f"gt{k}"]).cuda()
        r.update(p, g)
        m.update(p, g)
        d.update(p * np.float32(scale), g * np.float32(scale))
    assert abs(r.measure() - float(G["rmse"])) <= 2e-5 * float(G["rmse"])
    assert abs(m.measure() - float(G["mae"])) <= 2e-5 * float(G["mae"])
    # rmse, a1, a2, a3 (fp32 device reductions against numpy's pairwise fp32 sums; a pixel exactly on a ratio threshold may
    # fall on either side: 1 of 17 000 pixels = 6e-5)
    np.testing.assert_allclose(np.asarray(d.measure())[:4], G["depth_measure"], rtol=2e-4, atol=1e-4)
    f, p_, r_ = fscore(torch.from_numpy(G["fs_d1"]).cuda(), torch.from_numpy(G["fs_d2"]).cuda(), float(G["fs_threshold"]))
    np.testing.assert_allclose(f.cpu().numpy(), G["fs_f"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(p_.cpu().numpy(), G["fs_p"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(r_.cpu().numpy(), G["fs_r"], rtol=1e-6, atol=1e-7)
