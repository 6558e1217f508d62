"""G11: the oracle's restatement of the evaluation meters (oracle/metrics_ref.py) against what the reference's OWN
RMSEMeter / MAEMeter / DepthMeter (nerf/utils.py:226-372) and extern/fscore.py computed on the same frames
(tests/golden/make_g11_metrics.py).  The SSIM entry of DepthMeter is not in the fixture (scikit-image is not installed where
the reference was run); the HIP meters are compared with the same fixture in tests/test_metrics_gpu.py."""
import os

import numpy as np

from oracle import metrics_ref

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g11_metrics.npz"))
FRAMES = [(G[f"pred{k}"], G[f"gt{k}"]) for k in range(3)]


def test_rmse_and_mae_meters():
    rm = np.mean([metrics_ref.rmse(p, g) for p, g in FRAMES])
    ma = np.mean([metrics_ref.mae(p, g, 2.0) for p, g in FRAMES])
    assert abs(rm - float(G["rmse"])) <= 1e-6 * float(G["rmse"])
    assert abs(ma - float(G["mae"])) <= 1e-6 * float(G["mae"])


def test_depth_meter_first_four_entries():
    s = np.float32(G["scale"])
    for k, (p, g) in enumerate(FRAMES):
        got = metrics_ref.depth_errors((g * s) / s, (p * s) / s)[:4]
        np.testing.assert_allclose(np.array(got, dtype=np.float64), G["depth_per_frame"][k], rtol=1e-6, atol=0)
    assert 0.3 < G["depth_measure"][1] < 1.0  # (a1: a meaningful share of the pixels inside the 1.25 ratio)


def test_fscore():
    for b in range(2):
        want = float(G["fs_f"][b])
        got = metrics_ref.fscore(G["fs_d1"][b], G["fs_d2"][b], float(G["fs_threshold"]))
        assert abs(got - want) <= 1e-6
    assert float(G["fs_f"][1]) == 0.0  # nothing below the threshold: 0 / 0 -> 0
