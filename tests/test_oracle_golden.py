"""Oracle (oracle/*) vs golden vectors produced by the imported reference (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import encoders_ref, render_ref


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_g1_sample_pdf(golden_dir):
    g = _load(golden_dir, "g1_sample_pdf.npz")
    bins, w = torch.from_numpy(g["bins"]), torch.from_numpy(g["weights"])
    det = render_ref.sample_pdf(bins, w, 64, det=True)
    np.testing.assert_array_equal(det.numpy(), g["det"])
    rnd = render_ref.sample_pdf(bins, w, 64, det=False, u=torch.from_numpy(g["u"]))
    np.testing.assert_array_equal(rnd.numpy(), g["rnd"])


class _Stub(torch.nn.Module):
    def __init__(self, g):
        super().__init__()
        for n in ("a0", "a", "bump", "M", "Wc"):
            setattr(self, n, torch.nn.Parameter(torch.from_numpy(g[f"param_{n}"]).clone()))

    def density(self, x):
        r = x.norm(dim=-1)
        sigma = torch.exp(self.a0 + (x * self.a).sum(-1)) + self.bump[2] * torch.exp(
            -((r - self.bump[0]) / self.bump[1]) ** 2)
        return sigma, torch.tanh(x @ self.M.t() * 3.0)

    def color(self, x, d, mask, geo):
        rgbs = torch.zeros(mask.shape[0], 2, dtype=x.dtype)
        if not mask.any():
            return rgbs
        rgbs[mask] = torch.sigmoid(torch.cat([d[mask], geo[mask]], -1) @ self.Wc.t())
        return rgbs


@pytest.mark.parametrize("tag", ["eval", "train"])
def test_g2_renderer_run(golden_dir, tag):
    g = _load(golden_dir, "g2_renderer_run.npz")
    m = _Stub(g)
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    scale = 0.010784853507573345
    train = tag == "train"
    res = render_ref.run_lidar(torch.from_numpy(g["rays_o"]), torch.from_numpy(g["rays_d"]), m.density, m.color, aabb,
                               scale, 768, 64, perturb=train, training=train,
                               noise=torch.from_numpy(g["train_noise"]) if train else None,
                               u=torch.from_numpy(g["train_u"]) if train else None)
    np.testing.assert_allclose(res["depth_lidar"].detach().numpy(), g[f"{tag}_depth"][0], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(res["image_lidar"].detach().numpy(), g[f"{tag}_image"][0], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(res["weights_sum_lidar"].detach().numpy(), g[f"{tag}_ws"], rtol=1e-6, atol=1e-7)
    loss = ((res["depth_lidar"] * torch.from_numpy(g["cd"])).sum()
            + (res["image_lidar"] * torch.from_numpy(g["ci"][0])).sum()
            + (res["weights_sum_lidar"] * torch.from_numpy(g["cw"])).sum())
    loss.backward()
    for n in ("a0", "a", "bump", "M", "Wc"):
        np.testing.assert_allclose(getattr(m, n).grad.numpy(), g[f"{tag}_grad_{n}"], rtol=2e-4, atol=1e-5,
                                   err_msg=f"grad {n}")


def test_g4_freq_layout(golden_dir):
    g = _load(golden_dir, "g4_freq_encoder.npz")
    y = encoders_ref.freq_forward_exactcos(g["d"], 12)
    np.testing.assert_allclose(y, g["y"], rtol=0, atol=2e-6)
    # kernel form (cos as sin(x + pi/2) in float32): same layout, looser at high frequencies
    yk = encoders_ref.freq_forward(g["d"], 12)
    np.testing.assert_allclose(yk, g["y"], rtol=0, atol=3e-4)
    gd = encoders_ref.freq_backward(g["g"], g["y"], 3, 12)
    np.testing.assert_allclose(gd, g["gd"], rtol=2e-4, atol=2e-2)
    # torch twin used by the CPU baseline
    yt = render_ref.freq_encode_torch(torch.from_numpy(g["d"]), 12).numpy()
    np.testing.assert_allclose(yt, g["y"], rtol=0, atol=1e-6)


def test_g5_trunc_exp(golden_dir):
    g = _load(golden_dir, "g5_trunc_exp.npz")
    np.testing.assert_allclose(encoders_ref.trunc_exp_forward(g["x"]), g["y"], rtol=2e-7)
    np.testing.assert_allclose(encoders_ref.trunc_exp_backward(g["g"], g["x"]), g["gx"], rtol=2e-7)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = render_ref.trunc_exp(x)
    y.backward(torch.from_numpy(g["g"]))
    np.testing.assert_array_equal(y.detach().numpy(), g["y"])
    np.testing.assert_array_equal(x.grad.numpy(), g["gx"])


def test_g6_convert_oracle_matches_reference_bit_exactly(golden_dir):
    """oracle/convert_ref.py (vectorised) vs the reference's per-point loop (lidarnerf/convert.py:99-160, 194-237)."""
    from oracle import convert_ref
    g = _load(golden_dir, "g6_convert.npz")
    H, W, K = int(g["H"]), int(g["W"]), tuple(float(v) for v in g["K"])
    pano, inten = convert_ref.lidar_to_pano_with_intensities(g["pts"], H, W, K, 80)
    assert np.array_equal(pano, g["pano"]) and np.array_equal(inten, g["intensities"])
    assert (pano != 0).sum() > 10000
    back = convert_ref.pano_to_lidar_with_intensities(g["pano"].astype(np.float32),
                                                      g["intensities"].astype(np.float32), K)
    assert np.array_equal(back, g["back"])


# ---------------------------------------------------------------------------------------------- G7: BASELINE config 1
# The reference's own NeRFNetwork + NeRFRenderer.run (imported; tests/golden/make_golden.py g7): pins oracle/mlp_ref.py (the
# checker of the MFMA MLP kernels), oracle/encoders_ref.py (frequency encoder) and oracle/render_ref.py's RefFreqField
# (the cpu_baseline model of bench.py) to reference-PRODUCED numbers.
def _g7_stacks(g):
    return (("sigma", g["sig_in"], [g[f"w_sigma_net.{i}.weight"] for i in range(2)], g["sig_out"], g["sig_gout"],
             [g[f"sig_gw{i}"] for i in range(2)]),
            ("colour", np.concatenate([g["col_dir"], g["col_geo"]], 1),
             [g[f"w_lidar_color_net.{i}.weight"] for i in range(3)], g["col_pre"], g["col_gout"],
             [g[f"col_gw{i}"] for i in range(3)]))


def test_g7_mlp_oracle_matches_reference_linear_stacks(golden_dir):
    from oracle import mlp_ref
    g = _load(golden_dir, "g7_config1.npz")
    for name, x, mats, want, gout, gws in _g7_stacks(g):
        out, _ = mlp_ref.mlp_forward(x, mats, half=False)
        np.testing.assert_allclose(out, want, rtol=2e-5, atol=2e-5, err_msg=name)  # fp32 GEMMs there, float64 here
        gx, dW = mlp_ref.mlp_backward(x, mats, gout, half=False)
        for k, w in enumerate(gws):
            np.testing.assert_allclose(dW[k], w, rtol=0, atol=2e-6 * np.abs(w).max(), err_msg=f"{name} dW{k}")
        if name == "colour":
            np.testing.assert_allclose(gx[:, 75:], g["col_ggeo"], rtol=0, atol=2e-6 * np.abs(g["col_ggeo"]).max())


def test_g7_storage_model_distance(golden_dir):
    """What 16-bit storage costs on this fixture (the MODEL_DISTANCE table of tests/test_g7_config1_gpu.py): the oracle's
    model — exact dot products, one rounding per stored value — against the reference's fp32 numbers."""
    from oracle import mlp_ref
    g = _load(golden_dir, "g7_config1.npz")
    table = {(True, "sigma"): [1.6e-2, 4e-4], (True, "colour"): [1.3e-2, 1.4e-2, 4e-4],
             ("bf16", "sigma"): [4.8e-2, 3.4e-3], ("bf16", "colour"): [5.5e-2, 5.4e-2, 3.9e-3]}
    for half in (True, "bf16"):
        for name, x, mats, want, gout, gws in _g7_stacks(g):
            _, dW = mlp_ref.mlp_backward(x, mats, gout, half=half)
            for k, w in enumerate(gws):
                rel = np.linalg.norm(dW[k] - w) / np.linalg.norm(w)
                assert 0.5 * table[(half, name)][k] <= rel <= table[(half, name)][k], (half, name, k, rel)


def test_g7_freq_encoder_oracle(golden_dir):
    g = _load(golden_dir, "g7_config1.npz")
    np.testing.assert_allclose(encoders_ref.freq_forward(g["mlp_x"], 6), g["sig_in"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(encoders_ref.freq_forward(g["mlp_d"], 12), g["col_dir"], rtol=0, atol=2e-4)


@pytest.mark.parametrize("tag", ["eval", "train"])
def test_g7_config1_restatement_end_to_end(golden_dir, tag):
    """RefFreqField + run_lidar (what bench.py times as cpu_baseline) reproduce the reference's run: outputs, both losses
    and every weight gradient."""
    g = _load(golden_dir, "g7_config1.npz")
    f = render_ref.RefFreqField()
    with torch.no_grad():
        for i in range(2):
            f.sigma_net[i].weight.copy_(torch.from_numpy(g[f"w_sigma_net.{i}.weight"]))
        for i in range(3):
            f.lidar_color_net[i].weight.copy_(torch.from_numpy(g[f"w_lidar_color_net.{i}.weight"]))
    train = tag == "train"
    res = render_ref.run_lidar(torch.from_numpy(g["rays_o"][0]), torch.from_numpy(g["rays_d"][0]), f.density, f.color,
                               torch.tensor([-1.0, -1, -1, 1, 1, 1]), 0.010784853507573345, 768, 64, perturb=train,
                               training=train, noise=torch.from_numpy(g["train_noise"]) if train else None,
                               u=torch.from_numpy(g["train_u"]) if train else None)
    for k, kk in (("depth_lidar", "depth"), ("image_lidar", "image"), ("weights_sum_lidar", "ws")):
        np.testing.assert_allclose(res[k].detach().numpy().reshape(-1), g[f"{tag}_{kk}"].reshape(-1), rtol=1e-6, atol=1e-8)
    for loss_name in ("lin", "lidar"):
        f.zero_grad(set_to_none=True)
        if loss_name == "lin":
            loss = ((res["depth_lidar"] * torch.from_numpy(g["cd"])).sum()
                    + (res["image_lidar"] * torch.from_numpy(g["ci"][0])).sum()
                    + (res["weights_sum_lidar"] * torch.from_numpy(g["cw"])).sum())
        else:
            loss = render_ref.lidar_loss(res["depth_lidar"], res["image_lidar"], torch.from_numpy(g["gt"]))
        loss.backward(retain_graph=True)
        np.testing.assert_allclose(float(loss.detach()), float(g[f"{tag}_{loss_name}_loss"]), rtol=1e-6)
        for n, p in f.named_parameters():
            w = g[f"{tag}_{loss_name}_grad_{n}"]
            np.testing.assert_allclose(p.grad.numpy(), w, rtol=0, atol=2e-6 * np.abs(w).max(), err_msg=n)


@pytest.mark.parametrize("tag,patch", [("p1", None), ("p2x8", (2, 8)), ("p4x4", (4, 4))])
def test_g8_train_step_loss_restatements(golden_dir, tag, patch):
    """G8 = loss and d loss / d (depth, image) produced by the reference's OWN Trainer.train_step (nerf/utils.py:697-884,
    called unbound on a stub self; tests/golden/make_golden.py g8) for patch_size_lidar 1, [2, 8] and [4, 4].  Pins the two
    restatements of it: oracle/render_ref.py (lidar_loss + patch_grad_loss — the checker of the HIP loss kernels) and the
    product's own torch expressions (lidarnerf/nerf/train_step.py lidar_loss + patch_gradient_loss: what LidarTrainer runs
    on tensors the kernels do not cover).  fp32-tight: the same torch ops in a different grouping."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "lidar-nerf_amd"))
    from lidarnerf.nerf import train_step
    g = _load(golden_dir, "g8_train_step.npz")
    ad, ar, ai, ag = (float(v) for v in g["alphas"])
    scale = float(g["scale"])
    gt = torch.from_numpy(g["gt"])                                           # [1, N, 3]
    want_loss, want_gd, want_gi = float(g[f"{tag}_loss"]), g[f"{tag}_grad_depth"], g[f"{tag}_grad_image"]

    def check(loss, d, im):
        loss.backward()
        assert abs(float(loss) - want_loss) <= 2e-6 * abs(want_loss)
        np.testing.assert_allclose(d.grad.reshape(-1).numpy(), want_gd, rtol=1e-5, atol=1e-6 * np.abs(want_gd).max())
        np.testing.assert_allclose(im.grad.reshape(-1, 2).numpy(), want_gi, rtol=1e-5, atol=1e-9)

    # the oracle's restatement
    d = torch.from_numpy(g["depth"]).clone().requires_grad_(True)
    im = torch.from_numpy(g["image"]).clone().requires_grad_(True)
    loss = render_ref.lidar_loss(d, im, gt[0], ad, ar, ai)
    if patch:
        loss = loss + render_ref.patch_grad_loss(d, gt[0], patch[0], patch[1], scale, ag)
    check(loss, d, im)
    # the product's torch expressions
    d = torch.from_numpy(g["depth"])[None].clone().requires_grad_(True)
    im = torch.from_numpy(g["image"])[None].clone().requires_grad_(True)
    loss, pd, gd = train_step.lidar_loss({"depth_lidar": d, "image_lidar": im}, gt, ad, ar, ai)
    if patch:
        loss = loss + train_step.patch_gradient_loss(pd, gd, gt[..., 0], patch[0], patch[1], scale, ag)
    check(loss, d, im)
    assert np.abs(want_gd).max() > 1.0 and (np.abs(want_gd) > 0).mean() > 0.5   # the fixture exercises the terms
