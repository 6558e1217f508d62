"""HIP frequency / SH encoders vs the oracle (and the golden FreqEncoder vector from the imported reference)."""
import os

import numpy as np
import pytest
import torch

from oracle import encoders_ref

pytestmark = pytest.mark.gpu


def _dirs(n, seed):
    d = np.random.default_rng(seed).standard_normal((n, 3)).astype(np.float32)
    return (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("deg", [1, 4, 12])
def test_freq_forward_backward(deg):
    from gpu_util import call, dev, host
    x = _dirs(1000, 0)
    C = 3 + 6 * deg
    out = torch.empty((1000, C), device="cuda")
    call("lnh_freq_encode_forward", dev(x), 1000, 3, deg, C, out)
    want = encoders_ref.freq_forward(x, deg)
    # arguments reach 2^11: float32 sin of a float32 argument, a few ulp of the ARGUMENT spacing
    np.testing.assert_allclose(host(out), want, rtol=0, atol=2e-6 * 2 ** deg + 1e-6)
    g = np.random.default_rng(1).standard_normal((1000, C)).astype(np.float32)
    gi = torch.empty((1000, 3), device="cuda")
    call("lnh_freq_encode_backward", dev(g), out, 1000, 3, deg, C, gi)
    np.testing.assert_allclose(host(gi), encoders_ref.freq_backward(g, host(out), 3, deg), rtol=1e-4, atol=1e-2)


def test_freq_matches_reference_golden(golden_dir):
    from gpu_util import call, dev, host
    g = np.load(os.path.join(golden_dir, "g4_freq_encoder.npz"))
    n = g["d"].shape[0]
    out = torch.empty((n, 75), device="cuda")
    call("lnh_freq_encode_forward", dev(g["d"]), n, 3, 12, 75, out)
    # reference = torch.sin/cos on CPU; kernel evaluates cos as sin(x + fl(pi/2)) like freqencoder.cu:61
    np.testing.assert_allclose(host(out), g["y"], rtol=0, atol=3e-4)
    gi = torch.empty((n, 3), device="cuda")
    call("lnh_freq_encode_backward", dev(g["g"]), out, n, 3, 12, 75, gi)
    np.testing.assert_allclose(host(gi), g["gd"], rtol=2e-3, atol=0.5)


@pytest.mark.parametrize("degree", [1, 2, 3, 4])
def test_sh_forward_and_jacobian(degree):
    from gpu_util import call, dev, host
    d = _dirs(777, 2) * np.linspace(0.5, 1.5, 777, dtype=np.float32)[:, None]  # raw (un-normalised) directions
    C2 = degree * degree
    out = torch.empty((777, C2), device="cuda")
    dy = torch.empty((777, 3, C2), device="cuda")
    call("lnh_sh_encode_forward", dev(d), out, 777, 3, degree, dy)
    np.testing.assert_allclose(host(out), encoders_ref.sh_forward(d, degree), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(host(dy), encoders_ref.sh_jacobian_fd(d, degree), rtol=1e-4, atol=1e-4)
    g = np.random.default_rng(3).standard_normal((777, C2)).astype(np.float32)
    gi = torch.zeros((777, 3), device="cuda")
    call("lnh_sh_encode_backward", dev(g), dev(d), 777, 3, degree, dy, gi)
    np.testing.assert_allclose(host(gi), np.einsum("bc,bdc->bd", g, host(dy)), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("degree", [5, 6, 7, 8])
def test_sh_high_degree_forward_and_jacobian(degree):
    """Degrees 5..8 (shencoder.cu:90-300) from the structure-based restatement (validated against the explicit
    degree-4 table and scipy's harmonics on CPU), on raw un-normalised directions."""
    from gpu_util import call, dev, host
    d = _dirs(501, 5) * np.linspace(0.6, 1.3, 501, dtype=np.float32)[:, None]
    C2 = degree * degree
    out = torch.empty((501, C2), device="cuda")
    dy = torch.empty((501, 3, C2), device="cuda")
    call("lnh_sh_encode_forward", dev(d), out, 501, 3, degree, dy)
    want = encoders_ref.sh_forward_any(d, degree)
    scale = np.abs(want).max()
    np.testing.assert_allclose(host(out), want, rtol=2e-5, atol=2e-6 * scale)
    np.testing.assert_allclose(host(out)[:, :16], encoders_ref.sh_forward(d, 4), rtol=2e-5, atol=2e-6)  # same first 16
    J = encoders_ref.sh_jacobian_fd_any(d, degree)
    np.testing.assert_allclose(host(dy), J, rtol=2e-4, atol=2e-5 * np.abs(J).max())
    out2 = torch.empty((501, C2), device="cuda")
    call("lnh_sh_encode_forward", dev(d), out2, 501, 3, degree, None)  # without the Jacobian
    assert torch.equal(out, out2)
    g = np.random.default_rng(3).standard_normal((501, C2)).astype(np.float32)
    gi = torch.zeros((501, 3), device="cuda")
    call("lnh_sh_encode_backward", dev(g), dev(d), 501, 3, degree, dy, gi)
    np.testing.assert_allclose(host(gi), np.einsum("bc,bdc->bd", g, host(dy)), rtol=1e-4, atol=1e-3)


def test_sh_module_degree_8_autograd():
    from lidarnerf.shencoder import SHEncoder
    enc = SHEncoder(degree=8)
    assert enc.output_dim == 64
    d = torch.nn.functional.normalize(torch.randn(33, 3, device="cuda"), dim=-1).requires_grad_(True)
    y = enc(d)
    y.square().sum().backward()
    assert y.shape == (33, 64) and torch.isfinite(d.grad).all()
    # on the unit sphere sum_m Y_lm^2 = (2l+1)/4pi for every l (addition theorem): a transpose/sign-proof identity
    for l in range(8):
        s = y[:, l * l:(l + 1) * (l + 1)].square().sum(1)
        torch.testing.assert_close(s, torch.full_like(s, (2 * l + 1) / (4 * np.pi)), rtol=2e-5, atol=1e-6)


def test_sh_degree_limit_is_loud():
    from lidarnerf import _hip
    x = torch.rand((4, 3), device="cuda")
    out = torch.empty((4, 81), device="cuda")
    with pytest.raises(RuntimeError, match="degree"):
        _hip.call("lnh_sh_encode_forward", x.data_ptr(), out.data_ptr(), 4, 3, 9, None)
