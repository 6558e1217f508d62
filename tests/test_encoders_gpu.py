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


def test_sh_degree_limit_is_loud():
    from lidarnerf import _hip
    x = torch.rand((4, 3), device="cuda")
    out = torch.empty((4, 25), device="cuda")
    with pytest.raises(RuntimeError, match="degree"):
        _hip.call("lnh_sh_encode_forward", x.data_ptr(), out.data_ptr(), 4, 3, 5, None)
