"""Helpers for the -m gpu parity tests: call the C ABI through lidarnerf._hip with torch-owned device memory."""
import numpy as np
import torch

from lidarnerf import _hip

DEV = "cuda"


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()


def call(name, *args):
    _hip.call(name, *[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args])


def wgrad():
    """(pointer, bytes): the weight-gradient workspace the MLP backward entry points take as their last two arguments."""
    return _hip.wgrad_ws(DEV)
