"""NumPy restatements of the frequency / spherical-harmonics encoders and trunc_exp.

TEST INFRASTRUCTURE ONLY.  Follows /root/reference:
  lidarnerf/freqencoder/src/freqencoder.cu:34-63   kernel_freq           (layout [x | sin(2^f x) | cos(2^f x)]_f)
  lidarnerf/freqencoder/src/freqencoder.cu:68-101  kernel_freq_backward
  lidarnerf/encoding.py:6-47                       pure-torch FreqEncoder (same layout; pinned by golden G4)
  lidarnerf/shencoder/src/shencoder.cu:53-89       kernel_sh, degree <= 4 (16 real-SH polynomials)
  lidarnerf/activation.py:6-20                     trunc_exp
"""
import numpy as np


# ----------------------------------------------------------------------------- frequency
def freq_forward(x, degree):
    """x [B, D] float32 -> [B, D + 2*D*degree] float32.
    Column c >= D: col=c//D-1, d=c%D, f=col//2, phase=(col%2)*pi/2, out=sin(x[d]*2^f + phase)  (freqencoder.cu:52-62).
    The CUDA build uses __sinf (fast-math); this oracle evaluates sin in float64 and rounds once, the value any
    accurate implementation approximates."""
    x = np.asarray(x, dtype=np.float32)
    B, D = x.shape
    C = D + 2 * D * degree
    out = np.empty((B, C), dtype=np.float32)
    out[:, :D] = x
    for c in range(D, C):
        col, d = c // D - 1, c % D
        f, odd = col // 2, col % 2
        arg = np.ldexp(x[:, d], f).astype(np.float32)  # scalbnf(x, f): exact
        if odd:
            # the kernel adds a float32 pi/2 to the float32 argument before sin
            arg = (arg + np.float32(np.float32(3.141592653589793) / np.float32(2))).astype(np.float32)
        out[:, c] = np.sin(arg.astype(np.float64)).astype(np.float32)
    return out


def freq_forward_exactcos(x, degree):
    """Same layout but cos evaluated as cos() (the pure-torch twin, encoding.py:35-47)."""
    x = np.asarray(x, dtype=np.float32)
    outs = [x]
    for f in range(degree):
        a = np.ldexp(x, f).astype(np.float32)
        outs.append(np.sin(a.astype(np.float64)).astype(np.float32))
        outs.append(np.cos(a.astype(np.float64)).astype(np.float32))
    return np.concatenate(outs, axis=1)


def freq_backward(grad, outputs, D, degree):
    """freqencoder.cu:84-100: g_x[d] = g[d] + sum_f 2^f (g_sin * out_cos - g_cos * out_sin) from SAVED outputs."""
    g = np.asarray(grad, dtype=np.float32)
    o = np.asarray(outputs, dtype=np.float32)
    B = g.shape[0]
    res = g[:, :D].astype(np.float32).copy()
    for f in range(degree):
        base = D + 2 * D * f
        gs, gc = g[:, base:base + D], g[:, base + D:base + 2 * D]
        os_, oc = o[:, base:base + D], o[:, base + D:base + 2 * D]
        res = (res + np.float32(2.0 ** f) * (gs * oc - gc * os_)).astype(np.float32)
    return res


# ----------------------------------------------------------------------------- spherical harmonics
def sh_forward(d, degree):
    """shencoder.cu:53-89: polynomials in the RAW direction (x,y,z) (not re-normalised), degree in 1..4."""
    assert 1 <= degree <= 4
    d = np.asarray(d, dtype=np.float32)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    f = np.float32
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    out = np.empty((d.shape[0], degree * degree), dtype=np.float32)
    out[:, 0] = f(0.28209479177387814)
    if degree > 1:
        out[:, 1] = f(-0.48860251190291987) * y
        out[:, 2] = f(0.48860251190291987) * z
        out[:, 3] = f(-0.48860251190291987) * x
    if degree > 2:
        out[:, 4] = f(1.0925484305920792) * xy
        out[:, 5] = f(-1.0925484305920792) * yz
        out[:, 6] = f(0.94617469575755997) * z2 - f(0.31539156525251999)
        out[:, 7] = f(-1.0925484305920792) * xz
        out[:, 8] = f(0.54627421529603959) * x2 - f(0.54627421529603959) * y2
    if degree > 3:
        out[:, 9] = f(0.59004358992664352) * y * (f(-3.0) * x2 + y2)
        out[:, 10] = f(2.8906114426405538) * xy * z
        out[:, 11] = f(0.45704579946446572) * y * (f(1.0) - f(5.0) * z2)
        out[:, 12] = f(0.3731763325901154) * z * (f(5.0) * z2 - f(3.0))
        out[:, 13] = f(0.45704579946446572) * x * (f(1.0) - f(5.0) * z2)
        out[:, 14] = f(1.4453057213202769) * z * (x2 - y2)
        out[:, 15] = f(0.59004358992664352) * x * (-x2 + f(3.0) * y2)
    return out


def sh_jacobian_fd(d, degree, eps=1e-3):
    """Central finite differences of sh_forward in float64 (check for the analytic dy/dx of the HIP kernel)."""
    d = np.asarray(d, dtype=np.float64)
    J = np.empty((d.shape[0], 3, degree * degree))
    for k in range(3):
        dp, dm = d.copy(), d.copy()
        dp[:, k] += eps
        dm[:, k] -= eps
        J[:, k] = (_sh64(dp, degree) - _sh64(dm, degree)) / (2 * eps)
    return J


def _sh64(d, degree):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    o = [np.full_like(x, 0.28209479177387814)]
    if degree > 1:
        o += [-0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x]
    if degree > 2:
        o += [1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
              -1.0925484305920792 * xz, 0.54627421529603959 * (x2 - y2)]
    if degree > 3:
        o += [0.59004358992664352 * y * (-3 * x2 + y2), 2.8906114426405538 * xy * z,
              0.45704579946446572 * y * (1 - 5 * z2), 0.3731763325901154 * z * (5 * z2 - 3),
              0.45704579946446572 * x * (1 - 5 * z2), 1.4453057213202769 * z * (x2 - y2),
              0.59004358992664352 * x * (-x2 + 3 * y2)]
    return np.stack(o, axis=1)


# ----------------------------------------------------------------------------- trunc_exp
def trunc_exp_forward(x):
    """activation.py:11-13: exp(x) in float32."""
    return np.exp(np.asarray(x, dtype=np.float32).astype(np.float64)).astype(np.float32)


def trunc_exp_backward(g, x):
    """activation.py:17-19: g * exp(clamp(x, -15, 15))."""
    x = np.clip(np.asarray(x, dtype=np.float32), -15, 15)
    return (np.asarray(g, dtype=np.float32) * np.exp(x.astype(np.float64))).astype(np.float32)


# ---------------------------------------------------------------------------------------------- SH, degree 1..8
def sh_forward_any(d, degree):
    """Real spherical harmonics of shencoder.cu:53-300 for degree 1..8 in float64, from their structure:
    Y_l^m = N_lm * d^|m|/dz^|m| P_l(z) * {Re, Im}(x + iy)^|m|  (Condon-Shortley sign, index l*l + l + m) — the same
    polynomials in the RAW (x, y, z) as the reference's unrolled expressions.  Cross-checked against the explicit
    degree <= 4 table above and against scipy's complex harmonics on the unit sphere (tests/test_oracle_cross.py)."""
    from math import factorial, pi, sqrt
    from numpy.polynomial import legendre as npl
    from numpy.polynomial import polynomial as npp
    assert 1 <= degree <= 8
    d = np.asarray(d, dtype=np.float64)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    w = (x + 1j * y)
    out = np.empty((d.shape[0], degree * degree))
    for l in range(degree):
        p = npl.leg2poly([0] * l + [1])                      # P_l as an ordinary polynomial in z
        for m in range(0, l + 1):
            q = npp.polyval(z, npp.polyder(p, m) if m else p)
            if m == 0:
                out[:, l * l + l] = sqrt((2 * l + 1) / (4 * pi)) * q
            else:
                n = (-1) ** m * sqrt(2.0) * sqrt((2 * l + 1) / (4 * pi) * factorial(l - m) / factorial(l + m))
                wm = w ** m
                out[:, l * l + l + m] = n * q * wm.real
                out[:, l * l + l - m] = n * q * wm.imag
    return out


def sh_jacobian_fd_any(d, degree, eps=1e-4):
    d = np.asarray(d, dtype=np.float64)
    J = np.empty((d.shape[0], 3, degree * degree))
    for k in range(3):
        dp, dm = d.copy(), d.copy()
        dp[:, k] += eps
        dm[:, k] -= eps
        J[:, k] = (sh_forward_any(dp, degree) - sh_forward_any(dm, degree)) / (2 * eps)
    return J
