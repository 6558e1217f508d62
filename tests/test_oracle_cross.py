"""Cross-checks between the two independent restatements (C: lnh_oracle.c, NumPy: grid_ref/encoders_ref) + KATs."""
import numpy as np
import pytest

from oracle import c_oracle, grid_ref

H, L, CH = 16, 16, 2
PLS = grid_ref.per_level_scale(32768, H, L)
S = float(np.log2(PLS))
OFF = grid_ref.make_offsets(3, L, PLS, H, 19)


def test_level_table_matches_survey():
    # rows [4920, 21952, 97336, 421880, 524288 x 12], total 6 837 544
    rows = np.diff(OFF)
    assert list(rows[:4]) == [4920, 21952, 97336, 421880]
    assert all(rows[4:] == 524288)
    assert OFF[-1] == 6837544
    assert abs(PLS - 1.662476) < 1e-6
    for l in range(L):
        sc_c, res_c = c_oracle.grid_level(l, S, H)
        sc_n, res_n = grid_ref.level_geometry(l, S, H)
        assert sc_c == sc_n and res_c == res_n
    assert c_oracle.grid_level(0, S, H) == (np.float32(15.0), 16)


def _points(B, seed):
    r = np.random.default_rng(seed)
    x = r.random((B, 3), dtype=np.float32)
    x[0] = [0, 0, 0]
    x[1] = [1, 1, 1]
    x[2] = [1.0000001, 0.5, 0.5]  # out of range -> zeros
    x[3] = [0.5, -1e-7, 0.5]
    x[4] = [0.5, 0.5, 0.5]
    return x


@pytest.mark.parametrize("gridtype,align", [(0, False), (1, False), (0, True)])
def test_grid_indices_c_vs_numpy(gridtype, align):
    x = _points(513, 1)
    off = grid_ref.make_offsets(3, L, PLS, H, 19, align_corners=align)
    idx_c = c_oracle.grid_indices(x, off, CH, S, H, gridtype, align)
    for l in range(L):
        valid, rows, _ = grid_ref.corners(x, l, off, S, H, gridtype, align)
        np.testing.assert_array_equal(idx_c[l][valid], rows[valid] * CH)
        assert np.all(idx_c[l][~valid] == 0xFFFFFFFF)
        assert rows[valid].max() < off[l + 1] - off[l]


@pytest.mark.parametrize("gridtype,align", [(0, False), (1, False), (0, True)])
def test_grad_total_variation_c_vs_numpy(gridtype, align):
    """kernel_grad_tv (gridencoder.cu:695-807): C restatement vs the NumPy one, plus two hand-checkable properties."""
    x = _points(700, 11)
    off = grid_ref.make_offsets(3, L, PLS, H, 19, align_corners=align)
    emb = np.random.default_rng(12).standard_normal((int(off[-1]), CH)).astype(np.float32)
    a = c_oracle.grad_total_variation(x, emb, off, 1e-2, S, H, gridtype, align)
    b = grid_ref.grad_total_variation(x, emb, off, 1e-2, S, H, gridtype, align)
    np.testing.assert_allclose(a, b, rtol=0, atol=2e-6 * np.abs(b).max())
    # a constant table has no variation: zero gradient (0 * rsqrt(1e-9))
    flat = np.full_like(emb, 0.37)
    assert np.abs(c_oracle.grad_total_variation(x, flat, off, 1e-2, S, H, gridtype, align)).max() == 0
    # each visited (point, level, channel) adds at most weight / (2 D) * sqrt(2 D) in magnitude (Cauchy-Schwarz)
    one = c_oracle.grad_total_variation(x[4:5], emb, off, 1e-2, S, H, gridtype, align)
    assert 0 < np.abs(one).max() <= 1e-2 / 6 * np.sqrt(6) * (1 + 1e-5)
    assert (one != 0).sum() <= L * CH


def test_sph_from_ray_lands_on_the_sphere():
    """kernel_sph_from_ray (raymarching.cu:182-217): the coordinates invert to a point at |p| = radius on the ray."""
    r = np.random.default_rng(5)
    o = r.uniform(-0.5, 0.5, (256, 3)).astype(np.float32)
    d = r.standard_normal((256, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    radius = 2.0
    c = c_oracle.sph_from_ray(o, d, radius).astype(np.float64)
    theta, phi = (c[:, 0] + 1) * np.pi / 2, c[:, 1] * np.pi
    p = radius * np.stack([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)], 1)  # y up
    t = ((p - o) * d).sum(1)
    assert (t > 0).all()
    np.testing.assert_allclose(o + t[:, None] * d, p, atol=2e-5)


def test_hash_known_answers():
    # fast_hash (gridencoder.cu:53-67) on hand-computed values, level 15 (hashed, 2^19 rows)
    pg = np.array([[1, 1, 1], [5, 0, 0], [0, 3, 0], [123456, 654321, 999]], dtype=np.uint32)
    want = [(1 ^ 2654435761 ^ 805459861) % 2 ** 19, 5, (3 * 2654435761 % 2 ** 32) % 2 ** 19,
            ((123456) ^ (654321 * 2654435761 % 2 ** 32) ^ (999 * 805459861 % 2 ** 32)) % 2 ** 19]
    got = grid_ref.grid_index(pg, 2 ** 19, 32768, 0, False)
    assert list(got) == want


@pytest.mark.parametrize("dt", [np.float32, np.float16])
def test_grid_forward_backward_c_vs_numpy(dt):
    x = _points(300, 2)
    r = np.random.default_rng(0)
    emb = (r.random((int(OFF[-1]), CH), dtype=np.float32) * 2 - 1).astype(dt)
    out_c, _ = c_oracle.grid_forward(x, emb, OFF, S, H)
    out_n = grid_ref.forward(x, emb, OFF, S, H)
    if dt == np.float32:
        np.testing.assert_allclose(out_c, out_n, rtol=0, atol=1e-6)
    else:
        np.testing.assert_allclose(out_c.astype(np.float32), out_n.astype(np.float32), rtol=0, atol=2e-3)
    assert np.all(out_c[:, 2] == 0) and np.all(out_c[:, 3] == 0)
    g = (r.standard_normal((L, 300, CH))).astype(dt)
    ge_c = c_oracle.grid_backward(g, x, OFF, int(OFF[-1]), S, H)
    ge_n = grid_ref.backward(g, x, OFF, int(OFF[-1]), S, H)
    np.testing.assert_allclose(ge_c, ge_n, rtol=1e-6, atol=1e-7)


def test_grid_gradient_is_adjoint():
    # <forward(x; E), G> differentiated w.r.t. E must equal backward(G): linearity in the table
    x = _points(64, 3)
    r = np.random.default_rng(1)
    rows = int(OFF[-1])
    E = r.standard_normal((rows, CH)).astype(np.float32)
    G = r.standard_normal((L, 64, CH)).astype(np.float32)
    out, _ = c_oracle.grid_forward(x, E, OFF, S, H)
    ge = c_oracle.grid_backward(G, x, OFF, rows, S, H)
    np.testing.assert_allclose((out.astype(np.float64) * G).sum(), (ge * E).sum(), rtol=1e-5)


def test_grid_dy_dx_finite_difference():
    x = (np.random.default_rng(5).random((32, 3)) * 0.8 + 0.1).astype(np.float32)
    E = np.random.default_rng(6).standard_normal((int(OFF[-1]), CH)).astype(np.float32)
    offs = OFF[:5]  # coarse levels only: cells are wide enough for a stable central difference
    _, dy = c_oracle.grid_forward(x, E, offs, S, H, calc_dy_dx=True)
    eps = 1e-4
    for d in range(3):
        xp, xm = x.copy(), x.copy()
        xp[:, d] += eps
        xm[:, d] -= eps
        fp, _ = c_oracle.grid_forward(xp, E, offs, S, H)
        fm, _ = c_oracle.grid_forward(xm, E, offs, S, H)
        fd = (fp.astype(np.float64) - fm) / (xp[:, d] - xm[:, d]).astype(np.float64)[None, :, None]
        an = dy[:, :, d, :].transpose(1, 0, 2)
        close = np.isclose(an, fd, rtol=5e-2, atol=5e-2)
        assert close.mean() > 0.97  # cell-boundary crossings excepted


def test_morton_packbits_roundtrip():
    r = np.random.default_rng(0)
    c = r.integers(0, 128, size=(1000, 3), dtype=np.int32)
    c[0] = [0, 0, 0]
    c[1] = [127, 127, 127]
    c[2] = [1, 2, 4]
    m = c_oracle.morton3D(c)
    assert m[0] == 0 and m[1] == 128 ** 3 - 1
    assert m[2] == (1 | (0b1000 << 1) | (0b1000000 << 2))  # x bit0 -> bit0, y bit1 -> bit4, z bit2 -> bit8
    np.testing.assert_array_equal(c_oracle.morton3D_invert(m), c)
    assert len(np.unique(m)) == len(np.unique(c, axis=0))
    g = r.random(128 ** 3, dtype=np.float32)
    bits = c_oracle.packbits(g, 0.5)
    np.testing.assert_array_equal(np.unpackbits(bits, bitorder="little").astype(bool), g > 0.5)


def test_march_rays_train_properties():
    r = np.random.default_rng(3)
    Hh, casc, bound = 128, 1, 1.0
    dens = np.zeros(Hh ** 3, np.float32)
    # occupied ball of radius 0.5 in morton order
    idx = np.arange(Hh ** 3, dtype=np.int32)
    xyz = (c_oracle.morton3D_invert(idx).astype(np.float32) + 0.5) / Hh * 2 - 1
    dens[np.linalg.norm(xyz, axis=1) < 0.5] = 1.0
    bits = c_oracle.packbits(dens, 0.01)
    N = 64
    o = np.tile(np.array([[-0.9, 0.05, 0.02]], np.float32), (N, 1))
    d = r.standard_normal((N, 3)).astype(np.float32)
    d[:, 0] = np.abs(d[:, 0]) + 1.0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    nears, fars = c_oracle.near_far_from_aabb(o, d, [-1, -1, -1, 1, 1, 1], 0.05)
    M = N * 1024
    xyzs, dirs, deltas, rays, counter = c_oracle.march_rays_train(o, d, bits, bound, 0.0, 1024, casc, Hh, M, nears,
                                                                  fars, np.zeros(N, np.float32))
    assert counter[1] == N and counter[0] == rays[:, 2].sum() and counter[0] > 0
    # every emitted sample lies in an occupied cell and inside the ball (+ one cell of slack)
    P = counter[0]
    ci, occ = c_oracle.occupancy_lookup(xyzs[:P], deltas[:P, 0], bits, bound, casc, Hh)
    assert occ.all()
    assert np.linalg.norm(xyzs[:P], axis=1).max() < 0.5 + 2 * np.sqrt(3) / Hh
    # offsets are an exclusive scan of counts in ray order (deterministic oracle allocation)
    np.testing.assert_array_equal(rays[:, 1], np.concatenate([[0], np.cumsum(rays[:-1, 2])]))
    # composite with constant sigma/rgb: weights_sum = 1 - exp(-sigma * sum(dt)) per ray
    sig = np.full(P, 3.0, np.float32)
    rgb = np.full((P, 3), 0.5, np.float32)
    ws, dep, img = c_oracle.composite_rays_train_forward(sig, rgb, deltas[:P], rays, T_thresh=0.0)
    for n in range(N):
        s, k = rays[n, 1], rays[n, 2]
        want = 1 - np.exp(-3.0 * deltas[s:s + k, 0].astype(np.float64).sum())
        assert abs(ws[n] - want) < 1e-4
        assert abs(img[n, 0] - 0.5 * want) < 1e-4


def test_sh_structure_form_matches_explicit_table_and_scipy():
    """oracle/encoders_ref.sh_forward_any (degree 1..8, from the structure of the harmonics) vs the explicit degree <= 4
    restatement of shencoder.cu:53-89 on raw directions, and vs scipy's complex harmonics on the unit sphere."""
    import numpy as np
    from scipy.special import sph_harm_y
    from oracle import encoders_ref as e
    d = np.random.default_rng(0).normal(size=(300, 3))
    for deg in (1, 2, 3, 4):
        np.testing.assert_allclose(e.sh_forward_any(d, deg), e.sh_forward(d.astype(np.float32), deg), rtol=1e-5, atol=3e-6)
    u = d / np.linalg.norm(d, axis=1, keepdims=True)
    theta, phi = np.arccos(u[:, 2]), np.arctan2(u[:, 1], u[:, 0])
    Y = e.sh_forward_any(u, 8)
    for l in range(8):
        for m in range(-l, l + 1):
            c = sph_harm_y(l, abs(m), theta, phi)
            ref = c.real if m == 0 else np.sqrt(2) * (c.real if m > 0 else c.imag)
            np.testing.assert_allclose(Y[:, l * l + l + m], ref, rtol=0, atol=1e-12)


def test_bf16_rounding_of_the_mlp_oracle_matches_torch():
    """oracle/mlp_ref.round_bf16 (integer arithmetic on the float32 bit pattern) == torch's float32 -> bfloat16 conversion
    (round to nearest even) on random values, ties, subnormals, signed zeros and values next to the largest finite one."""
    import torch
    from oracle import mlp_ref
    r = np.random.default_rng(0)
    x = np.concatenate([
        r.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** r.integers(-30, 30, 20000).astype(np.float32),
        np.array([0.0, -0.0, 1.0, -1.0, 1.00390625, 1.01171875, 3.3895314e38, -3.3895314e38, 1e-40, -1e-40, 65504.0],
                 np.float32),
        # exact ties: upper 16 bits + 0x8000 with even / odd last kept bit
        (np.arange(0x3F80, 0x3F90, dtype=np.uint32) << 16 | 0x8000).view(np.float32),
    ])
    want = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    got = mlp_ref.round_bf16(x)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
