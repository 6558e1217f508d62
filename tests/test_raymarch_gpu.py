"""Occupancy-grid indexing / marching / ragged compositing (lnh_* raymarching entry points) vs the C oracle.
Integer results are compared bit-exactly; the atomic ray allocation is compared keyed by ray id."""
import numpy as np
import pytest
import torch

from oracle import c_oracle

pytestmark = pytest.mark.gpu


def test_morton_bit_exact():
    from gpu_util import call, dev, host
    c = np.random.default_rng(0).integers(0, 1024, size=(100000, 3), dtype=np.int32)
    out = torch.empty(100000, dtype=torch.int32, device="cuda")
    call("lnh_morton3D", dev(c), 100000, out)
    want = c_oracle.morton3D(c)
    np.testing.assert_array_equal(host(out), want)
    back = torch.empty((100000, 3), dtype=torch.int32, device="cuda")
    call("lnh_morton3D_invert", out, 100000, back)
    np.testing.assert_array_equal(host(back), c)
    # negative / high-bit indices follow the arithmetic-shift semantics of the reference
    idx = np.array([-1, -2, 0x7FFFFFFF, -2147483648, 5], dtype=np.int32)
    b2 = torch.empty((5, 3), dtype=torch.int32, device="cuda")
    call("lnh_morton3D_invert", dev(idx), 5, b2)
    np.testing.assert_array_equal(host(b2), c_oracle.morton3D_invert(idx))


def test_packbits_bit_exact_full_grid():
    from gpu_util import call, dev, host
    g = np.random.default_rng(1).random(128 ** 3, dtype=np.float32)
    g[:16] = 0.5  # exactly at the threshold: strict '>'
    out = torch.empty(128 ** 3 // 8, dtype=torch.uint8, device="cuda")
    call("lnh_packbits", dev(g), 128 ** 3 // 8, 0.5, out)
    np.testing.assert_array_equal(host(out), c_oracle.packbits(g, 0.5))


def _scene(cascade):
    Hh = 128
    r = np.random.default_rng(3)
    dens = np.zeros(cascade * Hh ** 3, np.float32)
    idx = np.arange(Hh ** 3, dtype=np.int32)
    xyz = (c_oracle.morton3D_invert(idx).astype(np.float32) + 0.5) / Hh * 2 - 1
    for cas in range(cascade):
        rr = np.linalg.norm(xyz * (2 ** cas), axis=1)
        dens[cas * Hh ** 3:(cas + 1) * Hh ** 3][(rr < 0.6 * 2 ** cas) & (r.random(Hh ** 3) < 0.7)] = 1.0
    return c_oracle.packbits(dens, 0.01), Hh


@pytest.mark.parametrize("cascade,bound", [(1, 1.0), (2, 2.0)])
def test_occupancy_lookup_bit_exact(cascade, bound):
    from gpu_util import call, dev, host
    bits, Hh = _scene(cascade)
    r = np.random.default_rng(5)
    N = 200000
    xyz = ((r.random((N, 3), dtype=np.float32) * 2 - 1) * bound * 1.1).astype(np.float32)
    dt = (r.random(N, dtype=np.float32) * 0.05).astype(np.float32)
    ci = torch.empty(N, dtype=torch.int32, device="cuda")
    occ = torch.empty(N, dtype=torch.uint8, device="cuda")
    call("lnh_occupancy_lookup", dev(xyz), dev(dt), dev(bits), bound, N, cascade, Hh, ci, occ)
    wci, wocc = c_oracle.occupancy_lookup(xyz, dt, bits, bound, cascade, Hh)
    np.testing.assert_array_equal(host(ci).view(np.uint32), wci)
    np.testing.assert_array_equal(host(occ), wocc)


@pytest.mark.parametrize("cascade,bound,dt_gamma", [(1, 1.0, 0.0), (2, 2.0, 1 / 128)])
def test_march_rays_train(cascade, bound, dt_gamma):
    from gpu_util import call, dev, host
    bits, Hh = _scene(cascade)
    r = np.random.default_rng(7)
    N = 1000
    o = (r.standard_normal((N, 3)) * 0.05 + np.array([-0.8 * bound, 0.1, 0.0])).astype(np.float32)
    d = r.standard_normal((N, 3)).astype(np.float32)
    d[:, 0] = np.abs(d[:, 0]) + 0.7
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears = torch.empty(N, device="cuda")
    fars = torch.empty(N, device="cuda")
    call("lnh_near_far_from_aabb", dev(o), dev(d), dev(aabb), N, 0.05, nears, fars)
    wn, wf = c_oracle.near_far_from_aabb(o, d, aabb, 0.05)
    np.testing.assert_array_equal(host(nears), wn)
    np.testing.assert_array_equal(host(fars), wf)
    noises = r.random(N, dtype=np.float32)
    M = N * 256
    xyzs = torch.zeros((M, 3), device="cuda")
    dirs = torch.zeros((M, 3), device="cuda")
    deltas = torch.zeros((M, 2), device="cuda")
    rays = torch.zeros((N, 3), dtype=torch.int32, device="cuda")
    counter = torch.zeros(2, dtype=torch.int32, device="cuda")
    call("lnh_march_rays_train", dev(o), dev(d), dev(bits), bound, dt_gamma, 1024, N, cascade, Hh, M, nears, fars, xyzs,
         dirs, deltas, rays, counter, dev(noises))
    wx, wd, wdl, wr, wc = c_oracle.march_rays_train(o, d, bits, bound, dt_gamma, 1024, cascade, Hh, M, wn, wf, noises)
    g_rays, g_cnt = host(rays), host(counter)
    np.testing.assert_array_equal(g_cnt, wc)
    assert wc[0] > 1000 and wc[0] <= M
    # ray table keyed by id: counts identical; offsets are a permutation-consistent allocation
    order = np.argsort(g_rays[:, 0])
    g_sorted = g_rays[order]
    np.testing.assert_array_equal(g_sorted[:, 0], np.arange(N))
    np.testing.assert_array_equal(g_sorted[:, 2], wr[:, 2])
    spans = sorted((int(a), int(a + b)) for _, a, b in g_rays if b > 0)
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))  # disjoint
    gx, gd, gdl = host(xyzs), host(dirs), host(deltas)
    for n in range(0, N, 7):
        a, k = g_sorted[n, 1], g_sorted[n, 2]
        b = wr[n, 1]
        np.testing.assert_array_equal(gx[a:a + k], wx[b:b + k])
        np.testing.assert_array_equal(gd[a:a + k], wd[b:b + k])
        np.testing.assert_array_equal(gdl[a:a + k], wdl[b:b + k])
    # ragged compositing on top of the marched samples
    P = int(wc[0])
    sig = (r.random(M, dtype=np.float32) * 30).astype(np.float32)
    rgb = r.random((M, 3), dtype=np.float32)
    ws = torch.zeros(N, device="cuda")
    dep = torch.zeros(N, device="cuda")
    img = torch.zeros((N, 3), device="cuda")
    call("lnh_composite_rays_train_forward", dev(sig), dev(rgb), deltas, rays, M, N, 1e-4, ws, dep, img)
    w_ws, w_dep, w_img = c_oracle.composite_rays_train_forward(sig, rgb, gdl, g_rays, 1e-4)
    np.testing.assert_allclose(host(ws), w_ws, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(host(dep), w_dep, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(host(img), w_img, rtol=1e-5, atol=1e-6)
    gws = r.standard_normal(N).astype(np.float32)
    gim = r.standard_normal((N, 3)).astype(np.float32)
    gs = torch.zeros(M, device="cuda")
    gc = torch.zeros((M, 3), device="cuda")
    call("lnh_composite_rays_train_backward", dev(gws), dev(gim), dev(sig), dev(rgb), deltas, rays, ws, img, M, N, 1e-4,
         gs, gc)
    w_gs, w_gc = c_oracle.composite_rays_train_backward(gws, gim, sig, rgb, gdl, g_rays, host(ws), host(img), 1e-4)
    np.testing.assert_allclose(host(gs), w_gs, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(host(gc), w_gc, rtol=1e-5, atol=1e-6)
    assert P > 0


@pytest.mark.parametrize("max_steps,dt_gamma", [(40, 0.0), (96, 1 / 64), (1024, 0.0)])
def test_march_rays_train_caps(max_steps, dt_gamma):
    """Coarse and fine lattices (max_steps sets dt_min: 1 .. 5 chunks of 64 lattice points per ray), a sample buffer that
    holds about half of what the rays ask for (the rays that arrive late keep their table entry and write nothing), rays
    that miss the box (near = far = FLT_MAX) and rays that start beyond `far` — bit-exact against the C oracle."""
    from gpu_util import call, dev, host
    cascade, bound = 1, 1.0
    bits, Hh = _scene(cascade)
    r = np.random.default_rng(17)
    N = 777                                                    # not a multiple of the 4 rays a workgroup takes
    o = (r.standard_normal((N, 3)) * 0.05 + np.array([-0.8 * bound, 0.1, 0.0])).astype(np.float32)
    d = r.standard_normal((N, 3)).astype(np.float32)
    d[:, 0] = np.abs(d[:, 0]) + 0.7
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o[::13] += np.array([0.0, 5.0, 0.0], np.float32)           # these miss the box
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    wn, wf = c_oracle.near_far_from_aabb(o, d, aabb, 0.05)
    wf[5::13] = wn[5::13] * 0.5                                # far in front of near: no lattice point is visited
    noises = r.random(N, dtype=np.float32)
    big = N * max_steps
    _, _, _, _, want_all = c_oracle.march_rays_train(o, d, bits, bound, dt_gamma, max_steps, cascade, Hh, big, wn, wf, noises)
    total = int(want_all[0])
    assert total > 0
    # a ray's samples are written iff offset + count <= M, and offsets depend on the arrival order: compare per ray, through
    # the oracle run with room for everything
    wx, wd, wdl, wr, wc = c_oracle.march_rays_train(o, d, bits, bound, dt_gamma, max_steps, cascade, Hh, big, wn, wf, noises)
    assert int(wr[:, 2].max()) <= max_steps   # (dt >= 2 sqrt(3) / max_steps: a ray cannot hold more lattice points)
    M = max(total // 2, 1)
    xyzs = torch.zeros((M, 3), device="cuda")
    dirs = torch.zeros((M, 3), device="cuda")
    deltas = torch.zeros((M, 2), device="cuda")
    rays = torch.zeros((N, 3), dtype=torch.int32, device="cuda")
    counter = torch.zeros(2, dtype=torch.int32, device="cuda")
    call("lnh_march_rays_train", dev(o), dev(d), dev(bits), bound, dt_gamma, max_steps, N, cascade, Hh, M, dev(wn), dev(wf),
         xyzs, dirs, deltas, rays, counter, dev(noises))
    g_rays, g_cnt = host(rays), host(counter)
    np.testing.assert_array_equal(g_cnt, wc)                   # every ray counted, dropped or not
    g_sorted = g_rays[np.argsort(g_rays[:, 0])]
    np.testing.assert_array_equal(g_sorted[:, 0], np.arange(N))
    np.testing.assert_array_equal(g_sorted[:, 2], wr[:, 2])
    assert int((g_sorted[::13, 2] != 0).sum()) == 0 and int((g_sorted[5::13, 2] != 0).sum()) == 0
    gx, gd, gdl = host(xyzs), host(dirs), host(deltas)
    kept = dropped = 0
    for n in range(N):
        a, k = int(g_sorted[n, 1]), int(g_sorted[n, 2])
        if k == 0:
            continue
        if a + k <= M:
            b = int(wr[n, 1])
            np.testing.assert_array_equal(gx[a:a + k], wx[b:b + k])
            np.testing.assert_array_equal(gd[a:a + k], wd[b:b + k])
            np.testing.assert_array_equal(gdl[a:a + k], wdl[b:b + k])
            kept += 1
        else:
            assert float(np.abs(gdl[a:M]).max(initial=0.0)) == 0.0   # nothing written into a dropped ray's slots
            dropped += 1
    assert kept > 0 and dropped > 0


@pytest.mark.parametrize("cascade,bound,dt_gamma", [(1, 1.0, 0.0), (2, 2.0, 1 / 128)])
def test_march_and_composite_rays_inference(cascade, bound, dt_gamma):
    """Inference loop of torch-ngp's run_cuda (raymarching.cu:808-928, 966-1053; raymarching.py:362-512): rounds of
    march_rays -> composite_rays over a shrinking set of alive rays, every round checked against the C restatement —
    sample positions / deltas and the alive table bit-exact, the accumulated sums to float rounding (expf)."""
    from lidarnerf import raymarching
    bits, Hh = _scene(cascade)
    r = np.random.default_rng(11)
    N, n_step = 700, 8
    o = (r.standard_normal((N, 3)) * 0.05 + np.array([-0.8 * bound, 0.1, 0.0])).astype(np.float32)
    d = r.standard_normal((N, 3)).astype(np.float32)
    d[:, 0] = np.abs(d[:, 0]) + 0.7
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    wn, wf = c_oracle.near_far_from_aabb(o, d, aabb, 0.05)
    t_o, t_d, t_bits = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), torch.from_numpy(bits).cuda()
    near, far = torch.from_numpy(wn).cuda(), torch.from_numpy(wf).cuda()
    alive = torch.arange(N, dtype=torch.int32, device="cuda")
    rays_t = near.clone()
    ws = torch.zeros(N, device="cuda")
    dep = torch.zeros(N, device="cuda")
    img = torch.zeros((N, 3), device="cuda")
    # host mirrors
    h_alive, h_t = np.arange(N, dtype=np.int32), wn.copy()
    h_ws, h_dep, h_img = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    n_alive, rounds = N, 0
    while n_alive > 0 and rounds < 40:
        xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, alive, rays_t, t_o, t_d, bound, t_bits, cascade, Hh,
                                                    near, far, -1, False, dt_gamma, 1024)
        wx, wd, wdl = c_oracle.march_rays(n_alive, n_step, h_alive, h_t, o, d, bound, dt_gamma, 1024, cascade, Hh, bits,
                                          wn, wf, np.zeros(n_alive, np.float32))
        np.testing.assert_array_equal(xyzs.cpu().numpy(), wx)
        np.testing.assert_array_equal(dirs.cpu().numpy(), wd)
        np.testing.assert_array_equal(deltas.cpu().numpy(), wdl)
        sig = (r.random(n_alive * n_step, dtype=np.float32) * 4).astype(np.float32)
        rgb = r.random((n_alive * n_step, 3), dtype=np.float32)
        raymarching.composite_rays(n_alive, n_step, alive, rays_t, torch.from_numpy(sig).cuda(),
                                   torch.from_numpy(rgb).cuda(), deltas, ws, dep, img, 1e-2)
        h_alive2, h_t, h_ws, h_dep, h_img = c_oracle.composite_rays(n_alive, n_step, h_alive, h_t, sig, rgb, wdl, h_ws,
                                                                    h_dep, h_img, 1e-2)
        g_alive = alive.cpu().numpy()
        np.testing.assert_array_equal(g_alive[:n_alive], h_alive2[:n_alive])
        np.testing.assert_allclose(ws.cpu().numpy(), h_ws, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(dep.cpu().numpy(), h_dep, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(img.cpu().numpy(), h_img, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(rays_t.cpu().numpy(), h_t, rtol=0, atol=0)
        # compaction of the survivors (the caller's job: renderer of torch-ngp does rays_alive[rays_alive >= 0])
        keep = g_alive[:n_alive] >= 0
        h_alive = g_alive[:n_alive][keep].astype(np.int32)
        n_alive = int(keep.sum())
        alive = torch.from_numpy(np.concatenate([h_alive, np.zeros(N - n_alive, np.int32)])).cuda()
        h_alive = alive.cpu().numpy()
        rounds += 1
    assert rounds >= 3 and n_alive == 0          # every ray left the grid or saturated
    assert float(ws.max()) <= 1.0 + 1e-5 and float(ws.max()) > 0.5


def test_sph_from_ray():
    """lnh_sph_from_ray vs the C restatement of kernel_sph_from_ray (raymarching.cu:182-217) — float transcendental
    functions on both sides (atan2f, sqrtf): agreement to a few ulp of the [-1, 1] coordinates; and through the module API
    (raymarching.py sph_from_ray)."""
    from gpu_util import call, dev, host
    from lidarnerf import raymarching
    r = np.random.default_rng(17)
    N = 4097
    o = r.uniform(-0.5, 0.5, (N, 3)).astype(np.float32)  # inside the smallest sphere: every ray has its far hit
    d = r.standard_normal((N, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[0] = [0, 1, 0]     # straight up: theta = 0
    d[1] = [1, 0, 0]
    d[2] = [0, 0, -1]
    for radius in (1.0, 2.0, 32.0):
        want = c_oracle.sph_from_ray(o, d, radius)
        out = torch.empty((N, 2), device="cuda")
        call("lnh_sph_from_ray", dev(o), dev(d), radius, N, out)
        got = host(out)
        # phi jumps by 2 at the +-pi seam (z = 0-, x < 0): compare on the circle
        dphi = np.abs(got[:, 1] - want[:, 1])
        dphi = np.minimum(dphi, 2 - dphi)
        assert np.abs(got[:, 0] - want[:, 0]).max() <= 2e-6 and dphi.max() <= 2e-6
        via_module = raymarching.sph_from_ray(dev(o), dev(d), radius)
        assert torch.equal(via_module, out)


def test_march_prologue_equals_the_separate_launches():
    """lnh_lidar_march_prologue (one launch in front of the marcher) against what it replaced: a fill of the constant near,
    lnh_near_far_from_aabb + torch.minimum(near * 81, far of the box) — bit for bit, rays that miss the box, rays along an
    axis (infinite reciprocal direction components) and a NaN exit included — and the clearing of four regions of awkward
    sizes and alignments, bytes around them untouched."""
    import ctypes as C
    from gpu_util import call
    r = np.random.default_rng(11)
    N = 5003
    o = (r.random((N, 3), dtype=np.float32) * 2.4 - 1.2).astype(np.float32)   # some origins outside the box
    d = r.normal(size=(N, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:40] = np.array([1.0, 0.0, 0.0], np.float32)          # axis-parallel: two reciprocal components are inf
    d[40:80] = np.array([0.0, 0.0, -1.0], np.float32)
    o[40:48] = np.array([1.0, 0.2, 0.3], np.float32)         # ... and on the box's +x face: (aabb[3] - ox) * inf = NaN
    aabb = torch.tensor([-1, -1, -1, 1, 1, 1], dtype=torch.float32, device="cuda")
    ro, rd = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    near = 0.005
    want_n = torch.full((N,), near, dtype=torch.float32, device="cuda")
    nb, fb = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    call("lnh_near_far_from_aabb", ro, rd, aabb, N, near, nb, fb)
    want_f = torch.minimum(want_n * 81.0, fb)
    # regions: (offset in floats, length in floats) inside guard-filled buffers
    bufs = [torch.full((n + 16,), 7.0, device="cuda") for n in (100003, 2, 17, 4096)]
    offs = (3, 1, 5, 0)
    lens = (100003 - 3, 2, 9, 4096)
    views = [b[o_:o_ + n] for b, o_, n in zip(bufs, offs, lens)]
    zp = (C.c_void_p * 4)(*[v.data_ptr() for v in views])
    zb = (C.c_uint64 * 4)(*[v.numel() * 4 for v in views])
    got_n, got_f = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    call("lnh_lidar_march_prologue", ro, rd, aabb, N, near, 81.0, got_n, got_f, C.cast(zp, C.c_void_p),
         C.cast(zb, C.c_void_p), 4)
    torch.cuda.synchronize()
    assert torch.equal(got_n, want_n)
    a, b = got_f.cpu().numpy(), want_f.cpu().numpy()
    assert np.isnan(b).any() and (b == np.float32(near) * np.float32(81.0)).any() and (b < 0.4).any()   # every kind occurs
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    for buf, o_, n in zip(bufs, offs, lens):
        h = buf.cpu().numpy()
        assert (h[o_:o_ + n] == 0).all() and (h[:o_] == 7.0).all() and (h[o_ + n:] == 7.0).all()
    # no rays / no regions: both halves on their own
    call("lnh_lidar_march_prologue", ro, rd, aabb, 0, near, 81.0, got_n, got_f, C.cast(zp, C.c_void_p), C.cast(zb, C.c_void_p), 1)
    got_f.fill_(-1.0)
    call("lnh_lidar_march_prologue", ro, rd, aabb, N, near, 81.0, got_n, got_f, None, None, 0)
    np.testing.assert_array_equal(got_f.cpu().numpy().view(np.uint32), b.view(np.uint32))
