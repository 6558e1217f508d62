"""ctypes front-end of oracle/lnh_oracle.c (CPU restatement of the reference CUDA).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never imported by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblnh_oracle.so")


def build(force=False):
    """Compile the C oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "lnh_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.lnh_oracle_f16_to_f32.restype = C.c_float
        _lib.lnh_oracle_f32_to_f16.restype = C.c_uint16
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def grid_level(level, S, H):
    sc, res = C.c_float(), C.c_uint32()
    lib().lnh_oracle_grid_level(C.c_uint32(level), C.c_float(S), C.c_uint32(H), C.byref(sc), C.byref(res))
    return np.float32(sc.value), int(res.value)


def grid_indices(inputs, offsets, C_, S, H, gridtype=0, align_corners=False):
    inputs, offsets = _f32(inputs), _i32(offsets)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    out = np.empty((L, B, 1 << D), dtype=np.uint32)
    lib().lnh_oracle_grid_indices(_p(inputs), _p(offsets), _p(out), C.c_uint32(B), C.c_uint32(D), C.c_uint32(C_),
                                  C.c_uint32(L), C.c_float(S), C.c_uint32(H), C.c_uint32(gridtype),
                                  C.c_int(int(align_corners)))
    return out


def grid_forward(inputs, embeddings, offsets, S, H, gridtype=0, align_corners=False, interp=0, calc_dy_dx=False):
    """embeddings float32 or float16 [rows, C].  Returns outputs [L,B,C] (same dtype) and dy_dx [B,L,D,C] or None."""
    inputs, offsets = _f32(inputs), _i32(offsets)
    emb = np.ascontiguousarray(embeddings)
    assert emb.dtype in (np.float32, np.float16)
    dtype = 0 if emb.dtype == np.float32 else 1
    B, D = inputs.shape
    L, Cc = offsets.shape[0] - 1, emb.shape[1]
    out = np.empty((L, B, Cc), dtype=emb.dtype)
    dy = np.empty((B, L, D, Cc), dtype=emb.dtype) if calc_dy_dx else None
    lib().lnh_oracle_grid_forward(_p(inputs), _p(emb), _p(offsets), _p(out), C.c_uint32(B), C.c_uint32(D),
                                  C.c_uint32(Cc), C.c_uint32(L), C.c_float(S), C.c_uint32(H), _p(dy),
                                  C.c_uint32(gridtype), C.c_int(int(align_corners)), C.c_uint32(interp), C.c_int(dtype))
    return out, dy


def grid_backward(grad, inputs, offsets, n_rows, S, H, gridtype=0, align_corners=False, interp=0,
                  quantize_contrib=True):
    """grad [L,B,C] float32/float16 -> grad_embeddings float64 [rows, C] (order-free sum)."""
    inputs, offsets = _f32(inputs), _i32(offsets)
    g = np.ascontiguousarray(grad)
    dtype = 0 if g.dtype == np.float32 else 1
    B, D = inputs.shape
    L, Cc = offsets.shape[0] - 1, g.shape[2]
    ge = np.zeros((n_rows, Cc), dtype=np.float64)
    lib().lnh_oracle_grid_backward(_p(g), _p(inputs), _p(offsets), _p(ge), C.c_uint32(B), C.c_uint32(D),
                                   C.c_uint32(Cc), C.c_uint32(L), C.c_float(S), C.c_uint32(H), C.c_uint32(gridtype),
                                   C.c_int(int(align_corners)), C.c_uint32(interp), C.c_int(dtype),
                                   C.c_int(int(quantize_contrib)))
    return ge


def grad_total_variation(inputs, embeddings, offsets, weight, S, H, gridtype=0, align_corners=False):
    """kernel_grad_tv (gridencoder.cu:695-807).  embeddings float32 or float16 [rows, C]; returns the increment of
    grad_embeddings as float64 [rows, C] (an order-free sum of what the kernel adds with atomics)."""
    inputs, offsets = _f32(inputs), _i32(offsets)
    emb = np.ascontiguousarray(embeddings)
    assert emb.dtype in (np.float32, np.float16)
    dtype = 0 if emb.dtype == np.float32 else 1
    if dtype == 1:  # the reference's kernel reads `inputs` as scalar_t too
        inputs = _f32(inputs.astype(np.float16))
    table = _f32(emb.astype(np.float32))
    B, D = inputs.shape
    L, Cc = offsets.shape[0] - 1, emb.shape[1]
    out = np.zeros((emb.shape[0], Cc), dtype=np.float64)
    lib().lnh_oracle_grad_tv(_p(inputs), _p(table), _p(offsets), _p(out), C.c_float(weight), C.c_uint32(B), C.c_uint32(D),
                             C.c_uint32(Cc), C.c_uint32(L), C.c_float(S), C.c_uint32(H), C.c_uint32(gridtype),
                             C.c_int(int(align_corners)), C.c_int(dtype))
    return out


def sph_from_ray(rays_o, rays_d, radius):
    """kernel_sph_from_ray (raymarching.cu:182-217) -> coords [N, 2] in [-1, 1]."""
    rays_o, rays_d = _f32(rays_o), _f32(rays_d)
    N = rays_o.shape[0]
    out = np.empty((N, 2), dtype=np.float32)
    lib().lnh_oracle_sph_from_ray(_p(rays_o), _p(rays_d), C.c_float(radius), C.c_uint32(N), _p(out))
    return out


def grid_input_backward(grad, dy_dx):
    g, dy = _f32(grad), _f32(dy_dx)
    L, B, Cc = g.shape
    D = dy.shape[2]
    out = np.empty((B, D), dtype=np.float32)
    lib().lnh_oracle_grid_input_backward(_p(g), _p(dy), _p(out), C.c_uint32(B), C.c_uint32(D), C.c_uint32(Cc),
                                         C.c_uint32(L))
    return out


def morton3D(coords):
    c = _i32(coords)
    out = np.empty(c.shape[0], dtype=np.int32)
    lib().lnh_oracle_morton3D(_p(c), C.c_uint32(c.shape[0]), _p(out))
    return out


def morton3D_invert(indices):
    i = _i32(indices)
    out = np.empty((i.shape[0], 3), dtype=np.int32)
    lib().lnh_oracle_morton3D_invert(_p(i), C.c_uint32(i.shape[0]), _p(out))
    return out


def packbits(grid, thresh):
    g = _f32(grid).reshape(-1)
    N = g.shape[0] // 8
    out = np.empty(N, dtype=np.uint8)
    lib().lnh_oracle_packbits(_p(g), C.c_uint32(N), C.c_float(thresh), _p(out))
    return out


def mip_levels(xyz, dt, cascade, H):
    xyz, dt = _f32(xyz), _f32(dt)
    N = xyz.shape[0]
    a, b = np.empty(N, dtype=np.int32), np.empty(N, dtype=np.int32)
    lib().lnh_oracle_mip_levels(_p(xyz), _p(dt), C.c_uint32(N), C.c_uint32(cascade), C.c_uint32(H), _p(a), _p(b))
    return a, b


def occupancy_lookup(xyz, dt, bitfield, bound, cascade, H):
    xyz, dt = _f32(xyz), _f32(dt)
    bf = np.ascontiguousarray(bitfield, dtype=np.uint8)
    N = xyz.shape[0]
    ci, occ = np.empty(N, dtype=np.uint32), np.empty(N, dtype=np.uint8)
    lib().lnh_oracle_occupancy_lookup(_p(xyz), _p(dt), _p(bf), C.c_float(bound), C.c_uint32(N), C.c_uint32(cascade),
                                      C.c_uint32(H), _p(ci), _p(occ))
    return ci, occ


def near_far_from_aabb(rays_o, rays_d, aabb, min_near):
    o, d, a = _f32(rays_o), _f32(rays_d), _f32(aabb)
    N = o.shape[0]
    n, f = np.empty(N, dtype=np.float32), np.empty(N, dtype=np.float32)
    lib().lnh_oracle_near_far_from_aabb(_p(o), _p(d), _p(a), C.c_uint32(N), C.c_float(min_near), _p(n), _p(f))
    return n, f


def march_rays_train(rays_o, rays_d, bitfield, bound, dt_gamma, max_steps, cascade, H, M, nears, fars, noises):
    o, d = _f32(rays_o), _f32(rays_d)
    bf = np.ascontiguousarray(bitfield, dtype=np.uint8)
    nears, fars, noises = _f32(nears), _f32(fars), _f32(noises)
    N = o.shape[0]
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    rays = np.zeros((N, 3), np.int32)
    counter = np.zeros(2, np.int32)
    lib().lnh_oracle_march_rays_train(_p(o), _p(d), _p(bf), C.c_float(bound), C.c_float(dt_gamma),
                                      C.c_uint32(max_steps), C.c_uint32(N), C.c_uint32(cascade), C.c_uint32(H),
                                      C.c_uint32(M), _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(rays),
                                      _p(counter), _p(noises))
    return xyzs, dirs, deltas, rays, counter


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
    s, c, dl, r = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
    M, N = s.shape[0], r.shape[0]
    ws, dep, img = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    lib().lnh_oracle_composite_rays_train_forward(_p(s), _p(c), _p(dl), _p(r), C.c_uint32(M), C.c_uint32(N),
                                                  C.c_float(T_thresh), _p(ws), _p(dep), _p(img))
    return ws, dep, img


def composite_rays_train_backward(grad_ws, grad_img, sigmas, rgbs, deltas, rays, ws, img, T_thresh=1e-4):
    gws, gi = _f32(grad_ws), _f32(grad_img)
    s, c, dl, r = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
    ws, img = _f32(ws), _f32(img)
    M, N = s.shape[0], r.shape[0]
    gs, gc = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
    lib().lnh_oracle_composite_rays_train_backward(_p(gws), _p(gi), _p(s), _p(c), _p(dl), _p(r), _p(ws), _p(img),
                                                   C.c_uint32(M), C.c_uint32(N), C.c_float(T_thresh), _p(gs), _p(gc))
    return gs, gc


def chamfer_nn(xyz1, xyz2):
    a, b = _f32(xyz1), _f32(xyz2)
    n, m = a.shape[0], b.shape[0]
    dist, idx = np.zeros(n, np.float32), np.zeros(n, np.int32)
    lib().lnh_oracle_chamfer_nn(_p(a), C.c_uint32(n), _p(b), C.c_uint32(m), _p(dist), _p(idx))
    return dist, idx


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, cascade, H, bitfield,
               nears, fars, noises):
    """raymarching.cu:808-928.  Returns xyzs, dirs, deltas [n_alive*n_step, 3|3|2] (unused slots zero)."""
    ra, rt = _i32(rays_alive), _f32(rays_t)
    o, d = _f32(rays_o), _f32(rays_d)
    bf = np.ascontiguousarray(bitfield, dtype=np.uint8)
    nears, fars, noises = _f32(nears), _f32(fars), _f32(noises)
    M = n_alive * n_step
    xyzs, dirs, deltas = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32), np.zeros((M, 2), np.float32)
    lib().lnh_oracle_march_rays(C.c_uint32(n_alive), C.c_uint32(n_step), _p(ra), _p(rt), _p(o), _p(d), C.c_float(bound),
                                C.c_float(dt_gamma), C.c_uint32(max_steps), C.c_uint32(cascade), C.c_uint32(H), _p(bf),
                                _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(noises))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """raymarching.cu:966-1053.  Returns updated COPIES (rays_alive, rays_t, weights_sum, depth, image)."""
    ra, rt = _i32(rays_alive).copy(), _f32(rays_t).copy()
    s, c, dl = _f32(sigmas), _f32(rgbs), _f32(deltas)
    ws, dep, img = _f32(weights_sum).copy(), _f32(depth).copy(), _f32(image).copy()
    lib().lnh_oracle_composite_rays(C.c_uint32(n_alive), C.c_uint32(n_step), C.c_float(T_thresh), _p(ra), _p(rt), _p(s),
                                    _p(c), _p(dl), _p(ws), _p(dep), _p(img))
    return ra, rt, ws, dep, img
