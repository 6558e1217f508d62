"""G9: the integer index functions against known answers derived from the CUDA sources with Python integers
(tests/golden/make_g9_kats.py — nothing of this repo is used to make them): the hash-grid row index
(gridencoder.cu:53-93, incl. the dense -> hash switch level, tiled grids, align_corners, D = 2 / 3 / 4 and the levels whose
dense stride wraps in uint32), the Morton code of the occupancy grid (raymarching.cu:71-95, incl. the signed shifts of
kernel_morton3D_invert, raymarching.cu:256-272), the occupancy cell lookup built on it (mip level from position and step,
cell, bit test: raymarching.cu:51-69, 386-408, on dyadic inputs for which every float operation is exact) and
kernel_packbits (286-306, incl. NaN / inf / values equal to the threshold).  CPU: both oracle restatements (NumPy, C).
GPU: the HIP kernels through lnh_grid_corner_indices / lnh_morton3D / lnh_morton3D_invert / lnh_occupancy_lookup /
lnh_packbits.  Bit-exact everywhere.

And the hash grid's VALUES (interpolation gridencoder.cu:95-263, scatter-add 265-362) on dyadic inputs, tables and gradients
for which every float operation is exact in the accumulator type — answers computed with Fractions that hold whatever the
CUDA compiler contracts into FMAs and in whatever order atomics arrive: both oracle restatements, lnh_grid_encode_forward,
the atomic backward and the bucketed backward, fp32 and fp16 tables, bit-exact."""
import os

import numpy as np
import pytest

from oracle import c_oracle, grid_ref

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_integer_kats.npz"))
H = int(G["base_resolution"])
S = 1.0  # per_level_scale 2


def _cases(ci):
    D, L, log2, gridtype, align = (int(v) for v in G["cfg"][ci])
    m = G["grid_cfg"] == ci
    return D, L, gridtype, bool(align), G["offsets"][ci, :L + 1].astype(np.int32), m


def _points(ci):
    """One input point per case whose corner-0 cell is the case's `base`: pos = x * scale + (0 | 0.5) with scale =
    resolution - 1 (gridencoder.cu:146-153); x = base / scale puts pos at base + 0.5 (align_corners: (base + 0.5) / scale),
    half a cell away from either neighbour — the float error of x * scale is < 0.01 at resolution 65536."""
    D, L, gridtype, align, off, m = _cases(ci)
    base = G["grid_base"][m][:, :D].astype(np.float64)
    scale = (G["grid_resolution"][m].astype(np.float64) - 1.0)[:, None]
    x = ((base + 0.5) / scale) if align else (base / scale)
    assert x.min() >= 0 and x.max() <= 1
    return x.astype(np.float32)


@pytest.mark.parametrize("ci", range(6))
def test_numpy_oracle_row_index(ci):
    D, L, gridtype, align, off, m = _cases(ci)
    pos, size, res = G["grid_pos"][m][:, :D], G["grid_hashmap_size"][m], G["grid_resolution"][m]
    for hs, r in sorted(set(zip(size.tolist(), res.tolist()))):
        k = (size == hs) & (res == r)
        got = grid_ref.grid_index(pos[k].astype(np.uint32), hs, r, gridtype, align)
        np.testing.assert_array_equal(got, G["grid_row"][m][k])
    np.testing.assert_array_equal(grid_ref.make_offsets(D, L, 2.0, H, int(G["cfg"][ci][2]), align_corners=align), off)


@pytest.mark.parametrize("ci", range(6))
def test_c_oracle_row_index(ci):
    D, L, gridtype, align, off, m = _cases(ci)
    idx = c_oracle.grid_indices(_points(ci), off, 1, S, H, gridtype, align)  # [L, K, 2^D], C = 1: index = row
    k = np.arange(int(m.sum()))
    np.testing.assert_array_equal(idx[G["grid_level"][m], k, G["grid_corner"][m]], G["grid_row"][m])


def test_oracle_morton():
    np.testing.assert_array_equal(c_oracle.morton3D(G["morton_coords"].view(np.int32)).view(np.uint32), G["morton_code"])
    np.testing.assert_array_equal(c_oracle.morton3D_invert(G["invert_in"].view(np.int32)).view(np.uint32), G["invert_out"])


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(6))
def test_hip_row_index(ci):
    import torch
    from gpu_util import call, dev, host
    D, L, gridtype, align, off, m = _cases(ci)
    x = _points(ci)
    for C_ in (1, 2):
        out = torch.empty((L, x.shape[0], 1 << D), dtype=torch.int32, device="cuda")
        call("lnh_grid_corner_indices", dev(x), torch.from_numpy(off), out, x.shape[0], D, C_, L, S, H, gridtype, int(align))
        got = host(out).view(np.uint32)[G["grid_level"][m], np.arange(x.shape[0]), G["grid_corner"][m]]
        np.testing.assert_array_equal(got, G["grid_row"][m] * C_)


@pytest.mark.gpu
def test_hip_morton():
    import torch
    from gpu_util import call, dev, host
    c = G["morton_coords"].view(np.int32)
    out = torch.empty(c.shape[0], dtype=torch.int32, device="cuda")
    call("lnh_morton3D", dev(c), c.shape[0], out)
    np.testing.assert_array_equal(host(out).view(np.uint32), G["morton_code"])
    i = G["invert_in"].view(np.int32)
    back = torch.empty((i.shape[0], 3), dtype=torch.int32, device="cuda")
    call("lnh_morton3D_invert", dev(i), i.shape[0], back)
    np.testing.assert_array_equal(host(back).view(np.uint32), G["invert_out"])


# ---- occupancy cell lookup (raymarching.cu:51-69, 386-408) and kernel_packbits (286-306)
def _bitfield(C_, Hh=128):
    """The field of tests/golden/make_g9_kats.py::occupancy_bit: bit `index` = ((index * 2654435761) >> 13) & 1."""
    idx = np.arange(C_ * Hh ** 3, dtype=np.uint64)
    return np.packbits((((idx * np.uint64(2654435761)) >> np.uint64(13)) & np.uint64(1)).astype(np.uint8), bitorder="little")


def _occ_groups():
    for C_, bound in sorted(set(zip(G["occ_C"].tolist(), G["occ_bound"].tolist()))):
        yield C_, bound, (G["occ_C"] == C_) & (G["occ_bound"] == bound)


def test_oracle_occupancy_lookup_and_packbits():
    for C_, bound, m in _occ_groups():
        ci, occ = c_oracle.occupancy_lookup(G["occ_xyz"][m], G["occ_dt"][m], _bitfield(C_), bound, C_, 128)
        np.testing.assert_array_equal(ci, G["occ_index"][m])
        np.testing.assert_array_equal(occ != 0, G["occ_bit"][m] != 0)
        a, b = c_oracle.mip_levels(G["occ_xyz"][m], G["occ_dt"][m], C_, 128)
        np.testing.assert_array_equal(np.maximum(a, b), G["occ_level"][m])
        # the index decomposes into the level and the Morton code of the cell
        np.testing.assert_array_equal(ci // 128 ** 3, G["occ_level"][m])
        np.testing.assert_array_equal(c_oracle.morton3D_invert((ci % 128 ** 3).astype(np.int32)), G["occ_cell"][m])
    np.testing.assert_array_equal(c_oracle.packbits(G["pack_grid"], float(G["pack_thresh"])), G["pack_bytes"])


@pytest.mark.gpu
def test_hip_occupancy_lookup_and_packbits():
    import torch
    from gpu_util import call, dev, host
    for C_, bound, m in _occ_groups():
        n = int(m.sum())
        ci = torch.empty(n, dtype=torch.int32, device="cuda")
        occ = torch.empty(n, dtype=torch.uint8, device="cuda")
        call("lnh_occupancy_lookup", dev(G["occ_xyz"][m]), dev(G["occ_dt"][m]), dev(_bitfield(C_)), float(bound), n, C_, 128, ci, occ)
        np.testing.assert_array_equal(host(ci).view(np.uint32), G["occ_index"][m])
        np.testing.assert_array_equal(host(occ) != 0, G["occ_bit"][m] != 0)
    out = torch.empty(G["pack_bytes"].shape[0], dtype=torch.uint8, device="cuda")
    call("lnh_packbits", dev(G["pack_grid"]), out.numel(), float(G["pack_thresh"]), out)
    np.testing.assert_array_equal(host(out), G["pack_bytes"])


# ---- hash-grid VALUES on exact arithmetic (gridencoder.cu:95-263 interpolation, 265-362 scatter-add)
def _value_table(f16):
    """tests/golden/make_g9_kats.py::table_value on every row of the 6-level table."""
    off = G["val_offsets"]
    row = np.arange(int(off[-1]), dtype=np.uint64)[:, None]
    ch = np.arange(2, dtype=np.uint64)[None, :]
    k = (((row * np.uint64(2654435761) + ch * np.uint64(40503) + np.uint64(12345)) >> np.uint64(7)) & np.uint64(15)).astype(np.int64) - 8
    return (k * 8).astype(np.float16) if f16 else (k / 8.0).astype(np.float32)


def _dense(rows, vals, n):
    out = np.zeros((n, 2), dtype=np.float64)
    out[rows] = vals
    return out


@pytest.mark.parametrize("tag", ["f32", "f16"])
def test_oracles_grid_values_on_exact_arithmetic(tag):
    f16 = tag == "f16"
    off, x = G["val_offsets"], G[f"val_{tag}_x"]
    tab = _value_table(f16)
    for fwd in (c_oracle.grid_forward(x, tab, off, S, H)[0], grid_ref.forward(x, tab, off, S, H)):
        np.testing.assert_array_equal(fwd.astype(np.float32), G[f"val_{tag}_fwd"])
    g = G[f"val_{tag}_grad"].astype(np.float16 if f16 else np.float32)
    want = _dense(G[f"val_{tag}_grad_rows"], G[f"val_{tag}_grad_table"], int(off[-1]))
    np.testing.assert_array_equal(c_oracle.grid_backward(g, x, off, int(off[-1]), S, H), want)
    np.testing.assert_array_equal(grid_ref.backward(g, x, off, int(off[-1]), S, H), want)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["f32", "f16"])
def test_hip_grid_values_on_exact_arithmetic(tag):
    import torch
    from gpu_util import call, dev, host
    from lidarnerf import _hip
    f16 = tag == "f16"
    dt, code = (torch.float16, 1) if f16 else (torch.float32, 0)
    off, x = G["val_offsets"], G[f"val_{tag}_x"]
    B, L, rows = x.shape[0], len(off) - 1, int(off[-1])
    tab, offh = dev(_value_table(f16)), torch.from_numpy(off)
    out = torch.empty((L, B, 2), dtype=dt, device="cuda")
    call("lnh_grid_encode_forward", dev(x), tab, offh, out, B, 3, 2, L, S, H, None, 0, 0, 0, code)
    np.testing.assert_array_equal(host(out).astype(np.float32), G[f"val_{tag}_fwd"])
    g = dev(G[f"val_{tag}_grad"].astype(np.float16 if f16 else np.float32))
    want = _dense(G[f"val_{tag}_grad_rows"], G[f"val_{tag}_grad_table"], rows)
    # the atomic path (the reference's own method) and the bucketed path (the product's)
    ge = torch.zeros((rows, 2), dtype=dt, device="cuda")
    call("lnh_grid_encode_backward", g, dev(x), None, offh, ge, B, 3, 2, L, S, H, None, None, 0, 0, 0, code)
    np.testing.assert_array_equal(host(ge).astype(np.float64), want)
    need = _hip.lib().lnh_grid_backward_workspace_size(offh.data_ptr(), B, 3, 2, L, S, H, 0, 0, code)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    ge2 = torch.zeros((rows, 2), dtype=dt, device="cuda")
    call("lnh_grid_encode_backward_ws", g, dev(x), offh, ge2, B, 3, 2, L, S, H, 0, 0, 0, code, ws, need)
    np.testing.assert_array_equal(host(ge2).astype(np.float64), want)
