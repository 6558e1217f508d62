#!/usr/bin/env python3
"""G9 — integer known-answer vectors for the two index functions no reference-produced golden vector reaches (the reference's
CUDA extensions do not compile here):

  * the hash-grid row index, lidarnerf/gridencoder/src/gridencoder.cu:53-93 (`fast_hash`, `get_grid_index`), with the level
    tables of lidarnerf/gridencoder/grid.py:179-192 (`offsets`);
  * the Morton code of the occupancy grid, lidarnerf/raymarching/src/raymarching.cu:71-95 (`__expand_bits`, `__morton3D`,
    `__morton3D_invert`), the cell lookup built on it (51-69, 386-408: mip level from position and step, cell, bit test) on
    inputs for which every float operation is exact, and `kernel_packbits` (286-306);
  * the interpolation and the scatter-add of the hash grid (gridencoder.cu:95-263, 265-362) on dyadic inputs / tables /
    gradients for which every float operation is exact in fp32 resp. fp16 — values that hold whatever the compiler
    contracts into FMAs and in whatever order atomics arrive.

Both are pure uint32 arithmetic, so the answers can be derived WITHOUT any of this repo's code: below, the C expressions are
evaluated with Python's unbounded integers and an explicit `& 0xFFFFFFFF` wherever C's uint32_t would wrap.  Nothing is imported
from `oracle/` or the package; the oracle (NumPy and C) and the HIP kernels are then checked against the file this writes
(tests/test_g9_integer_kats.py).  The Morton answers for 10-bit coordinates are cross-checked against the definition of a
Morton code (bit i of x -> bit 3i, y -> 3i+1, z -> 3i+2) inside this script.

    python tests/golden/make_g9_kats.py        # writes tests/golden/g9_integer_kats.npz

Cases (all with per_level_scale = 2, i.e. S = 1.0 exactly: the level resolution 16 * 2^l involves no float rounding):
  cfg 0  D=3 hash  2^19 rows  — levels 0-2 dense, level 3 is the dense->hash switch; level 12 (resolution 65536, R = 65537): the
                                 dense stride WRAPS in uint32 (65537^2 mod 2^32 = 131073, 131073 * 65537 mod 2^32 = 196609 <= 2^19),
                                 so `stride > hashmap_size` is false after the loop, the level is NOT hashed and its rows are
                                 formed with the wrapped strides — what the CUDA code computes (BASELINE's configs stop at 32768,
                                 where 32769^2 < 2^32 and the level is hashed as intended)
  cfg 1  D=3 hash  2^24 rows  — switch at level 4; level 12 wraps the same way
  cfg 2  D=3 tiled 2^19 rows  — gridtype 1: a level too large for the table keeps the PARTIAL dense sum (loop left early)
  cfg 3  D=3 hash  2^19 rows, align_corners (stride = resolution instead of resolution + 1)
  cfg 4  D=2 hash  2^19 rows
  cfg 5  D=4 hash  2^19 rows, 8 levels
Each case: a base cell `base` (the pos_grid of corner 0) and a corner number c; pos_grid = base + ((c >> d) & 1).
"""
import os

import numpy as np

M = 0xFFFFFFFF
PRIMES = (1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737)  # gridencoder.cu:56-58


def fast_hash(pos):  # gridencoder.cu:53-67
    result = 0
    for i, p in enumerate(pos):
        result ^= (p * PRIMES[i]) & M
    return result


def get_grid_index(gridtype, align_corners, hashmap_size, resolution, pos):  # gridencoder.cu:69-93, row (before * C + ch)
    stride, index, d, D = 1, 0, 0, len(pos)
    while d < D and stride <= hashmap_size:
        index = (index + pos[d] * stride) & M
        stride = (stride * (resolution if align_corners else resolution + 1)) & M
        d += 1
    if gridtype == 0 and stride > hashmap_size:
        index = fast_hash(pos)
    return index % hashmap_size


def level_offsets(D, L, H, log2_hashmap_size, align_corners):  # grid.py:179-192 with per_level_scale = 2
    offsets, offset = [], 0
    for i in range(L):
        resolution = H * 2 ** i
        params = min(2 ** log2_hashmap_size, (resolution if align_corners else resolution + 1) ** D)
        params = -(-params // 8) * 8
        offsets.append(offset)
        offset += params
    offsets.append(offset)
    assert offset < 2 ** 31
    return offsets


def expand_bits(v):  # raymarching.cu:71-77
    v = ((v * 0x00010001) & M) & 0xFF0000FF
    v = ((v * 0x00000101) & M) & 0x0F00F00F
    v = ((v * 0x00000011) & M) & 0xC30C30C3
    v = ((v * 0x00000005) & M) & 0x49249249
    return v


def morton3D(x, y, z):  # raymarching.cu:79-86
    return (expand_bits(x) | ((expand_bits(y) << 1) & M) | ((expand_bits(z) << 2) & M)) & M


def morton3D_invert(x):  # raymarching.cu:88-95
    x = x & 0x49249249
    x = (x | (x >> 2)) & 0xC30C30C3
    x = (x | (x >> 4)) & 0x0F00F00F
    x = (x | (x >> 8)) & 0xFF0000FF
    x = (x | (x >> 16)) & 0x0000FFFF
    return x


def interleave_by_definition(x, y, z):
    out = 0
    for i in range(10):
        out |= ((x >> i) & 1) << (3 * i) | ((y >> i) & 1) << (3 * i + 1) | ((z >> i) & 1) << (3 * i + 2)
    return out


def occupancy_bit(index):
    """The occupancy bitfield the lookups are checked on: bit `index` is set iff this is 1 (tests build the same field)."""
    return ((index * 2654435761) >> 13) & 1


def occupancy_kats():
    """Cell lookup of kernel_march_rays_train (raymarching.cu:386-408) with mip_from_pos / mip_from_dt (51-69): inputs are
    dyadic rationals, so every float operation of the CUDA code is exact and the answer follows with Fractions:
        level = max(frexp-exponent of max |x_i|, frexp-exponent of dt * H / 2), clamped to [0, C - 1]
        mip_bound = min(2^level, bound);  n_i = trunc(clamp((x_i / mip_bound + 1) * H / 2, 0, H - 1))
        index = level * H^3 + morton3D(n);  occ = bit `index` of the field."""
    from fractions import Fraction as Fr
    import math
    H = 128
    cases = []  # (C, bound, x, y, z, dt)
    pts = [Fr(0), Fr(1, 2), Fr(-1, 2), Fr(1), Fr(-1), Fr(63, 64), Fr(-127, 128), Fr(1, 128), Fr(-1, 256), Fr(3, 4), Fr(5, 4), Fr(2),
           Fr(-2), Fr(255, 128), Fr(3), Fr(-7, 2), Fr(4), Fr(-4), Fr(1, 1024), Fr(-33, 64)]
    dts = [Fr(1, 4096), Fr(1, 256), Fr(1, 128), Fr(1, 64), Fr(3, 128), Fr(1, 32), Fr(1, 16)]
    rng = Lcg(99)
    for C_, bound in ((1, 1), (2, 2), (3, 4)):
        inside = [p for p in pts if abs(p) <= bound]
        for i in range(140):
            x, y, z = (inside[rng.below(len(inside))] for _ in range(3))
            cases.append((C_, bound, x, y, z, dts[rng.below(len(dts))]))
        for p in inside:  # every special value once on every axis, with the smallest step
            cases += [(C_, bound, p, Fr(0), Fr(0), dts[0]), (C_, bound, Fr(0), p, Fr(1, 128), dts[0]), (C_, bound, Fr(1, 256), Fr(0), p, dts[1])]
    f32 = lambda q: float(np.float32(float(q)))
    out = {k: [] for k in ("occ_C", "occ_bound", "occ_xyz", "occ_dt", "occ_level", "occ_cell", "occ_index", "occ_bit")}
    for C_, bound, x, y, z, dt in cases:
        for q in (x, y, z, dt):
            assert Fr(f32(q)) == q  # float32 holds the input exactly
        mx = max(abs(x), abs(y), abs(z))
        e_pos = math.frexp(float(mx))[1] if mx else 0  # frexpf(0) = (0, 0)
        e_dt = math.frexp(float(dt * H / 2))[1]
        level = max(min(C_ - 1, max(0, e_pos)), min(C_ - 1, max(0, e_dt)))
        mip_bound = min(Fr(2) ** level, Fr(bound))
        cell = []
        for v in (x, y, z):
            t = (v / mip_bound + 1) * H / 2
            assert Fr(f32(v / mip_bound)) == v / mip_bound and Fr(f32(t)) == t  # the intermediate floats are exact too
            cell.append(int(min(max(t, 0), H - 1)))  # clamp, then the C cast truncates (all values >= 0)
        index = level * H ** 3 + morton3D(*cell)
        out["occ_C"].append(C_); out["occ_bound"].append(bound); out["occ_xyz"].append([f32(x), f32(y), f32(z)])
        out["occ_dt"].append(f32(dt)); out["occ_level"].append(level); out["occ_cell"].append(cell); out["occ_index"].append(index)
        out["occ_bit"].append(occupancy_bit(index))
    assert len(set(out["occ_level"])) == 3 and 0 < sum(out["occ_bit"]) < len(cases)
    return {"occ_C": np.array(out["occ_C"], np.int32), "occ_bound": np.array(out["occ_bound"], np.float32),
            "occ_xyz": np.array(out["occ_xyz"], np.float32), "occ_dt": np.array(out["occ_dt"], np.float32),
            "occ_level": np.array(out["occ_level"], np.int32), "occ_cell": np.array(out["occ_cell"], np.int32),
            "occ_index": np.array(out["occ_index"], np.uint32), "occ_bit": np.array(out["occ_bit"], np.uint8)}


def table_value(global_row, ch, f16):
    """The hash table the value KATs are computed on (tests build the same one): a small dyadic number per (row, channel) —
    fp32 set: k / 8 with k in -8 .. 7; fp16 set: 8 k."""
    k = (((global_row * 2654435761 + ch * 40503 + 12345) >> 7) & 15) - 8
    return k * 8 if f16 else k / 8.0


def grid_value_kats():
    """kernel_grid / kernel_grid_backward (gridencoder.cu:95-263, 265-362) — the INTERPOLATION and the scatter, linear, D = 3,
    hash grid, per_level_scale 2 (6 levels: 0-2 dense, 3-5 hashed, 2^19 rows) — on inputs for which every float operation of
    the CUDA code is EXACT, whatever it contracts into FMAs and in whatever order atomics arrive: x = j / 2^m, so pos =
    x * scale + 0.5 and its fraction are dyadic, weights have <= 3 (m + 1) bits, table values / gradients are small dyadic
    numbers, and every partial sum fits the accumulator type (fp32 set: m = 5; fp16 set — the reference accumulates in
    __half — m = 1 with table values and gradients multiples of 8).  Answers with Fractions + the integer row index above."""
    from fractions import Fraction as Fr
    D, L, log2 = 3, 6, 19
    offs = level_offsets(D, L, H, log2, 0)
    out = {"val_offsets": np.array(offs, dtype=np.int32)}
    rng = Lcg(77)
    for tag, m, f16 in (("f32", 5, False), ("f16", 1, True)):
        B = 48
        js = [[rng.below((1 << m) + 1) for _ in range(D)] for _ in range(B)]
        js[0], js[1], js[2] = [0, 0, 0], [1 << m] * 3, [1 << m, 0, 1 << (m - 1)]
        x = [[Fr(j, 1 << m) for j in row] for row in js]
        grad = [[[(rng.below(9) - 4) * (8 if f16 else 1) for _ in range(2)] for _ in range(B)] for _ in range(L)]
        fwd = [[[Fr(0), Fr(0)] for _ in range(B)] for _ in range(L)]
        gtab, gabs = {}, {}
        for l in range(L):
            res = H * 2 ** l
            scale = res - 1
            rows_l = offs[l + 1] - offs[l]
            for b in range(B):
                pos = [xd * scale + Fr(1, 2) for xd in x[b]]
                pg = [int(p) for p in pos]  # floor (pos >= 0)
                fr = [p - g for p, g in zip(pos, pg)]
                for c in range(1 << D):
                    w, cell = Fr(1), []
                    for d in range(D):
                        up = (c >> d) & 1
                        w *= fr[d] if up else 1 - fr[d]
                        cell.append(pg[d] + up)
                    row = offs[l] + get_grid_index(0, 0, rows_l, res, cell)
                    for ch in range(2):
                        fwd[l][b][ch] += w * Fr(table_value(row, ch, f16))
                        if w:
                            gtab[(row, ch)] = gtab.get((row, ch), Fr(0)) + w * grad[l][b][ch]
                            gabs[(row, ch)] = gabs.get((row, ch), Fr(0)) + abs(w * grad[l][b][ch])
        exact = lambda q, t: Fr(float(t(float(q)))) == q
        ft = np.float16 if f16 else np.float32
        assert all(exact(v, ft) for lv in fwd for pt in lv for v in pt) and all(exact(v, ft) for v in gtab.values())
        # ... and every PARTIAL sum, in any order: forward terms are multiples of 2^-21 (fp32) / integers (fp16) inside a sum of
        # weights 1; backward contributions are multiples of 2^-18 (fp32) / integers (fp16), so a row whose absolute sum stays
        # below 2^(24 - 18) (fp32) / 2048 (fp16) never needs more bits than the accumulator has
        assert max(gabs.values()) < (2048 if f16 else 64)
        rows_touched = sorted({r for r, _ in gtab})
        out[f"val_{tag}_x"] = np.array([[float(v) for v in row] for row in x], dtype=np.float32)
        out[f"val_{tag}_grad"] = np.array(grad, dtype=np.float32)
        out[f"val_{tag}_fwd"] = np.array([[[float(v) for v in pt] for pt in lv] for lv in fwd], dtype=np.float32)
        out[f"val_{tag}_grad_rows"] = np.array(rows_touched, dtype=np.int64)
        out[f"val_{tag}_grad_table"] = np.array([[float(gtab.get((r, ch), 0)) for ch in range(2)] for r in rows_touched],
                                                dtype=np.float32)
    return out


def packbits_kats():
    """kernel_packbits (raymarching.cu:286-306): bit i of byte n = grid[8 n + i] > thresh (strictly; NaN compares false)."""
    thresh = 0.5
    vals = [0.0, 0.5, 0.50000006, 0.49999997, 1.0, -1.0, float("inf"), float("-inf"), float("nan"), 1e-30, 0.75, 0.25, 3.0e38, -0.0, 0.5, 0.6]
    rng = Lcg(5)
    grid = [vals[rng.below(len(vals))] for _ in range(8 * 64)]
    grid[:16] = vals
    g32 = np.array(grid, dtype=np.float32)
    t32 = float(np.float32(thresh))
    by = []
    for n in range(len(grid) // 8):
        b = 0
        for i in range(8):
            v = float(g32[8 * n + i])
            if v == v and v > t32:
                b |= 1 << i
        by.append(b)
    return g32, thresh, np.array(by, dtype=np.uint8)


class Lcg:  # a generator that is its own specification (numerical recipes' constants)
    def __init__(self, seed):
        self.s = seed & M

    def below(self, n):
        self.s = (self.s * 1664525 + 1013904223) & M
        return (self.s >> 8) % n


CONFIGS = [  # D, L, log2_hashmap_size, gridtype, align_corners
    (3, 13, 19, 0, 0),
    (3, 13, 24, 0, 0),
    (3, 13, 19, 1, 0),
    (3, 13, 19, 0, 1),
    (2, 13, 19, 0, 0),
    (4, 8, 19, 0, 0),
]
H = 16


def main():
    rng = Lcg(9)
    rows = []  # cfg, level, hashmap_size, resolution, base[4], corner, pos[4], row, hashed
    offsets_all = np.zeros((len(CONFIGS), 17), dtype=np.int64)
    for ci, (D, L, log2, gridtype, align) in enumerate(CONFIGS):
        offs = level_offsets(D, L, H, log2, align)
        offsets_all[ci, :L + 1] = offs
        R = [H * 2 ** l for l in range(L)]
        sizes = [offs[l + 1] - offs[l] for l in range(L)]
        full = [(R[l] if align else R[l] + 1) ** D for l in range(L)]
        switch = next((l for l in range(L) if full[l] > 2 ** log2), L - 1)
        levels = sorted({0, max(switch - 1, 0), switch, min(switch + 1, L - 1), (switch + L) // 2, L - 1})
        for l in levels:
            top = R[l] - 2 if align else R[l] - 1  # largest base cell whose + 1 corner is still a grid vertex
            bases = [[0] * D, [top] * D, [top if d == 0 else 0 for d in range(D)], [0 if d == 0 else top for d in range(D)]]
            bases += [[rng.below(top + 1) for _ in range(D)] for _ in range(4)]
            for bi, base in enumerate(bases):
                corners = range(1 << D) if bi < 2 else (rng.below(1 << D), (1 << D) - 1)
                for c in corners:
                    pos = [base[d] + ((c >> d) & 1) for d in range(D)]
                    row = get_grid_index(gridtype, align, sizes[l], R[l], pos)
                    # was the row hashed?  (recomputed the long way: for the record only)
                    stride, d = 1, 0
                    while d < D and stride <= sizes[l]:
                        stride = (stride * (R[l] if align else R[l] + 1)) & M
                        d += 1
                    hashed = int(gridtype == 0 and stride > sizes[l])
                    rows.append([ci, l, sizes[l], R[l]] + (base + [0] * 4)[:4] + [c] + (pos + [0] * 4)[:4] + [row, hashed])
    rows = np.array(rows, dtype=np.int64)
    # spot facts the docstring states, asserted
    lvl12 = rows[(rows[:, 0] == 1) & (rows[:, 1] == 12)]
    assert len(lvl12) and not lvl12[:, -1].any() and (65537 * 65537) & M == 131073 and (131073 * 65537) & M == 196609
    assert rows[(rows[:, 0] == 0) & (rows[:, 1] == 2)][:, -1].sum() == 0 and rows[(rows[:, 0] == 0) & (rows[:, 1] == 3)][:, -1].all()

    # ---- Morton
    coords = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1), (1023, 1023, 1023), (1023, 0, 0), (0, 1023, 0), (0, 0, 1023),
              (127, 127, 127), (128, 0, 64), (5, 9, 3), (341, 682, 511), (512, 256, 128), (1, 2, 4), (1000, 10, 100)]
    coords += [tuple(rng.below(1024) for _ in range(3)) for _ in range(32)]
    for x, y, z in coords:
        assert morton3D(x, y, z) == interleave_by_definition(x, y, z)
        assert tuple(morton3D_invert(morton3D(x, y, z) >> s) for s in range(3)) == (x, y, z)
    # beyond 10 bits the code is whatever the masks leave (the CUDA code is called with coordinates < 128 only; still uint32)
    coords += [(1024, 0, 0), (1025, 2047, 4095), (0xFFFF, 0xFFFF, 0xFFFF), (0x12345, 0xABCDE, 0xFFFFF), (M, 0, M), (M, M, M)]
    mort = [morton3D(*c) for c in coords]
    inv_in = mort[:24] + [M, 0x80000000, 0x49249249, 0x92492492, 0x24924924, 0xDEADBEEF, 0x12345678, 0x3FFFFFFF, 0x40000000]
    # kernel_morton3D_invert, raymarching.cu:256-272: the index is read as a SIGNED int and `ind >> 1`, `ind >> 2` are arithmetic
    # shifts before the conversion to uint32 — with bit 31 set, `ind >> 2` carries the sign into bit 30, which the first mask
    # keeps (a logical shift would not): Python's >> on a negative int is that arithmetic shift
    def as_int32(v):
        return v - (1 << 32) if v & 0x80000000 else v
    inv_out = [[morton3D_invert((as_int32(v) >> s) & M) for s in range(3)] for v in inv_in]
    assert any(morton3D_invert((as_int32(v) >> 2) & M) != morton3D_invert(v >> 2) for v in inv_in)  # the case is in the set

    occ = occupancy_kats()
    occ.update(grid_value_kats())
    pb_grid, pb_thresh, pb_bytes = packbits_kats()

    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g9_integer_kats.npz")
    np.savez_compressed(
        out,
        cfg=np.array(CONFIGS, dtype=np.int64), base_resolution=np.int64(H), offsets=offsets_all,
        grid_cfg=rows[:, 0], grid_level=rows[:, 1], grid_hashmap_size=rows[:, 2].astype(np.uint32),
        grid_resolution=rows[:, 3].astype(np.uint32), grid_base=rows[:, 4:8].astype(np.uint32), grid_corner=rows[:, 8],
        grid_pos=rows[:, 9:13].astype(np.uint32), grid_row=rows[:, 13].astype(np.uint32), grid_hashed=rows[:, 14],
        morton_coords=np.array(coords, dtype=np.uint32), morton_code=np.array(mort, dtype=np.uint32),
        invert_in=np.array(inv_in, dtype=np.uint32), invert_out=np.array(inv_out, dtype=np.uint32),
        pack_grid=pb_grid, pack_thresh=np.float32(pb_thresh), pack_bytes=pb_bytes, **occ)
    print(f"{out}: {len(rows)} grid cases ({int(rows[:, -1].sum())} hashed), {len(coords)} Morton codes, {len(inv_in)} inversions, "
          f"{len(occ['occ_index'])} occupancy lookups, {len(pb_bytes)} packed bytes")


if __name__ == "__main__":
    main()
