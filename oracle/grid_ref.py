"""NumPy restatement of the reference hash-grid encoder (independent of lnh_oracle.c).

TEST INFRASTRUCTURE ONLY.  Follows /root/reference:
  lidarnerf/gridencoder/grid.py:141-235        (level table / offsets / input mapping)
  lidarnerf/gridencoder/src/gridencoder.cu:53-93   (fast_hash, get_grid_index — uint32 exact)
  lidarnerf/gridencoder/src/gridencoder.cu:95-209  (kernel_grid forward)
  lidarnerf/gridencoder/src/gridencoder.cu:265-362 (kernel_grid_backward)
Vectorised over points; loops over levels and the 2^D corners.
"""
import numpy as np

PRIMES = np.array([1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737], dtype=np.uint32)


def per_level_scale(desired_resolution, base_resolution, num_levels):
    """grid.py:158-161"""
    return np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))


def make_offsets(input_dim=3, num_levels=16, per_level_scale=2.0, base_resolution=16, log2_hashmap_size=19,
                 align_corners=False):
    """grid.py:179-193: rows per level = min(2^log2_hashmap, (res[+1])^D) rounded up to 8."""
    offsets, offset = [], 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(max_params, (resolution if align_corners else resolution + 1) ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offsets.append(offset)
        offset += n
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32)


def level_geometry(level, S, H):
    """gridencoder.cu:146-148.  exp2 evaluated in float64 and rounded once (same convention as lnh_oracle.c)."""
    e = np.float32(np.float32(level) * np.float32(S))
    p = np.float32(np.exp2(np.float64(e)))
    scale = np.float32(np.float32(p * np.float32(H)) - np.float32(1.0))
    resolution = int(np.ceil(scale)) + 1
    return scale, resolution


def grid_index(pos_grid, hashmap_size, resolution, gridtype=0, align_corners=False):
    """gridencoder.cu:69-93 on uint32 arrays pos_grid [B, D] -> row index [B] (before *C)."""
    D = pos_grid.shape[1]
    with np.errstate(over="ignore"):
        stride = np.uint32(1)
        index = np.zeros(pos_grid.shape[0], dtype=np.uint32)
        d = 0
        R = np.uint32(resolution if align_corners else resolution + 1)
        stride_int = 1  # python int mirror to evaluate the loop condition (uint32 wrap)
        while d < D and stride_int <= hashmap_size:
            index = index + pos_grid[:, d] * np.uint32(stride_int)
            stride_int = (stride_int * int(R)) & 0xFFFFFFFF
            d += 1
        if gridtype == 0 and stride_int > hashmap_size:
            h = np.zeros(pos_grid.shape[0], dtype=np.uint32)
            for dd in range(D):
                h ^= pos_grid[:, dd] * PRIMES[dd]
            index = h
    return index % np.uint32(hashmap_size)


def _fma32(a, b, c):
    """float32 fma emulated through float64 (exact product; one extra rounding, see DESIGN.md tolerance note)."""
    return (a.astype(np.float64) * np.float64(b) + np.float64(c)).astype(np.float32)


def corners(inputs, level, offsets, S, H, gridtype=0, align_corners=False, interp=0):
    """Returns (valid[B] bool, rows [B, 2^D] uint32 level-relative, weights [B, 2^D] float32)."""
    x = np.asarray(inputs, dtype=np.float32)
    B, D = x.shape
    scale, res = level_geometry(level, S, H)
    hm = int(offsets[level + 1] - offsets[level])
    valid = ~np.any((x < 0) | (x > 1), axis=1)
    pos = _fma32(x, scale, 0.0 if align_corners else 0.5)
    pg_f = np.floor(pos)
    pos = (pos - pg_f).astype(np.float32)
    # negative coordinates only occur for invalid points; clamp so the uint cast is defined
    pg = np.where(valid[:, None], pg_f, 0).astype(np.uint32)
    if interp == 1:
        pos = (pos * pos * (np.float32(3.0) - np.float32(2.0) * pos)).astype(np.float32)
    rows = np.empty((B, 1 << D), dtype=np.uint32)
    w = np.empty((B, 1 << D), dtype=np.float32)
    for c in range(1 << D):
        ww = np.ones(B, dtype=np.float32)
        pl = pg.copy()
        for d in range(D):
            if (c >> d) & 1:
                ww = (ww * pos[:, d]).astype(np.float32)
                pl[:, d] = pg[:, d] + np.uint32(1)
            else:
                ww = (ww * (np.float32(1) - pos[:, d])).astype(np.float32)
        rows[:, c] = grid_index(pl, hm, res, gridtype, align_corners)
        w[:, c] = ww
    return valid, rows, w


def forward(inputs, embeddings, offsets, S, H, gridtype=0, align_corners=False, interp=0):
    """kernel_grid: outputs [L, B, C] in the table dtype; accumulation in the table dtype."""
    emb = np.asarray(embeddings)
    x = np.asarray(inputs, dtype=np.float32)
    B = x.shape[0]
    L, Cc = len(offsets) - 1, emb.shape[1]
    out = np.zeros((L, B, Cc), dtype=emb.dtype)
    for l in range(L):
        valid, rows, w = corners(x, l, offsets, S, H, gridtype, align_corners, interp)
        base = int(offsets[l])
        acc = np.zeros((B, Cc), dtype=np.float32)
        for c in range(rows.shape[1]):
            g = emb[base + rows[:, c].astype(np.int64)].astype(np.float32)
            acc = (w[:, c:c + 1].astype(np.float64) * g.astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
            if emb.dtype == np.float16:
                acc = acc.astype(np.float16).astype(np.float32)
        acc[~valid] = 0
        out[l] = acc.astype(emb.dtype)
    return out


def backward(grad, inputs, offsets, n_rows, S, H, gridtype=0, align_corners=False, interp=0):
    """kernel_grid_backward as an order-free float64 scatter-add.  grad [L,B,C] -> [n_rows, C] float64."""
    g = np.asarray(grad)
    x = np.asarray(inputs, dtype=np.float32)
    L, B, Cc = g.shape
    ge = np.zeros((n_rows, Cc), dtype=np.float64)
    for l in range(L):
        valid, rows, w = corners(x, l, offsets, S, H, gridtype, align_corners, interp)
        base = int(offsets[l])
        gl = g[l].astype(np.float32)
        for c in range(rows.shape[1]):
            v = (w[:, c:c + 1] * gl).astype(np.float32)
            if g.dtype == np.float16:
                v = v.astype(np.float16).astype(np.float32)
            v = v[valid]
            np.add.at(ge, base + rows[valid, c].astype(np.int64), v.astype(np.float64))
    return ge


def grad_total_variation(inputs, embeddings, offsets, weight, S, H, gridtype=0, align_corners=False):
    """kernel_grad_tv (gridencoder.cu:695-807), second restatement (float64 arithmetic on the fp32 table values):
    increment of grad_embeddings [rows, C] for the cells visited by `inputs` [B, D] in [0, 1]."""
    emb = np.asarray(embeddings, dtype=np.float64)
    x = np.asarray(inputs, dtype=np.float32)
    B, D = x.shape
    out = np.zeros_like(emb)
    L = len(offsets) - 1
    valid = np.all((x >= 0) & (x <= 1), axis=1)
    for l in range(L):
        scale, res = level_geometry(l, S, H)
        hm = int(offsets[l + 1] - offsets[l])
        tab = emb[offsets[l]:offsets[l + 1]]
        pg = np.floor(_fma32(x, scale, 0.0 if align_corners else 0.5)).astype(np.int64)
        base = grid_index(pg.astype(np.uint32), hm, res, gridtype, align_corners).astype(np.int64)
        results = np.zeros((B, emb.shape[1]))
        idelta = np.zeros((B, emb.shape[1]))
        for d in range(D):
            for step, ok in ((1, pg[:, d] < res), (-1, pg[:, d] > 0)):
                q = pg.copy()
                q[:, d] = np.where(ok, q[:, d] + step, q[:, d])
                nb = grid_index(q.astype(np.uint32), hm, res, gridtype, align_corners).astype(np.int64)
                diff = (tab[base] - tab[nb]) * ok[:, None]
                results += diff
                idelta += diff * diff
        val = (weight / (2 * D)) * results / np.sqrt(idelta + 1e-9)
        np.add.at(out[offsets[l]:offsets[l + 1]], base[valid], val[valid])
    return out
