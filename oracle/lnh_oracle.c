/*
 * lnh_oracle.c — CPU restatement (plain C, scalar) of the reference's CUDA
 * kernels on the LiDAR-NeRF hot path.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product path
 * (lidar-nerf_amd/) never links, imports or calls anything in oracle/.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it restates.  The reference CUDA cannot be compiled in this
 * image (no nvcc), so this restatement is pinned by (a) the golden vectors
 * generated from the importable pure-PyTorch parts of the reference
 * (tests/golden/make_golden.py) and (b) an independent NumPy restatement
 * (oracle/grid_ref.py, oracle/raymarch_ref.py) that tests compare against.
 *
 * Build: make -C oracle   ->  oracle/_build/liblnh_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* fp16 helpers (gcc 11 / x86 has no _Float16): IEEE binary16, RNE          */
/* ------------------------------------------------------------------------ */
static uint16_t f32_to_f16_bits(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mant = x & 0x007fffffu;
    int32_t exp = (int32_t)((x >> 23) & 0xff);
    if (exp == 0xff) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u | (mant >> 13) : 0));
    }
    int32_t e = exp - 127 + 15;
    if (e >= 0x1f) return (uint16_t)(sign | 0x7c00u); /* overflow -> inf */
    if (e <= 0) {                                     /* subnormal / zero */
        if (e < -10) return (uint16_t)sign;
        mant |= 0x00800000u;
        uint32_t shift = (uint32_t)(14 - e);
        uint32_t half_m = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (half_m & 1))) half_m++;
        return (uint16_t)(sign | half_m);
    }
    uint32_t half_m = mant >> 13;
    uint32_t rem = mant & 0x1fffu;
    uint16_t h = (uint16_t)(sign | ((uint32_t)e << 10) | half_m);
    if (rem > 0x1000u || (rem == 0x1000u && (half_m & 1))) h++; /* may carry into exp: ok */
    return h;
}

static float f16_bits_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t mant = h & 0x3ffu;
    uint32_t x;
    if (exp == 0) {
        if (mant == 0) {
            x = sign;
        } else {
            int e = -1;
            do {
                e++;
                mant <<= 1;
            } while (!(mant & 0x400u));
            mant &= 0x3ffu;
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | (mant << 13);
        }
    } else if (exp == 0x1f) {
        x = sign | 0x7f800000u | (mant << 13);
    } else {
        x = sign | ((exp - 15 + 127) << 23) | (mant << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

static float round_f16(float f) { return f16_bits_to_f32(f32_to_f16_bits(f)); }

ORACLE_API uint16_t lnh_oracle_f32_to_f16(float f) { return f32_to_f16_bits(f); }
ORACLE_API float lnh_oracle_f16_to_f32(uint16_t h) { return f16_bits_to_f32(h); }

/* ------------------------------------------------------------------------ */
/* Hash grid — gridencoder/src/gridencoder.cu                               */
/* ------------------------------------------------------------------------ */

/* gridencoder.cu:53-67 fast_hash: xor_d(pos[d] * prime[d]) in uint32 */
static const uint32_t GRID_PRIMES[7] = {1u,          2654435761u, 805459861u, 3674653429u,
                                        2097192037u, 1434869437u, 2165219737u};

/* gridencoder.cu:69-93 get_grid_index */
static uint32_t grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners, uint32_t ch,
                           uint32_t hashmap_size, uint32_t resolution, const uint32_t *pos_grid) {
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pos_grid[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) {
        uint32_t h = 0;
        for (uint32_t d = 0; d < D; d++) h ^= pos_grid[d] * GRID_PRIMES[d];
        index = h;
    }
    return (index % hashmap_size) * C + ch;
}

/* gridencoder.cu:146-148: per-level scale / resolution.  exp2f is evaluated in
 * double and rounded once to float so that every implementation (this file,
 * NumPy, the HIP host launcher) agrees bit-for-bit on the level geometry. */
ORACLE_API void lnh_oracle_grid_level(uint32_t level, float S, uint32_t H, float *scale, uint32_t *resolution) {
    float e = (float)level * S;                   /* `level * S` is a float multiply in the kernel */
    float p = (float)exp2((double)e);             /* exp2f(level * S) */
    float sc = p * (float)H - 1.0f;               /* no contraction possible: product then subtract */
    *scale = sc;
    *resolution = (uint32_t)ceilf(sc) + 1u;
}

/* smoothstep (gridencoder.cu:43-51) */
static float smoothstep_f(float v) { return v * v * (3.0f - 2.0f * v); }
static float smoothstep_d(float v) { return 6 * v * (1.0f - v); }

/*
 * Corner indices + weights of one (point, level): gridencoder.cu:119-189.
 * Returns 0 when the point is out of [0,1]^D (kernel writes zeros), else 1.
 * idx[1<<D] are element indices (already * C) relative to the level base.
 */
static int grid_corners(const float *x, uint32_t D, uint32_t C, float scale, uint32_t resolution,
                        uint32_t hashmap_size, uint32_t gridtype, int align_corners, uint32_t interp,
                        uint32_t *idx, float *w, float *pos_out, float *pos_deriv, uint32_t *pos_grid_out) {
    for (uint32_t d = 0; d < D; d++)
        if (x[d] < 0 || x[d] > 1) return 0;
    float pos[8];
    uint32_t pg[8];
    for (uint32_t d = 0; d < D; d++) {
        /* `inputs[d] * scale + 0.5f` is contracted to one fma by nvcc (default -fmad=true) */
        float p = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
        float fl = floorf(p);
        pg[d] = (uint32_t)fl;
        p -= (float)pg[d];
        if (interp == 1) {
            if (pos_deriv) pos_deriv[d] = smoothstep_d(p);
            p = smoothstep_f(p);
        } else if (pos_deriv) {
            pos_deriv[d] = 1.0f;
        }
        pos[d] = p;
        if (pos_out) pos_out[d] = p;
        if (pos_grid_out) pos_grid_out[d] = pg[d];
    }
    for (uint32_t c = 0; c < (1u << D); c++) {
        float ww = 1;
        uint32_t pl[8];
        for (uint32_t d = 0; d < D; d++) {
            if ((c & (1u << d)) == 0) {
                ww *= 1 - pos[d];
                pl[d] = pg[d];
            } else {
                ww *= pos[d];
                pl[d] = pg[d] + 1;
            }
        }
        idx[c] = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pl);
        w[c] = ww;
    }
    return 1;
}

/* Debug/bit-exact contract: corner element-indices for every (level, point).
 * out_idx: [L, B, 2^D] uint32 (0xffffffff for out-of-bounds points). */
ORACLE_API void lnh_oracle_grid_indices(const float *inputs, const int32_t *offsets, uint32_t *out_idx, uint32_t B,
                                        uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                        int align_corners) {
    uint32_t nc = 1u << D;
    for (uint32_t l = 0; l < L; l++) {
        float scale;
        uint32_t res;
        lnh_oracle_grid_level(l, S, H, &scale, &res);
        uint32_t hm = (uint32_t)(offsets[l + 1] - offsets[l]);
        for (uint32_t b = 0; b < B; b++) {
            uint32_t idx[32];
            float w[32];
            uint32_t *o = out_idx + ((size_t)l * B + b) * nc;
            if (!grid_corners(inputs + (size_t)b * D, D, C, scale, res, hm, gridtype, align_corners, 0, idx, w, 0, 0,
                              0)) {
                for (uint32_t c = 0; c < nc; c++) o[c] = 0xffffffffu;
            } else {
                for (uint32_t c = 0; c < nc; c++) o[c] = idx[c];
            }
        }
    }
}

/*
 * kernel_grid forward (gridencoder.cu:95-263).  dtype: 0 = float table,
 * 1 = half table (values passed as uint16 bit patterns).  outputs [L,B,C] in
 * the table dtype; dy_dx (optional) [B, L, D, C].
 * Accumulation is in the table dtype (gridencoder.cu:173,198): for half every
 * `results[ch] += w * grid[..]` rounds to half (float fma, then narrowing).
 */
ORACLE_API void lnh_oracle_grid_forward(const float *inputs, const void *embeddings, const int32_t *offsets,
                                        void *outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                        uint32_t H, void *dy_dx, uint32_t gridtype, int align_corners, uint32_t interp,
                                        int dtype) {
    const float *ef = (const float *)embeddings;
    const uint16_t *eh = (const uint16_t *)embeddings;
    float *of = (float *)outputs;
    uint16_t *oh = (uint16_t *)outputs;
    uint32_t nc = 1u << D;
    for (uint32_t l = 0; l < L; l++) {
        float scale;
        uint32_t res;
        lnh_oracle_grid_level(l, S, H, &scale, &res);
        uint32_t hm = (uint32_t)(offsets[l + 1] - offsets[l]);
        size_t base = (size_t)(uint32_t)offsets[l] * C;
        for (uint32_t b = 0; b < B; b++) {
            uint32_t idx[32], pg[8];
            float w[32], pos[8], pderiv[8];
            size_t o = ((size_t)l * B + b) * C;
            int ok = grid_corners(inputs + (size_t)b * D, D, C, scale, res, hm, gridtype, align_corners, interp, idx, w,
                                  pos, pderiv, pg);
            if (!ok) {
                for (uint32_t ch = 0; ch < C; ch++) {
                    if (dtype == 0) of[o + ch] = 0; else oh[o + ch] = 0;
                }
                if (dy_dx) {
                    size_t dd = ((size_t)b * L + l) * D * C;
                    for (uint32_t k = 0; k < D * C; k++) {
                        if (dtype == 0) ((float *)dy_dx)[dd + k] = 0; else ((uint16_t *)dy_dx)[dd + k] = 0;
                    }
                }
                continue;
            }
            for (uint32_t ch = 0; ch < C; ch++) {
                float r = 0;
                for (uint32_t c = 0; c < nc; c++) {
                    float g = dtype == 0 ? ef[base + idx[c] + ch] : f16_bits_to_f32(eh[base + idx[c] + ch]);
                    r = fmaf(w[c], g, r);
                    if (dtype == 1) r = round_f16(r);
                }
                if (dtype == 0) of[o + ch] = r; else oh[o + ch] = f32_to_f16_bits(r);
            }
            if (dy_dx) { /* gridencoder.cu:214-262 */
                size_t dd = ((size_t)b * L + l) * D * C;
                for (uint32_t gd = 0; gd < D; gd++) {
                    for (uint32_t ch = 0; ch < C; ch++) {
                        float rg = 0;
                        for (uint32_t c = 0; c < (1u << (D - 1)); c++) {
                            float ww = scale;
                            uint32_t pl[8];
                            for (uint32_t nd = 0; nd < D - 1; nd++) {
                                uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                                if ((c & (1u << nd)) == 0) {
                                    ww *= 1 - pos[d];
                                    pl[d] = pg[d];
                                } else {
                                    ww *= pos[d];
                                    pl[d] = pg[d] + 1;
                                }
                            }
                            pl[gd] = pg[gd];
                            uint32_t il = grid_index(D, C, gridtype, align_corners, 0, hm, res, pl);
                            pl[gd] = pg[gd] + 1;
                            uint32_t ir = grid_index(D, C, gridtype, align_corners, 0, hm, res, pl);
                            float gl = dtype == 0 ? ef[base + il + ch] : f16_bits_to_f32(eh[base + il + ch]);
                            float gr = dtype == 0 ? ef[base + ir + ch] : f16_bits_to_f32(eh[base + ir + ch]);
                            float diff = gr - gl;
                            if (dtype == 1) diff = round_f16(diff); /* Half - Half -> Half */
                            rg += ww * diff * pderiv[gd];
                            if (dtype == 1) rg = round_f16(rg);
                        }
                        if (dtype == 0) ((float *)dy_dx)[dd + gd * C + ch] = rg;
                        else ((uint16_t *)dy_dx)[dd + gd * C + ch] = f32_to_f16_bits(rg);
                    }
                }
            }
        }
    }
}

/*
 * kernel_grid_backward (gridencoder.cu:265-362): scatter-add w*grad into the
 * gradient table.  The CUDA kernel uses atomics (order nondeterministic); this
 * oracle accumulates in DOUBLE per table cell and rounds once, which is the
 * order-free value every legal atomic ordering approximates.  grad: [L,B,C]
 * (dtype as table), grad_embeddings: double [rows*C] (caller zero-inits).
 * `quantize_contrib` = 1 reproduces the per-contribution half rounding
 * `(__half)(w * grad_cur[c])` at gridencoder.cu:350-351.
 */
ORACLE_API void lnh_oracle_grid_backward(const void *grad, const float *inputs, const int32_t *offsets,
                                         double *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                         float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                         int dtype, int quantize_contrib) {
    const float *gf = (const float *)grad;
    const uint16_t *gh = (const uint16_t *)grad;
    uint32_t nc = 1u << D;
    for (uint32_t l = 0; l < L; l++) {
        float scale;
        uint32_t res;
        lnh_oracle_grid_level(l, S, H, &scale, &res);
        uint32_t hm = (uint32_t)(offsets[l + 1] - offsets[l]);
        size_t base = (size_t)(uint32_t)offsets[l] * C;
        for (uint32_t b = 0; b < B; b++) {
            uint32_t idx[32];
            float w[32];
            if (!grid_corners(inputs + (size_t)b * D, D, C, scale, res, hm, gridtype, align_corners, interp, idx, w, 0,
                              0, 0))
                continue;
            size_t o = ((size_t)l * B + b) * C;
            for (uint32_t ch = 0; ch < C; ch++) {
                float g = dtype == 0 ? gf[o + ch] : f16_bits_to_f32(gh[o + ch]);
                for (uint32_t c = 0; c < nc; c++) {
                    float v = w[c] * g;
                    if (dtype == 1 && quantize_contrib) v = round_f16(v);
                    grad_embeddings[base + idx[c] + ch] += (double)v;
                }
            }
        }
    }
}

/*
 * kernel_grad_tv (gridencoder.cu:695-807): total-variation regulariser gradient of the cells visited by `inputs`.
 * Per (point, level): base cell pos_grid = floor(x * scale + 0.5) (the kernel never subtracts it: 738-742), its row
 * `index` and, per axis, the rows of the right (if cur < resolution) and left (if cur > 0) lattice neighbours;
 * results[ch] = sum of (grid[index] - grid[neighbour]), idelta[ch] = sum of their squares;
 * grad[index + ch] += (weight / (2 D)) * results[ch] * rsqrtf(idelta[ch] + 1e-9f)   (atomicAdd, 800-805).
 * Every local of the kernel is `scalar_t`: dtype = 1 (__half tables; `inputs` are __half there too, so the caller hands
 * over fp16-representable coordinates) rounds each difference, each running sum, each square and w * results to
 * fp16 (explicit __h* calls, no contraction); the atomicAdd operand is the fp16 rounding of the float product.
 * grad_out is double (order-free reference for an accumulation the GPU performs with atomics).
 */
static float rsqrt_ref(float v) { return (float)(1.0 / sqrt((double)v)); }

ORACLE_API void lnh_oracle_grad_tv(const float *inputs, const float *table, const int32_t *offsets, double *grad_out,
                                   float weight, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                   uint32_t gridtype, int align_corners, int dtype) {
#define TVR(v) (dtype == 1 ? round_f16(v) : (v))
    for (uint32_t l = 0; l < L; l++) {
        float scale;
        uint32_t res;
        lnh_oracle_grid_level(l, S, H, &scale, &res);
        uint32_t hm = (uint32_t)(offsets[l + 1] - offsets[l]);
        const float *tab = table + (size_t)(uint32_t)offsets[l] * C;
        double *g = grad_out + (size_t)(uint32_t)offsets[l] * C;
        const float w = TVR(weight / (float)(2 * D));
        for (uint32_t b = 0; b < B; b++) {
            const float *x = inputs + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) continue;
            uint32_t pg[8];
            for (uint32_t d = 0; d < D; d++) pg[d] = (uint32_t)floorf(fmaf(x[d], scale, align_corners ? 0.0f : 0.5f));
            uint32_t index = grid_index(D, C, gridtype, align_corners, 0, hm, res, pg);
            float results[8] = {0}, idelta[8] = {0};
            for (uint32_t d = 0; d < D; d++) {
                uint32_t cur = pg[d];
                for (int side = 0; side < 2; side++) {
                    if (side == 0 ? !(cur < res) : !(cur > 0)) continue;
                    pg[d] = side == 0 ? cur + 1 : cur - 1;
                    uint32_t in = grid_index(D, C, gridtype, align_corners, 0, hm, res, pg);
                    for (uint32_t ch = 0; ch < C; ch++) {
                        float gv = TVR(tab[index + ch] - tab[in + ch]);
                        results[ch] = TVR(results[ch] + gv);
                        idelta[ch] = TVR(idelta[ch] + TVR(gv * gv));
                    }
                }
                pg[d] = cur;
            }
            for (uint32_t ch = 0; ch < C; ch++) {
                float v = TVR(w * results[ch]) * rsqrt_ref(idelta[ch] + 1e-9f);
                g[index + ch] += (double)TVR(v);
            }
        }
    }
#undef TVR
}

/* kernel_input_backward (gridencoder.cu:364-390): grad_x[b,d] = sum_{l,c} grad[l,b,c]*dy_dx[b,l,d,c] (fp32 only) */
ORACLE_API void lnh_oracle_grid_input_backward(const float *grad, const float *dy_dx, float *grad_inputs, uint32_t B,
                                               uint32_t D, uint32_t C, uint32_t L) {
    for (uint32_t b = 0; b < B; b++)
        for (uint32_t d = 0; d < D; d++) {
            float r = 0;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t ch = 0; ch < C; ch++)
                    r += grad[((size_t)l * B + b) * C + ch] * dy_dx[(((size_t)b * L + l) * D + d) * C + ch];
            grad_inputs[(size_t)b * D + d] = r;
        }
}

/* ------------------------------------------------------------------------ */
/* Occupancy indexing + marching — raymarching/src/raymarching.cu            */
/* ------------------------------------------------------------------------ */

/* kernel_sph_from_ray (raymarching.cu:182-217): far intersection of o + t d with the sphere |p| = radius, expressed as
 * (theta, phi) with y up, normalised to [-1, 1]: coords = (2 theta / pi - 1, phi / pi).  Float arithmetic in the
 * reference's operation order (nvcc contracts a*b + c; the comparison is by tolerance, see the test). */
ORACLE_API void lnh_oracle_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N,
                                        float *coords) {
    const float RPI = 0.3183098861837907f;
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float A = dx * dx + dy * dy + dz * dz;
        const float Bh = ox * dx + oy * dy + oz * dz;
        const float Cc = ox * ox + oy * oy + oz * oz - radius * radius;
        const float t = (-Bh + sqrtf(Bh * Bh - A * Cc)) / A;
        const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
        const float theta = atan2f(sqrtf(x * x + z * z), y);
        const float phi = atan2f(z, x);
        coords[n * 2] = 2 * theta * RPI - 1;
        coords[n * 2 + 1] = phi * RPI;
    }
}

/* raymarching.cu:71-77 */
static uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
/* raymarching.cu:79-86 */
static uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
/* raymarching.cu:88-95 */
static uint32_t morton3d_invert(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

/* raymarching.cu:237-252 */
ORACLE_API void lnh_oracle_morton3D(const int32_t *coords, uint32_t N, int32_t *indices) {
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int32_t)morton3d((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}
/* raymarching.cu:256-279 (note: `ind >> k` is an arithmetic shift of a signed int) */
ORACLE_API void lnh_oracle_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords) {
    for (uint32_t n = 0; n < N; n++) {
        int32_t ind = indices[n];
        coords[n * 3 + 0] = (int32_t)morton3d_invert((uint32_t)(ind >> 0));
        coords[n * 3 + 1] = (int32_t)morton3d_invert((uint32_t)(ind >> 1));
        coords[n * 3 + 2] = (int32_t)morton3d_invert((uint32_t)(ind >> 2));
    }
}
/* raymarching.cu:286-319: N bytes, bit i = grid[8n+i] > thresh */
ORACLE_API void lnh_oracle_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield) {
    for (uint32_t n = 0; n < N; n++) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; i++) bits |= (grid[(size_t)n * 8 + i] > density_thresh) ? (uint8_t)(1u << i) : 0;
        bitfield[n] = bits;
    }
}

static float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
static float signf_(float x) { return copysignf(1.0f, x); }

/* raymarching.cu:51-60 */
static int mip_from_pos(float x, float y, float z, float max_cascade) {
    float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)e));
}
/* raymarching.cu:62-69 (dt * H * 0.5 evaluated in double: 0.5 is a double literal) */
static int mip_from_dt(float dt, float H, float max_cascade) {
    float mx = (float)((double)(dt * H) * 0.5);
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)e));
}

ORACLE_API void lnh_oracle_mip_levels(const float *xyz, const float *dt, uint32_t N, uint32_t C, uint32_t H,
                                      int32_t *mip_pos, int32_t *mip_dt) {
    for (uint32_t n = 0; n < N; n++) {
        mip_pos[n] = mip_from_pos(xyz[n * 3], xyz[n * 3 + 1], xyz[n * 3 + 2], (float)C);
        mip_dt[n] = mip_from_dt(dt[n], (float)H, (float)C);
    }
}

/* raymarching.cu:397-405: cell coordinate — the product is evaluated in double
 * (0.5 literal), narrowed to float by clamp(float,float,float), truncated. */
static int cell_coord(float x, float mip_rbound, uint32_t H) {
    /* x * mip_rbound + 1 is a float mul+add that nvcc contracts to one fma */
    double v = 0.5 * (double)fmaf(x, mip_rbound, 1.0f) * (double)H;
    return (int)clampf((float)v, 0.0f, (float)(H - 1));
}

typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, rH, H3, near, far, dt_min, dt_max, bound, dt_gamma;
    uint32_t C, H;
    const uint8_t *grid;
} march_ctx;

/* One marching decision at parameter t (raymarching.cu:379-439 / 464-532).
 * Returns occ; writes the clamped position, dt and the post-skip t. */
static int march_probe(const march_ctx *m, float t, float *px, float *py, float *pz, float *pdt, float *t_skip,
                       uint32_t *cell_index) {
    /* ox + t * dx: contracted to fma by nvcc's default -fmad=true */
    float x = clampf(fmaf(t, m->dx, m->ox), -m->bound, m->bound);
    float y = clampf(fmaf(t, m->dy, m->oy), -m->bound, m->bound);
    float z = clampf(fmaf(t, m->dz, m->oz), -m->bound, m->bound);
    float dt = clampf(t * m->dt_gamma, m->dt_min, m->dt_max);
    int lp = mip_from_pos(x, y, z, (float)m->C), ld = mip_from_dt(dt, (float)m->H, (float)m->C);
    int level = lp > ld ? lp : ld;
    float mip_bound = fminf(scalbnf(1.0f, level), m->bound);
    float mip_rbound = 1 / mip_bound;
    int nx = cell_coord(x, mip_rbound, m->H), ny = cell_coord(y, mip_rbound, m->H), nz = cell_coord(z, mip_rbound, m->H);
    /* `level * H3` is float (H3 is float) + uint32 morton -> float sum -> uint32 (raymarching.cu:407) */
    uint32_t index = (uint32_t)((float)level * m->H3 + (float)morton3d((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    int occ = (m->grid[index / 8] & (1u << (index % 8))) != 0;
    *px = x; *py = y; *pz = z; *pdt = dt;
    if (cell_index) *cell_index = index;
    if (!occ) {
        float tx = (((nx + 0.5f + 0.5f * signf_(m->dx)) * m->rH * 2 - 1) * mip_bound - x) * m->rdx;
        float ty = (((ny + 0.5f + 0.5f * signf_(m->dy)) * m->rH * 2 - 1) * mip_bound - y) * m->rdy;
        float tz = (((nz + 0.5f + 0.5f * signf_(m->dz)) * m->rH * 2 - 1) * mip_bound - z) * m->rdz;
        float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        do {
            t += clampf(t * m->dt_gamma, m->dt_min, m->dt_max);
        } while (t < tt);
        *t_skip = t;
    }
    return occ;
}

/*
 * kernel_march_rays_train (raymarching.cu:331-534).  Deterministic allocation:
 * rays are assigned offsets in ray-id order (the CUDA atomics give an arbitrary
 * order; tests compare keyed by ray id).  rays: [N,3] = (id, offset, count).
 * counter: [2] (total points, total rays) accumulated.
 */
ORACLE_API void lnh_oracle_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound,
                                            float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                            uint32_t M, const float *nears, const float *fars, float *xyzs, float *dirs,
                                            float *deltas, int32_t *rays, int32_t *counter, const float *noises) {
    const float SQRT3 = 1.7320508075688772f;
    for (uint32_t n = 0; n < N; n++) {
        march_ctx m;
        m.ox = rays_o[n * 3]; m.oy = rays_o[n * 3 + 1]; m.oz = rays_o[n * 3 + 2];
        m.dx = rays_d[n * 3]; m.dy = rays_d[n * 3 + 1]; m.dz = rays_d[n * 3 + 2];
        m.rdx = 1 / m.dx; m.rdy = 1 / m.dy; m.rdz = 1 / m.dz;
        m.rH = 1 / (float)H; m.H3 = (float)(H * H * H);
        m.near = nears[n]; m.far = fars[n];
        m.dt_min = 2 * SQRT3 / max_steps;
        m.dt_max = 2 * SQRT3 * (float)(1 << (C - 1)) / H;
        m.bound = bound; m.dt_gamma = dt_gamma; m.C = C; m.H = H; m.grid = grid;
        float t0 = m.near;
        t0 += clampf(t0 * dt_gamma, m.dt_min, m.dt_max) * noises[n];
        float t = t0;
        uint32_t num_steps = 0;
        float x, y, z, dt, ts;
        while (t < m.far && num_steps < max_steps) {
            if (march_probe(&m, t, &x, &y, &z, &dt, &ts, 0)) {
                num_steps++;
                t += dt;
            } else {
                t = ts;
            }
        }
        uint32_t point_index = (uint32_t)counter[0];
        counter[0] += (int32_t)num_steps;
        uint32_t ray_index = (uint32_t)counter[1];
        counter[1] += 1;
        rays[ray_index * 3] = (int32_t)n;
        rays[ray_index * 3 + 1] = (int32_t)point_index;
        rays[ray_index * 3 + 2] = (int32_t)num_steps;
        if (num_steps == 0) continue;
        if (point_index + num_steps > M) continue;
        float *px = xyzs + (size_t)point_index * 3, *pd = dirs + (size_t)point_index * 3,
              *pl = deltas + (size_t)point_index * 2;
        t = t0;
        uint32_t step = 0;
        float last_t = t;
        while (t < m.far && step < num_steps) {
            if (march_probe(&m, t, &x, &y, &z, &dt, &ts, 0)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
                t += dt;
                pl[0] = dt;
                pl[1] = t - last_t;
                last_t = t;
                px += 3; pd += 3; pl += 2;
                step++;
            } else {
                t = ts;
            }
        }
    }
}

/* Cell index / occupancy bit lookup for arbitrary points (raymarching.cu:386-408) */
ORACLE_API void lnh_oracle_occupancy_lookup(const float *xyz, const float *dt, const uint8_t *grid, float bound,
                                            uint32_t N, uint32_t C, uint32_t H, uint32_t *cell_index, uint8_t *occ) {
    for (uint32_t n = 0; n < N; n++) {
        float x = clampf(xyz[n * 3], -bound, bound), y = clampf(xyz[n * 3 + 1], -bound, bound),
              z = clampf(xyz[n * 3 + 2], -bound, bound);
        int lp = mip_from_pos(x, y, z, (float)C), ld = mip_from_dt(dt[n], (float)H, (float)C);
        int level = lp > ld ? lp : ld;
        float mip_bound = fminf(scalbnf(1.0f, level), bound);
        float mip_rbound = 1 / mip_bound;
        int nx = cell_coord(x, mip_rbound, H), ny = cell_coord(y, mip_rbound, H), nz = cell_coord(z, mip_rbound, H);
        float H3 = (float)(H * H * H);
        uint32_t index = (uint32_t)((float)level * H3 + (float)morton3d((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
        cell_index[n] = index;
        occ[n] = (grid[index / 8] & (1u << (index % 8))) != 0;
    }
}

/* raymarching.cu:104-157 near_far_from_aabb */
ORACLE_API void lnh_oracle_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N,
                                              float min_near, float *nears, float *fars) {
    const float FMAX = 3.402823466e+38f;
    for (uint32_t n = 0; n < N; n++) {
        float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        float rdx = 1 / rays_d[n * 3], rdy = 1 / rays_d[n * 3 + 1], rdz = 1 / rays_d[n * 3 + 2];
        float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx, tmp;
        if (near > far) { tmp = near; near = far; far = tmp; }
        float ny = (aabb[1] - oy) * rdy, fy = (aabb[4] - oy) * rdy;
        if (ny > fy) { tmp = ny; ny = fy; fy = tmp; }
        if (near > fy || ny > far) { nears[n] = fars[n] = FMAX; continue; }
        if (ny > near) near = ny;
        if (fy < far) far = fy;
        float nz = (aabb[2] - oz) * rdz, fz = (aabb[5] - oz) * rdz;
        if (nz > fz) { tmp = nz; nz = fz; fz = tmp; }
        if (near > fz || nz > far) { nears[n] = fars[n] = FMAX; continue; }
        if (nz > near) near = nz;
        if (fz < far) far = fz;
        if (near < min_near) near = min_near;
        nears[n] = near;
        fars[n] = far;
    }
}

/* raymarching.cu:577-655 composite_rays_train forward (3 channels, early stop) */
ORACLE_API void lnh_oracle_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                                        const int32_t *rays, uint32_t M, uint32_t N, float T_thresh,
                                                        float *weights_sum, float *depth, float *image) {
    for (uint32_t n = 0; n < N; n++) {
        uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) {
            weights_sum[index] = 0; depth[index] = 0;
            image[index * 3] = image[index * 3 + 1] = image[index * 3 + 2] = 0;
            continue;
        }
        const float *s = sigmas + offset, *c = rgbs + (size_t)offset * 3, *dl = deltas + (size_t)offset * 2;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            float alpha = 1.0f - expf(-s[0] * dl[0]);
            float weight = alpha * T;
            r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
            t += dl[1];
            d += weight * t;
            ws += weight;
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
            s++; c += 3; dl += 2;
        }
        weights_sum[index] = ws; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

/* raymarching.cu:690-772 composite_rays_train backward (no depth gradient) */
ORACLE_API void lnh_oracle_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                                         const float *sigmas, const float *rgbs, const float *deltas,
                                                         const int32_t *rays, const float *weights_sum,
                                                         const float *image, uint32_t M, uint32_t N, float T_thresh,
                                                         float *grad_sigmas, float *grad_rgbs) {
    for (uint32_t n = 0; n < N; n++) {
        uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) continue;
        const float *gi = grad_image + (size_t)index * 3;
        float gws = grad_weights_sum[index], ws_final = weights_sum[index];
        float rf = image[index * 3], gf = image[index * 3 + 1], bf = image[index * 3 + 2];
        const float *s = sigmas + offset, *c = rgbs + (size_t)offset * 3, *dl = deltas + (size_t)offset * 2;
        float *gs = grad_sigmas + offset, *gc = grad_rgbs + (size_t)offset * 3;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            float alpha = 1.0f - expf(-s[0] * dl[0]);
            float weight = alpha * T;
            r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
            ws += weight;
            T *= 1.0f - alpha;
            gc[0] = gi[0] * weight; gc[1] = gi[1] * weight; gc[2] = gi[2] * weight;
            gs[0] = dl[0] * (gi[0] * (T * c[0] - (rf - r)) + gi[1] * (T * c[1] - (gf - g)) +
                             gi[2] * (T * c[2] - (bf - b)) + gws * (1 - ws_final));
            if (T < T_thresh) break;
            s++; c += 3; dl += 2; gs++; gc += 3;
        }
    }
}

/* ---- chamfer nearest neighbour: extern/chamfer3D/chamfer3D.cu:9-138 (semantics: squared distance, first minimal
 * index; x*x + y*y + z*z with the two contractions nvcc applies by default). */
ORACLE_API void lnh_oracle_chamfer_nn(const float *xyz1, uint32_t n, const float *xyz2, uint32_t m, float *dist,
                                      int32_t *idx) {
    for (uint32_t j = 0; j < n; j++) {
        const float x1 = xyz1[j * 3], y1 = xyz1[j * 3 + 1], z1 = xyz1[j * 3 + 2];
        float best = 0.0f;
        int32_t best_i = 0;
        for (uint32_t k = 0; k < m; k++) {
            const float dx = xyz2[k * 3] - x1, dy = xyz2[k * 3 + 1] - y1, dz = xyz2[k * 3 + 2] - z1;
            const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (k == 0 || d < best) { best = d; best_i = (int32_t)k; }
        }
        dist[j] = best;
        idx[j] = best_i;
    }
}

/* ---- inference marching / compositing: raymarching.cu:808-928, 966-1053 */
ORACLE_API void lnh_oracle_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                                      const float *rays_o, const float *rays_d, float bound, float dt_gamma,
                                      uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid,
                                      const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                                      const float *noises) {
    const float SQRT3 = 1.7320508075688772f;
    for (uint32_t n = 0; n < n_alive; n++) {
        const uint32_t index = (uint32_t)rays_alive[n];
        march_ctx m;
        m.ox = rays_o[index * 3]; m.oy = rays_o[index * 3 + 1]; m.oz = rays_o[index * 3 + 2];
        m.dx = rays_d[index * 3]; m.dy = rays_d[index * 3 + 1]; m.dz = rays_d[index * 3 + 2];
        m.rdx = 1 / m.dx; m.rdy = 1 / m.dy; m.rdz = 1 / m.dz;
        m.rH = 1 / (float)H; m.H3 = (float)(H * H * H);
        m.near = nears[index]; m.far = fars[index];
        m.dt_min = 2 * SQRT3 / max_steps;
        m.dt_max = 2 * SQRT3 * (float)(1 << (C - 1)) / H;
        m.bound = bound; m.dt_gamma = dt_gamma; m.C = C; m.H = H; m.grid = grid;
        float t = rays_t[index];
        t += clampf(t * dt_gamma, m.dt_min, m.dt_max) * noises[n];
        float *px = xyzs + (size_t)n * n_step * 3, *pd = dirs + (size_t)n * n_step * 3,
              *pl = deltas + (size_t)n * n_step * 2;
        float last_t = t, x, y, z, dt, ts;
        uint32_t step = 0;
        while (t < m.far && step < n_step) {
            if (march_probe(&m, t, &x, &y, &z, &dt, &ts, 0)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
                t += dt;
                pl[0] = dt;
                pl[1] = t - last_t;
                last_t = t;
                px += 3; pd += 3; pl += 2;
                step++;
            } else {
                t = ts;
            }
        }
    }
}

ORACLE_API void lnh_oracle_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                                          float *rays_t, const float *sigmas, const float *rgbs, const float *deltas,
                                          float *weights_sum, float *depth, float *image) {
    for (uint32_t n = 0; n < n_alive; n++) {
        const uint32_t index = (uint32_t)rays_alive[n];
        const float *s = sigmas + (size_t)n * n_step, *c = rgbs + (size_t)n * n_step * 3,
                    *dl = deltas + (size_t)n * n_step * 2;
        float t = rays_t[index], ws = weights_sum[index], d = depth[index];
        float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
        uint32_t step = 0;
        while (step < n_step) {
            if (dl[0] == 0) break;
            const float alpha = 1.0f - expf(-s[0] * dl[0]);
            const float T = 1 - ws;
            const float weight = alpha * T;
            ws += weight;
            t += dl[1];
            d += weight * t;
            r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
            if (T < T_thresh) break;
            s++; c += 3; dl += 2;
            step++;
        }
        if (step < n_step) rays_alive[n] = -1;
        else rays_t[index] = t;
        weights_sum[index] = ws; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}
