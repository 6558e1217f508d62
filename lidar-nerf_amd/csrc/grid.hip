// grid.hip — multi-resolution hash / tiled grid encoder for gfx950 (MI355X).
//
// Semantics follow the reference kernels (lidarnerf/gridencoder/src/gridencoder.cu:53-390, 695-807); the
// implementation is organised for CDNA4:
//   * the level geometry (scale, resolution, dense/hash decision, modulo strategy) is resolved ONCE on the host and
//     handed to the kernel in kernarg SGPRs, so the per-corner index math is a handful of VALU ops with no
//     per-thread loop-carried stride test and no integer division on the hot levels;
//   * one 2D launch, blockIdx.y = level (dispatch order is x-fastest, so the chip works level-major and every XCD's
//     4 MiB L2 holds the ~2 MiB fp16 table of the level in flight);
//   * 64 consecutive lanes = 64 consecutive sample points (consecutive samples along a LiDAR ray): coarse-level
//     corner gathers of a wave hit a few cache lines; the [L,B,C] output store is a fully coalesced 256 B / wave;
//   * backward: packed `global_atomic_pk_add_f16` (fp16 tables) / `global_atomic_add_f32`, preceded by a
//     wave-level run-merge: lanes whose sample falls in the same cell as their lower neighbour (the common case on
//     coarse levels, where one cell spans many consecutive samples of a ray) are summed with a segmented shuffle
//     scan and only the run tail issues the atomic.
#include "common.h"

typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));
#include <type_traits>

#include <algorithm>
#include <cmath>

namespace {

struct LevelParams {
    float scale;
    uint32_t resolution;
    uint32_t hashmap_size;
    uint32_t offset;  // rows before this level
    uint32_t flags;   // bits 0-3: dims accumulated densely; bit4 hash; bit5 pow2 table; bit6 no wrap needed
};
struct GridMeta {
    LevelParams lv[LNH_MAX_LEVELS];
};
struct RowMap {
    uint32_t T_cur, T_tot, slot_off, B_all;
};
enum { LV_HASH = 16, LV_POW2 = 32, LV_NOWRAP = 64 };

// gridencoder.cu:55-57 (spatial-hash primes); folded to immediates after unrolling
__host__ __device__ constexpr uint32_t prime_of(int d) {
    return d == 0 ? 1u : d == 1 ? 2654435761u : d == 2 ? 805459861u : d == 3 ? 3674653429u
         : d == 4 ? 2097192037u : d == 5 ? 1434869437u : 2165219737u;
}

// Host: gridencoder.cu:146-148 + the stride loop of get_grid_index (69-84) evaluated once per level.
// exp2f(level*S) is evaluated in double and rounded once (same convention as the oracle).
int build_meta(GridMeta &m, const int32_t *offsets, uint32_t D, uint32_t L, float S, uint32_t H, uint32_t gridtype,
               bool align) {
    for (uint32_t l = 0; l < L; l++) {
        LevelParams &p = m.lv[l];
        float e = (float)l * S;
        float pw = (float)exp2((double)e);
        p.scale = pw * (float)H - 1.0f;
        p.resolution = (uint32_t)ceilf(p.scale) + 1u;
        int64_t hm = (int64_t)offsets[l + 1] - (int64_t)offsets[l];
        if (hm <= 0 || offsets[l] < 0) return -1;
        p.hashmap_size = (uint32_t)hm;
        p.offset = (uint32_t)offsets[l];
        uint32_t R = align ? p.resolution : p.resolution + 1;
        uint32_t stride = 1, nd = 0;
        uint64_t exact = 1;
        for (uint32_t d = 0; d < D && stride <= p.hashmap_size; d++) {
            stride *= R;  // uint32 wrap-around exactly like the device code it replaces
            exact *= R;
            nd++;
        }
        uint32_t f = nd;
        bool hash = (gridtype == 0 && stride > p.hashmap_size);
        if (hash) f |= LV_HASH;
        if ((p.hashmap_size & (p.hashmap_size - 1)) == 0) f |= LV_POW2;
        if (!hash && nd == D && exact <= p.hashmap_size) f |= LV_NOWRAP;
        p.flags = f;
    }
    return 0;
}

template <int D>
struct Cell {
    uint32_t term[D][2];  // per-dimension contribution to the row index for offset 0 / +1
    float frac[D];
    float deriv[D];
};

// Position inside the level lattice (gridencoder.cu:150-167).  Returns false for out-of-range points.
template <int D>
__device__ __forceinline__ bool locate(const float (&x)[D], const LevelParams &lv, bool align, uint32_t interp,
                                       Cell<D> &c) {
    bool ok = true;
#pragma unroll
    for (int d = 0; d < D; d++) ok = ok && !(x[d] < 0.0f || x[d] > 1.0f);
    if (!ok) return false;
    const uint32_t R = align ? lv.resolution : lv.resolution + 1;
    const bool hash = lv.flags & LV_HASH;
    const uint32_t nd = lv.flags & 15u;
    uint32_t stride = 1;
#pragma unroll
    for (int d = 0; d < D; d++) {
        float p = fmaf(x[d], lv.scale, align ? 0.0f : 0.5f);
        float fl = floorf(p);
        uint32_t g = (uint32_t)fl;
        p -= (float)g;
        if (interp == 1) {
            c.deriv[d] = 6.0f * p * (1.0f - p);
            p = p * p * (3.0f - 2.0f * p);
        } else {
            c.deriv[d] = 1.0f;
        }
        c.frac[d] = p;
        if (hash) {
            uint32_t t0 = g * prime_of(d);
            c.term[d][0] = t0;
            c.term[d][1] = t0 + prime_of(d);
        } else if ((uint32_t)d < nd) {
            uint32_t t0 = g * stride;
            c.term[d][0] = t0;
            c.term[d][1] = t0 + stride;
            stride *= R;
        } else {  // tiled grid whose stride already exceeded the table: dimension ignored (gridencoder.cu:81)
            c.term[d][0] = 0;
            c.term[d][1] = 0;
        }
    }
    return true;
}

template <int D>
__device__ __forceinline__ uint32_t corner_row(const Cell<D> &c, const LevelParams &lv, uint32_t corner) {
    uint32_t idx = 0;
    if (lv.flags & LV_HASH) {
#pragma unroll
        for (int d = 0; d < D; d++) idx ^= c.term[d][(corner >> d) & 1];
    } else {
#pragma unroll
        for (int d = 0; d < D; d++) idx += c.term[d][(corner >> d) & 1];
    }
    if (lv.flags & LV_POW2) idx &= lv.hashmap_size - 1;
    else if (!(lv.flags & LV_NOWRAP)) idx %= lv.hashmap_size;
    return idx;
}

template <int D>
__device__ __forceinline__ float corner_weight(const Cell<D> &c, uint32_t corner) {
    float w = 1.0f;
#pragma unroll
    for (int d = 0; d < D; d++) w *= ((corner >> d) & 1) ? c.frac[d] : (1.0f - c.frac[d]);
    return w;
}

template <typename T, int C>
struct Vec {
    T v[C];
};
template <typename T, int C>
__device__ __forceinline__ Vec<T, C> load_vec(const T *p) {
    Vec<T, C> r;
    if constexpr (sizeof(T) * C == 4) {
        uint32_t raw = *reinterpret_cast<const uint32_t *>(p);
        __builtin_memcpy(&r, &raw, 4);
    } else if constexpr (sizeof(T) * C == 8) {
        uint2 raw = *reinterpret_cast<const uint2 *>(p);
        __builtin_memcpy(&r, &raw, 8);
    } else if constexpr (sizeof(T) * C == 16) {
        uint4 raw = *reinterpret_cast<const uint4 *>(p);
        __builtin_memcpy(&r, &raw, 16);
    } else if constexpr (sizeof(T) * C == 32) {
        uint4 a = reinterpret_cast<const uint4 *>(p)[0], b = reinterpret_cast<const uint4 *>(p)[1];
        __builtin_memcpy(&r, &a, 16);
        __builtin_memcpy(reinterpret_cast<char *>(&r) + 16, &b, 16);
    } else {
#pragma unroll
        for (int i = 0; i < C; i++) r.v[i] = p[i];
    }
    return r;
}
template <typename T, int C>
__device__ __forceinline__ void store_vec(T *p, const Vec<T, C> &r) {
    if constexpr (sizeof(T) * C == 4) {
        uint32_t raw;
        __builtin_memcpy(&raw, &r, 4);
        *reinterpret_cast<uint32_t *>(p) = raw;
    } else if constexpr (sizeof(T) * C == 8) {
        uint2 raw;
        __builtin_memcpy(&raw, &r, 8);
        *reinterpret_cast<uint2 *>(p) = raw;
    } else if constexpr (sizeof(T) * C == 16) {
        uint4 raw;
        __builtin_memcpy(&raw, &r, 16);
        *reinterpret_cast<uint4 *>(p) = raw;
    } else {
#pragma unroll
        for (int i = 0; i < C; i++) p[i] = r.v[i];
    }
}

// ------------------------------------------------------------------------------------------------ forward
template <typename T, int D, int C, bool DYDX>
__global__ void __launch_bounds__(256)
k_grid_forward(const float *__restrict__ inputs, const T *__restrict__ table, T *__restrict__ outputs,
               T *__restrict__ dy_dx, uint32_t B, uint32_t L, GridMeta meta, uint32_t align_rt, uint32_t interp_rt,
               RowMap map) {
    const uint32_t b0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (b0 >= B) return;
    {  // (tools/probe_variants.py "fusion" turns this block into a loop over the levels: the encode -> MLP fusion probe)
    const uint32_t level = blockIdx.y;
    const LevelParams lv_rt = meta.lv[level];
    // optional row map: launch index b0 = r*T_cur + j addresses row r*T_tot + slot_off + j of buffers holding
    // B_all rows (coarse and fine samples of a ray side by side); identity when T_cur == 0
    const uint32_t b = map.T_cur ? (b0 / map.T_cur) * map.T_tot + map.slot_off + b0 % map.T_cur : b0;
    const uint32_t Bs = map.T_cur ? map.B_all : B;
    float x[D];
#pragma unroll
    for (int d = 0; d < D; d++) x[d] = inputs[(size_t)b * D + d];
    const T *tab = table + (size_t)lv_rt.offset * C;
    T *out = outputs + ((size_t)level * Bs + b) * C;
    // The level class is workgroup-uniform.  The two classes that make up the usual configuration (hashed power-of-two
    // level / dense level indexed in all D dimensions, both with linear interpolation and align_corners = false) get
    // their own straight-line copy of the body with the flags as compile-time constants; everything else takes the
    // generic copy with run-time branches.
    auto body = [&](auto mode_c) {
    constexpr int MODE = decltype(mode_c)::value;
    LevelParams lv = lv_rt;
    if constexpr (MODE == 1) lv.flags = LV_HASH | LV_POW2;
    if constexpr (MODE == 2) lv.flags = LV_NOWRAP | (uint32_t)D;
    const uint32_t align = MODE ? 0u : align_rt, interp = MODE ? 0u : interp_rt;
    Cell<D> cell;
    Vec<T, C> res;
#pragma unroll
    for (int ch = 0; ch < C; ch++) res.v[ch] = (T)0.0f;
    const bool ok = locate<D>(x, lv, align != 0, interp, cell);
    if (ok) {
        // issue all 2^D gathers before consuming any of them
        Vec<T, C> g[1 << D];
        if constexpr (D == 3 && C == 2) {
            // The x-neighbours of a cell edge lie close together in the table: rows r0, r0 + 1 when the x term is dense,
            // and r1 = r0 ^ (2^(t+1) - 1) on a hashed level (the x term of the hash is x itself; t = trailing ones of the
            // x coordinate).  Measured on MI355X (A/B builds over tools/bench_grid.py): a wave's gather costs per DISTINCT
            // line per INSTRUCTION — a second instruction touching the lines of the first still costs a third of a miss
            // (tag look-up), its width (4 / 8 / 16 bytes per lane) costs nothing — so each (y,z) corner pair is fetched
            // with one wide load wherever its two rows share an aligned block.
            const bool hash = lv.flags & LV_HASH;
            const bool x_dense = !hash && (lv.flags & 15u) >= 1 && (lv.flags & LV_NOWRAP);
            const bool hash_pair = hash && (lv.flags & LV_POW2);   // level-uniform
            const bool x_odd = cell.term[0][0] & 1u;               // per lane
            // All loads first, every use after the loop, and no per-lane branch around a load: a consumer (or the end of
            // a divergent region) pins an s_waitcnt and serialises the four (y,z) iterations into dependent round trips.
            // Lanes that need no second load read row 0 instead (one broadcast line): selecting the address keeps the
            // load unconditional.
            //
            // Hashed level, 4-byte rows: the aligned block of FOUR rows (16 bytes) around r0 also holds r1 for t <= 1,
            // i.e. for 3 lanes in 4; the others fetch r1 on its own (540 -> 505 us per 3.41 M points against the 2-row
            // blocks below; without the second load at all: 445 us).
            if constexpr (sizeof(T) * C == 4) {
            if (hash_pair) {
                Vec<T, 8> blk[4];
                Vec<T, C> solo[4];
                uint32_t r0s[4], r1s[4];
                const uint32_t xm = cell.term[0][0] ^ cell.term[0][1];   // 2^(t+1) - 1 (before the table mask)
                const bool far = (xm & (lv.hashmap_size - 1)) > 3u;
#pragma unroll
                for (uint32_t yz = 0; yz < 4; yz++) {
                    const uint32_t c0 = yz << 1;
                    const uint32_t r0 = corner_row<D>(cell, lv, c0);
                    const uint32_t r1 = corner_row<D>(cell, lv, c0 | 1u);
                    r0s[yz] = r0;
                    r1s[yz] = r1;
                    blk[yz] = load_vec<T, 8>(tab + (size_t)(r0 & ~3u) * C);
                    solo[yz] = load_vec<T, C>(tab + (size_t)(far ? r1 : 0u) * C);  // (probe anchor: tools/probe_variants.py "gather")
                }
#pragma unroll
                for (uint32_t yz = 0; yz < 4; yz++) {
                    const uint32_t c0 = yz << 1, c1 = c0 | 1u;
                    uint32_t w[4], so;
                    __builtin_memcpy(w, &blk[yz], 16);
                    __builtin_memcpy(&so, &solo[yz], 4);
                    const uint32_t i0 = r0s[yz] & 3u, i1 = r1s[yz] & 3u;
                    const uint32_t a0 = (i0 & 2u) ? ((i0 & 1u) ? w[3] : w[2]) : ((i0 & 1u) ? w[1] : w[0]);
                    const uint32_t a1 = (i1 & 2u) ? ((i1 & 1u) ? w[3] : w[2]) : ((i1 & 1u) ? w[1] : w[0]);
                    const uint32_t b1 = far ? so : a1;
                    __builtin_memcpy(&g[c0], &a0, 4);
                    __builtin_memcpy(&g[c1], &b1, 4);
                }
            }
            }
            if (!(sizeof(T) * C == 4 && hash_pair)) {
            //   pair[yz]: the aligned row pair holding corner c0 (hashed levels) / rows r0, r0+1 (dense x)
            //   solo[yz]: corner c1 on its own, fetched only where it is not the sibling of c0 (hashed level, odd x)
            Vec<T, 4> pair[4];
            Vec<T, C> solo[4];
            uint32_t r0s[4];
#pragma unroll
            for (uint32_t yz = 0; yz < 4; yz++) {
                const uint32_t c0 = yz << 1, c1 = c0 | 1u;
                const uint32_t r0 = corner_row<D>(cell, lv, c0);
                r0s[yz] = r0;
                if (x_dense || hash_pair) {
                    pair[yz] = load_vec<T, 4>(tab + (size_t)(x_dense ? r0 : (r0 & ~1u)) * C);
                    // lanes whose c1 IS the sibling read row 0 instead (one broadcast line): selecting the address
                    // keeps the load unconditional — an exec-masked load would be waited for at the end of its branch
                    if (hash_pair)
                        solo[yz] = load_vec<T, C>(tab + (size_t)(x_odd ? corner_row<D>(cell, lv, c1) : 0u) * C);
                } else {
                    g[c0] = load_vec<T, C>(tab + (size_t)r0 * C);
                    g[c1] = load_vec<T, C>(tab + (size_t)corner_row<D>(cell, lv, c1) * C);
                }
            }
            if (x_dense || hash_pair) {
#pragma unroll
                for (uint32_t yz = 0; yz < 4; yz++) {
                    const uint32_t c0 = yz << 1, c1 = c0 | 1u;
                    const bool swap = hash_pair && (r0s[yz] & 1u);  // hashed, r0 odd: its sibling r0 ^ 1 is the lower row
                    const bool own = hash_pair && x_odd;
                    g[c0].v[0] = swap ? pair[yz].v[2] : pair[yz].v[0];
                    g[c0].v[1] = swap ? pair[yz].v[3] : pair[yz].v[1];
                    g[c1].v[0] = own ? solo[yz].v[0] : (swap ? pair[yz].v[0] : pair[yz].v[2]);
                    g[c1].v[1] = own ? solo[yz].v[1] : (swap ? pair[yz].v[1] : pair[yz].v[3]);
                }
            }
            }
        } else {
#pragma unroll
            for (uint32_t c = 0; c < (1u << D); c++)
                g[c] = load_vec<T, C>(tab + (size_t)corner_row<D>(cell, lv, c) * C);
        }
#pragma unroll
        for (uint32_t c = 0; c < (1u << D); c++) {
            const float w = corner_weight<D>(cell, c);
            // accumulate in the table type, one rounding per corner (gridencoder.cu:173,198)
#pragma unroll
            for (int ch = 0; ch < C; ch++) {
                float acc = fmaf(w, (float)g[c].v[ch], (float)res.v[ch]);
                // keep the fp32 rounding of the fma visible: without this hipcc fuses fma+narrowing into
                // v_fma_mixlo_f16 (ONE rounding), while the reference rounds to fp32 and then to fp16
                if constexpr (sizeof(T) == 2) asm volatile("" : "+v"(acc));
                res.v[ch] = (T)acc;
            }
        }
    }
    store_vec<T, C>(out, res);
    if constexpr (DYDX) {
        // gridencoder.cu:214-262: d out / d x for every input dimension, layout [B, L, D, C]
        T *dd = dy_dx + (((size_t)b * L + level) * D) * C;
#pragma unroll
        for (int gd = 0; gd < D; gd++) {
            Vec<T, C> rg;
#pragma unroll
            for (int ch = 0; ch < C; ch++) rg.v[ch] = (T)0.0f;
            if (ok) {
#pragma unroll
                for (uint32_t k = 0; k < (1u << (D - 1)); k++) {
                    float w = lv.scale;
                    uint32_t corner = 0;
#pragma unroll
                    for (int nd = 0; nd < D - 1; nd++) {
                        const int d = (nd >= gd) ? nd + 1 : nd;
                        if ((k >> nd) & 1) {
                            w *= cell.frac[d];
                            corner |= 1u << d;
                        } else {
                            w *= 1.0f - cell.frac[d];
                        }
                    }
                    Vec<T, C> gl = load_vec<T, C>(tab + (size_t)corner_row<D>(cell, lv, corner) * C);
                    Vec<T, C> gr = load_vec<T, C>(tab + (size_t)corner_row<D>(cell, lv, corner | (1u << gd)) * C);
#pragma unroll
                    for (int ch = 0; ch < C; ch++) {
                        T diff = (T)(gr.v[ch] - gl.v[ch]);
                        rg.v[ch] = (T)((float)rg.v[ch] + w * (float)diff * cell.deriv[gd]);
                    }
                }
            }
            store_vec<T, C>(dd + gd * C, rg);
        }
    }
    };
    const bool plain = align_rt == 0 && interp_rt == 0;
    if (plain && (lv_rt.flags & (LV_HASH | LV_POW2)) == (LV_HASH | LV_POW2)) body(std::integral_constant<int, 1>{});
    else if (plain && !(lv_rt.flags & LV_HASH) && (lv_rt.flags & LV_NOWRAP) && (lv_rt.flags & 15u) == (uint32_t)D)
        body(std::integral_constant<int, 2>{});
    else body(std::integral_constant<int, 0>{});
    }  // level
}

// Debug kernel for the bit-exact index contract.
template <int D>
__global__ void __launch_bounds__(256)
k_grid_indices(const float *__restrict__ inputs, uint32_t *__restrict__ out, uint32_t B, uint32_t C, GridMeta meta,
               uint32_t align) {
    const uint32_t level = blockIdx.y;
    const LevelParams lv = meta.lv[level];
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float x[D];
#pragma unroll
    for (int d = 0; d < D; d++) x[d] = inputs[(size_t)b * D + d];
    Cell<D> cell;
    const bool ok = locate<D>(x, lv, align != 0, 0, cell);
    uint32_t *o = out + ((size_t)level * B + b) * (1u << D);
#pragma unroll
    for (uint32_t c = 0; c < (1u << D); c++) o[c] = ok ? corner_row<D>(cell, lv, c) * C : 0xffffffffu;
}

// ------------------------------------------------------------------------------------------------ backward
__device__ __forceinline__ void atomic_add_pair(half_t *p, float a, float b) {
    half2_t v = {(half_t)a, (half_t)b};
    __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) half2_t *)p, v);
}
__device__ __forceinline__ void atomic_add_pair(float *p, float a, float b) {
    unsafeAtomicAdd(p, a);
    unsafeAtomicAdd(p + 1, b);
}
__device__ __forceinline__ void atomic_add_one(float *p, float a) { unsafeAtomicAdd(p, a); }

// One thread = one (point, level, channel pair).  NC = channels per thread (1 or 2).
// MERGE: wave-level run merge of equal cells before the atomics (see file header).
template <typename T, int D, int C, int NC, bool MERGE>
__global__ void __launch_bounds__(256)
k_grid_backward(const T *__restrict__ grad, const float *__restrict__ inputs, T *__restrict__ grad_table, uint32_t B,
                GridMeta meta, uint32_t align, uint32_t interp) {
    constexpr int NP = C / NC;  // channel groups per point
    const uint32_t level = blockIdx.y;
    const LevelParams lv = meta.lv[level];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t / NP;
    const uint32_t ch = (t % NP) * NC;
    const bool in_range = b < B;
    float x[D];
#pragma unroll
    for (int d = 0; d < D; d++) x[d] = in_range ? inputs[(size_t)b * D + d] : -1.0f;
    Cell<D> cell;
    const bool ok = in_range && locate<D>(x, lv, align != 0, interp, cell);
    float g[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) g[c] = ok ? (float)grad[((size_t)level * B + b) * C + ch + c] : 0.0f;
    T *gt = grad_table + (size_t)lv.offset * C + ch;

    if (MERGE && NP == 1) {
        // Lanes are consecutive samples.  If this lane's base cell equals the previous lane's, both touch the same
        // 2^D table rows: pre-reduce w*g over such runs with a segmented inclusive scan; the LAST lane of each run
        // issues the atomics.  Cell identity = all per-dimension base terms equal (exact, hash or dense).
        const int lane = threadIdx.x & 63;
        uint32_t key[D];
#pragma unroll
        for (int d = 0; d < D; d++) key[d] = ok ? cell.term[d][0] : 0xffffffffu - lane;  // invalid lanes never match
        // NB: every shuffle must execute with all 64 lanes active — no short-circuit around it
        bool same_prev = true;
#pragma unroll
        for (int d = 0; d < D; d++) {
            const uint32_t up = __shfl_up(key[d], 1, 64);
            same_prev &= (up == key[d]);
        }
        same_prev &= lane > 0;
        const unsigned long long heads = __ballot(!same_prev);  // bit set where a run starts
        // distance to the start of my run
        const unsigned long long below = heads & ((2ull << lane) - 1ull);
        const int run_start = 63 - __builtin_clzll(below);
        const bool any_merge = (~heads) != 0ull;
        const unsigned long long heads_above = (lane == 63) ? 0ull : (heads >> (lane + 1));
        const bool is_tail = (lane == 63) || (heads_above & 1ull);
        if (any_merge) {  // wave-uniform
#pragma unroll
            for (uint32_t c = 0; c < (1u << D); c++) {
                const float w = ok ? corner_weight<D>(cell, c) : 0.0f;
                float v[NC];
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    v[k] = w * g[k];
                    if constexpr (sizeof(T) == 2) v[k] = (float)(half_t)v[k];  // per-contribution rounding (cu:350)
                }
#pragma unroll
                for (int k = 0; k < NC; k++) v[k] = wave_segscan_add(v[k], lane, run_start);
                if (ok && is_tail) {
                    T *p = gt + (size_t)corner_row<D>(cell, lv, c) * C;
                    if constexpr (NC == 2) atomic_add_pair(p, v[0], v[1]);
                    else atomic_add_one(p, v[0]);
                }
            }
            return;
        }
    }
    if (!ok) return;
#pragma unroll
    for (uint32_t c = 0; c < (1u << D); c++) {
        const float w = corner_weight<D>(cell, c);
        T *p = gt + (size_t)corner_row<D>(cell, lv, c) * C;
        if constexpr (NC == 2) atomic_add_pair(p, w * g[0], w * g[1]);
        else atomic_add_one(p, w * g[0]);
    }
}


// ------------------------------------------------------------------------------------------------ bucketed backward
// Device-scope atomics top out at ~20 G/s on MI355X no matter how the work is spread over XCDs (measured,
// scratch/bench_grid.py), and every LiDAR ray starts in the same few cells around the sensor, which serialises the
// coarse levels on a handful of addresses.  The fast path therefore never adds into HBM atomically:
//   pass 1 (k_grid_bwd_scatter): per (point, level) the run-merged corner contributions (row, w*g) are appended to
//           the pool of the table BUCKET they belong to (bucket = 8192 consecutive rows of one level); slots are
//           reserved with LDS counters + ONE global atomic per bucket per workgroup, so the pool writes of a
//           workgroup are contiguous per bucket;
//   pass 2 (k_grid_bwd_reduce): one workgroup per bucket accumulates its pool in a 128 KiB LDS image and adds the
//           touched rows into the gradient table with plain stores — it owns those rows.  The LDS image is 64-bit
//           FIXED POINT: measured on MI355X, ds_add_f32 sustains ~100 G adds/s chip-wide while ds_add_u64 keeps up
//           with the 6 TB/s pool stream (scratch/ldsbench), and integer accumulation makes the sum exact and
//           order-independent (fp16 contributions are multiples of 2^-24, so scale 2^24 loses nothing).
// A bucket whose pool overflows (non-uniform input) sends the excess to the level's SPILL list (full row + value, sized
// for the worst case); pass 2's workgroups of exactly those buckets (cursor > cap) scan the list and add their entries
// into the same fixed-point image.  Buckets with many entries are cut into slices (blockIdx.y of pass 2); the slices'
// integer images are written out and summed by the slice that finishes last.  Every contribution therefore enters ONE
// integer sum per table row, whichever pool / spill slot it landed in: the result does not depend on workgroup arrival
// order, for fp32 tables too (each contribution is truncated to 2^-40 on its own before the sum).
constexpr uint32_t kBucketRowsLog2 = 13;
constexpr uint32_t kBucketRows = 1u << kBucketRowsLog2;
constexpr uint32_t kMaxBucketsPerLevel = 64;

struct BucketPlan {
    uint32_t first_bucket[LNH_MAX_LEVELS + 1];  // prefix sum of buckets per level
    uint32_t cap[LNH_MAX_LEVELS];               // pool slots per bucket of that level
    uint64_t pool_off[LNH_MAX_LEVELS];          // byte offset of that level's pool region (16-byte aligned)
    uint64_t rows_off[LNH_MAX_LEVELS];          // byte offset of the level's 2-byte (row | code << 13) stream
    uint64_t spill_off[LNH_MAX_LEVELS];         // byte offset of the level's spill list (SpillEntry<T>, level-local rows)
    uint64_t partial_off;                       // byte offset of the slice images (kBucketRows * 2 int64 each)
    uint32_t spill_cap;                         // entries per spill list (= worst case of a level: B * 2^D)
    uint32_t partial_slots;
    uint32_t interleaved;                       // bit l: level l maps rows to buckets in 128-row groups (see bucket_of_row)
};

// ---- pool entries.  One entry carries the contributions of an x-NEIGHBOUR PAIR of corners (c0 = even corner index,
// c1 = c0 | 1): their rows differ in a way a 3-bit code describes, so the pair shares ONE 13-bit row field:
//   hashed power-of-two level:  r1 = r0 ^ ((2 << code) - 1)   (the x term of the hash is the x coordinate itself, so
//                               r0 ^ r1 = x ^ (x + 1) = 2^(t+1) - 1 with t = trailing ones of x; code = t <= 6)
//   every other level:          r1 = r0 + 1                    (dense x stride; code = 0)
//   code 7:                     single — only the first value applies (pairs the code cannot describe: t >= 7, one in
//                               128; a dense pair straddling a bucket boundary; every corner of the generic level
//                               classes: tiled grids, align_corners, smoothstep, non-power-of-two hash tables)
// Pool streams per level: values (two channel pairs: 8 bytes for fp16 tables, 16 for fp32) | 2-byte (row | code << 13):
// 10 bytes per PAIR of corners through HBM instead of 12, and half the LDS rank / stage / row-decode work per corner.
constexpr uint32_t kCodeSingle = 7;

// Rows -> buckets.  Hashed (and generic) levels: bucket = 8192 CONSECUTIVE rows — the hash spreads every batch evenly.
// Dense plain levels: rows are positions in space, every LiDAR ray leaves the same few cells around the sensor, and with
// consecutive rows level 0 is ONE bucket: one workgroup of the reduce pass worked 240 us on its 350 K entries, most of them
// on a handful of rows, while the 64 buckets of a hashed level took 85 us side by side — the whole pass waited for it
// (profiles/r04_reduce_levels.txt).  There the rows are dealt to up to 64 buckets in GROUPS OF 128: bucket = (row >> 7) & 63,
// position inside the bucket's image = (row >> 13) * 128 + (row & 127).  The eight corner rows of a cell differ by 1, R and
// R^2 (R = resolution + 1 >= 17), so the y / z neighbours of a hot cell land in other groups, i.e. other buckets; an x pair
// (r, r + 1) stays inside its group unless r & 127 == 127 (then it travels as two singles, one pair in 128).
constexpr uint32_t kGroupRowsLog2 = 7;
__host__ __device__ inline uint32_t bucket_of_row(uint32_t r, bool il) {
    return il ? (r >> kGroupRowsLog2) & (kMaxBucketsPerLevel - 1) : r >> kBucketRowsLog2;
}
__host__ __device__ inline uint32_t local_of_row(uint32_t r, bool il) {
    return il ? ((r >> kBucketRowsLog2) << kGroupRowsLog2) | (r & ((1u << kGroupRowsLog2) - 1)) : r & (kBucketRows - 1);
}
__host__ __device__ inline uint32_t row_of_local(uint32_t loc, uint32_t bk, bool il) {
    return il ? ((loc >> kGroupRowsLog2) << kBucketRowsLog2) | (bk << kGroupRowsLog2) | (loc & ((1u << kGroupRowsLog2) - 1))
              : (bk << kBucketRowsLog2) | loc;
}

template <typename T>
struct V2Of;
template <>
struct V2Of<half_t> {
    typedef half2_t type;
};
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <>
struct V2Of<float> {
    typedef f32x2_t type;
};
template <typename T>
using v2_t = typename V2Of<T>::type;

// spill-list entry: level-local row (19 bits) | code << 29, then the two channel pairs
template <typename T>
struct SpillEntry {
    uint32_t key;
    v2_t<T> a, b;
};

// value * 2^24 of an fp16 number as an exact 64-bit integer in 5 VALU operations: x = h * 2^24 is an integer with
// |x| < 2^40, so the double 1.5 * 2^52 + x is exact and its 52 mantissa bits hold 2^51 + x — low word = x mod 2^32, the 20
// bits above = 2^19 + floor(x / 2^32).  (The bit-twiddling form — (1024 | m) << (e - 1), negate — cost 12 per value and
// made the reduce pass VALU-bound: 34 VALU operations per pool entry against two ds_add_u64.)
__device__ __forceinline__ long long half_to_fixed24(half_t h) {
    const double d = __builtin_fma((double)(float)h, 16777216.0, 6755399441055744.0);
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, d);
    const int hi = (int)((uint32_t)(bits >> 32) & 0xFFFFFu) - 0x80000;
    return (long long)(((unsigned long long)(uint32_t)hi << 32) | (bits & 0xFFFFFFFFull));
}
// fixed-point images: fp16 contributions are exact multiples of 2^-24; fp32 ones get 2^-40 resolution, +-8e6 of range
template <typename T>
__device__ __forceinline__ void to_fixed(const v2_t<T> &v, long long &qa, long long &qb) {
    if constexpr (sizeof(T) == 2) {
        qa = half_to_fixed24(v[0]);
        qb = half_to_fixed24(v[1]);
    } else {
        qa = (long long)ldexp((double)v[0], 40);
        qb = (long long)ldexp((double)v[1], 40);
    }
}
template <typename T>
__device__ __forceinline__ v2_t<T> make_v2(float a, float b) {
    v2_t<T> r = {(T)a, (T)b};
    return r;
}
// second row of a pair (see above); `hashed` is workgroup-uniform
__device__ __forceinline__ uint32_t pair_row(uint32_t r0, uint32_t code, bool hashed) {
    return hashed ? r0 ^ ((2u << code) - 1u) : r0 + 1u;
}

// One thread = one (point, level): every workgroup reserves its pool slots with ONE global atomic per touched bucket
// for its 1024 points — the cursor atomics are device atomics too (~20 G/s), so fewer, larger reservations matter.
// CAP = LDS staging slots of a workgroup: 5 * 1024 serves the pair format (4 entries per point + the rare singles) with
// two workgroups per CU; 8 * 1024 is the worst case (launched when a level of the window is of a generic class: 8
// singles per point).  Entries beyond CAP (adversarial inputs: every x coordinate = 127 mod 128) bypass the staging and
// are written to their global slot by the thread that holds them.
template <typename T, int D, int CAP>
__global__ void __launch_bounds__(1024)
k_grid_bwd_scatter(const T *__restrict__ grad, const float *__restrict__ inputs, uint32_t B, GridMeta meta,
                   BucketPlan plan, char *__restrict__ pool_bytes, uint32_t *__restrict__ cursor,
                   uint32_t *__restrict__ spill_cursor, uint32_t align_rt, uint32_t interp_rt, uint32_t n_levels,
                   uint32_t level0, uint32_t b_begin, uint32_t B_all, T *__restrict__ grad_table) {
    // this launch handles the points [b_begin, b_begin + B) of a batch of B_all (large batches are walked in chunks)
    constexpr int C = 2, NCORN = 1 << D, NP = NCORN / 2, NTHREADS = 1024;
    typedef v2_t<T> V2;
    __shared__ uint2 lout[kMaxBucketsPerLevel];           // per bucket: {pool slot - staging slot, staging slots that fit}
    __shared__ uint32_t lsp[kMaxBucketsPerLevel];         // per bucket: spill slot - staging slot of the entries that do not
    __shared__ uint32_t lcnt[kMaxBucketsPerLevel];
    __shared__ uint32_t lstart[kMaxBucketsPerLevel + 1];  // first staging slot per bucket (workgroup-local)
    __shared__ __attribute__((aligned(16))) uint32_t skey[CAP];  // staged entries, bucket-sorted: row | code << 29
    __shared__ __attribute__((aligned(16))) V2 sa[CAP], sb[CAP];
    // level-fastest workgroup order: concurrently resident workgroups work on the same points at different levels
    // (the coordinates stay in L2, the cursor atomics spread over all levels' counters instead of 64 hot words).
    // (A persistent-workgroup variant of this kernel was measured 1.5x SLOWER: the hardware dispatcher overlaps the
    // phases of independent workgroups better than a barrier-separated item loop does.)
    const int lane = threadIdx.x & 63;
    const uint32_t level = level0 + blockIdx.x % n_levels, chunk = blockIdx.x / n_levels;  // window [level0, +n_levels)
    const LevelParams lv_rt = meta.lv[level];
    const uint32_t fb = plan.first_bucket[level], nb = plan.first_bucket[level + 1] - fb, cap = plan.cap[level];
    // level class resolved at compile time for the two usual classes (see k_grid_forward)
    auto body = [&](auto mode_c) {
    constexpr int MODE = decltype(mode_c)::value;
    LevelParams lv = lv_rt;
    if constexpr (MODE == 1) lv.flags = LV_HASH | LV_POW2;
    if constexpr (MODE == 2) lv.flags = LV_NOWRAP | (uint32_t)D;
    const uint32_t align = MODE ? 0u : align_rt, interp = MODE ? 0u : interp_rt;
    if (threadIdx.x < kMaxBucketsPerLevel) lcnt[threadIdx.x] = 0;

    V2 val[NCORN];      // (w * g0, w * g1) per corner, already in the pool's element type
    uint32_t row[NCORN];
    uint32_t xterm_xor;  // x term of corner 0 ^ x term of corner 1 (hashed levels: x ^ (x + 1))
    bool emit;
    {
        const uint32_t bl = chunk * blockDim.x + threadIdx.x;
        const bool in_range = bl < B;
        const uint32_t bc = b_begin + (in_range ? bl : 0);  // unconditional loads from a clamped index + selects
        float x[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            const float xv = inputs[(size_t)bc * D + d];
            x[d] = in_range ? xv : -1.0f;
        }
        const Vec<T, 2> gv = load_vec<T, 2>(grad + ((size_t)level * B_all + bc) * C);
        Cell<D> cell;
        bool ok = in_range;
#pragma unroll
        for (int d = 0; d < D; d++) ok = ok && !(x[d] < 0.0f || x[d] > 1.0f);
        // Lanes that are not `ok` never emit and never merge with a neighbour (unique key below): instead of zeroing
        // their 8 weights and 8 rows afterwards, they are located at the cube centre (3 selects) with a zero gradient
        // (every value stays finite — the scan multiplies foreign lanes by 0, and 0 * inf would poison a neighbour).
#pragma unroll
        for (int d = 0; d < D; d++) x[d] = ok ? x[d] : 0.5f;
        (void)locate<D>(x, lv, align != 0, interp, cell);
        const float g0 = ok ? (float)gv.v[0] : 0.0f, g1 = ok ? (float)gv.v[1] : 0.0f;
        // a sample whose upstream gradient is exactly zero (fp16 underflow behind an opaque surface, masked-out
        // rays) contributes nothing: it moves no entry of its own through the pool, but within three lanes of a sample
        // that has a gradient it stays a silent member of the run (k_grid_bwd_scatter_plain has the measurements:
        // dropped outright, it cut the runs of a trained field's coarse levels)
        const bool nonzero = ok && (g0 != 0.0f || g1 != 0.0f);
        {
            const unsigned long long in_m = __ballot(ok), nz_m = __ballot(nonzero);
            const unsigned long long nz1_m = nz_m | (nz_m << 1) | (nz_m >> 1);
            ok = ((in_m & (nz1_m | (nz1_m << 2) | (nz1_m >> 2))) >> lane) & 1ull;
        }
        // ---- run-merge inside 16-lane rows: consecutive lanes are consecutive samples of a ray, which share their cell
        //      on the coarser levels.  Runs are cut at DPP row starts (pure-VALU row_shr scan, no cross-row traffic), and
        //      a wave with fewer than kMinMerges mergeable lanes skips the scan altogether — on the fine levels almost
        //      every wave has SOME coincidental pair, and scanning 16 values to save one or two entries does not pay.
        constexpr int kMinMerges = 8;  // (4 / 16 / 24 measured within noise of each other)
        uint32_t key[D];
#pragma unroll
        for (int d = 0; d < D; d++) key[d] = ok ? cell.term[d][0] : 0xffffffffu - lane;
        // dense (coarse) levels: runs span whole waves and every entry lands in the same few buckets, so there the
        // full-wave scan (two extra cross-row steps) is worth its price
        const bool row_local = (lv.flags & LV_HASH) != 0;  // workgroup-uniform
        bool same_prev = row_local ? (lane & 15) != 0 : lane != 0;
#pragma unroll
        for (int d = 0; d < D; d++) {
            // wave_shr:1 (0x138): previous lane across rows
            const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key[d], 0x138, 0xf, 0xf, false);
            same_prev &= (up == key[d]);
        }
        unsigned long long heads = __ballot(!same_prev);
        const bool any_merge = __builtin_popcountll(~heads) >= kMinMerges;  // wave-uniform
        if (!any_merge) heads = ~0ull;
        const unsigned long long below = heads & ((2ull << lane) - 1ull);
        const int run_start = 63 - __builtin_clzll(below);
        const unsigned long long heads_above = (lane == 63) ? 0ull : (heads >> (lane + 1));
        // run tails; a zero-gradient sample that is a run of one emits nothing
        emit = ok && ((lane == 63) || (heads_above & 1ull)) && (nonzero || !((heads >> lane) & 1ull));
        xterm_xor = cell.term[0][0] ^ cell.term[0][1];
        const f32x2_t gg = {g0, g1};
#pragma unroll
        for (uint32_t c = 0; c < (uint32_t)NCORN; c++) {
            // w * (g0, g1) as one packed multiply, narrowed with one packed conversion: float product first, then the
            // per-contribution rounding to the table type, as the reference does (gridencoder.cu:350)
            const f32x2_t p = gg * corner_weight<D>(cell, c);
            val[c] = __builtin_convertvector(p, V2);
            row[c] = corner_row<D>(cell, lv, c);
        }
        if (any_merge) {  // wave-uniform: sum the runs in fp32 (one rounding of the sum when it is stored again)
            const SegScanMask sm = wave_segscan_mask(lane, run_start);
            float vv[2 * NCORN];
#pragma unroll
            for (uint32_t c = 0; c < (uint32_t)NCORN; c++) {
                vv[2 * c] = (float)val[c][0];
                vv[2 * c + 1] = (float)val[c][1];
            }
            row_segscan_add_n(vv, sm);
            if (!row_local) cross_segscan_add_n(vv, sm);
#pragma unroll
            for (uint32_t c = 0; c < (uint32_t)NCORN; c++) val[c] = make_v2<T>(vv[2 * c], vv[2 * c + 1]);
        }
    }
    // ---- pair the x-neighbours: entry p carries corners 2p and 2p + 1 unless the code cannot describe their rows
    uint32_t code[NP];
    bool single[NP], any_single = false;
    // hashed level: r0 ^ r1 = (x ^ (x + 1)) & (rows - 1) = 2^(t+1) - 1 for ALL four pairs of the point
    const uint32_t t_hash = (uint32_t)__builtin_popcount(xterm_xor & (lv.hashmap_size - 1u)) - 1u;
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const uint32_t r0 = row[2 * p];
        if constexpr (MODE == 1) {
            single[p] = t_hash >= kCodeSingle;
            code[p] = single[p] ? kCodeSingle : t_hash;
        } else if constexpr (MODE == 2) {
            single[p] = (r0 & (kBucketRows - 1)) == kBucketRows - 1;        // r1 = r0 + 1 opens the next bucket
            code[p] = single[p] ? kCodeSingle : 0u;
        } else {
            single[p] = true;
            code[p] = kCodeSingle;
        }
        any_single |= single[p];
    }
    const bool wave_singles = MODE == 0 ? true : (__ballot(emit && any_single) != 0ull);  // wave-uniform, rare
    __syncthreads();
    // ---- reserve pool slots: LDS rank per (bucket), one global atomic per touched bucket per workgroup
    uint32_t rank[NP], rank_x[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        // On dense (coarse) levels every emitting lane of the wave usually targets the SAME bucket: 64 returning
        // LDS atomics on one counter serialise, so aggregate — one lane adds the population count, the others
        // take their rank from the lane mask.  Mixed buckets (hashed levels) fall back to per-lane atomics.
        const uint32_t bk = row[2 * p] >> kBucketRowsLog2;
        rank[p] = 0;
        if (lv.flags & LV_HASH) {  // workgroup-uniform: hashed level, buckets are mixed -> per-lane atomics
            if (emit) rank[p] = atomicAdd(&lcnt[bk], 1u);
            continue;
        }
        const unsigned long long em = __ballot(emit);
        if (em == 0ull) continue;  // wave-uniform
        const int leader = __builtin_ctzll(em);
        const uint32_t bk0 = __shfl(bk, leader, 64);
        const bool uniform = __ballot(emit && bk != bk0) == 0ull;
        if (uniform) {
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&lcnt[bk0], (uint32_t)__builtin_popcountll(em));
            base = __shfl(base, leader, 64);
            rank[p] = base + (uint32_t)__builtin_popcountll(em & ((1ull << lane) - 1ull));
        } else if (emit) {
            rank[p] = atomicAdd(&lcnt[bk], 1u);
        }
    }
    if (wave_singles) {  // the second corners of pairs that travel as two singles
#pragma unroll
        for (int p = 0; p < NP; p++) {
            rank_x[p] = 0;
            if (emit && single[p]) rank_x[p] = atomicAdd(&lcnt[row[2 * p + 1] >> kBucketRowsLog2], 1u);
        }
    }
    __syncthreads();
    // global reservation + exclusive scan of the workgroup's bucket counts (first wave: one bucket counter per lane)
    if (threadIdx.x < 64) {
        static_assert(kMaxBucketsPerLevel == 64, "one bucket counter per lane of the first wave");
        const uint32_t n0 = lcnt[lane];
        uint32_t base = 0;
        if ((uint32_t)lane < nb && n0) base = atomicAdd(&cursor[fb + lane], n0);
        const uint32_t incl = wave_scan_add_u32(n0);  // DPP network: no LDS round trips while the atomics are in flight
        const uint32_t st = incl - n0;
        lstart[lane] = st;
        if (lane == 63) lstart[kMaxBucketsPerLevel] = incl;
        // staging slot pos of bucket bk goes to pool slot bk*cap + base + (pos - st) while base + (pos - st) < cap;
        // the `over` entries behind those go to consecutive slots of the level's spill list, reserved with ONE atomic
        // per workgroup (and none at all in the usual case of no overflow)
        const uint32_t fit = base < cap ? min(cap - base, n0) : 0u, over = n0 - fit;
        const uint32_t oincl = wave_scan_add_u32(over);
        uint32_t sp0 = 0;
        if (lane == 63 && oincl) sp0 = atomicAdd(&spill_cursor[level], oincl);
        sp0 = __shfl(sp0, 63, 64);
        lout[lane] = make_uint2((uint32_t)lane * cap + base - st, st + fit);
        lsp[lane] = sp0 + (oincl - over) - (st + fit);
    }
    __syncthreads();
    // pool streams of this level: values [slot][2] | rows [slot] | spill list; byte offsets inside a level fit 32 bits
    // (<= 4 M points per launch), so every store is base (scalar) + 32-bit lane offset — no 64-bit address arithmetic
    char *pvals = pool_bytes + plan.pool_off[level];
    char *prows = pool_bytes + plan.rows_off[level];
    char *spill = pool_bytes + plan.spill_off[level];
    // entry at staging position `pos` -> its pool slot (bucket-local row | code << 13) or, pool full, the spill list
    auto to_global = [&](uint32_t pos, uint32_t k, V2 a, V2 b, uint2 o) {
        if (pos < o.y) {
            const uint32_t slot = o.x + pos;
            struct Pair { V2 a, b; } pr = {a, b};
            *reinterpret_cast<Pair *>(pvals + slot * (uint32_t)sizeof(Pair)) = pr;
            *reinterpret_cast<unsigned short *>(prows + slot * 2u) =
                (unsigned short)((k & (kBucketRows - 1)) | ((k >> 29) << kBucketRowsLog2));
        } else {
            const uint32_t sp = lsp[(k & 0x7ffffu) >> kBucketRowsLog2] + pos;
            if (sp < plan.spill_cap) {
                SpillEntry<T> e = {k, a, b};
                *reinterpret_cast<SpillEntry<T> *>(spill + sp * (uint32_t)sizeof(SpillEntry<T>)) = e;
            } else {
                // The spill list is full too (it holds 1/16 of a level's worst case: a batch whose points crowd into a
                // few buckets of a DENSE level without merging): these entries go straight into the table with device
                // atomics — always correct, slow (~20 G/s), and the one place where a sum depends on arrival order.
                const uint32_t r0 = k & 0x7ffffu, cd = k >> 29;
                T *gt = grad_table + (size_t)lv.offset * 2;
                atomic_add_pair(gt + (size_t)r0 * 2, (float)a[0], (float)a[1]);
                if (cd != kCodeSingle) {
                    // (pairs never leave their bucket: a hashed pair differs in the low 7 bits, a dense one is r0 + 1)
                    atomic_add_pair(gt + (size_t)pair_row(r0, cd, MODE == 1) * 2, (float)b[0], (float)b[1]);
                }
            }
        }
    };
    // ---- stage the entries in LDS grouped by bucket, then stream them out: consecutive lanes write consecutive pool
    //      slots (a per-lane scatter of small stores costs one cache-line transaction per lane)
    auto put = [&](uint32_t r, uint32_t cd, V2 a, V2 b, uint32_t rk) {
        const uint32_t bk = r >> kBucketRowsLog2, pos = lstart[bk] + rk, k = r | (cd << 29);
        if (pos < (uint32_t)CAP) {
            skey[pos] = k;
            sa[pos] = a;
            sb[pos] = b;
        } else {
            to_global(pos, k, a, b, lout[bk]);
        }
    };
    if (emit) {
#pragma unroll
        for (int p = 0; p < NP; p++)
            put(row[2 * p], code[p], val[2 * p], val[2 * p + 1], rank[p]);
    }
    if (wave_singles && emit) {
#pragma unroll
        for (int p = 0; p < NP; p++)
            if (single[p])
                put(row[2 * p + 1], kCodeSingle, val[2 * p + 1], make_v2<T>(0.0f, 0.0f), rank_x[p]);
    }
    __syncthreads();
    // (workgroup-uniform, made scalar: the rounds below stop at it without touching LDS for slots nobody filled — on
    //  average a workgroup stages 2.1 entries per thread of the 5 it has room for)
    const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(lstart[kMaxBucketsPerLevel], (uint32_t)CAP));
    // every thread moves up to CAP / 1024 staged entries: all their LDS reads (entry, then the bucket's slot map) are
    // issued before the first use — one round trip for the batch instead of two dependent ones per entry.  (Moving PAIRS of
    // neighbouring entries per thread — 8-byte LDS reads, one 16-byte value store + one 4-byte row store per pair where both
    // sit in one bucket — was measured slower: 709 vs 677 us; the wide stores land on 8- / 2-byte boundaries.  So was ONE 16-byte
    // LDS record per staged entry instead of three 4-byte arrays: 755 vs 675 us.)
    constexpr int NW = CAP / NTHREADS;
    static_assert(CAP % NTHREADS == 0, "staging slots are dealt to the threads in rounds");
    uint32_t ks[NW];
    V2 as[NW], bs[NW];
    uint2 os[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) {
        if ((uint32_t)i * NTHREADS >= total) break;  // scalar branch
        const uint32_t pos = threadIdx.x + (uint32_t)i * NTHREADS, pc = pos < total ? pos : 0u;
        ks[i] = skey[pc];
        as[i] = sa[pc];
        bs[i] = sb[pc];
    }
#pragma unroll
    for (int i = 0; i < NW; i++) {
        if ((uint32_t)i * NTHREADS >= total) break;
        os[i] = lout[((ks[i] & 0x7ffffu) >> kBucketRowsLog2) & (kMaxBucketsPerLevel - 1)];
    }
#pragma unroll
    for (int i = 0; i < NW; i++) {
        if ((uint32_t)i * NTHREADS >= total) break;
        const uint32_t pos = threadIdx.x + (uint32_t)i * NTHREADS;
        if (pos < total) to_global(pos, ks[i], as[i], bs[i], os[i]);
    }
    };
    // the two plain classes (hashed power-of-two table / dense level, linear interpolation, align_corners off) are
    // k_grid_bwd_scatter_plain's levels: this kernel serves the generic classes of a window (8 singles per point)
    const bool plain = align_rt == 0 && interp_rt == 0;
    if (plain && (lv_rt.flags & (LV_HASH | LV_POW2)) == (LV_HASH | LV_POW2)) return;
    if (plain && !(lv_rt.flags & LV_HASH) && (lv_rt.flags & LV_NOWRAP) && (lv_rt.flags & 15u) == (uint32_t)D) return;
    body(std::integral_constant<int, 0>{});
}
// true when every corner of the level travels as a single (generic class): 8 entries per point
__host__ __device__ inline bool level_is_generic(const LevelParams &lv, uint32_t D, bool plain) {
    if (plain && (lv.flags & (LV_HASH | LV_POW2)) == (LV_HASH | LV_POW2)) return false;
    if (plain && !(lv.flags & LV_HASH) && (lv.flags & LV_NOWRAP) && (lv.flags & 15u) == D) return false;
    return true;
}

#ifndef LNH_SCATTER_THREADS
#define LNH_SCATTER_THREADS 1024
#endif
constexpr int kScatterThreads = LNH_SCATTER_THREADS;  // points (= threads) of a scatter workgroup; 5 staging slots each
// Paired write-out (round 5): a workgroup reserves an EVEN number of slots per bucket (an odd count is padded with one
// zero-valued single on the bucket's local row 0, which adds nothing), so every bucket's segment starts at an even staging
// slot AND at an even pool slot; a thread then writes TWO consecutive entries with one 16-byte value store and one 4-byte row
// store (fp16 tables; 2 x 16 + 4 for fp32) instead of two 8-byte and two 2-byte stores, and reads its staging with 8-byte
// LDS loads.  Round 3 tried pairs without the padding (the wide stores landed on 8- / 2-byte boundaries: 709 against 677 us).
#ifndef LNH_SCATTER_PAIRED
#define LNH_SCATTER_PAIRED 1
#endif
constexpr bool kScatterPaired = LNH_SCATTER_PAIRED != 0;
// ---- scatter pass for the two PLAIN level classes (hashed power-of-two tables / dense levels, linear interpolation,
// align_corners off — every level of the usual configuration).  Same contract as k_grid_bwd_scatter (which keeps the
// generic classes): same pool format, same cursors, same spill rules; what differs is the instruction budget.  The scatter
// pass is issue-bound (profiles/r03_grid_counters.json: 324 VALU + 168 SALU per wave), so this kernel is written to the
// instruction:
//   * every lane predicate that is a function of wave-wide facts (who continues a run, who emits, the six "take" masks of
//     the segmented scan) lives in SGPR PAIRS: built from two ballots with scalar shifts / ands and handed to the VALU as a
//     v_cndmask source or an exec mask (__builtin_amdgcn_inverse_ballot_w64) — no 64-bit lane arithmetic, no run_start;
//   * the scan executes only the steps the LONGEST run of the wave needs (scalar test on the masks): runs of 2 — the usual
//     case on the middle levels — cost one step instead of four;
//   * the range test is two min3 / max3 + two compares; locate() without its own second range test;
//   * a workgroup whose entries all fit their staging and pool slots (flag from the reserving wave; always, short of
//     adversarial inputs) stages and writes out WITHOUT per-entry tests: the staged key carries the bucket's LDS address
//     pre-shifted (key >> 16) and the pool's 16-bit (row | code << 13) in its low half, so the write-out of an entry is
//     one shift, one LDS read, one add, two shifts and two stores; LDS addresses of the staging reads are immediates;
//   * level / chunk come from a 2-D grid (x = level: the level-fastest order), not from a division.
// (amdgpu_num_sgpr: two 1024-thread workgroups per CU need 8 waves per SIMD, and gfx950 admits 8 only up to 80 SGPRs
//  including VCC / flat-scratch / XNACK — at the 84 the unconstrained allocation took, ONE workgroup per CU ran and the
//  pass took 959 us instead of 682; amdgpu_waves_per_eu: the same for the VGPR side, 64 at most — fp16 tables; fp32 tables
//  stage 100 KB and run one workgroup per CU anyway; profiles/r04_scatter_phases.txt)
template <typename T, int NTHREADS, int CAP>
__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_num_sgpr(72), amdgpu_waves_per_eu(sizeof(T) == 2 ? 8 : 4)))
k_grid_bwd_scatter_plain(const T *__restrict__ grad, const float *__restrict__ inputs, uint32_t B, GridMeta meta,
                         BucketPlan plan, char *__restrict__ pool_bytes, uint32_t *__restrict__ cursor,
                         uint32_t *__restrict__ spill_cursor, uint32_t level0, uint32_t b_begin, uint32_t B_all,
                         T *__restrict__ grad_table) {
    constexpr int NP = 4;
    typedef v2_t<T> V2;
    struct Pair { V2 a, b; };
    __shared__ uint32_t lcnt[kMaxBucketsPerLevel];
    __shared__ uint32_t lstart[kMaxBucketsPerLevel + 1];  // first staging slot per bucket; [64] = staged entries in all
    __shared__ uint32_t lox[kMaxBucketsPerLevel];          // pool slot - staging slot
    __shared__ uint32_t lofit[kMaxBucketsPerLevel];        // first staging slot of the bucket that does NOT fit its pool
    __shared__ uint32_t lsp[kMaxBucketsPerLevel];          // spill slot - staging slot of those
    __shared__ uint32_t lflag;                             // 1: everything fits staging and pool (the fast path)
    // staged key: bits 0-12 bucket-local row | 13-15 code | 16-23 bucket * 4
    __shared__ __attribute__((aligned(16))) uint32_t skey[CAP];
    // (the two value pairs of an entry in ONE 8-byte LDS record — one write per entry, one 16-byte read per written-out slot
    //  pair — was measured in round 5: 766 against 531 us, the random 8-byte LDS writes of the staging conflict; round 3 saw
    //  the same with a 16-byte record)
    __shared__ __attribute__((aligned(16))) V2 sa[CAP], sb[CAP];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t level = level0 + blockIdx.x, chunk = blockIdx.y;
    const LevelParams lv = meta.lv[level];
    const bool cls_hash = (lv.flags & (LV_HASH | LV_POW2)) == (LV_HASH | LV_POW2);
    const bool cls_dense = !(lv.flags & LV_HASH) && (lv.flags & LV_NOWRAP) && (lv.flags & 15u) == 3u;
    if (!cls_hash && !cls_dense) return;  // generic class: k_grid_bwd_scatter's level
    const uint32_t fb = plan.first_bucket[level], nb = plan.first_bucket[level + 1] - fb, cap = plan.cap[level];
    auto lds_at = [](uint32_t *base, uint32_t byte_off) -> uint32_t & {
        return *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(base) + byte_off);
    };
    auto body = [&](auto mode_c) {
    constexpr int MODE = decltype(mode_c)::value;  // 1 hashed, 2 dense
    LNH_MARK("A load+locate");
    if (tid < kMaxBucketsPerLevel) lcnt[tid] = 0;
    // ---- load (unconditional, clamped index), range test, cell
    const uint32_t bl = chunk * (uint32_t)NTHREADS + tid;
    const bool in_range = bl < B;
    const uint32_t bc = b_begin + (in_range ? bl : 0u);
    const float *px = inputs + (size_t)bc * 3;
    float x0 = px[0], x1 = px[1], x2 = px[2];
    const Vec<T, 2> gv = load_vec<T, 2>(grad + ((size_t)level * B_all + bc) * 2);
    __syncthreads();  // (the zeroed counters; the loads above stay in flight across it)
    // !(x < 0 || x > 1) per coordinate (gridencoder.cu:158-163), as max3 / min3: a NaN coordinate drops out of both
    bool ok = in_range & !(__builtin_fmaxf(__builtin_fmaxf(x0, x1), x2) > 1.0f) &
              !(__builtin_fminf(__builtin_fminf(x0, x1), x2) < 0.0f);
    const float g0 = (float)gv.v[0], g1 = (float)gv.v[1];
    // A sample whose upstream gradient is exactly zero contributes nothing.  It is not moved through the pool ON ITS OWN (see
    // emit_m below), but it stays a silent member of a run of same-cell neighbours: dropped outright — as rounds 3-4 did — it
    // CUT that run.  A field that has trained for a few hundred steps has such samples sprinkled along every ray: the coarse
    // levels' entry counts went from 0.47 M to 1.7 M (level 0) with them, one bucket of level 0 past its pool share and into
    // two slices, and the reduce pass from 310 to 680 us (profiles/r05_reduce_drift.txt).  Its zeros add nothing to a sum.
    bool nonzero;
    if constexpr (sizeof(T) == 2) {
        uint32_t raw;
        __builtin_memcpy(&raw, &gv, 4);
        nonzero = (raw & 0x7fff7fffu) != 0u;
    } else {
        nonzero = (g0 != 0.0f) | (g1 != 0.0f);
    }
    // Lanes that are not `ok` never emit and never join a run (masks below); they only must stay FINITE, because the scan
    // multiplies foreign lanes by 0: out-of-range coordinates are replaced by the cube centre.
    x0 = ok ? x0 : 0.5f;
    x1 = ok ? x1 : 0.5f;
    x2 = ok ? x2 : 0.5f;
    const float sc = lv.scale;
    const float p0 = fmaf(x0, sc, 0.5f), p1 = fmaf(x1, sc, 0.5f), p2 = fmaf(x2, sc, 0.5f);
    const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
    const uint32_t c0 = (uint32_t)f0, c1 = (uint32_t)f1, c2 = (uint32_t)f2;
    const float fr0 = p0 - f0, fr1 = p1 - f1, fr2 = p2 - f2;
    LNH_MARK("B rows");
    // ---- rows of the four even corners (pair p = y + 2 z holds corners 2p, 2p + 1)
    uint32_t r0[NP];
    uint32_t xm = 0;       // hashed: r1 = r0 ^ xm
    bool single[NP];       // the pair travels as two singles
    uint32_t codesh;       // hashed: code << 13 of all four pairs
    if constexpr (MODE == 1) {
        const uint32_t m = lv.hashmap_size - 1u;
        const uint32_t ty0 = c1 * prime_of(1), ty1 = ty0 + prime_of(1), tz0 = c2 * prime_of(2), tz1 = tz0 + prime_of(2);
        r0[0] = ((ty0 ^ tz0) ^ c0) & m;
        r0[1] = ((ty1 ^ tz0) ^ c0) & m;
        r0[2] = ((ty0 ^ tz1) ^ c0) & m;
        r0[3] = ((ty1 ^ tz1) ^ c0) & m;
        xm = (c0 ^ (c0 + 1u)) & m;  // 2^(t+1) - 1, t = trailing ones of x
        const uint32_t t_hash = (uint32_t)__builtin_popcount(xm) - 1u;
        const bool sg = t_hash >= kCodeSingle;
#pragma unroll
        for (int p = 0; p < NP; p++) single[p] = sg;
        codesh = (sg ? kCodeSingle : t_hash) << kBucketRowsLog2;
    } else {
        const uint32_t R = lv.resolution + 1u, RR = R * R;
        const uint32_t base = c0 + c1 * R + c2 * RR;
        r0[0] = base;
        r0[1] = base + R;
        r0[2] = base + RR;
        r0[3] = base + R + RR;
#pragma unroll
        for (int p = 0; p < NP; p++)  // r0 + 1 opens the next 128-row group, i.e. another bucket (bucket_of_row)
            single[p] = (r0[p] & ((1u << kGroupRowsLog2) - 1)) == (1u << kGroupRowsLog2) - 1;
        codesh = 0;
    }
    LNH_MARK("C masks");
    // ---- run-merge masks (SGPR pairs).  A lane CONTINUES the run of its lower neighbour when both are ok and lie in the
    //      same cell; hashed levels cut runs at 16-lane row starts (row-local scan), dense levels scan the whole wave.
    const bool same = ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)c0, 0x138, 0xf, 0xf, false) == c0) &
                      ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)c1, 0x138, 0xf, 0xf, false) == c1) &
                      ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)c2, 0x138, 0xf, 0xf, false) == c2);
    // members of runs: every in-range sample with a non-zero gradient, and a zero-gradient one within THREE LANES of such a
    // sample — the bridge inside a run (zero stretches of up to six samples are bridged from both ends); the long zero
    // stretches behind a surface, and whole waves of them, stay out of the masks, the scan and the pool as they always did.
    // Measured on fixed patterns (profiles/r05_reduce_drift.txt; backward = scatter + reduce, us): zeros sprinkled over 30 /
    // 50 / 70 % of the samples — dropped outright 1478 / - / 1696, every sample a member 791 / 712 / 633, this rule 786 / 708 /
    // 638; the far 50 / 85 % of every ray zero — 622 / 421, 692 / 570, 619 / 430.
    const unsigned long long in_m = __builtin_amdgcn_ballot_w64(ok);
    const unsigned long long nz_m = __builtin_amdgcn_ballot_w64(nonzero) & in_m;
    const unsigned long long nz1_m = nz_m | (nz_m << 1) | (nz_m >> 1);
    const unsigned long long ok_m = in_m & (nz1_m | (nz1_m << 2) | (nz1_m >> 2));
    constexpr unsigned long long kRowStarts = MODE == 1 ? 0x0001000100010001ull : 1ull;
    unsigned long long cont = __builtin_amdgcn_ballot_w64(same) & ok_m & (ok_m << 1) & ~kRowStarts;
    constexpr int kMinMerges = 8;  // a wave with fewer mergeable lanes skips the scan (4 / 16 / 24 measured within noise)
    if (__builtin_popcountll(cont) < kMinMerges) cont = 0ull;
    // run tails; a zero-gradient sample that continues nobody's run (a run of one) emits nothing
    const unsigned long long emit_m = ok_m & ~(cont >> 1) & (nz_m | cont);
    const bool emit = __builtin_amdgcn_inverse_ballot_w64(emit_m);
    LNH_MARK("F rank");
    // ---- rank inside the workgroup's (bucket) counters
    uint32_t bk4[NP], rank[NP], rank_x[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        // bucket * 4 = LDS byte offset of its counter (hashed: 8192 consecutive rows; dense: 128-row groups dealt to 64)
        bk4[p] = (r0[p] >> ((MODE == 1 ? kBucketRowsLog2 : kGroupRowsLog2) - 2)) & 0xfcu;
        rank[p] = 0;
        rank_x[p] = 0;
    }
    bool any_single = false;
#pragma unroll
    for (int p = 0; p < NP; p++) any_single |= single[p];
    const bool wave_singles = __builtin_amdgcn_ballot_w64(emit && any_single) != 0ull;  // wave-uniform, rare
    // The returning LDS atomics are issued BEFORE the value arithmetic: the LDS atomic unit retires ~3 lanes per clock and
    // CU (29 cycles per wave instruction here), the two resident workgroups of a CU run phase-locked, so nobody else hides
    // that time — the wave's own ~100 VALU instructions of weights / products / scan do (profiles/r04_scatter_phases.txt).
    // per-lane returning atomics.  (Dense levels: a wave emits a few run tails, and with the rows dealt to 64 buckets in
    // 128-row groups their four pairs go to four buckets — the wave-aggregated rank of round 3, one atomic for a whole
    // wave on the level's single bucket, has nothing left to aggregate and cost ~50 VALU instructions per wave.)
    if (emit) {
#pragma unroll
        for (int p = 0; p < NP; p++) rank[p] = atomicAdd(&lds_at(lcnt, bk4[p]), 1u);
    }
    if (wave_singles) {  // the second corners of pairs that travel as two singles
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const uint32_t r1 = MODE == 1 ? (r0[p] ^ xm) : r0[p] + 1u;
            if (emit && single[p]) rank_x[p] = atomicAdd(&lcnt[bucket_of_row(r1, MODE == 2)], 1u);
        }
    }
    LNH_MARK("D values");
    // ---- w * (g0, g1) per corner: float product, then the per-contribution rounding to the table type (gridencoder.cu:350).
    //      (Summing the fp32 PRODUCTS of a merged run and rounding once — 24 conversions less per scanning lane — was built
    //      in round 4 and dropped: 844 us against 838 for the backward, i.e. nothing, and rows whose every addend underflows
    //      in fp16 then hold a tiny sum where the reference and the oracle hold 0.)
    V2 val[8];
    {
        const float wx0 = 1.0f - fr0, wy0 = 1.0f - fr1, wz0 = 1.0f - fr2;
        const float wxy[4] = {wx0 * wy0, fr0 * wy0, wx0 * fr1, fr0 * fr1};
        const f32x2_t gg = {g0, g1};
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const float w = wxy[c & 3] * ((c & 4) ? fr2 : wz0);
            const f32x2_t p = gg * w;
            val[c] = __builtin_convertvector(p, V2);
        }
    }
    LNH_MARK("E scan");
    if (cont != 0ull) {  // wave-uniform: sum the runs in fp32 (one rounding of the sum when it is stored again)
        // take masks of the scan steps: step s adds the value 2^s lanes below iff the lanes (lane - 2^s, lane] all continue
        const unsigned long long k0 = cont, k1 = k0 & (k0 << 1), k2 = k1 & (k1 << 2), k3 = k2 & (k2 << 4);
        float vv[16];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            vv[2 * c] = (float)val[c][0];
            vv[2 * c + 1] = (float)val[c][1];
        }
        {
            const float m0 = __builtin_amdgcn_inverse_ballot_w64(k0) ? 1.0f : 0.0f;
            segscan_step_n<0>(vv, m0);
        }
        if (k1 != 0ull) {  // a run of 3 or more somewhere in the wave
            const float m1 = __builtin_amdgcn_inverse_ballot_w64(k1) ? 1.0f : 0.0f;
            segscan_step_n<1>(vv, m1);
            if (k2 != 0ull) {
                const float m2 = __builtin_amdgcn_inverse_ballot_w64(k2) ? 1.0f : 0.0f;
                segscan_step_n<2>(vv, m2);
                if (k3 != 0ull) {
                    const float m3 = __builtin_amdgcn_inverse_ballot_w64(k3) ? 1.0f : 0.0f;
                    segscan_step_n<3>(vv, m3);
                }
            }
        }
        if constexpr (MODE == 2) {
            // cross-row steps: lanes of rows 1, 3 (then 2, 3) whose run began before their row (before lane 32) take the
            // sum the last lane of the preceding row (lane 31) holds.  pre = lanes whose whole row prefix continues.
            unsigned long long pre = cont;
            pre &= (pre << 1) | 0x0001000100010001ull;
            pre &= (pre << 2) | 0x0003000300030003ull;
            pre &= (pre << 4) | 0x000f000f000f000full;
            pre &= (pre << 8) | 0x00ff00ff00ff00ffull;
            const unsigned long long k4 = pre & 0xffff0000ffff0000ull;
            const unsigned long long k5 = (pre & 0x0000ffff00000000ull) | (((pre >> 47) & 1ull) ? (pre & 0xffff000000000000ull) : 0ull);
            if (k4 != 0ull) {
                const float m4 = __builtin_amdgcn_inverse_ballot_w64(k4) ? 1.0f : 0.0f;
                segscan_step_n<4>(vv, m4);
            }
            if (k5 != 0ull) {  // (not nested: a run may cross the lane 31 | 32 boundary only)
                const float m5 = __builtin_amdgcn_inverse_ballot_w64(k5) ? 1.0f : 0.0f;
                segscan_step_n<5>(vv, m5);
            }
        }
#pragma unroll
        for (int c = 0; c < 8; c++) val[c] = make_v2<T>(vv[2 * c], vv[2 * c + 1]);
    }
    __syncthreads();
    LNH_MARK("G reserve");
    // ---- global reservation, in two halves around the staging pass: the first wave ISSUES one returning global atomic per
    //      touched bucket and publishes the workgroup-local exclusive scan of the counts (all the staging needs); the
    //      atomics' round trip (~66 us of the pass when it was waited for here) runs under the staging of all 16 waves, and
    //      the first wave turns the returned bases into pool / spill offsets only afterwards.
    uint32_t rs_n0 = 0, rs_base = 0, rs_st = 0, rs_incl = 0;
    bool pad_unstaged = false;  // (paired write-out) this lane's padding entry lies beyond the staging area
    if (tid < 64) {
        static_assert(kMaxBucketsPerLevel == 64, "one bucket counter per lane of the first wave");
        const uint32_t cnt = lcnt[lane];
        rs_n0 = kScatterPaired ? (cnt + 1u) & ~1u : cnt;  // slots reserved: even
        if (lane < nb && rs_n0) rs_base = atomicAdd(&cursor[fb + lane], rs_n0);
        rs_incl = wave_scan_add_u32(rs_n0);  // DPP network: no LDS round trips while the atomics are in flight
        rs_st = rs_incl - rs_n0;
        lstart[lane] = rs_st;
        if (lane == 63) lstart[kMaxBucketsPerLevel] = rs_incl;
        if (kScatterPaired && (cnt & 1u)) {  // the padding entry: a single with zero values on the bucket's local row 0
            const uint32_t q = rs_st + cnt;
            if (q < (uint32_t)CAP) {
                skey[q] = (kCodeSingle << kBucketRowsLog2) | (lane << 18);
                sa[q] = make_v2<T>(0.0f, 0.0f);
                sb[q] = make_v2<T>(0.0f, 0.0f);
            } else {
                pad_unstaged = true;
            }
        }
    }
    __syncthreads();
    LNH_MARK("H stage");
    // pool streams of this level: values [slot][2] | rows [slot] | spill list; byte offsets inside a level fit 32 bits
    // (<= 4 M points per launch), so every store is base (scalar) + 32-bit lane offset
    char *pvals = pool_bytes + plan.pool_off[level];
    char *prows = pool_bytes + plan.rows_off[level];
    const uint32_t total_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)lstart[kMaxBucketsPerLevel]);
    const uint32_t total = min(total_all, (uint32_t)CAP);
    const bool all_staged = total_all <= (uint32_t)CAP;  // workgroup-uniform; false only for adversarial inputs
    auto key_of = [&](uint32_t r, uint32_t csh) {  // staged key of level-local row r, code << 13
        return local_of_row(r, MODE == 2) | csh | (bucket_of_row(r, MODE == 2) << 18);
    };
    constexpr int NW = CAP / NTHREADS;
    static_assert(CAP % NTHREADS == 0, "staging slots are dealt to the threads in rounds");
    // ---- stage in bucket order.  Needs the workgroup-local scan only.  An entry whose position lies beyond the staging
    //      area (more than CAP entries in the workgroup: every x coordinate = 127 mod 128, ...) stays in its thread's
    //      registers and is written to its global slot by that thread after the bases have arrived.
    uint32_t pos[NP], pos_x[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) pos[p] = pos_x[p] = 0;
    auto stage = [&](auto checked_c) {  // (unswitched by hand: the usual, unchecked copy has no per-entry test)
        constexpr bool CHECKED = decltype(checked_c)::value;
        if (emit) {
#pragma unroll
            for (int p = 0; p < NP; p++) pos[p] = lds_at(lstart, bk4[p]);
#pragma unroll
            for (int p = 0; p < NP; p++) {
                pos[p] += rank[p];
                const uint32_t csh = MODE == 1 ? codesh : (single[p] ? kCodeSingle << kBucketRowsLog2 : 0u);
                if (!CHECKED || pos[p] < (uint32_t)CAP) {
                    skey[pos[p]] = local_of_row(r0[p], MODE == 2) | (csh | (bk4[p] << 16));
                    sa[pos[p]] = val[2 * p];
                    sb[pos[p]] = val[2 * p + 1];
                }
            }
        }
        if (wave_singles && emit) {  // the second corners of pairs that travel as two singles
#pragma unroll
            for (int p = 0; p < NP; p++)
                if (single[p]) {
                    const uint32_t r1 = MODE == 1 ? (r0[p] ^ xm) : r0[p] + 1u;
                    pos_x[p] = lstart[bucket_of_row(r1, MODE == 2)] + rank_x[p];
                    if (!CHECKED || pos_x[p] < (uint32_t)CAP) {
                        skey[pos_x[p]] = key_of(r1, kCodeSingle << kBucketRowsLog2);
                        sa[pos_x[p]] = val[2 * p + 1];
                        sb[pos_x[p]] = make_v2<T>(0.0f, 0.0f);
                    }
                }
        }
    };
    if (all_staged) stage(std::false_type{});
    else stage(std::true_type{});
    // ---- second half of the reservation (first wave): staging slot pos of bucket bk goes to pool slot bk * cap + base +
    //      (pos - st) while base + (pos - st) < cap; the `over` entries behind those go to consecutive slots of the level's
    //      spill list, reserved with ONE atomic per workgroup (and none at all in the usual case of no overflow)
    if (tid < 64) {
        const uint32_t fit = rs_base < cap ? min(cap - rs_base, rs_n0) : 0u, over = rs_n0 - fit;
        const uint32_t oincl = wave_scan_add_u32(over);
        uint32_t sp0 = 0;
        if (lane == 63 && oincl) sp0 = atomicAdd(&spill_cursor[level], oincl);
        sp0 = (uint32_t)__builtin_amdgcn_readlane((int)sp0, 63);
        lox[lane] = lane * cap + rs_base - rs_st;
        lofit[lane] = rs_st + fit;
        lsp[lane] = sp0 + (oincl - over) - (rs_st + fit);
        if (lane == 63) lflag = oincl == 0u ? 1u : 0u;  // 1: every entry of the workgroup has a pool slot
    }
    __syncthreads();
    LNH_MARK("I writeout");
    const bool fits = __builtin_amdgcn_readfirstlane((int)lflag) != 0;
    if (fits) {
        if constexpr (kScatterPaired) {
            // ---- write out in PAIRS of slots (see kScatterPaired): slot pair (2j, 2j + 1) of the staging area belongs to one
            //      bucket and goes to an even pool slot; all LDS reads of a thread before the first use
            constexpr int NW2 = (CAP + 2 * NTHREADS - 1) / (2 * NTHREADS);
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            struct V2x2 { V2 v[2]; };
            u32x2 ks2[NW2];
            V2x2 as2[NW2], bs2[NW2];
            uint32_t dl2[NW2];
#pragma unroll
            for (int i = 0; i < NW2; i++) {
                if ((uint32_t)i * 2 * NTHREADS >= total) break;  // scalar branch
                const uint32_t q = 2 * (tid + (uint32_t)i * NTHREADS);
                const uint32_t qc = q < (uint32_t)CAP ? q : 0u;  // (CAP is a multiple of 2: a pair never straddles the end)
                ks2[i] = *reinterpret_cast<const u32x2 *>(&skey[qc]);
                as2[i] = *reinterpret_cast<const V2x2 *>(&sa[qc]);
                bs2[i] = *reinterpret_cast<const V2x2 *>(&sb[qc]);
            }
#pragma unroll
            for (int i = 0; i < NW2; i++) {
                if ((uint32_t)i * 2 * NTHREADS >= total) break;
                dl2[i] = lds_at(lox, (ks2[i][0] >> 16) & 0xffu);
            }
#pragma unroll
            for (int i = 0; i < NW2; i++) {
                if ((uint32_t)i * 2 * NTHREADS >= total) break;
                const uint32_t q = 2 * (tid + (uint32_t)i * NTHREADS);
                if (q < total) {  // (total is even: the pair is whole)
                    const uint32_t slot = dl2[i] + q;
                    struct Pair2 { V2 a0, b0, a1, b1; };
                    Pair2 pr = {as2[i].v[0], bs2[i].v[0], as2[i].v[1], bs2[i].v[1]};
                    *reinterpret_cast<Pair2 *>(pvals + slot * (uint32_t)sizeof(Pair)) = pr;
                    *reinterpret_cast<uint32_t *>(prows + slot * 2u) = (ks2[i][0] & 0xffffu) | (ks2[i][1] << 16);
                }
            }
        } else {
        // ---- write out: consecutive lanes write consecutive pool slots; all LDS reads of a thread before the first use
        uint32_t ks[NW], dl[NW];
        V2 as[NW], bs[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) {
            if ((uint32_t)i * NTHREADS >= total) break;  // scalar branch
            // (slots at and beyond `total` hold stale bytes: read, never stored; their 8-bit bucket field stays inside lox)
            const uint32_t q = tid + (uint32_t)i * NTHREADS;
            ks[i] = skey[q];
            as[i] = sa[q];
            bs[i] = sb[q];
        }
#pragma unroll
        for (int i = 0; i < NW; i++) {
            if ((uint32_t)i * NTHREADS >= total) break;
            dl[i] = lds_at(lox, (ks[i] >> 16) & 0xffu);
        }
#pragma unroll
        for (int i = 0; i < NW; i++) {
            if ((uint32_t)i * NTHREADS >= total) break;
            const uint32_t q = tid + (uint32_t)i * NTHREADS;
            if (q < total) {
                const uint32_t slot = dl[i] + q;
                Pair pr = {as[i], bs[i]};
                *reinterpret_cast<Pair *>(pvals + slot * (uint32_t)sizeof(Pair)) = pr;
                *reinterpret_cast<unsigned short *>(prows + slot * 2u) = (unsigned short)ks[i];
            }
        }
        }
        if (all_staged) return;
    }
    LNH_MARK("J general");
    // ---- the general path: a bucket beyond its pool share (spill list, then atomics), entries beyond the staging area
    char *spill = pool_bytes + plan.spill_off[level];
    auto to_global = [&](uint32_t q, uint32_t k, V2 a, V2 b2) {
        const uint32_t bo = k >> 16;  // bucket * 4
        if (q < lds_at(lofit, bo)) {
            const uint32_t slot = lds_at(lox, bo) + q;
            Pair pr = {a, b2};
            *reinterpret_cast<Pair *>(pvals + slot * (uint32_t)sizeof(Pair)) = pr;
            *reinterpret_cast<unsigned short *>(prows + slot * 2u) = (unsigned short)k;
        } else {
            const uint32_t sp = lds_at(lsp, bo) + q;
            const uint32_t rr = row_of_local(k & (kBucketRows - 1), k >> 18, MODE == 2), cd = (k >> kBucketRowsLog2) & 7u;
            if (sp < plan.spill_cap) {
                SpillEntry<T> e = {rr | (cd << 29), a, b2};
                *reinterpret_cast<SpillEntry<T> *>(spill + sp * (uint32_t)sizeof(SpillEntry<T>)) = e;
            } else {
                // The spill list is full too (it holds 1/16 of a level's worst case): these entries go straight into the
                // table with device atomics — always correct, slow, and the one place where a sum depends on arrival order.
                T *gt = grad_table + (size_t)lv.offset * 2;
                atomic_add_pair(gt + (size_t)rr * 2, (float)a[0], (float)a[1]);
                if (cd != kCodeSingle)
                    atomic_add_pair(gt + (size_t)pair_row(rr, cd, MODE == 1) * 2, (float)b2[0], (float)b2[1]);
            }
        }
    };
    if (!fits)
        for (uint32_t q = tid; q < total; q += NTHREADS) to_global(q, skey[q], sa[q], sb[q]);
    if (pad_unstaged)  // (first wave: the padding entry of a bucket whose segment ends beyond the staging area)
        to_global(rs_st + rs_n0 - 1u, (kCodeSingle << kBucketRowsLog2) | (lane << 18), make_v2<T>(0.0f, 0.0f),
                  make_v2<T>(0.0f, 0.0f));
    if (!all_staged) {
        if (emit) {
#pragma unroll
            for (int p = 0; p < NP; p++)
                if (pos[p] >= (uint32_t)CAP)
                    to_global(pos[p], key_of(r0[p], MODE == 1 ? codesh : (single[p] ? kCodeSingle << kBucketRowsLog2 : 0u)),
                              val[2 * p], val[2 * p + 1]);
        }
        if (wave_singles && emit) {
#pragma unroll
            for (int p = 0; p < NP; p++)
                if (single[p] && pos_x[p] >= (uint32_t)CAP)
                    to_global(pos_x[p], key_of(MODE == 1 ? (r0[p] ^ xm) : r0[p] + 1u, kCodeSingle << kBucketRowsLog2),
                              val[2 * p + 1], make_v2<T>(0.0f, 0.0f));
        }
    }
    };
#ifdef LNH_ONLY_MODE  // tools/isa_sections.py: one body per census
    body(std::integral_constant<int, LNH_ONLY_MODE>{});
#else
    if (cls_hash) body(std::integral_constant<int, 1>{});
    else body(std::integral_constant<int, 2>{});
#endif
}

// A bucket with many entries (coarse dense levels: every ray passes the same few cells; very large batches) is split
// over up to kMaxSlices workgroups (blockIdx.y).  An unsplit bucket owns its rows and adds its image into the table with
// plain stores; the slices of a split bucket write their 64-bit integer images to the workspace and the slice that
// arrives last adds them up — integers, so the sum does not depend on the order.
// Entries per slice: 512 K keeps every hashed bucket whole up to 4 M points (a chunk).  Measured in round 4 with everything
// else as it is: 384 K the same, 256 K +4 %, 192 K and below 3.5 x (every hashed bucket split, its images through HBM).
// With the dense levels' rows dealt to 64 buckets, slices occur for concentrated batches only; lnh_grid_backward_set_slice_
// entries lowers the length so that tests reach this path with small inputs.
constexpr uint32_t kSliceEntries = 512 * 1024;
constexpr uint32_t kMaxSlices = 16;
uint32_t g_slice_entries = kSliceEntries;  // process-wide (set before sizing the workspace)

__device__ __forceinline__ uint32_t slices_of(uint32_t n_tot, uint32_t per) {
    const uint32_t s = (n_tot + per - 1) / per;
    return s > kMaxSlices ? kMaxSlices : s;
}
// Work items of pass 2 in dispatch order (blockIdx.x): first the EXTRA slices (slice 1.. of every split bucket — known
// only on the device, so the launch carries the host-side upper bound `n_extra` of them and the surplus workgroups
// exit after one look at the cursors), then slice 0 of every bucket of the window, longest first: buckets of the dense
// levels (few rows near the sensor take most entries -> same-address LDS adds), then the hashed levels from the finest
// (no run-merge, most entries) to the coarsest.  The dispatcher hands a free CU the next workgroup in that order, i.e.
// longest-processing-time-first list scheduling; with the plain (bucket, slice) grid the slices of the split buckets
// were dispatched behind everything else and 15 of 16 workgroups were empty, each waiting for a CU with 128 KiB of
// free LDS only to exit.
struct ReduceOrder {
    uint32_t level[LNH_MAX_LEVELS];      // levels of the window in processing order
    uint32_t first[LNH_MAX_LEVELS + 1];  // prefix sum of their bucket counts
    uint32_t n_levels, n_extra, bucket0, n_buckets;  // window: buckets [bucket0, bucket0 + n_buckets)
    uint32_t slice_entries;
};

template <typename T>
__global__ void __launch_bounds__(1024)
k_grid_bwd_reduce(T *__restrict__ grad_table, GridMeta meta, BucketPlan plan, const char *__restrict__ pool_bytes,
                  const uint32_t *__restrict__ cursor, const uint32_t *__restrict__ spill_cursor,
                  uint32_t *__restrict__ done, uint32_t L, ReduceOrder ord, uint32_t table_zero) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(smem_raw);  // [kBucketRows][2] fixed point
    __shared__ uint32_t sh_sum, sh_item[4], sh_wave[2][16];
    constexpr int K = sizeof(T) == 2 ? 24 : 40;  // scale of the fixed-point image (to_fixed)
    typedef v2_t<T> V2;
    // ---- which (bucket, slice) is this workgroup?
    const bool is_extra = blockIdx.x < ord.n_extra;
    uint32_t bid = ord.bucket0, slice = 0, slot0 = 0;
    if (!is_extra) {
        const uint32_t j = blockIdx.x - ord.n_extra;
        uint32_t k = 0;
        for (uint32_t q = 1; q < ord.n_levels; q++)
            if (j >= ord.first[q]) k = q;
        bid = plan.first_bucket[ord.level[k]] + (j - ord.first[k]);
    }
    if (is_extra || slices_of(cursor[bid], ord.slice_entries) > 1) {  // workgroup-uniform
        // block-wide exclusive scan over the window's buckets: extra slices before a bucket (-> which bucket an extra
        // workgroup serves) and image slots before it (-> where a split bucket keeps its slice images)
        const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
        const uint32_t sl = t < ord.n_buckets ? slices_of(cursor[ord.bucket0 + t], ord.slice_entries) : 0u;
        const uint32_t ex = sl > 1 ? sl - 1 : 0u, im = sl > 1 ? sl : 0u;
        const uint32_t in_ex = wave_scan_add_u32(ex), in_im = wave_scan_add_u32(im);
        if (lane == 63) {
            sh_wave[0][wv] = in_ex;
            sh_wave[1][wv] = in_im;
        }
        if (t == 0) sh_item[0] = 0;
        __syncthreads();
        uint32_t ex0 = in_ex - ex, im0 = in_im - im;
        for (uint32_t w = 0; w < wv; w++) {
            ex0 += sh_wave[0][w];
            im0 += sh_wave[1][w];
        }
        if (is_extra ? (ex && blockIdx.x >= ex0 && blockIdx.x < ex0 + ex) : (ord.bucket0 + t == bid)) {
            sh_item[0] = 1;
            sh_item[1] = ord.bucket0 + t;
            sh_item[2] = is_extra ? blockIdx.x - ex0 + 1 : 0u;
            sh_item[3] = im0;
        }
        __syncthreads();
        if (!sh_item[0]) return;  // surplus extra workgroup
        bid = sh_item[1];
        slice = sh_item[2];
        slot0 = sh_item[3];
    }
    uint32_t level = 0;
    for (uint32_t l = 0; l < L; l++)
        if (bid >= plan.first_bucket[l]) level = l;
    const LevelParams lv = meta.lv[level];
    const bool hashed = (lv.flags & (LV_HASH | LV_POW2)) == (LV_HASH | LV_POW2);  // pair rule of the level (pair_row)
    const uint32_t bk = bid - plan.first_bucket[level], cap = plan.cap[level];
    const uint32_t n_tot = cursor[bid];          // every entry reserved for this bucket, pool + spill
    const uint32_t n_all = min(n_tot, cap);      // ... of which in the pool
    const uint32_t slices = slices_of(n_tot, ord.slice_entries);
    if (slice >= slices) return;  // workgroup-uniform (n_tot == 0: nothing to add)
    const uint32_t i_begin = (uint32_t)((uint64_t)n_all * slice / slices);
    const uint32_t i_end = (uint32_t)((uint64_t)n_all * (slice + 1) / slices);
    const uint32_t n = i_end - i_begin;
    const bool il = (plan.interleaved >> level) & 1u;  // rows dealt to the buckets in 128-row groups (bucket_of_row)
    // table row of position r of this bucket's image (>= hashmap_size: no such row)
    auto table_row = [&](uint32_t r) { return row_of_local(r, bk, il); };
    // channel-planar image: acc[row] | acc[kBucketRows + row].  With the two channels of a row interleaved, each
    // 64-bit LDS add of a wave would touch only every other pair of banks and pay twice the conflicts; planar, the 64
    // random rows of an instruction spread over all banks.
    //
    // Element of the image: 64-bit FIXED POINT (ds_add_u64).  fp16 contributions are multiples of 2^-24 and enter exactly
    // (half_to_fixed24); fp32 ones are truncated to 2^-40 each.  Integer sums do not depend on the order of arrival.
    // (Measured and rejected in round 4: a DOUBLE image for fp16 tables — exact as well up to partial sums of 2^29, two
    // conversions per value instead of five VALU operations — runs the pass at 447 us against 330: ds_add_f64 is the slower
    // LDS atomic.  profiles/r04_reduce_levels.txt)
    typedef unsigned long long acc_t;
    acc_t *img = reinterpret_cast<acc_t *>(smem_raw);
    // (the image is cleared inside `stream`, AFTER the first pool loads have been issued: a workgroup's HBM round trip
    //  runs under its 128 KiB of LDS stores instead of behind them)
    auto clear_image = [&]() {
        for (uint32_t i = threadIdx.x; i < kBucketRows * 2; i += blockDim.x) acc[i] = 0ull;
        __syncthreads();
    };
    auto acc_add = [&](uint32_t idx, T v) {
        if constexpr (sizeof(T) == 2) atomicAdd(&img[idx], (unsigned long long)half_to_fixed24(v));
        else atomicAdd(&img[idx], (unsigned long long)(long long)ldexp((double)v, 40));
    };
    // No branch around the LDS adds: a single (code 7) adds ZERO for its absent second corner, to whatever row of the bucket
    // the pair rule yields.  (Per-entry `if`s cost an exec-mask round trip and a wait each; singles are one entry in 128.)
    auto add_entry = [&](auto hashed_c, uint32_t row0, uint32_t code, V2 a, V2 b) {
        constexpr bool HASHED = decltype(hashed_c)::value;
        const bool sgl = code == kCodeSingle;
        if constexpr (sizeof(T) == 2) {
            uint32_t wb = __builtin_bit_cast(uint32_t, b);
            wb = sgl ? 0u : wb;
            b = __builtin_bit_cast(V2, wb);
        } else {
            b[0] = sgl ? (T)0.0f : b[0];
            b[1] = sgl ? (T)0.0f : b[1];
        }
        const uint32_t row1 = (HASHED ? row0 ^ ((2u << code) - 1u) : row0 + 1u) & (kBucketRows - 1);
        acc_add(row0, a[0]);
        acc_add(kBucketRows + row0, a[1]);
        acc_add(row1, b[0]);
        acc_add(kBucketRows + row1, b[1]);
    };
    // One workgroup streams its whole slice [a_begin, a_end) of the level's slots.  The aligned QUADS inside it (4 entries =
    // 4 * sizeof(2 V2) bytes of values in 16-byte loads + 8 bytes of rows) take the main loop, with no per-entry range test;
    // the up to 3 + 3 entries in front of the first and behind the last full quad are added by the first threads, one each.
    auto stream = [&](auto hashed_c) {
        constexpr uint32_t UNROLL = sizeof(T) == 2 ? 2 : 1;  // (fp16: 1 / 2 / 4 quads in flight per lane measure the same)
        constexpr uint32_t VQ = sizeof(V2) * 2 * 4 / 16;     // 16-byte loads per quad: 2 (fp16) / 4 (fp32)
        const char *vbytes = pool_bytes + plan.pool_off[level], *rbytes = pool_bytes + plan.rows_off[level];
        const uint4_t *vals4 = reinterpret_cast<const uint4_t *>(vbytes);
        const uint2_t *rows4 = reinterpret_cast<const uint2_t *>(rbytes);
        const uint32_t a_begin = bk * cap + i_begin, a_end = a_begin + n;  // level-relative slots (< 2^32, checked)
        const uint32_t qf_begin = (a_begin + 3) >> 2, qf_end = a_end >> 2;
        const uint32_t nfull = qf_end > qf_begin ? qf_end - qf_begin : 0u;
        // every lane keeps UNROLL independent quads outstanding (non-temporal loads: the pool is written once and read
        // once), double-buffered: the loads of batch i+1 are in flight while the LDS adds of batch i execute
        const uint32_t stride = blockDim.x * UNROLL;
        uint4_t rv[2][UNROLL][VQ];
        uint2_t rr[2][UNROLL];
        auto fetch = [&](uint32_t j0, uint4_t (&v)[UNROLL][VQ], uint2_t (&r)[UNROLL]) {
#pragma unroll
            for (uint32_t u = 0; u < UNROLL; u++) {
                const uint32_t j = j0 + u * blockDim.x, q = qf_begin + (j < nfull ? j : nfull - 1);  // clamped, unconditional
#pragma unroll
                for (uint32_t w = 0; w < VQ; w++) v[u][w] = __builtin_nontemporal_load(vals4 + (size_t)q * VQ + w);
                r[u] = __builtin_nontemporal_load(rows4 + q);
            }
        };
        if (nfull) fetch(threadIdx.x, rv[0], rr[0]);
        clear_image();
        {
            const uint32_t lo_end = nfull ? qf_begin * 4 : a_end, hi_begin = nfull ? qf_end * 4 : a_end;
            const uint32_t n_lo = lo_end - a_begin, n_hi = a_end - hi_begin;
            if (threadIdx.x < n_lo + n_hi) {
                const uint32_t slot = threadIdx.x < n_lo ? a_begin + threadIdx.x : hi_begin + (threadIdx.x - n_lo);
                const V2 *pv = reinterpret_cast<const V2 *>(vbytes + (size_t)slot * (2 * sizeof(V2)));
                const uint32_t key = *reinterpret_cast<const unsigned short *>(rbytes + (size_t)slot * 2);
                add_entry(hashed_c, key & (kBucketRows - 1), key >> kBucketRowsLog2, pv[0], pv[1]);
            }
        }
        auto consume = [&](uint32_t j0, const uint4_t (&v)[UNROLL][VQ], const uint2_t (&r)[UNROLL]) {
#pragma unroll
            for (uint32_t u = 0; u < UNROLL; u++) {
                if (j0 + u * blockDim.x >= nfull) continue;  // (only in the last round of a lane: one test per quad)
                uint32_t words[VQ * 4];
#pragma unroll
                for (uint32_t w = 0; w < VQ; w++) {
                    words[4 * w] = v[u][w].x; words[4 * w + 1] = v[u][w].y;
                    words[4 * w + 2] = v[u][w].z; words[4 * w + 3] = v[u][w].w;
                }
                const uint32_t rw[2] = {r[u].x, r[u].y};
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    constexpr uint32_t EW = VQ;  // dwords per entry: 2 (fp16: a | b) / 4 (fp32: a0 a1 | b0 b1)
                    V2 a, b;
                    __builtin_memcpy(&a, &words[h * EW], sizeof(V2));
                    __builtin_memcpy(&b, &words[h * EW + EW / 2], sizeof(V2));
                    const uint32_t key = (rw[h >> 1] >> (16 * (h & 1))) & 0xffffu;
                    add_entry(hashed_c, key & (kBucketRows - 1), key >> kBucketRowsLog2, a, b);
                }
            }
        };
        uint32_t j0 = threadIdx.x;
        while (j0 < nfull) {  // unrolled by two so that the buffer index is a compile-time constant
            const uint32_t j1 = j0 + stride;
            if (j1 < nfull) fetch(j1, rv[1], rr[1]);
            consume(j0, rv[0], rr[0]);
            if (j1 >= nfull) break;
            const uint32_t j2 = j1 + stride;
            if (j2 < nfull) fetch(j2, rv[0], rr[0]);
            consume(j1, rv[1], rr[1]);
            j0 = j2;
        }
        // ---- this bucket overflowed its pool: its remaining entries sit somewhere in the level's spill list (among those
        //      of the level's other overflowing buckets).  Each slice filters its share of the list.
        if (n_tot > cap) {  // workgroup-uniform
            const SpillEntry<T> *spill = reinterpret_cast<const SpillEntry<T> *>(pool_bytes + plan.spill_off[level]);
            const uint32_t s_all = min(spill_cursor[level], plan.spill_cap);
            const uint32_t s_begin = (uint32_t)((uint64_t)s_all * slice / slices);
            const uint32_t s_end = (uint32_t)((uint64_t)s_all * (slice + 1) / slices);
            constexpr uint32_t SU = 4;
            for (uint32_t i0 = s_begin + threadIdx.x; i0 < s_end; i0 += blockDim.x * SU) {
                SpillEntry<T> e[SU];
#pragma unroll
                for (uint32_t u = 0; u < SU; u++) {
                    const uint32_t i = i0 + u * blockDim.x;
                    e[u] = spill[i < s_end ? i : s_end - 1];
                }
#pragma unroll
                for (uint32_t u = 0; u < SU; u++) {
                    const uint32_t i = i0 + u * blockDim.x;
                    const uint32_t r = e[u].key & 0x7ffffu;
                    if (i < s_end && bucket_of_row(r, il) == bk)
                        add_entry(hashed_c, local_of_row(r, il), e[u].key >> 29, e[u].a, e[u].b);
                }
            }
        }
    };
    if (hashed) stream(std::true_type{});
    else stream(std::false_type{});
    __syncthreads();
    // image element -> float: the exact integer sum * 2^-K, rounded once
    auto to_float = [&](acc_t q) -> float { return (float)ldexp((double)(long long)q, -K); };
    if (slices == 1) {
        // this workgroup owns the bucket's rows: table += image.  All row loads of a thread are issued before the first
        // use (a load -> add -> store chain per row would cost a memory round trip per row: 8 in a row per thread)
        T *gt = grad_table + (size_t)lv.offset * 2;
        constexpr uint32_t RPT = kBucketRows / 1024;
        Vec<T, 2> cur[RPT];
        // table_zero (workgroup-uniform): the caller has just cleared the gradient and this is its first chunk — the rows
        // hold zeros, nothing to read (27 MB per step and the round trip at the end of every workgroup).  Not so on a level
        // whose spill list ran full: the scatter pass has then added entries straight into the table with device atomics
        // (its last resort), and those rows must be read like any others.
        const bool rows_are_zero = table_zero && spill_cursor[level] <= plan.spill_cap;
#pragma unroll
        for (uint32_t i = 0; i < RPT; i++) {
            const uint32_t row = table_row(threadIdx.x + i * 1024u);
            if (rows_are_zero) {
                cur[i].v[0] = (T)0.0f;
                cur[i].v[1] = (T)0.0f;
            } else {
                cur[i] = load_vec<T, 2>(gt + 2 * (size_t)(row < lv.hashmap_size ? row : 0u));
            }
        }
#pragma unroll
        for (uint32_t i = 0; i < RPT; i++) {
            const uint32_t r = threadIdx.x + i * 1024u, row = table_row(r);
            const acc_t qa = img[r], qb = img[kBucketRows + r];
            if (row < lv.hashmap_size && (qa != (acc_t)0 || qb != (acc_t)0)) {
                cur[i].v[0] = (T)((float)cur[i].v[0] + to_float(qa));
                cur[i].v[1] = (T)((float)cur[i].v[1] + to_float(qb));
                store_vec<T, 2>(gt + 2 * (size_t)row, cur[i]);
            }
        }
    } else {
        // slice image -> workspace (coalesced 16-byte stores of the whole image); the slice that finishes LAST adds the
        // images up (exact sums: the result does not depend on which slice that is) and adds the rows into the table
        if (slot0 + slices > plan.partial_slots) return;  // (never: plan_buckets sizes the region for the worst case)
        char *images = const_cast<char *>(pool_bytes) + plan.partial_off;
        uint4 *dst = reinterpret_cast<uint4 *>(images) + (size_t)(slot0 + slice) * kBucketRows;
        const uint4 *src = reinterpret_cast<const uint4 *>(acc);
        for (uint32_t i = threadIdx.x; i < kBucketRows; i += blockDim.x) dst[i] = src[i];
        __threadfence();  // release (agent scope): this slice's image is visible before its arrival is
        __syncthreads();
        if (threadIdx.x == 0) sh_sum = atomicAdd(&done[bid], 1u);
        __syncthreads();
        if (sh_sum != slices - 1) return;  // workgroup-uniform
        __threadfence();  // acquire: the other slices' images (written by other CUs / XCDs) are read from memory
        const acc_t *simg = reinterpret_cast<const acc_t *>(images) + (size_t)slot0 * kBucketRows * 2;
        T *gt = grad_table + (size_t)lv.offset * 2;
        for (uint32_t r = threadIdx.x; r < kBucketRows; r += blockDim.x) {
            const uint32_t row = table_row(r);
            if (row >= lv.hashmap_size) continue;
            acc_t qa = (acc_t)0, qb = (acc_t)0;
            for (uint32_t sl = 0; sl < slices; sl++) {
                qa += simg[(size_t)sl * kBucketRows * 2 + r];
                qb += simg[(size_t)sl * kBucketRows * 2 + kBucketRows + r];
            }
            if (qa != (acc_t)0 || qb != (acc_t)0) {
                Vec<T, 2> cur = load_vec<T, 2>(gt + 2 * (size_t)row);
                cur.v[0] = (T)((float)cur.v[0] + to_float(qa));
                cur.v[1] = (T)((float)cur.v[1] + to_float(qb));
                store_vec<T, 2>(gt + 2 * (size_t)row, cur);
            }
        }
    }
}

// Host: bucket layout + pool sizing.  Returns the number of bytes of workspace needed:
//   [cursor: one u32 per bucket | spill cursor: one u32 per level | slice arrival counter: one u32 per bucket]
//                                                                      (zeroed by every launch)
//   [pool of every level: value stream | row stream] [spill list of every level] [slice images]
// Pool slots per bucket = even split of the level's expected entries (pair format: B * 2^D / 2, + 1/64 for the pairs
// that travel as two singles; generic level classes: B * 2^D) + 12.5 % + 12 sigma of a Poisson count (hashed levels are
// statistically even; correlated corners of neighbouring samples widen the spread, so the margin is generous) —
// whatever does not fit goes to the level's spill list (1/16 of the level's worst case B * 2^D), and beyond that into the
// table with device atomics.
constexpr uint32_t kCursorAlign = 256;
template <typename T>
uint64_t plan_buckets(BucketPlan &plan, const GridMeta &m, uint32_t L, uint32_t B, uint32_t D, bool plain,
                      uint32_t &total_buckets) {
    const uint64_t worst = (uint64_t)B << D;  // worst-case entries of one level (all singles)
    uint64_t bytes = 0;
    uint32_t nbt = 0;
    plan.interleaved = 0;
    for (uint32_t l = 0; l < L; l++) {
        // dense plain levels (k_grid_bwd_scatter_plain's MODE 2): 128-row groups dealt to up to 64 buckets
        const bool il = plain && !(m.lv[l].flags & LV_HASH) && (m.lv[l].flags & LV_NOWRAP) && (m.lv[l].flags & 15u) == D;
        if (il) plan.interleaved |= 1u << l;
        const uint32_t groups = (m.lv[l].hashmap_size + (1u << kGroupRowsLog2) - 1) >> kGroupRowsLog2;
        const uint32_t nb = il ? std::min<uint32_t>(groups, kMaxBucketsPerLevel)
                               : (m.lv[l].hashmap_size + kBucketRows - 1) / kBucketRows;
        plan.first_bucket[l] = nbt;
        const uint64_t expect = level_is_generic(m.lv[l], D, plain) ? worst : worst / 2 + worst / 128;
        const uint64_t mean = (expect + nb - 1) / nb;
        uint64_t cap = mean + mean / 8 + (uint64_t)(12.0 * sqrt((double)mean)) + 64;
        if (cap > worst) cap = worst;
        if (cap > 0xffffffffull) cap = 0xffffffffull;
        cap &= ~1ull;  // even: with even reservations (k_grid_bwd_scatter_plain) every slot PAIR of a workgroup stays inside a pool
        plan.cap[l] = (uint32_t)cap;
        const uint64_t slots = cap * nb;
        // value stream (2 V2 per slot) | row stream (2 B per slot), each padded so that aligned quad reads stay inside
        plan.pool_off[l] = bytes;
        const uint64_t vals_bytes = (slots * 2 * sizeof(v2_t<T>) + 64 + 15) / 16 * 16;
        plan.rows_off[l] = bytes + vals_bytes;
        bytes += vals_bytes + (slots * 2 + 8 + 15) / 16 * 16;
        nbt += nb;
    }
    plan.first_bucket[L] = nbt;
    total_buckets = nbt;
    // spill list of a level: 1/16 of its worst case (every entry of the level overflowing), at least 1 M entries (or the worst case itself: small batches never reach the atomics); what
    // does not fit there either is added with device atomics by the scatter pass itself (see to_global)
    uint64_t spill_cap = std::max<uint64_t>(worst / 16, 1u << 20);
    if (spill_cap > worst) spill_cap = worst;
    plan.spill_cap = spill_cap > 0xffffffffull ? 0xffffffffu : (uint32_t)spill_cap;
    for (uint32_t l = 0; l < L; l++) {
        plan.spill_off[l] = bytes;
        bytes += ((uint64_t)plan.spill_cap * sizeof(SpillEntry<T>) + 15) / 16 * 16;
    }
    // slice images: a split bucket has n > S entries (S = slice length of its level) and ceil(n / S) < 2 n / S slices, so
    // the split buckets of a level together have fewer than 2 * (entries of the level) / S of them
    uint64_t slots = 2 * (worst * L) / g_slice_entries + 1;
    if (slots > (uint64_t)nbt * kMaxSlices) slots = (uint64_t)nbt * kMaxSlices;
    plan.partial_slots = (uint32_t)slots;
    plan.partial_off = bytes;
    bytes += slots * kBucketRows * 2 * sizeof(unsigned long long);
    const uint64_t cursor_bytes = ((uint64_t)(2 * nbt + L) * 4 + kCursorAlign - 1) / kCursorAlign * kCursorAlign;
    return cursor_bytes + bytes;
}

// 128 KiB of dynamic LDS needs an opt-in per kernel and DEVICE (a process may drive several devices)
template <typename K>
void allow_big_lds(K kernel, size_t lds) {
    static unsigned long long done = 0;  // bit per device ordinal; the attribute call is idempotent, so a race is harmless
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(__atomic_load_n(&done, __ATOMIC_RELAXED) & bit)) {
        (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        __atomic_fetch_or(&done, bit, __ATOMIC_RELAXED);
    }
}

// Large batches are walked in chunks of at most kChunkPoints points (scatter + reduce per chunk, the table accumulates):
// up to 4 M points a bucket of a hashed level stays within one reduce slice (4 M * 8 / 64 buckets = 512 K entries);
// beyond that every bucket would be cut into slices whose images travel through HBM (measured at 16 384 rays x 832:
// 6.5 ms in one piece vs 4 x 1.2 ms in chunks), and the workspace would grow with the batch.  The sum stays a
// fixed-order, bit-reproducible one: integer per chunk, chunks added to the table in order.
constexpr uint32_t kChunkPoints = 4u << 20;
// `plain` = linear interpolation, align_corners off (the x-pair pool format: 4 entries per point and level); the generic
// level classes emit 8 singles per point and level, so their chunks are half as long — both fit the SAME workspace, whose
// size does not depend on an interpolation mode the sizing call does not know.
__host__ inline uint32_t chunk_points(uint32_t B, bool plain = true) {
    uint32_t n = div_up(B, kChunkPoints);
    if (!plain && B > 65536) n *= 2;
    return n <= 1 ? B : div_up(div_up(B, n), 1024) * 1024;
}

template <typename T>
int launch_backward_bucketed_chunk(const T *grad, const float *inputs, T *ge, uint32_t B, uint32_t L, const GridMeta &m,
                                   uint32_t align, uint32_t interp, void *workspace, uint64_t workspace_bytes,
                                   hipStream_t s, uint32_t level_begin, uint32_t level_end, uint32_t b_begin,
                                   uint32_t B_all, uint32_t B_plan, int phase = 0);

// Shortest chunk the call will use for a batch (256 K points, or the batch itself), and the longest chunk whose plan fits
// `workspace_bytes` (0: not even the shortest does).
constexpr uint32_t kMinChunkPoints = 256u << 10;
__host__ inline uint32_t halve_chunk(uint32_t B, uint32_t step) {
    return div_up(div_up(B, div_up(B, step) * 2), 1024) * 1024;
}
__host__ inline uint32_t min_chunk_points(uint32_t B, bool plain) {
    uint32_t step = chunk_points(B, plain);
    while (step > kMinChunkPoints) step = halve_chunk(B, step);
    return step;
}
template <typename T>
uint32_t fit_chunk(uint32_t B, const GridMeta &m, uint32_t L, bool plain, uint64_t workspace_bytes) {
    BucketPlan plan;
    uint32_t nbt = 0;
    for (uint32_t step = chunk_points(B, plain);; step = halve_chunk(B, step)) {
        if (plan_buckets<T>(plan, m, L, step, 3, plain, nbt) <= workspace_bytes) return step;
        if (step <= kMinChunkPoints) return 0;
    }
}

// split = 0: the whole backward of the levels [level_begin, level_end).
// split = 1 ("begin"): every chunk but the last completely (all levels), then the SCATTER pass of the last chunk.
// split = 2 ("finish"): the REDUCE pass of the last chunk for the levels [level_begin, level_end) — after it the gradient of
//           those levels is final.  A data-parallel caller runs begin once and finish per level window, handing each window
//           to the all-reduce while the next is being reduced: the scatter pass stays in one piece (cut into level windows it
//           loses its level-fastest interleave: 2 windows cost 444 + 453 us against 707 us, profiles/r03_bwd_pipeline.txt).
template <typename T>
int launch_backward_bucketed(const T *grad, const float *inputs, T *ge, uint32_t B, uint32_t L, const GridMeta &m,
                             uint32_t align, uint32_t interp, void *workspace, uint64_t workspace_bytes,
                             hipStream_t s, uint32_t level_begin = 0, uint32_t level_end = 0xffffffffu, int split = 0,
                             uint32_t flags = 0) {
    // flags (lnh_grid_encode_backward_ws_ex): LNH_BWD_WS_CLEARED — the head of the workspace is zero on entry (serves the
    // FIRST chunk's scatter pass); LNH_BWD_TABLE_ZERO — grad_embeddings is zero on entry (serves the first chunk's reduce pass)
    if (level_end > L) level_end = L;
    if (level_begin >= level_end) return LNH_OK;
    // The workspace serves one chunk at a time, so a SMALLER workspace than lnh_grid_backward_workspace_size() asks for is
    // not an error: the batch is walked in shorter chunks (fit_chunk halves the chunk until its plan fits; 3.09 GB / 1.60 /
    // 0.91 GB at 4096 rays x 832 cost 897 / 922 / 992 us, profiles/r04_workspace_chunks.txt).  Below the plan of the
    // shortest chunk the call fails and names that size.
    const uint32_t step = fit_chunk<T>(B, m, L, align == 0 && interp == 0, workspace_bytes);
    if (step == 0) {
        BucketPlan plan;
        uint32_t nbt = 0;
        const uint64_t least = plan_buckets<T>(plan, m, L, min_chunk_points(B, align == 0 && interp == 0), 3,
                                               align == 0 && interp == 0, nbt);
        lnh_set_error("grid backward: workspace too small (%llu bytes; this batch needs at least %llu — "
                      "lnh_grid_backward_workspace_size_min — and runs fastest with lnh_grid_backward_workspace_size)",
                      (unsigned long long)workspace_bytes, (unsigned long long)least);
        return LNH_ERR_INVALID_ARG;
    }
    for (uint32_t b0 = 0; b0 < B; b0 += step) {
        const bool last = b0 + step >= B;
        int phase = 0;
        uint32_t l0 = level_begin, l1 = level_end;
        if (split == 1) {
            l0 = 0;
            l1 = L;
            phase = last ? 1 : 0;
        } else if (split == 2) {
            if (!last) continue;
            phase = 2;
        }
        const bool first = b0 == 0;
        const int rc = launch_backward_bucketed_chunk<T>(grad, inputs, ge, std::min(step, B - b0), L, m, align, interp,
                                                         workspace, workspace_bytes, s, l0, l1, b0, B, step, phase,
                                                         first && (flags & LNH_BWD_WS_CLEARED),
                                                         first && (flags & LNH_BWD_TABLE_ZERO));
        if (rc) return rc;
    }
    return LNH_OK;
}

template <typename T>
int launch_backward_bucketed_chunk(const T *grad, const float *inputs, T *ge, uint32_t B, uint32_t L, const GridMeta &m,
                                   uint32_t align, uint32_t interp, void *workspace, uint64_t workspace_bytes,
                                   hipStream_t s, uint32_t level_begin, uint32_t level_end, uint32_t b_begin,
                                   uint32_t B_all, uint32_t B_plan, int phase, bool cursors_cleared, bool table_zero) {
    // phase 0: scatter + reduce; 1: (zeroed cursors +) scatter only; 2: reduce only, of a scatter an earlier call has run
    // cursors_cleared: the caller has zeroed the head of the workspace (lnh_grid_backward_workspace_clear_bytes) for this
    // chunk; table_zero: grad_table holds zeros — the reduce pass stores its sums instead of adding them to what it reads
    BucketPlan plan;
    uint32_t nbt = 0;
    const bool plain = align == 0 && interp == 0;
    const uint64_t need = plan_buckets<T>(plan, m, L, B_plan, 3, plain, nbt);  // (a full chunk's plan serves the last, shorter one)
    if (workspace == nullptr || workspace_bytes < need) {
        lnh_set_error("grid backward: workspace too small (%llu < %llu bytes)", (unsigned long long)workspace_bytes,
                      (unsigned long long)need);
        return LNH_ERR_INVALID_ARG;
    }
    if (((uint64_t)B << 3) > 0xffffffffull) {
        lnh_set_error("grid backward (bucketed): B = %u is too large for 32-bit pool slots", B);
        return LNH_ERR_UNSUPPORTED;
    }
    for (uint32_t l = 0; l < L; l++)
        if ((uint64_t)plan.cap[l] * (plan.first_bucket[l + 1] - plan.first_bucket[l]) > 0xffffffffull) {
            lnh_set_error("grid backward (bucketed): B = %u is too large for 32-bit pool slots", B);
            return LNH_ERR_UNSUPPORTED;
        }
    for (uint32_t l = 0; l < L; l++)
        if (plan.first_bucket[l + 1] - plan.first_bucket[l] > kMaxBucketsPerLevel) {
            lnh_set_error("grid backward (bucketed): level %u has more than %u buckets", l, kMaxBucketsPerLevel);
            return LNH_ERR_UNSUPPORTED;
        }
    const uint64_t cursor_bytes = ((uint64_t)(2 * nbt + L) * 4 + kCursorAlign - 1) / kCursorAlign * kCursorAlign;
    uint32_t *cursor = reinterpret_cast<uint32_t *>(workspace);
    uint32_t *spill_cursor = cursor + nbt, *done = spill_cursor + L;
    char *pool = reinterpret_cast<char *>(workspace) + cursor_bytes;
    if (phase != 2 && !cursors_cleared) {
        const int zrc = lnh_zero_async(cursor, cursor_bytes, s, "grid backward (cursor clear)");
        if (zrc != LNH_OK) return zrc;
    }
    // 1024 threads x 1 point: the per-workgroup cost that matters is the one returning device atomic per touched
    // bucket (measured: 256- and 512-thread workgroups are 2.3x / 1.5x slower); 5 staged pair entries per thread keep
    // the LDS staging at 60 KiB (two workgroups per CU).  A window with a generic-class level (8 singles per point)
    // takes the instantiation with the larger staging area.
    const uint32_t n_win = level_end - level_begin;
    bool generic = false, any_plain = false;
    for (uint32_t l = level_begin; l < level_end; l++) {
        const bool g = level_is_generic(m.lv[l], 3, plain);
        generic |= g;
        any_plain |= !g;
    }
    if (phase != 2) {
        // the plain classes' levels and the generic ones are served by their own kernel (a workgroup of the other kernel's
        // level exits at once); the usual configuration has plain levels only
        if (any_plain)
            LNH_LAUNCH((k_grid_bwd_scatter_plain<T, kScatterThreads, 5 * kScatterThreads>), dim3(n_win, div_up(B, kScatterThreads)),
                       dim3(kScatterThreads), 0, s, grad, inputs, B, m, plan, pool, cursor, spill_cursor, level_begin, b_begin,
                       B_all, ge);
        if (generic)
            LNH_LAUNCH((k_grid_bwd_scatter<T, 3, 6144>), dim3(div_up(B, 1024) * n_win), dim3(1024), 0, s, grad, inputs, B,
                       m, plan, pool, cursor, spill_cursor, align, interp, n_win, level_begin, b_begin, B_all, ge);
    }
    int rc = lnh_check_launch("lnh_grid_encode_backward_ws(scatter)");
    if (rc || phase == 1) return rc;
    auto k = k_grid_bwd_reduce<T>;
    const size_t lds = (size_t)kBucketRows * 2 * sizeof(unsigned long long);
    allow_big_lds(k, lds);
    ReduceOrder ord;
    ord.bucket0 = plan.first_bucket[level_begin];
    ord.n_buckets = plan.first_bucket[level_end] - ord.bucket0;
    ord.n_levels = 0;
    ord.first[0] = 0;
    auto push = [&](uint32_t l) {
        ord.level[ord.n_levels] = l;
        ord.first[ord.n_levels + 1] = ord.first[ord.n_levels] + plan.first_bucket[l + 1] - plan.first_bucket[l];
        ord.n_levels++;
    };
    for (uint32_t l = level_begin; l < level_end; l++)
        if (!(m.lv[l].flags & LV_HASH)) push(l);
    for (uint32_t l = level_end; l-- > level_begin;)
        if (m.lv[l].flags & LV_HASH) push(l);
    // slices beyond the first: sum over buckets of ceil(n / kSliceEntries) - 1 <= (entries of the window) / kSliceEntries
    const uint64_t extra = std::min<uint64_t>((uint64_t)ord.n_buckets * (kMaxSlices - 1),
                                              (((uint64_t)B << 3) * n_win) / g_slice_entries);
    ord.n_extra = (uint32_t)extra;
    ord.slice_entries = g_slice_entries;
    LNH_LAUNCH(k, dim3(ord.n_extra + ord.n_buckets), dim3(1024), lds, s, ge, m, plan, pool, cursor, spill_cursor, done, L,
               ord, table_zero ? 1u : 0u);
    return lnh_check_launch("lnh_grid_encode_backward_ws(reduce)");
}

// gridencoder.cu:364-390
template <typename T, int D, int C>
__global__ void k_grid_input_backward(const T *__restrict__ grad, const T *__restrict__ dy_dx,
                                      T *__restrict__ grad_inputs, uint32_t B, uint32_t L) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    T r = (T)0.0f;
    for (uint32_t l = 0; l < L; l++)
#pragma unroll
        for (int ch = 0; ch < C; ch++)
            r = (T)((float)r + (float)grad[((size_t)l * B + b) * C + ch] *
                                   (float)dy_dx[(((size_t)b * L + l) * D + d) * C + ch]);
    grad_inputs[t] = r;
}

// ------------------------------------------------------------------------------------------------ TV gradient
// gridencoder.cu:695-807
template <typename T, int D, int C>
__global__ void __launch_bounds__(256)
k_grad_tv(const T *__restrict__ inputs, const T *__restrict__ table, T *__restrict__ grad, float weight, uint32_t B,
          GridMeta meta, uint32_t align) {
    const uint32_t level = blockIdx.y;
    const LevelParams lv = meta.lv[level];
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float x[D];
#pragma unroll
    for (int d = 0; d < D; d++) x[d] = (float)inputs[(size_t)b * D + d];
    Cell<D> cell;
    if (!locate<D>(x, lv, align != 0, 0, cell)) return;
    const T *tab = table + (size_t)lv.offset * C;
    T *gt = grad + (size_t)lv.offset * C;
    // integer lattice position of the base corner and a row-index helper working on explicit coordinates
    uint32_t pg[D];
#pragma unroll
    for (int d = 0; d < D; d++) pg[d] = (uint32_t)floorf(fmaf(x[d], lv.scale, align ? 0.0f : 0.5f));
    auto row_of = [&](const uint32_t(&p)[D]) {
        const uint32_t R = align ? lv.resolution : lv.resolution + 1;
        const uint32_t nd = lv.flags & 15u;
        uint32_t idx = 0, stride = 1;
        if (lv.flags & LV_HASH) {
#pragma unroll
            for (int d = 0; d < D; d++) idx ^= p[d] * prime_of(d);
        } else {
#pragma unroll
            for (int d = 0; d < D; d++)
                if ((uint32_t)d < nd) {
                    idx += p[d] * stride;
                    stride *= R;
                }
        }
        return idx % lv.hashmap_size;
    };
    const uint32_t index = row_of(pg) * C;
    float results[C], idelta[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) results[ch] = idelta[ch] = 0.0f;
    const T w = (T)(weight / (2 * D));
#pragma unroll
    for (int d = 0; d < D; d++) {
        const uint32_t cur = pg[d];
        if (cur < lv.resolution) {
            pg[d] = cur + 1;
            const uint32_t ir = row_of(pg) * C;
#pragma unroll
            for (int ch = 0; ch < C; ch++) {
                T gv = (T)(tab[index + ch] - tab[ir + ch]);
                results[ch] = (float)(T)(results[ch] + (float)gv);
                idelta[ch] = (float)(T)(idelta[ch] + (float)(T)(gv * gv));
            }
        }
        if (cur > 0) {
            pg[d] = cur - 1;
            const uint32_t il = row_of(pg) * C;
#pragma unroll
            for (int ch = 0; ch < C; ch++) {
                T gv = (T)(tab[index + ch] - tab[il + ch]);
                results[ch] = (float)(T)(results[ch] + (float)gv);
                idelta[ch] = (float)(T)(idelta[ch] + (float)(T)(gv * gv));
            }
        }
        pg[d] = cur;
    }
#pragma unroll
    for (int ch = 0; ch < C; ch++) {
        // `w * results[ch]` is a scalar_t product in the reference (rounded to the table type), then a float product
        const float v = (float)(T)((float)w * results[ch]) * rsqrtf(idelta[ch] + 1e-9f);
        if constexpr (sizeof(T) == 4) {
            unsafeAtomicAdd((float *)gt + index + ch, v);
        } else {
            // scalar fp16 atomic through the aligned packed instruction: add (v,0) or (0,v)
            half_t *p = (half_t *)gt + index + ch;
            const bool hi = ((uintptr_t)p & 2) != 0;
            half_t *pa = hi ? p - 1 : p;
            atomic_add_pair(pa, hi ? 0.0f : v, hi ? v : 0.0f);
        }
    }
}

template <typename T, int D>
int launch_forward_c(const float *inputs, const T *emb, T *out, T *dy_dx, uint32_t B, uint32_t C, uint32_t L,
                     const GridMeta &m, uint32_t align, uint32_t interp, hipStream_t s, RowMap map = RowMap{0, 0, 0, 0}) {
    dim3 grid(div_up(B, 256), /* levels */ L), block(256);
#define LNH_FWD(CC)                                                                                               \
    if (dy_dx)                                                                                                    \
        LNH_LAUNCH((k_grid_forward<T, D, CC, true>), grid, block, 0, s, inputs, emb, out, dy_dx, B, L, m, \
                           align, interp, map);                                                                   \
    else                                                                                                          \
        LNH_LAUNCH((k_grid_forward<T, D, CC, false>), grid, block, 0, s, inputs, emb, out, dy_dx, B, L, m, \
                           align, interp, map);
    switch (C) {
        case 1: LNH_FWD(1) break;
        case 2: LNH_FWD(2) break;
        case 4: LNH_FWD(4) break;
        case 8: LNH_FWD(8) break;
        default: lnh_set_error("GridEncoding: C must be 1, 2, 4, or 8 (got %u)", C); return LNH_ERR_UNSUPPORTED;
    }
#undef LNH_FWD
    return lnh_check_launch("lnh_grid_encode_forward");
}


template <typename T, int D>
int launch_backward_c(const T *grad, const float *inputs, T *ge, uint32_t B, uint32_t C, uint32_t L,
                      const GridMeta &m, uint32_t align, uint32_t interp, hipStream_t s) {
    dim3 block(256);
    switch (C) {
        case 1:
            if constexpr (sizeof(T) == 2) {
                lnh_set_error("grid backward: fp16 tables need an even C (the reference forces fp32 when C is odd, "
                              "grid.py:54-57)");
                return LNH_ERR_UNSUPPORTED;
            } else {
                LNH_LAUNCH((k_grid_backward<T, D, 1, 1, true>), dim3(div_up(B, 256), L), block, 0, s, grad,
                                   inputs, ge, B, m, align, interp);
            }
            break;
        case 2:
            LNH_LAUNCH((k_grid_backward<T, D, 2, 2, true>), dim3(div_up(B, 256), L), block, 0, s, grad, inputs,
                               ge, B, m, align, interp);
            break;
        case 4:
            LNH_LAUNCH((k_grid_backward<T, D, 4, 2, false>), dim3(div_up((uint64_t)B * 2, 256), L), block, 0, s,
                               grad, inputs, ge, B, m, align, interp);
            break;
        case 8:
            LNH_LAUNCH((k_grid_backward<T, D, 8, 2, false>), dim3(div_up((uint64_t)B * 4, 256), L), block, 0, s,
                               grad, inputs, ge, B, m, align, interp);
            break;
        default: lnh_set_error("GridEncoding: C must be 1, 2, 4, or 8 (got %u)", C); return LNH_ERR_UNSUPPORTED;
    }
    return lnh_check_launch("lnh_grid_encode_backward");
}

template <typename T, int D>
int launch_input_backward_c(const T *grad, const T *dy_dx, T *gi, uint32_t B, uint32_t C, uint32_t L, hipStream_t s) {
    dim3 grid(div_up((uint64_t)B * D, 256)), block(256);
    switch (C) {
        case 1: LNH_LAUNCH((k_grid_input_backward<T, D, 1>), grid, block, 0, s, grad, dy_dx, gi, B, L); break;
        case 2: LNH_LAUNCH((k_grid_input_backward<T, D, 2>), grid, block, 0, s, grad, dy_dx, gi, B, L); break;
        case 4: LNH_LAUNCH((k_grid_input_backward<T, D, 4>), grid, block, 0, s, grad, dy_dx, gi, B, L); break;
        case 8: LNH_LAUNCH((k_grid_input_backward<T, D, 8>), grid, block, 0, s, grad, dy_dx, gi, B, L); break;
        default: return LNH_ERR_UNSUPPORTED;
    }
    return lnh_check_launch("lnh_grid_encode_backward(inputs)");
}

template <typename T, int D>
int launch_tv_c(const T *inputs, const T *emb, T *grad, float weight, uint32_t B, uint32_t C, uint32_t L,
                const GridMeta &m, uint32_t align, hipStream_t s) {
    dim3 grid(div_up(B, 256), L), block(256);
    switch (C) {
        case 1:
            if constexpr (sizeof(T) == 2) {
                lnh_set_error("grad_total_variation: fp16 needs an even C");
                return LNH_ERR_UNSUPPORTED;
            } else {
                LNH_LAUNCH((k_grad_tv<T, D, 1>), grid, block, 0, s, inputs, emb, grad, weight, B, m, align);
            }
            break;
        case 2: LNH_LAUNCH((k_grad_tv<T, D, 2>), grid, block, 0, s, inputs, emb, grad, weight, B, m, align); break;
        case 4: LNH_LAUNCH((k_grad_tv<T, D, 4>), grid, block, 0, s, inputs, emb, grad, weight, B, m, align); break;
        case 8: LNH_LAUNCH((k_grad_tv<T, D, 8>), grid, block, 0, s, inputs, emb, grad, weight, B, m, align); break;
        default: lnh_set_error("GridEncoding: C must be 1, 2, 4, or 8 (got %u)", C); return LNH_ERR_UNSUPPORTED;
    }
    return lnh_check_launch("lnh_grad_total_variation");
}

int check_common(const void *inputs, const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                 int dtype) {
    LNH_REQUIRE(inputs && offsets_host, LNH_ERR_INVALID_ARG, "grid: null inputs/offsets");
    LNH_REQUIRE(L >= 1 && L <= LNH_MAX_LEVELS, LNH_ERR_UNSUPPORTED, "grid: L must be in 1..%d (got %u)", LNH_MAX_LEVELS, L);
    LNH_REQUIRE(D >= 2 && D <= 5, LNH_ERR_UNSUPPORTED, "GridEncoding: D must be 2, 3, 4 or 5 (got %u)", D);
    LNH_REQUIRE(C == 1 || C == 2 || C == 4 || C == 8, LNH_ERR_UNSUPPORTED, "GridEncoding: C must be 1, 2, 4, or 8 (got %u)", C);
    LNH_REQUIRE(dtype == LNH_F32 || dtype == LNH_F16, LNH_ERR_UNSUPPORTED, "grid: dtype must be LNH_F32 or LNH_F16");
    (void)B;
    return LNH_OK;
}

}  // namespace

#define LNH_DISPATCH_D(D, CALL)                                   \
    switch (D) {                                                  \
        case 2: { constexpr int DD = 2; rc = CALL; } break;       \
        case 3: { constexpr int DD = 3; rc = CALL; } break;       \
        case 4: { constexpr int DD = 4; rc = CALL; } break;       \
        case 5: { constexpr int DD = 5; rc = CALL; } break;       \
        default: rc = LNH_ERR_UNSUPPORTED;                        \
    }

extern "C" {

int lnh_grid_encode_forward(const float *inputs, const void *embeddings, const int32_t *offsets_host, void *outputs,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void *dy_dx,
                            uint32_t gridtype, int align_corners, uint32_t interp, int dtype, lnh_stream_t stream) {
    int rc = check_common(inputs, offsets_host, B, D, C, L, dtype);
    if (rc) return rc;
    LNH_REQUIRE(embeddings && outputs, LNH_ERR_INVALID_ARG, "grid forward: null embeddings/outputs");
    if (B == 0) return LNH_OK;
    GridMeta m;
    LNH_REQUIRE(build_meta(m, offsets_host, D, L, S, H, gridtype, align_corners != 0) == 0, LNH_ERR_INVALID_ARG,
                "grid: offsets must be increasing and non-negative");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LNH_F32) {
        LNH_DISPATCH_D(D, (launch_forward_c<float, DD>(inputs, (const float *)embeddings, (float *)outputs,
                                                       (float *)dy_dx, B, C, L, m, align_corners != 0, interp, s)))
    } else {
        LNH_DISPATCH_D(D, (launch_forward_c<half_t, DD>(inputs, (const half_t *)embeddings, (half_t *)outputs,
                                                        (half_t *)dy_dx, B, C, L, m, align_corners != 0, interp, s)))
    }
    return rc;
}

int lnh_grid_encode_forward_mapped(const float *inputs_all, const void *embeddings, const int32_t *offsets_host,
                                   void *outputs_all, uint32_t B, uint32_t T_cur, uint32_t T_tot, uint32_t slot_off,
                                   uint32_t B_all, uint32_t C, uint32_t L, float S, uint32_t H, int dtype,
                                   lnh_stream_t stream) {
    int rc = check_common(inputs_all, offsets_host, B, 3, C, L, dtype);
    if (rc) return rc;
    LNH_REQUIRE(embeddings && outputs_all, LNH_ERR_INVALID_ARG, "grid forward (mapped): null embeddings/outputs");
    LNH_REQUIRE(T_cur >= 1 && slot_off + T_cur <= T_tot && B % T_cur == 0 && (uint64_t)(B / T_cur) * T_tot <= B_all,
                LNH_ERR_INVALID_ARG, "grid forward (mapped): inconsistent row map");
    if (B == 0) return LNH_OK;
    GridMeta m;
    LNH_REQUIRE(build_meta(m, offsets_host, 3, L, S, H, 0, false) == 0, LNH_ERR_INVALID_ARG,
                "grid: offsets must be increasing and non-negative");
    const RowMap map{T_cur, T_tot, slot_off, B_all};
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LNH_F32)
        return launch_forward_c<float, 3>(inputs_all, (const float *)embeddings, (float *)outputs_all, nullptr, B, C, L, m,
                                          0, 0, s, map);
    return launch_forward_c<half_t, 3>(inputs_all, (const half_t *)embeddings, (half_t *)outputs_all, nullptr, B, C, L, m, 0,
                                       0, s, map);
}

int lnh_grid_encode_backward(const void *grad, const float *inputs, const void *embeddings,
                             const int32_t *offsets_host, void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                             uint32_t L, float S, uint32_t H, const void *dy_dx, void *grad_inputs, uint32_t gridtype,
                             int align_corners, uint32_t interp, int dtype, lnh_stream_t stream) {
    (void)embeddings;
    int rc = check_common(inputs, offsets_host, B, D, C, L, dtype);
    if (rc) return rc;
    LNH_REQUIRE(grad && grad_embeddings, LNH_ERR_INVALID_ARG, "grid backward: null grad/grad_embeddings");
    LNH_REQUIRE((dy_dx == nullptr) == (grad_inputs == nullptr), LNH_ERR_INVALID_ARG,
                "grid backward: dy_dx and grad_inputs must be given together");
    if (B == 0) return LNH_OK;
    GridMeta m;
    LNH_REQUIRE(build_meta(m, offsets_host, D, L, S, H, gridtype, align_corners != 0) == 0, LNH_ERR_INVALID_ARG,
                "grid: offsets must be increasing and non-negative");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LNH_F32) {
        LNH_DISPATCH_D(D, (launch_backward_c<float, DD>((const float *)grad, inputs, (float *)grad_embeddings, B, C, L,
                                                        m, align_corners != 0, interp, s)))
        if (rc == LNH_OK && dy_dx)
            LNH_DISPATCH_D(D, (launch_input_backward_c<float, DD>((const float *)grad, (const float *)dy_dx,
                                                                  (float *)grad_inputs, B, C, L, s)))
    } else {
        LNH_DISPATCH_D(D, (launch_backward_c<half_t, DD>((const half_t *)grad, inputs, (half_t *)grad_embeddings, B, C,
                                                         L, m, align_corners != 0, interp, s)))
        if (rc == LNH_OK && dy_dx)
            LNH_DISPATCH_D(D, (launch_input_backward_c<half_t, DD>((const half_t *)grad, (const half_t *)dy_dx,
                                                                   (half_t *)grad_inputs, B, C, L, s)))
    }
    return rc;
}

uint64_t lnh_grid_backward_workspace_size(const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                          float S, uint32_t H, uint32_t gridtype, int align_corners, int dtype) {
    if (!offsets_host || L < 1 || L > LNH_MAX_LEVELS || D != 3 || C != 2 || B == 0) return 0;
    GridMeta m;
    if (build_meta(m, offsets_host, D, L, S, H, gridtype, align_corners != 0) != 0) return 0;
    BucketPlan plan;
    uint32_t nbt = 0;
    for (uint32_t l = 0; l < L; l++)
        if ((m.lv[l].hashmap_size + kBucketRows - 1) / kBucketRows > kMaxBucketsPerLevel) return 0;
    // the workspace serves one chunk at a time.  (The interpolation mode is not an argument here: size for the larger of the
    // two pool layouts it can select — each with ITS chunk length, see chunk_points.)
    uint64_t need = 0;
    for (int plain = 0; plain <= (align_corners ? 0 : 1); plain++) {
        const uint32_t Bc = chunk_points(B, plain != 0);
        const uint64_t n = dtype == LNH_F16 ? plan_buckets<half_t>(plan, m, L, Bc, D, plain != 0, nbt)
                                            : plan_buckets<float>(plan, m, L, Bc, D, plain != 0, nbt);
        need = n > need ? n : need;
    }
    return need;
}

uint64_t lnh_grid_backward_workspace_size_min(const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                              float S, uint32_t H, uint32_t gridtype, int align_corners, int dtype) {
    if (lnh_grid_backward_workspace_size(offsets_host, B, D, C, L, S, H, gridtype, align_corners, dtype) == 0) return 0;
    GridMeta m;
    (void)build_meta(m, offsets_host, D, L, S, H, gridtype, align_corners != 0);
    BucketPlan plan;
    uint32_t nbt = 0;
    uint64_t need = 0;
    for (int plain = 0; plain <= (align_corners ? 0 : 1); plain++) {
        const uint32_t Bc = min_chunk_points(B, plain != 0);
        const uint64_t n = dtype == LNH_F16 ? plan_buckets<half_t>(plan, m, L, Bc, D, plain != 0, nbt)
                                            : plan_buckets<float>(plan, m, L, Bc, D, plain != 0, nbt);
        need = n > need ? n : need;
    }
    return need;
}

void lnh_grid_backward_set_slice_entries(uint32_t entries) {
    g_slice_entries = entries == 0 ? kSliceEntries : std::min(std::max(entries, 1024u), kSliceEntries);
}

int lnh_grid_backward_plan_info(const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                uint32_t H, uint32_t gridtype, int align_corners, int dtype, uint32_t level,
                                uint32_t *out4) {
    LNH_REQUIRE(out4 && offsets_host, LNH_ERR_INVALID_ARG, "grid backward plan: null argument");
    LNH_REQUIRE(lnh_grid_backward_workspace_size(offsets_host, B, D, C, L, S, H, gridtype, align_corners, dtype) != 0 &&
                    level < L, LNH_ERR_UNSUPPORTED, "grid backward plan: configuration not served by the bucketed path");
    GridMeta m;
    (void)build_meta(m, offsets_host, D, L, S, H, gridtype, align_corners != 0);
    BucketPlan plan;
    uint32_t nbt = 0;
    if (dtype == LNH_F16) (void)plan_buckets<half_t>(plan, m, L, chunk_points(B, align_corners == 0), D, align_corners == 0, nbt);
    else (void)plan_buckets<float>(plan, m, L, chunk_points(B, align_corners == 0), D, align_corners == 0, nbt);
    out4[0] = plan.first_bucket[level + 1] - plan.first_bucket[level];
    out4[1] = plan.cap[level];
    out4[2] = kBucketRows;
    out4[3] = g_slice_entries;
    return LNH_OK;
}

int lnh_grid_encode_backward_ws(const void *grad, const float *inputs, const int32_t *offsets_host,
                                void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                void *workspace, uint64_t workspace_bytes, lnh_stream_t stream) {
    int rc = check_common(inputs, offsets_host, B, D, C, L, dtype);
    if (rc) return rc;
    LNH_REQUIRE(grad && grad_embeddings, LNH_ERR_INVALID_ARG, "grid backward: null grad/grad_embeddings");
    LNH_REQUIRE(D == 3 && C == 2, LNH_ERR_UNSUPPORTED,
                "grid backward (bucketed): only D == 3, C == 2 (use lnh_grid_encode_backward otherwise)");
    if (B == 0) return LNH_OK;
    GridMeta m;
    LNH_REQUIRE(build_meta(m, offsets_host, D, L, S, H, gridtype, align_corners != 0) == 0, LNH_ERR_INVALID_ARG,
                "grid: offsets must be increasing and non-negative");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LNH_F32)
        return launch_backward_bucketed<float>((const float *)grad, inputs, (float *)grad_embeddings, B, L, m,
                                               align_corners != 0, interp, workspace, workspace_bytes, s);
    return launch_backward_bucketed<half_t>((const half_t *)grad, inputs, (half_t *)grad_embeddings, B, L, m,
                                            align_corners != 0, interp, workspace, workspace_bytes, s);
}

int lnh_grid_encode_backward_ws_levels(const void *grad, const float *inputs, const int32_t *offsets_host,
                                       void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                       uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                       void *workspace, uint64_t workspace_bytes, uint32_t level_begin,
                                       uint32_t level_end, lnh_stream_t stream) {
    int rc = check_common(inputs, offsets_host, B, D, C, L, dtype);
    if (rc) return rc;
    LNH_REQUIRE(grad && grad_embeddings, LNH_ERR_INVALID_ARG, "grid backward: null grad/grad_embeddings");
    LNH_REQUIRE(D == 3 && C == 2, LNH_ERR_UNSUPPORTED,
                "grid backward (bucketed): only D == 3, C == 2 (use lnh_grid_encode_backward otherwise)");
    LNH_REQUIRE(level_begin <= level_end && level_end <= L, LNH_ERR_INVALID_ARG,
                "grid backward: need level_begin <= level_end <= L");
    if (B == 0) return LNH_OK;
    GridMeta m;
    LNH_REQUIRE(build_meta(m, offsets_host, D, L, S, H, gridtype, align_corners != 0) == 0, LNH_ERR_INVALID_ARG,
                "grid: offsets must be increasing and non-negative");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LNH_F32)
        return launch_backward_bucketed<float>((const float *)grad, inputs, (float *)grad_embeddings, B, L, m,
                                               align_corners != 0, interp, workspace, workspace_bytes, s, level_begin,
                                               level_end);
    return launch_backward_bucketed<half_t>((const half_t *)grad, inputs, (half_t *)grad_embeddings, B, L, m,
                                            align_corners != 0, interp, workspace, workspace_bytes, s, level_begin,
                                            level_end);
}

static int backward_ws_split(const void *grad, const float *inputs, const int32_t *offsets_host, void *grad_embeddings,
                             uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                             int align_corners, uint32_t interp, int dtype, void *workspace, uint64_t workspace_bytes,
                             uint32_t level_begin, uint32_t level_end, int split, lnh_stream_t stream) {
    int rc = check_common(inputs, offsets_host, B, D, C, L, dtype);
    if (rc) return rc;
    LNH_REQUIRE(grad && grad_embeddings, LNH_ERR_INVALID_ARG, "grid backward: null grad/grad_embeddings");
    LNH_REQUIRE(D == 3 && C == 2, LNH_ERR_UNSUPPORTED,
                "grid backward (bucketed): only D == 3, C == 2 (use lnh_grid_encode_backward otherwise)");
    LNH_REQUIRE(level_begin <= level_end && level_end <= L, LNH_ERR_INVALID_ARG,
                "grid backward: need level_begin <= level_end <= L");
    if (B == 0) return LNH_OK;
    GridMeta m;
    LNH_REQUIRE(build_meta(m, offsets_host, D, L, S, H, gridtype, align_corners != 0) == 0, LNH_ERR_INVALID_ARG,
                "grid: offsets must be increasing and non-negative");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LNH_F32)
        return launch_backward_bucketed<float>((const float *)grad, inputs, (float *)grad_embeddings, B, L, m,
                                               align_corners != 0, interp, workspace, workspace_bytes, s, level_begin,
                                               level_end, split);
    return launch_backward_bucketed<half_t>((const half_t *)grad, inputs, (half_t *)grad_embeddings, B, L, m,
                                            align_corners != 0, interp, workspace, workspace_bytes, s, level_begin,
                                            level_end, split);
}

int lnh_grid_encode_backward_ws_ex(const void *grad, const float *inputs, const int32_t *offsets_host, void *grad_embeddings,
                                   uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                   int align_corners, uint32_t interp, int dtype, void *workspace, uint64_t workspace_bytes,
                                   uint32_t level_begin, uint32_t level_end, int split, uint32_t flags, lnh_stream_t stream) {
    LNH_REQUIRE(split >= 0 && split <= 2 && (flags & ~(uint32_t)(LNH_BWD_WS_CLEARED | LNH_BWD_TABLE_ZERO)) == 0,
                LNH_ERR_INVALID_ARG, "grid backward: split must be 0 (whole) / 1 (begin) / 2 (finish), flags LNH_BWD_*");
    int rc = check_common(inputs, offsets_host, B, D, C, L, dtype);
    if (rc) return rc;
    LNH_REQUIRE(grad && grad_embeddings, LNH_ERR_INVALID_ARG, "grid backward: null grad/grad_embeddings");
    LNH_REQUIRE(D == 3 && C == 2, LNH_ERR_UNSUPPORTED,
                "grid backward (bucketed): only D == 3, C == 2 (use lnh_grid_encode_backward otherwise)");
    if (level_end > L) level_end = L;
    LNH_REQUIRE(level_begin <= level_end, LNH_ERR_INVALID_ARG, "grid backward: need level_begin <= level_end <= L");
    if (B == 0) return LNH_OK;
    GridMeta m;
    LNH_REQUIRE(build_meta(m, offsets_host, D, L, S, H, gridtype, align_corners != 0) == 0, LNH_ERR_INVALID_ARG,
                "grid: offsets must be increasing and non-negative");
    hipStream_t s = (hipStream_t)stream;
    if (split == 1) {  // begin: everything but the last reduce pass, over all levels
        level_begin = 0;
        level_end = L;
    }
    if (dtype == LNH_F32)
        return launch_backward_bucketed<float>((const float *)grad, inputs, (float *)grad_embeddings, B, L, m,
                                               align_corners != 0, interp, workspace, workspace_bytes, s, level_begin,
                                               level_end, split, flags);
    return launch_backward_bucketed<half_t>((const half_t *)grad, inputs, (half_t *)grad_embeddings, B, L, m,
                                            align_corners != 0, interp, workspace, workspace_bytes, s, level_begin,
                                            level_end, split, flags);
}

uint64_t lnh_grid_backward_workspace_clear_bytes(const int32_t *offsets_host, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                                 float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                                 int dtype, uint64_t workspace_bytes) {
    if (!offsets_host || D != 3 || C != 2 || L == 0 || L > LNH_MAX_LEVELS || B == 0) return 0;
    GridMeta m;
    if (build_meta(m, offsets_host, D, L, S, H, gridtype, align_corners != 0) != 0) return 0;
    const bool plain = align_corners == 0 && interp == 0;
    BucketPlan plan;
    uint32_t nbt = 0;
    uint32_t step;
    if (dtype == LNH_F32) {
        step = fit_chunk<float>(B, m, L, plain, workspace_bytes);
        if (step == 0) return 0;
        (void)plan_buckets<float>(plan, m, L, step, 3, plain, nbt);
    } else {
        step = fit_chunk<half_t>(B, m, L, plain, workspace_bytes);
        if (step == 0) return 0;
        (void)plan_buckets<half_t>(plan, m, L, step, 3, plain, nbt);
    }
    return ((uint64_t)(2 * nbt + L) * 4 + kCursorAlign - 1) / kCursorAlign * kCursorAlign;
}

int lnh_grid_encode_backward_ws_begin(const void *grad, const float *inputs, const int32_t *offsets_host,
                                      void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                      uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                      void *workspace, uint64_t workspace_bytes, lnh_stream_t stream) {
    return backward_ws_split(grad, inputs, offsets_host, grad_embeddings, B, D, C, L, S, H, gridtype, align_corners, interp,
                             dtype, workspace, workspace_bytes, 0, L, 1, stream);
}

int lnh_grid_encode_backward_ws_finish(const void *grad, const float *inputs, const int32_t *offsets_host,
                                       void *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                       uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp, int dtype,
                                       void *workspace, uint64_t workspace_bytes, uint32_t level_begin, uint32_t level_end,
                                       lnh_stream_t stream) {
    return backward_ws_split(grad, inputs, offsets_host, grad_embeddings, B, D, C, L, S, H, gridtype, align_corners, interp,
                             dtype, workspace, workspace_bytes, level_begin, level_end, 2, stream);
}

int lnh_grad_total_variation(const void *inputs, const void *embeddings, void *grad, const int32_t *offsets_host,
                             float weight, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             uint32_t gridtype, int align_corners, int dtype, lnh_stream_t stream) {
    int rc = check_common(inputs, offsets_host, B, D, C, L, dtype);
    if (rc) return rc;
    LNH_REQUIRE(embeddings && grad, LNH_ERR_INVALID_ARG, "grad_total_variation: null embeddings/grad");
    if (B == 0) return LNH_OK;
    GridMeta m;
    LNH_REQUIRE(build_meta(m, offsets_host, D, L, S, H, gridtype, align_corners != 0) == 0, LNH_ERR_INVALID_ARG,
                "grid: offsets must be increasing and non-negative");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == LNH_F32) {
        LNH_DISPATCH_D(D, (launch_tv_c<float, DD>((const float *)inputs, (const float *)embeddings, (float *)grad,
                                                  weight, B, C, L, m, align_corners != 0, s)))
    } else {
        LNH_DISPATCH_D(D, (launch_tv_c<half_t, DD>((const half_t *)inputs, (const half_t *)embeddings, (half_t *)grad,
                                                   weight, B, C, L, m, align_corners != 0, s)))
    }
    return rc;
}

int lnh_grid_corner_indices(const float *inputs, const int32_t *offsets_host, uint32_t *out_idx, uint32_t B,
                            uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                            int align_corners, lnh_stream_t stream) {
    int rc = check_common(inputs, offsets_host, B, D, C, L, LNH_F32);
    if (rc) return rc;
    LNH_REQUIRE(out_idx, LNH_ERR_INVALID_ARG, "grid indices: null output");
    if (B == 0) return LNH_OK;
    GridMeta m;
    LNH_REQUIRE(build_meta(m, offsets_host, D, L, S, H, gridtype, align_corners != 0) == 0, LNH_ERR_INVALID_ARG,
                "grid: offsets must be increasing and non-negative");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(div_up(B, 256), L), block(256);
    switch (D) {
        case 2: LNH_LAUNCH((k_grid_indices<2>), grid, block, 0, s, inputs, out_idx, B, C, m, (uint32_t)(align_corners != 0)); break;
        case 3: LNH_LAUNCH((k_grid_indices<3>), grid, block, 0, s, inputs, out_idx, B, C, m, (uint32_t)(align_corners != 0)); break;
        case 4: LNH_LAUNCH((k_grid_indices<4>), grid, block, 0, s, inputs, out_idx, B, C, m, (uint32_t)(align_corners != 0)); break;
        case 5: LNH_LAUNCH((k_grid_indices<5>), grid, block, 0, s, inputs, out_idx, B, C, m, (uint32_t)(align_corners != 0)); break;
    }
    return lnh_check_launch("lnh_grid_corner_indices");
}

}  // extern "C"
