// raymarch.hip — ray utilities and occupancy-grid machinery for gfx950.
// Semantics: lidarnerf/raymarching/src/raymarching.cu (near/far 104-157, sphere 182-217, Morton 71-95/237-279,
// packbits 286-306, march_rays_train 331-534, composite_rays_train 577-772).
//
// Built with -ffp-contract=off: every a*b+c that the reference's compiler fuses is written as an explicit fmaf
// below, everything else is evaluated exactly as written, so the integer results (cell index, occupancy bit, step
// counts, (id, offset, count) ray table) are bit-identical to the CPU oracle.
#include "common.h"

namespace {

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
__device__ __forceinline__ float signf(float x) { return copysignf(1.0f, x); }

// raymarching.cu:71-95 — 10-bit-per-axis Morton code via magic-number bit spreading
__device__ __forceinline__ uint32_t spread3(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton_encode(uint32_t x, uint32_t y, uint32_t z) {
    return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}
__device__ __forceinline__ uint32_t compact3(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

// raymarching.cu:51-69
__device__ __forceinline__ int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)e));
}
__device__ __forceinline__ int mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (float)((double)(dt * H) * 0.5);
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)e));
}
// raymarching.cu:397-405: product in double, clamp in float, truncate
__device__ __forceinline__ int cell_coord(float x, float mip_rbound, uint32_t H) {
    const double v = 0.5 * (double)fmaf(x, mip_rbound, 1.0f) * (double)H;
    return (int)clampf((float)v, 0.0f, (float)(H - 1));
}

struct Probe {
    float x, y, z, dt, t_next, tt;
    uint32_t index;
    bool occ;
};

struct MarchRay {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
};

// The step length at parameter t (raymarching.cu:389): the marcher only ever moves by this amount, occupied or not.
__device__ __forceinline__ float march_dt(float t, float dt_gamma, float dt_min, float dt_max) {
    return clampf(t * dt_gamma, dt_min, dt_max);
}

// What the marcher sees at parameter t (raymarching.cu:379-430): the clamped position, its cell's occupancy bit and — for
// an empty cell — tt, the parameter at which the ray leaves that cell.
__device__ __forceinline__ Probe probe_cell(const MarchRay &r, float t, const uint8_t *__restrict__ grid, float bound,
                                            float dt_gamma, float dt_min, float dt_max, uint32_t C, uint32_t H, float rH,
                                            float H3) {
    Probe p;
    p.x = clampf(fmaf(t, r.dx, r.ox), -bound, bound);
    p.y = clampf(fmaf(t, r.dy, r.oy), -bound, bound);
    p.z = clampf(fmaf(t, r.dz, r.oz), -bound, bound);
    p.dt = march_dt(t, dt_gamma, dt_min, dt_max);
    const int level = max(mip_from_pos(p.x, p.y, p.z, (float)C), mip_from_dt(p.dt, (float)H, (float)C));
    const float mip_bound = fminf(scalbnf(1.0f, level), bound);
    const float mip_rbound = 1 / mip_bound;
    const int nx = cell_coord(p.x, mip_rbound, H), ny = cell_coord(p.y, mip_rbound, H),
              nz = cell_coord(p.z, mip_rbound, H);
    p.index = (uint32_t)((float)level * H3 + (float)morton_encode((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    p.occ = (grid[p.index >> 3] & (1u << (p.index & 7u))) != 0;
    p.t_next = p.tt = t;
    if (!p.occ) {
        const float tx = (((nx + 0.5f + 0.5f * signf(r.dx)) * rH * 2 - 1) * mip_bound - p.x) * r.rdx;
        const float ty = (((ny + 0.5f + 0.5f * signf(r.dy)) * rH * 2 - 1) * mip_bound - p.y) * r.rdy;
        const float tz = (((nz + 0.5f + 0.5f * signf(r.dz)) * rH * 2 - 1) * mip_bound - p.z) * r.rdz;
        p.tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    }
    return p;
}

// One decision of the serial marcher at parameter t (raymarching.cu:379-439): probe_cell + the walk out of an empty cell.
__device__ __forceinline__ Probe probe(const MarchRay &r, float t, const uint8_t *__restrict__ grid, float bound,
                                       float dt_gamma, float dt_min, float dt_max, uint32_t C, uint32_t H, float rH,
                                       float H3) {
    Probe p = probe_cell(r, t, grid, bound, dt_gamma, dt_min, dt_max, C, H, rH, H3);
    if (!p.occ) {
        do {
            t += march_dt(t, dt_gamma, dt_min, dt_max);
        } while (t < p.tt);
        p.t_next = t;
    }
    return p;
}

// slab test of one ray against the box (raymarching.cu:104-177); FMAX / FMAX for a miss
__device__ __forceinline__ void near_far_ray(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                             const float *__restrict__ aabb, uint32_t n, float min_near, float &near_out,
                                             float &far_out) {
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float rdx = 1 / rays_d[n * 3], rdy = 1 / rays_d[n * 3 + 1], rdz = 1 / rays_d[n * 3 + 2];
    const float FMAX = 3.402823466e+38f;
    near_out = far_out = FMAX;
    float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx;
    if (near > far) { float t = near; near = far; far = t; }
    float ny = (aabb[1] - oy) * rdy, fy = (aabb[4] - oy) * rdy;
    if (ny > fy) { float t = ny; ny = fy; fy = t; }
    if (near > fy || ny > far) return;
    if (ny > near) near = ny;
    if (fy < far) far = fy;
    float nz = (aabb[2] - oz) * rdz, fz = (aabb[5] - oz) * rdz;
    if (nz > fz) { float t = nz; nz = fz; fz = t; }
    if (near > fz || nz > far) return;
    if (nz > near) near = nz;
    if (fz < far) far = fz;
    if (near < min_near) near = min_near;
    near_out = near;
    far_out = far;
}

__global__ void __launch_bounds__(128)
k_near_far(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ aabb,
           uint32_t N, float min_near, float *__restrict__ nears, float *__restrict__ fars) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float near, far;
    near_far_ray(rays_o, rays_d, aabb, n, min_near, near, far);
    nears[n] = near;
    fars[n] = far;
}

// What NeRFRenderer.run_cuda does in front of the marcher, in one launch (the occupancy-grid step is launch-bound: every
// tiny kernel costs ~5 us of device time inside a replayed graph): the LiDAR range of every ray — near = the constant
// `near`, far = min(near * far_factor, exit of the box) (renderer.py:129-138 with the box cut of nerf/renderer.py run_cuda) —
// and the clearing of up to four regions (the marcher's sample buffers, its counter, the colour buffer).
struct PrologueZero {
    uint32_t *p[4];
    uint64_t words[4];
    uint32_t first_block[5];
    uint32_t count;
};
__global__ void __launch_bounds__(256)
k_march_prologue(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ aabb,
                 uint32_t N, float near_c, float far_factor, float *__restrict__ nears, float *__restrict__ fars,
                 uint32_t ray_blocks, PrologueZero z) {
    if (blockIdx.x < ray_blocks) {
        const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
        if (n >= N) return;
        float nb, fb;
        near_far_ray(rays_o, rays_d, aabb, n, near_c, nb, fb);
        const float cap = near_c * far_factor;
        // torch.minimum(cap, far_box): a NaN exit (a ray along a face of the box) stays NaN
        nears[n] = near_c;
        fars[n] = fb != fb ? fb : (fb < cap ? fb : cap);
        return;
    }
    const uint32_t bz = blockIdx.x - ray_blocks;
    uint32_t r = 0;
    while (r + 1 < z.count && bz >= z.first_block[r + 1]) r++;
    const uint64_t b = bz - z.first_block[r], nb = z.first_block[r + 1] - z.first_block[r];
    uint32_t *p = z.p[r];
    const uint64_t n = z.words[r];
    const uint64_t head = (uint64_t)((16 - ((uintptr_t)p & 15)) & 15) / 4;
    const uint64_t h = head < n ? head : n, n4 = (n - h) / 4;
    uint4 *q = reinterpret_cast<uint4 *>(p + h);
    for (uint64_t i = b * 256 + threadIdx.x; i < n4; i += nb * 256) q[i] = make_uint4(0u, 0u, 0u, 0u);
    if (b == 0) {
        if (threadIdx.x < h) p[threadIdx.x] = 0u;
        const uint64_t done = h + n4 * 4;
        if (threadIdx.x < n - done) p[done + threadIdx.x] = 0u;
    }
}

__global__ void __launch_bounds__(128)
k_sph_from_ray(const float *__restrict__ rays_o, const float *__restrict__ rays_d, float radius, uint32_t N,
               float *__restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float A = dx * dx + dy * dy + dz * dz;
    const float Bh = ox * dx + oy * dy + oz * dz;
    const float Cc = ox * ox + oy * oy + oz * oz - radius * radius;
    const float t = (-Bh + sqrtf(Bh * Bh - A * Cc)) / A;
    const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
    const float theta = atan2f(sqrtf(x * x + z * z), y);
    const float phi = atan2f(z, x);
    coords[n * 2] = 2 * theta * 0.3183098861837907f - 1;
    coords[n * 2 + 1] = phi * 0.3183098861837907f;
}

__global__ void __launch_bounds__(256)
k_morton(const int32_t *__restrict__ coords, uint32_t N, int32_t *__restrict__ indices) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int32_t)morton_encode((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}
__global__ void __launch_bounds__(256)
k_morton_invert(const int32_t *__restrict__ indices, uint32_t N, int32_t *__restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int32_t ind = indices[n];  // arithmetic shifts of the signed value, as in the reference
    coords[n * 3] = (int32_t)compact3((uint32_t)(ind >> 0));
    coords[n * 3 + 1] = (int32_t)compact3((uint32_t)(ind >> 1));
    coords[n * 3 + 2] = (int32_t)compact3((uint32_t)(ind >> 2));
}

// One thread = one output byte = 8 floats read as two 16-byte loads (32 B contiguous per lane).
__global__ void __launch_bounds__(256)
k_packbits(const float *__restrict__ grid, uint32_t N, float thresh, uint8_t *__restrict__ bitfield) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float4 a = reinterpret_cast<const float4 *>(grid)[(size_t)n * 2];
    const float4 b = reinterpret_cast<const float4 *>(grid)[(size_t)n * 2 + 1];
    uint32_t bits = 0;
    bits |= (a.x > thresh) ? 1u : 0u;
    bits |= (a.y > thresh) ? 2u : 0u;
    bits |= (a.z > thresh) ? 4u : 0u;
    bits |= (a.w > thresh) ? 8u : 0u;
    bits |= (b.x > thresh) ? 16u : 0u;
    bits |= (b.y > thresh) ? 32u : 0u;
    bits |= (b.z > thresh) ? 64u : 0u;
    bits |= (b.w > thresh) ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
}

__global__ void __launch_bounds__(256)
k_occupancy_lookup(const float *__restrict__ xyz, const float *__restrict__ dt, const uint8_t *__restrict__ grid,
                   float bound, uint32_t N, uint32_t C, uint32_t H, uint32_t *__restrict__ cell_index,
                   uint8_t *__restrict__ occ) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float x = clampf(xyz[n * 3], -bound, bound), y = clampf(xyz[n * 3 + 1], -bound, bound),
                z = clampf(xyz[n * 3 + 2], -bound, bound);
    const int level = max(mip_from_pos(x, y, z, (float)C), mip_from_dt(dt[n], (float)H, (float)C));
    const float mip_bound = fminf(scalbnf(1.0f, level), bound);
    const float mip_rbound = 1 / mip_bound;
    const int nx = cell_coord(x, mip_rbound, H), ny = cell_coord(y, mip_rbound, H), nz = cell_coord(z, mip_rbound, H);
    const float H3 = (float)(H * H * H);
    const uint32_t index = (uint32_t)((float)level * H3 + (float)morton_encode((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    cell_index[n] = index;
    occ[n] = (grid[index >> 3] & (1u << (index & 7u))) != 0;
}

// raymarching.cu:331-534, ONE WAVE PER RAY.
// The reference walks a ray serially: at t it looks up the cell; occupied -> emit a sample, t += dt(t); empty -> t += dt(t)
// until t passes the cell's exit tt.  Either way t only ever moves by dt(t) = clamp(t * dt_gamma, dt_min, dt_max), so the
// parameters the walk can visit are the LATTICE t_0, t_1 = t_0 + dt(t_0), ... — the same floating-point recurrence whatever
// the grid holds.  The wave takes 64 lattice points at a time: every lane runs the recurrence up to its own point (64
// masked adds: the one serial part, a few hundred cycles), probes its cell in parallel, and lane-uniform bit arithmetic on
// the ballots replays the walk — runs of occupied lanes are emitted whole, an empty lane jumps to the first lane whose
// t >= its tt.  Same lattice, same probes, same comparisons as the serial walk: the (id, offset, count) table, the
// positions and the deltas are bit-identical to it (tests/test_raymarch_gpu.py against the C oracle).
// (Round 4: the lane-per-ray form ran 4096 rays as 64 one-wave workgroups, ~120 dependent iterations each, twice:
// 106 us per step of the NeRF-MVL-shaped bench.)
struct WalkState {
    float t_base;                  // lattice point of lane 0 of the next chunk
    float pending_tt;              // an empty cell's exit the walk has not reached yet (valid when pending)
    bool pending, done;
    uint32_t emitted;              // samples so far
    float last_t;                  // t after the last emitted sample (deltas[:, 1] = t_after - last_t)
};

constexpr uint32_t kMarchRaysPerGroup = 16;
__global__ void __launch_bounds__(64 * kMarchRaysPerGroup)
k_march_rays_train(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                   const uint8_t *__restrict__ grid, float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                   uint32_t C, uint32_t H, uint32_t M, const float *__restrict__ nears,
                   const float *__restrict__ fars, float *__restrict__ xyzs, float *__restrict__ dirs,
                   float *__restrict__ deltas, int32_t *__restrict__ rays, int32_t *__restrict__ counter,
                   const float *__restrict__ noises) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = blockIdx.x * kMarchRaysPerGroup + wv;
    const bool active = n < N;  // (wave-uniform; an idle wave of the last workgroup still meets the barriers)
    const uint32_t nc = active ? n : 0;
    __shared__ uint32_t s_count[kMarchRaysPerGroup], s_offset[kMarchRaysPerGroup], s_ray0;
    MarchRay r;
    r.ox = rays_o[nc * 3]; r.oy = rays_o[nc * 3 + 1]; r.oz = rays_o[nc * 3 + 2];
    r.dx = rays_d[nc * 3]; r.dy = rays_d[nc * 3 + 1]; r.dz = rays_d[nc * 3 + 2];
    r.rdx = 1 / r.dx; r.rdy = 1 / r.dy; r.rdz = 1 / r.dz;
    const float rH = 1 / (float)H, H3 = (float)(H * H * H);
    const float far = fars[nc];
    const float SQRT3 = 1.7320508075688772f;
    const float dt_min = 2 * SQRT3 / max_steps;
    const float dt_max = 2 * SQRT3 * (float)(1 << (C - 1)) / H;
    float t0 = nears[nc];
    t0 += march_dt(t0, dt_gamma, dt_min, dt_max) * noises[nc];
    const unsigned long long below = (1ull << lane) - 1;

    // One chunk of the walk.  Returns the mask of the lanes whose lattice point becomes a sample; p_out = their probes.
    auto chunk = [&](WalkState &w, Probe &p_out, float &t_lane) -> unsigned long long {
        float t = w.t_base;
#pragma unroll 8
        for (uint32_t i = 0; i < 63; i++)
            if (i < lane) t += march_dt(t, dt_gamma, dt_min, dt_max);
        t_lane = t;
        const float t_last = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 63));
        w.t_base = t_last + march_dt(t_last, dt_gamma, dt_min, dt_max);
        const bool valid_l = t < far;
        const unsigned long long valid = __ballot(valid_l);
        Probe p;
        p.occ = false;
        p.tt = t;
        if (valid_l) p = probe_cell(r, t, grid, bound, dt_gamma, dt_min, dt_max, C, H, rH, H3);
        p_out = p;
        const unsigned long long occm = __ballot(valid_l && p.occ);
        unsigned long long emit = 0;
        uint32_t pos = 0;
        if (w.pending) {  // still inside the empty cell an earlier chunk met
            const unsigned long long ge = __ballot(t >= w.pending_tt);
            if (!ge) {  // (t grows along the lanes: nobody here has left it; past `far` the walk is over all the same)
                if (valid != ~0ull) w.done = true;
                return 0;
            }
            pos = (uint32_t)__builtin_ctzll(ge);
            w.pending = false;
        }
        while (pos < 64) {
            if (!((valid >> pos) & 1)) { w.done = true; break; }                 // t >= far: the walk ends
            if (w.emitted + (uint32_t)__builtin_popcountll(emit) >= max_steps) { w.done = true; break; }
            if ((occm >> pos) & 1) {                                             // a run of occupied lattice points
                const unsigned long long rest = ~occm >> pos;
                const uint32_t run = rest ? (uint32_t)__builtin_ctzll(rest) : 64 - pos;
                emit |= (run >= 64 ? ~0ull : ((1ull << run) - 1)) << pos;
                pos += run;
            } else {                                                             // empty: on to the first t >= tt
                const float tt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p.tt), (int)pos));
                const unsigned long long ge = __ballot(t >= tt) & ~((2ull << pos) - 1);
                if (!ge) { w.pending = true; w.pending_tt = tt; break; }
                pos = (uint32_t)__builtin_ctzll(ge);
            }
        }
        // at most max_steps samples per ray: drop what a run emitted beyond
        while (w.emitted + (uint32_t)__builtin_popcountll(emit) > max_steps) {
            emit &= ~(1ull << (63 - __builtin_clzll(emit)));
            w.done = true;
        }
        if (valid != ~0ull) w.done = true;  // t >= far inside this chunk (also when an empty cell's exit lies beyond it)
        return emit;
    };

    // pass 1: count
    WalkState w{t0, 0.0f, false, !active, 0u, t0};
    while (!w.done) {
        Probe p;
        float t;
        const unsigned long long emit = chunk(w, p, t);
        w.emitted += (uint32_t)__builtin_popcountll(emit);
    }
    const uint32_t num_steps = w.emitted;
    // sample offsets and ray-table slots: ONE pair of atomics per workgroup for its 16 rays (the reference takes a pair per
    // ray, raymarching.cu:441-447; 4096 same-address returning atomics cost ~100 us by themselves — any disjoint
    // allocation is a valid arrival order)
    if (lane == 0) s_count[wv] = num_steps;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t here = min(kMarchRaysPerGroup, N - blockIdx.x * kMarchRaysPerGroup);
        uint32_t total = 0;
        for (uint32_t i = 0; i < here; i++) {
            s_offset[i] = total;
            total += s_count[i];
        }
        const uint32_t base = (uint32_t)atomicAdd(counter, (int32_t)total);
        s_ray0 = (uint32_t)atomicAdd(counter + 1, (int32_t)here);
        for (uint32_t i = 0; i < here; i++) s_offset[i] += base;
    }
    __syncthreads();
    if (!active) return;
    const uint32_t point_index = s_offset[wv], ray_index = s_ray0 + wv;
    if (lane == 0) {
        rays[ray_index * 3] = (int32_t)n;
        rays[ray_index * 3 + 1] = (int32_t)point_index;
        rays[ray_index * 3 + 2] = (int32_t)num_steps;
    }
    if (num_steps == 0) return;
    if (point_index + num_steps > M) return;

    // pass 2: the same walk, writing
    w = WalkState{t0, 0.0f, false, false, 0u, t0};
    while (!w.done) {
        Probe p;
        float t;
        const unsigned long long emit = chunk(w, p, t);
        if (!emit) continue;
        const float t_after = t + p.dt;  // (= the next lattice point: the serial walk's `t += dt`)
        const unsigned long long before = emit & below;
        // t after the previous sample: the emitted lane below this one, or the carry from the chunks before
        const int prev = before ? 63 - (int)__builtin_clzll(before) : 0;
        const float prev_after = __shfl(t_after, prev, 64);
        if ((emit >> lane) & 1) {
            const size_t row = (size_t)point_index + w.emitted + (uint32_t)__builtin_popcountll(before);
            xyzs[row * 3] = p.x; xyzs[row * 3 + 1] = p.y; xyzs[row * 3 + 2] = p.z;
            dirs[row * 3] = r.dx; dirs[row * 3 + 1] = r.dy; dirs[row * 3 + 2] = r.dz;
            deltas[row * 2] = p.dt;
            deltas[row * 2 + 1] = t_after - (before ? prev_after : w.last_t);
        }
        const int top = 63 - (int)__builtin_clzll(emit);
        w.last_t = __shfl(t_after, top, 64);
        w.emitted += (uint32_t)__builtin_popcountll(emit);
    }
}

// raymarching.cu:577-655
__global__ void __launch_bounds__(64)
k_composite_train_fwd(const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                      const float *__restrict__ deltas, const int32_t *__restrict__ rays, uint32_t M, uint32_t N,
                      float T_thresh, float *__restrict__ weights_sum, float *__restrict__ depth,
                      float *__restrict__ image) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1],
                   num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) {
        weights_sum[index] = 0; depth[index] = 0;
        image[index * 3] = 0; image[index * 3 + 1] = 0; image[index * 3 + 2] = 0;
        return;
    }
    const float *s = sigmas + offset, *c = rgbs + (size_t)offset * 3, *dl = deltas + (size_t)offset * 2;
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
    for (uint32_t step = 0; step < num_steps; step++) {
        const float alpha = 1.0f - expf(-s[0] * dl[0]);
        const float weight = alpha * T;
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

// raymarching.cu:690-772
__global__ void __launch_bounds__(64)
k_composite_train_bwd(const float *__restrict__ grad_ws, const float *__restrict__ grad_image,
                      const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                      const float *__restrict__ deltas, const int32_t *__restrict__ rays,
                      const float *__restrict__ weights_sum, const float *__restrict__ image, uint32_t M, uint32_t N,
                      float T_thresh, float *__restrict__ grad_sigmas, float *__restrict__ grad_rgbs) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1],
                   num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) return;
    const float gi0 = grad_image[index * 3], gi1 = grad_image[index * 3 + 1], gi2 = grad_image[index * 3 + 2];
    const float gws = grad_ws[index], ws_final = weights_sum[index];
    const float rf = image[index * 3], gf = image[index * 3 + 1], bf = image[index * 3 + 2];
    const float *s = sigmas + offset, *c = rgbs + (size_t)offset * 3, *dl = deltas + (size_t)offset * 2;
    float *gs = grad_sigmas + offset, *gc = grad_rgbs + (size_t)offset * 3;
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0;
    for (uint32_t step = 0; step < num_steps; step++) {
        const float alpha = 1.0f - expf(-s[0] * dl[0]);
        const float weight = alpha * T;
        r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
        ws += weight;
        T *= 1.0f - alpha;
        gc[0] = gi0 * weight; gc[1] = gi1 * weight; gc[2] = gi2 * weight;
        gs[0] = dl[0] * (gi0 * (T * c[0] - (rf - r)) + gi1 * (T * c[1] - (gf - g)) + gi2 * (T * c[2] - (bf - b)) +
                         gws * (1 - ws_final));
        if (T < T_thresh) break;
        s++; c += 3; dl += 2; gs++; gc += 3;
    }
}

// ------------------------------------------------------------------------------------------------ LiDAR ragged composite
// The reference's ragged compositing (above) is the RGB template: 3 channels, relative depth, NO depth gradient
// (raymarching.py:330).  The LiDAR path composites K channels (ray-drop, intensity: K = 2) and needs the ABSOLUTE
// depth sum(w * z) WITH its gradient (renderer.py:233-271 semantics on the marcher's ragged samples):
//   z_i = (xyz_i - o) . d  (distance of the sample along its unit ray),  alpha_i = 1 - exp(-sigma_i dt_i),
//   w_i = alpha_i T_i,  T_{i+1} = T_i (1 - alpha_i),  stop after the sample that takes T below T_thresh.
// Backward, for any composited quantity O = sum w_i c_i:  dO/dsigma_i = dt_i (T_{i+1} c_i - (O - O_{<=i})).
// One WAVE per ray, one lane per sample (chunks of 64 along the ray): T_i comes out of a multiplicative scan of (1 - alpha),
// the running sums the backward needs out of additive scans, the early stop out of a ballot.  (Round 4: the thread-per-ray
// form of these two kernels — the reference's shape, raymarching.cu:577-772 — ran 4096 rays as 64 single-wave workgroups
// with ~77 dependent iterations each: 41 + 33 us per step of the NeRF-MVL-shaped bench, latency from end to end.)
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float u = __shfl_up(v, o, 64);
        if (lane >= o) v *= u;
    }
    return v;
}

template <int K>
__global__ void __launch_bounds__(256)
k_lidar_composite_ragged_fwd(const float *__restrict__ sigmas, const float *__restrict__ feats,
                             const float *__restrict__ deltas, const float *__restrict__ xyzs,
                             const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                             const int32_t *__restrict__ rays, uint32_t M, uint32_t N, float T_thresh,
                             float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image) {
    const int lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;  // (wave-uniform)
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1],
                   num_steps = (uint32_t)rays[n * 3 + 2];
    float acc[K], ws = 0, d = 0;
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = 0;
    if (num_steps != 0 && offset + num_steps <= M) {
        const float ox = rays_o[index * 3], oy = rays_o[index * 3 + 1], oz = rays_o[index * 3 + 2];
        const float dx = rays_d[index * 3], dy = rays_d[index * 3 + 1], dz = rays_d[index * 3 + 2];
        float Tc = 1.0f;  // transmittance in front of this chunk
        for (uint32_t base = 0; base < num_steps; base += 64) {
            const uint32_t step = base + lane;
            const bool in = step < num_steps;
            const size_t i = (size_t)offset + (in ? step : 0);
            float alpha = 0.0f, z = 0.0f, f[K];
#pragma unroll
            for (int k = 0; k < K; k++) f[k] = 0.0f;
            if (in) {
                alpha = 1.0f - expf(-sigmas[i] * deltas[i * 2]);
                z = (xyzs[i * 3] - ox) * dx + (xyzs[i * 3 + 1] - oy) * dy + (xyzs[i * 3 + 2] - oz) * dz;
#pragma unroll
                for (int k = 0; k < K; k++) f[k] = feats[i * K + k];
            }
            const float incl = wave_scan_mul(1.0f - alpha, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            const float T = Tc * excl, Tn = Tc * incl;  // T_i, T_{i+1}
            // stop after the first sample that takes T below the threshold
            const unsigned long long stop = __ballot(in && Tn < T_thresh);
            const int last = stop ? __ffsll(stop) - 1 : 63;
            const float w = (in && lane <= last) ? alpha * T : 0.0f;
            ws += w;
            d += w * z;
#pragma unroll
            for (int k = 0; k < K; k++) acc[k] += w * f[k];
            if (stop) break;
            Tc = __shfl(Tn, 63, 64);
        }
        ws = wave_sum(ws);
        d = wave_sum(d);
#pragma unroll
        for (int k = 0; k < K; k++) acc[k] = wave_sum(acc[k]);
    }
    if (lane == 0) {
        weights_sum[index] = ws;
        depth[index] = d;
#pragma unroll
        for (int k = 0; k < K; k++) image[index * K + k] = acc[k];
    }
}

template <int K>
__global__ void __launch_bounds__(256)
k_lidar_composite_ragged_bwd(const float *__restrict__ grad_ws, const float *__restrict__ grad_depth,
                             const float *__restrict__ grad_image, const float *__restrict__ sigmas,
                             const float *__restrict__ feats, const float *__restrict__ deltas,
                             const float *__restrict__ xyzs, const float *__restrict__ rays_o,
                             const float *__restrict__ rays_d, const int32_t *__restrict__ rays,
                             const float *__restrict__ weights_sum, const float *__restrict__ depth,
                             const float *__restrict__ image, uint32_t M, uint32_t N, float T_thresh,
                             float *__restrict__ grad_sigmas, float *__restrict__ grad_feats) {
    const int lane = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1],
                   num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) return;  // (grad buffers are zero-initialised by the caller)
    const float ox = rays_o[index * 3], oy = rays_o[index * 3 + 1], oz = rays_o[index * 3 + 2];
    const float dx = rays_d[index * 3], dy = rays_d[index * 3 + 1], dz = rays_d[index * 3 + 2];
    const float gws = grad_ws[index], gd = grad_depth[index], ws_final = weights_sum[index], d_final = depth[index];
    float gi[K], fin[K], accc[K];  // accc / dc: the sums over the chunks in front of this one
#pragma unroll
    for (int k = 0; k < K; k++) {
        gi[k] = grad_image[index * K + k];
        fin[k] = image[index * K + k];
        accc[k] = 0;
    }
    float Tc = 1.0f, dc = 0.0f;
    for (uint32_t base = 0; base < num_steps; base += 64) {
        const uint32_t step = base + lane;
        const bool in = step < num_steps;
        const size_t i = (size_t)offset + (in ? step : 0);
        float alpha = 0.0f, z = 0.0f, dt = 0.0f, f[K];
#pragma unroll
        for (int k = 0; k < K; k++) f[k] = 0.0f;
        if (in) {
            dt = deltas[i * 2];
            alpha = 1.0f - expf(-sigmas[i] * dt);
            z = (xyzs[i * 3] - ox) * dx + (xyzs[i * 3 + 1] - oy) * dy + (xyzs[i * 3 + 2] - oz) * dz;
#pragma unroll
            for (int k = 0; k < K; k++) f[k] = feats[i * K + k];
        }
        const float incl = wave_scan_mul(1.0f - alpha, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float T = Tc * excl, Tn = Tc * incl;
        const unsigned long long stop = __ballot(in && Tn < T_thresh);
        const int last = stop ? __ffsll(stop) - 1 : 63;
        const bool use = in && lane <= last;
        const float w = use ? alpha * T : 0.0f;
        const float d = dc + wave_scan_add(w * z, lane);  // sum_{j <= i} w_j z_j
        float g = gws * (1.0f - ws_final) + gd * (Tn * z - (d_final - d));
#pragma unroll
        for (int k = 0; k < K; k++) {
            const float a = accc[k] + wave_scan_add(w * f[k], lane);
            g += gi[k] * (Tn * f[k] - (fin[k] - a));
            if (use) grad_feats[i * K + k] = gi[k] * w;
            accc[k] = __shfl(a, 63, 64);
        }
        if (use) grad_sigmas[i] = dt * g;
        if (stop) break;
        dc = __shfl(d, 63, 64);
        Tc = __shfl(Tn, 63, 64);
    }
}

// ------------------------------------------------------------------------------------------------ inference
// raymarching.cu:808-928: march the ALIVE rays for at most n_step occupied samples starting at their current t;
// slots a ray does not fill stay as the caller initialised them (zeros: delta == 0 marks the end of a ray).
__global__ void __launch_bounds__(64)
k_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive,
             const float *__restrict__ rays_t, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
             const uint8_t *__restrict__ grid, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
             const float *__restrict__ nears, const float *__restrict__ fars, float *__restrict__ xyzs,
             float *__restrict__ dirs, float *__restrict__ deltas, const float *__restrict__ noises) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const uint32_t index = (uint32_t)rays_alive[n];
    MarchRay r;
    r.ox = rays_o[index * 3]; r.oy = rays_o[index * 3 + 1]; r.oz = rays_o[index * 3 + 2];
    r.dx = rays_d[index * 3]; r.dy = rays_d[index * 3 + 1]; r.dz = rays_d[index * 3 + 2];
    r.rdx = 1 / r.dx; r.rdy = 1 / r.dy; r.rdz = 1 / r.dz;
    const float rH = 1 / (float)H, H3 = (float)(H * H * H);
    const float far = fars[index];
    const float SQRT3 = 1.7320508075688772f;
    const float dt_min = 2 * SQRT3 / max_steps;
    const float dt_max = 2 * SQRT3 * (float)(1 << (C - 1)) / H;
    float t = rays_t[index];
    t += clampf(t * dt_gamma, dt_min, dt_max) * noises[n];
    float *px = xyzs + (size_t)n * n_step * 3, *pd = dirs + (size_t)n * n_step * 3,
          *pl = deltas + (size_t)n * n_step * 2;
    float last_t = t;
    uint32_t step = 0;
    while (t < far && step < n_step) {
        const Probe p = probe(r, t, grid, bound, dt_gamma, dt_min, dt_max, C, H, rH, H3);
        if (p.occ) {
            px[0] = p.x; px[1] = p.y; px[2] = p.z;
            pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
            t += p.dt;
            pl[0] = p.dt;
            pl[1] = t - last_t;
            last_t = t;
            px += 3; pd += 3; pl += 2;
            step++;
        } else {
            t = p.t_next;
        }
    }
}

// raymarching.cu:966-1053: accumulate n_step samples into the per-ray running sums (T_i = 1 - sum of earlier weights);
// a ray that ends (delta == 0) or saturates (T < T_thresh) is marked dead (rays_alive[n] = -1), others save their t.
__global__ void __launch_bounds__(64)
k_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *__restrict__ rays_alive,
                 float *__restrict__ rays_t, const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                 const float *__restrict__ deltas, float *__restrict__ weights_sum, float *__restrict__ depth,
                 float *__restrict__ image) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
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

}  // namespace

extern "C" {

int lnh_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N, float min_near,
                           float *nears, float *fars, lnh_stream_t stream) {
    LNH_REQUIRE(rays_o && rays_d && aabb && nears && fars, LNH_ERR_INVALID_ARG, "near_far_from_aabb: null pointer");
    if (N == 0) return LNH_OK;
    LNH_LAUNCH(k_near_far, dim3(div_up(N, 128)), dim3(128), 0, (hipStream_t)stream, rays_o, rays_d, aabb, N,
                       min_near, nears, fars);
    return lnh_check_launch("lnh_near_far_from_aabb");
}

int lnh_lidar_march_prologue(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N, float near,
                             float far_factor, float *nears, float *fars, void *const *zero_ptrs,
                             const uint64_t *zero_bytes, uint32_t zero_count, lnh_stream_t stream) {
    LNH_REQUIRE(zero_count <= 4, LNH_ERR_INVALID_ARG, "lidar_march_prologue: at most 4 regions to clear");
    LNH_REQUIRE(N == 0 || (rays_o && rays_d && aabb && nears && fars), LNH_ERR_INVALID_ARG,
                "lidar_march_prologue: null pointer");
    LNH_REQUIRE(zero_count == 0 || (zero_ptrs && zero_bytes), LNH_ERR_INVALID_ARG, "lidar_march_prologue: null region list");
    PrologueZero z{};
    uint32_t blocks = 0;
    for (uint32_t i = 0; i < zero_count; i++) {
        if (!zero_bytes[i]) continue;
        LNH_REQUIRE(zero_ptrs[i] && ((uintptr_t)zero_ptrs[i] & 3) == 0 && (zero_bytes[i] & 3) == 0, LNH_ERR_INVALID_ARG,
                    "lidar_march_prologue: regions are 4-byte granular");
        const uint64_t words = zero_bytes[i] / 4, want = (words / 4 + 255) / 256 + 1;
        z.p[z.count] = (uint32_t *)zero_ptrs[i];
        z.words[z.count] = words;
        z.first_block[z.count] = blocks;
        blocks += (uint32_t)(want < 1024 ? want : 1024);
        z.count++;
    }
    z.first_block[z.count] = blocks;
    const uint32_t ray_blocks = div_up(N, 256);
    if (ray_blocks + blocks == 0) return LNH_OK;
    LNH_LAUNCH(k_march_prologue, dim3(ray_blocks + blocks), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, aabb, N,
               near, far_factor, nears, fars, ray_blocks, z);
    return lnh_check_launch("lnh_lidar_march_prologue");
}

int lnh_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords,
                     lnh_stream_t stream) {
    LNH_REQUIRE(rays_o && rays_d && coords, LNH_ERR_INVALID_ARG, "sph_from_ray: null pointer");
    if (N == 0) return LNH_OK;
    LNH_LAUNCH(k_sph_from_ray, dim3(div_up(N, 128)), dim3(128), 0, (hipStream_t)stream, rays_o, rays_d, radius,
                       N, coords);
    return lnh_check_launch("lnh_sph_from_ray");
}

int lnh_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, lnh_stream_t stream) {
    LNH_REQUIRE(coords && indices, LNH_ERR_INVALID_ARG, "morton3D: null pointer");
    if (N == 0) return LNH_OK;
    LNH_LAUNCH(k_morton, dim3(div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, coords, N, indices);
    return lnh_check_launch("lnh_morton3D");
}

int lnh_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, lnh_stream_t stream) {
    LNH_REQUIRE(coords && indices, LNH_ERR_INVALID_ARG, "morton3D_invert: null pointer");
    if (N == 0) return LNH_OK;
    LNH_LAUNCH(k_morton_invert, dim3(div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, indices, N, coords);
    return lnh_check_launch("lnh_morton3D_invert");
}

int lnh_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, lnh_stream_t stream) {
    LNH_REQUIRE(grid && bitfield, LNH_ERR_INVALID_ARG, "packbits: null pointer");
    LNH_REQUIRE(((uintptr_t)grid & 15) == 0, LNH_ERR_INVALID_ARG, "packbits: grid must be 16-byte aligned");
    if (N == 0) return LNH_OK;
    LNH_LAUNCH(k_packbits, dim3(div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, grid, N, density_thresh,
                       bitfield);
    return lnh_check_launch("lnh_packbits");
}

int lnh_occupancy_lookup(const float *xyz, const float *dt, const uint8_t *bitfield, float bound, uint32_t N,
                         uint32_t C, uint32_t H, uint32_t *cell_index, uint8_t *occ, lnh_stream_t stream) {
    LNH_REQUIRE(xyz && dt && bitfield && cell_index && occ, LNH_ERR_INVALID_ARG, "occupancy_lookup: null pointer");
    LNH_REQUIRE(C >= 1 && C <= 8 && H >= 1 && H <= 1024, LNH_ERR_INVALID_ARG, "occupancy_lookup: bad cascade / grid size");
    if (N == 0) return LNH_OK;
    LNH_LAUNCH(k_occupancy_lookup, dim3(div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, xyz, dt, bitfield,
                       bound, N, C, H, cell_index, occ);
    return lnh_check_launch("lnh_occupancy_lookup");
}

int lnh_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                         uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *nears,
                         const float *fars, float *xyzs, float *dirs, float *deltas, int32_t *rays, int32_t *counter,
                         const float *noises, lnh_stream_t stream) {
    LNH_REQUIRE(rays_o && rays_d && grid && nears && fars && xyzs && dirs && deltas && rays && counter && noises,
                LNH_ERR_INVALID_ARG, "march_rays_train: null pointer");
    LNH_REQUIRE(C >= 1 && C <= 8 && H >= 1 && H <= 1024 && max_steps >= 1, LNH_ERR_INVALID_ARG,
                "march_rays_train: bad cascade / grid size / max_steps");
    if (N == 0) return LNH_OK;
    LNH_LAUNCH(k_march_rays_train, dim3(div_up(N, kMarchRaysPerGroup)), dim3(64 * kMarchRaysPerGroup), 0,
               (hipStream_t)stream, rays_o, rays_d, grid,
                       bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises);
    return lnh_check_launch("lnh_march_rays_train");
}

int lnh_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays,
                                     uint32_t M, uint32_t N, float T_thresh, float *weights_sum, float *depth,
                                     float *image, lnh_stream_t stream) {
    LNH_REQUIRE(sigmas && rgbs && deltas && rays && weights_sum && depth && image, LNH_ERR_INVALID_ARG,
                "composite_rays_train_forward: null pointer");
    if (N == 0) return LNH_OK;
    LNH_LAUNCH(k_composite_train_fwd, dim3(div_up(N, 64)), dim3(64), 0, (hipStream_t)stream, sigmas, rgbs,
                       deltas, rays, M, N, T_thresh, weights_sum, depth, image);
    return lnh_check_launch("lnh_composite_rays_train_forward");
}

int lnh_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image, const float *sigmas,
                                      const float *rgbs, const float *deltas, const int32_t *rays,
                                      const float *weights_sum, const float *image, uint32_t M, uint32_t N,
                                      float T_thresh, float *grad_sigmas, float *grad_rgbs, lnh_stream_t stream) {
    LNH_REQUIRE(grad_weights_sum && grad_image && sigmas && rgbs && deltas && rays && weights_sum && image &&
                    grad_sigmas && grad_rgbs,
                LNH_ERR_INVALID_ARG, "composite_rays_train_backward: null pointer");
    if (N == 0) return LNH_OK;
    LNH_LAUNCH(k_composite_train_bwd, dim3(div_up(N, 64)), dim3(64), 0, (hipStream_t)stream, grad_weights_sum,
                       grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas,
                       grad_rgbs);
    return lnh_check_launch("lnh_composite_rays_train_backward");
}

int lnh_lidar_composite_rays_train_forward(const float *sigmas, const float *feats, const float *deltas,
                                           const float *xyzs, const float *rays_o, const float *rays_d,
                                           const int32_t *rays, uint32_t M, uint32_t N, uint32_t K, float T_thresh,
                                           float *weights_sum, float *depth, float *image, lnh_stream_t stream) {
    LNH_REQUIRE(sigmas && feats && deltas && xyzs && rays_o && rays_d && rays && weights_sum && depth && image,
                LNH_ERR_INVALID_ARG, "lidar_composite_rays_train_forward: null pointer");
    LNH_REQUIRE(K >= 1 && K <= 3, LNH_ERR_UNSUPPORTED, "lidar_composite_rays_train: K must be 1, 2 or 3 (got %u)", K);
    if (N == 0) return LNH_OK;
    hipStream_t s = (hipStream_t)stream;
#define LNH_RAGGED_FWD(KK)                                                                                          \
    LNH_LAUNCH(k_lidar_composite_ragged_fwd<KK>, dim3(div_up(N, 4)), dim3(256), 0, s, sigmas, feats, deltas, xyzs, \
               rays_o, rays_d, rays, M, N, T_thresh, weights_sum, depth, image)
    if (K == 1) LNH_RAGGED_FWD(1); else if (K == 2) LNH_RAGGED_FWD(2); else LNH_RAGGED_FWD(3);
#undef LNH_RAGGED_FWD
    return lnh_check_launch("lnh_lidar_composite_rays_train_forward");
}

int lnh_lidar_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_depth,
                                            const float *grad_image, const float *sigmas, const float *feats,
                                            const float *deltas, const float *xyzs, const float *rays_o,
                                            const float *rays_d, const int32_t *rays, const float *weights_sum,
                                            const float *depth, const float *image, uint32_t M, uint32_t N, uint32_t K,
                                            float T_thresh, float *grad_sigmas, float *grad_feats, lnh_stream_t stream) {
    LNH_REQUIRE(grad_weights_sum && grad_depth && grad_image && sigmas && feats && deltas && xyzs && rays_o && rays_d &&
                    rays && weights_sum && depth && image && grad_sigmas && grad_feats,
                LNH_ERR_INVALID_ARG, "lidar_composite_rays_train_backward: null pointer");
    LNH_REQUIRE(K >= 1 && K <= 3, LNH_ERR_UNSUPPORTED, "lidar_composite_rays_train: K must be 1, 2 or 3 (got %u)", K);
    if (N == 0) return LNH_OK;
    hipStream_t s = (hipStream_t)stream;
#define LNH_RAGGED_BWD(KK)                                                                                          \
    LNH_LAUNCH(k_lidar_composite_ragged_bwd<KK>, dim3(div_up(N, 4)), dim3(256), 0, s, grad_weights_sum, grad_depth, \
               grad_image, sigmas, feats, deltas, xyzs, rays_o, rays_d, rays, weights_sum, depth, image, M, N,     \
               T_thresh, grad_sigmas, grad_feats)
    if (K == 1) LNH_RAGGED_BWD(1); else if (K == 2) LNH_RAGGED_BWD(2); else LNH_RAGGED_BWD(3);
#undef LNH_RAGGED_BWD
    return lnh_check_launch("lnh_lidar_composite_rays_train_backward");
}

int lnh_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                   const float *rays_o, const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C,
                   uint32_t H, const uint8_t *grid, const float *nears, const float *fars, float *xyzs, float *dirs,
                   float *deltas, const float *noises, lnh_stream_t stream) {
    LNH_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && nears && fars && xyzs && dirs && deltas && noises,
                LNH_ERR_INVALID_ARG, "march_rays: null pointer");
    LNH_REQUIRE(C >= 1 && C <= 16 && H >= 1 && H <= 1024 && max_steps >= 1, LNH_ERR_INVALID_ARG,
                "march_rays: bad cascade / grid size / max_steps");
    if (n_alive == 0 || n_step == 0) return LNH_OK;
    LNH_LAUNCH(k_march_rays, dim3(div_up(n_alive, 64)), dim3(64), 0, (hipStream_t)stream, n_alive, n_step, rays_alive,
               rays_t, rays_o, rays_d, grid, bound, dt_gamma, max_steps, C, H, nears, fars, xyzs, dirs, deltas, noises);
    return lnh_check_launch("lnh_march_rays");
}

int lnh_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                       const float *sigmas, const float *rgbs, const float *deltas, float *weights_sum, float *depth,
                       float *image, lnh_stream_t stream) {
    LNH_REQUIRE(rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image, LNH_ERR_INVALID_ARG,
                "composite_rays: null pointer");
    if (n_alive == 0) return LNH_OK;
    LNH_LAUNCH(k_composite_rays, dim3(div_up(n_alive, 64)), dim3(64), 0, (hipStream_t)stream, n_alive, n_step, T_thresh,
               rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image);
    return lnh_check_launch("lnh_composite_rays");
}

}  // extern "C"
