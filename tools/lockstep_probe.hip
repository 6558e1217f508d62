// Lock-step LDS staging of the per-level features, timed once (north_star: "LDS staging of per-level features"; review item 7
// of round 3).  The only form of an encode -> sigma-net fusion that keeps the level-major locality of the tables: persistent
// workgroups (one per CU) hold a tile of points, walk the levels TOGETHER — a chip-wide barrier per level, so that every XCD's
// L2 still sees one level's table at a time — keep the 32 features of their points in LDS, and consume them there (the sigma
// net would run from LDS; here a stand-in reads every feature once and writes 4 bytes per point, which lower-bounds it).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -ffp-contract=off tools/lockstep_probe.hip \
//         lidar-nerf_amd/csrc/core.hip -o tools/bin/lockstep_probe && tools/bin/lockstep_probe
//
// Compared, on the same points (4096 LiDAR rays x 832 samples, fp16 tables, 16 levels to resolution 32768) and with the
// same per-point arithmetic (locate + 8 corner gathers, no wide-load tricks):
//   (a) level-major launch writing [L, B, 2] to HBM                      — the shape of the product's k_grid_forward
//   (b) lock-step: features to LDS, grid barrier per level, consume      — the fusion candidate
//   (c) the same without the barriers (workgroups drift apart)           — what round 3 measured on the product kernel
// plus the product's own forward for reference.  Results: profiles/r04_fusion_lockstep.txt.
#include "../lidar-nerf_amd/csrc/grid.hip"

#include <cstdlib>
#include <vector>

namespace {

template <typename T>
__device__ __forceinline__ Vec<T, 2> encode_point(const float *__restrict__ inputs, const T *__restrict__ table, uint32_t b,
                                                  const LevelParams &lv) {
    float x[3];
#pragma unroll
    for (int d = 0; d < 3; d++) x[d] = inputs[(size_t)b * 3 + d];
    Cell<3> cell;
    Vec<T, 2> res;
    res.v[0] = res.v[1] = (T)0.0f;
    if (!locate<3>(x, lv, false, 0, cell)) return res;
    const T *tab = table + (size_t)lv.offset * 2;
    Vec<T, 2> g[8];
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) g[c] = load_vec<T, 2>(tab + (size_t)corner_row<3>(cell, lv, c) * 2);
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
        const float w = corner_weight<3>(cell, c);
#pragma unroll
        for (int ch = 0; ch < 2; ch++) res.v[ch] = (T)fmaf(w, (float)g[c].v[ch], (float)res.v[ch]);
    }
    return res;
}

template <typename T>
__global__ void __launch_bounds__(256) k_level_major(const float *inputs, const T *table, T *out, uint32_t B, GridMeta meta) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, level = blockIdx.y;
    if (b >= B) return;
    store_vec<T, 2>(out + ((size_t)level * B + b) * 2, encode_point<T>(inputs, table, b, meta.lv[level]));
}

constexpr uint32_t kTile = 2048;  // points of a workgroup's tile: 2048 x 32 features x 2 B = 128 KiB of LDS

template <typename T, bool BARRIER>
__global__ void __launch_bounds__(1024) k_lockstep(const float *inputs, const T *table, float *consumed, uint32_t B, uint32_t L,
                                                   GridMeta meta, uint32_t *bar) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *feat = reinterpret_cast<T *>(smem);  // [kTile][32]
    const uint32_t n_wg = gridDim.x, supers = (B + n_wg * kTile - 1) / (n_wg * kTile);
    uint32_t epoch = 0;
    for (uint32_t s = 0; s < supers; s++) {
        const uint32_t base = (s * n_wg + blockIdx.x) * kTile;
        for (uint32_t level = 0; level < L; level++) {
            const LevelParams lv = meta.lv[level];
#pragma unroll
            for (uint32_t k = 0; k < kTile / 1024; k++) {
                const uint32_t p = threadIdx.x + k * 1024, b = base + p;
                Vec<T, 2> f;
                f.v[0] = f.v[1] = (T)0.0f;
                if (b < B) f = encode_point<T>(inputs, table, b, lv);
                store_vec<T, 2>(feat + (size_t)p * 32 + 2 * level, f);
            }
            if (BARRIER) {  // every workgroup finishes level l before anyone starts level l + 1 (nothing is exchanged: no fences)
                __syncthreads();
                epoch++;
                if (threadIdx.x == 0) {
                    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * n_wg) __builtin_amdgcn_s_sleep(2);
                }
                __syncthreads();
            }
        }
        __syncthreads();
        // stand-in for the sigma net: every feature of the tile is read from LDS once, 4 bytes per point leave the kernel
#pragma unroll
        for (uint32_t k = 0; k < kTile / 1024; k++) {
            const uint32_t p = threadIdx.x + k * 1024, b = base + p;
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < 32; j += 8) {
                const half8_t v = *reinterpret_cast<const half8_t *>(feat + (size_t)p * 32 + j);
#pragma unroll
                for (int i = 0; i < 8; i++) acc += (float)v[i];
            }
            if (b < B) consumed[b] = acc;
        }
        __syncthreads();
    }
}

float frand(uint64_t &s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((s >> 40) & 0xffffff) / 16777216.0f;
}

}  // namespace

int main() {
    const uint32_t N = 4096, T = 832, B = N * T, L = 16, H = 16;
    const double pls = exp2(log2(32768.0 / 16) / 15);
    const float S = (float)log2(pls), SCALE = 0.010784853507573345f;
    std::vector<int32_t> off(L + 1, 0);
    for (uint32_t l = 0; l < L; l++) {
        const uint64_t res = (uint64_t)ceil(16 * pow(pls, (double)l)), side = res + 1;
        uint64_t rows = std::min<uint64_t>(1u << 19, side * side * side);
        off[l + 1] = off[l] + (int32_t)((rows + 7) / 8 * 8);
    }
    // KITTI-360-shaped rays from near the origin, samples 1 m .. 81 m (tools/bench_grid.py lidar_points)
    std::vector<float> x((size_t)B * 3);
    uint64_t seed = 1;
    for (uint32_t r = 0; r < N; r++) {
        const float beta = (frand(seed) - 0.5f) * 6.2831853f, alpha = (2.0f - frand(seed) * 26.9f) / 180.0f * 3.14159265f;
        const float d[3] = {cosf(alpha) * cosf(beta), cosf(alpha) * sinf(beta), sinf(alpha)};
        const float o[3] = {(frand(seed) - 0.5f) * 0.02f, (frand(seed) - 0.5f) * 0.02f, (frand(seed) - 0.5f) * 0.02f};
        for (uint32_t i = 0; i < T; i++) {
            const float z = ((float)i / (T - 1)) * 80 * SCALE + SCALE + (frand(seed) - 0.5f) * (80 * SCALE / T);
            for (int k = 0; k < 3; k++) x[((size_t)r * T + i) * 3 + k] = (fminf(fmaxf(o[k] + d[k] * z, -1.0f), 1.0f) + 1.0f) / 2.0f;
        }
    }
    std::vector<_Float16> tab((size_t)off[L] * 2);
    for (auto &v : tab) v = (_Float16)((frand(seed) - 0.5f) * 2e-4f);
    float *dx, *dcons;
    _Float16 *dtab, *dout;
    uint32_t *dbar;
    hipMalloc(&dx, x.size() * 4);
    hipMalloc(&dtab, tab.size() * 2);
    hipMalloc(&dout, (size_t)L * B * 2 * 2);
    hipMalloc(&dcons, (size_t)B * 4);
    hipMalloc(&dbar, 256);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dtab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice);
    GridMeta m;
    if (build_meta(m, off.data(), 3, L, S, H, 0, false) != 0) return 1;
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto timed = [&](const char *name, auto &&fn) {
        float best = 1e9f;
        for (int r = 0; r < 6; r++) {
            hipEventRecord(e0);
            fn();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (r) best = ms < best ? ms : best;
        }
        printf("%-72s %8.1f us\n", name, best * 1e3f);
    };
    printf("%u points, 16 levels, fp16 tables, %d CUs\n", B, cus);
    timed("product forward (lnh_grid_encode_forward, [L,B,2] to HBM)", [&] {
        lnh_grid_encode_forward(dx, dtab, off.data(), dout, B, 3, 2, L, S, H, nullptr, 0, 0, 0, LNH_F16, nullptr);
    });
    timed("(a) plain gathers, level-major launch, [L,B,2] to HBM", [&] {
        hipLaunchKernelGGL(k_level_major<half_t>, dim3((B + 255) / 256, L), dim3(256), 0, 0, dx, dtab, dout, B, m);
    });
    const size_t lds = (size_t)kTile * 32 * 2;
    hipFuncSetAttribute((const void *)k_lockstep<half_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void *)k_lockstep<half_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    timed("(b) lock-step: features in LDS, grid barrier per level, consume in LDS", [&] {
        hipMemsetAsync(dbar, 0, 4, 0);
        hipLaunchKernelGGL((k_lockstep<half_t, true>), dim3(cus), dim3(1024), lds, 0, dx, dtab, dcons, B, L, m, dbar);
    });
    timed("(c) the same without the barriers (workgroups drift apart)", [&] {
        hipLaunchKernelGGL((k_lockstep<half_t, false>), dim3(cus), dim3(1024), lds, 0, dx, dtab, dcons, B, L, m, dbar);
    });
    // the two orders must have computed the same features: sum over levels of (a)'s output == (b)'s consumed value
    std::vector<_Float16> out((size_t)L * B * 2);
    std::vector<float> cons(B);
    hipMemcpy(out.data(), dout, out.size() * 2, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((k_lockstep<half_t, true>), dim3(cus), dim3(1024), lds, 0, dx, dtab, dcons, B, L, m, dbar);  // (bar keeps counting up: epochs restart) 
    hipMemsetAsync(dbar, 0, 4, 0);
    hipLaunchKernelGGL((k_lockstep<half_t, false>), dim3(cus), dim3(1024), lds, 0, dx, dtab, dcons, B, L, m, dbar);
    hipMemcpy(cons.data(), dcons, cons.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (uint32_t b = 0; b < B; b += 997) {
        float s = 0;
        for (uint32_t l = 0; l < L; l++) s += (float)out[((size_t)l * B + b) * 2] + (float)out[((size_t)l * B + b) * 2 + 1];
        worst = fmax(worst, fabs((double)s - cons[b]));
    }
    printf("max |sum of (a)'s features - (b)'s consumed value| over sampled points: %.3g\n", worst);
    return 0;
}
