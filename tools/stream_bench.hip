// stream_bench.hip — how fast can workgroups shaped like k_grid_bwd_reduce (1024 threads, 128 KiB LDS => one per CU)
// stream a 1.4 GB pool?  Variants: private contiguous region per workgroup (what the reduce pass does) vs regions
// interleaved at 4 KiB granularity across workgroups; with / without the LDS reservation (1 vs 2 workgroups per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/stream_bench.hip -o /tmp/stream_bench && /tmp/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int MODE, int UNROLL>
__global__ void __launch_bounds__(1024) k_stream(const u4 *__restrict__ src, uint64_t per_wg_vec, uint32_t *out) {
    extern __shared__ char lds[];
    const uint32_t wg = blockIdx.x, nwg = gridDim.x, t = threadIdx.x;
    u4 acc = {0, 0, 0, 0};
    if (MODE == 0) {  // private contiguous region
        const u4 *p = src + (uint64_t)wg * per_wg_vec;
        for (uint64_t i = t; i < per_wg_vec; i += 1024 * UNROLL) {
            u4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) v[u] = __builtin_nontemporal_load(p + (i + u * 1024 < per_wg_vec ? i + u * 1024 : 0));
#pragma unroll
            for (int u = 0; u < UNROLL; u++) acc ^= v[u];
        }
    } else {  // 4 KiB chunks (256 vectors) interleaved across workgroups
        const uint64_t chunks = per_wg_vec / 256;
        for (uint64_t c = 0; c < chunks; c += 4 * UNROLL) {
            u4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const uint64_t cc = c + (uint64_t)u * 4 + (t >> 8);
                v[u] = __builtin_nontemporal_load(src + ((cc < chunks ? cc : 0) * nwg + wg) * 256 + (t & 255));
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) acc ^= v[u];
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
    if (lds[t] == 77 && t == 9999) out[1] = 1;
}

int main() {
    const uint32_t nwg = 836;
    const uint64_t per_wg_bytes = 1664 * 1024;  // ~1.39 GB in total, a multiple of 4 KiB * 16
    const uint64_t per_wg_vec = per_wg_bytes / 16, total = per_wg_bytes * nwg;
    u4 *src; uint32_t *out;
    hipMalloc(&src, total); hipMalloc(&out, 64);
    hipMemset(src, 1, total);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, size_t lds, const char *name) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        float best = 1e9;
        for (int r = 0; r < 6; r++) {
            hipMemsetAsync(out, 0, 64, 0);   // also flushes nothing useful; the 1.4 GB do not fit any cache
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(1024), lds, 0, src, per_wg_vec, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
        }
        printf("%-44s %7.1f us  %6.2f TB/s\n", name, best * 1e3, total / (best * 1e-3) / 1e12);
    };
    run(k_stream<0, 4>, 128 * 1024, "private regions, 128 KiB LDS (1 WG/CU), u4");
    run(k_stream<0, 8>, 128 * 1024, "private regions, 128 KiB LDS (1 WG/CU), u8");
    run(k_stream<0, 4>, 0, "private regions, no LDS (2 WG/CU), u4");
    run(k_stream<1, 4>, 128 * 1024, "interleaved 4 KiB, 128 KiB LDS (1 WG/CU), u4");
    run(k_stream<1, 4>, 0, "interleaved 4 KiB, no LDS (2 WG/CU), u4");
    return 0;
}
