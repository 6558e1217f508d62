// wgrad.h — weight gradients of the tiny MLPs as FIXED-ORDER sums over workgroups.
//
// Every backward kernel of the MLP family ends with one partial sum of each weight gradient per workgroup.  Rounds 1-5 added
// those partials into the gradient with fp32 device atomics: the order of the additions — and with it the last bits of every
// weight gradient, and after a few hundred optimizer steps the whole training trajectory — depended on which workgroup
// arrived first.  Now the partials go to a scratch buffer ([workgroup][n] floats, plain stores) and a second, small launch on
// the same stream adds them up in INDEX order:
//   k_wgrad_reduce  a workgroup owns 16 consecutive float4 elements; its 256 threads are 16 slices x 16 elements, slice s
//                   adds the partials of workgroups [s * per, (s + 1) * per) in index order (loads issued eight deep, the adds
//                   in order), the 16 slice sums meet in LDS and are added in slice order; the result is added to the
//                   gradient (read-add-store: one writer per address).
// The summation tree depends on the number of partials only — the same batch gives the same bits on every run and every box.
// The reference forms these gradients with deterministic split-K GEMMs (lidarnerf/ffmlp/src/ffmlp.cu:1107-1142).
//
// (First form of round 6: the last workgroup to arrive — device-scope ticket counters — did the adding inside the backward
// kernel itself.  It cost 40-60 us per launch at 256 workgroups: every workgroup ends with a device-scope release, i.e. a
// write-back of its XCD's L2 with the kernel's whole dX stream still dirty in it, and the adding is two serial rounds of one
// workgroup.  The kernel boundary gives the same visibility for a ~2 us launch gap; no tickets, nothing to keep zeroed.)
//
// One workspace serves every launch of a stream (they use it one after the other); launches that may overlap in time — two
// streams, two processes on one device — need a workspace each.
#pragma once
#include "common.h"

constexpr uint32_t kWgradMaxBlocks = 512;              // workgroups of a backward launch (every launcher clamps to it)
constexpr uint32_t kWgradMaxFloats = 68 * 256;         // 128 -> 64 -> 64 -> 64 -> 16: 32 + 2 * 16 + 4 tiles of 16 x 16
constexpr uint32_t kWgradSlices = 16;

struct WgradWs {
    float *partials;  // [workgroups][n]
};

// 512 partials of the one-kernel MLP backwards' largest gradient, or 256 partials of a 256 x 256 matrix (mlp_wgrad.hip): 64 MiB
constexpr uint64_t kWgradWsFloats = 256ull * 256 * 256;
static_assert(kWgradWsFloats >= (uint64_t)kWgradMaxBlocks * kWgradMaxFloats, "weight-gradient workspace");
static inline uint64_t wgrad_ws_bytes() { return kWgradWsFloats * sizeof(float); }
static inline int wgrad_ws_open(void *ws, uint64_t bytes, WgradWs &out, const char *who) {
    if (ws == nullptr || bytes < wgrad_ws_bytes() || ((uintptr_t)ws & 15)) {
        lnh_set_error("%s: weight-gradient workspace missing, misaligned or too small (%llu bytes; lnh_wgrad_workspace_bytes() "
                      "= %llu, 16-byte aligned)", who, (unsigned long long)bytes, (unsigned long long)wgrad_ws_bytes());
        return LNH_ERR_INVALID_ARG;
    }
    out.partials = reinterpret_cast<float *>(ws);
    return LNH_OK;
}

// slot of this workgroup's partial sums (n floats, n % 4 == 0)
__device__ __forceinline__ float *wgrad_partial(const WgradWs &ws, uint32_t n) {
    return ws.partials + (size_t)blockIdx.x * n;
}

__device__ __forceinline__ void wgrad_add4(float *row_ptr, const float4 &s) {
    float4 *p = reinterpret_cast<float4 *>(row_ptr);
    float4 c = *p;
    c.x += s.x;
    c.y += s.y;
    c.z += s.z;
    c.w += s.w;
    *p = c;
}

// Where the sums go.  One 16 x 16 weight-gradient tile lives in a partial as 256 floats (r, lane): element r of lane
// 16 g + c is entry (row 4 g + r, column c) of the tile, so four consecutive floats of a partial are four consecutive COLUMNS
// of one row: a row-major gradient matrix takes them as one 16-byte read-add-store.  A partial is a sequence of tile ranges,
// one per matrix: tile `first + idx` of a range is the tile (idx / it, idx % it) of a row-major matrix with leading dimension
// ld (column tiles >= valid_it are padding and dropped; a stack of matrices [m][rows][ld] with rows = 16 * (tiles / it / m)
// is one range).
struct WgradRange {
    float *base;
    uint32_t first, it, valid_it, ld;
};
struct WgradTileMap {
    WgradRange r[3];  // ascending `first`; unused ranges: first = 0xffffffff
    __device__ __forceinline__ void operator()(uint32_t e4, const float4 &v) const {
        const uint32_t e = e4 * 4, tile = e >> 8, rr = (e >> 6) & 3, ln = e & 63, gg = ln >> 4, cc = ln & 15;
        const WgradRange &m = tile >= r[2].first ? r[2] : (tile >= r[1].first ? r[1] : r[0]);
        const uint32_t idx = tile - m.first, t = idx / m.it, i = idx % m.it;
        if (i < m.valid_it) wgrad_add4(m.base + (size_t)(16 * t + 4 * gg + rr) * m.ld + 16 * i + cc, v);
    }
};

namespace {
template <typename Emit>
__global__ void __launch_bounds__(256)
k_wgrad_reduce(const float4 *__restrict__ partials, uint32_t nb, uint32_t n4, Emit emit) {
    __shared__ float4 part[kWgradSlices][16];
    const uint32_t el = threadIdx.x & 15, s = threadIdx.x >> 4, e = blockIdx.x * 16 + el;
    const uint32_t per = (nb + kWgradSlices - 1) / kWgradSlices;
    const uint32_t j0 = s * per, j1 = min(nb, j0 + per);
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (e < n4) {
        constexpr uint32_t U = 8;
        uint32_t j = j0;
        for (; j + U <= j1; j += U) {
            float4 v[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) v[u] = partials[(size_t)(j + u) * n4 + e];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                acc.x += v[u].x;
                acc.y += v[u].y;
                acc.z += v[u].z;
                acc.w += v[u].w;
            }
        }
        for (; j < j1; j++) {
            const float4 v = partials[(size_t)j * n4 + e];
            acc.x += v.x;
            acc.y += v.y;
            acc.z += v.z;
            acc.w += v.w;
        }
    }
    part[s][el] = acc;
    __syncthreads();
    if (s == 0 && e < n4) {
        float4 t = part[0][el];
#pragma unroll
        for (uint32_t q = 1; q < kWgradSlices; q++) {
            t.x += part[q][el].x;
            t.y += part[q][el].y;
            t.z += part[q][el].z;
            t.w += part[q][el].w;
        }
        emit(e, t);
    }
}
}  // namespace

// the second launch: sums of `nb` partials of n floats each, handed to emit(e4, float4)
template <typename Emit>
static inline void wgrad_reduce_launch(const WgradWs &ws, uint32_t nb, uint32_t n, const Emit &emit, hipStream_t s) {
    const uint32_t n4 = n >> 2;
    hipLaunchKernelGGL((k_wgrad_reduce<Emit>), dim3((n4 + 15) / 16), dim3(256), 0, s,
                       reinterpret_cast<const float4 *>(ws.partials), nb, n4, emit);
}
