// lidar_field.hip — fused kernels of the LiDAR radiance-field step on gfx950 (the part of the reference that lives in
// lidarnerf/nerf/renderer.py:149-231 + lidarnerf/nerf/network.py:199-237 as dozens of PyTorch launches):
//
//   lnh_lidar_sample_points   rays x z -> sample positions mapped to the encoder's [0,1]^3 (renderer.py:164-167 +
//                             gridencoder/grid.py:213) in one pass.
//   lnh_lidar_merge_weights   sigma gathered into merged (sorted) order through the resampler's permutation +
//                             compositing weights (renderer.py:217-243) — the [N,T+t] sort/gather never happens.
//   (lidar_color.hip)
//   lnh_lidar_color_forward   LiDAR colour head (ray-drop, intensity) for every merged sample whose weight exceeds
//   lnh_lidar_color_backward  1e-4 (renderer.py:249-256, network.py:199-237) as ONE MFMA kernel each way.
//
// Two algebraic moves keep the colour head small:
//   (1) its first layer sees [freq(d) (75) | geo_feat (15)], and the direction d is constant along a ray, so
//       W0 [freq(d) | geo] = W0_dir freq(d) + W0_geo geo: the 75-wide part is evaluated ONCE PER RAY (a [N,64]
//       fp32 bias computed by the caller) and the per-sample matmul shrinks from K=90 to K=16;
//   (2) the 16-wide sample input is the sigma-net's raw output row itself (col 0 = density pre-activation, cols 1..15
//       = geo_feat): the weight column for col 0 is zero, so no slicing / concatenation / masked gather is needed —
//       the row is fetched through the merge permutation straight from where the sigma-net wrote it.
// Backward: one workgroup per ray; its four waves recompute, back-propagate in registers, cooperate on the weight
// gradients through LDS (see mlp_bwd.h), and emit the FULL gradient row of the sigma-net output in point order
// (col 0 = trunc_exp backward of the compositing gradient, lidarnerf/activation.py:17-19), so the merge is undone for
// free.  The per-ray sum of d(hidden0) gives the gradient of the direction part of W0 after one small GEMM.
#include "common.h"
#include "lidar_steps.h"

namespace {

// ------------------------------------------------------------------------------------------------ sample points
__global__ void __launch_bounds__(256)
k_sample_points(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ z,
                const float *__restrict__ aabb, float bound, uint32_t N, uint32_t T, uint32_t T_tot, uint32_t slot_off,
                float *__restrict__ x01) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * T) return;
    const uint32_t n = i / T;
    const float t = z[i];
    float *o = x01 + ((size_t)n * T_tot + slot_off + (i - n * T)) * 3;  // row n*T_tot + slot_off + j
#pragma unroll
    for (int d = 0; d < 3; d++) {
        // o + d * z  (separate multiply and add: the reference evaluates this with two PyTorch ops), clip to the
        // aabb, then (x + bound) / (2 bound) as GridEncoder.forward does
        float p = rays_o[n * 3 + d] + rays_d[n * 3 + d] * t;
        p = fminf(fmaxf(p, aabb[d]), aabb[3 + d]);
        o[d] = (p + bound) / (2 * bound);
    }
}

// The coarse pass of a step in one kernel: the stratified depths of lnh_lidar_coarse_samples (same arithmetic, bit for
// bit) and the grid coordinates of those samples.
__global__ void __launch_bounds__(256)
k_coarse_sample_points(const float *__restrict__ u, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                       const float *__restrict__ aabb, float bound, uint32_t N, uint32_t T, uint32_t T_tot, float near,
                       float far, float *__restrict__ z, float *__restrict__ x01) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * T) return;
    coarse_sample_point(idx, u, rays_o, rays_d, aabb, bound, T, T_tot, near, far, z, x01);
}

// ------------------------------------------------------------------------------------------------ merge + weights
__global__ void __launch_bounds__(256)
k_merge_weights(const float *__restrict__ z, const float *__restrict__ sigma_pt, const int32_t *__restrict__ perm,
                const float *__restrict__ sample_dist, uint32_t N, uint32_t T, float density_scale,
                float *__restrict__ sigma_m, float *__restrict__ weights) {
    const int lane = threadIdx.x & 63;
    const uint32_t ray = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= N) return;
    const float *zr = z + (size_t)ray * T;
    const float *sp = sigma_pt + (size_t)ray * T;
    const int32_t *pr = perm + (size_t)ray * T;
    const float sd = sample_dist[ray];
    // kChunks 64-sample chunks are requested at once (perm, z, then the gathered sigma) and scanned from registers:
    // one dependent round trip per chunk made this one-wave-per-ray kernel run at HBM latency (see lidar_render.hip)
    constexpr int kChunks = 7;
    float carry = 1.0f;
    for (uint32_t base0 = 0; base0 < T; base0 += 64 * kChunks) {
        float zi[kChunks], zn[kChunks], sg[kChunks];
        int32_t pi[kChunks];
#pragma unroll
        for (int u = 0; u < kChunks; u++) {
            const uint32_t i = base0 + u * 64 + lane, ic = i < T ? i : 0, in = i + 1 < T ? i + 1 : ic;
            pi[u] = pr[ic];
            zi[u] = zr[ic];
            zn[u] = zr[in];
        }
#pragma unroll
        for (int u = 0; u < kChunks; u++) sg[u] = sp[pi[u]];
#pragma unroll
        for (int u = 0; u < kChunks; u++) {
            const uint32_t i = base0 + u * 64 + lane;
            if (base0 + u * 64 >= T) break;  // wave-uniform
            float alpha = 0.0f, om = 1.0f;
            if (i < T) {
                const float delta = (i + 1 < T) ? (zn[u] - zi[u]) : sd;
                alpha = 1.0f - expf(-delta * density_scale * sg[u]);
                om = 1.0f - alpha + 1e-15f;
            }
            const float incl = wave_scan_mul(om, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            if (i < T) {
                sigma_m[(size_t)ray * T + i] = sg[u];
                weights[(size_t)ray * T + i] = alpha * (carry * excl);
            }
            carry *= __shfl(incl, 63, 64);
        }
    }
}

}  // namespace

extern "C" {

int lnh_lidar_sample_points(const float *rays_o, const float *rays_d, const float *z, const float *aabb, float bound,
                            uint32_t N, uint32_t T, uint32_t T_tot, uint32_t slot_off, float *x01, lnh_stream_t stream) {
    LNH_REQUIRE(rays_o && rays_d && z && aabb && x01, LNH_ERR_INVALID_ARG, "lidar_sample_points: null pointer");
    LNH_REQUIRE(bound > 0.0f, LNH_ERR_INVALID_ARG, "lidar_sample_points: bound must be positive");
    LNH_REQUIRE(slot_off + T <= T_tot, LNH_ERR_INVALID_ARG, "lidar_sample_points: slot_off + T must be <= T_tot");
    if ((uint64_t)N * T == 0) return LNH_OK;
    LNH_REQUIRE((uint64_t)N * T < 0xffffffffull, LNH_ERR_UNSUPPORTED, "lidar_sample_points: N*T must fit 32 bits");
    LNH_LAUNCH(k_sample_points, dim3(div_up((uint64_t)N * T, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, z,
               aabb, bound, N, T, T_tot, slot_off, x01);
    return lnh_check_launch("lnh_lidar_sample_points");
}

int lnh_lidar_coarse_sample_points(const float *u, const float *rays_o, const float *rays_d, const float *aabb, float bound,
                                   uint32_t N, uint32_t T, uint32_t T_tot, float near, float far, float *z, float *x01,
                                   lnh_stream_t stream) {
    LNH_REQUIRE(rays_o && rays_d && aabb && z && x01, LNH_ERR_INVALID_ARG, "lidar_coarse_sample_points: null pointer");
    LNH_REQUIRE(bound > 0.0f, LNH_ERR_INVALID_ARG, "lidar_coarse_sample_points: bound must be positive");
    LNH_REQUIRE(T <= T_tot, LNH_ERR_INVALID_ARG, "lidar_coarse_sample_points: T must be <= T_tot");
    if ((uint64_t)N * T == 0) return LNH_OK;
    LNH_REQUIRE((uint64_t)N * T_tot < 0xffffffffull, LNH_ERR_UNSUPPORTED, "lidar_coarse_sample_points: N*T must fit 32 bits");
    LNH_LAUNCH(k_coarse_sample_points, dim3(div_up((uint64_t)N * T, 256)), dim3(256), 0, (hipStream_t)stream, u, rays_o,
               rays_d, aabb, bound, N, T, T_tot, near, far, z, x01);
    return lnh_check_launch("lnh_lidar_coarse_sample_points");
}

int lnh_lidar_merge_weights(const float *z, const float *sigma_pt, const int32_t *perm, const float *sample_dist,
                            uint32_t N, uint32_t T, float density_scale, float *sigma_m, float *weights,
                            lnh_stream_t stream) {
    LNH_REQUIRE(z && sigma_pt && perm && sample_dist && sigma_m && weights, LNH_ERR_INVALID_ARG,
                "lidar_merge_weights: null pointer");
    if (N == 0 || T == 0) return LNH_OK;
    LNH_LAUNCH(k_merge_weights, dim3(div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, z, sigma_pt, perm, sample_dist, N,
               T, density_scale, sigma_m, weights);
    return lnh_check_launch("lnh_lidar_merge_weights");
}

}  // extern "C"
