// metrics.hip — nearest-neighbour pass of the chamfer distance used by the reference's evaluation (PointsMeter,
// lidarnerf/nerf/utils.py:377-413 -> extern/chamfer3D/chamfer3D.cu:9-138): for every point of cloud 1 the SQUARED
// distance to, and the index of, its nearest point in cloud 2 (first index on ties), brute force O(n*m).
// One lane per query point; cloud 2 streams through LDS in 1024-point tiles (every lane reads the same LDS address:
// broadcast), so the kernel is pure VALU: 8 flops per pair.
#include "common.h"

namespace {

constexpr int kTile = 1024;

__global__ void __launch_bounds__(256)
k_chamfer_nn(const float *__restrict__ xyz1, uint32_t n, const float *__restrict__ xyz2, uint32_t m,
             float *__restrict__ dist, int32_t *__restrict__ idx) {
    __shared__ float buf[kTile * 3];
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t jc = j < n ? j : (n ? n - 1 : 0);
    const float x1 = xyz1[(size_t)jc * 3], y1 = xyz1[(size_t)jc * 3 + 1], z1 = xyz1[(size_t)jc * 3 + 2];
    float best = 0.0f;
    int32_t best_i = 0;
    for (uint32_t k0 = 0; k0 < m; k0 += kTile) {
        const uint32_t cnt = min((uint32_t)kTile, m - k0);
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < cnt * 3; t += blockDim.x) buf[t] = xyz2[(size_t)k0 * 3 + t];
        __syncthreads();
        for (uint32_t k = 0; k < cnt; k++) {
            const float dx = buf[k * 3] - x1, dy = buf[k * 3 + 1] - y1, dz = buf[k * 3 + 2] - z1;
            // x*x + y*y + z*z as the reference's compiler contracts it (two fused multiply-adds)
            const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if ((k0 == 0 && k == 0) || d < best) {
                best = d;
                best_i = (int32_t)(k0 + k);
            }
        }
    }
    if (j < n) {
        dist[j] = best;
        idx[j] = best_i;
    }
}

}  // namespace

extern "C" {

int lnh_chamfer_nn(const float *xyz1, uint32_t n, const float *xyz2, uint32_t m, float *dist, int32_t *idx,
                   lnh_stream_t stream) {
    LNH_REQUIRE(dist && idx && (xyz1 || n == 0) && (xyz2 || m == 0), LNH_ERR_INVALID_ARG, "chamfer_nn: null pointer");
    if (n == 0) return LNH_OK;
    LNH_REQUIRE(m > 0, LNH_ERR_INVALID_ARG, "chamfer_nn: the second cloud is empty");
    LNH_LAUNCH(k_chamfer_nn, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, xyz1, n, xyz2, m, dist, idx);
    return lnh_check_launch("lnh_chamfer_nn");
}

}  // extern "C"
