// convert.hip — LiDAR point cloud <-> range image ("pano") on the GPU: the step before (data preparation) and after
// (point-cloud export) the training path.  Semantics: lidarnerf/convert.py:99-160 (lidar_to_pano_with_intensities) and
// 194-237 (pano_to_lidar_with_intensities), which run as per-point Python loops / NumPy broadcasts on the host.
//
// lidar -> pano: the reference walks the points in order and keeps, per pixel, the nearest point (strict '>' — the
// first of equally near points wins).  Here every point does one 64-bit atomicMin on its pixel with the key
// (float bits of dist << 32 | point index): distances are positive, so their bit patterns order like the floats, and
// the index breaks ties towards the earlier point.  A second pass resolves keys into (dist, intensity).
// Arithmetic follows the reference's float32 NumPy scalars (angles in fp32, constants rounded to fp32 where NumPy's
// weak-scalar promotion rounds them); atan2 is evaluated in double and rounded to float, which reproduces a
// correctly-rounded atan2f.
#include "common.h"

namespace {

constexpr double kPi = 3.14159265358979323846;

struct PanoGeom {
    float pi_f;        // float32(np.pi)
    float down_rad;    // float32(fov_down / 180 * pi)
    float col_step;    // float32(2 pi / W)
    float row_step;    // float32(fov / 180 * pi / H)
    float max_depth;
    uint32_t H, W;
};

__device__ __forceinline__ float atan2_rn(float y, float x) { return (float)atan2((double)y, (double)x); }

__global__ void __launch_bounds__(256)
k_lidar_to_pano_keys(const float *__restrict__ pts, uint32_t N, PanoGeom g, unsigned long long *__restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float x = pts[(size_t)i * 4], y = pts[(size_t)i * 4 + 1], z = pts[(size_t)i * 4 + 2];
    const float dist = sqrtf(x * x + y * y + z * z);  // np.linalg.norm over 3 float32 values
    if (!(dist < g.max_depth) || !(dist > 0.0f)) return;  // reference: `if dist >= max_depth: continue`
    const float beta = g.pi_f - atan2_rn(y, x);
    const float alpha = atan2_rn(z, sqrtf(x * x + y * y)) + g.down_rad;
    const float cf = rintf(beta / g.col_step);                 // Python round(): half to even
    const float rf = rintf((float)g.H - alpha / g.row_step);
    if (!(rf >= 0.0f) || !(cf >= 0.0f) || rf >= (float)g.H || cf >= (float)g.W) return;
    const uint32_t r = (uint32_t)rf, c = (uint32_t)cf;
    const unsigned long long key = ((unsigned long long)__float_as_uint(dist) << 32) | i;
    atomicMin(&keys[(size_t)r * g.W + c], key);
}

__global__ void __launch_bounds__(256)
k_lidar_to_pano_resolve(const float *__restrict__ pts, const unsigned long long *__restrict__ keys, uint32_t HW,
                        float *__restrict__ pano, float *__restrict__ intens) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const unsigned long long k = keys[p];
    const bool hit = k != ~0ull;
    pano[p] = hit ? __uint_as_float((uint32_t)(k >> 32)) : 0.0f;
    intens[p] = hit ? pts[(size_t)(uint32_t)k * 4 + 3] : 0.0f;
}

// pano -> points (dense [H*W,4] + validity; the caller compacts in pixel order like np.where)
__global__ void __launch_bounds__(256)
k_pano_to_lidar(const float *__restrict__ pano, const float *__restrict__ intens, uint32_t H, uint32_t W, float fov_up,
                float fov, float *__restrict__ pts, uint8_t *__restrict__ valid) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const uint32_t j = p / W, i = p % W;
    // convert.py:207-217: float32 index grids; W/2, W, H are Python scalars (weak), 2*np.pi and 180 likewise
    const float beta = -((float)i - (float)(W / 2.0)) / (float)W * 2.0f * (float)kPi;
    const float alpha = (fov_up - (float)j / (float)H * fov) / 180.0f * (float)kPi;
    const float ca = cosf(alpha), sa = sinf(alpha), cb = cosf(beta), sb = sinf(beta);
    const float d = pano[p];
    pts[(size_t)p * 4] = ca * cb * d;
    pts[(size_t)p * 4 + 1] = ca * sb * d;
    pts[(size_t)p * 4 + 2] = sa * d;
    pts[(size_t)p * 4 + 3] = intens ? intens[p] : 0.0f;
    valid[p] = d != 0.0f;
}

}  // namespace

extern "C" {

int lnh_lidar_to_pano(const float *points, uint32_t N, uint32_t H, uint32_t W, float fov_up, float fov, float max_depth,
                      void *keys_scratch, float *pano, float *intensities, lnh_stream_t stream) {
    LNH_REQUIRE((points || N == 0) && keys_scratch && pano && intensities, LNH_ERR_INVALID_ARG, "lidar_to_pano: null pointer");
    LNH_REQUIRE(H >= 1 && W >= 1 && (uint64_t)H * W < 0xffffffffull, LNH_ERR_INVALID_ARG, "lidar_to_pano: bad image size");
    LNH_REQUIRE(fov > 0.0f, LNH_ERR_INVALID_ARG, "lidar_to_pano: fov must be positive");
    hipStream_t s = (hipStream_t)stream;
    (void)hipGetLastError();
    LNH_REQUIRE(hipMemsetAsync(keys_scratch, 0xff, (size_t)H * W * 8, s) == hipSuccess, LNH_ERR_LAUNCH,
                "lidar_to_pano: hipMemsetAsync failed");
    PanoGeom g;
    g.pi_f = (float)kPi;
    g.down_rad = (float)(((double)fov - (double)fov_up) / 180.0 * kPi);
    g.col_step = (float)(2.0 * kPi / (double)W);
    g.row_step = (float)((double)fov / 180.0 * kPi / (double)H);
    g.max_depth = max_depth;
    g.H = H;
    g.W = W;
    if (N) {
        LNH_LAUNCH(k_lidar_to_pano_keys, dim3(div_up(N, 256)), dim3(256), 0, s, points, N, g,
                   (unsigned long long *)keys_scratch);
        int rc = lnh_check_launch("lnh_lidar_to_pano(keys)");
        if (rc) return rc;
    }
    LNH_LAUNCH(k_lidar_to_pano_resolve, dim3(div_up(H * W, 256)), dim3(256), 0, s, points,
               (const unsigned long long *)keys_scratch, H * W, pano, intensities);
    return lnh_check_launch("lnh_lidar_to_pano(resolve)");
}

int lnh_pano_to_lidar(const float *pano, const float *intensities, uint32_t H, uint32_t W, float fov_up, float fov,
                      float *points, uint8_t *valid, lnh_stream_t stream) {
    LNH_REQUIRE(pano && points && valid, LNH_ERR_INVALID_ARG, "pano_to_lidar: null pointer");
    LNH_REQUIRE(H >= 1 && W >= 1 && (uint64_t)H * W < 0xffffffffull, LNH_ERR_INVALID_ARG, "pano_to_lidar: bad image size");
    LNH_LAUNCH(k_pano_to_lidar, dim3(div_up(H * W, 256)), dim3(256), 0, (hipStream_t)stream, pano, intensities, H, W,
               fov_up, fov, points, valid);
    return lnh_check_launch("lnh_pano_to_lidar");
}

}  // extern "C"
