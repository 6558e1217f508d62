// lidar_steps.h — device bodies shared by the stand-alone entry points (lidar_field.hip, lidar_glue.hip) and the merged
// step prologue (lidar_glue.hip k_step_prologue): one definition of the arithmetic, whichever launch runs it.
#pragma once
#include "common.h"

// The coarse pass of a step for sample `idx` of N x T: the stratified depth of lnh_lidar_coarse_samples (renderer.py:147-161;
// torch.linspace evaluated symmetrically) and the grid coordinates of that sample (renderer.py:164-167, grid.py:213).
__device__ __forceinline__ void coarse_sample_point(uint32_t idx, const float *__restrict__ u, const float *__restrict__ rays_o,
                                                    const float *__restrict__ rays_d, const float *__restrict__ aabb,
                                                    float bound, uint32_t T, uint32_t T_tot, float near, float far,
                                                    float *__restrict__ z, float *__restrict__ x01) {
    const uint32_t n = idx / T, i = idx - n * T;
    const float step = T > 1 ? 1.0f / (float)(T - 1) : 0.0f;
    const float lin = i < T / 2 ? step * (float)i : 1.0f - step * (float)(T - 1 - i);
    float t = near + (far - near) * lin;
    if (u) t = t + (u[idx] - 0.5f) * ((far - near) / (float)T);
    z[idx] = t;
    float *o = x01 + ((size_t)n * T_tot + i) * 3;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        // o + d * z  (separate multiply and add: the reference evaluates this with two PyTorch ops), clip to the
        // aabb, then (x + bound) / (2 bound) as GridEncoder.forward does
        float p = rays_o[n * 3 + d] + rays_d[n * 3 + d] * t;
        p = fminf(fmaxf(p, aabb[d]), aabb[3 + d]);
        o[d] = (p + bound) / (2 * bound);
    }
}
