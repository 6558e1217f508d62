// lidar_ragged.hip — element-wise stages of the occupancy-grid render chain (nerf/fused.py FusedLidarRagged; BASELINE
// config 4) between the marcher, the encoder, the MFMA MLP kernels and the ragged compositing.  The marched samples of a
// step are a flat [M] list (M ~ 10^5 .. 10^6), the step is launch-bound: each stage is ONE launch here where tensor
// expressions would be three to six (network.py:162-237 on flat sample lists: cat / cast / slice / sigmoid).
//   lnh_ragged_points           x01 = (xyz + bound) / (2 bound)                      (gridencoder/grid.py:213)
//   lnh_ragged_pack_weights     fp32 master matrices -> flat 16-bit vectors: [ws0 | ws1], [wc0 padded to 96 cols | wc1 | wc2 padded to 16 rows]
//   lnh_ragged_color_input      [freq(d) (3 + 6 deg) | geo_feat = h16[:, 1:16] | 0] -> [M, 96]   (network.py:215-221)
//   lnh_ragged_color_input_rays the same rows written ray by ray (the direction terms once per ray)
//   lnh_ragged_color_output     rgb = sigmoid(y[:, :2])                                (network.py:224-231)
//   lnh_ragged_color_output_backward   d/dy of the above into a zero-padded [M, 16] row
//   lnh_ragged_grad_rows        gradient row of the sigma-net output: col 0 = g_sigma * density_scale * exp(clamp(h0, -15, 15))
//                               (trunc_exp backward, activation.py:17-19), cols 1..15 = colour head's d/d geo_feat
#include "common.h"
#include <algorithm>

namespace {

__global__ void __launch_bounds__(256)
k_ragged_points(const float *__restrict__ xyz, float bound, uint32_t n3, float *__restrict__ x01) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) x01[i] = (xyz[i] + bound) / (2 * bound);
}

template <typename E>
struct RaggedPackArgs {
    const float *ws0, *ws1, *wc0, *wc1, *wc2;
    uint32_t ld_s0, ld_s1, ld_c0, ld_c1, ld_c2, kin;  // kin = used columns of wc0 (n_dir + 15 <= 96)
    E *wsig, *wcol;
};
template <typename E>
__global__ void __launch_bounds__(256)
k_ragged_pack_weights(RaggedPackArgs<E> a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr uint32_t nS0 = 64 * 32, nS1 = 16 * 64, nC0 = 64 * 96, nC1 = 64 * 64, nC2 = 16 * 64;
    if (i < nS0) a.wsig[i] = (E)a.ws0[(i / 32) * a.ld_s0 + i % 32];
    else if (i < nS0 + nS1) a.wsig[i] = (E)a.ws1[((i - nS0) / 64) * a.ld_s1 + (i - nS0) % 64];
    if (i < nC0) {
        const uint32_t o = i / 96, c = i % 96;
        a.wcol[i] = c < a.kin ? (E)a.wc0[o * a.ld_c0 + c] : (E)0.0f;
    } else if (i < nC0 + nC1) {
        const uint32_t j = i - nC0;
        a.wcol[i] = (E)a.wc1[(j / 64) * a.ld_c1 + j % 64];
    } else if (i < nC0 + nC1 + nC2) {
        const uint32_t j = i - nC0 - nC1, o = j / 64;
        a.wcol[i] = o < 2 ? (E)a.wc2[o * a.ld_c2 + j % 64] : (E)0.0f;
    }
}

// one thread per (sample, column pair): 48 threads per sample write 96 columns as packed pairs
template <typename E>
__global__ void __launch_bounds__(256)
k_ragged_color_input(const float *__restrict__ dirs, const E *__restrict__ h16, uint32_t M, uint32_t kd,
                     E *__restrict__ cin) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t m = t / 48, c2 = (t % 48) * 2;
    if (m >= M) return;
    E out[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const uint32_t k = c2 + u;
        float v = 0.0f;
        if (k < kd) {  // frequency features, same operations in the same order as encoders.hip k_freq_forward
            const uint32_t q = k < 3 ? 0u : k - 3u, d = k < 3 ? k : q % 3u, f = q / 6u, c = (q / 3u) & 1u;
            const float x = dirs[(size_t)m * 3 + d];
            const float a = scalbnf(x, (int)f);
            v = k < 3 ? x : sinf(c ? a + 1.5707963267948966f : a);
        } else if (k < kd + 15) {
            v = (float)h16[(size_t)m * 16 + 1 + (k - kd)];
        }
        out[u] = (E)v;
    }
    cin[(size_t)m * 96 + c2] = out[0];
    cin[(size_t)m * 96 + c2 + 1] = out[1];
}

// The same rows, RAY BY RAY (one workgroup each): every marched sample of a ray carries the ray's direction (raymarching.cu:331-534 writes
// dirs = the ray's d for each of its samples), so the 2 * 6 * degree sines are evaluated once per ray instead of once per
// sample (the per-sample kernel above spends its 49 us per step of the NeRF-MVL-shaped bench on 312 K x 72 sinf).  48 lanes
// hold one column pair each; a wave walks every fourth row of the ray writing 192 contiguous bytes per row.  Rows no ray owns — the
// unused tail of the sample buffer, and the slots of a ray the marcher dropped for lack of room — still enter the weight
// gradient GEMM (with a zero output gradient), so they must hold finite numbers: the blocks behind the ray blocks find them
// by their delta == 0 (the marcher never wrote them; a marched sample has dt > 0) and zero them.
template <typename E>
__global__ void __launch_bounds__(256)
k_ragged_color_input_rays(const float *__restrict__ dirs, const E *__restrict__ h16, const int32_t *__restrict__ rays,
                          const float *__restrict__ deltas, uint32_t N, uint32_t M, uint32_t kd, uint32_t ray_blocks,
                          E *__restrict__ cin) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    struct alignas(4) Pair { E a, b; };
    if (blockIdx.x < ray_blocks) {
        const uint32_t n = blockIdx.x;  // one workgroup per ray: its four waves take every fourth row (a lone wave walking
        if (lane >= 48) return;         // ~80 rows is a chain of ~10 load round trips)
        const uint32_t offset = (uint32_t)rays[n * 3 + 1], count = (uint32_t)rays[n * 3 + 2];
        if (count == 0 || offset + count > M) return;  // (a dropped ray's slots: zeroed by the scanning blocks)
        const uint32_t c2 = lane * 2;
        float fv[2] = {0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t k = c2 + u;
            if (k < kd) {  // frequency features, same operations in the same order as encoders.hip k_freq_forward
                const uint32_t q = k < 3 ? 0u : k - 3u, d = k < 3 ? k : q % 3u, f = q / 6u, c = (q / 3u) & 1u;
                const float x = dirs[(size_t)offset * 3 + d];
                const float a = scalbnf(x, (int)f);
                fv[u] = k < 3 ? x : sinf(c ? a + 1.5707963267948966f : a);
            }
        }
        const bool g0 = c2 >= kd && c2 < kd + 15, g1 = c2 + 1 >= kd && c2 + 1 < kd + 15;
        const uint32_t h0 = 1 + (c2 - kd), h1 = 1 + (c2 + 1 - kd);
        Pair fixed{(E)fv[0], (E)fv[1]};
#pragma unroll 4  // (independent rows: the loads of several in flight)
        for (uint32_t j = wv; j < count; j += 4) {
            const size_t row = (size_t)offset + j;
            Pair o = fixed;
            if (g0) o.a = h16[row * 16 + h0];
            if (g1) o.b = h16[row * 16 + h1];
            *reinterpret_cast<Pair *>(cin + row * 96 + c2) = o;
        }
        return;
    }
    // rows nobody marched: 64 rows per wave at a time
    const uint32_t waves = (gridDim.x - ray_blocks) * 4, w = (blockIdx.x - ray_blocks) * 4 + wv;
    for (uint32_t r0 = w * 64; r0 < M; r0 += waves * 64) {
        const uint32_t r = r0 + lane;
        unsigned long long empty = __ballot(r < M && deltas[(size_t)r * 2] == 0.0f);
        while (empty) {
            const uint32_t b = (uint32_t)__ffsll(empty) - 1;
            empty &= empty - 1;
            if (lane < 48) *reinterpret_cast<Pair *>(cin + (size_t)(r0 + b) * 96 + lane * 2) = Pair{(E)0.0f, (E)0.0f};
        }
    }
}

template <typename E>
__global__ void __launch_bounds__(256)
k_ragged_color_output(const E *__restrict__ y, uint32_t M, float *__restrict__ rgb) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * 2) return;
    const float v = (float)y[(size_t)(i >> 1) * 16 + (i & 1)];
    rgb[i] = 1.0f / (1.0f + expf(-v));
}

template <typename E>
__global__ void __launch_bounds__(256)
k_ragged_color_output_bwd(const float *__restrict__ g_rgb, const float *__restrict__ rgb, uint32_t M, E *__restrict__ gy) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * 16) return;
    const uint32_t m = i >> 4, c = i & 15;
    float v = 0.0f;
    if (c < 2) {
        const float s = rgb[m * 2 + c];
        v = g_rgb[m * 2 + c] * s * (1.0f - s);
    }
    gy[i] = (E)v;
}

template <typename E>
__global__ void __launch_bounds__(256)
k_ragged_grad_rows(const float *__restrict__ g_sigma, float density_scale, const E *__restrict__ h16,
                   const E *__restrict__ gx, uint32_t kd, uint32_t M, E *__restrict__ g_h16) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * 16) return;
    const uint32_t m = i >> 4, c = i & 15;
    float v;
    if (c == 0) {
        const float h0 = fminf(fmaxf((float)h16[(size_t)m * 16], -15.0f), 15.0f);
        v = g_sigma[m] * density_scale * expf(h0);
    } else {
        v = (float)gx[(size_t)m * 96 + kd + c - 1];
    }
    g_h16[i] = (E)v;
}

template <typename E>
int pack(const float *ws0, uint32_t ld_s0, const float *ws1, uint32_t ld_s1, const float *wc0, uint32_t ld_c0, uint32_t kin,
         const float *wc1, uint32_t ld_c1, const float *wc2, uint32_t ld_c2, void *wsig, void *wcol, lnh_stream_t stream) {
    LNH_REQUIRE(ws0 && ws1 && wc0 && wc1 && wc2 && wsig && wcol, LNH_ERR_INVALID_ARG, "ragged_pack_weights: null pointer");
    LNH_REQUIRE(kin <= 96 && ld_s0 >= 32 && ld_s1 >= 64 && ld_c0 >= kin && ld_c1 >= 64 && ld_c2 >= 64, LNH_ERR_INVALID_ARG,
                "ragged_pack_weights: need n_in <= 96 and leading dimensions >= the rows");
    RaggedPackArgs<E> a{ws0, ws1, wc0, wc1, wc2, ld_s0, ld_s1, ld_c0, ld_c1, ld_c2, kin, (E *)wsig, (E *)wcol};
    LNH_LAUNCH(k_ragged_pack_weights<E>, dim3(div_up(64 * 96 + 64 * 64 + 16 * 64, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return lnh_check_launch("lnh_ragged_pack_weights");
}
template <typename E>
int color_input(const float *dirs, const void *h16, uint32_t M, uint32_t degree, void *cin, lnh_stream_t stream) {
    LNH_REQUIRE(dirs && h16 && cin, LNH_ERR_INVALID_ARG, "ragged_color_input: null pointer");
    const uint32_t kd = 3 + 6 * degree;
    LNH_REQUIRE(kd + 15 <= 96, LNH_ERR_UNSUPPORTED, "ragged_color_input: 3 + 6 * degree + 15 must fit 96 columns");
    LNH_REQUIRE((uint64_t)M * 48 < 0xffffffffull, LNH_ERR_UNSUPPORTED, "ragged_color_input: M too large");
    if (M == 0) return LNH_OK;
    LNH_LAUNCH(k_ragged_color_input<E>, dim3(div_up((uint64_t)M * 48, 256)), dim3(256), 0, (hipStream_t)stream, dirs,
               (const E *)h16, M, kd, (E *)cin);
    return lnh_check_launch("lnh_ragged_color_input");
}
template <typename E>
int color_input_rays(const float *dirs, const void *h16, const int32_t *rays, const float *deltas, uint32_t N, uint32_t M,
                     uint32_t degree, void *cin, lnh_stream_t stream) {
    LNH_REQUIRE(dirs && h16 && rays && deltas && cin, LNH_ERR_INVALID_ARG, "ragged_color_input_rays: null pointer");
    const uint32_t kd = 3 + 6 * degree;
    LNH_REQUIRE(kd + 15 <= 96, LNH_ERR_UNSUPPORTED, "ragged_color_input_rays: 3 + 6 * degree + 15 must fit 96 columns");
    LNH_REQUIRE((uint64_t)M * 96 < (1ull << 40), LNH_ERR_UNSUPPORTED, "ragged_color_input_rays: M too large");
    if (M == 0) return LNH_OK;
    const uint32_t ray_blocks = N, scan_blocks = std::min(div_up(M, 256), 512u);
    LNH_LAUNCH(k_ragged_color_input_rays<E>, dim3(ray_blocks + scan_blocks), dim3(256), 0, (hipStream_t)stream, dirs,
               (const E *)h16, rays, deltas, N, M, kd, ray_blocks, (E *)cin);
    return lnh_check_launch("lnh_ragged_color_input_rays");
}
template <typename E>
int color_output(const void *y, uint32_t M, float *rgb, lnh_stream_t stream) {
    LNH_REQUIRE(y && rgb, LNH_ERR_INVALID_ARG, "ragged_color_output: null pointer");
    if (M == 0) return LNH_OK;
    LNH_LAUNCH(k_ragged_color_output<E>, dim3(div_up((uint64_t)M * 2, 256)), dim3(256), 0, (hipStream_t)stream, (const E *)y,
               M, rgb);
    return lnh_check_launch("lnh_ragged_color_output");
}
template <typename E>
int color_output_bwd(const float *g_rgb, const float *rgb, uint32_t M, void *gy, lnh_stream_t stream) {
    LNH_REQUIRE(g_rgb && rgb && gy, LNH_ERR_INVALID_ARG, "ragged_color_output_backward: null pointer");
    LNH_REQUIRE((uint64_t)M * 16 < 0xffffffffull, LNH_ERR_UNSUPPORTED, "ragged_color_output_backward: M too large");
    if (M == 0) return LNH_OK;
    LNH_LAUNCH(k_ragged_color_output_bwd<E>, dim3(div_up((uint64_t)M * 16, 256)), dim3(256), 0, (hipStream_t)stream, g_rgb,
               rgb, M, (E *)gy);
    return lnh_check_launch("lnh_ragged_color_output_backward");
}
template <typename E>
int grad_rows(const float *g_sigma, float density_scale, const void *h16, const void *gx, uint32_t degree, uint32_t M,
              void *g_h16, lnh_stream_t stream) {
    LNH_REQUIRE(g_sigma && h16 && gx && g_h16, LNH_ERR_INVALID_ARG, "ragged_grad_rows: null pointer");
    const uint32_t kd = 3 + 6 * degree;
    LNH_REQUIRE(kd + 15 <= 96, LNH_ERR_UNSUPPORTED, "ragged_grad_rows: 3 + 6 * degree + 15 must fit 96 columns");
    LNH_REQUIRE((uint64_t)M * 16 < 0xffffffffull, LNH_ERR_UNSUPPORTED, "ragged_grad_rows: M too large");
    if (M == 0) return LNH_OK;
    LNH_LAUNCH(k_ragged_grad_rows<E>, dim3(div_up((uint64_t)M * 16, 256)), dim3(256), 0, (hipStream_t)stream, g_sigma,
               density_scale, (const E *)h16, (const E *)gx, kd, M, (E *)g_h16);
    return lnh_check_launch("lnh_ragged_grad_rows");
}

}  // namespace

extern "C" {

int lnh_ragged_points(const float *xyz, float bound, uint32_t M, float *x01, lnh_stream_t stream) {
    LNH_REQUIRE(xyz && x01, LNH_ERR_INVALID_ARG, "ragged_points: null pointer");
    LNH_REQUIRE(bound > 0.0f, LNH_ERR_INVALID_ARG, "ragged_points: bound must be positive");
    LNH_REQUIRE((uint64_t)M * 3 < 0xffffffffull, LNH_ERR_UNSUPPORTED, "ragged_points: M too large");
    if (M == 0) return LNH_OK;
    LNH_LAUNCH(k_ragged_points, dim3(div_up((uint64_t)M * 3, 256)), dim3(256), 0, (hipStream_t)stream, xyz, bound, M * 3, x01);
    return lnh_check_launch("lnh_ragged_points");
}

#define LNH_RAGGED_API(SFX, E)                                                                                               \
    int lnh_ragged_pack_weights##SFX(const float *ws0, uint32_t ld_s0, const float *ws1, uint32_t ld_s1, const float *wc0,  \
                                     uint32_t ld_c0, uint32_t n_in, const float *wc1, uint32_t ld_c1, const float *wc2,     \
                                     uint32_t ld_c2, void *wsig16, void *wcol16, lnh_stream_t stream) {                      \
        return pack<E>(ws0, ld_s0, ws1, ld_s1, wc0, ld_c0, n_in, wc1, ld_c1, wc2, ld_c2, wsig16, wcol16, stream);           \
    }                                                                                                                        \
    int lnh_ragged_color_input##SFX(const float *dirs, const void *h16, uint32_t M, uint32_t degree, void *cin,             \
                                    lnh_stream_t stream) {                                                                   \
        return color_input<E>(dirs, h16, M, degree, cin, stream);                                                            \
    }                                                                                                                        \
    int lnh_ragged_color_input_rays##SFX(const float *dirs, const void *h16, const int32_t *rays, const float *deltas,      \
                                         uint32_t N, uint32_t M, uint32_t degree, void *cin, lnh_stream_t stream) {          \
        return color_input_rays<E>(dirs, h16, rays, deltas, N, M, degree, cin, stream);                                      \
    }                                                                                                                        \
    int lnh_ragged_color_output##SFX(const void *y16, uint32_t M, float *rgb, lnh_stream_t stream) {                         \
        return color_output<E>(y16, M, rgb, stream);                                                                         \
    }                                                                                                                        \
    int lnh_ragged_color_output_backward##SFX(const float *grad_rgb, const float *rgb, uint32_t M, void *grad_y16,          \
                                              lnh_stream_t stream) {                                                         \
        return color_output_bwd<E>(grad_rgb, rgb, M, grad_y16, stream);                                                      \
    }                                                                                                                        \
    int lnh_ragged_grad_rows##SFX(const float *grad_sigma, float density_scale, const void *h16, const void *grad_cin,      \
                                  uint32_t degree, uint32_t M, void *grad_h16, lnh_stream_t stream) {                        \
        return grad_rows<E>(grad_sigma, density_scale, h16, grad_cin, degree, M, grad_h16, stream);                          \
    }
LNH_RAGGED_API(, half_t)
LNH_RAGGED_API(_bf16, __bf16)
#undef LNH_RAGGED_API

}  // extern "C"
