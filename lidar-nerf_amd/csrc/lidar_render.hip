// lidar_render.hip — per-ray volumetric compositing / importance resampling for LiDAR range-image rays (gfx950).
//
// The reference does this part in ~40 PyTorch launches over [N, T] tensors (lidarnerf/nerf/renderer.py:180-271,
// sample_pdf at 10-46).  Here one 64-lane wavefront owns one ray:
//   * samples are consumed in rounds of 64 (lane = sample within the round), so every global load/store of the
//     wave is one contiguous 256 B segment;
//   * the transmittance prod_{j<i}(1 - alpha_j + 1e-15) is an inclusive product scan inside the wave (shuffle
//     ladder) with a scalar carry between rounds — no [N,T+1] cumprod tensor ever reaches HBM;
//   * depth / image / weight sums are lane-partial accumulators reduced once per ray;
//   * backward recomputes the scan and uses  dL/dalpha_i = T_i c_i - (sum_{j>i} w_j c_j) / (1 - alpha_i + 1e-15),
//     the same quotient form torch.cumprod's autograd uses; it includes the DEPTH gradient (the reference's CUDA
//     compositor drops it, raymarching.py:330, but the LiDAR loss is depth dominated and the PyTorch path keeps it);
//   * resampling (stage-1 weights -> pdf -> cdf -> inverse-cdf lookup -> sort -> merge) keeps the ray's cdf and z in
//     LDS; the merge is rank arithmetic on two sorted runs, not a 832-element sort.
#include "common.h"

namespace {

constexpr float kEps = 1e-15f;  // renderer.py:189,241

struct RoundVals {
    float z, alpha, om, w;
};

// alpha/omega of sample i (renderer.py:233-241)
__device__ __forceinline__ void sample_alpha(const float *__restrict__ z, const float *__restrict__ sigma, uint32_t i,
                                             uint32_t T, float sample_dist, float density_scale, float &zi,
                                             float &delta, float &alpha, float &om, float &e) {
    zi = z[i];
    const float zn = (i + 1 < T) ? z[i + 1] : 0.0f;
    delta = (i + 1 < T) ? (zn - zi) : sample_dist;
    e = expf(-delta * density_scale * sigma[i]);
    alpha = 1.0f - e;
    om = 1.0f - alpha + kEps;
}

// The same from values already in registers (zn = z[i+1], ignored for the last sample).
__device__ __forceinline__ void alpha_of(float zi, float zn, bool has_next, float sg, float sample_dist,
                                         float density_scale, float &delta, float &alpha, float &om, float &e) {
    delta = has_next ? (zn - zi) : sample_dist;
    e = expf(-delta * density_scale * sg);
    alpha = 1.0f - e;
    om = 1.0f - alpha + kEps;
}
// The per-ray kernels below are one wave per ray walking the ray in 64-sample chunks with a scan carry: left as a plain
// loop every chunk pays its own HBM round trip (13 dependent ones at 832 samples — the kernels ran at that latency, not
// at bandwidth).  They therefore request kChunks chunks at once and scan them from registers.
constexpr int kChunks = 7;

// ------------------------------------------------------------------------------------------------ weights only
__global__ void __launch_bounds__(256)
k_lidar_weights(const float *__restrict__ z, const float *__restrict__ sigma, const float *__restrict__ sample_dist,
                uint32_t N, uint32_t T, float density_scale, float *__restrict__ weights) {
    const int lane = threadIdx.x & 63;
    const uint32_t ray = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= N) return;
    const float *zr = z + (size_t)ray * T, *sr = sigma + (size_t)ray * T;
    float *wr = weights + (size_t)ray * T;
    const float sd = sample_dist[ray];
    float carry = 1.0f;
    for (uint32_t base = 0; base < T; base += 64) {
        const uint32_t i = base + lane;
        float zi = 0, delta, alpha = 0.0f, om = 1.0f, e;
        if (i < T) sample_alpha(zr, sr, i, T, sd, density_scale, zi, delta, alpha, om, e);
        const float incl = wave_scan_mul(om, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        if (i < T) wr[i] = alpha * (carry * excl);
        carry *= __shfl(incl, 63, 64);
    }
}

// ------------------------------------------------------------------------------------------------ composite fwd
template <int K>
__global__ void __launch_bounds__(256)
k_lidar_composite_fwd(const float *__restrict__ z, const float *__restrict__ sigma, const float *__restrict__ rgb,
                      const float *__restrict__ sample_dist, uint32_t N, uint32_t T, float density_scale,
                      float *__restrict__ weights, float *__restrict__ weights_sum, float *__restrict__ depth,
                      float *__restrict__ image) {
    const int lane = threadIdx.x & 63;
    const uint32_t ray = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= N) return;
    const float *zr = z + (size_t)ray * T, *sr = sigma + (size_t)ray * T;
    const float *cr = rgb + (size_t)ray * T * K;
    float *wr = weights ? weights + (size_t)ray * T : nullptr;
    const float sd = sample_dist[ray];
    float carry = 1.0f, ws = 0.0f, dep = 0.0f, img[K];
#pragma unroll
    for (int k = 0; k < K; k++) img[k] = 0.0f;
    for (uint32_t base0 = 0; base0 < T; base0 += 64 * kChunks) {
        float zi[kChunks], zn[kChunks], sg[kChunks], c[kChunks][K];
#pragma unroll
        for (int u = 0; u < kChunks; u++) {
            const uint32_t i = base0 + u * 64 + lane, ic = i < T ? i : 0, in = i + 1 < T ? i + 1 : ic;
            zi[u] = zr[ic];
            zn[u] = zr[in];
            sg[u] = sr[ic];
#pragma unroll
            for (int k = 0; k < K; k++) c[u][k] = cr[(size_t)ic * K + k];
        }
#pragma unroll
        for (int u = 0; u < kChunks; u++) {
            const uint32_t i = base0 + u * 64 + lane;
            if (base0 + u * 64 >= T) break;  // wave-uniform
            float delta, alpha = 0.0f, om = 1.0f, e;
            if (i < T) alpha_of(zi[u], zn[u], i + 1 < T, sg[u], sd, density_scale, delta, alpha, om, e);
            const float incl = wave_scan_mul(om, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            const float w = alpha * (carry * excl);
            if (i < T && wr) wr[i] = w;
            ws += w;
            dep += i < T ? w * zi[u] : 0.0f;
#pragma unroll
            for (int k = 0; k < K; k++) img[k] += i < T ? w * c[u][k] : 0.0f;
            carry *= __shfl(incl, 63, 64);
        }
    }
    ws = wave_sum(ws);
    dep = wave_sum(dep);
#pragma unroll
    for (int k = 0; k < K; k++) img[k] = wave_sum(img[k]);
    if (lane == 0) {
        weights_sum[ray] = ws;
        depth[ray] = dep;
#pragma unroll
        for (int k = 0; k < K; k++) image[(size_t)ray * K + k] = img[k];
    }
}

// ------------------------------------------------------------------------------------------------ composite bwd
template <int K>
__global__ void __launch_bounds__(256)
k_lidar_composite_bwd(const float *__restrict__ g_ws, const float *__restrict__ g_depth,
                      const float *__restrict__ g_image, const float *__restrict__ z,
                      const float *__restrict__ sigma, const float *__restrict__ rgb,
                      const float *__restrict__ sample_dist, uint32_t N, uint32_t T, float density_scale,
                      float *__restrict__ grad_sigma, float *__restrict__ grad_rgb) {
    const int lane = threadIdx.x & 63;
    const uint32_t ray = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= N) return;
    const float *zr = z + (size_t)ray * T, *sr = sigma + (size_t)ray * T;
    const float *cr = rgb + (size_t)ray * T * K;
    float *gs = grad_sigma + (size_t)ray * T;
    float *gc = grad_rgb ? grad_rgb + (size_t)ray * T * K : nullptr;
    const float sd = sample_dist[ray];
    const float gws = g_ws ? g_ws[ray] : 0.0f, gdp = g_depth ? g_depth[ray] : 0.0f;
    float gim[K];
#pragma unroll
    for (int k = 0; k < K; k++) gim[k] = g_image ? g_image[(size_t)ray * K + k] : 0.0f;

    auto load_chunks = [&](uint32_t base0, float (&zi)[kChunks], float (&zn)[kChunks], float (&sg)[kChunks],
                           float (&ci)[kChunks]) {
#pragma unroll
        for (int u = 0; u < kChunks; u++) {
            const uint32_t i = base0 + u * 64 + lane, ic = i < T ? i : 0, in = i + 1 < T ? i + 1 : ic;
            zi[u] = zr[ic];
            zn[u] = zr[in];
            sg[u] = sr[ic];
            float c[K];
#pragma unroll
            for (int k = 0; k < K; k++) c[k] = cr[(size_t)ic * K + k];
            ci[u] = gws + gdp * zi[u];
#pragma unroll
            for (int k = 0; k < K; k++) ci[u] += gim[k] * c[k];
        }
    };
    // pass 1: total = sum_i w_i c_i with c_i = g_ws + g_depth z_i + sum_k g_img_k rgb_ik
    float carry = 1.0f, total = 0.0f;
    for (uint32_t base0 = 0; base0 < T; base0 += 64 * kChunks) {
        float zi[kChunks], zn[kChunks], sg[kChunks], ci[kChunks];
        load_chunks(base0, zi, zn, sg, ci);
#pragma unroll
        for (int u = 0; u < kChunks; u++) {
            const uint32_t i = base0 + u * 64 + lane;
            if (base0 + u * 64 >= T) break;  // wave-uniform
            float delta, alpha = 0.0f, om = 1.0f, e;
            if (i < T) alpha_of(zi[u], zn[u], i + 1 < T, sg[u], sd, density_scale, delta, alpha, om, e);
            const float incl = wave_scan_mul(om, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            total += i < T ? alpha * (carry * excl) * ci[u] : 0.0f;
            carry *= __shfl(incl, 63, 64);
        }
    }
    total = wave_sum(total);

    // pass 2: prefix of w c, gradients
    carry = 1.0f;
    float pref_carry = 0.0f;
    for (uint32_t base0 = 0; base0 < T; base0 += 64 * kChunks) {
        float zi[kChunks], zn[kChunks], sg[kChunks], ci[kChunks];
        load_chunks(base0, zi, zn, sg, ci);
#pragma unroll
        for (int u = 0; u < kChunks; u++) {
            const uint32_t i = base0 + u * 64 + lane;
            if (base0 + u * 64 >= T) break;  // wave-uniform
            float delta = 0.0f, alpha = 0.0f, om = 1.0f, e = 1.0f;
            if (i < T) alpha_of(zi[u], zn[u], i + 1 < T, sg[u], sd, density_scale, delta, alpha, om, e);
            const float cu = i < T ? ci[u] : 0.0f;
            const float incl = wave_scan_mul(om, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            const float Ti = carry * excl;
            const float w = alpha * Ti;
            const float wc = w * cu;
            const float pref = pref_carry + wave_scan_add(wc, lane);  // inclusive prefix of w c
            if (i < T) {
                const float suffix = total - pref;  // sum_{j>i} w_j c_j
                const float dalpha = Ti * cu - suffix / om;
                gs[i] = dalpha * (delta * density_scale * e);  // d alpha / d sigma = delta * s * exp(-delta s sigma)
                if (gc) {
#pragma unroll
                    for (int k = 0; k < K; k++) gc[(size_t)i * K + k] = w * gim[k];
                }
            }
            carry *= __shfl(incl, 63, 64);
            pref_carry = __shfl(pref, 63, 64);
        }
    }
}

// ------------------------------------------------------------------------------------------------ resample + merge
// One 64-thread workgroup (one wave) per ray.  Dynamic LDS layout (floats):
//   zs[T] | cdf[T] | key[P] | val[P] | cnt[n]   (P = n_new rounded up to a power of two, >= 64): 6.9 KB at 768 + 64, so
// that the 4096 one-wave workgroups of a batch are resident together (16 per CU) — the kernel is pure latency per ray
__global__ void __launch_bounds__(64)
k_lidar_resample(const float *__restrict__ z, const float *__restrict__ sigma, const float *__restrict__ sample_dist,
                 const float *__restrict__ u, uint32_t N, uint32_t T, uint32_t n_new, uint32_t P,
                 float density_scale, uint32_t sorted_new, float *__restrict__ new_z, float *__restrict__ z_out,
                 int32_t *__restrict__ perm, uint32_t sigma_stride, const float *__restrict__ rays_o,
                 const float *__restrict__ rays_d, const float *__restrict__ aabb, float bound, float *__restrict__ x01) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *zs = reinterpret_cast<float *>(smem_raw);
    float *cdf = zs + T;
    float *key = cdf + T;
    int *val = reinterpret_cast<int *>(key + P);
    int *cnt = val + P;

    const int lane = threadIdx.x;
    const uint32_t ray = blockIdx.x;
    const float *zr = z + (size_t)ray * T, *sr = sigma + (size_t)ray * sigma_stride;
    const float sd = sample_dist[ray];
    const uint32_t nb = T - 1;  // number of bins (z_mid entries) = cdf entries
    const uint32_t nw = T - 2;  // number of pdf weights: weights[1:-1]

    // 0) the ray's z and sigma rows into LDS with all loads in flight at once: the scan loop below would otherwise pay
    //    one HBM round trip per 64-sample chunk.  sigma shares its words with cdf[]: step 1 reads sigma[i] and then
    //    writes cdf[i] from the same lane, nothing else touches either before the barrier
    float *sg = cdf;
#pragma unroll 4
    for (uint32_t i = lane; i < T; i += 64) {
        zs[i] = zr[i];
        sg[i] = sr[i];
    }
    __syncthreads();
    // 1) stage-1 weights (renderer.py:180-194); keep w in cdf[] scratch (shifted: cdf[i] = w_i for i in 1..T-2)
    float carry = 1.0f, wsum = 0.0f;
    for (uint32_t base = 0; base < T; base += 64) {
        const uint32_t i = base + lane;
        float zi = 0, delta, alpha = 0.0f, om = 1.0f, e;
        if (i < T) sample_alpha(zs, sg, i, T, sd, density_scale, zi, delta, alpha, om, e);
        const float incl = wave_scan_mul(om, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float w = alpha * (carry * excl);
        if (i >= 1 && i + 1 < T) {
            const float wp = w + 1e-5f;  // renderer.py:17
            cdf[i] = wp;
            wsum += wp;
        }
        carry *= __shfl(incl, 63, 64);
    }
    wsum = wave_sum(wsum);
    __syncthreads();
    // 2) cdf = [0, cumsum(pdf)]  (renderer.py:18-20): cdf[k] for k = 1..nw is sum_{m<k} pdf_m, pdf_m = w_{m+1}/sum
    float run = 0.0f;
    for (uint32_t base = 0; base < nw; base += 64) {
        const uint32_t m = base + lane;
        const float pdf = (m < nw) ? cdf[m + 1] / wsum : 0.0f;
        const float inc = run + wave_scan_add(pdf, lane);
        __syncthreads();  // all lanes have read cdf[m+1] of this round before it is overwritten
        if (m < nw) cdf[m + 1] = inc;
        run = __shfl(inc, 63, 64);
    }
    if (lane == 0) cdf[0] = 0.0f;
    __syncthreads();
    // 3) inverse-cdf lookup per new sample (renderer.py:31-44)
    for (uint32_t base = 0; base < P; base += 64) {
        const uint32_t j = base + lane;
        float s = __builtin_inff();  // padding keys sort to the end
        if (j < n_new) {
            const float uj = u[(size_t)ray * n_new + j];
            // searchsorted(right=True): number of cdf entries <= u
            uint32_t lo = 0, hi = nb;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (cdf[mid] <= uj) lo = mid + 1; else hi = mid;
            }
            const uint32_t below = lo > 0 ? lo - 1 : 0;
            const uint32_t above = lo < nb - 1 ? lo : nb - 1;
            const float cb = cdf[below], ca = cdf[above];
            // z_mid = z[:-1] + 0.5 * deltas[:-1]  (renderer.py:196)
            const float bb = zs[below] + 0.5f * (zs[below + 1] - zs[below]);
            const float ba = zs[above] + 0.5f * (zs[above + 1] - zs[above]);
            float denom = ca - cb;
            if (denom < 1e-5f) denom = 1.0f;
            const float t = (uj - cb) / denom;
            s = bb + t * (ba - bb);
            if (!sorted_new) new_z[(size_t)ray * n_new + j] = s;
        }
        key[j] = s;
        val[j] = (int)j;
    }
    __syncthreads();
    // 4) bitonic sort of the P (key, val) pairs in LDS
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t jj = k >> 1; jj > 0; jj >>= 1) {
            for (uint32_t tix = lane; tix < P; tix += 64) {
                const uint32_t partner = tix ^ jj;
                if (partner > tix) {
                    const bool up = (tix & k) == 0;
                    const float a = key[tix], b = key[partner];
                    const int va = val[tix], vb = val[partner];
                    // total order (key, original index) so equal keys keep their original order
                    const bool gt = (a > b) || (a == b && va > vb);
                    if (gt == up) {
                        key[tix] = b; key[partner] = a;
                        val[tix] = vb; val[partner] = va;
                    }
                }
            }
            __syncthreads();
        }
    }
    // 5) merge ranks.  New element of sorted rank r sits after cnt_r = #{old z <= key_r} old elements
    //    (ties: old first, the stable order of a sort on concat([old, new])).
    for (uint32_t r = lane; r < n_new; r += 64) {
        const float s = key[r];
        uint32_t lo = 0, hi = T;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (zs[mid] <= s) lo = mid + 1; else hi = mid;
        }
        cnt[r] = (int)lo;
    }
    __syncthreads();
    // old element i lands at i + #{r : cnt_r <= i}; new element of rank r at cnt_r + r
    for (uint32_t i = lane; i < T; i += 64) {
        uint32_t lo = 0, hi = n_new;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint32_t)cnt[mid] <= i) lo = mid + 1; else hi = mid;
        }
        // (positions grow with i: consecutive lanes write consecutive or nearly consecutive words of the ray's row)
        z_out[(size_t)ray * (T + n_new) + i + lo] = zs[i];
        perm[(size_t)ray * (T + n_new) + i + lo] = (int)i;
    }
    for (uint32_t r = lane; r < n_new; r += 64) {
        const uint32_t pos = (uint32_t)cnt[r] + r;
        z_out[(size_t)ray * (T + n_new) + pos] = key[r];
        // sorted_new: the new samples are handed out in ascending order (slot T + r), so that the density pass sees
        // consecutive lanes = neighbouring positions along the ray, like the coarse samples
        perm[(size_t)ray * (T + n_new) + pos] = (int)(T + (sorted_new ? r : (uint32_t)val[r]));
        if (sorted_new) new_z[(size_t)ray * n_new + r] = key[r];
        if (x01) {
            // grid coordinates of the new samples (what lnh_lidar_sample_points computes for slots T .. T+n_new-1 of the
            // ray's rows in the combined [N, T+n_new] buffer; sorted_new order), same arithmetic
            float *o = x01 + ((size_t)ray * (T + n_new) + T + r) * 3;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                float p = rays_o[ray * 3 + d] + rays_d[ray * 3 + d] * key[r];
                p = fminf(fmaxf(p, aabb[d]), aabb[3 + d]);
                o[d] = (p + bound) / (2 * bound);
            }
        }
    }
}

}  // namespace

extern "C" {

int lnh_lidar_weights(const float *z, const float *sigma, const float *sample_dist, uint32_t N, uint32_t T,
                      float density_scale, float *weights, lnh_stream_t stream) {
    LNH_REQUIRE(z && sigma && sample_dist && weights, LNH_ERR_INVALID_ARG, "lidar_weights: null pointer");
    LNH_REQUIRE(T >= 1, LNH_ERR_INVALID_ARG, "lidar_weights: T must be >= 1");
    if (N == 0) return LNH_OK;
    LNH_LAUNCH(k_lidar_weights, dim3(div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, z, sigma, sample_dist, N,
                       T, density_scale, weights);
    return lnh_check_launch("lnh_lidar_weights");
}

int lnh_lidar_composite_forward(const float *z, const float *sigma, const float *rgb, const float *sample_dist,
                                uint32_t N, uint32_t T, uint32_t K, float density_scale, float *weights,
                                float *weights_sum, float *depth, float *image, lnh_stream_t stream) {
    LNH_REQUIRE(z && sigma && rgb && sample_dist && weights_sum && depth && image, LNH_ERR_INVALID_ARG,
                "lidar_composite_forward: null pointer");
    LNH_REQUIRE(T >= 1, LNH_ERR_INVALID_ARG, "lidar_composite_forward: T must be >= 1");
    if (N == 0) return LNH_OK;
    dim3 grid(div_up(N, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
        case 1: LNH_LAUNCH(k_lidar_composite_fwd<1>, grid, block, 0, s, z, sigma, rgb, sample_dist, N, T, density_scale, weights, weights_sum, depth, image); break;
        case 2: LNH_LAUNCH(k_lidar_composite_fwd<2>, grid, block, 0, s, z, sigma, rgb, sample_dist, N, T, density_scale, weights, weights_sum, depth, image); break;
        case 3: LNH_LAUNCH(k_lidar_composite_fwd<3>, grid, block, 0, s, z, sigma, rgb, sample_dist, N, T, density_scale, weights, weights_sum, depth, image); break;
        case 4: LNH_LAUNCH(k_lidar_composite_fwd<4>, grid, block, 0, s, z, sigma, rgb, sample_dist, N, T, density_scale, weights, weights_sum, depth, image); break;
        default: lnh_set_error("lidar_composite: K must be 1..4 (got %u)", K); return LNH_ERR_UNSUPPORTED;
    }
    return lnh_check_launch("lnh_lidar_composite_forward");
}

int lnh_lidar_composite_backward(const float *grad_weights_sum, const float *grad_depth, const float *grad_image,
                                 const float *z, const float *sigma, const float *rgb, const float *sample_dist,
                                 uint32_t N, uint32_t T, uint32_t K, float density_scale, float *grad_sigma,
                                 float *grad_rgb, lnh_stream_t stream) {
    LNH_REQUIRE(z && sigma && rgb && sample_dist && grad_sigma, LNH_ERR_INVALID_ARG,
                "lidar_composite_backward: null pointer");
    LNH_REQUIRE(T >= 1, LNH_ERR_INVALID_ARG, "lidar_composite_backward: T must be >= 1");
    if (N == 0) return LNH_OK;
    dim3 grid(div_up(N, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (K) {
        case 1: LNH_LAUNCH(k_lidar_composite_bwd<1>, grid, block, 0, s, grad_weights_sum, grad_depth, grad_image, z, sigma, rgb, sample_dist, N, T, density_scale, grad_sigma, grad_rgb); break;
        case 2: LNH_LAUNCH(k_lidar_composite_bwd<2>, grid, block, 0, s, grad_weights_sum, grad_depth, grad_image, z, sigma, rgb, sample_dist, N, T, density_scale, grad_sigma, grad_rgb); break;
        case 3: LNH_LAUNCH(k_lidar_composite_bwd<3>, grid, block, 0, s, grad_weights_sum, grad_depth, grad_image, z, sigma, rgb, sample_dist, N, T, density_scale, grad_sigma, grad_rgb); break;
        case 4: LNH_LAUNCH(k_lidar_composite_bwd<4>, grid, block, 0, s, grad_weights_sum, grad_depth, grad_image, z, sigma, rgb, sample_dist, N, T, density_scale, grad_sigma, grad_rgb); break;
        default: lnh_set_error("lidar_composite: K must be 1..4 (got %u)", K); return LNH_ERR_UNSUPPORTED;
    }
    return lnh_check_launch("lnh_lidar_composite_backward");
}

static int resample_launch(const float *z, const float *sigma, uint32_t sigma_stride, const float *sample_dist,
                           const float *u, uint32_t N, uint32_t T, uint32_t n_new, float density_scale,
                           uint32_t sorted_new, float *new_z, float *z_out, int32_t *perm, const float *rays_o,
                           const float *rays_d, const float *aabb, float bound, float *x01, lnh_stream_t stream) {
    LNH_REQUIRE(z && sigma && sample_dist && u && new_z && z_out && perm, LNH_ERR_INVALID_ARG,
                "lidar_resample: null pointer");
    LNH_REQUIRE(sigma_stride >= T, LNH_ERR_INVALID_ARG, "lidar_resample: sigma_stride %u < T %u", sigma_stride, T);
    LNH_REQUIRE(T >= 3, LNH_ERR_INVALID_ARG, "lidar_resample: needs T >= 3 coarse samples (got %u)", T);
    LNH_REQUIRE(n_new >= 1 && n_new <= 1024, LNH_ERR_UNSUPPORTED, "lidar_resample: n_new must be in 1..1024 (got %u)", n_new);
    uint32_t P = 64;
    while (P < n_new) P <<= 1;
    const size_t lds = sizeof(float) * (2 * (size_t)T + 2 * (size_t)P + n_new);
    LNH_REQUIRE(lds <= 160 * 1024, LNH_ERR_UNSUPPORTED, "lidar_resample: T=%u n_new=%u needs %zu B of LDS (> 160 KiB)", T,
                n_new, lds);
    if (N == 0) return LNH_OK;
    hipStream_t s = (hipStream_t)stream;
    if (lds > 64 * 1024)
        hipFuncSetAttribute((const void *)k_lidar_resample, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    LNH_LAUNCH(k_lidar_resample, dim3(N), dim3(64), lds, s, z, sigma, sample_dist, u, N, T, n_new, P,
                       density_scale, sorted_new, new_z, z_out, perm, sigma_stride, rays_o, rays_d, aabb, bound, x01);
    return lnh_check_launch("lnh_lidar_resample");
}

int lnh_lidar_resample_strided(const float *z, const float *sigma, uint32_t sigma_stride, const float *sample_dist,
                               const float *u, uint32_t N, uint32_t T, uint32_t n_new, float density_scale,
                               uint32_t sorted_new, float *new_z, float *z_out, int32_t *perm, lnh_stream_t stream) {
    return resample_launch(z, sigma, sigma_stride, sample_dist, u, N, T, n_new, density_scale, sorted_new, new_z, z_out, perm,
                           nullptr, nullptr, nullptr, 1.0f, nullptr, stream);
}

int lnh_lidar_resample_points(const float *z, const float *sigma, uint32_t sigma_stride, const float *sample_dist,
                              const float *u, uint32_t N, uint32_t T, uint32_t n_new, float density_scale, float *new_z,
                              float *z_out, int32_t *perm, const float *rays_o, const float *rays_d, const float *aabb,
                              float bound, float *x01, lnh_stream_t stream) {
    LNH_REQUIRE(rays_o && rays_d && aabb && x01, LNH_ERR_INVALID_ARG, "lidar_resample_points: null pointer");
    LNH_REQUIRE(bound > 0.0f, LNH_ERR_INVALID_ARG, "lidar_resample_points: bound must be positive");
    return resample_launch(z, sigma, sigma_stride, sample_dist, u, N, T, n_new, density_scale, 1, new_z, z_out, perm, rays_o,
                           rays_d, aabb, bound, x01, stream);
}

int lnh_lidar_resample(const float *z, const float *sigma, const float *sample_dist, const float *u, uint32_t N,
                       uint32_t T, uint32_t n_new, float density_scale, uint32_t sorted_new, float *new_z, float *z_out,
                       int32_t *perm, lnh_stream_t stream) {
    return resample_launch(z, sigma, T, sample_dist, u, N, T, n_new, density_scale, sorted_new, new_z, z_out, perm, nullptr,
                           nullptr, nullptr, 1.0f, nullptr, stream);
}

}  // extern "C"
