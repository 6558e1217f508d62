// lidar_glue.hip — the small element-wise stages around the LiDAR field that the reference leaves to ~100 PyTorch
// launches per step (renderer.py:129-161 sampling set-up, network.py:215-221 direction term, nerf/utils.py:712-746
// loss).  Each is one launch here; none of them is bandwidth- or compute-relevant, the point is the launch count.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ coarse samples
// z[n,i] = near + (far - near) * lin_i (+ (u[n,i] - 0.5) * (far - near) / T), lin = torch.linspace(0, 1, T)
// (renderer.py:147-161).  torch.linspace is evaluated symmetrically (start + step*i in the lower half, end -
// step*(T-1-i) in the upper half); the same form is used here so that both paths draw identical samples.
__global__ void __launch_bounds__(256)
k_coarse_samples(const float *__restrict__ u, uint32_t N, uint32_t T, float near, float far, float *__restrict__ z) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * T) return;
    const uint32_t i = idx % T;
    const float step = T > 1 ? 1.0f / (float)(T - 1) : 0.0f;
    const float lin = i < T / 2 ? step * (float)i : 1.0f - step * (float)(T - 1 - i);
    float v = near + (far - near) * lin;
    if (u) v = v + (u[idx] - 0.5f) * ((far - near) / (float)T);
    z[idx] = v;
}

// ------------------------------------------------------------------------------------------------ direction term
// enc16[n,k] = direction feature rounded to the MLP element type E (fp16 / bf16); cdir[n,o] = sum_k enc16[n,k] *
// E(W0[o,k])  (fp32 accumulate).  One workgroup = 4 rays x 64 outputs; the feature row of a ray is broadcast through LDS.
// FREQ: `enc` holds the raw directions [N,3] and the K = 3 + 6 * deg frequency features (encoders.hip k_freq_forward:
// x | sin(2^f x), sin(2^f x + pi/2) per band, the same operations in the same order) are formed here instead of being
// read — the direction encoder and the direction term in one launch.
template <typename E, bool FREQ>
__global__ void __launch_bounds__(256)
k_dir_term(const float *__restrict__ enc, const float *__restrict__ W0, uint32_t ldw, uint32_t N, uint32_t K,
           float *__restrict__ enc16, float *__restrict__ cdir) {
    __shared__ float row[4][128];
    const uint32_t r = threadIdx.x >> 6, o = threadIdx.x & 63;
    const uint32_t n = blockIdx.x * 4 + r;
    const bool valid = n < N;
    for (uint32_t k = o; k < K; k += 64) {
        float feat;
        if constexpr (FREQ) {
            const uint32_t q = k < 3 ? 0u : k - 3u, d = k < 3 ? k : q % 3u, f = q / 6u, c = (q / 3u) & 1u;
            const float x = enc[(size_t)(valid ? n : 0) * 3 + d];
            const float a = scalbnf(x, (int)f);
            feat = k < 3 ? x : sinf(c ? a + 1.5707963267948966f : a);
        } else {
            feat = enc[(size_t)(valid ? n : 0) * K + k];
        }
        const float v = (float)(E)feat;
        row[r][k] = v;
        if (valid) enc16[(size_t)n * K + k] = v;
    }
    __syncthreads();
    float acc = 0.0f;
    const float *w = W0 + (size_t)o * ldw;
    for (uint32_t k = 0; k < K; k++) acc = fmaf(row[r][k], (float)(E)w[k], acc);
    if (valid) cdir[(size_t)n * 64 + o] = acc;
}

// d cdir / d W0_dir:  gW[o, k] += sum_n S[n, o] * enc16[n, k]  (S = per-ray sum of dH0 from the colour backward).
// A library GEMM with M = 64, N = K <= 128 and a 4096-long reduction runs as one tile for ~43 us.  Two tiny passes
// instead: every workgroup reduces a 32-ray chunk (staged in LDS) to a [64, 128] partial in scratch, then one thread
// per output sums the partials.  (Atomics instead of the second pass were measured slower: 64*K addresses are so
// few cache lines that device atomics on them serialise.)
constexpr int kDirChunk = 32;
__global__ void __launch_bounds__(256)
k_dir_term_backward_partial(const float *__restrict__ S, const float *__restrict__ enc16, uint32_t N, uint32_t K,
                            float *__restrict__ partial) {
    constexpr int KPT = 32, CH = kDirChunk;  // k values per thread (k = kg + 4*j), rays per workgroup
    __shared__ float sS[CH][64];
    __shared__ __attribute__((aligned(16))) float sE[CH][128];
    const uint32_t o = threadIdx.x & 63, kg = threadIdx.x >> 6;
    const uint32_t n0 = blockIdx.x * CH, cnt = min((uint32_t)CH, N - n0);
    for (uint32_t i = threadIdx.x; i < CH * 64; i += 256) {  // coalesced staging, zero-filled beyond the valid part
        const uint32_t r = i >> 6;
        sS[r][i & 63] = r < cnt ? S[(size_t)(n0 + r) * 64 + (i & 63)] : 0.0f;
    }
    // enc is staged as [ray][kg][j] (k = kg + 4 j): the 32 values a thread multiplies by one S element are contiguous, so
    // the inner loop reads them with 8 wave-uniform 16-byte LDS loads instead of 32 4-byte ones (the kernel was bound by
    // its LDS instruction count)
    for (uint32_t i = threadIdx.x; i < CH * 128; i += 256) {
        const uint32_t r = i >> 7, k = i & 127;
        sE[r][(k & 3) * KPT + (k >> 2)] = (r < cnt && k < K) ? enc16[(size_t)(n0 + r) * K + k] : 0.0f;
    }
    __syncthreads();
    float acc[KPT];
#pragma unroll
    for (int j = 0; j < KPT; j++) acc[j] = 0.0f;
    for (uint32_t r = 0; r < CH; r++) {
        const float s = sS[r][o];
        const float4 *e4 = reinterpret_cast<const float4 *>(&sE[r][kg * KPT]);
#pragma unroll
        for (int q = 0; q < KPT / 4; q++) {
            const float4 e = e4[q];
            acc[4 * q + 0] = fmaf(s, e.x, acc[4 * q + 0]);
            acc[4 * q + 1] = fmaf(s, e.y, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(s, e.z, acc[4 * q + 2]);
            acc[4 * q + 3] = fmaf(s, e.w, acc[4 * q + 3]);
        }
    }
    float *out = partial + (size_t)blockIdx.x * 64 * 128;
#pragma unroll
    for (int j = 0; j < KPT; j++) out[(kg + 4 * j) * 64 + o] = acc[j];  // [k][o]: consecutive lanes, consecutive floats
}
__global__ void __launch_bounds__(256)
k_dir_term_backward_sum(const float *__restrict__ partial, uint32_t chunks, uint32_t K, float *__restrict__ gW,
                        uint32_t ldw) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // i = k*64 + o
    if (i >= K * 64) return;
    float a = 0.0f;  // blockIdx.y strides over the chunks: 8-way split keeps the dependent-load chain short
    for (uint32_t c = blockIdx.y; c < chunks; c += gridDim.y) a += partial[(size_t)c * 64 * 128 + i];
    const uint32_t k = i >> 6, o = i & 63;
    unsafeAtomicAdd(gW + (size_t)o * ldw + k, a);
}

// ------------------------------------------------------------------------------------------------ weight packing
// fp32 master weights (possibly strided views) -> the flat fp16 vectors the fused kernels read:
//   wsig = [ws0 (64x32) | ws1 (16x64)],  wcol = [W0g (64x16: col 0 zero, cols 1..15 = wc0[:, kd:kd+15]) | wc1 | wc2 padded to 16 rows]
template <typename E>
struct PackArgs {
    const float *ws0, *ws1, *wc0, *wc1, *wc2;
    uint32_t ld_s0, ld_s1, ld_c0, ld_c1, ld_c2, kd;
    E *wsig, *wcol;
};
template <typename E>
__global__ void __launch_bounds__(256)
k_pack_weights(PackArgs<E> a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr uint32_t nS0 = 64 * 32, nS1 = 16 * 64, nC0 = 64 * 16, nC1 = 64 * 64, nC2 = 16 * 64;
    if (i < nS0) {
        a.wsig[i] = (E)a.ws0[(i / 32) * a.ld_s0 + i % 32];
    } else if (i < nS0 + nS1) {
        const uint32_t j = i - nS0;
        a.wsig[i] = (E)a.ws1[(j / 64) * a.ld_s1 + j % 64];
    }
    if (i < nC0) {
        const uint32_t o = i / 16, c = i % 16;
        a.wcol[i] = c == 0 ? (E)0.0f : (E)a.wc0[o * a.ld_c0 + a.kd + c - 1];
    } else if (i < nC0 + nC1) {
        const uint32_t j = i - nC0;
        a.wcol[i] = (E)a.wc1[(j / 64) * a.ld_c1 + j % 64];
    } else if (i < nC0 + nC1 + nC2) {
        const uint32_t j = i - nC0 - nC1, o = j / 64;
        a.wcol[i] = o < 2 ? (E)a.wc2[o * a.ld_c2 + j % 64] : (E)0.0f;
    }
}

// ------------------------------------------------------------------------------------------------ loss
// nerf/utils.py:712-746 with the default criteria: per ray
//   l = a_d |pd - gd| + a_r (pr - gr)^2 + a_i (pi - gi)^2,  pd = depth * gr, gd = gt_depth * gr, pi = intensity * gr,
// loss = mean(l).  Emits the loss and d loss / d depth, d loss / d image in the same pass.
__global__ void __launch_bounds__(256)
k_lidar_loss(const float *__restrict__ depth, const float *__restrict__ image, const float *__restrict__ gt, uint32_t N,
             float a_d, float a_r, float a_i, float *__restrict__ loss, float *__restrict__ g_depth,
             float *__restrict__ g_image) {
    __shared__ float part[4];
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.0f;
    if (n < N) {
        const float gr = gt[n * 3], gi = gt[n * 3 + 1] * gr, gd = gt[n * 3 + 2] * gr;
        const float pr = image[n * 2], pi = image[n * 2 + 1] * gr, pd = depth[n] * gr;
        const float dd = pd - gd, dr = pr - gr, di = pi - gi;
        l = a_d * fabsf(dd) + a_r * dr * dr + a_i * di * di;
        const float inv = 1.0f / (float)N;
        const float sgn = dd > 0.0f ? 1.0f : (dd < 0.0f ? -1.0f : 0.0f);  // torch: sign(0) = 0
        g_depth[n] = a_d * sgn * gr * inv;
        g_image[n * 2] = 2.0f * a_r * dr * inv;
        g_image[n * 2 + 1] = 2.0f * a_i * di * gr * inv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(loss, (part[0] + part[1] + part[2] + part[3]) / (float)N);
}

// The same loss on the reference's PATCH epochs (nerf/utils.py:760-876, grad_loss, non-sobel): rays arrive as patches of
// px x py neighbouring pixels ([P, px, py] row-major), and the structural-gradient term
//   a_g * mean over (P, px, py-1) of | |pd_j - pd_j+1| * m_j - (gd_j - gd_j+1) * m_j |,   m_j = gr_j * (|gd_j - gd_j+1| < 0.01)
// (depths in metres: divided by `scale`; only the x term enters the loss) is added, with its gradient, in the same pass:
// thread n = ray n handles the per-ray loss and the pair (n, n + 1) of its patch row.
__global__ void __launch_bounds__(256)
k_lidar_loss_patch(const float *__restrict__ depth, const float *__restrict__ image, const float *__restrict__ gt,
                   uint32_t N, uint32_t py, float inv_scale, float a_d, float a_r, float a_i, float a_g,
                   float *__restrict__ loss, float *__restrict__ g_depth, float *__restrict__ g_image) {
    __shared__ float part[4];
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0.0f;
    if (n < N) {
        const float gr = gt[n * 3], gi = gt[n * 3 + 1] * gr, gd = gt[n * 3 + 2] * gr;
        const float pr = image[n * 2], pi = image[n * 2 + 1] * gr, pd = depth[n] * gr;
        const float dd = pd - gd, dr = pr - gr, di = pi - gi;
        const float inv = 1.0f / (float)N;
        l = (a_d * fabsf(dd) + a_r * dr * dr + a_i * di * di) * inv;
        const float sgn = dd > 0.0f ? 1.0f : (dd < 0.0f ? -1.0f : 0.0f);
        float gdep = a_d * sgn * gr * inv;
        g_image[n * 2] = 2.0f * a_r * dr * inv;
        g_image[n * 2 + 1] = 2.0f * a_i * di * gr * inv;
        // pairs (n, n + 1) and (n - 1, n) of this ray's patch row; N / py rows of py - 1 pairs each
        const uint32_t c = n % py;
        const float inv_pairs = a_g / ((float)(N / py) * (float)(py - 1));
        auto pair_term = [&](uint32_t j, float &dj, float &dj1) {  // pair (j, j + 1): value, d/d pred_j, d/d pred_j+1
            const float rj = gt[j * 3], rj1 = gt[(j + 1) * 3];
            const float pj = depth[j] * rj * inv_scale, pj1 = depth[j + 1] * rj1 * inv_scale;
            const float gj = gt[j * 3 + 2] * rj * inv_scale, gj1 = gt[(j + 1) * 3 + 2] * rj1 * inv_scale;
            const float gg = gj - gj1, dp = pj - pj1, pg = fabsf(dp);
            const float m = fabsf(gg) < 0.01f ? rj : 0.0f;
            const float e = pg * m - gg * m;
            const float se = e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f), sp = dp > 0.0f ? 1.0f : (dp < 0.0f ? -1.0f : 0.0f);
            dj = se * m * sp;
            dj1 = -dj;
            return fabsf(e);
        };
        float acc = 0.0f;
        if (c + 1 < py) {  // this thread owns the pair (n, n + 1): its value and d/d pred_n
            float dj, dj1;
            l += pair_term(n, dj, dj1) * inv_pairs;
            acc += dj;
        }
        if (c > 0) {       // d/d pred_n of the pair (n - 1, n)
            float dj, dj1;
            (void)pair_term(n - 1, dj, dj1);
            acc += dj1;
        }
        g_depth[n] = gdep + acc * inv_pairs * gr * inv_scale;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(loss, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace

template <typename E>
static int dir_term(const float *dir_features, const float *w0, uint32_t ldw, uint32_t N, uint32_t K, float *features16,
                    float *cdir, lnh_stream_t stream) {
    LNH_REQUIRE(dir_features && w0 && features16 && cdir, LNH_ERR_INVALID_ARG, "lidar_dir_term: null pointer");
    LNH_REQUIRE(K >= 1 && K <= 128 && ldw >= K, LNH_ERR_INVALID_ARG, "lidar_dir_term: need 1 <= K <= 128 and ldw >= K");
    if (N == 0) return LNH_OK;
    LNH_LAUNCH((k_dir_term<E, false>), dim3(div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, dir_features, w0, ldw, N, K,
               features16, cdir);
    return lnh_check_launch("lnh_lidar_dir_term");
}
template <typename E>
static int dir_term_freq(const float *dirs, uint32_t degree, const float *w0, uint32_t ldw, uint32_t N, float *features16,
                         float *cdir, lnh_stream_t stream) {
    const uint32_t K = 3 + 6 * degree;
    LNH_REQUIRE(dirs && w0 && features16 && cdir, LNH_ERR_INVALID_ARG, "lidar_dir_term_freq: null pointer");
    LNH_REQUIRE(K <= 128 && ldw >= K, LNH_ERR_INVALID_ARG, "lidar_dir_term_freq: need 3 + 6*degree <= 128 and ldw >= that");
    if (N == 0) return LNH_OK;
    LNH_LAUNCH((k_dir_term<E, true>), dim3(div_up(N, 4)), dim3(256), 0, (hipStream_t)stream, dirs, w0, ldw, N, K,
               features16, cdir);
    return lnh_check_launch("lnh_lidar_dir_term_freq");
}
template <typename E>
static int pack_weights(const float *ws0, uint32_t ld_s0, const float *ws1, uint32_t ld_s1, const float *wc0,
                        uint32_t ld_c0, uint32_t n_dir, const float *wc1, uint32_t ld_c1, const float *wc2, uint32_t ld_c2,
                        void *wsig16, void *wcol16, lnh_stream_t stream) {
    LNH_REQUIRE(ws0 && ws1 && wc0 && wc1 && wc2 && wsig16 && wcol16, LNH_ERR_INVALID_ARG,
                "lidar_pack_weights: null pointer");
    LNH_REQUIRE(ld_s0 >= 32 && ld_s1 >= 64 && ld_c0 >= n_dir + 15 && ld_c1 >= 64 && ld_c2 >= 64, LNH_ERR_INVALID_ARG,
                "lidar_pack_weights: leading dimension smaller than the row");
    PackArgs<E> a{ws0, ws1, wc0, wc1, wc2, ld_s0, ld_s1, ld_c0, ld_c1, ld_c2, n_dir, (E *)wsig16, (E *)wcol16};
    LNH_LAUNCH(k_pack_weights<E>, dim3(div_up(64 * 16 + 64 * 64 + 16 * 64, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return lnh_check_launch("lnh_lidar_pack_weights");
}
extern "C" {

int lnh_lidar_coarse_samples(const float *u, uint32_t N, uint32_t T, float near, float far, float *z,
                             lnh_stream_t stream) {
    LNH_REQUIRE(z, LNH_ERR_INVALID_ARG, "lidar_coarse_samples: null pointer");
    if ((uint64_t)N * T == 0) return LNH_OK;
    LNH_REQUIRE((uint64_t)N * T < 0xffffffffull, LNH_ERR_UNSUPPORTED, "lidar_coarse_samples: N*T must fit 32 bits");
    LNH_LAUNCH(k_coarse_samples, dim3(div_up((uint64_t)N * T, 256)), dim3(256), 0, (hipStream_t)stream, u, N, T, near,
               far, z);
    return lnh_check_launch("lnh_lidar_coarse_samples");
}

int lnh_lidar_dir_term(const float *dir_features, const float *w0, uint32_t ldw, uint32_t N, uint32_t K,
                       float *features16, float *cdir, lnh_stream_t stream) {
    return dir_term<half_t>(dir_features, w0, ldw, N, K, features16, cdir, stream);
}
int lnh_lidar_dir_term_bf16(const float *dir_features, const float *w0, uint32_t ldw, uint32_t N, uint32_t K,
                            float *features16, float *cdir, lnh_stream_t stream) {
    return dir_term<__bf16>(dir_features, w0, ldw, N, K, features16, cdir, stream);
}

int lnh_lidar_dir_term_freq(const float *dirs, uint32_t degree, const float *w0, uint32_t ldw, uint32_t N,
                            float *features16, float *cdir, lnh_stream_t stream) {
    return dir_term_freq<half_t>(dirs, degree, w0, ldw, N, features16, cdir, stream);
}
int lnh_lidar_dir_term_freq_bf16(const float *dirs, uint32_t degree, const float *w0, uint32_t ldw, uint32_t N,
                                 float *features16, float *cdir, lnh_stream_t stream) {
    return dir_term_freq<__bf16>(dirs, degree, w0, ldw, N, features16, cdir, stream);
}

int lnh_lidar_dir_term_backward(const float *ray_sum, const float *features16, uint32_t N, uint32_t K, float *scratch,
                                float *grad_w0, uint32_t ldw, lnh_stream_t stream) {
    LNH_REQUIRE(ray_sum && features16 && grad_w0 && scratch, LNH_ERR_INVALID_ARG, "lidar_dir_term_backward: null pointer");
    LNH_REQUIRE(K >= 1 && K <= 128 && ldw >= K, LNH_ERR_INVALID_ARG,
                "lidar_dir_term_backward: need 1 <= K <= 128 and ldw >= K");
    if (N == 0) return LNH_OK;
    const uint32_t chunks = div_up(N, kDirChunk);
    LNH_LAUNCH(k_dir_term_backward_partial, dim3(chunks), dim3(256), 0, (hipStream_t)stream, ray_sum, features16, N, K,
               scratch);
    int rc = lnh_check_launch("lnh_lidar_dir_term_backward(partial)");
    if (rc) return rc;
    LNH_LAUNCH(k_dir_term_backward_sum, dim3(div_up(K * 64, 256), chunks < 8 ? chunks : 8), dim3(256), 0, (hipStream_t)stream, scratch, chunks, K,
               grad_w0, ldw);
    return lnh_check_launch("lnh_lidar_dir_term_backward(sum)");
}

int lnh_lidar_pack_weights(const float *ws0, uint32_t ld_s0, const float *ws1, uint32_t ld_s1, const float *wc0,
                           uint32_t ld_c0, uint32_t n_dir, const float *wc1, uint32_t ld_c1, const float *wc2,
                           uint32_t ld_c2, void *wsig16, void *wcol16, lnh_stream_t stream) {
    return pack_weights<half_t>(ws0, ld_s0, ws1, ld_s1, wc0, ld_c0, n_dir, wc1, ld_c1, wc2, ld_c2, wsig16, wcol16, stream);
}
int lnh_lidar_pack_weights_bf16(const float *ws0, uint32_t ld_s0, const float *ws1, uint32_t ld_s1, const float *wc0,
                                uint32_t ld_c0, uint32_t n_dir, const float *wc1, uint32_t ld_c1, const float *wc2,
                                uint32_t ld_c2, void *wsig16, void *wcol16, lnh_stream_t stream) {
    return pack_weights<__bf16>(ws0, ld_s0, ws1, ld_s1, wc0, ld_c0, n_dir, wc1, ld_c1, wc2, ld_c2, wsig16, wcol16, stream);
}

int lnh_lidar_loss(const float *depth, const float *image, const float *gt, uint32_t N, float alpha_d, float alpha_r,
                   float alpha_i, float *loss, float *grad_depth, float *grad_image, lnh_stream_t stream) {
    LNH_REQUIRE(depth && image && gt && loss && grad_depth && grad_image, LNH_ERR_INVALID_ARG,
                "lidar_loss: null pointer");
    if (int zrc = lnh_zero_async(loss, sizeof(float), (hipStream_t)stream, "lidar_loss (accumulator clear)")) return zrc;
    if (N == 0) return LNH_OK;
    LNH_LAUNCH(k_lidar_loss, dim3(div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, depth, image, gt, N, alpha_d,
               alpha_r, alpha_i, loss, grad_depth, grad_image);
    return lnh_check_launch("lnh_lidar_loss");
}

int lnh_lidar_loss_patch(const float *depth, const float *image, const float *gt, uint32_t N, uint32_t px, uint32_t py,
                         float scale, float alpha_d, float alpha_r, float alpha_i, float alpha_grad, float *loss,
                         float *grad_depth, float *grad_image, lnh_stream_t stream) {
    LNH_REQUIRE(depth && image && gt && loss && grad_depth && grad_image, LNH_ERR_INVALID_ARG,
                "lidar_loss_patch: null pointer");
    LNH_REQUIRE(px >= 1 && py >= 2 && scale > 0.0f && N % (px * py) == 0, LNH_ERR_INVALID_ARG,
                "lidar_loss_patch: need py >= 2, scale > 0 and N a multiple of px * py");
    if (int zrc = lnh_zero_async(loss, sizeof(float), (hipStream_t)stream, "lidar_loss_patch (accumulator clear)")) return zrc;
    if (N == 0) return LNH_OK;
    LNH_LAUNCH(k_lidar_loss_patch, dim3(div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, depth, image, gt, N, py,
               1.0f / scale, alpha_d, alpha_r, alpha_i, alpha_grad, loss, grad_depth, grad_image);
    return lnh_check_launch("lnh_lidar_loss_patch");
}

}  // extern "C"
