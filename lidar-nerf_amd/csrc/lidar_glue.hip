// lidar_glue.hip — the small element-wise stages around the LiDAR field that the reference leaves to ~100 PyTorch
// launches per step (renderer.py:129-161 sampling set-up, network.py:215-221 direction term, nerf/utils.py:712-746
// loss).  Each is one launch here; none of them is bandwidth- or compute-relevant, the point is the launch count.
#include "common.h"
#include "lidar_steps.h"
#include "wgrad.h"

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
__device__ __forceinline__ void dir_term_block(uint32_t block, const float *__restrict__ enc, const float *__restrict__ W0,
                                               uint32_t ldw, uint32_t N, uint32_t K, float *__restrict__ enc16,
                                               float *__restrict__ cdir) {
    __shared__ float row[4][128];
    const uint32_t r = threadIdx.x >> 6, o = threadIdx.x & 63;
    const uint32_t n = block * 4 + r;
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
template <typename E, bool FREQ>
__global__ void __launch_bounds__(256)
k_dir_term(const float *__restrict__ enc, const float *__restrict__ W0, uint32_t ldw, uint32_t N, uint32_t K,
           float *__restrict__ enc16, float *__restrict__ cdir) {
    dir_term_block<E, FREQ>(blockIdx.x, enc, W0, ldw, N, K, enc16, cdir);
}

// d cdir / d W0_dir:  gW[o, k] += sum_n S[n, o] * enc16[n, k]  (S = per-ray sum of dH0 from the colour backward).
// A library GEMM with M = 64, N = K <= 128 and a 4096-long reduction runs as one tile for ~43 us; rounds 1-4 ran two
// passes (32-ray chunks staged in LDS to [64, 128] partials in scratch, then a sum over the 128 partials: 16 + 8 us).
// One launch: a workgroup owns the 64 outputs of ONE feature over a 1/16th of the rays (grid K x 16), its four waves taking
// every fourth ray (S rows are 256-byte lines, one lane per output; the feature is a wave-uniform value), eight rays in
// flight per wave; the four partial sums meet in LDS (fixed order).  The 16 segments leave their sums in scratch and a second
// launch adds them up in segment order and adds the result to the gradient (wgrad.h; round 5: 16 fp32 device atomics per
// address, i.e. an order that changed from run to run).  Block (0, 0) also copies the geo-feature columns
// of the colour head's first matrix, which the colour backward produced in its packed [64, 16] layout (col 0 unused), to
// columns K .. K+14 of the same gradient.
constexpr uint32_t kDirSegments = 16;
__global__ void __launch_bounds__(256)
k_dir_term_backward(const float *__restrict__ S, const float *__restrict__ enc16, uint32_t N, uint32_t K,
                    const float *__restrict__ g_w0g, float *__restrict__ gW, uint32_t ldw, WgradWs ws) {
    __shared__ float part[4][64];
    const uint32_t o = threadIdx.x & 63, w = threadIdx.x >> 6, k = blockIdx.x;
    const uint32_t per = (N + kDirSegments - 1) / kDirSegments;
    const uint32_t r0 = blockIdx.y * per, r1 = min(N, r0 + per);
    constexpr int U = 8;
    float acc[U];
#pragma unroll
    for (int u = 0; u < U; u++) acc[u] = 0.0f;
    uint32_t r = r0 + w;
    for (; r + 4 * (U - 1) < r1; r += 4 * U) {  // U independent chains: the loop waits for L2 once per U rays
        float sv[U], ev[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            sv[u] = S[(size_t)(r + 4 * u) * 64 + o];
            ev[u] = enc16[(size_t)(r + 4 * u) * K + k];
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc[u] = fmaf(sv[u], ev[u], acc[u]);
    }
    for (; r < r1; r += 4) acc[0] = fmaf(S[(size_t)r * 64 + o], enc16[(size_t)r * K + k], acc[0]);
    part[w][o] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (g_w0g && blockIdx.x == 0 && blockIdx.y == 0) {
        for (uint32_t i = threadIdx.x; i < 64 * 15; i += 256) {
            const uint32_t oo = i / 15, c = i % 15;
            gW[(size_t)oo * ldw + K + c] = g_w0g[oo * 16 + 1 + c];
        }
    }
    // partial of segment y: [K][64] floats; the second launch (k_wgrad_reduce, wgrad.h) adds the 16 segments in order
    if (w == 0) ws.partials[((size_t)blockIdx.y * K + k) * 64 + o] = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
}

// element e4 of the summed partial = outputs o .. o + 3 of feature k
struct DirTermEmit {
    float *gW;
    uint32_t ldw;
    __device__ __forceinline__ void operator()(uint32_t e4, const float4 &v) const {
        const uint32_t k = e4 >> 4, o = (e4 & 15) * 4;
        float *p = gW + (size_t)o * ldw + k;
        p[0] += v.x;
        p[ldw] += v.y;
        p[2 * (size_t)ldw] += v.z;
        p[3 * (size_t)ldw] += v.w;
    }
};

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
__device__ __forceinline__ void pack_weights_block(uint32_t block, const PackArgs<E> &a) {
    const uint32_t i = block * blockDim.x + threadIdx.x;
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
constexpr uint32_t kPackBlocks = (64 * 16 + 64 * 64 + 16 * 64 + 255) / 256;
template <typename E>
__global__ void __launch_bounds__(256)
k_pack_weights(PackArgs<E> a) {
    pack_weights_block<E>(blockIdx.x, a);
}

// ------------------------------------------------------------------------------------------------ step prologue
// Everything a render step needs before its first encode, in ONE launch (round 5; three launches of 5 / 9 / 16 us before):
// the weight packing (first blocks), the per-ray direction term with the frequency encoder folded in (next N / 4 blocks),
// and the coarse pass — stratified depths and their grid coordinates (the rest).  The bodies are the stand-alone kernels'.
struct PrologueArgs {
    const float *dirs, *w0;  // direction term: raw directions [N, 3], the colour head's first matrix (row stride ldw)
    uint32_t ldw, degree;
    float *enc16, *cdir;
    const float *u, *rays_o, *aabb;  // coarse pass (rays_d = dirs)
    float bound, near, far;
    uint32_t N, T, T_tot;
    float *z, *x01;
};
template <typename E>
__global__ void __launch_bounds__(256)
k_step_prologue(PackArgs<E> pk, PrologueArgs a) {
    const uint32_t dir_blocks = (a.N + 3) / 4;
    if (blockIdx.x < kPackBlocks) {
        pack_weights_block<E>(blockIdx.x, pk);
    } else if (blockIdx.x < kPackBlocks + dir_blocks) {
        dir_term_block<E, true>(blockIdx.x - kPackBlocks, a.dirs, a.w0, a.ldw, a.N, 3 + 6 * a.degree, a.enc16, a.cdir);
    } else {
        const uint32_t idx = (blockIdx.x - kPackBlocks - dir_blocks) * blockDim.x + threadIdx.x;
        if (idx < a.N * a.T)
            coarse_sample_point(idx, a.u, a.rays_o, a.dirs, a.aabb, a.bound, a.T, a.T_tot, a.near, a.far, a.z, a.x01);
    }
}

// ------------------------------------------------------------------------------------------------ loss
// nerf/utils.py:712-746 with the default criteria: per ray
//   l = a_d |pd - gd| + a_r (pr - gr)^2 + a_i (pi - gi)^2,  pd = depth * gr, gd = gt_depth * gr, pi = intensity * gr,
// loss = mean(l).  Emits the loss and d loss / d depth, d loss / d image in the same pass.
// ONE workgroup of 1024 threads walks the batch (a training batch is 4096 .. 16384 rays: 4 .. 16 per thread, a few
// microseconds next to a step of milliseconds): thread t adds its rays t, t + 1024, ... in that order, the lanes of a wave meet
// in a shuffle tree, the 16 waves in LDS in wave order — the same bits on every run, no accumulator to clear before the
// launch (rounds 1-4: a zero-fill launch + float atomics) and NO state outside the launch (round 5: per-workgroup partials
// and an arrival counter in static device memory, which two launches overlapping in time — an evaluation loss on another
// stream, two trainers in one process — would have corrupted for each other).
constexpr uint32_t kLossThreads = 1024;
__device__ __forceinline__ void loss_finish(float l, float scale_out, float *__restrict__ loss) {
    __shared__ float part[kLossThreads / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = part[0];
#pragma unroll
        for (uint32_t w = 1; w < kLossThreads / 64; w++) t += part[w];
        *loss = t * scale_out;
    }
}

// `grad_scale` (device scalar, or null): the gradients come out multiplied by it — the loss scale of the training step, so
// that backward() has nothing left to multiply (one element-wise launch less per step).
__global__ void __launch_bounds__(kLossThreads)
k_lidar_loss(const float *__restrict__ depth, const float *__restrict__ image, const float *__restrict__ gt, uint32_t N,
             float a_d, float a_r, float a_i, const float *__restrict__ grad_scale, float *__restrict__ loss,
             float *__restrict__ g_depth, float *__restrict__ g_image) {
    const float gsc = grad_scale ? *grad_scale : 1.0f;
    float l = 0.0f;
    for (uint32_t n = threadIdx.x; n < N; n += kLossThreads) {
        const float gr = gt[n * 3], gi = gt[n * 3 + 1] * gr, gd = gt[n * 3 + 2] * gr;
        const float pr = image[n * 2], pi = image[n * 2 + 1] * gr, pd = depth[n] * gr;
        const float dd = pd - gd, dr = pr - gr, di = pi - gi;
        l += a_d * fabsf(dd) + a_r * dr * dr + a_i * di * di;
        const float inv = 1.0f / (float)N;
        const float sgn = dd > 0.0f ? 1.0f : (dd < 0.0f ? -1.0f : 0.0f);  // torch: sign(0) = 0
        g_depth[n] = a_d * sgn * gr * inv * gsc;
        g_image[n * 2] = 2.0f * a_r * dr * inv * gsc;
        g_image[n * 2 + 1] = 2.0f * a_i * di * gr * inv * gsc;
    }
    loss_finish(l, 1.0f / (float)N, loss);
}

// The same loss on the reference's PATCH epochs (nerf/utils.py:760-876, grad_loss, non-sobel): rays arrive as patches of
// px x py neighbouring pixels ([P, px, py] row-major), and the structural-gradient term
//   a_g * mean over (P, px, py-1) of | |pd_j - pd_j+1| * m_j - (gd_j - gd_j+1) * m_j |,   m_j = gr_j * (|gd_j - gd_j+1| < 0.01)
// (depths in metres: divided by `scale`; only the x term enters the loss) is added, with its gradient, in the same pass:
// thread n = ray n handles the per-ray loss and the pair (n, n + 1) of its patch row.
__global__ void __launch_bounds__(kLossThreads)
k_lidar_loss_patch(const float *__restrict__ depth, const float *__restrict__ image, const float *__restrict__ gt,
                   uint32_t N, uint32_t py, float inv_scale, float a_d, float a_r, float a_i, float a_g,
                   const float *__restrict__ grad_scale, float *__restrict__ loss, float *__restrict__ g_depth,
                   float *__restrict__ g_image) {
    const float gsc = grad_scale ? *grad_scale : 1.0f;
    float l = 0.0f;
    for (uint32_t n = threadIdx.x; n < N; n += kLossThreads) {
        const float gr = gt[n * 3], gi = gt[n * 3 + 1] * gr, gd = gt[n * 3 + 2] * gr;
        const float pr = image[n * 2], pi = image[n * 2 + 1] * gr, pd = depth[n] * gr;
        const float dd = pd - gd, dr = pr - gr, di = pi - gi;
        const float inv = 1.0f / (float)N;
        l += (a_d * fabsf(dd) + a_r * dr * dr + a_i * di * di) * inv;
        const float sgn = dd > 0.0f ? 1.0f : (dd < 0.0f ? -1.0f : 0.0f);
        float gdep = a_d * sgn * gr * inv;
        g_image[n * 2] = 2.0f * a_r * dr * inv * gsc;
        g_image[n * 2 + 1] = 2.0f * a_i * di * gr * inv * gsc;
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
        g_depth[n] = (gdep + acc * inv_pairs * gr * inv_scale) * gsc;
    }
    loss_finish(l, 1.0f, loss);
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
template <typename E>
static int step_prologue(const float *ws0, uint32_t ld_s0, const float *ws1, uint32_t ld_s1, const float *wc0, uint32_t ld_c0,
                         uint32_t degree, const float *wc1, uint32_t ld_c1, const float *wc2, uint32_t ld_c2, void *wsig16,
                         void *wcol16, const float *u, const float *rays_o, const float *rays_d, const float *aabb, float bound,
                         uint32_t N, uint32_t T, uint32_t T_tot, float near, float far, float *z, float *x01, float *features16,
                         float *cdir, lnh_stream_t stream) {
    const uint32_t n_dir = 3 + 6 * degree;
    LNH_REQUIRE(ws0 && ws1 && wc0 && wc1 && wc2 && wsig16 && wcol16 && rays_o && rays_d && aabb && z && x01 && features16 && cdir,
                LNH_ERR_INVALID_ARG, "lidar_step_prologue: null pointer");
    LNH_REQUIRE(n_dir <= 128 && ld_s0 >= 32 && ld_s1 >= 64 && ld_c0 >= n_dir + 15 && ld_c1 >= 64 && ld_c2 >= 64,
                LNH_ERR_INVALID_ARG, "lidar_step_prologue: leading dimension smaller than the row (or 3 + 6 * degree > 128)");
    LNH_REQUIRE(bound > 0.0f && T <= T_tot, LNH_ERR_INVALID_ARG, "lidar_step_prologue: bound must be positive, T <= T_tot");
    LNH_REQUIRE((uint64_t)N * T_tot < 0xffffffffull, LNH_ERR_UNSUPPORTED, "lidar_step_prologue: N*T must fit 32 bits");
    PackArgs<E> pk{ws0, ws1, wc0, wc1, wc2, ld_s0, ld_s1, ld_c0, ld_c1, ld_c2, n_dir, (E *)wsig16, (E *)wcol16};
    PrologueArgs a{rays_d, wc0, ld_c0, degree, features16, cdir, u, rays_o, aabb, bound, near, far, N, T, T_tot, z, x01};
    const uint32_t blocks = kPackBlocks + (N + 3) / 4 + div_up((uint64_t)N * T, 256);
    LNH_LAUNCH(k_step_prologue<E>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pk, a);
    return lnh_check_launch("lnh_lidar_step_prologue");
}

extern "C" {

int lnh_lidar_step_prologue(const float *ws0, uint32_t ld_s0, const float *ws1, uint32_t ld_s1, const float *wc0, uint32_t ld_c0,
                            uint32_t degree, const float *wc1, uint32_t ld_c1, const float *wc2, uint32_t ld_c2, void *wsig16,
                            void *wcol16, const float *u, const float *rays_o, const float *rays_d, const float *aabb,
                            float bound, uint32_t N, uint32_t T, uint32_t T_tot, float near, float far, float *z, float *x01,
                            float *features16, float *cdir, lnh_stream_t stream) {
    return step_prologue<half_t>(ws0, ld_s0, ws1, ld_s1, wc0, ld_c0, degree, wc1, ld_c1, wc2, ld_c2, wsig16, wcol16, u, rays_o,
                                 rays_d, aabb, bound, N, T, T_tot, near, far, z, x01, features16, cdir, stream);
}
int lnh_lidar_step_prologue_bf16(const float *ws0, uint32_t ld_s0, const float *ws1, uint32_t ld_s1, const float *wc0,
                                 uint32_t ld_c0, uint32_t degree, const float *wc1, uint32_t ld_c1, const float *wc2,
                                 uint32_t ld_c2, void *wsig16, void *wcol16, const float *u, const float *rays_o,
                                 const float *rays_d, const float *aabb, float bound, uint32_t N, uint32_t T, uint32_t T_tot,
                                 float near, float far, float *z, float *x01, float *features16, float *cdir,
                                 lnh_stream_t stream) {
    return step_prologue<__bf16>(ws0, ld_s0, ws1, ld_s1, wc0, ld_c0, degree, wc1, ld_c1, wc2, ld_c2, wsig16, wcol16, u, rays_o,
                                 rays_d, aabb, bound, N, T, T_tot, near, far, z, x01, features16, cdir, stream);
}

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

int lnh_lidar_dir_term_backward(const float *ray_sum, const float *features16, uint32_t N, uint32_t K,
                                const float *grad_w0g, float *grad_w0, uint32_t ldw, void *wgrad_ws, uint64_t wgrad_ws_bytes,
                                lnh_stream_t stream) {
    LNH_REQUIRE(ray_sum && features16 && grad_w0, LNH_ERR_INVALID_ARG, "lidar_dir_term_backward: null pointer");
    WgradWs ws;
    if (int rc = wgrad_ws_open(wgrad_ws, wgrad_ws_bytes, ws, "lidar_dir_term_backward")) return rc;
    LNH_REQUIRE(K >= 1 && K <= 128 && ldw >= K + (grad_w0g ? 15u : 0u), LNH_ERR_INVALID_ARG,
                "lidar_dir_term_backward: need 1 <= K <= 128 and ldw >= K (+ 15 with grad_w0g)");
    if (N == 0 && !grad_w0g) return LNH_OK;
    LNH_LAUNCH(k_dir_term_backward, dim3(K, kDirSegments), dim3(256), 0, (hipStream_t)stream, ray_sum, features16,
               N, K, grad_w0g, grad_w0, ldw, ws);
    wgrad_reduce_launch(ws, kDirSegments, K * 64, DirTermEmit{grad_w0, ldw}, (hipStream_t)stream);
    return lnh_check_launch("lnh_lidar_dir_term_backward");
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
                   float alpha_i, const float *grad_scale, float *loss, float *grad_depth, float *grad_image,
                   lnh_stream_t stream) {
    LNH_REQUIRE(depth && image && gt && loss && grad_depth && grad_image, LNH_ERR_INVALID_ARG,
                "lidar_loss: null pointer");
    LNH_REQUIRE(N <= (1u << 26), LNH_ERR_UNSUPPORTED, "lidar_loss: at most 2^26 rays per call");
    if (N == 0) return lnh_zero_async(loss, sizeof(float), (hipStream_t)stream, "lidar_loss (no rays)");
    LNH_LAUNCH(k_lidar_loss, dim3(1), dim3(kLossThreads), 0, (hipStream_t)stream, depth, image, gt, N, alpha_d,
               alpha_r, alpha_i, grad_scale, loss, grad_depth, grad_image);
    return lnh_check_launch("lnh_lidar_loss");
}

int lnh_lidar_loss_patch(const float *depth, const float *image, const float *gt, uint32_t N, uint32_t px, uint32_t py,
                         float scale, float alpha_d, float alpha_r, float alpha_i, float alpha_grad,
                         const float *grad_scale, float *loss, float *grad_depth, float *grad_image, lnh_stream_t stream) {
    LNH_REQUIRE(depth && image && gt && loss && grad_depth && grad_image, LNH_ERR_INVALID_ARG,
                "lidar_loss_patch: null pointer");
    LNH_REQUIRE(px >= 1 && py >= 2 && scale > 0.0f && N % (px * py) == 0, LNH_ERR_INVALID_ARG,
                "lidar_loss_patch: need py >= 2, scale > 0 and N a multiple of px * py");
    LNH_REQUIRE(N <= (1u << 26), LNH_ERR_UNSUPPORTED, "lidar_loss_patch: at most 2^26 rays per call");
    if (N == 0) return lnh_zero_async(loss, sizeof(float), (hipStream_t)stream, "lidar_loss_patch (no rays)");
    LNH_LAUNCH(k_lidar_loss_patch, dim3(1), dim3(kLossThreads), 0, (hipStream_t)stream, depth, image, gt, N, py,
               1.0f / scale, alpha_d, alpha_r, alpha_i, alpha_grad, grad_scale, loss, grad_depth, grad_image);
    return lnh_check_launch("lnh_lidar_loss_patch");
}

}  // extern "C"
