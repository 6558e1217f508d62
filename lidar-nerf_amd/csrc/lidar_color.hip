// lidar_color.hip — LiDAR colour head (ray-drop, intensity) of the fused render step on gfx950 MFMA: forward and
// backward over the merged samples of a ray (renderer.py:249-256, network.py:199-237 of the reference as ONE kernel each
// way).  See lidar_field.hip for the two algebraic moves (per-ray direction term, sigma-net row as the input) and
// mlp_common.h for the register-resident layer chaining and the fp16 / bf16 element-type builds
// (lidar_color_bf16.hip compiles this file again with bf16 MFMA operands: entry points lnh_lidar_color_*_bf16).
#include "mlp_common.h"
#include "wgrad.h"
#include <type_traits>

namespace LNH_MLP_NS {
namespace {

// ------------------------------------------------------------------------------------------------ colour head
// flat fp16 weights: W0g [64][16] (col 0 = 0, cols 1..15 = geo columns of the first Linear) | W1 [64][64] | W2 [16][64]
constexpr int kW0g = 0, kW1 = 64 * 16, kW2 = kW1 + 64 * 64, kWTotal = kW2 + 16 * 64;
constexpr float kMaskThresh = 1e-4f;  // renderer.py:249

struct ColorArgs {
    const half_t *h16;      // [N*T,16] sigma-net raw outputs, point order
    const int32_t *perm;    // [N,T] merged position -> slot of the ray
    const float *weights;   // [N,T] merged order
    const float *cdir;      // [N,64] fp32: W0_dir * freq(d) per ray
    const half_t *W;        // flat fp16 (kWTotal)
    float *rgb;             // fwd out [N,T,2]
    const float *g_rgb;     // bwd in  [N,T,2], or null: then g_rgb[n,i,:] = weights[n,i] * g_image[n,:] is formed on the fly
    const float *g_image;   // bwd in  [N,2] (used when g_rgb is null)
    const float *g_sigma;   // bwd in  [N,T] merged order
    half_t *g_h16;          // bwd out [N*T,16] point order
    float *dW;              // bwd out flat fp32 (kWTotal): the sum over all workgroups is ADDED to it, in a fixed order (wgrad.h)
    WgradWs ws;             // bwd: scratch of that sum
    float *S;               // bwd out [N,64] fp32: sum over the ray of d(hidden0 pre-activation)
    uint32_t N, T;
    // fused forward tail only (k_color_forward_ray): merge + weights + colour + compositing of a ray in one kernel
    const float *z;         // [N,T] merged (sorted) depths
    const float *sigma_pt;  // [N,T] densities in point order
    const float *sample_dist;  // [N]
    float density_scale;
    float *sigma_m;         // out [N,T] densities in merged order
    float *weights_out;     // out [N,T]
    float *wsum, *depth, *image;  // out [N], [N], [N,2]
    // ragged rays (occupancy-grid sampling, BASELINE config 4): the samples of ray-table entry r are the rows
    // rays[3r+1] .. + rays[3r+2] of the flat [M, *] buffers, in order (no permutation, no weight mask: every marched
    // sample lies in an occupied cell), rays[3r] is the ray's index into the per-ray buffers (cdir, S)
    const int32_t *rays;
    uint32_t M;
    float gs_scale;         // bwd: d loss / d sigma is multiplied by it (the density scale of the ragged path; 1 dense)
};

// v_exp_f32 / v_rcp_f32 (1 ulp each) instead of the IEEE division and the range-reduced expf: the two outputs of a
// sample are computed by whole-wave instructions, and the exact forms cost ~40 of them per output
__device__ __forceinline__ float sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// exp of the clamped density pre-activation (trunc_exp backward, activation.py:14-17)
__device__ __forceinline__ float exp_clamped(float x) { return __expf(fminf(fmaxf(x, -15.0f), 15.0f)); }

__global__ void __launch_bounds__(256)
k_color_forward(ColorArgs a) {
    constexpr int HT = 4, HS = 2, NT = 4;
    const uint32_t lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    half8_t w0[HT], w1[HT][HS], w2[HS];
#pragma unroll
    for (int t = 0; t < HT; t++) {
        w0[t] = load_a_natural(a.W + kW0g, 16, 16 * t + c, 0, g, 16);
#pragma unroll
        for (int s = 0; s < HS; s++) w1[t][s] = load_a_nu(a.W + kW1, 64, 16 * t + c, s, g);
    }
#pragma unroll
    for (int s = 0; s < HS; s++) w2[s] = load_a_nu(a.W + kW2, 64, c, s, g);

    const uint32_t total = a.N * a.T;  // < 2^32 (checked by the launcher)
    // weight -> (mask) -> perm -> sigma-net row is a chain of three dependent round trips per 64 samples, and a SIMD holds
    // only four such waves: the kernel ran at the speed of that chain.  The weights and the permutation of the NEXT
    // span are requested before the current one is processed (8 registers), so a span waits for its rows only.
    auto load_wp = [&](uint32_t base, float (&wgt)[NT], uint32_t (&slot)[NT]) {
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const uint32_t m = base + n * 16 + c;
            const uint32_t mc = m < total ? m : 0;  // unconditional loads + selects
            const float wv = a.weights[mc];
            slot[n] = (uint32_t)a.perm[mc];
            wgt[n] = m < total ? wv : 0.0f;
        }
    };
    float wgt[NT], wgt_next[NT];
    uint32_t slot[NT], slot_next[NT];
    const uint32_t stride = nwaves * NT * 16;
    load_wp(wave * NT * 16, wgt, slot);
    for (uint32_t base = wave * NT * 16; base < total; base += stride) {
        load_wp(base + stride < total ? base + stride : base, wgt_next, slot_next);
        bool msk[NT];
        bool any = false;
#pragma unroll
        for (int n = 0; n < NT; n++) {
            msk[n] = wgt[n] > kMaskThresh;
            any |= msk[n];
        }
        if (!__any(any)) {  // whole 64-sample span transparent: colour is defined as 0 there
            if (g == 0) {
#pragma unroll
                for (int n = 0; n < NT; n++) {
                    const uint32_t m = base + n * 16 + c;
                    if (m < total) *reinterpret_cast<float2 *>(a.rgb + (size_t)m * 2) = make_float2(0.0f, 0.0f);
                }
            }
        } else {
#pragma unroll
            for (int n = 0; n < NT; n++) {
                const uint32_t m = base + n * 16 + c;
                const bool valid = m < total;
                const uint32_t mc = valid ? m : 0, ray = mc / a.T;
                const size_t src = (size_t)ray * a.T + (valid ? slot[n] : 0u);
                const half8_t bxl = *reinterpret_cast<const half8_t *>(a.h16 + src * 16 + (g < 2 ? 8 * g : 0));
                const half8_t bx = (valid && g < 2) ? bxl : zero_h8();
                f32x4 acc[HT];
#pragma unroll
                for (int t = 0; t < HT; t++) acc[t] = *reinterpret_cast<const f32x4 *>(a.cdir + (size_t)ray * 64 + 16 * t + 4 * g);
#pragma unroll
                for (int t = 0; t < HT; t++) acc[t] = MFMA16(w0[t], bx, acc[t]);
                half8_t bh[HS];
#pragma unroll
                for (int s = 0; s < HS; s++)
                    bh[s] = pack_pair_relu(acc[2 * s], acc[2 * s + 1]);
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    acc[t] = zero_f4();
#pragma unroll
                    for (int s = 0; s < HS; s++) acc[t] = MFMA16(w1[t][s], bh[s], acc[t]);
                }
#pragma unroll
                for (int s = 0; s < HS; s++)
                    bh[s] = pack_pair_relu(acc[2 * s], acc[2 * s + 1]);
                f32x4 o = zero_f4();
#pragma unroll
                for (int s = 0; s < HS; s++) o = MFMA16(w2[s], bh[s], o);
                if (valid && g == 0) {
                    // the unfused path rounds the MLP output to fp16 before the sigmoid (autocast Linear output)
                    const float r0 = msk[n] ? sigmoidf((float)(half_t)o[0]) : 0.0f;
                    const float r1 = msk[n] ? sigmoidf((float)(half_t)o[1]) : 0.0f;
                    *reinterpret_cast<float2 *>(a.rgb + (size_t)m * 2) = make_float2(r0, r1);
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NT; n++) {
            wgt[n] = wgt_next[n];
            slot[n] = slot_next[n];
        }
    }
}

// Forward tail of a ray in ONE kernel, one wave per ray: merged densities and compositing weights (what
// k_merge_weights computes: renderer.py:217-243), the colour head on the samples above the mask threshold, and the
// compositing sums (k_lidar_composite_fwd: weights_sum, depth = sum w z, image = sum w rgb).  The three kernels it
// replaces each walked the ray at HBM latency and handed weights / rgb to the next through HBM; here the weights and
// the permutation of the ray stay in LDS between the scan and the colour head (6.6 KB per wave at 832 samples).
// sigma_m, weights and rgb are still written: the backward pass reads them.  weights_sum and depth are summed in the
// order of k_lidar_composite_fwd (bit-identical); image is summed by the lanes that hold the colours (same value to fp32
// rounding).
__global__ void __launch_bounds__(256)
k_color_forward_ray(ColorArgs a) {
    constexpr int HT = 4, HS = 2, NT = 4, kChunks = 7;
    extern __shared__ __attribute__((aligned(16))) char smem_ray[];
    const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const uint32_t nw = blockDim.x >> 6, nwaves = gridDim.x * nw;
    float *wl = reinterpret_cast<float *>(smem_ray) + (size_t)wid * 2 * a.T;   // this wave's weights [T]
    uint32_t *sl = reinterpret_cast<uint32_t *>(wl + a.T);                     // ... and slots [T]
    half8_t w0[HT], w1[HT][HS], w2[HS];
#pragma unroll
    for (int t = 0; t < HT; t++) {
        w0[t] = load_a_natural(a.W + kW0g, 16, 16 * t + c, 0, g, 16);
#pragma unroll
        for (int s = 0; s < HS; s++) w1[t][s] = load_a_nu(a.W + kW1, 64, 16 * t + c, s, g);
    }
#pragma unroll
    for (int s = 0; s < HS; s++) w2[s] = load_a_nu(a.W + kW2, 64, c, s, g);

    for (uint32_t ray = blockIdx.x * nw + wid; ray < a.N; ray += nwaves) {
        const float *zr = a.z + (size_t)ray * a.T, *sp = a.sigma_pt + (size_t)ray * a.T;
        const int32_t *pr = a.perm + (size_t)ray * a.T;
        const float sd = a.sample_dist[ray];
        // ---- 1) merged densities, transmittance scan, weights; weights_sum and depth
        float carry = 1.0f, ws = 0.0f, dep = 0.0f;
        for (uint32_t base0 = 0; base0 < a.T; base0 += 64 * kChunks) {
            float zi[kChunks], zn[kChunks], sg[kChunks];
            int32_t pi[kChunks];
#pragma unroll
            for (int u = 0; u < kChunks; u++) {
                const uint32_t i = base0 + u * 64 + lane, ic = i < a.T ? i : 0, in = i + 1 < a.T ? i + 1 : ic;
                pi[u] = pr[ic];
                zi[u] = zr[ic];
                zn[u] = zr[in];
            }
#pragma unroll
            for (int u = 0; u < kChunks; u++) sg[u] = sp[pi[u]];
#pragma unroll
            for (int u = 0; u < kChunks; u++) {
                const uint32_t i = base0 + u * 64 + lane;
                if (base0 + u * 64 >= a.T) break;  // wave-uniform
                float alpha = 0.0f, om = 1.0f;
                if (i < a.T) {
                    const float delta = (i + 1 < a.T) ? (zn[u] - zi[u]) : sd;
                    alpha = 1.0f - expf(-delta * a.density_scale * sg[u]);
                    om = 1.0f - alpha + 1e-15f;
                }
                const float incl = wave_scan_mul(om, (int)lane);
                float excl = __shfl_up(incl, 1, 64);
                if (lane == 0) excl = 1.0f;
                const float w = alpha * (carry * excl);
                if (i < a.T) {
                    a.sigma_m[(size_t)ray * a.T + i] = sg[u];
                    a.weights_out[(size_t)ray * a.T + i] = w;
                    wl[i] = w;
                    sl[i] = (uint32_t)pi[u];
                }
                ws += w;
                dep += i < a.T ? w * zi[u] : 0.0f;
                carry *= __shfl(incl, 63, 64);
            }
        }
        ws = wave_sum(ws);
        dep = wave_sum(dep);
        // ---- 2) colour head on the spans that hold a sample above the mask threshold; image sums
        f32x4 cb[HT];
#pragma unroll
        for (int t = 0; t < HT; t++) cb[t] = *reinterpret_cast<const f32x4 *>(a.cdir + (size_t)ray * 64 + 16 * t + 4 * g);
        float img0 = 0.0f, img1 = 0.0f;
        for (uint32_t s0 = 0; s0 < a.T; s0 += NT * 16) {
            float wgt[NT];
            uint32_t slot[NT];
            bool valid[NT], msk[NT], any = false;
#pragma unroll
            for (int n = 0; n < NT; n++) {
                const uint32_t i = s0 + 16 * n + c;
                valid[n] = i < a.T;
                wgt[n] = valid[n] ? wl[i] : 0.0f;
                slot[n] = valid[n] ? sl[i] : 0u;
                msk[n] = wgt[n] > kMaskThresh;
                any |= msk[n];
            }
            float *rgb_row = a.rgb + ((size_t)ray * a.T + s0) * 2;
            if (!__any(any)) {  // whole span transparent: colour is defined as 0 there
                if (g == 0) {
#pragma unroll
                    for (int n = 0; n < NT; n++)
                        if (valid[n]) *reinterpret_cast<float2 *>(rgb_row + (16 * n + c) * 2) = make_float2(0.0f, 0.0f);
                }
                continue;
            }
            half8_t bx[NT];
#pragma unroll
            for (int n = 0; n < NT; n++) {
                const size_t src = (size_t)ray * a.T + slot[n];
                const half8_t v = *reinterpret_cast<const half8_t *>(a.h16 + src * 16 + (g < 2 ? 8 * g : 0));
                bx[n] = (valid[n] && g < 2) ? v : zero_h8();
            }
#pragma unroll
            for (int n = 0; n < NT; n++) {
                f32x4 acc[HT];
#pragma unroll
                for (int t = 0; t < HT; t++) acc[t] = MFMA16(w0[t], bx[n], cb[t]);
                half8_t bh[HS];
#pragma unroll
                for (int s = 0; s < HS; s++) bh[s] = pack_pair_relu(acc[2 * s], acc[2 * s + 1]);
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    acc[t] = zero_f4();
#pragma unroll
                    for (int s = 0; s < HS; s++) acc[t] = MFMA16(w1[t][s], bh[s], acc[t]);
                }
#pragma unroll
                for (int s = 0; s < HS; s++) bh[s] = pack_pair_relu(acc[2 * s], acc[2 * s + 1]);
                f32x4 o = zero_f4();
#pragma unroll
                for (int s = 0; s < HS; s++) o = MFMA16(w2[s], bh[s], o);
                if (valid[n] && g == 0) {
                    // the unfused path rounds the MLP output to fp16 before the sigmoid (autocast Linear output)
                    const float r0 = msk[n] ? sigmoidf((float)(half_t)o[0]) : 0.0f;
                    const float r1 = msk[n] ? sigmoidf((float)(half_t)o[1]) : 0.0f;
                    *reinterpret_cast<float2 *>(rgb_row + (16 * n + c) * 2) = make_float2(r0, r1);
                    img0 += wgt[n] * r0;
                    img1 += wgt[n] * r1;
                }
            }
        }
        // lanes 0..15 (g == 0) hold the partial image sums
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            img0 += __shfl_xor(img0, d, 64);
            img1 += __shfl_xor(img1, d, 64);
        }
        if (lane == 0) {
            a.wsum[ray] = ws;
            a.depth[ray] = dep;
            *reinterpret_cast<float2 *>(a.image + (size_t)ray * 2) = make_float2(img0, img1);
        }
    }
}

// The colour head on RAGGED rays (occupancy-grid sampling: renderer.run_cuda's samples, BASELINE config 4), one wave per
// ray-table entry: the same two algebraic moves as the dense chain — the direction part of the first layer once per RAY
// (cdir), the sample's sigma-net row as the 16-wide input — where rounds 2-4 assembled a [M, 96] input ([freq(d) | geo |
// 0]) and ran the generic 96 -> 64 -> 64 -> 16 MFMA MLP on it (three launches and a K = 96 first layer per sample).  No
// permutation and no weight mask here: every marched sample is evaluated.
__global__ void __launch_bounds__(256)
k_color_forward_ragged(ColorArgs a) {
    constexpr int HT = 4, HS = 2, NT = 4;
    const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const uint32_t nw = blockDim.x >> 6, nwaves = gridDim.x * nw;
    half8_t w0[HT], w1[HT][HS], w2[HS];
#pragma unroll
    for (int t = 0; t < HT; t++) {
        w0[t] = load_a_natural(a.W + kW0g, 16, 16 * t + c, 0, g, 16);
#pragma unroll
        for (int s = 0; s < HS; s++) w1[t][s] = load_a_nu(a.W + kW1, 64, 16 * t + c, s, g);
    }
#pragma unroll
    for (int s = 0; s < HS; s++) w2[s] = load_a_nu(a.W + kW2, 64, c, s, g);
    for (uint32_t r = blockIdx.x * nw + wid; r < a.N; r += nwaves) {
        const uint32_t rid = (uint32_t)a.rays[3 * r], off = (uint32_t)a.rays[3 * r + 1];
        uint32_t cnt = (uint32_t)a.rays[3 * r + 2];
        if (off + cnt > a.M) cnt = 0;  // (a ray the marcher dropped: its samples do not exist)
        if (cnt == 0) continue;
        f32x4 cb[HT];
#pragma unroll
        for (int t = 0; t < HT; t++) cb[t] = *reinterpret_cast<const f32x4 *>(a.cdir + (size_t)(rid < a.N ? rid : 0) * 64 + 16 * t + 4 * g);
        for (uint32_t s0 = 0; s0 < cnt; s0 += NT * 16) {
            half8_t bx[NT];
            bool valid[NT];
#pragma unroll
            for (int n = 0; n < NT; n++) {  // (all rows of the span requested before the first is used)
                const uint32_t i = s0 + 16 * n + c;
                valid[n] = i < cnt;
                const size_t row = (size_t)off + (valid[n] ? i : 0u);
                const half8_t v = *reinterpret_cast<const half8_t *>(a.h16 + row * 16 + (g < 2 ? 8 * g : 0));
                bx[n] = (valid[n] && g < 2) ? v : zero_h8();
            }
#pragma unroll
            for (int n = 0; n < NT; n++) {
                if (s0 + 16 * n >= cnt) break;  // wave-uniform
                f32x4 acc[HT];
#pragma unroll
                for (int t = 0; t < HT; t++) acc[t] = MFMA16(w0[t], bx[n], cb[t]);
                half8_t bh[HS];
#pragma unroll
                for (int s = 0; s < HS; s++) bh[s] = pack_pair_relu(acc[2 * s], acc[2 * s + 1]);
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    acc[t] = zero_f4();
#pragma unroll
                    for (int s = 0; s < HS; s++) acc[t] = MFMA16(w1[t][s], bh[s], acc[t]);
                }
#pragma unroll
                for (int s = 0; s < HS; s++) bh[s] = pack_pair_relu(acc[2 * s], acc[2 * s + 1]);
                f32x4 o = zero_f4();
#pragma unroll
                for (int s = 0; s < HS; s++) o = MFMA16(w2[s], bh[s], o);
                if (valid[n] && g == 0) {
                    // the unfused path rounds the MLP output to fp16 before the sigmoid (autocast Linear output)
                    const float r0 = sigmoidf((float)(half_t)o[0]), r1 = sigmoidf((float)(half_t)o[1]);
                    *reinterpret_cast<float2 *>(a.rgb + ((size_t)off + s0 + 16 * n + c) * 2) = make_float2(r0, r1);
                }
            }
        }
    }
}

// Wave-independent colour-head backward: one WAVE per ray at a time, 32 merged samples per iteration, no LDS tiles and
// no barriers in the sample loop.  Weight gradients contract over samples, so their MFMA operands are the TRANSPOSES of
// what the layer chain leaves in registers; a transpose of a packed fp16 fragment is one MFMA against an identity
// fragment (the A and B operand layouts are mirror images: row/col = lane & 15, same k enumeration), which is exact.
// Every wave owns all 24 gradient tiles (dW2 4, dW1 16, dW0g 4) in accumulators; the four waves of a workgroup are
// combined through LDS once at the end (wave order) and the workgroups' sums are added up in index order (wgrad.h): the
// gradient carries the same bits on every run.  (Rounds 1-5: one fp32 device atomic per weight and workgroup.)
// Round 5 built the transposes on gfx950's transposing LDS read as well (ds_read_b64_tr_b16: the sigma-net backward in
// mlp_bwd.h runs on it) — for THIS kernel, one wave per SIMD, both LDS forms lost to the identity MFMAs (283.9 / 291.4
// against 268.4 us: profiles/r05_color_backward_wgrad.txt) and were removed again.
// RAGGED: the rays are entries of a ray table over flat sample buffers (see ColorArgs::rays) instead of rows of [N, T] arrays.
template <bool FROM_IMAGE, bool RAGGED = false>  // FROM_IMAGE: d loss / d rgb formed as weights (x) a.g_image, else read from a.g_rgb
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_color_backward_wi(ColorArgs a) {
    static_assert(!(FROM_IMAGE && RAGGED), "the ragged compositing backward hands over d loss / d rgb itself");
    constexpr int HT = 4, HS = 2, NT = 2, NTILE = HT + HT * HT + HT;
    // The 28 weight fragments (112 registers) stay in registers for the whole kernel: the 24 gradient tiles live in
    // AGPRs (mfma16_acc_agpr), which leaves the VGPRs to the fragments and the pipeline state, and hipcc places the
    // fragments that do not fit there in AGPRs as well (an MFMA reads its A operand from either file).  Re-reading them
    // from LDS every iteration (48 ds_read_b128 issued just ahead of their MFMAs) cost 11 % of the kernel; sharing them
    // between two waves per SIMD that way was built in round 4 and lost (434 against 217 us: profiles/r04_color_backward_2wave.txt).
    enum { F_W0 = 0, F_W1 = 4, F_W2 = 12, F_W2T = 14, F_W1T = 18, F_W0T = 26, NFRAG = 28 };
    __shared__ float red[NTILE * 256];
    const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const uint32_t nw = blockDim.x >> 6, nwaves = gridDim.x * nw;
    const uint32_t wave = blockIdx.x * nw + (uint32_t)__builtin_amdgcn_readfirstlane((int)wid);  // known wave-uniform
    half8_t wreg[NFRAG];
    auto frag = [&](int i) -> half8_t {  // fragment i of this lane (see the F_* enumeration)
        if (i < F_W1) return load_a_natural(a.W + kW0g, 16, 16 * (i - F_W0) + c, 0, g, 16);
        if (i < F_W2) return load_a_nu(a.W + kW1, 64, 16 * ((i - F_W1) >> 1) + c, (i - F_W1) & 1, g);
        if (i < F_W2T) return load_a_nu(a.W + kW2, 64, c, i - F_W2, g);
        if (i < F_W1T) return load_at_natural(a.W + kW2, 64, 16 * (i - F_W2T) + c, 0, g, 16);
        if (i < F_W0T) return load_at_nu(a.W + kW1, 64, 16 * ((i - F_W1T) >> 1) + c, (i - F_W1T) & 1, g);
        return load_at_nu(a.W + kW0g, 16, c, i - F_W0T, g);
    };
#pragma unroll
    for (int i = 0; i < NFRAG; i++) wreg[i] = frag(i);
#define WF(i) wreg[(i)]
    // identity fragments: idn selects natural-k element c (k = 8g + j); idv[tt] selects nu-enumerated channel
    // 16 * (2s + tt) + c out of k-step s (element j of lane group g is channel 16 * (2s + (j >> 2)) + 4g + (j & 3))
    half8_t idn, idv[2];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        idn[j] = (8 * g + j == c) ? (half_t)1.0f : (half_t)0.0f;
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
            idv[tt][j] = ((uint32_t)(j >> 2) == (uint32_t)tt && 4 * g + (j & 3) == c) ? (half_t)1.0f : (half_t)0.0f;
    }
    f32x4 gW2[HT], gW1[HT][HT], gW0[HT];
#pragma unroll
    for (int t = 0; t < HT; t++) {
        gW2[t] = zero_f4();
        gW0[t] = zero_f4();
#pragma unroll
        for (int i = 0; i < HT; i++) gW1[t][i] = zero_f4();
    }
    auto pack2 = [](const f32x4 &lo, const f32x4 &hi) {
        half8_t r = {(half_t)lo[0], (half_t)lo[1], (half_t)lo[2], (half_t)lo[3],
                     (half_t)hi[0], (half_t)hi[1], (half_t)hi[2], (half_t)hi[3]};
        return r;
    };

    // ---- work items.  A wave walks its rays (wave, wave + nwaves, ...) and, inside a ray, only the 32-sample spans that
    // hold at least one sample above the mask threshold.  The iterator is wave-uniform (scalar registers): when it enters a
    // ray (a group of 32 spans of it) it reads the ray's weights one sample per lane, forms the span bits from ballots,
    // and stores the gradient rows of the TRANSPARENT spans right there — colour is defined as 0 on them, only the
    // compositing gradient of sigma flows back.  The loop below therefore sees active spans only, and its body has no
    // branch around the MFMA chain: with a transparent / active diamond inside the loop hipcc copied all 96
    // weight-gradient accumulators to other registers and back at every latch (192 moves per iteration).
    constexpr uint32_t kGroupSpans = 32, kGroupSamples = 32 * kGroupSpans, kGroupLoads = kGroupSamples / 64;
    const uint32_t nsteps = (a.T + 31) / 32, ngroups_dense = (nsteps + kGroupSpans - 1) / kGroupSpans;
    const uint32_t nrays = wave < a.N ? (a.N - wave + nwaves - 1) / nwaves : 0;
    struct Iter {
        uint32_t rr, grp, act;
        bool fresh, ray_any;
        uint32_t base, cnt, rid, ng;  // RAGGED: first row, sample count, ray index, span groups of the open ray
    };
    Iter it = {0u, 0u, 0u, true, false, 0u, 0u, 0u, ngroups_dense};
    struct Item {
        bool live;
        uint32_t ray, s0;
        uint32_t base, cnt, rid;  // (RAGGED)
    };
    // geometry of table entry r (RAGGED; wave-uniform scalar loads).  A ray whose samples do not fit the buffers was
    // dropped by the marcher: it has no samples here either.
    auto open_ray = [&](uint32_t r) {
        if constexpr (RAGGED) {
            const uint32_t off = (uint32_t)a.rays[3 * r + 1], cnt = (uint32_t)a.rays[3 * r + 2];
            it.rid = (uint32_t)a.rays[3 * r];
            it.base = off;
            it.cnt = off + cnt <= a.M ? cnt : 0u;
            it.ng = ((it.cnt + 31) / 32 + kGroupSpans - 1) / kGroupSpans;
            if (it.ng == 0) it.ng = 1;
        }
    };
    auto enter_group = [&](uint32_t ray, uint32_t grp) -> uint32_t {
        if constexpr (RAGGED) {  // every span that holds a sample is active; nothing transparent to write
            uint32_t act = 0;
#pragma unroll
            for (uint32_t sp = 0; sp < kGroupSpans; sp++)
                if ((grp * kGroupSpans + sp) * 32 < it.cnt) act |= 1u << sp;
            return act;
        }
        const uint32_t i0 = grp * kGroupSamples;
        bool v[kGroupLoads];
        uint32_t m[kGroupLoads];
        float w[kGroupLoads];
#pragma unroll
        for (uint32_t j = 0; j < kGroupLoads; j++) {
            const uint32_t i = i0 + j * 64 + lane;
            v[j] = i < a.T;
            m[j] = v[j] ? ray * a.T + i : 0;  // N*T < 2^32 (checked by the launcher)
            w[j] = a.weights[m[j]];
        }
        uint32_t act = 0;
#pragma unroll
        for (uint32_t j = 0; j < kGroupLoads; j++) {
            const unsigned long long b = __ballot(v[j] && w[j] > kMaskThresh);
            act |= ((uint32_t)b != 0u ? 1u : 0u) << (2 * j);
            act |= ((uint32_t)(b >> 32) != 0u ? 1u : 0u) << (2 * j + 1);
        }
        // rows of the transparent spans (lane = sample): d(row) = (g_sigma * exp(h0), 0, ..., 0); loads batched, all
        // unconditional from clamped indices; skipped (wave-uniform) when every span of the group is active
        bool tr[kGroupLoads], any_tr = false;
#pragma unroll
        for (uint32_t j = 0; j < kGroupLoads; j++) {
            tr[j] = v[j] && !((act >> (2 * j + (lane >> 5))) & 1u);
            any_tr |= tr[j];
        }
        if (__any(any_tr)) {
            uint32_t slot[kGroupLoads];
            float gs[kGroupLoads];
            half_t h0[kGroupLoads];
#pragma unroll
            for (uint32_t j = 0; j < kGroupLoads; j++) {
                slot[j] = (uint32_t)a.perm[m[j]];
                gs[j] = a.g_sigma[m[j]];
            }
#pragma unroll
            for (uint32_t j = 0; j < kGroupLoads; j++) {
                const size_t src = v[j] ? (size_t)ray * a.T + slot[j] : (size_t)0;
                h0[j] = a.h16[src * 16];
            }
#pragma unroll
            for (uint32_t j = 0; j < kGroupLoads; j++) {
                if (tr[j]) {
                    half8_t lo = zero_h8();
                    lo[0] = (half_t)(gs[j] * exp_clamped((float)h0[j]));
                    half8_t *q = reinterpret_cast<half8_t *>(a.g_h16 + ((size_t)ray * a.T + slot[j]) * 16);
                    q[0] = lo;
                    q[1] = zero_h8();
                }
            }
        }
        return act;
    };
    auto zero_ray_sum = [&](uint32_t ray) { a.S[(size_t)(RAGGED ? it.rid : ray) * 64 + lane] = 0.0f; };
    auto next_item = [&]() {
        Item I = {false, 0u, 0u, 0u, 0u, 0u};
        while (it.act == 0u) {
            if (!it.fresh) {
                if (++it.grp >= (RAGGED ? it.ng : ngroups_dense)) {
                    if (!it.ray_any && it.rr < nrays) zero_ray_sum(wave + it.rr * nwaves);  // a ray without any active span
                    it.ray_any = false;
                    it.grp = 0;
                    if (it.rr < nrays) it.rr++;
                    if (RAGGED && it.rr < nrays) open_ray(wave + it.rr * nwaves);
                }
            } else if (RAGGED && it.rr < nrays) {
                open_ray(wave + it.rr * nwaves);
            }
            if (it.rr >= nrays) return I;  // exhausted (stays exhausted: rr no longer moves)
            it.fresh = false;
            it.act = enter_group(wave + it.rr * nwaves, it.grp);
            it.ray_any = it.ray_any || it.act != 0u;
        }
        const uint32_t sp = (uint32_t)__builtin_ctz(it.act);
        it.act &= it.act - 1u;
        I.live = true;
        I.ray = wave + it.rr * nwaves;
        I.s0 = (it.grp * kGroupSpans + sp) * 32;
        if constexpr (RAGGED) {
            I.base = it.base;
            I.cnt = it.cnt;
            I.rid = it.rid;
        }
        return I;
    };

    // The per-sample operands form a dependent chain (perm -> sigma-net row) of HBM round trips; left in program order
    // they cost ~4 exposed latencies per step.  The item sequence is therefore software-pipelined: stage A (weights,
    // perm, incoming gradients) runs two items ahead, stage B (the gathered sigma-net row) one item ahead.  All loads
    // are unconditional from clamped addresses.
    struct StageA {
        bool live, valid[NT];
        uint32_t ray, m[NT], slot[NT];
        float wgt[NT], gs[NT];
        float2 gr[NT];
        uint32_t base, rid;  // (RAGGED)
    };
    auto load_a = [&](const Item &I) {
        StageA A;
        A.live = I.live;
        A.ray = I.ray;
        A.base = I.base;
        A.rid = I.rid;
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const uint32_t i = I.s0 + 16 * n + c;
            if constexpr (RAGGED) {
                A.valid[n] = I.live && i < I.cnt;
                A.m[n] = A.valid[n] ? I.base + i : 0;
                A.wgt[n] = 1.0f;  // every marched sample is evaluated (msk = valid)
                A.slot[n] = i;
                A.gs[n] = a.g_sigma[A.m[n]];
                A.gr[n] = *reinterpret_cast<const float2 *>(a.g_rgb + (size_t)A.m[n] * 2);
                continue;
            }
            A.valid[n] = I.live && i < a.T;
            A.m[n] = A.valid[n] ? I.ray * a.T + i : 0;
            A.wgt[n] = a.weights[A.m[n]];
            A.slot[n] = (uint32_t)a.perm[A.m[n]];
            A.gs[n] = a.g_sigma[A.m[n]];
            if constexpr (!FROM_IMAGE) {
                A.gr[n] = *reinterpret_cast<const float2 *>(a.g_rgb + (size_t)A.m[n] * 2);
            } else {  // what lnh_lidar_composite_backward would have written: w * d loss / d image
                const float2 gi = *reinterpret_cast<const float2 *>(a.g_image + (size_t)(I.live ? I.ray : 0) * 2);
                A.gr[n] = make_float2(A.wgt[n] * gi.x, A.wgt[n] * gi.y);
            }
        }
        return A;
    };
    auto src_of = [&](const StageA &A, int n) {
        if constexpr (RAGGED) return A.valid[n] ? (size_t)A.base + A.slot[n] : (size_t)0;
        else return A.valid[n] ? (size_t)A.ray * a.T + A.slot[n] : (size_t)0;
    };
    struct StageB { half8_t x[NT]; };
    auto load_b = [&](const StageA &A) {
        StageB B;
#pragma unroll
        for (int n = 0; n < NT; n++)
            B.x[n] = *reinterpret_cast<const half8_t *>(a.h16 + src_of(A, n) * 16 + (g < 2 ? 8 * g : 0));
        return B;
    };
    auto load_cb = [&](uint32_t ray, f32x4 (&cb)[HT]) {  // (RAGGED: `ray` is the ray INDEX rays[3r] of the table entry)
        const uint32_t r = ray < a.N ? ray : 0;
#pragma unroll
        for (int t = 0; t < HT; t++) cb[t] = *reinterpret_cast<const f32x4 *>(a.cdir + (size_t)r * 64 + 16 * t + 4 * g);
    };
    // S[ray][channel 16t + c] = sum over the ray's samples of dH0 (lane (g,c) holds samples 4g + r)
    auto store_ray_sum = [&](uint32_t ray, const float (&ssum)[HT]) {
#pragma unroll
        for (int t = 0; t < HT; t++) {
            float v = ssum[t];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g == 0) a.S[(size_t)ray * 64 + 16 * t + c] = v;
        }
    };
    StageA A0 = load_a(next_item()), A1 = load_a(next_item());
    StageB B0 = load_b(A0);
    f32x4 cb[HT], cb_next[HT];
    load_cb(RAGGED ? A0.rid : A0.ray, cb_next);
    float ssum[HT] = {0.0f, 0.0f, 0.0f, 0.0f};
    bool first_of_ray = true;

    while (A0.live) {
        const uint32_t ray = A0.ray;
        const StageA A2 = load_a(next_item());
        const StageB B1 = load_b(A1);
        if (first_of_ray) {
#pragma unroll
            for (int t = 0; t < HT; t++) {
                cb[t] = cb_next[t];
                ssum[t] = 0.0f;
            }
        }
        const bool last_of_ray = !A1.live || A1.ray != ray;
        if (last_of_ray) load_cb(RAGGED ? A1.rid : A1.ray, cb_next);   // the next item opens another ray: its direction term, one item ahead
        bool msk[NT];
        half8_t bx[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) {
            msk[n] = A0.valid[n] && A0.wgt[n] > kMaskThresh;
            bx[n] = (A0.valid[n] && g < 2) ? B0.x[n] : zero_h8();
        }
        {
            half8_t by[NT], bh0[NT][HS], bh1[NT][HS], bd1[NT][HS], bd0[NT][HS];
            // the two 16-sample tiles go through every layer side by side: one tile's MFMAs run while the other tile's
            // results are narrowed and activated (a single wave per SIMD has nothing else to hide MFMA latency with)
            f32x4 acc[NT][HT];
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int t = 0; t < HT; t++) acc[n][t] = MFMA16(WF(F_W0 + t), bx[n], cb[t]);
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int s = 0; s < HS; s++) bh0[n][s] = pack_pair_relu(acc[n][2 * s], acc[n][2 * s + 1]);
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    acc[n][t] = zero_f4();
#pragma unroll
                    for (int s = 0; s < HS; s++) acc[n][t] = MFMA16(WF(F_W1 + 2 * t + s), bh0[n][s], acc[n][t]);
                }
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int s = 0; s < HS; s++) bh1[n][s] = pack_pair_relu(acc[n][2 * s], acc[n][2 * s + 1]);
            f32x4 o[NT];
#pragma unroll
            for (int n = 0; n < NT; n++) {
                o[n] = zero_f4();
#pragma unroll
                for (int s = 0; s < HS; s++) o[n] = MFMA16(WF(F_W2 + s), bh1[n][s], o[n]);
            }
            // output gradient through the sigmoid (only outputs 0,1 exist; lanes g == 0 hold them)
#pragma unroll
            for (int n = 0; n < NT; n++) {
                by[n] = zero_h8();
                if (g == 0 && msk[n]) {
                    const float r0 = sigmoidf((float)(half_t)o[n][0]), r1 = sigmoidf((float)(half_t)o[n][1]);
                    by[n][0] = (half_t)(A0.gr[n].x * r0 * (1.0f - r0));
                    by[n][1] = (half_t)(A0.gr[n].y * r1 * (1.0f - r1));
                }
            }
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int t = 0; t < HT; t++) acc[n][t] = MFMA16(WF(F_W2T + t), by[n], zero_f4());
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int s = 0; s < HS; s++)
                    bd1[n][s] = pack_pair_relu_bwd(acc[n][2 * s], acc[n][2 * s + 1], bh1[n][s]);
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    acc[n][t] = zero_f4();
#pragma unroll
                    for (int s = 0; s < HS; s++) acc[n][t] = MFMA16(WF(F_W1T + 2 * t + s), bd1[n][s], acc[n][t]);
                }
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int s = 0; s < HS; s++)
                    bd0[n][s] = pack_pair_relu_bwd(acc[n][2 * s], acc[n][2 * s + 1], bh0[n][s]);
            // d(sigma-net row) = W0g^T dH0, col 0 <- trunc_exp backward of the compositing gradient
#pragma unroll
            for (int n = 0; n < NT; n++) {
                f32x4 dx = zero_f4();
#pragma unroll
                for (int s = 0; s < HS; s++) dx = MFMA16(WF(F_W0T + s), bd0[n][s], dx);
                if (A0.valid[n]) {
                    if (g == 0) dx[0] = (RAGGED ? A0.gs[n] * a.gs_scale : A0.gs[n]) * exp_clamped((float)bx[n][0]);
                    half4_t v = {(half_t)dx[0], (half_t)dx[1], (half_t)dx[2], (half_t)dx[3]};
                    *reinterpret_cast<half4_t *>(a.g_h16 + src_of(A0, n) * 16 + 4 * g) = v;
                }
            }
            // ---- sample-major operands (exact transposes) and the weight gradients
            const half8_t fy = pack2(MFMA16(by[0], idn, zero_f4()), MFMA16(by[1], idn, zero_f4()));
            const half8_t fx = pack2(MFMA16(bx[0], idn, zero_f4()), MFMA16(bx[1], idn, zero_f4()));
            half8_t fh0[HT], fd1[HT];
#pragma unroll
            for (int t = 0; t < HT; t++) {
                fh0[t] = pack2(MFMA16(bh0[0][t >> 1], idv[t & 1], zero_f4()), MFMA16(bh0[1][t >> 1], idv[t & 1], zero_f4()));
                fd1[t] = pack2(MFMA16(bd1[0][t >> 1], idv[t & 1], zero_f4()), MFMA16(bd1[1][t >> 1], idv[t & 1], zero_f4()));
            }
#pragma unroll
            for (int t = 0; t < HT; t++) {
                const half8_t fh1 = pack2(MFMA16(bh1[0][t >> 1], idv[t & 1], zero_f4()),
                                          MFMA16(bh1[1][t >> 1], idv[t & 1], zero_f4()));
                mfma16_acc_agpr(gW2[t], fy, fh1);  // dW2[o = 4g + r][16t + c]
                const f32x4 e0 = MFMA16(bd0[0][t >> 1], idv[t & 1], zero_f4());
                const f32x4 e1 = MFMA16(bd0[1][t >> 1], idv[t & 1], zero_f4());
                ssum[t] += (e0[0] + e0[1]) + (e0[2] + e0[3]) + (e1[0] + e1[1]) + (e1[2] + e1[3]);
                mfma16_acc_agpr(gW0[t], pack2(e0, e1), fx);  // dW0g[16t + 4g + r][c]
#pragma unroll
                for (int i = 0; i < HT; i++) mfma16_acc_agpr(gW1[t][i], fd1[t], fh0[i]);  // dW1[16t + 4g + r][16i + c]
            }
        }
        if (last_of_ray) store_ray_sum(RAGGED ? A0.rid : ray, ssum);
        first_of_ray = last_of_ray;
        A0 = A1;
        A1 = A2;
        B0 = B1;
    }
#pragma unroll
    for (int t = 0; t < HT; t++) {
        agpr_settle(gW2[t]);
        agpr_settle(gW0[t]);
#pragma unroll
        for (int i = 0; i < HT; i++) agpr_settle(gW1[t][i]);
    }

    // ---- combine the waves of the workgroup through LDS (wave order); the workgroups' sums are added in index order by the
    //      launcher's second launch (k_wgrad_reduce, wgrad.h)
    for (uint32_t w = 0; w < nw; w++) {
        if (wid == w) {
#pragma unroll
            for (int t = 0; t < HT; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float *q = red + (t * 4 + r) * 64 + lane;
                    *q = (w == 0 ? 0.0f : *q) + gW2[t][r];
                    q = red + ((HT + HT * HT + t) * 4 + r) * 64 + lane;
                    *q = (w == 0 ? 0.0f : *q) + gW0[t][r];
#pragma unroll
                    for (int i = 0; i < HT; i++) {
                        q = red + ((HT + t * HT + i) * 4 + r) * 64 + lane;
                        *q = (w == 0 ? 0.0f : *q) + gW1[t][i][r];
                    }
                }
        }
        __syncthreads();
    }
    {
        float4 *mine = reinterpret_cast<float4 *>(wgrad_partial(a.ws, NTILE * 256));
        for (uint32_t e = threadIdx.x; e < NTILE * 64; e += blockDim.x) mine[e] = reinterpret_cast<const float4 *>(red)[e];
    }
#undef WF
}
}  // namespace

extern "C" {

int LNH_MLP_FN(lnh_lidar_color_forward)(const void *h16, const int32_t *perm, const float *weights, const float *cdir,
                            const void *w16, uint32_t N, uint32_t T, float *rgb, lnh_stream_t stream) {
    LNH_REQUIRE(h16 && perm && weights && cdir && w16 && rgb, LNH_ERR_INVALID_ARG, "lidar_color_forward: null pointer");
    if (N == 0 || T == 0) return LNH_OK;
    LNH_REQUIRE((uint64_t)N * T < 0xffffffffull, LNH_ERR_UNSUPPORTED, "lidar_color_forward: N*T must fit 32 bits");
    ColorArgs a{};
    a.h16 = (const half_t *)h16; a.perm = perm; a.weights = weights; a.cdir = cdir; a.W = (const half_t *)w16;
    a.rgb = rgb; a.N = N; a.T = T;
    const uint32_t tiles = div_up((uint64_t)N * T, 4 * 16 * 4);
    LNH_LAUNCH(k_color_forward, dim3(tiles < 2048 ? tiles : 2048), dim3(256), 0, (hipStream_t)stream, a);
    return lnh_check_launch("lnh_lidar_color_forward");
}

int LNH_MLP_FN(lnh_lidar_color_composite_forward)(const float *z, const float *sigma_pt, const int32_t *perm,
                                                  const float *sample_dist, const void *h16, const float *cdir,
                                                  const void *w16, uint32_t N, uint32_t T, float density_scale,
                                                  float *sigma_m, float *weights, float *rgb, float *weights_sum,
                                                  float *depth, float *image, lnh_stream_t stream) {
    LNH_REQUIRE(z && sigma_pt && perm && sample_dist && h16 && cdir && w16 && sigma_m && weights && rgb && weights_sum &&
                    depth && image, LNH_ERR_INVALID_ARG, "lidar_color_composite_forward: null pointer");
    if (N == 0 || T == 0) return LNH_OK;
    LNH_REQUIRE((uint64_t)N * T < 0xffffffffull, LNH_ERR_UNSUPPORTED, "lidar_color_composite_forward: N*T must fit 32 bits");
    const size_t lds = (size_t)4 * 2 * T * sizeof(float);  // weights + slots of the four rays of a workgroup
    LNH_REQUIRE(lds <= 64 * 1024, LNH_ERR_UNSUPPORTED,
                "lidar_color_composite_forward: T=%u needs %zu B of LDS (> 64 KiB); use the three separate entry points", T, lds);
    ColorArgs a{};
    a.h16 = (const half_t *)h16; a.perm = perm; a.cdir = cdir; a.W = (const half_t *)w16; a.rgb = rgb; a.N = N; a.T = T;
    a.z = z; a.sigma_pt = sigma_pt; a.sample_dist = sample_dist; a.density_scale = density_scale;
    a.sigma_m = sigma_m; a.weights_out = weights; a.wsum = weights_sum; a.depth = depth; a.image = image;
    const uint32_t wgs = (N + 3) / 4;
    LNH_LAUNCH(k_color_forward_ray, dim3(wgs < 4096 ? wgs : 4096), dim3(256), lds, (hipStream_t)stream, a);
    return lnh_check_launch("lnh_lidar_color_composite_forward");
}

// second launch of the colour backward: the workgroups' partials [W2: 4 tiles | W1: 16 tiles | W0g: 4 tiles] added up in
// index order and added to grad_w (wgrad.h)
static void color_wgrad_reduce(const ColorArgs &a, uint32_t nwg, hipStream_t s) {
    const WgradTileMap map{{{a.dW + kW2, 0, 4, 4, 64}, {a.dW + kW1, 4, 4, 4, 64}, {a.dW + kW0g, 20, 1, 1, 16}}};
    wgrad_reduce_launch(a.ws, nwg, 24 * 256, map, s);
}

static int color_backward_launch(const float *grad_rgb, const float *grad_image, const float *grad_sigma, const void *h16, const int32_t *perm,
                             const float *weights, const float *cdir, const void *w16, uint32_t N, uint32_t T,
                             void *grad_h16, float *grad_w, float *ray_sum, void *wgrad_ws, uint64_t wgrad_ws_bytes,
                             lnh_stream_t stream) {
    LNH_REQUIRE((grad_rgb || grad_image) && grad_sigma && h16 && perm && weights && cdir && w16 && grad_h16 && grad_w && ray_sum,
                LNH_ERR_INVALID_ARG, "lidar_color_backward: null pointer");
    LNH_REQUIRE(((uintptr_t)grad_w & 15) == 0, LNH_ERR_INVALID_ARG, "lidar_color_backward: grad_w must be 16-byte aligned");
    if (N == 0 || T == 0) return LNH_OK;
    LNH_REQUIRE((uint64_t)N * T < 0xffffffffull, LNH_ERR_UNSUPPORTED, "lidar_color_backward: N*T must fit 32 bits");
    ColorArgs a{};
    a.h16 = (const half_t *)h16; a.perm = perm; a.weights = weights; a.cdir = cdir; a.W = (const half_t *)w16;
    a.g_rgb = grad_rgb; a.g_image = grad_image; a.g_sigma = grad_sigma; a.g_h16 = (half_t *)grad_h16; a.dW = grad_w; a.S = ray_sum;
    a.N = N; a.T = T;
    if (int rc = wgrad_ws_open(wgrad_ws, wgrad_ws_bytes, a.ws, "lidar_color_backward")) return rc;
    // persistent workgroups (one per CU): each leaves 6144 weight-gradient partials behind
    const uint32_t nwg = (N + 3) / 4 < 256 ? (N + 3) / 4 : 256;
    if (grad_rgb) {
        LNH_LAUNCH((k_color_backward_wi<false>), dim3(nwg), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        LNH_LAUNCH((k_color_backward_wi<true>), dim3(nwg), dim3(256), 0, (hipStream_t)stream, a);
    }
    color_wgrad_reduce(a, nwg, (hipStream_t)stream);
    return lnh_check_launch("lnh_lidar_color_backward");
}

int LNH_MLP_FN(lnh_lidar_color_backward)(const float *grad_rgb, const float *grad_sigma, const void *h16, const int32_t *perm,
                             const float *weights, const float *cdir, const void *w16, uint32_t N, uint32_t T,
                             void *grad_h16, float *grad_w, float *ray_sum, void *wgrad_ws, uint64_t wgrad_ws_bytes,
                             lnh_stream_t stream) {
    LNH_REQUIRE(grad_rgb, LNH_ERR_INVALID_ARG, "lidar_color_backward: null pointer");
    return color_backward_launch(grad_rgb, nullptr, grad_sigma, h16, perm, weights, cdir, w16, N, T, grad_h16, grad_w, ray_sum,
                                 wgrad_ws, wgrad_ws_bytes, stream);
}

int LNH_MLP_FN(lnh_lidar_color_backward_image)(const float *grad_image, const float *grad_sigma, const void *h16,
                                               const int32_t *perm, const float *weights, const float *cdir,
                                               const void *w16, uint32_t N, uint32_t T, void *grad_h16, float *grad_w,
                                               float *ray_sum, void *wgrad_ws, uint64_t wgrad_ws_bytes,
                                               lnh_stream_t stream) {
    LNH_REQUIRE(grad_image, LNH_ERR_INVALID_ARG, "lidar_color_backward_image: null pointer");
    return color_backward_launch(nullptr, grad_image, grad_sigma, h16, perm, weights, cdir, w16, N, T, grad_h16, grad_w,
                                 ray_sum, wgrad_ws, wgrad_ws_bytes, stream);
}


int LNH_MLP_FN(lnh_ragged_color_forward)(const void *h16, const int32_t *rays, const float *cdir, const void *w16, uint32_t N,
                                        uint32_t M, float *rgb, lnh_stream_t stream) {
    LNH_REQUIRE(h16 && rays && cdir && w16 && rgb, LNH_ERR_INVALID_ARG, "ragged_color_forward: null pointer");
    if (N == 0 || M == 0) return LNH_OK;
    ColorArgs a{};
    a.h16 = (const half_t *)h16; a.rays = rays; a.cdir = cdir; a.W = (const half_t *)w16; a.rgb = rgb; a.N = N; a.M = M;
    const uint32_t wgs = (N + 3) / 4;
    LNH_LAUNCH(k_color_forward_ragged, dim3(wgs < 4096 ? wgs : 4096), dim3(256), 0, (hipStream_t)stream, a);
    return lnh_check_launch("lnh_ragged_color_forward");
}

int LNH_MLP_FN(lnh_ragged_color_backward)(const float *grad_rgb, const float *grad_sigma, float density_scale, const void *h16,
                                         const int32_t *rays, const float *cdir, const void *w16, uint32_t N, uint32_t M,
                                         void *grad_h16, float *grad_w, float *ray_sum, void *wgrad_ws,
                                         uint64_t wgrad_ws_bytes, lnh_stream_t stream) {
    LNH_REQUIRE(grad_rgb && grad_sigma && h16 && rays && cdir && w16 && grad_h16 && grad_w && ray_sum, LNH_ERR_INVALID_ARG,
                "ragged_color_backward: null pointer");
    LNH_REQUIRE(((uintptr_t)grad_w & 15) == 0, LNH_ERR_INVALID_ARG, "ragged_color_backward: grad_w must be 16-byte aligned");
    if (N == 0 || M == 0) return LNH_OK;
    ColorArgs a{};
    a.h16 = (const half_t *)h16; a.rays = rays; a.cdir = cdir; a.W = (const half_t *)w16; a.g_rgb = grad_rgb;
    a.g_sigma = grad_sigma; a.g_h16 = (half_t *)grad_h16; a.dW = grad_w; a.S = ray_sum; a.N = N; a.M = M; a.T = 0;
    a.gs_scale = density_scale;
    if (int rc = wgrad_ws_open(wgrad_ws, wgrad_ws_bytes, a.ws, "ragged_color_backward")) return rc;
    const uint32_t nwg = (N + 3) / 4 < 256 ? (N + 3) / 4 : 256;
    LNH_LAUNCH((k_color_backward_wi<false, true>), dim3(nwg), dim3(256), 0, (hipStream_t)stream, a);
    color_wgrad_reduce(a, nwg, (hipStream_t)stream);
    return lnh_check_launch("lnh_ragged_color_backward");
}

}  // extern "C"

}  // namespace LNH_MLP_NS
