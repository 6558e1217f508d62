// lidar_color.hip — LiDAR colour head (ray-drop, intensity) of the fused render step on gfx950 MFMA: forward and
// backward over the merged samples of a ray (renderer.py:249-256, network.py:199-237 of the reference as ONE kernel each
// way).  See lidar_field.hip for the two algebraic moves (per-ray direction term, sigma-net row as the input) and
// mlp_common.h for the register-resident layer chaining and the fp16 / bf16 element-type builds
// (lidar_color_bf16.hip compiles this file again with bf16 MFMA operands: entry points lnh_lidar_color_*_bf16).
#include "mlp_common.h"

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
    const float *g_rgb;     // bwd in  [N,T,2]
    const float *g_sigma;   // bwd in  [N,T] merged order
    half_t *g_h16;          // bwd out [N*T,16] point order
    float *dW;              // bwd out flat fp32 (kWTotal), atomically accumulated
    float *S;               // bwd out [N,64] fp32: sum over the ray of d(hidden0 pre-activation)
    uint32_t N, T;
};

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

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
    for (uint32_t base = wave * NT * 16; base < total; base += nwaves * NT * 16) {
        float wgt[NT];
        bool msk[NT];
        bool any = false;
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const uint32_t m = base + n * 16 + c;
            const float wv = a.weights[m < total ? m : 0];  // unconditional load + select
            wgt[n] = m < total ? wv : 0.0f;
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
            continue;
        }
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const uint32_t m = base + n * 16 + c;
            const bool valid = m < total;
            const uint32_t mc = valid ? m : 0, ray = mc / a.T;
            const size_t src = (size_t)ray * a.T + (uint32_t)a.perm[mc];
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
}

// Wave-independent colour-head backward: one WAVE per ray at a time, 32 merged samples per iteration, no LDS tiles and
// no barriers in the sample loop.  Weight gradients contract over samples, so their MFMA operands are the TRANSPOSES of
// what the layer chain leaves in registers; a transpose of a packed fp16 fragment is one MFMA against an identity
// fragment (the A and B operand layouts are mirror images: row/col = lane & 15, same k enumeration), which is exact.
// Every wave owns all 24 gradient tiles (dW2 4, dW1 16, dW0g 4) in accumulators; the four waves of a workgroup are
// combined through LDS once at the end and flushed with one atomic per weight.
__global__ void __launch_bounds__(256)
k_color_backward_wi(ColorArgs a) {
    constexpr int HT = 4, HS = 2, NT = 2, NTILE = HT + HT * HT + HT;
    // Weight fragments live in LDS in per-lane order (one conflict-free ds_read_b128 each) instead of 112 VGPRs: the
    // 24 gradient tiles (96 accumulators) plus the pipeline state already fill the register file of a wave.
    enum { F_W0 = 0, F_W1 = 4, F_W2 = 12, F_W2T = 14, F_W1T = 18, F_W0T = 26, NFRAG = 28 };
    __shared__ half8_t wfrag[NFRAG][64];
    __shared__ float red[NTILE * 256];
    const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const uint32_t nw = blockDim.x >> 6, wave = blockIdx.x * nw + wid, nwaves = gridDim.x * nw;
    if (wid == 0) {
#pragma unroll
        for (int t = 0; t < HT; t++) {
            wfrag[F_W0 + t][lane] = load_a_natural(a.W + kW0g, 16, 16 * t + c, 0, g, 16);
            wfrag[F_W2T + t][lane] = load_at_natural(a.W + kW2, 64, 16 * t + c, 0, g, 16);
#pragma unroll
            for (int s = 0; s < HS; s++) {
                wfrag[F_W1 + 2 * t + s][lane] = load_a_nu(a.W + kW1, 64, 16 * t + c, s, g);
                wfrag[F_W1T + 2 * t + s][lane] = load_at_nu(a.W + kW1, 64, 16 * t + c, s, g);
            }
        }
#pragma unroll
        for (int s = 0; s < HS; s++) {
            wfrag[F_W2 + s][lane] = load_a_nu(a.W + kW2, 64, c, s, g);
            wfrag[F_W0T + s][lane] = load_at_nu(a.W + kW0g, 16, c, s, g);
        }
    }
    __syncthreads();
#define WF(i) (wfrag[(i)][lane])
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

    // The per-sample operands form a dependent chain (perm -> sigma-net row) of HBM round trips; left in program order
    // they cost ~4 exposed latencies per step.  The (ray, step) iteration space of the wave is therefore flattened and
    // software-pipelined: stage A (weights, perm, incoming gradients) runs two iterations ahead, stage B (the gathered
    // sigma-net row) one iteration ahead.  All loads are unconditional from clamped addresses.
    struct StageA {
        bool valid[NT];
        uint32_t ray, m[NT], slot[NT];
        float wgt[NT], gs[NT];
        float2 gr[NT];
    };
    const uint32_t nsteps = (a.T + 31) / 32;
    const uint32_t nrays = wave < a.N ? (a.N - wave + nwaves - 1) / nwaves : 0;
    const uint32_t K = nrays * nsteps;
    auto load_a = [&](uint32_t k) {
        StageA A;
        const uint32_t rr = k / nsteps, s0 = (k - rr * nsteps) * 32;
        A.ray = wave + rr * nwaves;
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const uint32_t i = s0 + 16 * n + c;
            A.valid[n] = k < K && i < a.T;
            A.m[n] = A.valid[n] ? A.ray * a.T + i : 0;  // N*T < 2^32 (checked by the launcher)
            A.wgt[n] = a.weights[A.m[n]];
            A.slot[n] = (uint32_t)a.perm[A.m[n]];
            A.gs[n] = a.g_sigma[A.m[n]];
            A.gr[n] = *reinterpret_cast<const float2 *>(a.g_rgb + (size_t)A.m[n] * 2);
        }
        return A;
    };
    auto src_of = [&](const StageA &A, int n) { return A.valid[n] ? (size_t)A.ray * a.T + A.slot[n] : (size_t)0; };
    struct StageB { half8_t x[NT]; };
    auto load_b = [&](const StageA &A) {
        StageB B;
#pragma unroll
        for (int n = 0; n < NT; n++)
            B.x[n] = *reinterpret_cast<const half8_t *>(a.h16 + src_of(A, n) * 16 + (g < 2 ? 8 * g : 0));
        return B;
    };
    auto load_cb = [&](uint32_t ray, f32x4 (&cb)[HT]) {
        const uint32_t r = ray < a.N ? ray : 0;
#pragma unroll
        for (int t = 0; t < HT; t++) cb[t] = *reinterpret_cast<const f32x4 *>(a.cdir + (size_t)r * 64 + 16 * t + 4 * g);
    };
    StageA A0 = load_a(0), A1 = load_a(1);
    StageB B0 = load_b(A0);
    f32x4 cb[HT], cb_next[HT];
    load_cb(wave, cb_next);
    float ssum[HT];

    for (uint32_t k = 0; k < K; k++) {
        __asm__ volatile("" ::: "memory");  // keep the weight-fragment LDS reads inside the loop (no hoist + spill)
        const uint32_t rr = k / nsteps, sidx = k - rr * nsteps;
        const uint32_t ray = wave + rr * nwaves;
        const StageA A2 = load_a(k + 2);
        const StageB B1 = load_b(A1);
        if (sidx == 0) {
#pragma unroll
            for (int t = 0; t < HT; t++) {
                cb[t] = cb_next[t];
                ssum[t] = 0.0f;
            }
            load_cb(ray + nwaves, cb_next);
        }
        bool msk[NT];
        half8_t bx[NT];
        bool any = false;
#pragma unroll
        for (int n = 0; n < NT; n++) {
            msk[n] = A0.valid[n] && A0.wgt[n] > kMaskThresh;
            any |= msk[n];
            bx[n] = (A0.valid[n] && g < 2) ? B0.x[n] : zero_h8();
        }
        if (!__any(any)) {
            // the wave's 32 samples are transparent: colour is defined as 0 there, only the compositing gradient of
            // sigma flows back
#pragma unroll
            for (int n = 0; n < NT; n++)
                if (A0.valid[n]) {
                    half4_t v = {(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
                    if (g == 0) v[0] = (half_t)(A0.gs[n] * expf(fminf(fmaxf((float)bx[n][0], -15.0f), 15.0f)));
                    *reinterpret_cast<half4_t *>(a.g_h16 + src_of(A0, n) * 16 + 4 * g) = v;
                }
        } else {
            half8_t by[NT], bh0[NT][HS], bh1[NT][HS], bd1[NT][HS], bd0[NT][HS];
#pragma unroll
            for (int n = 0; n < NT; n++) {
                f32x4 acc[HT];
#pragma unroll
                for (int t = 0; t < HT; t++) acc[t] = MFMA16(WF(F_W0 + t), bx[n], cb[t]);
#pragma unroll
                for (int s = 0; s < HS; s++)
                    bh0[n][s] = pack_pair_relu(acc[2 * s], acc[2 * s + 1]);
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    acc[t] = zero_f4();
#pragma unroll
                    for (int s = 0; s < HS; s++) acc[t] = MFMA16(WF(F_W1 + 2 * t + s), bh0[n][s], acc[t]);
                }
#pragma unroll
                for (int s = 0; s < HS; s++)
                    bh1[n][s] = pack_pair_relu(acc[2 * s], acc[2 * s + 1]);
                f32x4 o = zero_f4();
#pragma unroll
                for (int s = 0; s < HS; s++) o = MFMA16(WF(F_W2 + s), bh1[n][s], o);
                // output gradient through the sigmoid (only outputs 0,1 exist; lanes g == 0 hold them)
                by[n] = zero_h8();
                if (g == 0 && msk[n]) {
                    const float r0 = sigmoidf((float)(half_t)o[0]), r1 = sigmoidf((float)(half_t)o[1]);
                    by[n][0] = (half_t)(A0.gr[n].x * r0 * (1.0f - r0));
                    by[n][1] = (half_t)(A0.gr[n].y * r1 * (1.0f - r1));
                }
                f32x4 d[HT];
#pragma unroll
                for (int t = 0; t < HT; t++) d[t] = MFMA16(WF(F_W2T + t), by[n], zero_f4());
#pragma unroll
                for (int s = 0; s < HS; s++) bd1[n][s] = pack_pair_relu_bwd(d[2 * s], d[2 * s + 1], bh1[n][s]);
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    d[t] = zero_f4();
#pragma unroll
                    for (int s = 0; s < HS; s++) d[t] = MFMA16(WF(F_W1T + 2 * t + s), bd1[n][s], d[t]);
                }
#pragma unroll
                for (int s = 0; s < HS; s++) bd0[n][s] = pack_pair_relu_bwd(d[2 * s], d[2 * s + 1], bh0[n][s]);
                // d(sigma-net row) = W0g^T dH0, col 0 <- trunc_exp backward of the compositing gradient
                f32x4 dx = zero_f4();
#pragma unroll
                for (int s = 0; s < HS; s++) dx = MFMA16(WF(F_W0T + s), bd0[n][s], dx);
                if (A0.valid[n]) {
                    if (g == 0) dx[0] = A0.gs[n] * expf(fminf(fmaxf((float)bx[n][0], -15.0f), 15.0f));
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
                gW2[t] = MFMA16(fy, fh1, gW2[t]);  // dW2[o = 4g + r][16t + c]
                const f32x4 e0 = MFMA16(bd0[0][t >> 1], idv[t & 1], zero_f4());
                const f32x4 e1 = MFMA16(bd0[1][t >> 1], idv[t & 1], zero_f4());
                ssum[t] += (e0[0] + e0[1]) + (e0[2] + e0[3]) + (e1[0] + e1[1]) + (e1[2] + e1[3]);
                gW0[t] = MFMA16(pack2(e0, e1), fx, gW0[t]);  // dW0g[16t + 4g + r][c]
#pragma unroll
                for (int i = 0; i < HT; i++) gW1[t][i] = MFMA16(fd1[t], fh0[i], gW1[t][i]);  // dW1[16t + 4g + r][16i + c]
            }
        }
        A0 = A1;
        A1 = A2;
        B0 = B1;
        if (sidx + 1 < nsteps) continue;
        // ---- S[ray][channel 16t + c] = sum over the ray's samples of dH0 (lane (g,c) holds samples 4g + r)
#pragma unroll
        for (int t = 0; t < HT; t++) {
            float v = ssum[t];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g == 0) a.S[(size_t)ray * 64 + 16 * t + c] = v;
        }
    }

    // ---- combine the waves of the workgroup through LDS, then one atomic per weight
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
    for (uint32_t e = threadIdx.x; e < NTILE * 256; e += blockDim.x) {
        const uint32_t tile = e >> 8, r = (e >> 6) & 3, ln = e & 63, gg = ln >> 4, cc = ln & 15;
        const float v = red[e];
        if (tile < HT) {
            unsafeAtomicAdd(a.dW + kW2 + (size_t)(4 * gg + r) * 64 + 16 * tile + cc, v);
        } else if (tile < HT + HT * HT) {
            const uint32_t t = (tile - HT) / HT, i = (tile - HT) % HT;
            unsafeAtomicAdd(a.dW + kW1 + (size_t)(16 * t + 4 * gg + r) * 64 + 16 * i + cc, v);
        } else {
            const uint32_t t = tile - HT - HT * HT;
            unsafeAtomicAdd(a.dW + kW0g + (size_t)(16 * t + 4 * gg + r) * 16 + cc, v);
        }
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

int LNH_MLP_FN(lnh_lidar_color_backward)(const float *grad_rgb, const float *grad_sigma, const void *h16, const int32_t *perm,
                             const float *weights, const float *cdir, const void *w16, uint32_t N, uint32_t T,
                             void *grad_h16, float *grad_w, float *ray_sum, lnh_stream_t stream) {
    LNH_REQUIRE(grad_rgb && grad_sigma && h16 && perm && weights && cdir && w16 && grad_h16 && grad_w && ray_sum,
                LNH_ERR_INVALID_ARG, "lidar_color_backward: null pointer");
    if (N == 0 || T == 0) return LNH_OK;
    LNH_REQUIRE((uint64_t)N * T < 0xffffffffull, LNH_ERR_UNSUPPORTED, "lidar_color_backward: N*T must fit 32 bits");
    ColorArgs a{};
    a.h16 = (const half_t *)h16; a.perm = perm; a.weights = weights; a.cdir = cdir; a.W = (const half_t *)w16;
    a.g_rgb = grad_rgb; a.g_sigma = grad_sigma; a.g_h16 = (half_t *)grad_h16; a.dW = grad_w; a.S = ray_sum;
    a.N = N; a.T = T;
    // persistent workgroups: each flushes 6144 weight-gradient partials with device atomics (~20 G/s chip-wide), so
    // keep the workgroup count near the CU count rather than one per ray
    const uint32_t wgs = (N + 3) / 4;
    LNH_LAUNCH(k_color_backward_wi, dim3(wgs < 256 ? wgs : 256), dim3(256), 0, (hipStream_t)stream, a);
    return lnh_check_launch("lnh_lidar_color_backward");
}

}  // extern "C"

}  // namespace LNH_MLP_NS
