// mlp.hip — fully fused tiny MLP (forward / backward) on gfx950 MFMA.  See mlp_common.h for the register-resident
// layer chaining.  Semantics: lidarnerf/ffmlp/src/ffmlp.cu (forward 460-576, backward 578-733, weight gradients
// 1107-1263), lidarnerf/ffmlp/ffmlp.py:187-283, and the bias-free Linear stacks of lidarnerf/nerf/network.py:45-99.
//
// Forward: persistent waves; each wave keeps ALL weight fragments in VGPRs and streams point tiles (NT x 16 points
// per iteration): one 16-byte load per lane per k-step in, one 8-byte store per lane out.
// Backward: the same wave recomputes the hidden activations (cheaper than reading a saved [layers,B,hidden] buffer
// back from HBM), back-propagates in registers with transposed weight fragments, and accumulates the weight
// gradients dW_l = dH_l^T * A_{l-1} on MFMA with the contraction over points; the two operands of that product are
// transposed through a small per-wave LDS tile.  Per-wave partial dW are added to the fp32 gradient vector with
// one atomic per element per wave at the end of the kernel.
#include "mlp_bwd.h"

namespace LNH_MLP_NS {

int lnh_mlp_backward_h32(uint32_t in_ks, uint32_t nhm, const MlpBwdArgs &a, hipStream_t s);
int lnh_mlp_backward_nhm0(uint32_t in_ks, const MlpBwdArgs &a, hipStream_t s);
int lnh_mlp_backward_nhm1(uint32_t in_ks, const MlpBwdArgs &a, hipStream_t s);
int lnh_mlp_backward_nhm2(uint32_t in_ks, const MlpBwdArgs &a, hipStream_t s);
int lnh_density_mlp_backward_launch(const MlpBwdArgs &a, hipStream_t s);
// mlp_wide.hip: hidden 128 / 256 and nets with more than two hidden->hidden matrices (weights loaded where they are used)
int lnh_mlp_forward_wide(const void *inputs, const void *weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                         uint32_t hidden_dim, uint32_t n_hidden_mats, uint32_t activation, uint32_t output_activation,
                         void *forward_buffer, void *outputs, hipStream_t s);

namespace {

struct MlpArgs {
    const void *X;        // [B, in_dim] (IO::in_t: MLP element type, or level-major fp16 features for DensityIO)
    const half_t *W;      // flat weights
    half_t *Y;            // [B, 16]
    half_t *fb;           // NULL or [NHM+1, B, hidden]
    uint32_t B, in_dim, hidden, act, out_act;
    float *sigma;         // DensityIO only: fp32 exp(out[0]) per destination row
    IoDims io;
};

// ---------------------------------------------------------------------------------------------------- forward
// IN_KS = ceil(in_dim / 32); HT = hidden / 16 (M-tiles); NHM = hidden->hidden matrices; NT = point tiles / iteration
// FAST = (hidden activation ReLU, no output activation), resolved at compile time; otherwise runtime switches.
template <int IN_KS, int HT, int NHM, int NT, typename IO, bool FAST>
__global__ void __launch_bounds__(256)
k_mlp_forward(MlpArgs a) {
    constexpr int ACT = FAST ? (int)LNH_ACT_RELU : -1;
    constexpr int HS = HT / 2;  // k-steps over a hidden vector
    const uint32_t lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t hidden = HT * 16;

    // ---- weights -> registers (once per wave)
    const half_t *W0 = a.W;
    const half_t *Wh = W0 + (size_t)hidden * a.in_dim;
    const half_t *Wo = Wh + (size_t)NHM * hidden * hidden;
    half8_t w0[HT][IN_KS];
#pragma unroll
    for (int t = 0; t < HT; t++)
#pragma unroll
        for (int s = 0; s < IN_KS; s++) w0[t][s] = load_a_natural(W0, a.in_dim, 16 * t + c, s, g, a.in_dim);
    half8_t wh[NHM > 0 ? NHM : 1][HT][HS];
#pragma unroll
    for (int m = 0; m < NHM; m++)
#pragma unroll
        for (int t = 0; t < HT; t++)
#pragma unroll
            for (int s = 0; s < HS; s++) wh[m][t][s] = load_a_nu(Wh + (size_t)m * hidden * hidden, hidden, 16 * t + c, s, g);
    half8_t wo[HS];
#pragma unroll
    for (int s = 0; s < HS; s++) wo[s] = load_a_nu(Wo, hidden, c, s, g);

    const uint32_t act = a.act, out_act = a.out_act;
    // ---- inputs: B operand, natural k enumeration.  One-k-step nets (the sigma net) request the NEXT tile's rows before
    // the current tile is processed: the kernel is a read -> 3 layers -> write chain per tile at HBM latency otherwise
    auto load_in = [&](uint64_t base, half8_t (&bx)[NT][IN_KS]) {
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const uint64_t p = base + n * 16 + c;
#pragma unroll
            for (int s = 0; s < IN_KS; s++) {
                const uint32_t k0 = 32 * s + 8 * g;
                // unconditional load from a clamped address + register select (a predicated load would make hipcc
                // branch around every load and drain vmcnt per element)
                const bool ok = p < a.B && k0 < a.in_dim;
                const half8_t v = IO::load_x((const typename IO::in_t *)a.X, ok ? p : 0, ok ? k0 : 0, a.B, a.in_dim, a.io);
                bx[n][s] = ok ? v : zero_h8();
            }
        }
    };
    constexpr bool PREFETCH = IN_KS == 1 && IO::kDensity;  // (measured on the sigma net; the generic instantiations keep their registers)
    const uint64_t stride = (uint64_t)nwaves * NT * 16;
    half8_t bx[NT][IN_KS], bx_next[NT][IN_KS];
    if (PREFETCH) load_in((uint64_t)wave * NT * 16, bx);
    for (uint64_t base = (uint64_t)wave * NT * 16; base < a.B; base += stride) {
        if (PREFETCH) load_in(base + stride < a.B ? base + stride : base, bx_next);
        else load_in(base, bx);
        // ---- layer 0
        half8_t bh[NT][HS];
        {
            f32x4 acc[HT][NT];
#pragma unroll
            for (int t = 0; t < HT; t++)
#pragma unroll
                for (int n = 0; n < NT; n++) {
                    acc[t][n] = zero_f4();
#pragma unroll
                    for (int s = 0; s < IN_KS; s++) acc[t][n] = MFMA16(w0[t][s], bx[n][s], acc[t][n]);
                }
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int s = 0; s < HS; s++)
                    bh[n][s] = pack_pair_act<ACT>(acc[2 * s][n], acc[2 * s + 1][n], act);
        }
        auto save_hidden = [&](int layer) {
            if (!a.fb) return;
#pragma unroll
            for (int n = 0; n < NT; n++) {
                const uint64_t p = base + n * 16 + c;
                if (p >= a.B) continue;
                half_t *row = a.fb + ((size_t)layer * a.B + p) * hidden;
#pragma unroll
                for (int s = 0; s < HS; s++) {
                    half4_t lo = {bh[n][s][0], bh[n][s][1], bh[n][s][2], bh[n][s][3]};
                    half4_t hi = {bh[n][s][4], bh[n][s][5], bh[n][s][6], bh[n][s][7]};
                    *reinterpret_cast<half4_t *>(row + 32 * s + 4 * g) = lo;
                    *reinterpret_cast<half4_t *>(row + 32 * s + 16 + 4 * g) = hi;
                }
            }
        };
        save_hidden(0);
        // ---- hidden -> hidden
#pragma unroll
        for (int m = 0; m < NHM; m++) {
            f32x4 acc[HT][NT];
#pragma unroll
            for (int t = 0; t < HT; t++)
#pragma unroll
                for (int n = 0; n < NT; n++) {
                    acc[t][n] = zero_f4();
#pragma unroll
                    for (int s = 0; s < HS; s++) acc[t][n] = MFMA16(wh[m][t][s], bh[n][s], acc[t][n]);
                }
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int s = 0; s < HS; s++)
                    bh[n][s] = pack_pair_act<ACT>(acc[2 * s][n], acc[2 * s + 1][n], act);
            save_hidden(m + 1);
        }
        // ---- output layer (16 padded outputs)
#pragma unroll
        for (int n = 0; n < NT; n++) {
            f32x4 o = zero_f4();
#pragma unroll
            for (int s = 0; s < HS; s++) o = MFMA16(wo[s], bh[n][s], o);
            const uint64_t p = base + n * 16 + c;
            if (p < a.B) {
                if (!FAST && out_act != LNH_ACT_NONE) {
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = act_forward(out_act, o[r]);
                }
                half4_t v = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
                const uint64_t row = IO::out_row(p, a.io);
                *reinterpret_cast<half4_t *>(a.Y + row * 16 + 4 * g) = v;
                if constexpr (IO::kDensity) {
                    if (g == 0) a.sigma[row] = expf((float)v[0]);
                }
            }
        }
        if (PREFETCH) {
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int s = 0; s < IN_KS; s++) bx[n][s] = bx_next[n][s];
        }
    }
}

template <int IN_KS, int HT, int NHM, typename IO = RowMajorIO, int NT = 4>
int launch_fwd(const MlpArgs &a, hipStream_t s) {
    const uint32_t tiles = div_up(a.B, NT * 16 * 4);
    const uint32_t grid = tiles < 2048 ? tiles : 2048;
    if (a.act == LNH_ACT_RELU && a.out_act == LNH_ACT_NONE) {
        LNH_LAUNCH((k_mlp_forward<IN_KS, HT, NHM, NT, IO, true>), dim3(grid), dim3(256), 0, s, a);
    } else {
        LNH_LAUNCH((k_mlp_forward<IN_KS, HT, NHM, NT, IO, false>), dim3(grid), dim3(256), 0, s, a);
    }
    return lnh_check_launch("lnh_mlp_forward");
}

int check_shape(uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t n_hidden_mats) {
    LNH_REQUIRE(input_dim > 0 && input_dim % 16 == 0, LNH_ERR_UNSUPPORTED,
                "FFMLP input_dim should be 16 * m (m > 0), but got %u", input_dim);
    LNH_REQUIRE(input_dim <= 128, LNH_ERR_UNSUPPORTED, "fused MLP: input_dim <= 128 in this build (got %u)", input_dim);
    LNH_REQUIRE(output_dim == 16, LNH_ERR_UNSUPPORTED,
                "FFMLP current only supports (padded) output dim == 16, but got %u", output_dim);
    LNH_REQUIRE(hidden_dim == 64 || hidden_dim == 32, LNH_ERR_UNSUPPORTED,
                "fused MLP: hidden_dim must be 32, 64, 128 or 256 (16 runs zero-padded on 32), got %u", hidden_dim);
    LNH_REQUIRE(n_hidden_mats <= 2, LNH_ERR_UNSUPPORTED,
                "fused MLP: at most 2 hidden->hidden matrices in this build (got %u)", n_hidden_mats);
    return LNH_OK;
}

#define LNH_MLP_FWD_CASES(HT, ARGS)                                                                      \
        switch (key) {                                                                                    \
            case 10: rc = launch_fwd<1, HT, 0>(ARGS, s); break;                                           \
            case 11: rc = launch_fwd<1, HT, 1>(ARGS, s); break;                                           \
            case 12: rc = launch_fwd<1, HT, 2>(ARGS, s); break;                                           \
            case 20: rc = launch_fwd<2, HT, 0>(ARGS, s); break;                                           \
            case 21: rc = launch_fwd<2, HT, 1>(ARGS, s); break;                                           \
            case 22: rc = launch_fwd<2, HT, 2>(ARGS, s); break;                                           \
            case 30: rc = launch_fwd<3, HT, 0>(ARGS, s); break;                                           \
            case 31: rc = launch_fwd<3, HT, 1>(ARGS, s); break;                                           \
            case 32: rc = launch_fwd<3, HT, 2>(ARGS, s); break;                                           \
            case 40: rc = launch_fwd<4, HT, 0>(ARGS, s); break;                                           \
            case 41: rc = launch_fwd<4, HT, 1>(ARGS, s); break;                                           \
            case 42: rc = launch_fwd<4, HT, 2>(ARGS, s); break;                                           \
            default:                                                                                      \
                lnh_set_error("fused MLP: no kernel instance for input_dim=%u hidden=%u hidden_mats=%u",  \
                              input_dim, hidden_dim, n_hidden_mats);                                      \
                rc = LNH_ERR_UNSUPPORTED;                                                                 \
        }
#define LNH_MLP_FWD_DISPATCH(ARGS)                                                                       \
    {                                                                                                     \
        const uint32_t iks = (input_dim + 31) / 32;                                                       \
        const uint32_t key = iks * 10 + n_hidden_mats;                                                    \
        if (hidden_dim == 32) { LNH_MLP_FWD_CASES(2, ARGS) } else { LNH_MLP_FWD_CASES(4, ARGS) }          \
    }

}  // namespace

extern "C" {

int LNH_MLP_FN(lnh_mlp_forward)(const void *inputs, const void *weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                    uint32_t hidden_dim, uint32_t n_hidden_mats, uint32_t activation, uint32_t output_activation,
                    void *forward_buffer, void *outputs, lnh_stream_t stream) {
    LNH_REQUIRE(inputs && weights && outputs, LNH_ERR_INVALID_ARG, "mlp forward: null pointer");
    LNH_REQUIRE(activation <= LNH_ACT_NONE && output_activation <= LNH_ACT_NONE, LNH_ERR_INVALID_ARG,
                "mlp forward: unknown activation");
    if (hidden_dim == 128 || hidden_dim == 256 || (n_hidden_mats > 2 && (hidden_dim == 32 || hidden_dim == 64))) {
        if (B == 0) return LNH_OK;
        return lnh_mlp_forward_wide(inputs, weights, B, input_dim, output_dim, hidden_dim, n_hidden_mats, activation,
                                    output_activation, forward_buffer, outputs, (hipStream_t)stream);
    }
    int rc = check_shape(input_dim, output_dim, hidden_dim, n_hidden_mats);
    if (rc) return rc;
    if (B == 0) return LNH_OK;
    MlpArgs a{inputs, (const half_t *)weights, (half_t *)outputs, (half_t *)forward_buffer,
              B, input_dim, hidden_dim, activation, output_activation, nullptr, IoDims{1, 1, 0, 0}};
    hipStream_t s = (hipStream_t)stream;
    LNH_MLP_FWD_DISPATCH(a)
    return rc;
}

int LNH_MLP_FN(lnh_mlp_backward)(const void *grad, const void *inputs, const void *weights, uint32_t B, uint32_t input_dim,
                     uint32_t output_dim, uint32_t hidden_dim, uint32_t n_hidden_mats, uint32_t activation,
                     uint32_t output_activation, void *grad_inputs, float *grad_weights, void *wgrad_ws,
                     uint64_t wgrad_ws_bytes, lnh_stream_t stream) {
    LNH_REQUIRE(grad && inputs && weights && grad_weights, LNH_ERR_INVALID_ARG, "mlp backward: null pointer");
    LNH_REQUIRE(((uintptr_t)grad_weights & 15) == 0, LNH_ERR_INVALID_ARG, "mlp backward: grad_weights must be 16-byte aligned");
    LNH_REQUIRE(activation <= LNH_ACT_NONE && activation != LNH_ACT_SINE, LNH_ERR_UNSUPPORTED,
                "mlp backward: Sine needs stored pre-activations (unsupported by the reference as well, utils.h:626-630)");
    LNH_REQUIRE(output_activation == LNH_ACT_NONE, LNH_ERR_UNSUPPORTED,
                "mlp backward: output activation is not supported (ffmlp.py:196 'not supported currently')");
    LNH_REQUIRE(!(hidden_dim == 128 || hidden_dim == 256 || (n_hidden_mats > 2 && (hidden_dim == 32 || hidden_dim == 64))),
                LNH_ERR_UNSUPPORTED,
                "mlp backward: hidden_dim %u with %u hidden matrices has no one-kernel backward (its weight-gradient tiles do "
                "not fit a wave): call lnh_mlp_backward_data for the activation / input gradients and form dW_l = G_l^T "
                "A_(l-1) from backward_buffer and forward_buffer with a GEMM, as the reference does (ffmlp.cu:1107-1263)",
                hidden_dim, n_hidden_mats);
    int rc = check_shape(input_dim, output_dim, hidden_dim, n_hidden_mats);
    if (rc) return rc;
    if (B == 0) return LNH_OK;
    MlpBwdArgs a{(const half_t *)grad, inputs, (const half_t *)weights, grad_inputs,
                 grad_weights, B, input_dim, hidden_dim, activation, output_activation, IoDims{1, 1, 0, 0}, WgradWs{}};
    rc = wgrad_ws_open(wgrad_ws, wgrad_ws_bytes, a.ws, "mlp backward");
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t iks = (input_dim + 31) / 32;
    if (hidden_dim == 32) return lnh_mlp_backward_h32(iks, n_hidden_mats, a, s);
    switch (n_hidden_mats) {
        case 0: rc = lnh_mlp_backward_nhm0(iks, a, s); break;
        case 1: rc = lnh_mlp_backward_nhm1(iks, a, s); break;
        default: rc = lnh_mlp_backward_nhm2(iks, a, s); break;
    }
    return rc;
}


int LNH_MLP_FN(lnh_density_mlp_forward)(const void *features, const void *weights, uint32_t B, uint32_t T_cur, uint32_t T_tot,
                            uint32_t slot_off, uint32_t feat_rows, void *h16, float *sigma, lnh_stream_t stream) {
    LNH_REQUIRE(features && weights && h16 && sigma, LNH_ERR_INVALID_ARG, "density mlp forward: null pointer");
    LNH_REQUIRE(T_cur >= 1 && slot_off + T_cur <= T_tot && B % T_cur == 0, LNH_ERR_INVALID_ARG,
                "density mlp forward: need B %% T_cur == 0 and slot_off + T_cur <= T_tot");
    if (B == 0) return LNH_OK;
    MlpArgs a{features, (const half_t *)weights, (half_t *)h16, nullptr, B, 32, 64, LNH_ACT_RELU,
              LNH_ACT_NONE, sigma, IoDims{T_cur, T_tot, slot_off, feat_rows}};
    return launch_fwd<1, 4, 0, DensityIO, 2>(a, (hipStream_t)stream);
}

int LNH_MLP_FN(lnh_density_mlp_backward)(const void *grad_h16, const void *features, const void *weights, uint32_t B,
                             uint32_t T_cur, uint32_t T_tot, uint32_t slot_off, void *grad_features,
                             float *grad_weights, void *wgrad_ws, uint64_t wgrad_ws_bytes, lnh_stream_t stream) {
    const uint32_t feat_rows = 0;
    LNH_REQUIRE(grad_h16 && features && weights && grad_features && grad_weights, LNH_ERR_INVALID_ARG,
                "density mlp backward: null pointer");
    LNH_REQUIRE(((uintptr_t)grad_weights & 15) == 0, LNH_ERR_INVALID_ARG,
                "density mlp backward: grad_weights must be 16-byte aligned");
    LNH_REQUIRE(T_cur >= 1 && slot_off + T_cur <= T_tot && B % T_cur == 0, LNH_ERR_INVALID_ARG,
                "density mlp backward: need B %% T_cur == 0 and slot_off + T_cur <= T_tot");
    if (B == 0) return LNH_OK;
    MlpBwdArgs a{(const half_t *)grad_h16, features, (const half_t *)weights, grad_features,
                 grad_weights, B, 32, 64, LNH_ACT_RELU, LNH_ACT_NONE, IoDims{T_cur, T_tot, slot_off, feat_rows}, WgradWs{}};
    if (int rc = wgrad_ws_open(wgrad_ws, wgrad_ws_bytes, a.ws, "density mlp backward")) return rc;
    return lnh_density_mlp_backward_launch(a, (hipStream_t)stream);
}

}  // extern "C"

}  // namespace LNH_MLP_NS
