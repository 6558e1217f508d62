// mlp_wide.hip — the fused tiny MLP for the shapes whose weights do not fit a wave's registers: hidden 128 / 256 (the
// reference's remaining widths, lidarnerf/ffmlp/src/ffmlp.cu:756-800) and nets with more than two hidden->hidden matrices
// at any width.  Same formulation as mlp.hip (activations as MFMA B operands, `pack_pair` turning two accumulator tiles
// into the next layer's B fragment without LDS), with ONE difference: a weight fragment is loaded where it is used
// instead of once per wave.  All waves of the chip read the same <= 128 KB per matrix, which the vector L1 / the XCD's L2
// serve; a 16-point tile of a 256-wide layer costs 128 fragment loads (two 8-byte loads per lane each) next to its 128
// MFMAs.  What a fragment load is shared by is the number of 16-point tiles a wave keeps in flight: two at hidden 256 (128
// accumulator registers), four at 128 and below — measured at 1 M points, inference: one / two tiles 2196 / 335 us, two /
// four 1084 / 182 us for hidden 256 / 128 (profiles/r04_ffmlp_wide.txt).
//
// Backward, as the reference structures its own (ffmlp.cu:578-733 + 1107-1263: a fused kernel for the activation
// gradients, split-K GEMMs for the weight gradients):
//   lnh_mlp_backward_data — ONE fused kernel: dL/d(pre-activation) of every hidden layer (written to `backward_buffer`,
//                           [n_hidden_mats + 1, B, hidden]) and dL/d(input), from the output gradient, the post-
//                           activations lnh_mlp_forward saved in `forward_buffer`, and the TRANSPOSED weight matrices
//                           (so that the fragments of W^T are the same contiguous loads as those of W in the forward);
//   the weight gradients dW_l = G_l^T A_(l-1) are plain [hidden, B] x [B, K] GEMMs over those two buffers: library work
//   (rocBLAS / hipBLASLt; ffmlp/ffmlp.py does them through torch.mm).  96 .. 1024 accumulator tiles per matrix do not fit a
//   wave, and a hand-written split-K GEMM would be a worse rocBLAS.
#include "mlp_common.h"

namespace LNH_MLP_NS {

namespace {

struct WideFwdArgs {
    const half_t *X, *W;
    half_t *Y, *fb;
    uint32_t B, in_dim, nhm, act, out_act;
};
struct WideBwdArgs {
    const half_t *gy, *fb, *WT;  // WT = [W0^T (in x H) | Wh_m^T (H x H each) | Wo^T (H x 16)]
    half_t *gb, *gx;             // gb [nhm + 1, B, H]; gx [B, in] or NULL
    uint32_t B, in_dim, nhm, act;
};

constexpr int kMaxInKs = 4;  // input_dim <= 128

// row `p` of a [*, H] buffer as the B fragments of a 16-point tile (the layout pack_pair produces / save_rows stores)
template <int HS>
__device__ __forceinline__ void load_rows(const half_t *__restrict__ buf, uint64_t p, uint32_t H, uint32_t g, bool ok,
                                          half8_t (&b)[HS]) {
    const half_t *row = buf + (ok ? p : 0) * H;
#pragma unroll
    for (int s = 0; s < HS; s++) {
        const half4_t lo = *reinterpret_cast<const half4_t *>(row + 32 * s + 4 * g);
        const half4_t hi = *reinterpret_cast<const half4_t *>(row + 32 * s + 16 + 4 * g);
        const half8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        b[s] = ok ? v : zero_h8();
    }
}
template <int HS>
__device__ __forceinline__ void save_rows(half_t *__restrict__ buf, uint64_t p, uint32_t H, uint32_t g,
                                          const half8_t (&b)[HS]) {
    half_t *row = buf + p * H;
#pragma unroll
    for (int s = 0; s < HS; s++) {
        const half4_t lo = {b[s][0], b[s][1], b[s][2], b[s][3]}, hi = {b[s][4], b[s][5], b[s][6], b[s][7]};
        *reinterpret_cast<half4_t *>(row + 32 * s + 4 * g) = lo;
        *reinterpret_cast<half4_t *>(row + 32 * s + 16 + 4 * g) = hi;
    }
}
// gradient through the activation from the stored POST-activation (utils.h:609-664), packed like pack_pair
template <int ACT>
__device__ __forceinline__ half8_t pack_pair_act_bwd(const f32x4 &lo, const f32x4 &hi, const half8_t &h, uint32_t rt) {
    if constexpr (ACT == (int)LNH_ACT_RELU) {
        return pack_pair_relu_bwd(lo, hi, h);
    } else {
        half8_t r;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            r[j] = (half_t)act_bwd<ACT>(rt, lo[j], (float)h[j]);
            r[4 + j] = (half_t)act_bwd<ACT>(rt, hi[j], (float)h[4 + j]);
        }
        return r;
    }
}

// ---------------------------------------------------------------------------------------------------- forward
// LDSW (round 5, hidden 256): a hidden -> hidden matrix is 128 KB — the whole LDS budget of one workgroup per CU — and 128
// fragments per 16-point tile.  Loaded where they are used (L1 / L2), those fragment loads bounded the kernel (1086 us at
// 1 M points against 879 us for the library GEMM chain, profiles/r04_ffmlp_wide.txt); here the 8 waves of a 512-thread
// workgroup stage the matrix of the layer in LDS in FRAGMENT order ([tile t][k-step s][lane]: one conflict-free
// ds_read_b128 per fragment) once per layer and workgroup iteration (256 points), between two barriers.  The iteration
// count is the same for every wave of the grid (points beyond the batch are masked), so the barriers are uniform.
template <int HT, int NT, int ACT, bool LDSW = false>
__global__ void __launch_bounds__(LDSW ? 512 : 256)
k_mlp_forward_wide(WideFwdArgs a) {
    constexpr int HS = HT >= 2 ? HT / 2 : 1;
    constexpr uint32_t H = HT * 16;
    extern __shared__ __attribute__((aligned(16))) char smem_wide[];
    half8_t *wfrag = reinterpret_cast<half8_t *>(smem_wide);  // LDSW: [HT * HS][64]
    const uint32_t lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t wave = blockIdx.x * nw + wid;
    const uint32_t nwaves = gridDim.x * nw;
    const uint32_t in_ks = (a.in_dim + 31) / 32;
    const half_t *W0 = a.W, *Wh = W0 + (size_t)H * a.in_dim, *Wo = Wh + (size_t)a.nhm * H * H;
    const uint64_t stride = (uint64_t)nwaves * NT * 16;
    const uint64_t base_end = LDSW ? (a.B + stride - 1) / stride * stride : a.B;  // LDSW: every wave walks the same number of tiles
    for (uint64_t base = (uint64_t)wave * NT * 16; base < base_end; base += stride) {
        half8_t bx[NT][kMaxInKs];
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const uint64_t p = base + n * 16 + c;
#pragma unroll
            for (int s = 0; s < kMaxInKs; s++) {
                const uint32_t k0 = 32 * s + 8 * g;
                const bool ok = p < a.B && k0 < a.in_dim;
                const half8_t v = *reinterpret_cast<const half8_t *>(a.X + (ok ? p : 0) * a.in_dim + (ok ? k0 : 0));
                bx[n][s] = ok ? v : zero_h8();
            }
        }
        f32x4 acc[HT][NT];
        half8_t bh[NT][HS];
        auto pack = [&]() {
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int s = 0; s < HS; s++) bh[n][s] = pack_pair_act<ACT>(acc[2 * s][n], acc[2 * s + 1][n], a.act);
        };
        auto save = [&](uint32_t layer) {
            if (!a.fb) return;
#pragma unroll
            for (int n = 0; n < NT; n++) {
                const uint64_t p = base + n * 16 + c;
                if (p < a.B) save_rows<HS>(a.fb + (size_t)layer * a.B * H, p, H, g, bh[n]);
            }
        };
        // ---- layer 0
#pragma unroll
        for (int t = 0; t < HT; t++) {
#pragma unroll
            for (int n = 0; n < NT; n++) acc[t][n] = zero_f4();
#pragma unroll
            for (int s = 0; s < kMaxInKs; s++) {
                if (s < (int)in_ks) {  // (wave-uniform)
                    const half8_t w = load_a_natural(W0, a.in_dim, 16 * t + c, s, g, a.in_dim);
#pragma unroll
                    for (int n = 0; n < NT; n++) acc[t][n] = MFMA16(w, bx[n][s], acc[t][n]);
                }
            }
        }
        pack();
        save(0);
        // ---- hidden -> hidden
        for (uint32_t m = 0; m < a.nhm; m++) {
            const half_t *Wm = Wh + (size_t)m * H * H;
            if constexpr (LDSW) {
                __syncthreads();  // (every wave is done with the fragments of the previous layer / iteration)
                for (uint32_t f = wid; f < (uint32_t)(HT * HS); f += nw)
                    wfrag[f * 64 + lane] = load_a_nu(Wm, H, 16 * (f / HS) + c, f % HS, g);
                __syncthreads();
            }
#pragma unroll
            for (int t = 0; t < HT; t++) {
#pragma unroll
                for (int n = 0; n < NT; n++) acc[t][n] = zero_f4();
#pragma unroll
                for (int s = 0; s < HS; s++) {
                    const half8_t w = LDSW ? wfrag[(t * HS + s) * 64 + lane] : load_a_nu(Wm, H, 16 * t + c, s, g);
#pragma unroll
                    for (int n = 0; n < NT; n++) acc[t][n] = MFMA16(w, bh[n][s], acc[t][n]);
                }
            }
            pack();
            save(m + 1);
        }
        // ---- output layer (16 padded outputs)
        f32x4 o[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) o[n] = zero_f4();
#pragma unroll
        for (int s = 0; s < HS; s++) {
            const half8_t w = load_a_nu(Wo, H, c, s, g);
#pragma unroll
            for (int n = 0; n < NT; n++) o[n] = MFMA16(w, bh[n][s], o[n]);
        }
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const uint64_t p = base + n * 16 + c;
            if (p < a.B) {
                if (a.out_act != LNH_ACT_NONE) {
#pragma unroll
                    for (int r = 0; r < 4; r++) o[n][r] = act_forward(a.out_act, o[n][r]);
                }
                const half4_t v = {(half_t)o[n][0], (half_t)o[n][1], (half_t)o[n][2], (half_t)o[n][3]};
                *reinterpret_cast<half4_t *>(a.Y + p * 16 + 4 * g) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------- backward (data)
template <int HT, int NT, int ACT>
__global__ void __launch_bounds__(256)
k_mlp_backward_data_wide(WideBwdArgs a) {
    constexpr int HS = HT >= 2 ? HT / 2 : 1;
    constexpr uint32_t H = HT * 16;
    const uint32_t lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t in_t = a.in_dim / 16;  // 16-row tiles of dX^T
    const half_t *W0T = a.WT, *WhT = W0T + (size_t)a.in_dim * H, *WoT = WhT + (size_t)a.nhm * H * H;
    for (uint64_t base = (uint64_t)wave * NT * 16; base < a.B; base += (uint64_t)nwaves * NT * 16) {
        bool ok[NT];
        half8_t bg[NT][HS], hh[NT][HS];
        f32x4 acc[HT][NT];
        // ---- output layer: G^T = (Wo^T gy^T) . act'(A_last); the contraction runs over the 16 (padded) outputs
        half8_t by[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const uint64_t p = base + n * 16 + c;
            ok[n] = p < a.B;
            const bool use = ok[n] && g < 2;
            const half8_t v = *reinterpret_cast<const half8_t *>(a.gy + (use ? p : 0) * 16 + (use ? 8 * g : 0));
            by[n] = use ? v : zero_h8();
            load_rows<HS>(a.fb + (size_t)a.nhm * a.B * H, p, H, g, ok[n], hh[n]);
        }
#pragma unroll
        for (int t = 0; t < HT; t++) {
            const half8_t w = load_a_natural(WoT, 16, 16 * t + c, 0, g, 16);
#pragma unroll
            for (int n = 0; n < NT; n++) acc[t][n] = MFMA16(w, by[n], zero_f4());
        }
        auto mask_store = [&](uint32_t layer) {
#pragma unroll
            for (int n = 0; n < NT; n++) {
#pragma unroll
                for (int s = 0; s < HS; s++)
                    bg[n][s] = pack_pair_act_bwd<ACT>(acc[2 * s][n], acc[2 * s + 1][n], hh[n][s], a.act);
                if (ok[n]) save_rows<HS>(a.gb + (size_t)layer * a.B * H, base + n * 16 + c, H, g, bg[n]);
            }
        };
        mask_store(a.nhm);
        // ---- hidden layers, last to first: G_m^T = (Wh_m^T G_(m+1)^T) . act'(A_m)
        for (uint32_t m = a.nhm; m-- > 0;) {
            const half_t *Wm = WhT + (size_t)m * H * H;
#pragma unroll
            for (int n = 0; n < NT; n++) load_rows<HS>(a.fb + (size_t)m * a.B * H, base + n * 16 + c, H, g, ok[n], hh[n]);
#pragma unroll
            for (int t = 0; t < HT; t++) {
#pragma unroll
                for (int n = 0; n < NT; n++) acc[t][n] = zero_f4();
#pragma unroll
                for (int s = 0; s < HS; s++) {
                    const half8_t w = load_a_nu(Wm, H, 16 * t + c, s, g);
#pragma unroll
                    for (int n = 0; n < NT; n++) acc[t][n] = MFMA16(w, bg[n][s], acc[t][n]);
                }
            }
            mask_store(m);
        }
        // ---- input gradient: dX^T = W0^T G_0^T
        if (a.gx) {
            for (uint32_t t = 0; t < in_t; t++) {
                f32x4 d[NT];
#pragma unroll
                for (int n = 0; n < NT; n++) d[n] = zero_f4();
#pragma unroll
                for (int s = 0; s < HS; s++) {
                    const half8_t w = load_a_nu(W0T, H, 16 * t + c, s, g);
#pragma unroll
                    for (int n = 0; n < NT; n++) d[n] = MFMA16(w, bg[n][s], d[n]);
                }
#pragma unroll
                for (int n = 0; n < NT; n++) {
                    if (ok[n]) {
                        const half4_t v = {(half_t)d[n][0], (half_t)d[n][1], (half_t)d[n][2], (half_t)d[n][3]};
                        *reinterpret_cast<half4_t *>(a.gx + (base + n * 16 + c) * a.in_dim + 16 * t + 4 * g) = v;
                    }
                }
            }
        }
    }
}

template <int HT, int NT>
int launch_wide_fwd(const WideFwdArgs &a, hipStream_t s) {
    if constexpr (HT == 16) {  // hidden 256: hidden matrices staged in LDS (128 KiB), one 512-thread workgroup per CU
        if (a.nhm > 0) {
            const size_t lds = (size_t)HT * (HT / 2) * 64 * sizeof(half8_t);
            const uint32_t tiles = div_up(a.B, NT * 16 * 8), cus = (uint32_t)lnh_cu_count();
            const uint32_t grid = tiles < cus ? tiles : cus;
            auto k = a.act == LNH_ACT_RELU ? k_mlp_forward_wide<HT, NT, (int)LNH_ACT_RELU, true> : k_mlp_forward_wide<HT, NT, -1, true>;
            (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            LNH_LAUNCH(k, dim3(grid), dim3(512), lds, s, a);
            return lnh_check_launch("lnh_mlp_forward (wide, LDS-staged)");
        }
    }
    const uint32_t tiles = div_up(a.B, NT * 16 * 4), grid = tiles < 4096 ? tiles : 4096;
    if (a.act == LNH_ACT_RELU)
        LNH_LAUNCH((k_mlp_forward_wide<HT, NT, (int)LNH_ACT_RELU>), dim3(grid), dim3(256), 0, s, a);
    else
        LNH_LAUNCH((k_mlp_forward_wide<HT, NT, -1>), dim3(grid), dim3(256), 0, s, a);
    return lnh_check_launch("lnh_mlp_forward (wide)");
}
template <int HT, int NT>
int launch_wide_bwd(const WideBwdArgs &a, hipStream_t s) {
    const uint32_t tiles = div_up(a.B, NT * 16 * 4), grid = tiles < 4096 ? tiles : 4096;
    if (a.act == LNH_ACT_RELU)
        LNH_LAUNCH((k_mlp_backward_data_wide<HT, NT, (int)LNH_ACT_RELU>), dim3(grid), dim3(256), 0, s, a);
    else
        LNH_LAUNCH((k_mlp_backward_data_wide<HT, NT, -1>), dim3(grid), dim3(256), 0, s, a);
    return lnh_check_launch("lnh_mlp_backward_data");
}

int check_wide(uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t n_hidden_mats) {
    LNH_REQUIRE(input_dim > 0 && input_dim % 16 == 0, LNH_ERR_UNSUPPORTED,
                "FFMLP input_dim should be 16 * m (m > 0), but got %u", input_dim);
    LNH_REQUIRE(input_dim <= 128, LNH_ERR_UNSUPPORTED, "fused MLP: input_dim <= 128 in this build (got %u)", input_dim);
    LNH_REQUIRE(output_dim == 16, LNH_ERR_UNSUPPORTED,
                "FFMLP current only supports (padded) output dim == 16, but got %u", output_dim);
    LNH_REQUIRE(hidden_dim == 32 || hidden_dim == 64 || hidden_dim == 128 || hidden_dim == 256, LNH_ERR_UNSUPPORTED,
                "fused MLP: hidden_dim must be 32, 64, 128 or 256 (16 runs zero-padded on 32), got %u", hidden_dim);
    LNH_REQUIRE(n_hidden_mats <= 14, LNH_ERR_UNSUPPORTED, "fused MLP: at most 14 hidden->hidden matrices (got %u)",
                n_hidden_mats);
    return LNH_OK;
}

}  // namespace

// called by lnh_mlp_forward (mlp.hip) for the shapes its register-resident kernels do not serve
int lnh_mlp_forward_wide(const void *inputs, const void *weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                         uint32_t hidden_dim, uint32_t n_hidden_mats, uint32_t activation, uint32_t output_activation,
                         void *forward_buffer, void *outputs, hipStream_t s) {
    int rc = check_wide(input_dim, output_dim, hidden_dim, n_hidden_mats);
    if (rc) return rc;
    WideFwdArgs a{(const half_t *)inputs, (const half_t *)weights, (half_t *)outputs, (half_t *)forward_buffer, B, input_dim,
                  n_hidden_mats, activation, output_activation};
    switch (hidden_dim) {
        case 32: return launch_wide_fwd<2, 4>(a, s);
        case 64: return launch_wide_fwd<4, 4>(a, s);
        case 128: return launch_wide_fwd<8, 4>(a, s);
        default: return launch_wide_fwd<16, 2>(a, s);
    }
}

extern "C" {

int LNH_MLP_FN(lnh_mlp_backward_data)(const void *grad, const void *forward_buffer, const void *weights_t, uint32_t B,
                                      uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t n_hidden_mats,
                                      uint32_t activation, void *backward_buffer, void *grad_inputs, lnh_stream_t stream) {
    LNH_REQUIRE(grad && forward_buffer && weights_t && backward_buffer, LNH_ERR_INVALID_ARG, "mlp backward (data): null pointer");
    LNH_REQUIRE(activation <= LNH_ACT_NONE && activation != LNH_ACT_SINE, LNH_ERR_UNSUPPORTED,
                "mlp backward: Sine needs stored pre-activations (unsupported by the reference as well, utils.h:626-630)");
    int rc = check_wide(input_dim, output_dim, hidden_dim, n_hidden_mats);
    if (rc) return rc;
    if (B == 0) return LNH_OK;
    WideBwdArgs a{(const half_t *)grad, (const half_t *)forward_buffer, (const half_t *)weights_t, (half_t *)backward_buffer,
                  (half_t *)grad_inputs, B, input_dim, n_hidden_mats, activation};
    hipStream_t s = (hipStream_t)stream;
    switch (hidden_dim) {
        case 32: return launch_wide_bwd<2, 4>(a, s);
        case 64: return launch_wide_bwd<4, 4>(a, s);
        case 128: return launch_wide_bwd<8, 4>(a, s);
        default: return launch_wide_bwd<16, 2>(a, s);
    }
}

}  // extern "C"

}  // namespace LNH_MLP_NS
