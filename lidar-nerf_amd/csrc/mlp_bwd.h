// mlp_bwd.h — backward of the fused tiny MLP on gfx950 MFMA (included by the mlp_bwd_*.hip translation units).
//
// One 256-thread workgroup = 4 waves walks the batch in steps of PB = 4*NT*16 points.  Per step every wave
//   1. recomputes the hidden activations of ITS NT point tiles in registers (mlp_common.h chaining),
//   2. back-propagates dY -> dH_l -> dX in registers with transposed weight fragments,
// and the four waves COOPERATE on the weight gradients dW_l[o][i] = sum_p dH_l[p][o] * A_{l-1}[p][i]:
//   3. each wave drops its dH_l^T / A_{l-1}^T columns into two workgroup-shared LDS tiles ([channel][point] fp16),
//   4. after a barrier the 16x16 output tiles of dW_l are dealt round-robin to the waves; each wave contracts ITS
//      tiles over all PB points with MFMA (k = points) into persistent fp32 accumulators.
// The accumulators live in registers for the whole kernel; at its end every workgroup leaves its partial sums in scratch and
// the partials are added up in workgroup-index order (wgrad.h): bit-reproducible gradients, like the split-K CUTLASS GEMMs
// on saved activations the reference forms them with (lidarnerf/ffmlp/src/ffmlp.cu:1107-1263; here nothing but X, dY and W is
// read from HBM).  Rounds 1-5 flushed with one fp32 device atomic per element and workgroup.
#pragma once
#include "mlp_common.h"
#include "wgrad.h"

namespace LNH_MLP_NS {

struct MlpBwdArgs {
    const half_t *dY;  // [B,16]
    const void *X;     // [B,in_dim] MLP element type (RowMajorIO) / level-major fp16 features (DensityIO): IO::in_t
    const half_t *W;   // flat weights (MLP element type)
    void *dX;          // NULL or the gradient of X, same layout and type as X
    float *dW;         // flat fp32: the sum over the batch is ADDED to it (fixed summation order)
    uint32_t B, in_dim, hidden, act, out_act;
    IoDims io;
    WgradWs ws;        // scratch of that sum (wgrad.h)
};

template <int IN_KS, int HT, int NHM, int NT>
struct BwdCfg {
    static constexpr int HS = HT / 2;              // k-steps over a hidden vector
    static constexpr int IT = IN_KS * 2;           // 16-feature input tiles (upper bound, masked by in_dim)
    static constexpr int PB = 4 * NT * 16;         // points per workgroup step
    static constexpr int LDP = PB + 8;             // LDS row pitch (halves): 16-byte aligned rows, staggered banks
    static constexpr int PKS = PB / 32;            // k-steps over the points of a step
    static constexpr int ROWS_G = HT * 16;         // gradient tile rows (hidden, or 16 for dY)
    static constexpr int ROWS_A = (IT > HT ? IT : HT) * 16;  // activation tile rows
    static constexpr int NQ0 = (HT * IT + 3) / 4;  // dW0 tiles per wave
    static constexpr int NQH = (HT * HT + 3) / 4;  // dWh tiles per wave
    static constexpr int NQO = (HT + 3) / 4;       // dWo tiles per wave
    static constexpr size_t lds_bytes() { return (size_t)(ROWS_G + ROWS_A) * LDP * sizeof(half_t); }
};

template <int IN_KS, int HT, int NHM, int NT, typename IO, int ACT>
__global__ void __launch_bounds__(256)
k_mlp_backward(MlpBwdArgs a) {
    using Cfg = BwdCfg<IN_KS, HT, NHM, NT>;
    constexpr int HS = Cfg::HS, IT = Cfg::IT, PB = Cfg::PB, LDP = Cfg::LDP, PKS = Cfg::PKS;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    half_t *tileG = reinterpret_cast<half_t *>(smem_raw);  // [ROWS_G][LDP]  gradient^T
    half_t *tileA = tileG + Cfg::ROWS_G * LDP;             // [ROWS_A][LDP]  activation^T

    const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const uint32_t hidden = HT * 16, in_dim = a.in_dim, act = a.act, in_tiles = in_dim / 16;
    const bool want_dx = a.dX != nullptr;
    const uint32_t col0 = wid * NT * 16;  // this wave's first column inside the step

    const half_t *W0 = a.W;
    const half_t *Wh = W0 + (size_t)hidden * in_dim;
    const half_t *Wo = Wh + (size_t)NHM * hidden * hidden;
    float *dW0 = a.dW;
    float *dWh = dW0 + (size_t)hidden * in_dim;
    float *dWo = dWh + (size_t)NHM * hidden * hidden;

    // ---- weight fragments (registers, once per wave)
    half8_t w0[HT][IN_KS];
#pragma unroll
    for (int t = 0; t < HT; t++)
#pragma unroll
        for (int s = 0; s < IN_KS; s++) w0[t][s] = load_a_natural(W0, in_dim, 16 * t + c, s, g, in_dim);
    half8_t wh[NHM > 0 ? NHM : 1][HT][HS], whT[NHM > 0 ? NHM : 1][HT][HS];
#pragma unroll
    for (int m = 0; m < NHM; m++)
#pragma unroll
        for (int t = 0; t < HT; t++)
#pragma unroll
            for (int s = 0; s < HS; s++) {
                wh[m][t][s] = load_a_nu(Wh + (size_t)m * hidden * hidden, hidden, 16 * t + c, s, g);
                whT[m][t][s] = load_at_nu(Wh + (size_t)m * hidden * hidden, hidden, 16 * t + c, s, g);
            }
    half8_t woT[HT];  // dH_last^T = Wo^T dY^T: K = 16 outputs in one zero-padded k-step
#pragma unroll
    for (int t = 0; t < HT; t++) woT[t] = load_at_natural(Wo, hidden, 16 * t + c, 0, g, 16);
    half8_t w0T[IT][HS];
#pragma unroll
    for (int t = 0; t < IT; t++)
#pragma unroll
        for (int s = 0; s < HS; s++)
            w0T[t][s] = (want_dx && (uint32_t)t < in_tiles) ? load_at_nu(W0, in_dim, 16 * t + c, s, g) : zero_h8();

    // ---- this wave's share of the weight-gradient tiles
    f32x4 gW0[Cfg::NQ0], gWh[NHM > 0 ? NHM : 1][Cfg::NQH], gWo[Cfg::NQO];
#pragma unroll
    for (int q = 0; q < Cfg::NQ0; q++) gW0[q] = zero_f4();
#pragma unroll
    for (int m = 0; m < NHM; m++)
#pragma unroll
        for (int q = 0; q < Cfg::NQH; q++) gWh[m][q] = zero_f4();
#pragma unroll
    for (int q = 0; q < Cfg::NQO; q++) gWo[q] = zero_f4();

    auto put_packed = [&](half_t *tile, const half8_t (&v)[NT][HS]) {  // rows nu(s,g,j), columns of this wave
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int s = 0; s < HS; s++)
#pragma unroll
                for (int j = 0; j < 8; j++)
                    tile[(16 * (2 * s + (j >> 2)) + 4 * g + (j & 3)) * LDP + col0 + 16 * n + c] = v[n][s][j];
    };
    // contraction over the PB points of the step: acc += G^T rows [16*tg, +16) x A^T rows [16*ta, +16)
    auto wgrad_tile = [&](f32x4 acc, uint32_t tg, uint32_t ta) {
#pragma unroll
        for (int ks = 0; ks < PKS; ks++) {
            const half8_t fa = *reinterpret_cast<const half8_t *>(tileG + (16 * tg + c) * LDP + 32 * ks + 8 * g);
            const half8_t fb = *reinterpret_cast<const half8_t *>(tileA + (16 * ta + c) * LDP + 32 * ks + 8 * g);
            acc = MFMA16(fa, fb, acc);
        }
        return acc;
    };

    for (uint64_t step = (uint64_t)blockIdx.x * PB; step < a.B; step += (uint64_t)gridDim.x * PB) {
        const uint64_t base = step + col0;
        // ---- loads + forward recompute
        half8_t bx[NT][IN_KS], by[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const uint64_t p = base + n * 16 + c;
#pragma unroll
            for (int s = 0; s < IN_KS; s++) {
                const uint32_t k0 = 32 * s + 8 * g;
                // unconditional loads from clamped addresses + register selects (see mlp.hip)
                const bool ok = p < a.B && k0 < in_dim;
                const half8_t v = IO::load_x((const typename IO::in_t *)a.X, ok ? p : 0, ok ? k0 : 0, a.B, in_dim, a.io);
                bx[n][s] = ok ? v : zero_h8();
            }
            const bool oky = p < a.B && g < 2;
            const half8_t vy = *reinterpret_cast<const half8_t *>(a.dY + IO::out_row(oky ? p : 0, a.io) * 16 + (oky ? 8 * g : 0));
            by[n] = oky ? vy : zero_h8();
        }
        half8_t bh[NHM + 1][NT][HS];
#pragma unroll
        for (int n = 0; n < NT; n++) {
            f32x4 acc[HT];
#pragma unroll
            for (int t = 0; t < HT; t++) {
                acc[t] = zero_f4();
#pragma unroll
                for (int s = 0; s < IN_KS; s++) acc[t] = MFMA16(w0[t][s], bx[n][s], acc[t]);
            }
#pragma unroll
            for (int s = 0; s < HS; s++)
                bh[0][n][s] = pack_pair_act<ACT>(acc[2 * s], acc[2 * s + 1], act);
        }
#pragma unroll
        for (int m = 0; m < NHM; m++)
#pragma unroll
            for (int n = 0; n < NT; n++) {
                f32x4 acc[HT];
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    acc[t] = zero_f4();
#pragma unroll
                    for (int s = 0; s < HS; s++) acc[t] = MFMA16(wh[m][t][s], bh[m][n][s], acc[t]);
                }
#pragma unroll
                for (int s = 0; s < HS; s++)
                    bh[m + 1][n][s] = pack_pair_act<ACT>(acc[2 * s], acc[2 * s + 1], act);
            }

        // ---- output matrix: dWo[o][i] += sum_p dY[p][o] h_last[p][i]
        if (g < 2) {
#pragma unroll
            for (int n = 0; n < NT; n++)
#pragma unroll
                for (int j = 0; j < 8; j++) tileG[(8 * g + j) * LDP + col0 + 16 * n + c] = by[n][j];
        }
        put_packed(tileA, bh[NHM]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Cfg::NQO; q++) {
            const uint32_t i = 4 * q + wid;
            if (i < HT) gWo[q] = wgrad_tile(gWo[q], 0, i);
        }
        __syncthreads();

        // ---- dH_last^T = Wo^T dY^T through the last hidden activation
        half8_t bd[NT][HS];
#pragma unroll
        for (int n = 0; n < NT; n++) {
            f32x4 acc[HT];
#pragma unroll
            for (int t = 0; t < HT; t++) acc[t] = MFMA16(woT[t], by[n], zero_f4());
#pragma unroll
            for (int s = 0; s < HS; s++) {
                if constexpr (ACT == (int)LNH_ACT_RELU) {
                    bd[n][s] = pack_pair_relu_bwd(acc[2 * s], acc[2 * s + 1], bh[NHM][n][s]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        bd[n][s][j] = (half_t)act_bwd<ACT>(act, acc[2 * s][j], (float)bh[NHM][n][s][j]);
                        bd[n][s][4 + j] = (half_t)act_bwd<ACT>(act, acc[2 * s + 1][j], (float)bh[NHM][n][s][4 + j]);
                    }
                }
            }
        }
        // ---- hidden matrices, last to first
#pragma unroll
        for (int m = NHM - 1; m >= 0; m--) {
            put_packed(tileG, bd);
            put_packed(tileA, bh[m]);
            __syncthreads();
#pragma unroll
            for (int q = 0; q < Cfg::NQH; q++) {
                const uint32_t idx = 4 * q + wid;
                if (idx < HT * HT) gWh[m][q] = wgrad_tile(gWh[m][q], idx / HT, idx % HT);
            }
            __syncthreads();
#pragma unroll
            for (int n = 0; n < NT; n++) {
                f32x4 acc[HT];
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    acc[t] = zero_f4();
#pragma unroll
                    for (int s = 0; s < HS; s++) acc[t] = MFMA16(whT[m][t][s], bd[n][s], acc[t]);
                }
#pragma unroll
                for (int s = 0; s < HS; s++) {
                    if constexpr (ACT == (int)LNH_ACT_RELU) {
                        bd[n][s] = pack_pair_relu_bwd(acc[2 * s], acc[2 * s + 1], bh[m][n][s]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            bd[n][s][j] = (half_t)act_bwd<ACT>(act, acc[2 * s][j], (float)bh[m][n][s][j]);
                            bd[n][s][4 + j] = (half_t)act_bwd<ACT>(act, acc[2 * s + 1][j], (float)bh[m][n][s][4 + j]);
                        }
                    }
                }
            }
        }
        // ---- first matrix: dW0[o][i] += sum_p dH_0[p][o] x[p][i]
        put_packed(tileG, bd);
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int s = 0; s < IN_KS; s++)
#pragma unroll
                for (int j = 0; j < 8; j++) tileA[(32 * s + 8 * g + j) * LDP + col0 + 16 * n + c] = bx[n][s][j];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Cfg::NQ0; q++) {
            const uint32_t idx = 4 * q + wid;
            if (idx < HT * IT) gW0[q] = wgrad_tile(gW0[q], idx / IT, idx % IT);
        }
        __syncthreads();
        // ---- dX^T = W0^T dH_0^T
        if (want_dx) {
#pragma unroll
            for (int n = 0; n < NT; n++) {
                const uint64_t p = base + n * 16 + c;
#pragma unroll
                for (int t = 0; t < IT; t++) {
                    f32x4 acc = zero_f4();
#pragma unroll
                    for (int s = 0; s < HS; s++) acc = MFMA16(w0T[t][s], bd[n][s], acc);
                    if (p < a.B && (uint32_t)t < in_tiles) IO::store_dx((typename IO::in_t *)a.dX, p, t, g, a.B, in_dim, acc);
                }
            }
        }
    }

    // ---- this workgroup's partial sums, tile by tile (256 floats per tile: element r of lane 16 g + c = row 4g + r, col c),
    //      for the sum over workgroups in index order (k_wgrad_reduce, wgrad.h: the launcher's second launch)
    constexpr uint32_t NT0 = HT * IT, NTH = NHM * HT * HT, NTILES = NT0 + NTH + HT;
    static_assert(NTILES * 256 <= kWgradMaxFloats, "weight-gradient workspace: partial larger than the plan");
    float *mine = wgrad_partial(a.ws, NTILES * 256);
#pragma unroll
    for (int q = 0; q < Cfg::NQ0; q++) {
        const uint32_t idx = 4 * q + wid;
        if (idx < NT0) {
#pragma unroll
            for (int r = 0; r < 4; r++) mine[(idx * 4 + r) * 64 + lane] = gW0[q][r];
        }
    }
#pragma unroll
    for (int m = 0; m < NHM; m++)
#pragma unroll
        for (int q = 0; q < Cfg::NQH; q++) {
            const uint32_t idx = 4 * q + wid;
            if (idx < HT * HT) {
#pragma unroll
                for (int r = 0; r < 4; r++) mine[((NT0 + m * HT * HT + idx) * 4 + r) * 64 + lane] = gWh[m][q][r];
            }
        }
#pragma unroll
    for (int q = 0; q < Cfg::NQO; q++) {
        const uint32_t i = 4 * q + wid;
        if (i < HT) {
#pragma unroll
            for (int r = 0; r < 4; r++) mine[((NT0 + NTH + i) * 4 + r) * 64 + lane] = gWo[q][r];
        }
    }
}


// ---------------------------------------------------------------------------------------------------- wave-independent
// Backward for ONE-hidden-layer nets (NHM = 0: the sigma net) without workgroup barriers in the point loop.
//
// Every wave back-propagates its 32 points through the layer chain in registers (channel-major: lane (g, c) holds channels
// 4g + r of point c) and accumulates ALL weight-gradient tiles of the net in its own registers.  A weight gradient
// contracts over points, so its MFMA operands are the transposes of the chain's fragments: each packed fragment (hidden
// activation, its gradient, the input rows, the output gradient) is written to a wave-private LDS slab as it is formed and
// read back transposed with ds_read_b64_tr_b16 (mlp_common.h, "transposes through LDS").  Rounds 1-4 obtained the
// transposed operands by running the chain a second time with the MFMA operands swapped (the A and B layouts are mirror
// images, so the product comes out point-major) and by multiplying the quantities that are not products with an identity
// fragment: 22 MFMAs and a second copy of the activation arithmetic per 32 points, now 12 LDS writes and 22 LDS reads.
// Waves never wait for each other; the partial sums of a workgroup are combined through LDS once at the end of the
// kernel (wave order) and the workgroups' sums are added up in index order (wgrad.h).
template <int IN_KS, int HT>
struct WiCfg {
    static constexpr int IT = IN_KS * 2, NT = 2;
    // slabs of a wave: hidden activation (n, t), its gradient (n, t), input rows (n, i), output gradient (n)
    static constexpr int SL_H = 0, SL_D = NT * HT, SL_X = 2 * NT * HT, SL_Y = SL_X + NT * IT, NSLAB = SL_Y + NT;
    static constexpr int NTILE = HT + HT * IT;
    static constexpr size_t lds_bytes(int nwaves) {
        const size_t slabs = (size_t)nwaves * NSLAB * kSlabBytes, red = (size_t)NTILE * 256 * sizeof(float);
        return slabs > red ? slabs : red;
    }
};

#ifndef LNH_WI_WAVES
#define LNH_WI_WAVES 8  // waves per workgroup = 2 per SIMD
#endif
template <int IN_KS, int HT, typename IO, int ACT>
__global__ void __launch_bounds__(64 * LNH_WI_WAVES)
k_mlp_backward_wi(MlpBwdArgs a) {
    using Cfg = WiCfg<IN_KS, HT>;
    constexpr int HS = HT / 2, IT = Cfg::IT, NT = Cfg::NT;
    extern __shared__ __attribute__((aligned(16))) char smem_wi[];
    float *red = reinterpret_cast<float *>(smem_wi);  // (after the point loop, over the slabs)
    const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const uint32_t hidden = HT * 16, in_dim = a.in_dim, act = a.act, in_tiles = in_dim / 16;
    const bool want_dx = a.dX != nullptr;
    const uint32_t nw = blockDim.x >> 6;
    const uint32_t wave = blockIdx.x * nw + wid, nwaves = gridDim.x * nw;
    char *slabs = smem_wi + (size_t)__builtin_amdgcn_readfirstlane((int)wid) * Cfg::NSLAB * kSlabBytes;
    const uint32_t wr_off = (4 * c + g) * 8, rd_off = lane * 8;
    // natural-k fragments (input rows, output gradients): lane (g, c) holds values 8g .. 8g+7 of a 32-wide k-step =
    // chunk groups 2(g & 1), 2(g & 1) + 1 of the 16-wide tile g >> 1 of that step
    const uint32_t wr_nat = (g >> 1) * kSlabBytes + c * 32 + (g & 1) * 16;

    const half_t *W0 = a.W;
    const half_t *Wo = W0 + (size_t)hidden * in_dim;
    half8_t w0[HT][IN_KS], woT[HT], w0T[IT][HS];
#pragma unroll
    for (int t = 0; t < HT; t++) {
#pragma unroll
        for (int s = 0; s < IN_KS; s++) w0[t][s] = load_a_natural(W0, in_dim, 16 * t + c, s, g, in_dim);
        woT[t] = load_at_natural(Wo, hidden, 16 * t + c, 0, g, 16);
    }
#pragma unroll
    for (int t = 0; t < IT; t++)
#pragma unroll
        for (int s = 0; s < HS; s++)
            w0T[t][s] = (want_dx && (uint32_t)t < in_tiles) ? load_at_nu(W0, in_dim, 16 * t + c, s, g) : zero_h8();
    f32x4 gWo[HT], gW0[HT][IT];
#pragma unroll
    for (int t = 0; t < HT; t++) {
        gWo[t] = zero_f4();
#pragma unroll
        for (int i = 0; i < IT; i++) gW0[t][i] = zero_f4();
    }
    auto slab = [&](int q, int idx) { return slabs + (q + idx) * kSlabBytes; };

    // The kernel runs at 2 waves per SIMD, too few to hide the HBM latency of a tile's input rows behind another wave's
    // arithmetic: the rows of the NEXT tile are requested before the current tile is processed (software pipelining, one
    // tile deep: 153 -> 137 us; two deep measures the same and makes the wider instantiations spill).
    auto load_tile = [&](uint64_t base, half8_t (&bx)[NT][IN_KS], half8_t (&by)[NT]) {
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const uint64_t p = base + n * 16 + c;
#pragma unroll
            for (int s = 0; s < IN_KS; s++) {
                const uint32_t k0 = 32 * s + 8 * g;
                // unconditional loads from clamped addresses + register selects (see mlp.hip)
                const bool ok = p < a.B && k0 < in_dim;
                const half8_t v = IO::load_x((const typename IO::in_t *)a.X, ok ? p : 0, ok ? k0 : 0, a.B, in_dim, a.io);
                bx[n][s] = ok ? v : zero_h8();
            }
            const bool oky = p < a.B && g < 2;
            const half8_t vy = *reinterpret_cast<const half8_t *>(a.dY + IO::out_row(oky ? p : 0, a.io) * 16 + (oky ? 8 * g : 0));
            by[n] = oky ? vy : zero_h8();
        }
    };
    constexpr bool PREFETCH = IN_KS == 1;  // (the 64-input instantiations have no registers to spare)
    const uint64_t stride = (uint64_t)nwaves * NT * 16;
    half8_t nx[NT][IN_KS], ny[NT];
    if constexpr (PREFETCH) load_tile((uint64_t)wave * NT * 16, nx, ny);
    for (uint64_t base = (uint64_t)wave * NT * 16; base < a.B; base += stride) {
        half8_t bx[NT][IN_KS], by[NT];
        if constexpr (PREFETCH) {
#pragma unroll
            for (int n = 0; n < NT; n++) {
#pragma unroll
                for (int s = 0; s < IN_KS; s++) bx[n][s] = nx[n][s];
                by[n] = ny[n];
            }
            load_tile(base + stride, nx, ny);  // (beyond the batch: clamped addresses, zero fragments, never used)
        } else {
            load_tile(base, bx, by);
        }
#pragma unroll
        for (int n = 0; n < NT; n++) {
#pragma unroll
            for (int s = 0; s < IN_KS; s++)
                *reinterpret_cast<half8_t *>(slab(Cfg::SL_X, n * IT + 2 * s) + wr_nat) = bx[n][s];
            if (g < 2) *reinterpret_cast<half8_t *>(slab(Cfg::SL_Y, n) + wr_nat) = by[n];
        }
        // ---- channel-major chain: hidden, its gradient, dX
#pragma unroll
        for (int n = 0; n < NT; n++) {
            f32x4 h[HT], d[HT];
#pragma unroll
            for (int t = 0; t < HT; t++) {
                h[t] = zero_f4();
#pragma unroll
                for (int s = 0; s < IN_KS; s++) h[t] = MFMA16(w0[t][s], bx[n][s], h[t]);
                d[t] = MFMA16(woT[t], by[n], zero_f4());
            }
            half8_t bd[HS];
#pragma unroll
            for (int s = 0; s < HS; s++) {
                half8_t bh;
                if constexpr (ACT == (int)LNH_ACT_RELU) {
                    bh = pack_pair_relu(h[2 * s], h[2 * s + 1]);
                    bd[s] = pack_pair_relu_bwd(d[2 * s], d[2 * s + 1], bh);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        bh[j] = (half_t)act_fwd<ACT>(act, h[2 * s][j]);
                        bh[4 + j] = (half_t)act_fwd<ACT>(act, h[2 * s + 1][j]);
                        bd[s][j] = (half_t)act_bwd<ACT>(act, d[2 * s][j], (float)bh[j]);
                        bd[s][4 + j] = (half_t)act_bwd<ACT>(act, d[2 * s + 1][j], (float)bh[4 + j]);
                    }
                }
                slab_put_pair(slab(Cfg::SL_H, n * HT + 2 * s), wr_off, bh);
                slab_put_pair(slab(Cfg::SL_D, n * HT + 2 * s), wr_off, bd[s]);
            }
            if (want_dx) {
                const uint64_t p = base + n * 16 + c;
#pragma unroll
                for (int t = 0; t < IT; t++) {
                    f32x4 acc = zero_f4();
#pragma unroll
                    for (int s = 0; s < HS; s++) acc = MFMA16(w0T[t][s], bd[s], acc);
                    if (p < a.B && (uint32_t)t < in_tiles) IO::store_dx((typename IO::in_t *)a.dX, p, t, g, a.B, in_dim, acc);
                }
            }
        }
        // ---- weight gradients from the transposed (point-major) operands
        {
            const half8_t fy = slab_get_span(slab(Cfg::SL_Y, 0), kSlabBytes, rd_off);
            half8_t fx[IT];
#pragma unroll
            for (int i = 0; i < IT; i++) fx[i] = slab_get_span(slab(Cfg::SL_X, i), IT * kSlabBytes, rd_off);
#pragma unroll
            for (int t = 0; t < HT; t++) {
                const half8_t fh = slab_get_span(slab(Cfg::SL_H, t), HT * kSlabBytes, rd_off);
                const half8_t fd = slab_get_span(slab(Cfg::SL_D, t), HT * kSlabBytes, rd_off);
                gWo[t] = MFMA16(fy, fh, gWo[t]);  // dWo[o][16t + c]
#pragma unroll
                for (int i = 0; i < IT; i++) gW0[t][i] = MFMA16(fd, fx[i], gW0[t][i]);  // dW0[16t + .][16i + c]
            }
        }
    }

    // ---- combine the waves of the workgroup through LDS (wave order); the workgroups' sums are added in index order by the
    //      launcher's second launch (k_wgrad_reduce, wgrad.h)
    constexpr int NTILE = Cfg::NTILE;
    __syncthreads();  // (the reduction buffer lies over the slabs)
    for (uint32_t w = 0; w < nw; w++) {
        if (wid == w) {
#pragma unroll
            for (int t = 0; t < HT; t++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float *q = red + (t * 4 + r) * 64 + lane;
                    *q = (w == 0 ? 0.0f : *q) + gWo[t][r];
                }
#pragma unroll
                for (int i = 0; i < IT; i++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float *q = red + ((HT + t * IT + i) * 4 + r) * 64 + lane;
                        *q = (w == 0 ? 0.0f : *q) + gW0[t][i][r];
                    }
            }
        }
        __syncthreads();
    }
    {
        float4 *mine = reinterpret_cast<float4 *>(wgrad_partial(a.ws, NTILE * 256));
        for (uint32_t e = threadIdx.x; e < NTILE * 64; e += blockDim.x) mine[e] = reinterpret_cast<const float4 *>(red)[e];
    }
}

template <int IN_KS, int HT, typename IO = RowMajorIO>
int launch_mlp_backward_wi(const MlpBwdArgs &a, hipStream_t s) {
    const uint32_t iters = div_up(a.B, LNH_WI_WAVES * 32);
    const uint32_t grid = iters < 256 ? iters : 256;
    const size_t lds = WiCfg<IN_KS, HT>::lds_bytes(LNH_WI_WAVES);
    auto k = a.act == LNH_ACT_RELU ? k_mlp_backward_wi<IN_KS, HT, IO, (int)LNH_ACT_RELU> : k_mlp_backward_wi<IN_KS, HT, IO, -1>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    LNH_LAUNCH(k, dim3(grid), dim3(64 * LNH_WI_WAVES), lds, s, a);
    // partial = [dWo: HT tiles | dW0: HT x IT tiles]
    constexpr uint32_t IT = WiCfg<IN_KS, HT>::IT;
    float *dWo = a.dW + (size_t)a.hidden * a.in_dim;
    const WgradTileMap map{{{dWo, 0, HT, HT, a.hidden}, {a.dW, HT, IT, a.in_dim / 16, a.in_dim}, {nullptr, 0xffffffffu, 1, 0, 0}}};
    wgrad_reduce_launch(a.ws, grid, WiCfg<IN_KS, HT>::NTILE * 256, map, s);
    return lnh_check_launch("lnh_mlp_backward");
}

template <int IN_KS, int HT, int NHM, typename IO = RowMajorIO>
int launch_mlp_backward(const MlpBwdArgs &a, hipStream_t s) {
    constexpr int NT = 2;
    using Cfg = BwdCfg<IN_KS, HT, NHM, NT>;
    const size_t lds = Cfg::lds_bytes();
    const uint32_t steps = div_up(a.B, Cfg::PB);
    const uint32_t grid = steps < kWgradMaxBlocks ? steps : kWgradMaxBlocks;  // (one partial of every weight gradient each)
    auto k = a.act == LNH_ACT_RELU ? k_mlp_backward<IN_KS, HT, NHM, NT, IO, (int)LNH_ACT_RELU>
                                   : k_mlp_backward<IN_KS, HT, NHM, NT, IO, -1>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    LNH_LAUNCH(k, dim3(grid), dim3(256), lds, s, a);
    // partial = [dW0: HT x IT tiles | dWh: NHM x HT x HT tiles | dWo: HT tiles]
    constexpr uint32_t IT = IN_KS * 2, NT0 = HT * IT, NTH = NHM * HT * HT;
    float *dWh = a.dW + (size_t)a.hidden * a.in_dim, *dWo = dWh + (size_t)NHM * a.hidden * a.hidden;
    const WgradTileMap map{{{a.dW, 0, IT, a.in_dim / 16, a.in_dim}, {dWh, NTH ? NT0 : 0xffffffffu, HT, HT, a.hidden},
                            {dWo, NT0 + NTH, HT, HT, a.hidden}}};
    wgrad_reduce_launch(a.ws, grid, (NT0 + NTH + HT) * 256, map, s);
    return lnh_check_launch("lnh_mlp_backward");
}

}  // namespace LNH_MLP_NS
