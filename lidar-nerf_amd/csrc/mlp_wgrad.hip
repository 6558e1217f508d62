// mlp_wgrad.hip — weight gradients of the WIDE fused MLPs (hidden 128 / 256, or more than two hidden matrices):
//        dW[M, N] += G^T A      G [B, M] = gradient w.r.t. a layer's pre-activations (lnh_mlp_backward_data's backward_buffer),
//                               A [B, N] = that layer's input (forward_buffer / the MLP input), both 16-bit, dW fp32
// — a contraction over the BATCH (a million rows) into a 16 .. 256-wide square.  The reference runs these as split-K CUTLASS
// GEMMs on side streams (lidarnerf/ffmlp/src/ffmlp.cu:1107-1263, cutlass_matmul.h:481-616); rounds 4-5 of this package used
// the library's batched GEMM over 4096-row slices (torch.bmm, 128 / 237 us at B = 1 M for 128^2 / 256^2).
//
// One kernel, then the fixed-order sum of wgrad.h:
//   * a workgroup walks its share of the batch in stages of 64 rows (two MFMA k-steps of 32).  A stage's rows of G and A are copied global -> LDS as
//     they lie (16-byte loads, 16-byte stores: no VALU work) into 512-byte SLABS — one per (16-row tile, 16-channel tile),
//     chunk (row c, channel group q) at byte (4c + q) * 8 — from which gfx950's transposing LDS read (ds_read_b64_tr_b16,
//     mlp_common.h) returns in lane (g, c): channel c of rows 4g .. 4g + 3, i.e. an MFMA operand of the TRANSPOSED tile whose
//     k index runs over batch rows.  Two reads (the span's two row tiles) make one 8-value operand; G and A use the same row
//     enumeration, which is all a dot product asks.
//   * waves form a WM x WN grid over the output; a wave owns RM x RN = up to 4 x 4 tiles of 16 x 16 (64 accumulators) and
//     reads RM + RN operands per span for RM * RN MFMAs: 256 x 256 is 16 waves with 8 operand reads per 16 MFMAs each.
//   * two LDS stage buffers alternate: while stage s is multiplied, stage s + 1 moves from registers into the other buffer and
//     the loads of stage s + 2 are issued — ONE barrier per 64 rows (the first form of the kernel staged 32 rows between two
//     barriers and took 228 / 760 us at 128^2 / 256^2, B = 1 M; see tools/bench_wgrad.py).
//   * slabs are 544 bytes apart: consecutive lane pairs of a copy store land in consecutive slabs, and 512 would put all of
//     them on the same banks.
// Per workgroup one partial of M * N floats (tile-major, the MFMA D layout: WgradTileMap), summed in index order.
#include "mlp_common.h"
#include "wgrad.h"
#include <type_traits>

namespace LNH_MLP_NS {
namespace {

constexpr uint32_t kWgSlabStride = 544;  // 512 + 32: see above
constexpr uint32_t kWgSpan = 32;         // rows of one MFMA k-step
constexpr uint32_t kWgStage = 64;        // rows staged per barrier (two spans); two stages of LDS alternate

struct WideWgradArgs {
    const half_t *G, *A;  // [B, M], [B, N]
    uint32_t B, M, N;
    uint32_t WN, RM, RN;  // waves along N; tiles per wave along M / N = the kernel's template arguments (wave w: block (w / WN, w % WN))
    uint32_t stages_per_wg;
    WgradWs ws;
};

// LDS, per stage buffer: [G slabs: 4 row tiles x M/16][A slabs: 4 x N/16]; slab (row tile n, channel tile t) of an operand
// with T channel tiles at (n * T + t) * kWgSlabStride
template <int RM, int RN, int DEPTH, int NTHREADS, int MAXP = 4>
__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_mlp_wgrad_wide(WideWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t tid = threadIdx.x, nt = blockDim.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t TM = a.M >> 4, TN = a.N >> 4;
    constexpr uint32_t NRT = kWgStage / 16;  // row tiles per stage
    const uint32_t buf_bytes = NRT * (TM + TN) * kWgSlabStride;
    // this wave's block of output tiles
    const uint32_t bm = (wid / a.WN) * RM, bn = (wid % a.WN) * RN;
    const uint32_t rm = bm < TM ? min((uint32_t)RM, TM - bm) : 0u, rn = bn < TN ? min((uint32_t)RN, TN - bn) : 0u;
    f32x4 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; i++)
#pragma unroll
        for (int j = 0; j < RN; j++) acc[i][j] = zero_f4();

    // copy plan: 16-byte pieces of a stage; piece q of an operand with C channels: row q / (C / 8), channels 8 (q % (C / 8)) ..
    const uint32_t pcsG = kWgStage * (a.M >> 3), pcsA = kWgStage * (a.N >> 3), pcs = pcsG + pcsA;
    // (MAXP = 16-byte pieces per thread and stage: 64 * (256 + 256) / 8 = 4096 pieces at 1024 threads = 4)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 regs[DEPTH][MAXP];  // DEPTH stages of global loads in flight (2 where a wave has the registers: <= 8 waves)
    const uint32_t n_stages = (a.B + kWgStage - 1) / kWgStage;
    const uint32_t st0 = blockIdx.x * a.stages_per_wg, st1 = min(st0 + a.stages_per_wg, n_stages);
    // (the piece -> address arithmetic does not depend on the stage: once)
    const half_t *src[MAXP];
    uint32_t dst[MAXP], prow[MAXP], ldc[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; i++) {
        const uint32_t p = tid + i * nt;
        const bool isG = p < pcsG;
        const uint32_t q = isG ? p : p - pcsG, C_ = isG ? a.M : a.N, per_row = C_ >> 3, T = isG ? TM : TN;
        const uint32_t r = p < pcs ? q / per_row : 0u, ch = p < pcs ? (q % per_row) * 8 : 0u;
        prow[i] = p < pcs ? r : 0xffffffffu;
        ldc[i] = C_;
        src[i] = (isG ? a.G : a.A) + (size_t)r * C_ + ch;
        dst[i] = (isG ? 0u : NRT * TM * kWgSlabStride) + ((r >> 4) * T + (ch >> 4)) * kWgSlabStride +
                 (4 * (r & 15u) + ((ch & 15u) >> 2)) * 8;
    }
    auto fetch = [&](uint32_t st, u32x4 (&rg)[MAXP]) {
        const uint32_t row0 = st * kWgStage;
#pragma unroll
        for (int i = 0; i < MAXP; i++) {
            rg[i] = (u32x4){0u, 0u, 0u, 0u};
            if (prow[i] != 0xffffffffu && row0 + prow[i] < a.B)
                rg[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src[i] + (size_t)row0 * ldc[i]));
        }
    };
    auto stage = [&](char *buf, const u32x4 (&rg)[MAXP]) {
#pragma unroll
        for (int i = 0; i < MAXP; i++)
            if (prow[i] != 0xffffffffu) *reinterpret_cast<u32x4 *>(buf + dst[i]) = rg[i];
    };
    auto products = [&](const char *cur) {
        if (rm && rn) {
            const char *slabG = cur, *slabA = cur + NRT * TM * kWgSlabStride;
#pragma unroll
            for (uint32_t sp = 0; sp < kWgStage / kWgSpan; sp++) {
                // (a block that hangs over the edge of the output — 7 column tiles in blocks of 4 — reads its last valid tile
                //  again instead of branching; the products it forms there are never stored)
                half8_t fa[RM], fb[RN];
#pragma unroll
                for (int i = 0; i < RM; i++)
                    fa[i] = slab_get_span(slabG + (size_t)(2 * sp * TM + bm + min((uint32_t)i, rm - 1)) * kWgSlabStride,
                                          TM * kWgSlabStride, lane * 8);
#pragma unroll
                for (int j = 0; j < RN; j++)
                    fb[j] = slab_get_span(slabA + (size_t)(2 * sp * TN + bn + min((uint32_t)j, rn - 1)) * kWgSlabStride,
                                          TN * kWgSlabStride, lane * 8);
#pragma unroll
                for (int i = 0; i < RM; i++)
#pragma unroll
                    for (int j = 0; j < RN; j++) acc[i][j] = MFMA16(fa[i], fb[j], acc[i][j]);
            }
        }
    };
    // Stage st lives in LDS buffer (st - st0) & 1 and travelled through register set (st - st0) % DEPTH.  Iteration st: the
    // registers of stage st + 1 (loaded DEPTH iterations ago) go into the other buffer — its readers finished before the
    // barrier that ended the last iteration —, the loads of stage st + 1 + DEPTH are issued into the freed set, then this
    // stage's products; one barrier.
    if (st0 < st1) {
        fetch(st0, regs[0]);
        stage(lds, regs[0]);
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            if (st0 + 1 + d < st1) fetch(st0 + 1 + d, regs[d]);
    }
    __syncthreads();
    auto iteration = [&](uint32_t st, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        const uint32_t par = (st - st0) & 1u;
        if (st + 1 < st1) {
            stage(lds + (par ^ 1u) * buf_bytes, regs[SET]);
            if (st + 1 + DEPTH < st1) fetch(st + 1 + DEPTH, regs[SET]);
        }
        products(lds + par * buf_bytes);
        __syncthreads();
    };
    uint32_t st = st0;
    if constexpr (DEPTH == 2) {
        for (; st + 1 < st1; st += 2) {
            iteration(st, std::integral_constant<int, 0>{});
            iteration(st + 1, std::integral_constant<int, 1>{});
        }
        if (st < st1) iteration(st, std::integral_constant<int, 0>{});
    } else {
        for (; st < st1; st++) iteration(st, std::integral_constant<int, 0>{});
    }
    // partial: tile (tm, tn) at (tm * TN + tn) * 256 floats, element r of lane l at r * 64 + l
    float *mine = wgrad_partial(a.ws, a.M * a.N);
#pragma unroll
    for (int i = 0; i < RM; i++)
#pragma unroll
        for (int j = 0; j < RN; j++)
            if ((uint32_t)i < rm && (uint32_t)j < rn) {
                float *t = mine + (size_t)((bm + i) * TN + bn + j) * 256;
#pragma unroll
                for (int r = 0; r < 4; r++) t[r * 64 + lane] = acc[i][j][r];
            }
}

}  // namespace

extern "C" {

int LNH_MLP_FN(lnh_mlp_wgrad)(const void *grad, const void *acts, uint32_t B, uint32_t M, uint32_t N, float *grad_weights,
                              void *wgrad_ws, uint64_t wgrad_ws_bytes, lnh_stream_t stream) {
    LNH_REQUIRE(grad && acts && grad_weights, LNH_ERR_INVALID_ARG, "mlp wgrad: null pointer");
    LNH_REQUIRE(M >= 16 && N >= 16 && M <= 256 && N <= 256 && M % 16 == 0 && N % 16 == 0, LNH_ERR_UNSUPPORTED,
                "mlp wgrad: M and N must be multiples of 16 in 16 .. 256 (got %u x %u)", M, N);
    LNH_REQUIRE((((uintptr_t)grad | (uintptr_t)acts | (uintptr_t)grad_weights) & 15) == 0, LNH_ERR_INVALID_ARG,
                "mlp wgrad: pointers must be 16-byte aligned");
    WideWgradArgs a{};
    if (int rc = wgrad_ws_open(wgrad_ws, wgrad_ws_bytes, a.ws, "mlp wgrad")) return rc;
    if (B == 0) return LNH_OK;
    a.G = (const half_t *)grad; a.A = (const half_t *)acts; a.B = B; a.M = M; a.N = N;
    const uint32_t TM = M / 16, TN = N / 16;
    // tiles per wave: 4 x 4 where the output has 256 tiles, less on smaller outputs so that a workgroup still has ~8 waves
    // to hide the LDS round trips behind (every wave reads RM + RN operands per span for RM * RN products)
    a.RM = TM >= 4 ? 4 : TM;
    a.RN = TN >= 4 ? 4 : TN;
    while (div_up(TM, a.RM) * div_up(TN, a.RN) < 8 && a.RM * a.RN > 1) {
        if (a.RM >= a.RN) a.RM = (a.RM + 1) / 2; else a.RN = (a.RN + 1) / 2;
    }
    // (256 x 256 = 16 waves of 4 x 4 tiles at 128 registers has no room for a second stage of loads, and one workgroup fills a
    //  CU's LDS: 64 KB in flight per CU, 303 us at B = 1 M against the library's batched GEMM at 229 — the one shape where
    //  this kernel loses; eight waves of 4 x 8 tiles with two stages in flight need 260 registers and spill 149)
    const uint32_t WM = div_up(TM, a.RM);
    a.WN = div_up(TN, a.RN);
    const uint32_t waves = WM * a.WN;  // <= 16
    // at least 4 waves per workgroup and at most 4 copy pieces per thread (idle waves own no tiles, they only copy)
    const uint32_t copy_threads = (div_up(kWgStage * (M + N) / 8, 4) + 63) / 64 * 64;
    const uint32_t threads = max(64 * (waves < 4 ? 4 : waves), copy_threads);
    const uint32_t stages = div_up(B, kWgStage);
    // workgroups: two per CU where the workspace holds that many partials of M * N floats, at least 4 stages each
    const uint64_t cap = kWgradWsFloats / ((uint64_t)M * N);
    uint32_t nwg = stages / 4 ? stages / 4 : 1;
    if (nwg > kWgradMaxBlocks) nwg = kWgradMaxBlocks;
    if (nwg > cap) nwg = (uint32_t)cap;
    a.stages_per_wg = div_up(stages, nwg);
    nwg = div_up(stages, a.stages_per_wg);
    const size_t lds = (size_t)2 * (kWgStage / 16) * (TM + TN) * kWgSlabStride;
    LNH_REQUIRE(threads <= 1024 && kWgStage * (M + N) / 8 <= 4 * threads, LNH_ERR_UNSUPPORTED, "mlp wgrad: copy plan (internal)");
    hipStream_t s = (hipStream_t)stream;
    auto launch = [&](auto kernel) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        LNH_LAUNCH(kernel, dim3(nwg), dim3(threads), lds, s, a);
    };
    // two stages of loads in flight where a wave has the registers for them at four waves per SIMD (<= 4 tiles per wave: 8 spill)
    const bool deep = threads <= 512 && a.RM * a.RN <= 4;
#define LNH_WG_CASE(rm_, rn_)                                                                             \
    case rm_ * 8 + rn_:                                                                                   \
        if (deep) launch(k_mlp_wgrad_wide<rm_, rn_, 2, 512>); else launch(k_mlp_wgrad_wide<rm_, rn_, 1, 1024>); \
        break;
    switch (a.RM * 8 + a.RN) {
        LNH_WG_CASE(1, 1) LNH_WG_CASE(1, 2) LNH_WG_CASE(1, 3) LNH_WG_CASE(1, 4)
        LNH_WG_CASE(2, 1) LNH_WG_CASE(2, 2) LNH_WG_CASE(2, 3) LNH_WG_CASE(2, 4)
        LNH_WG_CASE(3, 1) LNH_WG_CASE(3, 2) LNH_WG_CASE(3, 3) LNH_WG_CASE(3, 4)
        LNH_WG_CASE(4, 1) LNH_WG_CASE(4, 2) LNH_WG_CASE(4, 3) LNH_WG_CASE(4, 4)
        default: lnh_set_error("mlp wgrad: tile plan (internal)"); return LNH_ERR_UNSUPPORTED;
    }
#undef LNH_WG_CASE
    const WgradTileMap map{{{grad_weights, 0, TN, TN, N}, {nullptr, 0xffffffffu, 1, 0, 0}, {nullptr, 0xffffffffu, 1, 0, 0}}};
    wgrad_reduce_launch(a.ws, nwg, M * N, map, s);
    return lnh_check_launch("lnh_mlp_wgrad");
}

}  // extern "C"

}  // namespace LNH_MLP_NS
