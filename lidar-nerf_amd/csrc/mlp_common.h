// mlp_common.h — register-resident tiny-MLP building blocks on gfx950 MFMA (v_mfma_f32_16x16x32_f16).
//
// Orientation: every layer is computed TRANSPOSED,  H^T[neuron, point] = W[neuron, k] * X^T[k, point]:
//   A operand = weight matrix (row m = output neuron),  B operand = activations (column n = sample point).
// With the 16x16 C/D layout (lane = 16*g + c holds D[4g + r][c], r = 0..3) a lane ends up with the values
// {neuron 16t + 4g + r} of ITS OWN point c.  The B operand of the next layer needs, per lane, 8 k-values of the same
// point c — and since a dot product does not care in which order k is enumerated as long as A and B agree, we DEFINE
// the k-enumeration of hidden-layer products as
//        k-step s, lane group g, element j  <->  neuron  nu(s, g, j) = 16 * (2s + (j >> 2)) + 4g + (j & 3)
// which is exactly what the lane already holds in its accumulators.  The activation is applied in registers, the
// result is narrowed to fp16 and fed straight back as the next B operand: no LDS, no cross-lane traffic, no global
// round trip between layers.  The weight fragments are gathered once per wave with the same enumeration.
//
// Replaces the shared-memory/WMMA chain of lidarnerf/ffmlp/src/ffmlp.cu:54-180,460-576 (semantics only).
//
// Element type: the whole MLP code base (this header, mlp_bwd.h, mlp.hip, mlp_bwd_*.hip, lidar_color.hip) is compiled
// TWICE into one library — fp16 operands (v_mfma_f32_16x16x32_f16; entry points lnh_*) and, from the *_bf16.hip
// wrapper translation units that define LNH_MLP_BF16, bf16 operands (v_mfma_f32_16x16x32_bf16; entry points
// lnh_*_bf16: BASELINE config 5, "fp16 hash features + bf16 MFMA MLP").  Each build lives in its own namespace, in
// which `half_t` / `half8_t` / ... name the MLP element type (so the kernels read the same either way); accumulation
// is fp32 in both.  The hash-grid features entering the sigma net and the feature gradients leaving it stay fp16
// (`feat_t`, the grid kernels' type) in both builds — DensityIO converts at that boundary.
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 feat_t;                                         // hash-grid features / their gradients: always fp16
typedef _Float16 feat2_t __attribute__((ext_vector_type(2)));

#ifdef LNH_MLP_BF16
#define LNH_MLP_NS lnh_mlp_bf16
#define LNH_MLP_FN(name) name##_bf16
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define LNH_MFMA16_MNEMONIC "v_mfma_f32_16x16x32_bf16"
#else
#define LNH_MLP_NS lnh_mlp_f16
#define LNH_MLP_FN(name) name
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define LNH_MFMA16_MNEMONIC "v_mfma_f32_16x16x32_f16"
#endif

namespace LNH_MLP_NS {
#ifdef LNH_MLP_BF16
typedef __bf16 half_t;  // (shadows the global fp16 typedefs inside this namespace)
typedef __bf16 half2_t __attribute__((ext_vector_type(2)));
typedef __bf16 half4_t __attribute__((ext_vector_type(4)));
typedef __bf16 half8_t __attribute__((ext_vector_type(8)));
constexpr bool kMlpBf16 = true;
#else
constexpr bool kMlpBf16 = false;
#endif

constexpr float kActK = 10.0f;  // utils.h squareplus / softplus sharpness

// lidarnerf/ffmlp/src/utils.h:479-531
__device__ __forceinline__ float act_forward(uint32_t a, float x) {
    switch (a) {
        case LNH_ACT_RELU: return x > 0.0f ? x : 0.0f;
        case LNH_ACT_EXPONENTIAL: return expf(x);
        case LNH_ACT_SINE: return sinf(x);
        case LNH_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
        case LNH_ACT_SQUAREPLUS: {
            const float y = x * kActK;
            return 0.5f * (y + sqrtf(y * y + 4.0f)) / kActK;
        }
        case LNH_ACT_SOFTPLUS: return logf(expf(x * kActK) + 1.0f) / kActK;
        default: return x;
    }
}
// lidarnerf/ffmlp/src/utils.h:609-664 (transfer from the stored POST-activation)
__device__ __forceinline__ float act_backward_post(uint32_t a, float g, float post) {
    switch (a) {
        case LNH_ACT_RELU: return post > 0.0f ? g : 0.0f;
        case LNH_ACT_EXPONENTIAL: return g * post;
        case LNH_ACT_SIGMOID: return g * post * (1.0f - post);
        case LNH_ACT_SQUAREPLUS: {
            const float y = post * kActK;
            return g * (y * y / (y * y + 1.0f));
        }
        case LNH_ACT_SOFTPLUS: return g * (1.0f - expf(-post * kActK));
        default: return g;
    }
}

// Compile-time activation selector: ACT >= 0 folds the switch away (the per-element runtime switch costs one scalar
// branch per value — ~1000 branches per tile iteration — and fences the MFMA schedule); ACT < 0 keeps it for the
// rarely used activations.
template <int ACT>
__device__ __forceinline__ float act_fwd(uint32_t rt, float x) {
    if constexpr (ACT >= 0) return act_forward((uint32_t)ACT, x);
    else return act_forward(rt, x);
}
template <int ACT>
__device__ __forceinline__ float act_bwd(uint32_t rt, float g, float post) {
    if constexpr (ACT >= 0) return act_backward_post((uint32_t)ACT, g, post);
    else return act_backward_post(rt, g, post);
}

__device__ __forceinline__ half8_t zero_h8() {
    const half_t o = (half_t)0.0f;
    half8_t z = {o, o, o, o, o, o, o, o};
    return z;
}
__device__ __forceinline__ f32x4 zero_f4() {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return z;
}

// acc += a * b with the accumulator tile pinned to ACCUMULATION registers (AGPRs).  The MLP translation units are built
// with VGPR-form MFMA destinations (build.py) because the layer chain feeds every product straight into VALU work; the
// weight-gradient tiles of the colour backward (lidar_color.hip) are the opposite case — 24 tiles that are only ever
// accumulated into for the whole kernel.  Left to the register allocator they share the 256 VGPRs with the pipeline
// state, and hipcc then moves all 96 of them to other registers and back around the loop latch (192 v_mov per
// iteration, a fifth of the colour backward's VALU time) and spills pipeline state into AGPRs besides.  In AGPRs they
// cost nothing.  The instruction is opaque to the compiler's hazard recogniser, so the two wait states a VALU-written
// A / B operand needs before an MFMA reads it are part of the statement (both operands always come from packed
// conversions here), and `agpr_settle` covers the MFMA -> v_accvgpr_read distance before the tiles are read back.
// (Not a general win: the sigma-net backward, two waves per SIMD and no branch in its loop, got 8 % slower with it.)
__device__ __forceinline__ void mfma16_acc_agpr(f32x4 &acc, const half8_t &a, const half8_t &b) {
    asm("s_nop 1\n\t" LNH_MFMA16_MNEMONIC " %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void agpr_settle(f32x4 &acc) { asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc)); }

// A fragment of a row-major [rows, ld] matrix with the NATURAL k enumeration (first layer: k = 32s + 8g + j).
__device__ __forceinline__ half8_t load_a_natural(const half_t *__restrict__ W, uint32_t ld, uint32_t row, uint32_t s,
                                                  uint32_t g, uint32_t kmax) {
    const uint32_t k0 = 32 * s + 8 * g;
    if (k0 >= kmax) return zero_h8();
    return *reinterpret_cast<const half8_t *>(W + (size_t)row * ld + k0);
}
// A fragment with the nu() enumeration (hidden layers): two 4-element groups.
__device__ __forceinline__ half8_t load_a_nu(const half_t *__restrict__ W, uint32_t ld, uint32_t row, uint32_t s,
                                             uint32_t g) {
    const half4_t lo = *reinterpret_cast<const half4_t *>(W + (size_t)row * ld + 32 * s + 4 * g);
    const half4_t hi = *reinterpret_cast<const half4_t *>(W + (size_t)row * ld + 32 * s + 16 + 4 * g);
    half8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}
// Transposed A fragment: A[m][k] = W[k-th row][m-th column] of a row-major [rows, ld] matrix, natural k.
// Used by the backward pass (dX^T = W^T dY^T).  kmax = number of valid rows of W.
__device__ __forceinline__ half8_t load_at_natural(const half_t *__restrict__ W, uint32_t ld, uint32_t col,
                                                   uint32_t s, uint32_t g, uint32_t kmax) {
    half8_t r;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t k = 32 * s + 8 * g + j;
        r[j] = k < kmax ? W[(size_t)k * ld + col] : (half_t)0.0f;
    }
    return r;
}
__device__ __forceinline__ half8_t load_at_nu(const half_t *__restrict__ W, uint32_t ld, uint32_t col, uint32_t s,
                                              uint32_t g) {
    half8_t r;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t k = 16 * (2 * s + (j >> 2)) + 4 * g + (j & 3);
        r[j] = W[(size_t)k * ld + col];
    }
    return r;
}

// Pack the accumulators of two consecutive M-tiles (2s, 2s+1) of one point tile into the next B operand.
template <typename F>
__device__ __forceinline__ half8_t pack_pair(const f32x4 &lo, const f32x4 &hi, F &&f) {
    half8_t r;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        r[j] = (half_t)f(lo[j]);
        r[4 + j] = (half_t)f(hi[j]);
    }
    return r;
}

// ReLU variant on packed 16-bit floats: narrow first (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32), then ONE packed op per
// PAIR of values — relu(narrow(x)) == narrow(relu(x)) exactly (rounding is monotonic and keeps the sign), at half the
// VALU instructions of the fp32 form.  These kernels are VALU-issue bound, so this is wall time.  fp16: v_pk_max_f16;
// bf16 (no packed bf16 max on gfx950): v_pk_max_i16 against 0 on the bit patterns — a negative float (sign bit set,
// -0 included) is a negative int16, a positive one keeps its bits.
__device__ __forceinline__ half8_t pack_pair_relu(const f32x4 &lo, const f32x4 &hi) {
    const half4_t a = __builtin_convertvector(lo, half4_t), b = __builtin_convertvector(hi, half4_t);
    const half8_t r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#ifdef LNH_MLP_BF16
    // (as a compiler-visible vector op, NOT inline asm: the result feeds the next MFMA directly, and the hazard
    //  recogniser inserts the VALU-write -> MFMA-read wait states only for instructions it can see)
    typedef short s8 __attribute__((ext_vector_type(8)));
    const s8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(half8_t, __builtin_elementwise_max(__builtin_bit_cast(s8, r), zero));
#else
    return __builtin_elementwise_max(r, zero_h8());
#endif
}
// ReLU backward on packed halves: narrow(d) where the stored (non-negative, never -0) activation h is non-zero, else 0.
// Two packed integer ops per PAIR of values: nz = min(bits(h), 1) per half (0 / 1), bits(d) * nz (16-bit low product:
// the bits themselves or 0; rounds 1-4: m = 0 - nz, d & m — three).  The min is inline asm so that the compiler cannot
// know nz is 0 / 1 (told so, it rewrites the product as 8 compares + 8 selects + 4 byte permutes per fragment); the
// product — the instruction whose result feeds an MFMA operand — is a vector multiply the compiler sees
// (v_pk_mul_lo_u16), because the hazard recogniser only pads instructions it knows (see pack_pair_relu).
__device__ __forceinline__ half8_t pack_pair_relu_bwd(const f32x4 &lo, const f32x4 &hi, const half8_t &h) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef unsigned short us8 __attribute__((ext_vector_type(8)));
    const half4_t a = __builtin_convertvector(lo, half4_t), b = __builtin_convertvector(hi, half4_t);
    const half8_t d = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    const u4 hw = __builtin_bit_cast(u4, h);
    u4 nz;
    const uint32_t ones = 0x00010001u;
#pragma unroll
    for (int i = 0; i < 4; i++) asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz[i]) : "v"(hw[i]), "v"(ones));
    return __builtin_bit_cast(half8_t, (us8)(__builtin_bit_cast(us8, d) * __builtin_bit_cast(us8, nz)));
}

// ---------------------------------------------------------------------------------- transposes through LDS (gfx950)
// Weight gradients contract over POINTS, the layer chain over CHANNELS: the chain leaves, per 16x16 tile, lane (g, c)
// holding channels 4g .. 4g+3 of point c (four 16-bit values = one 8-byte CHUNK), a weight-gradient MFMA operand wants
// lane (g, c) to hold channel c of four points.  ds_read_b64_tr_b16 is that transpose: within a group of 16 lanes,
// element j of lane l comes from the chunk whose address lane 4j + (l >> 2) supplies, element l & 3 of it.  A tile's 64
// chunks are kept as a SLAB of 512 bytes, chunk (point c, channel group q) at byte (4c + q) * 8; the read with the
// lane-linear address slab + 8 * lane then returns, in lane (g, c): channel c of points 4g .. 4g+3.  Two such reads
// (the two 16-point tiles of a 32-point span) are one 8-value operand with the point enumeration k(g, j) = tile j >> 2,
// point 4g + (j & 3) — the same for both operands of a product, which is all a dot product asks.  Slabs are private to
// a wave and LDS instructions of a wave execute in order: no barrier, no fence.  One LDS write per PAIR of tiles and one
// read per tile, no VALU instruction at all.  Used by the sigma-net backward (mlp_bwd.h, two waves per SIMD: 5 % faster
// than its second pass with swapped MFMA operands); the colour backward (one wave per SIMD) keeps its identity-fragment
// MFMAs — nothing runs under an LDS round trip there (profiles/r05_color_backward_wgrad.txt).
typedef short lds_tr_raw_t __attribute__((__vector_size__(4 * sizeof(short))));
constexpr uint32_t kSlabBytes = 512;
// chunks of a nu-enumerated packed pair (M-tiles 2s and 2s+1 of one point tile): slabs of the two tiles 512 bytes apart
__device__ __forceinline__ void slab_put_pair(char *slab2, uint32_t wr_off, const half8_t &v) {
    const half4_t lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<half4_t *>(slab2 + wr_off) = lo;
    *reinterpret_cast<half4_t *>(slab2 + kSlabBytes + wr_off) = hi;
}
__device__ __forceinline__ half4_t slab_get(const char *slab, uint32_t rd_off) {
    const lds_tr_raw_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) lds_tr_raw_t *)(slab + rd_off));
    return __builtin_bit_cast(half4_t, r);
}
// the 8-value operand of one channel tile over the 32 points of a span: slabs of point tiles 0 and 1, `stride` bytes apart
__device__ __forceinline__ half8_t slab_get_span(const char *slab_n0, uint32_t stride, uint32_t rd_off) {
    const half4_t a = slab_get(slab_n0, rd_off), b = slab_get(slab_n0 + stride, rd_off);
    const half8_t r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return r;
}
// activation resolved at compile time: ReLU takes the packed path, everything else the generic one
template <int ACT>
__device__ __forceinline__ half8_t pack_pair_act(const f32x4 &lo, const f32x4 &hi, uint32_t act_rt) {
    if constexpr (ACT == (int)LNH_ACT_RELU) return pack_pair_relu(lo, hi);
    else return pack_pair(lo, hi, [&](float v) { return act_fwd<ACT>(act_rt, v); });
}

// ---------------------------------------------------------------------------------------------------- I/O policies
// The MLP kernels are templated on an IO policy that says where a point's input row / output row / gradient rows
// live.  RowMajorIO is the FFMLP layout ([B,in] / [B,16]).  DensityIO is the fused sigma-net of the LiDAR field:
//   * input  = hash-grid features in the encoder's own level-major layout [L, B, 2] (no [B, L*2] permute copy);
//   * output = the 16 raw outputs as fp16 at a strided destination row (ray r, slot off + j of a [N, Ttot, 16]
//     buffer that holds coarse AND fine samples of a ray side by side) + sigma = exp(out[0]) as fp32
//     (trunc_exp forward, lidarnerf/activation.py:11-13, evaluated on the fp16-rounded pre-activation);
//   * backward reads the gradient rows from the same strided buffer and writes d(features) level-major.
struct IoDims {
    uint32_t T_cur, T_tot, slot_off;  // DensityIO: point p = r*T_cur + j  ->  destination row r*T_tot + slot_off + j
    uint32_t feat_rows;               // DensityIO: != 0 -> the feature buffer has feat_rows rows per level and is
                                      //            addressed by the destination row too (coarse + fine in one buffer)
};

struct RowMajorIO {
    static constexpr bool kDensity = false;
    typedef half_t in_t;  // inputs / input gradients have the MLP element type
    __device__ static __forceinline__ half8_t load_x(const in_t *X, uint64_t p, uint32_t k0, uint32_t B, uint32_t in_dim,
                                                     const IoDims &) {
        return *reinterpret_cast<const half8_t *>(X + p * in_dim + k0);
    }
    __device__ static __forceinline__ uint64_t out_row(uint64_t p, const IoDims &) { return p; }
    __device__ static __forceinline__ void store_dx(in_t *dX, uint64_t p, uint32_t t, uint32_t g, uint32_t B,
                                                    uint32_t in_dim, const f32x4 &acc) {
        half4_t v = {(half_t)acc[0], (half_t)acc[1], (half_t)acc[2], (half_t)acc[3]};
        *reinterpret_cast<half4_t *>(dX + p * in_dim + 16 * t + 4 * g) = v;
    }
};

struct DensityIO {
    static constexpr bool kDensity = true;
    typedef feat_t in_t;  // hash-grid features / feature gradients: fp16 whatever the MLP element type
    __device__ static __forceinline__ uint64_t out_row(uint64_t p, const IoDims &d) {
        const uint32_t p32 = (uint32_t)p, r = p32 / d.T_cur;  // B is 32-bit: no 64-bit software division
        return (uint64_t)r * d.T_tot + d.slot_off + (p32 - r * d.T_cur);
    }
    // features k0..k0+7 = levels k0/2 .. k0/2+3, two channels each: four 4-byte loads from [L,B,2]
    __device__ static __forceinline__ half8_t load_x(const in_t *X, uint64_t p0, uint32_t k0, uint32_t B0, uint32_t in_dim,
                                                     const IoDims &d) {
        half8_t r;
        const uint32_t l0 = k0 >> 1;
        const uint64_t p = d.feat_rows ? out_row(p0, d) : p0;
        const uint64_t B = d.feat_rows ? d.feat_rows : B0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const feat2_t v = *reinterpret_cast<const feat2_t *>(X + ((uint64_t)(l0 + i) * B + p) * 2);
#ifdef LNH_MLP_BF16  // fp16 feature -> fp32 (exact) -> bf16 (round to nearest even)
            r[2 * i] = (half_t)(float)v[0];
            r[2 * i + 1] = (half_t)(float)v[1];
#else
            r[2 * i] = v[0];
            r[2 * i + 1] = v[1];
#endif
        }
        return r;
    }
    // features 16t+4g+r -> levels 8t+2g, 8t+2g+1
    __device__ static __forceinline__ void store_dx(in_t *dX, uint64_t p, uint32_t t, uint32_t g, uint32_t B,
                                                    uint32_t in_dim, const f32x4 &acc) {
        const uint32_t l0 = 8 * t + 2 * g;
        feat2_t a = {(feat_t)acc[0], (feat_t)acc[1]}, b = {(feat_t)acc[2], (feat_t)acc[3]};
        *reinterpret_cast<feat2_t *>(dX + ((uint64_t)l0 * B + p) * 2) = a;
        *reinterpret_cast<feat2_t *>(dX + ((uint64_t)(l0 + 1) * B + p) * 2) = b;
    }
};

}  // namespace LNH_MLP_NS
