// common.h — shared host/device helpers of liblidarnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/lidarnerf_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));

#define LNH_WAVE 64

// Name of this build (lnh_build_variant): tools/probe_variants.py compiles its timing probes with -DLNH_VARIANT_TAG="<name>"
#ifndef LNH_VARIANT_TAG
#define LNH_VARIANT_TAG "product"
#endif
// Section marks for tools/isa_sections.py (an assembly comment per mark in -DLNH_ISA_MARKS builds, nothing otherwise)
#ifdef LNH_ISA_MARKS
#define LNH_MARK(name) asm volatile("; MARK " name)
#else
#define LNH_MARK(name) do { } while (0)
#endif

// thread-local last-error string (lnh_last_error)
void lnh_set_error(const char *fmt, ...);
int lnh_cu_count();  // compute units of the current device (cached per device); 256 on MI355X

#define LNH_REQUIRE(cond, code, ...)     \
    do {                                 \
        if (!(cond)) {                   \
            lnh_set_error(__VA_ARGS__);  \
            return (code);               \
        }                                \
    } while (0)

static inline int lnh_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        lnh_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return LNH_ERR_LAUNCH;
    }
    return LNH_OK;
}

// Launch with a clean error slot: hipGetLastError() is sticky per thread, so an unrelated earlier failure (e.g. a
// runtime probe made by another library) must not be reported as a failure of this launch.
#define LNH_LAUNCH(...)                      \
    do {                                     \
        (void)hipGetLastError();             \
        hipLaunchKernelGGL(__VA_ARGS__);     \
    } while (0)

static inline uint32_t div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// Zero fill as a KERNEL, not hipMemsetAsync.  A training step captured in a hipGraph (LidarTrainer graph mode) turns every
// hipMemsetAsync into a memset node, and on this runtime (ROCm 7.2) such nodes were seen to replay with a wrong fill byte:
// the 4-byte clear of the loss accumulator came back as 0xF0F0F0F0 / 0x70707070 / 0xD0D0D0D0 plus the sum, in one captured
// graph out of a few, every replay of that graph (tests/test_occupancy_gpu.py caught it).  A kernel node has no such field.
static __global__ void __launch_bounds__(256) k_lnh_zero_words(uint32_t *__restrict__ p, uint64_t n_words) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * 256) p[i] = 0u;
}
static inline int lnh_zero_async(void *p, uint64_t bytes, hipStream_t s, const char *what) {
    if (bytes == 0) return LNH_OK;
    if ((bytes & 3) || ((uintptr_t)p & 3)) {
        lnh_set_error("%s: zero fill needs 4-byte alignment", what);
        return LNH_ERR_INVALID_ARG;
    }
    const uint64_t n = bytes / 4;
    const uint32_t blocks = (uint32_t)(n / 256 < 1024 ? n / 256 + 1 : 1024);
    LNH_LAUNCH(k_lnh_zero_words, dim3(blocks), dim3(256), 0, s, (uint32_t *)p, n);
    return lnh_check_launch(what);
}

// ---- wave-level primitives (64 lanes) ------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive scan (sum) across the 64 lanes of a wave
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
// inclusive scan (product)
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o, 64);
        if (lane >= o) v *= t;
    }
    return v;
}

// inclusive scan (sum) of a 32-bit integer across the 64 lanes on the DPP network (row_shr x4, row_bcast:15, :31)
__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// ---- segmented inclusive add-scan over the 64 lanes on the DPP network (6 VALU ops, no LDS/bpermute traffic).
// `run_start` = first lane of the run this lane belongs to (runs are contiguous lane ranges).  After the four
// row_shr steps a lane holds the sum over [max(run_start, row_start) .. lane]; row_bcast:15 / :31 carry the partial
// sums of the preceding 16- / 32-lane blocks into lanes whose run started before their own block.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float lnh_dpp(float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_segscan_add(float v, int lane, int run_start) {
    float t;
    t = lnh_dpp<0x111, 0xf>(v); if (lane - 1 >= run_start) v += t;  // row_shr:1 (0 shifted in at row starts)
    t = lnh_dpp<0x112, 0xf>(v); if (lane - 2 >= run_start) v += t;
    t = lnh_dpp<0x114, 0xf>(v); if (lane - 4 >= run_start) v += t;
    t = lnh_dpp<0x118, 0xf>(v); if (lane - 8 >= run_start) v += t;
    t = lnh_dpp<0x142, 0xa>(v); if ((lane & 16) && run_start < (lane & ~15)) v += t;  // row_bcast:15 -> rows 1,3
    t = lnh_dpp<0x143, 0xc>(v); if (lane >= 32 && run_start < 32) v += t;             // row_bcast:31 -> rows 2,3
    return v;
}

// Same scan for MANY values of one lane: the six "take" decisions depend only on (lane, run_start), so they are
// turned into 0/1 multipliers once and every step becomes ONE v_fmac_f32 with a DPP source operand.
struct SegScanMask {
    float m[6];
};
__device__ __forceinline__ SegScanMask wave_segscan_mask(int lane, int run_start) {
    SegScanMask k;
    k.m[0] = (lane - 1 >= run_start) ? 1.0f : 0.0f;
    k.m[1] = (lane - 2 >= run_start) ? 1.0f : 0.0f;
    k.m[2] = (lane - 4 >= run_start) ? 1.0f : 0.0f;
    k.m[3] = (lane - 8 >= run_start) ? 1.0f : 0.0f;
    k.m[4] = ((lane & 16) && run_start < (lane & ~15)) ? 1.0f : 0.0f;
    k.m[5] = (lane >= 32 && run_start < 32) ? 1.0f : 0.0f;
    return k;
}
// Row-local variant: runs never cross a 16-lane DPP row (the caller starts a new run at every row start), so the scan
// is the four row_shr steps only — pure VALU, no cross-row broadcast.
__device__ __forceinline__ float row_segscan_add(float v, const SegScanMask &k) {
    v = fmaf(lnh_dpp<0x111, 0xf>(v), k.m[0], v);
    v = fmaf(lnh_dpp<0x112, 0xf>(v), k.m[1], v);
    v = fmaf(lnh_dpp<0x114, 0xf>(v), k.m[2], v);
    v = fmaf(lnh_dpp<0x118, 0xf>(v), k.m[3], v);
    return v;
}
// The same two scans over N values at once, as explicit v_fmac_f32_dpp: ONE instruction per value and step (the
// compiler's lowering of the intrinsic form is v_mov_b32 + v_mov_b32_dpp + half a v_pk_fma_f32).  Steps are emitted
// value-interleaved, so the 2-wait-state "VALU write -> DPP read" hazard of a value's next step is covered by the
// other values' instructions (N >= 3); the leading s_nop covers the producers of the first step.
template <int N>
__device__ __forceinline__ void row_segscan_add_n(float (&v)[N], const SegScanMask &k) {
    static_assert(N >= 3, "needs >= 2 independent instructions between the steps of one value");
    asm volatile("s_nop 1");
#pragma unroll
    for (int i = 0; i < N; i++)
        asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v[i]) : "v"(k.m[0]));
#pragma unroll
    for (int i = 0; i < N; i++)
        asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v[i]) : "v"(k.m[1]));
#pragma unroll
    for (int i = 0; i < N; i++)
        asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v[i]) : "v"(k.m[2]));
#pragma unroll
    for (int i = 0; i < N; i++)
        asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v[i]) : "v"(k.m[3]));
}
template <int N>
__device__ __forceinline__ void cross_segscan_add_n(float (&v)[N], const SegScanMask &k) {
    static_assert(N >= 3, "needs >= 2 independent instructions between the steps of one value");
    asm volatile("s_nop 1");
#pragma unroll
    for (int i = 0; i < N; i++)
        asm volatile("v_fmac_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v[i]) : "v"(k.m[4]));
#pragma unroll
    for (int i = 0; i < N; i++)
        asm volatile("v_fmac_f32_dpp %0, %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v[i]) : "v"(k.m[5]));
}
// ONE step of the segmented scan over N values (v_fmac_f32_dpp, value-interleaved like row_segscan_add_n): S = 0..3 are the
// row_shr:1/2/4/8 steps, 4 / 5 the row_bcast:15 / :31 steps.  `take` = 1.0f where the lane adds the shifted value, else 0.0f.
// Callers run only the steps the longest run of the wave needs.
template <int S, int N>
__device__ __forceinline__ void segscan_step_n(float (&v)[N], float take) {
    static_assert(N >= 3, "needs >= 2 independent instructions between the steps of one value");
    asm volatile("s_nop 1");
#pragma unroll
    for (int i = 0; i < N; i++) {
        if constexpr (S == 0)
            asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v[i]) : "v"(take));
        else if constexpr (S == 1)
            asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v[i]) : "v"(take));
        else if constexpr (S == 2)
            asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v[i]) : "v"(take));
        else if constexpr (S == 3)
            asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(v[i]) : "v"(take));
        else if constexpr (S == 4)
            asm volatile("v_fmac_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v[i]) : "v"(take));
        else
            asm volatile("v_fmac_f32_dpp %0, %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v[i]) : "v"(take));
    }
}
// the two cross-row steps that complete row_segscan_add to the full-wave scan
__device__ __forceinline__ float cross_segscan_add(float v, const SegScanMask &k) {
    v = fmaf(lnh_dpp<0x142, 0xa>(v), k.m[4], v);
    v = fmaf(lnh_dpp<0x143, 0xc>(v), k.m[5], v);
    return v;
}
__device__ __forceinline__ float wave_segscan_add(float v, const SegScanMask &k) {
    v = fmaf(lnh_dpp<0x111, 0xf>(v), k.m[0], v);
    v = fmaf(lnh_dpp<0x112, 0xf>(v), k.m[1], v);
    v = fmaf(lnh_dpp<0x114, 0xf>(v), k.m[2], v);
    v = fmaf(lnh_dpp<0x118, 0xf>(v), k.m[3], v);
    v = fmaf(lnh_dpp<0x142, 0xa>(v), k.m[4], v);
    v = fmaf(lnh_dpp<0x143, 0xc>(v), k.m[5], v);
    return v;
}
