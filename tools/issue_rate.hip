// Throughput probe: VALU / LDS-atomic wave-instructions per SIMD / CU at 1 .. 8 waves per SIMD (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/issue_rate.hip -o tools/bin/issue_rate && tools/bin/issue_rate
// One workgroup per CU (256 x w threads), every wave runs a loop of 16 independent copies of one instruction.
// Prints cycles (at 2.4 GHz) per wave-instruction PER SIMD (VALU kinds) or PER CU (LDS kinds).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ void __launch_bounds__(1024) probe(unsigned *out, int iters, unsigned seed) {
    extern __shared__ unsigned long long lds[];  // 64 KiB
    unsigned v[16];
    float f[16];
    for (int i = 0; i < 16; i++) { v[i] = threadIdx.x * 2654435761u + i * 40503u + seed; f[i] = threadIdx.x * 1e-3f + i; }
    for (unsigned i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    unsigned a32[16], a64[16], cf32[16];
    for (int i = 0; i < 16; i++) {
        a32[i] = ((v[i] >> 7) & 63u) * 4u;          // 64 counters, random (the scatter pass's rank counters)
        a64[i] = ((v[i] >> 5) & 8191u) * 8u;        // 8192 rows x 8 B, random (the reduce pass's image)
        cf32[i] = (((v[i] >> 9) & 63u) * 32u + (threadIdx.x & 31u)) * 4u;   // 64 counters x 32 lane copies: bank = lane
    }
    unsigned one = 1;
    unsigned long long one64 = 1;
    for (int it = 0; it < iters; it++) {
        if constexpr (KIND == 0) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 1) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 2) {
#define X(i) asm volatile("v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(f[i]) : "v"(f[(i + 5) & 15]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 3) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *)&f[(i) & 14]) : "v"(*(double *)&f[(i + 2) & 14]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 4) {   // returning 32-bit LDS atomics on 64 random counters
#define X(i) asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(v[i]) : "v"(a32[i]), "v"(one) : "memory");
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == 5) {   // the same, conflict-free addresses (bank = lane)
#define X(i) asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(v[i]) : "v"(cf32[i]), "v"(one) : "memory");
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == 6) {   // non-returning 64-bit adds on 8192 random rows
#define X(i) asm volatile("ds_add_u64 %0, %1" : : "v"(a64[i]), "v"(one64) : "memory");
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == 7) {   // non-returning 32-bit adds on 8192 random dwords
#define X(i) asm volatile("ds_add_u32 %0, %1" : : "v"(a64[i] >> 1), "v"(one) : "memory");
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == 8) {   // non-returning f32 adds on 8192 random dwords
#define X(i) asm volatile("ds_add_f32 %0, %1" : : "v"(a64[i] >> 1), "v"(f[0]) : "memory");
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == 9) {   // non-returning f64 adds
#define X(i) asm volatile("ds_add_f64 %0, %1" : : "v"(a64[i]), "v"(one64) : "memory");
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == 10) {  // plain 64-bit LDS writes to random rows (no atomicity)
#define X(i) asm volatile("ds_write_b64 %0, %1" : : "v"(a64[i]), "v"(one64) : "memory");
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == 11) {  // 64-bit adds, lanes of a wave on CONSECUTIVE rows (conflict-free)
#define X(i) asm volatile("ds_add_u64 %0, %1" : : "v"((((threadIdx.x & 63u) + (i) * 64u + (a64[i] & 0xe000u)) & 8191u) * 8u), "v"(one64) : "memory");
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == 12) {  // packed-half adds on 8192 random dwords
#define X(i) asm volatile("ds_pk_add_f16 %0, %1" : : "v"(a64[i] >> 1), "v"(one) : "memory");
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 16; i++) s ^= v[i] ^ __builtin_bit_cast(unsigned, f[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s ^ (unsigned)lds[threadIdx.x];
}

template <int KIND>
void run(const char *name, unsigned *out, bool per_cu) {
    printf("%-44s", name);
    for (int w = 1; w <= 4; w *= 2) {   // waves per SIMD: 1, 2, 4 (1024 threads = 4 per SIMD is the workgroup limit)
        const int threads = 256 * w, iters = 4000;
        hipFuncSetAttribute((const void *)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        probe<KIND><<<256, threads, 65536>>>(out, 10, 1);
        hipDeviceSynchronize();
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        float best = 1e9;
        for (int r = 0; r < 3; r++) {
            hipEventRecord(a);
            probe<KIND><<<256, threads, 65536>>>(out, iters, r);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        // wave-instructions per SIMD (or per CU) = iters * 16 * w (* 4 for the CU)
        const double n = (double)iters * 16 * w * (per_cu ? 4 : 1);
        printf("  w=%d %6.2f cyc", w, best * 1e-3 * 2.4e9 / n);
    }
    printf("   (%s)\n", per_cu ? "per CU" : "per SIMD");
}

int main() {
    unsigned *out;
    hipMalloc(&out, 256 * 1024 * 4);
    run<0>("v_add_f32", out, false);
    run<1>("v_mul_lo_u32", out, false);
    run<2>("v_fmac_f32_dpp row_shr:1", out, false);
    run<3>("v_pk_mul_f32", out, false);
    run<4>("ds_add_rtn_u32 64 random counters", out, true);
    run<5>("ds_add_rtn_u32 conflict-free (bank = lane)", out, true);
    run<6>("ds_add_u64 8192 random rows", out, true);
    run<11>("ds_add_u64 consecutive rows", out, true);
    run<7>("ds_add_u32 random dwords", out, true);
    run<8>("ds_add_f32 random dwords", out, true);
    run<9>("ds_add_f64 random rows", out, true);
    run<12>("ds_pk_add_f16 random dwords", out, true);
    run<10>("ds_write_b64 random rows", out, true);
    return 0;
}
