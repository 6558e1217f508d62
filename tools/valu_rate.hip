// Issue-rate probe for a few gfx950 VALU / MFMA instructions: one wave per SIMD, a loop of 16 independent copies of the
// instruction, HIP-event time -> cycles per wave-instruction relative to v_add_f32.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ void __launch_bounds__(512) probe(float *out, int iters) {
    float v[16], w[16];
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    for (int i = 0; i < 16; i++) { v[i] = threadIdx.x * 0.001f + i; w[i] = 1.0f + i * 1e-3f; }
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    h8 ha = {1, 2, 3, 4, 5, 6, 7, 8}, hb = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int it = 0; it < iters; it++) {
        if constexpr (KIND == 0) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 1) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 2) {
#define X(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 3) {
#define X(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 4) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *)&v[(i) & 14]) : "v"(*(double *)&w[(i) & 14]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 5) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 6) {
#define X(i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[(i) & 3]) : "v"(ha), "v"(hb));
            REP16(X)
#undef X
        } else if constexpr (KIND == 7) {
#define X(i) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 8) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 9) {
#define X(i) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
            REP16(X)
#undef X
        } else if constexpr (KIND == 10) {  // dependent MFMA chain on ONE accumulator
#define X(i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(ha), "v"(hb));
            REP16(X)
#undef X
        } else if constexpr (KIND == 11) {  // MFMA + 3 independent VALU
#define X(i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %6\n\tv_add_f32 %3, %3, %6" : "+v"(acc[(i) & 3]), "+v"(v[(i) & 15]), "+v"(v[(i + 1) & 15]), "+v"(v[(i + 2) & 15]) : "v"(ha), "v"(hb), "v"(w[0]));
            REP16(X)
#undef X
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += v[i];
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static int g_threads = 256;  // 256 = one wave per SIMD, 512 = two

template <int KIND>
void run(const char *name, float *out, float ref_ms, float *ms_out) {
    const int iters = 20000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    probe<KIND><<<256, g_threads>>>(out, 100);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(a);
        probe<KIND><<<256, g_threads>>>(out, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    const double ns_per = best * 1e6 / ((double)iters * 16);
    printf("%-28s %8.3f ms  %6.2f ns per wave-instruction", name, best, ns_per);
    if (ref_ms > 0) printf("  = %5.2f x v_add_f32", best / ref_ms);
    printf("\n");
    if (ms_out) *ms_out = best;
}

int main(int argc, char **argv) {
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    float ref = 0;
    run<0>("v_add_f32", out, 0, &ref);
    run<1>("v_cvt_pk_f16_f32", out, ref, nullptr);
    run<2>("v_cvt_pkrtz_f16_f32", out, ref, nullptr);
    run<7>("v_cvt_f16_f32", out, ref, nullptr);
    run<3>("v_pk_max_f16", out, ref, nullptr);
    run<9>("v_pk_min_u16", out, ref, nullptr);
    run<8>("v_and_b32", out, ref, nullptr);
    run<4>("v_pk_mul_f32", out, ref, nullptr);
    run<5>("v_exp_f32", out, ref, nullptr);
    run<6>("mfma 16x16x32 f16 (4 acc)", out, ref, nullptr);
    run<10>("mfma 16x16x32 f16 (1 acc)", out, ref, nullptr);
    run<11>("mfma + 3 v_add (per group)", out, ref, nullptr);
    // two waves per SIMD: the same per-wave instruction streams, twice the work — equal time = perfect interleave
    g_threads = 512;
    printf("-- two waves per SIMD (time for TWICE the instructions)\n");
    run<0>("v_add_f32", out, ref, nullptr);
    run<1>("v_cvt_pk_f16_f32", out, ref, nullptr);
    run<6>("mfma 16x16x32 f16 (4 acc)", out, ref, nullptr);
    run<11>("mfma + 3 v_add (per group)", out, ref, nullptr);
    return 0;
}
