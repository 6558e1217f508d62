// optim.hip — the optimizer step of the hash table as ONE pass.
// The reference's loop (nerf/utils.py:1206-1226: scaler.scale(loss).backward(); scaler.step(Adam); scaler.update())
// spends, on the 13.7 M-parameter table, a gradient cast fp16->fp32, GradScaler's unscale pass, the Adam kernel and
// next step's fp32->fp16 cast of the table: four trips over 55-220 MB.  Here: one finite check of the fp16 gradient
// (27 MB) and one kernel that reads g16, p, m, v and writes p, m, v and the fp16 copy the encode kernels read.
// Arithmetic = torch.optim.Adam (no weight decay, no amsgrad) as its fused CUDA kernel does it in fp32:
//   g = g16 * inv_scale;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// and GradScaler's contract: when found_inf != 0 nothing is written and the step counter does not advance.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256)
k_grad_check_f16(const uint4 *__restrict__ g, uint64_t n_vec, const half_t *__restrict__ tail, uint32_t n_tail,
                 float *__restrict__ found_inf) {
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 q = g[i];
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; k++)  // fp16 inf / nan: exponent field all ones
            bad |= ((w[k] & 0x7c00u) == 0x7c00u) || ((w[k] & 0x7c000000u) == 0x7c000000u);
    }
    if (blockIdx.x == 0 && threadIdx.x < n_tail)
        bad |= (__builtin_bit_cast(unsigned short, tail[threadIdx.x]) & 0x7c00u) == 0x7c00u;
    if (__any(bad) && (threadIdx.x & 63) == 0) *found_inf = 1.0f;  // benign race: every writer stores the same value
}

struct AdamArgs {
    float *p, *m, *v;
    const half_t *g16;
    half_t *p16;
    uint64_t n;
    double lr, beta1, beta2, eps;
    const float *inv_scale, *found_inf, *step_in;
    float *step_out;
    const float *lr_dev;  // non-null: the learning rate is read from device memory (a step captured in a hipGraph)
};

__global__ void __launch_bounds__(256)
k_adam_table(AdamArgs a) {
    const bool skip = *a.found_inf != 0.0f;
    const float t = *a.step_in + 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.step_out = skip ? *a.step_in : t;  // double-buffered step counter
    if (skip) return;
    const float inv = *a.inv_scale;
    // torch's kernel receives lr / betas / eps as doubles and lets them promote the fp32 state to double inside each
    // expression.  The moment updates do the same here (multiply-adds only: free), so m and v agree to the last bit or
    // two; the parameter update keeps fp32 for its square root and divisions (double versions of those cost more than
    // the kernel's memory time) and differs from torch's by an ulp of the parameter.
    const double bc1 = 1.0 - pow(a.beta1, (double)t), bc2 = 1.0 - pow(a.beta2, (double)t);
    const double w1 = 1.0 - a.beta1, w2 = 1.0 - a.beta2;
    const double lr = a.lr_dev ? (double)*a.lr_dev : a.lr;
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2), eps = (float)a.eps;
    auto update = [&](float &p, float &m, float &v, half_t gh) {
        const float g = (float)gh * inv;
        m = (float)((double)m + w1 * ((double)g - (double)m));  // lerp(m, g, 1 - beta1), |weight| < 0.5 branch
        v = (float)(a.beta2 * (double)v + w2 * (double)g * (double)g);
        p -= step_size * m / (sqrtf(v) / bc2_sqrt + eps);
        return (half_t)p;
    };
    const uint64_t n4 = a.n / 4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
        float4 p = reinterpret_cast<float4 *>(a.p)[i], m = reinterpret_cast<float4 *>(a.m)[i],
               v = reinterpret_cast<float4 *>(a.v)[i];
        const half4_t gh = reinterpret_cast<const half4_t *>(a.g16)[i];
        float *pp = &p.x, *mm = &m.x, *vv = &v.x;
        half4_t ph;
#pragma unroll
        for (int k = 0; k < 4; k++) ph[k] = update(pp[k], mm[k], vv[k], gh[k]);
        reinterpret_cast<float4 *>(a.p)[i] = p;
        reinterpret_cast<float4 *>(a.m)[i] = m;
        reinterpret_cast<float4 *>(a.v)[i] = v;
        reinterpret_cast<half4_t *>(a.p16)[i] = ph;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {  // tail
        const uint64_t i = n4 * 4 + threadIdx.x;
        a.p16[i] = update(a.p[i], a.m[i], a.v[i], a.g16[i]);
    }
}

}  // namespace

extern "C" {

int lnh_grad_check_f16(const void *grad16, uint64_t n, float *found_inf, lnh_stream_t stream) {
    LNH_REQUIRE(grad16 && found_inf, LNH_ERR_INVALID_ARG, "grad_check_f16: null pointer");
    LNH_REQUIRE(((uintptr_t)grad16 & 15) == 0, LNH_ERR_INVALID_ARG, "grad_check_f16: gradient must be 16-byte aligned");
    if (n == 0) return LNH_OK;
    const uint64_t n_vec = n / 8;
    const uint32_t blocks = (uint32_t)((n_vec + 255) / 256 < 2048 ? (n_vec + 255) / 256 + 1 : 2048);
    LNH_LAUNCH(k_grad_check_f16, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4 *)grad16, n_vec,
               (const half_t *)grad16 + n_vec * 8, (uint32_t)(n & 7), found_inf);
    return lnh_check_launch("lnh_grad_check_f16");
}

static int adam_table_step(float *param, float *exp_avg, float *exp_avg_sq, const void *grad16, void *param16, uint64_t n,
                           double lr, const float *lr_dev, double beta1, double beta2, double eps, const float *inv_scale,
                           const float *found_inf, const float *step_in, float *step_out, lnh_stream_t stream) {
    LNH_REQUIRE(param && exp_avg && exp_avg_sq && grad16 && param16 && inv_scale && found_inf && step_in && step_out,
                LNH_ERR_INVALID_ARG, "adam_table_step: null pointer");
    LNH_REQUIRE(step_in != step_out, LNH_ERR_INVALID_ARG, "adam_table_step: the step counter is double-buffered");
    LNH_REQUIRE((((uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0 &&
                    (((uintptr_t)grad16 | (uintptr_t)param16) & 7) == 0,
                LNH_ERR_INVALID_ARG, "adam_table_step: buffers must be 16-byte (fp32) / 8-byte (fp16) aligned");
    if (n == 0) return LNH_OK;
    AdamArgs a{param, exp_avg, exp_avg_sq, (const half_t *)grad16, (half_t *)param16, n, lr, beta1, beta2, eps,
               inv_scale, found_inf, step_in, step_out, lr_dev};
    const uint64_t n4 = n / 4;
    const uint32_t blocks = (uint32_t)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 + 1 : 4096);
    LNH_LAUNCH(k_adam_table, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return lnh_check_launch("lnh_adam_table_step");
}

int lnh_adam_table_step(float *param, float *exp_avg, float *exp_avg_sq, const void *grad16, void *param16, uint64_t n,
                        double lr, double beta1, double beta2, double eps, const float *inv_scale,
                        const float *found_inf, const float *step_in, float *step_out, lnh_stream_t stream) {
    return adam_table_step(param, exp_avg, exp_avg_sq, grad16, param16, n, lr, nullptr, beta1, beta2, eps, inv_scale,
                           found_inf, step_in, step_out, stream);
}

int lnh_adam_table_step_dlr(float *param, float *exp_avg, float *exp_avg_sq, const void *grad16, void *param16, uint64_t n,
                            const float *lr, double beta1, double beta2, double eps, const float *inv_scale,
                            const float *found_inf, const float *step_in, float *step_out, lnh_stream_t stream) {
    LNH_REQUIRE(lr, LNH_ERR_INVALID_ARG, "adam_table_step_dlr: null learning-rate pointer");
    return adam_table_step(param, exp_avg, exp_avg_sq, grad16, param16, n, 0.0, lr, beta1, beta2, eps, inv_scale,
                           found_inf, step_in, step_out, stream);
}

}  // extern "C"
