// encoders.hip — frequency and spherical-harmonics direction encoders for gfx950.
//
// Frequency: semantics of lidarnerf/freqencoder/src/freqencoder.cu:34-101 (layout [x | sin(2^f x) | cos(2^f x)]_f).
//   One thread produces one (point, input-dim) column group: it reads x[d] once and emits the 1+2*deg outputs of
//   that dimension, so each sin/cos argument is formed once per frequency instead of once per output element.
//   Accurate sinf/cosf are used (the CUDA build's __sinf is a documented fast-math approximation; parity tolerance
//   in tests/test_encoders_gpu.py).
// SH: the 16 real-SH polynomials of degree <= 4 (shencoder.cu:53-89) evaluated on the RAW direction, plus the
//   analytic Jacobian (shencoder.cu:281-831 restated by differentiating the same polynomials).
#include "common.h"
#include "sh_tables.h"

namespace {

__global__ void __launch_bounds__(256)
k_freq_forward(const float *__restrict__ inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
               float *__restrict__ outputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float x = inputs[t];
    float *o = outputs + (size_t)b * C;
    o[d] = x;
    for (uint32_t f = 0; f < deg; f++) {
        const float a = scalbnf(x, (int)f);
        // cos column: the reference evaluates sin(a + fl(pi/2)); keep that argument so the two agree to rounding
        o[D + (2 * f) * D + d] = sinf(a);
        o[D + (2 * f + 1) * D + d] = sinf(a + 1.5707963267948966f);
    }
}

__global__ void __launch_bounds__(256)
k_freq_backward(const float *__restrict__ grad, const float *__restrict__ outputs, uint32_t B, uint32_t D,
                uint32_t deg, uint32_t C, float *__restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float *g = grad + (size_t)b * C;
    const float *o = outputs + (size_t)b * C;
    float r = g[d];
    for (uint32_t f = 0; f < deg; f++) {
        const uint32_t s = D + 2 * f * D + d, c = s + D;
        r += scalbnf(1.0f, (int)f) * (g[s] * o[c] - g[c] * o[s]);
    }
    grad_inputs[t] = r;
}

template <bool DYDX>
__global__ void __launch_bounds__(256)
k_sh_forward(const float *__restrict__ inputs, float *__restrict__ outputs, uint32_t B, uint32_t degree,
             float *__restrict__ dy_dx) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t C2 = degree * degree;
    const float x = inputs[(size_t)b * 3], y = inputs[(size_t)b * 3 + 1], z = inputs[(size_t)b * 3 + 2];
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    float o[16];
    float dx[16], dy[16], dz[16];
    o[0] = 0.28209479177387814f;
    dx[0] = dy[0] = dz[0] = 0.0f;
    // l = 1
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    dx[1] = 0.0f; dy[1] = -0.48860251190291992f; dz[1] = 0.0f;
    dx[2] = 0.0f; dy[2] = 0.0f; dz[2] = 0.48860251190291992f;
    dx[3] = -0.48860251190291992f; dy[3] = 0.0f; dz[3] = 0.0f;
    // l = 2
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    dx[4] = 1.0925484305920792f * y;  dy[4] = 1.0925484305920792f * x;   dz[4] = 0.0f;
    dx[5] = 0.0f;                     dy[5] = -1.0925484305920792f * z;  dz[5] = -1.0925484305920792f * y;
    dx[6] = 0.0f;                     dy[6] = 0.0f;                      dz[6] = 1.8923493915151202f * z;
    dx[7] = -1.0925484305920792f * z; dy[7] = 0.0f;                      dz[7] = -1.0925484305920792f * x;
    dx[8] = 1.0925484305920792f * x;  dy[8] = -1.0925484305920792f * y;  dz[8] = 0.0f;
    // l = 3
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    dx[9] = -3.5402615395598609f * xy;
    dy[9] = -1.7701307697799304f * x2 + 1.7701307697799304f * y2;
    dz[9] = 0.0f;
    dx[10] = 2.8906114426405538f * yz; dy[10] = 2.8906114426405538f * xz; dz[10] = 2.8906114426405538f * xy;
    dx[11] = 0.0f;
    dy[11] = 0.45704579946446572f - 2.2852289973223288f * z2;
    dz[11] = -4.5704579946446566f * yz;
    dx[12] = 0.0f; dy[12] = 0.0f;
    dz[12] = 5.597644988851731f * z2 - 1.1195289977703462f;
    dx[13] = 0.45704579946446572f - 2.2852289973223288f * z2;
    dy[13] = 0.0f;
    dz[13] = -4.5704579946446566f * xz;
    dx[14] = 2.8906114426405538f * xz; dy[14] = -2.8906114426405538f * yz;
    dz[14] = 1.4453057213202769f * x2 - 1.4453057213202769f * y2;
    dx[15] = -1.7701307697799304f * x2 + 1.7701307697799304f * y2;
    dy[15] = 3.5402615395598609f * xy;
    dz[15] = 0.0f;
    float *out = outputs + (size_t)b * C2;
    for (uint32_t i = 0; i < C2; i++) out[i] = o[i];
    if constexpr (DYDX) {
        float *d0 = dy_dx + (size_t)b * 3 * C2;
        for (uint32_t i = 0; i < C2; i++) {
            d0[i] = dx[i];
            d0[C2 + i] = dy[i];
            d0[2 * C2 + i] = dz[i];
        }
    }
}

// Degrees 5..8 (shencoder.cu:90-300): the same functions, evaluated from their structure instead of 64 unrolled
// expressions — Y_l^m = Q_lm(z) * Re/Im (x + iy)^|m| with the polynomial coefficients of sh_tables.h (generated with
// exact arithmetic by gen_sh_tables.py).  Identical polynomials in (x, y, z) as the reference's, also off the unit sphere.
template <bool DYDX>
__global__ void __launch_bounds__(256)
k_sh_forward_generic(const float *__restrict__ inputs, float *__restrict__ outputs, uint32_t B, uint32_t degree,
                     float *__restrict__ dy_dx) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t C2 = degree * degree;
    const float x = inputs[(size_t)b * 3], y = inputs[(size_t)b * 3 + 1], z = inputs[(size_t)b * 3 + 2];
    float A[kShLMax], Bm[kShLMax], zp[kShLMax];  // Re / Im of (x+iy)^m, z^k
    A[0] = 1.0f; Bm[0] = 0.0f; zp[0] = 1.0f;
#pragma unroll
    for (int m = 1; m < kShLMax; m++) {
        A[m] = x * A[m - 1] - y * Bm[m - 1];
        Bm[m] = x * Bm[m - 1] + y * A[m - 1];
        zp[m] = zp[m - 1] * z;
    }
    float *out = outputs + (size_t)b * C2;
    float *d0 = DYDX ? dy_dx + (size_t)b * 3 * C2 : nullptr;
    for (uint32_t l = 0; l < degree; l++) {
        for (uint32_t am = 0; am <= l; am++) {
            float q = 0.0f, dq = 0.0f;
            for (uint32_t k = 0; k + am <= l; k++) {
                const float c = kShQ[l][am][k];
                q = fmaf(c, zp[k], q);
                if (k) dq = fmaf(c * (float)k, zp[k - 1], dq);
            }
            const float fm = (float)am;
            if (am == 0) {
                const uint32_t i = l * l + l;
                out[i] = q;
                if constexpr (DYDX) { d0[i] = 0.0f; d0[C2 + i] = 0.0f; d0[2 * C2 + i] = dq; }
            } else {
                const uint32_t ip = l * l + l + am, in = l * l + l - am;  // m = +am (cosine-like), m = -am (sine-like)
                out[ip] = q * A[am];
                out[in] = q * Bm[am];
                if constexpr (DYDX) {
                    d0[ip] = q * fm * A[am - 1];  d0[C2 + ip] = -q * fm * Bm[am - 1]; d0[2 * C2 + ip] = dq * A[am];
                    d0[in] = q * fm * Bm[am - 1]; d0[C2 + in] = q * fm * A[am - 1];   d0[2 * C2 + in] = dq * Bm[am];
                }
            }
        }
    }
}

// shencoder.cu:834-858 (accumulates into grad_inputs, which the caller zero-initialises)
__global__ void __launch_bounds__(256)
k_sh_backward(const float *__restrict__ grad, uint32_t B, uint32_t D, uint32_t degree,
              const float *__restrict__ dy_dx, float *__restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t / D;
    if (b >= B) return;
    const uint32_t d = t - b * D, C2 = degree * degree;
    const float *g = grad + (size_t)b * C2;
    const float *j = dy_dx + ((size_t)b * D + d) * C2;
    float r = grad_inputs[t];
    for (uint32_t ch = 0; ch < C2; ch++) r += g[ch] * j[ch];
    grad_inputs[t] = r;
}

}  // namespace

extern "C" {

int lnh_freq_encode_forward(const float *inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float *outputs,
                            lnh_stream_t stream) {
    LNH_REQUIRE(inputs && outputs, LNH_ERR_INVALID_ARG, "freq forward: null pointer");
    LNH_REQUIRE(D >= 1 && C == D + 2 * D * deg, LNH_ERR_INVALID_ARG, "freq forward: C (%u) must equal D + 2*D*deg (%u)",
                C, D + 2 * D * deg);
    if (B == 0) return LNH_OK;
    LNH_LAUNCH(k_freq_forward, dim3(div_up((uint64_t)B * D, 256)), dim3(256), 0, (hipStream_t)stream, inputs,
                       B, D, deg, C, outputs);
    return lnh_check_launch("lnh_freq_encode_forward");
}

int lnh_freq_encode_backward(const float *grad, const float *outputs, uint32_t B, uint32_t D, uint32_t deg,
                             uint32_t C, float *grad_inputs, lnh_stream_t stream) {
    LNH_REQUIRE(grad && outputs && grad_inputs, LNH_ERR_INVALID_ARG, "freq backward: null pointer");
    LNH_REQUIRE(D >= 1 && C == D + 2 * D * deg, LNH_ERR_INVALID_ARG, "freq backward: C (%u) must equal D + 2*D*deg", C);
    if (B == 0) return LNH_OK;
    LNH_LAUNCH(k_freq_backward, dim3(div_up((uint64_t)B * D, 256)), dim3(256), 0, (hipStream_t)stream, grad,
                       outputs, B, D, deg, C, grad_inputs);
    return lnh_check_launch("lnh_freq_encode_backward");
}

int lnh_sh_encode_forward(const float *inputs, float *outputs, uint32_t B, uint32_t D, uint32_t degree, float *dy_dx,
                          lnh_stream_t stream) {
    LNH_REQUIRE(inputs && outputs, LNH_ERR_INVALID_ARG, "sh forward: null pointer");
    LNH_REQUIRE(D == 3, LNH_ERR_UNSUPPORTED, "SH encoder only support input dim == 3 (got %u)", D);
    LNH_REQUIRE(degree >= 1 && degree <= 8, LNH_ERR_UNSUPPORTED, "SH encoder only supports degree in [1, 8] (got %u)",
                degree);
    if (B == 0) return LNH_OK;
    dim3 grid(div_up(B, 256)), block(256);
    if (degree > 4) {
        if (dy_dx)
            LNH_LAUNCH(k_sh_forward_generic<true>, grid, block, 0, (hipStream_t)stream, inputs, outputs, B, degree, dy_dx);
        else
            LNH_LAUNCH(k_sh_forward_generic<false>, grid, block, 0, (hipStream_t)stream, inputs, outputs, B, degree, dy_dx);
    } else if (dy_dx)
        LNH_LAUNCH(k_sh_forward<true>, grid, block, 0, (hipStream_t)stream, inputs, outputs, B, degree, dy_dx);
    else
        LNH_LAUNCH(k_sh_forward<false>, grid, block, 0, (hipStream_t)stream, inputs, outputs, B, degree, dy_dx);
    return lnh_check_launch("lnh_sh_encode_forward");
}

int lnh_sh_encode_backward(const float *grad, const float *inputs, uint32_t B, uint32_t D, uint32_t degree,
                           const float *dy_dx, float *grad_inputs, lnh_stream_t stream) {
    (void)inputs;
    LNH_REQUIRE(grad && dy_dx && grad_inputs, LNH_ERR_INVALID_ARG, "sh backward: null pointer");
    LNH_REQUIRE(D == 3 && degree >= 1 && degree <= 8, LNH_ERR_UNSUPPORTED, "sh backward: D must be 3, degree 1..8");
    if (B == 0) return LNH_OK;
    LNH_LAUNCH(k_sh_backward, dim3(div_up((uint64_t)B * D, 256)), dim3(256), 0, (hipStream_t)stream, grad, B, D,
                       degree, dy_dx, grad_inputs);
    return lnh_check_launch("lnh_sh_encode_backward");
}

}  // extern "C"
