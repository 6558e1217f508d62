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


// ------------------------------------------------------------------------------------------------ the whole optimizer step
// Round 5: everything the inner loop of train_one_epoch does after backward() (nerf/utils.py:1216-1226: scaler.step(optimizer),
// scaler.update(), lr_scheduler.step()) as TWO launches for ALL parameters — the hash table and the handful of small fp32
// tensors (the MLP weights).  Rounds 2-4 left the small tensors to torch: a zero fill of found_inf, a reciprocal, the
// foreach unscale / finite check, three launches of the capturable fused Adam, the loss-scale update, two scalar copies and
// the scheduler's fill of the learning-rate scalar — eleven launches of ~5 us for 21 k parameters.
//
// Scalars of the optimizer live in one device buffer (`state`, LNH_TRAIN_STATE_FLOATS floats), so that a captured step
// needs no host value: the loss scale and its growth counter (GradScaler), the Adam step count, the scheduler's step count
// and the learning rate lr0 * 0.1^min(it / iters, 1) (main_lidarnerf.py:408-410), formed on the device.
//   k_train_check   finite check of the fp16 table gradient and the small fp32 gradients; thread 0 first COMMITS the counters
//                   the previous step left in their `next` slots, then forms 1 / scale and the learning rate.  An inf / nan
//                   is recorded as a STAMP (it + 1) instead of a flag: a stale stamp of an earlier step is never equal to
//                   it + 1, so nothing ever has to be zeroed between steps, and a MAX all-reduce over ranks keeps it.
//   k_train_step    Adam on the table (as k_adam_table) in the first blocks, on the small tensors in the last ones;
//                   thread 0 of block 0 writes the `next` counters and the new loss scale (nobody reads those in this
//                   launch: every block works from the committed values and from 1 / scale).
enum { TS_SCALE = LNH_TS_SCALE, TS_GROWTH = LNH_TS_GROWTH, TS_FOUND = LNH_TS_FOUND, TS_INV = LNH_TS_INV,
       TS_INV_TABLE = LNH_TS_INV_TABLE, TS_LAST_SCALE = LNH_TS_LAST_SCALE, TS_T = LNH_TS_T, TS_IT = LNH_TS_IT, TS_LR = LNH_TS_LR,
       TS_T_NEXT = LNH_TS_T_NEXT, TS_IT_NEXT = LNH_TS_IT_NEXT, TS_SKIPPED = LNH_TS_SKIPPED };
static_assert(TS_SKIPPED < LNH_TRAIN_STATE_FLOATS, "state layout");

struct SmallSet {
    float *p[LNH_TRAIN_MAX_SMALL];
    const float *g[LNH_TRAIN_MAX_SMALL];
    uint32_t n[LNH_TRAIN_MAX_SMALL], off[LNH_TRAIN_MAX_SMALL];  // elements, offset into the flat moment buffers
    uint32_t count;
};

__global__ void __launch_bounds__(256)
k_train_check(float *__restrict__ state, const uint4 *__restrict__ g, uint64_t n_vec, const half_t *__restrict__ tail,
              uint32_t n_tail, SmallSet s, float div_table, float div_small, double lr0, double iters) {
    const float it = state[TS_IT_NEXT];  // (the committed value: thread 0 copies it to TS_IT below, nobody else writes it here)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        state[TS_T] = state[TS_T_NEXT];
        state[TS_IT] = it;
        const float scale = state[TS_SCALE], inv = 1.0f / scale;
        state[TS_INV] = inv / div_small;
        state[TS_INV_TABLE] = inv / div_table;
        state[TS_LAST_SCALE] = scale;
        const double x = (double)it / iters;
        state[TS_LR] = (float)(lr0 * pow(0.1, x < 1.0 ? x : 1.0));
    }
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
    // the small gradients (fp32: exponent field all ones), dealt to all blocks — a single block walking their ~21 k values
    // one dependent round trip after the other took 25 us
    for (uint32_t k = 0; k < s.count; k++) {
        if (!s.g[k]) continue;
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < s.n[k]; i += gridDim.x * blockDim.x)
            bad |= (__builtin_bit_cast(uint32_t, s.g[k][i]) & 0x7f800000u) == 0x7f800000u;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) state[TS_FOUND] = it + 1.0f;  // benign race: every writer stores the same value
}

struct TrainStepArgs {
    float *p, *m, *v;  // table (n == 0: the table is stepped elsewhere, e.g. shard by shard)
    const half_t *g16;
    half_t *p16;
    uint64_t n;
    SmallSet s;
    float *sm, *sv;
    float *state;
    double beta1, beta2, eps;
    float growth, backoff;
    uint32_t interval, table_blocks;
};

__global__ void __launch_bounds__(256)
k_train_step(TrainStepArgs a) {
    const float it = a.state[TS_IT], t0 = a.state[TS_T];
    const bool skip = a.state[TS_FOUND] == it + 1.0f;
    const float t = t0 + 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.state[TS_T_NEXT] = skip ? t0 : t;
        a.state[TS_IT_NEXT] = it + 1.0f;
        a.state[TS_SKIPPED] = skip ? 1.0f : 0.0f;
        // torch's amp_update_scale_ (growth 2, backoff 0.5, interval 2000 at the call site)
        float scale = a.state[TS_SCALE], grown = a.state[TS_GROWTH];
        if (skip) {
            scale *= a.backoff;
            grown = 0.0f;
        } else if (grown + 1.0f == (float)a.interval) {
            const float ns = scale * a.growth;
            if (isfinite(ns)) scale = ns;
            grown = 0.0f;
        } else {
            grown += 1.0f;
        }
        a.state[TS_SCALE] = scale;
        a.state[TS_GROWTH] = grown;
    }
    if (skip) return;
    const double bc1 = 1.0 - pow(a.beta1, (double)t), bc2 = 1.0 - pow(a.beta2, (double)t);
    const double w1 = 1.0 - a.beta1, w2 = 1.0 - a.beta2;
    const double lr = (double)a.state[TS_LR];
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2), eps = (float)a.eps;
    auto update = [&](float &p, float &m, float &v, float g) {  // (the arithmetic of k_adam_table, see there)
        m = (float)((double)m + w1 * ((double)g - (double)m));
        v = (float)(a.beta2 * (double)v + w2 * (double)g * (double)g);
        p -= step_size * m / (sqrtf(v) / bc2_sqrt + eps);
    };
    if (blockIdx.x < a.table_blocks) {
        const float inv = a.state[TS_INV_TABLE];
        const uint64_t n4 = a.n / 4;
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)a.table_blocks * blockDim.x) {
            float4 p = reinterpret_cast<float4 *>(a.p)[i], m = reinterpret_cast<float4 *>(a.m)[i],
                   v = reinterpret_cast<float4 *>(a.v)[i];
            const half4_t gh = reinterpret_cast<const half4_t *>(a.g16)[i];
            float *pp = &p.x, *mm = &m.x, *vv = &v.x;
            half4_t ph;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                update(pp[k], mm[k], vv[k], (float)gh[k] * inv);
                ph[k] = (half_t)pp[k];
            }
            reinterpret_cast<float4 *>(a.p)[i] = p;
            reinterpret_cast<float4 *>(a.m)[i] = m;
            reinterpret_cast<float4 *>(a.v)[i] = v;
            reinterpret_cast<half4_t *>(a.p16)[i] = ph;
        }
        if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {  // tail
            const uint64_t i = n4 * 4 + threadIdx.x;
            update(a.p[i], a.m[i], a.v[i], (float)a.g16[i] * inv);
            a.p16[i] = (half_t)a.p[i];
        }
        return;
    }
    // small tensors: the blocks behind the table's walk all of them together
    const float inv = a.state[TS_INV];
    const uint32_t nb = gridDim.x - a.table_blocks, b = blockIdx.x - a.table_blocks;
    for (uint32_t k = 0; k < a.s.count; k++) {
        if (!a.s.g[k]) continue;  // (no gradient this step: torch's Adam skips such a parameter too)
        for (uint32_t i = b * blockDim.x + threadIdx.x; i < a.s.n[k]; i += nb * blockDim.x)
            update(a.s.p[k][i], a.sm[a.s.off[k] + i], a.sv[a.s.off[k] + i], a.s.g[k][i] * inv);
    }
}

// up to 8 regions cleared by one launch (zero fills as kernels: see lnh_zero_async in common.h)
struct ZeroRegions {
    uint32_t *p[8];
    uint64_t words[8], first_block[9];
    uint32_t count;
};
__global__ void __launch_bounds__(256)
k_zero_regions(ZeroRegions z) {
    uint32_t r = 0;
    while (r + 1 < z.count && blockIdx.x >= z.first_block[r + 1]) r++;
    const uint64_t b = blockIdx.x - z.first_block[r], nb = z.first_block[r + 1] - z.first_block[r];
    uint32_t *p = z.p[r];
    const uint64_t n = z.words[r];
    // 16-byte stores over the aligned middle, single words at the ends
    const uint64_t head = (uint64_t)((16 - ((uintptr_t)p & 15)) & 15) / 4;
    const uint64_t h = head < n ? head : n, n4 = (n - h) / 4;
    uint4 *q = reinterpret_cast<uint4 *>(p + h);
    for (uint64_t i = b * 256 + threadIdx.x; i < n4; i += nb * 256) q[i] = make_uint4(0u, 0u, 0u, 0u);
    if (b == 0) {
        if (threadIdx.x < h) p[threadIdx.x] = 0u;
        const uint64_t done = h + n4 * 4;
        if (threadIdx.x < n - done) p[done + threadIdx.x] = 0u;
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


int lnh_zero_regions(void *const *ptrs, const uint64_t *bytes, uint32_t count, lnh_stream_t stream) {
    LNH_REQUIRE(count <= 8 && (count == 0 || (ptrs && bytes)), LNH_ERR_INVALID_ARG, "zero_regions: at most 8 regions");
    ZeroRegions z{};
    uint64_t blocks = 0;
    for (uint32_t i = 0; i < count; i++) {
        if (bytes[i] == 0) continue;
        LNH_REQUIRE(ptrs[i] && (bytes[i] & 3) == 0 && ((uintptr_t)ptrs[i] & 3) == 0, LNH_ERR_INVALID_ARG,
                    "zero_regions: region %u needs a 4-byte aligned pointer and size", i);
        z.p[z.count] = (uint32_t *)ptrs[i];
        z.words[z.count] = bytes[i] / 4;
        z.first_block[z.count] = blocks;
        const uint64_t want = (bytes[i] / 16 + 1023) / 1024;  // ~4 stores of 16 bytes per thread
        blocks += want < 1 ? 1 : (want > 1024 ? 1024 : want);
        z.count++;
    }
    if (z.count == 0) return LNH_OK;
    z.first_block[z.count] = blocks;
    LNH_LAUNCH(k_zero_regions, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, z);
    return lnh_check_launch("lnh_zero_regions");
}

static int fill_small(SmallSet &s, float *const *params, const float *const *grads, const uint32_t *numel, uint32_t count,
                      bool need_params, const char *who) {
    LNH_REQUIRE(count <= LNH_TRAIN_MAX_SMALL && (count == 0 || (grads && numel && (params || !need_params))),
                LNH_ERR_INVALID_ARG, "%s: at most %d small tensors", who, LNH_TRAIN_MAX_SMALL);
    uint32_t off = 0;
    s.count = count;
    for (uint32_t k = 0; k < count; k++) {
        LNH_REQUIRE(!need_params || params[k], LNH_ERR_INVALID_ARG, "%s: null parameter pointer %u", who, k);
        s.p[k] = need_params ? params[k] : nullptr;
        s.g[k] = grads[k];
        s.n[k] = numel[k];
        s.off[k] = off;
        off += numel[k];
    }
    return LNH_OK;
}

int lnh_train_check(float *state, const void *grad16, uint64_t n16, const float *const *small_grads,
                    const uint32_t *small_numel, uint32_t n_small, float div_table, float div_small, double lr0,
                    double iters, lnh_stream_t stream) {
    LNH_REQUIRE(state && (grad16 || n16 == 0), LNH_ERR_INVALID_ARG, "train_check: null pointer");
    LNH_REQUIRE(((uintptr_t)grad16 & 15) == 0, LNH_ERR_INVALID_ARG, "train_check: gradient must be 16-byte aligned");
    LNH_REQUIRE(div_table > 0.0f && div_small > 0.0f && iters > 0.0, LNH_ERR_INVALID_ARG, "train_check: divisors and iters must be > 0");
    SmallSet s{};
    if (int rc = fill_small(s, nullptr, small_grads, small_numel, n_small, false, "train_check")) return rc;
    const uint64_t n_vec = n16 / 8;
    const uint32_t blocks = (uint32_t)((n_vec + 255) / 256 < 2048 ? (n_vec + 255) / 256 + 1 : 2048);
    LNH_LAUNCH(k_train_check, dim3(blocks), dim3(256), 0, (hipStream_t)stream, state, (const uint4 *)grad16, n_vec,
               (const half_t *)grad16 + n_vec * 8, (uint32_t)(n16 & 7), s, div_table, div_small, lr0, iters);
    return lnh_check_launch("lnh_train_check");
}

int lnh_train_step(float *state, float *param, float *exp_avg, float *exp_avg_sq, const void *grad16, void *param16,
                   uint64_t n, float *const *small_params, const float *const *small_grads, const uint32_t *small_numel,
                   uint32_t n_small, float *small_exp_avg, float *small_exp_avg_sq, double beta1, double beta2, double eps,
                   double growth_factor, double backoff_factor, uint32_t growth_interval, lnh_stream_t stream) {
    LNH_REQUIRE(state, LNH_ERR_INVALID_ARG, "train_step: null state");
    LNH_REQUIRE(n == 0 || (param && exp_avg && exp_avg_sq && grad16 && param16), LNH_ERR_INVALID_ARG, "train_step: null table pointer");
    LNH_REQUIRE((((uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0 &&
                    (((uintptr_t)grad16 | (uintptr_t)param16) & 7) == 0,
                LNH_ERR_INVALID_ARG, "train_step: buffers must be 16-byte (fp32) / 8-byte (fp16) aligned");
    LNH_REQUIRE(n_small == 0 || (small_exp_avg && small_exp_avg_sq), LNH_ERR_INVALID_ARG, "train_step: null moment buffer");
    TrainStepArgs a{};
    if (int rc = fill_small(a.s, small_params, small_grads, small_numel, n_small, true, "train_step")) return rc;
    a.p = param; a.m = exp_avg; a.v = exp_avg_sq; a.g16 = (const half_t *)grad16; a.p16 = (half_t *)param16; a.n = n;
    a.sm = small_exp_avg; a.sv = small_exp_avg_sq; a.state = state;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.growth = (float)growth_factor; a.backoff = (float)backoff_factor; a.interval = growth_interval;
    const uint64_t n4 = n / 4;
    a.table_blocks = n ? (uint32_t)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 + 1 : 4096) : 1;  // (block 0 keeps the books)
    uint32_t small_total = 0;
    for (uint32_t k = 0; k < n_small; k++) small_total = small_total > small_numel[k] ? small_total : small_numel[k];
    const uint32_t small_blocks = n_small ? (small_total + 255) / 256 : 0;
    LNH_LAUNCH(k_train_step, dim3(a.table_blocks + (small_blocks > 64 ? 64 : small_blocks)), dim3(256), 0, (hipStream_t)stream, a);
    return lnh_check_launch("lnh_train_step");
}

}  // extern "C"
