// Flat-arena gradient clipping + AdamW for gfx950: three HBM-bound launches per step instead of
// ~200 per-tensor launches; no host sync (the clip coefficient, the loss==0 skip flag and the step
// counter live in device memory).  Replaces train.py:316-321.
#include "common.h"
#include "mi355det_optim.h"

using namespace mi355;

namespace {
struct OptState { int step; float clip_coef; float skip; float grad_norm; };

// Elements the reference's optimizer never touches: parameters with requires_grad == False and parameters that
// no forward uses (`p.grad is None`: torch's clip_grad_norm_ and AdamW skip them, train.py:320-321), as sorted,
// disjoint arena ranges [lo, hi).  `true` = element i is skipped.
__device__ __forceinline__ bool in_ranges(const long long* __restrict__ r, int n, long long i) {
    int lo = 0, hi = n;                          // first range with hi > i
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (r[2 * mid + 1] > i) hi = mid; else lo = mid + 1;
    }
    return lo < n && r[2 * lo] <= i;
}
}

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float4* __restrict__ g4, const float* __restrict__ g,
                                                         long long n, float* __restrict__ partials,
                                                         const long long* __restrict__ skip, int nskip) {
    __shared__ float red[4];
    const long long n4 = n >> 2;
    float acc = 0.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 v = g4[i];
        if (nskip > 0) {
            if (in_ranges(skip, nskip, 4 * i)) v.x = 0.0f;
            if (in_ranges(skip, nskip, 4 * i + 1)) v.y = 0.0f;
            if (in_ranges(skip, nskip, 4 * i + 2)) v.z = 0.0f;
            if (in_ranges(skip, nskip, 4 * i + 3)) v.w = 0.0f;
        }
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long j = (n4 << 2) + threadIdx.x;
        const float t = (nskip > 0 && in_ranges(skip, nskip, j)) ? 0.0f : g[j];
        acc += t * t;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

extern "C" int mi355_grad_sumsq(const float* grad, int64_t n, float* partials, void* stream) {
    return mi355_grad_sumsq_masked(grad, n, partials, nullptr, 0, stream);
}

extern "C" int mi355_grad_sumsq_masked(const float* grad, int64_t n, float* partials, const int64_t* skip_ranges,
                                       int nskip, void* stream) {
    MI355_REQUIRE(grad && partials && n > 0 && ((uintptr_t)grad & 15) == 0, "grad_sumsq: bad args");
    MI355_REQUIRE(nskip == 0 || skip_ranges, "grad_sumsq: null skip table");
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3(MI355_OPT_PARTIALS), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)grad, grad, (long long)n, partials, (const long long*)skip_ranges, nskip);
    return check_launch("grad_sumsq_kernel");
}

__global__ __launch_bounds__(256) void adamw_prepare_kernel(const float* __restrict__ partials,
                                                            const float* __restrict__ losses, float max_norm,
                                                            OptState* __restrict__ st) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < MI355_OPT_PARTIALS; i += 256) acc += (double)partials[i];   // fixed order
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        const float skip = (losses && (losses[0] + losses[1]) == 0.0f) ? 1.0f : 0.0f;   // train.py:316-317
        st->grad_norm = norm;
        st->clip_coef = fminf(1.0f, max_norm / (norm + 1e-6f));                         // clip_grad_norm_
        st->skip = skip;
        if (skip == 0.0f) st->step += 1;
    }
}

extern "C" int mi355_adamw_prepare(const float* partials, const float* losses, float max_norm, void* state,
                                   void* stream) {
    MI355_REQUIRE(partials && state, "adamw_prepare: null pointer");
    hipLaunchKernelGGL(adamw_prepare_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, losses,
                       max_norm, (OptState*)state);
    return check_launch("adamw_prepare_kernel");
}

__global__ __launch_bounds__(256) void adamw_flat_kernel(float4* __restrict__ p, float4* __restrict__ g,
                                                         float4* __restrict__ m, float4* __restrict__ v,
                                                         long long n4, float lr, float b1, float b2, float eps,
                                                         float wd, const OptState* __restrict__ st, int scale_grad,
                                                         const long long* __restrict__ skip, int nskip,
                                                         const float* __restrict__ hyper) {
    const OptState s = *st;
    if (s.skip != 0.0f) return;
    if (hyper) {                       // device-resident hyper-parameters (a captured launch follows an LR scheduler)
        lr = hyper[0]; b1 = hyper[1]; b2 = hyper[2]; eps = hyper[3]; wd = hyper[4];
    }
    const float bc1 = 1.0f - powf(b1, (float)s.step);
    const float bc2s = sqrtf(1.0f - powf(b2, (float)s.step));
    const float step_size = lr / bc1;
    const float decay = 1.0f - lr * wd;
    const float c = s.clip_coef;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (nskip > 0 && in_ranges(skip, nskip, 4 * i + e)) continue;     // untouched, like a `grad is None` parameter
            const float gr = G[e] * c;
            G[e] = gr;
            float x = P[e] * decay;
            M[e] = M[e] + (gr - M[e]) * (1.0f - b1);               // lerp_
            V[e] = V[e] * b2 + gr * gr * (1.0f - b2);
            const float denom = sqrtf(V[e]) / bc2s + eps;
            P[e] = x - step_size * (M[e] / denom);
        }
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (scale_grad) g[i] = gg;
    }
}

extern "C" int mi355_adamw_flat(float* param, float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, const void* state, int scale_grad,
                                void* stream) {
    return mi355_adamw_flat_masked(param, grad, m, v, n, lr, beta1, beta2, eps, weight_decay, state, scale_grad,
                                   nullptr, 0, stream);
}

static int adamw_launch(float* param, float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, const void* state, int scale_grad, const int64_t* skip_ranges,
                        int nskip, const float* hyper, void* stream) {
    MI355_REQUIRE(nskip == 0 || skip_ranges, "adamw_flat: null skip table");
    MI355_REQUIRE(param && grad && m && v && state && n > 0, "adamw_flat: null pointer");
    MI355_REQUIRE((n & 3) == 0, "adamw_flat: arena length must be a multiple of 4 (pad the arena)");
    MI355_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adamw_flat: alignment");
    const long long n4 = n >> 2;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (float4*)param, (float4*)grad, (float4*)m, (float4*)v, n4, lr, beta1, beta2, eps,
                       weight_decay, (const OptState*)state, scale_grad, (const long long*)skip_ranges, nskip, hyper);
    return check_launch("adamw_flat_kernel");
}

extern "C" int mi355_adamw_flat_masked(float* param, float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                                       float beta2, float eps, float weight_decay, const void* state, int scale_grad,
                                       const int64_t* skip_ranges, int nskip, void* stream) {
    return adamw_launch(param, grad, m, v, n, lr, beta1, beta2, eps, weight_decay, state, scale_grad, skip_ranges, nskip,
                        nullptr, stream);
}

extern "C" int mi355_adamw_flat_hyper(float* param, float* grad, float* m, float* v, int64_t n, const float* hyper,
                                      const void* state, int scale_grad, const int64_t* skip_ranges, int nskip,
                                      void* stream) {
    MI355_REQUIRE(hyper, "adamw_flat_hyper: null hyper-parameter vector");
    return adamw_launch(param, grad, m, v, n, 0.f, 0.f, 0.f, 0.f, 0.f, state, scale_grad, skip_ranges, nskip, hyper, stream);
}
