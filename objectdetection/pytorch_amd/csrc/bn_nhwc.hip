// Train-mode / cross-replica BatchNorm INSIDE the fused launch plans (gfx950, HBM-bound streaming passes
// over NHWC bf16 or fp32 activations).  The frozen-BN policy of the reference loop (train.py:273-277) folds
// BN into the conv weights; a model passed through `replace_w_sync_bn` (utils/utils.py:140-168) or left in
// train mode instead needs per-layer batch statistics:
//
//   forward   z = conv(x)  ->  [sum z | sum z^2] per channel  (-> all-reduce over the replicas)
//             -> mean, invstd (utils/sync_batchnorm/batchnorm.py:128-145: clamp(biased var, eps)^-1/2 when the
//                statistics are synchronised; F.batch_norm's (var + eps)^-1/2 for one replica, :65-69),
//                running statistics (momentum, UNBIASED variance)
//             -> y = act(z * (gamma * invstd) + (beta - mean * gamma * invstd) + residual)
//   backward  g = dL/dy  ->  [sum g | sum g * zhat]  (-> all-reduce)  ->  dgamma, dbeta (local sums) and
//             dz = gamma * invstd * (g - mean(g) - zhat * mean(g * zhat))
//
// All reductions go through per-block partial sums that are added in index order: bit-reproducible.
#include "conv_common.h"
#include "mi355det_bn.h"

using namespace mi355;

namespace {

// per-layer state written by the finalize pass, fp32 [5][C]
enum { ST_MEAN = 0, ST_INVSTD = 1, ST_SCALE = 2, ST_SHIFT = 3, ST_THROUGH = 4 };

template <bool F32>
__device__ __forceinline__ void load8(const void* base, long long elem, float (&v)[8]) {
    if (F32) {
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem);
        const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        const s8v a = *reinterpret_cast<const s8v*>(reinterpret_cast<const unsigned short*>(base) + elem);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf16_to_f32((unsigned short)a[e]);
    }
}

template <bool F32>
__device__ __forceinline__ void store8(void* base, long long elem, const float (&v)[8]) {
    if (F32) {
        float* p = reinterpret_cast<float*>(base) + elem;
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        s8v o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (short)f32_to_bf16_rn(v[e]);
        *reinterpret_cast<s8v*>(reinterpret_cast<unsigned short*>(base) + elem) = o;
    }
}

// partials[blk][0..C) = sum a, [C..2C) = sum a*a (forward) or sum a * (b - mean) * invstd (backward)
template <bool F32, bool BWD>
__global__ __launch_bounds__(256) void bn_stats_kernel(const void* __restrict__ a, const void* __restrict__ b,
                                                       const float* __restrict__ state, long long M, int C,
                                                       int rows_per_block, float* __restrict__ partials) {
    __shared__ float red[256 * 16];
    const int tid = threadIdx.x;
    const int nchunks = C >> 3;
    const int CT = nchunks < 256 ? nchunks : 256;        // column chunks handled concurrently
    const int phases = 256 / CT;
    const int ct = tid % CT, phase = tid / CT;
    const long long row0 = (long long)blockIdx.x * rows_per_block;
    const long long row1 = row0 + rows_per_block < M ? row0 + rows_per_block : M;
    float* out = partials + (long long)blockIdx.x * 2 * C;
    for (int c0 = 0; c0 < nchunks; c0 += CT) {
        const int cc = c0 + ct;
        float s0[8], s1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s0[e] = 0.0f; s1[e] = 0.0f; }
        if (cc < nchunks && phase < phases) {
            float mu[8], is[8];
            if (BWD) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { mu[e] = state[ST_MEAN * C + cc * 8 + e]; is[e] = state[ST_INVSTD * C + cc * 8 + e]; }
            }
            for (long long m = row0 + phase; m < row1; m += phases) {
                float va[8];
                load8<F32>(a, m * C + cc * 8, va);
                if (BWD) {
                    float vb[8];
                    load8<F32>(b, m * C + cc * 8, vb);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s0[e] += va[e]; s1[e] += va[e] * ((vb[e] - mu[e]) * is[e]); }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s0[e] += va[e]; s1[e] += va[e] * va[e]; }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s0[e]; red[tid * 16 + 8 + e] = s1[e]; }
        __syncthreads();
        // thread t < CT*16 sums one (chunk, which, element) over the phases in order
        for (int t = tid; t < CT * 16; t += 256) {
            const int chunk = t >> 4, w = t & 15;
            float acc = 0.0f;
            for (int ph = 0; ph < phases; ++ph) acc += red[(ph * CT + chunk) * 16 + w];
            const int col = (c0 + chunk) * 8 + (w & 7);
            if (c0 + chunk < nchunks) out[(w >> 3) * C + col] = acc;
        }
        __syncthreads();
    }
}

// sums[2C] = sum over the partial blocks; backward: dgamma += sums[C + c], dbeta += sums[c].
// A block owns 8 columns; its 32 row groups each add every 32nd partial row (four independent loads in flight, double
// accumulation), and one thread per column adds the 32 group sums in group order: a fixed order, so bit-reproducible,
// and ~nblk / 32 dependent loads deep instead of nblk (one thread per column walking up to 1024 partial rows measured
// 204 us per launch: 42 % of the GPU time of a synchronised-BN step, profiles/r03_syncbn_kernel_stats_before.txt).
constexpr int RED_COLS = 8, RED_GROUPS = 32;
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ partials, int nblk, int C,
                                                        float* __restrict__ sums, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, long long local_count) {
    __shared__ double red[RED_GROUPS][RED_COLS];
    if (local_count > 0 && blockIdx.x == 0 && threadIdx.x == 0) {       // the replica's element count rides the exchange
        sums[2 * C] = (float)(local_count >> 12);
        sums[2 * C + 1] = (float)(local_count & 4095);
    }
    const int col = threadIdx.x % RED_COLS, grp = threadIdx.x / RED_COLS;
    const int i = blockIdx.x * RED_COLS + col;
    const long long stride = 2LL * C;
    double acc = 0.0;
    if (i < 2 * C) {
        int b = grp;
        for (; b + 3 * RED_GROUPS < nblk; b += 4 * RED_GROUPS) {
            const float v0 = partials[(long long)b * stride + i];
            const float v1 = partials[(long long)(b + RED_GROUPS) * stride + i];
            const float v2 = partials[(long long)(b + 2 * RED_GROUPS) * stride + i];
            const float v3 = partials[(long long)(b + 3 * RED_GROUPS) * stride + i];
            acc += (double)v0; acc += (double)v1; acc += (double)v2; acc += (double)v3;
        }
        for (; b < nblk; b += RED_GROUPS) acc += (double)partials[(long long)b * stride + i];
    }
    red[grp][col] = acc;
    __syncthreads();
    if (grp != 0 || i >= 2 * C) return;
    double tot = 0.0;
#pragma unroll
    for (int g = 0; g < RED_GROUPS; ++g) tot += red[g][col];
    const float v = (float)tot;
    sums[i] = v;
    if (i < C) { if (dbeta) dbeta[i] += v; }
    else if (dgamma) dgamma[i - C] += v;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ sums, float count, int C, float eps,
                                                          float momentum, int sync_clamp, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, long long* __restrict__ nbt,
                                                          float* __restrict__ state) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    double n = (double)count;
    if (count <= 0.0f) {                                 // counted form: the all-reduced element count of the replicas
        n = (double)sums[2 * C] * 4096.0 + (double)sums[2 * C + 1];
        if (c == 0) state[MI355_BN_STATE_ROWS * C] = (float)(1.0 / n);
    }
    const double s = (double)sums[c], ss = (double)sums[C + c];
    const double mean = s / n;
    double sumvar = ss - s * mean;                       // batchnorm.py:131-132
    const double biased = sumvar / n;
    const double unbiased = n > 1.0 ? sumvar / (n - 1.0) : biased;
    float invstd, through = 1.0f;
    if (sync_clamp) {                                    // batchnorm.py:145: bias_var.clamp(eps) ** -0.5
        const float bv = (float)biased;
        through = bv >= eps ? 1.0f : 0.0f;               // the clamp passes no gradient to the variance below eps
        invstd = 1.0f / sqrtf(fmaxf(bv, eps));
    } else {                                             // F.batch_norm(training=True): (var + eps)^-1/2
        invstd = 1.0f / sqrtf(fmaxf((float)biased, 0.0f) + eps);
    }
    const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
    const float scale = g * invstd;
    state[ST_MEAN * C + c] = (float)mean;
    state[ST_INVSTD * C + c] = invstd;
    state[ST_SCALE * C + c] = scale;
    state[ST_SHIFT * C + c] = b - (float)mean * scale;
    state[ST_THROUGH * C + c] = through;
    if (running_mean) {
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

template <bool F32>
__global__ __launch_bounds__(256) void bn_apply_kernel(const void* __restrict__ z, const float* __restrict__ state,
                                                       const void* __restrict__ addend, int act, long long n8, int C,
                                                       void* __restrict__ y) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int c0 = (int)((i * 8) % C);
    float v[8];
    load8<F32>(z, i * 8, v);
    // per-channel coefficients as two 16-B loads per row (C % 8 == 0, 16-B aligned state) instead of eight scalar ones
    float sc[8], sh[8];
    load8<true>(state, (long long)ST_SCALE * C + c0, sc);
    load8<true>(state, (long long)ST_SHIFT * C + c0, sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
    if (addend) {
        float ad[8];
        load8<F32>(addend, i * 8, ad);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += ad[e];
    }
    if (act == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
    }
    store8<F32>(y, i * 8, v);
}

// dz = scale * (g - (sum g + zhat * through * sum(g zhat)) / n)
template <bool F32>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const void* __restrict__ g, const void* __restrict__ z,
                                                           const float* __restrict__ state, const float* __restrict__ sums,
                                                           float inv_count, long long n8, int C, void* __restrict__ dz) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    if (inv_count <= 0.0f) inv_count = state[MI355_BN_STATE_ROWS * C];
    const int c0 = (int)((i * 8) % C);
    float vg[8], vz[8];
    load8<F32>(g, i * 8, vg);
    load8<F32>(z, i * 8, vz);
    float mean[8], is[8], sc[8], th[8], s0[8], s1[8];      // 16-B coefficient loads (see bn_apply_kernel)
    load8<true>(state, (long long)ST_MEAN * C + c0, mean);
    load8<true>(state, (long long)ST_INVSTD * C + c0, is);
    load8<true>(state, (long long)ST_SCALE * C + c0, sc);
    load8<true>(state, (long long)ST_THROUGH * C + c0, th);
    load8<true>(sums, (long long)c0, s0);
    load8<true>(sums, (long long)C + c0, s1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float zhat = (vz[e] - mean[e]) * is[e];
        const float mg = s0[e] * inv_count, mgz = s1[e] * inv_count * th[e];
        vg[e] = sc[e] * (vg[e] - mg - zhat * mgz);
    }
    store8<F32>(dz, i * 8, vg);
}

}  // namespace

extern "C" int mi355_bn_nhwc_partial_blocks(int64_t M, int C, int* nblk, int* rows_per_block) {
    MI355_REQUIRE(M > 0 && C > 0 && nblk && rows_per_block, "bn_nhwc_partial_blocks: bad args");
    long long rows = cdiv(M, 1024);                      // <= 1024 blocks, >= 32 rows each
    if (rows < 32) rows = 32;
    *rows_per_block = (int)rows;
    *nblk = (int)cdiv(M, rows);
    return MI355_OK;
}

extern "C" int mi355_bn_nhwc_stats(const void* a, const void* b, const float* state, int f32, int64_t M, int C,
                                   int bwd, float* partials, void* stream) {
    MI355_REQUIRE(a && partials && M > 0 && C > 0 && C % 8 == 0, "bn_nhwc_stats: bad args (C %% 8)");
    MI355_REQUIRE(!bwd || (b && state), "bn_nhwc_stats: backward needs z and the layer state");
    MI355_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0, "bn_nhwc_stats: alignment");
    int nblk, rpb;
    mi355_bn_nhwc_partial_blocks(M, C, &nblk, &rpb);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(F, B) hipLaunchKernelGGL((bn_stats_kernel<F, B>), dim3(nblk), dim3(256), 0, st, a, b, state, (long long)M, C, rpb, partials)
    if (f32) { if (bwd) LAUNCH(true, true); else LAUNCH(true, false); }
    else     { if (bwd) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    return check_launch("bn_stats_kernel");
}

extern "C" int mi355_bn_nhwc_reduce(const float* partials, int nblk, int C, float* sums, float* dgamma, float* dbeta,
                                    void* stream) {
    MI355_REQUIRE(partials && sums && nblk > 0 && C > 0, "bn_nhwc_reduce: bad args");
    hipLaunchKernelGGL(bn_reduce_kernel, dim3((unsigned)cdiv(2 * C, RED_COLS)), dim3(256), 0, (hipStream_t)stream,
                       partials, nblk, C, sums, dgamma, dbeta, 0LL);
    return check_launch("bn_reduce_kernel");
}

extern "C" int mi355_bn_nhwc_reduce_counted(const float* partials, int nblk, int C, int64_t local_count, float* sums,
                                            float* dgamma, float* dbeta, void* stream) {
    MI355_REQUIRE(partials && sums && nblk > 0 && C > 0 && local_count > 0 && local_count < (1LL << 36),
                  "bn_nhwc_reduce_counted: bad args");
    hipLaunchKernelGGL(bn_reduce_kernel, dim3((unsigned)cdiv(2 * C, RED_COLS)), dim3(256), 0, (hipStream_t)stream,
                       partials, nblk, C, sums, dgamma, dbeta, (long long)local_count);
    return check_launch("bn_reduce_kernel");
}

extern "C" int mi355_bn_nhwc_finalize(const float* sums, float count, int C, float eps, float momentum, int sync_clamp,
                                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                                      int64_t* num_batches_tracked, float* state, void* stream) {
    MI355_REQUIRE(sums && state && C > 0, "bn_nhwc_finalize: bad args");
    MI355_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_nhwc_finalize: running stats come in pairs");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, count,
                       C, eps, momentum, sync_clamp, gamma, beta, running_mean, running_var,
                       (long long*)num_batches_tracked, state);
    return check_launch("bn_finalize_kernel");
}

extern "C" int mi355_bn_nhwc_apply(const void* z, const float* state, const void* addend, int act, int f32, int64_t M,
                                   int C, void* y, void* stream) {
    MI355_REQUIRE(z && state && y && M > 0 && C > 0 && C % 8 == 0 && (act == 0 || act == 1), "bn_nhwc_apply: bad args");
    MI355_REQUIRE(((uintptr_t)state & 15) == 0, "bn_nhwc_apply: the layer state must be 16-B aligned");
    const long long n8 = (long long)M * C / 8;
    const dim3 g((unsigned)cdiv(n8, 256)), b(256);
    if (f32) hipLaunchKernelGGL(bn_apply_kernel<true>, g, b, 0, (hipStream_t)stream, z, state, addend, act, n8, C, y);
    else     hipLaunchKernelGGL(bn_apply_kernel<false>, g, b, 0, (hipStream_t)stream, z, state, addend, act, n8, C, y);
    return check_launch("bn_apply_kernel");
}

extern "C" int mi355_bn_nhwc_bwd_apply(const void* g, const void* z, const float* state, const float* sums,
                                       float inv_count, int f32, int64_t M, int C, void* dz, void* stream) {
    MI355_REQUIRE(g && z && state && sums && dz && M > 0 && C > 0 && C % 8 == 0, "bn_nhwc_bwd_apply: bad args");
    MI355_REQUIRE(((uintptr_t)state & 15) == 0 && ((uintptr_t)sums & 15) == 0, "bn_nhwc_bwd_apply: state / sums must be 16-B aligned");
    const long long n8 = (long long)M * C / 8;
    const dim3 gr((unsigned)cdiv(n8, 256)), b(256);
    if (f32) hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, gr, b, 0, (hipStream_t)stream, g, z, state, sums, inv_count, n8, C, dz);
    else     hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, gr, b, 0, (hipStream_t)stream, g, z, state, sums, inv_count, n8, C, dz);
    return check_launch("bn_bwd_apply_kernel");
}
