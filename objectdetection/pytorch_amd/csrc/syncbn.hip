// Train-mode (cross-replica) BatchNorm building blocks on NCHW fp32 tensors, the layout the reference's
// SynchronizedBatchNorm2d sees (utils/sync_batchnorm/batchnorm.py:63-93): per-channel reductions and a
// per-channel affine combination.  HBM-bound: one read per reduction, two reads + one write per apply.
//
//   mi355_bn_reduce_nchw:  part[c][s] = ( sum p ,  sum p*p )                    (mode 0: sum x, sum x^2)
//                                      ( sum p ,  sum p*(q - mean[c])*invstd[c]) (mode 1: sum dy, sum dy*xhat)
//   mi355_bn_affine_nchw:  out = a[c]*p + b[c]*q + c0[c]
//       forward   y  = (x - mean)*invstd*w + bias           = a*x + c0          (b = 0)
//       backward  dx = w*invstd*(dy - mean(dy) - xhat*mean(dy*xhat)) = a*dy + b*x + c0
#include "common.h"
#include "mi355det_syncbn.h"

namespace {

constexpr int RB = 256;

__global__ __launch_bounds__(RB) void bn_reduce_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       int N, int C, int HW, int mode, float* __restrict__ part) {
    const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const long long per = (long long)N * HW;
    const long long lo = per * s / S, hi = per * (s + 1) / S;
    const float mu = mode ? mean[c] : 0.0f, is = mode ? invstd[c] : 0.0f;
    float a = 0.0f, b = 0.0f;
    for (long long i = lo + threadIdx.x; i < hi; i += RB) {
        const long long n = i / HW, r = i - n * HW;
        const long long off = (n * C + c) * HW + r;
        const float v = p[off];
        a += v;
        b += mode ? v * ((q[off] - mu) * is) : v * v;
    }
    __shared__ float sa[RB / 64], sb[RB / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        b += __shfl_down(b, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ta = 0.0f, tb = 0.0f;
        for (int w = 0; w < RB / 64; ++w) { ta += sa[w]; tb += sb[w]; }      // fixed order
        part[((long long)c * S + s) * 2] = ta;
        part[((long long)c * S + s) * 2 + 1] = tb;
    }
}

__global__ __launch_bounds__(256) void bn_affine_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                        const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ c0, float* __restrict__ out,
                                                        int C, int HW, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)((i / HW) % C);
    float v = a[c] * p[i] + c0[c];
    if (b) v += b[c] * q[i];
    out[i] = v;
}

}  // namespace

extern "C" int mi355_bn_reduce_splits(int N, int C, int HW, int* splits) {
    MI355_REQUIRE(splits && N > 0 && C > 0 && HW > 0, "bn_reduce_splits: bad args");
    const long long per = (long long)N * HW;
    long long s = 2048 / C;                            // ~8 blocks per CU in total
    if (s < 1) s = 1;
    if (s > per / 1024) s = per / 1024 > 0 ? per / 1024 : 1;
    *splits = (int)s;
    return MI355_OK;
}

extern "C" int mi355_bn_reduce_nchw(const float* p, const float* q, const float* mean, const float* invstd,
                                    int N, int C, int HW, int mode, int splits, float* part, void* stream) {
    MI355_REQUIRE(p && part && N > 0 && C > 0 && HW > 0 && splits > 0, "bn_reduce: bad args");
    MI355_REQUIRE(mode == 0 || (q && mean && invstd), "bn_reduce: mode 1 needs q, mean, invstd");
    hipLaunchKernelGGL(bn_reduce_kernel, dim3((unsigned)C, (unsigned)splits), dim3(RB), 0, (hipStream_t)stream,
                       p, q, mean, invstd, N, C, HW, mode, part);
    return mi355::check_launch("bn_reduce_kernel");
}

extern "C" int mi355_bn_affine_nchw(const float* p, const float* q, const float* a, const float* b, const float* c0,
                                    float* out, int N, int C, int HW, void* stream) {
    MI355_REQUIRE(p && a && c0 && out && N > 0 && C > 0 && HW > 0, "bn_affine: bad args");
    MI355_REQUIRE(!b || q, "bn_affine: b needs q");
    const long long total = (long long)N * C * HW;
    hipLaunchKernelGGL(bn_affine_kernel, dim3((unsigned)mi355::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       p, q, a, b, c0, out, C, HW, total);
    return mi355::check_launch("bn_affine_kernel");
}
