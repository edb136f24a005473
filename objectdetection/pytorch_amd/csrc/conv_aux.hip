// HBM-bound helpers around the MFMA conv kernels (gfx950): layout conversion, weight
// preparation (frozen-BN fold + bf16 shadow copies), parameter-gradient finishing, max-pool,
// nearest-upsample backward, ReLU mask, column sums.  All activation traffic is 16 B per lane.
#include "conv_common.h"

using namespace mi355;

// ----------------------------------------------------------------------- NCHW fp32 -> NHWC bf16
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, unsigned short* __restrict__ out,
                                    long long npix_total, int C, long long HW, int Cpad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // pixel index (n, hw)
    if (i >= npix_total) return;
    const long long n = i / HW, hw = i - n * HW;
    const float* src = in + n * C * HW + hw;
    unsigned short* dst = out + i * Cpad;
    for (int c0 = 0; c0 < Cpad; c0 += 8) {
        s8v o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            o[e] = (short)(c < C ? f32_to_bf16_rn(src[c * HW]) : (unsigned short)0);
        }
        *reinterpret_cast<s8v*>(dst + c0) = o;
    }
}

extern "C" int mi355_nchw_to_nhwc_bf16(const float* in, void* out, int N, int C, int H, int W,
                                       int Cpad, void* stream) {
    MI355_REQUIRE(in && out && N > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C && Cpad % 8 == 0,
                  "nchw_to_nhwc: bad args");
    const long long npix = (long long)N * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)cdiv(npix, 256)), dim3(256), 0,
                       (hipStream_t)stream, in, (unsigned short*)out, npix, C, (long long)H * W, Cpad);
    return check_launch("nchw_to_nhwc_kernel");
}

// ----------------------------------------------------------------------- space-to-depth stem
// The 7x7 / stride-2 / pad-3 stem conv over 3 channels (retinanet.py:245, utils_resnet via torchvision's ResNet stem) is run as
// a 4x4 / stride-1 conv over the 2x2 space-to-depth image: x'[n, i, j, (di, dj, c)] = x[n, c, 2i + di, 2j + dj] (12
// channels, padded to 16) and W'[k, (di, dj, c), a, b] = W[k, c, 2a + di - 1, 2b + dj - 1] (zero where the 7x7 index falls
// outside 0..6), pad 2 before / 1 after: out[p, q] = sum W[r, s] x[2p - 3 + r, 2q - 3 + s] with r - 3 = 2(a - 2) + di.
// 16 taps x 16 channels = 256 reduction columns instead of 49 x 8 = 392, 32-B instead of 16-B gather pieces: forward
// 118 -> 77 us, weight gradient 155 -> 90 us at 16 x 512 x 512 (tools/bench_stem.py).  W' is a derived buffer rebuilt
// before the weight preparation; its gradient is scattered back into dW after the parameter-gradient table.
__global__ void nchw_to_nhwc_s2d_kernel(const float* __restrict__ in, unsigned short* __restrict__ out,
                                        long long nq_total, int H, int W) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (n, y2, x2)
    if (i >= nq_total) return;
    const int W2 = W >> 1, H2 = H >> 1;
    const int x2 = (int)(i % W2);
    const long long r = i / W2;
    const int y2 = (int)(r % H2);
    const long long n = r / H2;
    const long long HW = (long long)H * W;
    const float* src = in + n * 3 * HW + (long long)(2 * y2) * W + 2 * x2;
    s8v o0, o1;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        unsigned short v = 0;
        if (e < 12) {
            const int q = e / 3, c = e - q * 3, di = q >> 1, dj = q & 1;
            v = f32_to_bf16_rn(src[c * HW + (long long)di * W + dj]);
        }
        if (e < 8) o0[e] = (short)v; else o1[e - 8] = (short)v;
    }
    s8v* dst = reinterpret_cast<s8v*>(out + i * 16);
    dst[0] = o0;
    dst[1] = o1;
}

extern "C" int mi355_nchw_to_nhwc_s2d_bf16(const float* in, void* out, int N, int H, int W, void* stream) {
    MI355_REQUIRE(in && out && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "nchw_to_nhwc_s2d: even H, W, 3 channels");
    const long long nq = (long long)N * (H / 2) * (W / 2);
    hipLaunchKernelGGL(nchw_to_nhwc_s2d_kernel, dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, (hipStream_t)stream, in,
                       (unsigned short*)out, nq, H, W);
    return check_launch("nchw_to_nhwc_s2d_kernel");
}

// wp [K][12][4][4] contiguous  <->  w [K][3][7][7] with element strides (sk, sc, sr, ss)
__global__ void stem_s2d_pack_kernel(const float* __restrict__ w, long long sk, long long sc, long long sr, long long ss,
                                     int K, float* __restrict__ wp, int unpack, float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * 192) return;
    const int k = i / 192, rem = i - k * 192;
    const int cp = rem >> 4, a = (rem >> 2) & 3, b = rem & 3;
    const int q = cp / 3, c = cp - q * 3, di = q >> 1, dj = q & 1;
    const int r = 2 * a + di - 1, s2 = 2 * b + dj - 1;
    const bool ok = r >= 0 && r < 7 && s2 >= 0 && s2 < 7;
    const long long o = k * sk + c * sc + r * sr + s2 * ss;
    if (!unpack) wp[i] = ok ? w[o] : 0.0f;
    else if (ok) dw[o] += wp[i];                       // every 7x7 element has exactly one image in wp: no race
}

extern "C" int mi355_stem_s2d_pack(const float* w, int64_t sk, int64_t sc, int64_t sr, int64_t ss, int K, float* wp,
                                   void* stream) {
    MI355_REQUIRE(w && wp && K > 0, "stem_s2d_pack: bad args");
    hipLaunchKernelGGL(stem_s2d_pack_kernel, dim3((unsigned)cdiv(K * 192, 256)), dim3(256), 0, (hipStream_t)stream, w, sk, sc,
                       sr, ss, K, wp, 0, nullptr);
    return check_launch("stem_s2d_pack_kernel");
}

extern "C" int mi355_stem_s2d_unpack_grad(const float* dwp, float* dw, int64_t sk, int64_t sc, int64_t sr, int64_t ss, int K,
                                          void* stream) {
    MI355_REQUIRE(dwp && dw && K > 0, "stem_s2d_unpack_grad: bad args");
    hipLaunchKernelGGL(stem_s2d_pack_kernel, dim3((unsigned)cdiv(K * 192, 256)), dim3(256), 0, (hipStream_t)stream, nullptr, sk,
                       sc, sr, ss, K, const_cast<float*>(dwp), 1, dw);
    return check_launch("stem_s2d_unpack_kernel");
}

// ----------------------------------------------------------------------- frozen-BN fold
__global__ void bn_fold_kernel(const float* g, const float* b, const float* m, const float* v,
                               const float* cb, float eps, int K, float* scale, float* shift) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float s = g[k] / sqrtf(v[k] + eps);
    scale[k] = s;
    shift[k] = b[k] - (m[k] - (cb ? cb[k] : 0.0f)) * s;
}

extern "C" int mi355_bn_fold(const float* gamma, const float* beta, const float* mean,
                             const float* var, const float* conv_bias, float eps, int K, float* scale,
                             float* shift, void* stream) {
    MI355_REQUIRE(gamma && beta && mean && var && scale && shift && K > 0, "bn_fold: bad args");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream,
                       gamma, beta, mean, var, conv_bias, eps, K, scale, shift);
    return check_launch("bn_fold_kernel");
}

// ----------------------------------------------------------------------- weight preparation
// One block = one (rs, 32 k, 32 c) tile: coalesced fp32 reads, LDS transpose, bf16 writes of
// both operand layouts.
__global__ __launch_bounds__(256) void prep_weight_kernel(
    const float* __restrict__ w, long long sk, long long sc, long long sr, long long ss, int K, int C,
    int R, int S, int Cpad, const float* __restrict__ scale, unsigned short* __restrict__ w_fwd,
    unsigned short* __restrict__ w_dgr, int Kpad) {
    __shared__ float tile[32][33];
    const int rs = blockIdx.z, r = rs / S, s = rs - r * S;
    const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int T = R * S;
    for (int j = ty; j < 32; j += 8) {
        const int k = k0 + j, c = c0 + tx;
        float v = 0.0f;
        if (k < K && c < C) v = w[k * sk + c * sc + r * sr + s * ss] * (scale ? scale[k] : 1.0f);
        tile[j][tx] = v;
        if (w_fwd && k < K && c < Cpad) w_fwd[((long long)k * T + rs) * Cpad + c] = f32_to_bf16_rn(v);
    }
    __syncthreads();
    if (w_dgr) {
        for (int j = ty; j < 32; j += 8) {
            const int c = c0 + j, k = k0 + tx;
            if (c < C && k < Kpad) w_dgr[((long long)c * T + rs) * Kpad + k] = f32_to_bf16_rn(tile[tx][j]);
        }
    }
}

extern "C" int mi355_prep_weight(const float* w, int64_t sk, int64_t sc, int64_t sr, int64_t ss, int K,
                                 int C, int R, int S, int Cpad, const float* scale, void* w_fwd,
                                 void* w_dgr, void* stream) {
    MI355_REQUIRE(w && K > 0 && C > 0 && R > 0 && S > 0 && Cpad >= C && Cpad % 8 == 0, "prep_weight: bad args");
    const int Kpad = (K + 7) / 8 * 8;
    dim3 grid((unsigned)cdiv(Cpad, 32), (unsigned)cdiv(Kpad, 32), (unsigned)(R * S));
    hipLaunchKernelGGL(prep_weight_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, (long long)sk,
                       (long long)sc, (long long)sr, (long long)ss, K, C, R, S, Cpad, scale,
                       (unsigned short*)w_fwd, (unsigned short*)w_dgr, Kpad);
    return check_launch("prep_weight_kernel");
}

// ----------------------------------------------------------------------- column sums
// out[k] += sum_m g[m][k].  One block streams `rows_per_block` consecutive rows with 16-B loads
// (thread = (row phase, 8-column chunk)), reduces the row phases through LDS and issues ONE
// atomic per column per block.
__global__ __launch_bounds__(256) void colsum_kernel(const unsigned short* __restrict__ g, long long M,
                                                     int K, int rows_per_block, float* __restrict__ out) {
    __shared__ float s_acc[2048 + 8];
    const int k8n = K >> 3;
    const int rpb = 256 / k8n;                       // row phases handled concurrently
    const int tid = threadIdx.x;
    const int rsub = tid / k8n, k8 = tid - rsub * k8n;
    const long long row0 = (long long)blockIdx.x * rows_per_block;
    const long long row1 = row0 + rows_per_block < M ? row0 + rows_per_block : M;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (rsub < rpb) {
        for (long long m = row0 + rsub; m < row1; m += rpb) {
            const s8v v = *reinterpret_cast<const s8v*>(g + m * K + k8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf16_to_f32((unsigned short)v[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s_acc[rsub * K + k8 * 8 + e] = acc[e];
    }
    __syncthreads();
    for (int k = tid; k < K; k += 256) {
        float t = 0.0f;
        for (int r = 0; r < rpb; ++r) t += s_acc[r * K + k];
        atomicAdd(out + k, t);
    }
}

extern "C" int mi355_colsum_bf16(const void* g, int64_t M, int K, float* out, void* stream) {
    MI355_REQUIRE(g && out && M > 0 && K > 0 && K % 8 == 0 && K <= 2048,
                  "colsum: K must be a multiple of 8, <= 2048");
    long long rows = (M + 1023) / 1024;              // ~1024 blocks
    if (rows < 64) rows = 64;
    const long long blocks = (M + rows - 1) / rows;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)g, (long long)M, K, (int)rows, out);
    return check_launch("colsum_kernel");
}

// ----------------------------------------------------------------------- parameter gradients
// One block per output channel k: dW = G*scale[k]; dgamma/dbeta of the folded frozen BN.
__global__ __launch_bounds__(256) void conv_param_grads_kernel(
    const float* __restrict__ G, const float* __restrict__ w, long long sk, long long sc, long long sr,
    long long ss, int C, int R, int S, int Cpad, const float* __restrict__ scale,
    const float* __restrict__ colsum, const float* __restrict__ mean, const float* __restrict__ var,
    float eps, float* __restrict__ dW, float* __restrict__ dgamma, float* __restrict__ dbeta,
    int accumulate, const float* __restrict__ conv_bias, float* __restrict__ dconv_bias) {
    const int k = blockIdx.x, tid = threadIdx.x;
    const int T = R * S;
    const float sck = scale ? scale[k] : 1.0f;
    float dot = 0.0f;
    const int total = T * Cpad;
    for (int i = tid; i < total; i += 256) {
        const int t = i / Cpad, c = i - t * Cpad;
        if (c >= C) continue;
        const int r = t / S, s = t - r * S;
        const float gv = G[(long long)k * total + i];
        const long long wo = k * sk + c * sc + r * sr + s * ss;
        if (dgamma) dot += w[wo] * gv;
        const float d = gv * sck;
        dW[wo] = accumulate ? dW[wo] + d : d;
    }
    if (dgamma || dbeta || dconv_bias) {
        __shared__ float red[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o, 64);
        if ((tid & 63) == 0) red[tid >> 6] = dot;
        __syncthreads();
        if (tid == 0) {
            const float cs = colsum[k];
            if (dgamma) {
                const float invstd = 1.0f / sqrtf(var[k] + eps);
                const float mu = mean[k] - (conv_bias ? conv_bias[k] : 0.0f);
                const float dg = invstd * ((red[0] + red[1] + red[2] + red[3]) - mu * cs);
                dgamma[k] = accumulate ? dgamma[k] + dg : dg;
            }
            if (dbeta) dbeta[k] = accumulate ? dbeta[k] + cs : cs;
            if (dconv_bias) dconv_bias[k] = accumulate ? dconv_bias[k] + sck * cs : sck * cs;
        }
    }
}

extern "C" int mi355_conv_param_grads(const float* G, const float* w, int64_t sk, int64_t sc,
                                      int64_t sr, int64_t ss, int K, int C, int R, int S, int Cpad,
                                      const float* scale, const float* colsum, const float* mean,
                                      const float* var, float eps, float* dW, float* dgamma,
                                      float* dbeta, int accumulate, const float* conv_bias,
                                      float* dconv_bias, void* stream) {
    MI355_REQUIRE(G && w && dW && K > 0 && C > 0 && R > 0 && S > 0 && Cpad >= C, "conv_param_grads: bad args");
    MI355_REQUIRE(!(dgamma || dbeta) || colsum, "conv_param_grads: colsum needed for dgamma/dbeta");
    MI355_REQUIRE(!dgamma || (mean && var), "conv_param_grads: mean/var needed for dgamma");
    hipLaunchKernelGGL(conv_param_grads_kernel, dim3((unsigned)K), dim3(256), 0, (hipStream_t)stream, G, w,
                       (long long)sk, (long long)sc, (long long)sr, (long long)ss, C, R, S, Cpad, scale,
                       colsum, mean, var, eps, dW, dgamma, dbeta, accumulate, conv_bias, dconv_bias);
    return check_launch("conv_param_grads_kernel");
}

// ----------------------------------------------------------------------- max-pool 3x3 s2 p1
__global__ void maxpool_fwd_kernel(const unsigned short* __restrict__ in, unsigned short* __restrict__ out,
                                   signed char* __restrict__ idx, int N, int H, int W, int C, int P, int Q,
                                   const FastDiv dc8, const FastDiv dQ, const FastDiv dP) {
    // (32-bit index arithmetic with launch-constant divisors: the 64-bit div / mod chain cost more than the loads)
    const int c8n = C >> 3;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned total = (unsigned)N * P * Q * c8n;
    if (t >= total) return;
    unsigned r = fdiv(t, dc8);
    const int c8 = (int)(t - r * c8n);
    const unsigned r2 = fdiv(r, dQ);
    const int q = (int)(r - r2 * Q);
    const int n = (int)fdiv(r2, dP);
    const int p = (int)(r2 - n * P);
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = -1; }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int h = 2 * p - 1 + a;
        if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int w = 2 * q - 1 + b;
            if ((unsigned)w >= (unsigned)W) continue;
            const s8v v = *reinterpret_cast<const s8v*>(in + (((long long)n * H + h) * W + w) * C + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = bf16_to_f32((unsigned short)v[e]);
                if (f > best[e] || bi[e] < 0) { best[e] = f; bi[e] = a * 3 + b; }   // first max wins
            }
        }
    }
    s8v o;
    unsigned long long packed = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] = (short)f32_to_bf16_rn(best[e]);
        packed |= (unsigned long long)(bi[e] & 0xff) << (8 * e);
    }
    *reinterpret_cast<s8v*>(out + (long long)t * 8) = o;
    *reinterpret_cast<unsigned long long*>(idx + (long long)t * 8) = packed;
}

extern "C" int mi355_maxpool3x3s2_fwd(const void* in, void* out, int8_t* idx, int N, int H, int W, int C,
                                      void* stream) {
    MI355_REQUIRE(in && out && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool_fwd: bad args");
    const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * P * Q * (C / 8);
    MI355_REQUIRE(total < 0x7fffffffLL, "maxpool_fwd: too many cells");
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)in, (unsigned short*)out, (signed char*)idx, N, H, W, C, P, Q,
                       make_fastdiv((unsigned)(C / 8)), make_fastdiv((unsigned)Q), make_fastdiv((unsigned)P));
    return check_launch("maxpool_fwd_kernel");
}

// A thread owns the 2 x 2 input block (2p', 2q') of one 8-channel chunk: the block lies in the windows
// (p'+a, q'+b), a, b in {0, 1}, whose (argmax, gradient) cells are loaded ONCE for the four pixels (a thread per
// input pixel loaded 2.25 cells per pixel on average).  Window (p, q) covers rows 2p-1 .. 2p+1: pixel row
// 2p'+dh sits at window row dh + 1 - 2a, likewise for columns; contributions are added in (p, q) order.
// relu_mask: the pool's input (ReLU output, input resolution) - or, POOLED, the pool's OUTPUT: behind a ReLU the
// argmax cell is positive exactly when the pooled value is (a window of zeros sends its gradient to a cell whose
// ReLU gradient is zero), so the quarter-size pooled map carries the whole mask.
template <bool POOLED>
__global__ void maxpool_bwd_kernel(const unsigned short* __restrict__ gout, const signed char* __restrict__ idx,
                                   unsigned short* __restrict__ gin, const unsigned short* __restrict__ relu_mask,
                                   int N, int H, int W, int C, int P, int Q,
                                   const FastDiv dc8, const FastDiv dW2, const FastDiv dH2) {
    const int c8n = C >> 3;
    const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (unsigned)N * H2 * W2 * c8n) return;
    const unsigned r = fdiv(t, dc8);
    const int c8 = (int)(t - r * c8n);
    const unsigned r2 = fdiv(r, dW2);
    const int q2 = (int)(r - r2 * W2);
    const int n = (int)fdiv(r2, dH2);
    const int p2 = (int)(r2 - n * H2);
    float acc[2][2][8];
#pragma unroll
    for (int i = 0; i < 32; ++i) (&acc[0][0][0])[i] = 0.0f;
    // the four cells' loads first, unconditionally (clamped addresses, an out-of-range cell gets slot 0xff): one
    // memory round trip per thread instead of one per cell behind its range test
    unsigned long long pk4[2][2];
    s8v g4[2][2], m4[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int p = min(p2 + a, P - 1), q = min(q2 + b, Q - 1);
            const long long o = (((long long)n * P + p) * Q + q) * C + c8 * 8;
            pk4[a][b] = *reinterpret_cast<const unsigned long long*>(idx + o);
            g4[a][b] = *reinterpret_cast<const s8v*>(gout + o);
            if (POOLED) m4[a][b] = *reinterpret_cast<const s8v*>(relu_mask + o);
        }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const unsigned long long pk = (p2 + a < P && q2 + b < Q) ? pk4[a][b] : ~0ull;
            s8v g = g4[a][b];
            if (POOLED) {
                const s8v m = m4[a][b];
#pragma unroll
                for (int e = 0; e < 8; ++e)                    // bf16 > 0  <=>  sign clear and not +-0
                    g[e] = ((unsigned short)m[e] & 0x8000u) || !((unsigned short)m[e] & 0x7fffu) ? (short)0 : g[e];
            }
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) {
                const int wr = dh + 1 - 2 * a;                 // window row of input row 2p'+dh
                if (wr < 0) continue;
#pragma unroll
                for (int dw = 0; dw < 2; ++dw) {
                    const int wc = dw + 1 - 2 * b;
                    if (wc < 0) continue;
                    const int slot = wr * 3 + wc;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if ((int)((pk >> (8 * e)) & 0xff) == slot) acc[dh][dw][e] += bf16_to_f32((unsigned short)g[e]);
                }
            }
        }
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
            const int h = 2 * p2 + dh, w = 2 * q2 + dw;
            if (h >= H || w >= W) continue;
            const long long o = ((((long long)n * H + h) * W + w) * c8n + c8) * 8;
            if (!POOLED && relu_mask) {
                const s8v m = *reinterpret_cast<const s8v*>(relu_mask + o);
#pragma unroll
                for (int e = 0; e < 8; ++e) if (!(bf16_to_f32((unsigned short)m[e]) > 0.0f)) acc[dh][dw][e] = 0.0f;
            }
            s8v v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (short)f32_to_bf16_rn(acc[dh][dw][e]);
            *reinterpret_cast<s8v*>(gin + o) = v;
        }
}

extern "C" int mi355_maxpool3x3s2_bwd(const void* gout, const int8_t* idx, void* gin, const void* relu_mask,
                                      int N, int H, int W, int C, void* stream) {
    MI355_REQUIRE(gout && idx && gin && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool_bwd: bad args");
    const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
    MI355_REQUIRE(total < 0x7fffffffLL, "maxpool_bwd: too many cells");
    hipLaunchKernelGGL(maxpool_bwd_kernel<false>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)gout, (const signed char*)idx, (unsigned short*)gin,
                       (const unsigned short*)relu_mask, N, H, W, C, P, Q, make_fastdiv((unsigned)(C / 8)),
                       make_fastdiv((unsigned)((W + 1) / 2)), make_fastdiv((unsigned)((H + 1) / 2)));
    return check_launch("maxpool_bwd_kernel");
}

extern "C" int mi355_maxpool3x3s2_bwd_pooled(const void* gout, const int8_t* idx, void* gin, const void* pooled,
                                             int N, int H, int W, int C, void* stream) {
    MI355_REQUIRE(gout && idx && gin && pooled && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0,
                  "maxpool_bwd_pooled: bad args");
    const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
    MI355_REQUIRE(total < 0x7fffffffLL, "maxpool_bwd: too many cells");
    hipLaunchKernelGGL(maxpool_bwd_kernel<true>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)gout, (const signed char*)idx, (unsigned short*)gin,
                       (const unsigned short*)pooled, N, H, W, C, P, Q, make_fastdiv((unsigned)(C / 8)),
                       make_fastdiv((unsigned)((W + 1) / 2)), make_fastdiv((unsigned)((H + 1) / 2)));
    return check_launch("maxpool_bwd_kernel<pooled>");
}

// ----------------------------------------------------------------------- nearest-2x upsample backward
__global__ void sumpool2x2_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst,
                                  int N, int H, int W, int C, int P, int Q, int accumulate) {
    const int c8n = C >> 3;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * P * Q * c8n;
    if (t >= total) return;
    int c8, q, p, n;
    split_cell(t, c8n, Q, P, c8, q, p, n);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (accumulate) {
        const s8v d = *reinterpret_cast<const s8v*>(dst + t * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bf16_to_f32((unsigned short)d[e]);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int h = 2 * p + a, w = 2 * q + b;
            if (h >= H || w >= W) continue;
            const s8v v = *reinterpret_cast<const s8v*>(src + (((long long)n * H + h) * W + w) * C + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf16_to_f32((unsigned short)v[e]);
        }
    s8v o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (short)f32_to_bf16_rn(acc[e]);
    *reinterpret_cast<s8v*>(dst + t * 8) = o;
}

extern "C" int mi355_sumpool2x2_bf16(const void* src, void* dst, int N, int H, int W, int C, int accumulate,
                                     void* stream) {
    MI355_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "sumpool2x2: bad args");
    const int P = (H + 1) / 2, Q = (W + 1) / 2;
    const long long total = (long long)N * P * Q * (C / 8);
    hipLaunchKernelGGL(sumpool2x2_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)src, (unsigned short*)dst, N, H, W, C, P, Q, accumulate);
    return check_launch("sumpool2x2_kernel");
}

// ----------------------------------------------------------------------- ReLU mask
__global__ void relu_mask_kernel(const s8v* __restrict__ x, const s8v* __restrict__ ref, s8v* __restrict__ y,
                                 long long n8) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const s8v a = x[i], r = ref[i];
    s8v o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = bf16_to_f32((unsigned short)r[e]) > 0.0f ? a[e] : (short)0;
    y[i] = o;
}

extern "C" int mi355_relu_mask_bf16(const void* x, const void* ref, void* y, int64_t n, void* stream) {
    MI355_REQUIRE(x && ref && y && n > 0 && n % 8 == 0, "relu_mask: n must be a multiple of 8");
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)cdiv(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const s8v*)x, (const s8v*)ref, (s8v*)y, (long long)(n / 8));
    return check_launch("relu_mask_kernel");
}

// ----------------------------------------------------------------------- table-driven variants
__device__ __forceinline__ int find_layer(const Mi355LayerDesc* __restrict__ t, int nlayers, int b) {
    int lo = 0, hi = nlayers - 1;                 // last layer with block_begin <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t[mid].block_begin <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// One block = one (tap, 32 k, 64 c) tile: two channels per thread (8-B reads where the master weights are
// channel-contiguous, 4-B bf16 pair stores), half the blocks of the 32 x 32 form and twice the bytes behind each
// block's descriptor lookup.
__global__ __launch_bounds__(256) void prep_weights_table_kernel(const Mi355LayerDesc* __restrict__ table,
                                                                 int nlayers) {
    __shared__ float tile[32][65];
    __shared__ float s_scale[32];
    const Mi355LayerDesc d = table[find_layer(table, nlayers, blockIdx.x)];
    int b = blockIdx.x - d.block_begin;
    const int ct = (d.Cpad + 63) / 64, kt = (d.Kpad + 31) / 32;
    const int c_tile = b % ct;  b /= ct;
    const int k_tile = b % kt;  b /= kt;
    const int rs = b, T = d.R * d.S, r = rs / d.S, s = rs - r * d.S;
    const int k0 = k_tile * 32, c0 = c_tile * 64;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (threadIdx.x < 32) {
        const int k = k0 + threadIdx.x;
        float sc = 1.0f;
        if (d.gamma && k < d.K) {
            sc = d.gamma[k] / sqrtf(d.var[k] + d.eps);
            if (rs == 0 && c_tile == 0) {
                d.scale[k] = sc;
                d.shift[k] = d.beta[k] - (d.mean[k] - (d.conv_bias ? d.conv_bias[k] : 0.0f)) * sc;
            }
        }
        s_scale[threadIdx.x] = sc;
    }
    __syncthreads();
    unsigned short* w_fwd = (unsigned short*)d.w_fwd;
    unsigned short* w_dgr = (unsigned short*)d.w_dgr;
    const bool f32 = d.f32_shadows != 0;               // fp32-accurate mode: the shadows stay fp32
    for (int j = ty; j < 32; j += 8) {
        const int k = k0 + j, c = c0 + 2 * tx;
        float v0 = 0.0f, v1 = 0.0f;
        if (k < d.K && c < d.C) {
            const float* wp = d.w + (k * d.sk + c * d.sc + r * d.sr + s * d.ss);
            if (d.sc == 1 && c + 1 < d.C && ((uintptr_t)wp & 7) == 0) {
                const float2 q = *reinterpret_cast<const float2*>(wp);
                v0 = q.x;  v1 = q.y;
            } else {
                v0 = wp[0];
                if (c + 1 < d.C) v1 = wp[d.sc];
            }
            v0 *= s_scale[j];  v1 *= s_scale[j];
        }
        tile[j][2 * tx] = v0;
        tile[j][2 * tx + 1] = v1;
        if (k < d.K && c < d.Cpad) {                   // (Cpad % 8 == 0: c + 1 < Cpad too, the pair is 4-B aligned)
            const long long o = ((long long)k * T + rs) * d.Cpad + c;
            if (f32) { ((float*)d.w_fwd)[o] = v0;  ((float*)d.w_fwd)[o + 1] = v1; }
            else *reinterpret_cast<unsigned*>(w_fwd + o) = (unsigned)f32_to_bf16_rn(v0) | ((unsigned)f32_to_bf16_rn(v1) << 16);
        }
    }
    __syncthreads();
    if (w_dgr) {
        for (int j = ty; j < 64; j += 8) {
            const int c = c0 + j, k = k0 + tx;
            if (c < d.C && k < d.Kpad) {
                if (f32) ((float*)d.w_dgr)[((long long)c * T + rs) * d.Kpad + k] = tile[tx][j];
                else w_dgr[((long long)c * T + rs) * d.Kpad + k] = f32_to_bf16_rn(tile[tx][j]);
            }
        }
    }
}

extern "C" int mi355_prep_weights_table(const Mi355LayerDesc* d_table, int nlayers, int nblocks, void* stream) {
    MI355_REQUIRE(d_table && nlayers > 0 && nblocks > 0, "prep_weights_table: bad args");
    hipLaunchKernelGGL(prep_weights_table_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream,
                       d_table, nlayers);
    return check_launch("prep_weights_table_kernel");
}

// sum of slabs [lo, hi) of one 16-B piece of a layer's raw weight gradient, in index order
// (slabs are written once by the weight-gradient launches and read once, here: non-temporal loads leave L2 / the Infinity
// Cache to the activations the rest of the backward pass still reads: +0.2 % on the R50 step, A/B of two builds)
typedef float pg_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 pg_load(const float4* p) {
    const pg_f4 v = __builtin_nontemporal_load(reinterpret_cast<const pg_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 pg_sum_slabs(const Mi355LayerDesc& d, long long off, int lo, int hi) {
    const float4* gp = reinterpret_cast<const float4*>(d.G + off);
    const long long ss4 = d.slab_stride >> 2;
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int sl = lo;
    for (; sl + 4 <= hi; sl += 4) {                      // four slab loads in flight, added in index order
        const float4 v0 = pg_load(gp + sl * ss4), v1 = pg_load(gp + (sl + 1) * ss4), v2 = pg_load(gp + (sl + 2) * ss4),
                     v3 = pg_load(gp + (sl + 3) * ss4);
        g4.x += v0.x; g4.y += v0.y; g4.z += v0.z; g4.w += v0.w;
        g4.x += v1.x; g4.y += v1.y; g4.z += v1.z; g4.w += v1.w;
        g4.x += v2.x; g4.y += v2.y; g4.z += v2.z; g4.w += v2.w;
        g4.x += v3.x; g4.y += v3.y; g4.z += v3.z; g4.w += v3.w;
    }
    for (; sl < hi; ++sl) {
        const float4 v = pg_load(gp + sl * ss4);
        g4.x += v.x; g4.y += v.y; g4.z += v.z; g4.w += v.w;
    }
    return g4;
}
// this part's share of a descriptor's slabs (nslabs 0 / 1: a single buffer)
__device__ __forceinline__ void pg_range(const Mi355LayerDesc& d, int part, int parts, int& lo, int& hi) {
    const int nsl = d.nslabs > 0 ? d.nslabs : 1;
    lo = (int)((long long)nsl * part / parts);
    hi = (int)((long long)nsl * (part + 1) / parts);
}
// sum over the slabs of a layer's dy column sum for channel k, by the whole block: thread t adds its contiguous share
// of the slabs in index order, the shares are combined by a fixed shuffle tree and the four waves in order (every
// thread gets the result).  (One thread walking 1024 slabs was a 1024-deep chain of dependent loads.)
__device__ __forceinline__ float pg_colsum(const Mi355LayerDesc& d, int k, int tid, float* s_red) {
    const int nsl = d.nslabs > 0 ? d.nslabs : 1;
    const int lo = (int)((long long)nsl * tid / 256), hi = (int)((long long)nsl * (tid + 1) / 256);
    float v = 0.0f;
    for (int sl = lo; sl < hi; ++sl) v += d.colsum[(long long)sl * d.Kpad + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
__device__ __forceinline__ float pg_scale(const Mi355LayerDesc& d, int k) {
    return d.gamma ? d.gamma[k] / sqrtf(d.var[k] + d.eps) : 1.0f;
}

// Descriptors that write the SAME dW (a conv shared by the pyramid levels, each level with its own folded BN, raw
// gradient and slab count) form a chain (share_next): the head's blocks add the members' contributions in chain
// order and store once, the members (atomic == 2) only produce their own BN gradients: bit-reproducible, where one
// atomicAdd per member left the order of the five additions to the scheduler.
__global__ __launch_bounds__(256) void param_grads_table_kernel(const Mi355LayerDesc* __restrict__ table,
                                                                int nlayers) {
    const Mi355LayerDesc d = table[find_layer(table, nlayers, blockIdx.x)];
    const int k = blockIdx.x - d.block_begin, tid = threadIdx.x;
    const int T = d.R * d.S;
    const float sck = pg_scale(d, k);
    const bool member = d.atomic == 2;
    const bool ow = d.atomic == 3;                     // sole writer of its gradients this step: store, do not add
    float dot = 0.0f;
    const int total = T * d.Cpad;
    // 16-B slab reads (Cpad % 8 == 0, rows and slabs 16-B aligned).  A row of few pieces (EfficientDet's 1x1 convs:
    // 4-36 pieces, but up to 1024 slabs of the 256 x 256 maps) leaves most of the block idle behind a chain of
    // dependent slab loads (0.27 ms per D0 step): the idle threads take contiguous parts of the slab range, the parts
    // are combined through LDS in part order (bit-reproducible).
    const int nvec = total >> 2;
    int lanes = 256, parts = 1;
    while (parts < 16 && (lanes >> 1) >= nvec) { lanes >>= 1; parts <<= 1; }      // lanes * parts == 256
    const int part = tid / lanes, v0 = tid - part * lanes;
    __shared__ float4 s_o[256];
    __shared__ float s_d[256];
    if (!member || d.dgamma) {
        for (int vb = 0; vb < nvec; vb += lanes) {
            const int i4 = (vb + v0) * 4;
            const bool act = vb + v0 < nvec;
            float oe[4] = {0.f, 0.f, 0.f, 0.f};
            float dp = 0.0f;
            int c0 = 0, r = 0, sx = 0;
            if (act) {
                int lo, hi;
                pg_range(d, part, parts, lo, hi);
                const float4 g4 = pg_sum_slabs(d, (long long)k * total + i4, lo, hi);
                const float ge[4] = {g4.x, g4.y, g4.z, g4.w};
                oe[0] = g4.x * sck; oe[1] = g4.y * sck; oe[2] = g4.z * sck; oe[3] = g4.w * sck;
                if (!member) {
                    for (int nx = d.share_next; nx != 0;) {
                        const Mi355LayerDesc& m = table[nx - 1];
                        const float sm = pg_scale(m, k);
                        pg_range(m, part, parts, lo, hi);
                        const float4 h4 = pg_sum_slabs(m, (long long)k * total + i4, lo, hi);
                        oe[0] += h4.x * sm; oe[1] += h4.y * sm; oe[2] += h4.z * sm; oe[3] += h4.w * sm;
                        nx = m.share_next;
                    }
                }
                const int t = i4 / d.Cpad;
                c0 = i4 - t * d.Cpad;
                r = t / d.S; sx = t - r * d.S;
                if (d.dgamma) {
                    const float* wp = d.w + (k * d.sk + c0 * d.sc + r * d.sr + sx * d.ss);
                    if (d.sc == 1 && c0 + 4 <= d.C && ((uintptr_t)wp & 15) == 0) {      // channel-contiguous weights: 16 B
                        const float4 w4 = *reinterpret_cast<const float4*>(wp);
                        dp += w4.x * ge[0];  dp += w4.y * ge[1];  dp += w4.z * ge[2];  dp += w4.w * ge[3];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c0 + e < d.C) dp += wp[e * d.sc] * ge[e];
                    }
                }
            }
            if (parts > 1) {                               // (uniform)
                s_o[tid] = make_float4(oe[0], oe[1], oe[2], oe[3]);
                s_d[tid] = dp;
                __syncthreads();
                if (part == 0 && act) {
                    for (int q = 1; q < parts; ++q) {
                        const float4 o = s_o[q * lanes + v0];
                        oe[0] += o.x; oe[1] += o.y; oe[2] += o.z; oe[3] += o.w;
                        dp += s_d[q * lanes + v0];
                    }
                }
                __syncthreads();
            }
            if (part == 0 && act) {
                dot += dp;
                if (!member) {
                    float* gp = d.dW + (k * d.sk + c0 * d.sc + r * d.sr + sx * d.ss);
                    if (ow && d.sc == 1 && c0 + 4 <= d.C && ((uintptr_t)gp & 15) == 0) {
                        *reinterpret_cast<float4*>(gp) = make_float4(oe[0], oe[1], oe[2], oe[3]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (c0 + e >= d.C) continue;
                            float* q = gp + e * d.sc;
                            if (d.atomic == 1) atomicAdd(q, oe[e]);
                            else if (ow) *q = oe[e];
                            else *q += oe[e];
                        }
                    }
                }
            }
        }
    }
    if (d.dgamma || d.dbeta || d.dconv_bias) {             // (uniform per block)
        __shared__ float red[4];
        __shared__ float cred[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o, 64);
        if ((tid & 63) == 0) red[tid >> 6] = dot;
        const float cs = pg_colsum(d, k, tid, cred);
        // dbeta is the layer's own BN beta, or (BN-less layer) the conv bias, which the chain shares like dW
        const bool beta_shared = d.gamma == nullptr;
        float bsum = cs, csum = sck * cs;
        if (!member) {
            for (int nx = d.share_next; nx != 0;) {
                const Mi355LayerDesc& m = table[nx - 1];
                const float mc = pg_colsum(m, k, tid, cred);
                bsum += mc;
                csum += pg_scale(m, k) * mc;
                nx = m.share_next;
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (d.dgamma) {
                const float invstd = 1.0f / sqrtf(d.var[k] + d.eps);
                const float mu = d.mean[k] - (d.conv_bias ? d.conv_bias[k] : 0.0f);
                const float dg = invstd * ((red[0] + red[1] + red[2] + red[3]) - mu * cs);
                if (ow) d.dgamma[k] = dg; else d.dgamma[k] += dg;
            }
            if (d.dbeta) {
                if (!beta_shared) { if (ow) d.dbeta[k] = cs; else d.dbeta[k] += cs; }
                else if (!member) {
                    if (d.atomic == 1) atomicAdd(d.dbeta + k, bsum);
                    else if (ow) d.dbeta[k] = bsum;
                    else d.dbeta[k] += bsum;
                }
            }
            if (d.dconv_bias && !member) {
                if (d.atomic == 1) atomicAdd(d.dconv_bias + k, csum);
                else if (ow) d.dconv_bias[k] = csum;
                else d.dconv_bias[k] += csum;
            }
        }
    }
}

extern "C" int mi355_param_grads_table(const Mi355LayerDesc* d_table, int nlayers, int nblocks, void* stream) {
    MI355_REQUIRE(d_table && nlayers > 0 && nblocks > 0, "param_grads_table: bad args");
    hipLaunchKernelGGL(param_grads_table_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream,
                       d_table, nlayers);
    return check_launch("param_grads_table_kernel");
}
