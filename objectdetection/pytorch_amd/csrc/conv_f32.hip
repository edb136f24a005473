// fp32-accurate execution mode of the conv engine (MI355_CONV_PRECISION=fp32): the same grouped
// "gather conv" / weight-gradient problem descriptors as conv_gather.hip / conv_wgrad.hip, but every
// activation, activation-gradient and weight shadow is fp32 and the contraction runs on the exact-f32
// matrix instruction v_mfma_f32_16x16x4_f32 (bitwise a k-ordered fmaf chain, 157 TFLOP/s chip peak:
// MI355X_MICROARCH.md "Matrix cores").  The reference trains in fp32 (train.py:259-321, no autocast):
// this mode is the one that meets the north-star "losses within 1e-4 relative" bound end to end; the
// bf16 mode is the throughput mode.
//
//   out[m][k] = act( sum_{tap,c} in[m @ tap][c] * w[k][tap][c] + bias[k] + addend[m][k] ) * (mask>0)
//   dw[k][tap][c] = sum_m dy[m][k] * in[m @ tap][c]        (per pixel chunk, slabs or atomics)
//
// Tiles: 64 (pixels) x 64 (channels) x 16 (reduction) per 256-thread block, 2 x 2 waves of 32 x 32
// (2 x 2 MFMA fragments); operands go global -> registers -> LDS (register prefetch of the next
// k-step while the current one is on the matrix cores).  Replaces the same reference calls as the
// bf16 kernels: architectures/retinanet.py:48-69, :92-115, :142-178, :272-287 and
// architectures/utils_resnet.py:25-41, :60-80 and their autograd gradients.
#include "conv_common.h"

using namespace mi355;

namespace {

constexpr int FM = 64, FN = 64, FK = 16;
constexpr int LDA = FK + 1;                  // floats per staged row ([row][k] image, odd stride)

struct FProb {
    const float* in;
    float* out;
    const float* addend;
    const float* mask;
    int N, H, W, C, P, Q, sh, sw;
    int TR, TS, dh0, dhs, dw0, dws;
    int r0, rs, s0, ss, S;
    int out_mh, out_oh, out_mw, out_ow;
    long long out_sN, out_sH, out_sW;
    long long add_sN, add_sH, add_sW;
    long long in_sN;
    int add_shift;
    int tile_begin;
    int M;
    FastDiv dPQ, dQ;
};

struct FLaunch {
    FProb p[MI355_CONV_MAX_PROBLEMS];
    const float* w;
    const float* bias;
    int nprob, K, T, act, n_tiles, flags;
};

__device__ __forceinline__ float activate_f32(float v, int act) {
    if (act == 1) return fmaxf(v, 0.0f);
    if (act == 2) return 1.0f / (1.0f + expf(-v));
    return v;
}

__global__ __launch_bounds__(256) void conv_gather_f32_kernel(const FLaunch L) {
    __shared__ float Xs[FM * LDA];
    __shared__ float Ws[FN * LDA];
    __shared__ long long s_outoff[FM];
    __shared__ long long s_addoff[FM];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile = blockIdx.x;
    const int m_tile = tile / L.n_tiles, n_tile = tile - m_tile * L.n_tiles;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MI355_CONV_MAX_PROBLEMS; ++i)
        if (i < L.nprob && m_tile >= L.p[i].tile_begin) pi = i;
    const FProb P = L.p[pi];
    const int m0 = (m_tile - P.tile_begin) * FM;
    const int n0 = n_tile * FN;
    const int PQ = P.P * P.Q;

    if (tid < FM) {
        const int m = m0 + tid;
        long long oo = -1, ao = -1;
        if (m < P.M) {
            const int n = (int)fdiv((unsigned)m, P.dPQ), rem = m - n * PQ;
            const int p = (int)fdiv((unsigned)rem, P.dQ), q = rem - p * P.Q;
            oo = n * P.out_sN + (long long)(p * P.out_mh + P.out_oh) * P.out_sH +
                 (long long)(q * P.out_mw + P.out_ow) * P.out_sW;
            ao = n * P.add_sN + (long long)(p >> P.add_shift) * P.add_sH +
                 (long long)(q >> P.add_shift) * P.add_sW;
        }
        s_outoff[tid] = oo;
        s_addoff[tid] = ao;
    }
    // The epilogue reads the offsets from every wave.  The barriers of the k loop used to be the only ones in front of
    // it, and a problem WITHOUT taps (the three empty parity classes of a 1x1 / stride-2 data gradient: dx = addend)
    // has no k step: waves 1 .. 3 then read whatever the LDS held - offsets of the previous tile when the kernel ran
    // alone (mostly harmless), another kernel's floats when the weight-gradient stream shared the CU: stores to wild
    // addresses, the intermittent GPU memory fault of the fp32 mode (round 2's unexplained abort; found with
    // MI355_TRACE at bwd[55] of the R50 plan, the data gradient of layer4.0.downsample).
    __syncthreads();

    // staging: thread = (row = tid / 4, 4-float chunk = tid % 4) of both tiles
    const int srow = tid >> 2, chunk = tid & 3;
    long long x_base = -1;
    int x_ih0 = 0, x_iw0 = 0;
    {
        const int m = m0 + srow;
        if (m < P.M) {
            const int n = (int)fdiv((unsigned)m, P.dPQ), rem = m - n * PQ;
            const int p = (int)fdiv((unsigned)rem, P.dQ), q = rem - p * P.Q;
            x_ih0 = p * P.sh + P.dh0;
            x_iw0 = q * P.sw + P.dw0;
            x_base = n * P.in_sN;
        }
    }
    const int kw = n0 + srow;
    const long long w_base = kw < L.K ? (long long)kw * L.T * P.C : -1;
    const int cq = P.C >> 2;                     // 4-float chunks per tap
    const int ntaps = P.TR * P.TS;
    const int nred = ntaps * cq;                 // reduction length in 4-float chunks
    const int ksteps = (nred + 3) >> 2;

    float4 xr, wr;
    auto fetch = [&](int ks) {
        const int J = ks * 4 + chunk;
        xr = make_float4(0.f, 0.f, 0.f, 0.f);
        wr = xr;
        if (J < nred) {
            const int tap = J / cq, c = (J - tap * cq) * 4;
            const int ti = tap / P.TS, tj = tap - ti * P.TS;
            const int ih = x_ih0 + ti * P.dhs, iw = x_iw0 + tj * P.dws;
            if (x_base >= 0 && (unsigned)ih < (unsigned)P.H && (unsigned)iw < (unsigned)P.W)
                xr = *reinterpret_cast<const float4*>(P.in + x_base + ((long long)ih * P.W + iw) * P.C + c);
            if (w_base >= 0)
                wr = *reinterpret_cast<const float4*>(L.w + w_base +
                                                      (long long)((P.r0 + ti * P.rs) * P.S + (P.s0 + tj * P.ss)) * P.C + c);
        }
    };

    const int wm = wv >> 1, wn = wv & 1;
    const int fr = lane & 15, fq = lane >> 4;
    f4v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f4v){0.f, 0.f, 0.f, 0.f};

    if (ksteps > 0) fetch(0);
    for (int ks = 0; ks < ksteps; ++ks) {
        float* xd = Xs + srow * LDA + chunk * 4;
        float* wd = Ws + srow * LDA + chunk * 4;
        xd[0] = xr.x; xd[1] = xr.y; xd[2] = xr.z; xd[3] = xr.w;
        wd[0] = wr.x; wd[1] = wr.y; wd[2] = wr.z; wd[3] = wr.w;
        __syncthreads();
        if (ks + 1 < ksteps) fetch(ks + 1);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            float a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = Ws[(wn * 32 + t * 16 + fr) * LDA + k4 * 4 + fq];     // A operand: rows = output channels
                b[t] = Xs[(wm * 32 + t * 16 + fr) * LDA + k4 * 4 + fq];     // B operand: cols = pixels
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: lane holds pixel (wm*32 + i*16 + fr), channels n0 + wn*32 + j*16 + fq*4 + r
    const bool mask_first = (L.flags & 1) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wm * 32 + i * 16 + fr;
        const long long oo = s_outoff[row];
        if (oo < 0) continue;
        const long long ao = s_addoff[row];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = n0 + wn * 32 + j * 16 + fq * 4 + r;
                if (k >= L.K) continue;
                float v = acc[i][j][r] + (L.bias ? L.bias[k] : 0.0f);
                const float mk = P.mask ? P.mask[oo + k] : 1.0f;
                if (P.mask && mask_first) v = mk > 0.0f ? v : 0.0f;
                if (P.addend) v += P.addend[ao + k];
                v = activate_f32(v, L.act);
                if (P.mask && !mask_first) v = mk > 0.0f ? v : 0.0f;
                P.out[oo + k] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient, fp32: block = (pixel chunk, tap, 64 k, 64 c); 16 pixel rows per stage.
constexpr int WR = 16, WLD = 64 + 16;        // staged rows x (64 + 16) floats: conflict-free b32 fragment reads

struct FWProb {
    const float* in;
    const float* dy;
    int N, H, W, C, P, Q, sh, sw;
    int TR, TS, dh0, dhs, dw0, dws;
    int r0, rs, s0, ss, S;
    int M, chunk_begin;
    long long in_sN, dy_sN;
    FastDiv dPQ, dQ;
};

struct FWLaunch {
    FWProb p[MI355_CONV_MAX_PROBLEMS];
    float* dw;
    float* colsum;
    long long slab_stride;
    int nprob, K, T, k_tiles, c_tiles, ntaps, split_rows;
};

__global__ __launch_bounds__(256) void conv_wgrad_f32_kernel(const FWLaunch L) {
    __shared__ __attribute__((aligned(16))) float Ys[WR * WLD];
    __shared__ __attribute__((aligned(16))) float Xs[WR * WLD];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int b = blockIdx.x;
    const int c_tile = b % L.c_tiles;  b /= L.c_tiles;
    const int k_tile = b % L.k_tiles;  b /= L.k_tiles;
    const int tap = b % L.ntaps;       b /= L.ntaps;
    const int chunk = b;
    FWProb P = L.p[0];
#pragma unroll
    for (int i = 1; i < MI355_CONV_MAX_PROBLEMS; ++i)
        if (i < L.nprob && chunk >= L.p[i].chunk_begin) P = L.p[i];
    const int m_begin = (chunk - P.chunk_begin) * L.split_rows;
    const int m_end = min(P.M, m_begin + L.split_rows);
    const int k0 = k_tile * 64, c0 = c_tile * 64;
    const int ti = tap / P.TS, tj = tap - ti * P.TS;
    const int dh = P.dh0 + ti * P.dhs, dw = P.dw0 + tj * P.dws;
    const int PQ = P.P * P.Q;

    const int srow = tid >> 4, c4 = (tid & 15) * 4;
    const bool ky_ok = k0 + c4 + 4 <= L.K;
    const bool cx_ok = c0 + c4 + 4 <= P.C;
    float4 yr, xr;
    auto fetch = [&](int step) {
        const int m = m_begin + step * WR + srow;
        yr = make_float4(0.f, 0.f, 0.f, 0.f);
        xr = yr;
        if (m < m_end) {
            const int n = (int)fdiv((unsigned)m, P.dPQ), rem = m - n * PQ;
            const int p = (int)fdiv((unsigned)rem, P.dQ), q = rem - p * P.Q;
            if (ky_ok) yr = *reinterpret_cast<const float4*>(P.dy + n * P.dy_sN + (long long)rem * L.K + k0 + c4);
            const int ih = p * P.sh + dh, iw = q * P.sw + dw;
            if (cx_ok && (unsigned)ih < (unsigned)P.H && (unsigned)iw < (unsigned)P.W)
                xr = *reinterpret_cast<const float4*>(P.in + n * P.in_sN + ((long long)ih * P.W + iw) * P.C + c0 + c4);
        }
    };

    const int wk = wv >> 1, wc = wv & 1;
    const int fr = lane & 15, fq = lane >> 4;
    f4v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f4v){0.f, 0.f, 0.f, 0.f};
    const bool do_colsum = L.colsum != nullptr && c_tile == 0 && tap == 0 && tid < 64;
    float csum = 0.0f;
    const int nsteps = (m_end - m_begin + WR - 1) / WR;
    if (nsteps > 0) fetch(0);
    for (int step = 0; step < nsteps; ++step) {
        *reinterpret_cast<float4*>(Ys + srow * WLD + c4) = yr;
        *reinterpret_cast<float4*>(Xs + srow * WLD + c4) = xr;
        __syncthreads();
        if (step + 1 < nsteps) fetch(step + 1);
        if (do_colsum) {
#pragma unroll
            for (int r = 0; r < WR; ++r) csum += Ys[r * WLD + tid];
        }
#pragma unroll
        for (int m4 = 0; m4 < WR / 4; ++m4) {
            float a[2], bb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = Ys[(m4 * 4 + fq) * WLD + wk * 32 + t * 16 + fr];    // A: rows = k, reduction = pixel
                bb[t] = Xs[(m4 * 4 + fq) * WLD + wc * 32 + t * 16 + fr];   // B: cols = c
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    if (do_colsum && k0 + tid < L.K) {
        if (L.slab_stride > 0) L.colsum[(long long)chunk * L.K + k0 + tid] = csum;
        else atomicAdd(L.colsum + k0 + tid, csum);
    }
    const long long wrow = (long long)L.T * P.C;
    const long long toff = (long long)((P.r0 + ti * P.rs) * P.S + (P.s0 + tj * P.ss)) * P.C;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wc * 32 + j * 16 + fr;
            if (c >= P.C) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + wk * 32 + i * 16 + fq * 4 + r;
                if (k >= L.K) continue;
                float* dst = L.dw + k * wrow + toff + c;
                if (L.slab_stride > 0) dst[(long long)chunk * L.slab_stride] = acc[i][j][r];
                else atomicAdd(dst, acc[i][j][r]);
            }
        }
}

// ---------------------------------------------------------------------------------------------
// HBM-bound helpers of the fp32 mode (conv_aux.hip holds the bf16 forms)
__global__ void nchw_to_nhwc_f32_kernel(const float* __restrict__ in, float* __restrict__ out, long long npix,
                                        int C, long long HW, int Cpad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const long long n = i / HW, hw = i - n * HW;
    const float* src = in + n * C * HW + hw;
    float* dst = out + i * Cpad;
    for (int c = 0; c < Cpad; ++c) dst[c] = c < C ? src[c * HW] : 0.0f;
}

__global__ void maxpool_fwd_f32_kernel(const float* __restrict__ in, float* __restrict__ out,
                                       signed char* __restrict__ idx, int N, int H, int W, int C, int P, int Q) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * P * Q * C) return;
    const int c = (int)(t % C);
    long long r = t / C;
    const int q = (int)(r % Q);  r /= Q;
    const int p = (int)(r % P);
    const int n = (int)(r / P);
    float best = -INFINITY;
    int bi = -1;
    for (int a = 0; a < 3; ++a) {
        const int h = 2 * p - 1 + a;
        if ((unsigned)h >= (unsigned)H) continue;
        for (int bq = 0; bq < 3; ++bq) {
            const int w = 2 * q - 1 + bq;
            if ((unsigned)w >= (unsigned)W) continue;
            const float f = in[(((long long)n * H + h) * W + w) * C + c];
            if (f > best || bi < 0) { best = f; bi = a * 3 + bq; }      // first max wins (torch CPU)
        }
    }
    out[t] = best;
    idx[t] = (signed char)bi;
}

template <bool POOLED>                                  // relu_mask at the pool's output resolution (see conv_aux.hip)
__global__ void maxpool_bwd_f32_kernel(const float* __restrict__ gout, const signed char* __restrict__ idx,
                                       float* __restrict__ gin, const float* __restrict__ relu_mask,
                                       int N, int H, int W, int C, int P, int Q) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * H * W * C) return;
    const int c = (int)(t % C);
    long long r = t / C;
    const int w = (int)(r % W);  r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    float acc = 0.0f;
    // windows (p, q) covering (h, w): 2p-1 <= h <= 2p+1
    for (int p = (h + 0) / 2; p <= (h + 1) / 2; ++p) {
        if (p >= P) continue;
        const int wr = h - (2 * p - 1);
        if (wr < 0 || wr > 2) continue;
        for (int q = (w + 0) / 2; q <= (w + 1) / 2; ++q) {
            if (q >= Q) continue;
            const int wc = w - (2 * q - 1);
            if (wc < 0 || wc > 2) continue;
            const long long o = (((long long)n * P + p) * Q + q) * C + c;
            if ((int)idx[o] == wr * 3 + wc && (!POOLED || relu_mask[o] > 0.0f)) acc += gout[o];
        }
    }
    if (!POOLED && relu_mask && !(relu_mask[t] > 0.0f)) acc = 0.0f;
    gin[t] = acc;
}

__global__ void sumpool2x2_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int H, int W,
                                      int C, int P, int Q, int accumulate) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * P * Q * C) return;
    const int c = (int)(t % C);
    long long r = t / C;
    const int q = (int)(r % Q);  r /= Q;
    const int p = (int)(r % P);
    const int n = (int)(r / P);
    float acc = accumulate ? dst[t] : 0.0f;
    for (int a = 0; a < 2; ++a)
        for (int bq = 0; bq < 2; ++bq) {
            const int h = 2 * p + a, w = 2 * q + bq;
            if (h >= H || w >= W) continue;
            acc += src[(((long long)n * H + h) * W + w) * C + c];
        }
    dst[t] = acc;
}

__global__ void relu_mask_f32_kernel(const float* __restrict__ x, const float* __restrict__ ref, float* __restrict__ y,
                                     long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = ref[i] > 0.0f ? x[i] : 0.0f;
}

}  // namespace

extern "C" int mi355_conv_gather_f32(const Mi355ConvLaunch* launch, void* stream) {
    MI355_REQUIRE(launch, "conv_gather_f32: null launch");
    const Mi355ConvLaunch& in = *launch;
    MI355_REQUIRE(in.nprob >= 1 && in.nprob <= MI355_CONV_MAX_PROBLEMS, "conv_gather_f32: nprob %d", in.nprob);
    MI355_REQUIRE(in.K > 0 && in.weight && in.T > 0, "conv_gather_f32: bad K/weight/T");
    FLaunch L;
    memset(&L, 0, sizeof(L));
    const int C = in.prob[0].C;
    MI355_REQUIRE(C > 0 && C % 4 == 0, "conv_gather_f32: C must be a multiple of 4");
    int tiles = 0;
    for (int i = 0; i < in.nprob; ++i) {
        const Mi355ConvProblem& s = in.prob[i];
        MI355_REQUIRE(!s.weight, "conv_gather_f32: per-problem weights are a bf16-launch feature");
        MI355_REQUIRE(s.in && s.out, "conv_gather_f32: problem %d null in/out", i);
        MI355_REQUIRE(s.C == C, "conv_gather_f32: C must be equal across problems");
        MI355_REQUIRE(s.N > 0 && s.H > 0 && s.W > 0 && s.P > 0 && s.Q > 0, "conv_gather_f32: problem %d bad dims", i);
        MI355_REQUIRE((long long)s.N * s.P * s.Q < 0x7fffffffLL, "conv_gather_f32: M too large");
        MI355_REQUIRE(((uintptr_t)s.in & 15) == 0 && ((uintptr_t)s.out & 3) == 0, "conv_gather_f32: alignment");
        FProb& d = L.p[i];
        d.in = (const float*)s.in;  d.out = (float*)s.out;
        d.addend = (const float*)s.addend;  d.mask = (const float*)s.mask;
        d.N = s.N; d.H = s.H; d.W = s.W; d.C = s.C; d.P = s.P; d.Q = s.Q; d.sh = s.sh; d.sw = s.sw;
        d.TR = s.TR; d.TS = s.TS; d.dh0 = s.dh0; d.dhs = s.dhs; d.dw0 = s.dw0; d.dws = s.dws;
        d.r0 = s.r0; d.rs = s.rs; d.s0 = s.s0; d.ss = s.ss; d.S = s.S;
        d.out_mh = s.out_mh; d.out_oh = s.out_oh; d.out_mw = s.out_mw; d.out_ow = s.out_ow;
        d.out_sN = s.out_sN; d.out_sH = s.out_sH; d.out_sW = s.out_sW;
        d.add_sN = s.add_sN; d.add_sH = s.add_sH; d.add_sW = s.add_sW; d.add_shift = s.add_shift;
        d.in_sN = s.in_sN ? s.in_sN : (long long)s.H * s.W * s.C;
        d.M = s.N * s.P * s.Q;
        d.dPQ = make_fastdiv((unsigned)(s.P * s.Q));
        d.dQ = make_fastdiv((unsigned)s.Q);
        d.tile_begin = tiles;
        tiles += (int)cdiv(d.M, FM);
    }
    MI355_REQUIRE(((uintptr_t)in.weight & 15) == 0, "conv_gather_f32: weight alignment");
    L.w = (const float*)in.weight;
    L.bias = in.bias;
    L.nprob = in.nprob; L.K = in.K; L.T = in.T; L.act = in.act; L.flags = in.flags;
    L.n_tiles = (int)cdiv(in.K, FN);
    const long long grid = (long long)tiles * L.n_tiles;
    MI355_REQUIRE(grid > 0 && grid < 0x7fffffffLL, "conv_gather_f32: grid");
    hipLaunchKernelGGL(conv_gather_f32_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, L);
    return check_launch("conv_gather_f32_kernel");
}

extern "C" int mi355_conv_wgrad_slabs_f32(const Mi355ConvLaunch* launch, float* dw, int64_t slab_stride,
                                          int max_slabs, float* colsum, int split_rows, void* stream) {
    MI355_REQUIRE(launch && dw, "conv_wgrad_f32: null argument");
    const Mi355ConvLaunch& in = *launch;
    MI355_REQUIRE(in.nprob >= 1 && in.nprob <= MI355_CONV_MAX_PROBLEMS, "conv_wgrad_f32: nprob");
    MI355_REQUIRE(in.K > 0 && in.K % 4 == 0 && in.T > 0, "conv_wgrad_f32: K must be a positive multiple of 4");
    MI355_REQUIRE(split_rows > 0 && split_rows % WR == 0, "conv_wgrad_f32: split_rows must be a multiple of 16");
    FWLaunch L;
    memset(&L, 0, sizeof(L));
    const int C = in.prob[0].C;
    const int ntaps = in.prob[0].TR * in.prob[0].TS;
    int chunks = 0;
    for (int i = 0; i < in.nprob; ++i) {
        const Mi355ConvProblem& s = in.prob[i];
        MI355_REQUIRE(s.in && s.out, "conv_wgrad_f32: problem %d null in/dy", i);
        MI355_REQUIRE(s.C == C && C % 4 == 0 && s.TR * s.TS == ntaps, "conv_wgrad_f32: C%%4, taps equal across problems");
        MI355_REQUIRE(((uintptr_t)s.in & 15) == 0 && ((uintptr_t)s.out & 15) == 0, "conv_wgrad_f32: alignment");
        const long long M = (long long)s.N * s.P * s.Q;
        MI355_REQUIRE(M > 0 && M < 0x7fffffffLL, "conv_wgrad_f32: M");
        FWProb& d = L.p[i];
        d.in = (const float*)s.in;  d.dy = (const float*)s.out;
        d.N = s.N; d.H = s.H; d.W = s.W; d.C = s.C; d.P = s.P; d.Q = s.Q; d.sh = s.sh; d.sw = s.sw;
        d.TR = s.TR; d.TS = s.TS; d.dh0 = s.dh0; d.dhs = s.dhs; d.dw0 = s.dw0; d.dws = s.dws;
        d.r0 = s.r0; d.rs = s.rs; d.s0 = s.s0; d.ss = s.ss; d.S = s.S;
        d.M = (int)M;
        d.in_sN = s.in_sN ? s.in_sN : (long long)s.H * s.W * s.C;
        d.dy_sN = s.out_sN ? s.out_sN : (long long)s.P * s.Q * in.K;
        d.dPQ = make_fastdiv((unsigned)(s.P * s.Q));
        d.dQ = make_fastdiv((unsigned)s.Q);
        d.chunk_begin = chunks;
        chunks += (int)cdiv(M, split_rows);
    }
    if (ntaps == 0) return MI355_OK;
    MI355_REQUIRE(slab_stride == 0 || chunks <= max_slabs, "conv_wgrad_f32: %d slabs > workspace for %d", chunks, max_slabs);
    L.dw = dw;
    L.colsum = colsum;
    L.slab_stride = slab_stride;
    L.nprob = in.nprob; L.K = in.K; L.T = in.T; L.ntaps = ntaps; L.split_rows = split_rows;
    L.k_tiles = (int)cdiv(in.K, 64);
    L.c_tiles = (int)cdiv(C, 64);
    const long long grid = (long long)chunks * ntaps * L.k_tiles * L.c_tiles;
    MI355_REQUIRE(grid > 0 && grid < 0x7fffffffLL, "conv_wgrad_f32: grid");
    hipLaunchKernelGGL(conv_wgrad_f32_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, L);
    return check_launch("conv_wgrad_f32_kernel");
}

extern "C" int mi355_nchw_to_nhwc_f32(const float* in, float* out, int N, int C, int H, int W, int Cpad, void* stream) {
    MI355_REQUIRE(in && out && N > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "nchw_to_nhwc_f32: bad args");
    const long long npix = (long long)N * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel, dim3((unsigned)cdiv(npix, 256)), dim3(256), 0, (hipStream_t)stream,
                       in, out, npix, C, (long long)H * W, Cpad);
    return check_launch("nchw_to_nhwc_f32_kernel");
}

extern "C" int mi355_maxpool3x3s2_fwd_f32(const float* in, float* out, int8_t* idx, int N, int H, int W, int C,
                                          void* stream) {
    MI355_REQUIRE(in && out && idx && N > 0 && H > 0 && W > 0 && C > 0, "maxpool_fwd_f32: bad args");
    const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * P * Q * C;
    hipLaunchKernelGGL(maxpool_fwd_f32_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       in, out, (signed char*)idx, N, H, W, C, P, Q);
    return check_launch("maxpool_fwd_f32_kernel");
}

extern "C" int mi355_maxpool3x3s2_bwd_f32(const float* gout, const int8_t* idx, float* gin, const float* relu_mask,
                                          int N, int H, int W, int C, void* stream) {
    MI355_REQUIRE(gout && idx && gin && N > 0 && H > 0 && W > 0 && C > 0, "maxpool_bwd_f32: bad args");
    const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * H * W * C;
    hipLaunchKernelGGL(maxpool_bwd_f32_kernel<false>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       gout, (const signed char*)idx, gin, relu_mask, N, H, W, C, P, Q);
    return check_launch("maxpool_bwd_f32_kernel");
}

extern "C" int mi355_maxpool3x3s2_bwd_pooled_f32(const float* gout, const int8_t* idx, float* gin, const float* pooled,
                                          int N, int H, int W, int C, void* stream) {
    MI355_REQUIRE(gout && idx && gin && pooled && N > 0 && H > 0 && W > 0 && C > 0, "maxpool_bwd_pooled_f32: bad args");
    const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * H * W * C;
    hipLaunchKernelGGL(maxpool_bwd_f32_kernel<true>, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       gout, (const signed char*)idx, gin, pooled, N, H, W, C, P, Q);
    return check_launch("maxpool_bwd_f32_kernel");
}

extern "C" int mi355_sumpool2x2_f32(const float* src, float* dst, int N, int H, int W, int C, int accumulate,
                                    void* stream) {
    MI355_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && C > 0, "sumpool2x2_f32: bad args");
    const int P = (H + 1) / 2, Q = (W + 1) / 2;
    const long long total = (long long)N * P * Q * C;
    hipLaunchKernelGGL(sumpool2x2_f32_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       src, dst, N, H, W, C, P, Q, accumulate);
    return check_launch("sumpool2x2_f32_kernel");
}

extern "C" int mi355_relu_mask_f32(const float* x, const float* ref, float* y, int64_t n, void* stream) {
    MI355_REQUIRE(x && ref && y && n > 0, "relu_mask_f32: bad args");
    hipLaunchKernelGGL(relu_mask_f32_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, ref, y, (long long)n);
    return check_launch("relu_mask_f32_kernel");
}
