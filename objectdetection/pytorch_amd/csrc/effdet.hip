// EfficientDet-specific HBM-bound kernels for gfx950: depthwise conv (TF-"same", fused frozen-BN
// affine, swish applied on load), squeeze-excite, BiFPN fast-attention nodes, zero-padded "same"
// max-pool.  NHWC bf16, 16 B (8 channels) per lane everywhere, fp32 math.
//
// Reference arithmetic: architectures/efficientnet.py:80-115, architectures/efficientdet.py:28-66,
// :195-267, architectures/utils_extra_efficientnet.py:9-96, architectures/utils_efficientnet.py:38-49.
#include "conv_common.h"
#include "mi355det_effdet.h"

using namespace mi355;

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float swishf_exact(float x) { return x * sigmoidf_(x); }          // fp32 SE vectors
__device__ __forceinline__ float dswishf_exact(float x) {    // utils_efficientnet.py:46-49
    const float s = sigmoidf_(x);
    return s * (1.0f + x * (1.0f - s));
}
// Activation maps are bf16: v_exp_f32 + v_rcp_f32 (1-2 ulp in fp32, far below the bf16 rounding that follows)
// instead of the accurate expf + IEEE division (~3x the instructions); every kernel that turns a stored
// pre-activation z into swish(z) uses this one function, so the forward, the weight gradient and a materialised
// copy see the same bf16 values.
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float swishf(float x) { return x * fast_sigmoid(x); }
__device__ __forceinline__ float dswishf(float x) {
    const float s = fast_sigmoid(x);
    return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ void load8(const unsigned short* p, float v[8]) {
    const s8v t = *reinterpret_cast<const s8v*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf16_to_f32((unsigned short)t[e]);
}
__device__ __forceinline__ void store8(unsigned short* p, const float v[8]) {
    s8v t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (short)f32_to_bf16_rn(v[e]);
    *reinterpret_cast<s8v*>(p) = t;
}
// fp32-accurate mode (MI355_CONV_PRECISION=fp32, RetinaPlan.prec): the generic (untiled) kernels below are templates
// over the storage type; fp32 tensors take the exact expf / IEEE-division swish
__device__ __forceinline__ void load8(const float* p, float v[8]) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(float* p, const float v[8]) {
    reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
template <typename T> struct Act;
template <> struct Act<unsigned short> {
    static __device__ __forceinline__ float sw(float x) { return swishf(x); }
    static __device__ __forceinline__ float dsw(float x) { return dswishf(x); }
};
template <> struct Act<float> {
    static __device__ __forceinline__ float sw(float x) { return swishf_exact(x); }
    static __device__ __forceinline__ float dsw(float x) { return dswishf_exact(x); }
};
inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }
// squeeze-excite backward, d/dz of gate * swish(z) with the pooled term: one explicit fma wherever it is evaluated
__device__ __forceinline__ float se_apply_mix(float gu, float gate, float gpool, float inv_hw, float dsw) {
    return fmaf(gu, gate, gpool * inv_hw) * dsw;
}
// BiFPN fast-attention mix, one explicit fma chain wherever a node is evaluated (bit-identical z in every kernel)
__device__ __forceinline__ float node_mix(float w0, float a, float w1, float b, float w2, float c) {
    return fmaf(w2, c, fmaf(w1, b, w0 * a));
}

}  // namespace

// ============================================================================ depthwise conv
__global__ void dw_prep_weight_kernel(const float* __restrict__ w, const float* __restrict__ scale, int C,
                                      int T, float* __restrict__ w_t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * T) return;
    const int t = i / C, c = i - t * C;
    w_t[i] = w[c * T + t] * (scale ? scale[c] : 1.0f);
}

extern "C" int mi355_dw_prep_weight(const float* w, const float* scale, int C, int k, float* w_t, void* stream) {
    MI355_REQUIRE(w && w_t && C > 0 && k > 0, "dw_prep_weight: bad args");
    hipLaunchKernelGGL(dw_prep_weight_kernel, dim3(blocks_for((long long)C * k * k)), dim3(256), 0,
                       (hipStream_t)stream, w, scale, C, k * k, w_t);
    return check_launch("dw_prep_weight_kernel");
}

// Every depthwise layer of a plan in one launch (blockIdx.y = table row): frozen-BN fold (same arithmetic as
// bn_fold_kernel, conv_aux.hip) + transposed, scaled taps.  96 tiny launches at the start of an EfficientDet-D0 step otherwise.
__global__ void dw_prep_table_kernel(const Mi355DwPrepDesc* __restrict__ table) {
    const Mi355DwPrepDesc d = table[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.C * d.T) return;
    const int t = i / d.C, c = i - t * d.C;
    float sc = 1.0f;
    if (d.gamma) {
        sc = d.gamma[c] / sqrtf(d.var[c] + d.eps);
        if (t == 0) {
            d.scale[c] = sc;
            d.shift[c] = d.beta[c] - d.mean[c] * sc;
        }
    }
    d.w_t[i] = d.w[c * d.T + t] * sc;
}

extern "C" int mi355_dw_prep_table(const Mi355DwPrepDesc* table, int n, int max_elems, void* stream) {
    MI355_REQUIRE(table && n > 0 && n <= 65535 && max_elems > 0, "dw_prep_table: bad args");
    hipLaunchKernelGGL(dw_prep_table_kernel, dim3(blocks_for(max_elems), (unsigned)n), dim3(256), 0,
                       (hipStream_t)stream, table);
    return check_launch("dw_prep_table_kernel");
}

template <typename T, bool SWISH>
__global__ void dwconv_fwd_kernel(const T* __restrict__ in, const float* __restrict__ w_t,
                                  const float* __restrict__ bias, T* __restrict__ out, int N, int H,
                                  int W, int C, int k, int stride, int pad_t, int pad_l, int P, int Q) {
    const int c8n = C >> 3;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * P * Q * c8n) return;
    int c8, q, p, n;
    split_cell(t, c8n, Q, P, c8, q, p, n);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bias ? bias[c8 * 8 + e] : 0.0f;
    for (int a = 0; a < k; ++a) {
        const int h = p * stride - pad_t + a;
        if ((unsigned)h >= (unsigned)H) continue;
        for (int b = 0; b < k; ++b) {
            const int w = q * stride - pad_l + b;
            if ((unsigned)w >= (unsigned)W) continue;
            float x[8];
            load8(in + (((long long)n * H + h) * W + w) * C + c8 * 8, x);
            const float* wt = w_t + (a * k + b) * C + c8 * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (SWISH ? Act<T>::sw(x[e]) : x[e]) * wt[e];
        }
    }
    store8(out + t * 8, acc);
}


// Stride-1 form of the forward: a thread produces FOUR neighbouring output pixels of one 8-channel chunk, so a
// kernel row costs 4 + K - 1 input loads for 4 K taps (K = 5: 10 loads per output instead of 25).
template <int K>
__global__ __launch_bounds__(256) void dwconv_fwd_s1_kernel(const unsigned short* __restrict__ in, const float* __restrict__ w_t,
                                                            const float* __restrict__ bias, unsigned short* __restrict__ out,
                                                            int N, int H, int W, int C, int pad_t, int pad_l, int P, int Q) {
    const int c8n = C >> 3, Q4 = (Q + 3) >> 2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * P * Q4 * c8n) return;
    int c8, q0, p, n;
    split_cell(t, c8n, Q4, P, c8, q0, p, n);
    q0 *= 4;
    float acc[4][8];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[o][e] = bias ? bias[c8 * 8 + e] : 0.0f;
#pragma unroll
    for (int a = 0; a < K; ++a) {
        const int h = p - pad_t + a;
        if ((unsigned)h >= (unsigned)H) continue;
        const unsigned short* row = in + ((long long)n * H + h) * W * C + c8 * 8;
        float xw[4 + K - 1][8];
#pragma unroll
        for (int j = 0; j < 4 + K - 1; ++j) {
            const int w = q0 - pad_l + j;
            if ((unsigned)w < (unsigned)W) load8(row + (long long)w * C, xw[j]);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) xw[j][e] = 0.0f;
            }
        }
#pragma unroll
        for (int b = 0; b < K; ++b) {
            float wt[8];
            const float4 w0 = *reinterpret_cast<const float4*>(w_t + (a * K + b) * C + c8 * 8);
            const float4 w1 = *reinterpret_cast<const float4*>(w_t + (a * K + b) * C + c8 * 8 + 4);
            wt[0] = w0.x; wt[1] = w0.y; wt[2] = w0.z; wt[3] = w0.w; wt[4] = w1.x; wt[5] = w1.y; wt[6] = w1.z; wt[7] = w1.w;
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[o][e] += xw[o + b][e] * wt[e];
        }
    }
    unsigned short* op = out + (((long long)n * P + p) * Q + q0) * C + c8 * 8;
#pragma unroll
    for (int o = 0; o < 4; ++o)
        if (q0 + o < Q) store8(op + (long long)o * C, acc[o]);
}


// LDS-tiled stride-1 "same" depthwise conv, used for the forward (FLIP = false) and for the data gradient
// (FLIP = true: correlation of dy with the flipped taps; epilogue x swish'(z), += dx).  A block owns an
// 8 x 32 output tile of NC8 8-channel chunks of one image: the (8 + K - 1) x (32 + K - 1) input halo tile is
// fetched ONCE (the direct kernels fetch every input K*K times through L1/L2), kept as [chunk][pixel] planes of
// 16-B cells (consecutive lanes = consecutive cells: conflict-free ds_read_b128), the taps' weights sit in
// LDS too, and the outputs go back through LDS so that a pixel's NC8 chunks are stored contiguously.
constexpr int DT_TH = 8, DT_TW = 32, DT_OS = DT_TH * DT_TW + 1;
// One launch covers up to MI355_DW_MAX_LEVELS feature maps that share the weights (the pyramid levels of an
// EfficientDet head conv): blockIdx.x walks the 8 x 32 tiles of all the maps, blockIdx.y = (image, chunk group).
struct DwTileLevel {
    const unsigned short* in;
    unsigned short* out;
    const unsigned short* zin;
    int H, W, tilesX, tile_begin;
    int P, Q;                      // output map (== H, W for the stride-1 "same" forms)
};
struct DwTileLaunch {
    DwTileLevel lv[MI355_DW_MAX_LEVELS];
    int nlevels, C, nc8, pad_t, pad_l, accumulate, xs;
    // NODE form (one level): the conv's input is the BiFPN fast-attention node z = w[0] in + w[1] in1' + w[2] in2
    // (efficientdet.py:213-265), built while the halo tile is staged and stored to `zout` by the tile that owns the
    // pixel (the backward reads z); in1' = nearest-2x upsample of in1 [N][H/2][W/2][C] when up1
    const unsigned short* in1;
    const unsigned short* in2;
    unsigned short* zout;
    const float* nodew;
    int up1;
    // squeeze-excite pooling of the OUTPUT (forward, one level): pool[(n * tiles + tile) * C + c] = pool_scale * sum over
    // the tile's pixels of swish(out) — the slab partials mi355_se_fc adds in order (efficientnet.py:98-100:
    // adaptive_avg_pool2d of swish(bn1(depthwise_conv(x)))); NULL = none
    float* pool;
    float pool_scale;
    // SEAPPLY form (data gradient, one level): the conv's input dy is the squeeze-excite backward
    // gz = (gu * gate[n][c] + gpool[n][c] / HW) * swish'(z) (mi355_se_bwd_apply), built while the halo tile is staged from
    // `in` = gu and stored to `ap_gz` by the tile that owns the pixel (the weight gradient reads it)
    const unsigned short* ap_z;
    const float* ap_gate;
    const float* ap_gpool;
    unsigned short* ap_gz;
    float ap_inv;
};
// S = 2 (forward only): the halo tile is kept as FOUR parity planes (input row parity, column parity), so that tap
// (a, b) of output column tx reads cell tx + (b >> 1) of plane (a & 1, b & 1): consecutive lanes = consecutive cells, as
// in the stride-1 form (a plain image would be read with stride 2: 2-way bank conflicts); swish once per loaded element
// (the direct stride-2 kernel recomputes it (k / 2)^2 = 2.25 / 6.25 times: it is bound by v_exp / v_rcp, 85 us for the
// 256^2 x 96 layer against a 50 us HBM floor).
template <int K, bool FLIP, bool INSW = false, bool NODE = false, int S = 1, bool SEAPPLY = false>
__global__ __launch_bounds__(256) void dw_tile_s1_kernel(const DwTileLaunch L, const float* __restrict__ w_t,
                                                         const float* __restrict__ bias) {
    constexpr int IH = (DT_TH - 1) * S + K, IW = (DT_TW - 1) * S + K;
    constexpr int PH = (IH + S - 1) / S, PW = (IW + S - 1) / S;          // one parity plane (S == 1: the tile itself)
    constexpr int NPX = S * S * PH * PW;
    static_assert(!(FLIP || NODE) || S == 1, "stride 2: forward only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nc8 = L.nc8, C = L.C, pad_t = L.pad_t, pad_l = L.pad_l, accumulate = L.accumulate;
    s8v* tile = reinterpret_cast<s8v*>(smem);                              // [nc8][NPX]
    float* wl = reinterpret_cast<float*>(smem + (size_t)nc8 * NPX * 16);     // [K*K][nc8*8]
    s8v* ot = reinterpret_cast<s8v*>(smem);                                // [nc8][DT_OS] output staging, aliases `tile`
    const int tid = threadIdx.x;
    // Block order (as in dw_wgrad_tile_kernel): the channel groups of a tile read pieces of the same 128-B lines; in
    // launch order (tiles fastest, then image x group) they run far apart and on different XCDs.  L.xs: XCD x takes a
    // contiguous range of logical ids with the GROUP fastest, so the groups of a tile share one L2 while the lines are
    // resident (a bijection of block ids: results unchanged).
    const int groups = (L.C >> 3) / L.nc8;
    int tile_id = blockIdx.x, img_group = blockIdx.y;
    if (L.xs) {
        const int lg = xcd_swizzle((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
        const int g = lg % groups, rest = lg / groups;
        tile_id = rest % (int)gridDim.x;
        img_group = (rest / (int)gridDim.x) * groups + g;
    }
    DwTileLevel P = L.lv[0];
#pragma unroll
    for (int i = 1; i < MI355_DW_MAX_LEVELS; ++i)
        if (i < L.nlevels && tile_id >= L.lv[i].tile_begin) P = L.lv[i];
    const int tl = tile_id - P.tile_begin;
    const int tyi = tl / P.tilesX, txi = tl - tyi * P.tilesX;
    const int q0 = txi * DT_TW, p0 = tyi * DT_TH;
    const int H = P.H, W = P.W, OP = P.P, OQ = P.Q;
    const unsigned short* __restrict__ in = P.in;
    unsigned short* __restrict__ out = P.out;
    const unsigned short* __restrict__ zin = P.zin;
    const int n = img_group / groups, c8b = (img_group - n * groups) * nc8;
    // ---- input halo tile
    for (int i = tid; i < IH * IW * nc8; i += 256) {
        const int lin = i / nc8, ch = i - lin * nc8;
        const int iy = lin / IW, ix = lin - iy * IW;
        const int px = S == 1 ? lin : (((iy & 1) * 2 + (ix & 1)) * (PH * PW) + (iy >> 1) * PW + (ix >> 1));
        const int h = p0 * S - pad_t + iy, w = q0 * S - pad_l + ix;
        s8v v = (s8v){0, 0, 0, 0, 0, 0, 0, 0};
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
            const long long o = (((long long)n * H + h) * W + w) * C + (c8b + ch) * 8;
            v = *reinterpret_cast<const s8v*>(in + o);
            if (NODE) {
                const long long o1 = L.up1 ? ((((long long)n * (H >> 1) + (h >> 1)) * (W >> 1) + (w >> 1)) * C + (c8b + ch) * 8) : o;
                const s8v b = *reinterpret_cast<const s8v*>(L.in1 + o1);
                s8v c = (s8v){0, 0, 0, 0, 0, 0, 0, 0};
                if (L.in2) c = *reinterpret_cast<const s8v*>(L.in2 + o);
                const float w0 = L.nodew[0], w1 = L.nodew[1], w2 = L.nodew[2];
#pragma unroll
                for (int e = 0; e < 8; ++e)                // (same expression as bifpn_node_kernel: bit-identical z)
                    v[e] = (short)f32_to_bf16_rn(node_mix(w0, bf16_to_f32((unsigned short)v[e]), w1, bf16_to_f32((unsigned short)b[e]),
                                                          w2, bf16_to_f32((unsigned short)c[e])));
                if (iy >= pad_t && iy < pad_t + DT_TH && ix >= pad_l && ix < pad_l + DT_TW)
                    *reinterpret_cast<s8v*>(L.zout + o) = v;   // this tile's own output pixels
            }
            if (SEAPPLY) {
                const s8v zc = *reinterpret_cast<const s8v*>(L.ap_z + o);
                const float* ga = L.ap_gate + (long long)n * C + (c8b + ch) * 8;
                const float* gp = L.ap_gpool + (long long)n * C + (c8b + ch) * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e)                // (same expression as se_bwd_apply_kernel: bit-identical gz)
                    v[e] = (short)f32_to_bf16_rn(se_apply_mix(bf16_to_f32((unsigned short)v[e]), ga[e], gp[e], L.ap_inv,
                                                              dswishf(bf16_to_f32((unsigned short)zc[e]))));
                if (iy >= pad_t && iy < pad_t + DT_TH && ix >= pad_l && ix < pad_l + DT_TW)
                    *reinterpret_cast<s8v*>(L.ap_gz + o) = v;
            }
        }
        if (INSW) {                                        // input stored as pre-activation z: swish once per loaded
#pragma unroll                                             // element, rounded to bf16 like a materialised swish(z)
            for (int e = 0; e < 8; ++e) v[e] = (short)f32_to_bf16_rn(swishf(bf16_to_f32((unsigned short)v[e])));
        }
        tile[ch * NPX + px] = v;
    }
    for (int i = tid; i < K * K * nc8 * 8; i += 256) {
        const int t = i / (nc8 * 8), c = i - t * (nc8 * 8);
        wl[i] = w_t[(FLIP ? K * K - 1 - t : t) * C + c8b * 8 + c];
    }
    __syncthreads();
    const int ty = tid >> 5, tx = tid & 31;
    s8v res[4];                                            // nc8 <= 4 chunks, results held until the tile is dead
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        if (ch >= nc8) break;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bias ? bias[(c8b + ch) * 8 + e] : 0.0f;
#pragma unroll
        for (int a = 0; a < K; ++a)
#pragma unroll
            for (int b = 0; b < K; ++b) {
                const s8v v = S == 1 ? tile[ch * NPX + (ty + a) * IW + tx + b]
                                     : tile[ch * NPX + ((a & 1) * 2 + (b & 1)) * (PH * PW) + (ty + (a >> 1)) * PW + tx + (b >> 1)];
                const float4 w0 = *reinterpret_cast<const float4*>(wl + (a * K + b) * nc8 * 8 + ch * 8);
                const float4 w1 = *reinterpret_cast<const float4*>(wl + (a * K + b) * nc8 * 8 + ch * 8 + 4);
                acc[0] += bf16_to_f32((unsigned short)v[0]) * w0.x;  acc[1] += bf16_to_f32((unsigned short)v[1]) * w0.y;
                acc[2] += bf16_to_f32((unsigned short)v[2]) * w0.z;  acc[3] += bf16_to_f32((unsigned short)v[3]) * w0.w;
                acc[4] += bf16_to_f32((unsigned short)v[4]) * w1.x;  acc[5] += bf16_to_f32((unsigned short)v[5]) * w1.y;
                acc[6] += bf16_to_f32((unsigned short)v[6]) * w1.z;  acc[7] += bf16_to_f32((unsigned short)v[7]) * w1.w;
            }
        // staging planes [chunk][pixel] (DT_OS = 257 cells apart): consecutive lanes write consecutive cells (a
        // [pixel][chunk] layout puts lanes 64-128 B apart: 16- to 32-way bank conflicts on every store)
        // (the data gradient stages bf16 too: fp32 cells doubled the LDS footprint to 56 KiB = 2 blocks per CU for
        //  a memory-bound kernel; the swish' / accumulate epilogue then rounds once more, 2^-9 relative)
        //  The staging planes reuse the input tile's LDS (24-29 KiB per block in all: 5-6 blocks per CU).
#pragma unroll
        for (int e = 0; e < 8; ++e) res[ch][e] = (short)f32_to_bf16_rn(acc[e]);
    }
    __syncthreads();                                       // every lane has finished reading the input tile
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
        if (ch < nc8) ot[ch * DT_OS + tid] = res[ch];
    __syncthreads();
    float psum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // SE pooling: this thread's pixels of chunk tid % nc8
    // ---- outputs: piece i = (pixel, chunk), a pixel's chunks contiguous in memory
    for (int i = tid; i < 256 * nc8; i += 256) {
        const int px = i / nc8, ch = i - px * nc8;
        const int p = p0 + (px >> 5), q = q0 + (px & 31);
        if (p >= OP || q >= OQ) continue;
        const long long off = (((long long)n * OP + p) * OQ + q) * C + (c8b + ch) * 8;
        if (!FLIP && L.pool) {
            const s8v sv = ot[ch * DT_OS + px];
#pragma unroll
            for (int e = 0; e < 8; ++e) psum[e] += swishf(bf16_to_f32((unsigned short)sv[e]));
        }
        if (FLIP) {
            const s8v sv = ot[ch * DT_OS + px];
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bf16_to_f32((unsigned short)sv[e]);
            if (zin) {
                float z[8];
                load8(zin + off, z);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dswishf(z[e]);
            }
            if (accumulate) {
                float o[8];
                load8(out + off, o);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += o[e];
            }
            store8(out + off, v);
        } else {
            *reinterpret_cast<s8v*>(out + off) = ot[ch * DT_OS + px];
        }
    }
    if (!FLIP && L.pool) {                                 // (uniform)
        // threads with the same tid % nc8 hold the same chunk (256 % nc8 == 0): xor steps over the higher lane bits,
        // then the four waves through LDS (the staging planes are dead by now)
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);           // [4 waves][nc8][8]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = psum[e];
            for (int o = 32; o >= nc8; o >>= 1) v += __shfl_xor(v, o, 64);
            psum[e] = v;
        }
        const int lane = tid & 63, wv = tid >> 6;
        if (lane < nc8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(wv * nc8 + lane) * 8 + e] = psum[e];
        }
        __syncthreads();
        if (tid < nc8 * 8) {
            const int ch = tid >> 3, e = tid & 7;
            const float t = (red[(0 * nc8 + ch) * 8 + e] + red[(1 * nc8 + ch) * 8 + e]) +
                            (red[(2 * nc8 + ch) * 8 + e] + red[(3 * nc8 + ch) * 8 + e]);
            L.pool[((long long)n * gridDim.x + tile_id) * C + (c8b + ch) * 8 + e] = t * L.pool_scale;
        }
    }
}

// levels[i]: x = the map the conv reads (forward: input; data gradient: dy), dy = the map it writes (forward:
// output; data gradient: dx), aux = zin (data gradient only)
struct DwNodeArgs { const void* in1; const void* in2; void* zout; const float* w; int up1; };
struct DwPoolArgs { float* pool; float scale; };
struct DwApplyArgs { const void* z; const float* gate; const float* gpool; void* gz; float inv; };

template <int K, bool FLIP, bool INSW = false, bool NODE = false, int S = 1, bool SEAPPLY = false>
static int launch_dw_tile(const Mi355DwLevel* levels, int nlevels, const float* w_t, const float* bias, int accumulate,
                          int N, int C, int pad_t, int pad_l, void* stream, const DwNodeArgs* node = nullptr,
                          const DwPoolArgs* pool = nullptr, const DwApplyArgs* apply = nullptr) {
    const int c8n = C / 8;
    DwTileLaunch L;
    memset(&L, 0, sizeof(L));
    if (pool) { L.pool = pool->pool; L.pool_scale = pool->scale; }
    if (SEAPPLY) {
        L.ap_z = (const unsigned short*)apply->z;  L.ap_gate = apply->gate;  L.ap_gpool = apply->gpool;
        L.ap_gz = (unsigned short*)apply->gz;  L.ap_inv = apply->inv;
    }
    if (NODE) {
        L.in1 = (const unsigned short*)node->in1;  L.in2 = (const unsigned short*)node->in2;
        L.zout = (unsigned short*)node->zout;  L.nodew = node->w;  L.up1 = node->up1;
    }
    L.nlevels = nlevels; L.C = C; L.pad_t = pad_t; L.pad_l = pad_l; L.accumulate = accumulate;
    static const int dw_xcd = [] { const char* e = getenv("MI355_DW_XCD"); return e ? atoi(e) : 1; }();
    L.xs = dw_xcd;
    L.nc8 = (S == 1 && c8n % 4 == 0) ? 4 : (c8n % 2 == 0 ? 2 : 1);        // stride 2: the halo tile is 4x the pixels
    long long tiles = 0;
    for (int i = 0; i < nlevels; ++i) {
        DwTileLevel& d = L.lv[i];
        d.in = (const unsigned short*)levels[i].x;  d.out = (unsigned short*)levels[i].dy;
        d.zin = (const unsigned short*)levels[i].aux;
        d.H = levels[i].H;  d.W = levels[i].W;
        d.P = S == 1 ? d.H : levels[i].P;  d.Q = S == 1 ? d.W : levels[i].Q;
        d.tilesX = (d.Q + DT_TW - 1) / DT_TW;
        d.tile_begin = (int)tiles;
        tiles += (long long)d.tilesX * ((d.P + DT_TH - 1) / DT_TH);
    }
    MI355_REQUIRE(tiles > 0 && tiles < 0x7fffffffLL && (long long)N * (c8n / L.nc8) <= 65535, "dw_tile: grid");
    constexpr int IH = (DT_TH - 1) * S + K, IW = (DT_TW - 1) * S + K;
    constexpr int NPX = S * S * ((IH + S - 1) / S) * ((IW + S - 1) / S);
    static_assert(NPX >= DT_OS, "output staging must fit in the input tile");
    const size_t lds = (size_t)L.nc8 * NPX * 16 + (size_t)K * K * L.nc8 * 32;
    const dim3 grid((unsigned)tiles, (unsigned)(N * (c8n / L.nc8)));
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)dw_tile_s1_kernel<K, FLIP, INSW, NODE, S, SEAPPLY>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((dw_tile_s1_kernel<K, FLIP, INSW, NODE, S, SEAPPLY>), grid, dim3(256), lds, (hipStream_t)stream, L, w_t, bias);
    return check_launch("dw_tile_s1_kernel");
}

// z = w[0] in0 + w[1] in1' + w[2] in2 (stored), out = dwconv_kxk(swish(z)) + bias: mi355_bifpn_node_fwd followed by
// mi355_dwconv_fwd(in_swish = 1) in one launch, bit-identical to the pair
extern "C" int mi355_bifpn_node_dwconv_fwd(const void* in0, const void* in1, int up1, const void* in2, const float* w,
                                           void* z, const float* w_t, const float* bias, void* out, int N, int H, int W,
                                           int C, int k, void* stream) {
    MI355_REQUIRE(in0 && in1 && w && z && w_t && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (k == 3 || k == 5),
                  "bifpn_node_dwconv_fwd: bad args (k = 3 or 5)");
    MI355_REQUIRE(!up1 || (H % 2 == 0 && W % 2 == 0), "bifpn_node_dwconv_fwd: upsample needs even H, W");
    Mi355DwLevel lv;
    memset(&lv, 0, sizeof(lv));
    lv.x = in0; lv.dy = out; lv.H = H; lv.W = W;
    const DwNodeArgs node = {in1, in2, z, w, up1};
    const int p = (k - 1) / 2;
    return k == 3 ? launch_dw_tile<3, false, true, true>(&lv, 1, w_t, bias, 0, N, C, p, p, stream, &node)
                  : launch_dw_tile<5, false, true, true>(&lv, 1, w_t, bias, 0, N, C, p, p, stream, &node);
}

static bool dw_levels_same_s1(const Mi355DwLevel* lv, int n, int k) {
    if (!(lv && n >= 1 && n <= MI355_DW_MAX_LEVELS && (k == 3 || k == 5))) return false;
    for (int i = 0; i < n; ++i)
        if (!(lv[i].x && lv[i].dy && lv[i].H > 0 && lv[i].W > 0 && lv[i].P == lv[i].H && lv[i].Q == lv[i].W &&
              lv[i].pad_t == (k - 1) / 2 && lv[i].pad_l == (k - 1) / 2))
            return false;
    return true;
}

extern "C" int mi355_dwconv_fwd_multi(const Mi355DwLevel* levels, int nlevels, int in_swish, const float* w_t,
                                      const float* bias, int N, int C, int k, void* stream) {
    MI355_REQUIRE(w_t && N > 0 && C > 0 && C % 8 == 0 && dw_levels_same_s1(levels, nlevels, k),
                  "dwconv_fwd_multi: stride-1 'same' 3x3 / 5x5 maps only (x = input, dy = output)");
    const int p = (k - 1) / 2;
    if (in_swish)
        return k == 3 ? launch_dw_tile<3, false, true>(levels, nlevels, w_t, bias, 0, N, C, p, p, stream)
                      : launch_dw_tile<5, false, true>(levels, nlevels, w_t, bias, 0, N, C, p, p, stream);
    return k == 3 ? launch_dw_tile<3, false>(levels, nlevels, w_t, bias, 0, N, C, p, p, stream)
                  : launch_dw_tile<5, false>(levels, nlevels, w_t, bias, 0, N, C, p, p, stream);
}

extern "C" int mi355_dwconv_dgrad_multi(const Mi355DwLevel* levels, int nlevels, const float* w_t, int N, int C, int k,
                                        int accumulate, void* stream) {
    MI355_REQUIRE(w_t && N > 0 && C > 0 && C % 8 == 0 && dw_levels_same_s1(levels, nlevels, k),
                  "dwconv_dgrad_multi: stride-1 'same' 3x3 / 5x5 maps only (x = dy, dy = dx, aux = zin)");
    const int p = k - 1 - (k - 1) / 2;
    return k == 3 ? launch_dw_tile<3, true>(levels, nlevels, w_t, nullptr, accumulate, N, C, p, p, stream)
                  : launch_dw_tile<5, true>(levels, nlevels, w_t, nullptr, accumulate, N, C, p, p, stream);
}

extern "C" int mi355_dwconv_fwd(const void* in, int in_swish, const float* w_t, const float* bias, void* out,
                                int N, int H, int W, int C, int k, int stride, int pad_t, int pad_l, int P, int Q,
                                void* stream) {
    MI355_REQUIRE(in && w_t && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && k > 0 && stride > 0,
                  "dwconv_fwd: bad args");
    if (stride == 1 && (k == 3 || k == 5) && P == H && Q == W && W >= 16 && H >= 8) {
        Mi355DwLevel lv;
        memset(&lv, 0, sizeof(lv));
        lv.x = in; lv.dy = out; lv.H = H; lv.W = W;
        if (in_swish)
            return k == 3 ? launch_dw_tile<3, false, true>(&lv, 1, w_t, bias, 0, N, C, pad_t, pad_l, stream)
                          : launch_dw_tile<5, false, true>(&lv, 1, w_t, bias, 0, N, C, pad_t, pad_l, stream);
        return k == 3 ? launch_dw_tile<3, false>(&lv, 1, w_t, bias, 0, N, C, pad_t, pad_l, stream)
                      : launch_dw_tile<5, false>(&lv, 1, w_t, bias, 0, N, C, pad_t, pad_l, stream);
    }
    if (stride == 2 && (k == 3 || k == 5) && Q >= 16 && P >= 8) {
        // LDS-tiled with parity planes (TF-"same": the pad of the even-size maps is asymmetric, pad_t / pad_l as given)
        Mi355DwLevel lv;
        memset(&lv, 0, sizeof(lv));
        lv.x = in; lv.dy = out; lv.H = H; lv.W = W; lv.P = P; lv.Q = Q;
        if (in_swish)
            return k == 3 ? launch_dw_tile<3, false, true, false, 2>(&lv, 1, w_t, bias, 0, N, C, pad_t, pad_l, stream)
                          : launch_dw_tile<5, false, true, false, 2>(&lv, 1, w_t, bias, 0, N, C, pad_t, pad_l, stream);
        return k == 3 ? launch_dw_tile<3, false, false, false, 2>(&lv, 1, w_t, bias, 0, N, C, pad_t, pad_l, stream)
                      : launch_dw_tile<5, false, false, false, 2>(&lv, 1, w_t, bias, 0, N, C, pad_t, pad_l, stream);
    }
    if (stride == 1 && !in_swish && (k == 3 || k == 5) && P == H && Q == W) {
        const long long tot4 = (long long)N * P * ((Q + 3) / 4) * (C / 8);
        if (k == 3)
            hipLaunchKernelGGL(dwconv_fwd_s1_kernel<3>, dim3(blocks_for(tot4)), dim3(256), 0, (hipStream_t)stream,
                               (const unsigned short*)in, w_t, bias, (unsigned short*)out, N, H, W, C, pad_t, pad_l, P, Q);
        else
            hipLaunchKernelGGL(dwconv_fwd_s1_kernel<5>, dim3(blocks_for(tot4)), dim3(256), 0, (hipStream_t)stream,
                               (const unsigned short*)in, w_t, bias, (unsigned short*)out, N, H, W, C, pad_t, pad_l, P, Q);
        return check_launch("dwconv_fwd_s1_kernel");
    }
    const long long total = (long long)N * P * Q * (C / 8);
    if (in_swish)
        hipLaunchKernelGGL((dwconv_fwd_kernel<unsigned short, true>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)in, w_t, bias, (unsigned short*)out, N, H, W, C, k, stride,
                           pad_t, pad_l, P, Q);
    else
        hipLaunchKernelGGL((dwconv_fwd_kernel<unsigned short, false>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)in, w_t, bias, (unsigned short*)out, N, H, W, C, k, stride,
                           pad_t, pad_l, P, Q);
    return check_launch("dwconv_fwd_kernel");
}

// mi355_dwconv_fwd + mi355_se_pool in one launch (the LDS-tiled forms only: k = 3 / 5, stride 1 "same" on maps >= 8 x 16,
// stride 2 onto maps >= 8 x 16): `pool_partial` [N][slabs][C] receives per-tile partial means of swish(out), slabs =
// mi355_dwconv_fwd_sepool_slabs(...) (0 = this geometry takes another kernel: call the two entry points).
extern "C" int mi355_dwconv_fwd_sepool_slabs(int H, int W, int k, int stride, int P, int Q, int* slabs) {
    MI355_REQUIRE(slabs && H > 0 && W > 0 && P > 0 && Q > 0, "dwconv_fwd_sepool_slabs: bad args");
    const bool ok = (k == 3 || k == 5) && ((stride == 1 && P == H && Q == W && W >= 16 && H >= 8) ||
                                           (stride == 2 && Q >= 16 && P >= 8));
    *slabs = ok ? ((Q + DT_TW - 1) / DT_TW) * ((P + DT_TH - 1) / DT_TH) : 0;
    return MI355_OK;
}

extern "C" int mi355_dwconv_fwd_sepool(const void* in, int in_swish, const float* w_t, const float* bias, void* out,
                                       float* pool_partial, int N, int H, int W, int C, int k, int stride, int pad_t,
                                       int pad_l, int P, int Q, void* stream) {
    int slabs = 0;
    MI355_REQUIRE(in && w_t && out && pool_partial && N > 0 && C > 0 && C % 8 == 0, "dwconv_fwd_sepool: bad args");
    mi355_dwconv_fwd_sepool_slabs(H, W, k, stride, P, Q, &slabs);
    MI355_REQUIRE(slabs > 0, "dwconv_fwd_sepool: geometry not served by the tiled kernel (see mi355_dwconv_fwd_sepool_slabs)");
    Mi355DwLevel lv;
    memset(&lv, 0, sizeof(lv));
    lv.x = in; lv.dy = out; lv.H = H; lv.W = W; lv.P = P; lv.Q = Q;
    const DwPoolArgs pool = {pool_partial, 1.0f / (float)((long long)P * Q)};
#define MI355_DWP(KK, SW, SS) launch_dw_tile<KK, false, SW, false, SS>(&lv, 1, w_t, bias, 0, N, C, pad_t, pad_l, stream, nullptr, &pool)
    if (stride == 1) {
        if (in_swish) return k == 3 ? MI355_DWP(3, true, 1) : MI355_DWP(5, true, 1);
        return k == 3 ? MI355_DWP(3, false, 1) : MI355_DWP(5, false, 1);
    }
    if (in_swish) return k == 3 ? MI355_DWP(3, true, 2) : MI355_DWP(5, true, 2);
    return k == 3 ? MI355_DWP(3, false, 2) : MI355_DWP(5, false, 2);
#undef MI355_DWP
}

template <typename T>
__global__ void dwconv_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w_t,
                                    T* __restrict__ dx, const T* __restrict__ zin, int N,
                                    int H, int W, int C, int k, int stride, int pad_t, int pad_l, int P, int Q,
                                    int accumulate) {
    const int c8n = C >> 3;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * H * W * c8n) return;
    int c8, w, h, n;
    split_cell(t, c8n, W, H, c8, w, h, n);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < k; ++a) {
        const int hp = h + pad_t - a;
        if (hp < 0 || hp % stride) continue;
        const int p = hp / stride;
        if (p >= P) continue;
        for (int b = 0; b < k; ++b) {
            const int wq = w + pad_l - b;
            if (wq < 0 || wq % stride) continue;
            const int q = wq / stride;
            if (q >= Q) continue;
            float g[8];
            load8(dy + (((long long)n * P + p) * Q + q) * C + c8 * 8, g);
            const float* wt = w_t + (a * k + b) * C + c8 * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += g[e] * wt[e];
        }
    }
    if (zin) {
        float z[8];
        load8(zin + t * 8, z);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] *= Act<T>::dsw(z[e]);
    }
    if (accumulate) {
        float o[8];
        load8(dx + t * 8, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += o[e];
    }
    store8(dx + t * 8, acc);
}

// Stride-2 data gradient: a thread owns the 2 x 2 input block (2h', 2w') of one 8-channel chunk.  The block lies
// under at most CN x CN output cells (CN = (K + 1) / 2 + 1): each (dy cell, tap) pair is visited once, with the tap
// index a = h + pad_t - 2p known from the cell (the per-pixel kernel walks all K x K taps with a modulo test each).
template <int K>
__global__ __launch_bounds__(256) void dwconv_dgrad_s2_kernel(
    const unsigned short* __restrict__ dy, const float* __restrict__ w_t, unsigned short* __restrict__ dx,
    const unsigned short* __restrict__ zin, int N, int H, int W, int C, int pad_t, int pad_l, int P, int Q,
    int accumulate) {
    constexpr int CN = (K + 1) / 2 + 1;
    const int c8n = C >> 3, H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * H2 * W2 * c8n) return;
    int c8, w2, h2, n;
    split_cell(t, c8n, W2, H2, c8, w2, h2, n);
    float acc[2][2][8];
#pragma unroll
    for (int i = 0; i < 32; ++i) (&acc[0][0][0])[i] = 0.0f;
    // first output cell that can reach input row 2h2 (+1): 2p - pad_t + a = h, a <= K-1  ->  p >= (h + pad_t - K + 1) / 2
    const int hb = 2 * h2 + pad_t, wb = 2 * w2 + pad_l;
    const int p_lo = (hb - K + 2) >> 1, q_lo = (wb - K + 2) >> 1;            // ceil((hb - K + 1) / 2), may be < 0
    const float* wc = w_t + c8 * 8;
#pragma unroll
    for (int i = 0; i < CN; ++i) {
        const int p = p_lo + i;
        if (p < 0 || p >= P) continue;
#pragma unroll
        for (int j = 0; j < CN; ++j) {
            const int q = q_lo + j;
            if (q < 0 || q >= Q) continue;
            float g[8];
            load8(dy + (((long long)n * P + p) * Q + q) * C + c8 * 8, g);
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) {
                const int a = hb + dh - 2 * p;
                if (a < 0 || a >= K) continue;
#pragma unroll
                for (int dw = 0; dw < 2; ++dw) {
                    const int b = wb + dw - 2 * q;
                    if (b < 0 || b >= K) continue;
                    const float4 w0 = *reinterpret_cast<const float4*>(wc + (a * K + b) * C);
                    const float4 w1 = *reinterpret_cast<const float4*>(wc + (a * K + b) * C + 4);
                    float* o = acc[dh][dw];
                    o[0] += g[0] * w0.x; o[1] += g[1] * w0.y; o[2] += g[2] * w0.z; o[3] += g[3] * w0.w;
                    o[4] += g[4] * w1.x; o[5] += g[5] * w1.y; o[6] += g[6] * w1.z; o[7] += g[7] * w1.w;
                }
            }
        }
    }
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
            const int h = 2 * h2 + dh, w = 2 * w2 + dw;
            if (h >= H || w >= W) continue;
            const long long o = ((((long long)n * H + h) * W + w) * c8n + c8) * 8;
            float* v = acc[dh][dw];
            if (zin) {
                float z[8];
                load8(zin + o, z);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dswishf(z[e]);
            }
            if (accumulate) {
                float old[8];
                load8(dx + o, old);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += old[e];
            }
            store8(dx + o, v);
        }
}

extern "C" int mi355_dwconv_dgrad(const void* dy, const float* w_t, void* dx, const void* zin, int N, int H,
                                  int W, int C, int k, int stride, int pad_t, int pad_l, int P, int Q,
                                  int accumulate, void* stream) {
    MI355_REQUIRE(dy && w_t && dx && N > 0 && C > 0 && C % 8 == 0 && k > 0 && stride > 0, "dwconv_dgrad: bad args");
    if (stride == 1 && (k == 3 || k == 5) && P == H && Q == W && W >= 16 && H >= 8 && pad_t == (k - 1) / 2 && pad_l == (k - 1) / 2)
    {
        Mi355DwLevel lv;
        memset(&lv, 0, sizeof(lv));
        lv.x = dy; lv.dy = dx; lv.aux = zin; lv.H = H; lv.W = W;
        return k == 3 ? launch_dw_tile<3, true>(&lv, 1, w_t, nullptr, accumulate, N, C, k - 1 - pad_t, k - 1 - pad_l, stream)
                      : launch_dw_tile<5, true>(&lv, 1, w_t, nullptr, accumulate, N, C, k - 1 - pad_t, k - 1 - pad_l, stream);
    }
    if (stride == 2 && (k == 3 || k == 5)) {
        const long long tot4 = (long long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
        if (k == 3)
            hipLaunchKernelGGL(dwconv_dgrad_s2_kernel<3>, dim3(blocks_for(tot4)), dim3(256), 0, (hipStream_t)stream,
                               (const unsigned short*)dy, w_t, (unsigned short*)dx, (const unsigned short*)zin, N, H, W,
                               C, pad_t, pad_l, P, Q, accumulate);
        else
            hipLaunchKernelGGL(dwconv_dgrad_s2_kernel<5>, dim3(blocks_for(tot4)), dim3(256), 0, (hipStream_t)stream,
                               (const unsigned short*)dy, w_t, (unsigned short*)dx, (const unsigned short*)zin, N, H, W,
                               C, pad_t, pad_l, P, Q, accumulate);
        return check_launch("dwconv_dgrad_s2_kernel");
    }
    const long long total = (long long)N * H * W * (C / 8);
    hipLaunchKernelGGL(dwconv_dgrad_kernel<unsigned short>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)dy, w_t, (unsigned short*)dx, (const unsigned short*)zin, N, H, W, C,
                       k, stride, pad_t, pad_l, P, Q, accumulate);
    return check_launch("dwconv_dgrad_kernel");
}

// mi355_se_bwd_apply + mi355_dwconv_dgrad (stride 1, "same", k = 3 / 5, maps >= 8 x 16) in one launch: the depthwise
// conv's output gradient gz = (gu * gate + gpool / HW) * swish'(z) is built while the halo tile is staged and stored
// (the weight gradient reads it); dx = conv-transpose(gz) [* swish'(zin)] [+ dx]; bit-identical to the pair.
extern "C" int mi355_dwconv_dgrad_seapply(const void* gu, const void* z, const float* gate, const float* gpool, void* gz,
                                          const float* w_t, void* dx, const void* zin, int N, int H, int W, int C, int k,
                                          int accumulate, void* stream) {
    MI355_REQUIRE(gu && z && gate && gpool && gz && w_t && dx && N > 0 && C > 0 && C % 8 == 0 && (k == 3 || k == 5) &&
                  W >= 16 && H >= 8, "dwconv_dgrad_seapply: stride-1 'same' 3x3 / 5x5 on maps >= 8 x 16 only");
    Mi355DwLevel lv;
    memset(&lv, 0, sizeof(lv));
    lv.x = gu; lv.dy = dx; lv.aux = zin; lv.H = H; lv.W = W;
    const DwApplyArgs ap = {z, gate, gpool, gz, 1.0f / (float)((long long)H * W)};
    const int p = k - 1 - (k - 1) / 2;
    return k == 3 ? launch_dw_tile<3, true, false, false, 1, true>(&lv, 1, w_t, nullptr, accumulate, N, C, p, p, stream, nullptr, nullptr, &ap)
                  : launch_dw_tile<5, true, false, false, 1, true>(&lv, 1, w_t, nullptr, accumulate, N, C, p, p, stream, nullptr, nullptr, &ap);
}

// block = 32 (tap, 8-channel chunk) items x 8 pixel lanes over a chunk of output rows; the pixel
// lanes are reduced through LDS so that a block issues ONE atomic per (tap, channel): contention on
// the tiny dw_t tensor is the number of row chunks (~100), not the number of threads.
constexpr int DW_IPB = 32, DW_PL = 8;
template <typename T, bool SWISH>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw_t,
    float* __restrict__ colsum, int N, int H, int W, int C, int k, int stride, int pad_t, int pad_l, int P, int Q,
    int rows_per_block) {
    __shared__ float s_acc[DW_PL][DW_IPB][9];
    const int c8n = C >> 3, NT = k * k;
    const int il = threadIdx.x % DW_IPB, pl = threadIdx.x / DW_IPB;
    const int item = blockIdx.x * DW_IPB + il;                 // (tap, c8)
    const bool live = item < NT * c8n;
    const int tap = live ? item / c8n : 0, c8 = live ? item - tap * c8n : 0;
    const int a = tap / k, b = tap - a * k;
    const int nrows = N * P;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(nrows, r0 + rows_per_block);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool do_cs = colsum != nullptr && tap == 0 && live;
    const int qlo = max(0, (pad_l - b + stride - 1) / stride);
    const int qhi = min(Q, (W - 1 + pad_l - b) / stride + 1);
    if (live) {
        for (int r = r0; r < r1; ++r) {
            const int n = r / P, p = r - n * P;
            const T* dyr = dy + (long long)r * Q * C + c8 * 8;
            if (do_cs) {
                for (int q = pl; q < Q; q += DW_PL) {
                    float g[8];
                    load8(dyr + (long long)q * C, g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) cs[e] += g[e];
                }
            }
            const int h = p * stride - pad_t + a;
            if ((unsigned)h >= (unsigned)H) continue;
            const T* xr = x + ((long long)n * H + h) * W * C + c8 * 8 + (long long)(b - pad_l) * C;
            int q = qlo + pl;
            for (; q + DW_PL < qhi; q += 2 * DW_PL) {
                float g0[8], v0[8], g1[8], v1[8];
                load8(dyr + (long long)q * C, g0);
                load8(xr + (long long)q * stride * C, v0);
                load8(dyr + (long long)(q + DW_PL) * C, g1);
                load8(xr + (long long)(q + DW_PL) * stride * C, v1);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    acc[e] += (SWISH ? Act<T>::sw(v0[e]) : v0[e]) * g0[e] + (SWISH ? Act<T>::sw(v1[e]) : v1[e]) * g1[e];
            }
            for (; q < qhi; q += DW_PL) {
                float g[8], v[8];
                load8(dyr + (long long)q * C, g);
                load8(xr + (long long)q * stride * C, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (SWISH ? Act<T>::sw(v[e]) : v[e]) * g[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s_acc[pl][il][e] = acc[e];
    __syncthreads();
    if (pl == 0 && live) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = 0.0f;
#pragma unroll
            for (int l = 0; l < DW_PL; ++l) t += s_acc[l][il][e];
            atomicAdd(dw_t + tap * C + c8 * 8 + e, t);
        }
    }
    if (colsum) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) s_acc[pl][il][e] = cs[e];
        __syncthreads();
        if (pl == 0 && do_cs) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = 0.0f;
#pragma unroll
                for (int l = 0; l < DW_PL; ++l) t += s_acc[l][il][e];
                atomicAdd(colsum + c8 * 8 + e, t);
            }
        }
    }
}


// Stride-1 form: a thread owns one KERNEL ROW a of one 8-channel chunk and walks whole output rows with a
// register window of the k input pixels under the current output pixel: per pixel ONE dy load and ONE new x
// load feed k taps (the per-tap form above re-reads both for every tap: k times the L1/L2 traffic).
// block = 32 (kernel row, chunk) items x 8 row lanes; lanes reduced through LDS, one atomic per (tap, channel).
// 8 (kernel row, chunk) items x 32 row lanes: layers have k * C/8 = 6 .. 432 items, blocks of 8 leave few idle
constexpr int DW_U = 8, DW1_IPB = 8, DW1_PL = 32;
template <int K, bool INSW = false, int S = 1>
__global__ __launch_bounds__(256) void dwconv_wgrad_s1_kernel(
    const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy, float* __restrict__ dw_t,
    float* __restrict__ colsum, int N, int H, int W, int C, int pad_t, int pad_l, int P, int Q, int rows_per_block,
    int qseg) {
    // S = stride (1 or 2): the window under output pixel q starts at input column q * S - pad_l and moves S pixels
    // per output pixel: S new loads feed K taps
    constexpr int U = S == 1 ? DW_U : DW_U / 2;
    __shared__ float s_acc[DW1_PL][DW1_IPB][9];
    const int c8n = C >> 3;
    const int il = threadIdx.x % DW1_IPB, pl = threadIdx.x / DW1_IPB;
    const int item = blockIdx.x * DW1_IPB + il;                 // (kernel row a, c8)
    const int qa = blockIdx.z * qseg, qb = min(Q, qa + qseg);  // this block's column segment of every row
    const bool live = item < K * c8n;
    const int a = live ? item / c8n : 0, c8 = live ? item - a * c8n : 0;
    const int nrows = N * P;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(nrows, r0 + rows_per_block);
    float acc[K][8], cs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < K; ++b)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[b][e] = 0.0f;
    const bool do_cs = colsum != nullptr && a == 0 && live;
    if (live) {
        for (int r = r0 + pl; r < r1; r += DW1_PL) {
            const int n = r / P, p = r - n * P;
            const int h = p * S - pad_t + a;
            const bool row_ok = (unsigned)h < (unsigned)H;
            if (!row_ok && !do_cs) continue;
            const unsigned short* dyr = dy + (long long)r * Q * C + c8 * 8;
            const unsigned short* xr = x + ((long long)n * H + (row_ok ? h : 0)) * W * C + c8 * 8;
            float win[K][8];                              // win[b] = x(h, q * S + b - pad_l)
#pragma unroll
            for (int b = 0; b < K - S; ++b) {
                const int w = qa * S + b - pad_l;         // window of output pixel q = qa, all but the last S taps
                if (row_ok && (unsigned)w < (unsigned)W) {
                    load8(xr + (long long)w * C, win[b + S]);
                    if (INSW) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) win[b + S][e] = bf16_to_f32(f32_to_bf16_rn(swishf(win[b + S][e])));
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) win[b + S][e] = 0.0f;
                }
            }
            // batches of U pixels: all loads of a batch are issued before the first use (a thread's loop is
            // otherwise one L2 round trip per pixel: 42 us for a 64 x 64 x 64 level against a 4 us floor)
            for (int q0 = qa; q0 < qb; q0 += U) {
                s8v xraw[U][S], graw[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = q0 + u;
                    const s8v zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int sx = 0; sx < S; ++sx) {
                        const int w = q * S + K - S + sx - pad_l;
                        xraw[u][sx] = (q < qb && row_ok && (unsigned)w < (unsigned)W)
                                          ? *reinterpret_cast<const s8v*>(xr + (long long)w * C) : zero;
                    }
                    graw[u] = q < qb ? *reinterpret_cast<const s8v*>(dyr + (long long)q * C) : zero;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int b = 0; b < K - S; ++b)
#pragma unroll
                        for (int e = 0; e < 8; ++e) win[b][e] = win[b + S][e];
                    float g[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
#pragma unroll
                        for (int sx = 0; sx < S; ++sx) {
                            float v = bf16_to_f32((unsigned short)xraw[u][sx][e]);
                            if (INSW) v = bf16_to_f32(f32_to_bf16_rn(swishf(v)));   // x is the pre-activation z
                            win[K - S + sx][e] = v;
                        }
                        g[e] = bf16_to_f32((unsigned short)graw[u][e]);
                    }
                    if (do_cs) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) cs[e] += g[e];
                    }
#pragma unroll
                    for (int b = 0; b < K; ++b)
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[b][e] += win[b][e] * g[e];
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < K; ++b) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) s_acc[pl][il][e] = acc[b][e];
        __syncthreads();
        if (pl == 0 && live) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = 0.0f;
#pragma unroll
                for (int l = 0; l < DW1_PL; ++l) t += s_acc[l][il][e];
                atomicAdd(dw_t + (a * K + b) * C + c8 * 8 + e, t);
            }
        }
    }
    if (colsum) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) s_acc[pl][il][e] = cs[e];
        __syncthreads();
        if (pl == 0 && do_cs) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = 0.0f;
#pragma unroll
                for (int l = 0; l < DW1_PL; ++l) t += s_acc[l][il][e];
                atomicAdd(colsum + c8 * 8 + e, t);
            }
        }
    }
}

extern "C" int mi355_dwconv_wgrad(const void* x, int in_swish, const void* dy, float* dw_t, float* colsum, int N,
                                  int H, int W, int C, int k, int stride, int pad_t, int pad_l, int P, int Q,
                                  void* stream) {
    MI355_REQUIRE(x && dy && dw_t && N > 0 && C > 0 && C % 8 == 0 && k > 0, "dwconv_wgrad: bad args");
    const int nrows = N * P;
    if ((stride == 1 && (k == 3 || k == 5) && P == H && Q == W) || (stride == 2 && (k == 3 || k == 5))) {
        const int items = k * (C / 8);
        const int bx = (items + DW1_IPB - 1) / DW1_IPB;
        const int qseg = 32, bz = (Q + qseg - 1) / qseg;      // column segments: a thread walks <= 32 pixels of a row
        int chunks = 2048 / (bx * bz);                        // ~2048 blocks, <= 512 atomics per (tap, channel)
        if (chunks * bz > 512) chunks = 512 / bz;
        if (chunks < 1) chunks = 1;
        int rows = (nrows + chunks - 1) / chunks;
        rows = (rows + DW1_PL - 1) / DW1_PL * DW1_PL;           // whole lane groups of rows
        const int by = (nrows + rows - 1) / rows;
        dim3 grid((unsigned)bx, (unsigned)by, (unsigned)bz);
#define MI355_DW1(KK, SW)                                                                                          \
    hipLaunchKernelGGL((dwconv_wgrad_s1_kernel<KK, SW>), grid, dim3(256), 0, (hipStream_t)stream,                    \
                       (const unsigned short*)x, (const unsigned short*)dy, dw_t, colsum, N, H, W, C, pad_t, pad_l, \
                       P, Q, rows, qseg)
        if (stride == 2) {
#define MI355_DW2(KK, SW)                                                                                          \
    hipLaunchKernelGGL((dwconv_wgrad_s1_kernel<KK, SW, 2>), grid, dim3(256), 0, (hipStream_t)stream,                 \
                       (const unsigned short*)x, (const unsigned short*)dy, dw_t, colsum, N, H, W, C, pad_t, pad_l, \
                       P, Q, rows, qseg)
            if (k == 3) { if (in_swish) MI355_DW2(3, true); else MI355_DW2(3, false); }
            else        { if (in_swish) MI355_DW2(5, true); else MI355_DW2(5, false); }
#undef MI355_DW2
        } else if (k == 3) { if (in_swish) MI355_DW1(3, true); else MI355_DW1(3, false); }
        else               { if (in_swish) MI355_DW1(5, true); else MI355_DW1(5, false); }
#undef MI355_DW1
        return check_launch("dwconv_wgrad_s1_kernel");
    }
    const int items = k * k * (C / 8);
    const int bx = (items + DW_IPB - 1) / DW_IPB;
    int chunks = 2048 / bx;                               // ~2048 blocks, <= 128-way atomic contention
    if (chunks > 128) chunks = 128;
    if (chunks < 1) chunks = 1;
    int rows = (nrows + chunks - 1) / chunks;
    if (rows < 1) rows = 1;
    const int by = (nrows + rows - 1) / rows;
    dim3 grid((unsigned)bx, (unsigned)by);
    if (in_swish)
        hipLaunchKernelGGL((dwconv_wgrad_kernel<unsigned short, true>), grid, dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)x, (const unsigned short*)dy, dw_t, colsum, N, H, W, C, k, stride,
                           pad_t, pad_l, P, Q, rows);
    else
        hipLaunchKernelGGL((dwconv_wgrad_kernel<unsigned short, false>), grid, dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)x, (const unsigned short*)dy, dw_t, colsum, N, H, W, C, k, stride,
                           pad_t, pad_l, P, Q, rows);
    return check_launch("dwconv_wgrad_kernel");
}

// LDS-tiled, bit-reproducible depthwise weight gradient (bf16 tensors, k = 3 / 5, stride 1 / 2), one launch for up to
// MI355_DW_MAX_LEVELS feature maps that share the weight (the pyramid levels of an EfficientDet head conv).  A block
// owns NC8 8-channel chunks and walks output tiles of 256 pixels (8 x 32, or 16 x 16 for maps <= 16 wide): the
// activation halo tile (swish applied once per loaded element when the layer stores its pre-activation) and the dy
// tile are fetched ONCE into [chunk][pixel] planes of 16-B cells (the sliding-window kernel above re-reads both k
// times through L1/L2 and a thread walks 32 pixels of a row in dependent batches: 30 us for a 64 x 64 x 64 pyramid
// level against a 4 us floor).  A thread owns (pixel lane of the tile, kernel row a, chunk) and keeps its k taps x 8
// channels in registers over all the tiles of the block: per step one dy cell and k activation cells (ds_read_b128,
// lanes = consecutive cells) feed 8 k multiply-adds issued as v_dot2c_f32_bf16 on the packed bf16 pairs with one half
// of the dy pair masked to zero (no bf16 -> fp32 unpacking: half the vector instructions).  At the end the 32 pixel
// lanes are summed by shuffles and the block stores ITS partial sums as one row of the layer's slab workspace
// [slab][(T + 1) * C] (taps, then the dy column sums): plain stores, no atomics, nothing to zero;
// mi355_dw_param_grads_table adds the rows in index order.
struct DwWgLevel {
    const unsigned short* x;
    const unsigned short* dy;
    int H, W, P, Q, pad_t, pad_l, tilesX, tilesXY, tile_begin;
};
struct DwWgLaunch {
    DwWgLevel lv[MI355_DW_MAX_LEVELS];
    int nlevels, ntiles, C, nc8, want_cs, xs;
};
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2bf(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2v, a), __builtin_bit_cast(bf2v, b), c, false);
}
typedef unsigned u4v __attribute__((ext_vector_type(4)));

template <int K, int S, bool INSW, int TW>
__global__ __launch_bounds__(32 * K * (K == 3 ? 4 : 2), (S == 2 ? 3 : (K == 3 ? 5 : 4))) void dw_wgrad_tile_kernel(const DwWgLaunch L, float* __restrict__ slabs) {
    constexpr int TH = 256 / TW, RP = 32 / TW;              // RP tile rows side by side in the 32 pixel lanes
    constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K, NLIN = IH * IW, NOUT = 256;
    // stride 2: the halo tile is kept as four parity planes (row parity, column parity) so that tap (a, b) of pixel
    // lane pcol reads cell pcol + (b >> 1) of plane (a & 1, b & 1): consecutive lanes = consecutive cells
    constexpr int PH = (IH + S - 1) / S, PW = (IW + S - 1) / S, NPX = S * S * PH * PW;
    constexpr int UJ = K == 3 ? 1 : 2;                      // unroll of the tile-row loop
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nc8 = L.nc8, C = L.C;
    u4v* xt = reinterpret_cast<u4v*>(smem);                 // [nc8][NPX]
    u4v* dt = xt + (size_t)nc8 * NPX;                       // [nc8][NOUT]
    const int tid = threadIdx.x;
    const int pl = tid & 31, slot = tid >> 5;               // slot = a * nc8 + ch
    const int a = slot / nc8, ch = slot - a * nc8;
    const int prow = pl / TW, pcol = pl - prow * TW;
    // Block order: the channel groups of a slab (= the same pixel tiles) read 16 .. 64-B pieces of the SAME 128-B lines,
    // and consecutive block ids go round-robin over the 8 XCDs (one L2 each): in plain order every XCD fetched every
    // line (PMC: 780 MB read by the 128^2 x 144 k5 / s2 launch for 94 MB of operands).  XCD x takes a contiguous range
    // of logical ids instead, i.e. all channel groups of a slab (a bijection: results unchanged; L.xs == 0: plain order).
    int bx = blockIdx.x, by = blockIdx.y;
    if (L.xs) {
        const int lg = xcd_swizzle((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
        by = lg / (int)gridDim.x;
        bx = lg - by * (int)gridDim.x;
    }
    const int c8b = bx * nc8;
    const int sc = tid % nc8, spx = tid / nc8;              // staging: this thread's chunk and first cell, stride 32 K
    float acc[K][8], cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        cs[e] = 0.0f;
#pragma unroll
        for (int b = 0; b < K; ++b) acc[b][e] = 0.0f;
    }
    const u4v zero = (u4v){0u, 0u, 0u, 0u};
    for (int t = by; t < L.ntiles; t += gridDim.y) {
        DwWgLevel P = L.lv[0];
#pragma unroll
        for (int i = 1; i < MI355_DW_MAX_LEVELS; ++i)
            if (i < L.nlevels && t >= L.lv[i].tile_begin) P = L.lv[i];
        const int tl = t - P.tile_begin;
        const int n = tl / P.tilesXY, r = tl - n * P.tilesXY;
        const int tyi = r / P.tilesX, txi = r - tyi * P.tilesX;
        const int p0 = tyi * TH, q0 = txi * TW;
        const int h0 = p0 * S - P.pad_t, w0 = q0 * S - P.pad_l;
        const unsigned short* xb = P.x + (long long)n * P.H * P.W * C + (c8b + sc) * 8;
        const unsigned short* gb = P.dy + (long long)n * P.P * P.Q * C + (c8b + sc) * 8;
        // every load of the tile is issued before the first use (a load -> swish -> LDS store loop is one HBM round
        // trip per cell: 22 us per tile)
        constexpr int XI = (NLIN + 32 * K - 1) / (32 * K), GI = (NOUT + 32 * K - 1) / (32 * K);  // sstep == 32 K
        u4v xr[XI], gr[GI];
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int px = spx + i * (32 * K);
            const int iy = px / IW, ix = px - iy * IW;
            const int h = h0 + iy, w = w0 + ix;
            xr[i] = zero;
            if (px < NLIN && (unsigned)h < (unsigned)P.H && (unsigned)w < (unsigned)P.W)
                xr[i] = *reinterpret_cast<const u4v*>(xb + ((long long)h * P.W + w) * C);
        }
#pragma unroll
        for (int i = 0; i < GI; ++i) {
            const int px = spx + i * (32 * K);
            const int p = p0 + px / TW, q = q0 + (px & (TW - 1));
            gr[i] = zero;
            if (px < NOUT && p < P.P && q < P.Q) gr[i] = *reinterpret_cast<const u4v*>(gb + ((long long)p * P.Q + q) * C);
        }
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int px = spx + i * (32 * K);
            u4v v = xr[i];
            if (INSW) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = swishf(__uint_as_float(v[e] << 16)), hi = swishf(__uint_as_float(v[e] & 0xffff0000u));
                    v[e] = (unsigned)f32_to_bf16_rn(lo) | ((unsigned)f32_to_bf16_rn(hi) << 16);
                }
            }
            if (px < NLIN) {
                const int iy = px / IW, ix = px - iy * IW;
                const int cell = S == 1 ? px : (((iy & 1) * 2 + (ix & 1)) * (PH * PW) + (iy >> 1) * PW + (ix >> 1));
                xt[sc * NPX + cell] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < GI; ++i) {
            const int px = spx + i * (32 * K);
            if (px < NOUT) dt[sc * NOUT + px] = gr[i];
        }
        __syncthreads();
        if (a < K) {
            // (fully unrolled, hipcc hoists all 8 (1 + K) ds_read_b128 and spills 130-350 VGPRs)
#pragma unroll UJ
            for (int j = 0; j < TH / RP; ++j) {
                const int p = j * RP + prow;
                const u4v gv = dt[ch * NOUT + p * TW + pcol];
                unsigned glo[4], ghi[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { glo[e] = gv[e] & 0x0000ffffu; ghi[e] = gv[e] & 0xffff0000u; }
#pragma unroll
                for (int e = 0; e < 4; ++e) {               // (every kernel row keeps the sums, row 0 stores them)
                    cs[2 * e] += __uint_as_float(gv[e] << 16);
                    cs[2 * e + 1] += __uint_as_float(ghi[e]);
                }
#pragma unroll
                for (int b = 0; b < K; ++b) {
                    const u4v xv = S == 1 ? xt[ch * NPX + (p + a) * IW + pcol + b]
                                          : xt[ch * NPX + ((a & 1) * 2 + (b & 1)) * (PH * PW) + (p + (a >> 1)) * PW + pcol + (b >> 1)];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[b][2 * e] = dot2bf(xv[e], glo[e], acc[b][2 * e]);
                        acc[b][2 * e + 1] = dot2bf(xv[e], ghi[e], acc[b][2 * e + 1]);
                    }
                }
            }
        }
        __syncthreads();
    }
    // the 32 pixel lanes of a (kernel row, chunk) slot are the 32 lanes of a half wave
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) cs[e] += __shfl_xor(cs[e], o, 64);
#pragma unroll
        for (int b = 0; b < K; ++b)
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) acc[b][e] += __shfl_xor(acc[b][e], o, 64);
    }
    if (pl == 0 && a < K) {
        float* row = slabs + (long long)by * (K * K + 1) * C + (c8b + ch) * 8;
#pragma unroll
        for (int b = 0; b < K; ++b) {
            float* dst = row + (a * K + b) * C;
            reinterpret_cast<float4*>(dst)[0] = make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
            reinterpret_cast<float4*>(dst)[1] = make_float4(acc[b][4], acc[b][5], acc[b][6], acc[b][7]);
        }
        if (a == 0 && L.want_cs) {
            float* dst = row + K * K * C;
            reinterpret_cast<float4*>(dst)[0] = make_float4(cs[0], cs[1], cs[2], cs[3]);
            reinterpret_cast<float4*>(dst)[1] = make_float4(cs[4], cs[5], cs[6], cs[7]);
        }
    }
}

// launch geometry shared by mi355_dwconv_wgrad_nslabs* and mi355_dwconv_wgrad_slabs*
static bool dw_wgrad_geometry(const Mi355DwLevel* lv, int nlevels, int N, int C, int k, int stride, DwWgLaunch* L,
                              int* tw, int* nslabs) {
    if (!(lv && nlevels >= 1 && nlevels <= MI355_DW_MAX_LEVELS && (k == 3 || k == 5) && (stride == 1 || stride == 2) &&
          C > 0 && C % 8 == 0 && N > 0))
        return false;
    const int c8n = C / 8, lim = (k == 3 ? 4 : 2) / stride;   // stride 2: the halo tile is four times the pixels
    int g = 1;
    for (int d = lim; d >= 1; --d)
        if (c8n % d == 0) { g = d; break; }
    int maxq = 0;
    for (int i = 0; i < nlevels; ++i) maxq = lv[i].Q > maxq ? lv[i].Q : maxq;
    *tw = maxq > 16 ? 32 : 16;
    const int th = 256 / *tw;
    long long nt = 0;
    for (int i = 0; i < nlevels; ++i) {
        if (!(lv[i].P > 0 && lv[i].Q > 0 && lv[i].H > 0 && lv[i].W > 0)) return false;
        if ((long long)N * lv[i].H * lv[i].W * C >= 0x7fffffffLL * 4LL) return false;
        const int tX = (lv[i].Q + *tw - 1) / *tw, tY = (lv[i].P + th - 1) / th;
        if (L) {
            DwWgLevel& d = L->lv[i];
            d.x = (const unsigned short*)lv[i].x; d.dy = (const unsigned short*)lv[i].dy;
            d.H = lv[i].H; d.W = lv[i].W; d.P = lv[i].P; d.Q = lv[i].Q; d.pad_t = lv[i].pad_t; d.pad_l = lv[i].pad_l;
            d.tilesX = tX; d.tilesXY = tX * tY; d.tile_begin = (int)nt;
        }
        nt += (long long)N * tX * tY;
    }
    if (nt > 0x7fffffffLL) return false;
    const int groups = c8n / g;
    int cap = 1536 / groups;                                // ~1536 blocks per launch, 16 .. 512 slab rows
    if (cap > 512) cap = 512;
    if (cap < 16) cap = 16;
    *nslabs = nt < cap ? (int)nt : cap;
    if (L) { L->nlevels = nlevels; L->ntiles = (int)nt; L->C = C; L->nc8 = g; }
    return true;
}

extern "C" int mi355_dwconv_wgrad_nslabs_multi(const Mi355DwLevel* levels, int nlevels, int N, int C, int k, int stride,
                                               int* nslabs) {
    int tw;
    MI355_REQUIRE(nslabs && dw_wgrad_geometry(levels, nlevels, N, C, k, stride, nullptr, &tw, nslabs),
                  "dwconv_wgrad_nslabs: k must be 3 or 5, stride 1 or 2, C %% 8 == 0, 1..%d levels", MI355_DW_MAX_LEVELS);
    return MI355_OK;
}

extern "C" int mi355_dwconv_wgrad_nslabs(int N, int C, int k, int stride, int P, int Q, int* nslabs) {
    Mi355DwLevel lv;
    memset(&lv, 0, sizeof(lv));
    lv.H = lv.W = 1; lv.P = P; lv.Q = Q;
    return mi355_dwconv_wgrad_nslabs_multi(&lv, 1, N, C, k, stride, nslabs);
}

template <int K, int S, bool INSW, int TW>
static void launch_dw_wgrad_tile(const DwWgLaunch& L, float* slabs, int nslabs, void* stream) {
    constexpr int TH = 256 / TW;
    constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
    constexpr int NPX = S * S * ((IH + S - 1) / S) * ((IW + S - 1) / S);
    const size_t lds = (size_t)L.nc8 * (NPX + 256) * 16;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)dw_wgrad_tile_kernel<K, S, INSW, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((dw_wgrad_tile_kernel<K, S, INSW, TW>), dim3((unsigned)(L.C / 8 / L.nc8), (unsigned)nslabs),
                       dim3(32 * K * L.nc8), lds, (hipStream_t)stream, L, slabs);
}

extern "C" int mi355_dwconv_wgrad_slabs_multi(const Mi355DwLevel* levels, int nlevels, int in_swish, float* slabs,
                                              int want_colsum, int N, int C, int k, int stride, void* stream) {
    DwWgLaunch L;
    memset(&L, 0, sizeof(L));
    int tw, nslabs;
    MI355_REQUIRE(slabs && dw_wgrad_geometry(levels, nlevels, N, C, k, stride, &L, &tw, &nslabs),
                  "dwconv_wgrad_slabs: k must be 3 or 5, stride 1 or 2, C %% 8 == 0, 1..%d levels", MI355_DW_MAX_LEVELS);
    for (int i = 0; i < nlevels; ++i) MI355_REQUIRE(levels[i].x && levels[i].dy, "dwconv_wgrad_slabs: level %d null x / dy", i);
    L.want_cs = want_colsum;
    static const int dw_xcd = [] { const char* e = getenv("MI355_DW_XCD"); return e ? atoi(e) : 1; }();
    L.xs = dw_xcd;
#define MI355_DWT2(KK, SS, SW)                                                                         \
    do {                                                                                               \
        if (tw == 32) launch_dw_wgrad_tile<KK, SS, SW, 32>(L, slabs, nslabs, stream);                  \
        else launch_dw_wgrad_tile<KK, SS, SW, 16>(L, slabs, nslabs, stream);                           \
    } while (0)
#define MI355_DWT(KK, SS)                                                                              \
    do {                                                                                               \
        if (in_swish) MI355_DWT2(KK, SS, true); else MI355_DWT2(KK, SS, false);                        \
    } while (0)
    if (k == 3 && stride == 1) MI355_DWT(3, 1);
    else if (k == 3) MI355_DWT(3, 2);
    else if (stride == 1) MI355_DWT(5, 1);
    else MI355_DWT(5, 2);
#undef MI355_DWT
#undef MI355_DWT2
    return check_launch("dw_wgrad_tile_kernel");
}

extern "C" int mi355_dwconv_wgrad_slabs(const void* x, int in_swish, const void* dy, float* slabs, int want_colsum,
                                        int N, int H, int W, int C, int k, int stride, int pad_t, int pad_l, int P,
                                        int Q, void* stream) {
    Mi355DwLevel lv;
    lv.x = x; lv.dy = dy; lv.aux = nullptr; lv.H = H; lv.W = W; lv.P = P; lv.Q = Q; lv.pad_t = pad_t; lv.pad_l = pad_l;
    return mi355_dwconv_wgrad_slabs_multi(&lv, 1, in_swish, slabs, want_colsum, N, C, k, stride, stream);
}

// a = swish(z), materialised once when several taps would otherwise recompute it (dw wgrad)
__global__ void swish_bf16_kernel(const unsigned short* __restrict__ z, unsigned short* __restrict__ a, long long n8) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n8) return;
    float v[8];
    load8(z + t * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = swishf(v[e]);
    store8(a + t * 8, v);
}

extern "C" int mi355_swish_bf16(const void* z, void* a, int64_t n, void* stream) {
    MI355_REQUIRE(z && a && n > 0 && n % 8 == 0, "swish_bf16: n must be a multiple of 8");
    hipLaunchKernelGGL(swish_bf16_kernel, dim3(blocks_for(n / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)z, (unsigned short*)a, (long long)(n / 8));
    return check_launch("swish_bf16_kernel");
}

__global__ void dw_param_grads_kernel(const float* __restrict__ G_t, const float* __restrict__ w,
                                      const float* __restrict__ scale, const float* __restrict__ colsum,
                                      const float* __restrict__ mean, const float* __restrict__ var, float eps, int C,
                                      int T, float* __restrict__ dW, float* __restrict__ dgamma,
                                      float* __restrict__ dbeta, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = scale ? scale[c] : 1.0f;
    float dot = 0.0f;
    for (int t = 0; t < T; ++t) {
        const float g = G_t[t * C + c];
        dot += w[c * T + t] * g;
        const float d = g * sc;
        dW[c * T + t] = accumulate ? dW[c * T + t] + d : d;
    }
    if (dgamma) {
        const float dg = (1.0f / sqrtf(var[c] + eps)) * (dot - mean[c] * colsum[c]);
        dgamma[c] = accumulate ? dgamma[c] + dg : dg;
    }
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + colsum[c] : colsum[c];
}

// All depthwise layers of a plan in ONE launch: blockIdx.y = row of the device-resident descriptor table
// (48 launches of ~10 us each at the end of the EfficientDet-D0 backward otherwise).  A layer's tap sums G_t and dy
// column sums arrive as `nslabs` partial rows `slab_stride` floats apart (mi355_dwconv_wgrad_slabs; 1 row for the
// atomic path): a block owns 32 channels, its 16 parts each add a contiguous sixteenth of the rows in index order
// and the parts are combined in a fixed order: bit-reproducible.
constexpr int DWG_MAXT = 25, DWG_CH = 32, DWG_PARTS = 16;
__global__ __launch_bounds__(DWG_CH * DWG_PARTS) void dw_param_grads_table_kernel(const Mi355DwGradDesc* __restrict__ table) {
    __shared__ float s_part[DWG_PARTS - 1][DWG_MAXT + 1][DWG_CH];
    const Mi355DwGradDesc d = table[blockIdx.y];
    if ((int)blockIdx.x * DWG_CH >= d.C) return;               // (uniform per block)
    const int cl = threadIdx.x % DWG_CH, part = threadIdx.x / DWG_CH;
    const int c = blockIdx.x * DWG_CH + cl;
    const bool live = c < d.C;
    const int nsl = d.nslabs > 0 ? d.nslabs : 1;
    const int lo = (int)((long long)nsl * part / DWG_PARTS), hi = (int)((long long)nsl * (part + 1) / DWG_PARTS);
    float g[DWG_MAXT + 1];
#pragma unroll
    for (int t = 0; t <= DWG_MAXT; ++t) g[t] = 0.0f;
    if (live) {
        // (a thread adds its slabs in index order; two slabs' loads are in flight together: the loop is one round trip
        //  per iteration, 512 slabs on 4 parts took 0.5 ms per step)
        int sl = lo;
        for (; sl + 2 <= hi; sl += 2) {
            const float* r0 = d.G_t + (long long)sl * d.slab_stride;
            const float* r1 = r0 + d.slab_stride;
            float v0[DWG_MAXT + 1], v1[DWG_MAXT + 1];
#pragma unroll
            for (int t = 0; t < DWG_MAXT; ++t) {
                v0[t] = t < d.T ? r0[t * d.C + c] : 0.0f;
                v1[t] = t < d.T ? r1[t * d.C + c] : 0.0f;
            }
            v0[DWG_MAXT] = d.colsum ? d.colsum[(long long)sl * d.slab_stride + c] : 0.0f;
            v1[DWG_MAXT] = d.colsum ? d.colsum[(long long)(sl + 1) * d.slab_stride + c] : 0.0f;
#pragma unroll
            for (int t = 0; t <= DWG_MAXT; ++t) g[t] = (g[t] + v0[t]) + v1[t];
        }
        for (; sl < hi; ++sl) {
            const float* row = d.G_t + (long long)sl * d.slab_stride;
#pragma unroll
            for (int t = 0; t < DWG_MAXT; ++t)
                if (t < d.T) g[t] += row[t * d.C + c];
            if (d.colsum) g[DWG_MAXT] += d.colsum[(long long)sl * d.slab_stride + c];
        }
    }
    if (part > 0) {
#pragma unroll
        for (int t = 0; t <= DWG_MAXT; ++t) s_part[part - 1][t][cl] = g[t];
    }
    __syncthreads();
    if (part != 0 || !live) return;
#pragma unroll 1
    for (int q = 0; q < DWG_PARTS - 1; ++q) {                                     // fixed order
#pragma unroll
        for (int t = 0; t <= DWG_MAXT; ++t) g[t] += s_part[q][t][cl];
    }
    const float sc = d.scale ? d.scale[c] : 1.0f;
    float dot = 0.0f;
#pragma unroll
    for (int t = 0; t < DWG_MAXT; ++t) {
        if (t < d.T) {
            dot += d.w[c * d.T + t] * g[t];
            const float v = g[t] * sc;
            d.dW[c * d.T + t] = d.accumulate ? d.dW[c * d.T + t] + v : v;
        }
    }
    const float csum = g[DWG_MAXT];
    if (d.dgamma) {
        const float dg = (1.0f / sqrtf(d.var[c] + d.eps)) * (dot - d.mean[c] * csum);
        d.dgamma[c] = d.accumulate ? d.dgamma[c] + dg : dg;
    }
    if (d.dbeta) d.dbeta[c] = d.accumulate ? d.dbeta[c] + csum : csum;
}

extern "C" int mi355_dw_param_grads_table(const Mi355DwGradDesc* table, int n, int max_c, void* stream) {
    MI355_REQUIRE(table && n > 0 && n <= 65535 && max_c > 0, "dw_param_grads_table: bad args");
    hipLaunchKernelGGL(dw_param_grads_table_kernel, dim3((unsigned)((max_c + DWG_CH - 1) / DWG_CH), (unsigned)n),
                       dim3(DWG_CH * DWG_PARTS), 0, (hipStream_t)stream, table);
    return check_launch("dw_param_grads_table_kernel");
}

extern "C" int mi355_dw_param_grads(const float* G_t, const float* w, const float* scale, const float* colsum,
                                    const float* mean, const float* var, float eps, int C, int k, float* dW,
                                    float* dgamma, float* dbeta, int accumulate, void* stream) {
    MI355_REQUIRE(G_t && w && dW && C > 0 && k > 0, "dw_param_grads: bad args");
    MI355_REQUIRE(!(dgamma || dbeta) || colsum, "dw_param_grads: colsum needed");
    MI355_REQUIRE(!dgamma || (mean && var), "dw_param_grads: mean/var needed");
    hipLaunchKernelGGL(dw_param_grads_kernel, dim3(blocks_for(C)), dim3(256), 0, (hipStream_t)stream, G_t, w, scale,
                       colsum, mean, var, eps, C, k * k, dW, dgamma, dbeta, accumulate);
    return check_launch("dw_param_grads_kernel");
}

// ============================================================================ squeeze-excite
// partial[n][slab][c] = scale * sum over a slab of pixels (summed in a fixed order by the consumer:
// bit-reproducible, no atomics, nothing to zero); grid (slabs, N)
template <typename T, bool SWISH, bool MUL>
__global__ __launch_bounds__(256) void se_reduce_kernel(const T* __restrict__ a,
                                                        const T* __restrict__ z, float* __restrict__ out,
                                                        int HW, int C, int rows_per_block, float scale) {
    __shared__ float s_acc[2048 + 8];
    const int c8n = C >> 3;
    const int n = blockIdx.y;
    const int tid = threadIdx.x;
    const int nphase = 256 / c8n > 0 ? 256 / c8n : 1;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(HW, r0 + rows_per_block);
    for (int cb = 0; cb < c8n; cb += 256) {                     // C/8 may exceed 256 (C up to 2112)
        const int c8 = cb + (tid % min(c8n, 256));
        const int ph = tid / min(c8n, 256);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (c8 < c8n && ph < nphase) {
#pragma unroll 4
            for (int m = r0 + ph; m < r1; m += nphase) {
                float v[8];
                load8(z + ((long long)n * HW + m) * C + c8 * 8, v);
                if (MUL) {
                    float g[8];
                    load8(a + ((long long)n * HW + m) * C + c8 * 8, g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += g[e] * (SWISH ? Act<T>::sw(v[e]) : v[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += SWISH ? Act<T>::sw(v[e]) : v[e];
                }
            }
        }
        if (c8n <= 256) {
            if (c8 < c8n && ph < nphase) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s_acc[ph * C + c8 * 8 + e] = acc[e];
            }
            __syncthreads();
            for (int c = tid; c < C; c += 256) {
                float t = 0.0f;
                for (int r = 0; r < nphase; ++r) t += s_acc[r * C + c];
                out[((long long)n * gridDim.x + blockIdx.x) * C + c] = t * scale;   // slab partial: no atomics
            }
            __syncthreads();
        } else if (c8 < c8n) {
#pragma unroll
            for (int e = 0; e < 8; ++e) out[((long long)n * gridDim.x + blockIdx.x) * C + c8 * 8 + e] = acc[e] * scale;
        }
    }
}

static void se_grid(int N, int HW, int* rows, dim3* grid) {
    // ~512 blocks in total: the per-sample FC kernels add the slab partials with C threads, 128 slabs per sample
    // (2048 blocks) made that sum a third of their 23-32 us
    int slabs = (512 + N - 1) / N;
    if (slabs > (HW + 63) / 64) slabs = (HW + 63) / 64;  // >= 64 rows per block
    if (slabs < 1) slabs = 1;
    *rows = (HW + slabs - 1) / slabs;
    *grid = dim3((unsigned)slabs, (unsigned)N);
}

extern "C" int mi355_se_slabs(int N, int HW, int* slabs) {
    MI355_REQUIRE(N > 0 && HW > 0 && slabs, "se_slabs: bad args");
    int rows; dim3 grid;
    se_grid(N, HW, &rows, &grid);
    *slabs = (int)grid.x;
    return MI355_OK;
}

extern "C" int mi355_se_pool(const void* z, int in_swish, float* pool, int N, int HW, int C, void* stream) {
    MI355_REQUIRE(z && pool && N > 0 && N <= 65535 && HW > 0 && C > 0 && C % 8 == 0, "se_pool: bad args");
    int rows; dim3 grid;
    se_grid(N, HW, &rows, &grid);
    const float sc = 1.0f / (float)HW;
    if (in_swish)
        hipLaunchKernelGGL((se_reduce_kernel<unsigned short, true, false>), grid, dim3(256), 0, (hipStream_t)stream, nullptr,
                           (const unsigned short*)z, pool, HW, C, rows, sc);
    else
        hipLaunchKernelGGL((se_reduce_kernel<unsigned short, false, false>), grid, dim3(256), 0, (hipStream_t)stream, nullptr,
                           (const unsigned short*)z, pool, HW, C, rows, sc);
    return check_launch("se_pool");
}

extern "C" int mi355_se_bwd_reduce(const void* gu, const void* z, float* ggate, int N, int HW, int C, void* stream) {
    MI355_REQUIRE(gu && z && ggate && N > 0 && N <= 65535 && HW > 0 && C > 0 && C % 8 == 0, "se_bwd_reduce: bad args");
    int rows; dim3 grid;
    se_grid(N, HW, &rows, &grid);
    hipLaunchKernelGGL((se_reduce_kernel<unsigned short, true, true>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)gu, (const unsigned short*)z, ggate, HW, C, rows, 1.0f);
    return check_launch("se_bwd_reduce");
}

// one block per sample: r = swish(W1 pool + b1); gate = sigmoid(W2 r + b2)
__global__ __launch_bounds__(256) void se_fc_kernel(const float* __restrict__ part, int slabs,
                                                    float* __restrict__ pool, const float* __restrict__ w1,
                                                    const float* __restrict__ b1, const float* __restrict__ w2,
                                                    const float* __restrict__ b2, float* __restrict__ rpre,
                                                    float* __restrict__ gate, int C, int Cse) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // pool[C] | r[Cse]
    float* s_pool = sm;
    float* s_r = sm + C;
    const int n = blockIdx.x, tid = threadIdx.x;
    {
        // the slab partials in a fixed order; narrow layers (C <= 128: up to 256 slabs from the depthwise forward's
        // tiles) split the slab range over the idle threads and combine the parts in order
        __shared__ float s_sp[256];
        int cl = 256, parts = 1;
        while (parts < 16 && (cl >> 1) >= C) { cl >>= 1; parts <<= 1; }       // cl * parts == 256, cl >= C when parts > 1
        const int pi = tid / cl, c0 = tid - pi * cl;
        const int lo = (int)((long long)slabs * pi / parts), hi = (int)((long long)slabs * (pi + 1) / parts);
        for (int c = c0; c < C; c += cl) {
            float t = 0.0f;
#pragma unroll 8
            for (int sl = lo; sl < hi; ++sl) t += part[((long long)n * slabs + sl) * C + c];
            if (parts == 1) {
                s_pool[c] = t;
                pool[(long long)n * C + c] = t;
            } else {
                s_sp[pi * cl + c] = t;
            }
        }
        if (parts > 1) {                                   // (uniform)
            __syncthreads();
            if (tid < C) {
                float t = s_sp[tid];
                for (int q = 1; q < parts; ++q) t += s_sp[q * cl + tid];
                s_pool[tid] = t;
                pool[(long long)n * C + tid] = t;
            }
        }
    }
    __syncthreads();
    // squeeze: one wave per reduced channel j, lanes stride over c (coalesced rows of w1), shuffle reduction
    // (a thread per j walked its own 1152-float row: 32 us of dependent loads for 55 k MACs)
    // (four rows per wave and round: their loads and shuffle chains overlap; one row per round was 12 dependent rounds
    //  for the 48 x 1152 layers)
    const int wave = tid >> 6, lane = tid & 63;
    for (int j0 = wave * 4; j0 < Cse; j0 += 16) {
        float t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        // (rows past Cse re-read the last row and are dropped: no branch around the loads, so that the unrolled loop
        //  keeps 16 of them in flight; parameters live in a flat arena at 4-B granularity: scalar loads)
#pragma unroll 4
        for (int c = lane; c < C; c += 64) {
            const float pv = s_pool[c];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] += w1[(long long)min(j0 + u, Cse - 1) * C + c] * pv;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t[u] += __shfl_xor(t[u], o, 64);
            if (lane == 0 && j0 + u < Cse) {
                const float v = t[u] + b1[j0 + u];
                rpre[(long long)n * Cse + j0 + u] = v;
                s_r[j0 + u] = swishf_exact(v);
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float t = b2[c];
#pragma unroll 8
        for (int j = 0; j < Cse; ++j) t += w2[c * Cse + j] * s_r[j];
        gate[(long long)n * C + c] = sigmoidf_(t);
    }
}

extern "C" int mi355_se_fc(const float* part, int slabs, float* pool, const float* w1, const float* b1,
                           const float* w2, const float* b2, float* rpre, float* gate, int N, int C, int Cse,
                           void* stream) {
    MI355_REQUIRE(part && slabs > 0 && pool && w1 && b1 && w2 && b2 && rpre && gate && N > 0 && C > 0 && Cse > 0,
                  "se_fc: bad args");
    hipLaunchKernelGGL(se_fc_kernel, dim3(N), dim3(256), (size_t)(C + Cse) * 4, (hipStream_t)stream, part, slabs, pool,
                       w1, b1, w2, b2, rpre, gate, C, Cse);
    return check_launch("se_fc_kernel");
}

__global__ __launch_bounds__(256) void se_fc_bwd_kernel(const float* __restrict__ pool, const float* __restrict__ w1,
                                                        const float* __restrict__ w2, const float* __restrict__ rpre,
                                                        const float* __restrict__ gate, const float* __restrict__ ggate,
                                                        int slabs, float* __restrict__ gpool, float* __restrict__ dw1,
                                                        float* __restrict__ db1, float* __restrict__ dw2,
                                                        float* __restrict__ db2, int C, int Cse) {
    extern __shared__ float sm[];                  // pool[C] | de[C] | r[Cse] | drpre[Cse]
    float* s_pool = sm;
    float* s_de = sm + C;
    float* s_r = sm + 2 * C;
    float* s_dr = sm + 2 * C + Cse;
    // grid = (image, slice): every block rebuilds the small vectors (de, r, dr: O(C * Cse) MACs), then owns
    // 1/gridDim.y of the outer-product atomics and of gpool; the bias gradients are added by slice 0 only
    const int n = blockIdx.x, tid = threadIdx.x, sl_i = blockIdx.y, sl_n = gridDim.y;
    for (int c = tid; c < C; c += 256) {
        s_pool[c] = pool[(long long)n * C + c];
        const float g = gate[(long long)n * C + c];
        float gg = 0.0f;
#pragma unroll 8
        for (int sl = 0; sl < slabs; ++sl) gg += ggate[((long long)n * slabs + sl) * C + c];   // fixed order
        const float de = gg * g * (1.0f - g);
        s_de[c] = de;
        if (sl_i == 0) atomicAdd(db2 + c, de);
    }
    for (int j = tid; j < Cse; j += 256) s_r[j] = swishf_exact(rpre[(long long)n * Cse + j]);
    __syncthreads();
    const int tot2 = C * Cse;
    const int lo2 = (int)((long long)tot2 * sl_i / sl_n), hi2 = (int)((long long)tot2 * (sl_i + 1) / sl_n);
    for (int i = lo2 + tid; i < hi2; i += 256) {
        const int c = i / Cse, j = i - c * Cse;
        atomicAdd(dw2 + i, s_de[c] * s_r[j]);
    }
    // dr[j] = sum_c w2[c][j] * de[c]: a thread owns rows c of w2 (16 contiguous j at a time), block reduction
    __shared__ float s_red[4][16];
    const int wave = tid >> 6, lane = tid & 63;
    for (int j0 = 0; j0 < Cse; j0 += 16) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = 0.0f;
        for (int c = tid; c < C; c += 256) {
            const float de = s_de[c];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (j0 + u < Cse) t[u] += w2[c * Cse + j0 + u] * de;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float v = t[u];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) s_red[wave][u] = v;
        }
        __syncthreads();
        if (tid < 16 && j0 + tid < Cse) {
            const int j = j0 + tid;
            const float v = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
            const float d = v * dswishf_exact(rpre[(long long)n * Cse + j]);
            s_dr[j] = d;
            if (sl_i == 0) atomicAdd(db1 + j, d);
        }
        __syncthreads();
    }
    for (int i = lo2 + tid; i < hi2; i += 256) {       // (same element count: Cse * C)
        const int j = i / C, c = i - j * C;
        atomicAdd(dw1 + i, s_dr[j] * s_pool[c]);
    }
    const int cl = (int)((long long)C * sl_i / sl_n), ch = (int)((long long)C * (sl_i + 1) / sl_n);
    for (int c = cl + tid; c < ch; c += 256) {
        float t = 0.0f;
#pragma unroll 8
        for (int j = 0; j < Cse; ++j) t += w1[j * C + c] * s_dr[j];
        gpool[(long long)n * C + c] = t;
    }
}

extern "C" int mi355_se_fc_bwd(const float* pool, const float* w1, const float* w2, const float* rpre,
                               const float* gate, const float* ggate, int slabs, float* gpool, float* dw1,
                               float* db1, float* dw2, float* db2, int N, int C, int Cse, void* stream) {
    MI355_REQUIRE(pool && w1 && w2 && rpre && gate && ggate && gpool && dw1 && db1 && dw2 && db2 && N > 0 && C > 0 &&
                  Cse > 0, "se_fc_bwd: bad args");
    const int slices = C >= 256 ? 8 : (C >= 64 ? 4 : 1);
    hipLaunchKernelGGL(se_fc_bwd_kernel, dim3(N, slices), dim3(256), (size_t)(2 * C + 2 * Cse) * 4, (hipStream_t)stream, pool,
                       w1, w2, rpre, gate, ggate, slabs, gpool, dw1, db1, dw2, db2, C, Cse);
    return check_launch("se_fc_bwd_kernel");
}

// The same backward without atomics, in two parts.  (1) per sample, the DATA path only: de = ggate * gate (1 - gate),
// dr = (W2^T de) * swish'(rpre), gpool = W1^T dr; grid (N, parts): every block rebuilds de and dr (a few 10 k MACs)
// and owns 1/parts of gpool.  dr: lanes = reduced channels j (rows of W2 are Cse contiguous floats: coalesced, no
// shuffles), the 8 waves split the C channels and their partials are added in wave order.  de and dr are kept for
// (2) mi355_se_param_grads_table: all squeeze-excite layers of a plan in ONE launch at the end of the backward,
// a thread per parameter element sums over the N samples in index order: bit-reproducible (the atomic form above
// issued 2 C Cse atomics per sample: 1.8 M for the 1152 x 48 layers of a batch of 16).
__global__ __launch_bounds__(512) void se_fc_bwd_data_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                             const float* __restrict__ rpre, const float* __restrict__ gate,
                                                             const float* __restrict__ ggate, int slabs,
                                                             float* __restrict__ de_out, float* __restrict__ dr_out,
                                                             float* __restrict__ gpool, int C, int Cse) {
    extern __shared__ float sm[];                  // de[C] | dr[Cse] | wave partials [8][64]
    float* s_de = sm;
    float* s_dr = sm + C;
    float* s_wp = sm + C + Cse;
    const int n = blockIdx.x, tid = threadIdx.x, part = blockIdx.y, parts = gridDim.y;
    for (int c = tid; c < C; c += 512) {
        const float g = gate[(long long)n * C + c];
        float gg = 0.0f;
#pragma unroll 8
        for (int sl = 0; sl < slabs; ++sl) gg += ggate[((long long)n * slabs + sl) * C + c];   // fixed order
        const float de = gg * g * (1.0f - g);
        s_de[c] = de;
        if (part == 0) de_out[(long long)n * C + c] = de;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    const int cw = (C + 7) / 8, c_lo = wave * cw, c_hi = min(C, c_lo + cw);
    for (int j0 = 0; j0 < Cse; j0 += 64) {
        const int j = j0 + lane;
        float t = 0.0f;
        if (j < Cse) {
#pragma unroll 8
            for (int c = c_lo; c < c_hi; ++c) t += w2[c * Cse + j] * s_de[c];
        }
        s_wp[wave * 64 + lane] = t;
        __syncthreads();
        if (wave == 0 && j < Cse) {
            float v = s_wp[lane];
#pragma unroll
            for (int w = 1; w < 8; ++w) v += s_wp[w * 64 + lane];
            const float d = v * dswishf_exact(rpre[(long long)n * Cse + j]);
            s_dr[j] = d;
            if (part == 0) dr_out[(long long)n * Cse + j] = d;
        }
        __syncthreads();
    }
    const int cl = (int)((long long)C * part / parts), ch = (int)((long long)C * (part + 1) / parts);
    for (int c = cl + tid; c < ch; c += 512) {
        float t = 0.0f;
#pragma unroll 8
        for (int j = 0; j < Cse; ++j) t += w1[j * C + c] * s_dr[j];
        gpool[(long long)n * C + c] = t;
    }
}

extern "C" int mi355_se_fc_bwd_data(const float* w1, const float* w2, const float* rpre, const float* gate,
                                    const float* ggate, int slabs, float* de, float* dr, float* gpool, int N, int C,
                                    int Cse, void* stream) {
    MI355_REQUIRE(w1 && w2 && rpre && gate && ggate && slabs > 0 && de && dr && gpool && N > 0 && C > 0 && Cse > 0,
                  "se_fc_bwd_data: bad args");
    const int parts = C >= 512 ? 4 : (C >= 128 ? 2 : 1);
    hipLaunchKernelGGL(se_fc_bwd_data_kernel, dim3(N, parts), dim3(512), (size_t)(C + Cse + 512) * 4, (hipStream_t)stream,
                       w1, w2, rpre, gate, ggate, slabs, de, dr, gpool, C, Cse);
    return check_launch("se_fc_bwd_data_kernel");
}

__global__ void se_param_grads_table_kernel(const Mi355SeGradDesc* __restrict__ table) {
    const Mi355SeGradDesc d = table[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int CC = d.C * d.Cse;
    if (i >= 2 * CC + d.C + d.Cse) return;
    float t = 0.0f;
    if (i < CC) {                                  // dw2[c][j] = sum_n de[n][c] * swish(rpre[n][j])
        const int c = i / d.Cse, j = i - c * d.Cse;
        for (int n = 0; n < d.N; ++n) t += d.de[(long long)n * d.C + c] * swishf_exact(d.rpre[(long long)n * d.Cse + j]);
        d.dw2[i] = t;
    } else if (i < 2 * CC) {                       // dw1[j][c] = sum_n dr[n][j] * pool[n][c]
        const int k = i - CC, j = k / d.C, c = k - j * d.C;
        for (int n = 0; n < d.N; ++n) t += d.dr[(long long)n * d.Cse + j] * d.pool[(long long)n * d.C + c];
        d.dw1[k] = t;
    } else if (i < 2 * CC + d.C) {                 // db2[c] = sum_n de[n][c]
        const int c = i - 2 * CC;
        for (int n = 0; n < d.N; ++n) t += d.de[(long long)n * d.C + c];
        d.db2[c] = t;
    } else {                                       // db1[j] = sum_n dr[n][j]
        const int j = i - 2 * CC - d.C;
        for (int n = 0; n < d.N; ++n) t += d.dr[(long long)n * d.Cse + j];
        d.db1[j] = t;
    }
}

extern "C" int mi355_se_param_grads_table(const Mi355SeGradDesc* table, int n, int max_elems, void* stream) {
    MI355_REQUIRE(table && n > 0 && n <= 65535 && max_elems > 0, "se_param_grads_table: bad args");
    hipLaunchKernelGGL(se_param_grads_table_kernel, dim3(blocks_for(max_elems), (unsigned)n), dim3(256), 0,
                       (hipStream_t)stream, table);
    return check_launch("se_param_grads_table_kernel");
}

template <typename T>
__global__ void se_scale_kernel(const T* __restrict__ z, const float* __restrict__ gate,
                                T* __restrict__ u, long long total, int HW, int C) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c8n = C >> 3;
    const int c8 = (int)(t % c8n);
    const long long n = t / ((long long)c8n * HW);
    float v[8];
    load8(z + t * 8, v);
    const float* g = gate + n * C + c8 * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = Act<T>::sw(v[e]) * g[e];
    store8(u + t * 8, v);
}

extern "C" int mi355_se_scale(const void* z, const float* gate, void* u, int N, int HW, int C, void* stream) {
    MI355_REQUIRE(z && gate && u && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "se_scale: bad args");
    const long long total = (long long)N * HW * (C / 8);
    hipLaunchKernelGGL(se_scale_kernel<unsigned short>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)z, gate, (unsigned short*)u, total, HW, C);
    return check_launch("se_scale_kernel");
}

template <typename T>
__global__ void se_bwd_apply_kernel(const T* __restrict__ gu, const T* __restrict__ z,
                                    const float* __restrict__ gate, const float* __restrict__ gpool,
                                    T* __restrict__ gz, long long total, int HW, int C) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c8n = C >> 3;
    const int c8 = (int)(t % c8n);
    const long long n = t / ((long long)c8n * HW);
    float g[8], v[8];
    load8(gu + t * 8, g);
    load8(z + t * 8, v);
    const float* ga = gate + n * C + c8 * 8;
    const float* gp = gpool + n * C + c8 * 8;
    const float inv = 1.0f / (float)HW;
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = se_apply_mix(g[e], ga[e], gp[e], inv, Act<T>::dsw(v[e]));
    store8(gz + t * 8, g);
}

extern "C" int mi355_se_bwd_apply(const void* gu, const void* z, const float* gate, const float* gpool, void* gz,
                                  int N, int HW, int C, void* stream) {
    MI355_REQUIRE(gu && z && gate && gpool && gz && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "se_bwd_apply: bad args");
    const long long total = (long long)N * HW * (C / 8);
    hipLaunchKernelGGL(se_bwd_apply_kernel<unsigned short>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)gu, (const unsigned short*)z, gate, gpool, (unsigned short*)gz, total, HW, C);
    return check_launch("se_bwd_apply_kernel");
}

// ============================================================================ BiFPN
__global__ void fusion_weights_fwd_kernel(const float* p, int n, float eps, float* w) {
    if (threadIdx.x != 0) return;
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s += fmaxf(p[i], 0.0f);
    for (int i = 0; i < 3; ++i) w[i] = i < n ? fmaxf(p[i], 0.0f) / (s + eps) : 0.0f;
}
extern "C" int mi355_fusion_weights_fwd(const float* p, int n, float eps, float* w, void* stream) {
    MI355_REQUIRE(p && w && n >= 1 && n <= 3, "fusion_weights_fwd: bad args");
    hipLaunchKernelGGL(fusion_weights_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p, n, eps, w);
    return check_launch("fusion_weights_fwd_kernel");
}
// all fusion nodes of a plan: rows of (p, n, w) pointers, one thread each
__global__ void fusion_weights_table_kernel(const Mi355FusionDesc* __restrict__ table, int rows, float eps) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const Mi355FusionDesc d = table[r];
    float s = 0.0f;
    for (int i = 0; i < d.n; ++i) s += fmaxf(d.p[i], 0.0f);
    for (int i = 0; i < 3; ++i) d.w[i] = i < d.n ? fmaxf(d.p[i], 0.0f) / (s + eps) : 0.0f;
}
extern "C" int mi355_fusion_weights_table(const Mi355FusionDesc* table, int rows, float eps, void* stream) {
    MI355_REQUIRE(table && rows > 0, "fusion_weights_table: bad args");
    hipLaunchKernelGGL(fusion_weights_table_kernel, dim3(blocks_for(rows)), dim3(256), 0, (hipStream_t)stream, table,
                       rows, eps);
    return check_launch("fusion_weights_table_kernel");
}
__global__ void fusion_weights_bwd_kernel(const float* p, int n, float eps, const float* gw, float* gp) {
    if (threadIdx.x != 0) return;
    float s = eps, dot = 0.0f;
    for (int i = 0; i < n; ++i) { const float r = fmaxf(p[i], 0.0f); s += r; dot += gw[i] * r; }
    // w_i = r_i / s  ->  dL/dr_j = gw_j / s - dot / s^2 ; relu'
    for (int j = 0; j < n; ++j) gp[j] = p[j] > 0.0f ? (gw[j] / s - dot / (s * s)) : 0.0f;
}
// one wave per node: gw is either the three sums themselves (nparts == 0) or `nparts` partial rows of 4 floats written by
// the blocks of mi355_bifpn_node_bwd; lane l adds rows l, l + 64, ... in order and the lanes are combined by a fixed
// shuffle tree: bit-reproducible
__global__ __launch_bounds__(64) void fusion_weights_bwd_table_kernel(const Mi355FusionBwdDesc* __restrict__ table, int rows, float eps) {
    const Mi355FusionBwdDesc d = table[blockIdx.x];
    const int lane = threadIdx.x;
    float g[3] = {0.0f, 0.0f, 0.0f};
    if (d.nparts >= 1) {
        for (int b = lane; b < d.nparts; b += 64)
#pragma unroll
            for (int i = 0; i < 3; ++i) g[i] += d.gw[b * 4 + i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) g[i] += __shfl_xor(g[i], o, 64);
    } else {
        for (int i = 0; i < d.n; ++i) g[i] = d.gw[i];
    }
    if (lane != 0) return;
    float s = eps, dot = 0.0f;
    for (int i = 0; i < d.n; ++i) { const float v = fmaxf(d.p[i], 0.0f); s += v; dot += g[i] * v; }
    for (int j = 0; j < d.n; ++j) d.gp[j] = d.p[j] > 0.0f ? (g[j] / s - dot / (s * s)) : 0.0f;
}
extern "C" int mi355_fusion_weights_bwd_table(const Mi355FusionBwdDesc* table, int rows, float eps, void* stream) {
    MI355_REQUIRE(table && rows > 0, "fusion_weights_bwd_table: bad args");
    hipLaunchKernelGGL(fusion_weights_bwd_table_kernel, dim3((unsigned)rows), dim3(64), 0, (hipStream_t)stream,
                       table, rows, eps);
    return check_launch("fusion_weights_bwd_table_kernel");
}
extern "C" int mi355_fusion_weights_bwd(const float* p, int n, float eps, const float* gw, float* gp, void* stream) {
    MI355_REQUIRE(p && gw && gp && n >= 1 && n <= 3, "fusion_weights_bwd: bad args");
    hipLaunchKernelGGL(fusion_weights_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p, n, eps, gw, gp);
    return check_launch("fusion_weights_bwd_kernel");
}

template <typename T, bool WGRAD>
__global__ __launch_bounds__(256) void bifpn_node_kernel(const T* __restrict__ gz,
                                                         const T* __restrict__ in0,
                                                         const T* __restrict__ in1, int up1,
                                                         const T* __restrict__ in2,
                                                         const float* __restrict__ w, T* __restrict__ z,
                                                         float* __restrict__ gw, int N, int H, int W, int C) {
    const int c8n = C >> 3;
    const long long total = (long long)N * H * W * c8n;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (t < total) {
        int c8, x, y, n;
        split_cell(t, c8n, W, H, c8, x, y, n);
        float a[8], b[8], c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        load8(in0 + t * 8, a);
        const long long o1 = up1 ? ((((long long)n * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * C + c8 * 8) : t * 8;
        load8(in1 + o1, b);
        if (in2) load8(in2 + t * 8, c);
        if (WGRAD) {
            float g[8];
            load8(gz + t * 8, g);
#pragma unroll
            for (int e = 0; e < 8; ++e) { d0 += g[e] * a[e]; d1 += g[e] * b[e]; d2 += g[e] * c[e]; }
        } else {
            const float w0 = w[0], w1 = w[1], w2 = w[2];
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = node_mix(w0, a[e], w1, b[e], w2, c[e]);
            store8(z + t * 8, o);
        }
    }
    if (WGRAD) {
        __shared__ float red[3][4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            d0 += __shfl_down(d0, o, 64); d1 += __shfl_down(d1, o, 64); d2 += __shfl_down(d2, o, 64);
        }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d0; red[1][threadIdx.x >> 6] = d1; red[2][threadIdx.x >> 6] = d2; }
        __syncthreads();
        if (threadIdx.x < 3) atomicAdd(gw + threadIdx.x, red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
    }
}

extern "C" int mi355_bifpn_node_fwd(const void* in0, const void* in1, int up1, const void* in2, const float* w,
                                    void* z, int N, int H, int W, int C, void* stream) {
    MI355_REQUIRE(in0 && in1 && w && z && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "bifpn_node_fwd: bad args");
    MI355_REQUIRE(!up1 || (H % 2 == 0 && W % 2 == 0), "bifpn_node_fwd: upsample needs even H, W");
    const long long total = (long long)N * H * W * (C / 8);
    hipLaunchKernelGGL((bifpn_node_kernel<unsigned short, false>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, nullptr,
                       (const unsigned short*)in0, (const unsigned short*)in1, up1, (const unsigned short*)in2, w,
                       (unsigned short*)z, nullptr, N, H, W, C);
    return check_launch("bifpn_node_kernel<fwd>");
}

extern "C" int mi355_bifpn_node_wgrad(const void* gz, const void* in0, const void* in1, int up1, const void* in2,
                                      float* gw, int N, int H, int W, int C, void* stream) {
    MI355_REQUIRE(gz && in0 && in1 && gw && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "bifpn_node_wgrad: bad args");
    const long long total = (long long)N * H * W * (C / 8);
    hipLaunchKernelGGL((bifpn_node_kernel<unsigned short, true>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)gz, (const unsigned short*)in0, (const unsigned short*)in1, up1,
                       (const unsigned short*)in2, nullptr, nullptr, gw, N, H, W, C);
    return check_launch("bifpn_node_kernel<wgrad>");
}

// The whole backward of a BiFPN node in one pass over gz (a node's backward was its fusion-weight reduction plus one
// axpy launch per input: 4-5 launches of 3-8 us for 2-3 MB of tensors): g_i (+)= w_i * gz for the three inputs (the
// half-resolution input of a top-down node takes the sum of its 2 x 2 gz pixels: a thread owns one such quad, so
// nothing is added across threads) and the block's partial <gz, in_i> sums go to its own row of gw (no atomics: bit-reproducible).
template <typename T>
__global__ __launch_bounds__(256) void bifpn_node_bwd_kernel(const T* __restrict__ gz, const T* __restrict__ in0,
                                                             const T* __restrict__ in1, int up1,
                                                             const T* __restrict__ in2, const float* __restrict__ w,
                                                             float* __restrict__ gw, T* __restrict__ g0, int acc0,
                                                             T* __restrict__ g1, int acc1, T* __restrict__ g2, int acc2,
                                                             int N, int H, int W, int C) {
    const int c8n = C >> 3;
    const int step = up1 ? 2 : 1, Hq = H / step, Wq = W / step;
    const long long total = (long long)N * Hq * Wq * c8n;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (t < total) {
        int c8, xq, yq, n;
        split_cell(t, c8n, Wq, Hq, c8, xq, yq, n);
        const float w0 = w[0], w1 = w[1], w2 = w[2];
        float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int dy = 0; dy < step; ++dy)
            for (int dx = 0; dx < step; ++dx) {
                const long long o = ((((long long)n * H + yq * step + dy) * W + xq * step + dx) * c8n + c8) * 8;
                float g[8], a[8], v[8];
                load8(gz + o, g);
                load8(in0 + o, a);
#pragma unroll
                for (int e = 0; e < 8; ++e) { d0 += g[e] * a[e]; sum[e] += g[e]; v[e] = w0 * g[e]; }
                if (acc0) {
                    float old[8];
                    load8(g0 + o, old);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += old[e];
                }
                store8(g0 + o, v);
                if (in2) {
                    float c[8];
                    load8(in2 + o, c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { d2 += g[e] * c[e]; v[e] = w2 * g[e]; }
                    if (acc2) {
                        float old[8];
                        load8(g2 + o, old);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += old[e];
                    }
                    store8(g2 + o, v);
                }
            }
        const long long o1 = t * 8;                       // in1 / g1: [N][Hq][Wq][C]
        float b[8], v[8];
        load8(in1 + o1, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) { d1 += sum[e] * b[e]; v[e] = w1 * sum[e]; }
        if (acc1) {
            float old[8];
            load8(g1 + o1, old);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += old[e];
        }
        store8(g1 + o1, v);
    }
    if (gw) {
        __shared__ float red[3][4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            d0 += __shfl_down(d0, o, 64); d1 += __shfl_down(d1, o, 64); d2 += __shfl_down(d2, o, 64);
        }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d0; red[1][threadIdx.x >> 6] = d1; red[2][threadIdx.x >> 6] = d2; }
        __syncthreads();
        // this block's row of the partial sums (plain stores; mi355_fusion_weights_bwd_table adds the rows in order)
        if (threadIdx.x < 3) gw[(long long)blockIdx.x * 4 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    }
}

extern "C" int mi355_bifpn_node_bwd_parts(int N, int H, int W, int C, int up1, int* nparts) {
    MI355_REQUIRE(nparts && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "bifpn_node_bwd_parts: bad args");
    const int step = up1 ? 2 : 1;
    *nparts = (int)blocks_for((long long)N * (H / step) * (W / step) * (C / 8));
    return MI355_OK;
}

template <typename T>
static int launch_bifpn_node_bwd(const void* gz, const void* in0, const void* in1, int up1, const void* in2, const float* w,
                                 float* gw, void* g0, int acc0, void* g1, int acc1, void* g2, int acc2, int N, int H, int W,
                                 int C, void* stream) {
    MI355_REQUIRE(gz && in0 && in1 && w && g0 && g1 && (!in2 || g2) && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0,
                  "bifpn_node_bwd: bad args");
    MI355_REQUIRE(!up1 || (H % 2 == 0 && W % 2 == 0), "bifpn_node_bwd: upsample needs even H, W");
    const int step = up1 ? 2 : 1;
    const long long total = (long long)N * (H / step) * (W / step) * (C / 8);
    hipLaunchKernelGGL(bifpn_node_bwd_kernel<T>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const T*)gz,
                       (const T*)in0, (const T*)in1, up1, (const T*)in2, w, gw, (T*)g0, acc0, (T*)g1, acc1, (T*)g2, acc2,
                       N, H, W, C);
    return check_launch("bifpn_node_bwd_kernel");
}

extern "C" int mi355_bifpn_node_bwd(const void* gz, const void* in0, const void* in1, int up1, const void* in2,
                                    const float* w, float* gw, void* g0, int acc0, void* g1, int acc1, void* g2, int acc2,
                                    int N, int H, int W, int C, void* stream) {
    return launch_bifpn_node_bwd<unsigned short>(gz, in0, in1, up1, in2, w, gw, g0, acc0, g1, acc1, g2, acc2, N, H, W, C, stream);
}

extern "C" int mi355_bifpn_node_bwd_f32(const void* gz, const void* in0, const void* in1, int up1, const void* in2,
                                        const float* w, float* gw, void* g0, int acc0, void* g1, int acc1, void* g2,
                                        int acc2, int N, int H, int W, int C, void* stream) {
    return launch_bifpn_node_bwd<float>(gz, in0, in1, up1, in2, w, gw, g0, acc0, g1, acc1, g2, acc2, N, H, W, C, stream);
}

template <typename T>
__global__ void axpy_kernel(const T* __restrict__ src, const float* __restrict__ w, int wi,
                            T* __restrict__ dst, long long n8, int accumulate) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n8) return;
    const float a = w ? w[wi] : 1.0f;
    float v[8], o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    load8(src + t * 8, v);
    if (accumulate) load8(dst + t * 8, o);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] += a * v[e];
    store8(dst + t * 8, o);
}

extern "C" int mi355_axpy_bf16(const void* src, const float* w, int wi, void* dst, int64_t n, int accumulate,
                               void* stream) {
    MI355_REQUIRE(src && dst && n > 0 && n % 8 == 0, "axpy: n must be a multiple of 8");
    hipLaunchKernelGGL(axpy_kernel<unsigned short>, dim3(blocks_for(n / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)src, w, wi, (unsigned short*)dst, (long long)(n / 8), accumulate);
    return check_launch("axpy_kernel");
}

// drop-connect (utils_efficientnet.py:85-94): dst = src / keep * floor(keep + u[n]) (+ addend), one uniform per image
template <typename T>
__global__ void drop_connect_kernel(const T* __restrict__ src, const float* __restrict__ u, float keep,
                                    const T* __restrict__ addend, T* __restrict__ dst,
                                    long long per_image8) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_image8) return;
    const int n = blockIdx.y;
    const float b = floorf(keep + u[n]);
    const long long t = (long long)n * per_image8 + i;
    float v[8], o[8];
    load8(src + t * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = v[e] / keep * b;
    if (addend) {
        float a[8];
        load8(addend + t * 8, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += a[e];
    }
    store8(dst + t * 8, o);
}

extern "C" int mi355_drop_connect(const void* src, const float* u, float keep, const void* addend, void* dst, int N,
                                  int64_t per_image, void* stream) {
    MI355_REQUIRE(src && u && dst && N > 0 && N <= 65535 && per_image > 0 && per_image % 8 == 0,
                  "drop_connect: bad args (per_image must be a multiple of 8)");
    MI355_REQUIRE(keep > 0.0f && keep <= 1.0f, "drop_connect: keep probability %g outside (0, 1]", (double)keep);
    hipLaunchKernelGGL(drop_connect_kernel<unsigned short>, dim3(blocks_for(per_image / 8), (unsigned)N), dim3(256), 0,
                       (hipStream_t)stream, (const unsigned short*)src, u, keep, (const unsigned short*)addend,
                       (unsigned short*)dst, (long long)(per_image / 8));
    return check_launch("drop_connect_kernel");
}

// backward of out = branch / keep * floor(keep + u[n]) + skip in one pass over dy: g_branch = dy / keep * b[n],
// g_skip (+)= dy (the two launches `drop_connect` + `axpy` read dy twice)
template <typename T>
__global__ void drop_connect_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ u, float keep,
                                        T* __restrict__ g_branch, T* __restrict__ g_skip, int acc_skip,
                                        long long per_image8) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_image8) return;
    const int n = blockIdx.y;
    const float b = floorf(keep + u[n]);
    const long long t = (long long)n * per_image8 + i;
    float v[8], o[8];
    load8(dy + t * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = v[e] / keep * b;
    store8(g_branch + t * 8, o);
    if (acc_skip) {
        float a[8];
        load8(g_skip + t * 8, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += a[e];
    }
    store8(g_skip + t * 8, v);
}

template <typename T>
static int launch_drop_connect_bwd(const void* dy, const float* u, float keep, void* g_branch, void* g_skip, int acc_skip,
                                   int N, int64_t per_image, void* stream) {
    MI355_REQUIRE(dy && u && g_branch && g_skip && N > 0 && N <= 65535 && per_image > 0 && per_image % 8 == 0,
                  "drop_connect_bwd: bad args (per_image must be a multiple of 8)");
    MI355_REQUIRE(keep > 0.0f && keep <= 1.0f, "drop_connect_bwd: keep probability %g outside (0, 1]", (double)keep);
    hipLaunchKernelGGL(drop_connect_bwd_kernel<T>, dim3(blocks_for(per_image / 8), (unsigned)N), dim3(256), 0,
                       (hipStream_t)stream, (const T*)dy, u, keep, (T*)g_branch, (T*)g_skip, acc_skip,
                       (long long)(per_image / 8));
    return check_launch("drop_connect_bwd_kernel");
}
extern "C" int mi355_drop_connect_bwd(const void* dy, const float* u, float keep, void* g_branch, void* g_skip,
                                      int acc_skip, int N, int64_t per_image, void* stream) {
    return launch_drop_connect_bwd<unsigned short>(dy, u, keep, g_branch, g_skip, acc_skip, N, per_image, stream);
}
extern "C" int mi355_drop_connect_bwd_f32(const void* dy, const float* u, float keep, void* g_branch, void* g_skip,
                                          int acc_skip, int N, int64_t per_image, void* stream) {
    return launch_drop_connect_bwd<float>(dy, u, keep, g_branch, g_skip, acc_skip, N, per_image, stream);
}

template <typename T>
__global__ void upsample_bwd_axpy_kernel(const T* __restrict__ src, const float* __restrict__ w, int wi,
                                         T* __restrict__ dst, int N, int H, int W, int C, int accumulate) {
    const int c8n = C >> 3, P = H >> 1, Q = W >> 1;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * P * Q * c8n) return;
    int c8, q, p, n;
    split_cell(t, c8n, Q, P, c8, q, p, n);
    const float a = w ? w[wi] : 1.0f;
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (accumulate) load8(dst + t * 8, o);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[8];
            load8(src + (((long long)n * H + 2 * p + i) * W + 2 * q + j) * C + c8 * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] += a * acc[e];
    store8(dst + t * 8, o);
}

extern "C" int mi355_upsample_bwd_axpy(const void* src, const float* w, int wi, void* dst, int N, int H, int W, int C,
                                       int accumulate, void* stream) {
    MI355_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0,
                  "upsample_bwd_axpy: bad args");
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(upsample_bwd_axpy_kernel<unsigned short>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)src, w, wi, (unsigned short*)dst, N, H, W, C, accumulate);
    return check_launch("upsample_bwd_axpy_kernel");
}

// ============================================================================ "same" max-pool (k3 s2)
static void same_pad(int n, int k, int s, int* before, int* out) {
    *out = (n + s - 1) / s;
    const int extra = s * (*out - 1) + 1 + (k - 1) - n;
    *before = extra / 2;
}

template <typename T>
__global__ void maxpool_same_fwd_kernel(const T* __restrict__ in, T* __restrict__ out,
                                        signed char* __restrict__ idx, int N, int H, int W, int C, int P, int Q,
                                        int pad_t, int pad_l) {
    const int c8n = C >> 3;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * P * Q * c8n) return;
    int c8, q, p, n;
    split_cell(t, c8n, Q, P, c8, q, p, n);
    float best[8];
    int bi[8];
    bool first = true;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int h = 2 * p - pad_t + a, w = 2 * q - pad_l + b;
            const bool inb = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};                 // zero padding takes part in the max
            if (inb) load8(in + (((long long)n * H + h) * W + w) * C + c8 * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (first || v[e] > best[e]) { best[e] = v[e]; bi[e] = inb ? a * 3 + b : -1; }
            first = false;
        }
    unsigned long long packed = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) packed |= (unsigned long long)(bi[e] & 0xff) << (8 * e);
    store8(out + t * 8, best);
    *reinterpret_cast<unsigned long long*>(idx + t * 8) = packed;
}

extern "C" int mi355_maxpool_same_fwd(const void* in, void* out, int8_t* idx, int N, int H, int W, int C,
                                      void* stream) {
    MI355_REQUIRE(in && out && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool_same_fwd: bad args");
    int pt, pl, P, Q;
    same_pad(H, 3, 2, &pt, &P);
    same_pad(W, 3, 2, &pl, &Q);
    const long long total = (long long)N * P * Q * (C / 8);
    hipLaunchKernelGGL(maxpool_same_fwd_kernel<unsigned short>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)in, (unsigned short*)out, (signed char*)idx, N, H, W, C, P, Q, pt, pl);
    return check_launch("maxpool_same_fwd_kernel");
}

template <typename T>
__global__ void maxpool_same_bwd_kernel(const T* __restrict__ gout, const signed char* __restrict__ idx,
                                        T* __restrict__ gin, int N, int H, int W, int C, int P, int Q,
                                        int pad_t, int pad_l, int accumulate) {
    const int c8n = C >> 3;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * H * W * c8n) return;
    int c8, w, h, n;
    split_cell(t, c8n, W, H, c8, w, h, n);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (accumulate) load8(gin + t * 8, acc);
    for (int a = 0; a < 3; ++a) {
        const int hp = h + pad_t - a;
        if (hp < 0 || (hp & 1)) continue;
        const int p = hp >> 1;
        if (p >= P) continue;
        for (int b = 0; b < 3; ++b) {
            const int wq = w + pad_l - b;
            if (wq < 0 || (wq & 1)) continue;
            const int q = wq >> 1;
            if (q >= Q) continue;
            const long long o = (((long long)n * P + p) * Q + q) * C + c8 * 8;
            const unsigned long long pk = *reinterpret_cast<const unsigned long long*>(idx + o);
            float g[8];
            load8(gout + o, g);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if ((int)(signed char)((pk >> (8 * e)) & 0xff) == a * 3 + b) acc[e] += g[e];
        }
    }
    store8(gin + t * 8, acc);
}

extern "C" int mi355_maxpool_same_bwd(const void* gout, const int8_t* idx, void* gin, int N, int H, int W, int C,
                                      int accumulate, void* stream) {
    MI355_REQUIRE(gout && idx && gin && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool_same_bwd: bad args");
    int pt, pl, P, Q;
    same_pad(H, 3, 2, &pt, &P);
    same_pad(W, 3, 2, &pl, &Q);
    const long long total = (long long)N * H * W * (C / 8);
    hipLaunchKernelGGL(maxpool_same_bwd_kernel<unsigned short>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)gout, (const signed char*)idx, (unsigned short*)gin, N, H, W, C, P, Q, pt,
                       pl, accumulate);
    return check_launch("maxpool_same_bwd_kernel");
}


// ============================================================================ fp32-accurate mode twins
// Same arguments as the bf16 entry points above with every activation / gradient tensor fp32 (C a multiple of 8); the
// generic kernels instantiated over float (exact swish).  A precision mode: none of the tiled / windowed fast paths.
extern "C" int mi355_dwconv_fwd_f32(const void* in, int in_swish, const float* w_t, const float* bias, void* out, int N,
                                    int H, int W, int C, int k, int stride, int pad_t, int pad_l, int P, int Q,
                                    void* stream) {
    MI355_REQUIRE(in && w_t && out && N > 0 && C > 0 && C % 8 == 0 && k > 0 && stride > 0, "dwconv_fwd_f32: bad args");
    const long long total = (long long)N * P * Q * (C / 8);
    if (in_swish)
        hipLaunchKernelGGL((dwconv_fwd_kernel<float, true>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)in, w_t, bias, (float*)out, N, H, W, C, k, stride, pad_t, pad_l, P, Q);
    else
        hipLaunchKernelGGL((dwconv_fwd_kernel<float, false>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)in, w_t, bias, (float*)out, N, H, W, C, k, stride, pad_t, pad_l, P, Q);
    return check_launch("dwconv_fwd_kernel<float>");
}

extern "C" int mi355_dwconv_dgrad_f32(const void* dy, const float* w_t, void* dx, const void* zin, int N, int H, int W,
                                      int C, int k, int stride, int pad_t, int pad_l, int P, int Q, int accumulate,
                                      void* stream) {
    MI355_REQUIRE(dy && w_t && dx && N > 0 && C > 0 && C % 8 == 0 && k > 0 && stride > 0, "dwconv_dgrad_f32: bad args");
    const long long total = (long long)N * H * W * (C / 8);
    hipLaunchKernelGGL(dwconv_dgrad_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)dy, w_t, (float*)dx, (const float*)zin, N, H, W, C, k, stride, pad_t, pad_l, P, Q,
                       accumulate);
    return check_launch("dwconv_dgrad_kernel<float>");
}

extern "C" int mi355_dwconv_wgrad_f32(const void* x, int in_swish, const void* dy, float* dw_t, float* colsum, int N,
                                      int H, int W, int C, int k, int stride, int pad_t, int pad_l, int P, int Q,
                                      void* stream) {
    MI355_REQUIRE(x && dy && dw_t && N > 0 && C > 0 && C % 8 == 0 && k > 0, "dwconv_wgrad_f32: bad args");
    const int nrows = N * P;
    const int items = k * k * (C / 8);
    const int bx = (items + DW_IPB - 1) / DW_IPB;
    int chunks = 2048 / bx;
    if (chunks > 128) chunks = 128;
    if (chunks < 1) chunks = 1;
    int rows = (nrows + chunks - 1) / chunks;
    if (rows < 1) rows = 1;
    const int by = (nrows + rows - 1) / rows;
    dim3 grid((unsigned)bx, (unsigned)by);
    if (in_swish)
        hipLaunchKernelGGL((dwconv_wgrad_kernel<float, true>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)x,
                           (const float*)dy, dw_t, colsum, N, H, W, C, k, stride, pad_t, pad_l, P, Q, rows);
    else
        hipLaunchKernelGGL((dwconv_wgrad_kernel<float, false>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)x,
                           (const float*)dy, dw_t, colsum, N, H, W, C, k, stride, pad_t, pad_l, P, Q, rows);
    return check_launch("dwconv_wgrad_kernel<float>");
}

extern "C" int mi355_se_pool_f32(const void* z, int in_swish, float* pool, int N, int HW, int C, void* stream) {
    MI355_REQUIRE(z && pool && N > 0 && N <= 65535 && HW > 0 && C > 0 && C % 8 == 0, "se_pool_f32: bad args");
    int rows; dim3 grid;
    se_grid(N, HW, &rows, &grid);
    const float sc = 1.0f / (float)HW;
    if (in_swish)
        hipLaunchKernelGGL((se_reduce_kernel<float, true, false>), grid, dim3(256), 0, (hipStream_t)stream, nullptr,
                           (const float*)z, pool, HW, C, rows, sc);
    else
        hipLaunchKernelGGL((se_reduce_kernel<float, false, false>), grid, dim3(256), 0, (hipStream_t)stream, nullptr,
                           (const float*)z, pool, HW, C, rows, sc);
    return check_launch("se_pool_f32");
}

extern "C" int mi355_se_bwd_reduce_f32(const void* gu, const void* z, float* ggate, int N, int HW, int C, void* stream) {
    MI355_REQUIRE(gu && z && ggate && N > 0 && N <= 65535 && HW > 0 && C > 0 && C % 8 == 0, "se_bwd_reduce_f32: bad args");
    int rows; dim3 grid;
    se_grid(N, HW, &rows, &grid);
    hipLaunchKernelGGL((se_reduce_kernel<float, true, true>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)gu,
                       (const float*)z, ggate, HW, C, rows, 1.0f);
    return check_launch("se_bwd_reduce_f32");
}

extern "C" int mi355_se_scale_f32(const void* z, const float* gate, void* u, int N, int HW, int C, void* stream) {
    MI355_REQUIRE(z && gate && u && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "se_scale_f32: bad args");
    const long long total = (long long)N * HW * (C / 8);
    hipLaunchKernelGGL(se_scale_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)z,
                       gate, (float*)u, total, HW, C);
    return check_launch("se_scale_kernel<float>");
}

extern "C" int mi355_se_bwd_apply_f32(const void* gu, const void* z, const float* gate, const float* gpool, void* gz,
                                      int N, int HW, int C, void* stream) {
    MI355_REQUIRE(gu && z && gate && gpool && gz && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "se_bwd_apply_f32: bad args");
    const long long total = (long long)N * HW * (C / 8);
    hipLaunchKernelGGL(se_bwd_apply_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)gu, (const float*)z, gate, gpool, (float*)gz, total, HW, C);
    return check_launch("se_bwd_apply_kernel<float>");
}

extern "C" int mi355_bifpn_node_fwd_f32(const void* in0, const void* in1, int up1, const void* in2, const float* w,
                                        void* z, int N, int H, int W, int C, void* stream) {
    MI355_REQUIRE(in0 && in1 && w && z && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "bifpn_node_fwd_f32: bad args");
    MI355_REQUIRE(!up1 || (H % 2 == 0 && W % 2 == 0), "bifpn_node_fwd_f32: upsample needs even H, W");
    const long long total = (long long)N * H * W * (C / 8);
    hipLaunchKernelGGL((bifpn_node_kernel<float, false>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       nullptr, (const float*)in0, (const float*)in1, up1, (const float*)in2, w, (float*)z, nullptr, N, H,
                       W, C);
    return check_launch("bifpn_node_kernel<float, fwd>");
}

extern "C" int mi355_bifpn_node_wgrad_f32(const void* gz, const void* in0, const void* in1, int up1, const void* in2,
                                          float* gw, int N, int H, int W, int C, void* stream) {
    MI355_REQUIRE(gz && in0 && in1 && gw && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "bifpn_node_wgrad_f32: bad args");
    const long long total = (long long)N * H * W * (C / 8);
    hipLaunchKernelGGL((bifpn_node_kernel<float, true>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)gz, (const float*)in0, (const float*)in1, up1, (const float*)in2, nullptr, nullptr,
                       gw, N, H, W, C);
    return check_launch("bifpn_node_kernel<float, wgrad>");
}

extern "C" int mi355_axpy_f32(const void* src, const float* w, int wi, void* dst, int64_t n, int accumulate, void* stream) {
    MI355_REQUIRE(src && dst && n > 0 && n % 8 == 0, "axpy_f32: n must be a multiple of 8");
    hipLaunchKernelGGL(axpy_kernel<float>, dim3(blocks_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const float*)src,
                       w, wi, (float*)dst, (long long)(n / 8), accumulate);
    return check_launch("axpy_kernel<float>");
}

extern "C" int mi355_drop_connect_f32(const void* src, const float* u, float keep, const void* addend, void* dst, int N,
                                      int64_t per_image, void* stream) {
    MI355_REQUIRE(src && u && dst && N > 0 && N <= 65535 && per_image > 0 && per_image % 8 == 0,
                  "drop_connect_f32: bad args (per_image must be a multiple of 8)");
    MI355_REQUIRE(keep > 0.0f && keep <= 1.0f, "drop_connect_f32: keep probability %g outside (0, 1]", (double)keep);
    hipLaunchKernelGGL(drop_connect_kernel<float>, dim3(blocks_for(per_image / 8), (unsigned)N), dim3(256), 0,
                       (hipStream_t)stream, (const float*)src, u, keep, (const float*)addend, (float*)dst,
                       (long long)(per_image / 8));
    return check_launch("drop_connect_kernel<float>");
}

extern "C" int mi355_upsample_bwd_axpy_f32(const void* src, const float* w, int wi, void* dst, int N, int H, int W, int C,
                                           int accumulate, void* stream) {
    MI355_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0,
                  "upsample_bwd_axpy_f32: bad args");
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(upsample_bwd_axpy_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)src, w, wi, (float*)dst, N, H, W, C, accumulate);
    return check_launch("upsample_bwd_axpy_kernel<float>");
}

extern "C" int mi355_maxpool_same_fwd_f32(const void* in, void* out, int8_t* idx, int N, int H, int W, int C, void* stream) {
    MI355_REQUIRE(in && out && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool_same_fwd_f32: bad args");
    int pt, pl, P, Q;
    same_pad(H, 3, 2, &pt, &P);
    same_pad(W, 3, 2, &pl, &Q);
    const long long total = (long long)N * P * Q * (C / 8);
    hipLaunchKernelGGL(maxpool_same_fwd_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)in, (float*)out, (signed char*)idx, N, H, W, C, P, Q, pt, pl);
    return check_launch("maxpool_same_fwd_kernel<float>");
}

extern "C" int mi355_maxpool_same_bwd_f32(const void* gout, const int8_t* idx, void* gin, int N, int H, int W, int C,
                                          int accumulate, void* stream) {
    MI355_REQUIRE(gout && idx && gin && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "maxpool_same_bwd_f32: bad args");
    int pt, pl, P, Q;
    same_pad(H, 3, 2, &pt, &P);
    same_pad(W, 3, 2, &pl, &Q);
    const long long total = (long long)N * H * W * (C / 8);
    hipLaunchKernelGGL(maxpool_same_bwd_kernel<float>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)gout, (const signed char*)idx, (float*)gin, N, H, W, C, P, Q, pt, pl, accumulate);
    return check_launch("maxpool_same_bwd_kernel<float>");
}
