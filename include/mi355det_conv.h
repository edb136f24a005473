/* mi355det — C ABI of the convolution engine (gfx950 MFMA implicit GEMM, NHWC bf16).
 *
 * Replaces the ATen/cuDNN calls behind the reference's `nn.Conv2d` + `BatchNorm2d(eval)` +
 * ReLU / residual-add / sigmoid / nearest-upsample-add chains:
 *   architectures/retinanet.py:48-69 (FPN), :92-115 / :142-178 (heads), :272-287 (backbone),
 *   architectures/utils_resnet.py:25-41, :60-80 (blocks).
 * The reference has no FFI here (pure Python on torch): see INTEGRATION.md for the binding.
 *
 * One generic "gather conv" covers forward convolution (any kernel/stride/pad), data-gradient
 * (stride 1: flipped taps; stride 2: one problem per output-parity class) and the five pyramid
 * levels of the shared-weight heads as ONE grouped launch.
 */
#ifndef MI355DET_CONV_H_
#define MI355DET_CONV_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_CONV_MAX_PROBLEMS 5

/* One problem of a grouped launch.  Output row m = (n*P + p)*Q + q.
 * Tap (i,j), i < TR, j < TS reads input pixel (p*sh + dh0 + i*dhs, q*sw + dw0 + j*dws) (zeros
 * outside [0,H)x[0,W)) and weight tap (r0 + i*rs)*S + (s0 + j*ss). */
typedef struct Mi355ConvProblem {
    const void* in;      /* bf16 NHWC [N][H][W][C] */
    void* out;           /* bf16 or fp32, element offset of (n,p,q,k) =
                            n*out_sN + (p*out_mh+out_oh)*out_sH + (q*out_mw+out_ow)*out_sW + k */
    const void* addend;  /* optional bf16, element offset n*add_sN + (p>>add_shift)*add_sH +
                            (q>>add_shift)*add_sW + k ; added before the activation */
    const void* mask;    /* optional bf16, same indexing as out: result *= (mask > 0) */
    int32_t N, H, W, C;
    int32_t P, Q;
    int32_t sh, sw;
    int32_t TR, TS, dh0, dhs, dw0, dws;
    int32_t r0, rs, s0, ss, S;
    int32_t out_mh, out_oh, out_mw, out_ow;
    int64_t out_sN, out_sH, out_sW;
    int64_t add_sN, add_sH, add_sW;
    int32_t add_shift;
    int32_t reserved;
    int64_t in_sN;       /* batch stride of `in` in elements; 0 = dense (H*W*C) */
    const void* weight;  /* optional: THIS problem's weights (same [K][T][C] layout) and bias instead of the launch's:
                            the pyramid levels of an EfficientDet head conv share the conv but fold a BatchNorm per
                            level (efficientdet.py:335-353: bn_list[level][i]); NULL = the launch's weight / bias.
                            bf16 gather launches (split-K included: the finish pass applies the problem's bias). */
    const float* bias;   /* used with `weight` (may be NULL) */
} Mi355ConvProblem;

typedef struct Mi355ConvLaunch {
    Mi355ConvProblem prob[MI355_CONV_MAX_PROBLEMS];
    int32_t nprob;
    int32_t K;            /* output channels */
    const void* weight;   /* bf16 [K][T][C] (T = full tap count, C = prob.C), reduction-contiguous */
    int32_t T;            /* taps per weight row */
    int32_t act;          /* 0 none, 1 relu, 2 sigmoid */
    const float* bias;    /* optional fp32 [K] (conv bias or folded frozen-BN shift) */
    int32_t out_f32;      /* 0: bf16 out, 1: fp32 out */
    int32_t flags;        /* bit 0: apply the mask BEFORE the addend: act(acc*mask + addend) */
    float* stats;         /* optional (mi355_conv_gather, one problem, bf16 out, no addend, K % 8 == 0): per-tile column
                             statistics of the STORED output for a train-mode BatchNorm behind the conv
                             (utils/sync_batchnorm/batchnorm.py:63-93: sum and square-sum per channel), fp32
                             [ceil(M / 128)][2][K] = the partial-sum layout of mi355_bn_nhwc_stats with one block per
                             128-row tile, so mi355_bn_nhwc_reduce(stats, ceil(M / 128), K, ...) finishes them: the
                             statistics pass over z disappears */
} Mi355ConvLaunch;

/* out = act( gatherconv(in, weight) + bias + addend ) * (mask > 0) for every problem. */
int mi355_conv_gather(const Mi355ConvLaunch* launch, void* stream);
/* Same result for launches with few output tiles and a long reduction (P6 = conv3x3 s2 over C5,
 * retinanet.py:64; the 16x16 stage): the taps x channels reduction is cut into `ksplit` ranges that run
 * as separate blocks and store fp32 partial tiles into slab `split` of `ws` (ksplit * sum over problems
 * of N*P*Q*K floats, no initialisation needed); one pass then sums the slabs in index order
 * (bit-reproducible), applies bias / addend / activation / mask and writes `out`. */
int mi355_conv_gather_splitk(const Mi355ConvLaunch* launch, int ksplit, float* ws, void* stream);

/* Weight gradient of the same gather conv, grouped over problems that share the weight:
 *   dw[k][t][c] += sum_m dy[m][k] * in[m @ tap t][c]        (fp32 atomics; zero dw first)
 * Problem fields used: in, N,H,W,C, P,Q, sh,sw, TR,TS,dh0,dhs,dw0,dws, r0,rs,s0,ss,S;
 * `out` holds the bf16 dy of that problem, [N][P*Q][K] with batch stride out_sN elements
 * (0 = dense). dw is fp32 [K][T][C]. */
int mi355_conv_wgrad(const Mi355ConvLaunch* launch, float* dw, int split_rows, void* stream);
/* Same, and additionally colsum[k] += sum_m dy[m][k] (fp32 [K], may be NULL): the bias / frozen-BN
 * beta gradient, reduced from the dy fragments the kernel already holds (no extra HBM pass). */
int mi355_conv_wgrad_colsum(const Mi355ConvLaunch* launch, float* dw, float* colsum, int split_rows,
                            void* stream);

/* Deterministic split-K variant: every pixel chunk writes its partial dW tile with plain stores into
 * slab `chunk` of the workspace (ws + chunk*slab_stride floats; slab_stride >= K*T*C), to be summed in
 * a fixed order by mi355_slab_reduce.  Plain stores run ~4-5x the fp32-atomic rate and the result is
 * bit-reproducible.  max_slabs = capacity of ws; mi355_conv_wgrad_num_slabs gives the count. */
int mi355_conv_wgrad_num_slabs(const Mi355ConvLaunch* launch, int split_rows, int* nslabs);
int mi355_conv_wgrad_slabs(const Mi355ConvLaunch* launch, float* ws, int64_t slab_stride, int max_slabs,
                           float* colsum, int split_rows, void* stream);
/* Multi-LAYER form: the problems are the same-shape convs of DIFFERENT layers (the identical bottleneck blocks of a
 * ResNet stage: their weight gradients are independent leaves of the backward pass, and each alone is a 20-40 us
 * launch that cannot fill 256 CUs).  Problem i writes its pixel chunks into slabs 0 .. of its OWN workspace ws[i]
 * (capacity max_slabs[i]) and its dy column sums into colsum[i] (may be NULL); K, T, C and the conv geometry are
 * those of the launch, as above.  Slab mode only. */
int mi355_conv_wgrad_multi(const Mi355ConvLaunch* launch, float* const* ws, int64_t slab_stride,
                           const int* max_slabs, float* const* colsum, int split_rows, void* stream);
/* out[i] (+)= sum_s ws[s*slab_stride + i], i < n (n, slab_stride multiples of 4; 16-B aligned) */
int mi355_slab_reduce(const float* ws, int nslabs, int64_t slab_stride, int64_t n, float* out,
                      int accumulate, void* stream);

/* ---- elementwise / layout helpers of the conv path --------------------------------------- */
/* fp32 NCHW [N][C][H][W] -> bf16 NHWC [N][H][W][Cpad] (channels >= C zero). */
int mi355_nchw_to_nhwc_bf16(const float* in, void* out, int N, int C, int H, int W, int Cpad,
                            void* stream);
/* Space-to-depth form of the ResNet stem (7x7 / stride 2 / pad 3 over 3 channels -> 4x4 / stride 1 / pad (2, 1) over the
 * 2x2-blocked image; the reference's `self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)`,
 * retinanet.py:245): the blocked input [N][H/2][W/2][16] (channels (di, dj, c), 12 used), the derived weights
 * wp [K][12][4][4] = w[k][c][2a + di - 1][2b + dj - 1] (0 outside the 7x7 window) and the scatter of their gradient back:
 * dw[k][c][r][s] += dwp[k][(di, dj, c)][a][b].  w / dw: element strides (sk, sc, sr, ss). */
int mi355_nchw_to_nhwc_s2d_bf16(const float* in, void* out, int N, int H, int W, void* stream);
int mi355_stem_s2d_pack(const float* w, int64_t sk, int64_t sc, int64_t sr, int64_t ss, int K, float* wp, void* stream);
int mi355_stem_s2d_unpack_grad(const float* dwp, float* dw, int64_t sk, int64_t sc, int64_t sr, int64_t ss, int K,
                               void* stream);
/* Weight preparation: fp32 [K][C][R][S] with arbitrary element strides (sk,sc,sr,ss) ->
 *   w_fwd bf16 [K][R*S][Cpad]  = w * scale[k]          (forward operand)
 *   w_dgr bf16 [C][R*S][K]     = w * scale[k]          (dgrad operand), either may be NULL.
 * scale (fp32 [K], may be NULL = 1) is the folded frozen-BN gamma/sqrt(var+eps). */
int mi355_prep_weight(const float* w, int64_t sk, int64_t sc, int64_t sr, int64_t ss, int K, int C,
                      int R, int S, int Cpad, const float* scale, void* w_fwd, void* w_dgr,
                      void* stream);
/* Frozen-BN fold: scale = gamma / sqrt(var + eps), shift = beta - (mean - conv_bias) * scale
 * (fp32 [K]); conv_bias (may be NULL) is the bias of a conv that ALSO feeds the BN
 * (efficientdet.py:45,52: pointwise conv with bias, then BN). */
int mi355_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                  const float* conv_bias, float eps, int K, float* scale, float* shift, void* stream);
/* Column sums: out[k] (+)= sum_m g[m][k], g bf16 [M][K] (bias / BN-beta gradients). */
int mi355_colsum_bf16(const void* g, int64_t M, int K, float* out, void* stream);
/* Finish a conv layer's parameter gradients from the raw wgrad G = X^T dY (fp32 [K][T][Cpad]):
 *   dW[k][c][r][s] (strides sk,sc,sr,ss; fp32) (+)= G[k][t][c] * scale[k]       (c < C)
 *   if gamma given (frozen BN, retinanet.py:266-270 keeps gamma/beta trainable):
 *     dgamma[k] (+)= invstd[k] * ( sum_{t,c} W[k][c][r][s]*G[k][t][c] - mean[k]*colsum[k] )
 *     dbeta[k]  (+)= colsum[k]
 *   if the conv has its own bias b in front of the BN: mean -> mean - b in dgamma, and
 *     dconv_bias[k] (+)= scale[k] * colsum[k]
 * accumulate != 0 adds to the existing gradient buffers. */
int mi355_conv_param_grads(const float* G, const float* w, int64_t sk, int64_t sc, int64_t sr,
                           int64_t ss, int K, int C, int R, int S, int Cpad, const float* scale,
                           const float* colsum, const float* mean, const float* var, float eps,
                           float* dW, float* dgamma, float* dbeta, int accumulate,
                           const float* conv_bias, float* dconv_bias, void* stream);
/* ---- table-driven (one launch for ALL layers) weight prep and parameter-gradient finishing ----
 * The per-layer entry points above cost ~5 us of GPU time each for ~125 + ~55 tiny launches per
 * step; these two take a device-resident table of descriptors and cover every layer in one launch.
 * nblocks = sum of the per-layer block counts (prep: ceil(Cpad/32)*ceil(Kpad/32)*R*S; grads: K);
 * `block_begin` of each descriptor is the running sum.  The table is plain data: build it once. */
typedef struct Mi355LayerDesc {
    const float* w;          /* fp32 master weight [K][C][R][S], element strides sk,sc,sr,ss */
    int64_t sk, sc, sr, ss;
    int32_t K, C, R, S, Cpad, Kpad;
    int32_t block_begin;     /* first block of this layer in the launch */
    float eps;
    int32_t atomic;          /* grads: 1 = another descriptor writes the same dW / dbias (weights shared across pyramid
                                levels) and the additions are fp32 atomics; 2 = this descriptor is a non-head member of
                                a share_next chain: its dW / shared-bias contribution is added by the chain's head;
                                3 = the only writer of its dW / dgamma / dbeta / dbias in this step: plain stores, the
                                caller need not zero them (0: += into zeroed gradients) */
    int32_t nslabs;          /* grads: G / colsum hold this many split-K slabs (0 or 1: a single buffer) */
    int32_t f32_shadows;     /* prep: != 0 writes w_fwd / w_dgr as fp32 (the fp32-accurate mode) instead of bf16 */
    int32_t share_next;      /* grads: 1-based table index of the next descriptor that writes the same dW (same K, C,
                                R, S; its own G / slabs / BN), 0 = none.  The head of a chain adds its members in chain
                                order and stores once: bit-reproducible shared-weight gradients */
    const float* gamma;      /* frozen BN (all four NULL when the layer has no BN) */
    const float* beta;
    const float* mean;
    const float* var;
    const float* conv_bias;  /* conv bias in front of the BN, or NULL */
    float* scale;            /* prep out: fp32 [K] (NULL without BN) */
    float* shift;
    void* w_fwd;             /* prep out: bf16 [K][R*S][Cpad] */
    void* w_dgr;             /* prep out: bf16 [C][R*S][Kpad] or NULL */
    const float* G;          /* grads in: raw wgrad fp32 [Kpad][R*S][Cpad] */
    const float* colsum;     /* grads in: fp32 [Kpad] */
    float* dW;               /* grads out (+=): same strides as w */
    float* dgamma;           /* (+=) or NULL */
    float* dbeta;            /* (+=): BN beta, or the conv bias of a BN-less layer, or NULL */
    float* dconv_bias;       /* (+=) or NULL */
    int64_t slab_stride;     /* grads: floats between consecutive G slabs (colsum slabs are Kpad apart);
                                the slabs are summed in index order: bit-reproducible gradients */
} Mi355LayerDesc;
int mi355_prep_weights_table(const Mi355LayerDesc* d_table, int nlayers, int nblocks, void* stream);
int mi355_param_grads_table(const Mi355LayerDesc* d_table, int nlayers, int nblocks, void* stream);

/* ---- fp32-accurate mode (MI355_CONV_PRECISION=fp32) ------------------------------------------------
 * The reference trains in fp32 (train.py:259-321: no autocast).  These entry points take the SAME launch
 * descriptors as the bf16 ones above, with every activation / gradient / weight-shadow pointer read as
 * fp32 (strides stay in elements, C a multiple of 4) and run the contraction on the exact-f32 matrix
 * instruction v_mfma_f32_16x16x4_f32.  Output is always fp32 (`out_f32` is ignored). */
int mi355_conv_gather_f32(const Mi355ConvLaunch* launch, void* stream);
/* slab_stride == 0: fp32 atomics straight into `ws` (= dW; zero it first), colsum likewise. */
int mi355_conv_wgrad_slabs_f32(const Mi355ConvLaunch* launch, float* ws, int64_t slab_stride, int max_slabs,
                               float* colsum, int split_rows, void* stream);
int mi355_nchw_to_nhwc_f32(const float* in, float* out, int N, int C, int H, int W, int Cpad, void* stream);
int mi355_maxpool3x3s2_fwd_f32(const float* in, float* out, int8_t* idx, int N, int H, int W, int C, void* stream);
int mi355_maxpool3x3s2_bwd_f32(const float* gout, const int8_t* idx, float* gin, const float* relu_mask,
                               int N, int H, int W, int C, void* stream);
int mi355_maxpool3x3s2_bwd_pooled_f32(const float* gout, const int8_t* idx, float* gin, const float* pooled,
                                      int N, int H, int W, int C, void* stream);
int mi355_sumpool2x2_f32(const float* src, float* dst, int N, int H, int W, int C, int accumulate, void* stream);
int mi355_relu_mask_f32(const float* x, const float* ref, float* y, int64_t n, void* stream);

/* MaxPool 3x3 s2 p1 (retinanet.py:188) NHWC bf16; idx int8 [N][P][Q][C] = argmax window slot. */
int mi355_maxpool3x3s2_fwd(const void* in, void* out, int8_t* idx, int N, int H, int W, int C,
                           void* stream);
int mi355_maxpool3x3s2_bwd(const void* gout, const int8_t* idx, void* gin, const void* relu_mask,
                           int N, int H, int W, int C, void* stream);
/* The same behind a ReLU (retinanet.py:275-276: relu, then maxpool), with the mask taken from the POOLED map
 * [N][P][Q][C] instead of the pool's input: the argmax cell of a window is positive exactly when the pooled value
 * is (a window of zeros hands its gradient to a cell whose ReLU gradient is zero).  A quarter of the mask bytes. */
int mi355_maxpool3x3s2_bwd_pooled(const void* gout, const int8_t* idx, void* gin, const void* pooled,
                                  int N, int H, int W, int C, void* stream);
/* Nearest-2x upsample backward: dst[n][p][q][c] (+)= sum of the 2x2 children of src
 * (src [N][H][W][C], dst [N][ceil(H/2)][ceil(W/2)][C], bf16). */
int mi355_sumpool2x2_bf16(const void* src, void* dst, int N, int H, int W, int C, int accumulate,
                          void* stream);
/* y = x * (ref > 0) elementwise, bf16 (ReLU backward with the saved output). n elements. */
int mi355_relu_mask_bf16(const void* x, const void* ref, void* y, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355DET_CONV_H_ */
