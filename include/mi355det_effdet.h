/* mi355det — C ABI of the EfficientDet-specific HBM-bound kernels (gfx950).
 *
 * Depthwise / squeeze-excite / BiFPN pieces of architectures/efficientnet.py:80-115 (MBConvBlock),
 * architectures/efficientdet.py:28-66 (SeparableConvBlock), :195-267 (BiFPN fast attention),
 * architectures/utils_extra_efficientnet.py:9-96 (TF-"same" padding, zero-padded max-pool),
 * architectures/utils_efficientnet.py:38-49 (swish forward / backward).  The 1x1 ("pointwise")
 * convolutions run on the MFMA gather conv of mi355det_conv.h.
 *
 * Tensors are NHWC bf16 with C a multiple of 8.  A tensor flagged `in_swish` is stored as the
 * PRE-activation z; its consumer applies swish(z) = z*sigmoid(z) while loading (the reference's
 * MemoryEfficientSwish also keeps only z), and the matching backward multiplies by
 * swish'(z) = s(z)*(1 + z*(1 - s(z))).
 */
#ifndef MI355DET_EFFDET_H_
#define MI355DET_EFFDET_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* depthwise weights: fp32 [C][1][k][k] (contiguous) * scale[c] -> fp32 [k*k][C] (tap-major) */
int mi355_dw_prep_weight(const float* w, const float* scale, int C, int k, float* w_t, void* stream);
/* out[n,p,q,c] = bias[c] + sum_taps f(in[n, p*s - pad_t + r, q*s - pad_l + t, c]) * w_t[r*k+t][c],
 * f = swish if in_swish else identity; zero padding (TF "same": pad_t/pad_l = extra/2). */
int mi355_dwconv_fwd(const void* in, int in_swish, const float* w_t, const float* bias, void* out,
                     int N, int H, int W, int C, int k, int stride, int pad_t, int pad_l, int P, int Q,
                     void* stream);
/* The forward with the squeeze-excite pooling of its output (efficientnet.py:94-100: x = swish(bn1(depthwise_conv(x)));
 * x_squeezed = adaptive_avg_pool2d(x, 1)) in the same launch: pool_partial [N][slabs][C] = per-tile partial means of
 * swish(out), to be added in order by mi355_se_fc (slabs = mi355_dwconv_fwd_sepool_slabs; 0: this geometry is not
 * served by the LDS-tiled kernel, use mi355_dwconv_fwd + mi355_se_pool). */
int mi355_dwconv_fwd_sepool_slabs(int H, int W, int k, int stride, int P, int Q, int* slabs);
int mi355_dwconv_fwd_sepool(const void* in, int in_swish, const float* w_t, const float* bias, void* out,
                            float* pool_partial, int N, int H, int W, int C, int k, int stride, int pad_t, int pad_l,
                            int P, int Q, void* stream);
/* dx = conv-transpose(dy) [* swish'(zin) if zin] [+ dx if accumulate] */
int mi355_dwconv_dgrad(const void* dy, const float* w_t, void* dx, const void* zin, int N, int H, int W,
                       int C, int k, int stride, int pad_t, int pad_l, int P, int Q, int accumulate,
                       void* stream);
/* mi355_se_bwd_apply followed by mi355_dwconv_dgrad in ONE launch (stride 1, "same", k = 3 / 5, maps >= 8 x 16): gz (stored:
 * the weight gradient reads it) is built while the data gradient's halo tile is staged; bit-identical to the two launches.
 * (efficientnet.py:94-104 backward: the gradient of swish(bn1(depthwise_conv(x))) * sigmoid(se(...)) w.r.t. the conv output) */
int mi355_dwconv_dgrad_seapply(const void* gu, const void* z, const float* gate, const float* gpool, void* gz,
                               const float* w_t, void* dx, const void* zin, int N, int H, int W, int C, int k,
                               int accumulate, void* stream);
/* dw_t[t][c] += sum_m f(x)[m@t][c] * dy[m][c];  colsum[c] += sum_m dy[m][c] (may be NULL) */
int mi355_dwconv_wgrad(const void* x, int in_swish, const void* dy, float* dw_t, float* colsum, int N,
                       int H, int W, int C, int k, int stride, int pad_t, int pad_l, int P, int Q,
                       void* stream);
/* The same sums, bit-reproducible (k = 3 / 5, stride 1 / 2, bf16 tensors): the launch stores `nslabs` partial rows
 * slabs[s][(k*k + 1) * C] = { dw_t partial [k*k][C], colsum partial [C] (written when want_colsum) } with plain
 * stores, nothing to zero; mi355_dw_param_grads_table adds the rows in index order (Mi355DwGradDesc.nslabs).
 * mi355_dwconv_wgrad_nslabs returns the row count the launch of that geometry writes.
 * (reference: the depthwise weight gradient of efficientnet.py:80-115 / utils_extra_efficientnet.py:9-47 as autograd
 * computes it; a shared head conv gives every pyramid level its own rows of one workspace) */
int mi355_dwconv_wgrad_nslabs(int N, int C, int k, int stride, int P, int Q, int* nslabs);
/* ... and for up to MI355_DW_MAX_LEVELS feature maps that share the depthwise weight (the five pyramid levels of an
 * EfficientDet head conv, efficientdet.py:269-359: `for feat, bn_list in zip(inputs, self.bn_list)` applies the same
 * conv_list[i] to every level) in ONE launch: the rows hold the sums over all the levels' pixels. */
#define MI355_DW_MAX_LEVELS 5
typedef struct Mi355DwLevel {
    const void* x; const void* dy;              /* bf16 NHWC [N][H][W][C], [N][P][Q][C] */
    const void* aux;                            /* mi355_dwconv_dgrad_multi: zin (may be NULL); otherwise unused */
    int32_t H, W, P, Q, pad_t, pad_l;
} Mi355DwLevel;
int mi355_dwconv_wgrad_nslabs_multi(const Mi355DwLevel* levels, int nlevels, int N, int C, int k, int stride,
                                    int* nslabs);
/* Forward and data gradient of a stride-1 "same" 3x3 / 5x5 depthwise conv over up to MI355_DW_MAX_LEVELS maps in one
 * launch (same arithmetic as mi355_dwconv_fwd / mi355_dwconv_dgrad per map).  forward: x = input, dy = output;
 * data gradient: x = dy, dy = dx, aux = zin (dx *= swish'(zin) when non-NULL), accumulate as mi355_dwconv_dgrad. */
int mi355_dwconv_fwd_multi(const Mi355DwLevel* levels, int nlevels, int in_swish, const float* w_t, const float* bias,
                           int N, int C, int k, void* stream);
int mi355_dwconv_dgrad_multi(const Mi355DwLevel* levels, int nlevels, const float* w_t, int N, int C, int k,
                             int accumulate, void* stream);
int mi355_dwconv_wgrad_slabs_multi(const Mi355DwLevel* levels, int nlevels, int in_swish, float* slabs,
                                   int want_colsum, int N, int C, int k, int stride, void* stream);
int mi355_dwconv_wgrad_slabs(const void* x, int in_swish, const void* dy, float* slabs, int want_colsum, int N,
                             int H, int W, int C, int k, int stride, int pad_t, int pad_l, int P, int Q,
                             void* stream);
/* a = swish(z) elementwise (n elements, multiple of 8): materialised input for dwconv_wgrad */
int mi355_swish_bf16(const void* z, void* a, int64_t n, void* stream);
/* finish depthwise parameter grads: dW[c][k][k] (+)= G_t[t][c]*scale[c]; frozen-BN dgamma/dbeta
 * (gamma may be NULL: plain conv). */
int mi355_dw_param_grads(const float* G_t, const float* w, const float* scale, const float* colsum,
                         const float* mean, const float* var, float eps, int C, int k, float* dW,
                         float* dgamma, float* dbeta, int accumulate, void* stream);

/* Start-of-step preparation of every depthwise layer in one launch: frozen-BN fold (scale = gamma / sqrt(var + eps),
 * shift = beta - mean * scale; gamma NULL: plain conv, scale / shift untouched) and w_t[t][c] = w[c][t] * scale[c]
 * (= mi355_bn_fold + mi355_dw_prep_weight per row).  max_elems = the largest C * T in the table. */
typedef struct Mi355DwPrepDesc {
    const float* gamma; const float* beta; const float* mean; const float* var; const float* w;
    float* scale; float* shift; float* w_t;
    float eps; int32_t C; int32_t T; int32_t pad_;
} Mi355DwPrepDesc;
int mi355_dw_prep_table(const Mi355DwPrepDesc* table, int n, int max_elems, void* stream);
/* Fast-attention fusion weights of every BiFPN node in one launch: w[0..2] = relu(p) / (sum relu(p) + eps)
 * (efficientdet.py:213-265), rows of (parameter, its length 2 or 3, output). */
typedef struct Mi355FusionDesc { const float* p; float* w; int32_t n; int32_t pad_; } Mi355FusionDesc;
int mi355_fusion_weights_table(const Mi355FusionDesc* table, int rows, float eps, void* stream);

/* ... and their backward (gp = d loss / d p from gw = d loss / d w) for all nodes in one launch. */
typedef struct Mi355FusionBwdDesc { const float* p; const float* gw; float* gp; int32_t n;
    int32_t nparts;  /* 0: gw = the n sums; >= 1: gw = nparts partial rows of 4 floats (mi355_bifpn_node_bwd) */
} Mi355FusionBwdDesc;
int mi355_fusion_weights_bwd_table(const Mi355FusionBwdDesc* table, int rows, float eps, void* stream);

/* The same for every depthwise layer of a plan in one launch: a device-resident table of `n` rows (borrowed device
 * pointers as above; dgamma / dbeta / scale / colsum / mean / var may be NULL), max_c = the largest C in it. */
typedef struct Mi355DwGradDesc {
    const float* G_t; const float* w; const float* scale; const float* colsum; const float* mean; const float* var;
    float* dW; float* dgamma; float* dbeta;
    float eps; int32_t C; int32_t T; int32_t accumulate;
    int32_t nslabs; int32_t slab_stride;   /* G_t / colsum: nslabs partial rows, slab_stride floats apart (T <= 25) */
} Mi355DwGradDesc;
int mi355_dw_param_grads_table(const Mi355DwGradDesc* table, int n, int max_c, void* stream);

/* squeeze-excite (efficientnet.py:98-104) */
/* The H*W reduction is split into `slabs` row slabs per sample (mi355_se_slabs); the reduce kernels
 * store per-slab partials [N][slabs][C] (no atomics, nothing to zero) and the tiny per-sample FC
 * kernels add them in a fixed order: bit-reproducible forward and backward. */
int mi355_se_slabs(int N, int HW, int* slabs);
int mi355_se_pool(const void* z, int in_swish, float* pool_partial /* [N][slabs][C] */, int N, int HW, int C,
                  void* stream);
int mi355_se_fc(const float* pool_partial, int slabs, float* pool /* [N][C] out */, const float* w1,
                const float* b1, const float* w2, const float* b2, float* rpre /* [N][Cse] */,
                float* gate /* [N][C] */, int N, int C, int Cse, void* stream);
int mi355_se_scale(const void* z, const float* gate, void* u, int N, int HW, int C, void* stream);
int mi355_se_bwd_reduce(const void* gu, const void* z, float* ggate_partial /* [N][slabs][C] */, int N, int HW,
                        int C, void* stream);
int mi355_se_fc_bwd(const float* pool, const float* w1, const float* w2, const float* rpre,
                    const float* gate, const float* ggate_partial, int slabs, float* gpool, float* dw1,
                    float* db1, float* dw2, float* db2, int N, int C, int Cse, void* stream);
int mi355_se_bwd_apply(const void* gu, const void* z, const float* gate, const float* gpool, void* gz,
                       int N, int HW, int C, void* stream);
/* mi355_se_fc_bwd without atomics, in two parts: the per-sample data path (de [N][C], dr [N][Cse] kept, gpool [N][C]) ... */
int mi355_se_fc_bwd_data(const float* w1, const float* w2, const float* rpre, const float* gate,
                         const float* ggate_partial, int slabs, float* de, float* dr, float* gpool, int N, int C,
                         int Cse, void* stream);
/* ... and the parameter gradients of every squeeze-excite layer of a plan in one launch (plain stores, sums over the
 * N samples in index order): dw2[c][j] = sum_n de[n][c] swish(rpre[n][j]), db2 = sum_n de, dw1[j][c] = sum_n dr[n][j]
 * pool[n][c], db1 = sum_n dr.  max_elems = the largest 2 C Cse + C + Cse in the table. */
typedef struct Mi355SeGradDesc {
    const float* pool; const float* rpre; const float* de; const float* dr;
    float* dw1; float* db1; float* dw2; float* db2;
    int32_t N, C, Cse, pad_;
} Mi355SeGradDesc;
int mi355_se_param_grads_table(const Mi355SeGradDesc* table, int n, int max_elems, void* stream);

/* BiFPN fast-attention node (efficientdet.py:213-265): w = relu(p) / (sum relu(p) + eps) */
int mi355_fusion_weights_fwd(const float* p, int n, float eps, float* w, void* stream);
int mi355_fusion_weights_bwd(const float* p, int n, float eps, const float* gw, float* gp, void* stream);
/* z = w[0]*in0 + w[1]*in1' + w[2]*in2; in1' = nearest-2x upsample of in1 when up1 != 0
 * (in1 is then [N][H/2][W/2][C]); in2 may be NULL.  w: device float[3]. */
int mi355_bifpn_node_fwd(const void* in0, const void* in1, int up1, const void* in2, const float* w,
                         void* z, int N, int H, int W, int C, void* stream);
/* gw[i] += <gz, in_i'> (device float[3], zeroed) */
int mi355_bifpn_node_wgrad(const void* gz, const void* in0, const void* in1, int up1, const void* in2,
                           float* gw, int N, int H, int W, int C, void* stream);
/* dst (+)= w[wi] * src  (same shape, n elements) */
int mi355_axpy_bf16(const void* src, const float* w, int wi, void* dst, int64_t n, int accumulate,
                    void* stream);
/* Drop-connect of an MBConv skip branch (utils_efficientnet.py:85-94, efficientnet.py:109-113):
 * dst[n] = src[n] / keep * floor(keep + u[n]) (+ addend[n]); u: device float[N] uniforms in [0, 1), one per image;
 * per_image = H*W*C elements.  Backward of the branch = the same call on the gradient with addend NULL. */
int mi355_drop_connect(const void* src, const float* u, float keep, const void* addend, void* dst, int N,
                       int64_t per_image, void* stream);
/* its backward in one pass over dy: g_branch = dy / keep * floor(keep + u[n]); g_skip = dy (+ g_skip when acc_skip) */
int mi355_drop_connect_bwd(const void* dy, const float* u, float keep, void* g_branch, void* g_skip, int acc_skip, int N,
                           int64_t per_image, void* stream);
int mi355_drop_connect_bwd_f32(const void* dy, const float* u, float keep, void* g_branch, void* g_skip, int acc_skip,
                               int N, int64_t per_image, void* stream);
/* mi355_bifpn_node_fwd + mi355_dwconv_fwd(in_swish = 1, stride 1, "same") in ONE launch: z (stored: the backward
 * reads it) is built while the depthwise conv's halo tile is staged; bit-identical to the two launches.
 * (efficientdet.py:213-265: `self.conv6_up(self.swish(weight[0] * p6_in + weight[1] * self.p6_upsample(p7_in)))` ...) */
int mi355_bifpn_node_dwconv_fwd(const void* in0, const void* in1, int up1, const void* in2, const float* w, void* z,
                                const float* w_t, const float* bias, void* out, int N, int H, int W, int C, int k,
                                void* stream);
/* The whole backward of a node in one pass: g_i = w[i] * gz (+ g_i when acc_i) for in0, in1 (2x2 sums of gz when up1:
 * g1 is [N][H/2][W/2][C]) and in2 (may be NULL); when gw is non-NULL every block stores its partial <gz, in_i> sums
 * into its own row gw[block][4] (mi355_bifpn_node_bwd_parts rows, plain stores, nothing to zero;
 * mi355_fusion_weights_bwd_table adds them in order).  (efficientdet.py:213-265 as autograd differentiates it.) */
int mi355_bifpn_node_bwd_parts(int N, int H, int W, int C, int up1, int* nparts);
int mi355_bifpn_node_bwd(const void* gz, const void* in0, const void* in1, int up1, const void* in2, const float* w,
                         float* gw, void* g0, int acc0, void* g1, int acc1, void* g2, int acc2, int N, int H, int W, int C,
                         void* stream);
int mi355_bifpn_node_bwd_f32(const void* gz, const void* in0, const void* in1, int up1, const void* in2, const float* w,
                             float* gw, void* g0, int acc0, void* g1, int acc1, void* g2, int acc2, int N, int H, int W,
                             int C, void* stream);
/* dst[N][H/2][W/2][C] (+)= w[wi] * (sum of the 2x2 children of src[N][H][W][C]) */
int mi355_upsample_bwd_axpy(const void* src, const float* w, int wi, void* dst, int N, int H, int W, int C,
                            int accumulate, void* stream);
/* TF-"same" max-pool k=3 s=2 with ZERO padding taking part in the max
 * (utils_extra_efficientnet.py:78-96); idx int8 = winning window slot, -1 = a pad won. */
int mi355_maxpool_same_fwd(const void* in, void* out, int8_t* idx, int N, int H, int W, int C, void* stream);
/* gin (+)= scatter(gout through idx) */
int mi355_maxpool_same_bwd(const void* gout, const int8_t* idx, void* gin, int N, int H, int W, int C,
                           int accumulate, void* stream);

/* ---- fp32-accurate mode (MI355_CONV_PRECISION=fp32 / model.precision = 'fp32') ----------------------------------
 * The same entry points with every activation / gradient tensor fp32 (the reference trains in fp32, train.py:259-321):
 * identical arguments and arithmetic, exact expf / division swish instead of the v_exp / v_rcp forms that are only
 * accurate to the bf16 rounding.  The 1x1 convs of this mode run on mi355_conv_gather_f32 (mi355det_conv.h). */
int mi355_dwconv_fwd_f32(const void* in, int in_swish, const float* w_t, const float* bias, void* out, int N, int H,
                         int W, int C, int k, int stride, int pad_t, int pad_l, int P, int Q, void* stream);
int mi355_dwconv_dgrad_f32(const void* dy, const float* w_t, void* dx, const void* zin, int N, int H, int W, int C,
                           int k, int stride, int pad_t, int pad_l, int P, int Q, int accumulate, void* stream);
int mi355_dwconv_wgrad_f32(const void* x, int in_swish, const void* dy, float* dw_t, float* colsum, int N, int H, int W,
                           int C, int k, int stride, int pad_t, int pad_l, int P, int Q, void* stream);
int mi355_se_pool_f32(const void* z, int in_swish, float* pool_partial, int N, int HW, int C, void* stream);
int mi355_se_scale_f32(const void* z, const float* gate, void* u, int N, int HW, int C, void* stream);
int mi355_se_bwd_reduce_f32(const void* gu, const void* z, float* ggate_partial, int N, int HW, int C, void* stream);
int mi355_se_bwd_apply_f32(const void* gu, const void* z, const float* gate, const float* gpool, void* gz, int N, int HW,
                           int C, void* stream);
int mi355_bifpn_node_fwd_f32(const void* in0, const void* in1, int up1, const void* in2, const float* w, void* z, int N,
                             int H, int W, int C, void* stream);
int mi355_bifpn_node_wgrad_f32(const void* gz, const void* in0, const void* in1, int up1, const void* in2, float* gw,
                               int N, int H, int W, int C, void* stream);
int mi355_axpy_f32(const void* src, const float* w, int wi, void* dst, int64_t n, int accumulate, void* stream);
int mi355_drop_connect_f32(const void* src, const float* u, float keep, const void* addend, void* dst, int N,
                           int64_t per_image, void* stream);
int mi355_upsample_bwd_axpy_f32(const void* src, const float* w, int wi, void* dst, int N, int H, int W, int C,
                                int accumulate, void* stream);
int mi355_maxpool_same_fwd_f32(const void* in, void* out, int8_t* idx, int N, int H, int W, int C, void* stream);
int mi355_maxpool_same_bwd_f32(const void* gout, const int8_t* idx, void* gin, int N, int H, int W, int C,
                               int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355DET_EFFDET_H_ */
