/* mi355det_bn.h — C ABI of train-mode / cross-replica BatchNorm inside the fused launch plans (NHWC bf16 or fp32).
 *
 * The reference's training loop freezes BatchNorm (train.py:273-277 -> `freeze_bn`, architectures/retinanet.py:266-270)
 * and the launch plans fold such layers into their convs.  A model left in train mode, or passed through
 * `replace_w_sync_bn` (utils/utils.py:140-168; BASELINE.json configs[2] "DP + SyncBN"), needs batch statistics per
 * layer.  These passes replace the tensor arithmetic of
 *     utils/sync_batchnorm/batchnorm.py:63-93  (SynchronizedBatchNorm2d.forward: sum, square-sum, normalise)
 *     utils/sync_batchnorm/batchnorm.py:128-145 (_compute_mean_std: mean, clamp(biased var, eps)^-1/2, running stats)
 *     torch.nn.functional.batch_norm(training=True) for one replica (batchnorm.py:65-69)
 * and of their autograd backward.  The cross-replica exchange itself ([sum | square-sum] forward,
 * [sum g | sum g*zhat] backward: `sums`, fp32 [2][C]) is one torch.distributed all-reduce issued by the plan
 * between the reduce and the finalize / backward-apply pass.  Conventions as mi355det.h; every reduction goes
 * through per-block partial sums added in index order (bit-reproducible).
 *
 * Layer state (fp32 [5][C], written by mi355_bn_nhwc_finalize): mean | invstd | scale = gamma*invstd |
 * shift = beta - mean*scale | through (0 where the synchronised form's clamp(var, eps) is active).
 */
#ifndef MI355DET_BN_H_
#define MI355DET_BN_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MI355_BN_STATE_ROWS 5

/* Partial-sum geometry for M rows: nblk blocks of rows_per_block rows; partials = fp32 [nblk][2][C]. */
int mi355_bn_nhwc_partial_blocks(int64_t M, int C, int* nblk, int* rows_per_block);
/* bwd == 0: partials[b] = (sum a, sum a*a) over block b's rows of a [M][C].
 * bwd != 0: partials[b] = (sum a, sum a * (z - mean) * invstd) with a = dL/dy, b = z, mean / invstd from `state`.
 * f32 != 0: a, b are fp32; else bf16.  C % 8 == 0. */
int mi355_bn_nhwc_stats(const void* a, const void* b, const float* state, int f32, int64_t M, int C, int bwd,
                        float* partials, void* stream);
/* sums[2][C] = sum over the nblk partial blocks (double accumulation, index order).
 * dgamma (+)= sums[1], dbeta (+)= sums[0] when given (backward: the LOCAL sums are the parameter gradients). */
int mi355_bn_nhwc_reduce(const float* partials, int nblk, int C, float* sums, float* dgamma, float* dbeta,
                         void* stream);
/* As mi355_bn_nhwc_reduce, and sums[2C], sums[2C + 1] = this replica's element count per channel (`local_count` = its
 * N*H*W) split as floor(count / 4096), count mod 4096 - both exact in fp32 and exact under the SUM all-reduce of up to
 * 4096 replicas.  The reference's SynchronizedBatchNorm2d adds the ACTUAL sizes of the replicas
 * (utils/sync_batchnorm/batchnorm.py:105-126 `sum_size`), which nn.DataParallel's scatter makes unequal when the batch
 * does not divide; sums has 2C + 2 elements and all of them go through the exchange. */
int mi355_bn_nhwc_reduce_counted(const float* partials, int nblk, int C, int64_t local_count, float* sums, float* dgamma,
                                 float* dbeta, void* stream);
/* sums (all-reduced over the replicas when synchronised) + element count -> layer state; running statistics
 * (may be NULL) updated with `momentum` and the UNBIASED variance; *num_batches_tracked += 1 when given.
 * sync_clamp != 0: invstd = clamp(biased var, eps)^-1/2 (batchnorm.py:145); 0: (biased var + eps)^-1/2. */
/* count <= 0: the count is read from sums[2C .. 2C + 2) (mi355_bn_nhwc_reduce_counted, after the exchange) and
 * state[5C] = 1 / count is stored for the backward pass (state then has 5C + 1 elements). */
int mi355_bn_nhwc_finalize(const float* sums, float count, int C, float eps, float momentum, int sync_clamp,
                           const float* gamma, const float* beta, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, float* state, void* stream);
/* y = act(z*scale + shift (+ addend)), act 0 none / 1 relu; z, addend, y: [M][C] bf16 or fp32. */
int mi355_bn_nhwc_apply(const void* z, const float* state, const void* addend, int act, int f32, int64_t M, int C,
                        void* y, void* stream);
/* dz = scale * (g - (sums[0] + zhat * through * sums[1]) * inv_count), zhat = (z - mean)*invstd; sums = the
 * (all-reduced) backward sums, inv_count = 1 / total element count per channel over the replicas. */
/* inv_count <= 0: read from state[5C] (the counted form of mi355_bn_nhwc_finalize). */
int mi355_bn_nhwc_bwd_apply(const void* g, const void* z, const float* state, const float* sums, float inv_count,
                            int f32, int64_t M, int C, void* dz, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355DET_BN_H_ */
