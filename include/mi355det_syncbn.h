/* mi355det_syncbn.h — C ABI of the train-mode / cross-replica BatchNorm building blocks (NCHW fp32).
 *
 * Replaces the tensor arithmetic of the reference's SynchronizedBatchNorm2d
 * (utils/sync_batchnorm/batchnorm.py:63-93 forward, :128-145 statistics) and its autograd backward; the
 * cross-replica exchange itself (one all-reduce of [sum | sumsq | count] forward, [sum dy | sum dy*xhat]
 * backward) stays in torch.distributed (objectdetection/pytorch_amd/dp.py).  Conventions as mi355det.h. */
#ifndef MI355DET_SYNCBN_H_
#define MI355DET_SYNCBN_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Number of partial sums per channel the reduction wants for this shape (>= 1). */
int mi355_bn_reduce_splits(int N, int C, int HW, int* splits);
/* part[c][s][0..1], s < splits:  mode 0: (sum p, sum p*p)   mode 1: (sum p, sum p * (q - mean[c]) * invstd[c])
 * over the s-th slice of the (n, hw) positions of channel c.  p, q: NCHW fp32.  Deterministic. */
int mi355_bn_reduce_nchw(const float* p, const float* q, const float* mean, const float* invstd,
                         int N, int C, int HW, int mode, int splits, float* part, void* stream);
/* out = a[c]*p + b[c]*q + c0[c]  (b, q may be NULL): normalisation forward (a = invstd*w, c0 = bias - mean*a)
 * and backward (a = w*invstd, b = -w*invstd^2*mean(dy*xhat), c0 = a*(mean*invstd*mean(dy*xhat) - mean(dy))). */
int mi355_bn_affine_nchw(const float* p, const float* q, const float* a, const float* b, const float* c0,
                         float* out, int N, int C, int HW, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355DET_SYNCBN_H_ */
