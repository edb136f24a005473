/* mi355det — C ABI of the flat-arena optimizer step (gfx950, HBM-bound streaming kernels).
 *
 * Replaces train.py:320-321 of the reference:
 *     torch.nn.utils.clip_grad_norm_(net_model.parameters(), 0.1)
 *     optimizer.step()                       # torch.optim.AdamW(lr=1e-4, weight_decay=5e-4), :259-260
 * and the skip of train.py:316-317 (`if bool(loss == 0): continue`) without a host sync.
 * All parameters / gradients / moments live in flat fp32 arenas (one RCCL all-reduce, one pass).
 */
#ifndef MI355DET_OPTIM_H_
#define MI355DET_OPTIM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Device-resident optimizer state: {int32 step, float clip_coef, float skip, float grad_norm}. */
#define MI355_OPT_STATE_BYTES 16
/* number of fp32 partial sums the norm pass writes (workspace = MI355_OPT_PARTIALS * 4 bytes) */
#define MI355_OPT_PARTIALS 1024

/* Pass 1: partials[b] = sum of squares of a strided slice of grad[n]. */
int mi355_grad_sumsq(const float* grad, int64_t n, float* partials, void* stream);
/* Pass 2 (one block): total L2 norm -> clip_coef = min(1, max_norm / (norm + 1e-6))
 * (clip_grad_norm_ semantics); skip = 1 when losses[0] + losses[1] == 0 (losses may be NULL);
 * step += 1 unless skipped. */
int mi355_adamw_prepare(const float* partials, const float* losses, float max_norm, void* state,
                        void* stream);
/* Pass 3: AdamW update over the flat arenas with the clip folded in (g = grad * clip_coef):
 *   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
 *   p -= (lr / (1-b1^step)) * m / (sqrt(v) / sqrt(1-b2^step) + eps)
 * (torch.optim.AdamW single-tensor formulas).  grad is scaled in place when scale_grad != 0 so
 * that `.grad` holds the clipped gradient as after clip_grad_norm_. */
int mi355_adamw_flat(float* param, float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, const void* state, int scale_grad,
                     void* stream);

/* The same two passes with a skip table: `skip_ranges` = device int64 [nskip][2], sorted disjoint arena ranges
 * [lo, hi) of elements that neither count in the norm nor are updated (no weight decay either).  These are
 * the parameters torch's clip_grad_norm_ / AdamW skip because `p.grad is None`: requires_grad == False
 * (the head-only fine-tune of train.py:206-216) and parameters no forward uses (the BiFPN fusion weights of
 * EfficientDet-D6/D7 with attention=False, efficientdet.py:269-317).  nskip == 0: the plain passes. */
int mi355_grad_sumsq_masked(const float* grad, int64_t n, float* partials, const int64_t* skip_ranges, int nskip,
                            void* stream);
int mi355_adamw_flat_masked(float* param, float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, const void* state, int scale_grad,
                            const int64_t* skip_ranges, int nskip, void* stream);

/* The masked update with the hyper-parameters read from DEVICE memory: hyper = float[5] {lr, beta1, beta2, eps,
 * weight_decay}.  For a launch captured in a hipGraph: by-value kernel arguments are frozen at capture, so a
 * ReduceLROnPlateau step (train.py:268, :346) between replays would be ignored; the host refreshes this vector. */
int mi355_adamw_flat_hyper(float* param, float* grad, float* m, float* v, int64_t n, const float* hyper,
                           const void* state, int scale_grad, const int64_t* skip_ranges, int nskip, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355DET_OPTIM_H_ */
