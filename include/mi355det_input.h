/* mi355det_input.h — C ABI of the input side of the training path: the per-sample
 * `load_image` scaling, `Normalizer`, the zero canvas of `Resizer` and the batch padding + NHWC->NCHW
 * permute of `collater`, done in one launch on the device from the decoded (uint8) images, so that the
 * host ships 3 bytes per pixel instead of a 12-byte fp32 NCHW batch.
 *
 * Replaces (reference file:line):
 *   datasettool/dataloader.py:92          img.astype(np.float32) / 255.0
 *   datasettool/dataloader.py:905-924     Normalizer: (image.astype(np.float32) - mean) / std  (float64 maths)
 *   datasettool/dataloader.py:872-880     Resizer mode 1: zero canvas, image at the top-left corner, fp32
 *   datasettool/dataloader.py:886         annots[:, :4] *= scale
 *   datasettool/dataloader.py:655-691     collater: zero-pad to the batch maximum, permute(0,3,1,2); annotations
 *                                         padded with -1 rows ([B,1,5] of -1 when no image has a box)
 * The image resampling itself (skimage.transform.resize, :866) stays with the caller.
 * Conventions as mi355det.h: borrowed device pointers, 0 / negative error code, no allocation, no sync. */
#ifndef MI355DET_INPUT_H_
#define MI355DET_INPUT_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MI355_IN_U8        1   /* packed images are uint8 HWC (else float32 HWC)                       */
#define MI355_IN_SCALE255  2   /* v = float32(v) / 255.0f first                 (dataloader.py:92)      */
#define MI355_IN_NORMALIZE 4   /* v = float32((double(v) - mean[c]) / std[c])   (dataloader.py:905-924) */

/* packed: the B images back to back, each HWC with 3 channels and no row padding.
 * meta[b] = {byte offset of image b in `packed`, h_b, w_b} (int64 x 3, device memory).
 * out: [B,3,H,W] fp32, every element written: the image at the top-left corner, zeros elsewhere. */
int mi355_collate_images(const void* packed, const int64_t* meta, int B, int H, int W, int flags,
                         const double* mean3, const double* std3, float* out, void* stream);

/* packed: all annotation rows back to back, float64 [R,5] = (x1,y1,x2,y2,label) as the datasets hold them.
 * meta[b] = {first row of image b, m_b} (int64 x 2, device).  scale: float64 [B] or NULL (Resizer's factor).
 * out: [B,M,5] fp32: row r < m_b -> (float32(x*scale_b) x4, float32(label)); the other rows -1.  M >= max m_b, M >= 1. */
int mi355_collate_annots(const double* packed, const int64_t* meta, const double* scale, int B, int M,
                         float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355DET_INPUT_H_ */
