/* mi355det — C ABI of the MI355X (gfx950) one-stage-detection hot path.
 *
 * The reference (Kai7/ObjectDetection.Pytorch) is 100 % Python: it has no FFI of its own, so
 * each entry point below names the reference *Python* function (file:line under
 * /root/reference) whose arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions (last row):
 *   - plain pointers and sizes only; device pointers are BORROWED (caller-owned storage);
 *   - every function returns 0 on success, a negative MI355_E* code otherwise; no exceptions
 *     cross the ABI; `mi355_last_error()` returns a thread-local message;
 *   - no hidden allocation, no host<->device sync, no stream creation: all work is enqueued on
 *     the `stream` argument (a hipStream_t passed as void*; NULL = the null stream);
 *   - thread-safe: no mutable globals besides the thread-local error string.
 */
#ifndef MI355DET_H_
#define MI355DET_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_OK 0
#define MI355_EINVAL (-1)   /* bad argument / unsupported shape */
#define MI355_ELAUNCH (-2)  /* hipLaunch / runtime error */
#define MI355_ENOSPC (-3)   /* workspace too small */

int mi355_version(void);
/* Order `side_stream` behind everything issued to `main_stream` so far (event record + stream wait). */
int mi355_stream_fork(void* main_stream, void* side_stream);
const char* mi355_last_error(void);

/* ---- anchors : architectures/anchors.py:38-60 (Anchors.forward), :62-97, :99-119 -------- */
/* Host, fp64: the per-cell anchor set of one level (generate_anchors). out: [nr*ns][4]. */
int mi355_anchors_base(double* out, double base_size, const double* ratios, int nr,
                       const double* scales, int ns);
/* Host: level_offsets[nlevels+1] = running anchor counts (ceil grid, anchors.py:43). */
int mi355_anchor_count(int height, int width, int nlevels, const int* strides, int per_cell,
                       int64_t* level_offsets);
/* Device: writes fp32 [A][4] (x1,y1,x2,y2); fp64 shift+add, one rounding to fp32 (anchors.py:58).
 * h_base: host fp64 [nlevels][per_cell][4]. nlevels <= 8, per_cell <= 15. */
int mi355_anchors_generate(float* d_out, int height, int width, int nlevels, const int* strides,
                           const double* h_base, int per_cell, void* stream);

/* ---- IoU assignment + focal / smooth-L1 loss : architectures/losses.py:6-23, 30-168 ------ */
/* Workspace bytes for fwd (per-block partial sums, reduced in a fixed order -> deterministic). */
int mi355_focal_loss_workspace_bytes(int B, int A, int C, int M, int64_t* bytes);
/* cls [B][A][C] probabilities, reg [B][A][4], anchors [A][4], annot [B][M][5] (label -1 = pad).
 * Outputs: losses[2] = {cls_loss, reg_loss} (batch means, losses.py:168);
 *          argmax [B][A] int32 (index into the FILTERED annotation rows, losses.py:50,66);
 *          state  [B][A] int8  (0 neg, 1 ignore, 2 pos, 3 image without GT);
 *          num_pos [B] int32. */
int mi355_focal_loss_fwd(const float* cls, const float* reg, const float* anchors,
                         const float* annot, int B, int A, int C, int M, float neg_thr,
                         float pos_thr, float alpha, float* losses, int32_t* argmax,
                         int8_t* state, int32_t* num_pos, void* workspace,
                         int64_t workspace_bytes, void* stream);
/* Backward of the same function.  grad_losses[2] = upstream grads of {cls_loss, reg_loss}
 * (device memory).  grad_cls_mode: 0 -> d/dprob as fp32 [B][A][C];
 *                                  1 -> d/dlogit (= d/dprob * p * (1-p)) as bf16 [B][A][C]
 *                                       (feeds the classification-head dgrad/wgrad directly).
 * grad_reg: fp32 [B][A][4] (mode 0) or bf16 (mode 1).  In mode 1, cell_anchors > 0 writes both
 * gradients per pyramid cell (cell_anchors anchors each) with padded pitches, i.e. as the NHWC
 * rows the head convs consume: element offset of grad_cls (b, a, c) =
 * (b*(A/cell_anchors) + a/cell_anchors)*cls_cell_pitch + (a%cell_anchors)*C + c, of grad_reg
 * (b, a, j) = (...)*reg_cell_pitch + (a%cell_anchors)*4 + j  (pad columns are never written:
 * zero the buffers once).  grad_cls_mode 2 = mode 1 with fp32 elements (the fp32-accurate conv mode). */
int mi355_focal_loss_bwd(const float* cls, const float* reg, const float* anchors,
                         const float* annot, const int32_t* argmax, const int8_t* state,
                         const int32_t* num_pos, const float* grad_losses, int B, int A, int C,
                         int M, float alpha, int grad_mode, void* grad_cls, void* grad_reg,
                         int cell_anchors, int cls_cell_pitch, int reg_cell_pitch, void* stream);

/* ---- decode + clip : architectures/utils.py:26-53 (BBoxTransform), :61-71 (ClipBoxes) ---- */
/* boxes [1 or B][A][4] (boxes_batched = 0/1), deltas [B][A][4] -> out [B][A][4].
 * clip != 0 applies ClipBoxes with (height, width). mean/std: host float[4]. */
int mi355_bbox_decode(const float* boxes, int boxes_batched, const float* deltas, int B, int A,
                      const float* mean4, const float* std4, int clip, int height, int width,
                      float* out, void* stream);
/* ClipBoxes alone, in place. */
int mi355_clip_boxes(float* boxes, int B, int A, int height, int width, void* stream);

/* ---- score filter + NMS : architectures/retinanet.py:300-318; torchvision.ops.nms -------- */
/* Row max / argmax over classes: cls [A][C] -> score [A], label [A] int32 (first max wins). */
int mi355_class_max(const float* cls, int A, int C, float* score, int32_t* label, void* stream);
/* Stable compaction of indices with score > thr (ascending index). d_count: device int32. */
int mi355_score_filter(const float* score, int A, float thr, int32_t* idx_out, int32_t* d_count,
                       void* stream);
int mi355_nms_workspace_bytes(int K, int64_t* bytes);
/* Greedy NMS over K candidates: boxes [K][4], scores [K]; keep_out [K] int32 receives the kept
 * candidate indices in descending-score order (ties: ascending index), d_num_keep device int32.
 * Suppression test: IoU > iou_thr, areas without +1 (torchvision semantics). */
int mi355_nms(const float* boxes, const float* scores, int K, float iou_thr, int32_t* keep_out,
              int32_t* d_num_keep, void* workspace, int64_t workspace_bytes, void* stream);
/* The pieces of mi355_nms, for the two-phase schedule of large candidate sets (objectdetection/pytorch_amd/ops_detect.py
 * `nms`: greedy NMS compares a box only with KEPT boxes of higher score, so the bit matrix is built for a head of the
 * sorted candidates, the tail is tested against the head's kept boxes only, and the survivors go round again; same
 * result as mi355_nms):
 * mi355_nms_sort: candidates in descending score order (ties: ascending index) -> sorted_boxes [K][4], sorted_idx [K]
 *   (their original indices); workspace >= 8 * next_pow2(max(K, 2048)) bytes.
 * mi355_nms_presorted: the matrix NMS over boxes already in that order; kept ORIGINAL indices (sorted_idx) are written
 *   to keep_out from 0 (append == 0) or appended behind *d_num_keep.
 * mi355_nms_filter_tail: the tail boxes (sorted, with their original indices) that no box keep[kept_begin ..
 *   *d_num_keep) suppresses (boxes = the ORIGINAL box array keep indexes), compacted in order -> out_boxes / out_idx,
 *   count -> *d_out_count; workspace >= 12 * ceil(Kt / 64) + 64 bytes. */
int mi355_nms_sort(const float* boxes, const float* scores, int K, float* sorted_boxes, int32_t* sorted_idx,
                   void* workspace, int64_t workspace_bytes, void* stream);
int mi355_nms_presorted_workspace_bytes(int K, int64_t* bytes);
int mi355_nms_presorted(const float* sorted_boxes, const int32_t* sorted_idx, int K, float iou_thr, int32_t* keep_out,
                        int32_t* d_num_keep, int append, void* workspace, int64_t workspace_bytes, void* stream);
int mi355_nms_filter_tail(const float* boxes, const int32_t* keep, const int32_t* d_num_keep, int kept_begin,
                          const float* tail_boxes, const int32_t* tail_idx, int Kt, float iou_thr, float* out_boxes,
                          int32_t* out_idx, int32_t* d_out_count, void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355DET_H_ */
