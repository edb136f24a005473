// Detection primitives for gfx950: anchors, IoU assignment + focal/smooth-L1 loss (fwd, bwd),
// box decode + clip, class max, score filter, bitmask NMS.
//
// All of these are HBM-bound streaming kernels: coalesced 16-B loads,
// one pass over the [B,A,C] probability tensor per direction, wave64 ballot/prefix for the
// compactions.  Integer results (assignment state / argmax, NMS keep list) must be bit-exact
// with the PyTorch reference, so this file is compiled with -ffp-contract=off and every
// IoU is the reference's op sequence in fp32 with a correctly rounded division.
//
// Reference arithmetic: architectures/anchors.py:38-119, architectures/losses.py:6-168,
// architectures/utils.py:26-71, architectures/retinanet.py:300-318 (all under /root/reference).
#include "common.h"

#include <math.h>

namespace mi355 {
thread_local char g_err[512] = {0};
}
using namespace mi355;

extern "C" int mi355_version(void) { return 1; }

// Fork: everything issued to `main_stream` so far happens-before whatever is issued to `side_stream`
// next (hipEventRecord + hipStreamWaitEvent on a small pool of timing-free events; one host call instead
// of three Python-level ones per weight-gradient launch of the two-stream schedule).
extern "C" int mi355_stream_fork(void* main_stream, void* side_stream) {
    // one pool of events per (thread, device): an event belongs to the device that was current when it was made,
    // and a thread may drive plans on several devices (nn.DataParallel calls forward from one thread per GPU).
    // Events are made lazily one at a time, so a failure leaves no half-built pool behind.
    constexpr int POOL = 64, MAXDEV = 16;
    static thread_local hipEvent_t ev[MAXDEV][POOL];
    static thread_local int made[MAXDEV] = {0};
    static thread_local int next[MAXDEV] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return mi355::fail(MI355_ELAUNCH, "stream_fork: device %d", dev);
    const int slot = next[dev];
    if (slot >= made[dev]) {
        if (hipEventCreateWithFlags(&ev[dev][slot], hipEventDisableTiming) != hipSuccess)
            return mi355::fail(MI355_ELAUNCH, "stream_fork: event pool");
        made[dev] = slot + 1;
    }
    hipEvent_t e = ev[dev][slot];
    next[dev] = (slot + 1) % POOL;
    if (hipEventRecord(e, (hipStream_t)main_stream) != hipSuccess) return mi355::fail(MI355_ELAUNCH, "stream_fork: record");
    if (hipStreamWaitEvent((hipStream_t)side_stream, e, 0) != hipSuccess) return mi355::fail(MI355_ELAUNCH, "stream_fork: wait");
    return MI355_OK;
}
extern "C" const char* mi355_last_error(void) { return g_err; }

// ============================================================================ anchors
extern "C" int mi355_anchors_base(double* out, double base_size, const double* ratios, int nr,
                                  const double* scales, int ns) {
    MI355_REQUIRE(out && ratios && scales && nr > 0 && ns > 0, "anchors_base: bad args");
    // anchors.py:62-97: side = base*scale; area = side*side; w = sqrt(area/ratio); h = w*ratio;
    // (x_ctr,y_ctr,w,h) = (0,0,w,h) -> x1 = 0 - w*0.5, x2 = w - w*0.5 (the in-place `-=` order).
    int i = 0;
    for (int r = 0; r < nr; ++r)
        for (int s = 0; s < ns; ++s, ++i) {
            double side = base_size * scales[s];
            double area = side * side;
            double w = sqrt(area / ratios[r]);
            double h = w * ratios[r];
            out[4 * i + 0] = 0.0 - w * 0.5;
            out[4 * i + 1] = 0.0 - h * 0.5;
            out[4 * i + 2] = w - w * 0.5;
            out[4 * i + 3] = h - h * 0.5;
        }
    return MI355_OK;
}

extern "C" int mi355_anchor_count(int height, int width, int nlevels, const int* strides,
                                  int per_cell, int64_t* level_offsets) {
    MI355_REQUIRE(height > 0 && width > 0 && nlevels > 0 && strides && per_cell > 0 && level_offsets,
                  "anchor_count: bad args");
    int64_t off = 0;
    for (int l = 0; l < nlevels; ++l) {
        MI355_REQUIRE(strides[l] > 0, "anchor_count: stride <= 0");
        level_offsets[l] = off;
        off += (int64_t)((height + strides[l] - 1) / strides[l]) *
               ((width + strides[l] - 1) / strides[l]) * per_cell;
    }
    level_offsets[nlevels] = off;
    return MI355_OK;
}

struct AnchorArgs {
    int nlevels, per_cell, width, height;
    int stride[8];
    int gw[8];
    long long offset[9];
    double base[8 * 15 * 4];
};

__global__ void anchors_kernel(float4* __restrict__ out, AnchorArgs a) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.offset[a.nlevels]) return;
    int l = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (k < a.nlevels && i >= a.offset[k]) l = k;
    long long j = i - a.offset[l];
    int n = (int)(j % a.per_cell);
    long long cell = j / a.per_cell;
    int x = (int)(cell % a.gw[l]);
    int y = (int)(cell / a.gw[l]);
    // anchors.py:100-101: (arange + 0.5) * stride in fp64, then base + shift (one fp64 rounding),
    // then .astype(float32) (:58).
    double sx = ((double)x + 0.5) * (double)a.stride[l];
    double sy = ((double)y + 0.5) * (double)a.stride[l];
    const double* b = &a.base[(l * 15 + n) * 4];
    float4 r;
    r.x = (float)(b[0] + sx);
    r.y = (float)(b[1] + sy);
    r.z = (float)(b[2] + sx);
    r.w = (float)(b[3] + sy);
    out[i] = r;
}

extern "C" int mi355_anchors_generate(float* d_out, int height, int width, int nlevels,
                                      const int* strides, const double* h_base, int per_cell,
                                      void* stream) {
    MI355_REQUIRE(d_out && strides && h_base, "anchors_generate: null pointer");
    MI355_REQUIRE(nlevels > 0 && nlevels <= 8 && per_cell > 0 && per_cell <= 15,
                  "anchors_generate: nlevels<=8, per_cell<=15");
    AnchorArgs a;
    a.nlevels = nlevels;
    a.per_cell = per_cell;
    a.width = width;
    a.height = height;
    int64_t offs[9];
    int rc = mi355_anchor_count(height, width, nlevels, strides, per_cell, offs);
    if (rc) return rc;
    for (int l = 0; l < 8; ++l) {
        a.stride[l] = l < nlevels ? strides[l] : 1;
        a.gw[l] = l < nlevels ? (width + strides[l] - 1) / strides[l] : 1;
    }
    for (int l = 0; l <= 8; ++l) a.offset[l] = offs[l < nlevels ? l : nlevels];
    for (int l = 0; l < nlevels; ++l)
        for (int n = 0; n < per_cell; ++n)
            for (int k = 0; k < 4; ++k) a.base[(l * 15 + n) * 4 + k] = h_base[(l * per_cell + n) * 4 + k];
    int64_t total = offs[nlevels];
    if (total == 0) return MI355_OK;
    hipLaunchKernelGGL(anchors_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, (float4*)d_out, a);
    return check_launch("anchors_kernel");
}

// ============================================================================ IoU + assignment
enum { ST_NEG = 0, ST_IGNORE = 1, ST_POS = 2, ST_NOGT = 3 };

// losses.py:6-23 with the reference's op order; every op rounds to fp32 (no FMA contraction).
__device__ __forceinline__ float iou_ref(float ax1, float ay1, float ax2, float ay2, float a_area,
                                         float bx1, float by1, float bx2, float by2, float b_area) {
    float iw = fminf(ax2, bx2) - fmaxf(ax1, bx1);
    float ih = fminf(ay2, by2) - fmaxf(ay1, by1);
    iw = fmaxf(iw, 0.0f);
    ih = fmaxf(ih, 0.0f);
    float inter = iw * ih;
    float ua = (a_area + b_area) - inter;
    ua = fmaxf(ua, 1e-8f);
    return __fdiv_rn(inter, ua);
}

// Compacts the rows of one image's annotations with label != -1 (losses.py:50) into LDS,
// preserving order, with a wave64 ballot + prefix popcount.  gt: [nvalid][6] = box, label, area.
// Called by all threads of the block; returns nvalid (uniform).
__device__ int compact_gt(const float* __restrict__ ann, int M, float* gt, int* s_count) {
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        int base = 0;
        for (int m0 = 0; m0 < M; m0 += 64) {
            int m = m0 + lane;
            float x1 = 0, y1 = 0, x2 = 0, y2 = 0, lab = -1.0f;
            if (m < M) {
                x1 = ann[m * 5 + 0];
                y1 = ann[m * 5 + 1];
                x2 = ann[m * 5 + 2];
                y2 = ann[m * 5 + 3];
                lab = ann[m * 5 + 4];
            }
            bool valid = (m < M) && (lab != -1.0f);
            unsigned long long mask = __ballot(valid);
            int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
            if (valid) {
                gt[pos * 6 + 0] = x1;
                gt[pos * 6 + 1] = y1;
                gt[pos * 6 + 2] = x2;
                gt[pos * 6 + 3] = y2;
                gt[pos * 6 + 4] = lab;
                gt[pos * 6 + 5] = (x2 - x1) * (y2 - y1);   // losses.py:7
            }
            base += __popcll(mask);
        }
        if (lane == 0) *s_count = base;
    }
    __syncthreads();
    return *s_count;
}

constexpr int FL_TILE = 256;   // anchors per block == threads per block

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_red[w] = v;
    __syncthreads();
    float r = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    __syncthreads();
    return r;
}

// smooth-L1 regression targets (losses.py:130-150) for one positive anchor.
__device__ __forceinline__ void reg_targets(const float4 an, const float* g, float t[4]) {
    float aw = an.z - an.x, ah = an.w - an.y;
    float acx = an.x + 0.5f * aw, acy = an.y + 0.5f * ah;
    float gw = g[2] - g[0], gh = g[3] - g[1];
    float gcx = g[0] + 0.5f * gw, gcy = g[1] + 0.5f * gh;   // centre BEFORE the clamp
    gw = fmaxf(gw, 1.0f);
    gh = fmaxf(gh, 1.0f);
    t[0] = __fdiv_rn(__fdiv_rn(gcx - acx, aw), 0.1f);
    t[1] = __fdiv_rn(__fdiv_rn(gcy - acy, ah), 0.1f);
    t[2] = __fdiv_rn(logf(__fdiv_rn(gw, aw)), 0.2f);
    t[3] = __fdiv_rn(logf(__fdiv_rn(gh, ah)), 0.2f);
}

template <int VEC>
__global__ __launch_bounds__(FL_TILE) void focal_fwd_kernel(
    const float* __restrict__ cls, const float* __restrict__ reg, const float4* __restrict__ anchors,
    const float* __restrict__ annot, int A, int C, int M, float neg_thr, float pos_thr, float alpha,
    int32_t* __restrict__ argmax_out, int8_t* __restrict__ state_out, float* __restrict__ part_cls,
    float* __restrict__ part_reg, int* __restrict__ part_pos) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* gt = smem;                                  // [M][6]
    __shared__ int s_state[FL_TILE];
    __shared__ int s_label[FL_TILE];
    __shared__ float s_red[4];
    __shared__ int s_count;

    const int b = blockIdx.y, tile = blockIdx.x, nblk = gridDim.x, tid = threadIdx.x;
    const int a0 = tile * FL_TILE;
    const int na = min(FL_TILE, A - a0);
    const int nvalid = compact_gt(annot + (size_t)b * M * 5, M, gt, &s_count);

    // ---- phase 1: one anchor per thread: IoU row max / argmax, thresholds, box loss
    float reg_sum = 0.0f;
    int is_pos = 0;
    int st = ST_NOGT, lab = -1, arg = 0;
    if (tid < na) {
        const int a = a0 + tid;
        if (nvalid > 0) {
            const float4 an = anchors[a];
            const float a_area = (an.z - an.x) * (an.w - an.y);          // losses.py:15
            float best = iou_ref(an.x, an.y, an.z, an.w, a_area, gt[0], gt[1], gt[2], gt[3], gt[5]);
            for (int g = 1; g < nvalid; ++g) {
                const float* q = gt + g * 6;
                float v = iou_ref(an.x, an.y, an.z, an.w, a_area, q[0], q[1], q[2], q[3], q[5]);
                if (v > best) { best = v; arg = g; }                      // ties: lowest index
            }
            st = ST_IGNORE;
            if (best < neg_thr) st = ST_NEG;                              // losses.py:79
            if (best >= pos_thr) st = ST_POS;                             // losses.py:80
            if (st == ST_POS) {
                is_pos = 1;
                lab = (int)(long long)gt[arg * 6 + 4];                    // .long() truncation
                float t[4];
                reg_targets(an, gt + arg * 6, t);
                const float4 r = *reinterpret_cast<const float4*>(reg + ((size_t)b * A + a) * 4);
                const float rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float d = fabsf(t[k] - rv[k]);                        // losses.py:154-160
                    reg_sum += (d <= (1.0f / 9.0f)) ? (0.5f * 9.0f) * (d * d) : d - (0.5f / 9.0f);
                }
            }
        }
        argmax_out[(size_t)b * A + a] = arg;
        state_out[(size_t)b * A + a] = (int8_t)st;
    }
    s_state[tid] = st;
    s_label[tid] = lab;
    __syncthreads();

    // ---- phase 2: stream the [na][C] probabilities of this tile once, 16 B per lane
    float csum = 0.0f;
    if (nvalid > 0) {
        const float* base = cls + ((size_t)b * A + a0) * C;
        const int nvec = (na * C) / VEC;
        for (int i = tid; i < nvec; i += FL_TILE) {
            float v[VEC];
            if (VEC == 4) {
                const float4 q = reinterpret_cast<const float4*>(base)[i];
                v[0] = q.x; v[1 % VEC] = q.y; v[2 % VEC] = q.z; v[3 % VEC] = q.w;
            } else {
                v[0] = base[i];
            }
            const int e0 = i * VEC;
            int al = e0 / C;
            int c = e0 - al * C;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const int s = s_state[al];
                if (s != ST_IGNORE) {
                    const float p = fminf(fmaxf(v[k], 1e-4f), 1.0f - 1e-4f);  // losses.py:62
                    const bool one = (s == ST_POS) && (c == s_label[al]);
                    // losses.py:104-111 with gamma = 2
                    // u = probability given to the true class, w = the other one: ONE logf per element, same
                    // operations and order as the two branches of the reference formula
                    const float q = 1.0f - p;
                    const float u = one ? p : q, w = one ? q : p;
                    const float fw = (one ? alpha : 1.0f - alpha) * (w * w);
                    const float bce = -logf(u);
                    csum += fw * bce;
                }
                if (++c == C) { c = 0; ++al; }
            }
        }
    }
    const float cs = block_sum_256(csum, s_red);
    const float rs = block_sum_256(reg_sum, s_red);
    const unsigned long long pm = __ballot(is_pos);
    __shared__ int s_pos[4];
    if ((tid & 63) == 0) s_pos[tid >> 6] = __popcll(pm);
    __syncthreads();
    if (tid == 0) {
        part_cls[(size_t)b * nblk + tile] = cs;
        part_reg[(size_t)b * nblk + tile] = rs;
        part_pos[(size_t)b * nblk + tile] = s_pos[0] + s_pos[1] + s_pos[2] + s_pos[3];
    }
}

// Fixed-order reduction of the per-block partials (deterministic), losses.py:118, 161, 168.
__global__ __launch_bounds__(1024) void focal_finalize_kernel(const float* part_cls, const float* part_reg,
                                                              const int* part_pos, int B, int nblk, float* losses,
                                                              int32_t* num_pos) {
    // one wave per image (16 at a time; one block of 256 threads walking the images one after another with a barrier
    // tree each took 22 us at batch 16): lane-strided partial sums in fp64, a fixed shuffle tree, then thread 0 adds the
    // images in index order
    extern __shared__ float s_img[];                   // [B][2]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int b = wv; b < B; b += 16) {
        double c = 0, r = 0;
        int p = 0;
        for (int i = lane; i < nblk; i += 64) {
            c += part_cls[(size_t)b * nblk + i];
            r += part_reg[(size_t)b * nblk + i];
            p += part_pos[(size_t)b * nblk + i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            c += __shfl_down(c, o, 64);
            r += __shfl_down(r, o, 64);
            p += __shfl_down(p, o, 64);
        }
        if (lane == 0) {
            num_pos[b] = p;
            s_img[2 * b] = __fdiv_rn((float)c, fmaxf((float)p, 1.0f));
            s_img[2 * b + 1] = p > 0 ? __fdiv_rn((float)r, (float)(4 * p)) : 0.0f;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float lc = 0.0f, lr = 0.0f;
        for (int b = 0; b < B; ++b) { lc += s_img[2 * b]; lr += s_img[2 * b + 1]; }
        losses[0] = __fdiv_rn(lc, (float)B);
        losses[1] = __fdiv_rn(lr, (float)B);
    }
}

extern "C" int mi355_focal_loss_workspace_bytes(int B, int A, int C, int M, int64_t* bytes) {
    MI355_REQUIRE(B > 0 && A > 0 && C > 0 && M >= 0 && bytes, "focal_loss_workspace_bytes: bad args");
    int64_t nblk = cdiv(A, FL_TILE);
    *bytes = (int64_t)B * nblk * 12;
    return MI355_OK;
}

extern "C" int mi355_focal_loss_fwd(const float* cls, const float* reg, const float* anchors,
                                    const float* annot, int B, int A, int C, int M, float neg_thr,
                                    float pos_thr, float alpha, float* losses, int32_t* argmax,
                                    int8_t* state, int32_t* num_pos, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
    MI355_REQUIRE(cls && reg && anchors && losses && argmax && state && num_pos && workspace,
                  "focal_loss_fwd: null pointer");
    MI355_REQUIRE(B > 0 && A > 0 && C > 0 && M >= 0 && (annot || M == 0), "focal_loss_fwd: bad shape");
    MI355_REQUIRE((int64_t)M * 24 <= 96 * 1024, "focal_loss_fwd: M too large for LDS (max 4096)");
    MI355_REQUIRE(B <= 65535, "focal_loss_fwd: B > 65535");
    int64_t need;
    mi355_focal_loss_workspace_bytes(B, A, C, M, &need);
    if (workspace_bytes < need) return fail(MI355_ENOSPC, "focal_loss_fwd: workspace %lld < %lld",
                                            (long long)workspace_bytes, (long long)need);
    const int nblk = (int)cdiv(A, FL_TILE);
    float* part_cls = (float*)workspace;
    float* part_reg = part_cls + (size_t)B * nblk;
    int* part_pos = (int*)(part_reg + (size_t)B * nblk);
    dim3 grid(nblk, B), block(FL_TILE);
    size_t lds = (size_t)(M > 0 ? M : 1) * 24;
    const bool vec4 = (C % 4 == 0) && (((uintptr_t)cls & 15) == 0);
    if (vec4)
        hipLaunchKernelGGL(focal_fwd_kernel<4>, grid, block, lds, (hipStream_t)stream, cls, reg,
                           (const float4*)anchors, annot, A, C, M, neg_thr, pos_thr, alpha, argmax,
                           state, part_cls, part_reg, part_pos);
    else
        hipLaunchKernelGGL(focal_fwd_kernel<1>, grid, block, lds, (hipStream_t)stream, cls, reg,
                           (const float4*)anchors, annot, A, C, M, neg_thr, pos_thr, alpha, argmax,
                           state, part_cls, part_reg, part_pos);
    int rc = check_launch("focal_fwd_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(1024), (size_t)B * 2 * sizeof(float), (hipStream_t)stream, part_cls,
                       part_reg, part_pos, B, nblk, losses, num_pos);
    return check_launch("focal_finalize_kernel");
}

__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);   // v_cvt_pk_bf16_f32: RNE, NaN-safe
}

template <int VEC, int MODE>
__global__ __launch_bounds__(FL_TILE) void focal_bwd_kernel(
    const float* __restrict__ cls, const float* __restrict__ reg, const float4* __restrict__ anchors,
    const float* __restrict__ annot, const int32_t* __restrict__ argmax,
    const int8_t* __restrict__ state, const int32_t* __restrict__ num_pos,
    const float* __restrict__ grad_losses, int B, int A, int C, int M, float alpha,
    void* __restrict__ grad_cls, void* __restrict__ grad_reg, int cell_anchors, int cls_pitch,
    int cell_pitch) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* gt = smem;
    __shared__ int s_state[FL_TILE];
    __shared__ int s_label[FL_TILE];
    __shared__ int s_count;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int a0 = blockIdx.x * FL_TILE;
    const int na = min(FL_TILE, A - a0);
    const int nvalid = compact_gt(annot + (size_t)b * M * 5, M, gt, &s_count);
    const int np = num_pos[b];
    // d(mean_b)/d(loss_b) = 1/B; cls: /max(np,1) (losses.py:118); reg: mean over np*4 (:161)
    const float gc = __fdiv_rn(__fdiv_rn(grad_losses[0], (float)B), fmaxf((float)np, 1.0f));
    const float gr = np > 0 ? __fdiv_rn(__fdiv_rn(grad_losses[1], (float)B), (float)(4 * np)) : 0.0f;

    int st = ST_NOGT, lab = -1;
    if (tid < na) {
        const int a = a0 + tid;
        st = nvalid > 0 ? (int)state[(size_t)b * A + a] : ST_NOGT;
        float g4[4] = {0.f, 0.f, 0.f, 0.f};
        if (st == ST_POS) {
            const int arg = argmax[(size_t)b * A + a];
            lab = (int)(long long)gt[arg * 6 + 4];
            float t[4];
            reg_targets(anchors[a], gt + arg * 6, t);
            const float4 r = *reinterpret_cast<const float4*>(reg + ((size_t)b * A + a) * 4);
            const float rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float diff = t[k] - rv[k];
                const float d = fabsf(diff);
                const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
                // d/dreg of where(d<=1/9, 4.5 d^2, d - 0.5/9) with d = |t - reg|
                g4[k] = gr * ((d <= (1.0f / 9.0f)) ? 9.0f * d : 1.0f) * (-sgn);
            }
        }
        if (MODE == 0) {
            reinterpret_cast<float4*>(grad_reg)[(size_t)b * A + a] = make_float4(g4[0], g4[1], g4[2], g4[3]);
        } else if (MODE == 2) {                    // fp32-accurate mode: fp32 rows, cell layout as mode 1
            float* dst = reinterpret_cast<float*>(grad_reg);
            if (cell_anchors > 0) {
                const int cell = a / cell_anchors, j = a - cell * cell_anchors;
                dst += ((size_t)b * (A / cell_anchors) + cell) * cell_pitch + j * 4;
            } else {
                dst += ((size_t)b * A + a) * 4;
            }
            dst[0] = g4[0]; dst[1] = g4[1]; dst[2] = g4[2]; dst[3] = g4[3];
        } else {
            ushort4 o;
            o.x = f32_to_bf16(g4[0]); o.y = f32_to_bf16(g4[1]);
            o.z = f32_to_bf16(g4[2]); o.w = f32_to_bf16(g4[3]);
            if (cell_anchors > 0) {
                const int cell = a / cell_anchors, j = a - cell * cell_anchors;
                unsigned short* dst = reinterpret_cast<unsigned short*>(grad_reg) +
                                      ((size_t)b * (A / cell_anchors) + cell) * cell_pitch + j * 4;
                dst[0] = o.x; dst[1] = o.y; dst[2] = o.z; dst[3] = o.w;
            } else {
                reinterpret_cast<ushort4*>(grad_reg)[(size_t)b * A + a] = o;
            }
        }
    }
    s_state[tid] = st;
    s_label[tid] = lab;
    __syncthreads();

    const size_t tbase = ((size_t)b * A + a0) * C;
    const float* base = cls + tbase;
    const int nvec = (na * C) / VEC;
    for (int i = tid; i < nvec; i += FL_TILE) {
        float v[VEC], g[VEC];
        if (VEC == 4) {
            const float4 q = reinterpret_cast<const float4*>(base)[i];
            v[0] = q.x; v[1 % VEC] = q.y; v[2 % VEC] = q.z; v[3 % VEC] = q.w;
        } else {
            v[0] = base[i];
        }
        const int e0 = i * VEC;
        int al = e0 / C;
        int c = e0 - al * C;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int s = s_state[al];
            float gv = 0.0f;
            const float x = v[k];
            // clamp passes gradient only inside [1e-4, 1-1e-4] (torch.clamp backward)
            if (s != ST_IGNORE && s != ST_NOGT && x >= 1e-4f && x <= 1.0f - 1e-4f) {
                const float p = x;
                const bool one = (s == ST_POS) && (c == s_label[al]);
                // positive: L = a (1-p)^2 (-log p)      -> dL/dp =    a  (2 q log p - q^2 / p)
                // negative: L = (1-a) p^2 (-log(1-p))   -> dL/dp = (1-a) (p^2 / q - 2 p log q)
                // with u = true-class probability, w = the other one both are  c (2 w log u - w^2 / u), c = a or
                // -(1-a): one logf and one division per element, no divergent branch, bit-identical values
                const float q = 1.0f - p;
                const float u = one ? p : q, w = one ? q : p;
                // (bf16 head gradients, MODE 1: the hardware log2 / reciprocal, ~1e-6 relative, far below the 4e-3 of the
                //  bf16 rounding that follows; the kernel is VALU-bound on the IEEE logf + division of the exact forms)
                const float lg = MODE == 1 ? __builtin_amdgcn_logf(u) * 0.693147180559945f : logf(u);
                const float dv = MODE == 1 ? (w * w) * __builtin_amdgcn_rcpf(u) : __fdiv_rn(w * w, u);
                const float t = 2.0f * w * lg - dv;
                const float d = one ? alpha * t : (1.0f - alpha) * (-t);
                gv = gc * d;
                if (MODE != 0) gv *= x * (1.0f - x);     // d prob / d logit of the head sigmoid
            }
            g[k] = gv;
            if (++c == C) { c = 0; ++al; }
        }
        if (MODE == 0) {
            float* o = reinterpret_cast<float*>(grad_cls) + tbase;
            if (VEC == 4) reinterpret_cast<float4*>(o)[i] = make_float4(g[0], g[1 % VEC], g[2 % VEC], g[3 % VEC]);
            else o[i] = g[0];
        } else if (MODE == 2) {
            const int aa = a0 + e0 / C, cc = e0 - (e0 / C) * C;
            float* o = reinterpret_cast<float*>(grad_cls);
            if (cell_anchors > 0) {
                const int cell = aa / cell_anchors;
                o += ((size_t)b * (A / cell_anchors) + cell) * cls_pitch + (size_t)(aa - cell * cell_anchors) * C + cc;
            } else {
                o += tbase + e0;
            }
            if (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(g[0], g[1 % VEC], g[2 % VEC], g[3 % VEC]);
            else o[0] = g[0];
        } else if (cell_anchors > 0) {
            // NHWC head-gradient rows: (cell, anchor-in-cell, class) with a padded cell pitch
            const int aa = a0 + e0 / C, cc = e0 - (e0 / C) * C;
            const int cell = aa / cell_anchors;
            unsigned short* o = reinterpret_cast<unsigned short*>(grad_cls) +
                                ((size_t)b * (A / cell_anchors) + cell) * cls_pitch +
                                (size_t)(aa - cell * cell_anchors) * C + cc;
            if (VEC == 4) {
                ushort4 w;
                w.x = f32_to_bf16(g[0]); w.y = f32_to_bf16(g[1 % VEC]);
                w.z = f32_to_bf16(g[2 % VEC]); w.w = f32_to_bf16(g[3 % VEC]);
                *reinterpret_cast<ushort4*>(o) = w;
            } else {
                o[0] = f32_to_bf16(g[0]);
            }
        } else {
            unsigned short* o = reinterpret_cast<unsigned short*>(grad_cls) + tbase;
            if (VEC == 4) {
                ushort4 w;
                w.x = f32_to_bf16(g[0]); w.y = f32_to_bf16(g[1 % VEC]);
                w.z = f32_to_bf16(g[2 % VEC]); w.w = f32_to_bf16(g[3 % VEC]);
                reinterpret_cast<ushort4*>(o)[i] = w;
            } else {
                o[i] = f32_to_bf16(g[0]);
            }
        }
    }
}

extern "C" int mi355_focal_loss_bwd(const float* cls, const float* reg, const float* anchors,
                                    const float* annot, const int32_t* argmax, const int8_t* state,
                                    const int32_t* num_pos, const float* grad_losses, int B, int A,
                                    int C, int M, float alpha, int grad_mode, void* grad_cls,
                                    void* grad_reg, int cell_anchors, int cls_cell_pitch,
                                    int reg_cell_pitch, void* stream) {
    MI355_REQUIRE(cls && reg && anchors && argmax && state && num_pos && grad_losses && grad_cls &&
                  grad_reg, "focal_loss_bwd: null pointer");
    MI355_REQUIRE(B > 0 && B <= 65535 && A > 0 && C > 0 && M >= 0, "focal_loss_bwd: bad shape");
    MI355_REQUIRE((int64_t)M * 24 <= 96 * 1024, "focal_loss_bwd: M too large");
    MI355_REQUIRE(grad_mode >= 0 && grad_mode <= 2, "focal_loss_bwd: grad_mode");
    MI355_REQUIRE(cell_anchors == 0 || (grad_mode >= 1 && A % cell_anchors == 0 &&
                                        reg_cell_pitch >= cell_anchors * 4 && reg_cell_pitch % 4 == 0 &&
                                        cls_cell_pitch >= cell_anchors * C && cls_cell_pitch % 4 == 0),
                  "focal_loss_bwd: bad cell layout");
    const int nblk = (int)cdiv(A, FL_TILE);
    dim3 grid(nblk, B), block(FL_TILE);
    size_t lds = (size_t)(M > 0 ? M : 1) * 24;
    const bool vec4 = (C % 4 == 0) && (((uintptr_t)cls & 15) == 0) && (((uintptr_t)grad_cls & 15) == 0);
#define LAUNCH_BWD(V, MD)                                                                        \
    hipLaunchKernelGGL((focal_bwd_kernel<V, MD>), grid, block, lds, (hipStream_t)stream, cls,    \
                       reg, (const float4*)anchors, annot, argmax, state, num_pos, grad_losses,  \
                       B, A, C, M, alpha, grad_cls, grad_reg, cell_anchors, cls_cell_pitch, reg_cell_pitch)
    if (vec4) { if (grad_mode == 0) LAUNCH_BWD(4, 0); else if (grad_mode == 1) LAUNCH_BWD(4, 1); else LAUNCH_BWD(4, 2); }
    else      { if (grad_mode == 0) LAUNCH_BWD(1, 0); else if (grad_mode == 1) LAUNCH_BWD(1, 1); else LAUNCH_BWD(1, 2); }
#undef LAUNCH_BWD
    return check_launch("focal_bwd_kernel");
}

// ============================================================================ decode + clip
struct Decode4 { float m[4]; float s[4]; };

__global__ void bbox_decode_kernel(const float4* __restrict__ boxes, int boxes_batched,
                                   const float4* __restrict__ deltas, long long total, int A,
                                   Decode4 ms, int clip, float wmax, float hmax,
                                   float4* __restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float4 bx = boxes[boxes_batched ? i : (i % A)];
    const float4 d = deltas[i];
    // utils.py:28-51, op for op
    const float w = bx.z - bx.x, h = bx.w - bx.y;
    const float cx = bx.x + 0.5f * w, cy = bx.y + 0.5f * h;
    const float dx = d.x * ms.s[0] + ms.m[0];
    const float dy = d.y * ms.s[1] + ms.m[1];
    const float dw = d.z * ms.s[2] + ms.m[2];
    const float dh = d.w * ms.s[3] + ms.m[3];
    const float pcx = cx + dx * w, pcy = cy + dy * h;
    const float pw = expf(dw) * w, ph = expf(dh) * h;
    float4 o;
    o.x = pcx - 0.5f * pw;
    o.y = pcy - 0.5f * ph;
    o.z = pcx + 0.5f * pw;
    o.w = pcy + 0.5f * ph;
    if (clip) {   // utils.py:65-69: one-sided clamps
        o.x = fmaxf(o.x, 0.0f);
        o.y = fmaxf(o.y, 0.0f);
        o.z = fminf(o.z, wmax);
        o.w = fminf(o.w, hmax);
    }
    out[i] = o;
}

extern "C" int mi355_bbox_decode(const float* boxes, int boxes_batched, const float* deltas, int B,
                                 int A, const float* mean4, const float* std4, int clip, int height,
                                 int width, float* out, void* stream) {
    MI355_REQUIRE(boxes && deltas && mean4 && std4 && out && B > 0 && A > 0, "bbox_decode: bad args");
    Decode4 ms;
    for (int k = 0; k < 4; ++k) { ms.m[k] = mean4[k]; ms.s[k] = std4[k]; }
    long long total = (long long)B * A;
    hipLaunchKernelGGL(bbox_decode_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)boxes, boxes_batched, (const float4*)deltas,
                       total, A, ms, clip, (float)(width - 1), (float)(height - 1), (float4*)out);
    return check_launch("bbox_decode_kernel");
}

__global__ void clip_boxes_kernel(float4* boxes, long long total, float wmax, float hmax) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float4 o = boxes[i];
    o.x = fmaxf(o.x, 0.0f);
    o.y = fmaxf(o.y, 0.0f);
    o.z = fminf(o.z, wmax);
    o.w = fminf(o.w, hmax);
    boxes[i] = o;
}

extern "C" int mi355_clip_boxes(float* boxes, int B, int A, int height, int width, void* stream) {
    MI355_REQUIRE(boxes && B > 0 && A > 0, "clip_boxes: bad args");
    long long total = (long long)B * A;
    hipLaunchKernelGGL(clip_boxes_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, (float4*)boxes, total, (float)(width - 1), (float)(height - 1));
    return check_launch("clip_boxes_kernel");
}

// ============================================================================ class max
// One wave per anchor row: lanes stride over the C classes, wave arg-max reduce (first max wins,
// as torch.max(dim) on CPU).
__global__ __launch_bounds__(256) void class_max_kernel(const float* __restrict__ cls, int A, int C,
                                                        float* __restrict__ score,
                                                        int32_t* __restrict__ label) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= A) return;
    const float* p = cls + row * C;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const float v = p[c];
        if (v > best || (v == best && c < bi)) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { score[row] = best; label[row] = bi; }
}

extern "C" int mi355_class_max(const float* cls, int A, int C, float* score, int32_t* label,
                               void* stream) {
    MI355_REQUIRE(cls && score && label && A > 0 && C > 0, "class_max: bad args");
    hipLaunchKernelGGL(class_max_kernel, dim3((unsigned)cdiv(A, 4)), dim3(256), 0, (hipStream_t)stream,
                       cls, A, C, score, label);
    return check_launch("class_max_kernel");
}

// ============================================================================ score filter
// Single block, order-preserving compaction: per-wave ballot + popcount prefix, running offset.
__global__ __launch_bounds__(1024) void score_filter_kernel(const float* __restrict__ score, int A,
                                                            float thr, int32_t* __restrict__ idx_out,
                                                            int32_t* __restrict__ d_count) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < A; i0 += 1024) {
        const int i = i0 + tid;
        const bool keep = (i < A) && (score[i] > thr);            // retinanet.py:302
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wave[wv] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wv; ++w) off += s_wave[w];
        if (keep) idx_out[off + __popcll(m & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += s_wave[w];
            s_base += t;
        }
        __syncthreads();
    }
    if (tid == 0) *d_count = s_base;
}

extern "C" int mi355_score_filter(const float* score, int A, float thr, int32_t* idx_out,
                                  int32_t* d_count, void* stream) {
    MI355_REQUIRE(score && idx_out && d_count && A > 0, "score_filter: bad args");
    hipLaunchKernelGGL(score_filter_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, score, A, thr,
                       idx_out, d_count);
    return check_launch("score_filter_kernel");
}

// ============================================================================ NMS
// 1) 64-bit keys (descending score, ascending index) sorted with a bitonic network
//    (LDS for spans <= 2048 keys, global passes above);
// 2) upper-triangular suppression bit matrix, 64x64 tiles, row box in registers, column boxes
//    in LDS;
// 3) one-block sweep: per 64-row tile resolve the diagonal word serially in a wave
//    (v_readlane), then OR the surviving rows into the running `removed` bitmap in parallel.
constexpr int SORT_CHUNK = 2048;

__device__ __forceinline__ unsigned int order_desc(float f) {
    unsigned int u = __float_as_uint(f);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;   // ascending-ordered
    return ~u;                                     // descending
}

__global__ void nms_make_keys_kernel(const float* __restrict__ scores, int K, int K2,
                                     unsigned long long* __restrict__ keys) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K2) return;
    keys[i] = i < K ? (((unsigned long long)order_desc(scores[i]) << 32) | (unsigned)i)
                    : 0xFFFFFFFFFFFFFFFFull;
}

__device__ __forceinline__ void cmpswap(unsigned long long& a, unsigned long long& b, bool up) {
    if ((a > b) == up) { unsigned long long t = a; a = b; b = t; }
}

// Sort every SORT_CHUNK-span fully (k = 2..SORT_CHUNK) in LDS.
__global__ __launch_bounds__(256) void bitonic_chunk_sort_kernel(unsigned long long* keys) {
    __shared__ unsigned long long s[SORT_CHUNK];
    const int tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * SORT_CHUNK;
    for (int i = tid; i < SORT_CHUNK; i += 256) s[i] = keys[base + i];
    __syncthreads();
    for (int k = 2; k <= SORT_CHUNK; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < SORT_CHUNK / 2; t += 256) {
                const int i = 2 * t - (t & (j - 1));
                const bool up = (((base + i) & k) == 0);
                cmpswap(s[i], s[i + j], up);
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < SORT_CHUNK; i += 256) keys[base + i] = s[i];
}

__global__ void bitonic_global_step_kernel(unsigned long long* keys, int K2, int k, int j) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= K2 / 2) return;
    const int i = 2 * t - (t & (j - 1));
    const bool up = ((i & k) == 0);
    unsigned long long a = keys[i], b = keys[i + j];
    if ((a > b) == up) { keys[i] = b; keys[i + j] = a; }
}

// Finish stage k for j = SORT_CHUNK/2 .. 1 in LDS.
__global__ __launch_bounds__(256) void bitonic_chunk_merge_kernel(unsigned long long* keys, int k) {
    __shared__ unsigned long long s[SORT_CHUNK];
    const int tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * SORT_CHUNK;
    for (int i = tid; i < SORT_CHUNK; i += 256) s[i] = keys[base + i];
    __syncthreads();
    for (int j = SORT_CHUNK >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < SORT_CHUNK / 2; t += 256) {
            const int i = 2 * t - (t & (j - 1));
            const bool up = (((base + i) & (size_t)k) == 0);
            cmpswap(s[i], s[i + j], up);
        }
        __syncthreads();
    }
    for (int i = tid; i < SORT_CHUNK; i += 256) keys[base + i] = s[i];
}

__global__ void nms_gather_boxes_kernel(const unsigned long long* __restrict__ keys,
                                        const float4* __restrict__ boxes, int K,
                                        float4* __restrict__ sorted_boxes,
                                        int32_t* __restrict__ sorted_idx) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const int src = (int)(keys[i] & 0xFFFFFFFFull);
    sorted_idx[i] = src;
    sorted_boxes[i] = boxes[src];
}

// Upper-triangle suppression bit matrix: mask[i][jt] bit c = (IoU(box i, box jt*64 + c) > thr) for j > i, i and j in
// descending-score order.  One block = row tile `it` (64 boxes, one per lane) x a share of the column tiles
// jt >= it (the four waves of a block take every fourth of them).  The column box of a step is wave-uniform:
// v_readlane -> scalar operands.  Decision without a division in the common case: q = inter * rcp(union) is within a
// few ulp of the correctly rounded quotient; only when q lies within 1e-5 (relative) of the threshold is the IEEE
// division (the reference arithmetic: `iou > thr` on the rounded quotient) evaluated.  The matrix is stored TILE-major,
// mask[(it * W + jt) * 64 + lane]: the 64 words of a (row tile, column tile) pair are one contiguous 512-B store (row-major
// put every lane's word on its own cache line: 17 M scattered 8-B stores per 46 706 boxes, ~1 ms of write amplification).
// Pairs with jt < it are never written nor read.
__device__ __forceinline__ float bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// IoU(a, b) > thr exactly as `iou > thr` on the correctly rounded quotient (see the mask kernel's comment)
__device__ __forceinline__ bool iou_over(const float4 a, float area_a, float bx, float by, float bz, float bw,
                                         float area_b, float thr, float lo, float hi) {
    const float w = fmaxf(fminf(a.z, bz) - fmaxf(a.x, bx), 0.0f);
    const float h = fmaxf(fminf(a.w, bw) - fmaxf(a.y, by), 0.0f);
    const float inter = w * h;
    const float uni = (area_a + area_b) - inter;
    const float q = inter * __builtin_amdgcn_rcpf(uni);
    bool over = q > hi;
    if (!(q > hi) && !(q < lo)) over = __fdiv_rn(inter, uni) > thr;     // near the threshold (or NaN): exact
    return over;
}

__global__ __launch_bounds__(256) void nms_mask_kernel(const float4* __restrict__ sb, int K, int W,
                                                       float thr, unsigned long long* __restrict__ mask) {
    const int it = blockIdx.x, part = blockIdx.y, nparts = gridDim.y;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = it * 64 + lane;
    const float4 a = i < K ? sb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float area_a = (a.z - a.x) * (a.w - a.y);
    const float lo = thr * (1.0f - 1e-5f), hi = thr * (1.0f + 1e-5f);
    for (int jt = it + part * 4 + wv; jt < W; jt += 4 * nparts) {
        const int j0 = jt * 64;
        const float4 mine = (j0 + lane < K) ? sb[j0 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float area_m = (mine.z - mine.x) * (mine.w - mine.y);
        unsigned long long bits = 0;
        auto pair = [&](int c) {
            const float bx = bcast(mine.x, c), by = bcast(mine.y, c), bz = bcast(mine.z, c), bw = bcast(mine.w, c);
            const float area_b = bcast(area_m, c);
            const bool over = iou_over(a, area_a, bx, by, bz, bw, area_b, thr, lo, hi);
            if (j0 + c > i && over) bits |= (1ull << c);
        };
        if (j0 + 64 <= K) {                                  // full column tile: fixed trip count, unrolled
#pragma unroll 8
            for (int c = 0; c < 64; ++c) pair(c);
        } else {
            for (int c = 0; c < K - j0; ++c) pair(c);
        }
        mask[((size_t)it * W + jt) * 64 + lane] = i < K ? bits : 0ull;      // (rows >= K are never read)
    }
}

// Greedy sweep in steps of SW_S = 1024 boxes (SW_T = 16 mask words).  The serial part of NMS is cheap when the
// bits are on chip (a scalar bit test per suppressed box, ~30 cycles per kept one); what made a tile-by-tile sweep
// slow is one dependent HBM round trip per 64 boxes.  Per step: `nms_resolve_kernel` (ONE block) stages the
// 1024 x 1024 diagonal bit block in LDS (coalesced: 16 lanes per row) and one wave walks the boxes in order against
// the running `removed` words, all decisions in scalar registers; `nms_propagate_kernel` (many blocks) then ORs the
// mask rows of the step's kept boxes into the `removed` words of all later boxes.
constexpr int SW_S = 1024, SW_T = 16, SW_LD = 17;          // LDS row stride in words (odd: conflict-free)

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(1024) void nms_resolve_kernel(const unsigned long long* __restrict__ mask,
                                                           const int32_t* __restrict__ sorted_idx, int K, int W, int step,
                                                           unsigned long long* __restrict__ removed,
                                                           unsigned long long* __restrict__ kept_bits,
                                                           int32_t* __restrict__ keep_out, int32_t* __restrict__ d_num_keep) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long diag[];      // [SW_S][SW_LD]
    __shared__ unsigned long long s_kept[SW_T];
    const int tid = threadIdx.x, lane = tid & 63;
    const int t0 = step * SW_T, r0 = step * SW_S;
    const int nr = min(SW_S, K - r0), nt = min(SW_T, W - t0);
#pragma unroll 4
    for (int ti = 0; ti < SW_T; ++ti) {                        // a wave reads one (row tile, word) pair: 512 B contiguous
        const int w = tid >> 6, row = ti * 64 + lane;
        unsigned long long v = 0ull;
        if (row < nr && w < nt && w >= ti) v = mask[((size_t)(t0 + ti) * W + t0 + w) * 64 + lane];
        diag[row * SW_LD + w] = v;
    }
    __syncthreads();
    if (tid < 64) {
        const int l16 = lane & 15;
        unsigned long long rem = (lane < nt) ? removed[t0 + lane] : ~0ull;       // lanes 0..15: this step's words
        unsigned long long kept = 0ull;
        for (int wd = 0; wd < nt; ++wd) {
            unsigned long long cur = readlane64(rem, wd);                          // scalar: the 64 boxes being walked
            const int nb = min(64, nr - wd * 64);
            unsigned long long kw = 0ull;
            const unsigned long long* rowp = diag + (wd * 64) * SW_LD + l16;
            const unsigned long long valid = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
            unsigned long long alive = ~cur & valid;                                // scalar; suppressed boxes cost nothing
            while (alive) {
                const int bit = __ffsll((long long)alive) - 1;                      // next box that survived so far: kept
                kw |= 1ull << bit;
                const unsigned long long row = rowp[bit * SW_LD];
                rem |= row;
                cur |= readlane64(row, wd);
                alive = ~cur & valid & ~((2ull << bit) - 1ull);                     // later boxes of this word still alive
            }
            if (lane == wd) kept = kw;
        }
        if (lane < SW_T) s_kept[lane] = lane < nt ? kept : 0ull;
        if (lane < nt) kept_bits[t0 + lane] = kept;
    }
    __syncthreads();
    // kept boxes -> keep_out, in order
    const int base = *d_num_keep;
    int before = 0, total = 0;
    for (int w = 0; w < SW_T; ++w) {
        const int c = __popcll(s_kept[w]);
        if (w < (tid >> 6)) before += c;
        total += c;
    }
    const unsigned long long mine = s_kept[tid >> 6];
    if (tid < nr && ((mine >> lane) & 1ull))
        keep_out[base + before + __popcll(mine & ((1ull << lane) - 1ull))] = sorted_idx[r0 + tid];
    __syncthreads();
    if (tid == 0) *d_num_keep = base + total;
}

// block = (64 later words) x (4 row groups); blockIdx.y = one of the step's 16 kept words: the rows of that word's kept
// boxes are OR-ed by 4 threads per removed word (<= 16 rows each, all loads independent) and merged with one atomicOr.
__global__ __launch_bounds__(256) void nms_propagate_kernel(const unsigned long long* __restrict__ mask, int K, int W,
                                                            int step, const unsigned long long* __restrict__ kept_bits,
                                                            unsigned long long* __restrict__ removed) {
    const int t0 = step * SW_T;
    const int kwi = blockIdx.y;
    if (t0 + kwi >= W) return;
    const unsigned long long kb = kept_bits[t0 + kwi];                             // uniform
    if (kb == 0ull) return;
    const int w = t0 + SW_T + blockIdx.x * 64 + (threadIdx.x & 63);
    const int grp = threadIdx.x >> 6;                                               // rows 16 grp .. 16 grp + 15 of the word
    unsigned long long sub = (kb >> (16 * grp)) & 0xffffull;
    if (w >= W || sub == 0ull) return;
    // tile-major mask: the 16 rows of this group are 128 contiguous bytes of the (row tile, word) pair
    const unsigned long long* rowbase = mask + ((size_t)(t0 + kwi) * W + w) * 64 + 16 * grp;
    unsigned long long acc = 0ull;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if ((sub >> r) & 1ull) acc |= rowbase[r];
    if (acc) atomicOr(removed + w, acc);
}

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// ---- two-phase NMS for large candidate sets --------------------------------------------------------------------
// Greedy NMS only ever compares a box with KEPT boxes of higher score.  With tens of thousands of candidates almost all
// of the K^2/2 pairs of the full bit matrix involve a box that is itself suppressed (46 706 random-initialised
// candidates: 4 052 kept, 1.09e9 pairs, 1.1 ms).  The host wrapper (ops_detect.nms) therefore runs the matrix NMS on a
// HEAD of the sorted candidates, tests the whole TAIL against the boxes the head kept (mi355_nms_filter_tail: tail x
// kept pairs only), compacts the survivors in order, and repeats on them: exact, because a tail box that a head-kept box
// suppresses is never kept and so never suppresses anything itself.
__global__ __launch_bounds__(256) void nms_filter_kernel(const float4* __restrict__ boxes, const int32_t* __restrict__ keep,
                                                         const int32_t* __restrict__ d_nk, int kept_begin,
                                                         const float4* __restrict__ tail, int Kt, float thr,
                                                         unsigned long long* __restrict__ surv_words) {
    __shared__ unsigned long long s_rem[4];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = blockIdx.x * 64 + lane;
    const float4 a = t < Kt ? tail[t] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float area_a = (a.z - a.x) * (a.w - a.y);
    const float lo = thr * (1.0f - 1e-5f), hi = thr * (1.0f + 1e-5f);
    const int nk = *d_nk;
    bool removed = false;
    for (int j0 = kept_begin + wv * 64; j0 < nk; j0 += 256) {
        const int j = j0 + lane;
        const float4 mine = j < nk ? boxes[keep[j]] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float area_m = (mine.z - mine.x) * (mine.w - mine.y);
        const int n = min(64, nk - j0);
        for (int c = 0; c < n; ++c) {
            const float bx = bcast(mine.x, c), by = bcast(mine.y, c), bz = bcast(mine.z, c), bw = bcast(mine.w, c);
            removed |= iou_over(a, area_a, bx, by, bz, bw, bcast(area_m, c), thr, lo, hi);
        }
    }
    const unsigned long long r = __ballot(removed);
    if (lane == 0) s_rem[wv] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long rem = (s_rem[0] | s_rem[1]) | (s_rem[2] | s_rem[3]);
        const int nb = min(64, Kt - blockIdx.x * 64);
        const unsigned long long valid = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
        surv_words[blockIdx.x] = ~rem & valid;
    }
}

// one block: exclusive prefix sum of the survivors per 64-box word -> word_off[w], total -> *d_count
__global__ __launch_bounds__(1024) void nms_survivor_scan_kernel(const unsigned long long* __restrict__ surv_words, int Wt,
                                                                 int32_t* __restrict__ word_off, int32_t* __restrict__ d_count) {
    __shared__ int s_sum[1024];
    const int tid = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < Wt; base += 1024) {
        const int w = base + tid;
        const int c = w < Wt ? __popcll(surv_words[w]) : 0;
        s_sum[tid] = c;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {                 // Hillis-Steele inclusive scan
            const int v = tid >= o ? s_sum[tid - o] : 0;
            __syncthreads();
            s_sum[tid] += v;
            __syncthreads();
        }
        if (w < Wt) word_off[w] = carry + s_sum[tid] - c;
        const int tot = s_sum[1023];
        __syncthreads();
        carry += tot;
    }
    if (tid == 0) *d_count = carry;
}

__global__ void nms_survivor_scatter_kernel(const unsigned long long* __restrict__ surv_words,
                                            const int32_t* __restrict__ word_off, const float4* __restrict__ tail,
                                            const int32_t* __restrict__ tail_idx, int Kt, float4* __restrict__ out_boxes,
                                            int32_t* __restrict__ out_idx) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Kt) return;
    const unsigned long long w = surv_words[t >> 6];
    const int b = t & 63;
    if (!((w >> b) & 1ull)) return;
    const int pos = word_off[t >> 6] + __popcll(w & ((1ull << b) - 1ull));
    out_boxes[pos] = tail[t];
    out_idx[pos] = tail_idx[t];
}

extern "C" int mi355_nms_filter_tail(const float* boxes, const int32_t* keep, const int32_t* d_num_keep, int kept_begin,
                                     const float* tail_boxes, const int32_t* tail_idx, int Kt, float iou_thr,
                                     float* out_boxes, int32_t* out_idx, int32_t* d_out_count, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
    MI355_REQUIRE(boxes && keep && d_num_keep && kept_begin >= 0 && tail_boxes && tail_idx && Kt > 0 && out_boxes && out_idx &&
                  d_out_count && workspace, "nms_filter_tail: bad args");
    MI355_REQUIRE(((uintptr_t)boxes & 15) == 0 && ((uintptr_t)tail_boxes & 15) == 0 && ((uintptr_t)out_boxes & 15) == 0 &&
                  ((uintptr_t)workspace & 15) == 0, "nms_filter_tail: alignment");
    const int Wt = (int)cdiv(Kt, 64);
    MI355_REQUIRE(workspace_bytes >= (int64_t)Wt * 12 + 64, "nms_filter_tail: workspace %lld < %lld",
                  (long long)workspace_bytes, (long long)Wt * 12 + 64);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* surv = (unsigned long long*)workspace;
    int32_t* word_off = (int32_t*)(surv + Wt);
    hipLaunchKernelGGL(nms_filter_kernel, dim3((unsigned)Wt), dim3(256), 0, st, (const float4*)boxes, keep, d_num_keep,
                       kept_begin, (const float4*)tail_boxes, Kt, iou_thr, surv);
    hipLaunchKernelGGL(nms_survivor_scan_kernel, dim3(1), dim3(1024), 0, st, surv, Wt, word_off, d_out_count);
    hipLaunchKernelGGL(nms_survivor_scatter_kernel, dim3((unsigned)cdiv(Kt, 256)), dim3(256), 0, st, surv, word_off,
                       (const float4*)tail_boxes, tail_idx, Kt, (float4*)out_boxes, out_idx);
    return check_launch("nms_filter_tail");
}

extern "C" int mi355_nms_workspace_bytes(int K, int64_t* bytes) {
    MI355_REQUIRE(K >= 0 && bytes, "nms_workspace_bytes: bad args");
    int64_t K2 = next_pow2(K < SORT_CHUNK ? SORT_CHUNK : K);
    int64_t W = cdiv(K, 64);
    *bytes = K2 * 8 + (int64_t)K * 16 + (int64_t)K * 4 + 64 + W * W * 64 * 8 + 256 + 2 * (W * 8 + 64);
    return MI355_OK;
}

extern "C" int mi355_nms_sort(const float* boxes, const float* scores, int K, float* sorted_boxes, int32_t* sorted_idx,
                              void* workspace, int64_t workspace_bytes, void* stream) {
    MI355_REQUIRE(boxes && scores && K > 0 && sorted_boxes && sorted_idx && workspace, "nms_sort: bad args");
    MI355_REQUIRE(((uintptr_t)boxes & 15) == 0 && ((uintptr_t)sorted_boxes & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
                  "nms_sort: alignment");
    const int K2 = next_pow2(K < SORT_CHUNK ? SORT_CHUNK : K);
    MI355_REQUIRE(workspace_bytes >= (int64_t)K2 * 8, "nms_sort: workspace %lld < %lld", (long long)workspace_bytes,
                  (long long)K2 * 8);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* keys = (unsigned long long*)workspace;
    hipLaunchKernelGGL(nms_make_keys_kernel, dim3(K2 / 256), dim3(256), 0, st, scores, K, K2, keys);
    hipLaunchKernelGGL(bitonic_chunk_sort_kernel, dim3(K2 / SORT_CHUNK), dim3(256), 0, st, keys);
    for (int k = SORT_CHUNK * 2; k <= K2; k <<= 1) {
        for (int j = k >> 1; j >= SORT_CHUNK; j >>= 1)
            hipLaunchKernelGGL(bitonic_global_step_kernel, dim3(K2 / 2 / 256), dim3(256), 0, st, keys, K2, k, j);
        hipLaunchKernelGGL(bitonic_chunk_merge_kernel, dim3(K2 / SORT_CHUNK), dim3(256), 0, st, keys, k);
    }
    hipLaunchKernelGGL(nms_gather_boxes_kernel, dim3((unsigned)cdiv(K, 256)), dim3(256), 0, st, keys,
                       (const float4*)boxes, K, (float4*)sorted_boxes, sorted_idx);
    return check_launch("nms_sort");
}

extern "C" int mi355_nms_presorted_workspace_bytes(int K, int64_t* bytes) {
    MI355_REQUIRE(K >= 0 && bytes, "nms_presorted_workspace_bytes: bad args");
    const int64_t W = cdiv(K, 64);
    *bytes = W * W * 64 * 8 + 256 + 2 * (W * 8 + 64);
    return MI355_OK;
}

// matrix NMS over boxes that are already in descending score order; the kept ORIGINAL indices (sorted_idx) are appended
// to keep_out behind *d_num_keep (append != 0) or written from 0
extern "C" int mi355_nms_presorted(const float* sorted_boxes, const int32_t* sorted_idx, int K, float iou_thr,
                                   int32_t* keep_out, int32_t* d_num_keep, int append, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
    MI355_REQUIRE(sorted_boxes && sorted_idx && K > 0 && keep_out && d_num_keep && workspace, "nms_presorted: bad args");
    MI355_REQUIRE(((uintptr_t)sorted_boxes & 15) == 0 && ((uintptr_t)workspace & 15) == 0, "nms_presorted: alignment");
    int64_t need;
    mi355_nms_presorted_workspace_bytes(K, &need);
    if (workspace_bytes < need) return fail(MI355_ENOSPC, "nms_presorted: workspace %lld < %lld",
                                            (long long)workspace_bytes, (long long)need);
    hipStream_t st = (hipStream_t)stream;
    const int W = (int)cdiv(K, 64);
    char* p = (char*)workspace;
    unsigned long long* mask = (unsigned long long*)p;              p += (int64_t)W * W * 64 * 8;
    unsigned long long* removed = (unsigned long long*)p;           p += (((int64_t)W * 8 + 63) / 64) * 64;
    unsigned long long* kept_bits = (unsigned long long*)p;
    const float4* sboxes = (const float4*)sorted_boxes;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(W, 4), dim3(256), 0, st, sboxes, K, W, iou_thr, mask);
    if (hipMemsetAsync(removed, 0, (size_t)W * 8, st) != hipSuccess ||
        (!append && hipMemsetAsync(d_num_keep, 0, 4, st) != hipSuccess))
        return fail(MI355_ELAUNCH, "nms: memset failed");
    static bool attr_set = false;
    constexpr size_t DIAG_BYTES = (size_t)SW_S * SW_LD * 8;           // 136 KiB of the 160 KiB LDS
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)nms_resolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DIAG_BYTES);
        attr_set = true;
    }
    const int nsteps = (int)cdiv(K, SW_S);
    for (int step = 0; step < nsteps; ++step) {
        hipLaunchKernelGGL(nms_resolve_kernel, dim3(1), dim3(1024), DIAG_BYTES, st, mask, sorted_idx, K, W, step, removed,
                           kept_bits, keep_out, d_num_keep);
        const int rest = W - (step + 1) * SW_T;
        if (rest > 0)
            hipLaunchKernelGGL(nms_propagate_kernel, dim3((unsigned)cdiv(rest, 64), SW_T), dim3(256), 0, st, mask, K, W, step,
                               kept_bits, removed);
    }
    return check_launch("nms");
}

extern "C" int mi355_nms(const float* boxes, const float* scores, int K, float iou_thr,
                         int32_t* keep_out, int32_t* d_num_keep, void* workspace,
                         int64_t workspace_bytes, void* stream) {
    MI355_REQUIRE(keep_out && d_num_keep && K >= 0, "nms: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (K == 0) {
        hipError_t e = hipMemsetAsync(d_num_keep, 0, 4, st);
        return e == hipSuccess ? MI355_OK : fail(MI355_ELAUNCH, "nms: memset failed");
    }
    MI355_REQUIRE(boxes && scores && workspace, "nms: null pointer");
    MI355_REQUIRE(((uintptr_t)boxes & 15) == 0 && ((uintptr_t)workspace & 255) == 0, "nms: alignment");
    int64_t need;
    mi355_nms_workspace_bytes(K, &need);
    if (workspace_bytes < need) return fail(MI355_ENOSPC, "nms: workspace %lld < %lld",
                                            (long long)workspace_bytes, (long long)need);
    const int K2 = next_pow2(K < SORT_CHUNK ? SORT_CHUNK : K);
    char* p = (char*)workspace;
    void* keys = p;                                     p += (int64_t)K2 * 8;
    float* sboxes = (float*)p;                          p += (int64_t)K * 16;
    int32_t* sidx = (int32_t*)p;                        p += (((int64_t)K * 4 + 63) / 64) * 64;
    const int rc = mi355_nms_sort(boxes, scores, K, sboxes, sidx, keys, (int64_t)K2 * 8, stream);
    if (rc) return rc;
    return mi355_nms_presorted(sboxes, sidx, K, iou_thr, keep_out, d_num_keep, 0, p, workspace_bytes - (p - (char*)workspace),
                               stream);
}
