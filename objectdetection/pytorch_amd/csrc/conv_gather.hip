// Grouped "gather conv" on gfx950 MFMA: forward convolution, data gradient and the five-level
// shared-weight head convs as one implicit-GEMM kernel.
//
//   out[m][k] = act( sum_{tap,c} in[m @ tap][c] * w[k][tap][c] + bias[k] + addend[m][k] ) * (mask>0)
//
// GEMM view: M = N*P*Q output pixels, N = K output channels, reduction = taps x C, both operands
// reduction-contiguous (NHWC activations, [K][tap][C] weights) = the natural bf16 MFMA layout.
// Tile 128x128x64, 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 accumulators.
// Operands are staged global->LDS with buffer_load_dwordx4 ... lds (LDS-DMA, no VGPR round
// trip); the per-lane SOURCE address does the im2col gather, out-of-range requests return
// zeros (padding / tails); the LDS image is lane-linear with the XOR swizzle applied on the
// source chunk and on the ds_read_b128 address (conflict-free for the b128 lane groups).
// Double-buffered: the DMA of k-step t+1 is in flight while k-step t runs on the matrix cores.
// Epilogue goes through LDS so that bias/addend/mask reads and the output stores are 16-B
// coalesced rows.
//
// Replaces: nn.Conv2d (+ frozen BatchNorm2d, ReLU, residual add, sigmoid, upsample-add) call
// chains of architectures/retinanet.py:48-69, :92-115, :142-178, :272-287 and
// architectures/utils_resnet.py:25-41, :60-80, and their autograd data-gradients.
#include "conv_common.h"

using namespace mi355;

namespace {

constexpr int BN = 128, BK = 64;
constexpr int EPI_LD = BN + 4;                           // fp32 staging row stride (floats): 64-row halves
constexpr int EPI_BYTES = 64 * EPI_LD * 4;               // 33792 B (the bf16 staging, 128 rows of BN + 4 elements, is the same size)

struct ProbDev {
    const unsigned short* in;
    void* out;
    const unsigned short* addend;
    const unsigned short* mask;
    const unsigned short* w;      // this problem's weights / bias (the launch's unless the problem overrides them)
    const float* bias;
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
    int in_bytes;
    FastDiv dPQ, dQ;
};

struct LaunchDev {
    ProbDev p[MI355_CONV_MAX_PROBLEMS];
    const unsigned short* w;
    const float* bias;
    int nprob, K, T, act, out_f32, n_tiles, w_bytes, flags;
    int ksplit, tiles_total;      // split-K: blockIdx = split * tiles_total + tile
    long long split_stride;       // floats between the partial-output slabs of consecutive splits
    float* stats;                 // per-tile column statistics of the stored output (Mi355ConvLaunch.stats) or null
};

__device__ __forceinline__ float activate(float v, int act) {
    if (act == 1) return fmaxf(v, 0.0f);
    // sigmoid over 63 M class scores per step: v_exp_f32 + v_rcp_f32 (1-2 ulp) instead of expf + IEEE division
    if (act == 2) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v));
    return v;
}

// Accumulators hold the TRANSPOSED tile: acc[i][j][r] = out[pixel i*16+fr][channel j*16+fq*4+r]
// (weights are the MFMA A operand), so a lane owns 4 consecutive channels of one pixel and the
// epilogue stages with 8/16-B LDS writes.  The bias is the accumulator's initial value.
template <int TA = 4>
__device__ __forceinline__ void init_acc(f4v (&acc)[TA][4], const LaunchDev& L, const float* __restrict__ bias, int n0, int wn, int fq) {
    {
        f4v b4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b4[j] = (f4v){0.f, 0.f, 0.f, 0.f};
        if (bias) {                                  // uniform branch; 4 loads in flight together
            if ((L.K & 3) == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    b4[j] = *reinterpret_cast<const f4v*>(bias + min(n0 + wn * 64 + j * 16 + fq * 4, L.K - 4));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        b4[j][r] = bias[min(n0 + wn * 64 + j * 16 + fq * 4 + r, L.K - 1)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    b4[j][r] = (n0 + wn * 64 + j * 16 + fq * 4 + r) < L.K ? b4[j][r] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < TA; ++i) acc[i][j] = b4[j];
    }
}

// Epilogue through LDS: bias is already in the accumulators; addend / activation / ReLU-backward mask,
// 16-B coalesced row stores.
// TA = 16-row accumulator fragments per wave: 4 (64x64 per wave, 2 wave columns, 128-wide tile) or 2
// (32x64 per wave, 4 wave rows, 64-wide tile: K <= 64 layers without empty wave columns).
// ST: the launch also leaves per-tile column statistics of its output (Mi355ConvLaunch.stats).  A template parameter, not a
// run-time branch: with the branch compiled in, the narrow generic kernel needs 103 instead of 94 VGPRs and loses its fifth
// block per CU (-2 % on the frozen-BN R50 step, A/B of two builds in one call).
template <int WM, int TA = 4, bool ST = false>
__device__ __forceinline__ void gather_epilogue(f4v (&acc)[TA][4], const LaunchDev& L, const ProbDev& P,
                                                unsigned char* smem, const long long* s_outoff,
                                                const long long* s_addoff, int n0, int m_tile) {
    constexpr int NTHREADS = WM * 128, BM = WM * 64, BNT = TA == 4 ? 128 : 64;
    constexpr int CT = BNT / 8, RP = NTHREADS / CT;    // threads per output row, rows per read pass
    constexpr int LD16 = BNT + 4, LD32 = BNT + 4;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wrow0 = TA == 4 ? (wv >> 1) * 64 : wv * 32, wn = TA == 4 ? (wv & 1) : 0;
    const int fr = lane & 15, fq = lane >> 4;
    const int col = (tid % CT) * 8;
    const int k0 = n0 + col;
    const int nvalid = min(8, L.K - k0);
    const bool full = nvalid == 8;
    const bool mask_first = (L.flags & 1) != 0;
    const bool fast = !L.out_f32 && !P.addend && (L.K & 7) == 0;
    if (fast) {
        // FAST: activation in registers, bf16 staging of the whole tile (one barrier), the read
        // side is a 16-B copy (+ the ReLU-backward mask as a bit select).
        unsigned short* st16 = reinterpret_cast<unsigned short*>(smem);
        constexpr int NP = BM / RP;
        long long oo[NP];
        s8v mk[NP];
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {      // mask reads first: latency hides under staging
            const int row = pass * RP + tid / CT;
            oo[pass] = k0 < L.K ? s_outoff[row] : -1;
            mk[pass] = (s8v){0, 0, 0, 0, 0, 0, 0, 0};
            if (P.mask && oo[pass] >= 0) mk[pass] = *reinterpret_cast<const s8v*>(P.mask + oo[pass] + k0);
        }
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s4v pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = (short)f32_to_bf16_rn(activate(acc[i][j][r], L.act));
                *reinterpret_cast<s4v*>(st16 + (wrow0 + i * 16 + fr) * LD16 + wn * 64 + j * 16 + fq * 4) = pk;
            }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            if (oo[pass] < 0) continue;
            const int row = pass * RP + tid / CT;
            const s4v lo = *reinterpret_cast<const s4v*>(st16 + row * LD16 + col);
            const s4v hi = *reinterpret_cast<const s4v*>(st16 + row * LD16 + col + 4);
            s8v o = (s8v){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if (P.mask) {
#pragma unroll
                for (int e = 0; e < 8; ++e)       // bf16 > 0  <=>  sign clear and not +-0
                    o[e] = ((unsigned short)mk[pass][e] & 0x8000u) || !((unsigned short)mk[pass][e] & 0x7fffu) ? (short)0 : o[e];
            }
            *reinterpret_cast<s8v*>(reinterpret_cast<unsigned short*>(P.out) + oo[pass] + k0) = o;
        }
        if constexpr (ST) {
            // Column statistics of the tile for a train-mode BatchNorm behind this conv: sum and square-sum of the
            // STORED (bf16-rounded) values, rows past M left out; a thread adds its rows in row order, the RP row threads
            // of a column are added in thread order by one thread per column: a fixed order (bit-reproducible).
            float fs[8], fq[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { fs[e] = 0.0f; fq[e] = 0.0f; }
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                if (oo[pass] < 0) continue;
                const int row = pass * RP + tid / CT;
                const s4v lo = *reinterpret_cast<const s4v*>(st16 + row * LD16 + col);
                const s4v hi = *reinterpret_cast<const s4v*>(st16 + row * LD16 + col + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = bf16_to_f32((unsigned short)lo[e]), b = bf16_to_f32((unsigned short)hi[e]);
                    fs[e] += a; fq[e] += a * a; fs[e + 4] += b; fq[e + 4] += b * b;
                }
            }
            __syncthreads();                                   // the staged tile has been read out
            float* red = reinterpret_cast<float*>(smem);       // [RP][2][BNT]
            const int rt = tid / CT;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(rt * 2 + 0) * BNT + col + e] = fs[e];
                red[(rt * 2 + 1) * BNT + col + e] = fq[e];
            }
            __syncthreads();
            for (int t = tid; t < 2 * BNT; t += NTHREADS) {
                const int which = t / BNT, c = t - which * BNT;
                if (n0 + c >= L.K) continue;
                float a = 0.0f;
                for (int r = 0; r < RP; ++r) a += red[(r * 2 + which) * BNT + c];
                L.stats[((long long)m_tile * 2 + which) * L.K + n0 + c] = a;
            }
        }
        return;
    }
    // GENERAL: fp32 staging in two 64-row halves (addend / fp32 output / ragged channel tails)
    float* epi = reinterpret_cast<float*>(smem);
    constexpr int PH = 64 / RP;                        // read passes per 64-row slice
    for (int half = 0; half < WM; ++half) {
        long long oo[PH];
        s8v mk[PH], ad[PH];
#pragma unroll
        for (int pass = 0; pass < PH; ++pass) {
            const int row = half * 64 + pass * RP + tid / CT;
            oo[pass] = k0 < L.K ? s_outoff[row] : -1;
            mk[pass] = (s8v){0, 0, 0, 0, 0, 0, 0, 0};
            ad[pass] = (s8v){0, 0, 0, 0, 0, 0, 0, 0};
            if (oo[pass] >= 0 && full) {
                if (P.mask) mk[pass] = *reinterpret_cast<const s8v*>(P.mask + oo[pass] + k0);
                if (P.addend) ad[pass] = *reinterpret_cast<const s8v*>(P.addend + s_addoff[row] + k0);
            }
        }
        if (wrow0 / 64 == half) {
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<f4v*>(epi + (wrow0 % 64 + i * 16 + fr) * LD32 + wn * 64 + j * 16 + fq * 4) = acc[i][j];
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < PH; ++pass) {
            if (oo[pass] < 0) continue;
            const int hrow = pass * RP + tid / CT;
            const f4v v0 = *reinterpret_cast<const f4v*>(epi + hrow * LD32 + col);
            const f4v v1 = *reinterpret_cast<const f4v*>(epi + hrow * LD32 + col + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (full) {
                if (P.mask && mask_first) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = bf16_to_f32((unsigned short)mk[pass][e]) > 0.0f ? v[e] : 0.0f;
                }
                if (P.addend) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bf16_to_f32((unsigned short)ad[pass][e]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = activate(v[e], L.act);
                if (P.mask && !mask_first) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = bf16_to_f32((unsigned short)mk[pass][e]) > 0.0f ? v[e] : 0.0f;
                }
                if (L.out_f32) {
                    float* op = reinterpret_cast<float*>(P.out) + oo[pass] + k0;
                    *reinterpret_cast<f4v*>(op) = (f4v){v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f4v*>(op + 4) = (f4v){v[4], v[5], v[6], v[7]};
                } else {
                    s8v o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (short)f32_to_bf16_rn(v[e]);
                    *reinterpret_cast<s8v*>(reinterpret_cast<unsigned short*>(P.out) + oo[pass] + k0) = o;
                }
            } else {                               // ragged channel tail (K % 8 != 0): scalar path
                const int row = half * 64 + hrow;
                const unsigned short* mp = P.mask ? P.mask + oo[pass] + k0 : nullptr;
                const unsigned short* ap = P.addend ? P.addend + s_addoff[row] + k0 : nullptr;
                for (int e = 0; e < nvalid; ++e) {
                    float x = v[e];
                    if (mp && mask_first) x = bf16_to_f32(mp[e]) > 0.0f ? x : 0.0f;
                    if (ap) x += bf16_to_f32(ap[e]);
                    x = activate(x, L.act);
                    if (mp && !mask_first) x = bf16_to_f32(mp[e]) > 0.0f ? x : 0.0f;
                    if (L.out_f32) (reinterpret_cast<float*>(P.out) + oo[pass] + k0)[e] = x;
                    else (reinterpret_cast<unsigned short*>(P.out) + oo[pass] + k0)[e] = f32_to_bf16_rn(x);
                }
            }
        }
        if (half + 1 < WM) __syncthreads();
    }
}

// WM = wave rows of 64 output pixels (WM 2: 128-row tile, 4 waves).  Variants, chosen per launch on the host:
// <STAGES 2, TA 4>: 128x128, double-buffered k-loop, 64 KiB LDS, 2 blocks/CU (split-K launches).
// <STAGES 1, TA 4>: 128x128, one stage, 33 KiB LDS, 3 blocks/CU: resident waves instead of a second stage hide
//                   the LDS-DMA latency (long many-tile reductions that the row-of-taps kernel cannot take).
// <.., TA 2>:       128x64 tile, four wave rows of 32x64 accumulators (half the registers: up to 5 blocks per
//                   CU): everything else - HBM-bound 1x1 convs, layers with K <= 64, small launches.
// (Measured and dropped this round: a 256x128 tile with a 3-stage counted-vmcnt ring at one block per CU and
//  a 4-stage 128x128 ring - both slower than more resident blocks.)
// TAP1: every problem of the launch has at most one tap (the 1x1 convs of the bottlenecks, forward and data gradient -
// a third of the R50 GPU time): the bounds test of a staged row does not change from k-step to k-step, so a thread keeps
// its four A offsets and its B offsets final up to the channel block and a DMA piece costs one add with a scalar
// instead of ~8 vector instructions (40 per k-step beside 16 MFMAs per wave: the SIMD's issue port, shared by up to five
// resident waves, paced these launches).
template <bool FLAT, int STAGES, int WM, int TA = 4, bool ST = false, bool TAP1 = false>
__global__ __launch_bounds__(WM * 128, TA == 2 ? 4 : 2) void conv_gather_kernel(const LaunchDev L) {
    static_assert(TA == 4 || WM == 2, "narrow tile: 4 waves");
    static_assert(!TAP1 || !FLAT, "single-tap form: channel-block k-steps");
    constexpr int NTHREADS = WM * 128, BM = WM * 64, BNT = TA == 4 ? 128 : 64;
    constexpr int A_BYTES = BM * BK * 2, STAGE_BYTES = A_BYTES + BNT * BK * 2;
    constexpr int NB = BNT * 8 / NTHREADS;            // B-tile DMA instructions per thread and stage
    constexpr int ROWSTEP = NTHREADS / 8;             // rows covered by one DMA instruction of the block
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ long long s_outoff[BM];
    __shared__ long long s_addoff[BM];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int split = L.ksplit > 1 ? blockIdx.x / L.tiles_total : 0;
    const int tile = L.ksplit > 1 ? blockIdx.x - split * L.tiles_total : xcd_swizzle(blockIdx.x, gridDim.x);
    const int m_tile = tile / L.n_tiles, n_tile = tile - m_tile * L.n_tiles;

    // ---- select the problem of this tile (scalar selects over the by-value descriptors)
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MI355_CONV_MAX_PROBLEMS; ++i)
        if (i < L.nprob && m_tile >= L.p[i].tile_begin) pi = i;
    ProbDev P = L.p[pi];                                 // one block copy out of the kernarg segment
    if (L.ksplit > 1) P.out = reinterpret_cast<float*>(P.out) + split * L.split_stride;
    const int m0 = (m_tile - P.tile_begin) * BM;
    const int n0 = n_tile * BNT;
    const int PQ = P.P * P.Q;

    // ---- per-row output / addend offsets (used by the epilogue)
    if (tid < BM) {
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

    // ---- staging geometry: thread owns LDS slot (row = tid/8 + 32 i, slot = tid%8), i = 0..3,
    //      and loads the global chunk kc = slot ^ (row & 7) into it.
    const int slot = tid & 7;
    const int kc = slot ^ ((tid >> 3) & 7);
    int a_base[4], a_ih0[4], a_iw0[4];
    unsigned b_rowoff[NB];
    const int wrow = L.T * P.C;                       // elements per weight row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + ROWSTEP * i;
        const int m = m0 + row;
        {
            const int mc = min(m, P.M - 1);
            const int n = (int)fdiv((unsigned)mc, P.dPQ), rem = mc - n * PQ;
            const int p = (int)fdiv((unsigned)rem, P.dQ), q = rem - p * P.Q;
            a_ih0[i] = m < P.M ? p * P.sh + P.dh0 : -0x40000000;   // tail rows: never valid
            a_iw0[i] = q * P.sw + P.dw0;
            a_base[i] = n * (int)P.in_sN + ((p * P.sh + P.dh0) * P.W + a_iw0[i]) * P.C;
        }
        if (i < NB) {
            const int nn = n0 + row;
            b_rowoff[i] = nn < L.K ? (unsigned)(nn * wrow) * 2u : OOB_OFFSET;
        }
    }
    if constexpr (TAP1) {                              // a_base / b_rowoff become the final byte offsets at channel block 0, or OOB
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = (unsigned)a_ih0[i] < (unsigned)P.H && (unsigned)a_iw0[i] < (unsigned)P.W;
            a_base[i] = ok ? (int)((unsigned)(a_base[i] + kc * 8) * 2u) : (int)OOB_OFFSET;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (b_rowoff[i] != OOB_OFFSET) b_rowoff[i] += (unsigned)((P.r0 * P.S + P.s0) * P.C + kc * 8) * 2u;
    }
    const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(P.in, (unsigned)P.in_bytes);
    const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(P.w, (unsigned)L.w_bytes);

    const int ntaps = P.TR * P.TS;
    const int cpt = P.C >> 3;                          // 16-B chunks per tap
    const int ksteps_all = FLAT ? (ntaps * cpt + 7) / 8 : ntaps * (P.C / BK);
    // split-K: this block reduces k-steps [ks_begin, ks_begin + ksteps) into slab `split` of the workspace
    const int ks_begin = L.ksplit > 1 ? (int)((long long)ksteps_all * split / L.ksplit) : 0;
    const int ksteps = (L.ksplit > 1 ? (int)((long long)ksteps_all * (split + 1) / L.ksplit) : ksteps_all) - ks_begin;

    // k-steps are issued strictly in order, so the (tap row, tap column, channel block) of the next
    // stage is carried as wave-uniform counters instead of being decoded from ks by division.
    int nx_ti = 0, nx_tj = 0, nx_c = 0;
    if (!FLAT && ks_begin > 0) {
        const int cb = P.C / BK, tap = ks_begin / cb;
        nx_c = (ks_begin - tap * cb) * BK;
        nx_ti = tap / P.TS;
        nx_tj = tap - nx_ti * P.TS;
    }
    auto issue = [&](int ks, int buf) {
        unsigned char* sa = smem + buf * STAGE_BYTES;
        unsigned char* sb = sa + A_BYTES;
        if constexpr (TAP1) {                               // (OOB_OFFSET + a channel offset stays out of range)
            const unsigned cd = (unsigned)nx_c * 2u;       // wave-uniform
            nx_c += BK;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (LDS_PTR(void))(sa + (i * NTHREADS + wv * 64) * 16),
                                                         16, (unsigned)a_base[i] + cd, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                // (the offset through a named temporary: with `b_rowoff[i] + cd` as the argument the HOST pass of hipcc 7.2
                //  drops this instantiation without a diagnostic and the library fails to load with an undefined stub)
                const unsigned off = b_rowoff[i] + cd;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (LDS_PTR(void))(sb + (i * NTHREADS + wv * 64) * 16),
                                                         16, off, 0, 0, 0);
            }
        } else {
        int ti, tj, c0;
        bool tap_ok = true;
        if (FLAT) {
            const int J = (ks_begin + ks) * 8 + kc;
            const int tap = J / cpt;
            c0 = (J - tap * cpt) * 8;
            ti = tap / P.TS;
            tj = tap - ti * P.TS;
            tap_ok = tap < ntaps;
        } else {
            ti = nx_ti;
            tj = nx_tj;
            c0 = nx_c + kc * 8;
            nx_c += BK;
            if (nx_c == P.C) {
                nx_c = 0;
                if (++nx_tj == P.TS) { nx_tj = 0; ++nx_ti; }
            }
        }
        const int dh = ti * P.dhs, dw = tj * P.dws;
        const int a_delta = (dh * P.W + dw) * P.C + c0;
        const unsigned b_delta = (unsigned)(((P.r0 + ti * P.rs) * P.S + (P.s0 + tj * P.ss)) * P.C + c0) * 2u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = tap_ok && (unsigned)(a_ih0[i] + dh) < (unsigned)P.H &&
                            (unsigned)(a_iw0[i] + dw) < (unsigned)P.W;
            const unsigned off = ok ? (unsigned)(a_base[i] + a_delta) * 2u : OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (LDS_PTR(void))(sa + (i * NTHREADS + wv * 64) * 16),
                                                     16, off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const unsigned off = (tap_ok && b_rowoff[i] != OOB_OFFSET) ? b_rowoff[i] + b_delta : OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (LDS_PTR(void))(sb + (i * NTHREADS + wv * 64) * 16),
                                                     16, off, 0, 0, 0);
        }
        }
    };

    const int wrow0 = TA == 4 ? (wv >> 1) * 64 : wv * 32, wn = TA == 4 ? (wv & 1) : 0;
    const int fr = lane & 15, fq = lane >> 4;
    f4v acc[TA][4];
    // ds_read_b128 byte offsets of this lane's fragments inside a stage (row r, chunk c):
    //   r*128 + ((c ^ (r & 7)) << 4); r & 7 == fr & 7 for every 16-row tile.
    const int a_rd = (wrow0 + fr) * 128;
    const int b_rd = A_BYTES + (wn * 64 + fr) * 128;
    const int sw0 = ((fq) ^ (fr & 7)) << 4;           // k-sub 0: chunk fq
    const int sw1 = ((4 + fq) ^ (fr & 7)) << 4;       // k-sub 1: chunk 4 + fq

    auto compute = [&](int buf) {
        const unsigned char* st = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int sw = sub ? sw1 : sw0;
            s8v af[TA], bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < TA) af[t] = *reinterpret_cast<const s8v*>(st + a_rd + t * 16 * 128 + sw);
                bf[t] = *reinterpret_cast<const s8v*>(st + b_rd + t * 16 * 128 + sw);
            }
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    if (ksteps > 0) issue(0, 0);
    // (bias loads are issued behind the first stage's DMA so that both latencies overlap)
    init_acc<TA>(acc, L, P.bias, n0, wn, fq);
    if (STAGES == 2) {
        __syncthreads();
        for (int ks = 0; ks < ksteps; ++ks) {
            const int buf = ks & 1;
            if (ks + 1 < ksteps) issue(ks + 1, buf ^ 1);
            compute(buf);
            __syncthreads();
        }
    } else {
        __syncthreads();
        for (int ks = 0; ks < ksteps; ++ks) {
            compute(0);
            __syncthreads();
            if (ks + 1 < ksteps) {
                issue(ks + 1, 0);
                __syncthreads();
            }
        }
    }

    gather_epilogue<WM, TA, ST>(acc, L, P, smem, s_outoff, s_addoff, n0, m0 / BM);
}


// ---------------------------------------------------------------------------------------------
// RING: the narrow tile (128 x 64, four wave rows of 32 x 64) with a THREE-stage LDS-DMA ring for the launches that
// cannot fill the chip with blocks (256-767 tiles: the 16^2 / 32^2 / 64^2 convs of ResNet layers 2-4 and the pyramid):
// with one or two blocks per CU nothing but the block's own pipeline hides the DMA latency, and the two-stage form
// exposes most of it every k-step (layer3 3x3: 0.8 us per k-step for 0.12 us of MFMA work).  Here the DMA of k-step
// t + 2 is issued while k-step t is on the matrix cores: one raw s_barrier per k-step, a counted s_waitcnt vmcnt that
// leaves the newest stage in flight, fragment reads as opaque asm (hipcc would otherwise order every visible LDS read
// behind ALL pending LDS-DMA with vmcnt(0): guide, "Pipelining across barriers").
//   iteration t:  vmcnt(pieces of stage t+1) -> s_barrier -> issue(t+2 -> buf (t+2)%3) -> read + MFMA buf t%3
// RAW: every thread waits for its own pieces of stage t, then the barrier, then anyone reads.  WAR: buf (t+2)%3 was read
// in iteration t-1; those ds_reads were retired (lgkmcnt(0)) before that wave's MFMAs, i.e. before it reached this barrier.
template <int OFF>
__device__ __forceinline__ s8v lds_b128(unsigned addr) {
    s8v v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

template <bool ST = false, bool TAP1 = false>          // TAP1: single-tap staging as in conv_gather_kernel
__global__ __launch_bounds__(256, 2) void conv_gather_ring_kernel(const LaunchDev L) {
    constexpr int NTHREADS = 256, BM = 128, BNT = 64, TA = 2, RS = 3;
    constexpr int A_BYTES = BM * BK * 2, STAGE_BYTES = A_BYTES + BNT * BK * 2;      // 24 KiB
    constexpr int NB = BNT * 8 / NTHREADS;                                            // 2
    constexpr int ROWSTEP = NTHREADS / 8;
    constexpr int PIECES = 4 + NB;                                                    // DMA instructions per thread and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ long long s_outoff[BM];
    __shared__ long long s_addoff[BM];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int m_tile = tile / L.n_tiles, n_tile = tile - m_tile * L.n_tiles;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MI355_CONV_MAX_PROBLEMS; ++i)
        if (i < L.nprob && m_tile >= L.p[i].tile_begin) pi = i;
    const ProbDev P = L.p[pi];
    const int m0 = (m_tile - P.tile_begin) * BM;
    const int n0 = n_tile * BNT;
    const int PQ = P.P * P.Q;

    if (tid < BM) {
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

    const int slot = tid & 7;
    const int kc = slot ^ ((tid >> 3) & 7);
    int a_base[4], a_ih0[4], a_iw0[4];
    unsigned b_rowoff[NB];
    const int wrow = L.T * P.C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + ROWSTEP * i;
        const int m = m0 + row;
        const int mc = min(m, P.M - 1);
        const int n = (int)fdiv((unsigned)mc, P.dPQ), rem = mc - n * PQ;
        const int p = (int)fdiv((unsigned)rem, P.dQ), q = rem - p * P.Q;
        a_ih0[i] = m < P.M ? p * P.sh + P.dh0 : -0x40000000;
        a_iw0[i] = q * P.sw + P.dw0;
        a_base[i] = n * (int)P.in_sN + ((p * P.sh + P.dh0) * P.W + a_iw0[i]) * P.C;
        if (i < NB) {
            const int nn = n0 + row;
            b_rowoff[i] = nn < L.K ? (unsigned)(nn * wrow) * 2u : OOB_OFFSET;
        }
    }
    if constexpr (TAP1) {                              // a_base / b_rowoff become the final byte offsets at channel block 0, or OOB
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = (unsigned)a_ih0[i] < (unsigned)P.H && (unsigned)a_iw0[i] < (unsigned)P.W;
            a_base[i] = ok ? (int)((unsigned)(a_base[i] + kc * 8) * 2u) : (int)OOB_OFFSET;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (b_rowoff[i] != OOB_OFFSET) b_rowoff[i] += (unsigned)((P.r0 * P.S + P.s0) * P.C + kc * 8) * 2u;
    }
    const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(P.in, (unsigned)P.in_bytes);
    const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(P.w, (unsigned)L.w_bytes);
    const int ksteps = P.TR * P.TS * (P.C / BK);

    int nx_ti = 0, nx_tj = 0, nx_c = 0;
    auto issue = [&](int buf) {                          // k-steps are issued strictly in order
        unsigned char* sa = smem + buf * STAGE_BYTES;
        unsigned char* sb = sa + A_BYTES;
        if constexpr (TAP1) {
            const unsigned cd = (unsigned)nx_c * 2u;       // wave-uniform
            nx_c += BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned off = (unsigned)a_base[i] + cd;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (LDS_PTR(void))(sa + (i * NTHREADS + wv * 64) * 16),
                                                         16, off, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const unsigned off = b_rowoff[i] + cd;     // (named temporary: see conv_gather_kernel)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (LDS_PTR(void))(sb + (i * NTHREADS + wv * 64) * 16),
                                                         16, off, 0, 0, 0);
            }
            return;
        }
        const int ti = nx_ti, tj = nx_tj, c0 = nx_c + kc * 8;
        nx_c += BK;
        if (nx_c == P.C) {
            nx_c = 0;
            if (++nx_tj == P.TS) { nx_tj = 0; ++nx_ti; }
        }
        const int dh = ti * P.dhs, dw = tj * P.dws;
        const int a_delta = (dh * P.W + dw) * P.C + c0;
        const unsigned b_delta = (unsigned)(((P.r0 + ti * P.rs) * P.S + (P.s0 + tj * P.ss)) * P.C + c0) * 2u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = (unsigned)(a_ih0[i] + dh) < (unsigned)P.H && (unsigned)(a_iw0[i] + dw) < (unsigned)P.W;
            const unsigned off = ok ? (unsigned)(a_base[i] + a_delta) * 2u : OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (LDS_PTR(void))(sa + (i * NTHREADS + wv * 64) * 16),
                                                     16, off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const unsigned off = b_rowoff[i] != OOB_OFFSET ? b_rowoff[i] + b_delta : OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (LDS_PTR(void))(sb + (i * NTHREADS + wv * 64) * 16),
                                                     16, off, 0, 0, 0);
        }
    };

    const int wrow0 = wv * 32;
    const int fr = lane & 15, fq = lane >> 4;
    f4v acc[TA][4];
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_PTR(void))smem;
    const unsigned a_rd = lds0 + (wrow0 + fr) * 128;
    const unsigned b_rd = lds0 + A_BYTES + fr * 128;
    const unsigned sw0 = ((fq) ^ (fr & 7)) << 4;
    const unsigned sw1 = ((4 + fq) ^ (fr & 7)) << 4;

    issue(0);
    init_acc<TA>(acc, L, P.bias, n0, 0, fq);                    // (bias loads: older than every later DMA in vmcnt order)
    if (ksteps > 1) issue(1);
    for (int ks = 0; ks < ksteps; ++ks) {
        if (ks + 1 < ksteps) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES) : "memory");
        else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ks + 2 < ksteps) issue((ks + 2) % RS);
        const unsigned st = (unsigned)((ks % RS) * STAGE_BYTES);
        s8v af[2][TA], bf[2][4];
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            af[0][t] = t == 0 ? lds_b128<0>(a_rd + st + sw0) : lds_b128<16 * 128>(a_rd + st + sw0);
            af[1][t] = t == 0 ? lds_b128<0>(a_rd + st + sw1) : lds_b128<16 * 128>(a_rd + st + sw1);
        }
        bf[0][0] = lds_b128<0>(b_rd + st + sw0);            bf[1][0] = lds_b128<0>(b_rd + st + sw1);
        bf[0][1] = lds_b128<16 * 128>(b_rd + st + sw0);     bf[1][1] = lds_b128<16 * 128>(b_rd + st + sw1);
        bf[0][2] = lds_b128<32 * 128>(b_rd + st + sw0);     bf[1][2] = lds_b128<32 * 128>(b_rd + st + sw1);
        bf[0][3] = lds_b128<48 * 128>(b_rd + st + sw0);     bf[1][3] = lds_b128<48 * 128>(b_rd + st + sw1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);                  // keep the MFMAs behind the wait (guide 5.4 rule 18)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int i = 0; i < TA; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[sub][j], af[sub][i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();                                        // every wave is done with the ring: the epilogue reuses it
    gather_epilogue<2, TA, ST>(acc, L, P, smem, s_outoff, s_addoff, n0, m0 / BM);
}


// ---------------------------------------------------------------------------------------------
// ROW3: 3x3 / stride 1 / "same" convs with C % 64 == 0 (tower and FPN convs, bottleneck conv2 and
// their data gradients).  The three taps of one kernel row read the SAME input rows shifted by one
// pixel, so the activation tile of a (kernel row, 64-channel block) is staged ONCE, two rows longer,
// and the column taps are ds_reads at shifted row addresses (lanes whose column leaves the image read
// a zero row).  Per k-step this is 4 weight + 4.25/3 activation LDS-DMA pieces per thread instead of
// 8 (WM = 2), and a third of the gather-address arithmetic: the generic kernel is bound by exactly
// those (vector-memory issue, not MFMA).  Staged row j of a group holds the input pixel below
// "virtual output pixel" m0 - 1 + j; output row i and column tap s read staged row i + s.
// STAGES = 1: one activation buffer + one weight buffer (33 KiB): three blocks per CU, the DMA latency is
// hidden by the resident waves instead of a second stage (as in the generic one-stage kernel).
template <int WM, int STAGES = 2, int TA = 4, bool ST = false>
__global__ __launch_bounds__(WM * 128, TA == 2 ? 4 : (STAGES == 1 ? 3 : 2)) void conv_gather_row3_kernel(const LaunchDev L) {
    static_assert(TA == 4 || WM == 2, "narrow tile: 4 waves");
    constexpr int NTHREADS = WM * 128, BM = WM * 64, BNT = TA == 4 ? 128 : 64;
    constexpr int AROWS = BM + 8;                      // staged rows (BM + 2 used), 8-row DMA granules
    constexpr int A_BYTES = AROWS * BK * 2;
    constexpr int Z_ROW = (AROWS - 1) * 128;           // staged rows >= BM + 2 are never in range: zeros
    constexpr int B_OFF = STAGES * A_BYTES;
    constexpr int B_BYTES = BNT * BK * 2;
    constexpr int NB = BNT * 8 / NTHREADS;
    constexpr int ROWSTEP = NTHREADS / 8;
    constexpr int NA = (AROWS + ROWSTEP - 1) / ROWSTEP;   // A DMA passes (the last one partial)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ long long s_outoff[BM];
    __shared__ long long s_addoff[BM];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int m_tile = tile / L.n_tiles, n_tile = tile - m_tile * L.n_tiles;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MI355_CONV_MAX_PROBLEMS; ++i)
        if (i < L.nprob && m_tile >= L.p[i].tile_begin) pi = i;
    const ProbDev P = L.p[pi];
    const int m0 = (m_tile - P.tile_begin) * BM;
    const int n0 = n_tile * BNT;
    const int PQ = P.P * P.Q;

    if (tid < BM) {
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

    // ---- staging geometry
    const int slot = tid & 7;
    const int kc = slot ^ ((tid >> 3) & 7);            // (row & 7) is the same for every pass
    int a_base[NA], a_p[NA];
    unsigned b_rowoff[NB];
    const int wrow = L.T * P.C;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int j = (tid >> 3) + ROWSTEP * i;
        const int v = m0 - 1 + j;                      // virtual output pixel of staged row j
        a_p[i] = -0x40000000;
        a_base[i] = 0;
        if (v >= 0 && v < P.M && j < BM + 2) {
            const int n = (int)fdiv((unsigned)v, P.dPQ), rem = v - n * PQ;
            const int p = (int)fdiv((unsigned)rem, P.dQ), q = rem - p * P.Q;
            a_p[i] = p;
            a_base[i] = n * (int)P.in_sN + (p * P.W + q) * P.C;
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int nn = n0 + (tid >> 3) + ROWSTEP * i;
        b_rowoff[i] = nn < L.K ? (unsigned)(nn * wrow) * 2u : OOB_OFFSET;
    }
    const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(P.in, (unsigned)P.in_bytes);
    const __amdgpu_buffer_rsrc_t rsrc_b = make_rsrc(P.w, (unsigned)L.w_bytes);
    const int nchunks = P.C / BK;
    const int ngroups = 3 * nchunks;                   // (kernel row, channel block)
    const int nsteps = 3 * ngroups;                    // x column tap

    // groups and steps are issued in order: carried counters, no division
    int ga_r = 0, ga_c = 0;                            // next A group
    auto issue_a = [&](int buf) {
        unsigned char* sa = smem + buf * A_BYTES;
        const int dh = P.dh0 + ga_r;
        const int delta = dh * P.W * P.C + ga_c + kc * 8;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i * ROWSTEP + (wv * 8) < AROWS) {      // wave-uniform: the last pass is partial
                const bool ok = (unsigned)(a_p[i] + dh) < (unsigned)P.H;
                const unsigned off = ok ? (unsigned)(a_base[i] + delta) * 2u : OOB_OFFSET;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (LDS_PTR(void))(sa + (i * NTHREADS + wv * 64) * 16),
                                                         16, off, 0, 0, 0);
            }
        }
        ga_c += BK;
        if (ga_c == P.C) { ga_c = 0; ++ga_r; }
    };
    int gb_r = 0, gb_c = 0, gb_s = 0;                  // next B step
    auto issue_b = [&](int buf) {
        unsigned char* sb = smem + B_OFF + buf * B_BYTES;
        const unsigned b_delta = (unsigned)(((P.r0 + gb_r * P.rs) * P.S + (P.s0 + gb_s * P.ss)) * P.C + gb_c + kc * 8) * 2u;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const unsigned off = b_rowoff[i] != OOB_OFFSET ? b_rowoff[i] + b_delta : OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (LDS_PTR(void))(sb + (i * NTHREADS + wv * 64) * 16),
                                                     16, off, 0, 0, 0);
        }
        if (++gb_s == 3) {
            gb_s = 0;
            gb_c += BK;
            if (gb_c == P.C) { gb_c = 0; ++gb_r; }
        }
    };

    const int wrow0 = TA == 4 ? (wv >> 1) * 64 : wv * 32, wn = TA == 4 ? (wv & 1) : 0;
    const int fr = lane & 15, fq = lane >> 4;
    f4v acc[TA][4];
    // fragment addresses: output row i = wm*64 + t*16 + fr reads staged row i + s (s = 0, 1, 2);
    // lanes whose column q + s - 1 leaves [0, Q) read the zero row instead.
    int a_rd[TA][3];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        const int i = wrow0 + t * 16 + fr;
        const int m = min(m0 + i, P.M - 1);
        const int n = (int)fdiv((unsigned)m, P.dPQ), rem = m - n * PQ;
        const int p = (int)fdiv((unsigned)rem, P.dQ), q = rem - p * P.Q;
        a_rd[t][0] = q > 0 ? i * 128 : Z_ROW;
        a_rd[t][1] = (i + 1) * 128;
        a_rd[t][2] = q < P.Q - 1 ? (i + 2) * 128 : Z_ROW;
    }
    int a_sw[3][2];                                    // (chunk ^ (staged row & 7)) << 4, per tap and k-sub
#pragma unroll
    for (int sft = 0; sft < 3; ++sft) {
        a_sw[sft][0] = (fq ^ ((fr + sft) & 7)) << 4;
        a_sw[sft][1] = ((4 + fq) ^ ((fr + sft) & 7)) << 4;
    }
    const int b_rd = (wn * 64 + fr) * 128;
    const int b_sw0 = (fq ^ (fr & 7)) << 4, b_sw1 = ((4 + fq) ^ (fr & 7)) << 4;

    if (STAGES == 2) {
        issue_a(0);
        issue_b(0);
    }
    init_acc<TA>(acc, L, P.bias, n0, wn, fq);

    int abuf = 0, step = 0;
    for (int group = 0; group < ngroups; ++group) {
#pragma unroll
        for (int sft = 0; sft < 3; ++sft) {            // static: fragment addresses stay in registers
            __syncthreads();                           // every DMA issued so far has landed (vmcnt(0))
            if (STAGES == 2) {
                if (sft == 0 && group + 1 < ngroups) issue_a(abuf ^ 1);   // last read two groups ago
                if (step + 1 < nsteps) issue_b((step + 1) & 1);
            } else {                                   // (the barrier above: every read of the buffers retired)
                if (sft == 0) issue_a(0);
                issue_b(0);
                __syncthreads();
            }
            const unsigned char* sa = smem + (STAGES == 2 ? abuf : 0) * A_BYTES;
            const unsigned char* sb = smem + B_OFF + (STAGES == 2 ? (step & 1) * B_BYTES : 0);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                s8v af[TA], bf[4];
#pragma unroll
                for (int t = 0; t < TA; ++t)
                    af[t] = *reinterpret_cast<const s8v*>(sa + a_rd[t][sft] + a_sw[sft][sub]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    bf[t] = *reinterpret_cast<const s8v*>(sb + b_rd + t * 16 * 128 + (sub ? b_sw1 : b_sw0));
#pragma unroll
                for (int i = 0; i < TA; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
            }
            ++step;
        }
        abuf ^= 1;
    }
    __syncthreads();
    gather_epilogue<WM, TA, ST>(acc, L, P, smem, s_outoff, s_addoff, n0, m0 / BM);
}

// Split-K finish: out = mask( act( sum_s ws[s] + bias (+ addend) ) ) with the generic epilogue's ordering;
// the partial slabs are summed in index order (bit-reproducible).
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ ws, int ksplit, long long split_stride,
                                                            const ProbDev P, const float* __restrict__ bias,
                                                            int K, int act, int out_f32, int mask_first) {
    const int kc8 = (K + 7) >> 3;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)P.M * kc8) return;
    const int m = (int)(idx / kc8), k0 = (int)(idx - (long long)m * kc8) * 8;
    const int PQ = P.P * P.Q;
    const int n = (int)fdiv((unsigned)m, P.dPQ), rem = m - n * PQ;
    const int p = (int)fdiv((unsigned)rem, P.dQ), q = rem - p * P.Q;
    const long long oo = n * P.out_sN + (long long)(p * P.out_mh + P.out_oh) * P.out_sH +
                         (long long)(q * P.out_mw + P.out_ow) * P.out_sW;
    const long long ao = n * P.add_sN + (long long)(p >> P.add_shift) * P.add_sH + (long long)(q >> P.add_shift) * P.add_sW;
    const float* w = ws + (long long)m * K + k0;
    const int nv = min(8, K - k0);
    if ((K & 7) == 0) {                                // 16-B path (every split-K launch of the plans)
        float x[8];
        {
            const float4 a = *reinterpret_cast<const float4*>(w), b = *reinterpret_cast<const float4*>(w + 4);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        }
        for (int sp = 1; sp < ksplit; ++sp) {
            const float4 a = *reinterpret_cast<const float4*>(w + sp * split_stride);
            const float4 b = *reinterpret_cast<const float4*>(w + sp * split_stride + 4);
            x[0] += a.x; x[1] += a.y; x[2] += a.z; x[3] += a.w; x[4] += b.x; x[5] += b.y; x[6] += b.z; x[7] += b.w;
        }
        s8v mk = (s8v){0, 0, 0, 0, 0, 0, 0, 0}, ad = mk;
        if (P.mask) mk = *reinterpret_cast<const s8v*>(P.mask + oo + k0);
        if (P.addend) ad = *reinterpret_cast<const s8v*>(P.addend + ao + k0);
        s8v o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = x[e] + (bias ? bias[k0 + e] : 0.0f);
            const bool on = bf16_to_f32((unsigned short)mk[e]) > 0.0f;
            if (P.mask && mask_first) v = on ? v : 0.0f;
            if (P.addend) v += bf16_to_f32((unsigned short)ad[e]);
            v = activate(v, act);
            if (P.mask && !mask_first) v = on ? v : 0.0f;
            x[e] = v;
            o[e] = (short)f32_to_bf16_rn(v);
        }
        if (out_f32) {
            float* op = reinterpret_cast<float*>(P.out) + oo + k0;
            *reinterpret_cast<float4*>(op) = make_float4(x[0], x[1], x[2], x[3]);
            *reinterpret_cast<float4*>(op + 4) = make_float4(x[4], x[5], x[6], x[7]);
        } else {
            *reinterpret_cast<s8v*>(reinterpret_cast<unsigned short*>(P.out) + oo + k0) = o;
        }
        return;
    }
    for (int e = 0; e < nv; ++e) {
        float x = w[e];
        for (int sp = 1; sp < ksplit; ++sp) x += w[sp * split_stride + e];
        if (bias) x += bias[k0 + e];
        const float mk = P.mask ? bf16_to_f32(P.mask[oo + k0 + e]) : 1.0f;
        if (P.mask && mask_first) x = mk > 0.0f ? x : 0.0f;
        if (P.addend) x += bf16_to_f32(P.addend[ao + k0 + e]);
        x = activate(x, act);
        if (P.mask && !mask_first) x = mk > 0.0f ? x : 0.0f;
        if (out_f32) reinterpret_cast<float*>(P.out)[oo + k0 + e] = x;
        else reinterpret_cast<unsigned short*>(P.out)[oo + k0 + e] = f32_to_bf16_rn(x);
    }
}

// ---------------------------------------------------------------------------------------------
// STREAM: 1x1 / stride-1 convs with a short reduction (C <= 256) whose weights fit in LDS (EfficientNet's expand /
// project convs on the 256^2 ... 64^2 maps and their data gradients).  The tiled kernels above top out at ~170 tiles
// per us whatever a tile moves (a 128-pixel tile of a 16 -> 96 conv carries 28 KiB: 2.8 TB/s); here a wave streams
// 16-pixel rows straight from global memory into the MFMA B operand (a lane's fragment IS 16 contiguous bytes of one
// pixel: no LDS, no barrier), the weights sit in LDS for the whole block, and the accumulators are the transposed tile
// with the 16 rows of a fragment pair permuted (row 4g + r of fragment t <-> channel 8g + 4t + r) so that a lane ends
// up with 8 consecutive output channels of one pixel: one 16-B store, 64 contiguous bytes per pixel and instruction.
struct PwStream {
    const unsigned short* x;
    const unsigned short* w;
    const float* bias;
    const unsigned short* addend;
    const unsigned short* mask;
    unsigned short* out;
    long long out_row, add_row;     // elements between consecutive pixels of out / addend
    int M, C, K, act, mask_first, ntiles;
    // WIN: a stride-1 conv whose tap row is a contiguous window of the NHWC input (TS * Cin elements, a multiple of
    // 32: the space-to-depth stem, 4 taps x 16 channels): reduction step cs = (tap row a, 32-element piece h of its
    // window), C = TR * TS * Cin "virtual" channels in the weights' own [K][tap][Cin] order
    int Cin, H, W, pad_t, pad_l, spr;   // spr = reduction steps per tap row
    FastDiv dHW, dW;
};

template <int CS, int PT, bool WIN = false>
__global__ __launch_bounds__(256) void conv_pw_stream_kernel(const PwStream S) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CP = CS * 32 + 8;                     // LDS row of a weight row (elements): 16-B aligned, banks staggered
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int KP = (S.K + 31) & ~31;
    float* s_bias = reinterpret_cast<float*>(smem + (size_t)KP * CP * 2);
    // [KP][CS * 32] <- w[K][C], zero beyond K / C; eight 16-B pieces per thread in flight (one at a time was eight
    // dependent round trips for a 38 KiB weight block: longer than streaming the block's pixels)
    for (int i0 = tid; i0 < KP * (CS * 4); i0 += 256 * 8) {
        s8v v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            const int k = i / (CS * 4), c = (i - k * (CS * 4)) * 8;
            v[u] = (s8v){0, 0, 0, 0, 0, 0, 0, 0};
            if (k < S.K && c < S.C) v[u] = *reinterpret_cast<const s8v*>(S.w + (long long)k * S.C + c);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            const int k = i / (CS * 4), c = (i - k * (CS * 4)) * 8;
            if (k < KP) *reinterpret_cast<s8v*>(smem + ((size_t)k * CP + c) * 2) = v[u];
        }
    }
    for (int i = tid; i < KP; i += 256) s_bias[i] = (S.bias && i < S.K) ? S.bias[i] : 0.0f;
    __syncthreads();

    const int p = lane & 15, g = lane >> 4;
    // A fragment (weights) of fragment t of a 32-channel group: row i = lane & 15 <-> channel 8 (i >> 2) + 4 t + (i & 3)
    const int a_row = 8 * (p >> 2) + (p & 3);
    const unsigned char* a_base = smem + ((size_t)a_row * CP + 8 * g) * 2;
    const int KG = KP >> 5;
    for (int st = blockIdx.x * 4 + wv; st < S.ntiles; st += gridDim.x * 4) {
        const int m0 = st * (PT * 16);
        s8v xf[PT][CS];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int m = m0 + pt * 16 + p;
            if (WIN) {
                const int mc = min(m, S.M - 1);
                const int n = (int)fdiv((unsigned)mc, S.dHW), rem = mc - n * S.H * S.W;
                const int py = (int)fdiv((unsigned)rem, S.dW), px = rem - py * S.W;
#pragma unroll
                for (int cs = 0; cs < CS; ++cs) {
                    const int a = cs / S.spr, e = (cs - a * S.spr) * 32 + 8 * g;
                    const int b = e / S.Cin, ch = e - b * S.Cin;
                    const int r = py + a - S.pad_t, col = px + b - S.pad_l;
                    xf[pt][cs] = (s8v){0, 0, 0, 0, 0, 0, 0, 0};
                    if (m < S.M && (unsigned)r < (unsigned)S.H && (unsigned)col < (unsigned)S.W)
                        xf[pt][cs] = *reinterpret_cast<const s8v*>(S.x + (((long long)n * S.H + r) * S.W + col) * S.Cin + ch);
                }
                continue;
            }
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) {
                const int c = cs * 32 + 8 * g;
                xf[pt][cs] = (s8v){0, 0, 0, 0, 0, 0, 0, 0};
                if (m < S.M && c < S.C) xf[pt][cs] = *reinterpret_cast<const s8v*>(S.x + (long long)m * S.C + c);
            }
        }
        for (int kg = 0; kg < KG; ++kg) {
            const int kk = kg * 32 + 8 * g;              // this lane's 8 output channels
            f4v acc[PT][2];
            {
                const f4v b0 = *reinterpret_cast<const f4v*>(s_bias + kk), b1 = *reinterpret_cast<const f4v*>(s_bias + kk + 4);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) { acc[pt][0] = b0; acc[pt][1] = b1; }
            }
            const unsigned char* ak = a_base + (size_t)kg * 32 * CP * 2;
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) {
                const s8v a0 = *reinterpret_cast<const s8v*>(ak + cs * 64);
                const s8v a1 = *reinterpret_cast<const s8v*>(ak + cs * 64 + 4 * CP * 2);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    acc[pt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, xf[pt][cs], acc[pt][0], 0, 0, 0);
                    acc[pt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, xf[pt][cs], acc[pt][1], 0, 0, 0);
                }
            }
            if (kk >= S.K) continue;                     // (K % 8 == 0: a lane's 8 channels are in or out together)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int m = m0 + pt * 16 + p;
                if (m >= S.M) continue;
                float v[8] = {acc[pt][0][0], acc[pt][0][1], acc[pt][0][2], acc[pt][0][3],
                              acc[pt][1][0], acc[pt][1][1], acc[pt][1][2], acc[pt][1][3]};
                s8v mk = (s8v){0, 0, 0, 0, 0, 0, 0, 0};
                if (S.mask) mk = *reinterpret_cast<const s8v*>(S.mask + (long long)m * S.out_row + kk);
                if (S.mask && S.mask_first) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = bf16_to_f32((unsigned short)mk[e]) > 0.0f ? v[e] : 0.0f;
                }
                if (S.addend) {
                    const s8v ad = *reinterpret_cast<const s8v*>(S.addend + (long long)m * S.add_row + kk);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bf16_to_f32((unsigned short)ad[e]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = activate(v[e], S.act);
                if (S.mask && !S.mask_first) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = bf16_to_f32((unsigned short)mk[e]) > 0.0f ? v[e] : 0.0f;
                }
                s8v o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (short)f32_to_bf16_rn(v[e]);
                *reinterpret_cast<s8v*>(S.out + (long long)m * S.out_row + kk) = o;
            }
        }
    }
}

template <int CS, int PT, bool WIN = false>
static void pw_stream_launch(const PwStream& S, int lds, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_pw_stream_kernel<CS, PT, WIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr = true;
    }
    int per_cu = (160 * 1024 - 4096) / (lds + 256);        // blocks of one CU by LDS (160 KiB), at most 6
    per_cu = per_cu > 6 ? 6 : (per_cu < 1 ? 1 : per_cu);
    const int nb = (S.ntiles + 3) / 4;
    const int grid = nb < 256 * per_cu ? nb : 256 * per_cu;
    hipLaunchKernelGGL((conv_pw_stream_kernel<CS, PT, WIN>), dim3((unsigned)grid), dim3(256), lds, st, S);
}

}  // namespace

// MI355_GATHER_ONE=0: the general staging for single-tap launches (A/B)
static bool one_tap_mode() {
    static const int on = [] { const char* e = getenv("MI355_GATHER_ONE"); return e ? atoi(e) : 1; }();
    return on != 0;
}

static int gather_launch(const Mi355ConvLaunch& in, int ksplit, float* ws, void* stream);

extern "C" int mi355_conv_gather(const Mi355ConvLaunch* launch, void* stream) {
    MI355_REQUIRE(launch, "conv_gather: null launch");
    return gather_launch(*launch, 1, nullptr, stream);
}

extern "C" int mi355_conv_gather_splitk(const Mi355ConvLaunch* launch, int ksplit, float* ws, void* stream) {
    MI355_REQUIRE(launch && ws && ksplit >= 1, "conv_gather_splitk: bad arguments");
    return gather_launch(*launch, ksplit, ws, stream);
}

static int gather_launch(const Mi355ConvLaunch& in, int ksplit, float* ws, void* stream) {
    MI355_REQUIRE(in.nprob >= 1 && in.nprob <= MI355_CONV_MAX_PROBLEMS, "conv_gather: nprob %d", in.nprob);
    MI355_REQUIRE(in.K > 0 && in.weight && in.T > 0, "conv_gather: bad K/weight/T");
    LaunchDev L;
    memset(&L, 0, sizeof(L));
    int tiles = 0;
    const int C = in.prob[0].C;
    MI355_REQUIRE(C > 0 && C % 8 == 0, "conv_gather: C must be a multiple of 8");
    // ---- streaming form of the short-reduction 1x1 convs (MI355_PW_STREAM: 0 off, 1 where it was measured ahead of the
    //      tiled kernels, 2 wherever it applies)
    const char* pw_env = getenv("MI355_PW_STREAM");         // (read per call: the tests switch it)
    int pw_mode = pw_env ? atoi(pw_env) : 1;
    if (in.stats) {
        MI355_REQUIRE(in.nprob == 1 && ksplit == 1 && !in.out_f32 && !in.prob[0].addend && in.K % 8 == 0 && !in.prob[0].mask,
                      "conv_gather: epilogue statistics need one problem, bf16 output, no addend / mask, K %% 8 == 0, no split-K");
        pw_mode = 0;                                        // (the streaming forms have their own epilogue)
    }
    if (pw_mode > 0 && in.nprob == 1 && ksplit == 1 && in.T > 1 && !in.out_f32 && in.K % 8 == 0 && C <= 32) {
        // window form: the space-to-depth stem (4 x 4 taps over 16 channels, stride 1): 79 -> 68 us at 16 x 256 x 256 (two
        // pixel rows per wave instead of one: 81 us, the registers cost a wave per SIMD)
        const Mi355ConvProblem& s = in.prob[0];
        const long long PQ = (long long)s.P * s.Q, M = (long long)s.N * PQ;
        const int CS = (in.T * C + 31) / 32, KP = (in.K + 31) & ~31;
        const int lds = KP * (CS * 32 + 8) * 2 + KP * 4;
        const bool ok =
            s.TR * s.TS == in.T && (s.TS * C) % 32 == 0 && CS <= 8 && s.sh == 1 && s.sw == 1 && s.dhs == 1 && s.dws == 1 &&
            s.r0 == 0 && s.rs == 1 && s.s0 == 0 && s.ss == 1 && s.S == s.TS && s.P == s.H && s.Q == s.W &&
            (s.in_sN == 0 || s.in_sN == (long long)s.H * s.W * C) && s.in && s.out && s.N > 0 && PQ > 0 &&
            s.out_mh == 1 && s.out_mw == 1 && s.out_oh == 0 && s.out_ow == 0 && s.out_sW >= in.K && s.out_sW % 8 == 0 &&
            s.out_sH == s.Q * s.out_sW && s.out_sN == PQ * s.out_sW && !s.addend && !s.mask &&
            (((uintptr_t)s.in | (uintptr_t)s.out) & 15) == 0 && M < 0x7fffffffLL && lds <= 65536;
        if (ok && (pw_mode > 1 || (M >= 131072 && in.K <= 144))) {
            PwStream S;
            memset(&S, 0, sizeof(S));
            S.x = (const unsigned short*)s.in;
            S.w = (const unsigned short*)(s.weight ? s.weight : in.weight);
            S.bias = s.weight ? s.bias : in.bias;
            S.out = (unsigned short*)s.out;
            S.out_row = s.out_sW;
            S.M = (int)M;  S.C = in.T * C;  S.K = in.K;  S.act = in.act;
            S.Cin = C;  S.H = s.H;  S.W = s.W;  S.pad_t = -s.dh0;  S.pad_l = -s.dw0;  S.spr = s.TS * C / 32;
            S.dHW = make_fastdiv((unsigned)(s.H * s.W));  S.dW = make_fastdiv((unsigned)s.W);
            MI355_REQUIRE(((uintptr_t)S.w & 15) == 0, "conv_gather: weight alignment");
            const int pt = CS <= 2 ? 4 : (CS <= 4 ? 2 : 1);
            S.ntiles = (int)cdiv(M, 16 * pt);
            hipStream_t st = (hipStream_t)stream;
            switch (CS) {
                case 1: pw_stream_launch<1, 4, true>(S, lds, st); break;
                case 2: pw_stream_launch<2, 4, true>(S, lds, st); break;
                case 3: pw_stream_launch<3, 2, true>(S, lds, st); break;
                case 4: pw_stream_launch<4, 2, true>(S, lds, st); break;
                case 5: pw_stream_launch<5, 1, true>(S, lds, st); break;
                case 6: pw_stream_launch<6, 1, true>(S, lds, st); break;
                case 7: pw_stream_launch<7, 1, true>(S, lds, st); break;
                default: pw_stream_launch<8, 1, true>(S, lds, st); break;
            }
            return check_launch("conv_pw_stream_kernel<window>");
        }
    }
    if (pw_mode > 0 && in.nprob == 1 && ksplit == 1 && in.T == 1 && !in.out_f32 && in.K % 8 == 0 && C <= 256) {
        const Mi355ConvProblem& s = in.prob[0];
        const long long PQ = (long long)s.P * s.Q, M = (long long)s.N * PQ;
        const int CS = (C + 31) / 32, KP = (in.K + 31) & ~31;
        const int lds = KP * (CS * 32 + 8) * 2 + KP * 4;
        const bool dense =
            s.TR == 1 && s.TS == 1 && s.sh == 1 && s.sw == 1 && s.dh0 == 0 && s.dw0 == 0 && s.P == s.H && s.Q == s.W &&
            (s.in_sN == 0 || s.in_sN == (long long)s.H * s.W * C) && s.in && s.out && s.N > 0 && PQ > 0 &&
            s.out_mh == 1 && s.out_mw == 1 && s.out_oh == 0 && s.out_ow == 0 && s.out_sW >= in.K && s.out_sW % 8 == 0 &&
            s.out_sH == s.Q * s.out_sW && s.out_sN == PQ * s.out_sW &&
            (!s.addend || (s.add_shift == 0 && s.add_sW % 8 == 0 && s.add_sH == s.Q * s.add_sW && s.add_sN == PQ * s.add_sW)) &&
            (((uintptr_t)s.in | (uintptr_t)s.out | (uintptr_t)s.addend | (uintptr_t)s.mask) & 15) == 0 &&
            M < 0x7fffffffLL && lds <= 65536;
        // Where it wins (per-launch A/B on EfficientDet-D0 and RetinaNet-R50 at batch 16, tools/profile_steps.py /
        // profile_plan.py with MI355_PW_STREAM=0 / 2): reductions the tiled kernels run in their flat form (C % 64 != 0,
        // partly empty tiles at ~170 tiles per us whatever a tile carries) on the large maps - 16 -> 96 at 256 x 256
        // 84 -> 50 us, 96 -> 16 53 -> 38, 32 -> 16 44 -> 21, 144 -> 24 at 128 x 128 24.5 -> 16.4, 144 -> 40 at 64 x 64
        // 10.1 -> 6.9.  Where it loses: C % 64 == 0 (RetinaNet 64 -> 256 at 128 x 128: 65 tiled vs 93 us), wide outputs
        // on small maps (a block stages K x C weights for a handful of pixels: 40 -> 240 at 64 x 64 15 vs 22 us,
        // 64 -> 320 at 16 x 16 6.6 vs 28.6) and everything the tiled kernel finishes in ~6 us.
        const bool wins = C % 64 != 0 && ((M >= 131072 && in.K <= 144) || (M >= 65536 && C >= 96 && in.K <= 48));
        if (dense && (pw_mode > 1 || wins)) {
            PwStream S;
            memset(&S, 0, sizeof(S));
            S.x = (const unsigned short*)s.in;
            S.w = (const unsigned short*)(s.weight ? s.weight : in.weight);
            S.bias = s.weight ? s.bias : in.bias;
            S.addend = (const unsigned short*)s.addend;
            S.mask = (const unsigned short*)s.mask;
            S.out = (unsigned short*)s.out;
            S.out_row = s.out_sW;  S.add_row = s.add_sW;
            S.M = (int)M;  S.C = C;  S.K = in.K;  S.act = in.act;  S.mask_first = in.flags & 1;
            MI355_REQUIRE(((uintptr_t)S.w & 15) == 0, "conv_gather: weight alignment");
            hipStream_t st = (hipStream_t)stream;
            const int pt = CS <= 2 ? 4 : (CS <= 4 ? 2 : 1);
            S.ntiles = (int)cdiv(M, 16 * pt);
            switch (CS) {
                case 1: pw_stream_launch<1, 4>(S, lds, st); break;
                case 2: pw_stream_launch<2, 4>(S, lds, st); break;
                case 3: pw_stream_launch<3, 2>(S, lds, st); break;
                case 4: pw_stream_launch<4, 2>(S, lds, st); break;
                case 5: pw_stream_launch<5, 1>(S, lds, st); break;
                case 6: pw_stream_launch<6, 1>(S, lds, st); break;
                case 7: pw_stream_launch<7, 1>(S, lds, st); break;
                default: pw_stream_launch<8, 1>(S, lds, st); break;
            }
            return check_launch("conv_pw_stream_kernel");
        }
    }
    const bool flat = C % BK != 0;
    int max_ksteps = 0;
    long long grid128 = 0;                             // 128x128 tiles of the launch
    for (int i = 0; i < in.nprob; ++i) {
        const int taps = in.prob[i].TR * in.prob[i].TS;
        const int ks = flat ? (taps * (C / 8) + 7) / 8 : taps * (C / BK);
        if (ks > max_ksteps) max_ksteps = ks;
        grid128 += cdiv((long long)in.prob[i].N * in.prob[i].P * in.prob[i].Q, 128) * cdiv(in.K, BN);
    }
    const bool splitk = ksplit > 1;
    // ---- kernel choice (per-launch A/B on R50 512^2 bs16 with tools/profile_plan.py; env knobs keep the
    //      alternatives reachable).
    // 128x64 tiles (four wave rows of 32x64 accumulators, ~95 VGPRs: up to five blocks per CU): faster than the
    // 128x128 tile everywhere (more bytes in flight for the HBM-bound 1x1 convs: 66 vs 90 us on 64->256 @128^2;
    // no empty wave column when K <= 64: stem 111 vs 209 us) except the long MFMA-bound reductions with many
    // tiles (3x3, C >= 128: head towers 102 vs 121 us), which keep the wide tile.
    static const int narrow_mode = [] {
        const char* e = getenv("MI355_NARROW");
        return e ? atoi(e) : 3;                        // 0 off, 1 K <= 64 only, 2 every launch, 3 auto
    }();
    const int taps0 = in.prob[0].TR * in.prob[0].TS;
    static const int wide_min_grid = [] { const char* e = getenv("MI355_WIDE_MIN_GRID"); return e ? atoi(e) : 1024; }();
    const bool wide_wins = taps0 >= 9 && C >= 128 && grid128 >= wide_min_grid;
    const bool narrow = !splitk && ((narrow_mode == 1 && in.K <= 64) || narrow_mode == 2 ||
                                    (narrow_mode == 3 && (in.K <= 64 || !wide_wins)));
    if (narrow) grid128 = grid128 / cdiv(in.K, BN) * cdiv(in.K, 64);
    // One stage (33 KiB of LDS, 3 blocks = 12 waves per CU) for <= 2 k-steps and for LONG reductions once the
    // launch has at least three tiles per CU: on the head tower convs 1010 TFLOP/s against 920 double-buffered
    // (2 blocks per CU); resident waves, not the depth of one block's pipeline, hide the LDS-DMA latency.
    static const int single_min_grid = [] { const char* e = getenv("MI355_SINGLE_MIN_GRID"); return e ? atoi(e) : 768; }();
    const bool many = single_min_grid > 0 && grid128 >= single_min_grid && !splitk;
    const bool single = (max_ksteps <= 2 || many) && !splitk;
    // Row-of-taps kernel for the wide 3x3 / stride-1 / same launches (MI355_ROW3=0 disables it).
    static const int row3_mode = [] {
        const char* e = getenv("MI355_ROW3");
        return e ? atoi(e) : 1;
    }();
    // ... and its 128x64 form for the narrow launches (MI355_ROW3N=0: the three-stage ring / generic kernel instead)
    static const int row3n_mode = [] {
        const char* e = getenv("MI355_ROW3N");
        return e ? atoi(e) : 1;
    }();
    bool row3 = !flat && row3_mode > 0 && !splitk && (!narrow || row3n_mode > 0);
    for (int i = 0; i < in.nprob && row3; ++i) {
        const Mi355ConvProblem& s = in.prob[i];
        // (data gradients walk the taps downwards: dw0 = +1, dws = -1; canonicalised below)
        row3 = s.TR == 3 && s.TS == 3 && s.sh == 1 && s.sw == 1 && (s.dhs == 1 || s.dhs == -1) &&
               ((s.dws == 1 && s.dw0 == -1) || (s.dws == -1 && s.dw0 == 1)) && s.Q == s.W;
    }
    constexpr int BM = 128;
    for (int i = 0; i < in.nprob; ++i) {
        const Mi355ConvProblem& s = in.prob[i];
        MI355_REQUIRE(s.in && s.out, "conv_gather: problem %d null in/out", i);
        MI355_REQUIRE(s.C == C, "conv_gather: C must be equal across problems");
        MI355_REQUIRE(s.N > 0 && s.H > 0 && s.W > 0 && s.P > 0 && s.Q > 0, "conv_gather: problem %d bad dims", i);
        MI355_REQUIRE(s.TR >= 0 && s.TS >= 0 && s.add_shift >= 0, "conv_gather: bad taps");
        const long long in_sN = s.in_sN ? s.in_sN : (long long)s.H * s.W * s.C;
        const long long in_bytes = ((long long)(s.N - 1) * in_sN + (long long)s.H * s.W * s.C) * 2;
        MI355_REQUIRE(in_bytes < 0x7fffffffLL, "conv_gather: input tensor >= 2 GiB");
        MI355_REQUIRE((long long)s.N * s.P * s.Q < 0x7fffffffLL, "conv_gather: M too large");
        MI355_REQUIRE(((uintptr_t)s.in & 15) == 0 && ((uintptr_t)s.out & 15) == 0, "conv_gather: alignment");
        ProbDev& d = L.p[i];
        d.in = (const unsigned short*)s.in;  d.out = s.out;
        d.addend = (const unsigned short*)s.addend;  d.mask = (const unsigned short*)s.mask;
        MI355_REQUIRE(!s.weight || ((uintptr_t)s.weight & 15) == 0, "conv_gather: per-problem weights need 16-B alignment");
        d.w = (const unsigned short*)(s.weight ? s.weight : in.weight);
        d.bias = s.weight ? s.bias : in.bias;
        d.N = s.N; d.H = s.H; d.W = s.W; d.C = s.C; d.P = s.P; d.Q = s.Q; d.sh = s.sh; d.sw = s.sw;
        d.TR = s.TR; d.TS = s.TS; d.dh0 = s.dh0; d.dhs = s.dhs; d.dw0 = s.dw0; d.dws = s.dws;
        d.r0 = s.r0; d.rs = s.rs; d.s0 = s.s0; d.ss = s.ss; d.S = s.S;
        d.out_mh = s.out_mh; d.out_oh = s.out_oh; d.out_mw = s.out_mw; d.out_ow = s.out_ow;
        d.out_sN = s.out_sN; d.out_sH = s.out_sH; d.out_sW = s.out_sW;
        d.add_sN = s.add_sN; d.add_sH = s.add_sH; d.add_sW = s.add_sW; d.add_shift = s.add_shift;
        d.in_sN = in_sN;
        d.M = s.N * s.P * s.Q;
        d.dPQ = make_fastdiv((unsigned)(s.P * s.Q));
        d.dQ = make_fastdiv((unsigned)s.Q);
        d.in_bytes = (int)in_bytes;
        if (row3) {                                    // ascending taps: tap i' = 2 - i
            if (d.dhs < 0) { d.dh0 += 2 * d.dhs; d.dhs = 1; d.r0 += 2 * d.rs; d.rs = -d.rs; }
            if (d.dws < 0) { d.dw0 += 2 * d.dws; d.dws = 1; d.s0 += 2 * d.ss; d.ss = -d.ss; }
        }
        d.tile_begin = tiles;
        tiles += (int)cdiv(d.M, BM);
    }
    const long long w_bytes = (long long)in.K * in.T * C * 2;
    MI355_REQUIRE(w_bytes < 0x7fffffffLL, "conv_gather: weight tensor >= 2 GiB");
    MI355_REQUIRE(((uintptr_t)in.weight & 15) == 0, "conv_gather: weight alignment");
    L.w = (const unsigned short*)in.weight;
    L.bias = in.bias;
    L.nprob = in.nprob; L.K = in.K; L.T = in.T; L.act = in.act; L.out_f32 = in.out_f32; L.flags = in.flags;
    L.stats = in.stats;
    L.n_tiles = (int)cdiv(in.K, narrow ? 64 : BN);
    L.w_bytes = (int)w_bytes;
    long long grid = (long long)tiles * L.n_tiles;
    MI355_REQUIRE(grid > 0 && grid < 0x7fffffffLL, "conv_gather: grid");
    L.ksplit = 1;
    L.tiles_total = (int)grid;
    ProbDev fin[MI355_CONV_MAX_PROBLEMS];
    if (splitk) {
        // partial tiles go to slab `split` of the fp32 workspace [ksplit][sum M][K] (plain stores); bias,
        // addend, activation and mask are applied by the finish kernel
        MI355_REQUIRE(max_ksteps >= ksplit, "conv_gather_splitk: ksplit %d > %d k-steps", ksplit, max_ksteps);
        long long off = 0;
        for (int i = 0; i < in.nprob; ++i) {
            ProbDev& d = L.p[i];
            fin[i] = d;
            d.out = ws + off;
            d.out_sN = (long long)d.P * d.Q * in.K; d.out_sH = (long long)d.Q * in.K; d.out_sW = in.K;
            d.out_mh = d.out_mw = 1; d.out_oh = d.out_ow = 0;
            d.addend = nullptr; d.mask = nullptr; d.bias = nullptr;
            off += (long long)d.M * in.K;
        }
        L.bias = nullptr; L.act = 0; L.out_f32 = 1;
        L.ksplit = ksplit;
        L.split_stride = off;
        grid *= ksplit;
        MI355_REQUIRE(grid < 0x7fffffffLL, "conv_gather_splitk: grid");
    }
    constexpr int STAGE2 = (128 + BN) * BK * 2;        // 32 KiB
    constexpr int NST = (128 + 64) * BK * 2;           // 24 KiB: narrow tile
    constexpr int ROW3_A = (128 + 8) * BK * 2, ROW3_B = BN * BK * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_gather_kernel<false, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE2);
        (void)hipFuncSetAttribute((const void*)conv_gather_kernel<true, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE2);
        (void)hipFuncSetAttribute((const void*)conv_gather_kernel<false, 2, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NST);
        (void)hipFuncSetAttribute((const void*)conv_gather_kernel<true, 2, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NST);
        (void)hipFuncSetAttribute((const void*)conv_gather_kernel<false, 1, 2, 2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NST);
        (void)hipFuncSetAttribute((const void*)conv_gather_row3_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ROW3_A + 2 * ROW3_B);
        attr_set = true;
    }
    const dim3 g((unsigned)grid), b256(256);
    hipStream_t st = (hipStream_t)stream;
    // three-stage ring for the narrow-tile launches that are too small for residency to hide the DMA latency
    static const int ring_mode = [] { const char* e = getenv("MI355_RING"); return e ? atoi(e) : 1; }();
    if (L.stats) {
        // The statistics forms (ST: separate instantiations, so that the plain kernels keep their registers): non-flat
        // reductions only, the same tile / stage choice as the plain launch would get.
        MI355_REQUIRE(!flat, "conv_gather: epilogue statistics need C %% 64 == 0");
        constexpr int ROW3N_B = 64 * BK * 2;
        static bool st_attr = false;
        if (!st_attr) {
            (void)hipFuncSetAttribute((const void*)conv_gather_row3_kernel<2, 2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ROW3_A + 2 * ROW3N_B);
            (void)hipFuncSetAttribute((const void*)conv_gather_row3_kernel<2, 2, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ROW3_A + 2 * ROW3_B);
            (void)hipFuncSetAttribute((const void*)conv_gather_ring_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * NST);
            (void)hipFuncSetAttribute((const void*)conv_gather_kernel<false, 2, 2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NST);
            (void)hipFuncSetAttribute((const void*)conv_gather_kernel<false, 2, 2, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE2);
            st_attr = true;
        }
        if (narrow && row3) {
            if (many) hipLaunchKernelGGL((conv_gather_row3_kernel<2, 1, 2, true>), g, b256, ROW3_A + ROW3N_B, st, L);
            else      hipLaunchKernelGGL((conv_gather_row3_kernel<2, 2, 2, true>), g, b256, 2 * ROW3_A + 2 * ROW3N_B, st, L);
        } else if (narrow && !single && ring_mode && max_ksteps >= 4) {
            hipLaunchKernelGGL(conv_gather_ring_kernel<true>, g, b256, 3 * NST, st, L);
        } else if (narrow) {
            if (single) hipLaunchKernelGGL((conv_gather_kernel<false, 1, 2, 2, true>), g, b256, NST, st, L);
            else        hipLaunchKernelGGL((conv_gather_kernel<false, 2, 2, 2, true>), g, b256, 2 * NST, st, L);
        } else if (row3) {
            if (many) hipLaunchKernelGGL((conv_gather_row3_kernel<2, 1, 4, true>), g, b256, ROW3_A + ROW3_B, st, L);
            else      hipLaunchKernelGGL((conv_gather_row3_kernel<2, 2, 4, true>), g, b256, 2 * ROW3_A + 2 * ROW3_B, st, L);
        } else if (single) {
            hipLaunchKernelGGL((conv_gather_kernel<false, 1, 2, 4, true>), g, b256, EPI_BYTES, st, L);
        } else {
            hipLaunchKernelGGL((conv_gather_kernel<false, 2, 2, 4, true>), g, b256, 2 * STAGE2, st, L);
        }
        return check_launch("conv_gather_kernel<statistics>");
    }
    if (narrow && row3) {
        constexpr int ROW3N_B = 64 * BK * 2;
        if (many) hipLaunchKernelGGL((conv_gather_row3_kernel<2, 1, 2>), g, b256, ROW3_A + ROW3N_B, st, L);
        else      hipLaunchKernelGGL((conv_gather_row3_kernel<2, 2, 2>), g, b256, 2 * ROW3_A + 2 * ROW3N_B, st, L);
        return check_launch("conv_gather_row3_kernel<narrow>");
    }
    if (narrow && !single && !flat && ring_mode && max_ksteps >= 4) {
        static bool ring_attr = false;
        if (!ring_attr) {
            (void)hipFuncSetAttribute((const void*)conv_gather_ring_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * NST);
            (void)hipFuncSetAttribute((const void*)conv_gather_ring_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * NST);
            ring_attr = true;
        }
        bool one1 = one_tap_mode();
        for (int i = 0; i < in.nprob && one1; ++i) one1 = in.prob[i].TR * in.prob[i].TS <= 1;
        if (one1) hipLaunchKernelGGL((conv_gather_ring_kernel<false, true>), g, b256, 3 * NST, st, L);
        else      hipLaunchKernelGGL((conv_gather_ring_kernel<false, false>), g, b256, 3 * NST, st, L);
        return check_launch("conv_gather_ring_kernel");
    }
    if (narrow) {
        if (single) {
            bool one = one_tap_mode() && !flat;
            for (int i = 0; i < in.nprob && one; ++i) one = in.prob[i].TR * in.prob[i].TS <= 1;
            if (flat)     hipLaunchKernelGGL((conv_gather_kernel<true, 1, 2, 2>), g, b256, NST, st, L);
            else if (one) hipLaunchKernelGGL((conv_gather_kernel<false, 1, 2, 2, false, true>), g, b256, NST, st, L);
            else          hipLaunchKernelGGL((conv_gather_kernel<false, 1, 2, 2>), g, b256, NST, st, L);
        } else {
            if (flat) hipLaunchKernelGGL((conv_gather_kernel<true, 2, 2, 2>), g, b256, 2 * NST, st, L);
            else      hipLaunchKernelGGL((conv_gather_kernel<false, 2, 2, 2>), g, b256, 2 * NST, st, L);
        }
    } else if (row3) {
        if (many) hipLaunchKernelGGL((conv_gather_row3_kernel<2, 1>), g, b256, ROW3_A + ROW3_B, st, L);
        else      hipLaunchKernelGGL((conv_gather_row3_kernel<2, 2>), g, b256, 2 * ROW3_A + 2 * ROW3_B, st, L);
    } else if (single) {
        if (flat) hipLaunchKernelGGL((conv_gather_kernel<true, 1, 2>), g, b256, EPI_BYTES, st, L);
        else      hipLaunchKernelGGL((conv_gather_kernel<false, 1, 2>), g, b256, EPI_BYTES, st, L);
    } else {
        if (flat) hipLaunchKernelGGL((conv_gather_kernel<true, 2, 2>), g, b256, 2 * STAGE2, st, L);
        else      hipLaunchKernelGGL((conv_gather_kernel<false, 2, 2>), g, b256, 2 * STAGE2, st, L);
    }
    if (splitk) {
        long long off = 0;
        for (int i = 0; i < in.nprob; ++i) {
            const long long n = (long long)fin[i].M * ((in.K + 7) / 8);
            hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, ws + off, ksplit,
                               L.split_stride, fin[i], fin[i].bias, in.K, in.act, in.out_f32, in.flags & 1);
            off += (long long)fin[i].M * in.K;
        }
    }
    return check_launch("conv_gather_kernel");
}
