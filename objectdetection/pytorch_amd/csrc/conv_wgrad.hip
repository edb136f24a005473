// Weight gradient of the gather conv on gfx950 MFMA, grouped over problems sharing a weight:
//
//   dw[k][tap][c] += sum_m dy[m][k] * in[m @ tap][c]
//
// GEMM view: rows = K output channels, cols = C input channels of ONE tap, reduction = output
// pixels m.  Both operands are stored pixel-major ([m][channel]), i.e. the reduction index is
// the SLOW one, so the MFMA fragments (8 consecutive reduction elements per lane) are fetched
// with ds_read_b64_tr_b16 (hardware transposed LDS read; semantics probed in
// tools/probe_gfx950.hip [2]).  Tiles: 128 (k) x 128 (c) per block, 64 pixels per stage,
// LDS-DMA staging (double-buffered), XOR swizzle on the 32-B column groups so that the
// transposed reads of a 32-lane half hit 32 distinct 8-B cells.  The pixel range is split over
// blocks (split_rows per block) and partial tiles are reduced with fp32 atomics issued as
// 256-B contiguous row segments from an LDS-staged epilogue.
//
// Replaces the autograd weight-gradient of nn.Conv2d for every conv of
// architectures/retinanet.py:272-287 (backbone, FPN, heads).
#include "conv_common.h"

#include <type_traits>

using namespace mi355;

namespace {

constexpr int TK = 128, TC = 128, RM = 64;
constexpr int EPI_LD = TC + 4;
constexpr int TAB_ROWS = 512;                             // pixel-row table: 512 rows x 16 B behind the stages
// NG = wave groups per block.  Each group of 4 waves owns 64 of the stage's 64*NG pixel rows and the
// full 128x128 tile; the groups' partial tiles are added through LDS before the block writes ONE
// partial tile, i.e. half the split-K slabs (or atomics) at the same number of resident waves.
constexpr int stage_bytes(int ng) { return (TK + TC) * RM * ng * 2; }           // 32 KiB per group
constexpr int lds_bytes(int ng) {
    return (2 * stage_bytes(ng) + TAB_ROWS * 16 > TK * EPI_LD * 4) ? 2 * stage_bytes(ng) + TAB_ROWS * 16 : TK * EPI_LD * 4;
}

constexpr int MAX_SORTED = 64;

struct WProb {
    const unsigned short* in;
    const unsigned short* dy;
    int N, H, W, C, P, Q, sh, sw;
    int TR, TS, dh0, dhs, dw0, dws;
    int r0, rs, s0, ss, S;
    int M, in_bytes, dy_bytes;
    int chunk_begin;
    int in_sN, dy_sN;
    FastDiv dPQ, dQ;
    // where this problem's partial tiles go: slab `local chunk` of its own workspace (slab mode), or the gradient
    // itself (atomic mode).  Problems of a shared-weight launch point into ONE workspace (slabs chunk_begin ..);
    // problems of a multi-layer launch (same shape, different layers) each have their own.
    float* dw;
    float* colsum;
};

struct WLaunch {
    WProb p[MI355_CONV_MAX_PROBLEMS];
    long long slab_stride;      // > 0: plain stores into slab `chunk` (dw + chunk*slab_stride), reduced later
    int nprob, K, T, k_tiles, c_tiles, ntaps, split_rows;
    // grouped launches: chunks sorted by length (longest first), so that the short chunks of the small
    // pyramid levels fill the tail of the launch instead of opening a second round of full-length blocks
    int ub;                                     // XCD-local unit size in logical blocks (1: none)
    int xs;                                     // > 0: XCD-aware block order in waves of xs blocks (wg_logical_block)
    int wide;                                   // slab mode, every dw / slab stride 16-B aligned: 16-B epilogue stores
    int nsorted;                                // 0: chunk -> problem by chunk_begin (single problem)
    unsigned char ch_prob[MAX_SORTED];
    int ch_mbegin[MAX_SORTED];
};

// Transposed LDS read as opaque inline asm.  hipcc orders a visible LDS read behind EVERY pending
// LDS-DMA (`s_waitcnt vmcnt(0)`: it cannot tell the two stage buffers apart), which serialised the
// prefetch of stage t+1 with the MFMAs of stage t.  The asm form is invisible to that pass; the stage
// being read was retired by the `vmcnt(0)` + barrier that ended the previous iteration, and its own
// completion is awaited with the explicit `lgkmcnt(0)` below.
template <int OFF>
__device__ __forceinline__ s4v lds_tr(unsigned addr) {
    s4v v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);      // keep the MFMAs behind the wait (guide §5.4 rule 18)
}

// Slab store of four columns; nt: the slab is written once and read once by the parameter-gradient table kernel, after the
// launch - kept out of the way of the operand rows that the other blocks of the chunk still want from L2.
__device__ __forceinline__ void slab_store4(float* dst, const float4 v, const bool nt) {
    if (nt) __builtin_nontemporal_store((f4v){v.x, v.y, v.z, v.w}, reinterpret_cast<f4v*>(dst));
    else    *reinterpret_cast<float4*>(dst) = v;
}

// XCD-aware block order.  Consecutive blockIdx go round-robin over the 8 XCDs (one L2 each), and the (tap / kernel
// row, k tile, c tile) blocks of a pixel chunk - consecutive LOGICAL ids - read the same dy / activation rows: in plain
// order every chunk's rows are fetched into all eight L2s (PMC: 391 MB read per head-tower launch for 90 MB of operands).
// Within each wave of `xs` blocks (the number resident at once: the dispatch order stays the logical order at that
// granularity, so the longest-first chunk order of a grouped launch still leaves the shortest chunks for the tail) XCD x
// takes a contiguous range of logical ids, i.e. whole chunks.  MI355_WG_XCD=0: plain order.
__device__ __forceinline__ int wg_logical_block(int bid, int nblocks, int xs) {
    if (xs <= 0) return bid;
    const int base = (bid / xs) * xs;
    const int n = min(xs, nblocks - base);
    return base + xcd_swizzle(bid - base, n);
}

// KH (K <= 64, NG 1): the upper half of the k-tile is empty; the four waves take 64 x 32 column strips of the
// 64 x 128 tile (NJ = 2 fragment columns) instead of two of them multiplying zeros.
template <bool FLAT, int NG, bool KH = false>
__global__ __launch_bounds__(256 * NG, 2) void conv_wgrad_kernel(const WLaunch L) {
    static_assert(!KH || NG == 1, "K <= 64 form: one wave group");
    constexpr int NJ = KH ? 2 : 4;
    constexpr int NTHREADS = 256 * NG, RMS = RM * NG;   // rows per stage
    constexpr int STAGE_BYTES = stage_bytes(NG), TAB_OFF = 2 * STAGE_BYTES, TAB_STEPS = TAB_ROWS / RMS;
    constexpr int ROWSTEP = NTHREADS / 16;              // rows covered by one DMA instruction of the block
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ float s_cs[TK];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    // block -> (chunk, tap, k_tile, c_tile).  Consecutive blockIdx round-robin over the 8 XCDs (one L2
    // each).  When the launch divides into a multiple of 8 equal units of `ub` consecutive logical
    // blocks that read the same rows (a whole pixel chunk, or its per-tap tile group), unit u runs on
    // XCD u % 8, so that its rows are fetched into ONE L2 instead of all eight (ub == 1: plain order).
    int b = blockIdx.x;
    if (L.ub > 1) {
        const int idx = blockIdx.x >> 3;
        b = ((blockIdx.x & 7) + 8 * (idx / L.ub)) * L.ub + idx % L.ub;
    } else {
        b = wg_logical_block(blockIdx.x, gridDim.x, L.xs);      // no exact unit split: contiguous logical ranges per XCD
    }
    const int c_tile = b % L.c_tiles;  b /= L.c_tiles;
    const int k_tile = b % L.k_tiles;  b /= L.k_tiles;
    const int tap = b % L.ntaps;       b /= L.ntaps;
    const int chunk = b;

    WProb P = L.p[0];
    int m_begin;
    if (L.nsorted > 0) {
        const int pi = L.ch_prob[chunk];
#pragma unroll
        for (int i = 1; i < MI355_CONV_MAX_PROBLEMS; ++i)
            if (i == pi) P = L.p[i];
        m_begin = L.ch_mbegin[chunk];
    } else {
#pragma unroll
        for (int i = 1; i < MI355_CONV_MAX_PROBLEMS; ++i)
            if (i < L.nprob && chunk >= L.p[i].chunk_begin) P = L.p[i];
        m_begin = (chunk - P.chunk_begin) * L.split_rows;
    }
    const int m_end = min(P.M, m_begin + L.split_rows);
    const int lchunk_id = m_begin / L.split_rows;       // slab of this block inside the problem's workspace
    const int k0 = k_tile * TK, c0 = c_tile * TC;

    const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(P.in, (unsigned)P.in_bytes);
    const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(P.dy, (unsigned)P.dy_bytes);

    // staging: thread owns LDS (row = tid/16 + 16 i, slot = tid%16) and loads logical chunk
    // slot ^ (f(row) << 1), f(row) = (row & 3) | ((row >> 3) & 1) << 2.
    const int srow = tid >> 4, slot = tid & 15;
    const int fsw = ((srow & 3) | (((srow >> 3) & 1) << 2)) << 1;
    const int lchunk = slot ^ fsw;
    const int ky = k0 + lchunk * 8;                    // dy channel of this thread's chunk
    const bool ky_ok = ky + 8 <= L.K;
    // input-side column chunk: one tap per block (tap-uniform) or, FLAT (C < 128), the tile's
    // 128 columns run over (tap, channel) so that small-C layers fill the MFMA tile
    int cx, dh, dw;
    bool cx_ok;
    if (FLAT) {
        const int cpt = P.C >> 3;
        const int J = c_tile * 16 + lchunk;
        const int tp = J / cpt;
        cx = (J - tp * cpt) * 8;
        const int ti = tp / P.TS, tj = tp - ti * P.TS;
        dh = P.dh0 + ti * P.dhs;
        dw = P.dw0 + tj * P.dws;
        cx_ok = tp < P.TR * P.TS;
    } else {
        const int ti = tap / P.TS, tj = tap - ti * P.TS;
        dh = P.dh0 + ti * P.dhs;
        dw = P.dw0 + tj * P.dws;
        cx = c0 + lchunk * 8;
        cx_ok = cx + 8 <= P.C;
    }

    // Pixel-row table: the (image, row, column) decomposition of the block's pixel rows is computed once
    // per 8 stages (2 rows per thread) instead of 4 rows per thread and stage; a stage then costs a
    // 16-B LDS read, two bound checks and two adds per row.
    //   .x = element offset of input pixel (p*sh, q*sw)   .y = (p*sh) << 16 | (q*sw)   .z = dy row offset
    int4* tab = reinterpret_cast<int4*>(smem + TAB_OFF);
    auto fill = [&](int step0) {
        for (int e = tid; e < TAB_ROWS; e += NTHREADS) {
            const int m = m_begin + step0 * RMS + e;
            int4 t = make_int4(0, 0x7fff0000, -1, 0);          // row beyond the chunk: never in range
            if (m < m_end) {
                const unsigned n = fdiv((unsigned)m, P.dPQ);
                const unsigned rem = (unsigned)m - n * (unsigned)(P.P * P.Q);
                const unsigned p = fdiv(rem, P.dQ);
                const unsigned q = rem - p * (unsigned)P.Q;
                const int ih0 = (int)p * P.sh, iw0 = (int)q * P.sw;
                t.x = (int)n * P.in_sN + (ih0 * P.W + iw0) * P.C;
                t.y = (ih0 << 16) | iw0;
                t.z = (int)n * P.dy_sN + (int)rem * L.K;
            }
            tab[e] = t;
        }
    };
    const int tapoff = (dh * P.W + dw) * P.C + cx;
    auto issue = [&](int step, int buf) {
        unsigned char* sy = smem + buf * STAGE_BYTES;
        unsigned char* sx = sy + TK * RMS * 2;
        int4 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = tab[(step & (TAB_STEPS - 1)) * RMS + srow + ROWSTEP * i];   // before any DMA
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ih = (t[i].y >> 16) + dh, iw = (t[i].y & 0xffff) + dw;
            const unsigned offy = (ky_ok && t[i].z >= 0) ? (unsigned)(t[i].z + ky) * 2u : OOB_OFFSET;
            const unsigned offx = (cx_ok && (unsigned)ih < (unsigned)P.H && (unsigned)iw < (unsigned)P.W)
                                      ? (unsigned)(t[i].x + tapoff) * 2u : OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (LDS_PTR(void))(sy + (i * NTHREADS + wv * 64) * 16),
                                                     16, offy, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (LDS_PTR(void))(sx + (i * NTHREADS + wv * 64) * 16),
                                                     16, offx, 0, 0, 0);
        }
    };

    f4v acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f4v){0.f, 0.f, 0.f, 0.f};

    // transposed-read addressing (16-lane group g, lane 4q+p): row 8g + q (+4), columns 4p..4p+3
    const int grp = wv >> 2;                           // pixel-row group of this wave
    const int wk = KH ? 0 : (wv >> 1) & 1, wc = KH ? wv & 3 : wv & 1;
    const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int f = (q4 | ((g & 1) << 2)) << 1;
    const int row_lo = 8 * g + q4;
    // logical 16-B chunk of (tile t of this wave, p4): 8*wk + 2t + (p4 >> 1) for dy, 8*wc + ... for x
    const int half8 = (p4 & 1) * 8;
    constexpr int XOFF = TK * RMS * 2;                 // x tile behind the dy tile of a stage
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_PTR(void))smem;
    unsigned ady[4], adx[4];                           // byte offsets inside a stage (rows 0..31: +0 / +4 rows)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ady[t] = (grp * RM + row_lo) * 256 + ((8 * wk + 2 * t + (p4 >> 1)) ^ f) * 16 + half8;
        adx[t] = (grp * RM + row_lo) * 256 + (((KH ? 4 : 8) * wc + 2 * (t & (NJ - 1)) + (p4 >> 1)) ^ f) * 16 + half8;
    }

    // column sums of dy (bias / BN-beta gradient): only the blocks owning (tap 0, c-tile 0) and
    // only their wc == 0 waves (the other column-waves hold the same dy fragments)
    const bool do_colsum = P.colsum != nullptr && c_tile == 0 && tap == 0 && wc == 0;
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    const int nsteps = (m_end - m_begin + RMS - 1) / RMS;
    fill(0);
    __syncthreads();
    if (nsteps > 0) issue(0, 0);
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        if (((step + 1) & (TAB_STEPS - 1)) == 0 && step + 1 < nsteps) {
            fill(step + 1);              // every table read of stages <= step happened before the last barrier
            __syncthreads();
        }
        if (step + 1 < nsteps) issue(step + 1, buf ^ 1);
        {
            const unsigned sb = lds0 + buf * STAGE_BYTES;
            s8v af[2][4], bf[2][NJ];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const s4v a0 = lds_tr<0>(sb + ady[t]), a1 = lds_tr<1024>(sb + ady[t]);
                af[0][t] = (s8v){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                if (t < NJ) {
                    const s4v b0 = lds_tr<XOFF>(sb + adx[t]), b1 = lds_tr<XOFF + 1024>(sb + adx[t]);
                    bf[0][t & (NJ - 1)] = (s8v){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                }
            }
            lds_wait();
#pragma unroll
            for (int t = 0; t < 4; ++t) {              // rows 32..63 fly while rows 0..31 are on the matrix cores
                const s4v a0 = lds_tr<8192>(sb + ady[t]), a1 = lds_tr<8192 + 1024>(sb + ady[t]);
                af[1][t] = (s8v){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                if (t < NJ) {
                    const s4v b0 = lds_tr<XOFF + 8192>(sb + adx[t]), b1 = lds_tr<XOFF + 8192 + 1024>(sb + adx[t]);
                    bf[1][t & (NJ - 1)] = (s8v){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                }
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                if (sub == 1) lds_wait();
                if (do_colsum) {                       // wave-uniform
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int e = 0; e < 8; ++e) csum[t] += bf16_to_f32((unsigned short)af[sub][t][e]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[sub][i], bf[sub][j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: the groups' partial tiles (and column sums) are added through LDS in group order,
    //      then written as 256-B contiguous row segments (plain slab stores, or fp32 atomics)
    float* epi = reinterpret_cast<float*>(smem);
    const int fr = lane & 15, fq = lane >> 4;
    float cs_v[4] = {0.f, 0.f, 0.f, 0.f};
    if (do_colsum) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = csum[t];
            v += __shfl_xor(v, 16, 64);                // the 4 lane groups hold disjoint pixels
            v += __shfl_xor(v, 32, 64);
            cs_v[t] = v;
            if (NG > 1 && grp == 1 && lane < 16) s_cs[wk * 64 + t * 16 + lane] = v;
        }
    }
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    epi[(wk * 64 + i * 16 + fq * 4 + r) * EPI_LD + wc * (KH ? 32 : 64) + j * 16 + fr] = acc[i][j][r];
    }
    __syncthreads();
    if (NG > 1) {
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        epi[(wk * 64 + i * 16 + fq * 4 + r) * EPI_LD + wc * (KH ? 32 : 64) + j * 16 + fr] += acc[i][j][r];
        }
        __syncthreads();
    }
    if (do_colsum && grp == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = k0 + wk * 64 + t * 16 + (lane & 15);
            if (lane < 16 && k < L.K) {
                const float v = cs_v[t] + (NG > 1 ? s_cs[wk * 64 + t * 16 + lane] : 0.0f);
                if (L.slab_stride > 0) P.colsum[(long long)lchunk_id * L.K + k] = v;   // slab mode: [chunk][K]
                else atomicAdd(P.colsum + k, v);
            }
        }
    }
    const long long wrow = (long long)L.T * P.C;
    if (L.wide) {
        // 16-B stores: lane & 31 owns four columns (C % 4 == 0: one tap, contiguous), the two halves of a wave two rows
        const int c = 4 * (lane & 31);
        long long co = -1;
        if (FLAT) {
            const int J8 = c_tile * TC + c;
            const int tp = J8 / P.C, cc = J8 - tp * P.C;
            const int ti = tp / P.TS, tj = tp - ti * P.TS;
            if (tp < P.TR * P.TS) co = (long long)((P.r0 + ti * P.rs) * P.S + (P.s0 + tj * P.ss)) * P.C + cc;
        } else {
            const int ti = tap / P.TS, tj = tap - ti * P.TS;
            if (c0 + c < P.C) co = (long long)((P.r0 + ti * P.rs) * P.S + (P.s0 + tj * P.ss)) * P.C + c0 + c;
        }
        if (co >= 0) {
            for (int rr = 2 * wv + (lane >> 5); rr < TK; rr += 8 * NG) {
                const int k = k0 + rr;
                if (k >= L.K) continue;
                slab_store4(P.dw + k * wrow + co + (long long)lchunk_id * L.slab_stride,
                            *reinterpret_cast<const float4*>(epi + rr * EPI_LD + c), L.wide == 2);
            }
        }
        return;
    }
    long long coff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {                      // weight-row offset of this lane's columns
        const int c = h * 64 + lane;
        coff[h] = -1;
        if (FLAT) {
            const int J8 = c_tile * TC + c;
            const int tp = J8 / P.C, cc = J8 - tp * P.C;
            const int ti = tp / P.TS, tj = tp - ti * P.TS;
            if (tp < P.TR * P.TS) coff[h] = (long long)((P.r0 + ti * P.rs) * P.S + (P.s0 + tj * P.ss)) * P.C + cc;
        } else {
            const int ti = tap / P.TS, tj = tap - ti * P.TS;
            if (c0 + c < P.C) coff[h] = (long long)((P.r0 + ti * P.rs) * P.S + (P.s0 + tj * P.ss)) * P.C + c0 + c;
        }
    }
    for (int rr = wv; rr < TK; rr += 4 * NG) {
        const int k = k0 + rr;
        if (k >= L.K) break;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (coff[h] >= 0) {
                float* dst = P.dw + k * wrow + coff[h];
                const float v = epi[rr * EPI_LD + h * 64 + lane];
                if (L.slab_stride > 0) dst[(long long)lchunk_id * L.slab_stride] = v;   // deterministic split-K
                else atomicAdd(dst, v);
            }
    }
}


// ---------------------------------------------------------------------------------------------
// ROW3 weight gradient: 3x3 / stride 1 / "same" convs with C % 64 == 0 and image width W | 64.
// A block owns ONE kernel row r, 128 output channels and 64 input channels, and accumulates the three
// column taps at once: the dy tile (64 pixels x 128 k) is staged once for the three taps and the
// activation tile once, two rows longer (66 x 64 channels); tap s reads it shifted by s rows through
// ds_read_b64_tr_b16 (pixels whose column q + s - 1 leaves the image read the zero row).  Per 64-pixel
// stage that is 25 KiB of LDS-DMA for 3 x (128 x 64 x 64) MACs instead of 3 x 32 KiB: the generic
// kernel's dominant cost (vector-memory issue + L2->LDS traffic, and the 8-XCD re-fetch of the rows).
// LDS image of the activation tile: 128-B rows, 16-B chunk c at position c ^ (gx(row) << 1),
// gx(row) = bit1(row) | bit3(row) << 1: the 8 rows of a 32-lane half of a transposed read (any of the
// three shifts) land in 8 distinct 32-B bank groups.
constexpr int R3_TC = 64, R3_XROWS = 72, R3_DY_BYTES = RM * TK * 2;      // x tile: R3_XROWS * R3_TC * 2 bytes
constexpr int R3_STAGE = 26624;                           // 16384 + 9216, 1 KiB aligned
constexpr int R3_TABN = 520, R3_TAB_OFF = 2 * R3_STAGE;
constexpr int R3_EPI_LD = R3_TC + 4;
constexpr int R3_LDS = (R3_TAB_OFF + R3_TABN * 16 > TK * R3_EPI_LD * 4) ? R3_TAB_OFF + R3_TABN * 16 : TK * R3_EPI_LD * 4;

__device__ __forceinline__ int r3_gx(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

// DYN: any image width.  The column of staged pixel i then changes from stage to stage, so the edge pixels (q == 0
// for the left tap, q == W - 1 for the right one) come from the pixel-row table (.w) instead of fixed fragment
// addresses: a lane fetches the flags of its four fragment pixels one stage ahead and swaps in the zero row.
// KH (K <= 64: stem-side layers, the regression output conv): the upper 64 rows of the k-tile are empty, so the four
// waves split the 64 x 64 tile into 64 x 16 column strips (NJ = 1 fragment column each) instead of two of them
// multiplying zeros.
template <bool DYN, bool KH>
__global__ __launch_bounds__(256, 2) void conv_wgrad_row3_kernel(const WLaunch L) {
    constexpr int NJ = KH ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int b = wg_logical_block(blockIdx.x, gridDim.x, L.xs);
    const int c_tile = b % L.c_tiles;  b /= L.c_tiles;
    const int k_tile = b % L.k_tiles;  b /= L.k_tiles;
    const int kr = b % 3;              b /= 3;          // kernel row of this block
    const int chunk = b;

    WProb P = L.p[0];
    int m_begin;
    if (L.nsorted > 0) {
        const int pi = L.ch_prob[chunk];
#pragma unroll
        for (int i = 1; i < MI355_CONV_MAX_PROBLEMS; ++i)
            if (i == pi) P = L.p[i];
        m_begin = L.ch_mbegin[chunk];
    } else {
#pragma unroll
        for (int i = 1; i < MI355_CONV_MAX_PROBLEMS; ++i)
            if (i < L.nprob && chunk >= L.p[i].chunk_begin) P = L.p[i];
        m_begin = (chunk - P.chunk_begin) * L.split_rows;
    }
    const int m_end = min(P.M, m_begin + L.split_rows);
    const int lchunk_id = m_begin / L.split_rows;
    const int k0 = k_tile * TK, c0 = c_tile * R3_TC;
    const int dh = P.dh0 + kr;                          // (dhs == 1)

    const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(P.in, (unsigned)P.in_bytes);
    const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(P.dy, (unsigned)P.dy_bytes);

    // dy staging as in the generic kernel: row = tid/16 + 16 i, slot = tid%16, logical chunk slot ^ (f << 1)
    const int srow = tid >> 4, slot = tid & 15;
    const int fsw = ((srow & 3) | (((srow >> 3) & 1) << 2)) << 1;
    const int ky = k0 + (slot ^ fsw) * 8;
    const bool ky_ok = ky + 8 <= L.K;
    // activation staging: row j = tid/8 + 32 i (i = 0, 1; wave 0 also j = 64 + tid/8), slot = tid%8
    const int xrow = tid >> 3, xslot = tid & 7;
    const int cx = c0 + ((xslot ^ (r3_gx(xrow) << 1)) << 3);    // gx(xrow + 32 i) == gx(xrow); rows 64.. below
    const int cx64 = c0 + ((xslot ^ (r3_gx(64 + xrow) << 1)) << 3);

    // pixel-row table, entry e + 1 <-> pixel m_begin + 64 * step0 + e, e = -1 .. 518:
    //   .x = element offset of input pixel (n, p, q)   .y = p (or far out of range)   .z = dy row offset / -1
    int4* tab = reinterpret_cast<int4*>(smem + R3_TAB_OFF);
    auto fill = [&](int step0) {
        for (int e = tid; e < R3_TABN; e += 256) {
            const int m = m_begin + step0 * RM + e - 1;
            int4 t = make_int4(0, -0x40000000, -1, 0);
            if (m >= 0 && m < P.M) {
                const unsigned n = fdiv((unsigned)m, P.dPQ);
                const unsigned rem = (unsigned)m - n * (unsigned)(P.P * P.Q);
                const unsigned p = fdiv(rem, P.dQ);
                const unsigned q = rem - p * (unsigned)P.Q;
                t.x = (int)n * P.in_sN + ((int)p * P.W + (int)q) * P.C;
                t.y = (int)p;
                t.w = (q == 0 ? 1 : 0) | ((int)q == P.W - 1 ? 2 : 0);
                if (m < m_end && e >= 1) t.z = (int)n * P.dy_sN + (int)rem * L.K;
            }
            tab[e] = t;
        }
    };
    const int xdelta = dh * P.W * P.C;
    auto issue = [&](int step, int buf) {
        unsigned char* sy = smem + buf * R3_STAGE;
        unsigned char* sx = sy + R3_DY_BYTES;
        const int tb = (step & 7) * RM;
        int4 ty[4], tx[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) ty[i] = tab[tb + srow + 16 * i + 1];            // before any DMA
#pragma unroll
        for (int i = 0; i < 3; ++i) tx[i] = tab[min(tb + xrow + 32 * i, R3_TABN - 1)];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned offy = (ky_ok && ty[i].z >= 0) ? (unsigned)(ty[i].z + ky) * 2u : OOB_OFFSET;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (LDS_PTR(void))(sy + (i * 256 + wv * 64) * 16),
                                                     16, offy, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i < 2 || wv == 0) {                                     // rows 64..71: wave 0 only
                const int j = xrow + 32 * i;
                const bool ok = j < RM + 2 && (unsigned)(tx[i].y + dh) < (unsigned)P.H;
                const unsigned offx = ok ? (unsigned)(tx[i].x + xdelta + (i < 2 ? cx : cx64)) * 2u : OOB_OFFSET;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (LDS_PTR(void))(sx + (i * 256 + wv * 64) * 16),
                                                         16, offx, 0, 0, 0);
            }
        }
    };

    f4v acc[3][4][NJ];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[s3][i][j] = (f4v){0.f, 0.f, 0.f, 0.f};

    const int wk = KH ? 0 : wv >> 1, wc = KH ? wv : wv & 1;
    const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int f = (q4 | ((g & 1) << 2)) << 1;
    const int row_lo = 8 * g + q4;
    const int half8 = (p4 & 1) * 8;
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_PTR(void))smem;
    unsigned ady[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) ady[t] = row_lo * 256 + ((8 * wk + 2 * t + (p4 >> 1)) ^ f) * 16 + half8;
    // activation fragment addresses: [sub][h][shift][t]; pixel i = 32 sub + row_lo + 4 h reads staged row i + s;
    // W | 64 and chunks start at multiples of 64, so the image column of pixel i is i mod W
    constexpr unsigned ZROW = R3_DY_BYTES + (R3_XROWS - 1) * 128;       // staged row 71: always zeros
    unsigned adx[2][2][3][NJ];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = sub * 32 + row_lo + 4 * h;
            const int q = i & (P.W - 1);
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const int row = i + s3;
                const bool edge = !DYN && ((s3 == 0 && q == 0) || (s3 == 2 && q == P.W - 1));
#pragma unroll
                for (int t = 0; t < NJ; ++t) {
                    const int lc = (KH ? 2 : 4) * wc + 2 * t + (p4 >> 1);
                    adx[sub][h][s3][t] = edge ? ZROW + half8
                                              : R3_DY_BYTES + row * 128 + ((lc ^ (r3_gx(row) << 1)) << 4) + half8;
                }
            }
        }

    const bool do_colsum = P.colsum != nullptr && c_tile == 0 && kr == 0 && wc == 0;
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    const int nsteps = (m_end - m_begin + RM - 1) / RM;
    // DYN: edge flags of this lane's fragment pixels (sub, h) of a stage, 2 bits each
    auto edge_flags = [&](int step) {
        const int tb = (step & 7) * RM + row_lo + 1;
        return tab[tb].w | (tab[tb + 4].w << 2) | (tab[tb + 32].w << 4) | (tab[tb + 36].w << 6);
    };
    fill(0);
    __syncthreads();
    int ef_next = DYN ? edge_flags(0) : 0;          // (table reads before the DMA issue: hipcc drains vmcnt in front
    if (nsteps > 0) issue(0, 0);                    //  of a plain LDS read that follows a buffer_load ... lds)
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        const int ef = ef_next;
        if (((step + 1) & 7) == 0 && step + 1 < nsteps) {
            fill(step + 1);
            __syncthreads();
        }
        if (step + 1 < nsteps) {
            if (DYN) ef_next = edge_flags(step + 1);       // (read before this step's closing barrier, like issue's rows)
            issue(step + 1, buf ^ 1);
        }
        const unsigned sb = lds0 + buf * R3_STAGE;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            s8v af[4], bf[3][NJ];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned a = sb + ady[t] + sub * 8192;
                const s4v a0 = lds_tr<0>(a), a1 = lds_tr<1024>(a);
                af[t] = (s8v){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            }
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
                for (int t = 0; t < NJ; ++t) {
                    unsigned x0 = adx[sub][0][s3][t], x1 = adx[sub][1][s3][t];
                    if (DYN && s3 != 1) {
                        const int bit = s3 == 0 ? 1 : 2;
                        x0 = (ef >> (4 * sub)) & bit ? ZROW + half8 : x0;
                        x1 = (ef >> (4 * sub + 2)) & bit ? ZROW + half8 : x1;
                    }
                    const s4v b0 = lds_tr<0>(sb + x0), b1 = lds_tr<0>(sb + x1);
                    bf[s3][t] = (s8v){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                }
            lds_wait();
            if (do_colsum) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 8; ++e) csum[t] += bf16_to_f32((unsigned short)af[t][e]);
            }
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[s3][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[s3][j], acc[s3][i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    if (do_colsum) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = csum[t];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int k = k0 + wk * 64 + t * 16 + (lane & 15);
            if (lane < 16 && k < L.K) {
                if (L.slab_stride > 0) P.colsum[(long long)lchunk_id * L.K + k] = v;
                else atomicAdd(P.colsum + k, v);
            }
        }
    }
    // ---- epilogue, one column tap at a time through LDS: 256-B contiguous row segments
    float* epi = reinterpret_cast<float*>(smem);
    const int fr = lane & 15, fq = lane >> 4;
    const long long wrow = (long long)L.T * P.C;
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    epi[(wk * 64 + i * 16 + fq * 4 + r) * R3_EPI_LD + wc * (KH ? 16 : 32) + j * 16 + fr] = acc[s3][i][j][r];
        __syncthreads();
        const long long coff = (long long)((P.r0 + kr * P.rs) * P.S + (P.s0 + s3 * P.ss)) * P.C + c0 + lane;
        if (L.wide) {                                   // 16-B stores: four 256-B row segments per instruction
            const int l16 = lane & 15;
            for (int rr = 4 * wv + (lane >> 4); rr < TK; rr += 16) {
                const int k = k0 + rr;
                if (k >= L.K) continue;
                slab_store4(P.dw + k * wrow + (coff - lane) + 4 * l16 + (long long)lchunk_id * L.slab_stride,
                            *reinterpret_cast<const float4*>(epi + rr * R3_EPI_LD + 4 * l16), L.wide == 2);
            }
        } else {
            for (int rr = wv; rr < TK; rr += 4) {
                const int k = k0 + rr;
                if (k >= L.K) break;
                float* dst = P.dw + k * wrow + coff;
                const float v = epi[rr * R3_EPI_LD + lane];
                if (L.slab_stride > 0) dst[(long long)lchunk_id * L.slab_stride] = v;
                else atomicAdd(dst, v);
            }
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------
// ROW3X weight gradient (round 3): the row-of-taps kernel as ONE eight-wave block per CU with a THREE-stage LDS-DMA
// ring.  Why: the four-wave form above runs two blocks per CU (198 VGPRs, 61.5 KiB), each with one stage of prefetch:
// its waves sit at the stage-closing barrier / vmcnt(0) for 47-64 % of their cycles (profiles/r01_pmc_sq_waves.txt) and
// the ablation of DESIGN.md section 4 prices the un-hidden LDS-DMA at 32 of its 138 us on the head towers.  Here the
// DMA of stage t + 2 is issued while stage t is on the matrix cores (counted `s_waitcnt vmcnt`, raw `s_barrier`, every
// LDS access of the loop as opaque asm so that hipcc does not drain the DMA queue in front of it: guide section 5,
// "Pipelining across barriers"), and the eight waves share ONE staging of a 128 k x 128 c x 3-tap tile: 34 KiB of
// L2 -> LDS fill per stage instead of 2 x 25 KiB for the same 2 x 48 MFMAs per SIMD.
//   block tile : kernel row r, 128 output channels, 128 input channels, the three column taps
//   waves      : 2 (k) x 4 (c); a wave owns 64 k x 32 c x 3 taps = 96 accumulator VGPRs, as in the four-wave form
//   stage      : 64 pixels: dy 64 x 128 k (16 KiB) + activation rows 72 x 128 c (18 KiB; 66 used, row 71 = zeros)
//   LDS        : 4 stages (136 KiB) + a 1024-entry pixel-row ring (8 KiB, refilled 8 stages at a time)
// Both LDS images have 256-B rows with the 16-B chunk c at slot c ^ (f(row) << 1), f(row) = row & 3 | bit3(row) << 2:
// the 8 rows of a 32-lane half of a transposed read (rows r0 .. r0 + 3 and r0 + 8 .. r0 + 11, any r0) then land in 8
// distinct 32-B bank groups.  Widths dividing the 64-pixel stage only (fixed edge addresses); C % 128 == 0, K >= 128.
constexpr int X3_TC = 128, X3_XROWS = 72;
constexpr int X3_DY_BYTES = RM * TK * 2;                    // 16 KiB
constexpr int X3_STAGE = X3_DY_BYTES + X3_XROWS * X3_TC * 2;    // 34 KiB
constexpr int X3_NST = 4;                                   // ring depth: X3_NST - 1 stages in flight beside the one on the matrix cores
constexpr int X3_TABN = 1024, X3_TAB_OFF = X3_NST * X3_STAGE;
constexpr int X3_EPI_LD = X3_TC + 4;
constexpr int X3_LDS = X3_TAB_OFF + X3_TABN * 8;
static_assert(TK * X3_EPI_LD * 4 <= X3_TAB_OFF, "epilogue staging must fit in the ring");

typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ i2v lds_read8(unsigned addr) {
    i2v v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write8(unsigned addr, i2v v) {
    asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ i4v lds_read16(unsigned addr) {
    i4v v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write16(unsigned addr, i4v v) {
    asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ int x3_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// ABL: diagnostic builds (MI355_X3_ABLATE, tools/bench_head_wgrad.py; 0 = the kernel): 1 no LDS-DMA in the loop, 2 no fragment
// reads in the loop, 4 no epilogue stores, 8 no MFMAs (sums: 3, 7, 9, 10, 11).  Results are garbage for ABL != 0.
// Head towers alone, us: 0: 132 | 1: 118 | 2: 114 | 3: 97 | 4: 125 | 7: 90 | 8: 93 | 9: 71 | 10: 69 | 11: 45 - the parts ADD
// (skeleton 45 incl. a second round of 20 short blocks, MFMAs +45 .. 52, reads +26, DMA +24).  What round 3 built on top of
// this loop, measured and took out again:
//  * the three column taps formed in registers from ONE set of transposed reads (14 reads per 24 MFMAs instead of 20,
//    v_alignbit + edge masks, ~30 VALU ops per group): 131 -> 138 - 142 us.  The LDS array is not the limit (a transposed read
//    is 2 LDS cycles), the vector issue slots beside the MFMAs are.
//  * ping-pong roles (waves w and w + 4 of a SIMD alternate a bare 48-MFMA phase with a load phase, two barriers per stage):
//    in-kernel cycle stamps show the compute phase AT its floor (800 cycles for 768 of MFMA) but a load phase of 1 370 - 1 460
//    cycles (reads alone 790 = 20 per read; pixel-row reads + offsets + five DMA pieces alone 930: a piece costs its wave
//    ~120 cycles with four waves issuing at once), so a stage takes 3 660 cycles against ~2 900 here: 134 -> 151 us.  s_setprio
//    for the loading wave: no change.  The load work of a stage (2 x 1 400 wave-cycles per SIMD) does not fit beside its
//    1 536 cycles of MFMA in either arrangement; less of it per MFMA needs a larger block tile than the register file holds.
//  * waves 4-7 issuing their DMA pieces half a stage after their SIMD partners: 140 vs 141 us.  Column sums off: no change.
// M32: the products as v_mfma_f32_32x32x16_bf16 (a wave's 64 k x 32 c x 3 taps = 3 x 2 tiles of 32 x 32).  Same operand
// bytes and the same 20 transposed reads per 32 pixels, but 12 MFMAs instead of 24: an MFMA holds the SIMD's vector issue
// port for 8 cycles whether it keeps the matrix pipe busy for 16 (16x16x32) or 32 cycles (32x32x16), and the port - shared
// by the two waves of a SIMD - is what paces this loop (the ablation times add, see above).  LDS images: 16-B chunk c of
// row r at slot c ^ ((r & 3) << 2): the two lane groups of a 32-lane half read the SAME four rows, 64 B each -> four
// distinct 64-B bank quarters; rows that differ by a multiple of four share the swizzle, so the reads of a lane differ
// by immediates (19 address registers instead of 28).
template <int ABL, bool M32 = false>
__global__ __launch_bounds__(512, 2) void conv_wgrad_row3x_kernel(const WLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform: scalar branches on the wave index
    int b = wg_logical_block(blockIdx.x, gridDim.x, L.xs);
    const int c_tile = b % L.c_tiles;  b /= L.c_tiles;
    const int k_tile = b % L.k_tiles;  b /= L.k_tiles;
    const int kr = b % 3;              b /= 3;          // kernel row of this block
    const int chunk = b;

    WProb P = L.p[0];
    int m_begin;
    if (L.nsorted > 0) {
        const int pi = L.ch_prob[chunk];
#pragma unroll
        for (int i = 1; i < MI355_CONV_MAX_PROBLEMS; ++i)
            if (i == pi) P = L.p[i];
        m_begin = L.ch_mbegin[chunk];
    } else {
#pragma unroll
        for (int i = 1; i < MI355_CONV_MAX_PROBLEMS; ++i)
            if (i < L.nprob && chunk >= L.p[i].chunk_begin) P = L.p[i];
        m_begin = (chunk - P.chunk_begin) * L.split_rows;
    }
    const int m_end = min(P.M, m_begin + L.split_rows);
    const int lchunk_id = m_begin / L.split_rows;
    const int k0 = k_tile * TK, c0 = c_tile * X3_TC;
    const int dh = P.dh0 + kr;                          // (dhs == 1)

    const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(P.in, (unsigned)P.in_bytes);
    const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(P.dy, (unsigned)P.dy_bytes);
    const unsigned lds0 = (unsigned)(uintptr_t)(LDS_PTR(void))smem;

    // staging: a DMA wave-instruction fills 4 rows of 256 B; piece pc = 8 i + wave (i = 0, 1) holds rows 4 pc .. 4 pc + 3,
    // lane -> (row 4 pc + lane / 16, slot lane % 16), source chunk slot ^ (f(row) << 1); f is the same for i = 0, 1.
    // Activation rows 64 .. 71 (pieces 16, 17): waves 0 and 1.
    auto swz = [](const int row) { return M32 ? (row & 3) << 2 : x3_f(row) << 1; };     // XOR on the 16-B chunk index
    const int srow = 4 * wv + (lane >> 4), slot = lane & 15;
    const int chunk8 = (slot ^ swz(srow)) << 3;
    const int ky = k0 + chunk8, cx = c0 + chunk8;
    const bool ky_ok = ky + 8 <= L.K;
    const int xrow2 = 64 + 4 * wv + (lane >> 4);                      // (waves 0, 1)
    const int cx2 = c0 + ((slot ^ swz(xrow2)) << 3);
    const int xdelta = dh * P.W * P.C;

    // pixel-row ring, entry (e + 1) & 1023 <-> pixel m_begin + e, e >= -1: the two buffer offsets a DMA lane adds its own
    // column to - .x = byte offset of input pixel (n, p + dh, q) (this block's kernel row folded in), .y = byte offset of
    // the dy row; OOB_OFFSET where the row is padding / outside the chunk (OOB_OFFSET + a column offset stays out of range)
    const unsigned tab0 = lds0 + X3_TAB_OFF;
    auto fill = [&](int s0) {                           // entries of stages s0 .. s0 + 7 (e = 64 s0 - 1 .. 64 s0 + 512)
        for (int i = tid; i < 514; i += 512) {
            const int e = 64 * s0 + i - 1;
            const int m = m_begin + e;
            i2v t = (i2v){(int)OOB_OFFSET, (int)OOB_OFFSET};
            if (m >= 0 && m < P.M) {
                const unsigned n = fdiv((unsigned)m, P.dPQ);
                const unsigned rem = (unsigned)m - n * (unsigned)(P.P * P.Q);
                const unsigned p = fdiv(rem, P.dQ);
                const unsigned q = rem - p * (unsigned)P.Q;
                if ((unsigned)((int)p + dh) < (unsigned)P.H)
                    t.x = (int)((unsigned)((int)n * P.in_sN + ((int)p * P.W + (int)q) * P.C + xdelta) * 2u);
                if (m < m_end && e >= 0) t.y = (int)((unsigned)((int)n * P.dy_sN + (int)rem * L.K) * 2u);
            }
            lds_write8(tab0 + (unsigned)((e + 1) & (X3_TABN - 1)) * 8u, t);
        }
    };
    // DMA of one stage in two halves: `issue_prep` reads this thread's five pixel-row entries (LDS, asm) and leaves them
    // pending; `issue_offsets` (behind an lgkmcnt(0)) turns them into buffer offsets; `issue_piece(k)` is one DMA
    // wave-instruction.  The loop runs the prep beside the MFMAs in front of the stage barrier and spreads the pieces
    // over the MFMAs behind it: a burst of five pieces + their table reads is ~600 cycles in which neither this wave nor
    // - after a barrier: in the same phase - its SIMD partner issues an MFMA.
    i2v tq[5];
    unsigned doff[5];
    int dstep = 0;
    auto issue_prep = [&](int step) {
        const int eb = 64 * step;                       // ring index of dy row r: eb + r + 1; of activation row j: eb + j
#pragma unroll
        for (int i = 0; i < 2; ++i) tq[i] = lds_read8(tab0 + (unsigned)((eb + srow + 32 * i + 1) & (X3_TABN - 1)) * 8u);
#pragma unroll
        for (int i = 0; i < 2; ++i) tq[2 + i] = lds_read8(tab0 + (unsigned)((eb + srow + 32 * i) & (X3_TABN - 1)) * 8u);
        tq[4] = lds_read8(tab0 + (unsigned)((eb + xrow2) & (X3_TABN - 1)) * 8u);
        dstep = step;
    };
    auto issue_offsets = [&]() {
        // every register of the five entries stays allocated until here: hipcc takes an asm output for written when the
        // statement ends, so a field nobody uses would be handed out as a temporary while the LDS read that fills it is
        // still in flight (seen: the address of the next table read computed in tq[0].x's register)
#pragma unroll
        for (int i = 0; i < 5; ++i) asm volatile("" :: "v"(tq[i]));
#pragma unroll
        for (int i = 0; i < 2; ++i) doff[i] = ky_ok ? (unsigned)tq[i].y + (unsigned)ky * 2u : OOB_OFFSET;
#pragma unroll
        for (int i = 0; i < 2; ++i) doff[2 + i] = (unsigned)tq[2 + i].x + (unsigned)cx * 2u;
        doff[4] = xrow2 < RM + 2 ? (unsigned)tq[4].x + (unsigned)cx2 * 2u : OOB_OFFSET;
    };
    auto issue_piece = [&](const int k) {               // k = 0, 1: dy; 2, 3: activation rows 0 .. 63; 4: rows 64 .. 71 (waves 0, 1)
        unsigned char* base = smem + (dstep % X3_NST) * X3_STAGE;
        if (k < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (LDS_PTR(void))(base + (8 * k + wv) * 1024), 16, doff[k], 0, 0, 0);
        else if (k < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (LDS_PTR(void))(base + X3_DY_BYTES + (8 * (k - 2) + wv) * 1024), 16, doff[k], 0, 0, 0);
        else if (wv < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (LDS_PTR(void))(base + X3_DY_BYTES + (16 + wv) * 1024), 16, doff[4], 0, 0, 0);
    };
    auto issue = [&](int step) {                        // the whole stage at once (prologue)
        issue_prep(step);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        issue_offsets();
#pragma unroll
        for (int k = 0; k < 5; ++k) issue_piece(k);
    };

    f4v acc[3][4][2];
    f16v acc32[3][2];                                       // M32: [tap][32-k tile]
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[s3][i][j] = (f4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc32[s3][kt][v] = 0.f;
    }

    const int wk = wv >> 2, wc = wv & 3;
    const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const int f = (q4 | ((g & 1) << 2)) << 1;
    const int row_lo = 8 * g + q4;
    const int half8 = (p4 & 1) * 8;
    unsigned ady[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) ady[t] = row_lo * 256 + ((8 * wk + 2 * t + (p4 >> 1)) ^ f) * 16 + half8;
    // activation fragment addresses [sub][h][shift][t]: pixel i = 32 sub + row_lo + 4 h reads staged row i + s;
    // W | 64 and chunks start at multiples of 64, so the image column of pixel i is i mod W
    constexpr unsigned ZROW = X3_DY_BYTES + (X3_XROWS - 1) * 256;      // staged row 71: always zeros
    unsigned adx[2][2][3][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = sub * 32 + row_lo + 4 * h;
            const int q = i & (P.W - 1);
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const int row = i + s3;
                const bool edge = (s3 == 0 && q == 0) || (s3 == 2 && q == P.W - 1);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int lc = 4 * wc + 2 * t + (p4 >> 1);
                    adx[sub][h][s3][t] = edge ? ZROW + half8
                                              : X3_DY_BYTES + row * 256 + ((lc ^ (x3_f(row) << 1)) << 4) + half8;
                }
            }
        }

    // M32 fragment addresses.  Lane group g: 16-column half kh = g & 1 of the 32-wide operand, pixel half hh = g >> 1 (8 pixels
    // of the 16 of an MFMA); a transposed read covers rows 32 sub + 16 pg + 8 hh + 4 h + q4 (+ tap).  A: base per 32-k tile;
    // B: one base for the middle tap, and for the edge taps one register per (sub, pg, h) - the zero row or base + offset.
    const int hh = g >> 1, kh = g & 1, rowm = 8 * hh + q4;
    unsigned adyM[2], adxM1 = 0, adxE[2][2][2][2];
    if constexpr (M32) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
            adyM[kt] = rowm * 256 + (((8 * wk + 4 * kt + 2 * kh + (p4 >> 1)) ^ (q4 << 2)) << 4) + half8;
        const int cb = 4 * wc + 2 * kh + (p4 >> 1);
        adxM1 = X3_DY_BYTES + (rowm + 1) * 256 + ((cb ^ (((q4 + 1) & 3) << 2)) << 4) + half8;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int pg = 0; pg < 2; ++pg)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int i = 32 * sub + 16 * pg + rowm + 4 * h;
                        const int q = i & (P.W - 1), row = i + 2 * e;
                        const bool edge = e == 0 ? q == 0 : q == P.W - 1;
                        adxE[e][sub][pg][h] = lds0 + (edge ? ZROW + half8
                                                           : X3_DY_BYTES + row * 256 + ((cb ^ ((row & 3) << 2)) << 4) + half8);
                    }
        adyM[0] += lds0;  adyM[1] += lds0;  adxM1 += lds0;    // (stage 0)
    }

    const bool do_colsum = P.colsum != nullptr && c_tile == 0 && kr == 0 && wc == 0;
    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    const int nsteps = (m_end - m_begin + RM - 1) / RM;

    // Fragment registers, double-buffered: while the 24 MFMAs of one 32-pixel group run, the 20 transposed reads of the
    // NEXT group are issued between them (one read behind each of the first 20 MFMAs: a read + its address add fit the 8
    // issue cycles a 16x16x32 MFMA leaves free), so a wave never stands at `lgkmcnt(0)` with an idle matrix pipe and
    // does not depend on its SIMD partner being in the opposite phase.  Raw 8-byte halves until the wait that
    // retires them; the s8v operands are formed after it.
    s4v fa0[2][4], fa1[2][4], fb0[2][3][2], fb1[2][3][2];
    s4v ga[2][2][2][2], gb[2][2][3][2];                     // M32: [buf][pg][32-k tile | tap][h]
    auto tr_off = [&](const unsigned a, const int off) {    // off = 8192 sub + 4096 pg + 1024 h: compile-time at every call
        return off == 0 ? lds_tr<0>(a) : off == 1024 ? lds_tr<1024>(a) : off == 4096 ? lds_tr<4096>(a) : off == 5120 ? lds_tr<5120>(a)
             : off == 8192 ? lds_tr<8192>(a) : off == 9216 ? lds_tr<9216>(a) : off == 12288 ? lds_tr<12288>(a) : lds_tr<13312>(a);
    };
    // M32: the 19 fragment addresses are kept ABSOLUTE (stage base included) and moved on once per stage (`m32_advance`):
    // a read is one LDS instruction, no address add beside the MFMAs
    auto m32_advance = [&](const int to_step) {              // from the buffer of stage to_step - 1 to that of to_step
        const int d = (to_step % X3_NST) == 0 ? -(X3_NST - 1) * X3_STAGE : X3_STAGE;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) adyM[kt] += d;
        adxM1 += d;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int pg = 0; pg < 2; ++pg)
#pragma unroll
                    for (int h = 0; h < 2; ++h) adxE[e][sub][pg][h] += d;
    };
    auto read_one = [&](const int buf, const int r, const unsigned sbn, const int subn) {   // r = 0 .. 19, all compile-time
        if constexpr (M32) {
            if (r < 8) {                                    // (sbn unused: the addresses carry their stage)
                const int pg = r >> 2, kt = (r >> 1) & 1, h = r & 1;
                ga[buf][pg][kt][h] = tr_off(adyM[kt], subn * 8192 + pg * 4096 + h * 1024);
            } else {
                const int rr = r - 8, pg = rr / 6, s3 = (rr % 6) >> 1, h = rr & 1;
                if (s3 == 1) gb[buf][pg][s3][h] = tr_off(adxM1, subn * 8192 + pg * 4096 + h * 1024);
                else         gb[buf][pg][s3][h] = lds_tr<0>(adxE[s3 >> 1][subn][pg][h]);
            }
            return;
        }
        if (r < 8) {
            const int t = r >> 1;
            const unsigned a = sbn + ady[t] + subn * 8192;
            if (r & 1) fa1[buf][t] = lds_tr<1024>(a);
            else       fa0[buf][t] = lds_tr<0>(a);
        } else {
            const int rr = r - 8, s3 = rr >> 2, t = (rr >> 1) & 1, h = rr & 1;
            const unsigned a = sbn + adx[subn][h][s3][t];
            if (h) fb1[buf][s3][t] = lds_tr<0>(a);
            else   fb0[buf][s3][t] = lds_tr<0>(a);
        }
    };
    // the 24 MFMAs of group `buf`, with the reads of group (sbn, subn) into the other buffer between them (rd: issue them)
    auto group = [&](const int buf, auto rd, const unsigned sbn, const int subn, const bool dma) {   // rd: std::true_type / false_type; dma: wave-uniform
        if constexpr (M32) {
            s8v a2[2][2], b2[2][3];
#pragma unroll
            for (int pg = 0; pg < 2; ++pg) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
                    a2[pg][kt] = (s8v){ga[buf][pg][kt][0][0], ga[buf][pg][kt][0][1], ga[buf][pg][kt][0][2], ga[buf][pg][kt][0][3],
                                       ga[buf][pg][kt][1][0], ga[buf][pg][kt][1][1], ga[buf][pg][kt][1][2], ga[buf][pg][kt][1][3]};
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3)
                    b2[pg][s3] = (s8v){gb[buf][pg][s3][0][0], gb[buf][pg][s3][0][1], gb[buf][pg][s3][0][2], gb[buf][pg][s3][0][3],
                                       gb[buf][pg][s3][1][0], gb[buf][pg][s3][1][1], gb[buf][pg][s3][1][2], gb[buf][pg][s3][1][3]};
            }
            if (do_colsum) {
#pragma unroll
                for (int pg = 0; pg < 2; ++pg)
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int e = 0; e < 8; ++e) csum[kt] += bf16_to_f32((unsigned short)a2[pg][kt][e]);
            }
            // 12 MFMAs; the 20 reads of the next group two by two behind the first ten, the five DMA pieces behind 1, 3, .. 9
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                const int pg = m / 6, s3 = (m % 6) >> 1, kt = m & 1;
                if constexpr (!(ABL & 8)) acc32[s3][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[pg][kt], b2[pg][s3], acc32[s3][kt], 0, 0, 0);
                else { asm volatile("" :: "v"(a2[pg][kt]), "v"(b2[pg][s3])); }
                if (m < 10) {
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (decltype(rd)::value && !(ABL & 2)) {
                        read_one(buf ^ 1, 2 * m, sbn, subn);
                        read_one(buf ^ 1, 2 * m + 1, sbn, subn);
                    }
                    if ((m & 1) == 1 && dma) issue_piece(m >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            return;
        }
        s8v af[4], bf[3][2];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            af[t] = (s8v){fa0[buf][t][0], fa0[buf][t][1], fa0[buf][t][2], fa0[buf][t][3],
                          fa1[buf][t][0], fa1[buf][t][1], fa1[buf][t][2], fa1[buf][t][3]};
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                bf[s3][t] = (s8v){fb0[buf][s3][t][0], fb0[buf][s3][t][1], fb0[buf][s3][t][2], fb0[buf][s3][t][3],
                                  fb1[buf][s3][t][0], fb1[buf][s3][t][1], fb1[buf][s3][t][2], fb1[buf][s3][t][3]};
        if (do_colsum) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) csum[t] += bf16_to_f32((unsigned short)af[t][e]);
        }
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            const int s3 = m >> 3, i = (m >> 1) & 3, j = m & 1;
            if constexpr (!(ABL & 8)) acc[s3][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[s3][j], acc[s3][i][j], 0, 0, 0);
            else { asm volatile("" :: "v"(af[i]), "v"(bf[s3][j])); }
            if (m < 20) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (decltype(rd)::value && !(ABL & 2)) read_one(buf ^ 1, m, sbn, subn);
                if ((m & 3) == 1 && dma) issue_piece(m >> 2);       // five pieces behind MFMAs 1, 5, 9, 13, 17
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    fill(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // prologue: the first X3_NST stages; stage 0 landed (the others may stay in flight: 4 pieces per wave and stage, 5 for
    // waves 0, 1)
#pragma unroll
    for (int i = 0; i < X3_NST; ++i)
        if (nsteps > i) issue(i);
    {
        const int fly = min(nsteps, X3_NST) - 1;                    // stages that may stay in flight
        auto wait_stages = [&](const int n) {                       // n wave-uniform
            if (n <= 0)      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (n == 1) { if (wv < 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else if (n == 2) { if (wv < 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            else             { if (wv < 2) asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
        };
        wait_stages(fly);
    }
    __builtin_amdgcn_s_barrier();
    if (nsteps > 0) {
#pragma unroll
        for (int r = 0; r < 20; ++r) read_one(0, r, lds0, 0);
    }
    for (int step = 0; step < nsteps; ++step) {
        const unsigned sb = lds0 + (unsigned)(step % X3_NST) * X3_STAGE;
        const unsigned sbn = lds0 + (unsigned)((step + 1) % X3_NST) * X3_STAGE;
        lds_wait();                                                 // group (step, 0) is in buffer 0
        const bool dma = !(ABL & 1) && step + X3_NST < nsteps;
        group(0, std::true_type{}, sb, 1, false);                   // ... and (step, 1) on its way into buffer 1
        if constexpr (M32) m32_advance(step + 1);                   // the next reads are those of (step + 1, 0)
        if (dma) issue_prep(step + X3_NST);                         // pixel rows of the stage to issue behind the barrier
        lds_wait();                                                 // this wave has read stage `step` out
        if (dma) issue_offsets();
        // stage step + 1 landed (stages step + 2 .. step + X3_NST - 1 may stay in flight)
        {
            const int fly = min(nsteps - step - 2, X3_NST - 2);
            if (fly <= 0)      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (fly == 1) { if (wv < 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else               { if (wv < 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        }
        __builtin_amdgcn_s_barrier();                               // every wave: stage step + 1 landed, stage step read out
        // (the pieces of stage step + X3_NST go out between the MFMAs of the next group, into the buffer of stage step)
        // ring entries of stages step + 6 .. step + 13: published by the NEXT iteration's barrier, first read by the
        // issue_prep of the iteration after that; the slots they replace belong to stages <= step - 2, whose pixel rows
        // were read five or more iterations ago
        if (((step + 6) & 7) == 0 && step + 6 < nsteps) fill(step + 6);
        // (step + 1, 0) on its way into buffer 0; behind the last stage the reads fetch stale LDS bytes into fragments
        // nobody uses (always inside the allocation): no branch in the MFMA stream
        group(1, std::true_type{}, sbn, 0, dma);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if (do_colsum && M32) {                                 // lane & 31 = k inside the 32-k tile, lane >> 5 = pixel half
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            float v = csum[kt];
            v += __shfl_xor(v, 32, 64);
            const int k = k0 + wk * 64 + kt * 32 + (lane & 31);
            if (lane < 32 && k < L.K) {
                if (L.slab_stride > 0) P.colsum[(long long)lchunk_id * L.K + k] = v;
                else atomicAdd(P.colsum + k, v);
            }
        }
    }
    if (do_colsum && !M32) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = csum[t];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int k = k0 + wk * 64 + t * 16 + (lane & 15);
            if (lane < 16 && k < L.K) {
                if (L.slab_stride > 0) P.colsum[(long long)lchunk_id * L.K + k] = v;
                else atomicAdd(P.colsum + k, v);
            }
        }
    }
    // ---- epilogue, one column tap at a time through LDS: 512-B contiguous row segments
    float* epi = reinterpret_cast<float*>(smem);
    const int fr = lane & 15, fq = lane >> 4;
    const long long wrow = (long long)L.T * P.C;
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        if constexpr (M32) {                                // D: column = lane & 31, row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    epi[(wk * 64 + kt * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5)) * X3_EPI_LD + wc * 32 + (lane & 31)] = acc32[s3][kt][v];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        epi[(wk * 64 + i * 16 + fq * 4 + r) * X3_EPI_LD + wc * 32 + j * 16 + fr] = acc[s3][i][j][r];
        }
        __syncthreads();
        const long long coff = (long long)((P.r0 + kr * P.rs) * P.S + (P.s0 + s3 * P.ss)) * P.C + c0 + lane;
        if (L.wide) {
            // 1 KiB per store instruction (two 512-B row segments): the tail of a block is store-ISSUE bound, and every
            // block of the launch reaches it at about the same time
            const int l32 = lane & 31;
            for (int rr = 2 * wv + (lane >> 5); rr < TK; rr += 16) {
                const int k = k0 + rr;
                if (k >= L.K) continue;
                const float4 v = *reinterpret_cast<const float4*>(epi + rr * X3_EPI_LD + 4 * l32);
                if (ABL & 4) { asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); continue; }
                slab_store4(P.dw + k * wrow + (coff - lane) + 4 * l32 + (long long)lchunk_id * L.slab_stride, v, L.wide == 2);
            }
        } else {
            for (int rr = wv; rr < TK; rr += 8) {
                const int k = k0 + rr;
                if (k >= L.K) break;
                float* dst = P.dw + k * wrow + coff;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float v = epi[rr * X3_EPI_LD + h * 64 + lane];
                    if (ABL & 4) { asm volatile("" :: "v"(v)); continue; }
                    if (L.slab_stride > 0) dst[(long long)lchunk_id * L.slab_stride + h * 64] = v;
                    else atomicAdd(dst + h * 64, v);
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

// MI355_WG_K64=0: K <= 64 layers through the 128-row k-tile forms (A/B)
static int wg_k64_on() {
    static const int on = [] { const char* e = getenv("MI355_WG_K64"); return e ? atoi(e) : 1; }();
    return on;
}

template <bool ALIGNED>
__global__ void slab_reduce_kernel(const float4* __restrict__ ws, int nslabs, long long slab4, long long n4,
                                   float* __restrict__ out, int accumulate) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (accumulate) {
        if (ALIGNED) a = reinterpret_cast<const float4*>(out)[i];
        else a = make_float4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
    }
    for (int s = 0; s < nslabs; ++s) {                       // fixed order: bit-reproducible
        const float4 v = ws[(long long)s * slab4 + i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (ALIGNED) reinterpret_cast<float4*>(out)[i] = a;
    else { out[4 * i] = a.x; out[4 * i + 1] = a.y; out[4 * i + 2] = a.z; out[4 * i + 3] = a.w; }
}

extern "C" int mi355_slab_reduce(const float* ws, int nslabs, int64_t slab_stride, int64_t n, float* out,
                                 int accumulate, void* stream) {
    MI355_REQUIRE(ws && out && nslabs > 0 && n > 0 && n % 4 == 0 && slab_stride % 4 == 0, "slab_reduce: bad args");
    MI355_REQUIRE(((uintptr_t)ws & 15) == 0, "slab_reduce: workspace alignment");
    const dim3 g((unsigned)cdiv(n / 4, 256)), b(256);
    if (((uintptr_t)out & 15) == 0)
        hipLaunchKernelGGL(slab_reduce_kernel<true>, g, b, 0, (hipStream_t)stream, (const float4*)ws, nslabs,
                           (long long)(slab_stride / 4), (long long)(n / 4), out, accumulate);
    else
        hipLaunchKernelGGL(slab_reduce_kernel<false>, g, b, 0, (hipStream_t)stream, (const float4*)ws, nslabs,
                           (long long)(slab_stride / 4), (long long)(n / 4), out, accumulate);
    return check_launch("slab_reduce_kernel");
}

extern "C" int mi355_conv_wgrad(const Mi355ConvLaunch* launch, float* dw, int split_rows, void* stream) {
    return mi355_conv_wgrad_colsum(launch, dw, nullptr, split_rows, stream);
}

extern "C" int mi355_conv_wgrad_colsum(const Mi355ConvLaunch* launch, float* dw, float* colsum,
                                       int split_rows, void* stream) {
    return mi355_conv_wgrad_slabs(launch, dw, 0, 0, colsum, split_rows, stream);
}

extern "C" int mi355_conv_wgrad_num_slabs(const Mi355ConvLaunch* launch, int split_rows, int* nslabs) {
    MI355_REQUIRE(launch && nslabs && split_rows > 0, "conv_wgrad_num_slabs: bad args");
    int chunks = 0;
    for (int i = 0; i < launch->nprob; ++i) {
        const Mi355ConvProblem& s = launch->prob[i];
        chunks += (int)cdiv((long long)s.N * s.P * s.Q, split_rows);
    }
    *nslabs = chunks;
    return MI355_OK;
}

static int wgrad_launch(const Mi355ConvLaunch& in, float* const* dws, int64_t slab_stride, const int* max_slabs,
                        float* const* colsums, bool multi, int split_rows, void* stream);

extern "C" int mi355_conv_wgrad_slabs(const Mi355ConvLaunch* launch, float* dw, int64_t slab_stride,
                                      int max_slabs, float* colsum, int split_rows, void* stream) {
    MI355_REQUIRE(launch && dw, "conv_wgrad: null argument");
    return wgrad_launch(*launch, &dw, slab_stride, &max_slabs, &colsum, false, split_rows, stream);
}

extern "C" int mi355_conv_wgrad_multi(const Mi355ConvLaunch* launch, float* const* ws, int64_t slab_stride,
                                      const int* max_slabs, float* const* colsum, int split_rows, void* stream) {
    MI355_REQUIRE(launch && ws && max_slabs && colsum && slab_stride > 0, "conv_wgrad_multi: null argument / slab mode only");
    for (int i = 0; i < launch->nprob; ++i) MI355_REQUIRE(ws[i], "conv_wgrad_multi: problem %d has no workspace", i);
    return wgrad_launch(*launch, ws, slab_stride, max_slabs, colsum, true, split_rows, stream);
}

static int wgrad_launch(const Mi355ConvLaunch& in, float* const* dws, int64_t slab_stride, const int* max_slabs,
                        float* const* colsums, bool multi, int split_rows, void* stream) {
    MI355_REQUIRE(in.nprob >= 1 && in.nprob <= MI355_CONV_MAX_PROBLEMS, "conv_wgrad: nprob");
    MI355_REQUIRE(in.K > 0 && in.K % 8 == 0 && in.T > 0, "conv_wgrad: K must be a positive multiple of 8");
    MI355_REQUIRE(split_rows > 0 && split_rows % RM == 0, "conv_wgrad: split_rows must be a multiple of 64");
    // MI355_WG_GROUPS=2: two wave groups per block when the chunk is a multiple of 128 rows (half the
    // slabs; measured slower end to end than two independent four-wave blocks per CU, so off by default)
    static const int max_groups = [] {
        const char* e = getenv("MI355_WG_GROUPS");
        return e ? atoi(e) : 1;
    }();
    const bool two = max_groups >= 2 && split_rows % (2 * RM) == 0;
    WLaunch L;
    memset(&L, 0, sizeof(L));
    const int C = in.prob[0].C;
    const int ntaps = in.prob[0].TR * in.prob[0].TS;
    int chunks = 0;
    for (int i = 0; i < in.nprob; ++i) {
        const Mi355ConvProblem& s = in.prob[i];
        MI355_REQUIRE(s.in && s.out, "conv_wgrad: problem %d null in/dy", i);
        MI355_REQUIRE(s.C == C && C % 8 == 0 && s.TR * s.TS == ntaps, "conv_wgrad: C%%8, taps equal across problems");
        const long long in_sN = s.in_sN ? s.in_sN : (long long)s.H * s.W * s.C;
        const long long dy_sN = s.out_sN ? s.out_sN : (long long)s.P * s.Q * in.K;
        const long long in_bytes = ((long long)(s.N - 1) * in_sN + (long long)s.H * s.W * s.C) * 2;
        const long long M = (long long)s.N * s.P * s.Q;
        const long long dy_bytes = ((long long)(s.N - 1) * dy_sN + (long long)s.P * s.Q * in.K) * 2;
        MI355_REQUIRE(in_bytes < 0x7fffffffLL && dy_bytes < 0x7fffffffLL && M > 0, "conv_wgrad: tensor >= 2 GiB");
        MI355_REQUIRE(((uintptr_t)s.in & 15) == 0 && ((uintptr_t)s.out & 15) == 0, "conv_wgrad: alignment");
        MI355_REQUIRE(s.H < 32000 && s.W < 32000, "conv_wgrad: H, W must be < 32000");
        WProb& d = L.p[i];
        d.in = (const unsigned short*)s.in;  d.dy = (const unsigned short*)s.out;
        d.N = s.N; d.H = s.H; d.W = s.W; d.C = s.C; d.P = s.P; d.Q = s.Q; d.sh = s.sh; d.sw = s.sw;
        d.TR = s.TR; d.TS = s.TS; d.dh0 = s.dh0; d.dhs = s.dhs; d.dw0 = s.dw0; d.dws = s.dws;
        d.r0 = s.r0; d.rs = s.rs; d.s0 = s.s0; d.ss = s.ss; d.S = s.S;
        d.M = (int)M; d.in_bytes = (int)in_bytes; d.dy_bytes = (int)dy_bytes;
        d.in_sN = (int)in_sN; d.dy_sN = (int)dy_sN;
        d.dPQ = make_fastdiv((unsigned)(s.P * s.Q));
        d.dQ = make_fastdiv((unsigned)s.Q);
        d.chunk_begin = chunks;
        const int mine = (int)cdiv(M, split_rows);
        if (multi) {                                   // one workspace per problem (= per layer), slabs 0 .. mine-1
            MI355_REQUIRE(mine <= max_slabs[i], "conv_wgrad_multi: problem %d: %d slabs > workspace for %d", i, mine, max_slabs[i]);
            d.dw = dws[i];
            d.colsum = colsums[i];
        } else {                                       // shared weight: one workspace, this problem's slabs start at `chunks`
            d.dw = dws[0] + (slab_stride > 0 ? (long long)chunks * slab_stride : 0);
            d.colsum = colsums[0] ? colsums[0] + (slab_stride > 0 ? (long long)chunks * in.K : 0) : nullptr;
        }
        chunks += mine;
    }
    if (ntaps == 0) return MI355_OK;
    // row-of-taps kernel: 3x3 / stride 1 / same, C % 64 == 0 (widths that do not divide 64: its table-driven edge form)
    static const int row3_on = [] {
        const char* e = getenv("MI355_WG_ROW3");
        return e ? atoi(e) : 1;
    }();
    static const int row3_dyn_on = [] {
        const char* e = getenv("MI355_WG_ROW3_DYN");
        return e ? atoi(e) : 1;
    }();
    bool row3 = row3_on && C % 64 == 0 && ntaps == 9, row3_dyn = false;   // (dyn: a width that does not divide 64)
    for (int i = 0; i < in.nprob && row3; ++i) {
        const Mi355ConvProblem& q = in.prob[i];
        row3 = q.TR == 3 && q.TS == 3 && q.sh == 1 && q.sw == 1 && q.dhs == 1 && q.dws == 1 && q.dw0 == -1 &&
               q.Q == q.W;
        const bool pow2 = (q.W & (q.W - 1)) == 0;
        if (q.W > 64 || !pow2) {
            // per-launch A/B at 640 x 640 / 512 x 512 (tools/profile_plan.py): ahead of the generic kernel for K >= 128
            // and C <= 256 (head towers 193 -> 188 us, layer3 group 224 -> 172 us, layer2 group 218 -> 192 us) and at
            // width 128 (layer1 group 153 -> 138 us); behind it for K = 64 at width 160 (also in the K <= 64 forms: 269 vs
            // 226 us) and for C = 512
            row3_dyn = true;
            row3 = row3 && row3_dyn_on && (pow2 || (in.K >= 128 && C <= 256));
        }
    }
    if (in.nprob > 1 && chunks <= MAX_SORTED) {
        int n = 0, rows[MAX_SORTED];
        for (int i = 0; i < in.nprob; ++i)
            for (int m = 0; m < L.p[i].M; m += split_rows) {
                rows[n] = L.p[i].M - m < split_rows ? L.p[i].M - m : split_rows;
                L.ch_prob[n] = (unsigned char)i;
                L.ch_mbegin[n] = m;
                ++n;
            }
        for (int a = 1; a < n; ++a)                      // stable insertion sort, longest first
            for (int b = a; b > 0 && rows[b] > rows[b - 1]; --b) {
                const int r = rows[b]; rows[b] = rows[b - 1]; rows[b - 1] = r;
                const unsigned char pp = L.ch_prob[b]; L.ch_prob[b] = L.ch_prob[b - 1]; L.ch_prob[b - 1] = pp;
                const int mb = L.ch_mbegin[b]; L.ch_mbegin[b] = L.ch_mbegin[b - 1]; L.ch_mbegin[b - 1] = mb;
            }
        L.nsorted = n;
    }
    L.slab_stride = slab_stride;
    MI355_REQUIRE(multi || slab_stride == 0 || chunks <= max_slabs[0], "conv_wgrad_slabs: %d slabs > workspace for %d", chunks, max_slabs[0]);
    L.nprob = in.nprob; L.K = in.K; L.T = in.T; L.ntaps = ntaps; L.split_rows = split_rows;
    const bool flat = C < TC && !row3;
    L.k_tiles = (int)cdiv(in.K, TK);
    L.c_tiles = flat ? (int)cdiv((long long)ntaps * C, TC) : (int)cdiv(C, TC);
    if (flat) L.ntaps = 1;
    // eight-wave ring form (conv_wgrad_row3x_kernel): widths dividing the stage, C % 128 == 0, K >= 128 (MI355_WG_ROW3X=0: off)
    static const int row3x_on = [] {
        const char* e = getenv("MI355_WG_ROW3X");
        return e ? atoi(e) : 1;
    }();
    const bool row3x = row3 && !row3_dyn && row3x_on && C % X3_TC == 0 && in.K >= 128;
    if (row3) {
        L.c_tiles = C / (row3x ? X3_TC : R3_TC);
        L.ntaps = 3;                                   // kernel rows; a block accumulates the three column taps
    }
    const long long grid = (long long)chunks * L.ntaps * L.k_tiles * L.c_tiles;
    MI355_REQUIRE(grid > 0 && grid < 0x7fffffffLL, "conv_wgrad: grid");
    static const int unit_mode = [] {
        const char* e = getenv("MI355_WG_UNITS");
        return e ? atoi(e) : 1;
    }();
    L.ub = 1;
    if (unit_mode && in.nprob == 1 && !row3) {                  // equal chunks (but the last): balanced iff #units % 8 == 0
        const int tpc = L.ntaps * L.k_tiles * L.c_tiles;
        const int cand[2] = {tpc, L.k_tiles * L.c_tiles};      // whole chunk, or one tap's tiles of a chunk
        for (int i = 0; i < 2; ++i)
            if (cand[i] > 1 && grid % cand[i] == 0 && (grid / cand[i]) % 8 == 0) { L.ub = cand[i]; break; }
    }
    static const int xcd_mode = [] {
        const char* e = getenv("MI355_WG_XCD");
        return e ? atoi(e) : 1;
    }();
    L.xs = !xcd_mode ? 0 : (row3x ? 256 : 512);            // blocks resident at once: one / two per CU
    static const int wide_mode = [] {                       // MI355_WG_WIDE=0: 4-B epilogue stores (A/B)
        const char* e = getenv("MI355_WG_WIDE");
        return e ? atoi(e) : 1;
    }();
    L.wide = wide_mode && slab_stride > 0 && slab_stride % 4 == 0 && C % 4 == 0;
    for (int i = 0; i < in.nprob && L.wide; ++i) L.wide = ((uintptr_t)L.p[i].dw & 15) == 0;
    static const int nt_mode = [] {                         // MI355_WG_NT=0: plain slab stores (A/B: +0.7 % on the R50 step with nt)
        const char* e = getenv("MI355_WG_NT");
        return e ? atoi(e) : 1;
    }();
    if (L.wide && nt_mode) L.wide = 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(1));
        (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(1));
        (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<true, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(1));
        (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(1));
        (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(2));
        (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(2));
        attr_set = true;
    }
    const dim3 g((unsigned)grid);
    hipStream_t st = (hipStream_t)stream;
    if (row3x) {
        static bool x3_attr = false;
        if (!x3_attr) {
            (void)hipFuncSetAttribute((const void*)conv_wgrad_row3x_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_row3x_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_row3x_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_row3x_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_row3x_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_row3x_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_row3x_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_row3x_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_row3x_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_row3x_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_row3x_kernel<11>, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS);
            x3_attr = true;
        }
        static const int m32_on = [] {                              // MI355_X3_M32=0: 16x16x32 MFMAs (A/B)
            const char* e = getenv("MI355_X3_M32");
            return e ? atoi(e) : 1;
        }();
        const char* ae = getenv("MI355_X3_ABLATE");               // diagnostic builds (not cached: a bench tool sweeps it)
        if (m32_on && (ae ? atoi(ae) : 0) == 0) hipLaunchKernelGGL((conv_wgrad_row3x_kernel<0, true>), g, dim3(512), X3_LDS, st, L);
        else switch (ae ? atoi(ae) : 0) {
            case 1: hipLaunchKernelGGL(conv_wgrad_row3x_kernel<1>, g, dim3(512), X3_LDS, st, L); break;
            case 2: hipLaunchKernelGGL(conv_wgrad_row3x_kernel<2>, g, dim3(512), X3_LDS, st, L); break;
            case 3: hipLaunchKernelGGL(conv_wgrad_row3x_kernel<3>, g, dim3(512), X3_LDS, st, L); break;
            case 4: hipLaunchKernelGGL(conv_wgrad_row3x_kernel<4>, g, dim3(512), X3_LDS, st, L); break;
            case 7: hipLaunchKernelGGL(conv_wgrad_row3x_kernel<7>, g, dim3(512), X3_LDS, st, L); break;
            case 8: hipLaunchKernelGGL(conv_wgrad_row3x_kernel<8>, g, dim3(512), X3_LDS, st, L); break;
            case 9: hipLaunchKernelGGL(conv_wgrad_row3x_kernel<9>, g, dim3(512), X3_LDS, st, L); break;
            case 10: hipLaunchKernelGGL(conv_wgrad_row3x_kernel<10>, g, dim3(512), X3_LDS, st, L); break;
            case 11: hipLaunchKernelGGL(conv_wgrad_row3x_kernel<11>, g, dim3(512), X3_LDS, st, L); break;
            default: hipLaunchKernelGGL(conv_wgrad_row3x_kernel<0>, g, dim3(512), X3_LDS, st, L);
        }
    } else if (row3) {
        static bool r3_attr = false;
        if (!r3_attr) {
            (void)hipFuncSetAttribute((const void*)conv_wgrad_row3_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, R3_LDS);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_row3_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, R3_LDS);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_row3_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, R3_LDS);
            (void)hipFuncSetAttribute((const void*)conv_wgrad_row3_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, R3_LDS);
            r3_attr = true;
        }
        const bool kh = wg_k64_on() && in.K <= 64;
        if (row3_dyn) {
            if (kh) hipLaunchKernelGGL((conv_wgrad_row3_kernel<true, true>), g, dim3(256), R3_LDS, st, L);
            else    hipLaunchKernelGGL((conv_wgrad_row3_kernel<true, false>), g, dim3(256), R3_LDS, st, L);
        } else {
            if (kh) hipLaunchKernelGGL((conv_wgrad_row3_kernel<false, true>), g, dim3(256), R3_LDS, st, L);
            else    hipLaunchKernelGGL((conv_wgrad_row3_kernel<false, false>), g, dim3(256), R3_LDS, st, L);
        }
    } else if (two) {
        if (flat) hipLaunchKernelGGL((conv_wgrad_kernel<true, 2>), g, dim3(512), lds_bytes(2), st, L);
        else      hipLaunchKernelGGL((conv_wgrad_kernel<false, 2>), g, dim3(512), lds_bytes(2), st, L);
    } else if (wg_k64_on() && in.K <= 64) {
        if (flat) hipLaunchKernelGGL((conv_wgrad_kernel<true, 1, true>), g, dim3(256), lds_bytes(1), st, L);
        else      hipLaunchKernelGGL((conv_wgrad_kernel<false, 1, true>), g, dim3(256), lds_bytes(1), st, L);
    } else {
        if (flat) hipLaunchKernelGGL((conv_wgrad_kernel<true, 1>), g, dim3(256), lds_bytes(1), st, L);
        else      hipLaunchKernelGGL((conv_wgrad_kernel<false, 1>), g, dim3(256), lds_bytes(1), st, L);
    }
    return check_launch("conv_wgrad_kernel");
}
