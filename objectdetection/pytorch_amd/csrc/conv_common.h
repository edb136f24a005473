// Device-side helpers shared by the MFMA conv kernels (gfx950).
#pragma once
#include "common.h"
#include "mi355det_conv.h"

typedef short s8v __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
#define LDS_PTR(T) __attribute__((address_space(3))) T*

namespace mi355 {

__device__ __forceinline__ float bf16_to_f32(unsigned short v) {
    return __uint_as_float(((unsigned int)v) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_rn(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}

// Bijective XCD-aware remap (cdna_hip_programming.md T1): blocks that share an XCD (bid % 8)
// get a contiguous range of tile ids, so that neighbouring tiles reuse the XCD's L2.
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
}

// Buffer resource over [base, base+bytes): out-of-range 16-B requests return zeros, which is
// how conv padding / tile tails are produced (probed on hardware: tools/probe_gfx950.hip [3]).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* p = (void*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

constexpr unsigned OOB_OFFSET = 0x80000000u;

// Division by a launch constant: floor(x / d) = umulhi(x, mul) >> shr for x < 2^31.
struct FastDiv {
    unsigned mul, shr, d;
};
inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    f.d = d;
    if (d <= 1) { f.mul = 0; f.shr = 0; f.d = 1; return f; }
    unsigned l = 0;
    while ((1u << l) < d) ++l;
    const unsigned p = 31 + l;
    f.mul = (unsigned)(((1ull << p) + d - 1) / d);
    f.shr = p - 32;
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned x, const FastDiv f) {
    return f.d == 1 ? x : (__umulhi(x, f.mul) >> f.shr);
}

// Flat cell index t -> (c8, x, y, n) for [n][y][x][c8] with extents (c8n, X, Y): 32-bit divisions whenever the index
// fits (always, for the tensors of a training step) - three 64-bit div / mod pairs per thread cost the elementwise
// kernels more instructions than their loads and arithmetic together.
__device__ __forceinline__ void split_cell(long long t, int c8n, int X, int Y, int& c8, int& x, int& y, int& n) {
    if ((unsigned long long)t >> 32) {
        c8 = (int)(t % c8n);
        long long r = t / c8n;
        x = (int)(r % X);  r /= X;
        y = (int)(r % Y);
        n = (int)(r / Y);
    } else {
        const unsigned u = (unsigned)t;
        const unsigned r = u / (unsigned)c8n;
        c8 = (int)(u - r * (unsigned)c8n);
        const unsigned r2 = r / (unsigned)X;
        x = (int)(r - r2 * (unsigned)X);
        const unsigned r3 = r2 / (unsigned)Y;
        y = (int)(r2 - r3 * (unsigned)Y);
        n = (int)r3;
    }
}

}  // namespace mi355
