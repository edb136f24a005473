// Input side of the path (include/mi355det_input.h): decoded images -> the [B,3,H,W] fp32 batch the
// models take, and the -1 padded annotation block, one launch each.  HBM-bound: 3 (uint8) or 12 (fp32)
// bytes read and 12 bytes written per pixel.
//
// Arithmetic follows the reference bit for bit (datasettool/dataloader.py:92, :905-924): the pixel is
// scaled in float32, normalised in float64 against the float64 mean / std and rounded to float32 once.
// For uint8 input only 3 x 256 distinct results exist: every block builds that table in LDS with the
// same operations (three float64 divisions per thread) and the pixel loop is a lookup.
#include "common.h"
#include "mi355det_input.h"

namespace {

constexpr int CB = 256;          // threads per block
constexpr int PIX_PER_BLOCK = 4096;

struct NormArgs { double mean[3], std[3]; };

__device__ inline float convert(float v, int c, int flags, const NormArgs& na) {
    if (flags & MI355_IN_SCALE255) v = v / 255.0f;
    if (flags & MI355_IN_NORMALIZE) v = (float)(((double)v - na.mean[c]) / na.std[c]);
    return v;
}

template <bool VEC4>
__global__ __launch_bounds__(CB) void collate_images_kernel(const uint8_t* __restrict__ packed,
                                                            const int64_t* __restrict__ meta, int H, int W,
                                                            int flags, NormArgs na, float* __restrict__ out) {
    __shared__ float lut[3][256];
    const int b = blockIdx.y;
    const bool u8 = flags & MI355_IN_U8;
    if (u8) {
#pragma unroll
        for (int c = 0; c < 3; ++c) lut[c][threadIdx.x] = convert((float)threadIdx.x, c, flags, na);
        __syncthreads();
    }
    const int64_t off = meta[3 * b];
    const int h = (int)meta[3 * b + 1], w = (int)meta[3 * b + 2];
    const int64_t plane = (int64_t)H * W;
    float* ob = out + (int64_t)b * 3 * plane;
    const int64_t first = (int64_t)blockIdx.x * PIX_PER_BLOCK;
    auto pixel = [&](int y, int x, float& r0, float& r1, float& r2) {
        r0 = r1 = r2 = 0.0f;
        if (y < h && x < w) {
            const int64_t e = ((int64_t)y * w + x) * 3;
            if (u8) {
                const uint8_t* s = packed + off + e;
                r0 = lut[0][s[0]]; r1 = lut[1][s[1]]; r2 = lut[2][s[2]];
            } else {
                const float* s = reinterpret_cast<const float*>(packed + off) + e;
                r0 = convert(s[0], 0, flags, na); r1 = convert(s[1], 1, flags, na); r2 = convert(s[2], 2, flags, na);
            }
        }
    };
    if (VEC4) {                                              // W % 4 == 0: four pixels of a row per thread, 16-byte stores
        for (int i = threadIdx.x * 4; i < PIX_PER_BLOCK; i += CB * 4) {
            const int64_t p = first + i;
            if (p >= plane) break;
            const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
            float4 v0, v1, v2;
            pixel(y, x, v0.x, v1.x, v2.x);
            pixel(y, x + 1, v0.y, v1.y, v2.y);
            pixel(y, x + 2, v0.z, v1.z, v2.z);
            pixel(y, x + 3, v0.w, v1.w, v2.w);
            *reinterpret_cast<float4*>(ob + p) = v0;
            *reinterpret_cast<float4*>(ob + plane + p) = v1;
            *reinterpret_cast<float4*>(ob + 2 * plane + p) = v2;
        }
        return;
    }
    for (int i = threadIdx.x; i < PIX_PER_BLOCK; i += CB) {
        const int64_t p = first + i;
        if (p >= plane) break;
        const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
        float r0, r1, r2;
        pixel(y, x, r0, r1, r2);
        ob[p] = r0; ob[plane + p] = r1; ob[2 * plane + p] = r2;
    }
}

__global__ __launch_bounds__(CB) void collate_annots_kernel(const double* __restrict__ packed,
                                                            const int64_t* __restrict__ meta,
                                                            const double* __restrict__ scale, int B, int M,
                                                            float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * CB + threadIdx.x;
    if (i >= (int64_t)B * M * 5) return;
    const int col = (int)(i % 5);
    const int64_t row = i / 5;
    const int b = (int)(row / M), r = (int)(row - (int64_t)b * M);
    float v = -1.0f;
    if (r < meta[2 * b + 1]) {
        double a = packed[(meta[2 * b] + r) * 5 + col];
        if (scale && col < 4) a *= scale[b];
        v = (float)a;
    }
    out[i] = v;
}

}  // namespace

extern "C" int mi355_collate_images(const void* packed, const int64_t* meta, int B, int H, int W, int flags,
                                    const double* mean3, const double* std3, float* out, void* stream) {
    MI355_REQUIRE(B >= 0 && H > 0 && W > 0, "collate_images: bad shape B=%d H=%d W=%d", B, H, W);
    if (B == 0) return MI355_OK;
    MI355_REQUIRE(packed && meta && out, "collate_images: null pointer");
    MI355_REQUIRE((flags & ~7) == 0, "collate_images: unknown flags %d", flags);
    MI355_REQUIRE(B <= 65535, "collate_images: B=%d > 65535", B);
    NormArgs na{{0, 0, 0}, {1, 1, 1}};
    if (flags & MI355_IN_NORMALIZE) {
        MI355_REQUIRE(mean3 && std3, "collate_images: MI355_IN_NORMALIZE needs mean and std");
        for (int c = 0; c < 3; ++c) {
            MI355_REQUIRE(std3[c] != 0.0, "collate_images: std[%d] = 0", c);
            na.mean[c] = mean3[c]; na.std[c] = std3[c];
        }
    }
    const int64_t plane = (int64_t)H * W;
    dim3 grid((unsigned)mi355::cdiv(plane, PIX_PER_BLOCK), (unsigned)B);
    if (W % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
        hipLaunchKernelGGL(collate_images_kernel<true>, grid, dim3(CB), 0, (hipStream_t)stream,
                           (const uint8_t*)packed, meta, H, W, flags, na, out);
    else
        hipLaunchKernelGGL(collate_images_kernel<false>, grid, dim3(CB), 0, (hipStream_t)stream,
                           (const uint8_t*)packed, meta, H, W, flags, na, out);
    return mi355::check_launch("collate_images");
}

extern "C" int mi355_collate_annots(const double* packed, const int64_t* meta, const double* scale, int B, int M,
                                    float* out, void* stream) {
    MI355_REQUIRE(B >= 0 && M >= 1, "collate_annots: bad shape B=%d M=%d", B, M);
    if (B == 0) return MI355_OK;
    MI355_REQUIRE(meta && out, "collate_annots: null pointer");
    const int64_t n = (int64_t)B * M * 5;
    hipLaunchKernelGGL(collate_annots_kernel, dim3((unsigned)mi355::cdiv(n, CB)), dim3(CB), 0, (hipStream_t)stream,
                       packed, meta, scale, B, M, out);
    return mi355::check_launch("collate_annots");
}
