// Where does the time of the max-pool backward (16 x 256 x 256 x 64 <- 16 x 128 x 128 x 64) go?  Its store pattern alone, a
// contiguous store of the same bytes, its load pattern alone, and the kernel itself (through libmi355det.so).
// hipcc --offload-arch=gfx950 -O3 -I include tools/probe_pool.hip -L objectdetection/pytorch_amd -lmi355det -o /tmp/probe_pool
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "mi355det_conv.h"
typedef short s8v __attribute__((ext_vector_type(8)));

__global__ void store_2x2(unsigned short* gin, int N, int H, int W, int C) {
    const int c8n = C >> 3, H2 = H >> 1, W2 = W >> 1;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (unsigned)N * H2 * W2 * c8n) return;
    const unsigned r = t / c8n; const int c8 = t - r * c8n;
    const unsigned r2 = r / W2; const int q2 = r - r2 * W2;
    const int n = r2 / H2, p2 = r2 - n * H2;
    s8v v = (s8v){1, 2, 3, 4, 5, 6, 7, (short)t};
    for (int dh = 0; dh < 2; ++dh)
        for (int dw = 0; dw < 2; ++dw)
            *reinterpret_cast<s8v*>(gin + ((((long long)n * H + 2 * p2 + dh) * W + 2 * q2 + dw) * c8n + c8) * 8) = v;
}
// a wave owns 2 rows x 8 pixels like store_2x2, but lane -> (pixel, c8) so that one store instruction is 1 KiB contiguous
__global__ void store_rows(unsigned short* gin, int N, int H, int W, int C) {
    const int c8n = C >> 3;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;      // one thread per input cell pair (h, h+1)
    if (t >= (unsigned)N * (H >> 1) * W * c8n) return;
    const unsigned r = t / c8n; const int c8 = t - r * c8n;
    const unsigned r2 = r / W; const int w = r - r2 * W;
    const int n = r2 / (H >> 1), p2 = r2 - n * (H >> 1);
    s8v v = (s8v){1, 2, 3, 4, 5, 6, 7, (short)t};
    for (int dh = 0; dh < 2; ++dh)
        *reinterpret_cast<s8v*>(gin + ((((long long)n * H + 2 * p2 + dh) * W + w) * c8n + c8) * 8) = v;
}
__global__ void store_flat(s8v* gin, long long n) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) gin[t] = (s8v){1, 2, 3, 4, 5, 6, 7, (short)t};
}
__global__ void load_2x2(const unsigned short* gout, const signed char* idx, const unsigned short* pooled, int* sink,
                         int N, int P, int Q, int C) {
    const int c8n = C >> 3;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (unsigned)N * P * Q * c8n) return;
    const unsigned r = t / c8n; const int c8 = t - r * c8n;
    const unsigned r2 = r / Q; const int q2 = r - r2 * Q;
    const int n = r2 / P, p2 = r2 - n * P;
    int acc = 0;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            const int p = min(p2 + a, P - 1), q = min(q2 + b, Q - 1);
            const long long o = (((long long)n * P + p) * Q + q) * C + c8 * 8;
            const unsigned long long pk = *reinterpret_cast<const unsigned long long*>(idx + o);
            const s8v g = *reinterpret_cast<const s8v*>(gout + o);
            const s8v m = *reinterpret_cast<const s8v*>(pooled + o);
            acc += (int)pk + g[0] + g[7] + m[1] + m[6];
        }
    if (acc == 0x12345678) sink[0] = acc;
}
template <class F> static float timed(F f, int n = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < n; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n;
}
int main() {
    const int N = 16, H = 256, W = 256, C = 64, P = 128, Q = 128;
    const long long nin = (long long)N * H * W * C, nout = (long long)N * P * Q * C;
    unsigned short *gin, *gout, *pooled; signed char* idx; int* sink;
    hipMalloc(&gin, nin * 2); hipMalloc(&gout, nout * 2); hipMalloc(&pooled, nout * 2); hipMalloc(&idx, nout); hipMalloc(&sink, 4);
    hipMemset(gout, 0x3c, nout * 2); hipMemset(pooled, 0x3c, nout * 2); hipMemset(idx, 4, nout);
    const unsigned cells = (unsigned)(N * P * Q * (C / 8));
    printf("store 2x2 pattern   %7.1f us (%.0f MB)\n", timed([&] { hipLaunchKernelGGL(store_2x2, dim3((cells + 255) / 256), dim3(256), 0, 0, gin, N, H, W, C); }), nin * 2 / 1e6);
    printf("store row pairs     %7.1f us\n", timed([&] { hipLaunchKernelGGL(store_rows, dim3((2 * cells + 255) / 256), dim3(256), 0, 0, gin, N, H, W, C); }));
    printf("store flat          %7.1f us\n", timed([&] { hipLaunchKernelGGL(store_flat, dim3((unsigned)((nin / 8 + 255) / 256)), dim3(256), 0, 0, (s8v*)gin, nin / 8); }));
    printf("load 4-cell pattern %7.1f us (%.0f MB unique)\n", timed([&] { hipLaunchKernelGGL(load_2x2, dim3((cells + 255) / 256), dim3(256), 0, 0, gout, idx, pooled, sink, N, P, Q, C); }), (nout * 5) / 1e6);
    printf("maxpool_bwd_pooled  %7.1f us\n", timed([&] { mi355_maxpool3x3s2_bwd_pooled(gout, (const int8_t*)idx, gin, pooled, N, H, W, C, nullptr); }));
    printf("maxpool_bwd (mask)  %7.1f us\n", timed([&] { mi355_maxpool3x3s2_bwd(gout, (const int8_t*)idx, gin, gin, N, H, W, C, nullptr); }));
    return 0;
}
