// Probe: throughput of returnless fp32 atomic adds on gfx950 by scope and footprint.
// hipcc --offload-arch=gfx950 -O3 tools/probe_atomics.hip -o /tmp/probe_atomics && /tmp/probe_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(float* buf, long long region_floats, int reps) {
    // every wave adds 64 consecutive floats per instruction; the block walks `reps` rows of 128 floats x 128
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long base = ((long long)blockIdx.x * 16384) % region_floats;
    for (int r = 0; r < reps; ++r)
        for (int rr = wv; rr < 128; rr += 4)
            for (int h = 0; h < 2; ++h) {
                float* p = buf + (base + rr * 128 + h * 64 + lane) % region_floats;
                if (MODE == 0) atomicAdd(p, 1.0f);
                else if (MODE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (MODE == 2) *p = 1.0f;
                else if (MODE == 3) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
}
template <int MODE>
void run(const char* name, float* buf, long long region, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, region, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, region, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double bytes = (double)blocks * 16384 * 4;
    printf("%-28s region %6.1f MB blocks %5d: %8.1f us  %7.2f TB/s\n", name, region * 4 / 1e6, blocks, ms * 1e3, bytes / ms / 1e9);
}
int main() {
    float* buf; hipMalloc(&buf, 1ll << 30); hipMemset(buf, 0, 1ll << 30);
    for (long long region : {1ll << 18, 1ll << 20, 1ll << 23, 1ll << 26}) {     // 1 MB, 4 MB, 32 MB, 256 MB
        for (int blocks : {512, 4096}) {
            run<0>("atomicAdd (agent)", buf, region, blocks);
            run<1>("fetch_add (workgroup scope)", buf, region, blocks);
            run<3>("fetch_add (system scope)", buf, region, blocks);
            run<2>("plain store", buf, region, blocks);
        }
    }
    return 0;
}
