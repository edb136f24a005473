// Hardware-assumption probe for the conv kernels (run once on a real MI355X):
//   hipcc --offload-arch=gfx950 -O2 tools/probe_gfx950.hip -o /tmp/probe && /tmp/probe
// Checks, with exact small-integer data against a host reference:
//  (1) v_mfma_f32_16x16x32_bf16 operand / accumulator lane maps (cdna_hip_programming.md §3);
//  (2) ds_read_b64_tr_b16 semantics used to feed A^T / B from [m][k] row-major LDS tiles
//      (the wgrad kernel's operand path);
//  (3) buffer_load_dwordx4 ... lds: lane-linear LDS destination and ZERO fill for
//      out-of-range offsets (the im2col halo / padding path of the conv kernels).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef short s8v __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
#define LDSP(T) __attribute__((address_space(3))) T*

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

// (1) D[16][16] = A[16][32] * B[32][16]; A row-major [i][k], Bt row-major [j][k].
__global__ void k_mfma(const unsigned short* A, const unsigned short* Bt, float* D) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    s8v a = *(const s8v*)(A + r * 32 + 8 * g);
    s8v b = *(const s8v*)(Bt + r * 32 + 8 * g);
    f4v acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D[(g * 4 + j) * 16 + r] = acc[j];   // row = 4*(lane>>4)+reg, col = lane&15
}

// (2) D[16][16] = sum_m X[m][i] * Y[m][j], X,Y row-major [32][16] in LDS, via tr reads.
__global__ void k_tr(const unsigned short* X, const unsigned short* Y, float* D, unsigned short* dump) {
    __shared__ __attribute__((aligned(16))) unsigned short sx[32 * 16], sy[32 * 16];
    const int l = threadIdx.x;
    for (int i = l; i < 512; i += 64) { sx[i] = X[i]; sy[i] = Y[i]; }
    __syncthreads();
    const int g = l >> 4, q = (l & 15) >> 2, p = l & 3;
    // lane 4q+p of group g addresses row (8g + q [+4]), cols 4p..4p+3
    s4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP(s4v))(sx + (8 * g + q) * 16 + 4 * p));
    s4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP(s4v))(sx + (8 * g + 4 + q) * 16 + 4 * p));
    s4v b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP(s4v))(sy + (8 * g + q) * 16 + 4 * p));
    s4v b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP(s4v))(sy + (8 * g + 4 + q) * 16 + 4 * p));
    s8v a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    s8v b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    for (int j = 0; j < 8; ++j) dump[l * 8 + j] = (unsigned short)a[j];
    f4v acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D[(g * 4 + j) * 16 + (l & 15)] = acc[j];
}

// (3) each lane requests 16 B at byte offset off[lane]; offsets >= nbytes must give zeros.
__global__ void k_buflds(const unsigned short* src, int nbytes, const int* off, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 8 * 2];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds[i] = 0x7777;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDSP(void))lds, 16, off[l], 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDSP(void))(lds + 512), 16, off[63 - l], 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = l; i < 1024; i += 64) out[i] = lds[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

int main() {
    int fails = 0;
    srand(1);
    {   // (1)
        std::vector<unsigned short> A(16 * 32), Bt(16 * 32);
        std::vector<float> Af(16 * 32), Bf(16 * 32), D(256), ref(256, 0.f);
        for (int i = 0; i < 512; ++i) { Af[i] = (float)(rand() % 7 - 3); Bf[i] = (float)(rand() % 5 - 2); A[i] = f2bf(Af[i]); Bt[i] = f2bf(Bf[i]); }
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) ref[i * 16 + j] += Af[i * 32 + k] * Bf[j * 32 + k];
        unsigned short *dA, *dB; float* dD;
        CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 1024));
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice));
        k_mfma<<<1, 64>>>(dA, dB, dD);
        CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 256; ++i) bad += D[i] != ref[i];
        printf("[1] mfma_16x16x32_bf16 lane maps: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    }
    {   // (2)
        std::vector<unsigned short> X(512), Y(512), dump(512);
        std::vector<float> Xf(512), Yf(512), D(256), ref(256, 0.f);
        for (int i = 0; i < 512; ++i) { Xf[i] = (float)(rand() % 7 - 3); Yf[i] = (float)(rand() % 5 - 2); X[i] = f2bf(Xf[i]); Y[i] = f2bf(Yf[i]); }
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int m = 0; m < 32; ++m) ref[i * 16 + j] += Xf[m * 16 + i] * Yf[m * 16 + j];
        unsigned short *dX, *dY, *dd; float* dD;
        CK(hipMalloc(&dX, 1024)); CK(hipMalloc(&dY, 1024)); CK(hipMalloc(&dD, 1024)); CK(hipMalloc(&dd, 1024));
        CK(hipMemcpy(dX, X.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dY, Y.data(), 1024, hipMemcpyHostToDevice));
        k_tr<<<1, 64>>>(dX, dY, dD, dd);
        CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(dump.data(), dd, 1024, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 256; ++i) bad += D[i] != ref[i];
        int bad2 = 0;   // expected: lane l element j == X[m = 8*(l>>4)+j][i = l&15]
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) bad2 += dump[l * 8 + j] != X[(8 * (l >> 4) + j) * 16 + (l & 15)];
        printf("[2] ds_read_b64_tr_b16 -> A^T.B via mfma: %s (%d mismatches; fragment map %d mismatches)\n", (bad || bad2) ? "FAIL" : "PASS", bad, bad2);
        if (bad2) { printf("    lane0 frag:"); for (int j = 0; j < 8; ++j) printf(" %04x", dump[j]); printf("\n    X col0:"); for (int m = 0; m < 8; ++m) printf(" %04x", X[m * 16]); printf("\n    lane1 frag:"); for (int j = 0; j < 8; ++j) printf(" %04x", dump[8 + j]); printf("\n    lane16 frag:"); for (int j = 0; j < 8; ++j) printf(" %04x", dump[128 + j]); printf("\n"); }
        fails += (bad || bad2) != 0;
    }
    {   // (3)
        const int n = 4096;
        std::vector<unsigned short> src(n), out(1024);
        for (int i = 0; i < n; ++i) src[i] = (unsigned short)(i + 1);
        std::vector<int> off(64);
        for (int l = 0; l < 64; ++l) off[l] = (l % 5 == 0) ? (n * 2 + 1024 * l) : ((l * 37) % (n / 8)) * 16;   // some OOB
        off[7] = 0x7ffffff0;                                                                                 // far OOB
        unsigned short *ds, *dout; int* doff;
        CK(hipMalloc(&ds, n * 2)); CK(hipMalloc(&dout, 2048)); CK(hipMalloc(&doff, 256));
        CK(hipMemcpy(ds, src.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(doff, off.data(), 256, hipMemcpyHostToDevice));
        k_buflds<<<1, 64>>>(ds, n * 2, doff, dout);
        CK(hipMemcpy(out.data(), dout, 2048, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int half = 0; half < 2; ++half) for (int l = 0; l < 64; ++l) {
            int o = off[half ? 63 - l : l];
            for (int j = 0; j < 8; ++j) {
                unsigned short expect = (o >= 0 && o + 16 <= n * 2) ? src[o / 2 + j] : 0;
                bad += out[half * 512 + l * 8 + j] != expect;
            }
        }
        printf("[3] buffer_load_dwordx4 lds: lane-linear dest + OOB zero fill: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
        if (bad) { printf("    lane0 (OOB) got:"); for (int j = 0; j < 8; ++j) printf(" %04x", out[j]); printf("\n    lane1 got:"); for (int j = 0; j < 8; ++j) printf(" %04x", out[8 + j]); printf(" expect %04x..\n", src[off[1] / 2]); }
        fails += bad != 0;
    }
    printf("probe: %s\n", fails ? "SOME FAILED" : "ALL PASS");
    return fails ? 1 : 0;
}
