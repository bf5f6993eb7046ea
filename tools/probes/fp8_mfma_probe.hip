// Probe: semantics of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3, all block scales 1.0) and v_cvt_pk_fp8_f32 on gfx950.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/fp8_probe tools/probes/fp8_mfma_probe.hip && /tmp/fp8_probe
// Assumed (and checked against a host reference here): lane l supplies row (l & 15) of its operand, 32 consecutive k starting at
// 32 * (l >> 4); D[r] of lane l is element (i = 4 * (l >> 4) + r of the FIRST operand, j = l & 15 of the SECOND operand);
// E8M0 scale byte 127 = 1.0; the converter produces the OCP e4m3 encoding the MFMA consumes (448 = 0x7E, saturating).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int v8i32;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void probe(const float* A, const float* B, float* D, unsigned* enc) {
    const int l = threadIdx.x, row = l & 15, kg = l >> 4;
    v8i32 va, vb;
    for (int w = 0; w < 8; ++w) {
        int x = 0, y = 0;
        const float* pa = A + row * 128 + kg * 32 + w * 4;
        const float* pb = B + row * 128 + kg * 32 + w * 4;
        x = __builtin_amdgcn_cvt_pk_fp8_f32(pa[0], pa[1], x, false);
        x = __builtin_amdgcn_cvt_pk_fp8_f32(pa[2], pa[3], x, true);
        y = __builtin_amdgcn_cvt_pk_fp8_f32(pb[0], pb[1], y, false);
        y = __builtin_amdgcn_cvt_pk_fp8_f32(pb[2], pb[3], y, true);
        va[w] = x; vb[w] = y;
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int r = 0; r < 4; ++r) D[(4 * kg + r) * 16 + row] = c[r];
    if (l == 0) {
        const float t[8] = {448.f, 1000.f, -448.f, 0.4375f, 1.0f, 0.001953125f, 240.f, 464.f};
        int u = 0, v = 0;
        u = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], u, false);
        u = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], u, true);
        v = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], v, false);
        v = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], v, true);
        enc[0] = (unsigned)u; enc[1] = (unsigned)v;
    }
}

int main() {
    float hA[16 * 128], hB[16 * 128], hD[256], ref[256];
    srand(1);
    for (int i = 0; i < 16 * 128; ++i) { hA[i] = (float)(rand() % 7 - 3); hB[i] = (float)(rand() % 9 - 4) * 0.5f; }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int k = 0; k < 128; ++k) s += (double)hA[i * 128 + k] * hB[j * 128 + k];
            ref[i * 16 + j] = (float)s;
        }
    float *dA, *dB, *dD; unsigned* dE; unsigned hE[2];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD); hipMalloc(&dE, 8);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dE);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); hipMemcpy(hE, dE, 8, hipMemcpyDeviceToHost);
    double err = 0, errT = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            err = fmax(err, fabs(hD[i * 16 + j] - ref[i * 16 + j]));
            errT = fmax(errT, fabs(hD[j * 16 + i] - ref[i * 16 + j]));
        }
    printf("max |D - ref| = %g   (transposed reading: %g)\n", err, errT);
    printf("fp8 encodings of {448, 1000, -448, 0.4375 | 1, 2^-9, 240, 464} = %08x %08x\n", hE[0], hE[1]);
    return 0;
}
