// Probe: operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3, block scales 2^0) on gfx950, as the persistent e4m3
// GEMM (ecamp_amd/csrc/gemm_q8.h, F8) assumes it:  lane l supplies row (l & 31) of its operand, 32 consecutive k starting at
// 32 * (l >> 5);  D[r] of lane l is element (i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) of the FIRST operand, j = l & 31 of the SECOND).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/fp8_probe32 tools/probes/fp8_mfma32_probe.hip && /tmp/fp8_probe32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int v8i32;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, row = l & 31, kg = l >> 5;
    v8i32 va, vb;
    for (int w = 0; w < 8; ++w) {
        int x = 0, y = 0;
        const float* pa = A + row * 64 + kg * 32 + w * 4;
        const float* pb = B + row * 64 + kg * 32 + w * 4;
        x = __builtin_amdgcn_cvt_pk_fp8_f32(pa[0], pa[1], x, false);
        x = __builtin_amdgcn_cvt_pk_fp8_f32(pa[2], pa[3], x, true);
        y = __builtin_amdgcn_cvt_pk_fp8_f32(pb[0], pb[1], y, false);
        y = __builtin_amdgcn_cvt_pk_fp8_f32(pb[2], pb[3], y, true);
        va[w] = x; vb[w] = y;
    }
    f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + row] = c[r];
}

int main() {
    static float hA[32 * 64], hB[32 * 64], hD[1024], ref[1024];
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) { hA[i] = (float)(rand() % 7 - 3); hB[i] = (float)(rand() % 9 - 4) * 0.5f; }
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double s = 0;
            for (int k = 0; k < 64; ++k) s += (double)hA[i * 64 + k] * hB[j * 64 + k];
            ref[i * 32 + j] = (float)s;
        }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double err = 0, errT = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            err = fmax(err, fabs(hD[i * 32 + j] - ref[i * 32 + j]));
            errT = fmax(errT, fabs(hD[j * 32 + i] - ref[i * 32 + j]));
        }
    printf("32x32x64 f8f6f4: max |D - ref| = %g   (transposed reading: %g)\n", err, errT);
    return 0;
}
