// Development probe: semantics and issue rate of v_mfma_f32_4x4x4_16B_bf16 (16 independent 4x4x4 products per wave) on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma4_probe tools/probes/mfma4_probe.hip && /tmp/mfma4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ short f2b(float f) { uint32_t u = __float_as_uint(f); return (short)(u >> 16); }
__global__ void sem(float* out) {
    const int lane = threadIdx.x, blk = lane >> 2, i = lane & 3;
    // A[row i][k] = 1 + i*10 + k ; B[k][col i] = (k == 0 ? 1 : 0) * (1 + blk) ... choose so that D tells the mapping
    bf16x4 a, b;
    for (int k = 0; k < 4; ++k) { a[k] = f2b((float)(1 + 10 * i + k)); b[k] = f2b(k == i ? (float)(blk + 1) : 0.f); }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
__global__ void rate(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x4 a = {(short)(0x3f80 + lane), 0x3f80, 0x3f80, 0x3f80}, b = {0x3f80, 0x3f80, (short)0x3f80, 0x3f80};
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c7, 0, 0, 0);
    }
    long long t1 = clock64();
    f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = s[0]; out[1] = (float)(t1 - t0) / (8.0f * iters); }
}
int main() {
    float* d; hipMalloc(&d, 4096); float h[256];
    sem<<<1, 64>>>(d); hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    // expectation if A = lane's row i, B = lane's column i, D column i held by lane: D[r][i] = sum_k A[r][k] B[k][i] = A[r][i]*(blk+1)
    printf("lane 0 (blk 0,i 0): %g %g %g %g   [expect A[r][0]*1 = 1 11 21 31]\n", h[0], h[1], h[2], h[3]);
    printf("lane 1 (blk 0,i 1): %g %g %g %g   [expect A[r][1]*1 = 2 12 22 32]\n", h[4], h[5], h[6], h[7]);
    printf("lane 6 (blk 1,i 2): %g %g %g %g   [expect A[r][2]*2 = 6 26 46 66]\n", h[24], h[25], h[26], h[27]);
    printf("lane 63(blk15,i 3): %g %g %g %g   [expect A[r][3]*16 = 64 224 384 544]\n", h[252], h[253], h[254], h[255]);
    rate<<<1, 64>>>(d, 4096); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("4x4x4 bf16 MFMA, one wave, 8 independent accumulators: %.2f cycles per instruction (2048 flop each)\n", h[1]);
    rate<<<1, 256>>>(d, 4096); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("same, 4 waves (one per SIMD): %.2f cycles per instruction per wave\n", h[1]);
    return 0;
}
