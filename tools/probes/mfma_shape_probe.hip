// Development probe: is the matrix pipes' sustained rate on RANDOM bf16 operands the same for v_mfma_f32_32x32x16_bf16 and
// v_mfma_f32_16x16x32_bf16?  (The vendor library's 256 x 256 x 64 kernel uses the 16 x 16 form and its K loop is 14 % faster than
// gemm_q8.h's; both forms have the same FLOP per pipe cycle on paper.)  One 128 x 128 wave tile per wave, 256 accumulators, operands in
// registers (no LDS, no memory), WPS waves per SIMD on every CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_shape_probe tools/probes/mfma_shape_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ bf16x8 rnd_frag(unsigned& x, int random_data) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) { x = x * 1664525u + 1013904223u; v[i] = random_data ? ((x & 0x807f807fu) | 0x3f003f00u | ((x >> 3) & 0x00800080u)) : 0u; }
    return __builtin_bit_cast(bf16x8, v);
}

template <int SHAPE>   // 32: 32x32x16 (4 x 4 blocks, 4 k-steps per K tile)   16: 16x16x32 (8 x 8 blocks, 2 k-steps per K tile)
__global__ __launch_bounds__(256) void loop(int ktiles, float* sink, int random_data) {
    unsigned x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[4][4];
        bf16x8 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { fa[i] = rnd_frag(x, random_data); fb[i] = rnd_frag(x, random_data); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fa[i]), "+v"(fb[i]));
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    } else {
        f32x4 acc[8][8];
        bf16x8 fa[8], fb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { fa[i] = rnd_frag(x, random_data); fb[i] = rnd_frag(x, random_data); }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
        for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(fa[i]), "+v"(fb[i]));
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    }
    if (s == 12345.678f) sink[0] = s;
}

template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float* sink; hipMalloc(&sink, 64);
    const int ktiles = 40000;   // ~50-100 ms per launch: long enough for the power controller to settle
    for (int rep = 0; rep < 2; ++rep)
        for (int wps = 1; wps <= 2; ++wps)
            for (int rnd = 0; rnd <= 1; ++rnd) {
                const int grid = 256 * wps;   // 256-thread workgroups: one wave per SIMD each; wps of them per CU
                const double fl = (double)grid * 4 * ktiles * 2.0 * 128 * 128 * 64;
                float t32 = timeit([&] { hipLaunchKernelGGL(loop<32>, dim3(grid), dim3(256), 0, 0, ktiles, sink, rnd); });
                float t16 = timeit([&] { hipLaunchKernelGGL(loop<16>, dim3(grid), dim3(256), 0, 0, ktiles, sink, rnd); });
                printf("waves/SIMD %d  %s operands:  32x32x16 %7.1f ms = %5.0f TF   16x16x32 %7.1f ms = %5.0f TF\n", wps, rnd ? "random" : "zero  ",
                       t32, fl / t32 / 1e9, t16, fl / t16 / 1e9);
            }
    return 0;
}
