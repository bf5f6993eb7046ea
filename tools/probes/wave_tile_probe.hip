// Development probe: what can the main loop of a 256 x 256 x 64 bf16 tile reach on one CU, by WAVE tile?
//   A  8 waves (two per SIMD), 128 x 64 per wave : 6 fragment reads (ds_read_b128, 1 KB each) + 8 MFMAs per k-step   (gemm_q8.h)
//   B  4 waves (one per SIMD), 128 x 128 per wave: 8 fragment reads + 16 MFMAs per k-step, fragments double-buffered in the wave
// Operands are whatever is in LDS (no result is checked); optional LDS-DMA of 64 KB per K tile (4 k-steps) from an L2-resident
// matrix, as the real loop issues it.  Prints the MFMA rate of the whole chip (every CU runs one workgroup).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/wave_tile_probe tools/probes/wave_tile_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int WAVES, int FM, int FN, bool DMA>
__global__ __launch_bounds__(WAVES * 64) void loop(const unsigned char* src, int ktiles, float* sink, int random_data) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 160 KB: one workgroup per CU
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {   // operands: zeros, or bf16 values in (-2, 2) with random mantissas -- the matrix pipes then toggle as in a real GEMM (power!)
        unsigned x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
        for (int i = threadIdx.x; i < 160 * 1024 / 4; i += WAVES * 64) {
            x = x * 1664525u + 1013904223u;
            reinterpret_cast<unsigned*>(lds)[i] = random_data ? ((x & 0x807f807fu) | 0x3f003f00u | ((x >> 3) & 0x00800080u)) : 0u;
        }
        __syncthreads();
    }
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // conflict-free fragment reads: lane l reads 16 B at l*16 of a 1-KB fragment image; fragments of a k-step 4 KB apart
    const unsigned char* base = lds + (wave % 4) * 16384 + lane * 16;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * 65536), 0, 1 << 24, 0x00020000);
    typedef void __attribute__((address_space(3))) lds_void;
    bf16x8 fa[2][FM], fb[2][FN];
    auto rd = [&](int buf, int ks) {
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[buf][i] = *reinterpret_cast<const bf16x8*>(base + ks * 4096 + i * 1024);
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[buf][j] = *reinterpret_cast<const bf16x8*>(base + 65536 + ks * 4096 + j * 1024);
    };
    rd(0, 0);
    for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1;
            rd(cur ^ 1, (ks + 1) & 3);                  // next k-step's fragments while this one's MFMAs run
            if (DMA) {                                   // 64 KB per K tile and CU: 16 KB per k-step = 16 pieces of 1 KB over the waves
#pragma unroll
                for (int p = 0; p < 16 / WAVES; ++p)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + 131072 + ((kt & 1) * 16 + (ks * 16 / 4)) * 0 + (p * WAVES + wave) * 1024),
                                                             16, lane * 16 + (p * WAVES + wave) * 1024 + ks * 16384, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][j], fa[cur][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) s += acc[i][j][0] + acc[i][j][7];
    if (s == 123.456f) sink[0] = s;
}

// C  4 waves (one per SIMD), 128 x 128 per wave as 8 x 8 blocks of v_mfma_f32_16x16x32_bf16 (what the vendor library's 256 x 256 x 64 kernel
//    uses): 16 fragment reads + 64 MFMAs per k-half (32 deep), fragments double-buffered in the wave
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <bool DMA>
__global__ __launch_bounds__(256) void loop16(const unsigned char* src, int ktiles, float* sink, int random_data) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    {
        unsigned x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
        for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) {
            x = x * 1664525u + 1013904223u;
            reinterpret_cast<unsigned*>(lds)[i] = random_data ? ((x & 0x807f807fu) | 0x3f003f00u | ((x >> 3) & 0x00800080u)) : 0u;
        }
        __syncthreads();
    }
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    const unsigned char* base = lds + wave * 16384 + lane * 16;   // A fragments of k-half h: 8 KB at h * 8192; B 64 KB further
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * 65536), 0, 1 << 24, 0x00020000);
    typedef void __attribute__((address_space(3))) lds_void;
    bf16x8 fa[2][8], fb[2][8];
    auto rd = [&](int buf, int h) {
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[buf][i] = *reinterpret_cast<const bf16x8*>(base + h * 8192 + i * 1024);
#pragma unroll
        for (int j = 0; j < 8; ++j) fb[buf][j] = *reinterpret_cast<const bf16x8*>(base + 65536 + h * 8192 + j * 1024);
    };
    rd(0, 0);
    for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            rd(h ^ 1, h ^ 1);
            if (DMA) {
#pragma unroll
                for (int p = 0; p < 8; ++p)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + 131072 + (p * 4 + wave) * 1024 % 28672), 16, lane * 16 + (p * 4 + wave) * 1024 + h * 32768, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[h][i], fb[h][j], acc[i][j], 0, 0, 0);
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 123.456f) sink[0] = s;
}

template <typename F> float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
    unsigned char* src; (void)hipMalloc(&src, 256 * 65536 + (1 << 24)); (void)hipMemset(src, 0, 256 * 65536 + (1 << 24));   // (the DMA'd bytes land outside the fragment images: rate only)
    float* sink; (void)hipMalloc(&sink, 4);
    const int ktiles = 8000, grid = 256;   // ~10 ms per launch: the power controller settles
    const size_t shm = 160 * 1024;
    const double flop = 2.0 * 256 * 256 * 64 * (double)ktiles * grid;
#define RUN(NAME, W, FM, FN, D)                                                                                              \
    do {                                                                                                                     \
        (void)hipFuncSetAttribute((const void*)loop<W, FM, FN, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);    \
        float ms = timeit([&] { loop<W, FM, FN, D><<<grid, W * 64, shm>>>(src, ktiles, sink, rnd); });                            \
        printf("%s %-58s %7.2f ms  %6.0f TFLOP/s  (%.0f cycles per K tile at 2.1 GHz)\n", rnd ? "random operands:" : "zero operands:  ", NAME, ms, flop / ms / 1e9, ms * 1e-3 * 2.1e9 / ktiles); \
    } while (0)
#define RUN16(NAME, D)                                                                                                       \
    do {                                                                                                                     \
        (void)hipFuncSetAttribute((const void*)loop16<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);             \
        float ms = timeit([&] { loop16<D><<<grid, 256, shm>>>(src, ktiles, sink, rnd); });                                   \
        printf("%s %-58s %7.2f ms  %6.0f TFLOP/s  (%.0f cycles per K tile at 2.1 GHz)\n", rnd ? "random operands:" : "zero operands:  ", NAME, ms, flop / ms / 1e9, ms * 1e-3 * 2.1e9 / ktiles); \
    } while (0)
    for (int rnd = 0; rnd < 2; ++rnd) {
    RUN("8 waves, 128 x 64 per wave, LDS reads + MFMA", 8, 4, 2, false);
    RUN("8 waves, 128 x 64 per wave, + LDS-DMA 64 KB per K tile", 8, 4, 2, true);
    RUN("4 waves, 128 x 128 per wave, LDS reads + MFMA", 4, 4, 4, false);
    RUN("4 waves, 128 x 128 per wave, + LDS-DMA 64 KB per K tile", 4, 4, 4, true);
    RUN16("4 waves, 128 x 128 per wave, 16x16x32 MFMAs, LDS reads + MFMA", false);
    RUN16("4 waves, 128 x 128 per wave, 16x16x32 MFMAs, + LDS-DMA", true);
    }
    return 0;
}
