// Development probe: how fast can ONE CU pull GEMM operand tiles from L2, by path?
//   mode 0  buffer_load_dwordx4 ... lds          (LDS-DMA, what gemm_q8.h uses)
//   mode 1  buffer_load_dwordx4 -> VGPR          (no LDS at all: the vector-memory return path alone)
//   mode 2  buffer_load_dwordx4 -> VGPR -> ds_write_b128   (register staging)
//   mode 3  mode 0 with `nt` (aux = 2) loads
// Access pattern of a 256x256x64 bf16 tile's operands: per "K tile" every wave of a 512-thread workgroup issues 8 instructions of
// 8 rows x 128 B (1 KB each; 64 KB per workgroup and K tile); the operand matrix is L2 / Infinity-Cache resident.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/load_path_probe tools/probes/load_path_probe.hip && /tmp/load_path_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void pull(const unsigned char* A, long ld_bytes, int rows, int ktiles, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // this workgroup's 512 rows (two 256-row operand tiles) of the matrix
    const long row0 = ((long)blockIdx.x * 512) % rows;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(A + row0 * ld_bytes), 0, (int)(512 * ld_bytes), 0x00020000);
    unsigned voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (i * 8 + wave) * 8 + (lane >> 3);
        voff[i] = (unsigned)(row * ld_bytes + ((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
    typedef void __attribute__((address_space(3))) lds_void;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        for (int kt = 0; kt < ktiles; ++kt) {
            const unsigned koff = (unsigned)kt * 128u;
            unsigned char* dst = lds + (kt & 1) * 65536;
            if (MODE == 0 || MODE == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + (i * 8 + wave) * 1024), 16, (int)(voff[i] + koff), 0, 0, MODE == 3 ? 2 : 0);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                u32x4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i] + koff, 0, 0);
                if (MODE == 2) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(dst + (i * 8 + wave) * 1024 + lane * 16) = v[i];
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc ^= v[i][0];
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 1) acc ^= *reinterpret_cast<unsigned*>(lds + threadIdx.x * 4);
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
    const int rows = 32768, K = 768;           // 50 MB operand: Infinity-Cache resident, 12 K tiles per row block
    const long ld = (long)K * 2;
    unsigned char* A; hipMalloc(&A, (size_t)rows * ld + 4096); hipMemset(A, 1, (size_t)rows * ld);
    unsigned* sink; hipMalloc(&sink, 4);
    const int ktiles = K / 64, iters = 40;
    const char* names[4] = {"buffer_load ... lds (DMA)", "buffer_load -> VGPR only", "buffer_load -> VGPR -> ds_write_b128", "buffer_load ... lds, nt"};
    for (int grid : {256, 64, 16}) {
        for (int mode = 0; mode < 4; ++mode) {
            auto run = [&] {
                const size_t shm = 131072;
                if (mode == 0) { hipFuncSetAttribute((const void*)pull<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); pull<0><<<grid, 512, shm>>>(A, ld, rows, ktiles, iters, sink); }
                if (mode == 1) { hipFuncSetAttribute((const void*)pull<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); pull<1><<<grid, 512, shm>>>(A, ld, rows, ktiles, iters, sink); }
                if (mode == 2) { hipFuncSetAttribute((const void*)pull<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); pull<2><<<grid, 512, shm>>>(A, ld, rows, ktiles, iters, sink); }
                if (mode == 3) { hipFuncSetAttribute((const void*)pull<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); pull<3><<<grid, 512, shm>>>(A, ld, rows, ktiles, iters, sink); }
            };
            const float ms = timeit(run);
            const double bytes = (double)grid * iters * ktiles * 65536.0;
            printf("grid %3d  %-40s %8.1f us  %6.2f TB/s chip  %6.1f GB/s per CU  (%.1f B/clk at 2.4 GHz)\n", grid, names[mode], ms * 1e3, bytes / ms / 1e9,
                   bytes / grid / ms / 1e6, bytes / grid / (ms * 1e-3) / 2.4e9);
        }
    }
    return 0;
}
