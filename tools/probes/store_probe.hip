// Development probe: HBM write bandwidth of tile-shaped store streams (what a GEMM epilogue produces) vs a linear fill.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_probe tools/probes/store_probe.hip && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// each block of 512 threads writes ROWS rows x SEGB bytes at row pitch `pitch` (bytes); tiles laid out row-major over the matrix
template <int NT>
__global__ __launch_bounds__(512) void tile_store(unsigned char* out, long pitch, int rows, int segb, int tiles_per_row, int ntiles) {
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tr = t / tiles_per_row, tc = t % tiles_per_row;
        unsigned char* base = out + (long)tr * rows * pitch + (long)tc * segb;
        const int lanes_per_row = segb / 16;
        const int rows_per_pass = 512 / lanes_per_row;
        const int r0 = threadIdx.x / lanes_per_row, c = threadIdx.x % lanes_per_row;
        u32x4 v = {(unsigned)t, 1u, 2u, 3u};
        for (int r = r0; r < rows; r += rows_per_pass) {
            u32x4* p = reinterpret_cast<u32x4*>(base + (long)r * pitch + c * 16);
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}
// MFMA-like scatter: lane (lrow = lane&15, lk = lane>>4) writes 16 B at row lrow, byte (lk*16 + h*64) of a 128-B row segment per wave
template <int NT>
__global__ __launch_bounds__(512) void frag_store(unsigned char* out, long pitch, int tiles_per_row, int ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lrow = lane & 15, lk = lane >> 4;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tr = t / tiles_per_row, tc = t % tiles_per_row;
        unsigned char* base = out + ((long)tr * 256 + (wave >> 2) * 128) * pitch + (long)tc * 512 + (wave & 3) * 128;
        u32x4 v = {(unsigned)t, 1u, 2u, 3u};
        for (int tm = 0; tm < 8; ++tm)
            for (int h = 0; h < 2; ++h) {
                u32x4* p = reinterpret_cast<u32x4*>(base + (long)(tm * 16 + lrow) * pitch + lk * 16 + h * 64);
                if (NT) __builtin_nontemporal_store(v, p); else *p = v;
            }
    }
}
// The Q8 epilogue's store stream: wave (wr, wc) of 8 owns 128 rows x 128 B (one cache line per row) of a 256-row x 512-B tile.
//   PAT 0  what gemm_q8.h issues: lane (l31, lh) writes 16 B at row tm*32 + l31, byte NH*64 + gq*32 + lh*16 -> 32 lines x 32 B per instruction
//   PAT 1  whole lines: instruction i writes rows i*8 + lane/8, byte (lane%8)*16                              ->  8 lines x 128 B per instruction
//   PAT 2  half lines:  instruction i writes rows (i/2)*16 + lane/4, byte (i%2)*64 + (lane%4)*16                -> 16 lines x 64 B per instruction
template <int PAT>
__global__ __launch_bounds__(512) void q8_store(unsigned char* out, long pitch, int tiles_per_row, int ntiles, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
    for (int rep = 0; rep < reps; ++rep)
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tr = t / tiles_per_row, tc = t % tiles_per_row;
        unsigned char* base = out + ((long)tr * 256 + (wave >> 2) * 128) * pitch + (long)tc * 512 + (wave & 3) * 128;
        u32x4 v = {(unsigned)t, 1u, 2u, (unsigned)rep};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            long row; int byte;
            if (PAT == 0) { const int tm = i >> 2, nh = (i >> 1) & 1, gq = i & 1; row = tm * 32 + l31; byte = nh * 64 + gq * 32 + lh * 16; }
            else if (PAT == 1) { row = i * 8 + (lane >> 3); byte = (lane & 7) * 16; }
            else { row = (i >> 1) * 16 + (lane >> 2); byte = (i & 1) * 64 + (lane & 3) * 16; }
            *reinterpret_cast<u32x4*>(base + row * pitch + byte) = v;
        }
    }
}
__global__ void linear_fill(u32x4* out, long n) {
    u32x4 v = {0u, 1u, 2u, 3u};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = v;
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
    const long M = 32768; const long pitches[2] = {60000, 6144};
    unsigned char* buf; hipMalloc(&buf, M * 60000 + 4096);
    {
        long n = M * 60000 / 16;
        float ms = timeit([&] { linear_fill<<<2048, 256>>>((u32x4*)buf, n); });
        printf("linear fill                         : %.2f TB/s\n", n * 16.0 / ms / 1e9);
    }
    {   // the Q8 epilogue's stream by lane pattern: HBM-sized output (206 MB, one pass) and an Infinity-Cache-sized one (16 MB, 12 passes)
        const char* pn[3] = {"32 lines x 32 B (gemm_q8.h)", " 8 lines x 128 B", "16 lines x 64 B"};
        for (int big = 1; big >= 0; --big) {
            const long pitch = big ? 4096 : 4096, rows = big ? 50432 : 4096;
            const int tpr = (int)(pitch / 512), nt = (int)(rows / 256) * tpr, reps = big ? 1 : 12;
            const double bytes = (double)nt * 256 * 512 * reps;
            for (int pat = 0; pat < 3; ++pat) {
                float ms = timeit([&] {
                    if (pat == 0) q8_store<0><<<256, 512>>>(buf, pitch, tpr, nt, reps);
                    if (pat == 1) q8_store<1><<<256, 512>>>(buf, pitch, tpr, nt, reps);
                    if (pat == 2) q8_store<2><<<256, 512>>>(buf, pitch, tpr, nt, reps);
                });
                printf("Q8 tile stream, %s output, %-28s: %.2f TB/s  (%.1f us per 256-tile round)\n", big ? "206 MB" : " 16 MB x 12", pn[pat], bytes / ms / 1e9,
                       ms * 1e3 / (bytes / (256.0 * 131072)));
            }
        }
    }
    for (int pi = 0; pi < 2; ++pi) {
        const long pitch = pitches[pi]; const long rowsM = pi == 0 ? M : 12800 * 8;  // keep ~2 GB / 0.63 GB
        const long mrows = pi == 0 ? 32768 : 102400;
        for (int segb : {512, 1024, 2048, 4096}) {
            if (pitch % 16 || segb > pitch) continue;
            const int tpr = (int)(pitch / segb); const int rows = 256 * 512 / segb;  // 128 KB per tile always
            const int nt = (int)(mrows / rows) * tpr;
            const double bytes = (double)nt * rows * segb;
            for (int grid : {256, 2048}) {
                float ms0 = timeit([&] { tile_store<0><<<grid, 512>>>(buf, pitch, rows, segb, tpr, nt); });
                float ms1 = timeit([&] { tile_store<1><<<grid, 512>>>(buf, pitch, rows, segb, tpr, nt); });
                printf("pitch %6ld tile %4d rows x %4d B grid %4d: %.2f TB/s   nontemporal %.2f TB/s\n", pitch, rows, segb, grid, bytes / ms0 / 1e9, bytes / ms1 / 1e9);
            }
        }
        const int tpr = (int)(pitch / 512); const int nt = (int)(mrows / 256) * tpr; const double bytes = (double)nt * 256 * 512;
        float ms0 = timeit([&] { frag_store<0><<<256, 512>>>(buf, pitch, tpr, nt); });
        float ms1 = timeit([&] { frag_store<1><<<256, 512>>>(buf, pitch, tpr, nt); });
        printf("pitch %6ld MFMA-fragment scatter 256x512B grid 256: %.2f TB/s   nontemporal %.2f TB/s\n", pitch, bytes / ms0 / 1e9, bytes / ms1 / 1e9);
    }
    return 0;
}
