// Issue rate of v_exp_f32, v_fma_f32, v_pk_fma_f32 and v_cvt_pk_bf16_f32 on one SIMD: cycles per wave64 instruction from a loop of
// independent instructions run by ONE wave per SIMD (4 waves per workgroup, 1 workgroup).  hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void probe(float* out, long long* cyc, int iters) {
    float a[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = (f32x2){a[i], a[i] + 1.f}; }
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
            if (OP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
            if (OP == 5) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i]));
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    const char* names[] = {"v_exp_f32", "v_fma_f32", "v_pk_fma_f32", "v_cvt_pk_bf16_f32", "v_pk_mul_f32", "v_max3_f32"};
    const int iters = 4096;
    for (int waves = 1; waves <= 4; waves *= 2)
        for (int op = 0; op < 6; ++op) {
            long long h = 0;
            for (int rep = 0; rep < 2; ++rep) {
                dim3 b(256 * waves);
                if (op == 0) hipLaunchKernelGGL(probe<0>, dim3(1), b, 0, 0, out, cyc, iters);
                if (op == 1) hipLaunchKernelGGL(probe<1>, dim3(1), b, 0, 0, out, cyc, iters);
                if (op == 2) hipLaunchKernelGGL(probe<2>, dim3(1), b, 0, 0, out, cyc, iters);
                if (op == 3) hipLaunchKernelGGL(probe<3>, dim3(1), b, 0, 0, out, cyc, iters);
                if (op == 4) hipLaunchKernelGGL(probe<4>, dim3(1), b, 0, 0, out, cyc, iters);
                if (op == 5) hipLaunchKernelGGL(probe<5>, dim3(1), b, 0, 0, out, cyc, iters);
                hipDeviceSynchronize();
                hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            }
            // clock64 counts at a fixed 100 MHz-class rate on gfx9? report raw per-instruction ticks and the ratio to v_fma below
            printf("%d wave(s)/SIMD  %-20s %8.3f ticks per wave-instruction per wave\n", waves, names[op], (double)h / (iters * 8.0));
        }
    return 0;
}
