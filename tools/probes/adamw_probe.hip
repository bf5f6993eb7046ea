// Development probe: the AdamW arena kernel's stream (4 f32 reads + 3 f32 writes + 1 bf16 write per element, 183 M elements) by variant:
//   0 as in optim.hip   1 non-temporal loads and stores   2 non-temporal stores only   3 two float4 per thread and iteration
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/adamw_probe tools/probes/adamw_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned short bf(float f) { unsigned u = __float_as_uint(f); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
template <int VAR> __device__ __forceinline__ f4 ld(const f4* p) { return (VAR == 1) ? __builtin_nontemporal_load(p) : *p; }
template <int VAR> __device__ __forceinline__ void st(f4* p, f4 v) { if (VAR == 1 || VAR == 2) __builtin_nontemporal_store(v, p); else *p = v; }
template <int VAR> __device__ __forceinline__ void one(float* p, const float* g, float* m, float* v, unsigned short* p16, long i, float lr, float wd, float& ss) {
    f4 pp = ld<VAR>((const f4*)p + i), gg = ld<VAR>((const f4*)g + i), mm = ld<VAR>((const f4*)m + i), vv = ld<VAR>((const f4*)v + i);
    us4 h;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float gr = gg[r]; ss += gr * gr;
        pp[r] *= 1.0f - lr * wd;
        mm[r] = 0.9f * mm[r] + 0.1f * gr;
        vv[r] = 0.95f * vv[r] + 0.05f * gr * gr;
        pp[r] -= lr * (mm[r] / (sqrtf(vv[r]) + 1e-8f));
        h[r] = bf(pp[r]);
    }
    st<VAR>((f4*)p + i, pp); st<VAR>((f4*)m + i, mm); st<VAR>((f4*)v + i, vv);
    if (VAR == 1 || VAR == 2) __builtin_nontemporal_store(h, (us4*)p16 + i); else ((us4*)p16)[i] = h;
}
template <int VAR>
__global__ __launch_bounds__(256) void adamw(float* p, const float* g, float* m, float* v, unsigned short* p16, const unsigned char* grp, long n4, float* sumsq) {
    float ss = 0.f;
    if (VAR == 3) {
        for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n4; i += (long)gridDim.x * blockDim.x * 2) {
            if (grp[i >> 4] >= 8) continue;
            one<0>(p, g, m, v, p16, i, 1e-4f, 0.05f, ss);
            one<0>(p, g, m, v, p16, i + 1, 1e-4f, 0.05f, ss);
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
            if (grp[i >> 4] >= 8) continue;
            one<VAR>(p, g, m, v, p16, i, 1e-4f, 0.05f, ss);
        }
    }
    if (ss == 12345.f) atomicAdd(sumsq, ss);
}
int main() {
    const long n = 183173120, n4 = n / 4;
    float *p, *g, *m, *v, *ss; unsigned short* p16; unsigned char* grp;
    hipMalloc(&p, n * 4); hipMalloc(&g, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4); hipMalloc(&p16, n * 2); hipMalloc(&grp, n / 64); hipMalloc(&ss, 4);
    hipMemset(p, 0, n * 4); hipMemset(g, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4); hipMemset(grp, 0, n / 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep)
    for (int nb : {2048, 8192, 32768}) {
        float t[4];
#define RUN(V) { adamw<V><<<nb, 256>>>(p, g, m, v, p16, grp, n4, ss); hipDeviceSynchronize(); hipEventRecord(a); for (int i = 0; i < 5; ++i) adamw<V><<<nb, 256>>>(p, g, m, v, p16, grp, n4, ss); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&t[V], a, b); t[V] /= 5; }
        RUN(0) RUN(1) RUN(2) RUN(3)
        printf("blocks %5d:  as is %.3f ms (%.2f TB/s)   nt loads+stores %.3f (%.2f)   nt stores %.3f (%.2f)   2 x float4 %.3f (%.2f)\n", nb,
               t[0], n * 30.0 / t[0] / 1e9, t[1], n * 30.0 / t[1] / 1e9, t[2], n * 30.0 / t[2] / 1e9, t[3], n * 30.0 / t[3] / 1e9);
    }
    return 0;
}
