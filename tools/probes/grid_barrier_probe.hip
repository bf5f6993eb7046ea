// Development probe: what does a grid-wide barrier cost on MI355X (256 persistent workgroups, one per CU), with the memory ordering a
// fused multi-phase kernel would need between phases (release of this workgroup's writes to device scope, acquire of everybody else's:
// the L2s of the eight XCDs are not coherent with each other)?  Compared with the cost of a dependent kernel boundary (two trivial
// kernels back to back on one stream).  Every workgroup writes a line per phase and reads its right neighbour's line of the previous
// phase (another XCD with the round-robin placement): a stale read is counted.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/grid_barrier_probe tools/probes/grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void phases(unsigned* data, unsigned* counter, int nphase, unsigned* stale, int fence) {
    const int nb = gridDim.x, b = blockIdx.x;
    unsigned bad = 0;
    for (int p = 1; p <= nphase; ++p) {
        if (threadIdx.x < 32) data[b * 32 + threadIdx.x] = (unsigned)p;                       // this phase's output
        if (fence) __threadfence();                                                            // release to device scope
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(p * nb)) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (fence) __threadfence();                                                            // acquire
        if (threadIdx.x < 32) {
            const unsigned v = __hip_atomic_load(&data[((b + 1) % nb) * 32 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bad += v < (unsigned)p;
        }
    }
    if (bad) atomicAdd(stale, bad);
}
__global__ void tiny(unsigned* data, int p) { if (threadIdx.x < 32) data[blockIdx.x * 32 + threadIdx.x] = (unsigned)p; }
int main() {
    unsigned *data, *counter, *stale;
    hipMalloc(&data, 256 * 32 * 4); hipMalloc(&counter, 4); hipMalloc(&stale, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int nphase = 2000;
    for (int fence = 0; fence <= 1; ++fence)
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(counter, 0, 4); hipMemset(stale, 0, 4); hipMemset(data, 0, 256 * 32 * 4);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(phases, dim3(256), dim3(256), 0, 0, data, counter, nphase, stale, fence);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            unsigned h; hipMemcpy(&h, stale, 4, hipMemcpyDeviceToHost);
            printf("grid barrier (256 workgroups, %s): %.2f us per phase, stale neighbour reads %u of %d\n", fence ? "with device-scope fences" : "atomics only", ms * 1e3 / nphase, h, nphase * 256 * 32);
        }
    hipEventRecord(a);
    for (int p = 1; p <= nphase; ++p) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, 0, data, p);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("dependent kernel boundary (trivial kernels back to back on one stream): %.2f us per launch\n", ms * 1e3 / nphase);
    return 0;
}
