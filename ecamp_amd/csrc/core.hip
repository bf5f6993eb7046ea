// Error reporting + ABI version for libecamp_hip.so
#include "common.h"
#include "../../include/ecamp_hip.h"
#include <stdarg.h>

thread_local char g_ecamp_err[512] = {0};

int ecamp_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_ecamp_err, sizeof(g_ecamp_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* ecamp_last_error(void) { return g_ecamp_err; }
extern "C" int ecamp_abi_version(void) { return ECAMP_ABI_VERSION; }
// 0: dtype code ECAMP_BF16 means bfloat16 (libecamp_hip.so); 1: it means IEEE half (libecamp_hip_f16.so, built with -DECAMP_HALF_F16)
extern "C" int ecamp_half_format(void) { return ECAMP_HALF_IS_F16; }

// Development aid (tools/hog_probe.py): `blocks` workgroups that spin for `cycles` shader clocks -- a stand-in for a communication
// kernel (RCCL all-reduce) that shares the GPU with the training step on another stream.
__global__ void dev_spin_kernel(long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
}
extern "C" int ecamp_dev_spin(int32_t blocks, int32_t threads, int64_t cycles, hipStream_t stream) {
    ECAMP_CHECK_ARG(blocks > 0 && threads > 0 && threads <= 1024 && cycles >= 0, "dev_spin: bad arguments");
    hipLaunchKernelGGL(dev_spin_kernel, dim3(blocks), dim3(threads), 0, stream, (long long)cycles);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
