// Small HBM-bound helpers: add, cast, column sums (bias gradients), broadcast add / sequence sums for the
// fusion layer's GAP token, Philox uniform noise, token mean.  All vectorised 4 elements per lane.
#include "common.h"

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float p[4], q[4];
        ld4<T>(a + i * 4, p);
        ld4<T>(b + i * 4, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] += q[r];
        st4<T>(y + i * 4, p);
    }
}
extern "C" int ecamp_add(const void* a, const void* b, void* y, int64_t n, int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(a && b && y && n % 4 == 0, "ecamp_add: bad args (n=%ld must be a multiple of 4)", (long)n);
    long n4 = n / 4;
    int nb = (int)((n4 + 255) / 256);
    if (nb > 4096) nb = 4096;
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(add_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)a, (const float*)b, (float*)y, n4);
    else hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n4);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ s, D* __restrict__ d, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float p[4];
        ld4<S>(s + i * 4, p);
        st4<D>(d + i * 4, p);
    }
}
extern "C" int ecamp_cast(const void* src, void* dst, int64_t n, int32_t src_dtype, int32_t dst_dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(src && dst && n % 4 == 0, "ecamp_cast: bad args (n=%ld)", (long)n);
    long n4 = n / 4;
    int nb = (int)((n4 + 255) / 256);
    if (nb > 4096) nb = 4096;
    if (src_dtype == ECAMP_F32 && dst_dtype == ECAMP_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(nb), dim3(256), 0, stream, (const float*)src, (bf16_t*)dst, n4);
    else if (src_dtype == ECAMP_BF16 && dst_dtype == ECAMP_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(nb), dim3(256), 0, stream, (const bf16_t*)src, (float*)dst, n4);
    else if (src_dtype == ECAMP_F32 && dst_dtype == ECAMP_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(nb), dim3(256), 0, stream, (const float*)src, (float*)dst, n4);
    else if (src_dtype == ECAMP_BF16 && dst_dtype == ECAMP_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(nb), dim3(256), 0, stream, (const bf16_t*)src, (bf16_t*)dst, n4);
    else return ecamp_set_error(-1, "ecamp_cast: bad dtypes");
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// y = x * alpha * (alpha_dev ? *alpha_dev : 1), computed in f32 (x and y may be the same buffer): the upstream gradient applied to a tensor
// that was computed for a unit upstream gradient (the chunked MLM head)
template <typename T>
__global__ void scale_kernel(const T* __restrict__ x, T* __restrict__ y, long n4, float alpha, const float* __restrict__ alpha_dev) {
    const float a = alpha_dev ? alpha * alpha_dev[0] : alpha;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float p[4];
        ld4<T>(x + i * 4, p);
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] *= a;
        st4<T>(y + i * 4, p);
    }
}
extern "C" int ecamp_scale(const void* x, void* y, int64_t n, float alpha, const float* alpha_dev, int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(x && y && n > 0 && n % 4 == 0, "ecamp_scale: bad args (n=%ld)", (long)n);
    const long n4 = n / 4;
    int nb = (int)((n4 + 255) / 256);
    if (nb > 4096) nb = 4096;
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(scale_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)x, (float*)y, n4, alpha, alpha_dev);
    else if (dtype == ECAMP_BF16) hipLaunchKernelGGL(scale_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, n4, alpha, alpha_dev);
    else return ecamp_set_error(-1, "ecamp_scale: bad dtype");
    ECAMP_LAUNCH_CHECK();
    return 0;
}

extern "C" int ecamp_zero(void* p, int64_t bytes, hipStream_t stream) {
    ECAMP_CHECK_ARG(p && bytes >= 0, "ecamp_zero: bad args");
    hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, stream);
    if (e != hipSuccess) return ecamp_set_error((int)e, "ecamp_zero: %s", hipGetErrorString(e));
    return 0;
}

// Zero only the 64-element blocks whose flag is set (the gradient arena's small tensors -- biases, LayerNorm, embeddings, tokens --
// that kernels ACCUMULATE into with atomics; weight matrices are overwritten by their weight-gradient GEMM instead).
__global__ __launch_bounds__(256) void zero_blocks_kernel(float* __restrict__ g, const unsigned char* __restrict__ flags, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        if (flags[i >> 4]) *reinterpret_cast<float4*>(g + i * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
}
extern "C" int ecamp_zero_blocks(float* g, const uint8_t* block_flags, int64_t n, hipStream_t stream) {
    ECAMP_CHECK_ARG(g && block_flags && n > 0 && n % 64 == 0, "ecamp_zero_blocks: bad args");
    const long n4 = n / 4;
    int nb = (int)((n4 + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(zero_blocks_kernel, dim3(nb), dim3(256), 0, stream, g, block_flags, n4);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// out[n] += alpha * sum_{m in rows selected} X[m*ld + n].   Row selection: all rows, or with period `per`
// only rows with (m % per) >= skip_lo  ("skip the cls row of every sample"), or only rows (m % per) < only_hi.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, long ld, long M, int N, float alpha, const float* __restrict__ alpha_dev, int per,
                                                     int lo, int hi, float* __restrict__ out) {
    __shared__ float sh[8][129];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 128 + tx * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < N) {
        for (long m = (long)blockIdx.y * 8 + ty; m < M; m += (long)gridDim.y * 8) {
            if (per > 0) {
                int t = (int)(m % per);
                if (t < lo || t >= hi) continue;
            }
            float p[4];
            ld4<T>(X + m * ld + c0, p);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += p[r];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[ty][tx * 4 + r] = acc[r];
    __syncthreads();
    if (threadIdx.x < 128) {
        int c = blockIdx.x * 128 + threadIdx.x;
        if (c < N) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += sh[j][threadIdx.x];
            atomicAdd(out + c, (alpha_dev ? alpha * alpha_dev[0] : alpha) * t);
        }
    }
}
extern "C" int ecamp_colsum(const void* X, int64_t ld, int64_t M, int64_t N, float alpha, const float* alpha_dev, int32_t period, int32_t lo,
                            int32_t hi, float* out, int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(X && out && N % 4 == 0 && ld % 4 == 0 && M > 0, "ecamp_colsum: bad args");
    int nbx = ceil_div(N, 128);
    int nby = ceil_div(M, 8 * 16);
    if (nby < 1) nby = 1;
    int cap = 2048 / nbx;
    if (cap < 1) cap = 1;
    if (nby > cap) nby = cap;
    dim3 grid(nbx, nby), block(256);
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, stream, (const float*)X, (long)ld, (long)M, (int)N, alpha, alpha_dev, period, lo, hi, out);
    else hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)X, (long)ld, (long)M, (int)N, alpha, alpha_dev, period, lo, hi, out);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// y[b, s, :] = x[b, s, :] + g[b, :]        (context_fusion.py:55: cross-attention output + gap_mlp(gap_token))
template <typename T>
__global__ void bcast_add_kernel(const T* __restrict__ x, const T* __restrict__ g, T* __restrict__ y, long B, int S, int H4) {
    long n4 = B * S * H4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        long b = i / ((long)S * H4);
        int h = (int)(i % H4);
        float p[4], q[4];
        ld4<T>(x + i * 4, p);
        ld4<T>(g + (b * H4 + h) * 4, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] += q[r];
        st4<T>(y + i * 4, p);
    }
}
extern "C" int ecamp_bcast_add(const void* x, const void* g, void* y, int64_t B, int32_t S, int32_t H, int32_t dtype,
                               hipStream_t stream) {
    ECAMP_CHECK_ARG(x && g && y && H % 4 == 0, "ecamp_bcast_add: bad args");
    long n4 = B * S * (H / 4);
    int nb = (int)((n4 + 255) / 256);
    if (nb > 4096) nb = 4096;
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(bcast_add_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)x, (const float*)g, (float*)y, (long)B, S, H / 4);
    else hipLaunchKernelGGL(bcast_add_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)g, (bf16_t*)y, (long)B, S, H / 4);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// out[b, :] = scale * sum_{s in [s0, s1)} x[b, s, :]     (deterministic; one block column-stripe per b)
// used for: GAP token forward (mean over patch tokens, model_ecamp.py:269) and gap gradient (sum over S)
template <typename T>
__global__ __launch_bounds__(256) void seq_sum_kernel(const T* __restrict__ x, T* __restrict__ out, int S, int H, int s0,
                                                      int s1, float scale) {
    __shared__ float sh[8][129];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long b = blockIdx.y;
    const int c0 = blockIdx.x * 128 + tx * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < H) {
        for (int s = s0 + ty; s < s1; s += 8) {
            float p[4];
            ld4<T>(x + (b * S + s) * (long)H + c0, p);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += p[r];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[ty][tx * 4 + r] = acc[r];
    __syncthreads();
    if (threadIdx.x < 128) {
        int c = blockIdx.x * 128 + threadIdx.x;
        if (c < H) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += sh[j][threadIdx.x];
            out[b * H + c] = from_f<T>(t * scale);
        }
    }
}
extern "C" int ecamp_seq_sum(const void* x, void* out, int64_t B, int32_t S, int32_t H, int32_t s0, int32_t s1, float scale,
                             int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(x && out && H % 4 == 0 && s0 >= 0 && s1 <= S && s0 < s1, "ecamp_seq_sum: bad args");
    dim3 grid(ceil_div(H, 128), (unsigned)B), block(256);
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(seq_sum_kernel<float>, grid, block, 0, stream, (const float*)x, (float*)out, S, H, s0, s1, scale);
    else hipLaunchKernelGGL(seq_sum_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)x, (bf16_t*)out, S, H, s0, s1, scale);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// y[b, s, :] (+)= g[b, :] * scale for s in [s0, s1), and optionally zero / keep other rows:
//   mode 0: y[b,s,:] = (s in range ? g*scale : 0)      mode 1: y[b,s,:] += (s in range ? g*scale : 0)
template <typename T>
__global__ void seq_bcast_kernel(const T* __restrict__ g, T* __restrict__ y, long B, int S, int H4, int s0, int s1,
                                 float scale, int mode) {
    long n4 = B * S * H4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        long b = i / ((long)S * H4);
        int s = (int)((i / H4) % S), h = (int)(i % H4);
        float q[4] = {0.f, 0.f, 0.f, 0.f};
        if (s >= s0 && s < s1) {
            ld4<T>(g + (b * H4 + h) * 4, q);
#pragma unroll
            for (int r = 0; r < 4; ++r) q[r] *= scale;
        }
        if (mode == 1) {
            float p[4];
            ld4<T>(y + i * 4, p);
#pragma unroll
            for (int r = 0; r < 4; ++r) q[r] += p[r];
        }
        st4<T>(y + i * 4, q);
    }
}
extern "C" int ecamp_seq_bcast(const void* g, void* y, int64_t B, int32_t S, int32_t H, int32_t s0, int32_t s1, float scale,
                               int32_t mode, int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(g && y && H % 4 == 0, "ecamp_seq_bcast: bad args");
    long n4 = B * S * (H / 4);
    int nb = (int)((n4 + 255) / 256);
    if (nb > 4096) nb = 4096;
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(seq_bcast_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)g, (float*)y, (long)B, S, H / 4, s0, s1, scale, mode);
    else hipLaunchKernelGGL(seq_bcast_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)g, (bf16_t*)y, (long)B, S, H / 4, s0, s1, scale, mode);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// development ABI (tests only): the dropout keep-mask every kernel of this library derives from (seed, offset) -- element e of the
// flattened tensor is kept iff halfword (e & 7) of Philox4x32-7(counter e >> 3) is >= round(65536 p) (dropout_scale, common.h) -- materialised as one
// byte per element, so that a test or the oracle can replay the SAME mask in plain PyTorch (attention probabilities: e = ((b * H + h)
// * Tq + i) * Tk + j; LayerNorm / embedding dropout: e = row * cols + col)
__global__ void dropout_mask_kernel(uint8_t* __restrict__ keep, long n, float p, uint64_t seed, uint64_t offset) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
        keep[e] = dropout_scale(seed, offset, (uint64_t)e, p, 1.0f) != 0.0f ? 1 : 0;
}
extern "C" int ecamp_dropout_mask(uint8_t* keep, int64_t n, float p, uint64_t seed, uint64_t offset, hipStream_t stream) {
    ECAMP_CHECK_ARG(keep && n > 0 && p >= 0.f && p < 1.f, "ecamp_dropout_mask: bad args");
    int nb = (int)((n + 255) / 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(nb), dim3(256), 0, stream, keep, (long)n, p, seed, offset);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// uniform [0,1) noise for MAE masking (stands in for torch.rand(N, L), model_ecamp.py:177)
__global__ void uniform_kernel(float* __restrict__ out, long n, uint64_t seed, uint64_t offset) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i * 4 < n; i += (long)gridDim.x * blockDim.x) {
        uint4 r = philox4x32(seed, offset, (uint64_t)i);
        float u[4] = {(float)(r.x >> 8) * (1.0f / 16777216.0f), (float)(r.y >> 8) * (1.0f / 16777216.0f),
                      (float)(r.z >> 8) * (1.0f / 16777216.0f), (float)(r.w >> 8) * (1.0f / 16777216.0f)};
        for (int k = 0; k < 4; ++k)
            if (i * 4 + k < n) out[i * 4 + k] = u[k];
    }
}
extern "C" int ecamp_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t stream) {
    ECAMP_CHECK_ARG(out && n > 0, "ecamp_uniform: bad args");
    long n4 = (n + 3) / 4;
    int nb = (int)((n4 + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(uniform_kernel, dim3(nb), dim3(256), 0, stream, out, (long)n, seed, offset);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// dx = dy * gelu'(pre)   (BertPredictionHeadTransform: dense -> GELU -> LayerNorm, bert_modeling.py:209)
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ pre, T* __restrict__ dx, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float p[4], q[4];
        ld4<T>(dy + i * 4, p);
        ld4<T>(pre + i * 4, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] *= gelu_grad_t<T>(q[r]);
        st4<T>(dx + i * 4, p);
    }
}
extern "C" int ecamp_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(dy && pre && dx && n % 4 == 0, "ecamp_gelu_bwd: bad args");
    long n4 = n / 4;
    int nb = (int)((n4 + 255) / 256);
    if (nb > 4096) nb = 4096;
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)dy, (const float*)pre, (float*)dx, n4);
    else hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)pre, (bf16_t*)dx, n4);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
