// LayerNorm forward / backward.  HBM-bound: one wave64 per row, 4-element vector
// accesses, wave-shuffle reductions, fp32 statistics.  The BERT post-LN form
//     y = LN( dropout(x) + residual )
// is fused: the Philox keep-mask is regenerated from (seed, offset, element index) in the backward pass,
// so no mask tensor is stored; `z = dropout(x) + residual` is written once because backward needs it.
#include "common.h"

template <typename T, int IT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ zout,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                     long rows, int cols, float eps, float drop_p, uint64_t seed,
                                                     uint64_t offset) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = cols >> 2;
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    float v[IT][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        int c = lane + 64 * i;
        if (c < nv) {
            ld4<T>(x + row * cols + c * 4, v[i]);
            if (drop_p > 0.f) {
                float m[4];
                dropout_scale4(seed, offset, (uint64_t)(row * nv + c), drop_p, inv_keep, m);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i][r] *= m[r];
            }
            if (res) {
                float q[4];
                ld4<T>(res + row * cols + c * 4, q);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i][r] += q[r];
            }
            if (zout) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i][r] = rnd<T>(v[i][r]);
                st4<T>(zout + row * cols + c * 4, v[i]);
            }
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i][r] = 0.f;
        }
    }
    const float mu = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        int c = lane + 64 * i;
        if (c < nv) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float d = v[i][r] - mu;
                q += d * d;
            }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)cols + eps);
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        int c = lane + 64 * i;
        if (c < nv) {
            float4 g = *reinterpret_cast<const float4*>(gamma + c * 4);
            float4 b = *reinterpret_cast<const float4*>(beta + c * 4);
            float o[4];
            o[0] = (v[i][0] - mu) * rs * g.x + b.x;
            o[1] = (v[i][1] - mu) * rs * g.y + b.y;
            o[2] = (v[i][2] - mu) * rs * g.z + b.z;
            o[3] = (v[i][3] - mu) * rs * g.w + b.w;
            st4<T>(y + row * cols + c * 4, o);
        }
    }
}

// dz = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma ;  dz += dres_in
// dx_drop = dz * keepmask/(1-p)   (gradient reaching the dense output through the dropout)
// dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy: per-lane column partials -> per-wave LDS slices (8 waves at a time) ->
// one global f32 atomic per column per workgroup.  One 16-wave workgroup per CU: a 12800-row call issues 0.4 M global atomics on
// the 48 cache lines of dgamma/dbeta instead of the 1.5 M of a 4-wave / 1024-block layout, whose per-line serialisation at L2
// held the small shapes at 2 TB/s.
// the 4-element vector of a row as it lies in memory (8 B of bf16 / 16 B of f32), converted when it is consumed
template <typename T> struct LnRaw;
template <> struct LnRaw<float> {
    typedef float4 type;
    static __device__ __forceinline__ void cvt(const float4& v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
};
template <> struct LnRaw<bf16_t> {
    typedef uint2 type;
    static __device__ __forceinline__ void cvt(const uint2& v, float (&o)[4]) {
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    }
};

#define LN_BWD_WAVES 16
template <typename T, int IT>
__global__ __launch_bounds__(LN_BWD_WAVES * 64) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const T* __restrict__ dres,
                                                     T* __restrict__ dz, T* __restrict__ dxdrop, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, long rows, int cols, float drop_p,
                                                     uint64_t seed, uint64_t offset) {
    extern __shared__ __attribute__((aligned(16))) float sh[];  // [8][cols]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = cols >> 2;
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    float ag[IT][4], ab[IT][4], gm[IT][4];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        int c = lane + 64 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) ag[i][r] = ab[i][r] = 0.f;
        if (c < nv) {
            float4 g = *reinterpret_cast<const float4*>(gamma + c * 4);
            gm[i][0] = g.x; gm[i][1] = g.y; gm[i][2] = g.z; gm[i][3] = g.w;
        } else {
            gm[i][0] = gm[i][1] = gm[i][2] = gm[i][3] = 0.f;
        }
    }
    // rows are software-pipelined: the loads of a wave's NEXT row are in flight while the current row is reduced, normalised and
    // stored (a row is a dependent chain load -> two wave reductions -> store; without the prefetch a CU has no load outstanding
    // for about half of it)
    typedef typename LnRaw<T>::type raw_t;
    raw_t rdy[IT], rz[IT], rq[IT];
    float nmu = 0.f, nrs = 0.f;
    const long stride = (long)gridDim.x * LN_BWD_WAVES;
    auto fetch = [&](long row) {
        nmu = mean[row];
        nrs = rstd[row];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                rdy[i] = *reinterpret_cast<const raw_t*>(dy + row * cols + c * 4);
                rz[i] = *reinterpret_cast<const raw_t*>(z + row * cols + c * 4);
                if (dres) rq[i] = *reinterpret_cast<const raw_t*>(dres + row * cols + c * 4);
            }
        }
    };
    long row = (long)blockIdx.x * LN_BWD_WAVES + wave;
    if (row < rows) fetch(row);
    for (; row < rows; row += stride) {
        const float mu = nmu, rs = nrs;
        float xh[IT][4], g[IT][4];
        raw_t cq[IT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            int c = lane + 64 * i;
            cq[i] = rq[i];
            if (c < nv) {
                float d[4], zz[4];
                LnRaw<T>::cvt(rdy[i], d);
                LnRaw<T>::cvt(rz[i], zz);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xh[i][r] = (zz[r] - mu) * rs;
                    g[i][r] = d[r] * gm[i][r];
                    s1 += g[i][r];
                    s2 += g[i][r] * xh[i][r];
                    ag[i][r] += d[r] * xh[i][r];
                    ab[i][r] += d[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) xh[i][r] = g[i][r] = 0.f;
            }
        }
        if (row + stride < rows) fetch(row + stride);
        s1 = wave_sum(s1) / (float)cols;
        s2 = wave_sum(s2) / (float)cols;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            int c = lane + 64 * i;
            if (c < nv) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = rs * (g[i][r] - s1 - xh[i][r] * s2);
                if (dres) {
                    float q[4];
                    LnRaw<T>::cvt(cq[i], q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] += q[r];
                }
                st4<T>(dz + row * cols + c * 4, o);
                if (dxdrop) {
                    if (drop_p > 0.f) {
                        float m[4];
                        dropout_scale4(seed, offset, (uint64_t)(row * nv + c), drop_p, inv_keep, m);
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = rnd<T>(o[r]) * m[r];
                    }
                    st4<T>(dxdrop + row * cols + c * 4, o);
                }
            }
        }
    }
    // workgroup reduction of the per-lane column partials, then one global atomic per column
    for (int pass = 0; pass < 2; ++pass) {
        float tot[2] = {0.f, 0.f};   // columns threadIdx.x and threadIdx.x + 1024 (cols <= 2048)
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
            if ((wave >> 3) == half) {
#pragma unroll
                for (int i = 0; i < IT; ++i) {
                    int c = lane + 64 * i;
                    if (c < nv)
                        *reinterpret_cast<float4*>(sh + (wave & 7) * cols + c * 4) =
                            pass == 0 ? make_float4(ag[i][0], ag[i][1], ag[i][2], ag[i][3]) : make_float4(ab[i][0], ab[i][1], ab[i][2], ab[i][3]);
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = threadIdx.x + j * LN_BWD_WAVES * 64;
                if (c < cols) {
#pragma unroll
                    for (int w = 0; w < 8; ++w) tot[j] += sh[w * cols + c];
                }
            }
        }
        float* dst = pass == 0 ? dgamma : dbeta;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = threadIdx.x + j * LN_BWD_WAVES * 64;
            if (c < cols) atomicAdd(dst + c, tot[j]);
        }
    }
}

template <typename T>
static int ln_fwd_launch(const void* x, const void* res, void* z, const float* gamma, const float* beta, void* y, float* mean,
                         float* rstd, long rows, int cols, float eps, float p, uint64_t seed, uint64_t off, hipStream_t st) {
    dim3 grid(ceil_div(rows, 4)), block(256);
    const int it = ceil_div(cols / 4, 64);
#define L(IT_) hipLaunchKernelGGL((ln_fwd_kernel<T, IT_>), grid, block, 0, st, (const T*)x, (const T*)res, (T*)z, gamma, beta, (T*)y, mean, rstd, rows, cols, eps, p, seed, off)
    if (it <= 1) L(1); else if (it <= 2) L(2); else if (it <= 3) L(3); else if (it <= 4) L(4); else L(8);
#undef L
    return 0;
}

extern "C" int ecamp_layernorm_fwd(const void* x, const void* residual, void* z_out, const float* gamma, const float* beta,
                                   void* y, float* mean, float* rstd, int64_t rows, int32_t cols, float eps, float drop_p,
                                   uint64_t seed, uint64_t offset, int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
    ECAMP_CHECK_ARG(cols % 4 == 0 && cols <= 2048 && rows > 0, "layernorm_fwd: cols=%d must be a multiple of 4 and <= 2048", cols);
    ECAMP_CHECK_ARG(!(residual || drop_p > 0.f) || z_out, "layernorm_fwd: fused residual/dropout needs z_out");
    if (dtype == ECAMP_F32) ln_fwd_launch<float>(x, residual, z_out, gamma, beta, y, mean, rstd, rows, cols, eps, drop_p, seed, offset, stream);
    else if (dtype == ECAMP_BF16) ln_fwd_launch<bf16_t>(x, residual, z_out, gamma, beta, y, mean, rstd, rows, cols, eps, drop_p, seed, offset, stream);
    else return ecamp_set_error(-1, "layernorm_fwd: bad dtype %d", dtype);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int ln_bwd_launch(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                         const void* dres, void* dz, void* dxdrop, float* dgamma, float* dbeta, long rows, int cols, float p,
                         uint64_t seed, uint64_t off, hipStream_t st) {
    int nb = ceil_div(rows, LN_BWD_WAVES);
    if (nb > 256) nb = 256;
    dim3 grid(nb), block(LN_BWD_WAVES * 64);
    size_t shm = (size_t)8 * cols * sizeof(float);
    const int it = ceil_div(cols / 4, 64);
#define L(IT_) hipLaunchKernelGGL((ln_bwd_kernel<T, IT_>), grid, block, shm, st, (const T*)dy, (const T*)z, mean, rstd, gamma, (const T*)dres, (T*)dz, (T*)dxdrop, dgamma, dbeta, rows, cols, p, seed, off)
    if (it <= 1) L(1); else if (it <= 2) L(2); else if (it <= 3) L(3); else if (it <= 4) L(4); else L(8);
#undef L
    return 0;
}

extern "C" int ecamp_layernorm_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                                   const void* dres_in, void* dz, void* dx_drop, float* dgamma, float* dbeta, int64_t rows,
                                   int32_t cols, float drop_p, uint64_t seed, uint64_t offset, int32_t dtype,
                                   hipStream_t stream) {
    ECAMP_CHECK_ARG(dy && z && mean && rstd && gamma && dz && dgamma && dbeta, "layernorm_bwd: null pointer");
    ECAMP_CHECK_ARG(cols % 4 == 0 && cols <= 2048 && rows > 0, "layernorm_bwd: cols=%d must be a multiple of 4 and <= 2048", cols);
    if (dtype == ECAMP_F32) ln_bwd_launch<float>(dy, z, mean, rstd, gamma, dres_in, dz, dx_drop, dgamma, dbeta, rows, cols, drop_p, seed, offset, stream);
    else if (dtype == ECAMP_BF16) ln_bwd_launch<bf16_t>(dy, z, mean, rstd, gamma, dres_in, dz, dx_drop, dgamma, dbeta, rows, cols, drop_p, seed, offset, stream);
    else return ecamp_set_error(-1, "layernorm_bwd: bad dtype %d", dtype);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
