// LayerNorm forward / backward.  HBM-bound: one wave64 per row, 4-element vector
// accesses, wave-shuffle reductions, fp32 statistics.  The BERT post-LN form
//     y = LN( dropout(x) + residual )
// is fused: the Philox keep-mask is regenerated from (seed, offset, element index) in the backward pass,
// so no mask tensor is stored; `z = dropout(x) + residual` is written once because backward needs it.
#include "common.h"

// VEC elements per lane access: 4 (16 B of f32, 8 B of bf16) or, for bf16 rows whose length is a multiple of 8, 8 (16 B -- the
// streaming optimum of this chip; a 768-column row is 96 such chunks = one full wave access and one half-filled one, instead of three
// 8-byte ones).  Round 4: the 8-byte form ran at 3 TB/s inside the step (fwd 27.7 us, bwd 31.8 us on 12800 x 768).
template <typename T, int VEC> struct LnVec;
template <> struct LnVec<float, 4> {
    typedef float4 raw_t;
    static __device__ __forceinline__ void cvt(const float4& v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    static __device__ __forceinline__ float4 pack(const float (&o)[4]) { return make_float4(o[0], o[1], o[2], o[3]); }
};
template <> struct LnVec<bf16_t, 4> {
    typedef uint2 raw_t;
    static __device__ __forceinline__ void cvt(const uint2& v, float (&o)[4]) {
        o[0] = h16_lo(v.x); o[1] = h16_hi(v.x);
        o[2] = h16_lo(v.y); o[3] = h16_hi(v.y);
    }
    static __device__ __forceinline__ uint2 pack(const float (&o)[4]) { return make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])); }
};
template <> struct LnVec<bf16_t, 8> {
    typedef uint4 raw_t;
    static __device__ __forceinline__ void cvt(const uint4& v, float (&o)[8]) {
        o[0] = h16_lo(v.x); o[1] = h16_hi(v.x);
        o[2] = h16_lo(v.y); o[3] = h16_hi(v.y);
        o[4] = h16_lo(v.z); o[5] = h16_hi(v.z);
        o[6] = h16_lo(v.w); o[7] = h16_hi(v.w);
    }
    static __device__ __forceinline__ uint4 pack(const float (&o)[8]) {
        return make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
    }
};
// keep-mask scales of the VEC elements of chunk `c` of a row of `nv` chunks (element index row * cols + c * VEC + r): a whole Philox call
// for an 8-wide chunk, half a call for a 4-wide one (common.h: eight consecutive elements share a call)
template <int VEC>
__device__ __forceinline__ void ln_dropout(uint64_t seed, uint64_t offset, long row, int nv, int c, float p, float inv_keep, float (&m)[VEC]) {
    if constexpr (VEC == 8) {
        dropout_scale8(seed, offset, (uint64_t)(row * nv + c), p, inv_keep, m);
    } else {
        static_assert(VEC == 4, "ln_dropout: 4- or 8-wide chunks");
        dropout_scale4(seed, offset, (uint64_t)(row * nv + c), p, inv_keep, m);
    }
}

template <typename T, int IT, int VEC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ zout,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                     long rows, int cols, float eps, float drop_p, uint64_t seed,
                                                     uint64_t offset, unsigned char* __restrict__ q8, const float* __restrict__ q8_scale,
                                                     float* __restrict__ q8_amax) {
    typedef LnVec<T, VEC> V;
    typedef typename V::raw_t raw_t;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    // q8 (fp8 forward, delayed scaling): an e4m3 copy of the (storage-rounded) output for the GEMM that consumes it, quantised with the
    // site's current scale, and this step's |y| maximum into the site's amax slots -- no separate amax / quantise pass over y
    const float q8_inv = q8 ? 1.0f / fmaxf(q8_scale[0], 1e-30f) : 0.f;
    float q8_max = 0.f;
    const int nv = cols / VEC;
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    // every load of the row is issued before the first value is consumed
    raw_t rx[IT], rr[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            rx[i] = *reinterpret_cast<const raw_t*>(x + row * cols + c * VEC);
            if (res) rr[i] = *reinterpret_cast<const raw_t*>(res + row * cols + c * VEC);
        }
    }
    float v[IT][VEC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            V::cvt(rx[i], v[i]);
            if (drop_p > 0.f) {
                float m[VEC];
                ln_dropout<VEC>(seed, offset, row, nv, c, drop_p, inv_keep, m);
#pragma unroll
                for (int r = 0; r < VEC; ++r) v[i][r] *= m[r];
            }
            if (res) {
                float q[VEC];
                V::cvt(rr[i], q);
#pragma unroll
                for (int r = 0; r < VEC; ++r) v[i][r] += q[r];
            }
            if (zout) {
#pragma unroll
                for (int r = 0; r < VEC; ++r) v[i][r] = rnd<T>(v[i][r]);
                *reinterpret_cast<raw_t*>(zout + row * cols + c * VEC) = V::pack(v[i]);
            }
#pragma unroll
            for (int r = 0; r < VEC; r += 4) s += (v[i][r] + v[i][r + 1]) + (v[i][r + 2] + v[i][r + 3]);
        } else {
#pragma unroll
            for (int r = 0; r < VEC; ++r) v[i][r] = 0.f;
        }
    }
    const float mu = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
#pragma unroll
            for (int r = 0; r < VEC; ++r) {
                const float d = v[i][r] - mu;
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
        const int c = lane + 64 * i;
        if (c < nv) {
            float o[VEC];
#pragma unroll
            for (int k = 0; k < VEC / 4; ++k) {
                const float4 g = *reinterpret_cast<const float4*>(gamma + c * VEC + 4 * k);
                const float4 b = *reinterpret_cast<const float4*>(beta + c * VEC + 4 * k);
                o[4 * k + 0] = (v[i][4 * k + 0] - mu) * rs * g.x + b.x;
                o[4 * k + 1] = (v[i][4 * k + 1] - mu) * rs * g.y + b.y;
                o[4 * k + 2] = (v[i][4 * k + 2] - mu) * rs * g.z + b.z;
                o[4 * k + 3] = (v[i][4 * k + 3] - mu) * rs * g.w + b.w;
            }
            *reinterpret_cast<raw_t*>(y + row * cols + c * VEC) = V::pack(o);
            if (q8) {
                unsigned int w[VEC / 4];
#pragma unroll
                for (int k = 0; k < VEC / 4; ++k) {
                    float u[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float yr = rnd<T>(o[4 * k + r]);
                        q8_max = fmaxf(q8_max, fabsf(yr));
                        u[r] = fminf(fmaxf(yr * q8_inv, -448.f), 448.f);
                    }
                    int ww = 0;
                    ww = __builtin_amdgcn_cvt_pk_fp8_f32(u[0], u[1], ww, false);
                    ww = __builtin_amdgcn_cvt_pk_fp8_f32(u[2], u[3], ww, true);
                    w[k] = (unsigned int)ww;
                }
                if constexpr (VEC == 8) *reinterpret_cast<uint2*>(q8 + row * cols + c * VEC) = make_uint2(w[0], w[1]);
                else *reinterpret_cast<unsigned int*>(q8 + row * cols + c * VEC) = w[0];
            }
        }
    }
    if (q8) {
        q8_max = wave_max(q8_max);
        if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(q8_amax + (blockIdx.x & 15) * 32), __float_as_uint(q8_max));
    }
}

// dz = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma ;  dz += dres_in
// dx_drop = dz * keepmask/(1-p)   (gradient reaching the dense output through the dropout)
// dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy: per-lane column partials -> per-wave LDS slices (8 waves at a time) ->
// one global f32 atomic per column per workgroup.  One 16-wave workgroup per CU: a 12800-row call issues 0.4 M global atomics on
// the 48 cache lines of dgamma/dbeta instead of the 1.5 M of a 4-wave / 1024-block layout, whose per-line serialisation at L2
// held the small shapes at 2 TB/s.  Sixteen waves per CU means 128 registers per wave: the row in flight is kept as it lies in
// memory (raw 16-B registers, converted twice) and gamma is re-read from L1 per row instead of living in registers, so that the
// 8-wide form of a 768- or 1024-column row (two chunks per lane: 16 + 16 column partials) stays clear of scratch.
// (the 8-wide form of a 768- / 1024-column row needs ~150 registers: it runs twelve waves per CU -- 168 registers -- whose one-row
// prefetch still keeps 12 x 3-4.5 KB per CU in flight)
// GREG: gamma lives in registers (the 4-wide forms: 12 values per lane for a 768-column row); otherwise it is staged once into LDS
// behind the reduction slices and read from there in both passes over a row (the 8-wide two-chunk form has no 16 registers to spare;
// re-reading it from L1 put eight dependent vector loads into every row's chain: 42 instead of 32 us per call inside the step)
template <typename T, int IT, int VEC, int LN_BWD_WAVES, bool GREG>
__global__ __launch_bounds__(LN_BWD_WAVES * 64) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const T* __restrict__ dres,
                                                     T* __restrict__ dz, T* __restrict__ dxdrop, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, long rows, int cols, float drop_p,
                                                     uint64_t seed, uint64_t offset) {
    typedef LnVec<T, VEC> V;
    typedef typename V::raw_t raw_t;
    extern __shared__ __attribute__((aligned(16))) float sh[];  // [8][cols]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = cols / VEC;
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    float ag[IT][VEC], ab[IT][VEC];
    float gmr[GREG ? IT : 1][VEC];
    float* shg = sh + 8 * cols;   // [cols] gamma (only when !GREG)
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int r = 0; r < VEC; ++r) ag[i][r] = ab[i][r] = 0.f;
        if (GREG) {
#pragma unroll
            for (int k = 0; k < VEC / 4; ++k) {
                const float4 g4 = c < nv ? *reinterpret_cast<const float4*>(gamma + c * VEC + 4 * k) : make_float4(0.f, 0.f, 0.f, 0.f);
                gmr[i][4 * k] = g4.x; gmr[i][4 * k + 1] = g4.y; gmr[i][4 * k + 2] = g4.z; gmr[i][4 * k + 3] = g4.w;
            }
        }
    }
    if (!GREG) {
        for (int c = threadIdx.x; c < cols; c += LN_BWD_WAVES * 64) shg[c] = gamma[c];
        __syncthreads();
    }
    auto gam = [&](int i, int c, int k, float (&o4)[4]) __attribute__((always_inline)) {   // gamma of columns c * VEC + 4 k .. + 3
        if (GREG) { o4[0] = gmr[GREG ? i : 0][4 * k]; o4[1] = gmr[GREG ? i : 0][4 * k + 1]; o4[2] = gmr[GREG ? i : 0][4 * k + 2]; o4[3] = gmr[GREG ? i : 0][4 * k + 3]; }
        else { const float4 g4 = *reinterpret_cast<const float4*>(shg + c * VEC + 4 * k); o4[0] = g4.x; o4[1] = g4.y; o4[2] = g4.z; o4[3] = g4.w; }
    };
    // rows are software-pipelined: the loads of a wave's NEXT row are in flight while the current row is reduced, normalised and
    // stored (a row is a dependent chain load -> two wave reductions -> store; without the prefetch a CU has no load outstanding
    // for about half of it)
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
                rdy[i] = *reinterpret_cast<const raw_t*>(dy + ((unsigned)(row * cols) + (unsigned)(c * VEC)));
                rz[i] = *reinterpret_cast<const raw_t*>(z + ((unsigned)(row * cols) + (unsigned)(c * VEC)));
            }
        }
    };
    // the residual gradient is only consumed at the end of a row: its prefetch for the next row is issued there (no second copy)
    auto fetch_q = [&](long row) {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) rq[i] = *reinterpret_cast<const raw_t*>(dres + ((unsigned)(row * cols) + (unsigned)(c * VEC)));
        }
    };
    long row = (long)blockIdx.x * LN_BWD_WAVES + wave;
    if (row < rows) { fetch(row); if (dres) fetch_q(row); }
    for (; row < rows; row += stride) {
        const float mu = nmu, rs = nrs;
        raw_t cdy[IT], cz[IT];   // the current row, as it lies in memory
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = lane + 64 * i;
            cdy[i] = rdy[i]; cz[i] = rz[i];
            if (c < nv) {
                float d[VEC], zz[VEC], gm[VEC];
                V::cvt(cdy[i], d);
                V::cvt(cz[i], zz);
#pragma unroll
                for (int k = 0; k < VEC / 4; ++k) {
                    float g4[4];
                    gam(i, c, k, g4);
                    gm[4 * k] = g4[0]; gm[4 * k + 1] = g4[1]; gm[4 * k + 2] = g4[2]; gm[4 * k + 3] = g4[3];
                }
#pragma unroll
                for (int r = 0; r < VEC; ++r) {
                    const float xh = (zz[r] - mu) * rs, g = d[r] * gm[r];
                    s1 += g;
                    s2 += g * xh;
                    ag[i][r] += d[r] * xh;
                    ab[i][r] += d[r];
                }
            }
        }
        if (row + stride < rows) fetch(row + stride);
        s1 = wave_sum(s1) / (float)cols;
        s2 = wave_sum(s2) / (float)cols;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float d[VEC], zz[VEC], o[VEC];
                V::cvt(cdy[i], d);
                V::cvt(cz[i], zz);
#pragma unroll
                for (int k = 0; k < VEC / 4; ++k) {
                    float gm[4];
                    gam(i, c, k, gm);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (zz[4 * k + r] - mu) * rs, g = d[4 * k + r] * gm[r];
                        o[4 * k + r] = rs * (g - s1 - xh * s2);
                    }
                }
                if (dres) {
                    float q[VEC];
                    V::cvt(rq[i], q);
#pragma unroll
                    for (int r = 0; r < VEC; ++r) o[r] += q[r];
                }
                *reinterpret_cast<raw_t*>(dz + ((unsigned)(row * cols) + (unsigned)(c * VEC))) = V::pack(o);
                if (dxdrop) {
                    if (drop_p > 0.f) {
                        float m[VEC];
                        ln_dropout<VEC>(seed, offset, row, nv, c, drop_p, inv_keep, m);
#pragma unroll
                        for (int r = 0; r < VEC; ++r) o[r] = rnd<T>(o[r]) * m[r];
                    }
                    *reinterpret_cast<raw_t*>(dxdrop + ((unsigned)(row * cols) + (unsigned)(c * VEC))) = V::pack(o);
                }
            }
        }
        if (dres && row + stride < rows) fetch_q(row + stride);
    }
    // workgroup reduction of the per-lane column partials, then one global atomic per column
    for (int pass = 0; pass < 2; ++pass) {
        float tot[2] = {0.f, 0.f};   // columns threadIdx.x and threadIdx.x + blockDim.x (cols <= 2 * blockDim.x, host-checked)
        for (int half = 0; half < (LN_BWD_WAVES + 7) / 8; ++half) {
            const int nsl = LN_BWD_WAVES - 8 * half < 8 ? LN_BWD_WAVES - 8 * half : 8;   // wave slices written in this half
            __syncthreads();
            if ((wave >> 3) == half) {
#pragma unroll
                for (int i = 0; i < IT; ++i) {
                    const int c = lane + 64 * i;
                    if (c < nv) {
#pragma unroll
                        for (int k = 0; k < VEC / 4; ++k)
                            *reinterpret_cast<float4*>(sh + (wave & 7) * cols + c * VEC + 4 * k) =
                                pass == 0 ? make_float4(ag[i][4 * k], ag[i][4 * k + 1], ag[i][4 * k + 2], ag[i][4 * k + 3])
                                          : make_float4(ab[i][4 * k], ab[i][4 * k + 1], ab[i][4 * k + 2], ab[i][4 * k + 3]);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = threadIdx.x + j * LN_BWD_WAVES * 64;
                if (c < cols) {
#pragma unroll
                    for (int w = 0; w < 8; ++w) if (w < nsl) tot[j] += sh[w * cols + c];
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
                         float* rstd, long rows, int cols, float eps, float p, uint64_t seed, uint64_t off, hipStream_t st,
                         void* q8 = nullptr, const float* q8_scale = nullptr, float* q8_amax = nullptr) {
    dim3 grid(ceil_div(rows, 4)), block(256);
    auto al16 = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
#define L(IT_, V_) hipLaunchKernelGGL((ln_fwd_kernel<T, IT_, V_>), grid, block, 0, st, (const T*)x, (const T*)res, (T*)z, gamma, beta, (T*)y, mean, rstd, rows, cols, eps, p, seed, off, (unsigned char*)q8, q8_scale, q8_amax)
    if constexpr (sizeof(T) == 2) {
        if (cols % 8 == 0 && al16(x) && al16(res) && al16(z) && al16(y)) {   // 16-B lane accesses
            const int it8 = ceil_div(cols / 8, 64);
            if (it8 <= 1) L(1, 8); else if (it8 <= 2) L(2, 8); else L(4, 8);
            return 0;
        }
    }
    const int it = ceil_div(cols / 4, 64);
    if (it <= 1) L(1, 4); else if (it <= 2) L(2, 4); else if (it <= 3) L(3, 4); else if (it <= 4) L(4, 4); else L(8, 4);
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
    // eight reduction slices (+ gamma for the !GREG form, which only serves <= 1024 columns): 8 * 2048 * 4 = 64 KB at the documented
    // maximum of 2048 columns, inside the default dynamic-LDS limit (the gamma slice used to be requested for every form,
    // 72 KB at 2048 columns without the opt-in)
    size_t shm = (size_t)8 * cols * sizeof(float);
    static const int variant = getenv("ECAMP_LN_BWD") ? atoi(getenv("ECAMP_LN_BWD")) : 0;   // development: 1 = 4-wide forms only
    auto al16 = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
#define L(IT_, V_, W_, G_)                                                                                               \
    do {                                                                                                                 \
        int nb = ceil_div(rows, W_);                                                                                     \
        if (nb > 256) nb = 256;                                                                                          \
        hipLaunchKernelGGL((ln_bwd_kernel<T, IT_, V_, W_, G_>), dim3(nb), dim3(W_ * 64), shm, st, (const T*)dy, (const T*)z, mean, rstd, gamma, \
                           (const T*)dres, (T*)dz, (T*)dxdrop, dgamma, dbeta, rows, cols, p, seed, off);      \
    } while (0)
    if constexpr (sizeof(T) == 2) {
        if (variant != 1 && cols % 8 == 0 && cols <= 1024 && al16(dy) && al16(z) && al16(dres) && al16(dz) && al16(dxdrop)) {   // 16-B lane accesses
            const int it8 = ceil_div(cols / 8, 64);
            // measured inside the step (profiles/r04_ln_variants.txt): one 16-B chunk per lane (<= 512 columns) beats two 8-B ones
            // (37.2 vs 38.3 us), the 1024-column row needs the two-chunk 16-B form (27.6 vs 62 us: the 4-wide form spills), and the
            // 768-column row is better off 4-wide with sixteen waves (32.4 vs 34.5 us)
            if (it8 <= 1) { L(1, 8, 16, true); return 0; }
            if (variant == 2 || cols > 768) { shm = (size_t)9 * cols * sizeof(float); L(2, 8, 12, false); return 0; }
        }
    }
    const int it = ceil_div(cols / 4, 64);
    if (it <= 1) L(1, 4, 16, true); else if (it <= 2) L(2, 4, 16, true); else if (it <= 3) L(3, 4, 16, true); else if (it <= 4) L(4, 4, 16, true); else L(8, 4, 16, true);
#undef L
    return 0;
}

// ecamp_layernorm_fwd that also leaves q8 [rows, cols] = e4m3(clamp(y / q8_scale[0], +-448)) and max|y| in the site's 16 amax slots
// (ecamp_quant_fp8_delayed's convention): the fp8 forward's quantisation folded into the producer of the GEMM input
extern "C" int ecamp_layernorm_fwd_q8(const void* x, const void* residual, void* z_out, const float* gamma, const float* beta,
                                      void* y, float* mean, float* rstd, int64_t rows, int32_t cols, float eps, float drop_p,
                                      uint64_t seed, uint64_t offset, void* q8, const float* q8_scale, float* q8_amax_slots,
                                      int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(x && gamma && beta && y && mean && rstd && q8 && q8_scale && q8_amax_slots, "layernorm_fwd_q8: null pointer");
    ECAMP_CHECK_ARG(cols % 4 == 0 && cols <= 2048 && rows > 0, "layernorm_fwd_q8: cols=%d must be a multiple of 4 and <= 2048", cols);
    ECAMP_CHECK_ARG(!(residual || drop_p > 0.f) || z_out, "layernorm_fwd_q8: fused residual/dropout needs z_out");
    ECAMP_CHECK_ARG(dtype == ECAMP_BF16 && (reinterpret_cast<uintptr_t>(q8) & 7) == 0, "layernorm_fwd_q8: bf16 rows, 8-byte aligned q8");
    ln_fwd_launch<bf16_t>(x, residual, z_out, gamma, beta, y, mean, rstd, rows, cols, eps, drop_p, seed, offset, stream, q8, q8_scale, q8_amax_slots);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

extern "C" int ecamp_layernorm_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                                   const void* dres_in, void* dz, void* dx_drop, float* dgamma, float* dbeta, int64_t rows,
                                   int32_t cols, float drop_p, uint64_t seed, uint64_t offset, int32_t dtype,
                                   hipStream_t stream) {
    ECAMP_CHECK_ARG(dy && z && mean && rstd && gamma && dz && dgamma && dbeta, "layernorm_bwd: null pointer");
    ECAMP_CHECK_ARG(cols % 4 == 0 && cols <= 2048 && rows > 0, "layernorm_bwd: cols=%d must be a multiple of 4 and <= 2048", cols);
    ECAMP_CHECK_ARG(rows * (int64_t)cols < (1ll << 30), "layernorm_bwd: rows * cols must stay below 2^30 (32-bit element offsets)");
    if (dtype == ECAMP_F32) ln_bwd_launch<float>(dy, z, mean, rstd, gamma, dres_in, dz, dx_drop, dgamma, dbeta, rows, cols, drop_p, seed, offset, stream);
    else if (dtype == ECAMP_BF16) ln_bwd_launch<bf16_t>(dy, z, mean, rstd, gamma, dres_in, dz, dx_drop, dgamma, dbeta, rows, cols, drop_p, seed, offset, stream);
    else return ecamp_set_error(-1, "layernorm_bwd: bad dtype %d", dtype);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
