// Image-side HBM-bound kernels of the ECAMP hot path:
//   bicubic 2x down-resize, MAE masking indices, im2col+gather of visible patches, token assembly,
//   decoder un-shuffle with mask-token fill, unpatchify + masked MSE, the super-resolution head
//   (bilinear x2 -> conv3x3 -> ReLU -> conv3x3 -> +skip -> ReLU) with its windowed MSE, and their backward.
// Masks are never materialised as pixel tensors (the reference kron()s them, model_ecamp.py:196-215):
// they are evaluated from mask[b, y/16, x/16] and the window bounds on the fly.
#include "common.h"
#include <stdlib.h>
#include <stdint.h>

// ---------------------------------------------------------------------------------------------
// K1  bicubic resize (aten upsample_bicubic2d semantics: A=-0.75, align_corners=False, no antialias)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void cubic_coeffs(float t, float (&w)[4]) {
    const float A = -0.75f;
    float x;
    x = t + 1.0f; w[0] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
    x = t;        w[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 1.0f - t; w[2] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 2.0f - t; w[3] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
}
__global__ void bicubic_kernel(const float* __restrict__ src, float* __restrict__ dst, long planes, int Hs, int Ws, int Hd,
                               int Wd, float sy, float sx) {
    long n = planes * Hd * Wd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int x = (int)(i % Wd), y = (int)((i / Wd) % Hd);
        long pl = i / ((long)Wd * Hd);
        float ry = sy * (y + 0.5f) - 0.5f, rx = sx * (x + 0.5f) - 0.5f;
        float fy = floorf(ry), fx = floorf(rx);
        float wy[4], wx[4];
        cubic_coeffs(ry - fy, wy);
        cubic_coeffs(rx - fx, wx);
        int iy = (int)fy, ix = (int)fx;
        const float* p = src + pl * (long)Hs * Ws;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            int yy = min(max(iy - 1 + a, 0), Hs - 1);
            float row = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                int xx = min(max(ix - 1 + b, 0), Ws - 1);
                row += wx[b] * p[(long)yy * Ws + xx];
            }
            acc += wy[a] * row;
        }
        dst[i] = acc;
    }
}
// Exact 2x down-resize (the only ratio on the hot path: 448 -> 224, model_ecamp.py:318): every output pixel has the same tap phase
// t = 0.5, rows 2y-1..2y+2, columns 2x-1..2x+2 (clamped).  A thread produces 4 consecutive outputs from 4 x 4 aligned float4 loads
// (the generic kernel issues 64 scalar loads for the same work) in the generic kernel's order of operations -> identical bits.
__global__ __launch_bounds__(256) void bicubic_half_kernel(const float* __restrict__ src, float* __restrict__ dst, long planes, int Hd, int Wd) {
    const int Ws = 2 * Wd, Hs = 2 * Hd, W4 = Wd >> 2;
    const long n = planes * Hd * W4;
    float w[4];
    cubic_coeffs(0.5f, w);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int tx = (int)(i % W4), y = (int)((i / W4) % Hd);
        const long pl = i / ((long)W4 * Hd);
        const float* p = src + pl * (long)Hs * Ws;
        const int c0 = 8 * tx;                                  // source column of output 4*tx is 2*(4*tx) = c0; taps start at c0-1
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(2 * y - 1 + a, 0), Hs - 1);
            const float* r = p + (long)yy * Ws;
            const float4 v1 = *reinterpret_cast<const float4*>(r + c0);
            const float4 v2 = *reinterpret_cast<const float4*>(r + c0 + 4);
            const float left = c0 > 0 ? r[c0 - 1] : r[0];                          // column clamp at the left border
            const float4 v3 = c0 + 8 < Ws ? *reinterpret_cast<const float4*>(r + c0 + 8) : make_float4(r[Ws - 1], r[Ws - 1], 0.f, 0.f);
            const float v[11] = {left, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y};   // columns c0-1 .. c0+9
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float row = 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b) row += w[b] * v[2 * o + b];
                acc[o] += w[a] * row;
            }
        }
        *reinterpret_cast<float4*>(dst + (pl * Hd + y) * (long)Wd + 4 * tx) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}
// ---- uint8 single-channel images (the dataset's item is a grayscale crop replicated to three channels and normalised
// with ONE mean / std, pretrain_datasets.py:47-52): the f32 [B,3,H,W] schema carries 12 bytes per pixel over PCIe and through every
// read, the crop itself 1.  The kernels normalise through the caller's 256-entry table lut[u] = ((float)u / 255 - mean) / std -- ToTensor +
// Normalize evaluated once per byte value in their f32 arithmetic -- so they see bit for bit the values the f32 schema would hold.
// the SR loss target: f32 [B,3,2R,2R], or u8 [B,2R,2R] normalised on the fly (every channel reads the same byte)
struct BigSrc {
    const float* f;
    const unsigned char* u8;
    const float* lut;
    __device__ __forceinline__ bool any() const { return f != nullptr || u8 != nullptr; }
    __device__ __forceinline__ float at(long b, int o, int Y, int X, int R2) const {
        return u8 ? lut[u8[(b * R2 + Y) * (long)R2 + X]] : f[((b * 3 + o) * (long)R2 + Y) * R2 + X];
    }
    __device__ __forceinline__ float2 at2(long b, int o, int Y, int X, int R2) const {   // pixels X, X + 1 (X even)
        if (u8) {
            const unsigned short w = *reinterpret_cast<const unsigned short*>(u8 + (b * R2 + Y) * (long)R2 + X);
            return make_float2(lut[w & 255u], lut[w >> 8]);
        }
        return *reinterpret_cast<const float2*>(f + ((b * 3 + o) * (long)R2 + Y) * R2 + X);
    }
};
// exact 2x down-resize of a u8 image into the three identical f32 planes of the model's `imgs`: the arithmetic of bicubic_half_kernel
// on the normalised pixels (identical bits), one twelfth of its source bytes
__global__ __launch_bounds__(256) void bicubic_half_u8_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, long B, int Hd, int Wd,
                                                              const float* __restrict__ lut_g) {
    const int Ws = 2 * Wd, Hs = 2 * Hd, W4 = Wd >> 2;
    const long n = B * Hd * W4;
    __shared__ float lut[256];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    __syncthreads();
    float w[4];
    cubic_coeffs(0.5f, w);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int tx = (int)(i % W4), y = (int)((i / W4) % Hd);
        const long b = i / ((long)W4 * Hd);
        const unsigned char* p = src + b * (long)Hs * Ws;
        const int c0 = 8 * tx;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(2 * y - 1 + a, 0), Hs - 1);
            const unsigned char* r = p + (long)yy * Ws;
            const uint2 m = *reinterpret_cast<const uint2*>(r + c0);            // columns c0 .. c0+7 (8-byte aligned: Ws % 8 == 0)
            const unsigned left = c0 > 0 ? r[c0 - 1] : r[0];
            const unsigned r0 = c0 + 8 < Ws ? r[c0 + 8] : r[Ws - 1], r1 = c0 + 8 < Ws ? r[c0 + 9] : r[Ws - 1];
            const unsigned u[11] = {left, m.x & 255u, (m.x >> 8) & 255u, (m.x >> 16) & 255u, m.x >> 24, m.y & 255u, (m.y >> 8) & 255u, (m.y >> 16) & 255u,
                                    m.y >> 24, r0, r1};
            float v[11];
#pragma unroll
            for (int k = 0; k < 11; ++k) v[k] = lut[u[k]];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float row = 0.f;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) row += w[bb] * v[2 * o + bb];
                acc[o] += w[a] * row;
            }
        }
        const float4 out = make_float4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
        for (int c = 0; c < 3; ++c) *reinterpret_cast<float4*>(dst + ((b * 3 + c) * Hd + y) * (long)Wd + 4 * tx) = out;
    }
}
// any other ratio (or no resize at all: Hs == Hd copies the normalised image into the three planes)
__global__ void bicubic_u8_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, long B, int Hs, int Ws, int Hd, int Wd, float sy, float sx,
                                  const float* __restrict__ lut) {
    const long n = B * Hd * Wd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wd), y = (int)((i / Wd) % Hd);
        const long b = i / ((long)Wd * Hd);
        const unsigned char* p = src + b * (long)Hs * Ws;
        float acc;
        if (Hs == Hd && Ws == Wd) {
            acc = lut[p[(long)y * Ws + x]];
        } else {
            const float ry = sy * (y + 0.5f) - 0.5f, rx = sx * (x + 0.5f) - 0.5f;
            const float fy = floorf(ry), fx = floorf(rx);
            float wy[4], wx[4];
            cubic_coeffs(ry - fy, wy);
            cubic_coeffs(rx - fx, wx);
            const int iy = (int)fy, ix = (int)fx;
            acc = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int yy = min(max(iy - 1 + a, 0), Hs - 1);
                float row = 0.f;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int xx = min(max(ix - 1 + bb, 0), Ws - 1);
                    row += wx[bb] * lut[p[(long)yy * Ws + xx]];
                }
                acc += wy[a] * row;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[((b * 3 + c) * Hd + y) * (long)Wd + x] = acc;
    }
}
extern "C" int ecamp_bicubic_resize_u8(const unsigned char* src, float* dst, int64_t B, int32_t Hs, int32_t Ws, int32_t Hd, int32_t Wd,
                                       const float* lut, hipStream_t stream) {
    ECAMP_CHECK_ARG(src && dst && lut && B > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "bicubic_u8: bad args");
    if (Hs == 2 * Hd && Ws == 2 * Wd && (Wd & 3) == 0 && ((uintptr_t)src & 7) == 0 && ((uintptr_t)dst & 15) == 0) {
        long n4 = B * Hd * (Wd >> 2);
        int nb4 = (int)((n4 + 255) / 256);
        if (nb4 > 16384) nb4 = 16384;
        hipLaunchKernelGGL(bicubic_half_u8_kernel, dim3(nb4), dim3(256), 0, stream, src, dst, (long)B, Hd, Wd, lut);
        ECAMP_LAUNCH_CHECK();
        return 0;
    }
    long n = B * Hd * Wd;
    int nb = (int)((n + 255) / 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(bicubic_u8_kernel, dim3(nb), dim3(256), 0, stream, src, dst, (long)B, Hs, Ws, Hd, Wd, (float)Hs / (float)Hd, (float)Ws / (float)Wd, lut);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
extern "C" int ecamp_bicubic_resize(const float* src, float* dst, int64_t planes, int32_t Hs, int32_t Ws, int32_t Hd,
                                    int32_t Wd, hipStream_t stream) {
    ECAMP_CHECK_ARG(src && dst && planes > 0, "bicubic: bad args");
    if (Hs == 2 * Hd && Ws == 2 * Wd && (Wd & 3) == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0) {
        long n4 = planes * Hd * (Wd >> 2);
        int nb4 = (int)((n4 + 255) / 256);
        if (nb4 > 16384) nb4 = 16384;
        hipLaunchKernelGGL(bicubic_half_kernel, dim3(nb4), dim3(256), 0, stream, src, dst, (long)planes, Hd, Wd);
        ECAMP_LAUNCH_CHECK();
        return 0;
    }
    long n = planes * Hd * Wd;
    int nb = (int)((n + 255) / 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(bicubic_kernel, dim3(nb), dim3(256), 0, stream, src, dst, (long)planes, Hs, Ws, Hd, Wd,
                       (float)Hs / (float)Hd, (float)Ws / (float)Wd);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// K3  MAE masking indices: stable rank-by-count of the per-sample noise row (== argsort twice)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_indices_kernel(const float* __restrict__ noise, int L, int len_keep,
                                                           int* __restrict__ ids_restore, int* __restrict__ ids_keep,
                                                           float* __restrict__ mask) {
    extern __shared__ float nz[];
    const long b = blockIdx.x;
    for (int i = threadIdx.x; i < L; i += 256) nz[i] = noise[b * L + i];
    __syncthreads();
    for (int i = threadIdx.x; i < L; i += 256) {
        float v = nz[i];
        int rank = 0;
        for (int j = 0; j < L; ++j) {
            float u = nz[j];
            rank += (u < v || (u == v && j < i)) ? 1 : 0;
        }
        ids_restore[b * L + i] = rank;
        mask[b * L + i] = rank >= len_keep ? 1.0f : 0.0f;
        if (rank < len_keep) ids_keep[b * len_keep + rank] = i;
    }
}
extern "C" int ecamp_mask_indices(const float* noise, int64_t B, int32_t L, int32_t len_keep, int32_t* ids_restore,
                                  int32_t* ids_keep, float* mask, hipStream_t stream) {
    ECAMP_CHECK_ARG(noise && ids_restore && ids_keep && mask && L > 0 && L <= 8192 && len_keep >= 0 && len_keep <= L,
                    "mask_indices: bad args");
    hipLaunchKernelGGL(mask_indices_kernel, dim3((unsigned)B), dim3(256), (size_t)L * 4, stream, noise, L, len_keep, ids_restore,
                       ids_keep, mask);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// K2a  im2col of the VISIBLE patches only (gather-before-embed): out[b*(Lk+1)+t, c*p*p+py*p+px]
//      row t=0 (cls slot) is zero so the same buffer drives the weight-gradient GEMM unchanged
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void im2col_gather_kernel(const float* __restrict__ imgs, const int* __restrict__ ids_keep, T* __restrict__ out,
                                     long B, int Lk, int C, int R, int p) {
    const int G = R / p, K = C * p * p, K4 = K / 4, Tt = Lk + 1;
    long n = B * Tt * K4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int k4 = (int)(i % K4);
        long row = i / K4;
        int t = (int)(row % Tt);
        long b = row / Tt;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            int patch = ids_keep[b * Lk + t - 1];
            int gy = patch / G, gx = patch % G;
            int k = k4 * 4;
            int c = k / (p * p), py = (k / p) % p, px = k % p;
            ld4<float>(imgs + ((b * C + c) * (long)R + gy * p + py) * R + gx * p + px, v);
        }
        st4<T>(out + row * K + k4 * 4, v);
    }
}
extern "C" int ecamp_im2col_gather(const float* imgs, const int32_t* ids_keep, void* out, int64_t B, int32_t Lk, int32_t C,
                                   int32_t R, int32_t p, int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(imgs && ids_keep && out && p % 4 == 0 && R % p == 0, "im2col_gather: bad args");
    long n = B * (Lk + 1) * (long)(C * p * p / 4);
    int nb = (int)((n + 255) / 256);
    if (nb > 8192) nb = 8192;
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(im2col_gather_kernel<float>, dim3(nb), dim3(256), 0, stream, imgs, ids_keep, (float*)out, (long)B, Lk, C, R, p);
    else hipLaunchKernelGGL(im2col_gather_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, imgs, ids_keep, (bf16_t*)out, (long)B, Lk, C, R, p);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// K2b/K4  in place on the patch-embed GEMM output x[B, Lk+1, D]:
//      x[b,0,:] = cls + pos[0] ;  x[b,t,:] += pos[1 + ids_keep[b,t-1]]     (model_ecamp.py:222,228-230)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void assemble_tokens_kernel(T* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos,
                                       const int* __restrict__ ids_keep, long B, int Lk, int D4) {
    const int Tt = Lk + 1;
    long n = B * Tt * D4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int d = (int)(i % D4);
        long row = i / D4;
        int t = (int)(row % Tt);
        long b = row / Tt;
        float v[4], q[4];
        if (t == 0) {
            ld4<float>(cls + d * 4, v);
            ld4<float>(pos + d * 4, q);
        } else {
            ld4<T>(x + i * 4, v);
            ld4<float>(pos + (long)(1 + ids_keep[b * Lk + t - 1]) * D4 * 4 + d * 4, q);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += q[r];
        st4<T>(x + i * 4, v);
    }
}
extern "C" int ecamp_assemble_tokens(void* x, const float* cls, const float* pos, const int32_t* ids_keep, int64_t B, int32_t Lk,
                                     int32_t D, int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(x && cls && pos && ids_keep && D % 4 == 0, "assemble_tokens: bad args");
    long n = B * (Lk + 1) * (long)(D / 4);
    int nb = (int)((n + 255) / 256);
    if (nb > 8192) nb = 8192;
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(assemble_tokens_kernel<float>, dim3(nb), dim3(256), 0, stream, (float*)x, cls, pos, ids_keep, (long)B, Lk, D / 4);
    else hipLaunchKernelGGL(assemble_tokens_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, (bf16_t*)x, cls, pos, ids_keep, (long)B, Lk, D / 4);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// K10  decoder un-shuffle (model_ecamp.py:245-251)
//   fwd: xd[b,0] = y[b,0] + dpos[0] ; xd[b,1+j] = (r=ids_restore[b,j]) < Lk ? y[b,1+r] : mask_token) + dpos[1+j]
//   bwd: dy[b,0] = dxd[b,0] ; dy[b,1+r] = dxd[b, 1+ids_keep[b,r]] ; dmask_token += sum over masked slots
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void unshuffle_fwd_kernel(const T* __restrict__ y, const int* __restrict__ ids_restore, const float* __restrict__ mtok,
                                     const float* __restrict__ dpos, T* __restrict__ xd, long B, int L, int Lk, int D4) {
    long n = B * (L + 1) * D4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int d = (int)(i % D4);
        long row = i / D4;
        int t = (int)(row % (L + 1));
        long b = row / (L + 1);
        float v[4], q[4];
        int src = 0;
        if (t > 0) {
            int r = ids_restore[b * L + t - 1];
            src = r < Lk ? 1 + r : -1;
        }
        if (src >= 0) ld4<T>(y + ((b * (Lk + 1) + src) * (long)D4 + d) * 4, v);
        else ld4<float>(mtok + d * 4, v);
        ld4<float>(dpos + ((long)t * D4 + d) * 4, q);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += q[r];
        st4<T>(xd + i * 4, v);
    }
}
template <typename T>
__global__ void unshuffle_bwd_kernel(const T* __restrict__ dxd, const int* __restrict__ ids_keep, T* __restrict__ dy, long B,
                                     int L, int Lk, int D4) {
    long n = B * (Lk + 1) * D4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int d = (int)(i % D4);
        long row = i / D4;
        int t = (int)(row % (Lk + 1));
        long b = row / (Lk + 1);
        int src = t == 0 ? 0 : 1 + ids_keep[b * Lk + t - 1];
        float v[4];
        ld4<T>(dxd + ((b * (L + 1) + src) * (long)D4 + d) * 4, v);
        st4<T>(dy + i * 4, v);
    }
}
// dmask_token[c] += sum_{b,j : ids_restore[b,j] >= Lk} dxd[b,1+j,c]
template <typename T>
__global__ __launch_bounds__(256) void masktok_grad_kernel(const T* __restrict__ dxd, const int* __restrict__ ids_restore,
                                                           float* __restrict__ out, long B, int L, int Lk, int D) {
    __shared__ float sh[8][129];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 128 + tx * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const long M = B * L;
    if (c0 < D) {
        for (long m = (long)blockIdx.y * 8 + ty; m < M; m += (long)gridDim.y * 8) {
            if (ids_restore[m] < Lk) continue;
            long b = m / L;
            int j = (int)(m % L);
            float p[4];
            ld4<T>(dxd + (b * (L + 1) + 1 + j) * (long)D + c0, p);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += p[r];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sh[ty][tx * 4 + r] = acc[r];
    __syncthreads();
    if (threadIdx.x < 128) {
        int c = blockIdx.x * 128 + threadIdx.x;
        if (c < D) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += sh[j][threadIdx.x];
            atomicAdd(out + c, t);
        }
    }
}
extern "C" int ecamp_unshuffle_fwd(const void* y, const int32_t* ids_restore, const float* mask_token, const float* dpos, void* xd,
                                   int64_t B, int32_t L, int32_t Lk, int32_t D, int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(y && ids_restore && mask_token && dpos && xd && D % 4 == 0, "unshuffle_fwd: bad args");
    long n = B * (L + 1) * (long)(D / 4);
    int nb = (int)((n + 255) / 256);
    if (nb > 8192) nb = 8192;
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(unshuffle_fwd_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)y, ids_restore, mask_token, dpos, (float*)xd, (long)B, L, Lk, D / 4);
    else hipLaunchKernelGGL(unshuffle_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)y, ids_restore, mask_token, dpos, (bf16_t*)xd, (long)B, L, Lk, D / 4);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
extern "C" int ecamp_unshuffle_bwd(const void* dxd, const int32_t* ids_restore, const int32_t* ids_keep, void* dy,
                                   float* dmask_token, int64_t B, int32_t L, int32_t Lk, int32_t D, int32_t dtype,
                                   hipStream_t stream) {
    ECAMP_CHECK_ARG(dxd && ids_restore && ids_keep && dy && dmask_token && D % 4 == 0, "unshuffle_bwd: bad args");
    long n = B * (Lk + 1) * (long)(D / 4);
    int nb = (int)((n + 255) / 256);
    if (nb > 8192) nb = 8192;
    int nbx = ceil_div(D, 128), nby = ceil_div(B * L, 8 * 16);
    int cap = 1024 / nbx;
    if (nby > cap) nby = cap;
    if (nby < 1) nby = 1;
    if (dtype == ECAMP_F32) {
        hipLaunchKernelGGL(unshuffle_bwd_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)dxd, ids_keep, (float*)dy, (long)B, L, Lk, D / 4);
        hipLaunchKernelGGL(masktok_grad_kernel<float>, dim3(nbx, nby), dim3(256), 0, stream, (const float*)dxd, ids_restore, dmask_token, (long)B, L, Lk, D);
    } else {
        hipLaunchKernelGGL(unshuffle_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)dxd, ids_keep, (bf16_t*)dy, (long)B, L, Lk, D / 4);
        hipLaunchKernelGGL(masktok_grad_kernel<bf16_t>, dim3(nbx, nby), dim3(256), 0, stream, (const bf16_t*)dxd, ids_restore, dmask_token, (long)B, L, Lk, D);
    }
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// K11/K13  unpatchify (model_ecamp.py:153-165) fused with the masked MSE numerator (:288-298)
//   pred[b, 1 + gy*G+gx, (py*p+px)*3 + c]  ->  pred_img[b, c, gy*p+py, gx*p+px]  (f32)
//   loss_sum[0] += sum mask[b,patch] * (pred_img - imgs)^2
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void unpatchify_mim_kernel(const T* __restrict__ pred, const float* __restrict__ imgs,
                                                             const float* __restrict__ mask, float* __restrict__ pred_img,
                                                             float* __restrict__ loss_sum, long B, int R, int p) {
    __shared__ float sh[4];
    const int G = R / p, L = G * G, PD = p * p * 3;
    long n = B * (long)R * R;
    float part = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        // i enumerates (b, patch, py, px) so that consecutive lanes read consecutive pred elements
        int px = (int)(i % p), py = (int)((i / p) % p);
        int patch = (int)((i / (p * p)) % L);
        long b = i / ((long)p * p * L);
        int gy = patch / G, gx = patch % G;
        const T* src = pred + ((b * (L + 1) + 1 + patch) * (long)PD) + (py * p + px) * 3;
        float m = mask[b * L + patch];
        int y = gy * p + py, x = gx * p + px;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = to_f<T>(src[c]);
            long o = ((b * 3 + c) * (long)R + y) * R + x;
            pred_img[o] = v;
            float d = v - imgs[o];
            part += m * d * d;
        }
    }
    part = block_sum_256(part, sh);
    if (threadIdx.x == 0) atomicAdd(loss_sum, part);
}
extern "C" int ecamp_unpatchify_mim(const void* pred, const float* imgs, const float* mask, float* pred_img, float* loss_sum,
                                    int64_t B, int32_t R, int32_t p, int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(pred && imgs && mask && pred_img && loss_sum, "unpatchify_mim: bad args");
    long n = B * (long)R * R;
    int nb = (int)((n + 255) / 256);
    if (nb > 4096) nb = 4096;
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(unpatchify_mim_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)pred, imgs, mask, pred_img, loss_sum, (long)B, R, p);
    else hipLaunchKernelGGL(unpatchify_mim_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)pred, imgs, mask, pred_img, loss_sum, (long)B, R, p);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// d_pred[b, t, :] (T, incl. a zero cls row) = gm * mask * (pred_img - imgs) + gs * dsr    (patchify of the image-space gradient)
//   gm_gs[0] = g_mim * 2 / N_mim , gm_gs[1] = g_res * 2 / N_res   (device scalars, so no host sync)
template <typename T>
__global__ void img_loss_bwd_kernel(const float* __restrict__ pred_img, const float* __restrict__ imgs,
                                    const float* __restrict__ mask, const float* __restrict__ dsr,
                                    const float* __restrict__ gm_gs, T* __restrict__ dpred, long B, int R, int p) {
    const int G = R / p, L = G * G, PD = p * p * 3;
    const float gm = gm_gs[0], gs = gm_gs[1];
    long n = B * (long)(L + 1) * p * p;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int px = (int)(i % p), py = (int)((i / p) % p);
        int t = (int)((i / (p * p)) % (L + 1));
        long b = i / ((long)p * p * (L + 1));
        T* dst = dpred + ((b * (L + 1) + t) * (long)PD) + (py * p + px) * 3;
        if (t == 0) {
            dst[0] = dst[1] = dst[2] = from_f<T>(0.f);
            continue;
        }
        int patch = t - 1, gy = patch / G, gx = patch % G;
        float m = mask[b * L + patch];
        int y = gy * p + py, x = gx * p + px;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            long o = ((b * 3 + c) * (long)R + y) * R + x;
            dst[c] = from_f<T>(gm * m * (pred_img[o] - imgs[o]) + gs * dsr[o]);
        }
    }
}
extern "C" int ecamp_img_loss_bwd(const float* pred_img, const float* imgs, const float* mask, const float* dsr,
                                  const float* gm_gs, void* dpred, int64_t B, int32_t R, int32_t p, int32_t dtype,
                                  hipStream_t stream) {
    ECAMP_CHECK_ARG(pred_img && imgs && mask && dsr && gm_gs && dpred, "img_loss_bwd: bad args");
    long n = B * (long)((R / p) * (R / p) + 1) * p * p;
    int nb = (int)((n + 255) / 256);
    if (nb > 4096) nb = 4096;
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(img_loss_bwd_kernel<float>, dim3(nb), dim3(256), 0, stream, pred_img, imgs, mask, dsr, gm_gs, (float*)dpred, (long)B, R, p);
    else hipLaunchKernelGGL(img_loss_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, pred_img, imgs, mask, dsr, gm_gs, (bf16_t*)dpred, (long)B, R, p);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// K12  super-resolution head (model_ecamp.py:28-46) + windowed MSE (:205-212,291-299)
//   u  = bilinear_x2(pred_img)                 (align_corners=False)
//   c1 = relu(conv1(u)) ; s = relu(conv2(c1) + u)
//   res_sum += sum_{window} (s - big)^2 ; ds = [window] * (s - big) * [s > 0]
//   window: super-patch rows [col_b, col_b+W) x cols [row_b, row_b+W) of 2p pixels (column indexes the H axis)
// Intermediates u, c1, ds, dc1, du live in HBM in the compute dtype; weights (168 floats) in constant-like args.
// ---------------------------------------------------------------------------------------------
struct SrW {
    float w1[81], b1[3], w2[81], b2[3];
};
struct SrP {  // device pointers to super_res.conv{1,2}.{weight,bias} (f32 master parameters)
    const float* w1; const float* b1; const float* w2; const float* b2;
};
__device__ __forceinline__ void load_srw(SrW& W, const SrP& p) {
    for (int i = threadIdx.x; i < 81; i += blockDim.x) {
        W.w1[i] = p.w1[i];
        W.w2[i] = p.w2[i];
    }
    if (threadIdx.x < 3) {
        W.b1[threadIdx.x] = p.b1[threadIdx.x];
        W.b2[threadIdx.x] = p.b2[threadIdx.x];
    }
    __syncthreads();
}

__device__ __forceinline__ void up2_taps(int Y, int H, int& y0, int& y1, float& w0, float& w1) {
    float src = fmaxf((Y + 0.5f) * 0.5f - 0.5f, 0.f);
    y0 = (int)src;
    y1 = min(y0 + 1, H - 1);
    w1 = src - (float)y0;
    w0 = 1.0f - w1;
}

// ---- fused, LDS-resident SR head -------------------------------------------------------------------------
// One workgroup walks 32x32 high-resolution tiles (= one super-patch each, so a tile is wholly inside or outside the
// loss window).  u, c1 (and in backward ds, dc1, du) live only in LDS: the forward reads pred_img + big once and
// writes one float; the backward re-derives everything from pred_img (cheap ALU) and writes only d/d pred_img.
// Zero padding of both convs is applied at the IMAGE border (values outside the image are 0, not conv outputs).
#define SRT 32

__device__ __forceinline__ float bilinear_at(const float* __restrict__ pl, int Y, int X, int R) {
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    up2_taps(Y, R, y0, y1, wy0, wy1);
    up2_taps(X, R, x0, x1, wx0, wx1);
    return wy0 * (wx0 * pl[y0 * R + x0] + wx1 * pl[y0 * R + x1]) + wy1 * (wx0 * pl[y1 * R + x0] + wx1 * pl[y1 * R + x1]);
}
// U[c][y][x] for the tile at (Y0, X0) extended by HALO; 0 outside the image
template <int HALO>
__device__ __forceinline__ void sr_fill_u(float* U, const float* __restrict__ pred_img, long b, int Y0, int X0, int R) {
    constexpr int E = SRT + 2 * HALO;
    const int R2 = 2 * R;
    for (int idx = threadIdx.x; idx < 3 * E * E; idx += 256) {
        int c = idx / (E * E), y = (idx / E) % E, x = idx % E;
        int Y = Y0 - HALO + y, X = X0 - HALO + x;
        U[idx] = (Y < 0 || Y >= R2 || X < 0 || X >= R2) ? 0.f : bilinear_at(pred_img + (b * 3 + c) * (long)R * R, Y, X, R);
    }
}
// dst (edge ED, halo HD) = relu?(bias + conv3x3(src (edge ES = ED+2))) ; 0 outside the image
template <int ED, int HD, bool RELU>
__device__ __forceinline__ void sr_conv_stage(float* dst, const float* src, const float* w, const float* bias, int Y0, int X0, int R2) {
    constexpr int ES = ED + 2;
    for (int idx = threadIdx.x; idx < ED * ED; idx += 256) {
        int y = idx / ED, x = idx % ED;
        int Y = Y0 - HD + y, X = X0 - HD + x;
        float acc[3] = {bias[0], bias[1], bias[2]};
        const bool in = Y >= 0 && Y < R2 && X >= 0 && X < R2;
        if (in) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float v = src[(i * ES + y + ky) * ES + x + kx];
#pragma unroll
                        for (int o = 0; o < 3; ++o) acc[o] += w[((o * 3 + i) * 3 + ky) * 3 + kx] * v;
                    }
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) dst[(o * ED + y) * ED + x] = in ? (RELU ? fmaxf(acc[o], 0.f) : acc[o]) : 0.f;
    }
}

// `sr_out` (optional, f32 [B,3,2R,2R]): the head's output image -- the reference's `self.super_res(pred_img)` (model_ecamp.py:28-46,
// 285) -- for every tile, not only the loss window (parity checks / visualisation; the training step passes null and touches only
// the window).  With `big` null only the image is produced.
__global__ __launch_bounds__(256) void sr_fused_fwd_kernel(const float* __restrict__ pred_img, BigSrc big,
                                                           const long* __restrict__ column, const long* __restrict__ row, SrP P,
                                                           float* __restrict__ loss_sum, long B, int R, int win, float* __restrict__ sr_out) {
    __shared__ SrW W;
    __shared__ float U[3 * 36 * 36];
    __shared__ float C1[3 * 34 * 34];
    __shared__ float sh[4];
    load_srw(W, P);
    const int R2 = 2 * R, G = R2 / SRT;
    float part = 0.f;
    for (long t = blockIdx.x; t < B * G * G; t += gridDim.x) {
        const long b = t / (G * G);
        const int ty = (int)((t / G) % G), tx = (int)(t % G);
        bool in_win = false;
        if (big.any()) {
            const int c0 = (int)column[b], r0 = (int)row[b];
            in_win = !(ty < c0 || ty >= c0 + win || tx < r0 || tx >= r0 + win);
        }
        if (!in_win && !sr_out) continue;  // block-uniform
        const int Y0 = ty * SRT, X0 = tx * SRT;
        __syncthreads();
        sr_fill_u<2>(U, pred_img, b, Y0, X0, R);
        __syncthreads();
        sr_conv_stage<34, 1, true>(C1, U, W.w1, W.b1, Y0, X0, R2);
        __syncthreads();
        for (int idx = threadIdx.x; idx < SRT * SRT; idx += 256) {
            int y = idx / SRT, x = idx % SRT;
            float acc[3] = {W.b2[0], W.b2[1], W.b2[2]};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float v = C1[(i * 34 + y + ky) * 34 + x + kx];
#pragma unroll
                        for (int o = 0; o < 3; ++o) acc[o] += W.w2[((o * 3 + i) * 3 + ky) * 3 + kx] * v;
                    }
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                float s = fmaxf(acc[o] + U[(o * 36 + y + 2) * 36 + x + 2], 0.f);
                const long at = ((b * 3 + o) * (long)R2 + Y0 + y) * R2 + X0 + x;
                if (sr_out) sr_out[at] = s;
                if (in_win) {
                    float d = s - big.at(b, o, Y0 + y, X0 + x, R2);
                    part += d * d;
                }
            }
        }
    }
    part = block_sum_256(part, sh);
    if (threadIdx.x == 0 && loss_sum) atomicAdd(loss_sum, part);
}

// ---- bf16 matrix-core variant of the SR head (compute_dtype = bf16) ---------------------------------------------------------
// A 3 -> 3 channel 3x3 convolution is nine 4x4 (out-channel x in-channel, padded from 3) matrix products per pixel.
// v_mfma_f32_4x4x4_16B_bf16 does sixteen independent 4x4x4 products per wave: block = lane/4, the lane (block, i) supplies row i of
// A, column i of B and receives column i of D (probed on hardware: tools/probes/mfma4_probe.hip).  With A = the tap's weight matrix
// (held in registers, the same in every block), B column = the 4 input channels of ONE pixel and D column = that pixel's 4 output
// channels, a wave convolves 64 pixels with 9 MFMAs and 9 eight-byte LDS reads per lane -- the f32 VALU form above needs 81 FMAs
// and 108 LDS reads per pixel and is LDS-issue-bound.  Tiles are kept channel-interleaved in LDS: [y][x][4] bf16 = 8 B per pixel.
// f32 accumulation; u, c1 (and ds, dc1 in backward) are rounded to bf16, the skip connection and the loss stay f32.
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#define MFMA4(A, B, C) ECAMP_MFMA_4x4x4((A), (B), (C))

__device__ __forceinline__ bf16x4_t sr_pack4(float a, float b, float c) {
    uint2 u = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, 0.f));
    return __builtin_bit_cast(bf16x4_t, u);
}
// A operand of tap (ky,kx): row i = lane&3 of M[i][k];  TRANS = false: M = W[oc=i][ic=k] (forward conv), true: M = W[oc=k][ic=i]
template <bool TRANS>
__device__ __forceinline__ void sr_load_taps(const float* __restrict__ w, int lane, bf16x4_t (&a)[9]) {
    const int i = lane & 3;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float v[3] = {0.f, 0.f, 0.f};
        if (i < 3) {
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = TRANS ? w[(k * 3 + i) * 9 + t] : w[(i * 3 + k) * 9 + t];
        }
        a[t] = sr_pack4(v[0], v[1], v[2]);
    }
}
// block-level reduction of NV per-thread partials into global f32 (one atomic per value per block)
template <int NV>
__device__ __forceinline__ void reduce_to_global(float (&v)[NV], float* __restrict__ out, float* sh /* [4][NV] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float t = wave_sum(v[k]);
        if (lane == 0) sh[wave * NV + k] = t;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < NV; k += 256) atomicAdd(out + k, sh[k] + sh[NV + k] + sh[2 * NV + k] + sh[3 * NV + k]);
}

// backward: dsr = d(0.5 * res_sum)/d pred_img (f32 [B,3,R,R]); gw[168] += {dW1[81], db1[3], dW2[81], db2[3]} (unscaled)
// Weight gradients: the 27 taps of each conv are split over the 4 waves (7,7,7,6); a lane walks 16 of the tile's 1024
// centre pixels and keeps 7 taps x 3 output channels per conv in registers ACROSS tiles (42 accumulators instead of 168),
// reduced across lanes once at the end of the kernel.
__device__ __forceinline__ int sr_tap_off(int tap, int E) {  // tap = (i*3 + ky)*3 + kx  ->  offset of (i, ky, kx) in a [3][E][E] tile
    int i = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    return (i * E + ky) * E + kx;
}

__global__ __launch_bounds__(256, 2) void sr_fused_bwd_kernel(const float* __restrict__ pred_img, BigSrc big,
                                                           const long* __restrict__ column, const long* __restrict__ row, SrP P,
                                                           float* __restrict__ dsr, float* __restrict__ gw, long B, int R, int win) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* U = smem;                       // 3 x 42 x 42   (halo 5)
    float* C1 = U + 3 * 42 * 42;           // 3 x 40 x 40   (halo 4)  -- reused for DU 3 x 34 x 34 (halo 1)
    float* DS = C1 + 3 * 40 * 40;          // 3 x 38 x 38   (halo 3)
    float* DC1 = DS + 3 * 38 * 38;         // 3 x 36 x 36   (halo 2)  -- first holds the 3 x 24 x 24 pred_img patch
    __shared__ SrW W;
    load_srw(W, P);
    const int R2 = 2 * R, G = R2 / SRT, PT = SRT / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tap0 = wave * 7, ntap = wave == 3 ? 6 : 7;
    int off1[7], off2[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        int tap = min(tap0 + t, 26);
        off1[t] = sr_tap_off(tap, 42);  // conv1 taps index U  (edge 42)
        off2[t] = sr_tap_off(tap, 40);  // conv2 taps index C1 (edge 40)
    }
    float a1[7][3], a2[7][3], bb1[3], bb2[3];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int o = 0; o < 3; ++o) a1[t][o] = a2[t][o] = 0.f;
#pragma unroll
    for (int o = 0; o < 3; ++o) bb1[o] = bb2[o] = 0.f;

    for (long t = blockIdx.x; t < B * G * G; t += gridDim.x) {
        const long b = t / (G * G);
        const int ty = (int)((t / G) % G), tx = (int)(t % G);
        const int c0 = (int)column[b], r0 = (int)row[b];
        const int Y0 = ty * SRT, X0 = tx * SRT;
        const int py = threadIdx.x / PT, px = threadIdx.x % PT;  // this thread's pred_img pixel of the 16x16 tile
        if (ty < c0 - 1 || ty > c0 + win || tx < r0 - 1 || tx > r0 + win) {  // no window pixel within reach: gradient is zero
#pragma unroll
            for (int c = 0; c < 3; ++c) dsr[((b * 3 + c) * (long)R + ty * PT + py) * R + tx * PT + px] = 0.f;
            continue;
        }
        __syncthreads();
        // (1) pred_img patch (24 x 24 x 3, clamped at the border) -> LDS with independent coalesced loads
        float* PP = DC1;
        const int ylo = Y0 / 2 - 4, xlo = X0 / 2 - 4;
        for (int idx = threadIdx.x; idx < 3 * 24 * 24; idx += 256) {
            int c = idx / 576, yy = (idx / 24) % 24, xx = idx % 24;
            int y = min(max(ylo + yy, 0), R - 1), x = min(max(xlo + xx, 0), R - 1);
            PP[idx] = pred_img[((b * 3 + c) * (long)R + y) * R + x];
        }
        // prefetch the `big` values this thread needs for ds (halo 3 region, window pixels only)
        float bigv[6][3];
        bool inw[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            int idx = threadIdx.x + 256 * j;
            int y = idx / 38, x = idx % 38;
            int Y = Y0 - 3 + y, X = X0 - 3 + x;
            bool in = idx < 38 * 38 && Y >= 0 && Y < R2 && X >= 0 && X < R2;
            if (in) {
                int gy = Y / SRT, gx = X / SRT;
                in = gy >= c0 && gy < c0 + win && gx >= r0 && gx < r0 + win;
            }
            inw[j] = in;
#pragma unroll
            for (int o = 0; o < 3; ++o) bigv[j][o] = in ? big.at(b, o, Y, X, R2) : 0.f;
        }
        __syncthreads();
        // (2) u on halo 5 from the patch
        for (int idx = threadIdx.x; idx < 3 * 42 * 42; idx += 256) {
            int c = idx / (42 * 42), y = (idx / 42) % 42, x = idx % 42;
            int Y = Y0 - 5 + y, X = X0 - 5 + x;
            float v = 0.f;
            if (Y >= 0 && Y < R2 && X >= 0 && X < R2) {
                int y0, y1, x0, x1;
                float wy0, wy1, wx0, wx1;
                up2_taps(Y, R, y0, y1, wy0, wy1);
                up2_taps(X, R, x0, x1, wx0, wx1);
                const float* pl = PP + c * 576;
                y0 -= ylo; y1 -= ylo; x0 -= xlo; x1 -= xlo;
                v = wy0 * (wx0 * pl[y0 * 24 + x0] + wx1 * pl[y0 * 24 + x1]) + wy1 * (wx0 * pl[y1 * 24 + x0] + wx1 * pl[y1 * 24 + x1]);
            }
            U[idx] = v;
        }
        __syncthreads();
        sr_conv_stage<40, 4, true>(C1, U, W.w1, W.b1, Y0, X0, R2);
        __syncthreads();
        // (3) ds on halo 3: [pixel in window] * (s - big) * [s > 0]
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            int idx = threadIdx.x + 256 * j;
            if (idx < 38 * 38) {
                int y = idx / 38, x = idx % 38;
                float acc[3] = {W.b2[0], W.b2[1], W.b2[2]};
                if (inw[j]) {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                float v = C1[(i * 40 + y + ky) * 40 + x + kx];
#pragma unroll
                                for (int o = 0; o < 3; ++o) acc[o] += W.w2[((o * 3 + i) * 3 + ky) * 3 + kx] * v;
                            }
                }
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    float d = 0.f;
                    if (inw[j]) {
                        float s = fmaxf(acc[o] + U[(o * 42 + y + 2) * 42 + x + 2], 0.f);
                        d = s > 0.f ? s - bigv[j][o] : 0.f;
                    }
                    DS[(o * 38 + y) * 38 + x] = d;
                }
            }
        }
        __syncthreads();
        // (4) dc1 on halo 2 = [c1 > 0] * conv2^T(ds)
        for (int idx = threadIdx.x; idx < 36 * 36; idx += 256) {
            int y = idx / 36, x = idx % 36;
            int Y = Y0 - 2 + y, X = X0 - 2 + x;
            const bool in = Y >= 0 && Y < R2 && X >= 0 && X < R2;
            float acc[3] = {0.f, 0.f, 0.f};
            if (in) {
#pragma unroll
                for (int o = 0; o < 3; ++o)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            float v = DS[(o * 38 + y + 2 - ky) * 38 + x + 2 - kx];
#pragma unroll
                            for (int i = 0; i < 3; ++i) acc[i] += W.w2[((o * 3 + i) * 3 + ky) * 3 + kx] * v;
                        }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) DC1[(i * 36 + y) * 36 + x] = (in && C1[(i * 40 + y + 2) * 40 + x + 2] > 0.f) ? acc[i] : 0.f;
        }
        __syncthreads();
        // (5) weight gradients over the 32x32 centre: conv2 from (ds, c1), conv1 from (dc1, u)
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            int idx = lane + 64 * j;
            int y = idx >> 5, x = idx & 31;
            float d2[3] = {DS[(0 * 38 + y + 3) * 38 + x + 3], DS[(1 * 38 + y + 3) * 38 + x + 3], DS[(2 * 38 + y + 3) * 38 + x + 3]};
            float d1[3] = {DC1[(0 * 36 + y + 2) * 36 + x + 2], DC1[(1 * 36 + y + 2) * 36 + x + 2], DC1[(2 * 36 + y + 2) * 36 + x + 2]};
            const float* c1p = C1 + (y + 3) * 40 + x + 3;   // tap (ky,kx) reads centre + (ky-1, kx-1) -> base shifted by (-1,-1) below
            const float* up = U + (y + 4) * 42 + x + 4;
#pragma unroll
            for (int tt = 0; tt < 7; ++tt) {
                if (tt < ntap) {
                    float v2 = c1p[off2[tt]], v1 = up[off1[tt]];
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        a2[tt][o] += d2[o] * v2;
                        a1[tt][o] += d1[o] * v1;
                    }
                }
            }
            if (wave == 3) {
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    bb2[o] += d2[o];
                    bb1[o] += d1[o];
                }
            }
        }
        __syncthreads();
        // (6) du on halo 1 (into the C1 buffer) = ds + conv1^T(dc1)
        float* DU = C1;
        for (int idx = threadIdx.x; idx < 34 * 34; idx += 256) {
            int y = idx / 34, x = idx % 34;
            int Y = Y0 - 1 + y, X = X0 - 1 + x;
            const bool in = Y >= 0 && Y < R2 && X >= 0 && X < R2;
            float acc[3] = {0.f, 0.f, 0.f};
            if (in) {
#pragma unroll
                for (int o = 0; o < 3; ++o)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            float v = DC1[(o * 36 + y + 2 - ky) * 36 + x + 2 - kx];
#pragma unroll
                            for (int i = 0; i < 3; ++i) acc[i] += W.w1[((o * 3 + i) * 3 + ky) * 3 + kx] * v;
                        }
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[i] += DS[(i * 38 + y + 2) * 38 + x + 2];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) DU[(i * 34 + y) * 34 + x] = in ? acc[i] : 0.f;
        }
        __syncthreads();
        // (7) transpose of the bilinear x2: each pred_img pixel gathers the <= 4x4 du values whose footprint touches it
        {
            const int y = ty * PT + py, x = tx * PT + px;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float acc = 0.f;
                for (int Y = max(2 * y - 1, 0); Y <= min(2 * y + 2, R2 - 1); ++Y) {
                    int y0, y1;
                    float wy0, wy1;
                    up2_taps(Y, R, y0, y1, wy0, wy1);
                    float wy = (y0 == y ? wy0 : 0.f) + (y1 == y ? wy1 : 0.f);
                    if (wy == 0.f) continue;
                    for (int X = max(2 * x - 1, 0); X <= min(2 * x + 2, R2 - 1); ++X) {
                        int x0, x1;
                        float wx0, wx1;
                        up2_taps(X, R, x0, x1, wx0, wx1);
                        float wx = (x0 == x ? wx0 : 0.f) + (x1 == x ? wx1 : 0.f);
                        if (wx != 0.f) acc += wy * wx * DU[(c * 34 + (Y - Y0 + 1)) * 34 + (X - X0 + 1)];
                    }
                }
                dsr[((b * 3 + c) * (long)R + y) * R + x] = acc;
            }
        }
    }
    // one cross-lane reduction per kernel: gw layout {dW1[81], db1[3], dW2[81], db2[3]}, dW[o][i][ky][kx] = index o*27 + tap
#pragma unroll
    for (int tt = 0; tt < 7; ++tt)
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float s1 = wave_sum(a1[tt][o]), s2 = wave_sum(a2[tt][o]);
            if (lane == 0 && tt < ntap) {
                atomicAdd(gw + o * 27 + tap0 + tt, s1);
                atomicAdd(gw + 84 + o * 27 + tap0 + tt, s2);
            }
        }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        float s1 = wave_sum(bb1[o]), s2 = wave_sum(bb2[o]);
        if (lane == 0 && wave == 3) {
            atomicAdd(gw + 81 + o, s1);
            atomicAdd(gw + 84 + 81 + o, s2);
        }
    }
}

__device__ __forceinline__ void sr_unpack3(uint2 q, float (&v)[3]) {   // the three channels of an interleaved bf16 pixel as f32
    v[0] = h16_lo(q.x);
    v[1] = h16_hi(q.x);
    v[2] = h16_lo(q.y);
}

// ---- backward on the matrix cores, pixel-PAIR form (the production kernel of compute_dtype = bf16) ---------------------------------
// The eight stages of sr_fused_bwd_kernel on bf16 tiles; what a lane does per instruction:
//  * a lane convolves TWO horizontally adjacent pixels: the 4 x 3 input pixels they share are six 16-B LDS reads (48 B per output
//    pixel instead of 72, a third of the read instructions).  For the reads to be 16-B aligned the tiles alternate their column
//    origin: u at -6 (even), c1 at -5 (pairs start on odd columns), ds at -4, dc1 at -3, du at -2 -- every stage computes one spare
//    column on each side.
//  * u comes from 2 x 2 blocks (the x2 bilinear filter has two phases: .25/.75 and .75/.25 on a 3 x 3 patch of pred_img).
//  * weight gradients run on the matrix cores too: ds_read_b64_tr_b16 turns the channel-interleaved pixels of 4 lanes into
//    "4 pixels of one channel" per lane, which is exactly a row of A (ds[o][4 pixels]) and a column of B (c1[i][the 4 pixels, shifted
//    by the tap]) of v_mfma_f32_4x4x4: D[o][i] accumulates the tap's 3 x 3 channel block over pixels, 16 blocks per wave; one MFMA
//    against a vector of ones gives the bias gradient.  2 + 18 transpose reads and 20 MFMAs per 64 pixels replace ~90 VALU
//    instructions per 64 pixels and wave of the tap-split f32 form.
//  * the 168 gradient sums leave the workgroup once, reduced through LDS, as three wave-wide atomics (the per-wave single-lane
//    atomics of the older kernels serialise on three cache lines).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
__device__ __forceinline__ bf16x4_t sr_px(unsigned a, unsigned b) { return __builtin_bit_cast(bf16x4_t, (u32x2_t){a, b}); }
__device__ __forceinline__ bf16x4_t sr_tr(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4_t __attribute__((address_space(3)))*)p);
}
// 3x3 stencils of the pixel pair at columns 1 and 2 of the 3 x 4 input window at p (tile edge ES pixels); two MFMA chains per pixel
template <int ES, bool FLIP>
__device__ __forceinline__ void sr_pair_conv(const unsigned char* p, const bf16x4_t (&a)[9], const f32x4_t init, f32x4_t& oA, f32x4_t& oB) {
    f32x4_t a0 = init, b0 = init, a1 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, b1 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(p + r * ES * 8);
        const u32x4_t hi = *reinterpret_cast<const u32x4_t*>(p + r * ES * 8 + 16);
        const bf16x4_t px[4] = {sr_px(lo[0], lo[1]), sr_px(lo[2], lo[3]), sr_px(hi[0], hi[1]), sr_px(hi[2], hi[3])};
        const int ky = FLIP ? 2 - r : r;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int kx = FLIP ? 2 - j : j;
            if ((r * 3 + j) & 1) {
                a1 = MFMA4(a[ky * 3 + kx], px[j], a1);
                b1 = MFMA4(a[ky * 3 + kx], px[j + 1], b1);
            } else {
                a0 = MFMA4(a[ky * 3 + kx], px[j], a0);
                b0 = MFMA4(a[ky * 3 + kx], px[j + 1], b0);
            }
        }
    }
    oA = a0 + a1;
    oB = b0 + b1;
}
// forward in the same pixel-pair form: u on rows / cols -2..33 from 2 x 2 blocks, c1 on rows -1..32 with column pairs from -1,
// s on the tile with column pairs from 0.  The skip connection and the loss stay f32.
__global__ __launch_bounds__(256) void sr_pair_fwd_kernel(const float* __restrict__ pred_img, BigSrc big,
                                                          const long* __restrict__ column, const long* __restrict__ row, SrP P,
                                                          float* __restrict__ loss_sum, long B, int R, int win) {
    __shared__ __attribute__((aligned(16))) unsigned char U16[36 * 36 * 8];   // u, rows / cols -2..33
    __shared__ __attribute__((aligned(16))) unsigned char C16[34 * 34 * 8];   // c1, rows / cols -1..32
    __shared__ __attribute__((aligned(16))) float UC[3 * 32 * 32];            // u on the tile itself in f32 (skip connection)
    __shared__ float PP[3 * 20 * 20];                                          // pred_img patch (rows/cols Y0/2-2 .. Y0/2+17, clamped)
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x4_t a1[9], a2[9];
    sr_load_taps<false>(P.w1, lane, a1);
    sr_load_taps<false>(P.w2, lane, a2);
    const f32x4_t bias1 = {P.b1[0], P.b1[1], P.b1[2], 0.f}, bias2 = {P.b2[0], P.b2[1], P.b2[2], 0.f};
    const int R2 = 2 * R, G = R2 / SRT, GG = G * G, NT = (int)(B * GG);
    float part = 0.f;
    for (int t = blockIdx.x; t < NT; t += gridDim.x) {
        const int b = t / GG, ty = (t - b * GG) / G, tx = t - b * GG - ty * G;
        const int c0 = (int)column[b], r0 = (int)row[b];
        if (ty < c0 || ty >= c0 + win || tx < r0 || tx >= r0 + win) continue;  // block-uniform
        const int Y0 = ty * SRT, X0 = tx * SRT;
        const int ylo = Y0 / 2 - 2, xlo = X0 / 2 - 2;
        __syncthreads();
        for (int idx = threadIdx.x; idx < 3 * 400; idx += 256) {
            const int c = idx / 400, yy = (idx / 20) % 20, xx = idx % 20;
            const int y = min(max(ylo + yy, 0), R - 1), x = min(max(xlo + xx, 0), R - 1);
            PP[idx] = pred_img[((b * 3 + c) * (long)R + y) * R + x];
        }
        __syncthreads();
        for (int id = threadIdx.x; id < 18 * 18; id += 256) {       // u in 2 x 2 blocks (see sr_pair_bwd_kernel)
            const int by = id / 18, bx = id - by * 18;
            float o[3][2][2];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* pl = PP + c * 400 + by * 20 + bx;
                float he[3], ho[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float v0 = pl[r * 20], v1 = pl[r * 20 + 1], v2 = pl[r * 20 + 2];
                    he[r] = 0.25f * v0 + 0.75f * v1;
                    ho[r] = 0.75f * v1 + 0.25f * v2;
                }
                o[c][0][0] = 0.25f * he[0] + 0.75f * he[1];
                o[c][0][1] = 0.25f * ho[0] + 0.75f * ho[1];
                o[c][1][0] = 0.75f * he[1] + 0.25f * he[2];
                o[c][1][1] = 0.75f * ho[1] + 0.25f * ho[2];
            }
            const int Y = Y0 - 2 + 2 * by, X = X0 - 2 + 2 * bx;       // both even: a block is inside or outside the image as a whole
            const bool in = Y >= 0 && Y < R2 && X >= 0 && X < R2;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                u32x4_t q;
                q[0] = in ? pack_bf16x2(o[0][e][0], o[1][e][0]) : 0u;
                q[1] = in ? pack_bf16x2(o[2][e][0], 0.f) : 0u;
                q[2] = in ? pack_bf16x2(o[0][e][1], o[1][e][1]) : 0u;
                q[3] = in ? pack_bf16x2(o[2][e][1], 0.f) : 0u;
                *reinterpret_cast<u32x4_t*>(U16 + ((2 * by + e) * 36 + 2 * bx) * 8) = q;
            }
            if (by >= 1 && by <= 16 && bx >= 1 && bx <= 16) {       // the tile itself (always inside the image)
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        *reinterpret_cast<float2*>(UC + (c * 32 + 2 * by - 2 + e) * 32 + 2 * bx - 2) = make_float2(o[c][e][0], o[c][e][1]);
            }
        }
        __syncthreads();
        for (int g = wave; g < (34 * 17 + 63) / 64; g += 4) {         // c1 = relu(conv1(u) + b1), 0 outside the image
            const int p = g * 64 + lane, pc = min(p, 34 * 17 - 1);
            const int y = pc / 17, xq = pc - y * 17;
            f32x4_t oA, oB;
            sr_pair_conv<36, false>(U16 + (y * 36 + 2 * xq) * 8, a1, bias1, oA, oB);
            const int Y = Y0 - 1 + y, X = X0 - 1 + 2 * xq;
            const bool iy = Y >= 0 && Y < R2, inA = iy && X >= 0 && X < R2, inB = iy && X + 1 >= 0 && X + 1 < R2;
            u32x4_t q;
            q[0] = inA ? pack_bf16x2(fmaxf(oA[0], 0.f), fmaxf(oA[1], 0.f)) : 0u;
            q[1] = inA ? pack_bf16x2(fmaxf(oA[2], 0.f), 0.f) : 0u;
            q[2] = inB ? pack_bf16x2(fmaxf(oB[0], 0.f), fmaxf(oB[1], 0.f)) : 0u;
            q[3] = inB ? pack_bf16x2(fmaxf(oB[2], 0.f), 0.f) : 0u;
            if (p < 34 * 17) *reinterpret_cast<u32x4_t*>(C16 + (y * 34 + 2 * xq) * 8) = q;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {                                  // s = relu(conv2(c1) + b2 + u); loss
            const int p = (wave + 4 * j) * 64 + lane, y = p >> 4, xq = p & 15;
            float2 bg[3];
#pragma unroll
            for (int o = 0; o < 3; ++o) bg[o] = big.at2(b, o, Y0 + y, X0 + 2 * xq, R2);
            f32x4_t oA, oB;
            sr_pair_conv<34, false>(C16 + (y * 34 + 2 * xq) * 8, a2, bias2, oA, oB);
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float2 uu = *reinterpret_cast<const float2*>(UC + (o * 32 + y) * 32 + 2 * xq);
                const float dA = fmaxf(oA[o] + uu.x, 0.f) - bg[o].x, dB = fmaxf(oB[o] + uu.y, 0.f) - bg[o].y;
                part += dA * dA + dB * dB;
            }
        }
    }
    part = block_sum_256(part, sh);
    if (threadIdx.x == 0) atomicAdd(loss_sum, part);
}

#define SRP_LDS_BYTES (44 * 44 * 8 + 40 * 42 * 8 + 38 * 40 * 8 + 36 * 38 * 8 + 3 * 34 * 36 * 4 + 4 * 9 * 4 * 8)
__global__ __launch_bounds__(256, 2) void sr_pair_bwd_kernel(const float* __restrict__ pred_img, BigSrc big,
                                                             const long* __restrict__ column, const long* __restrict__ row, SrP P,
                                                             float* __restrict__ dsr, float* __restrict__ gw, long B, int R, int win) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];   // the only LDS object (16-B aligned carve)
    unsigned char* U16 = smem8;                          // rows -6..37 x cols -6..37   (44 x 44 px of 8 B)
    unsigned char* C16 = U16 + 44 * 44 * 8;              // rows -4..35 x cols -5..36   (40 x 42)
    unsigned char* DS16 = C16 + 40 * 42 * 8;             // rows -3..34 x cols -4..35   (38 x 40)
    unsigned char* DC16 = DS16 + 38 * 40 * 8;            // rows -2..33 x cols -3..34   (36 x 38)
    float* DU = reinterpret_cast<float*>(DC16 + 36 * 38 * 8);   // [3] x rows -1..32 x cols -2..33 (34 x 36 f32); first the 3 x 24 x 24 pred_img patch
    float* PP = DU;
    bf16x4_t(*TAPS)[9][4] = reinterpret_cast<bf16x4_t(*)[9][4]>(reinterpret_cast<unsigned char*>(DU) + 3 * 34 * 36 * 4);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 144) {
        const int c = threadIdx.x / 36, t = (threadIdx.x / 4) % 9, i = threadIdx.x & 3;
        const float* w = (c == 0 || c == 3) ? P.w1 : P.w2;
        const bool trans = c >= 2;
        float v[3] = {0.f, 0.f, 0.f};
        if (i < 3) {
            for (int kk = 0; kk < 3; ++kk) v[kk] = trans ? w[(kk * 3 + i) * 9 + t] : w[(i * 3 + kk) * 9 + t];
        }
        TAPS[c][t][i] = sr_pack4(v[0], v[1], v[2]);
    }
#define SR_TAPS(NAME, C)                                                     \
    bf16x4_t NAME[9];                                                        \
    _Pragma("unroll") for (int t_ = 0; t_ < 9; ++t_) NAME[t_] = TAPS[(C)][t_][lane & 3]
    const f32x4_t bias1 = {P.b1[0], P.b1[1], P.b1[2], 0.f}, bias2 = {P.b2[0], P.b2[1], P.b2[2], 0.f}, zero4 = {0.f, 0.f, 0.f, 0.f};
    const bf16x4_t ones = {(short)H16_ONE, (short)H16_ONE, (short)H16_ONE, (short)H16_ONE};
    const int R2 = 2 * R, G = R2 / SRT, PT = SRT / 2;
    const long T = B * G * G;
    f32x4_t G1[9], G2[9], gb1 = zero4, gb2 = zero4;   // [tap][o] of input channel lane & 3, summed over this lane's block's pixels
#pragma unroll
    for (int q = 0; q < 9; ++q) G1[q] = G2[q] = zero4;
    const int py = threadIdx.x / PT, px = threadIdx.x % PT;  // this thread's pred_img pixel of a 16x16 tile

    // tile cursor: (b, ty, tx) of tile t, stepped by the grid size without divisions (all wave-uniform, 32-bit: B*G*G < 2^31, host-checked)
    struct Cur { int t, b, ty, tx; };
    const int GG = G * G, NT = (int)T;
    const int step_b = (int)gridDim.x / GG, step_r = (int)gridDim.x - step_b * GG, step_y = step_r / G, step_x = step_r - step_y * G;
    auto step = [&](Cur c) {
        c.t += (int)gridDim.x;
        c.tx += step_x;
        if (c.tx >= G) { c.tx -= G; ++c.ty; }
        c.ty += step_y;
        if (c.ty >= G) { c.ty -= G; ++c.b; }
        c.b += step_b;
        return c;
    };
    auto advance = [&](Cur c) {   // next reachable tile at or after c; tiles with no window pixel within reach get a zero gradient
        for (; c.t < NT; c = step(c)) {
            const int c0 = (int)column[c.b], r0 = (int)row[c.b];
            if (!(c.ty < c0 - 1 || c.ty > c0 + win || c.tx < r0 - 1 || c.tx > r0 + win)) break;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) dsr[((c.b * 3 + ch) * (long)R + c.ty * PT + py) * R + c.tx * PT + px] = 0.f;
        }
        return c;
    };
    float pp[7], bigv[3][3][2];
    bool inw[3];
    auto fetch_patch = [&](const Cur& cu) {
        const long b = cu.b;
        const int ylo = cu.ty * (SRT / 2) - 4, xlo = cu.tx * (SRT / 2) - 4;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int idx = threadIdx.x + 256 * k;
            const int c = idx / 576, yy = (idx / 24) % 24, xx = idx % 24;
            const int y = min(max(ylo + yy, 0), R - 1), x = min(max(xlo + xx, 0), R - 1);
            pp[k] = idx < 3 * 576 ? pred_img[((b * 3 + c) * (long)R + y) * R + x] : 0.f;
        }
    };
    auto fetch_big = [&](const Cur& cu) {   // the targets of pixel pair (wave + 4j)*64 + lane of the 38 x 20 pairs of the ds stage
        const long b = cu.b;
        const int Y0 = cu.ty * SRT, X0 = cu.tx * SRT;
        const int c0 = (int)column[b], r0 = (int)row[b];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int p = (wave + 4 * j) * 64 + lane;
            const int y = p / 20, xq = p - y * 20;
            const int Y = Y0 - 3 + y, X = X0 - 4 + 2 * xq;
            bool in = p < 38 * 20 && Y >= 0 && Y < R2 && X >= 0 && X < R2;
            if (in) {
                const int gy = Y / SRT, gx = X / SRT;
                in = gy >= c0 && gy < c0 + win && gx >= r0 && gx < r0 + win;
            }
            inw[j] = in;
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float2 v = in ? big.at2(b, o, Y, X, R2) : make_float2(0.f, 0.f);
                bigv[j][o][0] = v.x;
                bigv[j][o][1] = v.y;
            }
        }
    };
    Cur cur;
    cur.t = (int)blockIdx.x; cur.b = cur.t / GG; cur.ty = (cur.t - cur.b * GG) / G; cur.tx = cur.t - cur.b * GG - cur.ty * G;
    cur = advance(cur);
    if (cur.t < NT) {
        fetch_patch(cur);
        fetch_big(cur);
    }
    while (cur.t < NT) {
        const long b = cur.b;
        const int ty = cur.ty, tx = cur.tx;
        const int Y0 = ty * SRT, X0 = tx * SRT;
        __syncthreads();   // the previous tile's stage 8 is done with DU (= the patch buffer)
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int idx = threadIdx.x + 256 * k;
            if (idx < 3 * 576) PP[idx] = pp[k];
        }
        const Cur nxt = advance(step(cur));
        if (nxt.t < NT) fetch_patch(nxt);     // in flight until the next iteration
        __syncthreads();
        // (2) u on rows / cols -6..37 in 2 x 2 blocks: even outputs .25 p[y-1] + .75 p[y], odd ones .75 p[y] + .25 p[y+1]
        for (int id = threadIdx.x; id < 22 * 22; id += 256) {
            const int by = id / 22, bx = id - by * 22;
            float o[3][2][2];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* pl = PP + c * 576 + by * 24 + bx;
                float he[3], ho[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float v0 = pl[r * 24], v1 = pl[r * 24 + 1], v2 = pl[r * 24 + 2];
                    he[r] = 0.25f * v0 + 0.75f * v1;
                    ho[r] = 0.75f * v1 + 0.25f * v2;
                }
                o[c][0][0] = 0.25f * he[0] + 0.75f * he[1];
                o[c][0][1] = 0.25f * ho[0] + 0.75f * ho[1];
                o[c][1][0] = 0.75f * he[1] + 0.25f * he[2];
                o[c][1][1] = 0.75f * ho[1] + 0.25f * ho[2];
            }
            const int Y = Y0 - 6 + 2 * by, X = X0 - 6 + 2 * bx;       // both even: a block is inside or outside the image as a whole
            const bool in = Y >= 0 && Y < R2 && X >= 0 && X < R2;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                u32x4_t q;
                q[0] = in ? pack_bf16x2(o[0][e][0], o[1][e][0]) : 0u;
                q[1] = in ? pack_bf16x2(o[2][e][0], 0.f) : 0u;
                q[2] = in ? pack_bf16x2(o[0][e][1], o[1][e][1]) : 0u;
                q[3] = in ? pack_bf16x2(o[2][e][1], 0.f) : 0u;
                *reinterpret_cast<u32x4_t*>(U16 + ((2 * by + e) * 44 + 2 * bx) * 8) = q;
            }
        }
        __syncthreads();
        // (3) c1 = relu(conv1(u) + b1) on rows -4..35, column pairs from -5; 0 outside the image
        {
            SR_TAPS(a1, 0);
            for (int g = wave; g < (40 * 21 + 63) / 64; g += 4) {
                const int p = g * 64 + lane, pc = min(p, 40 * 21 - 1);
                const int y = pc / 21, xq = pc - y * 21;
                f32x4_t oA, oB;
                sr_pair_conv<44, false>(U16 + ((y + 1) * 44 + 2 * xq) * 8, a1, bias1, oA, oB);
                const int Y = Y0 - 4 + y, X = X0 - 5 + 2 * xq;
                const bool iy = Y >= 0 && Y < R2, inA = iy && X >= 0 && X < R2, inB = iy && X + 1 >= 0 && X + 1 < R2;
                u32x4_t q;
                q[0] = inA ? pack_bf16x2(fmaxf(oA[0], 0.f), fmaxf(oA[1], 0.f)) : 0u;
                q[1] = inA ? pack_bf16x2(fmaxf(oA[2], 0.f), 0.f) : 0u;
                q[2] = inB ? pack_bf16x2(fmaxf(oB[0], 0.f), fmaxf(oB[1], 0.f)) : 0u;
                q[3] = inB ? pack_bf16x2(fmaxf(oB[2], 0.f), 0.f) : 0u;
                if (p < 40 * 21) *reinterpret_cast<u32x4_t*>(C16 + (y * 42 + 2 * xq) * 8) = q;
            }
        }
        __syncthreads();
        // (4) ds = [pixel in window] * (s - big) * [s > 0], s = relu(conv2(c1) + b2 + u), on rows -3..34, column pairs from -4
        {
            SR_TAPS(a2, 1);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int g = wave + 4 * j;
                const int p = g * 64 + lane, pc = min(p, 38 * 20 - 1);
                const int y = pc / 20, xq = pc - y * 20;
                f32x4_t oA, oB;
                sr_pair_conv<42, false>(C16 + (y * 42 + 2 * xq) * 8, a2, bias2, oA, oB);
                float dA[3] = {0.f, 0.f, 0.f}, dB[3] = {0.f, 0.f, 0.f};
                if (inw[j]) {
                    const u32x4_t uu = *reinterpret_cast<const u32x4_t*>(U16 + ((y + 3) * 44 + 2 * xq + 2) * 8);
                    float uA[3], uB[3];
                    sr_unpack3(make_uint2(uu[0], uu[1]), uA);
                    sr_unpack3(make_uint2(uu[2], uu[3]), uB);
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        const float sA = fmaxf(oA[o] + uA[o], 0.f), sB = fmaxf(oB[o] + uB[o], 0.f);
                        dA[o] = sA > 0.f ? sA - bigv[j][o][0] : 0.f;
                        dB[o] = sB > 0.f ? sB - bigv[j][o][1] : 0.f;
                    }
                }
                if (p < 38 * 20)
                    *reinterpret_cast<u32x4_t*>(DS16 + (y * 40 + 2 * xq) * 8) =
                        (u32x4_t){pack_bf16x2(dA[0], dA[1]), pack_bf16x2(dA[2], 0.f), pack_bf16x2(dB[0], dB[1]), pack_bf16x2(dB[2], 0.f)};
            }
        }
        if (nxt.t < NT) fetch_big(nxt);       // bigv / inw are free again: request the next tile's targets
        __syncthreads();
        // (5) dc1 = [c1 > 0] * conv2^T(ds) on rows -2..33, column pairs from -3 (c1 is 0 outside the image, so is dc1)
        {
            SR_TAPS(a2t, 2);
            for (int g = wave; g < (36 * 19 + 63) / 64; g += 4) {
                const int p = g * 64 + lane, pc = min(p, 36 * 19 - 1);
                const int y = pc / 19, xq = pc - y * 19;
                f32x4_t oA, oB;
                sr_pair_conv<40, true>(DS16 + (y * 40 + 2 * xq) * 8, a2t, zero4, oA, oB);
                const u32x4_t cc = *reinterpret_cast<const u32x4_t*>(C16 + ((y + 2) * 42 + 2 * xq + 2) * 8);
                float cA[3], cB[3];
                sr_unpack3(make_uint2(cc[0], cc[1]), cA);
                sr_unpack3(make_uint2(cc[2], cc[3]), cB);
                float dA[3], dB[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    dA[i] = cA[i] > 0.f ? oA[i] : 0.f;
                    dB[i] = cB[i] > 0.f ? oB[i] : 0.f;
                }
                if (p < 36 * 19)
                    *reinterpret_cast<u32x4_t*>(DC16 + (y * 38 + 2 * xq) * 8) =
                        (u32x4_t){pack_bf16x2(dA[0], dA[1]), pack_bf16x2(dA[2], 0.f), pack_bf16x2(dB[0], dB[1]), pack_bf16x2(dB[2], 0.f)};
            }
        }
        __syncthreads();
        // (6) weight gradients over the 32 x 32 centre: conv2 from (ds, c1), conv1 from (dc1, u).  Lane l of a group of 64 pixels
        // (two rows) names pixel l; after the transpose read lane (block, ch) holds channel ch of the block's four pixels.
        // A wave's four groups are stacked (rows 8w .. 8w+7): the operands of kernel row 2 of one group are those of kernel row 0 of the next
        {
            bf16x4_t k2[3], k1[3];   // kernel-row-0 operands of the coming group (c1 / u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = 2 * (wave * 4 + j) + (lane >> 5), x = lane & 31;
                const bf16x4_t A2 = sr_tr(DS16 + ((y + 3) * 40 + x + 4) * 8);
                const bf16x4_t A1 = sr_tr(DC16 + ((y + 2) * 38 + x + 3) * 8);
                gb2 = MFMA4(A2, ones, gb2);
                gb1 = MFMA4(A1, ones, gb1);
                const unsigned char* cb = C16 + ((y + 3) * 42 + x + 4) * 8;   // tap (ky, kx) reads centre + (ky - 1, kx - 1)
                const unsigned char* ub = U16 + ((y + 5) * 44 + x + 5) * 8;
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    const int ky = q / 3, kx = q % 3;
                    const bf16x4_t B2 = (ky == 0 && j > 0) ? k2[kx] : sr_tr(cb + (ky * 42 + kx) * 8);
                    const bf16x4_t B1 = (ky == 0 && j > 0) ? k1[kx] : sr_tr(ub + (ky * 44 + kx) * 8);
                    G2[q] = MFMA4(A2, B2, G2[q]);
                    G1[q] = MFMA4(A1, B1, G1[q]);
                    if (ky == 2) { k2[kx] = B2; k1[kx] = B1; }
                }
            }
        }
        // (7) du = ds + conv1^T(dc1) on rows -1..32, column pairs from -2 (0 outside the image), and at once the horizontal half of
        // the transposed x2 bilinear filter: pred column x' gathers .25 du[2x'-1] + wa du[2x'] + wb du[2x'+1] + .25 du[2x'+2] with
        // wa = .75 (1 at x' = 0), wb = .75 (1 at x' = R-1); pair xq = x' + 1 of a row holds du[2x'], du[2x'+1], its lane neighbours
        // the other two (pairs 0 and 17 only serve as neighbours).  H: [3] x rows -1..32 x 16 columns, row pitch 24 floats.
        {
            SR_TAPS(a1t, 3);
            // groups of 62 pairs: lanes 0 and 63 recompute the neighbours' pairs so that every lane in between finds both in the wave
            for (int g = wave; g < (34 * 18 + 61) / 62; g += 4) {
                const int p = g * 62 - 1 + lane, pc = min(max(p, 0), 34 * 18 - 1);
                const int y = pc / 18, xq = pc - y * 18;
                f32x4_t oA, oB;
                sr_pair_conv<38, true>(DC16 + (y * 38 + 2 * xq) * 8, a1t, zero4, oA, oB);
                const u32x4_t dd = *reinterpret_cast<const u32x4_t*>(DS16 + ((y + 2) * 40 + 2 * xq + 2) * 8);
                float dA[3], dB[3];
                sr_unpack3(make_uint2(dd[0], dd[1]), dA);
                sr_unpack3(make_uint2(dd[2], dd[3]), dB);
                const int Y = Y0 - 1 + y, X = X0 - 2 + 2 * xq;     // X even: the pair is inside or outside as a whole
                const bool in = Y >= 0 && Y < R2 && X >= 0 && X < R2;
                const float wa = X == 0 ? 1.0f : 0.75f, wb = X == R2 - 2 ? 1.0f : 0.75f;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float va = in ? oA[i] + dA[i] : 0.f, vb = in ? oB[i] + dB[i] : 0.f;
                    const float left = __shfl_up(vb, 1, 64), right = __shfl_down(va, 1, 64);
                    const float h = wa * va + wb * vb + 0.25f * (left + right);
                    if (lane >= 1 && lane <= 62 && p < 34 * 18 && xq >= 1 && xq <= 16) DU[(i * 34 + y) * 24 + xq - 1] = h;
                }
            }
        }
        __syncthreads();
        // (8) the vertical half: pred row y' gathers .25 H[2y'-1] + wa H[2y'] + wb H[2y'+1] + .25 H[2y'+2] (rows outside the image hold 0)
        {
            const int y = ty * PT + py, x = tx * PT + px;
            const float wa = y == 0 ? 1.0f : 0.75f, wb = y == R - 1 ? 1.0f : 0.75f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* h = DU + (c * 34 + 2 * py) * 24 + px;   // H row of hi-res row 2y'-1
                dsr[((b * 3 + c) * (long)R + y) * R + x] = 0.25f * (h[0] + h[72]) + wa * h[24] + wb * h[48];
            }
        }
        cur = nxt;
    }
#undef SR_TAPS
    // the workgroup's 168 sums: blocks of a wave by shuffles (lanes with the same lane & 3), waves through LDS, then one atomic per
    // value: gw = {dW1[81], db1[3], dW2[81], db2[3]}, dW[o][i][ky][kx] at o*27 + i*9 + tap
    __syncthreads();
    float* RED = reinterpret_cast<float*>(U16);   // [4 waves][168]
    auto blocks_sum = [&](float v) {
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    const int ic = lane & 3;
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float s1 = blocks_sum(G1[q][o]), s2 = blocks_sum(G2[q][o]);
            if (lane < 3) {
                RED[wave * 168 + o * 27 + ic * 9 + q] = s1;
                RED[wave * 168 + 84 + o * 27 + ic * 9 + q] = s2;
            }
        }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const float s1 = blocks_sum(gb1[o]), s2 = blocks_sum(gb2[o]);
        if (lane == 0) {
            RED[wave * 168 + 81 + o] = s1;
            RED[wave * 168 + 165 + o] = s2;
        }
    }
    __syncthreads();
    if (threadIdx.x < 168) atomicAdd(gw + threadIdx.x, RED[threadIdx.x] + RED[168 + threadIdx.x] + RED[336 + threadIdx.x] + RED[504 + threadIdx.x]);
}

static BigSrc big_src(const void* big, const float* lut) {   // lut != null: `big` is the uint8 crop
    BigSrc s;
    s.f = lut ? nullptr : reinterpret_cast<const float*>(big);
    s.u8 = lut ? reinterpret_cast<const unsigned char*>(big) : nullptr;
    s.lut = lut;
    return s;
}
extern "C" int ecamp_sr_fwd(const float* pred_img, const void* big, const float* big_lut, const int64_t* column,
                            const int64_t* row, const float* w1, const float* b1, const float* w2, const float* b2, float* loss_sum, int64_t B,
                            int32_t R, int32_t super_patch, int32_t window, int32_t mode, hipStream_t stream) {
    ECAMP_CHECK_ARG(pred_img && big && column && row && w1 && b1 && w2 && b2 && loss_sum, "sr_fwd: null pointer");
    ECAMP_CHECK_ARG(!big_lut || ((uintptr_t)big & 1) == 0, "sr_fwd: the uint8 target must be 2-byte aligned");
    const BigSrc bs = big_src(big, big_lut);
    ECAMP_CHECK_ARG(super_patch == SRT && (2 * R) % SRT == 0, "sr_fwd: the fused SR head is built for 32-px super-patches (patch 16)");
    ECAMP_CHECK_ARG(mode == 0 || mode == 1, "sr_fwd: mode must be 0 (f32 VALU) or 1 (bf16 matrix cores)");
    SrP W = {w1, b1, w2, b2};
    long tiles = B * (2 * R / SRT) * (2 * R / SRT);
    int nb = (int)(tiles < 2048 ? tiles : 2048);
    if (mode == 1)
        hipLaunchKernelGGL(sr_pair_fwd_kernel, dim3(nb), dim3(256), 0, stream, pred_img, bs, (const long*)column, (const long*)row, W, loss_sum,
                           (long)B, R, window);
    else
        hipLaunchKernelGGL(sr_fused_fwd_kernel, dim3(nb), dim3(256), 0, stream, pred_img, bs, (const long*)column, (const long*)row, W, loss_sum,
                           (long)B, R, window, (float*)nullptr);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// The SR head's output image itself (f32 stencils): sr [B,3,2R,2R] = relu(conv2(relu(conv1(up2(pred_img)))) + up2(pred_img)),
// model_ecamp.py:28-46.  The training step never materialises it (ecamp_sr_fwd folds it into the loss); this entry serves the parity
// checks against the reference's `super_res` output and visualisation.
extern "C" int ecamp_sr_image(const float* pred_img, const float* w1, const float* b1, const float* w2, const float* b2, float* sr,
                              int64_t B, int32_t R, hipStream_t stream) {
    ECAMP_CHECK_ARG(pred_img && w1 && b1 && w2 && b2 && sr, "sr_image: null pointer");
    ECAMP_CHECK_ARG((2 * R) % SRT == 0, "sr_image: 2R must be a multiple of 32");
    SrP W = {w1, b1, w2, b2};
    long tiles = B * (2 * R / SRT) * (2 * R / SRT);
    int nb = (int)(tiles < 2048 ? tiles : 2048);
    hipLaunchKernelGGL(sr_fused_fwd_kernel, dim3(nb), dim3(256), 0, stream, pred_img, big_src(nullptr, nullptr), (const long*)nullptr, (const long*)nullptr, W,
                       (float*)nullptr, (long)B, R, 0, sr);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// Workspace of ecamp_sr_bwd(..., gw_ws, ...): {dW1[81], db1[3], dW2[81], db2[3]} f32, zeroed by the caller.
extern "C" int64_t ecamp_sr_bwd_workspace_bytes(void) { return 168 * 4; }

// dsr: f32 [B,3,R,R] = d(0.5*res_sum)/d pred_img;  gw_ws[168] += {dW1[81], db1[3], dW2[81], db2[3]} (unscaled; the caller
// folds g_res*2/N in when adding into the .grad views).
extern "C" int ecamp_sr_bwd(const float* pred_img, const void* big, const float* big_lut, const int64_t* column,
                            const int64_t* row, const float* w1, const float* b1, const float* w2, const float* b2, float* dsr, float* gw_ws,
                            int64_t B, int32_t R, int32_t super_patch, int32_t window, int32_t mode, hipStream_t stream) {
    ECAMP_CHECK_ARG(pred_img && big && column && row && w1 && b1 && w2 && b2 && dsr && gw_ws, "sr_bwd: null pointer");
    ECAMP_CHECK_ARG(!big_lut || ((uintptr_t)big & 1) == 0, "sr_bwd: the uint8 target must be 2-byte aligned");
    const BigSrc bs = big_src(big, big_lut);
    ECAMP_CHECK_ARG(super_patch == SRT && (2 * R) % SRT == 0, "sr_bwd: the fused SR head is built for 32-px super-patches (patch 16)");
    ECAMP_CHECK_ARG(mode == 0 || mode == 1, "sr_bwd: mode must be 0 (f32 VALU) or 1 (bf16 matrix cores)");
    SrP W = {w1, b1, w2, b2};
    long tiles = B * (2 * R / SRT) * (2 * R / SRT);
    ECAMP_CHECK_ARG(tiles < (1L << 30), "sr_bwd: too many tiles");
    int nb = (int)(tiles < 1024 ? tiles : 1024);
    size_t shm = (size_t)(3 * (42 * 42 + 40 * 40 + 38 * 38 + 36 * 36)) * sizeof(float);
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sr_fused_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        once = true;
    }
    if (mode == 1) {
        static int nbp = 0;
        if (nbp == 0) {
            const char* nbe = getenv("ECAMP_SR_BLOCKS");   // development: grid override
            nbp = nbe && atoi(nbe) > 0 ? atoi(nbe) : 512;  // two resident workgroups per CU
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sr_pair_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SRP_LDS_BYTES);
        }
        const int nb2 = (int)(tiles < nbp ? tiles : nbp);
        hipLaunchKernelGGL(sr_pair_bwd_kernel, dim3(nb2), dim3(256), SRP_LDS_BYTES, stream, pred_img, bs, (const long*)column, (const long*)row, W, dsr,
                           gw_ws, (long)B, R, window);
        ECAMP_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(sr_fused_bwd_kernel, dim3(nb), dim3(256), shm, stream, pred_img, bs, (const long*)column, (const long*)row, W, dsr, gw_ws,
                       (long)B, R, window);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// param_grad[k] += scale_dev[idx] * ws[k]   (folds the unscaled 168-float SR gradients into the f32 .grad views)
__global__ void scaled_accum_kernel(const float* __restrict__ ws, float* __restrict__ grad, const float* __restrict__ scale_dev,
                                    int idx, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) grad[i] += scale_dev[idx] * ws[i];
}
extern "C" int ecamp_scaled_accum(const float* ws, float* grad, const float* scale_dev, int32_t idx, int32_t n, hipStream_t stream) {
    ECAMP_CHECK_ARG(ws && grad && scale_dev && n > 0, "scaled_accum: bad args");
    hipLaunchKernelGGL(scaled_accum_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, ws, grad, scale_dev, idx, n);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
