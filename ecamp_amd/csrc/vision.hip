// Image-side HBM-bound kernels of the ECAMP hot path:
//   bicubic 2x down-resize, MAE masking indices, im2col+gather of visible patches, token assembly,
//   decoder un-shuffle with mask-token fill, unpatchify + masked MSE, the super-resolution head
//   (bilinear x2 -> conv3x3 -> ReLU -> conv3x3 -> +skip -> ReLU) with its windowed MSE, and their backward.
// Masks are never materialised as pixel tensors (the reference kron()s them, model_ecamp.py:196-215):
// they are evaluated from mask[b, y/16, x/16] and the window bounds on the fly.
#include "common.h"

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
extern "C" int ecamp_bicubic_resize(const float* src, float* dst, int64_t planes, int32_t Hs, int32_t Ws, int32_t Hd,
                                    int32_t Wd, hipStream_t stream) {
    ECAMP_CHECK_ARG(src && dst && planes > 0, "bicubic: bad args");
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

template <typename T>
__global__ void sr_up_kernel(const float* __restrict__ pred_img, T* __restrict__ u, long planes, int R) {
    const int R2 = 2 * R;
    long n = planes * R2 * R2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int X = (int)(i % R2), Y = (int)((i / R2) % R2);
        long pl = i / ((long)R2 * R2);
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        up2_taps(Y, R, y0, y1, wy0, wy1);
        up2_taps(X, R, x0, x1, wx0, wx1);
        const float* p = pred_img + pl * (long)R * R;
        float v = wy0 * (wx0 * p[y0 * R + x0] + wx1 * p[y0 * R + x1]) + wy1 * (wx0 * p[y1 * R + x0] + wx1 * p[y1 * R + x1]);
        u[i] = from_f<T>(v);
    }
}

// out[b,o,Y,X] = (relu?)( bias[o] + sum_{i,ky,kx} w[o,i,ky,kx] * in[b,i,Y+ky-1,X+kx-1] )
template <typename T>
__device__ __forceinline__ void conv3_at(const T* __restrict__ in, long b, int Y, int X, int R2, const float* w, float (&acc)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const T* pl = in + (b * 3 + i) * (long)R2 * R2;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            int yy = Y + ky - 1;
            if (yy < 0 || yy >= R2) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                int xx = X + kx - 1;
                if (xx < 0 || xx >= R2) continue;
                float v = to_f<T>(pl[(long)yy * R2 + xx]);
#pragma unroll
                for (int o = 0; o < 3; ++o) acc[o] += w[((o * 3 + i) * 3 + ky) * 3 + kx] * v;
            }
        }
    }
}
// transposed conv: out[b,i,Y,X] = sum_{o,ky,kx} w[o,i,ky,kx] * g[b,o,Y-ky+1,X-kx+1]
template <typename T>
__device__ __forceinline__ void conv3t_at(const T* __restrict__ g, long b, int Y, int X, int R2, const float* w, float (&acc)[3]) {
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const T* pl = g + (b * 3 + o) * (long)R2 * R2;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            int yy = Y - ky + 1;
            if (yy < 0 || yy >= R2) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                int xx = X - kx + 1;
                if (xx < 0 || xx >= R2) continue;
                float v = to_f<T>(pl[(long)yy * R2 + xx]);
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[i] += w[((o * 3 + i) * 3 + ky) * 3 + kx] * v;
            }
        }
    }
}

template <typename T>
__global__ void sr_conv1_kernel(const T* __restrict__ u, T* __restrict__ c1, SrP P, long B, int R2) {
    __shared__ SrW W;
    load_srw(W, P);
    long n = B * (long)R2 * R2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int X = (int)(i % R2), Y = (int)((i / R2) % R2);
        long b = i / ((long)R2 * R2);
        float acc[3] = {W.b1[0], W.b1[1], W.b1[2]};
        conv3_at<T>(u, b, Y, X, R2, W.w1, acc);
#pragma unroll
        for (int o = 0; o < 3; ++o) c1[((b * 3 + o) * (long)R2 + Y) * R2 + X] = from_f<T>(fmaxf(acc[o], 0.f));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sr_conv2_loss_kernel(const T* __restrict__ u, const T* __restrict__ c1,
                                                            const float* __restrict__ big, const long* __restrict__ column,
                                                            const long* __restrict__ row, T* __restrict__ ds,
                                                            float* __restrict__ loss_sum, SrP P, long B, int R2, int sp, int win) {
    __shared__ float sh[4];
    __shared__ SrW W;
    load_srw(W, P);
    long n = B * (long)R2 * R2;
    float part = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int X = (int)(i % R2), Y = (int)((i / R2) % R2);
        long b = i / ((long)R2 * R2);
        int gy = Y / sp, gx = X / sp;
        int c0 = (int)column[b], r0 = (int)row[b];
        bool in = gy >= c0 && gy < c0 + win && gx >= r0 && gx < r0 + win;
        float acc[3] = {W.b2[0], W.b2[1], W.b2[2]};
        if (in) conv3_at<T>(c1, b, Y, X, R2, W.w2, acc);
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            long idx = ((b * 3 + o) * (long)R2 + Y) * R2 + X;
            float g = 0.f;
            if (in) {
                float s = fmaxf(acc[o] + to_f<T>(u[idx]), 0.f);
                float d = s - big[idx];
                part += d * d;
                g = s > 0.f ? d : 0.f;
            }
            ds[idx] = from_f<T>(g);
        }
    }
    part = block_sum_256(part, sh);
    if (threadIdx.x == 0) atomicAdd(loss_sum, part);
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

// dc1 = conv2^T(ds) * [c1 > 0] ; dW2[o,i,ky,kx] += ds[o,Y,X] * c1[i,Y+ky-1,X+kx-1] ; db2[o] += ds[o,Y,X]
template <typename T>
__global__ __launch_bounds__(256) void sr_bwd2_kernel(const T* __restrict__ ds, const T* __restrict__ c1, T* __restrict__ dc1,
                                                      float* __restrict__ dw2 /*[81]*/, float* __restrict__ db2 /*[3]*/, SrP P,
                                                      long B, int R2) {
    __shared__ float sh[4 * 84];
    __shared__ SrW W;
    load_srw(W, P);
    long n = B * (long)R2 * R2;
    float g[84];
#pragma unroll
    for (int k = 0; k < 84; ++k) g[k] = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int X = (int)(i % R2), Y = (int)((i / R2) % R2);
        long b = i / ((long)R2 * R2);
        float acc[3] = {0.f, 0.f, 0.f};
        conv3t_at<T>(ds, b, Y, X, R2, W.w2, acc);
        float d[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            long idx = ((b * 3 + o) * (long)R2 + Y) * R2 + X;
            dc1[idx] = from_f<T>(to_f<T>(c1[idx]) > 0.f ? acc[o] : 0.f);
            d[o] = to_f<T>(ds[idx]);
            g[81 + o] += d[o];
        }
        if (d[0] != 0.f || d[1] != 0.f || d[2] != 0.f) {
#pragma unroll
            for (int ii = 0; ii < 3; ++ii) {
                const T* pl = c1 + (b * 3 + ii) * (long)R2 * R2;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    int yy = Y + ky - 1;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        int xx = X + kx - 1;
                        float v = (yy >= 0 && yy < R2 && xx >= 0 && xx < R2) ? to_f<T>(pl[(long)yy * R2 + xx]) : 0.f;
#pragma unroll
                        for (int o = 0; o < 3; ++o) g[((o * 3 + ii) * 3 + ky) * 3 + kx] += d[o] * v;
                    }
                }
            }
        }
    }
    reduce_to_global<84>(g, dw2, sh);  // dw2[0..80], db2 must directly follow: dw2[81..83]
    (void)db2;
}

// du = ds + conv1^T(dc1) ; dW1[o,i,ky,kx] += dc1[o,Y,X] * u[i,Y+ky-1,X+kx-1] ; db1[o] += dc1[o,Y,X]
template <typename T>
__global__ __launch_bounds__(256) void sr_bwd1_kernel(const T* __restrict__ ds, const T* __restrict__ dc1, const T* __restrict__ u,
                                                      T* __restrict__ du, float* __restrict__ dw1 /*[84]*/, SrP P, long B, int R2) {
    __shared__ float sh[4 * 84];
    __shared__ SrW W;
    load_srw(W, P);
    long n = B * (long)R2 * R2;
    float g[84];
#pragma unroll
    for (int k = 0; k < 84; ++k) g[k] = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int X = (int)(i % R2), Y = (int)((i / R2) % R2);
        long b = i / ((long)R2 * R2);
        float acc[3] = {0.f, 0.f, 0.f};
        conv3t_at<T>(dc1, b, Y, X, R2, W.w1, acc);
        float d[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            long idx = ((b * 3 + o) * (long)R2 + Y) * R2 + X;
            du[idx] = from_f<T>(acc[o] + to_f<T>(ds[idx]));
            d[o] = to_f<T>(dc1[idx]);
            g[81 + o] += d[o];
        }
        if (d[0] != 0.f || d[1] != 0.f || d[2] != 0.f) {
#pragma unroll
            for (int ii = 0; ii < 3; ++ii) {
                const T* pl = u + (b * 3 + ii) * (long)R2 * R2;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    int yy = Y + ky - 1;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        int xx = X + kx - 1;
                        float v = (yy >= 0 && yy < R2 && xx >= 0 && xx < R2) ? to_f<T>(pl[(long)yy * R2 + xx]) : 0.f;
#pragma unroll
                        for (int o = 0; o < 3; ++o) g[((o * 3 + ii) * 3 + ky) * 3 + kx] += d[o] * v;
                    }
                }
            }
        }
    }
    reduce_to_global<84>(g, dw1, sh);
}

// dsr[b,c,y,x] = sum over the <=4x4 upsampled pixels whose bilinear footprint touches (y,x) (transpose of sr_up)
template <typename T>
__global__ void sr_up_bwd_kernel(const T* __restrict__ du, float* __restrict__ dsr, long planes, int R) {
    const int R2 = 2 * R;
    long n = planes * (long)R * R;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int x = (int)(i % R), y = (int)((i / R) % R);
        long pl = i / ((long)R * R);
        const T* p = du + pl * (long)R2 * R2;
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
                if (wx != 0.f) acc += wy * wx * to_f<T>(p[(long)Y * R2 + X]);
            }
        }
        dsr[i] = acc;
    }
}

extern "C" int ecamp_sr_fwd(const float* pred_img, const float* big, const int64_t* column, const int64_t* row,
                            const float* w1, const float* b1, const float* w2, const float* b2, void* u, void* c1, void* ds,
                            float* loss_sum, int64_t B, int32_t R, int32_t super_patch, int32_t window, int32_t dtype,
                            hipStream_t stream) {
    ECAMP_CHECK_ARG(pred_img && big && column && row && w1 && b1 && w2 && b2 && u && c1 && ds && loss_sum, "sr_fwd: null pointer");
    SrP W = {w1, b1, w2, b2};
    const int R2 = 2 * R;
    long n = B * 3 * (long)R2 * R2, npx = B * (long)R2 * R2;
    int nb = (int)((n + 255) / 256), nbp = (int)((npx + 255) / 256);
    if (nb > 16384) nb = 16384;
    if (nbp > 16384) nbp = 16384;
    int nbl = nbp > 4096 ? 4096 : nbp;
    if (dtype == ECAMP_F32) {
        hipLaunchKernelGGL(sr_up_kernel<float>, dim3(nb), dim3(256), 0, stream, pred_img, (float*)u, (long)B * 3, R);
        hipLaunchKernelGGL(sr_conv1_kernel<float>, dim3(nbp), dim3(256), 0, stream, (const float*)u, (float*)c1, W, (long)B, R2);
        hipLaunchKernelGGL(sr_conv2_loss_kernel<float>, dim3(nbl), dim3(256), 0, stream, (const float*)u, (const float*)c1, big, (const long*)column, (const long*)row, (float*)ds, loss_sum, W, (long)B, R2, super_patch, window);
    } else {
        hipLaunchKernelGGL(sr_up_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, pred_img, (bf16_t*)u, (long)B * 3, R);
        hipLaunchKernelGGL(sr_conv1_kernel<bf16_t>, dim3(nbp), dim3(256), 0, stream, (const bf16_t*)u, (bf16_t*)c1, W, (long)B, R2);
        hipLaunchKernelGGL(sr_conv2_loss_kernel<bf16_t>, dim3(nbl), dim3(256), 0, stream, (const bf16_t*)u, (const bf16_t*)c1, big, (const long*)column, (const long*)row, (bf16_t*)ds, loss_sum, W, (long)B, R2, super_patch, window);
    }
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// Produces dsr (f32 [B,3,R,R], unscaled: d(0.5*res_sum)/d pred_img) and accumulates the UNSCALED conv gradients
// into gw_ws[168] = {dW1[81], db1[3], dW2[81], db2[3]} (caller scales by g_res*2/N when folding into .grad).
extern "C" int ecamp_sr_bwd(const void* u, const void* c1, const void* ds, const float* w1, const float* b1, const float* w2,
                            const float* b2, void* dc1, void* du, float* dsr, float* gw_ws, int64_t B, int32_t R, int32_t dtype,
                            hipStream_t stream) {
    ECAMP_CHECK_ARG(u && c1 && ds && w1 && b1 && w2 && b2 && dc1 && du && dsr && gw_ws, "sr_bwd: null pointer");
    SrP W = {w1, b1, w2, b2};
    const int R2 = 2 * R;
    long npx = B * (long)R2 * R2, nlo = B * 3 * (long)R * R;
    int nbp = (int)((npx + 255) / 256);
    if (nbp > 2048) nbp = 2048;
    int nbl = (int)((nlo + 255) / 256);
    if (nbl > 8192) nbl = 8192;
    if (dtype == ECAMP_F32) {
        hipLaunchKernelGGL(sr_bwd2_kernel<float>, dim3(nbp), dim3(256), 0, stream, (const float*)ds, (const float*)c1, (float*)dc1, gw_ws + 84, gw_ws + 84 + 81, W, (long)B, R2);
        hipLaunchKernelGGL(sr_bwd1_kernel<float>, dim3(nbp), dim3(256), 0, stream, (const float*)ds, (const float*)dc1, (const float*)u, (float*)du, gw_ws, W, (long)B, R2);
        hipLaunchKernelGGL(sr_up_bwd_kernel<float>, dim3(nbl), dim3(256), 0, stream, (const float*)du, dsr, (long)B * 3, R);
    } else {
        hipLaunchKernelGGL(sr_bwd2_kernel<bf16_t>, dim3(nbp), dim3(256), 0, stream, (const bf16_t*)ds, (const bf16_t*)c1, (bf16_t*)dc1, gw_ws + 84, gw_ws + 84 + 81, W, (long)B, R2);
        hipLaunchKernelGGL(sr_bwd1_kernel<bf16_t>, dim3(nbp), dim3(256), 0, stream, (const bf16_t*)ds, (const bf16_t*)dc1, (const bf16_t*)u, (bf16_t*)du, gw_ws, W, (long)B, R2);
        hipLaunchKernelGGL(sr_up_bwd_kernel<bf16_t>, dim3(nbl), dim3(256), 0, stream, (const bf16_t*)du, dsr, (long)B * 3, R);
    }
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
