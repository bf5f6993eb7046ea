// Optimizer-side HBM-bound kernels over the FLAT parameter / gradient arenas:
// global gradient sum-of-squares (misc.py:280-292) and fused decoupled-weight-decay Adam (torch AdamW
// semantics, main_pretrain.py:254) that also refreshes the bf16 shadow copy the MFMA GEMMs read.
#include "common.h"

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n4, float* __restrict__ out) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) atomicAdd(out, acc);
}
extern "C" int ecamp_sumsq(const float* x, int64_t n, float* out, hipStream_t stream) {
    ECAMP_CHECK_ARG(x && out && n % 4 == 0, "ecamp_sumsq: n=%ld must be a multiple of 4", (long)n);
    long n4 = n / 4;
    int nb = (int)((n4 + 255) / 256);
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, stream, x, n4, out);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ p16, long n4, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1, float rsqrt_bc2, float gscale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float pp[4], gg[4], mm[4], vv[4];
        ld4<float>(p + i * 4, pp);
        ld4<float>(g + i * 4, gg);
        ld4<float>(m + i * 4, mm);
        ld4<float>(v + i * 4, vv);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float gr = gg[r] * gscale;
            pp[r] *= 1.0f - lr * wd;
            mm[r] = b1 * mm[r] + (1.0f - b1) * gr;
            vv[r] = b2 * vv[r] + (1.0f - b2) * gr * gr;
            float denom = sqrtf(vv[r]) * rsqrt_bc2 + eps;
            pp[r] -= (lr / bc1) * (mm[r] / denom);
        }
        st4<float>(p + i * 4, pp);
        st4<float>(m + i * 4, mm);
        st4<float>(v + i * 4, vv);
        if (p16) st4<bf16_t>(p16 + i * 4, pp);
    }
}
extern "C" int ecamp_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
                           float beta2, float eps, float weight_decay, int64_t step, float grad_scale, hipStream_t stream) {
    ECAMP_CHECK_ARG(p && g && m && v && n % 4 == 0 && step >= 1, "ecamp_adamw: bad args (n=%ld)", (long)n);
    long n4 = n / 4;
    int nb = (int)((n4 + 255) / 256);
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(nb), dim3(256), 0, stream, p, g, m, v, (bf16_t*)p_bf16, n4, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Whole-arena AdamW: one launch for every parameter of the model.  Parameters are padded to 64-element blocks in
// the arena; `block_group[i]` (uint8) names the param_group of block i (255 = frozen / unused -> skipped), and
// each group carries its own (lr, weight_decay) so timm's decay / no-decay split (main_pretrain.py:253) and
// `lr_scale` groups need no extra launches.
struct GroupHyper {
    float lr[8];
    float wd[8];
};
__global__ __launch_bounds__(256) void adamw_grouped_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v,
                                                            bf16_t* __restrict__ p16, const unsigned char* __restrict__ grp,
                                                            long n4, GroupHyper hp, float b1, float b2, float eps, float bc1,
                                                            float rsqrt_bc2, float gscale, float* __restrict__ sumsq,
                                                            const float* __restrict__ ctl) {
    __shared__ float ss_sh[4];
    if (ctl) {   // dynamic loss scaling decided on the device (loss_scale_update_kernel): an overflowed step leaves every byte alone
        if (ctl[1] != 0.f) return;
        gscale = ctl[0]; bc1 = ctl[2]; rsqrt_bc2 = ctl[3];
    }
    float ss = 0.f;   // sum of squares of the (unscaled) gradients this thread consumed: the global grad-norm for free
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const unsigned gi = grp[i >> 4];  // 16 float4 per 64-element block
        if (gi >= 8) continue;
        const float lr = hp.lr[gi], wd = hp.wd[gi];
        float pp[4], gg[4], mm[4], vv[4];
        ld4<float>(p + i * 4, pp);
        ld4<float>(g + i * 4, gg);
        ld4<float>(m + i * 4, mm);
        ld4<float>(v + i * 4, vv);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float gr = gg[r] * gscale;
            ss += gr * gr;
            pp[r] *= 1.0f - lr * wd;
            mm[r] = b1 * mm[r] + (1.0f - b1) * gr;
            vv[r] = b2 * vv[r] + (1.0f - b2) * gr * gr;
            float denom = sqrtf(vv[r]) * rsqrt_bc2 + eps;
            pp[r] -= (lr / bc1) * (mm[r] / denom);
        }
        st4<float>(p + i * 4, pp);
        st4<float>(m + i * 4, mm);
        st4<float>(v + i * 4, vv);
        if (p16) st4<bf16_t>(p16 + i * 4, pp);
    }
    if (sumsq) {
        ss = block_sum_256(ss, ss_sh);
        if (threadIdx.x == 0) atomicAdd(sumsq, ss);
    }
}
extern "C" int ecamp_adamw_grouped(float* p, const float* g, float* m, float* v, void* p_bf16, const uint8_t* block_group,
                                   int64_t n, int32_t ngroups, const float* lr_host, const float* wd_host, float beta1,
                                   float beta2, float eps, int64_t step, float grad_scale, float* grad_sumsq, const float* ctl,
                                   hipStream_t stream) {
    ECAMP_CHECK_ARG(p && g && m && v && block_group && lr_host && wd_host, "ecamp_adamw_grouped: null pointer");
    ECAMP_CHECK_ARG(n % 64 == 0 && ngroups >= 1 && ngroups <= 8 && step >= 1, "ecamp_adamw_grouped: bad args (n=%ld, groups=%d)", (long)n, ngroups);
    GroupHyper hp;
    for (int i = 0; i < 8; ++i) {
        hp.lr[i] = i < ngroups ? lr_host[i] : 0.f;
        hp.wd[i] = i < ngroups ? wd_host[i] : 0.f;
    }
    long n4 = n / 4;
    int nb = (int)((n4 + 255) / 256);
    if (nb > 8192) nb = 8192;
    double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_grouped_kernel, dim3(nb), dim3(256), 0, stream, p, g, m, v, (bf16_t*)p_bf16, block_group, n4, hp, beta1,
                       beta2, eps, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale, grad_sumsq, ctl);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// torch.cuda.amp.GradScaler's unscale_ / step / update (util/misc.py:262-269; torch 1.13.1 grad_scaler.py, _amp_update_scale_) as ONE
// single-thread kernel, so that the host never reads the overflow flag: `sumsq` is sum(g^2) over the SCALED gradients (inf / nan when
// any element overflowed).  state = {scale, growth tracker, skipped steps, -}; opt_step = AdamW's count of steps actually taken;
// ctl = {1 / scale of THIS step, skip flag, 1 - beta1^step, 1 / sqrt(1 - beta2^step)} is what ecamp_adamw_grouped reads.  (The counters are
// f32: exact to 2^24 = 16.7 M steps, fifteen times the reference's 800-epoch schedule.)
__global__ void loss_scale_update_kernel(const float* __restrict__ sumsq, float* __restrict__ state, float* __restrict__ opt_step,
                                         float* __restrict__ ctl, float* __restrict__ norm_out, float growth, float backoff, float interval,
                                         float b1, float b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float s = sumsq[0], scale = state[0], inv = 1.0f / scale;
    const bool found = !(fabsf(s) <= 3.402823466e+38f);   // inf or nan
    if (norm_out) norm_out[0] = sqrtf(s) * inv;             // the norm of the un-scaled gradients; inf / nan on overflow, like the reference's
    ctl[0] = inv;
    ctl[1] = found ? 1.f : 0.f;
    if (found) {
        state[0] = scale * backoff;
        state[1] = 0.f;
        state[2] += 1.f;
    } else {
        const float ok = state[1] + 1.f;
        if (ok == interval) { state[0] = scale * growth; state[1] = 0.f; } else state[1] = ok;
        const float step = opt_step[0] + 1.f;
        opt_step[0] = step;
        ctl[2] = (float)(1.0 - pow((double)b1, (double)step));
        ctl[3] = (float)(1.0 / sqrt(1.0 - pow((double)b2, (double)step)));
    }
}
extern "C" int ecamp_loss_scale_update(const float* sumsq, float* state, float* opt_step, float* ctl, float* norm_out, float growth_factor,
                                       float backoff_factor, int32_t growth_interval, float beta1, float beta2, hipStream_t stream) {
    ECAMP_CHECK_ARG(sumsq && state && opt_step && ctl, "ecamp_loss_scale_update: null pointer");
    ECAMP_CHECK_ARG(growth_factor > 1.f && backoff_factor > 0.f && backoff_factor < 1.f && growth_interval >= 1, "ecamp_loss_scale_update: bad factors");
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, stream, sumsq, state, opt_step, ctl, norm_out, growth_factor, backoff_factor,
                       (float)growth_interval, beta1, beta2);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
