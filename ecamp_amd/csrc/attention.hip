// Multi-head attention forward / backward for the short sequences of the ECAMP hot path
// (SURVEY.md 2.3 K7 ViT MHSA T=50/197, K17 BERT self-attention S<=256 with key-padding mask and
// prob-dropout, K18 fusion cross-attention with 49 image tokens).
//
// Layout-agnostic: q/k/v/o are addressed as  base + b*sb + t*st + h*sh + d  (element strides), so the
// packed timm qkv buffer [B,T,3,H,hd] and HF's separate [B,S,H*hd] projections are read in place --
// no permute/transposes ever touch HBM.
//
// Structure (one workgroup = 4 waves = 64 query rows of one (batch, head); keys in chunks of 64):
//   S = (Q*scale) K^T on the matrix cores, K chunk staged in LDS (pitch hd+2 -> conflict-free B-operand reads)
//   softmax over the full key range held in accumulator registers (<= 256 keys), wave-shuffle row reductions
//   O = P V : P goes register(D layout) -> LDS -> A-operand, V chunk staged in LDS (pitch hd+16)
// All contractions use v_mfma_f32_16x16x4_f32 (exact f32), so one kernel serves the f32 parity mode and
// the bf16 mode (operands widened on load).  lse = max + log(sum) is saved for the backward pass, which
// recomputes P (no [T,T] tensor is ever written) and regenerates the dropout mask from Philox counters.
#include "attention.h"


// stage rows [r0, r0+64) x [0,HD) of a (b,h) slice into LDS as f32 with the given pitch (zeros past nrows)
template <typename T, int HD, int PITCH>
__device__ __forceinline__ void stage_rows(float* dst, const T* base, long st, int r0, int nrows, int tid) {
    constexpr int V4 = HD / 4;
#pragma unroll
    for (int i = 0; i < (64 * V4) / 256; ++i) {
        int idx = tid + 256 * i;
        int row = idx / V4, dv = idx % V4;
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        if (r0 + row < nrows) ld4<T>(base + (long)(r0 + row) * st + dv * 4, p);
        float* d = dst + row * PITCH + dv * 4;
        if (PITCH % 4 == 0) {
            *reinterpret_cast<float4*>(d) = make_float4(p[0], p[1], p[2], p[3]);
        } else {
            *reinterpret_cast<float2*>(d) = make_float2(p[0], p[1]);
            *reinterpret_cast<float2*>(d + 2) = make_float2(p[2], p[3]);
        }
    }
}

// A-operand fragments of 16 rows [r0+lrow] x HD, lane holds element d = 4*s + lk for s in [0, HD/4)
template <typename T, int HD>
__device__ __forceinline__ void load_rows_frag(float (&f)[HD / 4], const T* base, long st, int r0, int nrows, int lrow, int lk,
                                               float mul) {
    const bool ok = (r0 + lrow) < nrows;
    const T* p = base + (long)(r0 + lrow) * st + lk;
#pragma unroll
    for (int s = 0; s < HD / 4; ++s) f[s] = ok ? to_f<T>(p[s * 4]) * mul : 0.f;
}

// acc[t] += A(16 x HD, regs) * Bt(16 keys of tile t, HD)^T   with Bt rows in LDS (pitch P): acc[t][r] = C[4*lk+r][t*16+lrow]
template <int HD, int P>
__device__ __forceinline__ void mma_nt(f32x4 (&acc)[4], const float (&a)[HD / 4], const float* lds, int lrow, int lk) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float* b = lds + (t * 16 + lrow) * P + lk;
#pragma unroll
        for (int s = 0; s < HD / 4; ++s) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s * 4], acc[t], 0, 0, 0);
    }
}
// acc[dt] += A(16 x 64, LDS pitch PA, this wave's rows) * B(64 x HD, LDS pitch PB):  acc[dt][r] = C[4*lk+r][dt*16+lrow]
template <int HD, int PA, int PB>
__device__ __forceinline__ void mma_nn(f32x4 (&acc)[HD / 16], const float* la, const float* lb, int lrow, int lk) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        float a = la[lrow * PA + ks * 4 + lk];
        const float* b = lb + (ks * 4 + lk) * PB + lrow;
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[dt * 16], acc[dt], 0, 0, 0);
    }
}
// reduce over the 16 lanes that share lk (row reductions in the D layout)
__device__ __forceinline__ float row_max16(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float row_sum16(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#define PP 66  // pitch of the per-wave P / dS staging tiles (16 x 64)

// =============================================================================================
// forward
// =============================================================================================
template <typename T, int HD, int KCH>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
    constexpr int KP = HD + 2, VP = HD + 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* KV = smem;                 // 64 x VP
    float* PS = smem + 64 * VP;       // 4 x 16 x PP
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 15, lk = lane >> 4;
    const int b = blockIdx.y / a.H, h = blockIdx.y % a.H;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const T* qb = reinterpret_cast<const T*>(a.q) + b * a.q_sb + h * a.q_sh;
    const T* kb = reinterpret_cast<const T*>(a.k) + b * a.k_sb + h * a.k_sh;
    const T* vb = reinterpret_cast<const T*>(a.v) + b * a.v_sb + h * a.v_sh;
    T* ob = reinterpret_cast<T*>(a.o) + b * a.o_sb + h * a.o_sh;

    float qf[HD / 4];
    load_rows_frag<T, HD>(qf, qb, a.q_st, q0, a.Tq, lrow, lk, a.scale);

    f32x4 s[KCH * 4];
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        __syncthreads();
        stage_rows<T, HD, KP>(KV, kb, a.k_st, c * 64, a.Tk, tid);
        __syncthreads();
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_nt<HD, KP>(acc, qf, KV, lrow, lk);
#pragma unroll
        for (int t = 0; t < 4; ++t) s[c * 4 + t] = acc[t];
    }
    // mask + softmax (rows i = 4*lk + r, columns j = tile*16 + lrow)
    float mx[4] = {NEG_BIG, NEG_BIG, NEG_BIG, NEG_BIG};
#pragma unroll
    for (int t = 0; t < KCH * 4; ++t) {
        int j = t * 16 + lrow;
        bool ok = j < a.Tk && (a.key_mask == nullptr || a.key_mask[(long)b * a.Tk + j] != 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s[t][r] = ok ? s[t][r] : NEG_BIG;
            mx[r] = fmaxf(mx[r], s[t][r]);
        }
    }
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[r] = row_max16(mx[r]);
#pragma unroll
    for (int t = 0; t < KCH * 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float p = s[t][r] > 0.5f * NEG_BIG ? __expf(s[t][r] - mx[r]) : 0.f;
            s[t][r] = p;
            sum[r] += p;
        }
    const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sum[r] = row_sum16(sum[r]);
        int i = q0 + 4 * lk + r;
        if (lrow == 0 && i < a.Tq) a.lse[((long)b * a.H + h) * a.Tq + i] = mx[r] + __logf(sum[r]);
        sum[r] = 1.0f / sum[r];
    }
#pragma unroll
    for (int t = 0; t < KCH * 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float p = s[t][r] * sum[r];
            if (a.drop_p > 0.f) {
                uint64_t e = (((uint64_t)b * a.H + h) * a.Tq + (q0 + 4 * lk + r)) * (uint64_t)a.Tk + (t * 16 + lrow);
                p *= dropout_scale(a.seed, a.offset, e, a.drop_p, inv_keep);
            }
            s[t][r] = p;
        }
    // O = P V
    f32x4 o[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* ps = PS + wave * 16 * PP;
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        __syncthreads();
        stage_rows<T, HD, VP>(KV, vb, a.v_st, c * 64, a.Tk, tid);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) ps[(4 * lk + r) * PP + t * 16 + lrow] = s[c * 4 + t][r];
        __syncthreads();
        mma_nn<HD, PP, VP>(o, ps, KV, lrow, lk);
    }
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int i = q0 + 4 * lk + r;
            if (i < a.Tq) ob[(long)i * a.o_st + dt * 16 + lrow] = from_f<T>(o[dt][r]);
        }
}

// =============================================================================================
// backward, part 1: dQ (+ delta = rowsum(dO * O)), one workgroup per 64 query rows
// =============================================================================================
template <typename T, int HD, int KCH>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs a) {
    constexpr int KP = HD + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* KB = smem;                  // 64 x KP
    float* VB = smem + 64 * KP;        // 64 x KP
    float* PS = smem + 2 * 64 * KP;    // 4 x 16 x PP
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 15, lk = lane >> 4;
    const int b = blockIdx.y / a.H, h = blockIdx.y % a.H;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const long bh = (long)b * a.H + h;
    const T* qb = reinterpret_cast<const T*>(a.q) + b * a.q_sb + h * a.q_sh;
    const T* kb = reinterpret_cast<const T*>(a.k) + b * a.k_sb + h * a.k_sh;
    const T* vb = reinterpret_cast<const T*>(a.v) + b * a.v_sb + h * a.v_sh;
    const T* ob = reinterpret_cast<const T*>(a.o) + b * a.o_sb + h * a.o_sh;
    const T* gb = reinterpret_cast<const T*>(a.dout) + b * a.do_sb + h * a.do_sh;
    T* dqb = reinterpret_cast<T*>(a.dq) + b * a.dq_sb + h * a.dq_sh;

    float qf[HD / 4], gf[HD / 4];
    load_rows_frag<T, HD>(qf, qb, a.q_st, q0, a.Tq, lrow, lk, a.scale);
    load_rows_frag<T, HD>(gf, gb, a.do_st, q0, a.Tq, lrow, lk, 1.0f);
    // delta for row (q0 + lrow): sum_d dO*O ; lanes sharing lrow hold disjoint d
    float dl = 0.f;
    {
        const bool ok = (q0 + lrow) < a.Tq;
        const T* p = ob + (long)(q0 + lrow) * a.o_st + lk;
#pragma unroll
        for (int s = 0; s < HD / 4; ++s) dl += ok ? gf[s] * to_f<T>(p[s * 4]) : 0.f;
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        if (lk == 0 && ok) a.delta[bh * a.Tq + q0 + lrow] = dl;
    }
    float delta[4], lse[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        delta[r] = __shfl(dl, 4 * lk + r, 64);
        int i = q0 + 4 * lk + r;
        lse[r] = i < a.Tq ? a.lse[bh * a.Tq + i] : 0.f;
    }
    const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    f32x4 dq[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* ps = PS + wave * 16 * PP;
#pragma unroll 1
    for (int c = 0; c < KCH; ++c) {
        __syncthreads();
        stage_rows<T, HD, KP>(KB, kb, a.k_st, c * 64, a.Tk, tid);
        stage_rows<T, HD, KP>(VB, vb, a.v_st, c * 64, a.Tk, tid);
        __syncthreads();
        f32x4 s[4], dp[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) s[t] = dp[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_nt<HD, KP>(s, qf, KB, lrow, lk);
        mma_nt<HD, KP>(dp, gf, VB, lrow, lk);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int j = c * 64 + t * 16 + lrow;
            bool ok = j < a.Tk && (a.key_mask == nullptr || a.key_mask[(long)b * a.Tk + j] != 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int i = q0 + 4 * lk + r;
                float p = (ok && i < a.Tq) ? __expf(s[t][r] - lse[r]) : 0.f;
                float g = dp[t][r];
                if (a.drop_p > 0.f) {
                    uint64_t e = ((uint64_t)bh * a.Tq + i) * (uint64_t)a.Tk + j;
                    g *= dropout_scale(a.seed, a.offset, e, a.drop_p, inv_keep);
                }
                ps[(4 * lk + r) * PP + t * 16 + lrow] = p * (g - delta[r]);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes done before its own reads
        __builtin_amdgcn_wave_barrier();
        mma_nn<HD, PP, KP>(dq, ps, KB, lrow, lk);
    }
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int i = q0 + 4 * lk + r;
            if (i < a.Tq) dqb[(long)i * a.dq_st + dt * 16 + lrow] = from_f<T>(dq[dt][r] * a.scale);
        }
}

// =============================================================================================
// backward, part 2: dK, dV, one workgroup per 64 keys, looping over query chunks
// =============================================================================================
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnArgs a) {
    constexpr int KP = HD + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* QB = smem;                      // 64 x KP
    float* GB = smem + 64 * KP;            // 64 x KP   (dO)
    float* PS = smem + 2 * 64 * KP;        // 4 x 16 x PP  (P_dropped^T)
    float* DS = PS + 4 * 16 * PP;          // 4 x 16 x PP  (dS^T)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 15, lk = lane >> 4;
    const int b = blockIdx.y / a.H, h = blockIdx.y % a.H;
    const int j0 = blockIdx.x * 64 + wave * 16;
    const long bh = (long)b * a.H + h;
    const T* qb = reinterpret_cast<const T*>(a.q) + b * a.q_sb + h * a.q_sh;
    const T* kb = reinterpret_cast<const T*>(a.k) + b * a.k_sb + h * a.k_sh;
    const T* vb = reinterpret_cast<const T*>(a.v) + b * a.v_sb + h * a.v_sh;
    const T* gb = reinterpret_cast<const T*>(a.dout) + b * a.do_sb + h * a.do_sh;
    T* dkb = reinterpret_cast<T*>(a.dk) + b * a.dk_sb + h * a.dk_sh;
    T* dvb = reinterpret_cast<T*>(a.dv) + b * a.dv_sb + h * a.dv_sh;

    float kf[HD / 4], vf[HD / 4];
    load_rows_frag<T, HD>(kf, kb, a.k_st, j0, a.Tk, lrow, lk, a.scale);
    load_rows_frag<T, HD>(vf, vb, a.v_st, j0, a.Tk, lrow, lk, 1.0f);
    bool jok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int j = j0 + 4 * lk + r;
        jok[r] = j < a.Tk && (a.key_mask == nullptr || a.key_mask[(long)b * a.Tk + j] != 0);
    }
    const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    f32x4 dk[HD / 16], dv[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) dk[dt] = dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* ps = PS + wave * 16 * PP;
    float* ds = DS + wave * 16 * PP;
    const int nqc = (a.Tq + 63) / 64;
#pragma unroll 1
    for (int c = 0; c < nqc; ++c) {
        __syncthreads();
        stage_rows<T, HD, KP>(QB, qb, a.q_st, c * 64, a.Tq, tid);
        stage_rows<T, HD, KP>(GB, gb, a.do_st, c * 64, a.Tq, tid);
        __syncthreads();
        f32x4 st[4], dpt[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) st[t] = dpt[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma_nt<HD, KP>(st, kf, QB, lrow, lk);    // S^T[j][i]
        mma_nt<HD, KP>(dpt, vf, GB, lrow, lk);   // dP^T[j][i]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int i = c * 64 + t * 16 + lrow;
            bool iok = i < a.Tq;
            float lse = iok ? a.lse[bh * a.Tq + i] : 0.f;
            float dl = iok ? a.delta[bh * a.Tq + i] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = (iok && jok[r]) ? __expf(st[t][r] - lse) : 0.f;
                float m = 1.0f;
                if (a.drop_p > 0.f) {
                    uint64_t e = ((uint64_t)bh * a.Tq + i) * (uint64_t)a.Tk + (j0 + 4 * lk + r);
                    m = dropout_scale(a.seed, a.offset, e, a.drop_p, inv_keep);
                }
                ps[(4 * lk + r) * PP + t * 16 + lrow] = p * m;
                ds[(4 * lk + r) * PP + t * 16 + lrow] = p * (dpt[t][r] * m - dl);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        mma_nn<HD, PP, KP>(dv, ps, GB, lrow, lk);  // dV[j][d] += sum_i Pd^T[j][i] dO[i][d]
        mma_nn<HD, PP, KP>(dk, ds, QB, lrow, lk);  // dK[j][d] += sum_i dS^T[j][i] Q[i][d]
    }
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int j = j0 + 4 * lk + r;
            if (j < a.Tk) {
                dkb[(long)j * a.dk_st + dt * 16 + lrow] = from_f<T>(dk[dt][r] * a.scale);
                dvb[(long)j * a.dv_st + dt * 16 + lrow] = from_f<T>(dv[dt][r]);
            }
        }
}

// =============================================================================================
// Long key ranges in the exact-f32 path (compute_dtype = float32 at ViT-L/16 448^2: the decoder attends over 785 tokens,
// model_ecamp.py:240-264; timm Attention.forward).  The kernels above keep a query row's scores in accumulator registers
// (<= 256 keys); past that the parity mode uses these plain-f32 kernels: one wave per row, keys (or queries) 64 at a time,
// softmax in two passes over the keys (statistics, then probabilities from lse), FMA chains instead of MFMA.  Parity
// infrastructure of the product (exactness first); the bf16 production path has its own key-tiled online-softmax kernels.
// =============================================================================================
template <int HD>
__device__ __forceinline__ float dot_row(const float* __restrict__ a, const float* __restrict__ b) {
    float acc = 0.f;
#pragma unroll 8
    for (int d = 0; d < HD; ++d) acc = fmaf(a[d], b[d], acc);
    return acc;
}
template <int HD>
__global__ __launch_bounds__(256) void attn_long_fwd_kernel(AttnArgs a) {
    __shared__ float sq[4][HD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= (long)a.B * a.H * a.Tq) return;
    const int i = (int)(row % a.Tq), h = (int)((row / a.Tq) % a.H), b = (int)(row / ((long)a.Tq * a.H));
    const float* q = reinterpret_cast<const float*>(a.q) + b * a.q_sb + h * a.q_sh + (long)i * a.q_st;
    const float* kb = reinterpret_cast<const float*>(a.k) + b * a.k_sb + h * a.k_sh;
    const float* vb = reinterpret_cast<const float*>(a.v) + b * a.v_sb + h * a.v_sh;
    float* o = reinterpret_cast<float*>(a.o) + b * a.o_sb + h * a.o_sh + (long)i * a.o_st;
    for (int d = lane; d < HD; d += 64) sq[wave][d] = q[d] * a.scale;
    __builtin_amdgcn_wave_barrier();
    auto score = [&](int j) {
        const bool ok = j < a.Tk && (a.key_mask == nullptr || a.key_mask[(long)b * a.Tk + j] != 0);
        return ok ? dot_row<HD>(sq[wave], kb + (long)j * a.k_st) : NEG_BIG;
    };
    float mx = NEG_BIG, sm = 0.f;
    for (int j0 = 0; j0 < a.Tk; j0 += 64) {
        const float sc = score(j0 + lane);
        if (sc > 0.5f * NEG_BIG) {
            const float m2 = fmaxf(mx, sc);
            sm = sm * __expf(mx - m2) + __expf(sc - m2);
            mx = m2;
        }
    }
    const float wm = wave_max(mx);
    sm = mx > 0.5f * NEG_BIG ? sm * __expf(mx - wm) : 0.f;
    sm = wave_sum(sm);
    const float lse = wm + __logf(sm);
    if (lane == 0) a.lse[row] = lse;
    const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    float acc[(HD + 63) / 64];
#pragma unroll
    for (int t = 0; t < (HD + 63) / 64; ++t) acc[t] = 0.f;
    for (int j0 = 0; j0 < a.Tk; j0 += 64) {
        const int j = j0 + lane;
        const float sc = score(j);
        float p = sc > 0.5f * NEG_BIG ? __expf(sc - lse) : 0.f;
        if (a.drop_p > 0.f && j < a.Tk) p *= dropout_scale(a.seed, a.offset, (uint64_t)row * (uint64_t)a.Tk + j, a.drop_p, inv_keep);
        const int n = min(64, a.Tk - j0);
        for (int jj = 0; jj < n; ++jj) {
            const float pj = __shfl(p, jj, 64);
#pragma unroll
            for (int t = 0; t < (HD + 63) / 64; ++t) {
                const int d = lane + 64 * t;
                if (d < HD) acc[t] = fmaf(pj, vb[(long)(j0 + jj) * a.v_st + d], acc[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < (HD + 63) / 64; ++t) {
        const int d = lane + 64 * t;
        if (d < HD) o[d] = acc[t];
    }
}
// dQ of one query row per wave (+ delta[row] = dO . O for the dK/dV kernel)
template <int HD>
__global__ __launch_bounds__(256) void attn_long_bwd_dq_kernel(AttnArgs a) {
    __shared__ float sq[4][HD], sdo[4][HD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= (long)a.B * a.H * a.Tq) return;
    const int i = (int)(row % a.Tq), h = (int)((row / a.Tq) % a.H), b = (int)(row / ((long)a.Tq * a.H));
    const float* q = reinterpret_cast<const float*>(a.q) + b * a.q_sb + h * a.q_sh + (long)i * a.q_st;
    const float* kb = reinterpret_cast<const float*>(a.k) + b * a.k_sb + h * a.k_sh;
    const float* vb = reinterpret_cast<const float*>(a.v) + b * a.v_sb + h * a.v_sh;
    const float* o = reinterpret_cast<const float*>(a.o) + b * a.o_sb + h * a.o_sh + (long)i * a.o_st;
    const float* dO = reinterpret_cast<const float*>(a.dout) + b * a.do_sb + h * a.do_sh + (long)i * a.do_st;
    float* dq = reinterpret_cast<float*>(a.dq) + b * a.dq_sb + h * a.dq_sh + (long)i * a.dq_st;
    float part = 0.f;
    for (int d = lane; d < HD; d += 64) {
        sq[wave][d] = q[d] * a.scale;
        sdo[wave][d] = dO[d];
        part = fmaf(dO[d], o[d], part);
    }
    const float delta = wave_sum(part);
    if (lane == 0) a.delta[row] = delta;
    __builtin_amdgcn_wave_barrier();
    const float lse = a.lse[row];
    const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    float acc[(HD + 63) / 64];
#pragma unroll
    for (int t = 0; t < (HD + 63) / 64; ++t) acc[t] = 0.f;
    for (int j0 = 0; j0 < a.Tk; j0 += 64) {
        const int j = j0 + lane;
        float ds = 0.f;
        if (j < a.Tk && (a.key_mask == nullptr || a.key_mask[(long)b * a.Tk + j] != 0)) {
            const float p = __expf(dot_row<HD>(sq[wave], kb + (long)j * a.k_st) - lse);
            float dp = dot_row<HD>(sdo[wave], vb + (long)j * a.v_st);
            if (a.drop_p > 0.f) dp *= dropout_scale(a.seed, a.offset, (uint64_t)row * (uint64_t)a.Tk + j, a.drop_p, inv_keep);
            ds = p * (dp - delta);
        }
        const int n = min(64, a.Tk - j0);
        for (int jj = 0; jj < n; ++jj) {
            const float dj = __shfl(ds, jj, 64);
#pragma unroll
            for (int t = 0; t < (HD + 63) / 64; ++t) {
                const int d = lane + 64 * t;
                if (d < HD) acc[t] = fmaf(dj, kb[(long)(j0 + jj) * a.k_st + d], acc[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < (HD + 63) / 64; ++t) {
        const int d = lane + 64 * t;
        if (d < HD) dq[d] = acc[t] * a.scale;
    }
}
// dK and dV of one key row per wave
template <int HD>
__global__ __launch_bounds__(256) void attn_long_bwd_dkv_kernel(AttnArgs a) {
    __shared__ float sk[4][HD], sv[4][HD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long krow = (long)blockIdx.x * 4 + wave;
    if (krow >= (long)a.B * a.H * a.Tk) return;
    const int j = (int)(krow % a.Tk), h = (int)((krow / a.Tk) % a.H), b = (int)(krow / ((long)a.Tk * a.H));
    const float* qb = reinterpret_cast<const float*>(a.q) + b * a.q_sb + h * a.q_sh;
    const float* k = reinterpret_cast<const float*>(a.k) + b * a.k_sb + h * a.k_sh + (long)j * a.k_st;
    const float* v = reinterpret_cast<const float*>(a.v) + b * a.v_sb + h * a.v_sh + (long)j * a.v_st;
    const float* dOb = reinterpret_cast<const float*>(a.dout) + b * a.do_sb + h * a.do_sh;
    float* dk = reinterpret_cast<float*>(a.dk) + b * a.dk_sb + h * a.dk_sh + (long)j * a.dk_st;
    float* dv = reinterpret_cast<float*>(a.dv) + b * a.dv_sb + h * a.dv_sh + (long)j * a.dv_st;
    for (int d = lane; d < HD; d += 64) { sk[wave][d] = k[d]; sv[wave][d] = v[d]; }
    __builtin_amdgcn_wave_barrier();
    const bool kok = a.key_mask == nullptr || a.key_mask[(long)b * a.Tk + j] != 0;
    const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    float ak[(HD + 63) / 64], av[(HD + 63) / 64];
#pragma unroll
    for (int t = 0; t < (HD + 63) / 64; ++t) { ak[t] = 0.f; av[t] = 0.f; }
    const long r0 = ((long)b * a.H + h) * a.Tq;
    for (int i0 = 0; i0 < a.Tq; i0 += 64) {
        const int i = i0 + lane;
        float ds = 0.f, pz = 0.f;
        if (kok && i < a.Tq) {
            const float* q = qb + (long)i * a.q_st;
            const float p = __expf(dot_row<HD>(q, sk[wave]) * a.scale - a.lse[r0 + i]);
            float z = 1.f;
            if (a.drop_p > 0.f) z = dropout_scale(a.seed, a.offset, (uint64_t)(r0 + i) * (uint64_t)a.Tk + j, a.drop_p, inv_keep);
            const float dp = dot_row<HD>(dOb + (long)i * a.do_st, sv[wave]) * z;
            ds = p * (dp - a.delta[r0 + i]);
            pz = p * z;
        }
        const int n = min(64, a.Tq - i0);
        for (int ii = 0; ii < n; ++ii) {
            const float di = __shfl(ds, ii, 64), pi = __shfl(pz, ii, 64);
#pragma unroll
            for (int t = 0; t < (HD + 63) / 64; ++t) {
                const int d = lane + 64 * t;
                if (d < HD) {
                    ak[t] = fmaf(di, qb[(long)(i0 + ii) * a.q_st + d], ak[t]);
                    av[t] = fmaf(pi, dOb[(long)(i0 + ii) * a.do_st + d], av[t]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < (HD + 63) / 64; ++t) {
        const int d = lane + 64 * t;
        if (d < HD) { dk[d] = ak[t] * a.scale; dv[d] = av[t]; }
    }
}

// =============================================================================================
// host entries
// =============================================================================================
static int attn_check(const AttnArgs& a, int hd, int dtype, bool bwd) {
    ECAMP_CHECK_ARG(hd == 32 || hd == 64 || hd == 128, "attention: head_dim %d not in {32,64,128}", hd);
    ECAMP_CHECK_ARG(a.Tk >= 1 && a.Tq >= 1, "attention: empty sequence");
    ECAMP_CHECK_ARG(dtype == ECAMP_F32 || dtype == ECAMP_BF16, "attention: bad dtype");
    ECAMP_CHECK_ARG(a.drop_p >= 0.f && a.drop_p < 1.f, "attention: bad dropout p");
    const long m = dtype == ECAMP_BF16 ? 8 : 4;
    if (dtype == ECAMP_BF16) ECAMP_CHECK_ARG(a.q_st % m == 0 && a.q_sb % m == 0 && a.q_sh % m == 0 && a.o_st % 4 == 0 && a.o_sb % 4 == 0 && a.o_sh % 4 == 0,
                                             "attention(bf16): q strides must be multiples of 8, o strides of 4 elements");
    ECAMP_CHECK_ARG(a.k_st % m == 0 && a.k_sb % m == 0 && a.k_sh % m == 0 && a.v_st % m == 0 && a.v_sb % m == 0 && a.v_sh % m == 0,
                    "attention: k/v strides must be multiples of 4 (f32) / 8 (bf16) elements");
    if (bwd) ECAMP_CHECK_ARG(a.q_st % m == 0 && a.q_sb % m == 0 && a.q_sh % m == 0 && a.do_st % m == 0 && a.do_sb % m == 0 && a.do_sh % m == 0,
                             "attention: q/dO strides must be multiples of 4 elements");
    return 0;
}

// kernels that need more than the default 64 KiB of dynamic LDS opt in once per instantiation
template <typename K>
static void allow_lds(K kern, size_t bytes) {
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
#define LAUNCH_LDS(KERN, GRID, BLOCK, SHM, ST, ARGS)             \
    do {                                                         \
        static bool once_ = false;                               \
        if (!once_) { allow_lds(KERN, SHM); once_ = true; }      \
        hipLaunchKernelGGL(KERN, GRID, BLOCK, SHM, ST, ARGS);    \
    } while (0)

template <typename T, int HD>
static void fwd_dispatch(const AttnArgs& a, hipStream_t st) {
    if (a.Tk > 256) {   // f32 only (the bf16 path never comes here): one wave per query row
        hipLaunchKernelGGL((attn_long_fwd_kernel<HD>), dim3((unsigned)(((long)a.B * a.H * a.Tq + 3) / 4)), dim3(256), 0, st, a);
        return;
    }
    dim3 grid(ceil_div(a.Tq, 64), a.B * a.H), block(256);
    size_t shm = (size_t)(64 * (HD + 16) + 4 * 16 * PP) * sizeof(float);
    if (a.Tk <= 64) LAUNCH_LDS((attn_fwd_kernel<T, HD, 1>), grid, block, shm, st, a);
    else if (a.Tk <= 128) LAUNCH_LDS((attn_fwd_kernel<T, HD, 2>), grid, block, shm, st, a);
    else LAUNCH_LDS((attn_fwd_kernel<T, HD, 4>), grid, block, shm, st, a);
}
template <typename T, int HD>
static void bwd_dispatch(const AttnArgs& a, hipStream_t st) {
    if (a.Tk > 256) {
        hipLaunchKernelGGL((attn_long_bwd_dq_kernel<HD>), dim3((unsigned)(((long)a.B * a.H * a.Tq + 3) / 4)), dim3(256), 0, st, a);
        hipLaunchKernelGGL((attn_long_bwd_dkv_kernel<HD>), dim3((unsigned)(((long)a.B * a.H * a.Tk + 3) / 4)), dim3(256), 0, st, a);
        return;
    }
    dim3 grid(ceil_div(a.Tq, 64), a.B * a.H), block(256);
    size_t shm = (size_t)(2 * 64 * (HD + 2) + 4 * 16 * PP) * sizeof(float);
    if (a.Tk <= 64) LAUNCH_LDS((attn_bwd_dq_kernel<T, HD, 1>), grid, block, shm, st, a);
    else if (a.Tk <= 128) LAUNCH_LDS((attn_bwd_dq_kernel<T, HD, 2>), grid, block, shm, st, a);
    else LAUNCH_LDS((attn_bwd_dq_kernel<T, HD, 4>), grid, block, shm, st, a);
    dim3 grid2(ceil_div(a.Tk, 64), a.B * a.H);
    size_t shm2 = (size_t)(2 * 64 * (HD + 2) + 8 * 16 * PP) * sizeof(float);
    LAUNCH_LDS((attn_bwd_dkv_kernel<T, HD>), grid2, block, shm2, st, a);
}

extern "C" int64_t ecamp_attn_mask_bytes(int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t hd, int32_t dtype);
extern "C" int ecamp_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* key_mask,
                              int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t hd, const int64_t* q_strides,
                              const int64_t* k_strides, const int64_t* v_strides, const int64_t* o_strides, float scale,
                              float drop_p, uint64_t seed, uint64_t offset, int32_t dtype, void* drop_mask, hipStream_t stream) {
    ECAMP_CHECK_ARG(q && k && v && o && lse && q_strides && k_strides && v_strides && o_strides, "attn_fwd: null pointer");
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse; a.key_mask = key_mask;
    a.q_sb = q_strides[0]; a.q_st = q_strides[1]; a.q_sh = q_strides[2];
    a.k_sb = k_strides[0]; a.k_st = k_strides[1]; a.k_sh = k_strides[2];
    a.v_sb = v_strides[0]; a.v_st = v_strides[1]; a.v_sh = v_strides[2];
    a.o_sb = o_strides[0]; a.o_st = o_strides[1]; a.o_sh = o_strides[2];
    a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.scale = scale; a.drop_p = drop_p; a.seed = seed; a.offset = offset;
    a.drop_bits = (dtype == ECAMP_BF16 && drop_p > 0.f && ecamp_attn_mask_bytes(B, H, Tq, Tk, hd, dtype) > 0) ? reinterpret_cast<unsigned char*>(drop_mask) : nullptr;
    if (int rc = attn_check(a, hd, dtype, false)) return rc;
    const bool prof = ecamp_prof_active();
    if (prof) ecamp_prof_begin(ECAMP_PROF_ATTN, 4.0 * B * H * (double)Tq * Tk * hd, stream);
#define D(T_)                                                  \
    do {                                                       \
        if (hd == 32) fwd_dispatch<T_, 32>(a, stream);         \
        else if (hd == 64) fwd_dispatch<T_, 64>(a, stream);    \
        else fwd_dispatch<T_, 128>(a, stream);                 \
    } while (0)
    if (dtype == ECAMP_F32) D(float); else attn_bf16_fwd(a, hd, stream);
#undef D
    if (prof) ecamp_prof_end(stream);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// Bytes of the optional `drop_mask` buffer (32 per query row and head): 0 when the shape is not served by the head-resident bf16
// kernels under the current options -- the caller then passes NULL and every pass regenerates the mask from the Philox counters.
extern "C" int64_t ecamp_attn_mask_bytes(int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t hd, int32_t dtype) {
    if (dtype != ECAMP_BF16 || B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || Tk > 256) return 0;
    if (hd != 32 && hd != 64 && hd != 128) return 0;
    return attn_bf16_head_path(Tq, Tk, hd, false) ? (int64_t)B * H * Tq * 32 : 0;
}
// Workspace of ecamp_attn_bwd(..., delta_ws, ...): delta[b, h, i] = dO_i . O_i, one f32 per query row.
extern "C" int64_t ecamp_attn_bwd_workspace_bytes(int32_t B, int32_t H, int32_t Tq) { return (int64_t)B * H * Tq * 4; }

extern "C" int ecamp_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                              float* delta_ws, void* dq, void* dk, void* dv, const int32_t* key_mask, int32_t B, int32_t H,
                              int32_t Tq, int32_t Tk, int32_t hd, const int64_t* q_strides, const int64_t* k_strides,
                              const int64_t* v_strides, const int64_t* o_strides, const int64_t* do_strides,
                              const int64_t* dq_strides, const int64_t* dk_strides, const int64_t* dv_strides, float scale,
                              float drop_p, uint64_t seed, uint64_t offset, int32_t dtype, const void* drop_mask, hipStream_t stream) {
    ECAMP_CHECK_ARG(q && k && v && o && dout && lse && delta_ws && dq && dk && dv, "attn_bwd: null pointer");
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.v = v; a.o = const_cast<void*>(o); a.dout = dout; a.lse = const_cast<float*>(lse); a.delta = delta_ws;
    a.dq = dq; a.dk = dk; a.dv = dv; a.key_mask = key_mask;
    a.q_sb = q_strides[0]; a.q_st = q_strides[1]; a.q_sh = q_strides[2];
    a.k_sb = k_strides[0]; a.k_st = k_strides[1]; a.k_sh = k_strides[2];
    a.v_sb = v_strides[0]; a.v_st = v_strides[1]; a.v_sh = v_strides[2];
    a.o_sb = o_strides[0]; a.o_st = o_strides[1]; a.o_sh = o_strides[2];
    a.do_sb = do_strides[0]; a.do_st = do_strides[1]; a.do_sh = do_strides[2];
    a.dq_sb = dq_strides[0]; a.dq_st = dq_strides[1]; a.dq_sh = dq_strides[2];
    a.dk_sb = dk_strides[0]; a.dk_st = dk_strides[1]; a.dk_sh = dk_strides[2];
    a.dv_sb = dv_strides[0]; a.dv_st = dv_strides[1]; a.dv_sh = dv_strides[2];
    a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.scale = scale; a.drop_p = drop_p; a.seed = seed; a.offset = offset;
    a.drop_bits = (dtype == ECAMP_BF16 && drop_p > 0.f) ? reinterpret_cast<unsigned char*>(const_cast<void*>(drop_mask)) : nullptr;
    if (int rc = attn_check(a, hd, dtype, true)) return rc;
    const bool prof = ecamp_prof_active();
    if (prof) ecamp_prof_begin(ECAMP_PROF_ATTN, 8.0 * B * H * (double)Tq * Tk * hd, stream);
#define D(T_)                                                  \
    do {                                                       \
        if (hd == 32) bwd_dispatch<T_, 32>(a, stream);         \
        else if (hd == 64) bwd_dispatch<T_, 64>(a, stream);    \
        else bwd_dispatch<T_, 128>(a, stream);                 \
    } while (0)
    if (dtype == ECAMP_F32) D(float); else attn_bf16_bwd(a, hd, stream);
#undef D
    if (prof) ecamp_prof_end(stream);
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// =============================================================================================
// Attention PROBABILITIES (evaluation / visualisation only): probs[b,h,i,:] = softmax_j(scale * q_i . k_j + mask_j), f32.
// One wave per query row; lane owns keys lane, lane+64, ...  Not a training-path kernel: it exists because the reference's
// Visualization model returns the fusion layer's cross-attention probabilities (Visualization/module/context_fusion.py:45-57).
// =============================================================================================
template <typename T>
__global__ __launch_bounds__(256) void attn_probs_kernel(const T* __restrict__ q, const T* __restrict__ k, const int32_t* __restrict__ key_mask,
                                                         float* __restrict__ probs, int B, int H, int Tq, int Tk, int hd, long q_sb, long q_st,
                                                         long q_sh, long k_sb, long k_st, long k_sh, float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * H * Tq) return;
    const int i = (int)(row % Tq), h = (int)((row / Tq) % H), b = (int)(row / ((long)Tq * H));
    const T* qp = q + b * q_sb + i * q_st + h * q_sh;
    const T* kb = k + b * k_sb + h * k_sh;
    constexpr int MAXJ = 16;  // Tk <= 1024
    float s[MAXJ];
    float mx = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < MAXJ; ++jj) {
        const int j = lane + 64 * jj;
        s[jj] = -INFINITY;
        if (j < Tk) {
            const T* kp = kb + (long)j * k_st;
            float acc = 0.f;
            for (int d = 0; d < hd; ++d) acc += to_f<T>(qp[d]) * to_f<T>(kp[d]);
            acc *= scale;
            if (key_mask && key_mask[(long)b * Tk + j] == 0) acc += -3.4028234663852886e38f;  // + finfo(f32).min, as bert_modeling.py:92
            s[jj] = acc;
            mx = fmaxf(mx, acc);
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int jj = 0; jj < MAXJ; ++jj) {
        const int j = lane + 64 * jj;
        if (j < Tk) {
            s[jj] = __expf(s[jj] - mx);
            sum += s[jj];
        }
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    float* out = probs + row * Tk;
#pragma unroll
    for (int jj = 0; jj < MAXJ; ++jj) {
        const int j = lane + 64 * jj;
        if (j < Tk) out[j] = s[jj] * inv;
    }
}

extern "C" int ecamp_attn_probs(const void* q, const void* k, const int32_t* key_mask, float* probs, int32_t B, int32_t H, int32_t Tq,
                                int32_t Tk, int32_t hd, const int64_t* q_strides, const int64_t* k_strides, float scale, int32_t dtype,
                                hipStream_t stream) {
    ECAMP_CHECK_ARG(q && k && probs && q_strides && k_strides, "attn_probs: null pointer");
    ECAMP_CHECK_ARG(B > 0 && H > 0 && Tq > 0 && Tk > 0 && Tk <= 1024 && hd > 0, "attn_probs: bad shape (Tk <= 1024)");
    ECAMP_CHECK_ARG(dtype == ECAMP_F32 || dtype == ECAMP_BF16, "attn_probs: bad dtype %d", dtype);
    const long rows = (long)B * H * Tq;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == ECAMP_F32)
        hipLaunchKernelGGL(attn_probs_kernel<float>, grid, block, 0, stream, (const float*)q, (const float*)k, key_mask, probs, B, H, Tq, Tk, hd,
                           (long)q_strides[0], (long)q_strides[1], (long)q_strides[2], (long)k_strides[0], (long)k_strides[1], (long)k_strides[2], scale);
    else
        hipLaunchKernelGGL(attn_probs_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)q, (const bf16_t*)k, key_mask, probs, B, H, Tq, Tk, hd,
                           (long)q_strides[0], (long)q_strides[1], (long)q_strides[2], (long)k_strides[0], (long)k_strides[1], (long)k_strides[2], scale);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
