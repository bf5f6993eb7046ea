// Report-side HBM-bound kernels:
//   BertEmbeddings: word+position+token-type gather-sum -> LayerNorm(eps 1e-12) -> dropout, and its backward
//   (scatter-add into the f32 embedding-table gradients; the PAD row gets none: nn.Embedding(padding_idx=0));
//   weighted cross-entropy over the 30000-way MLM logits with the gradient written in place.
#include "common.h"

// one workgroup per sequence position s, waves stride over the batch -> the position-table gradient row is
// owned by exactly one workgroup (deterministic, no atomics); token-type / hot special-token / LN-affine
// gradients are reduced per workgroup and flushed with one atomic set.
template <typename T, int IT>
__global__ __launch_bounds__(256) void bert_embed_fwd_kernel(const long* __restrict__ ids, const long* __restrict__ type_ids,
                                                             const float* __restrict__ wword, const float* __restrict__ wpos,
                                                             const float* __restrict__ wtype, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, T* __restrict__ z, T* __restrict__ e,
                                                             float* __restrict__ mean, float* __restrict__ rstd, long B, int S,
                                                             int cols, float eps, float drop_p, uint64_t seed, uint64_t offset) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x;
    const int nv = cols >> 2;
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    for (long b = (long)blockIdx.y * 4 + wave; b < B; b += (long)gridDim.y * 4) {
        const long row = b * S + s;
        const long id = ids[row], ty = type_ids[row];
        float v[IT][4];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            int c = lane + 64 * i;
            if (c < nv) {
                float a[4], p[4], t[4];
                ld4<float>(wword + id * cols + c * 4, a);
                ld4<float>(wtype + ty * cols + c * 4, t);
                ld4<float>(wpos + (long)s * cols + c * 4, p);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i][r] = rnd<T>((a[r] + t[r]) + p[r]);
                st4<T>(z + row * cols + c * 4, v[i]);
                sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i][r] = 0.f;
            }
        }
        const float mu = wave_sum(sum) / (float)cols;
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
                float4 bb = *reinterpret_cast<const float4*>(beta + c * 4);
                float o[4] = {(v[i][0] - mu) * rs * g.x + bb.x, (v[i][1] - mu) * rs * g.y + bb.y,
                              (v[i][2] - mu) * rs * g.z + bb.z, (v[i][3] - mu) * rs * g.w + bb.w};
                if (drop_p > 0.f) {
                    float m[4];
                    dropout_scale4(seed, offset, (uint64_t)(row * nv + c), drop_p, inv_keep, m);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] *= m[r];
                }
                st4<T>(e + row * cols + c * 4, o);
            }
        }
    }
}

template <typename T, int IT>
__global__ __launch_bounds__(256) void bert_embed_bwd_kernel(const T* __restrict__ de, const T* __restrict__ z,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const long* __restrict__ ids,
                                                             const long* __restrict__ type_ids, float* __restrict__ gword,
                                                             float* __restrict__ gpos, float* __restrict__ gtype,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, long B, int S,
                                                             int cols, int pad_id, int hot0, int hot1, float drop_p,
                                                             uint64_t seed, uint64_t offset) {
    extern __shared__ __attribute__((aligned(16))) float sh[];  // [4][cols]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x;
    const int nv = cols >> 2;
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    // per-lane column partials: 0 dgamma, 1 dbeta, 2 pos row, 3/4 token types 0/1, 5/6 hot ids
    float acc[7][IT][4];
    float gm[IT][4];
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int i = 0; i < IT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[k][i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        int c = lane + 64 * i;
        if (c < nv) ld4<float>(gamma + c * 4, gm[i]);
        else gm[i][0] = gm[i][1] = gm[i][2] = gm[i][3] = 0.f;
    }
    for (long b = (long)blockIdx.y * 4 + wave; b < B; b += 4 * (long)gridDim.y) {   // gridDim.y workgroups share a position
        const long row = b * S + s;
        const long id = ids[row], ty = type_ids[row];
        const float mu = mean[row], rs = rstd[row];
        float xh[IT][4], g[IT][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            int c = lane + 64 * i;
            if (c < nv) {
                float d[4], zz[4];
                ld4<T>(de + row * cols + c * 4, d);
                ld4<T>(z + row * cols + c * 4, zz);
                if (drop_p > 0.f) {
                    float m[4];
                    dropout_scale4(seed, offset, (uint64_t)(row * nv + c), drop_p, inv_keep, m);
#pragma unroll
                    for (int r = 0; r < 4; ++r) d[r] *= m[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xh[i][r] = (zz[r] - mu) * rs;
                    g[i][r] = d[r] * gm[i][r];
                    s1 += g[i][r];
                    s2 += g[i][r] * xh[i][r];
                    acc[0][i][r] += d[r] * xh[i][r];
                    acc[1][i][r] += d[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) xh[i][r] = g[i][r] = 0.f;
            }
        }
        s1 = wave_sum(s1) / (float)cols;
        s2 = wave_sum(s2) / (float)cols;
        // A row of an ordinary token goes to its embedding row by atomics.  Issued from the 16-B-per-lane register layout an atomic instruction
        // touches 4 B in each of 64 chunks 16 B apart (eight 128-B lines a quarter full: 165 us per call at B = 256, the atomic units' line rate);
        // passed through this wave's LDS row first, lane l adds element 64 k + l: two full lines per instruction (round 6).
        const bool scatter = id != hot0 && id != hot1 && id != pad_id;   // wave-uniform (one row per wave)
        float* wrow = sh + wave * cols;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            int c = lane + 64 * i;
            if (c < nv) {
                float dz4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float dz = rs * (g[i][r] - s1 - xh[i][r] * s2);
                    dz4[r] = dz;
                    acc[2][i][r] += dz;
                    if (ty == 0) acc[3][i][r] += dz; else acc[4][i][r] += dz;
                    if (id == hot0) acc[5][i][r] += dz;
                    else if (id == hot1) acc[6][i][r] += dz;
                }
                if (scatter) *reinterpret_cast<float4*>(wrow + c * 4) = make_float4(dz4[0], dz4[1], dz4[2], dz4[3]);
            }
        }
        if (scatter) {
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes have landed (the row is private to the wave)
            __builtin_amdgcn_wave_barrier();
            float* dst = gword + id * cols;
            for (int c = lane; c < cols; c += 64) atomicAdd(dst + c, wrow[c]);
            __builtin_amdgcn_wave_barrier();       // the next row's writes stay behind these reads
        }
    }
    for (int k = 0; k < 7; ++k) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            int c = lane + 64 * i;
            if (c < nv) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = k == 0 ? acc[0][i][r] : k == 1 ? acc[1][i][r] : k == 2 ? acc[2][i][r] : k == 3 ? acc[3][i][r]
                              : k == 4 ? acc[4][i][r] : k == 5 ? acc[5][i][r] : acc[6][i][r];
                    sh[wave * cols + c * 4 + r] = v;
                }
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < cols; c += 256) {
            float t = sh[c] + sh[cols + c] + sh[2 * cols + c] + sh[3 * cols + c];
            if (k == 0) atomicAdd(dgamma + c, t);
            else if (k == 1) atomicAdd(dbeta + c, t);
            else if (k == 2) {
                if (gridDim.y == 1) gpos[(long)s * cols + c] += t;  // this workgroup owns position row s
                else atomicAdd(gpos + (long)s * cols + c, t);
            }
            else if (k == 3) atomicAdd(gtype + c, t);
            else if (k == 4) { if (t != 0.f) atomicAdd(gtype + cols + c, t); }
            else if (k == 5) { if (hot0 != pad_id && t != 0.f) atomicAdd(gword + (long)hot0 * cols + c, t); }
            else { if (hot1 != pad_id && t != 0.f) atomicAdd(gword + (long)hot1 * cols + c, t); }
        }
    }
}

extern "C" int ecamp_bert_embed_fwd(const int64_t* ids, const int64_t* type_ids, const float* word, const float* pos,
                                    const float* type, const float* gamma, const float* beta, void* z, void* e, float* mean,
                                    float* rstd, int64_t B, int32_t S, int32_t cols, float eps, float drop_p, uint64_t seed,
                                    uint64_t offset, int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(ids && type_ids && word && pos && type && gamma && beta && z && e && mean && rstd, "bert_embed_fwd: null pointer");
    ECAMP_CHECK_ARG(cols % 4 == 0 && cols <= 1024, "bert_embed_fwd: cols=%d must be a multiple of 4 and <= 1024", cols);
    int gy = (int)((B + 3) / 4);
    if (gy > 16) gy = 16;
    dim3 grid(S, gy), block(256);
    const int it = ceil_div(cols / 4, 64);
#define L(T_, IT_) hipLaunchKernelGGL((bert_embed_fwd_kernel<T_, IT_>), grid, block, 0, stream, (const long*)ids, (const long*)type_ids, word, pos, type, gamma, beta, (T_*)z, (T_*)e, mean, rstd, (long)B, S, cols, eps, drop_p, seed, offset)
    if (dtype == ECAMP_F32) { if (it <= 1) L(float, 1); else if (it <= 2) L(float, 2); else if (it <= 3) L(float, 3); else L(float, 4); }
    else { if (it <= 1) L(bf16_t, 1); else if (it <= 2) L(bf16_t, 2); else if (it <= 3) L(bf16_t, 3); else L(bf16_t, 4); }
#undef L
    ECAMP_LAUNCH_CHECK();
    return 0;
}

extern "C" int ecamp_bert_embed_bwd(const void* de, const void* z, const float* mean, const float* rstd, const float* gamma,
                                    const int64_t* ids, const int64_t* type_ids, float* gword, float* gpos, float* gtype,
                                    float* dgamma, float* dbeta, int64_t B, int32_t S, int32_t cols, int32_t pad_id,
                                    int32_t hot0, int32_t hot1, float drop_p, uint64_t seed, uint64_t offset, int32_t dtype,
                                    hipStream_t stream) {
    ECAMP_CHECK_ARG(de && z && mean && rstd && gamma && ids && type_ids && gword && gpos && gtype && dgamma && dbeta, "bert_embed_bwd: null pointer");
    ECAMP_CHECK_ARG(cols % 4 == 0 && cols <= 1024, "bert_embed_bwd: cols=%d must be a multiple of 4 and <= 1024", cols);
    // S workgroups alone leave half of the 256 CUs idle at S = 128: split the batch over gridDim.y workgroups per position
    int gy = (int)(B / 32);
    if (gy < 1) gy = 1;
    if (gy > 8) gy = 8;
    dim3 grid(S, gy), block(256);
    size_t shm = (size_t)4 * cols * sizeof(float);
    const int it = ceil_div(cols / 4, 64);
#define L(T_, IT_) hipLaunchKernelGGL((bert_embed_bwd_kernel<T_, IT_>), grid, block, shm, stream, (const T_*)de, (const T_*)z, mean, rstd, gamma, (const long*)ids, (const long*)type_ids, gword, gpos, gtype, dgamma, dbeta, (long)B, S, cols, pad_id, hot0, hot1, drop_p, seed, offset)
    if (dtype == ECAMP_F32) { if (it <= 1) L(float, 1); else if (it <= 2) L(float, 2); else if (it <= 3) L(float, 3); else L(float, 4); }
    else { if (it <= 1) L(bf16_t, 1); else if (it <= 2) L(bf16_t, 2); else if (it <= 3) L(bf16_t, 3); else L(bf16_t, 4); }
#undef L
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// weighted cross-entropy (bert_modeling.py:213-217): loss = mean_i( CE(logits_i, label_i) * w_i ) over ALL rows.
// One workgroup per row: pass 1 online max / sum-exp, pass 2 writes d loss / d logits in place:
//   dlogits[i, j] = w_i / M * (softmax_ij - [j == label_i])       loss_sum += w_i * (lse_i - logit[i, label_i])
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(T* __restrict__ logits, const long* __restrict__ labels,
                                                         const float* __restrict__ weights, float* __restrict__ loss_sum, int V,
                                                         long ld, float inv_count) {
    __shared__ float shm_[8];
    const long row = blockIdx.x;
    T* x = logits + row * ld;
    const int nv = V >> 2;
    float mx = -INFINITY, sm = 0.f;
    for (int c = threadIdx.x; c < nv; c += 256) {
        float p[4];
        ld4<T>(x + c * 4, p);
        float m4 = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3]));
        if (m4 > mx) {
            sm *= __expf(mx - m4);
            mx = m4;
        }
        sm += __expf(p[0] - mx) + __expf(p[1] - mx) + __expf(p[2] - mx) + __expf(p[3] - mx);
    }
    // block combine of (max, sum)
    float wm = wave_max(mx);
    sm = mx == -INFINITY ? 0.f : sm * __expf(mx - wm);   // a lane (or a whole wave) without elements: exp(-inf + inf) would be NaN
    sm = wave_sum(sm);
    if ((threadIdx.x & 63) == 0) {
        shm_[threadIdx.x >> 6] = wm;
        shm_[4 + (threadIdx.x >> 6)] = sm;
    }
    __syncthreads();
    float gmx = fmaxf(fmaxf(shm_[0], shm_[1]), fmaxf(shm_[2], shm_[3]));
    float gsm = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) gsm += shm_[k] == -INFINITY ? 0.f : shm_[4 + k] * __expf(shm_[k] - gmx);
    // a label outside [0, V) -- CrossEntropyLoss's ignore_index -100 (bert_modeling.py:212) -- contributes no loss and a zero
    // gradient row, and is never used as an index
    const long label = labels[row];
    const bool ign = label < 0 || label >= (long)V;
    const float w = ign ? 0.f : weights[row];
    if (threadIdx.x == 0 && !ign) {
        float lse = gmx + __logf(gsm);
        atomicAdd(loss_sum, w * (lse - to_f<T>(x[label])));
    }
    const float inv = 1.0f / gsm, sc = w * inv_count;
    __syncthreads();
    for (int c = threadIdx.x; c < nv; c += 256) {
        float p[4];
        ld4<T>(x + c * 4, p);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sftm = __expf(p[r] - gmx) * inv;
            p[r] = sc * (sftm - ((long)(c * 4 + r) == label ? 1.0f : 0.0f));
        }
        st4<T>(x + c * 4, p);
    }
}
// bf16 rows of up to 32768 logits: the row is read ONCE into registers (16 B per lane per load, NG loads per thread) and the
// gradient is written from them -- the two-pass kernel above reads every logit twice (2 GB per pass at B*S = 32768, V = 30000).
template <int NG>
__global__ __launch_bounds__(256) void ce_fwd_bwd_reg_kernel(bf16_t* __restrict__ logits, const long* __restrict__ labels,
                                                             const float* __restrict__ weights, float* __restrict__ loss_sum, int V,
                                                             long ld, float inv_count) {
    __shared__ float shm_[8];
    const long row = blockIdx.x;
    bf16_t* x = logits + row * ld;
    const int nv8 = V >> 3;
    uint4 q[NG];
    float mx = -INFINITY, sm = 0.f;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < nv8) {
            q[j] = *reinterpret_cast<const uint4*>(x + c * 8);
            const uint32_t w[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
            float p[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[2 * r] = h16_lo(w[r]);
                p[2 * r + 1] = h16_hi(w[r]);
            }
            const float m8 = fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), fmaxf(fmaxf(p[4], p[5]), fmaxf(p[6], p[7])));
            if (m8 > mx) {
                sm *= __expf(mx - m8);
                mx = m8;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) sm += __expf(p[r] - mx);
        }
    }
    float wm = wave_max(mx);
    sm = mx == -INFINITY ? 0.f : sm * __expf(mx - wm);   // a lane (or a whole wave) without elements: exp(-inf + inf) would be NaN
    sm = wave_sum(sm);
    if ((threadIdx.x & 63) == 0) {
        shm_[threadIdx.x >> 6] = wm;
        shm_[4 + (threadIdx.x >> 6)] = sm;
    }
    __syncthreads();
    const float gmx = fmaxf(fmaxf(shm_[0], shm_[1]), fmaxf(shm_[2], shm_[3]));
    float gsm = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) gsm += shm_[k] == -INFINITY ? 0.f : shm_[4 + k] * __expf(shm_[k] - gmx);
    const long label = labels[row];
    const bool ign = label < 0 || label >= (long)V;   // ignore_index (see the generic kernel)
    const float w = ign ? 0.f : weights[row];
    if (threadIdx.x == 0 && !ign) {
        float lse = gmx + __logf(gsm);
        atomicAdd(loss_sum, w * (lse - to_f<bf16_t>(x[label])));
    }
    const float inv = 1.0f / gsm, sc = w * inv_count;
    __syncthreads();   // thread 0 has read x[label] before anyone overwrites it
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < nv8) {
            const uint32_t wq[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
            float g[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                g[2 * r] = h16_lo(wq[r]);
                g[2 * r + 1] = h16_hi(wq[r]);
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) g[r] = sc * (__expf(g[r] - gmx) * inv - ((long)(c * 8 + r) == label ? 1.0f : 0.0f));
            uint4 o;
            o.x = pack_bf16x2(g[0], g[1]); o.y = pack_bf16x2(g[2], g[3]); o.z = pack_bf16x2(g[4], g[5]); o.w = pack_bf16x2(g[6], g[7]);
            *reinterpret_cast<uint4*>(x + c * 8) = o;
        }
    }
}
// Round 5: the same row-in-registers pass in THREE phases with ONE exp per logit (the kernel above evaluates two and rescales a running
// sum whenever its maximum moves: ~570 us of VALU time per 1.97 GB of logits against the ~500 us a pure read + write of them takes,
// profiles/r05_vocab_head.txt).  NT threads per row, NG 16-B loads per thread (V <= 8 NG NT):
//   1  load the row, maximum                                        -> block maximum
//   2  e = exp(x - max) kept in f32 registers over the loaded row, sum  -> block sum
//   3  d = e * (w / (M sum)), minus w / M at the label, packed, stored over the logits
// Lanes past the row hold -inf: they drop out of the maximum and add exp(-inf) = 0.  Measured directly behind the vocabulary GEMM
// (tools/vocab_head_probe.py, V = 30000): two-exp kernel 1.33x the time of an in-place multiply over the same bytes, this one with 512
// threads x 8 loads 1.25x, with 1024 x 4 (one row's 120 KB of e spread over all sixteen waves a CU's SIMDs take at 66 registers) 1.20-1.23x;
// forms that keep the row packed and evaluate exp twice (fewer registers, more rows per CU) 1.25-1.33x.
template <int NG, int NT>
__global__ __launch_bounds__(NT) void ce_fwd_bwd_row_kernel(bf16_t* __restrict__ logits, const long* __restrict__ labels,
                                                            const float* __restrict__ weights, float* __restrict__ loss_sum, int V,
                                                            long ld, float inv_count) {
    constexpr int NWV = NT / 64;
    __shared__ float shm_[2 * NWV];
    const long row = blockIdx.x;
    bf16_t* x = logits + row * ld;
    const int nv8 = V >> 3, wave = threadIdx.x >> 6;
    uint4 q[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int c = threadIdx.x + NT * j;
        q[j] = c < nv8 ? *reinterpret_cast<const uint4*>(x + c * 8) : make_uint4(H16_NEG_INF_X2, H16_NEG_INF_X2, H16_NEG_INF_X2, H16_NEG_INF_X2);
    }
    float e[NG][8];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const uint32_t w[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            e[j][2 * r] = h16_lo(w[r]);
            e[j][2 * r + 1] = h16_hi(w[r]);
            mx = fmaxf(mx, fmaxf(e[j][2 * r], e[j][2 * r + 1]));
        }
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) shm_[wave] = mx;
    __syncthreads();
    float gmx = shm_[0];
#pragma unroll
    for (int k = 1; k < NWV; ++k) gmx = fmaxf(gmx, shm_[k]);
    const long label = labels[row];
    const bool ign = label < 0 || label >= (long)V;   // ignore_index (see the generic kernel)
    const float w = ign ? 0.f : weights[row];
    const float xl = (threadIdx.x == 0 && !ign) ? to_f<bf16_t>(x[label]) : 0.f;   // read before anyone overwrites the row (barrier below)
    constexpr float LOG2E = 1.4426950408889634f;
    const float nb = -gmx * LOG2E;
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < NG; ++j)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            e[j][r] = __builtin_amdgcn_exp2f(fmaf(e[j][r], LOG2E, nb));
            sm += e[j][r];
        }
    sm = wave_sum(sm);
    if ((threadIdx.x & 63) == 0) shm_[NWV + wave] = sm;
    __syncthreads();
    float gsm = 0.f;
#pragma unroll
    for (int k = 0; k < NWV; ++k) gsm += shm_[NWV + k];
    if (threadIdx.x == 0 && !ign) atomicAdd(loss_sum, w * (gmx + __logf(gsm) - xl));
    const float sc = w * inv_count, f = sc / gsm;
    const int lg = ign ? -1 : (int)(label >> 3), lr = (int)(label & 7);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int c = threadIdx.x + NT * j;
        if (c < nv8) {
            float g[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) g[r] = e[j][r] * f;
            if (c == lg) {
#pragma unroll
                for (int r = 0; r < 8; ++r) g[r] -= r == lr ? sc : 0.f;
            }
            uint4 o;
            o.x = pack_bf16x2(g[0], g[1]); o.y = pack_bf16x2(g[2], g[3]); o.z = pack_bf16x2(g[4], g[5]); o.w = pack_bf16x2(g[6], g[7]);
            *reinterpret_cast<uint4*>(x + c * 8) = o;
        }
    }
}
extern "C" int ecamp_ce_fwd_bwd(void* logits, const int64_t* labels, const float* weights, float* loss_sum, int64_t M, int32_t V,
                                int64_t ld, float inv_count, int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(logits && labels && weights && loss_sum && V % 4 == 0 && ld % 4 == 0, "ce_fwd_bwd: bad args");
    dim3 grid((unsigned)M), block(256);
    if (dtype == ECAMP_BF16 && V % 8 == 0 && ld % 8 == 0 && V <= 32768 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0) {
        static const int ce_variant = getenv("ECAMP_CE_KERNEL") ? atoi(getenv("ECAMP_CE_KERNEL")) : 1;   // development A/B: 0 = the two-exp kernel of round 2
        if (ce_variant == 0) {
            if (V <= 16384) hipLaunchKernelGGL(ce_fwd_bwd_reg_kernel<8>, grid, block, 0, stream, (bf16_t*)logits, (const long*)labels, weights, loss_sum, V, (long)ld, inv_count);
            else hipLaunchKernelGGL(ce_fwd_bwd_reg_kernel<16>, grid, block, 0, stream, (bf16_t*)logits, (const long*)labels, weights, loss_sum, V, (long)ld, inv_count);
        } else {
#define CE_ROW(NG_, NT_) hipLaunchKernelGGL((ce_fwd_bwd_row_kernel<NG_, NT_>), grid, dim3(NT_), 0, stream, (bf16_t*)logits, (const long*)labels, weights, loss_sum, V, (long)ld, inv_count)
            if (V <= 4096) CE_ROW(1, 512); else if (V <= 8192) CE_ROW(2, 512); else if (V <= 16384) CE_ROW(4, 512); else CE_ROW(4, 1024);
#undef CE_ROW
        }
        ECAMP_LAUNCH_CHECK();
        return 0;
    }
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(ce_fwd_bwd_kernel<float>, grid, block, 0, stream, (float*)logits, (const long*)labels, weights, loss_sum, V, (long)ld, inv_count);
    else hipLaunchKernelGGL(ce_fwd_bwd_kernel<bf16_t>, grid, block, 0, stream, (bf16_t*)logits, (const long*)labels, weights, loss_sum, V, (long)ld, inv_count);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
