// Shared by the GEMM kernels (gemm.hip, gemm_q8.h) and by tools/gemm_lab: the argument block and the scalar epilogue.
#pragma once
#include "common.h"

struct GemmArgs {
    const void* A;
    const void* B;
    void* C;
    int M, N, K;
    long lda, ldb, ldc;
    const float* bias;      // [N] f32 or null
    const void* residual;   // T [M, ldr] or null
    long ldr;
    void* pre_out;          // T [M, ldp]: value before activation (saved for GELU backward) or null
    long ldp;
    const void* gmul;       // T [M, ldg]: multiply result by gelu'(gmul[m,n]) (dgrad through GELU) or null
    long ldg;
    int act;                // 0 none, 1 exact GELU, 2 exact GELU with the SAVED DERIVATIVE: pre_out receives gelu'(pre) (forward), gmul holds gelu' itself (data gradient)
    int out_f32;            // C is f32 regardless of operand type
    int accumulate;         // C += result (requires an f32 output); exclusive ownership -> plain read-modify-write
    float* rowsum;          // optional f32 [M]: rowsum[m] += alpha * sum_k opA[m,k]  (bias gradient inside the wgrad GEMM)
    float* partial;         // split-K: f32 slabs [gridDim.z][M*ldc-equivalent dense M x N] written with plain stores
    int k_per_split;        // multiple of the K tile; grid.z = number of splits
    int nbm, nbn;
    float alpha;            // result scale applied to the accumulator before the epilogue
    float alpha_out;        // the caller's alpha / alpha_dev, kept for the row-sum even when split-K resets the tile's own scale
    const float* alpha_dev_out;
    const float* alpha_dev; // optional device scalar multiplied into alpha (upstream loss gradient; avoids a host sync)
    const float* alpha_dev2; // e4m3 form: the second operand's per-tensor scale (the first is alpha_dev)
    void* q8_out;            // e4m3 form, GELU epilogue: a third output, the e4m3 copy of C [M, ldc bytes per row] for the NEXT GEMM ...
    const float* q8_scale;   // ... quantised with that GEMM's input scale (delayed scaling) ...
    float* q8_amax;          // ... and max|C| of this launch into its 16 amax slots
    int dbg;                // development switches of the P8 kernel (ECAMP_P8_DBG); 0 in production
    int wide;               // every [M, ld] epilogue operand is 16-B aligned at 8-column granularity (P8's 16-B epilogue)
    int nsplit;             // P8: number of split-K slices (the persistent kernel walks tiles x slices itself)
};

// ---- grouped weight gradients (gemm_q8.h, ITEMS form): several dW_p (+)= dY_p^T X_p sharing the contraction length (the four linear
// layers of one transformer block) as ONE launch whose work items are (tile, K range) pieces cut so that every workgroup gets the
// same number of K tiles (ranges run across tile boundaries).  Each piece writes a dense 256 x 256 f32 slab; a grouped reduce sums
// the consecutive slabs of a tile into C.
struct Q8Prob {
    const void* A;            // dY [K, lda]  (rows = contraction)
    const void* B;            // X  [K, ldb]
    long lda, ldb;
    int M, N;                 // output [M, N] = [out features, in features]
    float* rowsum;            // bias gradient [M] or null
    const float* alpha_dev_out;
    float alpha_out;
    int pad;
};
struct Q8ItemRec {            // 32 bytes, read with scalar loads
    int prob, m0, n0, kbeg, kend, slab, flags, pad;   // flags bit 0: this piece also sums the bias gradient of rows m0.. (first N-tile column)
};
struct Q8Group {
    const Q8ItemRec* items;
    const int* wg_first;      // [workgroups + 1]: item range of each workgroup
    float* slabs;             // [items][256][256] f32
    int nprob, pad;
    Q8Prob p[4];
};

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    // Blocks are dealt round-robin to the 8 XCDs; give every XCD a contiguous range of tiles so that
    // neighbouring tiles (same A rows, different weight columns) share one L2.  Bijective for any nblk.
    int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <typename T>
__device__ __forceinline__ void epilogue4(const GemmArgs& g, int m, int n0, f32x4 acc, int z) {
    if (m >= g.M || n0 >= g.N) return;
    const float al = g.alpha_dev ? g.alpha * g.alpha_dev[0] : g.alpha;
    float v[4] = {acc[0] * al, acc[1] * al, acc[2] * al, acc[3] * al};
    if (g.bias) {
        float4 b = *reinterpret_cast<const float4*>(g.bias + n0);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (g.act == 2 && g.pre_out) {   // "saved derivative" (bf16 only, host-checked): pre_out receives gelu'(x) of the rounded pre-activation x
        float d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_both_fast_f(rnd<T>(v[r]), d[r]);
        st4<T>(reinterpret_cast<T*>(g.pre_out) + (long)m * g.ldp + n0, d);
    } else {
        if (g.pre_out) st4<T>(reinterpret_cast<T*>(g.pre_out) + (long)m * g.ldp + n0, v);
        if (g.act == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_t<T>(g.pre_out ? rnd<T>(v[r]) : v[r]);
        }
    }
    if (g.gmul) {
        float p[4];
        ld4<T>(reinterpret_cast<const T*>(g.gmul) + (long)m * g.ldg + n0, p);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= (g.act == 2 ? p[r] : gelu_grad_t<T>(p[r]));   // act == 2: gmul already holds gelu'
    }
    if (g.residual) {
        float p[4];
        ld4<T>(reinterpret_cast<const T*>(g.residual) + (long)m * g.ldr + n0, p);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += p[r];
    }
    if (g.partial) {  // split-K slab of this z-slice: reduced (and scaled / accumulated) by splitk_reduce_kernel
        st4<float>(g.partial + ((long)z * g.M + m) * g.N + n0, v);
    } else if (g.out_f32) {
        float* c = reinterpret_cast<float*>(g.C) + (long)m * g.ldc + n0;
        if (g.accumulate) {  // each output element is owned by exactly one thread of one block: no atomics needed
            float o[4];
            ld4<float>(c, o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += o[r];
        }
        st4<float>(c, v);
    } else {
        st4<T>(reinterpret_cast<T*>(g.C) + (long)m * g.ldc + n0, v);
    }
}


// 8-element (one lane of a 16-B bf16 / 32-B f32 access) load / store
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&o)[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&o)[8]) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&o)[8]);
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&o)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
}
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float (&o)[8]) {
    uint4 v;
    v.x = pack_bf16x2(o[0], o[1]);
    v.y = pack_bf16x2(o[2], o[3]);
    v.z = pack_bf16x2(o[4], o[5]);
    v.w = pack_bf16x2(o[6], o[7]);
    *reinterpret_cast<uint4*>(p) = v;
}

