// GEMM family for the ECAMP hot path (SURVEY.md 2.3 K2,K6,K8,K9,K11,K15,K16,K19,K20 and all dgrad/wgrad).
//
//   C[M,N] (+)= epilogue( sum_k opA[m,k] * opB[k,n] )
//
// Each operand is described by a "contraction-contiguous" flag and a leading dimension:
//   a_kc=1: opA[m,k] = A[m*lda + k]      a_kc=0: opA[m,k] = A[k*lda + m]
//   b_kc=1: opB[k,n] = B[n*ldb + k]      b_kc=0: opB[k,n] = B[k*ldb + n]
// so one kernel family serves  fwd  Y = X W^T      (a_kc=1, b_kc=1)
//                              dgrad dX = dY W     (a_kc=1, b_kc=0)
//                              wgrad dW = dY^T X   (a_kc=0, b_kc=0, split-K + f32 atomics)
// without any transposed copy in HBM: strided operands are transposed in registers on their way to LDS.
//
// bf16 path: 128x128x64 block tile, 4 waves (2x2) of 64x64, v_mfma_f32_16x16x32_bf16, LDS rows of 128 B
// with a 16-B-chunk XOR swizzle; register-staged prefetch of the next K tile.  MFMA operands are swapped
// (weight/N fragment as A, activation/M fragment as B) so each lane owns 4 *consecutive output columns*
// and the epilogue stores 8 B (bf16) / 16 B (f32) per lane instead of scattered 2-B stores.
// f32 path (parity mode): same tiling with BK=16 on v_mfma_f32_16x16x4_f32 (exact f32 fma chain).
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
    int act;                // 0 none, 1 exact GELU
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
};

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    // Blocks are dealt round-robin to the 8 XCDs; give every XCD a contiguous range of tiles so that
    // neighbouring tiles (same A rows, different weight columns) share one L2.  Bijective for any nblk.
    int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <typename T>
__device__ __forceinline__ void epilogue4(const GemmArgs& g, int m, int n0, f32x4 acc) {
    if (m >= g.M || n0 >= g.N) return;
    const float al = g.alpha_dev ? g.alpha * g.alpha_dev[0] : g.alpha;
    float v[4] = {acc[0] * al, acc[1] * al, acc[2] * al, acc[3] * al};
    if (g.bias) {
        float4 b = *reinterpret_cast<const float4*>(g.bias + n0);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (g.pre_out) st4<T>(reinterpret_cast<T*>(g.pre_out) + (long)m * g.ldp + n0, v);
    if (g.act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_f(g.pre_out ? rnd<T>(v[r]) : v[r]);
    }
    if (g.gmul) {
        float p[4];
        ld4<T>(reinterpret_cast<const T*>(g.gmul) + (long)m * g.ldg + n0, p);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= gelu_grad_f(p[r]);
    }
    if (g.residual) {
        float p[4];
        ld4<T>(reinterpret_cast<const T*>(g.residual) + (long)m * g.ldr + n0, p);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += p[r];
    }
    if (g.partial) {  // split-K slab of this z-slice: reduced (and scaled / accumulated) by splitk_reduce_kernel
        st4<float>(g.partial + ((long)blockIdx.z * g.M + m) * g.N + n0, v);
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

// =============================================================================================
// bf16
// =============================================================================================
#define BM 128
#define BN 128
#define BK 64

// 128-B rows: two rows share one 256-B LDS bank row, so the 16-B chunk index is XORed with row>>1 -- the 16 rows of an MFMA
// fragment then land on 16 distinct 16-B slots for every b128 lane group (conflict-free reads and writes)
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// global -> registers, contraction-contiguous operand: 4 x 16 B per thread (row = c/8, chunk = c%8)
__device__ __forceinline__ void g2r_kc(const bf16_t* P, long ld, int row0, int nrows, int k0, int kend, int tid,
                                       uint4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + 256 * i;
        int row = c >> 3, kc = c & 7;
        int gr = row0 + row, gk = k0 + kc * 8;
        r[i] = (gr < nrows && gk < kend) ? *reinterpret_cast<const uint4*>(P + (long)gr * ld + gk) : make_uint4(0, 0, 0, 0);
    }
}
__device__ __forceinline__ void r2s_kc(unsigned char* lds, int tid, const uint4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + 256 * i;
        int row = c >> 3, kc = c & 7;
        *reinterpret_cast<uint4*>(lds + row * 128 + ((kc ^ swz(row)) << 4)) = r[i];
    }
}
// Output-contiguous operand P[k*ld + row] (weights in dgrad, both activations in wgrad): the tile is kept in LDS exactly
// as it lies in HBM -- 64 contraction rows x 128 outputs, 16-B loads / ds_write_b128, no register transposes -- and the
// MFMA fragments (8 consecutive k for one output) come from the gfx950 LDS transpose read ds_read_b64_tr_b16:
// in a 16-lane group, lane i = 4r+q points at row r, columns 4q..4q+3 of a 4x16 block and RECEIVES column i of it
// (semantics probed on hardware: tools/probes/tr16_probe.hip).  Row pitch 288 B keeps the 4 rows of a block on
// disjoint banks.
#define OC_PITCH 288
typedef __attribute__((ext_vector_type(4))) short v4s16;

__device__ __forceinline__ void g2r_oc(const bf16_t* P, long ld, int row0, int nrows, int k0, int kend, int tid,
                                       uint4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + 256 * i;
        int kr = c >> 4, ch = c & 15;
        int gk = k0 + kr, gr = row0 + ch * 8;
        r[i] = (gr < nrows && gk < kend) ? *reinterpret_cast<const uint4*>(P + (long)gk * ld + gr) : make_uint4(0, 0, 0, 0);
    }
}
__device__ __forceinline__ void r2s_oc(unsigned char* lds, int tid, const uint4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + 256 * i;
        int kr = c >> 4, ch = c & 15;
        *reinterpret_cast<uint4*>(lds + kr * OC_PITCH + ch * 16) = r[i];
    }
}
// fragment of 16 outputs [o0, o0+16) x 8 contraction steps starting at kb + 8*(lane>>4)
__device__ __forceinline__ bf16x8 frag_oc(const unsigned char* lds, int o0, int kb, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const unsigned char* p = lds + (kb + g * 8 + (i >> 2)) * OC_PITCH + (o0 + (i & 3) * 4) * 2;
    v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s16 __attribute__((address_space(3)))*)(p));
    v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s16 __attribute__((address_space(3)))*)(p + 4 * OC_PITCH));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ bf16x8 frag_kc(const unsigned char* lds, int row, int ch) {
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((ch ^ swz(row)) << 4));
}

template <bool A_KC, bool B_KC, bool ROWSUM>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
    constexpr int TILE_BYTES = 64 * OC_PITCH;  // >= BM*BK*2: one size fits both operand layouts
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILE_BYTES];
    unsigned char* ldsA = lds;
    unsigned char* ldsB = lds + TILE_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = xcd_remap(blockIdx.x, g.nbm * g.nbn);
    const int m0 = (wg / g.nbn) * BM, n0 = (wg % g.nbn) * BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // bias gradient for free: blocks of the first N-tile column also multiply the M-side fragments by an all-ones fragment
    // ROWSUM is a compile-time flag: its 16 extra accumulators would cost every other GEMM form a wave of occupancy
    const bool do_rowsum = ROWSUM && g.rowsum != nullptr && (wg % g.nbn) == 0 && (wave & 1) == 0;
    f32x4 accr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accr[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

    uint4 ra_kc[4], rb_kc[4];
    uint4 ra_oc[4], rb_oc[4];

    if (A_KC) g2r_kc(A, g.lda, m0, g.M, kbeg, kend, tid, ra_kc); else g2r_oc(A, g.lda, m0, g.M, kbeg, kend, tid, ra_oc);
    if (B_KC) g2r_kc(B, g.ldb, n0, g.N, kbeg, kend, tid, rb_kc); else g2r_oc(B, g.ldb, n0, g.N, kbeg, kend, tid, rb_oc);
    if (A_KC) r2s_kc(ldsA, tid, ra_kc); else r2s_oc(ldsA, tid, ra_oc);
    if (B_KC) r2s_kc(ldsB, tid, rb_kc); else r2s_oc(ldsB, tid, rb_oc);
    __syncthreads();

    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int lrow = lane & 15, lk = lane >> 4;

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool has_next = (k0 + BK) < kend;
        if (has_next) {
            if (A_KC) g2r_kc(A, g.lda, m0, g.M, k0 + BK, kend, tid, ra_kc); else g2r_oc(A, g.lda, m0, g.M, k0 + BK, kend, tid, ra_oc);
            if (B_KC) g2r_kc(B, g.ldb, n0, g.N, k0 + BK, kend, tid, rb_kc); else g2r_oc(B, g.ldb, n0, g.N, k0 + BK, kend, tid, rb_oc);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fm[4], fn[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fm[t] = A_KC ? frag_kc(ldsA, wm + t * 16 + lrow, kk * 4 + lk) : frag_oc(ldsA, wm + t * 16, kk * 32, lane);
                fn[t] = B_KC ? frag_kc(ldsB, wn + t * 16 + lrow, kk * 4 + lk) : frag_oc(ldsB, wn + t * 16, kk * 32, lane);
            }
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fn[tn]),
                        __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fm[tm]), acc[tm][tn], 0, 0, 0);
            if (do_rowsum) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    accr[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, ones),
                        __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fm[tm]), accr[tm], 0, 0, 0);
            }
        }
        __syncthreads();
        if (has_next) {
            if (A_KC) r2s_kc(ldsA, tid, ra_kc); else r2s_oc(ldsA, tid, ra_oc);
            if (B_KC) r2s_kc(ldsB, tid, rb_kc); else r2s_oc(ldsB, tid, rb_oc);
        }
        __syncthreads();
    }

    // D[i][j]: i (A-operand row) = weight/N index = 4*(lane>>4)+r ; j (B-operand col) = M index = lane&15
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
            epilogue4<bf16_t>(g, m0 + wm + tm * 16 + lrow, n0 + wn + tn * 16 + 4 * lk, acc[tm][tn]);
    if (do_rowsum && lk == 0) {  // every row of the ones-product is the same sum; lane (lk = 0, r = 0) owns column lrow
        const float al = g.alpha_dev_out ? g.alpha_out * g.alpha_dev_out[0] : g.alpha_out;
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            int m = m0 + wm + tm * 16 + lrow;
            if (m < g.M) atomicAdd(g.rowsum + m, al * accr[tm][0]);
        }
    }
}

// =============================================================================================
// f32 (parity mode): v_mfma_f32_16x16x4_f32, BK = 16, LDS tiles stored [k][row] with row pitch 144
// =============================================================================================
#define FK 16
#define FP 144

template <bool KC>
__device__ __forceinline__ void f32_g2r(const float* P, long ld, int row0, int nrows, int k0, int kend, int tid,
                                        float4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int c = tid + 256 * i;
        if (KC) {
            int row = c >> 2, kc = c & 3;
            int gr = row0 + row, gk = k0 + kc * 4;
            r[i] = (gr < nrows && gk < kend) ? *reinterpret_cast<const float4*>(P + (long)gr * ld + gk) : make_float4(0, 0, 0, 0);
        } else {
            int kr = c >> 5, rc = c & 31;
            int gk = k0 + kr, gr = row0 + rc * 4;
            r[i] = (gr < nrows && gk < kend) ? *reinterpret_cast<const float4*>(P + (long)gk * ld + gr) : make_float4(0, 0, 0, 0);
        }
    }
}
template <bool KC>
__device__ __forceinline__ void f32_r2s(float* S, int tid, const float4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int c = tid + 256 * i;
        if (KC) {
            int row = c >> 2, kc = c & 3;
            S[(kc * 4 + 0) * FP + row] = r[i].x;
            S[(kc * 4 + 1) * FP + row] = r[i].y;
            S[(kc * 4 + 2) * FP + row] = r[i].z;
            S[(kc * 4 + 3) * FP + row] = r[i].w;
        } else {
            int kr = c >> 5, rc = c & 31;
            *reinterpret_cast<float4*>(S + kr * FP + rc * 4) = r[i];
        }
    }
}

template <bool A_KC, bool B_KC, bool ROWSUM>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float SA[FK * FP];
    __shared__ __attribute__((aligned(16))) float SB[FK * FP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = xcd_remap(blockIdx.x, g.nbm * g.nbn);
    const int m0 = (wg / g.nbn) * BM, n0 = (wg % g.nbn) * BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const float* A = reinterpret_cast<const float*>(g.A);
    const float* B = reinterpret_cast<const float*>(g.B);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ROWSUM is a compile-time flag: its 16 extra accumulators would cost every other GEMM form a wave of occupancy
    const bool do_rowsum = ROWSUM && g.rowsum != nullptr && (wg % g.nbn) == 0 && (wave & 1) == 0;
    f32x4 accr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accr[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ra[2], rb[2];
    f32_g2r<A_KC>(A, g.lda, m0, g.M, kbeg, kend, tid, ra);
    f32_g2r<B_KC>(B, g.ldb, n0, g.N, kbeg, kend, tid, rb);
    f32_r2s<A_KC>(SA, tid, ra);
    f32_r2s<B_KC>(SB, tid, rb);
    __syncthreads();

    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int lrow = lane & 15, lk = lane >> 4;
    for (int k0 = kbeg; k0 < kend; k0 += FK) {
        const bool has_next = (k0 + FK) < kend;
        if (has_next) {
            f32_g2r<A_KC>(A, g.lda, m0, g.M, k0 + FK, kend, tid, ra);
            f32_g2r<B_KC>(B, g.ldb, n0, g.N, k0 + FK, kend, tid, rb);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float fm[4], fn[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fm[t] = SA[(kk * 4 + lk) * FP + wm + t * 16 + lrow];
                fn[t] = SB[(kk * 4 + lk) * FP + wn + t * 16 + lrow];
            }
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fn[tn], fm[tm], acc[tm][tn], 0, 0, 0);
            if (do_rowsum) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) accr[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, fm[tm], accr[tm], 0, 0, 0);
            }
        }
        __syncthreads();
        if (has_next) {
            f32_r2s<A_KC>(SA, tid, ra);
            f32_r2s<B_KC>(SB, tid, rb);
        }
        __syncthreads();
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
            epilogue4<float>(g, m0 + wm + tm * 16 + lrow, n0 + wn + tn * 16 + 4 * lk, acc[tm][tn]);
    if (do_rowsum && lk == 0) {
        const float al = g.alpha_dev_out ? g.alpha_out * g.alpha_dev_out[0] : g.alpha_out;
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            int m = m0 + wm + tm * 16 + lrow;
            if (m < g.M) atomicAdd(g.rowsum + m, al * accr[tm][0]);
        }
    }
}

// C[m,n] (+)= alpha * sum_z partial[z][m][n]   -- deterministic split-K combine (no f32 atomics: 16.5 M scattered
// atomics per weight gradient cost more than the GEMM itself on this chip)
__global__ void splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ C, long M, long N, long ldc, int splits,
                                     float alpha, const float* __restrict__ alpha_dev, int accumulate) {
    const long n4 = M * N / 4;
    const float al = alpha_dev ? alpha * alpha_dev[0] : alpha;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 acc = reinterpret_cast<const float4*>(partial)[i];
        for (int z = 1; z < splits; ++z) {
            float4 p = reinterpret_cast<const float4*>(partial + (long)z * M * N)[i];
            acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
        long e = i * 4, m = e / N, n = e % N;
        float* c = C + m * ldc + n;
        float4 o = accumulate ? *reinterpret_cast<float4*>(c) : make_float4(0.f, 0.f, 0.f, 0.f);
        o.x += al * acc.x; o.y += al * acc.y; o.z += al * acc.z; o.w += al * acc.w;
        *reinterpret_cast<float4*>(c) = o;
    }
}

// =============================================================================================
// host entry
// =============================================================================================
extern "C" int ecamp_gemm(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int a_kc, int64_t lda,
                          int b_kc, int64_t ldb, int64_t ldc, const float* bias, const void* residual, int64_t ldr,
                          void* pre_out, int64_t ldp, const void* gmul, int64_t ldg, int act, float alpha, const float* alpha_dev, int dtype,
                          int out_f32, int accumulate, int split_k, float* splitk_ws, float* rowsum, hipStream_t stream) {
    ECAMP_CHECK_ARG(A && B && C, "ecamp_gemm: null operand");
    ECAMP_CHECK_ARG(M > 0 && N > 0 && K > 0, "ecamp_gemm: bad shape %ld %ld %ld", (long)M, (long)N, (long)K);
    ECAMP_CHECK_ARG(dtype == ECAMP_F32 || dtype == ECAMP_BF16, "ecamp_gemm: bad dtype %d", dtype);
    const int vec = dtype == ECAMP_BF16 ? 8 : 4;   // elements per 16-B global access
    const int ovec = dtype == ECAMP_BF16 ? 8 : 4;  // output-contiguous operands are read 16 B at a time
    ECAMP_CHECK_ARG(N % 4 == 0, "ecamp_gemm: N=%ld must be a multiple of 4", (long)N);
    if (a_kc) ECAMP_CHECK_ARG(K % vec == 0 && lda % vec == 0, "ecamp_gemm: K/lda alignment (A k-contiguous)");
    else ECAMP_CHECK_ARG(M % ovec == 0 && lda % ovec == 0, "ecamp_gemm: M/lda alignment (A m-contiguous), M=%ld", (long)M);
    if (b_kc) ECAMP_CHECK_ARG(K % vec == 0 && ldb % vec == 0, "ecamp_gemm: K/ldb alignment (B k-contiguous)");
    else ECAMP_CHECK_ARG(N % ovec == 0 && ldb % ovec == 0, "ecamp_gemm: N/ldb alignment (B n-contiguous)");
    ECAMP_CHECK_ARG(!accumulate || out_f32 || dtype == ECAMP_F32, "ecamp_gemm: accumulate needs an f32 output");
    ECAMP_CHECK_ARG(!rowsum || (!a_kc && !b_kc), "ecamp_gemm: rowsum is built for the weight-gradient form (both operands strided) only");
    if (split_k < 1) split_k = 1;
    ECAMP_CHECK_ARG(split_k == 1 || (out_f32 || dtype == ECAMP_F32), "ecamp_gemm: split_k > 1 requires an f32 output");
    ECAMP_CHECK_ARG(split_k == 1 || splitk_ws, "ecamp_gemm: split_k > 1 requires a workspace of split_k*M*N floats");
    ECAMP_CHECK_ARG(split_k == 1 || (!bias && !residual && !pre_out && !gmul && !act), "ecamp_gemm: split-K has no epilogue");

    GemmArgs g;
    g.A = A; g.B = B; g.C = C;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.bias = bias; g.residual = residual; g.ldr = ldr; g.pre_out = pre_out; g.ldp = ldp; g.gmul = gmul; g.ldg = ldg;
    g.alpha = alpha;
    g.alpha_dev = alpha_dev;
    g.alpha_out = alpha;
    g.alpha_dev_out = alpha_dev;
    g.rowsum = rowsum;
    g.act = act; g.out_f32 = (out_f32 || dtype == ECAMP_F32) ? 1 : 0; g.accumulate = accumulate;
    const int ktile = dtype == ECAMP_BF16 ? BK : FK;
    long kps = (K + split_k - 1) / split_k;
    kps = ((kps + ktile - 1) / ktile) * ktile;
    split_k = (int)((K + kps - 1) / kps);
    g.k_per_split = (int)kps;
    g.partial = split_k > 1 ? splitk_ws : nullptr;
    if (split_k > 1) { g.alpha = 1.0f; g.alpha_dev = nullptr; }
    g.nbm = ceil_div(M, BM); g.nbn = ceil_div(N, BN);
    dim3 grid(g.nbm * g.nbn, 1, split_k), block(256);
#define LAUNCH(KERN)                                                             \
    do {                                                                         \
        if (a_kc && b_kc) hipLaunchKernelGGL((KERN<true, true, false>), grid, block, 0, stream, g);        \
        else if (a_kc && !b_kc) hipLaunchKernelGGL((KERN<true, false, false>), grid, block, 0, stream, g); \
        else if (!a_kc && b_kc) hipLaunchKernelGGL((KERN<false, true, false>), grid, block, 0, stream, g); \
        else if (rowsum) hipLaunchKernelGGL((KERN<false, false, true>), grid, block, 0, stream, g);        \
        else hipLaunchKernelGGL((KERN<false, false, false>), grid, block, 0, stream, g);                   \
    } while (0)
    const bool prof = ecamp_prof_active();
    if (prof) ecamp_prof_begin(dtype == ECAMP_BF16 ? ECAMP_PROF_GEMM_BF16 : ECAMP_PROF_GEMM_F32, 2.0 * (double)M * (double)N * (double)K, stream);
    if (dtype == ECAMP_BF16) LAUNCH(gemm_bf16_kernel); else LAUNCH(gemm_f32_kernel);
#undef LAUNCH
    if (split_k > 1) {
        long n4 = M * N / 4;
        int nb = (int)((n4 + 255) / 256);
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb), dim3(256), 0, stream, splitk_ws, reinterpret_cast<float*>(C), (long)M, (long)N,
                           (long)ldc, split_k, alpha, alpha_dev, accumulate);
    }
    if (prof) ecamp_prof_end(stream);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
