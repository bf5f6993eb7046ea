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
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "gemm_args.h"
void attn_set_head_mode(int on);   // attention_bf16.hip
#include "gemm_q8.h"
#include "gemm_q16.h"

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
    const int mblk = wg / g.nbn, nblk = wg % g.nbn;   // row-major: the grouped order of the P8 kernel measured 8 % slower here
    const int m0 = mblk * BM, n0 = nblk * BN;
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
    const bool do_rowsum = ROWSUM && g.rowsum != nullptr && nblk == 0 && (wave & 1) == 0;
    f32x4 accr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accr[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {(short)H16_ONE, (short)H16_ONE, (short)H16_ONE, (short)H16_ONE, (short)H16_ONE, (short)H16_ONE, (short)H16_ONE, (short)H16_ONE};

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
                    acc[tm][tn] = ECAMP_MFMA_16x16x32(fn[tn], fm[tm], acc[tm][tn]);
            if (do_rowsum) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    accr[tm] = ECAMP_MFMA_16x16x32(ones, fm[tm], accr[tm]);
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
            epilogue4<bf16_t>(g, m0 + wm + tm * 16 + lrow, n0 + wn + tn * 16 + 4 * lk, acc[tm][tn], blockIdx.z);
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
    const int mblk = wg / g.nbn, nblk = wg % g.nbn;   // row-major: the grouped order of the P8 kernel measured 8 % slower here
    const int m0 = mblk * BM, n0 = nblk * BN;
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
    const bool do_rowsum = ROWSUM && g.rowsum != nullptr && nblk == 0 && (wave & 1) == 0;
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
            epilogue4<float>(g, m0 + wm + tm * 16 + lrow, n0 + wn + tn * 16 + 4 * lk, acc[tm][tn], blockIdx.z);
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
// fp8 forward GEMM (BASELINE.json configs[4]: fp8 MFMA forward, bf16 gradients): Y = act((A8 . B8^T) * sa * sb + bias) (+ residual)
// with A8 [M,K], B8 [N,K] in OCP e4m3 (one byte per element, contraction-contiguous) and per-tensor scales sa, sb on the device.
// The tile, LDS image and fragment reads are those of the 128^2 bf16 kernel with K counted in BYTES (128 per step): a lane's two
// 16-B fragment reads (chunks lk and lk+4 of a row) are the 32 k-values it feeds to ONE v_mfma_scale_f32_16x16x128_f8f6f4 (block
// scales 2^0) -- twice the contraction depth of the bf16 kernel per LDS byte and per matrix-pipe cycle.  Which 32 k a lane holds
// does not matter as long as both operands use the same split (tools/probes/fp8_mfma_probe.hip checks the instruction's row /
// k-group / output mapping and the converter's encodings on hardware).
// =============================================================================================
typedef __attribute__((ext_vector_type(8))) int v8i32;

__device__ __forceinline__ void g2r_kc8(const unsigned char* P, long ld, int row0, int nrows, int k0, int kend, int tid, uint4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = tid + 256 * i;
        int row = c >> 3, kc = c & 7;
        int gr = row0 + row, gk = k0 + kc * 16;
        r[i] = (gr < nrows && gk < kend) ? *reinterpret_cast<const uint4*>(P + (long)gr * ld + gk) : make_uint4(0, 0, 0, 0);
    }
}
__device__ __forceinline__ v8i32 frag_kc8(const unsigned char* lds, int row, int lk) {
    const uint4 lo = *reinterpret_cast<const uint4*>(lds + row * 128 + ((lk ^ swz(row)) << 4));
    const uint4 hi = *reinterpret_cast<const uint4*>(lds + row * 128 + (((lk + 4) ^ swz(row)) << 4));
    return (v8i32){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
}

__global__ __launch_bounds__(256) void gemm_fp8_kernel(GemmArgs g0, const float* __restrict__ sa, const float* __restrict__ sb) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 128 * 128];
    unsigned char* ldsA = lds;
    unsigned char* ldsB = lds + 128 * 128;
    GemmArgs g = g0;
    g.alpha = g0.alpha * sa[0] * sb[0];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = xcd_remap(blockIdx.x, g.nbm * g.nbn);
    const int mblk = wg / g.nbn, nblk = wg % g.nbn;
    const int m0 = mblk * BM, n0 = nblk * BN;
    const unsigned char* A = reinterpret_cast<const unsigned char*>(g.A);
    const unsigned char* B = reinterpret_cast<const unsigned char*>(g.B);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 ra[4], rb[4];
    g2r_kc8(A, g.lda, m0, g.M, 0, g.K, tid, ra);
    g2r_kc8(B, g.ldb, n0, g.N, 0, g.K, tid, rb);
    r2s_kc(ldsA, tid, ra);
    r2s_kc(ldsB, tid, rb);
    __syncthreads();
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int lrow = lane & 15, lk = lane >> 4;
    for (int k0 = 0; k0 < g.K; k0 += 128) {
        const bool has_next = (k0 + 128) < g.K;
        if (has_next) {
            g2r_kc8(A, g.lda, m0, g.M, k0 + 128, g.K, tid, ra);
            g2r_kc8(B, g.ldb, n0, g.N, k0 + 128, g.K, tid, rb);
        }
        v8i32 fm[4], fn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            fm[t] = frag_kc8(ldsA, wm + t * 16 + lrow, lk);
            fn[t] = frag_kc8(ldsB, wn + t * 16 + lrow, lk);
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fn[tn], fm[tm], acc[tm][tn], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        __syncthreads();
        if (has_next) {
            r2s_kc(ldsA, tid, ra);
            r2s_kc(ldsB, tid, rb);
        }
        __syncthreads();
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
            epilogue4<bf16_t>(g, m0 + wm + tm * 16 + lrow, n0 + wn + tn * 16 + 4 * lk, acc[tm][tn], 0);
}

// |x|_max of a tensor into out[0] (which the caller zeroes): non-negative floats order like their bit patterns
template <typename T>
__global__ void amax_kernel(const T* __restrict__ x, float* __restrict__ out, long n4) {
    __shared__ float sh[4];
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float v[4];
        ld4<T>(x + i * 4, v);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
    }
}
// q = e4m3(x / scale), scale = max(amax, tiny) / 448 written to scale_out[0]; values are clamped first (the converter maps
// anything above 448 + half an ulp to NaN, it does not saturate)
template <typename T>
__global__ void quant_fp8_kernel(const T* __restrict__ x, const float* __restrict__ amax, unsigned int* __restrict__ q,
                                 float* __restrict__ scale_out, long n4) {
    const float sc = fmaxf(amax[0], 1e-20f) * (1.0f / 448.0f);
    const float inv = 1.0f / sc;
    if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = sc;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float v[4];
        ld4<T>(x + i * 4, v);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf(v[r] * inv, -448.f), 448.f);
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
        q[i] = (unsigned int)w;
    }
}

// ---- delayed scaling (round 4).  A GEMM input site owns one scale (f32, the one its producer quantises with and the GEMM dequantises
// with during THIS optimizer step) and 16 amax slots 128 B apart (what the producers of this step have seen so far: their waves
// atomicMax into slot (workgroup & 15), so that a 12800-wave LayerNorm does not serialise on one L2 line); ecamp_fp8_roll turns the
// slots into the next step's scale and clears them.  One pass over the activation instead of amax + quantise.
#define F8_SLOTS 16
#define F8_SLOT_STRIDE 32   // floats
template <typename T>
__global__ __launch_bounds__(256) void quant_fp8_delayed_kernel(const T* __restrict__ x, const float* __restrict__ scale, unsigned int* __restrict__ q,
                                                                float* __restrict__ amax_slots, long n4) {
    const float inv = 1.0f / fmaxf(scale[0], 1e-30f);
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float v[4];
        ld4<T>(x + i * 4, v);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf(v[r] * inv, -448.f), 448.f);
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
        q[i] = (unsigned int)w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(amax_slots + (blockIdx.x & (F8_SLOTS - 1)) * F8_SLOT_STRIDE), __float_as_uint(m));
}
__global__ void fp8_roll_kernel(float* __restrict__ amax_slots, float* __restrict__ scale, int n, float* __restrict__ hist, int hist_len, int hist_pos,
                                float margin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* s = amax_slots + (long)i * F8_SLOTS * F8_SLOT_STRIDE;
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < F8_SLOTS; ++k) { a = fmaxf(a, s[k * F8_SLOT_STRIDE]); s[k * F8_SLOT_STRIDE] = 0.f; }
    if (a <= 0.f) return;   // a site nobody fed this step keeps its scale (and its history)
    if (hist) {             // delayed scaling with a memory: the scale covers the largest maximum of the last hist_len fed steps
        float* h = hist + (long)i * hist_len;
        h[hist_pos % hist_len] = a;
        for (int k = 0; k < hist_len; ++k) a = fmaxf(a, h[k]);
    }
    scale[i] = a * margin * (1.0f / 448.0f);
}
extern "C" int ecamp_quant_fp8_delayed(const void* x, const float* scale, void* q, float* amax_slots, int64_t n, int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(x && scale && q && amax_slots && n > 0 && n % 4 == 0, "quant_fp8_delayed: bad args");
    ECAMP_CHECK_ARG(dtype == ECAMP_F32 || dtype == ECAMP_BF16, "quant_fp8_delayed: bad dtype %d", dtype);
    const long n4 = n / 4;
    int nb = (int)((n4 + 255) / 256);
    if (nb > 4096) nb = 4096;
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(quant_fp8_delayed_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)x, scale, (unsigned int*)q, amax_slots, n4);
    else hipLaunchKernelGGL(quant_fp8_delayed_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)x, scale, (unsigned int*)q, amax_slots, n4);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
extern "C" int ecamp_fp8_roll(float* amax_slots, float* scale, int32_t n, float* hist, int32_t hist_len, int32_t hist_pos, float margin, hipStream_t stream) {
    ECAMP_CHECK_ARG(amax_slots && scale && n > 0, "fp8_roll: bad args");
    ECAMP_CHECK_ARG(margin >= 1.0f && margin <= 16.0f && (!hist || (hist_len > 0 && hist_len <= 64 && hist_pos >= 0)), "fp8_roll: margin in [1, 16], history of 1..64 steps");
    hipLaunchKernelGGL(fp8_roll_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, amax_slots, scale, (int)n, hist, (int)hist_len, (int)hist_pos, margin);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
// ---- the fp8 copies of ALL weights in two launches per optimizer step (they were 92 amax + 92 quantise launches of 6-18 us each: 2.7 ms
// of a 69 ms step).  One workgroup per table item {first element, count (multiple of 4, <= 65536), scale id, -}: pass 0 leaves the item's
// |w| maximum in the slots of its scale id (several tensors may share one: BERT's fused query / key / value block), ecamp_fp8_roll turns
// the slots into scales, pass 1 quantises with them.  Exact current scaling: a weight matrix is quantised with its own maximum.
__global__ __launch_bounds__(256) void fp8_weights_kernel(const bf16_t* __restrict__ w, unsigned int* __restrict__ q, const int4* __restrict__ items,
                                                          float* __restrict__ amax_slots, const float* __restrict__ scales, int pass) {
    __shared__ float sh[4];
    const int4 it = items[blockIdx.x];
    const long first = (long)(unsigned)it.x;   // element offset (multiple of 4)
    const int n4 = it.y >> 2;
    if (pass == 0) {
        float m = 0.f;
        for (int i = threadIdx.x; i < n4; i += 256) {
            float v[4];
            ld4<bf16_t>(w + first + 4 * (long)i, v);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
            atomicMax(reinterpret_cast<unsigned int*>(amax_slots + (long)it.z * (F8_SLOTS * F8_SLOT_STRIDE) + (blockIdx.x & (F8_SLOTS - 1)) * F8_SLOT_STRIDE), __float_as_uint(m));
        }
    } else {
        const float inv = 1.0f / fmaxf(scales[it.z], 1e-30f);
        for (int i = threadIdx.x; i < n4; i += 256) {
            float v[4];
            ld4<bf16_t>(w + first + 4 * (long)i, v);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf(v[r] * inv, -448.f), 448.f);
            int ww = 0;
            ww = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], ww, false);
            ww = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], ww, true);
            q[(first >> 2) + i] = (unsigned int)ww;
        }
    }
}
extern "C" int ecamp_fp8_weights(const void* w_bf16, void* w8, const int32_t* items, int32_t nitems, float* amax_slots, const float* scales,
                                 int32_t pass, hipStream_t stream) {
    ECAMP_CHECK_ARG(w_bf16 && w8 && items && nitems > 0 && amax_slots && scales && (pass == 0 || pass == 1), "fp8_weights: bad args");
    ECAMP_CHECK_ARG((reinterpret_cast<uintptr_t>(items) & 15) == 0, "fp8_weights: the item table must be 16-byte aligned");
    hipLaunchKernelGGL(fp8_weights_kernel, dim3(nitems), dim3(256), 0, stream, (const bf16_t*)w_bf16, (unsigned int*)w8, (const int4*)items, amax_slots, scales, (int)pass);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
extern "C" int ecamp_amax(const void* x, float* out, int64_t n, int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(x && out && n > 0 && n % 4 == 0, "amax: bad args");
    ECAMP_CHECK_ARG(dtype == ECAMP_F32 || dtype == ECAMP_BF16, "amax: bad dtype %d", dtype);
    const long n4 = n / 4;
    int nb = (int)((n4 + 255) / 256);
    if (nb > 2048) nb = 2048;
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(amax_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)x, out, n4);
    else hipLaunchKernelGGL(amax_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)x, out, n4);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
extern "C" int ecamp_quant_fp8(const void* x, const float* amax, void* q, float* scale_out, int64_t n, int32_t dtype, hipStream_t stream) {
    ECAMP_CHECK_ARG(x && amax && q && scale_out && n > 0 && n % 4 == 0, "quant_fp8: bad args");
    ECAMP_CHECK_ARG(dtype == ECAMP_F32 || dtype == ECAMP_BF16, "quant_fp8: bad dtype %d", dtype);
    const long n4 = n / 4;
    int nb = (int)((n4 + 255) / 256);
    if (nb > 4096) nb = 4096;
    if (dtype == ECAMP_F32) hipLaunchKernelGGL(quant_fp8_kernel<float>, dim3(nb), dim3(256), 0, stream, (const float*)x, amax, (unsigned int*)q, scale_out, n4);
    else hipLaunchKernelGGL(quant_fp8_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)x, amax, (unsigned int*)q, scale_out, n4);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
static int p8_num_cu();
static int q8_env();
static long q8_min_items();
static long g_f8_q8_launches = 0;
extern "C" int64_t ecamp_gemm_f8_q8_launches(void) { return g_f8_q8_launches; }
extern "C" int ecamp_gemm_fp8(const void* A8, const void* B8, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                              int64_t ldc, const float* scale_a, const float* scale_b, const float* bias, const void* residual,
                              int64_t ldr, void* pre_out, int64_t ldp, int act, void* q8_out, const float* q8_scale, float* q8_amax_slots,
                              hipStream_t stream) {
    ECAMP_CHECK_ARG(!q8_out || (q8_scale && q8_amax_slots && (act == 1 || act == 2) && pre_out && !residual && ldc == N),
                    "ecamp_gemm_fp8: the e4m3 copy of the output needs its scale and amax slots, the GELU epilogue and a dense C");
    ECAMP_CHECK_ARG(A8 && B8 && C && scale_a && scale_b, "ecamp_gemm_fp8: null operand");
    ECAMP_CHECK_ARG(act >= 0 && act <= 2 && (act != 2 || pre_out), "ecamp_gemm_fp8: act=%d (0 none, 1 GELU, 2 GELU with the saved derivative in pre_out)", act);
    ECAMP_CHECK_ARG(M > 0 && N > 0 && K > 0, "ecamp_gemm_fp8: bad shape %ld %ld %ld", (long)M, (long)N, (long)K);
    ECAMP_CHECK_ARG(K % 16 == 0 && lda % 16 == 0 && ldb % 16 == 0 && N % 4 == 0, "ecamp_gemm_fp8: K, lda, ldb must be multiples of 16 and N of 4");
    {   // outputs past 2 GB (the vocabulary projection at B = 512: 65536 x 30000 bf16): two calls over row halves, as ecamp_gemm does
        const long lim = 0x7fffffffl;
        if (M >= 512 && !q8_out && (M * ldc * 2 > lim || (residual && M * ldr * 2 > lim) || (pre_out && M * ldp * 2 > lim))) {
            const int64_t m1 = (M / 2 + 255) / 256 * 256;
            auto rows = [](const void* p, int64_t r, int64_t ld, int64_t es) { return p ? (const void*)((const char*)p + r * ld * es) : nullptr; };
            int rc = ecamp_gemm_fp8(A8, B8, C, m1, N, K, lda, ldb, ldc, scale_a, scale_b, bias, residual, ldr, pre_out, ldp, act, nullptr, nullptr, nullptr, stream);
            if (rc) return rc;
            return ecamp_gemm_fp8(rows(A8, m1, lda, 1), B8, (void*)rows(C, m1, ldc, 2), M - m1, N, K, lda, ldb, ldc, scale_a, scale_b, bias, rows(residual, m1, ldr, 2),
                                  ldr, (void*)rows(pre_out, m1, ldp, 2), ldp, act, nullptr, nullptr, nullptr, stream);
        }
    }
    GemmArgs g;
    g.dbg = 0; g.wide = 0; g.nsplit = 1;
    g.A = A8; g.B = B8; g.C = C;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.bias = bias; g.residual = residual; g.ldr = ldr; g.pre_out = pre_out; g.ldp = ldp; g.gmul = nullptr; g.ldg = 0;
    g.alpha = 1.0f; g.alpha_dev = nullptr; g.alpha_dev2 = nullptr; g.alpha_out = 1.0f; g.alpha_dev_out = nullptr;
    g.q8_out = nullptr; g.q8_scale = nullptr; g.q8_amax = nullptr;
    g.rowsum = nullptr;
    g.act = act; g.out_f32 = 0; g.accumulate = 0;
    g.k_per_split = (int)K;
    g.partial = nullptr;
    g.nbm = ceil_div(M, BM); g.nbn = ceil_div(N, BN);
    const bool prof = ecamp_prof_active();
    if (prof) ecamp_prof_begin(ECAMP_PROF_GEMM_FP8, 2.0 * (double)M * (double)N * (double)K, stream);
    // the persistent 256 x 256 x 128 e4m3 form (gemm_q8.h, F8) from the same tile count up as the bf16 kernel, when its alignment /
    // size conditions hold; otherwise the 128^2 kernel below.  ECAMP_F8_Q8=0 keeps everything on the 128^2 kernel (development A/B).
    {
        static const int f8q8 = getenv("ECAMP_F8_Q8") ? atoi(getenv("ECAMP_F8_Q8")) : 1;
        const int q8m = q8_env();
        const int epi = (pre_out || act) ? ((pre_out && (act == 1 || act == 2) && !residual) ? 1 : -1) : residual ? 2 : 0;
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        const long lim = 0x7fffffffl;
        const long items8 = (long)ceil_div(M, 256) * ceil_div(N, 256);
        const bool legal = f8q8 && q8m != 0 && epi >= 0 && K >= 256 && N % 8 == 0 && ldc % 8 == 0 && al16(A8) && al16(B8) && al16(C) &&
                           M * lda <= lim && N * ldb <= lim && M * ldc * 2 <= lim && (!bias || al16(bias)) &&
                           (!pre_out || (ldp % 8 == 0 && al16(pre_out) && M * ldp * 2 <= lim)) &&
                           (!residual || (ldr % 8 == 0 && al16(residual) && M * ldr * 2 <= lim));
        if (legal && (q8m == 2 || items8 >= q8_min_items())) {
            typedef void (*f8_fn)(GemmArgs);
            const f8_fn fn = epi == 0 ? (f8_fn)gemm_f8_q8_kernel<0> : epi == 1 ? (f8_fn)gemm_f8_q8_kernel<1> : (f8_fn)gemm_f8_q8_kernel<2>;
            g.nbm = ceil_div(M, 256); g.nbn = ceil_div(N, 256); g.nsplit = 1; g.wide = 1;
            g.alpha_dev = scale_a; g.alpha_dev2 = scale_b;
            g.q8_out = q8_out; g.q8_scale = q8_scale; g.q8_amax = q8_amax_slots;
            const size_t shm = 10 * Q8_HALF;
            static bool attr[3] = {false, false, false};
            if (!attr[epi]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr[epi] = true; }
            const int ncu = p8_num_cu();
            hipLaunchKernelGGL(fn, dim3((unsigned)(items8 < ncu ? items8 : ncu)), dim3(512), shm, stream, g);
            ++g_f8_q8_launches;
            if (prof) ecamp_prof_end(stream);
            ECAMP_LAUNCH_CHECK();
            return 0;
        }
    }
    hipLaunchKernelGGL(gemm_fp8_kernel, dim3(g.nbm * g.nbn), dim3(256), 0, stream, g, scale_a, scale_b);
    if (prof) ecamp_prof_end(stream);
    if (q8_out) {   // the 128^2 kernel has no third output: one pass over C afterwards gives the same bytes
        const long n4 = M * N / 4;
        int nb = (int)((n4 + 255) / 256);
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(quant_fp8_delayed_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)C, q8_scale, (unsigned int*)q8_out, q8_amax_slots, n4);
    }
    ECAMP_LAUNCH_CHECK();
    return 0;
}

// =============================================================================================
// host entry
// =============================================================================================
// ---- kernel selection ------------------------------------------------------------------------------------------
static int p8_num_cu() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
        // development (profiles/r05_epilogue_scale.txt): persistent launches on fewer workgroups than CUs, so that launches of two streams
        // sit side by side instead of one behind the other
        const char* cap = getenv("ECAMP_GEMM_GRID_CAP");
        if (cap && atoi(cap) >= 32 && atoi(cap) < ncu) {
            ncu = atoi(cap);
            fprintf(stderr, "[ecamp_hip] WARNING: ECAMP_GEMM_GRID_CAP=%d -- a development switch: every persistent GEMM of this process runs on %d "
                            "workgroups instead of one per CU\n", ncu, ncu);
        }
    }
    return ncu;
}
// A persistent workgroup needs a whole CU (all 160 KB of LDS, all registers): when another kernel holds even one wave on a CU, the
// workgroup assigned there starts late and the launch takes up to twice as long (tools/hog_probe.py: +11-17 % on the step with a
// co-resident spinning kernel, against +1-5 % for the 128^2 kernel).  During the backward pass of a multi-GPU run RCCL's
// all-reduce workgroups are such co-tenants.  Two process-wide switches (ecamp_set_option): "p8_wgrad" = 0 keeps the
// weight-gradient form off the persistent kernel; "p8_wgrad_reserve_cus" = n launches it with n fewer workgroups than CUs, so
// that a communication kernel that is already resident (or arrives between two GEMMs) finds CUs without displacing a
// persistent workgroup.  The data-parallel wrapper sets the reserve; the forward pass has no communication beside it.
static int g_p8_wgrad = 1;
static int g_p8_wgrad_reserve = 0;
// "q8_bwd_grid" = n > 0 launches the DATA-GRADIENT form on min(items, n) workgroups instead of one per CU; n >= items gives one
// output tile per workgroup, i.e. the hardware dispatcher hands tiles to whichever CU is free.  That is what the data-parallel
// wrapper asks for: beside RCCL's all-reduce workgroups a persistent workgroup whose CU is taken starts late and holds its whole
// static share of the tiles back, while one-tile workgroups simply flow around the occupied CUs.  Measured cost on a GPU of its
// own (tools/grid_ab.sh): +0.15 ms per step for the data-gradient form (+0.4 ms if the forward form did the same, which it does not
// need: nothing communicates during forward) -- the cross-tile DMA prefetch of the persistent loop is worth that much and no more.
static int g_q8_bwd_grid = 0;

// ---- Q8 (gemm_q8.h): the persistent 256x256x64 kernel.  ECAMP_GEMM_Q8 / option "q8_mode": -1 automatic (default), 0 never, 2 whenever legal.
static int g_q8_mode = -2;
static int q8_env() {
    static const int v = getenv("ECAMP_GEMM_Q8") ? atoi(getenv("ECAMP_GEMM_Q8")) : -1;
    return g_q8_mode != -2 ? g_q8_mode : v;
}
// epilogue variant of a call (-1: none fits)
static int q8_epi(const float* bias, const void* residual, const void* pre_out, const void* gmul, int act, int out_f32) {
    if (out_f32) return (!bias && !residual && !pre_out && !gmul && !act) ? 4 : -1;
    if (gmul) return (!bias && !pre_out && (act == 0 || act == 2)) ? 3 : -1;
    if (pre_out || act) return (pre_out && (act == 1 || act == 2) && !residual) ? 1 : -1;
    if (residual) return 2;
    return 0;
}
static long g_q8_launches = 0;
extern "C" int64_t ecamp_gemm_q8_launches(void) { return g_q8_launches; }
typedef void (*q8_fn)(GemmArgs);
// "q8_sch" (development A/B; env ECAMP_Q8_SCH): bit 0 forward form, bit 1 data-gradient form, bit 2 weight-gradient form, bit 3 the
// grouped weight gradients on the lean stream (gemm_q8.h SCH = 1) instead of the round-3 stream
static int g_q8_sch = -1;
static int q8_sch() {
    static const int v = getenv("ECAMP_Q8_SCH") ? atoi(getenv("ECAMP_Q8_SCH")) : 7;
    return g_q8_sch >= 0 ? g_q8_sch : v;
}
static q8_fn q8_pick(int a_kc, int b_kc, int epi, bool rowsum) {
    const int sch = q8_sch();
#define Q8S(A, B, E, R) ((q8_fn)gemm_bf16_q8_kernel<A, B, E, 0, R, 1>)
    if (a_kc && b_kc && (sch & 1)) return epi == 0 ? Q8S(true, true, 0, false) : epi == 1 ? Q8S(true, true, 1, false) : epi == 2 ? Q8S(true, true, 2, false) : (q8_fn) nullptr;
    if (a_kc && !b_kc && (sch & 2)) return epi == 0 ? Q8S(true, false, 0, false) : epi == 2 ? Q8S(true, false, 2, false) : epi == 3 ? Q8S(true, false, 3, false) : (q8_fn) nullptr;
    if (!a_kc && !b_kc && epi == 4 && (sch & 4)) return rowsum ? Q8S(false, false, 4, true) : Q8S(false, false, 4, false);
#undef Q8S
    if (a_kc && b_kc) return epi == 0 ? gemm_bf16_q8_kernel<true, true, 0> : epi == 1 ? gemm_bf16_q8_kernel<true, true, 1> : epi == 2 ? gemm_bf16_q8_kernel<true, true, 2> : (q8_fn) nullptr;
    if (a_kc && !b_kc) return epi == 0 ? gemm_bf16_q8_kernel<true, false, 0> : epi == 2 ? gemm_bf16_q8_kernel<true, false, 2> : epi == 3 ? gemm_bf16_q8_kernel<true, false, 3> : (q8_fn) nullptr;
    if (!a_kc && !b_kc && epi == 4) return rowsum ? gemm_bf16_q8_kernel<false, false, 4, 0, true> : gemm_bf16_q8_kernel<false, false, 4, 0, false>;
    return nullptr;
}
static bool q8_legal(const void* A, const void* B, const void* C, int64_t M, int64_t N, int64_t K, int a_kc, int64_t lda, int b_kc, int64_t ldb, int64_t ldc,
                     const float* bias, const void* residual, int64_t ldr, const void* pre_out, int64_t ldp, const void* gmul, int64_t ldg, int act,
                     int dtype, int out_f32, int split_k, const float* splitk_ws, const float* rowsum) {
    if (dtype != ECAMP_BF16) return false;
    const int epi = q8_epi(bias, residual, pre_out, gmul, act, out_f32);
    if (epi < 0 || !q8_pick(a_kc, b_kc, epi, rowsum != nullptr)) return false;
    if (split_k < 1) split_k = 1;
    if (split_k > 1 && epi != 4) return false;
    long kps = (K + split_k - 1) / split_k;
    kps = (kps + 63) / 64 * 64;
    const long ns = (K + kps - 1) / kps, last = K - (ns - 1) * kps;
    if (kps < 128 || last <= 64) return false;   // every work item (the last slice included) has at least two K tiles
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const long lim = 0x7fffffffl;
    if (N % 8 || lda % 8 || ldb % 8 || !al16(A) || !al16(B) || !al16(C)) return false;
    if ((a_kc ? M * lda : K * lda) * 2 > lim || (b_kc ? N * ldb : K * ldb) * 2 > lim) return false;
    if (!a_kc && M % 8) return false;
    if (split_k > 1) { if (!al16(splitk_ws) || (long)split_k * M * N * 4 > lim) return false; }
    else if (ldc % 8 || M * ldc * (epi == 4 ? 4 : 2) > lim) return false;
    if (bias && !al16(bias)) return false;
    if (pre_out && (ldp % 8 || !al16(pre_out) || M * ldp * 2 > lim)) return false;
    if (gmul && (ldg % 8 || !al16(gmul) || M * ldg * 2 > lim)) return false;
    if (residual && (ldr % 8 || !al16(residual) || M * ldr * 2 > lim)) return false;
    return true;
}

// ---- Q16 (gemm_q16.h): four waves on the 16x16x32 MFMA, forward and data-gradient forms, 256- or 192-column tiles.
// "q16_mode" (env ECAMP_Q16): 0 never; 1 (default) where the 192-column tile removes idle last-round time (the 768-wide outputs of the model);
// 2 every eligible call, with the tile width the round count favours; 3 (tests) as 2 whatever the size
static int g_q16_mode = -1;
static int q16_mode() {
    static const int v = getenv("ECAMP_Q16") ? atoi(getenv("ECAMP_Q16")) : 1;
    return g_q16_mode >= 0 ? g_q16_mode : v;
}
static long g_q16_launches = 0;
extern "C" int64_t ecamp_gemm_q16_launches(void) { return g_q16_launches; }
// rounds of the chip a launch needs with TN-column tiles, in units of one 256 x 256 tile's time (a 256 x 192 tile costs ~0.79 of it:
// tools/gemm_lab, profiles/r05_vendor_vs_q8.txt)
static double q16_cost(int64_t M, int64_t N, int tn, int ncu) {
    const long tiles = (long)ceil_div(M, 256) * ceil_div(N, tn);
    return (double)((tiles + ncu - 1) / ncu) * (tn == 192 ? 0.79 : 1.0);
}
typedef void (*q16_fn)(GemmArgs);
static q16_fn q16_pick(int b_kc, int epi, int nw) {
    if (b_kc) {
        if (nw == 8) return epi == 0 ? (q16_fn)gemm_bf16_q16_kernel<0, 8, true> : (q16_fn)gemm_bf16_q16_kernel<2, 8, true>;
        return epi == 0 ? (q16_fn)gemm_bf16_q16_kernel<0, 6, true> : (q16_fn)gemm_bf16_q16_kernel<2, 6, true>;
    }
    if (nw == 8) return epi == 0 ? (q16_fn)gemm_bf16_q16_kernel<0, 8, false> : (q16_fn)gemm_bf16_q16_kernel<2, 8, false>;
    return epi == 0 ? (q16_fn)gemm_bf16_q16_kernel<0, 6, false> : (q16_fn)gemm_bf16_q16_kernel<2, 6, false>;
}

extern "C" int ecamp_set_option(const char* name, int32_t value) {
    ECAMP_CHECK_ARG(name != nullptr, "set_option: null name");
    if (strcmp(name, "q16_mode") == 0) { g_q16_mode = (value >= 0 && value <= 3) ? value : -1; return 0; }
    if (strcmp(name, "q8_mode") == 0) { g_q8_mode = (value == 0 || value == 2) ? value : -1; return 0; }   // -1 auto, 0 never, 2 whenever legal
    if (strcmp(name, "p8_wgrad") == 0) { g_p8_wgrad = value ? 1 : 0; return 0; }
    if (strcmp(name, "p8_wgrad_reserve_cus") == 0) { g_p8_wgrad_reserve = value < 0 ? 0 : value; return 0; }
    if (strcmp(name, "q8_bwd_grid") == 0) { g_q8_bwd_grid = value < 0 ? 0 : value; return 0; }
    if (strcmp(name, "q8_sch") == 0) { g_q8_sch = value; return 0; }
    if (strcmp(name, "attn_head") == 0) { attn_set_head_mode(value); return 0; }   // attention_bf16.hip: 1 head kernels (default), 0 streaming kernels
    return ecamp_set_error(-1, "set_option: unknown option '%s'", name);
}

// the Q8 kernel is selected from this many 256^2 work items up (measured: 150 tiles on 256 CUs still beat the 128^2 kernel by 10-20 %)
static long q8_min_items() {
    static const long v = getenv("ECAMP_Q8_MIN_ITEMS") ? atol(getenv("ECAMP_Q8_MIN_ITEMS")) : (long)(0.5 * p8_num_cu());
    return v;
}
// weight-gradient GEMMs: work items (tiles x split-K slices) the split is chosen for.  A slice costs an M x N f32 slab written and
// re-read, so fewer, longer items are cheaper per FLOP; on the side stream the rest of the chip is busy with the data-gradient chain
// anyway.  ECAMP_WGRAD_ITEMS overrides (development).
static long wgrad_target_items(int ncu) {
    static const long v = getenv("ECAMP_WGRAD_ITEMS") ? atol(getenv("ECAMP_WGRAD_ITEMS")) : 0;
    return v > 0 ? v : ncu;
}

extern "C" int ecamp_gemm_suggest_split(int64_t M, int64_t N, int64_t K, int a_kc, int b_kc, int dtype) {
    if (M <= 0 || N <= 0 || K <= 0) return 1;
    int ncu = p8_num_cu();
    if (!(a_kc && b_kc) && g_p8_wgrad_reserve > 0 && ncu - g_p8_wgrad_reserve >= 64) ncu -= g_p8_wgrad_reserve;   // as the launch does
    // 128^2 kernel: about four resident workgroups per CU
    const long tiles = (long)ceil_div(M, 128) * ceil_div(N, 128);
    long s_old = 1024 / tiles;
    if (s_old < 1) s_old = 1;
    const long cap = (K + 255) / 256;
    if (s_old > cap) s_old = cap;
    if (dtype != ECAMP_BF16 || q8_env() == 0 || (a_kc != 0) != (b_kc != 0) || (!a_kc && !b_kc && !g_p8_wgrad)) return (int)s_old;
    // persistent 256^2 kernel: one workgroup per CU walks the work items; pick the split count whose item count fills whole
    // rounds of the chip, preferring fewer splits (each split writes and re-reads an M x N f32 slab)
    const long t8 = (long)ceil_div(M, 256) * ceil_div(N, 256);
    const long tgt = (!a_kc && !b_kc) ? wgrad_target_items(ncu) : ncu;
    int best = 1;
    double best_score = -1.0;
    for (int sp = 1; sp <= 32; ++sp) {
        if (sp > 1 && K / sp < 512) break;
        const long items = t8 * sp;
        const long rounds = (items + tgt - 1) / tgt;
        const double score = (double)items / (double)(rounds * tgt) - 0.012 * (sp - 1);
        if (score > best_score + 1e-9) { best_score = score; best = sp; }
    }
    return t8 * best >= q8_min_items() ? best : (int)s_old;
}

// Workspace of ecamp_gemm(..., split_k, splitk_ws, ...): split_k stacked M x N f32 slabs (0 when split_k <= 1).  The split count is
// the caller's (ecamp_gemm_suggest_split recommends one); the library only ever lowers it (whole K tiles per slice).
extern "C" int64_t ecamp_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int32_t split_k) {
    (void)K;
    return split_k > 1 && M > 0 && N > 0 ? (int64_t)split_k * M * N * 4 : 0;
}

extern "C" int ecamp_gemm(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int a_kc, int64_t lda,
                          int b_kc, int64_t ldb, int64_t ldc, const float* bias, const void* residual, int64_t ldr,
                          void* pre_out, int64_t ldp, const void* gmul, int64_t ldg, int act, float alpha, const float* alpha_dev, int dtype,
                          int out_f32, int accumulate, int split_k, float* splitk_ws, float* rowsum, hipStream_t stream) {
    ECAMP_CHECK_ARG(A && B && C, "ecamp_gemm: null operand");
    ECAMP_CHECK_ARG(M > 0 && N > 0 && K > 0, "ecamp_gemm: bad shape %ld %ld %ld", (long)M, (long)N, (long)K);
    ECAMP_CHECK_ARG(dtype == ECAMP_F32 || dtype == ECAMP_BF16, "ecamp_gemm: bad dtype %d", dtype);
    ECAMP_CHECK_ARG(act >= 0 && act <= 2, "ecamp_gemm: act=%d (0 none, 1 GELU, 2 GELU with the saved derivative)", act);
    ECAMP_CHECK_ARG(act != 2 || (dtype == ECAMP_BF16 && !out_f32 && ((pre_out != nullptr) != (gmul != nullptr))),
                    "ecamp_gemm: act=2 (saved derivative) is the bf16 forward form with pre_out or the data-gradient form with gmul");
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

    // Row-contiguous forms whose [M, ld] operands pass 2 GB (the vocabulary projection at B = 512: 65536 x 30000 bf16) are run as
    // two calls over row halves, so that each half meets the 32-bit buffer offsets of the persistent kernels (q8_legal).
    if (a_kc && dtype == ECAMP_BF16 && split_k == 1 && !rowsum && M >= 512) {
        const long lim = 0x7fffffffl, esz = out_f32 ? 4 : 2;
        const bool big = M * ldc * esz > lim || M * lda * 2 > lim || (residual && M * ldr * 2 > lim) || (pre_out && M * ldp * 2 > lim) || (gmul && M * ldg * 2 > lim);
        if (big) {
            const int64_t m1 = (M / 2 + 255) / 256 * 256;
            auto rows = [](const void* p, int64_t r, int64_t ld, int64_t es) { return p ? (const void*)((const char*)p + r * ld * es) : nullptr; };
            int rc = ecamp_gemm(A, B, C, m1, N, K, a_kc, lda, b_kc, ldb, ldc, bias, residual, ldr, pre_out, ldp, gmul, ldg, act, alpha, alpha_dev, dtype,
                                out_f32, accumulate, 1, nullptr, nullptr, stream);
            if (rc) return rc;
            return ecamp_gemm(rows(A, m1, lda, 2), B, (void*)rows(C, m1, ldc, esz), M - m1, N, K, a_kc, lda, b_kc, ldb, ldc, bias, rows(residual, m1, ldr, 2), ldr,
                              (void*)rows(pre_out, m1, ldp, 2), ldp, rows(gmul, m1, ldg, 2), ldg, act, alpha, alpha_dev, dtype, out_f32, accumulate, 1, nullptr,
                              nullptr, stream);
        }
    }
    // ... and the weight-gradient form whose [K, ld] operands pass 2 GB as two calls over halves of the contraction, the second accumulating
    if (!a_kc && !b_kc && dtype == ECAMP_BF16 && out_f32 && K >= 1024) {
        const long lim = 0x7fffffffl;
        if (K * lda * 2 > lim || K * ldb * 2 > lim) {
            const int64_t k1 = (K / 2 + 255) / 256 * 256;
            int rc = ecamp_gemm(A, B, C, M, N, k1, a_kc, lda, b_kc, ldb, ldc, bias, residual, ldr, pre_out, ldp, gmul, ldg, act, alpha, alpha_dev, dtype, out_f32,
                                accumulate, split_k, splitk_ws, rowsum, stream);
            if (rc) return rc;
            return ecamp_gemm((const char*)A + k1 * lda * 2, (const char*)B + k1 * ldb * 2, C, M, N, K - k1, a_kc, lda, b_kc, ldb, ldc, bias, residual, ldr, pre_out, ldp,
                              gmul, ldg, act, alpha, alpha_dev, dtype, out_f32, 1, split_k, splitk_ws, rowsum, stream);
        }
    }

    GemmArgs g;
    g.dbg = 0; g.wide = 0; g.nsplit = 1;
    g.A = A; g.B = B; g.C = C;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.bias = bias; g.residual = residual; g.ldr = ldr; g.pre_out = pre_out; g.ldp = ldp; g.gmul = gmul; g.ldg = ldg;
    g.alpha = alpha;
    g.alpha_dev = alpha_dev;
    g.alpha_dev2 = nullptr;
    g.q8_out = nullptr; g.q8_scale = nullptr; g.q8_amax = nullptr;
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
    {
        const int q8m = q8_env();
        const long items8 = (long)ceil_div(M, 256) * ceil_div(N, 256) * split_k;
        if (q8m != 0 && (q8m == 2 || items8 >= q8_min_items() || q16_mode() == 3) && (a_kc || b_kc || g_p8_wgrad) &&
            q8_legal(A, B, C, M, N, K, a_kc, lda, b_kc, ldb, ldc, bias, residual, ldr, pre_out, ldp, gmul, ldg, act, dtype, g.out_f32, split_k, splitk_ws, rowsum)) {
            const int epi = q8_epi(bias, residual, pre_out, gmul, act, g.out_f32);
            {   // Q16: forward / data-gradient forms with a plain, bias or residual epilogue (same legality as Q8: 16-B alignment, < 2 GB)
                const int m16 = q16_mode(), ncu16 = p8_num_cu();
                if (m16 > 0 && a_kc && (epi == 0 || epi == 2) && split_k == 1 && !rowsum && (b_kc || ldb % 8 == 0)) {
                    const double c256 = q16_cost(M, N, 256, ncu16), c192 = q16_cost(M, N, 192, ncu16);
                    // (the 192-column tile only where it removes a good part of a round: the report side's qkv projection -- 6.0 rounds of
                    // 3/4-size tiles against 4.5 -> 5 -- measured 8 % SLOWER inside the step, profiles/r05_gemm_in_step_vs_lab.txt)
                    const int nw = c192 <= 0.9 * c256 ? 6 : 8;
                    if (m16 >= 2 || nw == 6) {
                        q16_fn f16 = q16_pick(b_kc, epi, nw);
                        g.nbm = ceil_div(M, 256); g.nbn = ceil_div(N, nw * 32);
                        g.nsplit = 1; g.wide = 1;
                        const long total16 = (long)g.nbm * g.nbn;
                        const size_t shm16 = 10 * Q8_HALF;
                        static q16_fn attr16[16];
                        static int n_attr16 = 0;
                        bool seen16 = false;
                        for (int i = 0; i < n_attr16; ++i) seen16 = seen16 || attr16[i] == f16;
                        if (!seen16) {
                            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(f16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm16);
                            if (n_attr16 < 16) attr16[n_attr16++] = f16;
                        }
                        const bool prof16 = ecamp_prof_active();
                        if (prof16) {
                            char tag[40];
                            snprintf(tag, sizeof tag, "q16:%c:e%d:%ld:%ld:%ld:w%d", b_kc ? 'f' : 'd', epi, (long)M, (long)N, (long)K, nw * 32);
                            ecamp_prof_begin(ECAMP_PROF_GEMM_BF16, 2.0 * (double)M * (double)N * (double)K, stream, tag);
                        }
                        long grid16 = ncu16;
                        {   // data-gradient form beside a co-tenant: the same switch as the eight-wave kernel's (g_q8_bwd_grid below) -- with
                            // it on, one output tile per workgroup and the hardware dispatcher deals the tiles
                            static const int env_bwd16 = getenv("ECAMP_Q8_BWD_GRID") ? atoi(getenv("ECAMP_Q8_BWD_GRID")) : 0;
                            const int bg16 = g_q8_bwd_grid > 0 ? g_q8_bwd_grid : env_bwd16;
                            if (!b_kc && bg16 > 0) grid16 = bg16;
                        }
                        hipLaunchKernelGGL(f16, dim3((unsigned)(total16 < grid16 ? total16 : grid16)), dim3(256), shm16, stream, g);
                        ++g_q16_launches;
                        if (prof16) ecamp_prof_end(stream);
                        ECAMP_LAUNCH_CHECK();
                        return 0;
                    }
                }
            }
            q8_fn fn = q8_pick(a_kc, b_kc, epi, rowsum != nullptr);
            g.nbm = ceil_div(M, 256); g.nbn = ceil_div(N, 256);
            g.nsplit = split_k; g.wide = 1;
            int ncu = p8_num_cu();
            if (!(a_kc && b_kc) && g_p8_wgrad_reserve > 0 && ncu - g_p8_wgrad_reserve >= 64) ncu -= g_p8_wgrad_reserve;
            const long total8 = (long)g.nbm * g.nbn * split_k;
            const size_t shm = 10 * Q8_HALF;   // the whole 160 KB LDS of a CU
            static q8_fn attr_done[32];
            static int n_attr = 0;
            bool seen = false;
            for (int i = 0; i < n_attr; ++i) seen = seen || attr_done[i] == fn;
            if (!seen) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
                if (n_attr < 32) attr_done[n_attr++] = fn;
            }
            const bool prof8 = ecamp_prof_active();
            if (prof8) {   // tag: form (f = forward x w^T, d = data gradient dy w, w = weight gradient dy^T x), epilogue, M N K, split
                char tag[40];
                snprintf(tag, sizeof tag, "q8:%c:e%d:%ld:%ld:%ld:s%d", a_kc && b_kc ? 'f' : a_kc ? 'd' : 'w', epi, (long)M, (long)N, (long)K, split_k);
                ecamp_prof_begin(ECAMP_PROF_GEMM_BF16, 2.0 * (double)M * (double)N * (double)K, stream, tag);
            }
            {   // data-gradient form beside a co-tenant (see g_q8_bwd_grid): the hardware dispatcher deals the items
                static const int env_bwd = getenv("ECAMP_Q8_BWD_GRID") ? atoi(getenv("ECAMP_Q8_BWD_GRID")) : 0;
                const int bg = g_q8_bwd_grid > 0 ? g_q8_bwd_grid : env_bwd;
                if (a_kc && !b_kc && bg > 0) ncu = bg;
            }
            hipLaunchKernelGGL(fn, dim3((unsigned)(total8 < ncu ? total8 : ncu)), dim3(512), shm, stream, g);
            ++g_q8_launches;
            if (split_k > 1) {
                long n4 = M * N / 4;
                int nb = (int)((n4 + 255) / 256);
                if (nb > 2048) nb = 2048;
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb), dim3(256), 0, stream, splitk_ws, reinterpret_cast<float*>(C), (long)M, (long)N,
                                   (long)ldc, split_k, alpha, alpha_dev, accumulate);
            }
            if (prof8) ecamp_prof_end(stream);
            ECAMP_LAUNCH_CHECK();
            return 0;
        }
    }
#define LAUNCH(KERN)                                                             \
    do {                                                                         \
        if (a_kc && b_kc) hipLaunchKernelGGL((KERN<true, true, false>), grid, block, 0, stream, g);        \
        else if (a_kc && !b_kc) hipLaunchKernelGGL((KERN<true, false, false>), grid, block, 0, stream, g); \
        else if (!a_kc && b_kc) hipLaunchKernelGGL((KERN<false, true, false>), grid, block, 0, stream, g); \
        else if (rowsum) hipLaunchKernelGGL((KERN<false, false, true>), grid, block, 0, stream, g);        \
        else hipLaunchKernelGGL((KERN<false, false, false>), grid, block, 0, stream, g);                   \
    } while (0)
    const bool prof = ecamp_prof_active();
    if (prof) {
        char tag[40];
        snprintf(tag, sizeof tag, "t128:%c:e-:%ld:%ld:%ld:s%d", a_kc && b_kc ? 'f' : a_kc ? 'd' : 'w', (long)M, (long)N, (long)K, split_k);
        ecamp_prof_begin(dtype == ECAMP_BF16 ? ECAMP_PROF_GEMM_BF16 : ECAMP_PROF_GEMM_F32, 2.0 * (double)M * (double)N * (double)K, stream, tag);
    }
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

// =============================================================================================
// Grouped weight gradients: the (up to four) dW_p (+)= alpha * dY_p^T X_p of one transformer block -- same contraction length
// `rows`, different shapes and operands -- as ONE launch of the Q8 kernel's item-table form (gemm_q8.h, ITEMS) plus one grouped
// reduce.  Per-layer launches give every CU exactly one (tile, K slice) item: 2.5 us of ring fill and 6.5 us of f32 slab
// stores with nothing to overlap per ~90 us launch, and 7-28 slabs per output tile (256 MB written and re-read per encoder
// block).  Here the K tiles of ALL tiles of the block are dealt out evenly, in order, to the workgroups (ranges run across
// tile boundaries), so a tile has 2-4 slabs and a workgroup one ring fill per block.
#include <map>
#include <string>
#include <vector>
struct WgTile { int prob, m0, n0, first, count; };
struct WgRedProb { float* C; long ldc; int M, N; float alpha; int accumulate; const float* alpha_dev; };
struct WgRedArgs { const WgTile* tiles; const float* slabs; int ntiles, pad; WgRedProb p[4]; };
__global__ __launch_bounds__(256) void wgrad_group_reduce_kernel(WgRedArgs a) {
    const int t = blockIdx.x >> 4, chunk = blockIdx.x & 15;
    const WgTile T = a.tiles[t];
    const WgRedProb P = T.prob == 0 ? a.p[0] : T.prob == 1 ? a.p[1] : T.prob == 2 ? a.p[2] : a.p[3];
    const float al = P.alpha_dev ? P.alpha * P.alpha_dev[0] : P.alpha;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = chunk * 1024 + j * 256 + threadIdx.x;            // float4 index inside the 256 x 256 tile
        const int r = idx >> 6, c = (idx & 63) * 4;
        const int gm = T.m0 + r, gn = T.n0 + c;
        if (gm >= P.M || gn >= P.N) continue;
        const float4* src = reinterpret_cast<const float4*>(a.slabs + (long)T.first * 65536) + idx;
        float4 acc = src[0];
        for (int z = 1; z < T.count; ++z) {
            const float4 p = src[(long)z * 16384];
            acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
        float* c_ = P.C + (long)gm * P.ldc + gn;
        float4 o = P.accumulate ? *reinterpret_cast<float4*>(c_) : make_float4(0.f, 0.f, 0.f, 0.f);
        o.x += al * acc.x; o.y += al * acc.y; o.z += al * acc.z; o.w += al * acc.w;
        *reinterpret_cast<float4*>(c_) = o;
    }
}

// The item table of a group is pure host arithmetic on the shapes; it lives in a host-side cache here and in a device buffer the
// CALLER owns (ecamp_wgrad_group_table fills a host image the caller uploads once per shape set): the library never allocates
// device memory and never synchronises -- ecamp_wgrad_group only enqueues two kernels, so it can be captured into a HIP graph.
// Image layout: items [nitems] (32 B each) | wg_first [nwg + 1] int32, padded to 32 B | tiles [ntiles] (20 B each).
struct WgPlan {
    std::vector<Q8ItemRec> items;
    std::vector<int> first;
    std::vector<WgTile> tiles;
    int nitems = 0, nwg = 0, ntiles = 0;
    size_t off_first = 0, off_tiles = 0, bytes = 0;
};
static std::map<std::string, WgPlan> g_wg_plans;

static const WgPlan* wg_plan(int n, const int64_t* n_out, const int64_t* k_in, const unsigned* has_bias, int64_t rows, int ncu) {
    std::string key((const char*)n_out, n * sizeof(int64_t));
    key.append((const char*)k_in, n * sizeof(int64_t)).append((const char*)has_bias, n * sizeof(unsigned)).append((const char*)&rows, 8).append((const char*)&ncu, 4);
    key.append((const char*)&g_p8_wgrad_reserve, sizeof(int));   // the reserve caps the piece count
    auto it = g_wg_plans.find(key);
    if (it != g_wg_plans.end()) return &it->second;
    const long KT = (rows + 63) / 64;
    WgPlan pl;
    std::vector<WgTile>& tiles = pl.tiles;
    for (int p = 0; p < n; ++p)
        for (int mb = 0; mb < ceil_div(n_out[p], 256); ++mb)
            for (int nb = 0; nb < ceil_div(k_in[p], 256); ++nb) tiles.push_back({p, mb * 256, nb * 256, 0, 0});
    std::vector<Q8ItemRec>& items = pl.items;
    std::vector<int>& first = pl.first;
    // ECAMP_WGRAD_PLAN=0: the round-2 dealing (K tiles of all tiles dealt out evenly in tile order; ranges run across tile boundaries)
    static const int plan_mode = getenv("ECAMP_WGRAD_PLAN") ? atoi(getenv("ECAMP_WGRAD_PLAN")) : 1;
    const long T = (long)tiles.size();
    int cap = p8_num_cu();                            // workgroups the launch may use: every CU, minus the data-parallel reserve
    if (g_p8_wgrad_reserve > 0 && cap - g_p8_wgrad_reserve >= 64) cap -= g_p8_wgrad_reserve;
    long S = (ncu + T / 2) / T;                       // K segments per tile: one (tile, segment) piece per workgroup
    while (S > 1 && (T * S > cap || KT / S < 4)) --S;
    if (plan_mode == 1 && S >= 1 && T * S >= ncu / 2 && T * S <= cap) {
        // Segment-major plan: every output tile is cut into the same S contiguous K segments and each workgroup owns ONE piece.  The
        // pieces are ordered (segment, tile) and workgroup w -- which the hardware places on XCD w % 8 -- takes piece
        // (w % 8) * P/8 + w / 8: the ~P/8 workgroups of an XCD start together on neighbouring tiles of the SAME K segment, walk the
        // contraction in step and share the dY / X strips of each K tile in that XCD's L2 (a dY strip is common to all tiles of an
        // M-block row, an X strip to a column): with the dealing above every workgroup streamed its own two strips, 4x the operands.
        const long P = T * S;
        for (long t = 0; t < T; ++t) { tiles[t].first = (int)(t * S); tiles[t].count = (int)S; }
        std::vector<long> order(P);                   // order[j] = piece (segment-major) of the j-th position in XCD-contiguous order
        for (long j = 0; j < P; ++j) order[j] = j;
        const long q8 = P / 8, r8 = P % 8;
        for (long w = 0; w < P; ++w) {
            const long xcd = w % 8, slot = w / 8;
            const long j = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;   // bijective for any P (xcd_remap)
            const long seg = order[j] / T, t = order[j] % T;
            const long k0 = KT * seg / S, k1 = KT * (seg + 1) / S;
            const long kend = k1 * 64 < rows ? k1 * 64 : rows;
            WgTile& Tt = tiles[t];
            first.push_back((int)items.size());
            items.push_back({Tt.prob, Tt.m0, Tt.n0, (int)(k0 * 64), (int)kend, (int)(t * S + seg), (Tt.n0 == 0 && has_bias[Tt.prob]) ? 1 : 0, 0});
        }
    } else {
    // the unit of the dealing is a PAIR of K tiles (a tile's last unit also takes its odd K tile, if it has one: 197 K tiles for the
    // 12608 rows of ViT-L/448): every piece has >= 2 K tiles, whatever the row count
    const long U = KT / 2;
    const long units = (long)tiles.size() * U;
    long q = (units + ncu - 1) / ncu;
    if (q < 1) q = 1;
    long u = 0;
    while (u < units) {
        first.push_back((int)items.size());
        long take = q < units - u ? q : units - u;
        while (take > 0) {
            const long t = u / U, j0 = u % U, len = take < U - j0 ? take : U - j0;
            WgTile& T = tiles[t];
            if (T.count == 0) T.first = (int)items.size();
            ++T.count;
            const long k0 = 2 * j0, k1 = j0 + len == U ? KT : 2 * (j0 + len);
            const long kend = k1 * 64 < rows ? k1 * 64 : rows;
            items.push_back({T.prob, T.m0, T.n0, (int)(k0 * 64), (int)kend, (int)items.size(), (T.n0 == 0 && has_bias[T.prob]) ? 1 : 0, 0});
            u += len; take -= len;
        }
    }
    }
    first.push_back((int)items.size());
    pl.nitems = (int)items.size(); pl.nwg = (int)first.size() - 1; pl.ntiles = (int)tiles.size();
    pl.off_first = items.size() * sizeof(Q8ItemRec);
    pl.off_tiles = pl.off_first + (first.size() * sizeof(int) + 31) / 32 * 32;
    pl.bytes = pl.off_tiles + tiles.size() * sizeof(WgTile);
    return &(g_wg_plans[key] = std::move(pl));
}
// Workgroups of the grouped launch: three quarters of the CUs.  It runs beside the data-gradient chain of the next block; with one
// workgroup on every CU that chain's kernels wait for whole 250-us items (39.7-39.9 ms per step against 39.4 with per-layer
// launches), with 192 they do not (39.4-39.9 against 39.7-40.1 on the same boxes; 160: +0.6 ms, 64: +4 ms).  ECAMP_WGRAD_GROUP_CUS overrides.
static int wg_ncu() {
    static const int env = getenv("ECAMP_WGRAD_GROUP_CUS") ? atoi(getenv("ECAMP_WGRAD_GROUP_CUS")) : 0;
    if (env > 0) return env < p8_num_cu() ? env : p8_num_cu();   // (the workspace is sized for at most one workgroup per CU)
    int ncu = p8_num_cu();
    if (g_p8_wgrad_reserve > 0 && ncu - g_p8_wgrad_reserve >= 64) ncu -= g_p8_wgrad_reserve;
    const int q = p8_num_cu() * 3 / 4;
    return ncu < q ? ncu : q;
}
// 1 if the group can run as one launch (otherwise the caller issues per-layer ecamp_gemm calls)
extern "C" int ecamp_wgrad_group_supported(int32_t n, const int64_t* n_out, const int64_t* k_in, int64_t rows) {
    if (n < 1 || n > 4 || !n_out || !k_in || rows < 256) return 0;
    long tiles = 0;
    for (int p = 0; p < n; ++p) {
        if (n_out[p] % 8 || k_in[p] % 8 || n_out[p] <= 0 || k_in[p] <= 0) return 0;
        if (rows * n_out[p] * 2 > 0x7fffffffl || rows * k_in[p] * 2 > 0x7fffffffl) return 0;
        tiles += (long)ceil_div(n_out[p], 256) * ceil_div(k_in[p], 256);
    }
    return tiles >= 16 ? 1 : 0;
}
// bytes of `ws`: one 256 x 256 f32 slab per work item (0: not supported)
extern "C" int64_t ecamp_wgrad_group_workspace_bytes(int32_t n, const int64_t* n_out, const int64_t* k_in, int64_t rows) {
    if (!ecamp_wgrad_group_supported(n, n_out, k_in, rows)) return 0;
    long tiles = 0;
    for (int p = 0; p < n; ++p) tiles += (long)ceil_div(n_out[p], 256) * ceil_div(k_in[p], 256);
    return (tiles + p8_num_cu() + 1) * 262144;    // every workgroup boundary adds at most one piece
}
static long g_wg_launches = 0;
extern "C" int64_t ecamp_wgrad_group_launches(void) { return g_wg_launches; }
// workgroups a grouped launch will use for the caller's `workgroups` argument (0 = the library's choice): part of the table's identity
extern "C" int ecamp_wgrad_group_workgroups(int32_t workgroups) {
    if (workgroups > 0) return workgroups < p8_num_cu() ? workgroups : p8_num_cu();
    return wg_ncu();
}
// upper bound of the item-table image for any workgroup count (0: group not supported)
extern "C" int64_t ecamp_wgrad_group_table_bytes(int32_t n, const int64_t* n_out, const int64_t* k_in, int64_t rows) {
    if (!ecamp_wgrad_group_supported(n, n_out, k_in, rows)) return 0;
    long tiles = 0;
    for (int p = 0; p < n; ++p) tiles += (long)ceil_div(n_out[p], 256) * ceil_div(k_in[p], 256);
    const long ncu = p8_num_cu();
    return (tiles + ncu + 1) * (long)sizeof(Q8ItemRec) + ((ncu + 2) * 4 + 31) / 32 * 32 + tiles * (long)sizeof(WgTile) + 64;
}
// fills `host_table` (HOST memory, ecamp_wgrad_group_table_bytes bytes) with the item table of this group for
// ecamp_wgrad_group_workgroups(workgroups) workgroups; returns the bytes used (< 0: error).  has_bias[p] != 0: gb[p] will be given.
extern "C" int64_t ecamp_wgrad_group_table(int32_t n, const int64_t* n_out, const int64_t* k_in, const int32_t* has_bias, int64_t rows,
                                           int32_t workgroups, void* host_table) {
    if (!n_out || !k_in || !has_bias || !host_table || !ecamp_wgrad_group_supported(n, n_out, k_in, rows))
        return ecamp_set_error(-1, "wgrad_group_table: unsupported group or null pointer");
    unsigned hb[4] = {0, 0, 0, 0};
    for (int p = 0; p < n; ++p) hb[p] = has_bias[p] ? 1u : 0u;
    const WgPlan* pl = wg_plan(n, n_out, k_in, hb, rows, ecamp_wgrad_group_workgroups(workgroups));
    char* out = (char*)host_table;
    memcpy(out, pl->items.data(), pl->items.size() * sizeof(Q8ItemRec));
    memcpy(out + pl->off_first, pl->first.data(), pl->first.size() * sizeof(int));
    memcpy(out + pl->off_tiles, pl->tiles.data(), pl->tiles.size() * sizeof(WgTile));
    return (int64_t)pl->bytes;
}
// gw[p] [n_out[p], k_in[p]] (f32, contiguous) (+)= alpha * dy[p]^T x[p];  gb[p] [n_out[p]] (f32 or null) += alpha * column sums of dy[p].
// dy[p] [rows, n_out[p]], x[p] [rows, k_in[p]] bf16, row-contiguous.  accumulate[p] = 0 overwrites gw[p].
extern "C" int ecamp_wgrad_group(int32_t n, const void* const* dy, const void* const* x, float* const* gw, float* const* gb, const int64_t* n_out,
                                 const int64_t* k_in, int64_t rows, float alpha, const float* alpha_dev, const int32_t* accumulate, float* ws,
                                 const void* table, int64_t table_bytes, int32_t workgroups, hipStream_t stream) {
    ECAMP_CHECK_ARG(dy && x && gw && gb && n_out && k_in && accumulate && ws && table, "wgrad_group: null pointer");
    ECAMP_CHECK_ARG(((uintptr_t)table & 31) == 0, "wgrad_group: the item table must be 32-byte aligned");
    ECAMP_CHECK_ARG(ecamp_wgrad_group_supported(n, n_out, k_in, rows), "wgrad_group: unsupported group (1-4 layers, rows >= 256, dims %% 8 == 0, < 2 GB operands)");
    unsigned hb[4] = {0, 0, 0, 0};
    bool any_bias = false;
    for (int p = 0; p < n; ++p) {
        ECAMP_CHECK_ARG(dy[p] && x[p] && gw[p], "wgrad_group: null operand");
        ECAMP_CHECK_ARG(((uintptr_t)dy[p] & 15) == 0 && ((uintptr_t)x[p] & 15) == 0 && ((uintptr_t)gw[p] & 15) == 0, "wgrad_group: operands must be 16-byte aligned");
        hb[p] = gb[p] ? 1u : 0u;
        any_bias = any_bias || gb[p];
    }
    const int nwg = ecamp_wgrad_group_workgroups(workgroups);   // caller's choice (how much of the chip to leave to what runs beside it) or the default
    const WgPlan* pl = wg_plan(n, n_out, k_in, hb, rows, nwg);   // host arithmetic (cached): counts and offsets of the caller's device image
    // the device image must be the one ecamp_wgrad_group_table wrote for THIS plan: a process-wide switch flipped through the raw C
    // ecamp_set_option between building the table and this call changes the piece count, and with it every offset below
    ECAMP_CHECK_ARG(table_bytes == (int64_t)pl->bytes, "wgrad_group: the item table (%ld bytes) was not built for the current plan (%ld bytes: CU reserve "
                    "or workgroup count changed since ecamp_wgrad_group_table) -- rebuild it", (long)table_bytes, (long)pl->bytes);
    const char* tb = (const char*)table;
    Q8Group G;
    memset(&G, 0, sizeof(G));
    G.items = (const Q8ItemRec*)tb; G.wg_first = (const int*)(tb + pl->off_first); G.slabs = ws; G.nprob = n;
    WgRedArgs R;
    memset(&R, 0, sizeof(R));
    R.tiles = (const WgTile*)(tb + pl->off_tiles); R.slabs = ws; R.ntiles = pl->ntiles;
    for (int p = 0; p < n; ++p) {
        G.p[p].A = dy[p]; G.p[p].B = x[p]; G.p[p].lda = n_out[p]; G.p[p].ldb = k_in[p]; G.p[p].M = (int)n_out[p]; G.p[p].N = (int)k_in[p];
        G.p[p].rowsum = gb[p]; G.p[p].alpha_out = alpha; G.p[p].alpha_dev_out = alpha_dev;
        R.p[p].C = gw[p]; R.p[p].ldc = k_in[p]; R.p[p].M = (int)n_out[p]; R.p[p].N = (int)k_in[p]; R.p[p].alpha = alpha; R.p[p].alpha_dev = alpha_dev;
        R.p[p].accumulate = accumulate[p];
    }
    GemmArgs g;   // the item-table form takes operands and shapes from the group; these only keep the kernel's unused set-up code in range
    memset(&g, 0, sizeof(g));
    g.A = dy[0]; g.B = x[0]; g.C = ws; g.M = (int)n_out[0]; g.N = (int)k_in[0]; g.K = (int)rows; g.lda = n_out[0]; g.ldb = k_in[0]; g.ldc = k_in[0];
    g.out_f32 = 1; g.alpha = 1.0f; g.alpha_out = alpha; g.nbm = 1; g.nbn = 1; g.nsplit = 1; g.k_per_split = (int)rows; g.wide = 1; g.partial = ws;
    const size_t shm = 10 * Q8_HALF;
    typedef void (*q8i_fn)(GemmArgs, Q8Group);
    const bool lean = (q8_sch() & 8) != 0;
    const q8i_fn fn = any_bias ? (lean ? (q8i_fn)gemm_bf16_q8_items_kernel<true, 1> : (q8i_fn)gemm_bf16_q8_items_kernel<true, 0>)
                               : (lean ? (q8i_fn)gemm_bf16_q8_items_kernel<false, 1> : (q8i_fn)gemm_bf16_q8_items_kernel<false, 0>);
    static bool attr[4] = {false, false, false, false};
    if (!attr[any_bias + 2 * lean]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        attr[any_bias + 2 * lean] = true;
    }
    double work = 0.0;
    for (int p = 0; p < n; ++p) work += 2.0 * (double)n_out[p] * (double)k_in[p] * (double)rows;
    const bool prof = ecamp_prof_active();
    if (prof) {
        char tag[40];
        snprintf(tag, sizeof tag, "grp:w:n%d:%ld:%ld:%ld:wg%d", n, (long)rows, (long)n_out[0], (long)k_in[0], pl->nwg);
        ecamp_prof_begin(ECAMP_PROF_GEMM_BF16, work, stream, tag);
    }
    hipLaunchKernelGGL(fn, dim3(pl->nwg), dim3(512), shm, stream, g, G);
    g_q8_launches += n;
    ++g_wg_launches;
    hipLaunchKernelGGL(wgrad_group_reduce_kernel, dim3(pl->ntiles * 16), dim3(256), 0, stream, R);
    if (prof) ecamp_prof_end(stream);
    ECAMP_LAUNCH_CHECK();
    return 0;
}
