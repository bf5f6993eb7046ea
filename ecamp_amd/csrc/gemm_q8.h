// "Q8": persistent 256x256x64 bf16 GEMM on v_mfma_f32_32x32x16_bf16, eight waves (2 x 4) of 128x64, one workgroup per CU.
//
// Structure (round 2; replaces the K=32 DMA-ring kernel "P8" of round 1 wherever it applied):
//  * K tile = 64, two K tiles resident in LDS (128 KB).  Each operand tile is two HALF-TILES of 128 rows (A_0/A_1: the rows of
//    wave row 0/1; B_0/B_1: the columns of wave columns 0-1 / 2-3), 16 KB each, filled by direct L2->LDS DMA
//    (buffer_load_dwordx4 ... lds: 8 instructions per wave per K tile) through per-half-tile buffer descriptors built with scalar
//    ALU only: the per-lane offsets are kernel constants and rows / contraction steps past the end read as zero.
//  * A K tile is FOUR STEPS per wave, one 64x32 quadrant of the wave's 128x64 block each: {4, 8 or 12 ds_read_b128; 8 MFMAs of
//    32x32x16 = 256 matrix cycles}, quadrants (m0,n0) (m0,n1) (m1,n1) (m1,n0) so that consecutive steps share one operand.
//  * ONE s_barrier per K tile.  Wave row 0 meets it after its last MFMA step, wave row 1 BEFORE its last MFMA step: after every
//    barrier the two waves of a SIMD (w and w+4) start in opposite halves of a step -- one multiplies while the other reads LDS /
//    issues DMA -- and nothing re-synchronises them inside the K tile, so the matrix pipe is handed back and forth by
//    the hardware arbiter instead of by barriers (the first version had two barriers per step: 130 idle matrix cycles each).
//  * The DMA of K tile t+1 is issued right after the barrier that retires K tile t-1 (whose slots it overwrites) and is waited for,
//    with a counted s_waitcnt, just before the next barrier: a full K tile of latency cover.  The K-tile stream is ONE flat
//    sequence over all the output tiles a workgroup processes, so the next tile's first K tile is in LDS before the current
//    tile's epilogue starts.
//  * Epilogue in four pieces: quadrant q of an output tile is final after step q of the tile's last K tile and is stored right
//    after that step (the last one after the barrier, in the first K tile of the next output tile, whose MFMAs start from C = 0).
//    The stores are buffer stores (bounds by descriptor, no exec-masked branches), so their COUNT is exact and the DMA wait
//    can be "all but the n stores issued after the DMA": vmcnt retires in order on gfx9 and loads and stores share it -- a
//    counted wait is never early, and with the DMA issued ahead of a K tile's stores it never waits for fresh stores either.
//  * LDS images are DMA-linear (128-B rows, 8 rows per wave piece); the bank swizzle (16-B chunk ^ ((row >> 1) & 7)) is applied
//    to the lane's SOURCE offset and again on the fragment reads (conflict-free for the 32-row b128 fragments, both row maps).
//  * N-side fragment row i is mapped to tile column (i&3) | i3<<2 | i2<<3 | i4<<4, so that with the N fragment as the MFMA's
//    A operand a lane ends up with 8 consecutive output columns per 8 accumulator registers: every epilogue access is 16 B.
#pragma once
#include "gemm_args.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 hw_bf16x8;
typedef __attribute__((ext_vector_type(4))) short q8_v4s16;

#define Q8_HALF 16384
#define Q8_GLDS16(SRC, DST) \
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(SRC), (void __attribute__((address_space(3)))*)(DST), 16, 0, 0)

static __device__ __attribute__((aligned(16))) unsigned int q8_zero16[4] = {0u, 0u, 0u, 0u};

// one output tile (and split-K slice) of the persistent kernel; every field is wave-uniform
struct Q8Item {
    int m0, n0, kbeg, kend, nt, z, ncol;
};
__device__ __forceinline__ Q8Item q8_decode(const GemmArgs& g, int v, int total) {
    const unsigned f = (unsigned)xcd_remap(v, total), ntile = (unsigned)(g.nbm * g.nbn);
    // grouped order inside a split: 8 M-blocks are walked for one N-block before the next N-block, so the ~32 tiles an XCD works
    // on at a time form an 8 x 4 patch and consecutive rounds keep the 8 M panels in its L2
    const unsigned z = f / ntile, tile = f - z * ntile;
    const unsigned gw = 8u * (unsigned)g.nbn, grp = tile / gw, in = tile - grp * gw, first = grp * 8u;
    const unsigned gsz = min(8u, (unsigned)g.nbm - first);
    const unsigned nb = in / gsz, mb = first + (in - nb * gsz);
    Q8Item it;
    it.m0 = (int)mb * 256; it.n0 = (int)nb * 256; it.z = (int)z; it.ncol = (int)nb;
    it.kbeg = (int)z * g.k_per_split;
    it.kend = min(g.K, it.kbeg + g.k_per_split);
    it.nt = (it.kend - it.kbeg + 63) >> 6;
    return it;
}

// ---- DMA of one half-tile (16 KB = 16 wave pieces of 1 KB; every wave issues pieces `wave` and `8 + wave`) ----------------------
// buffer_load_dwordx4 ... lds through a buffer descriptor that is rebuilt (scalar ALU only) for every half-tile: base = first
// element of the half-tile, num_records = bytes from there to the end of the operand's valid range.  The per-lane offsets are
// computed ONCE per kernel (they only depend on the lane and the leading dimension); rows past the end of the matrix (kc) and
// contraction rows past kend (oc) fall outside the descriptor and read as zero, so the loop has no clamps and no selects.
//   kc operand P[row*ld + k]: piece = 8 rows x 128 B; LDS position (row, j) holds global 16-B chunk j ^ ((row>>1)&7)
//   oc operand P[k*ld + row]: half-tile kept as it lies in HBM, 64 k-rows x 256 B; piece = 4 k-rows; position (kr, j) holds
//                             chunk j ^ ((kr&3)<<2), which puts the four k-rows of a transpose-read block on disjoint banks
template <bool KC>
__device__ __forceinline__ unsigned q8_voff(int i, int wave, int lane, long ld) {
    const int pi = i * 8 + wave;
    if (KC) {
        const int row = pi * 8 + (lane >> 3);
        const int kc = (lane & 7) ^ ((row >> 1) & 7);
        return (unsigned)(((long)row * ld + kc * 8) * 2);
    } else {
        const int kr = pi * 4 + (lane >> 4);
        const int oc = (lane & 15) ^ ((kr & 3) << 2);
        return (unsigned)(((long)kr * ld + oc * 8) * 2);
    }
}
// `base`/`rec`: wave-uniform first byte of the half-tile and bytes from there to the end of the valid range (<= 0: nothing valid);
// `krem`: contraction elements left from this K tile's first column (kc operands: chunks at k >= krem read as zero)
template <bool KC, int PIECES = 3>
__device__ __forceinline__ void q8_stage_half(const unsigned char* base, int rec, int krem, unsigned char* dst,
                                              const unsigned (&voff)[2], int wave, int lane) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, rec < 0 ? 0 : rec, 0x00020000);
    unsigned v0 = voff[0], v1 = voff[1];
    if (KC && krem < 64) {   // last, partial K tile (uniform branch)
        const int kc0 = (lane & 7) ^ (((wave * 8 + (lane >> 3)) >> 1) & 7);   // same key for both pieces (64 rows apart)
        if (kc0 * 8 >= krem) { v0 = 0xFFFFFF00u; v1 = 0xFFFFFF00u; }
    }
    typedef void __attribute__((address_space(3))) lds_void;
    if (PIECES & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + wave * 1024), 16, (int)v0, 0, 0, 0);
    if (PIECES & 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + 8192 + wave * 1024), 16, (int)v1, 0, 0, 0);
}

// ---- epilogue of 8 consecutive outputs of one row (the host only selects this kernel when every [M, ld] epilogue operand is
// 16-B aligned at 8-column granularity, N % 8 == 0 and every matrix is < 2 GB).  EPI is a compile-time selection of what the
// epilogue can do -- with every option tested at run time the epilogue's branches push the kernel over its 256 registers:
//   0  bf16 C = alpha*acc (+bias)            1  ... + save pre-activation + exact GELU        2  ... + residual
//   3  bf16 C = alpha*acc * gelu'(gmul) (+residual)                                            4  f32: split-K slab, or C (+= old)
// Stores per quadrant (exact, whatever the bounds): 4 for EPI 0/2/3, 8 for EPI 1/4.
template <int EPI> struct Q8Epi { static constexpr int NST = (EPI == 1 || EPI == 4) ? 8 : 4; };
typedef unsigned int q8_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ q8_u32x4 q8_pack8(const float (&v)[8]) {
    return (q8_u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
}

// DBG bits (development, template parameter): 1 = no MFMA, 2 = no DMA, 4 = no epilogue, 8 = no s_setprio, 16 = no barrier in the
// loop, 32 = no fragment reads (timing decomposition only: 16 and 32 give wrong results)
template <bool A_KC, bool B_KC, int EPI, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_bf16_q8_kernel(GemmArgs g) {
    constexpr int NST = Q8Epi<EPI>::NST;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // K tile 0 {A_0 A_1 B_0 B_1} | K tile 1; the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int total = g.nbm * g.nbn * g.nsplit, G = (int)gridDim.x;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);
    const int l31 = lane & 31, lh = lane >> 5;
    // N-side fragment row -> tile column (see header)
    const int ncol = (l31 & 3) | (((l31 >> 3) & 1) << 2) | (((l31 >> 2) & 1) << 3) | ((l31 >> 4) << 4);

    // per-lane fragment offsets inside a half-tile (without K-tile base and quadrant offset)
    //   kc: row*128 + ((2*ks + lh) ^ key(row)) * 16, one per k-step (the XOR does not commute with the k-step offset)
    //   oc: (ks*16 + 8*kb + r)*256 + ((chunk ^ (r<<2)) * 16) + within, one per 32-output tile index (the XOR touches the tile bits)
    unsigned offM[4], offN[4];
    if (A_KC) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) offM[ks] = (unsigned)(wr * Q8_HALF + l31 * 128 + (((2 * ks + lh) ^ ((l31 >> 1) & 7)) << 4));
    } else {
        const int i16 = lane & 15, ob = (lane >> 4) & 1, kb = lane >> 5, r = i16 >> 2, q = i16 & 3;
#pragma unroll
        for (int t = 0; t < 4; ++t) {   // tile index t = 2*mh + tm of the wave's 128 rows
            const int col = t * 32 + 16 * ob + 4 * q;
            offM[t] = (unsigned)(wr * Q8_HALF + (8 * kb + r) * 256 + ((((col >> 3) ^ (r << 2)) & 15) << 4) + (col & 7) * 2);
        }
    }
    if (B_KC) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            offN[ks] = (unsigned)((2 + (wc >> 1)) * Q8_HALF + ((wc & 1) * 64 + ncol) * 128 + (((2 * ks + lh) ^ ((ncol >> 1) & 7)) << 4));
    } else {
        const int i16 = lane & 15, ob = (lane >> 4) & 1, kb = lane >> 5, r = i16 >> 2, q = i16 & 3;
#pragma unroll
        for (int t = 0; t < 2; ++t) {   // t = nh; the pointer of quarter q covers the 4 outputs at 4*(q>>1) + 8*(q&1) (column remap)
            const int col = (wc & 1) * 64 + t * 32 + 16 * ob + 4 * (q >> 1) + 8 * (q & 1);
            offN[t] = (unsigned)((2 + (wc >> 1)) * Q8_HALF + (8 * kb + r) * 256 + ((((col >> 3) ^ (r << 2)) & 15) << 4) + (col & 7) * 2);
        }
        offN[2] = offN[3] = 0;
    }

    f32x16 acc[4][2];

    // ---- DMA cursor over the flat K-tile stream; all of it wave-uniform (SGPRs): byte cursors of the A and B half-tile 0 of the
    // K tile to stage next, bytes left in their valid ranges, contraction elements left in the staged output tile
    int pv = (int)blockIdx.x;
    bool pdone = pv >= total;
    const unsigned char *sa_base, *sb_base;
    int sa_rec, sb_rec, p_krem;
    const int a_half = A_KC ? (int)g.lda * 256 : 256, a_step = A_KC ? 128 : (int)g.lda * 128;
    const int b_half = B_KC ? (int)g.ldb * 256 : 256, b_step = B_KC ? 128 : (int)g.ldb * 128;
#define Q8_NEXT_ITEM()                                                                                                   \
    do {                                                                                                                 \
        const Q8Item n_ = q8_decode(g, pv, total);                                                                       \
        p_krem = n_.kend - n_.kbeg;                                                                                      \
        if (A_KC) { sa_base = (const unsigned char*)(A + ((long)n_.m0 * g.lda + n_.kbeg)); sa_rec = (int)((((long)(g.M - n_.m0)) * g.lda - n_.kbeg) * 2); } \
        else      { sa_base = (const unsigned char*)(A + ((long)n_.kbeg * g.lda + n_.m0)); sa_rec = (int)(((long)p_krem * g.lda - n_.m0) * 2); }             \
        if (B_KC) { sb_base = (const unsigned char*)(B + ((long)n_.n0 * g.ldb + n_.kbeg)); sb_rec = (int)((((long)(g.N - n_.n0)) * g.ldb - n_.kbeg) * 2); } \
        else      { sb_base = (const unsigned char*)(B + ((long)n_.kbeg * g.ldb + n_.n0)); sb_rec = (int)(((long)p_krem * g.ldb - n_.n0) * 2); }             \
    } while (0)
    if (!pdone) Q8_NEXT_ITEM();
    int wslot = 0;                                     // K-tile slot (0/1) the next staged K tile goes to
    unsigned voffA[2], voffB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { voffA[i] = q8_voff<A_KC>(i, wave, lane, g.lda); voffB[i] = q8_voff<B_KC>(i, wave, lane, g.ldb); }
    // stage pieces PCS (bit 0: piece `wave`, bit 1: piece `8 + wave`) of half-tile PART (0: A_0, 1: A_1, 2: B_0, 3: B_1) of the next
    // K tile of the stream; the cursor moves on with the second piece of part 3
#define Q8_STAGE_PCS(PART, PCS)                                                                                          \
    do {                                                                                                                 \
        if (!pdone) {                                                                                                    \
            unsigned char* d_ = lds + wslot * (4 * Q8_HALF) + (PART) * Q8_HALF;                                          \
            if (!(DBG & 2)) {                                                                                            \
                if ((PART) == 0) q8_stage_half<A_KC, (PCS)>(sa_base, sa_rec, p_krem, d_, voffA, wave, lane);             \
                if ((PART) == 1) q8_stage_half<A_KC, (PCS)>(sa_base + a_half, sa_rec - a_half, p_krem, d_, voffA, wave, lane);  \
                if ((PART) == 2) q8_stage_half<B_KC, (PCS)>(sb_base, sb_rec, p_krem, d_, voffB, wave, lane);             \
                if ((PART) == 3) q8_stage_half<B_KC, (PCS)>(sb_base + b_half, sb_rec - b_half, p_krem, d_, voffB, wave, lane);  \
            }                                                                                                            \
            if ((PART) == 3 && ((PCS) & 2)) {                                                                            \
                wslot ^= 1;                                                                                              \
                p_krem -= 64;                                                                                            \
                sa_base += a_step; sa_rec -= a_step; sb_base += b_step; sb_rec -= b_step;                                \
                if (p_krem <= 0) {                                                                                       \
                    pv += G;                                                                                             \
                    if (pv < total) Q8_NEXT_ITEM(); else pdone = true;                                                   \
                }                                                                                                        \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)
#define Q8_STAGE_PART(PART) Q8_STAGE_PCS(PART, 3)
    // every DMA of mine has landed, except that the `N_` youngest vector-memory operations (stores issued after it) may be pending
#define Q8_WAIT_DMA(N_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory")

    hw_bf16x8 xm[4] = {}, xn[2] = {}, ym[4] = {}, yn[2] = {};   // two fragment groups (one k-step of 16 each): 4 M-side + 2 N-side
    int rslot = 0;                                    // K-tile slot being multiplied

    // fragment I (0..5, in the order the MFMAs consume them: n0 m0 m1 n1 m2 m3) of k-step KS of the K tile at SK into group (FM, FN)
#define Q8_RD1(FM, FN, SK, KS, I)                                                                                        \
    do {                                                                                                                 \
        constexpr int isn_ = ((I) == 0 || (I) == 3), idx_ = (I) == 0 ? 0 : (I) == 3 ? 1 : (I) < 3 ? (I) - 1 : (I) - 2;   \
        if (!(DBG & 32)) {                                                                                               \
            if (isn_) {                                                                                                  \
                if (B_KC) FN[idx_] = *reinterpret_cast<const hw_bf16x8*>((SK) + offN[KS] + idx_ * 32 * 128);             \
                else {                                                                                                   \
                    const unsigned char* p_ = (SK) + offN[idx_] + (KS) * 16 * 256;                                       \
                    q8_v4s16 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((q8_v4s16 __attribute__((address_space(3)))*)(p_));            \
                    q8_v4s16 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((q8_v4s16 __attribute__((address_space(3)))*)(p_ + 4 * 256));  \
                    FN[idx_] = __builtin_bit_cast(hw_bf16x8, (bf16x8){lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]});   \
                }                                                                                                        \
            } else {                                                                                                     \
                if (A_KC) FM[idx_] = *reinterpret_cast<const hw_bf16x8*>((SK) + offM[KS] + idx_ * 32 * 128);             \
                else {                                                                                                   \
                    const unsigned char* p_ = (SK) + offM[idx_] + (KS) * 16 * 256;                                       \
                    q8_v4s16 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((q8_v4s16 __attribute__((address_space(3)))*)(p_));            \
                    q8_v4s16 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((q8_v4s16 __attribute__((address_space(3)))*)(p_ + 4 * 256));  \
                    FM[idx_] = __builtin_bit_cast(hw_bf16x8, (bf16x8){lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]});   \
                }                                                                                                        \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)
#define Q8_READ_GROUP(FM, FN, SK, KS) \
    do { Q8_RD1(FM, FN, SK, KS, 0); Q8_RD1(FM, FN, SK, KS, 1); Q8_RD1(FM, FN, SK, KS, 2); Q8_RD1(FM, FN, SK, KS, 3); Q8_RD1(FM, FN, SK, KS, 4); Q8_RD1(FM, FN, SK, KS, 5); } while (0)
    // MFMA J (0..7) of a group: quadrants in the order (m0 n0) (m0 n1) (m1 n1) (m1 n0), two 32x32 tiles each
#define Q8_MFMA1(FM, FN, J, ZERO)                                                                                        \
    do {                                                                                                                 \
        constexpr int q_ = (J) >> 1, mh_ = q_ >> 1, nh_ = (q_ == 1 || q_ == 2) ? 1 : 0, tm_ = 2 * mh_ + ((J) & 1);       \
        if (DBG & 1) {                                                                                                   \
            asm volatile("" ::"v"(FN[nh_]), "v"(FM[tm_]));                                                               \
            if (ZERO) acc[tm_][nh_] = zero16;                                                                            \
        } else {                                                                                                         \
            acc[tm_][nh_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FN[nh_], FM[tm_], (ZERO) ? zero16 : acc[tm_][nh_], 0, 0, 0); \
        }                                                                                                                \
    } while (0)

    // ---- epilogue ---------------------------------------------------------------------------------------------------------------
    // buffer descriptors of the outputs (kernel constants; every byte offset fits 32 bits, checked by the host)
    const int esz = EPI == 4 ? 4 : 2;
    const long ldo = (EPI == 4 && g.partial) ? (long)g.N : g.ldc;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(
        (EPI == 4 && g.partial) ? (void*)g.partial : g.C, 0,
        (int)(unsigned)(((EPI == 4 && g.partial) ? (long)g.nsplit * g.M : (long)g.M) * ldo * esz), 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(EPI == 1 ? g.pre_out : g.C, 0, (int)(unsigned)((long)g.M * g.ldp * 2), 0x00020000);
    const unsigned lane_o = (unsigned)((l31 * ldo + 8 * lh) * esz);          // lane part of an output offset
    const unsigned lane_p = (unsigned)((l31 * g.ldp + 8 * lh) * 2);
    const unsigned lane_g = (unsigned)((l31 * g.ldg + 8 * lh) * 2);
    const unsigned lane_r = (unsigned)((l31 * g.ldr + 8 * lh) * 2);
    const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 3 ? g.gmul : g.C), 0, (int)(unsigned)((long)g.M * g.ldg * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)((EPI == 2 || EPI == 3) && g.residual ? g.residual : g.C), 0,
                                                                        (int)(unsigned)((long)g.M * g.ldr * 2), 0x00020000);
    // quadrant (MH, NH) of the output tile at (tm0, tn0), split tz: 2 tiles x 2 groups of 8 consecutive columns per lane.
    // The accumulators are only READ: the first K tile of the next output tile starts from C = 0 in the MFMA itself.
    auto store_quadrant = [&](int tm0, int tn0, int tz, auto mh_c, auto nh_c) {
        constexpr int MH = decltype(mh_c)::value, NH = decltype(nh_c)::value;
        if (DBG & 4) {   // keep the accumulators (and so the MFMAs) alive without storing them
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) asm volatile("" ::"v"(acc[2 * MH + tm][NH]));
            return;
        }
        const int mb = tm0 + wr * 128 + MH * 64;             // wave-uniform first row / column of the quadrant
        const int nb = tn0 + wc * 64 + NH * 32;
        const bool edge = nb + 32 > g.N;                     // uniform: some groups of 8 lie past N
        // wave-uniform addresses in the constant address space, pinned to SGPRs: scalar loads (lgkmcnt) that do not touch the DMA
        // queue's vmcnt (a vector load here makes hipcc drain the whole queue with vmcnt(0))
        typedef const float __attribute__((address_space(4))) cfloat4;
        float al = g.alpha;
        if (g.alpha_dev) {
            float ad = *(cfloat4*)g.alpha_dev;
            asm volatile("" : "+s"(ad));
            al *= ad;
        }
        float bias[2][8];
#pragma unroll
        for (int gq = 0; gq < 2; ++gq)
#pragma unroll
            for (int r = 0; r < 8; ++r) bias[gq][r] = 0.f;
        if (EPI <= 2 && g.bias) {
            // each group of 8 is clamped on its own, so groups inside N read exactly their columns
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                cfloat4* b0 = (cfloat4*)(g.bias + min(nb + 16 * gq, g.N - 8));
                cfloat4* b1 = (cfloat4*)(g.bias + min(nb + 16 * gq + 8, g.N - 8));
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float x0 = b0[r], x1 = b1[r];
                    asm volatile("" : "+s"(x0), "+s"(x1));
                    bias[gq][r] = lh ? x1 : x0;
                }
            }
        }
        // byte offsets: uniform tile part + lane part; a group past N gets offset 2^31, past every descriptor (< 2 GB, host-checked): dropped / reads 0
        unsigned uo[2][2], up[2][2], ug[2][2], ur[2][2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                const long row = mb + tm * 32, col = nb + 16 * gq;
                const bool oob = edge && (nb + 16 * gq + 8 * lh >= g.N);
                uo[tm][gq] = oob ? 0x80000000u : (unsigned)(((EPI == 4 && g.partial ? (long)tz * g.M : 0) + row) * ldo + col) * esz + lane_o;
                if (EPI == 1) up[tm][gq] = oob ? 0x80000000u : (unsigned)((row * g.ldp + col) * 2) + lane_p;
                if (EPI == 3) ug[tm][gq] = oob ? 0x80000000u : (unsigned)((row * g.ldg + col) * 2) + lane_g;
                if (EPI == 2 || EPI == 3) ur[tm][gq] = oob ? 0x80000000u : (unsigned)((row * g.ldr + col) * 2) + lane_r;
            }
        // all loads of the quadrant ahead of its first store
        q8_u32x4 qg[2][2], qr[2][2], qo[2][2][2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                if (EPI == 3) qg[tm][gq] = __builtin_amdgcn_raw_buffer_load_b128(rG, ug[tm][gq], 0, 0);
                if (EPI == 2 || (EPI == 3 && g.residual)) qr[tm][gq] = __builtin_amdgcn_raw_buffer_load_b128(rR, ur[tm][gq], 0, 0);
                if (EPI == 4 && !g.partial && g.accumulate) {
                    qo[tm][gq][0] = __builtin_amdgcn_raw_buffer_load_b128(rC, uo[tm][gq], 0, 0);
                    qo[tm][gq][1] = __builtin_amdgcn_raw_buffer_load_b128(rC, uo[tm][gq] + 16, 0, 0);
                }
            }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = acc[2 * MH + tm][NH][8 * gq + r] * al;
                if (EPI <= 2 && g.bias) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += bias[gq][r];
                }
                if (EPI == 1) {
                    __builtin_amdgcn_raw_buffer_store_b128(q8_pack8(v), rP, up[tm][gq], 0, 0);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = gelu_t<bf16_t>(rnd<bf16_t>(v[r]));
                }
                if (EPI == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[2 * r] *= gelu_grad_t<bf16_t>(__uint_as_float(qg[tm][gq][r] << 16));
                        v[2 * r + 1] *= gelu_grad_t<bf16_t>(__uint_as_float(qg[tm][gq][r] & 0xffff0000u));
                    }
                }
                if (EPI == 2 || (EPI == 3 && g.residual)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[2 * r] += __uint_as_float(qr[tm][gq][r] << 16);
                        v[2 * r + 1] += __uint_as_float(qr[tm][gq][r] & 0xffff0000u);
                    }
                }
                if (EPI == 4) {
                    if (!g.partial && g.accumulate) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[r] += __uint_as_float(qo[tm][gq][0][r]); v[4 + r] += __uint_as_float(qo[tm][gq][1][r]); }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128((q8_u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rC, uo[tm][gq], 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128((q8_u32x4){__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, rC, uo[tm][gq] + 16, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(q8_pack8(v), rC, uo[tm][gq], 0, 0);
                }
            }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define Q8_STORE_Q(TM0, TN0, TZ, Q)                                                           \
    do {                                                                                      \
        if ((Q) == 0) store_quadrant(TM0, TN0, TZ, I0(), I0());                               \
        if ((Q) == 1) store_quadrant(TM0, TN0, TZ, I0(), I1());                               \
        if ((Q) == 2) store_quadrant(TM0, TN0, TZ, I1(), I1());                               \
        if ((Q) == 3) store_quadrant(TM0, TN0, TZ, I1(), I0());                               \
    } while (0)
#define Q8_SB() __builtin_amdgcn_sched_barrier(0)
    // one group of a K tile: 8 MFMAs on group (CM, CN) with the six fragment reads of the NEXT group (NM, NN) <- k-step NKS of the
    // K tile at NSK issued one by one between them.  DA / DB: DMA half-tile parts of this group (-1: none), one instruction per
    // slot; wave row 0 issues in the slots after MFMA 0-3, wave row 1 after MFMA 4-7: an LDS-DMA issue blocks its wave for
    // 60-180 cycles (the CU's address path takes one wave instruction at a time) and the two waves of a SIMD are in lock step,
    // so with both in the same slot the matrix pipe would idle; this way the SIMD's other wave multiplies meanwhile.
    // EPI_HOOK: the previous output tile's quadrants are stored ahead of the MFMAs that overwrite them.  No explicit wait
    // for (CM, CN): hipcc counts the ds_reads itself (lgkmcnt(n) per MFMA).
#define Q8_DMA_SLOT(J, DA, DB)                                                                                           \
    do {                                                                                                                 \
        if (((J) & 3) == 0 && (DA) >= 0 && wr == ((J) >> 2)) { Q8_STAGE_PCS((DA) < 0 ? 0 : (DA), 1); Q8_SB(); }          \
        if (((J) & 3) == 1 && (DA) >= 0 && wr == ((J) >> 2)) { Q8_STAGE_PCS((DA) < 0 ? 0 : (DA), 2); Q8_SB(); }          \
        if (((J) & 3) == 2 && (DB) >= 0 && wr == ((J) >> 2)) { Q8_STAGE_PCS((DB) < 0 ? 0 : (DB), 1); Q8_SB(); }          \
        if (((J) & 3) == 3 && (DB) >= 0 && wr == ((J) >> 2)) { Q8_STAGE_PCS((DB) < 0 ? 0 : (DB), 2); Q8_SB(); }          \
    } while (0)
#define Q8_GROUP(CM, CN, NM, NN, NSK, NKS, DA, DB, EPI_HOOK, ZERO)                                                       \
    do {                                                                                                                 \
        if (!(DBG & 8)) __builtin_amdgcn_s_setprio(1);                                                                   \
        if (EPI_HOOK) { if (have_pend) { Q8_STORE_Q(pm0, pn0, pz, 0); } Q8_SB(); }                                       \
        Q8_MFMA1(CM, CN, 0, ZERO); Q8_SB(); Q8_RD1(NM, NN, NSK, NKS, 0); Q8_SB(); Q8_DMA_SLOT(0, DA, DB);                \
        Q8_MFMA1(CM, CN, 1, ZERO); Q8_SB(); Q8_RD1(NM, NN, NSK, NKS, 1); Q8_SB(); Q8_DMA_SLOT(1, DA, DB);                \
        if (EPI_HOOK) { if (have_pend) { Q8_STORE_Q(pm0, pn0, pz, 1); } Q8_SB(); }                                       \
        Q8_MFMA1(CM, CN, 2, ZERO); Q8_SB(); Q8_RD1(NM, NN, NSK, NKS, 2); Q8_SB(); Q8_DMA_SLOT(2, DA, DB);                \
        Q8_MFMA1(CM, CN, 3, ZERO); Q8_SB(); Q8_RD1(NM, NN, NSK, NKS, 3); Q8_SB(); Q8_DMA_SLOT(3, DA, DB);                \
        if (EPI_HOOK) { if (have_pend) { Q8_STORE_Q(pm0, pn0, pz, 2); } Q8_SB(); }                                       \
        Q8_MFMA1(CM, CN, 4, ZERO); Q8_SB(); Q8_RD1(NM, NN, NSK, NKS, 4); Q8_SB(); Q8_DMA_SLOT(4, DA, DB);                \
        Q8_MFMA1(CM, CN, 5, ZERO); Q8_SB(); Q8_RD1(NM, NN, NSK, NKS, 5); Q8_SB(); Q8_DMA_SLOT(5, DA, DB);                \
        if (EPI_HOOK) { if (have_pend) { Q8_STORE_Q(pm0, pn0, pz, 3); } Q8_SB(); }                                       \
        Q8_MFMA1(CM, CN, 6, ZERO); Q8_SB(); Q8_DMA_SLOT(6, DA, DB);                                                      \
        Q8_MFMA1(CM, CN, 7, ZERO); Q8_SB(); Q8_DMA_SLOT(7, DA, DB);                                                      \
        if (!(DBG & 8)) __builtin_amdgcn_s_setprio(0);                                                                   \
    } while (0)

    // ---- one K tile = four groups (k-steps of 16), fragments double-buffered X/Y one group ahead.
    // FIRST: first K tile of an output tile: the previous tile's quadrants are stored between the quadrants of group 0, whose
    //        MFMAs start from C = 0; its DMA was issued whole at the previous tile's last barrier, so groups 0-1 issue none.
    // LAST:  last K tile: at its barrier the whole next-but-one K tile is issued (before the epilogue's stores: counted waits).
    // Literal flags: three straight-line copies of the body (no accumulator is live across a branch that writes it).
#define Q8_KTILE(FIRST, LAST)                                                                                            \
    do {                                                                                                                 \
        const unsigned char* sK = lds + rslot * (4 * Q8_HALF);                                                           \
        const unsigned char* sN = lds + (rslot ^ 1) * (4 * Q8_HALF);                                                     \
        Q8_GROUP(xm, xn, ym, yn, sK, 1, (FIRST) ? -1 : 1, (FIRST) ? -1 : 2, FIRST, FIRST);                               \
        Q8_GROUP(ym, yn, xm, xn, sK, 2, (FIRST) ? -1 : 3, -1, false, false);                                             \
        Q8_GROUP(xm, xn, ym, yn, sK, 3, -1, -1, false, false);                                                           \
        /* group 3: all my reads of this K tile are done; my DMA of the next K tile has landed; barrier = published + slot free */ \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
        if ((FIRST) && have_pend) Q8_WAIT_DMA(4 * NST); else Q8_WAIT_DMA(0);                                             \
        if (!(DBG & 16)) __builtin_amdgcn_s_barrier();                                                                   \
        Q8_SB();                                                                                                         \
        if (LAST) { Q8_STAGE_PART(0); Q8_STAGE_PART(1); Q8_STAGE_PART(2); Q8_STAGE_PART(3); Q8_SB(); }                   \
        Q8_GROUP(ym, yn, xm, xn, sN, 0, (LAST) ? -1 : 0, -1, false, false);                                              \
        rslot ^= 1;                                                                                                      \
    } while (0)

    // prologue: K tiles 0 and 1 issued, K tile 0 landed and published, its first group requested
    Q8_STAGE_PART(0); Q8_STAGE_PART(1); Q8_STAGE_PART(2); Q8_STAGE_PART(3);
    const bool two_ = !pdone;
    Q8_STAGE_PART(0); Q8_STAGE_PART(1); Q8_STAGE_PART(2); Q8_STAGE_PART(3);
    if (two_) Q8_WAIT_DMA(8); else Q8_WAIT_DMA(0);
    __builtin_amdgcn_s_barrier();
    Q8_READ_GROUP(xm, xn, lds, 0);
    if ((DBG & 32) && A_KC && B_KC) {   // timing decomposition: real (random) fragments, read once and never again
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xm[i] = *reinterpret_cast<const hw_bf16x8*>(lds + offM[0] + i * 4096);
            ym[i] = *reinterpret_cast<const hw_bf16x8*>(lds + offM[1] + i * 4096);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            xn[i] = *reinterpret_cast<const hw_bf16x8*>(lds + offN[0] + i * 4096);
            yn[i] = *reinterpret_cast<const hw_bf16x8*>(lds + offN[1] + i * 4096);
        }
    }

    // ---- main loop over this workgroup's output tiles (every tile has at least two K tiles: the host guarantees K/split >= 128)
    bool have_pend = false;
    int pm0 = 0, pn0 = 0, pz = 0;
    for (int cv = (int)blockIdx.x; cv < total; cv += G) {
        const Q8Item cit = q8_decode(g, cv, total);
        const int cm0 = cit.m0, cn0 = cit.n0, cz = cit.z, cnt = cit.nt;
        Q8_KTILE(true, false);
#pragma unroll 1
        for (int t = 2; t < cnt; ++t) Q8_KTILE(false, false);
        Q8_KTILE(false, true);
        have_pend = true; pm0 = cm0; pn0 = cn0; pz = cz;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the dangling request of the group after the last
    if (have_pend) { Q8_STORE_Q(pm0, pn0, pz, 0); Q8_STORE_Q(pm0, pn0, pz, 1); Q8_STORE_Q(pm0, pn0, pz, 2); Q8_STORE_Q(pm0, pn0, pz, 3); }
#undef Q8_GROUP
#undef Q8_STORE_Q
#undef Q8_SB
#undef Q8_READ_GROUP
#undef Q8_MFMA1
#undef Q8_RD1
#undef Q8_STAGE_PART
#undef Q8_STAGE_PCS
#undef Q8_DMA_SLOT
#undef Q8_KTILE
#undef Q8_NEXT_ITEM
#undef Q8_WAIT_DMA
}
