// "Q8": persistent 256x256x64 bf16 GEMM on v_mfma_f32_32x32x16_bf16, eight waves (2 x 4) of 128x64, one workgroup per CU.
//
// Structure (round 2; replaces the K=32 DMA-ring kernel "P8" for every form it is built for):
//  * K tile = 64.  Each operand tile is two HALF-TILES of 128 rows (A_0/A_1: the rows of wave row 0/1; B_0/B_1: the columns of
//    wave columns 0-1 / 2-3), 16 KB each, filled by direct L2->LDS DMA (global_load_lds_dwordx4: 2 pieces per wave per half-tile).
//    Half-tiles live in two rings of NSLOT slots (NSLOT = 4: 128 KB, two K tiles; NSLOT = 5: all 160 KB, 2.5 K tiles).
//  * A K tile is FOUR PHASES per wave, one 64x32 quadrant of the wave's 128x64 block each (8 MFMAs of 32x32x16 = 256 matrix
//    cycles): quadrants (m0,n0) (m0,n1) (m1,n1) (m1,n0), so consecutive phases share one operand's fragments and a phase reads
//    4, 8 or 12 ds_read_b128.  A phase is {fragment reads for this phase; one half-tile of DMA for a later K tile; s_barrier;
//    8 MFMAs; s_barrier}.  Wave row 1 runs ONE BARRIER INTERVAL behind wave row 0: the two waves of a SIMD (w and w+4) are
//    always in opposite halves of a phase -- one multiplies while the other reads LDS and issues DMA -- which is what keeps
//    the matrix pipe fed without either wave having to overlap its own memory instructions with its own MFMAs.
//  * DMA runs LEAD = NSLOT+1 half-tiles ahead of the phase that issues it and is only ever waited for with a counted
//    s_waitcnt vmcnt(2*(LEAD-4)) once per K tile (phase 3), one full phase before the first read of that K tile.  The
//    half-tile stream is ONE flat sequence over all the output tiles a workgroup processes, so the next tile's first K tiles
//    are in LDS before the current tile's epilogue starts.  vmcnt retires in order on gfx9 (loads and stores share it), so
//    a counted wait is never early; stores in the queue only make it conservative.
//  * Epilogue in four pieces: quadrant q of an output tile is final after phase q of the tile's last K tile and is stored in
//    the read half of the NEXT phase (the last one in the first phase of the following tile), so the stores of a tile are
//    spread over four phases and no accumulator copy is needed.
//  * LDS images are DMA-linear (128-B rows, 8 rows per wave piece); the bank swizzle (16-B chunk ^ ((row >> 1) & 7)) is applied
//    to the lane's SOURCE address and again on the fragment reads (conflict-free for the 32-row b128 fragments, both row maps).
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
template <bool KC>
__device__ __forceinline__ void q8_stage_half(const unsigned char* base, int rec, int krem, unsigned char* dst,
                                              const unsigned (&voff)[2], int wave, int lane) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, rec < 0 ? 0 : rec, 0x00020000);
    unsigned v0 = voff[0], v1 = voff[1];
    if (KC && krem < 64) {   // last, partial K tile (uniform branch)
        const int kc0 = (lane & 7) ^ (((wave * 8 + (lane >> 3)) >> 1) & 7);   // same key for both pieces (64 rows apart)
        if (kc0 * 8 >= krem) { v0 = 0xFFFFFF00u; v1 = 0xFFFFFF00u; }
    }
    typedef void __attribute__((address_space(3))) lds_void;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + wave * 1024), 16, (int)v0, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + 8192 + wave * 1024), 16, (int)v1, 0, 0, 0);
}

// ---- epilogue of 8 consecutive outputs of one row (the host only selects this kernel when every [M, ld] epilogue operand is
// 16-B aligned at 8-column granularity and N % 8 == 0, so a group of 8 is either wholly inside the matrix or wholly outside).
// EPI is a compile-time selection of what the epilogue can do -- with every option tested at run time the epilogue's branches
// push the kernel over its 256 registers:
//   0  bf16 C = alpha*acc (+bias)            1  ... + save pre-activation + exact GELU        2  ... + residual
//   3  bf16 C = alpha*acc * gelu'(gmul) (+residual)                                            4  f32: split-K slab, or C (+= old)
template <int EPI>
__device__ __forceinline__ void q8_epi8(const GemmArgs& g, int m, int n, float (&v)[8], int z, const float (&bias)[8],
                                        float al, const uint4& qg, const uint4& qr) {
    if (m >= g.M || n >= g.N) return;
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] *= al;
    if (EPI <= 2 && g.bias) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += bias[r];
    }
    if (EPI == 1) {
        st8<bf16_t>(reinterpret_cast<bf16_t*>(g.pre_out) + (long)m * g.ldp + n, v);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = gelu_t<bf16_t>(rnd<bf16_t>(v[r]));
    }
    if (EPI == 3) {
        const uint32_t w[4] = {qg.x, qg.y, qg.z, qg.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[2 * r] *= gelu_grad_t<bf16_t>(__uint_as_float(w[r] << 16));
            v[2 * r + 1] *= gelu_grad_t<bf16_t>(__uint_as_float(w[r] & 0xffff0000u));
        }
    }
    if (EPI == 2 || (EPI == 3 && g.residual)) {
        const uint32_t w[4] = {qr.x, qr.y, qr.z, qr.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[2 * r] += __uint_as_float(w[r] << 16);
            v[2 * r + 1] += __uint_as_float(w[r] & 0xffff0000u);
        }
    }
    if (EPI == 4) {
        if (g.partial) {
            st8<float>(g.partial + ((long)z * g.M + m) * g.N + n, v);
        } else {
            float* c = reinterpret_cast<float*>(g.C) + (long)m * g.ldc + n;
            if (g.accumulate) {
                float o[8];
                ld8<float>(c, o);
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] += o[r];
            }
            st8<float>(c, v);
        }
    } else {
        st8<bf16_t>(reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n, v);
    }
}

// DBG bits (development, GemmArgs.dbg): 1 = no MFMA, 2 = no DMA, 4 = no epilogue, 16 = epilogue in one piece after the tile
template <bool A_KC, bool B_KC, int EPI, int NSLOT, int DBG = 0>
__global__ __launch_bounds__(512) void gemm_bf16_q8_kernel(GemmArgs g) {
    static_assert(NSLOT == 4 || NSLOT == 5, "ring of 4 or 5 half-tile slots per operand");
    constexpr int LEAD = NSLOT + 1;                 // half-tiles the DMA runs ahead of the phase that issues it
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // A ring | B ring; the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int total = g.nbm * g.nbn * g.nsplit, G = (int)gridDim.x;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);
    const int l31 = lane & 31, lh = lane >> 5;
    // N-side fragment row -> tile column (see header)
    const int ncol = (l31 & 3) | (((l31 >> 3) & 1) << 2) | (((l31 >> 2) & 1) << 3) | ((l31 >> 4) << 4);

    // per-lane fragment offsets inside a half-tile (without slot base and quadrant offset)
    //   kc: row*128 + ((2*ks + lh) ^ key(row)) * 16, one per k-step (the XOR does not commute with the k-step offset)
    //   oc: (ks*16 + 8*kb + r)*256 + ((chunk ^ (r<<2)) * 16) + within, one per 32-output tile index (the XOR touches the tile bits)
    unsigned offM[4], offN[4];
    if (A_KC) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) offM[ks] = (unsigned)(l31 * 128 + (((2 * ks + lh) ^ ((l31 >> 1) & 7)) << 4));
    } else {
        const int i16 = lane & 15, ob = (lane >> 4) & 1, kb = lane >> 5, r = i16 >> 2, q = i16 & 3;
#pragma unroll
        for (int t = 0; t < 4; ++t) {   // tile index t = 2*mh + tm of the wave's 128 rows
            const int col = t * 32 + 16 * ob + 4 * q;
            offM[t] = (unsigned)((8 * kb + r) * 256 + ((((col >> 3) ^ (r << 2)) & 15) << 4) + (col & 7) * 2);
        }
    }
    if (B_KC) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) offN[ks] = (unsigned)(((wc & 1) * 64 + ncol) * 128 + (((2 * ks + lh) ^ ((ncol >> 1) & 7)) << 4));
    } else {
        const int i16 = lane & 15, ob = (lane >> 4) & 1, kb = lane >> 5, r = i16 >> 2, q = i16 & 3;
#pragma unroll
        for (int t = 0; t < 2; ++t) {   // t = nh; the pointer of quarter q covers the 4 outputs at 4*(q>>1) + 8*(q&1) (column remap)
            const int col = (wc & 1) * 64 + t * 32 + 16 * ob + 4 * (q >> 1) + 8 * (q & 1);
            offN[t] = (unsigned)((8 * kb + r) * 256 + ((((col >> 3) ^ (r << 2)) & 15) << 4) + (col & 7) * 2);
        }
        offN[2] = offN[3] = 0;
    }

    f32x16 acc[4][2];

    // ---- DMA cursor over the flat half-tile stream ---------------------------------------------------------------------------
    // all of it wave-uniform (SGPRs): byte cursors of the A and B half-tile 0 of the K tile being staged, bytes left in their valid
    // ranges, contraction elements left in the staged output tile
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
    int wA = 0, wB = 0;
    unsigned voffA[2], voffB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { voffA[i] = q8_voff<A_KC>(i, wave, lane, g.lda); voffB[i] = q8_voff<B_KC>(i, wave, lane, g.ldb); }
    int nstaged = 0;                                   // half-tiles issued so far (for the tail waits)
#define Q8_STAGE(KIND)                                                                                                   \
    do {                                                                                                                 \
        if (!pdone) {                                                                                                    \
            if (!(DBG & 2)) {                                                                                            \
                if ((KIND) == 0) q8_stage_half<A_KC>(sa_base, sa_rec, p_krem, lds + wA * Q8_HALF, voffA, wave, lane);    \
                if ((KIND) == 1) q8_stage_half<A_KC>(sa_base + a_half, sa_rec - a_half, p_krem, lds + wA * Q8_HALF, voffA, wave, lane);           \
                if ((KIND) == 2) q8_stage_half<B_KC>(sb_base, sb_rec, p_krem, lds + (NSLOT + wB) * Q8_HALF, voffB, wave, lane);                   \
                if ((KIND) == 3) q8_stage_half<B_KC>(sb_base + b_half, sb_rec - b_half, p_krem, lds + (NSLOT + wB) * Q8_HALF, voffB, wave, lane); \
            }                                                                                                            \
            if ((KIND) < 2) wA = wA == NSLOT - 1 ? 0 : wA + 1; else wB = wB == NSLOT - 1 ? 0 : wB + 1;                   \
            ++nstaged;                                                                                                   \
            if ((KIND) == 3) {                                                                                           \
                p_krem -= 64;                                                                                            \
                sa_base += a_step; sa_rec -= a_step; sb_base += b_step; sb_rec -= b_step;                                \
                if (p_krem <= 0) {                                                                                       \
                    pv += G;                                                                                             \
                    if (pv < total) Q8_NEXT_ITEM(); else pdone = true;                                                   \
                }                                                                                                        \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)

    // wait until at most `n_` of my half-tiles are in flight (n_ in 0..2)
#define Q8_WAIT_HALVES(N_)                                                         \
    do {                                                                           \
        const int n_ = (N_);                                                       \
        if (n_ >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");              \
        else if (n_ == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");         \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      \
    } while (0)

    // prologue: LEAD half-tiles in flight, K tile 0 landed and published
    Q8_STAGE(0); Q8_STAGE(1); Q8_STAGE(2); Q8_STAGE(3); Q8_STAGE(0);
    if (LEAD == 6) Q8_STAGE(1);
    Q8_WAIT_HALVES(nstaged - 4);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // wave row 1 runs one barrier interval behind wave row 0

    hw_bf16x8 fm[2][4], fn[4];
    int rA = 0, rB = 0;                               // ring slots of A_0 / B_0 of the K tile being multiplied
    int consumed = 0;                                 // K tiles multiplied so far (x4 = half-tiles retired)

    // fragment reads ------------------------------------------------------------------------------------------------------------
#define Q8_READ_FM(MH)                                                                                                   \
    do {                                                                                                                 \
        if (A_KC) {                                                                                                      \
            _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                                                             \
                _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                         \
                    fm[tm][ks] = *reinterpret_cast<const hw_bf16x8*>(sA + offM[ks] + ((MH) * 64 + tm * 32) * 128);       \
        } else {                                                                                                         \
            _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                                                             \
                _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                       \
                    const unsigned char* p_ = sA + offM[2 * (MH) + tm] + ks * 16 * 256;                                  \
                    q8_v4s16 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((q8_v4s16 __attribute__((address_space(3)))*)(p_));            \
                    q8_v4s16 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((q8_v4s16 __attribute__((address_space(3)))*)(p_ + 4 * 256));  \
                    fm[tm][ks] = __builtin_bit_cast(hw_bf16x8, (bf16x8){lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]}); \
                }                                                                                                        \
        }                                                                                                                \
    } while (0)
#define Q8_READ_FN(NH)                                                                                                   \
    do {                                                                                                                 \
        if (B_KC) {                                                                                                      \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                             \
                fn[ks] = *reinterpret_cast<const hw_bf16x8*>(sB + offN[ks] + (NH) * 32 * 128);                           \
        } else {                                                                                                         \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                           \
                const unsigned char* p_ = sB + offN[(NH)] + ks * 16 * 256;                                               \
                q8_v4s16 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((q8_v4s16 __attribute__((address_space(3)))*)(p_));                \
                q8_v4s16 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((q8_v4s16 __attribute__((address_space(3)))*)(p_ + 4 * 256));      \
                fn[ks] = __builtin_bit_cast(hw_bf16x8, (bf16x8){lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]});         \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)
#define Q8_MFMA(MH, NH, FIRST)                                                                                           \
    do {                                                                                                                 \
        if (DBG & 1) {   /* keep the fragment reads alive */                                                           \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                           \
                asm volatile("" ::"v"(fn[ks]), "v"(fm[0][ks]), "v"(fm[1][ks]));                                          \
                if (FIRST) { acc[2 * (MH)][(NH)] = zero16; acc[2 * (MH) + 1][(NH)] = zero16; }                           \
            }                                                                                                            \
        } else {                                                                                                         \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                             \
                _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                                                         \
                    acc[2 * (MH) + tm][(NH)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                  \
                        fn[ks], fm[tm][ks], ((FIRST) && ks == 0) ? zero16 : acc[2 * (MH) + tm][(NH)], 0, 0, 0);          \
        }                                                                                                                \
    } while (0)

    // epilogue of quadrant (MH, NH) of the output tile at (m0, n0), split z: 2 tiles x 2 groups of 8 consecutive columns per lane.
    // The accumulators are only READ: the first K tile of the next output tile starts from C = 0 in the MFMA itself.
    auto store_quadrant = [&](int tm0, int tn0, int tz, auto mh_c, auto nh_c) {
        constexpr int MH = decltype(mh_c)::value, NH = decltype(nh_c)::value;
        if (DBG & 4) {   // keep the accumulators (and so the MFMAs) alive without storing them
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) asm volatile("" ::"v"(acc[2 * MH + tm][NH]));
            return;
        }
        const int nb = tn0 + wc * 64 + NH * 32;              // wave-uniform: first column of the quadrant
        const int n_l = nb + 8 * lh;                         // + 16*gq below
        const int m_l = tm0 + wr * 128 + MH * 64 + l31;      // + 32*tm below
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
        const bf16_t* gm = reinterpret_cast<const bf16_t*>(g.gmul);
        const bf16_t* rs = reinterpret_cast<const bf16_t*>(g.residual);
        uint4 qg[2][2], qr[2][2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                qg[tm][gq] = make_uint4(0, 0, 0, 0);
                qr[tm][gq] = make_uint4(0, 0, 0, 0);
                const int m = min(m_l + tm * 32, g.M - 1), n = min(n_l + 16 * gq, g.N - 8);
                if (EPI == 3) qg[tm][gq] = *reinterpret_cast<const uint4*>(gm + (long)m * g.ldg + n);
                if (EPI == 2 || (EPI == 3 && rs)) qr[tm][gq] = *reinterpret_cast<const uint4*>(rs + (long)m * g.ldr + n);
            }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = acc[2 * MH + tm][NH][8 * gq + r];
                q8_epi8<EPI>(g, m_l + tm * 32, n_l + 16 * gq, v, tz, bias[gq], al, qg[tm][gq], qr[tm][gq]);
            }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ---- one K tile = four phases.  FIRST: first K tile of an output tile (C = 0; the previous tile's last quadrant is stored in
    // phase 0).  LAST: last K tile (quadrants stored as they become final).  Literal flags: three straight-line copies of the
    // body, so that no accumulator is live across a branch that writes it (which costs a second register copy of all of them).
#define Q8_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define Q8_MULT(MH, NH, FIRST)                   \
    do {                                         \
        Q8_BAR();                                \
        __builtin_amdgcn_s_setprio(1);           \
        Q8_MFMA(MH, NH, FIRST);                  \
        __builtin_amdgcn_s_setprio(0);           \
        Q8_BAR();                                \
    } while (0)
#define Q8_KTILE(FIRST, LAST)                                                                                            \
    do {                                                                                                                 \
        const unsigned char* sA = lds + (rA + wr >= NSLOT ? rA + wr - NSLOT : rA + wr) * Q8_HALF;                         \
        const int sb_ = rB + (wc >> 1);                                                                                  \
        const unsigned char* sB = lds + (NSLOT + (sb_ >= NSLOT ? sb_ - NSLOT : sb_)) * Q8_HALF;                          \
        /* phase 0: quadrant (m0, n0) */                                                                                 \
        Q8_READ_FN(0);                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        Q8_READ_FM(0);                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        Q8_STAGE((0 + LEAD) & 3);                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if ((FIRST) && have_pend) store_quadrant(pm0, pn0, pz, I1(), I0());                                              \
        Q8_MULT(0, 0, FIRST);                                                                                            \
        /* phase 1: quadrant (m0, n1) */                                                                                 \
        Q8_READ_FN(1);                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        Q8_STAGE((1 + LEAD) & 3);                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if (LAST) store_quadrant(cm0, cn0, cz, I0(), I0());                                                              \
        Q8_MULT(0, 1, FIRST);                                                                                            \
        /* phase 2: quadrant (m1, n1) */                                                                                 \
        Q8_READ_FM(1);                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        Q8_STAGE((2 + LEAD) & 3);                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if (LAST) store_quadrant(cm0, cn0, cz, I0(), I1());                                                              \
        Q8_MULT(1, 1, FIRST);                                                                                            \
        /* phase 3: quadrant (m1, n0); the wait that publishes the NEXT K tile (read from the next phase on) */          \
        Q8_READ_FN(0);                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        Q8_STAGE((3 + LEAD) & 3);                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        Q8_WAIT_HALVES(nstaged - 4 * (consumed + 2));                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if (LAST) store_quadrant(cm0, cn0, cz, I1(), I1());                                                              \
        Q8_MULT(1, 0, FIRST);                                                                                            \
        rA = rA + 2 >= NSLOT ? rA + 2 - NSLOT : rA + 2;                                                                  \
        rB = rB + 2 >= NSLOT ? rB + 2 - NSLOT : rB + 2;                                                                  \
        ++consumed;                                                                                                      \
    } while (0)

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
    if (have_pend) store_quadrant(pm0, pn0, pz, I1(), I0());
    if (wr == 0) __builtin_amdgcn_s_barrier();        // pairs with wave row 1's extra barrier
#undef Q8_KTILE
#undef Q8_MULT
#undef Q8_BAR
#undef Q8_STAGE
#undef Q8_NEXT_ITEM
#undef Q8_WAIT_HALVES
#undef Q8_READ_FM
#undef Q8_READ_FN
#undef Q8_MFMA
}
