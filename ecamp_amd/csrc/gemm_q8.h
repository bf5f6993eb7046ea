// "Q8": persistent 256x256x64 bf16 GEMM on v_mfma_f32_32x32x16_bf16, eight waves (2 x 4) of 128x64, one workgroup per CU.
//
// What the measurements on MI355X say (tools/gemm_lab, profiles/r02_gemm_lab_*.txt, profiles/r03_gemm_lab_*.txt) and how the kernel
// answers them:
//  * With random operands the matrix pipes are power-limited: a pure MFMA stream (no LDS, no DMA) runs at ~1.6 PFLOP/s.  A 256^2 tile
//    needs 64 KB of operands per K = 64 tile through the CU's vector-memory path, and its 128 KB of bf16 output leave through the
//    same path at ~16 B/clk: the loop below runs at ~90 % of what that path delivers under MFMA load, and the epilogue's stores ADD
//    their time to it however they are arranged (round 3: a tile packed to bf16 and stored one instruction per phase over the next
//    four K tiles takes exactly as long as the burst) -- so the epilogue is the plain burst.
//  * Operand tiles are HALF-TILES of 128 rows x 64 k (16 KB: A_0/A_1 = rows of wave row 0/1, B_0/B_1 = columns of wave columns
//    0-1 / 2-3) in two rings of 5 slots (all 160 KB of LDS = 2.5 K tiles).  They are filled by buffer_load_dwordx4 ... lds through
//    per-half-tile descriptors built with scalar ALU only (per-lane offsets are kernel constants; rows / contraction steps past
//    the end read as zero), one half-tile (2 instructions per wave) per phase, as one flat stream over the workgroup's tiles, and
//    waited for with a counted s_waitcnt once per K tile.
//  * PHASE SCHEDULE (round 3).  A K tile is four PHASES (k-steps of 16).  In a phase a wave first LOADS -- the six fragment reads of
//    this very phase (4 M-side + 2 N-side, one register set), its two DMA instructions, the waits -- then, after a barrier, MULTIPLIES:
//    its 8 independent MFMAs (4 x 2 tiles of 32x32) back to back at raised priority, then a second barrier.  Wave row 1 runs ONE
//    BARRIER BEHIND wave row 0, and the two waves of a SIMD belong to different rows: while one multiplies the other loads.  The matrix
//    pipe sees one uncontended MFMA stream at a time, and the LDS / DMA issue stalls of the loading wave cost no MFMA slot (a wave
//    blocked issuing an LDS-DMA piece cannot issue MFMAs; interleaving everything in every wave -- the round-2 schedule: fragments a
//    group ahead in a register double buffer, one barrier per K tile -- left both waves of a SIMD stalled at the same moments:
//    forward / data-gradient / weight-gradient forms 7-15 % slower on every shape, tools/gemm_lab --ph=0 before its removal).
//    The single fragment set also frees 24 registers (180 instead of 215 in the plain forward form).
//  * Epilogue: the previous output tile is stored at the head of the next tile's first phase, whose MFMAs start from C = 0 (no
//    accumulator copy).  Buffer stores (bounds by descriptor, no exec-masked branches); they are older than every DMA the tile's
//    first wait covers, and vmcnt retires in order.
//  * LDS images are DMA-linear (128-B rows, 8 rows per wave piece); the bank swizzle (16-B chunk ^ ((row >> 1) & 7)) is applied
//    to the lane's SOURCE offset and again on the fragment reads (conflict-free for the 32-row b128 fragments, both row maps:
//    SQ_LDS_BANK_CONFLICT = 0).  Strided operands stay as they lie in HBM (64 k-rows x 256 B) and are read with
//    ds_read_b64_tr_b16 -- through inline asm: hipcc puts s_waitcnt vmcnt(0) in front of the builtin whenever an LDS-DMA is in flight.
//  * N-side fragment row i is mapped to tile column (i&3) | i3<<2 | i2<<3 | i4<<4, so that with the N fragment as the MFMA's
//    A operand a lane ends up with 8 consecutive output columns per 8 accumulator registers: every epilogue access is 16 B.
#pragma once
#include "gemm_args.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef hw_h16x8 hw_bf16x8;   // MFMA operand: eight elements of the build's 16-bit format (common.h)
typedef __attribute__((ext_vector_type(4))) short q8_v4s16;
typedef unsigned int q8_u32x4_t __attribute__((ext_vector_type(4)));

#define Q8_HALF 16384
#define Q8_GLDS16(SRC, DST) \
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(SRC), (void __attribute__((address_space(3)))*)(DST), 16, 0, 0)

static __device__ __attribute__((aligned(16))) unsigned int q8_zero16[4] = {0u, 0u, 0u, 0u};

// one output tile (and split-K slice) of the persistent kernel; every field is wave-uniform
struct Q8Item {
    int m0, n0, kbeg, kend, nt, z, ncol;
};
template <int KT_SHIFT = 6>   // log2 of the K tile's depth in elements: 64 bf16 or 128 e4m3 (a 128-byte row either way)
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
    it.nt = (it.kend - it.kbeg + (1 << KT_SHIFT) - 1) >> KT_SHIFT;
    // the divisions above run on the VALU (v_rcp) and come back through v_readfirstlane; naming every result a scalar HERE keeps hipcc
    // from moving the whole chain behind them (pointers, record counts: the buffer descriptors) into VGPRs, which costs a waterfall
    // loop around every DMA instruction (cdna_hip_programming.md T20)
    it.m0 = __builtin_amdgcn_readfirstlane(it.m0); it.n0 = __builtin_amdgcn_readfirstlane(it.n0); it.z = __builtin_amdgcn_readfirstlane(it.z);
    it.ncol = __builtin_amdgcn_readfirstlane(it.ncol); it.kbeg = __builtin_amdgcn_readfirstlane(it.kbeg);
    it.kend = __builtin_amdgcn_readfirstlane(it.kend); it.nt = __builtin_amdgcn_readfirstlane(it.nt);
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
template <bool KC, int ES = 2>   // ES: bytes per element (2 bf16, 1 e4m3)
__device__ __forceinline__ unsigned q8_voff(int i, int wave, int lane, long ld) {
    const int pi = i * 8 + wave;
    if (KC) {
        const int row = pi * 8 + (lane >> 3);
        const int kc = (lane & 7) ^ ((row >> 1) & 7);
        return (unsigned)((long)row * ld * ES + kc * 16);
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

// ---- fragment registers ----------------------------------------------------------------------------------------------------------
// contraction-contiguous operand: one ds_read_b128 that hipcc tracks itself.  Strided operand: two ds_read_b64_tr_b16 in inline
// asm (see header); the halves stay separate values until `wait()` has named them after an explicit s_waitcnt lgkmcnt(0).
template <bool KC> struct Q8Frag;
template <> struct Q8Frag<true> {
    hw_bf16x8 v;
    template <int OFF> __device__ __forceinline__ void read(const unsigned char* p) { v = *reinterpret_cast<const hw_bf16x8*>(p + OFF); }
    __device__ __forceinline__ hw_bf16x8 get() const { return v; }
};
template <> struct Q8Frag<false> {
    q8_v4s16 lo, hi;
    template <int OFF> __device__ __forceinline__ void read(const unsigned char* p) {
        const unsigned a = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(OFF));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(OFF + 4 * 256));
    }
    __device__ __forceinline__ hw_bf16x8 get() const {
        return __builtin_bit_cast(hw_bf16x8, (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
    }
};
// e4m3 operand (contraction-contiguous only): the 32 bytes a lane feeds to one v_mfma_scale_f32_32x32x64_f8f6f4 = two 16-B chunks of its row
typedef __attribute__((ext_vector_type(8))) int q8_v8i32;
struct Q8Frag8 {
    q8_u32x4_t lo, hi;
    template <int OFF> __device__ __forceinline__ void read(const unsigned char* plo, const unsigned char* phi) {
        lo = *reinterpret_cast<const q8_u32x4_t*>(plo + OFF);
        hi = *reinterpret_cast<const q8_u32x4_t*>(phi + OFF);
    }
    __device__ __forceinline__ q8_v8i32 get() const { return (q8_v8i32){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]}; }
};
// the explicit wait of a group whose fragments include asm reads: names every half so no consumer is scheduled above it
__device__ __forceinline__ void q8_wait4(Q8Frag<false> (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo), "+v"(f[3].hi));
}
__device__ __forceinline__ void q8_wait2(Q8Frag<false> (&f)[2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi));
}
__device__ __forceinline__ void q8_wait4(Q8Frag<true> (&)[4]) {}
__device__ __forceinline__ void q8_wait2(Q8Frag<true> (&)[2]) {}

// ---- epilogue of 8 consecutive outputs of one row (the host only selects this kernel when every [M, ld] epilogue operand is
// 16-B aligned at 8-column granularity, N % 8 == 0 and every matrix is < 2 GB).  EPI is a compile-time selection of what the
// epilogue can do -- with every option tested at run time the epilogue's branches push the kernel over its 256 registers:
//   0  bf16 C = alpha*acc (+bias)            1  ... + save pre-activation + exact GELU        2  ... + residual
//   3  bf16 C = alpha*acc * gelu'(gmul) (+residual)                                            4  f32: split-K slab, or C (+= old)
typedef unsigned int q8_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ q8_u32x4 q8_pack8(const float (&v)[8]) {
    return (q8_u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
}

// DBG bits (development, template parameter): 1 = no MFMA, 2 = no DMA, 4 = no epilogue, 1024 = the epilogue runs but every store / load is out of range (dropped), 8 = no s_setprio, 16 = no barriers in the
// loop, 32 = no fragment reads, 64 = every DMA re-reads the tile's first K tile (timing decomposition only: 16, 32, 64 give wrong results),
// 128 / 256 = nt / sc1 output stores, 512 = workgroups start in four phases 9 us apart (lock-step epilogue bursts: the stagger costs what it saves)
// ROWSUM (weight-gradient form only): rowsum[m] += alpha * sum_k opA[m,k] -- the bias gradient -- summed on the VALU from the M-side
// fragments by the first wave column of the tiles in the first N-tile column, added with 4 buffer atomics per wave and tile.
// ITEMS (weight-gradient form only): the work items come from a table (Q8Group, gemm_args.h) instead of the tile x split arithmetic;
// operands and leading dimensions change with the item's problem, every piece stores a dense f32 slab.
// SCH: 0 = the round-3 stream (descriptors rebuilt per half-tile, bookkeeping spread over the load interval; the item-table form),
// 1 = lean stream (every other form).  Measured and removed again (profiles/r04_gemm_lab_sch.txt): the lean stream's bookkeeping in the
// matrix interval (-1..3 %), the DMA issue in the matrix interval as well (= sch 0), two k-steps per phase = four barriers per K tile
// instead of eight (+-1 %), and wave-specialised producers -- 8 consumer + 4 producer waves, the whole operand stream in waves of its own
// (tools/probes/gemm_producer_waves_fragment.hip.txt) -- within 3 % of this kernel on every shape.  Every schedule lands on the same
// ~1.75 us per 256 x 256 x 64 step: the loop waits for the DATA (the L2 -> LDS path beside a power-limited MFMA stream), not for whoever
// issues the loads, for bookkeeping, or for barriers.
// F8: both operands are OCP e4m3 (one byte per element, contraction-contiguous; BASELINE.json configs[4]).  A K tile is still a
// 128-byte row per operand row -- 128 elements instead of 64 -- so the LDS images, the rings, the DMA pieces and the counted waits
// are the bf16 kernel's byte for byte; a phase is four v_mfma_scale_f32_32x32x64_f8f6f4 (64 matrix-pipe cycles each, block scales
// 2^0: twice the bf16 rate per byte moved) on one k-step of 64 and one half of the wave's rows, the N-side fragments staying in
// registers between the two halves.  The per-tensor scales are device scalars multiplied into alpha (alpha_dev, alpha_dev2).
template <bool A_KC, bool B_KC, int EPI, int DBG, bool ROWSUM, bool ITEMS, int SCH = 0, bool F8 = false>
__device__ __forceinline__ void q8_body(const GemmArgs& g, const Q8Group& GR) {
    static_assert(!F8 || (A_KC && B_KC && SCH == 1 && !ROWSUM && !ITEMS && DBG == 0), "the e4m3 form is the forward form on the lean stream");
    constexpr int ES = F8 ? 1 : 2;            // bytes per operand element
    constexpr int KT = F8 ? 128 : 64;         // elements per K tile
    constexpr int KT_SHIFT = F8 ? 7 : 6;
    static_assert(!ITEMS || (!A_KC && !B_KC && EPI == 4 && DBG == 0), "the item-table form is the weight-gradient form");
    static_assert(!ROWSUM || (!A_KC && !B_KC && EPI == 4), "rowsum is built for the weight-gradient form");
    constexpr int NSLOT = 5;                          // half-tile slots per operand ring
    constexpr int ST_AUX = (DBG & 128) ? 2 : (DBG & 256) ? 16 : 0;   // cache policy of the bf16 output stores: 2 = nt, 16 = sc1 (write-through)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // A ring (5 x 16 KB) | B ring (5 x 16 KB); the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // ITEMS: this workgroup's consecutive pieces [it_beg, total) of the table; otherwise items blockIdx.x, + G, ... of the arithmetic order
    typedef const int __attribute__((address_space(4))) cint4;
    const int it_beg = ITEMS ? *(cint4*)(GR.wg_first + blockIdx.x) : (int)blockIdx.x;
    const int total = ITEMS ? *(cint4*)(GR.wg_first + blockIdx.x + 1) : g.nbm * g.nbn * g.nsplit, G = ITEMS ? 1 : (int)gridDim.x;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);
    // field F of problem P_ of the group (scalar selects: a run-time index into the by-value argument would put it in scratch)
    // (selects of VALUES read once here: a select between the fields' addresses keeps the whole argument block in scratch)
#define Q8_PROB_FIELD(F) const auto gp0_##F = GR.p[0].F, gp1_##F = GR.p[1].F, gp2_##F = GR.p[2].F, gp3_##F = GR.p[3].F
    Q8_PROB_FIELD(A); Q8_PROB_FIELD(B); Q8_PROB_FIELD(lda); Q8_PROB_FIELD(ldb); Q8_PROB_FIELD(M); Q8_PROB_FIELD(rowsum);
    Q8_PROB_FIELD(alpha_out); Q8_PROB_FIELD(alpha_dev_out);
#undef Q8_PROB_FIELD
    float* const gr_slabs = GR.slabs;
    // (prvalues on purpose: inside a by-reference lambda a ?: over the NAMES is a select between their addresses, and the four values
    // end up as a table in scratch that every item set-up loads from)
#define Q8_VAL_(X) static_cast<std::decay_t<decltype(X)>>(X)
#define Q8_PROB(P_, F) ((P_) == 0 ? Q8_VAL_(gp0_##F) : (P_) == 1 ? Q8_VAL_(gp1_##F) : (P_) == 2 ? Q8_VAL_(gp2_##F) : Q8_VAL_(gp3_##F))
    auto item_at = [&](int i) {   // record i of the table, through scalar loads (constant address space, wave-uniform index)
        typedef int q8_i32x8 __attribute__((ext_vector_type(8)));
        typedef const q8_i32x8 __attribute__((address_space(4))) crec4;
        const q8_i32x8 w = *(crec4*)(reinterpret_cast<const int*>(GR.items) + (long)i * 8);
        Q8ItemRec r;
#define Q8_U(X) __builtin_amdgcn_readfirstlane(X)
        r.prob = Q8_U(w[0]); r.m0 = Q8_U(w[1]); r.n0 = Q8_U(w[2]); r.kbeg = Q8_U(w[3]); r.kend = Q8_U(w[4]); r.slab = Q8_U(w[5]); r.flags = Q8_U(w[6]); r.pad = 0;
#undef Q8_U
        return r;
    };
    auto uniform_ptr = [](const void* q) {   // tells the compiler that a selected pointer is wave-uniform (it feeds scalar loads and descriptors)
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        return (const void*)(((unsigned long long)hi << 32) | lo);
    };
    const int l31 = lane & 31, lh = lane >> 5;
    // N-side fragment row -> tile column (see header)
    const int ncol = (l31 & 3) | (((l31 >> 3) & 1) << 2) | (((l31 >> 2) & 1) << 3) | ((l31 >> 4) << 4);

    // per-lane fragment offsets inside a half-tile
    //   kc: row*128 + ((2*ks + lh) ^ key(row)) * 16, one per k-step (the XOR does not commute with the k-step offset); tile t at + t*4096
    //   oc: (8*kb + r)*256 + ((chunk ^ (r<<2)) * 16) + within, one per 32-output tile (the XOR touches the tile bits); k-step ks at + ks*4096
    unsigned offM[4], offN[4];
    if (A_KC) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) offM[ks] = (unsigned)(l31 * 128 + (((2 * ks + lh) ^ ((l31 >> 1) & 7)) << 4));
    } else {
        const int i16 = lane & 15, ob = (lane >> 4) & 1, kb = lane >> 5, r = i16 >> 2, q = i16 & 3;
#pragma unroll
        for (int t = 0; t < 4; ++t) {   // 32-row tile t of the wave's 128 rows
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
        for (int t = 0; t < 2; ++t) {   // the pointer of quarter q covers the 4 outputs at 4*(q>>1) + 8*(q&1) (column remap)
            const int col = (wc & 1) * 64 + t * 32 + 16 * ob + 4 * (q >> 1) + 8 * (q & 1);
            offN[t] = (unsigned)((8 * kb + r) * 256 + ((((col >> 3) ^ (r << 2)) & 15) << 4) + (col & 7) * 2);
        }
        offN[2] = offN[3] = 0;
    }

    if (F8) {   // chunk pair (4 * ks8 + 2 * lh, + 1) of k-step ks8: offM / offN [2 * ks8 + j]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = 4 * (q >> 1) + 2 * lh + (q & 1);
            offM[q] = (unsigned)(l31 * 128 + ((ch ^ ((l31 >> 1) & 7)) << 4));
            offN[q] = (unsigned)(((wc & 1) * 64 + ncol) * 128 + ((ch ^ ((ncol >> 1) & 7)) << 4));
        }
    }
    f32x16 acc[4][2];

    // ---- DMA cursor over the flat K-tile stream; all of it wave-uniform (SGPRs): byte cursors of the A and B half-tile 0 of the
    // K tile being staged, bytes left in their valid ranges, contraction elements left in the staged output tile.  The four parts
    // of a K tile are staged in the order A_0, B_0 ("early": their ring slots were vacated two K tiles ago), A_1, B_1 ("late").
    int pv = it_beg;
    bool pdone = pv >= total;
    const unsigned char *sa_base, *sb_base;
    int sa_rec, sb_rec, p_krem;
    const int a_half = A_KC ? (int)g.lda * 256 : 256;     // bytes from half-tile 0 to half-tile 1 (128 rows)
    int a_step = A_KC ? 128 : (int)g.lda * 128;   // (ITEMS: follows the staged item's problem)
    const int b_half = B_KC ? (int)g.ldb * 256 : 256;
    int b_step = B_KC ? 128 : (int)g.ldb * 128;
    long s_lda = ITEMS ? -1 : g.lda, s_ldb = ITEMS ? -1 : g.ldb;   // leading dimensions the per-lane DMA offsets were built for
    unsigned voffA[2], voffB[2];
    if (!ITEMS) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { voffA[i] = q8_voff<A_KC>(i, wave, lane, g.lda); voffB[i] = q8_voff<B_KC>(i, wave, lane, g.ldb); }
    }
#define Q8_NEXT_ITEM()                                                                                                   \
    do {                                                                                                                 \
        if (ITEMS) {                                                                                                     \
            const Q8ItemRec r_ = item_at(pv);                                                                            \
            const long la_ = Q8_PROB(r_.prob, lda), lb_ = Q8_PROB(r_.prob, ldb);                                         \
            const bf16_t* A_ = reinterpret_cast<const bf16_t*>(uniform_ptr(Q8_PROB(r_.prob, A)));                        \
            const bf16_t* B_ = reinterpret_cast<const bf16_t*>(uniform_ptr(Q8_PROB(r_.prob, B)));                        \
            p_krem = r_.kend - r_.kbeg;                                                                                  \
            sa_base = (const unsigned char*)(A_ + ((long)r_.kbeg * la_ + r_.m0)); sa_rec = (int)(((long)p_krem * la_ - r_.m0) * 2); \
            sb_base = (const unsigned char*)(B_ + ((long)r_.kbeg * lb_ + r_.n0)); sb_rec = (int)(((long)p_krem * lb_ - r_.n0) * 2); \
            if (la_ != s_lda) { s_lda = la_; a_step = (int)la_ * 128; voffA[0] = q8_voff<A_KC>(0, wave, lane, la_); voffA[1] = q8_voff<A_KC>(1, wave, lane, la_); } \
            if (lb_ != s_ldb) { s_ldb = lb_; b_step = (int)lb_ * 128; voffB[0] = q8_voff<B_KC>(0, wave, lane, lb_); voffB[1] = q8_voff<B_KC>(1, wave, lane, lb_); } \
        } else {                                                                                                         \
        const Q8Item n_ = q8_decode(g, pv, total);                                                                   \
        p_krem = n_.kend - n_.kbeg;                                                                                      \
        if (A_KC) { sa_base = (const unsigned char*)(A + ((long)n_.m0 * g.lda + n_.kbeg)); sa_rec = (int)((((long)(g.M - n_.m0)) * g.lda - n_.kbeg) * 2); } \
        else      { sa_base = (const unsigned char*)(A + ((long)n_.kbeg * g.lda + n_.m0)); sa_rec = (int)(((long)p_krem * g.lda - n_.m0) * 2); }             \
        if (B_KC) { sb_base = (const unsigned char*)(B + ((long)n_.n0 * g.ldb + n_.kbeg)); sb_rec = (int)((((long)(g.N - n_.n0)) * g.ldb - n_.kbeg) * 2); } \
        else      { sb_base = (const unsigned char*)(B + ((long)n_.kbeg * g.ldb + n_.n0)); sb_rec = (int)(((long)p_krem * g.ldb - n_.n0) * 2); }             \
        }                                                                                                                \
    } while (0)
    if (!pdone) Q8_NEXT_ITEM();
    int wA = 0, wB = 0;                                // ring slots the next A / B half-tile goes to
    // stage pieces PCS (bit 0: piece `wave`, bit 1: piece `8 + wave`) of part PART (0: A_0, 1: B_0, 2: A_1, 3: B_1) of the K tile
    // being staged; the cursor moves on with the second piece of part 3
#define Q8_STAGE_PCS(PART, PCS)                                                                                          \
    do {                                                                                                                 \
        if (!pdone) {                                                                                                    \
            if (!(DBG & 2)) {                                                                                            \
                if ((PART) == 0) q8_stage_half<A_KC, (PCS)>(sa_base, sa_rec, p_krem, lds + wA * Q8_HALF, voffA, wave, lane);                         \
                if ((PART) == 1) q8_stage_half<B_KC, (PCS)>(sb_base, sb_rec, p_krem, lds + (NSLOT + wB) * Q8_HALF, voffB, wave, lane);             \
                if ((PART) == 2) q8_stage_half<A_KC, (PCS)>(sa_base + a_half, sa_rec - a_half, p_krem, lds + wA * Q8_HALF, voffA, wave, lane);       \
                if ((PART) == 3) q8_stage_half<B_KC, (PCS)>(sb_base + b_half, sb_rec - b_half, p_krem, lds + (NSLOT + wB) * Q8_HALF, voffB, wave, lane); \
            }                                                                                                            \
            if ((PCS) & 2) {                                                                                             \
                if ((PART) == 0 || (PART) == 2) wA = wA == NSLOT - 1 ? 0 : wA + 1; else wB = wB == NSLOT - 1 ? 0 : wB + 1; \
            }                                                                                                            \
            if ((PART) == 3 && ((PCS) & 2)) {                                                                            \
                p_krem -= 64;                                                                                            \
                if (!(DBG & 64)) { sa_base += a_step; sa_rec -= a_step; sb_base += b_step; sb_rec -= b_step; }           \
                if (p_krem <= 0) {                                                                                       \
                    pv += G;                                                                                             \
                    if (pv < total) Q8_NEXT_ITEM(); else pdone = true;                                                   \
                }                                                                                                        \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)
#define Q8_STAGE_PART(PART) Q8_STAGE_PCS(PART, 3)
    // every DMA of mine has landed, except that the `N_` youngest vector-memory operations (issued after it) may be pending
#define Q8_WAIT_DMA(N_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory")

    // ---- LEAN STREAM (SCH >= 1; round 4).  Counter evidence (profiles/r04_vendor_vs_q8_pmc.txt): the load interval of a phase is the
    // critical path -- the multiplying wave row waits at the barrier for the loading row, 37-49 % of all wave-cycles are parked -- and two
    // thirds of its instructions were bookkeeping (3.3x the vendor kernel's SALU, 6x its barrier / wait instructions per tile step).  So:
    //  * ONE descriptor per operand, valid for both half-tiles (the second half-tile is a per-lane offset), ADVANCED per K tile (two adds,
    //    a clamped subtract) instead of rebuilt per half-tile;
    //  * ring positions are LDS byte offsets (wave piece folded in): M0 is one move;
    //  * no `pdone` tests: an exhausted stream keeps issuing through a descriptor of ZERO records -- nothing is fetched, the slots it
    //    zero-fills have been consumed (same ring discipline), every counted wait keeps its count, and the kernel drains them before it ends;
    //  * the partial-K lane mask is folded into the per-lane offsets when the stream enters / leaves a tail K tile (twice per item at
    //    most), not evaluated in every part;
    //  * the decode of the NEXT output tile (a division chain) runs one tile ahead in a matrix interval, between the MFMAs; the stream's own
    //    bookkeeping -- ring advance, descriptor advance, the stream's item decode -- sits behind the DMA issue of the load interval (in the
    //    matrix interval it measured 1-3 % slower on every form: profiles/r04_gemm_lab_sch.txt).
    const unsigned char *qa = reinterpret_cast<const unsigned char*>(A), *qb = reinterpret_cast<const unsigned char*>(B);
    int qa_rec = 0, qb_rec = 0, q_krem = 1 << 30, qv = it_beg;
    bool q_tail = false;
    unsigned cvA[4] = {0u, 0u, 0u, 0u}, cvB[4] = {0u, 0u, 0u, 0u};   // per-lane source offsets [half * 2 + piece]
    int dA = wave * 1024, dB = NSLOT * Q8_HALF + wave * 1024;         // LDS byte offset the next A / B half-tile (this wave's first piece) goes to
    long q_lda = ITEMS ? -1 : g.lda, q_ldb = ITEMS ? -1 : g.ldb;
    auto q_cv = [&](bool tail) __attribute__((always_inline)) {   // (re)build the per-lane offsets for the current leading dimensions; tail: lanes past q_krem read as zero
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            cvA[i] = q8_voff<A_KC, ES>(i, wave, lane, q_lda); cvA[2 + i] = cvA[i] + (unsigned)(A_KC ? q_lda * 128 * ES : 256);
            cvB[i] = q8_voff<B_KC, ES>(i, wave, lane, q_ldb); cvB[2 + i] = cvB[i] + (unsigned)(B_KC ? q_ldb * 128 * ES : 256);
        }
        if ((A_KC || B_KC) && tail) {
            const int kc0 = (lane & 7) ^ (((wave * 8 + (lane >> 3)) >> 1) & 7);   // same key for all four pieces (they lie 64 / 128 rows apart)
            if (kc0 * (16 / ES) >= q_krem) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { if (A_KC) cvA[i] = 0xFFFFFF00u; if (B_KC) cvB[i] = 0xFFFFFF00u; }
            }
        }
    };
    // the stream enters item qv.  (A macro, not a lambda: inside a by-reference lambda the Q8_PROB selects become selects between the
    // ADDRESSES of the captured values, and the argument block ends up as a table in scratch that every item set-up loads from.)
#define Q9_ITEM()                                                                                                        \
    do {                                                                                                                 \
        if (ITEMS) {                                                                                                     \
            const Q8ItemRec r_ = item_at(qv);                                                                            \
            const long la_ = Q8_PROB(r_.prob, lda), lb_ = Q8_PROB(r_.prob, ldb);                                         \
            const bf16_t* A_ = reinterpret_cast<const bf16_t*>(uniform_ptr(Q8_PROB(r_.prob, A)));                        \
            const bf16_t* B_ = reinterpret_cast<const bf16_t*>(uniform_ptr(Q8_PROB(r_.prob, B)));                        \
            q_krem = r_.kend - r_.kbeg;                                                                                  \
            qa = (const unsigned char*)(A_ + ((long)r_.kbeg * la_ + r_.m0)); qa_rec = (int)(((long)q_krem * la_ - r_.m0) * 2); \
            qb = (const unsigned char*)(B_ + ((long)r_.kbeg * lb_ + r_.n0)); qb_rec = (int)(((long)q_krem * lb_ - r_.n0) * 2); \
            a_step = (int)la_ * 128; b_step = (int)lb_ * 128;                                                            \
            if (la_ != q_lda || lb_ != q_ldb) { q_lda = la_; q_ldb = lb_; q_cv(false); }                                 \
        } else {                                                                                                         \
            const Q8Item n_ = q8_decode<KT_SHIFT>(g, qv, total);                                                         \
            const unsigned char* Ab_ = reinterpret_cast<const unsigned char*>(g.A);                                      \
            const unsigned char* Bb_ = reinterpret_cast<const unsigned char*>(g.B);                                      \
            q_krem = n_.kend - n_.kbeg;                                                                                  \
            if (A_KC) { qa = Ab_ + ((long)n_.m0 * g.lda + n_.kbeg) * ES; qa_rec = (int)((((long)(g.M - n_.m0)) * g.lda - n_.kbeg) * ES); } \
            else      { qa = Ab_ + ((long)n_.kbeg * g.lda + n_.m0) * ES; qa_rec = (int)(((long)q_krem * g.lda - n_.m0) * ES); }             \
            if (B_KC) { qb = Bb_ + ((long)n_.n0 * g.ldb + n_.kbeg) * ES; qb_rec = (int)((((long)(g.N - n_.n0)) * g.ldb - n_.kbeg) * ES); } \
            else      { qb = Bb_ + ((long)n_.kbeg * g.ldb + n_.n0) * ES; qb_rec = (int)(((long)q_krem * g.ldb - n_.n0) * ES); }             \
        }                                                                                                                \
        qa_rec = max(qa_rec, 0); qb_rec = max(qb_rec, 0);                                                                \
    } while (0)
    // issue the two DMA instructions of part PART (0: A half 0, 1: B half 0, 2: A half 1, 3: B half 1) of the K tile being staged
#define Q9_ISSUE(PART)                                                                                                   \
    do {                                                                                                                 \
        if (!(DBG & 2)) {                                                                                                \
            typedef void __attribute__((address_space(3))) lds_void_;                                                    \
            constexpr bool isA_ = (((PART) & 1) == 0);                                                                   \
            constexpr int h_ = (PART) >> 1;                                                                              \
            const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)(isA_ ? qa : qb), 0, isA_ ? qa_rec : qb_rec, 0x00020000); \
            unsigned char* d_ = lds + (isA_ ? dA : dB);                                                                  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_void_*)d_, 16, (int)(isA_ ? cvA[2 * h_] : cvB[2 * h_]), 0, 0, 0);               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_void_*)(d_ + 8192), 16, (int)(isA_ ? cvA[2 * h_ + 1] : cvB[2 * h_ + 1]), 0, 0, 0); \
        }                                                                                                                \
    } while (0)
    // the bookkeeping behind part PART: ring position; behind part 3 the stream moves on one K tile (and, at an item's end, one item)
#define Q9_ADVANCE(PART)                                                                                                 \
    do {                                                                                                                 \
        if (((PART) & 1) == 0) { dA += Q8_HALF; if (dA >= NSLOT * Q8_HALF) dA -= NSLOT * Q8_HALF; }                      \
        else                   { dB += Q8_HALF; if (dB >= 2 * NSLOT * Q8_HALF) dB -= NSLOT * Q8_HALF; }                  \
        if ((PART) == 3) {                                                                                               \
            q_krem -= KT;                                                                                                \
            if (!(DBG & 64)) { qa += a_step; qb += b_step; qa_rec = max(qa_rec - a_step, 0); qb_rec = max(qb_rec - b_step, 0); } \
            if (q_krem <= 0) {                                                                                           \
                qv += G;                                                                                                 \
                if (qv < total) Q9_ITEM(); else { qa_rec = 0; qb_rec = 0; q_krem = 1 << 30; }                             \
            }                                                                                                            \
            if (A_KC || B_KC) {                                                                                          \
                const bool tl_ = q_krem < KT;                                                                            \
                if (tl_ != q_tail) { q_tail = tl_; q_cv(tl_); }                                                          \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)

    Q8Frag<A_KC> xm[4];                               // the fragments of one phase (a k-step of 16): 4 M-side + 2 N-side
    Q8Frag<B_KC> xn[2];
    int rA = 0, rB = 0;                               // ring slots of A_0 / B_0 of the K tile being multiplied
    float rs[4] = {0.f, 0.f, 0.f, 0.f}, rsp[4] = {0.f, 0.f, 0.f, 0.f};   // ROWSUM: running / finished-tile partial sums of row l31 of tile tm
    bool rs_on = false, rsp_on = false;
    int rsp_m0 = 0, rsp_prob = 0;
    // this lane's 8 contraction values of every M-side fragment summed into rs[]: four v_dot2c_f32_bf16 per fragment against a pair of
    // bf16 ones -- or zeros for the waves / tiles that do not sum (rs_ones), so the code is branch-free and can sit between the MFMAs
    // of a phase, where the matrix pipe hides it (no builtin lowers to the instruction on gfx950: inline asm, non-volatile)
    unsigned rs_ones = 0u;
#define Q8_RS_ACC(FM)                                                                                                    \
    do {                                                                                                                 \
        typedef unsigned int q8_u32x4_ __attribute__((ext_vector_type(4)));                                              \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                               \
            _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                                          \
                const q8_u32x4_ w_ = __builtin_bit_cast(q8_u32x4_, FM[t_].get());                                        \
                asm(ECAMP_DOT2C " %0, %1, %2" : "+v"(rs[t_]) : "v"(w_[j_]), "v"(rs_ones));                           \
            }                                                                                                            \
        }                                                                                                                \
    } while (0)

    // fragment I (0..5, in the order the MFMAs consume them: n0 m0 m1 n1 m2 m3) of k-step KS from the half-tiles at SM / SN
#define Q8_RD1(FM, FN, SM, SN, KS, I)                                                                                    \
    do {                                                                                                                 \
        constexpr int isn_ = ((I) == 0 || (I) == 3), idx_ = (I) == 0 ? 0 : (I) == 3 ? 1 : (I) < 3 ? (I) - 1 : (I) - 2;   \
        if (!(DBG & 32)) {                                                                        \
            if (isn_) { if (B_KC) FN[idx_].template read<idx_ * 4096>((SN) + offN[KS]); else FN[idx_].template read<(KS) * 4096>((SN) + offN[idx_]); } \
            else      { if (A_KC) FM[idx_].template read<idx_ * 4096>((SM) + offM[KS]); else FM[idx_].template read<(KS) * 4096>((SM) + offM[idx_]); } \
        }                                                                                                                \
    } while (0)
#define Q8_READ_GROUP(FM, FN, SM, SN, KS) \
    do { Q8_RD1(FM, FN, SM, SN, KS, 0); Q8_RD1(FM, FN, SM, SN, KS, 1); Q8_RD1(FM, FN, SM, SN, KS, 2); Q8_RD1(FM, FN, SM, SN, KS, 3); Q8_RD1(FM, FN, SM, SN, KS, 4); Q8_RD1(FM, FN, SM, SN, KS, 5); } while (0)
    // MFMA J (0..7) of a group: quadrants in the order (m0 n0) (m0 n1) (m1 n1) (m1 n0), two 32x32 tiles each
#define Q8_MFMA1(FM, FN, J, ZERO)                                                                                        \
    do {                                                                                                                 \
        constexpr int q_ = (J) >> 1, mh_ = q_ >> 1, nh_ = (q_ == 1 || q_ == 2) ? 1 : 0, tm_ = 2 * mh_ + ((J) & 1);       \
        if (DBG & 1) {                                                                                            \
            asm volatile("" ::"v"(FN[nh_].get()), "v"(FM[tm_].get()));                                                   \
            if (ZERO) acc[tm_][nh_] = zero16;                                                                            \
        } else {                                                                                                         \
            acc[tm_][nh_] = ECAMP_MFMA_32x32x16(FN[nh_].get(), FM[tm_].get(), (ZERO) ? zero16 : acc[tm_][nh_]); \
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
    const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc((F8 && EPI == 1 && g.q8_out) ? g.q8_out : g.C, 0, (int)(unsigned)((long)g.M * g.ldc), 0x00020000);
    float q8_max = 0.f;   // e4m3 form, GELU epilogue: this wave's running max|C| (one atomic per wave at the end of the kernel)
    const unsigned lane_o = (unsigned)((l31 * ldo + 8 * lh) * esz);          // lane part of an output offset
    const unsigned lane_p = (unsigned)((l31 * g.ldp + 8 * lh) * 2);
    const unsigned lane_g = (unsigned)((l31 * g.ldg + 8 * lh) * 2);
    const unsigned lane_r = (unsigned)((l31 * g.ldr + 8 * lh) * 2);
    const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 3 ? g.gmul : g.C), 0, (int)(unsigned)((long)g.M * g.ldg * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)((EPI == 2 || EPI == 3) && g.residual ? g.residual : g.C), 0,
                                                                        (int)(unsigned)((long)g.M * g.ldr * 2), 0x00020000);
    // Bias of the wave's 64 output columns, per lane the 4 x 8 values it adds (NH, group of 8, half): requested ONCE per output tile by
    // eight 16-B loads behind the previous tile's epilogue and landed by the first K tile's counted DMA wait (they are older than the
    // parts that wait leaves in flight).  Round 5: the epilogue used to fetch them per quadrant through scalar loads -- four exposed
    // scalar-memory round trips and 190 move / select instructions per tile with the matrix pipes idle (tools/epilogue_scale.py: 7-8 us
    // of every 256 x 256 tile are not its K loop).  Inline asm: hipcc drains the whole DMA queue (vmcnt(0)) in front of the first use of
    // a vector load it tracks; columns past N read 0 (descriptor bounds; N % 8 == 0).
    q8_u32x4 bq[2][2][2];
    typedef unsigned q8_u32x4s __attribute__((ext_vector_type(4)));
    constexpr bool HAS_BIAS = (EPI <= 2) && !ITEMS;
    const unsigned bias_lane = (unsigned)((wc * 64 + 8 * lh) * 4);
    auto bias_request = [&](int tn0) __attribute__((always_inline)) {
        if constexpr (HAS_BIAS) {
            if (g.bias) {
                const unsigned long long bp = (unsigned long long)g.bias;
                const q8_u32x4s rs_ = {(unsigned)bp, (unsigned)(bp >> 32) & 0xffffu, (unsigned)g.N * 4u, 0x00020000u};
                const unsigned vo = (unsigned)tn0 * 4u + bias_lane;
                // ONE statement: the descriptor may reach its SGPRs through v_readlane (the kernel spills scalars), and hipcc's hazard
                // recogniser does not look inside inline asm -- a VALU-written SGPR needs five wait states before a VMEM instruction reads it
                // (without the s_nop the e4m3 GELU kernel faulted on a stale descriptor word)
#define Q8_BOF(NH_, GQ_, H_) "n"(((NH_) * 32 + (GQ_) * 16 + (H_) * 4) * 4)
                asm volatile("s_nop 4\n\t"
                             "buffer_load_dwordx4 %0, %8, %9, 0 offen offset:%10\n\t"
                             "buffer_load_dwordx4 %1, %8, %9, 0 offen offset:%11\n\t"
                             "buffer_load_dwordx4 %2, %8, %9, 0 offen offset:%12\n\t"
                             "buffer_load_dwordx4 %3, %8, %9, 0 offen offset:%13\n\t"
                             "buffer_load_dwordx4 %4, %8, %9, 0 offen offset:%14\n\t"
                             "buffer_load_dwordx4 %5, %8, %9, 0 offen offset:%15\n\t"
                             "buffer_load_dwordx4 %6, %8, %9, 0 offen offset:%16\n\t"
                             "buffer_load_dwordx4 %7, %8, %9, 0 offen offset:%17"
                             : "=&v"(bq[0][0][0]), "=&v"(bq[0][0][1]), "=&v"(bq[0][1][0]), "=&v"(bq[0][1][1]),
                               "=&v"(bq[1][0][0]), "=&v"(bq[1][0][1]), "=&v"(bq[1][1][0]), "=&v"(bq[1][1][1])
                             : "v"(vo), "s"(rs_), Q8_BOF(0, 0, 0), Q8_BOF(0, 0, 1), Q8_BOF(0, 1, 0), Q8_BOF(0, 1, 1),
                               Q8_BOF(1, 0, 0), Q8_BOF(1, 0, 1), Q8_BOF(1, 1, 0), Q8_BOF(1, 1, 1));
#undef Q8_BOF
            }
        }
    };
    // (names the eight results behind the wait that landed them: nothing may copy the registers while the loads travel)
    auto bias_landed = [&]() __attribute__((always_inline)) {
        if constexpr (HAS_BIAS)
            asm volatile("" : "+v"(bq[0][0][0]), "+v"(bq[0][0][1]), "+v"(bq[0][1][0]), "+v"(bq[0][1][1]),
                              "+v"(bq[1][0][0]), "+v"(bq[1][0][1]), "+v"(bq[1][1][0]), "+v"(bq[1][1][1]));
    };
    // quadrant (MH, NH) of the output tile at (tm0, tn0), split tz: 2 tiles x 2 groups of 8 consecutive columns per lane.
    // The accumulators are only READ: the first K tile of the next output tile starts from C = 0 in the MFMA itself.
    auto store_quadrant = [&](int tm0, int tn0, int tz, auto mh_c, auto nh_c) {
        constexpr int MH = decltype(mh_c)::value, NH = decltype(nh_c)::value;
        constexpr int NTM = 2;   // 32-row blocks of this quadrant
        if (DBG & 4) {   // keep the accumulators (and so the MFMAs) alive without storing them
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm) asm volatile("" ::"v"(acc[2 * MH + tm][NH]));
            return;
        }
        if constexpr (ITEMS) {   // dense 256 x 256 f32 slab `tz`: no bounds, no scaling (the grouped reduce applies alpha), 8 stores
            const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(gr_slabs + (long)tz * 65536), 0, 262144, 0x00020000);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {
                    const unsigned o = (unsigned)((((wr * 128 + MH * 64 + tm * 32 + l31) * 256) + wc * 64 + NH * 32 + 16 * gq + 8 * lh) * 4);
                    const f32x16& a_ = acc[2 * MH + tm][NH];
                    __builtin_amdgcn_raw_buffer_store_b128((q8_u32x4){__float_as_uint(a_[8 * gq]), __float_as_uint(a_[8 * gq + 1]), __float_as_uint(a_[8 * gq + 2]), __float_as_uint(a_[8 * gq + 3])}, rI, o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128((q8_u32x4){__float_as_uint(a_[8 * gq + 4]), __float_as_uint(a_[8 * gq + 5]), __float_as_uint(a_[8 * gq + 6]), __float_as_uint(a_[8 * gq + 7])}, rI, o + 16, 0, 0);
                }
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
        if (F8 && g.alpha_dev2) {
            float ad = *(cfloat4*)g.alpha_dev2;
            asm volatile("" : "+s"(ad));
            al *= ad;
        }
        float bias[2][8];
#pragma unroll
        for (int gq = 0; gq < 2; ++gq)
#pragma unroll
            for (int r = 0; r < 8; ++r) bias[gq][r] = HAS_BIAS ? __uint_as_float(bq[NH][gq][r >> 2][r & 3]) : 0.f;
        // byte offsets: uniform tile part + lane part; a group past N gets offset 2^31, past every descriptor (< 2 GB, host-checked): dropped / reads 0
        unsigned uo[2][2], up[2][2], ug[2][2], ur[2][2];
#pragma unroll
        for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                const long row = mb + tm * 32, col = nb + 16 * gq;
                // (rows past M fall outside the descriptors by themselves -- except inside the stack of split-K slabs)
                const bool oob = (DBG & 1024) || (edge && (nb + 16 * gq + 8 * lh >= g.N)) || (EPI == 4 && g.partial && mb + tm * 32 + l31 >= g.M);   // (DBG 1024: every store dropped by its descriptor)
                uo[tm][gq] = oob ? 0x80000000u : (unsigned)(((EPI == 4 && g.partial ? (long)tz * g.M : 0) + row) * ldo + col) * esz + lane_o;
                if (EPI == 1) up[tm][gq] = oob ? 0x80000000u : (unsigned)((row * g.ldp + col) * 2) + lane_p;
                if (EPI == 3) ug[tm][gq] = oob ? 0x80000000u : (unsigned)((row * g.ldg + col) * 2) + lane_g;
                if (EPI == 2 || EPI == 3) ur[tm][gq] = oob ? 0x80000000u : (unsigned)((row * g.ldr + col) * 2) + lane_r;
            }
        // e4m3 form with the GELU epilogue: the e4m3 copy of the output for the next GEMM (8 bytes per lane and group)
        float q8_inv = 0.f;
        if (F8 && EPI == 1 && g.q8_out) {
            float qs = *(cfloat4*)g.q8_scale;
            asm volatile("" : "+s"(qs));
            q8_inv = 1.0f / fmaxf(qs, 1e-30f);
        }
        // all loads of the quadrant ahead of its first store
        q8_u32x4 qg[2][2], qr[2][2], qo[2][2][2];
#pragma unroll
        for (int tm = 0; tm < NTM; ++tm)
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
        for (int tm = 0; tm < NTM; ++tm)
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
                    if (g.act == 2) {   // saved derivative: the second output is gelu'(x), not x (uniform branch)
                        float d_[8];
#pragma unroll
                        for (int r = 0; r < 8; r += 2) {
                            f32x2_t g_;
                            const f32x2_t y_ = gelu_both_fast_f2((f32x2_t){rnd<bf16_t>(v[r]), rnd<bf16_t>(v[r + 1])}, g_);
                            v[r] = y_[0]; v[r + 1] = y_[1];
                            d_[r] = g_[0]; d_[r + 1] = g_[1];
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(q8_pack8(d_), rP, up[tm][gq], 0, ST_AUX);
                    } else {
                    __builtin_amdgcn_raw_buffer_store_b128(q8_pack8(v), rP, up[tm][gq], 0, ST_AUX);
#pragma unroll
                    for (int r = 0; r < 8; r += 2) {   // pair form: packed math (common.h)
                        const f32x2_t y_ = gelu_fast_f2((f32x2_t){rnd<bf16_t>(v[r]), rnd<bf16_t>(v[r + 1])});
                        v[r] = y_[0]; v[r + 1] = y_[1];
                    }
                    }
                    if (F8 && g.q8_out) {
                        float u[8];
                        // rows past M (an edge tile's zero-filled operand rows give gelu(bias) there) are not part of C: they must not
                        // enter the site's maximum; their stores are dropped by the descriptors anyway
                        const bool row_in = mb + tm * 32 + l31 < g.M && uo[tm][gq] != 0x80000000u;
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const float yr = rnd<bf16_t>(v[r]);   // what the bf16 output holds: fused == quantising C afterwards, bit for bit
                            if (row_in) q8_max = fmaxf(q8_max, fabsf(yr));
                            u[r] = fminf(fmaxf(yr * q8_inv, -448.f), 448.f);
                        }
                        int w0 = 0, w1 = 0;
                        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(u[0], u[1], w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(u[2], u[3], w0, true);
                        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(u[4], u[5], w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(u[6], u[7], w1, true);
                        typedef unsigned int q8_u32x2_ __attribute__((ext_vector_type(2)));
                        // (same element offsets as C, one byte per element instead of two)
                        __builtin_amdgcn_raw_buffer_store_b64((q8_u32x2_){(unsigned)w0, (unsigned)w1}, rQ, uo[tm][gq] == 0x80000000u ? 0x80000000u : (uo[tm][gq] >> 1), 0, 0);
                    }
                }
                if (EPI == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const f32x2_t lg_ = {h16_lo(qg[tm][gq][r]), h16_hi(qg[tm][gq][r])};
                        const f32x2_t gp_ = g.act == 2 ? lg_ : gelu_grad_fast_f2(lg_);   // act == 2: the forward pass saved gelu' itself (uniform branch)
                        v[2 * r] *= gp_[0];
                        v[2 * r + 1] *= gp_[1];
                    }
                }
                if (EPI == 2 || (EPI == 3 && g.residual)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[2 * r] += h16_lo(qr[tm][gq][r]);
                        v[2 * r + 1] += h16_hi(qr[tm][gq][r]);
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
                    __builtin_amdgcn_raw_buffer_store_b128(q8_pack8(v), rC, uo[tm][gq], 0, ST_AUX);
                }
            }
    };
    // ROWSUM: the finished tile's partial sums: the two k-halves of a row sit in lanes l and l+32; lanes < 32 add (buffer atomics:
    // exactly 4 instructions whatever the bounds, like the stores)
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)(ROWSUM && g.rowsum ? (void*)g.rowsum : g.C), 0, (int)(unsigned)((long)g.M * 4), 0x00020000);
    auto rowsum_flush = [&]() {
        typedef const float __attribute__((address_space(4))) cfloat4;
        float al = ITEMS ? Q8_PROB(rsp_prob, alpha_out) : g.alpha_out;
        const float* adp = ITEMS ? (const float*)uniform_ptr(Q8_PROB(rsp_prob, alpha_dev_out)) : g.alpha_dev_out;
        if (adp) { float ad = *(cfloat4*)adp; asm volatile("" : "+s"(ad)); al *= ad; }
        __amdgpu_buffer_rsrc_t rSi = rS;
        if (ITEMS) rSi = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(Q8_PROB(rsp_prob, rowsum)), 0, (int)(unsigned)((long)Q8_PROB(rsp_prob, M) * 4), 0x00020000);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float tot = (rsp[t] + __shfl_xor(rsp[t], 32, 64)) * al;
            const unsigned off = lh ? 0x80000000u : (unsigned)((rsp_m0 + wr * 128 + t * 32 + l31) * 4);
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(tot, rSi, off, 0, 0);
        }
        rsp_on = false;
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
    // ---- one PHASE = one k-step of 16 of a K tile: [load interval: the six fragment reads of this phase, one half-tile part of DMA,
    // the waits] barrier [matrix interval: 8 MFMAs back to back at raised priority] barrier.  Wave row 1 runs one barrier behind wave
    // row 0, so the two intervals of the two waves of a SIMD alternate (see header).  DMA of K tile t, one part per phase, in stream
    // order: A_1(t+1), B_1(t+1), A_0(t+2), B_0(t+2).  Ring reuse (5 half-tile slots per operand): A_1 / B_1 of K tile u overwrite
    // A_0 / B_0 of K tile u-2, A_0 / B_0 of K tile u the slot of A_1 / B_1 of K tile u-3; the last reads of a slot are retired
    // (lgkmcnt(0)) before the barrier that precedes the overwriting DMA's issue, for either wave row.  Phase 3 waits for the four
    // parts of K tile t+1 (the four DMA instructions of phases 2-3 are younger and stay in flight); each row's wait is followed by
    // a barrier before anyone reads K tile t+1.
    // PRE_HOOK (first phase of an output tile): the previous tile's epilogue -- its accumulators are overwritten by this phase's
    // MFMAs, which start from C = 0 -- and the bias-gradient atomics; all older than the DMA the next wait covers.
#define Q8_PHASE(KS, DPART, WAIT, ZERO, PRE_HOOK)                                                                     \
    do {                                                                                                                 \
        if (PRE_HOOK) {                                                                                                  \
            if (have_pend) { Q8_STORE_Q(pm0, pn0, pz, 0); Q8_STORE_Q(pm0, pn0, pz, 1); Q8_STORE_Q(pm0, pn0, pz, 2); Q8_STORE_Q(pm0, pn0, pz, 3); } \
            bias_request(cn0);                                                                                           \
            if (ROWSUM) { if (rsp_on) rowsum_flush(); }                                                                  \
            Q8_SB();                                                                                                     \
        }                                                                                                                \
        Q8_READ_GROUP(xm, xn, sM, sN, KS); Q8_SB();                                                                      \
        Q8_STAGE_PART(DPART); Q8_SB();                                                                                   \
        if (WAIT) { if (pdone) Q8_WAIT_DMA(0); else Q8_WAIT_DMA(4); }   /* in flight: A_0 and B_0 of K tile t+2 */   \
        if (!A_KC || !B_KC) { q8_wait4(xm); q8_wait2(xn); }                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); Q8_SB();                                                      \
        if (!(DBG & 16)) __builtin_amdgcn_s_barrier();                                                                   \
        Q8_SB();                                                                                                         \
        if (!(DBG & 8)) __builtin_amdgcn_s_setprio(1);                                                                   \
        if (ROWSUM) { Q8_RS_ACC(xm); }   /* no scheduling fence: the compiler spreads these between the MFMAs */          \
        Q8_MFMA1(xm, xn, 0, ZERO); Q8_MFMA1(xm, xn, 1, ZERO); Q8_MFMA1(xm, xn, 2, ZERO); Q8_MFMA1(xm, xn, 3, ZERO);      \
        Q8_MFMA1(xm, xn, 4, ZERO); Q8_MFMA1(xm, xn, 5, ZERO); Q8_MFMA1(xm, xn, 6, ZERO); Q8_MFMA1(xm, xn, 7, ZERO);      \
        Q8_SB();                                                                                                         \
        if (!(DBG & 8)) __builtin_amdgcn_s_setprio(0);                                                                   \
        if (!(DBG & 16)) __builtin_amdgcn_s_barrier();                                                                   \
        Q8_SB();                                                                                                         \
    } while (0)
#define Q8_KTILE(FIRST)                                                                                               \
    do {                                                                                                                 \
        const int ra_ = rA + wr, rb_ = rB + (wc >> 1);                                                                   \
        const unsigned char* sM = lds + (ra_ >= NSLOT ? ra_ - NSLOT : ra_) * Q8_HALF;                                    \
        const unsigned char* sN = lds + (NSLOT + (rb_ >= NSLOT ? rb_ - NSLOT : rb_)) * Q8_HALF;                          \
        Q8_PHASE(0, 2, false, FIRST, FIRST);                                                                          \
        Q8_PHASE(1, 3, false, false, false);                                                                          \
        Q8_PHASE(2, 0, false, false, false);                                                                          \
        Q8_PHASE(3, 1, true, false, false);                                                                           \
        rA = rA + 2 >= NSLOT ? rA + 2 - NSLOT : rA + 2;                                                                  \
        rB = rB + 2 >= NSLOT ? rB + 2 - NSLOT : rB + 2;                                                                  \
    } while (0)

    // ---- the lean-stream phase: [load interval: six fragment reads, the part's two DMA instructions, the bookkeeping behind the part,
    // the waits] barrier [matrix interval: MFMA 0-3, (MX_HOOK: the decode of the next output tile), MFMA 4-7] barrier.  The counted wait
    // of phase 3 covers K tile t+1: in flight behind it are A_0 / B_0 of K tile t+2 (4 instructions).
#define Q9_PHASE(KS, DPART, WAIT, ZERO, PRE_HOOK, MX_HOOK)                                                              \
    do {                                                                                                                 \
        if (PRE_HOOK) {                                                                                                  \
            if (have_pend) { Q8_STORE_Q(pm0, pn0, pz, 0); Q8_STORE_Q(pm0, pn0, pz, 1); Q8_STORE_Q(pm0, pn0, pz, 2); Q8_STORE_Q(pm0, pn0, pz, 3); } \
            bias_request(cn0);                                                                                           \
            if (ROWSUM) { if (rsp_on) rowsum_flush(); }                                                                  \
            Q8_SB();                                                                                                     \
        }                                                                                                                \
        Q8_READ_GROUP(xm, xn, sM, sN, KS); Q8_SB();                                                                      \
        Q9_ISSUE(DPART); Q8_SB();                                                                                        \
        Q9_ADVANCE(DPART); Q8_SB();                                                                                      \
        if (WAIT) Q8_WAIT_DMA(4);                                                                                        \
        if (!A_KC || !B_KC) { q8_wait4(xm); q8_wait2(xn); }                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); Q8_SB();                                                      \
        if (!(DBG & 16)) __builtin_amdgcn_s_barrier();                                                                   \
        Q8_SB();                                                                                                         \
        if (!(DBG & 8)) __builtin_amdgcn_s_setprio(1);                                                                   \
        Q8_MFMA1(xm, xn, 0, ZERO); Q8_MFMA1(xm, xn, 1, ZERO); Q8_SB();                                                   \
        Q8_MFMA1(xm, xn, 2, ZERO); Q8_MFMA1(xm, xn, 3, ZERO); Q8_SB();                                                   \
        if (MX_HOOK) { if (cv + G < total) Q9_CDECODE(cv + G); }                                                           \
        Q8_SB();                                                                                                         \
        if (ROWSUM) { Q8_RS_ACC(xm); }   /* no scheduling fence behind it: the compiler spreads these between the MFMAs */ \
        Q8_MFMA1(xm, xn, 4, ZERO); Q8_MFMA1(xm, xn, 5, ZERO); Q8_MFMA1(xm, xn, 6, ZERO); Q8_MFMA1(xm, xn, 7, ZERO);      \
        Q8_SB();                                                                                                         \
        if (!(DBG & 8)) __builtin_amdgcn_s_setprio(0);                                                                   \
        if (!(DBG & 16)) __builtin_amdgcn_s_barrier();                                                                   \
        Q8_SB();                                                                                                         \
    } while (0)
    // ---- the e4m3 phase: k-step KS8 (64 deep) x row half MH of the wave's tile.  MH = 0 also reads the two N-side fragments of the
    // k-step, which stay in registers for MH = 1.  Four MFMAs of 64 matrix-pipe cycles = the bf16 phase's 256.
    Q8Frag8 ym[2], yn[2];
#define Q9F_MFMA(TMI, NH, ZERO)                                                                                          \
    acc[2 * MH_ + (TMI)][NH] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(yn[NH].get(), ym[TMI].get(), (ZERO) ? zero16 : acc[2 * MH_ + (TMI)][NH], \
                                                                               0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F)
#define Q9F_PHASE(KS8, MH, DPART, WAIT, ZERO, PRE_HOOK, MX_HOOK)                                                         \
    do {                                                                                                                 \
        constexpr int MH_ = (MH);                                                                                        \
        if (PRE_HOOK) {                                                                                                  \
            if (have_pend) { Q8_STORE_Q(pm0, pn0, pz, 0); Q8_STORE_Q(pm0, pn0, pz, 1); Q8_STORE_Q(pm0, pn0, pz, 2); Q8_STORE_Q(pm0, pn0, pz, 3); } \
            bias_request(cn0);                                                                                           \
            Q8_SB();                                                                                                     \
        }                                                                                                                \
        if ((MH) == 0) {                                                                                                 \
            yn[0].template read<0>(sN + offN[2 * (KS8)], sN + offN[2 * (KS8) + 1]);                                       \
        }                                                                                                                \
        ym[0].template read<(2 * (MH)) * 4096>(sM + offM[2 * (KS8)], sM + offM[2 * (KS8) + 1]);                           \
        ym[1].template read<(2 * (MH) + 1) * 4096>(sM + offM[2 * (KS8)], sM + offM[2 * (KS8) + 1]);                       \
        if ((MH) == 0) {                                                                                                 \
            yn[1].template read<4096>(sN + offN[2 * (KS8)], sN + offN[2 * (KS8) + 1]);                                    \
        }                                                                                                                \
        Q8_SB();                                                                                                         \
        Q9_ISSUE(DPART); Q8_SB();                                                                                        \
        if (WAIT) Q8_WAIT_DMA(4);                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); Q8_SB();                                                      \
        __builtin_amdgcn_s_barrier();                                                                                    \
        Q8_SB();                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                                   \
        Q9F_MFMA(0, 0, ZERO); Q9F_MFMA(0, 1, ZERO); Q8_SB();                                                             \
        Q9_ADVANCE(DPART);   /* (load-interval placement measured equal on this form: tools/fp8_bench.py) */                   \
        if (MX_HOOK) { if (cv + G < total) Q9_CDECODE(cv + G); }                                                         \
        Q8_SB();                                                                                                         \
        Q9F_MFMA(1, 1, ZERO); Q9F_MFMA(1, 0, ZERO);                                                                      \
        Q8_SB();                                                                                                         \
        __builtin_amdgcn_s_setprio(0);                                                                                   \
        __builtin_amdgcn_s_barrier();                                                                                    \
        Q8_SB();                                                                                                         \
    } while (0)
#define Q9F_KTILE(FIRST)                                                                                              \
    do {                                                                                                                 \
        const int ra_ = rA + wr, rb_ = rB + (wc >> 1);                                                                   \
        const unsigned char* sM = lds + (ra_ >= NSLOT ? ra_ - NSLOT : ra_) * Q8_HALF;                                    \
        const unsigned char* sN = lds + (NSLOT + (rb_ >= NSLOT ? rb_ - NSLOT : rb_)) * Q8_HALF;                          \
        Q9F_PHASE(0, 0, 2, false, FIRST, FIRST, false);                                                               \
        Q9F_PHASE(0, 1, 3, false, FIRST, false, FIRST);                                                               \
        Q9F_PHASE(1, 0, 0, false, false, false, false);                                                               \
        Q9F_PHASE(1, 1, 1, true, false, false, false);                                                                \
        rA = rA + 2 >= NSLOT ? rA + 2 - NSLOT : rA + 2;                                                                  \
        rB = rB + 2 >= NSLOT ? rB + 2 - NSLOT : rB + 2;                                                                  \
    } while (0)
#define Q9_KTILE(FIRST)                                                                                               \
    do {                                                                                                                 \
        const int ra_ = rA + wr, rb_ = rB + (wc >> 1);                                                                   \
        const unsigned char* sM = lds + (ra_ >= NSLOT ? ra_ - NSLOT : ra_) * Q8_HALF;                                    \
        const unsigned char* sN = lds + (NSLOT + (rb_ >= NSLOT ? rb_ - NSLOT : rb_)) * Q8_HALF;                          \
        Q9_PHASE(0, 2, false, FIRST, FIRST, false);                                                                   \
        Q9_PHASE(1, 3, false, false, false, FIRST);                                                                   \
        Q9_PHASE(2, 0, false, false, false, false);                                                                   \
        Q9_PHASE(3, 1, true, false, false, false);                                                                    \
        rA = rA + 2 >= NSLOT ? rA + 2 - NSLOT : rA + 2;                                                                  \
        rB = rB + 2 >= NSLOT ? rB + 2 - NSLOT : rB + 2;                                                                  \
    } while (0)

    if (DBG & 512) {   // probe: workgroups start in four phases ~9 us apart (are the lock-step epilogue bursts of a round the cost?)
        const long long t0 = wall_clock64();   // 100 MHz
        const long long d = (long long)(blockIdx.x & 3) * 900;
        while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(8);
    }
    // prologue: K tile 0 and A_0 / B_0 of K tile 1 issued, K tile 0 landed and published
    if constexpr (SCH == 0) {
        Q8_STAGE_PART(0); Q8_STAGE_PART(1); Q8_STAGE_PART(2); Q8_STAGE_PART(3);
        Q8_STAGE_PART(0); Q8_STAGE_PART(1);
    } else {
        if (!ITEMS) q_cv(false);
        if (qv < total) Q9_ITEM();
        Q9_ISSUE(0); Q9_ADVANCE(0); Q9_ISSUE(1); Q9_ADVANCE(1); Q9_ISSUE(2); Q9_ADVANCE(2); Q9_ISSUE(3); Q9_ADVANCE(3);
        Q9_ISSUE(0); Q9_ADVANCE(0); Q9_ISSUE(1); Q9_ADVANCE(1);
    }
    Q8_WAIT_DMA(4);    // in flight: A_0 and B_0 of the second K tile
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // wave row 1 runs one barrier behind (wave row 0 makes up for it at the end)
    if constexpr ((DBG & 32) != 0 && A_KC && B_KC) {   // timing decomposition: real (random) fragments, read once and never again
#pragma unroll
        for (int i = 0; i < 4; ++i) xm[i].v = *reinterpret_cast<const hw_bf16x8*>(lds + offM[0] + i * 4096);
#pragma unroll
        for (int i = 0; i < 2; ++i) xn[i].v = *reinterpret_cast<const hw_bf16x8*>(lds + NSLOT * Q8_HALF + offN[0] + i * 4096);
    }

    // ---- main loop over this workgroup's output tiles (every tile has at least two K tiles: the host guarantees K/split >= 128)
    bool have_pend = false;
    int pm0 = 0, pn0 = 0, pz = 0;
    // lean stream: the next output tile is decoded one tile ahead, in a matrix interval of the current tile's first K tile
    int nm0 = 0, nn0 = 0, nz = 0, ncnt = 0;
    bool nrs = false;
#define Q9_CDECODE(V_)                                                                                                   \
    do {                                                                                                                 \
        if (ITEMS) {                                                                                                     \
            const Q8ItemRec cr = item_at(V_);                                                                            \
            nm0 = cr.m0; nn0 = cr.prob; nz = cr.slab; ncnt = (cr.kend - cr.kbeg + 63) >> 6;                              \
            if (ROWSUM) nrs = (cr.flags & 1) != 0 && wc == 0 && Q8_PROB(cr.prob, rowsum) != nullptr;                     \
        } else {                                                                                                         \
            const Q8Item cit = q8_decode<KT_SHIFT>(g, V_, total);                                                        \
            nm0 = cit.m0; nn0 = cit.n0; nz = cit.z; ncnt = cit.nt;                                                       \
            if (ROWSUM) nrs = g.rowsum != nullptr && wc == 0 && cit.ncol == 0;                                           \
        }                                                                                                                \
    } while (0)
    if constexpr (SCH != 0) {
        if (it_beg < total) Q9_CDECODE(it_beg);
        for (int cv = it_beg; cv < total; cv += G) {
            const int cm0 = nm0, cn0 = nn0, cz = nz, cnt = ncnt;
            if (ROWSUM) { rs_on = nrs; rs_ones = rs_on ? H16_ONE_X2 : 0u; }
            if constexpr (F8) {
                Q9F_KTILE(true); bias_landed();
#pragma unroll 1
                for (int t = 1; t < cnt; ++t) Q9F_KTILE(false);
            } else {
                Q9_KTILE(true); bias_landed();
#pragma unroll 1
                for (int t = 1; t < cnt; ++t) Q9_KTILE(false);
            }
            have_pend = true; pm0 = cm0; pn0 = cn0; pz = cz;
            if (ROWSUM) {
                rsp_on = rs_on; rsp_m0 = cm0; rsp_prob = cn0;
#pragma unroll
                for (int t = 0; t < 4; ++t) { rsp[t] = rs[t]; rs[t] = 0.f; }
            }
        }
        Q8_WAIT_DMA(0);   // the zero-length loads an exhausted stream keeps issuing still write their (zero) pieces into this workgroup's LDS
    } else
    for (int cv = it_beg; cv < total; cv += G) {
        int cm0, cn0, cz, cnt;   // ITEMS: cn0 carries the problem index and cz the slab index into the epilogue
        if (ITEMS) {
            const Q8ItemRec cr = item_at(cv);
            cm0 = cr.m0; cn0 = cr.prob; cz = cr.slab; cnt = (cr.kend - cr.kbeg + 63) >> 6;
            if (ROWSUM) { rs_on = (cr.flags & 1) != 0 && wc == 0 && Q8_PROB(cr.prob, rowsum) != nullptr; rs_ones = rs_on ? H16_ONE_X2 : 0u; }
        } else {
            const Q8Item cit = q8_decode(g, cv, total);
            cm0 = cit.m0; cn0 = cit.n0; cz = cit.z; cnt = cit.nt;
            if (ROWSUM) { rs_on = g.rowsum != nullptr && wc == 0 && cit.ncol == 0; rs_ones = rs_on ? H16_ONE_X2 : 0u; }
        }
        Q8_KTILE(true); bias_landed();
#pragma unroll 1
        for (int t = 1; t < cnt; ++t) Q8_KTILE(false);
        have_pend = true; pm0 = cm0; pn0 = cn0; pz = cz;
        if (ROWSUM) {
            rsp_on = rs_on; rsp_m0 = cm0; rsp_prob = cn0;
#pragma unroll
            for (int t = 0; t < 4; ++t) { rsp[t] = rs[t]; rs[t] = 0.f; }
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();   // wave row 0's share of the row lag
    if (have_pend) { Q8_STORE_Q(pm0, pn0, pz, 0); Q8_STORE_Q(pm0, pn0, pz, 1); Q8_STORE_Q(pm0, pn0, pz, 2); Q8_STORE_Q(pm0, pn0, pz, 3); }
    if (ROWSUM) { if (rsp_on) rowsum_flush(); }
    if (F8 && EPI == 1 && g.q8_out) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q8_max = fmaxf(q8_max, __shfl_xor(q8_max, o, 64));
        if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(g.q8_amax + ((blockIdx.x + wave) & 15) * 32), __float_as_uint(q8_max));
    }
#undef Q8_STORE_Q
#undef Q8_SB
#undef Q8_READ_GROUP
#undef Q8_MFMA1
#undef Q8_RD1
#undef Q8_STAGE_PART
#undef Q8_STAGE_PCS
#undef Q8_RS_ACC
#undef Q8_KTILE
#undef Q8_PHASE
#undef Q9_KTILE
#undef Q9F_KTILE
#undef Q9F_PHASE
#undef Q9F_MFMA
#undef Q9_PHASE
#undef Q9_ISSUE
#undef Q9_ADVANCE
#undef Q9_ITEM
#undef Q9_CDECODE
#undef Q8_NEXT_ITEM
#undef Q8_WAIT_DMA
}
#undef Q8_PROB
#undef Q8_VAL_

template <bool A_KC, bool B_KC, int EPI, int DBG = 0, bool ROWSUM = false, int SCH = 0>
__global__ __launch_bounds__(512) void gemm_bf16_q8_kernel(GemmArgs g) {
    Q8Group none;   // never read in this form
    q8_body<A_KC, B_KC, EPI, DBG, ROWSUM, false, SCH>(g, none);
}
// e4m3 forward form (per-tensor scales in g.alpha_dev / g.alpha_dev2)
template <int EPI>
__global__ __launch_bounds__(512) void gemm_f8_q8_kernel(GemmArgs g) {
    Q8Group none;
    q8_body<true, true, EPI, 0, false, false, 1, true>(g, none);
}
// grouped weight gradients: `g` only supplies the fields the item-table form does not take from the group (none of the operands)
template <bool ROWSUM, int SCH = 0>
__global__ __launch_bounds__(512) void gemm_bf16_q8_items_kernel(GemmArgs g, Q8Group GR) {
    q8_body<false, false, 4, 0, ROWSUM, true, SCH>(g, GR);
}
