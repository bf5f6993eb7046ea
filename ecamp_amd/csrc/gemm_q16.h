// "Q16": the persistent bf16 GEMM as FOUR waves (2 x 2, one per SIMD) on v_mfma_f32_16x16x32_bf16 (round 5).  Forward form
// (y = x w^T (+ b) (+ residual): nn.Linear forward, model_ecamp.py:233-234,254-255 / bert_modeling.py:131) and data-gradient form
// (dx = dy w (+ residual): its autograd), bf16 outputs, tile 256 x 256 (NW = 8) or 256 x 192 (NW = 6).
//
// Why it exists beside gemm_q8.h (profiles/r05_vendor_vs_q8.txt, r05_gemm_in_step_vs_lab.txt):
//  * POWER.  On random operands the chip is power-limited (rounds 3-4).  The same 4096^3 problem, back to back until the clocks settle: the
//    eight-wave 32x32x16 kernel holds 1770 MHz at the 1370 W cap, the four-wave 32x32x16 lab kernel 1862 MHz, THIS stream 2051 MHz -- the
//    16x16x32 MFMA costs less energy per FLOP, and although this compiler-scheduled loop spends ~10 % more cycles per K tile than the
//    eight-wave kernel it is 4-7 % faster in the sustained regime (bert inter forward 68.6 vs 74.1 us; the vendor library's hand-scheduled
//    kernel of the same shape, MT256x256x64_MI16x16x1: 66.7 us).
//  * TILE QUANTISATION, the largest GEMM loss of the step: 4.0 of 21.6 ms of in-step GEMM time is spent in last rounds with part of the
//    chip idle.  The model's 768-wide outputs (encoder proj / fc2, every dx of width 768, the BERT dense layers, the vocabulary head's
//    data gradient) are 150 / 384 tiles of 256 x 256 on 256 CUs (41 % / 25 % of the last round idle); as 256 x 192 tiles -- natural with a
//    16-wide MFMA: wave tile 128 x 96 = 8 x 6 MFMA tiles -- they are 200 / 512 tiles (0.78 / 2.0 rounds of 3/4-size tiles): -9..-12 % per
//    launch, and the vocabulary data gradient (K = 30000: half the chip idle for 700 us in its second round) 2.0 rounds exactly.
//
// Structure (the vendor kernel's, read as a yardstick: four waves of 128 x 128, fragments of k-step s+1 read under the MFMAs of k-step s,
// ONE other instruction behind every MFMA pair).  LDS images, rings (5 half-tile slots of 16 KB per operand), DMA pieces and the descriptor
// stream are gemm_q8.h's.  A K tile is two k-steps of 32; per k-step 8 NW MFMAs with, one behind every second MFMA: the 8 + NW fragment
// reads of the next k-step (across the K tile boundary too), the wave's DMA pieces of two half-tile parts, the stream bookkeeping.  One
// counted DMA wait + ONE barrier per K tile (between the k-steps) publishes K tile t+1 and retires every read of K tile t.
//
// Fragments.  M side (always contraction-contiguous): one ds_read_b128 = 16 rows x 32 k (lane: row l & 15, 16-B chunk 4 ks + (l >> 4)) under
// gemm_q8.h's XOR key (row >> 1) & 7 -- conflict-free for the 16-lane groups of ds_read_b128.  The N side is the MFMA's A operand, so a
// lane's four accumulator registers are four consecutive output COLUMNS; fragment row i of MFMA tile t is tile column
// 32 (t >> 1) + 8 (i >> 2) + 4 (t & 1) + (i & 3): the two tiles of a pair give a lane eight consecutive columns = one 16-B store, a store
// instruction writes 64 contiguous bytes of each of 16 rows.  Contraction-contiguous N side (forward form): that row set collides under
// the M-side key, so its half-tiles are staged under the key ((row >> 1) & 1) | (((row >> 3) & 3) << 1) (conflict-free, by enumeration).
// Strided N side (data-gradient form: w as it lies in HBM, 64 k-rows x 256 B per half-tile, chunk j of k-row kr at position
// j ^ ((kr & 3) << 2) as in gemm_q8.h): two ds_read_b64_tr_b16 per fragment -- lane (r = (l & 15) >> 2, q = l & 3) of 16-lane group g
// addresses k-row 32 ks + 8 g + r (+ 4 for the second read), columns 32 (t >> 1) + 8 q + 4 (t & 1) .. + 3, and receives column (l & 15)
// of the group's 4 x 16 block = its fragment row's four k values; the two k-groups of a 32-lane half meet on the same banks (2-way).
#pragma once
#include "gemm_q8.h"

typedef __attribute__((ext_vector_type(4))) float q16_f32x4;

__device__ __forceinline__ int q16_key_m(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int q16_key_n(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }

// N-side fragment of one MFMA tile: contraction-contiguous (one b128 the compiler tracks) or strided (two transpose reads in inline asm,
// halves named by the explicit wait at the k-step boundary -- hipcc puts vmcnt(0) in front of the builtin while an LDS-DMA is in flight)
template <bool KC> struct Q16FragN;
template <> struct Q16FragN<true> {
    hw_bf16x8 v;
    __device__ __forceinline__ void read(const unsigned char* p, int off) { v = *reinterpret_cast<const hw_bf16x8*>(p + off); }
    __device__ __forceinline__ hw_bf16x8 get() const { return v; }
};
template <> struct Q16FragN<false> {
    q8_v4s16 lo, hi;
    template <int OFF> __device__ __forceinline__ void read_tr(const unsigned char* p) {
        const unsigned a = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(OFF));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"(OFF + 4 * 256));
    }
    __device__ __forceinline__ hw_bf16x8 get() const {
        return __builtin_bit_cast(hw_bf16x8, (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
    }
};

// EPI: 0 bf16 C = alpha*acc (+bias)   2 ... + residual.   NW: MFMA tiles per wave along N (8: 256-column tile, 6: 192-column tile).
// B_KC: the N-side operand is contraction-contiguous (forward form) or strided (data-gradient form).
template <int EPI, int NW, bool B_KC, int DBG = 0>   // DBG (lab only): 1 no MFMA, 2 no DMA, 4 no fragment reads, 8 conflict-free (wrong) transpose reads
__device__ __forceinline__ void q16_body(const GemmArgs& g) {
    constexpr int NSLOT = 5;
    constexpr int TN = 32 * NW;                       // tile columns
    constexpr int NPB = B_KC ? NW / 2 : 4;            // DMA pieces of a B half-tile per wave (kc: 16 NW rows = 2 NW pieces over 4 waves; strided: always 16)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // A ring (5 x 16 KB) | B ring (5 x 16 KB); the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int nbn = (g.N + TN - 1) / TN;
    const int total = g.nbm * nbn, G = (int)gridDim.x, it_beg = (int)blockIdx.x;
    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(g.A);
    const unsigned char* Bb = reinterpret_cast<const unsigned char*>(g.B);
    const int l15 = lane & 15, lq = lane >> 4;
    // tile order: as q8_decode (8 M-blocks walked for one N-block before the next; XCD-contiguous ranges), with TN-wide N blocks
    auto decode = [&](int v, int& m0, int& n0) __attribute__((always_inline)) {
        const unsigned f = (unsigned)xcd_remap(v, total);
        const unsigned gw = 8u * (unsigned)nbn, grp = f / gw, in = f - grp * gw, first = grp * 8u;
        const unsigned gsz = min(8u, (unsigned)g.nbm - first);
        const unsigned nb = in / gsz, mb = first + (in - nb * gsz);
        m0 = __builtin_amdgcn_readfirstlane((int)mb * 256);
        n0 = __builtin_amdgcn_readfirstlane((int)nb * TN);
    };
    const int nt = (g.K + 63) >> 6;

    // per-lane fragment offsets inside a half-tile
    unsigned offM[2], offN[4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) offM[ks] = (unsigned)(l15 * 128 + (((4 * ks + lq) ^ q16_key_m(l15)) << 4));
    if (B_KC) {      // per k-step of 32; MFMA tile t at + (t & 1) * 512 + (t >> 1) * 4096
        const int nrow = 8 * (l15 >> 2) + (l15 & 3);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offN[ks] = (unsigned)(nrow * 128 + (((4 * ks + lq) ^ q16_key_n(nrow)) << 4));
        offN[2] = offN[3] = 0;
    } else {         // per tile pair T = t >> 1 (the XOR touches the pair bits); tile parity at + 8 B, k-step at + 8192, second read at + 1024
        const int r = l15 >> 2, q = l15 & 3;
#pragma unroll
        for (int T = 0; T < 4; ++T) offN[T] = (unsigned)((8 * lq + r) * 256 + ((((4 * T + q) ^ (r << 2)) & 15) << 4)) + ((DBG & 8) ? (unsigned)(lq & 1) * 8u : 0u);
        // (DBG & 8, lab only, WRONG results: the odd k-groups of a 32-lane half read the other 8-byte half of their chunks -- the pattern
        // without the 2-way bank meeting of the two k-groups; prices what that meeting costs: profiles/r06_q16_dgrad_conflicts.txt)
    }
    q16_f32x4 acc[8][NW];

    // ---- the operand stream (lean, as gemm_q4.h): wave w owns pieces 4 w .. 4 w + 3 of an A half-tile and NPB pieces of a B half-tile
    const unsigned char *qa = Ab, *qb = Bb;
    int qa_rec = 0, qb_rec = 0, q_krem = 1 << 30, qv = it_beg;
    bool q_tail = false;
    unsigned cvA[8], cvB[2 * NPB];   // [half * pieces + j]
    int dA = wave * 4096, dB = NSLOT * Q8_HALF + wave * NPB * 1024;
    const int b_step = B_KC ? 128 : (int)g.ldb * 128;   // bytes per K tile along the N-side operand
    auto q_cv = [&](bool tail) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (wave * 4 + j) * 8 + (lane >> 3);
            const int kc = (lane & 7) ^ q16_key_m(row);
            cvA[j] = (unsigned)((long)row * g.lda * 2 + kc * 16); cvA[4 + j] = cvA[j] + (unsigned)(g.lda * 256);
            if (tail && kc * 8 >= q_krem) cvA[j] = cvA[4 + j] = 0xFFFFFF00u;
        }
#pragma unroll
        for (int j = 0; j < NPB; ++j) {
            if (B_KC) {
                const int row = (wave * NPB + j) * 8 + (lane >> 3);
                const int kc = (lane & 7) ^ q16_key_n(row);
                cvB[j] = (unsigned)((long)row * g.ldb * 2 + kc * 16); cvB[NPB + j] = cvB[j] + (unsigned)(g.ldb * 32 * NW);   // second half-tile: 16 NW rows on
                if (tail && kc * 8 >= q_krem) cvB[j] = cvB[NPB + j] = 0xFFFFFF00u;
            } else {   // 4 k-rows x 256 B per piece (k-rows past the contraction fall outside the descriptor: zero)
                const int kr = (wave * 4 + j) * 4 + (lane >> 4);
                const int oc = (lane & 15) ^ ((kr & 3) << 2);
                cvB[j] = (unsigned)(((long)kr * g.ldb + oc * 8) * 2); cvB[NPB + j] = cvB[j] + (unsigned)(32 * NW);             // second half-tile: 16 NW columns on
            }
        }
    };
#define Q16_ITEM()                                                                                                       \
    do {                                                                                                                 \
        int m0_, n0_;                                                                                                    \
        decode(qv, m0_, n0_);                                                                                            \
        q_krem = g.K;                                                                                                    \
        qa = Ab + ((long)m0_ * g.lda) * 2; qa_rec = (int)((((long)(g.M - m0_)) * g.lda) * 2);                            \
        if (B_KC) { qb = Bb + ((long)n0_ * g.ldb) * 2; qb_rec = (int)((((long)(g.N - n0_)) * g.ldb) * 2); }              \
        else      { qb = Bb + (long)n0_ * 2; qb_rec = (int)(((long)g.K * g.ldb - n0_) * 2); }                            \
        qa_rec = max(qa_rec, 0); qb_rec = max(qb_rec, 0);                                                                \
    } while (0)
    typedef void __attribute__((address_space(3))) lds_void_;
    // piece J of part PART (0: A half 0, 1: B half 0, 2: A half 1, 3: B half 1); B parts have NPB pieces (J >= NPB: nothing)
#define Q16_ISSUE1(PART, J)                                                                                              \
    do {                                                                                                                 \
        constexpr bool isA_ = (((PART) & 1) == 0);                                                                       \
        constexpr int h_ = (PART) >> 1;                                                                                  \
        if (!(DBG & 2) && (isA_ || (J) < NPB)) {                                                                         \
            const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)(isA_ ? qa : qb), 0, isA_ ? qa_rec : qb_rec, 0x00020000); \
            unsigned char* d_ = lds + (isA_ ? dA : dB) + (J) * 1024;                                                     \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_void_*)d_, 16, (int)(isA_ ? cvA[4 * h_ + ((J) & 3)] : cvB[NPB * h_ + ((J) < NPB ? (J) : 0)]), 0, 0, 0); \
        }                                                                                                                \
    } while (0)
#define Q16_ADVANCE(PART)                                                                                                \
    do {                                                                                                                 \
        if (((PART) & 1) == 0) { dA += Q8_HALF; if (dA >= NSLOT * Q8_HALF) dA -= NSLOT * Q8_HALF; }                      \
        else                   { dB += Q8_HALF; if (dB >= 2 * NSLOT * Q8_HALF) dB -= NSLOT * Q8_HALF; }                  \
        if ((PART) == 3) {                                                                                               \
            q_krem -= 64;                                                                                                \
            qa += 128; qb += b_step; qa_rec = max(qa_rec - 128, 0); qb_rec = max(qb_rec - b_step, 0);                    \
            if (q_krem <= 0) {                                                                                           \
                qv += G;                                                                                                 \
                if (qv < total) Q16_ITEM(); else { qa_rec = 0; qb_rec = 0; q_krem = 1 << 30; }                           \
            }                                                                                                            \
            const bool tl_ = q_krem < 64;                                                                                \
            if (tl_ != q_tail) { q_tail = tl_; q_cv(tl_); }                                                              \
        }                                                                                                                \
    } while (0)
#define Q16_ISSUE_ALL(PART) do { Q16_ISSUE1(PART, 0); Q16_ISSUE1(PART, 1); Q16_ISSUE1(PART, 2); Q16_ISSUE1(PART, 3); } while (0)

    // ---- epilogue: MFMA tile pair (2 p, 2 p + 1) of M tile i = 16 rows x 32 columns; a lane holds row l15 and the eight columns 8 lq .. 8 lq + 7
    const long ldo = g.ldc;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)(unsigned)((long)g.M * ldo * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 2 && g.residual ? g.residual : g.C), 0, (int)(unsigned)((long)g.M * g.ldr * 2), 0x00020000);
    const unsigned lane_o = (unsigned)((l15 * ldo + 8 * lq) * 2), lane_r = (unsigned)((l15 * g.ldr + 8 * lq) * 2);
#define Q16_SB() __builtin_amdgcn_sched_barrier(0)
    // 192-column tiles (the register budget allows it: 140-164 of 256 VGPRs): the bias of the wave's 96 columns, per lane the 3 x 8 values
    // it adds, is requested ONCE per output tile by six 16-B loads at the head of the tile's first K tile and landed by that K tile's
    // counted DMA wait -- as in gemm_q8.h (one s_nop-opened asm statement; tests/test_isa.py checks both invariants on the built code).
    // The 256-column form has no registers to spare and fetches it per 32-column block through scalar loads (below).
    constexpr bool BIAS_PF = (NW == 6);
    q8_u32x4 bq[3][2];
    typedef unsigned q16_u32x4s __attribute__((ext_vector_type(4)));
    const unsigned bias_lane = (unsigned)((wc * (16 * NW) + 8 * lq) * 4);
    auto bias_request = [&](int tn0) __attribute__((always_inline)) {
        if constexpr (BIAS_PF) {
            if (g.bias) {
                const unsigned long long bp = (unsigned long long)g.bias;
                const q16_u32x4s rs_ = {(unsigned)bp, (unsigned)(bp >> 32) & 0xffffu, (unsigned)g.N * 4u, 0x00020000u};
                const unsigned vo = (unsigned)tn0 * 4u + bias_lane;
                asm volatile("s_nop 4\n\t"
                             "buffer_load_dwordx4 %0, %6, %7, 0 offen\n\t"
                             "buffer_load_dwordx4 %1, %6, %7, 0 offen offset:16\n\t"
                             "buffer_load_dwordx4 %2, %6, %7, 0 offen offset:128\n\t"
                             "buffer_load_dwordx4 %3, %6, %7, 0 offen offset:144\n\t"
                             "buffer_load_dwordx4 %4, %6, %7, 0 offen offset:256\n\t"
                             "buffer_load_dwordx4 %5, %6, %7, 0 offen offset:272"
                             : "=&v"(bq[0][0]), "=&v"(bq[0][1]), "=&v"(bq[1][0]), "=&v"(bq[1][1]), "=&v"(bq[2][0]), "=&v"(bq[2][1])
                             : "v"(vo), "s"(rs_));
            }
        }
    };
    auto bias_landed = [&]() __attribute__((always_inline)) {
        if constexpr (BIAS_PF) asm volatile("" : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[2][0]), "+v"(bq[2][1]));
    };
    // one 32-column block P of the wave's tile, all eight 16-row MFMA tiles of it: the bias of the block's columns is fetched ONCE (wave-uniform
    // scalar loads: lgkmcnt, they do not touch the DMA queue's vmcnt), the eight residual segments are requested together ahead of the first
    // store (a vector load makes hipcc wait vmcnt(0) at its first use: one exposed latency per block, four or three per tile -- per 16-row
    // tile it was thirty-two per tile and cost the step 1.7 ms)
    auto store_block = [&](int tm0, int tn0, auto p_c) __attribute__((always_inline)) {
        constexpr int P = decltype(p_c)::value;
        typedef const float __attribute__((address_space(4))) cfloat4;
        const int nb = tn0 + wc * (16 * NW) + P * 32;
        float al = g.alpha;
        if (g.alpha_dev) { float ad = *(cfloat4*)g.alpha_dev; asm volatile("" : "+s"(ad)); al *= ad; }
        const bool oob = nb + 8 * lq >= g.N;     // (N % 8 == 0, host-checked: a group of 8 columns is inside or outside as a whole)
        float bsel[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) bsel[r] = 0.f;
        if (BIAS_PF && g.bias) {
#pragma unroll
            for (int r = 0; r < 8; ++r) bsel[r] = __uint_as_float(bq[P < 3 ? P : 0][r >> 2][r & 3]);
        } else if (g.bias) {
            // each group of 8 columns is clamped on its own, so a group inside N reads exactly its columns (a clamped group belongs to lanes
            // whose store is dropped anyway)
            cfloat4* b0 = (cfloat4*)(g.bias + min(nb, g.N - 8));
            cfloat4* b1 = (cfloat4*)(g.bias + min(nb + 8, g.N - 8));
            cfloat4* b2 = (cfloat4*)(g.bias + min(nb + 16, g.N - 8));
            cfloat4* b3 = (cfloat4*)(g.bias + min(nb + 24, g.N - 8));
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float x0 = b0[r], x1 = b1[r], x2 = b2[r], x3 = b3[r];
                asm volatile("" : "+s"(x0), "+s"(x1), "+s"(x2), "+s"(x3));
                bsel[r] = lq == 0 ? x0 : lq == 1 ? x1 : lq == 2 ? x2 : x3;
            }
        }
        unsigned uo[8];
        q8_u32x4 qr[8];
#pragma unroll
        for (int I = 0; I < 8; ++I) {
            const int mb = tm0 + wr * 128 + I * 16;
            uo[I] = oob ? 0x80000000u : (unsigned)(((long)mb * ldo + nb) * 2) + lane_o;
            if (EPI == 2) {
                const unsigned ur = oob ? 0x80000000u : (unsigned)(((long)mb * g.ldr + nb) * 2) + lane_r;
                qr[I] = __builtin_amdgcn_raw_buffer_load_b128(rR, ur, 0, 0);
            }
        }
        Q16_SB();
#pragma unroll
        for (int I = 0; I < 8; ++I) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {   // explicit AGPR reads (left to itself hipcc copies every accumulator to VGPRs behind the K loop)
                float x0, x1;
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x0) : "a"(acc[I][2 * P][r]));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x1) : "a"(acc[I][2 * P + 1][r]));
                v[r] = fmaf(x0, al, bsel[r]); v[4 + r] = fmaf(x1, al, bsel[4 + r]);
            }
            if (EPI == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[2 * r] += h16_lo(qr[I][r]); v[2 * r + 1] += h16_hi(qr[I][r]); }
            }
            __builtin_amdgcn_raw_buffer_store_b128(q8_pack8(v), rC, uo[I], 0, 0);
            Q16_SB();
        }
    };
#define Q16_STORE_TILE(TM0, TN0)                                                                                         \
    do {                                                                                                                 \
        store_block(TM0, TN0, std::integral_constant<int, 0>()); store_block(TM0, TN0, std::integral_constant<int, 1>());  \
        store_block(TM0, TN0, std::integral_constant<int, 2>());                                                         \
        if (NW == 8) store_block(TM0, TN0, std::integral_constant<int, (NW == 8 ? 3 : 0)>());                            \
    } while (0)
    const q16_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- fragments: two register sets (k-step parity); the set of k-step s + 1 is read under the MFMAs of k-step s -- across the K tile
    // boundary too (k-step 1 reads k-step 0 of the next K tile, which the barrier in front of it has published)
    hw_bf16x8 fa[2][8];
    Q16FragN<B_KC> fb[2][NW];
#define Q16_RDA(S, KS, I, SM_) do { if (!(DBG & 4)) fa[S][I] = *reinterpret_cast<const hw_bf16x8*>((SM_) + offM[KS] + (I) * 2048); } while (0)
#define Q16_RDB(S, KS, T, SN_)                                                                                           \
    do {                                                                                                                 \
        if (!(DBG & 4)) {                                                                                                \
            if constexpr (B_KC) fb[S][T].read((SN_), (int)offN[KS] + ((T) & 1) * 512 + ((T) >> 1) * 4096);               \
            else fb[S][T].template read_tr<((T) & 1) * 8 + (KS) * 8192>((SN_) + offN[(T) >> 1]);                         \
        }                                                                                                                \
    } while (0)
    // the asm transpose reads of set S are named behind an explicit wait before their first use (the M side's b128 reads are the
    // compiler's own; one lgkmcnt(0) covers both -- every read of a k-step was issued at least 16 MFMA pairs earlier)
    auto wait_set = [&](Q16FragN<false> (&f)[NW]) __attribute__((always_inline)) {
        if constexpr (NW == 8)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo), "+v"(f[3].hi),
                         "+v"(f[4].lo), "+v"(f[4].hi), "+v"(f[5].lo), "+v"(f[5].hi), "+v"(f[6].lo), "+v"(f[6].hi), "+v"(f[7].lo), "+v"(f[7].hi));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].lo), "+v"(f[0].hi), "+v"(f[1].lo), "+v"(f[1].hi), "+v"(f[2].lo), "+v"(f[2].hi), "+v"(f[3].lo), "+v"(f[3].hi),
                         "+v"(f[4].lo), "+v"(f[4].hi), "+v"(f[5].lo), "+v"(f[5].hi));
    };
#define Q16_WAIT_SET(S) do { if constexpr (!B_KC) { wait_set(fb[S]); Q16_SB(); } } while (0)
    // MFMA j of a k-step: M tile j / NW, N tile j % NW (consecutive MFMAs keep the M-side operand and walk the N tiles)
#define Q16_MFMA(S, J, ZERO)                                                                                             \
    do {                                                                                                                 \
        constexpr int i_ = (J) / NW, t_ = (J) % NW;                                                                      \
        if (!(DBG & 1)) acc[i_][t_] = ECAMP_MFMA_16x16x32(fb[S][t_].get(), fa[S][i_], (ZERO) ? zero4 : acc[i_][t_]); \
    } while (0)
    // one k-step of 32: 8 NW MFMAs with ONE other action behind every second one: the NW + 8 fragment reads of the next k-step (set NS,
    // k-step NKS of the K tile at SM_ / SN_) in the order its MFMAs consume them, then the DMA pieces of part DP0, its bookkeeping (part
    // 3's moves the stream to the next K tile: it must precede the next part's pieces), part DP1 likewise.  Scheduling fences pin the
    // order (a clump of reads or DMA behind a few MFMAs outlasts their shadow).  Slot s (0 .. 4 NW - 1) sits behind MFMA 2 s + 1.
#define Q16_SLOT(S, NS, NKS, SM_, SN_, DP0, DP1, s_)                                                                     \
    do {                                                                                                                 \
        constexpr int s__ = (s_);                                                                                        \
        if (s__ < NW) Q16_RDB(NS, NKS, (s__ < NW ? s__ : 0), SN_);                                                       \
        else if (s__ < NW + 8) Q16_RDA(NS, NKS, (s__ >= NW && s__ < NW + 8 ? s__ - NW : 0), SM_);                        \
        else if (s__ < NW + 12) { Q16_ISSUE1(DP0, (s__ - NW - 8) & 3); }                                                 \
        else if (s__ == NW + 12) { Q16_ADVANCE(DP0); }                                                                   \
        else if (s__ < NW + 17) { Q16_ISSUE1(DP1, (s__ - NW - 13) & 3); }                                                \
        else if (s__ == NW + 17) { Q16_ADVANCE(DP1); }                                                                   \
    } while (0)
#define Q16_PAIR(S, ZERO, NS, NKS, SM_, SN_, DP0, DP1, s_)                                                               \
    do {                                                                                                                 \
        Q16_MFMA(S, 2 * (s_), ZERO); Q16_MFMA(S, 2 * (s_) + 1, ZERO); Q16_SB();                                          \
        Q16_SLOT(S, NS, NKS, SM_, SN_, DP0, DP1, s_); Q16_SB();                                                          \
    } while (0)
#define Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, s_) \
    do { Q16_PAIR(S, Z, NS, NKS, SM_, SN_, D0, D1, (s_)); Q16_PAIR(S, Z, NS, NKS, SM_, SN_, D0, D1, (s_) + 1); Q16_PAIR(S, Z, NS, NKS, SM_, SN_, D0, D1, (s_) + 2); Q16_PAIR(S, Z, NS, NKS, SM_, SN_, D0, D1, (s_) + 3); } while (0)
#define Q16_KSTEP(S, Z, NS, NKS, SM_, SN_, D0, D1)                                                                       \
    do {                                                                                                                 \
        Q16_SB();                                                                                                        \
        Q16_WAIT_SET(S);                                                                                                 \
        Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, 0);  Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, 4);                   \
        Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, 8);  Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, 12);                  \
        Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, 16); Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, 20);                  \
        if (NW == 8) { Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, (NW == 8 ? 24 : 0)); Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, (NW == 8 ? 28 : 0)); } \
    } while (0)
    int rA = 0, rB = 0;   // ring slots of A_0 / B_0 of the K tile being multiplied
    // K tile t.  Staged during k-step 0: A_0(t+2), B_0(t+2) -- into the slots of A_1(t-1), B_1(t-1), unread since the barrier of K tile t-1;
    // during k-step 1, behind this K tile's barrier: A_1(t+2), B_1(t+2) -- into the slots of A_0(t), B_0(t) (k-step 1's fragments were read
    // during k-step 0).  Every part is issued a whole K tile before the barrier that publishes it: inside the training step the operands
    // come from HBM behind a dependent kernel, and with half a K tile of lead (B_1(t+1) issued at the head of K tile t, the first version)
    // every launch paid 5-15 us more than the same call alone (profiles/r05_gemm_in_step_vs_lab.txt).  The barrier between the k-steps,
    // behind a counted wait that leaves the pieces of A_0 / B_0(t+2) in flight, publishes K tile t+1 and retires every read of K tile t.
#define Q16_RS(R_, ADD_) ((R_) + (ADD_) >= NSLOT ? (R_) + (ADD_) - NSLOT : (R_) + (ADD_))
#define Q16_KTILE(FIRST)                                                                                                 \
    do {                                                                                                                 \
        const unsigned char* sM = lds + Q16_RS(rA, wr) * Q8_HALF;                                                        \
        const unsigned char* sN = lds + (NSLOT + Q16_RS(rB, wc)) * Q8_HALF;                                              \
        const unsigned char* sMn = lds + Q16_RS(rA, 2 + wr) * Q8_HALF;                                                   \
        const unsigned char* sNn = lds + (NSLOT + Q16_RS(rB, 2 + wc)) * Q8_HALF;                                         \
        Q16_KSTEP(0, FIRST, 1, 1, sM, sN, 0, 1);                                                                         \
        if (NPB == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); Q16_SB();                                                     \
        __builtin_amdgcn_s_barrier(); Q16_SB();                                                                          \
        Q16_KSTEP(1, false, 0, 0, sMn, sNn, 2, 3);                                                                       \
        rA = Q16_RS(rA, 2); rB = Q16_RS(rB, 2);                                                                          \
    } while (0)

    // prologue: K tiles 0 and 1 issued, K tile 0 landed and published, its first k-step's fragments read
    q_cv(false);
    if (qv < total) Q16_ITEM();
    Q16_ISSUE_ALL(0); Q16_ADVANCE(0); Q16_ISSUE_ALL(1); Q16_ADVANCE(1); Q16_ISSUE_ALL(2); Q16_ADVANCE(2); Q16_ISSUE_ALL(3); Q16_ADVANCE(3);
    Q16_ISSUE_ALL(0); Q16_ADVANCE(0); Q16_ISSUE_ALL(1); Q16_ADVANCE(1); Q16_ISSUE_ALL(2); Q16_ADVANCE(2); Q16_ISSUE_ALL(3); Q16_ADVANCE(3);
    if (NPB == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");   // in flight: K tile 1
    __builtin_amdgcn_s_barrier();
    {
        const unsigned char* sM = lds + wr * Q8_HALF;
        const unsigned char* sN = lds + (NSLOT + wc) * Q8_HALF;
        Q16_RDB(0, 0, 0, sN); Q16_RDB(0, 0, 1, sN); Q16_RDB(0, 0, 2, sN); Q16_RDB(0, 0, 3, sN); Q16_RDB(0, 0, 4, sN); Q16_RDB(0, 0, 5, sN);
        if (NW == 8) { Q16_RDB(0, 0, (NW == 8 ? 6 : 0), sN); Q16_RDB(0, 0, (NW == 8 ? 7 : 0), sN); }
        Q16_RDA(0, 0, 0, sM); Q16_RDA(0, 0, 1, sM); Q16_RDA(0, 0, 2, sM); Q16_RDA(0, 0, 3, sM);
        Q16_RDA(0, 0, 4, sM); Q16_RDA(0, 0, 5, sM); Q16_RDA(0, 0, 6, sM); Q16_RDA(0, 0, 7, sM);
    }

    for (int cv = it_beg; cv < total; cv += G) {
        int cm0, cn0;
        decode(cv, cm0, cn0);
        bias_request(cn0);
        Q16_KTILE(true);
        bias_landed();
#pragma unroll 1
        for (int t = 1; t < nt; ++t) Q16_KTILE(false);
        Q16_WAIT_SET(0);   // the next tile's first fragments (already requested): named before the epilogue's code moves registers around
        Q16_STORE_TILE(cm0, cn0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the exhausted stream's zero-length loads still write their (zero) pieces into this workgroup's LDS
#undef Q16_RS
#undef Q16_ITEM
#undef Q16_ISSUE1
#undef Q16_ISSUE_ALL
#undef Q16_ADVANCE
#undef Q16_STORE_TILE
#undef Q16_SB
#undef Q16_RDA
#undef Q16_RDB
#undef Q16_WAIT_SET
#undef Q16_MFMA
#undef Q16_SLOT
#undef Q16_PAIR
#undef Q16_PAIR4
#undef Q16_KSTEP
#undef Q16_KTILE
}

template <int EPI, int NW, bool B_KC, int DBG = 0>
__global__ __launch_bounds__(256) void gemm_bf16_q16_kernel(GemmArgs g) {
    q16_body<EPI, NW, B_KC, DBG>(g);
}
