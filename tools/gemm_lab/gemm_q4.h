// "Q4": the persistent 256x256x64 bf16 GEMM as FOUR waves of 128x128, one per SIMD (round 4).  LAB ONLY -- not part of the product
// library: it ties the eight-wave kernel (profiles/r04_q4_and_power.txt) and the measurements it made possible say why.
//
// Why it was built: the counter comparison with the vendor library's kernel (profiles/r04_vendor_vs_q8_pmc.txt) says that kernel runs
// one wave per SIMD: a third less LDS traffic (a 128 x 128 wave tile reads 32 fragments per K tile where two 128 x 64 tiles read 48),
// few barriers, few scalar instructions -- and every schedule variant of the eight-wave kernel lands on the same time per K tile.
//
// Structure.  Same LDS images, rings (5 half-tile slots per operand), DMA pieces, descriptors and tile order as gemm_q8.h: half-tile A_r
// holds the 128 rows of wave row r, B_c the 128 columns of wave column c.  A K tile is ONE instruction stream per wave: for each of its
// four k-steps of 16, sixteen MFMAs (4 x 4 tiles of 32 x 32) with ONE other instruction in the shadow of each: the eight fragment reads
// of the NEXT k-step (across the K tile boundary too: k-step 3 reads k-step 0 of the next K tile), this wave's four DMA pieces of one
// half-tile part, the stream bookkeeping (scheduling fences pin the order); the fragments are double buffered in registers.  One
// counted DMA wait and ONE barrier per K tile (between k-steps 2 and 3) publish K tile t+1.  0 scratch, 117-125 VGPRs + 256 AGPRs.
//
// What it showed (profiles/r04_q4_and_power.txt): bit-exact at the first run; the plain forward form within -3 .. +3 % of the eight-wave
// kernel on every shape, the same time per K tile (slope of a K sweep) to 1 %; the GELU epilogue 8-15 % slower (one wave per SIMD has
// nobody to overlap its 256 accumulators' arithmetic with).  Issuing all DMA parts right behind the barrier (three k-steps of lead
// instead of two, EARLY) is 5-9 % SLOWER: the loop does not wait for DMA latency.  With all-zero operands the same instruction
// stream needs 1.12 us per K tile instead of 1.64 (eight-wave kernel: 1.12 / 1.64): the loop is bound by the socket's power limit
// on the matrix pipes + operand movement, not by its structure.
#pragma once
#include "../../ecamp_amd/csrc/gemm_q8.h"

template <bool KC>
__device__ __forceinline__ unsigned q4_voff(int pi, int lane, long ld) {   // per-lane source offset of piece `pi` (0..15) of a half-tile
    if (KC) {
        const int row = pi * 8 + (lane >> 3);
        const int kc = (lane & 7) ^ ((row >> 1) & 7);
        return (unsigned)((long)row * ld * 2 + kc * 16);
    } else {
        const int kr = pi * 4 + (lane >> 4);
        const int oc = (lane & 15) ^ ((kr & 3) << 2);
        return (unsigned)(((long)kr * ld + oc * 8) * 2);
    }
}

// EPI: 0 bf16 C = alpha*acc (+bias)   1 ... + save pre-activation + exact GELU   2 ... + residual
// EARLY: the DMA lead.  false: one half-tile part per k-step (B_1(t+1), A_0(t+2), B_0(t+2) | barrier | A_1(t+2)): the youngest part the
// barrier waits for was issued two k-steps earlier.  true: all four parts that fit the ring are issued in k-step 3, right behind the
// barrier that freed their slots (A_1, B_1 of t+2 and A_0, B_0 of t+3): the youngest part a barrier waits for is three k-steps old.
// (A variant that issued B_1(t+2) in front of the epilogue's stores and let the stores stay in flight across the next K tile's
// counted wait -- gfx9 counts loads and stores in one in-order counter -- measured 1-3 % slower: removed.)
template <int EPI, bool EARLY, int DBG = 0>   // DBG (lab only): 1 no MFMA, 2 no DMA, 4 no fragment reads
__global__ __launch_bounds__(256) void gemm_bf16_q4_kernel(GemmArgs g) {
    constexpr int NSLOT = 5;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // A ring (5 x 16 KB) | B ring (5 x 16 KB); the ONLY LDS object
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int total = g.nbm * g.nbn * g.nsplit, G = (int)gridDim.x, it_beg = (int)blockIdx.x;
    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(g.A);
    const unsigned char* Bb = reinterpret_cast<const unsigned char*>(g.B);
    const int l31 = lane & 31, lh = lane >> 5;
    const int ncol = (l31 & 3) | (((l31 >> 3) & 1) << 2) | (((l31 >> 2) & 1) << 3) | ((l31 >> 4) << 4);
    unsigned offM[4], offN[4];   // per k-step; 32-row / 32-column block i at + i * 4096
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        offM[ks] = (unsigned)(l31 * 128 + (((2 * ks + lh) ^ ((l31 >> 1) & 7)) << 4));
        offN[ks] = (unsigned)(ncol * 128 + (((2 * ks + lh) ^ ((ncol >> 1) & 7)) << 4));
    }
    f32x16 acc[4][4];

    // ---- the operand stream (lean): this wave owns pieces 4 * wave .. 4 * wave + 3 of every half-tile
    const unsigned char *qa = Ab, *qb = Bb;
    int qa_rec = 0, qb_rec = 0, q_krem = 1 << 30, qv = it_beg;
    bool q_tail = false;
    unsigned cvA[8], cvB[8];   // [half * 4 + j]
    int dA = wave * 4096, dB = NSLOT * Q8_HALF + wave * 4096;
    auto q_cv = [&](bool tail) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pi = wave * 4 + j;
            cvA[j] = q4_voff<true>(pi, lane, g.lda); cvA[4 + j] = cvA[j] + (unsigned)(g.lda * 256);
            cvB[j] = q4_voff<true>(pi, lane, g.ldb); cvB[4 + j] = cvB[j] + (unsigned)(g.ldb * 256);
            if (tail) {
                const int row = pi * 8 + (lane >> 3);
                const int kc0 = (lane & 7) ^ ((row >> 1) & 7);
                if (kc0 * 8 >= q_krem) { cvA[j] = cvA[4 + j] = 0xFFFFFF00u; cvB[j] = cvB[4 + j] = 0xFFFFFF00u; }
            }
        }
    };
#define Q4_ITEM()                                                                                                        \
    do {                                                                                                                 \
        const Q8Item n_ = q8_decode(g, qv, total);                                                                       \
        q_krem = n_.kend - n_.kbeg;                                                                                      \
        qa = Ab + ((long)n_.m0 * g.lda + n_.kbeg) * 2; qa_rec = (int)((((long)(g.M - n_.m0)) * g.lda - n_.kbeg) * 2);    \
        qb = Bb + ((long)n_.n0 * g.ldb + n_.kbeg) * 2; qb_rec = (int)((((long)(g.N - n_.n0)) * g.ldb - n_.kbeg) * 2);    \
        qa_rec = max(qa_rec, 0); qb_rec = max(qb_rec, 0);                                                                \
    } while (0)
    typedef void __attribute__((address_space(3))) lds_void_;
    // piece J (0..3) of part PART (0: A half 0, 1: B half 0, 2: A half 1, 3: B half 1)
#define Q4_ISSUE1(PART, J)                                                                                               \
    if ((PART) >= 0 && !(DBG & 2)) do {                                                                                                \
        constexpr bool isA_ = (((PART) & 1) == 0);                                                                       \
        constexpr int h_ = (PART) >> 1;                                                                                  \
        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)(isA_ ? qa : qb), 0, isA_ ? qa_rec : qb_rec, 0x00020000); \
        unsigned char* d_ = lds + (isA_ ? dA : dB) + (J) * 1024;                                                         \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_void_*)d_, 16, (int)(isA_ ? cvA[4 * h_ + (J)] : cvB[4 * h_ + (J)]), 0, 0, 0); \
    } while (0)
#define Q4_ADVANCE(PART)                                                                                                 \
    if ((PART) >= 0) do {                                                                                                \
        if (((PART) & 1) == 0) { dA += Q8_HALF; if (dA >= NSLOT * Q8_HALF) dA -= NSLOT * Q8_HALF; }                      \
        else                   { dB += Q8_HALF; if (dB >= 2 * NSLOT * Q8_HALF) dB -= NSLOT * Q8_HALF; }                  \
        if ((PART) == 3) {                                                                                               \
            q_krem -= 64;                                                                                                \
            qa += 128; qb += 128; qa_rec = max(qa_rec - 128, 0); qb_rec = max(qb_rec - 128, 0);                          \
            if (q_krem <= 0) {                                                                                           \
                qv += G;                                                                                                 \
                if (qv < total) Q4_ITEM(); else { qa_rec = 0; qb_rec = 0; q_krem = 1 << 30; }                            \
            }                                                                                                            \
            const bool tl_ = q_krem < 64;                                                                                \
            if (tl_ != q_tail) { q_tail = tl_; q_cv(tl_); }                                                              \
        }                                                                                                                \
    } while (0)
#define Q4_ISSUE_ALL(PART) do { Q4_ISSUE1(PART, 0); Q4_ISSUE1(PART, 1); Q4_ISSUE1(PART, 2); Q4_ISSUE1(PART, 3); } while (0)

    // ---- epilogue (the accumulators are only read: the next tile's first k-step starts from C = 0)
    const long ldo = g.ldc;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)(unsigned)((long)g.M * ldo * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(EPI == 1 ? g.pre_out : g.C, 0, (int)(unsigned)((long)g.M * g.ldp * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 2 && g.residual ? g.residual : g.C), 0, (int)(unsigned)((long)g.M * g.ldr * 2), 0x00020000);
    const unsigned lane_o = (unsigned)((l31 * ldo + 8 * lh) * 2), lane_p = (unsigned)((l31 * g.ldp + 8 * lh) * 2), lane_r = (unsigned)((l31 * g.ldr + 8 * lh) * 2);
    auto store_block = [&](int tm0, int tn0, auto tm_c, auto tn_c) __attribute__((always_inline)) {   // 32 x 32 block (TM, TN) of the wave's tile
        constexpr int TM = decltype(tm_c)::value, TN = decltype(tn_c)::value;
        typedef const float __attribute__((address_space(4))) cfloat4;
        const int mb = tm0 + wr * 128 + TM * 32, nb = tn0 + wc * 128 + TN * 32;
        float al = g.alpha;
        if (g.alpha_dev) { float ad = *(cfloat4*)g.alpha_dev; asm volatile("" : "+s"(ad)); al *= ad; }
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            const bool oob = nb + 16 * gq + 8 * lh >= g.N;
            const long row = mb, col = nb + 16 * gq;
            const unsigned uo = oob ? 0x80000000u : (unsigned)((row * ldo + col) * 2) + lane_o;
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {   // an explicit AGPR read: left to itself hipcc copies all 256 accumulators to VGPRs behind the K loop and spills
                float x;
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[TM][TN][8 * gq + r]));
                v[r] = x * al;
            }
            if (g.bias) {
                cfloat4* b0 = (cfloat4*)(g.bias + min(nb + 16 * gq, g.N - 8));
                cfloat4* b1 = (cfloat4*)(g.bias + min(nb + 16 * gq + 8, g.N - 8));
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float x0 = b0[r], x1 = b1[r];
                    asm volatile("" : "+s"(x0), "+s"(x1));
                    v[r] += lh ? x1 : x0;
                }
            }
            if (EPI == 1) {
                const unsigned up = oob ? 0x80000000u : (unsigned)((row * g.ldp + col) * 2) + lane_p;
                __builtin_amdgcn_raw_buffer_store_b128(q8_pack8(v), rP, up, 0, 0);
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = gelu_t<bf16_t>(rnd<bf16_t>(v[r]));
            }
            if (EPI == 2) {
                const unsigned ur = oob ? 0x80000000u : (unsigned)((row * g.ldr + col) * 2) + lane_r;
                const q8_u32x4 qr = __builtin_amdgcn_raw_buffer_load_b128(rR, ur, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[2 * r] += __uint_as_float(qr[r] << 16); v[2 * r + 1] += __uint_as_float(qr[r] & 0xffff0000u); }
            }
            __builtin_amdgcn_raw_buffer_store_b128(q8_pack8(v), rC, uo, 0, 0);
        }
    };
    // (a scheduling fence behind every block: left alone, hipcc hoists the accumulator reads of many blocks ahead and spills arch VGPRs)
#define Q4_STORE_ROW(TM0, TN0, TM)                                                                                       \
    do {                                                                                                                 \
        store_block(TM0, TN0, std::integral_constant<int, TM>(), std::integral_constant<int, 0>()); __builtin_amdgcn_sched_barrier(0); \
        store_block(TM0, TN0, std::integral_constant<int, TM>(), std::integral_constant<int, 1>()); __builtin_amdgcn_sched_barrier(0); \
        store_block(TM0, TN0, std::integral_constant<int, TM>(), std::integral_constant<int, 2>()); __builtin_amdgcn_sched_barrier(0); \
        store_block(TM0, TN0, std::integral_constant<int, TM>(), std::integral_constant<int, 3>()); __builtin_amdgcn_sched_barrier(0); \
    } while (0)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ---- fragments: two register sets (k-step parity); the set of k-step ks + 1 is read during the MFMAs of k-step ks -- ACROSS the K tile
    // boundary too: k-step 3 reads k-step 0 of the next K tile, which the barrier in front of it has published
    hw_bf16x8 fa[2][4], fb[2][4];
#define Q4_SB() __builtin_amdgcn_sched_barrier(0)
#define Q4_RDA(S, KS, I, SM_) if (!(DBG & 4)) fa[S][I] = *reinterpret_cast<const hw_bf16x8*>((SM_) + offM[KS] + (I) * 4096)
#define Q4_RDB(S, KS, I, SN_) if (!(DBG & 4)) fb[S][I] = *reinterpret_cast<const hw_bf16x8*>((SN_) + offN[KS] + (I) * 4096)
#define Q4_MFMA(S, I, ZERO) if (!(DBG & 1)) acc[(I) & 3][(I) >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[S][(I) >> 2], fa[S][(I) & 3], (ZERO) ? zero16 : acc[(I) & 3][(I) >> 2], 0, 0, 0)
    // one k-step: sixteen MFMAs, ONE other action in the shadow of each (a wave issues in order: a clump of reads and DMA behind four MFMAs
    // outlasts the fourth one's 32 cycles and idles the matrix pipe).  The eight fragment reads of the next k-step (set NS, k-step NKS of
    // the K tile at SM_ / SN_) come in the order its MFMAs consume them; four DMA pieces of part DPART; the stream bookkeeping last.
#define Q4_KSTEP(S, ZERO, NS, NKS, SM_, SN_, DPART)                                                                      \
    do {                                                                                                                 \
        Q4_SB();                                                                                                         \
        Q4_MFMA(S, 0, ZERO);  Q4_SB(); Q4_RDB(NS, NKS, 0, SN_); Q4_SB();                                                 \
        Q4_MFMA(S, 1, ZERO);  Q4_SB(); Q4_RDA(NS, NKS, 0, SM_); Q4_SB();                                                 \
        Q4_MFMA(S, 2, ZERO);  Q4_SB(); Q4_RDA(NS, NKS, 1, SM_); Q4_SB();                                                 \
        Q4_MFMA(S, 3, ZERO);  Q4_SB(); Q4_RDA(NS, NKS, 2, SM_); Q4_SB();                                                 \
        Q4_MFMA(S, 4, ZERO);  Q4_SB(); Q4_RDA(NS, NKS, 3, SM_); Q4_SB();                                                 \
        Q4_MFMA(S, 5, ZERO);  Q4_SB(); Q4_RDB(NS, NKS, 1, SN_); Q4_SB();                                                 \
        Q4_MFMA(S, 6, ZERO);  Q4_SB(); Q4_RDB(NS, NKS, 2, SN_); Q4_SB();                                                 \
        Q4_MFMA(S, 7, ZERO);  Q4_SB(); Q4_RDB(NS, NKS, 3, SN_); Q4_SB();                                                 \
        Q4_MFMA(S, 8, ZERO);  Q4_SB(); Q4_ISSUE1(DPART, 0); Q4_SB();                                                     \
        Q4_MFMA(S, 9, ZERO);  Q4_SB();                                                                                   \
        Q4_MFMA(S, 10, ZERO); Q4_SB(); Q4_ISSUE1(DPART, 1); Q4_SB();                                                     \
        Q4_MFMA(S, 11, ZERO); Q4_SB();                                                                                   \
        Q4_MFMA(S, 12, ZERO); Q4_SB(); Q4_ISSUE1(DPART, 2); Q4_SB();                                                     \
        Q4_MFMA(S, 13, ZERO); Q4_SB();                                                                                   \
        Q4_MFMA(S, 14, ZERO); Q4_SB(); Q4_ISSUE1(DPART, 3); Q4_SB();                                                     \
        Q4_MFMA(S, 15, ZERO); Q4_SB(); Q4_ADVANCE(DPART); Q4_SB();                                                       \
    } while (0)
    // k-step 3 with EARLY: the next K tile's first fragments and sixteen DMA pieces (parts 2, 3, 0, 1 in stream order), one per MFMA
#define Q4_KSTEP3E(S, NS, SM_, SN_)                                                                                      \
    do {                                                                                                                 \
        Q4_SB();                                                                                                         \
        Q4_MFMA(S, 0, false);  Q4_SB(); Q4_RDB(NS, 0, 0, SN_); Q4_ISSUE1(2, 0); Q4_SB();                                 \
        Q4_MFMA(S, 1, false);  Q4_SB(); Q4_RDA(NS, 0, 0, SM_); Q4_ISSUE1(2, 1); Q4_SB();                                 \
        Q4_MFMA(S, 2, false);  Q4_SB(); Q4_RDA(NS, 0, 1, SM_); Q4_ISSUE1(2, 2); Q4_SB();                                 \
        Q4_MFMA(S, 3, false);  Q4_SB(); Q4_RDA(NS, 0, 2, SM_); Q4_ISSUE1(2, 3); Q4_ADVANCE(2); Q4_SB();                  \
        Q4_MFMA(S, 4, false);  Q4_SB(); Q4_RDA(NS, 0, 3, SM_); Q4_ISSUE1(3, 0); Q4_SB();                                 \
        Q4_MFMA(S, 5, false);  Q4_SB(); Q4_RDB(NS, 0, 1, SN_); Q4_ISSUE1(3, 1); Q4_SB();                                 \
        Q4_MFMA(S, 6, false);  Q4_SB(); Q4_RDB(NS, 0, 2, SN_); Q4_ISSUE1(3, 2); Q4_SB();                                 \
        Q4_MFMA(S, 7, false);  Q4_SB(); Q4_RDB(NS, 0, 3, SN_); Q4_ISSUE1(3, 3); Q4_ADVANCE(3); Q4_SB();                  \
        Q4_MFMA(S, 8, false);  Q4_SB(); Q4_ISSUE1(0, 0); Q4_SB();                                                        \
        Q4_MFMA(S, 9, false);  Q4_SB(); Q4_ISSUE1(0, 1); Q4_SB();                                                        \
        Q4_MFMA(S, 10, false); Q4_SB(); Q4_ISSUE1(0, 2); Q4_SB();                                                        \
        Q4_MFMA(S, 11, false); Q4_SB(); Q4_ISSUE1(0, 3); Q4_ADVANCE(0); Q4_SB();                                         \
        Q4_MFMA(S, 12, false); Q4_SB(); Q4_ISSUE1(1, 0); Q4_SB();                                                        \
        Q4_MFMA(S, 13, false); Q4_SB(); Q4_ISSUE1(1, 1); Q4_SB();                                                        \
        Q4_MFMA(S, 14, false); Q4_SB(); Q4_ISSUE1(1, 2); Q4_SB();                                                        \
        Q4_MFMA(S, 15, false); Q4_SB(); Q4_ISSUE1(1, 3); Q4_ADVANCE(1); Q4_SB();                                         \
    } while (0)
    int rA = 0, rB = 0;   // ring slots of A_0 / B_0 of the K tile being multiplied
    // a K tile t.  Staged during its k-steps 0..3: B_1(t+1), A_0(t+2), B_0(t+2), A_1(t+2) -- the last one into the slot of A_0(t), which is
    // free once every wave is past the barrier between k-steps 2 and 3 (all of tile t's fragments have been read by then).  That barrier,
    // behind a counted wait that leaves the eight pieces of A_0 / B_0 (t+2) in flight, publishes K tile t+1.
#define Q4_SLOT(R_, ADD_) ((R_) + (ADD_) >= NSLOT ? (R_) + (ADD_) - NSLOT : (R_) + (ADD_))
#define Q4_KTILE(FIRST)                                                                                                  \
    do {                                                                                                                 \
        const unsigned char* sM = lds + Q4_SLOT(rA, wr) * Q8_HALF;                                                       \
        const unsigned char* sN = lds + (NSLOT + Q4_SLOT(rB, wc)) * Q8_HALF;                                             \
        const unsigned char* sMn = lds + Q4_SLOT(rA, 2 + wr) * Q8_HALF;                                                  \
        const unsigned char* sNn = lds + (NSLOT + Q4_SLOT(rB, 2 + wc)) * Q8_HALF;                                        \
        Q4_KSTEP(0, FIRST, 1, 1, sM, sN, (EARLY ? -1 : 3));                                                              \
        Q4_KSTEP(1, false, 0, 2, sM, sN, (EARLY ? -1 : 0));                                                              \
        Q4_KSTEP(0, false, 1, 3, sM, sN, (EARLY ? -1 : 1));                                                              \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); Q4_SB();                                                      \
        __builtin_amdgcn_s_barrier(); Q4_SB();                                                                           \
        if (EARLY) Q4_KSTEP3E(1, 0, sMn, sNn); else                                                                      \
        Q4_KSTEP(1, false, 0, 0, sMn, sNn, 2);                                                                           \
        rA = Q4_SLOT(rA, 2); rB = Q4_SLOT(rB, 2);                                                                        \
    } while (0)

    // prologue: K tile 0 and A_0 / B_0 / A_1 of K tile 1 issued, K tile 0 landed and published, its first k-step's fragments read
    q_cv(false);
    if (qv < total) Q4_ITEM();
    Q4_ISSUE_ALL(0); Q4_ADVANCE(0); Q4_ISSUE_ALL(1); Q4_ADVANCE(1); Q4_ISSUE_ALL(2); Q4_ADVANCE(2); Q4_ISSUE_ALL(3); Q4_ADVANCE(3);
    Q4_ISSUE_ALL(0); Q4_ADVANCE(0); Q4_ISSUE_ALL(1); Q4_ADVANCE(1); Q4_ISSUE_ALL(2); Q4_ADVANCE(2);
    if (EARLY) {   // ... and B_1 of K tile 1, A_0 / B_0 of K tile 2: the whole ring
        Q4_ISSUE_ALL(3); Q4_ADVANCE(3); Q4_ISSUE_ALL(0); Q4_ADVANCE(0); Q4_ISSUE_ALL(1); Q4_ADVANCE(1);
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
        const unsigned char* sM = lds + wr * Q8_HALF;
        const unsigned char* sN = lds + (NSLOT + wc) * Q8_HALF;
        Q4_RDB(0, 0, 0, sN); Q4_RDA(0, 0, 0, sM); Q4_RDA(0, 0, 1, sM); Q4_RDA(0, 0, 2, sM); Q4_RDA(0, 0, 3, sM);
        Q4_RDB(0, 0, 1, sN); Q4_RDB(0, 0, 2, sN); Q4_RDB(0, 0, 3, sN);
    }

    // (the epilogue sits at the END of its tile's iteration, in front of the next tile's first K tile whose operands are already in LDS
    // and whose first fragments are already in registers: with the accumulators live across the tile loop's back edge hipcc shuffles
    // ~250 of them through VGPRs at the loop head)
    for (int cv = it_beg; cv < total; cv += G) {
        const Q8Item cit = q8_decode(g, cv, total);
        Q4_KTILE(true);
#pragma unroll 1
        for (int t = 1; t < cit.nt; ++t) Q4_KTILE(false);
        Q4_STORE_ROW(cit.m0, cit.n0, 0); Q4_STORE_ROW(cit.m0, cit.n0, 1); Q4_STORE_ROW(cit.m0, cit.n0, 2); Q4_STORE_ROW(cit.m0, cit.n0, 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the exhausted stream's zero-length loads still write their (zero) pieces into this workgroup's LDS
#undef Q4_SLOT
#undef Q4_ITEM
#undef Q4_ISSUE1
#undef Q4_ISSUE_ALL
#undef Q4_ADVANCE
#undef Q4_STORE_ROW
#undef Q4_SB
#undef Q4_RDA
#undef Q4_RDB
#undef Q4_MFMA
#undef Q4_KSTEP
#undef Q4_KSTEP3E
#undef Q4_KTILE
}
