// "Q16": the persistent bf16 GEMM as FOUR waves (2 x 2, one per SIMD) on v_mfma_f32_16x16x32_bf16 (round 5).  Forward form (both operands
// contraction-contiguous: y = x w^T, the nn.Linear forward of model_ecamp.py:233-234,254-255 / bert_modeling.py:131).
//
// Why: (1) VERDICT r4 item 1 -- the vendor library's kernel for this form (Custom_Cijk_..._MT256x256x64_MI16x16x1, read as a yardstick in
// profiles/r05_vendor_loop_isa.txt) is four waves of 128 x 128 on the 16 x 16 x 32 MFMA: 128 MFMAs, 32 ds_read_b128, 16 LDS-DMA pieces and 3
// barriers per wave and K tile, ONE other instruction in the shadow of each MFMA pair, fragments of k-step s+1 read under the MFMAs of k-step s
// (across the K tile boundary too).  gemm_q4.h has that shape on the 32 x 32 x 16 MFMA and ties the eight-wave kernel; this is the same
// stream on the vendor's MFMA shape.  (2) The 16-wide MFMA makes a 192-column tile natural (wave tile 128 x 96 = 8 x 6 MFMA tiles): the
// model's 768-wide outputs (encoder proj / fc2, every dx of width 768) are 150 tiles of 256 x 256 on 256 CUs -- 41 % of the chip idle --
// and 200 tiles of 256 x 192.
//
// Layout.  LDS images, rings (5 half-tile slots of 16 KB per operand), DMA pieces (8 rows x 128 B per wave instruction) and the
// descriptor stream are gemm_q8.h's / gemm_q4.h's.  Fragments: one ds_read_b128 = 16 rows x 32 k (lane: row l & 15, 16-B chunk
// 4 * ks + (l >> 4) of the row's 128 B).  The M side reads rows 16 i + (l & 15) under the XOR key (row >> 1) & 7 of gemm_q8.h
// (conflict-free for the 16-lane groups of ds_read_b128: checked by enumeration, tools/probes/lds_swizzle_search.py conventions).
// The N side is the MFMA's A operand, so that a lane's four accumulator registers are four consecutive output COLUMNS; its fragment
// row i of MFMA tile t is tile column 32 (t >> 1) + 8 (i >> 2) + 4 (t & 1) + (i & 3): the two MFMA tiles of a pair give a lane eight
// consecutive columns = one 16-B store, and a store instruction writes 64 contiguous bytes of each of 16 rows.  That row set
// {0-3, 8-11, 16-19, 24-27} collides under gemm_q8.h's key; the N-side half-tiles are therefore staged under the key
// ((row >> 1) & 1) | (((row >> 3) & 3) << 1) (conflict-free for it, same enumeration).
#pragma once
#include "../../ecamp_amd/csrc/gemm_q8.h"

typedef __attribute__((ext_vector_type(4))) float q16_f32x4;

__device__ __forceinline__ int q16_key_m(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int q16_key_n(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }

// EPI: 0 bf16 C = alpha*acc (+bias)   2 ... + residual.   NW: MFMA tiles per wave along N (8: 256-column tile, 6: 192-column tile)
template <int EPI, int NW, int DBG = 0>   // DBG (lab only): 1 no MFMA, 2 no DMA, 4 no fragment reads
__global__ __launch_bounds__(256) void gemm_bf16_q16_kernel(GemmArgs g) {
    constexpr int NSLOT = 5;
    constexpr int TN = 32 * NW;            // tile columns
    constexpr int NPB = NW / 2;            // DMA pieces of a B half-tile per wave (a half-tile = 16 NW rows = 2 NW pieces over 4 waves)
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

    // per-lane fragment offsets inside a half-tile, per k-step of 32 (ks = 0, 1)
    const int nrow = 8 * (l15 >> 2) + (l15 & 3);
    unsigned offM[2], offN[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        offM[ks] = (unsigned)(l15 * 128 + (((4 * ks + lq) ^ q16_key_m(l15)) << 4));
        offN[ks] = (unsigned)(nrow * 128 + (((4 * ks + lq) ^ q16_key_n(nrow)) << 4));
    }
    q16_f32x4 acc[8][NW];

    // ---- the operand stream (lean, as gemm_q4.h): wave w owns pieces 4 w .. 4 w + 3 of an A half-tile and NPB pieces of a B half-tile
    const unsigned char *qa = Ab, *qb = Bb;
    int qa_rec = 0, qb_rec = 0, q_krem = 1 << 30, qv = it_beg;
    bool q_tail = false;
    unsigned cvA[8], cvB[2 * NPB];   // [half * pieces + j]
    int dA = wave * 4096, dB = NSLOT * Q8_HALF + wave * NPB * 1024;
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
            const int row = (wave * NPB + j) * 8 + (lane >> 3);
            const int kc = (lane & 7) ^ q16_key_n(row);
            cvB[j] = (unsigned)((long)row * g.ldb * 2 + kc * 16); cvB[NPB + j] = cvB[j] + (unsigned)(g.ldb * 32 * NW);   // second half-tile: 16 NW rows on
            if (tail && kc * 8 >= q_krem) cvB[j] = cvB[NPB + j] = 0xFFFFFF00u;
        }
    };
#define Q16_ITEM()                                                                                                       \
    do {                                                                                                                 \
        int m0_, n0_;                                                                                                    \
        decode(qv, m0_, n0_);                                                                                            \
        q_krem = g.K;                                                                                                    \
        qa = Ab + ((long)m0_ * g.lda) * 2; qa_rec = (int)((((long)(g.M - m0_)) * g.lda) * 2);                            \
        qb = Bb + ((long)n0_ * g.ldb) * 2; qb_rec = (int)((((long)(g.N - n0_)) * g.ldb) * 2);                            \
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
            qa += 128; qb += 128; qa_rec = max(qa_rec - 128, 0); qb_rec = max(qb_rec - 128, 0);                          \
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
    auto store_pair = [&](int tm0, int tn0, auto i_c, auto p_c) __attribute__((always_inline)) {
        constexpr int I = decltype(i_c)::value, P = decltype(p_c)::value;
        typedef const float __attribute__((address_space(4))) cfloat4;
        const int mb = tm0 + wr * 128 + I * 16, nb = tn0 + wc * (16 * NW) + P * 32;
        float al = g.alpha;
        if (g.alpha_dev) { float ad = *(cfloat4*)g.alpha_dev; asm volatile("" : "+s"(ad)); al *= ad; }
        const bool oob = nb + 8 * lq >= g.N;
        const unsigned uo = oob ? 0x80000000u : (unsigned)(((long)mb * ldo + nb) * 2) + lane_o;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // explicit AGPR reads (see gemm_q4.h: left to itself hipcc copies every accumulator to VGPRs behind the K loop)
            float x0, x1;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x0) : "a"(acc[I][2 * P][r]));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x1) : "a"(acc[I][2 * P + 1][r]));
            v[r] = x0 * al; v[4 + r] = x1 * al;
        }
        if (g.bias) {
            // 8 consecutive columns nb + 8 lq ..: four possible groups per pair, selected per lane
            const int cb = min(nb, g.N - 32);
            cfloat4* b = (cfloat4*)(g.bias + cb);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float x0 = b[r], x1 = b[8 + r], x2 = b[16 + r], x3 = b[24 + r];
                asm volatile("" : "+s"(x0), "+s"(x1), "+s"(x2), "+s"(x3));
                v[r] += lq == 0 ? x0 : lq == 1 ? x1 : lq == 2 ? x2 : x3;
            }
        }
        if (EPI == 2) {
            const unsigned ur = oob ? 0x80000000u : (unsigned)(((long)mb * g.ldr + nb) * 2) + lane_r;
            const q8_u32x4 qr = __builtin_amdgcn_raw_buffer_load_b128(rR, ur, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[2 * r] += __uint_as_float(qr[r] << 16); v[2 * r + 1] += __uint_as_float(qr[r] & 0xffff0000u); }
        }
        __builtin_amdgcn_raw_buffer_store_b128(q8_pack8(v), rC, uo, 0, 0);
    };
#define Q16_SB() __builtin_amdgcn_sched_barrier(0)
#define Q16_STORE_ROW(TM0, TN0, I)                                                                                       \
    do {                                                                                                                 \
        store_pair(TM0, TN0, std::integral_constant<int, I>(), std::integral_constant<int, 0>()); Q16_SB();              \
        store_pair(TM0, TN0, std::integral_constant<int, I>(), std::integral_constant<int, 1>()); Q16_SB();              \
        store_pair(TM0, TN0, std::integral_constant<int, I>(), std::integral_constant<int, 2>()); Q16_SB();              \
        if (NW == 8) { store_pair(TM0, TN0, std::integral_constant<int, I>(), std::integral_constant<int, (NW == 8 ? 3 : 0)>()); Q16_SB(); } \
    } while (0)
    const q16_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- fragments: two register sets (k-step parity); the set of k-step s + 1 is read under the MFMAs of k-step s -- across the K tile
    // boundary too (k-step 1 reads k-step 0 of the next K tile, which the barrier in front of it has published)
    hw_bf16x8 fa[2][8], fb[2][NW];
#define Q16_RDA(S, KS, I, SM_) do { if (!(DBG & 4)) fa[S][I] = *reinterpret_cast<const hw_bf16x8*>((SM_) + offM[KS] + (I) * 2048); } while (0)
#define Q16_RDB(S, KS, T, SN_) do { if (!(DBG & 4)) fb[S][T] = *reinterpret_cast<const hw_bf16x8*>((SN_) + offN[KS] + ((T) & 1) * 512 + ((T) >> 1) * 4096); } while (0)
    // MFMA j of a k-step: M tile j / NW, N tile j % NW (consecutive MFMAs keep the M-side operand and walk the N tiles)
#define Q16_MFMA(S, J, ZERO)                                                                                             \
    do {                                                                                                                 \
        constexpr int i_ = (J) / NW, t_ = (J) % NW;                                                                      \
        if (!(DBG & 1)) acc[i_][t_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[S][t_], fa[S][i_], (ZERO) ? zero4 : acc[i_][t_], 0, 0, 0); \
    } while (0)
    // one k-step of 32: 8 NW MFMAs with ONE other action behind every second one: the 8 + NW fragment reads of the next k-step (set NS,
    // k-step NKS of the K tile at SM_ / SN_) in the order its MFMAs consume them, then the DMA pieces of parts DP0 and DP1 (-1: none), the
    // stream bookkeeping last.  Scheduling fences pin the order (a clump of reads or DMA behind a few MFMAs outlasts their shadow).
    // slot s (0 .. 4 NW - 1) sits behind MFMA 2 s + 1
#define Q16_SLOT(S, NS, NKS, SM_, SN_, DP0, DP1, s_)                                                                     \
    do {                                                                                                                 \
        constexpr int s__ = (s_);                                                                                        \
        if (s__ < NW) Q16_RDB(NS, NKS, (s__ < NW ? s__ : 0), SN_);                                                       \
        else if (s__ < NW + 8) Q16_RDA(NS, NKS, (s__ >= NW && s__ < NW + 8 ? s__ - NW : 0), SM_);                        \
        else if (s__ < NW + 12) { if ((DP0) >= 0) Q16_ISSUE1((DP0) < 0 ? 0 : (DP0), (s__ - NW - 8) & 3); }               \
        else if (s__ == NW + 12) { if ((DP0) >= 0) Q16_ADVANCE((DP0) < 0 ? 0 : (DP0)); }   /* before DP1's pieces: part 3's advance moves the stream to the next K tile */ \
        else if (s__ < NW + 17) { if ((DP1) >= 0) Q16_ISSUE1((DP1) < 0 ? 0 : (DP1), (s__ - NW - 13) & 3); }              \
        else if (s__ == NW + 17) { if ((DP1) >= 0) Q16_ADVANCE((DP1) < 0 ? 0 : (DP1)); }                                 \
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
        Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, 0);  Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, 4);                   \
        Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, 8);  Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, 12);                  \
        Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, 16); Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, 20);                  \
        if (NW == 8) { Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, (NW == 8 ? 24 : 0)); Q16_PAIR4(S, Z, NS, NKS, SM_, SN_, D0, D1, (NW == 8 ? 28 : 0)); } \
    } while (0)
    int rA = 0, rB = 0;   // ring slots of A_0 / B_0 of the K tile being multiplied
    // K tile t.  Staged during k-step 0: B_1(t+1), A_0(t+2); during k-step 1 (behind the barrier that frees the slot of A_0(t)): B_0(t+2),
    // A_1(t+2).  The barrier between the k-steps, behind a counted wait that leaves the pieces of A_0(t+2) in flight, publishes K tile t+1
    // (its last part, B_1(t+1), was issued at the head of this K tile's k-step 0) and retires every read of K tile t (k-step 1's fragments
    // were read during k-step 0).
#define Q16_RS(R_, ADD_) ((R_) + (ADD_) >= NSLOT ? (R_) + (ADD_) - NSLOT : (R_) + (ADD_))
#define Q16_KTILE(FIRST)                                                                                                 \
    do {                                                                                                                 \
        const unsigned char* sM = lds + Q16_RS(rA, wr) * Q8_HALF;                                                        \
        const unsigned char* sN = lds + (NSLOT + Q16_RS(rB, wc)) * Q8_HALF;                                              \
        const unsigned char* sMn = lds + Q16_RS(rA, 2 + wr) * Q8_HALF;                                                   \
        const unsigned char* sNn = lds + (NSLOT + Q16_RS(rB, 2 + wc)) * Q8_HALF;                                         \
        Q16_KSTEP(0, FIRST, 1, 1, sM, sN, 3, 0);                                                                         \
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); Q16_SB();                                                     \
        __builtin_amdgcn_s_barrier(); Q16_SB();                                                                          \
        Q16_KSTEP(1, false, 0, 0, sMn, sNn, 1, 2);                                                                       \
        rA = Q16_RS(rA, 2); rB = Q16_RS(rB, 2);                                                                          \
    } while (0)

    // prologue: K tile 0 and A_0 / B_0 / A_1 of K tile 1 issued, K tile 0 landed and published, its first k-step's fragments read
    q_cv(false);
    if (qv < total) Q16_ITEM();
    Q16_ISSUE_ALL(0); Q16_ADVANCE(0); Q16_ISSUE_ALL(1); Q16_ADVANCE(1); Q16_ISSUE_ALL(2); Q16_ADVANCE(2); Q16_ISSUE_ALL(3); Q16_ADVANCE(3);
    Q16_ISSUE_ALL(0); Q16_ADVANCE(0); Q16_ISSUE_ALL(1); Q16_ADVANCE(1); Q16_ISSUE_ALL(2); Q16_ADVANCE(2);
    if (NPB == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");   // in flight: A_0, B_0, A_1 of K tile 1
    __builtin_amdgcn_s_barrier();
    {
        const unsigned char* sM = lds + wr * Q8_HALF;
        const unsigned char* sN = lds + (NSLOT + wc) * Q8_HALF;
#pragma unroll
        for (int t = 0; t < NW; ++t) Q16_RDB(0, 0, t, sN);
#pragma unroll
        for (int i = 0; i < 8; ++i) Q16_RDA(0, 0, i, sM);
    }

    for (int cv = it_beg; cv < total; cv += G) {
        int cm0, cn0;
        decode(cv, cm0, cn0);
        Q16_KTILE(true);
#pragma unroll 1
        for (int t = 1; t < nt; ++t) Q16_KTILE(false);
        Q16_STORE_ROW(cm0, cn0, 0); Q16_STORE_ROW(cm0, cn0, 1); Q16_STORE_ROW(cm0, cn0, 2); Q16_STORE_ROW(cm0, cn0, 3);
        Q16_STORE_ROW(cm0, cn0, 4); Q16_STORE_ROW(cm0, cn0, 5); Q16_STORE_ROW(cm0, cn0, 6); Q16_STORE_ROW(cm0, cn0, 7);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the exhausted stream's zero-length loads still write their (zero) pieces into this workgroup's LDS
#undef Q16_RS
#undef Q16_ITEM
#undef Q16_ISSUE1
#undef Q16_ISSUE_ALL
#undef Q16_ADVANCE
#undef Q16_STORE_ROW
#undef Q16_SB
#undef Q16_RDA
#undef Q16_RDB
#undef Q16_MFMA
#undef Q16_SLOT
#undef Q16_PAIR
#undef Q16_PAIR4
#undef Q16_KSTEP
#undef Q16_KTILE
}
