// bf16 attention on v_mfma_f32_16x16x32_bf16 -- the production path; the exact-f32 variant
// in attention.hip stays as the parity-mode path.
//
// One workgroup = 4 waves = 64 query rows (fwd, dQ) or 64 keys (dK/dV) of one (batch, head).  K/V/Q/dO chunks of 64 rows
// are staged ONCE per workgroup into LDS exactly as they lie in HBM (row-major, 16-B copies, pitch hd*2+32 B) and feed the
// matrix cores two ways from the same image:
//   frag_rows : ds_read_b128 -> 8 consecutive head-dim elements of one row        (contraction over head_dim:  Q K^T, dO V^T)
//   frag_tr   : ds_read_b64_tr_b16 x2 -> 8 consecutive ROWS of one head-dim column (contraction over tokens: P V, dS K, P^T dO, dS^T Q)
// so no [B,H,T,hd] permute, no K^T / V^T copy and no score matrix ever touches HBM.  MFMA operands are ordered so that each
// lane owns 4 consecutive elements of the OUTPUT's contiguous axis: probabilities are written to the per-wave LDS tile with
// 8-B stores and row reductions need 2 wave shuffles (xor 16, 32) instead of 4.
// Softmax / lse / delta / dropout (Philox, regenerated in backward) are f32 in registers; P and dS are rounded to bf16 only as
// MFMA operands, like the reference's autocast attention.
#include "attention.h"
#include <type_traits>

// tools/probes/attn_phase_probe.hip compiles this file with -DATTN_TS: wave 0 of every workgroup stamps the 100 MHz wall clock at the phase
// boundaries of the head-resident kernels (nothing of it exists in the library build)
#ifdef ATTN_TS
__device__ long long attn_ts_buf[8 * 8192];
#define ATTN_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) attn_ts_buf[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define ATTN_STAMP(i) do { } while (0)
#endif

typedef __attribute__((ext_vector_type(4))) short v4s16;
typedef hw_h16x8 bf16x8_t;

#define MFMA(a, b, c) ECAMP_MFMA_16x16x32(a, b, c)

// softmax in base 2: scores are scaled by scale*log2(e) once, exp(x) becomes the native v_exp_f32 (2^x) with no extra multiply, and
// a masked score (-1e30) needs no select: 2^(-1e30 - max(m, -1e29)) is 0.  lse is still stored in natural-log units.
#define ATTN_LOG2E 1.4426950408889634f
#define ATTN_LN2 0.6931471805599453f
__device__ __forceinline__ float attn_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int HD> struct TileCfg { static constexpr int PB = HD * 2 + 32; static constexpr int BYTES = 64 * PB; };

// rows [r0, r0+64) x HD of a (b,h) slice -> LDS (zeros past nrows)
template <int HD>
__device__ __forceinline__ void stage_tile(unsigned char* lds, const bf16_t* base, long st, int r0, int nrows, int tid) {
    constexpr int CPR = HD / 8, PB = TileCfg<HD>::PB;
#pragma unroll
    for (int i = 0; i < (64 * CPR) / 256; ++i) {
        int idx = tid + 256 * i;
        int row = idx / CPR, ch = idx % CPR;
        uint4 v = (r0 + row < nrows) ? *reinterpret_cast<const uint4*>(base + (long)(r0 + row) * st + ch * 8) : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(lds + row * PB + ch * 16) = v;
    }
}
// same, for a workgroup whose idle waves have already returned (nthr = 64 * active waves)
template <int HD>
__device__ __forceinline__ void stage_tile_n(unsigned char* lds, const bf16_t* base, long st, int r0, int nrows, int tid, int nthr) {
    constexpr int CPR = HD / 8, PB = TileCfg<HD>::PB;
    for (int idx = tid; idx < 64 * CPR; idx += nthr) {
        int row = idx / CPR, ch = idx % CPR;
        uint4 v = (r0 + row < nrows) ? *reinterpret_cast<const uint4*>(base + (long)(r0 + row) * st + ch * 8) : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(lds + row * PB + ch * 16) = v;
    }
}
template <int HD>
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* lds, int row, int kchunk) {
    return *reinterpret_cast<const bf16x8*>(lds + row * TileCfg<HD>::PB + kchunk * 16);
}
// 16 head-dim columns [o0, o0+16) x 8 consecutive rows starting at kb + 8*(lane>>4)
template <int HD>
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char* lds, int o0, int kb, int lane) {
    constexpr int PB = TileCfg<HD>::PB;
    const int i = lane & 15, g = lane >> 4;
    const unsigned char* p = lds + (kb + g * 8 + (i >> 2)) * PB + (o0 + (i & 3) * 4) * 2;
    v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s16 __attribute__((address_space(3)))*)(p));
    v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s16 __attribute__((address_space(3)))*)(p + 4 * PB));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// A/B fragments straight from global memory for the 16 rows a wave owns (row = r0 + lane&15, 8 elems at ks*32 + 8*(lane>>4))
template <int HD>
__device__ __forceinline__ void load_row_frags(bf16x8 (&f)[HD / 32], const bf16_t* base, long st, int r0, int nrows, int lane) {
    const int row = r0 + (lane & 15);
    const bf16_t* p = base + (long)row * st + (lane >> 4) * 8;
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
        uint4 v = row < nrows ? *reinterpret_cast<const uint4*>(p + ks * 32) : make_uint4(0, 0, 0, 0);
        f[ks] = __builtin_bit_cast(bf16x8, v);
    }
}
// per-wave 16 x 64 bf16 tile (P or dS), 128-B rows, 16-B chunks XOR-swizzled by row
__device__ __forceinline__ void ptile_write4(unsigned char* t, int row, int col0, const float (&v)[4]) {
    uint2 u;
    u.x = pack_bf16x2(v[0], v[1]);
    u.y = pack_bf16x2(v[2], v[3]);
    int chunk = col0 >> 3;
    *reinterpret_cast<uint2*>(t + row * 128 + ((chunk ^ (row & 7)) << 4) + (col0 & 7) * 2) = u;
}
__device__ __forceinline__ bf16x8 ptile_frag(const unsigned char* t, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(t + row * 128 + ((chunk ^ (row & 7)) << 4));
}
__device__ __forceinline__ float red4_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float red4_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v, float mul) {
    uint2 u;
    u.x = pack_bf16x2(v[0] * mul, v[1] * mul);
    u.y = pack_bf16x2(v[2] * mul, v[3] * mul);
    *reinterpret_cast<uint2*>(p) = u;
}

// keep-mask scales of 4 consecutive attention probabilities (keys j0..j0+3 of query row `row`): element index e = row*Tk + j under the
// library's convention (common.h: halfword (e & 7) of Philox(counter e >> 3)).  When Tk and j0 are multiples of 4 the four elements are
// half of ONE Philox call; otherwise one call per element.
__device__ __forceinline__ void attn_drop4(const AttnArgs& a, uint64_t row, int j0, float inv_keep, float (&m)[4]) {
    const uint64_t e0 = row * (uint64_t)a.Tk + (uint64_t)j0;
    if ((a.Tk & 3) == 0) {
        dropout_scale4(a.seed, a.offset, e0 >> 2, a.drop_p, inv_keep, m);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = dropout_scale(a.seed, a.offset, e0 + r, a.drop_p, inv_keep);
    }
}

// The dK/dV kernels hold the transposed tile: a lane owns ONE key kj and 4 consecutive query rows i0..i0+3, whose mask halfwords
// live in four different Philox outputs (one per row).  The 4 lanes of a quad own keys 4m..4m+3, i.e. the same half (two words) of
// each of those outputs: lane x of the quad computes the output of row i0+x, and the quad exchanges the two words with DPP quad
// broadcasts -- one Philox call per lane instead of four.  Same mask(e) as everywhere else.
__device__ __forceinline__ void attn_drop4_col(const AttnArgs& a, uint64_t row0, int kj, int li, float inv_keep, float (&m)[4]) {
    if ((a.Tk & 3) == 0) {
        const int x = li & 3;
        const uint64_t e4 = ((row0 + (uint64_t)x) * (uint64_t)a.Tk + (uint64_t)(kj & ~3)) >> 2;   // the quad's group of four keys in row i0+x
        const uint4 P = philox4x32(a.seed, a.offset, e4 >> 1);
        const uint32_t thr = dropout_thr(a.drop_p);
#define ECAMP_QUAD_WORD(R)                                                                                      \
        do {                                                                                                    \
            /* (whether a row's group is the low or the high half of its call depends on the row: Tk / 4 may be odd) */ \
            const uint32_t a0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)((e4 & 1) ? P.z : P.x), (R) * 0x55, 0xf, 0xf, false); \
            const uint32_t a1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)((e4 & 1) ? P.w : P.y), (R) * 0x55, 0xf, 0xf, false); \
            m[(R)] = dropout_pick(x < 2 ? a0 : a1, x & 1, thr, inv_keep);                                        \
        } while (0)
        ECAMP_QUAD_WORD(0); ECAMP_QUAD_WORD(1); ECAMP_QUAD_WORD(2); ECAMP_QUAD_WORD(3);
#undef ECAMP_QUAD_WORD
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = dropout_scale(a.seed, a.offset, (row0 + (uint64_t)r) * (uint64_t)a.Tk + (uint64_t)kj, a.drop_p, inv_keep);
    }
}

// =============================================================================================
template <int HD, int KCH>
__global__ __launch_bounds__(256) void attn16_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* KV = smem;                         // 64-row K (then V) chunk
    unsigned char* PT = smem + TileCfg<HD>::BYTES;    // 4 per-wave P tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int b = blockIdx.y / a.H, h = blockIdx.y % a.H;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const long bh = (long)b * a.H + h;
    const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_sb + h * a.q_sh;
    const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_sb + h * a.k_sh;
    const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_sb + h * a.v_sh;
    bf16_t* ob = reinterpret_cast<bf16_t*>(a.o) + b * a.o_sb + h * a.o_sh;

    bf16x8 qf[HD / 32];
    load_row_frags<HD>(qf, qb, a.q_st, q0, a.Tq, lane);

    f32x4 s[KCH * 4];  // s[c*4+jt][r] = S[i = q0+li][j = c*64 + jt*16 + 4g + r]
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        __syncthreads();
        stage_tile<HD>(KV, kb, a.k_st, c * 64, a.Tk, tid);
        __syncthreads();
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) acc = MFMA(frag_rows<HD>(KV, jt * 16 + li, ks * 4 + g), qf[ks], acc);
            s[c * 4 + jt] = acc;
        }
    }
    float mx = NEG_BIG;
#pragma unroll
    for (int t = 0; t < KCH * 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int j = t * 16 + 4 * g + r;
            bool ok = j < a.Tk && (a.key_mask == nullptr || a.key_mask[(long)b * a.Tk + j] != 0);
            s[t][r] = ok ? s[t][r] * a.scale : NEG_BIG;
            mx = fmaxf(mx, s[t][r]);
        }
    mx = red4_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < KCH * 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float p = s[t][r] > 0.5f * NEG_BIG ? __expf(s[t][r] - mx) : 0.f;
            s[t][r] = p;
            sum += p;
        }
    sum = red4_sum(sum);
    const int qi = q0 + li;
    if (g == 0 && qi < a.Tq) a.lse[bh * a.Tq + qi] = mx + __logf(sum);
    const float inv = 1.0f / sum;
    const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;

    f32x4 o[HD / 16];  // o[dt][r] = O[i = q0+li][d = dt*16 + 4g + r]
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned char* pt = PT + wave * 2048;
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        __syncthreads();
        stage_tile<HD>(KV, vb, a.v_st, c * 64, a.Tk, tid);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = s[c * 4 + jt][r] * inv;
            if (a.drop_p > 0.f) {
                float dm[4];
                attn_drop4(a, (uint64_t)bh * a.Tq + qi, c * 64 + jt * 16 + 4 * g, inv_keep, dm);
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] *= dm[r];
            }
            ptile_write4(pt, li, jt * 16 + 4 * g, p);
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 pf = ptile_frag(pt, li, kk * 4 + g);
#pragma unroll
            for (int dt = 0; dt < HD / 16; ++dt) o[dt] = MFMA(frag_tr<HD>(KV, dt * 16, kk * 32, lane), pf, o[dt]);
        }
    }
    if (qi < a.Tq) {
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) store4(ob + (long)qi * a.o_st + dt * 16 + 4 * g, o[dt], 1.0f);
    }
}

// =============================================================================================
// Forward for ANY key length: keys streamed in chunks of 64 with the online-softmax recurrence (running max m, running
// sum l, O rescaled by exp(m_old - m_new)).  Used when Tk > 256 (ViT-L/16 at 448^2: decoder sequence 785).
template <int HD, bool PART>
__device__ __forceinline__ void attn16_fwd_long_body(const AttnArgs& a, int nthr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* KT = smem;
    unsigned char* VT = smem + TileCfg<HD>::BYTES;
    unsigned char* PT = smem + 2 * TileCfg<HD>::BYTES;
    int* KM = reinterpret_cast<int*>(PT + 4 * 2048);    // attend-flags of the 64 keys of the staged chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int b = blockIdx.y / a.H, h = blockIdx.y % a.H;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const long bh = (long)b * a.H + h;
    const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_sb + h * a.q_sh;
    const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_sb + h * a.k_sh;
    const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_sb + h * a.v_sh;
    bf16_t* ob = reinterpret_cast<bf16_t*>(a.o) + b * a.o_sb + h * a.o_sh;
    bf16x8 qf[HD / 32];
    load_row_frags<HD>(qf, qb, a.q_st, q0, a.Tq, lane);
    const int qi = q0 + li;
    const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const float sc2 = a.scale * ATTN_LOG2E;
    float m = NEG_BIG, l = 0.f;
    f32x4 o[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned char* pt = PT + wave * 2048;
    const int nkc = (a.Tk + 63) / 64;
#pragma unroll 1
    for (int c = 0; c < nkc; ++c) {
        __syncthreads();
        if (PART) {
            stage_tile_n<HD>(KT, kb, a.k_st, c * 64, a.Tk, tid, nthr);
            stage_tile_n<HD>(VT, vb, a.v_st, c * 64, a.Tk, tid, nthr);
        } else {
            stage_tile<HD>(KT, kb, a.k_st, c * 64, a.Tk, tid);
            stage_tile<HD>(VT, vb, a.v_st, c * 64, a.Tk, tid);
        }
        if (tid < 64) {
            const int j = c * 64 + tid;
            KM[tid] = (j < a.Tk && (a.key_mask == nullptr || a.key_mask[(long)b * a.Tk + j] != 0)) ? 1 : 0;
        }
        __syncthreads();
        f32x4 s[4];
        float cmx = NEG_BIG;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) acc = MFMA(frag_rows<HD>(KT, jt * 16 + li, ks * 4 + g), qf[ks], acc);
            const int4 km = *reinterpret_cast<const int4*>(KM + jt * 16 + 4 * g);
            const int kmv[4] = {km.x, km.y, km.z, km.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[r] = kmv[r] != 0 ? acc[r] * sc2 : NEG_BIG;
                cmx = fmaxf(cmx, acc[r]);
            }
            s[jt] = acc;
        }
        cmx = red4_max(cmx);
        const float mn = fmaxf(m, cmx);
        const float alpha = attn_exp2(m - mn);  // first chunk: 2^(-1e30 - finite) = 0, and l = o = 0 anyway
        const float mns = fmaxf(mn, -1e29f);    // a row with no valid key so far keeps every p at 0
        float csum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = attn_exp2(s[jt][r] - mns);
                csum += p[r];
            }
            if (a.drop_p > 0.f) {
                float dm[4];
                attn_drop4(a, (uint64_t)bh * a.Tq + qi, c * 64 + jt * 16 + 4 * g, inv_keep, dm);
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] *= dm[r];
            }
            ptile_write4(pt, li, jt * 16 + 4 * g, p);
        }
        l = l * alpha + red4_sum(csum);
        m = mn;
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 pf = ptile_frag(pt, li, kk * 4 + g);
#pragma unroll
            for (int dt = 0; dt < HD / 16; ++dt) o[dt] = MFMA(frag_tr<HD>(VT, dt * 16, kk * 32, lane), pf, o[dt]);
        }
    }
    if (qi < a.Tq) {
        if (g == 0) a.lse[bh * a.Tq + qi] = (m + __log2f(l)) * ATTN_LN2;
        const float inv = 1.0f / l;
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) store4(ob + (long)qi * a.o_st + dt * 16 + 4 * g, o[dt], inv);
    }
}

template <int HD>
__global__ __launch_bounds__(256) void attn16_fwd_long_kernel(AttnArgs a) {
    const int nact = min(4, (a.Tq - (int)blockIdx.x * 64 + 15) >> 4);   // see attn16_bwd_dq_kernel
    if (nact < 4) {
        if ((int)(threadIdx.x >> 6) >= nact) return;
        attn16_fwd_long_body<HD, true>(a, nact * 64);
    } else {
        attn16_fwd_long_body<HD, false>(a, 256);
    }
}

// =============================================================================================
// dQ (+ delta).  D layouts: S/dP [j = 4g+r][i = li]; dQ [d = 4g+r][i = li]
template <int HD, int KCH, bool PART>
__device__ __forceinline__ void attn16_bwd_dq_body(const AttnArgs& a, int nthr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* KT = smem;
    unsigned char* VT = smem + TileCfg<HD>::BYTES;
    unsigned char* ST = smem + 2 * TileCfg<HD>::BYTES;  // 4 per-wave dS tiles
    int* KM = reinterpret_cast<int*>(ST + 4 * 2048);    // attend-flags of the 64 keys of the staged chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int b = blockIdx.y / a.H, h = blockIdx.y % a.H;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const long bh = (long)b * a.H + h;
    const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_sb + h * a.q_sh;
    const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_sb + h * a.k_sh;
    const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_sb + h * a.v_sh;
    const bf16_t* ob = reinterpret_cast<const bf16_t*>(a.o) + b * a.o_sb + h * a.o_sh;
    const bf16_t* gb = reinterpret_cast<const bf16_t*>(a.dout) + b * a.do_sb + h * a.do_sh;
    bf16_t* dqb = reinterpret_cast<bf16_t*>(a.dq) + b * a.dq_sb + h * a.dq_sh;

    bf16x8 qf[HD / 32], gf[HD / 32], of[HD / 32];
    load_row_frags<HD>(qf, qb, a.q_st, q0, a.Tq, lane);
    load_row_frags<HD>(gf, gb, a.do_st, q0, a.Tq, lane);
    load_row_frags<HD>(of, ob, a.o_st, q0, a.Tq, lane);
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += bf2f((bf16_t)gf[ks][e]) * bf2f((bf16_t)of[ks][e]);
    dl = red4_sum(dl);
    const int qi = q0 + li;
    const bool qok = qi < a.Tq;
    if (g == 0 && qok) a.delta[bh * a.Tq + qi] = dl;
    const float lse2 = (qok ? a.lse[bh * a.Tq + qi] : 0.f) * ATTN_LOG2E, sc2 = a.scale * ATTN_LOG2E;
    const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;

    f32x4 dq[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned char* st = ST + wave * 2048;
    const int nkc = KCH ? KCH : (a.Tk + 63) / 64;  // KCH == 0: any key length (ViT-L/448 decoder, T = 785)
#pragma unroll 1
    for (int c = 0; c < nkc; ++c) {
        __syncthreads();
        if (PART) {
            stage_tile_n<HD>(KT, kb, a.k_st, c * 64, a.Tk, tid, nthr);
            stage_tile_n<HD>(VT, vb, a.v_st, c * 64, a.Tk, tid, nthr);
        } else {
            stage_tile<HD>(KT, kb, a.k_st, c * 64, a.Tk, tid);
            stage_tile<HD>(VT, vb, a.v_st, c * 64, a.Tk, tid);
        }
        if (tid < 64) {   // key validity once per chunk (it was a global load per score inside the tile loop)
            const int j = c * 64 + tid;
            KM[tid] = (j < a.Tk && (a.key_mask == nullptr || a.key_mask[(long)b * a.Tk + j] != 0)) ? 1 : 0;
        }
        __syncthreads();
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) {
                s = MFMA(frag_rows<HD>(KT, jt * 16 + li, ks * 4 + g), qf[ks], s);
                dp = MFMA(frag_rows<HD>(VT, jt * 16 + li, ks * 4 + g), gf[ks], dp);
            }
            const int4 km = *reinterpret_cast<const int4*>(KM + jt * 16 + 4 * g);
            const int kmv[4] = {km.x, km.y, km.z, km.w};
            float ds[4], dm[4] = {1.f, 1.f, 1.f, 1.f};
            if (a.drop_p > 0.f) attn_drop4(a, (uint64_t)bh * a.Tq + qi, c * 64 + jt * 16 + 4 * g, inv_keep, dm);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = qok && kmv[r] != 0;
                float p = ok ? attn_exp2(fmaf(s[r], sc2, -lse2)) : 0.f;
                ds[r] = p * (dp[r] * dm[r] - dl);
            }
            ptile_write4(st, li, jt * 16 + 4 * g, ds);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 sf = ptile_frag(st, li, kk * 4 + g);
#pragma unroll
            for (int dt = 0; dt < HD / 16; ++dt) dq[dt] = MFMA(frag_tr<HD>(KT, dt * 16, kk * 32, lane), sf, dq[dt]);
        }
    }
    if (qok) {
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) store4(dqb + (long)qi * a.dq_st + dt * 16 + 4 * g, dq[dt], a.scale);
    }
}

// The last 64-row workgroup of a sequence may own fewer than four 16-row tiles (T = 197: one): its idle waves return before the
// first barrier (a barrier only waits for the waves still alive) and the staging loops run over the remaining threads.  Written
// as two instantiations of one body so that the full workgroups keep exactly the code (and registers) they had.
template <int HD, int KCH>
__global__ __launch_bounds__(256) void attn16_bwd_dq_kernel(AttnArgs a) {
    const int nact = min(4, (a.Tq - (int)blockIdx.x * 64 + 15) >> 4);
    if (nact < 4) {
        if ((int)(threadIdx.x >> 6) >= nact) return;
        attn16_bwd_dq_body<HD, KCH, true>(a, nact * 64);
    } else {
        attn16_bwd_dq_body<HD, KCH, false>(a, 256);
    }
}

// =============================================================================================
// dK, dV.  D layouts: S/dP [i = 4g+r][j = li]; dK/dV [d = 4g+r][j = li]
template <int HD, bool PART>
__device__ __forceinline__ void attn16_bwd_dkv_body(const AttnArgs& a, int nthr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* QT = smem;
    unsigned char* GT = smem + TileCfg<HD>::BYTES;
    unsigned char* PT = smem + 2 * TileCfg<HD>::BYTES;  // 4 per-wave P_dropped^T tiles [j][i]
    unsigned char* ST = PT + 4 * 2048;                  // 4 per-wave dS^T tiles
    float* LS = reinterpret_cast<float*>(ST + 4 * 2048);  // lse and delta of the 64 query rows of the staged chunk (2 x 64 floats)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int b = blockIdx.y / a.H, h = blockIdx.y % a.H;
    const int j0 = blockIdx.x * 64 + wave * 16;
    const long bh = (long)b * a.H + h;
    const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_sb + h * a.q_sh;
    const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_sb + h * a.k_sh;
    const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_sb + h * a.v_sh;
    const bf16_t* gb = reinterpret_cast<const bf16_t*>(a.dout) + b * a.do_sb + h * a.do_sh;
    bf16_t* dkb = reinterpret_cast<bf16_t*>(a.dk) + b * a.dk_sb + h * a.dk_sh;
    bf16_t* dvb = reinterpret_cast<bf16_t*>(a.dv) + b * a.dv_sb + h * a.dv_sh;

    bf16x8 kf[HD / 32], vf[HD / 32];
    load_row_frags<HD>(kf, kb, a.k_st, j0, a.Tk, lane);
    load_row_frags<HD>(vf, vb, a.v_st, j0, a.Tk, lane);
    const int kj = j0 + li;
    const bool jok = kj < a.Tk && (a.key_mask == nullptr || a.key_mask[(long)b * a.Tk + kj] != 0);
    const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f, sc2 = a.scale * ATTN_LOG2E;
    f32x4 dk[HD / 16], dv[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) dk[dt] = dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned char* pt = PT + wave * 2048;
    unsigned char* st = ST + wave * 2048;
    const int nqc = (a.Tq + 63) / 64;
#pragma unroll 1
    for (int c = 0; c < nqc; ++c) {
        __syncthreads();
        if (PART) {
            stage_tile_n<HD>(QT, qb, a.q_st, c * 64, a.Tq, tid, nthr);
            stage_tile_n<HD>(GT, gb, a.do_st, c * 64, a.Tq, tid, nthr);
        } else {
            stage_tile<HD>(QT, qb, a.q_st, c * 64, a.Tq, tid);
            stage_tile<HD>(GT, gb, a.do_st, c * 64, a.Tq, tid);
        }
        if (tid < 64) {   // per-row statistics once per chunk (they were 32 global loads per lane and chunk inside the tile loop)
            const int i = c * 64 + tid;
            LS[tid] = i < a.Tq ? a.lse[bh * a.Tq + i] * ATTN_LOG2E : 0.f;   // base-2 units
            LS[64 + tid] = i < a.Tq ? a.delta[bh * a.Tq + i] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) {
                s = MFMA(frag_rows<HD>(QT, it * 16 + li, ks * 4 + g), kf[ks], s);
                dp = MFMA(frag_rows<HD>(GT, it * 16 + li, ks * 4 + g), vf[ks], dp);
            }
            float pd[4], ds[4], dm[4] = {1.f, 1.f, 1.f, 1.f};
            if (a.drop_p > 0.f) attn_drop4_col(a, (uint64_t)bh * a.Tq + (c * 64 + it * 16 + 4 * g), kj, li, inv_keep, dm);
            const f32x4 lse4 = *reinterpret_cast<const f32x4*>(LS + it * 16 + 4 * g);
            const f32x4 dl4 = *reinterpret_cast<const f32x4*>(LS + 64 + it * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int i = c * 64 + it * 16 + 4 * g + r;
                bool ok = jok && i < a.Tq;
                float p = ok ? attn_exp2(fmaf(s[r], sc2, -lse4[r])) : 0.f;
                pd[r] = p * dm[r];
                ds[r] = p * (dp[r] * dm[r] - dl4[r]);
            }
            ptile_write4(pt, li, it * 16 + 4 * g, pd);
            ptile_write4(st, li, it * 16 + 4 * g, ds);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 pf = ptile_frag(pt, li, kk * 4 + g), sf = ptile_frag(st, li, kk * 4 + g);
#pragma unroll
            for (int dt = 0; dt < HD / 16; ++dt) {
                dv[dt] = MFMA(frag_tr<HD>(GT, dt * 16, kk * 32, lane), pf, dv[dt]);
                dk[dt] = MFMA(frag_tr<HD>(QT, dt * 16, kk * 32, lane), sf, dk[dt]);
            }
        }
    }
    if (kj < a.Tk) {
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) {
            store4(dkb + (long)kj * a.dk_st + dt * 16 + 4 * g, dk[dt], a.scale);
            store4(dvb + (long)kj * a.dv_st + dt * 16 + 4 * g, dv[dt], 1.0f);
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void attn16_bwd_dkv_kernel(AttnArgs a) {
    const int nact = min(4, (a.Tk - (int)blockIdx.x * 64 + 15) >> 4);   // 16-key tiles of this workgroup that hold a key
    if (nact < 4) {
        if ((int)(threadIdx.x >> 6) >= nact) return;
        attn16_bwd_dkv_body<HD, true>(a, nact * 64);
    } else {
        attn16_bwd_dkv_body<HD, false>(a, 256);
    }
}

// =============================================================================================
// "Resident" forward: one workgroup per (batch, head); ALL keys/values are staged into LDS once, then each wave walks 16-row
// query tiles with no further workgroup barrier.  Used for Tk <= 128 (the encoder and the report side).
template <int HD, int KCH>
__global__ __launch_bounds__(256) void attn16r_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TB = TileCfg<HD>::BYTES;
    unsigned char* KT = smem;
    unsigned char* VT = smem + KCH * TB;
    unsigned char* PT = smem + 2 * KCH * TB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const long bh = blockIdx.x;
    const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_sb + h * a.q_sh;
    const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_sb + h * a.k_sh;
    const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_sb + h * a.v_sh;
    bf16_t* ob = reinterpret_cast<bf16_t*>(a.o) + b * a.o_sb + h * a.o_sh;
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        stage_tile<HD>(KT + c * TB, kb, a.k_st, c * 64, a.Tk, tid);
        stage_tile<HD>(VT + c * TB, vb, a.v_st, c * 64, a.Tk, tid);
    }
    __syncthreads();
    const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    unsigned char* pt = PT + wave * 2048;
    unsigned long long kok = 0ull;  // bit (t*4 + r): key t*16 + 4g + r is attended (one register pair instead of a spilled bool array)
#pragma unroll
    for (int t = 0; t < KCH * 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int j = t * 16 + 4 * g + r;
            bool ok = j < a.Tk && (a.key_mask == nullptr || a.key_mask[(long)b * a.Tk + j] != 0);
            kok |= (unsigned long long)(ok ? 1 : 0) << (t * 4 + r);
        }
    for (int q0 = wave * 16; q0 < a.Tq; q0 += 64) {
        bf16x8 qf[HD / 32];
        load_row_frags<HD>(qf, qb, a.q_st, q0, a.Tq, lane);
        f32x4 s[KCH * 4];
#pragma unroll
        for (int c = 0; c < KCH; ++c)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < HD / 32; ++ks) acc = MFMA(frag_rows<HD>(KT + c * TB, jt * 16 + li, ks * 4 + g), qf[ks], acc);
                s[c * 4 + jt] = acc;
            }
        float mx = NEG_BIG;
#pragma unroll
        for (int t = 0; t < KCH * 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[t][r] = ((kok >> (t * 4 + r)) & 1ull) ? s[t][r] * (a.scale * ATTN_LOG2E) : NEG_BIG;
                mx = fmaxf(mx, s[t][r]);
            }
        mx = red4_max(mx);
        const float mxs = fmaxf(mx, -1e29f);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < KCH * 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = attn_exp2(s[t][r] - mxs);
                s[t][r] = p;
                sum += p;
            }
        sum = red4_sum(sum);
        const int qi = q0 + li;
        if (g == 0 && qi < a.Tq) a.lse[bh * a.Tq + qi] = (mx + __log2f(sum)) * ATTN_LN2;
        const float inv = 1.0f / sum;
        f32x4 o[HD / 16];
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                float p[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] = s[c * 4 + jt][r] * inv;
                if (a.drop_p > 0.f) {
                    float dm[4];
                    attn_drop4(a, (uint64_t)bh * a.Tq + qi, c * 64 + jt * 16 + 4 * g, inv_keep, dm);
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[r] *= dm[r];
                }
                ptile_write4(pt, li, jt * 16 + 4 * g, p);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 pf = ptile_frag(pt, li, kk * 4 + g);
#pragma unroll
                for (int dt = 0; dt < HD / 16; ++dt) o[dt] = MFMA(frag_tr<HD>(VT + c * TB, dt * 16, kk * 32, lane), pf, o[dt]);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
        if (qi < a.Tq) {
#pragma unroll
            for (int dt = 0; dt < HD / 16; ++dt) store4(ob + (long)qi * a.o_st + dt * 16 + 4 * g, o[dt], 1.0f);
        }
    }
}

// =============================================================================================
// "Head" kernels: one workgroup per (batch, head), everything the head needs staged ONCE, no workgroup barrier inside the tile
// loops and NO LDS round trip for the probabilities.  Every attention of this model is short (50 / 128 / 197 tokens) and tiny in
// FLOPs: the measured bounds of the 64-row streaming kernels above were the per-chunk barrier + global-latency chain, ~13 VALU
// instructions per score element, and 256^2 scores computed for a 197^2 problem (profiles/r02_pmc_sq_attention.txt).
//   * LDS images are unpadded, 16-B chunks XOR-swizzled by row (hswz), zero-filled up to a multiple of 32 rows.
//   * S^T = K Q^T on v_mfma_f32_16x16x32_bf16 leaves lane (li, g) with scores of query li and keys 16t + 4g + r.  Two key tiles
//     (t, t+1) of the SAME lane are exactly one B operand of the next MFMA if its contraction index is taken in the order
//     kappa = 8g + e  <->  key 16(t + e/4) + 4g + e%4: the A operand (V^T, K^T, dO^T, Q^T via ds_read_b64_tr_b16) simply reads
//     its two 4-row groups from rows 16t + 4g and 16(t+1) + 4g.  P / dS never leave the registers.
//   * One multiply puts a score in base-2 units (an fma where a tile pair holds masked / padded keys: additive -1e30), then
//     max / subtract, v_exp_f32, sum, pack; the normalisation is applied to O once.
//   * Tile loops run over the 32-key pairs that exist (7 x 13 for T = 197, not 8 x 16).
//   * Backward is ONE kernel: phase 1 (K, V resident; a wave owns 16 queries) produces dQ and leaves lse / delta in LDS; phase 2
//     (Q, dO staged over the same LDS; a wave owns 16 keys) produces dK and dV.  The operands a wave owns come straight from
//     global memory as MFMA fragments (the second read of each tensor is an L2 hit: the same workgroup has just staged it).
template <int HD> struct HeadCfg {
    static constexpr int RB = HD * 2, CPR = HD / 8;   // row bytes, 16-B chunks per row
    // XOR key of row r (period 16 rows, 4 bits per row, row 0 in the low nibble).  Found by search (tools/probes/lds_swizzle_search.py)
    // against the lane groups of MI355X_MICROARCH.md "LDS": conflict-free for the ds_read_b128 row fragments (groups {0-3,12-15,
    // 20-27}, ...), for the ds_read_b64_tr_b16 column fragments (32-lane halves) and for the ds_write_b128 staging (8 contiguous
    // lanes); the plain (row / rows-per-bank-window) key it replaces left 2-way conflicts on a third of the LDS cycles.
    static constexpr unsigned long long KEYS = HD == 32 ? 0x3033030132330101ull : HD == 64 ? 0x6056142366024421ull : 0xE0CA8436CE0A3864ull;
};
template <int HD> __device__ __forceinline__ int hswz(int row) { return (int)((HeadCfg<HD>::KEYS >> (4 * (row & 15))) & 15ull); }
template <int HD> __device__ __forceinline__ unsigned haddr(int row, int chunk) {
    return (unsigned)(row * HeadCfg<HD>::RB + ((chunk ^ hswz<HD>(row)) << 4));
}
// rows [0, nrows) of a (b, h) slice -> swizzled image of `prow` rows (zeros past nrows); four independent loads in flight per thread
template <int HD>
__device__ __forceinline__ void head_stage(unsigned char* img, const bf16_t* base, long st, int nrows, int prow, int tid, int nthr) {
    constexpr int CPR = HeadCfg<HD>::CPR;
    const int n = prow * CPR;
    for (int i0 = tid; i0 < n; i0 += 4 * nthr) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = i0 + u * nthr, row = idx / CPR, ch = idx % CPR;
            v[u] = (idx < n && row < nrows) ? *reinterpret_cast<const uint4*>(base + (long)row * st + ch * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = i0 + u * nthr, row = idx / CPR, ch = idx % CPR;
            if (idx < n) *reinterpret_cast<uint4*>(img + haddr<HD>(row, ch)) = v[u];
        }
    }
}
// Two images of `prow` rows staged together, U 16-B loads of EACH in flight per thread before the first LDS write (round 5).  One image at a
// time, four loads per thread and round, the hd 128 backward pass (256 threads, 2 x 32 KB per phase) opened each of its two phases with four
// memory round trips in series -- about half of a workgroup's own time on a kernel whose waves wait 50 % of their cycles
// (profiles/r05_pmc_sq_all_kernels.txt); with U = 8 it is one.  The registers are free at both points (no accumulator is live yet).
template <int HD, int U>
__device__ __forceinline__ void head_stage_pair(unsigned char* imgA, const bf16_t* baseA, long stA, unsigned char* imgB, const bf16_t* baseB, long stB,
                                                int nrows, int prow, int tid, int nthr) {
    constexpr int CPR = HeadCfg<HD>::CPR;
    const int n = prow * CPR;
    for (int i0 = tid; i0 < n; i0 += U * nthr) {
        uint4 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = i0 + u * nthr, row = idx / CPR, ch = idx % CPR;
            const bool in = idx < n && row < nrows;
            va[u] = in ? *reinterpret_cast<const uint4*>(baseA + (long)row * stA + ch * 8) : make_uint4(0, 0, 0, 0);
            vb[u] = in ? *reinterpret_cast<const uint4*>(baseB + (long)row * stB + ch * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = i0 + u * nthr, row = idx / CPR, ch = idx % CPR;
            if (idx < n) {
                *reinterpret_cast<uint4*>(imgA + haddr<HD>(row, ch)) = va[u];
                *reinterpret_cast<uint4*>(imgB + haddr<HD>(row, ch)) = vb[u];
            }
        }
    }
}
// packed f32 pairs: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 cost one VALU slot for two lanes' worth of work
// (tools/probes/valu_rate_probe.hip: same issue rate as v_fma_f32; v_exp_f32 1.5x)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 hexp2(f32x2 x) { return (f32x2){attn_exp2(x[0]), attn_exp2(x[1])}; }
__device__ __forceinline__ uint32_t hpk(f32x2 v) { return pack_bf16x2(v[0], v[1]); }
__device__ __forceinline__ bf16x8 hpack8(const float (&v)[8]) {
    const uint4 u = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    return __builtin_bit_cast(bf16x8, u);
}
// additive key bias (0 attend / -1e30 masked or padded) of `prow` keys
__device__ __forceinline__ void head_key_bias(float* KB, const AttnArgs& a, int b, int prow, int tid, int nthr) {
    for (int j = tid; j < prow; j += nthr)
        KB[j] = (j < a.Tk && (a.key_mask == nullptr || a.key_mask[(long)b * a.Tk + j] != 0)) ? 0.f : NEG_BIG;
}
// Per-lane LDS byte offsets of the two fragment kinds for rows 0..15 of an image; a 16-row tile t adds t * 16 * RB (the swizzle has
// a period of 16 rows), so the tile loops only add constants.
template <int HD> struct HeadOff {
    unsigned fr[HD / 32];   // ds_read_b128 of row li, 16-B chunk 4 ks + g
    unsigned tr[HD / 16];   // ds_read_b64_tr_b16 of rows 4g + (li >> 2), columns 16 dt + 4 (li & 3)
    __device__ __forceinline__ HeadOff(int li, int g) {
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) fr[ks] = haddr<HD>(li, ks * 4 + g);
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) {
            const int cb = (dt * 16 + (li & 3) * 4) * 2;
            tr[dt] = haddr<HD>(4 * g + (li >> 2), cb >> 4) + (cb & 15);
        }
    }
};
template <int HD> __device__ __forceinline__ bf16x8 hfr(const unsigned char* img, const HeadOff<HD>& o, int t, int ks) {
    return *reinterpret_cast<const bf16x8*>(img + o.fr[ks] + t * (16 * HeadCfg<HD>::RB));
}
// transposed A operand over the 32 image rows of tile pair pp (see the k-order note above)
template <int HD> __device__ __forceinline__ bf16x8 htr(const unsigned char* img, const HeadOff<HD>& o, int pp, int dt) {
    const unsigned char* p = img + o.tr[dt] + pp * (32 * HeadCfg<HD>::RB);
    v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s16 __attribute__((address_space(3)))*)(p));
    v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s16 __attribute__((address_space(3)))*)(p + 16 * HeadCfg<HD>::RB));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// Forward, two passes over the resident keys: (1) row maximum, (2) scores again -> p -> O.  Recomputing a score tile costs HD/32
// MFMAs and LDS reads; keeping all of them costs 4 * Tk/16 registers per lane and with them the occupancy, which is worth more
// here (measured on T = 197: 89 VGPRs / 2 workgroups per CU 71 us, 154 VGPRs / 1 workgroup 98 us).
// Tiles are walked in pairs (32 keys); key tiles below `nfull` hold only attended keys and skip the bias.
// FLAGS 0: no key mask, no dropout (the image side).  FLAGS 1: key bias on every tile + dropout if drop_p > 0 (the report side).
template <int HD, int FLAGS>
__global__ __launch_bounds__(512, 4) void attn_head_fwd_kernel(AttnArgs a) {   // (four waves per SIMD = two 8-wave workgroups per CU: at 154 registers the report side ran one, its loads never under its arithmetic)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RB = HeadCfg<HD>::RB;
    const int nthr = blockDim.x, nw = nthr >> 6;
    const int nkp = (((a.Tk + 15) >> 4) + 1) >> 1, prow = nkp * 32;
    const int pfull = FLAGS ? 0 : a.Tk >> 5;          // pairs of key tiles with no padded key
    unsigned char* KI = smem;
    unsigned char* VI = smem + prow * RB;
    float* KB = reinterpret_cast<float*>(smem + 2 * prow * RB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const long bh = blockIdx.x;
    const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_sb + h * a.q_sh;
    const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_sb + h * a.k_sh;
    const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_sb + h * a.v_sh;
    bf16_t* ob = reinterpret_cast<bf16_t*>(a.o) + b * a.o_sb + h * a.o_sh;
    // the fragments a wave owns come from global memory; the first block's are requested before the staging loads (requesting the
    // next block's a block ahead, into a second register set, measured the same: 228 vs 230 us on the T = 197 backward)
    bf16x8 qf[HD / 32];
    if (wave * 16 < a.Tq) load_row_frags<HD>(qf, qb, a.q_st, wave * 16, a.Tq, lane);
    head_stage_pair<HD, 4>(KI, kb, a.k_st, VI, vb, a.v_st, a.Tk, prow, tid, nthr);
    head_key_bias(KB, a, b, prow, tid, nthr);
    __syncthreads();
    const HeadOff<HD> off(li, g);
    const float sc2 = a.scale * ATTN_LOG2E;
    const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    for (int q0 = wave * 16; q0 < a.Tq; q0 += nw * 16) {
        // scores of key tile t in base-2 units (the product, not the raw MFMA result, is what fmaxf sees: no canonicalising v_max x, x)
        auto scores = [&](int t, bool biased) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) acc = MFMA(hfr<HD>(KI, off, t, ks), qf[ks], acc);
            if (biased) {
                const f32x4 kb4 = *reinterpret_cast<const f32x4*>(KB + t * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = fmaf(acc[r], sc2, kb4[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] *= sc2;
            }
            return acc;
        };
        float mx = NEG_BIG;
#pragma unroll 2
        for (int pp = 0; pp < pfull; ++pp) {
            const f32x4 s0 = scores(2 * pp, false), s1 = scores(2 * pp + 1, false);
            mx = fmaxf(fmaxf(fmaxf(mx, fmaxf(s0[0], s0[1])), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
        }
#pragma unroll 1
        for (int pp = pfull; pp < nkp; ++pp) {
            const f32x4 s0 = scores(2 * pp, true), s1 = scores(2 * pp + 1, true);
            mx = fmaxf(fmaxf(fmaxf(mx, fmaxf(s0[0], s0[1])), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
        }
        mx = red4_max(mx);
        const float mxs = fmaxf(mx, -1e29f);   // a row with no valid key keeps every p at 0
        const int qi = q0 + li;
        float sum = 0.f;
        f32x4 o[HD / 16];  // o[dt][r] = O[i = q0+li][d = dt*16 + 4g + r]
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto pair = [&](int pp, bool biased) {
            const f32x4 s0 = scores(2 * pp, biased), s1 = scores(2 * pp + 1, biased);
            float p[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = attn_exp2(s0[r] - mxs);
                p[4 + r] = attn_exp2(s1[r] - mxs);
                sum += p[r] + p[4 + r];
            }
            if (FLAGS && a.drop_p > 0.f) {
                float dm[4], dn[4];
                if ((a.Tk & 7) == 0) {
                    // A lane owns keys 4g..4g+3 of both tiles of the pair: two halves of two different Philox calls (eight keys each).  Lanes g
                    // and g ^ 1 of a query row own the other halves: the even one evaluates the call of the first tile's eight keys, the odd
                    // one the second tile's, and they swap the two words the other needs (ds_swizzle, lanes l <-> l ^ 16) -- one call per
                    // lane and pair instead of two; same mask(e) as everywhere else (common.h).
                    const bool odd = (g & 1) != 0;
                    const uint64_t e8 = ((((uint64_t)bh * a.Tq + qi) * (uint64_t)a.Tk) >> 3) + (uint64_t)(pp * 4 + (odd ? 2 : 0) + (g >> 1));
                    const uint4 P = philox4x32(a.seed, a.offset, e8);
                    const uint32_t r0 = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(odd ? P.x : P.z), 0x401f);   // xor_mask 0x10, and_mask 0x1f
                    const uint32_t r1 = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(odd ? P.y : P.w), 0x401f);
                    const uint32_t wa0 = odd ? r0 : P.x, wa1 = odd ? r1 : P.y, wb0 = odd ? P.z : r0, wb1 = odd ? P.w : r1;
                    const uint32_t thr = dropout_thr(a.drop_p);
                    dm[0] = dropout_pick(wa0, 0, thr, inv_keep); dm[1] = dropout_pick(wa0, 1, thr, inv_keep);
                    dm[2] = dropout_pick(wa1, 0, thr, inv_keep); dm[3] = dropout_pick(wa1, 1, thr, inv_keep);
                    dn[0] = dropout_pick(wb0, 0, thr, inv_keep); dn[1] = dropout_pick(wb0, 1, thr, inv_keep);
                    dn[2] = dropout_pick(wb1, 0, thr, inv_keep); dn[3] = dropout_pick(wb1, 1, thr, inv_keep);
                } else {
                    attn_drop4(a, (uint64_t)bh * a.Tq + qi, pp * 32 + 4 * g, inv_keep, dm);
                    attn_drop4(a, (uint64_t)bh * a.Tq + qi, pp * 32 + 16 + 4 * g, inv_keep, dn);
                }
                unsigned bits = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] *= dm[r];
                    p[4 + r] *= dn[r];
                    bits |= (dm[r] != 0.f ? 1u << r : 0u) | (dn[r] != 0.f ? 16u << r : 0u);
                }
                // the mask leaves as bits for the backward pass (the only Philox evaluation of this score)
                if (a.drop_bits && qi < a.Tq) a.drop_bits[(((long)bh * a.Tq + qi) * 4 + g) * 8 + pp] = (unsigned char)bits;
            }
            const bf16x8 pf = hpack8(p);
#pragma unroll
            for (int dt = 0; dt < HD / 16; ++dt) o[dt] = MFMA(htr<HD>(VI, off, pp, dt), pf, o[dt]);
        };
#pragma unroll 2
        for (int pp = 0; pp < pfull; ++pp) pair(pp, false);
#pragma unroll 1
        for (int pp = pfull; pp < nkp; ++pp) pair(pp, true);
        sum = red4_sum(sum);
        if (qi < a.Tq) {
            if (g == 0) a.lse[bh * a.Tq + qi] = (mxs + __log2f(sum)) * ATTN_LN2;
            const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
            for (int dt = 0; dt < HD / 16; ++dt) store4(ob + (long)qi * a.o_st + dt * 16 + 4 * g, o[dt], inv);
        }
        if (q0 + nw * 16 < a.Tq) load_row_frags<HD>(qf, qb, a.q_st, q0 + nw * 16, a.Tq, lane);
    }
}

template <int HD, int FLAGS>
__global__ __launch_bounds__(256, (FLAGS == 0 && HD == 32) ? 5 : (FLAGS == 0 && HD == 64) ? 4 : 2) void attn_head_bwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RB = HeadCfg<HD>::RB;
    const int nthr = blockDim.x, nw = nthr >> 6;
    const int nkp = (((a.Tk + 15) >> 4) + 1) >> 1, nqp = (((a.Tq + 15) >> 4) + 1) >> 1, prow = max(nkp, nqp) * 32;
    const int pfull = FLAGS ? 0 : a.Tk >> 5;          // pairs of key tiles with no padded key
    unsigned char* XI = smem;                 // phase 1: K     phase 2: Q
    unsigned char* YI = smem + prow * RB;     // phase 1: V     phase 2: dO
    float* KB = reinterpret_cast<float*>(smem + 2 * prow * RB);   // key bias, prow entries
    float* LS = KB + prow;                                        // lse in base-2 units (+1e30 past Tq: p = 0)
    float* DL = LS + prow;                                        // delta = rowsum(dO * O)
    unsigned char* MB = reinterpret_cast<unsigned char*>(DL + prow);   // phase 2 (saved dropout bits only): [g][pp][row] bytes, 32 * prow
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const long bh = blockIdx.x;
    const bool use_bits = FLAGS && a.drop_p > 0.f && a.drop_bits != nullptr;   // the forward pass left the keep-mask as bits
    const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_sb + h * a.q_sh;
    const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + b * a.k_sb + h * a.k_sh;
    const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + b * a.v_sb + h * a.v_sh;
    const bf16_t* ob = reinterpret_cast<const bf16_t*>(a.o) + b * a.o_sb + h * a.o_sh;
    const bf16_t* gb = reinterpret_cast<const bf16_t*>(a.dout) + b * a.do_sb + h * a.do_sh;
    bf16_t* dqb = reinterpret_cast<bf16_t*>(a.dq) + b * a.dq_sb + h * a.dq_sh;
    bf16_t* dkb = reinterpret_cast<bf16_t*>(a.dk) + b * a.dk_sb + h * a.dk_sh;
    bf16_t* dvb = reinterpret_cast<bf16_t*>(a.dv) + b * a.dv_sb + h * a.dv_sh;
    const float sc2 = a.scale * ATTN_LOG2E;
    const float inv_keep = a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
    const HeadOff<HD> off(li, g);

    ATTN_STAMP(0);
    // ---- phase 1: dQ (+ lse, delta -> LDS) -------------------------------------------------------------------------------------
    // the fragments a wave owns come from global memory; the first block's are requested before the staging loads
    bf16x8 qf[HD / 32], gf[HD / 32], of[HD / 32];
    float lse_c = 0.f;
    uint2 mw = make_uint2(0u, 0u);   // this lane's mask bytes of its query row: byte pp = key tiles 2pp (low nibble) and 2pp + 1
    auto load_q = [&](int q0, bf16x8 (&q_)[HD / 32], bf16x8 (&g_)[HD / 32], bf16x8 (&o_)[HD / 32], float& l_) {
        load_row_frags<HD>(q_, qb, a.q_st, q0, a.Tq, lane);
        load_row_frags<HD>(g_, gb, a.do_st, q0, a.Tq, lane);
        load_row_frags<HD>(o_, ob, a.o_st, q0, a.Tq, lane);
        l_ = q0 + li < a.Tq ? a.lse[bh * a.Tq + q0 + li] : 0.f;
        if (use_bits && q0 + li < a.Tq) mw = *reinterpret_cast<const uint2*>(a.drop_bits + (((long)bh * a.Tq + q0 + li) * 4 + g) * 8);
    };
    if (wave * 16 < a.Tq) load_q(wave * 16, qf, gf, of, lse_c);
    head_stage_pair<HD, (HD == 128 ? 8 : 4)>(XI, kb, a.k_st, YI, vb, a.v_st, a.Tk, prow, tid, nthr);
    head_key_bias(KB, a, b, prow, tid, nthr);
    for (int i = ((a.Tq + 15) & ~15) + tid; i < prow; i += nthr) { LS[i] = 1e30f; DL[i] = 0.f; }
    __syncthreads();
    ATTN_STAMP(1);
    for (int q0 = wave * 16; q0 < a.Tq; q0 += nw * 16) {
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += bf2f((bf16_t)gf[ks][e]) * bf2f((bf16_t)of[ks][e]);
        dl = red4_sum(dl);
        const int qi = q0 + li;
        const bool qok = qi < a.Tq;
        const float lse2 = qok ? lse_c * ATTN_LOG2E : 1e30f;
        if (g == 0) {
            LS[qi] = lse2;
            DL[qi] = qok ? dl : 0.f;
            if (qok && a.delta) a.delta[bh * a.Tq + qi] = dl;
        }
        f32x4 dq[HD / 16];  // dq[dt][r] = dQ[i = q0+li][d = dt*16 + 4g + r]
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // the arithmetic between the score MFMAs and the dQ MFMAs of one 16-key tile: e = s * scale*log2e (+ key bias) - lse;
        // ds = 2^e * (dp (* dropout) - delta), two scores per VALU instruction -> two packed words of the pair's dS operand
        auto soft = [&](int pp, int hf, bool biased, const f32x4& s, const f32x4& dp, uint32_t& w0, uint32_t& w1) __attribute__((always_inline)) {
            const int t = 2 * pp + hf;   // (an odd tile count: the last tile is all padding -- zero rows, biased keys)
            f32x2 e0 = {s[0], s[1]}, e1 = {s[2], s[3]}, d0 = {dp[0], dp[1]}, d1 = {dp[2], dp[3]};
            const f32x2 sc = {sc2, sc2}, nl = {-lse2, -lse2}, dlv = {dl, dl};
            if (biased) {
                const f32x4 kb4 = *reinterpret_cast<const f32x4*>(KB + t * 16 + 4 * g);
                e0 = e0 * sc + ((f32x2){kb4[0], kb4[1]} + nl);
                e1 = e1 * sc + ((f32x2){kb4[2], kb4[3]} + nl);
            } else {
                e0 = e0 * sc + nl;
                e1 = e1 * sc + nl;
            }
            if (FLAGS && a.drop_p > 0.f) {
                float dm[4];
                if (use_bits) {
                    const unsigned nib = ((pp < 4 ? mw.x : mw.y) >> (8 * (pp & 3) + 4 * hf)) & 15u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dm[r] = (nib >> r) & 1u ? inv_keep : 0.f;
                } else {
                    attn_drop4(a, (uint64_t)bh * a.Tq + qi, t * 16 + 4 * g, inv_keep, dm);
                }
                d0 *= (f32x2){dm[0], dm[1]};
                d1 *= (f32x2){dm[2], dm[3]};
            }
            w0 = hpk(hexp2(e0) * (d0 - dlv));
            w1 = hpk(hexp2(e1) * (d1 - dlv));
        };
        auto pair = [&](int pp, bool biased) {
            uint32_t dsp[4];
            {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int t = 2 * pp + hf;
                    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < HD / 32; ++ks) {
                        s = MFMA(hfr<HD>(XI, off, t, ks), qf[ks], s);
                        dp = MFMA(hfr<HD>(YI, off, t, ks), gf[ks], dp);
                    }
                    soft(pp, hf, biased, s, dp, dsp[2 * hf], dsp[2 * hf + 1]);
                }
                const bf16x8 sf = __builtin_bit_cast(bf16x8, make_uint4(dsp[0], dsp[1], dsp[2], dsp[3]));
#pragma unroll
                for (int dt = 0; dt < HD / 16; ++dt) dq[dt] = MFMA(htr<HD>(XI, off, pp, dt), sf, dq[dt]);
            }
        };
#pragma unroll 2
        for (int pp = 0; pp < pfull; ++pp) pair(pp, false);
#pragma unroll 1
        for (int pp = pfull; pp < nkp; ++pp) pair(pp, true);
        if (qok) {
#pragma unroll
            for (int dt = 0; dt < HD / 16; ++dt) store4(dqb + (long)qi * a.dq_st + dt * 16 + 4 * g, dq[dt], a.scale);
        }
        if (q0 + nw * 16 < a.Tq) load_q(q0 + nw * 16, qf, gf, of, lse_c);
    }
    ATTN_STAMP(2);
    // ---- phase 2: dK, dV ---------------------------------------------------------------------------------------------------------
    bf16x8 kf[HD / 32], vf[HD / 32];
    if (wave * 16 < a.Tk) {
        load_row_frags<HD>(kf, kb, a.k_st, wave * 16, a.Tk, lane);
        load_row_frags<HD>(vf, vb, a.v_st, wave * 16, a.Tk, lane);
    }
    __syncthreads();
    ATTN_STAMP(3);
    head_stage_pair<HD, (HD == 128 ? 8 : 4)>(XI, qb, a.q_st, YI, gb, a.do_st, a.Tq, prow, tid, nthr);
    if (use_bits) {   // the head's mask bytes, transposed so that the four query rows of a lane are four consecutive bytes
        for (int i = tid; i < a.Tq * 4; i += nthr) {
            const int row = i >> 2, gg = i & 3;
            const uint2 w = *reinterpret_cast<const uint2*>(a.drop_bits + ((long)bh * a.Tq * 4 + i) * 8);
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) MB[(gg * 8 + pp) * prow + row] = (unsigned char)(((pp < 4 ? w.x : w.y) >> (8 * (pp & 3))) & 255u);
        }
    }
    __syncthreads();
    ATTN_STAMP(4);
    for (int j0 = wave * 16; j0 < a.Tk; j0 += nw * 16) {
        const int kj = j0 + li;
        const bool jok = KB[kj] == 0.f;
        const f32x2 kb2 = {KB[kj], KB[kj]};     // 0 or -1e30
        // this lane's key in the forward pass's byte layout: tile pair, 4-key group, bit position
        const unsigned char* mcol = MB + ((((kj & 15) >> 2) * 8 + (kj >> 5)) * prow);
        const int mbit = 4 * ((kj >> 4) & 1) + (kj & 3);
        f32x4 dk[HD / 16], dv[HD / 16];   // [dt][r] = dK / dV[j = j0+li][d = dt*16 + 4g + r]
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) dk[dt] = dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int ip = 0; ip < nqp; ++ip) {
            uint32_t pdp[4], dsp[4];
            // one 16-query tile of the pair: p = 2^(s*scale*log2e - lse), P (with the dropout scale) and dS = p * (dp (* dropout) - delta)
            auto soft2 = [&](int hf, const f32x4& s, const f32x4& dp) __attribute__((always_inline)) {
                const int it = 2 * ip + hf;   // (an odd tile count: the last tile is all padding -- zero rows, lse = +1e30)
                const f32x4 lse4 = *reinterpret_cast<const f32x4*>(LS + it * 16 + 4 * g);
                const f32x4 dl4 = *reinterpret_cast<const f32x4*>(DL + it * 16 + 4 * g);
                const f32x2 sc = {sc2, sc2};
                f32x2 d0 = {dp[0], dp[1]}, d1 = {dp[2], dp[3]};
                // (a masked key: its score is NOT part of lse, so 2^(s - lse) can be anything -- 1e6 was seen 230 steps into a run; in IEEE half
                // that is inf, inf x 0 in the dV accumulation is NaN, and the "x 0" at the store below keeps it NaN: the key's bias goes into the
                // exponent, as in phase 1, and the probability is exactly 0)
                f32x2 e0 = (f32x2){s[0], s[1]} * sc - (f32x2){lse4[0], lse4[1]}, e1 = (f32x2){s[2], s[3]} * sc - (f32x2){lse4[2], lse4[3]};
                if (FLAGS) { e0 += kb2; e1 += kb2; }
                f32x2 p0 = hexp2(e0);
                f32x2 p1 = hexp2(e1);
                if (FLAGS && a.drop_p > 0.f) {
                    float dm[4];
                    if (use_bits) {
                        const unsigned m4 = *reinterpret_cast<const unsigned*>(mcol + it * 16 + 4 * g) >> mbit;
#pragma unroll
                        for (int r = 0; r < 4; ++r) dm[r] = (m4 >> (8 * r)) & 1u ? inv_keep : 0.f;
                    } else {
                        attn_drop4_col(a, (uint64_t)bh * a.Tq + (it * 16 + 4 * g), kj, li, inv_keep, dm);
                    }
                    const f32x2 m0 = {dm[0], dm[1]}, m1 = {dm[2], dm[3]};
                    d0 *= m0; d1 *= m1;
                    pdp[2 * hf] = hpk(p0 * m0);
                    pdp[2 * hf + 1] = hpk(p1 * m1);
                } else {
                    pdp[2 * hf] = hpk(p0);
                    pdp[2 * hf + 1] = hpk(p1);
                }
                dsp[2 * hf] = hpk(p0 * (d0 - (f32x2){dl4[0], dl4[1]}));
                dsp[2 * hf + 1] = hpk(p1 * (d1 - (f32x2){dl4[2], dl4[3]}));
            };
            {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int it = 2 * ip + hf;
                    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < HD / 32; ++ks) {
                        s = MFMA(hfr<HD>(XI, off, it, ks), kf[ks], s);
                        dp = MFMA(hfr<HD>(YI, off, it, ks), vf[ks], dp);
                    }
                    soft2(hf, s, dp);
                }
                const bf16x8 pf = __builtin_bit_cast(bf16x8, make_uint4(pdp[0], pdp[1], pdp[2], pdp[3]));
                const bf16x8 sf = __builtin_bit_cast(bf16x8, make_uint4(dsp[0], dsp[1], dsp[2], dsp[3]));
#pragma unroll
                for (int dt = 0; dt < HD / 16; ++dt) {
                    dv[dt] = MFMA(htr<HD>(YI, off, ip, dt), pf, dv[dt]);
                    dk[dt] = MFMA(htr<HD>(XI, off, ip, dt), sf, dk[dt]);
                }
            }
        }
        if (kj < a.Tk) {
            const float mk = jok ? a.scale : 0.f, mv = jok ? 1.0f : 0.f;   // a masked key got probability 0 in the forward pass
#pragma unroll
            for (int dt = 0; dt < HD / 16; ++dt) {
                store4(dkb + (long)kj * a.dk_st + dt * 16 + 4 * g, dk[dt], mk);
                store4(dvb + (long)kj * a.dv_st + dt * 16 + 4 * g, dv[dt], mv);
            }
        }
        if (j0 + nw * 16 < a.Tk) {
            load_row_frags<HD>(kf, kb, a.k_st, j0 + nw * 16, a.Tk, lane);
            load_row_frags<HD>(vf, vb, a.v_st, j0 + nw * 16, a.Tk, lane);
        }
    }
    ATTN_STAMP(5);
}

template <typename K>
static void lds_optin(K kern, size_t bytes) {
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
#define LAUNCH_R(KERN, GRID, SHM, ST, ARGS)                  \
    do {                                                     \
        static bool once_ = false;                           \
        if (!once_) { lds_optin(KERN, SHM); once_ = true; }  \
        hipLaunchKernelGGL(KERN, GRID, dim3(256), SHM, ST, ARGS); \
    } while (0)

// =============================================================================================
// (the opt-in is for the largest size any later call of the same instantiation may ask for, not for this call's)
#define LAUNCH_H(KERN, GRID, BLOCK, SHM, ST, ARGS)           \
    do {                                                     \
        static bool once_ = false;                           \
        if (!once_) { lds_optin(KERN, LDS_MAX); once_ = true; }  \
        hipLaunchKernelGGL(KERN, GRID, dim3(BLOCK), SHM, ST, ARGS); \
    } while (0)
#define LDS_MAX (160 * 1024)
// ECAMP_ATTN_HEAD=0 keeps the 64-row streaming kernels (A/B measurements)
static long g_head_launches = 0;   // development ABI: launches of the head-resident kernels so far (tests assert that they really ran)
extern "C" int64_t ecamp_attn_head_launches(void) { return g_head_launches; }
static int g_head_mode = -1;   // ecamp_set_option("attn_head", v): -1 = the environment decides
void attn_set_head_mode(int on) { g_head_mode = on < 0 ? -1 : (on ? 1 : 0); }
static bool head_enabled() {
    static const int env = [] { const char* e = getenv("ECAMP_ATTN_HEAD"); return e ? atoi(e) : 1; }();
    return (g_head_mode >= 0 ? g_head_mode : env) != 0;
}
// workgroup size: one wave per 16-row tile up to `cap` waves (ECAMP_ATTN_WAVES overrides the cap: tuning)
static int head_waves(int tiles, int cap) {
    static const int env = [] { const char* e = getenv("ECAMP_ATTN_WAVES"); return e ? atoi(e) : 0; }();
    if (env > 0) cap = env;
    return tiles < cap ? tiles : cap;
}
template <int HD>
static bool head_fwd(const AttnArgs& a, hipStream_t st) {
    const int nkp = ((a.Tk + 15) / 16 + 1) / 2, prow = nkp * 32;
    const size_t shm = (size_t)2 * prow * HeadCfg<HD>::RB + (size_t)prow * sizeof(float);
    if (!head_enabled() || shm > LDS_MAX) return false;
    const dim3 grid(a.B * a.H);
    const int nw = head_waves((a.Tq + 15) / 16, 8);
    if (a.key_mask != nullptr || a.drop_p > 0.f) LAUNCH_H((attn_head_fwd_kernel<HD, 1>), grid, nw * 64, shm, st, a);
    else LAUNCH_H((attn_head_fwd_kernel<HD, 0>), grid, nw * 64, shm, st, a);
    ++g_head_launches;
    return true;
}
template <int HD>
static bool head_bwd(const AttnArgs& a, hipStream_t st) {
    const int nkt = (a.Tk + 15) / 16, nqt = (a.Tq + 15) / 16, nkp = (nkt + 1) / 2, nqp = (nqt + 1) / 2, prow = (nkp > nqp ? nkp : nqp) * 32;
    const bool bits = a.drop_bits != nullptr && a.drop_p > 0.f;
    const size_t shm = (size_t)2 * prow * HeadCfg<HD>::RB + (size_t)3 * prow * sizeof(float) + (bits ? (size_t)32 * prow : 0);
    if (!head_enabled() || shm > LDS_MAX) return false;
    const dim3 grid(a.B * a.H);
    // four waves: two (hd = 128: the register file) to five workgroups share a CU and one's staging overlaps another's tile loops
    // (T = 197, hd = 32: 230 us against 260 us with eight; T = 128, hd = 128: 147 against 161)
    const int nw = head_waves(nkt > nqt ? nkt : nqt, 4);
    if (a.key_mask != nullptr || a.drop_p > 0.f) LAUNCH_H((attn_head_bwd_kernel<HD, 1>), grid, nw * 64, shm, st, a);
    else LAUNCH_H((attn_head_bwd_kernel<HD, 0>), grid, nw * 64, shm, st, a);
    ++g_head_launches;
    return true;
}
template <int HD>
static void fwd16(const AttnArgs& a, hipStream_t st) {
    if (head_fwd<HD>(a, st)) return;
    // Tk > 128: the all-scores-in-registers kernels need 64 score registers per lane and spill (272 B/lane of scratch at 4 chunks);
    // the online-softmax kernel keeps one chunk of scores live and is faster from 129 keys up (decoder T=197, S=256, ViT-L T=785)
    if (a.Tk > 128) {
        dim3 grid(ceil_div(a.Tq, 64), a.B * a.H);
        hipLaunchKernelGGL((attn16_fwd_long_kernel<HD>), grid, dim3(256), (size_t)2 * TileCfg<HD>::BYTES + 4 * 2048 + 64 * sizeof(int), st, a);
        return;
    }
    {
        const int kch = a.Tk <= 64 ? 1 : a.Tk <= 128 ? 2 : 4;
        const size_t shm = (size_t)2 * kch * TileCfg<HD>::BYTES + 4 * 2048;
        if (shm <= LDS_MAX) {
            dim3 grid(a.B * a.H);
            if (kch == 1) LAUNCH_R((attn16r_fwd_kernel<HD, 1>), grid, shm, st, a);
            else if (kch == 2) LAUNCH_R((attn16r_fwd_kernel<HD, 2>), grid, shm, st, a);
            else LAUNCH_R((attn16r_fwd_kernel<HD, 4>), grid, shm, st, a);
            return;
        }
    }
    dim3 grid(ceil_div(a.Tq, 64), a.B * a.H), block(256);
    size_t shm = TileCfg<HD>::BYTES + 4 * 2048;
    if (a.Tk <= 64) hipLaunchKernelGGL((attn16_fwd_kernel<HD, 1>), grid, block, shm, st, a);
    else if (a.Tk <= 128) hipLaunchKernelGGL((attn16_fwd_kernel<HD, 2>), grid, block, shm, st, a);
    else hipLaunchKernelGGL((attn16_fwd_kernel<HD, 4>), grid, block, shm, st, a);
}
template <int HD>
static void bwd16(const AttnArgs& a, hipStream_t st) {
    // The backward kernels run as 64-row workgroups streaming 64-key (dQ) / 64-query (dK, dV) chunks.  "Resident" variants that
    // stage all of K/V (or Q/dO) once per (batch, head), like the forward above, were built and measured slower on MI355X: a quarter
    // of the workgroups and 2-3x the LDS footprint cost more occupancy than the saved barriers return (rejected).
    if (head_bwd<HD>(a, st)) return;
    dim3 block(256);
    dim3 grid(ceil_div(a.Tq, 64), a.B * a.H);
    size_t shm = 2 * TileCfg<HD>::BYTES + 4 * 2048 + 64 * sizeof(int);
    if (a.Tk <= 64) hipLaunchKernelGGL((attn16_bwd_dq_kernel<HD, 1>), grid, block, shm, st, a);
    else if (a.Tk <= 128) hipLaunchKernelGGL((attn16_bwd_dq_kernel<HD, 2>), grid, block, shm, st, a);
    else if (a.Tk <= 256) hipLaunchKernelGGL((attn16_bwd_dq_kernel<HD, 4>), grid, block, shm, st, a);
    else hipLaunchKernelGGL((attn16_bwd_dq_kernel<HD, 0>), grid, block, shm, st, a);
    dim3 grid2(ceil_div(a.Tk, 64), a.B * a.H);
    size_t shm2 = 2 * TileCfg<HD>::BYTES + 8 * 2048 + 128 * sizeof(float);
    hipLaunchKernelGGL((attn16_bwd_dkv_kernel<HD>), grid2, block, shm2, st, a);
}
bool attn_bf16_head_path(int Tq, int Tk, int hd, bool backward) {
    const int nkt = (Tk + 15) / 16, nqt = (Tq + 15) / 16, nkp = (nkt + 1) / 2, nqp = (nqt + 1) / 2;
    const size_t rb = hd == 32 ? HeadCfg<32>::RB : hd == 64 ? HeadCfg<64>::RB : HeadCfg<128>::RB;
    size_t shm;
    if (backward) { const size_t prow = (size_t)(nkp > nqp ? nkp : nqp) * 32; shm = 2 * prow * rb + 3 * prow * sizeof(float) + 32 * prow; }
    else { const size_t prow = (size_t)nkp * 32; shm = 2 * prow * rb + prow * sizeof(float); }
    return head_enabled() && shm <= LDS_MAX;
}
void attn_bf16_fwd(const AttnArgs& a, int hd, hipStream_t st) {
    if (hd == 32) fwd16<32>(a, st); else if (hd == 64) fwd16<64>(a, st); else fwd16<128>(a, st);
}
void attn_bf16_bwd(const AttnArgs& a, int hd, hipStream_t st) {
    if (hd == 32) bwd16<32>(a, st); else if (hd == 64) bwd16<64>(a, st); else bwd16<128>(a, st);
}
