// Shared device/host helpers for libecamp_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define ECAMP_F32 0
#define ECAMP_BF16 1

// The 16-bit storage format is a property of the BUILD: libecamp_hip.so keeps bfloat16 (the benchmarked mode), libecamp_hip_f16.so -- the same
// sources with -DECAMP_HALF_F16 -- keeps IEEE half, the format the reference's torch.cuda.amp.autocast() computes in (main_pretrain.py:139,
// engine_pretrain.py:44).  Everything that touches the bits goes through the helpers below (h16 <-> f32, the packed pair forms, the MFMA
// macros, H16_ONE / H16_NEG_INF); `bf16_t` is the historical name of "one raw 16-bit element" in either build, and dtype code ECAMP_BF16
// means "the library's 16-bit format" (ecamp_half_format() says which).
#ifdef ECAMP_HALF_F16
#define ECAMP_HALF_IS_F16 1
typedef _Float16 hw_h16;
#else
#define ECAMP_HALF_IS_F16 0
typedef __bf16 hw_h16;
#endif
typedef unsigned short bf16_t;  // raw 16-bit element (bfloat16 bits, or IEEE half bits in the f16 build)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;  // MFMA bf16 operand (8 x bf16 = 4 VGPRs)

extern thread_local char g_ecamp_err[512];
int ecamp_set_error(int code, const char* fmt, ...);

#define ECAMP_CHECK_ARG(cond, ...)                         \
    do {                                                   \
        if (!(cond)) return ecamp_set_error(-1, __VA_ARGS__); \
    } while (0)

#define ECAMP_LAUNCH_CHECK()                                                        \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) return ecamp_set_error((int)e_, "%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------------------------------------
typedef hw_h16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef hw_h16 hw_h16x4 __attribute__((ext_vector_type(4)));
typedef hw_h16 hw_h16x8 __attribute__((ext_vector_type(8)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
#if ECAMP_HALF_IS_F16
constexpr unsigned short H16_ONE = 0x3C00, H16_NEG_INF = 0xFC00;
#define ECAMP_DOT2C "v_dot2c_f32_f16"    // f32 += two 16-bit products (inline asm of the bias-gradient row sums, gemm_q8.h)
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// the two elements of a packed pair (low half first): v_cvt_f32_f16 reads either half of the register directly
__device__ __forceinline__ float h16_lo(uint32_t w) { return (float)__builtin_bit_cast(hw_bf16x2, w)[0]; }
__device__ __forceinline__ float h16_hi(uint32_t w) { return (float)__builtin_bit_cast(hw_bf16x2, w)[1]; }
#else
constexpr unsigned short H16_ONE = 0x3F80, H16_NEG_INF = 0xFF80;
#define ECAMP_DOT2C "v_dot2c_f32_bf16"
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float h16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float h16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
#endif
constexpr uint32_t H16_ONE_X2 = (uint32_t)H16_ONE * 0x10001u, H16_NEG_INF_X2 = (uint32_t)H16_NEG_INF * 0x10001u;
// f32 -> 16 bit, round-to-nearest-even: gfx950 has both conversions in hardware (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, two values per
// instruction); the integer emulation the bf16 one replaces cost ~8 VALU operations per value and dominated the GEMM epilogues
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (hw_h16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((hw_f32x2){lo, hi}, hw_bf16x2));
}
// The matrix instructions of the 16-bit format (operands: eight / four raw elements per lane, any 16- / 8-byte type)
#if ECAMP_HALF_IS_F16
#define ECAMP_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hw_h16x8, a), __builtin_bit_cast(hw_h16x8, b), c, 0, 0, 0)
#define ECAMP_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hw_h16x8, a), __builtin_bit_cast(hw_h16x8, b), c, 0, 0, 0)
#define ECAMP_MFMA_4x4x4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(hw_h16x4, a), __builtin_bit_cast(hw_h16x4, b), c, 0, 0, 0)
#else
#define ECAMP_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(hw_h16x8, a), __builtin_bit_cast(hw_h16x8, b), c, 0, 0, 0)
#define ECAMP_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hw_h16x8, a), __builtin_bit_cast(hw_h16x8, b), c, 0, 0, 0)
typedef short bf16x4_raw __attribute__((ext_vector_type(4)));
#define ECAMP_MFMA_4x4x4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(bf16x4_raw, a), __builtin_bit_cast(bf16x4_raw, b), c, 0, 0, 0)
#endif

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }
// value after a round trip through storage type T (used so fwd statistics match what bwd re-reads)
template <typename T> __device__ __forceinline__ float rnd(float v) { return to_f<T>(from_f<T>(v)); }

// 4-element vector load/store (16 B for f32, 8 B for bf16); pointers must be suitably aligned
template <typename T> __device__ __forceinline__ void ld4(const T* p, float (&o)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float (&o)[4]) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = h16_lo(v.x); o[1] = h16_hi(v.x);
    o[2] = h16_lo(v.y); o[3] = h16_hi(v.y);
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float (&o)[4]);
template <> __device__ __forceinline__ void st4<float>(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float (&o)[4]) {
    uint2 v;
    v.x = pack_bf16x2(o[0], o[1]);
    v.y = pack_bf16x2(o[2], o[3]);
    *reinterpret_cast<uint2*>(p) = v;
}

// ---------------------------------------------------------------------------------------------
// wave64 reductions (all 64 lanes end with the result)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block reduction for blockDim.x == 256 (4 waves); `sh` must hold >= 4 floats. All threads get the sum.
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---------------------------------------------------------------------------------------------
// Philox4x32 counter RNG (Salmon et al. 2011, "Parallel random numbers: as easy as 1, 2, 3"), SEVEN rounds: Philox4x32-7, the smallest
// round count of that paper that passes BigCrush (10, its default, is a safety margin; rounds 1-4 of this library ran 10).  One call
// yields 4 x 32 random bits for counter (idx_lo, idx_hi, offset_lo, offset_hi) under key `seed`.  Forward and backward regenerate
// identical masks from (seed, offset, element index): no mask tensor ever touches HBM.
// Why the round count matters: a round is two 32 x 32 -> 64-bit multiplies, quarter-rate VALU work -- at ten rounds and four elements per
// call the mask cost 25 full-rate instruction slots per element, 18 us of a 44 us LayerNorm + dropout call on [32768, 768] (31 us is
// what its bytes take) and as much again in every report-side attention.
constexpr int PHILOX_ROUNDS = 7;
__device__ __forceinline__ uint4 philox4x32(uint64_t seed, uint64_t offset, uint64_t idx) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
#pragma unroll
    for (int r = 0; r < PHILOX_ROUNDS; ++r) {
        // one 32x32 -> 64-bit multiply per product (v_mad_u64_u32) instead of v_mul_hi_u32 + v_mul_lo_u32: integer multiplies are the
        // slow VALU operations and Philox is most of the report-side attention's arithmetic
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
// THE dropout convention of this library (round 5; every kernel and ecamp_dropout_mask): element e of the flattened tensor is kept iff
// halfword (e & 7) of Philox(counter e >> 3) -- words x, y, z, w in turn, low half first -- is >= round(p * 65536).  EIGHT consecutive
// elements share one call (rounds 1-4: four, 24 bits each); the keep probability is 1 - round(65536 p) / 65536, within 8e-6 of 1 - p
// (p = 0.1: 0.8999939), and the survivors are scaled by the caller's 1 / (1 - p).
__device__ __forceinline__ uint32_t dropout_thr(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }
__device__ __forceinline__ float dropout_pick(uint32_t word, int half, uint32_t thr, float inv_keep) {
    return (half ? (word >> 16) : (word & 0xffffu)) >= thr ? inv_keep : 0.0f;
}
// keep-mask scale (0 or inv_keep) of element e
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t offset, uint64_t e, float p, float inv_keep) {
    const uint4 r = philox4x32(seed, offset, e >> 3);
    const int h = (int)(e & 7);
    const uint32_t w = (h >> 1) == 0 ? r.x : (h >> 1) == 1 ? r.y : (h >> 1) == 2 ? r.z : r.w;
    return dropout_pick(w, h & 1, dropout_thr(p), inv_keep);
}
// ... of the four elements 4 e4 .. 4 e4 + 3 (half a call: words x, y for even e4, z, w for odd)
__device__ __forceinline__ void dropout_scale4(uint64_t seed, uint64_t offset, uint64_t e4, float p, float inv_keep,
                                               float (&s)[4]) {
    const uint4 r = philox4x32(seed, offset, e4 >> 1);
    const uint32_t wa = (e4 & 1) ? r.z : r.x, wb = (e4 & 1) ? r.w : r.y, thr = dropout_thr(p);
    s[0] = dropout_pick(wa, 0, thr, inv_keep); s[1] = dropout_pick(wa, 1, thr, inv_keep);
    s[2] = dropout_pick(wb, 0, thr, inv_keep); s[3] = dropout_pick(wb, 1, thr, inv_keep);
}
// ... of the eight elements 8 e8 .. 8 e8 + 7 (one whole call)
__device__ __forceinline__ void dropout_scale8(uint64_t seed, uint64_t offset, uint64_t e8, float p, float inv_keep,
                                               float (&s)[8]) {
    const uint4 r = philox4x32(seed, offset, e8);
    const uint32_t thr = dropout_thr(p);
    s[0] = dropout_pick(r.x, 0, thr, inv_keep); s[1] = dropout_pick(r.x, 1, thr, inv_keep);
    s[2] = dropout_pick(r.y, 0, thr, inv_keep); s[3] = dropout_pick(r.y, 1, thr, inv_keep);
    s[4] = dropout_pick(r.z, 0, thr, inv_keep); s[5] = dropout_pick(r.z, 1, thr, inv_keep);
    s[6] = dropout_pick(r.w, 0, thr, inv_keep); s[7] = dropout_pick(r.w, 1, thr, inv_keep);
}

// exact (erf) GELU and its derivative -- nn.GELU() default in timm Mlp / HF "gelu"
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
// The same two functions for bf16 activations.  The library erff is two divergent polynomial branches (~36 VALU instructions per
// element, ~50 with the derivative's separate exp) and the GELU epilogues of the fc1 / fc2-dgrad GEMMs run it on 1.3 G elements per
// step with nothing to overlap; Abramowitz-Stegun 7.1.26 is branch-free, shares exp(-x^2/2) between erf and the density, and its
// 6e-7 absolute error (f32 arithmetic) is four orders below a bf16 ulp.  The f32 parity mode keeps erff.
__device__ __forceinline__ float erf_as_f(float z, float& e) {   // -> erf(z);  e = exp(-z*z)
    const float a = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    e = __builtin_amdgcn_exp2f(-1.4426950408889634f * a * a);
    return copysignf(fmaf(-p, e, 1.0f), z);
}
// Forward only needs erf, and Abramowitz-Stegun 7.1.28 gives erfc(z) = (1 + a1 z + ... + a6 z^6)^-16 (|error| <= 3e-7) with ONE
// quarter-rate instruction (v_rcp_f32) instead of two (7.1.26 needs v_rcp_f32 and v_exp_f32): the GELU epilogue of the fc1 GEMMs is VALU
// time with the matrix pipes idle (profiles/r04_q4_and_power.txt section 8), and the two transcendentals were 44 % of it.  The
// polynomial is in a = |x| with the 1/sqrt(2) folded into the coefficients, split into even and odd parts, and
// x * Phi(x) = x/2 + |x| * (1 - erfc(|x|/sqrt 2)) / 2 needs no sign select: everything but the two multiply-adds by |x| packs
// (v_pk_fma_f32 / v_pk_mul_f32 have no |x| modifier) -- ~12 issue slots per element instead of ~18.  The pair form exists because
// hipcc stops packing the scalar form on its own once the constants are fma literals.
// Over all 65 280 finite bf16 inputs: max |error| 7.0e-7 (7.1.26: 4.4e-7); 9 results with |y| > 1e-4 round to the other bf16
// neighbour than the exact function (7.1.26: 2).  Huge |x|: x*x = inf -> s = inf -> erfc = 0 -> x or 0, no NaN.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_fast_f2(f32x2_t x) {
    const f32x2_t x2 = x * x;
    f32x2_t E = __builtin_elementwise_fma((f32x2_t)(5.3829750000e-06f), x2, (f32x2_t)(3.8003575000e-05f));
    E = __builtin_elementwise_fma(E, x2, (f32x2_t)(2.1141006150e-02f));
    E = __builtin_elementwise_fma(E, x2, (f32x2_t)(1.0f));
    f32x2_t O = __builtin_elementwise_fma((f32x2_t)(4.8890635643e-05f), x2, (f32x2_t)(3.2776263241e-03f));
    O = __builtin_elementwise_fma(O, x2, (f32x2_t)(4.9867346967e-02f));
    f32x2_t s = {fmaf(fabsf(x[0]), O[0], E[0]), fmaf(fabsf(x[1]), O[1], E[1])};
    s *= s; s *= s; s *= s; s *= s;
    const f32x2_t r = {__builtin_amdgcn_rcpf(s[0]), __builtin_amdgcn_rcpf(s[1])};   // erfc(|x| / sqrt 2)
    const f32x2_t q = __builtin_elementwise_fma((f32x2_t)(-0.5f), r, (f32x2_t)(0.5f));
    const f32x2_t hx = x * 0.5f;
    return (f32x2_t){fmaf(fabsf(x[0]), q[0], hx[0]), fmaf(fabsf(x[1]), q[1], hx[1])};
}
__device__ __forceinline__ float gelu_fast_f(float x) {   // the same operations on one value (bit-identical to a lane of the pair form)
    const float x2 = x * x;
    const float E = fmaf(fmaf(fmaf(5.3829750000e-06f, x2, 3.8003575000e-05f), x2, 2.1141006150e-02f), x2, 1.0f);
    const float O = fmaf(fmaf(4.8890635643e-05f, x2, 3.2776263241e-03f), x2, 4.9867346967e-02f);
    float s = fmaf(fabsf(x), O, E);
    s *= s; s *= s; s *= s; s *= s;
    const float q = fmaf(-0.5f, __builtin_amdgcn_rcpf(s), 0.5f);
    return fmaf(fabsf(x), q, x * 0.5f);
}
// GELU and its derivative together (act == 2, "saved derivative": the forward epilogue stores gelu'(x) in place of the pre-activation x,
// and the data-gradient epilogue of the following layer multiplies by it -- no erf / exp at all in the backward pass).  y is
// bit-identical to gelu_fast_f2; Phi = 1/2 + sign(x) (1 - erfc(|x|/sqrt 2))/2 reuses its q; the density costs the one v_exp_f32.
__device__ __forceinline__ f32x2_t gelu_both_fast_f2(f32x2_t x, f32x2_t& g) {
    const f32x2_t x2 = x * x;
    f32x2_t E = __builtin_elementwise_fma((f32x2_t)(5.3829750000e-06f), x2, (f32x2_t)(3.8003575000e-05f));
    E = __builtin_elementwise_fma(E, x2, (f32x2_t)(2.1141006150e-02f));
    E = __builtin_elementwise_fma(E, x2, (f32x2_t)(1.0f));
    f32x2_t O = __builtin_elementwise_fma((f32x2_t)(4.8890635643e-05f), x2, (f32x2_t)(3.2776263241e-03f));
    O = __builtin_elementwise_fma(O, x2, (f32x2_t)(4.9867346967e-02f));
    f32x2_t s = {fmaf(fabsf(x[0]), O[0], E[0]), fmaf(fabsf(x[1]), O[1], E[1])};
    s *= s; s *= s; s *= s; s *= s;
    const f32x2_t r = {__builtin_amdgcn_rcpf(s[0]), __builtin_amdgcn_rcpf(s[1])};
    const f32x2_t q = __builtin_elementwise_fma((f32x2_t)(-0.5f), r, (f32x2_t)(0.5f));
    const f32x2_t hx = x * 0.5f;
    const f32x2_t m = x2 * -0.72134752044448170368f;                                 // -x^2/2 * log2(e)
    const f32x2_t e = {__builtin_amdgcn_exp2f(m[0]), __builtin_amdgcn_exp2f(m[1])};
    const f32x2_t phi = {0.5f + copysignf(q[0], x[0]), 0.5f + copysignf(q[1], x[1])};
    g = __builtin_elementwise_fma(x * 0.39894228040143267794f, e, phi);
    return (f32x2_t){fmaf(fabsf(x[0]), q[0], hx[0]), fmaf(fabsf(x[1]), q[1], hx[1])};
}
__device__ __forceinline__ float gelu_both_fast_f(float x, float& g) {   // the same operations on one value
    f32x2_t g2;
    const f32x2_t y = gelu_both_fast_f2((f32x2_t){x, x}, g2);
    g = g2[0];
    return y[0];
}
__device__ __forceinline__ float gelu_grad_fast_f(float x) {
    float e;
    const float cdf = 0.5f * (1.0f + erf_as_f(x * 0.70710678118654752440f, e));
    return fmaf(x * 0.39894228040143267794f, e, cdf);
}
// The same operations in the same order on two values (bit-identical to the scalar form), written out so that everything except
// |z|, the sign transfer and the two quarter-rate instructions is a packed instruction (the gelu' epilogue of the fc2 data gradient).
__device__ __forceinline__ f32x2_t gelu_grad_fast_f2(f32x2_t x) {
    const f32x2_t z = x * 0.70710678118654752440f;
    const f32x2_t a = {fabsf(z[0]), fabsf(z[1])};
    const f32x2_t d = __builtin_elementwise_fma((f32x2_t)(0.3275911f), a, (f32x2_t)(1.0f));
    const f32x2_t t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2_t p = __builtin_elementwise_fma((f32x2_t)(1.061405429f), t, (f32x2_t)(-1.453152027f));
    p = __builtin_elementwise_fma(p, t, (f32x2_t)(1.421413741f));
    p = __builtin_elementwise_fma(p, t, (f32x2_t)(-0.284496736f));
    p = __builtin_elementwise_fma(p, t, (f32x2_t)(0.254829592f));
    p *= t;
    const f32x2_t m = (a * -1.4426950408889634f) * a;
    const f32x2_t e = {__builtin_amdgcn_exp2f(m[0]), __builtin_amdgcn_exp2f(m[1])};
    const f32x2_t er = __builtin_elementwise_fma(-p, e, (f32x2_t)(1.0f));                    // erf(|z|)
    const f32x2_t ers = {copysignf(er[0], z[0]), copysignf(er[1], z[1])};
    const f32x2_t cdf = (ers + 1.0f) * 0.5f;
    return __builtin_elementwise_fma(x * 0.39894228040143267794f, e, cdf);
}
template <typename T> __device__ __forceinline__ float gelu_t(float x);
template <> __device__ __forceinline__ float gelu_t<float>(float x) { return gelu_f(x); }
template <> __device__ __forceinline__ float gelu_t<bf16_t>(float x) { return gelu_fast_f(x); }
template <typename T> __device__ __forceinline__ float gelu_grad_t(float x);
template <> __device__ __forceinline__ float gelu_grad_t<float>(float x) { return gelu_grad_f(x); }
template <> __device__ __forceinline__ float gelu_grad_t<bf16_t>(float x) { return gelu_grad_fast_f(x); }

// profile.hip
int ecamp_prof_active();
void ecamp_prof_begin(int cat, double work, hipStream_t s, const char* tag = nullptr);
void ecamp_prof_end(hipStream_t s);
#define ECAMP_PROF_GEMM_BF16 0
#define ECAMP_PROF_GEMM_F32 1
#define ECAMP_PROF_ATTN 2
#define ECAMP_PROF_GEMM_FP8 3

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
