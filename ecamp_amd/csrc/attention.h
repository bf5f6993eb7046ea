// Shared argument block of the attention kernels (attention.hip: exact-f32 MFMA path; attention_bf16.hip: bf16 MFMA path).
#pragma once
#include "common.h"

struct AttnArgs {
    const void* q; const void* k; const void* v; void* o;
    const void* dout; void* dq; void* dk; void* dv;
    float* lse; float* delta;
    const int32_t* key_mask;  // [B, Tk], nonzero = attend; or null
    long q_sb, q_st, q_sh, k_sb, k_st, k_sh, v_sb, v_st, v_sh, o_sb, o_st, o_sh;
    long dq_sb, dq_st, dq_sh, dk_sb, dk_st, dk_sh, dv_sb, dv_st, dv_sh, do_sb, do_st, do_sh;
    int B, H, Tq, Tk;
    float scale, drop_p;
    uint64_t seed, offset;
    // Dropout keep-mask of the attention probabilities as BITS, written by the head-resident forward kernel and read by its backward
    // (one Philox evaluation per score instead of three); null: every kernel regenerates the mask from the Philox counters.
    // Layout [B*H][Tq][4][8] bytes: byte (row, g, pp) = keys pp*32 + 4g + r (bits r = 0..3) and pp*32 + 16 + 4g + r (bits 4 + r).
    unsigned char* drop_bits;
};

#define NEG_BIG (-1.0e30f)

void attn_bf16_fwd(const AttnArgs& a, int hd, hipStream_t st);
void attn_set_head_mode(int on);   // ecamp_set_option("attn_head", ...): -1 back to the environment's choice
void attn_bf16_bwd(const AttnArgs& a, int hd, hipStream_t st);
bool attn_bf16_head_path(int Tq, int Tk, int hd, bool backward);   // would the head-resident kernels serve this shape (LDS, option)?
