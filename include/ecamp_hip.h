/* libecamp_hip.so -- C ABI of the MI355X-native ECAMP pre-training hot path.
 *
 * The reference (ToniChopp/ECAMP) is 100 % Python: it has NO native interface to mirror.
 * Each entry point below therefore cites the reference *Python call site* whose ATen/cuDNN/cuBLAS kernels it
 * replaces (paths relative to ECAMP/Pre-training/).  Conventions:
 *   - raw device pointers + explicit sizes / element strides; dtype enum 0 = f32, 1 = bf16 (storage of activations);
 *     parameters, LayerNorm statistics, losses and all parameter gradients are always f32
 *   - every call only ENQUEUES work on `stream` (never synchronises, never allocates, never frees)
 *   - returns 0 on success, <0 on argument errors, >0 = hipError_t; message via ecamp_last_error() (thread-local)
 *   - parameter-gradient outputs ACCUMULATE (+=) into the caller's f32 buffers (gradient accumulation,
 *     main_pretrain.py:147-153); activation-gradient outputs are overwritten
 *   - dropout is a Philox4x32-7 stream keyed by (seed, offset, element index >> 3; one 16-bit draw per element): backward regenerates the mask
 */
#ifndef ECAMP_HIP_H
#define ECAMP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* ecampStream_t; /* == hipStream_t */

#define ECAMP_F32 0
#define ECAMP_BF16 1

/* Bumped whenever an exported signature changes (2: ecamp_wgrad_group gained table_bytes, ecamp_gemm_fp8 its q8_* arguments;
 * 3: round 5 -- ecamp_dropout_mask, ecamp_prof_dump, ecamp_fp8_roll's amax history and margin; 4: round 6 -- ecamp_adamw_grouped's
 * `ctl`, ecamp_loss_scale_update, ecamp_half_format, the resample entry points).  ecamp_abi_version()
 * returns the value the library was BUILT with; a consumer compares it with the header it was compiled against -- the Python binding
 * (ecamp_amd/_lib.py) refuses a library of another version, which is what protects an A/B of two builds (ECAMP_LIB, tools/ab_lib.sh)
 * from calling an older build with a newer argument list. */
#define ECAMP_ABI_VERSION 4
int ecamp_abi_version(void);
const char* ecamp_last_error(void);
/* The 16-bit activation format of THIS build -- what dtype code ECAMP_BF16 (1) stores.  0: bfloat16 (libecamp_hip.so, the benchmarked
 * mode).  1: IEEE half (libecamp_hip_f16.so: the same sources compiled with -DECAMP_HALF_F16), the format the reference's
 * torch.cuda.amp.autocast() computes its linear layers and attention products in (main_pretrain.py:139, engine_pretrain.py:44); it is
 * run with dynamic loss scaling (util/misc.py:251-271).  Accumulation, LayerNorm / softmax statistics, losses, parameters and their
 * gradients are f32 in both.  The e4m3 forward (ecamp_gemm_fp8 and its producers) exists in the bfloat16 build only. */
int ecamp_half_format(void);

/* ---- dense contractions -------------------------------------------------------------------------------------
 * C[M,N] (+)= epi( alpha * (alpha_dev ? *alpha_dev : 1) * sum_k opA[m,k] opB[k,n] ); a_kc/b_kc = operand is contiguous along the contraction.
 * Replaces every nn.Linear / Conv2d-as-GEMM on the path: timm PatchEmbed.proj (model_ecamp.py:60,220), Block
 * qkv/proj/fc1/fc2 (:66-68,80-82,233-234,254-255), decoder_embed/pred (:74,85,242,259), bert_mlp (:99,268), HF
 * Bert* dense layers (bert_modeling.py:113-131, context_fusion.py:32-72), MLM decoder (bert_modeling.py:209), and
 * their autograd dgrad/wgrad.  Epilogue: +bias[n]; save pre-activation; exact-erf GELU; *gelu'(gmul[m,n]);
 * +residual[m,n].  act: 0 none; 1 GELU (pre_out, if given, receives the pre-activation; gmul holds a pre-activation and the
 * result is multiplied by gelu'(gmul)); 2 (bf16 only) GELU with the SAVED DERIVATIVE: pre_out receives gelu'(pre-activation)
 * and, in the data-gradient call, gmul holds that derivative and multiplies the result as it is -- torch's GeluBackward
 * (nn.GELU at timm Mlp.act / HF BertIntermediate.intermediate_act_fn) without erf / exp in the backward pass, at the price of
 * one more bf16 rounding of the derivative.  out_f32/accumulate: f32 output added into C (weight gradients).  split_k > 1: the contraction is cut
 * into slabs written to `splitk_ws` (split_k*M*N floats) and combined by a deterministic reduce kernel (no atomics).
 * rowsum (optional, f32 [M]): rowsum[m] += alpha * sum_k opA[m,k] -- the bias gradient, computed inside the wgrad GEMM from the
 * M-side fragments it already holds (v_dot2c_f32_bf16 sums placed between the MFMAs) instead of a separate pass over dY. */
int ecamp_gemm(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int a_kc, int64_t lda, int b_kc,
               int64_t ldb, int64_t ldc, const float* bias, const void* residual, int64_t ldr, void* pre_out, int64_t ldp,
               const void* gmul, int64_t ldg, int act, float alpha, const float* alpha_dev, int dtype, int out_f32, int accumulate,
               int split_k, float* splitk_ws, float* rowsum, ecampStream_t stream);
/* Split count the library recommends for ecamp_gemm(M, N, K, ...) with an f32 accumulated output (the weight-gradient call of
 * torch.autograd for nn.Linear, e.g. timm Mlp.fc1 at model_ecamp.py:233): it depends on which kernel the shape selects
 * (128^2 tiles, or the persistent 256^2 kernel whose work items should fill whole rounds of the chip).  Pure host arithmetic. */
int ecamp_gemm_suggest_split(int64_t M, int64_t N, int64_t K, int a_kc, int b_kc, int dtype);
/* Grouped weight gradients (no reference counterpart: the reference's autograd issues one addmm per nn.Linear; timm Block at
 * model_ecamp.py:66-68,233-234, BertLayer at bert_layers.py): gw[p] [n_out[p], k_in[p]] (f32, contiguous) (+)= alpha * dy[p]^T x[p] and
 * gb[p] [n_out[p]] (f32, may be null) += alpha * column sums of dy[p], for the 1-4 linear layers of one block that share the row count
 * `rows` of dy[p] [rows, n_out[p]] / x[p] [rows, k_in[p]] (bf16, row-contiguous), as ONE persistent launch + one reduce.
 * accumulate[p] = 0 overwrites gw[p].  ws: ecamp_wgrad_group_workspace_bytes(...) bytes.  workgroups: 0 = three quarters of the CUs
 * (what is left runs whatever is queued beside it), otherwise at most one per CU.  ecamp_wgrad_group_supported == 0: issue per-layer ecamp_gemm calls.
 * table: DEVICE copy (32-byte aligned) of the group's item table -- pure host arithmetic on (shapes, has_bias, rows, workgroup count)
 * that ecamp_wgrad_group_table writes into HOST memory of ecamp_wgrad_group_table_bytes(...) bytes; the caller uploads it once per
 * shape set and keeps it (the library never allocates and never synchronises: the call is two kernel launches, capturable into a
 * HIP graph).  ecamp_wgrad_group_workgroups(w) = the workgroup count the launch uses for argument w (part of the table's identity).
 * table_bytes = what ecamp_wgrad_group_table returned for that image: a table built for another plan (the CU reserve or the workgroup
 * count changed in between) is refused instead of read at the wrong offsets. */
int ecamp_wgrad_group_supported(int32_t n, const int64_t* n_out, const int64_t* k_in, int64_t rows);
int64_t ecamp_wgrad_group_workspace_bytes(int32_t n, const int64_t* n_out, const int64_t* k_in, int64_t rows);
int64_t ecamp_wgrad_group_table_bytes(int32_t n, const int64_t* n_out, const int64_t* k_in, int64_t rows);
int64_t ecamp_wgrad_group_table(int32_t n, const int64_t* n_out, const int64_t* k_in, const int32_t* has_bias, int64_t rows,
                                int32_t workgroups, void* host_table);
int ecamp_wgrad_group_workgroups(int32_t workgroups);
int ecamp_wgrad_group(int32_t n, const void* const* dy, const void* const* x, float* const* gw, float* const* gb, const int64_t* n_out,
                      const int64_t* k_in, int64_t rows, float alpha, const float* alpha_dev, const int32_t* accumulate, float* ws,
                      const void* table, int64_t table_bytes, int32_t workgroups, ecampStream_t stream);
/* Workspace sizes (bytes) the caller allocates and passes in -- the library never allocates:
 *   ecamp_gemm_workspace_bytes      `splitk_ws` of ecamp_gemm for this split count (0 when split_k <= 1)
 *   ecamp_attn_bwd_workspace_bytes  `delta_ws` of ecamp_attn_bwd (one f32 per query row)
 *   ecamp_sr_bwd_workspace_bytes    `gw_ws` of ecamp_sr_bwd (168 f32, zeroed by the caller) */
int64_t ecamp_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int32_t split_k);
int64_t ecamp_attn_bwd_workspace_bytes(int32_t B, int32_t H, int32_t Tq);
int64_t ecamp_sr_bwd_workspace_bytes(void);
/* Process-wide switches with no reference counterpart (the "p8_" prefix is historical: the persistent one-workgroup-per-CU GEMM).
 * "p8_wgrad" (default 1): 0 keeps weight-gradient GEMMs off the persistent kernel.  "p8_wgrad_reserve_cus" (default 0): launch the
 * backward-pass forms of that kernel with this many fewer workgroups than CUs -- set by the data-parallel wrapper, whose all-reduce
 * kernels share the CUs during backward.  "q8_bwd_grid" (default 0 = one workgroup per CU; env ECAMP_Q8_BWD_GRID): n > 0 launches the
 * data-gradient form on min(output tiles, n) workgroups -- with n past the tile count every workgroup computes ONE tile and the
 * hardware dispatcher deals the tiles to whichever CUs the communication kernels leave free (+0.15 ms per step alone on a GPU);
 * set by the data-parallel wrapper as well. */
int ecamp_set_option(const char* name, int32_t value);
/* "q8_mode" (ecamp_set_option): -1 automatic (default), 0 never, 2 whenever its alignment / size conditions hold -- the
 * persistent 256x256x64 kernel (csrc/gemm_q8.h) that serves the forward, data-gradient and weight-gradient forms. */
/* "q16_mode" (ecamp_set_option; env ECAMP_Q16): the four-wave v_mfma_f32_16x16x32_bf16 kernel (csrc/gemm_q16.h; forward and data-gradient forms
 * with a plain / bias / residual epilogue, 256 x 256 or 256 x 192 tiles).  0 never; 1 (default) where the 192-column tile removes idle
 * last-round time -- the model's 768-wide outputs; 2 every eligible call; 3 as 2 whatever the size (tests); -1 back to the environment */
/* "attn_head" (ecamp_set_option): 1 (default; env ECAMP_ATTN_HEAD) one workgroup per (batch, head) with everything resident in LDS
 * for sequences that fit (<= 256 tokens here), 0 the 64-row streaming kernels for every length, -1 back to the environment's choice */

/* ---- fp8 forward (BASELINE.json configs[4]: "fp8 MFMA forward (bf16 grads) for QKV/MLP GEMMs"; no reference counterpart -- the
 * reference runs these nn.Linear layers under torch.cuda.amp, main_pretrain.py:138).  Per-tensor scaling, OCP e4m3:
 *   ecamp_amax      out[0] = max(out[0], max|x|)               (caller zeroes out[0]; n % 4 == 0)
 *   ecamp_quant_fp8 scale_out[0] = max(amax[0], tiny) / 448;  q[i] = e4m3(clamp(x[i] / scale, +-448))   (one byte per element)
 *   ecamp_gemm_fp8  C[M,N] (bf16) = act((A8[M,K] . B8[N,K]^T) * scale_a[0] * scale_b[0] + bias) (+ residual); act 0 none, 1 exact GELU, 2 exact GELU with gelu'(pre-activation) saved to pre_out (see ecamp_gemm),
 *                   with the bf16 pre-activation saved to pre_out -- the forward of timm Attention.qkv / proj and Mlp.fc1 / fc2
 *                   (call sites model_ecamp.py:233-234, 254-255).  K, lda, ldb multiples of 16 bytes.  f32 accumulation with unit block
 *                   scales: v_mfma_scale_f32_32x32x64_f8f6f4 in the persistent 256 x 256 x 128 kernel (csrc/gemm_q8.h, F8; every shape of
 *                   the model at B >= 64), v_mfma_scale_f32_16x16x128_f8f6f4 in the 128^2 kernel that serves small / unaligned shapes. */
int ecamp_amax(const void* x, float* out, int64_t n, int32_t dtype, ecampStream_t stream);
/* Delayed per-tensor scaling (round 4): a GEMM input site owns scale[0] (f32: what its producer quantises with and ecamp_gemm_fp8
 * dequantises with during this optimizer step) and 16 amax slots 32 floats apart (512 floats per site; what this step's producers saw).
 *   ecamp_quant_fp8_delayed  q[i] = e4m3(clamp(x[i] / scale[0], +-448)); amax_slots[(workgroup & 15) * 32] = max(., |x|)  -- one pass
 *   ecamp_fp8_roll           per site i < n: a = max over its slots; slots = 0; if a > 0: hist[i][hist_pos % hist_len] = a (hist nullable);
 *                            scale[i] = margin * max(a, hist[i][...]) / 448   (once per optimizer step).  hist = NULL, margin = 1: exact
 *                            current scaling (the weights).  Activation sites: the largest maximum of the last hist_len fed steps times a
 *                            margin, so that an activation that grows from one step to the next does not saturate at +-448
 *   ecamp_layernorm_fwd_q8   ecamp_layernorm_fwd that also writes the e4m3 copy of y (same conventions): the quantisation folded into
 *                            the producer of the GEMM input (nn.LayerNorm sites model_ecamp.py:69,84,235,256; BertSelfOutput / BertOutput) */
int ecamp_quant_fp8_delayed(const void* x, const float* scale, void* q, float* amax_slots, int64_t n, int32_t dtype, ecampStream_t stream);
int ecamp_fp8_roll(float* amax_slots, float* scale, int32_t n, float* hist, int32_t hist_len, int32_t hist_pos, float margin, ecampStream_t stream);
/* The e4m3 copies of all weights of a flat bf16 arena in two launches per optimizer step (exact per-matrix scaling): items = DEVICE table
 * of int32[4] {first element (multiple of 4), count (multiple of 4, <= 65536), scale id, 0}, one workgroup each.  pass 0: the item's
 * max|w| into the amax slots of its scale id; (ecamp_fp8_roll); pass 1: w8[i] = e4m3(clamp(w[i] / scales[id], +-448)). */
int ecamp_fp8_weights(const void* w_bf16, void* w8, const int32_t* items, int32_t nitems, float* amax_slots, const float* scales,
                      int32_t pass, ecampStream_t stream);
int ecamp_layernorm_fwd_q8(const void* x, const void* residual, void* z_out, const float* gamma, const float* beta, void* y, float* mean,
                           float* rstd, int64_t rows, int32_t cols, float eps, float drop_p, uint64_t seed, uint64_t offset, void* q8,
                           const float* q8_scale, float* q8_amax_slots, int32_t dtype, ecampStream_t stream);
int ecamp_quant_fp8(const void* x, const float* amax, void* q, float* scale_out, int64_t n, int32_t dtype, ecampStream_t stream);
int ecamp_gemm_fp8(const void* A8, const void* B8, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                   const float* scale_a, const float* scale_b, const float* bias, const void* residual, int64_t ldr, void* pre_out,
                   int64_t ldp, int act, void* q8_out, const float* q8_scale, float* q8_amax_slots, ecampStream_t stream);
/* q8_out (nullable; needs act = 1 or 2, pre_out, no residual, ldc = N): also leave the e4m3 copy of C for the NEXT dense layer (timm Mlp.fc2 /
 * BertOutput.dense behind the GELU), quantised with that layer's input scale q8_scale[0], and max|C| in its amax slots -- the fp8
 * forward's quantisation folded into the epilogue that produces the activation (ecamp_quant_fp8_delayed's conventions). */

/* ---- LayerNorm (nn.LayerNorm eps 1e-6: model_ecamp.py:69,84,235,256 + timm Block norms; HF LN eps 1e-12:
 * BertSelfOutput/BertOutput/BertEmbeddings/transform).  y = LN(z), z = dropout(x) + residual (both optional). */
int ecamp_layernorm_fwd(const void* x, const void* residual, void* z_out, const float* gamma, const float* beta, void* y,
                        float* mean, float* rstd, int64_t rows, int32_t cols, float eps, float drop_p, uint64_t seed,
                        uint64_t offset, int32_t dtype, ecampStream_t stream);
int ecamp_layernorm_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                        const void* dres_in, void* dz, void* dx_drop, float* dgamma, float* dbeta, int64_t rows, int32_t cols,
                        float drop_p, uint64_t seed, uint64_t offset, int32_t dtype, ecampStream_t stream);

/* ---- attention (timm Attention.forward: softmax(q k^T * hd^-1/2) v; HF BertSelfAttention 4.42.4 incl. the
 * cross-attention mode of context_fusion.py:45-53).  strides = {batch, token, head} in elements, head_dim contiguous.
 * key_mask: int32 [B,Tk], nonzero = attend (the additive finfo.min mask of bert_modeling.py:92), or NULL.
 * drop_mask (optional, ecamp_attn_mask_bytes(...) bytes): the dropout keep-mask of the probabilities as bits, written by the forward
 * call and read by the backward call that is given the same buffer -- the Philox stream is then evaluated once per score instead of
 * three times (forward, dQ pass, dK/dV pass).  NULL (or a size of 0: shapes the bit form does not serve): the backward pass
 * regenerates the mask from (seed, offset); both forms give bit-identical results. */
int64_t ecamp_attn_mask_bytes(int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t hd, int32_t dtype);
int ecamp_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* key_mask, int32_t B,
                   int32_t H, int32_t Tq, int32_t Tk, int32_t hd, const int64_t* q_strides, const int64_t* k_strides,
                   const int64_t* v_strides, const int64_t* o_strides, float scale, float drop_p, uint64_t seed, uint64_t offset,
                   int32_t dtype, void* drop_mask, ecampStream_t stream);
int ecamp_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                   float* delta_ws, void* dq, void* dk, void* dv, const int32_t* key_mask, int32_t B, int32_t H, int32_t Tq,
                   int32_t Tk, int32_t hd, const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides,
                   const int64_t* o_strides, const int64_t* do_strides, const int64_t* dq_strides, const int64_t* dk_strides,
                   const int64_t* dv_strides, float scale, float drop_p, uint64_t seed, uint64_t offset, int32_t dtype,
                   const void* drop_mask, ecampStream_t stream);
/* Attention probabilities softmax(scale * q k^T + mask), f32 [B,H,Tq,Tk] (Tk <= 1024): the tensor the reference's Visualization
 * model returns from the fusion layer's cross-attention (Visualization/module/context_fusion.py:45-57,
 * Visualization/module/model_ecamp.py:308-319).  Evaluation only. */
int ecamp_attn_probs(const void* q, const void* k, const int32_t* key_mask, float* probs, int32_t B, int32_t H, int32_t Tq, int32_t Tk,
                     int32_t hd, const int64_t* q_strides, const int64_t* k_strides, float scale, int32_t dtype, ecampStream_t stream);

/* ---- elementwise / reductions ---- */
int ecamp_add(const void* a, const void* b, void* y, int64_t n, int32_t dtype, ecampStream_t stream);
int ecamp_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, int32_t dtype, ecampStream_t stream);
int ecamp_cast(const void* src, void* dst, int64_t n, int32_t src_dtype, int32_t dst_dtype, ecampStream_t stream);
/* y = x * alpha * (alpha_dev ? *alpha_dev : 1) in f32 arithmetic; x == y allowed (autograd's scaling of a gradient tensor by an upstream
 * scalar, bert_modeling.py:213-217 -> loss.backward()) */
int ecamp_scale(const void* x, void* y, int64_t n, float alpha, const float* alpha_dev, int32_t dtype, ecampStream_t stream);
int ecamp_zero(void* p, int64_t bytes, ecampStream_t stream);
/* out[n] += alpha * sum over rows m (optionally only rows with lo <= m % period < hi) of X[m*ld+n]: bias grads,
 * cls-token grad (model_ecamp.py:228-230). */
int ecamp_colsum(const void* X, int64_t ld, int64_t M, int64_t N, float alpha, const float* alpha_dev, int32_t period, int32_t lo, int32_t hi,
                 float* out, int32_t dtype, ecampStream_t stream);
int ecamp_bcast_add(const void* x, const void* g, void* y, int64_t B, int32_t S, int32_t H, int32_t dtype,
                    ecampStream_t stream); /* context_fusion.py:55 */
int ecamp_seq_sum(const void* x, void* out, int64_t B, int32_t S, int32_t H, int32_t s0, int32_t s1, float scale, int32_t dtype,
                  ecampStream_t stream); /* model_ecamp.py:269 gap token; grad of the broadcast */
int ecamp_seq_bcast(const void* g, void* y, int64_t B, int32_t S, int32_t H, int32_t s0, int32_t s1, float scale, int32_t mode,
                    int32_t dtype, ecampStream_t stream);
int ecamp_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, ecampStream_t stream); /* model_ecamp.py:177 */

/* ---- image side ---- */
int ecamp_bicubic_resize(const float* src, float* dst, int64_t planes, int32_t Hs, int32_t Ws, int32_t Hd, int32_t Wd,
                         ecampStream_t stream); /* model_ecamp.py:318 */
/* The same resize from the dataset's grayscale crop as uint8 [B,Hs,Ws] (pretrain_datasets.py:47-52: Grayscale(3) + ToTensor +
 * Normalize with ONE mean / std, i.e. three identical channels): dst f32 [B,3,Hd,Wd] = resize(lut[src]), the three planes written
 * from one evaluation.  lut: 256 f32 on the device, lut[u] = ((float)u / 255 - mean) / std in f32 arithmetic (ToTensor + Normalize
 * per byte value), which makes the result bit for bit what ecamp_bicubic_resize gives on the f32 [B,3,Hs,Ws] image for the exact 2x
 * ratio of the hot path and for Hs == Hd (the normalised image itself).  One byte per pixel crosses PCIe and is read instead of twelve. */
int ecamp_bicubic_resize_u8(const uint8_t* src, float* dst, int64_t B, int32_t Hs, int32_t Ws, int32_t Hd, int32_t Wd, const float* lut,
                            ecampStream_t stream);
int ecamp_mask_indices(const float* noise, int64_t B, int32_t L, int32_t len_keep, int32_t* ids_restore, int32_t* ids_keep,
                       float* mask, ecampStream_t stream); /* model_ecamp.py:168-193 */
int ecamp_im2col_gather(const float* imgs, const int32_t* ids_keep, void* out, int64_t B, int32_t Lk, int32_t C, int32_t R,
                        int32_t p, int32_t dtype, ecampStream_t stream); /* model_ecamp.py:220 + :185 */
int ecamp_assemble_tokens(void* x, const float* cls, const float* pos, const int32_t* ids_keep, int64_t B, int32_t Lk, int32_t D,
                          int32_t dtype, ecampStream_t stream); /* model_ecamp.py:222,228-230 */
/* The dataset's image transform on the device (pretrain_datasets.py:47-52,113-115: RandomResizedCrop(448, bicubic) + RandomHorizontalFlip +
 * Grayscale): B crops of a pre-decoded uint8 grayscale radiograph -> dst uint8 [B, out, out], equal BYTE FOR BYTE to
 * PIL's img.crop(box).resize((out, out), BICUBIC) (+ FLIP_LEFT_RIGHT, convert('L')) on the same pixels -- Pillow's two-pass antialiased
 * resample (Resample.c: double-precision coefficients -> 22-bit fixed point, uint8 intermediate) restated in csrc/augment.hip.
 * src: the crops' bytes back to back (each h x w, contiguous); table int64 [B, 6] on the device = {byte offset in src, h, w, flip (0/1),
 * first row of the sample in the intermediate (prefix sum of h), 0}; kmax >= the tap count of the batch's largest scale factor
 * (2 * ceil(2 * max(1, size / out)) + 1); tmp_rows = sum of h; max_h = largest h; ws from ecamp_resample_crops_workspace_bytes;
 * err_flag: int32 on the device, set to 1 if a sample needs more than kmax taps (the result is then undefined). */
int64_t ecamp_resample_crops_workspace_bytes(int64_t B, int32_t out, int32_t kmax, int64_t tmp_rows);
int ecamp_resample_crops_u8(const uint8_t* src, const int64_t* table, uint8_t* dst, int64_t B, int32_t out, int32_t kmax, int64_t tmp_rows,
                            int32_t max_h, void* ws, int64_t ws_bytes, int32_t* err_flag, ecampStream_t stream);
int ecamp_unshuffle_fwd(const void* y, const int32_t* ids_restore, const float* mask_token, const float* dpos, void* xd, int64_t B,
                        int32_t L, int32_t Lk, int32_t D, int32_t dtype, ecampStream_t stream); /* model_ecamp.py:245-251 */
int ecamp_unshuffle_bwd(const void* dxd, const int32_t* ids_restore, const int32_t* ids_keep, void* dy, float* dmask_token,
                        int64_t B, int32_t L, int32_t Lk, int32_t D, int32_t dtype, ecampStream_t stream);
int ecamp_unpatchify_mim(const void* pred, const float* imgs, const float* mask, float* pred_img, float* loss_sum, int64_t B,
                         int32_t R, int32_t p, int32_t dtype, ecampStream_t stream); /* model_ecamp.py:153-165,288-298 */
int ecamp_img_loss_bwd(const float* pred_img, const float* imgs, const float* mask, const float* dsr, const float* gm_gs,
                       void* dpred, int64_t B, int32_t R, int32_t p, int32_t dtype, ecampStream_t stream);
int ecamp_sr_fwd(const float* pred_img, const void* big, const float* big_lut, const int64_t* column,
                 const int64_t* row, const float* w1, const float* b1, const float* w2, const float* b2, float* loss_sum, int64_t B,
                 int32_t R, int32_t super_patch, int32_t window, int32_t mode,
                 ecampStream_t stream); /* model_ecamp.py:28-46,196-215,291-299; fused, LDS-resident.
                 mode 0: f32 VALU stencils (parity); mode 1: bf16 matrix cores (4x4x4 MFMA per tap), f32 accumulate / skip / loss.
                 big: the loss target, f32 [B,3,2R,2R] (big_lut = NULL) or the uint8 crop [B,2R,2R] normalised on the fly through
                 big_lut[256] (see ecamp_bicubic_resize_u8; same bits as the f32 image) */
/* The SR head's output image itself, f32 [B,3,2R,2R] = super_res(pred_img) (model_ecamp.py:28-46,285): never materialised by the
 * training step (ecamp_sr_fwd folds it into the loss); for parity checks against the reference's activation and visualisation. */
int ecamp_sr_image(const float* pred_img, const float* w1, const float* b1, const float* w2, const float* b2, float* sr, int64_t B,
                   int32_t R, ecampStream_t stream);
int ecamp_sr_bwd(const float* pred_img, const void* big, const float* big_lut, const int64_t* column,
                 const int64_t* row, const float* w1, const float* b1, const float* w2, const float* b2, float* dsr, float* gw_ws,
                 int64_t B, int32_t R, int32_t super_patch, int32_t window, int32_t mode, ecampStream_t stream);
int ecamp_scaled_accum(const float* ws, float* grad, const float* scale_dev, int32_t idx, int32_t n, ecampStream_t stream);

/* ---- report side ---- */
int ecamp_bert_embed_fwd(const int64_t* ids, const int64_t* type_ids, const float* word, const float* pos, const float* type,
                         const float* gamma, const float* beta, void* z, void* e, float* mean, float* rstd, int64_t B, int32_t S,
                         int32_t cols, float eps, float drop_p, uint64_t seed, uint64_t offset, int32_t dtype,
                         ecampStream_t stream); /* HF BertEmbeddings, bert_modeling.py:113 */
int ecamp_bert_embed_bwd(const void* de, const void* z, const float* mean, const float* rstd, const float* gamma, const int64_t* ids,
                         const int64_t* type_ids, float* gword, float* gpos, float* gtype, float* dgamma, float* dbeta, int64_t B,
                         int32_t S, int32_t cols, int32_t pad_id, int32_t hot0, int32_t hot1, float drop_p, uint64_t seed,
                         uint64_t offset, int32_t dtype, ecampStream_t stream);
int ecamp_ce_fwd_bwd(void* logits, const int64_t* labels, const float* weights, float* loss_sum, int64_t M, int32_t V, int64_t ld,
                     float inv_count, int32_t dtype, ecampStream_t stream); /* bert_modeling.py:213-217 */

/* ---- optimizer side ---- */
/* optimizer.zero_grad() (main_pretrain.py:169) without touching the weight matrices: zero the 64-element blocks of the gradient arena
 * whose flag byte is non-zero (n = arena length, a multiple of 64).  The weight-gradient GEMMs of the next backward pass overwrite
 * their outputs (ecamp_gemm accumulate = 0) instead of adding to a zeroed buffer. */
int ecamp_zero_blocks(float* g, const uint8_t* block_flags, int64_t n, ecampStream_t stream);
int ecamp_sumsq(const float* x, int64_t n, float* out, ecampStream_t stream); /* util/misc.py:280-292 */
int ecamp_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1, float beta2,
                float eps, float weight_decay, int64_t step, float grad_scale,
                ecampStream_t stream); /* torch.optim.AdamW, main_pretrain.py:254 */

int ecamp_adamw_grouped(float* p, const float* g, float* m, float* v, void* p_bf16, const uint8_t* block_group, int64_t n,
                        int32_t ngroups, const float* lr_host, const float* wd_host, float beta1, float beta2, float eps,
                        int64_t step, float grad_scale, float* grad_sumsq, const float* ctl,
                        ecampStream_t stream); /* whole-arena AdamW with timm's decay/no-decay groups, main_pretrain.py:253-254;
                                                * grad_sumsq (nullable, caller-zeroed): += sum of (g * grad_scale)^2 over the updated
                                                * elements -- the global gradient norm of util/misc.py:280-292 without a second pass;
                                                * ctl (nullable, device f32[4] written by ecamp_loss_scale_update): {grad_scale, skip, bias
                                                * corrections} read on the DEVICE instead of `step` / `grad_scale` -- a skipped step writes nothing */
/* GradScaler's unscale_ + step + update (util/misc.py:262-269: `self._scaler.unscale_`, `.step(optimizer)`, `.update()`) decided on the
 * device, so the training loop never waits for the overflow flag: sumsq = sum(g^2) over the scaled gradients (ecamp_sumsq);
 * state f32[4] = {scale, growth tracker, skipped steps, unused}; opt_step f32[1] = optimizer steps actually taken (bias corrections);
 * ctl f32[4] = output for ecamp_adamw_grouped; norm_out (nullable) = sqrt(sumsq) / scale. */
int ecamp_loss_scale_update(const float* sumsq, float* state, float* opt_step, float* ctl, float* norm_out, float growth_factor,
                            float backoff_factor, int32_t growth_interval, float beta1, float beta2, ecampStream_t stream);

/* ---- optional in-process timing (bench.py roofline): HIP-event pairs around every GEMM / attention launch ---- */
int ecamp_prof_enable(int on);
int ecamp_prof_collect(int category, double* total_ms, double* total_work, int64_t* count); /* 0 gemm bf16, 1 gemm f32, 2 attention, 3 gemm fp8; <0 clears */
/* event pairs the timing facility currently holds: bounded (a pool of 4096 pairs whose finished records are folded into running
 * totals), however many steps run under `main_pretrain.py --profile` */
int64_t ecamp_prof_live_events(void);

/* ---- development ABI: NOT part of the drop-in boundary.  A C consumer sees these only with -DECAMP_DEV_ABI; the library always
 * exports them and the Python binding (which parses this header) binds them for tests/ and tools/ only.  No reference counterpart.
 *   ecamp_dev_spin              `blocks` workgroups spinning for `cycles` shader clocks on `stream` (tools/hog_probe.py: a stand-in
 *                               for a communication kernel sharing the GPU with the training step)
 *   ecamp_gemm_q8_launches      GEMM calls routed to the persistent 256x256x64 kernel so far (tests assert that it really ran)
 *   ecamp_wgrad_group_launches  grouped weight-gradient launches issued so far (tests assert that the grouped path really ran)
 *   ecamp_gemm_f8_q8_launches   ecamp_gemm_fp8 calls routed to the persistent 256 x 256 x 128 e4m3 kernel so far
 *   ecamp_attn_head_launches    launches of the head-resident bf16 attention kernels (one workgroup per (batch, head)) so far
 *   ecamp_dropout_mask          keep[e] = 1 iff element e of a tensor survives dropout(p) under (seed, offset): the Philox mask every kernel
 *                               of this library regenerates (attention probabilities: e = ((b*H + h)*Tq + i)*Tk + j; LayerNorm / embedding
 *                               dropout: e = row*cols + col), as bytes -- lets the tests and the oracle replay the reference's dropout
 *                               sites (context_fusion.py:28-57, bert_modeling.py:113,131) under the SAME masks in plain PyTorch */
#ifdef ECAMP_DEV_ABI
int ecamp_dropout_mask(uint8_t* keep, int64_t n, float p, uint64_t seed, uint64_t offset, ecampStream_t stream);
int ecamp_dev_spin(int32_t blocks, int32_t threads, int64_t cycles, ecampStream_t stream);
int64_t ecamp_gemm_q8_launches(void);
int64_t ecamp_wgrad_group_launches(void);
int64_t ecamp_gemm_f8_q8_launches(void);
int64_t ecamp_gemm_q16_launches(void);   /* GEMM calls routed to the four-wave 16x16x32 kernel (csrc/gemm_q16.h) so far */
int64_t ecamp_attn_head_launches(void);
int64_t ecamp_prof_dump(char* buf, int64_t cap);   /* per (form, epilogue, shape) totals of the profiled GEMM launches: "<tag> <n> <ms> <flop>" lines */
#endif

#ifdef __cplusplus
}
#endif
#endif
