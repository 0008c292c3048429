/*
 * pram_hip.h — C ABI of libpram_hip.so, the MI355X (gfx950) kernels underneath the PRAM
 * per-query hot path (SFD2 extract -> SegNetViT recognise -> GML/AdaGML match + Sinkhorn).
 *
 * The reference (feixue94/pram) is pure Python on stock torch ops and has no FFI; the boundary
 * it exposes is dict-in/dict-out nn.Modules.  This header is the native
 * boundary a maintainer binds underneath those modules (ctypes stub in INTEGRATION.md); each
 * entry names the reference torch-op site (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host"; fp32 unless typed otherwise.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing syncs.
 *   - no allocation inside; workspaces are passed in and sized by the *_workspace_bytes query.
 *   - return 0 on success, negative on error (PRAM_E_*); pram_last_error() gives the text.
 *   - ragged batches: `*_lens` are per-batch-element int32 device arrays (NULL = every
 *     element has the padded max length).  The reference has no masks; a padded row/key is
 *     never read into a softmax, so each element computes exactly its B=1 result.
 *   - token matrices are row-major [rows][ld]; heads are 64-wide column blocks.
 *
 * Limits (each is checked and comes back as PRAM_E_ARG with a message, never as a wrong result):
 *   - attention heads are exactly 64 wide (the shipped models: hidden 256 = 4 heads; pram_amd's Python layer raises on any
 *     other hidden_dim before it gets here);
 *   - keypoint selection sorts a top-k of up to 8192 keypoints per frame in LDS; larger bounds sort in the caller's workspace
 *     (pram_select_keypoints_workspace_bytes accounts for it) and max_keypoints >= h * w keeps everything unsorted;
 *   - AdaGML pruning handles token sets of at most 8192 tokens (pram_adagml_prune_f32);
 *   - LayerNorm rows are at most 1024 wide; the split-fp16 GEMMs need K % 32 == 0 (other shapes: the exact-fp32 entry);
 *   - the split-fp16 operands carry value * s in fp16 (s = 16 by default, pram_x3_set_act_scale): a finite |x| >= 65520 / s
 *     (4094.97) does not fit.  This one is NOT a silent limit
 *     either: see "range guard" below (pram_set_status_word) — the kernels report it, and pram_amd's Python layer re-runs the
 *     call on the exact-fp32 entries (or raises).
 */
#ifndef PRAM_HIP_H
#define PRAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRAM_OK 0
#define PRAM_E_ARG (-1)      /* bad shape / alignment / null pointer */
#define PRAM_E_LAUNCH (-2)   /* hipLaunchKernel failed */
#define PRAM_E_UNSUPPORTED (-3)

int pram_hip_version(void);
const char* pram_last_error(void);

/* ---------------------------------------------------------------- range guard of the split-fp16 ("x3") entries
 * The x3 entries carry every fp32 activation as two fp16 parts of value * s (s = the activation scale below, 16 by default).  A finite
 * value with |s x| >= 65520 rounds to
 * +-inf in its hi part and the result of the launch is garbage (usually NaN — but NaN scores turn into ordinary-looking
 * indices further down).  Kernel launches are asynchronous, so this cannot come back as a return code of the launch: every x3
 * kernel that splits fp32 values (GEMM / convolution operand staging, the plane-writing epilogues) ORs PRAM_STATUS_X3_RANGE
 * into a caller-owned DEVICE status word when it meets such a value.  The word is sticky until reset.  NaN inputs are not
 * flagged: they propagate as NaN, exactly as in the reference's fp32 arithmetic; +-inf inputs are flagged.
 *   pram_set_status_word : registers the word for the CURRENT device (NULL detaches; without a word nothing is reported).
 *   pram_read_status_word: enqueues a copy of the word to *host_out on `stream`, waits for the stream, and (reset != 0) clears
 *                          the word if it was set.  A caller that sees PRAM_STATUS_X3_RANGE re-runs on the *_f32 entries. */
#define PRAM_STATUS_X3_RANGE 1u
int pram_set_status_word(unsigned int* device_word);
/* Activation scale.  The planes carry value * s with s = 16 by default: the range is |x| < 65520 / s = 4094.97 and values down
 * to 2^-25 / s keep their last bit (the lo part is an fp16 subnormal below 2^-14).  A model whose activations are larger (trained
 * checkpoints; the range guard trips) lowers s instead of leaving the split path: s = 1 carries |x| < 65520 with an absolute floor
 * of 3e-8, s = 2^-4 a million with 4.8e-7 — the pair (hi, lo) is a 22-bit significand whatever the scale, only the floor moves.
 *   pram_x3_set_act_scale(s): s = a power of two in [2^-12, 2^4] -> sets the scale for the CALLING HOST THREAD and returns the
 *   previous one; s <= 0 only queries; anything else returns -1 and changes nothing.  Every x3 entry reads it at launch and
 *   hands it to its kernel by value: planes written under one setting must be consumed under the same setting (pram_amd's models
 *   set it around their forward, nets/_blocks.py); a captured hipGraph replays with the value it was captured under. */
float pram_x3_set_act_scale(float scale);
int pram_read_status_word(unsigned int* host_out, int reset, void* stream);

/* ---------------------------------------------------------------- token linear algebra */

/* flags for pram_linear_f32 */
#define PRAM_LIN_ROTARY 1  /* rotate column pairs (c, c+32) of each 64-wide head for c < rot_cols */

/* out[m][n] = alpha * ( [A0 | A1] · Wᵀ + bias ) + residual          (nn.Linear sites:
 * nets/segnetvit.py:87-95,157-164; nets/gml.py:118-126,151-159,220,235; K6 in SURVEY.md §2.1)
 *   A0 [m][lda0] supplies k in [0,k0), A1 [m][lda1] supplies k in [k0,k0+k1) (the torch.cat of
 *   segnetvit.py:106); a1 may be NULL with k1 = 0.  W is [n][k0+k1] (torch layout).
 *   PRAM_LIN_ROTARY: W rows were pre-permuted so that within each 64-wide head the even
 *   rotary dims are columns 0..31 and the odd dims 32..63; cos/sin are [m][32]
 *   (apply_cached_rotary_emb, segnetvit.py:21-23).
 * Constraints: k0 % 32 == 0 when k1 > 0; (k0+k1) % 4 == 0; lda % 4 == 0. */
int pram_linear_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1,
                    const float* w, const float* bias, const float* residual, int ldr,
                    float* out, int ldo, int m, int n, float alpha, int flags,
                    const float* rot_cos, const float* rot_sin, int rot_cols, void* stream);

/* fp16-operand variant for BASELINE config C5 ("fp16 MFMA path"): identical contract, but `w16` is the weight
 * matrix pre-converted to IEEE fp16 ([n][k0+k1]) and the activations are rounded to fp16 on their way into LDS;
 * products on v_mfma_f32_32x32x16_f16, fp32 accumulate and epilogue.  Needs (k0+k1) % 8 == 0 and, with a
 * second segment, k0 % 64 == 0.  Not for the fp32 parity configurations. */
int pram_linear_f16_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1,
                        const void* w16, const float* bias, const float* residual, int ldr,
                        float* out, int ldo, int m, int n, float alpha, int flags,
                        const float* rot_cos, const float* rot_sin, int rot_cols, void* stream);

/* pram_linear_f16_f32 on a ragged token matrix, as pram_linear_ragged_f32 / pram_linear_x3_ragged_f32: rows are sequences of
 * t_pad rows, sequence s has lens[s] (device int32) valid ones; output tiles without a valid row are skipped and only valid rows
 * are stored (`out` may be a persistent buffer whose other rows belong to someone else: AdaGML commits the matching descriptors
 * of the pairs that stop at a layer, nets/adagml.py:386-396).  lens == NULL: pram_linear_f16_f32. */
int pram_linear_f16_ragged_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1,
                               const void* w16, const float* bias, const float* residual, int ldr,
                               float* out, int ldo, int m, int n, float alpha, int flags,
                               const float* rot_cos, const float* rot_sin, int rot_cols,
                               const int* lens, int t_pad, void* stream);

/* pram_linear_f16_f32 that also (or only: out may be NULL) writes the result rounded to fp16 — the q / k / v operand
 * of pram_attention_h16_f32, which would otherwise round the fp32 result itself while staging it (same values). */
int pram_linear_f16_h16(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1,
                        const void* w16, const float* bias, const float* residual, int ldr,
                        float* out, int ldo, void* out16, int ldo16, int m, int n, float alpha, int flags,
                        const float* rot_cos, const float* rot_sin, int rot_cols, void* stream);

/* Split-fp16 ("x3") variant: fp32-class results on the fp16 matrix pipe (same nn.Linear sites; default path).
 * Every fp32 operand x is carried as two fp16 numbers, x * s = hi + lo (s a power of two), and every product as
 * three v_mfma_f32_32x32x16_f16 accumulated in fp32: a.b ~= (a_hi.b_hi + a_hi.b_lo + a_lo.b_hi) / (s_a s_b) — the
 * dropped lo.lo term and the truncation of hi + lo are 2^-22 relative, the class of fp32 rounding itself.
 *   w_hi / w_lo : the weight matrix [n][k0+k1] * w_scale split on the host into two fp16 planes
 *                 (pram_amd/ops.py::split_weight; w_scale = the power of two that puts max|w| in [2^13, 2^14));
 *   activations : fp32 in HBM, split with s = 16 while they are staged (|x| must stay below 4094);
 *   out_hi / out_lo (optional, both or neither; then `out` may be NULL): the result * 16 as split planes
 *                 [m][ldo16] fp16 — the q / k / v operand format of pram_attention_x3_f32.
 * Needs (k0+k1) % 8 == 0 and, with a second segment, k0 % 32 == 0. */
int pram_linear_x3_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1,
                       const void* w_hi, const void* w_lo, float w_scale, const float* bias,
                       const float* residual, int ldr, float* out, int ldo, void* out_hi, void* out_lo,
                       int ldo16, int m, int n, float alpha, int flags, const float* rot_cos,
                       const float* rot_sin, int rot_cols, void* stream);

/* Profiling aid (no product caller): with PRAM_GEMM_ABLATE=4 in the environment the wide split-fp16 GEMM adds, per workgroup,
 * the shader-clock cycles of each main-loop phase to eight device counters: [0] issue + MFMA k-steps, [1] wait for the
 * chunk's loads, [2] commit (split + LDS writes), [3] barrier, [4] whole main loop, [5] workgroups, [6] epilogue;
 * [8 + 8 w + i] = eight time stamps of wave w of workgroup 0 in its fourth chunk (gemm_core_x3w.h).
 * out72 = HOST array of 72 counters; reset != 0 clears them after the read. */
int pram_debug_gemm_phases(unsigned long long* out72, int reset);

/* The q | k | v projection of an attention block in one call (nets/segnetvit.py:87-95, nets/gml.py:151-159), split-fp16 path:
 * columns [0, vt_col0) leave as row-major split planes [m][ldo16] (out_hi / out_lo: the q / k operands of pram_attention_x3_f32,
 * rotary applied to the first rot_cols of them when PRAM_LIN_ROTARY is set); the last heads * 64 columns — the values — leave
 * as the transposed, key-permuted planes [m / t_seq][heads][64][t_seq] that pram_attention_x3_vt would build from them (zeros
 * for tokens >= lens[s]; lens may be NULL).  Rows are sequences of t_seq tokens, t_seq % 64 == 0; n == vt_col0 + heads * 64. */
int pram_linear_x3_qkv_f32(const float* a0, int lda0, int k0, const void* w_hi, const void* w_lo, float w_scale,
                           const float* bias, void* out_hi, void* out_lo, int ldo16, void* vt_hi, void* vt_lo, int vt_col0,
                           int heads, int t_seq, int m, int n, int flags, const float* rot_cos, const float* rot_sin,
                           int rot_cols, const int* lens, void* stream);

/* pram_linear_x3_f32 with the activations already split: [A0 | A1] given as fp16 planes (value * s = hi + lo, s = pram_x3_set_act_scale's value in force at the PRODUCER's launch and here, 16 by default; [m][lda]
 * halves) written by the epilogues of pram_linear_x3[p]_f32 / pram_attention_x3_f32 / pram_layernorm_gelu_x3.  Both operands are
 * then staged with plain 16-byte copies (the fp32-input form splits A again in every column tile and is instruction-issue
 * bound).  k0, k1 multiples of 32; lda multiples of 8. */
int pram_linear_x3p_f32(const void* a0_hi, const void* a0_lo, int lda0, int k0, const void* a1_hi, const void* a1_lo,
                        int lda1, int k1, const void* w_hi, const void* w_lo, float w_scale, const float* bias,
                        const float* residual, int ldr, float* out, int ldo, void* out_hi, void* out_lo, int ldo16,
                        int m, int n, float alpha, int flags, const float* rot_cos, const float* rot_sin, int rot_cols,
                        void* stream);

/* Ragged token matrices (AdaGML's pruned / stopped pairs): rows are sequences of t_pad rows, sequence s has lens[s] (device
 * int32) valid ones.  Output tiles that contain no valid row are skipped and their outputs left untouched; everything else is
 * the un-ragged call.  (The reference shrinks its tensors with boolean indexing instead, nets/adagml.py:354-372.) */
int pram_linear_ragged_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1,
                           const float* w, const float* bias, const float* residual, int ldr,
                           float* out, int ldo, int m, int n, float alpha, int flags,
                           const float* rot_cos, const float* rot_sin, int rot_cols,
                           const int* lens, int t_pad, void* stream);
int pram_linear_x3_ragged_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1,
                              const void* w_hi, const void* w_lo, float w_scale, const float* bias,
                              const float* residual, int ldr, float* out, int ldo, void* out_hi, void* out_lo,
                              int ldo16, int m, int n, float alpha, int flags, const float* rot_cos,
                              const float* rot_sin, int rot_cols, const int* lens, int t_pad, void* stream);

/* Batched C_b = alpha * A_b · B_bᵀ (einsum 'bmd,bnd->bmn', nets/gml.py:282; K12).
 * A_b = a + b*stride_a, [m_max][lda]; B_b [n_max][ldb]; C_b [m_max][ldc]. */
int pram_bgemm_nt_f32(const float* a, int lda, long long stride_a, const float* b, int ldb,
                      long long stride_b, float* c, int ldc, long long stride_c, int batch,
                      int m_max, int n_max, int k, float alpha, void* stream);

/* pram_bgemm_nt_f32 on the split-fp16 path: both operands as split planes (value * s = hi + lo at the activation scale in force, what the projection epilogues
 * write: pram_linear_x3_f32 out_hi / out_lo): a [m_max][lda], b [n_max][ldb == k] per batch element; strides in halves for the
 * planes, in floats for c.  k % 32 == 0, lda and the plane strides % 8 == 0.  Three fp16 MFMAs per product, fp32-class result. */
int pram_bgemm_nt_x3p_f32(const void* a_hi, const void* a_lo, int lda, long long stride_a, const void* b_hi, const void* b_lo,
                          int ldb, long long stride_b, float* c, int ldc, long long stride_c, int batch, int m_max, int n_max,
                          int k, float alpha, void* stream);

/* y = GELU(LayerNorm(x)) rowwise, eps 1e-5, exact erf GELU (nn.LayerNorm + nn.GELU,
 * nets/segnetvit.py:92-93,161-162; K11).  In place when y == x.  cols <= 1024, cols % 4 == 0 */
int pram_layernorm_gelu_f32(const float* x, int ldx, float* y, int ldy, const float* gamma,
                            const float* beta, int rows, int cols, float eps, void* stream);

/* the same, skipping the rows beyond their sequence's length (see pram_linear_ragged_f32) */
int pram_layernorm_gelu_ragged_f32(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta,
                                   int rows, int cols, float eps, const int* lens, int t_pad, void* stream);

/* Fourier positional encoding (normalize_keypoints nets/utils.py:17-24 +
 * LearnableFourierPositionalEncoding nets/segnetvit.py:35-40; K7):
 *   nk = (kpts - (cx,cy)) / scale ; proj = Wr·nk ; cos_out/sin_out [rows][32].
 * Pass cx = cy = 0, scale = 1 for pre-normalised keypoints. */
int pram_fourier_encoding_f32(const float* kpts, const float* wr, float cx, float cy, float scale,
                              float* cos_out, float* sin_out, int rows, void* stream);

/* ---------------------------------------------------------------- attention (graded kernel) */

/* Flash-style multi-head attention, head_dim 64, exact-fp32 MFMA (Attention.forward
 * nets/segnetvit.py:73-76; for cross attention (nets/gml.py:175-179) see pram_attention_cross_f32 below; K8/K9):
 *   out[b, i, h*64:(h+1)*64] = softmax_j( scale * q[b,i,h]·k[b,j,h] ) · v[b,j,h]
 * q rows of batch b start at row b*m_max (k/v: b*n_max).  lse2 (optional) receives
 * log2-domain log-sum-exp [batch][heads][m_max] for pram_attention_colmean_f32. */
int pram_attention_f32(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                       float* out, int ldo, float* lse2, const int* q_lens, const int* k_lens,
                       int batch, int heads, int m_max, int n_max, float scale, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Keys are reduced in chunks of 512 folded in order.  A launch with fewer than 512 (batch, head, 128-row) units
 * — one to four query frames — cannot fill the chip, so when the caller passes a workspace of at least this many
 * bytes the chunks run as separate workgroups and a second kernel applies the same fold; the result is bit-identical
 * with or without the workspace.  Returns 0 when the launch would not be split (pass NULL / 0 then). */
size_t pram_attention_workspace_bytes(int batch, int heads, int m_max, int n_max);

/* The "fp16 MFMA path" of BASELINE config C5: same contract as pram_attention_f32 (fp32 tensors in HBM),
 * but Q/K/V and the probabilities are rounded to fp16 and multiplied on v_mfma_f32_32x32x16_f16 (fp32
 * accumulate, fp32 softmax).  ~1e-3 relative error: NOT for the fp32 parity configs. */
int pram_attention_f16_f32(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                           float* out, int ldo, float* lse2, const int* q_lens, const int* k_lens,
                           int batch, int heads, int m_max, int n_max, float scale, void* stream);

/* pram_attention_f16_f32 on q / k / v that are already fp16 in HBM (ld* in halves, multiples of 8): bit-identical
 * output, half the operand traffic, two K/V tiles in flight.  kv_shift = 0 for self attention; = pairs with
 * batch = 2*pairs for the single-launch cross attention (see pram_attention_cross_f32). */
int pram_attention_h16_f32(const void* q16, int ldq, const void* k16, int ldk, const void* v16, int ldv, float* out,
                           int ldo, float* lse2, const int* q_lens, const int* k_lens, int batch, int heads,
                           int m_max, int n_max, float scale, int kv_shift, void* stream);

/* Split-fp16 ("x3") flash attention — the default attention of the fp32 parity path (same einsum -> softmax -> einsum
 * sites as pram_attention_f32).  q / k are row-major split planes written by pram_linear_x3_f32 (value * s = hi + lo at the activation scale in force,
 * ld* in halves, multiples of 8; heads are 64-wide column blocks); vt_hi / vt_lo are the TRANSPOSED value planes of the
 * key side built by pram_attention_x3_vt: [batch][heads][64][tv], tv = n_max rounded up to 64.  S = K Q^T and O = P V are
 * fp16 MFMAs with fp32 accumulation: three per product for the scores and, by default, three per product for P V (the
 * probabilities enter as two fp16 parts like every other operand: 48 MFMAs per 64-key tile and 32-query wave);
 * pram_attention_x3_set_p_split(0) carries the probabilities as ONE fp16 from 1024 keys on (two MFMAs per P V product, 40 per
 * tile — not the default, see below); pram_attention_x3_mfma_per_tile reports what a launch issues.  Output fp32.
 * kv_shift as in pram_attention_h16_f32 (cross attention: batch = 2 * pairs, kv_shift = pairs).
 * Key chunks: from 1024 keys on the keys are reduced in chunks (pram_attention_x3_set_chunk_keys: default 4096 keys, i.e. one chunk for every shipped configuration), each chunk
 * normalised and all chunks folded in chunk order.  One workgroup normally walks all chunks of its 128 query rows and folds them
 * in registers; an under-filled launch (pram_attention_x3_is_split: one or two query frames, the reference's online loop
 * localization/loc_by_rec_online.py:109-133) runs groups of chunks as a second grid dimension, parks them in `workspace`
 * (pram_attention_x3_workspace_bytes; 0 when a sequence is a single chunk) and folds them in a second kernel: the SAME fold in
 * the SAME order, so the output is bit-identical whichever mode ran.  workspace == NULL (or too small): always fused. */
size_t pram_attention_x3_workspace_bytes(int batch, int heads, int m_max, int n_max);
int pram_attention_x3_is_split(int batch, int heads, int m_max, int n_max);
/* tuning / test knob: under-filled launches are split along the keys into as many groups of 512-key chunks as bring the grid to
 * `workgroups` (default 256 = one per CU; 0 = never; negative restores the default); returns the previous value.  The mode never
 * changes a result bit.  pram_attention_x3_is_split returns the number of groups (1 = fused). */
int pram_attention_x3_set_split_target(int workgroups);
/* keys per chunk (a multiple of 128; default 4096 or the environment's PRAM_ATTN_CHUNK_KEYS), process-wide: set it before the first
 * launch — it fixes where EVERY launch folds its partial soft-maxes (results move in their last bits, consistently for all batch
 * sizes).  Smaller chunks let shorter key sets use the split mode; a fused walk pays a spill round trip at every chunk end
 * (+17 % kernel time at 512 keys per chunk for 2048-key sets, +6 % at 2048 for 4096-key sets).  0 = query; returns the value. */
int pram_attention_x3_set_chunk_keys(int keys);
int pram_attention_x3_mfma_per_tile(int n_max);
/* probabilities in P V from 1024 keys on: 1 = two fp16 parts (three MFMAs per product; default), 0 = one fp16 (two MFMAs, ~15 % less
 * attention time, 2^-12 relative rounding per probability: logits 7e-4..1.3e-3 instead of 4e-5 from the fp32 oracle on flat attention: outside the 1e-3 parity bar on some shapes, an opt-in that is NOT parity-gated);
 * also PRAM_ATTN_P=split|fp16 in the environment.  Negative = query; returns the value in force. */
int pram_attention_x3_set_p_split(int split);
int pram_attention_x3_f32(const void* q_hi, const void* q_lo, int ldq, const void* k_hi, const void* k_lo, int ldk,
                          const void* vt_hi, const void* vt_lo, float* out, int ldo, float* lse2,
                          const int* q_lens, const int* k_lens, int batch, int heads, int m_max, int n_max,
                          float scale, int kv_shift, void* workspace, size_t workspace_bytes, void* stream);

/* Column means of the soft-max matrix of the pram_attention_x3_f32 call that produced lse2 (AdaGML's token scores,
 * nets/adagml.py:92-104): colmean[kb][j] = mean over heads and over the q_lens[b] query rows of softmax_row(scale q k^T)[i][j],
 * kb = (b + kv_shift) % batch, [batch][n_max] fp32; entries j >= k_lens[kb] are left untouched.  q / k: the same split planes,
 * lse2: [batch][heads][m_max] (log2 domain).  Second Q K^T pass on the fp16 matrix pipe, three MFMAs per product; no atomics. */
int pram_attention_x3_colmean_f32(const void* q_hi, const void* q_lo, int ldq, const void* k_hi, const void* k_lo, int ldk,
                                  const float* lse2, float* colmean, const int* q_lens, const int* k_lens, int batch,
                                  int heads, int m_max, int n_max, float scale, int kv_shift, void* stream);

/* Single-product ("fp16 MFMA path", BASELINE C5) flash attention on the software-pipelined kernel of the split path: q / k = fp16
 * row-major [rows][ld] (pram_linear_f16_h16), vt16 = the transposed, key-permuted fp16 values [batch][heads][64][tv] written by
 * pram_attention_x3_vt with NULL lo planes.  One fp16 MFMA per product; the probabilities are rounded to fp16 like the inputs
 * (this path's own, looser tolerance).  Output fp32; lse2 / lens / kv_shift as pram_attention_x3_f32. */
int pram_attention_h16t_f32(const void* q16, int ldq, const void* k16, int ldk, const void* vt16, float* out, int ldo,
                            float* lse2, const int* q_lens, const int* k_lens, int batch, int heads, int m_max,
                            int n_max, float scale, int kv_shift, void* stream);

/* Row-major value planes [seqs * t_max][ldv] (head h at columns 64 h ..) -> the transposed, key-permuted planes
 * pram_attention_x3_f32 stages with plain 16-byte copies: [seqs][heads][64][tv], tv = t_max rounded up to 64, position of
 * token t = 64 (t / 64) + perm(t % 64) (the order in which the S^T accumulator registers hold the keys); tokens
 * t >= lens[seq] (NULL: t_max) are written as zeros.  v_lo == vt_lo == NULL: one plane only (the fp16 values of
 * pram_attention_h16t_f32). */
int pram_attention_x3_vt(const void* v_hi, const void* v_lo, int ldv, void* vt_hi, void* vt_lo, const int* lens,
                         int seqs, int heads, int t_max, void* stream);

/* Both directions of CrossMultiHeadAttention.forward (nets/gml.py:175-179; adagml.py:222-231) in ONE launch:
 * 2*pairs sequences of t_max rows each, sequences 0..pairs-1 = set 0, pairs..2*pairs-1 = set 1; sequence s takes
 * its queries from rows s*t_max.. of qk (the shared to_qk projection) and its keys / values from sequence
 * (s + pairs) mod 2*pairs of qk / v.  lens [2*pairs] (optional).  Per sequence the arithmetic is exactly
 * pram_attention_f32's (same kernel, same tiling), so the result equals two separate calls bit for bit. */
int pram_attention_cross_f32(const float* qk, int ldqk, const float* v, int ldv, float* out, int ldo, float* lse2,
                             const int* lens, int pairs, int heads, int t_max, float scale, void* workspace,
                             size_t workspace_bytes /* pram_attention_workspace_bytes(2*pairs, heads, t_max, t_max) */, void* stream);
int pram_attention_cross_f16_f32(const float* qk, int ldqk, const float* v, int ldv, float* out, int ldo, float* lse2,
                                 const int* lens, int pairs, int heads, int t_max, float scale, void* stream);
/* Column means for the same pairing: colmean [2*pairs][t_max]; row kb holds, per token of sequence kb, the mean
 * over heads and over the queries of sequence (kb + pairs) mod 2*pairs (adagml.py:229). */
int pram_attention_cross_colmean_f32(const float* qk, int ldqk, const float* lse2, float* colmean, const int* lens,
                                     int pairs, int heads, int t_max, float scale, void* stream);

/* Column means of the attention matrix (AdaGML Attention.forward nets/adagml.py:148,229; K10):
 *   colmean[b][j] = 1/(heads*m_b) * sum_h sum_i softmax(scale q k^T)[b,h,i,j] */
int pram_attention_colmean_f32(const float* q, int ldq, const float* k, int ldk, const float* lse2,
                               float* colmean, const int* q_lens, const int* k_lens, int batch,
                               int heads, int m_max, int n_max, float scale, void* stream);

/* ---------------------------------------------------------------- optimal transport + matches */

size_t pram_sinkhorn_workspace_bytes(int batch, int m_max, int n_max);

/* sink_algorithm (nets/gml.py:27-46; K13): plain-domain Sinkhorn on the dustbin-augmented
 * (m+1)x(n+1) matrix, `iters` iterations, eps 1e-8.  dist [batch][m_max][ldd].
 * bin_score: device pointer to the scalar parameter.  Outputs:
 *   p_out (optional) [batch][m_max+1][ldp] the final P*u*v (for parity with sink_algorithm)
 *   compute_matches (nets/gml.py:304-319; K14) fused into the last pass:
 *   matches0 [batch][m_max] int64 (-1 none), matches1 [batch][n_max], mscores0/1 fp32.
 * Rows >= m_b / cols >= n_b of the outputs are filled with -1 / 0. */
int pram_sinkhorn_match_f32(const float* dist, int ldd, const int* m_lens, const int* n_lens,
                            const float* bin_score, int iters, float match_threshold,
                            float* p_out, int ldp, long long* matches0, long long* matches1,
                            float* mscores0, float* mscores1, int batch, int m_max, int n_max,
                            void* workspace, void* stream);

/* dual_softmax (nets/gml.py:20-24; K15) + compute_matches; same outputs as above. */
int pram_dual_softmax_match_f32(const float* dist, int ldd, const int* m_lens, const int* n_lens,
                                const float* bin_score, float match_threshold, float* p_out, int ldp,
                                long long* matches0, long long* matches1, float* mscores0,
                                float* mscores1, int batch, int m_max, int n_max, void* workspace,
                                void* stream);

/* ---------------------------------------------------------------- AdaGML token pruning (K10) */

/* PoolingLayer tail + pruning bookkeeping (nets/adagml.py:137,354-372,516-531), one token set per
 * workgroup: conf = sigmoid(logit); n_below[s] = #(conf < thr) over the set's current tokens (the
 * check_if_stop numerator).  If lens_in[s] >= n_min_tokens the tokens with conf > thr are compacted,
 * order preserved, into x_out / cos_out / sin_out / ind_out (else all are copied); lens_out[s] = kept.
 * x [sets][t_max][ldx], cos/sin [sets][t_max][32], ind [sets][t_max] (original token ids).
 * Out-of-place only; rows at and beyond lens_out[s] of the outputs are not written.  conf_out (optional) [sets][t_max]
 * receives the confidences.  row_map: scratch, int32 [sets][t_max] (source row of every destination row). */
int pram_adagml_prune_f32(const float* conf_logit, float thr, int n_min_tokens, const int* lens_in,
                          const float* x_in, const float* cos_in, const float* sin_in, const int* ind_in,
                          float* x_out, float* cos_out, float* sin_out, int* ind_out, int* lens_out,
                          int* n_below, float* conf_out, int* row_map, int sets, int t_max, int ldx, void* stream);

/* Scatter the matches of the pruned sets back to full size (nets/adagml.py:382-396):
 * out_matches[b][ind0[i]] = ind1[matches0[i]] where matches0[i] >= 0; out_scores[b][ind0[i]] = mscores0[i].
 * out_* [batch][m_full] must be pre-filled with -1 / 0. */
int pram_adagml_scatter_f32(const long long* matches0, const float* mscores0, const int* ind0, const int* ind1,
                            const int* lens0, int batch, int t_max, int m_full, long long* out_matches,
                            float* out_scores, void* stream);

/* ---------------------------------------------------------------- SFD2 (NHWC feature maps) */

/* Implicit-GEMM convolution, NHWC, exact-fp32 MFMA (ResNet4x conv stack nets/sfd2.py:281-293,
 * 331-333; K1):  out = act( (conv(in, w) + bias) * scale + shift + residual )
 *   in [b][h][w][cin]; w [cout][ks][ks][cin] (repacked OIHW); out [b][ho][wo][cout],
 *   ho = (h + 2*pad - ks)/stride + 1.  ks in {1,3}; pad = ks/2; cin % 32 == 0, or cin == 4
 *   (conv1a, RGB + zero channel).  scale/shift = folded eval BatchNorm (NULL = identity). */
int pram_conv2d_nhwc_f32(const float* in, int batch, int h, int w, int cin, const float* wgt,
                         const float* bias, const float* scale, const float* shift,
                         const float* residual, float* out, int cout, int ks, int stride, int relu,
                         void* stream);

/* fp16-operand variant (BASELINE C5): wgt16 = [cout][ks][ks][cin] in fp16, cin % 64 == 0. */
int pram_conv2d_nhwc_f16_f32(const float* in, int batch, int h, int w, int cin, const void* wgt16,
                             const float* bias, const float* scale, const float* shift,
                             const float* residual, float* out, int cout, int ks, int stride, int relu,
                             void* stream);

/* Split-fp16 ("x3") variant (see pram_linear_x3_f32): wgt_hi / wgt_lo = [cout][ks][ks][cin] * w_scale as two fp16
 * planes, cin % 32 == 0; the im2col rows are split (s = 16) while they are staged.  fp32 in, fp32 out. */
int pram_conv2d_nhwc_x3_f32(const float* in, int batch, int h, int w, int cin, const void* wgt_hi, const void* wgt_lo,
                            float w_scale, const float* bias, const float* scale, const float* shift,
                            const float* residual, float* out, int cout, int ks, int stride, int relu, void* stream);

/* Grouped 3x3 convolution of the ResBlock (groups = 32, 8 ch/group; nets/sfd2.py:98-99,113-115).
 * w [c][3][3][c/groups]. */
int pram_conv3x3_grouped_nhwc_f32(const float* in, int batch, int h, int w, int c, const float* wgt,
                                  const float* scale, const float* shift, float* out, int groups,
                                  int relu, void* stream);

/* pram_conv2d_nhwc_x3_f32 followed, in the same kernel, by F.normalize over the channels of every output pixel (x / max(||x||, 1e-12)):
 * SFD2's descriptor head (convDb -> normalize, reference nets/sfd2.py:331-333) without the second pass over the map.  cout <= 128 (one
 * workgroup holds a pixel's whole channel vector), cin % 32 == 0.  The squared sums are added in another order than
 * pram_l2norm_rows_f32 adds them: the two agree to rounding, not bit for bit. */
int pram_conv2d_nhwc_x3_l2norm_f32(const float* in, int batch, int h, int w, int cin, const void* wgt_hi, const void* wgt_lo,
                                   float w_scale, const float* bias, const float* scale, const float* shift, const float* residual,
                                   float* out, int cout, int ks, int stride, int relu, void* stream);

/* pram_conv2d_nhwc_x3_f32 whose result leaves as the split operand of the next split-fp16 layer instead of fp32: out_hi =
 * fp16(s y), out_lo = fp16(s y - out_hi) (s = the activation scale in force, 16 by default), [batch][ho][wo][cout] each — the same four bytes per value, and the consumer needs
 * neither registers nor vector instructions to stage it (LDS-DMA).  cin % 32 == 0, cout even; |y| >= 4095 is reported through the
 * range guard (status word).  Used between a ResBlock's first 1x1 and its grouped 3x3. */
int pram_conv2d_nhwc_x3_planes(const float* in, int batch, int h, int w, int cin, const void* wgt_hi, const void* wgt_lo,
                               float w_scale, const float* bias, const float* scale, const float* shift, const float* residual,
                               void* out_hi, void* out_lo, int cout, int ks, int stride, int relu, void* stream);

/* The grouped 3x3 (8 channels per group) on the split-fp16 path: a block-diagonal product on the matrix pipe (a 32-channel block
 * is four groups; a 16-deep k-step carries one tap of two groups), the input taken as the planes above, windows double-buffered in
 * LDS by DMA under the MFMAs of a persistent workgroup per CU, weights in registers — the vector-ALU kernel above is bound by the
 * LDS broadcasts of its weights.  w_hi / w_lo: the [c][3][3][8] weights * w_scale as fp16 planes; c % 64 == 0.  fp32-class (three
 * fp16 products, fp32 accumulation): ~3e-7 relative from pram_conv3x3_grouped_nhwc_f32. */
int pram_conv3x3_grouped_planes_x3_f32(const void* in_hi, const void* in_lo, int batch, int h, int w, int c, const void* w_hi,
                                       const void* w_lo, float w_scale, const float* scale, const float* shift, float* out,
                                       int groups, int relu, void* stream);

/* SFD2's first two convolutions (nets/sfd2.py:135-139,281-282: conv1a 3 -> 64 3x3 stride 1, conv1b 64 -> 64 3x3 stride 2, each bias
 * -> BN -> ReLU) in ONE launch on the split-fp16 path: the 480 x 640 x 64 map between them (1.26 GB for 16 frames: the largest
 * round trip of the step) never exists — a workgroup computes the (2 * 8 + 1) x (2 * 16 + 1) window of conv1a outputs it needs
 * from the image, in LDS.  img: NHWC4 fp32 (pram_image_to_nhwc4_f32), or with img_nchw3 != 0 the reference's own NCHW layout
 * [batch][3][h][w] (the repack kernel is then not needed); out [batch][(h - 1) / 2 + 1][(w - 1) / 2 + 1][64].
 * wa: conv1a's [64][3][3][4] weights flattened to [64][36], zero-padded to [64][48], * wa_scale, as (hi, lo) fp16 planes;
 * wb: conv1b's [64][3][3][64] weights * wb_scale as (hi, lo) planes (the operand of pram_conv2d_nhwc_x3_f32); b? / s? / t?: bias and
 * eval-mode BatchNorm scale / shift of each layer.  fp32-class, not bit-identical to the two-kernel form (whose conv1a is the
 * exact-fp32 MFMA kernel). */
int pram_sfd2_conv1_x3_f32(const float* img, int batch, int h, int w, const void* wa_hi, const void* wa_lo, float wa_scale,
                           const float* ba, const float* sa, const float* ta, const void* wb_hi, const void* wb_lo, float wb_scale,
                           const float* bb, const float* sb, const float* tb, float* out, int img_nchw3, void* stream);

/* NCHW [b][3][h][w] -> NHWC4 [b][h][w][4] (zero 4th channel) */
int pram_image_to_nhwc4_f32(const float* img, float* out, int batch, int h, int w, void* stream);
/* NHWC [b][h][w][c] -> NCHW [b][c][h][w] (for the reference-layout dict entries) */
int pram_nhwc_to_nchw_f32(const float* in, float* out, int batch, int h, int w, int c, void* stream);

/* softmax over 65 channels, drop dustbin, 8x8 depth-to-space (nets/sfd2.py:294-300; K2).
 * logits NHWC [b][hc][wc][65] -> score [b][8hc][8wc] */
int pram_score_map_f32(const float* logits, float* score, int batch, int hc, int wc, void* stream);

/* simple_nms (nets/sfd2.py:20-35; K3): 1 + 2 suppression rounds, window 2r+1 (r <= 4), exact equality.
 * workspace: pram_simple_nms_workspace_bytes (four fp32 images per frame). */
size_t pram_simple_nms_workspace_bytes(int batch, int h, int w);
int pram_simple_nms_f32(const float* score, float* nms, int batch, int h, int w, int radius,
                        void* workspace, void* stream);

size_t pram_select_keypoints_workspace_bytes(int batch, int h, int w, int max_keypoints);

/* threshold / min-keypoint fallback / remove_borders / top-k (nets/sfd2.py:306-329,38-50; K4).
 * Canonical order: if more than max_keypoints candidates survive, (score desc, flat index asc);
 * otherwise row-major.  fallback_ref: -1 = each image tests its own count (per-query
 * semantics), >= 0 = every image uses that image's count (reference tests element 0).
 * kpts [b][max_keypoints][2] (x,y) fp32, scores [b][max_keypoints], counts [b].
 * max_keypoints > 0: up to 8192 the top-k is sorted in LDS, beyond that in the workspace (the reference has no bound,
 * nets/sfd2.py:38-50); >= h*w = keep all (the reference's max_keypoints < 0, nets/sfd2.py:324: row-major order, never sorted). */
int pram_select_keypoints_f32(const float* nms, int batch, int h, int w, float conf_th,
                              int min_keypoints, int border, int max_keypoints, int fallback_ref,
                              float* kpts, float* scores, int* counts, void* workspace, void* stream);

/* sample_descriptors / ResNet4x.sample (nets/sfd2.py:53-64,348-369; K5): bilinear
 * grid_sample(align_corners=True, zero pad) of an NHWC map at keypoints, optional L2 norm,
 * optional per-pixel pre-normalisation of the map taps is NOT applied (pass a normalised map).
 * fmap [b][fh][fw][c]; kpts [b][n_max][2]; out [b][n_max][c].  c in {128, 256}. */
int pram_sample_nhwc_f32(const float* fmap, int batch, int fh, int fw, int c, const float* kpts,
                         const int* lens, int n_max, int s, int l2norm, float* out, void* stream);

/* F.normalize(x, dim=channel) of an NHWC map in place (nets/sfd2.py:333). rows = b*h*w */
int pram_l2norm_rows_f32(float* x, int rows, int cols, void* stream);

/* score lookup of ResNet4x.sample (nets/sfd2.py:367): out[b][i] = score_map[b*map_stride + y_i*w + x_i]
 * (map_stride = 0 reproduces the reference's score_map[0, ...]) */
int pram_score_lookup_f32(const float* score_map, long long map_stride, int h, int w, const float* kpts,
                          const int* lens, int batch, int n_max, float* out, void* stream);

/* ---------------------------------------------------------------- edges of the path ("next" rows) */

/* F.interpolate(mode='bilinear', align_corners=True) on planar maps [planes][h][w] -> [planes][oh][ow]
 * (score map of frames whose sides are not multiples of 8, nets/sfd2.py:301-303; multi-scale
 * extraction nets/sfd2.py:412-415). */
int pram_resize_bilinear_f32(const float* in, float* out, int planes, int h, int w, int oh, int ow,
                             void* stream);

/* Recogniser epilogue (Frame.add_segmentations, localization/frame.py:96-121): per token
 * seg_scores = softmax(logits) (optional output), non_bg_mask = seg_scores[0] < bg_threshold,
 * seg_ids = argmax(logits) - 1 (first occurrence), n_non_bg[b] = sum(non_bg_mask). */
int pram_seg_epilogue_f32(const float* logits, const int* lens, int batch, int n_max, int n_class,
                          float bg_threshold, float* seg_scores, int* seg_ids, int* non_bg_mask,
                          int* n_non_bg, void* stream);

/* torch.topk(x, k = cols, dim = -1) = full descending sort of each row (MultiMap3D.process_segmentations,
 * localization/multimap3d.py:348-350); ties in canonical (value desc, index asc) order. cols <= 1024. */
int pram_row_sort_desc_f32(const float* x, int ld, int rows, int cols, float* vals, long long* idx,
                           void* stream);

/* Landmark vote of MultiMap3D.process_segmentations (localization/multimap3d.py:348-379) on the sorted class lists of
 * pram_row_sort_desc_f32 (sorted_ids / sorted_vals [n][c], c <= 1024): rank by rank, the landmarks that tokens put at sorted
 * position k — background 0 and landmarks seen at an earlier rank skipped — ordered by token count (descending, ties by
 * ascending id), until `topk` are collected.  Outputs (device): win_sid / win_rank / win_count [topk], *n_win, the winners'
 * tokens in ascending order tokens [topk][n] (first win_count[w] of row w valid) and the mean of their rank-k scores. */
int pram_seg_vote(const long long* sorted_ids, const float* sorted_vals, int n, int c, int topk, int* win_sid, int* win_rank,
                  int* win_count, int* n_win, int* tokens, float* mean_score, void* stream);

/* Row top-2 of a batched matrix (largest = 1: sim.topk(2), nearest_neighbor.py:5-17; largest = 0:
 * topk(largest=False), singlemap3d.py:428).  v0/v1 best and second best values, i0 index of the best
 * (lowest index on ties).  x [batch][m_max][ld]. */
int pram_row_top2_f32(const float* x, int ld, long long stride, const int* row_lens, const int* col_lens,
                      int batch, int m_max, int n_max, int largest, float* v0, float* v1, long long* i0,
                      void* stream);

/* Projection-refinement matching (SingleMap3D.refine_pose_by_projection, singlemap3d.py:416-433):
 * dist[i][j] = sqrt(2 - 2*sim[i][j] + 1e-6) + (||kpts[i] - proj_uv[:, j]|| >= range ? 100 : 0); per query
 * row the two smallest distances d0 <= d1 and the index of d0.  sim [m][ld] (= q_descs @ ref_descs^T),
 * kpts [m][2], proj_uv [2][n] (u row then v row). */
int pram_proj_dist_top2_f32(const float* sim, int ld, const float* kpts, const float* proj_uv, int m, int n,
                            float range, float* d0, float* d1, long long* i0, void* stream);

/* The same with the projected points in float64 (what the reference actually holds: it projects in float64, so the pixel
 * error and its `>= 2 * threshold` test are float64): proj_uv = [2][ldu] doubles, first n columns valid. */
int pram_proj_dist_top2_f64uv(const float* sim, int ld, const float* kpts, const double* proj_uv, int ldu, int m, int n,
                              double range, float* d0, float* d1, long long* i0, void* stream);

/* Projection of the map points into the query camera + frustum test + ordered compaction
 * (SingleMap3D.refine_pose_by_projection, localization/singlemap3d.py:405-415), float64 like the reference:
 *   p = K (Tcw [X 1])[:3];  u = p0 / p2,  v = p1 / p2;  keep = 0 < p2 < 100, 0 <= u < im_w, 0 <= v < im_h.
 * xyz [n][3]; K [3][3], Tcw [4][4] row-major (device); uvd [3][n] = u, v, depth of EVERY point; mask [n] int32;
 * keep_idx [n] = original indices of the survivors in their original order, uv_keep [2][n] their (u, v) (first *count
 * columns valid); count = number of survivors (device int). */
int pram_project_points_f64(const double* xyz, const double* K, const double* Tcw, int n, double im_w, double im_h,
                            double* uvd, int* mask, int* keep_idx, double* uv_keep, int* count, void* stream);

/* Per-layer bookkeeping of the batched AdaGML loop (nets/adagml.py:352-380 per pair; here `pairs` pairs at once, no host read):
 * commits the pruned token counts (lens_new / n_below of pram_adagml_prune_f32; NULL on layer 0) for the still-active pairs,
 * evaluates check_if_stop (adagml.py:522-531: 1 - below / (m + n) > 0.95, fp32) — every active pair stops when `last` — and lets the
 * pairs that stop here commit lens_final / ind_final / stop_layer.  Token sets 0..pairs-1 are the query sets, pairs..2*pairs-1 the
 * reference sets.  Outputs: lens_out (current counts), lens_stop (counts of the pairs stopping at this layer, 0 elsewhere: the
 * ragged out_proj GEMM), lens_eff (counts of the pairs still active afterwards, 0 elsewhere: the next layer).  The *_in / *_out
 * state buffers must differ (ping-pong). */
int pram_adagml_layer_state(const int* active_in, int* active_out, const int* lens_in, int* lens_out, const int* lens_new,
                            const int* n_below, const float* num_points, int* tiny, int* stop_layer, int* lens_final,
                            int* lens_stop, int* lens_eff, const int* ind, int* ind_final, int pairs, int t_max, int layer,
                            int last, void* stream);
/* score4[token] = (col_self, col_cross, 0, 0): the two attention scores per token the pooling head reads (adagml.py:132). */
int pram_adagml_scores4_f32(const float* col_self, const float* col_cross, float* score4, long long tokens, void* stream);
/* pram_adagml_prune_f32 with the logits at a stride (column 0 of the pooling head's padded [rows][4] output). */
int pram_adagml_prune_ld_f32(const float* conf_logit, int ld_logit, float thr, int n_min_tokens, const int* lens_in,
                             const float* x_in, const float* cos_in, const float* sin_in, const int* ind_in,
                             float* x_out, float* cos_out, float* sin_out, int* ind_out, int* lens_out,
                             int* n_below, float* conf_out, int* row_map, int sets, int t_max, int ldx, void* stream);

/* ---------------------------------------------------------------- MLP tail as a GEMM pair with the LayerNorm between them
 * Linear -> LayerNorm -> GELU -> Linear (+ residual): the tail of every attention block (nets/segnetvit.py:87-95,104-106;
 * nets/gml.py:118-126,151-162), SegNetViT's seg / sc heads (segnetvit.py:157-172) — without a stand-alone LayerNorm + GELU pass
 * over the hidden layer.  The first GEMM's weights are centred over its outputs on the host (w[j] - mean_j w[j], b[j] - mean b:
 * LayerNorm is invariant to the shift), so its output is h - mean(h); it also writes the rows' sums of squares, one partial per
 * 64-column block (row_ssq [parts][m], parts = pram_linear_x3_ssq_parts(m, n, k0 + k1) = ceil(n / 64); the consumer adds them in
 * ascending order, so the statistics do not depend on the tile configuration a launch picked).  The second GEMM applies
 * GELU(hidden * rstd * gamma + beta), rstd = 1 / sqrt(sum_p row_ssq[p][row] / k + eps), to its A operand while staging it
 * (GELU = t Phi(t) through a degree-7 fit of the Gaussian tail, |error| <= 7.5e-8 |t|).  Split-fp16 path; lens / t_pad as pram_linear_x3_ragged_f32. */
int pram_linear_x3_ssq_parts(int m, int n, int k);
int pram_linear_x3_ssq_f32(const float* a0, int lda0, int k0, const float* a1, int lda1, int k1, const void* w_hi, const void* w_lo,
                           float w_scale, const float* bias, float* out, int ldo, float* row_ssq, int m, int n,
                           const int* lens, int t_pad, void* stream);
int pram_linear_x3_lngelu_f32(const float* hidden, int ldh, int k, const void* w_hi, const void* w_lo, float w_scale, const float* bias,
                              const float* residual, int ldr, float* out, int ldo, int m, int n, const float* ln_ssq, int parts,
                              const float* gamma, const float* beta, float eps, const int* lens, int t_pad, void* stream);

/* ---------------------------------------------------------------- fp16 MFMA path (BASELINE C5) with fp16 intermediates in HBM
 * The single-product path rounds every GEMM / attention operand to fp16 while it stages it; here the producers write those
 * operands as fp16 (2 bytes per element of HBM traffic): q | k as fp16 rows and v straight into the transposed key-permuted layout
 * (pram_linear_f16_qkv_h16), the attention context (pram_attention_h16t_h16), the MLP's hidden layer (pram_linear_f16_ssq_h16: host-
 * centred weights, rows' sums of squares from the fp32 values), normalised and GELU-ed while the second GEMM stages it
 * (pram_linear_f16_lngelu_f32).  The residual stream stays fp32.  This path's own tolerance. */
int pram_linear_f16_qkv_h16(const float* a0, int lda0, int k0, const void* w16, const float* bias, void* out16, int ldo16, void* vt16,
                            int vt_col0, int heads, int t_seq, int m, int n, int flags, const float* rot_cos, const float* rot_sin,
                            int rot_cols, const int* lens, void* stream);
int pram_attention_h16t_h16(const void* q16, int ldq, const void* k16, int ldk, const void* vt16, void* out16, int ldo16,
                            float* lse2, const int* q_lens, const int* k_lens, int batch, int heads, int m_max,
                            int n_max, float scale, int kv_shift, void* stream);
int pram_linear_f16_ssq_h16(const float* a0, int lda0, int k0, const void* a1_16, int lda1, int k1, const void* w16, const float* bias,
                            void* out16, int ldo16, float* row_ssq, int m, int n, void* stream);
int pram_linear_f16_lngelu_f32(const void* hidden16, int ldh, int k, const void* w16, const float* bias, const float* residual, int ldr,
                               float* out, int ldo, int m, int n, const float* ln_ssq, int parts, const float* gamma,
                               const float* beta, float eps, void* stream);

/* Fixed-size per-query result record rec [batch][k][6] fp32 = x, y, score, landmark id, match index, match score — what the
 * single all-gather of the query-sharded job carries (the reference hands the same fields to its pose solver,
 * localization/singlemap3d.py:155-170).  landmark (int32 [batch][k]) and matches0 / mscores0 (int64 / fp32 [batch][km], km <= k:
 * only the first km keypoints of a query went through the matcher) may be NULL: 0, and -1 / 0 beyond km. */
int pram_pack_record_f32(const float* kpts, const float* scores, const int* landmark, const long long* matches0,
                         const float* mscores0, int batch, int k, int km, float* rec, void* stream);

/* Frame staging — replaces, per query frame, `torch.from_numpy(img / 255).permute(2, 0, 1).cuda().float()` followed by
 * `tvf.Normalize(mean, std)` (localization/loc_by_rec_online.py:98-106, nets/sfd2.py:14-17): frames_hwc3 = uint8 [batch][h][w][3] as
 * cv2.imread delivers them (channel order untouched, like the reference), lut = fp32 [3][256] with lut[c][v] = ((v / 255) - mean[c]) /
 * std[c] computed by the caller with the reference's own operations, out = fp32 [batch][3][h][w].  h * w a multiple of 4. */
int pram_stage_frames_u8(const void* frames_hwc3, const float* lut, float* out_nchw, int batch, int h, int w, void* stream);

/* dst[0 .. count) <- value, 32-bit words, on the stream (torch.zeros / torch.full of the host-side glue without a framework kernel). */
int pram_fill_u32(void* dst, unsigned int value, size_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PRAM_HIP_H */
