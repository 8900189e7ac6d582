/* ctclip_hip.h -- C ABI of libctclip_hip.so: the MI355X (gfx950) kernels of the CT-CLIP training hot path.
 *
 * The reference (ibrahimethemhamamci/CT-CLIP) is pure Python on PyTorch: its "FFI" for this path is the set of
 * torch / HuggingFace / vector-quantize-pytorch calls listed in SURVEY.md section 8(a).  Each entry point below
 * replaces one of those calls; the comment cites the reference line it stands in for (paths relative to the
 * reference repository: attention.py and ctvit.py live in transformer_maskgit/transformer_maskgit/, ct_clip.py in
 * CT_CLIP/ct_clip/, CTCLIPTrainer.py in scripts/).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to a contiguous (or row-strided, where an ld* argument exists) buffer;
 *   - dtype codes: 0 = float32, 1 = bfloat16 (raw 16-bit);  statistics, biases, gradients of parameters are float32;
 *   - the library never allocates, never synchronises and only uses the stream it is handed;
 *   - return value 0 = success; -1 bad argument, -2 unsupported shape/dtype, -3 workspace too small,
 *     -(1000 + hipError_t) launch failure; ctclip_last_error() gives the text (thread-local);
 *   - "ACCUMULATED" outputs are += (zero them first if a plain result is wanted).
 * The ctypes binding used by the Python host is ct_clip_amd/_lib.py; INTEGRATION.md shows the reference-side stub.
 */
#ifndef CTCLIP_HIP_H
#define CTCLIP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct ihipStream_t* hipStream_t;

/* per-(sequence, head) transposed copy [seq][h][d][Lp] used as MFMA A-operands by the attention kernels.  [replaces the rearrange 'b n (h d) -> b h n d' of attention.py:145 and HF BertSelfAttention.transpose_for_scores] */
int ctclip_head_transpose(const void* x, void* xt, int nseq, int H, int L, int Lp, int D, int64_t ldx, int dtype, hipStream_t stream);

/* l2norm(q)*q_scale / l2norm(k)*k_scale per head (attention.py:152-154). */
int ctclip_qk_norm_fwd(const void* x, const float* scale_vec, void* y, float* inv, int64_t M, int H, int D, int64_t ldx, int64_t ldy, int dtype, hipStream_t stream);

/* bytes of workspace ctclip_qk_norm_bwd needs for the per-workgroup partial sums of the learned-scale gradient (deterministic two-stage sum). [workspace query of ctclip_qk_norm_bwd (autograd through attention.py:152-156)] */
int64_t ctclip_qk_norm_bwd_workspace(int64_t M, int H, int D);

/* backward of the above; dscale (D) ACCUMULATED (+=) through per-workgroup partials in `workspace` (>= ctclip_qk_norm_bwd_workspace bytes), summed in a fixed order. [replaces autograd through l2norm + the learned scales, attention.py:22-23,152-156] */
int ctclip_qk_norm_bwd(const void* dy, const void* x, const float* inv, const float* scale_vec, void* dx, float* dscale, int64_t M, int H, int D, int64_t lddy, int64_t ldx, int64_t lddx, int dtype, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* softmax(scale*q k^T + bias[h] + keymask[seq]) v (attention.py:156-178; HF BertSelfAttention). */
int ctclip_attn_fwd(const void* q, const void* k, const void* vt, const float* bias, int bias_gh, int bias_gw, const float* keymask, void* out, float* lse, int nseq, int H, int L, int Lp, int D, int64_t ldq, int64_t ldk, int64_t ldo, float scale, float dropout_p, uint64_t dropout_seed, int dtype, hipStream_t stream);

/* bytes of workspace ctclip_attn_bwd needs when dbias is requested (per-split partial dBias slabs). [workspace query of ctclip_attn_bwd (autograd through attention.py:158-178; HF BertSelfAttention)] */
int64_t ctclip_attn_bwd_workspace(int nseq, int H, int L);

/* backward of the above: dq, dk, dv and (optional, ACCUMULATED) dbias (H,L,L). [replaces autograd through einsum / softmax / dropout / einsum, attention.py:158-178 and HF BertSelfAttention.forward] */
int ctclip_attn_bwd(const void* q, const void* k, const void* v, const void* qt, const void* kt, const void* o, const void* dout, const void* dot, const float* lse, const float* bias, int bias_gh, int bias_gw, const float* keymask, float* delta, void* dq, void* dk, void* dv, float* dbias, int nseq, int H, int L, int Lp, int D, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq, int64_t lddk, int64_t lddv, float scale, float dropout_p, uint64_t dropout_seed, int dtype, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* 1 when the second-generation cosine-attention kernels (csrc/attn2*.hip) serve the shape: bf16, d_head 32, L a multiple of 32 in [64, 1024]; with a position-bias table (has_bias) the grid gh x gw must equal L, gw % 8 == 0, (2gh-1)(2gw-1) <= 4096 classes.  attention.py:145-178. */
int ctclip_attn2_supported(int H, int L, int D_, int bias_gh, int bias_gw, int has_bias);

/* attention.py:152-160 for the second-generation kernels: q (M, H*32), k, v row-major bf16 -> head-planar [H][M][32] operands q~ = l2norm(q) * q_scale * (scale * log2 e), k^ = l2norm(k) * k_scale, v (copy) + the inverse row norms qinv, kinv (M, H) f32 that the backward needs. */
int ctclip_attn2_prep(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldk, int64_t ldv, const float* q_scale, const float* k_scale, float scale, void* qh, void* kh, void* vh, float* qinv, float* kinv, int64_t M, int H, hipStream_t stream);

/* softmax(q~ k^T + bias) v on head-planar operands (attention.py:162-178).  tab: the (nclass, H) f32 continuous-position-bias table of attention.py:257-276 (nclass = (2 bias_gh - 1)(2 bias_gw - 1)) or null; out (M, ldo) row-major bf16; lse2 [H][M] f32 log2-domain log-sum-exp for the backward. */
int ctclip_attn2_fwd(const void* qh, const void* kh, const void* vh, const float* tab, int bias_gh, int bias_gw, const float* q_scale, const float* k_scale, float scale, void* out, int64_t ldo, float* lse2, int nseq, int H, int L, hipStream_t stream);

/* bytes of workspace for ctclip_attn2_bwd: dO' / delta' planes passed from the query pass to the key pass, and (with a bias table) the per-workgroup dBias slabs and bins of the deterministic fold. [workspace query of ctclip_attn2_bwd (autograd through attention.py:152-178)] */
int64_t ctclip_attn2_bwd_workspace(int nseq, int H, int L, int bias_gh, int bias_gw);

/* backward of ctclip_attn2_fwd: head-planar dq~, dk^, dv [H][M][32] bf16 (the l2norm / scale backward is ctclip_attn2_unprep) and, when dtab is non-null, the gradient of the position-bias table (nclass, H) f32, ACCUMULATED (+=) in a fixed summation order. [replaces autograd through attention.py:158-178 incl. the position-bias gradient of attention.py:257-276] */
int ctclip_attn2_bwd(const void* qh, const void* kh, const void* vh, const float* tab, int bias_gh, int bias_gw, const float* q_scale, const float* k_scale, float scale, const void* o, int64_t ldo, const void* dout, int64_t lddo, const float* lse2, void* dqh, void* dkh, void* dvh, float* dtab, int nseq, int H, int L, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* The position-bias table gradient of ctclip_attn2_bwd as a call of its own (the scatter-add of dBias over the (2h-1)(2w-1) offset classes that autograd performs through `ContinuousPositionBias`, attention.py:257-276): dtab (ncls, H) OVERWRITTEN.  `workspace` must be the SAME buffer (>= ctclip_attn2_bwd_workspace(nseq, H, L, bias_gh, bias_gw)) a preceding ctclip_attn2_bwd(..., dtab = NULL, ...) of the same problem was given: its first two regions hold dO' and delta' of the query pass.  The host runs this call on a side stream, under the rest of the layer's backward. */
int ctclip_attn2_bwd_dbias(const void* qh, const void* kh, const void* vh, const float* tab, int bias_gh, int bias_gw, const float* q_scale, const float* k_scale, float scale, const float* lse2, float* dtab, int nseq, int H, int L, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* bytes of workspace ctclip_attn2_unprep needs for the partial sums of the q_scale / k_scale gradients. [workspace query of ctclip_attn2_unprep (autograd through attention.py:152-156)] */
int64_t ctclip_attn2_unprep_workspace(void);

/* bytes of workspace ctclip_attn2_bwd_tok needs (ctclip_attn2_bwd's regions + the per-workgroup k_scale partials). [workspace query of ctclip_attn2_bwd_tok (autograd through attention.py:152-178)] */
int64_t ctclip_attn2_bwd_tok_workspace(int nseq, int H, int L, int bias_gh, int bias_gw);

/* ctclip_attn2_bwd (backward of the cosine attention, attention.py:145-178) with the k / v half of ctclip_attn2_unprep folded into the slab key pass: dqh head-planar as before, dk (M, lddk) / dv (M, lddv) ROW-MAJOR with the l2norm backward of k applied (kinv = the forward's inverse norms (M, H)), dk_scale (32) ACCUMULATED, dtab as in ctclip_attn2_bwd (NULL: ctclip_attn2_bwd_dbias later from the same workspace).  Follow with ctclip_attn2_unprep_q.  CTCLIP_EUNSUPPORTED when the slab kernels do not serve the shape. */
int ctclip_attn2_bwd_tok(const void* qh, const void* kh, const void* vh, const float* tab, int bias_gh, int bias_gw, const float* q_scale, const float* k_scale, float scale, const void* o, int64_t ldo, const void* dout, int64_t lddo, const float* lse2, const float* kinv, void* dqh, void* dk, int64_t lddk, void* dv, int64_t lddv, float* dk_scale, float* dtab, int nseq, int H, int L, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* the q half of ctclip_attn2_unprep: head-planar dq^ -> row-major dq (M, lddq) through the l2norm backward of q (attention.py:152); dq_scale (32) ACCUMULATED; workspace >= ctclip_attn2_unprep_workspace(). */
int ctclip_attn2_unprep_q(const void* dqh, const void* qh, const float* qinv, const float* q_scale, float scale, void* dq, int64_t lddq, float* dq_scale, int64_t M, int H, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* backward of ctclip_attn2_prep: head-planar dq~, dk^, dv -> row-major dq (M, lddq), dk, dv bf16 through the l2norm backward; dq_scale, dk_scale (32) f32 are ACCUMULATED (+=), summed in a fixed order. [replaces autograd through l2norm + the learned scales + the head split, attention.py:145-156] */
int ctclip_attn2_unprep(const void* dqh, const void* dkh, const void* dvh, const void* qh, const void* kh, const float* qinv, const float* kinv, const float* q_scale, const float* k_scale, float scale, void* dq, void* dk, void* dv, int64_t lddq, int64_t lddk, int64_t lddv, float* dq_scale, float* dk_scale, int64_t M, int H, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* 1 when ctclip_attn2_bwd_fused serves the shape: bf16, d_head 32, L % 32 == 0, 256 <= L <= 576 (the one-pass kernel keeps Q~, dO'', the f32 dQ accumulators and the class table of one (sequence, head) in 160 KiB of LDS); with a bias table gw % 8 == 0. [capability query; attention.py:145-178] */
int ctclip_attn2_bwd_fused_supported(int nseq, int H, int L, int D_, int bias_gh, int bias_gw, int has_bias);

/* bytes of workspace ctclip_attn2_bwd_fused needs (staged table, per-workgroup table / scale partials, parked key-block accumulators); 0 when the shape is not served. [workspace query of ctclip_attn2_bwd_fused (autograd through attention.py:152-178)] */
int64_t ctclip_attn2_bwd_fused_workspace(int nseq, int H, int L, int bias_gh, int bias_gw);

/* Backward of ctclip_attn2_fwd in ONE pass over the score tiles: row-major dq (M, lddq), dk (M, lddk), dv (M, lddv) w.r.t. the projections (l2norm / learned-scale backward applied), dq_scale / dk_scale (32) ACCUMULATED, dtab (ncls, H) OVERWRITTEN when non-null (fixed-point scatter: deterministic). qh / kh / vh head-planar operands and qinv / kinv (M, H) of ctclip_attn2_prep or ctclip_gemm_headnorm; o / dout (M, ldo / lddo); lse2 [H][M]. CTCLIP_EUNSUPPORTED when the shape is not served (then ctclip_attn2_bwd_tok). [torch autograd through F.normalize, q_scale / k_scale, einsum('b h i d, b h j d'), + attn_bias, softmax, einsum('b h i j, b h j d') of attention.py:152-178, and the gather of the bias table attention.py:257-276] */
int ctclip_attn2_bwd_fused(const void* qh, const void* kh, const void* vh, const float* tab, int bias_gh, int bias_gw, const float* q_scale, const float* k_scale, float scale, const void* o, int64_t ldo, const void* dout, int64_t lddo, const float* lse2, const float* qinv, const float* kinv, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, float* dq_scale, float* dk_scale, float* dtab, int nseq, int H, int L, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* 1 when ctclip_attn_short_* serves the shape (bf16, d_head 32, 1 <= L <= 32 tokens per sequence, no bias, no mask): CTViT's temporal transformer (attention.py:145-178 on (b h w, t, d) sequences, ctvit.py:297-305). */
int ctclip_attn_short_supported(int L, int D, int dtype);

/* out[(s L + i), h*32 + :] = softmax_j(scale * <l2norm(q_i) q_scale, l2norm(k_j) k_scale>) v_j for nseq sequences of L tokens, H heads (attention.py:145-178): q (nseq*L, ldq >= H*32), kv (nseq*L, ldkv >= 2*H*32) = [k | v], out (nseq*L, ldo) row-major bf16; q_scale, k_scale (32) f32.  One wave per (sequence, head); nothing is saved for the backward. */
int ctclip_attn_short_fwd(const void* q, int64_t ldq, const void* kv, int64_t ldkv, const float* q_scale, const float* k_scale, void* out, int64_t ldo, int nseq, int H, int L, float scale, hipStream_t stream);

/* bytes of workspace ctclip_attn_short_bwd needs (one 64-float row of learned-scale gradient partials per workgroup). [workspace query of ctclip_attn_short_bwd (autograd through attention.py:145-178 in the temporal transformer, ctvit.py:187,303)] */
int64_t ctclip_attn_short_bwd_workspace(int nseq, int H);

/* backward of ctclip_attn_short_fwd from q, kv and dout alone (the 32 x 32 softmax is recomputed): dq (nseq*L, lddq), dkv (nseq*L, lddkv) = [dk | dv] bf16 are overwritten; dq_scale, dk_scale (32) f32 are ACCUMULATED (+=) when non-null, in a fixed summation order. [replaces autograd through attention.py:145-178 in the temporal transformer (ctvit.py:187,303)] */
int ctclip_attn_short_bwd(const void* q, int64_t ldq, const void* kv, int64_t ldkv, const float* q_scale, const float* k_scale, const void* dout, int64_t lddo, void* dq, int64_t lddq, void* dkv, int64_t lddkv, float* dq_scale, float* dk_scale, int nseq, int H, int L, float scale, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* thread-local message of the last failing call. [C-ABI plumbing: the reference has no FFI of its own, it raises Python exceptions] */
const char* ctclip_last_error(void);

/* ABI version (1). [C-ABI plumbing, no reference counterpart] */
int ctclip_abi_version(void);

/* "gfx950".  [C-ABI plumbing, no reference counterpart] */
const char* ctclip_target_arch(void);

/* state: two 64-bit words in DEVICE memory { dropout seed offset, optimiser step } or NULL (off, the default). While set, ctclip_dropout / ctclip_relu_dropout / ctclip_attn_fwd / ctclip_attn_bwd add state[0] to the seed they are given and ctclip_adam_step takes its step count from state[1], read when the kernels RUN: a captured hipGraph of the training step replays with fresh dropout masks and the right bias correction. Process-wide; the caller keeps the memory alive. [no reference counterpart: the reference is eager PyTorch (CTCLIPTrainer.py:249-264), its RNG and step counters live on the host] */
int ctclip_set_step_state(const void* state);

/* state[0] += an odd 64-bit constant, state[1] += 1 in one launch (put it first in the captured step). [no reference counterpart: see ctclip_set_step_state] */
int ctclip_advance_step_state(void* state, hipStream_t s);

/* x + PEG(x): causal-padded depthwise Conv3d 3x3x3 (attention.py:56-84,324). */
int ctclip_peg_fwd(const void* x, const float* w, const float* bias, void* y, int64_t B, int D1, int D2, int D3, int C, int dtype, hipStream_t stream);

/* ctclip_peg_fwd (attention.py:63-84 + the residual of :324) on the compensated residual stream: s = x + e_in (may be NULL) + conv(x) + bias in f32, y = bf16(s), e_out = bf16(s - y).  bf16 grids the LDS-marching kernels serve; CTCLIP_EUNSUPPORTED otherwise. */
int ctclip_peg_fwd_comp(const void* x, const float* w, const float* bias, const void* e_in, void* y, void* e_out, int64_t B, int D1, int D2, int D3, int C, int dtype, hipStream_t stream);

/* bytes of workspace ctclip_peg_bwd needs when dw is requested (per-workgroup partial weight gradients of the deterministic two-stage sum). [workspace query of ctclip_peg_bwd (autograd through attention.py:63-84,324)] */
int64_t ctclip_peg_bwd_workspace(int64_t B, int D1, int D2, int C);

/* backward of the above; dw (C,27) / db (C) ACCUMULATED (+=), may be NULL; with dw, `workspace` holds >= ctclip_peg_bwd_workspace bytes of per-workgroup partials (two-stage sum in a fixed order, no atomics). [replaces autograd through nn.Conv3d(groups=dim) + F.pad + the residual, attention.py:63-84,324] */
int ctclip_peg_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, float* db, int64_t B, int D1, int D2, int D3, int C, int dtype, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* ClassFine / CT-LiPro head (scripts/ct_lipro_train.py:33-36): out = relu(x) * dropout_mask / (1 - p) when dy is null, else the gradient dy * mask / (1 - p) * [x > 0]; f32, n % 4 == 0; the mask is Philox(seed, element, stream_id) as in ctclip_dropout. */
int ctclip_relu_dropout(const float* x, const float* dy, float* out, int64_t n, float p, uint64_t seed, uint32_t stream_id, hipStream_t s);

/* torch.nn.BCEWithLogitsLoss(pos_weight) with mean reduction (ct_lipro_train.py:84,104): logits, targets (B, C) f32, pos_weight (C) f32 or null -> loss (1) and, when non-null, dlogits (B, C) = d loss / d logits. */
int ctclip_bce_logits(const float* logits, const float* targets, const float* pos_weight, float* loss, float* dlogits, int B, int C, hipStream_t s);

/* VocabFine objective of one prompt group (scripts/ct_vocabfine_train.py:112-121): sims (n, 2) f32 = (true prompt, false prompt) similarities -> softmax over each pair, MSE against (1, 0): loss (1), dsims (n, 2) or null. */
int ctclip_pair_softmax_mse(const float* sims, float* loss, float* dsims, int n, hipStream_t s);

/* CTCLIP.forward without return_loss (ct_clip.py:771,796,805-807): sims[p] = <l2norm(text_p), l2norm(image_p)> * exp(temperature) with broadcasting (nt == ni, or one side has one row: two prompts against one volume).  dsims null: forward, sims (max(nt, ni)) f32.  dsims non-null: backward, dtext (nt, D), dimage (ni, D), dtemp (1) are overwritten. */
int ctclip_latent_similarity(const float* text, const float* image, const float* temperature, const float* dsims, float* sims, float* dtext, float* dimage, float* dtemp, int nt, int ni, int D, hipStream_t s);

/* nn.Linear / F.linear forward, grad-input and grad-weight (attention.py:48,51,119,120,125; ctvit.py:173; ct_clip.py:549,762; HF BERT dense layers). C = alpha*op(A) op(B)^T + bias + residual (+C). With a split over K > 1 (split_k > 1, or auto on f32 output) a workspace of ctclip_gemm_workspace bytes is required (CTCLIP_EWORKSPACE otherwise). */
int ctclip_gemm(const void* A, const void* B, void* C, const float* bias, const void* residual, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int a_kc, int b_kc, int in_dtype, int out_dtype, int res_dtype, int accumulate, int split_k, float alpha, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* bytes of workspace ctclip_gemm needs (split-K partial slabs; 0 when the split is 1); split_k <= 0 = auto. REQUIRED whenever the split is > 1: ctclip_gemm then returns CTCLIP_EWORKSPACE without it (round 3 removed the float-atomic fallback: every reduction is deterministic). [workspace query of ctclip_gemm (nn.Linear, attention.py:48,51,119,120,125)] */
int64_t ctclip_gemm_workspace(int64_t M, int64_t N, int64_t K, int in_dtype, int split_k);

/* bytes of workspace ctclip_gemm_argmax needs. [workspace query of ctclip_gemm_argmax (the code search of vector-quantize-pytorch called at ctvit.py:403)] */
int64_t ctclip_gemm_argmax_workspace(int64_t M, int64_t N);

/* vector_quantize_pytorch CosineSimCodebook: argmax_c <x_n, e_c> (ctvit.py:403) without materialising the distance matrix. */
int ctclip_gemm_argmax(const void* A, const void* B, int64_t* out_idx, float* out_val, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int in_dtype, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* The vector quantiser's code search (vector_quantize_pytorch CosineSimCodebook.forward, call site ctvit.py:403) on the RAW bf16 tokens: arg-max over codes c of A[m] . (hi_c + lo_c), A (M, K) the tokens as they are, B2 (C, 2 K) the unit codebook's rows [hi | lo] (ctclip_l2norm_split3 order 2).  The arg-max does not see a row's norm and a bf16 token has no low part: two products per (token, code, dim) instead of the three of the [hi|hi|lo] x [hi|lo|hi] form, no expanded copy of the tokens (the kernel reads A twice along k).  out_val = the winning dot product (not normalised by |A[m]|).  bf16, K % 64 == 0, ceil(M / 256) x ceil(C / 256) >= 160 tiles; other shapes: CTCLIP_EUNSUPPORTED (use ctclip_gemm_argmax).  workspace >= ctclip_gemm_argmax_workspace(M, C). */
int ctclip_gemm_argmax_hilo(const void* A, const void* B2, int64_t* out_idx, float* out_val, int64_t M, int64_t C, int64_t K, int64_t lda, int64_t ldb, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* FeedForward[1].weight (2 * inner, K) f32 = [x rows | gate rows] (attention.py:48) -> the bf16 operand of ctclip_gemm_geglu: (2 * hp, ldo >= K) with row 8 q + r = x row 4 q + r and row 8 q + 4 + r = gate row 4 q + r (r < 4); rows of padded features (>= inner) are zero. */
int ctclip_geglu_weight_interleave(const float* w, void* out, int inner, int hp, int K, int64_t ldo, hipStream_t stream);

/* Feed-forward in-projection + GEGLU in one launch (attention.py:39-48): u (M, ldu >= 2 hp) = [x | gate] = A B^T in the layout ctclip_geglu_bwd reads, g (M, ldg >= hp) = x * gelu_erf(gate); A (M, lda) bf16, B = the output of ctclip_geglu_weight_interleave.  Returns -2 (unsupported) when M or 2 hp is not a multiple of 256 or the shape does not fill the chip: the caller then runs ctclip_gemm + ctclip_geglu_fwd. */
int ctclip_gemm_geglu(const void* A, const void* B, void* U, void* G, int64_t M, int hp, int64_t K, int64_t lda, int64_t ldb, int64_t ldu, int64_t ldg, int dtype, hipStream_t stream);

/* Backward of ctclip_gemm_geglu through the GEGLU by RECOMPUTATION (replaces torch autograd through `x * F.gelu(gate)` of attention.py:39-42 and the activation it would keep): the same GEMM A B^T rebuilds (x, gate) in f32 and the epilogue writes dU (M, lddu >= 2 hp) = [dG * gelu(gate) | dG * x * gelu'(gate)] from dG (M, lddg >= hp), bf16.  The forward then stores no u (pass U = NULL to ctclip_gemm_geglu) and the streaming ctclip_geglu_bwd pass is not needed.  Same eligibility as ctclip_gemm_geglu (CTCLIP_EUNSUPPORTED otherwise). */
int ctclip_gemm_geglu_bwd(const void* A, const void* B, const void* dG, void* dU, int64_t M, int hp, int64_t K, int64_t lda, int64_t ldb, int64_t lddg, int64_t lddu, int dtype, hipStream_t stream);

/* Backward of the feed-forward block between FeedForward[4] and the GEGLU in ONE launch (replaces torch autograd through `Linear(inner, dim)` and `x * F.gelu(gate)`, attention.py:39-51): dU (M, lddu >= 2 hp) = [dg * gelu(gate) | dg * x * gelu'(gate)] where dg = dY W_out exists only in the accumulators (A = dY (M, K = model width) bf16, B = W_out^T (hp, ldb >= K), hidden feature j in row j) and U = [x | gate] (M, ldu >= 2 hp) is what ctclip_gemm_geglu stored.  No dg tensor, no ctclip_geglu_bwd pass.  CTCLIP_EUNSUPPORTED when the shape does not fill whole 256-row tiles / 128-column halves (caller: ctclip_gemm + ctclip_geglu_bwd). */
int ctclip_gemm_dgeglu(const void* A, const void* B, const void* U, void* dU, int64_t M, int hp, int64_t K, int64_t lda, int64_t ldb, int64_t ldu, int64_t lddu, int dtype, hipStream_t stream);

/* a residual add of the transformer (attention.py:326,331: `x = attn(x) + x`, `x = ff(x) + x`) on a COMPENSATED bf16 residual stream: s = A B^T + residual + comp in f32 (comp = the residue of `residual`), C = bf16(s), E = bf16(s - C) (the rounding residue the next add takes back in): replaces nn.Linear + the torch add, whose bf16 storage rounds the stream at every add.  bf16, M % 256 == 0, N % 128 == 0, K % 64 == 0; CTCLIP_EUNSUPPORTED otherwise. */
int ctclip_gemm_residual_comp(const void* A, const void* B, void* C, void* E, const void* residual, const void* comp, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int dtype, hipStream_t stream);

/* the q and k|v projections of the spatial attention with the operand layout of the attention kernels written by the GEMM (replaces nn.Linear at attention.py:141-143 + the head split / l2norm / learned scale of :145-154, i.e. nn.Linear + ctclip_attn2_prep): A (M, K) bf16, B (nsec * 256, K) bf16; per 256-column section s (8 heads x 32): inv_s != NULL -> out_s[h][m][d] = bf16(a_m . b_n) / max(|head row|, 1e-12) * scale_s[d] * mult_s, inv_s[m * 8 + h] = the inverse norm; inv_s == NULL -> head-planar copy (v).  CTCLIP_EUNSUPPORTED unless bf16, M % 256 == 0, K % 64 == 0, 1 <= nsec <= 3. */
int ctclip_gemm_headnorm(const void* A, const void* B, int64_t M, int nsec, int64_t K, int64_t lda, int64_t ldb, void* out0, float* inv0, const float* scale0, float mult0, void* out1, float* inv1, const float* scale1, float mult1, void* out2, float* inv2, const float* scale2, float mult2, int dtype, hipStream_t stream);

/* Tuning / test knob: which epilogue families of ctclip_gemm / ctclip_gemm_geglu / ctclip_gemm_dgeglu run on the two-workgroups-per-CU kernel (bit 0 plain / residual, bit 1 GEGLU forward, bit 2 GEGLU backward; negative = environment CTCLIP_GEMM_NT2 / built-in default); returns the previous mask; results are bit-identical either way. [no reference counterpart: torch picks its GEMM kernels inside ATen for nn.Linear, attention.py:48,51,119,120,125] */
int ctclip_gemm_nt2_select(int mask);

/* Weight AND bias gradient of a Linear layer in one launch: dW (n_out x k_in, f32, row stride ldw) (+)= dy^T x and db (n_out, f32) (+)= the column sums of dy; dy (T, n_out) and x (T, k_in) bf16 token-major (row strides lddy, ldx: column views allowed); the column sums ride the dW GEMM's A fragments (one more MFMA per fragment against ones): no second pass over dy, no reduce launch; deterministic.  CTCLIP_EUNSUPPORTED when the shape is not served (callers compose ctclip_gemm (0,0) + ctclip_colsum). [replaces autograd through nn.Linear WITH bias: HF BertSelfAttention query / key / value, BertSelfOutput.dense, BertIntermediate.dense, BertOutput.dense -- the text tower the reference calls at ct_clip.py:685-686] */
int ctclip_gemm_dw_db(const void* dy, const void* x, float* dW, float* db, int64_t T, int64_t n_out, int64_t k_in, int64_t lddy, int64_t ldx, int64_t ldw, int accumulate, hipStream_t stream);

/* bytes of workspace ctclip_visual_latent_fwd needs (split-K partial sums of the 294912-wide projection, summed in a fixed order). [workspace query of ctclip_visual_latent_fwd (to_visual_latent, ct_clip.py:549,771)] */
int64_t ctclip_visual_latent_fwd_workspace(int Bm, int N, int64_t K);

/* CTCLIP.to_visual_latent: Linear(h*w*dim -> dim_latent, no bias) at M = batch (ct_clip.py:564,767); `workspace` >= ctclip_visual_latent_fwd_workspace bytes (split-K partials, fixed summation order). */
int ctclip_visual_latent_fwd(const void* X, const void* W, float* Y, int Bm, int N, int64_t K, int dtype, void* workspace, int64_t workspace_bytes, hipStream_t s);

/* backward of the above (dX and dW); Bm 1..8 rows per call, 1..24 in f32 (the VocabFine step's 18 pooled vectors in one pass over the weight and its gradient). [replaces autograd through to_visual_latent = nn.Linear(dim_image, dim_latent, bias=False), ct_clip.py:549,771] */
int ctclip_visual_latent_bwd(const float* dY, const void* X, const void* W, void* dX, float* dW, int Bm, int N, int64_t K, int accumulate, int dtype, hipStream_t s);

/* l2norm + logits*exp(temperature) + symmetric InfoNCE, forward and backward (ct_clip.py:771,796,845-901). */
int ctclip_clip_loss(const float* text_latents, const float* image_latents, const float* temperature, float* out, float* logits, float* d_text, float* d_image, float* d_temperature, int G, int Dl, hipStream_t s);

/* The middle of the CLIP loss for ANY gathered batch size (ct_clip.py:845-901 with G = world size x batch beyond the 128 pairs ctclip_clip_loss keeps in LDS): S (G, G) f32 holds the cosines <u_t, u_v> of the l2-normalised latents on entry (f32 ctclip_gemm) and exp(temperature) * d loss / d logits on return (the factor both latent-gradient GEMMs need); out = [loss, exp(temperature)]; d_temperature (+=) = sum dS * logits.  workspace >= 16 * G bytes.  Fixed summation order. */
int ctclip_clip_loss_logits(float* S, int64_t lds, const float* temperature, float* out, float* d_temperature, int G, float* workspace, int64_t workspace_bytes, hipStream_t s);

/* Backward of F.normalize on rows (ct_clip.py:49-50,771): out = inv * (du - u <u, du>) with u = raw * inv; all f32, (rows, cols) contiguous. */
int ctclip_l2norm_bwd_rows(const float* raw, const float* inv, const float* du, float* out, int rows, int cols, hipStream_t s);

/* dst[i] += src[i] over n f32 values (n % 4 == 0, 16-byte aligned): the row blocks of a stacked weight gradient (one GEMM over [x | gate]) added into the flat gradient buffer. [replaces autograd gradient accumulation (AccumulateGrad) into param.grad under scripts/CTCLIPTrainer.py:259] */
int ctclip_accumulate_f32(float* dst, const float* src, int64_t n, hipStream_t s);

/* x *= scalar[0] (device scalar; scales the saved loss gradients by the upstream grad). [replaces the temperature multiply of ct_clip.py:796,805-807,845] */
int ctclip_scale_by_scalar(float* x, const float* scalar, int64_t n, hipStream_t s);

/* GEGLU: gelu(gate) * x (attention.py:39-42) on the padded [x | gate] layout. */
int ctclip_geglu_fwd(const void* u, void* g, int64_t M, int Hp, int dtype, hipStream_t s);

/* backward of GEGLU. [replaces autograd through GEGLU.forward, attention.py:39-43] */
int ctclip_geglu_bwd(const void* dg, const void* u, void* du, int64_t M, int Hp, int dtype, hipStream_t s);

/* erf-GELU (HF BertIntermediate). */
int ctclip_gelu_fwd(const void* u, void* h, int64_t n, int dtype, hipStream_t s);

/* backward of erf-GELU. [replaces autograd through HF BertIntermediate GELU (gelu = erf form)] */
int ctclip_gelu_bwd(const void* dh, const void* u, void* du, int64_t n, int dtype, hipStream_t s);

/* nn.LeakyReLU(0.1) of ContinuousPositionBias (attention.py:19-20,247-250). */
int ctclip_leaky_relu_fwd(const float* x, float* y, int64_t n, float slope, hipStream_t s);

/* backward of LeakyReLU. [replaces autograd through nn.LeakyReLU(0.1) of ContinuousPositionBias, attention.py:247-252] */
int ctclip_leaky_relu_bwd(const float* dy, const float* x, float* dx, int64_t n, float slope, hipStream_t s);

/* bytes of workspace ctclip_colsum needs (per-row-block partial sums of the deterministic two-stage reduction). [workspace query of ctclip_colsum (bias gradients of nn.Linear: HF BERT dense layers, ctvit.py:173)] */
int64_t ctclip_colsum_workspace(int64_t M, int N);

/* bias gradients: out[n] += sum_m x[m][n]; `workspace` >= ctclip_colsum_workspace bytes (row-block partials, fixed summation order). [replaces autograd of the bias of nn.Linear (HF BERT dense layers, ctvit.py:173, attention.py:247-252)] */
int ctclip_colsum(const void* x, float* out, int64_t M, int N, int64_t ld, int dtype, void* workspace, int64_t workspace_bytes, hipStream_t s);

/* rearrange '(b t)(h w) d <-> (b h w) t d' between the spatial and temporal phases (ctvit.py:297-305). */
int ctclip_permute0213(const void* x, void* y, int64_t A, int B, int C, int D, int dtype, hipStream_t s);

/* y (C,R) = x (R,C)^T: transposed weight shadows for the grad-input GEMMs. [builds the transposed weight operand of the grad-input GEMM of nn.Linear (attention.py:48,51,119,120,125)] */
int ctclip_transpose2d(const void* x, void* y, int R, int C, int64_t ldx, int64_t ldy, int dtype, hipStream_t s);

/* torch.mean(enc_image, dim=1) (ct_clip.py:724). */
int ctclip_pool_fwd(const void* x, void* y, int64_t B, int t, int64_t R, int dtype, int out_dtype, hipStream_t s);

/* backward of the depth mean-pool. [replaces autograd through torch.mean(enc_image, dim=1), ct_clip.py:724] */
int ctclip_pool_bwd(const void* dy, void* dx, int64_t B, int t, int64_t R, int dtype, int out_dtype, hipStream_t s);

/* f32 master weight -> padded compute-dtype shadow (optionally scaled per column). [replaces tensor.to(dtype) casts around the latent projections (ct_clip.py:762-771) and gradient bucket staging (accelerate DDP, scripts/CTCLIPTrainer.py:138-140)] */
int ctclip_convert_pad(const void* src, void* dst, const float* colscale, int64_t rows, int64_t cols, int64_t lds_, int64_t rows_dst, int64_t cols_dst, int64_t ldd, int src_dtype, int dst_dtype, hipStream_t s);

/* ContinuousPositionBias: gather the per-offset MLP table to (heads, hw, hw) (attention.py:261-276). */
int ctclip_cpb_expand(const float* tab, float* bias, int H, int gh, int gw, hipStream_t s);

/* backward of the gather (deterministic segmented sum). [replaces autograd through the rel_pos_indices gather of ContinuousPositionBias.forward, attention.py:261-276] */
int ctclip_cpb_reduce(const float* dbias, float* dtab, int H, int gh, int gw, hipStream_t s);

/* nn.Dropout of HF BertEmbeddings / BertSelfOutput / BertOutput fused with the residual add that follows it (modeling_bert.py BertSelfOutput.forward): y = x * mask / (1 - p) (+ residual); the mask is Philox4x32-10(seed, element / 4, stream_id), so the backward re-applies the same call to dy. */
int ctclip_dropout(const void* x, const void* residual, void* y, int64_t n, float p, uint64_t seed, uint32_t stream_id, int dtype, hipStream_t s);

/* test helper: writes the 0 / 1 keep mask (nseq, H, L, L) u8 that ctclip_attn_fwd / bwd apply to the attention probabilities for (dropout_p, dropout_seed). [test helper: the attention-probability dropout mask of HF BertSelfAttention as the kernels draw it] */
int ctclip_attn_dropout_mask(float* mask, int nseq, int H, int L, float p, uint64_t seed, hipStream_t s);

/* HF BertEmbeddings: word + position + token_type(0) lookup. */
int ctclip_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0, void* x, int64_t rows, int Tlen, int Hd, int dtype, hipStream_t s);

/* quantize = embed[ind] (vector_quantize_pytorch, ctvit.py:403). */
int ctclip_vq_gather(const float* embed, const int64_t* idx, void* out, int64_t M, int d, int dtype, hipStream_t s);

/* VQ EMA buffer update (decay 0.8) of cluster_size and embed. [replaces the EMA codebook update of vector-quantize-pytorch (ema_inplace + laplace smoothing + l2norm) behind ctvit.py:403] */
int ctclip_vq_ema_update(float* cluster, float* embed, const float* bins, const float* esum, int C, int d, float decay, hipStream_t s);

/* parameter-space epilogue of the patch-embedding backward (replaces the autograd of nn.LayerNorm(K) + nn.Linear(K, d), ctvit.py:172-173, with the LayerNorm affine folded into the GEMM): from G = dZ^T xhat (N x K) and dbp = colsum(dZ): dW (+)= G * gamma1 + dbp (x) beta1, dgamma1 (+)= sum_n W G, dbeta1 (+)= W^T dbp; all f32, fixed summation order. */
int ctclip_patch_embed_param_bwd(const float* G, const float* W, const float* gamma1, const float* beta1, const float* dbp, float* dW, float* dgamma1, float* dbeta1, int N, int K, int accumulate, hipStream_t s);

/* One thread spins for `microseconds` (0 .. 1 000 000) on stream s without touching memory: the busy kernel with which ct_clip_amd/streams.py probes whether a side stream runs beside the default stream (HIP multiplexes streams onto a few hardware queues). [no reference counterpart: the reference has one stream (scripts/CTCLIPTrainer.py:249-264)] */
int ctclip_spin(int64_t microseconds, hipStream_t s);

/* F.layer_norm (attention.py:28-35,47; ctvit.py:174; HF BertLayerNorm). gamma/beta may be NULL. */
int ctclip_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows, int cols, float eps, int dtype, hipStream_t stream);

/* bytes of workspace ctclip_layernorm_bwd needs when dgamma/dbeta are requested. [workspace query of ctclip_layernorm_bwd (autograd through F.layer_norm, attention.py:28-35,47)] */
int64_t ctclip_layernorm_bwd_workspace(int64_t rows, int cols);

/* LayerNorm backward, first half: dx (+ add1 + add2) and the per-workgroup partial sums of dgamma / dbeta into `partials` (ctclip_layernorm_bwd_workspace bytes; may be null = no parameter gradients). [replaces autograd through F.layer_norm / nn.LayerNorm (attention.py:28-35,47; ctvit.py:174), split so that the parameter-gradient fold can run on another stream] */
int ctclip_layernorm_bwd_partials(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx, const void* add1, const void* add2, int64_t rows, int cols, int dtype, void* partials, int64_t partials_bytes, hipStream_t stream);

/* LayerNorm backward, second half: dgamma / dbeta (either may be null) += the fixed-order sum of the partial rows ctclip_layernorm_bwd_partials wrote for the same (rows, cols); deterministic. [the weight / bias gradient of F.layer_norm, attention.py:28-35,47] */
int ctclip_layernorm_bwd_reduce(const void* partials, float* dgamma, float* dbeta, int64_t rows, int cols, hipStream_t stream);

/* backward of the above; dgamma/dbeta are ACCUMULATED (+=), may be NULL. [replaces autograd through F.layer_norm, attention.py:28-35,47,333, ctvit.py:172,174 and HF LayerNorm] */
int ctclip_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta, const void* add1, const void* add2, int64_t rows, int cols, int dtype, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* CTViT.to_patch_emb[0:2]: Rearrange 'b c (t pt)(h p1)(w p2) -> b t h w (c pt p1 p2)' + LayerNorm statistics (ctvit.py:171-172). */
int ctclip_patch_ln_fwd(const float* video, void* out, int64_t B, int F, int H, int W, int pt, int p1, int p2, int kpad, float eps, int out_dtype, hipStream_t stream);

/* F.normalize(x, dim=-1) (attention.py:22-23; ct_clip.py:49-50; VQ l2norm). */
int ctclip_l2norm_rows(const void* x, void* y, float* inv, int64_t rows, int cols, int64_t ldx, float eps, int in_dtype, int out_dtype, hipStream_t stream);

/* F.normalize rows as the three-term bf16 expansion the vector-quantiser code search multiplies on the matrix cores (vector_quantize_pytorch 1.1.2 CosineSimCodebook.forward computes the distances in f32): y (rows, 3 cols) bf16 = [hi | hi | lo] (order 0, tokens) or [hi | lo | hi] (order 1, codes) with hi = bf16(x^), lo = bf16(x^ - hi); order 2: y (rows, 2 cols) = [hi | lo], the codebook operand of ctclip_gemm_argmax_hilo; y null: inverse norms only; inv (rows) f32 inverse norms or null. */
int ctclip_l2norm_split3(const void* x, void* y, float* inv, int64_t rows, int cols, int64_t ldx, float eps, int in_dtype, int order, hipStream_t stream);

/* bytes of workspace ctclip_grad_norm_clip needs. [workspace query of ctclip_grad_norm_clip (accelerator.clip_grad_norm_, scripts/CTCLIPTrainer.py:259-260)] */
int64_t ctclip_grad_norm_workspace(void);

/* accelerator.clip_grad_norm_(params, 0.5) (CTCLIPTrainer.py:259-260): out = [norm, clip coefficient]. */
int ctclip_grad_norm_clip(const float* g, int64_t n, const float* extra_sq, float max_norm, float* out, void* workspace, int64_t workspace_bytes, hipStream_t s);

/* torch.optim.Adam / AdamW(lr, betas, eps, weight_decay).step() over a flat buffer (optimizer.py:24-32; CTCLIPTrainer.py:262).  weight_decay > 0 is the decoupled AdamW decay; decay_mask4 (one byte per 4 parameters, 1 = decayed) or null (all decayed) implements the reference's no-decay group for parameters with ndim < 2 (transformer_maskgit/optimizer.py:3-8).  clip: the 2-float result of ctclip_grad_norm_clip or null. */
int ctclip_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step, float weight_decay, const float* clip, const uint8_t* decay_mask4, hipStream_t s);

/* ctclip_adam_step followed by optimizer.zero_grad() in ONE pass over the flat buffers: every gradient is overwritten with zero right after it has been read (4 more bytes written per parameter instead of a separate fill launch over the 1.1-GB gradient buffer). [replaces optim.step(); optim.zero_grad() at scripts/CTCLIPTrainer.py:259-264] */
int ctclip_adam_step_zero_grad(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step, float weight_decay, const float* clip, const uint8_t* decay_mask4, hipStream_t s);

/* scripts/data.py:92-162 (CTReportDataset.nii_img_to_tensor) without the file decode: src = the (H, W, D) voxel array as nibabel returns it (src_dtype 0 int16, 1 f32, 2 f64, device memory) -> HU = slope * v + intercept, trilinear resample to target_xy / target_z mm (F.interpolate align_corners=False, new size int(n * spacing / target)), clip to [hu_lo, hu_hi], / hu_div, centre crop / pad with pad_value -> out (out_d, out_h, out_w) f32 = (240, 480, 480). */
int ctclip_preprocess_volume(const void* src, int src_dtype, int H, int W, int D, double slope, double intercept, double xy_spacing, double z_spacing, double target_xy, double target_z, float* out, int out_h, int out_w, int out_d, double hu_lo, double hu_hi, double hu_div, float pad_value, hipStream_t stream);

/* bytes of workspace ctclip_segment_sum needs (row histogram, scan and the row order of each segment). [workspace query of ctclip_segment_sum (EMA statistics of vector-quantize-pytorch behind ctvit.py:403; nn.Embedding backward of HF BertEmbeddings)] */
int64_t ctclip_segment_sum_workspace(int64_t M, int nseg);

/* out[key[r]] (+)= rowscale[r] * x[r, :d] summed in ascending row order per segment (deterministic scatter-add: the EMA statistics of the vector quantiser and the embedding-table gradients of HF BertEmbeddings).  keys (M) int64 or null (then key(r) = r % key_mod); x (M, ldx) f32 / bf16; rowscale (M) f32 or null; out (nseg, d) f32; counts_f (nseg) f32 row counts or null; accumulate 0 overwrites. */
int ctclip_segment_sum(const int64_t* keys, int key_mod, const void* x, int64_t ldx, const float* rowscale, float* out, float* counts_f, int64_t M, int d, int nseg, int accumulate, int in_dtype, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* Batched refresh of the bf16 weight shadows after the optimiser step (replaces the per-parameter `.to(bf16)` / `.t().contiguous()` / re-layout copies a torch module makes when its weights change; scripts/CTCLIPTrainer.py:259-263 is followed by nothing of the kind because torch computes from the f32 weights): jobs = DEVICE array of njobs records of 12 int64 {src, dst, src_ld, dst_ld, src_rows, src_cols, dst_rows, dst_cols, map, aux, transposed, tile0} sorted by tile0 (tile0 of job i = sum over the jobs before it of ceil(dst_rows / 64) * ceil(dst_cols / 64)), ntiles = the total.  src f32 (src_rows, src_cols), dst bf16 (dst_rows, dst_cols); plain: dst[r][c] = src[map(r)][c], transposed: dst[r][c] = src[map(c)][r], 0 outside the source.  map 0: identity; 1: GEGLU [x | pad | gate | pad] split (aux = inner, half = mapped extent / 2); 2: ctclip_geglu_weight_interleave's row order (aux = inner). */
int ctclip_shadow_refresh(const void* jobs, int njobs, int64_t ntiles, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
