/*
 * vqattack_hip.h -- C ABI of the MI355X (gfx950) kernels behind VQAttack's PGD hot path.
 *
 * Boundary contract (last row): plain device pointers, element / row counts,
 * scalars and a hipStream_t; every entry point returns an int (0 = VQA_OK, otherwise a negative VQA_ERR_*
 * or a positive hipError_t), never allocates, never synchronises and never touches the host side of
 * its buffers, so a caller may capture any sequence of them into a hipGraph.  All tensors are fp32,
 * contiguous unless strides are passed explicitly.  "Reference" below is ericyinyzy/VQAttack; paths are
 * relative to its root, A-ch = ALBEF_VQAttack/cleverhans/cleverhans/torch, V-ch = the VLMO_VQAttack copy.
 *
 * The library is loaded by vqattack_amd/_hip.py (ctypes); INTEGRATION.md shows the stub a maintainer
 * of the reference would add to call it from the reference's own cleverhans modules.
 */
#ifndef VQATTACK_HIP_H
#define VQATTACK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vqa_stream_t; /* a hipStream_t (NULL = the legacy default stream) */

#define VQA_OK 0
#define VQA_ERR_NULL (-1)     /* a required pointer is NULL */
#define VQA_ERR_SHAPE (-2)    /* a count/stride argument is out of the supported range */
#define VQA_ERR_ALIGN (-3)    /* a pointer that must be 4-byte aligned is not */

/* mode bits shared by the image-update entry points */
#define VQA_CLIP 1u           /* clamp the result to [cmin, cmax] (the reference's clip_min/clip_max not None) */
#define VQA_CHECK_RANGE 2u    /* atomically OR 1 into *flag when an INPUT x element is outside [cmin, cmax] or NaN
                                 (reference: torch.all(ge(x, clip_min)) / le(x, clip_max) sanity flags,
                                 A-ch/attacks/projected_gradient_descent.py:95-105) */
/* further bits the kernels may OR into a *flag word */
#define VQA_FLAG_RANGE 1      /* set by VQA_CHECK_RANGE */
#define VQA_FLAG_BAD_LABEL 2  /* vqa_ce_rows: a label is neither ignore_index nor in [0, V) (torch device-asserts) */
#define VQA_FLAG_DEGENERATE 4 /* optimize_linear self-check (A-ch/utils.py:101-104,110-116) would fail: norm=1 with a sample
                                 whose max |grad| is 0 or NaN, norm=2 with a non-finite sum of squares */

int vqa_abi_version(void);
const char* vqa_error_string(int code);

#ifdef VQA_TUNING
/* NOT part of the shipped library: libvqattack_hip.so runs one launch shape per kernel, fixed at compile time.  A build
 * with -DVQA_TUNING (`python -m vqattack_amd.build --tuning` -> lib/libvqattack_hip_tuning.so, used by tools/ only)
 * compiles the alternatives the sweeps of DESIGN.md compare and exposes them as process-wide knobs:
 *   option 0: resident workgroups per CU the grid is capped at (1..64, default 8)
 *   option 1: non-temporal hints, bit0 = gradient/second-stream loads, bit1 = result stores, bit2 = first/third-stream
 *             loads, bit3 = result stores only when the result exceeds the 256 MB Infinity Cache (default 1|4|8 = 13)
 *   option 2: 16-byte tiles in flight per lane and stream in vqa_linf_step (2, 4 or 8; default 4)
 *   option 3: tile-to-workgroup mapping, 0 = round-robin tiles (default), 1 = one contiguous chunk per workgroup
 *   option 4: workgroup size of the register-resident cross-entropy kernel (256, 512 or 1024; default 512)
 *   option 5: logits loads of the cross-entropy kernel, 2 = non-temporal only above 512 MB of logits (default),
 *             3 = always non-temporal (A/B measurements)
 *   option 6: grid of the cosine-loss kernel, 0 = exactly the resident workgroups (occupancy x CUs, default),
 *             n = 1..8 workgroups per CU
 *   option 7: rows in flight per wavefront in the cosine-loss kernel (1 or 2; default 2)
 *   option 8: non-temporal hints of the cosine-loss kernel, bit0 = loads of `a`, bit1 = gradient stores, bit2 = loads
 *             of the targets `b` (default 4)
 *   option 9: dQ-from-dS^T attention kernel, 1 = dS^T tile staged through LDS with 16-byte loads (default), 0 = direct
 *             dword loads into the MFMA operand (round 3's form)
 *   option 10: block-glue kernels, bit0 = non-temporal loads of the activation streams, bit1 = non-temporal stores of GELU
 *             results larger than the Infinity Cache (default 3) */
int vqa_set_option(int option, int value);
#endif


/* ---------------------------------------------------------------- L-infinity image update (hot)
 * sign(g) follows torch.sign: sign(+-0) = 0, sign(NaN) = 0.  clamp propagates NaN like torch.clamp.
 * All four are pure elementwise streams; `out` may alias `x` (in-place) but no other aliasing.
 */

/* PGD start point: out = clamp(x + clamp(eta, -eps, eps), cmin, cmax); eta == NULL means eta = 0.
 * Replaces A-ch/attacks/projected_gradient_descent.py:110-120 (+ the range flags :95-105). 8-12 B/element. */
int vqa_linf_init(const float* x, const float* eta, float* out, size_t n, float eps, float cmin, float cmax,
                  unsigned mode, int* flag, vqa_stream_t stream);

/* FGM update only: out = clamp(x + eps_iter * sign(g), cmin, cmax).
 * Replaces A-ch/utils.py:86,127 + A-ch/attacks/fast_gradient_method.py:151-160. 12 B/element. */
int vqa_linf_fgm(const float* x, const float* g, float* out, size_t n, float eps_iter, float cmin, float cmax,
                 unsigned mode, int* flag, vqa_stream_t stream);

/* Fused PGD iteration tail (the north-star kernel): FGM update, then projection on the eps-ball around x0:
 *   a   = clamp(x + eps_iter * sign(g), cmin, cmax)
 *   e   = clamp(a - x0, -eps, eps)
 *   out = clamp(x0 + e, cmin, cmax)
 * Replaces rows 5-8 and 10-13 of SURVEY.md section 2.3 (A-ch/attacks/fast_gradient_method.py:151-160 +
 * A-ch/attacks/projected_gradient_descent.py:146-151). 16 B/element (reads x, g, x0; writes out). */
int vqa_linf_step(const float* x, const float* g, const float* x0, float* out, size_t n, float eps_iter,
                  float eps, float cmin, float cmax, unsigned mode, int* flag, vqa_stream_t stream);

/* Projection only: out = clamp(x0 + clamp(adv - x0, -eps, eps), cmin, cmax).
 * Replaces A-ch/attacks/projected_gradient_descent.py:146-151 on its own. 12 B/element. */
int vqa_linf_project(const float* adv, const float* x0, float* out, size_t n, float eps, float cmin,
                     float cmax, unsigned mode, vqa_stream_t stream);

/* clip_eta(norm=inf): out = clamp(eta, -eps, eps).  A-ch/utils.py:21. */
int vqa_clip_eta_linf(const float* eta, float* out, size_t n, float eps, vqa_stream_t stream);

/* optimize_linear(norm=inf): out = eps * sign(g).  A-ch/utils.py:86,127. */
int vqa_optimize_linear_linf(const float* g, float* out, size_t n, float eps, vqa_stream_t stream);

/* zero_out_clipped_grads: out = (x <= cmin && sign(grad) < 0) || (x >= cmax && sign(grad) > 0) ? 0 : grad.
 * A-ch/utils.py:131-149 (defined in the reference's utils, unused by its attack drivers). 12 B/element. */
int vqa_zero_out_clipped_grads(const float* grad, const float* x, float* out, size_t n, float cmin, float cmax,
                               vqa_stream_t stream);

/* ---------------------------------------------------------------- per-sample reductions (L2 / L1 norms)
 * Deterministic two-stage reductions: stage 1 writes per-block partials into `ws`, stage 2 combines them in a
 * fixed order, so results are bitwise reproducible run to run.  `ws` must hold vqa_reduce_ws_bytes() bytes.
 */
size_t vqa_reduce_ws_bytes(int batch, size_t n_per_sample);

/* out[b] = sum_i (t[b,i] - (sub ? sub[b,i] : 0))^2.   A-ch/utils.py:33-35 and :106. 4-8 B/element. */
int vqa_sumsq_per_sample(const float* t, const float* sub, float* out, int batch, size_t n_per_sample,
                         float* ws, vqa_stream_t stream);

/* amax[b] = max_i |g[b,i]|, ties[b] = #{i : |g[b,i]| == amax[b]}.  A-ch/utils.py:95-100.  */
int vqa_absmax_ties_per_sample(const float* g, float* amax, float* ties, int batch, size_t n_per_sample,
                               float* ws, vqa_stream_t stream);

/* FGM update, L2:  out = clamp(x + eps_iter * (g / sqrt(max(1e-12, sumsq_g[b]))), cmin, cmax).
 * A-ch/utils.py:106-107,127 + fast_gradient_method.py:152-160. */
int vqa_l2_fgm(const float* x, const float* g, const float* sumsq_g, float* out, int batch,
               size_t n_per_sample, float eps_iter, float cmin, float cmax, unsigned mode, int* flag,
               vqa_stream_t stream);

/* Projection, L2: eta = adv - x0; eta *= min(1, eps / sqrt(max(1e-12, sumsq_eta[b]))); out = clamp(x0 + eta).
 * A-ch/utils.py:31-39 + projected_gradient_descent.py:146-151. */
int vqa_l2_project(const float* adv, const float* x0, const float* sumsq_eta, float* out, int batch,
                   size_t n_per_sample, float eps, float cmin, float cmax, unsigned mode, vqa_stream_t stream);

/* FGM update, L1: out = clamp(x + eps_iter * sign(g) * [|g| == amax[b]] / ties[b], cmin, cmax).
 * A-ch/utils.py:88-101,127. */
int vqa_l1_fgm(const float* x, const float* g, const float* amax, const float* ties, float* out, int batch,
               size_t n_per_sample, float eps_iter, float cmin, float cmax, unsigned mode, int* flag,
               vqa_stream_t stream);

/* Per-sample scaling used by the stand-alone utils:
 *   kind 0 (clip_eta, norm=2):        out = t * min(1, eps / sqrt(max(1e-12, stat[b])))           A-ch/utils.py:31-39
 *   kind 1 (optimize_linear, norm=2): out = eps * (t / sqrt(max(1e-12, stat[b])))                 A-ch/utils.py:106-107,127
 *   kind 2 (optimize_linear, norm=1): out = eps * (sign(t) * [|t| == stat[b]] / stat2[b])         A-ch/utils.py:88-101,127
 * flag (nullable): kinds 1 and 2 OR VQA_FLAG_DEGENERATE into it when the reference's self-check assert of
 * optimize_linear (A-ch/utils.py:101-104, :110-116: the result must have unit norm) would fire for a sample; the same bit
 * is set by vqa_l2_fgm / vqa_l1_fgm when their flag pointer is given.  No host sync: the caller reads the word once.
 */
int vqa_scale_per_sample(const float* t, const float* stat, const float* stat2, float* out, int batch,
                         size_t n_per_sample, float eps, int kind, int* flag, vqa_stream_t stream);

/* ---------------------------------------------------------------- cross-modal loss reduction
 * Rows of D contiguous floats, addressed as row(o, i) = base + o*stride0 + i*stride1 (strides in ELEMENTS) for
 * o < rows0, i < rows1 -- this is how the reference's `out[k][:, :feat_len, :]` truncation views are consumed
 * without a copy.  For every row pair:  c = sum_d (a_d / max(|a|, cos_eps)) * (b_d / max(|b|, cos_eps))
 * (torch.nn.CosineSimilarity semantics of torch 2.x).  The kernel accumulates  -c  per block into
 * partial[] and, when ga != NULL, writes d(gscale * sum(-c)) / d a  into ga (same addressing as a, with its own
 * strides) -- loss value and the gradient w.r.t. the model output in ONE pass over a and b.
 * row_mask (nullable, uint8 row WEIGHTS): row (o, i) is weighted by w = row_mask[(o % mask_period) * rows1 + i];
 * w == 0 rows contribute nothing and get a zero gradient whatever they hold, NaN included (padded text tokens of a
 * batched adapter; they must be readable: the kernel for rows of 256, 512, ... floats reads them), w == 2 counts a
 * row twice (the VLMO loss takes the [CLS] row both on its own and as a token, V-ch/attacks/fast_gradient_method.py:111).
 * Replaces nn.CosineSimilarity + negate + two torch.sum calls and their autograd backward:
 * A-ch/attacks/fast_gradient_method.py:98,120-127; V-ch/attacks/fast_gradient_method.py:102-114.
 * D must be a multiple of 4 and <= 2048; a, b, ga 16-byte aligned with strides multiples of 4.
 * Algorithmic bytes: 8*D per row (loss only) or 12*D per row (loss + gradient).
 * `partial` must hold vqa_neg_cos_partials() floats: one slot per workgroup plus the arrival counter of the in-kernel
 * fold, which must be ZERO before the first launch that uses the buffer (the folding workgroup resets it; launches that
 * share a buffer must be ordered on one stream).
 * loss_out (nullable): the workgroup that arrives last folds the partials in index order and writes
 *   loss_out[0] = (accumulate ? loss_out[0] : 0) + gscale * sum(partial)      -- bitwise reproducible, no second launch.
 * loss_out == NULL: partials only; fold them with vqa_sum_partials(partial, vqa_neg_cos_partials(), ...).
 * The grid is the number of workgroups resident at once (occupancy x compute units, queried from the device).
 */
int vqa_neg_cos_partials(void);
int vqa_neg_cos_rows(const float* a, const float* b, float* ga, float* partial, const uint8_t* row_mask,
                     long mask_period, long rows0, long rows1, int D, long a_stride0, long a_stride1,
                     long b_stride0, long b_stride1, long g_stride0, long g_stride1, float gscale,
                     float cos_eps, float* loss_out, int accumulate, vqa_stream_t stream);

/* The same pass over n_layers <= vqa_neg_cos_max_layers() feature maps of identical shape and strides in ONE launch
 * (the 13 / 25 per-layer maps of an encoder, never packed into one tensor): a, b, ga are HOST arrays of n_layers device
 * base pointers, copied into the kernel arguments (capture-safe); ga == NULL -> loss only.  Replaces the reference's
 * torch.cat / torch.stack feature packing (ALBEF_attack/adv_attack.py:124-125, vlmo_module.py:1435-1444) together
 * with the loss ops listed above.  Algorithmic bytes: n_layers * rows * 12*D (8*D without the gradient). */
int vqa_neg_cos_max_layers(void);
int vqa_neg_cos_rows_multi(const float* const* a, const float* const* b, float* const* ga, int n_layers,
                           float* partial, const uint8_t* row_mask, long mask_period, long rows0, long rows1, int D,
                           long a_stride0, long a_stride1, long b_stride0, long b_stride1, long g_stride0,
                           long g_stride1, float gscale, float cos_eps, float* loss_out, int accumulate,
                           vqa_stream_t stream);

/* dst[0] = (accumulate ? dst[0] : 0) + scale * sum_{i<count} partial[i], summed in index order by one workgroup.
 * Turns the partials of one or more vqa_neg_cos_rows launches into the scalar loss on the device
 * (the reference's float(loss.cpu()) host sync per step, projected_gradient_descent.py:145, is deferred). */
int vqa_sum_partials(const float* partial, int count, float* dst, int accumulate, float scale,
                     vqa_stream_t stream);

/* Masked-LM cross entropy with ignore_index over K label sets, loss and gradient in one launch:
 *   loss = sum_g sum_k mean_{r in group g : labels[k][r] != ignore} ( logsumexp(logits[r,:]) - logits[r, labels[k][r]] )
 * rows_per_group == 0: ONE group = F.cross_entropy's mean over all rows (the reference's op on whatever batch it is
 * given); rows_per_group == L: one group per SAMPLE of a (B*L)-row batch = the batch-1 reference's loss summed over
 * the samples, whose per-sample gradients equal the batch-1 gradients whatever the other samples' labels are.
 * row_loss[r] (scratch, `rows` floats) receives row r's share; loss_out (nullable) receives
 *   loss_out[0] = (accumulate ? loss_out[0] : 0) + gscale * sum_r row_loss[r]      (summed in row order, in the launch);
 * grad (nullable, (rows, V) contiguous) receives gscale * d loss / d logits.  labels is int64 [K][rows];
 * scratch holds vqa_ce_scratch_floats(K, groups) floats (the fold's arrival counters + reciprocal valid counts per label
 * set and group; the launch initialises it).  K <= vqa_ce_max_label_sets().
 * A label that is neither ignore_index nor in [0, V) makes the loss NaN and ORs VQA_FLAG_BAD_LABEL into *flag
 * (flag nullable); torch raises a device assert for it, it is never silently ignored.
 * exp() is the hardware exponential (v_exp_f32, ~2 ulp); parity with torch is stated as 1e-4 relative in the tests.
 * Replaces F.cross_entropy(out[0].view(-1, 30522), y[0][...].view(-1), ignore_index=-100), its K-fold repetition for
 * 3-d labels and the autograd backward: A-ch/attacks/fast_gradient_method.py:131-142, V-ch/...:115-126.
 * A row whose K labels are all ignore_index -- every position but the [MASK]-ed answer pieces in the reference's
 * workload, ALBEF_attack/adv_attack.py:433-558 -- has zero loss and a zero gradient whatever its logits are: they are not
 * read and no exponential is evaluated; the row costs its zero gradient store -- or nothing at all:
 * row_state (nullable, `rows` bytes, requires grad): for a caller that hands the SAME gradient buffer to every launch of an
 * attack (zero-filled once, together with row_state, when it is allocated).  row_state[r] != 0 means "the buffer's row r
 * holds a live gradient"; a live row writes its gradient and sets the byte, a dead row stores zeros only if the byte is
 * set and clears it.  With the labels of an attack fixed, dead rows are never written after the allocation.
 * Algorithmic bytes: 8*V per live row (read the logits once from HBM, write the gradient once); per dead row 4*V without
 * row_state, 0 with it. */
int vqa_ce_max_label_sets(void);
long vqa_ce_scratch_floats(int K, long groups);
int vqa_ce_rows(const float* logits, long row_stride, const int64_t* labels, int K, long rows, int V,
                long ignore_index, long rows_per_group, float* scratch, float* grad, float* row_loss, float gscale,
                float* loss_out, int accumulate, int* flag, unsigned char* row_state, vqa_stream_t stream);

/* ---------------------------------------------------------------- text side
 * dst[b, k, :] = src[b, idx[k], :]  for src (B, L, D), idx int64[K] with 0 <= idx[k] < L.
 * Replaces `x[1].grad[:, text_emb_pick]`, A-ch/attacks/fast_gradient_method_vl.py:120 (V-ch: :130). */
int vqa_gather_rows(const float* src, const int64_t* idx, float* dst, int B, int L, int K, int D,
                    vqa_stream_t stream);

/* Candidate-substitution scoring (update_adv_text + dir_sim, ALBEF_attack/adv_attack.py:284-298,325-333;
 * vlmo_module.py:1632-1702).  For candidate c with (sample s, token position p, gradient row k, vocabulary id v):
 *   e   = LayerNorm(word[v] + pos[p] + type[0]; gamma, beta, ln_eps)         (BERT embeddings, position-wise)
 *   d   = e - e_ori[s, p]
 *   out[c] = cos( d / max(|d|, 1e-12),  g / max(|g|, 1e-12) ; eps = 1e-6 ),   g = grad[s, k]
 * cand is int32[n_cand][4] = {s, p, k, v}.  word (V, D), pos (P, D), type (>=1, D), e_ori (S, L, D), grad (S, K, D).
 * D multiple of 4, <= 2048. */
int vqa_cand_dir_sim(const float* word, const float* pos, const float* type, const float* gamma,
                     const float* beta, float ln_eps, const float* e_ori, const float* grad,
                     const int32_t* cand, float* out, int n_cand, int L, int K, int D, vqa_stream_t stream);

/* Masked-token embedding substitution: BERT text embeddings for explicit triples {destination row, token position,
 * vocabulary id}:  dst[row, :] = LayerNorm(word[id] + type[0] + pos[position]; gamma, beta, ln_eps), dst viewed as
 * (rows, D).  With all B*L rows it embeds a question batch in one launch; with a short list it rewrites only the rows
 * of the words the joint attack just substituted.  Replaces the re-tokenise + `text_embeddings(adv_text_ids)` calls
 * between PGD blocks (ALBEF_attack/adv_attack.py:643-645,369-384; models/xbert.py:189-216).
 * triples int32 [n][3]; D multiple of 4, <= 2048; the caller guarantees rows are distinct and in range. */
int vqa_embed_tokens(const float* word, const float* pos, const float* type, const float* gamma, const float* beta,
                     float ln_eps, const int32_t* triples, int n, float* dst, int D, vqa_stream_t stream);

/* Greedy acceptance of the word substitutions of one probe step, for the whole batch, on the device
 * (update_adv_text's sort + acceptance loop, ALBEF_attack/adv_attack.py:299-323; vlmo_module.py:1676-1700), with the
 * bag-of-embeddings stand-in for the reference's TF-Hub sentence encoder (:315-318):
 *   sim(ids) = <m(ori), m(ids)> / (|m(ori)| |m(ids)| + 1e-12),  m(ids) = mean over non-pad (id != 0) tokens of table[id].
 * cand int32 [n][4] = {sample, position, grad row, vocabulary id} (as for vqa_cand_dir_sim); order int32 [n] lists the
 * candidate indices grouped by sample and, within a sample, by descending dir_sim score (stable); seg int32 [B + 1]:
 * order[seg[s] .. seg[s+1]) belong to sample s.  Per sample, in that order: skip a candidate whose position already
 * took a substitution; accept it iff the similarity between the ORIGINAL question and the current question with that
 * position replaced exceeds the threshold, which then rises to that similarity.
 * ori_ids, cur_ids int64 (B, L), L <= 64; cur_ids is updated in place; new_id int32 (B, L) receives the accepted id per
 * position or -1, acc_rank (nullable, int32 (B, L)) the order in which the sample accepted it (0, 1, ...) or -1.
 * table fp32 (V, E), E <= 512.  One wavefront per sample. */
int vqa_greedy_accept(const int32_t* cand, const int32_t* order, const int32_t* seg, int B, int L,
                      const int64_t* ori_ids, int64_t* cur_ids, int32_t* new_id, int32_t* acc_rank, const float* table,
                      int V, int E, float threshold, vqa_stream_t stream);

/* ---------------------------------------------------------------- white-box attention (fp32, head dimension 64)
 * o = softmax(scale * q k^T + bias) v  per (batch, head), exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32),
 * flash-style (the Sq x Sk scores never reach HBM).  The frozen white boxes' `Attention.forward`
 * (vlmo/modules/multiway_transformer.py:88-118; ALBEF_attack/models/vit.py) is a callee of the attack's hot path.
 * q (B, H, Sq, 64), k / v (B, H, Sk, 64), o (B, H, Sq, 64) are addressed through element strides
 * strides[12] = {q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss, o_sh} (batch, sequence, head; the head
 * dimension is dense; all multiples of 4, bases 16-byte aligned) -- q, k, v may be views of one packed qkv tensor.
 * bias (nullable): additive fp32 (relative-position bias and/or -inf key padding), bias_strides[3] = {batch, head, query
 * row} in elements (batch / head stride may be 0, all multiples of 4), key stride 1.  The kernels read bias rows in
 * whole tiles of 32 keys: from every row start, ceil32(Sk) floats must be readable (pad the rows; what lies beyond Sk
 * is masked, never used).  lse (B, H, Sq) receives log-sum-exp of the scores.
 * scores (nullable; vqa_attn_scores_floats(B, H, Sq, Sk) floats, 16-byte aligned): when given, the forward also stores
 * the pre-softmax scores scale * q k^T + bias -- a forward that will be differentiated hands them to vqa_attn_bwd, whose
 * key-block kernel then skips the q k^T product and the bias.  The buffer is OPAQUE: per (batch, head) a grid of 32 x 32
 * tiles over ceil128(Sq) x ceil128(Sk), each tile in the register order of the wave that produced it (csrc/attn.hip);
 * only vqa_attn_bwd of the same B, H, Sq, Sk reads it.
 * key_hole (nullable, int32 (B, 2)): keys [key_hole[2b], key_hole[2b+1]) of batch element b score -inf for every head
 * and query -- the padded text tokens of a question shorter than the batch's text length (key padding of the reference's
 * attention mask).  It lets a ragged batch share ONE (1, H, S, S) relative-position slab (batch stride 0) instead of a
 * per-sample (B, H, S, S) bias.  The stored scores carry the -inf, so the scores-based backward needs nothing else; the
 * score-recomputing forms of vqa_attn_bwd do not know the hole: callers that use them pass the padding inside `bias`.
 * nsplit / split_ws (ABI v4): nsplit == 1 (split_ws ignored) is the form above: one workgroup per (batch, head, 128
 * queries) walks all key tiles.  nsplit > 1 is for SMALL batches -- the reference's own batch 1 gives 12 heads x 5
 * blocks = 60 workgroups for 256 CUs, each a serial chain over ~19 tiles: the key loop is cut into nsplit parts
 * (nsplit <= ceil(Sk / 32)), every part is its own workgroup writing partial accumulators + (running maximum, row sum)
 * into split_ws (vqa_attn_split_ws_floats(B, H, Sq, Sk, nsplit) floats, 16-byte aligned), and a second kernel combines
 * the parts in index order (flash-decoding reduction; deterministic, results equal the unsplit form to fp32 rounding). */
int vqa_attn_fwd(const float* q, const float* k, const float* v, const float* bias, float* o, float* lse, float* scores,
                 int B, int H, int Sq, int Sk, const long* strides, const long* bias_strides, float scale,
                 const int* key_hole, int nsplit, float* split_ws, vqa_stream_t stream);
long vqa_attn_scores_floats(int B, int H, int Sq, int Sk);
long vqa_attn_split_ws_floats(int B, int H, int Sq, int Sk, int nsplit);

/* Gradients of the above w.r.t. q, k, v given go = d loss / d o (the bias is frozen: no gradient).  Deterministic (no
 * float atomics: bitwise reproducible), two forms:
 *   ds_ws != NULL (vqa_attn_bwd_ws_floats(B, H, Sq, Sk) floats, 16-byte aligned, contents irrelevant): 5 products.  A
 *     streaming pre-pass writes delta (B, H, Sq) = rowsum(go . o); the kernel that owns key blocks computes dk, dv and
 *     stores the dS tiles it forms on the way, transposed, into ds_ws (tiled like `scores`); a third kernel forms dq
 *     from them.  With
 *     scores != NULL (what vqa_attn_fwd stored for the same operands) the key-block kernel reads the scores instead of
 *     recomputing them: 4 products.
 *   ds_ws == NULL: 7 products, no workspace: one kernel owns query blocks (dq; it also writes delta), one owns key
 *     blocks (dk, dv); both recompute the probabilities from lse.
 * grad_strides[12] = {go_sb, go_ss, go_sh, dq_sb, dq_ss, dq_sh, dk_sb, dk_ss, dk_sh, dv_sb, dv_ss, dv_sh} (elements;
 * dq / dk / dv may be the three slices of one packed (B, S, 3, H, 64) gradient buffer).
 * nsplit / split_ws (ABI v4; needs ds_ws and scores): as in vqa_attn_fwd -- the key-block kernel's loop over query
 * tiles and the dq kernel's loop over key tiles are cut into nsplit parts (nsplit <= min(ceil(Sq / 32), ceil(Sk / 32)))
 * whose partial dk / dv / dq are summed in part order by a small kernel (the same split_ws serves both, in turn). */
int vqa_attn_bwd(const float* q, const float* k, const float* v, const float* bias, const float* o, const float* go,
                 const float* lse, const float* scores, float* delta, float* dq, float* dk, float* dv, float* ds_ws, int B,
                 int H, int Sq, int Sk, const long* strides, const long* bias_strides, const long* grad_strides,
                 float scale, int nsplit, float* split_ws, vqa_stream_t stream);
long vqa_attn_bwd_ws_floats(int B, int H, int Sq, int Sk);

/* ---------------------------------------------------------------- input pipeline
 * Pillow-exact bicubic resize of an 8-bit interleaved image (H, W, C), C <= 4, then ToTensor + Normalize into planar
 * fp32 -- what `transforms.Resize((res, res), interpolation=Image.BICUBIC)`, `ToTensor()`, `Normalize(0.5, 0.5)` do on
 * the host in the reference (ALBEF_attack/dataset/__init__.py:17,35-39; vlmo/transforms/square_transform.py:11-18).
 * kk int32 [out][ksize] fixed-point taps (22 fractional bits) and bounds int32 [out][2] = {first source index, tap
 * count} are Pillow's precompute_coeffs + normalize_coeffs_8bpc tables (vqattack_amd/preprocess.py builds them).
 *   pass 1: dst[y, xo, c] uint8 (H, W_out, C)         = horizontal resampling of src
 *   pass 2: dst[c, yo, x] fp32  (C, H_out, W)          = (vertical resampling of src / 255 - mean) / std
 *           (kk == NULL: no vertical resampling, requires H_in == H_out -- conversion only)
 * dst of pass 2 is typically `batch + b*3*S*S`, the image's slot of the attack's (B, 3, S, S) tensor. */
int vqa_resize_bicubic_h_u8(const uint8_t* src, int h, int w_in, int c, const int32_t* kk, const int32_t* bounds,
                            int ksize, int w_out, uint8_t* dst, vqa_stream_t stream);
int vqa_resize_bicubic_v_normalize(const uint8_t* src, int h_in, int w, int c, const int32_t* kk,
                                   const int32_t* bounds, int ksize, int h_out, float mean, float stdv, float* dst,
                                   vqa_stream_t stream);

/* ---------------------------------------------------------------- white-box block glue (callee of the hot path)
 * The frozen encoders' pre-LN transformer block -- reference: VLMO_VQAttack/vlmo/modules/multiway_transformer.py:184-201
 * (x = x + gamma_1 * attn(norm1(x)); text tokens through norm2_text / mlp_text, image tokens through norm2_imag /
 * mlp_imag, split at max_text_len :193-197; x = x + gamma_2 * mlp) and ALBEF's ViT block (ALBEF_attack/models/vit.py) --
 * executed without an autograd graph by vqattack_amd/whitebox/_fused.py: library GEMMs + vqa_attn_* + these four entry
 * points.  Rows are D contiguous floats (D % 4 == 0, D <= 1024), every pointer 16-byte aligned.
 * Token layout for the modality split: a batch element has `period` rows (tokens), the first `split` are text tokens
 * (segment 0), the rest image tokens (segment 1); a "split" tensor is two contiguous buffers, (B*split, D) and
 * (B*(period-split), D) -- what the two expert GEMMs read and write.  period == 0: no split anywhere.
 *
 * vqa_ln_fwd: optional prologue  x_out = x + rscale * r  (r0 != NULL; r given whole as r0, or split as r0 / r1;
 *   rscale NULL = 1), then  y = LayerNorm(x_out; gamma, beta, eps)  with (gamma1, beta1) for segment-1 rows when given,
 *   y written whole (y1 == NULL) or split (y0 / y1); mean[row], rstd[row] saved for the backward.  Without a prologue
 *   x_out is not written.  16 B/element with prologue, 8 without.  Replaces torch.addcmul + torch.cat + two slice copies
 *   + F.layer_norm per stage (multiway_transformer.py:186-199).
 * vqa_ln_bwd: dx = (g_a ? g_a : 0) + (g_inj ? g_inj : 0) + dLayerNorm/dx(dy; x, mean, rstd, gamma)  (frozen gamma/beta:
 *   input gradient only); dy whole (dy1 == NULL) or split; g_a = gradient arriving over the residual path, g_inj = the
 *   loss kernel's gradient of this feature map (vqa_neg_cos_rows_multi's ga[layer]); and, when dr0 != NULL,
 *   dr = rscale * dx (rscale NULL = 1) written whole or split (dr1): the gradient of the branch the forward prologue
 *   added.  20-28 B/element.  Replaces layer_norm's backward, the gradient-accumulation adds, addcmul's backward and the
 *   slice / cat backward copies.
 * vqa_gelu_fwd / vqa_gelu_bwd: a = gelu(h) / dh = da * gelu'(h), exact erf form (nn.GELU() of the reference's Mlp,
 *   multiway_transformer.py:36-55); dh may alias da. */
int vqa_ln_fwd(const float* x, const float* r0, const float* r1, const float* rscale, float* x_out,
               const float* gamma0, const float* beta0, const float* gamma1, const float* beta1, float* y0, float* y1,
               float* mean, float* rstd, long rows, int D, long period, long split, float eps, vqa_stream_t stream);
int vqa_ln_bwd(const float* dy0, const float* dy1, const float* x, const float* mean, const float* rstd,
               const float* gamma0, const float* gamma1, const float* g_a, const float* g_inj, const float* rscale,
               float* dx, float* dr0, float* dr1, long rows, int D, long period, long split, vqa_stream_t stream);
int vqa_gelu_fwd(const float* h, float* a, size_t n, vqa_stream_t stream);
int vqa_gelu_bwd(const float* h, const float* da, float* dh, size_t n, vqa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VQATTACK_HIP_H */
