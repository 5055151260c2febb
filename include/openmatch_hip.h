/*
 * openmatch_hip.h — C ABI of the MI355X (gfx950) dense-retrieval hot path.
 *
 * The reference (thunlp/OpenMatch v2) has no native boundary of its own: every
 * FLOP of its hot path runs inside un-vendored third-party libraries
 * (HF transformers, torch ATen, faiss, NCCL).  This header is the boundary a
 * maintainer would bind instead of those libraries; each entry point names the
 * reference call site it replaces (paths relative to the reference tree,
 * `src/openmatch/...`; `HF:` = installed transformers 5.15).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *     pointers are borrowed, never owned or freed by the library;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *     nothing synchronises the device unless the entry point says so;
 *   - no hidden allocation on the hot loop: scratch is a caller-provided
 *     workspace whose size comes from the matching *_workspace_bytes();
 *   - return value: 0 = OK, non-zero = error, message via om_last_error()
 *     (thread-local, valid until the next failing call on that thread);
 *   - matrices are row-major, leading dimensions in ELEMENTS.
 */
#ifndef OPENMATCH_HIP_H
#define OPENMATCH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OM_ABI_VERSION 5   /* 5 (round 5): om_grad_sqnorm, om_adamw_step, om_loss_scale_update, om_encoder_packed_supported; OM_F16 training */

/* element types */
#define OM_F32 0
#define OM_BF16 1
#define OM_F16 2 /* IEEE half: the search shadow index; the inference encoder's float16 mode (OmEncoderConfig.dtype) */

/* GEMM epilogue activation */
#define OM_ACT_NONE 0
#define OM_ACT_GELU_ERF 1  /* HF "gelu"      (HF:activations.py GELUActivation)   */
#define OM_ACT_RELU 2      /* HF "relu"      (T5 DenseReluDense)                  */
#define OM_ACT_GELU_TANH 3 /* HF "gelu_new"  (T5 v1.1 gated act)                  */
#define OM_ACT_MUL_RESID 0x100 /* flag: multiply by `resid` instead of adding it (gated FFN) */
#define OM_ACT_PRE_GRAD 0x200  /* flag (erf-GELU training epilogue, 16-bit output): `pre_act` receives gelu'(v) instead of the pre-activation
                                * v -- the backward's dgrad then multiplies by it (OM_ACT_MUL_RESID) instead of evaluating gelu' */

/* encoder architecture */
#define OM_ARCH_BERT 0 /* HF:models/bert/modeling_bert.py  BertModel       */
#define OM_ARCH_T5 1   /* HF:models/t5/modeling_t5.py      T5EncoderModel  */

/* pooling — modeling/dense_retrieval_model.py:145-150 */
#define OM_POOL_NONE 0
#define OM_POOL_FIRST 1
#define OM_POOL_MEAN 2 /* utils.py:233-235 mean_pooling */

/* search precision */
#define OM_SEARCH_F32 0           /* exact f32 MFMA scan                                   */
#define OM_SEARCH_F16_RESCORE 1   /* f16 MFMA candidate scan + exact f32 re-score (same ids)  */

const char* om_last_error(void);
int om_abi_version(void);

/* Number of HIP devices visible, or -1 with om_last_error() set. */
int om_device_count(void);

/* Measurement aid (bench.py's `roofline` object): while enabled, every launch of the dense
 * contraction kernel (class 0: bf16 encoder/scan GEMM, class 1: f32 GEMM, class 2: filtered
 * index scan) is bracketed by hipEvents on the launch stream.  om_kernel_timing_read()
 * synchronises those events, returns the summed kernel time / launch count / algorithmic
 * FLOPs (2*M*N*K) of the class since the last reset, and resets it. */
#define OM_TIMING_GEMM_BF16 0
#define OM_TIMING_GEMM_F32 1
#define OM_TIMING_SCAN 2
/* Debug hook: when `buf` is non-NULL every 256-row-tile GEMM workgroup writes 32 shader-clock
 * stamps (start, prologue, per-K-step, epilogue) to buf[32*block]; NULL switches it off. */
void om_debug_gemm_trace(unsigned long long* buf);
/* Debug hook for A/B measurements inside one process: 0 = default tile-generation selection; 6 = never the persistent
 * generation 7 (gemm_wide7.h); 70 = generation 7 with one tile per workgroup (no cross-tile prefetch). */
void om_debug_gemm_gen(int gen);
/* Run-time switches for A/B measurements and tests (initialised from the environment variable of the same
 * name on first use): OM_OPT_ENCODER_FUSED_LN 1 = LayerNorm / RMSNorm fused across the encoder GEMMs where the
 * shapes allow (default), 0 = one normalisation kernel per site; OM_OPT_ENCODER_DEBUG 1 = log the path taken. */
#define OM_OPT_ENCODER_FUSED_LN 0
#define OM_OPT_ENCODER_DEBUG 1
#define OM_OPT_ATTENTION_FAST 2   /* 1 (default): bf16 inference attention on the low-instruction-count kernel; 0: the generic kernel; bit 1 (tests): the
                                     tile-at-a-time kernels that serve more than 256 tokens (forward with dropout, backward) at every length; bit 2 (A/B):
                                     beyond 256 tokens the first online-softmax kernel instead of the chunked fast one (round 6) */
#define OM_OPT_SCAN_GEN7 3        /* 1 (default): f16 index scan of wide query batches on the persistent generation-7 kernel; 0: generation 6 */
#define OM_OPT_SCAN_GROWTH 4      /* fast schedule of the index scan: rows scanned per round grow by this many percent of the rows already
                                    * scanned (default 60; smaller = more rounds, tighter thresholds, fewer appends per tile) */
#define OM_OPT_WGRAD_DEBUG 5     /* 0 (default); A/B switches of the weight-gradient kernel: bit 3 the register-staged kernel for everything, value >> 4 =
                                   * workgroups aimed at / 64.  Bits 1 (plain stores) and 2 (one step of the token loop) break the result: they are timing
                                   * probes and act only in a probe build (-DOM_PROBE_KERNELS); the shipped library ignores them (round 6) */
#define OM_OPT_GEMM_GROUP_M 6     /* row tiles per group of the persistent GEMM's tile walk (default 8): the group's A panels stay in an
                                   * XCD's L2 while its column tiles are swept */
#define OM_OPT_SCAN_QGROUP 7      /* query tiles (256 queries each) an XCD keeps resident in its L2 during the index scan (default 8) */
#define OM_OPT_TRAIN_WGRAD_STREAM 8 /* bit 0 (default 1): the BERT backward's weight-gradient GEMMs run on a second stream beside the data-gradient
                                    * chain (neither fills the GPU alone at training batch sizes); bit 1 (round 5, A/B, off): the weight transposes the
                                    * backward needs are launched by the training FORWARD on that stream -- measured 3 % SLOWER (the memory-bound transposes
                                    * take more from the forward's contractions than the 170 us they save: profiles/r05_train_ab_v2_*.jsonl);
                                    * bit 2 (A/B): the LayerNorm backward adds its d_gamma / d_beta sums with atomics, as before round 5 (default:
                                    * per-block partial sums + one reduce per layer group, a fixed order); bit 3 (A/B): that kernel without the
                                    * prefetch of the next row; bits 1-3 clear and bit 0 clear: everything on the caller's stream */
#define OM_OPT_ATTENTION_DEBUG 9  /* 0 (default); timing experiments on the bf16 attention kernel at L in (64, 128]: bit 0 no K / V fetch,
                                   * bit 1 no arithmetic, bit 2 no stores (results are garbage) */
#define OM_OPT_ENCODER_PINGPONG 10 /* 1 (default): the fused bf16 encoder's kernels alternate their walk direction over the token rows so
                                   * each starts on the rows its producer wrote last (memory-side cache hits); 0: always first to last */
#define OM_OPT_ENCODER_TWO_PLANE 11 /* bit mask (env OM_ENCODER_TWO_PLANE, default 3): the fused BERT encoder keeps its pre-LayerNorm residual stream in TWO
                                   * 16-bit planes (value = hi + lo; the reference's autocast keeps it in f32) for +2 bytes per element at the two
                                   * residual sites of a layer -- bit 0 bfloat16 (round 3: 1 - cos against the fp32 chain 1e-5 instead of 4.8e-5),
                                   * bit 1 float16 (round 6: the headline format inside the reference's own float16 autocast); bit 2 (opt-in, float16): the
                                   * second plane in EIGHT bits (e5m2 of the remainder * 2^10, a kernel-private layout) -- + 2.6 % passages/s at the same cosine / dot
                                   * ratios, one more swapped tie on the config-1 fixture's MRR@10; 0: one plane */
#define OM_OPT_GEMM_VARIANT 12     /* 0 (default): automatic tile-generation choice; 1 | 2 | 6 pin a generation (A/B measurements) */
#define OM_OPT_SEARCH_DEBUG 13     /* bit 0: om_sim_topk logs every round (rows done, chunk, list lengths) to stderr; bit 2 (A/B): the small-batch scan
                                    * fetches the index with the default cache policy instead of non-temporal loads (env OM_SEARCH_DEBUG) */
#define OM_OPT_TRAIN_WGRAD_BATCH 14 /* layers per deferred weight-gradient launch of the bf16 BERT backward (default 4; 0: one launch per
                                     * weight gradient as in round 2): the backward keeps every layer's dY and one om_gemm_tn_acc_batch
                                     * launch per group of layers computes their weight gradients (env OM_TRAIN_WGRAD_BATCH) */
#define OM_OPT_GEMM_MAX_GRID 15    /* 0 (default): the persistent 16-bit GEMM takes every CU; > 0: at most this many workgroups (one per CU) --
                                    * two half-batch encoder forwards on two streams share the chip with 128 each */
#define OM_OPT_GEMM_CONT 16        /* bit mask (env OM_GEMM_CONT, default 495 = bits 0-3, 5, 6, 7, 8) of the persistent 16-bit GEMM's continuous ring (the K loop of a tile
                                    * prefetches the next tile's first two steps; epilogue and accumulator initialisation of the next tile interleaved).
                                    * bit 0: the variants without a residual; bit 1: the one-plane residual variants; bit 2: the f16 index scan of wide
                                    * query batches; bit 3: (no effect since round 5: the continuous GEMM kernels exist on 16 x 16 x 32 MFMAs only);
                                    * bit 4 (A/B, off): plain whole-tile bf16 shapes prefer the continuous kernels even when they leave CUs idle;
                                    * bit 5: the training forward's FFN1 (gelu + gelu' to the tape) on the continuous kernel with a two-output
                                    * epilogue; bit 6: generation 2 priced at its measured 0.55 of a 256 x 256 tile's rate when choosing between it and the
                                    * continuous kernel for plain whole-tile 16-bit shapes; bit 7 (round 5): plain float16 contractions follow the tile-choice model as
                                    * bfloat16 does (cleared: every whole-tile float16 shape on the persistent kernel); bits 8 / 9 (round 6): the float16 / bfloat16 TWO-plane residual variants on the continuous
                                    * ring (cleared -- bit 9 by default -- : the restart-per-tile kernel of round 3); a cleared bit 0 / 1: the ring restarts per tile as in round 3 */
#define OM_OPT_TRAIN_TAPE_GRAD 17  /* 1 (default): the bf16 BERT training forward keeps gelu'(f) on its tape instead of f (env OM_TRAIN_TAPE_GRAD) */
#define OM_OPT_TRAIN_RES32 18      /* 1 (default): the bf16 BERT training FORWARD keeps its residual stream in f32, as the reference's autocast does (layer_norm
                                    * runs and returns fp32): pre-LayerNorm sums in f32 on the tape, every LayerNorm output also unrounded for the next
                                    * residual add; 0: 16-bit residual stream as in rounds 1-4 (env OM_TRAIN_RES32; ~3 % faster, 2.4 x further from the
                                    * reference's fp32 gradients on tests/golden/train_base.npz) */
#define OM_OPT_GEMM_SKINNY_M 19    /* 16-bit contractions of at most this many rows run on the weight-streaming kernel (gemm_skinny.hip: one workgroup per
                                      16 output columns, K split over its waves) instead of the 128- / 256-column tiles, and the encoder forward takes its unfused path
                                      (normalisations as kernels) up to that many token rows; env OM_GEMM_SKINNY_M, default 1024, 0: off */
#define OM_OPT_GEMM_SKINNY_CFG 20  /* A/B: 0 (default) the kernel's own choice; MT * 10000 + NT * 100 + NW pins the tiles per wave and the K split (gemm_skinny.hip) */
#define OM_OPT_FEW_ROWS_LN_FUSE 21 /* round 6: 16-bit BERT forwards of at most this many token rows (default 64; env OM_FEW_ROWS_LN_FUSE; 0: off) launch no
                                      LayerNorm kernels between the embedding and the last layer: the contraction that consumes a LayerNorm's output
                                      normalises its operand rows itself, the one that adds it re-derives the element (gemm_skinny.hip; same bits) */
#define OM_OPT_COUNT 22
int om_debug_option(int opt, int value);
/* the attention kernel alone (bf16 qkv [B*L, 3H] -> ctx [B*L, H]; mask [B, L] int64), for timing: csrc/kernels.h omk_attention */
int om_debug_attention(const void* qkv, void* ctx, const int64_t* mask, int64_t B, int L, int H, int heads, void* stream);
/* self-check of the LayerNorm row reduction (csrc/ln_row.h): every group of 64 consecutive floats of `in` summed by the __shfl_xor butterfly
 * (out_shuffle[g]) and by its DPP / permlane form (out_dpp[g]); the two must agree bit for bit (tests/test_gpu_parity.py) */
int om_debug_wave_sum_check(const float* in, float* out_shuffle, float* out_dpp, int64_t groups, void* stream);
int om_kernel_timing_enable(int enable);
int om_kernel_timing_read(int kernel_class, double* total_ms, int64_t* launches, double* flops);

/* ------------------------------------------------------------------------
 * Dense contraction  C[M,N] = act(A[M,K] · B[N,K]^T + bias[N]) + resid[M,N]
 * (torch.nn.Linear layout: B is the [out,in] weight).  Replaces the ATen/BLAS
 * GEMMs under every nn.Linear of HF BertLayer / T5Block and linear.py:22-23.
 * in_dtype: OM_F32 (exact f32 MFMA, k-ordered fmaf chain), OM_BF16 or OM_F16 (16-bit MFMA, f32 accumulate;
 * OM_F16 with out_dtype OM_F16 or OM_F32, inference epilogues only).  bias (f32) and resid (out_dtype) may be NULL.
 * Requires K * sizeof(in) % 128 == 0.
 * ------------------------------------------------------------------------ */
int om_gemm_nt(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb,
               int out_dtype, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
               const float* bias, const void* resid, int64_t ldr, int act, void* stream);

/* ------------------------------------------------------------------------
 * Weight-gradient contraction  C[N,K] += A[M,N]^T · B[M,K],  bias[N] += column sums of A
 * (A = dY, B = X, both row-major bf16 as the backward pass holds them; C, bias f32,
 * ACCUMULATED into -- zero them first for a plain product).  Replaces autograd's
 * dW = dY^T X / db = sum(dY) of every nn.Linear under DRModel.forward + backward
 * (modeling/dense_retrieval_model.py:89-131).  Requires N % 128 == 0, K % 128 == 0,
 * lda / ldb multiples of 8, 16-byte aligned operands; bias may be NULL.
 * ------------------------------------------------------------------------ */
int om_gemm_tn_acc(int in_dtype, const void* A, int64_t lda, const void* B, int64_t ldb,
                   float* C, int64_t ldc, float* bias, int64_t M, int64_t N, int64_t K, void* stream);

/* The same contraction for MANY (A, B, C, bias) quadruples over one token count M in ONE launch (round 3): what the
 * training backward uses for the weight gradients of a whole group of layers -- autograd's dW = dY^T X of every nn.Linear,
 * deferred to the end of the group (modeling/dense_retrieval_model.py:89-131 + loss.backward()).  256 x 256 output tiles
 * over the whole token axis, no split and no atomics: every C / bias element is read, added to and written by exactly one
 * workgroup, so C and bias must not be touched by anything else while the launch runs.  Requires N % 256 == 0,
 * K % 256 == 0, M >= 32 (tokens past the last multiple of 32 go through the om_gemm_tn_acc kernels on the same stream),
 * lda / ldb multiples of 8, 16-byte aligned operands; bias may be NULL. */
typedef struct OmTnProblem {
  const void* A; const void* B; float* C; float* bias;   /* dY [M,N], X [M,K] (bf16), dW [N,K], db [N] (f32, accumulated) */
  int64_t lda, ldb, ldc, N, K;
} OmTnProblem;
int om_gemm_tn_acc_batch(int in_dtype, const OmTnProblem* problems, int n, int64_t M, void* stream);

/* ------------------------------------------------------------------------
 * Encoder forward:  ids -> hidden [B,L,H] -> pooled/head/normalised reps [B,D]
 * Replaces  lm(**items) + pooling + head + F.normalize  in
 * modeling/dense_retrieval_model.py:133-155 (DRModel.encode), i.e. the whole
 * HF BertModel.forward (HF:models/bert/modeling_bert.py:623-684) or
 * T5Stack.forward (HF:models/t5/modeling_t5.py) in eval mode.
 * ------------------------------------------------------------------------ */
typedef struct OmLayerWeights {
  /* matrices: compute dtype (OM_F32 / OM_BF16 / OM_F16), [out,in] row-major */
  const void* qkv_w;   /* [3H,H]  rows: query | key | value                       */
  const float* qkv_b;  /* [3H] or NULL (T5)                                       */
  const void* o_w;     /* [H,H]                                                   */
  const float* o_b;    /* [H] or NULL                                             */
  const float* ln1_g;  /* BERT: attention.output.LayerNorm ; T5: layer[0].layer_norm */
  const float* ln1_b;  /* NULL for T5 (RMSNorm)                                   */
  const void* ffn1_w;  /* [F,H]   BERT intermediate.dense / T5 wi (wi_0 if gated) */
  const float* ffn1_b; /* [F] or NULL                                             */
  const void* ffn1g_w; /* [F,H]   T5 v1.1 wi_1 (linear gate) or NULL              */
  const void* ffn2_w;  /* [H,F]                                                   */
  const float* ffn2_b; /* [H] or NULL                                             */
  const float* ln2_g;  /* BERT: output.LayerNorm ; T5: layer[1].layer_norm        */
  const float* ln2_b;
} OmLayerWeights;

typedef struct OmEncoderConfig {
  int arch;          /* OM_ARCH_*                                                 */
  int dtype;         /* compute dtype of matrices/activations: OM_F32 | OM_BF16 | OM_F16 (OM_F16: om_encoder_forward only,
                      * BERT-family erf-GELU encoders -- the reference's `--fp16` is torch.cuda.amp
                      * float16, retriever/dense_retriever.py:76; same kernels and MFMA rate as OM_BF16, 11-bit mantissas) */
  int hidden;        /* H                                                         */
  int n_layers;
  int n_heads;
  int head_dim;      /* 64                                                        */
  int ffn;           /* F                                                         */
  int vocab;
  int max_pos;       /* BERT position table rows                                  */
  int type_vocab;    /* BERT token-type table rows                                */
  int act;           /* OM_ACT_*                                                  */
  float ln_eps;      /* 1e-12 BERT, 1e-6 T5                                       */
  int rel_buckets;   /* T5 relative_attention_num_buckets (32)                    */
  int rel_max_dist;  /* T5 relative_attention_max_distance (128)                  */
  int pooling;       /* OM_POOL_*                                                 */
  int head_in;       /* LinearHead input dim  (0 = no head)                       */
  int head_out;      /* LinearHead output dim                                     */
  int normalize;     /* F.normalize(reps, dim=1)                                  */
} OmEncoderConfig;

typedef struct OmEncoderWeights {
  const float* word_emb;  /* [vocab,H] f32                                        */
  const float* pos_emb;   /* [max_pos,H] f32 (BERT)                               */
  const float* type_emb;  /* [type_vocab,H] f32 (BERT)                            */
  const float* emb_ln_g;  /* BERT embeddings.LayerNorm                            */
  const float* emb_ln_b;
  const OmLayerWeights* layers_host; /* HOST array [n_layers] of device pointers  */
  const float* final_ln_g; /* T5 final_layer_norm.weight                          */
  const float* rel_bias;   /* T5 block[0] relative_attention_bias [buckets,heads] f32 */
  const float* head_w;     /* LinearHead weight [head_out,head_in] f32, or NULL   */
  const void* folded;      /* LayerNorm-folded weights made by om_encoder_fold_weights (ABI v4), or NULL: folded per forward */
} OmEncoderWeights;

/* LayerNorm / RMSNorm folded into the weights that consume the normalised tensor (the 16-bit fused path): size of the
 * buffer (0 when the configuration has no fused path) and the one-off computation into a caller-owned, 256-byte aligned
 * device buffer.  Redo it whenever an encoder weight changes; om_encoder_forward reads it through OmEncoderWeights::folded. */
size_t om_encoder_fold_bytes(const OmEncoderConfig* cfg);
int om_encoder_fold_weights(const OmEncoderConfig* cfg, const OmEncoderWeights* w, void* folded, size_t bytes, void* stream);

size_t om_encoder_workspace_bytes(const OmEncoderConfig* cfg, int64_t B, int64_t L);

/* T5 relative-position bucket of `relative_position` = key_pos - query_pos, bidirectional
 * (HF:models/t5/modeling_t5.py T5Attention._relative_position_bucket).  Host function. */
int om_t5_relative_bucket(int relative_position, int num_buckets, int max_distance);

/* input_ids / attention_mask / token_type_ids: int64 [B,L] as the reference's
 * collators produce them (dataset/data_collator.py:27-38,78-83); token_type_ids
 * may be NULL (treated as 0; always ignored for T5).
 * out_hidden: [B,L,H] in cfg->dtype, or NULL.  out_reps: f32 [B,D]
 * (D = head_out if head else H), or NULL when pooling == OM_POOL_NONE. */
int om_encoder_forward(const OmEncoderConfig* cfg, const OmEncoderWeights* w,
                       const int64_t* input_ids, const int64_t* attention_mask,
                       const int64_t* token_type_ids, int64_t B, int64_t L,
                       void* out_hidden, float* out_reps, void* workspace,
                       size_t workspace_bytes, void* stream);

/* The same representations from PACKED rows: the reference pads every sequence of a batch to one length
 * (dataset/data_collator.py:27-38 `padding='max_length'`, or the longest of the batch) and HF runs every layer over the
 * padding; a padded key is masked out of every softmax, so the rows of a sequence up to its last unmasked token do not
 * depend on what follows them.  This entry keeps only those rows, back to back ([CLS] of sequence b at row cu[b]), runs
 * the embedding, all contractions and the normalisations over `packed_rows` rows instead of B * L, attention per
 * sequence over its own rows, and pools from them: the representations om_encoder_forward returns, for
 * sum(lengths) / (B * L) of the work.  16-bit configurations with the fused path (hidden, ffn multiples of 256;
 * BERT-family: erf-GELU, float16 or bfloat16; T5 encoders: no gated feed-forward), L <= 1024 (round 6; was 256), pooling set (no out_hidden).
 * packed_rows: the caller's bound on the token count -- sum over sequences of (1 + index of the last unmasked token) --
 * rounded up to a multiple of 256, >= 512.  The bound is checked on the device: a batch that holds more tokens returns
 * NaN in every representation (no host synchronisation, never a truncated batch).
 * Workspace: om_encoder_workspace_bytes_packed(cfg, B, L, packed_rows). */
/* 1 when om_encoder_forward_packed takes (cfg, B, L, packed_rows) under the current run-time switches (OM_OPT_*), else 0:
 * the host layer asks before choosing the packed entry and falls back to om_encoder_forward.  gated_ffn: T5 v1.1 layers (ffn1g_w). */
int om_encoder_packed_supported(const OmEncoderConfig* cfg, int gated_ffn, int64_t B, int64_t L, int64_t packed_rows);
size_t om_encoder_workspace_bytes_packed(const OmEncoderConfig* cfg, int64_t B, int64_t L, int64_t packed_rows);
int om_encoder_forward_packed(const OmEncoderConfig* cfg, const OmEncoderWeights* w,
                              const int64_t* input_ids, const int64_t* attention_mask,
                              const int64_t* token_type_ids, int64_t B, int64_t L, int64_t packed_rows,
                              float* out_reps, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * One decoder position of a T5 encoder-decoder over the encoder's output (inference):
 * what the reference computes for T5 backbones that are not --encoder_only --
 * DRModel.encode modeling/dense_retrieval_model.py:137-141 (decoder_input_ids = zeros([B,1]),
 * reps = decoder last_hidden_state[:, 0]) and the monoT5 scores of RRModel.encode
 * modeling/reranking_model.py:110-114 (two columns of the LM head over that state).
 * cfg: the ENCODER's config (hidden, heads, ffn, act, ln_eps, dtype); matrices in cfg->dtype,
 * [out,in] row-major, no biases (T5).  Self-attention over a single position needs only Wv, Wo.
 * enc_hidden: [B,L,H] in cfg->dtype = om_encoder_forward's out_hidden (after the final norm).
 * out_hidden: f32 [B,H], the decoder stack's output after its final RMSNorm. */
typedef struct OmT5DecoderLayer {
  const void* sa_v_w;    /* [H,H] layer[0].SelfAttention.v                                   */
  const void* sa_o_w;    /* [H,H] layer[0].SelfAttention.o                                   */
  const float* sa_ln_g;  /* layer[0].layer_norm                                              */
  const void* ca_q_w;    /* [H,H] layer[1].EncDecAttention.q                                 */
  const void* ca_kv_w;   /* [2H,H] rows: EncDecAttention.k | EncDecAttention.v               */
  const void* ca_o_w;    /* [H,H] layer[1].EncDecAttention.o                                 */
  const float* ca_ln_g;  /* layer[1].layer_norm                                              */
  const void* ffn1_w;    /* [F,H] wi (wi_0 if gated)                                         */
  const void* ffn1g_w;   /* [F,H] wi_1 or NULL                                               */
  const void* ffn2_w;    /* [H,F] wo                                                         */
  const float* ffn_ln_g; /* layer[2].layer_norm                                              */
} OmT5DecoderLayer;
typedef struct OmT5DecoderWeights {
  const float* start_emb;   /* [H] f32: shared.weight[decoder_start_token_id]                */
  const float* final_ln_g;  /* decoder.final_layer_norm.weight                               */
  const OmT5DecoderLayer* layers_host; /* HOST array [n_layers] of device pointers           */
  int n_layers;
} OmT5DecoderWeights;
size_t om_t5_decoder_workspace_bytes(const OmEncoderConfig* cfg, int64_t B, int64_t L);
int om_t5_decoder_step(const OmEncoderConfig* cfg, const OmT5DecoderWeights* w, const void* enc_hidden,
                       const int64_t* attention_mask, int64_t B, int64_t L, float* out_hidden,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Training through the decoder position (reference: autograd under DRModel.encode :137-141 / RRModel.encode :110-114 in
 * train mode).  `dropout` = HF config.dropout_rate (0 in eval), applied at HF's sites from hashes of (seed, site,
 * index); the forward keeps a caller-owned tape, the backward ADDS weight gradients into caller-zeroed f32 buffers laid
 * out like the weights and WRITES d_enc_hidden [B,L,H] (cfg->dtype), the gradient w.r.t. enc_hidden, which
 * om_encoder_train_backward_hidden takes.  start_emb [H]: gradient of the one embedding row the decoder reads (point it
 * at row decoder_start of the shared table's gradient; NULL skips it). */
typedef struct OmT5DecoderLayerGrads {
  float* sa_v_w;  float* sa_o_w; float* sa_ln_g;
  float* ca_q_w;  float* ca_kv_w; float* ca_o_w; float* ca_ln_g;
  float* ffn1_w;  float* ffn1g_w; float* ffn2_w; float* ffn_ln_g;
} OmT5DecoderLayerGrads;
typedef struct OmT5DecoderGrads {
  float* start_emb;
  float* final_ln_g;
  const OmT5DecoderLayerGrads* layers_host;   /* HOST array [n_layers] of device pointers */
} OmT5DecoderGrads;
size_t om_t5_decoder_tape_bytes(const OmEncoderConfig* cfg, int n_layers, int64_t B, int64_t L);
size_t om_t5_decoder_train_workspace_bytes(const OmEncoderConfig* cfg, int n_layers, int64_t B, int64_t L);
int om_t5_decoder_train_forward(const OmEncoderConfig* cfg, const OmT5DecoderWeights* w, const void* enc_hidden,
                                const int64_t* attention_mask, int64_t B, int64_t L, float dropout, uint64_t seed,
                                void* tape, size_t tape_bytes, float* out_hidden, void* workspace,
                                size_t workspace_bytes, void* stream);
int om_t5_decoder_train_backward(const OmEncoderConfig* cfg, const OmT5DecoderWeights* w, const void* enc_hidden,
                                 const int64_t* attention_mask, int64_t B, int64_t L, float dropout, uint64_t seed,
                                 const void* tape, const float* d_out, const OmT5DecoderGrads* grads,
                                 void* d_enc_hidden, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of y[B,D] = x[B,K] W[D,K]^T in f32 (om_gemm_nt forward): dw [D,K] = dy^T x (written), dx [B,K] = dy W.
 * Either output may be NULL.  The LinearHead / LM-head rows behind a trained T5 decoder position. */
int om_linear_f32_backward(const float* dy, const float* x, const float* w, float* dw, float* dx, int B, int D, int K,
                           void* stream);

/* ------------------------------------------------------------------------
 * Encoder forward + backward for training (BERT; sequence length <= 128).
 * Replaces the autograd graph HF builds under DRModel.forward in train mode
 * (modeling/dense_retrieval_model.py:89-131 -> HF BertModel with dropout) and
 * `loss.backward()` through it (trainer/dense_trainer.py:102-108).
 *
 * The forward saves one "tape" of activations (caller-provided memory); dropout masks are
 * regenerated from (seed, element index) in the backward, never stored.  Gradients are ADDED
 * (f32 atomics: split-K weight gradients, bias / LayerNorm / embedding reductions) into
 * caller-provided f32 buffers laid out like the weights; ZERO them before the call (or pass the
 * same buffers to several calls to accumulate).
 * ------------------------------------------------------------------------ */
typedef struct OmLayerGrads {
  float* qkv_w;  float* qkv_b;  float* o_w;    float* o_b;   float* ln1_g; float* ln1_b;
  float* ffn1_w; float* ffn1_b; float* ffn2_w; float* ffn2_b; float* ln2_g; float* ln2_b;
  float* ffn1g_w;                   /* T5 v1.1 gate projection wi_1 (ABI v2)          */
} OmLayerGrads;

typedef struct OmEncoderGrads {
  float* word_emb; float* pos_emb; float* type_emb; float* emb_ln_g; float* emb_ln_b;
  const OmLayerGrads* layers_host;  /* HOST array [n_layers] of device pointers */
  float* head_w;                    /* [head_out, head_in] or NULL              */
  float* final_ln_g;                /* T5: final RMSNorm weight [hidden] (ABI v2)     */
  float* rel_bias;                  /* T5: relative_attention_bias [buckets, heads]   */
} OmEncoderGrads;

size_t om_encoder_tape_bytes(const OmEncoderConfig* cfg, int64_t B, int64_t L);
size_t om_encoder_train_workspace_bytes(const OmEncoderConfig* cfg, int64_t B, int64_t L);

/* hidden_dropout / attn_dropout: HF config hidden_dropout_prob / attention_probs_dropout_prob
 * (0 disables).  out_reps: f32 [B,D] as om_encoder_forward.  Sequence length: up to 512 tokens in the 16-bit formats (round 6: above 256 the
 * attention runs on kernels that keep one score tile in registers; the reference trains at whatever length its collator pads to,
 * dataset/data_collator.py:13-24), up to 192 in float32. */
int om_encoder_train_forward(const OmEncoderConfig* cfg, const OmEncoderWeights* w,
                             const int64_t* input_ids, const int64_t* attention_mask,
                             const int64_t* token_type_ids, int64_t B, int64_t L,
                             float hidden_dropout, float attn_dropout, uint64_t seed, void* tape,
                             size_t tape_bytes, float* out_reps, void* workspace,
                             size_t workspace_bytes, void* stream);

/* d_reps: f32 [B,D] gradient of the loss w.r.t. out_reps.  Same ids / mask / dropout / seed /
 * tape as the matching forward call. */
int om_encoder_train_backward(const OmEncoderConfig* cfg, const OmEncoderWeights* w,
                              const int64_t* input_ids, const int64_t* attention_mask,
                              const int64_t* token_type_ids, int64_t B, int64_t L,
                              float hidden_dropout, float attn_dropout, uint64_t seed,
                              const void* tape, const float* d_reps, const OmEncoderGrads* grads,
                              void* workspace, size_t workspace_bytes, void* stream);

/* Packed rows in TRAINING (round 5; the training-side counterpart of om_encoder_forward_packed).  The reference's train collator pads
 * every query to q_max_len and every passage to p_max_len (dataset/data_collator.py:13-24) and the model computes over the padding;
 * here the contractions, normalisations, the tape and the weight gradients run over `packed_rows` rows -- each sequence's tokens up
 * to its last unmasked one, back to back.  ids / mask / token types keep their [B, L] layout; packed_rows is a multiple of 256 that
 * is >= the token count (computed on the host from the collator's lengths) and < B * L.  16-bit BERT-family and (round 6) T5 encoder
 * configurations with widths of 256, L <= 256, pooling first / mean: ask om_encoder_train_packed_supported.  Same representations and gradients as the
 * padded pair up to the order of the sums over rows; a bound below the token count turns out_reps into NaN. */
int om_encoder_train_packed_supported(const OmEncoderConfig* cfg, int64_t B, int64_t L, int64_t packed_rows);
size_t om_encoder_tape_bytes_packed(const OmEncoderConfig* cfg, int64_t B, int64_t L, int64_t packed_rows);
size_t om_encoder_train_workspace_bytes_packed(const OmEncoderConfig* cfg, int64_t B, int64_t L, int64_t packed_rows);
int om_encoder_train_forward_packed(const OmEncoderConfig* cfg, const OmEncoderWeights* w,
                                    const int64_t* input_ids, const int64_t* attention_mask,
                                    const int64_t* token_type_ids, int64_t B, int64_t L, int64_t packed_rows,
                                    float hidden_dropout, float attn_dropout, uint64_t seed,
                                    void* tape, size_t tape_bytes, float* out_reps,
                                    void* workspace, size_t workspace_bytes, void* stream);
int om_encoder_train_backward_packed(const OmEncoderConfig* cfg, const OmEncoderWeights* w,
                                     const int64_t* input_ids, const int64_t* attention_mask,
                                     const int64_t* token_type_ids, int64_t B, int64_t L, int64_t packed_rows,
                                     float hidden_dropout, float attn_dropout, uint64_t seed,
                                     const void* tape, const float* d_reps, const OmEncoderGrads* grads,
                                     void* workspace, size_t workspace_bytes, void* stream);

/* The same pair with the stack's output / its gradient at the boundary instead of pooled representations: out_hidden
 * and d_hidden are [B,L,H] in cfg->dtype (T5: after the final RMSNorm and its dropout).  cfg->pooling / head / normalize
 * are ignored.  What the T5 decoder position (om_t5_decoder_train_*) sits on. */
int om_encoder_train_forward_hidden(const OmEncoderConfig* cfg, const OmEncoderWeights* w,
                                    const int64_t* input_ids, const int64_t* attention_mask,
                                    const int64_t* token_type_ids, int64_t B, int64_t L,
                                    float hidden_dropout, float attn_dropout, uint64_t seed, void* tape,
                                    size_t tape_bytes, void* out_hidden, void* workspace,
                                    size_t workspace_bytes, void* stream);
int om_encoder_train_backward_hidden(const OmEncoderConfig* cfg, const OmEncoderWeights* w,
                                     const int64_t* input_ids, const int64_t* attention_mask,
                                     const int64_t* token_type_ids, int64_t B, int64_t L,
                                     float hidden_dropout, float attn_dropout, uint64_t seed,
                                     const void* tape, const void* d_hidden, const OmEncoderGrads* grads,
                                     void* workspace, size_t workspace_bytes, void* stream);

/* Gradient all-reduce overlapped with the backward (multi-GPU training): hand the NEXT om_encoder_train_backward on this
 * thread an array of n_layers + 1 hipEvent_t.  events[l] (l = n_layers-1 .. 0) is recorded on the backward's stream once
 * every kernel that writes layer l's gradients has been enqueued, events[n_layers] after the embedding gradients: a
 * side stream that waits for events[l] can all-reduce layer l's slice of the gradient arena while the rest of the
 * backward still runs.  NULL entries are skipped; the array is consumed by one backward. */
int om_encoder_train_set_layer_events(void* const* events, int n);

/* ------------------------------------------------------------------------
 * Optimizer step of the training loop: global-norm gradient clipping + AdamW + refresh of the packed compute-dtype
 * copies of the updated weights, for a whole model in three launches.  Replaces what the reference inherits from HF
 * Trainer.train per optimizer step (trainer/dense_trainer.py:27-108: torch.nn.utils.clip_grad_norm_(max_grad_norm),
 * torch.optim.AdamW.step) and the re-packing of the encoder's 16-bit weight matrices that would follow it here.
 *
 * tensors / chunks are DEVICE arrays the caller builds once per set of buffers: `tensors[t]` describes one parameter
 * (p, g, m, v: f32 [n]; g == NULL: the parameter received no gradient and is left alone; shadow0 / shadow1: up to two
 * copies of p in shadow*_dtype (OM_F32 | OM_BF16 | OM_F16) that are rewritten with the updated values -- the packed
 * weights OmLayerWeights points at -- or NULL); `chunks[2 c], chunks[2 c + 1]` = (tensor index, chunk index within it)
 * for every OM_ADAM_CHUNK elements of every tensor, one workgroup each.
 *   om_grad_sqnorm   out_sq[0] = sum over all tensors of |g|^2 (f32; fixed summation order); partial: n_chunks floats of scratch
 *   om_adamw_step    torch.optim.AdamW's update with bias corrections for `step` (counted from 1) on
 *                    g' = g * grad_scale * min(1, max_norm / (|g * grad_scale|_2 + 1e-6))   (max_norm <= 0: no clipping);
 *                    gnorm_sq = om_grad_sqnorm's out_sq (required when clipping).  skip_nonfinite != 0: an inf / nan norm
 *                    leaves every buffer untouched (the skipped step of a float16 GradScaler); om_loss_scale_update reads the same scalar.
 * ------------------------------------------------------------------------ */
#define OM_ADAM_CHUNK 16384
typedef struct OmAdamTensor {
  float* p; const float* g; float* m; float* v;
  void* shadow0; void* shadow1;
  int64_t n;
  float weight_decay;
  int shadow0_dtype, shadow1_dtype, reserved;
} OmAdamTensor;
int om_grad_sqnorm(const OmAdamTensor* tensors, const int32_t* chunks, int n_chunks, float* partial, float* out_sq, void* stream);
int om_adamw_step(const OmAdamTensor* tensors, const int32_t* chunks, int n_chunks, float lr, float beta1, float beta2, float eps,
                  int64_t step, const float* gnorm_sq, float max_norm, float grad_scale, int skip_nonfinite,
                  const float* scale_state /* NULL, or state4 of om_loss_scale_update: gradients are multiplied by state4[1] as well, and the
                                              bias corrections use step - state4[3] (GradScaler does not call step() on a skipped step) */,
                  void* stream);
/* Dynamic loss scale of float16 training (torch.cuda.amp.GradScaler.update; the reference's --fp16 through HF Trainer,
 * trainer/dense_trainer.py:141-149) on the device: state4 = {scale, 1 / scale, clean steps, skipped steps}; a non-finite
 * gnorm_sq[0] (om_grad_sqnorm of the SCALED gradients) halves the scale, `growth_interval` finite steps in a row double it. */
int om_loss_scale_update(const float* gnorm_sq, float* state4, int growth_interval, void* stream);

/* ------------------------------------------------------------------------
 * Exact inner-product search.  Replaces faiss.IndexFlatIP.add / .search
 * (retriever/dense_retriever.py:38-41,105,180) and faiss-GPU sharding (:43-58).
 * ------------------------------------------------------------------------ */

/* index.add(): make the 16-bit shadow copy of rows [0,N) and accumulate the rounding
 * statistics the certified candidate margin needs.  The shadow is IEEE f16, not bf16: same
 * MFMA rate, 8x smaller rounding error, which is what keeps the certified margin (and with it
 * the candidate lists) narrow on anisotropic embedding sets.
 * stats: device float[2] = {max_i ||p_i - f16(p_i)||_2 , max_i ||f16(p_i)||_2}, updated with
 * max (initialise to 0 before the first add; +inf if a value overflows f16 -> f32 scan). */
int om_index_to_f16(const float* rows_f32, int64_t N, int d, void* rows_f16, float* stats,
                    void* stream);

size_t om_sim_topk_workspace_bytes(int64_t n_queries, int d, int k);

/* D,I = index.search(x, k):  scores[Q,k] f32 sorted descending, ids[Q,k] int64 =
 * id_offset + row, padded with (-3.4028235e38, -1) when N < k (faiss semantics).
 * Ties are ordered by ascending row.  mode OM_SEARCH_F16_RESCORE needs
 * index_f16 + stats from om_index_to_f16; returned scores are always the
 * exact f32 inner products.  Synchronises `stream` internally (reads back
 * overflow flags between scan rounds).  k <= 2048. */
int om_sim_topk(int mode, const float* queries, int64_t n_queries, const float* index_f32,
                const void* index_f16, const float* stats, int64_t N, int d, int k,
                int64_t id_offset, float* out_scores, int64_t* out_ids, void* workspace,
                size_t workspace_bytes, void* stream);

/* Diagnostics of the calling thread's last om_sim_topk: out[0] = scan precision that produced
 * the result (0 f32, 1 f16+rescore), [1] = scan rounds, [2] = overflow fallbacks (chunks redone
 * densely), [3] = longest candidate list, [4] = 1 if the certified f16 margin was too wide and
 * the call fell back to the f32 scan. */
void om_sim_topk_info(int64_t out[8]);

/* Merge W partial results (utils.py:215-229 merge_retrieval_results_by_score /
 * faiss shard merge): parts are [W][Q,k_in] row-major, each row descending;
 * entries with id < 0 are padding.  Output [Q,k_out] descending; ties keep
 * (part, position) order (python's stable sorted(reverse=True)). */
int om_topk_merge(const float* part_scores, const int64_t* part_ids, int W, int64_t n_queries,
                  int k_in, int k_out, float* out_scores, int64_t* out_ids, void* stream);

/* ------------------------------------------------------------------------
 * In-batch-negatives contrastive loss, forward + backward in one call.
 * Replaces  scores = q @ p.T ; CrossEntropyLoss(mean)(scores, arange(Q)*n_psg)
 * (modeling/dense_retrieval_model.py:113-122, loss.py:9-15) and its autograd.
 *   q [Qg,d], p [Pg,d] f32 (already all-gathered when negatives_x_device);
 *   loss = loss_scale * mean_i CE(scores[i,:], i*n_psg);
 *   d_q [q_rows,d], d_p [p_rows,d]: gradient of loss w.r.t. the LOCAL slices
 *   q[q_row0 : q_row0+q_rows], p[p_row0 : p_row0+p_rows] (the reference's
 *   all_gather re-inserts only the local tensor: :247-258).
 * scores (f32 [Qg,Pg]) may be NULL; d_q / d_p may be NULL (forward only).
 * workspace: (2*Qg*Pg + Qg) floats.
 * ------------------------------------------------------------------------ */
int om_contrastive_fwd_bwd(const float* q, const float* p, int Qg, int Pg, int d, int n_psg,
                           float loss_scale, int q_row0, int q_rows, int p_row0, int p_rows,
                           float* loss, float* scores, float* d_q, float* d_p, float* workspace,
                           void* stream);

/* The same with the full call surface of `F.cross_entropy(logits, target, reduction=...)` that the reference's
 * loss callables expose (loss.py:9-15: `target=None, reduction='mean'` are parameters of SimpleContrastiveLoss.__call__;
 * :27-31 forwards them through DistributedContrastiveLoss):
 *   target   int64 [Qg] class index per row, or NULL for the in-batch positive i*n_psg; -100 rows are ignored
 *            (torch's ignore_index: loss 0, no gradient, not counted by the mean);
 *   reduction 0 mean | 1 sum | 2 none (loss is then [Qg]; row_grad [Qg], or NULL for ones, is the upstream
 *            gradient of each row's loss -- call once with d_q = d_p = NULL for the forward, again for the backward).
 * workspace: (2*Qg*Pg + Qg + 1) floats. */
int om_contrastive_fwd_bwd_ex(const float* q, const float* p, int Qg, int Pg, int d, const int64_t* target, int n_psg,
                              int reduction, const float* row_grad, float loss_scale, int q_row0, int q_rows,
                              int p_row0, int p_rows, float* loss, float* scores, float* d_q, float* d_p,
                              float* workspace, void* stream);

/* ------------------------------------------------------------------------
 * Multi-GPU collectives over RCCL / xGMI (one process per GPU).  What the reference does through torch.distributed
 * + NCCL and faiss-GPU, behind plain pointers:
 *   om_allgather_rows   DRModel.dist_gather_tensor (modeling/dense_retrieval_model.py:247-258; loss.py:33-38):
 *                       recv[w*rows:(w+1)*rows] = rank w's rows, rank-major
 *   om_allreduce_grads  the gradient averaging DistributedDataParallel does under HF Trainer
 *                       (trainer/dense_trainer.py:27-108): in place over one flat f32 buffer
 *   om_exchange_topk    the shard-merge traffic of the faiss-GPU index (retriever/dense_retriever.py:43-58),
 *                       re-cut by query range: block w of this shard's [world][q_block][k] candidates goes to rank w
 *                       (follow with om_topk_merge on the received [world][q_block][k])
 * A communicator is created from a 128-byte unique id: rank 0 calls om_comm_unique_id, the host layer broadcasts it
 * over whatever rendezvous it has (openmatch_amd: the torch.distributed store), every rank calls om_comm_init with
 * its HIP device current.  RCCL is bound at run time; all calls are asynchronous on `stream`.
 * ------------------------------------------------------------------------ */
#define OM_COMM_ID_BYTES 128
int om_comm_unique_id(void* id128);
int om_comm_init(const void* id128, int world, int rank, void** comm);
int om_comm_destroy(void* comm);
int om_comm_count(void* comm, int* count);   /* ranks of the communicator as RCCL reports them (ncclCommCount) */
int om_allgather_rows(void* comm, const void* send, void* recv, int64_t rows, int64_t row_bytes, void* stream);
int om_allreduce_grads(void* comm, float* buf, int64_t n, int average, void* stream);
int om_exchange_topk(void* comm, int world, const float* D, const int64_t* I, int64_t q_block, int k, float* recvD,
                     int64_t* recvI, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OPENMATCH_HIP_H */
