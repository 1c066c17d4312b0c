/*
 * pgmi.h -- C ABI of libpgmi.so, the MI355X (gfx950) masked-LM scorer behind ProteinGym's
 * ESM zero-shot path.
 *
 * The reference (OATML-Markslab/ProteinGym, /root/reference) has no FFI: a "baseline" plugs in
 * through (1) the per-baseline CLI + per-assay CSV and (2) the in-process seam
 *     model, alphabet = pretrained.load_model_and_alphabet(path)
 *                         (proteingym/baselines/esm/compute_fitness.py:349, esm/pretrained.py:24-28)
 *     model(tokens_int64[B,T])["logits"] -> f32 [B,T,33]
 *                         (compute_fitness.py:502; esm/model/esm1.py:116,177; esm/model/esm2.py:76,130)
 * Every entry point below names the reference code it replaces.  The Python host
 * (proteingym_amd/) binds these with ctypes; INTEGRATION.md shows the stub a ProteinGym
 * maintainer would add.
 *
 * Conventions: plain C types only; return 0 on success, negative PGMI_E* otherwise (never
 * throws); the caller owns every host buffer; the library owns device memory and one HIP
 * stream per model handle.  A handle is not thread-safe; distinct handles are independent.
 * pgmi_last_error() returns a thread-local message for the last failing call.
 */
#ifndef PGMI_H
#define PGMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGMI_ABI_VERSION 4

/* error codes */
#define PGMI_OK 0
#define PGMI_EINVAL (-1)   /* bad argument / unsupported shape */
#define PGMI_ENOMEM (-2)   /* device allocation failed */
#define PGMI_EHIP (-3)     /* HIP runtime error (message in pgmi_last_error) */
#define PGMI_ENODEV (-4)   /* no usable GPU */
#define PGMI_EPARSE (-5)   /* malformed mutant string / wild-type mismatch */
#define PGMI_EOVERFLOW (-6) /* 16-bit modes: an activation left the fp16/bf16 range (re-run in fp32) */

/* architectures: esm/model/esm1.py (arch "roberta_large": ESM-1b, ESM-1v) and esm/model/esm2.py */
#define PGMI_ARCH_ESM1B 1
#define PGMI_ARCH_ESM2 2
/* Tranception: GPT2-style causal LM with grouped ALiBi, depth-wise conv on q/k/v, squared ReLU
 * (proteingym/baselines/tranception/tranception/model_pytorch.py) */
#define PGMI_ARCH_TRANCEPTION 3
#define PGMI_ARCH_MSA 4          /* MSA Transformer (esm_msa1b): axial attention, esm/model/msa_transformer.py */

/* GEMM operand precision.  Residual stream, LayerNorm statistics, softmax and every
 * accumulator are fp32 in all modes. */
#define PGMI_PREC_FP32 0  /* v_mfma_f32_32x32x2_f32: exact fp32 products (parity-gated mode) */
#define PGMI_PREC_BF16 1  /* bf16 operands, fp32 accumulate (throughput mode; error is measured, not assumed) */
#define PGMI_PREC_F16X3 2 /* split-fp16 3-pass error-compensated GEMM (fp32-class accuracy on the f16 MFMA pipe) */

/* token ids of the 33-symbol ESM alphabet (esm/data.py:151-157, esm/constants.py:8) */
#define PGMI_TOK_CLS 0
#define PGMI_TOK_PAD 1
#define PGMI_TOK_EOS 2
#define PGMI_TOK_UNK 3
#define PGMI_TOK_MASK 32
#define PGMI_VOCAB 33

typedef struct pgmi_config {
    int32_t abi_version;          /* = PGMI_ABI_VERSION */
    int32_t arch;                 /* PGMI_ARCH_* */
    int32_t layers;               /* encoder_layers */
    int32_t embed_dim;            /* D  (encoder_embed_dim) */
    int32_t heads;                /* H  (encoder_attention_heads); head_dim D/H: 64, an even value below 64 (ESM2 8M/35M/150M: 16/24/32), or 128 (ESM2-15B) */
    int32_t ffn_dim;              /* F  (encoder_ffn_embed_dim; 4*D for ESM2, esm2.py:52) */
    int32_t vocab;                /* = 33 */
    int32_t max_positions;        /* ESM-1b learned positions (table has max_positions+2 rows, modules.py:246-251); 0 for ESM2 */
    int32_t token_dropout;        /* esm1.py:125-131 / esm2.py:85-91 */
    int32_t emb_layer_norm_before;/* pretrained.py:80-82,98 */
    int32_t precision;            /* PGMI_PREC_* */
    int32_t max_rows;             /* workspace rows (B*T per internal chunk); 0 = default */
    float ln_eps;                 /* LayerNorm epsilon; 0 = 1e-5 (ESM: modules.py:80-81; Tranception: config.layer_norm_epsilon) */
} pgmi_config;

typedef struct pgmi_model pgmi_model;
typedef struct pgmi_assay pgmi_assay;
typedef struct pgmi_pppl pgmi_pppl;

/* ---- library ---------------------------------------------------------------------------- */
int pgmi_abi_version(void);
int pgmi_device_count(void);
const char* pgmi_last_error(void);

/* Number of fp32 elements of the flat weight blob for cfg, in this order (all row-major,
 * nn.Linear layout y = x W^T + b, esm/modules.py; names as in SURVEY.md Appendix A):
 *   embed_tokens[V,D]; (ESM1B) embed_positions[max_positions+2,D];
 *   (emb_layer_norm_before) w[D],b[D];
 *   per layer: self_attn_layer_norm w,b; q_proj W[D,D],b; k_proj W,b; v_proj W,b; out_proj W,b;
 *              final_layer_norm w,b; fc1 W[F,D],b[F]; fc2 W[D,F],b[D];
 *   emb_layer_norm_after w,b; lm_head.dense W[D,D],b; lm_head.layer_norm w,b; lm_head.bias[V].
 * (lm_head.weight is tied to embed_tokens, esm1.py:101-105.)  Returns <0 on a bad cfg. */
int64_t pgmi_weight_count(const pgmi_config* cfg);

/* Replaces pretrained.load_model_and_alphabet + model.cuda() (compute_fitness.py:349-353):
 * uploads the blob to `device`, packs it for the selected precision, allocates workspace.
 * embed_tokens must be the value load_state_dict leaves in the tied embed_tokens/lm_head.weight
 * parameter, i.e. after the host applied pretrained.py:97 (<mask> row zeroing for v1 files). */
int pgmi_model_create(const pgmi_config* cfg, const float* weights, int64_t n_weights,
                      int device, pgmi_model** out);
void pgmi_model_destroy(pgmi_model* m);
int pgmi_model_device(const pgmi_model* m);

/* ---- forward ---------------------------------------------------------------------------- */
/* Replaces `torch.log_softmax(model(tokens)["logits"], -1)` (compute_fitness.py:476,502):
 * tokens int32 [B,T] row-major (host), out f32 [B,T,V] (host).  <pad> tokens are honoured as
 * the reference does (embedding rows zeroed, keys masked).  Chunked internally. */
int pgmi_token_logprobs(pgmi_model* m, const int32_t* tokens, int B, int T, float* out);

/* The masked-marginals inner loop (compute_fitness.py:489-503) for B ready-made rows:
 * row b is forwarded with tokens[b, mask_pos[b]] replaced by <mask>; out[b,:] =
 * log_softmax(logits[b, mask_pos[b], :]).  tokens/mask_pos host int32, out host f32 [B,V]. */
int pgmi_masked_logprobs(pgmi_model* m, const int32_t* tokens, const int32_t* mask_pos,
                         int B, int T, float* out);

/* ---- per-assay pipeline, inputs resident in HBM --------------------------------------------
 * pgmi_assay_create uploads everything one DMS assay needs:
 *   wt_tokens  int32 [n_tok]   cls + residues + eos   (BatchConverter, esm/data.py:262-297)
 *   positions  int32 [P]       token positions to mask (any subset of [0,n_tok); the reference
 *                              runs all n_tok, compute_fitness.py:489 -- rows no mutant reads
 *                              may be skipped without changing any output)
 *   window                     model window (1024): for n_tok > window each position gets
 *                              get_optimal_window(i, n_tok, window) (utils/scoring_utils.py:43-52,
 *                              compute_fitness.py:492-495)
 *   sub_pos/sub_wt/sub_mt int32 [n_sub], mut_off int64 [n_mut+1]
 *                              flattened substitutions of every mutant: token position (1+idx),
 *                              wild-type and mutant token ids (label_row, compute_fitness.py:240-250)
 * pgmi_assay_run then executes the whole hot path on the device: masked windows -> forward ->
 * head on masked rows -> log-softmax table [n_tok,V] (NaN rows where not computed) ->
 * per-mutant score = sum_subs (f32(lp[mt]-lp[wt])) accumulated in double.
 * scores_host / table_host may be NULL; scores_dev (device pointer, double[n_mut]) may be NULL. */
int pgmi_assay_create(pgmi_model* m, const int32_t* wt_tokens, int n_tok,
                      const int32_t* positions, int P, int window,
                      const int32_t* sub_pos, const int32_t* sub_wt, const int32_t* sub_mt,
                      const int64_t* mut_off, int64_t n_mut, pgmi_assay** out);
int pgmi_assay_run(pgmi_model* m, pgmi_assay* a, double* scores_host, float* table_host,
                   double* scores_dev);
void pgmi_assay_destroy(pgmi_assay* a);

/* ---- pseudo-perplexity over variable-length sequences, library resident in HBM (BASELINE config 5) ----
 * Replaces the `--scoring-strategy pseudo-ppl` driver (compute_fitness.py:515-529: one compute_pppl call per
 * row of the `mutated_sequence` column) and compute_pppl itself (:258-279).
 * pgmi_pppl_create uploads a whole library ONCE:
 *   tokens   uint8 [seq_off[n_seq]]  every sequence as BatchConverter writes it: <cls> + residues + <eos>
 *                                    (esm/data.py:286-295), concatenated; no <pad>
 *   seq_off  int64 [n_seq+1]         seq_off[0] = 0
 * pgmi_pppl_run scores the sequences [first, first+count) (a shard of the library: ranks take disjoint
 * ranges).  For a sequence of L residues the reference loops i in range(1, L-1), masks TOKEN i and reads
 * log p(sequence[i]) = log-prob of token i+1's identity at position i (its off-by-one; residues 0 and L-1
 * are never scored, residue L-2's identity is scored at residue L-3's position, L <= 2 gives 0.0): all of it
 * is reproduced.  The (sequence, i) rows are enumerated on the device from the resident tokens; sequences of
 * DIFFERENT lengths share a batch: the run is ordered by length, a batch's T is its longest member, shorter
 * members are <pad>-filled and masked per sequence (key mask, position count and token-dropout ratio over the
 * non-pad tokens, exactly what the reference computes for a sequence alone).  No windowing, like the
 * reference: ESM-1b/1v fail above max_positions tokens ("Sequence length ... above maximum sequence length").
 *   scores_host double [count]  sum(log_probs): left-to-right double sum of the f32 terms (python's sum)
 *   terms_host  float  [pgmi_pppl_rows(first,count)]  optional: the terms, sequence by sequence in library order
 *   scores_dev  device double [count], optional
 * pgmi_pppl_rows: number of masked forwards (rows) the range needs.  pgmi_pppl_stats: rows, batches, real and
 * padded token counts of the last run (packing efficiency = tokens / padded_tokens). */
int pgmi_pppl_create(pgmi_model* m, const uint8_t* tokens, const int64_t* seq_off, int64_t n_seq, pgmi_pppl** out);
int pgmi_pppl_run(pgmi_model* m, pgmi_pppl* lib, int64_t first, int64_t count, double* scores_host,
                  float* terms_host, double* scores_dev);
int64_t pgmi_pppl_rows(const pgmi_pppl* lib, int64_t first, int64_t count);
int pgmi_pppl_stats(const pgmi_pppl* lib, int64_t* rows, int64_t* batches, int64_t* tokens, int64_t* padded_tokens);
void pgmi_pppl_destroy(pgmi_pppl* lib);

/* ---- host-side mutant parsing (label_row's string handling, compute_fitness.py:240-250) -----
 * text: n_mut NUL-free mutant strings ("A25G:L30P") concatenated, str_off int64 [n_mut+1].
 * sequence: wild type (len seq_len); offset_idx as --offset-idx.  Two-pass: call with
 * sub_* == NULL to get the substitution count in *n_sub, then with buffers of that size.
 * Fails with PGMI_EPARSE on a wild-type mismatch ("The listed wildtype does not match the
 * provided sequence") or a malformed token.  Letters map through the ESM alphabet
 * (unknown -> <unk>, esm/data.py:125-128). */
int pgmi_parse_mutants(const char* text, const int64_t* str_off, int64_t n_mut,
                       const char* sequence, int seq_len, int offset_idx,
                       int32_t* sub_pos, int32_t* sub_wt, int32_t* sub_mt,
                       int64_t* mut_off, int64_t* n_sub);

/* ---- host-side scoring from a log-prob table (label_row's arithmetic, compute_fitness.py:240-250) -----
 * table: float32 [n_rows][vocab] log-probabilities (row = token position, <cls> = 0), e.g. a table merged from position shards
 * or returned by pgmi_assay_run; sub_* / mut_off as pgmi_parse_mutants returns them.  scores[i] = sum over the substitutions of
 * mutant i, in the order of its string, of the float32 difference table[pos][mt] - table[pos][wt], accumulated in double -- the
 * reference's `.item()` sum and the device's score_mutants_kernel, bit for bit.  Host code, no GPU needed; rows that were never
 * computed (NaN) give NaN scores.  PGMI_EINVAL on a position or token outside the table. */
int pgmi_score_mutants(const float* table, int n_rows, int vocab, const int32_t* sub_pos, const int32_t* sub_wt,
                       const int32_t* sub_mt, const int64_t* mut_off, int64_t n_mut, double* scores);

/* get_optimal_window (proteingym/utils/scoring_utils.py:43-52) */
void pgmi_optimal_window(int position, int seq_len_with_special, int model_window,
                         int* start, int* end);

/* ---- profiling (HIP events on the model's stream) ----------------------------------------- */
#define PGMI_K_EMBED 0
#define PGMI_K_LAYERNORM 1
#define PGMI_K_GEMM_QKV 2
#define PGMI_K_ATTENTION 3
#define PGMI_K_GEMM_OUT 4
#define PGMI_K_GEMM_FC1 5
#define PGMI_K_GEMM_FC2 6
#define PGMI_K_HEAD 7
#define PGMI_K_SCORE 8
#define PGMI_K_KEPT_ROWS 9   /* last layer after attention, on the kept (masked) rows only: gathers + out-projection + LN + FFN */
#define PGMI_K_COUNT 10
/* on != 0: every launch of the classes above is bracketed by hipEventRecord on the stream. */
int pgmi_profile_enable(pgmi_model* m, int on);
/* Sum of event-measured milliseconds, launch count and algorithmic FLOPs / bytes for a class
 * since the last reset. */
int pgmi_profile_get(pgmi_model* m, int kernel_class, double* ms, int64_t* launches,
                     double* flops, double* bytes);
int pgmi_profile_reset(pgmi_model* m);
int pgmi_synchronize(pgmi_model* m);

/* Test hooks of the GEMM launchers (bit-neutral: they only change how a launch is cut into work items / row chunks):
 * "gemm_half_tail" (0: no half-height tail items; default 1), "gemm_max_rows" (> 0: cut every launch into row chunks of at most
 * that many rows), "att_xcd_local" (1: the dense attention launches walk their blocks in the XCD-local order; 0: the (query block,
 * head, sequence) grid; -1, default: by shape -- same bits, for interleaved timing), "att_v3" (dense head_dim-64 attention: -1, default:
 * the software-pipelined kernel from seven query tiles per sequence on; 0: the round-3 kernel everywhere; 1: the pipelined kernel
 * wherever it is defined -- same bits).  The library reads PGMI_GEMM_HALF_TAIL / PGMI_GEMM_MAX_ROWS when a model is created (and at the
 * model-less pgmi_op_* / pgmi_bench_* entries) for the options nobody has set through this call: an explicit value stays in force until
 * -1 hands "gemm_half_tail" / "gemm_max_rows" back to the environment.  Process-wide; returns PGMI_EINVAL for another name. */
int pgmi_set_option(const char* name, int64_t value);

/* ---- single ops (numerics tests compare each against the torch op it replaces) -------------
 * All pointers are host; each call uploads, runs the production kernel, downloads. */
int pgmi_op_layernorm(int device, const float* x, const float* w, const float* b,
                      int rows, int D, float eps, float* y);               /* modules.py:80-81 */
int pgmi_op_gemm(int device, int precision, const float* A, const float* W, const float* bias,
                 const float* residual, int M, int N, int K, int epilogue /*0 none,1 gelu,2 squared relu; +256 (f16x3):
                 the split-fp16-plane output epilogue, its planes returned rebuilt as fp32*/,
                 float* C);          /* C = epi(A W^T + bias) + residual; modules.py:134-140 */
int pgmi_op_attention(int device, int precision, const float* qkv, const int32_t* kv_len,
                      int B, int T, int H, int rotary, float* ctx);
                                     /* multihead_attention.py:354-395; qkv [B*T,3*H*64], q pre-scaled */

/* ---- Tranception (arch PGMI_ARCH_TRANCEPTION; vocab 25, max_positions = n_ctx, precision f16x3) ------
 * Weight blob order (fp32, names as in the HF state dict, Conv1D weights as stored = [in,out]):
 *   transformer.wte.weight [V,D];
 *   per layer h.{i}: ln_1 w,b; attn.c_attn W[D,3D], b[3D];
 *        attn.{query,key,value}_depthwiseconv.{0,1,2}.conv weight [64,k] (k = 3,5,7), bias [64]  (in that order);
 *        attn.c_proj W[D,D], b; ln_2 w,b; mlp.c_fc W[D,F], b[F]; mlp.c_proj W[F,D], b[D];
 *   transformer.ln_f w,b; lm_head.weight [V,D].
 *
 * pgmi_tr_token_logprobs: replaces log_softmax(model(input_ids, attention_mask).logits)
 *   (model_pytorch.py:731-783); tokens int32 [B,T] right-padded with [PAD]=3; out f32 [B,T,V].
 * pgmi_tr_sequence_loglik: the scoring reduction of tranception/utils/scoring_utils.py:97-128 --
 *   out[b] = sum_{t < lens[b]-1} log p(tokens[b,t+1] | tokens[b,<=t]) -- with the inference-time
 *   retrieval fusion of model_pytorch.py:806-830 when log_prior != NULL: for sequence b, logit rows
 *   [prior_a0[b], prior_a0[b]+prior_n[b]) are replaced by (1-alpha)*logp + alpha*log_prior[row],
 *   row = prior_row0[b] + i (or prior_row0[b] + n-1-i when prior_flip[b] != 0: right-to-left scoring).
 *   lens[b] counts [CLS] and [SEP].  log_prior is f32 [P,V] (host). */
int pgmi_tr_token_logprobs(pgmi_model* m, const int32_t* tokens, int B, int T, float* out);
int pgmi_tr_sequence_loglik(pgmi_model* m, const int32_t* tokens, const int32_t* lens, int B, int T,
                            const float* log_prior, int P, const int32_t* prior_a0, const int32_t* prior_row0,
                            const int32_t* prior_n, const int32_t* prior_flip, float alpha, float* out);
/* pgmi_tr_sequence_loglik_shared: the same quantity for B sequences of EXACTLY T tokens each (no padding), computed with the work
 *   the reference's loop repeats shared: the reference forwards every mutated sequence in full, once per reading direction
 *   (tranception/utils/scoring_utils.py:97-128 inside :77-150; model_pytorch.py:878-928), although the model is causal (attention
 *   model_pytorch.py:155-183, depth-wise convolution :73-88) and a mutated sequence equals the wild type up to its first mutated token.
 *   ref[b] names the sequence of this call whose prefix sequence b shares (its "root": the wild type cut to the same window); a root
 *   has ref[b] == b and is forwarded in full.  For every other sequence only the rows from its first difference from its root on go
 *   through LayerNorm, the GEMMs and the head (the depth-wise convolution and the attention also recompute the head of that token's
 *   32-token tile from the root's rows, so that tiles stay whole); keys, values, convolution history and log-probability rows before
 *   that are the root's.  Every row
 *   goes through the same kernels with the same inputs in the same order as in pgmi_tr_sequence_loglik: out[b] has the same bits.
 *   token_logprobs (optional, f32 [B,T,V]): the rows pgmi_tr_token_logprobs would return.  rows_forwarded (optional): token rows that
 *   went through the network (B*T for the unshared call). */
int pgmi_tr_sequence_loglik_shared(pgmi_model* m, const int32_t* tokens, const int32_t* ref, int B, int T,
                                   const float* log_prior, int P, const int32_t* prior_a0, const int32_t* prior_row0,
                                   const int32_t* prior_n, const int32_t* prior_flip, float alpha, float* out,
                                   float* token_logprobs, int64_t* rows_forwarded);

/* Tuning utility: times `iters` launches of the production GEMM (device-resident random operands,
 * HIP events) for one shape; variant selects the launch parameters (negative or below 1000 = library default;
 * 1000 + t: see gemm_f16.hip set_tune).  split_out: 0 fp32 output, 1 split fp16 planes (the next GEMM's operand),
 * 2 fp32 output with the in-place residual of the out-projection / FC2, 3 the fused QKV epilogue (N = 3 D).
 * Writes the mean milliseconds per launch. */
int pgmi_bench_gemm(int device, int precision, int M, int N, int K, int epilogue, int split_out,
                    int variant, int iters, double* ms_per_launch);
/* The same for several variants on ONE set of operands, timed in `rounds` interleaved rounds of `iters` launches
 * each (within-process A/B); ms_out[v] = median over the rounds of the mean milliseconds per launch. */
int pgmi_bench_gemm_ab(int device, int precision, int M, int N, int K, int epilogue, int split_out,
                       const int* variants, int n_variants, int rounds, int iters, double* ms_out);

/* ---- MSA Transformer (arch PGMI_ARCH_MSA; vocab 33, head_dim 64, precision f16x3) --------------------
 * Replaces MSATransformer.forward (proteingym/baselines/esm/esm/model/msa_transformer.py:146-205; tied
 * row attention esm/axial_attention.py:33-168, column attention :171-297) and the masked-marginals loop
 * of compute_fitness.py:380-394.
 * Config: max_positions = args.max_positions (1024), emb_layer_norm_before = 1, token_dropout = 0,
 * max_rows >= roundup(R,32) * roundup(T,32) for the largest token grid.
 * Weight blob (fp32, nn.Linear [out,in]), state-dict names after the loader's row/column swap
 * (pretrained.py:110-116):
 *   embed_tokens.weight [33,D] (the tied lm_head.weight), embed_positions.weight [max_positions+2, D],
 *   msa_position_embedding [1024, D] (a [1024,1] parameter is broadcast by the host),
 *   emb_layer_norm_before.{weight,bias};
 *   per layer: row_self_attention.{layer_norm.{weight,bias}, layer.q_proj.{weight,bias}, k_proj, v_proj,
 *              out_proj}, column_self_attention.{same}, feed_forward_layer.{layer_norm.{weight,bias},
 *              layer.fc1.{weight,bias}, layer.fc2.{weight,bias}};
 *   emb_layer_norm_after.{weight,bias}; lm_head.dense.{weight,bias}; lm_head.layer_norm.{weight,bias};
 *   lm_head.bias [33].
 *
 * pgmi_msa_token_logprobs: log_softmax(model(tokens[None])["logits"])[0] for one alignment.
 *   tokens int32 [R][T] (<cls> first in every row, no <pad>), out float32 [R][T][33].
 * pgmi_msa_masked_logprobs: for i < n, mask column positions[i] of the FIRST row, run the columns
 *   [starts[i], min(T, starts[i] + window)) of all rows, keep log-probabilities of that cell:
 *   out float32 [n][33].  (window = T when T <= 1024; else 1024 with starts from pgmi_optimal_window,
 *   exactly as compute_fitness.py:383-388 crops.)  The alignment stays resident in HBM across the n forwards. */
int pgmi_msa_token_logprobs(pgmi_model* m, const int32_t* tokens, int R, int T, float* out);
int pgmi_msa_masked_logprobs(pgmi_model* m, const int32_t* tokens, int R, int T, int window,
                             const int32_t* positions, const int32_t* starts, int n, float* out);

/* ---- alignment pre-processing ---------------------------------------------
 * Cluster sizes (inverse sequence weights) of an alignment: replaces the numba kernel
 * calc_num_cluster_members_nogaps_parallel (proteingym/utils/weights.py:164-216, called by
 * calc_weights_fast :13-53) and MSA_processing.compute_weight
 * (baselines/tranception/tranception/utils/msa_utils.py:341-352).
 *   matrix  int8 [N][L]  symbols mapped to 0..29; invalid_value marks gaps / lower-case columns
 *   counts_out int32 [N] #{j : matches(i,j) / nongap(i) > identity_threshold}, self included;
 *                        0 for a sequence with no valid symbol (the reference gives it weight 0)
 *   kernel_ms  optional: duration of the pair-count kernel (HIP events on its stream)
 * The double-precision predicate of the reference is evaluated exactly (per non-gap length on the
 * host, integer compares on the device): results are bit-identical to the reference's counts. */
int pgmi_msa_cluster_counts(int device, const int8_t* matrix, int64_t N, int64_t L, int invalid_value,
                            double identity_threshold, int32_t* counts_out, double* kernel_ms);

#ifdef __cplusplus
}
#endif
#endif /* PGMI_H */
