// Shared declarations for libpgmi (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/pgmi.h"

namespace pgmi {

void set_error(const char* fmt, ...);

#define PGMI_HIP(expr)                                                                   \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            pgmi::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                            __FILE__, __LINE__);                                         \
            return PGMI_EHIP;                                                            \
        }                                                                                \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;
constexpr int kHeadDim = 64;

enum Epilogue { EPI_NONE = 0, EPI_GELU = 1, EPI_SQRELU = 2 };   // erf-GELU (ESM), squared ReLU (Tranception)

// f16x3 activation split: x ~= hi + lo * 2^-11 with hi = fp16(x) and lo = fp16((x - hi) * 2^11).
// The scaled lo keeps its 11 bits for every |x| >= 2^-14 (an unscaled lo would be an fp16
// subnormal whenever |x| < 0.125); |x| below fp16's normal range goes entirely into lo, so no
// subnormal ever reaches the MFMA inputs.  The GEMM multiplies lo by (w_hi * 2^-11), which is
// exact because weights are pre-scaled to ~2^13.
constexpr float kLoScale = 2048.0f;
// The split q planes consumed by attention_f16x3_v2_kernel are pre-multiplied by log2(e): its online softmax works in base 2
// (v_exp_f32 is 2^x) and the scores come out of the MFMAs already in base-2 units -- 16 multiplies per 32 x 32 tile saved.
constexpr float kQLog2e = 1.4426950408889634f;

// K-interleaved layout of an f16x3 GEMM operand [rows][K] (K % 32 == 0): every group of 32 consecutive k is stored as the
// 32 hi halfs (64 B) followed by the 32 lo halfs (64 B), so a row's share of a 32-deep K tile is ONE 128-byte line.
// Returns the half index of the hi value of (row, k); its lo value sits 32 halfs further.  Row pitch: 2 K halfs.
__host__ __device__ __forceinline__ size_t ki_off(size_t row, int k, int K) {
    return row * (size_t)(2 * K) + (size_t)(k >> 5) * 64 + (size_t)(k & 31);
}
#if defined(__HIPCC__)
__device__ __forceinline__ void split_act(float x, _Float16& hi, _Float16& lo) {
    hi = (fabsf(x) < 6.103515625e-05f) ? (_Float16)0.0f : (_Float16)x;
    lo = (_Float16)((x - (float)hi) * kLoScale);
}
#endif

// ---- elementwise.hip ---------------------------------------------------------------------
// tokens_out[b,t] = (t == mask_pos[b]) ? <mask> : wt[start[b] + t]
void launch_make_masked_windows(const int32_t* wt, const int32_t* win_start, const int32_t* mask_rel,
                                int B, int T, int32_t* tokens_out, hipStream_t s);
// tokens[b, mask_pos[b]] = <mask> (in place)
void launch_apply_mask(int32_t* tokens, const int32_t* mask_pos, int B, int T, hipStream_t s);
// per sequence: scale[b] (token-dropout rescale), pos_idx[b,t] (learned-position index), kv_len[b]
void launch_seq_stats(const int32_t* tokens, int B, int T, int token_dropout, float* scale,
                      int32_t* pos_idx, int32_t* kv_len, hipStream_t s);
void launch_zero_pad_rows(const int32_t* tokens, int rows, int D, float* x, hipStream_t s);
void launch_embed(const int32_t* tokens, const float* scale, const int32_t* pos_idx,
                  const float* embed_tokens, const float* embed_positions, int token_dropout,
                  int rows, int T, int D, float* x, hipStream_t s);
void launch_layernorm(const float* x, const float* w, const float* b, int rows, int D, float eps,
                      float* y, hipStream_t s);
// y[i,:] = x[row_idx[i],:]
void launch_gather_rows(const float* x, const int32_t* row_idx, int n, int D, float* y, hipStream_t s);
// idx[i] = first + i * stride, i < n
void launch_strided_index(int first, int stride, int n, int32_t* idx, hipStream_t s);
// out[r,:] = log_softmax(h[r,:] @ E^T + bias); E [V,D]
void launch_vocab_logsoftmax(const float* h, const float* E, const float* bias, int rows, int D,
                             int V, float* out, int32_t* nonfinite, hipStream_t s);
void launch_layernorm16(const float* x, const float* w, const float* b, int rows, int D, float eps,
                        unsigned short* y16, size_t plane, int mode, hipStream_t s);
void launch_scatter_rows(const float* src, const int32_t* dst_row, int n, int V, float* table,
                         hipStream_t s);
void launch_row_index(const int32_t* mask_rel, int B, int T, int32_t* out, hipStream_t s);
void launch_fill_f32(float* p, int64_t n, float v, hipStream_t s);
// Tranception: per-sequence sum over t < len-1 of log p(tok[t+1] | tok[<=t]) from lp [B*T,V]
// (scoring_utils.py:118-128), with the retrieval fusion (model_pytorch.py:806-830) on rows
// [a0, a0+n) when prior != nullptr: value = (1-alpha)*lp + alpha*prior[row0 +/- i].
void launch_seq_loglik(const float* lp, const int32_t* tokens, const int32_t* lens, int B, int T, int V,
                       const float* prior, const int32_t* a0, const int32_t* row0, const int32_t* n,
                       const int32_t* flip, float alpha, float* out, hipStream_t s);
void launch_seq_loglik_ragged(const float* lp, const int32_t* tokens, const int32_t* seq_off, const int32_t* seq_p,
                              const int32_t* seq_root, int B, int T, int V, const float* prior, const int32_t* a0,
                              const int32_t* row0, const int32_t* n, const int32_t* flip, float alpha, float* out, hipStream_t s);
void launch_score_mutants(const float* table, int V, const int32_t* sub_pos, const int32_t* sub_wt,
                          const int32_t* sub_mt, const int64_t* mut_off, int64_t n_mut,
                          double* scores, hipStream_t s);
// H = 64-lane slot groups per token; rot_halves = table rows per token (2 for head_dim 128: the slot group's parity picks the row)
void launch_rotary(float* qkv, const float* cos_t, const float* sin_t, int rows, int T, int H,
                   hipStream_t s, int rot_halves = 1);
// pseudo-perplexity (compute_fitness.py:258-279): rows enumerated on the device from a resident sequence library
void launch_make_pppl_rows(const uint8_t* tok8, const int64_t* seq_off, const int32_t* sid, const int64_t* rp, int J,
                           int64_t g0, int bc, int T, int32_t* tokens, int32_t* row_idx, int32_t* target, hipStream_t s);
void launch_pppl_pick(const float* lp, const int32_t* target, int bc, int V, float* terms, hipStream_t s);
void launch_pppl_sum(const float* terms, const int64_t* rp, const int32_t* sid, int J, int64_t first, double* out, hipStream_t s);

// ---- gemm_f32.hip ------------------------------------------------------------------------
// C[M,N] = epi(A[M,K] W[N,K]^T + bias[N]) (+ residual[M,N]); K % 32 == 0.
// ---- msa_transformer.hip ------------------------------------------------------------------
void launch_msa_window_tokens(const int32_t* full, int R, int Tfull, int start, int Tw, int mask_col, int32_t* out, hipStream_t s);
void launch_add_row_embedding(float* x, const float* pe, int R, int C, int D, hipStream_t s);
void launch_permute_rows(const float* src, float* dst, int A, int B, int D, hipStream_t s);
// tied row attention operands for the 16-bit pipe (msa_transformer.hip)
void launch_tied_prep_qk(const float* qkv, int64_t M, int D, unsigned short* q16, unsigned short* k16, hipStream_t s);
void launch_pack_vt16(const float* qkv, int R, int C, int Kp, int H, unsigned short* Vt, hipStream_t s);
int launch_tied_softmax16(const float* part, int H, int S, int C, int Kp, float scale, unsigned short* P, hipStream_t s);
float tied_w_scale();

int launch_gemm_f32(const float* A, const float* W, const float* bias, const float* residual,
                    float* C, int M, int N, int K, int epilogue, hipStream_t s);

// ---- gemm_f16.hip ------------------------------------------------------------------------
void gemm_options_from_env();                      // PGMI_GEMM_HALF_TAIL / PGMI_GEMM_MAX_ROWS (test hooks): at model creation, never per launch
int gemm_set_option(const char* name, long long value);
// 16-bit-plane GEMM: C = epi(A W^T * out_scale + bias) (+ residual).  A/W: `planes` planes of
// fp16 (hi, lo) or one bf16 plane, K-contiguous rows.  Exactly one of Cf (fp32) / Ch (planes).
int launch_gemm16(const unsigned short* A, size_t a_plane, const unsigned short* W, size_t w_plane,
                  const float* bias, const float* residual, float* Cf, unsigned short* Ch, size_t c_plane,
                  int M, int N, int K, int epilogue, float out_scale, int planes, bool bf, int variant,
                  hipStream_t s);
// H = number of 64-lane slot groups per token (= heads, or 2 x heads for head_dim 128); rot_halves = rotary table rows per
// token (1, or 2 for head_dim 128: slot group parity selects the frequencies)
int launch_gemm16_qkv(const unsigned short* A, size_t a_plane, const unsigned short* W, size_t w_plane,
                      const float* bias, int M, int D, int K, float out_scale, unsigned short* qk16, size_t qk_plane,
                      unsigned short* vt16, size_t vt_plane, const float* cos_t, const float* sin_t, int rotary,
                      int T, int H, int variant, hipStream_t s, int rot_halves = 1, bool bf = false);
struct XMap;                                       // gemm16x_kernel.h: batched / strided operand and output maps
int launch_gemm16_ex(const unsigned short* A, const unsigned short* W, float* Cf, unsigned short* Ch, int M, int N, int K,
                     float out_scale, XMap xm, int nbatch, hipStream_t s);
// x [n/K rows][K] fp32 -> mode 0: f16x3 weight (hi/lo of x*scale), 1: bf16 plane, 2: f16x3 activation split; the
// f16x3 forms are written K-interleaved (ki_off)
void launch_split16(const float* x, int64_t n, float scale, int mode, int K, unsigned short* out, hipStream_t s);

// ---- attention_f32.hip -------------------------------------------------------------------
// qkv [B*T, 3*H*64] (q pre-scaled by 1/8); kv_len[b] (nullable) = valid keys.  Output: ctx fp32
// [B*T, H*64] (out_mode 0), or fp16 hi/lo planes (1) / one bf16 plane (2) in ctx16.
int launch_attention_f32(const float* qkv, const int32_t* kv_len, int B, int T, int H, float* ctx,
                         unsigned short* ctx16, size_t plane, int out_mode, hipStream_t s, int head_dim = 64);   // 128: heads of two adjacent slot groups

// ---- attention_f16.hip -------------------------------------------------------------------
// Split-fp16 (f16x3) attention on the 16-bit MFMA pipe: operands as attention-ready fp16 planes (from the fused QKV projection, or
// from a prep pass over qkv != nullptr: rotary / Tranception depth-wise conv fused), tiles moved with direct-to-LDS loads.  Scratch: qk16 (2 planes, stride qk_plane >= B*T*2*H*64 halfs) and vt16
// (2 planes, stride vt_plane >= B*H*64*roundup(T,32) halfs).
int launch_attention_f16x3_v2(const float* qkv, const int32_t* kv_len, const float* cos_t, const float* sin_t,
                              int rotary, int B, int T, int H, unsigned short* qk16, size_t qk_plane,
                              unsigned short* vt16, size_t vt_plane, float* ctx, unsigned short* ctx16, size_t plane,
                              int out_mode, hipStream_t s, const float* conv = nullptr, const float* slopes = nullptr,
                              int head_dim = 64);     // 128: H heads of two 64-lane slot groups (ESM2-15B), fused-QKV operands only

// Tranception prefix-shared scoring: device arrays that describe a launch over SUFFIXES of sequences (attention_f16.hip RagMap).
struct AttRagged {
    const int32_t* seq_off;        // [sequences] first packed row of the sequence (its token seq_p): residual stream, q | k | v inputs, context
    const int32_t* seq_p;          // [sequences] first token the sequence owns (0 for a root); a = seq_p rounded down to a multiple of 32
    const int32_t* seq_q;          // [sequences] row of the q | k operand planes of the sequence's token a
    const int32_t* seq_root;       // [sequences] the sequence whose rows stand for the tokens before seq_p
    const uint32_t* seq_vt;        // [sequences] offset (halfs, per plane) of the sequence's V^T block [H][64][roundup(T - a, 32)]
    const int32_t* tile_seq;       // [n_tiles] 32-token tiles from token a on: sequence, tile index
    const int32_t* tile_j;
    const int32_t* blk_seq;        // [n_blocks] blocks of att16_waves_per_block(T) query tiles: sequence, block index inside the suffix
    const int32_t* blk_j;
    int n_tiles, n_blocks;
};
int att16_waves_per_block(int T);
int att_set_option(const char* name, long long value);      // "att_xcd_local": block order of the dense attention launches (A/B only)
int launch_attention_tr_ragged(const float* qkv, const float* conv, const float* slopes, int T, int H, const AttRagged& rg,
                               unsigned short* qk16, size_t qk_plane, unsigned short* vt16, size_t vt_plane, unsigned short* ctx16,
                               size_t plane, hipStream_t s);

}  // namespace pgmi
