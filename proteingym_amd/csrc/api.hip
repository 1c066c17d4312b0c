// C ABI of libpgmi.so (include/pgmi.h) and the host-side orchestration of the ESM forward.
//
// Forward order follows /root/reference/proteingym/baselines/esm/esm/model/esm1.py:116-177 and
// esm/model/esm2.py:76-130; the per-layer order follows esm/modules.py:120-142.
#include <math.h>
#include <stdlib.h>
#include <cmath>
#include <algorithm>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "common.h"
#include "gemm16x_kernel.h"          // XMap (the tied row attention's operand maps); no kernel is instantiated here

namespace pgmi {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// 16-bit operand of one Linear weight [N,K]: f16x3 = fp16 (hi, lo) of W*2^s in the K-interleaved layout (common.h
// ki_off: per row, groups of 32 hi halfs + 32 lo halfs), bf16 = one plane; out_scale = 2^-s is applied in the GEMM epilogue.
struct W16 {
    unsigned short* p = nullptr;
    size_t plane = 0;
    float out_scale = 1.0f;
};

struct Layer {
    float *ln1_w, *ln1_b, *wqkv, *bqkv, *wo, *bo, *ln2_w, *ln2_b, *w1, *b1, *w2, *b2;
    W16 wqkv16, wo16, w116, w216;
    float* conv = nullptr;        // Tranception: [3][4][64][8] right-aligned 7-tap filters + bias (attention_f16.hip)
    // MSA Transformer: ln1/wqkv/wo = tied row attention, c_* = column attention, ln2/w1/w2 = feed forward
    float *c_ln_w = nullptr, *c_ln_b = nullptr, *c_bqkv = nullptr, *c_bo = nullptr;
    W16 c_wqkv16, c_wo16;
};

struct ProfEvent {
    hipEvent_t start, stop;
    int cls;
};

}  // namespace pgmi

using namespace pgmi;

struct pgmi_assay;
struct pgmi_pppl;

struct pgmi_model {
    pgmi_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<void*> allocs;          // everything to hipFree
    std::vector<pgmi_assay*> assays;    // live assays created on this model (orphaned on destroy)
    std::vector<pgmi_pppl*> pppls;      // live pseudo-ppl libraries (same rule)
    // weights
    float *embed_tokens = nullptr, *embed_positions = nullptr;
    float *lnb_w = nullptr, *lnb_b = nullptr, *lna_w = nullptr, *lna_b = nullptr;
    float *hd_w = nullptr, *hd_b = nullptr, *hln_w = nullptr, *hln_b = nullptr, *h_bias = nullptr;
    std::vector<Layer> layers;
    W16 hd16;
    float *tr_lm_head = nullptr, *tr_zero_bias = nullptr, *tr_slopes = nullptr;   // Tranception head / ALiBi slopes
    float* tr_prior = nullptr;                          // device copy of the retrieval log-prior [P,V]
    int tr_prior_rows = 0;
    int32_t* tr_meta = nullptr;                         // prefix-shared scoring: the chunk's index arrays (TrChunk)
    size_t tr_meta_cap = 0;
    // MSA Transformer
    float* msa_pe = nullptr;                            // msa_position_embedding [1024, D]
    float* xt = nullptr;                                // residual stream in column-major token order
    int32_t *msa_full = nullptr, *msa_kv_len = nullptr; // device copy of the MSA token grid; [C] = R
    size_t msa_full_cap = 0;
    float *tied_part = nullptr, *tied_p = nullptr, *tied_vt = nullptr;   // split-K scores, probabilities, V^T
    size_t tied_part_cap = 0, tied_p_cap = 0, tied_vt_cap = 0;
    int msa_kv_R = 0, msa_kv_C = 0;
    float ln_eps = 1e-5f;
    unsigned short *h16 = nullptr, *g16 = nullptr;     // activation planes [planes][R*D], [planes][R*F]
    size_t h16_plane = 0, g16_plane = 0;
    unsigned short *qk16 = nullptr, *vt16 = nullptr;   // attention operands (f16x3): [2][R*2D], [2][R*D]
    size_t qk16_plane = 0, vt16_plane = 0;
    int32_t* nonfinite = nullptr;
    int gemm_variant = 0;
    int keep_rows = 1;                                 // last layer's row-local stages on the kept rows only (PGMI_KEEP_ROWS)
    int last_B = 0, last_T = 0;
    int dh = kHeadDim;    // true head dim; heads are laid out in 64-lane slot groups (pgmi_model_create)
    int rot_halves = 1;   // slot groups per head: 1, or 2 for head_dim 128
    int Hs = 0;           // slot groups per token = heads * rot_halves
    int Da = 0;           // attention width = heads * 64 (== embed_dim when dh == 64)
    float *rot_cos = nullptr, *rot_sin = nullptr;
    int rot_len = 0;
    // workspace
    int max_rows = 0;
    float *x = nullptr, *h = nullptr, *qkv = nullptr, *g = nullptr, *lp = nullptr, *denom = nullptr;
    int32_t *tokens = nullptr, *pos_idx = nullptr, *kv_len = nullptr, *row_idx = nullptr, *aux_i = nullptr;
    // profiling
    bool prof = false;
    std::vector<ProfEvent> events;
    size_t events_used = 0;
    double prof_ms[PGMI_K_COUNT] = {0};
    int64_t prof_n[PGMI_K_COUNT] = {0};
    double prof_flops[PGMI_K_COUNT] = {0};
    double prof_bytes[PGMI_K_COUNT] = {0};
};

struct pgmi_assay {
    pgmi_model* m = nullptr;
    int n_tok = 0, P = 0, T = 0;
    int64_t n_mut = 0, n_sub = 0;
    std::vector<void*> allocs;
    int32_t *wt = nullptr, *positions = nullptr, *win_start = nullptr, *mask_rel = nullptr;
    int32_t *sub_pos = nullptr, *sub_wt = nullptr, *sub_mt = nullptr;
    int64_t* mut_off = nullptr;
    float* table = nullptr;
    double* scores = nullptr;
};

// A library of variable-length sequences resident in HBM for pseudo-perplexity scoring (config 5).
struct pgmi_pppl {
    pgmi_model* m = nullptr;
    int64_t N = 0;
    std::vector<int64_t> off;           // host copy of seq_off [N+1]
    std::vector<void*> allocs;
    uint8_t* tok8 = nullptr;            // all tokens, one byte each
    int64_t* off_dev = nullptr;
    int64_t last_rows = 0, last_chunks = 0, last_tokens = 0, last_padded = 0;   // statistics of the last run
};

namespace {

template <typename T>
int dev_alloc(std::vector<void*>& pool, T** p, size_t n) {
    void* q = nullptr;
    if (n == 0) n = 1;
    hipError_t e = hipMalloc(&q, n * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
        return PGMI_ENOMEM;
    }
    pool.push_back(q);
    *p = static_cast<T*>(q);
    return PGMI_OK;
}

template <typename T>
int dev_upload(std::vector<void*>& pool, T** p, const T* host, size_t n) {
    int rc = dev_alloc(pool, p, n);
    if (rc) return rc;
    if (n) PGMI_HIP(hipMemcpy(*p, host, n * sizeof(T), hipMemcpyHostToDevice));
    return PGMI_OK;
}

struct ProfScope {
    pgmi_model* m;
    ProfEvent* ev = nullptr;
    ProfScope(pgmi_model* m_, int cls, double flops, double bytes) : m(m_) {
        if (!m->prof) return;
        if (m->events_used == m->events.size()) {
            ProfEvent e;
            if (hipEventCreate(&e.start) != hipSuccess || hipEventCreate(&e.stop) != hipSuccess) return;
            m->events.push_back(e);
        }
        ev = &m->events[m->events_used++];
        ev->cls = cls;
        m->prof_n[cls] += 1;
        m->prof_flops[cls] += flops;
        m->prof_bytes[cls] += bytes;
        hipEventRecord(ev->start, m->stream);
    }
    ~ProfScope() {
        if (ev) hipEventRecord(ev->stop, m->stream);
    }
};

int prof_drain(pgmi_model* m) {
    if (m->events_used == 0) return PGMI_OK;
    PGMI_HIP(hipStreamSynchronize(m->stream));
    for (size_t i = 0; i < m->events_used; ++i) {
        float ms = 0.f;
        PGMI_HIP(hipEventElapsedTime(&ms, m->events[i].start, m->events[i].stop));
        m->prof_ms[m->events[i].cls] += ms;
    }
    m->events_used = 0;
    return PGMI_OK;
}

int check_cfg(const pgmi_config* c) {
    if (!c) { set_error("null config"); return PGMI_EINVAL; }
    if (c->abi_version != PGMI_ABI_VERSION) { set_error("ABI version mismatch: got %d, library is %d", c->abi_version, PGMI_ABI_VERSION); return PGMI_EINVAL; }
    if (c->arch != PGMI_ARCH_ESM1B && c->arch != PGMI_ARCH_ESM2 && c->arch != PGMI_ARCH_TRANCEPTION && c->arch != PGMI_ARCH_MSA) { set_error("unknown arch %d", c->arch); return PGMI_EINVAL; }
    if (c->layers <= 0 || c->embed_dim <= 0 || c->heads <= 0 || c->ffn_dim <= 0) { set_error("non-positive model dimension"); return PGMI_EINVAL; }
    {
        // head_dim 64 natively; smaller head dims (ESM2 8M/35M/150M: 16/24/32) run zero-padded to 64 lanes per head;
        // head_dim 128 (ESM2-15B: pretrained.py:387-394) as two 64-lane slot groups per head (see pgmi_model_create)
        const int dh = c->embed_dim / c->heads;
        const bool esm = c->arch == PGMI_ARCH_ESM1B || c->arch == PGMI_ARCH_ESM2;
        const bool ok = c->embed_dim % c->heads == 0 &&
                        (dh == kHeadDim || (dh < kHeadDim && dh % 2 == 0 && esm) || (dh == 2 * kHeadDim && esm));
        if (!ok) { set_error("unsupported head_dim %d (embed_dim %d / heads %d): this build supports head_dim 64, even head dims below 64 and head_dim 128 (ESM)", dh, c->embed_dim, c->heads); return PGMI_EINVAL; }
    }
    if (c->embed_dim % 32 || c->ffn_dim % 32) { set_error("embed_dim and ffn_dim must be multiples of 32"); return PGMI_EINVAL; }
    if (c->arch == PGMI_ARCH_TRANCEPTION) {
        if (c->vocab != 25) { set_error("Tranception vocab must be 25"); return PGMI_EINVAL; }
        if (c->heads % 4) { set_error("Invalid number of heads. Tranception requires the number of heads to be a multiple of 4."); return PGMI_EINVAL; }
        if (c->precision != PGMI_PREC_F16X3) { set_error("Tranception is available in precision f16x3 only"); return PGMI_EINVAL; }
        if (c->max_positions <= 0) { set_error("Tranception needs max_positions = n_ctx"); return PGMI_EINVAL; }
    } else if (c->vocab != PGMI_VOCAB) { set_error("vocab must be %d", PGMI_VOCAB); return PGMI_EINVAL; }
    if (c->arch == PGMI_ARCH_ESM1B && c->max_positions <= 0) { set_error("ESM-1b arch needs max_positions"); return PGMI_EINVAL; }
    if (c->arch == PGMI_ARCH_MSA) {
        if (c->max_positions <= 0) { set_error("MSA Transformer needs max_positions"); return PGMI_EINVAL; }
        if (c->embed_dim != c->heads * kHeadDim) { set_error("MSA Transformer: head_dim must be 64"); return PGMI_EINVAL; }
        if (c->precision != PGMI_PREC_F16X3) { set_error("MSA Transformer is available in precision f16x3 only"); return PGMI_EINVAL; }
    }
    if (c->precision != PGMI_PREC_FP32 && c->precision != PGMI_PREC_F16X3 && c->precision != PGMI_PREC_BF16) { set_error("unknown precision %d", c->precision); return PGMI_EINVAL; }
    // f16x3: K tiles of 32 (checked above); the bf16 GEMM's K tile is 64
    if (c->precision == PGMI_PREC_BF16 && (c->embed_dim % 64 || c->ffn_dim % 64)) { set_error("precision bf16 needs embed_dim and ffn_dim to be multiples of 64"); return PGMI_EINVAL; }
    return PGMI_OK;
}

// trailing-only padding, at least one real token per sequence
int check_tokens(const int32_t* tokens, int B, int T) {
    for (int b = 0; b < B; ++b) {
        const int32_t* t = tokens + (size_t)b * T;
        bool seen_pad = false;
        if (t[0] == PGMI_TOK_PAD) { set_error("sequence %d is empty (all <pad>)", b); return PGMI_EINVAL; }
        for (int i = 0; i < T; ++i) {
            if (t[i] < 0 || t[i] >= PGMI_VOCAB) { set_error("token id %d out of range at [%d,%d]", t[i], b, i); return PGMI_EINVAL; }
            if (t[i] == PGMI_TOK_PAD) seen_pad = true;
            else if (seen_pad) { set_error("interior <pad> at [%d,%d]: only trailing padding is supported", b, i); return PGMI_EINVAL; }
        }
    }
    return PGMI_OK;
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// Upload a Linear weight [n_elems] as 16-bit planes.  f16x3: W*2^s with 2^s chosen so that
// max|W|*2^s lies in [8192, 16384): hi stays far below fp16's 65504 and lo = fp16(W' - hi) stays in
// the normal range for every element within 2^-15 of the largest.  bf16: one plane, no scaling.
int make_w16(std::vector<void*>& pool, const float* host, size_t n, size_t K, int precision, hipStream_t s, W16* out) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, fabsf(host[i]));
    float scale = 1.0f;
    const int planes = (precision == PGMI_PREC_F16X3) ? 2 : 1;
    if (precision == PGMI_PREC_F16X3 && mx > 0.f && std::isfinite(mx)) scale = exp2f(floorf(log2f(16384.0f / mx)));
    float* tmp = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&tmp), n * sizeof(float));
    if (e != hipSuccess) { set_error("hipMalloc failed: %s", hipGetErrorString(e)); return PGMI_ENOMEM; }
    int rc = dev_alloc(pool, &out->p, n * planes);
    if (rc) { hipFree(tmp); return rc; }
    e = hipMemcpy(tmp, host, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_split16(tmp, (int64_t)n, scale, precision == PGMI_PREC_BF16 ? 1 : 0, (int)K, out->p, s);
        e = hipStreamSynchronize(s);
    }
    hipFree(tmp);
    if (e != hipSuccess) { set_error("weight split failed: %s", hipGetErrorString(e)); return PGMI_EHIP; }
    out->plane = n;
    out->out_scale = 1.0f / scale;
    return PGMI_OK;
}

int ensure_rotary(pgmi_model* m, int T) {
    if (m->cfg.arch != PGMI_ARCH_ESM2 || T <= m->rot_len) return PGMI_OK;
    // rotary_embedding.py:40,52-58: inv_freq = 1/10000^(2i/d) in f32; freqs = t * inv_freq (f32);
    // emb = cat(freqs, freqs); cos/sin taken in f32.
    const int n = std::max(T, 1026);
    const int rh = m->rot_halves;                          // table rows per token: slot-group parity for head_dim 128
    std::vector<float> c((size_t)n * rh * 64), s((size_t)n * rh * 64);
    float inv[64];
    const int half = m->dh / 2;                            // rotary pairs are (j, j + dh/2)
    for (int i = 0; i < half; ++i) inv[i] = 1.0f / powf(10000.0f, (float)(2 * i) / (float)m->dh);
    // slots i and 32+i of slot group g hold dims j and j + dh/2 with j = i (dh <= 64) or 32 g + i (dh 128)
    for (int t = 0; t < n; ++t)
        for (int g = 0; g < rh; ++g)
            for (int i = 0; i < 32; ++i) {
                const int j = (rh == 1) ? i : 32 * g + i;
                const float f = j < half ? (float)t * inv[j] : 0.0f;       // padded slots (dh < 64): cos 1, sin 0
                const size_t o = ((size_t)t * rh + g) * 64;
                c[o + i] = c[o + 32 + i] = cosf(f);
                s[o + i] = s[o + 32 + i] = sinf(f);
            }
    int rc = dev_upload(m->allocs, &m->rot_cos, c.data(), c.size());
    if (rc) return rc;
    rc = dev_upload(m->allocs, &m->rot_sin, s.data(), s.size());
    if (rc) return rc;
    m->rot_len = n;
    return PGMI_OK;
}

// y = epi(in W^T + b) (+ residual).  fp32 mode: in32 -> fp32 out.  16-bit modes: in16 planes ->
// either fp32 out (out32) or 16-bit planes (out16).
int linear(pgmi_model* m, const float* in32, const unsigned short* in16, size_t in_plane, const float* W32,
           const W16& w16, const float* bias, const float* residual, float* out32, unsigned short* out16,
           size_t out_plane, int M, int N, int K, int epi) {
    if (m->cfg.precision == PGMI_PREC_FP32)
        return launch_gemm_f32(in32, W32, bias, residual, out32, M, N, K, epi, m->stream);
    const bool bf = m->cfg.precision == PGMI_PREC_BF16;
    return launch_gemm16(in16, in_plane, w16.p, w16.plane, bias, residual, out32, out16, out_plane, M, N, K, epi,
                         w16.out_scale, bf ? 1 : 2, bf, m->gemm_variant, m->stream);
}

// Runs the encoder on tokens already in m->tokens [B,T]; leaves the residual stream in m->x.
// keep != nullptr (device, n_keep row indices into [B*T]): the caller reads only these rows of the output (the masked
// position of every sequence: compute_fitness.py:503 `token_probs[:, i]`, :274-276).  Everything after the last layer's
// attention is row-local (out-projection, LayerNorm, FFN: modules.py:126-141), so the last layer gathers the kept rows
// of the attention context and of the residual stream and runs those stages on n_keep rows; m->x then holds the kept
// rows COMPACTED (row j = keep[j]) and *compacted is set.  The kept rows are bit-identical to the full evaluation: every
// kernel on the way computes a row from that row's inputs only, in an order that does not depend on the row count
// (tests/test_gpu_esm.py::test_last_layer_kept_rows_bit_identical).  PGMI_KEEP_ROWS=0 turns it off.
int run_encoder(pgmi_model* m, int B, int T, const int32_t* keep = nullptr, int n_keep = 0, bool* compacted = nullptr) {
    const pgmi_config& c = m->cfg;
    const int M = B * T, D = c.embed_dim, F = c.ffn_dim, H = c.heads, Da = m->Da;
    hipStream_t s = m->stream;
    if (c.arch == PGMI_ARCH_ESM1B && T > c.max_positions) {
        set_error("Sequence length %d above maximum sequence length of %d", T, c.max_positions);   // modules.py:256-260
        return PGMI_EINVAL;
    }
    int rc = ensure_rotary(m, T);
    if (rc) return rc;
    if (m->vt16 && (B != m->last_B || T != m->last_T)) {
        // pad keys (t >= T inside the last 32-key tile) are never written by the fused QKV epilogue:
        // they must hold finite data (their softmax weight is exactly 0)
        PGMI_HIP(hipMemsetAsync(m->vt16, 0, m->vt16_plane * 2 * sizeof(unsigned short), s));
        m->last_B = B;
        m->last_T = T;
    }
    {
        ProfScope p(m, PGMI_K_EMBED, 0, (double)M * D * 4);
        launch_seq_stats(m->tokens, B, T, c.token_dropout, m->denom, m->pos_idx, m->kv_len, s);
        launch_embed(m->tokens, m->denom, m->pos_idx, m->embed_tokens, m->embed_positions, c.token_dropout, M, T, D, m->x, s);
        if (c.emb_layer_norm_before) {
            launch_layernorm(m->x, m->lnb_w, m->lnb_b, M, D, 1e-5f, m->x, s);
            launch_zero_pad_rows(m->tokens, M, D, m->x, s);
        }
    }
    const double ln_bytes = 2.0 * M * D * 4;
    const int prec = c.precision;
    const int mode16 = (prec == PGMI_PREC_F16X3) ? 1 : 2;          // LN / attention output mode
    for (int l = 0; l < c.layers; ++l) {
        const Layer& L = m->layers[l];
        { ProfScope p(m, PGMI_K_LAYERNORM, 0, ln_bytes);
          if (prec == PGMI_PREC_FP32) launch_layernorm(m->x, L.ln1_w, L.ln1_b, M, D, 1e-5f, m->h, s);
          else launch_layernorm16(m->x, L.ln1_w, L.ln1_b, M, D, 1e-5f, m->h16, m->h16_plane, mode16, s); }
        const bool fused_qkv = prec == PGMI_PREC_F16X3;          // attention operands straight from the QKV projection's epilogue
        { ProfScope p(m, PGMI_K_GEMM_QKV, 2.0 * M * 3 * D * D, 0);
          if (fused_qkv)
              rc = launch_gemm16_qkv(m->h16, m->h16_plane, L.wqkv16.p, L.wqkv16.plane, L.bqkv, M, Da, D, L.wqkv16.out_scale,
                                     m->qk16, m->qk16_plane, m->vt16, m->vt16_plane, m->rot_cos, m->rot_sin,
                                     c.arch == PGMI_ARCH_ESM2, T, m->Hs, m->gemm_variant, s, m->rot_halves);
          else
              rc = linear(m, m->h, m->h16, m->h16_plane, L.wqkv, L.wqkv16, L.bqkv, nullptr, m->qkv, nullptr, 0, M, 3 * Da, D, EPI_NONE);
          if (rc) return rc; }
        { ProfScope p(m, PGMI_K_ATTENTION, 4.0 * M * T * D, 0);
          const bool v2 = prec == PGMI_PREC_F16X3;
          if (c.arch == PGMI_ARCH_ESM2 && !v2) launch_rotary(m->qkv, m->rot_cos, m->rot_sin, M, T, m->Hs, s, m->rot_halves);
          if (v2)
              rc = launch_attention_f16x3_v2(fused_qkv ? nullptr : m->qkv, m->kv_len, m->rot_cos, m->rot_sin, c.arch == PGMI_ARCH_ESM2, B, T, H,
                                             m->qk16, m->qk16_plane, m->vt16, m->vt16_plane, nullptr, m->h16,
                                             m->h16_plane, 1, s, nullptr, nullptr, m->rot_halves * kHeadDim);
          else
              rc = launch_attention_f32(m->qkv, m->kv_len, B, T, H, m->h, m->h16, m->h16_plane,
                                        prec == PGMI_PREC_FP32 ? 0 : mode16, s, m->rot_halves * kHeadDim);
          if (rc) return rc; }
        if (keep && m->keep_rows && l == c.layers - 1) {
            const int R = n_keep;
            ProfScope p(m, PGMI_K_KEPT_ROWS, 2.0 * R * D * (Da + 2.0 * F), 0);
            launch_gather_rows(m->x, keep, R, D, m->qkv, s);                       // residual rows (qkv is free after attention)
            if (prec == PGMI_PREC_FP32) launch_gather_rows(m->h, keep, R, Da, m->g, s);
            else        // a 16-bit context row is one contiguous run (K-interleaved hi|lo: 4 Da bytes; bf16: 2 Da bytes)
                launch_gather_rows(reinterpret_cast<const float*>(m->h16), keep, R, prec == PGMI_PREC_F16X3 ? Da : Da / 2,
                                   reinterpret_cast<float*>(m->g16), s);
            rc = linear(m, m->g, m->g16, m->g16_plane, L.wo, L.wo16, L.bo, m->qkv, m->x, nullptr, 0, R, D, Da, EPI_NONE);
            if (rc) return rc;
            if (prec == PGMI_PREC_FP32) launch_layernorm(m->x, L.ln2_w, L.ln2_b, R, D, 1e-5f, m->h, s);
            else launch_layernorm16(m->x, L.ln2_w, L.ln2_b, R, D, 1e-5f, m->h16, m->h16_plane, mode16, s);
            rc = linear(m, m->h, m->h16, m->h16_plane, L.w1, L.w116, L.b1, nullptr,
                        prec == PGMI_PREC_FP32 ? m->g : nullptr, prec == PGMI_PREC_FP32 ? nullptr : m->g16, m->g16_plane,
                        R, F, D, EPI_GELU);
            if (rc) return rc;
            rc = linear(m, m->g, m->g16, m->g16_plane, L.w2, L.w216, L.b2, m->x, m->x, nullptr, 0, R, D, F, EPI_NONE);
            if (rc) return rc;
            if (compacted) *compacted = true;
            break;
        }
        { ProfScope p(m, PGMI_K_GEMM_OUT, 2.0 * M * D * D, 0);
          rc = linear(m, m->h, m->h16, m->h16_plane, L.wo, L.wo16, L.bo, m->x, m->x, nullptr, 0, M, D, Da, EPI_NONE);
          if (rc) return rc; }
        { ProfScope p(m, PGMI_K_LAYERNORM, 0, ln_bytes);
          if (prec == PGMI_PREC_FP32) launch_layernorm(m->x, L.ln2_w, L.ln2_b, M, D, 1e-5f, m->h, s);
          else launch_layernorm16(m->x, L.ln2_w, L.ln2_b, M, D, 1e-5f, m->h16, m->h16_plane, mode16, s); }
        { ProfScope p(m, PGMI_K_GEMM_FC1, 2.0 * M * F * D, 0);
          rc = linear(m, m->h, m->h16, m->h16_plane, L.w1, L.w116, L.b1, nullptr,
                      prec == PGMI_PREC_FP32 ? m->g : nullptr, prec == PGMI_PREC_FP32 ? nullptr : m->g16, m->g16_plane,
                      M, F, D, EPI_GELU);
          if (rc) return rc; }
        { ProfScope p(m, PGMI_K_GEMM_FC2, 2.0 * M * F * D, 0);
          rc = linear(m, m->g, m->g16, m->g16_plane, L.w2, L.w216, L.b2, m->x, m->x, nullptr, 0, M, D, F, EPI_NONE);
          if (rc) return rc; }
    }
    PGMI_HIP(hipGetLastError());
    return PGMI_OK;
}

// LM head (modules.py:322-328) + log-softmax on R rows.  If row_idx != null the rows are
// gathered from m->x first (masked positions only), else R must be the full M rows of m->x.
// Result in m->lp [R,V].
int run_head(pgmi_model* m, int R, const int32_t* row_idx) {
    const pgmi_config& c = m->cfg;
    const int D = c.embed_dim;
    hipStream_t s = m->stream;
    ProfScope p(m, PGMI_K_HEAD, 2.0 * R * D * (D + c.vocab), 0);
    const int prec = c.precision;
    const float* src = m->x;
    if (row_idx) {
        launch_gather_rows(m->x, row_idx, R, D, m->h, s);
        src = m->h;
    }
    if (prec == PGMI_PREC_FP32) launch_layernorm(src, m->lna_w, m->lna_b, R, D, 1e-5f, m->h, s);
    else launch_layernorm16(src, m->lna_w, m->lna_b, R, D, 1e-5f, m->h16, m->h16_plane, prec == PGMI_PREC_F16X3 ? 1 : 2, s);
    int rc = linear(m, m->h, m->h16, m->h16_plane, m->hd_w, m->hd16, m->hd_b, nullptr, m->g, nullptr, 0, R, D, D, EPI_GELU);
    if (rc) return rc;
    launch_layernorm(m->g, m->hln_w, m->hln_b, R, D, 1e-5f, m->g, s);
    launch_vocab_logsoftmax(m->g, m->embed_tokens, m->h_bias, R, D, c.vocab, m->lp, m->nonfinite, s);
    PGMI_HIP(hipGetLastError());
    return PGMI_OK;
}


// Encoder + LM head where only the rows row_idx [R] (device) of the [B*T] outputs are read.  Result in m->lp [R,V].
int run_rows(pgmi_model* m, int B, int T, int R, const int32_t* row_idx) {
    bool compacted = false;
    int rc = run_encoder(m, B, T, row_idx, R, &compacted);
    if (rc) return rc;
    return run_head(m, R, compacted ? nullptr : row_idx);
}

// ALiBi slopes, grouped: tranception/model_pytorch.py:50-71 (get_slopes(n, "grouped_alibi"))
static void alibi_slopes_pow2(int n, std::vector<double>& out) {
    const double start = pow(2.0, -pow(2.0, -(log2((double)n) - 3.0)));
    double v = start;
    for (int i = 0; i < n; ++i) { out.push_back(v); v *= start; }
}
static std::vector<double> alibi_slopes(int n) {
    std::vector<double> r;
    const double l2 = log2((double)n);
    if (l2 == floor(l2)) { alibi_slopes_pow2(n, r); return r; }
    const int c = 1 << (int)floor(l2);
    alibi_slopes_pow2(c, r);
    std::vector<double> e = alibi_slopes(2 * c);
    for (int i = 0; i < (int)e.size() && (int)r.size() < n; i += 2) r.push_back(e[i]);
    return r;
}

// transpose an HF Conv1D weight [in,out] into nn.Linear layout [out,in], optionally scaling the
// first `scaled_cols` output columns (the q block) by `scale`
static void conv1d_to_linear(const float* w, size_t in, size_t out, size_t scaled_cols, float scale, std::vector<float>& dst) {
    dst.resize(in * out);
    for (size_t o = 0; o < out; ++o) {
        const float sc = o < scaled_cols ? scale : 1.0f;
        for (size_t i = 0; i < in; ++i) dst[o * in + i] = w[i * out + o] * sc;
    }
}

int create_tranception(pgmi_model* m, const pgmi_config* cfg, const float* w, int64_t n_weights) {
    const size_t D = cfg->embed_dim, F = cfg->ffn_dim, V = cfg->vocab, H = cfg->heads;
    const float* p = w;
    int rc = 0;
#define TRY(e) do { rc = (e); if (rc) return rc; } while (0)
    TRY(dev_upload(m->allocs, &m->embed_tokens, p, V * D)); p += V * D;
    const float qscale = 1.0f / sqrtf((float)kHeadDim);
    m->layers.resize(cfg->layers);
    std::vector<float> lin, bq(3 * D), conv(3 * 4 * 64 * 8);
    static const int ksz[3] = {3, 5, 7};
    for (int l = 0; l < cfg->layers; ++l) {
        Layer& L = m->layers[l];
        TRY(dev_upload(m->allocs, &L.ln1_w, p, D)); p += D;
        TRY(dev_upload(m->allocs, &L.ln1_b, p, D)); p += D;
        conv1d_to_linear(p, D, 3 * D, D, qscale, lin); p += D * 3 * D;
        TRY(make_w16(m->allocs, lin.data(), lin.size(), D, cfg->precision, m->stream, &L.wqkv16));
        for (size_t i = 0; i < 3 * D; ++i) bq[i] = p[i] * (i < D ? qscale : 1.0f);
        p += 3 * D;
        TRY(dev_upload(m->allocs, &L.bqkv, bq.data(), bq.size()));
        // conv table: group 0 = identity; groups 1..3 = kernels 3,5,7 right-aligned in 7 taps
        std::fill(conv.begin(), conv.end(), 0.0f);
        for (int which = 0; which < 3; ++which) {
            for (int d = 0; d < 64; ++d) conv[((which * 4 + 0) * 64 + d) * 8 + 6] = 1.0f;
            for (int ki = 0; ki < 3; ++ki) {
                const int k = ksz[ki];
                for (int d = 0; d < 64; ++d)
                    for (int j = 0; j < k; ++j) conv[((which * 4 + ki + 1) * 64 + d) * 8 + (7 - k) + j] = p[d * k + j];
                p += 64 * k;
                // the q projection is pre-scaled by 1/sqrt(dh): scale the q-conv bias the same way
                for (int d = 0; d < 64; ++d) conv[((which * 4 + ki + 1) * 64 + d) * 8 + 7] = p[d] * (which == 0 ? qscale : 1.0f);
                p += 64;
            }
        }
        TRY(dev_upload(m->allocs, &L.conv, conv.data(), conv.size()));
        conv1d_to_linear(p, D, D, 0, 1.0f, lin); p += D * D;
        TRY(make_w16(m->allocs, lin.data(), lin.size(), D, cfg->precision, m->stream, &L.wo16));
        TRY(dev_upload(m->allocs, &L.bo, p, D)); p += D;
        TRY(dev_upload(m->allocs, &L.ln2_w, p, D)); p += D;
        TRY(dev_upload(m->allocs, &L.ln2_b, p, D)); p += D;
        conv1d_to_linear(p, D, F, 0, 1.0f, lin); p += D * F;
        TRY(make_w16(m->allocs, lin.data(), lin.size(), D, cfg->precision, m->stream, &L.w116));
        TRY(dev_upload(m->allocs, &L.b1, p, F)); p += F;
        conv1d_to_linear(p, F, D, 0, 1.0f, lin); p += F * D;
        TRY(make_w16(m->allocs, lin.data(), lin.size(), F, cfg->precision, m->stream, &L.w216));
        TRY(dev_upload(m->allocs, &L.b2, p, D)); p += D;
    }
    TRY(dev_upload(m->allocs, &m->lna_w, p, D)); p += D;
    TRY(dev_upload(m->allocs, &m->lna_b, p, D)); p += D;
    TRY(dev_upload(m->allocs, &m->tr_lm_head, p, V * D)); p += V * D;
    if (p - w != n_weights) { set_error("internal: blob walk mismatch"); return PGMI_EINVAL; }
    std::vector<float> zb(V, 0.0f), sl;
    TRY(dev_upload(m->allocs, &m->tr_zero_bias, zb.data(), zb.size()));
    std::vector<double> quarter = alibi_slopes((int)H / 4);          // grouped: slopes of n/4 heads, tiled 4x
    for (int rep = 0; rep < 4; ++rep)
        for (double v : quarter) sl.push_back((float)v);
    TRY(dev_upload(m->allocs, &m->tr_slopes, sl.data(), sl.size()));
#undef TRY
    return PGMI_OK;
}

// Tranception forward on tokens in m->tokens [B,T]; leaves log-probabilities in m->lp [B*T, V].
int run_tranception(pgmi_model* m, int B, int T) {
    const pgmi_config& c = m->cfg;
    const int M = B * T, D = c.embed_dim, F = c.ffn_dim, H = c.heads;
    hipStream_t s = m->stream;
    if (T > c.max_positions) { set_error("sequence of %d tokens exceeds the model context n_ctx=%d", T, c.max_positions); return PGMI_EINVAL; }
    int rc = 0;
    if (B != m->last_B || T != m->last_T) {
        PGMI_HIP(hipMemsetAsync(m->vt16, 0, m->vt16_plane * 2 * sizeof(unsigned short), s));
        m->last_B = B;
        m->last_T = T;
    }
    { ProfScope p(m, PGMI_K_EMBED, 0, (double)M * D * 4);
      launch_gather_rows(m->embed_tokens, m->tokens, M, D, m->x, s); }       // wte[input_ids]; no positional embedding
    const double ln_bytes = 2.0 * M * D * 4;
    for (int l = 0; l < c.layers; ++l) {
        const Layer& L = m->layers[l];
        { ProfScope p(m, PGMI_K_LAYERNORM, 0, ln_bytes);
          launch_layernorm16(m->x, L.ln1_w, L.ln1_b, M, D, m->ln_eps, m->h16, m->h16_plane, 1, s); }
        { ProfScope p(m, PGMI_K_GEMM_QKV, 2.0 * M * 3 * D * D, 0);
          rc = linear(m, nullptr, m->h16, m->h16_plane, nullptr, L.wqkv16, L.bqkv, nullptr, m->qkv, nullptr, 0, M, 3 * D, D, EPI_NONE);
          if (rc) return rc; }
        { ProfScope p(m, PGMI_K_ATTENTION, 2.0 * M * T * D, 0);
          rc = launch_attention_f16x3_v2(m->qkv, nullptr, nullptr, nullptr, 0, B, T, H, m->qk16, m->qk16_plane, m->vt16,
                                         m->vt16_plane, nullptr, m->h16, m->h16_plane, 1, s, L.conv, m->tr_slopes);
          if (rc) return rc; }
        { ProfScope p(m, PGMI_K_GEMM_OUT, 2.0 * M * D * D, 0);
          rc = linear(m, nullptr, m->h16, m->h16_plane, nullptr, L.wo16, L.bo, m->x, m->x, nullptr, 0, M, D, D, EPI_NONE);
          if (rc) return rc; }
        { ProfScope p(m, PGMI_K_LAYERNORM, 0, ln_bytes);
          launch_layernorm16(m->x, L.ln2_w, L.ln2_b, M, D, m->ln_eps, m->h16, m->h16_plane, 1, s); }
        { ProfScope p(m, PGMI_K_GEMM_FC1, 2.0 * M * F * D, 0);
          rc = linear(m, nullptr, m->h16, m->h16_plane, nullptr, L.w116, L.b1, nullptr, nullptr, m->g16, m->g16_plane, M, F, D, EPI_SQRELU);
          if (rc) return rc; }
        { ProfScope p(m, PGMI_K_GEMM_FC2, 2.0 * M * F * D, 0);
          rc = linear(m, nullptr, m->g16, m->g16_plane, nullptr, L.w216, L.b2, m->x, m->x, nullptr, 0, M, D, F, EPI_NONE);
          if (rc) return rc; }
    }
    { ProfScope p(m, PGMI_K_HEAD, 2.0 * M * D * c.vocab, 0);
      launch_layernorm(m->x, m->lna_w, m->lna_b, M, D, m->ln_eps, m->h, s);
      launch_vocab_logsoftmax(m->h, m->tr_lm_head, m->tr_zero_bias, M, D, c.vocab, m->lp, m->nonfinite, s); }
    PGMI_HIP(hipGetLastError());
    return PGMI_OK;
}


// ---- MSA Transformer -------------------------------------------------------------------------------
// Blob order (include/pgmi.h): embed_tokens, embed_positions [(max_positions+2), D], msa_position_embedding
// [1024, D], emb_layer_norm_before; per layer {row attention: ln, q, k, v, out; column attention: same;
// feed forward: ln, fc1, fc2}; emb_layer_norm_after; lm_head dense, layer_norm, bias.
int create_msa(pgmi_model* m, const pgmi_config* cfg, const float* w, int64_t n_weights) {
    const size_t D = cfg->embed_dim, F = cfg->ffn_dim, V = cfg->vocab;
    const float* p = w;
    int rc = 0;
#define TRY(e) do { rc = (e); if (rc) return rc; } while (0)
    TRY(dev_upload(m->allocs, &m->embed_tokens, p, V * D)); p += V * D;
    { const size_t n = (size_t)(cfg->max_positions + 2) * D; TRY(dev_upload(m->allocs, &m->embed_positions, p, n)); p += n; }
    TRY(dev_upload(m->allocs, &m->msa_pe, p, (size_t)1024 * D)); p += (size_t)1024 * D;
    TRY(dev_upload(m->allocs, &m->lnb_w, p, D)); p += D;
    TRY(dev_upload(m->allocs, &m->lnb_b, p, D)); p += D;
    const float qscale = 1.0f / sqrtf((float)kHeadDim);      // axial_attention.py:48,212 (exact 1/8); the tied
    m->layers.resize(cfg->layers);                           // rows' extra 1/sqrt(R) is applied to the scores
    std::vector<float> wq(3 * D * D), bq(3 * D);
    auto attn = [&](float** ln_w, float** ln_b, W16* wqkv, float** bqkv, W16* wo, float** bo) -> int {
        TRY(dev_upload(m->allocs, ln_w, p, D)); p += D;
        TRY(dev_upload(m->allocs, ln_b, p, D)); p += D;
        for (int k = 0; k < 3; ++k) {
            const float sc = (k == 0) ? qscale : 1.0f;
            for (size_t i = 0; i < D * D; ++i) wq[k * D * D + i] = p[i] * sc;
            p += D * D;
            for (size_t i = 0; i < D; ++i) bq[k * D + i] = p[i] * sc;
            p += D;
        }
        TRY(make_w16(m->allocs, wq.data(), wq.size(), D, cfg->precision, m->stream, wqkv));
        TRY(dev_upload(m->allocs, bqkv, bq.data(), bq.size()));
        TRY(make_w16(m->allocs, p, D * D, D, cfg->precision, m->stream, wo)); p += D * D;
        TRY(dev_upload(m->allocs, bo, p, D)); p += D;
        return PGMI_OK;
    };
    for (int l = 0; l < cfg->layers; ++l) {
        Layer& L = m->layers[l];
        TRY(attn(&L.ln1_w, &L.ln1_b, &L.wqkv16, &L.bqkv, &L.wo16, &L.bo));
        TRY(attn(&L.c_ln_w, &L.c_ln_b, &L.c_wqkv16, &L.c_bqkv, &L.c_wo16, &L.c_bo));
        TRY(dev_upload(m->allocs, &L.ln2_w, p, D)); p += D;
        TRY(dev_upload(m->allocs, &L.ln2_b, p, D)); p += D;
        TRY(make_w16(m->allocs, p, F * D, D, cfg->precision, m->stream, &L.w116)); p += F * D;
        TRY(dev_upload(m->allocs, &L.b1, p, F)); p += F;
        TRY(make_w16(m->allocs, p, D * F, F, cfg->precision, m->stream, &L.w216)); p += D * F;
        TRY(dev_upload(m->allocs, &L.b2, p, D)); p += D;
    }
    TRY(dev_upload(m->allocs, &m->lna_w, p, D)); p += D;
    TRY(dev_upload(m->allocs, &m->lna_b, p, D)); p += D;
    TRY(make_w16(m->allocs, p, D * D, D, cfg->precision, m->stream, &m->hd16)); p += D * D;
    TRY(dev_upload(m->allocs, &m->hd_b, p, D)); p += D;
    TRY(dev_upload(m->allocs, &m->hln_w, p, D)); p += D;
    TRY(dev_upload(m->allocs, &m->hln_b, p, D)); p += D;
    TRY(dev_upload(m->allocs, &m->h_bias, p, V)); p += V;
#undef TRY
    if (p - w != n_weights) { set_error("internal: blob walk mismatch"); return PGMI_EINVAL; }
    return PGMI_OK;
}

template <typename T>
static int ensure_cap(pgmi_model* m, T** p, size_t* cap, size_t need) {
    if (need <= *cap) return PGMI_OK;
    // grown buffers are owned by the model's pool; the old one stays in the pool until destroy (shapes
    // change rarely: once per alignment)
    T* q = nullptr;
    int rc = dev_alloc(m->allocs, &q, need);
    if (rc) return rc;
    *p = q;
    *cap = need;
    return PGMI_OK;
}

// MSA Transformer forward on the token grid in m->tokens [R, C] (one alignment); leaves the residual
// stream (row-major token order) in m->x.  msa_transformer.py:146-205.
// keep_col >= 0: the caller reads only token (row 0, column keep_col) of the output (masked-marginals: compute_fitness.py:418-423
// `token_probs[:, 0, i]`).  In the LAST layer everything after the tied row attention's value update is then needed for the R
// tokens of that column only -- out-projection of the row attention, the whole column attention (one column = one sequence of R
// rows) -- and the feed-forward for the single token (0, keep_col); the same kernels on the same rows, so the kept token is
// bit-identical to the full evaluation (tests/test_gpu_msa_transformer.py).  m->x row 0 then holds that token (*compacted = true).
int run_msa(pgmi_model* m, int R, int C, int keep_col = -1, bool* compacted = nullptr) {
    const pgmi_config& c = m->cfg;
    const int M = R * C, D = c.embed_dim, F = c.ffn_dim, H = c.heads;
    hipStream_t s = m->stream;
    if (R > 1024) { set_error("Using model with MSA position embedding trained on maximum MSA depth of 1024, but received %d alignments.", R); return PGMI_EINVAL; }
    if (C > c.max_positions) { set_error("Sequence length %d above maximum sequence length of %d", C, c.max_positions); return PGMI_EINVAL; }
    const int Rp = (R + 31) / 32 * 32, Cp = (C + 31) / 32 * 32;
    if ((int64_t)Rp * Cp > m->max_rows) { set_error("alignment of %d x %d tokens exceeds the workspace (%d rows): create the model with max_rows >= %lld", R, C, m->max_rows, (long long)Rp * Cp); return PGMI_EINVAL; }
    int rc = 0;
    // split the (r, d) contraction of the tied scores so that the launch fills the chip: S divides R, ~2 rounds of tiles at most
    const int Kp = (C + 63) / 64 * 64;                      // the update GEMM's K (columns j), zero-padded
    int S = 1;
    { const int rows_last = C % 256, tm = (rows_last > 0 && rows_last <= 128) ? (C + 127) / 128 : (C + 255) / 256;
      const int tiles = tm * ((C + 255) / 256) * H;
      for (int cand = 1; cand <= 16; ++cand) if (R % cand == 0 && tiles * cand <= 640) S = cand; }
    rc = ensure_cap(m, &m->tied_part, &m->tied_part_cap, (size_t)H * S * C * Kp); if (rc) return rc;
    rc = ensure_cap(m, &m->tied_p, &m->tied_p_cap, (size_t)H * C * Kp); if (rc) return rc;           // split planes: 4 bytes per element like fp32
    rc = ensure_cap(m, &m->tied_vt, &m->tied_vt_cap, (size_t)H * R * 64 * Kp); if (rc) return rc;
    if ((unsigned long long)M * D * 4ull >= (1ull << 32) || (unsigned long long)H * R * 64 * Kp * 4ull >= (1ull << 32)) {
        set_error("alignment of %d x %d tokens exceeds the 32-bit offset range of the tied row attention's operands", R, C);
        return PGMI_EINVAL;
    }
    if (R != m->msa_kv_R || C != m->msa_kv_C) {
        std::vector<int32_t> kv((size_t)C, R);
        PGMI_HIP(hipMemcpyAsync(m->msa_kv_len, kv.data(), (size_t)C * 4, hipMemcpyHostToDevice, s));
        PGMI_HIP(hipStreamSynchronize(s));
        // pad keys of the column attention (rows >= R inside the last 32-key tile) must hold finite data
        PGMI_HIP(hipMemsetAsync(m->vt16, 0, m->vt16_plane * 2 * sizeof(unsigned short), s));
        m->msa_kv_R = R; m->msa_kv_C = C;
        m->last_B = C; m->last_T = R;
    }
    { ProfScope p(m, PGMI_K_EMBED, 0, (double)M * D * 4);
      launch_seq_stats(m->tokens, R, C, 0, m->denom, m->pos_idx, m->kv_len, s);
      launch_embed(m->tokens, m->denom, m->pos_idx, m->embed_tokens, m->embed_positions, 0, M, C, D, m->x, s);
      launch_add_row_embedding(m->x, m->msa_pe, R, C, D, s);
      launch_layernorm(m->x, m->lnb_w, m->lnb_b, M, D, 1e-5f, m->x, s); }
    const double ln_bytes = 2.0 * M * D * 4;
    for (int l = 0; l < c.layers; ++l) {
        const Layer& L = m->layers[l];
        // ---- tied row attention (axial_attention.py:108-168) ----
        { ProfScope p(m, PGMI_K_LAYERNORM, 0, ln_bytes);
          launch_layernorm16(m->x, L.ln1_w, L.ln1_b, M, D, 1e-5f, m->h16, m->h16_plane, 1, s); }
        { ProfScope p(m, PGMI_K_GEMM_QKV, 2.0 * M * 3 * D * D, 0);
          rc = linear(m, nullptr, m->h16, m->h16_plane, nullptr, L.wqkv16, L.bqkv, nullptr, m->qkv, nullptr, 0, M, 3 * D, D, EPI_NONE);
          if (rc) return rc; }
        { ProfScope p(m, PGMI_K_ATTENTION, 4.0 * (double)C * C * R * D, 0);
          // operands for the 16-bit pipe (msa_transformer.hip): q -> m->h16 (free once the projection has read it), k -> m->g16, V^T
          unsigned short* q16 = m->h16;
          unsigned short* k16 = m->g16;
          unsigned short* vt16 = reinterpret_cast<unsigned short*>(m->tied_vt);
          unsigned short* p16 = reinterpret_cast<unsigned short*>(m->tied_p);
          launch_tied_prep_qk(m->qkv, M, D, q16, k16, s);
          launch_pack_vt16(m->qkv, R, C, Kp, H, vt16, s);
          XMap g1{};                                       // scores: batch = (head, K split); K walks (r, d): runs of 64 d a row of tokens apart
          g1.a_row_bytes = g1.w_row_bytes = (unsigned int)D * 4u;
          g1.a_bytes = g1.w_bytes = (unsigned int)((size_t)M * D * 4);
          g1.k_run_log2 = 1; g1.a_run_bytes = g1.w_run_bytes = (unsigned int)((size_t)C * D * 4);
          g1.batch_inner = S;
          g1.a_b0 = g1.w_b0 = (unsigned int)((size_t)(R / S) * C * D * 4); g1.a_b1 = g1.w_b1 = 64u * 4u;
          g1.c_b0 = (long long)C * Kp; g1.c_b1 = (long long)S * C * Kp; g1.ldc = Kp;
          rc = launch_gemm16_ex(q16, k16, m->tied_part, nullptr, C, (C + 3) / 4 * 4, (R / S) * 64, 1.0f / tied_w_scale(), g1, H * S, s);
          if (rc) return rc;
          rc = launch_tied_softmax16(m->tied_part, H, S, C, Kp, 1.0f / sqrtf((float)R), p16, s);
          if (rc) return rc;
          XMap g2{};                                       // update: batch = head, output scattered to the context rows [r, i] x columns [h, d]
          g2.a_row_bytes = g2.w_row_bytes = (unsigned int)Kp * 4u;
          g2.a_bytes = (unsigned int)((size_t)H * C * Kp * 4); g2.w_bytes = (unsigned int)((size_t)H * R * 64 * Kp * 4);
          g2.batch_inner = H;
          g2.a_b0 = (unsigned int)((size_t)C * Kp * 4); g2.w_b0 = (unsigned int)((size_t)R * 64 * Kp * 4);
          g2.o_ld = D; g2.o_rows_per_n64 = C; g2.o_col_per_batch = 64;
          rc = launch_gemm16_ex(p16, vt16, nullptr, m->h16, C, R * 64, Kp, 1.0f / tied_w_scale(), g2, H, s);
          if (rc) return rc; }
        if (keep_col >= 0 && m->keep_rows && l == c.layers - 1) {
            ProfScope p(m, PGMI_K_KEPT_ROWS, 2.0 * R * D * (2.0 * D + 3.0 * D) + 4.0 * R * R * D + 4.0 * D * F, 0);
            // the column's tokens (r, keep_col), r = 0 .. R-1: rows r * C + keep_col of the (r, c) order
            launch_strided_index(keep_col, C, R, m->row_idx, s);
            launch_gather_rows(m->x, m->row_idx, R, D, m->qkv, s);                                     // residual rows
            launch_gather_rows(reinterpret_cast<const float*>(m->h16), m->row_idx, R, D, reinterpret_cast<float*>(m->g16), s);   // context rows (K-interleaved: 4 D bytes)
            rc = linear(m, nullptr, m->g16, m->g16_plane, nullptr, L.wo16, L.bo, m->qkv, m->xt, nullptr, 0, R, D, D, EPI_NONE);
            if (rc) return rc;
            // column attention of this one column: a sequence of R rows (m->xt rows 0 .. R-1)
            launch_layernorm16(m->xt, L.c_ln_w, L.c_ln_b, R, D, 1e-5f, m->h16, m->h16_plane, 1, s);
            rc = launch_gemm16_qkv(m->h16, m->h16_plane, L.c_wqkv16.p, L.c_wqkv16.plane, L.c_bqkv, R, D, D, L.c_wqkv16.out_scale,
                                   m->qk16, m->qk16_plane, m->vt16, m->vt16_plane, nullptr, nullptr, 0, R, H, m->gemm_variant, s);
            if (rc) return rc;
            rc = launch_attention_f16x3_v2(nullptr, m->msa_kv_len, nullptr, nullptr, 0, 1, R, H, m->qk16, m->qk16_plane, m->vt16,
                                           m->vt16_plane, nullptr, m->h16, m->h16_plane, 1, s);
            if (rc) return rc;
            rc = linear(m, nullptr, m->h16, m->h16_plane, nullptr, L.c_wo16, L.c_bo, m->xt, m->xt, nullptr, 0, R, D, D, EPI_NONE);
            if (rc) return rc;
            // feed-forward for token (0, keep_col) = row 0 of the column
            launch_layernorm16(m->xt, L.ln2_w, L.ln2_b, 1, D, 1e-5f, m->h16, m->h16_plane, 1, s);
            rc = linear(m, nullptr, m->h16, m->h16_plane, nullptr, L.w116, L.b1, nullptr, nullptr, m->g16, m->g16_plane, 1, F, D, EPI_GELU);
            if (rc) return rc;
            rc = linear(m, nullptr, m->g16, m->g16_plane, nullptr, L.w216, L.b2, m->xt, m->x, nullptr, 0, 1, D, F, EPI_NONE);
            if (rc) return rc;
            if (compacted) *compacted = true;
            break;
        }
        { ProfScope p(m, PGMI_K_GEMM_OUT, 2.0 * M * D * D, 0);
          rc = linear(m, nullptr, m->h16, m->h16_plane, nullptr, L.wo16, L.bo, m->x, m->x, nullptr, 0, M, D, D, EPI_NONE);
          if (rc) return rc; }
        // ---- column attention (axial_attention.py:232-275): ordinary attention over the R rows of a column ----
        { ProfScope p(m, PGMI_K_LAYERNORM, 0, 2 * ln_bytes);
          launch_permute_rows(m->x, m->xt, R, C, D, s);                       // -> token order (c, r)
          launch_layernorm16(m->xt, L.c_ln_w, L.c_ln_b, M, D, 1e-5f, m->h16, m->h16_plane, 1, s); }
        { ProfScope p(m, PGMI_K_GEMM_QKV, 2.0 * M * 3 * D * D, 0);
          rc = launch_gemm16_qkv(m->h16, m->h16_plane, L.c_wqkv16.p, L.c_wqkv16.plane, L.c_bqkv, M, D, D, L.c_wqkv16.out_scale,
                                 m->qk16, m->qk16_plane, m->vt16, m->vt16_plane, nullptr, nullptr, 0, R, H, m->gemm_variant, s);
          if (rc) return rc; }
        { ProfScope p(m, PGMI_K_ATTENTION, 4.0 * (double)M * R * D, 0);
          rc = launch_attention_f16x3_v2(nullptr, m->msa_kv_len, nullptr, nullptr, 0, C, R, H, m->qk16, m->qk16_plane, m->vt16,
                                         m->vt16_plane, nullptr, m->h16, m->h16_plane, 1, s);
          if (rc) return rc; }
        { ProfScope p(m, PGMI_K_GEMM_OUT, 2.0 * M * D * D, 0);
          rc = linear(m, nullptr, m->h16, m->h16_plane, nullptr, L.c_wo16, L.c_bo, m->xt, m->xt, nullptr, 0, M, D, D, EPI_NONE);
          if (rc) return rc; }
        { ProfScope p(m, PGMI_K_LAYERNORM, 0, 2 * ln_bytes);
          launch_permute_rows(m->xt, m->x, C, R, D, s);                       // back to (r, c)
          launch_layernorm16(m->x, L.ln2_w, L.ln2_b, M, D, 1e-5f, m->h16, m->h16_plane, 1, s); }
        // ---- feed forward (modules.py:409-432) ----
        { ProfScope p(m, PGMI_K_GEMM_FC1, 2.0 * M * F * D, 0);
          rc = linear(m, nullptr, m->h16, m->h16_plane, nullptr, L.w116, L.b1, nullptr, nullptr, m->g16, m->g16_plane, M, F, D, EPI_GELU);
          if (rc) return rc; }
        { ProfScope p(m, PGMI_K_GEMM_FC2, 2.0 * M * F * D, 0);
          rc = linear(m, nullptr, m->g16, m->g16_plane, nullptr, L.w216, L.b2, m->x, m->x, nullptr, 0, M, D, F, EPI_NONE);
          if (rc) return rc; }
    }
    PGMI_HIP(hipGetLastError());
    return PGMI_OK;
}

// ---- Tranception: prefix-shared scoring ------------------------------------------------------------
// The reference forwards every mutated sequence in full, in both reading directions (scoring_utils.py:77-150, model_pytorch.py:878-928).
// The model is causal (attention model_pytorch.py:155-183; depth-wise convolution :73-88): every hidden state of a sequence before its
// first token that differs from the wild type IS the wild type's.  One chunk of work = ROOT sequences forwarded in full plus sequences
// that own only the rows from seq_p on (seq_p = the first token that differs from the root): LayerNorm, the four GEMMs of a layer and
// the head run on the packed suffix rows (row-local); the convolution takes its history -- and the head of seq_p's 32-token tile, which
// the attention wants whole -- from the root's input rows of the same launch, the attention its earlier key tiles from the root's
// operand planes, and the per-sequence reduction reads the root's log-probability rows before seq_p.
// Every row is computed by the same kernels from the same inputs in the same order as in a full forward: the same bits.
struct TrChunk {
    std::vector<int32_t> seq;                         // call-level index of every chunk-local sequence (a root may repeat over chunks)
    std::vector<int32_t> off, p, q, root;             // packed row of token p; first own token; operand row of its tile's first token; chunk-local root
    std::vector<uint32_t> vt;                         // V^T block offset (halfs per plane)
    std::vector<int32_t> tile_seq, tile_j, blk_seq, blk_j, tokens;
    int rows = 0, padded = 0;                         // packed rows; operand rows (every sequence from its tile on, rounded up to whole tiles)
    double att_flops = 0;
    int add(int call_index, const int32_t* tok, int T, int p0, int root_local, int D) {
        const int qrows = 32 * att16_waves_per_block(T);  // queries per attention block
        const int local = (int)seq.size(), a0 = p0 / 32 * 32, n = T - a0;
        seq.push_back(call_index);
        off.push_back(rows);
        p.push_back(p0);
        q.push_back(padded);
        root.push_back(root_local < 0 ? local : root_local);
        vt.push_back((uint32_t)((size_t)padded * (size_t)D));
        for (int j = 0; j < (n + 31) / 32; ++j) { tile_seq.push_back(local); tile_j.push_back(j); }
        for (int j = 0; j < (n + qrows - 1) / qrows; ++j) { blk_seq.push_back(local); blk_j.push_back(j); }
        tokens.insert(tokens.end(), tok + p0, tok + T);
        rows += T - p0;
        padded += (n + 31) / 32 * 32;
        att_flops += 2.0 * D * ((double)T * T - (double)p0 * p0);         // 4 D per (query, visible key) pair
        return local;
    }
};

// Runs one chunk: tokens (packed), index arrays and the retrieval arguments are uploaded, the forward leaves the suffix rows'
// log-probabilities in m->lp [rows, V] and the per-sequence reductions in m->denom [sequences].
int run_tranception_shared(pgmi_model* m, TrChunk& ck, int T, const float* prior_dev, const int32_t* a0, const int32_t* r0,
                           const int32_t* pn, const int32_t* fl, float alpha) {
    const pgmi_config& c = m->cfg;
    const int M = ck.rows, D = c.embed_dim, F = c.ffn_dim, H = c.heads, V = c.vocab, S = (int)ck.seq.size();
    hipStream_t s = m->stream;
    // attention blocks with the most key tiles first: the launch's tail is made of the short ones
    {
        std::vector<int> order(ck.blk_seq.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        const int qrows = 32 * att16_waves_per_block(T);
        auto keys = [&](int i) { return std::min(T, ck.p[ck.blk_seq[i]] / 32 * 32 + (ck.blk_j[i] + 1) * qrows); };
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return keys(x) > keys(y); });
        std::vector<int32_t> bs(order.size()), bj(order.size());
        for (size_t i = 0; i < order.size(); ++i) { bs[i] = ck.blk_seq[order[i]]; bj[i] = ck.blk_j[order[i]]; }
        ck.blk_seq.swap(bs);
        ck.blk_j.swap(bj);
    }
    const size_t nt = ck.tile_seq.size(), nb = ck.blk_seq.size();
    const size_t need = (size_t)9 * S + 2 * nt + 2 * nb;
    int rc = ensure_cap(m, &m->tr_meta, &m->tr_meta_cap, std::max(need, (size_t)1 << 16));
    if (rc) return rc;
    std::vector<int32_t> meta(need);
    int32_t* p = meta.data();
    auto put = [&](const void* src, size_t n) { memcpy(p, src, n * 4); p += n; return (int32_t*)(m->tr_meta + (p - n - meta.data())); };
    std::vector<int32_t> pa(S, 0), pr(S, 0), pc(S, 0), pf(S, 0);
    if (prior_dev)
        for (int i = 0; i < S; ++i) { pa[i] = a0[ck.seq[i]]; pr[i] = r0[ck.seq[i]]; pc[i] = pn[ck.seq[i]]; pf[i] = fl[ck.seq[i]]; }
    AttRagged rg{};
    rg.seq_off = put(ck.off.data(), S);
    rg.seq_p = put(ck.p.data(), S);
    rg.seq_q = put(ck.q.data(), S);
    rg.seq_root = put(ck.root.data(), S);
    rg.seq_vt = reinterpret_cast<const uint32_t*>(put(ck.vt.data(), S));
    const int32_t* d_pa = put(pa.data(), S);
    const int32_t* d_pr = put(pr.data(), S);
    const int32_t* d_pc = put(pc.data(), S);
    const int32_t* d_pf = put(pf.data(), S);
    rg.tile_seq = put(ck.tile_seq.data(), nt);
    rg.tile_j = put(ck.tile_j.data(), nt);
    rg.blk_seq = put(ck.blk_seq.data(), nb);
    rg.blk_j = put(ck.blk_j.data(), nb);
    rg.n_tiles = (int)nt;
    rg.n_blocks = (int)nb;
    PGMI_HIP(hipMemcpyAsync(m->tr_meta, meta.data(), need * 4, hipMemcpyHostToDevice, s));
    PGMI_HIP(hipMemcpyAsync(m->tokens, ck.tokens.data(), (size_t)M * 4, hipMemcpyHostToDevice, s));
    PGMI_HIP(hipStreamSynchronize(s));                 // the host vectors go out of scope with the caller's chunk
    m->last_B = m->last_T = 0;                          // the V^T planes now hold another layout: the dense path clears them again
    { ProfScope ps(m, PGMI_K_EMBED, 0, (double)M * D * 4);
      launch_gather_rows(m->embed_tokens, m->tokens, M, D, m->x, s); }
    const double ln_bytes = 2.0 * M * D * 4;
    for (int l = 0; l < c.layers; ++l) {
        const Layer& L = m->layers[l];
        { ProfScope ps(m, PGMI_K_LAYERNORM, 0, ln_bytes);
          launch_layernorm16(m->x, L.ln1_w, L.ln1_b, M, D, m->ln_eps, m->h16, m->h16_plane, 1, s); }
        { ProfScope ps(m, PGMI_K_GEMM_QKV, 2.0 * M * 3 * D * D, 0);
          rc = linear(m, nullptr, m->h16, m->h16_plane, nullptr, L.wqkv16, L.bqkv, nullptr, m->qkv, nullptr, 0, M, 3 * D, D, EPI_NONE);
          if (rc) return rc; }
        { ProfScope ps(m, PGMI_K_ATTENTION, ck.att_flops, 0);
          rc = launch_attention_tr_ragged(m->qkv, L.conv, m->tr_slopes, T, H, rg, m->qk16, m->qk16_plane, m->vt16, m->vt16_plane,
                                          m->h16, m->h16_plane, s);
          if (rc) return rc; }
        { ProfScope ps(m, PGMI_K_GEMM_OUT, 2.0 * M * D * D, 0);
          rc = linear(m, nullptr, m->h16, m->h16_plane, nullptr, L.wo16, L.bo, m->x, m->x, nullptr, 0, M, D, D, EPI_NONE);
          if (rc) return rc; }
        { ProfScope ps(m, PGMI_K_LAYERNORM, 0, ln_bytes);
          launch_layernorm16(m->x, L.ln2_w, L.ln2_b, M, D, m->ln_eps, m->h16, m->h16_plane, 1, s); }
        { ProfScope ps(m, PGMI_K_GEMM_FC1, 2.0 * M * F * D, 0);
          rc = linear(m, nullptr, m->h16, m->h16_plane, nullptr, L.w116, L.b1, nullptr, nullptr, m->g16, m->g16_plane, M, F, D, EPI_SQRELU);
          if (rc) return rc; }
        { ProfScope ps(m, PGMI_K_GEMM_FC2, 2.0 * M * F * D, 0);
          rc = linear(m, nullptr, m->g16, m->g16_plane, nullptr, L.w216, L.b2, m->x, m->x, nullptr, 0, M, D, F, EPI_NONE);
          if (rc) return rc; }
    }
    { ProfScope ps(m, PGMI_K_HEAD, 2.0 * M * D * V, 0);
      launch_layernorm(m->x, m->lna_w, m->lna_b, M, D, m->ln_eps, m->h, s);
      launch_vocab_logsoftmax(m->h, m->tr_lm_head, m->tr_zero_bias, M, D, V, m->lp, m->nonfinite, s); }
    { ProfScope ps(m, PGMI_K_SCORE, 0, (double)S * T * 8);
      launch_seq_loglik_ragged(m->lp, m->tokens, rg.seq_off, rg.seq_p, rg.seq_root, S, T, V, prior_dev, d_pa, d_pr, d_pc, d_pf, alpha,
                               m->denom, s); }
    PGMI_HIP(hipGetLastError());
    return PGMI_OK;
}

// fp16 range check for the 16-bit modes: the vocabulary kernel raises the flag when a computed
// log-probability is NaN/inf (an activation exceeded fp16's 65504 upstream).
int check_nonfinite(pgmi_model* m) {
    if (m->cfg.precision == PGMI_PREC_FP32) return PGMI_OK;
    int32_t flag = 0;
    PGMI_HIP(hipMemcpyAsync(&flag, m->nonfinite, 4, hipMemcpyDeviceToHost, m->stream));
    PGMI_HIP(hipStreamSynchronize(m->stream));
    if (flag) {
        PGMI_HIP(hipMemsetAsync(m->nonfinite, 0, 4, m->stream));
        set_error("non-finite log-probabilities: an activation left the fp16/bf16 range in precision mode %d; "
                  "re-run with precision fp32", m->cfg.precision);
        return PGMI_EOVERFLOW;
    }
    return PGMI_OK;
}

}  // namespace

extern "C" {

int pgmi_abi_version(void) { return PGMI_ABI_VERSION; }

const char* pgmi_last_error(void) { return g_err; }

int pgmi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int64_t pgmi_weight_count(const pgmi_config* c) {
    if (!c || c->layers <= 0 || c->embed_dim <= 0 || c->ffn_dim <= 0) return -1;
    const int64_t D = c->embed_dim, F = c->ffn_dim, V = c->vocab;
    if (c->arch == PGMI_ARCH_TRANCEPTION) {
        const int64_t conv = 3 * ((64 * 3 + 64) + (64 * 5 + 64) + (64 * 7 + 64));
        return V * D + (int64_t)c->layers * (2 * D + (D * 3 * D + 3 * D) + conv + (D * D + D) + 2 * D + (D * F + F) + (F * D + D)) + 2 * D + V * D;
    }
    if (c->arch == PGMI_ARCH_MSA) {
        const int64_t attn = 2 * D + 4 * (D * D + D);
        return V * D + (int64_t)(c->max_positions + 2) * D + 1024 * D + 2 * D +
               (int64_t)c->layers * (2 * attn + 2 * D + (F * D + F) + (D * F + D)) + 2 * D + (D * D + D) + 2 * D + V;
    }
    int64_t n = V * D;
    if (c->arch == PGMI_ARCH_ESM1B) n += (int64_t)(c->max_positions + 2) * D;
    if (c->emb_layer_norm_before) n += 2 * D;
    n += (int64_t)c->layers * (2 * D + 4 * (D * D + D) + 2 * D + (F * D + F) + (D * F + D));
    n += 2 * D + (D * D + D) + 2 * D + V;
    return n;
}

int pgmi_model_create(const pgmi_config* cfg, const float* w, int64_t n_weights, int device, pgmi_model** out) {
    if (!out) { set_error("null out"); return PGMI_EINVAL; }
    *out = nullptr;
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!w || n_weights != pgmi_weight_count(cfg)) {
        set_error("weight blob has %lld elements, config needs %lld", (long long)n_weights, (long long)pgmi_weight_count(cfg));
        return PGMI_EINVAL;
    }
    const int ndev = pgmi_device_count();
    if (ndev <= 0) { set_error("no HIP device visible (libpgmi has no CPU fallback)"); return PGMI_ENODEV; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d visible)", device, ndev); return PGMI_EINVAL; }
    PGMI_HIP(hipSetDevice(device));
    pgmi_model* m = new pgmi_model();
    m->cfg = *cfg;
    m->device = device;
#define TRY(e) do { rc = (e); if (rc) { pgmi_model_destroy(m); return rc; } } while (0)
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); delete m; return PGMI_EHIP; }
    const size_t D = cfg->embed_dim, F = cfg->ffn_dim, V = cfg->vocab;
    m->dh = cfg->embed_dim / cfg->heads;
    m->rot_halves = m->dh > kHeadDim ? 2 : 1;
    m->Hs = cfg->heads * m->rot_halves;
    m->Da = m->Hs * kHeadDim;
    m->ln_eps = cfg->ln_eps > 0.f ? cfg->ln_eps : 1e-5f;
    const float* p = w;
    if (cfg->arch == PGMI_ARCH_TRANCEPTION) {
        TRY(create_tranception(m, cfg, w, n_weights));
    } else if (cfg->arch == PGMI_ARCH_MSA) {
        TRY(create_msa(m, cfg, w, n_weights));
    } else {
    // embed_tokens == the tied lm_head.weight (esm1.py:101-105).  The host passes the matrix that
    // load_state_dict leaves in the tied parameter (pretrained.py:97,216), see proteingym_amd/esm.py.
    TRY(dev_upload(m->allocs, &m->embed_tokens, p, V * D));
    p += V * D;
    if (cfg->arch == PGMI_ARCH_ESM1B) {
        const size_t n = (size_t)(cfg->max_positions + 2) * D;
        TRY(dev_upload(m->allocs, &m->embed_positions, p, n));
        p += n;
    }
    if (cfg->emb_layer_norm_before) {
        TRY(dev_upload(m->allocs, &m->lnb_w, p, D)); p += D;
        TRY(dev_upload(m->allocs, &m->lnb_b, p, D)); p += D;
    }
    // head layout: every head owns 64 lanes of the attention kernels; dim j of a head sits in slot
    // j (first half) or 32 + (j - dh/2) (second half) so that rotary pairs (j, j + dh/2) are the
    // kernels' pairs (i, i + 32).  dh == 64 is the identity layout; smaller heads leave zero slots
    // (zero weight rows -> q,k,v slots exactly 0 -> scores and context unchanged).
    const size_t H = cfg->heads, dh = m->dh, Da = m->Da;
    // head_dim 128: a head is two slot groups; group g in {0,1} holds dims 32 g + i (slots i < 32) and 64 + 32 g + i (slots 32 + i), so
    // the rotary partners (j, j + 64) are again the kernels' pairs (i, i + 32) inside ONE 64-column wave tile of the QKV epilogue.
    auto slot = [&](size_t col) -> size_t {
        const size_t h = col / dh, j = col % dh;
        if (dh > 64) return (2 * h + ((j >> 5) & 1)) * 64 + ((j >> 6) << 5) + (j & 31);
        return h * 64 + (j < dh / 2 ? j : 32 + (j - dh / 2));
    };
    const float qscale = 1.0f / sqrtf((float)dh);           // multihead_attention.py:261 (exact 1/8 for dh 64)
    m->layers.resize(cfg->layers);
    std::vector<float> wq(3 * Da * D, 0.0f), bq(3 * Da, 0.0f), wo_r(D * Da, 0.0f);
    (void)H;
    for (int l = 0; l < cfg->layers; ++l) {
        Layer& L = m->layers[l];
        TRY(dev_upload(m->allocs, &L.ln1_w, p, D)); p += D;
        TRY(dev_upload(m->allocs, &L.ln1_b, p, D)); p += D;
        for (int k = 0; k < 3; ++k) {            // fused [3Da, D] projection, q rows pre-scaled
            const float sc = (k == 0) ? qscale : 1.0f;
            for (size_t o = 0; o < D; ++o) {
                float* dst = &wq[(k * Da + slot(o)) * D];
                for (size_t i = 0; i < D; ++i) dst[i] = p[o * D + i] * sc;
            }
            p += D * D;
            for (size_t o = 0; o < D; ++o) bq[k * Da + slot(o)] = p[o] * sc;
            p += D;
        }
        const bool f32w = cfg->precision == PGMI_PREC_FP32;
        if (f32w) TRY(dev_upload(m->allocs, &L.wqkv, wq.data(), wq.size()));
        else TRY(make_w16(m->allocs, wq.data(), wq.size(), D, cfg->precision, m->stream, &L.wqkv16));
        TRY(dev_upload(m->allocs, &L.bqkv, bq.data(), bq.size()));
        for (size_t o = 0; o < D; ++o)           // out-proj [D, Da]: input columns follow the slot layout
            for (size_t i = 0; i < D; ++i) wo_r[o * Da + slot(i)] = p[o * D + i];
        if (f32w) TRY(dev_upload(m->allocs, &L.wo, wo_r.data(), wo_r.size()));
        else TRY(make_w16(m->allocs, wo_r.data(), wo_r.size(), Da, cfg->precision, m->stream, &L.wo16));
        p += D * D;
        TRY(dev_upload(m->allocs, &L.bo, p, D)); p += D;
        TRY(dev_upload(m->allocs, &L.ln2_w, p, D)); p += D;
        TRY(dev_upload(m->allocs, &L.ln2_b, p, D)); p += D;
        if (f32w) TRY(dev_upload(m->allocs, &L.w1, p, F * D));
        else TRY(make_w16(m->allocs, p, F * D, D, cfg->precision, m->stream, &L.w116));
        p += F * D;
        TRY(dev_upload(m->allocs, &L.b1, p, F)); p += F;
        if (f32w) TRY(dev_upload(m->allocs, &L.w2, p, D * F));
        else TRY(make_w16(m->allocs, p, D * F, F, cfg->precision, m->stream, &L.w216));
        p += D * F;
        TRY(dev_upload(m->allocs, &L.b2, p, D)); p += D;
    }
    TRY(dev_upload(m->allocs, &m->lna_w, p, D)); p += D;
    TRY(dev_upload(m->allocs, &m->lna_b, p, D)); p += D;
    if (cfg->precision == PGMI_PREC_FP32) TRY(dev_upload(m->allocs, &m->hd_w, p, D * D));
    else TRY(make_w16(m->allocs, p, D * D, D, cfg->precision, m->stream, &m->hd16));
    p += D * D;
    TRY(dev_upload(m->allocs, &m->hd_b, p, D)); p += D;
    TRY(dev_upload(m->allocs, &m->hln_w, p, D)); p += D;
    TRY(dev_upload(m->allocs, &m->hln_b, p, D)); p += D;
    TRY(dev_upload(m->allocs, &m->h_bias, p, V)); p += V;
    if (p - w != n_weights) { set_error("internal: blob walk mismatch"); pgmi_model_destroy(m); return PGMI_EINVAL; }
    }

    m->max_rows = cfg->max_rows > 0 ? cfg->max_rows : 98304;
    if (m->max_rows < 2048) m->max_rows = 2048;
    {
    const size_t R = m->max_rows, Da = m->Da, Dw = std::max(D, Da);   // h / h16 hold LN output [.,D] and attention context [.,Da]
    TRY(dev_alloc(m->allocs, &m->x, R * D));
    TRY(dev_alloc(m->allocs, &m->h, R * Dw));
    TRY(dev_alloc(m->allocs, &m->qkv, R * 3 * Da));
    const bool f32mode = cfg->precision == PGMI_PREC_FP32;
    TRY(dev_alloc(m->allocs, &m->g, R * (f32mode ? std::max(F, D) : D)));
    if (!f32mode) {
        const size_t planes = cfg->precision == PGMI_PREC_F16X3 ? 2 : 1;
        m->h16_plane = R * Dw;
        m->g16_plane = R * F;
        TRY(dev_alloc(m->allocs, &m->h16, m->h16_plane * planes));
        TRY(dev_alloc(m->allocs, &m->g16, m->g16_plane * planes));
    }
    TRY(dev_alloc(m->allocs, &m->nonfinite, (size_t)1));
    PGMI_HIP(hipMemset(m->nonfinite, 0, 4));
    if (cfg->precision == PGMI_PREC_F16X3) {
        m->qk16_plane = R * 2 * Da;
        m->vt16_plane = R * Da;
        TRY(dev_alloc(m->allocs, &m->qk16, m->qk16_plane * 2));
        TRY(dev_alloc(m->allocs, &m->vt16, m->vt16_plane * 2));
        PGMI_HIP(hipMemset(m->vt16, 0, m->vt16_plane * 2 * sizeof(unsigned short)));
    }
    m->keep_rows = env_int("PGMI_KEEP_ROWS", 1);
    gemm_options_from_env();                             // the GEMM launchers' test hooks: read here, not per launch
    m->gemm_variant = env_int("PGMI_GEMM_VARIANT", 0);   // tuning only (gemm_f16.hip set_tune); below 1000 = the product configuration
    if (cfg->arch == PGMI_ARCH_MSA) {
        TRY(dev_alloc(m->allocs, &m->xt, R * D));
        TRY(dev_alloc(m->allocs, &m->msa_kv_len, (size_t)2048));
    }
    TRY(dev_alloc(m->allocs, &m->lp, R * V));
    TRY(dev_alloc(m->allocs, &m->denom, R));
    TRY(dev_alloc(m->allocs, &m->tokens, R));
    TRY(dev_alloc(m->allocs, &m->pos_idx, R));
    TRY(dev_alloc(m->allocs, &m->kv_len, R));
    TRY(dev_alloc(m->allocs, &m->row_idx, R));
    TRY(dev_alloc(m->allocs, &m->aux_i, R));
    }
#undef TRY
    *out = m;
    return PGMI_OK;
}

void pgmi_model_destroy(pgmi_model* m) {
    if (!m) return;
    hipSetDevice(m->device);
    if (m->stream) hipStreamSynchronize(m->stream);
    for (pgmi_assay* a : m->assays) {           // assays outliving their model become inert handles
        for (void* p : a->allocs) hipFree(p);
        a->allocs.clear();
        a->m = nullptr;
    }
    m->assays.clear();
    for (pgmi_pppl* q : m->pppls) {
        for (void* p : q->allocs) hipFree(p);
        q->allocs.clear();
        q->m = nullptr;
    }
    m->pppls.clear();
    for (auto& e : m->events) { hipEventDestroy(e.start); hipEventDestroy(e.stop); }
    for (void* p : m->allocs) hipFree(p);
    if (m->stream) hipStreamDestroy(m->stream);
    delete m;
}

int pgmi_model_device(const pgmi_model* m) { return m ? m->device : -1; }

int pgmi_set_option(const char* name, int64_t value) {
    if (!name) { set_error("null option name"); return PGMI_EINVAL; }
    int rc = gemm_set_option(name, (long long)value);
    if (rc) rc = att_set_option(name, (long long)value);
    if (rc) set_error("unknown option '%s' (gemm_half_tail, gemm_max_rows, att_xcd_local, att_v3)", name);
    return rc;
}

int pgmi_synchronize(pgmi_model* m) {
    if (!m) { set_error("null model"); return PGMI_EINVAL; }
    PGMI_HIP(hipStreamSynchronize(m->stream));
    return PGMI_OK;
}

int pgmi_token_logprobs(pgmi_model* m, const int32_t* tokens, int B, int T, float* out) {
    if (!m || !tokens || !out || B <= 0 || T <= 0) { set_error("bad argument"); return PGMI_EINVAL; }
    if (T + 31 > m->max_rows) { set_error("T=%d exceeds workspace rows %d", T, m->max_rows); return PGMI_EINVAL; }
    int rc = check_tokens(tokens, B, T);
    if (rc) return rc;
    PGMI_HIP(hipSetDevice(m->device));
    const int per = std::max(1, m->max_rows / ((T + 31) / 32 * 32));     // B * roundup(T,32) <= max_rows
    const int V = m->cfg.vocab;
    for (int b0 = 0; b0 < B; b0 += per) {
        const int bc = std::min(per, B - b0);
        PGMI_HIP(hipMemcpyAsync(m->tokens, tokens + (size_t)b0 * T, (size_t)bc * T * 4, hipMemcpyHostToDevice, m->stream));
        rc = run_encoder(m, bc, T);
        if (rc) return rc;
        rc = run_head(m, bc * T, nullptr);
        if (rc) return rc;
        PGMI_HIP(hipMemcpyAsync(out + (size_t)b0 * T * V, m->lp, (size_t)bc * T * V * 4, hipMemcpyDeviceToHost, m->stream));
        PGMI_HIP(hipStreamSynchronize(m->stream));
    }
    return check_nonfinite(m);
}

int pgmi_masked_logprobs(pgmi_model* m, const int32_t* tokens, const int32_t* mask_pos, int B, int T, float* out) {
    if (!m || !tokens || !mask_pos || !out || B <= 0 || T <= 0) { set_error("bad argument"); return PGMI_EINVAL; }
    if (T > m->max_rows) { set_error("T=%d exceeds workspace rows %d", T, m->max_rows); return PGMI_EINVAL; }
    int rc = check_tokens(tokens, B, T);
    if (rc) return rc;
    for (int b = 0; b < B; ++b)
        if (mask_pos[b] < 0 || mask_pos[b] >= T) { set_error("mask_pos[%d]=%d out of range", b, mask_pos[b]); return PGMI_EINVAL; }
    PGMI_HIP(hipSetDevice(m->device));
    const int per = std::max(1, m->max_rows / ((T + 31) / 32 * 32));
    const int V = m->cfg.vocab;
    std::vector<int32_t> ridx;
    for (int b0 = 0; b0 < B; b0 += per) {
        const int bc = std::min(per, B - b0);
        ridx.resize(bc);
        for (int b = 0; b < bc; ++b) ridx[b] = b * T + mask_pos[b0 + b];
        PGMI_HIP(hipMemcpyAsync(m->tokens, tokens + (size_t)b0 * T, (size_t)bc * T * 4, hipMemcpyHostToDevice, m->stream));
        PGMI_HIP(hipMemcpyAsync(m->aux_i, mask_pos + b0, (size_t)bc * 4, hipMemcpyHostToDevice, m->stream));
        PGMI_HIP(hipMemcpyAsync(m->row_idx, ridx.data(), (size_t)bc * 4, hipMemcpyHostToDevice, m->stream));
        launch_apply_mask(m->tokens, m->aux_i, bc, T, m->stream);
        rc = run_rows(m, bc, T, bc, m->row_idx);
        if (rc) return rc;
        PGMI_HIP(hipMemcpyAsync(out + (size_t)b0 * V, m->lp, (size_t)bc * V * 4, hipMemcpyDeviceToHost, m->stream));
        PGMI_HIP(hipStreamSynchronize(m->stream));
    }
    return check_nonfinite(m);
}


static int msa_upload(pgmi_model* m, const int32_t* tokens, int R, int T) {
    if (!m || m->cfg.arch != PGMI_ARCH_MSA) { set_error("model is not an MSA Transformer"); return PGMI_EINVAL; }
    if (!tokens || R <= 0 || T <= 0) { set_error("bad argument"); return PGMI_EINVAL; }
    for (int64_t i = 0; i < (int64_t)R * T; ++i) {
        if (tokens[i] < 0 || tokens[i] >= PGMI_VOCAB) { set_error("token id %d out of range at [%lld,%lld]", tokens[i], (long long)(i / T), (long long)(i % T)); return PGMI_EINVAL; }
        if (tokens[i] == PGMI_TOK_PAD) { set_error("<pad> at [%lld,%lld]: the rows of an alignment must have equal length", (long long)(i / T), (long long)(i % T)); return PGMI_EINVAL; }
    }
    PGMI_HIP(hipSetDevice(m->device));
    int rc = ensure_cap(m, &m->msa_full, &m->msa_full_cap, (size_t)R * T);
    if (rc) return rc;
    PGMI_HIP(hipMemcpyAsync(m->msa_full, tokens, (size_t)R * T * 4, hipMemcpyHostToDevice, m->stream));
    return PGMI_OK;
}

int pgmi_msa_token_logprobs(pgmi_model* m, const int32_t* tokens, int R, int T, float* out) {
    if (!out) { set_error("bad argument"); return PGMI_EINVAL; }
    int rc = msa_upload(m, tokens, R, T);
    if (rc) return rc;
    launch_msa_window_tokens(m->msa_full, R, T, 0, T, -1, m->tokens, m->stream);
    rc = run_msa(m, R, T);
    if (rc) return rc;
    rc = run_head(m, R * T, nullptr);
    if (rc) return rc;
    PGMI_HIP(hipMemcpyAsync(out, m->lp, (size_t)R * T * m->cfg.vocab * 4, hipMemcpyDeviceToHost, m->stream));
    PGMI_HIP(hipStreamSynchronize(m->stream));
    return check_nonfinite(m);
}

int pgmi_msa_masked_logprobs(pgmi_model* m, const int32_t* tokens, int R, int T, int window, const int32_t* positions,
                             const int32_t* starts, int n, float* out) {
    if (!positions || !starts || !out || n < 0 || window <= 0) { set_error("bad argument"); return PGMI_EINVAL; }
    int rc = msa_upload(m, tokens, R, T);
    if (rc) return rc;
    const int V = m->cfg.vocab;
    for (int i = 0; i < n; ++i) {
        const int st = starts[i], pos = positions[i];
        const int Tw = std::min(window, T - st);             // python slicing [:, :, start:end] truncates at the end
        if (st < 0 || st >= T || pos < st || pos >= st + Tw) { set_error("position %d outside its window [%d,%d)", pos, st, st + Tw); return PGMI_EINVAL; }
        launch_msa_window_tokens(m->msa_full, R, T, st, Tw, pos, m->tokens, m->stream);
        const int32_t row = pos - st;                        // row 0 of the grid, column pos - start
        bool compacted = false;
        rc = run_msa(m, R, Tw, row, &compacted);
        if (rc) return rc;
        if (!compacted) PGMI_HIP(hipMemcpyAsync(m->row_idx, &row, 4, hipMemcpyHostToDevice, m->stream));
        rc = run_head(m, 1, compacted ? nullptr : m->row_idx);
        if (rc) return rc;
        PGMI_HIP(hipMemcpyAsync(out + (size_t)i * V, m->lp, (size_t)V * 4, hipMemcpyDeviceToHost, m->stream));
        PGMI_HIP(hipStreamSynchronize(m->stream));
    }
    return check_nonfinite(m);
}

int pgmi_score_mutants(const float* table, int n_rows, int vocab, const int32_t* sub_pos, const int32_t* sub_wt, const int32_t* sub_mt,
                       const int64_t* mut_off, int64_t n_mut, double* scores) {
    // compute_fitness.py:240-250 on the host: the arithmetic of score_mutants_kernel (elementwise.hip)
    if (!table || !mut_off || !scores || n_rows <= 0 || vocab <= 0 || n_mut < 0) { set_error("bad argument"); return PGMI_EINVAL; }
    const int64_t n_sub = mut_off[n_mut];
    if (n_sub > 0 && (!sub_pos || !sub_wt || !sub_mt)) { set_error("bad argument"); return PGMI_EINVAL; }
    for (int64_t k = 0; k < n_sub; ++k)
        if (sub_pos[k] < 0 || sub_pos[k] >= n_rows || sub_wt[k] < 0 || sub_wt[k] >= vocab || sub_mt[k] < 0 || sub_mt[k] >= vocab) {
            set_error("substitution %lld reads table[%d][%d / %d] of a [%d][%d] table", (long long)k, sub_pos[k], sub_wt[k], sub_mt[k], n_rows, vocab);
            return PGMI_EINVAL;
        }
    for (int64_t i = 0; i < n_mut; ++i) {
        double sc = 0.0;
        for (int64_t k = mut_off[i]; k < mut_off[i + 1]; ++k) {
            const float* rowp = table + (size_t)sub_pos[k] * vocab;
            const float d = rowp[sub_mt[k]] - rowp[sub_wt[k]];
            sc += (double)d;
        }
        scores[i] = sc;
    }
    return PGMI_OK;
}

void pgmi_optimal_window(int position, int n, int window, int* start, int* end) {
    // proteingym/utils/scoring_utils.py:43-52
    const int half = window / 2;
    int s, e;
    if (n <= window) { s = 0; e = n; }
    else if (position < half) { s = 0; e = window; }
    else if (position >= n - half) { s = n - window; e = n; }
    else { s = std::max(0, position - half); e = std::min(n, position + half); }
    if (start) *start = s;
    if (end) *end = e;
}

int pgmi_assay_create(pgmi_model* m, const int32_t* wt_tokens, int n_tok, const int32_t* positions, int P,
                      int window, const int32_t* sub_pos, const int32_t* sub_wt, const int32_t* sub_mt,
                      const int64_t* mut_off, int64_t n_mut, pgmi_assay** out) {
    if (!out) { set_error("null out"); return PGMI_EINVAL; }
    *out = nullptr;
    if (!m || !wt_tokens || n_tok <= 0 || P < 0 || (P > 0 && !positions) || window <= 0 || n_mut < 0) { set_error("bad argument"); return PGMI_EINVAL; }
    for (int i = 0; i < n_tok; ++i)
        if (wt_tokens[i] < 0 || wt_tokens[i] >= PGMI_VOCAB || wt_tokens[i] == PGMI_TOK_PAD) { set_error("wt token %d invalid at %d", wt_tokens[i], i); return PGMI_EINVAL; }
    const int T = std::min(n_tok, window);
    if (T > m->max_rows) { set_error("T=%d exceeds workspace rows %d", T, m->max_rows); return PGMI_EINVAL; }
    std::vector<int32_t> ws(P), mr(P);
    for (int i = 0; i < P; ++i) {
        if (positions[i] < 0 || positions[i] >= n_tok) { set_error("position %d out of range", positions[i]); return PGMI_EINVAL; }
        int s, e;
        pgmi_optimal_window(positions[i], n_tok, window, &s, &e);
        if (e - s != T) { set_error("internal: window length %d != %d", e - s, T); return PGMI_EINVAL; }
        ws[i] = s;
        mr[i] = positions[i] - s;
    }
    const int64_t n_sub = n_mut ? mut_off[n_mut] : 0;
    for (int64_t k = 0; k < n_sub; ++k)
        if (sub_pos[k] < 0 || sub_pos[k] >= n_tok || sub_wt[k] < 0 || sub_wt[k] >= PGMI_VOCAB || sub_mt[k] < 0 || sub_mt[k] >= PGMI_VOCAB) {
            set_error("substitution %lld out of range", (long long)k);
            return PGMI_EINVAL;
        }
    PGMI_HIP(hipSetDevice(m->device));
    pgmi_assay* a = new pgmi_assay();
    a->m = m; a->n_tok = n_tok; a->P = P; a->T = T; a->n_mut = n_mut; a->n_sub = n_sub;
    int rc;
#define TRY(e) do { rc = (e); if (rc) { pgmi_assay_destroy(a); return rc; } } while (0)
    TRY(dev_upload(a->allocs, &a->wt, wt_tokens, (size_t)n_tok));
    TRY(dev_upload(a->allocs, &a->positions, positions, (size_t)P));
    TRY(dev_upload(a->allocs, &a->win_start, ws.data(), (size_t)P));
    TRY(dev_upload(a->allocs, &a->mask_rel, mr.data(), (size_t)P));
    TRY(dev_upload(a->allocs, &a->sub_pos, sub_pos, (size_t)n_sub));
    TRY(dev_upload(a->allocs, &a->sub_wt, sub_wt, (size_t)n_sub));
    TRY(dev_upload(a->allocs, &a->sub_mt, sub_mt, (size_t)n_sub));
    TRY(dev_upload(a->allocs, &a->mut_off, mut_off, (size_t)(n_mut + 1)));
    TRY(dev_alloc(a->allocs, &a->table, (size_t)n_tok * PGMI_VOCAB));
    TRY(dev_alloc(a->allocs, &a->scores, (size_t)n_mut));
#undef TRY
    m->assays.push_back(a);
    *out = a;
    return PGMI_OK;
}

void pgmi_assay_destroy(pgmi_assay* a) {
    if (!a) return;
    if (a->m) {
        hipSetDevice(a->m->device);
        hipStreamSynchronize(a->m->stream);
        auto& v = a->m->assays;
        v.erase(std::remove(v.begin(), v.end(), a), v.end());
    }
    for (void* p : a->allocs) hipFree(p);
    delete a;
}

int pgmi_assay_run(pgmi_model* m, pgmi_assay* a, double* scores_host, float* table_host, double* scores_dev) {
    if (!m || !a || a->m != m) { set_error("bad model/assay handle (assay belongs to another or a destroyed model)"); return PGMI_EINVAL; }
    PGMI_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const int T = a->T, V = m->cfg.vocab;
    const int per = std::max(1, m->max_rows / ((T + 31) / 32 * 32));
    launch_fill_f32(a->table, (int64_t)a->n_tok * V, NAN, s);
    for (int p0 = 0; p0 < a->P; p0 += per) {
        const int bc = std::min(per, a->P - p0);
        launch_make_masked_windows(a->wt, a->win_start + p0, a->mask_rel + p0, bc, T, m->tokens, s);
        // rows to keep: b*T + mask_rel[b]  (compute_fitness.py:503: token_probs[:, i-start])
        launch_row_index(a->mask_rel + p0, bc, T, m->row_idx, s);
        int rc = run_rows(m, bc, T, bc, m->row_idx);
        if (rc) return rc;
        launch_scatter_rows(m->lp, a->positions + p0, bc, V, a->table, s);
    }
    {
        ProfScope p(m, PGMI_K_SCORE, 0, (double)a->n_sub * 20);
        launch_score_mutants(a->table, V, a->sub_pos, a->sub_wt, a->sub_mt, a->mut_off, a->n_mut, a->scores, s);
    }
    PGMI_HIP(hipGetLastError());
    if (scores_dev && a->n_mut) PGMI_HIP(hipMemcpyAsync(scores_dev, a->scores, (size_t)a->n_mut * 8, hipMemcpyDeviceToDevice, s));
    if (scores_host && a->n_mut) PGMI_HIP(hipMemcpyAsync(scores_host, a->scores, (size_t)a->n_mut * 8, hipMemcpyDeviceToHost, s));
    if (table_host) PGMI_HIP(hipMemcpyAsync(table_host, a->table, (size_t)a->n_tok * V * 4, hipMemcpyDeviceToHost, s));
    PGMI_HIP(hipStreamSynchronize(s));
    return check_nonfinite(m);
}

// ---- pseudo-perplexity over a resident library of variable-length sequences (BASELINE config 5) --------
int pgmi_pppl_create(pgmi_model* m, const uint8_t* tokens, const int64_t* seq_off, int64_t n_seq, pgmi_pppl** out) {
    if (!out) { set_error("null out"); return PGMI_EINVAL; }
    *out = nullptr;
    if (!m || !tokens || !seq_off || n_seq <= 0 || n_seq > 0x7fffffff) { set_error("bad argument"); return PGMI_EINVAL; }
    if (m->cfg.arch != PGMI_ARCH_ESM1B && m->cfg.arch != PGMI_ARCH_ESM2) { set_error("pseudo-ppl needs an ESM-1b/1v/ESM2 model"); return PGMI_EINVAL; }
    if (seq_off[0] != 0) { set_error("seq_off[0] must be 0"); return PGMI_EINVAL; }
    for (int64_t n = 0; n < n_seq; ++n) {
        const int64_t len = seq_off[n + 1] - seq_off[n];
        // BatchConverter output: <cls> + residues + <eos> (esm/data.py:286-295); an empty sequence still has 2 tokens
        if (len < 2 || len > (1 << 24)) { set_error("sequence %lld has %lld tokens", (long long)n, (long long)len); return PGMI_EINVAL; }
        const uint8_t* t = tokens + seq_off[n];
        for (int64_t i = 0; i < len; ++i)
            if (t[i] >= PGMI_VOCAB || t[i] == PGMI_TOK_PAD) { set_error("token id %d invalid at sequence %lld, position %lld", (int)t[i], (long long)n, (long long)i); return PGMI_EINVAL; }
    }
    PGMI_HIP(hipSetDevice(m->device));
    pgmi_pppl* q = new pgmi_pppl();
    q->m = m;
    q->N = n_seq;
    q->off.assign(seq_off, seq_off + n_seq + 1);
    int rc = dev_upload(q->allocs, &q->tok8, tokens, (size_t)seq_off[n_seq]);
    if (!rc) rc = dev_upload(q->allocs, &q->off_dev, seq_off, (size_t)n_seq + 1);
    if (rc) { for (void* p : q->allocs) hipFree(p); delete q; return rc; }
    m->pppls.push_back(q);
    *out = q;
    return PGMI_OK;
}

void pgmi_pppl_destroy(pgmi_pppl* q) {
    if (!q) return;
    if (q->m) {
        hipSetDevice(q->m->device);
        hipStreamSynchronize(q->m->stream);
        auto& v = q->m->pppls;
        v.erase(std::remove(v.begin(), v.end(), q), v.end());
    }
    for (void* p : q->allocs) hipFree(p);
    delete q;
}

int64_t pgmi_pppl_rows(const pgmi_pppl* q, int64_t first, int64_t count) {
    if (!q || first < 0 || count < 0 || first + count > q->N) return -1;
    int64_t r = 0;
    for (int64_t n = first; n < first + count; ++n) r += std::max<int64_t>(0, q->off[n + 1] - q->off[n] - 4);
    return r;
}

int pgmi_pppl_run(pgmi_model* m, pgmi_pppl* q, int64_t first, int64_t count, double* scores_host, float* terms_host,
                  double* scores_dev) {
    if (!m || !q || q->m != m) { set_error("bad model/library handle (library belongs to another or a destroyed model)"); return PGMI_EINVAL; }
    if (first < 0 || count <= 0 || first + count > q->N) { set_error("sequence range [%lld, %lld) outside the library of %lld", (long long)first, (long long)(first + count), (long long)q->N); return PGMI_EINVAL; }
    PGMI_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const int J = (int)count, V = m->cfg.vocab;
    // sequences of the run in descending token length (stable): every chunk's T is its first row's length, and the
    // rows that share a chunk differ by the few residues an indel library's lengths differ by
    std::vector<int32_t> sid(J);
    for (int j = 0; j < J; ++j) sid[j] = (int32_t)(first + j);
    auto len_of = [&](int32_t n) { return q->off[n + 1] - q->off[n]; };
    std::stable_sort(sid.begin(), sid.end(), [&](int32_t a, int32_t b) { return len_of(a) > len_of(b); });
    std::vector<int64_t> rp((size_t)J + 1);
    rp[0] = 0;
    for (int j = 0; j < J; ++j) rp[j + 1] = rp[j] + std::max<int64_t>(0, len_of(sid[j]) - 4);   // i in range(1, L-1): L-2 rows
    const int64_t R = rp[J];
    const int64_t Tmax = len_of(sid[0]);
    if (R > 0 && Tmax + 31 > m->max_rows) { set_error("T=%lld exceeds workspace rows %d", (long long)Tmax, m->max_rows); return PGMI_EINVAL; }
    if (R > 0 && m->cfg.arch == PGMI_ARCH_ESM1B && Tmax > m->cfg.max_positions) {
        set_error("Sequence length %lld above maximum sequence length of %d", (long long)Tmax, m->cfg.max_positions);   // modules.py:256-260 (no windowing in compute_pppl)
        return PGMI_EINVAL;
    }
    std::vector<void*> pool;
    auto cleanup = [&]() { for (void* p : pool) hipFree(p); };
    int32_t* d_sid = nullptr;
    int64_t* d_rp = nullptr;
    float* d_terms = nullptr;
    double* d_out = nullptr;
    int rc = dev_upload(pool, &d_sid, sid.data(), (size_t)J);
    if (!rc) rc = dev_upload(pool, &d_rp, rp.data(), (size_t)J + 1);
    if (!rc) rc = dev_alloc(pool, &d_terms, (size_t)R);
    if (!rc) rc = dev_alloc(pool, &d_out, (size_t)J);
    if (rc) { cleanup(); return rc; }
    q->last_rows = R; q->last_chunks = 0; q->last_tokens = 0; q->last_padded = 0;
    int j0 = 0;
    for (int64_t g0 = 0; g0 < R;) {
        while (rp[j0 + 1] <= g0) ++j0;                       // sequence holding row g0: the longest one left
        const int T = (int)len_of(sid[j0]);
        const int per = std::max(1, m->max_rows / ((T + 31) / 32 * 32));
        const int bc = (int)std::min<int64_t>(per, R - g0);
        launch_make_pppl_rows(q->tok8, q->off_dev, d_sid, d_rp, J, g0, bc, T, m->tokens, m->row_idx, m->aux_i, s);
        rc = run_rows(m, bc, T, bc, m->row_idx);
        if (rc) { hipStreamSynchronize(s); cleanup(); return rc; }
        launch_pppl_pick(m->lp, m->aux_i, bc, V, d_terms + g0, s);
        q->last_chunks += 1;
        q->last_padded += (int64_t)bc * T;
        g0 += bc;
    }
    for (int j = 0; j < J; ++j) q->last_tokens += (rp[j + 1] - rp[j]) * len_of(sid[j]);
    {
        ProfScope p(m, PGMI_K_SCORE, 0, (double)R * 4);
        launch_pppl_sum(d_terms, d_rp, d_sid, J, first, d_out, s);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && scores_dev) e = hipMemcpyAsync(scores_dev, d_out, (size_t)J * 8, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess && scores_host) e = hipMemcpyAsync(scores_host, d_out, (size_t)J * 8, hipMemcpyDeviceToHost, s);
    std::vector<float> sorted_terms;
    if (e == hipSuccess && terms_host && R > 0) {
        sorted_terms.resize((size_t)R);
        e = hipMemcpyAsync(sorted_terms.data(), d_terms, (size_t)R * 4, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    cleanup();
    if (e != hipSuccess) { set_error("pseudo-ppl run failed: %s", hipGetErrorString(e)); return PGMI_EHIP; }
    if (terms_host && R > 0) {                               // caller order: sequence first, first+1, ... each with its rows in order
        std::vector<int64_t> dst((size_t)J + 1, 0);
        for (int k = 0; k < J; ++k) dst[k + 1] = dst[k] + std::max<int64_t>(0, len_of((int32_t)(first + k)) - 4);
        for (int j = 0; j < J; ++j)
            std::copy(sorted_terms.begin() + rp[j], sorted_terms.begin() + rp[j + 1], terms_host + dst[sid[j] - first]);
    }
    return check_nonfinite(m);
}

int pgmi_pppl_stats(const pgmi_pppl* q, int64_t* rows, int64_t* chunks, int64_t* tokens, int64_t* padded_tokens) {
    if (!q) { set_error("null library"); return PGMI_EINVAL; }
    if (rows) *rows = q->last_rows;
    if (chunks) *chunks = q->last_chunks;
    if (tokens) *tokens = q->last_tokens;
    if (padded_tokens) *padded_tokens = q->last_padded;
    return PGMI_OK;
}

int pgmi_parse_mutants(const char* text, const int64_t* str_off, int64_t n_mut, const char* sequence,
                       int seq_len, int offset_idx, int32_t* sub_pos, int32_t* sub_wt, int32_t* sub_mt,
                       int64_t* mut_off, int64_t* n_sub_out) {
    if (!text || !str_off || !sequence || n_mut < 0 || !n_sub_out) { set_error("bad argument"); return PGMI_EINVAL; }
    // alphabet: esm/constants.py:8 + esm/data.py:151-157 ("ESM-1b"/"roberta_large")
    static const char* standard = "LAGVSERTIDPKQNFYMHWCXBUZO.-";
    int32_t idx[256];
    for (int i = 0; i < 256; ++i) idx[i] = PGMI_TOK_UNK;
    for (int i = 0; standard[i]; ++i) idx[(unsigned char)standard[i]] = 4 + i;
    const bool fill = sub_pos && sub_wt && sub_mt && mut_off;
    int64_t n = 0;
    for (int64_t i = 0; i < n_mut; ++i) {
        if (fill) mut_off[i] = n;
        const char* p = text + str_off[i];
        const char* end = text + str_off[i + 1];
        while (p < end) {                       // one "A25G" token up to ':' (row.split(":"))
            const char* q = p;
            while (q < end && *q != ':') ++q;
            const int64_t len = q - p;
            if (len < 3) { set_error("malformed mutation '%.*s' in mutant %lld", (int)len, p, (long long)i); return PGMI_EPARSE; }
            const char wt = p[0], mt = q[-1];
            long pos = 0;
            bool neg = false;
            const char* d = p + 1;
            if (*d == '-') { neg = true; ++d; }
            if (d >= q - 1) { set_error("malformed mutation '%.*s' in mutant %lld", (int)len, p, (long long)i); return PGMI_EPARSE; }
            for (; d < q - 1; ++d) {
                if (*d < '0' || *d > '9') { set_error("malformed mutation '%.*s' in mutant %lld", (int)len, p, (long long)i); return PGMI_EPARSE; }
                pos = pos * 10 + (*d - '0');
                if (pos > 100000000) { set_error("position overflow in mutant %lld", (long long)i); return PGMI_EPARSE; }
            }
            if (neg) pos = -pos;
            const long k = pos - offset_idx;    // idx of label_row (compute_fitness.py:243)
            // the reference would IndexError for idx >= len; a negative idx would silently wrap in
            // python -- no dataset relies on that, it is rejected here.
            if (k < 0 || k >= seq_len) { set_error("mutation '%.*s': position %ld out of range for sequence of length %d", (int)len, p, pos, seq_len); return PGMI_EPARSE; }
            if (sequence[k] != wt) { set_error("The listed wildtype does not match the provided sequence ('%.*s': sequence has %c)", (int)len, p, sequence[k]); return PGMI_EPARSE; }
            if (fill) {
                sub_pos[n] = (int32_t)(1 + k);  // "add 1 for BOS" (compute_fitness.py:248-249)
                sub_wt[n] = idx[(unsigned char)wt];
                sub_mt[n] = idx[(unsigned char)mt];
            }
            ++n;
            p = (q < end) ? q + 1 : q;
            if (q < end && p == end) { set_error("trailing ':' in mutant %lld", (long long)i); return PGMI_EPARSE; }
        }
        if (str_off[i + 1] == str_off[i]) { set_error("empty mutant string at row %lld", (long long)i); return PGMI_EPARSE; }
    }
    if (fill) mut_off[n_mut] = n;
    *n_sub_out = n;
    return PGMI_OK;
}

int pgmi_profile_enable(pgmi_model* m, int on) {
    if (!m) { set_error("null model"); return PGMI_EINVAL; }
    int rc = prof_drain(m);
    m->prof = on != 0;
    return rc;
}

int pgmi_profile_reset(pgmi_model* m) {
    if (!m) { set_error("null model"); return PGMI_EINVAL; }
    int rc = prof_drain(m);
    for (int i = 0; i < PGMI_K_COUNT; ++i) { m->prof_ms[i] = 0; m->prof_n[i] = 0; m->prof_flops[i] = 0; m->prof_bytes[i] = 0; }
    return rc;
}

int pgmi_profile_get(pgmi_model* m, int k, double* ms, int64_t* launches, double* flops, double* bytes) {
    if (!m || k < 0 || k >= PGMI_K_COUNT) { set_error("bad argument"); return PGMI_EINVAL; }
    int rc = prof_drain(m);
    if (rc) return rc;
    if (ms) *ms = m->prof_ms[k];
    if (launches) *launches = m->prof_n[k];
    if (flops) *flops = m->prof_flops[k];
    if (bytes) *bytes = m->prof_bytes[k];
    return PGMI_OK;
}

// ---- single-op entry points for the numerics tests -------------------------------------------
int pgmi_op_layernorm(int device, const float* x, const float* w, const float* b, int rows, int D, float eps, float* y) {
    if (!x || !w || !b || !y || rows <= 0 || D <= 0 || D % 4) { set_error("bad argument"); return PGMI_EINVAL; }
    if (pgmi_device_count() <= 0) { set_error("no HIP device visible"); return PGMI_ENODEV; }
    PGMI_HIP(hipSetDevice(device));
    std::vector<void*> pool;
    float *dx, *dw, *db, *dy;
    int rc = 0;
    if ((rc = dev_upload(pool, &dx, x, (size_t)rows * D)) || (rc = dev_upload(pool, &dw, w, (size_t)D)) ||
        (rc = dev_upload(pool, &db, b, (size_t)D)) || (rc = dev_alloc(pool, &dy, (size_t)rows * D))) {
        for (void* p : pool) hipFree(p);
        return rc;
    }
    launch_layernorm(dx, dw, db, rows, D, eps, dy, nullptr);
    hipError_t e = hipMemcpy(y, dy, (size_t)rows * D * 4, hipMemcpyDeviceToHost);
    for (void* p : pool) hipFree(p);
    if (e != hipSuccess) { set_error("layernorm op failed: %s", hipGetErrorString(e)); return PGMI_EHIP; }
    return PGMI_OK;
}

int pgmi_op_gemm(int device, int precision, const float* A, const float* W, const float* bias, const float* residual,
                 int M, int N, int K, int epilogue, float* C) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) { set_error("bad argument"); return PGMI_EINVAL; }
    if (precision != PGMI_PREC_FP32 && precision != PGMI_PREC_F16X3 && precision != PGMI_PREC_BF16) { set_error("unknown precision %d", precision); return PGMI_EINVAL; }
    if (pgmi_device_count() <= 0) { set_error("no HIP device visible"); return PGMI_ENODEV; }
    PGMI_HIP(hipSetDevice(device));
    gemm_options_from_env();                             // a model-less entry of the tests: the hooks are read per call here
    std::vector<void*> pool;
    float *dA, *dW = nullptr, *dB = nullptr, *dR = nullptr, *dC;
    int rc = 0;
    if ((rc = dev_upload(pool, &dA, A, (size_t)M * K)) ||
        (precision == PGMI_PREC_FP32 && (rc = dev_upload(pool, &dW, W, (size_t)N * K))) ||
        (bias && (rc = dev_upload(pool, &dB, bias, (size_t)N))) ||
        (residual && (rc = dev_upload(pool, &dR, residual, (size_t)M * N))) ||
        (rc = dev_alloc(pool, &dC, (size_t)M * N))) {
        for (void* p : pool) hipFree(p);
        return rc;
    }
    // epilogue: EPI_* in the low byte; + 256 (f16x3 only, no residual, N % 32 == 0): run the SPLIT-PLANE output epilogue and return its
    // planes rebuilt as fp32 (tests: a row's bits through the half- and full-height items)
    const int epi = epilogue & 255;
    const bool split_planes = (epilogue & 256) != 0;
    if (split_planes && (precision != PGMI_PREC_F16X3 || residual || (N % 32))) { for (void* p : pool) hipFree(p); set_error("split-plane GEMM op: f16x3, no residual, N %% 32 == 0"); return PGMI_EINVAL; }
    if (precision == PGMI_PREC_FP32) {
        rc = launch_gemm_f32(dA, dW, dB, dR, dC, M, N, K, epi, nullptr);
    } else {
        const bool bf = precision == PGMI_PREC_BF16;
        const int planes = bf ? 1 : 2;
        W16 w16;
        unsigned short* a16 = nullptr;
        rc = make_w16(pool, W, (size_t)N * K, (size_t)K, precision, nullptr, &w16);
        if (!rc) rc = dev_alloc(pool, &a16, (size_t)M * K * planes);
        if (!rc) {
            launch_split16(dA, (int64_t)M * K, 1.0f, bf ? 1 : 2, K, a16, nullptr);
            if (split_planes) {
                // the split-plane epilogue (the next GEMM's operand): run it, then rebuild fp32 = hi + lo 2^-11 from the K-interleaved planes
                unsigned short* c16 = nullptr;
                rc = dev_alloc(pool, &c16, (size_t)M * N * 2);
                if (!rc) rc = launch_gemm16(a16, (size_t)M * K, w16.p, w16.plane, dB, nullptr, nullptr, c16, (size_t)M * N, M, N, K, epi,
                                            w16.out_scale, planes, bf, env_int("PGMI_GEMM_VARIANT", 0), nullptr);
                if (!rc) {
                    std::vector<unsigned short> h((size_t)M * N * 2);
                    hipError_t e2 = hipMemcpy(h.data(), c16, h.size() * 2, hipMemcpyDeviceToHost);
                    if (e2 != hipSuccess) { set_error("gemm op failed: %s", hipGetErrorString(e2)); rc = PGMI_EHIP; }
                    for (size_t m = 0; m < (size_t)M && !rc; ++m)
                        for (int n = 0; n < N; ++n) {
                            const size_t o = ki_off(m, n, N);
                            _Float16 hi, lo;
                            memcpy(&hi, &h[o], 2); memcpy(&lo, &h[o + 32], 2);
                            C[m * N + n] = (float)hi + (float)lo * (1.0f / kLoScale);
                        }
                }
                for (void* p : pool) hipFree(p);
                return rc;
            }
            rc = launch_gemm16(a16, (size_t)M * K, w16.p, w16.plane, dB, dR, dC, nullptr, 0, M, N, K, epi,
                               w16.out_scale, planes, bf, env_int("PGMI_GEMM_VARIANT", 0), nullptr);
        }
    }
    hipError_t e = hipMemcpy(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    for (void* p : pool) hipFree(p);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("gemm op failed: %s", hipGetErrorString(e)); return PGMI_EHIP; }
    return PGMI_OK;
}

int pgmi_tr_token_logprobs(pgmi_model* m, const int32_t* tokens, int B, int T, float* out) {
    if (!m || !tokens || !out || B <= 0 || T <= 0) { set_error("bad argument"); return PGMI_EINVAL; }
    if (m->cfg.arch != PGMI_ARCH_TRANCEPTION) { set_error("not a Tranception model"); return PGMI_EINVAL; }
    for (int64_t i = 0; i < (int64_t)B * T; ++i)
        if (tokens[i] < 0 || tokens[i] >= m->cfg.vocab) { set_error("token id %d out of range", tokens[i]); return PGMI_EINVAL; }
    if (T + 31 > m->max_rows) { set_error("T=%d exceeds workspace rows %d", T, m->max_rows); return PGMI_EINVAL; }
    PGMI_HIP(hipSetDevice(m->device));
    const int per = std::max(1, m->max_rows / ((T + 31) / 32 * 32));
    const int V = m->cfg.vocab;
    for (int b0 = 0; b0 < B; b0 += per) {
        const int bc = std::min(per, B - b0);
        PGMI_HIP(hipMemcpyAsync(m->tokens, tokens + (size_t)b0 * T, (size_t)bc * T * 4, hipMemcpyHostToDevice, m->stream));
        int rc = run_tranception(m, bc, T);
        if (rc) return rc;
        PGMI_HIP(hipMemcpyAsync(out + (size_t)b0 * T * V, m->lp, (size_t)bc * T * V * 4, hipMemcpyDeviceToHost, m->stream));
        PGMI_HIP(hipStreamSynchronize(m->stream));
    }
    return check_nonfinite(m);
}

int pgmi_tr_sequence_loglik(pgmi_model* m, const int32_t* tokens, const int32_t* lens, int B, int T,
                            const float* log_prior, int P, const int32_t* prior_a0, const int32_t* prior_row0,
                            const int32_t* prior_n, const int32_t* prior_flip, float alpha, float* out) {
    if (!m || !tokens || !lens || !out || B <= 0 || T <= 0) { set_error("bad argument"); return PGMI_EINVAL; }
    if (m->cfg.arch != PGMI_ARCH_TRANCEPTION) { set_error("not a Tranception model"); return PGMI_EINVAL; }
    if (log_prior && (!prior_a0 || !prior_row0 || !prior_n || !prior_flip || P <= 0)) { set_error("incomplete retrieval arguments"); return PGMI_EINVAL; }
    const int V = m->cfg.vocab;
    for (int b = 0; b < B; ++b) {
        if (lens[b] < 1 || lens[b] > T) { set_error("lens[%d]=%d out of range", b, lens[b]); return PGMI_EINVAL; }
        for (int t = 0; t < T; ++t) {
            const int tk = tokens[(size_t)b * T + t];
            if (tk < 0 || tk >= V) { set_error("token id %d out of range at [%d,%d]", tk, b, t); return PGMI_EINVAL; }
        }
        if (log_prior && prior_n[b] > 0 &&
            (prior_a0[b] < 0 || prior_a0[b] + prior_n[b] > T - 1 || prior_row0[b] < 0 || prior_row0[b] + prior_n[b] > P)) {
            set_error("retrieval slice of sequence %d out of range", b);
            return PGMI_EINVAL;
        }
    }
    if (T + 31 > m->max_rows) { set_error("T=%d exceeds workspace rows %d", T, m->max_rows); return PGMI_EINVAL; }
    PGMI_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    if (log_prior) {
        if (P > m->tr_prior_rows) {
            float* np_ = nullptr;
            int rc = dev_alloc(m->allocs, &np_, (size_t)P * V);
            if (rc) return rc;
            m->tr_prior = np_;
            m->tr_prior_rows = P;
        }
        PGMI_HIP(hipMemcpyAsync(m->tr_prior, log_prior, (size_t)P * V * 4, hipMemcpyHostToDevice, s));
    }
    const int per = std::max(1, m->max_rows / ((T + 31) / 32 * 32));
    for (int b0 = 0; b0 < B; b0 += per) {
        const int bc = std::min(per, B - b0);
        PGMI_HIP(hipMemcpyAsync(m->tokens, tokens + (size_t)b0 * T, (size_t)bc * T * 4, hipMemcpyHostToDevice, s));
        PGMI_HIP(hipMemcpyAsync(m->kv_len, lens + b0, (size_t)bc * 4, hipMemcpyHostToDevice, s));
        int32_t *da0 = nullptr, *dr0 = nullptr, *dn = nullptr, *dfl = nullptr;
        if (log_prior) {            // four small int arrays share aux_i / row_idx / pos_idx (pos_idx is unused by Tranception)
            da0 = m->aux_i; dr0 = m->row_idx; dn = m->pos_idx; dfl = m->pos_idx + bc;
            PGMI_HIP(hipMemcpyAsync(da0, prior_a0 + b0, (size_t)bc * 4, hipMemcpyHostToDevice, s));
            PGMI_HIP(hipMemcpyAsync(dr0, prior_row0 + b0, (size_t)bc * 4, hipMemcpyHostToDevice, s));
            PGMI_HIP(hipMemcpyAsync(dn, prior_n + b0, (size_t)bc * 4, hipMemcpyHostToDevice, s));
            PGMI_HIP(hipMemcpyAsync(dfl, prior_flip + b0, (size_t)bc * 4, hipMemcpyHostToDevice, s));
        }
        int rc = run_tranception(m, bc, T);
        if (rc) return rc;
        { ProfScope p(m, PGMI_K_SCORE, 0, (double)bc * T * 8);
          launch_seq_loglik(m->lp, m->tokens, m->kv_len, bc, T, V, log_prior ? m->tr_prior : nullptr, da0, dr0, dn, dfl, alpha,
                            m->denom, s); }
        PGMI_HIP(hipMemcpyAsync(out + b0, m->denom, (size_t)bc * 4, hipMemcpyDeviceToHost, s));
        PGMI_HIP(hipStreamSynchronize(s));
    }
    return check_nonfinite(m);
}

int pgmi_tr_sequence_loglik_shared(pgmi_model* m, const int32_t* tokens, const int32_t* ref, int B, int T,
                                   const float* log_prior, int P, const int32_t* prior_a0, const int32_t* prior_row0,
                                   const int32_t* prior_n, const int32_t* prior_flip, float alpha, float* out, float* token_logprobs,
                                   int64_t* rows_forwarded) {
    if (!m || !tokens || !ref || !out || B <= 0 || T <= 0) { set_error("bad argument"); return PGMI_EINVAL; }
    if (m->cfg.arch != PGMI_ARCH_TRANCEPTION) { set_error("not a Tranception model"); return PGMI_EINVAL; }
    if (log_prior && (!prior_a0 || !prior_row0 || !prior_n || !prior_flip || P <= 0)) { set_error("incomplete retrieval arguments"); return PGMI_EINVAL; }
    if (T > m->cfg.max_positions) { set_error("sequence of %d tokens exceeds the model context n_ctx=%d", T, m->cfg.max_positions); return PGMI_EINVAL; }
    const int V = m->cfg.vocab, D = m->cfg.embed_dim;
    const int Tpad = (T + 31) / 32 * 32;
    if (2 * Tpad > m->max_rows) { set_error("T=%d exceeds workspace rows %d", T, m->max_rows); return PGMI_EINVAL; }
    for (int b = 0; b < B; ++b) {
        if (ref[b] < 0 || ref[b] >= B || ref[ref[b]] != ref[b]) { set_error("ref[%d]=%d is not a root (a sequence that is its own reference)", b, ref[b]); return PGMI_EINVAL; }
        for (int t = 0; t < T; ++t) {
            const int tk = tokens[(size_t)b * T + t];
            if (tk < 0 || tk >= V) { set_error("token id %d out of range at [%d,%d]", tk, b, t); return PGMI_EINVAL; }
        }
        if (log_prior && prior_n[b] > 0 &&
            (prior_a0[b] < 0 || prior_a0[b] + prior_n[b] > T - 1 || prior_row0[b] < 0 || prior_row0[b] + prior_n[b] > P)) {
            set_error("retrieval slice of sequence %d out of range", b);
            return PGMI_EINVAL;
        }
    }
    PGMI_HIP(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    if (log_prior) {
        if (P > m->tr_prior_rows) {
            float* np_ = nullptr;
            int rc = dev_alloc(m->allocs, &np_, (size_t)P * V);
            if (rc) return rc;
            m->tr_prior = np_;
            m->tr_prior_rows = P;
        }
        PGMI_HIP(hipMemcpyAsync(m->tr_prior, log_prior, (size_t)P * V * 4, hipMemcpyHostToDevice, s));
    }
    // first own token of every sequence: its first difference from its root (a copy of the root: the last token)
    std::vector<int> a0(B, 0);
    std::vector<std::vector<int>> members(B);
    std::vector<int> roots;
    for (int b = 0; b < B; ++b) {
        if (ref[b] == b) { roots.push_back(b); continue; }
        const int32_t *x = tokens + (size_t)b * T, *y = tokens + (size_t)ref[b] * T;
        int p = 0;
        while (p < T && x[p] == y[p]) ++p;
        a0[b] = std::min(p, T - 1);
        members[ref[b]].push_back(b);
    }
    const int cap = m->max_rows;
    int64_t forwarded = 0;
    std::vector<float> lp_host;
    TrChunk ck;
    auto flush = [&]() -> int {
        if (ck.seq.empty()) return PGMI_OK;
        int rc = run_tranception_shared(m, ck, T, log_prior ? m->tr_prior : nullptr, prior_a0, prior_row0, prior_n, prior_flip, alpha);
        if (rc) return rc;
        const int S = (int)ck.seq.size();
        std::vector<float> res(S);
        PGMI_HIP(hipMemcpyAsync(res.data(), m->denom, (size_t)S * 4, hipMemcpyDeviceToHost, s));
        if (token_logprobs) {
            lp_host.resize((size_t)ck.rows * V);
            PGMI_HIP(hipMemcpyAsync(lp_host.data(), m->lp, lp_host.size() * 4, hipMemcpyDeviceToHost, s));
        }
        PGMI_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < S; ++i) out[ck.seq[i]] = res[i];
        if (token_logprobs)
            for (int i = 0; i < S; ++i) {
                float* dst = token_logprobs + (size_t)ck.seq[i] * T * V;
                const int a = ck.p[i], r = ck.root[i];
                if (a > 0) memcpy(dst, lp_host.data() + (size_t)ck.off[r] * V, (size_t)a * V * 4);
                memcpy(dst + (size_t)a * V, lp_host.data() + (size_t)ck.off[i] * V, (size_t)(T - a) * V * 4);
            }
        forwarded += ck.rows;
        ck = TrChunk();
        return PGMI_OK;
    };
    for (int r : roots) {
        auto padded_rows = [&](int b) { return (T - a0[b] / 32 * 32 + 31) / 32 * 32; };
        const int first = members[r].empty() ? 0 : padded_rows(members[r][0]);
        if (!ck.seq.empty() && ck.padded + Tpad + first > cap) { int rc = flush(); if (rc) return rc; }
        int rl = ck.add(r, tokens + (size_t)r * T, T, 0, -1, D);
        for (int b : members[r]) {
            if (ck.padded + padded_rows(b) > cap) {
                int rc = flush();
                if (rc) return rc;
                rl = ck.add(r, tokens + (size_t)r * T, T, 0, -1, D);           // the root again: its rows serve the rest of the group
            }
            ck.add(b, tokens + (size_t)b * T, T, a0[b], rl, D);
        }
    }
    int rc = flush();
    if (rc) return rc;
    if (rows_forwarded) *rows_forwarded = forwarded;
    return check_nonfinite(m);
}

int pgmi_bench_gemm_ab(int device, int precision, int M, int N, int K, int epilogue, int split_out, const int* variants,
                       int n_variants, int rounds, int iters, double* ms_out) {
    if (M <= 0 || N <= 0 || K <= 0 || iters <= 0 || rounds <= 0 || n_variants <= 0 || !variants || !ms_out) { set_error("bad argument"); return PGMI_EINVAL; }
    if (pgmi_device_count() <= 0) { set_error("no HIP device visible"); return PGMI_ENODEV; }
    PGMI_HIP(hipSetDevice(device));
    gemm_options_from_env();
    std::vector<void*> pool;
    auto cleanup = [&]() { for (void* p : pool) hipFree(p); };
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
    unsigned int st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) * (1.0f / 8388608.0f)) - 1.0f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hW) v = rnd() * 0.03f;
    for (auto& v : hb) v = rnd();
    float *dA, *dW = nullptr, *dB, *dC = nullptr;
    unsigned short *a16 = nullptr, *c16 = nullptr;
    int rc = 0;
    if ((rc = dev_upload(pool, &dA, hA.data(), hA.size())) || (rc = dev_upload(pool, &dB, hb.data(), hb.size()))) { cleanup(); return rc; }
    hipEvent_t e0, e1;
    PGMI_HIP(hipEventCreate(&e0));
    PGMI_HIP(hipEventCreate(&e1));
    const bool f32 = precision == PGMI_PREC_FP32;
    const bool bf = precision == PGMI_PREC_BF16;
    const int planes = bf ? 1 : 2;
    W16 w16;
    if (f32) {
        if ((rc = dev_upload(pool, &dW, hW.data(), hW.size())) || (rc = dev_alloc(pool, &dC, (size_t)M * N))) { cleanup(); return rc; }
    } else {
        if ((rc = make_w16(pool, hW.data(), hW.size(), (size_t)K, precision, nullptr, &w16)) ||
            (rc = dev_alloc(pool, &a16, (size_t)M * K * planes))) { cleanup(); return rc; }
        launch_split16(dA, (int64_t)M * K, 1.0f, bf ? 1 : 2, K, a16, nullptr);
        if (split_out == 1) rc = dev_alloc(pool, &c16, (size_t)M * N * planes);
        else if (split_out != 3) rc = dev_alloc(pool, &dC, (size_t)M * N);
        if (rc) { cleanup(); return rc; }
    }
    // split_out 2: fp32 output with the in-place residual of the out-projection / FC2 (x += ...); 3: the fused QKV epilogue
    // (attention operands; N = 3 D, sequences of 288 tokens when M allows)
    const bool fused_qkv = split_out == 3 && !f32 && !bf;
    const int Tq = (M % 288 == 0) ? 288 : M;
    unsigned short *qk16 = nullptr, *vt16 = nullptr;
    size_t qk_plane = 0, vt_plane = 0;
    if (fused_qkv) {
        if (N % 3 || (N / 3) % 64) { set_error("fused QKV bench needs N = 3 D, D %% 64 == 0"); cleanup(); return PGMI_EINVAL; }
        const size_t Tp = (size_t)(Tq + 31) / 32 * 32;
        qk_plane = (size_t)M * 2 * (N / 3);
        vt_plane = (size_t)(M / Tq) * (N / 3) * Tp;
        if ((rc = dev_alloc(pool, &qk16, qk_plane * 2)) || (rc = dev_alloc(pool, &vt16, vt_plane * 2))) { cleanup(); return rc; }
    }
    auto run = [&](int var) -> int {
        if (f32) return launch_gemm_f32(dA, dW, dB, split_out == 2 ? dC : nullptr, dC, M, N, K, epilogue, nullptr);
        if (fused_qkv)
            return launch_gemm16_qkv(a16, (size_t)M * K, w16.p, w16.plane, dB, M, N / 3, K, w16.out_scale, qk16, qk_plane, vt16, vt_plane,
                                     nullptr, nullptr, 0, Tq, N / 3 / kHeadDim, var, nullptr);
        const bool planes_out = split_out == 1;
        return launch_gemm16(a16, (size_t)M * K, w16.p, w16.plane, dB, split_out == 2 ? dC : nullptr, planes_out ? nullptr : dC,
                             planes_out ? c16 : nullptr, (size_t)M * N, M, N, K, epilogue, w16.out_scale, planes, bf, var, nullptr);
    };
    std::vector<std::vector<double>> samples(n_variants);
    for (int v = 0; v < n_variants && !rc; ++v) rc = run(variants[v] >= 0 ? variants[v] : env_int("PGMI_GEMM_VARIANT", 0));   // warm-up
    for (int r = 0; r < rounds && !rc; ++r)
        for (int v = 0; v < n_variants && !rc; ++v) {              // interleaved rounds: variants see the same clocks / temperature
            const int var = variants[v] >= 0 ? variants[v] : env_int("PGMI_GEMM_VARIANT", 0);
            hipEventRecord(e0, nullptr);
            for (int i = 0; i < iters && !rc; ++i) rc = run(var);
            hipEventRecord(e1, nullptr);
            hipError_t e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e != hipSuccess) { set_error("bench failed: %s", hipGetErrorString(e)); rc = PGMI_EHIP; }
            samples[v].push_back(ms / iters);
        }
    if (!rc)
        for (int v = 0; v < n_variants; ++v) {
            std::sort(samples[v].begin(), samples[v].end());
            ms_out[v] = samples[v][samples[v].size() / 2];         // median over the rounds
        }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    cleanup();
    return rc;
}

int pgmi_bench_gemm(int device, int precision, int M, int N, int K, int epilogue, int split_out, int variant,
                    int iters, double* ms_per_launch) {
    return pgmi_bench_gemm_ab(device, precision, M, N, K, epilogue, split_out, &variant, 1, 1, iters, ms_per_launch);
}

int pgmi_op_attention(int device, int precision, const float* qkv, const int32_t* kv_len, int B, int T, int H,
                      int rotary, float* ctx) {
    if (!qkv || !ctx || B <= 0 || T <= 0 || H <= 0) { set_error("bad argument"); return PGMI_EINVAL; }
    if (pgmi_device_count() <= 0) { set_error("no HIP device visible"); return PGMI_ENODEV; }
    PGMI_HIP(hipSetDevice(device));
    std::vector<void*> pool;
    float *dq, *dc;
    int32_t* dl = nullptr;
    const size_t D = (size_t)H * kHeadDim;
    int rc = 0;
    if ((rc = dev_upload(pool, &dq, qkv, (size_t)B * T * 3 * D)) || (kv_len && (rc = dev_upload(pool, &dl, kv_len, (size_t)B))) ||
        (rc = dev_alloc(pool, &dc, (size_t)B * T * D))) {
        for (void* p : pool) hipFree(p);
        return rc;
    }
    pgmi_model tmp;
    tmp.cfg.arch = PGMI_ARCH_ESM2;
    if (rotary) rc = ensure_rotary(&tmp, T);
    if (!rc && precision == PGMI_PREC_F16X3) {
        const size_t Tp = (size_t)(T + 31) / 32 * 32;
        unsigned short *qk = nullptr, *vt = nullptr;
        rc = dev_alloc(pool, &qk, (size_t)B * T * 2 * D * 2);
        if (!rc) rc = dev_alloc(pool, &vt, (size_t)B * Tp * D * 2);
        if (!rc) rc = launch_attention_f16x3_v2(dq, dl, tmp.rot_cos, tmp.rot_sin, rotary, B, T, H, qk, (size_t)B * T * 2 * D,
                                                vt, (size_t)B * Tp * D, dc, nullptr, 0, 0, nullptr);
    } else if (!rc) {
        if (rotary) launch_rotary(dq, tmp.rot_cos, tmp.rot_sin, B * T, T, H, nullptr);
        rc = launch_attention_f32(dq, dl, B, T, H, dc, nullptr, 0, 0, nullptr);
    }
    hipDeviceSynchronize();
    for (void* p : tmp.allocs) hipFree(p);
    hipError_t e = hipMemcpy(ctx, dc, (size_t)B * T * D * 4, hipMemcpyDeviceToHost);
    for (void* p : pool) hipFree(p);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("attention op failed: %s", hipGetErrorString(e)); return PGMI_EHIP; }
    return PGMI_OK;
}

}  // extern "C"
