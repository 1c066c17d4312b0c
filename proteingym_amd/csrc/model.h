// Internal header of the C ABI's translation units (api_*.hip): the device-resident model / assay / library state behind the opaque
// handles of include/pgmi.h and the host-side helpers they share.  Nothing here is part of the ABI.
//   api_model.hip        errors, configuration check, weight upload, model create / destroy, options, profiling
//   api_esm.hip          ESM-1b / ESM-1v / ESM2 forward (run_encoder, run_head), masked-marginals assays, pseudo-ppl libraries
//   api_tranception.hip  Tranception forward, dense and prefix-shared; token log-probs and sequence log-likelihoods
//   api_msa.hip          MSA Transformer forward (tied row attention, column attention)
//   api_host.hip         host-only entries: mutant parser, table -> scores, optimal window
//   api_ops.hip          single-op and timing entries for the numerics tests and the A/B scripts
#pragma once
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

namespace pgmi {

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

template <typename T>
int ensure_cap(pgmi_model* m, T** p, size_t* cap, size_t need) {
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

// ---- shared helpers (api_model.hip unless noted) ----
int prof_drain(pgmi_model* m);
int check_cfg(const pgmi_config* c);
int check_tokens(const int32_t* tokens, int B, int T);
int env_int(const char* name, int dflt);
int make_w16(std::vector<void*>& pool, const float* host, size_t n, size_t K, int precision, hipStream_t s, W16* out);
int linear(pgmi_model* m, const float* in32, const unsigned short* in16, size_t in_plane, const float* W32,
           const W16& w16, const float* bias, const float* residual, float* out32, unsigned short* out16,
           size_t out_plane, int M, int N, int K, int epi);
int check_nonfinite(pgmi_model* m);
// api_esm.hip
int ensure_rotary(pgmi_model* m, int T);
int run_encoder(pgmi_model* m, int B, int T, const int32_t* keep = nullptr, int n_keep = 0, bool* compacted = nullptr);
int run_head(pgmi_model* m, int R, const int32_t* row_idx);
int run_rows(pgmi_model* m, int B, int T, int R, const int32_t* row_idx);
// api_tranception.hip
int create_tranception(pgmi_model* m, const pgmi_config* cfg, const float* w, int64_t n_weights);
int run_tranception(pgmi_model* m, int B, int T);
// api_msa.hip
int create_msa(pgmi_model* m, const pgmi_config* cfg, const float* w, int64_t n_weights);
int run_msa(pgmi_model* m, int R, int C, int keep_col = -1, bool* compacted = nullptr);

}  // namespace pgmi
