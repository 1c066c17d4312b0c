// ESM-1b / ESM-1v / ESM2: the forward (esm/model/esm1.py:116-177, esm2.py:76-130, modules.py:120-142), masked-marginals assays
// (compute_fitness.py:486-514) and pseudo-perplexity libraries (compute_fitness.py:258-279,515-529) behind include/pgmi.h.
#include "model.h"

namespace pgmi {

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

// Runs the encoder on tokens already in m->tokens [B,T]; leaves the residual stream in m->x.
// keep != nullptr (device, n_keep row indices into [B*T]): the caller reads only these rows of the output (the masked
// position of every sequence: compute_fitness.py:503 `token_probs[:, i]`, :274-276).  Everything after the last layer's
// attention is row-local (out-projection, LayerNorm, FFN: modules.py:126-141), so the last layer gathers the kept rows
// of the attention context and of the residual stream and runs those stages on n_keep rows; m->x then holds the kept
// rows COMPACTED (row j = keep[j]) and *compacted is set.  The kept rows are bit-identical to the full evaluation: every
// kernel on the way computes a row from that row's inputs only, in an order that does not depend on the row count
// (tests/test_gpu_esm.py::test_last_layer_kept_rows_bit_identical).  PGMI_KEEP_ROWS=0 turns it off.
int run_encoder(pgmi_model* m, int B, int T, const int32_t* keep, int n_keep, bool* compacted) {
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
        // attention operands straight from the QKV projection's epilogue -- in the bf16 mode too (round 6): the epilogue splits the fp32
        // accumulators whatever the GEMM's operand type was, so that mode's attention runs on the 16-bit pipe as well
        const bool fused_qkv = prec != PGMI_PREC_FP32;
        { ProfScope p(m, PGMI_K_GEMM_QKV, 2.0 * M * 3 * D * D, 0);
          if (fused_qkv)
              rc = launch_gemm16_qkv(m->h16, m->h16_plane, L.wqkv16.p, L.wqkv16.plane, L.bqkv, M, Da, D, L.wqkv16.out_scale,
                                     m->qk16, m->qk16_plane, m->vt16, m->vt16_plane, m->rot_cos, m->rot_sin,
                                     c.arch == PGMI_ARCH_ESM2, T, m->Hs, m->gemm_variant, s, m->rot_halves, prec == PGMI_PREC_BF16);
          else
              rc = linear(m, m->h, m->h16, m->h16_plane, L.wqkv, L.wqkv16, L.bqkv, nullptr, m->qkv, nullptr, 0, M, 3 * Da, D, EPI_NONE);
          if (rc) return rc; }
        { ProfScope p(m, PGMI_K_ATTENTION, 4.0 * M * T * D, 0);
          const bool v2 = prec != PGMI_PREC_FP32;
          if (c.arch == PGMI_ARCH_ESM2 && !v2) launch_rotary(m->qkv, m->rot_cos, m->rot_sin, M, T, m->Hs, s, m->rot_halves);
          if (prec == PGMI_PREC_F16X3)
              rc = launch_attention_f16x3_v2(fused_qkv ? nullptr : m->qkv, m->kv_len, m->rot_cos, m->rot_sin, c.arch == PGMI_ARCH_ESM2, B, T, H,
                                             m->qk16, m->qk16_plane, m->vt16, m->vt16_plane, nullptr, m->h16,
                                             m->h16_plane, 1, s, nullptr, nullptr, m->rot_halves * kHeadDim);
          else if (v2)         // bf16 mode: the context rows leave as one bf16 plane (the out-projection's operand)
              rc = launch_attention_f16x3_v2(nullptr, m->kv_len, m->rot_cos, m->rot_sin, c.arch == PGMI_ARCH_ESM2, B, T, H,
                                             m->qk16, m->qk16_plane, m->vt16, m->vt16_plane, nullptr, m->h16, m->h16_plane, 2, s, nullptr, nullptr,
                                             m->rot_halves * kHeadDim);
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

}  // namespace pgmi

extern "C" {

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
        // a call can hold hours of forwards (CAPSD_AAV2S: 1.6e8 rows): the fp16 range flag is read every 64 chunks
        // (~20 s at T = 737), not only at the end, so that the caller's fp32 re-run starts when the overflow happens
        if ((q->last_chunks & 63) == 0 && g0 < R && (rc = check_nonfinite(m)) != PGMI_OK) { cleanup(); return rc; }
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

}  // extern "C"
