// MSA Transformer (esm/model/msa_transformer.py, esm/axial_attention.py): model creation, the forward and its C entries.
#include "model.h"

namespace pgmi {

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


// MSA Transformer forward on the token grid in m->tokens [R, C] (one alignment); leaves the residual
// stream (row-major token order) in m->x.  msa_transformer.py:146-205.
// keep_col >= 0: the caller reads only token (row 0, column keep_col) of the output (masked-marginals: compute_fitness.py:418-423
// `token_probs[:, 0, i]`).  In the LAST layer everything after the tied row attention's value update is then needed for the R
// tokens of that column only -- out-projection of the row attention, the whole column attention (one column = one sequence of R
// rows) -- and the feed-forward for the single token (0, keep_col); the same kernels on the same rows, so the kept token is
// bit-identical to the full evaluation (tests/test_gpu_msa_transformer.py).  m->x row 0 then holds that token (*compacted = true).
int run_msa(pgmi_model* m, int R, int C, int keep_col, bool* compacted) {
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

}  // namespace pgmi

extern "C" {

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

}  // extern "C"
