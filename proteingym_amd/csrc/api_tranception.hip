// Tranception (tranception/model_pytorch.py): model creation, the dense forward, prefix-shared scoring and their C entries.
#include "model.h"

namespace pgmi {

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

}  // namespace pgmi

extern "C" {

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

}  // extern "C"
