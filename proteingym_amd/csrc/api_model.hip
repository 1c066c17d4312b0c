// C ABI of libpgmi.so (include/pgmi.h) and the host-side orchestration of the ESM forward.
//
// Forward order follows /root/reference/proteingym/baselines/esm/esm/model/esm1.py:116-177 and
// esm/model/esm2.py:76-130; the per-layer order follows esm/modules.py:120-142.
#include "model.h"

namespace pgmi {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

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

}  // namespace pgmi

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
    if (cfg->precision != PGMI_PREC_FP32) {              // (bf16 mode: its attention runs on the split-fp16 operands as well)
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

}  // extern "C"
