// Host-only entries of include/pgmi.h: label_row's string handling and arithmetic (compute_fitness.py:240-250) and
// get_optimal_window (utils/scoring_utils.py:43-52).  No device code.
#include "model.h"


extern "C" {

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

}  // extern "C"
