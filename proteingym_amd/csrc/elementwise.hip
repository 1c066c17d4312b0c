// HBM-bound kernels of the ESM forward: masked-window construction, embedding (+token-dropout
// rescale, learned positions), LayerNorm, rotary, row gather/scatter, the vocabulary
// projection + log-softmax on the kept rows, and the per-mutant table lookup.
// One wave64 per row wherever a row reduction is needed (shuffle reductions, no LDS).
#include "common.h"

namespace pgmi {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- masked windows: compute_fitness.py:490-495 ------------------------------------------
__global__ void make_masked_windows_kernel(const int32_t* __restrict__ wt,
                                           const int32_t* __restrict__ win_start,
                                           const int32_t* __restrict__ mask_rel, int B, int T,
                                           int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * T) return;
    const int b = (int)(i / T), t = (int)(i % T);
    out[i] = (t == mask_rel[b]) ? PGMI_TOK_MASK : wt[win_start[b] + t];
}
void launch_make_masked_windows(const int32_t* wt, const int32_t* win_start, const int32_t* mask_rel,
                                int B, int T, int32_t* out, hipStream_t s) {
    const int64_t n = (int64_t)B * T;
    hipLaunchKernelGGL(make_masked_windows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       wt, win_start, mask_rel, B, T, out);
}

__global__ void apply_mask_kernel(int32_t* tokens, const int32_t* __restrict__ mask_pos, int B, int T) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) tokens[(int64_t)b * T + mask_pos[b]] = PGMI_TOK_MASK;
}
void launch_apply_mask(int32_t* tokens, const int32_t* mask_pos, int B, int T, hipStream_t s) {
    hipLaunchKernelGGL(apply_mask_kernel, dim3((B + 255) / 256), dim3(256), 0, s, tokens, mask_pos, B, T);
}

// ---- per-sequence statistics ---------------------------------------------------------------
// scale[b] = 1 or the token-dropout factor (1-0.15*0.8)/(1 - n_mask/n_nonpad)  (esm1.py:125-131);
// the embed kernel applies it as (x*0.88f)/denom, the reference's op order.
// pos_idx[b,t] = cumsum(tok != pad)*(tok != pad) + pad_idx          (modules.py:261-262)
// kv_len[b]    = index of the last non-pad token + 1 (the host only admits trailing padding, so
//                this is the number of valid keys for the attention mask).
__global__ void seq_stats_kernel(const int32_t* __restrict__ tokens, int B, int T, int token_dropout,
                                 float* __restrict__ denom, int32_t* __restrict__ pos_idx,
                                 int32_t* __restrict__ kv_len) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const int32_t* tk = tokens + (int64_t)b * T;
    int running = 0, n_mask = 0, last_valid = -1;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const int tok = (t < T) ? tk[t] : PGMI_TOK_PAD;
        const int valid = (t < T) && (tok != PGMI_TOK_PAD);
        // inclusive prefix sum of `valid` within the wave
        const unsigned long long bal = __ballot(valid);
        const unsigned long long below = bal & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
        const int incl = __popcll(below);
        if (t < T) pos_idx[(int64_t)b * T + t] = valid ? (running + incl + PGMI_TOK_PAD) : PGMI_TOK_PAD;
        running += __popcll(bal);
        n_mask += __popcll(__ballot((t < T) && tok == PGMI_TOK_MASK));
        if (bal) last_valid = t0 + 63 - __clzll(bal);
    }
    if (lane == 0) {
        float d = 1.0f;
        if (token_dropout) d = 1.0f - (float)n_mask / (float)running;
        denom[b] = d;
        kv_len[b] = last_valid + 1;
    }
}
void launch_seq_stats(const int32_t* tokens, int B, int T, int token_dropout, float* denom,
                      int32_t* pos_idx, int32_t* kv_len, hipStream_t s) {
    hipLaunchKernelGGL(seq_stats_kernel, dim3((B + 3) / 4), dim3(256), 0, s, tokens, B, T, token_dropout,
                       denom, pos_idx, kv_len);
}

// ---- embedding: esm1.py:123-139 / esm2.py:83-94 -------------------------------------------
// x = E[tok] (zero for <mask> under token dropout) * 0.88 / denom[b] + Wpos[pos_idx]; pad rows -> 0.
__global__ void embed_kernel(const int32_t* __restrict__ tokens, const float* __restrict__ denom,
                             const int32_t* __restrict__ pos_idx, const float* __restrict__ E,
                             const float* __restrict__ P, int token_dropout, int rows, int T, int D,
                             float* __restrict__ x) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int tok = tokens[row];
    const int b = row / T;
    const float dn = denom[b];
    const f32x4* e = reinterpret_cast<const f32x4*>(E + (size_t)tok * D);
    const f32x4* p = P ? reinterpret_cast<const f32x4*>(P + (size_t)pos_idx[row] * D) : nullptr;
    f32x4* xo = reinterpret_cast<f32x4*>(x + (size_t)row * D);
    const bool is_pad = tok == PGMI_TOK_PAD;
    const bool zero_emb = token_dropout && tok == PGMI_TOK_MASK;
    for (int i = lane; i < D / 4; i += 64) {
        f32x4 v = zero_emb ? f32x4{0.f, 0.f, 0.f, 0.f} : e[i];
        if (token_dropout) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (v[c] * 0.88f) / dn;
        }
        if (p) {
            const f32x4 pv = p[i];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] += pv[c];
        }
        if (is_pad) v = f32x4{0.f, 0.f, 0.f, 0.f};
        xo[i] = v;
    }
}
void launch_embed(const int32_t* tokens, const float* denom, const int32_t* pos_idx, const float* E,
                  const float* P, int token_dropout, int rows, int T, int D, float* x, hipStream_t s) {
    hipLaunchKernelGGL(embed_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, tokens, denom, pos_idx, E, P,
                       token_dropout, rows, T, D, x);
}

// zero the rows of <pad> tokens (esm1.py:138-139), needed after emb_layer_norm_before
__global__ void zero_pad_rows_kernel(const int32_t* __restrict__ tokens, int rows, int D, float* __restrict__ x) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows || tokens[row] != PGMI_TOK_PAD) return;
    for (int i = lane; i < D; i += 64) x[(size_t)row * D + i] = 0.f;
}
void launch_zero_pad_rows(const int32_t* tokens, int rows, int D, float* x, hipStream_t s) {
    hipLaunchKernelGGL(zero_pad_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, tokens, rows, D, x);
}

// ---- LayerNorm (torch.nn.LayerNorm, modules.py:80-81): biased variance, eps inside sqrt ----
// One wave per row, the row held in registers (D <= 64*4*NV), two-pass mean / variance.
// Algorithmic bytes: 2*D*4 per row (read + write); HBM-bound.
// OUTMODE 0: fp32 y;  1: fp16 hi/lo, K-interleaved (f16x3 GEMM operand; `plane` unused);  2: bf16 plane
template <int NV, int OUTMODE>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x,   // may alias y (in place)
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bsh, int rows, int D,
                                                        float eps, float* y, unsigned short* y16,
                                                        size_t plane) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nv = D >> 2;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * D);
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        v[i] = (c < nv) ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = v[i][k] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    const f32x4* wr = reinterpret_cast<const f32x4*>(w);
    const f32x4* br = reinterpret_cast<const f32x4*>(bsh);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const f32x4 wv = wr[c], bv = br[c];
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (v[i][k] - mean) * rstd * wv[k] + bv[k];
            if constexpr (OUTMODE == 0) {
                reinterpret_cast<f32x4*>(y + (size_t)row * D)[c] = o;
            } else if constexpr (OUTMODE == 1) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 hi, lo;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    _Float16 a, b2;
                    split_act(o[k], a, b2);
                    hi[k] = a;
                    lo[k] = b2;
                }
                // K-interleaved GEMM operand (common.h ki_off).  Lanes (2m, 2m+1) hold k..k+3 and k+4..k+7 of one 8-group:
                // the even lane stores the 16 bytes of hi values, the odd lane the 16 bytes of lo values (one exchange of
                // 8 bytes with the neighbour), so 8 lanes write one full 128-byte line with a single store each.
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                const u32x2 H = __builtin_bit_cast(u32x2, hi), L = __builtin_bit_cast(u32x2, lo);
                const bool odd = lane & 1;
                const unsigned int s0 = odd ? H[0] : L[0], s1 = odd ? H[1] : L[1];
                const unsigned int r0 = __shfl_xor(s0, 1), r1 = __shfl_xor(s1, 1);
                const u32x4 out = odd ? u32x4{r0, r1, L[0], L[1]} : u32x4{H[0], H[1], r0, r1};
                unsigned short* dst = y16 + ki_off((size_t)row, 4 * (c & ~1), D) + (odd ? 32 : 0);
                *reinterpret_cast<u32x4*>(dst) = out;
            } else {
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                unsigned short b[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float ok = o[k];     // copy first: bit_cast of a vector-element lvalue reads element 0
                    unsigned int u = __builtin_bit_cast(unsigned int, ok);
                    u += 0x7fffu + ((u >> 16) & 1u);
                    b[k] = (unsigned short)(u >> 16);
                }
                u32x2 pk;
                pk[0] = b[0] | ((unsigned)b[1] << 16);
                pk[1] = b[2] | ((unsigned)b[3] << 16);
                reinterpret_cast<u32x2*>(y16 + (size_t)row * D)[c] = pk;
            }
        }
    }
}
template <int OUTMODE>
static void launch_ln_mode(const float* x, const float* w, const float* b, int rows, int D, float eps,
                           float* y, unsigned short* y16, size_t plane, hipStream_t s) {
    const dim3 grid((rows + 3) / 4), block(256);
    const int nv = (D / 4 + 63) / 64;
    if (nv <= 1) hipLaunchKernelGGL((layernorm_kernel<1, OUTMODE>), grid, block, 0, s, x, w, b, rows, D, eps, y, y16, plane);
    else if (nv <= 2) hipLaunchKernelGGL((layernorm_kernel<2, OUTMODE>), grid, block, 0, s, x, w, b, rows, D, eps, y, y16, plane);
    else if (nv <= 5) hipLaunchKernelGGL((layernorm_kernel<5, OUTMODE>), grid, block, 0, s, x, w, b, rows, D, eps, y, y16, plane);
    else if (nv <= 10) hipLaunchKernelGGL((layernorm_kernel<10, OUTMODE>), grid, block, 0, s, x, w, b, rows, D, eps, y, y16, plane);
    else hipLaunchKernelGGL((layernorm_kernel<20, OUTMODE>), grid, block, 0, s, x, w, b, rows, D, eps, y, y16, plane);
}
void launch_layernorm(const float* x, const float* w, const float* b, int rows, int D, float eps,
                      float* y, hipStream_t s) {
    launch_ln_mode<0>(x, w, b, rows, D, eps, y, nullptr, 0, s);
}
// mode 1: fp16 hi/lo planes, mode 2: bf16
void launch_layernorm16(const float* x, const float* w, const float* b, int rows, int D, float eps,
                        unsigned short* y16, size_t plane, int mode, hipStream_t s) {
    if (mode == 1) launch_ln_mode<1>(x, w, b, rows, D, eps, nullptr, y16, plane, s);
    else launch_ln_mode<2>(x, w, b, rows, D, eps, nullptr, y16, plane, s);
}

// ---- rotary (rotary_embedding.py:11-20,47-69): half-split rotation of q (already scaled) and k
// qkv [rows, 3*H*64]; tables cos/sin [T, 64] (emb = cat(freqs,freqs)) built on the host in f32
// exactly as the reference does.  One thread handles the pair (d, d+32) of one head.
__global__ void rotary_kernel(float* __restrict__ qkv, const float* __restrict__ cos_t,
                              const float* __restrict__ sin_t, int64_t n, int T, int H, int rh) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;                      // n = rows * 2 * H * 32
    const int d = (int)(i & 31);
    const int64_t j = i >> 5;
    const int h = (int)(j % H);              // 64-lane slot group: a head, or half a head of 128 dims
    const int64_t k2 = j / H;
    const int which = (int)(k2 & 1);         // 0 = q, 1 = k
    const int64_t row = k2 >> 1;
    const int t = (int)(row % T);
    float* p = qkv + row * (size_t)(3 * H * 64) + (size_t)which * H * 64 + h * 64 + d;
    const float x1 = p[0], x2 = p[32];
    const int tr = (t * rh + (h % rh)) * 64;  // table row: one per token, or per (token, slot-group parity) for head_dim 128 (api_esm.hip ensure_rotary)
    const float c1 = cos_t[tr + d], s1 = sin_t[tr + d];
    const float c2 = cos_t[tr + d + 32], s2 = sin_t[tr + d + 32];
    p[0] = x1 * c1 + (-x2) * s1;             // x*cos + rotate_half(x)*sin, first half: -x2
    p[32] = x2 * c2 + x1 * s2;               // second half: +x1
}
void launch_rotary(float* qkv, const float* cos_t, const float* sin_t, int rows, int T, int H, hipStream_t s, int rot_halves) {
    const int64_t n = (int64_t)rows * 2 * H * 32;
    hipLaunchKernelGGL(rotary_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, qkv, cos_t, sin_t, n, T, H, rot_halves);
}

// ---- row gather / scatter -------------------------------------------------------------------
__global__ void gather_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx, int n,
                                   int D, float* __restrict__ y) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const f32x4* src = reinterpret_cast<const f32x4*>(x + (size_t)idx[row] * D);
    f32x4* dst = reinterpret_cast<f32x4*>(y + (size_t)row * D);
    for (int i = lane; i < D / 4; i += 64) dst[i] = src[i];
}
void launch_gather_rows(const float* x, const int32_t* idx, int n, int D, float* y, hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, s, x, idx, n, D, y);
}

__global__ void strided_index_kernel(int first, int stride, int n, int32_t* __restrict__ idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = first + i * stride;
}
void launch_strided_index(int first, int stride, int n, int32_t* idx, hipStream_t s) {
    hipLaunchKernelGGL(strided_index_kernel, dim3((n + 255) / 256), dim3(256), 0, s, first, stride, n, idx);
}

__global__ void scatter_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ dst_row,
                                    int n, int V, float* __restrict__ table) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * V) return;
    const int rI = i / V, c = i % V;
    table[(size_t)dst_row[rI] * V + c] = src[i];
}
void launch_scatter_rows(const float* src, const int32_t* dst_row, int n, int V, float* table, hipStream_t s) {
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((n * V + 255) / 256), dim3(256), 0, s, src, dst_row, n, V, table);
}

__global__ void row_index_kernel(const int32_t* __restrict__ mask_rel, int B, int T, int32_t* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out[b] = b * T + mask_rel[b];
}
void launch_row_index(const int32_t* mask_rel, int B, int T, int32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(row_index_kernel, dim3((B + 255) / 256), dim3(256), 0, s, mask_rel, B, T, out);
}

__global__ void fill_f32_kernel(float* p, int64_t n, float v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
void launch_fill_f32(float* p, int64_t n, float v, hipStream_t s) {
    hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
}

// ---- vocabulary projection + log-softmax (modules.py:327 ; compute_fitness.py:502) ---------
// One wave per row: logits[v] = <h, E[v]> + bias[v] for the 33 symbols, then x - max - log(sum exp).
__global__ __launch_bounds__(256) void vocab_logsoftmax_kernel(const float* __restrict__ h,
                                                               const float* __restrict__ E,
                                                               const float* __restrict__ bias, int rows,
                                                               int D, int V, float* __restrict__ out,
                                                               int32_t* __restrict__ nonfinite) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    // The 33 (25) logits of a row and their log-sum-exp are accumulated in DOUBLE and rounded to fp32 once: the work is negligible
    // (V dot products of D per kept row) and the result carries no rounding of its own on top of the fp32 input row -- a
    // pseudo-ppl score sums ~700 of these values, where a systematic last-place bias of an fp32 log-softmax (values ~ -16: one
    // ulp = 1.9e-6) would add up to 1e-3 (tests/test_gpu_parity_real_width.py).
    const float* hr = h + (size_t)row * D;
    double my_logit = -INFINITY;           // lane v (< V) ends up owning logit v
    for (int v = 0; v < V; ++v) {
        const float* ev = E + (size_t)v * D;
        double acc = 0.0;
        for (int i = lane; i < D; i += 64) acc = fma((double)hr[i], (double)ev[i], acc);
        acc = wave_sum_d(acc);
        if (lane == v) my_logit = acc + (double)bias[v];
    }
    const float mx = wave_max((float)my_logit);
    const double ex = (lane < V) ? exp(my_logit - (double)mx) : 0.0;
    const double lse = log(wave_sum_d(ex));
    const float res = (float)((my_logit - (double)mx) - lse);
    if (lane < V) {
        out[(size_t)row * V + lane] = res;
        if (nonfinite && !(fabsf(res) <= 3.0e38f)) atomicOr(nonfinite, 1);   // NaN/inf: fp16 overflow upstream
    }
}
void launch_vocab_logsoftmax(const float* h, const float* E, const float* bias, int rows, int D, int V,
                             float* out, int32_t* nonfinite, hipStream_t s) {
    hipLaunchKernelGGL(vocab_logsoftmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, h, E, bias, rows, D, V, out, nonfinite);
}

// ---- Tranception sequence log-likelihood (one wave per sequence) ------------------------------
__global__ __launch_bounds__(256) void seq_loglik_kernel(const float* __restrict__ lp, const int32_t* __restrict__ tokens,
                                                         const int32_t* __restrict__ lens, int B, int T, int V,
                                                         const float* __restrict__ prior, const int32_t* __restrict__ a0,
                                                         const int32_t* __restrict__ row0, const int32_t* __restrict__ n,
                                                         const int32_t* __restrict__ flip, float alpha,
                                                         float* __restrict__ out) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const int len = lens[b];
    float acc = 0.f;
    for (int t = lane; t < len - 1; t += 64) {
        const int tgt = tokens[(size_t)b * T + t + 1];
        float v = lp[((size_t)b * T + t) * V + tgt];
        if (prior && n[b] > 0 && t >= a0[b] && t < a0[b] + n[b]) {
            const int i = t - a0[b];
            const int row = flip[b] ? row0[b] + (n[b] - 1 - i) : row0[b] + i;
            v = (1.0f - alpha) * v + alpha * prior[(size_t)row * V + tgt];
        }
        acc += v;
    }
    acc = wave_sum(acc);
    if (lane == 0) out[b] = acc;
}
// The same reduction over SUFFIX rows (Tranception prefix-shared scoring, common.h AttRagged): sequence b owns the packed rows of its
// tokens seq_p[b] .. T-1 (log-probabilities in lp, ids in tokens, both packed); the rows before seq_p[b] are its root's (the model is
// causal: the same tokens give the same rows).  Same lane partition, same order, same arithmetic as seq_loglik_kernel: with the same
// row values the sum has the same bits.
__global__ __launch_bounds__(256) void seq_loglik_ragged_kernel(const float* __restrict__ lp, const int32_t* __restrict__ tokens,
                                                                const int32_t* __restrict__ seq_off, const int32_t* __restrict__ seq_p,
                                                                const int32_t* __restrict__ seq_root, int B, int T, int V,
                                                                const float* __restrict__ prior, const int32_t* __restrict__ a0,
                                                                const int32_t* __restrict__ row0, const int32_t* __restrict__ n,
                                                                const int32_t* __restrict__ flip, float alpha,
                                                                float* __restrict__ out) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const int a = seq_p[b], own0 = seq_off[b] - a, root0 = seq_off[seq_root[b]];
    float acc = 0.f;
    for (int t = lane; t < T - 1; t += 64) {
        const int tgt = tokens[(t + 1 < a) ? root0 + t + 1 : own0 + t + 1];
        float v = lp[(size_t)((t < a) ? root0 + t : own0 + t) * V + tgt];
        if (prior && n[b] > 0 && t >= a0[b] && t < a0[b] + n[b]) {
            const int i = t - a0[b];
            const int row = flip[b] ? row0[b] + (n[b] - 1 - i) : row0[b] + i;
            v = (1.0f - alpha) * v + alpha * prior[(size_t)row * V + tgt];
        }
        acc += v;
    }
    acc = wave_sum(acc);
    if (lane == 0) out[b] = acc;
}
void launch_seq_loglik_ragged(const float* lp, const int32_t* tokens, const int32_t* seq_off, const int32_t* seq_p,
                              const int32_t* seq_root, int B, int T, int V, const float* prior, const int32_t* a0,
                              const int32_t* row0, const int32_t* n, const int32_t* flip, float alpha, float* out, hipStream_t s) {
    hipLaunchKernelGGL(seq_loglik_ragged_kernel, dim3((B + 3) / 4), dim3(256), 0, s, lp, tokens, seq_off, seq_p, seq_root, B, T, V,
                       prior, a0, row0, n, flip, alpha, out);
}
void launch_seq_loglik(const float* lp, const int32_t* tokens, const int32_t* lens, int B, int T, int V,
                       const float* prior, const int32_t* a0, const int32_t* row0, const int32_t* n,
                       const int32_t* flip, float alpha, float* out, hipStream_t s) {
    hipLaunchKernelGGL(seq_loglik_kernel, dim3((B + 3) / 4), dim3(256), 0, s, lp, tokens, lens, B, T, V, prior, a0, row0, n,
                       flip, alpha, out);
}

// ---- label_row (compute_fitness.py:240-250): score = sum_subs f32(lp[mt] - lp[wt]) in double ---
__global__ void score_mutants_kernel(const float* __restrict__ table, int V,
                                     const int32_t* __restrict__ sub_pos,
                                     const int32_t* __restrict__ sub_wt,
                                     const int32_t* __restrict__ sub_mt,
                                     const int64_t* __restrict__ mut_off, int64_t n_mut,
                                     double* __restrict__ scores) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_mut) return;
    double sc = 0.0;
    for (int64_t k = mut_off[i]; k < mut_off[i + 1]; ++k) {
        const float* rowp = table + (size_t)sub_pos[k] * V;
        const float d = rowp[sub_mt[k]] - rowp[sub_wt[k]];
        sc += (double)d;
    }
    scores[i] = sc;
}
void launch_score_mutants(const float* table, int V, const int32_t* sub_pos, const int32_t* sub_wt,
                          const int32_t* sub_mt, const int64_t* mut_off, int64_t n_mut, double* scores,
                          hipStream_t s) {
    if (n_mut <= 0) return;
    hipLaunchKernelGGL(score_mutants_kernel, dim3((unsigned)((n_mut + 255) / 256)), dim3(256), 0, s, table, V,
                       sub_pos, sub_wt, sub_mt, mut_off, n_mut, scores);
}


// ---- pseudo-perplexity rows (compute_fitness.py:258-279), enumerated on the device ----------------------
// A library of variable-length sequences is resident as one byte string tok8 (cls + residues + eos per
// sequence, seq_off[n] .. seq_off[n+1]).  sid[j] lists the sequences of a run in descending token length,
// rp[j] is the number of (sequence, masked position) rows before sid[j] in that order: sequence n with `len`
// tokens (L = len-2 residues) contributes the rows i = 1 .. L-2 (the reference loops i in range(1, len(seq)-1)
// and masks TOKEN i).  Row g of the run = (sid[j], i = 1 + g - rp[j]) with rp[j] <= g < rp[j+1]: one wave per row
// finds j by bisection, then writes the T tokens of the row (<pad> beyond the sequence, <mask> at i), the
// flat index of the masked token and the target token: the reference scores alphabet.get_idx(sequence[i]) =
// token i+1 at masked token i (its off-by-one, reproduced).
__global__ __launch_bounds__(256) void make_pppl_rows_kernel(const uint8_t* __restrict__ tok8, const int64_t* __restrict__ seq_off,
                                                             const int32_t* __restrict__ sid, const int64_t* __restrict__ rp, int J,
                                                             int64_t g0, int bc, int T, int32_t* __restrict__ tokens,
                                                             int32_t* __restrict__ row_idx, int32_t* __restrict__ target) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= bc) return;
    const int64_t g = g0 + b;
    int lo = 0, hi = J;                                   // largest j with rp[j] <= g (rp[J] = total rows > g)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rp[mid] <= g) lo = mid; else hi = mid;
    }
    const int n = sid[lo];
    const int i = 1 + (int)(g - rp[lo]);
    const int64_t o = seq_off[n];
    const int len = (int)(seq_off[n + 1] - o);
    int32_t* out = tokens + (int64_t)b * T;
    for (int t = lane; t < T; t += 64) out[t] = (t == i) ? PGMI_TOK_MASK : (t < len ? (int32_t)tok8[o + t] : PGMI_TOK_PAD);
    if (lane == 0) {
        row_idx[b] = b * T + i;
        target[b] = (int32_t)tok8[o + i + 1];
    }
}
void launch_make_pppl_rows(const uint8_t* tok8, const int64_t* seq_off, const int32_t* sid, const int64_t* rp, int J,
                           int64_t g0, int bc, int T, int32_t* tokens, int32_t* row_idx, int32_t* target, hipStream_t s) {
    hipLaunchKernelGGL(make_pppl_rows_kernel, dim3((bc + 3) / 4), dim3(256), 0, s, tok8, seq_off, sid, rp, J, g0, bc, T,
                       tokens, row_idx, target);
}

// terms[g0 + b] = lp[b, target[b]]   (token_probs[0, i, alphabet.get_idx(sequence[i])], compute_fitness.py:275)
__global__ void pppl_pick_kernel(const float* __restrict__ lp, const int32_t* __restrict__ target, int bc, int V,
                                 float* __restrict__ terms) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < bc) terms[b] = lp[(size_t)b * V + target[b]];
}
void launch_pppl_pick(const float* lp, const int32_t* target, int bc, int V, float* terms, hipStream_t s) {
    hipLaunchKernelGGL(pppl_pick_kernel, dim3((bc + 255) / 256), dim3(256), 0, s, lp, target, bc, V, terms);
}

// out[sid[j] - first] = sum(log_probs): python's left-to-right sum of the f32 terms' double values (:279)
__global__ void pppl_sum_kernel(const float* __restrict__ terms, const int64_t* __restrict__ rp, const int32_t* __restrict__ sid,
                                int J, int64_t first, double* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= J) return;
    double acc = 0.0;
    for (int64_t g = rp[j]; g < rp[j + 1]; ++g) acc += (double)terms[g];
    out[sid[j] - first] = acc;
}
void launch_pppl_sum(const float* terms, const int64_t* rp, const int32_t* sid, int J, int64_t first, double* out, hipStream_t s) {
    if (J <= 0) return;
    hipLaunchKernelGGL(pppl_sum_kernel, dim3((J + 127) / 128), dim3(128), 0, s, terms, rp, sid, J, first, out);
}

}  // namespace pgmi
