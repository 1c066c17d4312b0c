// Operand preparation for the split-fp16 attention when the operands do NOT come from the fused QKV projection's epilogue: the op-level
// entry (rotary) and Tranception (causal depth-wise convolution on q | k | v).  Layout of the planes: attention_f16.hip.
#include <stdlib.h>
#include <string.h>
#include <algorithm>

#include "attention_f16_common.h"

namespace pgmi {

// Prep pass for operands that do NOT come from the fused QKV projection (the op-level entry pgmi_op_attention): fp32 q|k|v rows
// [M][3D] -> the planes above, ESM2 rotary applied on the way (rotary_embedding.py:11-20).
__global__ __launch_bounds__(256) void qkv_prep_kernel(
    const float* __restrict__ qkv, const float* __restrict__ cos_t, const float* __restrict__ sin_t,
    int rotary, int T, int H, int Tp, unsigned short* __restrict__ qk16,
    size_t qk_plane, unsigned short* __restrict__ vt16, size_t vt_plane) {
    const int b = blockIdx.z, h = blockIdx.y, t0 = blockIdx.x * 32;
    const int tid = threadIdx.x;
    const int D = H * kHeadDim;
    const size_t RS = (size_t)3 * D;
    // ---- q and k: units of (token, which, 4 dims d..d+3 and the rotary partner d+32..d+35) ----------
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int u = tid + 256 * it;
        const int which = u >> 8, tok = (u >> 3) & 31, c = u & 7;
        const int t = t0 + tok;
        if (t < T) {
            const float* src = qkv + ((size_t)b * T + t) * RS + (size_t)which * D + h * kHeadDim + 4 * c;
            f32x4 x1 = *reinterpret_cast<const f32x4*>(src);
            f32x4 x2 = *reinterpret_cast<const f32x4*>(src + 32);
            if (rotary) {           // rotary_embedding.py:11-20: x*cos + rotate_half(x)*sin
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(cos_t + t * 64 + 4 * c);
                const f32x4 s1 = *reinterpret_cast<const f32x4*>(sin_t + t * 64 + 4 * c);
                const f32x4 c2 = *reinterpret_cast<const f32x4*>(cos_t + t * 64 + 32 + 4 * c);
                const f32x4 s2 = *reinterpret_cast<const f32x4*>(sin_t + t * 64 + 32 + 4 * c);
                f32x4 y1, y2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y1[e] = x1[e] * c1[e] + (-x2[e]) * s1[e];
                    y2[e] = x2[e] * c2[e] + x1[e] * s2[e];
                }
                x1 = y1;
                x2 = y2;
            }
            unsigned short* dst = qk16 + ((size_t)b * T + t) * (2 * D) + (size_t)which * D + h * kHeadDim + 4 * c;
            if (which == 0) {                        // base-2 softmax downstream: q carries log2(e) (common.h kQLog2e)
#pragma unroll
                for (int e = 0; e < 4; ++e) { x1[e] *= kQLog2e; x2[e] *= kQLog2e; }
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x4 x = half ? x2 : x1;
                _Float16 hh[4], ll[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xe = x[e];
                    split_act(xe, hh[e], ll[e]);
                }
                *reinterpret_cast<u32x2*>(dst + 32 * half) = u32x2{pack_h2(hh[0], hh[1]), pack_h2(hh[2], hh[3])};
                *reinterpret_cast<u32x2*>(dst + qk_plane + 32 * half) = u32x2{pack_h2(ll[0], ll[1]), pack_h2(ll[2], ll[3])};
            }
        }
    }
    // ---- v: thread (d, kq) transposes keys 8kq .. 8kq+7 of dimension d ------------------------------
    {
        const int d = tid & 63, kq = tid >> 6;
        _Float16 hh[8], ll[8];
        const float* vsrc = qkv + ((size_t)b * T) * RS + 2 * D + h * kHeadDim + d;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int t = t0 + 8 * kq + e;
            const float x = (t < T) ? vsrc[(size_t)t * RS] : 0.0f;
            split_act(x, hh[e], ll[e]);
        }
        // key 8kq+e -> position with bits 2,3 swapped: 16(kq>>1) + 8(e>>2) + 4(kq&1) + (e&3)
        unsigned short* row = vt16 + (((size_t)b * H + h) * kHeadDim + d) * Tp + t0 + 16 * (kq >> 1) + 4 * (kq & 1);
        *reinterpret_cast<u32x2*>(row) = u32x2{pack_h2(hh[0], hh[1]), pack_h2(hh[2], hh[3])};
        *reinterpret_cast<u32x2*>(row + 8) = u32x2{pack_h2(hh[4], hh[5]), pack_h2(hh[6], hh[7])};
        *reinterpret_cast<u32x2*>(row + vt_plane) = u32x2{pack_h2(ll[0], ll[1]), pack_h2(ll[2], ll[3])};
        *reinterpret_cast<u32x2*>(row + vt_plane + 8) = u32x2{pack_h2(ll[4], ll[5]), pack_h2(ll[6], ll[7])};
    }
}

// `conv` (Tranception, tranception/model_pytorch.py:73-88,240-251): per (q|k|v, head group h / (H/4), channel) a causal 7-tap
// depth-wise filter + bias, conv[((which*4 + group)*64 + d)*8 + j]; taps are right-aligned (kernel sizes 3/5/7 have leading zeros,
// group 0 is the identity), tap j multiplies the token t-6+j of the same sequence (zero before the sequence start); entry 7 is the bias.
// Tranception flavour of the prep pass (conv != nullptr, no rotary): the 38 token rows a 32-token tile
// needs (6 rows of causal history) are staged ONCE in LDS with coalesced float4 loads and the 7-tap
// filters are read from an LDS copy, instead of 7 strided global loads per output and per-tap scalar weight
// loads (computing the taps from global memory ran at 2.3 TB/s; this pass is HBM-bound: 4 B in + 4 B out per element).
// RAG (see RagMap below): blockIdx.x walks a list of (sequence, 32-token tile from the tile of its first own token on); the input rows
// of tokens before the sequence's first own token -- the causal history, and the head of that first tile -- are its ROOT's rows (same
// tokens up to there: the same pre-convolution q | k | v, bit for bit), so the tile's operand rows come out whole.
template <bool RAG>
__global__ __launch_bounds__(256) void qkv_prep_conv_kernel(
    const float* __restrict__ qkv, const float* __restrict__ conv, int T, int H, int Tp,
    unsigned short* __restrict__ qk16, size_t qk_plane, unsigned short* __restrict__ vt16, size_t vt_plane,
    const int32_t* __restrict__ seq_off, const int32_t* __restrict__ seq_p, const int32_t* __restrict__ seq_q, const int32_t* __restrict__ seq_root,
    const uint32_t* __restrict__ seq_vt, const int32_t* __restrict__ ent_seq, const int32_t* __restrict__ ent_j) {
    constexpr int RSTR = 196;                                 // 192 floats (q|k|v of one head) + pad
    constexpr int NLD = (38 * 48 + 255) / 256;                // float4 loads per thread: all issued before the first LDS store
    __shared__ __attribute__((aligned(16))) float raw[38 * RSTR];
    __shared__ __attribute__((aligned(16))) float cwl[3 * 8 * 64];   // [which][tap (7 = bias)][d]
    const int b = RAG ? ent_seq[blockIdx.x] : blockIdx.z, h = blockIdx.y;
    // p0: the sequence's first own token, a0: the 32-token tile it lies in, t0: absolute position of this tile's first token;
    // rowof(t) = packed INPUT row of token t (own rows from p0 on, the root's before); orow(t) = row of the q | k operand planes
    const int p0 = RAG ? seq_p[b] : 0, a0 = p0 & ~31;
    const int t0 = RAG ? a0 + ent_j[blockIdx.x] * 32 : blockIdx.x * 32;
    const int own0 = RAG ? seq_off[b] - p0 : b * T, root0 = RAG ? seq_off[seq_root[b]] : b * T;
    const int oq0 = RAG ? seq_q[b] - a0 : b * T;
    auto rowof = [&](int t) -> size_t { return (size_t)((RAG && t < p0) ? root0 + t : own0 + t); };
    auto orow = [&](int t) -> size_t { return (size_t)(oq0 + t); };
    const int Tpo = RAG ? (T - a0 + 31) / 32 * 32 : Tp;
    const int tid = threadIdx.x;
    const int D = H * kHeadDim;
    const size_t RS = (size_t)3 * D;
    const int group = h / (H / 4);
    f32x4 ld[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {                           // 8 independent 16-byte loads in flight per thread.  UNCONDITIONAL loads
        const int i = min(tid + 256 * k, 38 * 48 - 1);        // from clamped (always valid) addresses, zeroed afterwards: a load inside
        const int row = i / 48, seg = (i % 48) >> 4, c4 = i & 15;   // a per-lane `if` makes hipcc branch around every load and wait for it
        const int t = min(max(t0 - 6 + row, 0), T - 1);       // before the next one (8 serial round trips: measured 33 % slower than the rolled loop)
        ld[k] = *reinterpret_cast<const f32x4*>(qkv + rowof(t) * RS + (size_t)seg * D + h * kHeadDim + c4 * 4);
    }
    float cw[6];                                              // the head group's filters: 3 x 64 x 8 floats = 6 per thread, in flight with the rows
#pragma unroll                                                // (a rolled loop waited for every one of them in turn: 6 serial round trips per workgroup)
    for (int k = 0; k < 6; ++k) {
        const int i = tid + 256 * k;
        cw[k] = conv[((size_t)((i >> 9) * 4 + group) * kHeadDim + ((i >> 3) & 63)) * 8 + (i & 7)];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int i = tid + 256 * k;
        cwl[((i >> 9) * 8 + (i & 7)) * 64 + ((i >> 3) & 63)] = cw[k];
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = tid + 256 * k;
        const int row = i / 48, seg = (i % 48) >> 4, c4 = i & 15;
        const int t = t0 - 6 + row;
        const bool in_seq = t >= 0 && t < T;                  // rows before the sequence start (causal history) and beyond its end are zeros
        if (i < 38 * 48) *reinterpret_cast<f32x4*>(&raw[row * RSTR + seg * 64 + c4 * 4]) = in_seq ? ld[k] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    __syncthreads();
    // ---- q and k: unit (token, which, dims 8c .. 8c+7): one 16-byte store per plane, 8 lanes per 128-byte row segment ----------
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int u = tid + 256 * it;
        const int which = u >> 8, tok = (u >> 3) & 31, c = u & 7;
        const int t = t0 + tok;
        if (t < T) {
            f32x4 x1 = *reinterpret_cast<const f32x4*>(&cwl[(which * 8 + 7) * 64 + 8 * c]);
            f32x4 x2 = *reinterpret_cast<const f32x4*>(&cwl[(which * 8 + 7) * 64 + 8 * c + 4]);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(&cwl[(which * 8 + j) * 64 + 8 * c]);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(&cwl[(which * 8 + j) * 64 + 8 * c + 4]);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(&raw[(tok + j) * RSTR + which * 64 + 8 * c]);
                const f32x4 a2 = *reinterpret_cast<const f32x4*>(&raw[(tok + j) * RSTR + which * 64 + 8 * c + 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x1[e] = fmaf(w1[e], a1[e], x1[e]);
                    x2[e] = fmaf(w2[e], a2[e], x2[e]);
                }
            }
            if (which == 0) {                        // base-2 softmax downstream: q carries log2(e) (common.h kQLog2e)
#pragma unroll
                for (int e = 0; e < 4; ++e) { x1[e] *= kQLog2e; x2[e] *= kQLog2e; }
            }
            _Float16 hh[8], ll[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v1 = x1[e], v2 = x2[e];
                split_act(v1, hh[e], ll[e]);
                split_act(v2, hh[4 + e], ll[4 + e]);
            }
            unsigned short* dst = qk16 + orow(t) * (2 * D) + (size_t)which * D + h * kHeadDim + 8 * c;
            *reinterpret_cast<u32x4*>(dst) = u32x4{pack_h2(hh[0], hh[1]), pack_h2(hh[2], hh[3]), pack_h2(hh[4], hh[5]), pack_h2(hh[6], hh[7])};
            *reinterpret_cast<u32x4*>(dst + qk_plane) = u32x4{pack_h2(ll[0], ll[1]), pack_h2(ll[2], ll[3]), pack_h2(ll[4], ll[5]), pack_h2(ll[6], ll[7])};
        }
    }
    // ---- v: thread (d, p) produces the 8 CONSECUTIVE POSITIONS 8p .. 8p+7 of row d of the transposed tile = keys 16a + 4b + 0..3 and
    //      16a + 8 + 4b + 0..3 (a = p >> 1, b = p & 1: positions are keys with bits 2 and 3 swapped) -> one 16-byte store per plane ----
    {
        const int d = tid & 63, p = tid >> 6;
        const int k1 = 16 * (p >> 1) + 4 * (p & 1);
        _Float16 hh[8], ll[8];
        float w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = cwl[(2 * 8 + j) * 64 + d];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int key = k1 + (e & 3) + 8 * (e >> 2);
            float y = w[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) y = fmaf(w[j], raw[(key + j) * RSTR + 128 + d], y);
            if (t0 + key >= T) y = 0.0f;
            split_act(y, hh[e], ll[e]);
        }
        unsigned short* row = RAG ? vt16 + (size_t)seq_vt[b] + ((size_t)h * kHeadDim + d) * Tpo + (t0 - a0) + 8 * p
                                  : vt16 + (((size_t)b * H + h) * kHeadDim + d) * Tp + t0 + 8 * p;
        *reinterpret_cast<u32x4*>(row) = u32x4{pack_h2(hh[0], hh[1]), pack_h2(hh[2], hh[3]), pack_h2(hh[4], hh[5]), pack_h2(hh[6], hh[7])};
        *reinterpret_cast<u32x4*>(row + vt_plane) = u32x4{pack_h2(ll[0], ll[1]), pack_h2(ll[2], ll[3]), pack_h2(ll[4], ll[5]), pack_h2(ll[6], ll[7])};
    }
}

void launch_qkv_prep(dim3 grid, hipStream_t s, const float* qkv, const float* cos_t, const float* sin_t, int rotary, int T, int H, int Tp,
                     unsigned short* qk16, size_t qk_plane, unsigned short* vt16, size_t vt_plane) {
    hipLaunchKernelGGL(qkv_prep_kernel, grid, dim3(256), 0, s, qkv, cos_t, sin_t, rotary, T, H, Tp, qk16, qk_plane, vt16, vt_plane);
}
void launch_qkv_prep_conv(dim3 grid, hipStream_t s, const float* qkv, const float* conv, int T, int H, int Tp, unsigned short* qk16, size_t qk_plane,
                          unsigned short* vt16, size_t vt_plane, const RagMap* rag) {
    if (rag)
        hipLaunchKernelGGL(qkv_prep_conv_kernel<true>, grid, dim3(256), 0, s, qkv, conv, T, H, Tp, qk16, qk_plane, vt16, vt_plane,
                           rag->seq_off, rag->seq_p, rag->seq_q, rag->seq_root, rag->seq_vt, rag->ent_seq, rag->ent_j);
    else
        hipLaunchKernelGGL(qkv_prep_conv_kernel<false>, grid, dim3(256), 0, s, qkv, conv, T, H, Tp, qk16, qk_plane, vt16, vt_plane,
                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

}  // namespace pgmi
