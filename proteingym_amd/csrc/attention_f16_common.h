// Shared by attention_f16.hip (4-wave kernels, prep passes) and attention_f16_pp.hip (the two-role 8-wave kernel).
#pragma once
#include "common.h"

namespace pgmi {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int AKT = 32;                       // keys per tile
constexpr float kAttDefer = 4.0f;             // lag allowed before O is rescaled (base-2 units; see the rescale in the kernel): +3.5 % (T = 288) / +4.5 % (T = 1024) over 0
constexpr int K_CH = AKT * 8;                 // chunks per K plane
constexpr int V_CH = 64 * 4;                  // chunks per V^T plane
constexpr int A_STAGE = 2 * K_CH + 2 * V_CH;  // chunks per buffer (hi+lo planes of K and V^T) = 16 KB

__device__ __forceinline__ unsigned int pack_h2(_Float16 a, _Float16 b) {
    const h2 t = {a, b};
    return __builtin_bit_cast(unsigned int, t);
}
__device__ __forceinline__ f32x16 mfma_h(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}


// DH = 64: one head = 64 lanes of the operand planes (head dims below 64 arrive zero-padded).  DH = 128 (ESM2-15B): a head is two
// adjacent 64-lane slot groups of the planes; S sums 8 k16 steps instead of 4, O has 4 d tiles instead of 2.
// Measured and NOT kept (round 4, scripts/att_bench.py, profiles/r4/README.md): one workgroup of 8 / 9 waves per (sequence, head)
// (K / V^T read once instead of once per query block; every query tile of T = 288 in one block) -4 % / -36 %; the score MFMAs of key
// tile kt + 1 issued inside the softmax of tile kt (own accumulators, 4-stage ring, 235 VGPRs) -3 ... -5 % at every shape.  Two
// waves share a SIMD's matrix pipe AND its VALU issue: work moved between them, or between a wave's own phases, does not net.
// RAG (Tranception prefix-shared scoring, api.hip run_tranception_shared): the launch holds SUFFIXES of sequences of T tokens.  Sequence
// b owns the packed rows [seq_off[b], seq_off[b] + T - seq_p[b]) of the residual stream / context = its tokens seq_p[b] .. T-1, and in
// the attention operand planes the rows [seq_q[b], seq_q[b] + T - a) and the V^T block seq_vt[b] (row pitch roundup(T - a, 32)) for the
// tokens from a = seq_p[b] rounded down to a multiple of 32 (the prep pass fills the head of that tile from the root's inputs).  The keys
// before a are those of its ROOT sequence seq_root[b], which is in the same launch with seq_p = 0 (the model is causal: a sequence that
// equals its root up to token seq_p - 1 has the root's K and V there, bit for bit).  Key tiles keep their ABSOLUTE alignment -- tile
// kt = keys 32 kt .. 32 kt + 31 -- and every query tile holds the same 32 queries as in a full forward, so each row goes through the
// same tiles in the same order: the same bits.  Context rows of the tokens before seq_p are not written.
// blockIdx.x indexes a list of (sequence, query block) entries.
struct RagMap {
    const int32_t* seq_off;
    const int32_t* seq_p;
    const int32_t* seq_q;
    const int32_t* seq_root;
    const uint32_t* seq_vt;        // halfs, per plane
    const int32_t* ent_seq;        // per entry of the launch's list (query blocks here, 32-token tiles in the prep pass)
    const int32_t* ent_j;
};

// Context rows as ONE bf16 plane, row-major [rows][H * DH] (OUT 3: the bf16 throughput mode's out-projection operand).  The split-plane
// epilogue's lane exchange with two dwords per (lane, column group): lane (r, kh) holds columns 8 g + 4 kh .. + 3 of its row; one
// v_permlane32_swap per dword gives every lane 8 consecutive columns = one 16-byte store.  All 64 lanes take part in the swaps.
template <int ND>
__device__ __forceinline__ void store_ctx_bf16(const f32x16 (&om)[ND], const f32x16 (&oc)[ND], float inv, float inv_lo, bool row_ok,
                                               unsigned short* row_head, int kh) {
    auto rne = [](float f) -> unsigned int {
        unsigned int u = __builtin_bit_cast(unsigned int, f);
        u += 0x7fffu + ((u >> 16) & 1u);
        return u >> 16;
    };
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            unsigned int w[2][2];
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const int g = 2 * gp + gi;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(oc[dt][4 * g + e], inv_lo, om[dt][4 * g + e]) * inv;
                w[gi][0] = rne(v[0]) | (rne(v[1]) << 16);
                w[gi][1] = rne(v[2]) | (rne(v[3]) << 16);
            }
            unsigned int first[2], second[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const auto sw = __builtin_amdgcn_permlane32_swap(w[0][k], w[1][k], false, false);
                first[k] = sw[0];
                second[k] = sw[1];
            }
            if (row_ok) *reinterpret_cast<u32x4*>(row_head + dt * 32 + 8 * (2 * gp + kh)) = u32x4{first[0], first[1], second[0], second[1]};
        }
}

// attention_f16_prep.hip: operands for launches that do not come from the fused QKV projection
void launch_qkv_prep(dim3 grid, hipStream_t s, const float* qkv, const float* cos_t, const float* sin_t, int rotary, int T, int H, int Tp,
                     unsigned short* qk16, size_t qk_plane, unsigned short* vt16, size_t vt_plane);
void launch_qkv_prep_conv(dim3 grid, hipStream_t s, const float* qkv, const float* conv, int T, int H, int Tp, unsigned short* qk16, size_t qk_plane,
                          unsigned short* vt16, size_t vt_plane, const RagMap* rag);          // rag == nullptr: dense (grid = tiles x H x B)
// attention_f16_v3.hip: the software-pipelined dense kernel
bool att_v3_serves(int T, const float* conv, const float* slopes, int head_dim);
void att_v3_set_option(int value);
int launch_att16v3(int out_mode, int wpb, dim3 grid, const unsigned short* qk16, size_t qk_plane, const unsigned short* vt16, size_t vt_plane,
                   const int32_t* kv_len, int T, int H, int Tp, float* ctx, unsigned short* ctx16, hipStream_t s, int dense_nblk, int nseq);

}  // namespace pgmi
