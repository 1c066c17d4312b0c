// Kernels specific to the MSA Transformer (axial attention) forward on gfx950.
//
// Reference: proteingym/baselines/esm/esm/model/msa_transformer.py:146-205 (forward),
// esm/axial_attention.py:33-168 (tied row attention), :171-297 (column attention).
//
// Token grid [R rows (sequences), C columns], residual stream x fp32 [R*C, D] in row-major token order
// (row r, column c) -> r*C + c.
//   * tied row attention: S_h[i,j] = sum_{r,d} q[r,i,h,d] k[r,j,h,d] is ONE [C,C] score matrix per head,
//     shared by all rows.  Both contractions run on the fp32 matrix pipe through the strided/batched
//     form of gemm_f32 (no packing of q,k: the K axis walks (r, d) with a row stride); this file holds
//     the pieces around them: the softmax over the split-K partial sums and the V transpose.
//   * column attention: for each column an ordinary attention over the R rows -> the residual stream
//     is permuted to column-major token order and the regular f16x3 attention path (fused QKV epilogue
//     + DMA-ring kernel) runs with batch = C, T = R.
// All kernels here are HBM-bound data movement or tiny reductions.
#include "common.h"

namespace pgmi {

// ---- token grid for one masked position --------------------------------------------------------
// out[r, t] = full[r, start + t], except out[0, mask_col - start] = <mask>   (compute_fitness.py:381-388)
__global__ void msa_window_tokens_kernel(const int32_t* __restrict__ full, int R, int Tfull, int start, int Tw,
                                         int mask_col, int32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * Tw) return;
    const int r = i / Tw, t = i % Tw;
    int v = full[(size_t)r * Tfull + start + t];
    if (r == 0 && start + t == mask_col) v = PGMI_TOK_MASK;
    out[i] = v;
}
void launch_msa_window_tokens(const int32_t* full, int R, int Tfull, int start, int Tw, int mask_col, int32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(msa_window_tokens_kernel, dim3((R * Tw + 255) / 256), dim3(256), 0, s, full, R, Tfull, start, Tw, mask_col, out);
}

// ---- x[(r*C + c), :] += pe[r, :]   (msa_position_embedding, msa_transformer.py:160-166) -----------
__global__ void add_row_embedding_kernel(float* __restrict__ x, const float* __restrict__ pe, int R, int C, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // float4 index
    const int d4 = D / 4;
    if (i >= (int64_t)R * C * d4) return;
    const int64_t row = i / d4;
    const int c4 = (int)(i % d4);
    const int r = (int)(row / C);
    f32x4 v = reinterpret_cast<f32x4*>(x)[i];
    const f32x4 e = reinterpret_cast<const f32x4*>(pe)[(size_t)r * d4 + c4];
    v += e;
    reinterpret_cast<f32x4*>(x)[i] = v;
}
void launch_add_row_embedding(float* x, const float* pe, int R, int C, int D, hipStream_t s) {
    const int64_t n = (int64_t)R * C * (D / 4);
    hipLaunchKernelGGL(add_row_embedding_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, pe, R, C, D);
}

// ---- token-order permutation: dst[(b*A + a), :] = src[(a*B + b), :] -------------------------------
// (A, B) = (R, C) turns row-major token order into column-major; (C, R) turns it back.
__global__ void permute_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int A, int B, int D) {
    const int d4 = D / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)A * B * d4) return;
    const int64_t drow = i / d4;
    const int c4 = (int)(i % d4);
    const int b = (int)(drow / A), a = (int)(drow % A);
    reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[((size_t)a * B + b) * d4 + c4];
}
void launch_permute_rows(const float* src, float* dst, int A, int B, int D, hipStream_t s) {
    const int64_t n = (int64_t)A * B * (D / 4);
    hipLaunchKernelGGL(permute_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, A, B, D);
}

// ---- tied row attention on the 16-bit matrix pipe (axial_attention.py:112-168) -------------------------------------
// S_h[i,j] = sum_{r,d} q[r,i,h,d] k[r,j,h,d] and O[r,i,h,:] = sum_j P_h[i,j] v[r,j,h,:] are batched GEMMs of gemm16x_kernel (XMap form,
// gemm_f16.hip launch_gemm16_ex): q and P are its A operand (activation split: hi = fp16(x), lo = fp16((x - hi) 2^11)), k and V^T its
// W operand, which the kernel wants like a weight: hi = fp16(w 2^s), lo = fp16(w 2^s - hi), 2^-s in the epilogue -- here s = 6
// (|k|, |v| up to 1 023 stay inside fp16; larger values end in the non-finite guard, never in a wrong number).
constexpr float kTiedWScale = 64.0f;

__device__ __forceinline__ void store_split8(unsigned short* dst, const float (&x)[8], bool as_weight) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        _Float16 h0, l0, h1, l1;
        if (as_weight) {
            h0 = (_Float16)x[2 * e]; l0 = (_Float16)(x[2 * e] - (float)h0);
            h1 = (_Float16)x[2 * e + 1]; l1 = (_Float16)(x[2 * e + 1] - (float)h1);
        } else {
            split_act(x[2 * e], h0, l0);
            split_act(x[2 * e + 1], h1, l1);
        }
        const h2 a = {h0, h1}, b = {l0, l1};
        hi[e] = __builtin_bit_cast(unsigned int, a);
        lo[e] = __builtin_bit_cast(unsigned int, b);
    }
    *reinterpret_cast<u32x4*>(dst) = hi;                 // 8 consecutive k of one 32-group: 16 B of its hi half, 16 B of its lo half
    *reinterpret_cast<u32x4*>(dst + 32) = lo;
}

// q16 [M][D] (A operand) and k16 [M][D] (W operand), K-interleaved rows (common.h ki_off), from the fp32 qkv [M][3 D] of the row
// attention's projection.  One thread = 8 consecutive columns of q or of k.
__global__ __launch_bounds__(256) void tied_prep_qk_kernel(const float* __restrict__ qkv, int64_t M, int D,
                                                           unsigned short* __restrict__ q16, unsigned short* __restrict__ k16) {
    const int per_row = D / 8;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * per_row * 2) return;
    const int which = (int)(i / (M * per_row));
    const int64_t u = i - (int64_t)which * M * per_row;
    const int64_t row = u / per_row;
    const int c = (int)(u % per_row) * 8;
    const float* src = qkv + row * 3 * D + (int64_t)which * D + c;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
    const float sc = which ? kTiedWScale : 1.0f;
    const float x[8] = {a[0] * sc, a[1] * sc, a[2] * sc, a[3] * sc, b[0] * sc, b[1] * sc, b[2] * sc, b[3] * sc};
    store_split8((which ? k16 : q16) + ki_off((size_t)row, c, D), x, which != 0);
}
void launch_tied_prep_qk(const float* qkv, int64_t M, int D, unsigned short* q16, unsigned short* k16, hipStream_t s) {
    const int64_t n = M * (D / 8) * 2;
    hipLaunchKernelGGL(tied_prep_qk_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, qkv, M, D, q16, k16);
}
float tied_w_scale() { return kTiedWScale; }

// Vt16[h][(r*64 + d)][j] (W operand of the update GEMM, K = j padded with zeros to Kp, K-interleaved) = v[r, j, h, d] 2^s
__global__ __launch_bounds__(256) void pack_vt16_kernel(const float* __restrict__ qkv, int R, int C, int Kp, int H,
                                                        unsigned short* __restrict__ Vt) {
    __shared__ float tile[64][65];
    const int jt = blockIdx.x, r = blockIdx.y, h = blockIdx.z;
    const int Da = H * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;             // 64 x 4
    for (int jj = ty; jj < 64; jj += 4) {
        const int j = jt * 64 + jj;
        tile[jj][tx] = (j < C) ? qkv[((size_t)r * C + j) * 3 * Da + 2 * Da + h * 64 + tx] * kTiedWScale : 0.0f;
    }
    __syncthreads();
    // thread (d, g): 8 consecutive j of row d, two passes cover the 64 j of the tile
    for (int it = 0; it < 2; ++it) {
        const int u = threadIdx.x + 256 * it, d = u >> 3, g = u & 7;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = tile[8 * g + e][d];
        store_split8(Vt + ki_off(((size_t)h * R + r) * 64 + d, jt * 64 + 8 * g, Kp), x, true);
    }
}
void launch_pack_vt16(const float* qkv, int R, int C, int Kp, int H, unsigned short* Vt, hipStream_t s) {
    hipLaunchKernelGGL(pack_vt16_kernel, dim3(Kp / 64, R, H), dim3(256), 0, s, qkv, R, C, Kp, H, Vt);
}

// ---- tied-attention softmax -------------------------------------------------------------------------
// part [H][S][C][Kp] split-K partial scores (fixed summation order: deterministic) ->
// P16 [H][C][Kp] = softmax_j<C(scale * sum_s part) as the update GEMM's A operand (activation split, K-interleaved), zeros in
// the padding columns j >= C.  axial_attention.py:72-74 (scaling = dh^-0.5 / sqrt(R); dh^-0.5 is folded into the q projection),
// :165 softmax.  One wave per (h, i) row; lane l owns columns 16 l .. 16 l + 15 (C <= 1024).
__global__ __launch_bounds__(256) void tied_softmax16_kernel(const float* __restrict__ part, int H, int S, int C, int Kp,
                                                             float scale, unsigned short* __restrict__ P) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);                  // (h, i)
    const int lane = threadIdx.x & 63;
    if (row >= H * C) return;
    const int h = row / C, i = row % C;
    const int j0 = lane * 16;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = 0.0f;
    if (j0 < Kp)
        for (int s = 0; s < S; ++s) {
            const float* src = part + (((size_t)h * S + s) * C + i) * Kp + j0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * q + e] += t[e];
            }
        }
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        v[e] *= scale;
        if (j0 + e < C) mx = fmaxf(mx, v[e]);
    }
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        v[e] = (j0 + e < C) ? expf(v[e] - mx) : 0.0f;
        sum += v[e];
    }
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    const float inv = 1.0f / sum;
    if (j0 < Kp) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float x[8] = {v[8 * g] * inv, v[8 * g + 1] * inv, v[8 * g + 2] * inv, v[8 * g + 3] * inv,
                                v[8 * g + 4] * inv, v[8 * g + 5] * inv, v[8 * g + 6] * inv, v[8 * g + 7] * inv};
            store_split8(P + ki_off((size_t)h * C + i, j0 + 8 * g, Kp), x, false);
        }
    }
}
int launch_tied_softmax16(const float* part, int H, int S, int C, int Kp, float scale, unsigned short* P, hipStream_t s) {
    if (C > 1024 || Kp > 1024 || (Kp % 64)) { set_error("tied row attention supports at most 1024 columns, got %d", C); return PGMI_EINVAL; }
    hipLaunchKernelGGL(tied_softmax16_kernel, dim3((H * C + 3) / 4), dim3(256), 0, s, part, H, S, C, Kp, scale, P);
    return PGMI_OK;
}

}  // namespace pgmi
