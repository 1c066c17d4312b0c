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

// ---- tied-attention softmax -------------------------------------------------------------------------
// part [H][S][C][Cp] split-K partial scores (fixed summation order: deterministic) ->
// P [H][C][Cp] = softmax_j<C(scale * sum_s part), zeros in the padding columns j >= C.
// axial_attention.py:72-74 (scaling = dh^-0.5 / sqrt(R); dh^-0.5 is folded into the q projection),
// :165 softmax.  One wave per (h, i) row; C <= 1024 -> at most 16 values per lane.
__global__ __launch_bounds__(256) void tied_softmax_kernel(const float* __restrict__ part, int H, int S, int C, int Cp,
                                                           float scale, float* __restrict__ P) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);                  // (h, i)
    const int lane = threadIdx.x & 63;
    if (row >= H * C) return;
    const int h = row / C, i = row % C;
    float v[16];
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int j = e * 64 + lane;
        float acc = 0.0f;
        if (j < C) {
            for (int s = 0; s < S; ++s) acc += part[(((size_t)h * S + s) * C + i) * Cp + j];
            acc *= scale;
            mx = fmaxf(mx, acc);
        }
        v[e] = acc;
    }
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int j = e * 64 + lane;
        v[e] = (j < C) ? expf(v[e] - mx) : 0.0f;
        sum += v[e];
    }
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int j = e * 64 + lane;
        if (j < Cp) P[((size_t)h * C + i) * Cp + j] = v[e] * inv;
    }
}
int launch_tied_softmax(const float* part, int H, int S, int C, int Cp, float scale, float* P, hipStream_t s) {
    if (C > 1024 || Cp > 1024) { set_error("tied row attention supports at most 1024 columns, got %d", C); return PGMI_EINVAL; }
    hipLaunchKernelGGL(tied_softmax_kernel, dim3((H * C + 3) / 4), dim3(256), 0, s, part, H, S, C, Cp, scale, P);
    return PGMI_OK;
}

// ---- V transpose for the tied update: Vt[h][(r*64 + d)][j] = qkv[(r*C + j), 2*Da + h*64 + d], 0 for j >= C ----
__global__ __launch_bounds__(256) void pack_vt_kernel(const float* __restrict__ qkv, int R, int C, int Cp, int H,
                                                      float* __restrict__ Vt) {
    __shared__ float tile[64][65];
    const int jt = blockIdx.x, r = blockIdx.y, h = blockIdx.z;
    const int Da = H * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;             // 64 x 4
    for (int jj = ty; jj < 64; jj += 4) {
        const int j = jt * 64 + jj;
        tile[jj][tx] = (j < C) ? qkv[((size_t)r * C + j) * 3 * Da + 2 * Da + h * 64 + tx] : 0.0f;
    }
    __syncthreads();
    for (int d = ty; d < 64; d += 4) {
        const int j = jt * 64 + tx;
        if (j < Cp) Vt[(((size_t)h * R + r) * 64 + d) * Cp + j] = tile[tx][d];
    }
}
void launch_pack_vt(const float* qkv, int R, int C, int Cp, int H, float* Vt, hipStream_t s) {
    hipLaunchKernelGGL(pack_vt_kernel, dim3((Cp + 63) / 64, R, H), dim3(256), 0, s, qkv, R, C, Cp, H, Vt);
}

}  // namespace pgmi
