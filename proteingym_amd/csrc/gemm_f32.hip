// fp32 GEMM on the CDNA4 matrix cores: C = epi(A W^T + bias) (+ residual).
//
// Replaces nn.Linear on the ESM hot path (q/k/v/out projections, fc1+GELU, fc2+residual:
// /root/reference/proteingym/baselines/esm/esm/modules.py:134-140,
// esm/multihead_attention.py:258-261,394).  Parity-gated mode: operands stay fp32 and go
// through v_mfma_f32_32x32x2_f32, whose result is bit-for-bit an fmaf chain over k
// (cdna_hip_programming.md section 3), i.e. plain fp32 arithmetic like the reference's CPU
// sgemm, only in a different summation order.
//
// Tiling (wave64, one MFMA pipe per SIMD):
//   workgroup  128(M) x 128(N) x 32(K), 4 waves as 2x2, 2 workgroups per CU (2 waves/SIMD)
//   wave       64 x 64 = 2x2 MFMA tiles of 32x32 -> 64 accumulator registers
//   LDS        A and W tiles K-contiguous, row stride 36 floats (144 B): ds_read_b128 of 16
//              different rows hits 16 different 16-B bank slots (conflict-free), 16-B aligned
//   K order    lane (r, kh) reads 4 consecutive k at 8g+4kh..+3 for BOTH operands, so one
//              ds_read_b128 feeds four MFMAs (k pairs (8g+e, 8g+4+e)); any k permutation is a
//              valid dot-product order as long as A and W agree
//   pipeline   global->VGPR loads of tile t+1 are issued before the 64 MFMAs of tile t and
//              written to the other LDS buffer after them: one barrier per K tile
//   grid       1-D, XCD-aware: block ids are remapped so each XCD owns a contiguous range of
//              tiles, walked in groups of 8 M-tiles x all N-tiles so A panels and W panels are
//              re-used out of that XCD's L2
// Roofline: MFMA-bound (arithmetic intensity 128*128*2/((128+128)*4) = 32 FLOP/B per K step
// against a 157 TF / ~5 TB/s L2->LDS ridge); peak 157.3 TFLOP/s.
#include "common.h"

namespace pgmi {

constexpr int BM = 128, BN = 128, BK = 32, LDS_STRIDE = 36;
constexpr int GROUP_M = 8;

__device__ __forceinline__ float gelu_erf(float x) {
    return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(
    const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
    const float* residual, float* C, int M, int N, int K, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BM * LDS_STRIDE];
    float* As = lds;                         // [2][BM][LDS_STRIDE]
    float* Bs = lds + 2 * BM * LDS_STRIDE;   // [2][BN][LDS_STRIDE]

    // --- XCD-aware, grouped tile order ---------------------------------------------------
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int q = nwg >> 3, r8 = nwg & 7;
    const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int width = GROUP_M * tiles_n;
    const int group = wgid / width;
    const int first_m = group * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tm = first_m + (wgid % width) % gsz;
    const int tn = (wgid % width) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, kh = lane >> 5;
    const int64_t lda = K, ldw = K;
    auto koff_a = [&](int kt) -> int64_t { return (int64_t)kt * BK; };     // element offset of K tile kt inside a row
    auto koff_w = [&](int kt) -> int64_t { return (int64_t)kt * BK; };

    // --- global -> register staging: 4 float4 of A and 4 of W per thread per K tile ------
    const float* a_src[4];
    const float* w_src[4];
    int lds_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;       // float4 index in the 128 x 8 tile
        const int row = f >> 3, c4 = f & 7;
        const int am = min(m0 + row, M - 1);
        const int wnr = min(n0 + row, N - 1);
        a_src[i] = A + (size_t)am * lda + c4 * 4;
        w_src[i] = W + (size_t)wnr * ldw + c4 * 4;
        lds_off[i] = row * LDS_STRIDE + c4 * 4;
    }
    f32x4 a_st[4], w_st[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_st[i] = *reinterpret_cast<const f32x4*>(a_src[i]);
        w_st[i] = *reinterpret_cast<const f32x4*>(w_src[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f32x4*>(As + lds_off[i]) = a_st[i];
        *reinterpret_cast<f32x4*>(Bs + lds_off[i]) = w_st[i];
    }
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.0f;

    const int a_frag = (wm * 64 + r) * LDS_STRIDE + kh * 4;
    const int b_frag = (wn * 64 + r) * LDS_STRIDE + kh * 4;
    const int nk = K / BK;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a_st[i] = *reinterpret_cast<const f32x4*>(a_src[i] + koff_a(kt + 1));
                w_st[i] = *reinterpret_cast<const f32x4*>(w_src[i] + koff_w(kt + 1));
            }
        }
        const float* Ab = As + cur * BM * LDS_STRIDE + a_frag;
        const float* Bb = Bs + cur * BN * LDS_STRIDE + b_frag;
        // Two-level sum: the 32 products of a K tile go through one MFMA chain that starts at 0 (`part`), and the tiles' partial
        // sums are added in fp32 afterwards.  One chain over the whole K (2 560 dependent roundings at K = 5120) put this mode
        // 2x further from exact arithmetic than the reference's blocked CPU GEMM on long sums (736-term pseudo-ppl sums, depth-5
        // multi-mutants: VERDICT r3); K / 32 + 16 roundings per element are closer to what a blocked sgemm does.  The 64 adds per
        // K tile issue under the 64 MFMAs of the next tile.
        f32x16 part[2][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDS_STRIDE + g * 8);
                bf[i] = *reinterpret_cast<const f32x4*>(Bb + i * 32 * LDS_STRIDE + g * 8);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        part[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e],
                                                                          (g == 0 && e == 0) ? f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f} : part[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][j][v] += part[i][j][v];
        if (more) {
            float* Aw = As + (cur ^ 1) * BM * LDS_STRIDE;
            float* Bw = Bs + (cur ^ 1) * BN * LDS_STRIDE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x4*>(Aw + lds_off[i]) = a_st[i];
                *reinterpret_cast<f32x4*>(Bw + lds_off[i]) = w_st[i];
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    // --- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (v&3)+8*(v>>2)+4*(lane>>5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + r;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * kh;
                if (m < M) {
                    float val = acc[i][j][v] + bv;
                    if (EPI == EPI_GELU) val = gelu_erf(val);
                    const size_t o = (size_t)m * N + n;
                    if (residual) val = residual[o] + val;
                    C[o] = val;
                }
            }
        }
    }
}

int launch_gemm_f32(const float* A, const float* W, const float* bias, const float* residual,
                    float* C, int M, int N, int K, int epilogue, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0) {
        set_error("gemm_f32: unsupported shape M=%d N=%d K=%d (K must be a multiple of %d)", M, N, K, BK);
        return PGMI_EINVAL;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const dim3 grid(tiles_m * tiles_n), block(256);
    if (epilogue == EPI_GELU)
        hipLaunchKernelGGL((gemm_f32_kernel<EPI_GELU>), grid, block, 0, s, A, W, bias, residual, C, M, N, K, tiles_m, tiles_n);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<EPI_NONE>), grid, block, 0, s, A, W, bias, residual, C, M, N, K, tiles_m, tiles_n);
    PGMI_HIP(hipGetLastError());
    return PGMI_OK;
}

}  // namespace pgmi
