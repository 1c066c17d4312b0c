// GEMMs on the 16-bit matrix cores: C = epi(A W^T * 2^-s + bias) (+ residual).
//
// Replaces nn.Linear on the ESM hot path (modules.py:134-140, multihead_attention.py:258-261, 394) in the f16x3
// (default, parity-gated) and bf16 (throughput, not parity-gated) modes.
//
// f16x3: every fp32 operand x is carried as two fp16 planes, hi = fp16(x) and a low part (weights:
// lo = fp16(x - hi); activations: lo = fp16((x - hi) 2^11), common.h split_act); the kernel accumulates
//     a_hi w_hi + a_hi w_lo + (a_lo 2^11)(w_hi 2^-11)
// in the fp32 MFMA accumulator (the dropped a_lo w_lo term is 2^-22 relative).  fp16 x fp16 products are exact in
// fp32, so the result has ~22 mantissa bits -- measured fp32-class on the full 33- and 36-layer models --
// at 3 MFMAs of the 2.5 PFLOP/s pipe per product block instead of 1 MFMA of the 157 TFLOP/s fp32 pipe.  Weights are
// pre-scaled by a per-tensor power of two 2^s (exact) so that their lo plane stays in fp16's normal range; 2^-s is
// applied in the epilogue.  Activations arrive already split (the producing LayerNorm / GELU / attention epilogue
// writes both planes: the same 4 bytes per element as fp32).
//
// Operand layout in HBM ("K-interleaved planes", common.h ki_off): a row of K elements is stored as K/32 groups of
// 128 bytes = 32 hi halfs followed by the 32 lo halfs of the same k.  One row's share of a 32-deep K tile is then ONE
// full 128-byte line: 8 consecutive lanes fetch it with one dwordx4 each.  With two separate planes (round 1) the same
// data was two 64-byte half lines, every line was requested twice (the other half one K tile later, long evicted from
// the 32 KB L1) and the texture-address unit handled twice the lines: measured with the phase-timing instantiation,
// 288 -> 325 TFLOP/s on the FC2 shape from the access pattern alone.
//
// Roofline: MFMA-bound; peak 2.5 PFLOP/s of 16-bit MFMA = 833 TFLOP/s of fp32-equivalent algorithmic FLOPs in f16x3.
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"

namespace pgmi {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// erf-GELU (modules.py:17-24, nn.GELU) without calling erff: 17 VALU ops instead of ~38 per value, which matters
// because the FC1 epilogue runs while the matrix pipe idles (18 % of FC1 at K = 1280, 30 % at K = 768).
//   gelu(x) = max(x,0) - 0.5|x| erfc(|x|/sqrt2),   erfc(z) ~= t Q(t) exp(-z^2),  t = 1/(1 + p z)
// (Abramowitz-Stegun 7.1.26 form, Q of degree 5 re-fitted by minimax to the product 0.5|x| erfc: fit error 3.7e-9).
// The erfc form has no 1 + erf cancellation for x < 0: against fp64 the fp32 evaluation is within 2.4e-7 abs
// (torch's fp32 gelu: 1.2e-6), mean 1.9e-8 (torch: 4.6e-8) -- scripts/fit_gelu.py.
__device__ __forceinline__ float gelu_erf16(float x) {
    const float ax = fabsf(x);
    const float z = fminf(ax * 0.70710678118654752440f, 13.0f);
    const float t = __builtin_amdgcn_rcpf(fmaf(3.973660903e-01f, z, 1.0f));
    float q = -2.134610164e-01f;
    q = fmaf(q, t, 7.781763039e-01f);
    q = fmaf(q, t, -4.744764895e-01f);
    q = fmaf(q, t, 5.422912625e-01f);
    q = fmaf(q, t, 1.330677808e-01f);
    q = fmaf(q, t, 2.344017484e-01f);
    const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
    return fmaf(-0.5f * ax, t * q * e, fmaxf(x, 0.0f));
}

// Attention operands straight from the fused QKV projection (OUT 2): q|k as split planes qk16 [2][M][2D] (ESM2 rotary
// applied here, rotary_embedding.py:11-20), v as the transposed, key-permuted planes vt16 [2][B*H*64][Tp] that
// attention_f16.hip consumes.
struct QkvOut {
    unsigned short* vt16;
    size_t vt_plane;
    const float* cos_t;
    const float* sin_t;
    int T, H, Tp, rotary;
    int rot_halves;       // rotary table rows per token: 1, or 2 for head_dim 128 (row = slot-group parity)
};

template <bool BF>
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (BF) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    }
}

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
    unsigned int u = __builtin_bit_cast(unsigned int, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// ---- bf16 (throughput mode, NOT parity-gated): plain bf16 operands, one MFMA per product block ----------------------
// Tiling (wave64): workgroup (WM*TM*32) x (WN*TN*32) x 32, WM*WN waves, each wave TM x TN MFMA tiles of 32x32
// (v_mfma_f32_32x32x16_bf16).  Operands are swapped (MFMA "A" = weight rows, "B" = activation rows) so that the
// accumulator puts 4 consecutive output columns in a lane: epilogue loads/stores are 8/16-byte vectors.  LDS tiles are
// K-contiguous 64-byte rows with an XOR swizzle on the 16-byte chunk (conflict-free ds_read_b128), double buffered,
// global -> VGPR -> LDS staging with the loads of tile t+1 in flight during the MFMAs of tile t; one barrier per K tile.
// OUT: 0 = fp32 [M,N]; 1 = one bf16 plane [M,N].
template <int WM, int WN, int TM, int TN, int EPI, int OUT>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_bf16_kernel(
    const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, const float* __restrict__ bias,
    const float* residual, float* Cf, unsigned short* Ch, int M, int N, int K, int tiles_m, int tiles_n) {
    constexpr int BK = 32, CPR = 4;
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_CH = BM * CPR, W_CH = BN * CPR;
    constexpr int A_LD = (A_CH + NT - 1) / NT, W_LD = (W_CH + NT - 1) / NT;
    constexpr int STAGE = A_CH + W_CH;
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];  // [2][STAGE]

    // XCD-aware grouped tile order (see gemm_f32.hip)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r8 = nwg & 7;
    const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    constexpr int GROUP_M = 8;
    const int width = GROUP_M * tiles_n;
    const int group = wgid / width, first_m = group * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tm = first_m + (wgid % width) % gsz, tn = (wgid % width) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, kh = lane >> 5;
    auto swz = [](int row, int c) -> int { return c ^ ((row >> 2) & 3); };

    const u32x4* a_src[A_LD];
    const u32x4* w_src[W_LD];
    int a_dst[A_LD], w_dst[W_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int f = tid + NT * i, row = (f % A_CH) / CPR, c = f % CPR;
        a_src[i] = reinterpret_cast<const u32x4*>(A + (size_t)min(m0 + row, M - 1) * K) + c;
        a_dst[i] = row * CPR + swz(row, c);
    }
#pragma unroll
    for (int i = 0; i < W_LD; ++i) {
        const int f = tid + NT * i, row = (f % W_CH) / CPR, c = f % CPR;
        w_src[i] = reinterpret_cast<const u32x4*>(W + (size_t)min(n0 + row, N - 1) * K) + c;
        w_dst[i] = A_CH + row * CPR + swz(row, c);
    }
    u32x4 a_st[A_LD], w_st[W_LD];
    auto stage_load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            if (A_CH % NT == 0 || tid + NT * i < A_CH) a_st[i] = a_src[i][kt * CPR];
#pragma unroll
        for (int i = 0; i < W_LD; ++i)
            if (W_CH % NT == 0 || tid + NT * i < W_CH) w_st[i] = w_src[i][kt * CPR];
    };
    auto stage_store = [&](int buf) {
        u32x4* base = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            if (A_CH % NT == 0 || tid + NT * i < A_CH) base[a_dst[i]] = a_st[i];
#pragma unroll
        for (int i = 0; i < W_LD; ++i)
            if (W_CH % NT == 0 || tid + NT * i < W_CH) base[w_dst[i]] = w_st[i];
    };
    stage_load(0);
    stage_store(0);
    __syncthreads();

    f32x16 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[j][i][v] = 0.0f;

    const int nk = K / BK;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) stage_load(kt + 1);
        const u32x4* Ab = lds + cur * STAGE;
        const u32x4* Wb = Ab + A_CH;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int c = ks * 2 + kh;
            u32x4 af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 32 + r;
                af[i] = Ab[row * CPR + swz(row, c)];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = (wn * TN + j) * 32 + r;
                wf[j] = Wb[row * CPR + swz(row, c)];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[j][i] = mfma16<true>(wf[j], af[i], acc[j][i]);
        }
        if (more) stage_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // ---- epilogue: lane holds column m = m_base + r of C^T, rows n = (v&3) + 8(v>>2) + 4kh ----
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + (wm * TM + i) * 32 + r;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + (wn * TN + j) * 32 + 8 * g + 4 * kh;
                if (n >= N) continue;                    // N % 4 == 0 is required by the launcher
                const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 val;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[j][i][4 * g + e] + bv[e];
                    if (EPI == EPI_GELU) t = gelu_erf16(t);
                    if (EPI == EPI_SQRELU) { t = fmaxf(t, 0.0f); t = t * t; }
                    val[e] = t;
                }
                const size_t o = (size_t)m * N + n;
                if (residual) {
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(residual + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = rv[e] + val[e];
                }
                if constexpr (OUT == 0) {
                    *reinterpret_cast<f32x4*>(Cf + o) = val;
                } else {
                    u32x2 pk;
                    pk[0] = f32_to_bf16_rne(val[0]) | ((unsigned)f32_to_bf16_rne(val[1]) << 16);
                    pk[1] = f32_to_bf16_rne(val[2]) | ((unsigned)f32_to_bf16_rne(val[3]) << 16);
                    *reinterpret_cast<u32x2*>(Ch + o) = pk;
                }
            }
        }
    }
}

template <int WM, int WN, int TM, int TN>
static int launch_bf16_cfg(const unsigned short* A, const unsigned short* W, const float* bias, const float* residual, float* Cf,
                           unsigned short* Ch, int M, int N, int K, int epilogue, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr size_t lds_bytes = (size_t)2 * (BM + BN) * 4 * 16;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const dim3 grid(tiles_m * tiles_n), block(WM * WN * 64);
#define PGMI_LAUNCH_BF16(EPI_, OUT_)                                                                     \
    do {                                                                                                 \
        auto kfn = gemm_bf16_kernel<WM, WN, TM, TN, EPI_, OUT_>;                                          \
        if (lds_bytes > 65536) {                                                                         \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PGMI_EHIP; } \
        }                                                                                                \
        hipLaunchKernelGGL(kfn, grid, block, lds_bytes, s, A, W, bias, residual, Cf, Ch, M, N, K, tiles_m, tiles_n); \
    } while (0)
    const int out = Ch ? 1 : 0;
    if (epilogue == EPI_GELU) { if (out) PGMI_LAUNCH_BF16(EPI_GELU, 1); else PGMI_LAUNCH_BF16(EPI_GELU, 0); }
    else if (epilogue == EPI_SQRELU) { if (out) PGMI_LAUNCH_BF16(EPI_SQRELU, 1); else PGMI_LAUNCH_BF16(EPI_SQRELU, 0); }
    else { if (out) PGMI_LAUNCH_BF16(EPI_NONE, 1); else PGMI_LAUNCH_BF16(EPI_NONE, 0); }
#undef PGMI_LAUNCH_BF16
    PGMI_HIP(hipGetLastError());
    return PGMI_OK;
}

// =================================================================================================
// Persistent ping-pong kernel (f16x3, 256 x 256 x 32 tile, 8 waves = 2(M) x 4(N), wave tile 128 x 64).
// One workgroup per CU walks a list of work items; per item the K loop is the ping-pong schedule of
// gemm16_kernel<PP> (the two waves of a SIMD run one phase apart: one issues its 24 MFMAs while the other reads
// fragments / stages the next K tile).  What the persistent form adds:
//   * no workgroup launch/retire gap between tiles, the output stores of tile i drain under the prologue and
//     main loop of tile i+1, and (STG 0) the first K tile of item i+1 is loaded into the staging registers
//     BEFORE the epilogue of item i runs;
//   * the tail of the launch is balanced: with T tiles on G workgroups the last partial round (T mod G tiles,
//     1610 tiles on 256 CUs = 6.29 rounds for the N = 1280 GEMMs) is cut into `split` K slices per tile, so every
//     CU gets a slice instead of 29 % of the chip working a full round.  Sliced items leave their raw fp32
//     accumulators in a workspace (fully coalesced 16-byte stores in accumulator order); splitk_fix_kernel adds
//     the slices in a fixed order (deterministic) and applies scale, bias and the residual.  Only the fp32-output
//     GEMMs (out-projection, FC2) use slices; their N = D gives the few tiles that make the tail matter.
// LDS image of a K tile (per operand): [256 rows][8 chunks of 16 B] = the row's 128-byte line (4 hi chunks, 4 lo chunks)
// with the chunk index XORed by (row >> 1) & 7: conflict-free ds_read_b128 fragment reads at a 128-byte row pitch, and
// both staging forms write whole rows.  STG 0: global -> VGPR -> LDS staging.  STG 1: global -> LDS DMA (the swizzle
// moves to the SOURCE chunk, which stays inside the row's line); the wave's share of tile kt+1 is issued in the first
// memory phase of tile kt and waited for at the LAST barrier of tile kt (two to three phases of latency cover).
// =================================================================================================
struct TilePlan {
    int tiles_m, tiles_n;
    int n_main;       // tiles [0, n_main) in launch order: one full-K item each
    int split;        // every later tile is cut into `split` K slices (<= 1: none)
    int n_items;      // n_main + (tiles - n_main) * split (or * 2 with half)
    int half;         // 1: every later tile is cut into its upper and lower 128 rows instead (two items, full K each)
    float* ws;        // raw accumulators of the sliced items
    unsigned long long* diag;   // DIAG instantiation only: [2 waves][kDiagSamples][2] shader-clock stamps (barrier arrival, release)
    int diag_flags;             // DIAG instantiation only (ablations, wrong numbers): 1 no global loads in the loop, 2 no
                                // ds_write staging, 4 loads issued in memory phase 1, 8 no fragment reads after the first K tile, 32 no s_setprio, 64 static
                                // priority 1 for the late waves only
};
constexpr int kDiagSamples = 1024;

constexpr int XBM = 256, XBN = 256, XNT = 512, XCPR = 8;   // 8 chunks = one 128-byte line per row and K tile (hi | lo)
constexpr int X_OP_CH = XBM * XCPR;                        // 16-byte chunks of one operand tile
constexpr int X_STAGE = 2 * X_OP_CH;                       // chunks per stage = 64 KB

__device__ __forceinline__ void x_tile_coords(int wgid, int tiles_m, int tiles_n, int& tm, int& tn) {
    constexpr int GROUP_M = 8;
    const int width = GROUP_M * tiles_n;
    const int group = wgid / width, first_m = group * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    tm = first_m + (wgid % width) % gsz;
    tn = (wgid % width) / gsz;
}

template <int EPI, int OUT, int STG, int DFLAGS = -1>     // DFLAGS >= 0: tuning-only phase-timing instantiation with these ablation flags
__global__ __launch_bounds__(XNT, 2) void gemm16x_kernel(
    const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, const float* __restrict__ bias,
    const float* residual, float* Cf, unsigned short* Ch, size_t c_plane, int M, int N, int K, float out_scale,
    TilePlan tp, QkvOut qo) {
    constexpr int WN = 4, TMX = 4, TN = 2, LD = 4;               // TMX: 32-row MFMA tiles per wave of a full item (a half item: 2)
    constexpr bool DIAG = DFLAGS >= 0;
    constexpr bool DMA = STG == 1 || STG == 3;                    // 3 (tuning): DMA issued after the fragment reads of memory phase 1
    constexpr int kFlags = DIAG ? DFLAGS % 1000 : 0;              // DFLAGS >= 1000: the DMA form (launched with STG 1)
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];  // [2][X_STAGE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // provably wave-uniform: scalar branches, SGPR LDS bases
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, kh = lane >> 5;
    const bool late = wave >= 4;                                  // the second wave of every SIMD

    // staging geometry (the same for both operands): slot f = tid + 512 i is (row (tid >> 3) + 64 i, chunk tid & 7) -- 8
    // consecutive lanes move one row's 128-byte line -- at LDS index row * 8 + (chunk ^ ((row >> 1) & 7)); the swizzle does
    // not depend on i, so one LDS index + 512 i serves all four, plus one 32-bit byte offset per row and operand.
    const int row_lo = tid >> 3, c8 = tid & 7, sw8 = (row_lo >> 1) & 7;
    const int dst0 = DMA ? tid : row_lo * XCPR + (c8 ^ sw8);
    const int csrc = DMA ? (c8 ^ sw8) : c8;               // DMA: lane-linear slot, swizzled SOURCE chunk
    // buffer descriptors built from kernel arguments only (provably wave-uniform): loads take a 32-bit per-lane byte offset
    // and the K-tile offset as an SGPR -- no 64-bit address arithmetic in the memory phases
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A), 0, (int)((unsigned int)M * (unsigned int)K * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, (int)((unsigned int)N * (unsigned int)K * 4u), 0x00020000);
    unsigned int a_off[LD], w_off[LD];
    u32x4 a_st[LD], w_st[LD];
    int m0 = 0, n0 = 0, kt0 = 0, kt1 = 0, slice_item = -1, a_ld = LD;
    bool half_item = false;                                       // the upper or lower 128 rows of a tile (tail of the item list)
    auto decode = [&](int item) {                                 // sets m0, n0, [kt0, kt1), slice_item and the source offsets
        const int nk = K / 32;
        int wgid;
        if (item < tp.n_main) {
            // XCD-aware order: item i runs on XCD i % 8 (grid size is a multiple of 8); every XCD walks a contiguous
            // run of the grouped tile order so that its L2 keeps the live A panels and W tiles
            const int nwg = tp.n_main, xcd = item & 7, q = nwg >> 3, r8 = nwg & 7;
            wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (item >> 3);
            kt0 = 0; kt1 = nk; slice_item = -1;
        } else if (tp.half) {
            wgid = tp.n_main + ((item - tp.n_main) >> 1);
            kt0 = 0; kt1 = nk; slice_item = -1;
        } else {
            const int rel = item - tp.n_main, ks = rel % tp.split;
            wgid = tp.n_main + rel / tp.split;
            kt0 = (int)((long long)nk * ks / tp.split);
            kt1 = (int)((long long)nk * (ks + 1) / tp.split);
            slice_item = rel;
        }
        int tm, tn;
        x_tile_coords(wgid, tp.tiles_m, tp.tiles_n, tm, tn);
        half_item = tp.half && item >= tp.n_main;
        a_ld = half_item ? LD / 2 : LD;                           // A rows staged per K tile: 64 per instruction
        m0 = __builtin_amdgcn_readfirstlane(tm * XBM + (half_item ? ((item - tp.n_main) & 1) * (XBM / 2) : 0));
        n0 = __builtin_amdgcn_readfirstlane(tn * XBN);
        kt0 = __builtin_amdgcn_readfirstlane(kt0); kt1 = __builtin_amdgcn_readfirstlane(kt1);
        slice_item = __builtin_amdgcn_readfirstlane(slice_item);
#pragma unroll
        for (int i = 0; i < LD; ++i) {                            // row pitch 4 K bytes; the launcher guarantees rows * 4 K < 4 GiB
            a_off[i] = (unsigned int)min(m0 + row_lo + 64 * i, M - 1) * (unsigned int)K * 4u + (unsigned int)csrc * 16u;
            w_off[i] = (unsigned int)min(n0 + row_lo + 64 * i, N - 1) * (unsigned int)K * 4u + (unsigned int)csrc * 16u;
        }
    };
    auto stage_load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < LD; ++i) a_st[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)a_off[i], kt * 128, 0));
#pragma unroll
        for (int i = 0; i < LD; ++i) w_st[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)w_off[i], kt * 128, 0));
    };
    auto stage_store = [&](int buf) {
        u32x4* base = lds + buf * X_STAGE + dst0;
#pragma unroll
        for (int i = 0; i < LD; ++i) base[XNT * i] = a_st[i];
#pragma unroll
        for (int i = 0; i < LD; ++i) base[X_OP_CH + XNT * i] = w_st[i];
    };
    auto issue_tile = [&](int kt, int buf) {                      // STG 1: 1 KiB per wave-instruction, wave-uniform LDS base
        u32x4* base = lds + buf * X_STAGE + wave * 64;
#pragma unroll
        for (int i = 0; i < LD; ++i)
            if (i < a_ld)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + XNT * i), 16, (int)a_off[i], kt * 128, 0, 0);
#pragma unroll
        for (int i = 0; i < LD; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(base + X_OP_CH + XNT * i), 16, (int)w_off[i], kt * 128, 0, 0);
    };
    // phase boundary: everything issued before stays before, this wave's LDS traffic has landed; VMEM stays in flight
    // DIAG (tuning-only instantiation): waves 0 and 4 of workgroup 0 stamp the shader clock when they ARRIVE at every phase
    // barrier, with a tag saying what the phase was (0 mem1, 1 cmp1, 3 mem2, 4 cmp2, 2 other).  The stamp is an SMEM read
    // issued before the barrier and consumed at the next one: a wave that waits at the barrier pays nothing for it; the wave
    // that arrives last (a computing wave) pays its latency at the start of its next -- memory -- phase.  (A second stamp
    // after the barrier made every instrumented compute phase wait for the SMEM round trip: +200-350 clocks.)  The host
    // takes release(k) = max over the two waves of arrival(k).
    const bool diag_on = DIAG && blockIdx.x == 0 && (wave == 0 || wave == 4);
    unsigned long long d_arr = 0;
    int d_n = 0, d_tag = 0;
    auto phase_impl = [&](bool vm, int tag) {
        __builtin_amdgcn_sched_barrier(0);
        if (DIAG && diag_on) {
            if (d_n > 0 && d_n <= kDiagSamples && lane == 0) {
                unsigned long long* q = tp.diag + ((size_t)(wave >> 2) * kDiagSamples + (d_n - 1)) * 2;
                q[0] = d_arr; q[1] = (unsigned long long)d_tag + 1;
            }
        }
        if (vm && !(DIAG && (kFlags & 128))) __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0xC07F);                                    // lgkmcnt(0)
        if (DIAG && diag_on) { d_arr = __builtin_readcyclecounter(); d_tag = tag; ++d_n; }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto phase = [&](int tag = 2) { phase_impl(false, tag); };
    auto phase_vm = [&](int tag = 2) { phase_impl(true, tag); };   // the same, plus this wave's DMA has landed (DMA form)

    f32x16 acc[TN][TMX];
    u32x4 af[2][TMX], wf[2][TN], whs[TN];
    const int fsw = (r >> 1) & 7;                                 // fragment rows are (multiple of 32) + r: the row swizzle is per lane
    auto read_frags = [&](auto tmc, const u32x4* Ab, const u32x4* Wb, int ks) {
        constexpr int TM = decltype(tmc)::value;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int c = (p * 4 + ks * 2 + kh) ^ fsw;            // chunk p*4 + (2 ks + kh) of the row's line, swizzled
#pragma unroll
            for (int i = 0; i < TM; ++i) af[p][i] = Ab[((wm * TM + i) * 32 + r) * XCPR + c];
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[p][j] = Wb[((wn * TN + j) * 32 + r) * XCPR + c];
        }
    };
    // w_hi 2^-11 (exact: weights are pre-scaled to ~2^13), the third operand of the split product.  Computed at the head of
    // the wave's own COMPUTE phase, interleaved with the first MFMAs (which do not need it): in the memory phase the partner
    // wave holds priority and these eight VALU ops sat on the critical path to the phase barrier (measured: a memory phase
    // with nothing but them still took 600-700 clocks).
    auto scale_whi = [&]() {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 sc = {(_Float16)(1.0f / kLoScale), (_Float16)(1.0f / kLoScale)};
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned int u = wf[0][j][e];
                const h2 t = __builtin_bit_cast(h2, u) * sc;
                whs[j][e] = __builtin_bit_cast(unsigned int, t);
            }
    };
    auto mfmas = [&](auto tmc) {
        constexpr int TM = decltype(tmc)::value;
        scale_whi();
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = mfma16<false>(wf[1][j], af[0][i], acc[j][i]);    // w_lo a_hi
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = mfma16<false>(whs[j], af[1][i], acc[j][i]);      // (w_hi 2^-11)(a_lo 2^11)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = mfma16<false>(wf[0][j], af[0][i], acc[j][i]);    // w_hi a_hi
        // one MFMA, one VALU: the eight scalings ride in the shadow of the first eight MFMAs
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * TN * TM - 8, 0);
    };

    int item = blockIdx.x;
    if (item >= tp.n_items) return;
    bool dma_ahead = false;                                       // the item's first K tile was issued before the previous epilogue
    decode(item);
    if constexpr (STG == 0) stage_load(kt0);
    // the fp32-output epilogue has the registers to spare for the next item's first K tile; the split-plane and
    // attention-operand epilogues do not (the prefetch spilled 12-23 registers): they fetch after the epilogue
    constexpr bool kPrefetchAcrossEpilogue = STG == 0 && OUT == 0;
    // one item, start to finish; TM (compile time) = 32-row MFMA tiles per wave: 4, or 2 for a half item.  Returns "more items".
    auto run_item = [&](auto tmc) -> bool {
        constexpr int TM = decltype(tmc)::value;
        // ---- prologue: first K tile of the item into buffer 0 (the LDS patches of the previous epilogue are done:
        //      every wave passed the barrier below only after finishing its own patch reads) ----
        __syncthreads();
        if constexpr (STG == 0) {
            stage_store(0);
            if (kt0 + 1 < kt1) stage_load(kt0 + 1);
            __syncthreads();
        } else {
            if (!dma_ahead) issue_tile(kt0, 0);
            phase_vm();
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[j][i][v] = 0.0f;
        int cur = 0;
        if (DIAG && (kFlags & 64) && late) __builtin_amdgcn_s_setprio(1);
        if (late) phase();
        for (int kt = kt0; kt < kt1; ++kt) {
            const u32x4* Ab = lds + cur * X_STAGE;
            const u32x4* Wb = Ab + X_OP_CH;
            // -- memory phase 1: fragments of the first k16 step (STG 1: DMA of tile kt+1 into the other buffer, last read
            //    two phases ago by the other wave group) --
            if (!(DIAG && (kFlags & 96))) __builtin_amdgcn_s_setprio(0);
            // the DMA goes first: with the fragment reads ahead of it the memory phase outlasts the partner's 24 MFMAs
            // (tools/mfma_phase.hip: 829 -> 797 clocks per phase on the bare loop); STG 3 keeps the old order for A/B
            if (STG == 1 && kt + 1 < kt1 && !(DIAG && (kFlags & 16) && kt > kt0)) issue_tile(kt + 1, cur ^ 1);
            if (!(DIAG && (kFlags & 8) && kt > kt0)) read_frags(tmc, Ab, Wb, 0);
            if (STG == 3 && kt + 1 < kt1) issue_tile(kt + 1, cur ^ 1);
            if (DIAG && (kFlags & 4) && kt > kt0 && kt + 1 < kt1) stage_load(kt + 1);
            phase(0);
            // -- compute phase 1 --
            if (!(DIAG && (kFlags & 96))) __builtin_amdgcn_s_setprio(1);
            mfmas(tmc);
            phase(1);
            // -- memory phase 2: fragments of the second k16 step; STG 0: tile kt+1 registers -> LDS, loads of kt+2 --
            if (!(DIAG && (kFlags & 96))) __builtin_amdgcn_s_setprio(0);
            if (!(DIAG && (kFlags & 8) && kt > kt0)) read_frags(tmc, Ab, Wb, 1);
            if (STG == 0 && kt + 1 < kt1) {
                __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0); lgkmcnt / expcnt untouched
                if (!(DIAG && (kFlags & 2))) stage_store(cur ^ 1);
                if (kt + 2 < kt1 && !(DIAG && (kFlags & 5))) stage_load(kt + 2);
            }
            if (DMA && late) phase_vm(3); else phase(3);        // late waves close tile kt here: their DMA share must have landed
            // -- compute phase 2 --
            if (!(DIAG && (kFlags & 96))) __builtin_amdgcn_s_setprio(1);
            mfmas(tmc);
            if (DMA && !late) phase_vm(4); else phase(4);       // early waves close tile kt here
            cur ^= 1;
        }
        __builtin_amdgcn_s_setprio(0);
        if (!late) phase();

        // ---- the item just finished; fetch the next one's first K tile before the epilogue (STG 0) ----
        const int em0 = m0, en0 = n0, eslice = slice_item;
        item += gridDim.x;
        const bool more = item < tp.n_items;
        if (more) {
            decode(item);
            if constexpr (kPrefetchAcrossEpilogue) stage_load(kt0);
            // DMA form: the fp32-output epilogue does not touch LDS and both K-tile buffers are free after the last phase barrier,
            // so the next item's first K tile is already in flight while this item's epilogue runs
            if constexpr (DMA && OUT == 0) { issue_tile(kt0, 0); dma_ahead = true; }
        }

        if (eslice >= 0) {
            // raw accumulators, accumulator order: [item][wave][j][i][q][lane] f32x4 -> 1 KiB per store instruction
            f32x4* dst = reinterpret_cast<f32x4*>(tp.ws) + ((size_t)eslice * 8 + wave) * (TN * TM * 4 * 64) + lane;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        dst[((j * TM + i) * 4 + q4) * 64] = f32x4{acc[j][i][4 * q4], acc[j][i][4 * q4 + 1], acc[j][i][4 * q4 + 2], acc[j][i][4 * q4 + 3]};
        } else if constexpr (OUT == 2) {
            const int Dm = N / 3;
            const int nb = en0 + wn * 64;                      // first column of this wave's head
            if (nb < N) {
            const int which = nb / Dm, hcol = nb - which * Dm, hh = hcol >> 6;
            constexpr int SPQ = 144;
            unsigned char* patch_q = reinterpret_cast<unsigned char*>(lds) + wave * (2 * 32 * SPQ);
            const bool staged_q = which < 2;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = em0 + (wm * TM + i) * 32 + r;
                const bool row_ok = m < M;
                if (!staged_q && !row_ok) continue;
                const int bb = m / qo.T, t = m - bb * qo.T;
                if (row_ok)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = 8 * g + 4 * kh;             // dims d0..d0+3 (x0) and d0+32.. (x1) of the head
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + nb + d0);
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + nb + 32 + d0);
                    float x0[4], x1[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x0[e] = acc[0][i][4 * g + e] * out_scale + b0[e];
                        x1[e] = acc[1][i][4 * g + e] * out_scale + b1[e];
                    }
                    if (which == 0) {                          // attention's softmax is base 2: q carries log2(e) (common.h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { x0[e] *= kQLog2e; x1[e] *= kQLog2e; }
                    }
                    if (which < 2) {
                        if (qo.rotary) {                       // rotary_embedding.py:11-20
                            const int tr = (t * qo.rot_halves + (hh % qo.rot_halves)) * 64;
                            const f32x4 c1 = *reinterpret_cast<const f32x4*>(qo.cos_t + tr + d0);
                            const f32x4 s1 = *reinterpret_cast<const f32x4*>(qo.sin_t + tr + d0);
                            const f32x4 c2 = *reinterpret_cast<const f32x4*>(qo.cos_t + tr + 32 + d0);
                            const f32x4 s2 = *reinterpret_cast<const f32x4*>(qo.sin_t + tr + 32 + d0);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float y0 = x0[e] * c1[e] + (-x1[e]) * s1[e];
                                const float y1 = x1[e] * c2[e] + x0[e] * s2[e];
                                x0[e] = y0;
                                x1[e] = y1;
                            }
                        }
                        h4 hi0, lo0, hi1, lo1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            _Float16 a, b2;
                            split_act(x0[e], a, b2); hi0[e] = a; lo0[e] = b2;
                            split_act(x1[e], a, b2); hi1[e] = a; lo1[e] = b2;
                        }
                        unsigned char* cell = patch_q + r * SPQ + d0 * 2;
                        *reinterpret_cast<h4*>(cell) = hi0;
                        *reinterpret_cast<h4*>(cell + 32 * SPQ) = lo0;
                        *reinterpret_cast<h4*>(cell + 64) = hi1;
                        *reinterpret_cast<h4*>(cell + 32 * SPQ + 64) = lo1;
                    } else {
                        const int tk = t & 31;
                        const int pos = (t & ~31) + ((tk & 0x13) | ((tk & 4) << 1) | ((tk & 8) >> 1));   // swap key bits 2,3
                        unsigned short* col = qo.vt16 + (((size_t)bb * qo.H + hh) * kHeadDim + d0) * qo.Tp + pos;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            _Float16 a0, l0, a1, l1;
                            split_act(x0[e], a0, l0);
                            split_act(x1[e], a1, l1);
                            unsigned short* c0 = col + (size_t)e * qo.Tp;
                            unsigned short* c1p = c0 + (size_t)32 * qo.Tp;
                            c0[0] = __builtin_bit_cast(unsigned short, a0);
                            c0[qo.vt_plane] = __builtin_bit_cast(unsigned short, l0);
                            c1p[0] = __builtin_bit_cast(unsigned short, a1);
                            c1p[qo.vt_plane] = __builtin_bit_cast(unsigned short, l1);
                        }
                    }
                }
                if (staged_q) {
                    __builtin_amdgcn_wave_barrier();
                    const int m_base = em0 + (wm * TM + i) * 32;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int q = lane + 64 * k, row = q >> 3, cc = q & 7;
                        const u32x4 vh = *reinterpret_cast<const u32x4*>(patch_q + row * SPQ + cc * 16);
                        const u32x4 vl = *reinterpret_cast<const u32x4*>(patch_q + (32 + row) * SPQ + cc * 16);
                        if (m_base + row < M) {
                            unsigned short* dst = Ch + (size_t)(m_base + row) * (2 * Dm) + (size_t)which * Dm + hcol + cc * 8;
                            *reinterpret_cast<u32x4*>(dst) = vh;
                            *reinterpret_cast<u32x4*>(dst + c_plane) = vl;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            }
        } else {
            // lane holds column m = m_base + r of C^T, rows n = (v&3) + 8(v>>2) + 4kh; split-plane output leaves through a
            // per-wave LDS transpose as full 128-byte row segments (see gemm16_kernel)
            // per-wave LDS patch: 32 rows x (256 B in OUTPUT order: group 0 hi | group 0 lo | group 1 hi | group 1 lo) + 16 B pad
            constexpr int SP = 272;
            const bool staged = (OUT == 1) && (N % 8 == 0) && (en0 + (wn * TN + TN) * 32 <= N);
            unsigned char* patch = reinterpret_cast<unsigned char*>(lds) + wave * (32 * SP);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = em0 + (wm * TM + i) * 32 + r;
                const bool row_ok = m < M;
                if (!staged && !row_ok) continue;
                if (row_ok)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = en0 + (wn * TN + j) * 32 + 8 * g + 4 * kh;
                        if (n >= N) continue;                    // N % 4 == 0 is required by the launcher
                        const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                        f32x4 val;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = acc[j][i][4 * g + e] * out_scale + bv[e];
                            if (EPI == EPI_GELU) t = gelu_erf16(t);
                            if (EPI == EPI_SQRELU) { t = fmaxf(t, 0.0f); t = t * t; }     // tranception/activations.py:79-84
                            val[e] = t;
                        }
                        const size_t o = (size_t)m * N + n;
                        if constexpr (OUT == 0) {
                            if (residual) {
                                const f32x4 rv = *reinterpret_cast<const f32x4*>(residual + o);
#pragma unroll
                                for (int e = 0; e < 4; ++e) val[e] = rv[e] + val[e];
                            }
                            *reinterpret_cast<f32x4*>(Cf + o) = val;
                        } else {
                            h4 hi, lo;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                _Float16 a, b;
                                split_act(val[e], a, b);
                                hi[e] = a;
                                lo[e] = b;
                            }
                            if (staged) {
                                unsigned char* cell = patch + r * SP + j * 128 + (8 * g + 4 * kh) * 2;
                                *reinterpret_cast<h4*>(cell) = hi;
                                *reinterpret_cast<h4*>(cell + 64) = lo;
                            } else {
                                unsigned short* dst = Ch + ki_off((size_t)m, n, N);
                                *reinterpret_cast<h4*>(dst) = hi;
                                *reinterpret_cast<h4*>(dst + 32) = lo;
                            }
                        }
                    }
                }
                if (OUT == 1 && staged) {
                    // K-interleaved output: the wave's 64 columns are two 32-column groups = 2 x (64 B hi | 64 B lo) = 256
                    // contiguous bytes per row: 16 lanes write one row
                    __builtin_amdgcn_wave_barrier();
                    const int m_base = em0 + (wm * TM + i) * 32;
                    const size_t ncol0 = (size_t)en0 + (size_t)wn * TN * 32;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int q = lane + 64 * k;                 // 16-byte chunk: row q/16, chunk q%16 of the 256-byte run
                        const int row = q >> 4, cc = q & 15;
                        const u32x4 v = *reinterpret_cast<const u32x4*>(patch + row * SP + cc * 16);
                        if (m_base + row < M)
                            *reinterpret_cast<u32x4*>(Ch + (size_t)(m_base + row) * (2 * (size_t)N) + ncol0 * 2 + cc * 8) = v;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        return more;
    };
    using Full = std::integral_constant<int, TMX>;
    using Half = std::integral_constant<int, TMX / 2>;
    while (true) {
        bool more;
        if constexpr (OUT != 2 && DMA && !DIAG) more = half_item ? run_item(Half{}) : run_item(Full{});
        else more = run_item(Full{});
        if (!more) break;
        if constexpr (STG == 0 && !kPrefetchAcrossEpilogue) stage_load(kt0);
    }
}

// Sum of the K slices of the sliced tiles (fixed order: deterministic) + the fp32 epilogue of gemm16x_kernel<EPI_NONE, 0>:
// C = residual + (sum acc) * out_scale + bias.  One workgroup per sliced tile, same thread <-> element map as the GEMM.
__global__ __launch_bounds__(XNT) void splitk_fix_kernel(const float* __restrict__ ws, int split, int n_main, int tiles_m, int tiles_n,
                                                         const float* __restrict__ bias, const float* residual, float* Cf,
                                                         int M, int N, float out_scale) {
    constexpr int WN = 4, TM = 4, TN = 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, r = lane & 31, kh = lane >> 5;
    int tm, tn;
    x_tile_coords(n_main + blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * XBM, n0 = tn * XBN;
    const f32x4* src = reinterpret_cast<const f32x4*>(ws) + ((size_t)blockIdx.x * split * 8 + wave) * (TN * TM * 4 * 64) + lane;
    const size_t slice_stride = (size_t)8 * (TN * TM * 4 * 64);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + (wm * TM + i) * 32 + r;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + (wn * TN + j) * 32 + 8 * g + 4 * kh;
                f32x4 a = src[((j * TM + i) * 4 + g) * 64];
                for (int s = 1; s < split; ++s) {
                    const f32x4 b = src[s * slice_stride + ((j * TM + i) * 4 + g) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] += b[e];
                }
                if (m >= M || n >= N) continue;
                const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                const size_t o = (size_t)m * N + n;
                f32x4 val;
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = a[e] * out_scale + bv[e];
                if (residual) {
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(residual + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = rv[e] + val[e];
                }
                *reinterpret_cast<f32x4*>(Cf + o) = val;
            }
        }
}

// Per-device launch state (a process may drive several devices through distinct model handles: nothing here is shared
// between devices).  Indexed by the current HIP device of the calling thread; written under a mutex because two handles on two
// devices may launch from two host threads.
constexpr int kMaxDevices = 64;
struct XDeviceState {
    int num_cus = 0;
    float* splitk_ws = nullptr;                                   // K-sliced tail workspace (PGMI_GEMM_SPLITK=1 only)
    size_t splitk_ws_bytes = 0;
};
static XDeviceState g_xdev[kMaxDevices];
static std::mutex g_xdev_mu;

static int x_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    return dev;
}

static int x_num_cus() {
    const int dev = x_current_device();
    std::lock_guard<std::mutex> lk(g_xdev_mu);
    int& n = g_xdev[dev].num_cus;
    if (!n) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
        n -= n % 8;                                              // the XCD-aware item order wants a multiple of 8
        if (n <= 0) n = 8;
    }
    return n;
}

// stg: 0 register staging, 1 direct-to-LDS DMA, 3 the same with the DMA issued after the fragment reads (tuning).  splitk: allow K-sliced tail items (fp32-output GEMMs only).
static int launch_gemm16x_one(const unsigned short* A, const unsigned short* W,
                              const float* bias, const float* residual, float* Cf, unsigned short* Ch, size_t c_plane,
                              int M, int N, int K, int epilogue, float out_scale, int stg, bool splitk, hipStream_t s,
                              const QkvOut* qkv) {
    TilePlan tp{};
    tp.tiles_m = (M + XBM - 1) / XBM;
    tp.tiles_n = (N + XBN - 1) / XBN;
    const int T = tp.tiles_m * tp.tiles_n, G = x_num_cus(), nk = K / 32;
    tp.n_main = T; tp.split = 1; tp.n_items = T;
    if ((unsigned long long)std::max(M, N) * (unsigned long long)K * 4ull >= (1ull << 32)) {      // launch_gemm16x chunks M below this
        set_error("gemm16x: operand of %d x %d split elements exceeds the 32-bit offset range", std::max(M, N), K);
        return PGMI_EINVAL;
    }
    const int rem = T % G;
    // K-sliced tails are OPT-IN (PGMI_GEMM_SPLITK=1): a sliced tile adds its K range in a different association than a full
    // tile, so a row's bits would depend on how many rows travel with it -- and the scorer guarantees that every way of
    // batching an assay (whole, position chunks, any number of GPUs) gives bit-identical scores (tests/test_gpu_cli.py).
    // Measured gain when enabled: FC2 386 -> 402 TFLOP/s at the BLAT shape (6.29 -> 6.33 rounds of tiles), +1.2 % per step.
    const int want_split = getenv("PGMI_GEMM_SPLITK") ? atoi(getenv("PGMI_GEMM_SPLITK")) : 0;
    if (splitk && want_split && Cf && !qkv && epilogue == EPI_NONE && rem > 0 && 4 * rem <= 3 * G) {
        int split = std::min(std::min(G / rem, 8), nk / 4);       // every slice keeps >= 4 K tiles
        if (split >= 2) {
            const size_t need = (size_t)rem * split * XBM * XBN * sizeof(float);
            float* ws = nullptr;
            {
                std::lock_guard<std::mutex> lk(g_xdev_mu);
                XDeviceState& st = g_xdev[x_current_device()];
                if (need > st.splitk_ws_bytes) {
                    if (st.splitk_ws) hipFree(st.splitk_ws);
                    st.splitk_ws = nullptr; st.splitk_ws_bytes = 0;
                    const size_t cap = std::max(need, (size_t)G * XBM * XBN * sizeof(float));
                    if (hipMalloc(reinterpret_cast<void**>(&st.splitk_ws), cap) == hipSuccess) st.splitk_ws_bytes = cap;
                }
                ws = st.splitk_ws;
            }
            if (ws) {
                tp.n_main = T - rem; tp.split = split; tp.n_items = tp.n_main + rem * split; tp.ws = ws;
            }
        }
    }
    // Half-height tail (every output kind but the fused QKV, DMA form): the last, partial round of tiles leaves G - rem CUs idle for a whole tile
    // time (N = 1280 at the BLAT shape: 6.29 rounds cost 7).  Its tiles are cut into their upper and lower 128 rows -- two
    // items on two CUs, each over the full K range in the same order, so every output element is computed exactly as in a
    // full tile (bit-identical; unlike the K slices above) -- when all the halves still fit one round.
    const int want_half = getenv("PGMI_GEMM_HALF_TAIL") ? atoi(getenv("PGMI_GEMM_HALF_TAIL")) : 1;   // read per launch: the tests toggle it
    if (want_half && tp.split <= 1 && !qkv && (stg == 1 || stg == 3) && rem > 0 && 2 * rem <= G) {
        tp.n_main = T - rem; tp.half = 1; tp.n_items = tp.n_main + 2 * rem;
    }
    QkvOut qo{};
    if (qkv) qo = *qkv;
    const size_t lds_bytes = (size_t)2 * X_STAGE * 16;
    const dim3 grid(std::min(G, tp.n_items)), block(XNT);
    if (stg == 2) {                                               // tuning only: phase-timing instantiation (fp32-out GEMM)
        if (!Cf || qkv || epilogue != EPI_NONE) { set_error("gemm16x diag: fp32-output GEMM without activation only"); return PGMI_EINVAL; }
        static unsigned long long* dbuf = nullptr;
        const size_t n = (size_t)2 * kDiagSamples * 2;
        if (!dbuf && hipMalloc(reinterpret_cast<void**>(&dbuf), n * 8) != hipSuccess) { set_error("diag alloc failed"); return PGMI_ENOMEM; }
        PGMI_HIP(hipMemsetAsync(dbuf, 0, n * 8, s));
        tp.diag = dbuf;
        tp.diag_flags = getenv("PGMI_GEMM_DIAG_FLAGS") ? atoi(getenv("PGMI_GEMM_DIAG_FLAGS")) : 0;
#define PGMI_DIAG_CASE(F_)                                                                                          \
        case F_: {                                                                                                  \
            auto kfn = gemm16x_kernel<EPI_NONE, 0, (F_ >= 1000 ? 1 : 0), F_>;                                       \
            PGMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
            hipLaunchKernelGGL(kfn, grid, block, lds_bytes, s, A, W, bias, residual, Cf, Ch, c_plane, M, N, K, out_scale, tp, qo); \
        } break;
        switch (tp.diag_flags) {
            // register staging: 0 as shipped, 11 no loads / LDS writes / fragment reads.  DMA form (+1000): 1000 as shipped, 1016 no DMA
            // after the first K tile, 1008 no fragment reads, 1024 neither, 1128 no wait for the DMA (wrong numbers, timing only)
            PGMI_DIAG_CASE(0) PGMI_DIAG_CASE(11) PGMI_DIAG_CASE(1000) PGMI_DIAG_CASE(1016) PGMI_DIAG_CASE(1008) PGMI_DIAG_CASE(1024) PGMI_DIAG_CASE(1128)
            default: set_error("gemm16x diag: flags %d not instantiated", tp.diag_flags); return PGMI_EINVAL;
        }
#undef PGMI_DIAG_CASE
        if (tp.split > 1)
            hipLaunchKernelGGL(splitk_fix_kernel, dim3(T - tp.n_main), dim3(XNT), 0, s, tp.ws, tp.split, tp.n_main, tp.tiles_m,
                               tp.tiles_n, bias, residual, Cf, M, N, out_scale);
        std::vector<unsigned long long> h(n);
        PGMI_HIP(hipMemcpyAsync(h.data(), dbuf, n * 8, hipMemcpyDeviceToHost, s));
        PGMI_HIP(hipStreamSynchronize(s));
        // per wave: work = release(k-1) -> arrival(k), wait = arrival(k) -> release(k), release(k) = the later arrival of the two
        static int printed = 0;
        if (!printed++) {
            fprintf(stderr, "[gemm16x diag] flags %d\n", tp.diag_flags);
            const unsigned long long* q0 = h.data();
            const unsigned long long* q1 = h.data() + (size_t)kDiagSamples * 2;
            for (int g = 0; g < 2; ++g) {
                const unsigned long long* q = g ? q1 : q0;
                double work[5] = {0, 0, 0, 0, 0}, wait[5] = {0, 0, 0, 0, 0};
                int cnt[5] = {0, 0, 0, 0, 0};
                for (int k = 16; k < kDiagSamples && q0[2 * k + 1] && q1[2 * k + 1]; ++k) {     // skip the pipeline fill
                    const int tag = (int)q[2 * k + 1] - 1;
                    const unsigned long long rel_prev = std::max(q0[2 * (k - 1)], q1[2 * (k - 1)]);
                    const unsigned long long rel = std::max(q0[2 * k], q1[2 * k]);
                    if (tag < 0 || tag > 4 || q[2 * k] < rel_prev) continue;
                    work[tag] += (double)(q[2 * k] - rel_prev);
                    wait[tag] += (double)(rel - q[2 * k]);
                    cnt[tag]++;
                }
                fprintf(stderr, "[gemm16x diag] %s waves: ", g ? "late " : "early");
                static const char* nm[5] = {"mem1", "cmp1", "other", "mem2", "cmp2"};
                for (int p : {0, 1, 3, 4})
                    fprintf(stderr, "%s work %.0f wait %.0f | ", nm[p], cnt[p] ? work[p] / cnt[p] : 0.0, cnt[p] ? wait[p] / cnt[p] : 0.0);
                fprintf(stderr, "(shader clocks, mean over %d K tiles)\n", cnt[0]);
            }
        }
        return PGMI_OK;
    }
#define PGMI_LAUNCH16X(EPI_, OUT_, STG_)                                                                  \
    do {                                                                                                 \
        auto kfn = gemm16x_kernel<EPI_, OUT_, STG_>;                                                      \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                           \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);  \
        if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PGMI_EHIP; } \
        hipLaunchKernelGGL(kfn, grid, block, lds_bytes, s, A, W, bias, residual, Cf, Ch, c_plane, M, N, K,  \
                           out_scale, tp, qo);                                                           \
    } while (0)
#define PGMI_LAUNCH16X_S(EPI_, OUT_) do { if constexpr (OUT_ == 0) { if (stg == 3) { PGMI_LAUNCH16X(EPI_, OUT_, 3); break; } } \
        if (stg) PGMI_LAUNCH16X(EPI_, OUT_, 1); else PGMI_LAUNCH16X(EPI_, OUT_, 0); } while (0)
    if (qkv) PGMI_LAUNCH16X_S(EPI_NONE, 2);
    else {
        const int out = Ch ? 1 : 0;
        if (epilogue == EPI_GELU) { if (out) PGMI_LAUNCH16X_S(EPI_GELU, 1); else PGMI_LAUNCH16X_S(EPI_GELU, 0); }
        else if (epilogue == EPI_SQRELU) { if (out) PGMI_LAUNCH16X_S(EPI_SQRELU, 1); else PGMI_LAUNCH16X_S(EPI_SQRELU, 0); }
        else { if (out) PGMI_LAUNCH16X_S(EPI_NONE, 1); else PGMI_LAUNCH16X_S(EPI_NONE, 0); }
    }
#undef PGMI_LAUNCH16X_S
#undef PGMI_LAUNCH16X
    if (tp.split > 1)
        hipLaunchKernelGGL(splitk_fix_kernel, dim3(T - tp.n_main), dim3(XNT), 0, s, tp.ws, tp.split, tp.n_main, tp.tiles_m,
                           tp.tiles_n, bias, residual, Cf, M, N, out_scale);
    PGMI_HIP(hipGetLastError());
    return PGMI_OK;
}

// The kernel addresses its operands with 32-bit byte offsets from a buffer descriptor (no 64-bit address arithmetic in the
// memory phases): one launch covers at most 2^32 bytes of A, i.e. rows * K * 4 < 4 GiB.  Larger activations (ESM2-15B FC2:
// K = 20480 allows 52 428 rows; an MSA Transformer alignment of 400 x 1024 tokens at K = 3072) are cut into row chunks --
// rows are independent and every row is computed exactly as in one launch (bit-identical), the chunks run back to back on the
// stream.  Fused QKV output: chunks are whole sequences (the V^T scatter is per (sequence, head)).
static int launch_gemm16x(const unsigned short* A, const unsigned short* W,
                          const float* bias, const float* residual, float* Cf, unsigned short* Ch, size_t c_plane,
                          int M, int N, int K, int epilogue, float out_scale, int stg, bool splitk, hipStream_t s,
                          const QkvOut* qkv = nullptr) {
    const unsigned long long lim = (1ull << 32) - 1;
    if ((unsigned long long)N * (unsigned long long)K * 4ull > lim) {
        set_error("gemm16x: weight of %d x %d split elements exceeds the 32-bit offset range", N, K);
        return PGMI_EINVAL;
    }
    const char* tr = getenv("PGMI_GEMM_MAX_ROWS");                      // tests: force chunking at small shapes (read per launch)
    const long long test_rows = tr ? atoll(tr) : 0;
    long long max_rows = (long long)(lim / ((unsigned long long)K * 4ull));
    if (test_rows > 0) max_rows = std::min(max_rows, test_rows);
    if (M <= max_rows) return launch_gemm16x_one(A, W, bias, residual, Cf, Ch, c_plane, M, N, K, epilogue, out_scale, stg, splitk, s, qkv);
    long long per = qkv ? (max_rows / qkv->T) * qkv->T : (max_rows / XBM) * XBM;
    if (per <= 0) per = qkv ? 0 : max_rows;
    if (per <= 0) { set_error("gemm16x: one sequence of %d tokens x K = %d exceeds the 32-bit offset range", qkv->T, K); return PGMI_EINVAL; }
    for (long long m0 = 0; m0 < M; m0 += per) {
        const int mc = (int)std::min<long long>(per, M - m0);
        const unsigned short* Ac = A + (size_t)m0 * (size_t)K * 2;                     // K-interleaved rows: 2 K halfs each
        int rc;
        if (qkv) {
            QkvOut q = *qkv;                                                           // rows m0.. are sequences m0 / T ..
            q.vt16 = qkv->vt16 + (size_t)(m0 / qkv->T) * (size_t)qkv->H * kHeadDim * (size_t)qkv->Tp;
            rc = launch_gemm16x_one(Ac, W, bias, nullptr, nullptr, Ch + (size_t)m0 * (size_t)(2 * (N / 3)), c_plane, mc, N, K, epilogue,
                                    out_scale, stg, splitk, s, &q);
        } else {
            rc = launch_gemm16x_one(Ac, W, bias, residual ? residual + (size_t)m0 * N : nullptr, Cf ? Cf + (size_t)m0 * N : nullptr,
                                    Ch ? Ch + (size_t)m0 * (size_t)(2 * N) : nullptr, c_plane, mc, N, K, epilogue, out_scale, stg, splitk, s, nullptr);
        }
        if (rc) return rc;
    }
    return PGMI_OK;
}

// f16x3 variants (tuning; 0 is the product's): 0 persistent ping-pong kernel with global->LDS DMA staging; 2 the same with
// the DMA issued after the fragment reads; 1 / 3 register staging; 13 phase-timing diagnostics (PGMI_GEMM_DIAG_FLAGS).  K-sliced tails only with PGMI_GEMM_SPLITK=1 (not with 3).
int launch_gemm16(const unsigned short* A, size_t a_plane, const unsigned short* W, size_t w_plane,
                  const float* bias, const float* residual, float* Cf, unsigned short* Ch, size_t c_plane,
                  int M, int N, int K, int epilogue, float out_scale, int planes, bool bf, int variant,
                  hipStream_t s) {
    (void)a_plane; (void)w_plane;                 // f16x3 operands are K-interleaved (no plane stride); bf16 has one plane
    if (M <= 0 || N <= 0 || K <= 0 || (K % 64) != 0 || (N % 4) != 0 || (!Cf && !Ch) || (Cf && Ch)) {
        set_error("gemm16: unsupported shape/args M=%d N=%d K=%d (K %% 64 == 0, N %% 4 == 0 required)", M, N, K);
        return PGMI_EINVAL;
    }
    if (planes == 2 && !bf) {
        if (Ch && (N % 32) != 0) { set_error("gemm16: split output needs N %% 32 == 0 (K-interleaved operand of the next GEMM), got %d", N); return PGMI_EINVAL; }
        switch (variant) {
            case 1: return launch_gemm16x(A, W, bias, residual, Cf, Ch, c_plane, M, N, K, epilogue, out_scale, 0, true, s);
            case 2: return launch_gemm16x(A, W, bias, residual, Cf, Ch, c_plane, M, N, K, epilogue, out_scale, 3, true, s);
            case 3: return launch_gemm16x(A, W, bias, residual, Cf, Ch, c_plane, M, N, K, epilogue, out_scale, 0, false, s);
            case 13: return launch_gemm16x(A, W, bias, residual, Cf, Ch, c_plane, M, N, K, epilogue, out_scale, 2, false, s);
            // measured (profiles/r2/README.md): with buffer loads the DMA form wins for every output kind (FFN 368 -> 379 TFLOP/s
            // against register staging for the fp32-output GEMMs)
            default: return launch_gemm16x(A, W, bias, residual, Cf, Ch, c_plane, M, N, K, epilogue, out_scale, 1, true, s);
        }
    }
    if (planes == 1 && bf) {
        if (out_scale != 1.0f) { set_error("gemm16: bf16 weights are not pre-scaled"); return PGMI_EINVAL; }
        if ((long long)M * N >= 1 << 22) return launch_bf16_cfg<2, 4, 4, 2>(A, W, bias, residual, Cf, Ch, M, N, K, epilogue, s);
        return launch_bf16_cfg<2, 2, 2, 2>(A, W, bias, residual, Cf, Ch, M, N, K, epilogue, s);
    }
    set_error("gemm16: unsupported mode planes=%d bf=%d", planes, (int)bf);
    return PGMI_EINVAL;
}

// Fused QKV projection for the f16x3 attention: writes qk16 (q|k split planes, [M][2D]) and vt16
// (transposed key-permuted V planes) instead of an fp32 [M,3D] tensor; rotary applied to q,k if set.
int launch_gemm16_qkv(const unsigned short* A, size_t a_plane, const unsigned short* W, size_t w_plane,
                      const float* bias, int M, int D, int K, float out_scale, unsigned short* qk16, size_t qk_plane,
                      unsigned short* vt16, size_t vt_plane, const float* cos_t, const float* sin_t, int rotary,
                      int T, int H, int variant, hipStream_t s, int rot_halves) {
    (void)a_plane; (void)w_plane;
    if (M <= 0 || D <= 0 || (K % 64) || (D % 64) || M % T || rot_halves < 1) {
        set_error("gemm16_qkv: unsupported shape M=%d D=%d K=%d T=%d", M, D, K, T);
        return PGMI_EINVAL;
    }
    QkvOut qo{vt16, vt_plane, cos_t, sin_t, T, H, (T + 31) / 32 * 32, rotary, rot_halves};
    return launch_gemm16x(A, W, bias, nullptr, nullptr, qk16, qk_plane, M, 3 * D, K, EPI_NONE, out_scale, (variant == 1 || variant == 3) ? 0 : 1,
                          false, s, &qo);
}

// ---- fp32 -> 16-bit operands (weights at load time; activations in the op-level tests and the MSA tied-attention path) ----
// mode 0: f16x3 weight: hi = fp16(x*scale), lo = fp16(x*scale - hi), K-interleaved rows of length K
// mode 1: bf16 (single plane, RNE)
// mode 2: f16x3 activation split (lo scaled by 2^11, see split_act), K-interleaved rows of length K
__global__ void split16_kernel(const float* __restrict__ x, int64_t n, float scale, int mode, int K,
                               unsigned short* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i] * scale;
    if (mode == 1) {
        out[i] = f32_to_bf16_rne(v);
        return;
    }
    _Float16 hi, lo;
    if (mode == 2) {
        split_act(v, hi, lo);
    } else {
        hi = (_Float16)v;
        lo = (_Float16)(v - (float)hi);
    }
    const size_t o = ki_off((size_t)(i / K), (int)(i % K), K);
    out[o] = __builtin_bit_cast(unsigned short, hi);
    out[o + 32] = __builtin_bit_cast(unsigned short, lo);
}
void launch_split16(const float* x, int64_t n, float scale, int mode, int K, unsigned short* out, hipStream_t s) {
    hipLaunchKernelGGL(split16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, scale, mode, K, out);
}

}  // namespace pgmi
