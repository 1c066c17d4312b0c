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
#include <string.h>
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"
#include "gemm16x_kernel.h"            // typedefs, gelu_erf16, QkvOut, TilePlan, gemm16x_kernel (the persistent ping-pong kernel)

namespace pgmi {

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
                for (int e = 0; e < 4; ++e) val[e] = acc[j][i][4 * g + e] + bv[e];
                if (EPI == EPI_GELU) val = gelu_erf16(val);
                if (EPI == EPI_SQRELU)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float t = fmaxf(val[e], 0.0f); val[e] = t * t; }
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

// Per-device launch state (a process may drive several devices through distinct model handles: nothing here is shared
// between devices).  Indexed by the current HIP device of the calling thread; written under a mutex because two handles on two
// devices may launch from two host threads.
constexpr int kMaxDevices = 64;
static int g_num_cus[kMaxDevices];
static std::mutex g_xdev_mu;

static int x_num_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    std::lock_guard<std::mutex> lk(g_xdev_mu);
    int& n = g_num_cus[dev];
    if (!n) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
        n -= n % 8;                                              // the XCD-aware item order wants a multiple of 8
        if (n <= 0) n = 8;
    }
    return n;
}

// Row panels per group of the grouped tile order.  An XCD's 32 concurrent tiles then cover ~g row panels x 32/g column panels: per K
// step g A slices + 32/g W slices miss its L2.  W (the layer's weights, 6.5 - 26 MB) is shared by every tile of the launch and
// stays in the Infinity Cache; A (the activations, 0.4 - 1.7 GB) streams from HBM: fewer A slices per step win although the slice
// count is the same -- measured (scripts/gemm_ab.py, BLAT shape) g = 8 -> 4: QKV 400 -> 414, out 374 -> 383, FC2 419 -> 429 TFLOP/s,
// FC1 unchanged; g = 2 within noise of 4, 16 / 32 / 64 worse.  Tile order does not touch a row's arithmetic: same bits.
// FC1 (N = 5120: 20 column panels) runs at the same speed with 4 and with 8 but fetches less with 8 (3.26 vs 3.47 GB per launch: with
// 20 column panels a group of 4 rows is 80 tiles = 2.5 rounds of an XCD, and the partial rounds straddle two groups): wide outputs keep 8.
constexpr int kGroupM = 4, kGroupMWide = 8, kWideTilesN = 16;
// Test hooks (bit-neutral: both only change how a launch is cut into items / row chunks).  Read from the environment when a model is
// created and at the op-level entries (gemm_options_from_env), changed on a live model through pgmi_set_option -- never per launch.
//   PGMI_GEMM_HALF_TAIL / "gemm_half_tail": 0 = no half-height tail items;  PGMI_GEMM_MAX_ROWS / "gemm_max_rows": force row chunks
// An option set through pgmi_set_option stays in force: creating another model or calling an op-level entry re-reads the environment
// only for the options nobody has set explicitly (value -1 hands an option back to the environment / its default).  Process-wide and
// not synchronised: they are test hooks, changed between launches by the thread that launches.
struct GemmOptions { int half_tail = 1; long long max_rows = 0; bool half_tail_set = false, max_rows_set = false; };
static GemmOptions g_opt;
void gemm_options_from_env() {
    const char* h = getenv("PGMI_GEMM_HALF_TAIL");
    const char* r = getenv("PGMI_GEMM_MAX_ROWS");
    if (!g_opt.half_tail_set) g_opt.half_tail = h ? atoi(h) : 1;
    if (!g_opt.max_rows_set) g_opt.max_rows = r ? atoll(r) : 0;
}
int gemm_set_option(const char* name, long long value) {
    if (!strcmp(name, "gemm_half_tail")) {
        g_opt.half_tail_set = value >= 0;
        if (value >= 0) g_opt.half_tail = (int)value; else gemm_options_from_env();
        return PGMI_OK;
    }
    if (!strcmp(name, "gemm_max_rows")) {
        g_opt.max_rows_set = value >= 0;
        if (value >= 0) g_opt.max_rows = value; else gemm_options_from_env();
        return PGMI_OK;
    }
    return PGMI_EINVAL;
}

// Launch parameters: variants >= 1000 of launch_gemm16 override the row panels per group for the interleaved A/B of
// scripts/gemm_ab.py (tile order does not touch a row's arithmetic: same bits).
struct GemmTune { int group_m = 0; };
static thread_local GemmTune g_tune;

static int launch_gemm16x_one(const unsigned short* A, const unsigned short* W,
                              const float* bias, const float* residual, float* Cf, unsigned short* Ch, size_t c_plane,
                              int M, int N, int K, int epilogue, float out_scale, hipStream_t s, const QkvOut* qkv, bool bf = false) {
    const unsigned long long eb = bf ? 2ull : 4ull;                  // operand bytes per k element (gemm16x_kernel.h: BF)
    const GemmTune tune = g_tune;
    TilePlan tp{};
    tp.group_m = tune.group_m > 0 ? tune.group_m : ((N + XBN - 1) / XBN >= kWideTilesN ? kGroupMWide : kGroupM);
    tp.tiles_m = (M + XBM - 1) / XBM;
    tp.tiles_n = (N + XBN - 1) / XBN;
    const int T = tp.tiles_m * tp.tiles_n, G = x_num_cus();
    tp.n_main = T;
    if ((unsigned long long)std::max(M, N) * (unsigned long long)K * eb >= (1ull << 32) ||
        (Cf && (unsigned long long)M * (unsigned long long)N * 4ull >= (1ull << 31))) {      // launch_gemm16x chunks M below this
        set_error("gemm16x: operand of %d x %d split elements (or %d x %d outputs) exceeds the 32-bit offset range", std::max(M, N), K, M, N);
        return PGMI_EINVAL;
    }
    const int rem = T % G;
    // Half-height tail (every output kind but the fused QKV): the last, partial round of tiles leaves G - rem CUs idle for a whole
    // tile time (N = 1280 at the BLAT shape: 6.29 rounds cost 7).  Its tiles are cut into their upper and lower 128 rows -- two
    // items on two CUs, each over the full K range in the same order, so every output element is computed exactly as in a
    // full tile (bit-identical: a row's bits must not depend on how many rows travel with it, tests/test_gpu_cli.py) -- when
    // all the halves still fit one round.
    const int want_half = g_opt.half_tail;
    if (want_half && !qkv && rem > 0 && 2 * rem <= G) {
        tp.n_main = T - rem; tp.half = 1; tp.n_tail = 2 * rem;
    }
    const int n_items = tp.n_main + tp.n_tail;
    const dim3 grid(std::min(G, n_items)), block(XNT);
    QkvOut qo{};
    if (qkv) qo = *qkv;
    const XMap xmap{};
#define PGMI_LAUNCH16X(EPI_, OUT_)                                                                        \
    do {                                                                                                 \
        auto kfn = bf ? gemm16x_kernel<EPI_, OUT_, false, true> : gemm16x_kernel<EPI_, OUT_>;             \
        const size_t lds_bytes = X_LDS_BYTES;                                                            \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                           \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);  \
        if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PGMI_EHIP; } \
        hipLaunchKernelGGL(kfn, grid, block, lds_bytes, s, A, W, bias, residual, Cf, Ch, c_plane, M, N, K,  \
                           out_scale, tp, qo, xmap);                                                     \
    } while (0)
#define PGMI_LAUNCH16X_O(EPI_) do { if (Ch) PGMI_LAUNCH16X(EPI_, 1); else PGMI_LAUNCH16X(EPI_, 0); } while (0)
    if (qkv) PGMI_LAUNCH16X(EPI_NONE, 2);
    else if (epilogue == EPI_GELU) PGMI_LAUNCH16X_O(EPI_GELU);
    else if (epilogue == EPI_SQRELU) PGMI_LAUNCH16X_O(EPI_SQRELU);
    else PGMI_LAUNCH16X_O(EPI_NONE);
#undef PGMI_LAUNCH16X_O
#undef PGMI_LAUNCH16X
    PGMI_HIP(hipGetLastError());
    return PGMI_OK;
}

// Batched / strided form (XMap, gemm16x_kernel.h): the MSA Transformer's tied row attention.  M, N, K are ONE batch's; nbatch batches
// share the launch (xm.tiles_per_batch is filled in here).  fp32 output (Cf; always through the LDS-transpose epilogue) or split
// planes (Ch; dense or scattered by xm.o_*).  M just above a multiple of 128: 128-row tiles throughout (TilePlan.half = 2).
int launch_gemm16_ex(const unsigned short* A, const unsigned short* W, float* Cf, unsigned short* Ch, int M, int N, int K,
                     float out_scale, XMap xm, int nbatch, hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0 || (K % 64) != 0 || (N % 4) != 0 || (!Cf && !Ch) || (Cf && Ch) || nbatch < 1 ||
        (Ch && (N % 64) != 0) || (xm.a_run_bytes && (K / 32) % (1 << xm.k_run_log2) != 0)) {
        set_error("gemm16_ex: unsupported shape/args M=%d N=%d K=%d batches=%d", M, N, K, nbatch);
        return PGMI_EINVAL;
    }
    TilePlan tp{};
    tp.group_m = kGroupM;
    const int rows_last = M % XBM;                                // rows in the last 256-row panel
    const bool all_half = rows_last > 0 && rows_last <= XBM / 2;  // ... at most half of it: 128-row tiles waste less
    tp.tiles_m = all_half ? (M + XBM / 2 - 1) / (XBM / 2) : (M + XBM - 1) / XBM;
    tp.tiles_n = (N + XBN - 1) / XBN;
    const int per = tp.tiles_m * tp.tiles_n, T = per * nbatch, G = x_num_cus();
    xm.tiles_per_batch = per;
    if (xm.batch_inner < 1) xm.batch_inner = 1;
    if (all_half) { tp.half = 2; tp.n_main = 0; tp.n_tail = T; } else { tp.n_main = T; }
    const dim3 grid(std::min(G, T)), block(XNT);
    QkvOut qo{};
    const float* nobias = nullptr;
    const size_t lds_bytes = X_LDS_BYTES;
#define PGMI_LAUNCH16X_EX(OUT_)                                                                           \
    do {                                                                                                 \
        auto kfn = gemm16x_kernel<EPI_NONE, OUT_, true>;                                                  \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                           \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);  \
        if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PGMI_EHIP; } \
        hipLaunchKernelGGL(kfn, grid, block, lds_bytes, s, A, W, nobias, nobias, Cf, Ch, (size_t)0, M, N, K,  \
                           out_scale, tp, qo, xm);                                                       \
    } while (0)
    if (Ch) PGMI_LAUNCH16X_EX(1); else PGMI_LAUNCH16X_EX(0);
#undef PGMI_LAUNCH16X_EX
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
                          int M, int N, int K, int epilogue, float out_scale, hipStream_t s, const QkvOut* qkv = nullptr, bool bf = false) {
    const unsigned long long lim = (1ull << 32) - 1;
    const unsigned long long eb = bf ? 2ull : 4ull;
    if ((unsigned long long)N * (unsigned long long)K * eb > lim) {
        set_error("gemm16x: weight of %d x %d split elements exceeds the 32-bit offset range", N, K);
        return PGMI_EINVAL;
    }
    const long long test_rows = g_opt.max_rows;                          // tests: force chunking at small shapes
    long long max_rows = (long long)(lim / ((unsigned long long)K * eb));
    if (Cf) max_rows = std::min(max_rows, (long long)(((1ull << 31) - 1) / ((unsigned long long)N * 4ull)));   // the fp32 epilogue's buffer offsets
    if (test_rows > 0) max_rows = std::min(max_rows, test_rows);
    if (M <= max_rows) return launch_gemm16x_one(A, W, bias, residual, Cf, Ch, c_plane, M, N, K, epilogue, out_scale, s, qkv, bf);
    long long per = qkv ? (max_rows / qkv->T) * qkv->T : (max_rows / XBM) * XBM;
    if (per <= 0) per = qkv ? 0 : max_rows;
    if (per <= 0) { set_error("gemm16x: one sequence of %d tokens x K = %d exceeds the 32-bit offset range", qkv->T, K); return PGMI_EINVAL; }
    for (long long m0 = 0; m0 < M; m0 += per) {
        const int mc = (int)std::min<long long>(per, M - m0);
        const unsigned short* Ac = A + (size_t)m0 * (size_t)K * (bf ? 1 : 2);          // K-interleaved rows: 2 K halfs each (bf16: K)
        int rc;
        if (qkv) {
            QkvOut q = *qkv;                                                           // rows m0.. are sequences m0 / T ..
            q.vt16 = qkv->vt16 + (size_t)(m0 / qkv->T) * (size_t)qkv->H * kHeadDim * (size_t)qkv->Tp;
            rc = launch_gemm16x_one(Ac, W, bias, nullptr, nullptr, Ch + (size_t)m0 * (size_t)(2 * (N / 3)), c_plane, mc, N, K, epilogue,
                                    out_scale, s, &q, bf);
        } else {
            rc = launch_gemm16x_one(Ac, W, bias, residual ? residual + (size_t)m0 * N : nullptr, Cf ? Cf + (size_t)m0 * N : nullptr,
                                    Ch ? Ch + (size_t)m0 * (size_t)((bf ? 1 : 2) * N) : nullptr, c_plane, mc, N, K, epilogue, out_scale, s, nullptr, bf);
        }
        if (rc) return rc;
    }
    return PGMI_OK;
}

// variant (tuning; everything below 1000 is the product's configuration): 1000 + t sets the round-4 launch parameters for the
// interleaved A/B of scripts/gemm_ab.py -- t bits 0-3: row panels per group (0: by shape).
static void set_tune(int variant) {
    g_tune = GemmTune{};
    if (variant >= 1000) {
        const int t = variant - 1000;
        g_tune.group_m = t & 15;
    }
}

int launch_gemm16(const unsigned short* A, size_t a_plane, const unsigned short* W, size_t w_plane,
                  const float* bias, const float* residual, float* Cf, unsigned short* Ch, size_t c_plane,
                  int M, int N, int K, int epilogue, float out_scale, int planes, bool bf, int variant,
                  hipStream_t s) {
    (void)a_plane; (void)w_plane;                 // f16x3 operands are K-interleaved (no plane stride); bf16 has one plane
    // K: whole 32-deep K tiles for the f16x3 kernel (one 128-byte line per row and tile; an odd tile count is fine: ESM2-35M has
    // K = 480 = 15 tiles), 64 for the bf16 kernel's K tile
    const int k_step = (planes == 2 && !bf) ? 32 : 64;
    if (M <= 0 || N <= 0 || K <= 0 || (K % k_step) != 0 || (N % 4) != 0 || (!Cf && !Ch) || (Cf && Ch)) {
        set_error("gemm16: unsupported shape/args M=%d N=%d K=%d (K %% %d == 0, N %% 4 == 0 required)", M, N, K, k_step);
        return PGMI_EINVAL;
    }
    if (planes == 2 && !bf) {
        if (Ch && (N % 32) != 0) { set_error("gemm16: split output needs N %% 32 == 0 (K-interleaved operand of the next GEMM), got %d", N); return PGMI_EINVAL; }
        set_tune(variant);
        const int rc = launch_gemm16x(A, W, bias, residual, Cf, Ch, c_plane, M, N, K, epilogue, out_scale, s);
        g_tune = GemmTune{};
        return rc;
    }
    if (planes == 1 && bf) {
        if (out_scale != 1.0f) { set_error("gemm16: bf16 weights are not pre-scaled"); return PGMI_EINVAL; }
        if (Ch && (N % 4) != 0) { set_error("gemm16: bf16 plane output needs N %% 4 == 0, got %d", N); return PGMI_EINVAL; }
        // round 6: the persistent ping-pong kernel in its one-plane form (gemm16x_kernel.h BF); the round-1 kernel is kept for the A/B
        // (variant 2000: scripts/gemm_ab.py)
        if (variant != 2000) {
            set_tune(variant);
            const int rc = launch_gemm16x(A, W, bias, residual, Cf, Ch, c_plane, M, N, K, epilogue, 1.0f, s, nullptr, true);
            g_tune = GemmTune{};
            return rc;
        }
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
                      int T, int H, int variant, hipStream_t s, int rot_halves, bool bf) {
    (void)a_plane; (void)w_plane;
    if (M <= 0 || D <= 0 || (K % (bf ? 64 : 32)) || (D % 64) || M % T || rot_halves < 1) {
        set_error("gemm16_qkv: unsupported shape M=%d D=%d K=%d T=%d", M, D, K, T);
        return PGMI_EINVAL;
    }
    QkvOut qo{vt16, vt_plane, cos_t, sin_t, T, H, (T + 31) / 32 * 32, rotary, rot_halves};
    set_tune(variant);
    const int rc = launch_gemm16x(A, W, bias, nullptr, nullptr, qk16, qk_plane, M, 3 * D, K, EPI_NONE, out_scale, s, &qo, bf);
    g_tune = GemmTune{};
    return rc;
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
