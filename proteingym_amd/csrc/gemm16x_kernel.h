// gemm16x_kernel: the persistent ping-pong f16x3 GEMM kernel of libpgmi (launched by gemm_f16.hip).
#pragma once
#include <type_traits>

#include "common.h"

namespace pgmi {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// erf-GELU (modules.py:17-24, nn.GELU) in 10 VALU issue slots per 2 values + one quarter-rate exp2 per value.  The FC1 epilogue runs while
// the matrix pipe idles and its cost is VALU issue: two waves per SIMD x 128 values per lane; erff costs ~38 instructions, the round-2
// form (Abramowitz-Stegun t = 1 / (1 + p z), Q(t) exp(-z^2): 17 instructions, an rcp AND an exp) ~78 issue cycles per value, this ~38.
//   gelu(x) = max(x, 0) - a Phi(-a),  a = min(|x|, 6),  Phi(-a) = 2^P(a)
// P of degree 6, fitted by reweighted least squares to the ABSOLUTE error of a 2^P(a) on [0, 6] (5.1e-8; beyond 6 the term is below
// 6e-9 and held there); written in b = -a so that the last step is one fma.  No 1 + erf cancellation for x < 0.  Against fp64 the fp32
// evaluation is within 2.8e-7 abs on [-10, 10] (torch's fp32 gelu: 1.2e-6), mean 3.0e-8 (torch: 6.1e-8) -- scripts/fit_gelu.py.
// Four values at a time so that the seven fmas are PACKED instructions (v_pk_fma_f32: two values per issue slot; from scalar code
// hipcc prefers v_fmaak_f32 with a literal, one value per slot).
__device__ __forceinline__ f32x4 gelu_erf16(f32x4 x) {
    auto all = [](float v) { return f32x4{v, v, v, v}; };
    const f32x4 b = __builtin_elementwise_max(-__builtin_elementwise_abs(x), all(-6.0f));
    // x * 0 in the first step: a NaN or an infinity in x must come out as NaN (max / min return their OTHER operand for a NaN, so b and
    // max(x, 0) alone would turn NaN into -6e-9: the LM head's GELU would swallow the non-finite values the fp16 range guard looks for)
    f32x4 q = __builtin_elementwise_fma(x, all(0.0f), all(3.309269596e-05f));
    q = __builtin_elementwise_fma(q, b, all(7.692188374e-04f));
    q = __builtin_elementwise_fma(q, b, all(8.080714382e-03f));
    q = __builtin_elementwise_fma(q, b, all(5.341210216e-02f));
    q = __builtin_elementwise_fma(q, b, all(-4.587709904e-01f));
    q = __builtin_elementwise_fma(q, b, all(1.151201725e+00f));
    q = __builtin_elementwise_fma(q, b, all(-9.999930859e-01f));
    const f32x4 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1]), __builtin_amdgcn_exp2f(q[2]), __builtin_amdgcn_exp2f(q[3])};
    return __builtin_elementwise_fma(b, e, __builtin_elementwise_max(x, all(0.0f)));
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

// =================================================================================================
// Persistent ping-pong kernel (f16x3, 256 x 256 x 32 tile, 8 waves = 2(M) x 4(N), wave tile 128 x 64).
// One workgroup per CU walks a list of work items; per item the K loop is a ping-pong schedule: the two waves of a SIMD
// run one phase apart, one issues its 24 MFMAs while the other reads fragments / issues the DMA of the next K tile.
// What the persistent form adds:
//   * no workgroup launch/retire gap between tiles; the first K tile of item i+1 is in flight (DMA into buffer 0) while
//     the epilogue of item i runs -- the epilogues' LDS patches start at buffer 1, which is free until the next K loop;
//   * the tail of the launch is balanced: with T tiles on G workgroups the tiles of the last, partial round (1610 tiles on
//     256 CUs = 6.29 rounds for the N = 1280 GEMMs) are cut into their upper and lower 128 rows -- two items, each over the
//     full K range in the same order, so every output element is computed exactly as in a full tile (bit-identical);
// Measured and NOT kept (round 4, scripts/gemm_ab.py, profiles/r4/README.md): starting the XCDs 2 - 12 us apart, or the CU slots
// of an XCD 0.1 - 0.7 us apart, so that the epilogues' 67 MB of stores (+ 67 MB of residual reads) per round of items do not hit
// the fabric in the same microseconds -- no gain at any setting; with the epilogue's stores dropped (timing probe) the
// out-projection runs 8.8 % faster, with its residual loads dropped 6.7 %, with both 13 %: the waves wait for their own stores to
// drain at the first vmcnt(0) of the next item (gfx9 counts loads and stores on ONE in-order counter).
// LDS image of a K tile (per operand): [256 rows][8 chunks of 16 B] = the row's 128-byte line (4 hi chunks, 4 lo chunks)
// with the chunk index XORed by (row >> 1) & 7: conflict-free ds_read_b128 fragment reads at a 128-byte row pitch.  Staging
// is global -> LDS DMA (buffer_load ... lds; the swizzle is applied to the SOURCE chunk, which stays inside the row's line);
// the wave's share of tile kt+1 is issued in the first memory phase of tile kt and waited for at the LAST barrier of tile kt.
// =================================================================================================
struct TilePlan {
    int tiles_m, tiles_n;
    int n_main;       // items [0, n_main): one full tile each (XCD-aware grouped order); workgroup b runs b, b + G, ...
    int n_tail;       // tail items n_main + h, h < n_tail, one per workgroup at most (half = 1: halves of the last round's tiles)
    int half;         // 1: tail item h is the upper (h even) or lower (h odd) 128 rows of tile n_main + h / 2;
                      // 2: EVERY item is a 128-row tile (tiles_m counts 128-row panels, n_main = 0, n_tail = all of them): M just above a
                      //    multiple of 128 (the tied row attention's C = 287 columns) wastes a third of a 256-row panel less
    int group_m;      // row panels per group of the grouped tile order (gemm_f16.hip kGroupM)
};

// How the GEMM's indices map onto memory when an operand is not a dense [rows][K] matrix / the output not a dense [M][N] one: the tied
// row attention of the MSA Transformer (axial_attention.py:112-168) contracts over (alignment row r, head dim d) with q, k used in
// place -- K runs of 64 elements a row stride apart --, one batch per (head, K split), and scatters its update back to [r, i, h, d].
// All zero = dense, one batch.
struct XMap {
    unsigned int a_row_bytes, w_row_bytes;     // byte stride between consecutive operand rows (0: 4 K)
    unsigned int a_bytes, w_bytes;             // extent of the operand arrays for the buffer descriptors (0: rows x 4 K)
    int k_run_log2;                            // K tiles (of 32) per contiguous run, log2 (0 with a_run_bytes == 0: one run = dense)
    unsigned int a_run_bytes, w_run_bytes;     // byte stride between runs
    int tiles_per_batch, batch_inner;          // tiles per batch (0: one batch); batch b = outer * batch_inner + inner
    unsigned int a_b0, a_b1, w_b0, w_b1;       // operand byte offsets per inner / outer batch index
    long long c_b0, c_b1;                      // fp32 output: element offsets per inner / outer batch index
    int ldc;                                   // fp32 output: row pitch in elements (0: N)
    int o_ld;                                  // split-plane output scatter (0: dense [M][N]): row pitch in columns, ...
    int o_rows_per_n64;                        // ... output row = m + (n / 64) * o_rows_per_n64, ...
    int o_col_per_batch;                       // ... column = batch * o_col_per_batch + n % 64
};

constexpr int XBM = 256, XBN = 256, XNT = 512, XCPR = 8;   // 8 chunks = one 128-byte line per row and K tile (hi | lo)
constexpr int X_OP_CH = XBM * XCPR;                        // 16-byte chunks of one operand tile
constexpr int X_STAGE = 2 * X_OP_CH;                       // chunks per stage = 64 KB
constexpr int X_PATCH_BYTES = 72 * 1024;                   // epilogue patches: buffer 1 and 8 KB beyond it (8 waves x <= 9 KB)
constexpr size_t X_LDS_BYTES = (size_t)(X_STAGE * 16 + X_PATCH_BYTES);   // 136 KB

__device__ __forceinline__ void x_tile_coords(int wgid, int tiles_m, int tiles_n, int& tm, int& tn, int GROUP_M) {
    const int width = GROUP_M * tiles_n;
    const int group = wgid / width, first_m = group * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    tm = first_m + (wgid % width) % gsz;
    tn = (wgid % width) / gsz;
}

// OUT 0: fp32 [M,N] (+ residual); 1: split fp16 planes, K-interleaved (the next GEMM's operand); 2: attention operands (QkvOut).
// XM: the batched / strided form (XMap); false = dense operands, one batch: xm is ignored (and costs nothing).
// BF (round 6, the un-gated bf16 throughput mode): operands are ONE bf16 plane, row-major [rows][K].  A K tile is then 64 deep -- the same 128
// bytes per row and tile, so staging, LDS image and fragment reads are unchanged -- and a compute phase issues 16 MFMAs (k16 steps ks and
// 2 + ks of the tile: the chunks the f16x3 form reads as the hi and the lo half of one step) instead of the 24 of the three split products.
template <int EPI, int OUT, bool XM = false, bool BF = false>
__global__ __launch_bounds__(XNT, 2) void gemm16x_kernel(
    const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, const float* __restrict__ bias,
    const float* residual, float* Cf, unsigned short* Ch, size_t c_plane, int M, int N, int K, float out_scale,
    TilePlan tp, QkvOut qo, XMap xm_arg) {
    static_assert(!(XM && BF), "the batched / strided form is f16x3 only");
    constexpr unsigned int EB = BF ? 2u : 4u;                     // operand bytes per k element: one bf16, or the hi and the lo half
    const XMap xm = XM ? xm_arg : XMap{};                         // dense instantiations: every xm test below folds away
    constexpr int WN = 4, TMX = 4, TN = 2, LD = 4;               // TMX: 32-row MFMA tiles per wave of a full item (a half item: 2)
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];  // [2][X_STAGE] + patches
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // provably wave-uniform: scalar branches, SGPR LDS bases
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, kh = lane >> 5;
    const bool late = wave >= 4;                                  // the second wave of every SIMD
    // epilogue patches: from buffer 1 on (free between the last phase barrier of an item and the first memory phase of the
    // next one, which every wave enters through a barrier after its own epilogue) -- buffer 0 takes the next item's first K tile
    unsigned char* const patches = reinterpret_cast<unsigned char*>(lds + X_STAGE);

    // staging geometry (the same for both operands): slot f = tid + 512 i is (row (tid >> 3) + 64 i, chunk tid & 7) -- 8
    // consecutive lanes move one row's 128-byte line -- into the lane-linear LDS slot; the chunk swizzle is applied to the
    // SOURCE: one 32-bit byte offset per row and operand.
    const int row_lo = tid >> 3, c8 = tid & 7, sw8 = (row_lo >> 1) & 7;
    const int csrc = c8 ^ sw8;
    // buffer descriptors built from kernel arguments only (provably wave-uniform): loads take a 32-bit per-lane byte offset
    // and the K-tile offset as an SGPR -- no 64-bit address arithmetic in the memory phases
    const unsigned int a_row_bytes = xm.a_row_bytes ? xm.a_row_bytes : (unsigned int)K * EB;
    const unsigned int w_row_bytes = xm.w_row_bytes ? xm.w_row_bytes : (unsigned int)K * EB;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A), 0, (int)(xm.a_bytes ? xm.a_bytes : (unsigned int)M * (unsigned int)K * EB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, (int)(xm.w_bytes ? xm.w_bytes : (unsigned int)N * (unsigned int)K * EB), 0x00020000);
    unsigned int a_off[LD], w_off[LD];
    int m0 = 0, n0 = 0, a_ld = LD, batch = 0;
    unsigned int a_base = 0, w_base = 0;                          // the batch's operand offsets (SGPRs)
    bool half_item = false;                                       // a 128-row item: half of a tile (tail of the item list), or every item (half == 2)
    const int nk = BF ? K / 64 : K / 32;                          // K tiles of 128 bytes per row
    const int n_items = tp.n_main + tp.n_tail;
    auto decode = [&](int item) {                                 // sets m0, n0, half_item, batch and the source offsets
        const bool all_half = tp.half == 2;
        int wgid;
        if (item < tp.n_main || all_half) {
            // XCD-aware order: item i runs on XCD i % 8 (grid size is a multiple of 8); every XCD walks a contiguous
            // run of the grouped tile order so that its L2 keeps the live A panels and W tiles
            const int nwg = all_half ? n_items : tp.n_main, xcd = item & 7, q = nwg >> 3, r8 = nwg & 7;
            wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (item >> 3);
        } else {
            wgid = tp.n_main + ((item - tp.n_main) >> 1);
        }
        batch = 0;
        if (xm.tiles_per_batch) {
            batch = wgid / xm.tiles_per_batch;
            wgid -= batch * xm.tiles_per_batch;
            const int bi = batch % xm.batch_inner, bo = batch / xm.batch_inner;
            a_base = __builtin_amdgcn_readfirstlane((unsigned int)bi * xm.a_b0 + (unsigned int)bo * xm.a_b1);
            w_base = __builtin_amdgcn_readfirstlane((unsigned int)bi * xm.w_b0 + (unsigned int)bo * xm.w_b1);
            batch = __builtin_amdgcn_readfirstlane(batch);
        }
        int tm, tn;
        x_tile_coords(wgid, tp.tiles_m, tp.tiles_n, tm, tn, tp.group_m);
        half_item = all_half || (tp.half && item >= tp.n_main);
        a_ld = half_item ? LD / 2 : LD;                           // A rows staged per K tile: 64 per instruction
        m0 = __builtin_amdgcn_readfirstlane(all_half ? tm * (XBM / 2) : tm * XBM + (half_item ? ((item - tp.n_main) & 1) * (XBM / 2) : 0));
        n0 = __builtin_amdgcn_readfirstlane(tn * XBN);
#pragma unroll
        for (int i = 0; i < LD; ++i) {                            // the launcher guarantees that every offset stays below 4 GiB
            a_off[i] = (unsigned int)min(m0 + row_lo + 64 * i, M - 1) * a_row_bytes + (unsigned int)csrc * 16u;
            w_off[i] = (unsigned int)min(n0 + row_lo + 64 * i, N - 1) * w_row_bytes + (unsigned int)csrc * 16u;
        }
    };
    auto issue_tile = [&](int kt, int buf) {                      // 1 KiB per wave-instruction, wave-uniform LDS base
        u32x4* base = lds + buf * X_STAGE + wave * 64;
        // K tile kt of the item: dense rows: 128 kt; K runs (xm): run kt >> k_run_log2, tile kt & mask inside it
        const int run = kt >> xm.k_run_log2, in_run = kt - (run << xm.k_run_log2);
        const int ka = (int)(a_base + (xm.a_run_bytes ? (unsigned int)run * xm.a_run_bytes + (unsigned int)in_run * 128u : (unsigned int)kt * 128u));
        const int kw = (int)(w_base + (xm.w_run_bytes ? (unsigned int)run * xm.w_run_bytes + (unsigned int)in_run * 128u : (unsigned int)kt * 128u));
#pragma unroll
        for (int i = 0; i < LD; ++i)
            if (i < a_ld)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + XNT * i), 16, (int)a_off[i], ka, 0, 0);
#pragma unroll
        for (int i = 0; i < LD; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(base + X_OP_CH + XNT * i), 16, (int)w_off[i], kw, 0, 0);
    };
    // phase boundary: everything issued before stays before, this wave's LDS traffic has landed; VMEM stays in flight
    auto phase = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                                    // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto phase_vm = [&]() {                                                    // the same, plus this wave's DMA has landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0070);                                    // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

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
        if constexpr (BF) {                                       // two k16 steps of plain bf16 products
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[j][i] = mfma16<true>(wf[p][j], af[p][i], acc[j][i]);
            return;
        }
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

    // ---- this workgroup's item list: b, b + G, ... (items >= n_main are 128-row items) ----
    int item = blockIdx.x;
    if (item >= n_items) return;
    decode(item);
    issue_tile(0, 0);
    // one item, start to finish; TM (compile time) = 32-row MFMA tiles per wave: 4, or 2 for a half item.  Returns "more items".
    auto run_item = [&](auto tmc) -> bool {
        constexpr int TM = decltype(tmc)::value;
        // ---- prologue: the item's first K tile has been in flight into buffer 0 since before the previous epilogue ----
        phase_vm();
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[j][i][v] = 0.0f;
        int cur = 0;
        if (late) phase();
        for (int kt = 0; kt < nk; ++kt) {
            const u32x4* Ab = lds + cur * X_STAGE;
            const u32x4* Wb = Ab + X_OP_CH;
            // -- memory phase 1: DMA of tile kt+1 into the other buffer (last read two phases ago by the other wave group), then
            //    the fragments of the first k16 step.  The DMA goes first: with the fragment reads ahead of it the memory phase
            //    outlasts the partner's 24 MFMAs (tools/mfma_phase.hip: 829 -> 797 clocks per phase on the bare loop) --
            __builtin_amdgcn_s_setprio(0);
            if (kt + 1 < nk) issue_tile(kt + 1, cur ^ 1);
            read_frags(tmc, Ab, Wb, 0);
            phase();
            // -- compute phase 1 --
            __builtin_amdgcn_s_setprio(1);
            mfmas(tmc);
            phase();
            // -- memory phase 2: fragments of the second k16 step --
            __builtin_amdgcn_s_setprio(0);
            read_frags(tmc, Ab, Wb, 1);
            if (late) phase_vm(); else phase();                 // late waves close tile kt here: their DMA share must have landed
            // -- compute phase 2 --
            __builtin_amdgcn_s_setprio(1);
            mfmas(tmc);
            if (!late) phase_vm(); else phase();                // early waves close tile kt here
            cur ^= 1;
        }
        __builtin_amdgcn_s_setprio(0);
        if (!late) phase();

        // ---- the item just finished: both K-tile buffers are free after the last phase barrier; the next item's first K
        //      tile is in flight while this item's epilogue runs ----
        const int em0 = m0, en0 = n0, eb = batch;
        item += gridDim.x;
        const bool more = item < n_items;
        if (more) {
            decode(item);
            issue_tile(0, 0);
        }

        if constexpr (OUT == 2) {
            const int Dm = N / 3;
            const int nb = en0 + wn * 64;                      // first column of this wave's head
            if (nb < N) {
            const int which = nb / Dm, hcol = nb - which * Dm, hh = hcol >> 6;
            constexpr int SPQ = 144;
            unsigned char* patch_q = patches + wave * (2 * 32 * SPQ);
            const bool staged_q = which < 2;
            // the wave's bias values, once per item: a load inside the block loop is followed by a wait for EVERYTHING in
            // flight (vmcnt counts the stores of the previous block as well) -- one memory round trip per 4 values
            f32x4 bq0[4], bq1[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bq0[g] = *reinterpret_cast<const f32x4*>(bias + nb + 8 * g + 4 * kh);
                bq1[g] = *reinterpret_cast<const f32x4*>(bias + nb + 32 + 8 * g + 4 * kh);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) asm volatile("" :: "v"(bq0[g]), "v"(bq1[g]));      // waited for HERE, not inside the block loop
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = em0 + (wm * TM + i) * 32 + r;
                const bool row_ok = m < M;
                if (!staged_q && !row_ok) continue;
                const int bb = m / qo.T, t = m - bb * qo.T;
                // ESM2: the row's rotary table entries for all four column groups, loaded TOGETHER before the group loop (loads inside it
                // were waited for one group at a time, and the wait -- vmcnt counts stores too -- included the V^T stores just issued)
                // (a table row holds every angle twice: slots i and 32 + i are the pair (j, j + dh / 2) of one frequency -- api_esm.hip ensure_rotary)
                f32x4 rc[4], rs[4];
                if (which < 2 && qo.rotary) {
                    const int tr = (min(t, qo.T - 1) * qo.rot_halves + (hh % qo.rot_halves)) * 64;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        rc[g] = *reinterpret_cast<const f32x4*>(qo.cos_t + tr + 8 * g + 4 * kh);
                        rs[g] = *reinterpret_cast<const f32x4*>(qo.sin_t + tr + 8 * g + 4 * kh);
                    }
                }
                if (row_ok)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = 8 * g + 4 * kh;             // dims d0..d0+3 (x0) and d0+32.. (x1) of the head
                    const f32x4 b0 = bq0[g], b1 = bq1[g];
                    float x0[4], x1[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x0[e] = fmaf(acc[0][i][4 * g + e], out_scale, b0[e]);      // explicit fma in every epilogue: left to -ffp-contract, the full- and
                        x1[e] = fmaf(acc[1][i][4 * g + e], out_scale, b1[e]);      // half-height instantiations could round differently
                    }
                    if (which == 0) {                          // attention's softmax is base 2: q carries log2(e) (common.h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { x0[e] *= kQLog2e; x1[e] *= kQLog2e; }
                    }
                    if (which < 2) {
                        if (qo.rotary) {                       // rotary_embedding.py:11-20
                            const f32x4 c1 = rc[g], s1 = rs[g], c2 = rc[g], s2 = rs[g];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float y0 = fmaf(x0[e], c1[e], (-x1[e]) * s1[e]);
                                const float y1 = fmaf(x1[e], c2[e], x0[e] * s2[e]);
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
        } else if constexpr (OUT == 0) {
            // fp32 (+ residual) output through a per-wave LDS transpose, one 32 x 32 accumulator tile at a time.  Patch: 32 rows
            // x 128 B, the 16-byte chunk index XORed with row & 7: the accumulator-order writes (ds_write_b128: 8 rows of one
            // chunk column per group) and the row-order reads (ds_read_b128: four rows of 4 chunks per group) are conflict-free.
            // In row order lane q holds chunk q & 7 of rows (q >> 3) + 8 k: 8 lanes cover one full 128-byte line, a load / store
            // instruction 8 full lines (the accumulator layout gives 32 rows x 32 bytes per instruction: 4 x the line accesses).
            // Loads and stores are BUFFER operations whose offset is pushed out of range for rows >= M / columns >= N: no branch
            // around any of them.  A memory operation under a branch makes the compiler wait for everything in flight at the next
            // use (vmcnt(0), which counts the stores already issued as well): one store round trip per 4 values.
            unsigned char* patch = patches + wave * (32 * 128);
            const int cc = lane & 7, rq = lane >> 3;
            const unsigned int kOob = 0x80000000u;                // the launcher keeps M N 4 below 2^31
            const unsigned int ldc = xm.ldc ? (unsigned int)xm.ldc : (unsigned int)N;
            const long long cb = xm.tiles_per_batch ? (long long)(eb % xm.batch_inner) * xm.c_b0 + (long long)(eb / xm.batch_inner) * xm.c_b1 : 0;
            const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(Cf + cb, 0, (int)((unsigned int)M * ldc * 4u), 0x00020000);
            const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual ? residual : Cf) + cb, 0, (int)((unsigned int)M * ldc * 4u), 0x00020000);
            unsigned int coff[TN];                                // byte offset of the lane's four columns inside a row, or out of range
            f32x4 bv[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = en0 + (wn * TN + j) * 32 + cc * 4;  // N % 4 == 0 is required by the launcher
                coff[j] = n < N ? (unsigned int)n * 4u : kOob;
                bv[j] = (bias && n < N) ? *reinterpret_cast<const f32x4*>(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(bv[j]));                     // waited for HERE, not inside the block loop
            auto c_off = [&](int m, int j) -> int {
                return (int)((m < M && coff[j] != kOob) ? (unsigned int)m * ldc * 4u + coff[j] : kOob);
            };
            // residual rows of block i + 1 are loaded before block i is processed: one memory latency per item, not per block
            u32x4 rv[2][TN][4];
            auto load_residual = [&](int i, u32x4 (&dst)[TN][4]) {
                const int m_base = em0 + (wm * TM + i) * 32;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        dst[j][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsR, c_off(m_base + rq + 8 * k, j), 0, 0));
            };
            if (residual) load_residual(0, rv[0]);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m_base = em0 + (wm * TM + i) * 32;
                if (residual && i + 1 < TM) load_residual(i + 1, rv[(i + 1) & 1]);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4*>(patch + r * 128 + (((2 * g + kh) ^ (r & 7)) << 4)) =
                            f32x4{acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int row = rq + 8 * k;
                        const f32x4 a = *reinterpret_cast<const f32x4*>(patch + row * 128 + ((cc ^ (row & 7)) << 4));
                        f32x4 val;
#pragma unroll
                        for (int e = 0; e < 4; ++e) val[e] = fmaf(a[e], out_scale, bv[j][e]);
                        if (EPI == EPI_GELU) val = gelu_erf16(val);
                        if (EPI == EPI_SQRELU)   // tranception/activations.py:79-84
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float t = fmaxf(val[e], 0.0f); val[e] = t * t; }
                        if (residual) {
                            const f32x4 rr = __builtin_bit_cast(f32x4, rv[i & 1][j][k]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) val[e] = rr[e] + val[e];
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), rsC, c_off(m_base + row, j), 0, 0);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        } else {
            // OUT 1.  Lane holds column m = m_base + r of C^T, rows n = (v&3) + 8(v>>2) + 4kh; split-plane output leaves through a
            // per-wave LDS transpose as full 128-byte row segments.  (Round 4 tried whole 16-byte chunks per lane -- one
            // v_permlane32_swap per dword between the lane pair (r, kh = 0 / 1), ds_write_b128 into an XOR-swizzled 256-byte row,
            // no bank conflicts: FC1 -0.8 % -- and did not keep it: the full- and the half-height instantiations then disagreed in the
            // last bits on rows holding values below fp16's normal range (tests/test_gpu_tranception.py::test_token_logprobs_do_not_…),
            // although every op-level comparison on random data was bit-identical.)
            if constexpr (BF) {
                // bf16 plane out, row-major [M][N] (the next GEMM's operand): the wave's 64 columns are 128 contiguous bytes per row;
                // through a per-wave LDS patch (32 rows x 128 B + 16 B pad) so that 8 lanes store one full line
                constexpr int SPB = 144;
                const bool staged = (N % 8 == 0) && (en0 + (wn * TN + TN) * 32 <= N);
                unsigned char* patch = patches + wave * (32 * SPB);
                f32x4 bvs[TN][4];
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = en0 + (wn * TN + j) * 32 + 8 * g + 4 * kh;
                        bvs[j][g] = (bias && n < N) ? *reinterpret_cast<const f32x4*>(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) asm volatile("" :: "v"(bvs[j][g]));
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = em0 + (wm * TM + i) * 32 + r;
                    const bool row_ok = m < M;
                    if (!staged && !row_ok) continue;
                    if (row_ok)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n = en0 + (wn * TN + j) * 32 + 8 * g + 4 * kh;
                            if (n >= N) continue;
                            const f32x4 bv = bvs[j][g];
                            f32x4 val;
#pragma unroll
                            for (int e = 0; e < 4; ++e) val[e] = fmaf(acc[j][i][4 * g + e], out_scale, bv[e]);
                            if (EPI == EPI_GELU) val = gelu_erf16(val);
                            if (EPI == EPI_SQRELU)
#pragma unroll
                            for (int e = 0; e < 4; ++e) { const float t = fmaxf(val[e], 0.0f); val[e] = t * t; }
                            u32x2 pk;
                            pk[0] = (unsigned int)f32_to_bf16_rne(val[0]) | ((unsigned int)f32_to_bf16_rne(val[1]) << 16);
                            pk[1] = (unsigned int)f32_to_bf16_rne(val[2]) | ((unsigned int)f32_to_bf16_rne(val[3]) << 16);
                            if (staged) *reinterpret_cast<u32x2*>(patch + r * SPB + j * 64 + (8 * g + 4 * kh) * 2) = pk;
                            else *reinterpret_cast<u32x2*>(Ch + (size_t)m * (size_t)N + n) = pk;
                        }
                    if (staged) {
                        __builtin_amdgcn_wave_barrier();
                        const int m_base = em0 + (wm * TM + i) * 32;
                        const size_t ncol0 = (size_t)en0 + (size_t)wn * TN * 32;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int q = lane + 64 * k, row = q >> 3, cc = q & 7;
                            const u32x4 v = *reinterpret_cast<const u32x4*>(patch + row * SPB + cc * 16);
                            if (m_base + row < M) *reinterpret_cast<u32x4*>(Ch + (size_t)(m_base + row) * (size_t)N + ncol0 + cc * 8) = v;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            } else {
            // per-wave LDS patch: 32 rows x (256 B in OUTPUT order: group 0 hi | group 0 lo | group 1 hi | group 1 lo) + 16 B pad
            constexpr int SP = 272;
            const bool staged = (N % 8 == 0) && (en0 + (wn * TN + TN) * 32 <= N);
            unsigned char* patch = patches + wave * (32 * SP);
            // the wave's bias values, once per item (see OUT 2 above)
            f32x4 bvs[TN][4];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = en0 + (wn * TN + j) * 32 + 8 * g + 4 * kh;
                    bvs[j][g] = (bias && n < N) ? *reinterpret_cast<const f32x4*>(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) asm volatile("" :: "v"(bvs[j][g]));              // waited for HERE, not inside the block loop
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
                        const f32x4 bv = bvs[j][g];
                        f32x4 val;
#pragma unroll
                        for (int e = 0; e < 4; ++e) val[e] = fmaf(acc[j][i][4 * g + e], out_scale, bv[e]);
                        if (EPI == EPI_GELU) val = gelu_erf16(val);
                        if (EPI == EPI_SQRELU)   // tranception/activations.py:79-84
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float t = fmaxf(val[e], 0.0f); val[e] = t * t; }
                        {
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
                if (staged) {
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
                        if (m_base + row < M) {
                            if (xm.o_ld)      // scatter (XMap): the wave's 64 columns are one (n / 64) block of the batch's output columns
                                *reinterpret_cast<u32x4*>(Ch + ((size_t)(m_base + row) + (ncol0 >> 6) * (size_t)xm.o_rows_per_n64) * (2 * (size_t)xm.o_ld) +
                                                          (size_t)eb * (size_t)xm.o_col_per_batch * 2 + cc * 8) = v;
                            else
                                *reinterpret_cast<u32x4*>(Ch + (size_t)(m_base + row) * (2 * (size_t)N) + ncol0 * 2 + cc * 8) = v;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            }
        }
        return more;
    };
    using Full = std::integral_constant<int, TMX>;
    using Half = std::integral_constant<int, TMX / 2>;
    while (true) {
        bool more;
        if constexpr (OUT != 2) more = half_item ? run_item(Half{}) : run_item(Full{});
        else more = run_item(Full{});
        if (!more) break;
    }
}

}  // namespace pgmi
