// Probe (round 5): can TWO independent 4-wave workgroups per CU -- one wave of each per SIMD, naturally out of step, so that one's
// epilogue runs under the other's K loop -- beat the ONE 8-wave ping-pong workgroup of gemm16x_kernel on the f16x3 GEMM's work?
//
// Both kernels do the GEMM's real memory work on synthetic operands (K-interleaved rows of K = 1280 elements = 5 120 bytes, DMA straight
// into LDS, the fragment reads, the 3 x 8 MFMAs per k16 step with the eight v_pk_mul_f16 riding on them) and walk a list of items
// (tiles) of `nk` K tiles each; after every item an FC1-like epilogue (scale + bias, erf-GELU, fp16 split, 32 dwordx4 stores per wave:
// the real code of gemm16x_kernel.h / common.h on the accumulators, results stored lane-contiguously) unless EPI == 0.
//   pingpong8: 512 threads, 256 x 256 tile, two 64 KB stages, 4 barriers per K tile (the product's loop)
//   duo4:      256 threads, 128 x 256 tile (wave tile 128 x 64 as in the product), a 3-slot ring of K16 half-stages (3 x 24 KB = 72 KB,
//              80 KB with the patch space: two workgroups per CU), ONE barrier per K16 half-stage; SPLIT = 0: a row's K16 share is one
//              contiguous 64-byte segment (what a [16 hi | 16 lo] layout would give), SPLIT = 1: the present [32 hi | 32 lo] layout,
//              i.e. two 32-byte pieces 64 bytes apart per row and half-stage (every 128-byte line requested by two instructions)
// Prints algorithmic TFLOP/s (MFMA FLOPs / 3) of every variant, interleaved round by round on the same box.
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mfma_duo tools/mfma_duo.hip && tools/mfma_duo [rounds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>

#include "../proteingym_amd/csrc/gemm16x_kernel.h"     // gelu_erf16, split_act (common.h), typedefs; no kernel is instantiated from it

using namespace pgmi;
typedef _Float16 hh2 __attribute__((ext_vector_type(2)));

constexpr int K_ELEMS = 1280, ROW_BYTES = K_ELEMS * 4, NKT = K_ELEMS / 32;      // 40 K tiles per item
constexpr int M_ROWS = 82368, N_ROWS = 5120;                                    // FC1 of the BLAT-shaped step

__device__ __forceinline__ f32x16 mf(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// the wave's 24 MFMAs of one k16 step (gemm16x_kernel.h mfmas()): w_lo a_hi, (w_hi 2^-11) a_lo, w_hi a_hi
template <int TM>
__device__ __forceinline__ void step24(f32x16 (&acc)[2][TM], u32x4 (&af)[2][TM], u32x4 (&wf)[2][2]) {
    u32x4 whs[2];
    const hh2 sc = {(_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f)};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const unsigned u = wf[0][j][e]; whs[j][e] = __builtin_bit_cast(unsigned, __builtin_bit_cast(hh2, u) * sc); }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = mf(wf[1][j], af[0][i], acc[j][i]);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = mf(whs[j], af[1][i], acc[j][i]);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = mf(wf[0][j], af[0][i], acc[j][i]);
#pragma unroll
    for (int k = 0; k < 8; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); }
    __builtin_amdgcn_sched_group_barrier(0x008, 6 * TM - 8, 0);
}

// f16 / f64 denormal mode of THIS wave = flush (MODE.FP_DENORM[3:2] = 0): v_cvt_pk_f16_f32 then returns +-0 for results below fp16's
// normal range, which is what split_act's compare + select does by hand
__device__ __forceinline__ void f16_denorm_flush() { __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0); }

// split_act for a pair, without the compare + select (needs f16_denorm_flush): hi = fp16(x) packed, lo = fp16((x - hi) 2^11) through one
// packed multiply and a mixed-precision fma per value (x 2^11 and the fma are exact)
__device__ __forceinline__ void split_pair(float x0, float x1, hh2& hi, hh2& lo) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    hi = hh2{(_Float16)x0, (_Float16)x1};
    const f2 xs = f2{x0, x1} * f2{2048.0f, 2048.0f};
    const float l0 = fmaf((float)hi[0], -2048.0f, xs[0]), l1 = fmaf((float)hi[1], -2048.0f, xs[1]);
    lo = hh2{(_Float16)l0, (_Float16)l1};
}

__global__ void split_selftest(const float* x, int n, unsigned short* hi_out, unsigned short* lo_out, unsigned short* hi_ref, unsigned short* lo_ref) {
    f16_denorm_flush();
    const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 1 >= n) return;
    hh2 h, l;
    split_pair(x[i], x[i + 1], h, l);
    hi_out[i] = __builtin_bit_cast(unsigned short, h[0]); hi_out[i + 1] = __builtin_bit_cast(unsigned short, h[1]);
    lo_out[i] = __builtin_bit_cast(unsigned short, l[0]); lo_out[i + 1] = __builtin_bit_cast(unsigned short, l[1]);
    _Float16 a, b;
    split_act(x[i], a, b); hi_ref[i] = __builtin_bit_cast(unsigned short, a); lo_ref[i] = __builtin_bit_cast(unsigned short, b);
    split_act(x[i + 1], a, b); hi_ref[i + 1] = __builtin_bit_cast(unsigned short, a); lo_ref[i + 1] = __builtin_bit_cast(unsigned short, b);
}

// FC1-like epilogue of one wave: 128 values per lane through scale + bias, GELU, split; 32 dwordx4 stores per wave (lane-contiguous:
// every store instruction writes 1 KB), accumulators cleared.
// FAST 0: split_act; 1: packed split; 3: the arithmetic only (stores behind a flag that is never set); 4: the stores only (raw accumulators)
template <int FAST, int TM = 4>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[2][TM], unsigned short* out, int lane, float scale, bool do_store = true) {
    u32x4* dst = reinterpret_cast<u32x4*>(out) + lane;
    if (FAST == 4) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    dst[((i * 2 + j) * 4 + g) * 64] = u32x4{__builtin_bit_cast(unsigned, acc[j][i][4 * g]), __builtin_bit_cast(unsigned, acc[j][i][4 * g + 1]),
                                                             __builtin_bit_cast(unsigned, acc[j][i][4 * g + 2]), __builtin_bit_cast(unsigned, acc[j][i][4 * g + 3])};
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[j][i][v] = 0.f;
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[j][i][4 * g + e], scale, 0.01f * e);
                if (FAST != 5 && FAST != 7) v = gelu_erf16(v);
                if (FAST == 7) {                                               // gelu_erf16 with its four v_exp_f32 replaced by fmas (wrong values, same shape)
                    auto all = [](float t) { return f32x4{t, t, t, t}; };
                    const f32x4 bq = __builtin_elementwise_max(-__builtin_elementwise_abs(v), all(-6.0f));
                    f32x4 q = __builtin_elementwise_fma(v, all(0.0f), all(3.309269596e-05f));
                    q = __builtin_elementwise_fma(q, bq, all(7.692188374e-04f));
                    q = __builtin_elementwise_fma(q, bq, all(8.080714382e-03f));
                    q = __builtin_elementwise_fma(q, bq, all(5.341210216e-02f));
                    q = __builtin_elementwise_fma(q, bq, all(-4.587709904e-01f));
                    q = __builtin_elementwise_fma(q, bq, all(1.151201725e+00f));
                    q = __builtin_elementwise_fma(q, bq, all(-9.999930859e-01f));
                    q = __builtin_elementwise_fma(q, bq, all(0.25f));
                    v = __builtin_elementwise_fma(bq, q, __builtin_elementwise_max(v, all(0.0f)));
                }
                hh2 a, b, c, d;
                if (FAST == 6) {
                    a = hh2{(_Float16)v[0], (_Float16)v[1]}; b = hh2{(_Float16)v[2], (_Float16)v[3]}; c = a; d = b;
                } else if (FAST == 1) {
                    split_pair(v[0], v[1], a, c);
                    split_pair(v[2], v[3], b, d);
                } else {
                    _Float16 h[4], l[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) split_act(v[e], h[e], l[e]);
                    a = hh2{h[0], h[1]}; b = hh2{h[2], h[3]}; c = hh2{l[0], l[1]}; d = hh2{l[2], l[3]};
                }
                const u32x4 pk = u32x4{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d)};
                if (FAST == 3 || FAST >= 5) { asm volatile("" :: "v"(pk)); if (do_store) dst[((i * 2 + j) * 4 + g) * 64] = pk; }
                else dst[((i * 2 + j) * 4 + g) * 64] = pk;
            }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[j][i][v] = 0.f;
}

// item -> (row panel of `bm` rows, column panel of 256): grouped order, 8 row panels per group (gemm_f16.hip kGroupMWide), XCD-aware
__device__ __forceinline__ void item_coords(int item, int n_items, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int xcd = item & 7, q = n_items >> 3, r8 = n_items & 7;
    const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (item >> 3);
    const int G = 8, width = G * tiles_n, group = wgid / width, first = group * G, gsz = min(tiles_m - first, G);
    tm = first + (wgid % width) % gsz;
    tn = (wgid % width) / gsz;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// CW = 1: the item's first wait is COUNTED -- vmcnt(32): the eight DMAs of its first K tile were issued before the previous epilogue's 32
// stores, so they have landed once at most 32 operations are outstanding -- instead of the product's vmcnt(0), which waits for the stores
// VAR 2 / 3: the CU slots of an XCD start 1 / 3 us apart (slot = blockIdx.x / 8 mod 32): do the epilogues' store bursts cost less when the 32 CUs behind
// one L2-fabric port do not burst together?   VAR 4: the next item's first TWO K tiles are in flight before the epilogue's stores (the probe has no LDS
// patches, both stages are free), with counted waits across the stores: the stores get two K tiles of time to drain before anybody waits for them.
template <int EPI, int CW>
__global__ __launch_bounds__(512, 2) void pingpong8(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, unsigned short* out,
                                                    int n_items, float* sink) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];                  // [2][4096]
    if (EPI == 2) f16_denorm_flush();
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = wave >= 4;
    const int wm = wave >> 2, wn = wave & 3, r = lane & 31, kh = lane >> 5, fsw = (r >> 1) & 7;
    const int tiles_m = (M_ROWS + 255) / 256, tiles_n = N_ROWS / 256;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A), 0, (int)((unsigned)M_ROWS * ROW_BYTES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, (int)((unsigned)N_ROWS * ROW_BYTES), 0x00020000);
    const int row_lo = tid >> 3, c8 = tid & 7, csrc = c8 ^ ((row_lo >> 1) & 7);
    unsigned a_off[4], w_off[4];
    auto decode = [&](int item) {
        int tm, tn;
        item_coords(item, n_items, tiles_m, tiles_n, tm, tn);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a_off[i] = (unsigned)min(tm * 256 + row_lo + 64 * i, M_ROWS - 1) * ROW_BYTES + csrc * 16u;
            w_off[i] = (unsigned)min(tn * 256 + row_lo + 64 * i, N_ROWS - 1) * ROW_BYTES + csrc * 16u;
        }
    };
    auto issue = [&](int kt, int buf) {
        u32x4* base = lds + buf * 4096 + wave * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + 512 * i), 16, (int)a_off[i], kt * 128, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(base + 2048 + 512 * i), 16, (int)w_off[i], kt * 128, 0, 0);
    };
    auto phase = [&](bool vm) {
        __builtin_amdgcn_sched_barrier(0);
        if (vm) __builtin_amdgcn_s_waitcnt(0x0070); else __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x16 acc[2][4];
    u32x4 af[2][4], wf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[j][i][v] = 0.f;
    auto frags = [&](const u32x4* Ab, const u32x4* Wb, int ks) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int c = (p * 4 + ks * 2 + kh) ^ fsw;
#pragma unroll
            for (int i = 0; i < 4; ++i) af[p][i] = Ab[((wm * 4 + i) * 32 + r) * 8 + c];
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[p][j] = Wb[((wn * 2 + j) * 32 + r) * 8 + c];
        }
    };
    int item = blockIdx.x;
    if (item >= n_items) return;
    if (CW == 2 || CW == 3) {                                                    // 100 MHz wall clock
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(((blockIdx.x >> 3) & 31) * (CW == 2 ? 100 : 300));
        while (__builtin_amdgcn_s_memrealtime() < t_end) __builtin_amdgcn_s_sleep(8);
    }
    decode(item);
    issue(0, 0);
    float keep = 0.f;
    bool first = true;
    while (true) {
        if (CW == 4 && EPI && !first) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x8078);                                   // vmcnt(40) lgkmcnt(0): tile 0 landed; tile 1's eight DMAs and the 32 stores may be out
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        } else if (CW == 1 && EPI && !first) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x8070);                                   // vmcnt(32) lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        } else {
            phase(true);
        }
        const bool was_first = first;
        first = false;
        int cur = 0;
        if (late) phase(false);
        for (int kt = 0; kt < NKT; ++kt) {
            const u32x4* Ab = lds + cur * 4096;
            const u32x4* Wb = Ab + 2048;
            const bool ahead = CW == 4 && EPI && !was_first && kt == 0;          // tile 1 of this item is already in flight, behind it the stores
            __builtin_amdgcn_s_setprio(0);
            if (kt + 1 < NKT && !ahead) issue(kt + 1, cur ^ 1);
            frags(Ab, Wb, 0);
            phase(false);
            __builtin_amdgcn_s_setprio(1);
            step24<4>(acc, af, wf);
            phase(false);
            __builtin_amdgcn_s_setprio(0);
            frags(Ab, Wb, 1);
            auto close = [&](bool vm) {                                           // the wave group's last barrier of the K tile
                if (vm && ahead) {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_waitcnt(0x8070);                           // vmcnt(32): tile 1 landed, the stores may still be out
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    phase(vm);
                }
            };
            close(late);
            __builtin_amdgcn_s_setprio(1);
            step24<4>(acc, af, wf);
            close(!late);
            cur ^= 1;
        }
        __builtin_amdgcn_s_setprio(0);
        if (!late) phase(false);
        const int done = item;
        item += gridDim.x;
        const bool more = item < n_items;
        if (more) { decode(item); issue(0, 0); if (CW == 4 && EPI) issue(1, 1); }
        if (EPI) epilogue<(EPI == 2 ? 1 : EPI == 1 ? 0 : EPI)>(acc, out + ((size_t)done * 8 + wave) * 16384, lane, 1e-4f, n_items < 0);
        else { keep += acc[0][0][0]; }
        if (!more) break;
    }
    if (!EPI) {                                                                  // every accumulator stays live
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int v = 0; v < 16; ++v) keep += acc[j][i][v];
        sink[blockIdx.x * 512 + tid] = keep;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
template <int EPI, int SPLIT>
__global__ __launch_bounds__(256, 2) void duo4(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, unsigned short* out,
                                               int n_items, float* sink) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];                  // [3][1536] chunks (+ patch space up to 80 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave, r = lane & 31, kh = lane >> 5, fsw = (r >> 2) & 3;
    const int tiles_m = (M_ROWS + 127) / 128, tiles_n = N_ROWS / 256;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A), 0, (int)((unsigned)M_ROWS * ROW_BYTES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, (int)((unsigned)N_ROWS * ROW_BYTES), 0x00020000);
    // staging: instruction i of wave w fills chunks (4 i + w) 64 .. + 63 of the half-stage = 16 rows x 4 chunks; lane = (row >> 0 & 15, chunk c4)
    const int row_in = lane >> 2, c4 = lane & 3;
    unsigned a_off[2], w_off[4];
    auto decode = [&](int item) {
        int tm, tn;
        item_coords(item, n_items, tiles_m, tiles_n, tm, tn);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 16 * (4 * i + wave) + row_in;                         // 0 .. 127
            const int hc = c4 ^ ((row >> 2) & 3);
            const unsigned within = SPLIT ? (unsigned)((hc >> 1) * 64 + (hc & 1) * 16) : (unsigned)(hc * 16);
            a_off[i] = (unsigned)min(tm * 128 + row, M_ROWS - 1) * ROW_BYTES + within;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 16 * (4 * i + wave) + row_in;                         // 0 .. 255
            const int hc = c4 ^ ((row >> 2) & 3);
            const unsigned within = SPLIT ? (unsigned)((hc >> 1) * 64 + (hc & 1) * 16) : (unsigned)(hc * 16);
            w_off[i] = (unsigned)min(tn * 256 + row, N_ROWS - 1) * ROW_BYTES + within;
        }
    };
    auto issue = [&](int h, int slot) {                                           // half-stage h of the item: K tile h >> 1, k16 step h & 1
        u32x4* base = lds + slot * 1536 + wave * 64;
        const int so = (h >> 1) * 128 + (h & 1) * (SPLIT ? 32 : 64);
#pragma unroll
        for (int i = 0; i < 2; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + 256 * i), 16, (int)a_off[i], so, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(base + 512 + 256 * i), 16, (int)w_off[i], so, 0, 0);
    };
    f32x16 acc[2][4];
    u32x4 af[2][4], wf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[j][i][v] = 0.f;
    auto frags = [&](const u32x4* Ab, const u32x4* Wb) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int c = (p * 2 + kh) ^ fsw;
#pragma unroll
            for (int i = 0; i < 4; ++i) af[p][i] = Ab[(i * 32 + r) * 4 + c];
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[p][j] = Wb[((wn * 2 + j) * 32 + r) * 4 + c];
        }
    };
    int item = blockIdx.x;
    if (item >= n_items) return;
    decode(item);
    issue(0, 0);
    issue(1, 1);
    float keep = 0.f;
    constexpr int NH = 2 * NKT;
    bool first = true;
    while (true) {
        int slot = 0;
        for (int h = 0; h < NH; ++h) {
            // this wave's share of half-stage h has landed (the six DMAs of h + 1 may still be in flight); after the barrier everybody's has,
            // and everybody has read half-stage h - 1 (its fragment reads were waited for before its MFMAs): its slot takes h + 2.
            // Half-stages 0 and 1 of an item were issued BEFORE the previous item's 32 epilogue stores: counted across them (vmcnt(38)).
            __builtin_amdgcn_sched_barrier(0);
            if (EPI && !first && h < 2) __builtin_amdgcn_s_waitcnt(0x8F76);       // vmcnt(38)
            else if (h + 1 < NH) __builtin_amdgcn_s_waitcnt(0x0F76);             // vmcnt(6)
            else __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (h + 2 < NH) issue(h + 2, slot == 0 ? 2 : slot - 1);
            const u32x4* Ab = lds + slot * 1536;
            frags(Ab, Ab + 512);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);                                   // lgkmcnt(0)
            __builtin_amdgcn_s_setprio(1);
            step24<4>(acc, af, wf);
            __builtin_amdgcn_s_setprio(0);
            slot = slot == 2 ? 0 : slot + 1;
        }
        first = false;
        const int done = item;
        item += gridDim.x;
        const bool more = item < n_items;
        __builtin_amdgcn_s_barrier();                                             // every wave is through its last fragment reads: the ring restarts at slot 0
        if (more) { decode(item); issue(0, 0); issue(1, 1); }
        if (EPI) epilogue<(EPI == 2 ? 1 : 0)>(acc, out + ((size_t)done * 4 + wave) * 16384, lane, 1e-4f);
        else { keep += acc[0][0][0]; }
        if (!more) break;
    }
    if (!EPI) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int v = 0; v < 16; ++v) keep += acc[j][i][v];
        sink[blockIdx.x * 256 + tid] = keep;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// duo8: TWO 8-wave ping-pong workgroups per CU -- 128 x 256 tiles, wave tile 64 x 64 (TM 2 x TN 2: 64 accumulator registers, 128 VGPRs:
// four waves per SIMD, two of each workgroup), each workgroup with the product's own ping-pong inside (waves 4-7 one phase behind waves 0-3,
// 12 MFMAs per compute phase, K16 half-stages in a 3-slot ring = 72 KB).  A workgroup ALONE can keep the matrix pipe busy (unlike duo4's lone
// wave per SIMD): while one is in its epilogue the other has the pipe to itself.  Price: twice the fragment reads per MFMA, half-length phases.
template <int EPI>
__global__ __launch_bounds__(512, 4) void duo8(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, unsigned short* out,
                                               int n_items, float* sink) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];                  // [3][1536] chunks
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = wave >= 4;
    const int wm = (wave >> 1) & 1, wn = (wave & 1) | ((wave >> 2) << 1);       // early waves 0-3 and late waves 4-7 each cover 2 x 2 of the 2 (M) x 4 (N) wave tiles
    const int r = lane & 31, kh = lane >> 5, fsw = (r >> 2) & 3;
    const int tiles_m = (M_ROWS + 127) / 128, tiles_n = N_ROWS / 256;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A), 0, (int)((unsigned)M_ROWS * ROW_BYTES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, (int)((unsigned)N_ROWS * ROW_BYTES), 0x00020000);
    // staging: the half-stage's 24 wave-instructions (8 for A's 128 rows, 16 for W's 256 rows; 16 rows x 4 chunks each); wave w issues w, w + 8, w + 16
    const int row_in = lane >> 2, c4 = lane & 3;
    unsigned off[3];
    auto decode = [&](int item) {
        int tm, tn;
        item_coords(item, n_items, tiles_m, tiles_n, tm, tn);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int q = wave + 8 * i;
            const bool isA = q < 8;
            const int row = 16 * (isA ? q : q - 8) + row_in;
            const int hc = c4 ^ ((row >> 2) & 3);
            const unsigned within = (unsigned)((hc >> 1) * 64 + (hc & 1) * 16);
            off[i] = isA ? (unsigned)min(tm * 128 + row, M_ROWS - 1) * ROW_BYTES + within : (unsigned)min(tn * 256 + row, N_ROWS - 1) * ROW_BYTES + within;
        }
    };
    auto issue = [&](int h, int slot) {
        u32x4* base = lds + slot * 1536 + wave * 64;
        const int so = (h >> 1) * 128 + (h & 1) * 32;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base), 16, (int)off[0], so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(base + 512), 16, (int)off[1], so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(base + 1024), 16, (int)off[2], so, 0, 0);
    };
    f32x16 acc[2][2];
    u32x4 af[2][2], wf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[j][i][v] = 0.f;
    auto frags = [&](int slot) {
        const u32x4* Ab = lds + slot * 1536;
        const u32x4* Wb = Ab + 512;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int c = (p * 2 + kh) ^ fsw;
#pragma unroll
            for (int i = 0; i < 2; ++i) af[p][i] = Ab[((wm * 2 + i) * 32 + r) * 4 + c];
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[p][j] = Wb[((wn * 2 + j) * 32 + r) * 4 + c];
        }
    };
    // phase ends: every wave's LDS traffic has landed; at the end of ODD global phases also this wave's share of the NEXT half-stage (at most
    // `allow` younger operations may still be out: the three DMAs of the half-stage after it, and behind an epilogue its 16 stores)
    auto bar_lgkm = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto bar_vm = [&](int allow) {
        __builtin_amdgcn_sched_barrier(0);
        if (allow >= 19) __builtin_amdgcn_s_waitcnt(0x4073);                     // vmcnt(19) lgkmcnt(0)
        else if (allow >= 3) __builtin_amdgcn_s_waitcnt(0x0073);                 // vmcnt(3) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0x0070);                                 // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    int item = blockIdx.x;
    if (item >= n_items) return;
    decode(item);
    issue(0, 0);
    issue(1, 1);
    float keep = 0.f;
    constexpr int NH = 2 * NKT;
    bool first = true;
    while (true) {
        // half-stage 0 landed everywhere (half-stage 1 and, behind an epilogue, its 16 stores may be out)
        bar_vm((EPI && !first) ? 19 : 3);
        if (late) bar_lgkm();                                                     // late waves run one phase behind
        int slot = 0;
        for (int h = 0; h < NH; ++h) {
            const int behind = (EPI && !first && h == 0) ? 19 : (h + 2 < NH ? 3 : 0);   // what may be out when half-stage h + 1 must have landed
            // -- memory phase: this wave's share of half-stage h + 2 into the slot half-stage h - 1 left, then the fragments of h --
            __builtin_amdgcn_s_setprio(0);
            if (h + 2 < NH) issue(h + 2, slot == 0 ? 2 : slot - 1);
            frags(slot);
            if (late) bar_vm(behind); else bar_lgkm();                            // late: an odd global phase ends here
            // -- compute phase --
            __builtin_amdgcn_s_setprio(1);
            step24<2>(acc, af, wf);
            if (late) bar_lgkm(); else bar_vm(behind);                            // early: an odd global phase ends here
            slot = slot == 2 ? 0 : slot + 1;
        }
        __builtin_amdgcn_s_setprio(0);
        if (!late) bar_lgkm();                                                    // realign: every wave is through its last fragment reads
        first = false;
        const int done = item;
        item += gridDim.x;
        const bool more = item < n_items;
        if (more) { decode(item); issue(0, 0); issue(1, 1); }
        if (EPI) epilogue<0, 2>(acc, out + ((size_t)done * 8 + wave) * 8192, lane, 1e-4f);
        else { keep += acc[0][0][0]; }
        if (!more) break;
    }
    if (!EPI) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int v = 0; v < 16; ++v) keep += acc[j][i][v];
        sink[blockIdx.x * 512 + tid] = keep;
    }
}

template <typename Kfn>
static float time_one(Kfn kfn, dim3 grid, dim3 block, size_t lds, const unsigned short* A, const unsigned short* W, unsigned short* out, int n_items, float* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kfn, grid, block, lds, 0, A, W, out, n_items, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 5;
    unsigned short *A, *W, *out; float* sink;
    const size_t a_bytes = (size_t)M_ROWS * ROW_BYTES, w_bytes = (size_t)N_ROWS * ROW_BYTES;
    const int items8 = ((M_ROWS + 255) / 256) * (N_ROWS / 256), items4 = ((M_ROWS + 127) / 128) * (N_ROWS / 256);
    hipMalloc(&A, a_bytes); hipMalloc(&W, w_bytes); hipMalloc(&sink, 512 * 512 * 4);
    hipMalloc(&out, (size_t)items8 * 8 * 16384 * 2 + (1 << 20));
    std::vector<unsigned short> h(1 << 23);
    srand(1);
    for (auto& v : h) { const unsigned e = 12 + rand() % 4; v = (unsigned short)(((rand() & 1) << 15) | (e << 10) | (rand() & 0x3ff)); }
    for (size_t o = 0; o < a_bytes; o += h.size() * 2) hipMemcpy((char*)A + o, h.data(), std::min(h.size() * 2, a_bytes - o), hipMemcpyHostToDevice);
    for (size_t o = 0; o < w_bytes; o += h.size() * 2) hipMemcpy((char*)W + o, h.data(), std::min(h.size() * 2, w_bytes - o), hipMemcpyHostToDevice);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount - prop.multiProcessorCount % 8;
    auto d00 = duo4<0, 0>; auto d10 = duo4<1, 0>; auto d11 = duo4<1, 1>;
    for (auto f : {reinterpret_cast<const void*>(d00), reinterpret_cast<const void*>(d10), reinterpret_cast<const void*>(d11)})
        hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    typedef void (*Kfn)(const unsigned short*, const unsigned short*, unsigned short*, int, float*);
    struct Variant { const char* name; Kfn fn; int duo; };
    const Variant vs[] = {
        {"pingpong8 no epilogue", pingpong8<0, 0>, 0},
        {"pingpong8 + FC1-like epilogue", pingpong8<1, 0>, 0},
        {"pingpong8 + epilogue, arithmetic only (no stores)", pingpong8<3, 0>, 0},
        {"pingpong8 + epilogue, stores only (raw accumulators)", pingpong8<4, 0>, 0},
        {"pingpong8 arithmetic only: scale + bias + split, no GELU", pingpong8<5, 0>, 0},
        {"pingpong8 arithmetic only: GELU, plain conversion (no split)", pingpong8<6, 0>, 0},
        {"pingpong8 arithmetic only: GELU without v_exp_f32, split", pingpong8<7, 0>, 0},
        {"pingpong8 + epilogue, counted first wait", pingpong8<1, 1>, 0},
        {"pingpong8 + epilogue, packed split (no compare)", pingpong8<2, 0>, 0},
        {"pingpong8 + epilogue, CU slots 1 us apart", pingpong8<1, 2>, 0},
        {"pingpong8 + epilogue, two K tiles in flight before the stores", pingpong8<1, 4>, 0},
        {"duo4 contiguous no epilogue", d00, 1},
        {"duo4 contiguous + epilogue", d10, 1},
        {"duo4 split-32B + epilogue", d11, 1},
        {"duo8 (two 8-wave ping-pong workgroups per CU, 64 x 64 wave tiles) no epilogue", duo8<0>, 2},
        {"duo8 + FC1-like epilogue", duo8<1>, 2},
    };
    constexpr int NV = sizeof(vs) / sizeof(vs[0]);
    for (const Variant& v : vs)
        hipFuncSetAttribute(reinterpret_cast<const void*>(v.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (v.duo ? 80 : 136) * 1024);
    const double flops = 2.0 * M_ROWS * (double)N_ROWS * K_ELEMS;             // algorithmic (one product per element pair)
    printf("FC1 shape %d x %d x %d: %d tiles of 256 x 256 / %d tiles of 128 x 256, %d CUs\n", M_ROWS, N_ROWS, K_ELEMS, items8, items4, cus);
    std::vector<std::vector<float>> ms(NV);
    for (int r = 0; r <= rounds; ++r) {
        float t[NV];
        for (int k = 0; k < NV; ++k)
            t[k] = vs[k].duo ? time_one(vs[k].fn, dim3(2 * cus), dim3(vs[k].duo == 1 ? 256 : 512), 80 * 1024, A, W, out, items4, sink)
                             : time_one(vs[k].fn, dim3(cus), dim3(512), 136 * 1024, A, W, out, items8, sink);
        if (r == 0) continue;                                                  // warm-up round
        for (int k = 0; k < NV; ++k) ms[k].push_back(t[k]);
        printf("round %d:", r);
        for (int k = 0; k < NV; ++k) printf(" %.1f", flops / t[k] / 1e9);
        printf("  TFLOP/s\n");
        fflush(stdout);
    }
    for (int k = 0; k < NV; ++k) {
        std::sort(ms[k].begin(), ms[k].end());
        const float med = ms[k][ms[k].size() / 2];
        printf("%-64s median %.3f ms = %.1f TFLOP/s algorithmic\n", vs[k].name, med, flops / med / 1e9);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("HIP error: %s\n", hipGetErrorString(e)); return 1; }
    return 0;
}
