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
__device__ __forceinline__ void step24(f32x16 (&acc)[2][4], u32x4 (&af)[2][4], u32x4 (&wf)[2][2]) {
    u32x4 whs[2];
    const hh2 sc = {(_Float16)(1.0f / 2048.0f), (_Float16)(1.0f / 2048.0f)};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const unsigned u = wf[0][j][e]; whs[j][e] = __builtin_bit_cast(unsigned, __builtin_bit_cast(hh2, u) * sc); }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = mf(wf[1][j], af[0][i], acc[j][i]);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = mf(whs[j], af[1][i], acc[j][i]);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = mf(wf[0][j], af[0][i], acc[j][i]);
#pragma unroll
    for (int k = 0; k < 8; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); }
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
}

// FC1-like epilogue of one wave: 128 values per lane through scale + bias, GELU, split; 32 dwordx4 stores per wave (lane-contiguous:
// every store instruction writes 1 KB), accumulators cleared.
__device__ __forceinline__ void epilogue(f32x16 (&acc)[2][4], unsigned short* out, int lane, float scale) {
    u32x4* dst = reinterpret_cast<u32x4*>(out) + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[j][i][4 * g + e], scale, 0.01f * e);
                v = gelu_erf16(v);
                _Float16 h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split_act(v[e], h[e], l[e]);
                const hh2 a = {h[0], h[1]}, b = {h[2], h[3]}, c = {l[0], l[1]}, d = {l[2], l[3]};
                dst[((i * 2 + j) * 4 + g) * 64] = u32x4{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, c),
                                                         __builtin_bit_cast(unsigned, d)};
            }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
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
template <int EPI, int CW>
__global__ __launch_bounds__(512, 2) void pingpong8(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, unsigned short* out,
                                                    int n_items, float* sink) {
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];                  // [2][4096]
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
    decode(item);
    issue(0, 0);
    float keep = 0.f;
    bool first = true;
    while (true) {
        if (CW && EPI && !first) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x8070);                                   // vmcnt(32) lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        } else {
            phase(true);
        }
        first = false;
        int cur = 0;
        if (late) phase(false);
        for (int kt = 0; kt < NKT; ++kt) {
            const u32x4* Ab = lds + cur * 4096;
            const u32x4* Wb = Ab + 2048;
            __builtin_amdgcn_s_setprio(0);
            if (kt + 1 < NKT) issue(kt + 1, cur ^ 1);
            frags(Ab, Wb, 0);
            phase(false);
            __builtin_amdgcn_s_setprio(1);
            step24(acc, af, wf);
            phase(false);
            __builtin_amdgcn_s_setprio(0);
            frags(Ab, Wb, 1);
            phase(late);
            __builtin_amdgcn_s_setprio(1);
            step24(acc, af, wf);
            phase(!late);
            cur ^= 1;
        }
        __builtin_amdgcn_s_setprio(0);
        if (!late) phase(false);
        const int done = item;
        item += gridDim.x;
        const bool more = item < n_items;
        if (more) { decode(item); issue(0, 0); }
        if (EPI) epilogue(acc, out + ((size_t)done * 8 + wave) * 16384, lane, 1e-4f);
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
            step24(acc, af, wf);
            __builtin_amdgcn_s_setprio(0);
            slot = slot == 2 ? 0 : slot + 1;
        }
        first = false;
        const int done = item;
        item += gridDim.x;
        const bool more = item < n_items;
        __builtin_amdgcn_s_barrier();                                             // every wave is through its last fragment reads: the ring restarts at slot 0
        if (more) { decode(item); issue(0, 0); issue(1, 1); }
        if (EPI) epilogue(acc, out + ((size_t)done * 4 + wave) * 16384, lane, 1e-4f);
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
    auto pp0 = pingpong8<0, 0>; auto pp1 = pingpong8<1, 0>; auto pp2 = pingpong8<1, 1>;
    auto d00 = duo4<0, 0>; auto d01 = duo4<0, 1>; auto d10 = duo4<1, 0>; auto d11 = duo4<1, 1>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(pp0), hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(pp1), hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(pp2), hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
    for (auto f : {reinterpret_cast<const void*>(d00), reinterpret_cast<const void*>(d01), reinterpret_cast<const void*>(d10), reinterpret_cast<const void*>(d11)})
        hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    const double flops = 2.0 * M_ROWS * (double)N_ROWS * K_ELEMS;             // algorithmic (one product per element pair)
    constexpr int NV = 7;
    const char* names[NV] = {"pingpong8 no epilogue", "duo4 contiguous no epilogue", "duo4 split-32B no epilogue", "pingpong8 + FC1-like epilogue",
                             "pingpong8 + epilogue, counted wait", "duo4 contiguous + epilogue", "duo4 split-32B + epilogue"};
    printf("FC1 shape %d x %d x %d: %d tiles of 256 x 256 / %d tiles of 128 x 256, %d CUs\n", M_ROWS, N_ROWS, K_ELEMS, items8, items4, cus);
    std::vector<std::vector<float>> ms(NV);
    for (int r = 0; r <= rounds; ++r) {
        float t[NV];
        t[0] = time_one(pp0, dim3(cus), dim3(512), 136 * 1024, A, W, out, items8, sink);
        t[1] = time_one(d00, dim3(2 * cus), dim3(256), 80 * 1024, A, W, out, items4, sink);
        t[2] = time_one(d01, dim3(2 * cus), dim3(256), 80 * 1024, A, W, out, items4, sink);
        t[3] = time_one(pp1, dim3(cus), dim3(512), 136 * 1024, A, W, out, items8, sink);
        t[4] = time_one(pp2, dim3(cus), dim3(512), 136 * 1024, A, W, out, items8, sink);
        t[5] = time_one(d10, dim3(2 * cus), dim3(256), 80 * 1024, A, W, out, items4, sink);
        t[6] = time_one(d11, dim3(2 * cus), dim3(256), 80 * 1024, A, W, out, items4, sink);
        if (r == 0) continue;                                                  // warm-up round
        for (int k = 0; k < NV; ++k) ms[k].push_back(t[k]);
        printf("round %d:", r);
        for (int k = 0; k < NV; ++k) printf("  %.3f ms = %.1f", t[k], flops / t[k] / 1e9);
        printf("  TFLOP/s\n");
        fflush(stdout);
    }
    for (int k = 0; k < NV; ++k) {
        std::sort(ms[k].begin(), ms[k].end());
        const float med = ms[k][ms[k].size() / 2];
        printf("%-34s median %.3f ms = %.1f TFLOP/s algorithmic\n", names[k], med, flops / med / 1e9);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("HIP error: %s\n", hipGetErrorString(e)); return 1; }
    return 0;
}
