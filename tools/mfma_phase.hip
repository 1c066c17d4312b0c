// Probe: what does each ingredient of the ping-pong GEMM's K loop cost?  One workgroup of 8 waves per CU, the two waves of a
// SIMD alternate memory / compute phases across s_barrier exactly like gemm16x_kernel (4 barriers per K tile, 24 MFMAs per
// compute phase: ideal 768 clocks per phase).
//   mode 0: one wave per SIMD, 24 MFMAs (8 accumulators x 3) back to back, no barriers
//   mode 1: ping-pong skeleton, no memory work
//   mode 3: + the 8 v_pk_mul_f16 interleaved with the first 8 MFMAs and s_setprio around the phases
//   mode 4: + 12 ds_read_b128 per memory phase (the GEMM's fragment addresses, conflict-free)
//   mode 5: + 8 buffer_load_dwordx4 ... lds per wave per K tile in memory phase 1, vmcnt(0) at the tile's last barrier;
//             all workgroups stream the same 64 KiB per K tile (L2 hits after the first toucher)
//   mode 6: as 5, but every workgroup streams its own data (HBM)
//   mode 7: as 5, but the 8 DMA instructions are issued 4 in memory phase 1 and 4 in memory phase 2 (every phase of the
//             workgroup then carries 16 KiB of DMA instead of 32 / 32 / 0 / 0)
//   mode 9: DMA as in 5 but NO fragment reads; mode 10: as 9 and the memory phases do not wait on lgkmcnt at all
//   mode 11: as 5 with the DMA issued BEFORE the fragment reads; mode 13: as 5 but the COMPUTING wave issues the DMA (two per
//             four MFMAs in compute phase 1), the memory phases only read fragments
//   mode 14: DMA first, 4 in memory phase 1 and 4 in memory phase 2
//   mode 15: as 11, but every fourth K tile is the workgroup's own data (HBM): the GEMM's L2 miss ratio (~25 %)
// Prints shader clocks per K tile / 4 (= per phase) from s_memtime at the loop ends, and raw MFMA TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(const u4* __restrict__ in, size_t in_bytes, float* out, unsigned long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) u4 lds[];      // [2][4096] chunks of 16 B = 128 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = wave >= 4;
    if (MODE == 0 && late) return;
    for (int i = tid; i < 8192; i += 512) lds[i] = in[i];
    __syncthreads();
    u4 af[2][4], wf[2][2], whs[2];
    for (int p = 0; p < 2; ++p) { for (int i = 0; i < 4; ++i) af[p][i] = in[(tid * 16 + p * 4 + i) & 0xffff]; for (int j = 0; j < 2; ++j) wf[p][j] = in[(tid * 16 + 8 + p * 2 + j) & 0xffff]; }
    whs[0] = wf[0][0]; whs[1] = wf[0][1];
    f16v acc[2][4];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 4; ++i) for (int v = 0; v < 16; ++v) acc[j][i][v] = 0.f;
    const int r = lane & 31, kh = lane >> 5, fsw = (r >> 1) & 7, wm = wave >> 2, wn = wave & 3;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u4*>(in), 0, (int)(unsigned)in_bytes, 0x00020000);
    const int voff = lane * 16;
    auto mfma = [&](const u4& a, const u4& b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0); };
    auto compute = [&]() {
        if (MODE >= 3) {
            __builtin_amdgcn_s_setprio(1);
            const h2 sc = {(_Float16)0.5f, (_Float16)0.5f};
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) whs[j][e] = __builtin_bit_cast(unsigned, __builtin_bit_cast(h2, wf[0][j][e]) * sc);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = mfma(wf[1][j], af[0][i], acc[j][i]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = mfma(whs[j], af[1][i], acc[j][i]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = mfma(wf[0][j], af[0][i], acc[j][i]);
        if (MODE == 13) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        } else if (MODE >= 3) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
    };
    auto read_frags = [&](const u4* Ab, const u4* Wb, int ks) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int c = (p * 4 + ks * 2 + kh) ^ fsw;
#pragma unroll
            for (int i = 0; i < 4; ++i) af[p][i] = Ab[((wm * 4 + i) * 32 + r) * 8 + c];
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[p][j] = Wb[((wn * 2 + j) * 32 + r) * 8 + c];
        }
    };
    auto issue = [&](int kt, int buf, int i0 = 0, int i1 = 8) {
        u4* base = lds + buf * 4096 + wave * 64;
        const bool own = MODE == 6 || (MODE == 15 && (kt & 3) == 0);
        const unsigned so = (own ? (unsigned)((blockIdx.x * 64 + (kt & 63)) * 65536u) : (unsigned)((kt & 1023) * 65536u)) + wave * 1024;
#pragma unroll
        for (int i = i0; i < i1; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(base + 512 * i), 16, voff, (int)(so + i * 8192), 0, 0);
    };
    auto phase = [&](bool vm) {
        __builtin_amdgcn_sched_barrier(0);
        if (vm) __builtin_amdgcn_s_waitcnt(MODE == 10 ? 0x0F70 : 0x0070); else if (MODE != 10) __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    unsigned long long t0 = 0, t1 = 0;
    if (MODE == 0) {
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) { compute(); compute(); compute(); compute(); }
        t1 = __builtin_readcyclecounter();
    } else {
        if (MODE >= 5) { issue(0, 0); }
        phase(MODE >= 5);
        if (late) phase(false);
        t0 = __builtin_readcyclecounter();
        int cur = 0;
        for (int kt = 0; kt < iters; ++kt) {
            const u4* Ab = lds + cur * 4096;
            const u4* Wb = Ab + 2048;
            if (MODE >= 3) __builtin_amdgcn_s_setprio(0);
            if ((MODE == 11 || MODE == 15) && kt + 1 < iters) issue(kt + 1, cur ^ 1);
            if (MODE == 14 && kt + 1 < iters) issue(kt + 1, cur ^ 1, 0, 4);
            if (MODE >= 4 && MODE < 9 || MODE >= 11) read_frags(Ab, Wb, 0);
            if (MODE >= 5 && MODE < 11 && kt + 1 < iters) issue(kt + 1, cur ^ 1, 0, MODE == 7 ? 4 : 8);
            phase(false);
            if (MODE == 13 && kt + 1 < iters) issue(kt + 1, cur ^ 1);
            compute();
            phase(false);
            if (MODE >= 3) __builtin_amdgcn_s_setprio(0);
            if (MODE == 14 && kt + 1 < iters) issue(kt + 1, cur ^ 1, 4, 8);
            if (MODE >= 4 && MODE < 9 || MODE >= 11) read_frags(Ab, Wb, 1);
            if (MODE == 7 && kt + 1 < iters) issue(kt + 1, cur ^ 1, 4, 8);
            phase(MODE >= 5 && late);
            compute();
            phase(MODE >= 5 && !late);
            cur ^= 1;
        }
        t1 = __builtin_readcyclecounter();
        if (!late) phase(false);
    }
    float s = 0.f;
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 4; ++i) for (int v = 0; v < 16; ++v) s += acc[j][i][v];
    out[blockIdx.x * 512 + tid] = s;
    if (blockIdx.x == 0 && lane == 0) clk[wave] = t1 - t0;
}

template <int MODE>
static void run(const u4* in, size_t in_bytes, float* out, unsigned long long* clk, int iters, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto kfn = k<MODE>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL(kfn, dim3(256), dim3(512), 131072, 0, in, in_bytes, out, clk, 100);
    hipDeviceSynchronize();
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kfn, dim3(256), dim3(512), 131072, 0, in, in_bytes, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[8]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        const int waves = MODE == 0 ? 4 : 8;
        const double fl = 256.0 * waves * iters * 48 * 32768.0 * (MODE == 0 ? 2 : 1);
        printf("mode %d: %.0f clocks per phase (ideal 768); %.1f TFLOP/s raw MFMA = %.1f algorithmic, %.2f ms, shader clock %.2f GHz\n",
               MODE, (double)h[0] / iters / 4, fl / ms / 1e9, fl / ms / 1e9 / 3, ms, (double)h[0] / (ms * 1e6));
        fflush(stdout);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 10000;
    const int reps = argc > 2 ? atoi(argv[2]) : 4;
    const size_t in_bytes = (size_t)1 << 30;
    u4* in; float* out; unsigned long long* clk;
    hipMalloc(&in, in_bytes); hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&clk, 64);
    unsigned short* h = (unsigned short*)malloc(1 << 24);
    srand(1);
    for (int i = 0; i < (1 << 23); ++i) {
        unsigned e = 13 + rand() % 3;
        h[i] = (unsigned short)(((rand() & 1) << 15) | (e << 10) | (rand() & 0x3ff));
    }
    for (size_t o = 0; o < in_bytes; o += (1 << 24)) hipMemcpy((char*)in + o, h, 1 << 24, hipMemcpyHostToDevice);
    run<0>(in, in_bytes, out, clk, iters, reps);
    run<1>(in, in_bytes, out, clk, iters, reps);
    run<3>(in, in_bytes, out, clk, iters, reps);
    run<4>(in, in_bytes, out, clk, iters, reps);
    run<5>(in, in_bytes, out, clk, iters, reps);
    run<11>(in, in_bytes, out, clk, iters, reps);
    run<15>(in, in_bytes, out, clk, iters, reps);
    run<14>(in, in_bytes, out, clk, iters, reps);
    run<7>(in, in_bytes, out, clk, iters, reps);
    run<9>(in, in_bytes, out, clk, iters, reps);
    run<10>(in, in_bytes, out, clk, iters, reps);
    run<13>(in, in_bytes, out, clk, iters, reps);
    run<6>(in, in_bytes, out, clk, iters, reps);
    return 0;
}
