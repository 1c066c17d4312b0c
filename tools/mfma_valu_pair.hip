// Round 6 probe: can the two waves of a SIMD overlap an MFMA stream (wave w) with a VALU stream (wave w + 4)?
// One 512-thread workgroup per CU.  Waves 0-3 ("M") run 24 v_mfma_f32_32x32x16_f16 per step on six accumulators, waves 4-7 ("V") run a
// softmax-like VALU block per step (16 v_exp_f32 + ~100 fma / packed ops on independent chains).  Variants: M alone, V alone, both,
// both with a raw s_barrier per step, M's accumulators in ArchVGPRs or in AGPRs (inline asm), V's work interleaved into the SAME wave.
//     hipcc --offload-arch=gfx950 -O3 -o tools/mfma_valu_pair tools/mfma_valu_pair.hip && tools/mfma_valu_pair
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool AGPR>
__device__ __forceinline__ void mfma(f32x16& acc, h8 a, h8 b) {
    if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}

__device__ __forceinline__ void valu_block(float (&x)[16], float& l) {
    // ~ the softmax of one 32 x 32 score tile per lane: combine, max, exp2, sum, split into two fp16 planes
    float m = x[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = fmaxf(m, x[i]);
    m = fmaxf(m, __shfl_xor(m, 32));
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(x[i] - m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    l = l * 0.5f + s;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
        const fp16x2 hi = __builtin_amdgcn_cvt_pkrtz(x[i], x[i + 1]);
        const float l0 = fmaf((float)hi[0], -2048.f, x[i] * 2048.f), l1 = fmaf((float)hi[1], -2048.f, x[i + 1] * 2048.f);
        const fp16x2 lo = __builtin_amdgcn_cvt_pkrtz(l0, l1);
        x[i] = (float)lo[0] + (float)hi[1] * 1e-3f - 3.0f;         // keep a dependency into the next step, values stay bounded
        x[i + 1] = (float)lo[1] + (float)hi[0] * 1e-3f - 3.0f;
    }
}

// MODE bits: 1 = M waves run MFMAs, 2 = V waves run VALU, 4 = s_barrier per step, 8 = AGPR accumulators, 16 = same-wave interleave (waves 0-3 do both, 4-7 idle),
//            32 = M at s_setprio 1, 64 = both halves run the interleaved stream (with 16), 128 = sched_group_barrier pattern 1 MFMA : 6 VALU (with 16)
template <int MODE>
__global__ __launch_bounds__(512, 2) void pair_kernel(int iters, float* out) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_m = wave < 4;
    constexpr bool AG = (MODE & 8) != 0;
    f32x16 acc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[k][v] = 0.f;
    h8 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[k][e] = (_Float16)(0.001f * (threadIdx.x + e + k)); b[k][e] = (_Float16)(0.002f * (threadIdx.x - e + k)); }
    float x[16], l = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 0.01f * (threadIdx.x + i);
    if (MODE & 32) { if (is_m) __builtin_amdgcn_s_setprio(1); }
    for (int it = 0; it < iters; ++it) {
        const bool both = (MODE & 64) != 0;
        const bool do_m = (MODE & 1) && (is_m || both), do_v = (MODE & 2) && ((MODE & 16) ? (is_m || both) : !is_m);
        if (do_m) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 6; ++k) mfma<AG>(acc[k], a[(r + k) & 3], b[(r * 3 + k) & 3]);
        }
        if (do_v) valu_block(x, l);
        if (MODE & 128) {
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            }
        }
        if (MODE & 4) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    }
    float s = l;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int v = 0; v < 16; ++v) s += acc[k][v];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
static double run(int iters, float* out, int cus) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(pair_kernel<MODE>, dim3(cus), dim3(512), 0, 0, 10, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(pair_kernel<MODE>, dim3(cus), dim3(512), 0, 0, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best * 1e3 / iters;                  // us per step
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 4000;
    printf("%s, %d CUs; us per step (24 MFMAs = 768 matrix cycles per M wave; one softmax-like VALU block per V wave)\n", p.name, cus);
#define R(MODE, what) printf("  %-78s %.4f us\n", what, run<MODE>(iters, out, cus))
    R(1, "M alone (ArchVGPR accumulators)");
    R(9, "M alone (AGPR accumulators)");
    R(2, "V alone");
    R(3, "M + V on partner waves (ArchVGPR), free-running");
    R(11, "M + V on partner waves (AGPR), free-running");
    R(35, "M + V on partner waves (ArchVGPR), M at s_setprio 1");
    R(7, "M + V on partner waves (ArchVGPR), s_barrier per step");
    R(15, "M + V on partner waves (AGPR), s_barrier per step");
    R(19, "M + V interleaved in the SAME wave (ArchVGPR), partner idle");
    R(19 + 128, "M + V in the SAME wave, 1 MFMA : 6 VALU pattern, partner idle");
    R(19 + 64, "M + V in the SAME wave, BOTH waves of every SIMD run it (per step of each)");
    R(19 + 64 + 128, "M + V in the SAME wave, 1 : 6 pattern, BOTH waves of every SIMD run it");
    R(19 + 64 + 128 + 4, "... with an s_barrier per step");
    R(1 + 64, "M alone on BOTH waves of every SIMD (48 MFMAs per SIMD and step)");
    R(2 + 16 + 64, "V alone on BOTH waves of every SIMD");
    return 0;
}
