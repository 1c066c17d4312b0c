// Phase timing of libpgmi's persistent ping-pong GEMM (proteingym_amd/csrc/gemm16x_kernel.h): the tuning-only instantiations
// gemm16x_kernel<EPI_NONE, 0, STG, DFLAGS >= 0> stamp the shader clock at every phase barrier (waves 0 and 4 of workgroup 0) and
// can ablate parts of the loop.  Built on its own so that the product library carries the production instantiations only:
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/gemm_diag tools/gemm_diag.hip
//   tools/gemm_diag [flags] [M N K]        flags: 0 as shipped (register staging), 11 no loads / LDS writes / fragment reads;
//                                          DMA form: 1000 as shipped, 1016 no DMA after the first K tile, 1008 no fragment reads,
//                                          1024 neither, 1128 no wait for the DMA  (ablations give wrong numbers: timing only)
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "../proteingym_amd/csrc/gemm16x_kernel.h"

namespace pgmi {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
}
using namespace pgmi;

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

// random split-fp16 operand [rows][K] in the K-interleaved layout (common.h ki_off)
__global__ void fill_kernel(unsigned short* p, size_t rows, int K, unsigned int seed, float scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * (size_t)K) return;
    unsigned int x = (unsigned int)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    const float v = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
    _Float16 hi, lo;
    split_act(v, hi, lo);
    const size_t o = ki_off(i / K, (int)(i % K), K);
    p[o] = __builtin_bit_cast(unsigned short, hi);
    p[o + 32] = __builtin_bit_cast(unsigned short, lo);
}

template <int F>
static void launch(dim3 grid, size_t lds_bytes, const unsigned short* A, const unsigned short* W, float* C, int M, int N, int K, TilePlan tp) {
    auto kfn = gemm16x_kernel<EPI_NONE, 0, (F >= 1000 ? 1 : 0), F>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    QkvOut qo{};
    hipLaunchKernelGGL(kfn, grid, dim3(XNT), lds_bytes, nullptr, A, W, (const float*)nullptr, (const float*)nullptr, C, (unsigned short*)nullptr,
                       (size_t)0, M, N, K, 1.0f, tp, qo);
}

int main(int argc, char** argv) {
    const int flags = argc > 1 ? atoi(argv[1]) : 1000;
    const int M = argc > 4 ? atoi(argv[2]) : 82368, N = argc > 4 ? atoi(argv[3]) : 1280, K = argc > 4 ? atoi(argv[4]) : 5120;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int G = prop.multiProcessorCount;
    G -= G % 8;
    unsigned short *A, *W;
    float* C;
    unsigned long long* dbuf;
    CK(hipMalloc(&A, (size_t)M * K * 4));
    CK(hipMalloc(&W, (size_t)N * K * 4));
    CK(hipMalloc(&C, (size_t)M * N * 4));
    const size_t nd = (size_t)2 * kDiagSamples * 2;
    CK(hipMalloc(&dbuf, nd * 8));
    CK(hipMemset(dbuf, 0, nd * 8));
    fill_kernel<<<(unsigned)(((size_t)M * K + 255) / 256), 256>>>(A, M, K, 1u, 1.0f);
    fill_kernel<<<(unsigned)(((size_t)N * K + 255) / 256), 256>>>(W, N, K, 2u, 0.03f);
    TilePlan tp{};
    tp.tiles_m = (M + XBM - 1) / XBM;
    tp.tiles_n = (N + XBN - 1) / XBN;
    tp.n_main = tp.n_items = tp.tiles_m * tp.tiles_n;
    tp.split = 1;
    tp.group_m = 4;
    tp.diag = dbuf;
    tp.diag_flags = flags;
    const size_t lds_bytes = (size_t)2 * X_STAGE * 16;
    const dim3 grid(std::min(G, tp.n_items));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    switch (flags) {
        case 0: launch<0>(grid, lds_bytes, A, W, C, M, N, K, tp); break;
        case 11: launch<11>(grid, lds_bytes, A, W, C, M, N, K, tp); break;
        case 1000: launch<1000>(grid, lds_bytes, A, W, C, M, N, K, tp); break;
        case 1016: launch<1016>(grid, lds_bytes, A, W, C, M, N, K, tp); break;
        case 1008: launch<1008>(grid, lds_bytes, A, W, C, M, N, K, tp); break;
        case 1024: launch<1024>(grid, lds_bytes, A, W, C, M, N, K, tp); break;
        case 1128: launch<1128>(grid, lds_bytes, A, W, C, M, N, K, tp); break;
        default: fprintf(stderr, "flags %d not instantiated\n", flags); return 1;
    }
    hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("[gemm16x diag] flags %d  M %d N %d K %d: %.3f ms = %.1f TFLOP/s algorithmic (under instrumentation)\n", flags, M, N, K, ms,
           2.0 * M * N * K / ms / 1e9);
    std::vector<unsigned long long> h(nd);
    CK(hipMemcpy(h.data(), dbuf, nd * 8, hipMemcpyDeviceToHost));
    // per wave: work = release(k-1) -> arrival(k), wait = arrival(k) -> release(k), release(k) = the later arrival of the two
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
        printf("[gemm16x diag] %s waves: ", g ? "late " : "early");
        static const char* nm[5] = {"mem1", "cmp1", "other", "mem2", "cmp2"};
        for (int p : {0, 1, 3, 4})
            printf("%s work %.0f wait %.0f | ", nm[p], cnt[p] ? work[p] / cnt[p] : 0.0, cnt[p] ? wait[p] / cnt[p] : 0.0);
        printf("(shader clocks, mean over %d K tiles)\n", cnt[0]);
    }
    return 0;
}
