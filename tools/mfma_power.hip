// Upper bound probe: MFMA-only loop (no LDS / global traffic) with random fp16 operands,
// the same 3-MFMA-per-tile-pair pattern as the f16x3 GEMM.  Prints TFLOP/s (raw MFMA).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void k(const u4* __restrict__ in, float* out, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    u4 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 0xffff]; b[i] = in[(tid * 8 + 4 + i) & 0xffff]; }
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[i]), __builtin_bit_cast(h8, b[(i + 1) & 3]), acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[(i + 2) & 3]), __builtin_bit_cast(h8, b[i]), acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[i]), __builtin_bit_cast(h8, b[i]), acc[i], 0, 0, 0);
        }
        // perturb operands a little so the compiler keeps them live and data keeps toggling
        a[it & 3][0] ^= (unsigned)it * 2654435761u & 0x03ff03ffu;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
    out[tid] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    u4* in; float* out;
    const int nblk = 256 * 1, nthr = 512;
    hipMalloc(&in, 65536 * sizeof(u4)); hipMalloc(&out, nblk * nthr * sizeof(float));
    unsigned short* h = (unsigned short*)malloc(65536 * 16);
    srand(1);
    for (int i = 0; i < 65536 * 8; ++i) {               // random fp16 in (-2, 2): sign, exponent 13..15, random mantissa
        unsigned e = 13 + rand() % 3;
        h[i] = (unsigned short)(((rand() & 1) << 15) | (e << 10) | (rand() & 0x3ff));
    }
    hipMemcpy(in, h, 65536 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<8>, dim3(nblk), dim3(nthr), 0, 0, in, out, 100);
    hipDeviceSynchronize();
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<8>, dim3(nblk), dim3(nthr), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)nblk * 8 * iters * 12 * 32768.0;
        printf("mfma-only f16 32x32x16: %.1f TFLOP/s raw (%.2f ms)\n", fl / ms / 1e9, ms); fflush(stdout);
    }
    return 0;
}
