// How long a workgroup's burst of output stores takes to be acknowledged (s_waitcnt vmcnt(0)) when 1, 8, 32 (one XCD's worth), 64,
// 128 or 256 workgroups burst at the same moment -- the question behind the GEMM epilogues' ~8 us per item (gemm16x_kernel.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/store_drain tools/store_drain.hip && tools/store_drain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

// every workgroup of 512 threads writes (and optionally first reads) `kb` KB of its own region, 16 B per lane per instruction, rows of
// 256 B per 16 lanes like the GEMM's epilogue; only workgroups with (blockIdx % stride) == 0 take part, so that `stride` = 8 keeps one XCD
__global__ __launch_bounds__(512) void burst(float* out, const float* in, int kb, int stride, int do_read, long long* ticks, int rounds, int fresh) {
    if (blockIdx.x % stride) return;
    const int n = kb * 1024 / 16 / 512;                                 // 16-byte pieces per thread
    long long t_acc = 0;
    for (int r = 0; r < rounds; ++r) {
        // a fresh region every round (fresh = 1: nothing of it in L2 / the Infinity Cache, like a GEMM's output tile), or the same one
        const size_t base = ((size_t)(fresh ? r : 0) * gridDim.x + blockIdx.x) * (size_t)kb * 256;          // floats
        __syncthreads();
        const long long t0 = wall_clock64();
        f32x4 v = {1.f, 2.f, 3.f, (float)r};
        for (int i = 0; i < n; ++i) {
            const size_t o = base + ((size_t)i * 512 + threadIdx.x) * 4;
            if (do_read) { const f32x4 x = *reinterpret_cast<const f32x4*>(in + o); v += x; }
            *reinterpret_cast<f32x4*>(out + o) = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        t_acc += wall_clock64() - t0;
        // idle gap so that the next burst starts from a drained memory system, like a K loop between epilogues
        const long long t1 = wall_clock64();
        while (wall_clock64() - t1 < 2000) __builtin_amdgcn_s_sleep(32);   // 20 us
    }
    if (threadIdx.x == 0) ticks[blockIdx.x] = t_acc;
}

int main() {
    const int G = 256, kb = 256, rounds = 20;
    float *out, *in;
    long long* ticks;
    CK(hipMalloc(&out, (size_t)rounds * G * kb * 1024));
    CK(hipMalloc(&in, (size_t)rounds * G * kb * 1024));
    CK(hipMemset(in, 0, (size_t)rounds * G * kb * 1024));
    CK(hipMalloc(&ticks, G * sizeof(long long)));
    std::vector<long long> h(G);
    for (int fresh = 1; fresh >= 0; --fresh)
    for (int do_read = 0; do_read < 2; ++do_read)
        for (int stride : {256, 32, 8, 4, 2, 1}) {
            CK(hipMemset(ticks, 0, G * sizeof(long long)));
            hipLaunchKernelGGL(burst, dim3(G), dim3(512), 0, 0, out, in, kb, stride, do_read, ticks, rounds, fresh);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), ticks, G * sizeof(long long), hipMemcpyDeviceToHost));
            double sum = 0, mx = 0; int n = 0;
            for (int b = 0; b < G; b += stride) { const double us = h[b] / (double)rounds / 100.0; sum += us; mx = std::max(mx, us); ++n; }
            printf("%s %s %3d workgroups bursting (%s): mean %.2f us, max %.2f us per 256 KB burst  -> %.2f TB/s aggregate\n",
                   fresh ? "fresh region " : "same region  ", do_read ? "read+write" : "write     ", n, stride == 8 ? "ONE XCD" : stride > 8 ? "spread over XCDs" : "all XCDs",
                   sum / n, mx, n * (do_read ? 2.0 : 1.0) * kb * 1024 / (sum / n * 1e-6) / 1e12);
        }
    return 0;
}
