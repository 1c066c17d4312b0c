// Sequence-cluster sizes of an alignment (inverse sequence weights) on gfx950.
//
// Replaces the O(N^2 L) numba kernel `calc_num_cluster_members_nogaps_parallel`
// (proteingym/utils/weights.py:164-216, called from calc_weights_fast :13-53) and the per-sequence
// one-hot dot products of Tranception's `MSA_processing.compute_weight`
// (baselines/tranception/tranception/utils/msa_utils.py:341-352):
//   count[i] = #{ j : matches(i,j) / nongap(i) > identity_threshold },  j == i included,
//   matches(i,j) = #{ k : m[i,k] == m[j,k] and m[i,k] != invalid_value }.
//
// Integer/bit work on the vector ALU, no matrix cores: every symbol is a 5-bit code stored as five
// bit planes of 32 alignment columns each.  Gaps get code 30 in the "i" copy and 31 in the "j"
// copy, so a gap never equals anything; columns beyond L are 0 in both copies, so they never
// differ.  mismatches(i,j) over one 32-column word = popcount(OR_p (Pi[p] ^ Pj[p])): five v_xor,
// two v_or3, one accumulating v_bcnt = 8 VALU ops per 32 columns per pair, and
// matches = L - mismatches.  The floating-point test is hoisted out of the pair loop: for every i
// the host finds, with the same double division the reference evaluates, the smallest match count
// that passes, so the kernel compares integers (bit-exact with the reference predicate).
//
// mismatches(i,j) is symmetric (only the threshold depends on i), so only the upper triangle of tile
// pairs is computed and off-diagonal tiles credit both sides -- the trick of the reference's serial
// variant (weights.py:141-157) applied tile-wise.
// One workgroup = 128 "i" sequences x a slice of the "j" tiles; a lane owns 8 x 8 pairs in
// registers; planes stream global -> registers -> LDS one 32-column word per stage (double-buffered,
// one barrier per stage, pipelined across j-tile boundaries); 2 workgroups per CU.
#include "common.h"

#include <stdlib.h>
#include <vector>

namespace pgmi {

constexpr int kTile = 128;          // sequences per tile side
constexpr int kPlanes = 5;
constexpr uint32_t kGapI = 30, kGapJ = 31;

// codes int8 [N][L] -> planes [W][5][Npad] for both copies; nongap[i]; flags |= 1 on a symbol outside 0..29
__global__ void msa_encode_kernel(const int8_t* __restrict__ codes, int64_t N, int64_t L, int invalid, int64_t Npad, int W,
                                  uint32_t* __restrict__ Pi, uint32_t* __restrict__ Pj, int32_t* __restrict__ nongap,
                                  int32_t* __restrict__ flags) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int w = blockIdx.y;
    if (n >= Npad) return;
    uint32_t pi[kPlanes] = {0, 0, 0, 0, 0}, pj[kPlanes] = {0, 0, 0, 0, 0};
    int ng = 0;
    for (int b = 0; b < 32; ++b) {
        int64_t k = (int64_t)w * 32 + b;
        if (k >= L) break;
        uint32_t ci, cj;
        if (n < N) {
            int c = codes[n * L + k];
            if (c == invalid) { ci = kGapI; cj = kGapJ; }
            else {
                if (c < 0 || c > 29) { atomicOr(flags, 1); c = 0; }
                ci = cj = (uint32_t)c;
                ++ng;
            }
        } else { ci = kGapI; cj = kGapJ; }                      // padding sequences: all gaps
        for (int p = 0; p < kPlanes; ++p) {
            pi[p] |= ((ci >> p) & 1u) << b;
            pj[p] |= ((cj >> p) & 1u) << b;
        }
    }
    for (int p = 0; p < kPlanes; ++p) {
        Pi[((int64_t)w * kPlanes + p) * Npad + n] = pi[p];
        Pj[((int64_t)w * kPlanes + p) * Npad + n] = pj[p];
    }
    if (n < N && ng) atomicAdd(&nongap[n], ng);
}

__global__ __launch_bounds__(256, 2) void msa_count_kernel(const uint32_t* __restrict__ Pi, const uint32_t* __restrict__ Pj,
                                                            const int32_t* __restrict__ max_mism, int64_t Npad, int W,
                                                            int jtiles, int jtiles_per_block, int32_t* __restrict__ counts) {
    // one 32-column word of both tiles per stage, double-buffered: [buffer][copy][plane][sequence]
    __shared__ uint32_t lds[2][2][kPlanes][kTile];
    __shared__ int32_t s_cj[kTile];                        // j-side credits of the current off-diagonal tile
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const int64_t i0 = (int64_t)blockIdx.x * kTile;
    // mismatches(i,j) is symmetric, so only tiles with j-tile >= i-tile are visited; an off-diagonal
    // tile credits both its i rows (against max_mism[i]) and its j rows (against max_mism[j])
    const int jt0 = (int)blockIdx.x + blockIdx.y * jtiles_per_block;
    const int jt1 = min(jtiles, jt0 + jtiles_per_block);
    if (jt0 >= jt1) return;
    int mm[8], mmj[8], cnt[8];
    for (int r = 0; r < 8; ++r) { mm[r] = max_mism[i0 + ti * 8 + r]; cnt[r] = 0; mmj[r] = -1; }
    if (tid < kTile) s_cj[tid] = 0;
    // staging: 5 planes x 128 sequences x 2 copies = 1280 dwords per stage, 5 per thread
    constexpr int kPerThread = kPlanes * kTile * 2 / 256;
    uint32_t stg[kPerThread];
    auto fetch = [&](int jt, int w) {
        for (int q = 0; q < kPerThread; ++q) {
            int e = q * 256 + tid;                         // [copy][plane][seq]
            int seq = e & (kTile - 1), pl = (e >> 7) % kPlanes, cp = e / (kTile * kPlanes);
            const uint32_t* P = cp ? Pj : Pi;
            int64_t base = cp ? (int64_t)jt * kTile : i0;
            stg[q] = P[((int64_t)w * kPlanes + pl) * Npad + base + seq];
        }
    };
    auto stash = [&](int buf) {
        uint32_t* dst = &lds[buf][0][0][0];
        for (int q = 0; q < kPerThread; ++q) dst[q * 256 + tid] = stg[q];
    };
    const int steps = (jt1 - jt0) * W;                     // (j tile, word) pairs, pipelined across tile boundaries
    int acc[8][8];
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) acc[r][c] = 0;
    fetch(jt0, 0);
    stash(0);
    __syncthreads();
    int jt = jt0, w = 0;
    for (int st = 0; st < steps; ++st) {
        int nw = w + 1, njt = jt;
        if (nw == W) { nw = 0; ++njt; }
        const bool more = st + 1 < steps;
        if (more) fetch(njt, nw);                          // next stage's global loads fly during the compares
        {
            const int buf = st & 1;
            uint32_t a[kPlanes][8], b[kPlanes][8];
            for (int p = 0; p < kPlanes; ++p) {
                const uint4* pa = reinterpret_cast<const uint4*>(&lds[buf][0][p][ti * 8]);
                uint4 a0 = pa[0], a1 = pa[1];
                a[p][0] = a0.x; a[p][1] = a0.y; a[p][2] = a0.z; a[p][3] = a0.w;
                a[p][4] = a1.x; a[p][5] = a1.y; a[p][6] = a1.z; a[p][7] = a1.w;
                // lane column tj owns j = tj*4 .. +3 and 64 + tj*4 .. +3: 16-byte reads contiguous over the 16 lanes
                uint4 b0 = *reinterpret_cast<const uint4*>(&lds[buf][1][p][tj * 4]);
                uint4 b1 = *reinterpret_cast<const uint4*>(&lds[buf][1][p][64 + tj * 4]);
                b[p][0] = b0.x; b[p][1] = b0.y; b[p][2] = b0.z; b[p][3] = b0.w;
                b[p][4] = b1.x; b[p][5] = b1.y; b[p][6] = b1.z; b[p][7] = b1.w;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    uint32_t d = (a[0][r] ^ b[0][c]) | (a[1][r] ^ b[1][c]) | (a[2][r] ^ b[2][c]) |
                                 (a[3][r] ^ b[3][c]) | (a[4][r] ^ b[4][c]);
                    acc[r][c] += __popc(d);
                }
        }
        bool flush = false;
        if (w == W - 1) {                                  // tile finished: threshold and reset
            if (jt == (int)blockIdx.x) {                   // diagonal tile: every ordered pair is in it
                for (int r = 0; r < 8; ++r)
                    for (int c = 0; c < 8; ++c) { cnt[r] += (acc[r][c] <= mm[r]) ? 1 : 0; acc[r][c] = 0; }
            } else {
                int cj[8];
                for (int c = 0; c < 8; ++c) {
                    cj[c] = 0;
                    mmj[c] = max_mism[(int64_t)jt * kTile + (c >> 2) * 64 + tj * 4 + (c & 3)];
                }
                for (int r = 0; r < 8; ++r)
                    for (int c = 0; c < 8; ++c) {
                        cnt[r] += (acc[r][c] <= mm[r]) ? 1 : 0;
                        cj[c] += (acc[r][c] <= mmj[c]) ? 1 : 0;
                        acc[r][c] = 0;
                    }
                for (int c = 0; c < 8; ++c)
                    if (cj[c]) atomicAdd(&s_cj[(c >> 2) * 64 + tj * 4 + (c & 3)], cj[c]);
                flush = true;
            }
        }
        if (more) stash((st + 1) & 1);                     // the other buffer: last read before the previous barrier
        __syncthreads();
        if (flush) {                                       // uniform: w and jt are the same for every lane
            if (tid < kTile) {
                int v = s_cj[tid];
                if (v) { atomicAdd(&counts[(int64_t)jt * kTile + tid], v); s_cj[tid] = 0; }
            }
            // no extra barrier: the next writes to s_cj happen after at least one more __syncthreads
            // only when W > 1; with W == 1 every stage ends a tile, so order them explicitly
            if (W == 1) __syncthreads();
        }
        w = nw; jt = njt;
    }
    // the 16 lanes tj = 0..15 of one ti hold partial counts of the same 8 sequences
    for (int r = 0; r < 8; ++r) {
        int v = cnt[r];
        for (int off = 8; off >= 1; off >>= 1) v += __shfl_xor(v, off, 16);
        if (tj == 0 && v) atomicAdd(&counts[i0 + ti * 8 + r], v);
    }
}

}  // namespace pgmi

using namespace pgmi;

extern "C" int pgmi_msa_cluster_counts(int device, const int8_t* matrix, int64_t N, int64_t L, int invalid_value,
                                       double identity_threshold, int32_t* counts_out, double* kernel_ms) {
    if (!matrix || !counts_out || N <= 0 || L <= 0) { set_error("pgmi_msa_cluster_counts: null pointer or empty alignment"); return PGMI_EINVAL; }
    if (!(identity_threshold >= 0.0 && identity_threshold < 1.0)) { set_error("identity_threshold must be in [0, 1), got %g", identity_threshold); return PGMI_EINVAL; }
    if (L > (1 << 24) || N > (int64_t)1 << 30) { set_error("alignment too large (N=%lld, L=%lld)", (long long)N, (long long)L); return PGMI_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("no HIP device"); return PGMI_ENODEV; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d visible)", device, ndev); return PGMI_EINVAL; }
    PGMI_HIP(hipSetDevice(device));
    const int W = (int)((L + 31) / 32);
    const int64_t Npad = (N + kTile - 1) / kTile * kTile;
    const int jtiles = (int)(Npad / kTile);
    int8_t* d_codes = nullptr; uint32_t *d_pi = nullptr, *d_pj = nullptr; int32_t *d_ng = nullptr, *d_mm = nullptr, *d_cnt = nullptr, *d_flag = nullptr;
    hipStream_t s = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = PGMI_OK;
    std::vector<int32_t> ng((size_t)N), mm((size_t)Npad, -1), thr_of((size_t)L + 1);
    auto cleanup = [&]() {
        hipFree(d_codes); hipFree(d_pi); hipFree(d_pj); hipFree(d_ng); hipFree(d_mm); hipFree(d_cnt); hipFree(d_flag);
        if (e0) hipEventDestroy(e0);
        if (e1) hipEventDestroy(e1);
        if (s) hipStreamDestroy(s);
    };
#define MSA_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(_e)); cleanup(); return PGMI_EHIP; } } while (0)
    const size_t plane_bytes = (size_t)W * kPlanes * Npad * sizeof(uint32_t);
    MSA_HIP(hipStreamCreate(&s));
    MSA_HIP(hipEventCreate(&e0));
    MSA_HIP(hipEventCreate(&e1));
    MSA_HIP(hipMalloc(&d_codes, (size_t)N * L));
    MSA_HIP(hipMalloc(&d_pi, plane_bytes));
    MSA_HIP(hipMalloc(&d_pj, plane_bytes));
    MSA_HIP(hipMalloc(&d_ng, Npad * sizeof(int32_t)));
    MSA_HIP(hipMalloc(&d_mm, Npad * sizeof(int32_t)));
    MSA_HIP(hipMalloc(&d_cnt, Npad * sizeof(int32_t)));
    MSA_HIP(hipMalloc(&d_flag, sizeof(int32_t)));
    MSA_HIP(hipMemcpyAsync(d_codes, matrix, (size_t)N * L, hipMemcpyHostToDevice, s));
    MSA_HIP(hipMemsetAsync(d_ng, 0, Npad * sizeof(int32_t), s));
    MSA_HIP(hipMemsetAsync(d_cnt, 0, Npad * sizeof(int32_t), s));
    MSA_HIP(hipMemsetAsync(d_flag, 0, sizeof(int32_t), s));
    {
        dim3 grid((unsigned)((Npad + 255) / 256), (unsigned)W);
        hipLaunchKernelGGL(msa_encode_kernel, grid, dim3(256), 0, s, d_codes, N, L, invalid_value, Npad, W, d_pi, d_pj, d_ng, d_flag);
    }
    int32_t flag = 0;
    MSA_HIP(hipMemcpyAsync(ng.data(), d_ng, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MSA_HIP(hipMemcpyAsync(&flag, d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MSA_HIP(hipStreamSynchronize(s));
    if (flag) { set_error("alignment holds a symbol outside 0..29 that is not invalid_value (%d)", invalid_value); cleanup(); return PGMI_EINVAL; }
    // smallest match count m with (double)m / (double)nongap > threshold, per non-gap length (the
    // reference's own predicate, weights.py:207 / msa_utils.py:345); max mismatches = L - m
    for (int64_t g = 1; g <= L; ++g) {
        int64_t m = (int64_t)(identity_threshold * (double)g);
        if (m < 0) m = 0;
        while (m > 0 && (double)(m - 1) / (double)g > identity_threshold) --m;
        while (m <= g && !((double)m / (double)g > identity_threshold)) ++m;
        thr_of[(size_t)g] = (m <= g) ? (int32_t)(L - m) : -1;
    }
    for (int64_t i = 0; i < N; ++i) mm[(size_t)i] = ng[(size_t)i] > 0 ? thr_of[(size_t)ng[(size_t)i]] : -1;
    MSA_HIP(hipMemcpyAsync(d_mm, mm.data(), (size_t)Npad * sizeof(int32_t), hipMemcpyHostToDevice, s));
    {
        // triangular schedule: block (I, y) visits j tiles I + y*per .. ; slices of <= 32 tiles keep the
        // uneven rows balanced, and small alignments still get >= ~2048 workgroups where possible
        int per = 32;
        while (per > 1 && (int64_t)jtiles * ((jtiles + per - 1) / per) < 4096) per >>= 1;
        int gy = (jtiles + per - 1) / per;
        MSA_HIP(hipEventRecord(e0, s));
        hipLaunchKernelGGL(msa_count_kernel, dim3((unsigned)jtiles, (unsigned)gy), dim3(256), 0, s, d_pi, d_pj, d_mm, Npad, W, jtiles, per, d_cnt);
        MSA_HIP(hipEventRecord(e1, s));
    }
    MSA_HIP(hipMemcpyAsync(counts_out, d_cnt, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MSA_HIP(hipStreamSynchronize(s));
    MSA_HIP(hipGetLastError());
    if (kernel_ms) { float ms = 0; hipEventElapsedTime(&ms, e0, e1); *kernel_ms = ms; }
#undef MSA_HIP
    cleanup();
    return rc;
}
