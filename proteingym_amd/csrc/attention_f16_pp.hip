// The barrier-locked two-role 8-wave form of the split-fp16 attention (see attention_f16.hip for the math, the operand planes and the
// 4-wave kernel whose arithmetic this one repeats row by row).  A file of its own: its compiler flags are its own (build_native.py).
#include <stdlib.h>
#include <string.h>
#include <algorithm>

#include "attention_f16_common.h"

namespace pgmi {

// ---- Round 6: the barrier-locked two-role ("ping-pong") form of the dense head_dim-64 kernel ------------------------------------
// The 4-wave kernel above runs two independent workgroups per CU: the two waves of a SIMD each walk their own chain (12 S MFMAs ->
// max / exp2 / split (~660 VALU-issue cycles) -> 12 P V MFMAs (~768 matrix cycles per tile)) and nothing keeps them out of phase, so
// a SIMD completes a wave-tile every ~1 500 cycles = the SUM of the two (profiles/r4, r5).  Here ONE 8-wave workgroup owns the CU and
// the two waves of every SIMD alternate roles between raw s_barriers, as gemm16x_kernel.h does for the GEMM and
// MI355X_MICROARCH.md ("Two waves per SIMD") describes for an 8-wave attention loop:
//     segment X (matrix, s_setprio 1):  P V of key tile j  +  S = K Q^T of key tile j + 1      -- 24 MFMAs from registers
//     segment Y (vector / memory):      the DMA share of a later tile, the V^T fragments of tile j + 1, the online softmax of tile
//                                       j + 1 (scores -> P hi | lo), the K fragments of tile j + 2
// waves 0-3 run X while waves 4-7 run Y and vice versa, one segment apart, two barriers per key tile.
// Work list: the launch's query tiles in (sequence, head, tile) order; a workgroup takes EIGHT CONSECUTIVE ones, which for
// sequences of at least seven tiles span at most two (sequence, head) pairs -- T = 288's nine tiles are 8 + (1 with seven of the next
// head) instead of three quarter-idle blocks.  The ring therefore holds up to two K / V^T streams per stage (2 x 16 KB x 4 stages);
// a wave reads the stream of its own pair (a scalar select of the LDS base).
// A row goes through exactly the arithmetic of attention_f16x3_v2_kernel, tile by tile in the same order: the same bits
// (tests/test_gpu_ops.py::test_attention_pp_bits_equal_v2), so which of the two kernels serves a shape is a launch option.
constexpr int PP_WAVES = 8, PP_NSTG = 4, PP_STREAM_CH = A_STAGE, PP_STAGE_CH = 2 * A_STAGE;     // chunks of 16 B

// DBG (timing probes only, wrong results; scripts/att_bench.py --ab att_pp=1,att_pp=17,...: option value = 1 + 16 DBG):
//   1 = no DMA waits in the loop, 2 = no softmax arithmetic, 4 = no MFMAs, 8 = no fragment reads, 16 = no DMA issue in the loop
//   32 = no s_setprio at all, 64 = the Y segment (not X) at priority 1, 128 = MFMA accumulators in AGPRs (correct results)
template <int OUT, int DBG = 0>
__global__ __launch_bounds__(PP_WAVES * 64, 2) void attention_f16x3_pp_kernel(
    const unsigned short* __restrict__ qk16, size_t qk_plane, const unsigned short* __restrict__ vt16, size_t vt_plane,
    const int32_t* __restrict__ kv_len, int T, int H, int Tp, float* __restrict__ ctx, unsigned short* __restrict__ ctx16,
    int n32, int total_tiles, int n_wg) {
    if (DBG & 128) asm volatile("; an AGPR operand: hipcc then selects the AGPR form of every MFMA of this kernel" ::"a"(0.0f));
    constexpr float defer_thr = kAttDefer;
    constexpr int DH = 64, KCPR = 8, KCH = AKT * KCPR, VCH = DH * 4, NS = 4, ND = 2;
    constexpr int NDMA = 2;                                        // wave-instructions (1 KiB) per wave, stream and tile: 16 / 8
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];   // [PP_NSTG][2 streams][A_STAGE]
    // XCD-chunked order: workgroup i runs on XCD i % 8; an XCD walks a contiguous run of the list, so the workgroups that share a
    // pair's K / V^T stream (two at T = 288, four or five at T = 1024) are neighbours in time on ONE L2
    int wg;
    {
        const int i = (int)blockIdx.x, xcd = i & 7, q = n_wg >> 3, r8 = n_wg & 7;
        wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (i >> 3);
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = wave >= 4;                                  // the second wave of every SIMD: one segment behind
    const int r = lane & 31, kh = lane >> 5;
    const int D = H * DH;
    const int f0 = wg * PP_WAVES;
    const int pair0 = f0 / n32, pair_last = min(f0 + PP_WAVES - 1, total_tiles - 1) / n32;
    const int ns = pair_last - pair0 + 1;                         // 1 or 2 streams (the launcher guarantees n32 >= 7)
    const int f = f0 + wave;
    const bool active = f < total_tiles;
    const int pair = min(f, total_tiles - 1) / n32;
    const int strm = __builtin_amdgcn_readfirstlane(pair - pair0);
    const int b = pair / H, h = pair - b * H;
    const int q0 = (min(f, total_tiles - 1) - pair * n32) * 32;   // first query of the wave's tile
    // per stream: sequence, head, key count, tiles
    int sb[2], sh[2], sTk[2], snkt[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ps = min(pair0 + s, pair_last);
        sb[s] = ps / H;
        sh[s] = ps - sb[s] * H;
        sTk[s] = kv_len ? kv_len[sb[s]] : T;
        snkt[s] = (sTk[s] + AKT - 1) / AKT;
    }
    const int nkt = __builtin_amdgcn_readfirstlane(max(snkt[0], snkt[1]));          // the workgroup's loop count
    const int Tk = strm ? sTk[1] : sTk[0];
    const int nkt_w = __builtin_amdgcn_readfirstlane(active ? (strm ? snkt[1] : snkt[0]) : 0);   // key tiles this wave computes on

    const size_t seq_halfs = (size_t)T * (2 * D), vt_halfs = (size_t)DH * Tp;
    const unsigned long long qk_bytes = std::min<unsigned long long>(0xFFFFFFFFull, (unsigned long long)qk_plane * 2ull + seq_halfs * 2ull);
    const unsigned long long vt_bytes = std::min<unsigned long long>(0xFFFFFFFFull, (unsigned long long)vt_plane * 2ull + vt_halfs * 2ull);
    const __amdgpu_buffer_rsrc_t rsQK0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(qk16) + (size_t)sb[0] * seq_halfs, 0, (int)(unsigned int)qk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsQK1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(qk16) + (size_t)sb[1] * seq_halfs, 0, (int)(unsigned int)qk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsVT0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(vt16) + ((size_t)sb[0] * H + sh[0]) * vt_halfs, 0, (int)(unsigned int)vt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsVT1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(vt16) + ((size_t)sb[1] * H + sh[1]) * vt_halfs, 0, (int)(unsigned int)vt_bytes, 0x00020000);

    // DMA map of one stream's stage: the 4-wave kernel's (its comment above `voff`), 16 wave-instructions over 8 waves
    int voff[NDMA], slot0[NDMA], sstep[NDMA], pbase[NDMA], krow[NDMA];
    bool is_k[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int wi = wave + PP_WAVES * i;
        slot0[i] = wi * 64;
        is_k[i] = wi < 2 * KCH / 64;
        if (is_k[i]) {
            const int p = wi / (KCH / 64), key = ((wi % (KCH / 64)) * 64 + lane) / KCPR;
            const int c = (lane % KCPR) ^ ((key >> 1) & 7);
            krow[i] = key;
            voff[i] = c * 16;                                       // + row * row bytes, per tile (rows clamped to the sequence)
            pbase[i] = (int)((unsigned int)p * (unsigned int)qk_plane * 2u + (unsigned int)D * 2u);
            sstep[i] = 0;
        } else {
            const int wv = wi - 2 * KCH / 64, p = wv / (VCH / 64), g = (wv % (VCH / 64)) * 64 + lane;
            const int d = g >> 2, c = (g & 3) ^ ((d >> 2) & 3);
            krow[i] = 0;
            voff[i] = d * Tp * 2 + c * 16;
            pbase[i] = (int)((unsigned int)p * (unsigned int)vt_plane * 2u);
            sstep[i] = (AKT / 8) * 16;
        }
    }
    auto issue_stream = [&](int kt_s, int stage, int s) {           // tile kt_s of stream s (already clamped to the stream's tiles)
        u32x4* base = lds + stage * PP_STAGE_CH + s * PP_STREAM_CH;
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            if (is_k[i]) {
                // K rows past the sequence's last token are clamped to it (finite, masked by Tk): never into the next sequence
                const int vo = min(kt_s * AKT + krow[i], T - 1) * (2 * D) * 2 + voff[i];
                const int so = pbase[i] + (s ? sh[1] : sh[0]) * DH * 2;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(s ? rsQK1 : rsQK0, (__attribute__((address_space(3))) void*)(base + slot0[i]), 16, vo, so, 0, 0);
            } else {
                const int so = pbase[i] + kt_s * sstep[i];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(s ? rsVT1 : rsVT0, (__attribute__((address_space(3))) void*)(base + slot0[i]), 16, voff[i], so, 0, 0);
            }
        }
    };
    // every wave issues NDMA instructions per stream: the per-wave count per tile is 2 ns (wave-uniform, the same for all waves)
    auto issue_tile = [&](int kt) {
        const int stage = kt & (PP_NSTG - 1);
        issue_stream(min(kt, snkt[0] - 1), stage, 0);
        if (ns > 1) issue_stream(min(kt, snkt[1] - 1), stage, 1);
    };
    // this wave's DMA share of a tile has landed once at most `younger` tiles issued after it remain in flight
    auto wait_tile = [&](int younger) {
        if (DBG & 1) return;
        if (ns > 1) {
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * NDMA / 2) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NDMA / 2) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    auto phase = [&]() {                                            // segment boundary: nothing moves across it; LDS traffic of this wave done
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                         // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: tile 0 into stage 0, the eight Q tiles through stages 1-2 (8 waves x 8 KB), then tiles 1 .. 3 ----
    issue_tile(0);
    u32x4 qh[NS], ql[NS];
    {
        u32x4* qbase = lds + PP_STAGE_CH + wave * (2 * KCH);
        const __amdgpu_buffer_rsrc_t rsQ = strm ? rsQK1 : rsQK0;
        constexpr int NQ = 2 * KCH / 64;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int fq = i * 64 + lane, pq = fq / KCH, row = (fq % KCH) / KCPR;
            const int c = (fq % KCPR) ^ ((row >> 1) & 7);
            const int vo = (int)(((unsigned int)min(q0 + row, T - 1) * (unsigned int)(2 * D) + (unsigned int)(h * DH)) * 2u + (unsigned int)c * 16u);
            const int so = (int)((unsigned int)pq * (unsigned int)qk_plane * 2u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (__attribute__((address_space(3))) void*)(qbase + i * 64), 16, vo, so, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ci = r * KCPR + ((2 * s + kh) ^ ((r >> 1) & 7));
            qh[s] = qbase[ci];
            ql[s] = qbase[KCH + ci];
        }
    }
    phase();                                                        // every wave holds its Q; tile 0 is visible; stages 1-2 are free
#pragma unroll
    for (int t = 1; t < PP_NSTG; ++t)
        if (t < nkt) issue_tile(t);
    const int issued0 = min(PP_NSTG - 1, nkt - 1);                  // newest tile in flight after the prologue

    const u32x4* const my = lds + strm * PP_STREAM_CH;              // + stage * PP_STAGE_CH: the wave's own stream
    u32x4 kf[NS][2], vf[2][ND][2];                                  // fragments of the next S tile / the next P V tile (hi, lo)
    u32x4 ph[2], pl[2];
    f32x16 om[ND], oc[ND], sm, sc;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int v = 0; v < 16; ++v) { om[dt][v] = 0.f; oc[dt][v] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    constexpr float kInvLo = 1.0f / kLoScale;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    auto read_k = [&](int kt) {
        if (DBG & 8) return;
        const u32x4* Kb = my + (kt & (PP_NSTG - 1)) * PP_STAGE_CH;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ci = r * KCPR + ((2 * s + kh) ^ ((r >> 1) & 7));
            kf[s][0] = Kb[ci];
            kf[s][1] = Kb[KCH + ci];
        }
    };
    auto read_v = [&](int kt) {
        if (DBG & 8) return;
        const u32x4* Vb = my + (kt & (PP_NSTG - 1)) * PP_STAGE_CH + 2 * KCH;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                const int d = dt * 32 + r;
                const int ci = d * 4 + ((2 * m + kh) ^ ((d >> 2) & 3));
                vf[m][dt][0] = Vb[ci];
                vf[m][dt][1] = Vb[VCH + ci];
            }
    };
    // segment X of step j: K fragments of tile j + 1 (their latency hides under the P V MFMAs), P V of tile j, S of tile j + 1
    // (each in the 4-wave kernel's MFMA order)
    auto seg_x = [&](int j) {
        if (!(DBG & 96)) __builtin_amdgcn_s_setprio(1);
        if (j + 1 < nkt_w) read_k(j + 1);
        if (!(DBG & 4) && j >= 0 && j < nkt_w) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    oc[dt] = mfma_h(vf[m][dt][0], pl[m], oc[dt]);
                    oc[dt] = mfma_h(vf[m][dt][1], ph[m], oc[dt]);
                    om[dt] = mfma_h(vf[m][dt][0], ph[m], om[dt]);
                }
        }
        if (!(DBG & 4) && j + 1 < nkt_w) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                sc = mfma_h(kf[s][0], ql[s], s == 0 ? zero16 : sc);
                sc = mfma_h(kf[s][1], qh[s], sc);
                sm = mfma_h(kf[s][0], qh[s], s == 0 ? zero16 : sm);
            }
        }
        if (!(DBG & 96)) __builtin_amdgcn_s_setprio(0);
    };
    // the online softmax of key tile kt on the scores in sm / sc -> P (hi | lo fragments); the 4-wave kernel's code, value for value
    auto softmax = [&](int kt) {
        float st[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) st[v] = fmaf(sc[v], kInvLo, sm[v]);
        if (kt * AKT + AKT > Tk) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int key = kt * AKT + (v & 3) + 8 * (v >> 2) + 4 * kh;
                if (key >= Tk) st[v] = -INFINITY;
            }
        }
        float mloc = st[0];
#pragma unroll
        for (int v = 1; v < 16; ++v) mloc = fmaxf(mloc, st[v]);
        {
            const unsigned int mu = __builtin_bit_cast(unsigned int, mloc);
            const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
            const unsigned int s0 = sw[0], s1 = sw[1];
            mloc = fmaxf(__builtin_bit_cast(float, s0), __builtin_bit_cast(float, s1));
        }
        const float m_new = fmaxf(m_run, mloc);
        if (!__all(m_new <= m_run + defer_thr)) {                   // deferred, per-row rescale (see the 4-wave kernel)
            const bool moved = m_new > m_run + defer_thr;
            const float alpha = moved ? __builtin_amdgcn_exp2f(m_run - m_new) : 1.0f;
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int v = 0; v < 16; ++v) { om[dt][v] *= alpha; oc[dt][v] *= alpha; }
            if (moved) m_run = m_new;
        }
        const float mb = m_run - 10.0f;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
            const f32x2 dlt = f32x2{st[v], st[v + 1]} - f32x2{mb, mb};
            st[v] = __builtin_amdgcn_exp2f(dlt[0]);
            st[v + 1] = __builtin_amdgcn_exp2f(dlt[1]);
        }
        l_run += ((st[0] + st[1]) + (st[2] + st[3])) + ((st[4] + st[5]) + (st[6] + st[7])) +
                 (((st[8] + st[9]) + (st[10] + st[11])) + ((st[12] + st[13]) + (st[14] + st[15])));
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p0 = st[8 * m + 2 * e], p1 = st[8 * m + 2 * e + 1];
                typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
                const fp16x2 hi2 = __builtin_amdgcn_cvt_pkrtz(p0, p1);
                const f32x2 ps = f32x2{p0, p1} * f32x2{kLoScale, kLoScale};
                const float l0 = fmaf((float)hi2[0], -kLoScale, ps[0]), l1 = fmaf((float)hi2[1], -kLoScale, ps[1]);
                const fp16x2 lo2 = __builtin_amdgcn_cvt_pkrtz(l0, l1);
                ph[m][e] = __builtin_bit_cast(unsigned int, hi2);
                pl[m][e] = __builtin_bit_cast(unsigned int, lo2);
            }
    };
    // segment Y of step j: this wave's DMA share of tile j + 4 (into the stage of tile j: last read -- V^T -- in the other half's
    // Y(j - 1), one barrier ago at the latest), the V^T fragments and the softmax of tile j + 1
    auto seg_y = [&](int j) {
        if (DBG & 64) __builtin_amdgcn_s_setprio(1);
        if (!(DBG & 16) && j + 4 >= PP_NSTG && j + 4 < nkt) issue_tile(j + 4);
        if (j + 1 < nkt_w) {
            read_v(j + 1);
            if (!(DBG & 2)) softmax(j + 1);
        }
        if (DBG & 64) __builtin_amdgcn_s_setprio(0);
    };
    // tile n of this wave's share must have landed (`newest` = the newest tile this wave has issued)
    auto land = [&](int n, int newest) {
        if (n < nkt) wait_tile(min(newest, nkt - 1) - n);
    };

    // Schedule.  Every wave runs  X(j) | Y(j)  per key tile, a barrier after each; the late waves run ONE SEGMENT BEHIND (one extra
    // barrier before the loop): while waves 0-3 are in X the partners on their SIMDs are in Y and vice versa.  Tile n sits in stage
    // n % 4.  Its K rows are first read by the early waves at the head of X(n - 1), so every wave's share must have landed at the
    // barrier before that: the end of the early waves' Y(n - 2) = the end of the late waves' X(n - 2).
    if (late) {
        land(1, issued0);
        phase();
    }
    for (int j = -1; j < nkt - 1; ++j) {
        seg_x(j);
        if (late) land(j + 2, max(j + 3, PP_NSTG - 1));         // the barrier before the early waves' X(j + 1), which reads K(j + 2)
        phase();
        seg_y(j);
        if (!late) land(j + 2, max(j + 4, PP_NSTG - 1));        // the same barrier, seen from the early waves
        phase();
    }
    seg_x(nkt - 1);
    // the early waves' last barrier releases the late waves' last X; nobody reads LDS afterwards, and a wave that has ended no
    // longer counts in a barrier: the early waves' epilogue runs under the late waves' last 12 MFMAs
    if (!late) phase();
    __builtin_amdgcn_sched_barrier(0);

    if (active) {
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = 1.0f / l_tot;
        if (OUT == 1) {
            const bool row_ok = q0 + r < T;
            unsigned short* rowp = ctx16 + (size_t)(b * T + min(q0 + r, T - 1)) * (size_t)(2 * D) + (size_t)(ND * h) * 64;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    unsigned int w[2][4];
#pragma unroll
                    for (int gi = 0; gi < 2; ++gi) {
                        const int g = 2 * gp + gi;
                        _Float16 hh[4], ll[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) split_act(fmaf(oc[dt][4 * g + e], kInvLo, om[dt][4 * g + e]) * inv, hh[e], ll[e]);
                        w[gi][0] = pack_h2(hh[0], hh[1]); w[gi][1] = pack_h2(hh[2], hh[3]);
                        w[gi][2] = pack_h2(ll[0], ll[1]); w[gi][3] = pack_h2(ll[2], ll[3]);
                    }
                    unsigned int first[4], second[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(w[0][k], w[1][k], false, false);
                        first[k] = sw[0];
                        second[k] = sw[1];
                    }
                    if (row_ok) {
                        unsigned short* dst = rowp + dt * 64 + 8 * (2 * gp + kh);
                        *reinterpret_cast<u32x4*>(dst) = u32x4{first[0], first[1], second[0], second[1]};
                        *reinterpret_cast<u32x4*>(dst + 32) = u32x4{first[2], first[3], second[2], second[3]};
                    }
                }
        } else if (q0 + r < T) {
            const size_t off = (size_t)(b * T + q0 + r) * D + (size_t)h * DH + 4 * kh;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float val[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = fmaf(oc[dt][4 * g + e], kInvLo, om[dt][4 * g + e]) * inv;
                    *reinterpret_cast<f32x4*>(ctx + off + dt * 32 + 8 * g) = f32x4{val[0], val[1], val[2], val[3]};
                }
        }
    }
}

static int g_att_pp = -1;
int att_pp_set_option(long long value) { g_att_pp = (int)value; return PGMI_OK; }            // -1: by shape (see att_pp_serves), 0: never, 1: wherever the kernel is defined
bool att_pp_serves(int T, const float* conv, const float* slopes, int head_dim) {
    const int n32 = (T + 31) / 32;
    return g_att_pp != 0 && !conv && !slopes && head_dim == 64 && n32 >= 7;
}

int launch_att16_pp(const unsigned short* qk16, size_t qk_plane, const unsigned short* vt16, size_t vt_plane, const int32_t* kv_len,
                           int B, int T, int H, int Tp, float* ctx, unsigned short* ctx16, int out_mode, hipStream_t s) {
    const int n32 = (T + 31) / 32;
    const long long total = (long long)B * H * n32;
    if (total > 0x7FFFFFF0ll) { set_error("attention_f16x3_pp: too many query tiles"); return PGMI_EINVAL; }
    const int n_wg = (int)((total + PP_WAVES - 1) / PP_WAVES);
    constexpr size_t lds_bytes = (size_t)PP_NSTG * PP_STAGE_CH * 16;
    void (*kfn)(const unsigned short*, size_t, const unsigned short*, size_t, const int32_t*, int, int, int, float*, unsigned short*, int, int, int) =
        out_mode ? attention_f16x3_pp_kernel<1> : attention_f16x3_pp_kernel<0>;
    const int dbg = g_att_pp > 1 ? (g_att_pp - 1) / 16 : 0;           // timing probes (split-plane output only)
    if (dbg && out_mode) {
        switch (dbg) {
            case 1: kfn = attention_f16x3_pp_kernel<1, 1>; break;
            case 2: kfn = attention_f16x3_pp_kernel<1, 2>; break;
            case 4: kfn = attention_f16x3_pp_kernel<1, 4>; break;
            case 8: kfn = attention_f16x3_pp_kernel<1, 8>; break;
            case 6: kfn = attention_f16x3_pp_kernel<1, 6>; break;
            case 17: kfn = attention_f16x3_pp_kernel<1, 17>; break;
            case 14: kfn = attention_f16x3_pp_kernel<1, 14>; break;
            case 32: kfn = attention_f16x3_pp_kernel<1, 32>; break;
            case 64: kfn = attention_f16x3_pp_kernel<1, 64>; break;
            case 128: kfn = attention_f16x3_pp_kernel<1, 128>; break;
            case 130: kfn = attention_f16x3_pp_kernel<1, 130>; break;
            case 132: kfn = attention_f16x3_pp_kernel<1, 132>; break;
            default: set_error("attention_f16x3_pp: unknown probe %d", dbg); return PGMI_EINVAL;
        }
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PGMI_EHIP; }
    hipLaunchKernelGGL(kfn, dim3(n_wg), dim3(PP_WAVES * 64), lds_bytes, s, qk16, qk_plane, vt16, vt_plane, kv_len, T, H, Tp, ctx, ctx16, n32, (int)total, n_wg);
    return PGMI_OK;
}


}  // namespace pgmi
