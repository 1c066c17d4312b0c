// The software-pipelined form of the dense head_dim-64 split-fp16 attention (round 6).  Math, operand planes and the round-3 kernel whose
// arithmetic this one repeats row by row: attention_f16.hip.
#include <stdlib.h>
#include <string.h>
#include <algorithm>

#include "attention_f16_common.h"

namespace pgmi {

// ---- Round 6: the software-pipelined form of the dense head_dim-64 kernel -----------------------------------------------------------
// What bounds attention_f16x3_v2_kernel (measured, profiles/r6): per key tile a wave issues 12 S MFMAs, then ~150 VALU instructions of
// online softmax, then 12 P V MFMAs -- a dependent chain -- and the two waves of a SIMD (one from each of the CU's two workgroups) do NOT
// hide each other's phases: on gfx950 an MFMA stream of one wave and a VALU stream of its SIMD partner take at least the SUM of their
// times (tools/mfma_valu_pair.hip: 0.46 us + 0.40 us alone, 0.82 - 1.02 us side by side), whatever the priorities; a barrier-locked
// two-role workgroup built on that overlap was bit-identical and 12 - 23 % slower (git history, profiles/r6/att_ab_1_*).  What does
// overlap is VALU work in the shadow of THE SAME wave's MFMAs (0.59 us per MFMA + VALU unit with both waves of a SIMD running such a
// stream).  So this kernel gives every wave independent matrix and vector work in the same stretch of its instruction stream:
//     step kt:   P V of key tile kt - 1   (12 MFMAs: P was finished in step kt - 1)
//                softmax of key tile kt   (VALU: its scores were finished in step kt - 1)
//                S = K Q^T of tile kt + 1 (12 MFMAs, after the softmax has read the previous scores out of the accumulators)
// The arithmetic of a row is the v2 kernel's, operation for operation in the same order (the deferred rescale of O by alpha(kt) still
// sits between P V (kt - 1) and P V (kt)): bit-identical (tests/test_gpu_ops.py::test_attention_v3_bits_equal_v2), so which kernel serves
// a shape is a launch option ("att_v3").  LDS: the v2 ring, but a stage holds the pair the step needs TOGETHER: bundle m = {K tile m,
// V^T tile m - 2}, needed in step m - 1, issued two steps ahead.
template <int WPB, int OUT>
__global__ __launch_bounds__(WPB * 64, 2) void attention_f16x3_v3_kernel(
    const unsigned short* __restrict__ qk16, size_t qk_plane, const unsigned short* __restrict__ vt16, size_t vt_plane,
    const int32_t* __restrict__ kv_len, int T, int H, int Tp, float* __restrict__ ctx, unsigned short* __restrict__ ctx16,
    int dense_nblk, int nseq) {
    constexpr float defer_thr = kAttDefer;
    constexpr int DH = 64, NSTG = 3, KCPR = 8, KCH = AKT * KCPR, VCH = DH * 4, STG_CH = 2 * KCH + 2 * VCH;
    constexpr int NWI = STG_CH / 64, NDMA = (NWI + WPB - 1) / WPB, NS = 4, ND = 2;
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];   // [NSTG][STG_CH]
    int b, h, qblk;
    if (dense_nblk > 0) {                                         // XCD-local order (see the v2 kernel)
        const int within = (int)blockIdx.x % (8 * dense_nblk);
        const int pair = ((int)blockIdx.x / (8 * dense_nblk)) * 8 + (within & 7);
        if (pair >= nseq * H) return;
        qblk = within >> 3;
        b = pair / H;
        h = pair - b * H;
    } else {
        b = blockIdx.z;
        h = blockIdx.y;
        qblk = blockIdx.x;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kh = lane >> 5;
    const int D = H * DH;
    const int Tk = kv_len ? kv_len[b] : T;
    const int q0 = (qblk * WPB + wave) * 32;
    const bool active = q0 < T;
    const int nkt = (Tk + AKT - 1) / AKT;
    const size_t seq_halfs = (size_t)T * (2 * D), vt_halfs = (size_t)DH * Tp;
    const unsigned long long qk_bytes = std::min<unsigned long long>(0xFFFFFFFFull, (unsigned long long)qk_plane * 2ull + seq_halfs * 2ull);
    const unsigned long long vt_bytes = std::min<unsigned long long>(0xFFFFFFFFull, (unsigned long long)vt_plane * 2ull + vt_halfs * 2ull);
    const __amdgpu_buffer_rsrc_t rsQK = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(qk16) + (size_t)b * seq_halfs, 0, (int)(unsigned int)qk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsVT = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(vt16) + ((size_t)b * H + h) * vt_halfs, 0, (int)(unsigned int)vt_bytes, 0x00020000);

    // DMA map of a stage: the v2 kernel's (K hi | K lo | V^T hi | V^T lo, 16 wave-instructions of 1 KiB over WPB waves).  Instruction i
    // of wave w is wave-instruction w + WPB i: with WPB in {1, 2, 4} its tensor and plane are compile-time (i < 8 / WPB: K), only the
    // quarter of the plane it covers depends on the wave -- scalar arithmetic at issue time, no per-wave tables in SGPRs (a kernel that
    // runs out of SGPRs keeps uniform values in VGPRs and hipcc then wraps every DMA in a readfirstlane loop).
    static_assert(WPB == 1 || WPB == 2 || WPB == 4, "the DMA map needs WPB to divide 8");
    constexpr int NK = 8 / WPB;                                    // K instructions per wave and bundle; the rest are V^T
    int kvoff[NK], krow[NK], vvoff[NDMA - NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const int q4 = (wave + WPB * i) & 3;                       // quarter of the K plane: keys 8 q4 .. 8 q4 + 7
        const int key = q4 * 8 + (lane >> 3);
        krow[i] = key;
        kvoff[i] = ((lane & 7) ^ ((key >> 1) & 7)) * 16;
    }
#pragma unroll
    for (int i = 0; i < NDMA - NK; ++i) {
        const int q4 = (wave + WPB * (i + NK)) & 3;                // quarter of the V^T plane: dims 16 q4 .. 16 q4 + 15
        const int g = q4 * 64 + lane, d = g >> 2, c = (g & 3) ^ ((d >> 2) & 3);
        vvoff[i] = d * Tp * 2 + c * 16;
    }
    // bundle m -> stage m % 3: K tile min(m, nkt - 1) (rows clamped to the sequence's last token: finite, masked by Tk) and V^T tile
    // clamp(m - 2) -- the clamped copies are never read, they keep the per-wave DMA count constant for the counted waits
    // piece i (0 .. NDMA - 1) of bundle m into `stage`; on == false: the offset is pushed out of the descriptor's range -- the
    // instruction still counts in vmcnt (the counted waits need a constant number per step) but fetches nothing
    auto issue_piece = [&](int i, int m, int stage, bool on) {        // (i is a constant after unrolling at every call site)
        u32x4* base = lds + stage * STG_CH;
        const int kk = min(m, nkt - 1), vv = min(max(m - 2, 0), nkt - 1);
        if (i < NK) {
            const int wi = wave + WPB * i, p = (WPB * i) >> 2;
            const int kr = krow[i < NK ? i : 0], ko = kvoff[i < NK ? i : 0];
            const int vo = on ? min(kk * AKT + kr, T - 1) * (2 * D) * 2 + ko : -16;
            const int so = (int)((unsigned int)p * (unsigned int)qk_plane * 2u + (unsigned int)(D + h * DH) * 2u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQK, (__attribute__((address_space(3))) void*)(base + wi * 64), 16, vo, so, 0, 0);
        } else {
            const int wi = wave + WPB * i, p = (WPB * i - 8) >> 2;
            const int so = (int)((unsigned int)p * (unsigned int)vt_plane * 2u) + vv * ((AKT / 8) * 16);
            const int vv0 = vvoff[i >= NK ? i - NK : 0];
            const int vo = on ? vv0 : -16;             // (named locals: with an array expression written in the call hipcc 7.2 silently drops the kernel's host stub)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsVT, (__attribute__((address_space(3))) void*)(base + wi * 64), 16, vo, so, 0, 0);
        }
    };
    auto issue_bundle = [&](int m, int stage, bool on) {
#pragma unroll
        for (int i = 0; i < NDMA; ++i) issue_piece(i, m, stage, on);
    };
    const int last_bundle = nkt + 1;                              // V^T tile nkt - 1 travels in bundle nkt + 1

    // ---- prologue (the v2 kernel's): bundle 0 (K tile 0) into stage 0, the Q tiles through stages 1-2, then bundles 1 and 2 ----
    issue_bundle(0, 0, true);
    u32x4 qh[NS], ql[NS];
    {
        u32x4* qbase = lds + STG_CH + wave * (2 * KCH);
        if (active) {
            constexpr int NQ = 2 * KCH / 64;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int f = i * 64 + lane, pq = f / KCH, row = (f % KCH) / KCPR;
                const int c = (f % KCPR) ^ ((row >> 1) & 7);
                const int vo = (int)(((unsigned int)min(q0 + row, T - 1) * (unsigned int)(2 * D) + (unsigned int)(h * DH)) * 2u + (unsigned int)c * 16u);
                const int so = (int)((unsigned int)pq * (unsigned int)qk_plane * 2u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQK, (__attribute__((address_space(3))) void*)(qbase + i * 64), 16, vo, so, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (active) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int ci = r * KCPR + ((2 * s + kh) ^ ((r >> 1) & 7));
                qh[s] = qbase[ci];
                ql[s] = qbase[KCH + ci];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                               // every wave holds its Q in registers: stages 1 and 2 are free again; bundle 0 is visible
        asm volatile("" ::: "memory");
    }
    issue_bundle(1, 1, true);
    issue_bundle(2, 2, true);

    f32x16 om[ND], oc[ND], sm, sc;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int v = 0; v < 16; ++v) { om[dt][v] = 0.f; oc[dt][v] = 0.f; }
#pragma unroll
    for (int v = 0; v < 16; ++v) { sm[v] = 0.f; sc[v] = 0.f; }
    u32x4 ph[2], pl[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) { ph[m][e] = 0u; pl[m][e] = 0u; }
    float m_run = -INFINITY, l_run = 0.f;
    constexpr float kInvLo = 1.0f / kLoScale;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));

    // One step on the stage that holds bundle kt + 1 = {K tile kt + 1, V^T tile kt - 1}: P V of tile kt - 1, softmax of tile kt, scores
    // of tile kt + 1.  Only TWO full steps are instantiated (with and without the key mask of the sequence's last key tile) plus the
    // closing P V: every further variant made hipcc's register allocation worse (P and the Q fragments went to scratch with five).  The
    // first step runs its P V on P = 0 (exact zeros added to O = 0: same bits), the last one computes scores of a clamped K tile that
    // nobody reads.
    // The full step, in six sub-steps separated by sched_barriers (nothing moves across them): every sub-step issues the LDS reads of
    // the NEXT sub-step's MFMA operands (16 registers: one m of V^T, or one k16 slice s of K and Q), runs its own 6 or 3 MFMAs and its
    // share of the vector work -- at most 32 fragment registers live at any time (loading a whole tile's operands up front needs 64 - 96
    // and sent P and the Q fragments to scratch).
    auto vfrag = [&](const u32x4* Vb, int m, u32x4 (&fh)[ND], u32x4 (&fl)[ND]) {
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            const int d = dt * 32 + r;
            const int ci = d * 4 + ((2 * m + kh) ^ ((d >> 2) & 3));
            fh[dt] = Vb[ci];
            fl[dt] = Vb[VCH + ci];
        }
    };
    auto kfrag = [&](const u32x4* Kb, int s, u32x4& kh_, u32x4& kl_) {
        const int ci = r * KCPR + ((2 * s + kh) ^ ((r >> 1) & 7));
        kh_ = Kb[ci];
        kl_ = Kb[KCH + ci];
    };
    auto pv = [&](int m, const u32x4 (&fh)[ND], const u32x4 (&fl)[ND]) {
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
            oc[dt] = mfma_h(fh[dt], pl[m], oc[dt]);
            oc[dt] = mfma_h(fl[dt], ph[m], oc[dt]);
            om[dt] = mfma_h(fh[dt], ph[m], om[dt]);
        }
    };
    auto split_p = [&](const float (&st)[16], int m, int e) {        // P -> hi by truncation, lo = (p - hi) 2^11 (the v2 kernel's)
        const float p0 = st[8 * m + 2 * e], p1 = st[8 * m + 2 * e + 1];
        typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
        const fp16x2 hi2 = __builtin_amdgcn_cvt_pkrtz(p0, p1);
        const f32x2 ps = f32x2{p0, p1} * f32x2{kLoScale, kLoScale};
        const float l0 = fmaf((float)hi2[0], -kLoScale, ps[0]), l1 = fmaf((float)hi2[1], -kLoScale, ps[1]);
        const fp16x2 lo2 = __builtin_amdgcn_cvt_pkrtz(l0, l1);
        ph[m][e] = __builtin_bit_cast(unsigned int, hi2);
        pl[m][e] = __builtin_bit_cast(unsigned int, lo2);
    };
    // The rescale of O by alpha(kt) -- decided in step kt, due after P V (kt - 1) and before P V (kt) -- is applied at the HEAD of step
    // kt + 1 (and of the closing P V): a branch in the middle of a step would cut its straight-line block, and hipcc then sinks the exp2 /
    // split half of the softmax below the branch, behind all 24 MFMAs (the first build of this kernel: the v2 kernel's phases again).
    bool pend = false;
    float pend_alpha = 1.0f;
    auto apply_pending = [&]() {
        if (pend) {
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int v = 0; v < 16; ++v) { om[dt][v] *= pend_alpha; oc[dt][v] *= pend_alpha; }
        }
    };
    auto step = [&](auto MKc, int kt, const u32x4* stage, int dma_stage) {
        constexpr bool MK = decltype(MKc)::value;
        apply_pending();
        const u32x4* Kb = stage;
        const u32x4* Vb = stage + 2 * KCH;
        u32x4 va_h[ND], va_l[ND], vb_h[ND], vb_l[ND];
        u32x4 k0h, k0l, k1h, k1l;
        float st[16];
        // The step is ONE straight-line block cut into 24 slots by sched_barriers (nothing moves across them): a slot = one MFMA + the
        // slice of the softmax that rides in its shadow (at most ~7 single-issue instructions: a wave issues in order, so vector work
        // overlaps a matrix instruction only if it FOLLOWS it in the stream and fits under its 32 cycles) + the LDS reads of operands
        // three or more slots ahead.  Slots 0-11: P V (kt - 1); 12-23: scores of tile kt + 1.
#define PGMI_SLOT() __builtin_amdgcn_sched_barrier(0)
        // head: the V^T fragments are requested; while they travel: the scores of tile kt leave the accumulators (+ key mask)
        vfrag(Vb, 0, va_h, va_l);
        vfrag(Vb, 1, vb_h, vb_l);
#pragma unroll
        for (int v = 0; v < 16; ++v) st[v] = fmaf(sc[v], kInvLo, sm[v]);
        if constexpr (MK) {                                          // the sequence's last key tile
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int key = kt * AKT + (v & 3) + 8 * (v >> 2) + 4 * kh;
                if (key >= Tk) st[v] = -INFINITY;
            }
        }
        float ma = st[0];
#pragma unroll
        for (int v = 1; v < 8; ++v) ma = fmaxf(ma, st[v]);
        PGMI_SLOT();
        // ---- slots 0-5: P V (kt - 1), m = 0 ----  (measured and not kept: the whole row maximum before the first MFMA, which waits for
        //      the V^T fragments anyway: +2.7 % / +5.6 % over v2 at T = 288 / 1024 instead of +4.5 % / +9 %)
        oc[0] = mfma_h(va_h[0], pl[0], oc[0]);
        float mloc = st[8];
#pragma unroll
        for (int v = 9; v < 16; ++v) mloc = fmaxf(mloc, st[v]);
        mloc = fmaxf(mloc, ma);
        PGMI_SLOT();
        oc[0] = mfma_h(va_l[0], ph[0], oc[0]);
        {
            const unsigned int mu = __builtin_bit_cast(unsigned int, mloc);
            const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
            const unsigned int s0 = sw[0], s1 = sw[1];
            mloc = fmaxf(__builtin_bit_cast(float, s0), __builtin_bit_cast(float, s1));
        }
        const float m_new = fmaxf(m_run, mloc);
        PGMI_SLOT();
        om[0] = mfma_h(va_h[0], ph[0], om[0]);
        // deferred, per-row rescale (the v2 kernel's rule): alpha == 1 and the reference unchanged for a row that moved by less
        const bool moved = m_new > m_run + defer_thr;
        pend = !__all(m_new <= m_run + defer_thr);
        pend_alpha = moved ? __builtin_amdgcn_exp2f(m_run - m_new) : 1.0f;
        l_run *= pend_alpha;                                         // (x 1.0f is exact)
        asm volatile("" : "+v"(l_run));                              // NOT contracted with the row sum below: the v2 kernel's two roundings (hipcc fuses __fmul_rn too)
        m_run = moved ? m_new : m_run;
        const float mb = m_run - 10.0f;
        PGMI_SLOT();
        auto exp_pairs = [&](int v0) {                               // two pairs: packed subtract, two exp2 each
#pragma unroll
            for (int v = v0; v < v0 + 4; v += 2) {
                const f32x2 dlt = f32x2{st[v], st[v + 1]} - f32x2{mb, mb};
                st[v] = __builtin_amdgcn_exp2f(dlt[0]);
                st[v + 1] = __builtin_amdgcn_exp2f(dlt[1]);
            }
        };
        oc[1] = mfma_h(va_h[1], pl[0], oc[1]);
        exp_pairs(0);
        PGMI_SLOT();
        oc[1] = mfma_h(va_l[1], ph[0], oc[1]);
        exp_pairs(4);
        PGMI_SLOT();
        om[1] = mfma_h(va_h[1], ph[0], om[1]);
        exp_pairs(8);
        PGMI_SLOT();
        // ---- slots 6-11: P V (kt - 1), m = 1 ----
        oc[0] = mfma_h(vb_h[0], pl[1], oc[0]);
        exp_pairs(12);
        PGMI_SLOT();
        oc[0] = mfma_h(vb_l[0], ph[1], oc[0]);
        const float sa = ((st[0] + st[1]) + (st[2] + st[3])) + ((st[4] + st[5]) + (st[6] + st[7]));
        PGMI_SLOT();
        om[0] = mfma_h(vb_h[0], ph[1], om[0]);
        kfrag(Kb, 0, k0h, k0l);
        l_run += sa + (((st[8] + st[9]) + (st[10] + st[11])) + ((st[12] + st[13]) + (st[14] + st[15])));
        PGMI_SLOT();
        oc[1] = mfma_h(vb_h[1], pl[1], oc[1]);
        split_p(st, 0, 0);                                           // (P of m = 0 may be overwritten: its six MFMAs have issued)
        PGMI_SLOT();
        oc[1] = mfma_h(vb_l[1], ph[1], oc[1]);
        split_p(st, 0, 1);
        PGMI_SLOT();
        om[1] = mfma_h(vb_h[1], ph[1], om[1]);
        kfrag(Kb, 1, k1h, k1l);
        split_p(st, 0, 2);
        PGMI_SLOT();
        // ---- slots 12-23: scores of tile kt + 1, one k16 slice per three slots; the vector work ends in slot 16, the DMA of bundle
        //      kt + 3 (into the stage of bundle kt, free since this step's barrier) rides in the slots after it ----
        sc = mfma_h(k0h, ql[0], zero16);
        split_p(st, 0, 3);
        PGMI_SLOT();
        sc = mfma_h(k0l, qh[0], sc);
        split_p(st, 1, 0);
        PGMI_SLOT();
        sm = mfma_h(k0h, qh[0], zero16);
        kfrag(Kb, 2, k0h, k0l);
        split_p(st, 1, 1);
        PGMI_SLOT();
        sc = mfma_h(k1h, ql[1], sc);
        split_p(st, 1, 2);
        PGMI_SLOT();
        sc = mfma_h(k1l, qh[1], sc);
        split_p(st, 1, 3);
        PGMI_SLOT();
        sm = mfma_h(k1h, qh[1], sm);
        kfrag(Kb, 3, k1h, k1l);
        PGMI_SLOT();
        const bool dma_on = kt + 3 <= last_bundle;
        auto dma_slot = [&](int j) {                                 // pieces j, j + 6, j + 12 ... of the bundle
#pragma unroll
            for (int i = j; i < NDMA; i += 6) issue_piece(i, kt + 3, dma_stage, dma_on);
        };
        sc = mfma_h(k0h, ql[2], sc);
        dma_slot(0);
        PGMI_SLOT();
        sc = mfma_h(k0l, qh[2], sc);
        dma_slot(1);
        PGMI_SLOT();
        sm = mfma_h(k0h, qh[2], sm);
        dma_slot(2);
        PGMI_SLOT();
        sc = mfma_h(k1h, ql[3], sc);
        dma_slot(3);
        PGMI_SLOT();
        sc = mfma_h(k1l, qh[3], sc);
        dma_slot(4);
        PGMI_SLOT();
        sm = mfma_h(k1h, qh[3], sm);
        dma_slot(5);
#undef PGMI_SLOT
    };
    auto closing_pv = [&](const u32x4* stage) {                      // P V of the last key tile
        apply_pending();
        const u32x4* Vb = stage + 2 * KCH;
        u32x4 va_h[ND], va_l[ND], vb_h[ND], vb_l[ND];
        vfrag(Vb, 0, va_h, va_l);
        vfrag(Vb, 1, vb_h, vb_l);
        pv(0, va_h, va_l);
        pv(1, vb_h, vb_l);
    };
    using T1 = std::integral_constant<bool, true>;
    using T0 = std::integral_constant<bool, false>;

    // scores of tile 0 (bundle 0 is visible since the prologue barrier)
    if (active && nkt > 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ci = r * KCPR + ((2 * s + kh) ^ ((r >> 1) & 7));
            const u32x4 kfh = lds[ci], kfl = lds[KCH + ci];
            sc = mfma_h(kfh, ql[s], s == 0 ? zero16 : sc);
            sc = mfma_h(kfl, qh[s], sc);
            sm = mfma_h(kfh, qh[s], s == 0 ? zero16 : sm);
        }
    }
    auto wait_bundle = [&](int younger) {
        if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    int cur = 1;                                                     // stage of bundle kt + 1
    auto step_head = [&](int kt) {
        // bundle kt + 1 (issued two steps ago) has landed; bundle kt + 2 may stay in flight
        wait_bundle(1);
        __builtin_amdgcn_s_barrier();              // bundle kt + 1 visible to all waves; the stage of bundle kt (read in step kt - 1) is free
        asm volatile("" ::: "memory");
    };
    auto idle_dma = [&](int kt) {                                    // a wave without a query tile still moves its share of the bundles
        issue_bundle(kt + 3, cur == 0 ? NSTG - 1 : cur - 1, kt + 3 <= last_bundle);
    };
    for (int kt = 0; kt < nkt - 1; ++kt) {
        step_head(kt);
        if (active) step(T0{}, kt, lds + cur * STG_CH, cur == 0 ? NSTG - 1 : cur - 1);
        else idle_dma(kt);
        cur = (cur == NSTG - 1) ? 0 : cur + 1;
    }
    step_head(nkt - 1);                                              // the sequence's last key tile: masked by Tk
    if (active) step(T1{}, nkt - 1, lds + cur * STG_CH, cur == 0 ? NSTG - 1 : cur - 1);
    else idle_dma(nkt - 1);
    cur = (cur == NSTG - 1) ? 0 : cur + 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // bundle nkt + 1: the V^T tile of the last key tile
    asm volatile("" ::: "memory");
    if (active) closing_pv(lds + cur * STG_CH);

    // (Measured and not kept, profiles/r6/att_ab_4_*: the split-plane rows through a per-wave LDS transpose -- 16 lanes per 256-byte row, 8 lines
    // per store instruction instead of 32 partial ones, the GEMM's OUT 1 epilogue -- is within 0.5 % of the lane-exchange stores at every shape.)
    if (active) {
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = 1.0f / l_tot;
        if (OUT == 3) {
            store_ctx_bf16<ND>(om, oc, inv, kInvLo, q0 + r < T, ctx16 + (size_t)(b * T + min(q0 + r, T - 1)) * (size_t)D + (size_t)h * DH, kh);
        } else if (OUT == 1) {
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

// Launch option "att_v3": -1 (default) = by shape -- from seven query tiles per sequence on, where the interleaved A/B of scripts/att_bench.py
// has it ahead (profiles/r6/att_ab_3_*: +4.5 % at T = 288, +9 % at T = 502 / 1024, +7 % at T = 739 with key masks; +-2 % below) --, 0 = never,
// 1 = wherever the kernel is defined (dense, head_dim 64, no causal / ALiBi flavour).  Same bits either way.
static int g_att_v3 = -1;
void att_v3_set_option(int value) { g_att_v3 = value; }
bool att_v3_serves(int T, const float* conv, const float* slopes, int head_dim) {
    if (g_att_v3 == 0 || conv || slopes || head_dim != 64) return false;
    return g_att_v3 > 0 || (T + 31) / 32 >= 7;
}

template <int WPB, int OUT>
static void launch_att16v3_one(dim3 grid, const unsigned short* qk16, size_t qk_plane, const unsigned short* vt16, size_t vt_plane, const int32_t* kv_len,
                               int T, int H, int Tp, float* ctx, unsigned short* ctx16, hipStream_t s, int dense_nblk, int nseq) {
    constexpr size_t lds_bytes = (size_t)3 * A_STAGE * 16;
    auto kfn = attention_f16x3_v3_kernel<WPB, OUT>;
    if (lds_bytes > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL((attention_f16x3_v3_kernel<WPB, OUT>), grid, dim3(WPB * 64), lds_bytes, s, qk16, qk_plane, vt16, vt_plane, kv_len, T, H, Tp, ctx, ctx16,
                       dense_nblk, nseq);
}
int launch_att16v3(int out_mode, int wpb, dim3 grid, const unsigned short* qk16, size_t qk_plane, const unsigned short* vt16, size_t vt_plane,
                          const int32_t* kv_len, int T, int H, int Tp, float* ctx, unsigned short* ctx16, hipStream_t s, int dense_nblk, int nseq) {
#define PGMI_V3(W)                                                                                                                          \
    do {                                                                                                                                    \
        if (out_mode == 2) launch_att16v3_one<W, 3>(grid, qk16, qk_plane, vt16, vt_plane, kv_len, T, H, Tp, ctx, ctx16, s, dense_nblk, nseq); \
        else if (out_mode) launch_att16v3_one<W, 1>(grid, qk16, qk_plane, vt16, vt_plane, kv_len, T, H, Tp, ctx, ctx16, s, dense_nblk, nseq);   \
        else launch_att16v3_one<W, 0>(grid, qk16, qk_plane, vt16, vt_plane, kv_len, T, H, Tp, ctx, ctx16, s, dense_nblk, nseq);            \
    } while (0)
    switch (wpb) {
        case 1: PGMI_V3(1); break;
        case 2: PGMI_V3(2); break;
        default: PGMI_V3(4); break;
    }
#undef PGMI_V3
    return PGMI_OK;
}

}  // namespace pgmi
