// Fused multi-head self-attention on the 16-bit matrix pipe with split-fp16 (f16x3) operands.
//
// Math: q k^T, fp32 softmax, P v (reference: /root/reference/proteingym/baselines/esm/esm/multihead_attention.py:357-387;
// Tranception: tranception/model_pytorch.py:155-183).  One wave = 32 queries of one head, S^T = K Q^T so the query lives in the
// lane, P is consumed straight from the accumulator registers.  Every product is evaluated as
//     x y  ~=  x_hi y_hi + 2^-11 (x_hi y_lo + x_lo y_hi),   x_hi = fp16(x), x_lo = fp16((x - x_hi) 2^11)
// with v_mfma_f32_32x32x16_f16: 24 MFMAs per (32q x 32k) tile.  The 2^-11 terms are kept in their own accumulators and folded in
// fp32 (S = main + corr/2048 before the softmax, O likewise at the end), so no fp16 subnormal is ever produced.
// Operands arrive already split (from the fused QKV projection's epilogue, gemm16x_kernel.h OUT 2, or from the prep passes below):
//   qk16 [plane][M][2D]      q | k, row-major (hi plane, lo 2^11 plane); q carries head_dim^-1/2 log2(e)
//   vt16 [plane][B*H*64][Tp] V transposed per (sequence, head), keys of each 32-key tile stored with bits 2 and 3 of the key index
//                            swapped (the order the S^T accumulator holds them), pad keys = 0
#include <stdlib.h>
#include <string.h>
#include <algorithm>

#include "attention_f16_common.h"

namespace pgmi {

template <int WPB, int OUT, int NSTG, int DH = 64, bool RAG = false>
__global__ __launch_bounds__(WPB * 64) void attention_f16x3_v2_kernel(
    const unsigned short* __restrict__ qk16, size_t qk_plane, const unsigned short* __restrict__ vt16,
    size_t vt_plane, const int32_t* __restrict__ kv_len, const float* __restrict__ slopes, int T, int H,
    int Tp, float* __restrict__ ctx, unsigned short* __restrict__ ctx16, size_t plane, RagMap rag, int dense_nblk, int nseq) {
    constexpr float defer_thr = kAttDefer;
    // slopes != nullptr selects the Tranception flavour (tranception/model_pytorch.py:155-183): causal
    // mask (key <= query) and the grouped-ALiBi bias slope[h] * key added to the scaled scores.
    constexpr int NT = WPB * 64;
    // A tile is 16 wave-instructions of 64 chunks (K hi, K lo, V^T hi, V^T lo: 1024 x 16 B).  Every
    // wave issues the same number NDMA of them (counted vmcnt needs a per-wave constant): when
    // 16 % WPB != 0 the surplus slots re-issue instruction (i - 16), i.e. write identical bytes twice.
    constexpr int KCPR = DH / 8;                        // 16-byte chunks per key row of a K plane
    constexpr int KCH = AKT * KCPR, VCH = DH * 4;       // chunks per K plane / V^T plane of a tile
    constexpr int STG_CH = 2 * KCH + 2 * VCH;           // chunks per stage (16 KB at DH 64, 32 KB at DH 128)
    constexpr int NWI = STG_CH / 64;                    // wave-instructions per tile
    constexpr int NDMA = (NWI + WPB - 1) / WPB;
    constexpr int NS = DH / 16, ND = DH / 32;
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];   // [NSTG][STG_CH]

    // Dense launches (dense_nblk > 0) are ONE-dimensional in an XCD-local order (workgroup i runs on XCD i % 8): the dense_nblk query
    // blocks of one (sequence, head) are the workgroups i, i + 8, i + 16 ... of a group of 8 dense_nblk consecutive ones -- the same
    // XCD, dispatched together -- so that the K and V^T tiles every one of them streams reach that XCD's L2 once instead of once per
    // query block (the (nblk, H, B) grid, dense_nblk == 0, puts them on different XCDs: 2.45 x the algorithmic bytes fetched at
    // T = 288, profiles/r4).
    int b, h, qblk_dense;
    if (RAG) {
        b = rag.ent_seq[blockIdx.x];
        h = blockIdx.y;
        qblk_dense = 0;
    } else if (dense_nblk > 0) {
        const int within = (int)blockIdx.x % (8 * dense_nblk);
        const int pair = ((int)blockIdx.x / (8 * dense_nblk)) * 8 + (within & 7);
        if (pair >= nseq * H) return;
        qblk_dense = within >> 3;
        b = pair / H;
        h = pair - b * H;
    } else {
        b = blockIdx.z;
        h = blockIdx.y;
        qblk_dense = blockIdx.x;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: scalar branches, SGPR DMA bases
    const int r = lane & 31, kh = lane >> 5;
    const int D = H * DH;
    const int Tk = (!RAG && kv_len) ? kv_len[b] : T;
    // RAG: p0 = first token the sequence owns, a0 = its tile; row0 / rrow0 = operand row of ITS token a0 / of its root's token 0; kt0 = first
    // own key tile; orow0 + t = context (output) row of token t >= p0
    const int p0 = RAG ? rag.seq_p[b] : 0, a0 = p0 & ~(AKT - 1), kt0 = a0 / AKT;
    const int qblk = RAG ? rag.ent_j[blockIdx.x] : qblk_dense;
    const int row0 = RAG ? rag.seq_q[b] : b * T, rrow0 = RAG ? rag.seq_q[rag.seq_root[b]] : 0;
    const int orow0 = RAG ? rag.seq_off[b] - p0 : b * T;
    const int Tpo = RAG ? (T - a0 + 31) / 32 * 32 : Tp;            // row pitch of the sequence's own V^T block
    const int q0 = a0 + (qblk * WPB + wave) * 32;                  // absolute position of the wave's first query
    const bool active = q0 < T;
    // (Two workgroups share a CU, one wave of each per SIMD.  Measured and not kept, profiles/r3: a static priority for the wave in
    // the odd hardware slot, -3 %; the lane <-> lane + 32 max exchange through LDS instead of v_permlane32_swap, -1.3 %; 8-byte
    // epilogue stores, -8 %; Q fragments straight from global memory, -2 %.)

    // Q fragments: lane (r,kh) holds Q[q0+r][16s + 8kh .. +7] of both planes.  Loaded straight from the planes each of the 2 NS
    // 16-byte loads of a wave touches 32 different rows (one per lane pair): 8 x 32 row segments for an 8 KB tile.  With
    // kQviaLds the wave's Q tile is instead moved like a K tile -- 8 lanes per 128-byte row, DMA into LDS in the K tile's swizzled
    // image (ring stages 1 and 2 are still free) -- and the fragments are read from there: a quarter of the row segments on the
    // texture-address path, which this kernel keeps busy (9 % of wave-cycles with its FIFO full).
    constexpr bool q_lds = NSTG >= 3;
    u32x4 qh[NS], ql[NS];
    if (!q_lds) {
        const int qrow = min(q0 + r, T - 1);
        const unsigned short* qp = qk16 + ((size_t)row0 + (qrow - a0)) * (2 * D) + h * DH + kh * 8;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            qh[s] = *reinterpret_cast<const u32x4*>(qp + s * 16);
            ql[s] = *reinterpret_cast<const u32x4*>(qp + qk_plane + s * 16);
        }
    }

    // DMA map: LDS slot f (0 .. STG_CH-1 within a stage) <- global chunk
    //   f in [0, 2 KCH):        K plane p = f / KCH, key = (f % KCH) / KCPR, slot chunk c' = f % KCPR, source chunk c = c' ^ swz(key)
    //                           swz(key) = (key >> 1) & 7 for 128-byte rows (DH 64), key & 15 for 256-byte rows (DH 128)
    //   f in [2 KCH, STG_CH):   V^T plane p, row d = (g >> 2) % DH, c' = g & 3, source chunk c = c' ^ ((d >> 2) & 3)
    // Every wave-instruction (64 slots) lies inside ONE plane of one tensor, so the tensor, the plane and the (sequence, head,
    // tile) part of the address are wave-uniform: they go into the buffer descriptor and the SGPR offset, the lane keeps one
    // 32-bit byte offset per instruction -- no per-tile address arithmetic.
    // The descriptors are based at THIS block's sequence / (sequence, head) in the hi plane, so the 32-bit offsets only span one plane
    // stride plus one sequence (the launcher checks plane * 2 + extent < 4 GiB): operand arrays themselves may be larger than 4 GiB
    // (an MSA Transformer workspace of 1024 x 1024 tokens).  b and h are block-uniform: the bases stay in SGPRs.
    const size_t seq_halfs = (size_t)T * (2 * D), vt_halfs = (size_t)DH * Tp;
    const unsigned long long qk_bytes = std::min<unsigned long long>(0xFFFFFFFFull, (unsigned long long)qk_plane * 2ull + seq_halfs * 2ull);
    const unsigned long long vt_bytes = std::min<unsigned long long>(0xFFFFFFFFull, (unsigned long long)vt_plane * 2ull + vt_halfs * 2ull);
    const __amdgpu_buffer_rsrc_t rsQK = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(qk16) + (size_t)row0 * (2 * D), 0, (int)(unsigned int)qk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsVT = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(vt16) + (RAG ? (size_t)rag.seq_vt[b] + (size_t)h * DH * Tpo : ((size_t)b * H + h) * vt_halfs), 0, (int)(unsigned int)vt_bytes, 0x00020000);
    // RAG: the root's planes for the key tiles before kt0 (the root owns all of its T tokens: V^T row pitch Tp)
    const __amdgpu_buffer_rsrc_t rsQKr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(qk16) + (size_t)rrow0 * (2 * D), 0, (int)(unsigned int)qk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsVTr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(vt16) + (RAG ? (size_t)rag.seq_vt[rag.seq_root[b]] + (size_t)h * DH * Tp : (size_t)0), 0, (int)(unsigned int)vt_bytes, 0x00020000);
    // causal: keys beyond the block's last query tile are never needed (uniform bound for the block)
    const bool causal = slopes != nullptr;
    const int last_q = min(T, a0 + (qblk * WPB + WPB) * 32);
    const int nkt = causal ? (min(Tk, last_q) + AKT - 1) / AKT : (Tk + AKT - 1) / AKT;
    // voff_last: the same offsets for the LAST key tile with its K rows clamped to the sequence's last row (finite, masked by Tk) --
    // the only tile that can reach past row T-1, i.e. into the next sequence or past the end of the operand
    int voff[NDMA], voff_last[NDMA], voff_root[NDMA], sbase[NDMA], sstep[NDMA], slot0[NDMA];
    bool is_k[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        int wi = wave + WPB * i;                          // wave-instruction index 0 .. NWI-1 (+ duplicates)
        if (wi >= NWI) wi -= NWI;
        slot0[i] = wi * 64;
        is_k[i] = wi < 2 * KCH / 64;
        if (is_k[i]) {
            const int p = wi / (KCH / 64), key = ((wi % (KCH / 64)) * 64 + lane) / KCPR;
            const int c = (lane % KCPR) ^ (DH == 64 ? ((key >> 1) & 7) : (key & 15));
            voff[i] = voff_root[i] = key * (2 * D) * 2 + c * 16;
            voff_last[i] = min(key, T - 1 - (nkt - 1) * AKT) * (2 * D) * 2 + c * 16;
            sbase[i] = (int)((unsigned int)p * (unsigned int)qk_plane * 2u + (unsigned int)(D + h * DH) * 2u);
            sstep[i] = AKT * (2 * D) * 2;
        } else {
            const int wv = wi - 2 * KCH / 64, p = wv / (VCH / 64), g = (wv % (VCH / 64)) * 64 + lane;
            const int d = g >> 2, c = (g & 3) ^ ((d >> 2) & 3);
            voff[i] = voff_last[i] = d * Tpo * 2 + c * 16;
            voff_root[i] = d * Tp * 2 + c * 16;
            sbase[i] = (int)((unsigned int)p * (unsigned int)vt_plane * 2u);
            sstep[i] = (AKT / 8) * 16;
        }
    }
    auto issue_tile = [&](int kt, int buf) {
        u32x4* base = lds + buf * STG_CH;
        const bool from_root = RAG && kt < kt0;               // wave-uniform: a scalar select of the descriptor
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int vo = from_root ? voff_root[i] : ((kt == nkt - 1) ? voff_last[i] : voff[i]);
            const int so = sbase[i] + (from_root ? kt : kt - kt0) * sstep[i];
            // (soffset goes through a named local: with the array expression written in the call hipcc 7.2 silently drops the kernel's host stub)
            if (is_k[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(from_root ? rsQKr : rsQK, (__attribute__((address_space(3))) void*)(base + slot0[i]), 16, vo, so, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(from_root ? rsVTr : rsVT, (__attribute__((address_space(3))) void*)(base + slot0[i]), 16, vo, so, 0, 0);
        }
    };

    // NSTG-deep LDS ring, K/V tiles prefetched NSTG-1 ahead with counted vmcnt (the DMAs stay in
    // flight across the barrier; a __syncthreads() would drain them)
    constexpr float kLog2e = 1.4426950408889634f;
    const float slope2 = causal ? slopes[h] * kLog2e : 0.0f;      // the q planes carry log2(e) (kQLog2e), the ALiBi term must too
    if (q_lds) {
        issue_tile(0, 0);
        u32x4* qbase = lds + STG_CH + wave * (2 * KCH);             // this wave's half of stages 1 .. 2: [2 planes][32 rows][KCPR chunks]
        if (active) {
            constexpr int NQ = 2 * KCH / 64;                        // wave-instructions of 1 KiB
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int f = i * 64 + lane, pq = f / KCH, row = (f % KCH) / KCPR;
                const int c = (f % KCPR) ^ (DH == 64 ? ((row >> 1) & 7) : (row & 15));
                const int vo = (int)(((unsigned int)(min(q0 + row, T - 1) - a0) * (unsigned int)(2 * D) + (unsigned int)(h * DH)) * 2u + (unsigned int)c * 16u);
                const int so = (int)((unsigned int)pq * (unsigned int)qk_plane * 2u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQK, (__attribute__((address_space(3))) void*)(qbase + i * 64), 16, vo, so, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's share of tile 0 and its own Q tile have landed
        if (active) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int ci = r * KCPR + ((2 * s + kh) ^ (DH == 64 ? ((r >> 1) & 7) : (r & 15)));
                qh[s] = qbase[ci];
                ql[s] = qbase[KCH + ci];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                               // every wave holds its Q in registers: stages 1 and 2 are free again
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 1; t < NSTG - 1; ++t)
            if (t < nkt) issue_tile(t, t);
    } else {
#pragma unroll
        for (int t = 0; t < NSTG - 1; ++t)
            if (t < nkt) issue_tile(t, t);
    }

    f32x16 om[ND], oc[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int v = 0; v < 16; ++v) { om[dt][v] = 0.f; oc[dt][v] = 0.f; }
    // base-2 online softmax with P scaled by 2^10 (keeps every P hi in fp16's normal range; the
    // factor cancels in O / l because l accumulates the same scaled P)
    float m_run = -INFINITY, l_run = 0.f;
    constexpr float kInvLo = 1.0f / kLoScale;

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto wanted = [&](int kt) -> bool { return active && !(causal && kt * AKT > q0 + 31); };   // causal: tiles above the diagonal are skipped
    // S^T = K Q^T of one key tile: main and 2^-11 correction accumulators
    auto scores = [&](int buf, f32x16& sm, f32x16& sc) {
        const u32x4* Kb = lds + buf * STG_CH;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ci = r * KCPR + ((2 * s + kh) ^ (DH == 64 ? ((r >> 1) & 7) : (r & 15)));
            const u32x4 kfh = Kb[ci], kfl = Kb[KCH + ci];
            sc = mfma_h(kfh, ql[s], s == 0 ? zero16 : sc);       // s == 0: the accumulator operand is the inline constant 0
            sc = mfma_h(kfl, qh[s], sc);
            sm = mfma_h(kfh, qh[s], s == 0 ? zero16 : sm);
        }
    };
    // tile t of this wave's DMA share has landed once at most `younger` tiles issued after it remain in flight
    auto wait_tile = [&](int younger) {
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        wait_tile(min(NSTG - 2, nkt - 1 - kt));
        __builtin_amdgcn_s_barrier();          // tile kt visible to all waves; slot of tile kt-1 is free
        asm volatile("" ::: "memory");
        if (kt + NSTG - 1 < nkt) issue_tile(kt + NSTG - 1, (cur == 0) ? NSTG - 1 : cur - 1);
        if (wanted(kt)) {
            const u32x4* Vb = lds + cur * STG_CH + 2 * KCH;
            f32x16 sm, sc;
            scores(cur, sm, sc);
            float st[16];
            if (causal) {
                // ALiBi: slope * key index (model_pytorch.py:167-168), key = 32 kt + 4 kh + c_v with c_v a compile-time constant per
                // accumulator register; the causal mask (:161-165) only on tiles that reach beyond the wave's first query -- every
                // tile below the diagonal band is fully visible and skips the 16 compares and selects
                const float kb = (float)(kt * AKT + 4 * kh);                // key index as a float: small integers, every sum below is exact
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    st[v] = fmaf(slope2, kb + (float)((v & 3) + 8 * (v >> 2)), fmaf(sc[v], kInvLo, sm[v]));
                if (kt * AKT + AKT - 1 > q0) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int key = kt * AKT + (v & 3) + 8 * (v >> 2) + 4 * kh;
                        if (key > q0 + r) st[v] = -INFINITY;
                    }
                }
            } else {
#pragma unroll
                for (int v = 0; v < 16; ++v) st[v] = fmaf(sc[v], kInvLo, sm[v]);
            }
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
            {                                             // v_permlane32_swap: both halves of the query's row without an LDS round trip
                const unsigned int mu = __builtin_bit_cast(unsigned int, mloc);
                const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
                const unsigned int s0 = sw[0], s1 = sw[1];
                mloc = fmaxf(__builtin_bit_cast(float, s0), __builtin_bit_cast(float, s1));
            }
            const float m_new = fmaxf(m_run, mloc);
            // Deferred rescale: the running reference m_run may lag the true row maximum by up to defer_thr (base-2 units).  P is
            // then up to 2^(10 + defer_thr) instead of 2^10 -- still exact to 22 bits in its hi | lo split and, for defer_thr <= 5,
            // inside fp16's range -- and O and l carry the same factor, which cancels in O / l.  The O-wide rescale (36 packed
            // multiplies on 64 accumulator registers + their wait for the previous P V MFMAs) then runs on the first key tiles
            // only instead of on nearly every tile.  defer_thr = 0: rescale whenever any lane's maximum moved (exact running max).
            // The branch is wave-wide (some row moved by more than defer_thr), the new reference is PER ROW: a row that moved by less keeps
            // alpha == 1 and its reference, so its bits depend on its own scores only -- not on which rows share its wave.
            if (!__all(m_new <= m_run + defer_thr)) {
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
            for (int v = 0; v < 16; v += 2) {                    // packed fp32 subtract (v_pk_add_f32): same values, half the issue slots
                const f32x2 dlt = f32x2{st[v], st[v + 1]} - f32x2{mb, mb};
                st[v] = __builtin_amdgcn_exp2f(dlt[0]);          // P * 2^10 in [0, 1024 * 2^defer_thr]
                st[v + 1] = __builtin_amdgcn_exp2f(dlt[1]);
            }
            // row sum as a tree (four independent chains of depth 2 + 2): a 16-long serial add chain is 16 dependent-issue latencies
            l_run += ((st[0] + st[1]) + (st[2] + st[3])) + ((st[4] + st[5]) + (st[6] + st[7])) +
                     (((st[8] + st[9]) + (st[10] + st[11])) + ((st[12] + st[13]) + (st[14] + st[15])));
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                u32x4 ph, pl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {                    // hi by truncation (pkrtz), lo = (p - hi) 2^11
                    const float p0 = st[8 * m + 2 * e], p1 = st[8 * m + 2 * e + 1];
                    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
                    const fp16x2 hi2 = __builtin_amdgcn_cvt_pkrtz(p0, p1);
                    // (p - hi) 2^11 as one mixed-precision fma on the fp16 hi (v_fma_mix_f32): p 2^11 (one packed multiply for
                    // the pair) and the fma are both exact
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 ps = f32x2{p0, p1} * f32x2{kLoScale, kLoScale};
                    const float l0 = fmaf((float)hi2[0], -kLoScale, ps[0]), l1 = fmaf((float)hi2[1], -kLoScale, ps[1]);
                    const fp16x2 lo2 = __builtin_amdgcn_cvt_pkrtz(l0, l1);
                    ph[e] = __builtin_bit_cast(unsigned int, hi2);
                    pl[e] = __builtin_bit_cast(unsigned int, lo2);
                }
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    const int d = dt * 32 + r;
                    const int ci = d * 4 + ((2 * m + kh) ^ ((d >> 2) & 3));
                    const u32x4 vfh = Vb[ci], vfl = Vb[VCH + ci];
                    oc[dt] = mfma_h(vfh, pl, oc[dt]);
                    oc[dt] = mfma_h(vfl, ph, oc[dt]);
                    om[dt] = mfma_h(vfh, ph, om[dt]);
                }
            }
        }
        cur = (cur == NSTG - 1) ? 0 : cur + 1;
    }

    if (active) {
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        if (OUT == 3) {          // one bf16 plane, row-major (attention_f16_common.h)
            store_ctx_bf16<ND>(om, oc, 1.0f / l_tot, kInvLo, q0 + r < T && q0 + r >= p0,
                               ctx16 + (size_t)(orow0 + max(min(q0 + r, T - 1), p0)) * (size_t)D + (size_t)h * DH, kh);
        } else if (OUT == 1) {
            // Split-plane output, 16-byte stores: lane (r, kh) holds columns 8g + 4kh .. + 3 of its query row for g = 0 .. 3; one
            // v_permlane32_swap per dword hands lane (r, 0) its partner's half of an even g and lane (r, 1) its partner's half of
            // the following odd g, so every lane owns 8 consecutive columns = one dwordx4 per plane: 8 store instructions per
            // lane instead of 16 of half the width (a row-per-lane store touches 32-64 lines per instruction: the epilogue is bound
            // by store issue, not by bytes).  All 64 lanes take part in the swaps; rows beyond T only skip the stores.
            const float inv = 1.0f / l_tot;
            const bool row_ok = q0 + r < T && q0 + r >= p0;
            unsigned short* rowp = ctx16 + (size_t)(orow0 + max(min(q0 + r, T - 1), p0)) * (size_t)(2 * D) + (size_t)(ND * h) * 64;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    unsigned int w[2][4];                        // [g parity][hi0 hi1 lo0 lo1]
#pragma unroll
                    for (int gi = 0; gi < 2; ++gi) {
                        const int g = 2 * gp + gi;
                        _Float16 hh[4], ll[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) split_act(fmaf(oc[dt][4 * g + e], kInvLo, om[dt][4 * g + e]) * inv, hh[e], ll[e]);
                        w[gi][0] = pack_h2(hh[0], hh[1]); w[gi][1] = pack_h2(hh[2], hh[3]);
                        w[gi][2] = pack_h2(ll[0], ll[1]); w[gi][3] = pack_h2(ll[2], ll[3]);
                    }
                    unsigned int first[4], second[4];            // columns c .. c + 3 and c + 4 .. c + 7 of this lane's 8-column run
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(w[0][k], w[1][k], false, false);
                        first[k] = sw[0];                        // kh 0: own g even          kh 1: partner's (kh 0) g odd
                        second[k] = sw[1];                       // kh 0: partner's g even    kh 1: own g odd
                    }
                    if (row_ok) {
                        unsigned short* dst = rowp + dt * 64 + 8 * (2 * gp + kh);
                        *reinterpret_cast<u32x4*>(dst) = u32x4{first[0], first[1], second[0], second[1]};
                        *reinterpret_cast<u32x4*>(dst + 32) = u32x4{first[2], first[3], second[2], second[3]};
                    }
                }
        } else if (q0 + r < T && q0 + r >= p0) {
            const float inv = 1.0f / l_tot;
            const size_t off = (size_t)(orow0 + q0 + r) * D + (size_t)h * DH + 4 * kh;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float val[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = fmaf(oc[dt][4 * g + e], kInvLo, om[dt][4 * g + e]) * inv;
                    const size_t oo = off + dt * 32 + 8 * g;
                    if constexpr (OUT == 0) {
                        *reinterpret_cast<f32x4*>(ctx + oo) = f32x4{val[0], val[1], val[2], val[3]};
                    } else {
                        _Float16 hh[4], ll[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) split_act(val[e], hh[e], ll[e]);
                        // K-interleaved GEMM operand (common.h ki_off): column h*DH + dt*32 + 8g + 4kh of a row of D
                        unsigned short* dst = ctx16 + (size_t)(orow0 + q0 + r) * (size_t)(2 * D) + (size_t)(ND * h + dt) * 64 + 8 * g + 4 * kh;
                        *reinterpret_cast<u32x2*>(dst) = u32x2{pack_h2(hh[0], hh[1]), pack_h2(hh[2], hh[3])};
                        *reinterpret_cast<u32x2*>(dst + 32) = u32x2{pack_h2(ll[0], ll[1]), pack_h2(ll[2], ll[3])};
                    }
                }
        }
    }
}

// Measured and NOT kept (round 5, profiles/r5/att_ab_2_persistent_kernel_with_prefetch.log; the code is in git history, commit "experiment:
// persistent attention kernel"): the PERSISTENT form of the dense head_dim-64 kernel -- a workgroup walks a contiguous run of (sequence, head,
// query block) items, the next item's first K / V^T tile and Q tile in flight (DMA) under the current item's epilogue arithmetic, its stores
// issued after the next Q tile is read; bits equal to this kernel's -- is 9 % slower at T = 288, 13 % at T = 1024, 5 % at T = 152: the per-item
// state (descriptors, bounds) becomes mutable and leaves the SGPRs (54 spilled), 225 VGPRs instead of 188, one more barrier per item, and what the
// prefetch hides is small beside that.  The XCD-local block ORDER alone (the query blocks of one (sequence, head) on one XCD, K / V^T fetched
// into its L2 once) is +2.5 % at T = 1024 and -2 % at T = 288 (att_ab_1_xcd_local_order.log): fetch traffic is not what bounds this kernel.

template <int WPB, int OUT, int NSTG, int DH, bool RAG = false>
static int launch_att16v2_one(dim3 grid, const unsigned short* qk16, size_t qk_plane, const unsigned short* vt16, size_t vt_plane,
                              const int32_t* kv_len, const float* slopes, int T, int H, int Tp, float* ctx, unsigned short* ctx16,
                              size_t plane, hipStream_t s, RagMap rag = RagMap{}, int dense_nblk = 0, int nseq = 0) {
    constexpr size_t lds_bytes = (size_t)NSTG * (DH * 16) * 16;          // stage = DH * 16 chunks of 16 B
    auto kfn = attention_f16x3_v2_kernel<WPB, OUT, NSTG, DH, RAG>;
    if (lds_bytes > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PGMI_EHIP; }
    }
    hipLaunchKernelGGL(kfn, grid, dim3(WPB * 64), lds_bytes, s, qk16, qk_plane, vt16, vt_plane, kv_len, slopes, T, H, Tp, ctx, ctx16, plane, rag, dense_nblk, nseq);
    return PGMI_OK;
}

template <int OUT, int NSTG>
static int launch_att16v2_mode(int wpb, dim3 grid, const unsigned short* qk16, size_t qk_plane,
                               const unsigned short* vt16, size_t vt_plane, const int32_t* kv_len,
                               const float* slopes, int T, int H, int Tp, float* ctx, unsigned short* ctx16,
                               size_t plane, hipStream_t s, int dense_nblk, int nseq) {
    const RagMap none{};
    switch (wpb) {
        case 1: return launch_att16v2_one<1, OUT, NSTG, 64>(grid, qk16, qk_plane, vt16, vt_plane, kv_len, slopes, T, H, Tp, ctx, ctx16, plane, s, none, dense_nblk, nseq);
        case 2: return launch_att16v2_one<2, OUT, NSTG, 64>(grid, qk16, qk_plane, vt16, vt_plane, kv_len, slopes, T, H, Tp, ctx, ctx16, plane, s, none, dense_nblk, nseq);
        case 3: return launch_att16v2_one<3, OUT, NSTG, 64>(grid, qk16, qk_plane, vt16, vt_plane, kv_len, slopes, T, H, Tp, ctx, ctx16, plane, s, none, dense_nblk, nseq);
        default: return launch_att16v2_one<4, OUT, NSTG, 64>(grid, qk16, qk_plane, vt16, vt_plane, kv_len, slopes, T, H, Tp, ctx, ctx16, plane, s, none, dense_nblk, nseq);
    }
}

// Launch option (pgmi_set_option "att_xcd_local", default 1): 1 = the one-dimensional XCD-local order of the dense launches, 0 = the
// (query block, head, sequence) grid of rounds 1-4 (kept for the interleaved A/B of scripts/att_bench.py: block order does not touch a
// row's arithmetic, same bits).
static int g_att_xcd_local = -1;     // -1: by shape -- XCD-local from two query blocks per sequence on.  (Round 5, v2 kernel: +2.5 % at T = 1024, -2 % at
                                     // T = 288, so it was used from eight blocks on; with the pipelined kernel, round 6: +5 % at T = 152, +3.5 % at 230 / 256, 0 at 288,
                                     // +1.6 / +0.7 / +3 % at 322 / 352 / 382, +3 % at 502, +2 % at 739, +6 % at 902; causal: -1 ... +3 %: profiles/r6/att_ab_6_*)
int att_set_option(const char* name, long long value) {
    if (!strcmp(name, "att_xcd_local")) { g_att_xcd_local = (int)value; return PGMI_OK; }
    if (!strcmp(name, "att_v3")) { att_v3_set_option((int)value); return PGMI_OK; }
    return PGMI_EINVAL;
}
// grid of a dense launch of nblk query blocks x H heads x B sequences, and the dense_nblk argument that goes with it
static dim3 dense_grid(int nblk, int H, int B, int* dense_nblk) {
    if (g_att_xcd_local == 0 || (g_att_xcd_local < 0 && nblk < 2)) { *dense_nblk = 0; return dim3(nblk, H, B); }
    *dense_nblk = nblk;
    const long long pairs = (long long)H * B, groups = (pairs + 7) / 8;
    return dim3((unsigned)(groups * 8 * nblk), 1, 1);
}

// Query tiles (waves) per workgroup for sequences of T tokens, head_dim 64: the tiles are spread evenly over ceil(n / 4) blocks
int att16_waves_per_block(int T) {
    const int n32 = (T + 31) / 32, nblk = (n32 + 3) / 4;
    const int wpb = (n32 + nblk - 1) / nblk;
    return wpb == 3 ? 4 : wpb;                    // measured: a 4th (idle) wave that only helps loading beats 3-wave blocks
}

// qkv fp32 [B*T, 3D] -> (rotary) -> split planes -> attention.  Scratch: qk16 2 planes of B*T*2D
// halfs (plane stride qk_plane), vt16 2 planes of B*H*64*Tp halfs (plane stride vt_plane), Tp = T
// rounded up to 32.
int launch_attention_f16x3_v2(const float* qkv, const int32_t* kv_len, const float* cos_t, const float* sin_t,
                              int rotary, int B, int T, int H, unsigned short* qk16, size_t qk_plane,
                              unsigned short* vt16, size_t vt_plane, float* ctx, unsigned short* ctx16, size_t plane,
                              int out_mode, hipStream_t s, const float* conv, const float* slopes, int head_dim) {
    if (B <= 0 || T <= 0 || H <= 0 || out_mode < 0 || out_mode > 2 || (head_dim != 64 && head_dim != 128)) {      // out_mode 2: one bf16 plane
        set_error("attention_f16x3_v2: bad arguments B=%d T=%d H=%d out=%d head_dim=%d", B, T, H, out_mode, head_dim);
        return PGMI_EINVAL;
    }
    const int n32 = (T + 31) / 32, Tp = n32 * 32;
    int rc = PGMI_OK;
    // The K / V^T DMA addresses both planes of ONE sequence / (sequence, head) through a buffer descriptor based there, with 32-bit
    // byte offsets: the lo plane sits one plane stride further, so plane stride + one sequence must stay below 4 GiB (the arrays
    // themselves may be larger).  Every supported model fits with room (ESM2-15B at 98 304 rows: 2.0e9 bytes per q|k plane; an MSA
    // Transformer workspace of 1024 x 1024 tokens: 3.2e9); anything beyond is refused here instead of wrapping.
    {
        const unsigned long long seq_bytes = (unsigned long long)T * 2ull * (unsigned long long)H * (unsigned long long)head_dim * 2ull;
        const unsigned long long vt_bytes = (unsigned long long)head_dim * (unsigned long long)Tp * 2ull;
        if ((unsigned long long)qk_plane * 2ull + seq_bytes >= (1ull << 32) || (unsigned long long)vt_plane * 2ull + vt_bytes >= (1ull << 32)) {
            set_error("attention_f16x3_v2: operand planes of %zu / %zu halfs exceed the 32-bit offset range of the K / V^T DMA "
                      "(plane stride + one sequence must stay below 4 GiB: create the model with a smaller max_rows)", qk_plane, vt_plane);
            return PGMI_EINVAL;
        }
    }
    if (head_dim == 128) {
        // ESM2-15B class: H heads of 128 = 2 H slot groups of 64 in the operand planes; only the fused-QKV operand path
        // (no prep pass), no causal / ALiBi flavour; 2-stage ring (2 x 32 KB) so that two workgroups share a CU
        if (qkv || conv || slopes) { set_error("attention_f16x3_v2: head_dim 128 needs operands from the fused QKV projection"); return PGMI_EINVAL; }
        const int nblk = (n32 + 3) / 4;
        int dn = 0;
        const dim3 grid = dense_grid(nblk, H, B, &dn);
        if (out_mode == 0) rc = launch_att16v2_one<4, 0, 3, 128>(grid, qk16, qk_plane, vt16, vt_plane, kv_len, nullptr, T, H, Tp, ctx, ctx16, plane, s, RagMap{}, dn, B);
        else if (out_mode == 2) rc = launch_att16v2_one<4, 3, 3, 128>(grid, qk16, qk_plane, vt16, vt_plane, kv_len, nullptr, T, H, Tp, ctx, ctx16, plane, s, RagMap{}, dn, B);
        else rc = launch_att16v2_one<4, 1, 3, 128>(grid, qk16, qk_plane, vt16, vt_plane, kv_len, nullptr, T, H, Tp, ctx, ctx16, plane, s, RagMap{}, dn, B);
        if (rc) return rc;
        PGMI_HIP(hipGetLastError());
        return PGMI_OK;
    }
    if (qkv && conv && rotary) { set_error("attention_f16x3_v2: depth-wise convolution and rotary together are not a model this library knows"); return PGMI_EINVAL; }
    if (qkv && conv)       // Tranception: LDS-staged depth-wise conv + split
        launch_qkv_prep_conv(dim3(n32, H, B), s, qkv, conv, T, H, Tp, qk16, qk_plane, vt16, vt_plane, nullptr);
    else if (qkv)          // operands not prepared by the fused QKV epilogue: run the prep pass
        launch_qkv_prep(dim3(n32, H, B), s, qkv, cos_t, sin_t, rotary, T, H, Tp, qk16, qk_plane, vt16, vt_plane);
    const int wpb = att16_waves_per_block(T), nblk = (n32 + wpb - 1) / wpb;
    int dn = 0;
    const dim3 grid = dense_grid(nblk, H, B, &dn);
    if (att_v3_serves(T, conv, slopes, head_dim)) rc = launch_att16v3(out_mode, wpb, grid, qk16, qk_plane, vt16, vt_plane, kv_len, T, H, Tp, ctx, ctx16, s, dn, B);
    else if (out_mode == 0) rc = launch_att16v2_mode<0, 3>(wpb, grid, qk16, qk_plane, vt16, vt_plane, kv_len, slopes, T, H, Tp, ctx, ctx16, plane, s, dn, B);
    else if (out_mode == 2) rc = launch_att16v2_mode<3, 3>(wpb, grid, qk16, qk_plane, vt16, vt_plane, kv_len, slopes, T, H, Tp, ctx, ctx16, plane, s, dn, B);
    else rc = launch_att16v2_mode<1, 3>(wpb, grid, qk16, qk_plane, vt16, vt_plane, kv_len, slopes, T, H, Tp, ctx, ctx16, plane, s, dn, B);
    if (rc) return rc;
    PGMI_HIP(hipGetLastError());
    return PGMI_OK;
}

// Tranception prefix-shared scoring (RagMap): depth-wise conv prep over a list of n_tiles (sequence, 32-token tile) entries, then causal
// grouped-ALiBi attention over a list of n_blocks (sequence, block of att16_waves_per_block(T) query tiles) entries; split-plane context rows out (packed rows).
int launch_attention_tr_ragged(const float* qkv, const float* conv, const float* slopes, int T, int H, const AttRagged& rg,
                               unsigned short* qk16, size_t qk_plane, unsigned short* vt16, size_t vt_plane, unsigned short* ctx16,
                               size_t plane, hipStream_t s) {
    if (T <= 0 || H <= 0 || !qkv || !conv || !slopes || rg.n_tiles <= 0 || rg.n_blocks <= 0) {
        set_error("attention_tr_ragged: bad arguments T=%d H=%d tiles=%d blocks=%d", T, H, rg.n_tiles, rg.n_blocks);
        return PGMI_EINVAL;
    }
    const int Tp = (T + 31) / 32 * 32;
    {
        const unsigned long long seq_bytes = (unsigned long long)T * 2ull * (unsigned long long)H * 64ull * 2ull;
        const unsigned long long vt_bytes = 64ull * (unsigned long long)Tp * 2ull;
        if ((unsigned long long)qk_plane * 2ull + seq_bytes >= (1ull << 32) || (unsigned long long)vt_plane * 2ull + vt_bytes >= (1ull << 32)) {
            set_error("attention_tr_ragged: operand planes exceed the 32-bit offset range of the K / V^T DMA (create the model with a smaller max_rows)");
            return PGMI_EINVAL;
        }
    }
    const RagMap tiles{rg.seq_off, rg.seq_p, rg.seq_q, rg.seq_root, rg.seq_vt, rg.tile_seq, rg.tile_j};
    launch_qkv_prep_conv(dim3(rg.n_tiles, H, 1), s, qkv, conv, T, H, Tp, qk16, qk_plane, vt16, vt_plane, &tiles);
    // the same instantiation (waves per block) as the dense launch of T tokens: a row is computed by the same code
    const RagMap rag{rg.seq_off, rg.seq_p, rg.seq_q, rg.seq_root, rg.seq_vt, rg.blk_seq, rg.blk_j};
    const dim3 grid(rg.n_blocks, H, 1);
    int rc;
    switch (att16_waves_per_block(T)) {
        case 1: rc = launch_att16v2_one<1, 1, 3, 64, true>(grid, qk16, qk_plane, vt16, vt_plane, nullptr, slopes, T, H, Tp, nullptr, ctx16, plane, s, rag); break;
        case 2: rc = launch_att16v2_one<2, 1, 3, 64, true>(grid, qk16, qk_plane, vt16, vt_plane, nullptr, slopes, T, H, Tp, nullptr, ctx16, plane, s, rag); break;
        default: rc = launch_att16v2_one<4, 1, 3, 64, true>(grid, qk16, qk_plane, vt16, vt_plane, nullptr, slopes, T, H, Tp, nullptr, ctx16, plane, s, rag); break;
    }
    if (rc) return rc;
    PGMI_HIP(hipGetLastError());
    return PGMI_OK;
}

}  // namespace pgmi
