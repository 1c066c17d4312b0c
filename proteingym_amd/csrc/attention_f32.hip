// Fused multi-head self-attention, fp32 operands on the matrix cores (parity-gated mode).
//
// Replaces  S = q k^T ; softmax_fp32(S) ; O = P v  of
// /root/reference/proteingym/baselines/esm/esm/multihead_attention.py:357-387 (ESM-1v reaches
// the same math through F.multi_head_attention_forward, :196-230).  q arrives pre-scaled by
// head_dim^-0.5 (folded into the packed q_proj weights; 1/8 is exact).  Non-causal; keys
// >= kv_len[b] are masked with -inf exactly like key_padding_mask (:370-376).  The [B*H,T,T]
// score tensor is never materialised.
//
// One wave owns 32 query rows of one (sequence, head); up to 4 waves (same head) share the
// K/V tiles of 32 keys staged in LDS.  Everything is arranged so that no cross-lane data
// movement is needed except one lane<->lane+32 exchange per tile for the row max:
//   S^T = K Q^T  (A = K tile from LDS, B = Q held in 32 VGPRs)  -> C layout puts query
//        r = lane&31 in the lane and 16 of the tile's 32 keys in the lane's registers;
//   online softmax per lane (row stats live in the lane that owns the query);
//   O^T = V^T P^T (A = V^T read from LDS, B = P straight from the S^T accumulator
//        registers: register v of lane (r,kh) is P[r][key (v&3)+8(v>>2)+4kh], which is
//        exactly the B-operand layout for the k pair (key, key+4)) -> query again in the
//        lane, so the rescale by exp(m_old-m_new) and the final 1/l are per-lane scalars.
// MFMA: v_mfma_f32_32x32x2_f32, 64 per (32q x 32k) tile = 4096 cycles for 262144 FLOP = the
// fp32 matrix peak; the ~100 VALU ops of the softmax hide behind the second wave on the SIMD.
#include "common.h"

namespace pgmi {

constexpr int KT = 32;          // keys per tile

// DH = 64 (heads of <= 64 dims, zero-padded slots) or 128 (ESM2-15B: a head is two adjacent 64-lane slot groups of the
// projection; pretrained.py:387-394): the S^T contraction runs over DH / 8 fragments, O^T has DH / 32 column tiles.
template <int WPB, int OUT, int DH = 64>
__global__ __launch_bounds__(WPB * 64) void attention_f32_kernel(
    const float* __restrict__ qkv, const int32_t* __restrict__ kv_len, int T, int H,
    float* __restrict__ ctx, unsigned short* __restrict__ ctx16, size_t plane) {
    constexpr int NT = WPB * 64;
    constexpr int KS_STRIDE = DH + 4;         // K tile row stride (floats): conflict-free ds_read_b128
    constexpr int VS_STRIDE = DH;
    constexpr int NF4 = KT * DH / 4;          // float4 units per tensor per tile
    constexpr int NL = (NF4 + NT - 1) / NT;   // float4 loads per thread per tensor per tile
    constexpr int C4 = DH / 4;                // float4 units per key row
    constexpr int NG = DH / 8, ND = DH / 32;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][KT][KS_STRIDE] K, then [2][KT][VS_STRIDE] V
    float* Ks = lds;
    float* Vs = lds + 2 * KT * KS_STRIDE;

    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, kh = lane >> 5;
    const int D = H * DH;
    const size_t RS = (size_t)3 * D;
    const float* base = qkv + (size_t)b * T * RS + (size_t)h * DH;
    const int Tk = kv_len ? kv_len[b] : T;
    const int q0 = (blockIdx.x * WPB + wave) * 32;
    const bool active = q0 < T;

    // Q fragment: lane (r,kh) holds Q[q0+r][8g+4kh+e]
    f32x4 qf[NG];
    {
        const int qrow = min(q0 + r, T - 1);
        const float* qp = base + (size_t)qrow * RS + kh * 4;
#pragma unroll
        for (int g = 0; g < NG; ++g) qf[g] = *reinterpret_cast<const f32x4*>(qp + g * 8);
    }

    // staging of K/V tiles through registers
    f32x4 k_st[NL], v_st[NL];
    auto stage_load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int f = tid + NT * i;
            if (f < NF4) {
                const int key = kt * KT + f / C4, c4 = f % C4;
                if (key < T) {
                    const float* p = base + (size_t)key * RS + D + c4 * 4;
                    k_st[i] = *reinterpret_cast<const f32x4*>(p);
                    v_st[i] = *reinterpret_cast<const f32x4*>(p + D);
                } else {
                    k_st[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    v_st[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int f = tid + NT * i;
            if (f < NF4) {
                const int key = f / C4, c4 = f % C4;
                *reinterpret_cast<f32x4*>(Ks + buf * KT * KS_STRIDE + key * KS_STRIDE + c4 * 4) = k_st[i];
                *reinterpret_cast<f32x4*>(Vs + buf * KT * VS_STRIDE + key * VS_STRIDE + c4 * 4) = v_st[i];
            }
        }
    };

    const int nkt = (Tk + KT - 1) / KT;      // tiles holding at least one valid key
    stage_load(0);
    stage_store(0);
    __syncthreads();

    f32x16 o[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int v = 0; v < 16; ++v) o[dt][v] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) stage_load(kt + 1);
        if (active) {
            const float* Kb = Ks + cur * KT * KS_STRIDE + r * KS_STRIDE + kh * 4;
            const float* Vb = Vs + cur * KT * VS_STRIDE + r;
            f32x16 st;
#pragma unroll
            for (int v = 0; v < 16; ++v) st[v] = 0.f;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(Kb + g * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[g][e], st, 0, 0, 0);
            }
            if (kt * KT + KT > Tk) {          // wave-uniform: tile straddles the valid-key limit
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int key = kt * KT + (v & 3) + 8 * (v >> 2) + 4 * kh;
                    if (key >= Tk) st[v] = -INFINITY;
                }
            }
            float mloc = st[0];
#pragma unroll
            for (int v = 1; v < 16; ++v) mloc = fmaxf(mloc, st[v]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
            const float m_new = fmaxf(m_run, mloc);
            const float alpha = expf(m_run - m_new);
            float psum = 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                st[v] = expf(st[v] - m_new);
                psum += st[v];
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int v = 0; v < 16; ++v) o[dt][v] *= alpha;
            // O^T += V^T P^T : k pair for register v is (key_v, key_v + 4), key_v = (v&3)+8(v>>2)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int key = (v & 3) + 8 * (v >> 2);
                const float* vp = Vb + (key + 4 * kh) * VS_STRIDE;
#pragma unroll
                for (int dt = 0; dt < ND; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[dt * 32], st[v], o[dt], 0, 0, 0);
            }
        }
        if (more) stage_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    if (active) {
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        if (q0 + r < T) {
            const float inv = 1.0f / l_tot;
            const size_t off = ((size_t)b * T + q0 + r) * D + (size_t)h * DH + 4 * kh;
#pragma unroll
            for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 val;
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = o[dt][4 * g + e] * inv;
                    const size_t oo = off + dt * 32 + 8 * g;
                    if constexpr (OUT == 0) {
                        *reinterpret_cast<f32x4*>(ctx + oo) = val;
                    } else if constexpr (OUT == 1) {      // fp16 hi/lo planes for the f16x3 out-projection
                        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                        h4 hi, lo;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            _Float16 a, b2;
                            split_act(val[e], a, b2);
                            hi[e] = a;
                            lo[e] = b2;
                        }
                        // K-interleaved GEMM operand (common.h ki_off): column h*DH + dt*32 + 8g + 4kh of a row of D
                        unsigned short* dst = ctx16 + ((size_t)b * T + q0 + r) * (size_t)(2 * D) + (size_t)(ND * h + dt) * 64 + 8 * g + 4 * kh;
                        *reinterpret_cast<h4*>(dst) = hi;
                        *reinterpret_cast<h4*>(dst + 32) = lo;
                    } else {                              // bf16
                        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                        unsigned short bb[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float ve = val[e];   // copy first: bit_cast of a vector-element lvalue reads element 0
                            unsigned int u = __builtin_bit_cast(unsigned int, ve);
                            u += 0x7fffu + ((u >> 16) & 1u);
                            bb[e] = (unsigned short)(u >> 16);
                        }
                        u32x2 pk;
                        pk[0] = bb[0] | ((unsigned)bb[1] << 16);
                        pk[1] = bb[2] | ((unsigned)bb[3] << 16);
                        *reinterpret_cast<u32x2*>(ctx16 + oo) = pk;
                    }
                }
        }
    }
}

template <int WPB, int OUT, int DH>
static int launch_att_one(dim3 grid, const float* qkv, const int32_t* kv_len, int T, int H, float* ctx, unsigned short* ctx16,
                          size_t plane, hipStream_t s) {
    constexpr size_t lds_bytes = (size_t)(2 * KT * (DH + 4) + 2 * KT * DH) * sizeof(float);      // 33 KB at DH 64, 66.6 KB at DH 128
    auto kfn = attention_f32_kernel<WPB, OUT, DH>;
    if (lds_bytes > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PGMI_EHIP; }
    }
    hipLaunchKernelGGL(kfn, grid, dim3(WPB * 64), lds_bytes, s, qkv, kv_len, T, H, ctx, ctx16, plane);
    return PGMI_OK;
}

template <int OUT, int DH>
static int launch_att_mode(int wpb, dim3 grid, const float* qkv, const int32_t* kv_len, int T, int H,
                           float* ctx, unsigned short* ctx16, size_t plane, hipStream_t s) {
    switch (wpb) {
        case 1: return launch_att_one<1, OUT, DH>(grid, qkv, kv_len, T, H, ctx, ctx16, plane, s);
        case 2: return launch_att_one<2, OUT, DH>(grid, qkv, kv_len, T, H, ctx, ctx16, plane, s);
        case 3: return launch_att_one<3, OUT, DH>(grid, qkv, kv_len, T, H, ctx, ctx16, plane, s);
        default: return launch_att_one<4, OUT, DH>(grid, qkv, kv_len, T, H, ctx, ctx16, plane, s);
    }
}

int launch_attention_f32(const float* qkv, const int32_t* kv_len, int B, int T, int H, float* ctx,
                         unsigned short* ctx16, size_t plane, int out_mode, hipStream_t s, int head_dim) {
    if (B <= 0 || T <= 0 || H <= 0 || (head_dim != 64 && head_dim != 128) || out_mode < 0 || out_mode > 2) {
        set_error("attention_f32: bad arguments B=%d T=%d H=%d head_dim=%d out=%d", B, T, H, head_dim, out_mode);
        return PGMI_EINVAL;
    }
    const int n32 = (T + 31) / 32;
    const int nblk = (n32 + 3) / 4;
    const int wpb = (n32 + nblk - 1) / nblk;
    const dim3 grid(nblk, H, B);
    int rc;
    if (head_dim == 128) {
        if (out_mode == 0) rc = launch_att_mode<0, 128>(wpb, grid, qkv, kv_len, T, H, ctx, ctx16, plane, s);
        else if (out_mode == 1) rc = launch_att_mode<1, 128>(wpb, grid, qkv, kv_len, T, H, ctx, ctx16, plane, s);
        else rc = launch_att_mode<2, 128>(wpb, grid, qkv, kv_len, T, H, ctx, ctx16, plane, s);
    } else {
        if (out_mode == 0) rc = launch_att_mode<0, 64>(wpb, grid, qkv, kv_len, T, H, ctx, ctx16, plane, s);
        else if (out_mode == 1) rc = launch_att_mode<1, 64>(wpb, grid, qkv, kv_len, T, H, ctx, ctx16, plane, s);
        else rc = launch_att_mode<2, 64>(wpb, grid, qkv, kv_len, T, H, ctx, ctx16, plane, s);
    }
    if (rc) return rc;
    PGMI_HIP(hipGetLastError());
    return PGMI_OK;
}

}  // namespace pgmi
