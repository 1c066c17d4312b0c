// Single-op and timing entries of include/pgmi.h (numerics tests against torch ops; interleaved A/B of launch parameters).
#include "model.h"


extern "C" {

// ---- single-op entry points for the numerics tests -------------------------------------------
int pgmi_op_layernorm(int device, const float* x, const float* w, const float* b, int rows, int D, float eps, float* y) {
    if (!x || !w || !b || !y || rows <= 0 || D <= 0 || D % 4) { set_error("bad argument"); return PGMI_EINVAL; }
    if (pgmi_device_count() <= 0) { set_error("no HIP device visible"); return PGMI_ENODEV; }
    PGMI_HIP(hipSetDevice(device));
    std::vector<void*> pool;
    float *dx, *dw, *db, *dy;
    int rc = 0;
    if ((rc = dev_upload(pool, &dx, x, (size_t)rows * D)) || (rc = dev_upload(pool, &dw, w, (size_t)D)) ||
        (rc = dev_upload(pool, &db, b, (size_t)D)) || (rc = dev_alloc(pool, &dy, (size_t)rows * D))) {
        for (void* p : pool) hipFree(p);
        return rc;
    }
    launch_layernorm(dx, dw, db, rows, D, eps, dy, nullptr);
    hipError_t e = hipMemcpy(y, dy, (size_t)rows * D * 4, hipMemcpyDeviceToHost);
    for (void* p : pool) hipFree(p);
    if (e != hipSuccess) { set_error("layernorm op failed: %s", hipGetErrorString(e)); return PGMI_EHIP; }
    return PGMI_OK;
}

int pgmi_op_gemm(int device, int precision, const float* A, const float* W, const float* bias, const float* residual,
                 int M, int N, int K, int epilogue, float* C) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) { set_error("bad argument"); return PGMI_EINVAL; }
    if (precision != PGMI_PREC_FP32 && precision != PGMI_PREC_F16X3 && precision != PGMI_PREC_BF16) { set_error("unknown precision %d", precision); return PGMI_EINVAL; }
    if (pgmi_device_count() <= 0) { set_error("no HIP device visible"); return PGMI_ENODEV; }
    PGMI_HIP(hipSetDevice(device));
    gemm_options_from_env();                             // a model-less entry of the tests: the hooks are read per call here
    std::vector<void*> pool;
    float *dA, *dW = nullptr, *dB = nullptr, *dR = nullptr, *dC;
    int rc = 0;
    if ((rc = dev_upload(pool, &dA, A, (size_t)M * K)) ||
        (precision == PGMI_PREC_FP32 && (rc = dev_upload(pool, &dW, W, (size_t)N * K))) ||
        (bias && (rc = dev_upload(pool, &dB, bias, (size_t)N))) ||
        (residual && (rc = dev_upload(pool, &dR, residual, (size_t)M * N))) ||
        (rc = dev_alloc(pool, &dC, (size_t)M * N))) {
        for (void* p : pool) hipFree(p);
        return rc;
    }
    // epilogue: EPI_* in the low byte; + 256 (f16x3 only, no residual, N % 32 == 0): run the SPLIT-PLANE output epilogue and return its
    // planes rebuilt as fp32 (tests: a row's bits through the half- and full-height items)
    const int epi = epilogue & 255;
    const bool split_planes = (epilogue & 256) != 0;
    if (split_planes && (precision == PGMI_PREC_FP32 || residual || (N % (precision == PGMI_PREC_F16X3 ? 32 : 4)))) {
        for (void* p : pool) hipFree(p);
        set_error("16-bit-plane GEMM op: f16x3 (N %% 32 == 0) or bf16 (N %% 4 == 0), no residual");
        return PGMI_EINVAL;
    }
    if (precision == PGMI_PREC_FP32) {
        rc = launch_gemm_f32(dA, dW, dB, dR, dC, M, N, K, epi, nullptr);
    } else {
        const bool bf = precision == PGMI_PREC_BF16;
        const int planes = bf ? 1 : 2;
        W16 w16;
        unsigned short* a16 = nullptr;
        rc = make_w16(pool, W, (size_t)N * K, (size_t)K, precision, nullptr, &w16);
        if (!rc) rc = dev_alloc(pool, &a16, (size_t)M * K * planes);
        if (!rc) {
            launch_split16(dA, (int64_t)M * K, 1.0f, bf ? 1 : 2, K, a16, nullptr);
            if (split_planes) {
                // the split-plane epilogue (the next GEMM's operand): run it, then rebuild fp32 = hi + lo 2^-11 from the K-interleaved planes
                unsigned short* c16 = nullptr;
                rc = dev_alloc(pool, &c16, (size_t)M * N * 2);
                if (!rc) rc = launch_gemm16(a16, (size_t)M * K, w16.p, w16.plane, dB, nullptr, nullptr, c16, (size_t)M * N, M, N, K, epi,
                                            w16.out_scale, planes, bf, env_int("PGMI_GEMM_VARIANT", 0), nullptr);
                if (!rc) {
                    std::vector<unsigned short> h((size_t)M * N * 2);
                    hipError_t e2 = hipMemcpy(h.data(), c16, h.size() * 2, hipMemcpyDeviceToHost);
                    if (e2 != hipSuccess) { set_error("gemm op failed: %s", hipGetErrorString(e2)); rc = PGMI_EHIP; }
                    for (size_t m = 0; m < (size_t)M && !rc && bf; ++m)           // bf16 plane, row-major
                        for (int n = 0; n < N; ++n) {
                            const unsigned int u = (unsigned int)h[m * N + n] << 16;
                            memcpy(&C[m * N + n], &u, 4);
                        }
                    for (size_t m = 0; m < (size_t)M && !rc && !bf; ++m)
                        for (int n = 0; n < N; ++n) {
                            const size_t o = ki_off(m, n, N);
                            _Float16 hi, lo;
                            memcpy(&hi, &h[o], 2); memcpy(&lo, &h[o + 32], 2);
                            C[m * N + n] = (float)hi + (float)lo * (1.0f / kLoScale);
                        }
                }
                for (void* p : pool) hipFree(p);
                return rc;
            }
            rc = launch_gemm16(a16, (size_t)M * K, w16.p, w16.plane, dB, dR, dC, nullptr, 0, M, N, K, epi,
                               w16.out_scale, planes, bf, env_int("PGMI_GEMM_VARIANT", 0), nullptr);
        }
    }
    hipError_t e = hipMemcpy(C, dC, (size_t)M * N * 4, hipMemcpyDeviceToHost);
    for (void* p : pool) hipFree(p);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("gemm op failed: %s", hipGetErrorString(e)); return PGMI_EHIP; }
    return PGMI_OK;
}

int pgmi_bench_gemm_ab(int device, int precision, int M, int N, int K, int epilogue, int split_out, const int* variants,
                       int n_variants, int rounds, int iters, double* ms_out) {
    if (M <= 0 || N <= 0 || K <= 0 || iters <= 0 || rounds <= 0 || n_variants <= 0 || !variants || !ms_out) { set_error("bad argument"); return PGMI_EINVAL; }
    if (pgmi_device_count() <= 0) { set_error("no HIP device visible"); return PGMI_ENODEV; }
    PGMI_HIP(hipSetDevice(device));
    gemm_options_from_env();
    std::vector<void*> pool;
    auto cleanup = [&]() { for (void* p : pool) hipFree(p); };
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
    unsigned int st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) * (1.0f / 8388608.0f)) - 1.0f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hW) v = rnd() * 0.03f;
    for (auto& v : hb) v = rnd();
    float *dA, *dW = nullptr, *dB, *dC = nullptr;
    unsigned short *a16 = nullptr, *c16 = nullptr;
    int rc = 0;
    if ((rc = dev_upload(pool, &dA, hA.data(), hA.size())) || (rc = dev_upload(pool, &dB, hb.data(), hb.size()))) { cleanup(); return rc; }
    hipEvent_t e0, e1;
    PGMI_HIP(hipEventCreate(&e0));
    PGMI_HIP(hipEventCreate(&e1));
    const bool f32 = precision == PGMI_PREC_FP32;
    const bool bf = precision == PGMI_PREC_BF16;
    const int planes = bf ? 1 : 2;
    W16 w16;
    if (f32) {
        if ((rc = dev_upload(pool, &dW, hW.data(), hW.size())) || (rc = dev_alloc(pool, &dC, (size_t)M * N))) { cleanup(); return rc; }
    } else {
        if ((rc = make_w16(pool, hW.data(), hW.size(), (size_t)K, precision, nullptr, &w16)) ||
            (rc = dev_alloc(pool, &a16, (size_t)M * K * planes))) { cleanup(); return rc; }
        launch_split16(dA, (int64_t)M * K, 1.0f, bf ? 1 : 2, K, a16, nullptr);
        if (split_out == 1) rc = dev_alloc(pool, &c16, (size_t)M * N * planes);
        else if (split_out != 3) rc = dev_alloc(pool, &dC, (size_t)M * N);
        if (rc) { cleanup(); return rc; }
    }
    // split_out 2: fp32 output with the in-place residual of the out-projection / FC2 (x += ...); 3: the fused QKV epilogue
    // (attention operands; N = 3 D, sequences of 288 tokens when M allows)
    const bool fused_qkv = split_out == 3 && !f32 && !bf;
    const int Tq = (M % 288 == 0) ? 288 : M;
    unsigned short *qk16 = nullptr, *vt16 = nullptr;
    size_t qk_plane = 0, vt_plane = 0;
    if (fused_qkv) {
        if (N % 3 || (N / 3) % 64) { set_error("fused QKV bench needs N = 3 D, D %% 64 == 0"); cleanup(); return PGMI_EINVAL; }
        const size_t Tp = (size_t)(Tq + 31) / 32 * 32;
        qk_plane = (size_t)M * 2 * (N / 3);
        vt_plane = (size_t)(M / Tq) * (N / 3) * Tp;
        if ((rc = dev_alloc(pool, &qk16, qk_plane * 2)) || (rc = dev_alloc(pool, &vt16, vt_plane * 2))) { cleanup(); return rc; }
    }
    auto run = [&](int var) -> int {
        if (f32) return launch_gemm_f32(dA, dW, dB, split_out == 2 ? dC : nullptr, dC, M, N, K, epilogue, nullptr);
        if (fused_qkv)
            return launch_gemm16_qkv(a16, (size_t)M * K, w16.p, w16.plane, dB, M, N / 3, K, w16.out_scale, qk16, qk_plane, vt16, vt_plane,
                                     nullptr, nullptr, 0, Tq, N / 3 / kHeadDim, var, nullptr);
        const bool planes_out = split_out == 1;
        return launch_gemm16(a16, (size_t)M * K, w16.p, w16.plane, dB, split_out == 2 ? dC : nullptr, planes_out ? nullptr : dC,
                             planes_out ? c16 : nullptr, (size_t)M * N, M, N, K, epilogue, w16.out_scale, planes, bf, var, nullptr);
    };
    std::vector<std::vector<double>> samples(n_variants);
    for (int v = 0; v < n_variants && !rc; ++v) rc = run(variants[v] >= 0 ? variants[v] : env_int("PGMI_GEMM_VARIANT", 0));   // warm-up
    for (int r = 0; r < rounds && !rc; ++r)
        for (int v = 0; v < n_variants && !rc; ++v) {              // interleaved rounds: variants see the same clocks / temperature
            const int var = variants[v] >= 0 ? variants[v] : env_int("PGMI_GEMM_VARIANT", 0);
            hipEventRecord(e0, nullptr);
            for (int i = 0; i < iters && !rc; ++i) rc = run(var);
            hipEventRecord(e1, nullptr);
            hipError_t e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e != hipSuccess) { set_error("bench failed: %s", hipGetErrorString(e)); rc = PGMI_EHIP; }
            samples[v].push_back(ms / iters);
        }
    if (!rc)
        for (int v = 0; v < n_variants; ++v) {
            std::sort(samples[v].begin(), samples[v].end());
            ms_out[v] = samples[v][samples[v].size() / 2];         // median over the rounds
        }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    cleanup();
    return rc;
}

int pgmi_bench_gemm(int device, int precision, int M, int N, int K, int epilogue, int split_out, int variant,
                    int iters, double* ms_per_launch) {
    return pgmi_bench_gemm_ab(device, precision, M, N, K, epilogue, split_out, &variant, 1, 1, iters, ms_per_launch);
}

int pgmi_op_attention(int device, int precision, const float* qkv, const int32_t* kv_len, int B, int T, int H,
                      int rotary, float* ctx) {
    if (!qkv || !ctx || B <= 0 || T <= 0 || H <= 0) { set_error("bad argument"); return PGMI_EINVAL; }
    if (pgmi_device_count() <= 0) { set_error("no HIP device visible"); return PGMI_ENODEV; }
    PGMI_HIP(hipSetDevice(device));
    std::vector<void*> pool;
    float *dq, *dc;
    int32_t* dl = nullptr;
    const size_t D = (size_t)H * kHeadDim;
    int rc = 0;
    if ((rc = dev_upload(pool, &dq, qkv, (size_t)B * T * 3 * D)) || (kv_len && (rc = dev_upload(pool, &dl, kv_len, (size_t)B))) ||
        (rc = dev_alloc(pool, &dc, (size_t)B * T * D))) {
        for (void* p : pool) hipFree(p);
        return rc;
    }
    pgmi_model tmp;
    tmp.cfg.arch = PGMI_ARCH_ESM2;
    if (rotary) rc = ensure_rotary(&tmp, T);
    if (!rc && precision == PGMI_PREC_F16X3) {
        const size_t Tp = (size_t)(T + 31) / 32 * 32;
        unsigned short *qk = nullptr, *vt = nullptr;
        rc = dev_alloc(pool, &qk, (size_t)B * T * 2 * D * 2);
        if (!rc) rc = dev_alloc(pool, &vt, (size_t)B * Tp * D * 2);
        if (!rc) rc = launch_attention_f16x3_v2(dq, dl, tmp.rot_cos, tmp.rot_sin, rotary, B, T, H, qk, (size_t)B * T * 2 * D,
                                                vt, (size_t)B * Tp * D, dc, nullptr, 0, 0, nullptr);
    } else if (!rc) {
        if (rotary) launch_rotary(dq, tmp.rot_cos, tmp.rot_sin, B * T, T, H, nullptr);
        rc = launch_attention_f32(dq, dl, B, T, H, dc, nullptr, 0, 0, nullptr);
    }
    hipDeviceSynchronize();
    for (void* p : tmp.allocs) hipFree(p);
    hipError_t e = hipMemcpy(ctx, dc, (size_t)B * T * D * 4, hipMemcpyDeviceToHost);
    for (void* p : pool) hipFree(p);
    if (rc) return rc;
    if (e != hipSuccess) { set_error("attention op failed: %s", hipGetErrorString(e)); return PGMI_EHIP; }
    return PGMI_OK;
}

}  // extern "C"
