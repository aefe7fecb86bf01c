// Shared fp16 GEMM epilogue: lane owns row m and four consecutive columns n..n+3 (C^T accumulator layout).
#pragma once
#include "common.h"

template <int EPI>
__device__ __forceinline__ void epilogue_store(const GemmArgs& a, int m, int n, float4_t acc) {
    if (m >= a.M || n >= a.N) return;
    half_t* C = (half_t*)a.C;
    long crow = m;
    const half_t* posrow = nullptr;
    if (EPI == EPI_PATCH) {
        int b = m / a.rows_in, p = m - b * a.rows_in;
        crow = (long)b * a.rows_out + 1 + p;
        posrow = (const half_t*)a.pos + (long)(1 + p) * a.N;
    }
    const bool vec = (n + 3 < a.N) && ((a.ldc & 3) == 0) && (EPI != EPI_BIAS_RES || (a.ldres & 3) == 0);
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_QGELU || EPI == EPI_BIAS_RES) {
        const half_t* bp = (const half_t*)a.bias + n;
        if (vec) {
            half4_t b4 = *(const half4_t*)bp;
#pragma unroll
            for (int r = 0; r < 4; ++r) bias[r] = (float)b4[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < a.N) bias[r] = (float)bp[r];
        }
    }
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float x = acc[r];
        if (EPI == EPI_BIAS_QGELU && a.gelu_mode) {          // the one-rounding form (common.h quick_gelu_f32x2), as the 256-row kernels
            const float xb = x + bias[r];
            x = (float)(half_t)(xb * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xb * -2.4554669595930157f)));
        } else {
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_QGELU || EPI == EPI_BIAS_RES) x = (float)(half_t)(x + bias[r]);
            else x = (float)(half_t)x;
            if (EPI == EPI_BIAS_QGELU) x = quick_gelu_h(x);
        }
        if (EPI == EPI_SCALE) x = x * a.scale;
        v[r] = x;
    }
    half_t* cp = C + crow * a.ldc + n;
    if (vec) {
        if (EPI == EPI_BIAS_RES) {
            half4_t r4 = *(const half4_t*)((const half_t*)a.res + (long)m * a.ldres + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)r4[r];
        }
        if (EPI == EPI_PATCH) {
            half4_t p4 = *(const half4_t*)(posrow + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)p4[r];
        }
        half4_t o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
        *(half4_t*)cp = o;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (n + r < a.N) {
                float x = v[r];
                if (EPI == EPI_BIAS_RES) x += (float)((const half_t*)a.res)[(long)m * a.ldres + n + r];
                if (EPI == EPI_PATCH) x += (float)posrow[n + r];
                cp[r] = (half_t)x;
            }
        }
    }
}

