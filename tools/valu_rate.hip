// Issue cost of the vector instructions a GEMM epilogue is made of, in shader cycles per wave instruction per SIMD, with ONE and
// with TWO (and four) waves per SIMD issuing the same independent stream (s_memtime brackets; 4 blocks on a few CUs so that
// the clock is not power-limited).  Decides between epilogue forms: is v_pk_*_f32 one or two issue slots, what does a
// transcendental cost, do two co-resident waves issue faster than one.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate.bin && tools/valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float float2_t __attribute__((ext_vector_type(2)));

enum { OP_FMA, OP_PK_FMA32, OP_PK_MUL32, OP_PK_ADD32, OP_EXP32, OP_RCP32, OP_EXP16, OP_RCP16, OP_PK_MUL16, OP_PK_FMA16,
       OP_CVT_PK, OP_MIX32, OP_MIXLO, OP_MIN32, OP_SWAP16, OP_DOT2, OP_EXP16_SDWA, OP_MFMA_FMA, OP_EXP_PKFMA1, OP_EXP_PKFMA2, OP_EXP_PKFMA4, OP_EXP_FMA4, OP_COUNT };
static const char* const kNames[OP_COUNT] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_exp_f32", "v_rcp_f32",
    "v_exp_f16", "v_rcp_f16", "v_pk_mul_f16", "v_pk_fma_f16", "v_cvt_pk_f16_f32", "v_fma_mix_f32", "v_fma_mixlo_f16", "v_min_f32",
    "v_permlane16_swap_b32", "v_dot2_f32_f16", "v_exp_f16_sdwa(hi)", "mfma16x16x32 + 4 v_fma_f32",
    "v_exp_f32 + 1 v_pk_fma_f32 (interleaved; per instruction)", "v_exp_f32 + 2 v_pk_fma_f32", "v_exp_f32 + 4 v_pk_fma_f32", "v_exp_f32 + 4 v_fma_f32"};

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void rate_kernel(float* out, long long* cyc, int iters, float seed) {
    float r[8];
    float2_t p[8];
    typedef float float4_t __attribute__((ext_vector_type(4)));
    typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
    float4_t acc = {0.f, 0.f, 0.f, 0.f};
    half8_t ha, hb;
    for (int i = 0; i < 8; ++i) {
        r[i] = seed + 0.001f * (threadIdx.x + i);
        p[i] = (float2_t){r[i], r[i] * 0.5f};
        ha[i] = (_Float16)r[i]; hb[i] = (_Float16)(0.5f * r[i]);
    }
    const float a = 1.0001f, b = 0.0001f;
    const float2_t a2 = {a, a}, b2 = {b, b};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (OP == OP_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_PK_FMA32) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a2), "v"(b2));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_PK_MUL32) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_PK_ADD32) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(b2));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_EXP32) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_RCP32) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_EXP16) {
#define X(i) asm volatile("v_exp_f16 %0, %0" : "+v"(r[i]));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_EXP16_SDWA) {
#define X(i) asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(r[i]));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_RCP16) {
#define X(i) asm volatile("v_rcp_f16 %0, %0" : "+v"(r[i]));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_PK_MUL16) {
#define X(i) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(r[i]) : "v"(a));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_PK_FMA16) {
#define X(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_CVT_PK) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_MIX32) {
#define X(i) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(r[i]) : "v"(a), "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_MIXLO) {
#define X(i) asm volatile("v_fma_mixlo_f16 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(r[i]) : "v"(a), "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_MIN32) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_SWAP16) {
#define X(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[i]), "+v"(p[i][0]));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_DOT2) {
#define X(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
                REP8(X)
#undef X
            } else if constexpr (OP == OP_EXP_PKFMA1 || OP == OP_EXP_PKFMA2 || OP == OP_EXP_PKFMA4 || OP == OP_EXP_FMA4) {
                // r04: does the transcendental unit run BESIDE the main vector ALU?  One v_exp_f32 followed by k independent (packed) fmas,
                // 8 x per u: if the per-instruction cost falls below the weighted mean of the two stand-alone costs, it does -- and an
                // exponential evaluated as a polynomial on the main ALU could run beside the transcendental ones (attention_v5.hip).
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i])); \
             asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a2), "v"(b2)); \
             if constexpr (OP == OP_EXP_PKFMA2 || OP == OP_EXP_PKFMA4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 1) & 7]) : "v"(a2), "v"(b2)); \
             if constexpr (OP == OP_EXP_PKFMA4) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 2) & 7]) : "v"(a2), "v"(b2)); \
                                                  asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 3) & 7]) : "v"(a2), "v"(b2)); }
#define Y(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i])); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[i][0]) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[i][1]) : "v"(a), "v"(b)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 1) & 7][0]) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 1) & 7][1]) : "v"(a), "v"(b));
                if constexpr (OP == OP_EXP_FMA4) { REP8(Y) } else { REP8(X) }
#undef X
#undef Y
            } else if constexpr (OP == OP_MFMA_FMA) {
                // what a vector instruction costs BESIDE the matrix pipe: one MFMA + 4 independent fmas, 2 x per u
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc, 0, 0, 0);
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
                X(0) X(1) X(2) X(3)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, ha, acc, 0, 0, 0);
                X(4) X(5) X(6) X(7)
#undef X
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc[0] + acc[3];
    for (int i = 0; i < 8; ++i) s += r[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(float* out, long long* cyc, int threads) {
    const int iters = 2000, blocks = 8;                 // a few CUs only: no power throttling, s_memtime counts shader cycles
    rate_kernel<OP><<<blocks, threads>>>(out, cyc, 10, 1.0f);
    hipDeviceSynchronize();
    rate_kernel<OP><<<blocks, threads>>>(out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    const int nw = blocks * threads / 64;
    std::vector<long long> h(nw);
    hipMemcpy(h.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const int per_u = OP == OP_EXP_PKFMA1 ? 16 : OP == OP_EXP_PKFMA2 ? 24 : (OP == OP_EXP_PKFMA4 || OP == OP_EXP_FMA4) ? 40 : 8;   // instructions per u (x 8 u per iteration)
    const double per_wave = (double)h[nw / 2] / (iters * 8.0 * per_u);        // cycles per instruction of ONE wave's stream
    const double per_simd = per_wave / (threads / 256.0);                    // ... per instruction issued on the SIMD
    printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr_per_wave\": %.2f, \"cycles_per_instr_per_simd\": %.2f}\n",
           kNames[OP], threads / 256, per_wave, per_simd);
}

template <int OP>
void run_all(float* out, long long* cyc) {
    run<OP>(out, cyc, 256);
    run<OP>(out, cyc, 512);
    run<OP>(out, cyc, 1024);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 8 * 1024 * sizeof(float));
    hipMalloc(&cyc, 8 * 16 * sizeof(long long));
    run_all<OP_FMA>(out, cyc); run_all<OP_PK_FMA32>(out, cyc); run_all<OP_PK_MUL32>(out, cyc); run_all<OP_PK_ADD32>(out, cyc);
    run_all<OP_EXP32>(out, cyc); run_all<OP_RCP32>(out, cyc); run_all<OP_EXP16>(out, cyc); run_all<OP_EXP16_SDWA>(out, cyc);
    run_all<OP_RCP16>(out, cyc);
    run_all<OP_PK_MUL16>(out, cyc); run_all<OP_PK_FMA16>(out, cyc); run_all<OP_CVT_PK>(out, cyc); run_all<OP_MIX32>(out, cyc);
    run_all<OP_MIXLO>(out, cyc); run_all<OP_MIN32>(out, cyc); run_all<OP_SWAP16>(out, cyc); run_all<OP_DOT2>(out, cyc);
    run_all<OP_MFMA_FMA>(out, cyc);
    run_all<OP_EXP_PKFMA1>(out, cyc); run_all<OP_EXP_PKFMA2>(out, cyc); run_all<OP_EXP_PKFMA4>(out, cyc); run_all<OP_EXP_FMA4>(out, cyc);
    return 0;
}
