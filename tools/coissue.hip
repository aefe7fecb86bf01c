// Can vector instructions run in the shadow of MFMAs on one SIMD -- from the SAME wave (interleaved stream) and from a SECOND wave
// (one wave issues only MFMAs, its SIMD neighbour only vector work)?  Decides whether the attention kernels (softmax beside the
// QK / PV products) can be sped up by software pipelining or role-split waves, or whether matrix and vector time simply add.
// 8 blocks on a few CUs (no power throttling); s_memtime brackets; cycles per loop iteration.
//   hipcc --offload-arch=gfx950 -O3 tools/coissue.hip -o tools/coissue.bin && tools/coissue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

// MODE 0: per iteration 4 x mfma_32x32x16 (4 independent accumulators) each followed by NV vector instructions
// MODE 1: per iteration 8 x mfma_16x16x32 (same matrix-pipe time) each followed by NV / 2 vector instructions
// VOP 0: v_fma_f32, 1: v_exp_f32, 2: v_cvt_pk_f16_f32, 3: v_max3_f32
// ROLE 0: every wave runs the mixed stream; ROLE 1: waves 0-3 of the block only the MFMAs, waves 4-7 only the vector instructions
template <int MODE, int NV, int VOP, int ROLE>
__global__ void k(float* out, long long* cyc, int iters) {
    float r[16];
    half8_t ha, hb;
    for (int i = 0; i < 16; ++i) r[i] = 1.0f + 0.001f * (threadIdx.x + i);
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)r[i]; hb[i] = (_Float16)(0.5f * r[i]); }
    float16_t A[4];
    float4_t B[8];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) A[j][i] = 0.f;
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) B[j][i] = 0.f;
    const float a = 1.0001f, b = 0.0001f;
    const int wave = threadIdx.x >> 6;
    const bool do_m = ROLE == 0 || wave < 4, do_v = ROLE == 0 || wave >= 4;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (do_m) {
                if (MODE == 0) A[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, A[j], 0, 0, 0);
                else {
                    B[2 * j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, B[2 * j], 0, 0, 0);
                }
            }
            if (do_v) {
#pragma unroll
                for (int v = 0; v < (MODE == 0 ? NV : NV / 2); ++v) {
                    float& x = r[(j * 4 + v) & 15];
                    if (VOP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
                    else if (VOP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                    else if (VOP == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(a));
                    else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
                }
            }
            if (MODE == 1) {
                if (do_m) B[2 * j + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, ha, B[2 * j + 1], 0, 0, 0);
                if (do_v) {
#pragma unroll
                    for (int v = 0; v < NV / 2; ++v) {
                        float& x = r[(j * 4 + v + 8) & 15];
                        if (VOP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
                        else if (VOP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                        else if (VOP == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(a));
                        else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
                    }
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += A[j][0] + A[j][15];
    for (int j = 0; j < 8; ++j) s += B[j][0];
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

static const char* const kV[4] = {"v_fma_f32", "v_exp_f32", "v_cvt_pk_f16_f32", "v_max3_f32"};

template <int MODE, int NV, int VOP, int ROLE>
void run(float* out, long long* cyc, int threads) {
    const int iters = 2000, blocks = 8;
    k<MODE, NV, VOP, ROLE><<<blocks, threads>>>(out, cyc, 10);
    hipDeviceSynchronize();
    k<MODE, NV, VOP, ROLE><<<blocks, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    const int wpb = threads / 64, nw = blocks * wpb;
    std::vector<long long> h(nw);
    hipMemcpy(h.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<long long> m, v;
    for (int i = 0; i < nw; ++i) ((ROLE == 1 && (i % wpb) >= 4) ? v : m).push_back(h[i]);
    std::sort(m.begin(), m.end());
    std::sort(v.begin(), v.end());
    printf("{\"mfma\": \"%s\", \"vector_per_mfma32\": %d, \"vop\": \"%s\", \"role_split\": %d, \"waves_per_simd\": %d, "
           "\"cycles_per_4mfma32_group\": %.1f", MODE == 0 ? "32x32x16" : "2 x 16x16x32", NV, kV[VOP], ROLE, threads / 256,
           (double)m[m.size() / 2] / iters);
    if (ROLE == 1) printf(", \"vector_wave_cycles_per_group\": %.1f", (double)v[v.size() / 2] / iters);
    printf("}\n");
}

template <int MODE, int VOP>
void sweep(float* out, long long* cyc) {
    run<MODE, 0, VOP, 0>(out, cyc, 256);
    run<MODE, 4, VOP, 0>(out, cyc, 256);
    run<MODE, 8, VOP, 0>(out, cyc, 256);
    run<MODE, 12, VOP, 0>(out, cyc, 256);
    run<MODE, 16, VOP, 0>(out, cyc, 256);
    run<MODE, 8, VOP, 0>(out, cyc, 512);
    run<MODE, 8, VOP, 0>(out, cyc, 1024);
    run<MODE, 8, VOP, 1>(out, cyc, 512);     // wave w: MFMAs only, wave w + 4 (same SIMD): 8 vector instructions per MFMA slot
    run<MODE, 16, VOP, 1>(out, cyc, 512);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 8 * 1024 * sizeof(float));
    hipMalloc(&cyc, 8 * 16 * sizeof(long long));
    sweep<0, 0>(out, cyc); sweep<0, 1>(out, cyc); sweep<0, 2>(out, cyc); sweep<0, 3>(out, cyc);
    sweep<1, 0>(out, cyc); sweep<1, 1>(out, cyc);
    return 0;
}
