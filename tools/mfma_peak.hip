// Measures the fp16 MFMA rate this device sustains on RANDOM operands with no memory traffic at all
// (the ceiling any GEMM here can reach at the clock the chip holds under MFMA load; cdna guide rule 28).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mfma_loop(const _Float16* src, float* out, int iters) {
    half8_t fa[8], fb[4];
    for (int i = 0; i < 8; ++i) fa[i] = *(const half8_t*)(src + ((threadIdx.x * 8 + i) * 8) % 4096);
    for (int i = 0; i < 4; ++i) fb[i] = *(const half8_t*)(src + ((threadIdx.x * 4 + i + 77) * 8) % 4096);
    float4_t acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float float16_t __attribute__((ext_vector_type(16)));

// the same 128 x 64 wave block as 4 x 2 tiles of v_mfma_f32_32x32x16_f16 (128 accumulator registers either way)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mfma_loop32(const _Float16* src, float* out, int iters) {
    half8_t fa[4], fb[2];
    for (int i = 0; i < 4; ++i) fa[i] = *(const half8_t*)(src + ((threadIdx.x * 4 + i) * 8) % 4096);
    for (int i = 0; i < 2; ++i) fb[i] = *(const half8_t*)(src + ((threadIdx.x * 2 + i + 77) * 8) % 4096);
    float16_t acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)                      // four 16-deep steps = the K extent of two 16x16x32 steps
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int WAVES>
void run32(const _Float16* d, float* o, int blocks_per_cu) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop32<WAVES><<<grid, WAVES * 64>>>(d, o, 100);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        mfma_loop32<WAVES><<<grid, WAVES * 64>>>(d, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)grid * WAVES * iters * 32 * 2.0 * 32 * 32 * 16;
        printf("32x32x16: waves/block %d blocks/CU %d: %.3f ms  %.0f TFLOP/s\n", WAVES, blocks_per_cu, ms, fl / ms / 1e9);
    }
}

template <int WAVES>
void run(const _Float16* d, float* o, int blocks_per_cu) {
    const int iters = 4000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<WAVES><<<grid, WAVES * 64>>>(d, o, 100);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        mfma_loop<WAVES><<<grid, WAVES * 64>>>(d, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)grid * WAVES * iters * 32 * 2.0 * 16 * 16 * 32;
        printf("waves/block %d blocks/CU %d: %.3f ms  %.0f TFLOP/s\n", WAVES, blocks_per_cu, ms, fl / ms / 1e9);
    }
}

int main() {
    _Float16* h = (_Float16*)malloc(4096 * 2);
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    _Float16* d; float* o;
    hipMalloc(&d, 4096 * 2); hipMalloc(&o, 256 * 8 * 512 * 4);
    hipMemcpy(d, h, 4096 * 2, hipMemcpyHostToDevice);
    run<4>(d, o, 1);   // 1 wave per SIMD
    run<8>(d, o, 1);   // 2 waves per SIMD
    run<4>(d, o, 2);
    run32<4>(d, o, 1);
    run32<8>(d, o, 1);
    return 0;
}
