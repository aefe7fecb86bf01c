// Round 6, item 4 (in_proj + attention fused per (image, head)): what the GEMM phase of such a kernel could sustain.
//
// One workgroup of the fused design computes Q | K | V of ONE head of ONE image: x_b [197, 768] . W_h^T [768, 192].  The tile is
// 197 rows (not a multiple of anything the MFMA shapes like) by 192 columns, against the 256 x 256 tiles of gemm_f16_v5.hip.  This probe
// runs the K loop of both geometries with the operands ALREADY IN LDS (the layout of the product kernel: 128-byte rows, 16-byte chunks
// XOR-swizzled by the row; fragment reads = ds_read_b128; one barrier per 64-wide K-tile; no global traffic at all), i.e. the rate the
// matrix pipes reach when LDS fragment reads are the only thing feeding them -- an upper bound for either kernel's K loop:
//
//   G0  256 x 256, 8 waves as 2 x 4, wave block 128 x 64 = 8 x 4 tiles of 16x16x32: 12 fragment reads per 32 MFMAs   (today's in_proj)
//   G1  208 x 192, 8 waves as 2 x 4, wave blocks (112 | 96) x 48 = (7 | 6) x 3 tiles: 10 (9) reads per 21 (18) MFMAs (197 rows in 13 x 16)
//   G2  224 x 192, 8 waves as 4 x 2, wave blocks (64,64,64,32) x 96 of 32x32x16: 5 (4) reads per 6 (3) MFMAs        (197 rows in 7 x 32)
//
// "useful" TFLOP/s counts 197 rows for G1 / G2 (the padding rows are computed and thrown away).
//   hipcc --offload-arch=gfx950 -O3 tools/fused_qkv_probe.hip -o tools/fused_qkv_probe.bin && tools/fused_qkv_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }   // byte offset in a [rows][64] fp16 tile

// 16x16x32 geometries.  BM_T / BN_T: 16-row / 16-column tiles of the workgroup; waves 2 (M) x 4 (N); wave wm takes the first ceil or the
// remaining floor half of the row tiles.
template <int BM_T, int BN_T>
__global__ __launch_bounds__(512) void kloop16(const half_t* src, float* out, int ktiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = BM_T * 16, BN = BN_T * 16;
    constexpr int TM_HI = (BM_T + 1) / 2, TN = BN_T / 4;
    char* sA = smem;
    char* sB = smem + BM * 128;
    for (int i = threadIdx.x; i < (BM + BN) * 8; i += 512) *(half8_t*)(smem + i * 16) = *(const half8_t*)(src + (i * 8) % 8192);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3;
    const int tm0 = wm * TM_HI, tm_n = wm == 0 ? TM_HI : BM_T - TM_HI;
    const int fr = lane & 15, fk = lane >> 4;
    float4_t acc[TM_HI][TN];
#pragma unroll
    for (int i = 0; i < TM_HI; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (float4_t){0, 0, 0, 0};
    for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8_t fa[TM_HI], fb[TN];
#pragma unroll
            for (int i = 0; i < TM_HI; ++i)
                if (i < tm_n) fa[i] = *(const half8_t*)(sA + swz((tm0 + i) * 16 + fr, ks * 4 + fk));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const half8_t*)(sB + swz((wn * TN + j) * 16 + fr, ks * 4 + fk));
#pragma unroll
            for (int i = 0; i < TM_HI; ++i)
                if (i < tm_n) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();                                   // the K-tile boundary of the double-buffered loop
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < TM_HI; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// 32x32x16 geometry G2: 7 row tiles of 32 over 4 wave rows (2, 2, 2, 1), 6 column tiles of 32 over 2 wave columns (3 each).
__global__ __launch_bounds__(512) void kloop32(const half_t* src, float* out, int ktiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 224, BN = 192;
    char* sA = smem;
    char* sB = smem + BM * 128;
    for (int i = threadIdx.x; i < (BM + BN) * 8; i += 512) *(half8_t*)(smem + i * 16) = *(const half8_t*)(src + (i * 8) % 8192);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int tm0 = wm * 2, tm_n = wm == 3 ? 1 : 2;
    const int fr = lane & 31, fk = lane >> 5;
    float16_t acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                   // 16-deep steps: a lane reads 8 halves = one 16-byte chunk
            half8_t fa[2], fb[3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (i < tm_n) fa[i] = *(const half8_t*)(sA + swz((tm0 + i) * 32 + fr, ks * 2 + fk));
#pragma unroll
            for (int j = 0; j < 3; ++j) fb[j] = *(const half8_t*)(sB + swz((wn * 3 + j) * 32 + fr, ks * 2 + fk));
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (i < tm_n) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) s += acc[i][j][0] + acc[i][j][15];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, size_t lds, int rows_real, int cols, const half_t* d, float* o) {
    const int grid = 256, ktiles = 12 * 400;               // 400 tiles' worth of a K = 768 loop per workgroup, one workgroup per CU
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<grid, 512, lds>>>(d, o, 24);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        kern<<<grid, 512, lds>>>(d, o, ktiles);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double useful = (double)grid * ktiles * 2.0 * rows_real * cols * 64;
        printf("%-44s %8.3f ms  useful %6.0f TFLOP/s  = %5.2f us per (K = 768) tile per CU\n", name, ms, useful / ms / 1e9, ms * 1e3 / 400);
    }
}

int main() {
    half_t* h = (half_t*)malloc(8192 * 2);
    srand(1);
    for (int i = 0; i < 8192; ++i) h[i] = (half_t)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    half_t* d; float* o;
    hipMalloc(&d, 8192 * 2); hipMalloc(&o, 256 * 512 * 4);
    hipMemcpy(d, h, 8192 * 2, hipMemcpyHostToDevice);
    run("G0 256 x 256, 16x16x32 (today's tile)", kloop16<16, 16>, (256 + 256) * 128, 256, 256, d, o);
    run("G1 208 x 192, 16x16x32 (197 rows in 13 x 16)", kloop16<13, 12>, (208 + 192) * 128, 197, 192, d, o);
    run("G2 224 x 192, 32x32x16 (197 rows in 7 x 32)", kloop32, (224 + 192) * 128, 197, 192, d, o);
    return 0;
}
