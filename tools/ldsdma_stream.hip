// Load-path ceiling for the GEMM: every workgroup (512 threads, 1 per CU) streams 64 KiB "K-tiles" into LDS with
// global_load_lds_dwordx4 and does nothing else.  Compares the GEMM's access shape (256+256 rows x 128 B at a row
// stride) with fully contiguous 64 KiB chunks, L2-resident vs streaming footprints, drained vs counted waits.
//   hipcc --offload-arch=gfx950 -O3 tools/ldsdma_stream.hip -o tools/ldsdma_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// MODE 0: strided rows (row r of the tile at base + r*stride), MODE 1: contiguous chunk.  DEPTH: tiles in flight (1 or 2)
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void stream(const char* buf, size_t footprint, long stride, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t tiles = footprint / 65536;
    size_t t = (size_t)blockIdx.x * 977 % tiles;
    auto issue = [&](int slot, size_t tile) {
        const char* base = buf + tile * 65536;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ins = wave * 8 + j;                  // 64 instructions of 1 KiB per tile
            const char* src;
            if (MODE == 0) {
                const int row = ins * 8 + (lane >> 3);     // 512 rows x 128 B
                src = buf + ((tile * 512 + row) * (size_t)stride) % (footprint - 128) / 128 * 128 + (lane & 7) * 16;
            } else {
                src = base + ins * 1024 + lane * 16;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + slot * 65536 + ins * 1024), 16, 0, 0);
        }
    };
    if (DEPTH == 2) { issue(0, t); t = (t + 256) % tiles; }
    for (int i = 0; i < iters; ++i) {
        issue(DEPTH == 2 ? ((i + 1) & 1) : 0, t);
        t = (t + 256) % tiles;
        if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = ((int*)smem)[lane];
}

template <int MODE, int DEPTH>
void run(const char* d, size_t footprint, long stride, int* sink, const char* name) {
    const int iters = 400;
    hipFuncSetAttribute((const void*)stream<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    stream<MODE, DEPTH><<<256, 512, 131072>>>(d, footprint, stride, 20, sink);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        stream<MODE, DEPTH><<<256, 512, 131072>>>(d, footprint, stride, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = 256.0 * iters * 65536;
    printf("%-44s footprint %6zu MiB: %.3f ms, %.2f us per 64 KiB tile, %.1f GB/s per CU, %.2f TB/s chip\n", name,
           footprint >> 20, best, best * 1e3 / iters, bytes / 256 / best / 1e6, bytes / best / 1e9);
}

int main() {
    const size_t big = (size_t)1 << 30;
    char* d; int* sink;
    hipMalloc(&d, big); hipMalloc(&sink, 4096);
    hipMemset(d, 1, big);
    for (size_t fp : {(size_t)16 << 20, (size_t)128 << 20, big}) {
        run<1, 1>(d, fp, 0, sink, "contiguous 64 KiB, drained each tile");
        run<1, 2>(d, fp, 0, sink, "contiguous 64 KiB, 2 tiles in flight");
        run<0, 1>(d, fp, 1536, sink, "512 rows x 128 B @ stride 1536, drained");
        run<0, 2>(d, fp, 1536, sink, "512 rows x 128 B @ stride 1536, 2 in flight");
        run<0, 2>(d, fp, 6144, sink, "512 rows x 128 B @ stride 6144, 2 in flight");
    }
    return 0;
}
