// Prototype (timing + checksum only): can ONE wave per SIMD keep the matrix pipe fed from LDS-DMA-staged operands with
// v_mfma_f32_32x32x16_f16, and does vector work interleaved into that wave's MFMA stream cost anything?  If both answers are
// good, a 4-wave workgroup with 512 registers per lane could run a tile's epilogue inside the NEXT tile's K loop (the one thing
// the 8-wave kernel of gemm_f16_v5.hip cannot do: its accumulators fill the register file).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/k9_proto.hip -o /tmp/k9 && /tmp/k9
// C[M,N] = A[M,K] W[N,K]^T, 256 x 256 x 64 tiles, 4 waves as 2 x 2, each wave 128 x 128 = 4 x 4 MFMA tiles (256 accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 256, BN = 256, BK = 64, STAGE = (BM + BN) * BK * 2;   // 64 KiB per K-tile

// LDS image of an operand tile: 128-byte rows (64 halves), 16-byte chunk c of row r in slot c ^ ((r >> 1) & 7): a 32-row
// ds_read_b128 fragment (rows l % 32, chunk by l / 32) is conflict free
template <int VALU_PER_MFMA, bool EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k9(const half_t* __restrict__ A, const half_t* __restrict__ W, half_t* __restrict__ C, int M, int N, int K, int tiles_n, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN, nk = K / BK;

    // staging: per K-tile 64 LDS-DMA instructions of 8 rows x 128 B (32 for A, 32 for W); wave w issues 8 + 8.  Scalar base +
    // 32-bit lane offset from inline asm: no vector address arithmetic in the loop, and hipcc does not know a DMA is in flight
    const int srow = lane >> 3, slot = lane & 7;
    unsigned oa[8], ob[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = wave * 64 + j * 8 + srow;                       // tile row of this lane in instruction j
        const int chunk = slot ^ ((r >> 1) & 7);
        oa[j] = (unsigned)(((long)(min(m0 + r, M - 1) - m0) * K + chunk * 8) * 2);
        ob[j] = (unsigned)(((long)(min(n0 + r, N - 1) - n0) * K + chunk * 8) * 2);
    }
    const char* Abase = (const char*)A + (long)m0 * K * 2;
    const char* Wbase = (const char*)W + (long)n0 * K * 2;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
    auto dma = [&](unsigned voff, const char* sbase, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
    };
    // instruction q (0..15) of this wave's share of K-tile kt into buffer buf
    auto dma_q = [&](int buf, int kt, int q) {
        const unsigned dst = lds0 + buf * STAGE + wave * 64 * 128 + (q & 7) * 1024 + (q >= 8 ? BM * 128 : 0);
        if (q < 8) dma(oa[q & 7], Abase + (long)kt * 128, __builtin_amdgcn_readfirstlane(dst));
        else dma(ob[q & 7], Wbase + (long)kt * 128, __builtin_amdgcn_readfirstlane(dst));
    };

    float16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    const int fr = lane & 31, fk = lane >> 5;
    // fragment addresses: row (block row + i*32 + fr), chunk ks*2 + fk.  (row >> 1) & 7 == (fr >> 1) & 7 for every i, so one
    // VGPR per (operand, k-step) and the immediate i * 4096 address everything
    int aoff[4], boff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int sl = ((ks * 2 + fk) ^ ((fr >> 1) & 7)) << 4;
        aoff[ks] = (wm * 128 + fr) * 128 + sl;
        boff[ks] = BM * 128 + (wn * 128 + fr) * 128 + sl;
    }
    half8_t fa[2][4], fb[2][4];
    // read #g (0..7) of k-step ks into register set p
    auto read_one = [&](const char* buf, int ks, int p, int g) {
        if (g < 4) fa[p][g] = *(const half8_t*)(buf + aoff[ks] + g * 4096);
        else fb[p][g - 4] = *(const half8_t*)(buf + boff[ks] + (g - 4) * 4096);
    };
    float dummy[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) dummy[q] = (float)lane * 0.001f + q;
    // MFMA #t (0..15) of a k-step on register set p: j fastest, so B fragment j and A fragment i are reused over 4 / 1 steps
    auto mfma_one = [&](int p, int t) {
        const int i = t >> 2, j = t & 3;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[p][j], fa[p][i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < VALU_PER_MFMA; ++q) dummy[q & 7] = __builtin_fmaf(dummy[q & 7], 1.0001f, 0.5f);
    };
#define FENCE() __builtin_amdgcn_sched_barrier(0)

#pragma unroll
    for (int q = 0; q < 16; ++q) dma_q(0, 0, q);
    if (nk > 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) dma_q(1, 1, q);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // tile 0 landed (tile 1's 16 instructions may be in flight)
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    FENCE();
#pragma unroll
    for (int g = 0; g < 8; ++g) read_one(smem, 0, 0, g);
    FENCE();
    for (int kt = 0; kt < nk; ++kt) {
        const char* cur = smem + (kt & 1) * STAGE;
        const char* nxt = smem + ((kt + 1) & 1) * STAGE;
        // k-steps 0..2: this step's 16 MFMAs with the next step's 8 fragment reads dealt between them
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                mfma_one(ks & 1, 2 * g);
                mfma_one(ks & 1, 2 * g + 1);
                read_one(cur, ks + 1, (ks + 1) & 1, g);
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            FENCE();
        }
        // k-step 3: first half of its MFMAs, the tile boundary, then the second half with the new tile's first reads and the
        // LDS-DMA of the tile after it dealt between them
#pragma unroll
        for (int t = 0; t < 8; ++t) mfma_one(1, t);
        FENCE();
        // (unconditional: MFMAs inside an if / else would make phi copies of 128 accumulator registers.  After the last tile
        // the reads fetch stale LDS that nobody uses.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // tile kt+1 landed (issued a K-tile ago)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // every fragment of tile kt is in registers
        __builtin_amdgcn_s_barrier();                              // ... for every wave: buffer kt & 1 is free
        FENCE();
        const bool more = kt + 2 < nk;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            read_one(nxt, 0, 0, g);
            mfma_one(1, 8 + g);
            if (more) { dma_q(kt & 1, kt + 2, 2 * g); dma_q(kt & 1, kt + 2, 2 * g + 1); }
        }
        FENCE();
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += dummy[q];
    if (EPI) {      // plain (slow) store of the result for the checksum run
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    // D = mfma(Wfrag, Afrag): rows = N side (j), cols = M side (i): lane fr = M row, regs = N cols
                    const int m = m0 + wm * 128 + i * 32 + fr;
                    const int n = n0 + wn * 128 + j * 32 + (k >> 2) * 8 + fk * 4 + (k & 3);
                    if (m < M && n < N) C[(long)m * N + n] = (half_t)acc[i][j][k];
                }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][7] + acc[i][j][15];
    }
    if (s == 12345.678f) sink[tid] = s;
}

template <int V, bool EPI>
float run(const half_t* A, const half_t* W, half_t* C, int M, int N, int K, float* sink, int reps) {
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const size_t lds = 2 * STAGE;
    hipFuncSetAttribute((const void*)k9<V, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k9<V, EPI>), dim3(tiles_m * tiles_n), dim3(256), lds, 0, A, W, C, M, N, K, tiles_n, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k9<V, EPI>), dim3(tiles_m * tiles_n), dim3(256), lds, 0, A, W, C, M, N, K, tiles_n, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
    // correctness on a small shape first
    {
        const int M = 512, N = 512, K = 256;
        std::vector<half_t> hA((size_t)M * K), hW((size_t)N * K), hC((size_t)M * N);
        srand(3);
        for (auto& v : hA) v = (half_t)((rand() / (float)RAND_MAX - 0.5f));
        for (auto& v : hW) v = (half_t)((rand() / (float)RAND_MAX - 0.5f));
        half_t *A, *W, *C; float* sink;
        hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&C, hC.size() * 2); hipMalloc(&sink, 4096);
        hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
        run<0, true>(A, W, C, M, N, K, sink, 1);
        hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int t = 0; t < 4000; ++t) {
            const int m = rand() % M, n = rand() % N;
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
            worst = std::max(worst, std::abs(ref - (double)hC[(size_t)m * N + n]));
        }
        printf("check 512x512x256: max abs err %.4f %s\n", worst, worst < 0.02 ? "OK" : "WRONG");
        hipFree(A); hipFree(W); hipFree(C);
    }
    const int B = argc > 1 ? atoi(argv[1]) : 768, M = B * 197;
    half_t *A, *W; float* sink;
    const size_t na = (size_t)M * 3072, nw = (size_t)3072 * 3072;
    std::vector<half_t> h(1 << 20);
    for (auto& v : h) v = (half_t)((rand() / (float)RAND_MAX - 0.5f));
    hipMalloc(&A, na * 2); hipMalloc(&W, nw * 2); hipMalloc(&sink, 4096);
    for (size_t o = 0; o < na; o += h.size()) hipMemcpy(A + o, h.data(), std::min(h.size(), na - o) * 2, hipMemcpyHostToDevice);
    for (size_t o = 0; o < nw; o += h.size()) hipMemcpy(W + o, h.data(), std::min(h.size(), nw - o) * 2, hipMemcpyHostToDevice);
    struct { const char* name; int N, K; } shapes[] = {{"qkv", 2304, 768}, {"out_proj", 768, 768}, {"c_fc", 3072, 768}, {"c_proj", 768, 3072}};
    for (auto& sh : shapes) {
        const double fl = 2.0 * M * sh.N * sh.K;
        const float t0 = run<0, false>(A, W, nullptr, M, sh.N, sh.K, sink, 10);
        const float t8 = run<8, false>(A, W, nullptr, M, sh.N, sh.K, sink, 10);
        const float t20 = run<20, false>(A, W, nullptr, M, sh.N, sh.K, sink, 10);
        printf("%-9s M=%d N=%d K=%d  K loop alone: %.1f us = %.0f TFLOP/s | +8 VALU per MFMA %.1f us | +20 VALU per MFMA %.1f us\n",
               sh.name, M, sh.N, sh.K, t0, fl / t0 / 1e6, t8, t20);
    }
    return 0;
}
