// Scaled-dot-product self attention for hd = 64 heads (nn.MultiheadAttention as used at
// clip/model.py:171,184-188): softmax(q k^T / sqrt(hd) + mask) v per (sequence, head).
//
// fp16 kernel (image tower: L = 197/577 no mask; text tower: L <= 77 causal):
//   * one workgroup = up to four 16-row query tiles (one per wave) of one (sequence, head);
//     key/value blocks of 64 keys are staged in LDS and shared by the four waves;
//   * both products are issued "swapped" so the query index always sits on the MFMA lane
//     (lane & 15): S^T = K Q^T, then O^T = V^T P^T.  The row max / row sum are then two
//     shuffles (xor 16, 32), P never leaves registers (the S^T accumulator IS the P^T operand
//     under a key permutation that the V^T fragment mirrors), and the online-softmax rescale of
//     O^T is lane local;
//   * V is transposed while it is staged ([d][key] rows of 136 B, conflict-free ds_read_b64);
//     K rows are XOR-swizzled 128-byte rows (conflict-free ds_read_b128).
// fp32 kernel (aggregator, L = n_ctx + shots <= 128): one thread per query row, K/V in LDS.
#include "common.h"

namespace {

constexpr int KB = 64;      // keys per block
constexpr int VT_LD = 68;   // halves per V^T row (64 keys + 4 pad -> 136 B)

template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn_f16_v0(const half_t* __restrict__ qkv, half_t* __restrict__ out,
                                                   int L, int Lq, int H, int nT, int nWG, float scale_log2e) {
    // Lq <= L: only the first Lq query rows of every sequence are computed, output rows are b*Lq + q
    __shared__ __attribute__((aligned(16))) half_t sK[KB * 64];
    __shared__ __attribute__((aligned(16))) half_t sVt[64 * VT_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int D = H * 64, ld = 3 * D;
    const int wg = blockIdx.x % nWG, bh = blockIdx.x / nWG;
    const int h = bh % H, b = bh / H;
    const int t0 = (wg * nT) / nWG, t1 = ((wg + 1) * nT) / nWG;
    const int qt = t0 + wave;
    const bool active = qt < t1;
    const half_t* base = qkv + (long)b * L * ld + h * 64;
    const int q = qt * 16 + fr;
    const int qc = min(q, L - 1);

    half8_t qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *(const half8_t*)(base + (long)qc * ld + ks * 32 + fg * 8);

    float m_run = -INFINITY, l_run = 0.f;
    float4_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int kmax = CAUSAL ? min(L, t1 * 16) : L;
    for (int k0 = 0; k0 < kmax; k0 += KB) {
        __syncthreads();
        // ---- stage K block (64 keys x 128 B)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + 256 * i, key = id >> 3, c = id & 7;
            const int kc = min(k0 + key, L - 1);
            uint4 v = *(const uint4*)(base + D + (long)kc * ld + c * 8);
            *(uint4*)(sK + key * 64 + ((c ^ (key & 7)) << 3)) = v;
        }
        // ---- stage V block transposed: thread = (4 keys) x (4 d)
        {
            const int dg = tid & 15, kg = tid >> 4;
            half4_t r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kc = min(k0 + kg * 4 + i, L - 1);
                r[i] = *(const half4_t*)(base + 2 * D + (long)kc * ld + dg * 4);
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                half4_t w = {r[0][d], r[1][d], r[2][d], r[3][d]};
                *(half4_t*)(sVt + (dg * 4 + d) * VT_LD + kg * 4) = w;
            }
        }
        __syncthreads();

        if (active && (!CAUSAL || k0 <= qt * 16 + 15)) {
            float4_t s[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                s[nt] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    half8_t kf = *(const half8_t*)(sK + (nt * 16 + fr) * 64 + ((((ks << 2) + fg) ^ (fr & 7)) << 3));
                    s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], s[nt], 0, 0, 0);
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + nt * 16 + fg * 4 + r;
                    const bool ok = (key < L) && (!CAUSAL || key <= q);
                    const float v = ok ? s[nt][r] * scale_log2e : -INFINITY;
                    s[nt][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = exp2f(m_run - m_new);
            float psum = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = exp2f(s[nt][r] - m_new);
                    s[nt][r] = p;
                    psum += p;
                }
            psum += __shfl_xor(psum, 16, 64);
            psum += __shfl_xor(psum, 32, 64);
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                half8_t pf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pf[j] = (half_t)s[2 * s2][j];
                    pf[4 + j] = (half_t)s[2 * s2 + 1][j];
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const half_t* vr = sVt + (dt * 16 + fr) * VT_LD + s2 * 32 + fg * 4;
                    half4_t v0 = *(const half4_t*)vr;
                    half4_t v1 = *(const half4_t*)(vr + 16);
                    half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[dt], 0, 0, 0);
                }
            }
        }
    }
    if (active && q < Lq) {
        const float inv = 1.0f / l_run;
        half_t* op = out + ((long)b * Lq + q) * D + h * 64 + fg * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            half4_t w = {(half_t)(o[dt][0] * inv), (half_t)(o[dt][1] * inv), (half_t)(o[dt][2] * inv), (half_t)(o[dt][3] * inv)};
            *(half4_t*)(op + dt * 16) = w;
        }
    }
}

// fp32, L <= 128: one thread per query row, single-pass online softmax.
__global__ __launch_bounds__(128) void attn_f32_small(const float* __restrict__ qkv, float* __restrict__ out,
                                                      int L, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sK = sm;
    float* sV = sm + L * 64;
    const int tid = threadIdx.x;
    const int h = blockIdx.x % H, b = blockIdx.x / H;
    const int D = H * 64, ld = 3 * D;
    const float* base = qkv + (long)b * L * ld + h * 64;
    for (int i = tid; i < L * 16; i += blockDim.x) {
        const int row = i >> 4, c = i & 15;
        *(float4_t*)(sK + row * 64 + c * 4) = *(const float4_t*)(base + D + (long)row * ld + c * 4);
        *(float4_t*)(sV + row * 64 + c * 4) = *(const float4_t*)(base + 2 * D + (long)row * ld + c * 4);
    }
    __syncthreads();
    if (tid >= L) return;
    float qv[64], o[64];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float4_t t = *(const float4_t*)(base + (long)tid * ld + c * 4);
        qv[c * 4] = t[0]; qv[c * 4 + 1] = t[1]; qv[c * 4 + 2] = t[2]; qv[c * 4 + 3] = t[3];
    }
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int key = 0; key < L; ++key) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 64; ++d) s += qv[d] * sK[key * 64 + d];
        s *= scale;
        const float mn = fmaxf(m, s);
        const float alpha = __expf(m - mn), p = __expf(s - mn);
        l = l * alpha + p;
#pragma unroll
        for (int d = 0; d < 64; ++d) o[d] = o[d] * alpha + p * sV[key * 64 + d];
        m = mn;
    }
    const float inv = 1.0f / l;
    float* op = out + ((long)b * L + tid) * D + h * 64;
#pragma unroll
    for (int c = 0; c < 16; ++c)
        *(float4_t*)(op + c * 4) = (float4_t){o[c * 4] * inv, o[c * 4 + 1] * inv, o[c * 4 + 2] * inv, o[c * 4 + 3] * inv};
}

}  // namespace

int launch_attention_f16_v1(const half_t* qkv, half_t* out, int B, int L, int Lq, int H, int causal, hipStream_t s);  // attention_v1.hip


int launch_attention_f16_v3(const half_t* qkv, half_t* out, int B, int L, int Lq, int H, int causal, hipStream_t s);  // attention_v3.hip
int launch_attention_f16_v5(const half_t* qkv, half_t* out, int B, int L, int Lq, int H, int causal, int mode, hipStream_t s);  // attention_v5.hip
#ifdef OVMR_EXPERIMENTS
int launch_attention_f16_v6(const half_t* qkv, half_t* out, int B, int L, int Lq, int H, int causal, int mode, hipStream_t s);  // attention_v6.hip
int launch_attention_f16_v4(const half_t* qkv, half_t* out, int B, int L, int Lq, int H, int causal, int mode, hipStream_t s);  // attention_v4.hip
#endif

int launch_attention_f16(const half_t* qkv, half_t* out, int B, int L, int H, int causal, int variant, hipStream_t s) {
    return launch_attention_f16_q(qkv, out, B, L, L, H, causal, variant, s);
}

int launch_attention_f16_q(const half_t* qkv, half_t* out, int B, int L, int Lq, int H, int causal, int variant, hipStream_t s) {
    if (B <= 0 || L <= 0 || Lq <= 0) return 0;
    if (Lq > L) return -2;
    if (variant >= 1 && Lq == L && L <= 32) {   // short sequences (text prompts truncated to their last needed row): one wave per (sequence, head)
        const long row0 = 0;
        const int rc = launch_attention_f16_short(qkv, out, 1, &B, &L, &row0, H, causal, s);
        if (rc != -100) return rc;
    }
#ifdef OVMR_EXPERIMENTS   // variant 4: variant 3's arithmetic with free-running producer / consumer waves (LDS flags instead of the barrier),
    // the vehicle of the r02 ablations (400 + mode: timing-only); same speed as 3, so only the experiment build carries it
    if (variant == 4 || (variant >= 400 && variant < 1000)) {
        const int rc = launch_attention_f16_v4(qkv, out, B, L, Lq, H, causal, variant == 4 ? 0 : variant - 400, s);
        if (rc != -100) return rc;
        variant = 1;
    }
#else
    if (variant == 4) variant = 3;
#endif
    if (variant == 3) {               // single-pass kernel for the ViT-B/16 image shape; long sequences (ViT-L: L = 257 / 577) as variant 5
        int rc = launch_attention_f16_v3(qkv, out, B, L, Lq, H, causal, s);
        if (rc != -100) return rc;
        rc = launch_attention_f16_v5(qkv, out, B, L, Lq, H, causal, 0, s);
        if (rc != -100) return rc;
        variant = 1;
    }
#ifdef OVMR_EXPERIMENTS
    if (variant == 6 || (variant >= 60 && variant < 68)) {   // two query tiles per wave (60 + mode: attention_v6.hip; slower than variant 5)
        const int rc = launch_attention_f16_v6(qkv, out, B, L, Lq, H, causal, variant == 6 ? 0 : variant - 60, s);
        if (rc != -100) return rc;
        variant = 1;
    }
    if (variant >= 50 && variant < 306) {   // variant 5's scheduling experiments (50 + mode; 66 = the folded exponent)
        const int rc = launch_attention_f16_v5(qkv, out, B, L, Lq, H, causal, variant - 50, s);
        if (rc != -100) return rc;
        variant = 1;
    }
#endif
    if (variant == 5) {               // 32x32x16 flash kernel (non-causal, L >= 256); other shapes as variant 1
        const int rc = launch_attention_f16_v5(qkv, out, B, L, Lq, H, causal, 0, s);
        if (rc != -100) return rc;
        variant = 1;
    }
    if (variant == 1 && L >= 128) {   // short (text) sequences: one key block, the plain kernel is faster (tools/attn_bench.py)
        int rc = launch_attention_f16_v1(qkv, out, B, L, Lq, H, causal, s);
        if (rc != -100) return rc;
    }
    const int nT = (Lq + 15) / 16, nWG = (nT + 3) / 4;
    const float sl2e = 0.125f * 1.4426950408889634f;   // hd^-0.5 * log2(e), hd = 64
    const dim3 grid((unsigned)((long)B * H * nWG));
    if (causal) hipLaunchKernelGGL(attn_f16_v0<true>, grid, dim3(256), 0, s, qkv, out, L, Lq, H, nT, nWG, sl2e);
    else hipLaunchKernelGGL(attn_f16_v0<false>, grid, dim3(256), 0, s, qkv, out, L, Lq, H, nT, nWG, sl2e);
    return (int)hipGetLastError();
}

int launch_attention_f32(const float* qkv, float* out, int B, int L, int H, hipStream_t s) {
    if (B <= 0 || L <= 0) return 0;
    if (L > 128) return -2;
    const size_t lds = (size_t)2 * L * 64 * sizeof(float);
    hipLaunchKernelGGL(attn_f32_small, dim3(B * H), dim3(128), lds, s, qkv, out, L, H, 0.125f);
    return (int)hipGetLastError();
}
