// Device half of the test transform, first part: Resize(R, bicubic | bilinear) + CenterCrop(R) of decoded uint8 images, BIT-EQUAL to
// PIL.Image.resize + crop (torchvision Resize / CenterCrop of Dassl.pytorch/dassl/data/transforms/transforms.py:495-526).
//
// PIL resamples 8-bit images in two integer passes (Pillow src/libImaging/Resample.c: ImagingResampleHorizontal_8bpc, then
// ImagingResampleVertical_8bpc on the uint8 intermediate): ss = 2^21 + sum pixel * k, k the filter weights in 22-bit fixed point,
// out = clip8(ss >> 22).  The weights and windows depend on the input SIZE only and are built on the host exactly as PIL builds them
// (ovmr_amd/resize.py: double precision, PIL's rounding); the passes below are integer arithmetic, so nothing can differ.
// Only the R x R crop window is computed: R columns of the input rows that window needs, then R rows.
// HBM-bound byte work: ~w * ny * 3 bytes read + ny * R * 3 written and read back + R * R * 3 written per image.
#include "../../include/ovmr_hip.h"
#include "common.h"

namespace {

constexpr int RS_PRECISION = 22;

__device__ __forceinline__ uint8_t clip8(int v) { return (uint8_t)min(max(v >> RS_PRECISION, 0), 255); }

// Horizontal pass.  A workgroup takes RPB consecutive needed rows of one image: the bytes of those rows that the crop window's columns
// reach (xmin of column 0 .. xmin + n of column R - 1: the windows are monotonic) are brought into LDS with 16-byte loads -- the pixels
// are 3-byte groups at arbitrary alignment, a thread reading its own taps from global memory does byte loads at a 3 * scale stride
// (0.5 TB/s, profiles/r05n_pmc_head.json) -- then every thread convolves from LDS.  Frames start at 16-byte-aligned offsets and are padded
// to 16 bytes in the arena (loader.py), so the aligned chunks around a row segment stay inside the image's own padded extent.
constexpr int RS_RPB = 4;                  // rows per workgroup
constexpr int RS_SEG = 12 * 1024;          // LDS bytes per row segment (a 4000-pixel-wide row); wider rows take the direct path

__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ pixels, const ovmr_resize_job* __restrict__ jobs,
                                                       const int32_t* __restrict__ tables, uint8_t* __restrict__ tmp, int R) {
    __shared__ __attribute__((aligned(16))) uint8_t seg[RS_RPB][RS_SEG];
    const ovmr_resize_job jb = jobs[blockIdx.y];
    const int row0 = blockIdx.x * RS_RPB;
    if (jb.passthrough || row0 >= jb.ny) return;
    const int rows = min(RS_RPB, jb.ny - row0);
    const int32_t* t = tables + jb.table;
    const int xlo = t[0], xhi = t[2 * (R - 1)] + t[2 * (R - 1) + 1];          // columns the window reaches
    const long frame_end = jb.in_offset + (long)jb.w * jb.h * 3;
    const long g00 = jb.in_offset + ((long)(jb.y0 + row0) * jb.w + xlo) * 3;  // first byte of the first row's segment
    const int seg_bytes = (xhi - xlo) * 3;
    const bool staged = seg_bytes + 16 <= RS_SEG;
    if (staged) {
        for (int r = 0; r < rows; ++r) {
            const long g0 = g00 + (long)r * jb.w * 3, a0 = g0 & ~15L;
            const int chunks = (int)((g0 + seg_bytes - a0 + 15) >> 4);
            for (int i = threadIdx.x; i < chunks; i += 256) {
                const long src = a0 + 16L * i;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (src < ((frame_end + 15) & ~15L)) v = *(const uint4*)(pixels + src);
                *(uint4*)(&seg[r][16 * i]) = v;
            }
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < rows * R; idx += 256) {
        const int r = idx / R, xo = idx - r * R;
        const int x0 = t[2 * xo], n = t[2 * xo + 1];
        const int32_t* k = t + 2 * R + xo * jb.ksize_h;
        int a0 = 1 << (RS_PRECISION - 1), a1 = a0, a2 = a0;
        if (staged) {
            const long g0 = g00 + (long)r * jb.w * 3;
            const uint8_t* src = &seg[r][(int)(g0 & 15) + (x0 - xlo) * 3];
            for (int i = 0; i < n; ++i) {
                const int kv = k[i];
                a0 += (int)src[3 * i] * kv;
                a1 += (int)src[3 * i + 1] * kv;
                a2 += (int)src[3 * i + 2] * kv;
            }
        } else {
            const uint8_t* src = pixels + jb.in_offset + ((long)(jb.y0 + row0 + r) * jb.w + x0) * 3;
            for (int i = 0; i < n; ++i) {
                const int kv = k[i];
                a0 += (int)src[3 * i] * kv;
                a1 += (int)src[3 * i + 1] * kv;
                a2 += (int)src[3 * i + 2] * kv;
            }
        }
        uint8_t* dst = tmp + jb.tmp_offset + ((long)(row0 + r) * R + xo) * 3;
        dst[0] = clip8(a0);
        dst[1] = clip8(a1);
        dst[2] = clip8(a2);
    }
}

// one thread = one output pixel: three channels
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ pixels, const ovmr_resize_job* __restrict__ jobs,
                                                       const int32_t* __restrict__ tables, const uint8_t* __restrict__ tmp,
                                                       uint8_t* __restrict__ out, int R) {
    const ovmr_resize_job jb = jobs[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= R * R) return;
    uint8_t* dst = out + ((long)blockIdx.y * R * R + idx) * 3;
    if (jb.passthrough) {                                 // already the R x R crop (resized on the host): copy
        const uint8_t* src = pixels + jb.in_offset + (long)idx * 3;
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
        return;
    }
    const int yo = idx / R, xo = idx - yo * R;
    const int32_t* t = tables + jb.table + 2 * R + R * jb.ksize_h;
    const int y0 = t[2 * yo], n = t[2 * yo + 1];
    const int32_t* k = t + 2 * R + yo * jb.ksize_v;
    const uint8_t* src = tmp + jb.tmp_offset + ((long)y0 * R + xo) * 3;
    int a0 = 1 << (RS_PRECISION - 1), a1 = a0, a2 = a0;
    for (int i = 0; i < n; ++i) {
        const int kv = k[i];
        const uint8_t* s = src + (long)i * R * 3;
        a0 += (int)s[0] * kv;
        a1 += (int)s[1] * kv;
        a2 += (int)s[2] * kv;
    }
    dst[0] = clip8(a0);
    dst[1] = clip8(a1);
    dst[2] = clip8(a2);
}

}  // namespace

int launch_resize_crop_u8(const uint8_t* pixels, const ovmr_resize_job* jobs, int n, const int32_t* tables, uint8_t* tmp, int max_ny,
                          uint8_t* out, int R, hipStream_t s) {
    if (n <= 0) return 0;
    if (max_ny > 0)
        hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((max_ny + RS_RPB - 1) / RS_RPB), (unsigned)n), dim3(256), 0, s, pixels, jobs, tables, tmp, R);
    hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)((R * R + 255) / 256), (unsigned)n), dim3(256), 0, s, pixels, jobs, tables, tmp, out, R);
    return (int)hipGetLastError();
}
