// Colour augmentations of the pre-train front end on the device (row f3 of SURVEY.md 8f, the part round 2 left on the
// host): what the reference's DataLoader workers run per sample before the geometric part (tools/ssl_train.py:176-201),
//   albu.ColorJitter(0.4, 0.4, 0.4, 0.1, p=0.8)  ->  albu.ToGray(p=0.2)  ->  OneOf(GaussianBlur([19,23], [0.1,2.0]), Sharpen)
// on uint8 RGB images [N][H][W][3]: the whole 1024x1024 tile for the target views (src/utils/data/bcss.py:166-170), the
// 224x224 crop for the context views.  The random decisions (which adjustments, their order and factors, kernel sizes)
// are inputs drawn by the caller; this file is the arithmetic.
//
// PARITY UNPINNED: albumentations / cv2 are third-party and absent here; their published 8-bit arithmetic is restated
// (oracle/augment_oracle.py states the same, line for line, in numpy and is the checker):
//   brightness / contrast  look-up tables in float64, truncated:  v*f  |  v*f + mean(gray)*(1-f)
//   saturation             cv2.addWeighted(img, f, gray, 1-f) in fp32, rounded to nearest-even, saturated
//   hue                    H plane of cv2's 8-bit HSV image shifted by 180*f (mod 180), back through the float formula
//   gray                   cv2 RGB2GRAY, 14-bit fixed point (4899, 9617, 1868)
//   GaussianBlur           separable, float64-normalised taps cast to fp32, BORDER_REFLECT_101, rows first, one rounding
//                          (OpenCV's bit-exact 8-bit path with 8-bit taps is not reproduced)
//   Sharpen                3x3 correlation with (1-a)*identity + a*[[-1,-1,-1],[-1,8+l,-1],[-1,-1,-1]], REFLECT_101
// Every floating-point operation is written with the explicit round-to-nearest intrinsics: no contraction into FMAs, the
// results equal the numpy statement bit for bit.  HBM-bound byte work: 3 B in + 3 B out per pixel and stage.
#include "common.h"
#include "../../include/msfwsi_hip.h"

// hipcc contracts a*b + c into one FMA by default, also through __fmul_rn / __fadd_rn (plain operators underneath): with
// the contraction the 3x3 sharpening sums landed on the other side of .5 ties (162 of 15360 values one level off the numpy
// statement).  This file is therefore compiled with -ffp-contract=off (Makefile): every product is rounded before it is
// added, as numpy / cv2's scalar code do.  (A file-scope `#pragma clang fp contract(off)` does not reach the header
// intrinsics' own operators.)

namespace {

constexpr int OP_NONE = 0, OP_BRIGHTNESS = 1, OP_CONTRAST = 2, OP_SATURATION = 3, OP_HUE = 4, OP_GRAY = 5;
constexpr int FILT_NONE = 0, FILT_BLUR = 1, FILT_SHARPEN = 2;
constexpr int kMaxTaps = 32;

__device__ __forceinline__ int gray_u8(int r, int g, int b) { return (r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14; }

__device__ __forceinline__ unsigned char trunc_u8(double x) {  // np.clip(x, 0, 255).astype(uint8)
    x = x < 0.0 ? 0.0 : (x > 255.0 ? 255.0 : x);
    return (unsigned char)(int)x;
}

__device__ __forceinline__ unsigned char round_u8(float x) {  // saturate_cast<uchar>(cvRound(x))
    const float r = rintf(x);
    return (unsigned char)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
}

// cv2 RGB2HSV, 8-bit, H in [0,180): 12-bit fixed-point division tables sdiv[v] = round(255*4096/v), hdiv[d] = round(180*4096/(6 d))
__device__ __forceinline__ void rgb2hsv_u8(int r, int g, int b, int& h, int& s, int& v) {
    v = max(max(r, g), b);
    const int vmin = min(min(r, g), b);
    const int diff = v - vmin;
    const int sdiv = v ? __double2int_rn((double)(255 << 12) / (double)v) : 0;
    const int hdiv = diff ? __double2int_rn((double)(180 << 12) / (6.0 * (double)diff)) : 0;
    s = (int)(((long)diff * sdiv + (1 << 11)) >> 12);
    int hh = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    hh = (int)(((long)hh * hdiv + (1 << 11)) >> 12);  // arithmetic shift: floor, as cv2's
    h = hh < 0 ? hh + 180 : hh;
}

// cv2 HSV2RGB, 8-bit through the float formula
__device__ __forceinline__ void hsv2rgb_u8(int h, int s, int v, unsigned char& r, unsigned char& g, unsigned char& b) {
    const float fv = __fmul_rn((float)v, 1.0f / 255.0f);
    float fb = fv, fg = fv, fr = fv;
    if (s != 0) {
        const float fs = __fmul_rn((float)s, 1.0f / 255.0f);
        const float hh = __fmul_rn((float)h, 6.0f / 180.0f);
        int sector = (int)floorf(hh);
        const float f = __fsub_rn(hh, (float)sector);
        sector = sector % 6;
        float tab[4];
        tab[0] = fv;
        tab[1] = __fmul_rn(fv, __fsub_rn(1.0f, fs));
        tab[2] = __fmul_rn(fv, __fsub_rn(1.0f, __fmul_rn(fs, f)));
        tab[3] = __fmul_rn(fv, __fsub_rn(1.0f, __fmul_rn(fs, __fsub_rn(1.0f, f))));
        // (b, g, r) per sector: {1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0}
        const int ib = sector == 0 ? 1 : sector == 1 ? 1 : sector == 2 ? 3 : sector == 3 ? 0 : sector == 4 ? 0 : 2;
        const int ig = sector == 0 ? 3 : sector == 1 ? 0 : sector == 2 ? 0 : sector == 3 ? 2 : sector == 4 ? 1 : 1;
        const int ir = sector == 0 ? 0 : sector == 1 ? 2 : sector == 2 ? 1 : sector == 3 ? 1 : sector == 4 ? 3 : 0;
        fb = ib == 0 ? tab[0] : ib == 1 ? tab[1] : ib == 2 ? tab[2] : tab[3];
        fg = ig == 0 ? tab[0] : ig == 1 ? tab[1] : ig == 2 ? tab[2] : tab[3];
        fr = ir == 0 ? tab[0] : ir == 1 ? tab[1] : ir == 2 ? tab[2] : tab[3];
    }
    r = round_u8(__fmul_rn(fr, 255.0f));
    g = round_u8(__fmul_rn(fg, 255.0f));
    b = round_u8(__fmul_rn(fb, 255.0f));
}

// sums[n] += sum over the image of gray(pixel): integer partial sums, exact in fp64 whatever the order of the additions
__global__ void gray_sum_kernel(const unsigned char* __restrict__ img, long npix, double* __restrict__ sums) {
    const int n = blockIdx.y;
    const unsigned char* p = img + (long)n * npix * 3;
    unsigned long long acc = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x)
        acc += (unsigned)gray_u8(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]);
    __shared__ unsigned long long sh[256];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && sh[0] != 0) atomicAdd(sums + n, (double)sh[0]);
}

__global__ void color_stage_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, long npix,
                                   const int* __restrict__ op, const double* __restrict__ factor,
                                   const double* __restrict__ gray_sum) {
    const int n = blockIdx.y;
    const int o = op[n];
    const unsigned char* p = in + (long)n * npix * 3;
    unsigned char* q = out + (long)n * npix * 3;
    if (o == OP_NONE && p == q) return;
    const double f = factor != nullptr ? factor[n] : 0.0;
    const double mean = (o == OP_CONTRAST && gray_sum != nullptr) ? gray_sum[n] / (double)npix : 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const int r = p[i * 3], g = p[i * 3 + 1], b = p[i * 3 + 2];
        unsigned char ro = (unsigned char)r, go = (unsigned char)g, bo = (unsigned char)b;
        if (o == OP_BRIGHTNESS) {
            ro = trunc_u8(__dmul_rn((double)r, f));
            go = trunc_u8(__dmul_rn((double)g, f));
            bo = trunc_u8(__dmul_rn((double)b, f));
        } else if (o == OP_CONTRAST) {
            if (f == 0.0) {
                ro = go = bo = (unsigned char)(int)(mean + 0.5);
            } else {
                const double off = __dmul_rn(mean, 1.0 - f);
                ro = trunc_u8(__dadd_rn(__dmul_rn((double)r, f), off));
                go = trunc_u8(__dadd_rn(__dmul_rn((double)g, f), off));
                bo = trunc_u8(__dadd_rn(__dmul_rn((double)b, f), off));
            }
        } else if (o == OP_SATURATION) {
            const float a = (float)f, bt = (float)(1.0 - f);
            const float gy = __fmul_rn((float)gray_u8(r, g, b), bt);
            ro = round_u8(__fadd_rn(__fmul_rn((float)r, a), gy));
            go = round_u8(__fadd_rn(__fmul_rn((float)g, a), gy));
            bo = round_u8(__fadd_rn(__fmul_rn((float)b, a), gy));
        } else if (o == OP_HUE) {
            if (f != 0.0) {
                int h, s, v;
                rgb2hsv_u8(r, g, b, h, s, v);
                const double hs = fmod(__dadd_rn((double)h, __dmul_rn(180.0, f)), 180.0);  // np.mod: sign of the divisor
                h = (int)(unsigned char)(int)(hs < 0.0 ? hs + 180.0 : hs);
                hsv2rgb_u8(h, s, v, ro, go, bo);
            }
        } else if (o == OP_GRAY) {
            ro = go = bo = (unsigned char)gray_u8(r, g, b);
        }
        q[i * 3] = ro; q[i * 3 + 1] = go; q[i * 3 + 2] = bo;
    }
}

__device__ __forceinline__ int reflect101(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

// rows of the separable Gaussian: tmp[y][x][c] = sum_i taps[i] * src[y][reflect(x + i - r)][c]   (images with kind == BLUR)
__global__ void blur_rows_kernel(const unsigned char* __restrict__ in, float* __restrict__ tmp, int H, int W,
                                 const int* __restrict__ kind, const int* __restrict__ ksize,
                                 const float* __restrict__ taps) {
    const int n = blockIdx.y;
    if (kind[n] != FILT_BLUR) return;
    const int ks = ksize[n], r = ks / 2;
    const float* t = taps + (long)n * kMaxTaps;
    const long npix = (long)H * W;
    const unsigned char* p = in + (long)n * npix * 3;
    float* q = tmp + (long)n * npix * 3;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long row = (i - x) * 3;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int k = 0; k < ks; ++k) {
            const unsigned char* s = p + row + (long)reflect101(x + k - r, W) * 3;
            a0 = __fadd_rn(a0, __fmul_rn(t[k], (float)s[0]));
            a1 = __fadd_rn(a1, __fmul_rn(t[k], (float)s[1]));
            a2 = __fadd_rn(a2, __fmul_rn(t[k], (float)s[2]));
        }
        q[i * 3] = a0; q[i * 3 + 1] = a1; q[i * 3 + 2] = a2;
    }
}

// columns of the Gaussian (from tmp), the 3x3 sharpening correlation (from the image), or a copy
__global__ void filter_finish_kernel(const unsigned char* __restrict__ in, const float* __restrict__ tmp,
                                     unsigned char* __restrict__ out, int H, int W, const int* __restrict__ kind,
                                     const int* __restrict__ ksize, const float* __restrict__ taps) {
    const int n = blockIdx.y;
    const int kd = kind[n];
    const long npix = (long)H * W;
    const unsigned char* p = in + (long)n * npix * 3;
    unsigned char* q = out + (long)n * npix * 3;
    if (kd == FILT_NONE && p == q) return;
    const float* t = taps + (long)n * kMaxTaps;
    const int ks = ksize[n], r = ks / 2;
    const float* tp = tmp + (long)n * npix * 3;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)(i / W);
        if (kd == FILT_BLUR) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            for (int k = 0; k < ks; ++k) {
                const float* s = tp + ((long)reflect101(y + k - r, H) * W + x) * 3;
                a0 = __fadd_rn(a0, __fmul_rn(t[k], s[0]));
                a1 = __fadd_rn(a1, __fmul_rn(t[k], s[1]));
                a2 = __fadd_rn(a2, __fmul_rn(t[k], s[2]));
            }
            q[i * 3] = round_u8(a0); q[i * 3 + 1] = round_u8(a1); q[i * 3 + 2] = round_u8(a2);
        } else if (kd == FILT_SHARPEN) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const unsigned char* s = p + ((long)reflect101(y + dy, H) * W + reflect101(x + dx, W)) * 3;
                    const float m = t[(dy + 1) * 3 + dx + 1];
                    a0 = __fadd_rn(a0, __fmul_rn(m, (float)s[0]));
                    a1 = __fadd_rn(a1, __fmul_rn(m, (float)s[1]));
                    a2 = __fadd_rn(a2, __fmul_rn(m, (float)s[2]));
                }
            q[i * 3] = round_u8(a0); q[i * 3 + 1] = round_u8(a1); q[i * 3 + 2] = round_u8(a2);
        } else {
            q[i * 3] = p[i * 3]; q[i * 3 + 1] = p[i * 3 + 1]; q[i * 3 + 2] = p[i * 3 + 2];
        }
    }
}

inline dim3 image_grid(long npix, int N) {
    long bx = (npix + 255) / 256;
    if (bx > 1024) bx = 1024;
    return dim3((unsigned)bx, (unsigned)N);
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int msfwsi_gray_sum(const unsigned char* img, int N, int H, int W, double* sums, void* stream) {
    MSFWSI_CHECK_ARG(img && sums && N > 0 && N <= 65535 && H > 0 && W > 0);
    hipLaunchKernelGGL(gray_sum_kernel, image_grid((long)H * W, N), dim3(256), 0, ST(stream), img, (long)H * W, sums);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_color_stage(const unsigned char* in, unsigned char* out, int N, int H, int W, const int* op,
                                  const double* factor, const double* gray_sum, void* stream) {
    MSFWSI_CHECK_ARG(in && out && op && N > 0 && N <= 65535 && H > 0 && W > 0);
    hipLaunchKernelGGL(color_stage_kernel, image_grid((long)H * W, N), dim3(256), 0, ST(stream), in, out, (long)H * W, op,
                       factor, gray_sum);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_blur_sharpen(const unsigned char* in, unsigned char* out, float* tmp, int N, int H, int W,
                                   const int* kind, const int* ksize, const float* taps, void* stream) {
    MSFWSI_CHECK_ARG(in && out && tmp && kind && ksize && taps && N > 0 && N <= 65535 && H >= 16 && W >= 16);
    MSFWSI_CHECK_ARG(in != out);  // the stencils read neighbours
    hipLaunchKernelGGL(blur_rows_kernel, image_grid((long)H * W, N), dim3(256), 0, ST(stream), in, tmp, H, W, kind, ksize,
                       taps);
    hipLaunchKernelGGL(filter_finish_kernel, image_grid((long)H * W, N), dim3(256), 0, ST(stream), in, tmp, out, H, W, kind,
                       ksize, taps);
    return msfwsi_launch_status();
}
