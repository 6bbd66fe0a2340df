// U-Net decoder pieces of the fine-tune model (row f2 of SURVEY.md 8f; BASELINE config 5) on gfx950, all HBM-bound:
//   * nearest-neighbour x2 upsampling fused with the skip concatenation, and its adjoint
//       smp DecoderBlock.forward: x = F.interpolate(x, scale_factor=2, mode="nearest"); x = torch.cat([x, skip], 1)
//       (segmentation_models_pytorch/decoders/unet/decoder.py; called from reference src/models/hooknet.py:24-27,95-98)
//   * the "hook": centre crop of the context decoder's block-1 output, x[:, :, 12:20, 12:20]
//       (reference src/models/hooknet.py:29-32), and its adjoint (add into the cropped window)
//   * multiclass soft Dice loss from logits, forward + backward
//       smp.losses.DiceLoss(MULTICLASS_MODE, classes=[1..n], from_logits=True) (reference tools/ssl_finetune.py:287-288):
//       p = softmax(logits); per class c over ALL pixels of the batch: dice_c = 2 sum(p_c t_c) / max(sum(p_c + t_c), eps);
//       loss = mean over the selected classes of (1 - dice_c) * [sum t_c > 0]
//   * NHWC storage -> NCHW fp32 (the logits handed back to the caller)
// segmentation_models_pytorch is a third-party dependency outside the reference tree (absent from this image): its
// published algorithm is restated, parity unpinned.  Tensors are NHWC, channel counts multiples of the 16-byte chunk.
#include "common.h"
#include "../../include/msfwsi_hip.h"

namespace {

constexpr int kT = 256;
constexpr int kMaxCls = 32;

inline unsigned ugrid(long total) {
    long b = (total + kT - 1) / kT;
    if (b > 256 * 32) b = 256 * 32;
    return (unsigned)(b < 1 ? 1 : b);
}

// out[n][H][W][Cx+Cs] = [ x[n][H/2][W/2][:] | skip[n][H][W][:] ]
template <typename T>
__global__ void upcat_fwd_kernel(const T* __restrict__ x, const T* __restrict__ skip, T* __restrict__ out, int N, int h,
                                 int w, int Cx, int Cs) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int H = 2 * h, W = 2 * w, cx = Cx / VEC, cs = Cs / VEC, ct = cx + cs;
    const long total = (long)N * H * W * ct;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ct);
        const long pix = i / ct;
        const int ww = (int)(pix % W);
        const int hh = (int)((pix / W) % H);
        const long n = pix / ((long)W * H);
        uint4 v;
        if (c < cx) v = *reinterpret_cast<const uint4*>(x + ((n * h + (hh >> 1)) * w + (ww >> 1)) * Cx + c * VEC);
        else v = *reinterpret_cast<const uint4*>(skip + pix * Cs + (c - cx) * VEC);
        *reinterpret_cast<uint4*>(out + pix * (Cx + Cs) + c * VEC) = v;
    }
}

// dx[n][h][w][:] = sum over the 2x2 window of dout[..., :Cx] (fp32 sum, one rounding);  dskip = dout[..., Cx:]
template <typename T>
__global__ void upcat_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dx, T* __restrict__ dskip, int N, int h, int w,
                                 int Cx, int Cs) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int H = 2 * h, W = 2 * w, cx = Cx / VEC, cs = Cs / VEC, Ct = Cx + Cs;
    const long nx = (long)N * h * w * cx, ns = (long)N * H * W * cs;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nx + ns; i += (long)gridDim.x * blockDim.x) {
        if (i < nx) {
            const int c = (int)(i % cx);
            const long pix = i / cx;
            const int ww = (int)(pix % w);
            const int hh = (int)((pix / w) % h);
            const long n = pix / ((long)w * h);
            float acc[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float f[VEC];
                    unpack16<T>(*reinterpret_cast<const uint4*>(dout + ((n * H + 2 * hh + a) * W + 2 * ww + b) * Ct + c * VEC),
                                f);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] += f[e];
                }
            *reinterpret_cast<uint4*>(dx + pix * Cx + c * VEC) = pack16<T>(acc);
        } else if (dskip != nullptr) {
            const long j = i - nx;
            const int c = (int)(j % cs);
            const long pix = j / cs;
            *reinterpret_cast<uint4*>(dskip + pix * Cs + c * VEC) =
                *reinterpret_cast<const uint4*>(dout + pix * Ct + Cx + c * VEC);
        }
    }
}

// fwd: out[n][ch][cw][C] = x[n][y0+..][x0+..][C];  bwd (ACC): x[window] += out  (fp32 add, one rounding)
template <typename T, bool BWD>
__global__ void crop_kernel(T* __restrict__ x, T* __restrict__ out, int N, int H, int W, int C, int y0, int x0, int ch,
                            int cw) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int cv = C / VEC;
    const long total = (long)N * ch * cw * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        const long pix = i / cv;
        const int ww = (int)(pix % cw);
        const int hh = (int)((pix / cw) % ch);
        const long n = pix / ((long)cw * ch);
        T* xp = x + ((n * H + y0 + hh) * W + x0 + ww) * C + c * VEC;
        T* op = out + pix * C + c * VEC;
        if constexpr (!BWD) {
            *reinterpret_cast<uint4*>(op) = *reinterpret_cast<const uint4*>(xp);
        } else {
            float a[VEC], b[VEC];
            unpack16<T>(*reinterpret_cast<const uint4*>(xp), a);
            unpack16<T>(*reinterpret_cast<const uint4*>(op), b);
#pragma unroll
            for (int e = 0; e < VEC; ++e) a[e] += b[e];
            *reinterpret_cast<uint4*>(xp) = pack16<T>(a);
        }
    }
}

// NHWC storage [M][CP] (first C channels) -> NCHW fp32 [N][C][HW]
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int N, int C, long HW, int CP) {
    const long total = (long)N * C * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long hw = i % HW;
        const int c = (int)((i / HW) % C);
        const long n = i / (HW * C);
        y[i] = load_elem<T>(x, (size_t)((n * HW + hw) * CP + c));
    }
}

// ---- Dice ------------------------------------------------------------------------------------------------------------
// NC = compile-time bound of the class count (8 / 16 / 32): every per-class array is indexed by fully unrolled loops
// and stays in registers.  (Round 3: arrays of kMaxCls indexed by runtime loops put dice_reduce_kernel's accumulators
// into 528 bytes of scratch per lane.)
template <typename T, int NC>
__device__ __forceinline__ void softmax_px(const T* lp, int C1, float (&p)[NC]) {
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        p[c] = c < C1 ? load_elem<T>(lp, c) : -INFINITY;
        mx = fmaxf(mx, p[c]);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        p[c] = c < C1 ? expf(p[c] - mx) : 0.f;
        s += p[c];
    }
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < NC; ++c) p[c] *= inv;
}

// sums[3][C1] (fp64) += { sum p_c t_c, sum p_c, sum t_c } over all pixels
template <typename T, int NC>
__global__ void dice_reduce_kernel(const T* __restrict__ logits, const long* __restrict__ target, long M, int C1, int CP,
                                   double* __restrict__ sums) {
    __shared__ float sh[3 * kMaxCls];
    for (int i = threadIdx.x; i < 3 * C1; i += blockDim.x) sh[i] = 0.f;
    __syncthreads();
    float acc[3][NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[0][c] = acc[1][c] = acc[2][c] = 0.f;
    for (long m = blockIdx.x * (long)blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        float p[NC];
        softmax_px<T, NC>(logits + m * CP, C1, p);
        const int t = (int)target[m];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            acc[1][c] += p[c];
            const bool hit = t == c;
            acc[0][c] += hit ? p[c] : 0.f;
            acc[2][c] += hit ? 1.f : 0.f;
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = wave_sum(acc[k][c]);
            if ((threadIdx.x & 63) == 0 && c < C1) atomicAdd(&sh[k * C1 + c], v);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * C1; i += blockDim.x) atomicAdd(sums + i, (double)sh[i]);
}

// loss += weight * mean_{c in classes} (1 - dice_c) [T_c > 0];  coef[0][c] = dL/dI_c, coef[1][c] = dL/dS_c  (S = P + T)
__global__ void dice_finalize_kernel(const double* __restrict__ sums, int C1, unsigned class_mask, double eps,
                                     double smooth, double weight, double* loss, float* __restrict__ coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int ncls = 0;
    for (int c = 0; c < C1; ++c) ncls += (class_mask >> c) & 1u;
    double total = 0.0;
    for (int c = 0; c < C1; ++c) {
        coef[c] = coef[C1 + c] = 0.f;
        if (!((class_mask >> c) & 1u)) continue;
        const double I = sums[c], S = sums[C1 + c] + sums[2 * C1 + c], Tc = sums[2 * C1 + c];
        if (!(Tc > 0.0)) continue;  // loss *= (y_true.sum(dims) > 0)
        const double den = S + smooth > eps ? S + smooth : eps;
        const double dice = (2.0 * I + smooth) / den;
        total += 1.0 - dice;
        const double w = weight / (double)ncls;
        coef[c] = (float)(-w * 2.0 / den);
        coef[C1 + c] = (float)(S + smooth > eps ? w * (2.0 * I + smooth) / (den * den) : 0.0);
    }
    *loss += weight * total / (double)ncls;
}

// dlogits[m][j] = gs * p_j (dp_j - sum_c p_c dp_c),  dp_c = a_c [t == c] + b_c;  channels C1..CP-1 get 0
template <typename T, int NC>
__global__ void dice_bwd_kernel(const T* __restrict__ logits, const long* __restrict__ target, long M, int C1, int CP,
                                const float* __restrict__ coef, const float* __restrict__ grad_scale,
                                T* __restrict__ dlogits) {
    const float gs = grad_scale != nullptr ? *grad_scale : 1.f;
    float ca[NC], cb[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        ca[c] = c < C1 ? coef[c] : 0.f;
        cb[c] = c < C1 ? coef[C1 + c] : 0.f;
    }
    for (long m = blockIdx.x * (long)blockDim.x + threadIdx.x; m < M; m += (long)gridDim.x * blockDim.x) {
        float p[NC], dp[NC];
        softmax_px<T, NC>(logits + m * CP, C1, p);
        const int t = (int)target[m];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            dp[c] = (t == c ? ca[c] : 0.f) + cb[c];
            dot = fmaf(p[c], dp[c], dot);
        }
        T* o = dlogits + m * CP;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < CP) store_elem<T>(o, c, c < C1 ? gs * p[c] * (dp[c] - dot) : 0.f);
        for (int c = NC; c < CP; ++c) store_elem<T>(o, c, 0.f);
    }
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int msfwsi_upcat_fwd(int dtype, const void* x, const void* skip, void* out, int N, int h, int w, int Cx, int Cs,
                                void* stream) {
    MSFWSI_CHECK_ARG(msfwsi_dtype_ok(dtype) && x && out && N > 0 && h > 0 && w > 0 && Cx > 0 && Cs >= 0);
    MSFWSI_CHECK_ARG((Cs == 0) == (skip == nullptr));
    const int vec = msfwsi_vec_of(dtype);
    MSFWSI_CHECK_ARG(Cx % vec == 0 && Cs % vec == 0);
    const long total = (long)N * 4 * h * w * ((Cx + Cs) / vec);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(upcat_fwd_kernel<T>, dim3(ugrid(total)), dim3(kT), 0, ST(stream), (const T*)x,
                           (const T*)skip, (T*)out, N, h, w, Cx, Cs));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_upcat_bwd(int dtype, const void* dout, void* dx, void* dskip, int N, int h, int w, int Cx, int Cs,
                                void* stream) {
    MSFWSI_CHECK_ARG(msfwsi_dtype_ok(dtype) && dout && dx && N > 0 && h > 0 && w > 0 && Cx > 0 && Cs >= 0);
    MSFWSI_CHECK_ARG(Cs > 0 || dskip == nullptr);
    const int vec = msfwsi_vec_of(dtype);
    MSFWSI_CHECK_ARG(Cx % vec == 0 && Cs % vec == 0);
    const long total = (long)N * h * w * (Cx / vec) + (dskip != nullptr ? (long)N * 4 * h * w * (Cs / vec) : 0);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(upcat_bwd_kernel<T>, dim3(ugrid(total)), dim3(kT), 0, ST(stream), (const T*)dout,
                           (T*)dx, (T*)dskip, N, h, w, Cx, Cs));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_crop(int dtype, void* x, void* out, int N, int H, int W, int C, int y0, int x0, int ch, int cw,
                           int backward, void* stream) {
    MSFWSI_CHECK_ARG(msfwsi_dtype_ok(dtype) && x && out && N > 0 && C > 0 && C % msfwsi_vec_of(dtype) == 0);
    MSFWSI_CHECK_ARG(y0 >= 0 && x0 >= 0 && ch > 0 && cw > 0 && y0 + ch <= H && x0 + cw <= W);
    const long total = (long)N * ch * cw * (C / msfwsi_vec_of(dtype));
    if (backward) {
        MSFWSI_WITH_T(dtype, hipLaunchKernelGGL((crop_kernel<T, true>), dim3(ugrid(total)), dim3(kT), 0, ST(stream), (T*)x,
                               (T*)out, N, H, W, C, y0, x0, ch, cw));
    } else {
        MSFWSI_WITH_T(dtype, hipLaunchKernelGGL((crop_kernel<T, false>), dim3(ugrid(total)), dim3(kT), 0, ST(stream), (T*)x,
                               (T*)out, N, H, W, C, y0, x0, ch, cw));
    }
    return msfwsi_launch_status();
}

extern "C" int msfwsi_nhwc_to_nchw(int dtype, const void* x, float* y, int N, int C, long HW, int CP, void* stream) {
    MSFWSI_CHECK_ARG(msfwsi_dtype_ok(dtype) && x && y && N > 0 && C > 0 && HW > 0 && CP >= C);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(ugrid((long)N * C * HW)), dim3(kT), 0, ST(stream),
                           (const T*)x, y, N, C, HW, CP));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_dice_loss(int dtype, const void* logits, const long* target, long M, int C1, int CP,
                                unsigned class_mask, double eps, double smooth, double weight, double* sums,
                                double* loss, float* coef, const float* grad_scale, void* dlogits, void* stream) {
    MSFWSI_CHECK_ARG(msfwsi_dtype_ok(dtype) && logits && target && sums && loss && coef && M > 0);
    MSFWSI_CHECK_ARG(C1 > 0 && C1 <= kMaxCls && CP >= C1 && CP % msfwsi_vec_of(dtype) == 0);
#define MSFWSI_DICE_NC(KERNEL, GRID, ...)                                                                              \
    do {                                                                                                               \
        if (C1 <= 8) { MSFWSI_WITH_T(dtype, hipLaunchKernelGGL((KERNEL<T, 8>), GRID, dim3(kT), 0, ST(stream), __VA_ARGS__)); }   \
        else if (C1 <= 16) { MSFWSI_WITH_T(dtype, hipLaunchKernelGGL((KERNEL<T, 16>), GRID, dim3(kT), 0, ST(stream), __VA_ARGS__)); } \
        else { MSFWSI_WITH_T(dtype, hipLaunchKernelGGL((KERNEL<T, 32>), GRID, dim3(kT), 0, ST(stream), __VA_ARGS__)); }  \
    } while (0)
    MSFWSI_DICE_NC(dice_reduce_kernel, dim3(ugrid(M) > 1024 ? 1024 : ugrid(M)), (const T*)logits, target, M, C1, CP, sums);
    int rc = msfwsi_launch_status();
    if (rc != MSFWSI_OK) return rc;
    hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(64), 0, ST(stream), sums, C1, class_mask, eps, smooth, weight,
                       loss, coef);
    rc = msfwsi_launch_status();
    if (rc != MSFWSI_OK || dlogits == nullptr) return rc;
    MSFWSI_DICE_NC(dice_bwd_kernel, dim3(ugrid(M)), (const T*)logits, target, M, C1, CP, coef, grad_scale, (T*)dlogits);
    return msfwsi_launch_status();
}
