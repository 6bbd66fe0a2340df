// Validation metrics of the fine-tune / evaluation loops on gfx950 (integer, HBM-bound):
//   pred = argmax over classes of the segmentation logits          (tools/ssl_finetune.py:526  torch.argmax(preds, dim=1))
//   tp / fp / fn / tn per image and class                          (:527-533  smp.metrics.get_stats(pred-1, target-1,
//                                                                    mode="multiclass", ignore_index=-1, num_classes=C);
//                                                                    tools/evaluate.py:285-305 the same)
// segmentation_models_pytorch is a third-party dependency that is not part of the reference tree; its published
// algorithm (functional._get_stats_multiclass, smp >= 0.3.2) is restated: per image, with `ignore = target == ignore_index`
// both maps set to -1 where ignored, tp[c] = #{pred == target == c}, fp[c] = #{pred == c} - tp[c],
// fn[c] = #{target == c} - tp[c], tn[c] = L - tp - fp - fn - #ignored; values outside [0, C) fall out of the histograms.
#include "common.h"
#include "../../include/msfwsi_hip.h"

namespace {

constexpr int kMaxClasses = 64;

// counts[img][4][C] (int64, zero-initialised): slot 0 = tp, 1 = #pred==c, 2 = #target==c, 3[0] = #ignored
// LOGITS: pred = argmax_c logits[img][c][pix] (first maximum wins, as torch.argmax) + pred_shift; otherwise pred is read.
template <typename T, bool LOGITS>
__global__ void seg_count_kernel(const T* __restrict__ logits, const long* __restrict__ pred_in,
                                 const long* __restrict__ target, long L, int nch, int C, long pred_shift,
                                 long target_shift, long ignore_index, int has_ignore,
                                 unsigned long long* __restrict__ counts) {
    __shared__ unsigned int h[3 * kMaxClasses + 1];
    const int img = blockIdx.y;
    for (int i = threadIdx.x; i < 3 * C + 1; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const long base = (long)img * L;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < L; p += (long)gridDim.x * blockDim.x) {
        long pr;
        if constexpr (LOGITS) {
            const T* lp = logits + (long)img * nch * L + p;
            float best = load_elem<T>(lp, 0);
            int arg = 0;
            for (int c = 1; c < nch; ++c) {
                const float v = load_elem<T>(lp, (size_t)c * L);
                if (v > best || (v != v && best == best)) {  // NaN counts as the maximum, like torch.argmax
                    best = v;
                    arg = c;
                }
            }
            pr = (long)arg + pred_shift;
        } else {
            pr = pred_in[base + p] + pred_shift;
        }
        long tg = target[base + p] + target_shift;
        if (has_ignore && tg == ignore_index) {
            atomicAdd(&h[3 * C], 1u);
            continue;  // both maps are -1 there: outside every histogram
        }
        if (pr >= 0 && pr < C) atomicAdd(&h[1 * C + (int)pr], 1u);
        if (tg >= 0 && tg < C) {
            atomicAdd(&h[2 * C + (int)tg], 1u);
            if (pr == tg) atomicAdd(&h[(int)tg], 1u);
        }
    }
    __syncthreads();
    unsigned long long* dst = counts + (long)img * 4 * C;
    for (int i = threadIdx.x; i < 3 * C + 1; i += blockDim.x)
        if (h[i] != 0) atomicAdd(dst + i, (unsigned long long)h[i]);
}

__global__ void seg_finalize_kernel(const unsigned long long* __restrict__ counts, int N, int C, long L,
                                    long* __restrict__ tp, long* __restrict__ fp, long* __restrict__ fn,
                                    long* __restrict__ tn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int img = i / C, c = i - img * C;
    const unsigned long long* src = counts + (long)img * 4 * C;
    const long t = (long)src[c], oc = (long)src[C + c], tc = (long)src[2 * C + c], ign = (long)src[3 * C];
    tp[i] = t;
    fp[i] = oc - t;
    fn[i] = tc - t;
    tn[i] = L - t - (oc - t) - (tc - t) - ign;
}

// scores[0..2] = micro F1 / IoU / accuracy over all images and classes; scores[3 + c], [3 + C + c], [3 + 2C + c] = the
// same per class on the counts summed over images (reduction=None on tp.sum(0) ..., ssl_finetune.py:544-551);
// 0/0 -> zero_division (smp default 1.0)
__global__ void seg_scores_kernel(const long* __restrict__ tp, const long* __restrict__ fp, const long* __restrict__ fn,
                                  const long* __restrict__ tn, int N, int C, double zero_division,
                                  double* __restrict__ scores) {
    const int c = threadIdx.x;  // thread C handles the micro reduction
    if (c > C) return;
    double a = 0, b = 0, d = 0, e = 0;
    for (int i = 0; i < N; ++i)
        for (int k = (c == C ? 0 : c); k < (c == C ? C : c + 1); ++k) {
            a += (double)tp[i * C + k];
            b += (double)fp[i * C + k];
            d += (double)fn[i * C + k];
            e += (double)tn[i * C + k];
        }
    auto div = [&](double num, double den) { return den == 0.0 ? zero_division : num / den; };
    const double f1 = div(2.0 * a, 2.0 * a + d + b), iou = div(a, a + b + d), acc = div(a + e, a + b + d + e);
    if (c == C) {
        scores[0] = f1;
        scores[1] = iou;
        scores[2] = acc;
    } else {
        scores[3 + c] = f1;
        scores[3 + C + c] = iou;
        scores[3 + 2 * C + c] = acc;
    }
}

// smp reductions "micro-imagewise" (out[0..2]: per image the counts summed over classes, score per image, mean over
// images) and "macro-imagewise" (out[3..5]: score per (image, class), mean over both).  One workgroup.
__global__ void seg_scores_imagewise_kernel(const long* __restrict__ tp, const long* __restrict__ fp,
                                            const long* __restrict__ fn, const long* __restrict__ tn, int N, int C,
                                            double zero_division, double* __restrict__ out) {
    __shared__ double red[6][256];
    auto div = [&](double num, double den) { return den == 0.0 ? zero_division : num / den; };
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        double a = 0, b = 0, d = 0, e = 0;
        for (int k = 0; k < C; ++k) {
            const double ta = (double)tp[i * C + k], tb = (double)fp[i * C + k], td = (double)fn[i * C + k],
                         te = (double)tn[i * C + k];
            a += ta; b += tb; d += td; e += te;
            acc[3] += div(2.0 * ta, 2.0 * ta + td + tb);
            acc[4] += div(ta, ta + tb + td);
            acc[5] += div(ta + te, ta + tb + td + te);
        }
        acc[0] += div(2.0 * a, 2.0 * a + d + b);
        acc[1] += div(a, a + b + d);
        acc[2] += div(a + e, a + b + d + e);
    }
    for (int j = 0; j < 6; ++j) red[j][threadIdx.x] = acc[j];
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int j = 0; j < 6; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 6) out[threadIdx.x] = red[threadIdx.x][0] / (threadIdx.x < 3 ? (double)N : (double)N * (double)C);
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int msfwsi_seg_stats(int logits_dtype, const void* logits, int nch, const long* pred, const long* target,
                                int N, long L, int C, long pred_shift, long target_shift, long ignore_index,
                                int has_ignore, unsigned long long* counts, long* tp, long* fp, long* fn, long* tn,
                                void* stream) {
    MSFWSI_CHECK_ARG(target && counts && tp && fp && fn && tn && N > 0 && L > 0 && C > 0 && C <= kMaxClasses);
    MSFWSI_CHECK_ARG((logits != nullptr) != (pred != nullptr));
    MSFWSI_CHECK_ARG(logits == nullptr || (msfwsi_dtype_ok(logits_dtype) && nch > 0));
    long bx = (L + 255) / 256;
    if (bx > 256) bx = 256;
    const dim3 grid((unsigned)bx, (unsigned)N);
    if (logits != nullptr) {
        MSFWSI_WITH_T(logits_dtype, hipLaunchKernelGGL((seg_count_kernel<T, true>), grid, dim3(256), 0, ST(stream),
                               (const T*)logits, (const long*)nullptr, target, L, nch, C, pred_shift, target_shift,
                               ignore_index, has_ignore, counts));
    } else {
        hipLaunchKernelGGL((seg_count_kernel<float, false>), grid, dim3(256), 0, ST(stream), (const float*)nullptr, pred,
                           target, L, 0, C, pred_shift, target_shift, ignore_index, has_ignore, counts);
    }
    int rc = msfwsi_launch_status();
    if (rc != MSFWSI_OK) return rc;
    hipLaunchKernelGGL(seg_finalize_kernel, dim3((N * C + 255) / 256), dim3(256), 0, ST(stream), counts, N, C, L, tp, fp,
                       fn, tn);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_seg_scores(const long* tp, const long* fp, const long* fn, const long* tn, int N, int C,
                                 double zero_division, double* scores, void* stream) {
    MSFWSI_CHECK_ARG(tp && fp && fn && tn && scores && N > 0 && C > 0 && C <= kMaxClasses);
    hipLaunchKernelGGL(seg_scores_kernel, dim3(1), dim3(128), 0, ST(stream), tp, fp, fn, tn, N, C, zero_division, scores);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_seg_scores_imagewise(const long* tp, const long* fp, const long* fn, const long* tn, int N, int C,
                                           double zero_division, double* scores, void* stream) {
    MSFWSI_CHECK_ARG(tp && fp && fn && tn && scores && N > 0 && C > 0 && C <= kMaxClasses);
    hipLaunchKernelGGL(seg_scores_imagewise_kernel, dim3(1), dim3(256), 0, ST(stream), tp, fp, fn, tn, N, C,
                       zero_division, scores);
    return msfwsi_launch_status();
}
