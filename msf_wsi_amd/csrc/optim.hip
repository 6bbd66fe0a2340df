// Loss and optimizer kernels of the MSF-WSI pre-train step on gfx950 (all HBM-bound):
//   * fused negative-cosine (SimSiam) loss forward+backward, one wavefront per row
//     (reference: nn.CosineSimilarity(dim=1) terms of tools/ssl_train.py:422,448-466)
//   * non-finite check + dynamic loss-scale update (torch.cuda.amp.GradScaler, ssl_train.py:100,472-474)
//   * flat multi-tensor Adam (torch.optim.Adam defaults, ssl_train.py:309,473) that also refreshes the
//     bf16 compute copy of the weights in the same pass
//   * fp32 -> storage-type casts with channel padding (stem weights / their gradients)
#include "common.h"
#include "../../include/msfwsi_hip.h"

namespace {

template <typename T>
__global__ void cosine_loss_kernel(const T* __restrict__ p, const T* __restrict__ z, long rows, int d, float coef,
                                   const float* __restrict__ loss_scale, float eps, double* loss_accum,
                                   T* __restrict__ dp) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int lane = threadIdx.x & 63;
    const long row = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    if (row >= rows) return;
    const T* pr = p + row * d;
    const T* zr = z + row * d;
    float dot = 0.f, pp = 0.f, zz = 0.f;
    for (int i = lane * VEC; i < d; i += 64 * VEC) {
        float a[VEC], b[VEC];
        unpack16<T>(*reinterpret_cast<const uint4*>(pr + i), a);
        unpack16<T>(*reinterpret_cast<const uint4*>(zr + i), b);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            dot = fmaf(a[e], b[e], dot);
            pp = fmaf(a[e], a[e], pp);
            zz = fmaf(b[e], b[e], zz);
        }
    }
    dot = wave_sum(dot);
    pp = wave_sum(pp);
    zz = wave_sum(zz);
    const float np_raw = sqrtf(pp), nz_raw = sqrtf(zz);
    const float np = fmaxf(np_raw, eps), nz = fmaxf(nz_raw, eps);
    const float inv = 1.f / (np * nz);
    const float cosv = dot * inv;
    if (lane == 0 && loss_accum != nullptr) atomicAdd(loss_accum, (double)coef * (double)cosv);
    if (dp != nullptr) {
        const float gs = coef * (loss_scale != nullptr ? *loss_scale : 1.f);
        // d cos / d p = z/(np*nz) - cos * p / np^2   (second term vanishes while the norm is clamped)
        const float a1 = gs * inv;
        const float a2 = np_raw > eps ? gs * cosv / (np * np) : 0.f;
        T* dr = dp + row * d;
        for (int i = lane * VEC; i < d; i += 64 * VEC) {
            float a[VEC], b[VEC];
            unpack16<T>(*reinterpret_cast<const uint4*>(pr + i), a);
            unpack16<T>(*reinterpret_cast<const uint4*>(zr + i), b);
#pragma unroll
            for (int e = 0; e < VEC; ++e) a[e] = a1 * b[e] - a2 * a[e];
            *reinterpret_cast<uint4*>(dr + i) = pack16<T>(a);
        }
    }
}

// ---- InfoNCE variant (BASELINE.json north_star wording; the reference itself has only the cosine loss) ----
// xhat = x / max(||x||, eps) per row, inv[row] = 1 / max(||x||, eps); one wavefront per row
template <typename T>
__global__ void row_l2norm_kernel(const T* __restrict__ x, T* __restrict__ out, float* __restrict__ inv, long rows, int d,
                                  float eps) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int lane = threadIdx.x & 63;
    const long row = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    if (row >= rows) return;
    const T* xr = x + row * d;
    float ss = 0.f;
    for (int i = lane * VEC; i < d; i += 64 * VEC) {
        float a[VEC];
        unpack16<T>(*reinterpret_cast<const uint4*>(xr + i), a);
#pragma unroll
        for (int e = 0; e < VEC; ++e) ss = fmaf(a[e], a[e], ss);
    }
    ss = wave_sum(ss);
    const float s = 1.f / fmaxf(sqrtf(ss), eps);
    if (lane == 0) inv[row] = s;
    for (int i = lane * VEC; i < d; i += 64 * VEC) {
        float a[VEC];
        unpack16<T>(*reinterpret_cast<const uint4*>(xr + i), a);
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] *= s;
        *reinterpret_cast<uint4*>(out + row * d + i) = pack16<T>(a);
    }
}

// dx = inv * (dxhat - xhat <xhat, dxhat>)   (the clamp is inactive for non-degenerate rows)
template <typename T>
__global__ void row_l2norm_bwd_kernel(const T* __restrict__ xhat, const T* __restrict__ dxhat,
                                      const float* __restrict__ inv, T* __restrict__ dx, long rows, int d) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int lane = threadIdx.x & 63;
    const long row = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    if (row >= rows) return;
    const T* xr = xhat + row * d;
    const T* gr = dxhat + row * d;
    float dot = 0.f;
    for (int i = lane * VEC; i < d; i += 64 * VEC) {
        float a[VEC], b[VEC];
        unpack16<T>(*reinterpret_cast<const uint4*>(xr + i), a);
        unpack16<T>(*reinterpret_cast<const uint4*>(gr + i), b);
#pragma unroll
        for (int e = 0; e < VEC; ++e) dot = fmaf(a[e], b[e], dot);
    }
    dot = wave_sum(dot);
    const float s = inv[row];
    for (int i = lane * VEC; i < d; i += 64 * VEC) {
        float a[VEC], b[VEC];
        unpack16<T>(*reinterpret_cast<const uint4*>(xr + i), a);
        unpack16<T>(*reinterpret_cast<const uint4*>(gr + i), b);
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] = s * (b[e] - a[e] * dot);
        *reinterpret_cast<uint4*>(dx + row * d + i) = pack16<T>(a);
    }
}

// cross entropy of logits[row][:] * inv_tau against label = label0 + row, one 256-thread workgroup per row:
//   loss += coef * (logsumexp - logit[label] * inv_tau);  logits <- coef * gs * inv_tau * (softmax - onehot)  (in place)
template <typename T>
__global__ void softmax_ce_kernel(T* __restrict__ logits, long rows, int n, long label0, float inv_tau, float coef,
                                  const float* __restrict__ grad_scale, double* loss_accum, int write_grad) {
    __shared__ float red[8];
    const long row = blockIdx.x;
    T* lr = logits + row * (long)n;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += blockDim.x) mx = fmaxf(mx, load_elem<T>(lr, i) * inv_tau);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < n; i += blockDim.x) sum += expf(load_elem<T>(lr, i) * inv_tau - mx);
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wv] = sum;
    __syncthreads();
    sum = red[4] + red[5] + red[6] + red[7];
    const long label = label0 + row;
    if (tid == 0 && loss_accum != nullptr) {
        const float lse = mx + logf(sum);
        atomicAdd(loss_accum, (double)coef * (double)(lse - load_elem<T>(lr, label) * inv_tau));
    }
    if (write_grad) {
        __syncthreads();  // every thread has read the label logit's neighbours before anything is overwritten
        const float g = coef * inv_tau * (grad_scale != nullptr ? *grad_scale : 1.f);
        const float inv_sum = 1.f / sum;
        for (int i = tid; i < n; i += blockDim.x) {
            const float pr = expf(load_elem<T>(lr, i) * inv_tau - mx) * inv_sum;
            store_elem<T>(lr, i, g * (pr - (i == label ? 1.f : 0.f)));
        }
    }
}

__global__ void nonfinite_check_kernel(const float* __restrict__ g, long n, float* found) {
    bool bad = false;
    const long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        bad |= !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad |= !isfinite(g[n4 * 4 + threadIdx.x]);
    if (__any(bad) && (threadIdx.x & 63) == 0) *found = 1.f;
}

// torch._amp_update_scale_: scale/growth_tracker live on the device so no step ever syncs on them
__global__ void scaler_update_kernel(float* scale, int* growth_tracker, const float* found, float growth_factor,
                                     float backoff_factor, int growth_interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (*found > 0.f) {
        *scale = *scale * backoff_factor;
        *growth_tracker = 0;
    } else {
        const int succ = *growth_tracker + 1;
        if (succ == growth_interval) {
            const float ns = *scale * growth_factor;
            if (isfinite(ns)) *scale = ns;
            *growth_tracker = 0;
        } else {
            *growth_tracker = succ;
        }
    }
}

// torch.optim.Adam (no amsgrad, no weight decay), same operation order as _single_tensor_adam:
//   m = lerp(m, g, 1-b1); v = b2*v + (1-b2)*g*g; p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps)
// step_dev != NULL: the step count lives on the device (advanced by step_advance_kernel only on steps the
// GradScaler does not skip, exactly like optimizer.step under scaler.step) and the bias corrections are formed here
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float beta1, float beta2, float eps, float step_size,
                            float bc2_sqrt, const int* __restrict__ step_dev, const float* __restrict__ loss_scale,
                            const float* __restrict__ found, unsigned short* __restrict__ p_bf16, int lowp_f16) {
    if (found != nullptr && *found > 0.f) return;  // GradScaler.step skips the update
    if (step_dev != nullptr) {
        const double t = (double)*step_dev;
        step_size = (float)((double)lr / (1.0 - pow((double)beta1, t)));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, t));
    }
    const float inv_scale = loss_scale != nullptr ? 1.f / *loss_scale : 1.f;
    const long n4 = n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pp = &pv.x;
        const float* gg = &gv.x;
        float* mm = &mv.x;
        float* vq = &vv.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * inv_scale;
            mm[e] = mm[e] + (gr - mm[e]) * (1.f - beta1);
            vq[e] = vq[e] * beta2 + (1.f - beta2) * gr * gr;
            const float denom = sqrtf(vq[e]) / bc2_sqrt + eps;
            pp[e] = pp[e] - step_size * (mm[e] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (p_bf16 != nullptr) {
            uint2 o;
            if (lowp_f16) {
                o.x = pack2_f16(pp[0], pp[1]);
                o.y = pack2_f16(pp[2], pp[3]);
            } else {
                o.x = pack2_bf16(pp[0], pp[1]);
                o.y = pack2_bf16(pp[2], pp[3]);
            }
            reinterpret_cast<uint2*>(p_bf16)[i] = o;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = n4 * 4 + threadIdx.x;
        const float gr = g[i] * inv_scale;
        m[i] = m[i] + (gr - m[i]) * (1.f - beta1);
        v[i] = v[i] * beta2 + (1.f - beta2) * gr * gr;
        const float denom = sqrtf(v[i]) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (m[i] / denom);
        if (p_bf16 != nullptr) p_bf16[i] = lowp_f16 ? float_to_f16_bits(p[i]) : float_to_bf16_bits(p[i]);
    }
}

// Adam's "step" advances only when the update is applied (found == 0): torch's GradScaler.step does not call
// optimizer.step on an overflowed step, so bias corrections and the checkpointed step never count skipped steps
__global__ void step_advance_kernel(int* step, const float* found) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && !(found != nullptr && *found > 0.f)) *step += 1;
}

__global__ void cast_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long n, int f16) {
    const long n4 = n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        uint2 o;
        if (f16) {
            o.x = pack2_f16(v.x, v.y);
            o.y = pack2_f16(v.z, v.w);
        } else {
            o.x = pack2_bf16(v.x, v.y);
            o.y = pack2_bf16(v.z, v.w);
        }
        reinterpret_cast<uint2*>(dst)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float x = src[n4 * 4 + threadIdx.x];
        dst[n4 * 4 + threadIdx.x] = f16 ? float_to_f16_bits(x) : float_to_bf16_bits(x);
    }
}

// storage type -> fp32 (the fold algebra of DESIGN.md 3.1 runs on exactly the weights the MFMA multiplies)
template <typename T>
__global__ void upcast_kernel(const T* __restrict__ src, float* __restrict__ dst, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = load_elem<T>(src, i);
}

// rows x cols fp64 block with leading dimension ld := 0 (one slot of a sharded [nshard][slots][C] accumulator)
__global__ void zero_f64_2d_kernel(double* __restrict__ p, long rows, int cols, long ld) {
    const long total = rows * cols;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        p[(i / cols) * ld + (i % cols)] = 0.0;
}

// [rows][C] fp32 -> [rows][CP] storage type, zero padded (stem weights: C=3 -> CP=8 / 4)
template <typename T>
__global__ void pad_cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long rows, int C, int CP) {
    const long total = rows * CP;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / CP;
        const int c = (int)(i - r * CP);
        store_elem<T>(dst, i, c < C ? src[r * C + c] : 0.f);
    }
}
// adjoint: dst[rows][C] += src[rows][CP][:C]
__global__ void unpad_add_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows, int C, int CP) {
    const long total = rows * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        atomicAdd(dst + i, src[r * CP + c]);  // the other view's stream adds into the same gradient
    }
}

inline unsigned sgrid(long total, int threads) {
    long b = (total + threads - 1) / threads;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int msfwsi_cosine_loss(int dtype, const void* p, const void* z, long rows, int d, float coef,
                                  const float* loss_scale, float eps, double* loss_accum, void* dp, void* stream) {
    MSFWSI_CHECK_ARG(msfwsi_dtype_ok(dtype) && p && z && rows > 0 && d > 0);
    MSFWSI_CHECK_ARG(d % msfwsi_vec_of(dtype) == 0);
    const long blocks = (rows + 3) / 4;
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(cosine_loss_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, ST(stream),
                           (const T*)p, (const T*)z, rows, d, coef, loss_scale, eps, loss_accum, (T*)dp));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_row_l2norm(int dtype, const void* x, void* out, float* inv, long rows, int d, float eps,
                                 void* stream) {
    MSFWSI_CHECK_ARG(msfwsi_dtype_ok(dtype) && x && out && inv && rows > 0 && d > 0 && d % msfwsi_vec_of(dtype) == 0);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(row_l2norm_kernel<T>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST(stream),
                           (const T*)x, (T*)out, inv, rows, d, eps));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_row_l2norm_bwd(int dtype, const void* xhat, const void* dxhat, const float* inv, void* dx, long rows,
                                     int d, void* stream) {
    MSFWSI_CHECK_ARG(msfwsi_dtype_ok(dtype) && xhat && dxhat && inv && dx && rows > 0 && d > 0);
    MSFWSI_CHECK_ARG(d % msfwsi_vec_of(dtype) == 0);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(row_l2norm_bwd_kernel<T>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                           ST(stream), (const T*)xhat, (const T*)dxhat, inv, (T*)dx, rows, d));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_softmax_ce(int dtype, void* logits, long rows, int n, long label0, float inv_tau, float coef,
                                 const float* grad_scale, double* loss_accum, int write_grad, void* stream) {
    MSFWSI_CHECK_ARG(msfwsi_dtype_ok(dtype) && logits && rows > 0 && n > 0 && label0 >= 0 && label0 + rows <= n);
    MSFWSI_CHECK_ARG(rows <= 0x7fffffffL && inv_tau > 0.f);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(softmax_ce_kernel<T>, dim3((unsigned)rows), dim3(256), 0, ST(stream), (T*)logits,
                           rows, n, label0, inv_tau, coef, grad_scale, loss_accum, write_grad));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_nonfinite_check(const float* g, long n, float* found, void* stream) {
    MSFWSI_CHECK_ARG(g && found && n > 0);
    hipLaunchKernelGGL(nonfinite_check_kernel, dim3(sgrid(n / 4 + 1, 256)), dim3(256), 0, ST(stream), g, n, found);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_scaler_update(float* scale, int* growth_tracker, const float* found, float growth_factor,
                                    float backoff_factor, int growth_interval, void* stream) {
    MSFWSI_CHECK_ARG(scale && growth_tracker && found && growth_interval > 0);
    hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, ST(stream), scale, growth_tracker, found,
                       growth_factor, backoff_factor, growth_interval);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_adam(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                           float eps, long step, const int* step_dev, const float* loss_scale, const float* found,
                           void* p_lowp, int lowp_dtype, void* stream) {
    MSFWSI_CHECK_ARG(p && g && m && v && n > 0 && (step >= 1 || step_dev != nullptr));
    MSFWSI_CHECK_ARG(p_lowp == nullptr || lowp_dtype == MSFWSI_DT_BF16 || lowp_dtype == MSFWSI_DT_F16);
    void* p_bf16 = p_lowp;
    float step_size = 0.f, bc2_sqrt = 1.f;
    if (step_dev == nullptr) {
        const double bc1 = 1.0 - pow((double)beta1, (double)step);
        const double bc2 = 1.0 - pow((double)beta2, (double)step);
        step_size = (float)((double)lr / bc1);
        bc2_sqrt = (float)sqrt(bc2);
    }
    hipLaunchKernelGGL(adam_kernel, dim3(sgrid(n / 4 + 1, 256)), dim3(256), 0, ST(stream), p, g, m, v, n, lr, beta1,
                       beta2, eps, step_size, bc2_sqrt, step_dev, loss_scale, found, (unsigned short*)p_bf16,
                       lowp_dtype == MSFWSI_DT_F16 ? 1 : 0);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_adam_step_advance(int* step, const float* found, void* stream) {
    MSFWSI_CHECK_ARG(step != nullptr);
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, ST(stream), step, found);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_cast_lowp(int dtype, const float* src, void* dst, long n, void* stream) {
    MSFWSI_CHECK_ARG(src && dst && n > 0 && (dtype == MSFWSI_DT_BF16 || dtype == MSFWSI_DT_F16));
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(sgrid(n / 4 + 1, 256)), dim3(256), 0, ST(stream), src,
                       (unsigned short*)dst, n, dtype == MSFWSI_DT_F16 ? 1 : 0);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_upcast_f32(int dtype, const void* src, float* dst, long n, void* stream) {
    MSFWSI_CHECK_ARG(src && dst && n > 0 && (dtype == MSFWSI_DT_BF16 || dtype == MSFWSI_DT_F16));
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(upcast_kernel<T>, dim3(sgrid(n, 256)), dim3(256), 0, ST(stream), (const T*)src,
                           dst, n));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_zero_f64_2d(double* p, long rows, int cols, long ld, void* stream) {
    MSFWSI_CHECK_ARG(p && rows > 0 && cols > 0 && ld >= cols);
    hipLaunchKernelGGL(zero_f64_2d_kernel, dim3(sgrid(rows * cols, 256)), dim3(256), 0, ST(stream), p, rows, cols, ld);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_pad_cast(int dtype, const float* src, void* dst, long rows, int C, int CP, void* stream) {
    MSFWSI_CHECK_ARG(msfwsi_dtype_ok(dtype) && src && dst && rows > 0 && C > 0 && CP >= C);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(pad_cast_kernel<T>, dim3(sgrid(rows * CP, 256)), dim3(256), 0, ST(stream), src,
                           (T*)dst, rows, C, CP));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_unpad_add(const float* src, float* dst, long rows, int C, int CP, void* stream) {
    MSFWSI_CHECK_ARG(src && dst && rows > 0 && C > 0 && CP >= C);
    hipLaunchKernelGGL(unpad_add_kernel, dim3(sgrid(rows * C, 256)), dim3(256), 0, ST(stream), src, dst, rows, C, CP);
    return msfwsi_launch_status();
}
