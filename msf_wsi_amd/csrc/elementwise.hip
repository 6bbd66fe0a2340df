// Streaming (HBM-bound) kernels of the MSF-WSI pre-train step on gfx950: BatchNorm finalize / apply /
// backward, residual add + ReLU, stem max-pool, global average pool, jigsaw row gather, 2-D copies.
// All activation tensors are NHWC viewed as [M][C] with C contiguous; every thread moves 16-byte chunks.
// Per-channel reductions: a thread owns one 16-byte channel chunk and walks rows; partials are combined
// through LDS and leave as fp64 atomics into `nshard` replicas (memory-side atomics, spread over rows).
#include "common.h"
#include "../../include/msfwsi_hip.h"

namespace {

constexpr int kThreads = 256;

// ---------------------------------------------------------------------------------------------
// column-reduction scaffolding
// ---------------------------------------------------------------------------------------------
struct ColGrid {
    int cw;         // chunk columns handled by one block
    int nrl;        // row lanes per block
    int rows_per_block;
    dim3 grid;
};

inline ColGrid make_col_grid(long M, int C, int vec, int target_blocks) {
    ColGrid g;
    const int cpr = C / vec;
    g.cw = cpr < kThreads ? cpr : kThreads;
    g.nrl = kThreads / g.cw;
    const int colblocks = (cpr + g.cw - 1) / g.cw;
    long rowblocks = target_blocks / colblocks;
    if (rowblocks < 1) rowblocks = 1;
    long rpb = (M + rowblocks - 1) / rowblocks;
    // every row lane sums at least 16 rows before its block's fp64 atomics: the heads' tensors (256 .. 4096 rows, up
    // to 18432 columns) were cut into 2-row blocks whose 8 M atomics took 100 us per launch (95 GB/s)
    if (rpb < 16L * g.nrl) rpb = 16L * g.nrl;
    rowblocks = (M + rpb - 1) / rpb;
    if (rowblocks > 65535) {
        rowblocks = 65535;
        rpb = (M + rowblocks - 1) / rowblocks;
        rowblocks = (M + rpb - 1) / rpb;
    }
    g.rows_per_block = (int)rpb;
    g.grid = dim3((unsigned)colblocks, (unsigned)rowblocks);
    return g;
}

// combine per-thread partial sums acc[NS][VEC] over the block's row lanes and add them (fp64 atomics)
// to sums[shard][NS][C].  smem must hold kThreads*NS*VEC floats.
template <int NS, int VEC>
__device__ __forceinline__ void col_commit(float (&acc)[NS][VEC], float* smem, int cw, int nrl, int cc, int rl,
                                           bool active, int chan0, int C, double* sums, int nshard) {
    const int tid = threadIdx.x;
    if (nrl > 1) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int e = 0; e < VEC; ++e) smem[(s * VEC + e) * kThreads + tid] = active ? acc[s][e] : 0.f;
        __syncthreads();
        if (active && rl == 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float t = 0.f;
                    for (int r = 0; r < nrl; ++r) t += smem[(s * VEC + e) * kThreads + r * cw + cc];
                    acc[s][e] = t;
                }
        }
    }
    if (active && rl == 0) {
        double* base = sums + (long)(blockIdx.y % nshard) * NS * C;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int e = 0; e < VEC; ++e) atomicAdd(base + (long)s * C + chan0 + e, (double)acc[s][e]);
    }
}

// ---------------------------------------------------------------------------------------------
// NCHW fp32 image -> NHWC (channel-padded) storage type.  reference: the H2D'd batch of
// tools/ssl_train.py:430-438 entering conv1 (src/models/resnet.py:174,234)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int C, int HW, int CP) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const long total = (long)N * HW * (CP / VEC);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % (CP / VEC));
        const long pix = i / (CP / VEC);
        const int n = (int)(pix / HW);
        const int hw = (int)(pix - (long)n * HW);
        float f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int c = cv * VEC + e;
            f[e] = c < C ? x[((long)n * C + c) * HW + hw] : 0.f;
        }
        *reinterpret_cast<uint4*>(y + pix * CP + cv * VEC) = pack16<T>(f);
    }
}

// ---------------------------------------------------------------------------------------------
// Space-to-depth form of the stem (conv 7x7 / stride 2 / pad 3 on 3 channels, src/models/resnet.py:174):
//   y[n][i][j][(a*2+b)*3 + c] = x[n][c][2i+a][2j+b]   (12 channels, zero-padded to 16),   H, W even
//   out(p, q) = sum_{r,s,c} W[r][s][c] x[c][2p-3+r][2q-3+s]  with  2p-3+r = 2(p-2+ri) + a  <=>  r = 2 ri + a - 1
// => a 4x4 / stride-1 conv with padding 2 (top/left; the output keeps H/2 rows) on y with weights
//   W2[k][ri][si][(a*2+b)*3 + c] = W[k][2ri+a-1][2si+b-1][c]   (0 where an index is -1, and in channels 12..15).
// A filter row is then 4 taps x 16 channels = 64 contiguous elements = whole k slabs (K = 256 instead of the 448 of the
// 7-row-taps form with its 8-padded channels), and the input tensor has half the bytes.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_s2d_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int H, int W) {
    constexpr int VEC = ElemTraits<T>::VEC, CP = 16;
    const int H2 = H / 2, W2 = W / 2;
    const long total = (long)N * H2 * W2 * (CP / VEC);
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(t % (CP / VEC));
        const long pix = t / (CP / VEC);
        const int j = (int)(pix % W2);
        const int i = (int)((pix / W2) % H2);
        const long n = pix / ((long)W2 * H2);
        float f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int ch = cv * VEC + e;
            const int ab = ch / 3, c = ch - ab * 3;
            f[e] = ch < 12 ? x[((n * 3 + c) * H + 2 * i + (ab >> 1)) * W + 2 * j + (ab & 1)] : 0.f;
        }
        *reinterpret_cast<uint4*>(y + pix * CP + cv * VEC) = pack16<T>(f);
    }
}

// w fp32 [K][7][7][3] -> out T [K][4][4][16]
template <typename T>
__global__ void stem_s2d_weights_kernel(const float* __restrict__ w, T* __restrict__ out, int K) {
    const int total = K * 4 * 4 * 16;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int ch = t & 15, si = (t >> 4) & 3, ri = (t >> 6) & 3, k = t >> 8;
        const int ab = ch / 3, c = ch - ab * 3;
        const int r = 2 * ri + (ab >> 1) - 1, s = 2 * si + (ab & 1) - 1;
        store_elem<T>(out, t, (ch < 12 && r >= 0 && s >= 0) ? w[((k * 7 + r) * 7 + s) * 3 + c] : 0.f);
    }
}

// adjoint: dw fp32 [K][7][7][3] += dw2 fp32 [K][4][4][16] gathered back (atomic: both views' streams add here)
__global__ void stem_s2d_wfold_kernel(const float* __restrict__ dw2, float* dw, int K) {
    const int total = K * 7 * 7 * 3;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int c = t % 3, s = (t / 3) % 7, r = (t / 21) % 7, k = t / 147;
        const int ri = (r + 1) >> 1, a = (r + 1) & 1, si = (s + 1) >> 1, b = (s + 1) & 1;
        atomicAdd(dw + t, dw2[((k * 4 + ri) * 4 + si) * 16 + (a * 2 + b) * 3 + c]);
    }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm (training) finalize: sums -> mean / invstd / fused scale+shift, running-stat update.
// reference: nn.BatchNorm2d / BatchNorm1d in train mode (resnet.py:175,59-62; backbone.py:15,18,21,28):
// biased variance for normalisation, unbiased into running_var, momentum 0.1, eps 1e-5.
// ---------------------------------------------------------------------------------------------
// Shard reduction of the finalize kernels: 32 lanes per channel, lane j sums the shards j, j+32, ... (normally one:
// NSHARD = 32) and a butterfly over the 32 lanes adds them -- one load latency instead of a serial chain of `nshard`
// dependent fp64 additions behind as many loads (10-12 us per launch measured, 1232 launches per step; now ~4 us).
// The summation order is fixed, so the result is the same on every run.
constexpr int kFinLanes = 32;                       // lanes per channel
constexpr int kFinChans = 256 / kFinLanes;          // channels per 256-thread workgroup
__device__ __forceinline__ double fin_reduce(double v) {
#pragma unroll
    for (int o = kFinLanes / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kFinLanes);
    return v;
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, int nshard, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* running_mean, float* running_var, long* nbt,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                                   float* __restrict__ invstd_out) {
    const int j = threadIdx.x % kFinLanes;
    const int c = blockIdx.x * kFinChans + threadIdx.x / kFinLanes;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt != nullptr) *nbt += 1;
    double s = 0.0, q = 0.0;
    if (c < C)
        for (int k = j; k < nshard; k += kFinLanes) {
            s += sums[((long)k * 2 + 0) * C + c];
            q += sums[((long)k * 2 + 1) * C + c];
        }
    s = fin_reduce(s);
    q = fin_reduce(q);
    if (c >= C || j != 0) return;
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma != nullptr ? gamma[c] : 1.f;
    const float b = beta != nullptr ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - (float)mean * sc;
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    if (running_mean != nullptr) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// BatchNorm in eval mode (nn.BatchNorm*.eval() with track_running_stats): the affine map comes from the running
// statistics, nothing is reduced or exchanged: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean*scale
__global__ void bn_eval_coeffs_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int C,
                                      float* __restrict__ scale, float* __restrict__ shift,
                                      float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.f / sqrtf(running_var[c] + eps);
    const float sc = (gamma != nullptr ? gamma[c] : 1.f) * invstd;
    scale[c] = sc;
    shift[c] = (beta != nullptr ? beta[c] : 0.f) - running_mean[c] * sc;
    mean_out[c] = running_mean[c];
    invstd_out[c] = invstd;
}

// sum `nshard` replicas of a length-n fp64 vector (the packed message of the cross-replica exchange)
__global__ void shard_sum_kernel(const double* __restrict__ in, int nshard, int n, double* __restrict__ out) {
    const int j = threadIdx.x % kFinLanes;  // see bn_finalize_kernel
    const int i = blockIdx.x * kFinChans + threadIdx.x / kFinLanes;
    double s = 0.0;
    if (i < n)
        for (int k = j; k < nshard; k += kFinLanes) s += in[(long)k * n + i];
    s = fin_reduce(s);
    if (i < n && j == 0) out[i] = s;
}

// ---------------------------------------------------------------------------------------------
// y = act(scale*c + shift [+ identity | + id_scale*identity + id_shift])
// reference: bn2/bn3 + residual add + ReLU (resnet.py:76-80,131-138), and the projector's final
// BatchNorm1d(affine=False) (backbone.py:21) with relu=0, identity=null
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void bn_act_kernel(const T* __restrict__ c, const float* __restrict__ scale, const float* __restrict__ shift,
                              const T* __restrict__ ident, const float* __restrict__ id_scale,
                              const float* __restrict__ id_shift, int relu, T* __restrict__ out, long M, int C) {
    // plain grid-stride loop: measured faster in the step (92.8 ms) than a 2x-unrolled variant with hoisted
    // coefficient loads (106.8 ms) and than one workgroup per 16 KiB (+8 ms on the step)
    constexpr int VEC = ElemTraits<T>::VEC;
    const int cpr = C / VEC;
    const long total = M * cpr;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpr) * VEC;
        float f[VEC];
        unpack16<T>(*reinterpret_cast<const uint4*>(c + i * VEC), f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] = fmaf(f[e], scale[ch + e], shift[ch + e]);
        if (ident != nullptr) {
            float g[VEC];
            unpack16<T>(*reinterpret_cast<const uint4*>(ident + i * VEC), g);
            if (id_scale != nullptr) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) f[e] += fmaf(g[e], id_scale[ch + e], id_shift[ch + e]);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) f[e] += g[e];
            }
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[e] = fmaxf(f[e], 0.f);
        }
        *reinterpret_cast<uint4*>(out + i * VEC) = pack16<T>(f);
    }
}

// ---------------------------------------------------------------------------------------------
// stem: out = maxpool3x3/s2/p1(relu(scale*c0+shift)), plus the window index of the first maximum
// reference: bn1 -> relu -> maxpool (resnet.py:235-237)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void stem_pool_fwd_kernel(const T* __restrict__ c0, const float* __restrict__ scale,
                                     const float* __restrict__ shift, T* __restrict__ out,
                                     unsigned char* __restrict__ amax, int N, int H, int W, int C, int P, int Q) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int cpr = C / VEC;
    const long total = (long)N * P * Q * cpr;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % cpr);
        long pix = i / cpr;
        const int q = (int)(pix % Q);
        pix /= Q;
        const int p = (int)(pix % P);
        const int n = (int)(pix / P);
        const int ch = cv * VEC;
        float sc[VEC], sh[VEC], best[VEC];
        int bidx[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            sc[e] = scale[ch + e];
            sh[e] = shift[ch + e];
            best[e] = -INFINITY;
            bidx[e] = 0;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int h = 2 * p - 1 + r;
            if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int w = 2 * q - 1 + s;
                if ((unsigned)w >= (unsigned)W) continue;
                float f[VEC];
                unpack16<T>(*reinterpret_cast<const uint4*>(c0 + (((long)n * H + h) * W + w) * C + ch), f);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    // round to the storage type first: the pooled value must equal a stored activation
                    const float v = round_to<T>(fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f));
                    if (v > best[e]) {
                        best[e] = v;
                        bidx[e] = r * 3 + s;
                    }
                }
            }
        }
        *reinterpret_cast<uint4*>(out + i * VEC) = pack16<T>(best);
        unsigned char* ap = amax + i * VEC;
        if (VEC == 8) {
            uint2 pk;
            pk.x = bidx[0] | (bidx[1] << 8) | (bidx[2] << 16) | (bidx[3] << 24);
            pk.y = bidx[4 % VEC] | (bidx[5 % VEC] << 8) | (bidx[6 % VEC] << 16) | (bidx[7 % VEC] << 24);
            *reinterpret_cast<uint2*>(ap) = pk;
        } else {
            *reinterpret_cast<unsigned*>(ap) = bidx[0] | (bidx[1] << 8) | (bidx[2] << 16) | (bidx[3] << 24);
        }
    }
}

// The same by COLUMN WALK (round 4, the default): a thread owns one output column q and one channel chunk and walks down
// the output rows of one image.  Window row 2p+1 of output row p is window row 2(p+1)-1 of output row p+1: its
// transformed values' horizontal maximum (value + column index, the first one on ties) is carried in registers, so each
// input row is loaded and transformed ONCE per column -- 6 loads per output instead of 9 -- and the result (value and
// first-maximum position in (r, s) order) is the per-window kernel's bit for bit.
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_fwd_walk_kernel(const T* __restrict__ c0, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, T* __restrict__ out,
                                                                 unsigned char* __restrict__ amax, int N, int H, int W, int C,
                                                                 int P, int Q, int cpp, int qgroups, long ntask) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cols = 64 / cpp;
    const int kc = lane % cpp, jc = lane / cpp;
    const int ch = kc * VEC;
    const long task = (long)blockIdx.x * 4 + wave;
    if (task >= ntask) return;
    const int n = (int)(task / qgroups);
    const int q = (int)(task - (long)n * qgroups) * cols + jc;
    if (q >= Q) return;
    float sc[VEC], sh[VEC], cv[VEC];
    int ci[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        sc[e] = scale[ch + e];
        sh[e] = shift[ch + e];
        cv[e] = -INFINITY;  // carried odd row (none above the first window row)
        ci[e] = 0;
    }
    for (int p = 0; p < P; ++p) {
        float best[VEC];
        int bidx[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            best[e] = cv[e];
            bidx[e] = ci[e];       // r = 0: position code = s of the carried row
            cv[e] = -INFINITY;
            ci[e] = 0;
        }
#pragma unroll
        for (int r = 1; r < 3; ++r) {
            const int h = 2 * p - 1 + r;
            if (h >= H) continue;
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
                const int w = 2 * q - 1 + s2;
                if ((unsigned)w >= (unsigned)W) continue;
                float f[VEC];
                unpack16<T>(*reinterpret_cast<const uint4*>(c0 + (((long)n * H + h) * W + w) * C + ch), f);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float v = round_to<T>(fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f));
                    if (v > best[e]) {
                        best[e] = v;
                        bidx[e] = r * 3 + s2;
                    }
                    if (r == 2 && v > cv[e]) {  // the odd row's own first maximum: row r = 0 of the next window
                        cv[e] = v;
                        ci[e] = s2;
                    }
                }
            }
        }
        const long o = (((long)n * P + p) * Q + q) * C + ch;
        *reinterpret_cast<uint4*>(out + o) = pack16<T>(best);
        unsigned char* ap = amax + o;
        if (VEC == 8) {
            uint2 pk;
            pk.x = bidx[0] | (bidx[1] << 8) | (bidx[2] << 16) | (bidx[3] << 24);
            pk.y = bidx[4 % VEC] | (bidx[5 % VEC] << 8) | (bidx[6 % VEC] << 16) | (bidx[7 % VEC] << 24);
            *reinterpret_cast<uint2*>(ap) = pk;
        } else {
            *reinterpret_cast<unsigned*>(ap) = bidx[0] | (bidx[1] << 8) | (bidx[2] << 16) | (bidx[3] << 24);
        }
    }
}

// backward of the above: g0 = relu'(.) * sum over the (<= 4) windows that selected this pixel of dp,
// plus the BatchNorm-backward sums  S1 = sum g0,  S2 = sum g0*c0
template <typename T>
__global__ void stem_pool_bwd_kernel(const T* __restrict__ dp, const unsigned char* __restrict__ amax,
                                     const T* __restrict__ c0, const float* __restrict__ scale,
                                     const float* __restrict__ shift, T* __restrict__ g0, double* sums, int nshard,
                                     const float* __restrict__ k1, const float* __restrict__ k2,
                                     const float* __restrict__ k3, const T* __restrict__ dact, int N, int H, int W, int C,
                                     int P, int Q, int cw, int nrl, int rows_per_block) {
    // dact (nullable): a second gradient of the stem ACTIVATION relu(bn1(c0)) itself, added before the gate -- the
    // U-Net skip connection taken before the max-pool (smp ResNetEncoder stage 1)
    // two modes (BatchNorm's backward needs the sums of ALL of g before any dc can be formed):
    //   k1 == null: g = gated pool gradient, sums += {sum g, sum g*c}; g0 (if given) = g
    //   k1 != null: g0 = k1*g + k2*c + k3  (g re-derived on the fly instead of being written and re-read)
    constexpr int VEC = ElemTraits<T>::VEC;
    extern __shared__ float smem_f[];
    const int tid = threadIdx.x;
    const int cc = tid % cw, rl = tid / cw;
    const int chunk = blockIdx.x * cw + cc;
    const bool active = rl < nrl && chunk * VEC < C;
    const int ch = chunk * VEC;
    float acc[2][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[0][e] = acc[1][e] = 0.f;
    if (active) {
        float sc[VEC], sh[VEC], q1[VEC], q2[VEC], q3[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            sc[e] = scale[ch + e];
            sh[e] = shift[ch + e];
            q1[e] = k1 != nullptr ? k1[ch + e] : 0.f;
            q2[e] = k1 != nullptr ? k2[ch + e] : 0.f;
            q3[e] = k1 != nullptr ? k3[ch + e] : 0.f;
        }
        const long M = (long)N * H * W;
        const long rbeg = (long)blockIdx.y * rows_per_block;
        const long rend = rbeg + rows_per_block < M ? rbeg + rows_per_block : M;
        for (long m = rbeg + rl; m < rend; m += nrl) {
            const unsigned mu = (unsigned)m;  // N*H*W < 2^31 (checked by the launcher): 32-bit divisions
            const unsigned t = mu / (unsigned)W;
            const int w = (int)(mu - t * (unsigned)W);
            const int n = (int)(t / (unsigned)H);
            const int h = (int)(t - (unsigned)n * (unsigned)H);
            float g[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) g[e] = 0.f;
            // windows (p,q) with 2p-1 <= h <= 2p+1
            const int p_lo = h >> 1, p_hi = (h + 1) >> 1;
            const int q_lo = w >> 1, q_hi = (w + 1) >> 1;
            for (int p = p_lo; p <= p_hi; ++p) {
                if (p >= P) continue;
                const int r = h - (2 * p - 1);
                for (int q = q_lo; q <= q_hi; ++q) {
                    if (q >= Q) continue;
                    const int s = w - (2 * q - 1);
                    const long o = (((long)n * P + p) * Q + q) * C + ch;
                    float d[VEC];
                    unpack16<T>(*reinterpret_cast<const uint4*>(dp + o), d);
                    // the VEC argmax bytes of the chunk in one load
                    unsigned long long ab;
                    if constexpr (VEC == 8) ab = *reinterpret_cast<const unsigned long long*>(amax + o);
                    else ab = *reinterpret_cast<const unsigned*>(amax + o);
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if ((int)((ab >> (8 * e)) & 0xffu) == r * 3 + s) g[e] += d[e];
                }
            }
            float x[VEC];
            unpack16<T>(*reinterpret_cast<const uint4*>(c0 + m * C + ch), x);
            if (dact != nullptr) {
                float da[VEC];
                unpack16<T>(*reinterpret_cast<const uint4*>(dact + m * C + ch), da);
#pragma unroll
                for (int e = 0; e < VEC; ++e) g[e] += da[e];
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if (!(fmaf(x[e], sc[e], sh[e]) > 0.f)) g[e] = 0.f;
                g[e] = round_to<T>(g[e]);
                acc[0][e] += g[e];
                acc[1][e] = fmaf(g[e], x[e], acc[1][e]);
            }
            if (k1 != nullptr) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) g[e] = fmaf(q1[e], g[e], fmaf(q2[e], x[e], q3[e]));
            }
            if (g0 != nullptr) *reinterpret_cast<uint4*>(g0 + m * C + ch) = pack16<T>(g);
        }
    }
    if (sums != nullptr) col_commit<2, VEC>(acc, smem_f, cw, nrl, cc, rl, active, ch, C, sums, nshard);
}

// ---------------------------------------------------------------------------------------------
// The same by COLUMN WALK (round 4, the default for the training path: g0 + sums, no second gradient, no k1 mode).
// The per-pixel kernel above loads the gradient + argmax chunks of every window that covers a pixel: 2.25 window loads
// (24 bytes each) per pixel, 344 bytes through the vector memory path per 64 bytes of output, and each window row is
// fetched by three input rows at three different times (measured 2.8 TB/s algorithmic, 1.55x over-fetch).  Here a thread
// owns one window column q and one channel chunk and walks down the window rows p of ONE image:
//   * it loads window (p, q) ONCE (16 bytes of gradient + VEC argmax bytes, kept packed), keeps the previous row's in
//     registers and takes the right-hand neighbour's (p, q+1) from the lane CPP further on (ds_bpermute; the last
//     column of a wave loads it itself) -- 24 bytes of window data per FOUR output pixels instead of 216;
//   * per step it emits the 2 x 2 pixels (2p-1, 2p) x (2q, 2q+1): a pixel in an odd row / column sits under two window
//     rows / columns, one in an even row / column under one, so every contribution is in registers by then;
//   * c0 is read and g0 written exactly once, in whole 128-byte lines, and the walk keeps each wave on one DRAM page run.
// position codes r*3+s of the window (p', q') that selected pixel (h, w): r = h - (2p'-1), s = w - (2q'-1).
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void pool_take(float (&g)[ElemTraits<T>::VEC], const uint4& dpk, unsigned long long ab,
                                          unsigned code) {
    constexpr int VEC = ElemTraits<T>::VEC;
    float d[VEC];
    unpack16<T>(dpk, d);
#pragma unroll
    for (int e = 0; e < VEC; ++e)
        if (((unsigned)(ab >> (8 * e)) & 0xffu) == code) g[e] += d[e];
}

template <typename T>
__global__ __launch_bounds__(256) void stem_pool_bwd_walk_kernel(
    const T* __restrict__ dp, const unsigned char* __restrict__ amax, const T* __restrict__ c0,
    const float* __restrict__ scale, const float* __restrict__ shift, T* __restrict__ g0, double* sums, int nshard,
    int N, int H, int W, int C, int P, int Q, int cpp /* chunks per pixel = C / VEC */, int qgroups, long ntask) {
    constexpr int VEC = ElemTraits<T>::VEC;
    __shared__ float red[4][2][64 * 8];  // [wave][slot][chunk * VEC + e], C <= 64 * 8 / ... (C == cpp * VEC <= 512)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cols = 64 / cpp;            // window columns per wave
    const int kc = lane % cpp, jc = lane / cpp;
    const int ch = kc * VEC;
    const long task = (long)blockIdx.x * 4 + wave;  // (image, column group)
    float acc[2][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[0][e] = acc[1][e] = 0.f;
    if (task < ntask) {
        const int n = (int)(task / qgroups);
        const int q = (int)(task - (long)n * qgroups) * cols + jc;
        const bool qok = q < Q;
        const bool rok = q + 1 < Q;            // the right-hand neighbour column exists
        const bool last_col = jc == cols - 1;  // its window data is not in this wave: loaded directly
        float sc[VEC], sh[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            sc[e] = scale[ch + e];
            sh[e] = shift[ch + e];
        }
        const int w0 = 2 * q, w1 = 2 * q + 1;
        const bool w0ok = qok && w0 < W, w1ok = qok && w1 < W;
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        uint4 pv = z4, pvR = z4;                                   // window row p-1 (own column, right neighbour)
        unsigned long long pa = ~0ull, paR = ~0ull;                // no position code matches 0xff
        auto load_win = [&](int p, int qq, uint4& d, unsigned long long& a) {
            const long o = (((long)n * P + p) * Q + qq) * C + ch;
            d = *reinterpret_cast<const uint4*>(dp + o);
            if constexpr (VEC == 8) a = *reinterpret_cast<const unsigned long long*>(amax + o);
            else a = 0xffffffff00000000ull | *reinterpret_cast<const unsigned*>(amax + o);
        };
        auto emit = [&](int h, int w, float (&g)[VEC]) {
            const long m = ((long)n * H + h) * W + w;
            float x[VEC];
            unpack16<T>(*reinterpret_cast<const uint4*>(c0 + m * C + ch), x);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if (!(fmaf(x[e], sc[e], sh[e]) > 0.f)) g[e] = 0.f;
                g[e] = round_to<T>(g[e]);
                acc[0][e] += g[e];
                acc[1][e] = fmaf(g[e], x[e], acc[1][e]);
            }
            if (g0 != nullptr) *reinterpret_cast<uint4*>(g0 + m * C + ch) = pack16<T>(g);
        };
        for (int p = 0; p <= P; ++p) {
            // window row p (none at p == P: only the last odd input row 2P-1 is left to emit)
            uint4 cv = z4, cvR = z4;
            unsigned long long ca = ~0ull, caR = ~0ull;
            if (p < P) {
                if (qok) load_win(p, q, cv, ca);
                // the neighbour's chunk from the lane cpp further on (every lane takes part in the shuffle)
                cvR.x = __shfl_down(cv.x, cpp, 64);
                cvR.y = __shfl_down(cv.y, cpp, 64);
                cvR.z = __shfl_down(cv.z, cpp, 64);
                cvR.w = __shfl_down(cv.w, cpp, 64);
                const unsigned alo = __shfl_down((unsigned)ca, cpp, 64), ahi = __shfl_down((unsigned)(ca >> 32), cpp, 64);
                caR = ((unsigned long long)ahi << 32) | alo;
                if (last_col) {
                    cvR = z4;
                    caR = ~0ull;
                    if (rok) load_win(p, q + 1, cvR, caR);
                }
                if (!rok) {
                    cvR = z4;
                    caR = ~0ull;
                }
            }
            const int ho = 2 * p - 1, he = 2 * p;
            if (ho >= 0 && ho < H) {  // odd row 2p-1: windows p-1 (r = 2) and p (r = 0)
                if (w0ok) {
                    float g[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) g[e] = 0.f;
                    pool_take<T>(g, pv, pa, 7u);
                    pool_take<T>(g, cv, ca, 1u);
                    emit(ho, w0, g);
                }
                if (w1ok) {
                    float g[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) g[e] = 0.f;
                    pool_take<T>(g, pv, pa, 8u);
                    pool_take<T>(g, pvR, paR, 6u);
                    pool_take<T>(g, cv, ca, 2u);
                    pool_take<T>(g, cvR, caR, 0u);
                    emit(ho, w1, g);
                }
            }
            if (p < P && he < H) {    // even row 2p: window p only (r = 1)
                if (w0ok) {
                    float g[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) g[e] = 0.f;
                    pool_take<T>(g, cv, ca, 4u);
                    emit(he, w0, g);
                }
                if (w1ok) {
                    float g[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) g[e] = 0.f;
                    pool_take<T>(g, cv, ca, 5u);
                    pool_take<T>(g, cvR, caR, 3u);
                    emit(he, w1, g);
                }
            }
            pv = cv;
            pa = ca;
            pvR = cvR;
            paR = caR;
        }
    }
    if (sums == nullptr) return;
    // per channel: sum over the wave's columns (lanes kc, kc + cpp, ...), then over the four waves, then fp64 atomics
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float v = acc[s][e];
            for (int off = cpp; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            acc[s][e] = v;
        }
    if (lane < cpp) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < VEC; ++e) red[wave][s][lane * VEC + e] = acc[s][e];
    }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += 256) {
        const int s = i / C, c = i - s * C;
        const float t = red[0][s][c] + red[1][s][c] + red[2][s][c] + red[3][s][c];
        atomicAdd(sums + ((long)(blockIdx.x % nshard) * 2 + s) * C + c, (double)t);
    }
}

// ---------------------------------------------------------------------------------------------
// global average pool over H*W (resnet.py:244-250): one wave per (image, 16-byte channel chunk)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void gap_fwd_kernel(const T* __restrict__ y, T* __restrict__ out, int N, int HW, int C, int cw,
                               int nrl, T* __restrict__ sout = nullptr, int W = 0, FastDiv div_w = FastDiv{}) {
    // one workgroup per (channel block, image): lanes run along channels (coalesced 16-byte chunks of a pixel
    // row), `nrl` row lanes stride over the pixels, partial sums meet in LDS
    constexpr int VEC = ElemTraits<T>::VEC;
    extern __shared__ float smem_f[];
    const int tid = threadIdx.x;
    const int cc = tid % cw, rl = tid / cw;
    const int chunk = blockIdx.x * cw + cc;
    const int n = blockIdx.y;
    const bool active = rl < nrl && chunk * VEC < C;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    if (active) {
        const T* base = y + (long)n * HW * C + chunk * VEC;
        // sout (nullable): the pixels of the even rows and columns as a dense [N][H/2][W/2][C] tensor -- the operand of the next
        // stage's strided downsample branch (msfwsi_pixel_stride), taken from the pass that reads y anyway
        const int Qs = (W + 1) >> 1;
        T* sbase = sout != nullptr ? sout + (long)n * ((HW / W + 1) >> 1) * Qs * C + chunk * VEC : nullptr;
        for (int i = rl; i < HW; i += nrl) {
            const uint4 v = *reinterpret_cast<const uint4*>(base + (long)i * C);
            float f[VEC];
            unpack16<T>(v, f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += f[e];
            if (sbase != nullptr) {
                const unsigned h = fast_div((unsigned)i, div_w), w = (unsigned)i - h * (unsigned)W;
                if (((h | w) & 1u) == 0) *reinterpret_cast<uint4*>(sbase + (long)((h >> 1) * Qs + (w >> 1)) * C) = v;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) smem_f[e * kThreads + tid] = active ? acc[e] : 0.f;
    __syncthreads();
    if (active && rl == 0) {
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float t = 0.f;
            for (int r = 0; r < nrl; ++r) t += smem_f[e * kThreads + r * cw + cc];
            acc[e] = t * inv;
        }
        *reinterpret_cast<uint4*>(out + (long)n * C + chunk * VEC) = pack16<T>(acc);
    }
}

// ---------------------------------------------------------------------------------------------
// backward at a residual-block output  y = relu(bn(c_main) + identity):
//   g = (dy + gapg/HW) * (y > 0);  S1 = sum g;  S2 = sum g*c_main;  S3 = sum g*c_ds (downsample branch)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void block_end_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ gapg,
                                     float gap_scale, const T* __restrict__ c_main, const T* __restrict__ c_ds,
                                     T* __restrict__ g_out, double* sums, int nshard, long M, int HW, int C, int cw,
                                     int nrl, int rows_per_block) {
    constexpr int VEC = ElemTraits<T>::VEC;
    extern __shared__ float smem_f[];
    const int tid = threadIdx.x;
    const int cc = tid % cw, rl = tid / cw;
    const int chunk = blockIdx.x * cw + cc;
    const bool active = rl < nrl && chunk * VEC < C;
    const int ch = chunk * VEC;
    float acc[3][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[0][e] = acc[1][e] = acc[2][e] = 0.f;
    if (active) {
        const long rbeg = (long)blockIdx.y * rows_per_block;
        const long rend = rbeg + rows_per_block < M ? rbeg + rows_per_block : M;
        constexpr int UNR = 2;
        for (long m = rbeg + rl; m < rend; m += (long)nrl * UNR) {
            uint4 vd[UNR], vy[UNR], vg[UNR], vm[UNR], vs[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long mm = m + (long)u * nrl;
                if (mm < rend) {
                    const long o = mm * C + ch;
                    if (dy != nullptr) vd[u] = *reinterpret_cast<const uint4*>(dy + o);
                    if (gapg != nullptr) vg[u] = *reinterpret_cast<const uint4*>(gapg + (mm / HW) * C + ch);
                    vy[u] = *reinterpret_cast<const uint4*>(y + o);
                    if (c_main != nullptr) vm[u] = *reinterpret_cast<const uint4*>(c_main + o);
                    if (c_ds != nullptr) vs[u] = *reinterpret_cast<const uint4*>(c_ds + o);
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long mm = m + (long)u * nrl;
                if (mm < rend) {
                    const long o = mm * C + ch;
                    float g[VEC], yy[VEC];
                    if (dy != nullptr) {
                        unpack16<T>(vd[u], g);
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) g[e] = 0.f;
                    }
                    if (gapg != nullptr) {
                        float gg[VEC];
                        unpack16<T>(vg[u], gg);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) g[e] = fmaf(gg[e], gap_scale, g[e]);
                    }
                    unpack16<T>(vy[u], yy);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        g[e] = yy[e] > 0.f ? round_to<T>(g[e]) : 0.f;
                        acc[0][e] += g[e];
                    }
                    if (c_main != nullptr) {  // null: the main-branch sum g*c comes from msfwsi_fold_dots
                        float cm[VEC];
                        unpack16<T>(vm[u], cm);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[1][e] = fmaf(g[e], cm[e], acc[1][e]);
                    }
                    if (c_ds != nullptr) {
                        float cd[VEC];
                        unpack16<T>(vs[u], cd);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[2][e] = fmaf(g[e], cd[e], acc[2][e]);
                    }
                    *reinterpret_cast<uint4*>(g_out + o) = pack16<T>(g);
                }
            }
        }
    }
    col_commit<3, VEC>(acc, smem_f, cw, nrl, cc, rl, active, ch, C, sums, nshard);
}

// backward through an inner activation a = relu(scale*c+shift) (scale==null: identity activation):
//   g = da * (scale*c+shift > 0);  S1 = sum g;  S2 = sum g*c
template <typename T>
__global__ void act_bwd_reduce_kernel(const T* __restrict__ da, const T* __restrict__ c,
                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                      T* __restrict__ g_out, double* sums, int nshard, long M, int C, int cw, int nrl,
                                      int rows_per_block) {
    constexpr int VEC = ElemTraits<T>::VEC;
    extern __shared__ float smem_f[];
    const int tid = threadIdx.x;
    const int cc = tid % cw, rl = tid / cw;
    const int chunk = blockIdx.x * cw + cc;
    const bool active = rl < nrl && chunk * VEC < C;
    const int ch = chunk * VEC;
    float acc[2][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[0][e] = acc[1][e] = 0.f;
    if (active) {
        float sc[VEC], sh[VEC];
        const bool masked = scale != nullptr;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            sc[e] = masked ? scale[ch + e] : 0.f;
            sh[e] = masked ? shift[ch + e] : 1.f;
        }
        const long rbeg = (long)blockIdx.y * rows_per_block;
        const long rend = rbeg + rows_per_block < M ? rbeg + rows_per_block : M;
        constexpr int UNR = 4;
        for (long m = rbeg + rl; m < rend; m += (long)nrl * UNR) {
            uint4 vg[UNR], vx[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long mm = m + (long)u * nrl;
                if (mm < rend) {
                    vg[u] = *reinterpret_cast<const uint4*>(da + mm * C + ch);
                    vx[u] = *reinterpret_cast<const uint4*>(c + mm * C + ch);
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long mm = m + (long)u * nrl;
                if (mm < rend) {
                    float g[VEC], x[VEC];
                    unpack16<T>(vg[u], g);
                    unpack16<T>(vx[u], x);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        if (!(fmaf(x[e], sc[e], sh[e]) > 0.f)) g[e] = 0.f;
                        acc[0][e] += g[e];
                        acc[1][e] = fmaf(g[e], x[e], acc[1][e]);
                    }
                    if (masked) *reinterpret_cast<uint4*>(g_out + mm * C + ch) = pack16<T>(g);
                }
            }
        }
    }
    col_commit<2, VEC>(acc, smem_f, cw, nrl, cc, rl, active, ch, C, sums, nshard);
}

// BatchNorm backward coefficients from the reduced sums (torch batch_norm_backward, train mode):
//   dx = k1*g + k2*c + k3,   dgamma += invstd*(S2 - mean*S1),   dbeta += S1
// `which` selects the S2 slot (1 = main branch, 2 = downsample branch); ns = slots per shard.
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ sums, int nshard, int ns, int which, int C,
                                       double count, const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ invstd, float* dgamma, float* dbeta,
                                       float* __restrict__ k1, float* __restrict__ k2, float* __restrict__ k3) {
    const int j = threadIdx.x % kFinLanes;  // see bn_finalize_kernel
    const int c = blockIdx.x * kFinChans + threadIdx.x / kFinLanes;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int k = j; k < nshard; k += kFinLanes) {
            s1 += sums[((long)k * ns + 0) * C + c];
            s2 += sums[((long)k * ns + which) * C + c];
        }
    s1 = fin_reduce(s1);
    s2 = fin_reduce(s2);
    if (c >= C || j != 0) return;
    const double mu = mean[c], is = invstd[c];
    const double dot = is * (s2 - mu * s1);  // sum g * xhat
    // atomics: the two views of an encoder run their backward passes on two streams and share these accumulators
    if (dgamma != nullptr) atomicAdd(dgamma + c, (float)dot);
    if (dbeta != nullptr) atomicAdd(dbeta + c, (float)s1);
    const double a = (gamma != nullptr ? (double)gamma[c] : 1.0) * is;
    const double m1 = s1 / count, m2 = dot / count;
    k1[c] = (float)a;
    k2[c] = (float)(-a * m2 * is);
    k3[c] = (float)(-a * m1 + a * m2 * is * mu);
}

constexpr int kFlatUnr = 4;  // 16-byte chunks per thread of bn_bwd_apply (one workgroup = 16 KiB per tensor)

template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ c, const float* __restrict__ k1,
                                    const float* __restrict__ k2, const float* __restrict__ k3, T* __restrict__ dc,
                                    long M, int C) {
    constexpr int VEC = ElemTraits<T>::VEC;
    constexpr int UNR = kFlatUnr;
    const int cpr = C / VEC;
    const long total = M * cpr;
    const long i0 = (long)blockIdx.x * (kThreads * UNR) + threadIdx.x;  // see bn_act_kernel
    uint4 vg[UNR], vc[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const long i = i0 + u * kThreads;
        if (i < total) {
            vg[u] = *reinterpret_cast<const uint4*>(g + i * VEC);
            vc[u] = *reinterpret_cast<const uint4*>(c + i * VEC);
        }
    }
    int ch_cur = -1;
    float q1[VEC], q2[VEC], q3[VEC];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const long i = i0 + u * kThreads;
        if (i < total) {
            const int ch = (int)(i % cpr) * VEC;
            if (ch != ch_cur) {
                ch_cur = ch;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    q1[e] = k1[ch + e];
                    q2[e] = k2[ch + e];
                    q3[e] = k3[ch + e];
                }
            }
            float a[VEC], x[VEC];
            unpack16<T>(vg[u], a);
            unpack16<T>(vc[u], x);
#pragma unroll
            for (int e = 0; e < VEC; ++e) a[e] = fmaf(q1[e], a[e], fmaf(q2[e], x[e], q3[e]));
            *reinterpret_cast<uint4*>(dc + i * VEC) = pack16<T>(a);
        }
    }
}

// column sums of [M][C] (bias gradient of the predictor's last Linear, backbone.py:30): out[c] += sum_m x
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, double* sums, int nshard, long M, int C, int cw, int nrl,
                              int rows_per_block) {
    constexpr int VEC = ElemTraits<T>::VEC;
    extern __shared__ float smem_f[];
    const int tid = threadIdx.x;
    const int cc = tid % cw, rl = tid / cw;
    const int chunk = blockIdx.x * cw + cc;
    const bool active = rl < nrl && chunk * VEC < C;
    const int ch = chunk * VEC;
    float acc[1][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[0][e] = 0.f;
    if (active) {
        const long rbeg = (long)blockIdx.y * rows_per_block;
        const long rend = rbeg + rows_per_block < M ? rbeg + rows_per_block : M;
        for (long m = rbeg + rl; m < rend; m += nrl) {
            float f[VEC];
            unpack16<T>(*reinterpret_cast<const uint4*>(x + m * C + ch), f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[0][e] += f[e];
        }
    }
    col_commit<1, VEC>(acc, smem_f, cw, nrl, cc, rl, active, ch, C, sums, nshard);
}

// Column statistics {sum x, sum x^2} of [M][C] accumulated in fp64 from the first add on.  The heads' BatchNorm1d
// (backbone.py:15,18,21,28) sees Linear outputs whose batch mean is 10..100x their batch deviation (pooled features of
// different tiles are nearly equal), so var = E[x^2] - mean^2 cancels 3-4 digits: fp32 partial sums (the GEMM
// epilogue's) would leave invstd with 1e-4 relative error there; fp64 sums make the cancellation harmless.
template <typename T>
__global__ void colstats_kernel(const T* __restrict__ x, double* sums, int nshard, long M, int C, int cw, int nrl,
                                int rows_per_block) {
    constexpr int VEC = ElemTraits<T>::VEC;
    extern __shared__ double smem_d[];
    const int tid = threadIdx.x;
    const int cc = tid % cw, rl = tid / cw;
    const int chunk = blockIdx.x * cw + cc;
    const bool active = rl < nrl && chunk * VEC < C;
    const int ch = chunk * VEC;
    double s1[VEC], s2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s1[e] = s2[e] = 0.0;
    if (active) {
        const long rbeg = (long)blockIdx.y * rows_per_block;
        const long rend = rbeg + rows_per_block < M ? rbeg + rows_per_block : M;
        for (long m = rbeg + rl; m < rend; m += nrl) {
            float f[VEC];
            unpack16<T>(*reinterpret_cast<const uint4*>(x + m * C + ch), f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const double d = (double)f[e];
                s1[e] += d;
                s2[e] = fma(d, d, s2[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        smem_d[(0 * VEC + e) * kThreads + tid] = active ? s1[e] : 0.0;
        smem_d[(1 * VEC + e) * kThreads + tid] = active ? s2[e] : 0.0;
    }
    __syncthreads();
    if (active && rl == 0) {
        double* dst = sums + (long)(blockIdx.y % nshard) * 2 * C;
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                double t = 0.0;
                for (int r = 0; r < nrl; ++r) t += smem_d[(w * VEC + e) * kThreads + r * cw + cc];
                atomicAdd(dst + (long)w * C + ch + e, t);
            }
    }
}

// bn_act with the column sums of its OUTPUT: out = relu(scale*c+shift), sums[c] += sum_m out  (the folded bn3
// backward needs sum_p a2, see fold_weights_kernel; one pass instead of bn_act + colsum)
template <typename T>
__global__ void bn_act_sum_kernel(const T* __restrict__ c, const float* __restrict__ scale,
                                  const float* __restrict__ shift, T* __restrict__ out, double* sums, int nshard,
                                  long M, int C, int cw, int nrl, int rows_per_block) {
    constexpr int VEC = ElemTraits<T>::VEC;
    extern __shared__ float smem_f[];
    const int tid = threadIdx.x;
    const int cc = tid % cw, rl = tid / cw;
    const int chunk = blockIdx.x * cw + cc;
    const bool active = rl < nrl && chunk * VEC < C;
    const int ch = chunk * VEC;
    float acc[1][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[0][e] = 0.f;
    if (active) {
        float sc[VEC], sh[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            sc[e] = scale[ch + e];
            sh[e] = shift[ch + e];
        }
        const long rbeg = (long)blockIdx.y * rows_per_block;
        const long rend = rbeg + rows_per_block < M ? rbeg + rows_per_block : M;
        constexpr int UNR = 4;  // rows in flight per thread (memory-level parallelism)
        for (long m = rbeg + rl; m < rend; m += (long)nrl * UNR) {
            uint4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long mm = m + (long)u * nrl;
                if (mm < rend) v[u] = *reinterpret_cast<const uint4*>(c + mm * C + ch);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long mm = m + (long)u * nrl;
                if (mm < rend) {
                    float f[VEC];
                    unpack16<T>(v[u], f);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        f[e] = round_to<T>(fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f));
                        acc[0][e] += f[e];
                    }
                    *reinterpret_cast<uint4*>(out + mm * C + ch) = pack16<T>(f);
                }
            }
        }
    }
    col_commit<1, VEC>(acc, smem_f, cw, nrl, cc, rl, active, ch, C, sums, nshard);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm backward folded into the weights of the 1x1 conv that produced its input (Bottleneck conv3 + bn3,
// src/models/resnet.py:115-117): with c = W a the terms of  dc = k1*g + k2*c + k3  that depend on c reduce to
// small [K][C] / [C][C] matrices, so the 4x-wide c never has to be kept, re-made or re-read in backward:
//   sum_p g*c [k]   = sum_c W[k][c] * M[k][c]                     M = g^T a      (fold_dots)
//   dW              = k1 o M + k2 o (W A) + k3 (x) sa             A = a^T a, sa = sum_p a   (fold_weights)
//   W^T dc          = (k1 o W)^T g + (W^T diag(k2) W) a + W^T k3
// ---------------------------------------------------------------------------------------------
__global__ void fold_dots_kernel(const float* __restrict__ W, const float* __restrict__ Mm, double* __restrict__ out,
                                 int K, int C) {
    // one wave per output channel k
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= K) return;
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int c = lane; c < C; c += 64) acc += (double)W[(long)k * C + c] * (double)Mm[(long)k * C + c];
    acc = wave_sum_d(acc);
    if (lane == 0) out[k] = acc;
}

// out[k] = sum_c W[k][c] * v[c]  (fp64; sum over pixels of c = W a from the column sums of a)
__global__ void fold_matvec_kernel(const float* __restrict__ W, const double* __restrict__ v, double* __restrict__ out,
                                   int K, int C) {
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= K) return;
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int c = lane; c < C; c += 64) acc += (double)W[(long)k * C + c] * v[c];
    acc = wave_sum_d(acc);
    if (lane == 0) out[k] = acc;
}

// out[k][0:C1] = s1[k]*W1[k][:],  out[k][C1:C1+C2] = s2[k]*W2[k][:],  shift[k] = b1[k] + b2[k]: two BatchNorms folded
// into the weight rows of the two 1x1 convs they follow (Bottleneck tail + downsample branch in one GEMM)
__global__ void row_scale_cat_kernel(const float* __restrict__ W1, const float* __restrict__ s1, int C1,
                                     const float* __restrict__ W2, const float* __restrict__ s2, int C2,
                                     const float* __restrict__ b1, const float* __restrict__ b2,
                                     float* __restrict__ out, float* __restrict__ shift, int K) {
    const int k = blockIdx.x;
    const int Ct = C1 + C2;
    const float a = s1[k], b = s2[k];
    for (int c = threadIdx.x; c < Ct; c += blockDim.x)
        out[(long)k * Ct + c] = c < C1 ? a * W1[(long)k * C1 + c] : b * W2[(long)k * C2 + (c - C1)];
    if (threadIdx.x == 0) shift[k] = b1[k] + b2[k];
}

constexpr int kFoldRows = 16;
__global__ void fold_weights_kernel(const float* __restrict__ W, const float* __restrict__ Mm,
                                    const float* __restrict__ WA, const float* __restrict__ k1,
                                    const float* __restrict__ k2, const float* __restrict__ k3,
                                    const double* __restrict__ sa, float* __restrict__ dW, float* __restrict__ Wk1,
                                    float* __restrict__ Wk2, double* bvec64, int K, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int kb = blockIdx.y * kFoldRows;
    const int ke = kb + kFoldRows < K ? kb + kFoldRows : K;
    const float s = (float)sa[c];
    for (int k = kb; k < ke; ++k) {
        const long o = (long)k * C + c;
        const float w = W[o], a = k1[k], b = k2[k], d = k3[k];
        atomicAdd(dW + o, fmaf(a, Mm[o], fmaf(b, WA[o], d * s)));  // shared with the other view's stream
        Wk1[o] = a * w;
        Wk2[o] = b * w;
    }
    // bvec64[c] += sum_k k3[k] W[k][c]: this block's 16 rows in row order (fp32, fixed), the blocks' partial sums added in
    // fp64 -- their arrival order then moves the sum by ~1e-16, invisible once the caller rounds it to fp32
    // (msfwsi_add_f64_to_f32), where fp32 atomics moved the folded backward's bias by 1e-7 from run to run.  (One block
    // walking the whole column in a fixed order was measured 2.7 ms/step slower: 2048 dependent loads in one wave.)
    float bacc = 0.f;
    for (int k = kb; k < ke; ++k) bacc = fmaf(k3[k], W[(long)k * C + c], bacc);
    atomicAdd(bvec64 + c, (double)bacc);
}

__global__ void add_f64_kernel(const double* __restrict__ in, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += in[i];
}

__global__ void add_f64_to_f32_kernel(const double* __restrict__ in, float* out, int n, float alpha) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(out + i, alpha * (float)in[i]);
}

// ---------------------------------------------------------------------------------------------
// jigsaw row gather / scatter (backbone.py:147-158): out[b*K+k] = in[b*K+idx[b][k]]
// scatter=1 is its adjoint: out[b*K+idx[b][k]] (+)= in[b*K+k]   (idx rows are permutations)
// ---------------------------------------------------------------------------------------------
// strided pixel subsampling of an NHWC tensor and its adjoint (the operand of a stride-s 1x1 conv as a dense tensor,
// and the zero-stuffed full-resolution gradient of that operand):
//   expand == 0: lo[n][p][q][:] = full[n][p*s][q*s][:]            (one thread per 16-byte chunk of lo)
//   expand == 1: full[n][h][w][:] = (h%s == 0 && w%s == 0) ? lo[n][h/s][w/s][:] : 0   (one thread per chunk of full)
template <typename T>
__global__ void pixel_stride_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W, int P, int Q,
                                    int C, int s, int expand) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int cpr = C / VEC;
    const long total = (long)N * (expand ? (long)H * W : (long)P * Q) * cpr;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % cpr);
        long pix = i / cpr;
        if (!expand) {
            const int q = (int)(pix % Q);
            pix /= Q;
            const int p = (int)(pix % P);
            const long n = pix / P;
            const long src = ((n * H + (long)p * s) * W + (long)q * s) * C + cv * VEC;
            *reinterpret_cast<uint4*>(out + i * VEC) = *reinterpret_cast<const uint4*>(in + src);
        } else {
            const int w = (int)(pix % W);
            pix /= W;
            const int h = (int)(pix % H);
            const long n = pix / H;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (h % s == 0 && w % s == 0 && h / s < P && w / s < Q)
                v = *reinterpret_cast<const uint4*>(in + ((n * P + h / s) * Q + w / s) * C + cv * VEC);
            *reinterpret_cast<uint4*>(out + i * VEC) = v;
        }
    }
}

template <typename T>
__global__ void rows_permute_kernel(const T* __restrict__ in, const long* __restrict__ idx, T* __restrict__ out,
                                    int B, int K, int C, int scatter, int accumulate) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int cpr = C / VEC;
    const long total = (long)B * K * cpr;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % cpr);
        const long row = i / cpr;
        const long b = row / K;
        const long j = idx[row];
        const long other = b * K + j;
        const long srow = scatter ? row : other, drow = scatter ? other : row;
        uint4 v = *reinterpret_cast<const uint4*>(in + srow * C + cv * VEC);
        if (accumulate) {
            float a[VEC], d[VEC];
            unpack16<T>(v, a);
            unpack16<T>(*reinterpret_cast<const uint4*>(out + drow * C + cv * VEC), d);
#pragma unroll
            for (int e = 0; e < VEC; ++e) a[e] += d[e];
            v = pack16<T>(a);
        }
        *reinterpret_cast<uint4*>(out + drow * C + cv * VEC) = v;
    }
}

// strided 2-D copy / accumulate (fuser concat and its adjoint, backbone.py:195-202)
template <typename T>
__global__ void copy2d_kernel(const T* __restrict__ src, long src_ld, T* __restrict__ dst, long dst_ld, long rows,
                              int cols, int accumulate) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int cpr = cols / VEC;
    const long total = rows * cpr;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % cpr);
        const long r = i / cpr;
        uint4 v = *reinterpret_cast<const uint4*>(src + r * src_ld + cv * VEC);
        if (accumulate) {
            float a[VEC], d[VEC];
            unpack16<T>(v, a);
            unpack16<T>(*reinterpret_cast<const uint4*>(dst + r * dst_ld + cv * VEC), d);
#pragma unroll
            for (int e = 0; e < VEC; ++e) a[e] += d[e];
            v = pack16<T>(a);
        }
        *reinterpret_cast<uint4*>(dst + r * dst_ld + cv * VEC) = v;
    }
}

inline unsigned stream_grid(long total) {
    long b = (total + kThreads - 1) / kThreads;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// one workgroup per kFlatUnr*256 chunks (no grid-stride loop)
inline unsigned flat_grid(long total) {
    const long per = (long)kThreads * kFlatUnr;
    return (unsigned)((total + per - 1) / per);
}

inline bool dtype_ok(int dt) { return msfwsi_dtype_ok(dt); }
inline int vec_of(int dt) { return msfwsi_vec_of(dt); }

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int msfwsi_nchw_to_nhwc(int dtype, const float* x, void* y, int N, int C, int H, int W, int CP,
                                   void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && x && y && N > 0 && C > 0 && H > 0 && W > 0 && CP >= C);
    MSFWSI_CHECK_ARG(CP % vec_of(dtype) == 0);
    const long total = (long)N * H * W * (CP / vec_of(dtype));
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(stream_grid(total)), dim3(kThreads), 0, ST(stream), x,
                           (T*)y, N, C, H * W, CP));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_nchw_to_s2d(int dtype, const float* x, void* y, int N, int H, int W, void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && x && y && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0);
    const long total = (long)N * (H / 2) * (W / 2) * (16 / vec_of(dtype));
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(nchw_to_s2d_kernel<T>, dim3(stream_grid(total)), dim3(kThreads), 0, ST(stream), x,
                           (T*)y, N, H, W));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_stem_s2d_weights(int dtype, const float* w, void* out, int K, void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && w && out && K > 0);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(stem_s2d_weights_kernel<T>, dim3((K * 256 + 255) / 256), dim3(256), 0, ST(stream),
                           w, (T*)out, K));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_stem_s2d_wfold(const float* dw2, float* dw, int K, void* stream) {
    MSFWSI_CHECK_ARG(dw2 && dw && K > 0);
    hipLaunchKernelGGL(stem_s2d_wfold_kernel, dim3((K * 147 + 255) / 256), dim3(256), 0, ST(stream), dw2, dw, K);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_bn_finalize(const double* sums, int nshard, int C, double count, const float* gamma,
                                  const float* beta, float eps, float momentum, float* running_mean,
                                  float* running_var, long* num_batches_tracked, float* scale, float* shift,
                                  float* mean, float* invstd, void* stream) {
    MSFWSI_CHECK_ARG(sums && nshard >= 1 && C > 0 && count > 0 && scale && shift && mean && invstd);
    MSFWSI_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + kFinChans - 1) / kFinChans), dim3(256), 0, ST(stream), sums, nshard, C, count,
                       gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, scale, shift, mean,
                       invstd);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_bn_eval_coeffs(const float* running_mean, const float* running_var, const float* gamma,
                                     const float* beta, float eps, int C, float* scale, float* shift, float* mean,
                                     float* invstd, void* stream) {
    MSFWSI_CHECK_ARG(running_mean && running_var && C > 0 && scale && shift && mean && invstd);
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 127) / 128), dim3(128), 0, ST(stream), running_mean, running_var,
                       gamma, beta, eps, C, scale, shift, mean, invstd);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_shard_sum(const double* in, int nshard, int n, double* out, void* stream) {
    MSFWSI_CHECK_ARG(in && out && nshard >= 1 && n > 0);
    hipLaunchKernelGGL(shard_sum_kernel, dim3((n + kFinChans - 1) / kFinChans), dim3(256), 0, ST(stream), in, nshard, n, out);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_bn_act(int dtype, const void* c, const float* scale, const float* shift, const void* ident,
                             const float* id_scale, const float* id_shift, int relu, void* out, long M, int C,
                             void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && c && scale && shift && out && M > 0 && C > 0 && C % vec_of(dtype) == 0);
    MSFWSI_CHECK_ARG((id_scale == nullptr) == (id_shift == nullptr));
    const long total = M * (C / vec_of(dtype));
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(bn_act_kernel<T>, dim3(stream_grid(total)), dim3(kThreads), 0, ST(stream),
                           (const T*)c, scale, shift, (const T*)ident, id_scale, id_shift, relu, (T*)out,
                           M, C));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_bn_act_sum(int dtype, const void* c, const float* scale, const float* shift, void* out,
                                 double* sums, int nshard, long M, int C, void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && c && scale && shift && out && sums && nshard >= 1 && M > 0);
    MSFWSI_CHECK_ARG(C % vec_of(dtype) == 0);
    const int vec = vec_of(dtype);
    ColGrid cg = make_col_grid(M, C, vec, 2048);
    const size_t lds = (size_t)kThreads * vec * sizeof(float);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(bn_act_sum_kernel<T>, cg.grid, dim3(kThreads), lds, ST(stream), (const T*)c,
                                            scale, shift, (T*)out, sums, nshard, M, C, cg.cw, cg.nrl,
                                            cg.rows_per_block));
    return msfwsi_launch_status();
}

static msfwsi_tunable g_pool_bwd_walk{1};  // msfwsi_set_tuning(16, .): 0 = the per-window / per-pixel max-pool kernels (A/B); 1 = column walk
extern "C" __attribute__((visibility("hidden"))) long msfwsi_pool_bwd_set_walk(long v, int write) {
    const long old = g_pool_bwd_walk;
    if (write) g_pool_bwd_walk = v;
    return old;
}

extern "C" int msfwsi_stem_pool_fwd(int dtype, const void* c0, const float* scale, const float* shift, void* out,
                                    unsigned char* argmax, int N, int H, int W, int C, void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && c0 && scale && shift && out && argmax && N > 0 && H > 1 && W > 1);
    MSFWSI_CHECK_ARG(C % vec_of(dtype) == 0);
    const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
    {   // column walk: each input row loaded once per output column (see stem_pool_fwd_walk_kernel)
        const int cpp = C / vec_of(dtype);
        if (g_pool_bwd_walk && cpp >= 1 && cpp <= 64 && (cpp & (cpp - 1)) == 0) {
            const int cols = 64 / cpp;
            const int qgroups = (Q + cols - 1) / cols;
            const long ntask = (long)N * qgroups;
            const long nblk = (ntask + 3) / 4;
            if (nblk > 0x7fffffffL) return MSFWSI_EINVAL;
            MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(stem_pool_fwd_walk_kernel<T>, dim3((unsigned)nblk), dim3(256), 0,
                                   ST(stream), (const T*)c0, scale, shift, (T*)out, argmax, N, H, W, C, P, Q, cpp, qgroups,
                                   ntask));
            return msfwsi_launch_status();
        }
    }
    const long total = (long)N * P * Q * (C / vec_of(dtype));
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(stem_pool_fwd_kernel<T>, dim3(stream_grid(total)), dim3(kThreads), 0, ST(stream),
                           (const T*)c0, scale, shift, (T*)out, argmax, N, H, W, C, P, Q));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_stem_pool_bwd(int dtype, const void* dp, const unsigned char* argmax, const void* c0,
                                    const float* scale, const float* shift, void* g0, double* sums, int nshard,
                                    const float* k1, const float* k2, const float* k3, const void* dact, int N, int H,
                                    int W, int C, void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && dp && argmax && c0 && scale && shift && (g0 || sums) && nshard >= 1);
    MSFWSI_CHECK_ARG((k1 == nullptr) == (k2 == nullptr) && (k1 == nullptr) == (k3 == nullptr));
    MSFWSI_CHECK_ARG(k1 == nullptr || g0 != nullptr);
    MSFWSI_CHECK_ARG(N > 0 && H > 1 && W > 1 && C % vec_of(dtype) == 0 && (long)N * H * W <= 0x7fffffffL);
    const int P = (H + 2 - 3) / 2 + 1, Q = (W + 2 - 3) / 2 + 1;
    const int vec = vec_of(dtype);
    const size_t lds = (size_t)kThreads * 2 * vec * sizeof(float);
    // column walk (see stem_pool_bwd_walk_kernel): the training path's form -- gated gradient + sums, chunks per pixel a
    // power of two <= 64 (the stem: 64 channels = 8 chunks of 16-bit / 16 of fp32)
    const int cpp = C / vec;
    if (g_pool_bwd_walk && k1 == nullptr && dact == nullptr && cpp >= 1 && cpp <= 64 && (cpp & (cpp - 1)) == 0 &&
        C <= 512) {
        const int cols = 64 / cpp;
        const int qgroups = (Q + cols - 1) / cols;
        const long ntask = (long)N * qgroups;
        const long nblk = (ntask + 3) / 4;
        if (nblk > 0x7fffffffL) return MSFWSI_EINVAL;
        MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(stem_pool_bwd_walk_kernel<T>, dim3((unsigned)nblk), dim3(256), 0, ST(stream),
                               (const T*)dp, argmax, (const T*)c0, scale, shift, (T*)g0, sums, nshard, N, H, W, C, P, Q,
                               cpp, qgroups, ntask));
        return msfwsi_launch_status();
    }
    ColGrid g = make_col_grid((long)N * H * W, C, vec, 2048);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(stem_pool_bwd_kernel<T>, g.grid, dim3(kThreads), lds, ST(stream), (const T*)dp,
                           argmax, (const T*)c0, scale, shift, (T*)g0, sums, nshard, k1, k2, k3, (const T*)dact, N, H, W, C,
                           P, Q, g.cw, g.nrl, g.rows_per_block));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_gap_fwd(int dtype, const void* y, void* out, int N, int HW, int C, void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && y && out && N > 0 && HW > 0 && C % vec_of(dtype) == 0);
    MSFWSI_CHECK_ARG(N <= 65535);
    const int vec = vec_of(dtype);
    const int cpr = C / vec;
    const int cw = cpr < kThreads ? cpr : kThreads;
    int nrl = kThreads / cw;
    if (nrl > HW) nrl = HW;
    const dim3 grid((unsigned)((cpr + cw - 1) / cw), (unsigned)N);
    const size_t lds = (size_t)kThreads * vec * sizeof(float);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(gap_fwd_kernel<T>, grid, dim3(kThreads), lds, ST(stream), (const T*)y,
                           (T*)out, N, HW, C, cw, nrl));
    return msfwsi_launch_status();
}

// msfwsi_gap_fwd + msfwsi_pixel_stride(stride 2, expand 0) in one pass over y [N][H][W][C]: out = the pooled features, sout
// [N][ceil(H/2)][ceil(W/2)][C] = y[:, ::2, ::2, :]
extern "C" int msfwsi_gap_fwd_stride2(int dtype, const void* y, void* out, void* sout, int N, int H, int W, int C,
                                      void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && y && out && sout && N > 0 && H > 0 && W > 0 && C % vec_of(dtype) == 0);
    MSFWSI_CHECK_ARG(N <= 65535 && (long)H * W <= 0x7fffffffL);
    const int vec = vec_of(dtype);
    const int cpr = C / vec;
    const int cw = cpr < kThreads ? cpr : kThreads;
    const int HW = H * W;
    int nrl = kThreads / cw;
    if (nrl > HW) nrl = HW;
    const dim3 grid((unsigned)((cpr + cw - 1) / cw), (unsigned)N);
    const size_t lds = (size_t)kThreads * vec * sizeof(float);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(gap_fwd_kernel<T>, grid, dim3(kThreads), lds, ST(stream), (const T*)y,
                           (T*)out, N, HW, C, cw, nrl, (T*)sout, W, make_fastdiv((unsigned)W)));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_block_end_bwd(int dtype, const void* dy, const void* y, const void* gapg, float gap_scale,
                                    const void* c_main, const void* c_ds, void* g, double* sums, int nshard, long M,
                                    int HW, int C, void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && y && g && sums && nshard >= 1 && M > 0 && HW > 0);
    MSFWSI_CHECK_ARG(C % vec_of(dtype) == 0 && (dy != nullptr || gapg != nullptr));
    const int vec = vec_of(dtype);
    ColGrid cg = make_col_grid(M, C, vec, 2048);
    const size_t lds = (size_t)kThreads * 3 * vec * sizeof(float);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(block_end_bwd_kernel<T>, cg.grid, dim3(kThreads), lds, ST(stream), (const T*)dy,
                           (const T*)y, (const T*)gapg, gap_scale, (const T*)c_main,
                           (const T*)c_ds, (T*)g, sums, nshard, M, HW, C, cg.cw, cg.nrl, cg.rows_per_block));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_act_bwd_reduce(int dtype, const void* da, const void* c, const float* scale,
                                     const float* shift, void* g, double* sums, int nshard, long M, int C,
                                     void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && da && c && sums && nshard >= 1 && M > 0 && C % vec_of(dtype) == 0);
    MSFWSI_CHECK_ARG((scale == nullptr) == (shift == nullptr));
    MSFWSI_CHECK_ARG(scale == nullptr || g != nullptr);
    const int vec = vec_of(dtype);
    ColGrid cg = make_col_grid(M, C, vec, 2048);
    const size_t lds = (size_t)kThreads * 2 * vec * sizeof(float);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(act_bwd_reduce_kernel<T>, cg.grid, dim3(kThreads), lds, ST(stream), (const T*)da,
                           (const T*)c, scale, shift, (T*)g, sums, nshard, M, C, cg.cw, cg.nrl,
                           cg.rows_per_block));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_bn_bwd_finalize(const double* sums, int nshard, int nslots, int which, int C, double count,
                                      const float* gamma, const float* mean, const float* invstd, float* dgamma,
                                      float* dbeta, float* k1, float* k2, float* k3, void* stream) {
    MSFWSI_CHECK_ARG(sums && nshard >= 1 && nslots >= 2 && which >= 1 && which < nslots && C > 0 && count > 0);
    MSFWSI_CHECK_ARG(mean && invstd && k1 && k2 && k3);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + kFinChans - 1) / kFinChans), dim3(256), 0, ST(stream), sums, nshard, nslots,
                       which, C, count, gamma, mean, invstd, dgamma, dbeta, k1, k2, k3);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_bn_bwd_apply(int dtype, const void* g, const void* c, const float* k1, const float* k2,
                                   const float* k3, void* dc, long M, int C, void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && g && c && k1 && k2 && k3 && dc && M > 0 && C % vec_of(dtype) == 0);
    const long total = M * (C / vec_of(dtype));
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(flat_grid(total)), dim3(kThreads), 0, ST(stream),
                           (const T*)g, (const T*)c, k1, k2, k3, (T*)dc, M, C));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_colsum(int dtype, const void* x, double* sums, int nshard, long M, int C, void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && x && sums && nshard >= 1 && M > 0 && C % vec_of(dtype) == 0);
    const int vec = vec_of(dtype);
    ColGrid cg = make_col_grid(M, C, vec, 1024);
    const size_t lds = (size_t)kThreads * vec * sizeof(float);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(colsum_kernel<T>, cg.grid, dim3(kThreads), lds, ST(stream), (const T*)x, sums,
                           nshard, M, C, cg.cw, cg.nrl, cg.rows_per_block));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_colstats(int dtype, const void* x, double* sums, int nshard, long M, int C, void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && x && sums && nshard >= 1 && M > 0 && C % vec_of(dtype) == 0);
    const int vec = vec_of(dtype);
    ColGrid cg = make_col_grid(M, C, vec, 512);
    const size_t lds = (size_t)kThreads * 2 * vec * sizeof(double);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(colstats_kernel<T>, cg.grid, dim3(kThreads), lds, ST(stream), (const T*)x, sums,
                           nshard, M, C, cg.cw, cg.nrl, cg.rows_per_block));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_fold_dots(const float* W, const float* M, double* out, int K, int C, void* stream) {
    MSFWSI_CHECK_ARG(W && M && out && K > 0 && C > 0);
    hipLaunchKernelGGL(fold_dots_kernel, dim3((K + 3) / 4), dim3(256), 0, ST(stream), W, M, out, K, C);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_fold_matvec(const float* W, const double* v, double* out, int K, int C, void* stream) {
    MSFWSI_CHECK_ARG(W && v && out && K > 0 && C > 0);
    hipLaunchKernelGGL(fold_matvec_kernel, dim3((K + 3) / 4), dim3(256), 0, ST(stream), W, v, out, K, C);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_row_scale_cat(const float* W1, const float* s1, int C1, const float* W2, const float* s2, int C2,
                                    const float* b1, const float* b2, float* out, float* shift, int K, void* stream) {
    MSFWSI_CHECK_ARG(W1 && s1 && W2 && s2 && b1 && b2 && out && shift && C1 > 0 && C2 > 0 && K > 0);
    hipLaunchKernelGGL(row_scale_cat_kernel, dim3(K), dim3(256), 0, ST(stream), W1, s1, C1, W2, s2, C2, b1, b2, out,
                       shift, K);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_fold_weights(const float* W, const float* M, const float* WA, const float* k1, const float* k2,
                                   const float* k3, const double* sa, float* dW, float* Wk1, float* Wk2, double* bvec,
                                   int K, int C, void* stream) {
    MSFWSI_CHECK_ARG(W && M && WA && k1 && k2 && k3 && sa && dW && Wk1 && Wk2 && bvec && K > 0 && C > 0);
    hipLaunchKernelGGL(fold_weights_kernel, dim3((C + 63) / 64, (K + kFoldRows - 1) / kFoldRows), dim3(64), 0,
                       ST(stream), W, M, WA, k1, k2, k3, sa, dW, Wk1, Wk2, bvec, K, C);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_add_f64(const double* in, double* out, int n, void* stream) {
    MSFWSI_CHECK_ARG(in && out && n > 0);
    hipLaunchKernelGGL(add_f64_kernel, dim3((n + 255) / 256), dim3(256), 0, ST(stream), in, out, n);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_add_f64_to_f32(const double* in, float* out, int n, float alpha, void* stream) {
    MSFWSI_CHECK_ARG(in && out && n > 0);
    hipLaunchKernelGGL(add_f64_to_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, ST(stream), in, out, n, alpha);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_rows_permute(int dtype, const void* in, const long* idx, void* out, int B, int K, int C,
                                   int scatter, int accumulate, void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && in && idx && out && B > 0 && K > 0 && C % vec_of(dtype) == 0);
    const long total = (long)B * K * (C / vec_of(dtype));
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(rows_permute_kernel<T>, dim3(stream_grid(total)), dim3(kThreads), 0, ST(stream),
                           (const T*)in, idx, (T*)out, B, K, C, scatter, accumulate));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_pixel_stride(int dtype, const void* in, void* out, int N, int H, int W, int C, int stride,
                                   int expand, void* stream) {
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && in && out && N > 0 && H > 0 && W > 0 && stride >= 1);
    MSFWSI_CHECK_ARG(C % vec_of(dtype) == 0);
    const int P = (H - 1) / stride + 1, Q = (W - 1) / stride + 1;
    const long total = (long)N * (expand ? (long)H * W : (long)P * Q) * (C / vec_of(dtype));
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(pixel_stride_kernel<T>, dim3(stream_grid(total)), dim3(kThreads), 0,
                                            ST(stream), (const T*)in, (T*)out, N, H, W, P, Q, C, stride, expand));
    return msfwsi_launch_status();
}

extern "C" int msfwsi_copy2d(int dtype, const void* src, long src_ld, void* dst, long dst_ld, long rows, int cols,
                             int accumulate, void* stream) {
    const int vec = vec_of(dtype);
    MSFWSI_CHECK_ARG(dtype_ok(dtype) && src && dst && rows > 0 && cols > 0 && cols % vec == 0);
    MSFWSI_CHECK_ARG(src_ld % vec == 0 && dst_ld % vec == 0);
    const long total = rows * (cols / vec);
    MSFWSI_WITH_T(dtype, hipLaunchKernelGGL(copy2d_kernel<T>, dim3(stream_grid(total)), dim3(kThreads), 0, ST(stream),
                           (const T*)src, src_ld, (T*)dst, dst_ld, rows, cols, accumulate));
    return msfwsi_launch_status();
}
