// Weight-gradient GEMM for conv / linear layers on gfx950:
//
//   dW[co][(r,s,ci)] += sum_m dY[m][co] * act(X[img(m), p*stride-pad+r, q*stride-pad+s, ci])
//
// (the wgrad half of autograd's conv2d/linear backward that the reference reaches through
// `scaler.scale(loss).backward()`, tools/ssl_train.py:472).  The reduction runs over output pixels m,
// so both operand tiles are staged in their natural [m][channel] layout and the MFMA fragments are
// fetched with the gfx950 transposed LDS read (ds_read_b64_tr_b16) for bf16, plain b32 reads for fp32.
// `act` is the producer's BatchNorm+ReLU recomputed on the fly from the saved raw conv output, so the
// normalised activation is never stored.  Split over m across workgroups; partial tiles are added
// with fp32 global atomics (one accumulator register = two 128-byte row segments per wave).
#include "common.h"
#include "../../include/msfwsi_hip.h"

namespace {

struct WgradParams {
    const void* x;
    const void* dy;
    float* dw;
    const float* pro_scale;
    const float* pro_shift;
    int N, H, W, C;
    int P, Q, K;
    int R, S, stride, pad;
    int M, Jtot;
    int rows_per_split;
    int ntile_i;
};

template <typename T, int BI, int BJ>
struct WgradCfg {
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int BKM = ElemTraits<T>::BK;  // pixels per stage
    static constexpr int LDI = (sizeof(T) == 2) ? (BI + 32) : (BI + 4);
    static constexpr int LDJ = (sizeof(T) == 2) ? (BJ + 32) : (BJ + 4);
    static constexpr int WI = (BI >= 128 || BJ <= 64) ? 2 : 1;  // waves along co
    static constexpr int WJ = 4 / WI;
    static constexpr int TI = BI / WI / 32;
    static constexpr int TJ = BJ / WJ / 32;
    static constexpr int A_CHUNKS = BKM * (BI / VEC) / 256;
    static constexpr int B_CHUNKS = BKM * (BJ / VEC) / 256;
    static constexpr int A_BYTES = BKM * LDI * (int)sizeof(T);
    static constexpr int B_BYTES = BKM * LDJ * (int)sizeof(T);
    static constexpr int LDS_BYTES = 2 * (A_BYTES + B_BYTES);
    static_assert(TI >= 1 && TJ >= 1, "tile too small");
    static_assert(A_CHUNKS >= 1 && B_CHUNKS >= 1, "tile too small");
};

template <typename T>
struct WFrag;
template <>
struct WFrag<float> {
    typedef f32x4 type;
};
template <>
struct WFrag<__bf16> {
    typedef bf16x8 type;
};

// fragment of a natural-layout [k][col] LDS tile: 32 columns starting at col0, one k-group `ks`
template <typename T, int LD>
__device__ __forceinline__ typename WFrag<T>::type read_tr_frag(const T* tile, int ks, int col0, int lane) {
    typedef typename WFrag<T>::type frag_t;
    if constexpr (sizeof(T) == 2) {
        const int li = lane & 15, G = lane >> 4;
        const int q = li >> 2, p = li & 3;
        const T* a0 = tile + (ks * 16 + (G >> 1) * 8 + q) * LD + col0 + (G & 1) * 16 + p * 4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * LD));
        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(frag_t, both);
    } else {
        frag_t t;
        const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = reinterpret_cast<const float*>(tile)[(ks * 8 + lh * 4 + e) * LD + col0 + l31];
        return t;
    }
}

template <typename T, int BI, int BJ>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams prm) {
    typedef WgradCfg<T, BI, BJ> Cfg;
    constexpr int VEC = Cfg::VEC, BKM = Cfg::BKM, LDI = Cfg::LDI, LDJ = Cfg::LDJ;
    constexpr int WI = Cfg::WI, TI = Cfg::TI, TJ = Cfg::TJ;
    constexpr int A_CHUNKS = Cfg::A_CHUNKS, B_CHUNKS = Cfg::B_CHUNKS;
    constexpr int CPI = BI / VEC, CPJ = BJ / VEC;
    typedef typename WFrag<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* As = reinterpret_cast<T*>(smem);                     // [2][BKM][LDI]  dY tile
    T* Bs = reinterpret_cast<T*>(smem + 2 * Cfg::A_BYTES);  // [2][BKM][LDJ]  activation tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave % WI, wj = wave / WI;
    const int l31 = lane & 31, lh = lane >> 5;

    const int tile_i = blockIdx.x % prm.ntile_i;
    const int tile_j = blockIdx.x / prm.ntile_i;
    const int i0 = tile_i * BI, j0 = tile_j * BJ;
    const int mbeg = blockIdx.y * prm.rows_per_split;
    const int mend = min(prm.M, mbeg + prm.rows_per_split);

    const T* __restrict__ x = reinterpret_cast<const T*>(prm.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(prm.dy);
    const int PQ = prm.P * prm.Q;
    const bool has_pro = prm.pro_scale != nullptr;

    // this thread's fixed activation column chunk: j -> (r, s, ci)
    const int cj = tid % CPJ;
    const int jcol = j0 + cj * VEC;
    const bool j_ok = jcol < prm.Jtot;
    int fr = 0, fs = 0, fc = 0;
    if (j_ok) {
        const int rs = jcol / prm.C;
        fc = jcol - rs * prm.C;
        fr = rs / prm.S;
        fs = rs - fr * prm.S;
    }
    float psc[VEC], psh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        psc[e] = 1.f;
        psh[e] = 0.f;
    }
    if (has_pro && j_ok) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            psc[e] = prm.pro_scale[fc + e];
            psh[e] = prm.pro_shift[fc + e];
        }
    }
    const int ci_a = tid % CPI;  // dY column chunk
    const bool i_ok = (i0 + ci_a * VEC) < prm.K;

    uint4 a_reg[A_CHUNKS], b_reg[B_CHUNKS];
    bool b_ok[B_CHUNKS];

    auto load_global = [&](int mb) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            const int krow = tid / CPI + i * (256 / CPI);
            const int m = mb + krow;
            a_reg[i] = make_uint4(0, 0, 0, 0);
            if (m < mend && i_ok) a_reg[i] = *reinterpret_cast<const uint4*>(dy + (long)m * prm.K + i0 + ci_a * VEC);
        }
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) {
            const int krow = tid / CPJ + i * (256 / CPJ);
            const int m = mb + krow;
            b_reg[i] = make_uint4(0, 0, 0, 0);
            bool ok = (m < mend) && j_ok;
            if (ok) {
                const int img = m / PQ;
                const int rem = m - img * PQ;
                const int p = rem / prm.Q;
                const int q = rem - p * prm.Q;
                const int h = p * prm.stride - prm.pad + fr;
                const int w = q * prm.stride - prm.pad + fs;
                ok = (unsigned)h < (unsigned)prm.H && (unsigned)w < (unsigned)prm.W;
                if (ok) b_reg[i] = *reinterpret_cast<const uint4*>(x + (((long)img * prm.H + h) * prm.W + w) * prm.C + fc);
            }
            b_ok[i] = ok;
        }
    };
    auto store_lds = [&](int buf) {
        T* Ab = As + buf * (BKM * LDI);
        T* Bb = Bs + buf * (BKM * LDJ);
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            const int krow = tid / CPI + i * (256 / CPI);
            *reinterpret_cast<uint4*>(Ab + krow * LDI + ci_a * VEC) = a_reg[i];
        }
#pragma unroll
        for (int i = 0; i < B_CHUNKS; ++i) {
            const int krow = tid / CPJ + i * (256 / CPJ);
            uint4 v = b_reg[i];
            if (has_pro && b_ok[i]) {
                float f[VEC];
                unpack16<T>(v, f);
#pragma unroll
                for (int e = 0; e < VEC; ++e) f[e] = fmaxf(fmaf(f[e], psc[e], psh[e]), 0.f);
                v = pack16<T>(f);
            }
            *reinterpret_cast<uint4*>(Bb + krow * LDJ + cj * VEC) = v;
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

    auto compute = [&](int buf) {
        const T* Ab = As + buf * (BKM * LDI);
        const T* Bb = Bs + buf * (BKM * LDJ);
#pragma unroll
        for (int ks = 0; ks < BKM / (2 * VEC); ++ks) {
            frag_t af[TI], bf[TJ];
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) af[ti] = read_tr_frag<T, LDI>(Ab, ks, (wi * TI + ti) * 32, lane);
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj) bf[tj] = read_tr_frag<T, LDJ>(Bb, ks, (wj * TJ + tj) * 32, lane);
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                for (int tj = 0; tj < TJ; ++tj) {
                    if constexpr (sizeof(T) == 2) {
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ti], bf[tj], acc[ti][tj], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ti][e], bf[tj][e], acc[ti][tj], 0, 0, 0);
                    }
                }
        }
    };

    const int nk = (mend - mbeg + BKM - 1) / BKM;
    if (nk <= 0) return;
    load_global(mbeg);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_global(mbeg + (kt + 1) * BKM);
        compute(buf);
        if (kt + 1 < nk) store_lds(buf ^ 1);
        __syncthreads();
    }

    // epilogue: D[co][j]: lane -> j (contiguous in dW rows), registers -> co
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
            const int j = j0 + (wj * TJ + tj) * 32 + l31;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = i0 + (wi * TI + ti) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                if (co < prm.K && j < prm.Jtot) atomicAdd(prm.dw + (long)co * prm.Jtot + j, acc[ti][tj][reg]);
            }
        }
}

template <typename T, int BI, int BJ>
int launch_wgrad(WgradParams& prm, int target_blocks, hipStream_t stream) {
    typedef WgradCfg<T, BI, BJ> Cfg;
    prm.ntile_i = (prm.K + BI - 1) / BI;
    const int ntile_j = (prm.Jtot + BJ - 1) / BJ;
    const long tiles = (long)prm.ntile_i * ntile_j;
    long splits = (target_blocks + tiles - 1) / tiles;
    const long max_splits = (prm.M + Cfg::BKM - 1) / Cfg::BKM;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    long rows = (prm.M + splits - 1) / splits;
    rows = (rows + Cfg::BKM - 1) / Cfg::BKM * Cfg::BKM;
    splits = (prm.M + rows - 1) / rows;
    prm.rows_per_split = (int)rows;
    if (tiles > 0x7fffffffL) return MSFWSI_EINVAL;
    hipLaunchKernelGGL((wgrad_kernel<T, BI, BJ>), dim3((unsigned)tiles, (unsigned)splits), dim3(256),
                       Cfg::LDS_BYTES, stream, prm);
    return msfwsi_launch_status();
}

}  // namespace

extern "C" int msfwsi_conv_wgrad(const msfwsi_conv_desc* d, const void* x, const void* dy, float* dw,
                                 const float* pro_scale, const float* pro_shift, int target_blocks,
                                 void* stream) {
    if (d == nullptr || x == nullptr || dy == nullptr || dw == nullptr) return MSFWSI_EINVAL;
    if (d->dtype != MSFWSI_DT_F32 && d->dtype != MSFWSI_DT_BF16) return MSFWSI_EUNSUPPORTED;
    const int vec = d->dtype == MSFWSI_DT_BF16 ? 8 : 4;
    if (d->C % vec != 0 || d->K % vec != 0) return MSFWSI_EUNSUPPORTED;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->P <= 0 || d->Q <= 0 || d->R <= 0 || d->S <= 0) return MSFWSI_EINVAL;
    if ((pro_scale == nullptr) != (pro_shift == nullptr)) return MSFWSI_EINVAL;
    if ((long)d->N * d->P * d->Q > 0x7fffffffL || (long)d->N * d->H * d->W > 0x7fffffffL) return MSFWSI_EINVAL;
    WgradParams prm{};
    prm.x = x; prm.dy = dy; prm.dw = dw;
    prm.pro_scale = pro_scale; prm.pro_shift = pro_shift;
    prm.N = d->N; prm.H = d->H; prm.W = d->W; prm.C = d->C;
    prm.P = d->P; prm.Q = d->Q; prm.K = d->K;
    prm.R = d->R; prm.S = d->S; prm.stride = d->stride; prm.pad = d->pad;
    prm.M = d->N * d->P * d->Q;
    prm.Jtot = d->R * d->S * d->C;
    if (target_blocks <= 0) target_blocks = 1024;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool small_i = d->K <= 64;
    const bool small_j = prm.Jtot <= 64;
    if (d->dtype == MSFWSI_DT_BF16) {
        if (small_i && small_j) return launch_wgrad<__bf16, 64, 64>(prm, target_blocks, st);
        if (small_i) return launch_wgrad<__bf16, 64, 128>(prm, target_blocks, st);
        if (small_j) return launch_wgrad<__bf16, 128, 64>(prm, target_blocks, st);
        return launch_wgrad<__bf16, 128, 128>(prm, target_blocks, st);
    }
    if (small_i && small_j) return launch_wgrad<float, 64, 64>(prm, target_blocks, st);
    if (small_i) return launch_wgrad<float, 64, 128>(prm, target_blocks, st);
    if (small_j) return launch_wgrad<float, 128, 64>(prm, target_blocks, st);
    return launch_wgrad<float, 128, 128>(prm, target_blocks, st);
}
