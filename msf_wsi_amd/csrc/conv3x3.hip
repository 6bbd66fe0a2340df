// 3x3 / stride 1 / pad 1 convolution (forward and input-gradient) with the input patch staged ONCE per channel
// slab in LDS and reused by all nine filter taps -- the gfx950 kernel for the convs that carry 45 % (ResNet-50)
// to 92 % (ResNet-18) of the encoder FLOPs (reference: conv3x3, src/models/resnet.py:25-28, and its
// convolution_backward(input) reached through tools/ssl_train.py:472).
//
// Why: the generic gather-GEMM (igemm.hip) fetches every input element once per tap, i.e. nine times through
// L2 -> LDS, and waits on that traffic (measured: LDS-DMA latency-bound, ~30 % MFMA utilisation).  Here a
// workgroup owns BM = 256 consecutive raster pixels of ONE image and BN output channels.  Per 64-byte channel
// slab it DMAs the BM + 2W + 2 input pixels those outputs touch ("halo rows": raster order makes tap (r,s) a
// CONSTANT row offset (r-1)*W + (s-1)), a whole slab (= nine tap steps) ahead of use, while the nine per-tap
// weight tiles stream through a 4-slot ring three steps ahead.  Vertical padding = halo rows outside the
// image read a zero page; horizontal padding = lanes whose pixel sits in column 0 / W-1 zero their fragment for
// the s=0 / s=2 taps.  Every byte arrives by global_load_lds_dwordx4; each wave counts its own DMA
// instructions so the per-step wait is an exact `s_waitcnt vmcnt(n)`.
//
// Layout, swizzle, MFMA operand roles and the epilogue (LDS transpose -> 16-byte row stores, BatchNorm sums,
// fused ReLU-gate + BatchNorm-backward sums for DGRAD) are those of igemm.hip.
#include "common.h"
#include "../../include/msfwsi_hip.h"

namespace {

__device__ __attribute__((aligned(256))) unsigned int g_zero_page3[64];

struct C3Params {
    const void* src;   // [N][H][W][C]   (forward: x; dgrad: dY with C = forward Cout)
    const void* wgt;   // forward weight [K][3][3][Cin]
    void* out;         // [N][H][W][Nout]
    double* stats;     // [nshard][2][Nout], nullable
    const void* resid; // [M][Nout], nullable
    const void* mask_c;
    const float* mask_scale;
    const float* mask_shift;
    int N, H, W, C, Nout;
    int nshard, ntile_n, tiles_per_img;
};

template <typename T>
using Frag3 = MmaFrag<T>;

__device__ __forceinline__ void dma16c(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int swz3(int row, int c) { return c ^ ((row >> 2) & 3); }
template <int ROWB>
__device__ __forceinline__ int nat_off3(int k, int cb) {
    const int g = (ROWB >= 256) ? (k & 3) : ((k >> 1) & 1);
    return k * ROWB + ((((cb >> 6) ^ g) << 6) | (cb & 63));
}

__device__ __forceinline__ void wait_vmcnt(int allowed) {
    // exact counted wait; `allowed` is wave-uniform
    if (allowed <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (allowed == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (allowed == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (allowed == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (allowed == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (allowed == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (allowed == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (allowed == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

template <typename T, int BN, bool DGRAD>
struct C3Cfg {
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int BK = ElemTraits<T>::BK;
    static constexpr int BM = 256, NW = 8, WM = 4, WN = 2;
    static constexpr int TM = BM / WM / 32;  // 2
    static constexpr int TN = BN / WN / 32;  // 2 (BN=128) or 1 (BN=64)
    static constexpr int MAXW = 56;
    static constexpr int HALO_ROWS = ((BM + 2 * MAXW + 2 + 15) / 16) * 16;  // 384
    static constexpr int A_BYTES = HALO_ROWS * 64;                          // 24 KiB per slab
    static constexpr int ROWB = BN * (int)sizeof(T);                        // natural weight row (DGRAD)
    static constexpr int B_BYTES = DGRAD ? BK * ROWB : BN * 64;             // one tap: 8 KiB (BN=128)
    static constexpr int NRING = 4;
    static constexpr int LDC = BN + VEC;
    static constexpr int AB_BYTES = 2 * A_BYTES + NRING * B_BYTES;
    static constexpr int C_BYTES = BM * LDC * (int)sizeof(T) + NW * BN * 2 * (int)sizeof(float);
    static constexpr int LDS_BYTES = AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES;
};

template <typename T, int BN, bool DGRAD>
__global__ __launch_bounds__(512) void conv3x3_kernel(const C3Params prm) {
    typedef C3Cfg<T, BN, DGRAD> Cfg;
    constexpr int VEC = Cfg::VEC, BK = Cfg::BK, BM = Cfg::BM, NW = Cfg::NW, WM = Cfg::WM;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, LDC = Cfg::LDC, ROWB = Cfg::ROWB;
    constexpr int NT = 64 * NW;
    typedef typename Frag3<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                        // [2][HALO_ROWS][64 B]
    char* Bs = smem + 2 * Cfg::A_BYTES;     // [NRING][B_BYTES]
    T* Cs = reinterpret_cast<T*>(smem);     // epilogue: [BM][LDC] then red[NW][BN][2]
    float* red = reinterpret_cast<float*>(smem + BM * LDC * (int)sizeof(T));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, lh = lane >> 5;

    const unsigned wgid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wgid % prm.ntile_n;
    const int tile_mi = wgid / prm.ntile_n;
    const int img = tile_mi / prm.tiles_per_img;
    const int p0 = (tile_mi - img * prm.tiles_per_img) * BM;  // first output pixel (raster index in the image)
    const int n0 = tile_n * BN;
    const int HW = prm.H * prm.W, W = prm.W;
    const int halo_rows = BM + 2 * W + 2;

    const T* __restrict__ src = reinterpret_cast<const T*>(prm.src) + (long)img * HW * prm.C;
    const T* __restrict__ wgt = reinterpret_cast<const T*>(prm.wgt);
    const char* zero = reinterpret_cast<const char*>(g_zero_page3);

    const int nslab = prm.C / BK;
    const int nstep = nslab * 9;
    int issued = 0;          // DMA instructions this wave has issued so far
    int b_mark[Cfg::NRING];  // value of `issued` right after the weight tile of ring slot i was issued
#pragma unroll
    for (int i = 0; i < Cfg::NRING; ++i) b_mark[i] = 0;

    // halo rows of slab s: row hr <-> pixel p0 - W - 1 + hr of this image (zero outside)
    auto fetch_A = [&](int s, int buf) {
        char* Ab = As + buf * Cfg::A_BYTES;
        for (int g = wave; g * 16 < halo_rows; g += NW) {
            const int hr = g * 16 + (lane >> 2);
            const int p = p0 - W - 1 + hr;
            const int kc = swz3(hr, lane & 3);
            const bool ok = hr < halo_rows && p >= 0 && p < HW;
            const void* gp = ok ? reinterpret_cast<const void*>(src + (long)p * prm.C + s * BK + kc * VEC)
                                : reinterpret_cast<const void*>(zero);
            dma16c(gp, Ab + g * 1024);
            ++issued;
        }
    };
    // weight tile of step j = (slab, tap): forward [n][k] rows of 64 B; dgrad natural [k][n], flipped tap
    auto fetch_B = [&](int j, int slot) {
        char* Bb = Bs + slot * Cfg::B_BYTES;
        const int s = j / 9, t = j - s * 9;
        if (!DGRAD) {
            for (int g = wave; g * 16 < BN; g += NW) {
                const int row = g * 16 + (lane >> 2);
                const int kc = swz3(row, lane & 3);
                const int n = n0 + row;
                const void* gp = n < prm.Nout
                                     ? reinterpret_cast<const void*>(wgt + ((long)n * 9 + t) * prm.C + s * BK + kc * VEC)
                                     : reinterpret_cast<const void*>(zero);
                dma16c(gp, Bb + g * 1024);
                ++issued;
            }
        } else {
            constexpr int CPRW = ROWB / 16, RPI = 1024 / ROWB;
            for (int g = wave; g * RPI < BK; g += NW) {
                const int krow = g * RPI + lane / CPRW;
                const int cp = lane % CPRW;
                const int gsw = (ROWB >= 256) ? (krow & 3) : ((krow >> 1) & 1);
                const int n = n0 + ((((cp >> 2) ^ gsw) << 2) | (cp & 3)) * VEC;
                const int co = s * BK + krow;  // source channel = forward output channel
                // dX[p] += dY[p + (1-r)W + (1-s)] * W[co][r][s][ci]: halo offset index t' = 8 - t  <->  tap t
                const void* gp = n < prm.Nout ? reinterpret_cast<const void*>(wgt + ((long)co * 9 + (8 - t)) * prm.Nout + n)
                                              : reinterpret_cast<const void*>(zero);
                dma16c(gp, Bb + g * 1024);
                ++issued;
            }
        }
        b_mark[slot] = issued;
    };

    // per-lane horizontal-padding flags of the TM pixels this lane owns as MFMA columns
    bool edge_l[TM], edge_r[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int p = p0 + (wm * TM + tm) * 32 + l31;
        const int col = p % W;
        edge_l[tm] = col == 0;
        edge_r[tm] = col == W - 1;
    }

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

    auto compute = [&](int abuf, int slot, int t) {
        const char* Ab = As + abuf * Cfg::A_BYTES;
        const char* Bb = Bs + slot * Cfg::B_BYTES;
        const int r = t / 3, s = t - r * 3;
        const int roff = r * W + s;  // halo row of output row 0 for this tap
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag_t xf[TM], wf[TN];
            const int cidx = ks * 2 + lh;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int row = (wm * TM + tm) * 32 + l31 + roff;
                xf[tm] = *reinterpret_cast<const frag_t*>(Ab + row * 64 + swz3(row, cidx) * 16);
                const bool kill = (s == 0 && edge_l[tm]) || (s == 2 && edge_r[tm]);
                if (kill) xf[tm] = frag_t{};
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int ncol = (wn * TN + tn) * 32;
                if (!DGRAD) {
                    const int row = ncol + l31;
                    wf[tn] = *reinterpret_cast<const frag_t*>(Bb + row * 64 + swz3(row, cidx) * 16);
                } else if constexpr (sizeof(T) == 2) {
                    const int li = lane & 15, G = lane >> 4;
                    const int q = li >> 2, p = li & 3;
                    const int kbase = ks * 16 + (G >> 1) * 8 + q;
                    const int cb = (ncol + (G & 1) * 16 + p * 4) * 2;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(Bb + nat_off3<ROWB>(kbase, cb)));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(Bb + nat_off3<ROWB>(kbase + 4, cb)));
                    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    wf[tn] = __builtin_bit_cast(frag_t, both);
                } else {
                    frag_t tt;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        tt[e] = *reinterpret_cast<const float*>(Bb + nat_off3<ROWB>(ks * 8 + lh * 4 + e, (ncol + l31) * 4));
                    wf[tn] = tt;
                }
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    mma32<T>(acc[tn][tm], wf[tn], xf[tm]);
                }
        }
    };

    // ---------------- pipeline: A one slab (nine steps) ahead, weights three steps ahead ----------------
    fetch_A(0, 0);
    fetch_B(0, 0);
    if (nstep > 1) fetch_B(1, 1);
    if (nstep > 2) fetch_B(2, 2);
    int slot = 0;
    for (int j = 0; j < nstep; ++j) {
        const int s = j / 9, t = j - s * 9;
        // everything issued up to and including weight tile j must have landed (A(s) was issued before it,
        // except for slab 0 whose A precedes B(0) as well): allow only the instructions issued after it
        wait_vmcnt(issued - b_mark[slot]);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (j + 3 < nstep) fetch_B(j + 3, (slot + 3) & 3);
        if (t == 0 && s + 1 < nslab) fetch_A(s + 1, (s + 1) & 1);
        compute(s & 1, slot, t);
        slot = (slot + 1) & 3;
    }
    __syncthreads();

    // ---------------- epilogue (as igemm.hip) ----------------
    const int m_img0 = img * HW;  // global row of this image's pixel 0
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ncol = (wn * TN + tn) * 32 + 8 * g + 4 * lh;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int row = (wm * TM + tm) * 32 + l31;
                T* dst = Cs + row * LDC + ncol;
#pragma unroll
                for (int e = 0; e < 4; ++e) store_elem<T>(dst, e, acc[tn][tm][g * 4 + e]);
            }
        }
    }
    __syncthreads();
    constexpr int CPR = BN / VEC;
    constexpr int RPP = NT / CPR;
    const int cc = tid % CPR;
    const int rr = tid / CPR;
    const int ncol = n0 + cc * VEC;
    const bool col_ok = ncol < prm.Nout;
    T* __restrict__ out = reinterpret_cast<T*>(prm.out);
    const T* __restrict__ resid = reinterpret_cast<const T*>(prm.resid);
    const T* __restrict__ mask_c = reinterpret_cast<const T*>(prm.mask_c);
    float ssum[VEC], ssq[VEC], msc[VEC], msh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) ssum[e] = ssq[e] = 0.f;
    if (mask_c != nullptr && col_ok) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            msc[e] = prm.mask_scale[ncol + e];
            msh[e] = prm.mask_shift[ncol + e];
        }
    }
#pragma unroll 2
    for (int pass = 0; pass < BM / RPP; ++pass) {
        const int row = rr + pass * RPP;
        const int p = p0 + row;
        if (p < HW && col_ok) {
            uint4 v = *reinterpret_cast<const uint4*>(Cs + row * LDC + cc * VEC);
            const long off = (long)(m_img0 + p) * prm.Nout + ncol;
            if (resid != nullptr) {
                float f[VEC], g[VEC];
                unpack16<T>(v, f);
                unpack16<T>(*reinterpret_cast<const uint4*>(resid + off), g);
#pragma unroll
                for (int e = 0; e < VEC; ++e) f[e] += g[e];
                v = pack16<T>(f);
            }
            if (mask_c != nullptr) {
                float f[VEC], cv[VEC];
                unpack16<T>(v, f);
                unpack16<T>(*reinterpret_cast<const uint4*>(mask_c + off), cv);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    if (!(fmaf(cv[e], msc[e], msh[e]) > 0.f)) f[e] = 0.f;
                    ssum[e] += f[e];
                    ssq[e] = fmaf(f[e], cv[e], ssq[e]);
                }
                v = pack16<T>(f);
                *reinterpret_cast<uint4*>(out + off) = v;
            } else {
                *reinterpret_cast<uint4*>(out + off) = v;
                if (prm.stats != nullptr) {
                    float f[VEC];
                    unpack16<T>(v, f);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        ssum[e] += f[e];
                        ssq[e] = fmaf(f[e], f[e], ssq[e]);
                    }
                }
            }
        }
    }
    if (prm.stats != nullptr) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
#pragma unroll
            for (int off = CPR; off < 64; off <<= 1) {
                ssum[e] += __shfl_xor(ssum[e], off, 64);
                ssq[e] += __shfl_xor(ssq[e], off, 64);
            }
        }
        if (lane < CPR) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                red[(wave * BN + lane * VEC + e) * 2 + 0] = ssum[e];
                red[(wave * BN + lane * VEC + e) * 2 + 1] = ssq[e];
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * BN; i += NT) {
            const int col = i % BN, which = i / BN;
            if (n0 + col < prm.Nout) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) t += red[(w * BN + col) * 2 + which];
                double* dst = prm.stats + ((long)(tile_mi % prm.nshard) * 2 + which) * prm.Nout + n0 + col;
                atomicAdd(dst, (double)t);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 64 -> 64 channels (layer1 of every ResNet here): WEIGHTS-STATIONARY, persistent workgroups.
//
// All nine taps of the 64 x 64 filter (72 KiB in a 2-byte type) stay in LDS for the life of the workgroup, which walks
// a contiguous range of 256-pixel tiles of the batch's global raster (images back to back: a tap that would cross an
// image's top / bottom / left / right edge zeroes the fragment, as the horizontal edges always did).  Per tile the
// 384 halo pixels (256 + 2W + 2, both channel slabs, 48 KiB) are loaded ONCE -- by ordinary global loads into
// registers while the previous tile computes, written to LDS between two barriers -- and the 72 MFMAs of a wave run
// back to back from LDS: four barriers per tile instead of one per (slab, tap) step, no LDS-DMA request in the MFMA
// stream (a wave stalls in the issue of one), 9x less L2 -> LDS traffic than the gather kernel, and the halo rows a
// tile shares with its predecessor come from the same CU's L2 slice.  The BatchNorm sums accumulate in registers
// over all tiles of the workgroup and reach memory with one set of atomics.
// ---------------------------------------------------------------------------------------------------------------
struct C3WParams {
    const void* src;   // [Mtot][64]
    const void* wgt;   // [64][3][3][64]
    void* out;         // [Mtot][64]
    double* stats;
    const void* resid;
    const void* mask_c;
    const float* mask_scale;
    const float* mask_shift;
    const float* pro_scale;  // nullable: the input is the producer's RAW conv output, relu(scale*c + shift) is applied
    const float* pro_shift;  //   on the way from the staging registers to LDS (the activation is never materialised)
    int H, W;
    long Mtot;
    int ntiles, tiles_per_wg, nshard;
    FastDiv div_w, div_h;
};

#ifndef MSFWSI_C3W_BM
#define MSFWSI_C3W_BM 256  // 512 (halo 640 rows, 4 accumulator tiles per wave) spills: measured 430 vs 764 TFLOP/s
#endif
template <typename T>
struct C3WCfg {
    static constexpr int VEC = ElemTraits<T>::VEC, BK = ElemTraits<T>::BK;  // 8, 32
    static constexpr int BM = MSFWSI_C3W_BM, BN = 64, CH = 64, NW = 8, WM = 4, WN = 2, TM = BM / WM / 32;
    static constexpr int MAXW = 64;  // 56: ResNet layer1 at 224^2; 64: at the 256^2 tiles of the fine-tune model
    static constexpr int HALO_ROWS = ((BM + 2 * MAXW + 2 + 63) / 64) * 64;  // 448
    static constexpr int PLANE = HALO_ROWS * 64 + 128;  // one channel slab of the halo; +128 B: the two slabs of a pixel
                                                        // (written by neighbouring lanes) start 32 banks apart
    static constexpr int W_TILE = 4096;                 // one (slab, tap) weight tile: [64 n][64 B] or [32 k][128 B]
    static constexpr int W_BYTES = 18 * W_TILE;
    static constexpr int A_BYTES = 2 * PLANE;
    static constexpr int LDC = BN + VEC;
    static constexpr int C_BYTES = BM * LDC * (int)sizeof(T) + NW * BN * 2 * (int)sizeof(float);
    static constexpr int LDS_BYTES = W_BYTES + (A_BYTES > C_BYTES ? A_BYTES : C_BYTES);
    static constexpr int A_LOADS = HALO_ROWS * 8 / (64 * NW);  // 16-byte chunks per thread and tile: 10 (6)
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
    static_assert(sizeof(T) == 2 && A_LOADS * 64 * NW == HALO_ROWS * 8, "2-byte types, whole chunks per thread");
};

template <typename T, bool DGRAD>
__global__ __launch_bounds__(512) void conv3x3_ws_kernel(const C3WParams prm) {
    typedef C3WCfg<T> Cfg;
    constexpr int VEC = Cfg::VEC, BM = Cfg::BM, BN = Cfg::BN, NW = Cfg::NW, WM = Cfg::WM, TM = Cfg::TM;
    constexpr int LDC = Cfg::LDC, NT = 64 * NW, PLANE = Cfg::PLANE, AL = Cfg::A_LOADS;
    constexpr int ROWB = 128;  // natural weight row (DGRAD): 64 input channels
    typedef typename Frag3<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ws = smem;                                     // 18 weight tiles
    char* As = smem + Cfg::W_BYTES;                      // 2 halo planes
    T* Cs = reinterpret_cast<T*>(smem + Cfg::W_BYTES);   // epilogue tile over the halo planes
    float* red = reinterpret_cast<float*>(smem + Cfg::W_BYTES + BM * LDC * (int)sizeof(T));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, lh = lane >> 5;
    const int W = prm.W, H = prm.H;
    const T* __restrict__ src = reinterpret_cast<const T*>(prm.src);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(prm.wgt);
    T* __restrict__ out = reinterpret_cast<T*>(prm.out);
    const T* __restrict__ resid = reinterpret_cast<const T*>(prm.resid);
    const T* __restrict__ mask_c = reinterpret_cast<const T*>(prm.mask_c);

    const int t_beg = blockIdx.x * prm.tiles_per_wg;
    const int t_end = min(prm.ntiles, t_beg + prm.tiles_per_wg);
    if (t_beg >= t_end) return;

    // ---- weights -> LDS, once (18 tiles x 4 KiB = 72 pieces of 1 KiB: nine per wave) ----
    for (int g = wave; g < 72; g += NW) {
        const int wt = g >> 2, q = g & 3;  // tile (slab, tap), quarter
        const int sl = wt / 9, t = wt - sl * 9;
        const void* gp;
        if (!DGRAD) {
            const int row = q * 16 + (lane >> 2);  // output channel n
            const int kc = swz3(row, lane & 3);
            gp = wgt + ((long)row * 9 + t) * 64 + sl * 32 + kc * VEC;
        } else {
            const int krow = q * 8 + (lane >> 3);  // forward output channel within the slab
            const int cp = lane & 7;
            const int gsw = (krow >> 1) & 1;
            const int n = ((((cp >> 2) ^ gsw) << 2) | (cp & 3)) * VEC;
            gp = wgt + ((long)(sl * 32 + krow) * 9 + (8 - t)) * 64 + n;  // flipped tap, see conv3x3_kernel
        }
        dma16c(gp, Ws + g * 1024);
    }

    // ---- per-thread staging map of the halo: chunk i*512 + tid -> (halo row, slab, 16-byte chunk) ----
    int a_lds[AL];
    int a_hr[AL], a_col[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int idx = i * NT + tid;
        const int hr = idx >> 3, c8 = idx & 7;
        a_hr[i] = hr;
        a_col[i] = c8 * VEC;
        a_lds[i] = (c8 >> 2) * PLANE + hr * 64 + swz3(hr, c8 & 3) * 16;
    }
    uint4 a_reg[AL];
    unsigned a_valid = 0;  // bit i: chunk i of the staged tile lies inside the tensor (padding stays zero)
    auto load_A = [&](int tile) {
        const long g0 = (long)tile * BM - W - 1;
        a_valid = 0;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const long g = g0 + a_hr[i];
            a_reg[i] = make_uint4(0, 0, 0, 0);
            if (g >= 0 && g < prm.Mtot && a_hr[i] < BM + 2 * W + 2) {
                a_reg[i] = *reinterpret_cast<const uint4*>(src + g * 64 + a_col[i]);
                a_valid |= 1u << i;
            }
        }
    };
    // producer BatchNorm + ReLU of this thread's channels (every chunk it stages has the same 8: 512 threads % 8 == 0)
    const bool has_pro = prm.pro_scale != nullptr;
    float qsc[VEC], qsh[VEC];
    if (has_pro) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            qsc[e] = prm.pro_scale[a_col[0] + e];
            qsh[e] = prm.pro_shift[a_col[0] + e];
        }
    }

    // epilogue row-chunk map
    constexpr int CPR = BN / VEC, RPP = NT / CPR, NP = BM / RPP;  // 8 chunks per row, 64 rows per pass, 4 passes
    const int cc = tid % CPR, rr = tid / CPR;
    const int ncol = cc * VEC;
    float ssum[VEC], ssq[VEC], msc[VEC], msh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) ssum[e] = ssq[e] = 0.f;
    if (mask_c != nullptr) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            msc[e] = prm.mask_scale[ncol + e];
            msh[e] = prm.mask_shift[ncol + e];
        }
    }

    load_A(t_beg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // weights (and the first halo) have landed
    for (int tile = t_beg; tile < t_end; ++tile) {
        __syncthreads();  // the previous tile's epilogue is done with the region the halo planes share
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            uint4 v = a_reg[i];
            if (has_pro && ((a_valid >> i) & 1u)) {
                float f[VEC];
                unpack16<T>(v, f);
#pragma unroll
                for (int e = 0; e < VEC; ++e) f[e] = fmaxf(fmaf(f[e], qsc[e], qsh[e]), 0.f);
                v = pack16<T>(f);
            }
            *reinterpret_cast<uint4*>(As + a_lds[i]) = v;
        }
        __syncthreads();
        // next tile's halo and this tile's epilogue operands: in flight while the MFMAs run
        if (tile + 1 < t_end) load_A(tile + 1);
        uint4 r_res[NP], r_msk[NP];
        const long m0 = (long)tile * BM;
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const long m = m0 + rr + ps * RPP;
            if (m < prm.Mtot) {
                if (resid != nullptr) r_res[ps] = *reinterpret_cast<const uint4*>(resid + m * 64 + ncol);
                if (mask_c != nullptr) r_msk[ps] = *reinterpret_cast<const uint4*>(mask_c + m * 64 + ncol);
            }
        }

        // image-edge flags of the TM pixels this lane owns as MFMA columns
        bool e_top[TM], e_bot[TM], e_l[TM], e_r[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const unsigned g = (unsigned)(m0 + (wm * TM + tm) * 32 + l31);
            const unsigned yy = fast_div(g, prm.div_w);
            const unsigned x = g - yy * (unsigned)W;
            const unsigned y = yy - fast_div(yy, prm.div_h) * (unsigned)H;
            e_top[tm] = y == 0;
            e_bot[tm] = y == (unsigned)(H - 1);
            e_l[tm] = x == 0;
            e_r[tm] = x == (unsigned)(W - 1);
        }

        f32x16 acc[TM];
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[b][j] = 0.f;

#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int r = t / 3, s2 = t - r * 3;
                const int roff = r * W + s2;
                const char* Ab = As + sl * PLANE;
                const char* Bb = Ws + (sl * 9 + t) * Cfg::W_TILE;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    frag_t xf[TM], wf;
                    const int cidx = ks * 2 + lh;
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        const int row = (wm * TM + tm) * 32 + l31 + roff;
                        xf[tm] = *reinterpret_cast<const frag_t*>(Ab + row * 64 + swz3(row, cidx) * 16);
                        const bool kill = (r == 0 && e_top[tm]) || (r == 2 && e_bot[tm]) || (s2 == 0 && e_l[tm]) ||
                                          (s2 == 2 && e_r[tm]);
                        if (kill) xf[tm] = frag_t{};
                    }
                    const int ncl = wn * 32;
                    if (!DGRAD) {
                        const int row = ncl + l31;
                        wf = *reinterpret_cast<const frag_t*>(Bb + row * 64 + swz3(row, cidx) * 16);
                    } else {
                        const int li = lane & 15, G = lane >> 4;
                        const int q = li >> 2, pp = li & 3;
                        const int kbase = ks * 16 + (G >> 1) * 8 + q;
                        const int cb = (ncl + (G & 1) * 16 + pp * 4) * 2;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(Bb + nat_off3<ROWB>(kbase, cb)));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(Bb + nat_off3<ROWB>(kbase + 4, cb)));
                        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        wf = __builtin_bit_cast(frag_t, both);
                    }
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) mma32<T>(acc[tm], wf, xf[tm]);
                }
            }
        }
        __syncthreads();  // every wave has read its last halo fragment: the region becomes the output tile

        // ---- epilogue: accumulators -> LDS tile -> 16-byte row chunks ----
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nc = wn * 32 + 8 * g + 4 * lh;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int row = (wm * TM + tm) * 32 + l31;
                T* dst = Cs + row * LDC + nc;
#pragma unroll
                for (int e = 0; e < 4; ++e) store_elem<T>(dst, e, acc[tm][g * 4 + e]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int row = rr + ps * RPP;
            const long m = m0 + row;
            if (m < prm.Mtot) {
                uint4 v = *reinterpret_cast<const uint4*>(Cs + row * LDC + cc * VEC);
                const long off = m * 64 + ncol;
                if (resid != nullptr) {
                    float f[VEC], g[VEC];
                    unpack16<T>(v, f);
                    unpack16<T>(r_res[ps], g);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) f[e] += g[e];
                    v = pack16<T>(f);
                }
                if (mask_c != nullptr) {
                    float f[VEC], cv[VEC];
                    unpack16<T>(v, f);
                    unpack16<T>(r_msk[ps], cv);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        if (!(fmaf(cv[e], msc[e], msh[e]) > 0.f)) f[e] = 0.f;
                        ssum[e] += f[e];
                        ssq[e] = fmaf(f[e], cv[e], ssq[e]);
                    }
                    v = pack16<T>(f);
                    *reinterpret_cast<uint4*>(out + off) = v;
                } else {
                    *reinterpret_cast<uint4*>(out + off) = v;
                    if (prm.stats != nullptr) {
                        float f[VEC];
                        unpack16<T>(v, f);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            ssum[e] += f[e];
                            ssq[e] = fmaf(f[e], f[e], ssq[e]);
                        }
                    }
                }
            }
        }
    }

    if (prm.stats != nullptr) {
        __syncthreads();  // the last tile's row passes have read the tile the scratch follows
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
#pragma unroll
            for (int off = CPR; off < 64; off <<= 1) {
                ssum[e] += __shfl_xor(ssum[e], off, 64);
                ssq[e] += __shfl_xor(ssq[e], off, 64);
            }
        }
        if (lane < CPR) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                red[(wave * BN + lane * VEC + e) * 2 + 0] = ssum[e];
                red[(wave * BN + lane * VEC + e) * 2 + 1] = ssq[e];
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * BN; i += NT) {
            const int col = i % BN, which = i / BN;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[(w * BN + col) * 2 + which];
            double* dst = prm.stats + ((long)(blockIdx.x % prm.nshard) * 2 + which) * 64 + col;
            atomicAdd(dst, (double)t);
        }
    }
}

msfwsi_tunable g_c3_stationary{1};  // msfwsi_set_tuning(9, .): 0 = the 64 -> 64 layers on the per-tile kernels

template <typename T, bool DGRAD>
int launch_c3w(C3WParams& prm, hipStream_t stream) {
    typedef C3WCfg<T> Cfg;
    prm.ntiles = (int)((prm.Mtot + Cfg::BM - 1) / Cfg::BM);
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
        ncu = 256;
    (void)hipGetLastError();
    // one persistent workgroup per CU (120 KiB of LDS each), contiguous tile ranges; short launches spread thinner
    prm.tiles_per_wg = (prm.ntiles + ncu - 1) / ncu;
    const int nblk = (prm.ntiles + prm.tiles_per_wg - 1) / prm.tiles_per_wg;
    auto kern = conv3x3_ws_kernel<T, DGRAD>;
    static bool attr_done = false;
    if (!attr_done) {
        if (int e = msfwsi_raise_lds(reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES)) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), Cfg::LDS_BYTES, stream, prm);
    return msfwsi_launch_status();
}

// the weights-stationary kernel serves: 2-byte types, 64 -> 64 channels, 3x3 / stride 1 / pad 1, W <= 56
bool c3w_ok(const msfwsi_conv_desc* d) {
    return g_c3_stationary && d->dtype != MSFWSI_DT_F32 && d->C == 64 && d->K == 64 && d->R == 3 && d->S == 3 &&
           d->stride == 1 && d->pad == 1 && d->P == d->H && d->Q == d->W && d->W <= C3WCfg<__bf16>::MAXW && d->W >= 3 &&
           d->H >= 3 && (long)d->H * d->W >= 128 && (long)d->N * d->H * d->W <= 0x7fffffffL;
}

template <typename T, int BN, bool DGRAD>
int launch_c3(C3Params& prm, hipStream_t stream) {
    typedef C3Cfg<T, BN, DGRAD> Cfg;
    const int HW = prm.H * prm.W;
    prm.tiles_per_img = (HW + Cfg::BM - 1) / Cfg::BM;
    prm.ntile_n = (prm.Nout + BN - 1) / BN;
    const long nblk = (long)prm.N * prm.tiles_per_img * prm.ntile_n;
    if (nblk <= 0 || nblk > 0x7fffffffL) return MSFWSI_EINVAL;
    auto kern = conv3x3_kernel<T, BN, DGRAD>;
    static bool attr_done = false;  // one attribute per instantiation
    if (!attr_done && Cfg::LDS_BYTES > 64 * 1024) {
        if (int e = msfwsi_raise_lds(reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES)) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), Cfg::LDS_BYTES, stream, prm);
    return msfwsi_launch_status();
}

}  // namespace

// 1 if the halo kernel handles this geometry (3x3, stride 1, pad 1, 14 <= W <= 56, channel slabs of 64 bytes)
extern "C" int msfwsi_conv3x3_supported(const msfwsi_conv_desc* d) {
    if (d == nullptr) return 0;
    if (!msfwsi_dtype_ok(d->dtype)) return 0;
    const int bk = d->dtype == MSFWSI_DT_F32 ? 16 : 32;
    if (d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1) return 0;
    if (d->W > 56 || d->H * d->W < 128) return 0;
    if (d->C % bk != 0 || d->K % bk != 0) return 0;
    return 1;
}

// 1 if the weights-stationary persistent kernel would serve this geometry (callers prefer it over the gather kernel)
extern "C" int msfwsi_conv3x3_stationary(const msfwsi_conv_desc* d) {
    return d != nullptr && msfwsi_dtype_ok(d->dtype) && c3w_ok(d) ? 1 : 0;
}
extern "C" __attribute__((visibility("hidden"))) long msfwsi_c3_set_stationary(long v, int write) {
    const long old = g_c3_stationary;
    if (write) g_c3_stationary = v;
    return old;
}

extern "C" int msfwsi_conv3x3_fwd(const msfwsi_conv_desc* d, const void* x, const void* w, void* y, double* stats,
                                  int nshard, const float* pro_scale, const float* pro_shift, void* stream) {
    if (d == nullptr || !msfwsi_dtype_ok(d->dtype)) return MSFWSI_EUNSUPPORTED;
    const bool ws = c3w_ok(d);  // (a little wider than the per-tile kernel: rows of up to 64 pixels)
    if (!ws && !msfwsi_conv3x3_supported(d)) return MSFWSI_EUNSUPPORTED;
    MSFWSI_CHECK_ARG((pro_scale == nullptr) == (pro_shift == nullptr));
    if (pro_scale != nullptr && !ws) return MSFWSI_EUNSUPPORTED;  // fused prologue: weights-stationary kernel only
    MSFWSI_CHECK_ARG(x != nullptr && w != nullptr && y != nullptr && (stats == nullptr || nshard >= 1));
    MSFWSI_CHECK_ARG((long)d->N * d->H * d->W <= 0x7fffffffL);
    C3Params prm{};
    prm.src = x; prm.wgt = w; prm.out = y; prm.stats = stats; prm.nshard = nshard > 0 ? nshard : 1;
    prm.N = d->N; prm.H = d->H; prm.W = d->W; prm.C = d->C; prm.Nout = d->K;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (ws) {
        C3WParams wp{};
        wp.src = x; wp.wgt = w; wp.out = y; wp.stats = stats; wp.nshard = prm.nshard;
        wp.pro_scale = pro_scale; wp.pro_shift = pro_shift;
        wp.H = d->H; wp.W = d->W; wp.Mtot = (long)d->N * d->H * d->W;
        wp.div_w = make_fastdiv((unsigned)d->W); wp.div_h = make_fastdiv((unsigned)d->H);
        if (d->dtype == MSFWSI_DT_BF16) return launch_c3w<__bf16, false>(wp, st);
        return launch_c3w<_Float16, false>(wp, st);
    }
    MSFWSI_WITH_T(d->dtype, return d->K <= 64 ? launch_c3<T, 64, false>(prm, st) : launch_c3<T, 128, false>(prm, st));
    return MSFWSI_EINVAL;
}

extern "C" int msfwsi_conv3x3_dgrad(const msfwsi_conv_desc* d, const void* dy, const void* w, void* dx,
                                    const void* resid, const void* mask_c, const float* mask_scale,
                                    const float* mask_shift, double* sums, int nshard, void* stream) {
    if (d == nullptr || !msfwsi_dtype_ok(d->dtype)) return MSFWSI_EUNSUPPORTED;
    if (!c3w_ok(d) && !msfwsi_conv3x3_supported(d)) return MSFWSI_EUNSUPPORTED;
    MSFWSI_CHECK_ARG(dy != nullptr && w != nullptr && dx != nullptr);
    MSFWSI_CHECK_ARG((mask_c == nullptr) == (mask_scale == nullptr) && (mask_c == nullptr) == (mask_shift == nullptr));
    MSFWSI_CHECK_ARG((mask_c == nullptr) == (sums == nullptr) && (sums == nullptr || nshard >= 1));
    MSFWSI_CHECK_ARG((long)d->N * d->H * d->W <= 0x7fffffffL);
    C3Params prm{};
    prm.src = dy; prm.wgt = w; prm.out = dx; prm.resid = resid;
    prm.mask_c = mask_c; prm.mask_scale = mask_scale; prm.mask_shift = mask_shift;
    prm.stats = sums; prm.nshard = nshard > 0 ? nshard : 1;
    prm.N = d->N; prm.H = d->H; prm.W = d->W; prm.C = d->K; prm.Nout = d->C;  // stride 1: same H, W
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (c3w_ok(d)) {
        C3WParams wp{};
        wp.src = dy; wp.wgt = w; wp.out = dx; wp.resid = resid;
        wp.mask_c = mask_c; wp.mask_scale = mask_scale; wp.mask_shift = mask_shift;
        wp.stats = sums; wp.nshard = prm.nshard;
        wp.H = d->H; wp.W = d->W; wp.Mtot = (long)d->N * d->H * d->W;
        wp.div_w = make_fastdiv((unsigned)d->W); wp.div_h = make_fastdiv((unsigned)d->H);
        if (d->dtype == MSFWSI_DT_BF16) return launch_c3w<__bf16, true>(wp, st);
        return launch_c3w<_Float16, true>(wp, st);
    }
    MSFWSI_WITH_T(d->dtype, return d->C <= 64 ? launch_c3<T, 64, true>(prm, st) : launch_c3<T, 128, true>(prm, st));
    return MSFWSI_EINVAL;
}
