// The ResNet stem conv (conv1: 7x7 / stride 2 / pad 3, 3 -> 64 channels; src/models/resnet.py:174) in its
// space-to-depth form -- a 4x4 / stride-1 conv with pad 2 (top/left) over [N][H/2][W/2][16] -- as a WEIGHTS-STATIONARY,
// persistent kernel (the third of the family: conv3x3_ws_kernel, wgrad_os_kernel).
//
// The gather kernel fetched 512 bytes per output pixel through L2 -> LDS for 128 bytes of output on a 128 x 64 tile
// (43 FLOP per ingested byte; measured 356 TFLOP/s, 1.7 TB/s).  Here the 32 KiB filter stays in LDS; a workgroup walks a
// contiguous range of 256-position tiles of a ZERO-PADDED raster of the batch (rows of W+2 positions, images of H+2
// rows: the two pad positions / rows between neighbours are every tap's out-of-image value) and loads the
// 256 + 3(W+2) + 3 halo positions of a tile (32 bytes each) ONCE, by plain global loads into registers while the
// previous tile computes.  Two adjacent taps of a filter row are 64 contiguous bytes of the halo = one 32-deep k slab, so
// the MFMA fragments are plain ds_read_b128 at constant offsets; the 16-byte halves of a position are swapped on every
// other group of 8 positions, which makes those reads conflict-free.  Outputs of pad positions are dropped in the row
// pass; BatchNorm sums accumulate in registers over the workgroup's tiles.  Two workgroups share a CU (69 KiB of LDS
// each), so one's loads / stores overlap the other's MFMAs.
#include "common.h"
#include "../../include/msfwsi_hip.h"

namespace {

struct StemParams {
    const void* x;    // [N][H][W][16]   space-to-depth input
    const void* wgt;  // [64][4][4][16]
    void* out;        // [N][H][W][64]   raw conv output
    double* stats;    // [nshard][2][64], nullable
    int N, H, W;
    long npos;        // N * (H+2) * (W+2)
    int ntiles, tiles_per_wg, nshard;
    FastDiv div_img, div_wp;
};

struct StemCfg {
    static constexpr int BM = 256, BN = 64, NW = 8, WM = 4, WN = 2, TM = 2, VEC = 8;
    static constexpr int MAXW = 128;  // 112: 224^2 tiles; 128: the 256^2 tiles of the fine-tune model
    static constexpr int HALO = ((BM + 3 * (MAXW + 2) + 3 + 63) / 64) * 64;  // 704 positions
    static constexpr int W_BYTES = 8 * 4096;                                 // eight (filter row, tap pair) slabs
    static constexpr int A_BYTES = HALO * 32;
    static constexpr int LDC = BN + VEC;
    static constexpr int C_BYTES = BM * LDC * 2;
    static constexpr int RED_BYTES = NW * BN * 2 * (int)sizeof(float);
    static constexpr int AC_BYTES = A_BYTES > C_BYTES ? A_BYTES : C_BYTES;   // the output tile reuses the halo's bytes
    static constexpr int LDS_BYTES = W_BYTES + AC_BYTES;
    static constexpr int A_LOADS = (HALO * 2 + 64 * NW - 1) / (64 * NW);     // 16-byte chunks per thread and tile: 3
    static_assert(RED_BYTES <= AC_BYTES, "statistics scratch reuses the tile region");
};

__device__ __forceinline__ int swzs(int row, int c) { return c ^ ((row >> 2) & 3); }
__device__ __forceinline__ int halo_off(int pos, int half) { return pos * 32 + ((half ^ ((pos >> 3) & 1)) << 4); }

template <typename T>
__global__ __launch_bounds__(512, 4) void stem_ws_kernel(const StemParams prm) {
    typedef StemCfg Cfg;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NW = Cfg::NW, WM = Cfg::WM, TM = Cfg::TM, VEC = Cfg::VEC;
    constexpr int LDC = Cfg::LDC, NT = 64 * NW, AL = Cfg::A_LOADS;
    static_assert(sizeof(T) == 2, "2-byte storage types");
    typedef typename MmaFrag<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ws = smem;
    char* As = smem + Cfg::W_BYTES;
    T* Cs = reinterpret_cast<T*>(smem + Cfg::W_BYTES);
    float* red = reinterpret_cast<float*>(smem + Cfg::W_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, lh = lane >> 5;
    const int W = prm.W, H = prm.H, Wp = W + 2, Hp = H + 2;
    const T* __restrict__ x = reinterpret_cast<const T*>(prm.x);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(prm.wgt);
    T* __restrict__ out = reinterpret_cast<T*>(prm.out);

    const int t_beg = blockIdx.x * prm.tiles_per_wg;
    const int t_end = min(prm.ntiles, t_beg + prm.tiles_per_wg);
    if (t_beg >= t_end) return;

    // ---- filter -> LDS, once: slab sl = (filter row r, tap pair sp) is [64 n][64 B], 32 pieces of 1 KiB ----
    for (int g = wave; g < 32; g += NW) {
        const int sl = g >> 2, q = g & 3;
        const int row = q * 16 + (lane >> 2);
        const int kc = swzs(row, lane & 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wgt + (long)row * 256 + sl * 32 + kc * VEC),
                                         (__attribute__((address_space(3))) void*)(Ws + g * 1024), 16, 0, 0);
    }

    // padded-raster position -> (image, row, column); advanced by BM positions per tile with carries
    auto locate = [&](long q, int& img, int& y, int& xx) {
        if (q < 0) {  // before the batch (q >= -(2 Wp + 2) > -Hp Wp): the pad rows of "image -1"
            img = -1;
            const long r = q + (long)Hp * Wp;
            y = (int)(r / Wp);
            xx = (int)(r - (long)y * Wp);
            return;
        }
        img = (int)fast_div((unsigned)q, prm.div_img);
        const unsigned rem = (unsigned)q - (unsigned)img * (unsigned)(Hp * Wp);
        y = (int)fast_div(rem, prm.div_wp);
        xx = (int)(rem - (unsigned)y * (unsigned)Wp);
    };
    const int adv_y = BM / Wp, adv_x = BM - adv_y * Wp;
    auto advance = [&](int& img, int& y, int& xx) {
        xx += adv_x;
        y += adv_y;
        if (xx >= Wp) {
            xx -= Wp;
            ++y;
        }
        while (y >= Hp) {
            y -= Hp;
            ++img;
        }
    };

    // halo staging: chunk i*512 + tid -> (halo position, 16-byte half); halo position hr <-> q0 - 2 Wp - 2 + hr
    int a_lds[AL], a_img[AL], a_y[AL], a_x[AL];
    bool a_in[AL];
    const int half = tid & 1;
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int hr = (i * NT + tid) >> 1;
        a_in[i] = hr < BM + 3 * Wp + 3;
        a_lds[i] = halo_off(hr < Cfg::HALO ? hr : 0, half);
        if (hr >= Cfg::HALO) a_lds[i] = -1;
        locate((long)t_beg * BM - 2 * Wp - 2 + hr, a_img[i], a_y[i], a_x[i]);
    }
    uint4 a_reg[AL];
    auto load_A = [&]() {  // loads the tile the slot positions point at, then advances them to the next tile
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            a_reg[i] = make_uint4(0, 0, 0, 0);
            if (a_in[i] && (unsigned)a_img[i] < (unsigned)prm.N && a_y[i] < H && a_x[i] < W)
                a_reg[i] = *reinterpret_cast<const uint4*>(x + (((long)a_img[i] * H + a_y[i]) * W + a_x[i]) * 16 + half * VEC);
            advance(a_img[i], a_y[i], a_x[i]);
        }
    };

    // row pass: chunk cc of rows rr + ps*64; the output position of each of this thread's rows, advanced per tile
    constexpr int CPR = BN / VEC, RPP = NT / CPR, NP = BM / RPP;  // 8 chunks per row, 64 rows per pass, 4 passes
    const int cc = tid % CPR, rr = tid / CPR;
    int o_img[NP], o_y[NP], o_x[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) locate((long)t_beg * BM + rr + ps * RPP, o_img[ps], o_y[ps], o_x[ps]);
    float ssum[VEC], ssq[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) ssum[e] = ssq[e] = 0.f;

    load_A();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the filter (and the first halo) have landed
    for (int tile = t_beg; tile < t_end; ++tile) {
        __syncthreads();  // the previous tile's row pass is done with the bytes the halo shares
#pragma unroll
        for (int i = 0; i < AL; ++i)
            if (a_lds[i] >= 0) *reinterpret_cast<uint4*>(As + a_lds[i]) = a_reg[i];
        __syncthreads();
        if (tile + 1 < t_end) load_A();  // in flight while the MFMAs run

        f32x16 acc[TM];
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[b][j] = 0.f;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            const int roff = (sl >> 1) * Wp + 2 * (sl & 1);  // halo position of an output position's tap (r, 2 sp)
            const char* Bb = Ws + sl * 4096;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int j = ks * 2 + lh;  // 16-byte k chunk of the slab: tap (j >> 1), channel half (j & 1)
                frag_t xf[TM], wf;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const int pos = (wm * TM + tm) * 32 + l31 + roff + (j >> 1);
                    xf[tm] = *reinterpret_cast<const frag_t*>(As + halo_off(pos, j & 1));
                }
                const int nrow = wn * 32 + l31;
                wf = *reinterpret_cast<const frag_t*>(Bb + nrow * 64 + swzs(nrow, j) * 16);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) mma32<T>(acc[tm], wf, xf[tm]);
            }
        }
        __syncthreads();  // every wave has read its last halo fragment: the bytes become the output tile

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
            if ((unsigned)o_img[ps] < (unsigned)prm.N && o_y[ps] < H && o_x[ps] < W) {  // pad positions produce nothing
                const int row = rr + ps * RPP;
                const uint4 v = *reinterpret_cast<const uint4*>(Cs + row * LDC + cc * VEC);
                *reinterpret_cast<uint4*>(out + (((long)o_img[ps] * H + o_y[ps]) * W + o_x[ps]) * 64 + cc * VEC) = v;
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
            advance(o_img[ps], o_y[ps], o_x[ps]);
        }
    }

    if (prm.stats != nullptr) {
        __syncthreads();  // the last row pass has read the tile the scratch overlays
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
#pragma unroll
            for (int off = CPR; off < 64; off <<= 1) {
                ssum[e] += __shfl_xor(ssum[e], off, 64);
                ssq[e] += __shfl_xor(ssq[e], off, 64);
            }
        }
        const int lane_e = fresh_lane(), tid_e = wave * 64 + lane_e;  // re-derived: not carried through the main loop
        if (lane_e < CPR) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                red[(wave * BN + lane_e * VEC + e) * 2 + 0] = ssum[e];
                red[(wave * BN + lane_e * VEC + e) * 2 + 1] = ssq[e];
            }
        }
        __syncthreads();
        for (int i = tid_e; i < 2 * BN; i += NT) {
            const int col = i % BN, which = i / BN;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[(w * BN + col) * 2 + which];
            atomicAdd(prm.stats + ((long)(blockIdx.x % prm.nshard) * 2 + which) * 64 + col, (double)t);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the same conv: dW[co][r][s][c] = sum over positions of dY[pos][co] * x[pos + (r-2)(W+2) + (s-2)][c],
// OUTPUT-STATIONARY like wgrad_os_kernel (wgrad.hip): the 64 x 256 gradient is 16 MFMA tiles; a workgroup has FOUR waves,
// wave w owns the four tiles of k slabs 2w and 2w+1 (a slab = filter row, tap pair: 32 columns = 64 contiguous bytes of
// the halo per position) for both halves of the output channels -- two transposed reads per MFMA (eight waves with two
// tiles each needed three and were bound by them: 3.24 ms); two workgroups share a CU.  dY (256 positions x 128 B) and the activation halo (32 B per position) of a chunk of the
// zero-padded raster are loaded once through registers; the fragments come from the transposed LDS read -- for the
// activation with rows that OVERLAP (32-byte stride, 64 bytes wide), which the per-lane addresses of that read allow.
// ---------------------------------------------------------------------------------------------------------------
struct StemWgParams {
    const void* x;   // [N][H][W][16]
    const void* dy;  // [N][H][W][64]
    float* dw;       // [64][4][4][16]
    // BNBWD: dy is the GATED gradient g of the stem's BatchNorm output; the conv-output gradient the weight gradient
    // needs, dc = k1*g + k2*c + k3 (BatchNorm backward, rounded to the storage type exactly as msfwsi_bn_bwd_apply
    // rounds it), is formed on the way from the registers to LDS: neither written nor re-read
    const void* c0;  // [N][H][W][64] raw conv output
    const float *k1, *k2, *k3;
    int N, H, W;
    long npos;
    int nchunks, chunks_per_wg;
    FastDiv div_img, div_wp;
};

struct StemWgCfg {
    static constexpr int BP = 256, NW = 4;
    static constexpr int HALO = StemCfg::HALO;
    static constexpr int DY_BYTES = BP * 128, A_BYTES = HALO * 32;
    static constexpr int K_BYTES = 3 * 64 * 4;                                 // BNBWD: k1, k2, k3 of the 64 channels
    static constexpr int LDS_BYTES = DY_BYTES + A_BYTES + K_BYTES;             // 55 KiB
    static constexpr int DY_LOADS = BP * 8 / (64 * NW);                        // 8 chunks per thread
    static constexpr int A_LOADS = (HALO * 2 + 64 * NW - 1) / (64 * NW);       // 6
};

// natural [row][128 B] image with the 64-byte block swizzle of wgrad.hip
__device__ __forceinline__ int dy_off(int k, int cb) { return k * 128 + ((((cb >> 6) ^ ((k >> 1) & 1)) << 6) | (cb & 63)); }

template <typename T, bool BNBWD>
__global__ __launch_bounds__(256, 2) void stem_wgrad_os_kernel(const StemWgParams prm) {
    typedef StemWgCfg Cfg;
    constexpr int BP = Cfg::BP, NT = 64 * Cfg::NW, DL = Cfg::DY_LOADS, AL = Cfg::A_LOADS, VEC = 8;
    static_assert(sizeof(T) == 2, "2-byte storage types");
    typedef typename MmaFrag<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ds = smem;
    char* As = smem + Cfg::DY_BYTES;
    float* Ks = reinterpret_cast<float*>(smem + Cfg::DY_BYTES + Cfg::A_BYTES);  // [3][64], BNBWD only

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int W = prm.W, H = prm.H, Wp = W + 2, Hp = H + 2;
    const T* __restrict__ x = reinterpret_cast<const T*>(prm.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(prm.dy);
    const T* __restrict__ c0 = reinterpret_cast<const T*>(prm.c0);
    if constexpr (BNBWD) {
        // the coefficients live in LDS and are re-read per chunk: held in registers for the whole kernel they cost 24
        // VGPRs of a budget that is exhausted (256 at two workgroups per CU) and the kernel spilled
        if (tid < 192) Ks[tid] = (tid < 64 ? prm.k1 : (tid < 128 ? prm.k2 : prm.k3))[tid & 63];
    }

    const int c_beg = blockIdx.x * prm.chunks_per_wg;
    const int c_end = min(prm.nchunks, c_beg + prm.chunks_per_wg);
    if (c_beg >= c_end) return;

    auto locate = [&](long q, int& img, int& y, int& xx) {
        if (q < 0) {
            img = -1;
            const long r = q + (long)Hp * Wp;
            y = (int)(r / Wp);
            xx = (int)(r - (long)y * Wp);
            return;
        }
        img = (int)fast_div((unsigned)q, prm.div_img);
        const unsigned rem = (unsigned)q - (unsigned)img * (unsigned)(Hp * Wp);
        y = (int)fast_div(rem, prm.div_wp);
        xx = (int)(rem - (unsigned)y * (unsigned)Wp);
    };
    const int adv_y = BP / Wp, adv_x = BP - adv_y * Wp;
    auto advance = [&](int& img, int& y, int& xx) {
        xx += adv_x;
        y += adv_y;
        if (xx >= Wp) {
            xx -= Wp;
            ++y;
        }
        while (y >= Hp) {
            y -= Hp;
            ++img;
        }
    };

    const int c8 = tid & 7, half = tid & 1;
    int d_lds[DL], d_img[DL], d_y[DL], d_x[DL];
    int a_lds[AL], a_img[AL], a_y[AL], a_x[AL];
    bool a_in[AL];
#pragma unroll
    for (int i = 0; i < DL; ++i) {
        const int row = (i * NT + tid) >> 3;
        d_lds[i] = dy_off(row, c8 * 16);
        locate((long)c_beg * BP + row, d_img[i], d_y[i], d_x[i]);
    }
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int hr = (i * NT + tid) >> 1;
        a_in[i] = hr < BP + 3 * Wp + 3;
        a_lds[i] = hr < Cfg::HALO ? halo_off(hr, half) : -1;
        locate((long)c_beg * BP - 2 * Wp - 2 + hr, a_img[i], a_y[i], a_x[i]);
    }
    uint4 d_reg[DL], a_reg[AL];
    uint4 c_reg[BNBWD ? DL : 1];
    unsigned d_ok = 0;  // BNBWD: bit i = chunk i lies inside an image (pad positions stay zero: no k3 there)
    auto load_chunk = [&]() {
        d_ok = 0;
        if constexpr (BNBWD) {
            // register diet (this variant carries 8 more 16-byte loads): the raster position of load i is the one of
            // load 0 plus 32 i -- derived here instead of carried as eight (image, row, column) triples
#pragma unroll
            for (int i = 0; i < DL; ++i) {
                int xi = d_x[0] + (NT / 8) * i, yi = d_y[0], im = d_img[0];
                while (xi >= Wp) {
                    xi -= Wp;
                    ++yi;
                }
                while (yi >= Hp) {
                    yi -= Hp;
                    ++im;
                }
                d_reg[i] = make_uint4(0, 0, 0, 0);
                c_reg[i] = make_uint4(0, 0, 0, 0);
                if ((unsigned)im < (unsigned)prm.N && yi < H && xi < W) {
                    const long o = (((long)im * H + yi) * W + xi) * 64 + c8 * VEC;
                    d_reg[i] = *reinterpret_cast<const uint4*>(dy + o);
                    c_reg[i] = *reinterpret_cast<const uint4*>(c0 + o);
                    d_ok |= 1u << i;
                }
            }
            advance(d_img[0], d_y[0], d_x[0]);
        } else
#pragma unroll
        for (int i = 0; i < DL; ++i) {
            d_reg[i] = make_uint4(0, 0, 0, 0);
            if ((unsigned)d_img[i] < (unsigned)prm.N && d_y[i] < H && d_x[i] < W) {
                const long o = (((long)d_img[i] * H + d_y[i]) * W + d_x[i]) * 64 + c8 * VEC;
                d_reg[i] = *reinterpret_cast<const uint4*>(dy + o);
            }
            advance(d_img[i], d_y[i], d_x[i]);
        }
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            a_reg[i] = make_uint4(0, 0, 0, 0);
            if (a_in[i] && (unsigned)a_img[i] < (unsigned)prm.N && a_y[i] < H && a_x[i] < W)
                a_reg[i] = *reinterpret_cast<const uint4*>(x + (((long)a_img[i] * H + a_y[i]) * W + a_x[i]) * 16 + half * VEC);
            advance(a_img[i], a_y[i], a_x[i]);
        }
    };

    // transposed fragments: 16 positions from `row0`, 32 columns
    auto read_dy = [&](int row0, int col0) -> frag_t {
        const int li = lane & 15, G = lane >> 4;
        const int q = li >> 2, p = li & 3;
        const int kb = row0 + (G >> 1) * 8 + q;
        const int cb = (col0 + (G & 1) * 16 + p * 4) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Ds + dy_off(kb, cb)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Ds + dy_off(kb + 4, cb)));
        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(frag_t, both);
    };
    // activation: "row" k = 64 bytes starting at halo position hp0 + k (the two taps of the slab), columns = (tap, channel)
    auto read_x = [&](int hp0) -> frag_t {
        const int li = lane & 15, G = lane >> 4;
        const int q = li >> 2, p = li & 3;
        const int cbyte = ((G & 1) * 16 + p * 4) * 2;  // 0 .. 63: position offset cbyte >> 5, half (cbyte >> 4) & 1
        const int k0 = hp0 + (G >> 1) * 8 + q + (cbyte >> 5);
        const int hf = (cbyte >> 4) & 1, in = cbyte & 15;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(As + halo_off(k0, hf) + in));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(As + halo_off(k0 + 4, hf) + in));
        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(frag_t, both);
    };

    f32x16 acc[2][2];  // [slab 2w + s][output-channel half]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;
    // halo position of an output position's tap (r, 2 sp) of slab sl = r*2 + sp: slabs 2w and 2w+1 share the filter row w
    const int roff0 = wave * Wp, roff1 = wave * Wp + 2;

    load_chunk();
    for (int chunk = c_beg; chunk < c_end; ++chunk) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < DL; ++i) {
            if constexpr (BNBWD) {
                if ((d_ok >> i) & 1u) {  // dc = k1*g + k2*c + k3 for this lane's 8 channels (c8 is fixed per thread)
                    float g[VEC], cv[VEC];
                    unpack16<T>(d_reg[i], g);
                    unpack16<T>(c_reg[i], cv);
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        g[e] = fmaf(Ks[c8 * VEC + e], g[e], fmaf(Ks[64 + c8 * VEC + e], cv[e], Ks[128 + c8 * VEC + e]));
                    d_reg[i] = pack16<T>(g);
                }
            }
            *reinterpret_cast<uint4*>(Ds + d_lds[i]) = d_reg[i];
        }
#pragma unroll
        for (int i = 0; i < AL; ++i)
            if (a_lds[i] >= 0) *reinterpret_cast<uint4*>(As + a_lds[i]) = a_reg[i];
        __syncthreads();
        if (chunk + 1 < c_end) load_chunk();
        frag_t af[2][2], bf[2][2];
        auto fetch_frags = [&](int ks, int s) {
            af[s][0] = read_dy(ks * 16, 0);
            af[s][1] = read_dy(ks * 16, 32);
            bf[s][0] = read_x(ks * 16 + roff0);
            bf[s][1] = read_x(ks * 16 + roff1);
        };
        auto mma_step = [&](int s) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) mma32<T>(acc[sl][cb], af[s][cb], bf[s][sl]);
        };
        fetch_frags(0, 0);
#pragma unroll 1
        for (int kk = 0; kk < BP / 32; ++kk) {
            fetch_frags(2 * kk + 1, 1);
            mma_step(0);
            if (kk + 1 < BP / 32) fetch_frags(2 * kk + 2, 0);
            mma_step(1);
        }
    }

    // dW[co][256]: lane -> column (slab 2w + sl, 32 columns), registers -> co
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int j = (2 * wave + sl) * 32 + l31;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                atomicAdd(prm.dw + (long)co * 256 + j, acc[sl][cb][reg]);
            }
    }
}

template <typename T>
int launch_stem_wgrad(StemWgParams& prm, hipStream_t stream) {
    typedef StemWgCfg Cfg;
    prm.nchunks = (int)((prm.npos + Cfg::BP - 1) / Cfg::BP);
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
        ncu = 256;
    (void)hipGetLastError();
    const int slots = 2 * ncu;  // two four-wave workgroups per CU (registers: ~200 per lane)
    prm.chunks_per_wg = (prm.nchunks + slots - 1) / slots;
    const int nblk = (prm.nchunks + prm.chunks_per_wg - 1) / prm.chunks_per_wg;
    if (prm.c0 != nullptr)
        hipLaunchKernelGGL((stem_wgrad_os_kernel<T, true>), dim3((unsigned)nblk), dim3(64 * Cfg::NW), Cfg::LDS_BYTES,
                           stream, prm);
    else
        hipLaunchKernelGGL((stem_wgrad_os_kernel<T, false>), dim3((unsigned)nblk), dim3(64 * Cfg::NW), Cfg::LDS_BYTES,
                           stream, prm);
    return msfwsi_launch_status();
}

msfwsi_tunable g_stem_ws{1};  // msfwsi_set_tuning(12, .): 0 = the stem on the gather kernel
msfwsi_tunable g_stem_os_min_pos{32L * 512 * 256};  // msfwsi_set_tuning(13, .): smallest padded raster the weight-gradient kernel takes

template <typename T>
int launch_stem_ws(StemParams& prm, hipStream_t stream) {
    typedef StemCfg Cfg;
    prm.ntiles = (int)((prm.npos + Cfg::BM - 1) / Cfg::BM);
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
        ncu = 256;
    (void)hipGetLastError();
    const int slots = 2 * ncu;  // two workgroups per CU
    prm.tiles_per_wg = (prm.ntiles + slots - 1) / slots;
    const int nblk = (prm.ntiles + prm.tiles_per_wg - 1) / prm.tiles_per_wg;
    auto kern = stem_ws_kernel<T>;
    static bool attr_done = false;
    if (!attr_done) {
        if (int e = msfwsi_raise_lds(reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES)) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), Cfg::LDS_BYTES, stream, prm);
    return msfwsi_launch_status();
}

}  // namespace

extern "C" __attribute__((visibility("hidden"))) long msfwsi_stem_set_ws(long v, int write) {
    const long old = g_stem_ws;
    if (write) g_stem_ws = v;
    return old;
}

// weight gradient of the space-to-depth stem on the output-stationary kernel; MSFWSI_EUNSUPPORTED where it does not
// apply (msfwsi_conv_wgrad then takes the gather kernel).  From ~32 chunks per workgroup (16 384 atomics each at the end).
extern "C" __attribute__((visibility("hidden"))) int msfwsi_stem_os_wgrad(const msfwsi_conv_desc* d, const void* x,
                                                                         const void* dy, float* dw, const void* c0,
                                                                         const float* k1, const float* k2,
                                                                         const float* k3, void* stream) {
    if (!g_stem_ws || d->dtype == MSFWSI_DT_F32 || d->C != 16 || d->K != 64 || d->R != 4 || d->S != 4 || d->stride != 1 ||
        d->pad != 2 || d->P != d->H || d->Q != d->W || d->W > StemCfg::MAXW || d->W < 2 || d->H < 2)
        return MSFWSI_EUNSUPPORTED;
    const long npos = (long)d->N * (d->H + 2) * (d->W + 2);
    if (npos > 0x7fffffffL || npos < g_stem_os_min_pos) return MSFWSI_EUNSUPPORTED;
    StemWgParams prm{};
    prm.x = x; prm.dy = dy; prm.dw = dw;
    prm.c0 = c0; prm.k1 = k1; prm.k2 = k2; prm.k3 = k3;
    prm.N = d->N; prm.H = d->H; prm.W = d->W; prm.npos = npos;
    prm.div_img = make_fastdiv((unsigned)((d->H + 2) * (d->W + 2)));
    prm.div_wp = make_fastdiv((unsigned)(d->W + 2));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == MSFWSI_DT_BF16) return launch_stem_wgrad<__bf16>(prm, st);
    return launch_stem_wgrad<_Float16>(prm, st);
}
extern "C" __attribute__((visibility("hidden"))) long msfwsi_stem_set_os_min(long v, int write) {
    const long old = g_stem_os_min_pos;
    if (write) g_stem_os_min_pos = v;
    return old;
}

// the space-to-depth stem on the weights-stationary kernel; MSFWSI_EUNSUPPORTED where it does not apply (the caller,
// msfwsi_stem_conv_fwd, then takes the gather kernel)
extern "C" __attribute__((visibility("hidden"))) int msfwsi_stem_ws_fwd(int dtype, const void* x, const void* w, void* y,
                                                                       double* stats, int nshard, int N, int H, int W,
                                                                       int CP, int K, int R, int S, int stride, int pad,
                                                                       int P, int Q, void* stream) {
    if (!g_stem_ws || dtype == MSFWSI_DT_F32 || CP != 16 || K != 64 || R != 4 || S != 4 || stride != 1 || pad != 2 ||
        P != H || Q != W || W > StemCfg::MAXW || W < 2 || H < 2 || (long)N * (H + 2) * (W + 2) > 0x7fffffffL)
        return MSFWSI_EUNSUPPORTED;
    StemParams prm{};
    prm.x = x; prm.wgt = w; prm.out = y; prm.stats = stats; prm.nshard = nshard > 0 ? nshard : 1;
    prm.N = N; prm.H = H; prm.W = W;
    prm.npos = (long)N * (H + 2) * (W + 2);
    prm.div_img = make_fastdiv((unsigned)((H + 2) * (W + 2)));
    prm.div_wp = make_fastdiv((unsigned)(W + 2));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MSFWSI_DT_BF16) return launch_stem_ws<__bf16>(prm, st);
    return launch_stem_ws<_Float16>(prm, st);
}
