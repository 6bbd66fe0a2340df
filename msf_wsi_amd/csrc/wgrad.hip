// Weight-gradient GEMM for conv / linear layers on gfx950:
//
//   dW[co][(r,s,ci)] += sum_m dY[m][co] * act(X[img(m), p*stride-pad+r, q*stride-pad+s, ci])
//
// (the wgrad half of autograd's conv2d/linear backward that the reference reaches through
// `scaler.scale(loss).backward()`, tools/ssl_train.py:472).  The reduction runs over output pixels m,
// so both operand tiles are staged in their natural [pixel][channel] layout and the MFMA fragments are
// fetched with the gfx950 transposed LDS read (ds_read_b64_tr_b16) for bf16, plain b32 reads for fp32.
// Rows are unpadded; their 64-byte blocks are XOR-swizzled with the pixel index so that the four rows of
// one transposed read hit the four quarters of the 256-byte bank row.  The dY tile always arrives by LDS-DMA
// (global_load_lds_dwordx4, swizzle on the per-lane source address); the activation tile does too unless
// the producer's BatchNorm+ReLU has to be recomputed on the way in (`act`, from the saved raw conv output, so
// the normalised activation is never stored) -- then it is register-staged into the same image.  With both
// tiles on DMA the loop keeps two pixel-slabs in flight behind a counted vmcnt.  Split over m across
// workgroups; partial tiles are added with fp32 global atomics (one accumulator register = two 128-byte row
// segments per wave).
#include "common.h"
#include "../../include/msfwsi_hip.h"

#ifndef MSFWSI_WGRAD_BIG_WAVES
#define MSFWSI_WGRAD_BIG_WAVES 16  // waves of the 256 x 256 tile (8: 128 x 64 per wave, 16: 64 x 64)
#endif
#ifndef MSFWSI_WGRAD_BIG_STAGES
#define MSFWSI_WGRAD_BIG_STAGES 4  // LDS stages of the 256 x 256 tile (32 KiB each); 4: the DMA requests of slab kt+3 are issued BEFORE the barrier of iteration kt (see the pixel loop)
#endif
#ifndef MSFWSI_WGRAD_SMALL_STAGES
#define MSFWSI_WGRAD_SMALL_STAGES 3  // LDS stages of the 4-wave tiles (A/B: make EXTRA=-DMSFWSI_WGRAD_SMALL_STAGES=2)
#endif
#ifndef MSFWSI_WGRAD_PIPE
#define MSFWSI_WGRAD_PIPE 1  // fragment reads software-pipelined ACROSS the slab barrier (0: the round-2..5 loop; A/B: make EXTRA=-DMSFWSI_WGRAD_PIPE=0)
#endif
#ifndef MSFWSI_FETCH_FIRST
#define MSFWSI_FETCH_FIRST 1  // DMA requests of slab kt+2 before the MFMAs of slab kt (0: after them; A/B: make EXTRA=-DMSFWSI_FETCH_FIRST=0)
#endif

namespace {

__device__ __attribute__((aligned(256))) unsigned int g_wzero_page[64];

struct WgradParams {
    const void* x;
    const void* dy;
    float* dw;
    double* dw64;  // if set: the pixel splits accumulate in fp64 instead (msfwsi_gram: the order of the atomic additions
                   // then moves the sum by ~1e-16, invisible after rounding to fp32 -- run-to-run reproducible forward)
    const float* pro_scale;
    const float* pro_shift;
    void* aout;    // XPRO, 1x1 / stride 1 only, nullable: relu(pro_scale * x + pro_shift) [M][C], written by the tiles of row 0
    int N, H, W, C;
    int P, Q, K;
    int R, S, stride, pad;
    int M, Jtot;
    int rows_per_split;
    int ntile_i;
    FastDiv div_pq, div_q;
    int store;     // 1: ONE pixel split, the tile is STORED (dw = ..., no atomics, dw need not be cleared): msfwsi_conv_wgrad_store
};

constexpr int wgrad_waves(int bi, int bj) { return (bi == 256 && bj == 256) ? MSFWSI_WGRAD_BIG_WAVES : 4; }
constexpr int wgrad_threads(int bi, int bj) { return 64 * wgrad_waves(bi, bj); }

template <typename T, int BI, int BJ, bool XPRO = false>
struct WgradCfg {
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int BKM = ElemTraits<T>::BK;  // pixels per stage
    static constexpr int ROWI = BI * (int)sizeof(T);
    static constexpr int ROWJ = BJ * (int)sizeof(T);
    // 256 x 256 for the deep layers: per 32-pixel slab the workgroup takes in 32 KiB for 2 x 256 x 256 x 32 FLOP,
    // 128 FLOP per ingested byte against 64 for the 128 x 128 tile.  Measured with the MFMAs removed, `buffer_load ...
    // lds` delivers 15.5 TB/s chip-wide out of L2: a roof of 0.99 PFLOP/s for the small tile and 1.98 for this one.
    // Sixteen waves of 64 x 64 (2 DMA pieces and 8 MFMAs per wave and slab) rather than eight of 128 x 64: a wave
    // stalls in the ISSUE of a DMA piece while earlier ones land (about one piece per 150 ns and wave), so the
    // requests are spread over as many waves as the register file allows (+5-9 % over eight waves).
    static constexpr int NW = wgrad_waves(BI, BJ);
    static constexpr int WI = NW == 16 ? 4 : ((BI >= 128 || BJ <= 64) ? 2 : 1);  // waves along co
    static constexpr int WJ = NW / WI;
    static constexpr int TI = BI / WI / 32;
    static constexpr int TJ = BJ / WJ / 32;
    static constexpr int A_BYTES = BKM * ROWI;
    static constexpr int B_BYTES = BKM * ROWJ;
    static constexpr int A_IT = A_BYTES / (1024 * NW);  // 1-KiB DMA instructions per wave per slab
    static constexpr int B_IT = B_BYTES / (1024 * NW);
    static constexpr int NST = XPRO ? 2 : (NW >= 8 ? MSFWSI_WGRAD_BIG_STAGES : MSFWSI_WGRAD_SMALL_STAGES);
    static constexpr int LDS_BYTES = NST * (A_BYTES + B_BYTES);
    static_assert(TI >= 1 && TJ >= 1, "tile too small");
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small");
};

template <typename T>
using WFrag = MmaFrag<T>;

__device__ __forceinline__ void wdma16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// same through a buffer descriptor: per-lane byte offset + wave-uniform byte offset, out-of-range -> zeros.
// (kept in a __device__ helper: the builtin does not exist for the host pass of a __global__ template)
__device__ __forceinline__ void wdma16_buf(__amdgpu_buffer_rsrc_t rsrc, void* lds_wave_base, int voff, int soff) {
    lds_dma16_buf(rsrc, lds_wave_base, voff, soff);  // inline asm: see common.h
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N == 0 || N == 2 || N == 3 || N == 4 || N == 6 || N == 8 || N == 16, "vmcnt literal table");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}

template <int ROWB>
__device__ __forceinline__ int wswz(int k) {
    return (ROWB >= 256) ? (k & 3) : ((k >> 1) & 1);
}
// byte offset of byte column `cb` of natural-tile row k
template <int ROWB>
__device__ __forceinline__ int wnat_off(int k, int cb) {
    return k * ROWB + ((((cb >> 6) ^ wswz<ROWB>(k)) << 6) | (cb & 63));
}

// fragment of a natural-layout [k][col] LDS tile: 32 columns starting at col0, one k-group `ks`
template <typename T, int ROWB>
__device__ __forceinline__ typename WFrag<T>::type read_tr_frag(const char* tile, int ks, int col0, int lane) {
    typedef typename WFrag<T>::type frag_t;
    if constexpr (sizeof(T) == 2) {
        const int li = lane & 15, G = lane >> 4;
        const int q = li >> 2, p = li & 3;
        const int kb = ks * 16 + (G >> 1) * 8 + q;
        const int cb = (col0 + (G & 1) * 16 + p * 4) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(tile + wnat_off<ROWB>(kb, cb)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(tile + wnat_off<ROWB>(kb + 4, cb)));
        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(frag_t, both);
    } else {
        frag_t t;
        const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            t[e] = *reinterpret_cast<const float*>(tile + wnat_off<ROWB>(ks * 8 + lh * 4 + e, (col0 + l31) * 4));
        return t;
    }
}

// LIN: stride 1 and P == H, Q == W ("same" padding or 1x1): the source pixel of output pixel m at tap (r,s) is
// m + (r-pad)*W + (s-pad), linear in m.  Both tiles then arrive by `buffer_load_dwordx4 ... lds` whose per-lane
// byte offsets are constant over the whole pixel loop; the slab only moves the scalar offset, rows past the end of
// the split / tensor fall outside the buffer range (zeros), and the k loop carries no address arithmetic for 1x1
// filters and ~13 VALU per piece (image-border test) for 3x3.  (The generic path spends two integer divisions per
// piece and slab: measured 21 VALU instructions per MFMA.)
template <typename T, int BI, int BJ, bool XPRO, bool LIN>
__global__ __launch_bounds__(wgrad_threads(BI, BJ)) void wgrad_kernel(const WgradParams prm) {
    static_assert(!(XPRO && LIN), "the linear fast path is pure DMA");
    typedef WgradCfg<T, BI, BJ, XPRO> Cfg;
    constexpr int VEC = Cfg::VEC, BKM = Cfg::BKM, ROWI = Cfg::ROWI, ROWJ = Cfg::ROWJ;
    constexpr int WI = Cfg::WI, TI = Cfg::TI, TJ = Cfg::TJ, NW = Cfg::NW;
    constexpr int A_IT = Cfg::A_IT, B_IT = Cfg::B_IT;
    constexpr int CPI = ROWI / 16, CPJ = ROWJ / 16;      // 16-byte chunks per row
    constexpr int RPI_A = 1024 / ROWI, RPI_B = 1024 / ROWJ;  // rows per DMA instruction
    typedef typename WFrag<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                            // [NST][BKM][ROWI]  dY tile
    char* Bs = smem + Cfg::NST * Cfg::A_BYTES;  // [NST][BKM][ROWJ]  activation tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave % WI, wj = wave / WI;
    const int l31 = lane & 31, lh = lane >> 5;

    // tiles of one pixel split share their dY / activation rows: keep them on one XCD
    const unsigned wgid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const unsigned tile_lin = wgid % gridDim.x, split = wgid / gridDim.x;
    const int tile_i = tile_lin % prm.ntile_i;
    const int tile_j = tile_lin / prm.ntile_i;
    const int i0 = tile_i * BI, j0 = tile_j * BJ;
    const int mbeg = split * prm.rows_per_split;
    const int mend = min(prm.M, mbeg + prm.rows_per_split);

    const T* __restrict__ x = reinterpret_cast<const T*>(prm.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(prm.dy);
    const char* zero = reinterpret_cast<const char*>(g_wzero_page);
    const int PQ = prm.P * prm.Q;

    // ---- fixed per-lane staging map (the slab advances by BKM rows, a multiple of 4: swizzle unchanged) ----
    int a_row[A_IT], a_col[A_IT];
    bool a_colok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int krow = (i * NW + wave) * RPI_A + lane / CPI;
        const int cp = lane % CPI;
        const int cl = (((cp >> 2) ^ wswz<ROWI>(krow)) << 2) | (cp & 3);
        a_row[i] = krow;
        a_col[i] = i0 + cl * VEC;
        a_colok[i] = a_col[i] < prm.K;
    }
    int b_row[B_IT], b_r[B_IT], b_s[B_IT], b_c[B_IT];
    bool b_colok[B_IT];
    float psc[XPRO ? B_IT : 1][VEC], psh[XPRO ? B_IT : 1][VEC];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int krow = (i * NW + wave) * RPI_B + lane / CPJ;
        const int cp = lane % CPJ;
        const int cl = (((cp >> 2) ^ wswz<ROWJ>(krow)) << 2) | (cp & 3);
        const int jcol = j0 + cl * VEC;
        b_row[i] = krow;
        b_colok[i] = jcol < prm.Jtot;
        const int jj = b_colok[i] ? jcol : 0;
        const int rs = jj / prm.C;
        b_c[i] = jj - rs * prm.C;
        b_r[i] = rs / prm.S;
        b_s[i] = rs - b_r[i] * prm.S;
        if (XPRO) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                psc[i][e] = b_colok[i] ? prm.pro_scale[b_c[i] + e] : 1.f;
                psh[i][e] = b_colok[i] ? prm.pro_shift[b_c[i] + e] : 0.f;
            }
        }
    }

    uint4 b_reg[XPRO ? B_IT : 1];
    bool b_ok[XPRO ? B_IT : 1];

    // ---- linear fast path: buffer descriptors + constant per-lane offsets ----
    constexpr int ES = (int)sizeof(T);
    constexpr int OOB = (int)0x80000000;
    int la_voff[LIN ? A_IT : 1], lb_voff[LIN ? B_IT : 1], lb_eff[LIN ? B_IT : 1];
    int lb_rem[LIN ? B_IT : 1], lb_pbad[LIN ? B_IT : 1], lb_qbad[LIN ? B_IT : 1];
    int l_soff_a = 0, l_soff_b = 0;
    const bool taps = prm.R * prm.S > 1;
    const float inv_q = 1.0f / (float)prm.Q, inv_pq = 1.0f / (float)PQ;
    const long dy_bytes = (long)(mend > mbeg ? mend - mbeg : 0) * prm.K * ES;
    const __amdgpu_buffer_rsrc_t srd_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dy) + (long)mbeg * prm.K, 0, (int)dy_bytes, 0x00020000);
    const long dr_min = -((long)prm.pad * prm.W + prm.pad);
    const long xbase = (long)mbeg + dr_min;  // pixel index of the descriptor base (may precede the tensor)
    const long x_bytes_full = ((long)prm.M - xbase) * prm.C * ES;
    const __amdgpu_buffer_rsrc_t srd_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(x) + xbase * prm.C, 0, (int)(x_bytes_full > 0x7fffffffL ? 0x7fffffffL : x_bytes_full), 0x00020000);
    if constexpr (LIN) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) la_voff[i] = a_colok[i] ? (a_row[i] * prm.K + (a_col[i])) * ES : OOB;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const long dr = (long)(b_r[i] - prm.pad) * prm.W + (b_s[i] - prm.pad);
            lb_voff[i] = b_colok[i] ? (int)(((b_row[i] + dr - dr_min) * prm.C + b_c[i]) * ES) : OOB;
            lb_eff[i] = lb_voff[i];
            // a tap leaves the image only on one row (p_bad) and / or one column (q_bad) of output pixels
            const int hr = b_r[i] - prm.pad, ws = b_s[i] - prm.pad;
            lb_pbad[i] = hr < 0 ? -hr - 1 : (hr > 0 ? prm.P - hr : -1);
            lb_qbad[i] = ws < 0 ? -ws - 1 : (ws > 0 ? prm.Q - ws : -1);
            lb_rem[i] = (mbeg + b_row[i]) % PQ;
        }
    }

    auto fetch = [&](int mb, int st) {
        char* Ab = As + st * Cfg::A_BYTES;
        char* Bb = Bs + st * Cfg::B_BYTES;
        if constexpr (LIN) {  // slabs are fetched in order: the scalar offsets and the pixel position advance by BKM
#pragma unroll
            for (int i = 0; i < A_IT; ++i)
                wdma16_buf(srd_a, Ab + (i * NW + wave) * 1024, la_voff[i], l_soff_a);
            if (taps) {
#pragma unroll
                for (int i = 0; i < B_IT; ++i) {
                    const int rem = lb_rem[i];
                    const int pp = (int)(((float)rem + 0.5f) * inv_q);  // exact: rem < 2^21
                    const int qq = rem - pp * prm.Q;
                    // |hr|,|ws| <= 1 for the 3x3 filters this path serves (wider taps use the generic kernel)
                    lb_eff[i] = (pp != lb_pbad[i] && qq != lb_qbad[i]) ? lb_voff[i] : OOB;
                    int nr = rem + BKM;
                    nr -= (int)(((float)nr + 0.5f) * inv_pq) * PQ;
                    lb_rem[i] = nr;
                }
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
                wdma16_buf(srd_b, Bb + (i * NW + wave) * 1024, lb_eff[i], l_soff_b);
            l_soff_a += BKM * prm.K * ES;
            l_soff_b += BKM * prm.C * ES;
            return;
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int m = mb + a_row[i];
            const bool ok = m < mend && a_colok[i];
            const void* g = ok ? reinterpret_cast<const void*>(dy + (long)m * prm.K + a_col[i])
                               : reinterpret_cast<const void*>(zero);
            wdma16(g, Ab + (i * NW + wave) * 1024);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int m = mb + b_row[i];
            bool ok = (m < mend) && b_colok[i];
            long off = 0;
            if (ok) {
                const int img = (int)fast_div((unsigned)m, prm.div_pq);
                const int rem = m - img * PQ;
                const int p = (int)fast_div((unsigned)rem, prm.div_q);
                const int q = rem - p * prm.Q;
                const int h = p * prm.stride - prm.pad + b_r[i];
                const int w = q * prm.stride - prm.pad + b_s[i];
                ok = (unsigned)h < (unsigned)prm.H && (unsigned)w < (unsigned)prm.W;
                off = (((long)img * prm.H + h) * prm.W + w) * prm.C + b_c[i];
            }
            if (XPRO) {
                b_ok[i] = ok;
                b_reg[i] = make_uint4(0, 0, 0, 0);
                if (ok) b_reg[i] = *reinterpret_cast<const uint4*>(x + off);
            } else {
                const void* g = ok ? reinterpret_cast<const void*>(x + off) : reinterpret_cast<const void*>(zero);
                wdma16(g, Bb + (i * NW + wave) * 1024);
            }
        }
    };
    // (mb: the first pixel row of the slab being committed -- the activation by-product's address, 1x1 convs only)
    T* __restrict__ aout = XPRO && tile_i == 0 ? reinterpret_cast<T*>(prm.aout) : nullptr;
    auto commit = [&](int st, int mb) {
        if (XPRO) {
            char* Bb = Bs + st * Cfg::B_BYTES;
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                uint4 v = b_reg[i];
                if (b_ok[i]) {
                    float f[VEC];
                    unpack16<T>(v, f);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) f[e] = fmaxf(fmaf(f[e], psc[i][e], psh[i][e]), 0.f);
                    v = pack16<T>(f);
                    // the normalised operand is what msfwsi_bn_act would write: the tiles of output row 0 store it, so that the
                    // launches that need it materialised (the two-source input gradient of the folded tail) skip that pass
                    if (aout != nullptr) *reinterpret_cast<uint4*>(aout + (long)(mb + b_row[i]) * prm.C + b_c[i]) = v;
                }
                *reinterpret_cast<uint4*>(Bb + (i * NW + wave) * 1024 + lane * 16) = v;
            }
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

    auto compute = [&](int st) {
        const char* Ab = As + st * Cfg::A_BYTES;
        const char* Bb = Bs + st * Cfg::B_BYTES;
#pragma unroll
        for (int ks = 0; ks < BKM / (2 * VEC); ++ks) {
            frag_t af[TI], bf[TJ];
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) af[ti] = read_tr_frag<T, ROWI>(Ab, ks, (wi * TI + ti) * 32, lane);
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj) bf[tj] = read_tr_frag<T, ROWJ>(Bb, ks, (wj * TJ + tj) * 32, lane);
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                for (int tj = 0; tj < TJ; ++tj) mma32<T>(acc[ti][tj], af[ti], bf[tj]);
        }
    };

    const int nk = (mend - mbeg + BKM - 1) / BKM;
    if (nk <= 0) return;
#if MSFWSI_WGRAD_PIPE
    if constexpr (!XPRO) {
        // ---- Pixel loop, fragment reads software-pipelined ACROSS the slab barrier (round 6) ----
        // The round-2..5 loop was: wait + barrier, DMA requests, ALL 16 transposed reads of the slab, `s_waitcnt lgkmcnt(0)`,
        // 8 MFMAs -- every wave's first MFMA of a slab waited out the LDS latency right behind the barrier, and the four
        // waves of a SIMD did so TOGETHER (the barrier puts them in phase): the matrix pipe idled for that latency once per
        // 1024-cycle slab (hipcc's -S: `s_waitcnt lgkmcnt(0)` in front of the first v_mfma of every iteration), and ~20 VALU
        // address additions per slab competed with the MFMAs for issue slots.  Now a slab's two k-groups live in two
        // register sets: the reads of (slab kt, group 1) are issued BEFORE the MFMAs of (kt, group 0), the wave then waits
        // for its DMA pieces of slab kt+1 AND for its own outstanding reads, joins the barrier, issues the DMA requests of
        // slab kt+3 and the reads of (kt+1, group 0), and only then runs the MFMAs of (kt, group 1) -- whose operands were
        // requested a whole MFMA group earlier.  Every MFMA group so finds its operands in registers; the latency of a read is
        // hidden behind the other group's four MFMAs (128 cycles per wave, 512 per SIMD).
        // Stage reuse: the requests of slab kt+3 overwrite the stage of slab kt.  Every wave has waited for ALL its reads
        // of slab kt (lgkmcnt(0)) before the barrier behind which those requests are issued, so three stages suffice.
        // Addresses: the per-lane offset of a fragment inside a stage image is computed once (fa / fb: the block swizzle of
        // a row depends on bits of the row index that neither the k-group nor the +4-row half changes); k-group and half are
        // immediates of the ds_read, the stage one addition per fragment column and slab.
        constexpr int DMA_PER_SLAB = A_IT + B_IT;
        static_assert(DMA_PER_SLAB >= 2 && DMA_PER_SLAB <= 4, "vmcnt literal table");
        constexpr int NST = Cfg::NST;
        static_assert(NST == 3 || NST == 4, "stages");
        // NST == 4 (the 256 x 256 tile, one workgroup per CU: 128 KiB): the requests of slab kt+3 go into the stage of slab
        // kt-1, which every wave left before the PREVIOUS barrier -- so they are issued right after the MFMAs of (kt, group 0)
        // and BEFORE this iteration's wait + barrier.  A wave stalls in the issue of an LDS-DMA piece (60-185 cycles per piece,
        // MI355X_MICROARCH.md); placed behind the barrier, all four waves of a SIMD stood in that stall together with an
        // empty matrix pipe; placed here the stall runs under the wave's own four queued MFMAs (128 cycles of pipe time,
        // 512 per SIMD) and under the barrier wait itself.  Three slabs are in flight instead of two.
        constexpr bool EARLY = NST == 4;
        static_assert(BKM / (2 * VEC) == 2, "two k-groups per slab: two register sets");
        int fa[TI], fb[TJ];
        if constexpr (sizeof(T) == 2) {
            const int li = lane & 15, G = lane >> 4;
            const int kb0 = (G >> 1) * 8 + (li >> 2);
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) fa[ti] = wnat_off<ROWI>(kb0, ((wi * TI + ti) * 32 + (G & 1) * 16 + (li & 3) * 4) * 2);
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj) fb[tj] = wnat_off<ROWJ>(kb0, ((wj * TJ + tj) * 32 + (G & 1) * 16 + (li & 3) * 4) * 2);
        }
        auto read_group = [&](int st, int ks, frag_t(&af)[TI], frag_t(&bf)[TJ]) {
            const char* Ab = As + st * Cfg::A_BYTES;
            const char* Bb = Bs + st * Cfg::B_BYTES;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) {
                    const char* pa = Ab + fa[ti] + ks * 16 * ROWI;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + 4 * ROWI));
                    af[ti] = __builtin_bit_cast(frag_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int tj = 0; tj < TJ; ++tj) {
                    const char* pb = Bb + fb[tj] + ks * 16 * ROWJ;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + 4 * ROWJ));
                    bf[tj] = __builtin_bit_cast(frag_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            } else {
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) af[ti] = read_tr_frag<T, ROWI>(Ab, ks, (wi * TI + ti) * 32, lane);
#pragma unroll
                for (int tj = 0; tj < TJ; ++tj) bf[tj] = read_tr_frag<T, ROWJ>(Bb, ks, (wj * TJ + tj) * 32, lane);
            }
        };
        auto mma_group = [&](const frag_t(&af)[TI], const frag_t(&bf)[TJ]) {
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                for (int tj = 0; tj < TJ; ++tj) mma32<T>(acc[ti][tj], af[ti], bf[tj]);
        };
        // prologue: three slabs requested, slab 0 landed and visible, its group 0 on the way to registers
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < nk) fetch(mbeg + i * BKM, i);
        if (nk >= 3) wait_vmcnt<2 * DMA_PER_SLAB>();
        else if (nk == 2) wait_vmcnt<DMA_PER_SLAB>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        frag_t a0[TI], b0[TJ], a1[TI], b1[TJ];
        read_group(0, 0, a0, b0);
        int st_c = 0;
        int st_f = 3;  // EARLY: the stage the next requests go to (slab kt+3 -> stage (kt+3) mod 4)
        // (the last slab is peeled off: ONE straight-line body, so that hipcc keeps one set of accumulator registers -- with
        //  the final slab as a branch of the loop body it renamed the accumulators per branch and spilled 96 registers)
        for (int kt = 0; kt + 1 < nk; ++kt) {
            read_group(st_c, 1, a1, b1);
            mma_group(a0, b0);
            const int st_n = st_c == NST - 1 ? 0 : st_c + 1;
            if constexpr (EARLY) {
                if (kt + 3 < nk) fetch(mbeg + (kt + 3) * BKM, st_f);
                st_f = st_f == NST - 1 ? 0 : st_f + 1;
                // my DMA pieces of slab kt+1 have landed (slabs kt+2, kt+3, if any, stay in flight); my reads of slab kt are done
                if (kt + 3 < nk) wait_vmcnt<2 * DMA_PER_SLAB>();
                else if (kt + 2 < nk) wait_vmcnt<DMA_PER_SLAB>();
                else wait_vmcnt<0>();
            } else {
                // my DMA pieces of slab kt+1 have landed (slab kt+2, if any, stays in flight); my reads of slab kt are done
                if (kt + 2 < nk) wait_vmcnt<DMA_PER_SLAB>();
                else wait_vmcnt<0>();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (EARLY) {
                read_group(st_n, 0, a0, b0);
                mma_group(a1, b1);
            } else {
#if MSFWSI_FETCH_FIRST
                if (kt + 3 < nk) fetch(mbeg + (kt + 3) * BKM, st_c);
                read_group(st_n, 0, a0, b0);
                mma_group(a1, b1);
#else
                read_group(st_n, 0, a0, b0);
                mma_group(a1, b1);
                if (kt + 3 < nk) fetch(mbeg + (kt + 3) * BKM, st_c);
#endif
            }
            st_c = st_n;
        }
        read_group(st_c, 1, a1, b1);
        mma_group(a0, b0);
        mma_group(a1, b1);
#else
    if constexpr (!XPRO) {
        constexpr int DMA_PER_SLAB = A_IT + B_IT;
        static_assert(DMA_PER_SLAB >= 2 && DMA_PER_SLAB <= 4, "vmcnt literal table");
        // NST stages: slab kt is consumed while kt+1 .. kt+NST-2 are in flight
        constexpr int NST = Cfg::NST, AHEAD = NST - 1;
#pragma unroll
        for (int i = 0; i < AHEAD; ++i)
            if (i < nk) fetch(mbeg + i * BKM, i);
        int st_c = 0, st_f = AHEAD;
        for (int kt = 0; kt < nk; ++kt) {
            // wait until only the slabs after kt are outstanding (DMA pieces complete in order)
            const int rem = min(nk - 1 - kt, AHEAD - 1);
            if (rem >= 2) wait_vmcnt<2 * DMA_PER_SLAB>();
            else if (rem == 1) wait_vmcnt<DMA_PER_SLAB>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // DMA requests of slab kt+NST-1 (its stage was read in kt-1, every wave is past this iteration's barrier) and the
            // MFMAs of slab kt: requests first since the DMA is issued from inline asm (see igemm.hip's main loop and
            // common.h lds_dma16_buf; round 2's "MFMAs first" belonged to the drained pipeline)
#if MSFWSI_FETCH_FIRST
            if (kt + AHEAD < nk) fetch(mbeg + (kt + AHEAD) * BKM, st_f);
            compute(st_c);
#else
            compute(st_c);
            if (kt + AHEAD < nk) fetch(mbeg + (kt + AHEAD) * BKM, st_f);
#endif
            st_c = st_c == NST - 1 ? 0 : st_c + 1;
            st_f = st_f == NST - 1 ? 0 : st_f + 1;
        }
#endif
    } else {
        fetch(mbeg, 0);
        commit(0, mbeg);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) fetch(mbeg + (kt + 1) * BKM, buf ^ 1);
            compute(buf);
            if (kt + 1 < nk) commit(buf ^ 1, mbeg + (kt + 1) * BKM);
            __syncthreads();
        }
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
                if (co < prm.K && j < prm.Jtot) {
                    if (prm.dw64 != nullptr) atomicAdd(prm.dw64 + (long)co * prm.Jtot + j, (double)acc[ti][tj][reg]);
                    else if (prm.store) prm.dw[(long)co * prm.Jtot + j] = acc[ti][tj][reg];
                    else atomicAdd(prm.dw + (long)co * prm.Jtot + j, acc[ti][tj][reg]);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// 64 -> 64 channels, 3x3 / stride 1 / pad 1 (layer1 of every ResNet here): OUTPUT-STATIONARY, persistent workgroups.
//
// The whole gradient (64 x 9 x 64 fp32 = 36 MFMA tiles) lives in the accumulators of ONE 8-wave workgroup: wave w owns
// tap w (four 32x32 tiles), the four tiles of tap 8 go to waves 0-3 -- nine tiles per SIMD.  The workgroup walks a
// contiguous range of 256-position chunks of a ZERO-PADDED raster of the batch (rows of W+1, images of H+1 rows: the
// one pad column / row between neighbours is every tap's out-of-image value, so no tap needs a mask; dY is zero on pad
// positions).  Per chunk, dY (256 rows) and the activation halo (256 + 2(W+1) + 2 rows) are loaded ONCE by plain global
// loads into registers while the previous chunk computes -- optionally with the producer's relu(scale*c + shift) applied
// on the way to LDS, so the activation is never materialised for this kernel either -- and all nine taps read their
// shifted rows of the same halo through the transposed LDS read.  The gather kernel fetched every activation element
// nine times through L2 -> LDS on a 64 x 128 tile (43 FLOP per ingested byte: 594 TFLOP/s of a 660 roof).
// ---------------------------------------------------------------------------------------------------------------
struct WgWParams {
    const void* x;   // [N][H][W][64]  activation (or the producer's raw conv output with pro_*)
    const void* dy;  // [N][H][W][64]
    float* dw;       // [64][3][3][64]
    const float* pro_scale;
    const float* pro_shift;
    int N, H, W;
    long npos;       // N * (H+1) * (W+1) positions of the padded raster
    int nchunks, chunks_per_wg;
    FastDiv div_img, div_wp;
};

struct WgWCfg {
    static constexpr int BP = 256, NW = 8, MAXW = 56;  // (64 would need a seventh staged chunk per thread: spills, 1.10 -> 1.95 ms)
    static constexpr int HALO = ((BP + 2 * (MAXW + 1) + 2 + 63) / 64) * 64;  // 384
    static constexpr int DY_BYTES = BP * 128, A_BYTES = HALO * 128;
    static constexpr int LDS_BYTES = DY_BYTES + A_BYTES;                      // 80 KiB
    static constexpr int DY_LOADS = BP * 8 / (64 * NW), A_LOADS = HALO * 8 / (64 * NW);  // 4, 6 chunks per thread
};

// fragment of a natural-layout [row][128 B] LDS image: 32 columns from col0, 16 rows from ABSOLUTE row `row0`
// (the block swizzle is a function of the absolute row, so a tap's shifted view needs no copy)
template <typename T>
__device__ __forceinline__ typename WFrag<T>::type read_tr_rows(const char* img, int row0, int col0, int lane) {
    typedef typename WFrag<T>::type frag_t;
    const int li = lane & 15, G = lane >> 4;
    const int q = li >> 2, p = li & 3;
    const int kb = row0 + (G >> 1) * 8 + q;
    const int cb = (col0 + (G & 1) * 16 + p * 4) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(img + wnat_off<128>(kb, cb)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(img + wnat_off<128>(kb + 4, cb)));
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(frag_t, both);
}

template <typename T>
__global__ __launch_bounds__(512) void wgrad_os_kernel(const WgWParams prm) {
    typedef WgWCfg Cfg;
    constexpr int BP = Cfg::BP, NT = 64 * Cfg::NW, DL = Cfg::DY_LOADS, AL = Cfg::A_LOADS;
    constexpr int VEC = 8;
    static_assert(sizeof(T) == 2, "2-byte storage types");
    typedef typename WFrag<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ds = smem;                   // [BP][128 B]   dY rows (position-major)
    char* As = smem + Cfg::DY_BYTES;   // [HALO][128 B] activation rows: row hr <-> position chunk*BP - Wp - 1 + hr

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int W = prm.W, H = prm.H, Wp = W + 1;
    const T* __restrict__ x = reinterpret_cast<const T*>(prm.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(prm.dy);

    const int c_beg = blockIdx.x * prm.chunks_per_wg;
    const int c_end = min(prm.nchunks, c_beg + prm.chunks_per_wg);
    if (c_beg >= c_end) return;

    // staging map: chunk i*512 + tid -> (row, 16-byte column chunk); the 8 channels are the same for every i
    const int c8 = tid & 7;
    int d_lds[DL], a_lds[AL];
#pragma unroll
    for (int i = 0; i < DL; ++i) {
        const int row = (i * NT + tid) >> 3;
        d_lds[i] = wnat_off<128>(row, c8 * 16);
    }
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int row = (i * NT + tid) >> 3;
        a_lds[i] = wnat_off<128>(row, c8 * 16);
    }
    const bool has_pro = prm.pro_scale != nullptr;
    float qsc[VEC], qsh[VEC];
    if (has_pro) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            qsc[e] = prm.pro_scale[c8 * VEC + e];
            qsh[e] = prm.pro_shift[c8 * VEC + e];
        }
    }
    uint4 d_reg[DL], a_reg[AL];
    unsigned a_valid = 0;
    // (image, row, column) of every staged slot's position, advanced by BP positions per chunk with carries instead of
    // two divisions per slot and chunk (the staging arithmetic stood in front of every chunk's MFMAs)
    int d_img[DL], d_y[DL], d_x[DL], a_img[AL], a_y[AL], a_x[AL];
    const int Hp = H + 1;
    auto locate = [&](long q, int& img, int& y, int& xx) {  // q may be negative (before the batch): image -1
        if (q < 0) {
            img = -1;
            const long r = q + (long)Hp * Wp;  // q >= -(Wp+1) > -Hp*Wp
            y = (int)(r / Wp);
            xx = (int)(r - (long)y * Wp);
            return;
        }
        img = (int)fast_div((unsigned)q, prm.div_img);
        const unsigned rem = (unsigned)q - (unsigned)img * (unsigned)(Hp * Wp);
        y = (int)fast_div(rem, prm.div_wp);
        xx = (int)(rem - (unsigned)y * (unsigned)Wp);
    };
    const int adv_y = BP / Wp, adv_x = BP - adv_y * Wp;  // BP positions = adv_y rows + adv_x columns
    auto advance = [&](int& img, int& y, int& xx) {
        xx += adv_x;
        y += adv_y;
        if (xx >= Wp) {
            xx -= Wp;
            ++y;
        }
        while (y >= Hp) {  // small images: a chunk may span several
            y -= Hp;
            ++img;
        }
    };
#pragma unroll
    for (int i = 0; i < DL; ++i) locate((long)c_beg * BP + ((i * NT + tid) >> 3), d_img[i], d_y[i], d_x[i]);
#pragma unroll
    for (int i = 0; i < AL; ++i) locate((long)c_beg * BP - Wp - 1 + ((i * NT + tid) >> 3), a_img[i], a_y[i], a_x[i]);
    auto load_chunk = [&]() {  // loads the chunk the slot positions point at, then advances them to the next
#pragma unroll
        for (int i = 0; i < DL; ++i) {
            d_reg[i] = make_uint4(0, 0, 0, 0);
            if ((unsigned)d_img[i] < (unsigned)prm.N && d_y[i] < H && d_x[i] < W)
                d_reg[i] = *reinterpret_cast<const uint4*>(dy + (((long)d_img[i] * H + d_y[i]) * W + d_x[i]) * 64 + c8 * VEC);
            advance(d_img[i], d_y[i], d_x[i]);
        }
        a_valid = 0;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const int hr = (i * NT + tid) >> 3;
            a_reg[i] = make_uint4(0, 0, 0, 0);
            if (hr < BP + 2 * Wp + 2 && (unsigned)a_img[i] < (unsigned)prm.N && a_y[i] < H && a_x[i] < W) {
                a_reg[i] = *reinterpret_cast<const uint4*>(x + (((long)a_img[i] * H + a_y[i]) * W + a_x[i]) * 64 + c8 * VEC);
                a_valid |= 1u << i;
            }
            advance(a_img[i], a_y[i], a_x[i]);
        }
    };

    // tiles: acc[cb*2+ib] = tap `wave`, output-channel block cb, input-channel block ib; acc[4] = tap 8, block (wave>>1, wave&1)
    f32x16 acc[5];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[a][j] = 0.f;
    const bool extra = wave < 4;
    const int roff_w = (wave / 3) * Wp + (wave % 3);  // halo row of a position's tap-`wave` neighbour: + r*Wp + s
    const int roff_8 = 2 * Wp + 2;
    const int xcb = wave >> 1, xib = wave & 1;

    load_chunk();
    for (int chunk = c_beg; chunk < c_end; ++chunk) {
        __syncthreads();  // every wave has read the previous chunk's last fragment
#pragma unroll
        for (int i = 0; i < DL; ++i) *reinterpret_cast<uint4*>(Ds + d_lds[i]) = d_reg[i];
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
        if (chunk + 1 < c_end) load_chunk();  // in flight while the MFMAs run
        // fragments of step ks+1 are requested before the MFMAs of step ks (two waves per SIMD and four or five MFMAs
        // per step do not cover an LDS round trip: measured 30 % MFMA utilisation without this)
        frag_t af[2][2], bf[2][2], bx[2];
        auto fetch_frags = [&](int ks, int s) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) af[s][cb] = read_tr_rows<T>(Ds, ks * 16, cb * 32, lane);
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) bf[s][ib] = read_tr_rows<T>(As, ks * 16 + roff_w, ib * 32, lane);
            if (extra) bx[s] = read_tr_rows<T>(As, ks * 16 + roff_8, xib * 32, lane);
        };
        auto mma_step = [&](int s) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) mma32<T>(acc[cb * 2 + ib], af[s][cb], bf[s][ib]);
            if (extra) mma32<T>(acc[4], xcb ? af[s][1] : af[s][0], bx[s]);
        };
        fetch_frags(0, 0);
#pragma unroll 1  // a real loop of step pairs: the two fragment sets are static registers, nothing else is hoisted
        for (int kk = 0; kk < BP / 32; ++kk) {
            fetch_frags(2 * kk + 1, 1);
            mma_step(0);
            if (kk + 1 < BP / 32) fetch_frags(2 * kk + 2, 0);
            mma_step(1);
        }
    }

    // dW[co][t][ci] += partial: lane -> ci (contiguous), registers -> co
    const int lane_e = fresh_lane();  // re-derived (common.h): kept since kernel entry it cost 16-20 bytes of scratch per lane
    const int l31_e = lane_e & 31, lh_e = lane_e >> 5;
    auto flush = [&](const f32x16& a, int t, int cb, int ib) {
        const int j = t * 64 + ib * 32 + l31_e;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh_e;
            atomicAdd(prm.dw + (long)co * 576 + j, a[reg]);
        }
    };
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) flush(acc[cb * 2 + ib], wave, cb, ib);
    if (extra) flush(acc[4], 8, xcb, xib);
}

msfwsi_tunable g_wgrad_os{1};  // msfwsi_set_tuning(10, .): 0 = the 64 -> 64 3x3 weight gradient on the gather kernel
msfwsi_tunable g_wgrad_os_min_pos{32L * 256 * 256};  // msfwsi_set_tuning(11, .): smallest padded raster the kernel takes

bool wgrad_os_ok(const msfwsi_conv_desc* d) {
    return g_wgrad_os && d->dtype != MSFWSI_DT_F32 && d->C == 64 && d->K == 64 && d->R == 3 && d->S == 3 &&
           d->stride == 1 && d->pad == 1 && d->P == d->H && d->Q == d->W && d->W <= WgWCfg::MAXW && d->W >= 2 &&
           (long)d->N * (d->H + 1) * (d->W + 1) <= 0x7fffffffL &&
           // each workgroup ends with 36 864 atomic adds: worth it from ~32 chunks per workgroup (measured: 13 chunks
           // 0.114 vs 0.101 ms for the gather kernel, 203 chunks 1.10 vs 1.66 ms)
           (long)d->N * (d->H + 1) * (d->W + 1) >= g_wgrad_os_min_pos;
}

template <typename T>
int launch_wgrad_os(const msfwsi_conv_desc* d, const void* x, const void* dy, float* dw, const float* ps,
                    const float* psh, hipStream_t stream) {
    WgWParams prm{};
    prm.x = x; prm.dy = dy; prm.dw = dw; prm.pro_scale = ps; prm.pro_shift = psh;
    prm.N = d->N; prm.H = d->H; prm.W = d->W;
    prm.npos = (long)d->N * (d->H + 1) * (d->W + 1);
    prm.nchunks = (int)((prm.npos + WgWCfg::BP - 1) / WgWCfg::BP);
    prm.div_img = make_fastdiv((unsigned)((d->H + 1) * (d->W + 1)));
    prm.div_wp = make_fastdiv((unsigned)(d->W + 1));
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
        ncu = 256;
    (void)hipGetLastError();
    prm.chunks_per_wg = (prm.nchunks + ncu - 1) / ncu;
    const int nblk = (prm.nchunks + prm.chunks_per_wg - 1) / prm.chunks_per_wg;
    auto kern = wgrad_os_kernel<T>;
    static bool attr_done = false;
    if (!attr_done) {
        if (int e = msfwsi_raise_lds(reinterpret_cast<const void*>(kern), WgWCfg::LDS_BYTES)) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), WgWCfg::LDS_BYTES, stream, prm);
    return msfwsi_launch_status();
}

msfwsi_tunable g_wgrad_lin{1};  // msfwsi_set_tuning(2, .): 0 = always the generic staging
// msfwsi_set_tuning(15, .): cap on the pixel splits of the gather weight-gradient kernel (0 = none).  With 1 every
// gradient tile is summed by ONE workgroup in pixel order: the step is then reproducible bit for bit (the splits add
// their partial sums with fp32 atomics in arrival order, 4e-7 jitter per tensor) -- what a chaotic 30-step test needs
msfwsi_tunable g_wgrad_max_splits{0};
msfwsi_tunable g_wgrad_big{1};  // msfwsi_set_tuning(6, .): 0 = never the 256 x 256 tile

// workgroups of `kern` that fit the device at once (all splits carry equal work, so a grid that overshoots this by
// one workgroup costs a whole second round: measured 0.75 vs 0.56 ms for 1025 vs 1020 workgroups)
template <typename K>
int resident_slots(K kern, int threads, int lds_bytes) {
    int per_cu = 0, dev = 0, ncu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, (size_t)lds_bytes) != hipSuccess) per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        ncu = 0;
    (void)hipGetLastError();
    return per_cu > 0 && ncu > 0 ? per_cu * ncu : 1024;
}

template <typename T, int BI, int BJ>
int launch_wgrad(WgradParams& prm, int target_blocks, hipStream_t stream) {
    typedef WgradCfg<T, BI, BJ> Cfg;
    prm.ntile_i = (prm.K + BI - 1) / BI;
    const int ntile_j = (prm.Jtot + BJ - 1) / BJ;
    const long tiles = (long)prm.ntile_i * ntile_j;
    if (tiles > 0x7fffffffL) return MSFWSI_EINVAL;
    constexpr int lds_pro = WgradCfg<T, BI, BJ, true>::LDS_BYTES;
    constexpr int lds_dma = WgradCfg<T, BI, BJ, false>::LDS_BYTES;
    constexpr int threads = 64 * Cfg::NW;
    const bool pro = prm.pro_scale != nullptr;
    if constexpr (lds_dma > 65536 || lds_pro > 65536) {  // more than the default dynamic LDS limit: raised once per kernel
        if constexpr (lds_pro > 65536) {
            if (int e = msfwsi_raise_lds(reinterpret_cast<const void*>(wgrad_kernel<T, BI, BJ, true, false>), lds_pro)) return e;
        }
        if (int e = msfwsi_raise_lds(reinterpret_cast<const void*>(wgrad_kernel<T, BI, BJ, false, true>), lds_dma)) return e;
        if (int e = msfwsi_raise_lds(reinterpret_cast<const void*>(wgrad_kernel<T, BI, BJ, false, false>), lds_dma)) return e;
    }
    // pixel splits: fill the resident workgroup slots once, never overshoot them (target_blocks > 0 overrides)
    long splits;
    if (target_blocks > 0) {
        splits = (target_blocks + tiles - 1) / tiles;
    } else {
        // same occupancy for the generic and the linear DMA instance; two host threads of the process may launch at once
        // (Engine._run_views): relaxed atomics, both would store the same value
        static std::atomic<int> slots_pro{0}, slots_dma{0};
        std::atomic<int>& cache = pro ? slots_pro : slots_dma;
        int slots = cache.load(std::memory_order_relaxed);
        if (slots == 0) {
            slots = pro ? resident_slots(wgrad_kernel<T, BI, BJ, true, false>, threads, lds_pro)
                        : resident_slots(wgrad_kernel<T, BI, BJ, false, true>, threads, lds_dma);
            cache.store(slots, std::memory_order_relaxed);
        }
        splits = slots / tiles;
        // Small launches (the context passes, N = 256): every split ends with an atomic add of its whole tile (fp64 for a
        // Gram matrix) -- at 196 rows per split the adds of a 256 x 256 tile move five times the bytes of the operands.  At
        // least ~1024 rows per split, but never fewer than 64 workgroups (`profiles/r05_kbench_gram_splits.txt`: the
        // 14 x 14 x 256 Gram launch 0.138 -> 0.055 ms, 7 x 7 x 512 0.096 -> 0.047).  Launches of the heads (M < 8192) keep
        // the old rule.
        if (prm.M >= 8192) {
            long want = prm.M / 1024;
            const long floor_splits = (64 + tiles - 1) / tiles;
            if (want < floor_splits) want = floor_splits;
            if (splits > want) splits = want;
        }
    }
    const long max_splits = (prm.M + Cfg::BKM - 1) / Cfg::BKM;
    if (splits > max_splits) splits = max_splits;
    if (g_wgrad_max_splits > 0 && splits > g_wgrad_max_splits) splits = g_wgrad_max_splits;
    if (prm.store) splits = 1;  // the tile is written, not added: one workgroup owns it
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    long rows = (prm.M + splits - 1) / splits;
    rows = (rows + Cfg::BKM - 1) / Cfg::BKM * Cfg::BKM;
    splits = (prm.M + rows - 1) / rows;
    prm.rows_per_split = (int)rows;
    // linear fast path: stride 1, same-size output, taps at most one pixel outside, offsets within 2 GiB windows
    const bool lin = g_wgrad_lin && prm.stride == 1 && prm.P == prm.H && prm.Q == prm.W && prm.pad <= 1 &&
                     prm.R <= 3 && prm.S <= 3 && prm.R == 2 * prm.pad + 1 && prm.S == 2 * prm.pad + 1 &&
                     (rows + 2L * prm.W + 4) * (prm.C > prm.K ? prm.C : prm.K) * (long)sizeof(T) < 0x7fffffffL;
    if (pro)
        hipLaunchKernelGGL((wgrad_kernel<T, BI, BJ, true, false>), dim3((unsigned)tiles, (unsigned)splits), dim3(threads),
                           lds_pro, stream, prm);
    else if (lin)
        hipLaunchKernelGGL((wgrad_kernel<T, BI, BJ, false, true>), dim3((unsigned)tiles, (unsigned)splits), dim3(threads),
                           lds_dma, stream, prm);
    else
        hipLaunchKernelGGL((wgrad_kernel<T, BI, BJ, false, false>), dim3((unsigned)tiles, (unsigned)splits), dim3(threads),
                           lds_dma, stream, prm);
    return msfwsi_launch_status();
}

}  // namespace

extern "C" __attribute__((visibility("hidden"))) long msfwsi_wgrad_set_lin(long v, int write) {
    const long old = g_wgrad_lin;
    if (write) g_wgrad_lin = v;
    return old;
}
extern "C" __attribute__((visibility("hidden"))) long msfwsi_wgrad_set_max_splits(long v, int write) {
    const long old = g_wgrad_max_splits;
    if (write) g_wgrad_max_splits = v;
    return old;
}
extern "C" __attribute__((visibility("hidden"))) long msfwsi_wgrad_set_big(long v, int write) {
    const long old = g_wgrad_big;
    if (write) g_wgrad_big = v;
    return old;
}
extern "C" __attribute__((visibility("hidden"))) long msfwsi_wgrad_set_os(long v, int write) {
    const long old = g_wgrad_os;
    if (write) g_wgrad_os = v;
    return old;
}
extern "C" __attribute__((visibility("hidden"))) long msfwsi_wgrad_set_os_min(long v, int write) {
    const long old = g_wgrad_os_min_pos;
    if (write) g_wgrad_os_min_pos = v;
    return old;
}

// 1 if msfwsi_conv_wgrad serves this geometry with the output-stationary kernel (whose BatchNorm+ReLU prologue is free:
// callers then pass pro_scale / pro_shift instead of materialising the activation)
extern "C" int msfwsi_conv_wgrad_stationary(const msfwsi_conv_desc* d) { return d != nullptr && wgrad_os_ok(d) ? 1 : 0; }

extern "C" __attribute__((visibility("hidden"))) int msfwsi_stem_os_wgrad(const msfwsi_conv_desc* d, const void* x,
                                                                         const void* dy, float* dw, const void* c0,
                                                                         const float* k1, const float* k2,
                                                                         const float* k3, void* stream);

// Weight gradient of the space-to-depth stem with the BatchNorm backward of its output applied on the fly:
// dw += (k1*g + k2*c0 + k3)^T x, the bracket rounded to the storage type as msfwsi_bn_bwd_apply rounds it.
// MSFWSI_EUNSUPPORTED where the output-stationary stem kernel does not apply (the caller then runs msfwsi_bn_bwd_apply
// + msfwsi_conv_wgrad).
extern "C" int msfwsi_stem_wgrad_bnbwd(const msfwsi_conv_desc* d, const void* x, const void* g, const void* c0,
                                       const float* k1, const float* k2, const float* k3, float* dw, void* stream) {
    if (d == nullptr || x == nullptr || g == nullptr || c0 == nullptr || k1 == nullptr || k2 == nullptr || k3 == nullptr ||
        dw == nullptr)
        return MSFWSI_EINVAL;
    if (!msfwsi_dtype_ok(d->dtype)) return MSFWSI_EUNSUPPORTED;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0) return MSFWSI_EINVAL;
    return msfwsi_stem_os_wgrad(d, x, g, dw, c0, k1, k2, k3, stream);
}

static int wgrad_generic(const msfwsi_conv_desc* d, const void* x, const void* dy, float* dw, double* dw64,
                         const float* pro_scale, const float* pro_shift, int target_blocks, hipStream_t st,
                         int store = 0, void* aout = nullptr) {
    WgradParams prm{};
    prm.x = x; prm.dy = dy; prm.dw = dw; prm.dw64 = dw64; prm.store = store;
    prm.pro_scale = pro_scale; prm.pro_shift = pro_shift; prm.aout = aout;
    prm.N = d->N; prm.H = d->H; prm.W = d->W; prm.C = d->C;
    prm.P = d->P; prm.Q = d->Q; prm.K = d->K;
    prm.R = d->R; prm.S = d->S; prm.stride = d->stride; prm.pad = d->pad;
    prm.M = d->N * d->P * d->Q;
    prm.Jtot = d->R * d->S * d->C;
    prm.div_pq = make_fastdiv((unsigned)(d->P * d->Q));
    prm.div_q = make_fastdiv((unsigned)d->Q);
    const bool small_i = d->K <= 64;
    const bool small_j = prm.Jtot <= 64;
    // 2-byte types, both extents in whole 256-wide tiles, no recomputed prologue: the eight-wave tile
    const bool big = g_wgrad_big && d->dtype != MSFWSI_DT_F32 && d->K % 256 == 0 && prm.Jtot % 256 == 0 &&
                     pro_scale == nullptr;
    if (big) {
        if (d->dtype == MSFWSI_DT_BF16) return launch_wgrad<__bf16, 256, 256>(prm, target_blocks, st);
        return launch_wgrad<_Float16, 256, 256>(prm, target_blocks, st);
    }
    MSFWSI_WITH_T(d->dtype, {
        if (small_i && small_j) return launch_wgrad<T, 64, 64>(prm, target_blocks, st);
        if (small_i) return launch_wgrad<T, 64, 128>(prm, target_blocks, st);
        if (small_j) return launch_wgrad<T, 128, 64>(prm, target_blocks, st);
        return launch_wgrad<T, 128, 128>(prm, target_blocks, st);
    });
    return MSFWSI_EINVAL;
}

extern "C" int msfwsi_conv_wgrad(const msfwsi_conv_desc* d, const void* x, const void* dy, float* dw,
                                 const float* pro_scale, const float* pro_shift, int target_blocks,
                                 void* stream) {
    if (d == nullptr || x == nullptr || dy == nullptr || dw == nullptr) return MSFWSI_EINVAL;
    if (!msfwsi_dtype_ok(d->dtype)) return MSFWSI_EUNSUPPORTED;
    const int vec = msfwsi_vec_of(d->dtype);
    if (d->C % vec != 0 || d->K % vec != 0) return MSFWSI_EUNSUPPORTED;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->P <= 0 || d->Q <= 0 || d->R <= 0 || d->S <= 0) return MSFWSI_EINVAL;
    if ((pro_scale == nullptr) != (pro_shift == nullptr)) return MSFWSI_EINVAL;
    if ((long)d->N * d->P * d->Q > 0x7fffffffL || (long)d->N * d->H * d->W > 0x7fffffffL) return MSFWSI_EINVAL;
    if (pro_scale == nullptr) {  // the space-to-depth stem (stem.hip)
        const int rc = msfwsi_stem_os_wgrad(d, x, dy, dw, nullptr, nullptr, nullptr, nullptr, stream);
        if (rc != MSFWSI_EUNSUPPORTED) return rc;
    }
    if (wgrad_os_ok(d)) {
        hipStream_t st0 = reinterpret_cast<hipStream_t>(stream);
        if (d->dtype == MSFWSI_DT_BF16) return launch_wgrad_os<__bf16>(d, x, dy, dw, pro_scale, pro_shift, st0);
        return launch_wgrad_os<_Float16>(d, x, dy, dw, pro_scale, pro_shift, st0);
    }
    return wgrad_generic(d, x, dy, dw, nullptr, pro_scale, pro_shift, target_blocks, reinterpret_cast<hipStream_t>(stream));
}

// msfwsi_conv_wgrad for a 1x1 / stride 1 conv whose operand is the producer's raw output under its BatchNorm + ReLU, with the
// normalised operand as a by-product: the register staging that applies relu(scale * x + shift) also stores it to act_out
// (bit for bit msfwsi_bn_act's output).  The folded Bottleneck tail's M = g^T a2 launch uses it so that a2 -- which the
// two-source input gradient needs materialised -- costs no pass of its own (src/models/resnet.py:128-131 backwards).
extern "C" int msfwsi_conv_wgrad_act(const msfwsi_conv_desc* d, const void* x, const void* dy, float* dw,
                                     const float* pro_scale, const float* pro_shift, void* act_out, int target_blocks,
                                     void* stream) {
    if (d == nullptr || x == nullptr || dy == nullptr || dw == nullptr || pro_scale == nullptr || pro_shift == nullptr ||
        act_out == nullptr || act_out == x)
        return MSFWSI_EINVAL;
    if (d->dtype != MSFWSI_DT_BF16 && d->dtype != MSFWSI_DT_F16) return MSFWSI_EUNSUPPORTED;
    if (d->R != 1 || d->S != 1 || d->stride != 1 || d->pad != 0 || d->P != d->H || d->Q != d->W) return MSFWSI_EUNSUPPORTED;
    if (d->C % 8 != 0 || d->K % 8 != 0) return MSFWSI_EUNSUPPORTED;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0) return MSFWSI_EINVAL;
    if ((long)d->N * d->H * d->W > 0x7fffffffL) return MSFWSI_EINVAL;
    return wgrad_generic(d, x, dy, dw, nullptr, pro_scale, pro_shift, target_blocks, reinterpret_cast<hipStream_t>(stream), 0,
                         act_out);
}

// dw = dy^T x, STORED: every gradient tile is computed by one workgroup over all pixels and written once -- no atomics,
// no read-modify-write of dw, dw need not be zero beforehand.  For weight gradients that receive exactly one launch per
// step and have few rows: the heads' Linear layers (both views stacked: 512 / 8192 rows), above all the fuser's
// 18432 x 18432 matrices (1.36 GB of fp32 each: accumulated, they are read and written; stored, written -- and the
// step's clear of the flat gradient buffer skips them).
extern "C" int msfwsi_conv_wgrad_store(const msfwsi_conv_desc* d, const void* x, const void* dy, float* dw, void* stream) {
    if (d == nullptr || x == nullptr || dy == nullptr || dw == nullptr) return MSFWSI_EINVAL;
    if (!msfwsi_dtype_ok(d->dtype)) return MSFWSI_EUNSUPPORTED;
    const int vec = msfwsi_vec_of(d->dtype);
    if (d->C % vec != 0 || d->K % vec != 0) return MSFWSI_EUNSUPPORTED;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->P <= 0 || d->Q <= 0 || d->R <= 0 || d->S <= 0) return MSFWSI_EINVAL;
    if ((long)d->N * d->P * d->Q > 0x7fffffffL || (long)d->N * d->H * d->W > 0x7fffffffL) return MSFWSI_EINVAL;
    return wgrad_generic(d, x, dy, dw, nullptr, nullptr, nullptr, 0, reinterpret_cast<hipStream_t>(stream), 1);
}

// Gram matrix A += a^T a of an NHWC activation (d: the 1x1 / stride-1 conv C -> C whose "weight gradient" with x = dy = a
// it is), accumulated in fp64.  The folded Bottleneck tail derives bn3's batch statistics from it: with fp32
// atomics the order of the pixel splits moved the statistics by ~1e-6 from run to run, which 16-bit storage and a deep
// BatchNorm chain amplify to a visibly different step; in fp64 the forward is reproducible to the last bit in practice.
extern "C" int msfwsi_gram(const msfwsi_conv_desc* d, const void* a, double* A64, void* stream) {
    if (d == nullptr || a == nullptr || A64 == nullptr) return MSFWSI_EINVAL;
    if (!msfwsi_dtype_ok(d->dtype)) return MSFWSI_EUNSUPPORTED;
    const int vec = msfwsi_vec_of(d->dtype);
    if (d->R != 1 || d->S != 1 || d->stride != 1 || d->pad != 0 || d->C != d->K || d->C % vec != 0) return MSFWSI_EUNSUPPORTED;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->P != d->H || d->Q != d->W) return MSFWSI_EINVAL;
    if ((long)d->N * d->H * d->W > 0x7fffffffL) return MSFWSI_EINVAL;
    return wgrad_generic(d, a, a, nullptr, A64, nullptr, nullptr, 0, reinterpret_cast<hipStream_t>(stream));
}
