// Image-stationary 3x3 / stride 1 / pad 1 convolution for the deep layers of the encoders on gfx950 (16-bit storage):
// conv2 of the Bottlenecks of layer2 (28x28, 128 -> 128) and layer3 (14x14, 256 -> 256), forward and input gradient
// (reference src/models/resnet.py:25-28,128; backward through scaler.scale(loss).backward(), tools/ssl_train.py:472).
//
// Why a third 3x3 kernel.  The gather kernel (igemm.hip) re-fetches every activation row NINE times from L2 into LDS -- one
// `buffer_load ... lds` piece per (tap, 64-byte slab), a workgroup barrier per slab -- and runs these layers at 0.9-1.0
// PFLOP/s, 40 % MFMA utilisation: its waves stall in the issue of the DMA pieces and at 72-144 barriers per tile.  Here
//   * a workgroup owns a BAND of an image: BH full rows (the whole 14x14 image; 7 rows of a 28x28 one) = 196 output pixels
//     = seven 32-row MFMA tiles.  The band with its halo ((BH+2) x (W+2) positions, zero outside the image) is read from
//     HBM ONCE, optionally transformed on the way (the producer's BatchNorm+ReLU; or the BatchNorm backward dc = k1 g + k2
//     c + k3, written back for the weight gradient), and parked in LDS: a filter tap is then nothing but a constant added
//     to the fragment's LDS address;
//   * after one barrier every wave runs alone: wave w owns output channels 32w .. 32w+31 of all 196 pixels (7 accumulator
//     tiles); its weight fragments stream from L2 in MFMA order (msfwsi_img3x3_pack_weights: 1 KiB per fragment, 9 * C/16
//     of them, consumed strictly in sequence) through a four-deep register ring of hand-counted asm loads (handload.h);
//     7 MFMAs per fragment, one ds_read_b128 per MFMA, no barrier, no DMA, no address arithmetic beyond one XOR per read.
// Weight traffic: every workgroup streams the whole filter (1.2 MB at 256 channels) from L2 -- 4.8 GB per N = 4096 launch
// at ~15 TB/s, below the MFMA time; activations: one pass.
#include "common.h"
#include "handload.h"
#include "../../include/msfwsi_hip.h"

#ifndef MSFWSI_IMG_TN
#define MSFWSI_IMG_TN 1  // 32-channel blocks per wave: 1 = eight / four waves of 32 channels, two per SIMD; 2 = four / two
#endif                   // waves of 64 (half the LDS reads per MFMA, one wave per SIMD: slower but for the 14x14 forward)
#ifndef MSFWSI_IMG_STAGGER
#define MSFWSI_IMG_STAGGER 40  // x 64 clocks between the eight start phases of the first round (0: off); -6 % on the 14x14 forward, neutral elsewhere (profiles/r05_img3_stagger.txt)
#endif
#ifndef MSFWSI_IMG_ABLATE
#define MSFWSI_IMG_ABLATE 0  // diagnostic builds (tools/build_variant.sh), WRONG RESULTS: 1 no staging loads, 2 one tap instead
#endif                       // of nine, 4 no output stores / mask loads, 8 no weight re-loads, 16 no LDS reads in the k loop, 32 epilogue accesses as 8 rows x 128 B

namespace {

struct Img3Params {
    const void* src;    // [N][H][W][C]: PRO 0 the operand, PRO 1 the producer's raw conv output, PRO 2 the gated gradient g
    const void* src_c;  // PRO 2: the raw conv output whose BatchNorm is differentiated
    const float* p0;    // PRO 1: scale  PRO 2: k1
    const float* p1;    // PRO 1: shift  PRO 2: k2
    const float* p2;    //               PRO 2: k3
    void* aout;         // PRO 2, nullable: dc written back [N][H][W][C]
    const void* wpk;    // [KO/32][9 * C/16][64][8]
    void* out;          // [N][H][W][KO]
    double* stats;      // forward: [nshard][2][KO] += {sum y, sum y^2} of the stored outputs; gradient: {sum g, sum g*c}
    int nshard;
    const void* mask_c;       // gradient: [N][H][W][KO] raw conv output whose BatchNorm+ReLU gates it, nullable
    const float* mask_scale;
    const float* mask_shift;
    void* act_out;            // gradient, nullable (needs mask_c): relu(mask_scale * mask_c + mask_shift) [N][H][W][KO], = msfwsi_bn_act
    int N, H;
};

// Chunk XOR of a staged pixel row.  A ds_read_b128 is served in four groups of 16 lanes (MI355X guide, LDS table), one LDS
// cycle per group when its 16 lanes fall on 16 different 16-byte slots of the 256-byte bank line.  A group's lanes hold 16
// output pixels with raster indices p that are distinct mod 16, and tap (r, s) sends pixel p to the staged row of LINEAR
// index G = p + r * IW + s (pitch IW, not the padded pitch IW + 2): keyed on G, the XOR is distinct across the group for
// every tap.  (Keyed on the padded position, as in round 5's first build, two image rows met on the same slots: 45 % of the
// LDS cycles of the 14x14 kernel were conflict cycles, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.)  Rows of >= 256 bytes
// rotate through 16 chunk positions; at 64 channels two 128-byte rows share a bank line, told apart by G's parity (= the
// position's parity, IW being even), and rotate together through the 8 chunks of a row.
template <int C>
__device__ __forceinline__ int img_swz(int G) { return C >= 128 ? (G & 15) : ((G >> 1) & 7); }

template <typename T, int C, int KO, int BH, int IW, int PRO, bool DGRAD, int TN>
__global__ __launch_bounds__(KO * 2 / TN, 2 / TN) void img3x3_kernel(const Img3Params prm) {
    constexpr int NW = KO / (32 * TN), NT = 64 * NW;
    constexpr int PW = IW + 2, PP = (BH + 2) * PW;  // padded positions of the band
    constexpr int CPR = C / 8, ROWB = C * 2;
    constexpr int NCHT = (PP * CPR + NT - 1) / NT;   // 16-byte chunks staged per thread,
    constexpr int NPASS = (NCHT + 15) / 16;          // at most 16 in flight at a time
    constexpr int NCH = (NCHT + NPASS - 1) / NPASS;
    constexpr int MB = BH * IW;                      // output pixels of the band
    constexpr int TM = (MB + 31) / 32;
    constexpr int KC = C / 16;                       // k steps per tap
    constexpr int R = 4;                             // weight-fragment ring
    constexpr int SCR_PITCH = 80, SCR_BYTES = 32 * SCR_PITCH;
    static_assert(C % 64 == 0 && IW % 2 == 0 && KO % (32 * TN) == 0 && NT % CPR == 0 && KC % R == 0 && (TN == 1 || TN == 2), "image kernel geometry");
    typedef typename MmaFrag<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* img = smem;  // [PP][ROWB], chunk index XOR-swizzled by the position
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    char* scratch = smem + PP * ROWB + wave * SCR_BYTES;

    // Bands in memory order per XCD: workgroup b runs on XCD b % 8 (round-robin dispatch), so XCD x takes the x-th eighth of
    // the bands, in order -- neighbouring bands (which share two halo rows) meet in ONE L2 at about the same time.
    unsigned bid = blockIdx.x;
    if ((gridDim.x & 7u) == 0) bid = (bid & 7u) * (gridDim.x >> 3) + (bid >> 3);
    const int nband = prm.H / BH;
    const int image = bid / nband, band = bid - image * nband;
    const int row0 = band * BH;                                   // first image row of the band
    const long pix0 = ((long)image * prm.H + row0) * IW;          // first output pixel (bands are contiguous in memory)

#if MSFWSI_IMG_STAGGER
    // Workgroups of equal length that start together stay together: all 256 CUs stage at once (an HBM burst nobody computes
    // under), then all compute (HBM idle).  The first round starts in eight phases, MSFWSI_IMG_STAGGER * 64 clocks apart.
    if (blockIdx.x < 256u * (unsigned)((160 * 1024) / (PP * ROWB + NW * SCR_BYTES))) {
        const int g = (blockIdx.x >> 3) & 7;
        for (int i = 0; i < g; ++i) __builtin_amdgcn_s_sleep(MSFWSI_IMG_STAGGER);
    }
#endif
    // ---------------- stage the band + halo: every chunk requested up front (clamped addresses, no branches) ----------------
    {
        const int cc = tid % CPR;
        const char* src_img = reinterpret_cast<const char*>(prm.src) + (long)image * prm.H * IW * ROWB;
        const char* srcc_img = PRO == 2 ? reinterpret_cast<const char*>(prm.src_c) + (long)image * prm.H * IW * ROWB : nullptr;
        float c0[PRO ? 8 : 1], c1[PRO ? 8 : 1], c2[PRO == 2 ? 8 : 1];
        if constexpr (PRO != 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                c0[e] = prm.p0[cc * 8 + e];
                c1[e] = prm.p1[cc * 8 + e];
                if constexpr (PRO == 2) c2[e] = prm.p2[cc * 8 + e];
            }
        }
        char* aout_img = PRO == 2 && prm.aout != nullptr ? reinterpret_cast<char*>(prm.aout) + (long)image * prm.H * IW * ROWB : nullptr;
#pragma unroll  // (straight-line: behind a run-time loop hipcc's wait-count pass turns conservative and drains the weight ring
                // with a vmcnt(0) of its own at the top of every filter tap)
      for (int pass = 0; pass < NPASS; ++pass) {
        const int tbase = tid + pass * NCH * NT;
        uint4 v[NCH], vc[PRO == 2 ? NCH : 1];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int pos = (tbase + i * NT) / CPR;
            const int ph = pos / PW, pw = pos - ph * PW;
            const int hh = row0 + ph - 1, ww = pw - 1;
            const bool ok = pos < PP && hh >= 0 && hh < prm.H && ww >= 0 && ww < IW;
            const unsigned off = (unsigned)((ok ? hh * IW + ww : row0 * IW) * ROWB + cc * 16);
            v[i] = (MSFWSI_IMG_ABLATE & 1) ? make_uint4(off, 0, 0, 0) : *reinterpret_cast<const uint4*>(src_img + off);
            if constexpr (PRO == 2) vc[i] = (MSFWSI_IMG_ABLATE & 1) ? make_uint4(off, 0, 0, 0) : *reinterpret_cast<const uint4*>(srcc_img + off);
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int pos = (tbase + i * NT) / CPR;
            const int ph = pos / PW, pw = pos - ph * PW;
            const int hh = row0 + ph - 1, ww = pw - 1;
            const bool ok = pos < PP && hh >= 0 && hh < prm.H && ww >= 0 && ww < IW;
            uint4 t = v[i];
            if constexpr (PRO == 1) {
                float f[8];
                unpack16<T>(t, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaxf(fmaf(f[e], c0[e], c1[e]), 0.f);
                t = pack16<T>(f);
            } else if constexpr (PRO == 2) {
                float g[8], c[8];
                unpack16<T>(t, g);
                unpack16<T>(vc[i], c);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = fmaf(c0[e], g[e], fmaf(c1[e], c[e], c2[e]));  // = msfwsi_bn_bwd_apply
                t = pack16<T>(g);
                // every pixel is written by the ONE band that owns its row (halo rows belong to the neighbours)
                if (aout_img != nullptr && ok && ph >= 1 && ph <= BH)
                    *reinterpret_cast<uint4*>(aout_img + (unsigned)((hh * IW + ww) * ROWB + cc * 16)) = t;
            }
            if (!ok) t = make_uint4(0, 0, 0, 0);  // zero padding AFTER the transform (relu(shift) is not zero)
            if (pos < PP) *reinterpret_cast<uint4*>(img + pos * ROWB + ((cc ^ img_swz<C>(ph * IW + pw)) << 4)) = t;
        }
      }
    }
    // A wait hipcc can see: its wait-count pass otherwise carries "a staging load may still be writing v54 / v64" into the
    // k loop (the staging loads sit in exec-masked branches) and guards the first re-use of those registers with a
    // vmcnt(0) of its own at the top of EVERY filter tap -- which drains the weight ring nine times per workgroup.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); expcnt / lgkmcnt untouched
    // the first R steps' weight fragments of this wave's channel blocks (TN per step), then the only barrier
    const char* wb = reinterpret_cast<const char*>(prm.wpk) + (long)wave * TN * (9 * KC) * 1024;
    constexpr unsigned NBLK = 9 * KC * 1024;  // bytes between the fragment streams of two channel blocks
    u32x4 wf[R][TN];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n) pl_load16<true>(wf[i][n], wb, (unsigned)(n * NBLK + i * 1024 + lane * 16));
    __syncthreads();

    // ---------------- the k loop: 9 taps x KC steps, 7 * TN MFMAs per step ----------------
    int pos0[TM];  // padded position of the window origin of this lane's pixel in row tile tm (0 for the padding rows)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int p = tm * 32 + l31;
        const int h = p / IW, w = p - h * IW;
        pos0[tm] = p < MB ? h * PW + w : 0;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[tm][n][j] = 0.f;

    int step = 0;  // steps consumed so far (the ring slot is step % R: static inside the unrolled tap body)
#pragma unroll 1
    for (int tap = 0; tap < ((MSFWSI_IMG_ABLATE & 2) ? 1 : 9); ++tap) {
        const int r = tap / 3, s_ = tap - r * 3;
        const int shift = r * PW + s_;
        int rowb[TM], sw[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            rowb[tm] = (pos0[tm] + shift) * ROWB;
            sw[tm] = img_swz<C>(tm * 32 + l31 + r * IW + s_) ^ lh;  // chunk (2 kc + lh) ^ swz = (2 kc) ^ (lh ^ swz)
        }
        frag_t xc[TM], xn[TN == 2 ? TM : 1];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) xc[tm] = *reinterpret_cast<const frag_t*>(img + rowb[tm] + ((0 ^ sw[tm]) << 4));
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            // the ring holds R steps of TN fragments, oldest first: all but the (R - 1) * TN youngest have landed
            frag_t wfr[TN];
            if constexpr (TN == 1) {
                pl_wait<true, R - 1>(wf[kc % R][0]);
            } else {
                pl_wait<true, (R - 1) * TN>(wf[kc % R][0]);
                pl_wait<true, (R - 1) * TN>(wf[kc % R][1]);
            }
#pragma unroll
            for (int n = 0; n < TN; ++n) wfr[n] = __builtin_bit_cast(frag_t, wf[kc % R][n]);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                for (int n = 0; n < TN; ++n) mma32<T>(acc[tm][n], wfr[n], xc[tm]);
                if (kc + 1 < KC && !(MSFWSI_IMG_ABLATE & 16)) {
                    const frag_t nx = *reinterpret_cast<const frag_t*>(img + rowb[tm] + (((2 * (kc + 1)) ^ sw[tm]) << 4));
                    if constexpr (TN == 2) xn[tm] = nx; else xc[tm] = nx;
                }
            }
            if constexpr (TN == 2) {
                if (kc + 1 < KC) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) xc[tm] = xn[tm];
                }
            }
            // the step R ahead takes the slots just consumed (past the end: the last step again, never used)
            const int nxt = step + kc + R < 9 * KC ? step + kc + R : 9 * KC - 1;
#pragma unroll
            for (int n = 0; n < TN; ++n)
                if (!(MSFWSI_IMG_ABLATE & 8)) pl_load16<true>(wf[kc % R][n], wb, (unsigned)(n * NBLK + nxt * 1024 + lane * 16));
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                __builtin_amdgcn_sched_group_barrier(0x008, TN, 0);
                if (kc + 1 < KC && !(MSFWSI_IMG_ABLATE & 16)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        step += KC;
    }
    // the ring's trailing re-requests are still in flight: the waits NAME their registers, so that hipcc cannot hand them to
    // the epilogue's address arithmetic before the loads have landed (it did: tools/check_hand_waits.py)
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n) pl_drain<true>(wf[i][n]);

    // ---------------- epilogue: 16-byte row chunks through the wave's scratch; statistics / gate + sums ----------------
    const int q = lane & 3, r4 = lane >> 2;
#if MSFWSI_IMG_ABLATE & 32
    const int ab_row = (lane >> 3) + 8 * (wave & 1);         // 8 rows per access, the odd wave of a pair takes rows 8..15
    const int ab_col = (wave & ~1) * 32 + (lane & 7) * 8;    // 128 contiguous bytes: the pair's 64 channels
#define IMG_GROW(p_) (((p_) - r4 + ab_row) < MB ? ((p_) - r4 + ab_row) : MB - 1)
#define IMG_GCOL(c_) (ab_col)
#else
#define IMG_GROW(p_) (p_)
#define IMG_GCOL(c_) (c_)
#endif
    T* __restrict__ out = reinterpret_cast<T*>(prm.out) + pix0 * KO;
    const T* __restrict__ mask_c = DGRAD ? reinterpret_cast<const T*>(prm.mask_c) : nullptr;
    if (mask_c != nullptr) mask_c += pix0 * KO;
    T* __restrict__ act = DGRAD && prm.act_out != nullptr ? reinterpret_cast<T*>(prm.act_out) + pix0 * KO : nullptr;
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        const int ncol = (wave * TN + n) * 32 + q * 8;
        uint4 mk[DGRAD ? TM * 2 : 1];
        float msc[8], msh[8];
        if constexpr (DGRAD) {
            if (mask_c != nullptr) {
#pragma unroll
                for (int t = 0; t < TM * 2; ++t) {
                    const int p = (t >> 1) * 32 + (t & 1) * 16 + r4;
                    mk[t] = (MSFWSI_IMG_ABLATE & 4) ? make_uint4(p, 1, 2, 3) : *reinterpret_cast<const uint4*>(mask_c + (long)(p < MB ? IMG_GROW(p) : 0) * KO + IMG_GCOL(ncol));
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    msc[e] = prm.mask_scale[ncol + e];
                    msh[e] = prm.mask_shift[ncol + e];
                }
            }
        }
        float s0[8], s1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<uint2*>(scratch + l31 * SCR_PITCH + (8 * g + 4 * lh) * 2) =
                    pack4<T>(acc[tm][n][4 * g], acc[tm][n][4 * g + 1], acc[tm][n][4 * g + 2], acc[tm][n][4 * g + 3]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p = tm * 32 + i * 16 + r4;
                uint4 cv = *reinterpret_cast<const uint4*>(scratch + (i * 16 + r4) * SCR_PITCH + q * 16);
                if (p < MB) {
                    float f[8];
                    unpack16<T>(cv, f);
                    if constexpr (DGRAD) {
                        if (mask_c != nullptr) {
                            float c[8], a[8];
                            unpack16<T>(mk[tm * 2 + i], c);
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                a[e] = fmaxf(fmaf(c[e], msc[e], msh[e]), 0.f);
                                if (!(a[e] > 0.f)) f[e] = 0.f;
                                s0[e] += f[e];
                                s1[e] = fmaf(f[e], c[e], s1[e]);
                            }
                            cv = pack16<T>(f);
                            // the gating activation itself, for the weight gradient of this conv (its operand): the
                            // stand-alone msfwsi_bn_act pass would re-read c for it
                            if (act != nullptr) *reinterpret_cast<uint4*>(act + (long)IMG_GROW(p) * KO + IMG_GCOL(ncol)) = pack16<T>(a);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            s0[e] += f[e];
                            s1[e] = fmaf(f[e], f[e], s1[e]);
                        }
                    }
                    if (!(MSFWSI_IMG_ABLATE & 4) || cv.x == 0x12345u) *reinterpret_cast<uint4*>(out + (long)IMG_GROW(p) * KO + IMG_GCOL(ncol)) = cv;
                }
            }
        }
        if (prm.stats != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int off = 4; off < 64; off <<= 1) {
                    s0[e] += __shfl_xor(s0[e], off, 64);
                    s1[e] += __shfl_xor(s1[e], off, 64);
                }
            }
            if (lane < 4) {
                double* dst = prm.stats + (long)(blockIdx.x % prm.nshard) * 2 * KO + ncol;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    atomicAdd(dst + e, (double)s0[e]);
                    atomicAdd(dst + KO + e, (double)s1[e]);
                }
            }
        }
    }
}
#undef IMG_GROW
#undef IMG_GCOL

// wpk[n/32][step][lane][j], step = tap * (Ck/16) + c/16:  forward  W'(n, tap, c) = w[n][tap][c]         (w = [K][3][3][C])
//                                                          gradient W'(n, tap, c) = w[c][8 - tap][n]     (taps flipped)
template <typename U>
__global__ void img3x3_pack_kernel(const U* __restrict__ w, U* __restrict__ wpk, int K, int C, int dgrad) {
    const int nout = dgrad ? C : K, ck = dgrad ? K : C;
    const long total = (long)nout * 9 * ck;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long frag = i >> 9;
    const int steps = 9 * (ck >> 4);
    const int step = (int)(frag % steps), nb = (int)(frag / steps);
    const int tap = step / (ck >> 4), c = (step - tap * (ck >> 4)) * 16 + 8 * (lane >> 5) + j;
    const int n = nb * 32 + (lane & 31);
    wpk[i] = dgrad ? w[((long)c * 9 + (8 - tap)) * C + n] : w[((long)n * 9 + tap) * C + c];
}

// ---------------------------------------------------------------------------------------------------------------------
// Input gradient of the STRIDED conv2 (3x3 / stride 2 / pad 1, C -> C) of layer2.0 / layer3.0 (resnet.py:128 with
// stride 2): dx [N][2 IH][2 IW][C] from dy [N][IH][IW][C].  The gather kernel runs it as four launches (one per parity
// class of the output position), each reading dy and its part of the mask again and writing a quarter of the pixels.  Here a
// workgroup stages a band of dy ONCE (BH rows + one halo row below, IW + 1 columns: 196 gradient pixels = seven MFMA tiles)
// and makes four passes over it, one per parity (a, b) of the output position (2h + a, 2w + b):
//     dx(2h+a, 2w+b) = sum over r in R(a), s in R(b) of  dy(h + dh(a, r), w + dh(b, s)) . w[.][r][s][.]
//     R(0) = {1}, dh = 0;    R(1) = {0, 2}, dh(1, 0) = 1, dh(1, 2) = 0                      (1 + 2 + 2 + 4 = 9 taps in all)
// -- the same seven accumulator tiles, weight ring and address-shift taps as img3x3_kernel; each pass ends with the epilogue
// of its 196 output pixels (ReLU gate from the raw c1 at the output resolution, BatchNorm sums, optionally a1 = relu(bn1(c1))
// for the weight gradient, as in msfwsi_img3x3_dgrad).  PRO 2: dc = k1 dy + k2 c2 + k3 in the staging, written back.
// The filter is packed in pass order (msfwsi_img3x3_pack_weights, mode 2).
struct Img3S2Params {
    const void* src;    // [N][IH][IW][C]: PRO 0 the gradient operand, PRO 2 the gated gradient g
    const void* src_c;  // PRO 2: conv2's raw output c2
    const float* p0;    // PRO 2: k1, k2, k3
    const float* p1;
    const float* p2;
    void* aout;         // PRO 2, nullable: dc [N][IH][IW][C]
    const void* wpk;    // [C/32][9 * C/16][64][8], steps in pass order
    void* out;          // dx [N][2 IH][2 IW][C]
    double* stats;      // {sum g, sum g*c} of the gated result, [nshard][2][C]
    int nshard;
    const void* mask_c;  // [N][2 IH][2 IW][C] raw conv output whose BatchNorm+ReLU gates dx, nullable
    const float* mask_scale;
    const float* mask_shift;
    void* act_out;       // nullable (needs mask_c): relu(mask_scale * mask_c + mask_shift)
    int N, IH;
};

template <typename T, int C, int BH, int IW, int PRO>
__global__ __launch_bounds__(C * 2, 2) void img3x3_s2d_kernel(const Img3S2Params prm) {
    constexpr int NW = C / 32, NT = 64 * NW;
    constexpr int PW = IW + 1, PP = (BH + 1) * PW;  // staged positions: one halo row below, one halo column right
    constexpr int CPR = C / 8, ROWB = C * 2;
    constexpr int NCHT = (PP * CPR + NT - 1) / NT;
    constexpr int NPASS = (NCHT + 15) / 16;
    constexpr int NCH = (NCHT + NPASS - 1) / NPASS;
    constexpr int MB = BH * IW;
    constexpr int TM = (MB + 31) / 32;
    constexpr int KC = C / 16;
    constexpr int R = 4;
    constexpr int KU = KC > 8 ? 8 : KC;
    constexpr int OW = 2 * IW;
    constexpr int SCR_PITCH = 80, SCR_BYTES = 32 * SCR_PITCH + 256;  // + this wave's 2 x 32 fp32 statistics
    static_assert(C % 128 == 0 && IW % 2 == 0 && NT % CPR == 0 && KU % R == 0 && KC % KU == 0, "strided image kernel geometry");
    typedef typename MmaFrag<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* img = smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    char* scratch = smem + PP * ROWB + wave * SCR_BYTES;

    unsigned bid = blockIdx.x;  // bands in memory order per XCD, see img3x3_kernel
    if ((gridDim.x & 7u) == 0) bid = (bid & 7u) * (gridDim.x >> 3) + (bid >> 3);
    const int nband = prm.IH / BH;
    const int image = bid / nband, band = bid - image * nband;
    const int row0 = band * BH;

    // ---------------- stage the band of the gradient (+ halo row / column, zero outside the image) ----------------
    {
        const int cc = tid % CPR;
        const char* src_img = reinterpret_cast<const char*>(prm.src) + (long)image * prm.IH * IW * ROWB;
        const char* srcc_img = PRO == 2 ? reinterpret_cast<const char*>(prm.src_c) + (long)image * prm.IH * IW * ROWB : nullptr;
        float c0[PRO ? 8 : 1], c1[PRO ? 8 : 1], c2[PRO ? 8 : 1];
        if constexpr (PRO == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                c0[e] = prm.p0[cc * 8 + e];
                c1[e] = prm.p1[cc * 8 + e];
                c2[e] = prm.p2[cc * 8 + e];
            }
        }
        char* aout_img = PRO == 2 && prm.aout != nullptr ? reinterpret_cast<char*>(prm.aout) + (long)image * prm.IH * IW * ROWB : nullptr;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int tbase = tid + pass * NCH * NT;
            uint4 v[NCH], vc[PRO == 2 ? NCH : 1];
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int pos = (tbase + i * NT) / CPR;
                const int ph = pos / PW, pw = pos - ph * PW;
                const int hh = row0 + ph;
                const bool ok = pos < PP && hh < prm.IH && pw < IW;
                const unsigned off = (unsigned)((ok ? hh * IW + pw : row0 * IW) * ROWB + cc * 16);
                v[i] = *reinterpret_cast<const uint4*>(src_img + off);
                if constexpr (PRO == 2) vc[i] = *reinterpret_cast<const uint4*>(srcc_img + off);
            }
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int pos = (tbase + i * NT) / CPR;
                const int ph = pos / PW, pw = pos - ph * PW;
                const int hh = row0 + ph;
                const bool ok = pos < PP && hh < prm.IH && pw < IW;
                uint4 t = v[i];
                if constexpr (PRO == 2) {
                    float g[8], c[8];
                    unpack16<T>(t, g);
                    unpack16<T>(vc[i], c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] = fmaf(c0[e], g[e], fmaf(c1[e], c[e], c2[e]));  // = msfwsi_bn_bwd_apply
                    t = pack16<T>(g);
                    if (aout_img != nullptr && ok && ph < BH)  // the band that owns the row writes it
                        *reinterpret_cast<uint4*>(aout_img + (unsigned)((hh * IW + pw) * ROWB + cc * 16)) = t;
                }
                if (!ok) t = make_uint4(0, 0, 0, 0);
                if (pos < PP) *reinterpret_cast<uint4*>(img + pos * ROWB + ((cc ^ img_swz<C>(ph * IW + pw)) << 4)) = t;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), visible to hipcc's wait-count pass (see img3x3_kernel)
    __syncthreads();

    const char* wb = reinterpret_cast<const char*>(prm.wpk) + (long)wave * (9 * KC) * 1024;
    const int q = lane & 3, r4 = lane >> 2;
    const int ncol = wave * 32 + q * 8;
    const long opix0 = ((long)image * 2 * prm.IH + 2 * row0) * OW;  // first output pixel of the band's first output row
    T* __restrict__ out = reinterpret_cast<T*>(prm.out) + opix0 * C;
    const T* __restrict__ mask_c = reinterpret_cast<const T*>(prm.mask_c);
    if (mask_c != nullptr) mask_c += opix0 * C;
    T* __restrict__ act = prm.act_out != nullptr ? reinterpret_cast<T*>(prm.act_out) + opix0 * C : nullptr;
    // the passes' statistics meet in LDS (registers are the kernel's scarce resource: 7 accumulator tiles + ring + addresses)
    float* wstat = reinterpret_cast<float*>(scratch + 32 * SCR_PITCH);
    wstat[lane] = 0.f;

    int sbase = 0;  // steps of the passes before this one
#pragma unroll 1
    for (int par = 0; par < 4; ++par) {
        const int a = par >> 1, b = par & 1;
        const int ntap = (1 + a) * (1 + b);
        const int send = sbase + ntap * KC;
        u32x4 wf[R];
#pragma unroll
        for (int i = 0; i < R; ++i) pl_load16<true>(wf[i], wb, (unsigned)((sbase + i) * 1024 + lane * 16));
        f32x16 acc[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[tm][j] = 0.f;
        int step = sbase;
#pragma unroll 1
        for (int t = 0; t < ntap; ++t) {
            // taps of the pass in (r, s) order: rows r in R(a), columns s in R(b); the FIRST element of R(1) = {0, 2} reads one
            // row (column) further on
            const int tr = b ? (t >> 1) : t, ts = b ? (t & 1) : 0;
            const int dh = (a && tr == 0) ? 1 : 0, dw = (b && ts == 0) ? 1 : 0;
            const int shift = dh * PW + dw;
            int rowb[TM], sw[TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int p = tm * 32 + l31;  // (recomputed per tap: three VALU against 16 MFMAs; registers are scarce)
                const int h = p / IW, w = p - h * IW;
                rowb[tm] = ((p < MB ? h * PW + w : 0) + shift) * ROWB;
                sw[tm] = img_swz<C>(p + dh * IW + dw) ^ lh;
            }
            frag_t xc[TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) xc[tm] = *reinterpret_cast<const frag_t*>(img + rowb[tm] + ((0 ^ sw[tm]) << 4));
            // KU steps unrolled; at 256 channels two rounds of them (16 unrolled steps + the epilogue's state spill)
#pragma unroll 1
            for (int k0 = 0; k0 < KC; k0 += KU) {
#pragma unroll
                for (int u = 0; u < KU; ++u) {
                    const int kc = k0 + u;
                    pl_wait<true, R - 1>(wf[u % R]);
                    const frag_t wfr = __builtin_bit_cast(frag_t, wf[u % R]);
                    const int kn = kc + 1 < KC ? kc + 1 : KC - 1;  // (the tap's last step re-reads its own chunk: never used)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        mma32<T>(acc[tm], wfr, xc[tm]);
                        xc[tm] = *reinterpret_cast<const frag_t*>(img + rowb[tm] + (((2 * kn) ^ sw[tm]) << 4));
                    }
                    // the step R ahead takes the slot just consumed (past the pass's end: its last step again, never used)
                    const int nxt = step + kc + R < send ? step + kc + R : send - 1;
                    pl_load16<true>(wf[u % R], wb, (unsigned)(nxt * 1024 + lane * 16));
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            step += KC;
        }
#pragma unroll
        for (int i = 0; i < R; ++i) pl_drain<true>(wf[i]);  // the trailing re-requests land before anything re-uses their registers
        sbase = send;
        // an opaque zero: the epilogue's 14 output addresses depend only on the pass, and hipcc would otherwise compute them
        // BEFORE the k loop and carry them through it (-> 80 spilled registers)
        int zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
        const int r4z = r4 + zero;

        // ---- epilogue of the pass: output pixel (2h + a, 2w + b) of gradient pixel (h, w); mask operands in two halves ----
        float msc[8], msh[8], s0[8], s1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
        if (mask_c != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                msc[e] = prm.mask_scale[ncol + e];
                msh[e] = prm.mask_shift[ncol + e];
            }
        }
        constexpr int TH = (TM + 1) / 2;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int tm_lo = half * TH, tm_hi = half ? TM : TH;
            uint4 mk[TH * 2];
            if (mask_c != nullptr) {
#pragma unroll
                for (int t2 = 0; t2 < (tm_hi - tm_lo) * 2; ++t2) {
                    const int p = (tm_lo + (t2 >> 1)) * 32 + (t2 & 1) * 16 + r4z;
                    const int h = p / IW, w = p - h * IW;
                    const int op = p < MB ? (2 * h + a) * OW + 2 * w + b : 0;
                    mk[t2] = *reinterpret_cast<const uint4*>(mask_c + (long)op * C + ncol);
                }
            }
#pragma unroll
            for (int tm = tm_lo; tm < tm_hi; ++tm) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<uint2*>(scratch + l31 * SCR_PITCH + (8 * g + 4 * lh) * 2) =
                        pack4<T>(acc[tm][4 * g], acc[tm][4 * g + 1], acc[tm][4 * g + 2], acc[tm][4 * g + 3]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int p = tm * 32 + i * 16 + r4z;
                    uint4 cv = *reinterpret_cast<const uint4*>(scratch + (i * 16 + r4) * SCR_PITCH + q * 16);
                    if (p < MB) {
                        const int h = p / IW, w = p - h * IW;
                        const long op = (long)((2 * h + a) * OW + 2 * w + b) * C + ncol;
                        if (mask_c != nullptr) {
                            float f[8], c[8], av[8];
                            unpack16<T>(cv, f);
                            unpack16<T>(mk[(tm - tm_lo) * 2 + i], c);
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                av[e] = fmaxf(fmaf(c[e], msc[e], msh[e]), 0.f);
                                if (!(av[e] > 0.f)) f[e] = 0.f;
                                s0[e] += f[e];
                                s1[e] = fmaf(f[e], c[e], s1[e]);
                            }
                            cv = pack16<T>(f);
                            if (act != nullptr) *reinterpret_cast<uint4*>(act + op) = pack16<T>(av);
                        }
                        *reinterpret_cast<uint4*>(out + op) = cv;
                    }
                }
            }
        }
        if (prm.stats != nullptr) {  // this pass's sums: lanes with the same chunk q meet, lanes 0-3 add them to the wave's LDS row
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int off = 4; off < 64; off <<= 1) {
                    s0[e] += __shfl_xor(s0[e], off, 64);
                    s1[e] += __shfl_xor(s1[e], off, 64);
                }
            }
            if (lane < 4) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    wstat[q * 8 + e] += s0[e];
                    wstat[32 + q * 8 + e] += s1[e];
                }
            }
        }
    }
    if (prm.stats != nullptr && lane < 32) {  // (wave-private LDS row: program order suffices)
        double* dst = prm.stats + (long)(blockIdx.x % prm.nshard) * 2 * C + wave * 32 + lane;
        atomicAdd(dst, (double)wstat[lane]);
        atomicAdd(dst + C, (double)wstat[32 + lane]);
    }
}

// pass order of the strided gradient: step = sum of the earlier passes' taps * (C/16) + tap * (C/16) + k/16;
// W'(n, pass (a, b), tap (r, s), k) = w[k][r][s][n],  r in R(a), s in R(b), R(0) = {1}, R(1) = {0, 2}
template <typename U>
__global__ void img3x3_pack_s2d_kernel(const U* __restrict__ w, U* __restrict__ wpk, int C) {
    const long total = (long)C * 9 * C;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long frag = i >> 9;
    const int kcs = C >> 4, steps = 9 * kcs;
    const int step = (int)(frag % steps), nb = (int)(frag / steps);
    int tap = step / kcs;  // 0 .. 8 in pass order: 1 + 2 + 2 + 4
    const int k = (step - tap * kcs) * 16 + 8 * (lane >> 5) + j;
    int a, b;
    if (tap < 1) { a = 0; b = 0; }
    else if (tap < 3) { a = 0; b = 1; tap -= 1; }
    else if (tap < 5) { a = 1; b = 0; tap -= 3; }
    else { a = 1; b = 1; tap -= 5; }
    const int tr = b ? (tap >> 1) : tap, ts = b ? (tap & 1) : 0;
    const int r = a ? 2 * tr : 1, s_ = b ? 2 * ts : 1;
    const int n = nb * 32 + (lane & 31);
    wpk[i] = w[(((long)k * 3 + r) * 3 + s_) * C + n];
}

template <typename T, int C, int BH, int IW, int PRO>
int launch_img_s2d(const Img3S2Params& prm, hipStream_t stream) {
    constexpr int LDS = (BH + 1) * (IW + 1) * C * 2 + (C / 32) * (32 * 80 + 256);
    void (*kern)(const Img3S2Params) = img3x3_s2d_kernel<T, C, BH, IW, PRO>;
    if (LDS > 64 * 1024) {
        if (int e = msfwsi_raise_lds(reinterpret_cast<const void*>(kern), LDS)) return e;
    }
    const long nwg = (long)prm.N * (prm.IH / BH);
    if (nwg <= 0 || nwg > 0x7fffffffL) return MSFWSI_EINVAL;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(C * 2), LDS, stream, prm);
    return msfwsi_launch_status();
}

template <typename T, int PRO>
int dispatch_img_s2d(const msfwsi_conv_desc* d, const Img3S2Params& prm, hipStream_t st) {
    if (d->P == 28 && d->C == 128) return launch_img_s2d<T, 128, 7, 28, PRO>(prm, st);
    if (d->P == 14 && d->C == 256) return launch_img_s2d<T, 256, 14, 14, PRO>(prm, st);
    return MSFWSI_EUNSUPPORTED;
}

template <typename T, int C, int BH, int IW, int PRO, bool DGRAD>
int launch_img(const Img3Params& prm, hipStream_t stream) {
    constexpr int KO = C;
    constexpr int TN = MSFWSI_IMG_TN;
    constexpr int LDS = (BH + 2) * (IW + 2) * C * 2 + (KO / (32 * TN)) * 32 * 80;
    void (*kern)(const Img3Params) = img3x3_kernel<T, C, KO, BH, IW, PRO, DGRAD, TN>;
    if (LDS > 64 * 1024) {
        if (int e = msfwsi_raise_lds(reinterpret_cast<const void*>(kern), LDS)) return e;
    }
    const long nwg = (long)prm.N * (prm.H / BH);
    if (nwg <= 0 || nwg > 0x7fffffffL) return MSFWSI_EINVAL;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(KO * 2 / TN), LDS, stream, prm);
    return msfwsi_launch_status();
}

template <typename T, int PRO, bool DGRAD>
int dispatch_img(const msfwsi_conv_desc* d, const Img3Params& prm, hipStream_t st) {
    if (d->H == 14 && d->C == 256) return launch_img<T, 256, 14, 14, PRO, DGRAD>(prm, st);
    if (d->H == 28 && d->C == 128) return launch_img<T, 128, 7, 28, PRO, DGRAD>(prm, st);
    if (d->H == 56 && d->C == 64) return launch_img<T, 64, 4, 56, PRO, DGRAD>(prm, st);
    return MSFWSI_EUNSUPPORTED;
}

}  // namespace

extern "C" int msfwsi_img3x3_supported(const msfwsi_conv_desc* d) {
    if (d == nullptr || (d->dtype != MSFWSI_DT_BF16 && d->dtype != MSFWSI_DT_F16)) return 0;
    if (d->R != 3 || d->S != 3 || d->stride != 1 || d->pad != 1 || d->C != d->K || d->H != d->W || d->P != d->H || d->Q != d->W) return 0;
    return ((d->H == 14 && d->C == 256) || (d->H == 28 && d->C == 128) || (d->H == 56 && d->C == 64)) &&
                   (long)d->N * d->H * d->W <= 0x7fffffffL ? 1 : 0;
}

extern "C" int msfwsi_img3x3_pack_weights(int dtype, const void* w, void* wpk, int K, int C, int dgrad, void* stream) {
    MSFWSI_CHECK_ARG(w != nullptr && wpk != nullptr && K > 0 && C > 0 && dgrad >= 0 && dgrad <= 2);
    if ((dtype != MSFWSI_DT_BF16 && dtype != MSFWSI_DT_F16) || K % 32 != 0 || C % 32 != 0) return MSFWSI_EUNSUPPORTED;
    const long n = (long)K * 9 * C;
    if (dgrad == 2) {  // the strided gradient's pass order
        if (K != C) return MSFWSI_EUNSUPPORTED;
        hipLaunchKernelGGL(img3x3_pack_s2d_kernel<unsigned short>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const unsigned short*>(w),
                           reinterpret_cast<unsigned short*>(wpk), C);
        return msfwsi_launch_status();
    }
    hipLaunchKernelGGL(img3x3_pack_kernel<unsigned short>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const unsigned short*>(w),
                       reinterpret_cast<unsigned short*>(wpk), K, C, dgrad ? 1 : 0);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_img3x3_fwd(const msfwsi_conv_desc* d, const void* x, const float* pro_scale, const float* pro_shift,
                                 const void* wpk, void* y, double* stats, int nshard, void* stream) {
    if (!msfwsi_img3x3_supported(d)) return d == nullptr ? MSFWSI_EINVAL : MSFWSI_EUNSUPPORTED;
    MSFWSI_CHECK_ARG(x != nullptr && wpk != nullptr && y != nullptr && (stats == nullptr || nshard >= 1));
    MSFWSI_CHECK_ARG((pro_scale == nullptr) == (pro_shift == nullptr));
    Img3Params prm{};
    prm.src = x; prm.p0 = pro_scale; prm.p1 = pro_shift; prm.wpk = wpk; prm.out = y;
    prm.stats = stats; prm.nshard = nshard > 0 ? nshard : 1;
    prm.N = d->N; prm.H = d->H;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool pro = pro_scale != nullptr;
    if (d->dtype == MSFWSI_DT_BF16) return pro ? dispatch_img<__bf16, 1, false>(d, prm, st) : dispatch_img<__bf16, 0, false>(d, prm, st);
    return pro ? dispatch_img<_Float16, 1, false>(d, prm, st) : dispatch_img<_Float16, 0, false>(d, prm, st);
}

extern "C" int msfwsi_img3x3_dgrad(const msfwsi_conv_desc* d, const void* dy, const void* c, const float* k1, const float* k2,
                                   const float* k3, void* dc_out, const void* wpk, void* dx, const void* mask_c,
                                   const float* mask_scale, const float* mask_shift, void* act_out, double* sums, int nshard,
                                   void* stream) {
    if (!msfwsi_img3x3_supported(d)) return d == nullptr ? MSFWSI_EINVAL : MSFWSI_EUNSUPPORTED;
    MSFWSI_CHECK_ARG(dy != nullptr && wpk != nullptr && dx != nullptr);
    const bool pro = c != nullptr;
    MSFWSI_CHECK_ARG(pro == (k1 != nullptr) && pro == (k2 != nullptr) && pro == (k3 != nullptr) && (pro || dc_out == nullptr));
    MSFWSI_CHECK_ARG((mask_c == nullptr) == (mask_scale == nullptr) && (mask_c == nullptr) == (mask_shift == nullptr));
    MSFWSI_CHECK_ARG((mask_c == nullptr) == (sums == nullptr) && (sums == nullptr || nshard >= 1));
    MSFWSI_CHECK_ARG(act_out == nullptr || (mask_c != nullptr && act_out != mask_c && act_out != dx));
    // a band reads its halo rows from the gradient of the NEIGHBOURING bands: written back in place, a neighbour's rows could
    // already hold dc instead of g.  In place only where a workgroup owns the whole image (14 x 14).
    MSFWSI_CHECK_ARG(dc_out == nullptr || dc_out != dy || d->H == 14);
    Img3Params prm{};
    prm.src = dy; prm.src_c = c; prm.p0 = k1; prm.p1 = k2; prm.p2 = k3; prm.aout = dc_out;
    prm.wpk = wpk; prm.out = dx;
    prm.mask_c = mask_c; prm.mask_scale = mask_scale; prm.mask_shift = mask_shift; prm.act_out = act_out;
    prm.stats = sums; prm.nshard = nshard > 0 ? nshard : 1;
    prm.N = d->N; prm.H = d->H;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == MSFWSI_DT_BF16) return pro ? dispatch_img<__bf16, 2, true>(d, prm, st) : dispatch_img<__bf16, 0, true>(d, prm, st);
    return pro ? dispatch_img<_Float16, 2, true>(d, prm, st) : dispatch_img<_Float16, 0, true>(d, prm, st);
}

extern "C" int msfwsi_img3x3_s2_dgrad_supported(const msfwsi_conv_desc* d) {
    if (d == nullptr || (d->dtype != MSFWSI_DT_BF16 && d->dtype != MSFWSI_DT_F16)) return 0;
    if (d->R != 3 || d->S != 3 || d->stride != 2 || d->pad != 1 || d->C != d->K || d->H != d->W || d->P != d->Q) return 0;
    if (d->H != 2 * d->P) return 0;
    return ((d->P == 28 && d->C == 128) || (d->P == 14 && d->C == 256)) && (long)d->N * d->H * d->W <= 0x7fffffffL ? 1 : 0;
}

extern "C" int msfwsi_img3x3_s2_dgrad(const msfwsi_conv_desc* d, const void* dy, const void* c, const float* k1, const float* k2,
                                      const float* k3, void* dc_out, const void* wpk, void* dx, const void* mask_c,
                                      const float* mask_scale, const float* mask_shift, void* act_out, double* sums, int nshard,
                                      void* stream) {
    if (!msfwsi_img3x3_s2_dgrad_supported(d)) return d == nullptr ? MSFWSI_EINVAL : MSFWSI_EUNSUPPORTED;
    MSFWSI_CHECK_ARG(dy != nullptr && wpk != nullptr && dx != nullptr);
    const bool pro = c != nullptr;
    MSFWSI_CHECK_ARG(pro == (k1 != nullptr) && pro == (k2 != nullptr) && pro == (k3 != nullptr) && (pro || dc_out == nullptr));
    MSFWSI_CHECK_ARG((mask_c == nullptr) == (mask_scale == nullptr) && (mask_c == nullptr) == (mask_shift == nullptr));
    MSFWSI_CHECK_ARG((mask_c == nullptr) == (sums == nullptr) && (sums == nullptr || nshard >= 1));
    MSFWSI_CHECK_ARG(act_out == nullptr || (mask_c != nullptr && act_out != mask_c && act_out != dx));
    // bands read their halo row from the gradient of the band below: in place only where a workgroup owns the whole image
    MSFWSI_CHECK_ARG(dc_out == nullptr || dc_out != dy || d->P == 14);
    Img3S2Params prm{};
    prm.src = dy; prm.src_c = c; prm.p0 = k1; prm.p1 = k2; prm.p2 = k3; prm.aout = dc_out;
    prm.wpk = wpk; prm.out = dx;
    prm.mask_c = mask_c; prm.mask_scale = mask_scale; prm.mask_shift = mask_shift; prm.act_out = act_out;
    prm.stats = sums; prm.nshard = nshard > 0 ? nshard : 1;
    prm.N = d->N; prm.IH = d->P;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == MSFWSI_DT_BF16) return pro ? dispatch_img_s2d<__bf16, 2>(d, prm, st) : dispatch_img_s2d<__bf16, 0>(d, prm, st);
    return pro ? dispatch_img_s2d<_Float16, 2>(d, prm, st) : dispatch_img_s2d<_Float16, 0>(d, prm, st);
}
