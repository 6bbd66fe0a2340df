// Gather-GEMM ("implicit GEMM") for the MSF-WSI encoders and heads on gfx950.
//
//   out[m][n] = sum_k A(m,k) * B(k,n)        m = (image, p, q) output pixel, n = output channel
//
// FWD   : A(m,(r,s,c)) = act(src[img, p*stride-pad+r, q*stride-pad+s, c]),  B = W[n][r][s][c]
//         replaces nn.Conv2d (bias-free 3x3/1x1/7x7, reference src/models/resnet.py:25-33,174) and
//         nn.Linear (src/models/backbone.py:12-31, a 1x1 conv on an H=W=1 tensor); `act` is the
//         producer's BatchNorm+ReLU (resnet.py:69-71) applied while the tile is staged, so the
//         normalised activation never round-trips through HBM.  The epilogue emits the raw conv
//         output plus per-channel sum / sum-of-squares for the consumer BatchNorm (training mode).
// DGRAD : A(m,(r,s,co)) = dY[img, (p+pad-r)/stride, (q+pad-s)/stride, co],   B = W[co][r][s][n]
//         (the forward weight tensor read as a [k][n] operand -> no transposed weight copy).
//
// Activations are NHWC, weights are [Cout][R][S][Cin] (torch channels_last physical layout).
// Tiles: BM x BN outputs per workgroup (128x64 / 128x128 with 4 waves, 256x128 with 8), 64-byte k-slab per stage
// (32 16-bit / 16 fp32 values), three LDS stages (two slabs in flight behind a counted s_waitcnt vmcnt + s_barrier).
// Two kernels share the MFMA loop and the epilogue:
//   * igemm_dma_kernel (97 % of the time): every operand byte arrives by `buffer_load_dwordx4 ... lds`.  Operand
//     rows are 64 bytes = four 16-byte chunks stored UNPADDED with the chunk index XOR-swizzled by (row>>2)&3 (the
//     swizzle sits on the per-lane SOURCE offset), which makes every ds_read_b128 fragment read conflict-free.  The
//     per-lane byte offset of each piece is constant over the k loop; tap and channel slab move a wave-uniform scalar
//     offset; out-of-image taps / rows use an out-of-range offset (zeros).  Template switches keep instances lean:
//     EPI (epilogue feature class), TWO (second source tensor with its own k range), RUN (stem: filter-row runs).
//   * igemm_kernel (generic): `global_load_lds` with per-lane 64-bit addresses and a zero page, runtime tap
//     arithmetic, optional register-staged BatchNorm+ReLU prologue (APRO) -- stem-like channel counts, small heads.
// MFMA 32x32x16 (bf16 / f16) or 32x32x2 (exact fp32).  The weight tile is the MFMA A operand and the activation
// tile the B operand, so a lane's 16 accumulator registers hold 4x4 consecutive output channels of ONE pixel; the
// tile is then transposed through LDS and leaves as 16-byte row chunks (igemm_epilogue).
#include "common.h"
#include "../../include/msfwsi_hip.h"

#ifndef MSFWSI_ABLATE
#define MSFWSI_ABLATE 0  // diagnostic builds (tools/build_variant.sh): 1 = pure-DMA kernel without its k loop (epilogue only),
#endif                   // 2 = without the epilogue's global loads / stores (k loop + LDS transposition only), 3 = 3x3 launches
                         // without the activation DMA pieces of the filter taps s = 1, 2.  WRONG RESULTS.
#ifndef MSFWSI_IGEMM_PIPE
#define MSFWSI_IGEMM_PIPE 1  // igemm_dma_kernel: fragment reads pipelined across the slab barrier (0: the round-2..5 loop)
#endif
#ifndef MSFWSI_FETCH_FIRST
#define MSFWSI_FETCH_FIRST 1  // DMA requests of slab kt+2 before the MFMAs of slab kt (0: after them; A/B: make EXTRA=-DMSFWSI_FETCH_FIRST=0)
#endif

namespace {

__device__ __attribute__((aligned(256))) unsigned int g_zero_page[64];  // 256 bytes of zeros (never written)

struct IgemmParams {
    const void* src;
    const void* wgt;
    void* out;
    const float* pro_scale;  // per source channel, nullable
    const float* pro_shift;
    const float* bias;       // per output channel, nullable
    double* stats;           // [nshard][2][Nout], nullable
    const void* resid;       // [M][Nout], nullable (added in the epilogue)
    const void* gapg;        // [Nimg][Nout] storage type, nullable (added, times gap_scale)
    float gap_scale;
    const void* mask_c;      // [M][Nout] raw conv output whose BatchNorm+ReLU gates this gradient, nullable
    const float* mask_scale; // with mask_c: out = acc * (mask_scale*c + mask_shift > 0), stats = {sum g, sum g*c}
    const float* mask_shift;
    const float* post_scale; // per output channel, nullable: out = [relu](round(acc)*post_scale + post_shift + resid)
    const float* post_shift;
    int post_relu;
    int resid_stride;        // EPI 3: resid is [N][P/s][Q/s][Nout] and is added only at pixels with h % s == w % s == 0
    FastDiv div_pq, div_q;   //   (the zero-stuffed gradient of a strided downsample branch, never materialised)
    int pix_stride;          // RUN kernels: elements between consecutive source pixels (the k range of one filter row
    int s_run;               //   is a run of s_run pixels x pix_stride channels, padded to C = a multiple of the slab)
    const void* src2;        // DGRAD 1x1 only, nullable: second source [M][C2] whose k-range follows the first
    int C2;                  //   (out = src . W[0:C] + src2 . W[C:C+C2]; Ktot = C + C2)
    const float* pro2_scale; // nullable, [C2]: the second source is a RAW conv output c and the operand is relu(pro2_scale * c +
    const float* pro2_shift; //   pro2_shift) -- formed on the fragments read from LDS (msfwsi_conv_dgrad2_pro), never stored
    unsigned char* gate_out;        // gate bytes of [M][Nout/VEC] chunks (layout: gate_off, common.h), nullable: bit e = (out[m][VEC*chunk+e] > 0)
    const unsigned char* mask_bits; // same layout, nullable: gates this gradient instead of mask_c (stats = {sum g, 0})
    int N, H, W, C;          // source tensor
    int P, Q, Nout;          // output tensor
    int R, S, stride, pad;
    int M, Ktot;
    int nshard;
    int ntile_n;
    // Stride-2 3x3 input gradient by OUTPUT-PIXEL PARITY (pure-DMA kernel, DGRAD): launch (par_a, par_b) computes the dX
    // pixels (2i + par_a, 2j + par_b).  For them only the taps r = wr0 + 2 r', s = ws0 + 2 s' (r' < R, s' < S) meet a
    // dY pixel, so the launch is a stride-1 gradient of an R x S (1x1, 1x2, 2x1, 2x2) kernel over the [N][P][Q] grid of
    // dY with paddings (pad, pad_w) -- 9 tap-passes over M/4 rows each instead of 9 over M (3/4 of them on zeros) --
    // whose weights are read in place from the 3x3 tensor (w_S = 3 taps per row, w_RS = 9 per channel) and whose
    // rows are written to / read from the interleaved full-resolution pixels (epilogue: pix_of_row).
    int par_mode, par_a, par_b, pad_w, wr0, ws0, w_S, w_RS;
    FastDiv par_div_pq, par_div_q;
};

template <typename T>
using Frag = MmaFrag<T>;

template <typename T>
__device__ __forceinline__ void mma_step(f32x16& acc, const typename MmaFrag<T>::type& w,
                                         const typename MmaFrag<T>::type& x) {
    mma32<T>(acc, w, x);
}

// 16 bytes global -> LDS without touching VGPRs; lds_wave_base is wave-uniform, lane l lands at base + 16*l
__device__ __forceinline__ void dma16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T, int BM, int BN, int WM, int WN, bool DGRAD, bool APRO>
struct IgemmCfg {
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int BK = ElemTraits<T>::BK;  // 64-byte operand rows
    // natural [k][n] weight tile (DGRAD): unpadded rows of ROWB bytes, 64-byte blocks XOR-swizzled by the
    // row index so that the four rows of a transposed read hit the four quarters of the 256-byte bank row
    static constexpr int ROWB = BN * (int)sizeof(T);
    static constexpr int NAT_RPI = 1024 / ROWB;             // rows covered by one 1-KiB DMA instruction
    static constexpr int NW = WM * WN;                      // wavefronts per workgroup (4 or 8)
    static constexpr int NAT_IT = BK * ROWB / 1024 / NW;    // DMA instructions per wave per slab
    static constexpr int TM = BM / WM / 32;
    static constexpr int TN = BN / WN / 32;
    static constexpr int A_IT = BM / (16 * NW);  // 16-row groups per wave (NW waves x 16 rows x A_IT = BM)
    static constexpr int B_IT = BN / (16 * NW);
    static constexpr int LDC = BN + VEC;
    static constexpr int A_BYTES = BM * 64;
    static constexpr int B_BYTES = DGRAD ? BK * ROWB : BN * 64;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int NSTAGE_MAX = APRO ? 2 : 3;  // slab slots: register-staged prologue 2; pure DMA 3
    static constexpr int AB_BYTES = NSTAGE_MAX * STAGE_BYTES;
    static constexpr int C_BYTES = BM * LDC * (int)sizeof(T);
    static constexpr int MAIN_BYTES = AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES;
    static constexpr int RED_BYTES = NW * BN * 2 * (int)sizeof(float);
    static constexpr int LDS_BYTES = MAIN_BYTES + RED_BYTES;
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
    static_assert(A_IT >= 1 && B_IT >= 1 && NAT_IT >= 1, "tile too small");
};

// byte offset of element column byte `cb` (byte offset inside the logical row) of natural-tile row k
template <int ROWB>
__device__ __forceinline__ int nat_off(int k, int cb) {
    const int g = (ROWB >= 256) ? (k & 3) : ((k >> 1) & 1);
    return k * ROWB + ((((cb >> 6) ^ g) << 6) | (cb & 63));
}

// swizzled position of logical chunk c (0..3) of operand row `row`
__device__ __forceinline__ int swz(int row, int c) { return c ^ ((row >> 2) & 3); }

// Shared epilogue: accumulators -> LDS tile (storage type) -> coalesced 16-byte row chunks, with the optional
// bias / residual / pooled-gradient adds, the fused ReLU gate and the per-channel statistics.
// EPI selects which epilogue features are COMPILED IN (the others cost SGPRs/VGPRs even when unused: with all of
// them the input-gradient kernels spilled scalars): 0 = statistics / residual / pooled gradient / ReLU gate,
// 1 = consumer BatchNorm apply + identity + ReLU + gate bits (msfwsi_conv_fwd_post), 2 = everything (generic kernel)
template <typename T, int BM, int BN, int WM, int WN, bool DGRAD, bool APRO, int EPI, typename Acc>
__device__ __forceinline__ void igemm_epilogue(Acc& acc /* f32x16 [TN][TM] */, const IgemmParams& prm, char* smem,
                                               int tile_m, int m0, int n0) {
    constexpr bool STD = EPI != 1, POST = EPI == 1 || EPI == 2, LORES = EPI == 3;  // 3 = class 0 + strided residual
    typedef IgemmCfg<T, BM, BN, WM, WN, DGRAD, APRO> Cfg;
    constexpr int VEC = Cfg::VEC, TM = Cfg::TM, TN = Cfg::TN, NW = Cfg::NW, NT = 64 * Cfg::NW, LDC = Cfg::LDC;
    T* Cs = reinterpret_cast<T*>(smem);                             // [BM][LDC]
    float* red = reinterpret_cast<float*>(smem + Cfg::MAIN_BYTES);  // [NW][BN][2]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, lh = lane >> 5;
    const int PQ = prm.P * prm.Q;
    // row m of this launch -> pixel index of the output tensor (identity except in parity mode: row (n, i, j) of the
    // [N][P][Q] sub-grid is pixel (n, 2i + a, 2j + b) of the [N][2P][2Q] tensor)
    auto pix_of_row = [&](int m) -> long {
        if (!DGRAD || !prm.par_mode) return (long)m;
        const unsigned n = fast_div((unsigned)m, prm.par_div_pq);
        const unsigned rem = (unsigned)m - n * (unsigned)PQ;
        const unsigned i = fast_div(rem, prm.par_div_q);
        const unsigned j = rem - i * (unsigned)prm.Q;
        return ((long)n * (2 * prm.P) + 2 * i + prm.par_a) * (2 * prm.Q) + 2 * j + prm.par_b;
    };
    // ---------------- epilogue: accumulators -> LDS tile (storage type) ----------------
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ncol = (wn * TN + tn) * 32 + 8 * g + 4 * lh;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (prm.bias != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n0 + ncol + e < prm.Nout) bv[e] = prm.bias[n0 + ncol + e];
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int row = (wm * TM + tm) * 32 + l31;
                T* dst = Cs + row * LDC + ncol;
#pragma unroll
                for (int e = 0; e < 4; ++e) store_elem<T>(dst, e, acc[tn][tm][g * 4 + e] + bv[e]);
            }
        }
    }
    // (the barrier that completes the tile follows the operand prefetch below)

    // ---------------- row-chunk pass: coalesced 16-byte stores + per-channel statistics ----------------
    constexpr int CPR = BN / VEC;        // chunks per tile row
    constexpr int RPP = NT / CPR;        // rows per pass
    const int cc = tid % CPR;
    const int rr = tid / CPR;
    const int ncol = n0 + cc * VEC;
    const bool col_ok = MSFWSI_ABLATE == 2 ? (ncol < 0) : (ncol < prm.Nout);
    T* __restrict__ out = reinterpret_cast<T*>(prm.out);
    const T* __restrict__ resid = reinterpret_cast<const T*>(prm.resid);
    float ssum[VEC], ssq[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) ssum[e] = ssq[e] = 0.f;
    const T* __restrict__ mask_c = STD ? reinterpret_cast<const T*>(prm.mask_c) : nullptr;
    const float* post_scale = POST ? prm.post_scale : nullptr;
    const T* gapg = STD ? reinterpret_cast<const T*>(prm.gapg) : nullptr;
    const unsigned char* mask_bits = STD ? prm.mask_bits : nullptr;
    unsigned char* gate_out = POST ? prm.gate_out : nullptr;
    double* stats = STD ? prm.stats : nullptr;
    const int post_relu = POST ? prm.post_relu : 0;
    float msc[VEC], msh[VEC], psc[VEC], psh[VEC];
    if (post_scale != nullptr && col_ok) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            psc[e] = post_scale[ncol + e];
            psh[e] = prm.post_shift[ncol + e];
        }
    }
    if (mask_c != nullptr && col_ok) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            msc[e] = prm.mask_scale[ncol + e];
            msh[e] = prm.mask_shift[ncol + e];
        }
    }

    // Every epilogue operand that comes from global memory (residual / identity, the gate's activation, gate bits) is
    // requested for ALL row passes up front: a short-k 1x1 tile (4-8 slabs) otherwise spends more time in its NP
    // serialised load latencies here than in its k loop (measured 3.0-3.5 TB/s on those layers).  The accumulators are
    // dead by now, so the registers are free -- up to the 128 that four waves per SIMD allow: the class that can carry
    // two operands per pass (residual AND gate activation) prefetches in groups of four passes.
    constexpr int NP = BM / RPP;
    constexpr int GP = !STD ? NP : (NW == 4 && BN == 128 ? 2 : (NP > 4 ? 4 : NP));  // 128x128 / 4 waves: 64 AGPRs stay allocated
    static_assert(NP % GP == 0, "row passes must split into whole prefetch groups");
    const int Pl = LORES ? (prm.P - 1) / prm.resid_stride + 1 : 0, Ql = LORES ? (prm.Q - 1) / prm.resid_stride + 1 : 0;
#pragma unroll 1  // a real loop: unrolled, the scheduler hoists every group's loads to the top (130 VGPRs, 3 waves/SIMD)
    for (int grp = 0; grp < NP / GP; ++grp) {
    uint4 r_res[GP], r_msk[GP];
    unsigned r_bits[GP];
    bool r_has[GP];
#pragma unroll
    for (int pi = 0; pi < GP; ++pi) {
        const int pass = grp * GP + pi;
        const int m = m0 + rr + pass * RPP;
        const bool ok = m < prm.M && col_ok;
        r_has[pi] = false;
        r_bits[pi] = 0;
        if (ok) {
            const long pixm = pix_of_row(m);
            const long off = pixm * prm.Nout + ncol;
            if (resid != nullptr) {
                if constexpr (LORES) {
                    // pixel m = (n, h, w) of the [N][P][Q] output; the residual lives on the s-strided sub-grid
                    // the stride is 2 (the only one a ResNet stage boundary has; checked by the entry point): parity
                    // test and shift instead of two runtime modulos and two divisions per row pass
                    const unsigned n = fast_div((unsigned)m, prm.div_pq);
                    const unsigned rem = (unsigned)m - n * (unsigned)PQ;
                    const unsigned h = fast_div(rem, prm.div_q);
                    const unsigned w = rem - h * (unsigned)prm.Q;
                    if (((h | w) & 1u) == 0) {
                        const long lo = (((long)n * Pl + (h >> 1)) * Ql + (w >> 1)) * prm.Nout + ncol;
                        r_res[pi] = *reinterpret_cast<const uint4*>(resid + lo);
                        r_has[pi] = true;
                    }
                } else {
                    r_res[pi] = *reinterpret_cast<const uint4*>(resid + off);
                    r_has[pi] = true;
                }
            }
            if (mask_c != nullptr) r_msk[pi] = *reinterpret_cast<const uint4*>(mask_c + off);
            else if (mask_bits != nullptr) r_bits[pi] = mask_bits[gate_off(pixm, ncol / VEC, prm.Nout / VEC)];
        }
    }
    if (grp == 0) __syncthreads();  // the transposed tile is complete in LDS

#pragma unroll
    for (int pi = 0; pi < GP; ++pi) {
        const int pass = grp * GP + pi;
        const int row = rr + pass * RPP;
        const int m = m0 + row;
        if (m < prm.M && col_ok) {
            uint4 v = *reinterpret_cast<const uint4*>(Cs + row * LDC + cc * VEC);
            const long pixm = pix_of_row(m);
            const long off = pixm * prm.Nout + ncol;
            if (resid != nullptr || gapg != nullptr || post_scale != nullptr) {
                float f[VEC];
                unpack16<T>(v, f);
                if (post_scale != nullptr) {
                    // fused BatchNorm apply of the consumer (statistics known beforehand, see conv_fwd_post)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) f[e] = fmaf(f[e], psc[e], psh[e]);
                }
                if (r_has[pi]) {
                    float g[VEC];
                    unpack16<T>(r_res[pi], g);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) f[e] += g[e];
                }
                if (gapg != nullptr) {
                    float gp[VEC];
                    unpack16<T>(*reinterpret_cast<const uint4*>(gapg + (long)(m / PQ) * prm.Nout + ncol), gp);  // m / PQ = image, also in parity mode
#pragma unroll
                    for (int e = 0; e < VEC; ++e) f[e] = fmaf(gp[e], prm.gap_scale, f[e]);
                }
                if (post_relu) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) f[e] = fmaxf(f[e], 0.f);
                }
                v = pack16<T>(f);
                if (gate_out != nullptr) {
                    // one byte per 16-byte chunk: the ReLU gate of this output for the backward pass (read there
                    // instead of the 16 bytes of the activation itself)
                    gate_out[gate_off(pixm, ncol / VEC, prm.Nout / VEC)] = (unsigned char)gate_bits_of<T>(v);
                }
            }
            if (mask_c != nullptr) {
                // fused backward of the producer's relu(bn(c)): gate, then accumulate {sum g, sum g*c}
                float f[VEC], cv[VEC];
                unpack16<T>(v, f);
                unpack16<T>(r_msk[pi], cv);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    if (!(fmaf(cv[e], msc[e], msh[e]) > 0.f)) f[e] = 0.f;
                    ssum[e] += f[e];
                    ssq[e] = fmaf(f[e], cv[e], ssq[e]);
                }
                v = pack16<T>(f);
                *reinterpret_cast<uint4*>(out + off) = v;
            } else if (mask_bits != nullptr) {
                float f[VEC];
                unpack16<T>(v, f);
                const unsigned b = r_bits[pi];
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    if (!((b >> e) & 1u)) f[e] = 0.f;
                    ssum[e] += f[e];
                }
                v = pack16<T>(f);
                *reinterpret_cast<uint4*>(out + off) = v;
            } else {
                *reinterpret_cast<uint4*>(out + off) = v;
                if (stats != nullptr) {
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

    }  // prefetch group

    if (stats != nullptr) {
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
                double* dst = stats + ((long)(tile_m % prm.nshard) * 2 + which) * prm.Nout + n0 + col;
                atomicAdd(dst, (double)t);
            }
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, bool DGRAD, bool APRO>
__global__ __launch_bounds__(64 * WM * WN) void igemm_kernel(const IgemmParams prm) {
    typedef IgemmCfg<T, BM, BN, WM, WN, DGRAD, APRO> Cfg;
    constexpr int VEC = Cfg::VEC, BK = Cfg::BK, ROWB = Cfg::ROWB;
    constexpr bool PD = !APRO;                 // pure-DMA staging: deep (3-stage) pipeline with counted vmcnt
    constexpr int NST = PD ? 3 : 2;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, A_IT = Cfg::A_IT, B_IT = Cfg::B_IT;
    constexpr int NW = Cfg::NW, NT = 64 * Cfg::NW;
    constexpr int LDC = Cfg::LDC;
    typedef typename Frag<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                                           // [NST][BM][64 B]
    char* Bs = smem + Cfg::NSTAGE_MAX * Cfg::A_BYTES;          // [NST][BN][64 B] or [NST][BK][ROWB]
    T* Cs = reinterpret_cast<T*>(smem);                        // [BM][LDC]   (after the k loop)
    float* red = reinterpret_cast<float*>(smem + Cfg::MAIN_BYTES);  // [4][BN][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, lh = lane >> 5;

    const unsigned wgid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wgid % prm.ntile_n;
    const int tile_m = wgid / prm.ntile_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const T* __restrict__ src = reinterpret_cast<const T*>(prm.src);
    const T* __restrict__ wgt = reinterpret_cast<const T*>(prm.wgt);
    const char* zero = reinterpret_cast<const char*>(g_zero_page);

    const int PQ = prm.P * prm.Q;
    const int RS = prm.R * prm.S;
    // fast path: a k-slab never straddles two filter taps, and the taps fit a 32-bit validity mask
    const bool fastk = (prm.C % BK) == 0 && RS <= 32;

    // ---- staging map: group g = it*4 + wave covers operand rows 16g .. 16g+15; lane -> (row, slot) ----
    // lane l of the group owns LDS bytes [16g*64 + 16*l, +16): row = 16g + l/4, slot = l%4, and therefore
    // fetches the logical chunk kc = slot ^ swizzle(row).
    int a_hb[A_IT], a_wb[A_IT], a_kc[A_IT];
    long a_img[A_IT];
    bool a_rowok[A_IT];
    long a_base[A_IT];       // fast path: element offset of the (r=0,s=0) source pixel, channel a_kc*VEC
    unsigned a_mask[A_IT];   // fast path: bit r*S+s set iff that tap's source pixel exists
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int row = (i * NW + wave) * 16 + (lane >> 2);
        a_kc[i] = swz(row, lane & 3);
        const int m = m0 + row;
        a_rowok[i] = m < prm.M;
        const int mm = a_rowok[i] ? m : 0;
        const int img = mm / PQ;
        const int rem = mm - img * PQ;
        const int p = rem / prm.Q;
        const int q = rem - p * prm.Q;
        a_img[i] = (long)img * prm.H * prm.W;
        if (DGRAD) {
            a_hb[i] = p + prm.pad;
            a_wb[i] = q + prm.pad;
        } else {
            a_hb[i] = p * prm.stride - prm.pad;
            a_wb[i] = q * prm.stride - prm.pad;
        }
        a_mask[i] = 0;
        a_base[i] = 0;
        if (fastk) {
            unsigned mask = 0;
            for (int r = 0; r < prm.R; ++r)
                for (int s2 = 0; s2 < prm.S; ++s2) {
                    int h, w;
                    bool ok = a_rowok[i];
                    if (DGRAD) {
                        const int th = a_hb[i] - r, tw = a_wb[i] - s2;
                        if (prm.stride == 1) {
                            h = th;
                            w = tw;
                        } else {
                            ok = ok && (((th | tw) & 1) == 0);
                            h = th >> 1;
                            w = tw >> 1;
                        }
                        ok = ok && th >= 0 && tw >= 0;
                    } else {
                        h = a_hb[i] + r;
                        w = a_wb[i] + s2;
                    }
                    ok = ok && (unsigned)h < (unsigned)prm.H && (unsigned)w < (unsigned)prm.W;
                    mask |= (ok ? 1u : 0u) << (r * prm.S + s2);
                }
            a_mask[i] = mask;
            // pixel of tap (0,0); other taps differ by a wave-uniform delta (see tap_delta below).  For the
            // stride-2 input gradient valid taps have th, r of equal parity, so (th>>1) == (hb>>1) - (r>>1).
            const int bh = (DGRAD && prm.stride == 2) ? (a_hb[i] >> 1) : a_hb[i];
            const int bw = (DGRAD && prm.stride == 2) ? (a_wb[i] >> 1) : a_wb[i];
            a_base[i] = (a_img[i] + (long)bh * prm.W + bw) * prm.C + a_kc[i] * VEC;
        }
    }
    // weight-tile lane constants
    long b_off[DGRAD ? Cfg::NAT_IT : B_IT];
    bool b_nok[DGRAD ? Cfg::NAT_IT : B_IT];
    if (!DGRAD) {
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int row = (i * NW + wave) * 16 + (lane >> 2);
            const int n = n0 + row;
            b_nok[i] = n < prm.Nout;
            b_off[i] = (long)n * prm.Ktot + swz(row, lane & 3) * VEC;
        }
    } else {
#pragma unroll
        for (int i = 0; i < Cfg::NAT_IT; ++i) {
            constexpr int CPRW = ROWB / 16;
            const int krow = (i * NW + wave) * Cfg::NAT_RPI + lane / CPRW;
            const int cp = lane % CPRW;
            const int g = (ROWB >= 256) ? (krow & 3) : ((krow >> 1) & 1);
            const int n = n0 + ((((cp >> 2) ^ g) << 2) | (cp & 3)) * VEC;
            b_nok[i] = n < prm.Nout;
            b_off[i] = (long)krow * RS * prm.Nout + n;  // source channel co = krow at tap (0,0)
        }
    }

    uint4 a_reg[A_IT];
    bool a_ok[A_IT];
    float a_sc[APRO ? A_IT : 1][VEC], a_sh[APRO ? A_IT : 1][VEC];

    // running tap state (wave-uniform)
    int tap_r = 0, tap_s = 0, tap_c = 0;

    // issue the global traffic of one k-slab: DMA straight into stage `buf`, or loads into registers
    auto fetch = [&](int k0, int buf) {
        char* Ab = As + buf * Cfg::A_BYTES;
        char* Bb = Bs + buf * Cfg::B_BYTES;
        const int tap_t = (tap_r * prm.S + tap_s) & 31;
        const bool slab_ok = true;
        long tap_delta;  // element offset of tap (r,s) relative to tap (0,0)
        if (DGRAD)
            tap_delta = prm.stride == 1 ? -((long)tap_r * prm.W + tap_s) * prm.C
                                        : -((long)(tap_r >> 1) * prm.W + (tap_s >> 1)) * prm.C;
        else
            tap_delta = ((long)tap_r * prm.W + tap_s) * prm.C;
        // ---------------- activation tile ----------------
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            bool ok;
            long off;
            int ch;
            if (fastk) {
                ok = slab_ok && ((a_mask[i] >> tap_t) & 1u);
                ch = tap_c + a_kc[i] * VEC;
                off = a_base[i] + tap_delta + tap_c;
            } else {
                const int kk = k0 + a_kc[i] * VEC;
                int r, s;
                if (RS == 1) {
                    r = 0;
                    s = 0;
                    ch = kk;
                } else {
                    const int rs = kk / prm.C;
                    ch = kk - rs * prm.C;
                    r = rs / prm.S;
                    s = rs - r * prm.S;
                }
                int h, w;
                ok = a_rowok[i] && (kk < prm.Ktot);
                if (DGRAD) {
                    const int th = a_hb[i] - r, tw = a_wb[i] - s;
                    if (prm.stride == 1) {
                        h = th;
                        w = tw;
                    } else {  // stride 2
                        ok = ok && (((th | tw) & 1) == 0);
                        h = th >> 1;
                        w = tw >> 1;
                    }
                    ok = ok && th >= 0 && tw >= 0;
                } else {
                    h = a_hb[i] + r;
                    w = a_wb[i] + s;
                }
                ok = ok && (unsigned)h < (unsigned)prm.H && (unsigned)w < (unsigned)prm.W;
                off = ok ? ((a_img[i] + (long)h * prm.W + w) * prm.C) + ch : 0;
            }
            if (APRO) {
                a_ok[i] = ok;
                a_reg[i] = make_uint4(0, 0, 0, 0);
                if (ok) {
                    a_reg[i] = *reinterpret_cast<const uint4*>(src + off);
#pragma unroll
                    for (int e = 0; e < VEC; e += 4) {
                        const float4 sc = *reinterpret_cast<const float4*>(prm.pro_scale + ch + e);
                        const float4 sh = *reinterpret_cast<const float4*>(prm.pro_shift + ch + e);
                        a_sc[i][e + 0] = sc.x; a_sc[i][e + 1] = sc.y; a_sc[i][e + 2] = sc.z; a_sc[i][e + 3] = sc.w;
                        a_sh[i][e + 0] = sh.x; a_sh[i][e + 1] = sh.y; a_sh[i][e + 2] = sh.z; a_sh[i][e + 3] = sh.w;
                    }
                }
            } else {
                const void* g = ok ? reinterpret_cast<const void*>(src + off) : reinterpret_cast<const void*>(zero);
                dma16(g, Ab + (i * NW + wave) * 1024);
            }
        }
        // ---------------- weight tile ----------------
        if (!DGRAD) {
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                bool ok = b_nok[i] && slab_ok;
                if (!fastk) ok = ok && (k0 + swz((i * NW + wave) * 16 + (lane >> 2), lane & 3) * VEC) < prm.Ktot;
                const void* g = ok ? reinterpret_cast<const void*>(wgt + b_off[i] + k0)
                                   : reinterpret_cast<const void*>(zero);
                dma16(g, Bb + (i * NW + wave) * 1024);
            }
        } else {
#pragma unroll
            for (int i = 0; i < Cfg::NAT_IT; ++i) {
                constexpr int CPRW = ROWB / 16;  // 16-byte chunks per natural row
                bool ok = b_nok[i] && slab_ok;
                long off;
                if (fastk) {
                    off = b_off[i] + ((long)tap_c * RS + tap_t) * prm.Nout;
                } else {
                    const int krow = (i * NW + wave) * Cfg::NAT_RPI + lane / CPRW;
                    const int kk = k0 + krow;
                    int r, s, co;
                    if (RS == 1) {
                        r = 0;
                        s = 0;
                        co = kk;
                    } else {
                        const int rs = kk / prm.C;
                        co = kk - rs * prm.C;
                        r = rs / prm.S;
                        s = rs - r * prm.S;
                    }
                    ok = ok && kk < prm.Ktot;
                    off = b_off[i] + ((long)(co - krow) * RS + r * prm.S + s) * prm.Nout;
                }
                const void* gp = ok ? reinterpret_cast<const void*>(wgt + off) : reinterpret_cast<const void*>(zero);
                dma16(gp, Bb + (i * NW + wave) * 1024);
            }
        }
        if (fastk) {  // advance the tap for the next slab
            tap_c += BK;
            if (tap_c >= prm.C) {
                tap_c = 0;
                if (++tap_s == prm.S) {
                    tap_s = 0;
                    ++tap_r;
                }
            }
        }
    };

    // write the register-staged part of a slab into stage `buf`
    auto commit = [&](int buf) {
        if (APRO) {
            char* Ab = As + buf * Cfg::A_BYTES;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                uint4 v = a_reg[i];
                if (a_ok[i]) {
                    float f[VEC];
                    unpack16<T>(v, f);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) f[e] = fmaxf(fmaf(f[e], a_sc[i][e], a_sh[i][e]), 0.f);
                    v = pack16<T>(f);
                }
                *reinterpret_cast<uint4*>(Ab + (i * NW + wave) * 1024 + lane * 16) = v;
            }
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

    auto compute = [&](int buf) {
        const char* Ab = As + buf * Cfg::A_BYTES;
        const char* Bb = Bs + buf * Cfg::B_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag_t xf[TM], wf[TN];
            const int cidx = ks * 2 + lh;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int row = (wm * TM + tm) * 32 + l31;
                xf[tm] = *reinterpret_cast<const frag_t*>(Ab + row * 64 + swz(row, cidx) * 16);
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int ncol = (wn * TN + tn) * 32;
                if (!DGRAD) {
                    const int row = ncol + l31;
                    wf[tn] = *reinterpret_cast<const frag_t*>(Bb + row * 64 + swz(row, cidx) * 16);
                } else if constexpr (sizeof(T) == 2) {
                    // transposed LDS read: 16-lane group G -> columns 16*(G&1).., k-half G>>1
                    const int li = lane & 15, G = lane >> 4;
                    const int q = li >> 2, p = li & 3;
                    const int kbase = ks * 16 + (G >> 1) * 8 + q;
                    const int cb = (ncol + (G & 1) * 16 + p * 4) * 2;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(Bb + nat_off<ROWB>(kbase, cb)));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(Bb + nat_off<ROWB>(kbase + 4, cb)));
                    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    wf[tn] = __builtin_bit_cast(frag_t, both);
                } else {
                    frag_t t;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        t[e] = *reinterpret_cast<const float*>(Bb + nat_off<ROWB>(ks * 8 + lh * 4 + e, (ncol + l31) * 4));
                    wf[tn] = t;
                }
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) mma_step<T>(acc[tn][tm], wf[tn], xf[tm]);
        }
    };

    // ---------------- main loop ----------------
    const int nk = (prm.Ktot + BK - 1) / BK;
    if constexpr (PD) {
        // every byte arrives by LDS-DMA: keep TWO slabs in flight.  Per wave and slab exactly DMA_PER_SLAB
        // instructions are issued, so "slab kt has landed" == at most DMA_PER_SLAB newer ones outstanding.
        constexpr int DMA_PER_SLAB = A_IT + (DGRAD ? Cfg::NAT_IT : B_IT);
        fetch(0, 0);
        if (nk > 1) fetch(BK, 1);
        int st_c = 0, st_f = 2;  // stage being computed / stage the next fetch goes to
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) {
                if constexpr (DMA_PER_SLAB == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if constexpr (DMA_PER_SLAB == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();  // slab kt visible to all waves; stage st_f (read in kt-1) is free
            asm volatile("" ::: "memory");
            if (kt + 2 < nk) fetch((kt + 2) * BK, st_f);
            compute(st_c);
            st_c = st_c == 2 ? 0 : st_c + 1;
            st_f = st_f == 2 ? 0 : st_f + 1;
        }
        __syncthreads();
    } else {
        // register-staged activations (BatchNorm prologue): one slab ahead; __syncthreads drains the DMA too
        fetch(0, 0);
        commit(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) fetch((kt + 1) * BK, buf ^ 1);
            compute(buf);
            if (kt + 1 < nk) commit(buf ^ 1);
            __syncthreads();
        }
    }

    igemm_epilogue<T, BM, BN, WM, WN, DGRAD, APRO, 2>(acc, prm, smem, tile_m, m0, n0);
}

// ---------------------------------------------------------------------------------------------
// Pure-DMA kernel for the common case (no BatchNorm prologue, C % BK == 0, R*S <= 32): every operand byte
// arrives by `buffer_load_dwordx4 ... lds`.  Per lane the byte offset of each 16-byte piece is CONSTANT for the
// whole k loop; the filter tap and channel slab move only the wave-uniform scalar offset, and out-of-image
// taps / out-of-range rows use an out-of-range offset (the buffer range check then writes zeros).  The k loop
// therefore carries almost no address arithmetic: per slab one s_mov m0 + one buffer_load per piece.
// (The generic kernel above spends >100 SALU + ~45 VALU instructions per slab on 64-bit address selects --
// measured 13 SALU + 6 VALU per MFMA -- which, not the MFMA pipe or HBM, bounded it.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rsrc, void* lds_wave_base, int voff, int soff) {
    lds_dma16_buf(rsrc, lds_wave_base, voff, soff);  // inline asm: see common.h
}

// RUN (stem): the S filter taps of one filter row are contiguous in NHWC memory when the channel count is one chunk
// (7 taps x 8 padded channels = 56 elements), so a filter row is ONE tap whose "channels" are that run, padded to 64
// with a zero weight column: the 7x7/C=3 stem becomes 7 taps x 2 slabs on this kernel instead of the generic one.
template <typename T, int BM, int BN, int WM, int WN, bool DGRAD, int EPI = 0, bool TWO = false, bool RUN = false, bool P2 = false>
__global__ __launch_bounds__(64 * WM * WN) void igemm_dma_kernel(const IgemmParams prm) {
    static_assert(!(TWO && !DGRAD && EPI != 1) && !(DGRAD && EPI != 0 && EPI != 3) && !(!DGRAD && EPI == 3),
                  "second source: input gradient or post-epilogue forward; post epilogue: forward; strided residual: dgrad");
    static_assert(!P2 || (TWO && DGRAD && EPI == 0 && sizeof(T) == 2 && MSFWSI_IGEMM_PIPE),
                  "BatchNorm + ReLU on the second source: two-source input gradient, 16-bit storage, pipelined loop");
    static_assert(!(RUN && (DGRAD || TWO || EPI != 0)), "run mode: plain forward only");
    typedef IgemmCfg<T, BM, BN, WM, WN, DGRAD, false> Cfg;
    constexpr int VEC = Cfg::VEC, BK = Cfg::BK, ROWB = Cfg::ROWB;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, A_IT = Cfg::A_IT, B_IT = Cfg::B_IT;
    constexpr int NW = Cfg::NW;
    constexpr int ES = (int)sizeof(T);
    constexpr int NB = DGRAD ? Cfg::NAT_IT : B_IT;
    constexpr int OOB = (int)0x80000000;  // byte offset beyond num_records: the load returns zeros
    typedef typename Frag<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                                   // [3][BM][64 B]
    char* Bs = smem + Cfg::NSTAGE_MAX * Cfg::A_BYTES;  // [3][BN][64 B] or [3][BK][ROWB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, lh = lane >> 5;

    const unsigned wgid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = wgid % prm.ntile_n;
    const int tile_m = wgid / prm.ntile_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int PQ = prm.P * prm.Q;
    const int RS = prm.R * prm.S;
    const int sh = (DGRAD && prm.stride == 2) ? 1 : 0;

    // source pixel (linear index over [img][H][W], may be negative inside the padding) of filter tap (0,0)
    // 1x1 / stride 1 / no padding (most launches): the source pixel IS the output pixel -- no divisions at all
    const bool unit = RS == 1 && prm.stride == 1 && prm.pad == 0;
    auto tap0_pixel = [&](int m, int& hb, int& wb) -> long {
        if (unit) {
            hb = 0;
            wb = 0;
            return m;
        }
        const int img = m / PQ;
        const int rem = m - img * PQ;
        const int p = rem / prm.Q;
        const int q = rem - p * prm.Q;
        if (DGRAD) {
            hb = p + prm.pad;
            wb = q + (prm.par_mode ? prm.pad_w : prm.pad);
        } else {
            hb = p * prm.stride - prm.pad;
            wb = q * prm.stride - prm.pad;
        }
        return ((long)img * prm.H + (hb >> sh)) * prm.W + (wb >> sh);
    };
    // reference pixel of the workgroup: source = ref + lane offset (pix - pix(m0) + W+1 >= 0, the margin covers the
    // non-monotonic rows of the stride-2 input gradient) + tap shift (maxd - tap distance >= 0)
    int hb0, wb0;
    const long maxd = DGRAD ? ((long)((prm.R - 1) >> sh) * prm.W + ((prm.S - 1) >> sh)) : 0;
    const long ref_pix = tap0_pixel(m0, hb0, wb0) - (prm.W + 1) - maxd;
    const int pitch = RUN ? prm.pix_stride : prm.C;  // elements per source pixel
    const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(reinterpret_cast<const T*>(prm.src)) + ref_pix * pitch, 0, 0x7fffffff, 0x00020000);
    // (parity mode reads its taps in place from the whole 3x3 tensor: the descriptor must span all w_RS taps)
    const long wbytes = (DGRAD && prm.par_mode) ? (long)prm.Nout * prm.w_RS * prm.C * ES : (long)prm.Nout * prm.Ktot * ES;
    const __amdgpu_buffer_rsrc_t srd_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(prm.wgt), 0, (int)(wbytes < 0x7fffffffL ? wbytes : 0x7fffffffL), 0x00020000);

    // ---- per-lane constants: group g = it*NW + wave covers operand rows 16g .. 16g+15; lane l owns the LDS
    // bytes [16g*64 + 16*l, +16) = row 16g + l/4, slot l%4, and fetches the logical chunk slot ^ swizzle(row)
    int a_voff[A_IT];
    int a_voff2[TWO ? A_IT : 1];
    unsigned a_mask[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int row = (i * NW + wave) * 16 + (lane >> 2);
        const int kc = swz(row, lane & 3);
        const int m = m0 + row;
        const bool rowok = m < prm.M;
        int hb, wb;
        const long pix = tap0_pixel(rowok ? m : m0, hb, wb);
        unsigned mask = 0;
        if constexpr (RUN) {
            // bit r*nsl + slab: this lane's chunk of that slab is pixel s = slab*(BK/pitch) + kc*(VEC/pitch) of the run
            const int nsl = prm.C / BK;
            for (int r = 0; r < prm.R; ++r)
                for (int sl = 0; sl < nsl; ++sl) {
                    const int sx = (sl * BK + kc * VEC) / pitch;
                    const bool ok = rowok && sx < prm.s_run && (unsigned)(hb + r) < (unsigned)prm.H &&
                                    (unsigned)(wb + sx) < (unsigned)prm.W;
                    mask |= (ok ? 1u : 0u) << (r * nsl + sl);
                }
        } else
        for (int r = 0; r < prm.R; ++r)
            for (int s2 = 0; s2 < prm.S; ++s2) {
                int h, w;
                bool ok = rowok;
                if (DGRAD) {
                    const int th = hb - r, tw = wb - s2;
                    if (sh) ok = ok && (((th | tw) & 1) == 0);
                    h = th >> sh;
                    w = tw >> sh;
                    ok = ok && th >= 0 && tw >= 0;
                } else {
                    h = hb + r;
                    w = wb + s2;
                }
                ok = ok && (unsigned)h < (unsigned)prm.H && (unsigned)w < (unsigned)prm.W;
                mask |= (ok ? 1u : 0u) << (r * prm.S + s2);
            }
        a_mask[i] = mask;
        a_voff[i] = (int)(((pix - ref_pix - maxd) * pitch + kc * VEC) * ES);
        if constexpr (TWO) {  // second source (two-source 1x1 input gradient): same pixel, its own row pitch
            a_voff2[i] = rowok ? (int)(((pix - ref_pix - maxd) * prm.C2 + kc * VEC) * ES) : OOB;
        }
    }
    // optional second source (1x1 input gradient with a concatenated k range)
    const __amdgpu_buffer_rsrc_t srd_a2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(reinterpret_cast<const T*>(TWO ? prm.src2 : prm.src)) + ref_pix * (TWO ? prm.C2 : prm.C), 0,
        0x7fffffff, 0x00020000);
    int b_voff[NB];
    if (!DGRAD) {
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int row = (i * NW + wave) * 16 + (lane >> 2);
            const int n = n0 + row;
            b_voff[i] = n < prm.Nout ? (int)(((long)n * prm.Ktot + swz(row, lane & 3) * VEC) * ES) : OOB;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            constexpr int CPRW = ROWB / 16;
            const int krow = (i * NW + wave) * Cfg::NAT_RPI + lane / CPRW;
            const int cp = lane % CPRW;
            const int g = (ROWB >= 256) ? (krow & 3) : ((krow >> 1) & 1);
            const int n = n0 + ((((cp >> 2) ^ g) << 2) | (cp & 3)) * VEC;
            b_voff[i] = n < prm.Nout ? (int)(((long)krow * (prm.par_mode ? prm.w_RS : RS) * prm.Nout + n) * ES) : OOB;
        }
    }

    // running tap state (wave-uniform scalars)
    int tap_r = 0, tap_s = 0, tap_c = 0, tap_t = 0, k0 = 0;
    int soff_tap = 0;
    int voff_eff[A_IT];
    // (always_inline: with the second source's early return hipcc left this lambda OUT of line in every TWO instance --
    //  a call per slab and its by-reference captures k0 / tap_c / soff_tap in scratch, 12 bytes per lane)
    auto fetch = [&](int buf) __attribute__((always_inline)) {
        if constexpr (TWO) {
            // two-source launches are 1x1 / stride 1: ONE tap, no tap state.  Source 1 holds k in [0, C), source 2 (its own
            // row pitch C2) k in [C, C + C2); the weight rows follow k.  (Written with the tap counters of the general path
            // and an early return, hipcc kept those counters in a dynamically indexed scratch slot -- 12 bytes per lane and a
            // scratch read-modify-write behind `s_waitcnt vmcnt(0)`, i.e. a drained DMA pipeline, once per tile.)
            char* Ab2 = As + buf * Cfg::A_BYTES + wave * 1024;
            char* Bb2 = Bs + buf * Cfg::B_BYTES + wave * 1024;
            if (k0 >= prm.C) {
                const int soff2 = (k0 - prm.C) * ES;
#pragma unroll
                for (int i = 0; i < A_IT; ++i) dma16_buf(srd_a2, Ab2 + i * NW * 1024, a_voff2[i], soff2);
            } else {
                const int soff1 = k0 * ES;
#pragma unroll
                for (int i = 0; i < A_IT; ++i)
                    dma16_buf(srd_a, Ab2 + i * NW * 1024, (a_mask[i] & 1u) ? a_voff[i] : OOB, soff1);
            }
            const int soffb2 = DGRAD ? (int)((long)k0 * prm.Nout * ES) : k0 * ES;  // [k][n] rows / [n][k] columns
#pragma unroll
            for (int i = 0; i < NB; ++i) dma16_buf(srd_b, Bb2 + i * NW * 1024, b_voff[i], soffb2);
            k0 += BK;
            return;
        }
        if (RUN || tap_c == 0) {  // new filter tap: which rows have a source pixel, and the tap's scalar shift
            tap_t = RUN ? tap_r * (prm.C / BK) + tap_c / BK : tap_r * prm.S + tap_s;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) voff_eff[i] = ((a_mask[i] >> tap_t) & 1u) ? a_voff[i] : OOB;
            const long d = DGRAD ? maxd - ((long)(tap_r >> sh) * prm.W + (tap_s >> sh)) : (long)tap_r * prm.W + tap_s;
            soff_tap = (int)(d * pitch * ES);
        }
        char* Ab = As + buf * Cfg::A_BYTES + wave * 1024;
        char* Bb = Bs + buf * Cfg::B_BYTES + wave * 1024;
        const int soff_a = soff_tap + tap_c * ES;
#if MSFWSI_ABLATE == 3  // diagnostic: what would a 3x3 launch cost if the taps s = 1, 2 of a filter row re-used the staged rows of s = 0
        if (!(RS == 9 && tap_s != 0))
#endif
#pragma unroll
        for (int i = 0; i < A_IT; ++i) dma16_buf(srd_a, Ab + i * NW * 1024, voff_eff[i], soff_a);
        // weight rows of this slab: channel tap_c, tap tap_t of the [k][r][s][n] tensor (parity mode: the 3x3 tensor's tap
        // (wr0 + 2 tap_r, ws0 + 2 tap_s), read in place)
        const int tap_w = (DGRAD && prm.par_mode) ? (prm.wr0 + 2 * tap_r) * prm.w_S + prm.ws0 + 2 * tap_s : tap_t;
        const int soff_b = DGRAD ? (int)(((long)tap_c * (prm.par_mode ? prm.w_RS : RS) + tap_w) * prm.Nout * ES) : k0 * ES;
#pragma unroll
        for (int i = 0; i < NB; ++i) dma16_buf(srd_b, Bb + i * NW * 1024, b_voff[i], soff_b);
        k0 += BK;
        tap_c += BK;
        // (two-source launches are 1x1: one tap, tap_c simply runs on to C.  With the wrap below compiled in, hipcc turned
        //  the two counters of those instances into a dynamically indexed scratch slot -- 12 bytes per lane, a
        //  scratch_load / add / scratch_store behind an `s_waitcnt vmcnt(0)` that drained the LDS-DMA pipeline once per tile)
        if constexpr (!TWO) {
            if (tap_c >= prm.C) {
                tap_c = 0;
                if (++tap_s == prm.S) {
                    tap_s = 0;
                    ++tap_r;
                }
            }
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

    float* p2 = reinterpret_cast<float*>(smem + Cfg::MAIN_BYTES);  // [2][C2] (P2): the epilogue's reduction area
    if constexpr (P2) {
        for (int i = tid; i < prm.C2; i += 64 * NW) {
            p2[i] = prm.pro2_scale[i];
            p2[prm.C2 + i] = prm.pro2_shift[i];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // plain loads: done before the DMA pipeline's counted waits begin
        __syncthreads();
    }
    (void)p2;
    // fragments of k-group `ks` (0 / 1) of the slab in stage `buf`
    // (kslab: first k of the slab -- P2 only: a slab of the SECOND source (k >= C) holds a raw conv output, and the operand is
    //  relu(scale * c + shift) of it, formed here on the fragment: the arithmetic of msfwsi_bn_act, so the values the MFMAs
    //  see are bit for bit those of the materialised activation.  The per-channel scale / shift sit in the epilogue's
    //  reduction area of LDS (free during the k loop): LDS reads, so the counted vmcnt of the DMA pipeline is untouched)
    auto read_group = [&](int buf, int ks, frag_t(&xf)[TM], frag_t(&wf)[TN], int kslab) __attribute__((always_inline)) {
        const char* Ab = As + buf * Cfg::A_BYTES;
        const char* Bb = Bs + buf * Cfg::B_BYTES;
        const int cidx = ks * 2 + lh;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int row = (wm * TM + tm) * 32 + l31;
            xf[tm] = *reinterpret_cast<const frag_t*>(Ab + row * 64 + swz(row, cidx) * 16);
        }
        if constexpr (P2) {
            if (kslab >= prm.C) {  // wave-uniform
                const float* ps = p2 + (kslab - prm.C) + cidx * 8;
                float sc[8], sh[8];
                *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(ps);
                *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(ps + 4);
                *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(ps + prm.C2);
                *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(ps + prm.C2 + 4);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    float f[8];
                    unpack16<T>(__builtin_bit_cast(uint4, xf[tm]), f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f);
                    xf[tm] = __builtin_bit_cast(frag_t, pack16<T>(f));
                }
            }
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int ncol = (wn * TN + tn) * 32;
            if (!DGRAD) {
                const int row = ncol + l31;
                wf[tn] = *reinterpret_cast<const frag_t*>(Bb + row * 64 + swz(row, cidx) * 16);
            } else if constexpr (sizeof(T) == 2) {
                // transposed LDS read: 16-lane group G -> columns 16*(G&1).., k-half G>>1
                const int li = lane & 15, G = lane >> 4;
                const int q = li >> 2, p = li & 3;
                const int kbase = ks * 16 + (G >> 1) * 8 + q;
                const int cb = (ncol + (G & 1) * 16 + p * 4) * 2;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(Bb + nat_off<ROWB>(kbase, cb)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(Bb + nat_off<ROWB>(kbase + 4, cb)));
                const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                wf[tn] = __builtin_bit_cast(frag_t, both);
            } else {
                frag_t t;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    t[e] = *reinterpret_cast<const float*>(Bb + nat_off<ROWB>(ks * 8 + lh * 4 + e, (ncol + l31) * 4));
                wf[tn] = t;
            }
        }
    };
    auto mma_group = [&](const frag_t(&xf)[TM], const frag_t(&wf)[TN]) __attribute__((always_inline)) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) mma_step<T>(acc[tn][tm], wf[tn], xf[tm]);
    };
    auto compute = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag_t xf[TM], wf[TN];
            read_group(buf, ks, xf, wf, 0);
            mma_group(xf, wf);
        }
    };
    (void)compute;

    // ---------------- main loop: TWO slabs in flight, counted vmcnt ----------------
    const int nk = MSFWSI_ABLATE == 1 ? 0 : prm.Ktot / BK;
    constexpr int DMA_PER_SLAB = A_IT + NB;
    static_assert(DMA_PER_SLAB >= 2 && DMA_PER_SLAB <= 6, "unexpected DMA count per slab");
#if MSFWSI_IGEMM_PIPE
    // Fragment reads software-pipelined ACROSS the slab barrier, as in csrc/wgrad.hip's pixel loop (round 6): a slab's two
    // k-groups live in two register sets; the reads of (slab kt, group 1) are issued before the MFMAs of (kt, group 0), the
    // wave then waits for its DMA pieces of slab kt+1 and for its own outstanding reads, joins the barrier, requests slab
    // kt+3 into the stage of slab kt (every wave has finished reading it: lgkmcnt(0) sits before the barrier) and reads
    // (kt+1, group 0) before it runs the MFMAs of (kt, group 1).  Before: all fragment reads of a slab and an
    // `s_waitcnt lgkmcnt(0)` stood between the barrier and the slab's first MFMA, for the four waves of a SIMD at once.
    auto wait_slabs = [&](int n) __attribute__((always_inline)) {  // all but the youngest n slabs' pieces have landed
        constexpr int D = DMA_PER_SLAB;
        if (n >= 2) {
            if constexpr (D == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if constexpr (D == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if constexpr (D == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (D == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else if (n == 1) {
            if constexpr (D == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (D == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if constexpr (D == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (D == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    if (nk > 0) {
        fetch(0);
        if (nk > 1) fetch(1);
        if (nk > 2) fetch(2);
        wait_slabs(nk >= 3 ? 2 : nk - 1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        frag_t x0[TM], w0[TN], x1[TM], w1[TN];
        read_group(0, 0, x0, w0, 0);
        int st_c = 0;
        for (int kt = 0; kt + 1 < nk; ++kt) {  // (the last slab is peeled off: one straight-line body, one set of accumulators)
            read_group(st_c, 1, x1, w1, kt * BK);
            mma_group(x0, w0);
            const int st_n = st_c == 2 ? 0 : st_c + 1;
            wait_slabs(kt + 2 < nk ? 1 : 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 3 < nk) fetch(st_c);
            read_group(st_n, 0, x0, w0, (kt + 1) * BK);
            mma_group(x1, w1);
            st_c = st_n;
        }
        read_group(st_c, 1, x1, w1, (nk - 1) * BK);
        mma_group(x0, w0);
        mma_group(x1, w1);
    }
#else
    if (nk > 0) fetch(0);
    if (nk > 1) fetch(1);
    int st_c = 0, st_f = 2;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {
            if constexpr (DMA_PER_SLAB == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (DMA_PER_SLAB == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if constexpr (DMA_PER_SLAB == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (DMA_PER_SLAB == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // slab kt visible to all waves; stage st_f (read in kt-1) is free
        asm volatile("" ::: "memory");
        // Order of the DMA requests of slab kt+2 and the MFMAs of slab kt.  Round 2 measured "MFMAs first" faster (704 ->
        // 845 TFLOP/s on the weight-gradient kernel) -- but that was with hipcc's hidden `s_waitcnt vmcnt(0)` in front of the
        // transposed reads draining the pipeline every slab (common.h, lds_dma16_buf): a request issued before the MFMAs
        // then only delayed them.  With the requests in inline asm two slabs really are in flight, and requests first is
        // the faster order again (whole step 547.5 -> 545.3 ms, A/B on one box, two rounds).
#if MSFWSI_FETCH_FIRST
        if (kt + 2 < nk) fetch(st_f);
        compute(st_c);
#else
        compute(st_c);
        if (kt + 2 < nk) fetch(st_f);
#endif
        st_c = st_c == 2 ? 0 : st_c + 1;
        st_f = st_f == 2 ? 0 : st_f + 1;
    }
#endif
    __syncthreads();
    igemm_epilogue<T, BM, BN, WM, WN, DGRAD, false, EPI>(acc, prm, smem, tile_m, m0, n0);
}

msfwsi_tunable g_fast_dma{1};  // tunable through msfwsi_set_tuning(1, .): 0 = always the generic kernel

template <typename T, int BM, int BN, int WM, int WN, bool DGRAD, bool APRO>
int launch_igemm(IgemmParams& prm, hipStream_t stream) {
    typedef IgemmCfg<T, BM, BN, WM, WN, DGRAD, APRO> Cfg;
    const int ntm = (prm.M + BM - 1) / BM;
    prm.ntile_n = (prm.Nout + BN - 1) / BN;
    const long nblk = (long)ntm * prm.ntile_n;
    if (nblk <= 0 || nblk > 0x7fffffffL) return MSFWSI_EINVAL;
    // pure-DMA fast kernel when no BatchNorm prologue is applied and a k-slab never straddles two filter taps
    void (*kern)(const IgemmParams) = igemm_kernel<T, BM, BN, WM, WN, DGRAD, APRO>;
    bool dma = false;
    if constexpr (!APRO) {
        if (g_fast_dma && prm.C % Cfg::BK == 0 && prm.R * prm.S <= 32) {
            dma = true;
            if constexpr (DGRAD) {
                kern = prm.src2 != nullptr ? igemm_dma_kernel<T, BM, BN, WM, WN, true, 0, true>
                                           : igemm_dma_kernel<T, BM, BN, WM, WN, true, 0, false>;
                if (prm.pro2_scale != nullptr) {
                    if constexpr (sizeof(T) == 2 && MSFWSI_IGEMM_PIPE) {
                        if (prm.src2 == nullptr || prm.resid_stride > 1 || 2 * prm.C2 * (int)sizeof(float) > Cfg::RED_BYTES)
                            return MSFWSI_EUNSUPPORTED;
                        kern = igemm_dma_kernel<T, BM, BN, WM, WN, true, 0, true, false, true>;
                    } else {
                        return MSFWSI_EUNSUPPORTED;
                    }
                }
                if (prm.resid_stride > 1) {
                    if (prm.src2 != nullptr) return MSFWSI_EUNSUPPORTED;
                    kern = igemm_dma_kernel<T, BM, BN, WM, WN, true, 3, false>;
                }
            } else {
                kern = prm.post_scale != nullptr ? igemm_dma_kernel<T, BM, BN, WM, WN, false, 1, false>
                                                 : igemm_dma_kernel<T, BM, BN, WM, WN, false, 0, false>;
                if (prm.src2 != nullptr) {
                    if (prm.post_scale == nullptr) return MSFWSI_EUNSUPPORTED;
                    kern = igemm_dma_kernel<T, BM, BN, WM, WN, false, 1, true>;
                }
                if constexpr (BN == 64) {
                    if (prm.pix_stride > 0) kern = igemm_dma_kernel<T, BM, BN, WM, WN, false, 0, false, true>;
                }
            }
        }
    }
    if ((prm.src2 != nullptr || prm.resid_stride > 1 || prm.pro2_scale != nullptr) && !dma) return MSFWSI_EUNSUPPORTED;  // pure-DMA kernel features
    if (Cfg::LDS_BYTES > 64 * 1024) {
        if (int e = msfwsi_raise_lds(reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES)) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(64 * WM * WN), Cfg::LDS_BYTES, stream, prm);
    return msfwsi_launch_status();
}

msfwsi_tunable g_s2_parity{1};               // key 5: stride-2 3x3 input gradients by output-pixel parity (0 = one masked launch)
msfwsi_tunable g_big_tile_min_blocks{1024};  // tunable through msfwsi_set_tuning
msfwsi_tunable g_small_grid_blocks{100};     // key 4: 128x128 grids below this use 128x64 tiles (measured: helps <= 72 tiles, hurts at 144+)

template <typename T, bool DGRAD, bool APRO>
int dispatch_tile(IgemmParams& prm, hipStream_t stream) {
    if (prm.Nout <= 64) return launch_igemm<T, 128, 64, 2, 2, DGRAD, APRO>(prm, stream);
    // 256x128 / 8 waves moves 25 % fewer operand bytes through L1/LDS per MFMA (the per-CU 64 B/clk vector
    // memory path, not HBM, bounds these kernels); keep 128x128 while the grid would not fill the chip
    // (the register-staged BatchNorm-prologue variant loses with 8 waves: measured 0.55 -> 0.76 ms)
    if (!APRO && sizeof(T) == 2 && (long)((prm.M + 255) / 256) * ((prm.Nout + 127) / 128) >= g_big_tile_min_blocks)
    {
        return launch_igemm<T, 256, 128, 4, 2, DGRAD, APRO>(prm, stream);
    }
    // grids far below one workgroup per CU (small head GEMMs): halve the tile to put more CUs to work
    if (!APRO && (long)((prm.M + 127) / 128) * ((prm.Nout + 127) / 128) < g_small_grid_blocks)
        return launch_igemm<T, 128, 64, 2, 2, DGRAD, APRO>(prm, stream);
    return launch_igemm<T, 128, 128, 2, 2, DGRAD, APRO>(prm, stream);
}

int check_desc(const msfwsi_conv_desc* d) {
    if (d == nullptr) return MSFWSI_EINVAL;
    if (!msfwsi_dtype_ok(d->dtype)) return MSFWSI_EUNSUPPORTED;
    const int vec = msfwsi_vec_of(d->dtype);
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->K <= 0 || d->P <= 0 || d->Q <= 0) return MSFWSI_EINVAL;
    if (d->R <= 0 || d->S <= 0 || d->pad < 0) return MSFWSI_EINVAL;
    if (d->stride != 1 && d->stride != 2) return MSFWSI_EUNSUPPORTED;
    if (d->C % vec != 0 || d->K % vec != 0) return MSFWSI_EUNSUPPORTED;
    // output extent must be the convolution's
    if (d->P != (d->H + 2 * d->pad - d->R) / d->stride + 1) return MSFWSI_EINVAL;
    if (d->Q != (d->W + 2 * d->pad - d->S) / d->stride + 1) return MSFWSI_EINVAL;
    if ((long)d->N * d->P * d->Q > 0x7fffffffL || (long)d->N * d->H * d->W > 0x7fffffffL) return MSFWSI_EINVAL;
    return MSFWSI_OK;
}

}  // namespace

extern "C" __attribute__((visibility("hidden"))) long msfwsi_wgrad_set_lin(long v, int write);
extern "C" __attribute__((visibility("hidden"))) long msfwsi_wgrad_set_big(long v, int write);
extern "C" __attribute__((visibility("hidden"))) long msfwsi_c3_set_stationary(long v, int write);
extern "C" __attribute__((visibility("hidden"))) long msfwsi_wgrad_set_os(long v, int write);
extern "C" __attribute__((visibility("hidden"))) long msfwsi_wgrad_set_os_min(long v, int write);
extern "C" __attribute__((visibility("hidden"))) long msfwsi_stem_set_ws(long v, int write);
extern "C" __attribute__((visibility("hidden"))) long msfwsi_stem_set_os_min(long v, int write);
extern "C" __attribute__((visibility("hidden"))) long msfwsi_wgrad_set_max_splits(long v, int write);
extern "C" __attribute__((visibility("hidden"))) long msfwsi_pool_bwd_set_walk(long v, int write);
extern "C" __attribute__((visibility("hidden"))) long msfwsi_panel_set_hand(long v, int write);
extern "C" __attribute__((visibility("hidden"))) long msfwsi_panel_set_wide(long v, int write);

namespace {
long own_tunable(msfwsi_tunable& g, long v, int write) {
    const long old = g;
    if (write) g = v;
    return old;
}
// the switch behind `key`: its value before the call in *old, rewritten when `write`; false for an unknown key
bool tuning_access(int key, long v, int write, long* old) {
    switch (key) {
        case 0: *old = own_tunable(g_big_tile_min_blocks, v, write); return true;
        case 1: *old = own_tunable(g_fast_dma, v, write); return true;
        case 2: *old = msfwsi_wgrad_set_lin(v, write); return true;
        case 4: *old = own_tunable(g_small_grid_blocks, v, write); return true;
        case 5: *old = own_tunable(g_s2_parity, v, write); return true;
        case 6: *old = msfwsi_wgrad_set_big(v, write); return true;
        case 9: *old = msfwsi_c3_set_stationary(v, write); return true;
        case 10: *old = msfwsi_wgrad_set_os(v, write); return true;
        case 11: *old = msfwsi_wgrad_set_os_min(v, write); return true;
        case 12: *old = msfwsi_stem_set_ws(v, write); return true;
        case 13: *old = msfwsi_stem_set_os_min(v, write); return true;
        case 15: *old = msfwsi_wgrad_set_max_splits(v, write); return true;
        case 16: *old = msfwsi_pool_bwd_set_walk(v, write); return true;
        case 17: *old = msfwsi_panel_set_hand(v, write); return true;
        case 18: *old = msfwsi_panel_set_wide(v, write); return true;
    }
    return false;
}
}  // namespace

extern "C" int msfwsi_set_tuning(int key, long value) {
    long old;
    return tuning_access(key, value, 1, &old) ? MSFWSI_OK : MSFWSI_EINVAL;
}

extern "C" int msfwsi_get_tuning(int key, long* value) {
    MSFWSI_CHECK_ARG(value != nullptr);
    return tuning_access(key, 0, 0, value) ? MSFWSI_OK : MSFWSI_EINVAL;
}

extern "C" int msfwsi_conv_fwd(const msfwsi_conv_desc* d, const void* x, const void* w, void* y,
                               const float* pro_scale, const float* pro_shift, const float* bias,
                               double* stats, int nshard, void* stream) {
    int rc = check_desc(d);
    if (rc != MSFWSI_OK) return rc;
    MSFWSI_CHECK_ARG(x != nullptr && w != nullptr && y != nullptr);
    MSFWSI_CHECK_ARG((pro_scale == nullptr) == (pro_shift == nullptr));
    MSFWSI_CHECK_ARG(stats == nullptr || nshard >= 1);
    IgemmParams prm{};
    prm.src = x; prm.wgt = w; prm.out = y;
    prm.pro_scale = pro_scale; prm.pro_shift = pro_shift; prm.bias = bias;
    prm.stats = stats; prm.nshard = nshard > 0 ? nshard : 1;
    prm.N = d->N; prm.H = d->H; prm.W = d->W; prm.C = d->C;
    prm.P = d->P; prm.Q = d->Q; prm.Nout = d->K;
    prm.R = d->R; prm.S = d->S; prm.stride = d->stride; prm.pad = d->pad;
    prm.M = d->N * d->P * d->Q;
    prm.Ktot = d->R * d->S * d->C;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool pro = pro_scale != nullptr;
    MSFWSI_WITH_T(d->dtype, return pro ? dispatch_tile<T, false, true>(prm, st) : dispatch_tile<T, false, false>(prm, st));
    return MSFWSI_EINVAL;
}

extern "C" __attribute__((visibility("hidden"))) int msfwsi_stem_ws_fwd(int dtype, const void* x, const void* w, void* y,
                                                                       double* stats, int nshard, int N, int H, int W,
                                                                       int CP, int K, int R, int S, int stride, int pad,
                                                                       int P, int Q, void* stream);

extern "C" int msfwsi_stem_conv_fwd(int dtype, const void* x, const void* w_run, void* y, double* stats, int nshard,
                                    int N, int H, int W, int CP, int K, int R, int S, int stride, int pad, int P,
                                    int Q, void* stream) {
    MSFWSI_CHECK_ARG(msfwsi_dtype_ok(dtype) && x && w_run && y && N > 0 && H > 0 && W > 0 && K > 0 && K <= 64);
    MSFWSI_CHECK_ARG(stats == nullptr || nshard >= 1);
    {   // the space-to-depth stem (4x4 / stride 1 / pad 2 over 16 channels): weights-stationary kernel (stem.hip)
        const int rc = msfwsi_stem_ws_fwd(dtype, x, w_run, y, stats, nshard, N, H, W, CP, K, R, S, stride, pad, P, Q, stream);
        if (rc != MSFWSI_EUNSUPPORTED) return rc;
    }
    const int vec = msfwsi_vec_of(dtype), bk = dtype == MSFWSI_DT_F32 ? 16 : 32;
    // the run of S pixels x CP channels is padded to whole k slabs; a 16-byte chunk must be one pixel
    const int run = ((S * CP + bk - 1) / bk) * bk;
    if (CP % vec != 0 || K % vec != 0 || R * (run / bk) > 32 || (stride != 1 && stride != 2) || !g_fast_dma)
        return MSFWSI_EUNSUPPORTED;
    IgemmParams prm{};
    prm.src = x; prm.wgt = w_run; prm.out = y;
    prm.stats = stats; prm.nshard = nshard > 0 ? nshard : 1;
    prm.N = N; prm.H = H; prm.W = W; prm.C = run;
    // P, Q > 0: an explicitly cropped output extent (asymmetric padding: `pad` rows above / left, fewer below / right)
    prm.P = P > 0 ? P : (H + 2 * pad - R) / stride + 1; prm.Q = Q > 0 ? Q : (W + 2 * pad - S) / stride + 1; prm.Nout = K;
    if (prm.P > (H + 2 * pad - R) / stride + 1 || prm.Q > (W + 2 * pad - S) / stride + 1) return MSFWSI_EINVAL;
    prm.R = R; prm.S = 1; prm.stride = stride; prm.pad = pad;
    prm.pix_stride = CP; prm.s_run = S;
    if ((long)N * prm.P * prm.Q > 0x7fffffffL) return MSFWSI_EINVAL;
    prm.M = N * prm.P * prm.Q;
    prm.Ktot = R * run;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    MSFWSI_WITH_T(dtype, return (launch_igemm<T, 128, 64, 2, 2, false, false>(prm, st)));
    return MSFWSI_EINVAL;
}

extern "C" int msfwsi_conv_dgrad2(const msfwsi_conv_desc* d, const void* dy, const void* w_cat, void* dx,
                                  const void* src2, int C2, const float* bias, const void* mask_c,
                                  const float* mask_scale, const float* mask_shift, double* sums, int nshard,
                                  void* stream) {
    int rc = check_desc(d);
    if (rc != MSFWSI_OK) return rc;
    MSFWSI_CHECK_ARG(dy != nullptr && w_cat != nullptr && dx != nullptr && src2 != nullptr && C2 > 0);
    MSFWSI_CHECK_ARG((mask_c == nullptr) == (mask_scale == nullptr) && (mask_c == nullptr) == (mask_shift == nullptr));
    MSFWSI_CHECK_ARG((mask_c == nullptr) == (sums == nullptr) && (sums == nullptr || nshard >= 1));
    const int bk = d->dtype == MSFWSI_DT_F32 ? 16 : 32;
    // only the pure-DMA kernel knows the second source: 1x1 / stride 1, whole k slabs in both ranges
    if (d->R != 1 || d->S != 1 || d->stride != 1 || d->pad != 0 || d->K % bk != 0 || C2 % bk != 0 || !g_fast_dma)
        return MSFWSI_EUNSUPPORTED;
    IgemmParams prm{};
    prm.src = dy; prm.wgt = w_cat; prm.out = dx;
    prm.src2 = src2; prm.C2 = C2; prm.bias = bias;
    prm.mask_c = mask_c; prm.mask_scale = mask_scale; prm.mask_shift = mask_shift;
    prm.stats = sums; prm.nshard = nshard > 0 ? nshard : 1;
    prm.N = d->N; prm.H = d->P; prm.W = d->Q; prm.C = d->K;
    prm.P = d->H; prm.Q = d->W; prm.Nout = d->C;
    prm.R = 1; prm.S = 1; prm.stride = 1; prm.pad = 0;
    prm.M = d->N * d->H * d->W;
    prm.Ktot = d->K + C2;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    MSFWSI_WITH_T(d->dtype, return dispatch_tile<T, true, false>(prm, st));
    return MSFWSI_EINVAL;
}

extern "C" int msfwsi_conv_dgrad2_pro(const msfwsi_conv_desc* d, const void* dy, const void* w_cat, void* dx,
                                      const void* c2, int C2, const float* pro_scale, const float* pro_shift,
                                      const float* bias, const void* mask_c, const float* mask_scale,
                                      const float* mask_shift, double* sums, int nshard, void* stream) {
    int rc = check_desc(d);
    if (rc != MSFWSI_OK) return rc;
    MSFWSI_CHECK_ARG(dy != nullptr && w_cat != nullptr && dx != nullptr && c2 != nullptr && C2 > 0);
    MSFWSI_CHECK_ARG(pro_scale != nullptr && pro_shift != nullptr);
    MSFWSI_CHECK_ARG((mask_c == nullptr) == (mask_scale == nullptr) && (mask_c == nullptr) == (mask_shift == nullptr));
    MSFWSI_CHECK_ARG((mask_c == nullptr) == (sums == nullptr) && (sums == nullptr || nshard >= 1));
    // 16-bit storage, 1x1 / stride 1, whole k slabs in both ranges: the pure-DMA kernel's two-source instance
    if (d->dtype == MSFWSI_DT_F32 || d->R != 1 || d->S != 1 || d->stride != 1 || d->pad != 0 || d->K % 32 != 0 ||
        C2 % 32 != 0 || !g_fast_dma)
        return MSFWSI_EUNSUPPORTED;
    IgemmParams prm{};
    prm.src = dy; prm.wgt = w_cat; prm.out = dx;
    prm.src2 = c2; prm.C2 = C2; prm.bias = bias;
    prm.pro2_scale = pro_scale; prm.pro2_shift = pro_shift;
    prm.mask_c = mask_c; prm.mask_scale = mask_scale; prm.mask_shift = mask_shift;
    prm.stats = sums; prm.nshard = nshard > 0 ? nshard : 1;
    prm.N = d->N; prm.H = d->P; prm.W = d->Q; prm.C = d->K;
    prm.P = d->H; prm.Q = d->W; prm.Nout = d->C;
    prm.R = 1; prm.S = 1; prm.stride = 1; prm.pad = 0;
    prm.M = d->N * d->H * d->W;
    prm.Ktot = d->K + C2;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    MSFWSI_WITH_T(d->dtype, return dispatch_tile<T, true, false>(prm, st));
    return MSFWSI_EINVAL;
}

extern "C" int msfwsi_conv_fwd_post(const msfwsi_conv_desc* d, const void* x, const void* w, void* y,
                                    const float* post_scale, const float* post_shift, const void* ident, int relu,
                                    unsigned char* gate_out, void* stream) {
    int rc = check_desc(d);
    if (rc != MSFWSI_OK) return rc;
    MSFWSI_CHECK_ARG(x != nullptr && w != nullptr && y != nullptr && post_scale != nullptr && post_shift != nullptr);
    IgemmParams prm{};
    prm.src = x; prm.wgt = w; prm.out = y;
    prm.post_scale = post_scale; prm.post_shift = post_shift; prm.post_relu = relu ? 1 : 0;
    prm.resid = ident;
    prm.gate_out = gate_out;
    prm.nshard = 1;
    prm.N = d->N; prm.H = d->H; prm.W = d->W; prm.C = d->C;
    prm.P = d->P; prm.Q = d->Q; prm.Nout = d->K;
    prm.R = d->R; prm.S = d->S; prm.stride = d->stride; prm.pad = d->pad;
    prm.M = d->N * d->P * d->Q;
    prm.Ktot = d->R * d->S * d->C;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    MSFWSI_WITH_T(d->dtype, return dispatch_tile<T, false, false>(prm, st));
    return MSFWSI_EINVAL;
}

extern "C" int msfwsi_conv_fwd_post2(const msfwsi_conv_desc* d, const void* x, const void* w_cat, void* y,
                                     const void* src2, int C2, const float* post_scale, const float* post_shift,
                                     const void* ident, int relu, unsigned char* gate_out, void* stream) {
    int rc = check_desc(d);
    if (rc != MSFWSI_OK) return rc;
    MSFWSI_CHECK_ARG(x && w_cat && y && src2 && C2 > 0 && post_scale && post_shift);
    const int bk = d->dtype == MSFWSI_DT_F32 ? 16 : 32;
    if (d->R != 1 || d->S != 1 || d->stride != 1 || d->pad != 0 || d->C % bk != 0 || C2 % bk != 0 || !g_fast_dma)
        return MSFWSI_EUNSUPPORTED;
    IgemmParams prm{};
    prm.src = x; prm.wgt = w_cat; prm.out = y;
    prm.src2 = src2; prm.C2 = C2;
    prm.post_scale = post_scale; prm.post_shift = post_shift; prm.post_relu = relu ? 1 : 0;
    prm.resid = ident;
    prm.gate_out = gate_out;
    prm.nshard = 1;
    prm.N = d->N; prm.H = d->H; prm.W = d->W; prm.C = d->C;
    prm.P = d->P; prm.Q = d->Q; prm.Nout = d->K;
    prm.R = 1; prm.S = 1; prm.stride = 1; prm.pad = 0;
    prm.M = d->N * d->P * d->Q;
    prm.Ktot = d->C + C2;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    MSFWSI_WITH_T(d->dtype, return dispatch_tile<T, false, false>(prm, st));
    return MSFWSI_EINVAL;
}

extern "C" int msfwsi_conv_dgrad(const msfwsi_conv_desc* d, const void* dy, const void* w, void* dx,
                                 const void* resid, const void* gapg, float gap_scale, const void* mask_c,
                                 const float* mask_scale, const float* mask_shift, const unsigned char* mask_bits,
                                 double* sums, int nshard, int resid_stride, void* stream) {
    int rc = check_desc(d);
    if (rc != MSFWSI_OK) return rc;
    MSFWSI_CHECK_ARG(dy != nullptr && w != nullptr && dx != nullptr);
    MSFWSI_CHECK_ARG(resid_stride >= 0 && (resid_stride <= 1 || resid != nullptr));
    if (resid_stride > 2) return MSFWSI_EUNSUPPORTED;  // the kernel's strided-residual class tests pixel parity
    MSFWSI_CHECK_ARG((mask_c == nullptr) == (mask_scale == nullptr) && (mask_c == nullptr) == (mask_shift == nullptr));
    MSFWSI_CHECK_ARG(mask_c == nullptr || mask_bits == nullptr);
    MSFWSI_CHECK_ARG((mask_c == nullptr && mask_bits == nullptr) == (sums == nullptr) && (sums == nullptr || nshard >= 1));
    IgemmParams prm{};
    prm.src = dy; prm.wgt = w; prm.out = dx;
    prm.resid = resid; prm.gapg = gapg; prm.gap_scale = gap_scale;
    prm.mask_c = mask_c; prm.mask_scale = mask_scale; prm.mask_shift = mask_shift;
    prm.mask_bits = mask_bits;
    prm.resid_stride = resid_stride > 1 ? resid_stride : 0;
    prm.div_pq = make_fastdiv((unsigned)(d->H * d->W));
    prm.div_q = make_fastdiv((unsigned)d->W);
    prm.stats = sums; prm.nshard = nshard > 0 ? nshard : 1;
    // source = dY [N,P,Q,K]; output = dX [N,H,W,C]
    prm.N = d->N; prm.H = d->P; prm.W = d->Q; prm.C = d->K;
    prm.P = d->H; prm.Q = d->W; prm.Nout = d->C;
    prm.R = d->R; prm.S = d->S; prm.stride = d->stride; prm.pad = d->pad;
    prm.M = d->N * d->H * d->W;
    prm.Ktot = d->R * d->S * d->K;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int bk = d->dtype == MSFWSI_DT_F32 ? 16 : 32;
    if (g_s2_parity && g_fast_dma && d->stride == 2 && d->R == 3 && d->S == 3 && d->pad == 1 && d->H % 2 == 0 &&
        d->W % 2 == 0 && d->P == d->H / 2 && d->Q == d->W / 2 && d->K % bk == 0 && resid_stride <= 1) {
        // four launches, one per parity (a, b) of the dX pixel: see IgemmParams::par_mode
        prm.par_mode = 1; prm.stride = 1; prm.w_S = 3; prm.w_RS = 9;
        prm.P = d->P; prm.Q = d->Q;
        prm.M = d->N * d->P * d->Q;
        prm.par_div_pq = make_fastdiv((unsigned)(d->P * d->Q));
        prm.par_div_q = make_fastdiv((unsigned)d->Q);
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) {
                prm.par_a = a; prm.par_b = b;
                prm.R = a ? 2 : 1; prm.S = b ? 2 : 1;
                prm.pad = a; prm.pad_w = b;
                prm.wr0 = a ? 0 : 1; prm.ws0 = b ? 0 : 1;
                prm.Ktot = prm.R * prm.S * d->K;
                int rc2 = MSFWSI_EINVAL;
                MSFWSI_WITH_T(d->dtype, rc2 = (dispatch_tile<T, true, false>(prm, st)));
                if (rc2 != MSFWSI_OK) return rc2;
            }
        return MSFWSI_OK;
    }
    MSFWSI_WITH_T(d->dtype, return dispatch_tile<T, true, false>(prm, st));
    return MSFWSI_EINVAL;
}
