// Activation-stationary ("panel") 1x1 convolution for the short-k / wide-output layers of a Bottleneck on gfx950:
//
//   out[m][n] = epi( sum_k A(m,k) * B(k,n) ),      k = K <= 512 operand channels, n = Nout >= 128 output channels
//
//   forward  : conv3 of a Bottleneck (w -> 4w, reference src/models/resnet.py:131-138) with bn3 apply + identity + ReLU
//              (+ gate bits) in the epilogue, A = relu(bn2(c2)) formed from conv2's RAW output while the panel is staged
//              (resnet.py:128-130): the normalised activation a2 never exists in HBM.
//   backward : input gradient of conv1 (4w <- w, resnet.py:124), A = dc1 = k1*g + k2*c1 + k3 -- bn1's backward
//              (autograd's batch_norm_backward, tools/ssl_train.py:472) -- formed from the gated gradient g and the raw
//              conv output c1 while the panel is staged and written back once for the weight gradient; epilogue = the
//              identity-path gradient + the previous block's closing ReLU gate (bits) + sum(g) (+ pooled-feature gradient,
//              + the low-resolution residual of a strided downsample branch).
//
// Why a second kernel beside igemm_dma_kernel.  There a 256 x 128 tile walks its k range slab by slab with two slabs in
// flight; the n-tiles of one row block run side by side on one XCD and all wait for the same HBM miss of each slab, so a
// short-k launch is a chain of HBM latencies (8 slabs, 2 in flight) FOLLOWED by an epilogue whose operands can only be
// requested once the accumulators are dead -- the launch costs the SUM of the two phases (profiles/r04_ablation_short_k.txt).
// Here:
//   * the [BM x K] operand panel of a workgroup is requested AT ONCE (BM*K*2 bytes in flight per workgroup), transformed
//     in registers and parked in LDS for the workgroup's life: it is read from HBM once instead of once per n-tile
//     through L2, and the BatchNorm map costs no pass of its own;
//   * after ONE barrier every wave is on its own: wave w owns the 32-channel output blocks w, w+4, ... of all BM rows.
//     Weight fragments come straight from global memory (L2-resident, pre-packed in MFMA fragment order: one coalesced
//     1-KiB load per fragment, msfwsi_panel_pack_weights) into registers, a whole block ahead; the epilogue operands of
//     a block are requested before its MFMAs; the accumulators are transposed through a WAVE-PRIVATE LDS scratch.  No
//     barrier in the loop: one wave's epilogue traffic runs beside its neighbours' MFMAs and loads.
// 16-bit storage types only (fp32 launches keep the gather kernel).
#include "common.h"
#include "../../include/msfwsi_hip.h"

namespace {

struct PanelParams {
    const void* src;    // [M][K]: PRO 0 the operand itself, PRO 1 the producer's raw conv output, PRO 2 the gated gradient g
    const void* src_c;  // PRO 2: the raw conv output c whose BatchNorm is differentiated
    const float* p0;    // PRO 1: scale   PRO 2: k1   (per operand channel)
    const float* p1;    // PRO 1: shift   PRO 2: k2
    const float* p2;    //                PRO 2: k3
    void* aout;         // PRO 2, nullable: the transformed operand (dc) written back [M][K]
    const void* wpk;    // packed weights [Nout/32][K/16][64 lanes][8]
    void* out;          // [M][Nout]
    // EPI 1 (forward): out = [relu]( round(acc) * post_scale + post_shift + ident ), gate bits out
    const float* post_scale;
    const float* post_shift;
    const void* ident;  // [M][Nout], nullable
    int post_relu;
    unsigned char* gate_out;  // [M][Nout/8], nullable
    // EPI 0 / 3 (input gradient): out = gate( round(acc) + resid + gap_scale * gapg[image] ), sums[shard][0][n] += out
    const void* resid;  // [M][Nout] (EPI 3: [N][P/2][Q/2][Nout], added where h % 2 == w % 2 == 0), nullable
    const void* gapg;   // [N][Nout], nullable
    float gap_scale;
    const unsigned char* mask_bits;  // [M][Nout/8], nullable
    double* sums;                    // [nshard][2][Nout], nullable (slot 0)
    int nshard;
    int M, Nout, P, Q;
    FastDiv div_pq, div_q;
};

template <int K>
__device__ __forceinline__ int panel_swz(int row) {
    // chunk-index XOR that spreads the 16 rows of a ds_read_b128 lane group over the 16 16-byte slots of a 256-byte bank
    // row: rows are K*2 bytes apart (a multiple of 256 for K >= 128; 128 bytes for K = 64, where bit 0 of the row picks
    // the half of the bank row)
    return K >= 128 ? (row & 15) : ((row >> 1) & 7);
}

template <typename T>
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d);
template <>
__device__ __forceinline__ uint2 pack4<__bf16>(float a, float b, float c, float d) {
    return make_uint2(pack2_bf16(a, b), pack2_bf16(c, d));
}
template <>
__device__ __forceinline__ uint2 pack4<_Float16>(float a, float b, float c, float d) {
    return make_uint2(pack2_f16(a, b), pack2_f16(c, d));
}

// PRO: 0 none, 1 relu(scale*c + shift), 2 k1*g + k2*c + k3.   EPI: 1 forward post, 0 input gradient, 3 input gradient with
// the low-resolution (stride 2) residual.
template <typename T, int K, int BM, int PRO, int EPI>
__global__ __launch_bounds__(256, 2) void panel_kernel(const PanelParams prm) {
    constexpr int NT = 256, NW = 4;
    constexpr int CPR = K / 8;       // 16-byte chunks per operand row
    constexpr int RPP = NT / CPR;    // rows staged per pass
    constexpr int NPASS = BM / RPP;
    constexpr int TM = BM / 32;      // 32-row MFMA tiles per wave (every wave covers all BM rows)
    constexpr int KS = K / 16;       // MFMA k steps
    constexpr int ROWB = K * 2;
    constexpr int SCR_PITCH = 80;    // bytes per scratch row (32 channels = 64 bytes + 16: keeps ds_read_b128 aligned)
    constexpr int SCR_BYTES = 32 * SCR_PITCH;
    constexpr bool FWD = EPI == 1, LORES = EPI == 3;
    static_assert(K % 64 == 0 && BM % RPP == 0 && NT % CPR == 0 && BM % 32 == 0, "panel geometry");
    typedef typename MmaFrag<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* panel = smem;  // [BM][ROWB], chunk index XOR-swizzled by panel_swz(row)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    char* scratch = smem + BM * ROWB + wave * SCR_BYTES;  // [32][SCR_PITCH], private to this wave
    const long m0 = (long)blockIdx.x * BM;
    const int nblk = prm.Nout >> 5;

    // ---------------- stage the operand panel: every pass requested up front, transformed in registers ----------------
    // (loads are unconditional from a clamped row -- a row past the tensor's end re-reads the last valid one and its
    //  results are never stored: a divergent branch around each load made hipcc drain the queue between passes)
    const int rows_left = (int)((long)prm.M - m0 < BM ? (long)prm.M - m0 : BM);  // rows of this panel inside the tensor
    {
        const int cc = tid % CPR, rr = tid / CPR;
        const char* src_wg = reinterpret_cast<const char*>(prm.src) + m0 * ROWB;
        const char* srcc_wg = PRO == 2 ? reinterpret_cast<const char*>(prm.src_c) + m0 * ROWB : nullptr;
        uint4 v[NPASS], vc[PRO == 2 ? NPASS : 1];
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int row = rr + p * RPP;
            const unsigned off = (unsigned)(row < rows_left ? row : rows_left - 1) * ROWB + cc * 16;
            v[p] = *reinterpret_cast<const uint4*>(src_wg + off);
            if constexpr (PRO == 2) vc[p] = *reinterpret_cast<const uint4*>(srcc_wg + off);
        }
        float c0[PRO ? 8 : 1], c1[PRO ? 8 : 1], c2[PRO == 2 ? 8 : 1];
        if constexpr (PRO != 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                c0[e] = prm.p0[cc * 8 + e];
                c1[e] = prm.p1[cc * 8 + e];
                if constexpr (PRO == 2) c2[e] = prm.p2[cc * 8 + e];
            }
        }
        char* aout_wg = PRO == 2 && prm.aout != nullptr ? reinterpret_cast<char*>(prm.aout) + m0 * ROWB : nullptr;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int row = rr + p * RPP;
            uint4 t = v[p];
            if constexpr (PRO == 1) {
                float f[8];
                unpack16<T>(t, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaxf(fmaf(f[e], c0[e], c1[e]), 0.f);
                t = pack16<T>(f);
            } else if constexpr (PRO == 2) {
                float g[8], c[8];
                unpack16<T>(t, g);
                unpack16<T>(vc[p], c);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = fmaf(c0[e], g[e], fmaf(c1[e], c[e], c2[e]));  // = msfwsi_bn_bwd_apply
                t = pack16<T>(g);
                if (aout_wg != nullptr && row < rows_left)
                    *reinterpret_cast<uint4*>(aout_wg + (unsigned)(row * ROWB + cc * 16)) = t;
            }
            *reinterpret_cast<uint4*>(panel + row * ROWB + ((cc ^ panel_swz<K>(row)) << 4)) = t;
        }
    }
    // weight fragments of this wave's first block (after the staging loads have left their registers: with both live
    // the 256-channel input-gradient instances spilled)
    const T* __restrict__ wpk = reinterpret_cast<const T*>(prm.wpk);
    frag_t wf[KS];
    {
        const int cb0 = wave < nblk ? wave : nblk - 1;
        const T* wb = wpk + ((long)cb0 * KS * 64 + lane) * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[ks] = *reinterpret_cast<const frag_t*>(wb + ks * 512);
    }
    __syncthreads();  // the only workgroup barrier: from here on every wave runs alone

    // per-lane constants of the fragment reads: row tm*32 + l31, chunk (2 ks + lh) ^ swz(row) = (2 ks) ^ (lh ^ swz)
    const int xv = lh ^ panel_swz<K>(l31);
    const char* prow = panel + l31 * ROWB;
    // per-lane constants of the transposed epilogue: chunk q = lane & 3 of rows (lane >> 2) and 16 + (lane >> 2).
    // Every global access of the loop is "wave-uniform 64-bit base + 32-bit lane offset" (saddr form: one VGPR of address
    // for all row tiles instead of a 64-bit pair each -- with per-lane 64-bit addresses the 256-channel instances spilled)
    const int q = lane & 3, r4 = lane >> 2;
    const int nbyte = prm.Nout >> 3;  // gate bytes per row
    const int PQ = prm.P * prm.Q;
    const char* eop_wg = reinterpret_cast<const char*>(FWD ? prm.ident : prm.resid);
    const bool has_eop = eop_wg != nullptr;
    if (!LORES && has_eop) eop_wg += m0 * prm.Nout * 2;
    char* out_wg = reinterpret_cast<char*>(prm.out) + m0 * prm.Nout * 2;
    const unsigned char* mb_wg = (!FWD && prm.mask_bits != nullptr) ? prm.mask_bits + m0 * nbyte : nullptr;
    unsigned char* go_wg = (FWD && prm.gate_out != nullptr) ? prm.gate_out + m0 * nbyte : nullptr;
    const unsigned row_off = (unsigned)r4 * (unsigned)prm.Nout * 2u + (unsigned)q * 16u;  // byte offset of (row r4, chunk q)
    const unsigned bit_off = (unsigned)r4 * (unsigned)nbyte;

    for (int cb = wave; cb < nblk; cb += NW) {
        const int ncol = cb * 32 + q * 8;
        const unsigned lane_off = row_off + (unsigned)cb * 64u;
        // ---- epilogue operands of this block: in flight during its MFMAs ----
        uint4 er[TM * 2];
        unsigned ebits[2] = {0xffffffffu, 0xffffffffu};  // gate bytes of the TM*2 row groups, packed four to a register
        unsigned ehave = 0;                              // LORES: bit t = row group t has a residual
#pragma unroll
        for (int t = 0; t < TM * 2; ++t) {
            const int rbase = (t >> 1) * 32 + (t & 1) * 16;  // first row of the group (wave-uniform)
            const bool ok = rbase + r4 < rows_left;          // (a lane past the end re-reads row 0 of the panel)
            er[t] = make_uint4(0, 0, 0, 0);
            if (has_eop) {
                if constexpr (LORES) {
                    const unsigned m = (unsigned)(m0 + (ok ? rbase + r4 : 0));
                    const unsigned n = fast_div(m, prm.div_pq);
                    const unsigned rem = m - n * (unsigned)PQ;
                    const unsigned h = fast_div(rem, prm.div_q);
                    const unsigned w = rem - h * (unsigned)prm.Q;
                    const int Pl = (prm.P + 1) >> 1, Ql = (prm.Q + 1) >> 1;
                    // (odd pixels carry no residual: they read the value of the even pixel above / left and drop it)
                    // (32-bit byte offset from the tensor base: the entry point refuses low-resolution tensors of 4 GiB or more)
                    const unsigned lo = (((n * (unsigned)Pl + (h >> 1)) * (unsigned)Ql + (w >> 1)) * (unsigned)prm.Nout + (unsigned)ncol) * 2u;
                    er[t] = *reinterpret_cast<const uint4*>(eop_wg + lo);
                    if (((h | w) & 1u) == 0) ehave |= 1u << t;
                } else {
                    const unsigned off = ok ? (unsigned)rbase * (unsigned)prm.Nout * 2u + lane_off : (unsigned)cb * 64u + (unsigned)q * 16u;
                    er[t] = *reinterpret_cast<const uint4*>(eop_wg + off);
                }
            }
            if constexpr (!FWD) {
                if (mb_wg != nullptr) {  // the four lanes of a row read the same dword: one request
                    const unsigned boff = (ok ? (unsigned)rbase * (unsigned)nbyte + bit_off : 0u) + (unsigned)cb * 4u;
                    const unsigned dw = *reinterpret_cast<const unsigned*>(mb_wg + boff);
                    const unsigned b = (dw >> (8 * q)) & 0xffu;
                    ebits[t >> 2] = (ebits[t >> 2] & ~(0xffu << (8 * (t & 3)))) | (b << (8 * (t & 3)));
                }
            }
        }
        float psc[FWD ? 8 : 1], psh[FWD ? 8 : 1];
        if constexpr (FWD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                psc[e] = prm.post_scale[ncol + e];
                psh[e] = prm.post_shift[ncol + e];
            }
        }

        // ---- MFMAs of this block; the fragment just consumed is replaced by the next block's ----
        f32x16 acc[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[tm][j] = 0.f;
        const int cbn = cb + NW < nblk ? cb + NW : cb;  // (the last block re-requests its own fragments: no branch)
        const T* wn = wpk + (long)cbn * KS * 512;
        // activation fragments of step ks+1 are read while the MFMAs of step ks run (hipcc left to itself put each
        // ds_read_b128 right in front of the MFMA that consumes it, with a full lgkmcnt(0) wait between them)
        frag_t xc[TM], xn[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) xc[tm] = *reinterpret_cast<const frag_t*>(prow + tm * 32 * ROWB + ((0 ^ xv) << 4));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    xn[tm] = *reinterpret_cast<const frag_t*>(prow + tm * 32 * ROWB + (((2 * (ks + 1)) ^ xv) << 4));
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) mma32<T>(acc[tm], wf[ks], xc[tm]);
            wf[ks] = *reinterpret_cast<const frag_t*>(wn + ks * 512 + lane * 8);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA of step ks ...
                if (ks + 1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // ... then one fragment read of step ks+1
            }
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // the next block's weight fragment for this step
            __builtin_amdgcn_sched_barrier(0);  // nothing moves across k steps (hipcc otherwise interchanges the loops:
                                                // all k steps of one row tile, each read right before its MFMA)
            if (ks + 1 < KS) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) xc[tm] = xn[tm];
            }
        }

        // ---- epilogue: transpose each 32 x 32 tile through the wave's scratch, then 16-byte row chunks ----
        float ssum[FWD ? 1 : 8];
        if constexpr (!FWD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ssum[e] = 0.f;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            // lane (l31, lh) holds pixel l31, channels 8g + 4 lh + e in accumulator register 4g + e
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<uint2*>(scratch + l31 * SCR_PITCH + (8 * g + 4 * lh) * 2) =
                    pack4<T>(acc[tm][4 * g], acc[tm][4 * g + 1], acc[tm][4 * g + 2], acc[tm][4 * g + 3]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int t = tm * 2 + i;
                const int rbase = tm * 32 + i * 16;
                const uint4 cv = *reinterpret_cast<const uint4*>(scratch + (i * 16 + r4) * SCR_PITCH + q * 16);
                const bool ok = rbase + r4 < rows_left;
                float f[8];
                unpack16<T>(cv, f);
                if constexpr (FWD) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], psc[e], psh[e]);
                    if (has_eop) {
                        float id[8];
                        unpack16<T>(er[t], id);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += id[e];
                    }
                    if (prm.post_relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
                    }
                    if (go_wg != nullptr) {
                        const unsigned gb = gate_bits_of<T>(pack16<T>(f));  // (the pack is shared with the store below)
                        // the four lanes of a row hold four consecutive gate bytes: one dword store by the first of them
                        unsigned dw = gb << (8 * q);
                        dw |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)dw, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
                        dw |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)dw, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
                        if (ok && q == 0) *reinterpret_cast<unsigned*>(go_wg + ((unsigned)rbase * (unsigned)nbyte + bit_off + (unsigned)cb * 4u)) = dw;
                    }
                } else {
                    const bool addr = LORES ? ((ehave >> t) & 1u) != 0 : has_eop;
                    if (addr) {
                        float rs[8];
                        unpack16<T>(er[t], rs);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += rs[e];
                    }
                    if (prm.gapg != nullptr) {
                        const long m = m0 + (ok ? rbase + r4 : 0);
                        const long img = LORES ? (long)fast_div((unsigned)m, prm.div_pq) : m / PQ;
                        float gp[8];
                        unpack16<T>(*reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(prm.gapg) + img * prm.Nout + ncol), gp);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = fmaf(gp[e], prm.gap_scale, f[e]);
                    }
                    const unsigned b = (ebits[t >> 2] >> (8 * (t & 3))) & 0xffu;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (!((b >> e) & 1u)) f[e] = 0.f;
                        if (ok) ssum[e] += f[e];
                    }
                }
                if (ok) *reinterpret_cast<uint4*>(out_wg + ((unsigned)rbase * (unsigned)prm.Nout * 2u + lane_off)) = pack16<T>(f);
            }
        }
        if constexpr (!FWD) {
            if (prm.sums != nullptr) {
                // lanes with equal q hold the same 8 channels for different rows
#pragma unroll
                for (int e = 0; e < 8; ++e) {
#pragma unroll
                    for (int off = 4; off < 64; off <<= 1) ssum[e] += __shfl_xor(ssum[e], off, 64);
                }
                if (lane < 4) {
                    double* dst = prm.sums + (long)(blockIdx.x % prm.nshard) * 2 * prm.Nout + ncol;
#pragma unroll
                    for (int e = 0; e < 8; ++e) atomicAdd(dst + e, (double)ssum[e]);
                }
            }
        }
    }
}

// W(n, k) = w[n * stride_n + k * stride_k]  ->  wpk[n/32][k/16][lane][j] = W(32 (n/32) + (lane & 31), 16 (k/16) + 8 (lane >> 5) + j):
// the A operand of mfma_f32_32x32x16 for output-channel block n/32 and k step k/16, one contiguous KiB per fragment
template <typename T>
__global__ void panel_pack_kernel(const T* __restrict__ w, T* __restrict__ wpk, int Nout, int K, long stride_n, long stride_k) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per packed element
    if (i >= (long)Nout * K) return;
    const int j = (int)(i & 7);
    const int lane = (int)((i >> 3) & 63);
    const long frag = i >> 9;
    const int KS = K >> 4;
    const int ks = (int)(frag % KS);
    const int nb = (int)(frag / KS);
    const int n = nb * 32 + (lane & 31);
    const int k = ks * 16 + 8 * (lane >> 5) + j;
    wpk[i] = w[n * stride_n + k * stride_k];
}

template <typename T, int K, int BM, int PRO, int EPI>
int launch_panel(const PanelParams& prm, hipStream_t stream) {
    constexpr int LDS = BM * K * 2 + 4 * 32 * 80;
    void (*kern)(const PanelParams) = panel_kernel<T, K, BM, PRO, EPI>;
    if (LDS > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
    }
    const long nwg = ((long)prm.M + BM - 1) / BM;
    if (nwg <= 0 || nwg > 0x7fffffffL) return MSFWSI_EINVAL;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), LDS, stream, prm);
    return msfwsi_launch_status();
}

template <typename T, int PRO, int EPI>
int dispatch_panel_k(int K, const PanelParams& prm, hipStream_t stream) {
    switch (K) {
        case 64: return launch_panel<T, 64, 128, PRO, EPI>(prm, stream);
        case 128: return launch_panel<T, 128, 128, PRO, EPI>(prm, stream);
        case 256: return launch_panel<T, 256, 128, PRO, EPI>(prm, stream);
        case 512: return launch_panel<T, 512, 64, PRO, EPI>(prm, stream);
    }
    return MSFWSI_EUNSUPPORTED;
}

bool panel_shape_ok(int dtype, int K, int Nout, long M) {
    return (dtype == MSFWSI_DT_BF16 || dtype == MSFWSI_DT_F16) && (K == 64 || K == 128 || K == 256 || K == 512) &&
           Nout >= 128 && Nout % 32 == 0 && M > 0 && M <= 0x7fffffffL;
}

bool is_1x1(const msfwsi_conv_desc* d) {
    return d != nullptr && d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->P == d->H && d->Q == d->W;
}

}  // namespace

extern "C" int msfwsi_panel_supported(const msfwsi_conv_desc* d, int dgrad) {
    if (!is_1x1(d)) return 0;
    const long M = (long)d->N * d->H * d->W;
    return panel_shape_ok(d->dtype, dgrad ? d->K : d->C, dgrad ? d->C : d->K, M) ? 1 : 0;
}

extern "C" int msfwsi_panel_pack_weights(int dtype, const void* w, void* wpk, int Nout, int K, long stride_n, long stride_k,
                                         void* stream) {
    MSFWSI_CHECK_ARG(w != nullptr && wpk != nullptr && Nout > 0 && K > 0);
    if ((dtype != MSFWSI_DT_BF16 && dtype != MSFWSI_DT_F16) || Nout % 32 != 0 || K % 16 != 0) return MSFWSI_EUNSUPPORTED;
    const long n = (long)Nout * K;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // (bf16 and fp16 are both 2-byte payloads: the permutation does not look inside an element)
    hipLaunchKernelGGL(panel_pack_kernel<unsigned short>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const unsigned short*>(w), reinterpret_cast<unsigned short*>(wpk), Nout, K, stride_n,
                       stride_k);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_panel_fwd_post(const msfwsi_conv_desc* d, const void* x, const float* pro_scale, const float* pro_shift,
                                     const void* wpk, void* y, const float* post_scale, const float* post_shift,
                                     const void* ident, int relu, unsigned char* gate_out, void* stream) {
    if (d == nullptr) return MSFWSI_EINVAL;
    if (!msfwsi_panel_supported(d, 0)) return MSFWSI_EUNSUPPORTED;
    MSFWSI_CHECK_ARG(x != nullptr && wpk != nullptr && y != nullptr && post_scale != nullptr && post_shift != nullptr);
    MSFWSI_CHECK_ARG((pro_scale == nullptr) == (pro_shift == nullptr));
    PanelParams prm{};
    prm.src = x; prm.p0 = pro_scale; prm.p1 = pro_shift;
    prm.wpk = wpk; prm.out = y;
    prm.post_scale = post_scale; prm.post_shift = post_shift; prm.ident = ident; prm.post_relu = relu ? 1 : 0;
    prm.gate_out = gate_out;
    prm.nshard = 1;
    prm.M = d->N * d->H * d->W; prm.Nout = d->K; prm.P = d->H; prm.Q = d->W;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool pro = pro_scale != nullptr;
    if (d->dtype == MSFWSI_DT_BF16)
        return pro ? dispatch_panel_k<__bf16, 1, 1>(d->C, prm, st) : dispatch_panel_k<__bf16, 0, 1>(d->C, prm, st);
    return pro ? dispatch_panel_k<_Float16, 1, 1>(d->C, prm, st) : dispatch_panel_k<_Float16, 0, 1>(d->C, prm, st);
}

extern "C" int msfwsi_panel_dgrad(const msfwsi_conv_desc* d, const void* dy, const void* c, const float* k1, const float* k2,
                                  const float* k3, void* dc_out, const void* wpk, void* dx, const void* resid,
                                  int resid_stride, const void* gapg, float gap_scale, const unsigned char* mask_bits,
                                  double* sums, int nshard, void* stream) {
    if (d == nullptr) return MSFWSI_EINVAL;
    if (!msfwsi_panel_supported(d, 1)) return MSFWSI_EUNSUPPORTED;
    MSFWSI_CHECK_ARG(dy != nullptr && wpk != nullptr && dx != nullptr);
    const bool pro = c != nullptr;
    MSFWSI_CHECK_ARG(pro == (k1 != nullptr) && pro == (k2 != nullptr) && pro == (k3 != nullptr));
    MSFWSI_CHECK_ARG(pro || dc_out == nullptr);
    MSFWSI_CHECK_ARG(resid_stride >= 0 && (resid_stride <= 1 || resid != nullptr));
    if (resid_stride > 2) return MSFWSI_EUNSUPPORTED;
    if (resid_stride == 2 && (long)d->N * ((d->H + 1) / 2) * ((d->W + 1) / 2) * d->C * 2 >= (1L << 32)) return MSFWSI_EUNSUPPORTED;
    MSFWSI_CHECK_ARG((mask_bits == nullptr) == (sums == nullptr) && (sums == nullptr || nshard >= 1));
    PanelParams prm{};
    prm.src = dy; prm.src_c = c; prm.p0 = k1; prm.p1 = k2; prm.p2 = k3; prm.aout = dc_out;
    prm.wpk = wpk; prm.out = dx;
    prm.resid = resid; prm.gapg = gapg; prm.gap_scale = gap_scale;
    prm.mask_bits = mask_bits; prm.sums = sums; prm.nshard = nshard > 0 ? nshard : 1;
    prm.M = d->N * d->H * d->W; prm.Nout = d->C; prm.P = d->H; prm.Q = d->W;
    prm.div_pq = make_fastdiv((unsigned)(d->H * d->W));
    prm.div_q = make_fastdiv((unsigned)d->W);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool lores = resid_stride == 2;
    if (d->dtype == MSFWSI_DT_BF16) {
        if (lores) return pro ? dispatch_panel_k<__bf16, 2, 3>(d->K, prm, st) : dispatch_panel_k<__bf16, 0, 3>(d->K, prm, st);
        return pro ? dispatch_panel_k<__bf16, 2, 0>(d->K, prm, st) : dispatch_panel_k<__bf16, 0, 0>(d->K, prm, st);
    }
    if (lores) return pro ? dispatch_panel_k<_Float16, 2, 3>(d->K, prm, st) : dispatch_panel_k<_Float16, 0, 3>(d->K, prm, st);
    return pro ? dispatch_panel_k<_Float16, 2, 0>(d->K, prm, st) : dispatch_panel_k<_Float16, 0, 0>(d->K, prm, st);
}
