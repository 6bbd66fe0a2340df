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
#include "handload.h"
#include "../../include/msfwsi_hip.h"

#ifndef MSFWSI_PANEL_ABLATE
#define MSFWSI_PANEL_ABLATE 0  // diagnostic builds (tools/build_variant.sh), WRONG RESULTS: bit 0 = no output / gate stores, bit 1 = no
#endif                         // MFMAs, bit 2 = no weight-fragment reloads, bit 3 = no epilogue-operand loads, bit 4 = no staging loads,
                               // bit 5 = no epilogue arithmetic, bit 6 = no gate bits, bit 7 = epilogue loads / stores as 8 rows x 128 B per
                               // instruction (same bytes, half the L2 requests; wrong addresses)

#ifndef MSFWSI_PANEL_NT
#define MSFWSI_PANEL_NT 0  // 1: output / gate stores non-temporal (A/B: tools/build_variant.sh nt "-DMSFWSI_PANEL_NT=1")
#endif

namespace {

template <typename V>
__device__ __forceinline__ void pl_store(V* p, const V& v) {
#if MSFWSI_PANEL_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

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
    unsigned char* gate_out;  // gate bytes of the [M][Nout/8] chunks (layout: gate_off, common.h), nullable
    // EPI 0 / 3 (input gradient): out = gate( round(acc) + resid + gap_scale * gapg[image] ), sums[shard][0][n] += out
    const void* resid;  // [M][Nout] (EPI 3: [N][P/2][Q/2][Nout], added where h % 2 == w % 2 == 0), nullable
    const void* gapg;   // [N][Nout], nullable
    float gap_scale;
    const unsigned char* mask_bits;  // same layout, nullable
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

// The block loop of one wave.  Vector-memory operations per block, in program order (HAND mode: all unconditional):
//   k loop     : KS weight-fragment loads (the NEXT block's)
//   epilogue   : per row group t -- [gate-byte load], operand load (both the next block's), [gate store], output store;
//                then (forward) four loads of the next block's BatchNorm scale / shift chunks
// so every load is followed by exactly KS - 1 + NL + NS younger operations before its first consumer (NL loads, NS
// stores per epilogue): ONE constant serves every hand-counted wait.
template <typename T, int K, int BM, int EPI, bool HAND>
__device__ __forceinline__ void panel_blocks(const PanelParams& prm, char* panel, char* scratch, float* colsum,
                                             const char* gapl, long m0, int rows_left, int wave, int lane) {
    constexpr int NW = 4;
    constexpr int TM = BM / 32, NG = TM * 2, KS = K / 16, ROWB = K * 2;
    constexpr int SCR_PITCH = 80;
    constexpr bool FWD = EPI == 1, LORES = EPI == 3;
    typedef typename MmaFrag<T>::type frag_t;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nblk = prm.Nout >> 5;
    // per-lane constants of the transposed epilogue: chunk q = lane & 3 of rows (lane >> 2) and 16 + (lane >> 2).
    // Every global access of the loop is "wave-uniform 64-bit base + 32-bit lane offset" (saddr form: one VGPR of address
    // for all row tiles instead of a 64-bit pair each -- with per-lane 64-bit addresses the 256-channel instances spilled)
    const int q = lane & 3, r4 = lane >> 2;
    const int PQ = prm.P * prm.Q;
    const unsigned img0 = FWD ? 0u : fast_div((unsigned)m0, prm.div_pq);  // first image of the panel (gapl's first row)
    const char* eop_wg = reinterpret_cast<const char*>(FWD ? prm.ident : prm.resid);
    const bool has_eop = eop_wg != nullptr;
    if (!LORES && has_eop) eop_wg += m0 * prm.Nout * 2;
    char* out_wg = reinterpret_cast<char*>(prm.out) + m0 * prm.Nout * 2;
    // gate bytes (blocked layout, gate_off in common.h): the dwords of this panel's rows for block cb are contiguous
    const long gbase = ((m0 >> 7) * (long)nblk) * 512 + (m0 & 127) * 4;
    const unsigned char* mb_wg = (!FWD && prm.mask_bits != nullptr) ? prm.mask_bits + gbase : nullptr;
    unsigned char* go_wg = (FWD && prm.gate_out != nullptr) ? prm.gate_out + gbase : nullptr;
#if MSFWSI_PANEL_ABLATE & 128
    const unsigned row_off = (unsigned)(lane >> 3) * (unsigned)prm.Nout * 2u + (unsigned)(lane & 7) * 16u;
#else
    const unsigned row_off = (unsigned)r4 * (unsigned)prm.Nout * 2u + (unsigned)q * 16u;  // byte offset of (row r4, chunk q)
#endif
    const unsigned bit_off = (unsigned)r4 * 4u;  // this lane's row inside a 16-row group of gate dwords
    // HAND mode is entered only with an operand tensor / gate bytes present wherever the epilogue class has them: its
    // operation counts are compile-time constants
    constexpr int NL = NG * (FWD ? 1 : 2) + (FWD ? 4 : 0);  // loads per epilogue
    constexpr int NS = NG * (FWD ? 2 : 1);                  // stores per epilogue (forward: gate dword + output)
    constexpr int NWAIT = KS - 1 + NL + NS;
    static_assert(NWAIT <= 63, "operation count per block exceeds the vmcnt range");

    // The epilogue operands (identity / residual chunk, gate bytes) of row group t of output block cb.  Loads are
    // unconditional: a lane past the tensor's end re-reads row 0 of the panel, and in LORES mode an odd pixel reads the
    // value of its even neighbour and drops it (a divergent branch around a load makes hipcc drain the whole queue).
    u32x4 er[NG];
    unsigned ebw[FWD ? 1 : NG];  // the dword holding this row's four gate bytes of the block
    u32x4 pq[FWD ? 4 : 1];       // forward: scale[0:4], scale[4:8], shift[0:4], shift[4:8] of this lane's 8 output channels
    unsigned ehave = 0;          // LORES: bit t = row group t carries a residual (the same for every block)
    auto request = [&](int t, int cb) __attribute__((always_inline)) {
        const int rbase = (t >> 1) * 32 + (t & 1) * 16;  // first row of the group (wave-uniform)
        const bool ok = HAND || rbase + r4 < rows_left;
        if (MSFWSI_PANEL_ABLATE & 8) return;
        if constexpr (!FWD) {
            if (HAND || mb_wg != nullptr) {  // the four lanes of a row read the same dword: one request
                const unsigned boff = (ok ? (unsigned)rbase * 4u + bit_off : 0u) + (unsigned)cb * 512u;
                pl_load4<HAND>(ebw[t], mb_wg, boff);
            }
        }
        if (HAND || has_eop) {
            if constexpr (LORES) {
                const unsigned m = (unsigned)(m0 + (ok ? rbase + r4 : 0));
                const unsigned n = fast_div(m, prm.div_pq);
                const unsigned rem = m - n * (unsigned)PQ;
                const unsigned h = fast_div(rem, prm.div_q);
                const unsigned w = rem - h * (unsigned)prm.Q;
                const int Pl = (prm.P + 1) >> 1, Ql = (prm.Q + 1) >> 1;
                // (32-bit byte offset from the tensor base: the entry point refuses low-resolution tensors of 4 GiB or more)
                const unsigned lo = (((n * (unsigned)Pl + (h >> 1)) * (unsigned)Ql + (w >> 1)) * (unsigned)prm.Nout +
                                     (unsigned)(cb * 32 + q * 8)) * 2u;
                pl_load16<HAND>(er[t], eop_wg, lo);
                if (((h | w) & 1u) == 0) ehave |= 1u << t;
            } else {
                const unsigned off = (ok ? (unsigned)(MSFWSI_PANEL_ABLATE & 128 ? rbase / 2 + (cb & 1) * 64 : rbase) * (unsigned)prm.Nout * 2u + row_off : (unsigned)q * 16u) +
                                     (unsigned)(MSFWSI_PANEL_ABLATE & 128 ? (cb & ~1) : cb) * 64u;
                pl_load16<HAND>(er[t], eop_wg, off);
            }
        }
    };
    auto request_post = [&](int cb) __attribute__((always_inline)) {
        if constexpr (FWD) {
            const unsigned o = (unsigned)(cb * 32 + q * 8) * 4u;
            pl_load16<HAND>(pq[0], prm.post_scale, o);
            pl_load16<HAND>(pq[1], prm.post_scale, o + 16u);
            pl_load16<HAND>(pq[2], prm.post_shift, o);
            pl_load16<HAND>(pq[3], prm.post_shift, o + 16u);
        }
    };
#pragma unroll
    for (int t = 0; t < NG; ++t) {
        er[t] = (u32x4){0u, 0u, 0u, 0u};
        if constexpr (!FWD) ebw[t] = 0xffffffffu;
    }
    const int cb0 = wave < nblk ? wave : nblk - 1;  // (a wave without a block -- fewer than 4 blocks -- loads and drops)
#pragma unroll
    for (int t = 0; t < NG; ++t) request(t, cb0);
    request_post(cb0);
    // weight fragments of this wave's first block
    const char* wpk = reinterpret_cast<const char*>(prm.wpk);
    u32x4 wf[KS];
    {
        const char* wb = wpk + (long)cb0 * KS * 1024;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) pl_load16<HAND>(wf[ks], wb, (unsigned)(ks * 1024 + lane * 16));
    }
    if constexpr (HAND) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // once per workgroup; the loop's waits are counted

    // per-lane constants of the fragment reads: row tm*32 + l31, chunk (2 ks + lh) ^ swz(row) = (2 ks) ^ (lh ^ swz)
    const int xv = lh ^ panel_swz<K>(l31);
    const char* prow = panel + l31 * ROWB;

    for (int cb = wave; cb < nblk; cb += NW) {
        const int ncol = cb * 32 + q * 8;
        const unsigned lane_off = row_off + (unsigned)cb * 64u;
        const int cbn = cb + NW < nblk ? cb + NW : cb;  // (the last block re-requests its own operands: no branch)

        // ---- MFMAs of this block; the weight fragment just consumed is replaced by the next block's, the activation
        //      fragment just consumed by the next k step's (one register set: a read is four MFMAs ahead of its use) ----
        f32x16 acc[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[tm][j] = 0.f;
        const char* wn = wpk + (long)cbn * KS * 1024;
        frag_t xc[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) xc[tm] = *reinterpret_cast<const frag_t*>(prow + tm * 32 * ROWB + ((0 ^ xv) << 4));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            pl_wait<HAND, NWAIT>(wf[ks]);
            const frag_t wfr = __builtin_bit_cast(frag_t, wf[ks]);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                if (!(MSFWSI_PANEL_ABLATE & 2) || ks == 0) mma32<T>(acc[tm], wfr, xc[tm]);
                if (ks + 1 < KS)
                    xc[tm] = *reinterpret_cast<const frag_t*>(prow + tm * 32 * ROWB + (((2 * (ks + 1)) ^ xv) << 4));
            }
            if (!(MSFWSI_PANEL_ABLATE & 4)) pl_load16<HAND>(wf[ks], wn, (unsigned)(ks * 1024 + lane * 16));
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA of step ks ...
                if (ks + 1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // ... then one fragment read of step ks+1
            }
            __builtin_amdgcn_sched_barrier(0);  // nothing moves across k steps (hipcc otherwise interchanges the loops:
                                                // all k steps of one row tile, each read right before its MFMA)
        }

        // ---- epilogue: transpose each 32 x 32 tile through the wave's scratch, then 16-byte row chunks; each row group's
        //      operand registers are re-requested for the NEXT block as soon as they are consumed ----
        if constexpr (FWD) {
            // the four post loads close the previous epilogue: only the k loop's KS weight loads are younger
            pl_wait<HAND, KS>(pq[0]);
            pl_wait<HAND, KS>(pq[1]);
            pl_wait<HAND, KS>(pq[2]);
            pl_wait<HAND, KS>(pq[3]);
        }
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
                const bool ok = HAND || rbase + r4 < rows_left;
                float f[8];
                unpack16<T>(cv, f);
                uint4 pk;
                // this group's operands were requested NWAIT operations ago (one block's worth, as the weights)
                // (input gradient: the gate-byte load of the group is issued BEFORE its operand load, so one operation fewer
                //  is younger than the operand -- and the wait for the operand covers the gate bytes)
                if constexpr (FWD) pl_wait<HAND, NWAIT>(er[t]);
                else pl_wait<HAND, NWAIT - 1>(er[t], ebw[t]);
                if constexpr (MSFWSI_PANEL_ABLATE & 32) {
                    pk = cv;
                } else if constexpr (FWD) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        f[e] = fmaf(f[e], __uint_as_float(pq[e >> 2][e & 3]), __uint_as_float(pq[2 + (e >> 2)][e & 3]));
                    if (HAND || has_eop) {
                        float id[8];
                        unpack16<T>(as_uint4(er[t]), id);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += id[e];
                    }
                    if (prm.post_relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
                    }
                    pk = pack16<T>(f);
                } else {
                    const bool addr = LORES ? ((ehave >> t) & 1u) != 0 : (HAND || has_eop);
                    if (addr) {
                        float rs[8];
                        unpack16<T>(as_uint4(er[t]), rs);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += rs[e];
                    }
                    if (prm.gapg != nullptr) {
                        // pooled-feature gradient of the row's image.  HAND: from the two rows panel_kernel staged in LDS (a
                        // whole panel spans at most two images there: launch_panel) -- no vector-memory operation, the
                        // hand-counted waits stay exact; otherwise straight from memory on hipcc's own waits
                        const unsigned m = (unsigned)(m0 + (ok ? rbase + r4 : 0));
                        const unsigned img = fast_div(m, prm.div_pq);
                        uint4 gv;
                        if constexpr (HAND) gv = *reinterpret_cast<const uint4*>(gapl + ((img - img0) * (unsigned)prm.Nout + (unsigned)ncol) * 2u);
                        else gv = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(prm.gapg) + (long)img * prm.Nout + ncol);
                        float gp[8];
                        unpack16<T>(gv, gp);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = fmaf(gp[e], prm.gap_scale, f[e]);
                    }
                    const unsigned b = (ebw[t] >> (8 * q)) & 0xffu;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (!((b >> e) & 1u)) f[e] = 0.f;
                        if (ok) ssum[e] += f[e];
                    }
                    pk = pack16<T>(f);
                }
                request(t, cbn);  // this row group's operands of the wave's next block (registers just consumed)
                if constexpr (FWD) {
                    if ((HAND || go_wg != nullptr) && !(MSFWSI_PANEL_ABLATE & 64)) {
                        const unsigned gb = gate_bits_of<T>(pk);
                        // the four lanes of a row hold four consecutive gate bytes: every one of them stores the same dword
                        // (HAND mode counts its stores: no lane-dependent branch around one)
                        unsigned dw = gb << (8 * q);
                        dw |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)dw, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
                        dw |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)dw, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
                        if ((HAND || (ok && q == 0)) && !(MSFWSI_PANEL_ABLATE & 1))
                            pl_store(reinterpret_cast<unsigned*>(go_wg + ((unsigned)rbase * 4u + bit_off + (unsigned)cb * 512u)), dw);
                    }
                }
                if (ok && !(MSFWSI_PANEL_ABLATE & 1))
                    pl_store(reinterpret_cast<u32x4*>(out_wg + ((unsigned)(MSFWSI_PANEL_ABLATE & 128 ? rbase / 2 + (cb & 1) * 64 : rbase) * (unsigned)prm.Nout * 2u +
                                                        (MSFWSI_PANEL_ABLATE & 128 ? row_off + (unsigned)(cb & ~1) * 64u : lane_off))),
                             (u32x4){pk.x, pk.y, pk.z, pk.w});
            }
        }
        request_post(cbn);
        if constexpr (!FWD) {
            if (prm.sums != nullptr) {
                // lanes with equal q hold the same 8 channels for different rows; every channel of the panel is owned by
                // exactly one wave, so the column sums land in LDS by plain stores
#pragma unroll
                for (int e = 0; e < 8; ++e) {
#pragma unroll
                    for (int off = 4; off < 64; off <<= 1) ssum[e] += __shfl_xor(ssum[e], off, 64);
                }
                if (lane < 4) {
#pragma unroll
                    for (int e = 0; e < 8; e += 4)
                        *reinterpret_cast<float4*>(colsum + ncol + e) = make_float4(ssum[e], ssum[e + 1], ssum[e + 2], ssum[e + 3]);
                }
            }
        }
    }
    // the last block's re-requested operands: nobody reads them, but they are still in flight -- every destination register is
    // named by a draining wait before hipcc may re-use it (tools/check_hand_waits.py checks the code after the loop too)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) pl_drain<HAND>(wf[ks]);
#pragma unroll
    for (int t = 0; t < NG; ++t) {
        pl_drain<HAND>(er[t]);
        if constexpr (!FWD) pl_drain<HAND>(ebw[t]);
    }
    if constexpr (FWD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pl_drain<HAND>(pq[i]);
    }
}

// The WIDE form of the block loop (k <= 128, Nout a multiple of 64).  The launches of this kernel are bound by the L2 REQUEST
// rate, and a wave that owns 32 output channels moves its epilogue's operands and results as 16 rows x 64 bytes per
// instruction -- twice the requests of full 128-byte lines (an ablation with 8 rows x 128 bytes: -10 ... -15 %,
// profiles/r05_panel_ablation_128B_rows.txt).  Here a wave owns 64 CHANNELS of 64 ROWS: waves (rh, cg) = (wave & 1, wave >> 1)
// take the row half rh of the panel and the wide blocks cg, cg + 2, ... -- two weight fragments per k step (each used by two
// row tiles instead of four: the weight stream from L2 doubles, 25 M requests more per launch against 100 M fewer in the
// epilogue), the same 64 accumulator registers, one activation fragment read per two MFMAs instead of one.  Per wide block
// the operation counts of panel_blocks hold with 2 KS weight loads: the A fragment of step ks is followed by
// 2 KS - 1 + NL + NS younger operations, the B fragment by one fewer.
template <typename T, int K, int BM, int EPI, bool HAND>
__device__ __forceinline__ void panel_blocks_wide(const PanelParams& prm, char* panel, char* scratch, float* colsum,
                                                  const char* gapl, long m0, int rows_left, int wave, int lane) {
    constexpr int TMW = BM / 64, NG = TMW * 4, KS = K / 16, ROWB = K * 2;
    constexpr int SCRW = 144;  // scratch pitch: 32 rows x (64 channels = 128 bytes + 16)
    constexpr bool FWD = EPI == 1, LORES = EPI == 3;
    static_assert(BM == 128 && K <= 128, "wide blocks: 128-row panels, k <= 128 (two weight fragments per k step in registers)");
    typedef typename MmaFrag<T>::type frag_t;
    const int l31 = lane & 31, lh = lane >> 5;
    const int rh = wave & 1, cg = wave >> 1;
    const int nwb = prm.Nout >> 6;
    const int c8 = lane & 7, r8 = lane >> 3;  // chunk of the 128-byte row segment, row inside a group of 8
    const int PQ = prm.P * prm.Q;
    const unsigned img0 = FWD ? 0u : fast_div((unsigned)m0, prm.div_pq);
    const char* eop_wg = reinterpret_cast<const char*>(FWD ? prm.ident : prm.resid);
    const bool has_eop = eop_wg != nullptr;
    if (!LORES && has_eop) eop_wg += m0 * prm.Nout * 2;
    char* out_wg = reinterpret_cast<char*>(prm.out) + m0 * prm.Nout * 2;
    const long gbase = ((m0 >> 7) * (long)(prm.Nout >> 5)) * 512 + (m0 & 127) * 4;
    const unsigned char* mb_wg = (!FWD && prm.mask_bits != nullptr) ? prm.mask_bits + gbase : nullptr;
    unsigned char* go_wg = (FWD && prm.gate_out != nullptr) ? prm.gate_out + gbase : nullptr;
    const unsigned row_off = (unsigned)r8 * (unsigned)prm.Nout * 2u + (unsigned)c8 * 16u;
    const unsigned bit_off = (unsigned)r8 * 4u + (unsigned)(c8 >> 2) * 512u;  // this lane's row and 32-channel block inside a wide block
    constexpr int NL = NG * (FWD ? 1 : 2) + (FWD ? 4 : 0);
    constexpr int NS = NG * (FWD ? 2 : 1);
    constexpr int NWAIT = 2 * KS - 1 + NL + NS;
    static_assert(NWAIT <= 63, "operation count per wide block exceeds the vmcnt range");

    u32x4 er[NG];
    unsigned ebw[FWD ? 1 : NG];
    u32x4 pq[FWD ? 4 : 1];
    unsigned ehave = 0;
    auto request = [&](int t, int wb) __attribute__((always_inline)) {
        const int rbase = rh * 64 + (t >> 2) * 32 + (t & 3) * 8;  // first row of the group (wave-uniform)
        const bool ok = HAND || rbase + r8 < rows_left;
        if constexpr (!FWD) {
            if (HAND || mb_wg != nullptr) {
                const unsigned boff = (ok ? (unsigned)rbase * 4u + bit_off : (unsigned)(c8 >> 2) * 512u) + (unsigned)wb * 1024u;
                pl_load4<HAND>(ebw[t], mb_wg, boff);
            }
        }
        if (HAND || has_eop) {
            if constexpr (LORES) {
                const unsigned m = (unsigned)(m0 + (ok ? rbase + r8 : 0));
                const unsigned n = fast_div(m, prm.div_pq);
                const unsigned rem = m - n * (unsigned)PQ;
                const unsigned h = fast_div(rem, prm.div_q);
                const unsigned w = rem - h * (unsigned)prm.Q;
                const int Pl = (prm.P + 1) >> 1, Ql = (prm.Q + 1) >> 1;
                const unsigned lo = (((n * (unsigned)Pl + (h >> 1)) * (unsigned)Ql + (w >> 1)) * (unsigned)prm.Nout +
                                     (unsigned)(wb * 64 + c8 * 8)) * 2u;
                pl_load16<HAND>(er[t], eop_wg, lo);
                if (((h | w) & 1u) == 0) ehave |= 1u << t;
            } else {
                const unsigned off = (ok ? (unsigned)rbase * (unsigned)prm.Nout * 2u + row_off : (unsigned)c8 * 16u) + (unsigned)wb * 128u;
                pl_load16<HAND>(er[t], eop_wg, off);
            }
        }
    };
    auto request_post = [&](int wb) __attribute__((always_inline)) {
        if constexpr (FWD) {
            const unsigned o = (unsigned)(wb * 64 + c8 * 8) * 4u;
            pl_load16<HAND>(pq[0], prm.post_scale, o);
            pl_load16<HAND>(pq[1], prm.post_scale, o + 16u);
            pl_load16<HAND>(pq[2], prm.post_shift, o);
            pl_load16<HAND>(pq[3], prm.post_shift, o + 16u);
        }
    };
#pragma unroll
    for (int t = 0; t < NG; ++t) {
        er[t] = (u32x4){0u, 0u, 0u, 0u};
        if constexpr (!FWD) ebw[t] = 0xffffffffu;
    }
    const int wb0 = cg < nwb ? cg : nwb - 1;  // (a wave pair without a wide block loads and drops)
#pragma unroll
    for (int t = 0; t < NG; ++t) request(t, wb0);
    request_post(wb0);
    const char* wpk = reinterpret_cast<const char*>(prm.wpk);
    u32x4 wfa[KS], wfb[KS];
    {
        const char* wb_ = wpk + (long)(2 * wb0) * KS * 1024;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            pl_load16<HAND>(wfa[ks], wb_, (unsigned)(ks * 1024 + lane * 16));
            pl_load16<HAND>(wfb[ks], wb_, (unsigned)((KS + ks) * 1024 + lane * 16));
        }
    }
    if constexpr (HAND) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const int xv = lh ^ panel_swz<K>(l31);
    const char* prow = panel + (rh * 64 + l31) * ROWB;

    for (int wb = cg; wb < nwb; wb += 2) {
        const int ncol = wb * 64 + c8 * 8;
        const unsigned lane_off = row_off + (unsigned)wb * 128u;
        const int wbn = wb + 2 < nwb ? wb + 2 : wb;  // (the last wide block re-requests its own operands: no branch)

        f32x16 acc[2][TMW];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[n][tm][j] = 0.f;
        const char* wn = wpk + (long)(2 * wbn) * KS * 1024;
        frag_t xc[TMW];
#pragma unroll
        for (int tm = 0; tm < TMW; ++tm) xc[tm] = *reinterpret_cast<const frag_t*>(prow + tm * 32 * ROWB + ((0 ^ xv) << 4));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            pl_wait<HAND, NWAIT>(wfa[ks]);
            pl_wait<HAND, NWAIT - 1>(wfb[ks]);
            const frag_t fa = __builtin_bit_cast(frag_t, wfa[ks]);
            const frag_t fb = __builtin_bit_cast(frag_t, wfb[ks]);
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm) {
                mma32<T>(acc[0][tm], fa, xc[tm]);
                mma32<T>(acc[1][tm], fb, xc[tm]);
                if (ks + 1 < KS)
                    xc[tm] = *reinterpret_cast<const frag_t*>(prow + tm * 32 * ROWB + (((2 * (ks + 1)) ^ xv) << 4));
            }
            pl_load16<HAND>(wfa[ks], wn, (unsigned)(ks * 1024 + lane * 16));
            pl_load16<HAND>(wfb[ks], wn, (unsigned)((KS + ks) * 1024 + lane * 16));
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (ks + 1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        if constexpr (FWD) {
            pl_wait<HAND, 2 * KS>(pq[0]);
            pl_wait<HAND, 2 * KS>(pq[1]);
            pl_wait<HAND, 2 * KS>(pq[2]);
            pl_wait<HAND, 2 * KS>(pq[3]);
        }
        float ssum[FWD ? 1 : 8];
        if constexpr (!FWD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ssum[e] = 0.f;
        }
#pragma unroll
        for (int tm = 0; tm < TMW; ++tm) {
            // both 32-channel tiles of the row tile side by side: lane (l31, lh) holds pixel l31, channels 8g + 4 lh + e
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<uint2*>(scratch + l31 * SCRW + n * 64 + (8 * g + 4 * lh) * 2) =
                        pack4<T>(acc[n][tm][4 * g], acc[n][tm][4 * g + 1], acc[n][tm][4 * g + 2], acc[n][tm][4 * g + 3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = tm * 4 + i;
                const int rbase = rh * 64 + tm * 32 + i * 8;
                const uint4 cv = *reinterpret_cast<const uint4*>(scratch + (i * 8 + r8) * SCRW + c8 * 16);
                const bool ok = HAND || rbase + r8 < rows_left;
                float f[8];
                unpack16<T>(cv, f);
                uint4 pk;
                if constexpr (FWD) pl_wait<HAND, NWAIT>(er[t]);
                else pl_wait<HAND, NWAIT - 1>(er[t], ebw[t]);
                if constexpr (FWD) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        f[e] = fmaf(f[e], __uint_as_float(pq[e >> 2][e & 3]), __uint_as_float(pq[2 + (e >> 2)][e & 3]));
                    if (HAND || has_eop) {
                        float id[8];
                        unpack16<T>(as_uint4(er[t]), id);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += id[e];
                    }
                    if (prm.post_relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
                    }
                    pk = pack16<T>(f);
                } else {
                    const bool addr = LORES ? ((ehave >> t) & 1u) != 0 : (HAND || has_eop);
                    if (addr) {
                        float rs[8];
                        unpack16<T>(as_uint4(er[t]), rs);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += rs[e];
                    }
                    if (prm.gapg != nullptr) {
                        const unsigned m = (unsigned)(m0 + (ok ? rbase + r8 : 0));
                        const unsigned img = fast_div(m, prm.div_pq);
                        uint4 gv;
                        if constexpr (HAND) gv = *reinterpret_cast<const uint4*>(gapl + ((img - img0) * (unsigned)prm.Nout + (unsigned)ncol) * 2u);
                        else gv = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(prm.gapg) + (long)img * prm.Nout + ncol);
                        float gp[8];
                        unpack16<T>(gv, gp);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = fmaf(gp[e], prm.gap_scale, f[e]);
                    }
                    const unsigned b = (ebw[t] >> (8 * (c8 & 3))) & 0xffu;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (!((b >> e) & 1u)) f[e] = 0.f;
                        if (ok) ssum[e] += f[e];
                    }
                    pk = pack16<T>(f);
                }
                request(t, wbn);
                if constexpr (FWD) {
                    if (HAND || go_wg != nullptr) {
                        const unsigned gb = gate_bits_of<T>(pk);
                        unsigned dw = gb << (8 * (c8 & 3));
                        dw |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)dw, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
                        dw |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)dw, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
                        if (HAND || (ok && (c8 & 3) == 0))
                            pl_store(reinterpret_cast<unsigned*>(go_wg + ((unsigned)rbase * 4u + bit_off + (unsigned)wb * 1024u)), dw);
                    }
                }
                if (ok)
                    pl_store(reinterpret_cast<u32x4*>(out_wg + ((unsigned)rbase * (unsigned)prm.Nout * 2u + lane_off)),
                             (u32x4){pk.x, pk.y, pk.z, pk.w});
            }
        }
        request_post(wbn);
        if constexpr (!FWD) {
            if (prm.sums != nullptr) {
                // lanes with equal c8 hold the same 8 channels for different rows; the two row halves of the panel are two
                // waves: their column sums meet in LDS by atomic adds (two commutative adds onto zero: one result)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
#pragma unroll
                    for (int off = 8; off < 64; off <<= 1) ssum[e] += __shfl_xor(ssum[e], off, 64);
                }
                if (lane < 8) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) atomicAdd(colsum + ncol + e, ssum[e]);
                }
            }
        }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        pl_drain<HAND>(wfa[ks]);
        pl_drain<HAND>(wfb[ks]);
    }
#pragma unroll
    for (int t = 0; t < NG; ++t) {
        pl_drain<HAND>(er[t]);
        if constexpr (!FWD) pl_drain<HAND>(ebw[t]);
    }
    if constexpr (FWD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pl_drain<HAND>(pq[i]);
    }
}

// PRO: 0 none, 1 relu(scale*c + shift), 2 k1*g + k2*c + k3.   EPI: 1 forward post, 0 input gradient, 3 input gradient with
// the low-resolution (stride 2) residual.  WIDE: panel_blocks_wide (64 channels x 64 rows per wave and block).
template <typename T, int K, int BM, int PRO, int EPI, bool HAND, bool WIDE = false>
__global__ __launch_bounds__(256, HAND ? 2 : 1) void panel_kernel(const PanelParams prm) {
    constexpr int NT = 256, NW = 4;
    constexpr int CPR = K / 8;       // 16-byte chunks per operand row
    constexpr int RPP = NT / CPR;    // rows staged per pass
    constexpr int NPASS = BM / RPP;
    constexpr int ROWB = K * 2;
    constexpr int SCR_BYTES = WIDE ? 32 * 144 : 32 * 80;  // per wave: 32 rows x (32 / 64 channels + 16 bytes: keeps ds_read_b128 aligned)
    constexpr bool FWD = EPI == 1;
    static_assert(K % 64 == 0 && BM % RPP == 0 && NT % CPR == 0 && BM % 32 == 0, "panel geometry");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* panel = smem;  // [BM][ROWB], chunk index XOR-swizzled by panel_swz(row)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* scratch = smem + BM * ROWB + wave * SCR_BYTES;                          // private to this wave
    float* colsum = reinterpret_cast<float*>(smem + BM * ROWB + NW * SCR_BYTES);  // [Nout] (input gradient with sums)
    char* gapl = smem + BM * ROWB + NW * SCR_BYTES + (FWD || prm.sums == nullptr ? 0 : prm.Nout * 4);  // [2][Nout] T (HAND with gapg)
    const long m0 = (long)blockIdx.x * BM;
    const int rows_left = (int)((long)prm.M - m0 < BM ? (long)prm.M - m0 : BM);  // rows of this panel inside the tensor

    if constexpr (!FWD && HAND) {
        if (prm.gapg != nullptr) {
            // the pooled-feature gradients of the (at most two) images this panel touches: read by the epilogue from LDS
            const unsigned i0 = fast_div((unsigned)m0, prm.div_pq);
            const unsigned nimg = fast_div((unsigned)(prm.M - 1), prm.div_pq) + 1u;
            const int cpr = prm.Nout >> 3;
            for (int i = tid; i < 2 * cpr; i += NT) {
                const unsigned img = i0 + (i >= cpr ? 1u : 0u);
                const int ch = i >= cpr ? i - cpr : i;
                *reinterpret_cast<uint4*>(gapl + i * 16) = *reinterpret_cast<const uint4*>(
                    reinterpret_cast<const T*>(prm.gapg) + (long)(img < nimg ? img : nimg - 1) * prm.Nout + ch * 8);
            }
        }
    }
    // ---------------- stage the operand panel: every pass requested up front, transformed in registers ----------------
    // (loads are unconditional from a clamped row -- a row past the tensor's end re-reads the last valid one and its
    //  results are never stored: a divergent branch around each load made hipcc drain the queue between passes)
    {
        const int cc = tid % CPR, rr = tid / CPR;
        const char* src_wg = reinterpret_cast<const char*>(prm.src) + m0 * ROWB;
        const char* srcc_wg = PRO == 2 ? reinterpret_cast<const char*>(prm.src_c) + m0 * ROWB : nullptr;
        uint4 v[NPASS], vc[PRO == 2 ? NPASS : 1];
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int row = rr + p * RPP;
            const unsigned off = (unsigned)(row < rows_left ? row : rows_left - 1) * ROWB + cc * 16;
            v[p] = (MSFWSI_PANEL_ABLATE & 16) ? make_uint4(off, 1, 2, 3) : *reinterpret_cast<const uint4*>(src_wg + off);
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
    if constexpr (WIDE && !FWD) {
        if (prm.sums != nullptr)  // two waves add into every column sum: from zero
            for (int n = tid; n < prm.Nout; n += NT) colsum[n] = 0.f;
    }
    __syncthreads();  // the only workgroup barrier before the sums: from here on every wave runs alone

    // HAND (chosen by the launcher): every panel whole and every optional operand of the epilogue class present
    if constexpr (WIDE) panel_blocks_wide<T, K, BM, EPI, HAND>(prm, panel, scratch, colsum, gapl, m0, rows_left, wave, lane);
    else panel_blocks<T, K, BM, EPI, HAND>(prm, panel, scratch, colsum, gapl, m0, rows_left, wave, lane);

    if constexpr (!FWD) {
        if (prm.sums != nullptr) {
            // one coalesced pass of fp64 atomics per workgroup (512 contiguous bytes per wave instruction; issued per
            // block by four lanes, the same adds cost the workgroup more than its HBM traffic)
            __syncthreads();
            double* dst = prm.sums + (long)(blockIdx.x % prm.nshard) * 2 * prm.Nout;
            for (int n = tid; n < prm.Nout; n += NT) atomicAdd(dst + n, (double)colsum[n]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Gram matrix and column sums of a = relu(scale*c + shift) straight from the raw conv output c:
//     A64[i][j] += sum_p a[p][i] a[p][j]        sums[i] += sum_p a[p][i]          (a rounded to the storage type)
// -- what msfwsi_bn_act_sum + msfwsi_gram produce in two passes with the normalised activation written and re-read in
// between (the statistics of bn3 follow from them: Engine._gram_stats, resnet.py:131-133).  Persistent workgroups walk
// 128-row panels: the panel is staged exactly like panel_kernel's (same LDS image), the whole K x K matrix lives in the
// workgroup's accumulators (K/32 x K/32 MFMA tiles spread over its waves; both operands come from the SAME image by the
// transposed LDS read), the next panel's loads are in flight while the current one is multiplied; fp64 atomics at the end.
struct GramParams {
    const void* src;
    const float* scale;
    const float* shift;
    double* A64;   // [K][K]
    double* sums;  // [K]
    long M;
    int npanel;
};

// fragment of the panel image for 32 channels starting at col0 and the 16 rows of k group ks (operand of a product that
// sums over ROWS): ds_read_b64_tr_b16 gathers 4 rows x 16 columns per 16-lane group (csrc/wgrad.hip read_tr_frag, on this
// file's chunk-swizzled rows)
template <typename T, int K>
__device__ __forceinline__ typename MmaFrag<T>::type panel_tr_frag(const char* panel, int ks, int col0, int lane) {
    typedef typename MmaFrag<T>::type frag_t;
    constexpr int ROWB = K * 2;
    const int li = lane & 15, G = lane >> 4;
    const int q = li >> 2, p = li & 3;
    const int kb = ks * 16 + (G >> 1) * 8 + q;
    const int cb = (col0 + (G & 1) * 16 + p * 4) * 2;
    const int o0 = kb * ROWB + ((((cb >> 4) ^ panel_swz<K>(kb)) << 4) | (cb & 15));
    const int o1 = (kb + 4) * ROWB + ((((cb >> 4) ^ panel_swz<K>(kb + 4)) << 4) | (cb & 15));
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(panel + o0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(panel + o1));
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(frag_t, both);
}

template <typename T, int K, int NWV>
__global__ __launch_bounds__(64 * NWV) void panel_gram_kernel(const GramParams prm) {
    constexpr int BM = 128, NT = 64 * NWV;
    constexpr int CPR = K / 8, RPP = NT / CPR, NPASS = BM / RPP, ROWB = K * 2;
    constexpr int TT = K / 32, TPW = TT * TT / NWV;  // MFMA tiles per dimension / per wave
    static_assert(TT * TT % NWV == 0 && BM % RPP == 0 && NT % CPR == 0, "gram geometry");
    typedef typename MmaFrag<T>::type frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* panel = smem;
    float* colsum = reinterpret_cast<float*>(smem + BM * ROWB);  // [K]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cc = tid % CPR, rr = tid / CPR;

    float csum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[e] = 0.f;
    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

    const char* src = reinterpret_cast<const char*>(prm.src);
    uint4 v[NPASS];
    auto request = [&](int p) __attribute__((always_inline)) {
        const long m0 = (long)p * BM;
        const int rows_left = (int)(prm.M - m0 < BM ? prm.M - m0 : BM);
        const char* base = src + m0 * ROWB;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int row = rr + i * RPP;
            v[i] = *reinterpret_cast<const uint4*>(base + (unsigned)((row < rows_left ? row : rows_left - 1) * ROWB + cc * 16));
        }
    };
    int p = blockIdx.x;
    if (p < prm.npanel) request(p);
    for (; p < prm.npanel; p += gridDim.x) {
        const long m0 = (long)p * BM;
        const int rows_left = (int)(prm.M - m0 < BM ? prm.M - m0 : BM);
        // (the BatchNorm map is re-read per panel -- L1 hits -- instead of living in 16 registers beside the accumulators)
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = prm.scale[cc * 8 + e];
            sh[e] = prm.shift[cc * 8 + e];
        }
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int row = rr + i * RPP;
            float f[8];
            unpack16<T>(v[i], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = row < rows_left ? round_to<T>(fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f)) : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) csum[e] += f[e];
            *reinterpret_cast<uint4*>(panel + row * ROWB + ((cc ^ panel_swz<K>(row)) << 4)) = pack16<T>(f);
        }
        __syncthreads();
        if (p + (int)gridDim.x < prm.npanel) request(p + gridDim.x);  // the next panel's rows fly during this one's MFMAs
#pragma unroll 1  // (unrolled, hipcc hoists every step's transposed reads: the 256-channel instance spilled)
        for (int ks = 0; ks < BM / 16; ++ks) {
            const int t0 = wave * TPW;
            const frag_t af = panel_tr_frag<T, K>(panel, ks, (t0 / TT) * 32, lane);
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const frag_t bf = panel_tr_frag<T, K>(panel, ks, ((t0 + i) % TT) * 32, lane);
                mma32<T>(acc[i], af, bf);
            }
        }
        __syncthreads();  // every wave is done with the image before the next panel overwrites it
    }
    // column sums: the NT / CPR threads that share a chunk column combine their partial sums in a FIXED order -- through the
    // panel's LDS image, free after the loop's last barrier -- and one fp64 atomic per channel leaves the workgroup.  (Round 5
    // combined them with fp32 atomic adds in LDS: 16-32 addends in arrival order, i.e. a column sum that differed in its last
    // bits from run to run -- the seed of the run-to-run differences tools/race_check.py showed on the Bottleneck path: the
    // sum feeds bn3's mean, and 16-bit storage amplifies one flipped rounding through every BatchNorm below it.)
    {
        float* part = reinterpret_cast<float*>(panel);  // [NT / CPR][K]
#pragma unroll
        for (int e = 0; e < 8; ++e) part[rr * K + cc * 8 + e] = csum[e];
        __syncthreads();
        for (int i = tid; i < K; i += NT) {
            double t = 0.0;
            for (int r = 0; r < NT / CPR; ++r) t += (double)part[r * K + i];
            atomicAdd(prm.sums + i, t);
        }
    }
    (void)colsum;
    // D[i][j]: lane -> column j, registers -> rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = wave * TPW + i;
        const int r0 = (t / TT) * 32, c0 = (t % TT) * 32;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = r0 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            atomicAdd(prm.A64 + (long)row * K + c0 + l31, (double)acc[i][reg]);
        }
    }
}

template <typename T, int K, int NWV>
int launch_gram(const GramParams& prm, hipStream_t stream) {
    constexpr int LDS = 128 * K * 2 + K * 4;
    void (*kern)(const GramParams) = panel_gram_kernel<T, K, NWV>;
    // workgroups resident on the device at once (the kernel is persistent); one process drives one GPU, two host threads of it
    // may launch at once (Engine._run_views): a relaxed atomic, both would store the same value
    static std::atomic<int> slots_cache{0};
    int slots = slots_cache.load(std::memory_order_relaxed);
    if (LDS > 64 * 1024) {
        if (int e = msfwsi_raise_lds(reinterpret_cast<const void*>(kern), LDS)) return e;
    }
    if (slots == 0) {
        int dev = 0, cus = 0, per = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, reinterpret_cast<const void*>(kern), 64 * NWV, LDS) != hipSuccess ||
            cus <= 0 || per <= 0)
            return MSFWSI_EUNSUPPORTED;
        slots = cus * per;
        slots_cache.store(slots, std::memory_order_relaxed);
    }
    const int grid = prm.npanel < slots ? prm.npanel : slots;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * NWV), LDS, stream, prm);
    return msfwsi_launch_status();
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

msfwsi_tunable g_panel_wide{1};  // msfwsi_set_tuning(18, .): 0 = 32-channel blocks everywhere (panel_blocks), the A/B reference of the wide form
msfwsi_tunable g_panel_hand{1};  // msfwsi_set_tuning(17, .): 0 = every launch on hipcc's own waits (the A/B reference of the hand-counted ones)

template <typename T, int K, int BM, int PRO, int EPI>
int launch_panel(const PanelParams& prm, hipStream_t stream) {
    // wide blocks (panel_blocks_wide): k <= 128 on 128-row panels, whole 64-channel blocks
    constexpr bool CAN_WIDE = K <= 128 && BM == 128;
    const bool wide = CAN_WIDE && g_panel_wide && prm.Nout % 64 == 0;
    const int LDS = BM * K * 2 + 4 * 32 * (wide ? 144 : 80) + (EPI != 1 && prm.sums != nullptr ? prm.Nout * 4 : 0) +
                    (EPI != 1 && prm.gapg != nullptr ? prm.Nout * 4 : 0);
    // hand-counted loads (panel_blocks) need a fixed number of vector-memory operations per block: whole panels only, and
    // every optional operand of the epilogue class present (the engine's launches all are: M = N*H*W with N a multiple of
    // 128 tiles, identity + gate bytes / residual + gate bytes + sums)
    const bool hand = g_panel_hand && prm.M % BM == 0 && (EPI == 1 ? (prm.ident != nullptr && prm.gate_out != nullptr)
                                                   : (prm.resid != nullptr && prm.mask_bits != nullptr &&
                                                      (prm.gapg == nullptr || prm.P * prm.Q >= BM)));  // (a panel within two images)
    void (*kern)(const PanelParams) = hand ? panel_kernel<T, K, BM, PRO, EPI, true> : panel_kernel<T, K, BM, PRO, EPI, false>;
    if constexpr (CAN_WIDE) {
        if (wide) kern = hand ? panel_kernel<T, K, BM, PRO, EPI, true, true> : panel_kernel<T, K, BM, PRO, EPI, false, true>;
    }
    if (LDS > 64 * 1024) {
        if (int e = msfwsi_raise_lds(reinterpret_cast<const void*>(kern), LDS)) return e;
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

extern "C" __attribute__((visibility("hidden"))) long msfwsi_panel_set_wide(long v, int write) {
    const long old = g_panel_wide;
    if (write) g_panel_wide = v;
    return old;
}

extern "C" __attribute__((visibility("hidden"))) long msfwsi_panel_set_hand(long v, int write) {
    const long old = g_panel_hand;
    if (write) g_panel_hand = v;
    return old;
}

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

extern "C" int msfwsi_panel_gram(int dtype, const void* c, const float* scale, const float* shift, double* A64, double* sums,
                                 long M, int C, void* stream) {
    MSFWSI_CHECK_ARG(c != nullptr && scale != nullptr && shift != nullptr && A64 != nullptr && sums != nullptr && M > 0);
    if ((dtype != MSFWSI_DT_BF16 && dtype != MSFWSI_DT_F16) || (C != 64 && C != 128) || M > 0x7fffffffL * 64)
        return MSFWSI_EUNSUPPORTED;
    GramParams prm{};
    prm.src = c; prm.scale = scale; prm.shift = shift; prm.A64 = A64; prm.sums = sums; prm.M = M;
    prm.npanel = (int)((M + 127) / 128);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MSFWSI_DT_BF16) {
        if (C == 64) return launch_gram<__bf16, 64, 4>(prm, st);
        return launch_gram<__bf16, 128, 4>(prm, st);
    }
    if (C == 64) return launch_gram<_Float16, 64, 4>(prm, st);
    return launch_gram<_Float16, 128, 4>(prm, st);
}
