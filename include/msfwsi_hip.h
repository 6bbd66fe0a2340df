/* C ABI of libmsfwsi_hip.so: the gfx950 (MI355X) kernels behind the MSF-WSI pre-train step.
 *
 * The reference (Dylan-H-Wang/msf-wsi) has no FFI of its own: its hot path reaches the device through
 * torch operators called from src/models/{resnet,backbone}.py and tools/ssl_train.py.  Each entry point
 * below therefore names the reference operator call it replaces (file:line into the reference tree).
 * The Python host (msf_wsi_amd/) binds these with ctypes; see INTEGRATION.md for the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch's caching allocator); the library
 *     never allocates, frees or synchronises; `stream` is a hipStream_t passed as void*.
 *   - return value: 0 ok; <0 invalid argument (-1) / unsupported configuration (-2); >0 a hipError_t.
 *   - re-entrant: safe to call from the forward thread and autograd's backward thread.  The only state the library
 *     keeps is the PROCESS-GLOBAL tuning switches of msfwsi_set_tuning (relaxed atomics; they select between kernels that
 *     compute the same results and default to the measured-fastest choice).
 *   - activations are NHWC ("channels last"), viewed as [M = N*H*W][C]; dtype selects the storage type
 *     of activations AND weights (MSFWSI_DT_F32 exact-fp32 MFMA path, MSFWSI_DT_BF16 / MSFWSI_DT_F16 16-bit
 *     MFMA with fp32 accumulation); per-channel vectors, statistics and weight gradients are always fp32 / fp64.
 *   - channel counts must be multiples of one 16-byte chunk (4 fp32 / 8 bf16 elements).
 */
#ifndef MSFWSI_HIP_H
#define MSFWSI_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MSFWSI_DT_F32 0
#define MSFWSI_DT_BF16 1
#define MSFWSI_DT_F16 2

/* Geometry of one convolution (a Linear layer is R=S=1, H=W=1, N=rows). */
typedef struct msfwsi_conv_desc {
    int dtype;
    int N, H, W, C; /* input  [N,H,W,C]   */
    int P, Q, K;    /* output [N,P,Q,K]   */
    int R, S, stride, pad;
} msfwsi_conv_desc;

/* ---- dense contractions (MFMA) ------------------------------------------------------------------ */

/* y = conv(act(x), w) [+ bias];  act = relu(pro_scale[c]*x + pro_shift[c]) when pro_* != NULL (the
 * producer's BatchNorm+ReLU fused into the operand load; padding stays zero).  w: [K][R][S][C].
 * stats != NULL: adds per-channel sum(y) and sum(y*y) of the STORED outputs into
 * stats[blockshard][2][K] (fp64, caller zeroes; nshard replicas spread the atomics).
 * Replaces: conv3x3/conv1x1/conv1 forward, src/models/resnet.py:25-33,174,232-242 (+ the statistics
 * half of the following nn.BatchNorm2d in train mode, resnet.py:68-77,124-134) and nn.Linear forward,
 * src/models/backbone.py:12-31,161-186,205-212. */
int msfwsi_conv_fwd(const msfwsi_conv_desc* d, const void* x, const void* w, void* y, const float* pro_scale,
                    const float* pro_shift, const float* bias, double* stats, int nshard, void* stream);

/* y = [relu]( round(conv(x, w)) * post_scale[k] + post_shift[k] + ident ): a conv whose consumer BatchNorm
 * statistics are already known (from msfwsi_fold_matvec / msfwsi_fold_dots), so BatchNorm apply, the residual add and
 * the ReLU of src/models/resnet.py:131-138 (bn3 -> += identity -> relu) run in the conv epilogue and the raw conv
 * output never reaches HBM.  ident may be NULL.  gate_out (nullable): one byte per 16-byte chunk of y (vec = 4 fp32 / 8
 * 16-bit elements, cpr = K/vec chunks per pixel), bit e = (stored y > 0) for element e of the chunk -- the ReLU gate
 * msfwsi_conv_dgrad reads back.  Layout: the byte of (pixel m, chunk c) is at m*cpr + c when cpr % 4 != 0, otherwise at
 * ((m/128)*(cpr/4) + c/4)*512 + (m%128)*4 + c%4 -- the four gate bytes of a 32-channel block form a dword and the dwords
 * of 128 consecutive pixels are contiguous; the tensor then holds ceil(M/128)*128*cpr bytes. */
int msfwsi_conv_fwd_post(const msfwsi_conv_desc* d, const void* x, const void* w, void* y, const float* post_scale,
                         const float* post_shift, const void* ident, int relu, unsigned char* gate_out, void* stream);

/* msfwsi_conv_fwd_post with a second source: y = [relu]( round(x . w_cat[:, 0:C] + src2 . w_cat[:, C:C+C2]) *
 * post_scale + post_shift + ident ), w_cat = [K][C + C2] (each output row holds both weight rows), src2 = [N,P,Q,C2].
 * A Bottleneck with a downsample branch is ONE such launch: relu(bn3(conv3(a2)) + bn_d(conv_d(x))) with the two
 * BatchNorm scales folded into the weight rows and the shifts summed (src/models/resnet.py:131-138 with
 * self.downsample).  MSFWSI_EUNSUPPORTED unless 1x1 / stride 1 with C and C2 multiples of the k slab. */
int msfwsi_conv_fwd_post2(const msfwsi_conv_desc* d, const void* x, const void* w_cat, void* y, const void* src2, int C2,
                          const float* post_scale, const float* post_shift, const void* ident, int relu,
                          unsigned char* gate_out, void* stream);

/* dx = conv_transpose(dy, w) [+ resid] [+ gap_scale * gapg[image]]  (input gradient).  w is the forward
 * weight [K][R][S][C], read in place as the [k][n] operand.  resid: [N,H,W,C] added element-wise (the
 * identity-path gradient of a residual block); gapg: [N][C] broadcast over H*W (global-average-pool
 * gradient).  mask_c != NULL fuses the backward of the activation that produced the conv input,
 * a = relu(mask_scale*c + mask_shift): dx is gated by (mask_scale*c + mask_shift > 0) and
 * sums[shard][2][C] += {sum dx, sum dx*c} (what msfwsi_act_bwd_reduce would compute in a second pass).
 * mask_bits != NULL (instead of mask_c): the gate comes from the bytes msfwsi_conv_fwd_post wrote
 * (same layout, over [N*H*W] pixels of C/vec chunks); sums slot 0 += sum dx, slot 1 is left alone.
 * resid_stride s = 2 (others: MSFWSI_EUNSUPPORTED): resid is the LOW-resolution tensor [N][(H-1)/s+1][(W-1)/s+1][C] and is added only at pixels
 * with h % s == w % s == 0 -- the input gradient of a stride-s downsample branch without its zero-stuffed copy.
 * Replaces: autograd's convolution_backward(input) / linear backward(input) (+ threshold_backward and the
 * reduction half of batch_norm_backward) reached through scaler.scale(loss).backward(), ssl_train.py:472. */
int msfwsi_conv_dgrad(const msfwsi_conv_desc* d, const void* dy, const void* w, void* dx, const void* resid,
                      const void* gapg, float gap_scale, const void* mask_c, const float* mask_scale,
                      const float* mask_shift, const unsigned char* mask_bits, double* sums, int nshard,
                      int resid_stride, void* stream);

/* Stem convolution (7x7 / stride 2 on 3 channels, src/models/resnet.py:174,234) on the pure-DMA kernel: x is
 * [N,H,W,CP] with the channels zero-padded to ONE 16-byte chunk (CP = 4 fp32 / 8 16-bit), w_run is
 * [K][R][run] where run = S*CP rounded up to whole k slabs (32 16-bit / 16 fp32 elements) and the padding columns are
 * zero: the S taps of a filter row are contiguous in NHWC memory, so a filter row is one tap over `run` channels.
 * y [N,P,Q,K] raw conv output, stats as in msfwsi_conv_fwd.  MSFWSI_EUNSUPPORTED for other shapes (K > 64 ...).
 * CP may also be a multiple of the chunk (the space-to-depth form below: CP = 16, R = S = 4, stride 1, pad 2);
 * P, Q > 0 give an explicitly cropped output extent (asymmetric padding), 0 = the convolution formula. */
int msfwsi_stem_conv_fwd(int dtype, const void* x, const void* w_run, void* y, double* stats, int nshard, int N, int H,
                         int W, int CP, int K, int R, int S, int stride, int pad, int P, int Q, void* stream);

/* The stem in space-to-depth form: the 7x7 / stride-2 / pad-3 conv on 3 channels (src/models/resnet.py:174) equals a
 * 4x4 / stride-1 conv (2 rows of padding above / left, H/2 output rows) on y[n][i][j][(a*2+b)*3+c] = x[n][c][2i+a][2j+b]
 * (12 channels padded to 16) with W2[k][ri][si][(a*2+b)*3+c] = W[k][2ri+a-1][2si+b-1][c]: k range 256 instead of 448,
 * half the input bytes.  nchw_to_s2d: fp32 NCHW [N,3,H,W] (H, W even) -> storage [N,H/2,W/2,16];
 * stem_s2d_weights: fp32 [K][7][7][3] -> storage [K][4][4][16]; stem_s2d_wfold: dw[K][7][7][3] += the gathered
 * dw2[K][4][4][16] (the weight gradient computed on the space-to-depth operand, folded back). */
int msfwsi_nchw_to_s2d(int dtype, const float* x, void* y, int N, int H, int W, void* stream);
int msfwsi_stem_s2d_weights(int dtype, const float* w, void* out, int K, void* stream);
int msfwsi_stem_s2d_wfold(const float* dw2, float* dw, int K, void* stream);

/* Two-source 1x1 input gradient: dx = gate( dy . w_cat[0:K] + src2 . w_cat[K:K+C2] + bias ), w_cat = [K+C2][C]
 * (rows K.. are a second [C2][C] matrix), src2 = [N,H,W,C2], bias fp32 [C] nullable, gate / sums as in
 * msfwsi_conv_dgrad.  One launch for the folded bn3 backward da2 = g (k1 o W) + a2 (W^T diag(k2) W) + W^T k3
 * (see msfwsi_fold_weights).  MSFWSI_EUNSUPPORTED unless 1x1 / stride 1 with K and C2 multiples of the k slab. */
int msfwsi_conv_dgrad2(const msfwsi_conv_desc* d, const void* dy, const void* w_cat, void* dx, const void* src2, int C2,
                       const float* bias, const void* mask_c, const float* mask_scale, const float* mask_shift,
                       double* sums, int nshard, void* stream);
/* The same with the second source given as the RAW conv output c2 [N,H,W,C2] of the layer whose BatchNorm + ReLU produces
 * the operand: a2 = relu(pro_scale * c2 + pro_shift) (fp32 [C2] each; src/models/resnet.py:128-130) is formed on the
 * fragments the kernel reads from LDS -- the arithmetic of msfwsi_bn_act, so the result is BIT FOR BIT that of
 * msfwsi_conv_dgrad2 on the materialised a2 -- and a2 is neither written nor read: with mask_c = c2 (the gate of the
 * folded tail's input gradient) the launch reads g and c2 and writes da2.  16-bit storage types; otherwise as
 * msfwsi_conv_dgrad2 (MSFWSI_EUNSUPPORTED also when 2 C2 floats exceed the tile's reduction area in LDS). */
int msfwsi_conv_dgrad2_pro(const msfwsi_conv_desc* d, const void* dy, const void* w_cat, void* dx, const void* c2, int C2,
                           const float* pro_scale, const float* pro_shift, const float* bias, const void* mask_c,
                           const float* mask_scale, const float* mask_shift, double* sums, int nshard, void* stream);

/* ---- activation-stationary ("panel") 1x1 convolutions: short k, wide output (csrc/panel.hip) -------------------
 * For the w -> 4w / 4w <- w convolutions of a Bottleneck (src/models/resnet.py:124,131): the [128 x k] operand panel of
 * a workgroup is read from HBM once, optionally TRANSFORMED on the way into LDS -- the BatchNorm+ReLU of the producer
 * (resnet.py:128-130) or the BatchNorm backward dc = k1*g + k2*c + k3 -- and stays resident while every 32-channel
 * output block streams past it; weights come pre-packed in MFMA fragment order.  16-bit storage types, 1x1 / stride 1,
 * k in {64, 128, 256, 512}, output channels >= 128 and a multiple of 32; otherwise MSFWSI_EUNSUPPORTED (callers then
 * use msfwsi_conv_fwd_post / msfwsi_conv_dgrad). */

/* 1 if the panel kernels serve the 1x1 convolution d (dgrad = 0: forward, k = d->C, outputs d->K; 1: input gradient,
 * k = d->K, outputs d->C). */
int msfwsi_panel_supported(const msfwsi_conv_desc* d, int dgrad);

/* wpk[n/32][k/16][64][8] <- W(n, k) = w[n * stride_n + k * stride_k]: the weight operand in MFMA fragment order (one
 * contiguous KiB per 32-channel block and 16-deep k step).  Forward of W[K][C]: Nout = K, K = C, strides (C, 1); input
 * gradient (the same tensor read as [k = K][n = C]): Nout = C, K = K, strides (1, C).  Nout % 32 == 0, K % 16 == 0. */
int msfwsi_panel_pack_weights(int dtype, const void* w, void* wpk, int Nout, int K, long stride_n, long stride_k,
                              void* stream);

/* y = [relu]( round(act(x) . W^T) * post_scale + post_shift + ident ) with act(x) = relu(pro_scale*x + pro_shift) when
 * pro_* != NULL (x is then the producer's RAW conv output: bn2 + ReLU of resnet.py:128-130 applied while the panel is
 * staged), else x itself.  Everything else as msfwsi_conv_fwd_post.  Replaces conv3 + bn3 + residual + ReLU,
 * src/models/resnet.py:131-138, without the normalised operand ever being stored. */
int msfwsi_panel_fwd_post(const msfwsi_conv_desc* d, const void* x, const float* pro_scale, const float* pro_shift,
                          const void* wpk, void* y, const float* post_scale, const float* post_shift, const void* ident,
                          int relu, unsigned char* gate_out, void* stream);

/* dx = gate( round(dc . W) + resid + gap_scale * gapg[image] ), sums[shard][0][C] += dx   (msfwsi_conv_dgrad's epilogue
 * with mask_bits), where dc = k1*dy + k2*c + k3 per channel when c != NULL -- the BatchNorm backward
 * (msfwsi_bn_bwd_apply) of the layer whose raw output is c, formed while the panel is staged; dc_out (nullable) receives
 * dc [N,P,Q,K] for the weight gradient.  c == NULL: dc = dy.  resid_stride 2: low-resolution residual as in
 * msfwsi_conv_dgrad.  Replaces batch_norm_backward + convolution_backward(input) of conv1 / bn1 of a Bottleneck
 * (src/models/resnet.py:124-126) reached through scaler.scale(loss).backward(), tools/ssl_train.py:472. */
int msfwsi_panel_dgrad(const msfwsi_conv_desc* d, const void* dy, const void* c, const float* k1, const float* k2,
                       const float* k3, void* dc_out, const void* wpk, void* dx, const void* resid, int resid_stride,
                       const void* gapg, float gap_scale, const unsigned char* mask_bits, double* sums, int nshard,
                       void* stream);

/* A64[i][j] += sum_p a[p][i]*a[p][j], sums[i] += sum_p a[p][i] for a = relu(scale*c + shift) rounded to the storage type:
 * msfwsi_bn_act_sum + msfwsi_gram in ONE pass over the raw conv output c [M][C], the normalised activation never
 * stored (bn3's batch statistics follow from these two, src/models/resnet.py:131-133 with the Gram-matrix algebra of
 * msfwsi_fold_matvec / msfwsi_fold_dots).  A64 [C][C] and sums [C] fp64, caller zeroes.  16-bit storage types, C in {64,
 * 128} (the whole C x C matrix lives in one workgroup's accumulators); otherwise MSFWSI_EUNSUPPORTED. */
int msfwsi_panel_gram(int dtype, const void* c, const float* scale, const float* shift, double* A64, double* sums, long M,
                      int C, void* stream);

/* ---- image-stationary 3x3 convolution of the deep layers (csrc/img3x3.hip) ----------------------------------------
 * conv2 of the Bottlenecks of layer2 / layer3 (src/models/resnet.py:25-28,128: 3x3, stride 1, pad 1, C == K) at 28x28x128
 * and 14x14x256, 16-bit storage: a workgroup keeps a band of 196 output pixels with its halo in LDS (read from HBM once;
 * a filter tap is a constant added to the fragment address) and streams the filter in MFMA fragment order -- no barrier
 * and no LDS-DMA in the k loop.  msfwsi_img3x3_supported says whether a geometry qualifies; otherwise the entry points
 * return MSFWSI_EUNSUPPORTED and callers use msfwsi_conv_fwd / msfwsi_conv_dgrad. */
int msfwsi_img3x3_supported(const msfwsi_conv_desc* d);
/* wpk <- the filter w [K][3][3][C] in fragment order: [n/32][tap*(Ck/16) + c/16][64 lanes][8]; dgrad = 0: output channel n
 * = K index, operand channel c = C index; dgrad = 1: n = C index, c = K index, taps flipped (the transposed convolution);
 * dgrad = 2 (K == C): the strided gradient's order, see msfwsi_img3x3_s2_dgrad. */
int msfwsi_img3x3_pack_weights(int dtype, const void* w, void* wpk, int K, int C, int dgrad, void* stream);
/* y = conv3x3(act(x), W) with act = relu(pro_scale*x + pro_shift) when pro_* != NULL (x = the producer's raw conv output,
 * resnet.py:125-128 fused; the zero padding stays zero), else x.  stats as in msfwsi_conv_fwd. */
int msfwsi_img3x3_fwd(const msfwsi_conv_desc* d, const void* x, const float* pro_scale, const float* pro_shift,
                      const void* wpk, void* y, double* stats, int nshard, void* stream);
/* dx = gate(conv3x3_transpose(dc, W)), sums as in msfwsi_conv_dgrad with mask_c; dc = k1*dy + k2*c + k3 when c != NULL (the
 * BatchNorm backward of the layer whose raw output is c, formed while the band is staged; dc_out (nullable) receives it
 * for the weight gradient; it may alias dy only at 14x14, where a workgroup owns a whole image: bands of a 28x28 image read
 * their halo rows from each other's gradient), else dy.  act_out (nullable, needs mask_c): receives the gating activation
 * relu(mask_scale*mask_c + mask_shift) -- what msfwsi_bn_act would write -- which is this conv's forward operand and so the
 * operand of its weight gradient (src/models/resnet.py:125-128 backwards: three stand-alone passes of a Bottleneck's backward
 * become by-products of this launch). */
int msfwsi_img3x3_dgrad(const msfwsi_conv_desc* d, const void* dy, const void* c, const float* k1, const float* k2,
                        const float* k3, void* dc_out, const void* wpk, void* dx, const void* mask_c,
                        const float* mask_scale, const float* mask_shift, void* act_out, double* sums, int nshard,
                        void* stream);

/* The same for the STRIDED conv2 of layer2.0 / layer3.0 (3x3 / stride 2 / pad 1, C == K; src/models/resnet.py:128 with
 * stride 2): the input gradient dx [N][2P][2Q][C] from dy [N][P][Q][C] at P = 28, C = 128 and P = 14, C = 256 (d describes the
 * forward conv: H = 2P).  One launch instead of msfwsi_conv_dgrad's four parity launches: a band of dy is staged once and
 * serves the four parities of the output position.  Arguments as msfwsi_img3x3_dgrad: c / k1..k3 / dc_out at dy's
 * resolution (dc_out may alias dy only at P = 14), mask_c / act_out / sums at dx's.  wpk from msfwsi_img3x3_pack_weights
 * with dgrad = 2 (the passes' tap order). */
int msfwsi_img3x3_s2_dgrad_supported(const msfwsi_conv_desc* d);
int msfwsi_img3x3_s2_dgrad(const msfwsi_conv_desc* d, const void* dy, const void* c, const float* k1, const float* k2,
                           const float* k3, void* dc_out, const void* wpk, void* dx, const void* mask_c,
                           const float* mask_scale, const float* mask_shift, void* act_out, double* sums, int nshard,
                           void* stream);

/* Specialised 3x3 / stride 1 / pad 1 path: the input patch of 256 raster pixels (+ halo) is staged once per
 * channel slab in LDS and reused by all nine taps (see csrc/conv3x3.hip).  Same results as msfwsi_conv_fwd /
 * msfwsi_conv_dgrad without prologue/bias/gapg; `supported` tells whether a geometry qualifies.
 * Replaces: conv3x3 forward / backward(input), src/models/resnet.py:25-28. */
int msfwsi_conv3x3_supported(const msfwsi_conv_desc* d);
/* 1 if msfwsi_conv3x3_fwd / _dgrad serve this geometry with the weights-stationary persistent kernel (2-byte types,
 * 64 -> 64 channels: all nine taps of the filter resident in LDS, each halo pixel loaded once); callers then prefer
 * them over msfwsi_conv_fwd / msfwsi_conv_dgrad. */
int msfwsi_conv3x3_stationary(const msfwsi_conv_desc* d);
/* pro_scale / pro_shift (both or neither; only where msfwsi_conv3x3_stationary() is 1, else MSFWSI_EUNSUPPORTED): x is
 * the producer's RAW conv output and the conv sees relu(scale[c] * x + shift[c]) -- conv2 of a block reads bn1+ReLU of
 * conv1's output without that activation ever being stored (src/models/resnet.py:120-126 fused). */
int msfwsi_conv3x3_fwd(const msfwsi_conv_desc* d, const void* x, const void* w, void* y, double* stats, int nshard,
                       const float* pro_scale, const float* pro_shift, void* stream);
int msfwsi_conv3x3_dgrad(const msfwsi_conv_desc* d, const void* dy, const void* w, void* dx, const void* resid,
                         const void* mask_c, const float* mask_scale, const float* mask_shift, double* sums,
                         int nshard, void* stream);

/* dw[K][R][S][C] (fp32) += dy^T * act(x)   (weight gradient, split over pixels, fp32 atomics).
 * target_blocks: workgroup budget used to pick the split factor (<=0: the workgroups resident on the device at once).
 * Replaces: convolution_backward(weight) / linear backward(weight), tools/ssl_train.py:472. */
int msfwsi_conv_wgrad(const msfwsi_conv_desc* d, const void* x, const void* dy, float* dw, const float* pro_scale,
                      const float* pro_shift, int target_blocks, void* stream);
/* msfwsi_conv_wgrad for a 1x1 / stride-1 conv (16-bit storage) whose operand x is the producer's RAW output under pro_scale /
 * pro_shift, with the normalised operand as a by-product: act_out [N*H*W][C] receives relu(pro_scale*x + pro_shift), bit for
 * bit what msfwsi_bn_act writes.  MSFWSI_EUNSUPPORTED for other geometries / fp32.
 * Replaces: native_batch_norm (apply) + relu + convolution_backward(weight), src/models/resnet.py:128-131 backwards. */
int msfwsi_conv_wgrad_act(const msfwsi_conv_desc* d, const void* x, const void* dy, float* dw, const float* pro_scale,
                          const float* pro_shift, void* act_out, int target_blocks, void* stream);
/* dw[K][R][S][C] (fp32) = dy^T * x, STORED (not accumulated): one workgroup per gradient tile over all pixels, no
 * atomics, dw need not be cleared.  For gradients with ONE launch per step and few rows -- the heads' Linear layers
 * (both views stacked), whose fp32 matrices (fuser: 1.36 GB each) are then written once instead of cleared, read and
 * written.  Same results as msfwsi_conv_wgrad on a cleared dw up to the order of the fp32 sums.
 * Replaces: linear backward(weight), tools/ssl_train.py:472 (+ optimizer.zero_grad for these tensors, :471). */
int msfwsi_conv_wgrad_store(const msfwsi_conv_desc* d, const void* x, const void* dy, float* dw, void* stream);
/* Gram matrix A64[C][C] (fp64) += a^T a over the N*H*W pixels of the NHWC activation a (d: the 1x1 / stride-1 conv C -> C
 * whose weight gradient with x = dy = a it is; same MFMA kernels as msfwsi_conv_wgrad, the pixel splits accumulate in
 * fp64).  The folded Bottleneck tail takes bn3's batch statistics from it (sum c^2 = diag(W A W^T)); the fp64
 * accumulation makes them -- and with them the forward -- independent of the order of the atomic additions.
 * Replaces: the statistics pass of native_batch_norm over conv3's output, src/models/resnet.py:131-138. */
int msfwsi_gram(const msfwsi_conv_desc* d, const void* a, double* A64, void* stream);
/* The stem's weight gradient (space-to-depth form, 2-byte types) with the BatchNorm backward of bn1 applied on the
 * fly: dw[64][4][4][16] += (k1*g + k2*c0 + k3)^T x, where g is the gated gradient of bn1's output, c0 the raw conv
 * output and k1..k3 the coefficients of msfwsi_bn_bwd_finalize; the bracket is rounded to the storage type exactly as
 * msfwsi_bn_bwd_apply rounds it, but is neither written nor re-read (-2 passes over the largest tensor of the network).
 * MSFWSI_EUNSUPPORTED where the output-stationary stem kernel does not apply (other geometries / fp32 / small batches):
 * the caller then runs msfwsi_bn_bwd_apply + msfwsi_conv_wgrad.
 * Replaces: native_batch_norm_backward + convolution_backward(weight) of conv1/bn1, src/models/resnet.py:234-235. */
int msfwsi_stem_wgrad_bnbwd(const msfwsi_conv_desc* d, const void* x, const void* g, const void* c0, const float* k1,
                            const float* k2, const float* k3, float* dw, void* stream);
/* 1 if msfwsi_conv_wgrad serves this geometry with the output-stationary persistent kernel (2-byte types, 64 -> 64
 * channels, 3x3 / stride 1 / pad 1): the whole gradient stays in one workgroup's accumulators, every activation element
 * is loaded once, and pro_scale / pro_shift cost nothing -- callers pass them instead of materialising the activation. */
int msfwsi_conv_wgrad_stationary(const msfwsi_conv_desc* d);

/* ---- BatchNorm (training mode) ------------------------------------------------------------------ */

/* sums[nshard][2][C] (fp64: sum, sum of squares over `count` elements per channel) -> scale/shift
 * (gamma*invstd, beta-mean*scale), mean, invstd; updates running_mean/var (momentum, unbiased variance)
 * and num_batches_tracked when non-NULL.  gamma/beta NULL = affine=False.
 * Replaces: F.batch_norm(training=True) statistics of nn.BatchNorm2d / BatchNorm1d / SyncBatchNorm,
 * src/models/resnet.py:175 etc., src/models/backbone.py:15,18,21,28; tools/ssl_train.py:160. */
int msfwsi_bn_finalize(const double* sums, int nshard, int C, double count, const float* gamma, const float* beta,
                       float eps, float momentum, float* running_mean, float* running_var,
                       long* num_batches_tracked, float* scale, float* shift, float* mean, float* invstd,
                       void* stream);

/* BatchNorm in eval mode: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale (mean =
 * running_mean, invstd = 1/sqrt(running_var + eps)); no batch statistics, no cross-replica exchange.
 * Replaces: F.batch_norm(training=False) of nn.BatchNorm2d / BatchNorm1d under model.eval(),
 * src/models/resnet.py:175 etc. (feature extraction through the pre-trained encoders). */
int msfwsi_bn_eval_coeffs(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                          float eps, int C, float* scale, float* shift, float* mean, float* invstd, void* stream);

/* out[n] = sum over nshard replicas of in[shard][n]: packs the cross-replica (SyncBatchNorm) message. */
int msfwsi_shard_sum(const double* in, int nshard, int n, double* out, void* stream);

/* out = act(scale*c + shift [+ ident | + id_scale*ident + id_shift]); relu != 0 applies ReLU.
 * Replaces: bn2/bn3 apply + residual add + ReLU, resnet.py:76-80,131-138 (id_scale: the downsample
 * branch's BatchNorm, resnet.py:222-225); final projector BatchNorm1d(affine=False), backbone.py:21. */
int msfwsi_bn_act(int dtype, const void* c, const float* scale, const float* shift, const void* ident,
                  const float* id_scale, const float* id_shift, int relu, void* out, long M, int C, void* stream);

/* msfwsi_bn_act (relu, no identity) that also adds the column sums of its output into sums[shard][C] (fp64,
 * nshard replicas: 2048 workgroups adding into ONE replica serialise on 64 addresses for ~300 us): the normalised
 * conv3 operand of the folded bn3 forward / backward and its sum in one pass. */
int msfwsi_bn_act_sum(int dtype, const void* c, const float* scale, const float* shift, void* out, double* sums,
                      int nshard, long M, int C, void* stream);

/* Backward at a residual-block output y = relu(bn(c_main) + identity):
 * g = (dy + gap_scale*gapg[image]) * (y>0); sums[shard][3][C] += {sum g, sum g*c_main, sum g*c_ds}.
 * dy or gapg may be NULL (not both), c_ds may be NULL; c_main may be NULL (slot 1 untouched: the main-branch
 * sum then comes from msfwsi_fold_dots). */
int msfwsi_block_end_bwd(int dtype, const void* dy, const void* y, const void* gapg, float gap_scale,
                         const void* c_main, const void* c_ds, void* g, double* sums, int nshard, long M, int HW,
                         int C, void* stream);

/* Backward through an inner activation relu(scale*c+shift): g = da * (scale*c+shift > 0) (g may alias
 * da); sums[shard][2][C] += {sum g, sum g*c}.  scale == NULL: no activation (g not written, g = da). */
int msfwsi_act_bwd_reduce(int dtype, const void* da, const void* c, const float* scale, const float* shift,
                          void* g, double* sums, int nshard, long M, int C, void* stream);

/* BatchNorm backward coefficients: dc = k1*g + k2*c + k3; dgamma += sum g*xhat; dbeta += sum g.
 * sums[nshard][nslots][C]; `which` picks the g*c slot (1, or 2 for the downsample branch). */
int msfwsi_bn_bwd_finalize(const double* sums, int nshard, int nslots, int which, int C, double count,
                           const float* gamma, const float* mean, const float* invstd, float* dgamma,
                           float* dbeta, float* k1, float* k2, float* k3, void* stream);

/* dc = k1[c]*g + k2[c]*c + k3[c]  (batch_norm_backward input gradient). */
int msfwsi_bn_bwd_apply(int dtype, const void* g, const void* c, const float* k1, const float* k2,
                        const float* k3, void* dc, long M, int C, void* stream);

/* ---- stem / pooling / data movement --------------------------------------------------------------- */

/* fp32 NCHW image batch -> NHWC storage type with channels zero-padded to CP.
 * Replaces the layout the batch of tools/ssl_train.py:430-438 has when it enters conv1. */
int msfwsi_nchw_to_nhwc(int dtype, const float* x, void* y, int N, int C, int H, int W, int CP, void* stream);

/* out = maxpool3x3/s2/p1(relu(scale*c0+shift)); argmax = window slot (0..8) of the first maximum.
 * Replaces: bn1 apply + relu + maxpool, src/models/resnet.py:235-237. */
int msfwsi_stem_pool_fwd(int dtype, const void* c0, const float* scale, const float* shift, void* out,
                         unsigned char* argmax, int N, int H, int W, int C, void* stream);

/* Backward of relu + maxpool at the stem (resnet.py:235-237), g = relu'(.) * maxpool_backward(dp), in the two passes
 * BatchNorm's backward forces (all of sum g, sum g*c0 before any dc0):
 *   k1 == NULL: sums[shard][2][C] += {sum g, sum g*c0}; g0 (nullable) = g
 *   k1 != NULL: g0 = k1*g + k2*c0 + k3 (g re-derived from dp / argmax instead of written and re-read); sums nullable
 * dact (nullable, [N][H][W][C]): a gradient of the stem activation itself, added before the gate (the U-Net skip taken
 * before the max-pool: smp's ResNetEncoder stage 1, used by reference src/models/hooknet.py through smp.Unet). */
int msfwsi_stem_pool_bwd(int dtype, const void* dp, const unsigned char* argmax, const void* c0,
                         const float* scale, const float* shift, void* g0, double* sums, int nshard, const float* k1,
                         const float* k2, const float* k3, const void* dact, int N, int H, int W, int C, void* stream);

/* out[n][c] = mean over HW of y[n][hw][c].  Replaces: AdaptiveAvgPool2d((1,1)) + flatten on the four
 * stage outputs, src/models/resnet.py:244-250. */
int msfwsi_gap_fwd(int dtype, const void* y, void* out, int N, int HW, int C, void* stream);
/* The same pass also writing sout [N][ceil(H/2)][ceil(W/2)][C] = y[:, ::2, ::2, :] -- what msfwsi_pixel_stride(stride 2) makes
 * for the next stage's strided downsample conv (src/models/resnet.py:181-185, 137): one read of y for both. */
int msfwsi_gap_fwd_stride2(int dtype, const void* y, void* out, void* sout, int N, int H, int W, int C, void* stream);

/* BatchNorm backward folded into the weights of the 1x1 conv W[K][C] that produced the BatchNorm input
 * (Bottleneck conv3 -> bn3, src/models/resnet.py:115-117, backward by autograd in the reference): with c = W a,
 *   fold_dots:    out[k] = sum_c W[k][c]*M[k][c]  (= sum over pixels of g*c, M = g^T a from msfwsi_conv_wgrad)
 *   fold_weights: dW += k1 o M + k2 o WA + k3 (x) sa;  Wk1 = k1 o W;  Wk2 = k2 o W;  bvec[c] (fp64) += sum_k k3[k] W[k][c]
 * (WA = W (a^T a), sa = column sums of a, k1..k3 from msfwsi_bn_bwd_finalize).  All fp32 / fp64. */
int msfwsi_fold_dots(const float* W, const float* M, double* out, int K, int C, void* stream);
/* out[k] = [s1[k]*W1[k][0:C1] | s2[k]*W2[k][0:C2]] (fp32 [K][C1+C2]), shift[k] = b1[k] + b2[k]: two BatchNorm
 * affines folded into the rows of the two 1x1 weight matrices whose outputs are summed (msfwsi_conv_fwd_post2). */
int msfwsi_row_scale_cat(const float* W1, const float* s1, int C1, const float* W2, const float* s2, int C2,
                         const float* b1, const float* b2, float* out, float* shift, int K, void* stream);
/* out[k] = sum_c W[k][c]*v[c] (fp64).  With v = column sums of a and fold_dots(W, W (a^T a)) this yields the
 * BatchNorm statistics {sum c, sum c^2} of c = W a WITHOUT forming c: the forward of conv3 -> bn3 then runs the
 * conv once with the BatchNorm apply + residual + ReLU in its epilogue (msfwsi_conv_fwd_post). */
int msfwsi_fold_matvec(const float* W, const double* v, double* out, int K, int C, void* stream);
int msfwsi_fold_weights(const float* W, const float* M, const float* WA, const float* k1, const float* k2,
                        const float* k3, const double* sa, float* dW, float* Wk1, float* Wk2, double* bvec, int K, int C,
                        void* stream);

/* column sums of x[M][C] added into sums[shard][C] (fp64, nshard replicas against same-address atomic contention)
 * -- bias gradient of backbone.py:30's Linear; column sums of a folded BatchNorm's operand. */
int msfwsi_colsum(int dtype, const void* x, double* sums, int nshard, long M, int C, void* stream);

/* sums[nshard][2][C] += {sum_m x, sum_m x^2} of x[M][C], accumulated in fp64 throughout: the statistics of the heads'
 * BatchNorm1d (src/models/backbone.py:15,18,21,28), whose inputs have a batch mean far larger than their batch
 * deviation (E[x^2] - mean^2 cancels 3-4 digits; the GEMM epilogue's fp32 partial sums are not enough there). */
int msfwsi_colstats(int dtype, const void* x, double* sums, int nshard, long M, int C, void* stream);
int msfwsi_add_f64(const double* in, double* out, int n, void* stream); /* out[i] += in[i] (fp64 statistics vectors) */
int msfwsi_add_f64_to_f32(const double* in, float* out, int n, float alpha, void* stream);

/* scatter == 0: out[b*K+k] = in[b*K+idx[b][k]] (jigsaw un-shuffle, src/models/backbone.py:147-158);
 * scatter == 1: its adjoint, out[b*K+idx[b][k]] (+)= in[b*K+k].  idx: int64 [B][K] on the device. */
int msfwsi_rows_permute(int dtype, const void* in, const long* idx, void* out, int B, int K, int C, int scatter,
                        int accumulate, void* stream);

/* Strided pixel subsampling of an NHWC tensor [N,H,W,C] and its adjoint (P = (H-1)/stride+1, Q likewise):
 * expand == 0: out[N,P,Q,C] = in[:, ::stride, ::stride, :] -- the operand of a stride-s 1x1 conv (the downsample
 * branch, src/models/resnet.py:222-225) as a dense tensor; expand == 1: out[N,H,W,C] = in[N,P,Q,C] zero-stuffed
 * (the input gradient of that subsampling). */
int msfwsi_pixel_stride(int dtype, const void* in, void* out, int N, int H, int W, int C, int stride, int expand,
                        void* stream);

/* dst[r][0:cols] (+)= src[r][0:cols] with row strides (fuser concat, backbone.py:195-202, and adjoint). */
int msfwsi_copy2d(int dtype, const void* src, long src_ld, void* dst, long dst_ld, long rows, int cols,
                  int accumulate, void* stream);

/* ---- loss / optimizer ------------------------------------------------------------------------- */

/* *loss_accum += coef * sum_rows cos(p_row, z_row);  dp = (*loss_scale) * coef * dcos/dp  (z constant:
 * stop-gradient).  Replaces: nn.CosineSimilarity(dim=1)(p, z).mean() terms and their backward,
 * tools/ssl_train.py:422,448-466; coef = -0.5 * fuser_weight / rows. */
int msfwsi_cosine_loss(int dtype, const void* p, const void* z, long rows, int d, float coef,
                       const float* loss_scale, float eps, double* loss_accum, void* dp, void* stream);

/* ---- InfoNCE variant of the loss (BASELINE.json north_star: "InfoNCE-style contrastive loss ... all-gather of embeddings
 * for the cross-GPU negative set").  The reference has NO such code (tools/ssl_train.py:422,448-466 is the SimSiam cosine
 * loss): this is an optional mode of PretrainStep, parity unpinned (checked against a torch restatement:
 * F.normalize, matmul / temperature, F.cross_entropy).  Rows of p against ALL ranks' z (gathered), positives on the
 * diagonal: row_l2norm (xhat = x / max(||x||, eps), inv = 1 / max(||x||, eps)), its backward
 * dx = inv * (dxhat - xhat <xhat, dxhat>), and softmax_ce on logits [rows][n] (label = label0 + row):
 * *loss_accum += coef * (logsumexp(l / tau) - l[label] / tau); write_grad: logits <- coef * *grad_scale / tau *
 * (softmax - onehot) in place.  The two GEMMs are msfwsi_conv_fwd / msfwsi_conv_dgrad (1x1, H = W = 1). */
int msfwsi_row_l2norm(int dtype, const void* x, void* out, float* inv, long rows, int d, float eps, void* stream);
int msfwsi_row_l2norm_bwd(int dtype, const void* xhat, const void* dxhat, const float* inv, void* dx, long rows, int d,
                          void* stream);
int msfwsi_softmax_ce(int dtype, void* logits, long rows, int n, long label0, float inv_tau, float coef,
                      const float* grad_scale, double* loss_accum, int write_grad, void* stream);

/* GradScaler pieces (tools/ssl_train.py:100,472-474): *found = 1 if any gradient is inf/nan; scale update. */
int msfwsi_nonfinite_check(const float* g, long n, float* found, void* stream);
int msfwsi_scaler_update(float* scale, int* growth_tracker, const float* found, float growth_factor,
                         float backoff_factor, int growth_interval, void* stream);

/* One Adam step over a flat fp32 parameter group (torch.optim.Adam defaults, tools/ssl_train.py:309,473);
 * grads are divided by *loss_scale, the step is skipped when *found > 0; p_lowp != NULL also refreshes
 * the 16-bit compute copy (lowp_dtype = MSFWSI_DT_BF16 or MSFWSI_DT_F16).  step_dev != NULL: Adam's step
 * count is read from the device (the host `step` is ignored) and the bias corrections are formed in the kernel. */
int msfwsi_adam(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                float eps, long step, const int* step_dev, const float* loss_scale, const float* found, void* p_lowp,
                int lowp_dtype, void* stream);

/* *step += 1 unless *found > 0: torch's GradScaler.step (tools/ssl_train.py:473) does not call optimizer.step on an
 * overflowed step, so Adam's per-parameter "step" (bias correction, checkpoint) does not count skipped steps. */
int msfwsi_adam_step_advance(int* step, const float* found, void* stream);

/* fp32 -> bf16 / fp16 compute copy of a flat weight buffer (dtype = MSFWSI_DT_BF16 or MSFWSI_DT_F16) */
int msfwsi_cast_lowp(int dtype, const float* src, void* dst, long n, void* stream);
int msfwsi_pad_cast(int dtype, const float* src, void* dst, long rows, int C, int CP, void* stream);
/* bf16 / fp16 -> fp32 (replaces Tensor.float() on the 16-bit weight copies: the folded BatchNorm algebra of
 * src/models/resnet.py:131-138 is evaluated on exactly the weights the MFMA multiplies) */
int msfwsi_upcast_f32(int dtype, const void* src, float* dst, long n, void* stream);
/* p[r*ld + c] = 0 for r < rows, c < cols (fp64): clears one slot of a sharded statistics accumulator */
int msfwsi_zero_f64_2d(double* p, long rows, int cols, long ld, void* stream);
int msfwsi_unpad_add(const float* src, float* dst, long rows, int C, int CP, void* stream);

/* ---- validation metrics (row f4 of SURVEY.md 8f) -------------------------------------------------------------------
 * Per-image, per-class confusion counts of a multiclass segmentation, and the scores the reference logs.
 * Replaces: torch.argmax(preds, dim=1) + smp.metrics.get_stats(pred - 1, target - 1, mode="multiclass",
 * ignore_index=-1, num_classes=C) (tools/ssl_finetune.py:526-533, tools/evaluate.py:285-305) and smp.metrics.f1_score /
 * iou_score / accuracy with reduction "micro" and None (:535-551).  segmentation_models_pytorch (>= 0.3.2) is a
 * third-party dependency outside the reference tree: its published algorithm is restated, parity unpinned.
 * Either `logits` [N][nch][L] (storage dtype; pred = argmax over nch, first maximum) or `pred` [N][L] int64 is given;
 * pred_shift / target_shift are added first (the reference's "- 1"); counts [N][4][C] uint64 scratch, ZEROED by the
 * caller; tp/fp/fn/tn [N][C] int64 outputs.  Classes <= 64. */
int msfwsi_seg_stats(int logits_dtype, const void* logits, int nch, const long* pred, const long* target, int N, long L,
                     int C, long pred_shift, long target_shift, long ignore_index, int has_ignore,
                     unsigned long long* counts, long* tp, long* fp, long* fn, long* tn, void* stream);
/* scores[0..2] = micro F1, IoU, accuracy over all images and classes; scores[3+c], [3+C+c], [3+2C+c] = per-class F1, IoU,
 * accuracy on the counts summed over images; 0/0 -> zero_division (smp default 1.0).  scores: 3 + 3*C doubles. */
int msfwsi_seg_scores(const long* tp, const long* fp, const long* fn, const long* tn, int N, int C, double zero_division,
                      double* scores, void* stream);
/* smp's image-wise reductions (smp.metrics.f1_score(..., reduction="micro-imagewise"), tools/ssl_finetune.py:319):
 * scores[0..2] = F1, IoU, accuracy computed per image on its counts summed over classes, averaged over images
 * ("micro-imagewise"); scores[3..5] = the same scores per (image, class), averaged over both ("macro-imagewise"). */
int msfwsi_seg_scores_imagewise(const long* tp, const long* fp, const long* fn, const long* tn, int N, int C,
                                double zero_division, double* scores, void* stream);

/* ---- fine-tune model: U-Net decoder pieces (row f2 of SURVEY.md 8f, BASELINE config 5) --------------------------------
 * The arithmetic of HookNet's decoders lives in segmentation_models_pytorch (third party, outside the reference tree,
 * absent here): its published algorithm is restated, parity unpinned; call sites: src/models/hooknet.py:15-35,84-100.
 * out[n][2h][2w][Cx+Cs] = [nearest-x2(x) | skip]: DecoderBlock's F.interpolate(scale_factor=2, "nearest") + torch.cat. */
int msfwsi_upcat_fwd(int dtype, const void* x, const void* skip, void* out, int N, int h, int w, int Cx, int Cs,
                     void* stream);
/* adjoint: dx = 2x2 window sums of dout[..., :Cx]; dskip (nullable when Cs == 0) = dout[..., Cx:] */
int msfwsi_upcat_bwd(int dtype, const void* dout, void* dx, void* dskip, int N, int h, int w, int Cx, int Cs,
                     void* stream);
/* backward == 0: out[N][ch][cw][C] = x[:, y0:y0+ch, x0:x0+cw, :] (the hook x[:, :, 12:20, 12:20], hooknet.py:29-32);
 * backward != 0: x[window] += out (its adjoint, accumulated into the gradient that also comes from the next block) */
int msfwsi_crop(int dtype, void* x, void* out, int N, int H, int W, int C, int y0, int x0, int ch, int cw, int backward,
                void* stream);
/* y[N][C][HW] fp32 = x[N][HW][CP][:C] (the logits handed back to the caller in the reference's NCHW layout) */
int msfwsi_nhwc_to_nchw(int dtype, const void* x, float* y, int N, int C, long HW, int CP, void* stream);
/* smp.losses.DiceLoss(MULTICLASS_MODE, classes, from_logits=True) (tools/ssl_finetune.py:287-288) forward + backward on
 * NHWC logits [M][CP] (C1 real channels) and int64 targets [M]: *loss += weight * mean_{c in class_mask}
 * (1 - 2 sum p_c t_c / max(sum p_c + t_c, eps)) [sum t_c > 0]; dlogits (nullable) = *grad_scale * weight * dLoss/dlogits.
 * sums [3][C1] fp64 ZEROED by the caller; coef [2][C1] fp32 scratch.  C1 <= 32. */
int msfwsi_dice_loss(int dtype, const void* logits, const long* target, long M, int C1, int CP, unsigned class_mask,
                     double eps, double smooth, double weight, double* sums, double* loss, float* coef,
                     const float* grad_scale, void* dlogits, void* stream);

/* ---- tiling / normalising front end (row f3 of SURVEY.md 8f) --------------------------------------------------------
 * out[b][k] (fp32 [3][S][S]) = Normalize(HFlip?(resize_bilinear(crop(block perm[b][k] of img[b], box[b][k]), S x S)))
 * for the grid x grid blocks of an H x W x 3 uint8 tile; grid = 1 treats the whole tile as one block (context view).
 * Replaces the per-sample CPU work of BcssPretrainDataset.__getitem__ after the colour augmentations:
 * blockshaped (src/utils/data/bcss.py:203-216), target_grid[jigsaw_idx] (:176), RandomResizedCrop + HorizontalFlip +
 * Normalize + ToTensorV2 (tools/ssl_train.py:176-214) with the random decisions (boxes [B][K][4] = x0,y0,w,h inside the
 * block; flips [B][K]; perm [B][K]) supplied by the caller.  Bilinear resize in fp32 with cv2's half-pixel convention,
 * rounded to uint8 levels (cv2's fixed-point arithmetic is not reproduced: resize parity unpinned; a box of exactly
 * S x S is an exact copy).  perm / flips may be NULL. */
int msfwsi_tile_views(const unsigned char* img, int B, int H, int W, int grid, const long* perm, const int* boxes,
                      const unsigned char* flips, const float* mean, const float* std_, float max_pixel, int S,
                      float* out, void* stream);
/* The crop + bilinear resize of msfwsi_tile_views alone: out[b][k] (uint8 [S][S][3]) -- no flip, no Normalize.  The context
 * view's colour augmentations sit between its crop and its flip (tools/ssl_train.py:176-196). */
int msfwsi_tile_crops_u8(const unsigned char* img, int B, int H, int W, int grid, const long* perm, const int* boxes, int S,
                         unsigned char* out, void* stream);
/* ---- colour augmentations (row f3; PARITY UNPINNED: albumentations / cv2 arithmetic restated, see csrc/augment.hip) ----
 * Images: uint8 [N][H][W][3] RGB.  The random decisions are the caller's (msf_wsi_amd/augment.py draws them).
 * Replaces: albu.ColorJitter / ToGray / OneOf(GaussianBlur, Sharpen) of context_aug and target_aug,
 * tools/ssl_train.py:176-201, applied by BcssPretrainDataset.__getitem__, src/utils/data/bcss.py:166-170.
 *   msfwsi_gray_sum:     sums[n] += sum over the image of gray(pixel) (cv2 RGB2GRAY, 8-bit); sums ZEROED by the caller
 *   msfwsi_color_stage:  one adjustment per image: op[n] in {0 none, 1 brightness, 2 contrast, 3 saturation, 4 hue,
 *                        5 to-gray} with factor[n]; contrast reads gray_sum[n] (of the image as it enters this stage).
 *                        in == out is allowed (pointwise).  ColorJitter = four stages in the image's random order.
 *   msfwsi_blur_sharpen: kind[n] in {0 copy, 1 Gaussian blur with ksize[n] <= 31 taps, 2 Sharpen with a 3x3 matrix};
 *                        taps [N][32] fp32 (the 1-D Gaussian taps, or the 3x3 matrix row-major); tmp fp32 [N][H][W][3]
 *                        scratch; in != out; H, W >= 16. */
int msfwsi_gray_sum(const unsigned char* img, int N, int H, int W, double* sums, void* stream);
int msfwsi_color_stage(const unsigned char* in, unsigned char* out, int N, int H, int W, const int* op, const double* factor,
                       const double* gray_sum, void* stream);
int msfwsi_blur_sharpen(const unsigned char* in, unsigned char* out, float* tmp, int N, int H, int W, const int* kind,
                        const int* ksize, const float* taps, void* stream);
/* inv[r][perm[r][k]] = k: jigsaw_reverse_idx = argsort(jigsaw_idx) (src/utils/data/bcss.py:172) */
int msfwsi_inverse_perm(const long* perm, long* inv, long rows, int K, void* stream);

/* performance knobs (never change results): key 0 = minimum grid (in 256x128 tiles) from which the conv
 * kernels switch from the 128x128 / 4-wave tile to the 256x128 / 8-wave tile; key 1 = 0 disables the pure-DMA
 * (buffer_load ... lds) conv kernel, key 2 = 0 the linear-addressing weight-gradient path, key 4 = grid size (in
 * 128x128 tiles) below which 128x64 tiles are used, key 5 = 0 disables the parity-class form of the stride-2 3x3 input
 * gradient, key 6 = 0 the 256x256 / 16-wave weight-gradient tile, key 9 = 0 the weights-stationary 3x3
 * kernel of the 64 -> 64 layers, key 10 = 0 their output-stationary weight-gradient kernel, key 11 = smallest
 * padded raster (positions) that kernel takes, key 12 = 0 the stationary stem kernels (forward and weight gradient), key 13 =
 * smallest padded raster the stem's weight-gradient kernel takes, key 16 = 0 the column-walk max-pool forward / backward (default 1; 0 = one thread per
 * window / pixel) (A/B measurements, tests).
 * key 17 = 0 runs the panel kernels (csrc/panel.hip) on compiler-counted waits instead of the hand-counted ones (same
 * arithmetic, bit-identical results: the A/B reference of tools/check_hand_waits.py's static audit).
 * key 18 = 0 keeps the panel kernels on 32-channel blocks where they would take the wide form (a wave owns 64 channels of
 * 64 rows: 128-byte row segments in the epilogue; k <= 128); same products in another order of the fp32 sums.
 * One knob trades speed for run-to-run reproducibility: key 15 = cap on the pixel splits of the gather weight-gradient
 * kernel (0 = none; 1 = each gradient tile summed by one workgroup in pixel order instead of fp32 atomics in arrival
 * order -- the results then differ from the default's by rounding, 4e-7, and are the same on every run). */
int msfwsi_set_tuning(int key, long value);
/* the current value of a switch (fixtures save it before they flip one, and restore exactly that) */
int msfwsi_get_tuning(int key, long* value);

/* library identification: returns the gfx target string the code objects were built for */
const char* msfwsi_target(void);
/* build identification: 16 hex digits, sha256 over every source file this binary was built from (the .hip and .h files under csrc/, this
 * header, the Makefile and its EXTRA flags).  The loader compares it with the digest of the sources beside it (a binary that
 * travelled without its sources' edits is rebuilt, whatever the file times say), and bench.py replays committed counter
 * summaries (profiles/r*_pmc.json) only when they were taken with this very build. */
const char* msfwsi_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* MSFWSI_HIP_H */
