// On-device tiling / normalising front end of the pre-train step (row f3 of SURVEY.md 8f): what the reference's
// DataLoader workers do per sample after the colour augmentations (src/utils/data/bcss.py:164-182):
//   target view : blockshaped(img, 256, 256) -> 16 blocks (bcss.py:203-216), shuffled by jigsaw_idx = randperm(16)
//                 (:171-176), each block through misc_aug = RandomResizedCrop(224) + HorizontalFlip + Normalize +
//                 ToTensorV2 (tools/ssl_train.py:203-214)
//   context view: RandomResizedCrop(224) of the whole 1024x1024 tile + HorizontalFlip + Normalize + ToTensorV2 (:176-196)
//   jigsaw_reverse_idx = argsort(jigsaw_idx) (bcss.py:172)
// The random decisions (crop boxes, flips, permutations) are inputs; the colour augmentations (ColorJitter, ToGray, blur,
// sharpen) stay outside (albumentations arithmetic, unpinned).  One thread per output pixel, 3 channels; reads are
// gathered from the uint8 HWC tile, writes are coalesced fp32 CHW planes.  HBM-bound: 3 B in (per bilinear tap) + 12 B
// out per pixel.
// Arithmetic: crop + bilinear resize with the half-pixel convention of cv2.resize(INTER_LINEAR)
// (src = (dst + 0.5) * scale - 0.5, clamped) evaluated EXACTLY in integers -- coordinate split and the 4-tap blend --
// and rounded half-to-even to the uint8 level (the reference's
// intermediate image is uint8) -- cv2's fixed-point coefficients are NOT reproduced (cv2 / albumentations are absent:
// resize parity unpinned; a box of exactly 224x224 is an exact copy); then albumentations' Normalize in its own fp32
// operation order: img = float(img); img -= mean*255; img *= 1/(std*255).
#include "common.h"
#include "../../include/msfwsi_hip.h"

namespace {

struct TilerParams {
    const unsigned char* img;  // [B][H][W][3] uint8
    float* out;                // [B][K][3][S][S] fp32
    unsigned char* out_u8;     // instead: [B][K][S][S][3] uint8, the resized crop before flip / Normalize (msfwsi_tile_crops_u8)
    const long* perm;          // [B][K] block order (jigsaw_idx), nullable (identity)
    const int* boxes;          // [B][K][4] = x0, y0, w, h inside the block
    const unsigned char* flips;  // [B][K], nullable
    float mean255[3], denom[3];
    int B, H, W, K, grid, bh, bw, S;  // grid x grid blocks of bh x bw pixels
};

__global__ void tiler_kernel(const TilerParams p) {
    const long total = (long)p.B * p.K * p.S * p.S;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % p.S);
        const int y = (int)((i / p.S) % p.S);
        const long bk = i / ((long)p.S * p.S);
        const int k = (int)(bk % p.K);
        const int b = (int)(bk / p.K);
        const int src_blk = p.perm != nullptr ? (int)p.perm[bk] : k;  // target_grid[jigsaw_idx]: output k <- block perm[k]
        const int by = (src_blk / p.grid) * p.bh, bx = (src_blk % p.grid) * p.bw;
        const int* box = p.boxes + bk * 4;
        const int x0 = box[0], y0 = box[1], cw = box[2], ch = box[3];
        const int xo = (p.flips != nullptr && p.flips[bk]) ? p.S - 1 - x : x;  // HorizontalFlip AFTER the resize
        // cv2 half-pixel mapping into the crop, clamped to its edge
        // Source coordinate of cv2's half-pixel convention, src = (dst + 0.5) * c / S - 0.5 = ((2 dst + 1) c - S) / (2 S),
        // split EXACTLY with integers into its floor and a remainder in [0, 2S); the interpolation weight is
        // remainder / 2S (kept as the integer remainder: the blend below is integer arithmetic, bit-identical everywhere)
        const int twoS = 2 * p.S;
        const int nx = (2 * xo + 1) * cw - p.S, ny = (2 * y + 1) * ch - p.S;
        int ix = nx >= 0 ? nx / twoS : -((-nx + twoS - 1) / twoS);
        int iy = ny >= 0 ? ny / twoS : -((-ny + twoS - 1) / twoS);
        int rx = nx - ix * twoS, ry = ny - iy * twoS;  // weights rx / 2S, ry / 2S
        if (ix < 0) { ix = 0; rx = 0; }
        if (iy < 0) { iy = 0; ry = 0; }
        int ix1 = ix + 1, iy1 = iy + 1;
        if (ix1 >= cw) { ix1 = cw - 1; if (ix >= cw - 1) { ix = cw - 1; rx = 0; } }
        if (iy1 >= ch) { iy1 = ch - 1; if (iy >= ch - 1) { iy = ch - 1; ry = 0; } }
        const unsigned char* r0 = p.img + (((long)b * p.H + by + y0 + iy) * p.W + bx + x0) * 3;
        const unsigned char* r1 = p.img + (((long)b * p.H + by + y0 + iy1) * p.W + bx + x0) * 3;
        float* o = p.out + bk * 3 * p.S * p.S + (long)y * p.S + x;
        unsigned char* o8 = p.out_u8 + ((bk * p.S + y) * p.S + x) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // the blend in exact integers (like cv2's 8-bit path, which is fixed-point too):
            //   v = [(v00 (2S-rx) + v01 rx) (2S-ry) + (v10 (2S-rx) + v11 rx) ry] / (2S)^2, rounded half to even
            const int top = (int)r0[ix * 3 + c] * (twoS - rx) + (int)r0[ix1 * 3 + c] * rx;
            const int bot = (int)r1[ix * 3 + c] * (twoS - rx) + (int)r1[ix1 * 3 + c] * rx;
            const long num = (long)top * (twoS - ry) + (long)bot * ry;
            const long den = (long)twoS * twoS;
            long q = num / den;
            const long rem2 = 2 * (num - q * den);
            if (rem2 > den || (rem2 == den && (q & 1))) ++q;
            if (p.out_u8 != nullptr) {
                o8[c] = (unsigned char)q;
                continue;
            }
            const float v = (float)q;  // 0 .. 255: the reference's uint8 intermediate image
            o[(long)c * p.S * p.S] = __fmul_rn(__fsub_rn(v, p.mean255[c]), p.denom[c]);
        }
    }
}

__global__ void inverse_perm_kernel(const long* __restrict__ perm, long* __restrict__ inv, long rows, int K) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= rows * K) return;
    const long r = i / K;
    inv[r * K + perm[i]] = i - r * K;  // argsort of a permutation is its inverse
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int msfwsi_tile_views(const unsigned char* img, int B, int H, int W, int grid, const long* perm,
                                 const int* boxes, const unsigned char* flips, const float* mean, const float* std_,
                                 float max_pixel, int S, float* out, void* stream) {
    MSFWSI_CHECK_ARG(img && boxes && mean && std_ && out && B > 0 && H > 0 && W > 0 && grid > 0 && S > 0);
    MSFWSI_CHECK_ARG(H % grid == 0 && W % grid == 0);
    TilerParams p{};
    p.img = img; p.out = out; p.perm = perm; p.boxes = boxes; p.flips = flips;
    for (int c = 0; c < 3; ++c) {
        // albumentations.functional.normalize: mean32 * max_pixel, reciprocal(std32 * max_pixel) -- all fp32
        p.mean255[c] = mean[c] * max_pixel;
        p.denom[c] = 1.0f / (std_[c] * max_pixel);  // host code: IEEE fp32 division, as np.reciprocal(float32)
    }
    p.B = B; p.H = H; p.W = W; p.K = grid * grid; p.grid = grid; p.bh = H / grid; p.bw = W / grid; p.S = S;
    const long total = (long)B * p.K * S * S;
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(tiler_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), p);
    return msfwsi_launch_status();
}

// the crop + resize alone, uint8 HWC out: the context view's colour augmentations sit between the crop and the flip
// (tools/ssl_train.py:176-196: RandomResizedCrop, ColorJitter, ToGray, blur | sharpen, HorizontalFlip, Normalize)
extern "C" int msfwsi_tile_crops_u8(const unsigned char* img, int B, int H, int W, int grid, const long* perm,
                                    const int* boxes, int S, unsigned char* out, void* stream) {
    MSFWSI_CHECK_ARG(img && boxes && out && B > 0 && H > 0 && W > 0 && grid > 0 && S > 0);
    MSFWSI_CHECK_ARG(H % grid == 0 && W % grid == 0);
    TilerParams p{};
    p.img = img; p.out_u8 = out; p.perm = perm; p.boxes = boxes;
    p.B = B; p.H = H; p.W = W; p.K = grid * grid; p.grid = grid; p.bh = H / grid; p.bw = W / grid; p.S = S;
    const long total = (long)B * p.K * S * S;
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(tiler_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), p);
    return msfwsi_launch_status();
}

extern "C" int msfwsi_inverse_perm(const long* perm, long* inv, long rows, int K, void* stream) {
    MSFWSI_CHECK_ARG(perm && inv && rows > 0 && K > 0);
    hipLaunchKernelGGL(inverse_perm_kernel, dim3((unsigned)((rows * K + 255) / 256)), dim3(256), 0, ST(stream), perm, inv,
                       rows, K);
    return msfwsi_launch_status();
}
