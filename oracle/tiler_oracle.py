"""ORACLE -- test infrastructure only (nothing under msf_wsi_amd/ may import it).

numpy restatement of the geometric part of the reference's per-sample batch construction:
  * `blockshaped` -- src/utils/data/bcss.py:203-216 (same in paip.py / camelyon.py), restated verbatim in behaviour;
  * target_grid[jigsaw_idx], argsort(jigsaw_idx) -- bcss.py:171-176;
  * misc_aug's crop -> resize(224) -> HorizontalFlip -> Normalize -> ToTensorV2 -- tools/ssl_train.py:203-214.
PINNED: blockshaped / shuffle / flip / layout (plain numpy, checked here against the reference's own assertion
`target_grid.shape == (16, 256, 256, 3)` and its block order) and Normalize (albumentations.functional.normalize's
published fp32 operation order).  UNPINNED: the bilinear resize -- albumentations delegates to cv2.resize(INTER_LINEAR),
whose 8-bit path uses fixed-point coefficients; cv2 and albumentations are absent from this image, so the resize is
restated as fp32 bilinear interpolation with cv2's half-pixel convention, rounded to uint8 levels.  A crop box of exactly
224x224 makes the resize an exact copy: that case is fully pinned."""
import numpy as np


def blockshaped(arr, nrows, ncols):
    h, w, c = arr.shape
    assert h % nrows == 0 and w % ncols == 0
    return arr.reshape(h // nrows, nrows, -1, ncols, c).swapaxes(1, 2).reshape(-1, nrows, ncols, c)


def resize_bilinear_u8(img, size):
    """[h,w,3] uint8 -> [size,size,3] uint8-valued float32; cv2 half-pixel convention, fp32 lerps, round half to even"""
    h, w, _ = img.shape
    f32 = np.float32

    def coords(n_src, n_dst):
        sc = f32(n_src) / f32(n_dst)
        f = (np.arange(n_dst, dtype=f32) + f32(0.5)) * sc - f32(0.5)
        i0 = np.floor(f).astype(np.int64)
        a = (f - i0.astype(f32)).astype(f32)
        lo = i0 < 0
        i0[lo], a[lo] = 0, 0
        hi = i0 >= n_src - 1
        i0[hi], a[hi] = n_src - 1, 0
        i1 = np.minimum(i0 + 1, n_src - 1)
        return i0, i1, a

    y0, y1, ay = coords(h, size)
    x0, x1, ax = coords(w, size)
    im = img.astype(f32)
    ax_, ay_ = ax[None, :, None], ay[:, None, None]
    top = im[y0][:, x0] + ax_ * (im[y0][:, x1] - im[y0][:, x0])
    bot = im[y1][:, x0] + ax_ * (im[y1][:, x1] - im[y1][:, x0])
    v = top + ay_ * (bot - top)
    return np.clip(np.rint(v), 0, 255).astype(f32)


def normalize(img_f32, mean, std, max_pixel=255.0):
    """albumentations.functional.normalize: fp32 throughout"""
    mean = np.array(mean, dtype=np.float32) * np.float32(max_pixel)
    std = np.array(std, dtype=np.float32) * np.float32(max_pixel)
    denom = np.reciprocal(std, dtype=np.float32)
    out = img_f32.astype(np.float32).copy()
    out -= mean
    out *= denom
    return out


def view(img_u8, grid, perm, boxes, flips, mean, std, size=224):
    """one tile [H,W,3] uint8 -> [grid*grid, 3, size, size] float32"""
    h, w, _ = img_u8.shape
    blocks = blockshaped(img_u8, h // grid, w // grid)
    if perm is not None:
        blocks = blocks[np.asarray(perm)]
    out = []
    for k, blk in enumerate(blocks):
        x0, y0, cw, ch = [int(t) for t in boxes[k]]
        r = resize_bilinear_u8(blk[y0:y0 + ch, x0:x0 + cw], size)
        if flips is not None and flips[k]:
            r = r[:, ::-1]
        out.append(normalize(r, mean, std).transpose(2, 0, 1))
    return np.stack(out)
