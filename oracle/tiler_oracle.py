"""ORACLE -- test infrastructure only (nothing under msf_wsi_amd/ may import it).

numpy restatement of the geometric part of the reference's per-sample batch construction:
  * `blockshaped` -- src/utils/data/bcss.py:203-216 (same in paip.py / camelyon.py), restated verbatim in behaviour;
  * target_grid[jigsaw_idx], argsort(jigsaw_idx) -- bcss.py:171-176;
  * misc_aug's crop -> resize(224) -> HorizontalFlip -> Normalize -> ToTensorV2 -- tools/ssl_train.py:203-214.
PINNED: blockshaped / shuffle / flip / layout (plain numpy, checked here against the reference's own assertion
`target_grid.shape == (16, 256, 256, 3)` and its block order) and Normalize (albumentations.functional.normalize's
published fp32 operation order).  UNPINNED: the bilinear resize -- albumentations delegates to cv2.resize(INTER_LINEAR),
whose 8-bit path uses fixed-point coefficients (11-bit weights, two rounded passes); cv2 and albumentations are absent
from this image, so the resize is restated as EXACT bilinear interpolation with cv2's half-pixel convention (integer
arithmetic, one rounding half-to-even to the uint8 level): it can differ from cv2 by one level on a fraction of pixels.  A crop box of exactly
224x224 makes the resize an exact copy: that case is fully pinned."""
import numpy as np


def blockshaped(arr, nrows, ncols):
    h, w, c = arr.shape
    assert h % nrows == 0 and w % ncols == 0
    return arr.reshape(h // nrows, nrows, -1, ncols, c).swapaxes(1, 2).reshape(-1, nrows, ncols, c)


def resize_bilinear_u8(img, size):
    """[h,w,3] uint8 -> [size,size,3] uint8-valued float32.  cv2 half-pixel convention src = (dst + 0.5) n_src / n_dst - 0.5
    = ((2 dst + 1) n_src - n_dst) / (2 n_dst), split exactly into floor and remainder r in [0, 2 n_dst); the 4-tap blend
    with weights r / (2 n_dst) is evaluated in exact integers and rounded half to even"""
    h, w, _ = img.shape
    two = 2 * size

    def coords(n_src):
        num = (2 * np.arange(size, dtype=np.int64) + 1) * n_src - size
        i0 = np.floor_divide(num, two)
        r = num - i0 * two
        lo = i0 < 0
        i0[lo], r[lo] = 0, 0
        hi = i0 >= n_src - 1
        i0[hi], r[hi] = n_src - 1, 0
        i1 = np.minimum(i0 + 1, n_src - 1)
        return i0, i1, r

    y0, y1, ry = coords(h)
    x0, x1, rx = coords(w)
    im = img.astype(np.int64)
    rx_, ry_ = rx[None, :, None], ry[:, None, None]
    top = im[y0][:, x0] * (two - rx_) + im[y0][:, x1] * rx_
    bot = im[y1][:, x0] * (two - rx_) + im[y1][:, x1] * rx_
    num = top * (two - ry_) + bot * ry_
    den = two * two
    q = num // den
    rem2 = 2 * (num - q * den)
    q = q + ((rem2 > den) | ((rem2 == den) & (q % 2 == 1)))
    return q.astype(np.float32)


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
