"""ORACLE -- test infrastructure only (nothing under msf_wsi_amd/ may import it).

numpy restatement of the COLOUR part of the reference's per-sample batch construction (tools/ssl_train.py:176-201):
    albu.ColorJitter(0.4, 0.4, 0.4, 0.1, p=0.8), albu.ToGray(p=0.2),
    albu.OneOf([albu.GaussianBlur(blur_limit=[19, 23], sigma_limit=[0.1, 2.0]), albu.Sharpen()], p=0.5)
applied to uint8 RGB images -- the whole 1024x1024 tile for the target views (src/utils/data/bcss.py:166-170), the 224x224
crop for the context views (the list's order: crop, colour, flip, normalise).

PARITY UNPINNED.  albumentations and cv2 are third-party packages outside the reference tree and absent from this image
(SURVEY.md 8f row f3); what follows restates their PUBLISHED arithmetic for 8-bit images:
  * albumentations.augmentations.functional: adjust_brightness / contrast / saturation / hue _torchvision (look-up tables
    built in float64 and truncated to uint8; contrast around the mean of the 8-bit gray image; saturation as
    cv2.addWeighted with the gray image; hue as a look-up table on the H plane of cv2's 8-bit HSV image), to_gray,
    ColorJitter.apply's loop over a random order of the four adjustments;
  * cv2.cvtColor: RGB2GRAY in 14-bit fixed point (4899, 9617, 1868); RGB2HSV (8-bit, H in [0,180)) with the 12-bit
    division tables; HSV2RGB through the float formula, rounded to nearest-even;
  * cv2.GaussianBlur: separable, kernel exp(-(i-c)^2 / 2 sigma^2) normalised in float64, BORDER_REFLECT_101 -- restated
    in fp32 (rows first, one rounding to uint8 at the end); OpenCV's bit-exact 8-bit path quantises the kernel to 8
    fractional bits and is NOT reproduced;
  * albumentations.Sharpen: (1 - alpha) * identity + alpha * [[-1,-1,-1],[-1,8+lightness,-1],[-1,-1,-1]] through
    cv2.filter2D (correlation, BORDER_REFLECT_101, rounded to nearest-even, saturated).
No golden vector of the reference exists for this path; the known-answer checks in tests/test_augment.py are the
published constants (pure red -> gray 76, primary colours <-> HSV)."""
import numpy as np

OP_NONE, OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION, OP_HUE, OP_GRAY = 0, 1, 2, 3, 4, 5
FILT_NONE, FILT_BLUR, FILT_SHARPEN = 0, 1, 2


def rgb2gray_u8(img):
    """cv2.cvtColor(img, COLOR_RGB2GRAY), 8-bit: descale(R*4899 + G*9617 + B*1868, 14)"""
    v = img.astype(np.int64)
    return ((v[..., 0] * 4899 + v[..., 1] * 9617 + v[..., 2] * 1868 + (1 << 13)) >> 14).astype(np.uint8)


def _lut_u8(lut):
    return np.clip(lut, 0, 255).astype(np.uint8)  # truncation, as ndarray.astype does


def adjust_brightness(img, f):
    if f == 0:
        return np.zeros_like(img)
    if f == 1:
        return img
    return _lut_u8(np.arange(0, 256) * f)[img]


def adjust_contrast(img, f):
    if f == 1:
        return img
    mean = rgb2gray_u8(img).mean()  # float64 mean of the 8-bit gray image (exact: integer sum / count)
    if f == 0:
        return np.full_like(img, int(mean + 0.5))
    lut = np.arange(0, 256) * f
    lut = lut + mean * (1 - f)
    return _lut_u8(lut)[img]


def _round_u8(x):
    """saturate_cast<uchar>(cvRound(x)): nearest, ties to even"""
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


def adjust_saturation(img, f):
    if f == 1:
        return img
    gray = np.repeat(rgb2gray_u8(img)[..., None], 3, axis=-1)
    if f == 0:
        return gray
    a, b = np.float32(f), np.float32(1 - f)  # cv2.addWeighted: the 8-bit path works in float
    return _round_u8(img.astype(np.float32) * a + gray.astype(np.float32) * b)


_SDIV = np.zeros(256, dtype=np.int64)
_HDIV = np.zeros(256, dtype=np.int64)
_SDIV[1:] = np.rint((255 << 12) / (1.0 * np.arange(1, 256))).astype(np.int64)
_HDIV[1:] = np.rint((180 << 12) / (6.0 * np.arange(1, 256))).astype(np.int64)


def rgb2hsv_u8(img):
    """cv2.cvtColor(img, COLOR_RGB2HSV), 8-bit: H in [0,180), S and V in [0,255]; 12-bit fixed-point division tables"""
    v = img.astype(np.int64)
    r, g, b = v[..., 0], v[..., 1], v[..., 2]
    vmax = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = vmax - vmin
    s = (diff * _SDIV[vmax] + (1 << 11)) >> 12
    h = np.where(vmax == r, g - b, np.where(vmax == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * _HDIV[diff] + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return np.stack([h, s, vmax], axis=-1).astype(np.uint8)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])  # tab index of (b, g, r)


def hsv2rgb_u8(hsv):
    """cv2.cvtColor(hsv, COLOR_HSV2RGB), 8-bit through the float formula (fp32 operations in cv2's order)"""
    f32 = np.float32
    h = hsv[..., 0].astype(f32) * f32(6.0 / 180.0)
    s = hsv[..., 1].astype(f32) * f32(1.0 / 255.0)
    v = hsv[..., 2].astype(f32) * f32(1.0 / 255.0)
    sector = np.floor(h).astype(np.int64)
    fr = h - sector.astype(f32)
    sector = np.mod(sector, 6)
    one = f32(1.0)
    tab = np.stack([v, v * (one - s), v * (one - s * fr), v * (one - s * (one - fr))], axis=-1)
    idx = _SECTOR[sector]  # [..., 3] -> (b, g, r)
    bgr = np.take_along_axis(tab, idx, axis=-1)
    bgr = np.where((hsv[..., 1] == 0)[..., None], v[..., None], bgr)
    rgb = bgr[..., ::-1]
    return _round_u8(rgb * f32(255.0))


def adjust_hue(img, f):
    if f == 0:
        return img
    hsv = rgb2hsv_u8(img)
    lut = np.mod(np.arange(0, 256, dtype=np.int16) + 180 * f, 180).astype(np.uint8)
    hsv[..., 0] = lut[hsv[..., 0]]
    return hsv2rgb_u8(hsv)


def to_gray(img):
    return np.repeat(rgb2gray_u8(img)[..., None], 3, axis=-1)


_ADJUST = {OP_BRIGHTNESS: adjust_brightness, OP_CONTRAST: adjust_contrast, OP_SATURATION: adjust_saturation,
           OP_HUE: adjust_hue}


def color_jitter(img, order, factors):
    """ColorJitter.apply: the four adjustments in `order` (a permutation of OP_BRIGHTNESS..OP_HUE); factors[op]"""
    for op in order:
        img = _ADJUST[int(op)](img, float(factors[int(op)]))
    return img


def reflect101(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * n - 2 - i, i)


def gaussian_taps(ksize, sigma):
    """cv2.getGaussianKernel(ksize, sigma) (float64, sum 1), as fp32 taps"""
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)


def gaussian_blur(img, ksize, sigma):
    h, w, _ = img.shape
    taps = gaussian_taps(ksize, sigma)
    r = ksize // 2
    src = img.astype(np.float32)
    tmp = np.zeros_like(src)
    for i in range(ksize):  # rows first; fp32 multiply then add, taps in order
        tmp = tmp + taps[i] * src[:, reflect101(np.arange(w) + i - r, w)]
    out = np.zeros_like(src)
    for i in range(ksize):
        out = out + taps[i] * tmp[reflect101(np.arange(h) + i - r, h)]
    return _round_u8(out)


def sharpen_matrix(alpha, lightness):
    nochange = np.array([[0, 0, 0], [0, 1, 0], [0, 0, 0]], dtype=np.float64)
    effect = np.array([[-1, -1, -1], [-1, 8 + lightness, -1], [-1, -1, -1]], dtype=np.float64)
    return ((1 - alpha) * nochange + alpha * effect).astype(np.float32)


def sharpen(img, alpha, lightness):
    h, w, _ = img.shape
    m = sharpen_matrix(alpha, lightness)
    src = img.astype(np.float32)
    out = np.zeros_like(src)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            out = out + m[dy + 1, dx + 1] * src[reflect101(np.arange(h) + dy, h)][:, reflect101(np.arange(w) + dx, w)]
    return _round_u8(out)


def apply(img, dec):
    """one image through the reference's colour list; dec: dict of the per-image decisions
    {jitter: bool, order: [4], factors: {op: f}, gray: bool, filt: FILT_*, ksize, sigma, alpha, lightness}"""
    if dec["jitter"]:
        img = color_jitter(img, dec["order"], dec["factors"])
    if dec["gray"]:
        img = to_gray(img)
    if dec["filt"] == FILT_BLUR:
        img = gaussian_blur(img, dec["ksize"], dec["sigma"])
    elif dec["filt"] == FILT_SHARPEN:
        img = sharpen(img, dec["alpha"], dec["lightness"])
    return img
