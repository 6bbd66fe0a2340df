"""Row f3, colour part (tools/ssl_train.py:176-201: ColorJitter, ToGray, GaussianBlur | Sharpen): the numpy oracle against
the published constants of the arithmetic it restates (CPU; PARITY UNPINNED -- albumentations / cv2 are absent, no vector of
the reference exists for this path), the HIP kernels against the oracle bit for bit (GPU), and the pipeline in the
reference's order."""
import numpy as np
import pytest
import torch


def _img(n, h, w, seed=0):
    """smooth colourful test images with saturated corners (uint8 [n,h,w,3])"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    out = []
    for i in range(n):
        ph = rng.uniform(0, 6.28, 3)
        fr = rng.uniform(0.02, 0.3, 3)
        base = [127 + 120 * np.sin(fr[c] * xx + ph[c]) * np.cos(fr[(c + 1) % 3] * yy) for c in range(3)]
        im = np.stack(base, -1) + rng.normal(0, 12, (h, w, 3))
        im[:4, :4] = 255
        im[-4:, -4:] = 0
        im[:4, -4:] = (255, 0, 0)
        out.append(np.clip(im, 0, 255).astype(np.uint8))
    return np.stack(out)


def test_oracle_known_answers():
    from oracle import augment_oracle as ao

    px = lambda *rgb: np.array([[rgb]], dtype=np.uint8)
    # cv2 RGB2GRAY (0.299, 0.587, 0.114 in 14-bit fixed point): the documented values of the primaries
    assert [int(ao.rgb2gray_u8(px(*c))[0, 0]) for c in ((255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 255), (0, 0, 0))] \
        == [76, 150, 29, 255, 0]
    # cv2 8-bit HSV: H = degrees / 2, S and V in 0..255
    for rgb, hsv in (((255, 0, 0), (0, 255, 255)), ((0, 255, 0), (60, 255, 255)), ((0, 0, 255), (120, 255, 255)),
                     ((255, 255, 0), (30, 255, 255)), ((0, 255, 255), (90, 255, 255)), ((255, 0, 255), (150, 255, 255)),
                     ((128, 128, 128), (0, 0, 128)), ((0, 0, 0), (0, 0, 0))):
        assert tuple(int(t) for t in ao.rgb2hsv_u8(px(*rgb))[0, 0]) == hsv, rgb
        assert tuple(int(t) for t in ao.hsv2rgb_u8(np.array([[hsv]], dtype=np.uint8))[0, 0]) == rgb, hsv
    img = _img(1, 48, 40)[0]
    # identities albumentations short-cuts (factor 1, hue 0) hold for the general formulas too, except hue
    assert np.array_equal(ao.adjust_brightness(img, 1.0 + 1e-12 - 1e-12), img)
    assert np.array_equal(ao._lut_u8(np.arange(256) * 1.0)[img], img)
    assert np.array_equal(ao.adjust_hue(img, 0), img)
    g = ao.to_gray(img)
    assert np.array_equal(ao.to_gray(g), g) and np.array_equal(ao.adjust_saturation(img, 0), g)
    # a hue shift by half a turn and back returns within the 8-bit HSV quantisation
    back = ao.adjust_hue(ao.adjust_hue(img, 0.5), 0.5)
    assert np.abs(back.astype(int) - img.astype(int)).max() <= 6
    # contrast 0 -> the rounded gray mean everywhere; brightness 2 saturates
    m = int(ao.rgb2gray_u8(img).mean() + 0.5)
    assert (ao.adjust_contrast(img, 0) == m).all() and ao.adjust_brightness(img, 2.0).max() == 255
    # Gaussian taps sum to one; a constant image is a fixed point of both filters; reflect-101 indices
    assert abs(float(ao.gaussian_taps(23, 2.0).astype(np.float64).sum()) - 1) < 1e-6
    const = np.full((40, 36, 3), 93, np.uint8)
    assert np.array_equal(ao.gaussian_blur(const, 19, 0.7), const)
    assert abs(float(ao.sharpen_matrix(0.3, 0.75).sum()) - (0.7 + 0.3 * 0.75)) < 1e-6
    assert list(ao.reflect101(np.array([-2, -1, 0, 5, 6, 7]), 6)) == [2, 1, 0, 5, 4, 3]
    # blur smooths: total variation falls; sharpen raises it
    tv = lambda a: float(np.abs(np.diff(a.astype(float), axis=1)).sum())
    assert tv(ao.gaussian_blur(img, 19, 2.0)) < 0.5 * tv(img) < tv(ao.sharpen(img, 0.5, 1.0))


def test_decision_law_matches_the_reference_lists():
    """DeviceColorAug.decisions (host, no GPU): the rates and ranges of tools/ssl_train.py:176-201 -- ColorJitter(0.4, 0.4,
    0.4, 0.1, p=0.8) with a random order of its four adjustments, ToGray(p=0.2), OneOf([GaussianBlur([19,23], [0.1,2.0]),
    Sharpen()], p=0.5) -- and albumentations' kernel-size rule (randrange(19, 24); an even draw k becomes (k+1) % 24)"""
    from msf_wsi_amd.augment import DeviceColorAug, gaussian_taps, sharpen_matrix
    from oracle import augment_oracle as ao

    big = DeviceColorAug().decisions(40000, torch.Generator().manual_seed(1))
    near = lambda x, p: abs(float(x.float().mean()) - p) < 0.01
    assert near(big.jitter, 0.8) and near(big.gray, 0.2) and near(big.filt == 1, 0.25) and near(big.filt == 2, 0.25)
    assert near(big.ksize == 19, 0.2) and near(big.ksize == 21, 0.4) and near(big.ksize == 23, 0.4)
    for op, (lo, hi) in ((1, (0.6, 1.4)), (2, (0.6, 1.4)), (3, (0.6, 1.4)), (4, (-0.1, 0.1))):
        f = big.factors[:, op]
        assert float(f.min()) >= lo and float(f.max()) <= hi and abs(float(f.mean()) - (lo + hi) / 2) < 0.01
    assert float(big.sigma.min()) >= 0.1 and float(big.sigma.max()) <= 2.0
    assert float(big.alpha.min()) >= 0.2 and float(big.alpha.max()) <= 0.5
    assert float(big.lightness.min()) >= 0.5 and float(big.lightness.max()) <= 1.0
    assert all(sorted(o) == [1, 2, 3, 4] for o in big.order[:200].tolist())
    firsts = torch.bincount(big.order[:, 0].long(), minlength=5)[1:].float() / big.order.shape[0]
    assert float((firsts - 0.25).abs().max()) < 0.01  # every adjustment leads a quarter of the orders
    # the host-side tap tables are the oracle's, bit for bit
    for ks, sg in ((19, 0.1), (21, 1.0), (23, 2.0)):
        assert np.array_equal(gaussian_taps(ks, sg).numpy(), ao.gaussian_taps(ks, sg))
    assert np.array_equal(sharpen_matrix(0.37, 0.81).numpy(), ao.sharpen_matrix(0.37, 0.81))


def _decisions_for_oracle(dec, n):
    from oracle import augment_oracle as ao

    return {"jitter": bool(dec.jitter[n]), "order": [int(t) for t in dec.order[n]],
            "factors": {op: float(dec.factors[n, op]) for op in (1, 2, 3, 4)}, "gray": bool(dec.gray[n]),
            "filt": int(dec.filt[n]), "ksize": int(dec.ksize[n]), "sigma": float(dec.sigma[n]),
            "alpha": float(dec.alpha[n]), "lightness": float(dec.lightness[n])}


@pytest.mark.gpu
@pytest.mark.parametrize("op", [1, 2, 3, 4, 5], ids=["brightness", "contrast", "saturation", "hue", "gray"])
def test_color_stage_matches_oracle(hip_lib, op):
    from msf_wsi_amd import kernels as kn
    from oracle import augment_oracle as ao

    img = _img(6, 72, 56, seed=op)
    lo, hi = ((-0.1, 0.1) if op == 4 else (0.6, 1.4))
    f = np.random.default_rng(op).uniform(lo, hi, 6)
    f[4], f[5] = (0.0, -0.1) if op == 4 else (0.0, 1.0)      # the special-cased factors
    ops = torch.full((6,), op, dtype=torch.int32)
    ops[3] = 0                                                # an image this stage skips
    d = torch.from_numpy(img).cuda()
    sums = kn.gray_sum(d)
    assert torch.equal(sums.cpu(), torch.tensor([float(ao.rgb2gray_u8(im).astype(np.int64).sum()) for im in img],
                                                dtype=torch.float64))
    out = kn.color_stage(d.clone(), ops.cuda(), torch.from_numpy(f).cuda(), sums).cpu().numpy()
    fn = {1: ao.adjust_brightness, 2: ao.adjust_contrast, 3: ao.adjust_saturation, 4: ao.adjust_hue}
    for n in range(6):
        want = img[n] if n == 3 else (ao.to_gray(img[n]) if op == 5 else fn[op](img[n], float(f[n])))
        assert np.array_equal(out[n], want), (n, float(f[n]), int(np.abs(out[n].astype(int) - want.astype(int)).max()))


@pytest.mark.gpu
def test_blur_and_sharpen_match_oracle(hip_lib):
    from msf_wsi_amd import augment as aug, kernels as kn
    from oracle import augment_oracle as ao

    img = _img(7, 64, 80, seed=9)
    kind = torch.tensor([1, 2, 0, 1, 1, 2, 1], dtype=torch.int32)
    ksize = torch.tensor([19, 0, 0, 21, 23, 0, 23], dtype=torch.int32)
    sigma = [0.1, 0, 0, 1.0, 2.0, 0, 0.55]
    alpha, light = [0, 0.2, 0, 0, 0, 0.5, 0], [0, 0.5, 0, 0, 0, 1.0, 0]
    taps = torch.zeros(7, aug.MAX_TAPS)
    for n in range(7):
        if int(kind[n]) == 1:
            taps[n, :int(ksize[n])] = aug.gaussian_taps(int(ksize[n]), sigma[n])
            assert np.array_equal(taps[n, :int(ksize[n])].numpy(), ao.gaussian_taps(int(ksize[n]), sigma[n]))
        elif int(kind[n]) == 2:
            taps[n, :9] = aug.sharpen_matrix(alpha[n], light[n]).reshape(-1)
            assert np.array_equal(taps[n, :9].numpy().reshape(3, 3), ao.sharpen_matrix(alpha[n], light[n]))
    out = kn.blur_sharpen(torch.from_numpy(img).cuda(), kind.cuda(), ksize.cuda(), taps.cuda()).cpu().numpy()
    for n in range(7):
        want = (ao.gaussian_blur(img[n], int(ksize[n]), sigma[n]) if int(kind[n]) == 1 else
                ao.sharpen(img[n], alpha[n], light[n]) if int(kind[n]) == 2 else img[n])
        assert np.array_equal(out[n], want), (n, int(np.abs(out[n].astype(int) - want.astype(int)).max()))


@pytest.mark.gpu
def test_device_color_aug_matches_oracle_pipeline(hip_lib):
    """ColorJitter (each image its own order and factors) -> ToGray -> OneOf(blur, sharpen) as DeviceColorAug chains the
    kernels == the oracle's per-image pipeline, bit for bit; the decision rates are the reference's"""
    from msf_wsi_amd.augment import DeviceColorAug
    from oracle import augment_oracle as ao

    N = 24
    img = _img(N, 64, 64, seed=4)
    ca = DeviceColorAug(chunk=7)
    dec = ca.decisions(N, torch.Generator().manual_seed(11))
    assert dec.jitter.any() and (~dec.jitter).any() and dec.gray.any() and set(dec.filt.tolist()) == {0, 1, 2}
    assert set(dec.ksize.tolist()) <= {19, 21, 23} and all(sorted(o) == [1, 2, 3, 4] for o in dec.order.tolist())
    src = torch.from_numpy(img).cuda()
    out = ca.apply(src, dec).cpu().numpy()
    assert torch.equal(src.cpu(), torch.from_numpy(img))  # the input is left alone
    for n in range(N):
        want = ao.apply(img[n], _decisions_for_oracle(dec, n))
        assert np.array_equal(out[n], want), (n, _decisions_for_oracle(dec, n))
    big = ca.decisions(20000, torch.Generator().manual_seed(1))
    assert abs(float(big.jitter.float().mean()) - 0.8) < 0.02 and abs(float(big.gray.float().mean()) - 0.2) < 0.02
    assert abs(float((big.filt == 1).float().mean()) - 0.25) < 0.02 and abs(float((big.filt == 2).float().mean()) - 0.25) < 0.02
    assert abs(float((big.ksize == 19).float().mean()) - 0.2) < 0.02  # randrange(19, 24): 20 -> 21, 22 -> 23


@pytest.mark.gpu
def test_tiler_batch_with_colour_follows_the_reference_order(hip_lib):
    """DeviceTiler.batch(color=...): target views = colour list on the WHOLE tile, then split / shuffle / crop; context
    views = crop, colour list, flip, Normalize -- rebuilt here from the oracles with the same decisions"""
    from msf_wsi_amd import data
    from msf_wsi_amd.augment import DeviceColorAug
    from oracle import augment_oracle as ao, tiler_oracle as to

    B, hw, size = 2, 256, 32
    img = _img(B, hw, hw, seed=21)
    tiles = torch.from_numpy(img).cuda()
    tiler, ca = data.DeviceTiler(scale=4, size=size), DeviceColorAug()
    (c1, c2), (t1, t2), idx = tiler.batch((tiles, tiles), (tiles, tiles), torch.Generator().manual_seed(5), color=ca)
    assert c1.shape == (B, 3, size, size) and t1.shape == (B * 16, 3, size, size) and idx[0].shape == (B, 16)
    # replay the host decisions in the order batch() draws them
    gen = torch.Generator().manual_seed(5)
    for v, (cv, tv) in enumerate(((c1, t1), (c2, t2))):
        perm = torch.stack([torch.randperm(16, generator=gen) for _ in range(B)])
        cb, cf = tiler._decisions(B, 1, hw, hw, gen)
        tb, tf = tiler._decisions(B, 16, hw // 4, hw // 4, gen)
        tdec = ca.decisions(B, gen)
        cdec = ca.decisions(B, gen)
        for b in range(B):
            tile_aug = ao.apply(img[b], _decisions_for_oracle(tdec, b))
            want_t = to.view(tile_aug, 4, perm[b].numpy(), tb[b].numpy(), tf[b].numpy(), data.MEAN, data.STD, size)
            assert np.array_equal(tv.view(B, 16, 3, size, size)[b].cpu().numpy(), want_t)
            x0, y0, cw, ch = [int(t) for t in cb[b, 0]]
            crop = to.resize_bilinear_u8(img[b][y0:y0 + ch, x0:x0 + cw], size).astype(np.uint8)
            crop = ao.apply(crop, _decisions_for_oracle(cdec, b))
            if cf[b, 0]:
                crop = crop[:, ::-1]
            want_c = to.normalize(crop.astype(np.float32), data.MEAN, data.STD).transpose(2, 0, 1)
            assert np.array_equal(cv[b].cpu().numpy(), want_c)
