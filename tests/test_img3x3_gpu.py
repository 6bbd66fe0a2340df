"""Image-stationary 3x3 kernels of the deep layers (csrc/img3x3.hip) on a real MI355X, through the C ABI: conv2 of the
Bottlenecks of layer1 / layer2 / layer3 (reference src/models/resnet.py:25-28,125-128) at 56x56x64, 28x28x128 and 14x14x256 against a plain
PyTorch fp64 CPU convolution of the same seeded, storage-rounded operands (bf16 <= 1.5e-2, fp16 <= 2e-3 rel-L2, the bounds of
test_kernels_gpu.py), and against the gather kernel (msfwsi_conv_fwd / msfwsi_conv_dgrad) the engine ran before it."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16]
# N, H (= W), channels: the two served geometries; N = 3 / 5 leaves the last wave of workgroups ragged
GEOMS = [(3, 14, 256), (2, 28, 128), (5, 14, 256), (2, 56, 64), (4, 56, 64)]  # (4 x 14 bands: the XCD-ordered band map)


def tol(dt):
    return {torch.bfloat16: 1.5e-2, torch.float16: 2e-3}[dt]


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def rnd(shape, dt, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(dt)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("pro", [False, True])
def test_img3x3_fwd(hip_lib, dt, geom, pro):
    from msf_wsi_amd import kernels as kn

    N, H, Cn = geom
    g = torch.Generator().manual_seed(31)
    x = rnd((N, Cn, H, H), dt, g)
    w = rnd((Cn, Cn, 3, 3), dt, g, 1.0 / math.sqrt(Cn * 9))
    sc, sh = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.3 + 0.2  # relu(shift) != 0: padding must stay 0
    d = kn.conv_desc(dt, N, H, H, Cn, Cn, 3, 3, 1, 1)
    assert kn.img3x3_supported(d)
    a = F.relu(x.float() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).to(dt) if pro else x
    ref = F.conv2d(a.double(), w.double(), None, stride=1, padding=1)
    wd = nhwc(w).cuda()
    wpk = kn.img3x3_pack_weights(wd, torch.empty_like(wd), False)
    y = torch.empty(N, H, H, Cn, dtype=dt, device="cuda")
    stats = kn.new_stats(Cn)
    assert kn.img3x3_fwd(d, nhwc(x).cuda(), wpk, y, stats=stats, pro=(sc.cuda(), sh.cuda()) if pro else None)
    torch.cuda.synchronize()
    assert rel(y.float().cpu().permute(0, 3, 1, 2), ref) < tol(dt)
    s = stats.sum(0).cpu()
    yy = y.double().cpu().reshape(-1, Cn)
    assert torch.allclose(s[0], yy.sum(0), rtol=1e-5, atol=1e-4)
    assert torch.allclose(s[1], (yy * yy).sum(0), rtol=1e-5, atol=1e-4)
    # the gather kernel on the same operands: the same fp32 MFMA sums in another order
    y2 = torch.empty_like(y)
    kn.conv_fwd(d, nhwc(a).cuda(), wd, y2)
    torch.cuda.synchronize()
    assert rel(y, y2) < tol(dt) / 4


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("gate", [False, True])
@pytest.mark.parametrize("bnbwd", [False, True])
def test_img3x3_dgrad(hip_lib, dt, geom, gate, bnbwd):
    from msf_wsi_amd import kernels as kn

    N, H, Cn = geom
    g = torch.Generator().manual_seed(32)
    w = rnd((Cn, Cn, 3, 3), dt, g, 1.0 / math.sqrt(Cn * 9))
    dy = rnd((N, Cn, H, H), dt, g)
    c2 = rnd((N, Cn, H, H), dt, g)          # the raw output whose BatchNorm backward is fused
    c1 = rnd((N, Cn, H, H), dt, g)          # the raw input whose BatchNorm + ReLU gates the result
    k1, k2, k3 = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.1, torch.randn(Cn, generator=g) * 0.01
    sc, sh = torch.rand(Cn, generator=g) - 0.3, torch.randn(Cn, generator=g) * 0.3
    d = kn.conv_desc(dt, N, H, H, Cn, Cn, 3, 3, 1, 1)
    dyd, wd = nhwc(dy).cuda(), nhwc(w).cuda()
    wpk = kn.img3x3_pack_weights(wd, torch.empty_like(wd), True)
    dc_ref = dy
    kw = {}
    dc = None
    if bnbwd:
        # msfwsi_bn_bwd_apply's arithmetic: fp32 fma chain, rounded to the storage type
        v = lambda t: t.view(1, -1, 1, 1)
        dc_ref = torch.addcmul(torch.addcmul(v(k3), v(k2), c2.float()), v(k1), dy.float()).to(dt)
        dc = torch.empty(N, H, H, Cn, dtype=dt, device="cuda")
        kw = dict(bnbwd=(nhwc(c2).cuda(), k1.cuda(), k2.cuda(), k3.cuda()), dc_out=dc)
    ref = torch.nn.grad.conv2d_input((N, Cn, H, H), w.double(), dc_ref.double(), stride=1, padding=1)
    if gate:
        ref = ref * ((c1.float() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) > 0)
        sums = kn.new_stats(Cn)
        c1d = nhwc(c1).cuda()
        act = torch.empty(N, H, H, Cn, dtype=dt, device="cuda")
        kw.update(mask=(c1d, sc.cuda(), sh.cuda()), sums=sums, act_out=act)
    dx = torch.empty(N, H, H, Cn, dtype=dt, device="cuda")
    assert kn.img3x3_dgrad(d, dyd, wpk, dx, **kw)
    torch.cuda.synchronize()
    assert rel(dx.float().cpu().permute(0, 3, 1, 2), ref) < tol(dt) * 2
    if bnbwd:
        # the written-back operand is what msfwsi_bn_bwd_apply writes (one fma association may differ by an ulp of fp32
        # before the rounding: compare through the rounding, allowing the rare tie)
        got = dc.float().cpu().permute(0, 3, 1, 2)
        want = dc_ref.float()
        assert (got != want).float().mean().item() < 2e-3
        assert rel(got, want) < 1e-3
    if gate:
        s = sums.sum(0).cpu()
        gd = dx.double().cpu().reshape(-1, Cn)
        assert torch.allclose(s[0], gd.sum(0), rtol=1e-5, atol=1e-4)
        assert torch.allclose(s[1], (gd * c1d.double().cpu().reshape(-1, Cn)).sum(0), rtol=1e-5, atol=1e-4)
        # the by-product activation is bit for bit what the stand-alone pass writes
        want = torch.empty_like(act)
        kn.bn_act(c1d, sc.cuda(), sh.cuda(), want, relu=True)
        torch.cuda.synchronize()
        assert torch.equal(act, want)
    # the gather kernel on the same (materialised) gradient operand
    dx2 = torch.empty_like(dx)
    kw2 = dict(mask=kw["mask"], sums=kn.new_stats(Cn)) if gate else {}
    kn.conv_dgrad(d, nhwc(dc_ref).cuda(), wd, dx2, **kw2)
    torch.cuda.synchronize()
    assert rel(dx, dx2) < tol(dt) / 2


def test_img3x3_in_place_operand_only_where_a_workgroup_owns_the_image(hip_lib):
    """dc_out may alias dy at 14x14 (one workgroup per image); at 28x28 the bands read each other's halo rows: refused"""
    from msf_wsi_amd import kernels as kn
    from msf_wsi_amd._lib import MsfwsiHipError

    dt = torch.bfloat16
    g = torch.Generator().manual_seed(33)
    for H, Cn, ok in ((14, 256, True), (28, 128, False), (56, 64, False)):
        N = 2
        d = kn.conv_desc(dt, N, H, H, Cn, Cn, 3, 3, 1, 1)
        w = rnd((Cn, 3, 3, Cn), dt, g, 1.0 / math.sqrt(Cn * 9)).cuda()
        wpk = kn.img3x3_pack_weights(w, torch.empty_like(w), True)
        dy = rnd((N, H, H, Cn), dt, g).cuda()
        c2 = rnd((N, H, H, Cn), dt, g).cuda()
        k = [torch.rand(Cn, generator=g).cuda() for _ in range(3)]
        dx = torch.empty_like(dy)
        if not ok:
            with pytest.raises(MsfwsiHipError):
                kn.img3x3_dgrad(d, dy, wpk, dx, bnbwd=(c2, *k), dc_out=dy)
            continue
        dc = torch.empty_like(dy)
        dx2 = torch.empty_like(dy)
        assert kn.img3x3_dgrad(d, dy, wpk, dx2, bnbwd=(c2, *k), dc_out=dc)
        assert kn.img3x3_dgrad(d, dy, wpk, dx, bnbwd=(c2, *k), dc_out=dy)
        torch.cuda.synchronize()
        assert torch.equal(dy, dc) and torch.equal(dx, dx2)


def test_img3x3_other_geometries_are_declined(hip_lib):
    from msf_wsi_amd import kernels as kn

    for dt, N, H, Cn, K, R, stride in ((torch.bfloat16, 2, 7, 512, 512, 3, 1), (torch.bfloat16, 2, 56, 128, 128, 3, 1),
                                       (torch.float32, 2, 14, 256, 256, 3, 1), (torch.bfloat16, 2, 14, 256, 256, 1, 1),
                                       (torch.bfloat16, 2, 28, 128, 128, 3, 2), (torch.bfloat16, 2, 14, 256, 128, 3, 1)):
        d = kn.conv_desc(dt, N, H, H, Cn, K, R, R, stride, R // 2)
        assert not kn.img3x3_supported(d)
    d = kn.conv_desc(torch.bfloat16, 2, 7, 7, 512, 512, 3, 3, 1, 1)
    x = torch.zeros(2, 7, 7, 512, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(512, 3, 3, 512, dtype=torch.bfloat16, device="cuda")
    assert not kn.img3x3_fwd(d, x, w, torch.empty_like(x))
    assert not kn.img3x3_dgrad(d, x, w, torch.empty_like(x))


# N, P (= Q: the gradient's height), channels: dy [N, P, P, C] -> dx [N, 2P, 2P, C]
S2_GEOMS = [(3, 28, 128), (2, 14, 256), (5, 14, 256), (4, 28, 128)]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", S2_GEOMS)
@pytest.mark.parametrize("gate", [False, True])
@pytest.mark.parametrize("bnbwd", [False, True])
def test_img3x3_s2_dgrad(hip_lib, dt, geom, gate, bnbwd):
    """the strided conv2's input gradient in one launch (four parity passes over one staged band) against fp64
    conv2d_input and against msfwsi_conv_dgrad's four parity launches"""
    from msf_wsi_amd import kernels as kn

    N, P, Cn = geom
    H = 2 * P
    g = torch.Generator().manual_seed(34)
    w = rnd((Cn, Cn, 3, 3), dt, g, 1.0 / math.sqrt(Cn * 9))
    dy = rnd((N, Cn, P, P), dt, g)
    c2 = rnd((N, Cn, P, P), dt, g)
    c1 = rnd((N, Cn, H, H), dt, g)
    k1, k2, k3 = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.1, torch.randn(Cn, generator=g) * 0.01
    sc, sh = torch.rand(Cn, generator=g) - 0.3, torch.randn(Cn, generator=g) * 0.3
    d = kn.conv_desc(dt, N, H, H, Cn, Cn, 3, 3, 2, 1)
    assert (d.P, d.Q) == (P, P) and kn.img3x3_s2_dgrad_supported(d)
    dyd, wd = nhwc(dy).cuda(), nhwc(w).cuda()
    wpk = kn.img3x3_pack_weights(wd, torch.empty_like(wd), 2)
    dc_ref, kw, dc = dy, {}, None
    if bnbwd:
        v = lambda t: t.view(1, -1, 1, 1)
        dc_ref = torch.addcmul(torch.addcmul(v(k3), v(k2), c2.float()), v(k1), dy.float()).to(dt)
        dc = torch.empty(N, P, P, Cn, dtype=dt, device="cuda")
        kw = dict(bnbwd=(nhwc(c2).cuda(), k1.cuda(), k2.cuda(), k3.cuda()), dc_out=dc)
    ref = torch.nn.grad.conv2d_input((N, Cn, H, H), w.double(), dc_ref.double(), stride=2, padding=1)
    if gate:
        ref = ref * ((c1.float() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) > 0)
        sums = kn.new_stats(Cn)
        c1d = nhwc(c1).cuda()
        act = torch.empty(N, H, H, Cn, dtype=dt, device="cuda")
        kw.update(mask=(c1d, sc.cuda(), sh.cuda()), sums=sums, act_out=act)
    dx = torch.full((N, H, H, Cn), float("nan"), dtype=dt, device="cuda")  # every output pixel must be written
    assert kn.img3x3_s2_dgrad(d, dyd, wpk, dx, **kw)
    torch.cuda.synchronize()
    assert rel(dx.float().cpu().permute(0, 3, 1, 2), ref) < tol(dt) * 2
    if bnbwd:
        got, want = dc.float().cpu().permute(0, 3, 1, 2), dc_ref.float()
        assert (got != want).float().mean().item() < 2e-3 and rel(got, want) < 1e-3
    if gate:
        s = sums.sum(0).cpu()
        gd = dx.double().cpu().reshape(-1, Cn)
        assert torch.allclose(s[0], gd.sum(0), rtol=1e-5, atol=1e-4)
        assert torch.allclose(s[1], (gd * c1d.double().cpu().reshape(-1, Cn)).sum(0), rtol=1e-5, atol=1e-4)
        want = torch.empty_like(act)
        kn.bn_act(c1d, sc.cuda(), sh.cuda(), want, relu=True)
        torch.cuda.synchronize()
        assert torch.equal(act, want)
    dx2 = torch.empty_like(dx)
    kw2 = dict(mask=kw["mask"], sums=kn.new_stats(Cn)) if gate else {}
    kn.conv_dgrad(d, nhwc(dc_ref).cuda(), wd, dx2, **kw2)
    torch.cuda.synchronize()
    assert rel(dx, dx2) < tol(dt) / 2


def test_img3x3_s2_dgrad_declines_other_geometries(hip_lib):
    from msf_wsi_amd import kernels as kn
    from msf_wsi_amd._lib import MsfwsiHipError

    for dt, H, Cn, K, stride in ((torch.bfloat16, 28, 512, 512, 2), (torch.bfloat16, 56, 128, 128, 1),
                                 (torch.float32, 56, 128, 128, 2), (torch.bfloat16, 56, 128, 256, 2)):
        assert not kn.img3x3_s2_dgrad_supported(kn.conv_desc(dt, 2, H, H, Cn, K, 3, 3, stride, 1))
    d = kn.conv_desc(torch.bfloat16, 2, 28, 28, 512, 512, 3, 3, 2, 1)
    dy = torch.zeros(2, 14, 14, 512, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(512, 3, 3, 512, dtype=torch.bfloat16, device="cuda")
    assert not kn.img3x3_s2_dgrad(d, dy, w, torch.zeros(2, 28, 28, 512, dtype=torch.bfloat16, device="cuda"))
    # in place only where a workgroup owns the whole gradient image
    d = kn.conv_desc(torch.bfloat16, 2, 56, 56, 128, 128, 3, 3, 2, 1)
    dy = torch.zeros(2, 28, 28, 128, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(128, 3, 3, 128, dtype=torch.bfloat16, device="cuda")
    k = [torch.ones(128, device="cuda") for _ in range(3)]
    with pytest.raises(MsfwsiHipError):
        kn.img3x3_s2_dgrad(d, dy, w, torch.zeros(2, 56, 56, 128, dtype=torch.bfloat16, device="cuda"),
                           bnbwd=(torch.zeros_like(dy), *k), dc_out=dy)
