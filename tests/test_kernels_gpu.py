"""Per-kernel numerics on a real MI355X: every C-ABI entry point against a plain PyTorch fp32/fp64 CPU
reference of the same operator on the same seeded inputs (fp32: rel-L2 <= 2e-5; bf16 storage: <= 1.5e-2,
fp16 storage: <= 2e-3;
inputs pre-rounded to the storage type so only the kernel's own rounding is measured)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def tol(dt):
    return {torch.float32: 2e-5, torch.bfloat16: 1.5e-2, torch.float16: 2e-3}[dt]


def rel(a, b):
    a = a.double().cpu()
    b = b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def nhwc(x):  # NCHW cpu fp32 -> NHWC contiguous
    return x.permute(0, 2, 3, 1).contiguous()


def rnd(shape, dt, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(dt).float()


CONVS = [
    # N, H, W, C, K, R, stride, pad
    (2, 14, 14, 64, 64, 3, 1, 1),
    (3, 13, 11, 32, 128, 3, 2, 1),
    (2, 9, 9, 64, 256, 1, 1, 0),
    (2, 10, 10, 128, 64, 1, 2, 0),
    (2, 20, 20, 8, 64, 7, 2, 3),
    (5, 1, 1, 64, 16, 1, 1, 0),     # Linear 64->16, 5 rows
    (130, 1, 1, 16, 64, 1, 1, 0),   # Linear 16->64
    (9, 1, 1, 576, 144, 1, 1, 0),   # fuser predictor shapes
    (3, 6, 6, 256, 200, 3, 1, 1),   # K not a tile multiple
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", CONVS)
@pytest.mark.parametrize("pro", [False, True])
def test_conv_fwd(hip_lib, dt, geom, pro):
    from msf_wsi_amd import kernels as kn

    N, H, W, Cc, K, R, st, pad = geom
    g = torch.Generator().manual_seed(1)
    x = rnd((N, Cc, H, W), dt, g)
    w = rnd((K, Cc, R, R), dt, g, 1.0 / math.sqrt(Cc * R * R))
    bias = torch.randn(K, generator=g)
    sc = torch.rand(Cc, generator=g) + 0.5
    sh = torch.randn(Cc, generator=g) * 0.3
    xin = x
    if pro:
        xin = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).to(dt).float()
    ref = F.conv2d(xin.double(), w.double(), bias.double(), stride=st, padding=pad).float()
    d = kn.conv_desc(dt, N, H, W, Cc, K, R, R, st, pad)
    xd = nhwc(x).to(dt).cuda()
    wd = nhwc(w).to(dt).cuda()
    y = torch.empty(N, d.P, d.Q, K, dtype=dt, device="cuda")
    stats = kn.new_stats(K)
    kn.conv_fwd(d, xd, wd, y, pro=(sc.cuda(), sh.cuda()) if pro else None, bias=bias.cuda(), stats=stats)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert rel(got, ref) < tol(dt)
    s = stats.sum(0).cpu()
    yy = y.double().cpu().reshape(-1, K)
    assert torch.allclose(s[0], yy.sum(0), rtol=1e-5, atol=1e-4)
    assert torch.allclose(s[1], (yy * yy).sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", [(2, 14, 14, 64, 256, 1, 1, 0), (3, 7, 7, 128, 200, 1, 1, 0), (2, 9, 9, 32, 64, 3, 1, 1)])
@pytest.mark.parametrize("with_ident", [True, False])
def test_conv_fwd_post(hip_lib, dt, geom, with_ident):
    """conv with the consumer's BatchNorm apply + residual + ReLU in the epilogue (Bottleneck tail, resnet.py:131-138)
    and the Gram-matrix route to that BatchNorm's batch statistics (fold_matvec / fold_dots)"""
    from msf_wsi_amd import kernels as kn

    N, H, W, Cc, K, R, st, pad = geom
    g = torch.Generator().manual_seed(11)
    x = F.relu(rnd((N, Cc, H, W), dt, g) + 0.3).to(dt).float()
    w = rnd((K, Cc, R, R), dt, g, 1.0 / math.sqrt(Cc * R * R))
    ps, pb = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    d = kn.conv_desc(dt, N, H, W, Cc, K, R, R, st, pad)
    ident = rnd((N, K, d.P, d.Q), dt, g)
    c = F.conv2d(x.double(), w.double(), stride=st, padding=pad)
    ref = c.float().to(dt).double() * ps.view(1, -1, 1, 1) + pb.view(1, -1, 1, 1)
    if with_ident:
        ref = ref + ident.double()
    ref = F.relu(ref).float()
    y = torch.empty(N, d.P, d.Q, K, dtype=dt, device="cuda")
    xd, wd = nhwc(x).to(dt).cuda(), nhwc(w).to(dt).cuda()
    kn.conv_fwd_post(d, xd, wd, y, ps.cuda(), pb.cuda(), ident=nhwc(ident).to(dt).cuda() if with_ident else None)
    torch.cuda.synchronize()
    assert rel(y.float().cpu().permute(0, 3, 1, 2), ref) < tol(dt)
    if R == 1:  # statistics of c = W a from the Gram matrix of a
        A = torch.zeros(Cc, 1, 1, Cc, device="cuda")
        dsq = kn.conv_desc(dt, N, H, W, Cc, Cc, 1, 1, 1, 0)
        kn.conv_wgrad(dsq, xd, xd, A)
        sa = torch.zeros(Cc, dtype=torch.float64, device="cuda")
        kn.colsum(xd, sa)
        Wq = wd.float()
        WA = torch.empty(K, 1, 1, Cc, device="cuda")
        kn.conv_fwd(kn.conv_desc(torch.float32, K, 1, 1, Cc, Cc, 1, 1, 1, 0), Wq, A, WA)
        s = torch.zeros(2, K, dtype=torch.float64, device="cuda")
        kn.fold_matvec(Wq, sa, s[0])
        kn.fold_dots(Wq, WA, s[1])
        torch.cuda.synchronize()
        cc = c.permute(0, 2, 3, 1).reshape(-1, K)
        assert torch.allclose(s[0].cpu(), cc.sum(0), rtol=1e-5, atol=1e-4)
        assert torch.allclose(s[1].cpu(), (cc * cc).sum(0), rtol=2e-5, atol=1e-4)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(3, 10, 64, 64, 256), (2, 7, 128, 256, 200), (1, 5, 32, 64, 64)])
def test_conv_fwd_post2_two_sources(hip_lib, dt, shape):
    """y = relu(round(x . W1^T + s2 . W2^T) * scale + shift): a Bottleneck tail with its downsample branch"""
    from msf_wsi_amd import kernels as kn

    N, H, C1, C2, K = shape
    g = torch.Generator().manual_seed(16)
    x = rnd((N * H * H, C1), dt, g)
    s2 = rnd((N * H * H, C2), dt, g)
    W1 = rnd((K, C1), dt, g, 1.0 / math.sqrt(C1))
    W2 = rnd((K, C2), dt, g, 1.0 / math.sqrt(C2))
    ps, pb = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    acc = x.double() @ W1.double().t() + s2.double() @ W2.double().t()
    ref = F.relu(acc.float().to(dt).double() * ps.double() + pb.double()).float()
    d = kn.conv_desc(dt, N, H, H, C1, K, 1, 1, 1, 0)
    y = torch.empty(N, H, H, K, dtype=dt, device="cuda")
    bits = kn.gate_bytes(N * H * H, K, dt)
    wcat = torch.cat([W1, W2], 1).to(dt).cuda()
    assert kn.conv_fwd_post2(d, x.to(dt).cuda().view(N, H, H, C1), wcat, y, s2.to(dt).cuda().view(N, H, H, C2),
                             ps.cuda(), pb.cuda(), relu=True, gate_out=bits)
    torch.cuda.synchronize()
    assert rel(y.float().cpu().view(-1, K), ref) < tol(dt)
    vec = 4 if dt == torch.float32 else 8
    want = (y.reshape(-1, K // vec, vec) > 0).to(torch.int32) * (2 ** torch.arange(vec, device="cuda", dtype=torch.int32))
    assert torch.equal(kn.gate_unpack(bits, y.numel() // K, K, dt).to(torch.int32), want.sum(-1))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(3, 10, 256, 64, 64), (2, 7, 512, 128, 128), (1, 5, 64, 32, 32)])
def test_conv_dgrad2_two_sources(hip_lib, dt, shape):
    """dx = gate(dy . W1 + src2 . W2 + bias): the two-source 1x1 input gradient of the folded bn3 backward"""
    from msf_wsi_amd import kernels as kn

    N, H, K, Cc, C2 = shape
    g = torch.Generator().manual_seed(13)
    dy = rnd((N * H * H, K), dt, g)
    s2 = rnd((N * H * H, C2), dt, g)
    W1 = rnd((K, Cc), dt, g, 1.0 / math.sqrt(K))
    W2 = rnd((C2, Cc), dt, g, 1.0 / math.sqrt(C2))
    bias = torch.randn(Cc, generator=g) * 0.1
    c = rnd((N * H * H, Cc), dt, g)
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    ref = dy.double() @ W1.double() + s2.double() @ W2.double() + bias.double()
    gate = (c.double() * sc.double() + sh.double()) > 0
    refg = torch.where(gate, ref.float().to(dt).double(), torch.zeros_like(ref))
    d = kn.conv_desc(dt, N, H, H, Cc, K, 1, 1, 1, 0)
    dx = torch.empty(N, H, H, Cc, dtype=dt, device="cuda")
    sums = kn.new_stats(Cc)
    wcat = torch.cat([W1, W2], 0).to(dt).cuda()
    ok = kn.conv_dgrad2(d, dy.to(dt).cuda().view(N, H, H, K), wcat, dx, s2.to(dt).cuda().view(N, H, H, C2),
                        bias=bias.cuda(), mask=(c.to(dt).cuda().view(N, H, H, Cc), sc.cuda(), sh.cuda()), sums=sums)
    torch.cuda.synchronize()
    assert ok
    got = dx.double().cpu().view(-1, Cc)
    # gate decisions at |scale*c+shift| ~ 0 may differ between fp64 and the kernel's fp32 fma: compare where clear
    clear = ((c.double() * sc.double() + sh.double()).abs() > 1e-3)
    assert rel(torch.where(clear, got, refg), refg) < tol(dt)
    st = sums.sum(0).cpu()
    assert torch.allclose(st[0], got.sum(0), rtol=1e-6, atol=1e-4)
    assert torch.allclose(st[1], (got * c.to(dt).double()).sum(0), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(3, 10, 256, 64, 64), (2, 7, 512, 128, 128), (1, 5, 64, 32, 32), (2, 23, 256, 64, 64),
                                   (1, 9, 2048, 512, 512), (5, 14, 1024, 256, 256)])
@pytest.mark.parametrize("big", [False, True])
def test_conv_dgrad2_second_source_normalised_in_the_launch(hip_lib, dt, shape, big):
    """msfwsi_conv_dgrad2_pro: the second source is the RAW conv output c2 and the operand relu(scale * c2 + shift) is formed
    on the fragments inside the launch -- BIT FOR BIT the result (output and BatchNorm sums) of msfwsi_conv_dgrad2 on the
    activation msfwsi_bn_act materialises, with c2 also the gate's tensor as in the folded tail's backward
    (src/models/resnet.py:128-131 backwards); both tile classes (big: the 256 x 128 / 8-wave tile forced); ragged last tile;
    also against fp64"""
    from helpers import tuned
    from msf_wsi_amd import kernels as kn

    N, H, K, Cc, C2 = shape
    assert Cc == C2  # the folded tail: the gate's tensor IS the second source
    g = torch.Generator().manual_seed(17)
    M = N * H * H
    dy = rnd((M, K), dt, g).to(dt).cuda()
    c2 = rnd((M, C2), dt, g).to(dt).cuda()
    W1 = rnd((K, Cc), dt, g, 1.0 / math.sqrt(K))
    W2 = rnd((C2, Cc), dt, g, 1.0 / math.sqrt(C2))
    bias = (torch.randn(Cc, generator=g) * 0.1).cuda()
    sc, sh = (torch.rand(C2, generator=g) + 0.5).cuda(), (torch.randn(C2, generator=g) * 0.3).cuda()
    wcat = torch.cat([W1, W2], 0).to(dt).cuda()
    d = kn.conv_desc(dt, N, H, H, Cc, K, 1, 1, 1, 0)
    with tuned(hip_lib, {0: 1} if big else {}):
        a2 = torch.empty_like(c2)
        kn.bn_act(c2, sc, sh, a2, relu=True)
        want, got = torch.empty(M, Cc, dtype=dt, device="cuda"), torch.full((M, Cc), float("nan"), dtype=dt, device="cuda")
        s_want, s_got = kn.new_stats(Cc), kn.new_stats(Cc)
        assert kn.conv_dgrad2(d, dy, wcat, want, a2, bias=bias, mask=(c2, sc, sh), sums=s_want)
        ok = kn.conv_dgrad2(d, dy, wcat, got, c2, bias=bias, mask=(c2, sc, sh), sums=s_got, src2_pro=(sc, sh))
    torch.cuda.synchronize()
    if not ok:
        # the scale / shift vectors live in the tile's reduction area of LDS: the 128 x 64 tile of a small grid has room for
        # 256 channels; wider second sources are DECLINED there (the engine then reads the materialised activation)
        assert C2 > 256 and not big, shape
        return
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert torch.equal(s_got.sum(0), s_want.sum(0))
    a64 = torch.relu(c2.double() * sc.double() + sh.double()).to(dt).double()
    ref = dy.double() @ wcat[:K].double() + a64 @ wcat[K:].double() + bias.double()
    pre = c2.double() * sc.double() + sh.double()
    refg = torch.where(pre > 0, ref.float().to(dt).double(), torch.zeros_like(ref)).cpu()
    clear = (pre.abs() > 1e-3).cpu()
    assert rel(torch.where(clear, got.double().cpu(), refg), refg) < tol(dt)
    # fp32 storage has no such launch: the library declines, it does not fall back silently
    d32 = kn.conv_desc(torch.float32, 1, 4, 4, 32, 32, 1, 1, 1, 0)
    z = torch.zeros(16, 32, device="cuda")
    assert not kn.conv_dgrad2(d32, z, torch.zeros(64, 32, device="cuda"), torch.empty_like(z), z, src2_pro=(sc[:32], sh[:32]))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("hw", [(30, 30), (33, 17), (64, 64)])
def test_stem_conv_as_row_runs(hip_lib, dt, hw):
    """7x7 / stride 2 / pad 3 on 3 channels (resnet.py:174) as 7 row taps over runs of contiguous pixels"""
    from msf_wsi_amd import kernels as kn

    H, W = hw
    N, K, R, CP = 3, 64, 7, (4 if dt == torch.float32 else 8)
    g = torch.Generator().manual_seed(15)
    x = rnd((N, 3, H, W), dt, g)
    w = rnd((K, 3, R, R), dt, g, 1.0 / math.sqrt(3 * R * R))
    ref = F.conv2d(x.double(), w.double(), stride=2, padding=3).float()
    xp = torch.zeros(N, H, W, CP)
    xp[..., :3] = nhwc(x)
    bk = 16 if dt == torch.float32 else 32
    run = (R * CP + bk - 1) // bk * bk
    w_run = torch.zeros(K, R, run)
    wp = torch.zeros(K, R, R, CP)
    wp[..., :3] = nhwc(w)
    w_run[:, :, :R * CP] = wp.reshape(K, R, R * CP)
    P, Q = ref.shape[2], ref.shape[3]
    y = torch.empty(N, P, Q, K, dtype=dt, device="cuda")
    stats = kn.new_stats(K)
    assert kn.stem_conv_fwd(xp.to(dt).cuda(), w_run.to(dt).cuda(), y, stats, R, R, 2, 3)
    torch.cuda.synchronize()
    assert rel(y.float().cpu().permute(0, 3, 1, 2), ref) < tol(dt)
    yy = y.double().cpu().reshape(-1, K)
    assert torch.allclose(stats.sum(0).cpu()[0], yy.sum(0), rtol=1e-5, atol=1e-4)
    assert torch.allclose(stats.sum(0).cpu()[1], (yy * yy).sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("hw", [(8, 8), (7, 9)])
def test_conv_dgrad_lowres_residual(hip_lib, dt, hw):
    """resid_stride=2: the residual is the low-resolution tensor, added on the even-pixel sub-grid only"""
    from msf_wsi_amd import kernels as kn

    H, W = hw
    N, K, Cc = 3, 64, 256
    g = torch.Generator().manual_seed(18)
    dy = nhwc(rnd((N, K, H, W), dt, g)).to(dt).cuda()
    w = nhwc(rnd((K, Cc, 1, 1), dt, g, 0.2)).to(dt).cuda()
    P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    lo = nhwc(rnd((N, Cc, P, Q), dt, g)).to(dt).cuda()
    full = torch.empty(N, H, W, Cc, dtype=dt, device="cuda")
    kn.pixel_stride(lo, full, 2, expand=True)
    d = kn.conv_desc(dt, N, H, W, Cc, K, 1, 1, 1, 0)
    mc = nhwc(rnd((N, Cc, H, W), dt, g)).to(dt).cuda()
    one, zero = torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    a, b = torch.empty_like(full), torch.empty_like(full)
    sa, sb = kn.new_stats(Cc), kn.new_stats(Cc)
    kn.conv_dgrad(d, dy, w, a, resid=full, mask=(mc, one, zero), sums=sa)
    kn.conv_dgrad(d, dy, w, b, resid=lo, mask=(mc, one, zero), sums=sb, resid_stride=2)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert torch.allclose(sa.sum(0), sb.sum(0), rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("hw", [(8, 8), (7, 9), (1, 1)])
def test_pixel_stride_gather_and_expand(hip_lib, dt, hw):
    from msf_wsi_amd import kernels as kn

    H, W = hw
    N, Cc, s = 3, 64, 2
    g = torch.Generator().manual_seed(17)
    x = rnd((N, H, W, Cc), dt, g).to(dt).cuda()
    P, Q = (H - 1) // s + 1, (W - 1) // s + 1
    lo = torch.empty(N, P, Q, Cc, dtype=dt, device="cuda")
    kn.pixel_stride(x, lo, s, expand=False)
    assert torch.equal(lo, x[:, ::s, ::s, :].contiguous())
    full = torch.full((N, H, W, Cc), 7.0, dtype=dt, device="cuda")
    kn.pixel_stride(lo, full, s, expand=True)
    want = torch.zeros_like(x)
    want[:, ::s, ::s, :] = lo
    assert torch.equal(full, want)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(5000, 64), (333, 256), (70, 2048)])
def test_bn_act_sum(hip_lib, dt, shape):
    from msf_wsi_amd import kernels as kn

    M, Cc = shape
    g = torch.Generator().manual_seed(14)
    c = rnd((M, Cc), dt, g)
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    ref = F.relu(c * sc + sh).to(dt)
    out = torch.empty(M, Cc, dtype=dt, device="cuda")
    sa = torch.zeros(Cc, dtype=torch.float64, device="cuda")
    kn.bn_act_sum(c.to(dt).cuda(), sc.cuda(), sh.cuda(), out, sa)
    torch.cuda.synchronize()
    assert rel(out.float().cpu(), ref.float()) < (1e-6 if dt == torch.float32 else tol(dt))
    assert torch.allclose(sa.cpu(), out.double().cpu().sum(0), rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("dt", DTYPES)
def test_gate_bits_roundtrip(hip_lib, dt):
    """the ReLU gate written as bits by conv_fwd_post gates a later input gradient exactly like the activation"""
    from msf_wsi_amd import kernels as kn

    N, H, Cc, K = 3, 10, 32, 200 if dt == torch.float32 else 208
    g = torch.Generator().manual_seed(12)
    x = nhwc(rnd((N, Cc, H, H), dt, g)).to(dt).cuda()
    w = nhwc(rnd((K, Cc, 1, 1), dt, g, 0.2)).to(dt).cuda()
    ps, pb = (torch.rand(K, generator=g) + 0.5).cuda(), (torch.randn(K, generator=g) * 0.3).cuda()
    d = kn.conv_desc(dt, N, H, H, Cc, K, 1, 1, 1, 0)
    y = torch.empty(N, H, H, K, dtype=dt, device="cuda")
    bits = kn.gate_bytes(N * H * H, K, dt)
    kn.conv_fwd_post(d, x, w, y, ps, pb, relu=True, gate_out=bits)
    vec = 4 if dt == torch.float32 else 8
    want = (y.reshape(-1, K // vec, vec) > 0).to(torch.int32) * (2 ** torch.arange(vec, device="cuda", dtype=torch.int32))
    assert torch.equal(kn.gate_unpack(bits, y.numel() // K, K, dt).to(torch.int32), want.sum(-1))
    # a 1x1 conv K2 -> K whose input gradient is gated by y: bits vs activation
    K2 = 64
    d2 = kn.conv_desc(dt, N, H, H, K, K2, 1, 1, 1, 0)
    dy = nhwc(rnd((N, K2, H, H), dt, g)).to(dt).cuda()
    w2 = nhwc(rnd((K2, K, 1, 1), dt, g, 0.2)).to(dt).cuda()
    resid = nhwc(rnd((N, K, H, H), dt, g)).to(dt).cuda()
    one, zero = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
    dx_a, dx_b = torch.empty_like(y), torch.empty_like(y)
    s_a, s_b = kn.new_stats(K), kn.new_stats(K)
    kn.conv_dgrad(d2, dy, w2, dx_a, resid=resid, mask=(y, one, zero), sums=s_a)
    kn.conv_dgrad(d2, dy, w2, dx_b, resid=resid, mask_bits=bits, sums=s_b)
    torch.cuda.synchronize()
    assert torch.equal(dx_a, dx_b)
    assert torch.allclose(s_a.sum(0)[0], s_b.sum(0)[0], rtol=1e-12, atol=1e-9) and float(s_b.sum(0)[1].abs().max()) == 0.0


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", CONVS)
def test_conv_dgrad(hip_lib, dt, geom):
    from msf_wsi_amd import kernels as kn

    N, H, W, Cc, K, R, st, pad = geom
    g = torch.Generator().manual_seed(2)
    d = kn.conv_desc(dt, N, H, W, Cc, K, R, R, st, pad)
    w = rnd((K, Cc, R, R), dt, g, 1.0 / math.sqrt(K * R * R))
    dy = rnd((N, K, d.P, d.Q), dt, g)
    resid = rnd((N, Cc, H, W), dt, g)
    gapg = rnd((N, Cc), dt, g)
    ref = torch.nn.grad.conv2d_input((N, Cc, H, W), w.double(), dy.double(), stride=st, padding=pad)
    ref = ref + resid.double() + 0.25 * gapg.double().view(N, Cc, 1, 1)
    dx = torch.empty(N, H, W, Cc, dtype=dt, device="cuda")
    kn.conv_dgrad(d, nhwc(dy).to(dt).cuda(), nhwc(w).to(dt).cuda(), dx, resid=nhwc(resid).to(dt).cuda(),
                  gapg=gapg.to(dt).cuda(), gap_scale=0.25)
    torch.cuda.synchronize()
    assert rel(dx.float().cpu().permute(0, 3, 1, 2), ref) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", [CONVS[0], CONVS[1], CONVS[2], CONVS[8]])
def test_conv_dgrad_fused_activation_backward(hip_lib, dt, geom):
    """dgrad epilogue gating by the producer's relu(bn(c)) and accumulating {sum g, sum g*c}"""
    from msf_wsi_amd import kernels as kn

    N, H, W, Cc, K, R, st, pad = geom
    g = torch.Generator().manual_seed(12)
    d = kn.conv_desc(dt, N, H, W, Cc, K, R, R, st, pad)
    w = rnd((K, Cc, R, R), dt, g, 1.0 / math.sqrt(K * R * R))
    dy = rnd((N, K, d.P, d.Q), dt, g)
    c = rnd((N, Cc, H, W), dt, g)
    sc, sh = torch.rand(Cc, generator=g) - 0.3, torch.randn(Cc, generator=g) * 0.3  # some negative scales
    ref = torch.nn.grad.conv2d_input((N, Cc, H, W), w.double(), dy.double(), stride=st, padding=pad)
    gate = (c * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) > 0
    ref = ref * gate
    dx = torch.empty(N, H, W, Cc, dtype=dt, device="cuda")
    sums = kn.new_stats(Cc)
    cd = nhwc(c).to(dt).cuda()
    kn.conv_dgrad(d, nhwc(dy).to(dt).cuda(), nhwc(w).to(dt).cuda(), dx, mask=(cd, sc.cuda(), sh.cuda()), sums=sums)
    torch.cuda.synchronize()
    assert rel(dx.float().cpu().permute(0, 3, 1, 2), ref) < tol(dt)
    s = sums.sum(0).cpu()
    gd = dx.double().cpu().reshape(-1, Cc)
    assert torch.allclose(s[0], gd.sum(0), rtol=1e-5, atol=1e-4)
    assert torch.allclose(s[1], (gd * cd.double().cpu().reshape(-1, Cc)).sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", CONVS)
@pytest.mark.parametrize("pro", [False, True])
def test_conv_wgrad(hip_lib, dt, geom, pro):
    from msf_wsi_amd import kernels as kn

    N, H, W, Cc, K, R, st, pad = geom
    g = torch.Generator().manual_seed(3)
    d = kn.conv_desc(dt, N, H, W, Cc, K, R, R, st, pad)
    x = rnd((N, Cc, H, W), dt, g)
    dy = rnd((N, K, d.P, d.Q), dt, g)
    sc = torch.rand(Cc, generator=g) + 0.5
    sh = torch.randn(Cc, generator=g) * 0.3
    xin = x
    if pro:
        xin = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).to(dt).float()
    ref = torch.nn.grad.conv2d_weight(xin.double(), (K, Cc, R, R), dy.double(), stride=st, padding=pad)
    dw = torch.zeros(K, R, R, Cc, dtype=torch.float32, device="cuda")
    for tb in (1, 64):  # one split and many splits accumulate into the same buffer
        kn.conv_wgrad(d, nhwc(x).to(dt).cuda(), nhwc(dy).to(dt).cuda(), dw,
                      pro=(sc.cuda(), sh.cuda()) if pro else None, target_blocks=tb)
    torch.cuda.synchronize()
    got = dw.cpu().permute(0, 3, 1, 2) * 0.5
    assert rel(got, ref) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("rows,Cin,K", [(512, 512, 256), (96, 320, 136), (40, 64, 64), (1000, 256, 768)])
def test_conv_wgrad_store(hip_lib, dt, rows, Cin, K):
    """msfwsi_conv_wgrad_store: the Linear weight gradient WRITTEN by one launch (no atomics, no clear beforehand) -- the
    buffer starts full of NaN, every element must be overwritten with dy^T x (whole and ragged 256 / 128 / 64 tiles)"""
    from msf_wsi_amd import kernels as kn

    g = torch.Generator().manual_seed(31)
    d = kn.conv_desc(dt, rows, 1, 1, Cin, K, 1, 1, 1, 0)
    x = rnd((rows, Cin), dt, g)
    dy = rnd((rows, K), dt, g, 0.1)
    ref = dy.double().t() @ x.double()
    dw = torch.full((K, Cin), float("nan"), dtype=torch.float32, device="cuda")
    kn.conv_wgrad_store(d, x.to(dt).cuda(), dy.to(dt).cuda(), dw)
    kn.conv_wgrad_store(d, x.to(dt).cuda(), dy.to(dt).cuda(), dw)  # a second launch REPLACES, it does not add
    torch.cuda.synchronize()
    assert torch.isfinite(dw).all()
    assert rel(dw.cpu(), ref) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(300, 64), (7, 144), (5000, 16), (64, 2304)])
def test_batchnorm_train_fwd_bwd(hip_lib, dt, shape):
    """conv-epilogue stats -> bn_finalize -> bn_act, then act_bwd_reduce -> bn_bwd_finalize -> bn_bwd_apply
    against F.batch_norm + relu under autograd."""
    from msf_wsi_amd import kernels as kn

    M, Cn = shape
    g = torch.Generator().manual_seed(4)
    c = rnd((M, Cn), dt, g) * 1.5 + 0.3
    c = c.to(dt).float()
    gamma = torch.rand(Cn, generator=g) + 0.5
    beta = torch.randn(Cn, generator=g) * 0.2
    rm, rv = torch.randn(Cn, generator=g) * 0.1, torch.rand(Cn, generator=g) + 0.5
    da = rnd((M, Cn), dt, g)
    # reference
    cr = c.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True)
    br = beta.double().requires_grad_(True)
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    bn = F.batch_norm(cr, rm_ref, rv_ref, gr, br, training=True, momentum=0.1, eps=1e-5)
    a = F.relu(bn)
    a.backward(da.double())
    # device
    cd = c.to(dt).cuda()
    sums = kn.new_stats(Cn)
    sums[0, 0] = cd.double().sum(0)
    sums[0, 1] = (cd.double() ** 2).sum(0)
    scale, shift, mean, invstd = (torch.empty(Cn, device="cuda") for _ in range(4))
    rmd, rvd = rm.cuda(), rv.cuda()
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    kn.bn_finalize(sums, M, gamma.cuda(), beta.cuda(), 1e-5, 0.1, rmd, rvd, nbt, scale, shift, mean, invstd)
    out = torch.empty_like(cd)
    kn.bn_act(cd, scale, shift, out, relu=True)
    torch.cuda.synchronize()
    assert rel(out.float(), a.detach()) < tol(dt)
    assert torch.allclose(rmd.cpu().double(), rm_ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(rvd.cpu().double(), rv_ref, rtol=1e-5, atol=1e-6)
    assert int(nbt.item()) == 1
    # backward
    bs = kn.new_stats(Cn)
    gbuf = torch.empty_like(cd)
    kn.act_bwd_reduce(da.to(dt).cuda(), cd, scale, shift, gbuf, bs)
    dgamma, dbeta = torch.zeros(Cn, device="cuda"), torch.zeros(Cn, device="cuda")
    k1, k2, k3 = (torch.empty(Cn, device="cuda") for _ in range(3))
    kn.bn_bwd_finalize(bs, 2, 1, M, gamma.cuda(), mean, invstd, dgamma, dbeta, k1, k2, k3)
    dc = torch.empty_like(cd)
    kn.bn_bwd_apply(gbuf, cd, k1, k2, k3, dc)
    torch.cuda.synchronize()
    t = tol(dt) * (1 if dt == torch.float32 else 2)
    # activations exactly at the ReLU kink may flip in bf16; rel-L2 absorbs that
    assert rel(dc.float(), cr.grad) < max(t, 1e-4)
    assert rel(dgamma, gr.grad) < max(t, 1e-4)
    assert rel(dbeta, br.grad) < max(t, 1e-4)


@pytest.mark.parametrize("dt", DTYPES)
def test_residual_block_end(hip_lib, dt):
    from msf_wsi_amd import kernels as kn

    N, HW, Cn = 3, 25, 64
    M = N * HW
    g = torch.Generator().manual_seed(5)
    c = rnd((M, Cn), dt, g)
    cds = rnd((M, Cn), dt, g)
    s1, b1 = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.2
    s2, b2 = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.2
    y_ref = F.relu(c * s1 + b1 + cds * s2 + b2)
    out = torch.empty(M, Cn, dtype=dt, device="cuda")
    kn.bn_act(c.to(dt).cuda(), s1.cuda(), b1.cuda(), out, ident=cds.to(dt).cuda(), id_scale=s2.cuda(),
              id_shift=b2.cuda(), relu=True)
    assert rel(out.float(), y_ref) < tol(dt)
    out2 = torch.empty(M, Cn, dtype=dt, device="cuda")
    kn.bn_act(c.to(dt).cuda(), s1.cuda(), b1.cuda(), out2, ident=cds.to(dt).cuda(), relu=True)
    assert rel(out2.float(), F.relu(c * s1 + b1 + cds)) < tol(dt)
    # backward sums
    dy = rnd((M, Cn), dt, g)
    gap = rnd((N, Cn), dt, g)
    yv = out.float().cpu()
    gg = (dy + 0.04 * gap.repeat_interleave(HW, 0)).to(dt).float() * (yv > 0)
    gbuf = torch.empty(M, Cn, dtype=dt, device="cuda")
    sums = kn.new_stats(Cn, 3)
    kn.block_end_bwd(dy.to(dt).cuda(), out, gap.to(dt).cuda(), 0.04, c.to(dt).cuda(), cds.to(dt).cuda(), gbuf, sums,
                     HW)
    torch.cuda.synchronize()
    assert rel(gbuf.float(), gg) < tol(dt)
    s = sums.sum(0).cpu()
    gd = gbuf.double().cpu()
    assert torch.allclose(s[0], gd.sum(0), rtol=1e-5, atol=1e-4)
    assert torch.allclose(s[1], (gd * c.double()).sum(0), rtol=1e-5, atol=1e-4)
    assert torch.allclose(s[2], (gd * cds.double()).sum(0), rtol=1e-5, atol=1e-4)


@pytest.fixture
def pool_bwd_kernel(request, hip_lib):
    """which max-pool backward kernel msfwsi_stem_pool_bwd dispatches to: "walk" = the column walk (default,
    msfwsi_set_tuning(16, 1)), "pixel" = one thread per pixel (16 -> 0)"""
    assert hip_lib.msfwsi_set_tuning(16, 1 if request.param == "walk" else 0) == 0
    yield request.param
    assert hip_lib.msfwsi_set_tuning(16, 1) == 0


POOL_CASES = [(k, hw) for k in ("walk", "pixel") for hw in ((16, 16), (15, 13), (14, 17), (34, 70))]


@pytest.mark.parametrize("pool_bwd_kernel,hw", POOL_CASES, indirect=["pool_bwd_kernel"],
                         ids=[f"{k}-{h}x{w}" for k, (h, w) in POOL_CASES])
@pytest.mark.parametrize("dt", DTYPES)
def test_stem_pool(hip_lib, dt, hw, pool_bwd_kernel):
    from msf_wsi_amd import kernels as kn

    N, Cn = 2, 64
    H, W = hw
    g = torch.Generator().manual_seed(6)
    c0 = rnd((N, Cn, H, W), dt, g)
    sc, sh = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.3
    # the kernel forms scale*c + shift with ONE rounding (fma): the reference does the same -- the product of two fp32 numbers
    # is exact in fp64, the sum is rounded once to fp32 (keeps the 1e-6 gate instead of loosening it to 1e-5)
    lin = (c0.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float()
    a = F.relu(lin).to(dt).double().requires_grad_(True)
    p_ref = F.max_pool2d(a, 3, 2, 1)
    P, Q = p_ref.shape[2:]
    dp = rnd((N, Cn, P, Q), dt, g)
    p_ref.backward(dp.double())
    g_ref = a.grad * (a.detach() > 0)
    c0d = nhwc(c0).to(dt).cuda()
    out = torch.empty(N, P, Q, Cn, dtype=dt, device="cuda")
    am = torch.empty(N, P, Q, Cn, dtype=torch.uint8, device="cuda")
    kn.stem_pool_fwd(c0d, sc.cuda(), sh.cuda(), out, am, N, H, W, Cn)
    torch.cuda.synchronize()
    assert rel(out.float().cpu().permute(0, 3, 1, 2), p_ref.detach()) < 1e-6
    if pool_bwd_kernel == "walk":  # the column-walk forward equals the per-window kernel bit for bit, argmax codes included
        out2, am2 = torch.empty_like(out), torch.empty_like(am)
        hip_lib.msfwsi_set_tuning(16, 0)
        kn.stem_pool_fwd(c0d, sc.cuda(), sh.cuda(), out2, am2, N, H, W, Cn)
        hip_lib.msfwsi_set_tuning(16, 1)
        torch.cuda.synchronize()
        assert torch.equal(out, out2) and torch.equal(am, am2)
    g0 = torch.empty(N, H, W, Cn, dtype=dt, device="cuda")
    sums = kn.new_stats(Cn)
    kn.stem_pool_bwd(nhwc(dp).to(dt).cuda(), am, c0d, sc.cuda(), sh.cuda(), g0, sums, N, H, W, Cn)
    torch.cuda.synchronize()
    assert rel(g0.float().cpu().permute(0, 3, 1, 2), g_ref) < tol(dt)
    s = sums.sum(0).cpu()
    gd = g0.double().cpu().reshape(-1, Cn)
    assert torch.allclose(s[0], gd.sum(0), rtol=1e-5, atol=1e-4)
    assert torch.allclose(s[1], (gd * c0d.double().cpu().reshape(-1, Cn)).sum(0), rtol=1e-5, atol=1e-4)
    # sums-only pass, then the apply pass that re-derives g: dc = k1*g + k2*c0 + k3 without g in memory
    sums2 = kn.new_stats(Cn)
    kn.stem_pool_bwd(nhwc(dp).to(dt).cuda(), am, c0d, sc.cuda(), sh.cuda(), None, sums2, N, H, W, Cn)
    assert torch.allclose(sums2.sum(0).cpu(), s, rtol=1e-9, atol=1e-9)
    k1, k2, k3 = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.1, torch.randn(Cn, generator=g) * 0.1
    dc = torch.empty_like(g0)
    kn.stem_pool_bwd(nhwc(dp).to(dt).cuda(), am, c0d, sc.cuda(), sh.cuda(), dc, None, N, H, W, Cn,
                     k=(k1.cuda(), k2.cuda(), k3.cuda()))
    want = torch.empty_like(g0)
    kn.bn_bwd_apply(g0, c0d, k1.cuda(), k2.cuda(), k3.cuda(), want)
    torch.cuda.synchronize()
    assert torch.equal(dc, want)


@pytest.mark.parametrize("dt", DTYPES)
def test_gap_permute_copy_colsum(hip_lib, dt):
    from msf_wsi_amd import kernels as kn

    g = torch.Generator().manual_seed(7)
    N, HW, Cn = 5, 49, 128
    y = rnd((N, HW, Cn), dt, g)
    out = torch.empty(N, Cn, dtype=dt, device="cuda")
    kn.gap_fwd(y.to(dt).cuda(), out, N, HW, Cn)
    assert rel(out.float(), y.mean(1)) < tol(dt)
    B, K = 3, 16
    f = rnd((B * K, Cn), dt, g)
    idx = torch.stack([torch.randperm(K, generator=g) for _ in range(B)])
    ref = f.view(B, K, Cn)[torch.arange(B).view(-1, 1), idx].reshape(B * K, Cn)
    o = torch.empty(B * K, Cn, dtype=dt, device="cuda")
    kn.rows_permute(f.to(dt).cuda(), idx.cuda(), o, B, K, Cn)
    assert torch.equal(o.float().cpu(), ref)
    back = torch.zeros(B * K, Cn, dtype=dt, device="cuda")
    kn.rows_permute(o, idx.cuda(), back, B, K, Cn, scatter=True)
    assert torch.equal(back.float().cpu(), f)
    # concat-style copy
    dst = torch.zeros(B, 9 * Cn, dtype=dt, device="cuda")
    ctx = rnd((B, Cn), dt, g)
    kn.copy2d(ctx.to(dt).cuda(), 0, Cn, dst, 0, 9 * Cn, B, Cn)
    kn.copy2d(f.to(dt).cuda(), 0, K * Cn, dst, Cn, 9 * Cn, B, 8 * Cn)
    refc = torch.cat([ctx, f.view(B, K, Cn)[:, :8].flatten(1)], 1)
    assert torch.equal(dst.float().cpu(), refc)
    kn.copy2d(f.to(dt).cuda(), 0, K * Cn, dst, Cn, 9 * Cn, B, 8 * Cn, accumulate=True)
    refc[:, Cn:] = (refc[:, Cn:] * 2).to(dt).float()
    assert torch.equal(dst.float().cpu(), refc)
    cs = torch.zeros(Cn, dtype=torch.float64, device="cuda")
    kn.colsum(f.to(dt).cuda(), cs)
    assert torch.allclose(cs.cpu(), f.double().sum(0), rtol=1e-6, atol=1e-5)
    acc = torch.ones(Cn, device="cuda")
    kn.add_f64_to_f32(cs, acc, 2.0)
    assert torch.allclose(acc.cpu().double(), 1 + 2 * f.double().sum(0), rtol=1e-5, atol=1e-4)


def test_nchw_to_nhwc_and_pad(hip_lib):
    from msf_wsi_amd import kernels as kn

    g = torch.Generator().manual_seed(8)
    x = torch.randn(3, 3, 10, 12, generator=g)
    for dt, CP in ((torch.float32, 4), (torch.bfloat16, 8), (torch.float16, 8)):
        y = torch.empty(3, 10, 12, CP, dtype=dt, device="cuda")
        kn.nchw_to_nhwc(x.cuda(), y, CP)
        ref = torch.zeros(3, 10, 12, CP)
        ref[..., :3] = x.permute(0, 2, 3, 1)
        assert torch.equal(y.float().cpu(), ref.to(dt).float())
        w = torch.randn(64 * 49, 3, generator=g)
        wp = torch.empty(64 * 49, CP, dtype=dt, device="cuda")
        kn.pad_cast(w.cuda(), wp, 64 * 49, 3, CP)
        assert torch.equal(wp.float().cpu()[:, :3], w.to(dt).float())
        assert wp.float().cpu()[:, 3:].abs().max() == 0
        gw = torch.randn(64 * 49, CP, generator=g)
        acc = torch.ones(64 * 49, 3, device="cuda")
        kn.unpad_add(gw.cuda(), acc, 64 * 49, 3, CP)
        assert torch.allclose(acc.cpu(), 1 + gw[:, :3])


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(8, 64), (128, 512), (5, 4608)])
def test_cosine_loss(hip_lib, dt, shape):
    from msf_wsi_amd import kernels as kn

    rows, d = shape
    g = torch.Generator().manual_seed(9)
    p = rnd((rows, d), dt, g)
    z = rnd((rows, d), dt, g)
    pr = p.double().requires_grad_(True)
    w = 0.7
    loss = -(F.cosine_similarity(pr, z.double(), dim=1).mean()) * 0.5 * w
    (loss * 128.0).backward()
    acc = torch.zeros(1, dtype=torch.float64, device="cuda")
    dp = torch.empty(rows, d, dtype=dt, device="cuda")
    ls = torch.full((1,), 128.0, device="cuda")
    kn.cosine_loss(p.to(dt).cuda(), z.to(dt).cuda(), -0.5 * w / rows, acc, dp, ls)
    torch.cuda.synchronize()
    assert abs(acc.item() - loss.item()) < 1e-5
    assert rel(dp.float(), pr.grad) < tol(dt)


def test_adam_and_scaler(hip_lib):
    from msf_wsi_amd import kernels as kn

    g = torch.Generator().manual_seed(10)
    n = 10007
    p0 = torch.randn(n, generator=g)
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pr], lr=3e-3)
    p = p0.clone().cuda()
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pb = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    ph = torch.empty(n, dtype=torch.float16, device="cuda")
    scale = torch.full((1,), 1024.0, device="cuda")
    found = torch.zeros(1, device="cuda")
    for step in range(1, 4):
        gr = torch.randn(n, generator=g)
        pr.grad = gr.clone()
        opt.step()
        gd = (gr * 1024.0).cuda()
        kn.nonfinite_check(gd, found)
        kn.adam(p, gd, m, v, 3e-3, 0.9, 0.999, 1e-8, step, loss_scale=scale, found=found, p_lowp=pb)
    torch.cuda.synchronize()
    assert found.item() == 0
    assert torch.allclose(p.cpu(), pr.detach(), rtol=1e-6, atol=1e-7)
    assert torch.equal(pb.float().cpu(), p.cpu().bfloat16().float())
    kn.cast_lowp(p, ph)
    assert torch.equal(ph.float().cpu(), p.cpu().half().float())
    kn.cast_lowp(p, pb)
    assert torch.equal(pb.float().cpu(), p.cpu().bfloat16().float())
    # a non-finite gradient: flagged, step skipped, scale backs off
    gd = torch.randn(n, generator=g).cuda()
    gd[n - 2] = float("inf")
    before = p.clone()
    kn.nonfinite_check(gd, found)
    kn.adam(p, gd, m, v, 3e-3, 0.9, 0.999, 1e-8, 4, loss_scale=scale, found=found)
    tracker = torch.zeros(1, dtype=torch.int32, device="cuda")
    kn.scaler_update(scale, tracker, found, 2.0, 0.5, 2)
    torch.cuda.synchronize()
    assert found.item() == 1 and torch.equal(p, before) and scale.item() == 512.0
    found.zero_()
    kn.scaler_update(scale, tracker, found, 2.0, 0.5, 2)
    kn.scaler_update(scale, tracker, found, 2.0, 0.5, 2)
    torch.cuda.synchronize()
    assert scale.item() == 1024.0 and tracker.item() == 0


def test_bad_arguments_are_rejected(hip_lib):
    """the ABI refuses malformed geometry instead of launching"""
    from msf_wsi_amd import kernels as kn
    from msf_wsi_amd._lib import MsfwsiHipError

    d = kn.conv_desc(torch.float32, 1, 8, 8, 6, 8, 3, 3, 1, 1)  # C=6 is not a 16-byte multiple
    x = torch.zeros(1, 8, 8, 6, device="cuda")
    w = torch.zeros(8, 3, 3, 6, device="cuda")
    y = torch.zeros(1, 8, 8, 8, device="cuda")
    with pytest.raises(MsfwsiHipError):
        kn.conv_fwd(d, x, w, y)
    with pytest.raises(MsfwsiHipError):
        kn.conv_fwd(kn.conv_desc(torch.float32, 1, 8, 8, 8, 8, 3, 3, 1, 1), x.cpu(), w, y)


HALO = [  # N, H, W, C, K     (3x3, stride 1, pad 1)
    (2, 56, 56, 64, 64),
    (3, 28, 28, 128, 128),
    (5, 14, 14, 64, 256),    # one 196-pixel image per 256-row tile
    (2, 20, 23, 32, 96),     # odd width, HW not a multiple of the tile, K not a tile multiple
    (1, 16, 8, 96, 32),
    # 64 -> 64 channels: the weights-stationary persistent kernel (2-byte types); tiles run over the batch's global
    # raster, so these put image boundaries inside tiles
    (5, 14, 14, 64, 64),     # 196-pixel images: every tile spans two of them
    (3, 20, 23, 64, 64),     # odd width
    (1, 12, 11, 64, 64),     # a single ragged tile
    (2, 9, 56, 64, 64),
    (2, 5, 64, 64, 64),      # the widest row the stationary kernels' halo holds (layer1 of a 256 x 256 tile)
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", HALO)
def test_conv3x3_halo_fwd(hip_lib, dt, geom):
    """halo-in-LDS 3x3 kernel (csrc/conv3x3.hip) against F.conv2d, incl. the BatchNorm sums"""
    from msf_wsi_amd import kernels as kn

    N, H, W, Cc, K = geom
    if dt == torch.float32 and Cc % 16:
        pytest.skip("fp32 slabs are 16 channels")
    g = torch.Generator().manual_seed(21)
    x = rnd((N, Cc, H, W), dt, g)
    w = rnd((K, Cc, 3, 3), dt, g, 1.0 / math.sqrt(Cc * 9))
    d = kn.conv_desc(dt, N, H, W, Cc, K, 3, 3, 1, 1)
    if W > 56 and not kn.conv3x3_stationary(d):
        pytest.skip("rows wider than 56 pixels: stationary kernel (2-byte types) only")
    assert kn.conv3x3_supported(d) or kn.conv3x3_stationary(d)
    ref = F.conv2d(x.double(), w.double(), None, stride=1, padding=1)
    y = torch.empty(N, H, W, K, dtype=dt, device="cuda")
    stats = kn.new_stats(K)
    kn.conv3x3_fwd(d, nhwc(x).to(dt).cuda(), nhwc(w).to(dt).cuda(), y, stats=stats)
    torch.cuda.synchronize()
    assert rel(y.float().cpu().permute(0, 3, 1, 2), ref) < tol(dt)
    s = stats.sum(0).cpu()
    yy = y.double().cpu().reshape(-1, K)
    assert torch.allclose(s[0], yy.sum(0), rtol=1e-5, atol=1e-4)
    assert torch.allclose(s[1], (yy * yy).sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom", [g for g in HALO if g[3] == 64 and g[4] == 64])
def test_conv3x3_stationary_fused_prologue(hip_lib, dt, geom):
    """conv(relu(scale * c + shift)) with the BatchNorm + ReLU applied inside the weights-stationary kernel (padding stays
    zero in ACTIVATION space, image borders inside a tile included) against the materialised activation"""
    from msf_wsi_amd import kernels as kn

    N, H, W, Cc, K = geom
    g = torch.Generator().manual_seed(23)
    d = kn.conv_desc(dt, N, H, W, Cc, K, 3, 3, 1, 1)
    assert kn.conv3x3_stationary(d)
    x = rnd((N, Cc, H, W), dt, g)
    w = rnd((K, Cc, 3, 3), dt, g, 1.0 / math.sqrt(Cc * 9))
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3 + 0.2  # shift > 0: relu(shift) != 0
    a = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).to(dt).float()         # the stored activation
    ref = F.conv2d(a.double(), w.double(), None, stride=1, padding=1)
    y = torch.empty(N, H, W, K, dtype=dt, device="cuda")
    stats = kn.new_stats(K)
    kn.conv3x3_fwd(d, nhwc(x).to(dt).cuda(), nhwc(w).to(dt).cuda(), y, stats=stats, pro=(sc.cuda(), sh.cuda()))
    torch.cuda.synchronize()
    assert rel(y.float().cpu().permute(0, 3, 1, 2), ref) < tol(dt)
    yy = y.double().cpu().reshape(-1, K)
    assert torch.allclose(stats.sum(0).cpu()[0], yy.sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom", [g for g in HALO if g[3] == 64 and g[4] == 64 and g[2] <= 56] + [(9, 3, 2, 64, 64)])
@pytest.mark.parametrize("pro", [False, True])
def test_conv_wgrad_output_stationary(hip_lib, dt, geom, pro):
    """the 64 -> 64 3x3 weight gradient on the zero-padded raster (with and without the fused BatchNorm + ReLU of the
    producer) against fp64, and against the gather kernel on the same operands"""
    from msf_wsi_amd import kernels as kn

    N, H, W, Cc, K = geom
    g = torch.Generator().manual_seed(24)
    d = kn.conv_desc(dt, N, H, W, Cc, K, 3, 3, 1, 1)
    x = rnd((N, Cc, H, W), dt, g)
    dy = rnd((N, K, H, W), dt, g, 0.1)
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3 + 0.2
    xin = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).to(dt).float() if pro else x
    ref = torch.nn.grad.conv2d_weight(xin.double(), (K, Cc, 3, 3), dy.double(), stride=1, padding=1)
    xd, dyd = nhwc(x).to(dt).cuda(), nhwc(dy).to(dt).cuda()
    kw = dict(pro=(sc.cuda(), sh.cuda())) if pro else {}
    got = {}
    try:
        hip_lib.msfwsi_set_tuning(11, 0)  # small test shapes: lift the size threshold of the kernel
        assert kn.conv_wgrad_stationary(d)
        for on in (1, 0):
            hip_lib.msfwsi_set_tuning(10, on)
            dw = torch.zeros(K, 3, 3, Cc, device="cuda")
            kn.conv_wgrad(d, xd, dyd, dw, **kw)
            kn.conv_wgrad(d, xd, dyd, dw, **kw)  # accumulates
            torch.cuda.synchronize()
            got[on] = dw.cpu().permute(0, 3, 1, 2) * 0.5
    finally:
        hip_lib.msfwsi_set_tuning(10, 1)
        hip_lib.msfwsi_set_tuning(11, 32 * 256 * 256)
    assert rel(got[1], ref) < 2e-5
    assert rel(got[0], ref) < 2e-5


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", HALO)
@pytest.mark.parametrize("fused", [False, True])
def test_conv3x3_halo_dgrad(hip_lib, dt, geom, fused):
    from msf_wsi_amd import kernels as kn

    N, H, W, Cc, K = geom
    if dt == torch.float32 and (Cc % 16 or K % 16):
        pytest.skip("fp32 slabs are 16 channels")
    g = torch.Generator().manual_seed(22)
    d = kn.conv_desc(dt, N, H, W, Cc, K, 3, 3, 1, 1)
    if W > 56 and not kn.conv3x3_stationary(d):
        pytest.skip("rows wider than 56 pixels: stationary kernel (2-byte types) only")
    w = rnd((K, Cc, 3, 3), dt, g, 1.0 / math.sqrt(K * 9))
    dy = rnd((N, K, H, W), dt, g)
    resid = rnd((N, Cc, H, W), dt, g)
    c = rnd((N, Cc, H, W), dt, g)
    sc, sh = torch.rand(Cc, generator=g) - 0.3, torch.randn(Cc, generator=g) * 0.3
    ref = torch.nn.grad.conv2d_input((N, Cc, H, W), w.double(), dy.double(), stride=1, padding=1) + resid.double()
    dx = torch.empty(N, H, W, Cc, dtype=dt, device="cuda")
    kw = {}
    if fused:
        ref = ref * ((c * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) > 0)
        sums = kn.new_stats(Cc)
        cd = nhwc(c).to(dt).cuda()
        kw = dict(mask=(cd, sc.cuda(), sh.cuda()), sums=sums)
    kn.conv3x3_dgrad(d, nhwc(dy).to(dt).cuda(), nhwc(w).to(dt).cuda(), dx, resid=nhwc(resid).to(dt).cuda(), **kw)
    torch.cuda.synchronize()
    assert rel(dx.float().cpu().permute(0, 3, 1, 2), ref) < tol(dt) * (1 if dt == torch.float32 else 2)
    if fused:
        s = sums.sum(0).cpu()
        gd = dx.double().cpu().reshape(-1, Cc)
        assert torch.allclose(s[0], gd.sum(0), rtol=1e-5, atol=1e-4)
        assert torch.allclose(s[1], (gd * cd.double().cpu().reshape(-1, Cc)).sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom", [(3, 9, 7, 64, 256), (2, 14, 14, 128, 512), (1, 5, 5, 24, 40)])
def test_conv_wgrad_act_writes_the_normalised_operand(hip_lib, dt, geom):
    """msfwsi_conv_wgrad_act: the 1x1 weight gradient with the producer's BatchNorm + ReLU applied in the register staging,
    and the normalised operand stored from there -- the gradient equals msfwsi_conv_wgrad's with the same prologue (the
    same kernel instance and split: bit for bit under one workgroup per tile) and fp64, the by-product equals msfwsi_bn_act's
    output bit for bit; other geometries decline"""
    from helpers import tuned
    from msf_wsi_amd import kernels as kn

    N, H, W, Cc, K = geom
    g = torch.Generator().manual_seed(41)
    x = rnd((N * H * W, Cc), dt, g).to(dt).cuda()
    dy = rnd((N * H * W, K), dt, g, 0.1).to(dt).cuda()
    sc, sh = (torch.rand(Cc, generator=g) + 0.5).cuda(), (torch.randn(Cc, generator=g) * 0.3).cuda()
    d = kn.conv_desc(dt, N, H, W, Cc, K, 1, 1, 1, 0)
    want_a = torch.empty_like(x)
    kn.bn_act(x, sc, sh, want_a, relu=True)
    with tuned(hip_lib, {15: 1}):  # one workgroup per tile: no atomics, the sums repeat
        dw0 = torch.zeros(K, Cc, device="cuda")
        kn.conv_wgrad(d, x, dy, dw0, pro=(sc, sh))
        dw1 = torch.zeros(K, Cc, device="cuda")
        act = torch.full_like(x, float("nan"))
        assert kn.conv_wgrad_act(d, x, dy, dw1, (sc, sh), act)
    torch.cuda.synchronize()
    assert torch.equal(act, want_a)
    assert torch.equal(dw0, dw1)
    ref = dy.double().cpu().t() @ want_a.double().cpu()
    assert rel(dw1.cpu(), ref) < 2e-5
    d3 = kn.conv_desc(dt, N, H, W, Cc, K, 3, 3, 1, 1)
    assert not kn.conv_wgrad_act(d3, x, dy, torch.zeros(K, 9 * Cc, device="cuda"), (sc, sh), act)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", [(3, 8, 8, 64), (2, 7, 9, 32), (5, 14, 14, 256)])
def test_gap_fwd_stride2_is_gap_fwd_and_pixel_stride(hip_lib, dt, geom):
    """one pass over a stage output: pooled features bit for bit msfwsi_gap_fwd's, the strided copy bit for bit
    msfwsi_pixel_stride's (odd extents included)"""
    from msf_wsi_amd import kernels as kn

    N, H, W, Cn = geom
    g = torch.Generator().manual_seed(42)
    y = rnd((N, H, W, Cn), dt, g).to(dt).cuda()
    f0 = torch.empty(N, Cn, dtype=dt, device="cuda")
    kn.gap_fwd(y, f0, N, H * W, Cn)
    s0 = torch.empty(N, (H + 1) // 2, (W + 1) // 2, Cn, dtype=dt, device="cuda")
    kn.pixel_stride(y, s0, 2, expand=False)
    f1 = torch.full_like(f0, float("nan"))
    s1 = torch.full_like(s0, float("nan"))
    kn.gap_fwd_stride2(y, f1, s1, N, H, W, Cn)
    torch.cuda.synchronize()
    assert torch.equal(f0, f1) and torch.equal(s0, s1)
    assert torch.equal(s1, y[:, ::2, ::2, :].contiguous())
