"""Activation-stationary ("panel") 1x1 kernels (csrc/panel.hip) on a real MI355X, through the C ABI, against a plain
PyTorch fp64 CPU reference of the same operator chain on the same seeded, storage-rounded inputs -- the conv3 + bn3 +
identity + ReLU tail of a Bottleneck (reference src/models/resnet.py:128-138) and the bn1-backward + conv1 input gradient
(resnet.py:124-126 backwards).  bf16 storage <= 1.5e-2, fp16 storage <= 2e-3 rel-L2 (the bounds of test_kernels_gpu.py);
gate bits and the written-back BatchNorm-backward operand are compared exactly where the arithmetic allows it."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16]


def tol(dt):
    return {torch.bfloat16: 1.5e-2, torch.float16: 2e-3}[dt]


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def rnd(shape, dt, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(dt)


def unpack_bits(bits, M, Nout):
    """gate-byte buffer -> bool [M][Nout] (bit e of a byte = element e of that chunk)"""
    from msf_wsi_amd import kernels as kn

    b = kn.gate_unpack(bits, M, Nout, torch.bfloat16).cpu().to(torch.int32)  # (16-bit types: 8 elements per chunk)
    return ((b.unsqueeze(-1) >> torch.arange(8)) & 1).bool().view(M, Nout)


# N, H, W, k (operand channels), Nout: ragged last panel (M % 128 != 0), every supported k, one wave idle (Nout 96 -> unsupported)
GEOMS = [
    (3, 7, 9, 64, 256),
    (2, 14, 14, 128, 512),
    (5, 7, 7, 256, 1024),
    (3, 5, 5, 512, 2048),
    (1, 11, 13, 256, 160),   # 5 output blocks: waves with 2 and 1 blocks
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("pro", [False, True])
@pytest.mark.parametrize("with_ident", [False, True])
def test_panel_fwd_post(hip_lib, dt, geom, pro, with_ident):
    from msf_wsi_amd import kernels as kn

    N, H, W, Cn, K = geom
    M = N * H * W
    g = torch.Generator().manual_seed(11)
    x = rnd((M, Cn), dt, g)
    w = rnd((K, Cn), dt, g, 1.0 / math.sqrt(Cn))
    ident = rnd((M, K), dt, g)
    ps = torch.rand(K, generator=g) + 0.5
    pb = torch.randn(K, generator=g) * 0.2
    sc = torch.rand(Cn, generator=g) + 0.5
    sh = torch.randn(Cn, generator=g) * 0.3
    d = kn.conv_desc(dt, N, H, W, Cn, K, 1, 1, 1, 0)
    assert kn.panel_supported(d, False)
    wd = w.cuda()
    wpk = kn.panel_pack_weights(wd, torch.empty_like(wd), K, Cn, Cn, 1)
    y = torch.empty(M, K, dtype=dt, device="cuda")
    bits = kn.gate_bytes(M, K, dt, "cuda")
    assert kn.panel_fwd_post(d, x.cuda(), wpk, y, ps.cuda(), pb.cuda(), pro=(sc.cuda(), sh.cuda()) if pro else None,
                             ident=ident.cuda() if with_ident else None, relu=True, gate_out=bits)
    torch.cuda.synchronize()
    # fp64 reference of the same chain (the normalised operand rounded to the storage type, as the kernel stages it)
    a = x.double()
    if pro:
        a = torch.relu(a * sc.double() + sh.double()).to(dt).double()
    ref = (a @ w.double().t()) * ps.double() + pb.double()
    if with_ident:
        ref = ref + ident.double()
    ref = torch.relu(ref)
    assert rel(y, ref) < tol(dt)
    # gate bits are the sign of the STORED output, bit for bit
    assert torch.equal(unpack_bits(bits, M, K), y.cpu().float() > 0)
    # and the gather kernel's epilogue gives the same tensor up to the summation order
    y2 = torch.empty_like(y)
    bits2 = kn.gate_bytes(M, K, dt, "cuda")
    xin = x.cuda()
    if pro:
        xin = torch.empty_like(xin)
        kn.bn_act(x.cuda(), sc.cuda(), sh.cuda(), xin, relu=True)
    kn.conv_fwd_post(d, xin, wd, y2, ps.cuda(), pb.cuda(), ident=ident.cuda() if with_ident else None, relu=True,
                     gate_out=bits2)
    torch.cuda.synchronize()
    assert rel(y, y2.float()) < tol(dt)
    assert torch.equal(unpack_bits(bits2, M, K), y2.cpu().float() > 0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("bnbwd", [False, True])
@pytest.mark.parametrize("epi", ["plain", "resid", "bits", "bits+resid+gap"])
def test_panel_dgrad(hip_lib, dt, geom, bnbwd, epi):
    from msf_wsi_amd import kernels as kn

    N, H, W, Kc, Cin = geom  # forward conv: x [.., Cin] -> y [.., Kc]; the gradient maps dY [M, Kc] -> dX [M, Cin]
    M = N * H * W
    g = torch.Generator().manual_seed(12)
    dy = rnd((M, Kc), dt, g, 0.1)
    c = rnd((M, Kc), dt, g)
    w = rnd((Kc, Cin), dt, g, 1.0 / math.sqrt(Kc))
    k1 = torch.rand(Kc, generator=g) + 0.5
    k2 = torch.randn(Kc, generator=g) * 0.05
    k3 = torch.randn(Kc, generator=g) * 0.05
    resid = rnd((M, Cin), dt, g, 0.1)
    gapg = rnd((N, Cin), dt, g, 0.1)
    bits = kn.gate_pack(torch.randint(0, 256, (M, Cin // 8), dtype=torch.uint8, generator=g), Cin, dt)
    d = kn.conv_desc(dt, N, H, W, Cin, Kc, 1, 1, 1, 0)
    assert kn.panel_supported(d, True)
    wd = w.cuda()
    wpk = kn.panel_pack_weights(wd, torch.empty_like(wd), Cin, Kc, 1, Cin)  # W read as [k = Kc][n = Cin]
    dx = torch.empty(M, Cin, dtype=dt, device="cuda")
    dc_out = torch.empty(M, Kc, dtype=dt, device="cuda") if bnbwd else None
    kw = {}
    if "resid" in epi:
        kw["resid"] = resid.cuda()
    if "gap" in epi:
        kw.update(gapg=gapg.cuda(), gap_scale=1.0 / (H * W))
    sums = None
    if "bits" in epi:
        sums = kn.new_stats(Cin, 2, "cuda")
        kw.update(mask_bits=bits.cuda(), sums=sums)
    assert kn.panel_dgrad(d, dy.cuda(), wpk, dx, bnbwd=(c.cuda(), k1.cuda(), k2.cuda(), k3.cuda()) if bnbwd else None,
                          dc_out=dc_out, **kw)
    torch.cuda.synchronize()
    dc = dy.double()
    if bnbwd:
        dc = (k1.double() * dy.double() + k2.double() * c.double() + k3.double()).to(dt).double()
        # the written-back operand is exactly msfwsi_bn_bwd_apply's result
        ref_dc = torch.empty(M, Kc, dtype=dt, device="cuda")
        kn.bn_bwd_apply(dy.cuda(), c.cuda(), k1.cuda(), k2.cuda(), k3.cuda(), ref_dc)
        torch.cuda.synchronize()
        assert torch.equal(dc_out.cpu(), ref_dc.cpu())
        assert rel(dc_out, dc) < tol(dt)
    ref = dc @ w.double()
    if "resid" in epi:
        ref = ref + resid.double()
    if "gap" in epi:
        ref = ref + (gapg.double() / (H * W)).repeat_interleave(H * W, dim=0)
    if "bits" in epi:
        ref = ref * unpack_bits(bits, M, Cin).double()
    assert rel(dx, ref) < tol(dt)
    if sums is not None:
        # column sums of the gated gradient (taken in fp32 before the store's rounding): within the rounding noise of M
        # storage-type values of the fp64 sums
        got = sums.sum(dim=0)[0].cpu()
        want = ref.sum(dim=0)
        eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        bound = 4.0 * eps * math.sqrt(M) * ref.pow(2).mean().sqrt().item() + 1e-6
        assert (got - want).abs().max().item() <= bound
        assert sums[:, 1].abs().max().item() == 0.0  # slot 1 is left alone


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", [(3, 8, 8, 128, 256), (2, 14, 14, 256, 512), (2, 6, 10, 64, 128)])
def test_panel_dgrad_lowres_residual(hip_lib, dt, geom):
    """the strided-downsample residual, added on the even pixels only, equals msfwsi_conv_dgrad's resid_stride = 2"""
    from msf_wsi_amd import kernels as kn

    N, H, W, Kc, Cin = geom
    M = N * H * W
    g = torch.Generator().manual_seed(13)
    dy = rnd((M, Kc), dt, g, 0.1).cuda()
    w = rnd((Kc, Cin), dt, g, 1.0 / math.sqrt(Kc)).cuda()
    lo = rnd((N * (H // 2) * (W // 2), Cin), dt, g, 0.1).cuda()
    gapg = rnd((N, Cin), dt, g, 0.1).cuda()
    bits = kn.gate_pack(torch.randint(0, 256, (M, Cin // 8), dtype=torch.uint8, generator=g), Cin, dt).cuda()
    d = kn.conv_desc(dt, N, H, W, Cin, Kc, 1, 1, 1, 0)
    wpk = kn.panel_pack_weights(w, torch.empty_like(w), Cin, Kc, 1, Cin)
    a, b = torch.empty(M, Cin, dtype=dt, device="cuda"), torch.empty(M, Cin, dtype=dt, device="cuda")
    sa, sb = kn.new_stats(Cin, 2, "cuda"), kn.new_stats(Cin, 2, "cuda")
    assert kn.panel_dgrad(d, dy, wpk, a, resid=lo, resid_stride=2, gapg=gapg, gap_scale=1.0 / (H * W), mask_bits=bits, sums=sa)
    kn.conv_dgrad(d, dy, w, b, resid=lo, resid_stride=2, gapg=gapg, gap_scale=1.0 / (H * W), mask_bits=bits, sums=sb)
    torch.cuda.synchronize()
    # fp64 reference
    full = torch.zeros(N, H, W, Cin, dtype=torch.float64)
    full[:, ::2, ::2, :] = lo.cpu().double().view(N, H // 2, W // 2, Cin)
    ref = dy.cpu().double() @ w.cpu().double() + full.view(M, Cin) + (gapg.cpu().double() / (H * W)).repeat_interleave(H * W, dim=0)
    ref = ref * unpack_bits(bits, M, Cin).double()
    assert rel(a, ref) < tol(dt)
    assert rel(a, b.float()) < tol(dt)
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    bound = 4.0 * eps * math.sqrt(M) * ref.pow(2).mean().sqrt().item() + 1e-6
    assert (sa.sum(0)[0].cpu() - ref.sum(0)).abs().max().item() <= bound
    assert (sb.sum(0)[0].cpu() - ref.sum(0)).abs().max().item() <= bound


def test_panel_unsupported_shapes(hip_lib):
    """fp32, ragged k, narrow outputs and non-1x1 geometry are refused (the engine then keeps the gather kernel)"""
    from msf_wsi_amd import kernels as kn

    ok = kn.conv_desc(torch.bfloat16, 2, 7, 7, 256, 1024, 1, 1, 1, 0)
    assert kn.panel_supported(ok, False) and kn.panel_supported(ok, True) is False  # dgrad: k = 1024 is not a panel width
    assert not kn.panel_supported(kn.conv_desc(torch.float32, 2, 7, 7, 256, 1024, 1, 1, 1, 0), False)
    assert not kn.panel_supported(kn.conv_desc(torch.bfloat16, 2, 7, 7, 96, 1024, 1, 1, 1, 0), False)
    assert not kn.panel_supported(kn.conv_desc(torch.bfloat16, 2, 7, 7, 256, 64, 1, 1, 1, 0), False)
    assert not kn.panel_supported(kn.conv_desc(torch.bfloat16, 2, 7, 7, 256, 256, 3, 3, 1, 1), False)
    assert not kn.panel_supported(kn.conv_desc(torch.bfloat16, 2, 8, 8, 256, 512, 1, 1, 2, 0), False)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", [(64, 14, 14, 256, 1024), (32, 28, 28, 128, 512), (128, 7, 7, 512, 2048), (16, 56, 56, 64, 256)])
def test_hand_counted_waits_equal_compiler_waits(hip_lib, dt, geom):
    """whole panels at production channel counts (hundreds of workgroups, every wave with several blocks): the
    hand-counted `s_waitcnt vmcnt(N)` instances (msfwsi_set_tuning(17, 1), the default) give the SAME BITS as the instances
    whose loads hipcc counts itself -- a register read before its load has landed would differ (the dynamic counterpart of
    tools/check_hand_waits.py's static audit), forward with the fused BatchNorm+ReLU prologue and input gradient with the
    fused BatchNorm backward, three times each"""
    from msf_wsi_amd import kernels as kn

    N, H, W, Cn, Kw = geom
    M = N * H * W
    assert M % 128 == 0
    g = torch.Generator().manual_seed(14)
    a = rnd((M, Cn), dt, g).cuda()
    c1 = rnd((M, Cn), dt, g).cuda()
    w = rnd((Kw, Cn), dt, g, 1.0 / math.sqrt(Cn)).cuda()
    w1 = rnd((Cn, Kw), dt, g, 1.0 / math.sqrt(Cn)).cuda()
    ident = rnd((M, Kw), dt, g).cuda()
    ps, pb = (torch.rand(Kw, generator=g) + 0.5).cuda(), (torch.randn(Kw, generator=g) * 0.2).cuda()
    sc, sh = (torch.rand(Cn, generator=g) + 0.5).cuda(), (torch.randn(Cn, generator=g) * 0.3).cuda()
    k3 = (torch.randn(Cn, generator=g) * 0.05).cuda()
    bits_in = kn.gate_pack(torch.randint(0, 256, (M, Kw // 8), dtype=torch.uint8, generator=g), Kw, dt).cuda()
    d = kn.conv_desc(dt, N, H, W, Cn, Kw, 1, 1, 1, 0)
    d1 = kn.conv_desc(dt, N, H, W, Kw, Cn, 1, 1, 1, 0)
    wpk = kn.panel_pack_weights(w, torch.empty_like(w), Kw, Cn, Cn, 1)
    wpk1 = kn.panel_pack_weights(w1, torch.empty_like(w1), Kw, Cn, 1, Kw)

    def run():
        y = torch.empty(M, Kw, dtype=dt, device="cuda")
        bits = kn.gate_bytes(M, Kw, dt, "cuda")
        assert kn.panel_fwd_post(d, a, wpk, y, ps, pb, pro=(sc, sh), ident=ident, relu=True, gate_out=bits)
        dx = torch.empty(M, Kw, dtype=dt, device="cuda")
        dc = torch.empty(M, Cn, dtype=dt, device="cuda")
        sums = kn.new_stats(Kw, 2, "cuda")
        assert kn.panel_dgrad(d1, a, wpk1, dx, bnbwd=(c1, sc, sh, k3), dc_out=dc, resid=ident, mask_bits=bits_in, sums=sums)
        torch.cuda.synchronize()
        return y, bits, dx, dc, sums.sum(0)[0]

    try:
        assert hip_lib.msfwsi_set_tuning(17, 0) == 0
        ref = run()
        assert hip_lib.msfwsi_set_tuning(17, 1) == 0
        for _ in range(3):
            got = run()
            for name, r, t in zip(("y", "gate bits", "dx", "dc"), ref[:4], got[:4]):
                assert torch.equal(r, t), f"{name}: hand-counted instance differs from the compiler-counted one"
            # (the column sums are added by fp64 atomics in an order that varies: equal to fp64 rounding)
            assert (ref[4] - got[4]).abs().max().item() <= 1e-9 * max(1.0, ref[4].abs().max().item())
    finally:
        hip_lib.msfwsi_set_tuning(17, 1)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", [(32, 28, 28, 128, 512), (16, 56, 56, 64, 256), (3, 9, 7, 128, 192)])
def test_wide_blocks_equal_narrow_blocks(hip_lib, dt, geom):
    """the wide block form (k <= 128: a wave owns 64 channels of 64 rows, msfwsi_set_tuning(18, 1), the default) against
    the 32-channel form: every output element is the same MFMA chain over the same k order, so outputs, gate bytes and
    the written-back BatchNorm-backward operand are the SAME BITS; the column sums meet in another order (fp32 partials of
    two waves instead of one: equal to fp32 rounding).  Whole panels (hand-counted instances) and a ragged one"""
    from helpers import tuned
    from msf_wsi_amd import kernels as kn

    N, H, W, Cn, Kw = geom
    M = N * H * W
    g = torch.Generator().manual_seed(16)
    a = rnd((M, Cn), dt, g).cuda()
    c1 = rnd((M, Cn), dt, g).cuda()
    w = rnd((Kw, Cn), dt, g, 1.0 / math.sqrt(Cn)).cuda()
    w1 = rnd((Cn, Kw), dt, g, 1.0 / math.sqrt(Cn)).cuda()
    ident = rnd((M, Kw), dt, g).cuda()
    ps, pb = (torch.rand(Kw, generator=g) + 0.5).cuda(), (torch.randn(Kw, generator=g) * 0.2).cuda()
    sc, sh = (torch.rand(Cn, generator=g) + 0.5).cuda(), (torch.randn(Cn, generator=g) * 0.3).cuda()
    k3 = (torch.randn(Cn, generator=g) * 0.05).cuda()
    bits_in = kn.gate_pack(torch.randint(0, 256, (M, Kw // 8), dtype=torch.uint8, generator=g), Kw, dt).cuda()
    d = kn.conv_desc(dt, N, H, W, Cn, Kw, 1, 1, 1, 0)
    d1 = kn.conv_desc(dt, N, H, W, Kw, Cn, 1, 1, 1, 0)
    wpk = kn.panel_pack_weights(w, torch.empty_like(w), Kw, Cn, Cn, 1)
    wpk1 = kn.panel_pack_weights(w1, torch.empty_like(w1), Kw, Cn, 1, Kw)

    def run():
        y = torch.empty(M, Kw, dtype=dt, device="cuda")
        bits = kn.gate_bytes(M, Kw, dt, "cuda")
        assert kn.panel_fwd_post(d, a, wpk, y, ps, pb, pro=(sc, sh), ident=ident, relu=True, gate_out=bits)
        dx = torch.empty(M, Kw, dtype=dt, device="cuda")
        dc = torch.empty(M, Cn, dtype=dt, device="cuda")
        sums = kn.new_stats(Kw, 2, "cuda")
        assert kn.panel_dgrad(d1, a, wpk1, dx, bnbwd=(c1, sc, sh, k3), dc_out=dc, resid=ident, mask_bits=bits_in, sums=sums)
        torch.cuda.synchronize()
        return y, kn.gate_unpack(bits, M, Kw, dt), dx, dc, sums.sum(0)[0]

    with tuned(hip_lib, {18: 0}):
        ref = run()
    got = run()
    for name, r, t in zip(("y", "gate bits", "dx", "dc"), ref[:4], got[:4]):
        assert torch.equal(r, t), f"{name}: the wide form differs from the 32-channel form"
    assert (ref[4] - got[4]).abs().max().item() <= 1e-5 * max(1.0, ref[4].abs().max().item())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", [(3, 7, 9, 64), (2, 14, 14, 128), (700, 8, 8, 64), (300, 8, 8, 128)])
def test_panel_gram(hip_lib, dt, geom):
    """Gram matrix + column sums of relu(scale*c + shift) in one pass over the raw conv output (msfwsi_panel_gram): against
    fp64 on the rounded activation, and against the two-pass form it replaces (msfwsi_bn_act_sum + msfwsi_gram); ragged last
    panel, more panels than resident workgroups (persistent loop)"""
    from msf_wsi_amd import kernels as kn

    N, H, W, Cn = geom
    M = N * H * W
    g = torch.Generator().manual_seed(15)
    c = rnd((M, Cn), dt, g).cuda()
    sc = (torch.rand(Cn, generator=g) + 0.5).cuda()
    sh = (torch.randn(Cn, generator=g) * 0.3).cuda()
    A = kn.zeros((Cn, 1, 1, Cn), torch.float32, "cuda")
    sa = kn.zeros((Cn,), torch.float64, "cuda")
    assert kn.panel_gram(c, sc, sh, A, sa)
    a2 = torch.empty_like(c)
    sb = kn.zeros((Cn,), torch.float64, "cuda")
    kn.bn_act_sum(c, sc, sh, a2, sb)
    B = kn.zeros((Cn, 1, 1, Cn), torch.float32, "cuda")
    kn.gram(kn.conv_desc(dt, M, 1, 1, Cn, Cn, 1, 1, 1, 0), a2, B)
    torch.cuda.synchronize()
    a64 = a2.double().cpu()  # the rounded activation both forms multiply
    ref = a64.t() @ a64
    assert rel(A.view(Cn, Cn), ref) < 1e-6
    assert rel(B.view(Cn, Cn), ref) < 1e-6
    assert rel(sa, a64.sum(0)) < 1e-6 and rel(sb, a64.sum(0)) < 1e-6
    assert not kn.panel_gram(rnd((64, 256), dt, g).cuda(), torch.ones(256).cuda(), torch.zeros(256).cuda(),
                             kn.zeros((256, 1, 1, 256), torch.float32, "cuda"), kn.zeros((256,), torch.float64, "cuda"))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", [(8, 16, 16, 128, 256), (4, 16, 16, 256, 512), (16, 12, 12, 64, 256)])
@pytest.mark.parametrize("lowres", [False, True])
def test_panel_dgrad_pooled_gradient_from_lds(hip_lib, dt, geom, lowres):
    """whole panels with the pooled-feature gradient (the stage-boundary launches of the encoder backward): the
    hand-counted instance reads gapg from the two image rows staged in LDS; same bits as the compiler-counted instance
    that reads it from memory, and both against fp64"""
    from msf_wsi_amd import kernels as kn

    N, H, W, Kc, Cin = geom
    M = N * H * W
    assert M % 128 == 0 and H * W >= 128
    g = torch.Generator().manual_seed(16)
    dy = rnd((M, Kc), dt, g, 0.1).cuda()
    w = rnd((Kc, Cin), dt, g, 1.0 / math.sqrt(Kc)).cuda()
    full = rnd((M, Cin), dt, g, 0.1).cuda()
    lo = rnd((N * (H // 2) * (W // 2), Cin), dt, g, 0.1).cuda()
    gapg = rnd((N, Cin), dt, g, 0.5).cuda()
    bits = kn.gate_pack(torch.randint(0, 256, (M, Cin // 8), dtype=torch.uint8, generator=g), Cin, dt).cuda()
    d = kn.conv_desc(dt, N, H, W, Cin, Kc, 1, 1, 1, 0)
    wpk = kn.panel_pack_weights(w, torch.empty_like(w), Cin, Kc, 1, Cin)
    kw = dict(resid=lo, resid_stride=2) if lowres else dict(resid=full)

    def run():
        dx = torch.empty(M, Cin, dtype=dt, device="cuda")
        sums = kn.new_stats(Cin, 2, "cuda")
        assert kn.panel_dgrad(d, dy, wpk, dx, gapg=gapg, gap_scale=1.0 / (H * W), mask_bits=bits, sums=sums, **kw)
        torch.cuda.synchronize()
        return dx, sums.sum(0)[0]

    try:
        assert hip_lib.msfwsi_set_tuning(17, 0) == 0
        ref_dx, ref_s = run()
        assert hip_lib.msfwsi_set_tuning(17, 1) == 0
        dx, s = run()
    finally:
        hip_lib.msfwsi_set_tuning(17, 1)
    assert torch.equal(dx, ref_dx)
    assert (s - ref_s).abs().max().item() <= 1e-9 * max(1.0, ref_s.abs().max().item())
    res = torch.zeros(N, H, W, Cin, dtype=torch.float64)
    if lowres:
        res[:, ::2, ::2, :] = lo.cpu().double().view(N, H // 2, W // 2, Cin)
    else:
        res = full.cpu().double().view(N, H, W, Cin)
    ref = dy.cpu().double() @ w.cpu().double() + res.reshape(M, Cin) + (gapg.cpu().double() / (H * W)).repeat_interleave(H * W, dim=0)
    ref = ref * unpack_bits(bits, M, Cin).double()
    assert rel(dx, ref) < tol(dt)


@pytest.mark.parametrize("K", [64, 128])
def test_panel_gram_sums_repeat_bit_for_bit(hip_lib, K):
    """the fused Gram pass on the same operand twenty times: Gram matrix AND column sums equal bit for bit (the column sums
    of round 5 met in LDS through fp32 atomic adds in arrival order: last-bit differences from run to run that the 16-bit
    network amplified -- tests/test_fixes_gpu.py::test_step_is_reproducible_run_to_run_with_one_split)"""
    from msf_wsi_amd import kernels as kn

    g = torch.Generator().manual_seed(K)
    M = 128 * 515 + 37                      # several panels per persistent workgroup, a ragged last one
    c = (torch.randn(M, K, generator=g) * 1.7).to(torch.bfloat16).cuda()
    sc, sh = (torch.rand(K, generator=g) + 0.5).cuda(), (torch.randn(K, generator=g) * 0.3).cuda()
    first = None
    for _ in range(20):
        A = torch.zeros(K, K, dtype=torch.float32, device="cuda")
        sa = torch.zeros(K, dtype=torch.float64, device="cuda")
        assert kn.panel_gram(c, sc, sh, A, sa)
        torch.cuda.synchronize()
        # (cross-workgroup sums are fp64 atomics: their order moves the result by ~1e-16, below the fp32 values the
        #  engine rounds them to)
        cur = (A.clone(), sa.float().clone())
        if first is None:
            first = cur
        else:
            assert torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1])
