"""The kernel instances and tensor sizes that BASELINE config 2 (ResNet-50-derived, bf16, 256 tile pairs, 1 GPU) actually
runs -- everything the small-shape tests of test_kernels_gpu.py never reach:

  * the 256x128 / 8-wave `igemm_dma_kernel` tile (dispatch_tile picks it only on grids >= 1024 tiles; here it is forced
    with msfwsi_set_tuning(0, 1)) in every epilogue class: forward + statistics (EPI 0), forward + BatchNorm apply +
    identity + ReLU + gate bits (EPI 1), its two-source form, input gradient plain / gated by activation / gated by
    bits (EPI 0), two-source input gradient, strided residual (EPI 3) -- against torch fp64 on the CPU;
  * the 256x256 / 16-wave weight-gradient tile of the deep layers against the 128x128 one;
  * the weight-gradient kernel at production pixel counts (hundreds of pixel splits, linear-addressing path);
  * tensors beyond 2^31 BYTES (layer-1 activations of config 2 are 6.6 GB): the kernels address operands with 32-bit
    per-lane byte offsets against a per-workgroup 64-bit base, so a wrap would show as wrong images -- checked on whole
    images below, at and far beyond the 2^31-byte line;
  * the full-size config-2 step: finite loss, two BatchNorm updates per step, folded (Gram-matrix) statistics agree
    with the explicit ones at 12.8 M pixels per channel.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import test_kernels_gpu as tk
from test_kernels_gpu import nhwc, rel, rnd, tol

pytestmark = pytest.mark.gpu
LOWP = [torch.bfloat16, torch.float16]


@pytest.fixture
def big_tile(hip_lib):
    """force the production tile (256x128, 8 waves) on every 16-bit grid with more than 64 output channels"""
    hip_lib.msfwsi_set_tuning(0, 1)
    yield hip_lib
    hip_lib.msfwsi_set_tuning(0, 1024)


BIG_CONVS = [
    # N, H, W, C, K, R, stride, pad   (K > 64 and C > 64: both directions take the 256x128 tile)
    (2, 40, 40, 128, 256, 1, 1, 0),    # ragged M: 3200 pixels = 12.5 tiles
    (2, 28, 28, 128, 128, 3, 1, 1),
    (3, 27, 29, 128, 256, 3, 2, 1),    # stride-2 3x3 (conv2 of the first block of a stage), odd extents
    (3, 20, 20, 256, 512, 1, 2, 0),    # strided 1x1 (downsample branch)
    (1, 9, 9, 256, 200, 3, 1, 1),      # output channels not a tile multiple
]


@pytest.mark.parametrize("dt", LOWP)
@pytest.mark.parametrize("geom", BIG_CONVS)
def test_big_tile_conv_fwd_stats(big_tile, dt, geom):
    tk.test_conv_fwd(big_tile, dt, geom, False)


@pytest.mark.parametrize("dt", LOWP)
@pytest.mark.parametrize("geom", BIG_CONVS)
def test_big_tile_conv_dgrad(big_tile, dt, geom):
    tk.test_conv_dgrad(big_tile, dt, geom)


@pytest.mark.parametrize("dt", LOWP)
@pytest.mark.parametrize("geom", BIG_CONVS[:4])
def test_big_tile_conv_dgrad_gated(big_tile, dt, geom):
    tk.test_conv_dgrad_fused_activation_backward(big_tile, dt, geom)


@pytest.mark.parametrize("dt", LOWP)
@pytest.mark.parametrize("geom", [(2, 28, 28, 128, 512, 1, 1, 0), (3, 15, 17, 256, 200, 1, 1, 0)])
@pytest.mark.parametrize("with_ident", [True, False])
def test_big_tile_conv_fwd_post(big_tile, dt, geom, with_ident):
    tk.test_conv_fwd_post(big_tile, dt, geom, with_ident)


@pytest.mark.parametrize("dt", LOWP)
@pytest.mark.parametrize("shape", [(3, 30, 64, 256, 256), (2, 21, 128, 512, 200)])
def test_big_tile_conv_fwd_post2(big_tile, dt, shape):
    tk.test_conv_fwd_post2_two_sources(big_tile, dt, shape)


@pytest.mark.parametrize("dt", LOWP)
@pytest.mark.parametrize("shape", [(3, 30, 256, 128, 128), (2, 21, 1024, 256, 256)])
def test_big_tile_conv_dgrad2(big_tile, dt, shape):
    tk.test_conv_dgrad2_two_sources(big_tile, dt, shape)


@pytest.mark.parametrize("dt", LOWP)
def test_big_tile_gate_bits_and_lowres_residual(big_tile, dt):
    tk.test_gate_bits_roundtrip(big_tile, dt)               # K = 208 > 64: EPI 1 gate bits out, EPI 0 gate bits in
    tk.test_conv_dgrad_lowres_residual(big_tile, dt, (18, 22))  # EPI 3


@pytest.mark.parametrize("dt", LOWP)
@pytest.mark.parametrize("geom", [
    # N, H, W, C, K, R      (stride 1, same-size output: the linear-addressing path; M = 100 k .. 200 k pixels)
    (64, 56, 56, 64, 64, 1),
    (64, 56, 56, 64, 64, 3),
    (32, 56, 56, 256, 64, 1),
    (64, 28, 28, 128, 128, 3),
    (128, 28, 28, 512, 128, 1),
])
def test_wgrad_production_pixel_counts(hip_lib, dt, geom):
    """hundreds of pixel splits accumulating through fp32 atomics, both tile shapes, 1x1 and 3x3"""
    from msf_wsi_amd import kernels as kn

    N, H, W, Cc, K, R = geom
    pad = R // 2
    g = torch.Generator().manual_seed(31)
    x = rnd((N, Cc, H, W), dt, g)
    dy = rnd((N, K, H, W), dt, g, 0.05)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    ref = torch.nn.grad.conv2d_weight(x.double(), (K, Cc, R, R), dy.double(), stride=1, padding=pad)
    d = kn.conv_desc(dt, N, H, W, Cc, K, R, R, 1, pad)
    dw = torch.zeros(K, R, R, Cc, device="cuda")
    kn.conv_wgrad(d, nhwc(x).to(dt).cuda(), nhwc(dy).to(dt).cuda(), dw)
    torch.cuda.synchronize()
    # inputs are exact in the storage type and products accumulate in fp32: only the summation order differs
    assert rel(dw.cpu().permute(0, 3, 1, 2), ref) < 2e-5


@pytest.mark.parametrize("dt", LOWP)
@pytest.mark.parametrize("geom", [
    # N, H, W, C, K, R, stride   (both extents whole multiples of 256: the 256 x 256 tile)
    (64, 14, 14, 256, 256, 3, 1),     # linear addressing, image-border taps, 9 column tiles
    (64, 14, 14, 1024, 256, 1, 1),    # 1x1, four column tiles
    (96, 7, 7, 512, 2048, 1, 1),      # eight row tiles
    (48, 14, 14, 256, 256, 3, 2),     # stride 2: generic staging
    (3, 7, 7, 256, 512, 1, 1),        # fewer pixels than one slab per resident workgroup
])
def test_wgrad_big_tile(hip_lib, dt, geom):
    """the 256 x 256 / 16-wave weight-gradient instance against fp64, with the 128 x 128 instance on the same operands
    beside it"""
    from msf_wsi_amd import kernels as kn

    N, H, W, Cc, K, R, st = geom
    pad = R // 2
    g = torch.Generator().manual_seed(37)
    d = kn.conv_desc(dt, N, H, W, Cc, K, R, R, st, pad)
    x = rnd((N, Cc, H, W), dt, g)
    dy = rnd((N, K, d.P, d.Q), dt, g, 0.05)
    ref = torch.nn.grad.conv2d_weight(x.double(), (K, Cc, R, R), dy.double(), stride=st, padding=pad)
    xd, dyd = nhwc(x).to(dt).cuda(), nhwc(dy).to(dt).cuda()
    got = {}
    try:
        for big in (1, 0):
            hip_lib.msfwsi_set_tuning(6, big)
            dw = torch.zeros(K, R, R, Cc, device="cuda")
            kn.conv_wgrad(d, xd, dyd, dw)
            torch.cuda.synchronize()
            got[big] = dw.cpu().permute(0, 3, 1, 2)
    finally:
        hip_lib.msfwsi_set_tuning(6, 1)
    for key, val in got.items():
        assert rel(val, ref) < 2e-5, key


@pytest.mark.parametrize("dt", LOWP)
def test_conv3x3_stationary_persistent_ranges(hip_lib, dt):
    """the weights-stationary 64 -> 64 kernel with several tiles per persistent workgroup (784 tiles on 256 CUs), forward
    with BatchNorm sums and gated input gradient, against fp64 and against the gather kernels on the same operands"""
    from msf_wsi_amd import kernels as kn

    N, H, W, Cc = 64, 56, 56, 64
    g = torch.Generator().manual_seed(41)
    d = kn.conv_desc(dt, N, H, W, Cc, Cc, 3, 3, 1, 1)
    assert kn.conv3x3_stationary(d)
    x = rnd((N, Cc, H, W), dt, g)
    w = rnd((Cc, Cc, 3, 3), dt, g, 1.0 / math.sqrt(Cc * 9))
    ref = F.conv2d(x.double(), w.double(), None, stride=1, padding=1)
    xd, wd = nhwc(x).to(dt).cuda(), nhwc(w).to(dt).cuda()
    y, y2 = (torch.empty(N, H, W, Cc, dtype=dt, device="cuda") for _ in range(2))
    st = kn.new_stats(Cc)
    kn.conv3x3_fwd(d, xd, wd, y, stats=st)
    kn.conv_fwd(d, xd, wd, y2)
    torch.cuda.synchronize()
    assert rel(y.float().cpu().permute(0, 3, 1, 2), ref) < tol(dt)
    assert rel(y.float(), y2.float()) < 1e-3  # same products, another summation order, one rounding to the storage type
    yy = y.double().reshape(-1, Cc)
    assert torch.allclose(st.sum(0)[0], yy.sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(st.sum(0)[1], (yy * yy).sum(0), rtol=1e-5, atol=1e-3)
    # input gradient, gated by the producer's activation, with its BatchNorm-backward sums
    dy = rnd((N, Cc, H, W), dt, g)
    c = rnd((N, Cc, H, W), dt, g)
    sc, sh = torch.rand(Cc, generator=g) - 0.3, torch.randn(Cc, generator=g) * 0.3
    refd = torch.nn.grad.conv2d_input((N, Cc, H, W), w.double(), dy.double(), stride=1, padding=1)
    refd = refd * ((c * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) > 0)
    dyd, cd = nhwc(dy).to(dt).cuda(), nhwc(c).to(dt).cuda()
    dx = torch.empty(N, H, W, Cc, dtype=dt, device="cuda")
    sums = kn.new_stats(Cc)
    kn.conv3x3_dgrad(d, dyd, wd, dx, mask=(cd, sc.cuda(), sh.cuda()), sums=sums)
    torch.cuda.synchronize()
    assert rel(dx.float().cpu().permute(0, 3, 1, 2), refd) < 2 * tol(dt)
    gd = dx.double().reshape(-1, Cc)
    assert torch.allclose(sums.sum(0)[0], gd.sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(sums.sum(0)[1], (gd * cd.double().reshape(-1, Cc)).sum(0), rtol=1e-5, atol=1e-3)
    # weight gradient: output-stationary kernel (N = 64: its size threshold lifted) with the producer's BatchNorm + ReLU
    # fused into the staging, against fp64 on the materialised activation
    a = F.relu(c * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).to(dt).float()
    refw = torch.nn.grad.conv2d_weight(a.double(), (Cc, Cc, 3, 3), dy.double(), stride=1, padding=1)
    dw = torch.zeros(Cc, 3, 3, Cc, device="cuda")
    try:
        hip_lib.msfwsi_set_tuning(11, 0)
        assert kn.conv_wgrad_stationary(d)
        kn.conv_wgrad(d, cd, dyd, dw, pro=(sc.cuda(), sh.cuda()))
        torch.cuda.synchronize()
    finally:
        hip_lib.msfwsi_set_tuning(11, 32 * 256 * 256)
    assert rel(dw.cpu().permute(0, 3, 1, 2), refw) < 2e-5


# ------------------------------------------------------------------------------------------------
# beyond 2^31 bytes
# ------------------------------------------------------------------------------------------------
def _imgs(n_img, bytes_per_img):
    """image indices below, just around and far beyond the 2^31-byte line of a tensor"""
    line = (1 << 31) // bytes_per_img
    assert n_img * bytes_per_img > (1 << 31) + 4 * bytes_per_img
    return sorted({0, 1, line - 1, line, line + 1, (n_img + line) // 2, n_img - 1})


def _rand_nhwc(shape, dt, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    out = torch.empty(shape, dtype=dt, device="cuda")
    step = max(1, (1 << 28) // (out[0].numel()))
    for i in range(0, shape[0], step):  # chunks: no multi-GB fp32 temporary
        out[i:i + step] = (torch.randn((min(step, shape[0] - i),) + tuple(shape[1:]), generator=g, device="cuda")
                           * scale).to(dt)
    return out


def _nchw(t):  # NHWC device tensor (a few images) -> NCHW fp64 on the CPU
    return t.permute(0, 3, 1, 2).double().cpu()


@pytest.fixture
def freed():
    yield
    torch.cuda.empty_cache()


def test_over_2gib_conv1x1_fwd_and_post(hip_lib, freed):
    """layer-1 shapes of config 2: N = 4096 images of 56x56, 256 <-> 64 channels, bf16 (6.6 GB / 1.6 GB tensors)"""
    from msf_wsi_amd import kernels as kn

    dt = torch.bfloat16
    N, H, Cw, Cx = 4096, 56, 64, 256
    x = _rand_nhwc((N, H, H, Cx), dt, 1)                      # 6.6 GB
    g = torch.Generator().manual_seed(2)
    w = rnd((Cw, Cx, 1, 1), dt, g, 1 / 16.0)
    d = kn.conv_desc(dt, N, H, H, Cx, Cw, 1, 1, 1, 0)
    y = torch.empty(N, H, H, Cw, dtype=dt, device="cuda")
    stats = kn.new_stats(Cw)
    kn.conv_fwd(d, x, nhwc(w).to(dt).cuda(), y, stats=stats)   # 256 -> 64 (conv1 of a layer-1 block), EPI 0
    torch.cuda.synchronize()
    pick = _imgs(N, H * H * Cx * 2)
    ref = F.conv2d(_nchw(x[pick]), w.double())
    assert rel(_nchw(y[pick]), ref) < tol(dt)
    yy = y.view(-1, Cw)
    s = stats.sum(0)
    assert torch.allclose(s[0], yy.sum(0, dtype=torch.float64), rtol=1e-6, atol=5e-2)
    # 64 -> 256 with BatchNorm apply + identity + ReLU + gate bits (the fused Bottleneck tail, EPI 1): out 6.6 GB
    w2 = rnd((Cx, Cw, 1, 1), dt, g, 1 / 8.0)
    ps, pb = torch.rand(Cx, generator=g) + 0.5, torch.randn(Cx, generator=g) * 0.3
    d2 = kn.conv_desc(dt, N, H, H, Cw, Cx, 1, 1, 1, 0)
    out = torch.empty(N, H, H, Cx, dtype=dt, device="cuda")
    bits = kn.gate_bytes(N * H * H, Cx, dt)
    kn.conv_fwd_post(d2, y, nhwc(w2).to(dt).cuda(), out, ps.cuda(), pb.cuda(), ident=x, relu=True, gate_out=bits)
    torch.cuda.synchronize()
    pick2 = _imgs(N, H * H * Cx * 2)
    c = F.conv2d(_nchw(y[pick2]), w2.double()).float().to(dt).double()
    ref2 = F.relu(c * ps.double().view(1, -1, 1, 1) + pb.double().view(1, -1, 1, 1) + _nchw(x[pick2]))
    assert rel(_nchw(out[pick2]), ref2) < tol(dt)
    got_bits = kn.gate_unpack(bits, N * H * H, Cx, dt).view(N, H * H, Cx // 8)[pick2].to(torch.int32)
    want = ((out[pick2].reshape(len(pick2), H * H, Cx // 8, 8) > 0).to(torch.int32)
            * (2 ** torch.arange(8, device="cuda", dtype=torch.int32))).sum(-1)
    assert torch.equal(got_bits, want)


def test_over_2gib_conv3x3_fwd_dgrad_wgrad(hip_lib, freed):
    """3x3 / 64 channels at 56x56 with N = 6000 images: 2.4 GB operands on the gather kernel, the halo kernel and the
    linear-addressing weight gradient"""
    from msf_wsi_amd import kernels as kn

    dt = torch.bfloat16
    N, H, Cc = 6000, 56, 64
    x = _rand_nhwc((N, H, H, Cc), dt, 3)
    g = torch.Generator().manual_seed(4)
    w = rnd((Cc, Cc, 3, 3), dt, g, 1 / 24.0)
    wd = nhwc(w).to(dt).cuda()
    d = kn.conv_desc(dt, N, H, H, Cc, Cc, 3, 3, 1, 1)
    pick = _imgs(N, H * H * Cc * 2)
    y = torch.empty_like(x)
    kn.conv_fwd(d, x, wd, y)
    torch.cuda.synchronize()
    assert rel(_nchw(y[pick]), F.conv2d(_nchw(x[pick]), w.double(), padding=1)) < tol(dt)
    # input gradient: gather kernel and (if the library takes the shape) the halo kernel
    dx = torch.empty_like(x)
    kn.conv_dgrad(d, x, wd, dx)
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_input((len(pick), Cc, H, H), w.double(), _nchw(x[pick]), padding=1)
    assert rel(_nchw(dx[pick]), ref) < tol(dt)
    if kn.conv3x3_supported(d):
        dx.zero_()
        kn.conv3x3_dgrad(d, x, wd, dx)
        torch.cuda.synchronize()
        assert rel(_nchw(dx[pick]), ref) < tol(dt)
    # weight gradient: dy is zero except on the picked images, x is dense -- a wrapped address would pair the
    # gradient of an image beyond the line with the activations of another one
    dy = torch.zeros_like(x)
    gsel = _rand_nhwc((len(pick), H, H, Cc), dt, 5, 0.05)
    dy[pick] = gsel
    dw = torch.zeros(Cc, 3, 3, Cc, device="cuda")
    kn.conv_wgrad(d, x, dy, dw)
    torch.cuda.synchronize()
    refw = torch.nn.grad.conv2d_weight(_nchw(x[pick]), (Cc, Cc, 3, 3), _nchw(gsel), padding=1)
    assert rel(dw.cpu().permute(0, 3, 1, 2), refw) < 2e-5


def test_over_2gib_dgrad_gated_and_two_source(hip_lib, freed):
    """the dominant kernels of the profile on 6.6 GB outputs: 64 -> 256 input gradient with the producer's gate +
    BatchNorm sums (EPI 0), with gate bits + pooled-feature gradient, and the two-source form (256 + 64 -> 64)"""
    from msf_wsi_amd import kernels as kn

    dt = torch.bfloat16
    N, H, Cw, Cx = 4096, 56, 64, 256
    g = torch.Generator().manual_seed(6)
    dy = _rand_nhwc((N, H, H, Cw), dt, 7)                     # gradient of conv1's output (1.6 GB)
    w = rnd((Cw, Cx, 1, 1), dt, g, 1 / 8.0)                   # conv1: 256 -> 64
    wd = nhwc(w).to(dt).cuda()
    d = kn.conv_desc(dt, N, H, H, Cx, Cw, 1, 1, 1, 0)
    resid = _rand_nhwc((N, H, H, Cx), dt, 8)                  # 6.6 GB
    yprev = _rand_nhwc((N, H, H, Cx), dt, 9)                  # 6.6 GB: the previous block's output (its ReLU gate)
    one, zero = torch.ones(Cx, device="cuda"), torch.zeros(Cx, device="cuda")
    dx = torch.empty_like(resid)
    sums = kn.new_stats(Cx)
    kn.conv_dgrad(d, dy, wd, dx, resid=resid, mask=(yprev, one, zero), sums=sums)
    torch.cuda.synchronize()
    pick = _imgs(N, H * H * Cx * 2)

    def ref_dx(idx):
        r = torch.nn.grad.conv2d_input((len(idx), Cx, H, H), w.double(), _nchw(dy[idx])) + _nchw(resid[idx])
        return torch.where(_nchw(yprev[idx]) > 0, r.float().to(dt).double(), torch.zeros_like(r))

    assert rel(_nchw(dx[pick]), ref_dx(pick)) < tol(dt)
    s = sums.sum(0)
    assert torch.allclose(s[0], dx.view(-1, Cx).sum(0, dtype=torch.float64), rtol=1e-6, atol=5e-2)
    # gate bits + pooled-feature gradient
    lin = torch.empty(N * H * H, Cx // 8, dtype=torch.uint8, device="cuda")
    vec = torch.arange(8, device="cuda", dtype=torch.int32)
    for i in range(0, N, 256):  # bit e of byte = (y[m][8*chunk+e] > 0)
        blk = (yprev[i:i + 256].reshape(-1, Cx // 8, 8) > 0).to(torch.int32)
        lin[i * H * H:(i + 256) * H * H] = (blk << vec).sum(-1).to(torch.uint8)
    bits = kn.gate_pack(lin, Cx, dt)  # the kernels' (blocked) gate-byte layout
    del lin
    gapg = _rand_nhwc((N, Cx), dt, 10)
    dx2 = torch.empty_like(resid)
    sums2 = kn.new_stats(Cx)
    kn.conv_dgrad(d, dy, wd, dx2, resid=resid, gapg=gapg, gap_scale=1.0 / (H * H), mask_bits=bits, sums=sums2)
    torch.cuda.synchronize()
    r = (torch.nn.grad.conv2d_input((len(pick), Cx, H, H), w.double(), _nchw(dy[pick])) + _nchw(resid[pick])
         + gapg[pick].double().cpu().view(len(pick), Cx, 1, 1) / (H * H))
    want = torch.where(_nchw(yprev[pick]) > 0, r.float().to(dt).double(), torch.zeros_like(r))
    assert rel(_nchw(dx2[pick]), want) < tol(dt)
    del dx2, resid, gapg, bits
    # two-source: da2 = gate(g . W1 + a2 . W2 + b): g is [M][256] (6.6 GB), a2 / da2 / mask [M][64]
    W1 = rnd((Cx, Cw), dt, g, 1 / 16.0)
    W2 = rnd((Cw, Cw), dt, g, 1 / 8.0)
    bias = torch.randn(Cw, generator=g) * 0.1
    sc, sh = torch.rand(Cw, generator=g) + 0.5, torch.randn(Cw, generator=g) * 0.3
    d3 = kn.conv_desc(dt, N, H, H, Cw, Cx, 1, 1, 1, 0)      # conv3: 64 -> 256, its input gradient
    da = torch.empty_like(dy)
    s3 = kn.new_stats(Cw)
    wcat = torch.cat([W1, W2], 0).to(dt).cuda()
    c2 = _rand_nhwc((N, H, H, Cw), dt, 11)
    assert kn.conv_dgrad2(d3, yprev, wcat, da, dy, bias=bias.cuda(), mask=(c2, sc.cuda(), sh.cuda()), sums=s3)
    torch.cuda.synchronize()
    pick3 = _imgs(N, H * H * Cx * 2)
    G = yprev[pick3].double().cpu().view(-1, Cx)
    A2 = dy[pick3].double().cpu().view(-1, Cw)
    ref3 = G @ W1.double() + A2 @ W2.double() + bias.double()
    pre = c2[pick3].double().cpu().view(-1, Cw) * sc.double() + sh.double()
    want3 = torch.where(pre > 0, ref3.float().to(dt).double(), torch.zeros_like(ref3))
    got3 = da[pick3].double().cpu().view(-1, Cw)
    clear = pre.abs() > 1e-3
    assert rel(torch.where(clear, got3, want3), want3) < tol(dt)


def test_gram_statistics_at_production_row_count(hip_lib, freed):
    """bn3's batch statistics from the Gram matrix of conv3's operand (engine._gram_stats: sum c = W sum(a),
    sum c^2 = diag(W (a^T a) W^T), fp32 Gram accumulated over 12.8 M pixels, variance as E[c^2] - mean^2) against the
    statistics accumulated from the conv output itself, at the row count of config 2's layer 1 (N = 4096, 56x56)"""
    from msf_wsi_amd import kernels as kn
    from msf_wsi_amd.engine import Engine

    dt = torch.bfloat16
    N, H, Cw, Cx = 4096, 56, 64, 256
    a = _rand_nhwc((N, H, H, Cw), dt, 21)
    a.add_(0.5).clamp_(min=0)                                   # relu-like operand with a non-zero mean
    g = torch.Generator().manual_seed(22)
    w = rnd((Cx, Cw, 1, 1), dt, g, 1 / 8.0)
    wd = nhwc(w).to(dt).cuda()
    M = N * H * H
    # explicit: the conv with its statistics epilogue (sums over the bf16-rounded outputs)
    d = kn.conv_desc(dt, N, H, H, Cw, Cx, 1, 1, 1, 0)
    c = torch.empty(N, H, H, Cx, dtype=dt, device="cuda")
    stats = kn.new_stats(Cx)
    kn.conv_fwd(d, a, wd, c, stats=stats)
    bn_a, bn_b = torch.nn.BatchNorm2d(Cx).cuda().train(), torch.nn.BatchNorm2d(Cx).cuda().train()
    eng = Engine()
    st_explicit = eng._bn_finalize(stats, M, bn_a)
    del c
    # folded: Gram matrix + column sums of the operand
    A = torch.zeros(Cw, 1, 1, Cw, device="cuda")
    kn.conv_wgrad(kn.conv_desc(dt, N, H, H, Cw, Cw, 1, 1, 1, 0), a, a, A)
    sa = torch.zeros(Cw, dtype=torch.float64, device="cuda")
    kn.colsum(a, sa)
    st_fold = eng._gram_stats(wd, A, sa, bn_b, M, dt)
    torch.cuda.synchronize()
    # exact statistics of c = W a in fp64 from the exact Gram matrix (a is exact in bf16)
    a64 = torch.zeros(Cw, Cw, dtype=torch.float64, device="cuda")
    for i in range(0, N, 128):
        blk = a[i:i + 128].reshape(-1, Cw).double()
        a64 += blk.t() @ blk
    W64 = wd.view(Cx, Cw).double()
    mean = (W64 @ sa) / M
    var = ((W64 @ a64) * W64).sum(1) / M - mean * mean
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    for name, st in (("explicit", st_explicit), ("Gram", st_fold)):
        em = float((st.mean.double() - mean).abs().max() / mean.abs().max())
        ei = float(((st.invstd.double() - invstd) / invstd).abs().max())
        print(f"   {name} statistics at M = {M}: mean err {em:.1e}, invstd err {ei:.1e}")
        assert em < 1e-4 and ei < 1e-3, (name, em, ei)
    assert torch.allclose(bn_a.running_var, bn_b.running_var, rtol=2e-3)


# ------------------------------------------------------------------------------------------------
# full-size config 2
# ------------------------------------------------------------------------------------------------
def test_config2_full_size_properties(hip_lib, freed):
    """ResNet-50-derived model, bf16, 256 tile pairs of 224x224 on one GPU (BASELINE config 2, the bench workload)"""
    from helpers import install_hub_stub, MODEL_SEED
    from msf_wsi_amd.models import resnet as R
    from msf_wsi_amd.models.backbone import MSFWSI
    from msf_wsi_amd.train import PretrainStep, synthetic_batch

    free, total = torch.cuda.mem_get_info()
    if total < 250 * 2 ** 30:
        pytest.skip("needs the 288 GB of an MI355X")
    torch.cuda.empty_cache()
    install_hub_stub()
    torch.manual_seed(MODEL_SEED)
    with torch.device("cuda"):
        model = MSFWSI(R.resnet50, 4).train()
    ts = PretrainStep(model, lr=1e-3, global_batch=256, dtype=torch.bfloat16, arch="resnet50")
    batch = synthetic_batch(256, 224, 16, seed=0, device="cuda")
    eng = ts.engine

    def forward_only(fold):
        keep = (eng.fold_bn3_fwd, eng.fold_ds_fwd, eng.update_running)
        eng.fold_bn3_fwd = eng.fold_ds_fwd = fold
        eng.update_running = False
        try:
            with torch.no_grad():
                outs, rec = eng.model_forward(model, (batch[0][0], batch[1][0]), (batch[0][1], batch[1][1]), batch[2],
                                              torch.bfloat16, need_backward=False)
        finally:
            eng.fold_bn3_fwd, eng.fold_ds_fwd, eng.update_running = keep
        from helpers import reference_loop_loss

        loss, terms = reference_loop_loss(outs)
        feats = [f.float() for name in ("c0", "t0") for f in rec.enc[name].feats]
        return float(loss), terms.cpu(), feats

    l_fold, t_fold, z_fold = forward_only(True)
    l_expl, t_expl, z_expl = forward_only(False)
    print(f"config 2 forward loss: folded BatchNorm statistics {l_fold:.6f}, explicit {l_expl:.6f}")
    assert math.isfinite(l_fold) and math.isfinite(l_expl)
    # the Gram-matrix route to bn3 / downsample statistics (fp32 Gram over 12.8 M pixels per channel, E[c^2]-mean^2)
    # against statistics accumulated from the conv output itself: same network up to bf16 rounding of different tensors
    print("   per-term |difference|:", [f"{float(v):.1e}" for v in (t_fold - t_expl).abs()])
    assert float((t_fold - t_expl).abs().max()) <= 5e-3, (t_fold, t_expl)
    # pooled encoder features (the heads' BatchNorm1d then amplifies the bf16 noise of these nearly sample-independent
    # features of N(0,1) images by orders of magnitude -- the reference's own fp32 run is 2e-2 from its fp64 run on this
    # model, fixture r50_b8_s64 -- so the embeddings themselves are not comparable between two bf16 evaluation orders)
    fr = [float((a - b).norm() / b.norm()) for a, b in zip(z_fold, z_expl)]
    print("   pooled-feature rel-L2, folded vs explicit:", [f"{v:.1e}" for v in fr])
    # two bf16 evaluation orders of a 50-layer network: rounding-level after layers 1-2; on N(0,1) images the deeper
    # features of two bf16 evaluation orders drift apart (each BatchNorm re-normalises, i.e. amplifies, the accumulated
    # noise), which is why round 5 could only bound them by 0.25 -- a bound that catches nothing.  Round 6: the FULL-SIZE
    # trunk is pinned to the reference instead (tests/test_headline_geometry_gpu.py::
    # test_resnet50_trunk_224_full_size_replicated: 4 096 well-conditioned images, natural dispatch, every gradient tensor
    # against the fp64 reference); what stays here is what only the whole model at 256 tile pairs can show -- the statistics
    # of the shallow stages, the step's bookkeeping and the memory plan
    assert max(fr[0], fr[1], fr[4], fr[5]) < 1e-2
    del z_fold, z_expl
    torch.cuda.empty_cache()
    loss = float(ts.step(batch))
    torch.cuda.synchronize()
    assert math.isfinite(loss) and abs(loss - l_fold) <= 5e-3, (loss, l_fold)
    assert int(ts.found_inf.item()) == 0 and ts.t == 1
    nbt = {k: int(v) for k, v in model.state_dict().items() if k.endswith("num_batches_tracked")}
    assert len(nbt) == 154 and set(nbt.values()) == {2}       # two views per step, every BatchNorm module
    for gi in range(3):
        assert torch.isfinite(ts.flats.w[gi]).all()
    loss2 = float(ts.step(batch))
    assert math.isfinite(loss2)
    del ts, model, batch


S2_CONVS = [
    # N, H, W, C, K, R, stride, pad   (even extents: the parity-class form of the stride-2 3x3 input gradient)
    (2, 28, 28, 128, 128, 3, 2, 1),
    (3, 14, 18, 64, 256, 3, 2, 1),
    (1, 56, 56, 64, 64, 3, 2, 1),      # ResNet-18 first conv of layer2-like: 64 output channels -> 128x64 tile
    (2, 8, 6, 256, 256, 3, 2, 1),
]


@pytest.mark.parametrize("dt", tk.DTYPES)
@pytest.mark.parametrize("geom", S2_CONVS)
@pytest.mark.parametrize("big", [False, True])
def test_stride2_dgrad_by_parity(hip_lib, dt, geom, big):
    """4 launches over the (h % 2, w % 2) classes of the dX pixels (1x1, 1x2, 2x1, 2x2 sub-kernels read in place from the
    3x3 weights) == the single masked launch == torch; plain, with residual + pooled gradient, and gated with sums"""
    hip_lib.msfwsi_set_tuning(0, 1 if big else 1024)
    try:
        tk.test_conv_dgrad(hip_lib, dt, geom)
        tk.test_conv_dgrad_fused_activation_backward(hip_lib, dt, geom)
        # A/B against the single-launch form on the same inputs: identical up to the summation order of the taps
        from msf_wsi_amd import kernels as kn

        N, H, W, Cc, K, R, st, pad = geom
        g = torch.Generator().manual_seed(77)
        d = kn.conv_desc(dt, N, H, W, Cc, K, R, R, st, pad)
        dy = nhwc(rnd((N, K, d.P, d.Q), dt, g)).to(dt).cuda()
        w = nhwc(rnd((K, Cc, R, R), dt, g, 0.05)).to(dt).cuda()
        a, b = torch.empty(N, H, W, Cc, dtype=dt, device="cuda"), torch.empty(N, H, W, Cc, dtype=dt, device="cuda")
        kn.conv_dgrad(d, dy, w, a)
        assert hip_lib.msfwsi_set_tuning(5, 0) == 0
        kn.conv_dgrad(d, dy, w, b)
        torch.cuda.synchronize()
        assert rel(a.float(), b.float()) < (1e-6 if dt == torch.float32 else tol(dt))
    finally:
        assert hip_lib.msfwsi_set_tuning(5, 1) == 0
        assert hip_lib.msfwsi_set_tuning(0, 1024) == 0
