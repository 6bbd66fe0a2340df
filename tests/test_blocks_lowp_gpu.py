"""Per-BLOCK 16-bit parity on a real MI355X: percent-level sensitivity for the arithmetic that only
16-bit runs take.

The whole-model 16-bit gates (test_lowp_parity_gpu.py) hold the product to the distance of the REFERENCE UNDER AUTOCAST
from its fp64 run -- 20-30 % per gradient tensor at these batch sizes, so a 3-5 % error in a 2-byte-only kernel passes
them.  Here every residual block, the stem and every kind of head of the product is checked ON ITS OWN: the block runs in
16-bit storage on the product's own 16-bit block input (taken from a whole encoder pass on the well-conditioned trunk
case), forward and backward, against an fp64 restatement of THAT BLOCK (resnet.py:66-82 BasicBlock, :120-140 Bottleneck,
:234-237 stem, backbone.py:12-31 heads; plain torch.nn.functional, fp64, CPU) on the same input and the same incoming
gradient.  One block carries 3-6 roundings to the storage type, not the 50-layer chain's amplification, so the allowance
is a few units in the last place of the storage type -- a bound a 3 % error in any kernel of the block exceeds
(test_block_fault_injection: a BatchNorm scale off by 5 % / 2 % turns it red).

Bounds (rel-L2 per tensor) = about twice the worst value measured on the MI355X (gpurun_out/r4_sel4.log):
  bf16: outputs <= 1.2e-2 [blocks 4.5e-3, stem 2.7e-3, heads 7.4e-3], gradients <= 1.6e-2 [blocks 8.1e-3 (median
        4.7e-3), stem 3.2e-3, heads 7.0e-3; BatchNorm1d over the fuser's 16 rows: 2.0e-2 -> 3.2e-2]
  fp16: outputs <= 1.5e-3 [5.8e-4 / 3.4e-4 / 9.3e-4], gradients <= 2.0e-3 [blocks median 5.8e-4, heads 1.0e-3]
The BACKWARD comparisons run the fp64 block with the product's own ReLU gates / max-pool positions (see _block_oracle:
with free gates a 16-bit block's gradients sit 5e-2 (bf16) / 1.6e-2 (fp16) from fp64 -- gate flips, sqrt(eps))
The stationary kernels (weights-stationary 3x3, output-stationary weight gradient, the stem pair incl. stem_wgrad_bnbwd)
are forced onto these small shapes by lifting their size gates (conftest-style tuning keys), so the 2-byte-only kernels
are the ones measured."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import MODEL_SEED, load_golden, rel

pytestmark = pytest.mark.gpu

OUT_TOL = {torch.bfloat16: 1.2e-2, torch.float16: 1.5e-3}
GRAD_TOL = {torch.bfloat16: 1.6e-2, torch.float16: 2.0e-3}
SMALL_BATCH_GRAD_TOL = {torch.bfloat16: 3.2e-2, torch.float16: 2.0e-3}  # BatchNorm1d over 16 rows (the fuser heads' batch)
DTYPES = [torch.bfloat16, torch.float16]
IDS = ["bf16", "fp16"]


@pytest.fixture
def stationary_forced(hip_lib):
    """the shape-specialised persistent kernels take the small test shapes too (size gates lifted)"""
    from helpers import tuned

    with tuned(hip_lib, {9: 1, 10: 1, 12: 1, 11: 0, 13: 0}):
        yield


def _encoder(arch, gain=0.1):
    from msf_wsi_amd.models import resnet

    torch.manual_seed(MODEL_SEED)
    enc = resnet.__dict__[arch](zero_init_residual=False, return_features=True)
    enc.fc = torch.nn.Identity()
    last = ".bn2.weight" if arch == "resnet18" else ".bn3.weight"
    with torch.no_grad():
        for k, v in enc.state_dict().items():
            if k.startswith("layer") and k.endswith(last):
                v.mul_(gain)
    return enc


def _nchw64(t):
    return t.detach().double().cpu().permute(0, 3, 1, 2).contiguous()


def _bn64(x, bn):
    return F.batch_norm(x, None, None, bn["w"], bn["b"], training=True, eps=bn["eps"])


def _leaf(p):
    return p.detach().double().cpu().clone().requires_grad_(True)


class _GatedReLU(torch.autograd.Function):
    """relu whose gate (the set of active elements) is GIVEN: forward x * gate, backward grad * gate"""

    @staticmethod
    def forward(ctx, x, gate):
        ctx.save_for_backward(gate)
        return x * gate

    @staticmethod
    def backward(ctx, g):
        (gate,) = ctx.saved_tensors
        return g * gate, None


def _gate_of(c, st):
    """the ReLU gate the product's kernels derive from a kept raw conv output: fma(c, scale, shift) > 0 in fp32.  Evaluated
    here in fp64, where the product of a 16-bit value and an fp32 value is exact: the sign of the exact sum is the sign of
    the fused fp32 result (a separate fp32 multiply and add can fall on the other side for a handful of elements)"""
    pre = c.double() * st.scale.double() + st.shift.double()
    return (pre > 0).double().cpu().permute(0, 3, 1, 2).contiguous()


def _block_oracle(blk, rec, x64, dy64):
    """fp64 forward + backward of one residual block (resnet.py:66-82 / :120-140) with batch statistics of ITS input.

    Returns (true forward output, input gradient, parameter gradients).  The BACKWARD runs with the product's own ReLU
    gates (from the raw conv outputs and BatchNorm maps the product keeps, and from its block output): a 16-bit run
    rounds every pre-activation, so ~2^-8 (bf16) / 2^-11 (fp16) of the gates sit on the other side than in fp64, each
    moving its element's gradient by 100 % -- a sqrt(eps) effect (measured with free gates: 4.8e-2 / 1.6e-2 median per
    tensor) that would hide any percent-level arithmetic error.  Which way a borderline gate falls is part of the
    forward state, checked by the forward comparison; given the gates, the backward is pure arithmetic."""
    convs, bns = [], []
    for conv, bn in blk.main_branch():
        convs.append((_leaf(conv.weight), conv.stride, conv.padding))
        bns.append({"w": _leaf(bn.weight), "b": _leaf(bn.bias), "eps": bn.eps})
    ds = None
    if blk.downsample is not None:
        dc, db = blk.downsample[0], blk.downsample[1]
        ds = ((_leaf(dc.weight), dc.stride), {"w": _leaf(db.weight), "b": _leaf(db.bias), "eps": db.eps})
    x = x64.clone().requires_grad_(True)
    out = x
    for i, ((w, st, pad), bn) in enumerate(zip(convs, bns)):
        out = _bn64(F.conv2d(out, w, None, stride=st, padding=pad), bn)
        if i + 1 < len(convs):
            out = _GatedReLU.apply(out, _gate_of(rec.units[i].c, rec.units[i].st))
    ident = x if ds is None else _bn64(F.conv2d(x, ds[0][0], None, stride=ds[0][1]), ds[1])
    pre = out + ident
    y = _GatedReLU.apply(pre, (_nchw64(rec.y_out) > 0).double())
    y.backward(dy64)
    grads = {}
    for i, ((w, _, _), bn) in enumerate(zip(convs, bns)):
        grads[f"conv{i + 1}.weight"], grads[f"bn{i + 1}.weight"], grads[f"bn{i + 1}.bias"] = w.grad, bn["w"].grad, bn["b"].grad
    if ds is not None:
        grads["downsample.0.weight"], grads["downsample.1.weight"], grads["downsample.1.bias"] = (
            ds[0][0].grad, ds[1]["w"].grad, ds[1]["b"].grad)
    return F.relu(pre).detach(), x.grad, grads


def _check(name, got, want, tol, worst):
    r = rel(got, want)
    worst[name] = r
    return r <= tol


def run_blocks(arch, dtype, mutate=None, batch=None, size=None, only=None):
    """every residual block of `arch` (only: those whose name passes) on the product's own 16-bit block inputs; returns
    {tensor: rel-L2}, failures"""
    from msf_wsi_amd.engine import Engine, GradStore
    from oracle import msfwsi_oracle as orc

    vec, man = load_golden("r50enc_b16_s64_div")
    x = orc.diverse_images(batch or man["B"], size or man["size"], man["data_seed"])
    enc = _encoder(arch).cuda().train()
    eng = Engine()
    eng.update_running = False
    if mutate is not None:
        mutate()
    ps = eng.encoder_forward(enc, x.cuda(), dtype)
    torch.cuda.synchronize()
    blocks = [b for stage in enc.stages() for b in stage]
    names = [f"layer{si + 1}.{bi}" for si, stage in enumerate(enc.stages()) for bi in range(len(stage))]
    assert len(blocks) == len(ps.blocks)
    g = torch.Generator().manual_seed(11)
    worst, bad = {}, []
    for name, blk, rec in zip(names, blocks, ps.blocks):
        if only is not None and not only(name):
            continue
        dy = (torch.randn(rec.y_out.shape, generator=g) * 0.1).to(dtype).cuda()
        y_ref, dx_ref, g_ref = _block_oracle(blk, rec, _nchw64(rec.y_in), _nchw64(dy))
        if not _check(f"{name}: out", _nchw64(rec.y_out), y_ref, OUT_TOL[dtype], worst):
            bad.append(f"{name}: out")
        grads = GradStore()
        dx, _ = eng._block_bwd(rec, dy.clone(), None, grads, dtype)
        torch.cuda.synchronize()
        if not _check(f"{name}: dx", _nchw64(dx), dx_ref, GRAD_TOL[dtype], worst):
            bad.append(f"{name}: dx")
        params = dict(blk.named_parameters())
        for k, gr in g_ref.items():
            if not _check(f"{name}: d {k}", grads.logical(params[k]), gr, GRAD_TOL[dtype], worst):
                bad.append(f"{name}: d {k}")
    return worst, bad


def _report(tag, worst):
    outs = [v for k, v in worst.items() if k.endswith(": out")]
    grs = [v for k, v in worst.items() if not k.endswith(": out")]
    kmax = max(worst, key=worst.get)
    print(f"[{tag}] {len(worst)} tensors: outputs median {np.median(outs):.2e} max {max(outs):.2e}; gradients median "
          f"{np.median(grs):.2e} max {max(grs):.2e}; worst {kmax} {worst[kmax]:.2e}")


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("arch", ["resnet50", "resnet18"])
def test_every_block_within_a_few_ulp(hip_lib, stationary_forced, arch, dtype):
    """16 Bottlenecks (folded tails, two-source launches, strided / stride-1 downsample branches, the stationary 64-channel
    kernels) / 8 BasicBlocks: output, input gradient and every parameter gradient of each block against the fp64 block"""
    worst, bad = run_blocks(arch, dtype)
    _report(f"blocks {arch} {dtype}", worst)
    assert not bad, [(k, f"{worst[k]:.2e}") for k in bad]


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("arch", ["resnet50", "resnet18"])
def test_deep_blocks_on_the_image_stationary_kernels(hip_lib, monkeypatch, arch, dtype):
    """224 x 224 tiles put layer1 at 56x56x64, layer2 at 28x28x128 and layer3 at 14x14x256: conv2 of their stride-1 blocks
    runs on csrc/img3x3.hip (layer1: the gradient only) -- bn1 + ReLU inside the forward staging, bn2's backward inside the gradient staging (in place at 14x14,
    into a second buffer at 28x28), a1 for the weight gradient written by the gradient's gate.  Same fp64 block oracle and
    bounds as the 64 x 64 run; the launches are counted so that a silent fallback to the gather kernel fails the test."""
    from msf_wsi_amd import kernels as kn

    calls = {"fwd": 0, "dgrad": 0, "fused_bn": 0, "s2": 0}
    fwd, dgrad, s2dgrad = kn.img3x3_fwd, kn.img3x3_dgrad, kn.img3x3_s2_dgrad

    def count_s2(d, *a, **k):
        calls["s2"] += 1
        assert k.get("bnbwd") is not None and k.get("act_out") is not None
        return s2dgrad(d, *a, **k)

    def count_fwd(*a, **k):
        calls["fwd"] += 1
        return fwd(*a, **k)

    def count_dgrad(d, *a, **k):
        calls["dgrad"] += 1
        calls["fused_bn"] += k.get("bnbwd") is not None
        # layer1 (64 channels): only with the BatchNorm backward folded in, and without a1 (the output-stationary weight
        # gradient normalises c1 itself)
        assert (k.get("act_out") is None and k.get("bnbwd") is not None) if d.C == 64 else k.get("act_out") is not None
        return dgrad(d, *a, **k)

    monkeypatch.setenv("MSFWSI_ENGINE", "img3x3_min_fill=0")  # three images: far below the fill the engine asks for by default
    monkeypatch.setattr(kn, "img3x3_fwd", count_fwd)
    monkeypatch.setattr(kn, "img3x3_dgrad", count_dgrad)
    monkeypatch.setattr(kn, "img3x3_s2_dgrad", count_s2)
    stages = ("layer1.", "layer2.", "layer3.") if arch == "resnet50" else ("layer2.", "layer3.")
    # (resnet50: the strided first blocks of layer2 / layer3 too -- their conv2's input gradient is the one-launch strided kernel)
    deep = lambda name: name.startswith(stages) and (not name.endswith(".0") or (arch == "resnet50" and name[:6] in ("layer2", "layer3")))
    worst, bad = run_blocks(arch, dtype, batch=3, size=224, only=deep)
    _report(f"deep blocks {arch} {dtype}", worst)
    assert not bad, [(k, f"{worst[k]:.2e}") for k in bad]
    nblk = {"resnet50": 2 + 3 + 5, "resnet18": 1 + 1}[arch]
    assert len([k for k in worst if k.endswith(": out")]) == nblk + (2 if arch == "resnet50" else 0)
    if arch == "resnet50":   # conv2 of every Bottleneck of the three stages; bn2 folded into each gradient launch
        assert calls == {"fwd": 3 + 5, "dgrad": nblk, "fused_bn": nblk, "s2": 2}  # (layer1's forward stays weights-stationary)
    else:                    # BasicBlocks: both 3x3 convs forward (the strided blocks' conv2 too), conv2's gradient
        assert calls["fwd"] == 2 * nblk + 2 and calls["dgrad"] == nblk and calls["fused_bn"] == 0


def test_image_kernel_backward_in_image_chunks(hip_lib, monkeypatch):
    """Engine._img3_bwd_chunks: where the by-product activation a1 would be large, the gradient launch + weight gradient pair
    runs over chunks of images with one chunk-sized a1 / dc (memory: two views' a1 beside their da filled the card).  Forced
    here by a 1-byte threshold on four images: same fp64 block oracle, same bounds, and the launches counted"""
    from msf_wsi_amd import kernels as kn

    calls = {"s1": 0, "s2": 0}
    dgrad, s2dgrad = kn.img3x3_dgrad, kn.img3x3_s2_dgrad

    def count(which, fn):
        def wrapped(d, *a, **k):
            calls[which] += 1
            assert d.N == 1 and k.get("act_out") is not None and k["act_out"].shape[0] == 1
            return fn(d, *a, **k)
        return wrapped

    monkeypatch.setenv("MSFWSI_ENGINE", "img3x3_min_fill=0,img3x3_chunk_bytes=1")
    monkeypatch.setattr(kn, "img3x3_dgrad", count("s1", dgrad))
    monkeypatch.setattr(kn, "img3x3_s2_dgrad", count("s2", s2dgrad))
    only = lambda name: name in ("layer2.0", "layer2.1", "layer3.0")
    worst, bad = run_blocks("resnet50", torch.bfloat16, batch=4, size=224, only=only)
    _report("chunked image-kernel backward", worst)
    assert not bad, [(k, f"{worst[k]:.2e}") for k in bad]
    assert calls == {"s1": 4, "s2": 8}  # four one-image chunks per block


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_stem_within_a_few_ulp(hip_lib, stationary_forced, dtype):
    """conv1 -> bn1 -> relu -> maxpool (resnet.py:234-237) through stem_ws_kernel, stem_pool_fwd / _bwd and
    stem_wgrad_bnbwd (bn1's backward inside the weight-gradient staging): pooled output and d conv1.weight, d bn1"""
    from msf_wsi_amd.engine import Engine, GradStore
    from oracle import msfwsi_oracle as orc

    vec, man = load_golden("r50enc_b16_s64_div")
    x = orc.diverse_images(man["B"], man["size"], man["data_seed"])
    enc = _encoder("resnet50").cuda().train()
    eng = Engine()
    eng.update_running = False
    ps = eng.encoder_forward(enc, x.cuda(), dtype)
    assert ps.stem.s2d, "the space-to-depth stationary stem kernels must be the ones under test"
    g = torch.Generator().manual_seed(12)
    dy = (torch.randn(ps.pooled.shape, generator=g) * 0.1).to(dtype).cuda()
    # fp64 stem on the input as the product quantised it (conv1 reads 16-bit pixels)
    w = _leaf(enc.conv1.weight)
    bn = {"w": _leaf(enc.bn1.weight), "b": _leaf(enc.bn1.bias), "eps": enc.bn1.eps}
    xq = x.to(dtype).double()
    pre = _bn64(F.conv2d(xq, w, None, stride=2, padding=3), bn)
    y = F.max_pool2d(F.relu(pre), 3, 2, 1).detach()  # the forward as fp64 computes it
    # backward with the product's forward decisions (ReLU gate of c0, window position of each maximum): see _block_oracle
    a = _GatedReLU.apply(pre, _gate_of(ps.stem.c, ps.stem.st))
    code = ps.amax.long().cpu().permute(0, 3, 1, 2)                       # r * 3 + s per pooled element
    P, Q = code.shape[2:]
    H0, W0 = a.shape[2:]
    hh = (2 * torch.arange(P).view(1, 1, P, 1) - 1 + code // 3).clamp(0, H0 - 1)
    ww = (2 * torch.arange(Q).view(1, 1, 1, Q) - 1 + code % 3).clamp(0, W0 - 1)
    picked = torch.gather(a.flatten(2), 2, (hh * W0 + ww).flatten(2)).view(code.shape)
    assert rel(picked.detach(), y) < OUT_TOL[dtype]                       # the product's choices are maxima up to rounding
    picked.backward(_nchw64(dy))
    grads = GradStore()
    eng._stem_bwd(ps, dy.clone(), grads, dtype)
    torch.cuda.synchronize()
    worst = {}
    ok = [_check("stem: out", _nchw64(ps.pooled), y, OUT_TOL[dtype], worst),
          _check("stem: d conv1.weight", grads.logical(enc.conv1.weight), w.grad, GRAD_TOL[dtype], worst),
          _check("stem: d bn1.weight", grads.logical(enc.bn1.weight), bn["w"].grad, GRAD_TOL[dtype], worst),
          _check("stem: d bn1.bias", grads.logical(enc.bn1.bias), bn["b"].grad, GRAD_TOL[dtype], worst)]
    print(f"[stem {dtype}] " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    assert all(ok), worst


def _head_oracle(seq, rec, x64, dy64):
    """fp64 restatement of a projector / predictor Sequential (backbone.py:12-31) with batch statistics; the backward runs
    with the product's ReLU gates (see _block_oracle).  Returns (true forward output, dx, parameter gradients)."""
    x = x64.clone().requires_grad_(True)
    out, true_out, leaves, li = x, x64, {}, -1
    for i, m in enumerate(seq):
        if isinstance(m, torch.nn.Linear):
            li += 1
            w = leaves[f"{i}.weight"] = _leaf(m.weight)
            b = None
            if m.bias is not None:
                b = leaves[f"{i}.bias"] = _leaf(m.bias)
            out = F.linear(out, w, b)
            true_out = F.linear(true_out, w.detach(), None if b is None else b.detach())
        elif isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            wt = bs = None
            if m.affine:
                wt, bs = _leaf(m.weight), _leaf(m.bias)
                leaves[f"{i}.weight"], leaves[f"{i}.bias"] = wt, bs
            out = F.batch_norm(out, None, None, wt, bs, training=True, eps=m.eps)
            true_out = F.batch_norm(true_out, None, None, None if wt is None else wt.detach(),
                                    None if bs is None else bs.detach(), training=True, eps=m.eps)
        else:
            u = rec.units[li]
            gate = ((u.c.view(u.c.shape[0], -1).double() * u.st.scale.double() + u.st.shift.double()) > 0).double().cpu()
            out = _GatedReLU.apply(out, gate)
            true_out = F.relu(true_out)
    out.backward(dy64)
    return true_out.detach(), x.grad, {k: v.grad for k, v in leaves.items()}


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("width,rows", [(256, 256), (2048, 256), (4608, 16)], ids=["d256", "d2048", "fuser4608"])
def test_heads_within_a_few_ulp(hip_lib, width, rows, dtype):
    """projector (Linear-BN-ReLU x2, Linear-BN(affine=False)) and predictor (Linear-BN-ReLU-Linear+bias) chains of the
    product (engine.chain_forward / chain_backward) in 16-bit storage on 16-bit rows against the fp64 chain"""
    from msf_wsi_amd.engine import Engine, GradStore
    from msf_wsi_amd.models.backbone import make_predictor, make_projector

    torch.manual_seed(MODEL_SEED)
    g = torch.Generator().manual_seed(13)
    eng = Engine()
    eng.update_running = False
    worst, bad = {}, []
    for kind, make in (("projector", make_projector), ("predictor", make_predictor)):
        seq = (make(width, width) if kind == "projector" else make(width, width // 4)).cuda().train()  # backbone.py:70-100
        # pooled features are positive with a per-row level and pattern (what a ReLU network's GAP output looks like)
        x = (torch.rand(rows, width, generator=g) * (0.5 + torch.rand(rows, 1, generator=g))).to(dtype).cuda()
        dy = (torch.randn(rows, width, generator=g) * 0.1).to(dtype).cuda()
        rec = eng.chain_forward(seq, x, dtype)
        y_ref, dx_ref, g_ref = _head_oracle(seq, rec, x.double().cpu(), dy.double().cpu())
        grads = GradStore()
        dx = eng.chain_backward(rec, dy.clone(), grads, dtype)
        torch.cuda.synchronize()
        params = dict(seq.named_parameters())
        gtol = (GRAD_TOL if rows >= 64 else SMALL_BATCH_GRAD_TOL)[dtype]
        checks = [(f"{kind}{width}: out", rec.out, y_ref, OUT_TOL[dtype]), (f"{kind}{width}: dx", dx, dx_ref, gtol)]
        checks += [(f"{kind}{width}: d {k}", grads.logical(params[k]), v, gtol) for k, v in g_ref.items()]
        for name, got, want, tol in checks:
            if not _check(name, got, want, tol, worst):
                bad.append(name)
    print(f"[heads {width} x {rows} {dtype}] " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    assert not bad, [(k, f"{worst[k]:.2e}") for k in bad]


@pytest.mark.parametrize("factor", [1.05, 1.02], ids=["x1.05", "x1.02"])
def test_block_fault_injection(hip_lib, stationary_forced, monkeypatch, factor):
    """falsifiability at the percent level (the whole-model bf16 gate needed x1.25): the BatchNorm
    scale of ONE transient activation (a2 = relu(bn2(c2)) of a layer2 block, applied by bn_act_sum) multiplied by 1.05 /
    1.02 must put that block's output beyond the bf16 bound"""
    from msf_wsi_amd import kernels as kn

    real = kn.bn_act_sum
    hits = []

    def wrong(c, scale, shift, out, sums):
        if c.shape[-1] == 128 and not hits:
            hits.append(1)
            scale = scale * factor
        return real(c, scale, shift, out, sums)

    worst, bad = run_blocks("resnet50", torch.bfloat16, mutate=lambda: monkeypatch.setattr(kn, "bn_act_sum", wrong))
    assert hits, "the fault was never injected"
    print(f"[fault x{factor}] tensors beyond their bound: {[(k, round(worst[k], 4)) for k in bad]}")
    # (the block's OUTPUT cannot show it: bn3 behind the linear conv3 divides a uniform scale of conv3's operand out
    #  again -- it is the gradients flowing back through the mis-scaled activation that are off by the factor)
    assert any(k.startswith("layer2.0:") for k in bad), (factor, {k: v for k, v in worst.items() if k.startswith("layer2.0")})


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_block_gates_agree_with_free_fp64_gates(hip_lib, stationary_forced, dtype):
    """The backward comparisons above run the fp64 block WITH THE PRODUCT'S ReLU gates; this test closes the other half:
    with FREE gates (plain fp64 ReLU on the same 16-bit block input) the fraction of gates -- inner activations and block
    output -- that fall on the other side in the product is at most 2^-7 (bf16) / 2^-10 (fp16): a kernel that derived or
    stored its gates wrongly (msfwsi_conv_fwd_post's gate_out, the panel kernels' blocked gate bytes, the mask the input
    gradients re-derive from a raw conv output) would differ in percent of them.  And the gate BYTES the forward wrote equal
    the sign of the stored block output bit for bit."""
    from msf_wsi_amd import kernels as kn
    from msf_wsi_amd.engine import Engine
    from oracle import msfwsi_oracle as orc

    vec, man = load_golden("r50enc_b16_s64_div")
    x = orc.diverse_images(man["B"], man["size"], man["data_seed"])
    enc = _encoder("resnet50").cuda().train()
    eng = Engine()
    eng.update_running = False
    ps = eng.encoder_forward(enc, x.cuda(), dtype)
    torch.cuda.synchronize()
    blocks = [b for stage in enc.stages() for b in stage]
    bound = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    worst, with_bits = 0.0, 0
    for blk, rec in zip(blocks, ps.blocks):
        x64 = _nchw64(rec.y_in)
        out = x64
        main = blk.main_branch()
        for i, (conv, bn) in enumerate(main):
            w = conv.weight.detach().double().cpu()
            out = F.batch_norm(F.conv2d(out, w, None, stride=conv.stride, padding=conv.padding), None, None,
                               bn.weight.detach().double().cpu(), bn.bias.detach().double().cpu(), training=True, eps=bn.eps)
            if i + 1 < len(main):
                free = out > 0
                mine = _gate_of(rec.units[i].c, rec.units[i].st) > 0
                frac = (free != mine).double().mean().item()
                worst = max(worst, frac)
                assert frac <= bound, (f"gate of unit {i}", frac, bound)
                out = F.relu(out)
        ident = x64
        if blk.downsample is not None:
            dc, db = blk.downsample[0], blk.downsample[1]
            ident = F.batch_norm(F.conv2d(x64, dc.weight.detach().double().cpu(), None, stride=dc.stride), None, None,
                                 db.weight.detach().double().cpu(), db.bias.detach().double().cpu(), training=True, eps=db.eps)
        free_out = (out + ident) > 0
        mine_out = _nchw64(rec.y_out) > 0
        frac = (free_out != mine_out).double().mean().item()
        worst = max(worst, frac)
        assert frac <= bound, ("block output gate", frac, bound)
        if rec.gate_bits is not None:  # the bytes the fused tails wrote: (stored y > 0), bit for bit
            with_bits += 1
            M, Cn = rec.y_out.numel() // rec.y_out.shape[-1], rec.y_out.shape[-1]
            b = kn.gate_unpack(rec.gate_bits, M, Cn, dtype).to(torch.int32)
            got = ((b.unsqueeze(-1) >> torch.arange(8, device=b.device)) & 1).bool().view(M, Cn)
            assert torch.equal(got, rec.y_out.view(M, Cn) > 0)
    assert with_bits >= 12, with_bits  # every Bottleneck tail on the fused path carries gate bytes
    print(f"[free gates {dtype}] worst fraction of differing gates {worst:.2e} (bound {bound:.2e})")
