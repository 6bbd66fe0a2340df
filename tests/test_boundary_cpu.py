"""CPU: the drop-in boundary (module tree, state-dict contract, C-ABI symbols).  No compute calls: the product
has no CPU path, and these tests assert that too."""
import inspect
import os
import re

import pytest
import torch
import torch.nn as nn

from helpers import build_product, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol(hip_lib):
    """every function declared in include/msfwsi_hip.h is exported by the built library and bound in _lib.py"""
    from msf_wsi_amd import _lib

    header = open(os.path.join(ROOT, "include", "msfwsi_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(msfwsi_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 25
    assert declared - {"msfwsi_target", "msfwsi_build_id"} == set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(hip_lib, name), name
    assert hip_lib.msfwsi_target().decode() == "gfx950"


def test_build_id_is_the_digest_of_the_sources(hip_lib, tmp_path, monkeypatch):
    """the library carries the sha256 of the sources it was built from (csrc/target.hip, Makefile BUILD_ID): the exported
    msfwsi_build_id(), the marker read from the FILE (_lib.built_id) and the digest of the sources beside it
    (_lib.source_id) agree after a build; staleness is decided by that digest, not by file times"""
    from msf_wsi_amd import _lib

    bid = hip_lib.msfwsi_build_id().decode()
    assert re.fullmatch(r"[0-9a-f]{16}", bid), bid
    assert bid == _lib.built_id() == _lib.source_id()
    assert not _lib._stale()
    # an edited source under an OLDER file time than the binary (what rsync of a git-ignored .so produces): stale
    csrc = tmp_path / "csrc"
    csrc.mkdir()
    (tmp_path / "include").mkdir()
    (tmp_path / "pkg").mkdir()
    for path in _lib.id_files():
        rel = os.path.relpath(path, os.path.dirname(_lib.CSRC_DIR))
        dst = tmp_path / ("include/msfwsi_hip.h" if rel.endswith("msfwsi_hip.h") and "include" in rel else rel)
        dst.write_bytes(open(path, "rb").read())
    fake = tmp_path / "libmsfwsi_hip.so"
    fake.write_bytes(open(_lib.LIB_PATH, "rb").read())
    monkeypatch.setattr(_lib, "CSRC_DIR", str(csrc))
    monkeypatch.setattr(_lib, "_HERE", str(tmp_path / "pkg"))   # id_files: <_HERE>/../include/msfwsi_hip.h
    monkeypatch.setattr(_lib, "LIB_PATH", str(fake))
    assert _lib.source_id() == bid and not _lib._stale()
    src = csrc / "wgrad.hip"
    src.write_bytes(src.read_bytes() + b"// edited\n")
    os.utime(src, (1, 1))                                       # far OLDER than the binary
    assert os.path.getmtime(src) < os.path.getmtime(fake)
    assert _lib.source_id() != bid and _lib._stale()
    fake.write_bytes(b"no marker in here")
    assert _lib.built_id() == "" and _lib._stale()


def test_bench_replays_counters_only_of_this_build(hip_lib, tmp_path, monkeypatch):
    """bench.py's `roofline.traffic` / `mfma_util` are replayed from a committed rocprofv3 counter summary -- only from one
    taken with THIS build of the kernels (the summary's build_id, written by tools/pmc_summary.py); a summary of another
    build is named as `stale_profile`, never replayed beside a fresh time"""
    import json
    import sys

    sys.path.insert(0, ROOT)
    import bench

    mine = hip_lib.msfwsi_build_id().decode()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_entry("wgrad_kernelIxE") == {} and bench.pmc_step() == (None, None, None)
    tab = {"kernels": {"wgrad_kernelIxE": {"hbm_bytes_per_launch": 7.0, "launches_in_pass": 2, "mfma_util": 0.5}},
           "hbm_bytes_per_step": 11.0}
    (prof / "r98_pmc.json").write_text(json.dumps(dict(tab, build_id="0123456789abcdef")))
    assert bench.pmc_entry("wgrad_kernelIxE") == {"stale_profile": "r98_pmc.json"}
    assert bench.pmc_step() == (None, None, "r98_pmc.json")
    (prof / "r97_pmc.json").write_text(json.dumps(tab))        # no build id at all (the summaries of rounds 1-5): stale too
    assert bench.pmc_entry("wgrad_kernelIxE").get("stale_profile") and bench.pmc_step()[0] is None
    (prof / "r96_pmc.json").write_text(json.dumps(dict(tab, build_id=mine)))   # an OLDER file of this very build: replayed
    e = bench.pmc_entry("_ZN12_GLOBAL__N_112wgrad_kernelIxEEvNS")
    assert e["hbm_bytes_per_launch"] == 7.0 and e["source"] == "r96_pmc.json"
    assert bench.pmc_step() == (11.0, "r96_pmc.json", None)


def test_no_product_import_of_oracle():
    """the product must never route through the checker"""
    pkg = os.path.join(ROOT, "msf_wsi_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), os.path.join(dp, f)


def test_model_api_matches_reference_contract():
    from msf_wsi_amd.models import resnet
    from msf_wsi_amd.models.backbone import MSFWSI, make_predictor, make_projector

    sig = inspect.signature(MSFWSI.__init__)
    assert list(sig.parameters) == ["self", "base_encoder", "scale", "dim", "pred_dim", "mask_ratio", "use_checkpoint"]
    assert sig.parameters["dim"].default == 2048 and sig.parameters["pred_dim"].default == 512
    assert list(inspect.signature(MSFWSI.forward).parameters) == ["self", "x1", "x2", "jigsaw_idx"]
    for arch in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        assert callable(resnet.__dict__[arch])
    m = build_product("resnet18")
    assert m.K == 16 and m.n_keep == 8
    assert m.inter_dim.tolist() == [64, 128, 256, 512] and m.ms_inter_dim.tolist() == [576, 1152, 2304, 4608]
    assert isinstance(m.context_encoder.fc, nn.Identity)
    assert [type(x).__name__ for x in make_projector(64, 64)] == ["Linear", "BatchNorm1d", "ReLU", "Linear",
                                                                 "BatchNorm1d", "ReLU", "Linear", "BatchNorm1d"]
    pj = make_projector(64, 64)
    assert pj[7].affine is False and pj[0].bias is None
    pd = make_predictor(64, 16)
    assert pd[0].bias is None and pd[3].bias is not None and pd[3].out_features == 64
    # state-dict contract
    vec, man = load_golden("r18_b8_s64")
    sd = m.state_dict()
    assert len(sd) == 528 and [k for k in sd] == [k for k, _, _ in man["keys"]]
    assert "context_encoder.layer2.0.downsample.1.running_var" in sd and "inter_projector.3.6.weight" in sd
    assert "target_predictor.0.3.bias" in sd and sd["context_encoder.bn1.num_batches_tracked"].dtype == torch.int64
    assert sum(p.numel() for p in m.parameters()) == 123551584
    named = [n for n, _ in m.named_parameters()]
    assert [sum(n.startswith(p) for n in named) for p in ("context_", "target_", "inter_")] == [108, 108, 48]
    # conv weights sit in the kernels' [K][R][S][C] order without changing logical shape / values
    w = m.target_encoder.layer1[0].conv1.weight
    assert tuple(w.shape) == (64, 64, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous()


def test_survives_reference_wrappers():
    """SyncBatchNorm conversion (ssl_train.py:160) keeps keys; repr works (ssl_train.py:172)"""
    m = build_product("resnet18")
    keys = list(m.state_dict())
    m2 = nn.SyncBatchNorm.convert_sync_batchnorm(m)
    assert list(m2.state_dict()) == keys
    assert sum(isinstance(x, nn.SyncBatchNorm) for x in m2.modules()) == 88
    assert "MSFWSI" in repr(m2)


def test_resnet50_generalisation_builds():
    from msf_wsi_amd.models import resnet

    enc = resnet.resnet50(zero_init_residual=True, return_features=True)
    assert enc.fc.in_features == 2048
    keys = list(enc.state_dict())
    assert "layer1.0.conv3.weight" in keys and "layer1.0.downsample.0.weight" in keys and len(keys) == 320
    assert float(enc.layer1[0].bn3.weight.abs().sum()) == 0.0  # zero_init_residual
    assert float(enc.layer1[0].bn2.weight.sum()) == 64.0


def test_unsupported_variants_and_cpu_raise():
    from msf_wsi_amd._lib import MsfwsiHipError
    from msf_wsi_amd.models import resnet

    with pytest.raises(NotImplementedError):
        resnet.ResNet(resnet.Bottleneck, [1, 1, 1, 1], groups=32, width_per_group=4)
    m = build_product("resnet18")
    x = torch.zeros(2, 3, 64, 64)
    t = torch.zeros(32, 3, 64, 64)
    idx = [torch.arange(16).repeat(2, 1)] * 2
    with pytest.raises(MsfwsiHipError):
        m((x, t), (x, t), idx)


def test_pretrained_offline_dir(tmp_path, monkeypatch):
    from msf_wsi_amd.models import resnet

    ref = resnet.resnet18()
    torch.save(ref.state_dict(), tmp_path / "resnet18-f37072fd.pth")
    monkeypatch.setenv("MSFWSI_PRETRAINED_DIR", str(tmp_path))
    got = resnet.resnet18(pretrained=True)
    for (k, a), (_, b) in zip(ref.state_dict().items(), got.state_dict().items()):
        assert torch.equal(a, b), k


def test_panel_hand_counted_waits_audit():
    """static audit of the panel kernels' hand-counted `s_waitcnt vmcnt(N)` (tools/check_hand_waits.py): in hipcc's
    assembly of csrc/panel.hip every asm load is followed by at least N younger vector-memory operations before the wait that
    covers its first reader, and hipcc has put no wait, load or scratch access of its own into those loops"""
    import shutil
    import subprocess
    import sys as _sys

    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not (os.path.isfile(hipcc) and os.access(hipcc, os.X_OK)):
        pytest.skip("no hipcc on this machine ($HIPCC, PATH, /opt/rocm/bin): the audit reads hipcc's assembly")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([_sys.executable, os.path.join(root, "tools", "check_hand_waits.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 findings" in r.stdout


def test_gate_byte_layout_matches_the_header():
    """the ReLU-gate bytes' layout (include/msfwsi_hip.h at msfwsi_conv_fwd_post, csrc/common.h gate_off) restated from the
    header's words, against the host-side index kernels.gate_pack / gate_unpack use: linear where a row does not hold whole
    dwords of gate bytes, else 512-byte groups of 128 rows x 4 chunks; pack -> unpack is the identity, the buffer is padded
    to whole 128-row groups, and no two (row, chunk) pairs share a byte"""
    import torch

    from msf_wsi_amd import kernels as kn

    for rows, Cn, dt in ((5, 24, torch.bfloat16), (130, 64, torch.bfloat16), (257, 256, torch.float16), (128, 32, torch.float32),
                         (3, 2048, torch.bfloat16), (200, 96, torch.bfloat16)):
        vec = 4 if dt == torch.float32 else 8
        cpr = Cn // vec

        def header_offset(m, c):
            if cpr % 4:
                return m * cpr + c
            return ((m // 128) * (cpr // 4) + c // 4) * 512 + (m % 128) * 4 + c % 4

        idx = kn._gate_index(rows, cpr, "cpu")
        want = torch.tensor([[header_offset(m, c) for c in range(cpr)] for m in range(rows)])
        assert torch.equal(idx, want)
        n = kn.gate_numel(rows, Cn, dt)
        assert n == (rows * cpr if cpr % 4 else (rows + 127) // 128 * 128 * cpr)
        assert int(idx.max()) < n and idx.unique().numel() == rows * cpr
        g = torch.Generator().manual_seed(rows)
        lin = torch.randint(0, 256, (rows, cpr), generator=g, dtype=torch.uint8)
        buf = kn.gate_pack(lin, Cn, dt)
        assert buf.numel() == n and torch.equal(kn.gate_unpack(buf, rows, Cn, dt), lin)


def test_image_kernels_are_chosen_only_when_their_bands_fill_the_chip(monkeypatch):
    """Engine._img3_fills: one workgroup per band of an image -- 14 x 14: one band, one workgroup per CU; 28 x 28: four bands,
    two per CU; 56 x 56: fourteen bands, three per CU; the strided gradient is keyed on the gradient's height"""
    import types

    import torch

    from msf_wsi_amd import engine as eng_mod

    e = eng_mod.Engine.__new__(eng_mod.Engine)
    e.img3x3_min_fill = 1.0
    e._ncu = {"dev": 256}
    d = lambda N, H, stride=1: types.SimpleNamespace(N=N, H=H, P=H // stride, stride=stride)
    assert not e._img3_fills(d(255, 14), "dev") and e._img3_fills(d(256, 14), "dev")
    assert not e._img3_fills(d(127, 28), "dev") and e._img3_fills(d(128, 28), "dev")
    assert not e._img3_fills(d(54, 56), "dev") and e._img3_fills(d(55, 56), "dev")
    assert not e._img3_fills(d(127, 56, 2), "dev") and e._img3_fills(d(128, 56, 2), "dev")   # gradient 28 x 28
    assert not e._img3_fills(d(255, 28, 2), "dev") and e._img3_fills(d(256, 28, 2), "dev")   # gradient 14 x 14
    assert not e._img3_fills(d(4096, 7), "dev")
    e.img3x3_min_fill = 0.0
    assert e._img3_fills(d(1, 14), "dev")
