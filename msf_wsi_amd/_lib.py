"""ctypes binding of libmsfwsi_hip.so (C ABI declared in include/msfwsi_hip.h).

The product path has no CPU or eager-torch fallback: if the shared library is missing or a symbol is
absent, loading raises and every engine entry point fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSFWSI_LIB") or os.path.join(_HERE, "libmsfwsi_hip.so")  # override: A/B of two builds
CSRC_DIR = os.path.join(_HERE, "csrc")

DT_F32 = 0
DT_BF16 = 1
DT_F16 = 2


class ConvDesc(C.Structure):
    """mirror of `msfwsi_conv_desc`"""

    _fields_ = [(n, C.c_int) for n in ("dtype", "N", "H", "W", "C", "P", "Q", "K", "R", "S", "stride", "pad")]


class MsfwsiHipError(RuntimeError):
    pass


_vp, _i, _l, _f, _d = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double
_desc = C.POINTER(ConvDesc)

# name -> argtypes; must list every symbol declared in include/msfwsi_hip.h
SIGNATURES = {
    "msfwsi_conv_fwd": [_desc, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "msfwsi_conv_dgrad": [_desc, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "msfwsi_conv_wgrad": [_desc, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "msfwsi_conv_wgrad_act": [_desc, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "msfwsi_conv_wgrad_store": [_desc, _vp, _vp, _vp, _vp],
    "msfwsi_gram": [_desc, _vp, _vp, _vp],
    "msfwsi_stem_wgrad_bnbwd": [_desc, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "msfwsi_bn_finalize": [_vp, _i, _i, _d, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "msfwsi_bn_eval_coeffs": [_vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp],
    "msfwsi_shard_sum": [_vp, _i, _i, _vp, _vp],
    "msfwsi_bn_act": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _l, _i, _vp],
    "msfwsi_block_end_bwd": [_i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _l, _i, _i, _vp],
    "msfwsi_act_bwd_reduce": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _l, _i, _vp],
    "msfwsi_bn_bwd_finalize": [_vp, _i, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "msfwsi_bn_bwd_apply": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _vp],
    "msfwsi_nchw_to_nhwc": [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "msfwsi_stem_pool_fwd": [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "msfwsi_stem_pool_bwd": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "msfwsi_upcat_fwd": [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "msfwsi_upcat_bwd": [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "msfwsi_crop": [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "msfwsi_nhwc_to_nchw": [_i, _vp, _vp, _i, _i, _l, _i, _vp],
    "msfwsi_dice_loss": [_i, _vp, _vp, _l, _i, _i, C.c_uint, _d, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp],
    "msfwsi_gap_fwd": [_i, _vp, _vp, _i, _i, _i, _vp],
    "msfwsi_gap_fwd_stride2": [_i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "msfwsi_bn_act_sum": [_i, _vp, _vp, _vp, _vp, _vp, _i, _l, _i, _vp],
    "msfwsi_fold_matvec": [_vp, _vp, _vp, _i, _i, _vp],
    "msfwsi_stem_conv_fwd": [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "msfwsi_nchw_to_s2d": [_i, _vp, _vp, _i, _i, _i, _vp],
    "msfwsi_stem_s2d_weights": [_i, _vp, _vp, _i, _vp],
    "msfwsi_stem_s2d_wfold": [_vp, _vp, _i, _vp],
    "msfwsi_conv_fwd_post2": [_desc, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "msfwsi_conv_dgrad2": [_desc, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "msfwsi_conv_dgrad2_pro": [_desc, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "msfwsi_conv_fwd_post": [_desc, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "msfwsi_row_scale_cat": [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "msfwsi_fold_dots": [_vp, _vp, _vp, _i, _i, _vp],
    "msfwsi_fold_weights": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "msfwsi_colsum": [_i, _vp, _vp, _i, _l, _i, _vp],
    "msfwsi_colstats": [_i, _vp, _vp, _i, _l, _i, _vp],
    "msfwsi_add_f64": [_vp, _vp, _i, _vp],
    "msfwsi_add_f64_to_f32": [_vp, _vp, _i, _f, _vp],
    "msfwsi_rows_permute": [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "msfwsi_pixel_stride": [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "msfwsi_copy2d": [_i, _vp, _l, _vp, _l, _l, _i, _i, _vp],
    "msfwsi_cosine_loss": [_i, _vp, _vp, _l, _i, _f, _vp, _f, _vp, _vp, _vp],
    "msfwsi_row_l2norm": [_i, _vp, _vp, _vp, _l, _i, _f, _vp],
    "msfwsi_row_l2norm_bwd": [_i, _vp, _vp, _vp, _vp, _l, _i, _vp],
    "msfwsi_softmax_ce": [_i, _vp, _l, _i, _l, _f, _f, _vp, _vp, _i, _vp],
    "msfwsi_nonfinite_check": [_vp, _l, _vp, _vp],
    "msfwsi_scaler_update": [_vp, _vp, _vp, _f, _f, _i, _vp],
    "msfwsi_adam": [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _l, _vp, _vp, _vp, _vp, _i, _vp],
    "msfwsi_adam_step_advance": [_vp, _vp, _vp],
    "msfwsi_cast_lowp": [_i, _vp, _vp, _l, _vp],
    "msfwsi_pad_cast": [_i, _vp, _vp, _l, _i, _i, _vp],
    "msfwsi_unpad_add": [_vp, _vp, _l, _i, _i, _vp],
    "msfwsi_upcast_f32": [_i, _vp, _vp, _l, _vp],
    "msfwsi_zero_f64_2d": [_vp, _l, _i, _l, _vp],
    "msfwsi_seg_stats": [_i, _vp, _i, _vp, _vp, _i, _l, _i, _l, _l, _l, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "msfwsi_seg_scores": [_vp, _vp, _vp, _vp, _i, _i, _d, _vp, _vp],
    "msfwsi_seg_scores_imagewise": [_vp, _vp, _vp, _vp, _i, _i, _d, _vp, _vp],
    "msfwsi_tile_views": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp],
    "msfwsi_tile_crops_u8": [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp],
    "msfwsi_gray_sum": [_vp, _i, _i, _i, _vp, _vp],
    "msfwsi_color_stage": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "msfwsi_blur_sharpen": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "msfwsi_inverse_perm": [_vp, _vp, _l, _i, _vp],
    "msfwsi_img3x3_supported": [_desc],
    "msfwsi_img3x3_pack_weights": [_i, _vp, _vp, _i, _i, _i, _vp],
    "msfwsi_img3x3_fwd": [_desc, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "msfwsi_img3x3_dgrad": [_desc, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "msfwsi_img3x3_s2_dgrad_supported": [_desc],
    "msfwsi_img3x3_s2_dgrad": [_desc, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "msfwsi_panel_gram": [_i, _vp, _vp, _vp, _vp, _vp, _l, _i, _vp],
    "msfwsi_panel_supported": [_desc, _i],
    "msfwsi_panel_pack_weights": [_i, _vp, _vp, _i, _i, _l, _l, _vp],
    "msfwsi_panel_fwd_post": [_desc, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "msfwsi_panel_dgrad": [_desc, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _f, _vp, _vp, _i, _vp],
    "msfwsi_set_tuning": [_i, _l],
    "msfwsi_get_tuning": [_i, _vp],
    "msfwsi_conv3x3_supported": [_desc],
    "msfwsi_conv3x3_stationary": [_desc],
    "msfwsi_conv_wgrad_stationary": [_desc],
    "msfwsi_conv3x3_fwd": [_desc, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    "msfwsi_conv3x3_dgrad": [_desc, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
}

_lib = None


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH) or _stale():
        subprocess.run(["make", "-C", CSRC_DIR, "-j4"], check=True, stdout=subprocess.DEVNULL)
        if _stale():
            # make decides by file times; the binary and its objects are git-ignored and travel by rsync, so an object can
            # be NEWER than a source it was not built from.  The digest says the library is still not of these sources:
            # rebuild everything.
            subprocess.run(["make", "-C", CSRC_DIR, "-j4", "-B"], check=True, stdout=subprocess.DEVNULL)
            if _stale():
                raise MsfwsiHipError(f"{LIB_PATH}: build id {built_id()} != source digest {source_id()} after a full rebuild")
    return LIB_PATH


def id_files():
    """the files the build id covers, in the Makefile's order (csrc/Makefile: ID_FILES)"""
    names = sorted(f for f in os.listdir(CSRC_DIR) if f.endswith((".hip", ".h")))
    return [os.path.join(CSRC_DIR, f) for f in names] + [os.path.join(_HERE, "..", "include", "msfwsi_hip.h"),
                                                         os.path.join(CSRC_DIR, "Makefile")]


def source_id(extra: str = "") -> str:
    """sha256 over the library's sources as the Makefile forms it (`(cat $(ID_FILES); echo "$(EXTRA)") | sha256sum`), 16 hex"""
    import hashlib

    h = hashlib.sha256()
    for path in id_files():
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(extra.encode() + b"\n")
    return h.hexdigest()[:16]


_ID_MARK = b"MSFWSI_BUILD_ID="


def built_id(path: str = None) -> str:
    """the build id baked into a library FILE (csrc/target.hip), read without loading it; "" when it carries none"""
    try:
        with open(path or LIB_PATH, "rb") as f:
            blob = f.read()
    except OSError:
        return ""
    i = blob.find(_ID_MARK)
    return blob[i + len(_ID_MARK):i + len(_ID_MARK) + 16].decode("ascii", "replace") if i >= 0 else ""


def _stale() -> bool:
    """the library was not built from the sources beside it: by CONTENT (sha256), not by file times -- binaries are
    git-ignored and travel by rsync, where a pushed .so can be newer than an edited .hip"""
    try:
        return built_id() != source_id()
    except OSError:
        return True


def load() -> C.CDLL:
    """dlopen the library and type every entry point; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64; it must be the first HIP runtime mapped into the process, otherwise
    # this library binds to /opt/rocm's copy and the two runtimes disagree about devices (hipErrorNoDevice)
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise MsfwsiHipError(
            f"{LIB_PATH} not found: build it with `make -C {CSRC_DIR}` (or __graft_entry__.build()). "
            "There is no CPU fallback for the MSF-WSI hot path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MsfwsiHipError(f"symbol {name} missing from {LIB_PATH}") from e
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name in ("msfwsi_target", "msfwsi_build_id"):
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MsfwsiHipError(f"symbol {name} missing from {LIB_PATH}") from e
        fn.argtypes = []
        fn.restype = C.c_char_p
    # A/B of the C side's dispatch switches (include/msfwsi_hip.h, msfwsi_set_tuning): ONE variable,
    # MSFWSI_TUNING="key=value,key=value" (e.g. "15=1" = one pixel split per weight-gradient tile, "6=0" = no 256 x 256 tile);
    # round 5 had twelve variables, one per key
    spec = os.environ.get("MSFWSI_TUNING", "").strip()
    if spec:
        for item in spec.split(","):
            k, sep, v = item.strip().partition("=")
            if not sep or lib.msfwsi_set_tuning(int(k), int(v)) != 0:
                raise MsfwsiHipError(f"MSFWSI_TUNING: bad entry {item!r} (expected key=value with a key of msfwsi_set_tuning)")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        kind = {-1: "invalid argument", -2: "unsupported configuration"}.get(rc, "error")
        raise MsfwsiHipError(f"{what}: {kind} (rc={rc})")
    raise MsfwsiHipError(f"{what}: hipError_t {rc}")
