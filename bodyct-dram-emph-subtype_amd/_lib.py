"""ctypes binding of libdram_hip.so (the C ABI declared in include/dram_hip.h).

The product path has NO fallback: if the library is missing or fails to load,
``load()`` raises.  ``import torch`` happens first on purpose -- libdram_hip.so needs
``libamdhip64.so.7`` and must bind to the HIP runtime PyTorch has already loaded
(same soname), so that torch's streams and device pointers are valid inside it.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_void_p

import torch  # noqa: F401  (must precede CDLL: loads the HIP runtime we bind to)

from . import _build

P, I, LL, F, D, SZ = c_void_p, c_int, c_longlong, c_float, c_double, c_size_t


class DramConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("B", "D", "H", "W", "Cin", "Do", "Ho", "Wo", "Cout", "k", "stride", "pad", "dil", "flags")]


class DramTensorRef(ctypes.Structure):
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("n", ctypes.c_int64)]


class DramAugment(ctypes.Structure):
    _fields_ = [("flags", ctypes.c_int32), ("n_boxes", ctypes.c_int32), ("boxes", (ctypes.c_int32 * 6) * 10),
                ("flip_axes", ctypes.c_int32), ("sigma", c_float), ("box_lo", c_float * 3), ("box_hi", c_float * 3)]


class DramProfRecord(ctypes.Structure):
    _fields_ = [("family", ctypes.c_int32), ("variant", ctypes.c_int32), ("mfma_flops", c_double),
                ("alg_flops", c_double), ("hbm_bytes", c_double), ("ms", c_float), ("pad_", c_float)]


class DramChunkRef(ctypes.Structure):
    _fields_ = [("tensor", ctypes.c_int32), ("pad", ctypes.c_int32), ("offset", ctypes.c_int64)]


class DramPackRef(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("off_f", ctypes.c_int64), ("off_b", ctypes.c_int64), ("Cout", ctypes.c_int32),
                ("Cin", ctypes.c_int32), ("taps", ctypes.c_int32), ("pad", ctypes.c_int32)]


DP = ctypes.POINTER(DramConvDesc)

# name -> (restype, argtypes); mirrors include/dram_hip.h one to one
SIGNATURES = {
    "dram_version": (I, []),
    "dram_build_info": (c_char_p, []),
    "dram_abi_hash": (c_char_p, []),
    "dram_stream_capture_id": (ctypes.c_ulonglong, [P]),
    "dram_profile_family_name": (c_char_p, [I]),
    "dram_profile_family_is_mfma": (I, [I]),
    "dram_profile_start": (I, [I]),
    "dram_profile_stop": (I, []),
    "dram_profile_dropped": (I, []),
    "dram_profile_read": (I, [P, I]),
    "dram_pack_conv_weight": (I, [P, P, P, I, I, I, P]),
    "dram_conv3d_fwd": (I, [P, P, P, P, P, DP, P]),
    "dram_conv3d_bwd_data": (I, [P, P, P, P, P, DP, P]),
    "dram_conv3d_bwd_weight_workspace": (SZ, [DP]),
    "dram_conv3d_bwd_weight": (I, [P, P, P, DP, P, SZ, P]),
    "dram_conv_num_mtiles": (I, [DP]),
    "dram_conv_algo": (I, [DP]),
    "dram_conv1x1_applicable": (I, [DP]),
    "dram_conv1x1_num_stat_rows": (I, [DP]),
    "dram_conv1x1_fwd": (I, [P, P, P, P, P, DP, P]),
    "dram_conv1x1_bwd_data": (I, [P, P, P, P, P, DP, P]),
    "dram_conv1x1_bwd_weight_workspace": (SZ, [DP]),
    "dram_conv1x1_bwd_weight": (I, [P, P, P, DP, P, SZ, P]),
    "dram_wgrad_w2d_applicable": (I, [DP]),
    "dram_wgrad_w2d_workspace": (SZ, [DP]),
    "dram_wgrad_w2d": (I, [P, P, P, DP, P, SZ, P]),
    "dram_wino2d_applicable": (I, [DP]),
    "dram_wino2d_num_stat_rows": (I, [DP]),
    "dram_wino2d_pack_weight": (I, [P, P, P, I, I, P]),
    "dram_wino2d_conv3d_fwd": (I, [P, P, P, P, P, DP, P]),
    "dram_wino2d_conv3d_bwd_data": (I, [P, P, P, P, P, DP, P]),
    "dram_wino_applicable": (I, [DP]),
    "dram_conv_wgrad_algo": (I, [DP]),
    "dram_wino_num_points": (I, [DP]),
    "dram_wino_num_points_bwd": (I, [DP]),
    "dram_wino_pack_weight": (I, [P, P, P, DP, P]),
    "dram_wino_workspace": (SZ, [DP, I]),
    "dram_wino_num_stat_rows": (I, [DP]),
    "dram_wino_v_elems": (SZ, [DP]),
    "dram_wino_conv3d_fwd": (I, [P, P, P, P, P, P, DP, P, SZ, P]),
    "dram_wino_prologue_supported": (I, [DP]),
    "dram_wino_conv3d_fwd_cat": (I, [P, I, P, I, P, P, P, P, P, DP, P, SZ, P]),
    "dram_wino_conv3d_fwd_bn": (I, [P, P, P, P, P, P, P, P, DP, P, SZ, P]),
    "dram_wino_conv3d_bwd_data": (I, [P, P, P, P, P, DP, P, SZ, P]),
    "dram_wino_num_stat_rows_bwd": (I, [DP]),
    "dram_wino_conv3d_bwd_data_bn": (I, [P, P, P, P, P, P, P, P, P, DP, P, SZ, P]),
    "dram_wino_conv3d_bwd_weight": (I, [P, P, P, P, DP, P, SZ, P]),
    "dram_stem_num_tiles": (I, [I, I, I, I]),
    "dram_stem_fwd": (I, [P, P, P, P, I, I, I, I, P]),
    "dram_stem_bwd_weight_workspace": (SZ, [I, I, I, I]),
    "dram_stem_bwd_weight": (I, [P, P, P, I, I, I, I, P, SZ, P]),
    "dram_reduce_partials_stages": (I, [I]),
    "dram_reduce_partials": (I, [P, P, P, I, I, I, D, I, P]),
    "dram_fold_partials_stages": (I, [I]),
    "dram_fold_partials": (I, [P, P, P, P, P, I, I, I, D, I, P]),
    "dram_bn_fold_finalize": (I, [P, P, P, I, I, D, P, P, P, P, F, F, I, P, P, P, P, P]),
    "dram_bn_finalize": (I, [P, D, P, P, P, P, P, F, F, I, P, P, P, P, I, P]),
    "dram_bn_apply": (I, [P, P, P, P, I, I, I, I, I, P, I, I, I, I, I, I, P]),
    "dram_colsum_nparts": (I, [LL, I]),
    "dram_bn_bwd_reduce": (I, [P, P, P, P, P, P, P, P, LL, I, I, P]),
    "dram_bn_bwd_apply_nparts": (I, [LL, I]),
    "dram_bn_bwd_apply": (I, [P, P, P, P, P, P, P, P, P, D, P, P, P, LL, I, I, P]),
    "dram_colsum": (I, [P, P, LL, I, P]),
    "dram_maxpool_fwd": (I, [P, P, P, I, I, I, I, I, P]),
    "dram_maxpool_bwd": (I, [P, P, P, I, P, I, I, I, I, I, P]),
    "dram_bn_maxpool_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, P]),
    "dram_bn_maxpool_fwd_bf16": (I, [P, P, P, P, P, P, I, I, I, I, I, P]),
    "dram_upcat_fwd": (I, [P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "dram_upcat_bwd": (I, [P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "dram_upmix_stat_rows": (I, [LL]),
    "dram_upmix_axis_fwd": (I, [P, I, P, LL, I, I, I, P]),
    "dram_upmix_axis_fwd_final": (I, [P, P, P, I, P, LL, I, I, I, P]),
    "dram_upmix_axis_bwd": (I, [P, I, P, I, LL, I, I, I, P]),
    "dram_upmix_split_weight": (I, [P, P, P, I, I, I, P]),
    "dram_upmix_merge_wgrad": (I, [P, P, P, I, I, I, P]),
    "dram_head_nblk": (I, [LL]),
    "dram_head_fwd": (I, [P, P, P, P, I, I, I, P, P, I, I, I, I, I, I, P]),
    "dram_head_bwd_nparts": (I, [LL]),
    "dram_head_bwd": (I, [P, P, P, P, P, P, I, I, I, P, P, I, I, I, I, I, I, P]),
    # bf16 storage path (same argument lists as the fp32 namesakes; activation tensors are bf16)
    "dram_pack_conv_weight_bf16_multi": (I, [P, P, I, P, D, P]),
    "dram_pack_conv_weight_bf16_tiles": (LL, [I, I, I]),
    "dram_cast_f32_to_bf16": (I, [P, P, LL, P]),
    "dram_cast_bf16_to_f32": (I, [P, P, LL, P]),
    "dram_s2d_bf16": (I, [P, P, I, I, I, I, I, P]),
    "dram_d2s_bf16": (I, [P, P, P, P, I, I, I, I, I, P]),
    "dram_s2_embed_weight": (I, [P, P, I, I, P]),
    "dram_s2_extract_wgrad": (I, [P, P, I, I, P]),
    "dram_conv_bf16_supported": (I, [DP]),
    "dram_conv_bf16_num_stat_rows": (I, [DP]),
    "dram_pack_conv_weight_bf16": (I, [P, P, P, I, I, I, P]),
    "dram_conv3d_fwd_bf16": (I, [P, P, P, P, P, DP, P]),
    "dram_conv3d_bwd_data_bf16": (I, [P, P, P, P, P, DP, P]),
    "dram_conv3d_bwd_weight_bf16_workspace": (SZ, [DP]),
    "dram_conv3d_bwd_weight_bf16": (I, [P, P, P, DP, P, SZ, P]),
    "dram_stem_fwd_bf16": (I, [P, P, P, P, I, I, I, I, P]),
    "dram_stem_bwd_weight_bf16": (I, [P, P, P, I, I, I, I, P, SZ, P]),
    "dram_stem_fwd_bf16mm": (I, [P, P, P, P, I, I, I, I, P]),
    "dram_stem_bwd_weight_bf16mm": (I, [P, P, P, I, I, I, I, P, SZ, P]),
    "dram_bn_apply_bf16": (I, [P, P, P, P, I, I, I, I, I, P, I, I, I, I, I, I, P]),
    "dram_bn_bwd_reduce_bf16": (I, [P, P, P, P, P, P, P, P, LL, I, I, P]),
    "dram_bn_bwd_apply_bf16": (I, [P, P, P, P, P, P, P, P, P, D, P, P, P, LL, I, I, P]),
    "dram_colsum_bf16": (I, [P, P, LL, I, P]),
    "dram_maxpool_fwd_bf16": (I, [P, P, P, I, I, I, I, I, P]),
    "dram_maxpool_bwd_bf16": (I, [P, P, P, I, P, I, I, I, I, I, P]),
    "dram_upcat_fwd_bf16": (I, [P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "dram_upcat_bwd_bf16": (I, [P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "dram_head_fwd_bf16": (I, [P, P, P, P, I, I, I, P, P, I, I, I, I, I, I, P]),
    "dram_head_bwd_bf16": (I, [P, P, P, P, P, P, I, I, I, P, P, I, I, I, I, I, I, P]),
    "dram_segloss_nblk": (I, [LL]),
    "dram_segloss_fwd": (I, [P, P, P, P, P, I, I, I, P, I, I, I, I, F, P]),
    "dram_segloss_bwd": (I, [P, P, P, P, P, I, I, I, P, P, P, I, I, I, I, F, P]),
    "dram_regloss_tail": (I, [P, I, P, P, P, P, P, P, P, I, P, I, I, D, D, D, D, P, P, P, P]),
    "dram_upproject_nblk": (I, [LL]),
    "dram_upproject": (I, [P, P, P, P, I, I, I, I, I, I, I, P]),
    "dram_adam_multi": (I, [P, P, I, F, F, F, F, F, F, F, F, P]),
    "dram_adam_multi_dev": (I, [P, P, I, P, P]),
    "dram_sgd_multi": (I, [P, P, I, F, F, F, I, F, P]),
    "dram_window_stats_nblk": (I, [LL]),
    "dram_window_stats": (I, [P, P, LL, F, F, P]),
    "dram_prep_image": (I, [P, P, P, P, I, I, I, I, I, I, F, F, P]),
    "dram_prep_mask": (I, [P, P, P, I, I, I, I, I, I, P]),
    "dram_add": (I, [P, P, P, LL, P]),
    "dram_resample_paste": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dram_minmax_nblk": (I, [LL]),
    "dram_minmax": (I, [P, P, LL, P]),
    "dram_augment_image": (I, [P, P, P, P, I, I, I, P, P]),
    "dram_augment_mask": (I, [P, P, I, I, I, P, P]),
}

OPT_CHUNK = 16384
ABI_VERSION = 7
_LIB = None


def lib_path() -> str:
    return _build.LIB_PATH


def load(build: bool = True) -> ctypes.CDLL:
    """Load libdram_hip.so.  With hipcc present the (content-hash incremental) build runs first,
    so a library older than the sources is never loaded; either way the library must carry the
    fingerprint of THIS include/dram_hip.h -- a stale build would be called with shifted
    arguments (wild device writes), so it is refused."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if build and _build.have_hipcc():
        _build.build_library()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.dram_version() != ABI_VERSION:
        raise RuntimeError("libdram_hip.so ABI version mismatch")
    got, want = lib.dram_abi_hash().decode(), _build.abi_hash()
    if got != want:
        raise RuntimeError(f"libdram_hip.so was built from another include/dram_hip.h (library {got}, header {want}): "
                           "rebuild with __graft_entry__.build()")
    _LIB = lib
    return lib
