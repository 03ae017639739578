"""ctypes binding of the C-ABI library ``libmm2d3d_hip.so`` (declared in ``include/mm2d3d.h``).

The product path has NO fallback: if the HIP library is missing or fails to load, every
operator raises.  PyTorch is used only for device memory and the current HIP stream.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MM_LIB_PATH: a DIAGNOSTIC build of the same sources (tools/diag_lib.sh: parts of a kernel switched off at compile time) - the
# measurement tools' A/B lever, never a fallback: a missing file still raises
LIB_PATH = os.environ.get("MM_LIB_PATH") or os.path.join(_HERE, "libmm2d3d_hip.so")
CSRC = os.path.join(_HERE, "csrc")

_lib = None

vp, i32, i64, sz, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_float, C.c_double

# name -> (restype, argtypes); mirrors include/mm2d3d.h
_PROTOS = {
    "mm_last_error": (C.c_char_p, []),
    "mm_handle_sync_bytes": (sz, []),
    "mm_handle_fault_bytes": (sz, []),
    "mm_create": (i32, [i32, vp, sz, vp, sz, C.POINTER(vp)]),
    "mm_destroy": (i32, [vp]),
    "mm_set_option": (i32, [vp, i32, i32]),
    "mm_get_option": (i32, [vp, i32]),
    "mm_fault_poll": (i32, [vp]),
    "mm_hash_capacity": (i64, [i64]),
    "mm_dedupe_ws_bytes": (sz, [i64]),
    "mm_voxel_dedupe": (i32, [vp, i32, i64, vp, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp, i32, vp, sz, vp]),
    "mm_batch_lower_bound": (i32, [vp, vp, i32, vp, vp]),
    "mm_subm_neighbors": (i32, [vp, i64, i32, vp, vp, i64, vp, vp]),
    "mm_down_neighbors": (i32, [vp, i64, vp, i64, vp, vp]),
    "mm_rulebook_ws_bytes": (sz, [i64, i32]),
    "mm_rulebook_compact": (i32, [vp, i32, i64, vp, vp, vp, vp, vp, i32, vp, sz, vp]),
    "mm_rulebook_csr": (i32, [vp, i32, i64, vp, vp, i32, vp, sz, vp]),
    "mm_spconv_ws_bytes": (sz, [i64, i32, i32, i32]),
    "mm_spconv_apply": (i32, [vp, i32, i32, vp, i32, i32, i64, vp, vp, vp, vp, i32, vp, vp, i32, vp, i64, i32, i32, i32, i32, vp, sz, vp]),
    "mm_spconv_apply_packed": (i32, [vp, i32, i32, vp, i32, i32, i64, vp, vp, vp, vp, i32, vp, vp, i32, vp, i64, i32, i32, i32, vp, i32, vp, sz, vp]),
    "mm_spconv_dw_ws_bytes": (sz, [vp, i32, i32, i32]),
    "mm_spconv_dw": (i32, [vp, i32, i32, vp, i32, i32, vp, vp, vp, i32, vp, i32, i32, vp, sz, vp]),
    "mm_voxelize_ws_bytes": (sz, [i64, i32]),
    "mm_voxelize_batch": (i32, [vp, vp, vp, i32, vp, vp, i32, f32, i32, vp, vp, vp, vp, vp, vp, sz, vp]),
    "mm_project_batch": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "mm_collect_points": (i32, [vp, vp, i64, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "mm_up_neighbors": (i32, [vp, i64, vp, vp, vp]),
    "mm_os_table_ws_bytes": (sz, [i64, i32]),
    "mm_os_table_build": (i32, [vp, i32, i64, i32, i32, vp, vp, vp, vp, sz, vp]),
    "mm_spconv_os_pack_bytes": (sz, [i32, i32, i32]),
    "mm_spconv_os_pack_blocks": (i64, [i32, i32, i32]),
    "mm_spconv_os_pack_desc_fields": (i32, []),
    "mm_spconv_os_pack": (i32, [vp, i64, i32, i32, i32, i32, i32, i32, vp, vp]),
    "mm_spconv_os_pack_batch": (i32, [vp, i32, i64, vp]),
    "mm_spconv_os_apply": (i32, [vp, i32, i32, vp, i32, i32, vp, i32, vp, vp, vp, i64, i32, vp]),
    "mm_spconv_os_pack_bytes_bf16": (sz, [i32, i32, i32]),
    "mm_spconv_os_pack_bf16": (i32, [vp, i64, i32, i32, i32, i32, i32, i32, vp, vp]),
    "mm_spconv_os_pack_batch_bf16": (i32, [vp, i32, i64, vp]),
    "mm_spconv_os_apply_bf16": (i32, [vp, i32, i32, vp, i32, i32, vp, i32, vp, vp, vp, i64, i32, vp]),
    "mm_spconv_dw_bf16": (i32, [vp, i32, i32, vp, i32, i32, vp, vp, vp, i32, vp, i32, i32, vp, sz, vp]),
    "mm_spconv_os_pack_f16": (i32, [vp, i64, i32, i32, i32, i32, i32, i32, vp, vp]),
    "mm_spconv_os_pack_batch_f16": (i32, [vp, i32, i64, vp]),
    "mm_spconv_os_apply_f16": (i32, [vp, i32, i32, vp, i32, i32, vp, i32, vp, vp, vp, i64, i32, vp]),
    "mm_spconv_dw_f16": (i32, [vp, i32, i32, vp, i32, i32, vp, vp, vp, i32, vp, i32, i32, vp, sz, vp]),
    "mm_spconv_dw_partial": (i32, [i32, vp, i32, i32, vp, i32, i32, vp, vp, vp, i32, i32, vp, sz, vp, vp]),
    "mm_spconv_dw_desc_bytes": (i32, []),
    "mm_spconv_dw_reduce_blocks": (i64, [i32, i32]),
    "mm_spconv_dw_reduce_batch": (i32, [vp, i32, i64, vp]),
    "mm_bn_fwd_train_bf16": (i32, [vp, vp, i32, i64, i64, i32, vp, vp, vp, vp, f32, f32, f32, vp, i32, vp, vp, vp, sz, vp]),
    "mm_bn_fwd_eval_bf16": (i32, [vp, i32, i64, i32, vp, vp, vp, vp, f32, f32, vp, i32, vp]),
    "mm_bn_bwd_bf16": (i32, [vp, vp, i32, vp, i32, i64, i64, i32, vp, vp, vp, vp, f32, vp, i32, vp, vp, i32, vp, sz, vp]),
    "mm_bn_fwd_train_f16": (i32, [vp, vp, i32, i64, i64, i32, vp, vp, vp, vp, f32, f32, f32, vp, i32, vp, vp, vp, sz, vp]),
    "mm_bn_fwd_eval_f16": (i32, [vp, i32, i64, i32, vp, vp, vp, vp, f32, f32, vp, i32, vp]),
    "mm_bn_bwd_f16": (i32, [vp, vp, i32, vp, i32, i64, i64, i32, vp, vp, vp, vp, f32, vp, i32, vp, vp, i32, vp, sz, vp]),
    "mm_bn_ws_bytes": (sz, [i32]),
    "mm_bn_fwd_train": (i32, [vp, vp, i32, i64, i64, i32, vp, vp, vp, vp, f32, f32, f32, vp, i32, vp, vp, vp, sz, vp]),
    "mm_bn_fwd_eval": (i32, [vp, i32, i64, i32, vp, vp, vp, vp, f32, f32, vp, i32, vp]),
    "mm_bn_bwd": (i32, [vp, vp, i32, vp, i32, i64, i64, i32, vp, vp, vp, vp, f32, vp, i32, vp, vp, i32, vp, sz, vp]),
    "mm_point_ws_bytes": (sz, [i32, i32]),
    "mm_gate_fwd": (i32, [vp, i64, i32, vp, vp, vp, vp, vp]),
    "mm_gate_bwd": (i32, [vp, vp, vp, i64, i32, vp, vp, vp, vp, i32, vp, sz, vp]),
    "mm_segment_reduce": (i32, [vp, i32, i32, vp, vp, i64, i32, vp, i32, vp]),
    "mm_row_gather": (i32, [vp, i32, i32, vp, vp, i32, i64, vp, i32, vp]),
    "mm_linear_fwd": (i32, [vp, i32, i64, i32, i32, vp, vp, vp, i32, vp]),
    "mm_linear_bwd": (i32, [vp, i32, vp, i32, i64, i32, i32, vp, vp, i32, i32, vp, vp, i32, vp, sz, vp]),
    "mm_loss_ws_bytes": (sz, []),
    "mm_ce_fwd": (i32, [vp, i32, vp, vp, i64, i32, i64, vp, vp, sz, vp]),
    "mm_ce_bwd": (i32, [vp, i32, vp, vp, i64, i32, i64, vp, vp, vp, i32, vp]),
    "mm_kl_fwd": (i32, [vp, i32, vp, i32, i64, i32, vp, vp, sz, vp]),
    "mm_kl_bwd": (i32, [vp, i32, vp, i32, i64, i32, vp, vp, i32, vp]),
    "mm_lift_gather": (i32, [vp, i64, vp, i64, i32, vp, vp]),
    "mm_lift_scatter": (i32, [vp, i32, vp, vp, vp, i64, i64, vp, vp]),
    "mm_lift_scatter_runs": (i32, [vp, i32, vp, vp, vp, i64, i64, vp, vp]),
    "mm_lift_index_ws_bytes": (sz, [i64]),
    "mm_lift_index": (i32, [vp, vp, i32, i64, i32, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
    "mm_lift_gather_key": (i32, [vp, i64, i64, i64, i64, vp, i64, i32, i32, i32, vp, vp]),
    "mm_lift_scatter_key": (i32, [vp, i32, vp, vp, i64, i32, i32, i64, i64, i64, i64, vp, vp]),
    "mm_eval_confusion": (i32, [vp, i32, vp, i32, vp, i64, i32, i64, vp, vp]),
    "mm_adamw_step": (i32, [vp, vp, vp, vp, i64, f64, f64, f64, f64, f64, i64, f64, vp, i32, vp]),
    "mm_grad_nonfinite": (i32, [vp, i64, vp, vp]),
    "mm_amp_coef_bytes": (i32, []),
    "mm_amp_prepare": (i32, [vp, vp, i32, vp, i32, f64, f64, f64, f64, f64, f64, vp, vp]),
    "mm_adamw_step_dev": (i32, [vp, vp, vp, vp, i64, vp, vp]),
    "mm_amp_update": (i32, [vp, vp, vp, i32, f64, f64, i32, vp]),
    "mm_conv2d_gemm": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp,
                             vp, i32, i64, i32, vp, vp, i64, vp, i32, vp]),
    "mm_conv2d_3x3s1": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp, i32, vp, i32, vp]),
    "mm_conv2d_3x3s1_pair": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp, i32, vp, vp, i32, vp]),
    "mm_conv2d_dgrad_s2": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, i32, i32, i32, vp, i32, i32, vp, i32, vp]),
    "mm_conv2d_gemm_stat_rows": (i64, [i64, i32]),
    "mm_conv2d_3x3s1_stat_rows": (i64, [i32, i32, i32]),
    "mm_conv2d_wgrad_ws_bytes": (sz, [i64, i32, i32, i32]),
    "mm_conv2d_wgrad": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, i64, i64, i64, i32, vp, sz,
                              vp]),
    "mm_conv2d_wgrad3x3_pair": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp, i64, i64, i64, i32, vp, sz, vp]),
    "mm_conv2d_wgrad_slabs": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, sz, vp, vp]),
    "mm_conv2d_wgrad3x3_pair_slabs": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp, sz, vp, vp]),
    "mm_conv2d_wgrad_reduce_desc_bytes": (i32, []),
    "mm_conv2d_wgrad_reduce_blocks": (i64, [i32, i32, i32]),
    "mm_conv2d_wgrad_reduce_batch": (i32, [vp, i32, i64, vp]),
    "mm_conv2d_f32": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, i64, i64, i64, vp, vp]),
    "mm_conv2d_f32_wgrad_ws_bytes": (sz, [i64, i32, i32, i32, i32]),
    "mm_conv2d_f32_wgrad": (i32, [vp, i32, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, i64, i64, i64,
                                  i32, vp, sz, vp]),
    "mm_colsum_f32": (i32, [vp, i32, i64, i32, vp, i32, vp]),
    "mm_pack_weights_bf16": (i32, [vp, vp, i32, i32, i32, i32, i64, i64, i64, i64, vp]),
    "mm_pack_weights_bf16_batch": (i32, [vp, i32, i64, vp]),
    "mm_conv2d_stem7_stat_rows": (i64, [i32, i32, i32]),
    "mm_conv2d_stem7": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, i32, vp]),
    "mm_stem_prep": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "mm_bn2d_ws_bytes": (sz, [i32]),
    "mm_bn2d_fwd_train": (i32, [vp, vp, i32, vp, i32, i64, i64, i32, vp, vp, vp, vp, vp, f32, f32, i32, vp, i32, vp, vp, vp, sz, vp]),
    "mm_bn2d_single_launch": (i32, [vp, i64, i64, i32, i32]),
    "mm_bn2d_fwd_train_pre": (i32, [vp, i32, vp, i32, i64, i64, i32, vp, vp, vp, vp, vp, f32, f32, i32, vp, i32, vp, vp, vp, i64, vp, sz, vp]),
    "mm_bn2d_fwd_train_pre_pool": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, f32, f32, vp, i32, vp, vp, vp, vp, vp, i64, vp, sz, vp]),
    "mm_bn2d_bwd_pool": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp, sz, vp]),
    "mm_bn2d_fwd_train_pair": (i32, [vp, vp, vp, i64, i64, i32, f32, f32, i32, vp, sz, vp]),
    "mm_bn2d_bwd_pair": (i32, [vp, vp, vp, i32, i64, i64, i32, i32, vp, sz, vp]),
    "mm_bn2d_fwd_eval": (i32, [vp, i32, vp, i32, i64, i32, vp, vp, vp, vp, f32, i32, vp, i32, vp]),
    "mm_bn2d_bwd": (i32, [vp, vp, i32, vp, i32, vp, i32, vp, i32, i32, i64, i64, i32, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, i32, vp, sz, vp]),
    "mm_colsum_bf16": (i32, [vp, i32, i64, i32, vp, i32, vp, sz, vp]),
    "mm_copy_rows_bf16": (i32, [vp, i64, vp, i64, i64, i32, vp]),
    "mm_concat_bf16": (i32, [vp, vp, i32, vp, i64, i32, vp]),
    "mm_maxpool3x3s2_fwd": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    "mm_maxpool3x3s2_bwd": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, vp]),
    "mm_head_ws_bytes": (sz, [i32, i32, i32, i32, i32, i32, i32]),
    "mm_head_fwd": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, sz, vp]),
    "mm_head_bwd": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, sz, vp]),
}


class Bn2dFwdArgs(C.Structure):
    """include/mm2d3d.h mm_bn2d_fwd_args"""
    _fields_ = [("x", C.c_void_p), ("ld_x", C.c_int), ("res", C.c_void_p), ("ld_r", C.c_int), ("weight", C.c_void_p), ("bias", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p), ("y", C.c_void_p),
                ("ld_y", C.c_int), ("save_mean", C.c_void_p), ("save_invstd", C.c_void_p)]


class Bn2dBwdArgs(C.Structure):
    """include/mm2d3d.h mm_bn2d_bwd_args"""
    _fields_ = [("x", C.c_void_p), ("ld_x", C.c_int), ("dy", C.c_void_p), ("ld_dy", C.c_int), ("dy2", C.c_void_p), ("ld_dy2", C.c_int),
                ("yout", C.c_void_p), ("ld_y", C.c_int), ("weight", C.c_void_p), ("bias", C.c_void_p), ("save_mean", C.c_void_p),
                ("save_invstd", C.c_void_p), ("dx", C.c_void_p), ("ld_dx", C.c_int), ("dres", C.c_void_p), ("ld_dr", C.c_int),
                ("dweight", C.c_void_p), ("dbias", C.c_void_p)]


class HipLibraryMissing(RuntimeError):
    pass


# the dense 2D kernels built for IEEE fp16 storage (csrc/h16.h): same prototypes under the suffix _f16
H16_2D = {
    "mm_conv2d_gemm": "mm_conv2d_gemm_f16",
    "mm_conv2d_dgrad_s2": "mm_conv2d_dgrad_s2_f16",
    "mm_conv2d_3x3s1": "mm_conv2d_3x3s1_f16",
    "mm_conv2d_3x3s1_pair": "mm_conv2d_3x3s1_pair_f16",
    "mm_conv2d_gemm_stat_rows": "mm_conv2d_gemm_stat_rows_f16",
    "mm_conv2d_3x3s1_stat_rows": "mm_conv2d_3x3s1_stat_rows_f16",
    "mm_bn2d_fwd_train_pre": "mm_bn2d_fwd_train_pre_f16",
    "mm_bn2d_single_launch": "mm_bn2d_single_launch_f16",
    "mm_bn2d_fwd_train_pair": "mm_bn2d_fwd_train_pair_f16",
    "mm_bn2d_fwd_train_pre_pool": "mm_bn2d_fwd_train_pre_pool_f16",
    "mm_bn2d_bwd_pool": "mm_bn2d_bwd_pool_f16",
    "mm_bn2d_bwd_pair": "mm_bn2d_bwd_pair_f16",
    "mm_conv2d_wgrad_ws_bytes": "mm_conv2d_wgrad_ws_bytes_f16",
    "mm_conv2d_wgrad": "mm_conv2d_wgrad_f16",
    "mm_conv2d_wgrad3x3_pair": "mm_conv2d_wgrad3x3_pair_f16",
    "mm_conv2d_wgrad_slabs": "mm_conv2d_wgrad_slabs_f16",
    "mm_conv2d_wgrad3x3_pair_slabs": "mm_conv2d_wgrad3x3_pair_slabs_f16",
    "mm_conv2d_wgrad_reduce_desc_bytes": "mm_conv2d_wgrad_reduce_desc_bytes_f16",
    "mm_conv2d_wgrad_reduce_blocks": "mm_conv2d_wgrad_reduce_blocks_f16",
    "mm_conv2d_wgrad_reduce_batch": "mm_conv2d_wgrad_reduce_batch_f16",
    "mm_stem_prep": "mm_stem_prep_f16",
    "mm_conv2d_stem7_stat_rows": "mm_conv2d_stem7_stat_rows_f16",
    "mm_conv2d_stem7": "mm_conv2d_stem7_f16",
    "mm_pack_weights_bf16": "mm_pack_weights_f16",
    "mm_pack_weights_bf16_batch": "mm_pack_weights_f16_batch",
    "mm_bn2d_ws_bytes": "mm_bn2d_ws_bytes_f16",
    "mm_bn2d_fwd_train": "mm_bn2d_fwd_train_f16",
    "mm_bn2d_fwd_eval": "mm_bn2d_fwd_eval_f16",
    "mm_bn2d_bwd": "mm_bn2d_bwd_f16",
    "mm_colsum_bf16": "mm_colsum_f16",
    "mm_maxpool3x3s2_fwd": "mm_maxpool3x3s2_fwd_f16",
    "mm_maxpool3x3s2_bwd": "mm_maxpool3x3s2_bwd_f16",
    "mm_head_ws_bytes": "mm_head_ws_bytes_f16",
    "mm_head_fwd": "mm_head_fwd_f16",
    "mm_head_bwd": "mm_head_bwd_f16",
}
for _a, _b in H16_2D.items():
    _PROTOS[_b] = _PROTOS[_a]


def build(verbose: bool = False) -> str:
    """Compile every .hip source for gfx950 into the in-tree shared library (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libmm2d3d_hip.so failed:\n" + (r.stdout or "") + (r.stderr or ""))
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C mm2d3d_amd/csrc`). There is no CPU fallback in the product path."
            )
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)  # AttributeError if the library does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


# ---------------------------------------------------------------------------------------------- per-device handles
# include/mm2d3d.h: the library keeps no process-wide state; what outlives a call (the grid-barrier words and the fault word of
# the single-launch batch norms, their switches) lives in a handle over memory supplied from here.  The environment variables
# that used to be read inside the library are read HERE, once, and passed on explicitly.
OPT_BN2D_FUSED, OPT_BN3D_FUSED = 0, 1
SPCONV_DEFAULT, SPCONV_FP32, SPCONV_TWO_TERMS, SPCONV_DW_NARROW, SPCONV_DW16_ELEM = 0, 1, 2, 4, 8


def _env_int(name, default):
    v = os.environ.get(name)
    try:
        return int(v) if v not in (None, "") else default
    except ValueError:
        return default


def spconv_mode_from_env():
    """mode argument of the fp32 sparse engines (MM_SPCONV_FP32 / MM_SPCONV_SPLIT=2 / MM_DW_WIDE=0: diagnostics)."""
    mode = SPCONV_FP32 if os.environ.get("MM_SPCONV_FP32") else (SPCONV_TWO_TERMS if _env_int("MM_SPCONV_SPLIT", 3) == 2 else SPCONV_DEFAULT)
    if _env_int("MM_DW_WIDE", 1) == 0:
        mode |= SPCONV_DW_NARROW
    if _env_int("MM_DW16_ELEM", 0):
        mode |= SPCONV_DW16_ELEM
    return mode


class Handle:
    """mm_create over torch-allocated memory: zero-filled barrier words on the device, a pinned (device-mapped) fault word.
    ``bn2d_fused`` / ``bn3d_fused``: bit 0 = forward, bit 1 = backward single-launch kernels (default: MM_BN2D_FUSED /
    MM_BN_FUSED of the environment, else 3)."""

    def __init__(self, device=None, bn2d_fused=None, bn3d_fused=None):
        L = lib()
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("mm2d3d_amd: a handle belongs to a GPU (HIP path only, no CPU fallback)")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self._sync = torch.zeros(int(L.mm_handle_sync_bytes()), dtype=torch.uint8, device=dev)
        self._fault = torch.zeros(int(L.mm_handle_fault_bytes()), dtype=torch.uint8).pin_memory()
        torch.cuda.synchronize(dev)  # the zero fill has run before any kernel of any stream meets at these words
        out = vp()
        check(L.mm_create(dev.index, self._sync.data_ptr(), self._sync.numel(), self._fault.data_ptr(), self._fault.numel(),
                          C.byref(out)), "mm_create")
        self.h = out.value
        self.set(OPT_BN2D_FUSED, _env_int("MM_BN2D_FUSED", 3) if bn2d_fused is None else bn2d_fused)
        self.set(OPT_BN3D_FUSED, _env_int("MM_BN_FUSED", 3) if bn3d_fused is None else bn3d_fused)

    def set(self, option, value):
        """Returns the previous value."""
        prev = int(lib().mm_set_option(self.h, option, int(value)))
        if prev < 0:
            check(prev, "mm_set_option")
        return prev

    def get(self, option):
        return int(lib().mm_get_option(self.h, option))

    def fault_poll(self):
        return int(lib().mm_fault_poll(self.h))

    def close(self):
        if getattr(self, "h", None):
            lib().mm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DEFAULT_HANDLES = {}
_HANDLE_STACK = []


def handle(device=None):
    """The handle operators launch through: the innermost ``use()`` handle of the device, else the device's default handle."""
    idx = torch.cuda.current_device() if device is None or torch.device(device).index is None else torch.device(device).index
    for h in reversed(_HANDLE_STACK):
        if h.device.index == idx:
            return h
    h = _DEFAULT_HANDLES.get(idx)
    if h is None:
        h = _DEFAULT_HANDLES[idx] = Handle(torch.device("cuda", idx))
    return h


@contextlib.contextmanager
def use(h):
    """Operators of ``h``'s device launch through ``h`` inside this context (a trainer with its own switches)."""
    _HANDLE_STACK.append(h)
    try:
        yield h
    finally:
        _HANDLE_STACK.remove(h)


def bn2d_set_fused(mask: int, device=None) -> int:
    """Single-launch BatchNorm2d kernels of the CURRENT handle (bit 0 forward, bit 1 backward); returns the previous mask."""
    return handle(device).set(OPT_BN2D_FUSED, mask)


def bn3d_set_fused(mask: int, device=None) -> int:
    return handle(device).set(OPT_BN3D_FUSED, mask)


def fault_poll(device=None) -> int:
    """1 if a single-launch batch-norm kernel of the current handle gave up at its grid barrier since the last call."""
    idx = torch.cuda.current_device() if device is None or torch.device(device).index is None else torch.device(device).index
    hs = [h for h in _HANDLE_STACK if h.device.index == idx]
    if idx in _DEFAULT_HANDLES:
        hs.append(_DEFAULT_HANDLES[idx])
    bad = 0
    for h in hs:
        bad |= h.fault_poll()
    return bad


# Listeners called right BEFORE a grid-barrier kernel (a single-launch batch norm: csrc/fused_bn.h) is queued, with backward = True /
# False.  The data-parallel reducer keeps its collectives away from them (mm2d3d_amd/ddp.py, "tail" schedule): a barrier kernel
# needs every workgroup resident at once and must not share the GPU with a kernel that holds CUs for as long as a peer is late.
BARRIER_LISTENERS = []  # weakref.WeakMethod objects (a reducer that is gone stops listening)


def add_barrier_listener(bound_method):
    import weakref

    BARRIER_LISTENERS.append(weakref.WeakMethod(bound_method))


def before_barrier_kernel(backward, sparse=False):
    """``sparse``: the kernel belongs to the sparse branch's batch norms (short: a few thousand rows to a few hundred thousand)."""
    dead = False
    for ref in BARRIER_LISTENERS:
        cb = ref()
        if cb is None:
            dead = True
        else:
            cb(backward, sparse)
    if dead:
        BARRIER_LISTENERS[:] = [r for r in BARRIER_LISTENERS if r() is not None]


def exported_symbols():
    return sorted(_PROTOS)


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().mm_last_error()
        raise RuntimeError(f"libmm2d3d_hip {what} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


class _Workspace:
    """Grow-only scratch buffers, one per (device, slot, current stream): every kernel that uses a buffer is enqueued on the
    stream the buffer belongs to, so the stream's own order serialises them (ops run on the current
    torch HIP stream)."""

    def __init__(self):
        self.bufs = {}
        self.default_slot = "main"

    def get(self, nbytes: int, device, slot: str = None):
        # one buffer per (device, slot, stream): kernels of different streams never share scratch memory
        key = (device.index if device.index is not None else torch.cuda.current_device(), slot or self.default_slot,
               torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
            self.bufs[key] = buf
        return buf


workspace = _Workspace()


@contextlib.contextmanager
def workspace_slot(name):
    """Kernels enqueued on another stream must not share the scratch buffer of the main stream: inside this context
    ``workspace.get`` hands out the buffer ``name``."""
    old = workspace.default_slot
    workspace.default_slot = name
    try:
        yield
    finally:
        workspace.default_slot = old


def require_cuda(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"mm2d3d_amd: {name} must live on the GPU (HIP path only, no CPU fallback)")
