"""ctypes binding of libmiso_hip.so (C ABI in include/miso_hip.h).

The library is the product: there is no CPU or PyTorch fallback for the hot ops.
A missing library, or a call with tensors that are not on a HIP device, raises.
"""
from __future__ import annotations

import ctypes as C
import os

# torch must be imported BEFORE libmiso_hip.so is loaded: both need
# libamdhip64.so.7 and the process must end up with the ONE HIP runtime torch
# ships (loading ROCm's copy first leaves torch without a device).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MISO_HIP_LIB: a dev override for A/B runs of differently built libraries, tools/train_ab.sh)
LIB_PATH = os.environ.get("MISO_HIP_LIB") or os.path.join(_HERE, "libmiso_hip.so")

MAX_LEVELS = 8
MAX_LINEAR = 4
RAY_MAX_BINS = 64
ADAM_CHUNK = 64

F_ALIGN_CORNERS = 1
F_PAD_BORDER = 2
F_COORDS_NORMALIZED = 4
F_GRAD_OVERWRITE = 8
F_GRAD_SDF_SORTED = 16
F_GRAD_ZEROED = 32
F_CROWDED = 64
F_EXACT_F32 = 128
F_FULL_TRIPS = 256
F_ATLAS_NO_BOUND = 512
LOSS_SLOTS = 512

E_BADARG = 2001
E_UNSUPPORTED = 2002
E_TOOLARGE = 2003


class Level(C.Structure):
    _fields_ = [("data", C.c_void_p), ("grad", C.c_void_p),
                ("C", C.c_int32), ("Z", C.c_int32), ("Y", C.c_int32), ("X", C.c_int32),
                ("sC", C.c_int64), ("sZ", C.c_int64), ("sY", C.c_int64), ("sX", C.c_int64),
                ("grad_touched", C.c_void_p)]


class LmTrack(C.Structure):
    _fields_ = [("coords_frame", C.c_void_p), ("target", C.c_void_p), ("valid", C.c_void_p), ("frame_ids", C.c_void_p),
                ("stride_target", C.c_int64), ("stride_valid", C.c_int64), ("stride_frame_ids", C.c_int64),
                ("valid_is_bool", C.c_int32), ("n", C.c_int64), ("keyframe_id", C.c_int64), ("trunc_dist", C.c_float),
                ("R_base", C.c_void_p), ("t_base", C.c_void_p), ("rot_correction", C.c_void_p),
                ("trans_correction", C.c_void_p), ("loss_type", C.c_int32), ("gm_scale", C.c_float),
                ("lm_lambda", C.c_float), ("pose", C.c_void_p), ("coords_world", C.c_void_p), ("sdf", C.c_void_p),
                ("grad", C.c_void_p), ("ones", C.c_void_p), ("relu_mask", C.c_void_p), ("sums", C.c_void_p),
                ("info", C.c_void_p), ("sanitized", C.c_void_p)]


class TrackAdam(C.Structure):
    _fields_ = [("s", LmTrack), ("loss_type", C.c_int32), ("weight_sdf", C.c_float), ("gm_scale", C.c_float),
                ("grad_pred", C.c_void_p), ("adam_table", C.c_void_p), ("adam_table_len", C.c_int32),
                ("state", C.c_void_p), ("loss_ring", C.c_void_p), ("ring_len", C.c_int32)]


class Grid(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("ignore_mask", C.c_uint32),
                ("bound_min", C.c_float * 3), ("bound_max", C.c_float * 3),
                ("flags", C.c_uint32), ("level", Level * MAX_LEVELS)]


class Mlp(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("hidden_dim", C.c_int32), ("out_dim", C.c_int32),
                ("n_linear", C.c_int32),
                ("weight", C.c_void_p * MAX_LINEAR), ("bias", C.c_void_p * MAX_LINEAR)]


class Sorted(C.Structure):
    _fields_ = [("tiles_per_axis", C.c_int32), ("x_sorted", C.c_void_p), ("xn_sorted", C.c_void_p),
                ("perm", C.c_void_p), ("tile_offsets", C.c_void_p),
                ("pull_queue", C.c_void_p), ("pull_queue_ints", C.c_int64)]


ADAM_MAX_TENSORS = 8


class AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("active", C.c_void_p), ("touched", C.c_void_p), ("numel", C.c_int64), ("zero_grad", C.c_int32),
                ("reserved", C.c_int32)]


class RayFrames(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("normals", C.c_void_p), ("T_WC", C.c_void_p), ("R_wk", C.c_void_p),
                ("t_wk", C.c_void_p), ("frame_ids", C.c_void_p),
                ("n_frames", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


class RaySampling(C.Structure):
    _fields_ = [("min_depth", C.c_float), ("dist_behind_surf", C.c_float), ("trunc_dist", C.c_float),
                ("n_strat", C.c_int32), ("n_surf", C.c_int32), ("rays_per_frame", C.c_int32),
                ("bin_edges", C.POINTER(C.c_float))]


class AlignPair(C.Structure):
    _fields_ = [("dst_grid", Grid), ("coords_src", C.c_void_p), ("feats_src", C.c_void_p),
                ("ld_feats", C.c_int64), ("n", C.c_int64), ("gate_coords", C.c_void_p), ("gate_n", C.c_int64),
                ("gate_axis", C.c_void_p * 3), ("gate_dims", C.c_int32 * 3),
                ("src", C.c_int32), ("dst", C.c_int32), ("src_boxes", C.c_void_p)]


class Align(C.Structure):
    _fields_ = [("n_submaps", C.c_int32), ("n_pairs", C.c_int32), ("loss_type", C.c_int32),
                ("ring_iters", C.c_int32), ("save_poses", C.c_int32), ("vec4", C.c_int32),
                ("max_n", C.c_int64), ("max_gate_n", C.c_int64), ("max_gate_rows", C.c_int64),
                ("align_weight", C.c_float), ("overlap_thresh", C.c_float),
                ("reg_weight", C.c_float), ("reg_thresh_rad", C.c_float), ("reg_thresh_m", C.c_float),
                ("rel_change_thresh", C.c_float),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("R0", C.c_void_p), ("t0", C.c_void_p), ("plan", C.c_void_p), ("state", C.c_void_p),
                ("poses_ready", C.c_int32)]


# name -> (restype, argtypes); every symbol include/miso_hip.h declares
SIGNATURES = {
    "miso_version": (C.c_char_p, []),
    "miso_error_string": (C.c_char_p, [C.c_int]),
    "miso_encode_fwd": (C.c_int, [C.POINTER(Grid), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                  C.c_void_p]),
    "miso_encode_bwd": (C.c_int, [C.POINTER(Grid), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                  C.c_void_p, C.c_void_p]),
    "miso_encode_bwd2": (C.c_int, [C.POINTER(Grid), C.POINTER(Grid), C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_void_p]),
    "miso_mlp_packed_floats": (C.c_int64, [C.POINTER(Mlp)]),
    "miso_mlp_pack": (C.c_int, [C.POINTER(Mlp), C.c_void_p, C.c_void_p]),
    "miso_sdf_supported": (C.c_int, [C.POINTER(Grid), C.POINTER(Mlp)]),
    "miso_sdf_mask_words": (C.c_int64, [C.POINTER(Mlp)]),
    "miso_sdf_fwd": (C.c_int, [C.POINTER(Grid), C.POINTER(Mlp), C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_void_p, C.c_void_p, C.c_void_p]),
    "miso_sdf_bwd": (C.c_int, [C.POINTER(Grid), C.POINTER(Mlp), C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "miso_sdf_train_lds_bytes": (C.c_int64, [C.POINTER(Grid), C.POINTER(Mlp), C.c_int32]),
    "miso_sdf_bwd_rows": (C.c_int, [C.POINTER(Grid), C.POINTER(Mlp), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "miso_grid_pool_avg": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.POINTER(C.c_float), C.c_float,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "miso_atlas_plan_bytes": (C.c_int64, [C.c_int32]),
    "miso_atlas_plan_build": (C.c_int, [C.POINTER(Grid), C.c_int32, C.c_void_p]),
    "miso_atlas_sdf_fwd": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(Grid), C.c_void_p, C.POINTER(Mlp), C.c_void_p,
                                     C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p]),
    "miso_grad_pull_on_matrix_cores": (C.c_int, [C.POINTER(Grid), C.c_int32, C.c_int64, C.c_int64]),
    "miso_sort_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32]),
    "miso_sort_points": (C.c_int, [C.POINTER(Grid), C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "miso_sdf_fwd_sorted": (C.c_int, [C.POINTER(Grid), C.POINTER(Mlp), C.c_void_p, C.POINTER(Sorted),
                                      C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "miso_grad_pull": (C.c_int, [C.POINTER(Grid), C.POINTER(Sorted), C.c_int64, C.c_void_p, C.c_int64,
                                 C.c_int32, C.c_void_p]),
    "miso_lm_normal_eq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                    C.c_float, C.c_void_p, C.c_void_p]),
    "miso_encode_fwd_sorted": (C.c_int, [C.POINTER(Grid), C.POINTER(Sorted), C.c_int64, C.c_void_p, C.c_int64,
                                         C.c_void_p]),
    "miso_sdf_fwd_loss": (C.c_int, [C.POINTER(Grid), C.POINTER(Mlp), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float,
                                    C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "miso_sdf_fwd_sorted_loss": (C.c_int, [C.POINTER(Grid), C.POINTER(Mlp), C.c_void_p, C.POINTER(Sorted), C.c_int64,
                                           C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "miso_sdf_train_sorted": (C.c_int, [C.POINTER(Grid), C.POINTER(Mlp), C.c_void_p, C.POINTER(Sorted), C.c_int64,
                                        C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "miso_sdf_train": (C.c_int, [C.POINTER(Grid), C.POINTER(Mlp), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float,
                                 C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "miso_overlap_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                     C.c_void_p, C.c_void_p]),
    "miso_grad_pull_dx": (C.c_int, [C.POINTER(Grid), C.POINTER(Sorted), C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_void_p]),
    "miso_encode_bwd2_sorted": (C.c_int, [C.POINTER(Grid), C.POINTER(Grid), C.POINTER(Sorted), C.c_int64, C.c_void_p,
                                          C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "miso_encode_bwd_sorted": (C.c_int, [C.POINTER(Grid), C.POINTER(Sorted), C.c_int64, C.c_void_p, C.c_int64,
                                         C.c_void_p, C.c_void_p]),
    "miso_grad_pull_levels": (C.c_uint32, [C.POINTER(Grid), C.c_int32]),
    "miso_track_adam_step": (C.c_int, [C.POINTER(Grid), C.POINTER(Mlp), C.c_void_p, C.POINTER(TrackAdam), C.c_void_p]),
    "miso_lm_track_step": (C.c_int, [C.POINTER(Grid), C.POINTER(Mlp), C.c_void_p, C.POINTER(LmTrack), C.c_void_p]),
    "miso_adam_scalars_table": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32,
                                          C.c_void_p]),
    "miso_adam_bump": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "miso_loss_total_bump": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "miso_adam_active_multi": (C.c_int, [C.POINTER(AdamTensor), C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
                                         C.c_int32, C.c_void_p, C.c_void_p]),
    "miso_adam_step_dev_multi": (C.c_int, [C.POINTER(AdamTensor), C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p]),
    "miso_loss_total_bump_host": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                            C.c_void_p]),
    "miso_adam_step_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_void_p, C.c_int32, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "miso_mapping_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int,
                                     C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "miso_mapping_loss_rows": (C.c_int, [C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int64,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "miso_adam_touched": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int, C.c_void_p,
                                    C.c_void_p]),
    "miso_sdf_bwd_scattered_levels": (C.c_uint32, [C.POINTER(Grid), C.c_int32, C.c_int64]),
    "miso_sdf_bwd_push_levels": (C.c_uint32, [C.POINTER(Grid), C.c_int32, C.c_int64]),
    "miso_sdf_bwd_workspace_floats": (C.c_int64, [C.POINTER(Grid), C.c_int64]),
    "miso_sdf_bwd_sorted": (C.c_int, [C.POINTER(Grid), C.POINTER(Mlp), C.c_void_p, C.POINTER(Sorted),
                                      C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "miso_pair_latent": (C.c_int, [C.POINTER(Grid), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                   C.c_int, C.c_void_p, C.c_void_p]),
    "miso_mapping_loss": (C.c_int, [C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "miso_pull_queue_ints": (C.c_int64, [C.c_int64]),
    "miso_sample_rays_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32]),
    "miso_sample_rays": (C.c_int, [C.POINTER(RayFrames), C.POINTER(RaySampling), C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "miso_adam_active": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int, C.c_void_p,
                                   C.c_void_p]),
    "miso_rigid_by_index": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int,
                                      C.c_void_p, C.c_void_p]),
    "miso_mc_words": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "miso_mc_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "miso_mc_classify": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "miso_mc_emit": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                               C.c_void_p, C.c_void_p]),
    "miso_mc_vertices": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                   C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "miso_mc_case_table": (C.c_int, [C.c_void_p]),
    "miso_align_plan_bytes": (C.c_int64, [C.c_int32]),
    "miso_align_plan_build": (C.c_int, [C.POINTER(AlignPair), C.POINTER(Align), C.c_void_p]),
    "miso_align_src_boxes": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "miso_align_state_layout": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    "miso_align_iteration_a": (C.c_int, [C.POINTER(Align), C.c_void_p]),
    "miso_align_iteration_b": (C.c_int, [C.POINTER(Align), C.c_void_p]),
    "miso_adam_dense": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                  C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int,
                                  C.c_void_p]),
}

_lib = None


def load():
    """Load libmiso_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C miso_amd/csrc`.  miso_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().miso_error_string(rc).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {rc})")
