"""ctypes binding of libprcnn_pointops.so -- exactly the entry points declared in include/prcnn_pointops.h.

This is the reference-side binding a maintainer would add: raw device pointers + sizes +
a hipStream_t, no torch types cross the boundary.  The product path FAILS LOUDLY when the library is
missing: there is no CPU or PyTorch fallback.
"""
import ctypes
import os

from . import build as _build

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_L = ctypes.c_int64
_Z = ctypes.c_size_t
_D = ctypes.c_double



class TrainSrc(ctypes.Structure):
    """prcnn_train_src_t (include/prcnn_pointops.h)"""
    _fields_ = [("mode", _I), ("rows", _L), ("K", _I),
                ("in_", _P), ("ld_in", _I), ("pro_scale", _P), ("pro_shift", _P),
                ("xyz", _P), ("new_xyz", _P), ("idx", _P), ("feat", _P), ("ld_feat", _I),
                ("B", _I), ("N", _I), ("M", _I), ("ns", _I), ("C", _I),
                ("known", _P), ("idx3", _P), ("w3", _P), ("skip", _P), ("ld_known", _I), ("ld_skip", _I),
                ("n", _I), ("m", _I), ("C2", _I), ("C1", _I),
                ("mult", _P), ("rows_dev", _P), ("norm_rows", _L), ("seg_off", _P), ("seg_cnt", _P), ("row_grp", _P), ("groups", _I)]


class TrainLayer(ctypes.Structure):
    """prcnn_train_layer_t (include/prcnn_pointops.h)"""
    _fields_ = [("Nout", _I), ("W", _P), ("gamma", _P), ("beta", _P), ("eps", _F), ("momentum", _F),
                ("running_mean", _P), ("running_var", _P), ("y", _P), ("cst", _P), ("ld_c", _I), ("wpack", _P), ("wpack_t", _P),
                ("dW", _P), ("dgamma", _P), ("dbeta", _P)]


REQUIRED_ABI = 9                 # prcnn_abi_version() the signatures below describe

# name -> (restype, argtypes); mirrors include/prcnn_pointops.h one for one
SIGNATURES = {
    "prcnn_abi_version": (_I, []),
    "prcnn_last_error": (ctypes.c_char_p, []),
    "prcnn_build_id": (ctypes.c_char_p, []),
    "prcnn_fps": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "prcnn_fps_status": (_I, []),
    "prcnn_fps_order": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "prcnn_fps_mode": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "prcnn_ball_query_arith": (_I, [_P, _P, _I, _I, _I, _F, _I, _I, _P, _P]),
    "prcnn_three_nn_arith": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "prcnn_rpn_labels": (_I, [_P, _P, _P, _I, _I, _I, _F, _P, _P, _P]),
    "prcnn_gt_aug_edit": (_I, [_P, _P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "prcnn_host_pts_in_boxes3d": (_I, [_P, _P, _L, _L, _P]),
    "prcnn_host_roipool3d": (_I, [_P, _P, _P, _L, _L, _L, _L, _P, _P, _P]),
    "prcnn_gather": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "prcnn_gather_grad": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "prcnn_gather_rows": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P]),
    "prcnn_ball_query": (_I, [_P, _P, _I, _I, _I, _F, _I, _P, _P]),
    "prcnn_ball_query2": (_I, [_P, _P, _I, _I, _I, _F, _I, _P, _F, _I, _P, _P]),
    "prcnn_group": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "prcnn_group_grad": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "prcnn_three_nn": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "prcnn_three_interp": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "prcnn_three_interp_grad": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "prcnn_wpack_floats": (_Z, [_I, _I]),
    "prcnn_pack_weight": (_I, [_P, _I, _I, _I, _P, _P]),
    "prcnn_mlp_rows": (_I, [_P, _I, _L, _I, _P, _P, _I, _I, _P, _I, _I, _I, _P, _I, _P, _I, _P]),
    "prcnn_mlp_group": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _P, _P]),
    "prcnn_mlp_group_split": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _I, _P, _I, _I, _I, _P, _P]),
    "prcnn_mlp_interp": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P, _I, _I, _P]),
    "prcnn_mlp_rows_addinterp": (_I, [_P, _I, _I, _P, _P, _I, _I, _P, _I, _P, _P, _I, _I, _I, _P, _I, _I, _P]),
    "prcnn_wsplit_bytes": (_Z, [_I, _I]),
    "prcnn_pack_weight_split": (_I, [_P, _I, _I, _I, _P, _P]),
    "prcnn_mlp_chain_interp_split": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _P, _I, _I, _P]),
    "prcnn_mlp_chain_rows_split": (_I, [_P, _I, _L, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P]),
    "prcnn_mlp_rows_split": (_I, [_P, _I, _L, _I, _P, _P, _I, _P, _I, _I, _P, _I, _I, _I, _P, _I, _P, _I, _P]),
    "prcnn_mlp_rows_addinterp_split": (_I, [_P, _I, _I, _P, _P, _I, _P, _I, _I, _P, _I, _P, _P, _I, _I, _I, _P, _I, _I, _P]),
    "prcnn_mlp_chain_supported": (_I, [_I, _I, _P, _I]),
    "prcnn_mlp_chain_rows": (_I, [_P, _I, _L, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P]),
    "prcnn_mlp_chain_group": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "prcnn_mlp_chain_interp": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P]),
    "prcnn_maxpool_rows": (_I, [_P, _I, _L, _I, _I, _P, _I, _I, _P]),
    "prcnn_roipool3d": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "prcnn_roipool3d_work_bytes": (_Z, [_I, _I]),
    "prcnn_roipool3d_ws": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "prcnn_pts_in_boxes3d": (_I, [_P, _P, _I, _I, _P, _P]),
    "prcnn_boxes_overlap_bev": (_I, [_P, _I, _P, _I, _P, _P]),
    "prcnn_boxes_iou_bev": (_I, [_P, _I, _P, _I, _P, _P]),
    "prcnn_nms_workspace_bytes": (_Z, [_I]),
    "prcnn_nms": (_I, [_P, _I, _F, _I, _I, _P, _P, _P, _Z, _P]),
    "prcnn_decode_bbox_target": (_I, [_P, _I, _P, ctypes.c_long, _I, _D, _D, _I, _P, _I, _I, _D, _D, _I, _I, _P, _P]),
    "prcnn_proposal_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "prcnn_proposal_layer": (_I, [_P, _P, _I, _I, _I, _F, _F, _F, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P, _Z, _P]),
    "prcnn_nms_batched_workspace_bytes": (_Z, [_I, _I]),
    "prcnn_roipool3d_canonical": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P]),
    "prcnn_roipool3d_canonical_ws": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _Z, _P]),
    "prcnn_grid_bytes": (_Z, [_I, _I]),
    "prcnn_grid_build": (_I, [_P, _I, _I, _F, _I, _P, _Z, _P]),
    "prcnn_ball_query2_grid": (_I, [_P, _P, _P, _I, _I, _I, _F, _I, _P, _F, _I, _P, _P]),
    "prcnn_three_nn_grid": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "prcnn_rotate_iou_eval": (_I, [_P, _I, _P, _I, _I, _P, _P]),
    "prcnn_kitti_overlaps": (_I, [_I, _P, _P, _P, _P, _P, _I, _P, _P]),
    "prcnn_kitti_statistics": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _D, _P, _I, _I, _I, _P, _P, _P]),
    "prcnn_group_compact": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "prcnn_segmax_scatter": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _P, _I, _I, _P]),
    "prcnn_scatter_rows": (_I, [_P, _I, _P, _P, _I, _I, _P, _I, _I, _P]),
    "prcnn_scene_workspace_bytes": (_Z, [_L, _I]),
    "prcnn_scene_prepare": (_I, [_P, _P, _I, _L, _I, _P, _P, _P, _I, ctypes.c_uint32, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "prcnn_nms_batched": (_I, [_P, _P, _P, _I, _I, _F, _I, _I, _P, _P, _P, _Z, _P]),
    "prcnn_train_stack_work_bytes": (_Z, [_L, ctypes.POINTER(TrainLayer), _I, _I, _I]),
    "prcnn_train_stack_fwd": (_I, [ctypes.POINTER(TrainSrc), ctypes.POINTER(TrainLayer), _I, _I, _P, _I, _P, _I, _I, _P, _P, _Z, _P]),
    "prcnn_train_stack_bwd": (_I, [ctypes.POINTER(TrainSrc), ctypes.POINTER(TrainLayer), _I, _I, _P, _I, _P, _I, _P, _P, _I, _P, _Z, _P]),
    "prcnn_ref_trig": (_I, [_P, _P, _I, _I, _P, _P]),
    "prcnn_boxes_iou3d": (_I, [_P, _I, _P, _I, _P, _P]),
    "prcnn_proposal_target_sample": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _I, _I, ctypes.c_uint32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "prcnn_train_group_rows": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "prcnn_flat_rows_grad": (_I, [_P, _I, _P, _P, _L, _I, _P, _I, _P]),
    "prcnn_flat_rows_grad_work_bytes": (_Z, [_I, _I, _L]),
    "prcnn_flat_rows_grad_ws": (_I, [_P, _I, _P, _P, _L, _I, _P, _I, _P, _I, _I, _I, _P, _Z, _P]),
    "prcnn_group_rows_grad": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "prcnn_interp_rows_grad": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "prcnn_interp_rows_grad_work_bytes": (_Z, [_I, _I, _I]),
    "prcnn_interp_rows_grad_ws": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _I, _P, _Z, _P]),
}

_lib = None


class PointOpsError(RuntimeError):
    pass


def library_path():
    return _build.LIB


def lib():
    """Load (building first if hipcc is available and the .so is missing/stale) and type the library."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process BEFORE this library is
    # loaded, so that the library's DT_NEEDED libamdhip64 resolves to the same runtime instance; loaded the other way
    # round the process ends up with two runtimes and the second one reports "no ROCm-capable device".
    import torch  # noqa: F401
    path = _build.LIB
    if not os.path.exists(path):
        try:
            _build.build(verbose=False)
        except Exception as e:  # noqa: BLE001
            raise PointOpsError(
                "libprcnn_pointops.so is missing and could not be built (%s). The HIP extension is "
                "mandatory: there is no CPU fallback. Run `python -m pointrcnn_amd.build`." % e)
    elif not os.environ.get("PRCNN_POINTOPS_LIB") and _build.have_sources() and _build.library_id(path) != _build.source_id():
        # A library built from OTHER kernel sources than the ones next to this file would load fine (same symbols) and
        # silently compute with old kernels.  The digest baked into the .so (prcnn_build_id) is compared with the sources:
        # rebuild when a compiler is here (serialised by a lock file: under torchrun every rank gets here at once),
        # refuse otherwise.
        if not os.path.exists(_build.HIPCC):
            raise PointOpsError("%s is stale (built from %s, sources are %s) and hipcc is not available to rebuild it"
                                % (path, _build.library_id(path), _build.source_id()))
        try:
            _build.build(verbose=False)
        except Exception as e:  # noqa: BLE001
            raise PointOpsError("%s is stale and could not be rebuilt (%s)" % (path, e))
    try:
        handle = ctypes.CDLL(path)
    except OSError as e:
        raise PointOpsError("cannot load %s: %s" % (path, e))
    handle.prcnn_abi_version.restype = _I
    if handle.prcnn_abi_version() < REQUIRED_ABI:      # an older library fails HERE, not at a missing symbol or a changed signature later
        raise PointOpsError("%s implements ABI %d, this binding needs %d (include/prcnn_pointops.h)"
                            % (path, handle.prcnn_abi_version(), REQUIRED_ABI))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)      # AttributeError here == header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return handle


EUNSUPPORTED = -3                # PRCNN_EUNSUPPORTED: a valid request this build has no kernel for


def check(rc, what=""):
    if rc != 0:
        msg = lib().prcnn_last_error()
        raise PointOpsError("%s failed (code %d): %s" % (what or "prcnn call", rc, msg.decode() if msg else "?"))
