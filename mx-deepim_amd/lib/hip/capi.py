"""ctypes binding of libdeepim_hip.so (the C ABI declared in include/deepim_hip.h).

The product path has NO CPU fallback: importing this module without the built library, or calling
an op with non-CUDA tensors, raises.  Build with `make -C mx-deepim_amd/csrc` (or
`python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIM_HIP_LIB: another build of the same library (kernel experiments: tools/split_exp.sh); there is no fallback either way
LIB_PATH = os.environ.get("DIM_HIP_LIB") or os.path.normpath(os.path.join(_HERE, "..", "libdeepim_hip.so"))

P = ctypes.c_void_p  # device pointer / host array / stream
I = ctypes.c_int
F = ctypes.c_float
L = ctypes.c_long
D = ctypes.c_double

# name -> (restype, argtypes); mirrors include/deepim_hip.h one to one
SIGNATURES = {
    "dim_last_error": (ctypes.c_char_p, []),
    "dim_device_info": (I, [ctypes.c_char_p, I]),
    "dim_mask_bbox": (I, [P, I, I, I, I, I, F, P, P, P]),
    "dim_zoom_factor": (I, [P, P, P, P, I, I, I, P, P, P]),
    "dim_zoom_planes": (I, [P, P, P, I, I, I, I, I, I, I, P, I, P]),
    "dim_zoom_net_input": (I, [P, P, P, P, P, P, I, I, I, P, P, P, P, P, P]),
    "dim_zoom_net_input_ex": (I, [P, P, P, P, P, P, I, I, I, P, I, P]),
    "dim_zoom_trans": (I, [P, P, P, I, I, P]),
    "dim_se3_compose": (I, [P, P, P, P, I, I, P, P, P]),
    "dim_se3_delta": (I, [P, P, P, P, I, I, P, P, P]),
    "dim_se3_delta_matrix": (I, [P, P, P, P, I, I, P, P, P]),
    "dim_se3_compose_euler": (I, [P, P, P, P, I, I, P, P, P]),
    "dim_se3_delta_euler": (I, [P, P, P, P, I, I, P, P, P]),
    "dim_pose_to_KT": (I, [P, P, P, P, I, P]),
    "dim_transform3d_fwd": (I, [P, P, P, P, P, I, I, I, P, P, P]),
    "dim_transform3d_bwd": (I, [P, P, P, P, P, P, P, I, I, I, P, P, P]),
    "dim_depth_to_flow": (I, [P, P, P, P, I, I, I, P, P, P]),
    "dim_flow_epe_workspace_bytes": (L, [I]),
    "dim_flow_epe_sums": (I, [P, P, P, P, I, I, I, P, P, I, P]),
    "dim_refiner_create": (I, [P, P, P, P, I, P]),
    "dim_refiner_run": (I, [P, P, P, P, P, P, P, P, P, P, P]),
    "dim_refiner_destroy": (I, [P]),
    "dim_test_blobs_from_raw": (I, [P, P, P, I, I, I, F, P, F, P, P, P, P, P]),
    "dim_pair_blobs_from_raw": (I, [P, P, P, P, P, P, P, P, P, I, I, I, F, P, F, P, P, P, P, P, P, P, P, P, P, P]),
    "dim_mask_dilate": (I, [P, P, P, I, I, I, P]),
    "dim_calc_flow_labels": (I, [P, P, P, P, I, I, I, D, I, I, P, P, P]),
    "dim_point_clouds": (I, [P, P, P, P, I, I, P, P, P, P]),
    "dim_raster_workspace_bytes": (L, [I, I, I, I]),
    "dim_raster_render": (I, [P, P, P, P, I, I, I, P, P, P, P, P, I, I, I, F, F, I, P, F, P, P, P, P, P, P, P, P]),
    "dim_raster_render_lit": (I, [P, P, P, P, P, I, I, I, P, P, P, P, P, I, I, I, F, F, I, P, P, F, P, F, P, P, P, P, P, P, P, P]),
    "dim_raster_render_dirty": (I, [P, P, P, P, P, I, I, I, P, P, P, P, P, I, I, I, F, F, I, P, P, F, P, F, P, P, P, P, P, P, P, P, P]),
    "dim_modelnet_light_position": (I, [P, F, F, F, P, I, P]),
    "dim_box_mask": (I, [P, P, I, I, I, P, P]),
    "dim_conv2d_packed_weight_floats": (L, [I, I, I, I]),
    "dim_conv2d_pack_weight": (I, [P, P, I, I, I, I, P]),
    "dim_conv2d_workspace_floats": (L, [I, I, I, I, I, I, I, I, I, I]),
    "dim_conv2d_fwd": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, I, P]),
    "dim_conv2d_fwd_partial": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dim_splitk_reduce": (I, [P, P, P, L, I, I, F, P]),
    "dim_f32_to_bf16": (I, [P, P, L, P]),
    "dim_conv2d_pack_weight_bf16": (I, [P, P, I, I, I, I, I, P]),
    "dim_conv2d_dgrad_pack_weight_bf16": (I, [P, P, I, I, I, I, I, I, P]),
    "dim_deconv4x4s2_pack_weight_bf16": (I, [P, P, I, I, P]),
    "dim_fc_dgrad_pack_weight_bf16": (I, [P, P, I, I, I, I, P]),
    "dim_bf16_to_f32": (I, [P, P, L, P]),
    "dim_conv2d_fwd_bf16": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, I, P]),
    "dim_conv2d_fwd_ex_bf16": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, F, I, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dim_conv2d_dgrad_bf16": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dim_conv2d_dgrad_splitk_workspace_floats": (L, [I, I, I, I, I]),
    "dim_conv2d_dgrad_bf16_splitk": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dim_deconv4x4s2_fwd_bf16": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, I, I, P]),
    "dim_conv2d_wgrad_bf16_splits": (I, [I, I, I, I, I, I, I, I, I, I]),
    "dim_conv2d_wgrad_bf16": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dim_conv2d_fwd_ex": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, F, I, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dim_conv2d_pack_weight_padded": (I, [P, P, I, I, I, I, I, P]),
    "dim_conv2d_unpack_weight": (I, [P, P, I, I, I, I, I, F, I, P]),
    "dim_fc_unpack_weight": (I, [P, P, I, I, I, I, P]),
    "dim_fc_dgrad_pack_weight": (I, [P, P, I, I, I, I, P]),
    "dim_flow_loss_grad": (I, [P, P, P, P, L, F, F, P, P]),
    "dim_logistic_grad": (I, [P, P, P, P, L, F, P]),
    "dim_pm_l1_grad": (I, [P, P, P, P, L, F, F, P, P]),
    "dim_pm_loss_grad": (I, [P, P, P, P, L, F, F, I, F, P, P]),
    "dim_se3_dist_loss_grad": (I, [P, P, P, P, P, P, P, P, I, F, F, I, F, P, P]),
    "dim_quat_normalize": (I, [P, P, I, P]),
    "dim_pose_head_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, I, P]),
    "dim_fc_wgrad": (I, [P, P, P, P, I, I, I, P]),
    "dim_fc_wgrad_nhwc": (I, [P, P, P, I, I, I, I, I, P]),
    "dim_upsample16_bwd": (I, [P, P, P, I, I, I, I, I, I, I, F, P]),
    "dim_conv_small_cout_bwd_workspace_floats": (L, [I, I, I, I, I, I, I]),
    "dim_conv_small_cout_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dim_deconv4x4s2_tiny_bwd": (I, [P, I, P, I, I, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "dim_conv2d_tail_plan": (I, [I, I, I, I, I, I, P, P]),
    "dim_conv_auto_plan": (I, [L, I, I, I, P, P]),
    "dim_winograd_gemm_tile": (I, [I, L]),
    "dim_winograd_gemm_tile_planes": (I, [I, L, I]),
    "dim_set_winograd_split": (I, [I]),
    "dim_get_winograd_split": (I, []),
    "dim_winograd_packed_weight_floats": (L, [I, I, I]),
    "dim_winograd_workspace_floats": (L, [I, I, I, I, I, I]),
    "dim_winograd_pack_weight": (I, [P, P, I, I, I, P]),
    "dim_winograd_dgrad_pack_weight": (I, [P, P, I, I, I, P]),
    "dim_winograd3x3s2_packed_weight_floats": (L, [I, I]),
    "dim_winograd3x3s2_use": (I, [I, I, I, I]),
    "dim_winograd3x3s2_workspace_floats": (L, [I, I, I, I, I]),
    "dim_winograd3x3s2_pack_weight": (I, [P, P, I, I, P]),
    "dim_conv2d_fwd_winograd3x3s2": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, F, I, P, P]),
    "dim_winograd5x5s2_packed_weight_floats": (L, [I, I]),
    "dim_winograd5x5s2_workspace_floats": (L, [I, I, I, I, I]),
    "dim_winograd5x5s2_pack_weight": (I, [P, P, I, I, P]),
    "dim_conv2d_fwd_winograd5x5s2": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, F, I, P, P]),
    "dim_winograd5x5s2_dgrad_pack_weight": (I, [P, P, I, I, P]),
    "dim_conv2d_dgrad_winograd5x5s2": (I, [P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "dim_conv2d_wgrad_winograd_workspace_floats": (L, [I, I, I, I, I, I, I]),
    "dim_conv2d_wgrad_winograd": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, P]),
    "dim_fc_fwd_workspace_floats": (L, [I, I, I, I]),
    "dim_fc_fwd": (I, [P, P, P, P, P, I, I, I, I, I, F, P]),
    "dim_conv2d_fwd_winograd": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, F, I, I, P, P]),
    "dim_copy_words": (I, [P, P, L, P]),
    "dim_copy_rows": (I, [P, L, P, L, L, L, P]),
    "dim_add_rows": (I, [P, L, P, L, L, L, P]),
    "dim_fill_words": (I, [P, L, ctypes.c_uint, P]),
    "dim_sgd_momentum": (I, [P, P, P, L, F, F, F, F, P]),
    "dim_adam": (I, [P, P, P, P, L, F, F, F, F, F, F, P]),
    "dim_deconv4x4s2_packed_weight_floats": (L, [I, I]),
    "dim_deconv4x4s2_pack_weight": (I, [P, P, I, I, P]),
    "dim_deconv4x4s2_fwd": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, I, I, P]),
    "dim_deconv4x4s2_tiny_fwd": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dim_conv_small_cout_pack_weight": (I, [P, P, I, I, I, I, P]),
    "dim_conv_small_cout_fwd": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dim_upsample16_fwd": (I, [P, P, P, I, I, I, I, I, I, I, F, I, P]),
    "dim_conv2d_dgrad_packed_weight_floats": (L, [I, I, I, I, I, I]),
    "dim_conv2d_dgrad_pack_weight": (I, [P, P, I, I, I, I, I, I, P]),
    "dim_conv2d_dgrad": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dim_conv2d_wgrad_workspace_floats": (L, [I, I, I, I, I]),
    "dim_conv2d_wgrad": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dim_conv2d_dgrad_lrelu_workspace_floats": (L, [I, I, I, I, I]),
    "dim_conv2d_dgrad_bf16_lrelu": (I, [P, P, P, P, F, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dim_conv2d_wgrad_oihw": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, F, I, P]),
    "dim_bias_grad_workspace_floats": (L, [I, I]),
    "dim_bias_grad": (I, [P, P, P, I, I, I, I, I, P]),
    "dim_lrelu_bwd": (I, [P, I, I, P, I, I, L, I, F, P]),
    "dim_lrelu_bwd_bias_grad_workspace_floats": (L, [I, I]),
    "dim_lrelu_bwd_bias_grad": (I, [P, I, I, P, I, I, P, P, I, I, F, I, P]),
    "dim_fc_pack_weight": (I, [P, P, I, I, I, I, P]),
    "dim_pose_head_fwd": (I, [P, P, P, P, P, P, P, P, P, P, I, P]),
}

_lib = None


class DeepIMHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raise loudly if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DeepIMHipError(
                "libdeepim_hip.so not found at {} -- build it with `make -C mx-deepim_amd/csrc`; "
                "there is no CPU fallback for the refinement path".format(LIB_PATH))
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise DeepIMHipError("libdeepim_hip error {}: {}".format(rc, lib().dim_last_error().decode()))


def host_f32(values, n=None):
    """small HOST float array argument (K9, means3, ...) -> (keepalive ndarray, pointer)."""
    a = np.ascontiguousarray(np.asarray(values, dtype=np.float32).reshape(-1))
    if n is not None and a.size != n:
        raise ValueError("expected {} floats, got {}".format(n, a.size))
    return a, a.ctypes.data


def dptr(t, dtype=None):
    """device pointer of a contiguous CUDA torch tensor (None -> NULL)."""
    if t is None:
        return None
    import torch

    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise DeepIMHipError("expected a CUDA tensor (the HIP path has no CPU fallback), got {}".format(type(t)))
    if not t.is_contiguous():
        raise DeepIMHipError("tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise DeepIMHipError("expected dtype {}, got {}".format(dtype, t.dtype))
    return t.data_ptr()


def current_stream():
    import torch

    return torch.cuda.current_stream().cuda_stream


ROT_COORD = {"model": 0, "camera": 1, "camera_new": 2, "naive": 3}


def rot_coord_id(name):
    try:
        return ROT_COORD[name.lower()]
    except KeyError:
        raise Exception("Unknown rot_coord: {}".format(name))
