"""ctypes front-end of oracle/raster.c (software rasteriser + depth->flow restatement).

Oracle = test infrastructure.  `build()` compiles the C file with gcc (no GPU needed);
the resulting oracle/_build/liboracle.so is git-ignored but travels with gpurun snapshots.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "raster.c")
_OUT = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    os.makedirs(os.path.dirname(_OUT), exist_ok=True)
    if force or not os.path.exists(_OUT) or os.path.getmtime(_OUT) < os.path.getmtime(_SRC):
        # -ffp-contract=off: only the explicit fmaf() calls fuse, like the HIP kernels' __fmaf_rn
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", _OUT, _SRC, "-lm"])
    return _OUT


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def render(verts, uvs, faces, tex, R, t, K, H=480, W=640, znear=0.25, zfar=6.0, tex_bilinear=False):
    """Render_Py.render restatement -> (bgr (H,W,3) float32 0..255, depth (H,W) float32 metres)."""
    verts = np.ascontiguousarray(verts, dtype=np.float32)
    uvs = np.ascontiguousarray(uvs, dtype=np.float32)
    faces = np.ascontiguousarray(faces, dtype=np.int32)
    tex = np.ascontiguousarray(tex, dtype=np.uint8)
    R = np.ascontiguousarray(R, dtype=np.float32).reshape(9)
    t = np.ascontiguousarray(t, dtype=np.float32).reshape(3)
    K = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
    bgr = np.empty((H, W, 3), dtype=np.float32)
    depth = np.empty((H, W), dtype=np.float32)
    f = ctypes.c_float
    lib().dim_oracle_render(
        _p(verts, f), _p(uvs, f), _p(faces, ctypes.c_int32), ctypes.c_int(verts.shape[0]), ctypes.c_int(faces.shape[0]),
        _p(tex, ctypes.c_uint8), ctypes.c_int(tex.shape[0]), ctypes.c_int(tex.shape[1]),
        _p(R, f), _p(t, f), _p(K, f), ctypes.c_int(H), ctypes.c_int(W), f(znear), f(zfar),
        ctypes.c_int(1 if tex_bilinear else 0), _p(bgr, f), _p(depth, f))
    return bgr, depth


def render_lit(verts, normals, uvs, faces, tex, R, t, K, light_position, light_intensity, brightness_ratio=0.7, H=480, W=640,
               znear=0.25, zfar=6.0, tex_bilinear=False):
    """Render_Py_Light_ModelNet_Multi.render restatement (render_py_light_modelnet_multi.py:153-235) ->
    (bgr (H,W,3) float32 0..255 integral values, depth (H,W) float32 metres)."""
    verts = np.ascontiguousarray(verts, dtype=np.float32)
    normals = np.ascontiguousarray(normals, dtype=np.float32)
    uvs = np.ascontiguousarray(uvs, dtype=np.float32)
    faces = np.ascontiguousarray(faces, dtype=np.int32)
    tex = np.ascontiguousarray(tex, dtype=np.uint8)
    R = np.ascontiguousarray(R, dtype=np.float32).reshape(9)
    t = np.ascontiguousarray(t, dtype=np.float32).reshape(3)
    K = np.ascontiguousarray(K, dtype=np.float32).reshape(9)
    lp = np.ascontiguousarray(light_position, dtype=np.float32).reshape(3)
    li = np.ascontiguousarray(light_intensity, dtype=np.float32).reshape(3)
    bgr = np.empty((H, W, 3), dtype=np.float32)
    depth = np.empty((H, W), dtype=np.float32)
    f = ctypes.c_float
    lib().dim_oracle_render_lit(
        _p(verts, f), _p(normals, f), _p(uvs, f), _p(faces, ctypes.c_int32), ctypes.c_int(verts.shape[0]),
        ctypes.c_int(faces.shape[0]), _p(tex, ctypes.c_uint8), ctypes.c_int(tex.shape[0]), ctypes.c_int(tex.shape[1]),
        _p(R, f), _p(t, f), _p(K, f), ctypes.c_int(H), ctypes.c_int(W), f(znear), f(zfar),
        ctypes.c_int(1 if tex_bilinear else 0), _p(lp, f), _p(li, f), f(brightness_ratio), _p(bgr, f), _p(depth, f))
    return bgr, depth


def modelnet_light_position(pose, idx=2):
    """tester.py:204-225 / batch_updater_py_multi.py:233-255: light direction table entry `idx % 6`, halved, offset by the
    pose translation with y and z flipped (GL camera frame)."""
    table = [[1, 0, 1], [1, 1, 1], [0, 1, 1], [-1, 1, 1], [-1, 0, 1], [0, 0, 1]]
    lp = np.array(table[idx % 6], dtype=np.float64) * 0.5
    lp[0] += pose[0, 3]
    lp[1] -= pose[1, 3]
    lp[2] -= pose[2, 3]
    return lp


def gpu_flow(depth_src, depth_tgt, KT, Kinv):
    """lib/flow_c gpu_flow(depth_src[N,1,H,W], depth_tgt[N,1,H,W], KT[N,3,4], Kinv[3,3]) -> flow[N,2,H,W], valid[N,1,H,W]."""
    depth_src = np.ascontiguousarray(depth_src, dtype=np.float32)
    depth_tgt = np.ascontiguousarray(depth_tgt, dtype=np.float32)
    KT = np.ascontiguousarray(KT, dtype=np.float32)
    Kinv = np.ascontiguousarray(Kinv, dtype=np.float32)
    B, _, H, W = depth_src.shape
    flow = np.zeros((B, 2, H, W), dtype=np.float32)
    valid = np.zeros((B, 1, H, W), dtype=np.float32)
    f = ctypes.c_float
    if B > 0:
        lib().dim_oracle_flow(_p(depth_src, f), _p(depth_tgt, f), _p(KT, f), _p(Kinv, f), ctypes.c_int(B), ctypes.c_int(H),
                              ctypes.c_int(W), _p(flow, f), _p(valid, f))
    return flow, valid
