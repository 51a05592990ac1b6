"""oracle/_ref: the one piece of the reference's own NATIVE code on the path that compiles without CUDA / MXNet / OpenGL, built from
the sources where they lie (never copied): /root/reference/lib/flow_c/cpu_flow_kernel.cpp (+ cpu_flow.hpp) with
`g++ -O2 -ffp-contract=off -shared -fPIC` -> oracle/_ref/libcpu_flow_ref.so (git-ignored, travels to the GPU box with the snapshot).

Test infrastructure only.  What it pins: the projection arithmetic of the depth->flow path (A15).  `flow_cpp` is the older CPU
variant of the CUDA kernel: the same float32 back-projection x = (w Kinv0 + h Kinv1 + Kinv2) d, the same K T product and
division, the same (dy, dx) output order -- but another visibility predicate (d_src > 1e-3 and d_src - d_tgt[SAME pixel] < 3e-3, no
bounds test, no `valid` output) and a `KT += batch_idx * 12` inside the pixel loop that only leaves the pointer alone for batch 0.
So it is run with ONE pair per call, and compared with the oracle's restatement of the CUDA kernel on the pixels where BOTH
predicates hold: there the two must agree bit for bit (tests/test_oracle_ref.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(os.environ.get("DIM_REFERENCE_ROOT", "/root/reference"), "lib", "flow_c")   # where the reference is mounted
OUT = os.path.join(_HERE, "_ref", "libcpu_flow_ref.so")
SYMBOL = "_Z8flow_cppPfS_S_S_S_iiii"   # void flow_cpp(float*, float*, float*, float*, float*, int, int, int, int): C++ linkage in the reference


def build(force=False):
    """-> path of the built library, or None when the reference is not mounted (the GPU box: it uses the prebuilt file)"""
    src = os.path.join(REF_DIR, "cpu_flow_kernel.cpp")
    if not os.path.exists(src):
        return OUT if os.path.exists(OUT) else None
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-I", REF_DIR, "-o", OUT, src])
    return OUT


def flow_cpp(depth_src, depth_tgt, KT, Kinv):
    """the reference's function for ONE pair: depth_src / depth_tgt (H,W) float32, KT (3,4), Kinv (3,3) -> flow (2,H,W) in (dy, dx)"""
    path = build()
    if path is None:
        raise RuntimeError("oracle/_ref is not built and /root/reference is not mounted")
    fn = getattr(ctypes.CDLL(path), SYMBOL)
    fn.restype = None
    f = ctypes.POINTER(ctypes.c_float)
    depth_src = np.ascontiguousarray(depth_src, dtype=np.float32)
    depth_tgt = np.ascontiguousarray(depth_tgt, dtype=np.float32)
    KT = np.ascontiguousarray(KT, dtype=np.float32)
    Kinv = np.ascontiguousarray(Kinv, dtype=np.float32)
    H, W = depth_src.shape
    flow = np.full((2, H, W), np.nan, dtype=np.float32)
    fn(flow.ctypes.data_as(f), depth_src.ctypes.data_as(f), depth_tgt.ctypes.data_as(f), KT.ctypes.data_as(f), Kinv.ctypes.data_as(f),
       ctypes.c_int(1), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(0))
    return flow
