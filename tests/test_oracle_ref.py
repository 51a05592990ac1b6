"""oracle/_ref -- the reference's own compiled CPU flow kernel (lib/flow_c/cpu_flow_kernel.cpp, built by oracle/ref_build.py from the
sources where they lie) -- against the oracle's restatement of the CUDA kernel (oracle/raster.c dim_oracle_flow,
lib/flow_c/gpu_flow_kernel.cu:32-69).  The two kernels test visibility differently, but share the projection arithmetic: on every
pixel both call visible, the flow must agree BIT FOR BIT.  Runs wherever the prebuilt library is (here and on the GPU box)."""
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import native, ref_build, se3 as ose3

GOLD = os.path.join(ROOT, "tests", "golden")


def test_reference_cpu_flow_kernel_pins_the_projection_arithmetic():
    if ref_build.build() is None:
        pytest.skip("oracle/_ref not built and /root/reference not mounted")
    g = np.load(os.path.join(GOLD, "flow_golden.npz"))
    K = g["K"].astype(np.float32)
    Kinv = np.linalg.inv(K).astype(np.float32)
    n_both = 0
    for i in range(len(g["depth_src"])):
        d_src, d_tgt = g["depth_src"][i].astype(np.float32), g["depth_tgt"][i].astype(np.float32)
        R, t = ose3.calc_se3(g["pose_src"][i], g["pose_tgt"][i])
        KT = np.dot(K, np.concatenate([R, t.reshape(3, 1)], axis=1)).astype(np.float32)
        ref = ref_build.flow_cpp(d_src, d_tgt, KT, Kinv)
        assert not np.isnan(ref).any()                                  # every pixel written
        mine, valid = native.gpu_flow(d_src[None, None], d_tgt[None, None], KT[None], Kinv)
        ref_vis = (d_src > 1e-3) & ((d_src - d_tgt) < 3e-3)             # cpu_flow_kernel.cpp:28
        np.testing.assert_array_equal(ref[:, ~ref_vis], 0.0)
        both = ref_vis & (valid[0, 0] > 0)
        n_both += int(both.sum())
        np.testing.assert_array_equal(mine[0][:, both], ref[:, both])   # same float32 operations in the same order
    assert n_both > 1000
