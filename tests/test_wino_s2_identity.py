"""The algebra behind csrc/wino_s2.hip on the CPU: a 3x3 / stride-2 / pad-1 convolution as four phase images with minimal filtering
(F(4,1) on the even, F(4,2) on the odd phase, 81 planes per 4 x 4 output tile) equals the direct convolution exactly in float64, for
the integer-scaled matrices the kernels use and on maps that do not divide into tiles (tools/wino_s2_proto.py is the numpy statement)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import wino_s2_proto as proto  # noqa: E402

# the matrices of csrc/wino_s2.hip (B^T scaled to small integers, the fractions in G)
BT = np.array([[2, -1, -2, 1, 0], [0, 2, 1, -1, 0], [0, -2, 3, -1, 0], [0, -1, 0, 1, 0], [0, 2, -1, -2, 1]], float)
G = np.array([[0.5, 0], [0.5, 0.5], [1 / 6, -1 / 6], [1 / 6, 1 / 3], [0, 1]])
AT = np.array([[1, 1, 1, 1, 0], [0, 1, -1, 2, 0], [0, 1, 1, 4, 0], [0, 1, -1, 8, 1]], float)


def test_f42_identity_of_the_kernel_matrices():
    rng = np.random.RandomState(0)
    for _ in range(20):
        d, g = rng.randn(5), rng.randn(2)
        y = AT @ ((G @ g) * (BT @ d))
        ref = np.array([g[0] * d[k] + g[1] * d[k + 1] for k in range(4)])
        assert np.abs(y - ref).max() < 1e-13


def test_phase_image_form_equals_direct_convolution():
    rng = np.random.RandomState(1)
    for H, W, C, Co in ((8, 8, 3, 2), (9, 11, 4, 3), (15, 20, 5, 2), (7, 5, 2, 4)):
        x = rng.randn(H, W, C)
        w = rng.randn(Co, C, 3, 3)
        ref = proto.conv_direct(x, w, np.float64)
        got = proto.conv_wino_s2(x, w, (AT, G, BT), np.float64)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
    # and the Cook-Toom construction of the tool gives the same algebra for another point set
    got = proto.conv_wino_s2(x, w, proto.f42((0, 1, -1, -2)), np.float64)
    assert np.abs(got - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
