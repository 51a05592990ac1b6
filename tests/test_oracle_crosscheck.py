"""Independent cross-checks of the parts of the oracle that cannot be pinned by running the reference (MXNet is absent):
the GridGenerator / BilinearSampler restatement vs torch's affine_grid / grid_sample (align_corners=True, zero padding --
the same published semantics, implemented by a third party), and the Deconvolution + Crop restatement vs conv_transpose2d."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import flownet as oflow
from oracle import zoom as ozoom


def test_affine_grid_and_bilinear_sampler_vs_torch():
    rng = np.random.default_rng(3)
    B, C, H, W = 3, 2, 48, 64
    data = rng.normal(size=(B, C, H, W)).astype(np.float32)
    theta = np.zeros((B, 6), np.float32)
    theta[:, 0] = rng.uniform(0.3, 1.4, B)      # wx
    theta[:, 4] = rng.uniform(0.3, 1.4, B)      # wy
    theta[:, 2] = rng.uniform(-0.6, 0.6, B)     # tx  (pushes part of the window outside: zero padding is exercised)
    theta[:, 5] = rng.uniform(-0.6, 0.6, B)
    theta[:, 1] = rng.uniform(-0.1, 0.1, B)
    theta[:, 3] = rng.uniform(-0.1, 0.1, B)
    grid = ozoom.affine_grid(theta, H, W)
    tgrid = F.affine_grid(torch.from_numpy(theta).view(B, 2, 3), (B, C, H, W), align_corners=True)   # (B,H,W,2) = (x, y)
    np.testing.assert_allclose(grid.transpose(0, 2, 3, 1), tgrid.numpy(), atol=2e-6)
    out = ozoom.bilinear_sampler(data, grid)
    tout = F.grid_sample(torch.from_numpy(data), tgrid, mode="bilinear", padding_mode="zeros", align_corners=True).numpy()
    assert (np.abs(tout) > 0).mean() > 0.5 and (tout == 0).mean() > 0.02
    np.testing.assert_allclose(out, tout, atol=1e-4)  # one ulp of a float32 grid coordinate moves a sample by a few 1e-5 here


def test_deconv_crop_vs_conv_transpose():
    """Deconvolution(k4, s2, p0) + Crop(offset 1,1) (deepIM_flownet.py:225-238) = conv_transpose2d cropped to the reference size"""
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.normal(size=(2, 6, 8, 10)).astype(np.float32))
    w = torch.from_numpy(rng.normal(size=(6, 4, 4, 4)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(4,)).astype(np.float32))
    full = F.conv_transpose2d(x, w, b, stride=2)                 # (2,4,18,22)
    got = oflow.crop_like(full, (15, 20), (1, 1))
    np.testing.assert_array_equal(got.numpy(), full[:, :, 1:16, 1:21].numpy())
    # the x16 bilinear "upsampling" kernel (mx.init.Bilinear): interpolates a constant field to the same constant (interior)
    k = oflow.bilinear_kernel((2, 1, 32, 32))
    up = F.conv_transpose2d(torch.ones(1, 2, 6, 6), torch.from_numpy(k), stride=16, groups=2)
    np.testing.assert_allclose(up[:, :, 32:64, 32:64].numpy(), 1.0, atol=1e-6)
