"""Independent cross-checks of the parts of the oracle that cannot be pinned by running the reference (MXNet is absent):
the GridGenerator / BilinearSampler restatement vs torch's affine_grid / grid_sample (align_corners=True, zero padding --
the same published semantics, implemented by a third party), and the Deconvolution + Crop restatement vs conv_transpose2d."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import flownet as oflow
from oracle import native, se3 as ose3
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


def _raycast_depth(verts, faces, R, t, K, H, W, znear, zfar, step=4):
    """independent of the rasteriser: Moeller-Trumbore per pixel ray (every `step`-th pixel) against every triangle in camera space;
    depth = smallest Z in [znear, zfar] among the hits (what a z-buffered GL draw with near / far clipping and no face culling shows)"""
    cam = (verts.astype(np.float64) @ np.asarray(R, np.float64).T) + np.asarray(t, np.float64)
    v0, v1, v2 = cam[faces[:, 0]], cam[faces[:, 1]], cam[faces[:, 2]]
    e1, e2 = v1 - v0, v2 - v0
    ys, xs = np.mgrid[0:H:step, 0:W:step]
    d = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs, dtype=np.float64)], -1).reshape(-1, 3)   # Z = 1 rays
    out = np.zeros(d.shape[0])
    margin = np.zeros(d.shape[0])   # how far the winning hit is from any triangle edge (barycentric units): pixels on an edge are skipped
    for i, di in enumerate(d):
        p = np.cross(di, e2)
        det = (e1 * p).sum(1)
        ok = np.abs(det) > 1e-14
        inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
        tv = -v0
        u = (tv * p).sum(1) * inv
        q = np.cross(tv, e1)
        v = (q @ di) * inv
        z = (e2 * q).sum(1) * inv   # ray parameter = camera Z because d_z = 1
        hit = ok & (u >= 0) & (v >= 0) & (u + v <= 1) & (z >= znear) & (z <= zfar)
        if hit.any():
            k = np.flatnonzero(hit)[np.argmin(z[hit])]
            out[i] = z[k]
            margin[i] = min(u[k], v[k], 1 - u[k] - v[k])
    return out.reshape(ys.shape), margin.reshape(ys.shape)


@pytest.mark.parametrize("tz", [0.30, 0.12])
def test_near_plane_clipping_vs_ray_casting(tz):
    """GL clips against zNear = 0.25 (render_py_multi.py:152-169).  A 0.3 m mesh centred at Z = 0.30 straddles the plane; centred at
    Z = 0.12 it also has vertices BEHIND the eye (Z < 0), whose screen-space triangles do not exist.  The rasteriser's depth must
    be what ray casting against the same triangles gives: nearest surface at Z >= 0.25, the inside of the far wall where the front
    is cut away -- never a fragment in front of the plane, never a hole where a clipped triangle's remainder is visible."""
    from lib.utils import synthetic as syn

    rng = np.random.default_rng(5)
    v, uv, f = syn.make_mesh(rng, subdiv=2, diameter=0.3)
    tex = syn.make_texture(rng, size=64, cells=8)
    K = syn.LINEMOD_K
    q = rng.normal(size=4)
    R = ose3.quat2mat(q / np.linalg.norm(q))
    t = np.array([0.01, -0.02, tz])
    zc = (v @ R.T + t)[:, 2]
    assert zc.min() < 0.25 < zc.max() and (tz > 0.2 or zc.min() < 0.0)
    bgr, depth = native.render(v, uv, f, tex, R, t, K, znear=0.25, zfar=6.0)
    assert depth[depth > 0].min() >= 0.25
    want, margin = _raycast_depth(v, f, R.astype(np.float32).astype(np.float64), t.astype(np.float32).astype(np.float64), K, 480, 640, 0.25, 6.0, step=4)
    got = depth[::4, ::4]
    # compare away from triangle edges / silhouettes (sub-pixel snapping moves an edge by up to 1/256 px; ray casting does not snap)
    inner = margin > 0.02
    both = inner & (want > 0)
    assert both.sum() > 1000
    assert ((got > 0) == (want > 0))[inner | (want == 0)].mean() > 0.995
    sel = both & (got > 0)
    np.testing.assert_allclose(got[sel], want[sel], rtol=2e-4)
    assert (bgr[depth > 0].sum(-1) > 0).all()   # every covered pixel got a texel (texture has no black)
