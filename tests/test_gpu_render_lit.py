"""GPU parity of the lit (ModelNet) rasteriser variant vs the C software rasteriser of the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lib.utils import synthetic as syn  # noqa: E402
from oracle import native  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("tex_bilinear", [False, True])
def test_lit_render_vs_oracle(hip_lib, tex_bilinear):
    from lib.render_hip.render_py_light_modelnet_multi import Render_Py_Light_ModelNet_Multi, vertex_normals

    models = syn.make_models(seed=11, n_models=3, subdiv=3)
    rng = np.random.default_rng(5)
    gray = np.full((64, 64, 3), 200, np.uint8)
    gray[::2, ::3] = 140  # a little structure so that texture addressing matters
    meshes = [(v, vertex_normals(v, f).astype(np.float32), t, f) for v, t, f, _ in models]
    K = syn.LINEMOD_K
    rm = Render_Py_Light_ModelNet_Multi(None, gray, K, 640, 480, 0.25, 6.0, brightness_ratios=[0.7], meshes=meshes,
                                        tex_bilinear=tex_bilinear)
    cls, gt, _ = syn.sample_pairs(77, 4, n_classes=3)
    B = 4
    poses = torch.from_numpy(gt.astype(np.float32)).to(DEV)
    light_int = rng.uniform(0.9, 1.3, size=(B, 3)).astype(np.float32)  # > 1 exercises the [0,1] clamp
    lp_dev = rm.light_position(poses, idx=2)
    for b in range(B):
        want = native.modelnet_light_position(gt[b].astype(np.float32).astype(np.float64), idx=2)
        np.testing.assert_allclose(lp_dev[b].cpu().numpy(), want.astype(np.float32), rtol=0, atol=1e-7)
    bgr = torch.empty((B, 480, 640, 3), dtype=torch.float32, device=DEV)
    depth = torch.empty((B, 1, 480, 640), dtype=torch.float32, device=DEV)
    image = torch.empty((B, 3, 480, 640), dtype=torch.float32, device=DEV)
    mask = torch.empty((B, 1, 480, 640), dtype=torch.float32, device=DEV)
    pm = np.array([123.68, 116.779, 103.939], np.float32)
    rm.render_batch(torch.from_numpy(cls.astype(np.int32)).to(DEV), poses, lp_dev, torch.from_numpy(light_int).to(DEV), bgr=bgr,
                    depth=depth, image=image, mask=mask, plane_means=pm)
    got_bgr, got_d = bgr.cpu().numpy(), depth.cpu().numpy()[:, 0]
    for b in range(B):
        v, n, t, f = meshes[cls[b]]
        w_bgr, w_d = native.render_lit(v, n, t, f, gray, gt[b][:, :3], gt[b][:, 3], K, lp_dev[b].cpu().numpy(), light_int[b], 0.7,
                                       tex_bilinear=tex_bilinear)
        cov = (got_d[b] > 0) != (w_d > 0)
        assert cov.sum() <= 4, cov.sum()
        both = (got_d[b] > 0) & (w_d > 0)
        assert both.sum() > 500
        np.testing.assert_allclose(got_d[b][both], w_d[both], rtol=2e-6)
        diff = np.abs(got_bgr[b][both] - w_bgr[both])
        assert diff.max() <= 1.0, diff.max()  # integral grey levels; a rounding tie may fall the other way
        assert (diff > 0).mean() < 1e-3, (diff > 0).mean()
        assert w_bgr[both].std() > 3.0  # shading varies over the object
        assert got_bgr[b][both].max() <= 255.0
        # image blob = RGB planes minus plane means; mask = depth > 0.2
        np.testing.assert_array_equal(image[b, 0].cpu().numpy(), got_bgr[b][..., 2] - pm[0])
        np.testing.assert_array_equal(mask[b, 0].cpu().numpy(), (got_d[b] > 0.2).astype(np.float32))
    # reference-signature single render returns uint8 bgr + depth
    img, d = rm.render(int(cls[0]), gt[0][:, :3], gt[0][:, 3], lp_dev[0].cpu().numpy(), light_int[0], brightness_k=0, r_type="mat")
    assert img.dtype == np.uint8 and img.shape == (480, 640, 3)
    np.testing.assert_array_equal(img.astype(np.float32), got_bgr[0])
    np.testing.assert_array_equal(d, got_d[0])
