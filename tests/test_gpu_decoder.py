"""GPU parity of the decoder + flow / mask heads (FAST_TEST False graph) vs the torch-CPU oracle, piece by piece and end to end."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import flownet as oflow  # noqa: E402
from scene import make_scene, make_test_config  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops(hip_lib):
    assert torch.cuda.is_available()
    from lib.hip import ops as _ops

    return _ops


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


@pytest.mark.parametrize("shape", [(2, 64, 8, 10, 64, 15, 20), (1, 70, 15, 20, 128, 30, 40), (2, 32, 5, 7, 64, 11, 15)])
def test_deconv4x4s2_crop_lrelu(ops, shape):
    N, Cin, H, W, Cout, OH, OW = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cin, Cout, 4, 4), generator=g) / np.sqrt(Cin * 4)
    b = torch.randn((Cout,), generator=g)
    ref = F.leaky_relu(F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2)[:, :, 1:1 + OH, 1:1 + OW], 0.1).float()
    cs = ops.pad32(Cin)
    xin = torch.zeros((N, H, W, cs), device=DEV)
    xin[..., :Cin] = nhwc(x)
    y = torch.full((N, OH, OW, Cout + 40), -7.0, device=DEV)  # concat buffer: deconv lands at channel offset 8
    ops.deconv4x4s2_fwd(xin, Cin, ops.deconv4x4s2_pack_weight(w.to(DEV)), b.to(DEV), y, Cout, crop=1, slope=0.1, out_coff=8)
    got = y[..., 8:8 + Cout].permute(0, 3, 1, 2).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-5, rtol=1e-4)
    assert (y[..., :8] == -7).all() and (y[..., 8 + Cout:] == -7).all()  # neighbours of the channel range untouched


def test_deconv_tiny_and_small_cout_and_upsample16(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 2, 8, 10), generator=g)
    w = torch.randn((2, 2, 4, 4), generator=g)
    b = torch.randn((2,), generator=g)
    ref = F.conv_transpose2d(x, w, b, stride=2)[:, :, 1:16, 1:21]
    y = torch.zeros((2, 15, 20, 6), device=DEV)
    ops.deconv4x4s2_tiny_fwd(nhwc(x), 2, w.to(DEV), b.to(DEV), y, 2, crop=1, out_coff=3)
    np.testing.assert_allclose(y[..., 3:5].permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), atol=1e-5)
    # small-Cout conv on a zero-padded concat buffer (770 -> 800 channels), and the other two heads' shapes (rows of 20 and 10 pixels:
    # a wave owns 4 adjacent output pixels, the last group of a 10-pixel row is partial); a 7-pixel row; a 1x1 kernel
    for cout, cin, cpad, h, w_, k in ((1, 770, 800, 30, 40, 3), (2, 770, 800, 30, 40, 3), (2, 1026, 1056, 15, 20, 3), (2, 1024, 1024, 8, 10, 3),
                                      (1, 96, 96, 5, 7, 3), (2, 64, 64, 6, 9, 1)):
        xc = torch.randn((2, cin, h, w_), generator=g)
        wc = torch.randn((cout, cin, k, k), generator=g) / np.sqrt(cin * k * k)
        bc = torch.randn((cout,), generator=g)
        refc = F.conv2d(xc.double(), wc.double(), bc.double(), padding=k // 2).float()
        buf = torch.zeros((2, h, w_, cpad), device=DEV)
        buf[..., :cin] = nhwc(xc)
        out = ops.conv_small_cout_fwd(buf, cin, ops.conv_small_cout_pack_weight(wc.to(DEV)), bc.to(DEV), cout, KH=k, KW=k, pad=k // 2)
        np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), refc.numpy(), atol=2e-5, rtol=1e-4)
    # x16 frozen-bilinear deconvolution + Crop(8,8), grouped (flow) and sigmoid (mask)
    f = torch.randn((2, 2, 30, 40), generator=g)
    wk = torch.from_numpy(oflow.bilinear_kernel((2, 1, 32, 32)))
    ref_up = F.conv_transpose2d(f, wk, None, stride=16, groups=2)[:, :, 8:488, 8:648] * 20.0
    got = ops.upsample16_fwd(nhwc(f), wk.to(DEV), 480, 640, crop=8, scale=20.0)     # four output pixels per thread
    np.testing.assert_allclose(got.cpu().numpy(), ref_up.numpy(), atol=2e-5, rtol=1e-5)
    ref_odd = F.conv_transpose2d(f, wk, None, stride=16, groups=2)[:, :, 6:477, 6:637] * 20.0   # crop 6, 471 x 631: one pixel per thread
    got_odd = ops.upsample16_fwd(nhwc(f), wk.to(DEV), 471, 631, crop=6, scale=20.0)
    np.testing.assert_allclose(got_odd.cpu().numpy(), ref_odd.numpy(), atol=2e-5, rtol=1e-5)
    assert torch.equal(got[:, :, :400, :600], ops.upsample16_fwd(nhwc(f), wk.to(DEV), 400, 600, crop=8, scale=20.0))
    m = torch.randn((2, 1, 30, 40), generator=g)
    wk1 = torch.from_numpy(oflow.bilinear_kernel((1, 1, 32, 32)))
    ref_m = torch.sigmoid(F.conv_transpose2d(m, wk1, None, stride=16)[:, :, 8:488, 8:648])
    got_m = ops.upsample16_fwd(nhwc(m), wk1.to(DEV), 480, 640, crop=8, sigmoid=True)
    np.testing.assert_allclose(got_m.cpu().numpy(), ref_m.numpy(), atol=1e-6)


def test_full_graph_not_fast_test_vs_oracle(hip_lib):
    from deepim.core.tester import Predictor
    from deepim.symbols.deepIM_flownet import deepIM_flownet

    cfg = make_test_config(test_iter=1)
    cfg.TEST.FAST_TEST = False
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=False)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    rng = np.random.RandomState(1)
    params["trans_weight"] = (rng.randn(3, 256) * 0.002).astype(np.float32)
    params["mask_conv3_weight"] = (rng.randn(1, 770, 3, 3) * 0.05).astype(np.float32)  # make the mask head cross 0.5 / 0.2
    params["mask_conv3_bias"] = np.array([0.1], np.float32)
    scene = make_scene(B=2, seed=77, subdiv=3)
    pred = Predictor(cfg, params, 2)
    batch = {k: torch.as_tensor(v).to(DEV) for k, v in scene["blobs"].items()}
    out = pred.predict(batch)[0]
    ref = oflow.forward_test(params, scene["blobs"], scene["K"], cfg.network.PIXEL_MEANS, fast_test=False)
    net = pred.net
    c3 = net.concat3[..., :770].permute(0, 3, 1, 2).cpu().numpy()
    want = ref["concat3"].numpy()
    assert np.abs(c3 - want).max() <= 1e-4 * np.abs(want).max() + 1e-5
    assert net.concat3[..., 770:].abs().sum() == 0 and net.concat2[..., 1026:].abs().sum() == 0
    np.testing.assert_allclose(out["se3_output"].cpu().numpy(), ref["se3"], atol=1e-3)
    np.testing.assert_allclose(out["zoom_mask_observed_prob_iter_output"].cpu().numpy(), ref["zoom_mask_prob"], atol=1e-4)
    prob = ref["zoom_mask_prob"]
    assert 0.05 < (prob > 0.2).mean() < 0.95  # the binarisation branch is exercised
    mism = (out["mask_observed_pred_output"].cpu().numpy() != ref["mask_observed_pred"]).sum()
    assert mism <= 200, mism  # pixels whose probability sits within float noise of 0.2, or factor last-bit shifts
    fl, rfl = out["flow_est_crop_output"].cpu().numpy(), ref["flow_est_crop"]
    np.testing.assert_allclose(fl, rfl, atol=1e-3 * max(1.0, np.abs(rfl).max()))


def test_upsample16_backward_vs_autograd(ops):
    """backward of the frozen x16 bilinear deconvolution + Crop(8, 8) (flow: 2 grouped channels, mask: 1) vs torch autograd in float64,
    with a random (not bilinear) kernel so that every tap matters; edge windows hang over the crop on all four sides"""
    g = torch.Generator().manual_seed(21)
    for C in (2, 1):
        wk = torch.randn((C, 1, 32, 32), generator=g)
        f = torch.zeros((3, C, 30, 40), dtype=torch.float64, requires_grad=True)
        dout = torch.randn((3, C, 480, 640), generator=g)
        up = F.conv_transpose2d(f, wk.double(), None, stride=16, groups=C)[:, :, 8:488, 8:648] * 20.0
        up.backward(dout.double())
        df = torch.full((3, 30, 40, C), 5.0, device=DEV)
        ops.upsample16_bwd(dout.to(DEV), wk.to(DEV), df, crop=8, scale=20.0)
        ref = f.grad.permute(0, 2, 3, 1)
        assert (df.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-5
