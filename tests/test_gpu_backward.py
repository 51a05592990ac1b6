"""GPU parity of the conv backward kernels (dgrad / wgrad / bias grad / LeakyReLU') vs torch autograd on the CPU (float64)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops(hip_lib):
    assert torch.cuda.is_available()
    from lib.hip import ops as _ops

    return _ops


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


BWD_CASES = [
    # N, H, W, Cin, Cout, k, s, p, wgrad splits
    (2, 15, 20, 64, 64, 3, 1, 1, 1),
    (2, 15, 20, 64, 128, 3, 2, 1, 2),
    (1, 30, 40, 64, 128, 5, 2, 2, 3),
    (2, 9, 11, 128, 64, 3, 2, 1, 1),
    (1, 17, 23, 64, 128, 5, 2, 2, 1),   # odd sizes: phases of different extent
    (2, 8, 10, 256, 512, 3, 1, 1, 1),
]


@pytest.mark.parametrize("case", BWD_CASES)
def test_dgrad_wgrad_bias_vs_autograd(ops, case):
    N, H, W, Cin, Cout, k, s, p, splits = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn((N, Cin, H, W), generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn((Cout, Cin, k, k), generator=g, dtype=torch.float64) / np.sqrt(Cin * k * k)).requires_grad_()
    b = torch.randn((Cout,), generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, b, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    Ho, Wo = y.shape[2:]
    # dgrad into a wider buffer with accumulate on top of an existing value
    dx = torch.full((N, H, W, ops.pad64(Cin) + 64), 0.5, device=DEV)
    wd = ops.conv2d_dgrad_pack_weight(w.detach().float().to(DEV), s, p)
    ops.conv2d_dgrad(nhwc(dy.float()), Cout, wd, dx, Cin, k, k, s, p, accumulate=True)
    got = dx[..., :Cin].permute(0, 3, 1, 2).cpu().double() - 0.5
    scale = x.grad.abs().max().item()
    assert (got - x.grad).abs().max().item() <= 2e-5 * scale + 1e-5
    assert (dx[..., ops.pad64(Cin):] == 0.5).all()
    dx2 = torch.empty((N, H, W, ops.pad64(Cin)), device=DEV)
    ops.conv2d_dgrad(nhwc(dy.float()), Cout, wd, dx2, Cin, k, k, s, p, accumulate=False)
    assert (dx2[..., :Cin].permute(0, 3, 1, 2).cpu().double() - x.grad).abs().max().item() <= 2e-5 * scale + 1e-5
    # wgrad in the packed layout == pack(dW)
    dwp = torch.zeros_like(ops.conv2d_pack_weight(w.detach().float().to(DEV)))
    ops.conv2d_wgrad(nhwc(x.detach().float()), Cin, nhwc(dy.float()), Cout, k, k, s, p, dwp, splits=splits)
    ref = ops.conv2d_pack_weight(w.grad.float().to(DEV))
    wscale = w.grad.abs().max().item()
    assert (dwp - ref).abs().max().item() <= 3e-5 * wscale + 1e-5
    db = torch.empty(Cout, device=DEV)
    ops.bias_grad(nhwc(dy.float()), Cout, db)
    np.testing.assert_allclose(db.cpu().numpy(), b.grad.float().numpy(), rtol=1e-4, atol=1e-4)


def test_wgrad_first_layer_cin8_and_fc6(ops):
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 8, 33, 41), generator=g, dtype=torch.float64)
    w = torch.randn((64, 8, 7, 7), generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, None, stride=2, padding=3)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    dwp = torch.zeros_like(ops.conv2d_pack_weight(w.detach().float().to(DEV)))
    ops.conv2d_wgrad(nhwc(x.float()), 8, nhwc(dy.float()), 64, 7, 7, 2, 3, dwp, splits=4)
    ref = ops.conv2d_pack_weight(w.grad.float().to(DEV))
    n = 13 * 32 * 64   # the packed f32 weights; behind them the buffer holds flow_conv1's three-term image (forward only)
    assert (dwp[:n] - ref[:n]).abs().max().item() <= 3e-5 * w.grad.abs().max().item() + 1e-5
    # fc6 as an 8x10 "convolution": wgrad in the fc packed layout
    B = 3
    feat = torch.randn((B, 1024, 8, 10), generator=g, dtype=torch.float64)
    w6 = torch.randn((256, 81920), generator=g, dtype=torch.float64, requires_grad=True)
    out = F.linear(feat.reshape(B, -1), w6)
    dz = torch.randn(out.shape, generator=g, dtype=torch.float64)
    out.backward(dz)
    dwp6 = torch.zeros(256 * 81920, device=DEV)
    ops.conv2d_wgrad(nhwc(feat.float()), 1024, dz.float().reshape(B, 1, 1, 256).to(DEV), 256, 8, 10, 1, 0, dwp6)
    ref6 = ops.fc_pack_weight(w6.grad.float().to(DEV), 1024, 8, 10)
    assert (dwp6 - ref6).abs().max().item() <= 3e-5 * w6.grad.abs().max().item() + 1e-5


def test_lrelu_bwd(ops):
    g = torch.Generator().manual_seed(2)
    y = torch.randn((2, 5, 7, 96), generator=g).to(DEV)
    dy = torch.randn((2, 5, 7, 128), generator=g).to(DEV)
    ref = dy.clone()
    ref[..., 32:96] *= torch.where(y[..., 16:80] > 0, 1.0, 0.1)
    ops.lrelu_bwd(y, dy, 64, slope=0.1, y_coff=16, dy_coff=32)
    np.testing.assert_allclose(dy.cpu().numpy(), ref.cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("shape", [(2, 9, 11, 770, 800, 2), (1, 15, 20, 1026, 1056, 2), (3, 8, 10, 64, 64, 1), (2, 30, 40, 6, 8, 2),
                                   (1, 13, 17, 258, 260, 1)])
def test_conv_small_cout_backward(ops, shape):
    """3x3 / pad 1 heads with 1 or 2 output channels (Convolution1-3, mask_conv3): dX, dW, db vs torch autograd in float64; channel counts
    that are not multiples of 4 inside wider zero-padded rows (the concat buffers), pixel counts around the 64-pixel chunk"""
    N, H, W, Cin, cs, Cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn((N, Cin, H, W), generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn((Cout, Cin, 3, 3), generator=g, dtype=torch.float64) / np.sqrt(9 * Cin)).requires_grad_()
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, b, padding=1)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    xb = torch.zeros((N, H, W, cs), device=DEV)
    xb[..., :Cin] = nhwc(x.detach().float())
    dx = torch.full((N, H, W, cs), 0.5, device=DEV)
    dw = torch.empty((Cout, Cin, 3, 3), device=DEV)
    db = torch.empty((Cout,), device=DEV)
    ops.conv_small_cout_bwd(xb, Cin, nhwc(dy.float()), w.detach().float().to(DEV), dx, dw, db, accumulate_dx=True)
    np.testing.assert_allclose(dw.cpu().numpy(), w.grad.numpy(), rtol=1e-4, atol=2e-5 * w.grad.abs().max().item())
    np.testing.assert_allclose(db.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dx[..., :Cin].permute(0, 3, 1, 2).cpu().numpy() - 0.5, x.grad.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("shape", [(2, 5, 7, 96, 128, 64, 16, 32), (3, 33, 41, 64, 64, 64, 0, 0), (1, 120, 160, 136, 200, 132, 4, 8)])
def test_lrelu_bwd_bias_grad_fused(ops, shape):
    """one pass = lrelu_bwd followed by bias_grad (row counts below, at and far above one row block; channel ranges of wider buffers)"""
    N, H, W, Cy, Cdy, C, yo, dyo = shape
    g = torch.Generator().manual_seed(sum(shape))
    y = torch.randn((N, H, W, Cy), generator=g).to(DEV)
    dy = torch.randn((N, H, W, Cdy), generator=g).to(DEV)
    ref = dy.clone()
    ref[..., dyo:dyo + C] *= torch.where(y[..., yo:yo + C] > 0, 1.0, 0.1)
    db_ref = ref[..., dyo:dyo + C].double().sum(dim=(0, 1, 2))
    db = torch.full((C,), 7.0, device=DEV)
    ops.lrelu_bwd_bias_grad(y, dy, C, db, slope=0.1, y_coff=yo, dy_coff=dyo)
    np.testing.assert_allclose(dy.cpu().numpy(), ref.cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(db.cpu().numpy(), db_ref.cpu().numpy(), rtol=2e-5, atol=2e-4)
    ops.lrelu_bwd_bias_grad(y, ref.clone(), C, db, slope=1.0, y_coff=yo, dy_coff=dyo, accumulate=True)   # slope 1: dz unchanged, db += sums
    np.testing.assert_allclose(db.cpu().numpy(), 2 * db_ref.cpu().numpy(), rtol=2e-5, atol=4e-4)


@pytest.mark.parametrize("shape", [(2, 64, 8, 10, 128, 15, 20), (1, 70, 15, 20, 64, 30, 40)])
def test_deconv4x4s2_backward_via_conv_view(ops, shape):
    """Deconvolution(k4,s2)+Crop(1,1) backward as used by the decoder: dgrad = stride-2 convolution of dz with the deconv weight
    read as (O=Cin, I=Cout); wgrad through the same convolution view with the roles of x and dz swapped."""
    N, Cin, H, W, Cout, OH, OW = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn((N, Cin, H, W), generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn((Cin, Cout, 4, 4), generator=g, dtype=torch.float64) / np.sqrt(Cin * 4)).requires_grad_()
    y = F.conv_transpose2d(x, w, None, stride=2)[:, :, 1:1 + OH, 1:1 + OW]
    dz = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dz)
    cpad = ops.pad64(Cin)
    xin = torch.zeros((N, H, W, cpad), device=DEV)
    xin[..., :Cin] = nhwc(x.detach().float())
    dzb = torch.zeros((N, OH, OW, Cout + 64), device=DEV)  # dz lives at channel offset 32 of a wider buffer
    dzb[..., 32:32 + Cout] = nhwc(dz.float())
    dx = torch.empty((N, H, W, cpad), device=DEV)
    wd = ops.conv2d_pack_weight_padded(w.detach().float().to(DEV), cpad)
    ops.conv2d_fwd_ex(dzb, 32, Cout, wd, None, dx, 0, cpad, 4, 4, 2, 1, Ho=H, Wo=W)
    got = dx[..., :Cin].permute(0, 3, 1, 2).cpu().double()
    assert (got - x.grad).abs().max().item() <= 2e-5 * x.grad.abs().max().item() + 1e-6
    gp = torch.empty(16 * Cout * cpad, device=DEV)
    ops.conv2d_wgrad_ex(dzb, 32, Cout, xin, 0, cpad, 4, 4, 2, 1, gp)
    dw = torch.empty((Cin, Cout, 4, 4), device=DEV)
    ops.conv2d_unpack_weight(gp, dw, CoutPad=cpad)
    assert (dw.cpu().double() - w.grad).abs().max().item() <= 3e-5 * w.grad.abs().max().item() + 1e-6


def test_winograd_dgrad_pack_weight_matches_flipped_copy(hip_lib):
    """dim_winograd_dgrad_pack_weight == dim_winograd_pack_weight of the flipped, transposed kernel (bit for bit), m = 2 and 4"""
    from lib.hip import ops

    g = torch.Generator().manual_seed(5)
    for cout, cin in ((64, 128), (256, 64), (512, 512)):
        w = torch.randn(cout, cin, 3, 3, generator=g).to("cuda:0")
        for m in (2, 4):
            want = ops.winograd_pack_weight(w.flip(2, 3).transpose(0, 1).contiguous(), m=m)
            assert torch.equal(ops.winograd_dgrad_pack_weight(w, m=m), want)
