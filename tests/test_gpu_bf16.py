"""bf16 matrix-pipe twins of the convolution kernels (training mode of BASELINE configs[2]; the reference trains in fp32, so this is new
functionality with a DECLARED tolerance).

Two bars per kernel:
  exact   -- vs torch-CPU float64 on operands rounded to bf16 first: bf16 x bf16 products are exact in f32, so the kernel must agree to
             f32 accumulation error (1e-4 relative): this pins indexing, operand maps and the rounding mode (nearest even);
  declared -- vs the unrounded float64 result: per-tensor L2-relative error <= 6e-3 (2^-8 per operand, two operands, random signs).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
L2_BAR = 6e-3


@pytest.fixture(scope="module")
def ops(hip_lib):
    assert torch.cuda.is_available()
    from lib.hip import ops as _ops

    return _ops


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def r16(t):
    """round to bf16 (nearest even) and back to float64"""
    return t.float().bfloat16().double()


def l2rel(a, b):
    return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30))


def test_f32_bf16_round_trip(ops):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(100003, generator=g) * 3).to(DEV)
    h = ops.to_bf16(x)
    assert h.dtype == torch.bfloat16 and torch.equal(h, x.bfloat16())
    assert torch.equal(ops.from_bf16(h), x.bfloat16().float())


FWD_CASES = [
    # N, H, W, Cin, Cout, k, s, p, tile, splits
    (2, 60, 80, 8, 64, 7, 2, 3, 3, 1),
    (1, 37, 53, 8, 64, 7, 2, 3, 2, 1),
    # tile 6: the persistent first-layer kernel on the bf16 pipe (16 x 16 output blocks, two taps per MFMA; whole blocks / partial blocks
    # on both edges / a map smaller than one block / more blocks than CUs, so that workgroups walk several blocks through both buffers)
    (2, 64, 96, 8, 64, 7, 2, 3, 6, 1),
    (1, 37, 53, 8, 64, 7, 2, 3, 6, 1),
    (3, 9, 13, 8, 64, 7, 2, 3, 6, 1),
    (5, 290, 420, 8, 64, 7, 2, 3, 6, 1),
    (2, 30, 40, 64, 128, 5, 2, 2, 4, 1),
    (1, 23, 31, 64, 128, 5, 2, 2, 3, 1),
    (2, 15, 20, 256, 256, 3, 1, 1, 4, 1),
    (2, 15, 20, 256, 512, 3, 2, 1, 1, 1),
    (2, 15, 20, 512, 512, 3, 1, 1, 3, 3),
    (2, 8, 10, 512, 1024, 3, 2, 1, 4, 4),
    (1, 9, 11, 32, 64, 3, 1, 0, 2, 1),
    # tile 7: the LDS-halo kernel (8 x 16 output blocks x 128 channels; partial blocks, several channel slices, both strides)
    (2, 30, 40, 64, 128, 5, 2, 2, 7, 1),
    (1, 23, 31, 64, 128, 5, 2, 2, 7, 1),
    (2, 15, 20, 256, 256, 3, 1, 1, 7, 1),
    (2, 15, 20, 256, 512, 3, 2, 1, 7, 1),
    (1, 37, 53, 32, 128, 3, 1, 1, 7, 1),
    (3, 9, 11, 96, 128, 3, 2, 1, 7, 1),
    # tile 9: the stride-1 patch kernel (16 x 16 output blocks x 128 channels; partial blocks, 1 .. 8 channel slices, no padding)
    (2, 15, 20, 256, 256, 3, 1, 1, 9, 1),
    (1, 37, 53, 32, 128, 3, 1, 1, 9, 1),
    (2, 33, 17, 64, 128, 3, 1, 1, 9, 1),
    (1, 18, 35, 96, 256, 3, 1, 0, 9, 1),
    (1, 16, 16, 32, 128, 3, 1, 1, 9, 1),
    (2, 20, 33, 64, 64, 3, 1, 1, 9, 1),    # 64-channel tile
    (1, 17, 16, 32, 192, 3, 1, 1, 9, 1),   # 192 = 3 x 64
    # tile 9 at stride 2 (8 x 16 output blocks, even / odd input columns in two patch halves)
    (2, 30, 40, 64, 128, 5, 2, 2, 9, 1),
    (1, 23, 31, 64, 128, 5, 2, 2, 9, 1),
    (1, 37, 53, 32, 128, 5, 2, 2, 9, 1),
    (2, 15, 20, 256, 512, 3, 2, 1, 9, 1),
    (3, 9, 11, 96, 128, 3, 2, 1, 9, 1),
    (1, 32, 32, 32, 256, 3, 2, 1, 9, 1),
]


@pytest.mark.parametrize("case", FWD_CASES)
def test_conv_fwd_bf16(ops, case):
    N, H, W, Cin, Cout, k, s, p, tile, splits = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn((Cout,), generator=g) * 0.1
    ref_exact = F.leaky_relu(F.conv2d(r16(x), r16(w), b.double(), stride=s, padding=p), 0.1).permute(0, 2, 3, 1).numpy()
    ref_full = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p), 0.1).permute(0, 2, 3, 1).numpy()
    wp = ops.to_bf16(ops.conv2d_pack_weight(w.to(DEV)))
    y = ops.conv2d_fwd(nhwc(x), wp, b.to(DEV), Cout, k, k, s, p, slope=0.1, splits=splits, tile=tile).cpu().numpy()
    assert np.abs(y - ref_exact).max() <= 1e-4 * np.abs(ref_exact).max() + 2e-5
    assert l2rel(y, ref_full) <= L2_BAR


BWD_CASES = [
    # N, H, W, Cin, Cout, k, s, p, wgrad splits
    (1, 30, 40, 128, 128, 5, 2, 2, 3),   # stride-2 input gradient = 3x3 / 3x2 / 2x3 / 2x2 phase convolutions (tile 9 below)
    (2, 21, 17, 128, 64, 3, 2, 1, 1),    # 2x2 / 2x1 / 1x2 / 1x1 phases
    (2, 15, 20, 64, 64, 3, 1, 1, 1),
    (2, 15, 20, 64, 128, 3, 2, 1, 2),
    (1, 30, 40, 64, 128, 5, 2, 2, 3),
    (2, 9, 11, 128, 64, 3, 2, 1, 1),
    (1, 17, 23, 64, 128, 5, 2, 2, 1),
    (2, 8, 10, 256, 512, 3, 1, 1, 1),
    (1, 21, 35, 128, 256, 3, 1, 1, 2),
    (3, 13, 9, 32, 64, 3, 1, 1, 2),    # one chunk column group only partly filled (9 chunks, 4 per workgroup)
    # the patch form of the weight gradient (3x3 / stride 1, >= 1200 pixels, Cout % 128 == 0): 8 x 8 pixel blocks that hang over both map
    # edges, several input slices / output tiles, block ranges split unevenly, and a single split
    (2, 36, 44, 64, 128, 3, 1, 1, 7),
    (1, 40, 48, 96, 256, 3, 1, 1, 1),
    (3, 30, 40, 32, 128, 3, 1, 1, 100),   # more splits than blocks per split can fill: clamped
]


@pytest.mark.parametrize("case", BWD_CASES)
def test_dgrad_wgrad_bf16(ops, case):
    N, H, W, Cin, Cout, k, s, p, splits = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn((N, Cin, H, W), generator=g, dtype=torch.float64)
    w = torch.randn((Cout, Cin, k, k), generator=g, dtype=torch.float64) / np.sqrt(Cin * k * k)
    xr, wr = r16(x).requires_grad_(), r16(w).requires_grad_()
    y = F.conv2d(xr, wr, None, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(r16(dy))                      # exact bar: every operand of every product is a bf16 value
    xf, wf = x.clone().requires_grad_(), w.clone().requires_grad_()
    F.conv2d(xf, wf, None, stride=s, padding=p).backward(dy)   # declared bar: the unrounded gradients
    wd = ops.to_bf16(ops.conv2d_dgrad_pack_weight(w.float().to(DEV), s, p))
    dx = torch.empty((N, H, W, ops.pad64(Cin)), device=DEV)
    ops.conv2d_dgrad(nhwc(dy.float()), Cout, wd, dx, Cin, k, k, s, p, accumulate=False)
    got = dx[..., :Cin].permute(0, 3, 1, 2).cpu().double().numpy()
    assert np.abs(got - xr.grad.numpy()).max() <= 1e-4 * xr.grad.abs().max().item() + 1e-5
    assert l2rel(got, xf.grad.numpy()) <= L2_BAR
    if s == 1 and ops.pad64(Cin) % 128 == 0:   # stride-1 input gradients may take the LDS-halo kernel too
        dx7 = torch.empty((N, H, W, ops.pad64(Cin)), device=DEV)
        ops.conv2d_dgrad(nhwc(dy.float()), Cout, wd, dx7, Cin, k, k, s, p, accumulate=False, tile=7)
        got7 = dx7[..., :Cin].permute(0, 3, 1, 2).cpu().double().numpy()
        assert np.abs(got7 - xr.grad.numpy()).max() <= 1e-4 * xr.grad.abs().max().item() + 1e-5
    if True:   # ... and the stride-1 patch kernel (128- or 64-channel tiles), which also takes the phases of a stride-2 gradient
        dx9 = torch.full((N, H, W, ops.pad64(Cin)), 3.0, device=DEV)
        ops.conv2d_dgrad(nhwc(dy.float()), Cout, wd, dx9, Cin, k, k, s, p, accumulate=False, tile=9)
        got9 = dx9[..., :Cin].permute(0, 3, 1, 2).cpu().double().numpy()
        assert np.abs(got9 - xr.grad.numpy()).max() <= 1e-4 * xr.grad.abs().max().item() + 1e-5
        ops.conv2d_dgrad(nhwc(dy.float()), Cout, wd, dx9, Cin, k, k, s, p, accumulate=True, tile=9)   # out += result
        got9 = dx9[..., :Cin].permute(0, 3, 1, 2).cpu().double().numpy()
        assert np.abs(got9 - 2 * xr.grad.numpy()).max() <= 2e-4 * xr.grad.abs().max().item() + 2e-5
    # split-K through output-shaped slabs (the small maps): same result as the single launch up to f32 summation order; every phase of a
    # strided gradient needs >= splits K chunks (a 1-tap phase of a 32-wide dy has one)
    ksp = 2 if s == 1 or Cout >= 64 else 1
    if ksp > 1:
        dxs = torch.full((N, H, W, ops.pad64(Cin)), 5.0, device=DEV)
        ops.conv2d_dgrad(nhwc(dy.float()), Cout, wd, dxs, Cin, k, k, s, p, accumulate=False, tile=3, splits=ksp)
        gots = dxs[..., :Cin].permute(0, 3, 1, 2).cpu().double().numpy()
        assert np.abs(gots - xr.grad.numpy()).max() <= 1e-4 * xr.grad.abs().max().item() + 1e-5
        ops.conv2d_dgrad(nhwc(dy.float()), Cout, wd, dxs, Cin, k, k, s, p, accumulate=True, tile=3, splits=ksp)   # dx += result
        gots = dxs[..., :Cin].permute(0, 3, 1, 2).cpu().double().numpy()
        assert np.abs(gots - 2 * xr.grad.numpy()).max() <= 2e-4 * xr.grad.abs().max().item() + 2e-5
    dwp = torch.zeros_like(ops.conv2d_pack_weight(w.float().to(DEV)))
    ops.conv2d_wgrad(nhwc(x.float()), Cin, nhwc(dy.float()), Cout, k, k, s, p, dwp, splits=splits, bf16_mfma=True)
    ref = ops.conv2d_pack_weight(wr.grad.float().to(DEV))
    n = 13 * 32 * 64   # the packed f32 weights (behind them: flow_conv1's three-term image, forward only)
    assert (dwp[:n] - ref[:n]).abs().max().item() <= 1e-4 * wr.grad.abs().max().item() + 1e-5
    assert l2rel(dwp.cpu().numpy(), ops.conv2d_pack_weight(wf.grad.float().to(DEV)).cpu().numpy()) <= L2_BAR
    # the same gradient delivered in the MXNet layout with the slab sum folded into the layout converter: the bits of the two-step path
    two = ops.conv2d_unpack_weight(dwp, torch.empty((Cout, Cin, k, k), device=DEV))
    one = torch.full((Cout, Cin, k, k), 7.0, device=DEV)
    ops.conv2d_wgrad_oihw(nhwc(x.float()), Cin, nhwc(dy.float()), Cout, k, k, s, p, one, splits=splits, bf16_mfma=True)
    assert torch.equal(one, two)
    ops.conv2d_wgrad_oihw(nhwc(x.float()), Cin, nhwc(dy.float()), Cout, k, k, s, p, one, splits=splits, bf16_mfma=True, scale=0.5, accumulate=True)
    assert torch.equal(one, two + 0.5 * two)
    f32_two = torch.zeros_like(dwp)
    ops.conv2d_wgrad(nhwc(x.float()), Cin, nhwc(dy.float()), Cout, k, k, s, p, f32_two, splits=splits)
    f32_one = ops.conv2d_wgrad_oihw(nhwc(x.float()), Cin, nhwc(dy.float()), Cout, k, k, s, p, torch.empty_like(one), splits=splits)
    assert torch.equal(f32_one, ops.conv2d_unpack_weight(f32_two, torch.empty_like(one)))


@pytest.mark.parametrize("case", [
    # N, H, W, Cin (of dX / the activation below), Cout (of dy), k, s, p
    (2, 60, 80, 64, 128, 5, 2, 2),      # the four phase images of a 5x5 / stride-2 gradient, 64-channel tile
    (1, 47, 61, 128, 256, 5, 2, 2),     # odd map: phases of different sizes, partial 16 x 16 blocks
    (2, 30, 40, 256, 256, 3, 1, 1),     # 3x3 / stride 1
    (1, 37, 53, 128, 128, 3, 1, 1),
])
def test_dgrad_with_lrelu_and_bias_gradient_folded_in(ops, case):
    """dim_conv2d_dgrad_bf16_lrelu == dim_conv2d_dgrad_bf16 (tile 9) followed by dim_lrelu_bwd_bias_grad: dz bit for bit, db up to the
    summation order; and both against float64 on the bf16-rounded operands"""
    N, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    w = torch.randn((Cout, Cin, k, k), generator=g, dtype=torch.float64) / np.sqrt(Cout * k * k)
    dy = torch.randn((N, Cout, Ho, Wo), generator=g, dtype=torch.float64)
    y_act = torch.randn((N, Cin, H, W), generator=g)
    y_act[0, :, :2, :3] = 0.0                                  # exact zeros take the slope branch (y > 0 ? 1 : slope)
    wd = ops.to_bf16(ops.conv2d_dgrad_pack_weight(w.float().to(DEV), s, p))
    two = torch.empty((N, H, W, Cin), device=DEV)
    ops.conv2d_dgrad(nhwc(dy.float()), Cout, wd, two, Cin, k, k, s, p, accumulate=False, tile=9)
    db_two = torch.zeros(Cin, device=DEV)
    ops.lrelu_bwd_bias_grad(nhwc(y_act), two, Cin, db_two)
    one = torch.full((N, H, W, Cin), 9.0, device=DEV)
    db_one = torch.full((Cin,), 5.0, device=DEV)
    ops.conv2d_dgrad_lrelu(nhwc(dy.float()), Cout, wd, one, nhwc(y_act), Cin, k, k, s, p, db_one)
    assert torch.equal(one, two)
    scale = db_two.abs().max().item()
    assert (db_one - db_two).abs().max().item() <= 2e-5 * scale + 1e-6
    ops.conv2d_dgrad_lrelu(nhwc(dy.float()), Cout, wd, one, nhwc(y_act), Cin, k, k, s, p, db_one, accumulate_db=True)
    assert (db_one - 2 * db_two).abs().max().item() <= 4e-5 * scale + 2e-6
    dx_ref = torch.nn.grad.conv2d_input((N, Cin, H, W), r16(w), r16(dy), stride=s, padding=p)
    dz_ref = dx_ref * torch.where(y_act.double() > 0, 1.0, 0.1)
    got = one.permute(0, 3, 1, 2).cpu().double()
    assert (got - dz_ref).abs().max().item() <= 1e-4 * dz_ref.abs().max().item() + 1e-5
    assert (db_two.cpu().double() - dz_ref.sum((0, 2, 3))).abs().max().item() <= 1e-4 * dz_ref.sum((0, 2, 3)).abs().max().item() + 1e-3


def test_wgrad_bf16_first_layer_cin8_and_fc6(ops):
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 8, 33, 41), generator=g, dtype=torch.float64)
    w = torch.randn((64, 8, 7, 7), generator=g, dtype=torch.float64)
    xr, wr = r16(x), r16(w).requires_grad_()
    y = F.conv2d(xr, wr, None, stride=2, padding=3)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(r16(dy))
    dwp = torch.zeros_like(ops.conv2d_pack_weight(w.float().to(DEV)))
    ops.conv2d_wgrad(nhwc(x.float()), 8, nhwc(dy.float()), 64, 7, 7, 2, 3, dwp, splits=4, bf16_mfma=True)
    ref = ops.conv2d_pack_weight(wr.grad.float().to(DEV))
    n = 13 * 32 * 64   # the packed f32 weights (behind them: flow_conv1's three-term image, forward only)
    assert (dwp[:n] - ref[:n]).abs().max().item() <= 1e-4 * wr.grad.abs().max().item() + 1e-5
    # the MXNet-layout entry: the 8-lane layer has no tiled converter (slabs are summed first, then the element-wise converter); 85 small
    # slabs take the lane-parallel reduce first, exactly as the two-step path does
    xb = torch.randn((2, 8, 65, 81), generator=g, dtype=torch.float64)
    dyb = torch.randn((2, 64, 33, 41), generator=g, dtype=torch.float64)
    for xi, dyi, sp in ((x, dy, 4), (xb, dyb, 1000)):
        ops.conv2d_wgrad(nhwc(xi.float()), 8, nhwc(dyi.float()), 64, 7, 7, 2, 3, dwp, splits=sp, bf16_mfma=True)
        two = ops.conv2d_unpack_weight(dwp, torch.empty((64, 8, 7, 7), device=DEV))
        one = ops.conv2d_wgrad_oihw(nhwc(xi.float()), 8, nhwc(dyi.float()), 64, 7, 7, 2, 3, torch.empty_like(two), splits=sp, bf16_mfma=True)
        assert torch.equal(one, two)
    refb = torch.nn.grad.conv2d_weight(r16(xb), (64, 8, 7, 7), r16(dyb), stride=2, padding=3)
    assert (two.cpu().double() - refb).abs().max().item() <= 1e-4 * refb.abs().max().item() + 1e-5
    B = 3
    feat = torch.randn((B, 1024, 8, 10), generator=g, dtype=torch.float64)
    w6 = torch.randn((256, 81920), generator=g, dtype=torch.float64, requires_grad=True)
    out = F.linear(r16(feat).reshape(B, -1), w6)
    dz = torch.randn(out.shape, generator=g, dtype=torch.float64)
    out.backward(r16(dz))
    dwp6 = torch.zeros(256 * 81920, device=DEV)
    ops.conv2d_wgrad(nhwc(feat.float()), 1024, dz.float().reshape(B, 1, 1, 256).to(DEV), 256, 8, 10, 1, 0, dwp6, bf16_mfma=True)
    ref6 = ops.fc_pack_weight(w6.grad.float().to(DEV), 1024, 8, 10)
    assert (dwp6 - ref6).abs().max().item() <= 1e-4 * w6.grad.abs().max().item() + 1e-5


@pytest.mark.parametrize("shape", [(2, 64, 8, 10, 64, 15, 20), (1, 70, 15, 20, 128, 30, 40)])
def test_deconv4x4s2_bf16_forward_and_backward(ops, shape):
    N, Cin, H, W, Cout, OH, OW = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn((N, Cin, H, W), generator=g, dtype=torch.float64)
    w = torch.randn((Cin, Cout, 4, 4), generator=g, dtype=torch.float64) / np.sqrt(Cin * 4)
    b = torch.randn((Cout,), generator=g, dtype=torch.float64)
    xr, wr = r16(x).requires_grad_(), r16(w).requires_grad_()
    pre = F.conv_transpose2d(xr, wr, b, stride=2)[:, :, 1:1 + OH, 1:1 + OW]
    ref = F.leaky_relu(pre, 0.1)
    cs = ops.pad32(Cin)
    xin = torch.zeros((N, H, W, cs), device=DEV)
    xin[..., :Cin] = nhwc(x.float())
    y = torch.full((N, OH, OW, Cout + 40), -7.0, device=DEV)
    ops.deconv4x4s2_fwd(xin, Cin, ops.to_bf16(ops.deconv4x4s2_pack_weight(w.float().to(DEV))), b.float().to(DEV), y, Cout, crop=1, slope=0.1,
                        out_coff=8)
    got = y[..., 8:8 + Cout].permute(0, 3, 1, 2).cpu().double()
    assert (got - ref.detach()).abs().max().item() <= 1e-4 * ref.abs().max().item() + 2e-5
    assert (y[..., :8] == -7).all() and (y[..., 8 + Cout:] == -7).all()
    if Cout % 128 == 0:   # the four 2x2 phase convolutions as one batched launch of the stride-1 patch kernel
        y9 = torch.full((N, OH, OW, Cout + 40), -7.0, device=DEV)
        ops.deconv4x4s2_fwd(xin, Cin, ops.to_bf16(ops.deconv4x4s2_pack_weight(w.float().to(DEV))), b.float().to(DEV), y9, Cout, crop=1,
                            slope=0.1, out_coff=8, tile=9)
        got9 = y9[..., 8:8 + Cout].permute(0, 3, 1, 2).cpu().double()
        assert (got9 - ref.detach()).abs().max().item() <= 1e-4 * ref.abs().max().item() + 2e-5
        assert (y9[..., :8] == -7).all() and (y9[..., 8 + Cout:] == -7).all()
    # backward through the convolution view (decoder): dgrad = stride-2 conv of dz with the weight read as (O = Cin, I = Cout)
    dz = torch.randn(pre.shape, generator=g, dtype=torch.float64)
    pre.backward(r16(dz))
    cpad = ops.pad64(Cin)
    dzb = torch.zeros((N, OH, OW, Cout + 64), device=DEV)
    dzb[..., 32:32 + Cout] = nhwc(dz.float())
    dx = torch.empty((N, H, W, cpad), device=DEV)
    wd = ops.to_bf16(ops.conv2d_pack_weight_padded(w.float().to(DEV), cpad))
    ops.conv2d_fwd_ex(dzb, 32, Cout, wd, None, dx, 0, cpad, 4, 4, 2, 1, Ho=H, Wo=W)
    gotx = dx[..., :Cin].permute(0, 3, 1, 2).cpu().double()
    assert (gotx - xr.grad).abs().max().item() <= 1e-4 * xr.grad.abs().max().item() + 1e-6
    xin2 = torch.zeros((N, H, W, cpad), device=DEV)
    xin2[..., :Cin] = nhwc(x.float())
    gp = torch.empty(16 * Cout * cpad, device=DEV)
    ops.conv2d_wgrad_ex(dzb, 32, Cout, xin2, 0, cpad, 4, 4, 2, 1, gp, bf16_mfma=True)
    dw = torch.empty((Cin, Cout, 4, 4), device=DEV)
    ops.conv2d_unpack_weight(gp, dw, CoutPad=cpad)
    assert (dw.cpu().double() - wr.grad).abs().max().item() <= 1e-4 * wr.grad.abs().max().item() + 1e-6


def _pack_index_reference(w, cout_pad):
    """[chunk = (32-channel slice, kh, kw)][CoutPad][32] restated with torch indexing (dim_conv2d_pack_weight's documented layout)"""
    cout, cin, kh, kw = w.shape
    wp = torch.zeros((cin // 32, kh * kw, cout_pad, 32), dtype=w.dtype)
    wp[:, :, :cout, :] = w.reshape(cout, cin // 32, 32, kh * kw).permute(1, 3, 0, 2)
    return wp.reshape(-1)


def test_weight_packers_tiled_and_bf16(ops):
    """The LDS-tiled layout converters: the forward packer against an index restatement of its layout, pack -> unpack round trips
    (scale / accumulate), and every bf16 packer bit-for-bit against to_bf16 of its f32 twin (conv, padded, stride-1 / stride-2 input
    gradient incl. 5x5, deconvolution with a channel count that is not a multiple of 32, fc6's input gradient)."""
    g = torch.Generator().manual_seed(11)
    for cout, cin, k in ((64, 32, 3), (256, 128, 5), (96, 64, 1), (128, 96, 4)):
        w = torch.randn(cout, cin, k, k, generator=g)
        wd = w.to(DEV)
        wp = ops.conv2d_pack_weight(wd)
        assert torch.equal(wp.cpu(), _pack_index_reference(w, cout))
        assert torch.equal(ops.conv2d_pack_weight(wd, as_bf16=True), ops.to_bf16(wp))
        pad = ops.pad64(cout + 2)
        wpp = ops.conv2d_pack_weight_padded(wd, pad)
        assert torch.equal(wpp.cpu(), _pack_index_reference(w, pad))
        assert torch.equal(ops.conv2d_pack_weight_padded(wd, pad, as_bf16=True), ops.to_bf16(wpp))
        back = torch.full_like(wd, 3.0)
        ops.conv2d_unpack_weight(wpp, back, CoutPad=pad)
        assert torch.equal(back, wd)
        ops.conv2d_unpack_weight(wpp, back, CoutPad=pad, scale=0.5, accumulate=True)
        assert torch.equal(back, wd + 0.5 * wd)
        for s, p in ((1, k // 2), (2, k // 2)):
            if k == 1 and s == 2:
                continue
            assert torch.equal(ops.conv2d_dgrad_pack_weight(wd, s, p, as_bf16=True), ops.to_bf16(ops.conv2d_dgrad_pack_weight(wd, s, p)))
    for cin, cout in ((1026, 64), (64, 256)):
        wd = torch.randn(cin, cout, 4, 4, generator=g).to(DEV)
        assert torch.equal(ops.deconv4x4s2_pack_weight(wd, as_bf16=True), ops.to_bf16(ops.deconv4x4s2_pack_weight(wd)))
    wf = torch.randn(64, 64 * 3 * 5, generator=g).to(DEV)
    assert torch.equal(ops.fc_dgrad_pack_weight(wf, 64, 3, 5, as_bf16=True), ops.to_bf16(ops.fc_dgrad_pack_weight(wf, 64, 3, 5)))
    back = torch.empty_like(wf)
    ops.fc_unpack_weight(ops.fc_pack_weight(wf, 64, 3, 5), back, 64, 3, 5)
    assert torch.equal(back, wf)


def test_first_layer_bf16_linear_no_bias_and_padded_wgrad_rows(ops):
    """two API corners the executor relies on or documents: (i) the persistent first-layer kernel with slope 1 and no bias (the 10-channel
    input variant adds the mask group's convolution before the activation); (ii) dim_conv2d_wgrad_oihw delivering only the first
    Cout_rows rows of a channel-padded gradient"""
    g = torch.Generator().manual_seed(11)
    x = torch.randn((2, 8, 50, 70), generator=g)
    w = torch.randn((64, 8, 7, 7), generator=g) / np.sqrt(8 * 49)
    ref = F.conv2d(r16(x), r16(w), None, stride=2, padding=3).permute(0, 2, 3, 1).numpy()
    wp = ops.to_bf16(ops.conv2d_pack_weight(w.to(DEV)))
    y = ops.conv2d_fwd(nhwc(x), wp, None, 64, 7, 7, 2, 3, slope=1.0, splits=1, tile=6).cpu().numpy()
    assert np.abs(y - ref).max() <= 1e-4 * np.abs(ref).max() + 2e-5
    xx = torch.randn((2, 64, 12, 14), generator=g, dtype=torch.float64)
    dy = torch.randn((2, 128, 12, 14), generator=g, dtype=torch.float64)
    full = torch.nn.grad.conv2d_weight(r16(xx), (128, 64, 3, 3), r16(dy), stride=1, padding=1)
    part = torch.full((100, 64, 3, 3), 4.0, device=DEV)
    ops.conv2d_wgrad_oihw(nhwc(xx.float()), 64, nhwc(dy.float()), 128, 3, 3, 1, 1, part, splits=2, bf16_mfma=True)
    assert (part.cpu().double() - full[:100]).abs().max().item() <= 1e-4 * full.abs().max().item() + 1e-5
