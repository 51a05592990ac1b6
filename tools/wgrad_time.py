"""Time the bf16 weight-gradient kernel on the encoder's shapes at B = 16 (HIP events, 20 launches each).
usage: python tools/wgrad_time.py [layer ...]      (DIM_WGB_DBG=<bits> selects a timing-only ablation, see csrc/wgrad.hip)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import torch  # noqa: E402

from lib.hip import ops  # noqa: E402

# name: (H, W, Cin, Cout, k, stride, pad, splits) -- input map size, as MutableModule launches them
SHAPES = {"conv2": (240, 320, 64, 128, 5, 2, 2, 59), "conv3": (120, 160, 128, 256, 5, 2, 2, 15), "conv3_1": (60, 80, 256, 256, 3, 1, 1, 21),
          "conv4": (60, 80, 256, 512, 3, 2, 1, 10), "conv4_1": (30, 40, 512, 512, 3, 1, 1, 5), "conv5": (30, 40, 512, 512, 3, 2, 1, 5),
          "conv5_1": (15, 20, 512, 512, 3, 1, 1, 5), "conv6": (15, 20, 512, 1024, 3, 2, 1, 2), "conv6_1": (8, 10, 1024, 1024, 3, 1, 1, 1),
          # the first layer (8 input lanes) and deconv4's gradient through its convolution view (conv'(dz, k4, s2, p1) against x)
          "flow_conv1": (480, 640, 8, 64, 7, 2, 3, 256), "deconv4": (30, 40, 256, 1088, 4, 2, 1, 1)}
B = 16
dev = "cuda:0"
names = sys.argv[1:] or list(SHAPES)
for name in names:
    H, W, Cin, Cout, k, s, p, splits = SHAPES[name]
    Ho, Wo = ops.conv_out_hw(H, W, k, k, s, p)
    x = torch.randn((B, H, W, Cin), device=dev)
    dz = torch.randn((B, Ho, Wo, Cout), device=dev)
    n = ops.lib().dim_conv2d_packed_weight_floats(Cout, Cin, k, k)
    dw = torch.empty((n,), device=dev)
    ws = torch.empty((ops.lib().dim_conv2d_wgrad_workspace_floats(Cout, Cin, k, k, splits),), device=dev)
    for _ in range(3):
        ops.conv2d_wgrad(x, Cin, dz, Cout, k, k, s, p, dw, splits=splits, workspace=ws, bf16_mfma=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d_wgrad(x, Cin, dz, Cout, k, k, s, p, dw, splits=splits, workspace=ws, bf16_mfma=True)
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    flops = 2.0 * B * Ho * Wo * Cout * Cin * k * k
    print("{:8s} dbg={} {:8.1f} us  {:7.1f} TFLOP/s (incl. slab reduce)".format(name, os.environ.get("DIM_WGB_DBG", "0"), us, flops / us / 1e6))
