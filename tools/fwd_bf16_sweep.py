"""Sweep (tile, split-K) of the bf16 forward convolutions on the small maps at B = 16 (conv5 .. conv6_1): time incl. the slab reduce.
usage: python tools/fwd_bf16_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import torch  # noqa: E402

from lib.hip import ops  # noqa: E402

SHAPES = {"conv4": (60, 80, 256, 512, 3, 2, 1), "conv5": (30, 40, 512, 512, 3, 2, 1), "conv5_1": (15, 20, 512, 512, 3, 1, 1),
          "conv6": (15, 20, 512, 1024, 3, 2, 1), "conv6_1": (8, 10, 1024, 1024, 3, 1, 1)}
B, dev = 16, "cuda:0"
for name, (H, W, Cin, Cout, k, s, p) in SHAPES.items():
    Ho, Wo = ops.conv_out_hw(H, W, k, k, s, p)
    x = torch.randn((B, H, W, Cin), device=dev)
    w = ops.conv2d_pack_weight(torch.randn((Cout, Cin, k, k), device=dev) * 0.01, as_bf16=True)
    bias = torch.zeros((Cout,), device=dev)
    y = torch.empty((B, Ho, Wo, Cout), device=dev)
    ws = torch.empty((16 * B * Ho * Wo * Cout,), device=dev)
    out = []
    for tile in (4, 3):
        tiles = -(-B * Ho * Wo // (128 if tile == 4 else 64)) * (Cout // (128 if tile == 4 else 64))
        for sp in (1, 2, 3, 4, 5, 6, 8, 12):
            for _ in range(2):
                ops.conv2d_fwd(x, w, bias, Cout, k, k, s, p, splits=sp, tile=tile, out=y, workspace=ws)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv2d_fwd(x, w, bias, Cout, k, k, s, p, splits=sp, tile=tile, out=y, workspace=ws)
            e1.record()
            e1.synchronize()
            out.append("t{}x{}:{}wg:{:.0f}us".format(tile, sp, tiles * sp, e0.elapsed_time(e1) / 10 * 1e3))
    print(name, " ".join(out))
