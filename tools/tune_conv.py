"""Per-layer sweep of (tile, splits) for the encoder shapes at a given batch; prints ms and TFLOP/s.
usage: python tools/tune_conv.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import torch  # noqa: E402

from deepim.symbols.deepIM_flownet import ENCODER  # noqa: E402
from lib.hip import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda:0"
h, w, c = 480, 640, 8
layers = []
for name, cout, k, s, p in ENCODER:
    layers.append((name, h, w, c, cout, k, k, s, p))
    h, w = ops.conv_out_hw(h, w, k, k, s, p)
    c = cout
layers.append(("fc6", 8, 10, 1024, 256, 8, 10, 1, 0))
for name, h, w, c, cout, kh, kw, s, p in layers:
    x = torch.randn((B, h, w, c), device=dev)
    wt = torch.randn((cout, c, kh, kw), device=dev) * 0.01
    wp = ops.conv2d_pack_weight(wt) if name != "fc6" else ops.fc_pack_weight(wt.reshape(cout, -1), c, kh, kw)
    bias = torch.zeros(cout, device=dev)
    ho, wo = ops.conv_out_hw(h, w, kh, kw, s, p)
    flops = 2.0 * B * ho * wo * cout * c * kh * kw
    res = []
    for tile in (1, 2, 3, 4):
        if tile in (1, 4) and (cout % 128 or c == 8):
            continue
        for splits in ((1, 2, 3, 4, 6, 8) if name != "fc6" else (40, 80, 160, 320)):
            try:
                y = ops.conv2d_fwd(x, wp, bias, cout, kh, kw, s, p, splits=splits, tile=tile)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ws = torch.empty((max(4, ops.lib().dim_conv2d_workspace_floats(B, h, w, c, cout, kh, kw, s, p, splits)),), device=dev)
                n = 40  # sustained: short bursts run at boost clocks and flatter every candidate
                e0.record()
                for _ in range(n):
                    ops.conv2d_fwd(x, wp, bias, cout, kh, kw, s, p, splits=splits, tile=tile, out=y, workspace=ws)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / n
                res.append((ms, tile, splits))
            except Exception as e:  # noqa
                print("  skip", name, tile, splits, str(e)[:80])
    res.sort()
    best = res[0]
    print("{:10s} M={:8d} K={:6d} N={:5d}  best tile={} splits={}  {:.3f} ms  {:.1f} TF   | ".format(
        name, B * ho * wo, c * kh * kw, cout, best[1], best[2], best[0], flops / best[0] / 1e9) +
        "  ".join("t{}s{}:{:.3f}".format(t, sp, ms) for ms, t, sp in res[:6]), flush=True)
