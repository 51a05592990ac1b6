"""time the plane-GEMM launch of one Winograd layer (HIP events around the GEMM only).  usage: gemm_time.py [layer] [B] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import torch
from deepim.symbols.deepIM_flownet import ENCODER
from lib.hip import ops
layer = sys.argv[1] if len(sys.argv) > 1 else "conv3_1"; B = int(sys.argv[2]) if len(sys.argv) > 2 else 16; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
h, w, c = 480, 640, 8
for name, cout, k, s, p in ENCODER:
    if name == layer: break
    h, w = ops.conv_out_hw(h, w, k, k, s, p); c = cout
x = torch.randn((B, h, w, c), device="cuda:0"); wt = torch.randn((cout, c, k, k), device="cuda:0") * 0.01; bias = torch.zeros(cout, device="cuda:0")
if k == 3:
    ws = torch.empty(ops.lib().dim_winograd_workspace_floats(B, h, w, c, cout, 4), device="cuda:0")
    y = torch.empty((B, h, w, cout), device="cuda:0")
    wp = ops.winograd_pack_weight(wt, m=4)
    tiles = B * ((h + 3) // 4) * ((w + 3) // 4)
    flops = 2.0 * 36 * tiles * c * cout
    run = lambda tile, ev: ops.conv2d_fwd_winograd(x, c, wp, bias, cout, tile=tile, out=y, workspace=ws, m=4, events=ev)
else:
    ho, wo = (h + 1) // 2, (w + 1) // 2
    ws = torch.empty(ops.lib().dim_winograd5x5s2_workspace_floats(B, h, w, c, cout), device="cuda:0")
    y = torch.empty((B, ho, wo, cout), device="cuda:0")
    wp = ops.winograd5x5s2_pack_weight(wt)
    tiles = B * ((ho + 3) // 4) * ((wo + 3) // 4)
    flops = 2.0 * 36 * tiles * 4 * c * cout
    run = lambda tile, ev: ops.conv2d_fwd_winograd5x5s2(x, c, wp, bias, cout, tile=tile, out=y, workspace=ws, events=ev)
for tile in ((5, 4, 3) if cout % 256 == 0 else (4, 3)):
    tot, n = 0.0, 0
    for r in range(reps):
        ev = []
        run(tile, ev)
        torch.cuda.synchronize()
        if r >= 10:
            t = [1e3 * a.elapsed_time(b) for tag, a, b in ev if tag == "conv"][0]
            tot += t; n += 1
    print("%s tile %d variant %s: GEMM %.1f us  %.1f TFLOP/s executed" % (layer, tile, os.environ.get("DIM_GEMM_VARIANT", "0"), tot / n, flops / (tot / n) / 1e6))
