"""time the Winograd path vs the direct kernel for one 3x3/s1 encoder layer.  usage: one_wino.py LAYER [B] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import torch
from deepim.symbols.deepIM_flownet import ENCODER
from lib.hip import ops
layer = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 16; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
h, w, c = 480, 640, 8
for name, cout, k, s, p in ENCODER:
    if name == layer: break
    h, w = ops.conv_out_hw(h, w, k, k, s, p); c = cout
assert k == 3 and s == 1
x = torch.randn((B, h, w, c), device="cuda:0"); wt = torch.randn((cout, c, 3, 3), device="cuda:0") * 0.01; bias = torch.zeros(cout, device="cuda:0")
wd = ops.conv2d_pack_weight(wt)
ws = torch.empty(ops.lib().dim_winograd_workspace_floats(B, h, w, c, cout, 2), device="cuda:0")
flops = 2.0 * B * h * w * cout * c * 9
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
y = torch.empty((B, h, w, cout), device="cuda:0")
for m in (4,):
    wp = ops.winograd_pack_weight(wt, m=m)
    for tile in (4, 5, 6, 7):
        ev = []
        ops.conv2d_fwd_winograd(x, c, wp, bias, cout, tile=tile, out=y, workspace=ws, m=m, events=ev); torch.cuda.synchronize()
        parts = " ".join("%s %.1f us" % (t, 1e3 * a.elapsed_time(b)) for t, a, b in ev)
        ms = timeit(lambda: ops.conv2d_fwd_winograd(x, c, wp, bias, cout, tile=tile, out=y, workspace=ws, m=m))
        print("%s winograd F(%dx%d) tile %d: %.4f ms  (%.1f TF direct-equivalent)  [%s]" % (layer, m, m, tile, ms, flops / ms / 1e9, parts))
ms = timeit(lambda: ops.conv2d_fwd(x, wd, bias, cout, 3, 3, 1, 1, tile=4, out=y))
print("%s direct tile 4 s1: %.4f ms  %.1f TF" % (layer, ms, flops / ms / 1e9))
