"""time the phase-image Winograd path vs the direct kernel for a 5x5/s2 encoder layer.  usage: one_wino5.py conv2|conv3 [B] [reps]"""
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
assert k == 5 and s == 2
x = torch.randn((B, h, w, c), device="cuda:0"); wt = torch.randn((cout, c, 5, 5), device="cuda:0") * 0.01; bias = torch.zeros(cout, device="cuda:0")
wd = ops.conv2d_pack_weight(wt); wp = ops.winograd5x5s2_pack_weight(wt)
ws = torch.empty(max(ops.lib().dim_winograd5x5s2_workspace_floats(B, h, w, c, cout), 8 * 304 * 128 * 128), device="cuda:0")
ho, wo = h // 2, w // 2
flops = 2.0 * B * ho * wo * cout * c * 25
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
y = torch.empty((B, ho, wo, cout), device="cuda:0")
for tile in (4, 5, 6, 7):
    ev = []
    ops.conv2d_fwd_winograd5x5s2(x, c, wp, bias, cout, tile=tile, out=y, workspace=ws, events=ev); torch.cuda.synchronize()
    parts = " ".join("%s %.1f us" % (t, 1e3 * a.elapsed_time(b)) for t, a, b in ev)
    ms = timeit(lambda: ops.conv2d_fwd_winograd5x5s2(x, c, wp, bias, cout, tile=tile, out=y, workspace=ws))
    print("%s phase-Winograd tile %d: %.4f ms  (%.1f TF direct-equivalent)  [%s]" % (layer, tile, ms, flops / ms / 1e9, parts))
ms = timeit(lambda: ops.conv2d_fwd(x, wd, bias, cout, 5, 5, 2, 2, tile=4, splits=0, out=y, workspace=ws))
print("%s direct tile 4 auto: %.4f ms  %.1f TF" % (layer, ms, flops / ms / 1e9))
