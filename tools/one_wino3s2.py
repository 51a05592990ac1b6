"""Time a 3x3 / stride-2 encoder layer through the phase-image Winograd path against the direct kernel.
usage: one_wino3s2.py LAYER [B] [gemm tile] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import torch  # noqa: E402

from deepim.symbols.deepIM_flownet import ENCODER  # noqa: E402
from lib.hip import ops  # noqa: E402

layer = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 0
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
h, w, c = 480, 640, 8
for name, cout, k, s, p in ENCODER:
    if name == layer:
        break
    h, w = ops.conv_out_hw(h, w, k, k, s, p)
    c = cout
assert (k, s, p) == (3, 2, 1)
x = torch.randn((B, h, w, c), device="cuda:0")
wt = torch.randn((cout, c, 3, 3), device="cuda:0") * 0.01
bias = torch.zeros(cout, device="cuda:0")
wp = ops.winograd3x3s2_pack_weight(wt)
ws = torch.empty(ops.lib().dim_winograd3x3s2_workspace_floats(B, h, w, c, cout), device="cuda:0")
y = ops.conv2d_fwd_winograd3x3s2(x, c, wp, bias, cout, tile=tile, workspace=ws)
ev = []
ops.conv2d_fwd_winograd3x3s2(x, c, wp, bias, cout, tile=tile, out=y, workspace=ws, events=ev)
torch.cuda.synchronize()
parts = {t: a.elapsed_time(b_) * 1e3 for t, a, b_ in ev}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.conv2d_fwd_winograd3x3s2(x, c, wp, bias, cout, tile=tile, out=y, workspace=ws)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
wd = ops.conv2d_pack_weight(wt)
plan_tile, plan_splits = ops.conv_auto_plan(B * (h // 2) * (w // 2), cout, 9 * (c // 32), cin=c)
wsd = torch.empty(max(4, ops.lib().dim_conv2d_workspace_floats(B, h, w, c, cout, 3, 3, 2, 1, plan_splits)), device="cuda:0")
yd = ops.conv2d_fwd(x, wd, bias, cout, 3, 3, 2, 1, tile=plan_tile, splits=plan_splits, workspace=wsd)
torch.cuda.synchronize()
e0.record()
for _ in range(reps):
    ops.conv2d_fwd(x, wd, bias, cout, 3, 3, 2, 1, tile=plan_tile, splits=plan_splits, out=yd, workspace=wsd)
e1.record()
torch.cuda.synchronize()
md = e0.elapsed_time(e1) / reps
print("{} B={} gemm tile {}: winograd {:.4f} ms (in {:.0f} us, gemm {:.0f} us, out {:.0f} us) | direct (tile {}, splits {}) {:.4f} ms | max |diff| {:.2e}".format(
    layer, B, tile, ms, parts.get("wino_in", 0), parts.get("conv", 0), parts.get("wino_out", 0), plan_tile, plan_splits, md,
    (y - yd).abs().max().item()))
