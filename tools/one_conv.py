"""Run one encoder layer shape repeatedly (profiling target).  usage: python tools/one_conv.py LAYER [B] [tile] [splits] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import torch  # noqa: E402

from deepim.symbols.deepIM_flownet import ENCODER  # noqa: E402
from lib.hip import ops  # noqa: E402

layer = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 3
splits = int(sys.argv[4]) if len(sys.argv) > 4 else 1
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
h, w, c = 480, 640, 8
for name, cout, k, s, p in ENCODER:
    if name == layer:
        break
    h, w = ops.conv_out_hw(h, w, k, k, s, p)
    c = cout
x = torch.randn((B, h, w, c), device="cuda:0")
wt = torch.randn((cout, c, k, k), device="cuda:0") * 0.01
if os.environ.get("ZERO_DATA"):  # power / clock experiment: all-zero operands draw far less MFMA power
    x.zero_()
    wt.zero_()
wp = ops.conv2d_pack_weight(wt, as_bf16=bool(os.environ.get("BF16")))   # BF16=1: the bf16 pipe
bias = torch.zeros(cout, device="cuda:0")
ho, wo = ops.conv_out_hw(h, w, k, k, s, p)
flops = 2.0 * B * ho * wo * cout * c * k * k
y = ops.conv2d_fwd(x, wp, bias, cout, k, k, s, p, splits=splits, tile=tile)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.conv2d_fwd(x, wp, bias, cout, k, k, s, p, splits=splits, tile=tile, out=y)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("{} B={} tile={} splits={}: {:.4f} ms {:.1f} TF".format(layer, B, tile, splits, ms, flops / ms / 1e9))
