"""conv6 (3x3 / s2, 512 -> 1024 on 15 x 20, 16 pairs): direct kernel vs phase images + 81 three-term plane GEMMs"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import numpy as np, torch
from lib.hip import ops
torch.manual_seed(0)
N = 16
x = torch.randn((N, 15, 20, 512), device="cuda:0"); w = torch.randn((1024, 512, 3, 3), device="cuda:0") * 0.02; b = torch.zeros(1024, device="cuda:0")
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
wp = ops.conv2d_pack_weight(w)
ws = torch.empty(ops.lib().dim_conv2d_workspace_floats(N, 15, 20, 512, 1024, 3, 3, 2, 1, 0), device="cuda:0")
print("direct (auto plan): {:.1f} us".format(timed(lambda: ops.conv2d_fwd(x, wp, b, 1024, 3, 3, 2, 1, slope=0.1, splits=0, tile=0, workspace=ws))))
w3 = ops.winograd3x3s2_pack_weight(w)
wsw = torch.empty(ops.lib().dim_winograd3x3s2_workspace_floats(N, 15, 20, 512, 1024), device="cuda:0")
for split in (1, 0):
    ops.set_winograd_split(split)
    for tile in (7, 4, 6, 3):
        print("phase images, split {} tile {}: {:.1f} us".format(split, tile, timed(lambda: ops.conv2d_fwd_winograd3x3s2(x, 512, w3, b, 1024, slope=0.1, tile=tile, workspace=wsw))), flush=True)
