"""few-row layers (conv5_1: T = 320 tile rows, conv5: 81 planes): plane-GEMM time per tile, split vs f32"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import numpy as np, torch
from lib.hip import ops
torch.manual_seed(0)

def t_of(fn):
    ts = []
    for _ in range(6):
        ev = []
        fn(ev)
        torch.cuda.synchronize()
        ts.append([s.elapsed_time(e) * 1e3 for k, s, e in ev if k == "conv"][0])
    return min(ts[2:])

N = 16
x = torch.randn((N, 15, 20, 512), device="cuda:0"); w = torch.randn((512, 512, 3, 3), device="cuda:0") * 0.02; b = torch.zeros(512, device="cuda:0")
wp = ops.winograd_pack_weight(w, m=4)
x2 = torch.randn((N, 30, 40, 512), device="cuda:0"); wp2 = ops.winograd3x3s2_pack_weight(w)
for split in (1, 0):
    ops.set_winograd_split(split)
    for tile in (4, 6, 7, 3):
        a = t_of(lambda ev: ops.conv2d_fwd_winograd(x, 512, wp, b, 512, slope=1.0, tile=tile, m=4, events=ev))
        c = t_of(lambda ev: ops.conv2d_fwd_winograd3x3s2(x2, 512, wp2, b, 512, slope=1.0, tile=tile, events=ev))
        print("split {} tile {}: conv5_1 (36 planes) {:.1f} us   conv5 (81 planes) {:.1f} us".format(split, tile, a, c), flush=True)
