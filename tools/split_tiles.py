"""split kernel: GEMM time per tile choice on the big shapes"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import numpy as np, torch
from lib.hip import ops
torch.manual_seed(0)
for N, H, W, Cin, Cout, m in [(16, 60, 80, 256, 256, 4), (16, 30, 40, 512, 512, 4), (16, 60, 80, 128, 256, 4)]:
    x = torch.randn((N, H, W, Cin), device="cuda:0")
    w = torch.randn((Cout, Cin, 3, 3), device="cuda:0") * 0.02
    b = torch.zeros(Cout, device="cuda:0")
    wp = ops.winograd_pack_weight(w, m=m)
    for split in (1, 0):
        ops.set_winograd_split(split)
        for tile in (5, 4):
            ts = []
            for _ in range(6):
                ev = []
                ops.conv2d_fwd_winograd(x, Cin, wp, b, Cout, slope=1.0, tile=tile, m=m, events=ev)
                torch.cuda.synchronize()
                ts.append([s.elapsed_time(e) * 1e3 for k, s, e in ev if k == "conv"][0])
            print("N{} {}x{} {}->{} split {} tile {}: {:.1f} us".format(N, H, W, Cin, Cout, split, tile, min(ts[2:])), flush=True)
