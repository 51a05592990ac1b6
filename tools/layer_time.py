"""whole Winograd layers (input transform / plane GEMMs / output transform, us) for the encoder's shapes at 16 pairs"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import numpy as np, torch
from lib.hip import ops
torch.manual_seed(0)
tot = 0.0
for name, (N, H, W, Cin, Cout, tile) in {"conv3_1": (16, 60, 80, 256, 256, 5), "conv4_1": (16, 30, 40, 512, 512, 5), "conv5_1": (16, 15, 20, 512, 512, 7),
                                         "conv6_1": (16, 8, 10, 1024, 1024, 7)}.items():
    x = torch.randn((N, H, W, Cin), device="cuda:0"); w = torch.randn((Cout, Cin, 3, 3), device="cuda:0") * 0.02; b = torch.zeros(Cout, device="cuda:0")
    wp = ops.winograd_pack_weight(w, m=4)
    rows = []
    for _ in range(8):
        ev = []
        ops.conv2d_fwd_winograd(x, Cin, wp, b, Cout, slope=0.1, tile=tile, m=4, events=ev)
        torch.cuda.synchronize()
        rows.append([s.elapsed_time(e) * 1e3 for k, s, e in ev])
    r = np.median(np.array(rows[3:]), axis=0)
    tot += r.sum()
    print("{:8s} in {:6.1f}  gemm {:6.1f}  out {:6.1f}  sum {:6.1f}".format(name, r[0], r[1], r[2], r.sum()), flush=True)
print("total {:.1f}".format(tot))
