"""one shape through the three-term plane GEMM, a few launches (for rocprofv3 --pmc passes)"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import torch
from lib.hip import ops
torch.manual_seed(0)
N, H, W, Cin, Cout, m, tile = 16, 60, 80, 256, 256, 4, 5
x = torch.randn((N, H, W, Cin), device="cuda:0"); w = torch.randn((Cout, Cin, 3, 3), device="cuda:0") * 0.02; b = torch.zeros(Cout, device="cuda:0")
wp = ops.winograd_pack_weight(w, m=m)
for split in (1, 0):
    ops.set_winograd_split(split)
    for _ in range(4):
        ops.conv2d_fwd_winograd(x, Cin, wp, b, Cout, slope=1.0, tile=tile, m=m)
torch.cuda.synchronize()
xc = torch.randn((16, 480, 640, 8), device="cuda:0"); wc = torch.randn((64, 8, 7, 7), device="cuda:0") * 0.05
wpc = ops.conv2d_pack_weight(wc)
ops.set_winograd_split(1)
for _ in range(4):
    ops.conv2d_fwd(xc, wpc, None, 64, 7, 7, 2, 3, slope=0.1, splits=1, tile=6)
torch.cuda.synchronize()
