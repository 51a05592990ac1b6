"""Split (3 x bf16, six products) plane GEMMs against the f32 matrix pipe and a float64 reference, with launch times."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import numpy as np, torch
import torch.nn.functional as F
from lib.hip import ops

torch.manual_seed(0)
cases = [  # N, H, W, Cin, Cout, tile, m
    (2, 60, 80, 256, 256, 5, 4),
    (16, 60, 80, 256, 256, 5, 4),
    (16, 30, 40, 512, 512, 5, 4),
    (2, 60, 80, 64, 128, 4, 4),
    (16, 8, 10, 1024, 1024, 7, 4),
    (3, 30, 40, 256, 256, 5, 2),
]
for N, H, W, Cin, Cout, tile, m in cases:
    x = torch.randn((N, H, W, Cin), device="cuda:0")
    w = torch.randn((Cout, Cin, 3, 3), device="cuda:0") * (1.0 / np.sqrt(9 * Cin))
    b = torch.randn(Cout, device="cuda:0") * 0.1
    wp = ops.winograd_pack_weight(w, m=m)
    res = {}
    for split in (1, 0):
        ops.set_winograd_split(split)
        ev = []
        for _ in range(3):
            ev.clear()
            y = ops.conv2d_fwd_winograd(x, Cin, wp, b, Cout, slope=1.0, tile=tile, m=m, events=ev)
        torch.cuda.synchronize()
        res[split] = (y.cpu().numpy().astype(np.float64), {k: s.elapsed_time(e) * 1e3 for k, s, e in ev})
    ops.set_winograd_split(1)
    line = "N{} {}x{} {}->{} tile {} m {}:".format(N, H, W, Cin, Cout, tile, m)
    if N <= 3:
        ref = F.conv2d(x.cpu().double().permute(0, 3, 1, 2), w.cpu().double(), b.cpu().double(), padding=1).permute(0, 2, 3, 1).numpy()
        sc = np.abs(ref).max()
        line += " err/max|y| split {:.2e} f32 {:.2e}".format(np.abs(res[1][0] - ref).max() / sc, np.abs(res[0][0] - ref).max() / sc)
    line += " split-f32 {:.2e}  gemm us split {:.1f} f32 {:.1f}".format(np.abs(res[1][0] - res[0][0]).max() / np.abs(res[0][0]).max(),
                                                                       res[1][1]["conv"], res[0][1]["conv"])
    print(line, flush=True)
