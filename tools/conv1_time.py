"""flow_conv1 (7x7 / s2, 8 -> 64 channels, 16 x 480 x 640): three-term kernel vs the f32 pipe, values and time"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import numpy as np, torch
from lib.hip import ops
torch.manual_seed(0)
N = 16
x = torch.randn((N, 480, 640, 8), device="cuda:0") * 60.0
x[..., 6:] = 0
w = torch.randn((64, 8, 7, 7), device="cuda:0") * 0.05
w[:, 6:] = 0
b = torch.randn(64, device="cuda:0")
wp = ops.conv2d_pack_weight(w)
ys = {}
for split in ((1,) if os.environ.get("DIM_HIP_LIB") else (1, 0, 1)):
    ops.set_winograd_split(split)
    for _ in range(3):
        y = ops.conv2d_fwd(x, wp, b, 64, 7, 7, 2, 3, slope=0.1, splits=1, tile=6)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = ops.conv2d_fwd(x, wp, b, 64, 7, 7, 2, 3, slope=0.1, splits=1, tile=6)
    e1.record()
    torch.cuda.synchronize()
    ys[split] = y.cpu().numpy().astype(np.float64)
    print("split {}: {:.1f} us per launch".format(split, e0.elapsed_time(e1) * 100), flush=True)
if os.environ.get("DIM_HIP_LIB"):
    sys.exit(0)
import torch.nn.functional as F
ref = F.leaky_relu(F.conv2d(x[:2].cpu().double().permute(0, 3, 1, 2), w.cpu().double(), b.cpu().double(), stride=2, padding=3), 0.1).permute(0, 2, 3, 1).numpy()
sc = np.abs(ref).max()
print("err / max|y|: three-term {:.2e}  f32 pipe {:.2e}   (max|y| {:.1f})".format(np.abs(ys[1][:2] - ref).max() / sc, np.abs(ys[0][:2] - ref).max() / sc, sc))
