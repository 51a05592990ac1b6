"""Sweep the pixel-split count of the bf16 weight-gradient kernel per layer (B = 16): time vs resident-slot quantisation.
usage: python tools/wgrad_sweep.py [layer ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import torch  # noqa: E402

from lib.hip import ops  # noqa: E402
from wgrad_time import SHAPES  # noqa: E402

B, dev = 16, "cuda:0"
for name in (sys.argv[1:] or list(SHAPES)):
    H, W, Cin, Cout, k, s, p, cur = SHAPES[name]
    Ho, Wo = ops.conv_out_hw(H, W, k, k, s, p)
    x = torch.randn((B, H, W, Cin), device=dev)
    dz = torch.randn((B, Ho, Wo, Cout), device=dev)
    nchunks = -(-k * k // 4) if Cin == 8 else k * k * Cin // 32
    tiles = -(-nchunks // 4) * (Cout // 128 if Cout % 128 == 0 else Cout // 64)
    dw = torch.empty((ops.lib().dim_conv2d_packed_weight_floats(Cout, Cin, k, k),), device=dev)
    cands = sorted(set([cur] + [max(1, round(f * 768 / tiles)) for f in (0.5, 0.75, 1.0, 1.34, 1.67, 2.0, 3.0)] + [max(1, 768 // tiles), max(1, 1536 // tiles)]))
    if os.environ.get("DIM_SWEEP_SPLITS"):
        cands = [int(v) for v in os.environ["DIM_SWEEP_SPLITS"].split(",")]
    out = []
    for splits in cands:
        ws = torch.empty((ops.lib().dim_conv2d_wgrad_workspace_floats(Cout, Cin, k, k, splits),), device=dev)
        for _ in range(2):
            ops.conv2d_wgrad(x, Cin, dz, Cout, k, k, s, p, dw, splits=splits, workspace=ws, bf16_mfma=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv2d_wgrad(x, Cin, dz, Cout, k, k, s, p, dw, splits=splits, workspace=ws, bf16_mfma=True)
        e1.record()
        e1.synchronize()
        out.append((splits, tiles * splits, e0.elapsed_time(e1) / 10 * 1e3))
    print(name, "tiles", tiles, "current", cur, " ".join("{}:{}wg:{:.0f}us".format(*o) for o in out))
