"""Time fc6's forward (dim_fc_fwd: weight stream + partial reduce) at B pairs, cold weights: between two launches a 512 MB buffer is
streamed so that the 84 MB of weights come from HBM as they do inside a forward.  usage: fc_fwd_time.py [B]
env: DIM_FC16_DEPTH (1 / 2 / 3 chunks in flight), DIM_FC_WGS (workgroups = K groups)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import torch  # noqa: E402

from lib.hip import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.randn((B, 8, 10, 1024), device="cuda:0")
w = torch.randn((256, 81920), device="cuda:0") * 0.01
wp = ops.fc_pack_weight(w, 1024, 8, 10)
bias = torch.zeros(256, device="cuda:0")
big = torch.empty(128 << 20, device="cuda:0")
y = ops.fc_fwd(x, wp, bias, 256, slope=0.1)
torch.cuda.synchronize()
for cold in (True, False):
    ts = []
    for _ in range(20):
        if cold:
            big.add_(1.0)   # 1 GB of traffic: the weights leave the Infinity Cache
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.fc_fwd(x, wp, bias, 256, slope=0.1)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print("fc_fwd B={} {}: median {:.1f} us, min {:.1f} us  ({:.2f} TB/s of the 84 MB at the median)".format(
        B, "cold" if cold else "warm", ts[len(ts) // 2], ts[0], 83.9e6 / ts[len(ts) // 2] / 1e6))
