import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import torch, numpy as np
from lib.hip import ops
B = 16; d = "cuda:0"
p = {"fc7_weight": torch.randn(256, 256, device=d) * 0.05, "fc7_bias": torch.zeros(256, device=d), "rot_weight": torch.randn(4, 256, device=d) * 0.05,
     "rot_bias": torch.zeros(4, device=d), "trans_weight": torch.randn(3, 256, device=d) * 0.05, "trans_bias": torch.zeros(3, device=d)}
y6 = torch.randn(B, 256, device=d); zf = torch.ones(B, 4, device=d)
se3 = ops.pose_head_fwd(y6, p, zf)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): ops.pose_head_fwd(y6, p, zf, se3=se3)
e1.record(); torch.cuda.synchronize()
print("pose_head_fwd B=16: %.2f us per call (back-to-back launches)" % (e0.elapsed_time(e1) / 200 * 1e3))
