"""Time dim_depth_to_flow at 16 pairs of 480 x 640 (6.144 MB per pair: two planes read, three written)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lib.hip import ops  # noqa: E402
from lib.render_hip.render_py_multi import Render_Py  # noqa: E402
from lib.utils import synthetic as syn  # noqa: E402

B, d = 16, "cuda:0"
models = syn.make_models(seed=2333, n_models=1, subdiv=5)
rm = Render_Py(None, ["ape"], syn.LINEMOD_K, meshes=models)
cls, gt, init = syn.sample_pairs(5, B, n_classes=1)
ci = torch.from_numpy(cls.astype(np.int32)).to(d)
ds, dt = torch.empty((B, 1, 480, 640), device=d), torch.empty((B, 1, 480, 640), device=d)
rm.render_batch(ci, torch.from_numpy(init).to(d), depth=ds)
rm.render_batch(ci, torch.from_numpy(gt).to(d), depth=dt)
KT = ops.pose_to_KT(torch.from_numpy(init).to(d), torch.from_numpy(gt).to(d), syn.LINEMOD_K)
Kinv = np.linalg.inv(syn.LINEMOD_K).astype(np.float32)
flow, valid = ops.depth_to_flow(ds, dt, KT, Kinv)
big = torch.empty(128 << 20, device=d)
for cold in (True, False):
    ts = []
    for _ in range(30):
        if cold:
            big.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.depth_to_flow(ds, dt, KT, Kinv, flow=flow, valid=valid)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print("depth_to_flow B={} {}: median {:.1f} us  min {:.1f} us  -> {:.2f} TB/s of {:.1f} MB".format(B, "cold" if cold else "warm", ts[len(ts) // 2], ts[0],
                                                                                                   B * 6.144e6 / ts[len(ts) // 2] / 1e6, B * 6.144))
