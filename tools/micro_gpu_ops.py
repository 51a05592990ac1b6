"""GPU-side counterparts of tools/ref_micro_baseline.py: batched SE(3) compose / delta and depth->flow on cuda:0, HIP-event timed."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import numpy as np, torch
from lib.hip import ops

d = "cuda:0"
B = 16
rng = np.random.default_rng(0)
def rand_pose(n):
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)
    t = np.stack([rng.uniform(-.2, .2, n), rng.uniform(-.15, .15, n), rng.uniform(.6, 1.2, n)], 1)
    return np.concatenate([R, t[:, :, None]], 2).astype(np.float32)
def timeit(fn, n=200):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
ps, pt = torch.from_numpy(rand_pose(B)).to(d), torch.from_numpy(rand_pose(B)).to(d)
se3 = torch.from_numpy(rng.normal(size=(B, 7)).astype(np.float32)).to(d)
z3, o3 = np.zeros(3, np.float32), np.ones(3, np.float32)
out_pose = torch.empty((B, 3, 4), device=d)
K = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]], np.float32)
depth = torch.rand((B, 1, 480, 640), device=d) * 0.5 + 0.6
KT = ops.pose_to_KT(ps, pt, K)
flow, valid = torch.empty((B, 2, 480, 640), device=d), torch.empty((B, 1, 480, 640), device=d)
res = {"device": torch.cuda.get_device_name(0), "batch": B, "unit": "seconds per call (whole batch)"}
res["dim_se3_compose (16 poses)"] = timeit(lambda: ops.se3_compose(ps, se3, "CAMERA", z3, o3, out=out_pose))
res["dim_se3_delta (16 poses)"] = timeit(lambda: ops.se3_delta(ps, pt, "CAMERA", z3, o3))
res["dim_depth_to_flow 480x640 (16 pairs)"] = timeit(lambda: ops.depth_to_flow(depth, depth, KT, np.linalg.inv(K).astype(np.float32), flow=flow, valid=valid))
print(json.dumps(res, indent=1))
