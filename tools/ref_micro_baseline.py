"""micro-baselines of the reference's OWN importable numpy functions, timed on the host cores of the build
container (the reference cannot travel to the GPU box).  Writes profiles/r01_reference_micro_baselines.json.
    python tools/ref_micro_baseline.py          (needs /root/reference)"""
import json
import os
import sys
import time

import numpy as np

np.float = float  # noqa: numpy-2 shim for RT_transform.py:246-247
np.int = int
np.maximum_sctype = lambda t: np.longdouble
sys.path.insert(0, "/root/reference")
from lib.pair_matching import RT_transform as RT  # noqa: E402
from lib.pair_matching.flow import calc_flow  # noqa: E402
from lib.utils.pose_error import add, adi  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]])
rng = np.random.default_rng(0)


def rand_pose():
    q = rng.normal(size=4)
    R = RT.quat2mat(q / np.linalg.norm(q))
    return np.concatenate([R, np.array([[rng.uniform(-.2, .2)], [rng.uniform(-.15, .15)], [rng.uniform(.6, 1.2)]])], axis=1)


def timeit(fn, n):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n


ps, pt = rand_pose(), rand_pose()
q, td = rng.normal(size=4), rng.normal(size=3) * 0.05
z3, o3 = np.zeros(3), np.ones(3)
out = {"host": "build container, {} cores, numpy {}".format(os.cpu_count(), np.__version__), "unit": "seconds per call"}
out["RT_transform (per pose)"] = timeit(lambda: RT.RT_transform(ps, q, td, z3, o3, "CAMERA"), 2000)
out["calc_RT_delta QUAT (per pose)"] = timeit(lambda: RT.calc_RT_delta(ps, pt, z3, o3, "CAMERA", "QUAT"), 2000)
# calc_flow on a 480x640 pair: depth of a sphere in front of the camera (lib/pair_matching/flow.py:12-81)
ys, xs = np.meshgrid(np.arange(480), np.arange(640), indexing="ij")
d = np.where((xs - 320) ** 2 + (ys - 240) ** 2 < 120 ** 2, 0.9, 0.0).astype(np.float32)
p0 = rand_pose(); p0[:, 3] = [0, 0, 0.9]
p1 = p0.copy(); p1[:, 3] += [0.01, -0.01, 0.02]
out["calc_flow 480x640 (per pair)"] = timeit(lambda: calc_flow(d, p0, p1, K, d, thresh=3e-3, standard_rep=False), 3)
pts = rng.normal(size=(3000, 3)) * 0.05
out["add 3000 pts (per pose)"] = timeit(lambda: add(ps[:, :3], ps[:, 3], pt[:, :3], pt[:, 3], pts), 500)
out["adi 3000 pts (per pose)"] = timeit(lambda: adi(ps[:, :3], ps[:, 3], pt[:, :3], pt[:, 3], pts), 50)
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_reference_micro_baselines.json"), "w"), indent=1)
for k, v in out.items():
    print(k, v)
