"""time the rasteriser passes at batch 16 (normal poses vs object out of view)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import numpy as np, torch
from lib.render_hip.render_py_multi import Render_Py
from lib.utils import synthetic as syn
d = "cuda:0"; B = 16
models = syn.make_models(seed=2333, n_models=1, subdiv=5)
rm = Render_Py(None, ["ape"], syn.LINEMOD_K, meshes=models)
cls, gt, init = syn.sample_pairs(5, B, n_classes=1)
img = torch.empty((B, 3, 480, 640), device=d); dep = torch.empty((B, 1, 480, 640), device=d); msk = torch.empty((B, 1, 480, 640), device=d)
bbox = torch.zeros((B, 4), dtype=torch.int32, device=d)
ci = torch.from_numpy(cls.astype(np.int32)).to(d)
pm = np.array([103.939, 116.779, 123.68], np.float32)
def t(poses, n=100):
    p = torch.from_numpy(poses.astype(np.float32)).to(d)
    f = lambda: rm.render_batch(ci, p, image=img, depth=dep, mask=msk, bbox=bbox, plane_means=pm)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("render_batch normal: %.1f us" % t(gt))
far = gt.copy(); far[:, 0, 3] += 50.0
print("render_batch object out of view: %.1f us" % t(far))
# the refinement loop's shape: no depth plane, image + mask + bbox; then the same with the previous render's box as the dirty-box hint
bbox2 = torch.zeros((B, 4), dtype=torch.int32, device=d)
def t_loop(hint, n=100):
    p = torch.from_numpy(init.astype(np.float32)).to(d)
    boxes = [bbox, bbox2]
    def f(i):
        rm.render_batch(ci, p, image=img, mask=msk, bbox=boxes[i & 1], plane_means=pm, clean_bbox=boxes[(i + 1) & 1] if hint else None)
    f(0); f(1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): f(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("loop-shaped render (image + mask + bbox): %.1f us; with the dirty-box hint: %.1f us" % (t_loop(False), t_loop(True)))
