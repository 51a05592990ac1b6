"""Does splitting the 16 pairs of a refinement step into concurrent lanes pay?  Times K steps of (a) one Refiner with B pairs,
(b) L Refiners with B / L pairs each, every one replaying its own captured graph on its own stream (HBM-bound kernels of one lane
can run under the MFMA-bound GEMMs of another).  usage: two_lane_probe.py [B] [lanes] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from deepim.config.config import config as cfg, update_config
from deepim.core.tester import Predictor, Refiner
from deepim.symbols.deepIM_flownet import deepIM_flownet
from lib.render_hip.render_py_multi import Render_Py
from lib.utils import synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
LANES = int(sys.argv[2]) if len(sys.argv) > 2 else 2
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = "cuda:0"
update_config(os.path.join(ROOT, "mx-deepim_amd/experiments/deepim/cfgs/deepim_hip_LM_ape_test.yaml"))
sym = deepIM_flownet(); sym.get_symbol(cfg, is_train=False)
params = sym.init_weights(cfg, {}, {}, seed=0)
params["trans_weight"] = (np.random.RandomState(1).randn(3, 256) * 0.002).astype(np.float32)
models = syn.make_models(seed=2333, n_models=len(cfg.dataset.class_name), subdiv=5)


def make(b, seed):
    rm = Render_Py(None, cfg.dataset.class_name, cfg.dataset.INTRINSIC_MATRIX, zNear=cfg.dataset.ZNEAR, zFar=cfg.dataset.ZFAR, device=dev, meshes=models)
    pred = Predictor(cfg, params, b, device=dev)
    batch = syn.build_device_batch(rm, b, seed=seed, n_classes=len(models), pixel_means=cfg.network.PIXEL_MEANS, device=dev)
    r = Refiner(cfg, pred, rm, b, capture_graph=True)
    r.load(batch["image_observed"], batch["image_rendered"], batch["mask_observed"], batch["mask_rendered"], batch["src_pose"], batch["class_index"])
    r.refine(); torch.cuda.synchronize()
    return r


def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(STEPS): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3


one = make(B, 1000)
t1 = timed(one.refine)
print("one lane   B=%d: %.3f ms/step  %.1f refinements/s" % (B, t1, B / t1 * 1e3), flush=True)
lanes = [make(B // LANES, 1000 + i) for i in range(LANES)]
streams = [torch.cuda.Stream(device=dev) for _ in range(LANES)]


def multi():
    cur = torch.cuda.current_stream()
    for r, s in zip(lanes, streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            r.refine()
    for s in streams:
        cur.wait_stream(s)


tm = timed(multi)
print("%d lanes  B=%d each: %.3f ms/step  %.1f refinements/s" % (LANES, B // LANES, tm, B / tm * 1e3), flush=True)
ts = timed(lambda: [r.refine() for r in lanes])
print("%d lanes one after the other: %.3f ms/step" % (LANES, ts), flush=True)
