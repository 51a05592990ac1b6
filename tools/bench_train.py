"""Time one training iteration (forward + backward + update) at batch B on one GPU; prints per-phase ms.
usage: bench_train.py [B] [f32|bf16]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from deepim.config.config import config as cfg, update_config
from deepim.symbols.deepIM_flownet import deepIM_flownet
from deepim.core.module import MutableModule, fit_batch
from lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti
from lib.render_hip.render_py_multi import Render_Py
from lib.utils import synthetic as syn
from lib.hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
DTYPE = sys.argv[2] if len(sys.argv) > 2 else "f32"
update_config(os.path.join(ROOT, "mx-deepim_amd/experiments/deepim/cfgs/deepim_hip_LM_ape_test.yaml"))
cfg.TRAIN.lr = 1e-4
sym = deepIM_flownet(); sym.get_symbol(cfg, True)
params = sym.init_weights(cfg, {}, {}, seed=0)
models = syn.make_models(seed=2333, n_models=1, subdiv=5)
rm = Render_Py(None, cfg.dataset.class_name, cfg.dataset.INTRINSIC_MATRIX, meshes=models)
batch = syn.build_device_train_batch(rm, B, seed=5, models=models)
mod = MutableModule(cfg, params, B, compute_dtype=DTYPE)
upd = batchUpdaterPyMulti(cfg, 480, 640, render_machine=rm)
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
import json
res = {"B": B, "forward_ms": timed(lambda: mod.forward(batch)), "backward_ms": timed(lambda: mod.backward(batch)),
       "update_repack_ms": timed(lambda: mod.update(0.0))}
preds = mod.forward(batch)
res["batch_updater_ms"] = timed(lambda: upd.forward(batch, preds))
t = timed(lambda: (mod.forward_backward(batch), mod.update(1e-4)))
res["train_iteration_ms"] = t
res["pair_iterations_per_s"] = B / t * 1e3
res["config"] = "LINEMOD 'ape' training graph (encoder + decoder + flow / mask / point-matching losses), SGD momentum, %s, 1x MI355X, synthetic pairs" % DTYPE
for k, v in res.items():
    print(k, ("%.2f" % v) if isinstance(v, float) else v)
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()}))
