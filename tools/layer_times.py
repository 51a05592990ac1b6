"""Per-layer time / TFLOP/s of the encoder inside the real network forward (HIP events on the launch stream).  usage: layer_times.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from deepim.config.config import config as cfg, update_config
from deepim.symbols.deepIM_flownet import deepIM_flownet
from deepim.core.tester import Predictor
from lib.render_hip.render_py_multi import Render_Py
from lib.utils import synthetic as syn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
update_config(os.path.join(ROOT, "mx-deepim_amd/experiments/deepim/cfgs/deepim_hip_LM_ape_test.yaml"))
sym = deepIM_flownet(); sym.get_symbol(cfg, False)
params = sym.init_weights(cfg, {}, {}, seed=0)
models = syn.make_models(seed=2333, n_models=1, subdiv=5)
rm = Render_Py(None, cfg.dataset.class_name, cfg.dataset.INTRINSIC_MATRIX, meshes=models)
batch = syn.build_device_batch(rm, B, seed=5)
pred = Predictor(cfg, params, B)
net = pred.net
for _ in range(2):
    net.forward_test(batch)
torch.cuda.synchronize()
events = {}
reps = 10
for _ in range(reps):
    net.zoom(batch); net.encoder(events=events); net.head()
torch.cuda.synchronize()
tot = 0.0
for name, evs in events.items():
    info = net.layer_info[name]
    conv = [ev[1].elapsed_time(ev[2]) for ev in evs if ev[0] == "conv"]
    red = [ev[1].elapsed_time(ev[2]) for ev in evs if ev[0] != "conv"]
    ms = sum(conv) / reps
    tot += ms + sum(red) / reps
    print("%-11s tile=%d splits=%d  conv %.4f ms  %.1f TF   reduce %.4f ms" % (name, info["tile"], info["splits"], ms, info["flops"] / ms / 1e9, sum(red) / reps))
print("total %.3f ms per forward" % tot)
