"""Where does a loader-fed training batch lose time against private copies?  Per batch: host ms in next(), host ms in fit_batch (enqueue),
GPU ms between events around fit_batch, wall ms.  usage: python tools/loader_probe.py"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from deepim.config.config import config as cfg, update_config  # noqa: E402
from deepim.core.loader import PixelCache, TrainDataLoader  # noqa: E402
from deepim.core.module import MutableModule, fit_batch  # noqa: E402
from deepim.symbols.deepIM_flownet import deepIM_flownet  # noqa: E402
from lib.dataset.synthetic_files import write_synthetic_dataset  # noqa: E402
from lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti  # noqa: E402
from lib.render_hip.render_py_multi import Render_Py  # noqa: E402
from lib.utils import synthetic as syn  # noqa: E402

update_config(os.path.join(ROOT, "mx-deepim_amd/experiments/deepim/cfgs/deepim_hip_LM_ape_test.yaml"))
B, dev, nb = 16, "cuda:0", 6
models = syn.make_models(seed=2333, n_models=1, subdiv=5)
rm = Render_Py(None, cfg.dataset.class_name, cfg.dataset.INTRINSIC_MATRIX, meshes=models)
root = tempfile.mkdtemp(prefix="dim_probe_")
try:
    db = write_synthetic_dataset(root, rm, models, list(cfg.dataset.class_name), B * nb, seed=4242)
    cfg.dataset.model_dir = os.path.join(root, "models")
    sym = deepIM_flownet(); sym.get_symbol(cfg, is_train=True)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    upd = batchUpdaterPyMulti(cfg, 480, 640, render_machine=rm)
    mod = MutableModule(cfg, params, B, device=dev, compute_dtype=sys.argv[1] if len(sys.argv) > 1 else "f32")
    loader = TrainDataLoader(None, db, cfg, batch_size=B, shuffle=False, device=dev, workers=16, cache=PixelCache(dev, 8 << 30))
    for ep in range(4):
        loader.reset()
        torch.cuda.synchronize()
        t_next = t_fit = gpu = 0.0
        evs = []
        w0 = time.perf_counter()
        it = iter(loader)
        while True:
            a = time.perf_counter()
            try:
                batch = next(it)
            except StopIteration:
                break
            b = time.perf_counter()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fit_batch(mod, batch, upd, 1e-5)
            e1.record()
            c = time.perf_counter()
            evs.append((e0, e1))
            t_next += b - a
            t_fit += c - b
        torch.cuda.synchronize()
        wall = time.perf_counter() - w0
        gpu = sum(a.elapsed_time(b) for a, b in evs)
        print("epoch {}: wall {:.1f} ms/batch, host next() {:.2f}, host fit_batch {:.1f}, GPU fit_batch {:.1f}".format(
            ep, wall / nb * 1e3, t_next / nb * 1e3, t_fit / nb * 1e3, gpu / nb))
    loader.reset()
    clones = [{k: v.clone() for k, v in bt.items()} for bt in loader]
    loader.close()
    torch.cuda.synchronize()
    evs, w0, t_fit = [], time.perf_counter(), 0.0
    for bt in clones:
        b = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fit_batch(mod, bt, upd, 1e-5)
        e1.record()
        t_fit += time.perf_counter() - b
        evs.append((e0, e1))
    torch.cuda.synchronize()
    print("private copies: wall {:.1f} ms/batch, host fit_batch {:.1f}, GPU fit_batch {:.1f}".format(
        (time.perf_counter() - w0) / nb * 1e3, t_fit / nb * 1e3, sum(a.elapsed_time(b) for a, b in evs) / nb))
finally:
    shutil.rmtree(root, ignore_errors=True)
