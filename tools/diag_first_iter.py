"""Does the first forward_backward of a fresh MutableModule differ from the next ones on the same batch?  (uninitialised state hunt)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from test_gpu_baseline_configs import make_train_config, make_train_scene, _head_params, DEV
from deepim.core.module import MutableModule
from deepim.symbols.deepIM_flownet import deepIM_flownet
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
cfg = make_train_config()
cfg.dataset.class_name = ["ape", "can", "cat"]
sym = deepIM_flownet(); sym.get_symbol(cfg, is_train=True)
params = _head_params(sym, cfg, 0)
params["mask_conv3_weight"] = (np.random.RandomState(3).randn(1, 770, 3, 3) * 0.02).astype(np.float32)
B = 16
scene = make_train_scene(B=B, seed=777, subdiv=3, n_models=3)
dev = {k: torch.as_tensor(np.ascontiguousarray(v)).to(DEV) for k, v in scene["blobs"].items()}
mod = MutableModule(cfg, params, B, compute_dtype=dtype)
gs = []
for it in range(3):
    mod.forward_backward(dev)
    gs.append({k: v.copy() for k, v in mod.get_grads().items()})
for a, b in ((0, 1), (1, 2)):
    worst = sorted(((float(np.linalg.norm((gs[a][k] - gs[b][k]).ravel()) / (np.linalg.norm(gs[b][k].ravel()) + 1e-30)), k) for k in gs[0]), reverse=True)[:6]
    print("run %d vs %d:" % (a, b), ["%s %.2e" % (k, v) for v, k in worst])
perm = torch.as_tensor(np.random.RandomState(5).permutation(B), device=DEV)
mod.forward_backward({k: v[perm].contiguous() for k, v in dev.items()})
gp = {k: v.copy() for k, v in mod.get_grads().items()}
worst = sorted(((float(np.linalg.norm((gp[k] - gs[2][k]).ravel()) / (np.linalg.norm(gs[2][k].ravel()) + 1e-30)), k) for k in gp), reverse=True)[:6]
print("permuted vs run 2:", ["%s %.2e" % (k, v) for v, k in worst])
mod.forward_backward(dev)
g4 = mod.get_grads()
worst = sorted(((float(np.linalg.norm((g4[k] - gs[2][k]).ravel()) / (np.linalg.norm(gs[2][k].ravel()) + 1e-30)), k) for k in g4), reverse=True)[:3]
print("original again vs run 2:", ["%s %.2e" % (k, v) for v, k in worst])
# forward activations: original vs permuted batch (rows brought back into the original order)
from deepim.symbols.deepIM_flownet import ENCODER
mod.forward(dev)
a0 = {n: mod.net.acts[n].clone() for n, *_ in ENCODER}
mod.forward({k: v[perm].contiguous() for k, v in dev.items()})
inv = torch.argsort(perm)
for n, *_ in ENCODER:
    a1 = mod.net.acts[n][inv]
    d = (a1 - a0[n]).abs()
    nz = int((d > 0).sum())
    sign = int(((a1 > 0) != (a0[n] > 0)).sum())
    print("%-8s max|diff| %.3e (max|act| %.2e)  differing %d of %d  sign flips %d" % (n, d.max().item(), a0[n].abs().max().item(), nz, d.numel(), sign))
