import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import torch.nn.functional as F
from scene import make_train_config, make_train_scene
from deepim.symbols.deepIM_flownet import deepIM_flownet
from deepim.core.module import MutableModule
from lib.hip import ops
cfg = make_train_config()
sym = deepIM_flownet(); sym.get_symbol(cfg, True)
params = sym.init_weights(cfg, {}, {}, seed=0)
rng = np.random.RandomState(1)
params["trans_weight"] = (rng.randn(3, 256) * 0.002).astype(np.float32)
scene = make_train_scene(B=2, seed=99, subdiv=3)
mod = MutableModule(cfg, params, 2)
batch = {k: torch.as_tensor(np.ascontiguousarray(v)).cuda() for k, v in scene["blobs"].items()}
mod.forward_backward(batch)
net, w = mod.net, mod.w
B = 2
def rel(a, b): return float((a - b).abs().max() / (b.abs().max() + 1e-30))
# fc6 dgrad alone
d = torch.zeros((B, 1, 1, 81920), device="cuda")
ops.conv2d_fwd_ex(mod.dz6.view(B, 1, 1, 256), 0, 256, mod.dgrad_packed["fc6"], None, d, 0, 81920, 1, 1, 1, 0, accumulate=True)
ref = (mod.dz6.double() @ w["fc6_weight"].double()).view(B, 1024, 8, 10).permute(0, 2, 3, 1).reshape(B, 1, 1, 81920).float()
print("fc6 dgrad rel", rel(d, ref))
# conv1 dX alone
d1 = torch.zeros((B, 8, 10, 1024), device="cuda")
ops.conv_small_cout_bwd(net.acts["conv6_1"], 1024, mod.dflow6, w["Convolution1_weight"], d1, torch.empty_like(w["Convolution1_weight"]), torch.empty(2, device="cuda"))
ref1 = F.conv_transpose2d(mod.dflow6.permute(0, 3, 1, 2).double(), w["Convolution1_weight"].double(), padding=1).permute(0, 2, 3, 1).float()
print("conv1 dX rel", rel(d1, ref1))
# deconv5 dgrad alone
d5 = torch.zeros((B, 8, 10, 1024), device="cuda")
ops.conv2d_fwd_ex(mod.dconcat2, 512, 512, mod.dgrad_packed["deconv5"], None, d5, 0, 1024, 4, 4, 2, 1, Ho=8, Wo=10)
dz5 = mod.dconcat2[..., 512:1024].permute(0, 3, 1, 2).double()
full = torch.zeros((B, 512, 18, 22), dtype=torch.float64, device="cuda"); full[:, :, 1:16, 1:21] = dz5
ref5 = F.conv2d(full, w["deconv5_weight"].double(), stride=2).permute(0, 2, 3, 1).float()
print("deconv5 dgrad rel", rel(d5, ref5), ref5.shape)
# encoder dgrad conv6_1 (stride 1) and conv6 (stride 2) from the stored dz
for name, prev, cout, cin, k, s, p in (("conv6_1", "conv6", 1024, 1024, 3, 1, 1), ("conv6", "conv5_1", 1024, 512, 3, 2, 1), ("conv5_1", "conv5", 512, 512, 3, 1, 1)):
    dz = mod.dacts[name]
    x = net.acts[prev]
    out = torch.zeros_like(x)
    ops.conv2d_dgrad(dz, cout, mod.dgrad_packed[name], out, cin, k, k, s, p)
    xx = x.permute(0, 3, 1, 2).double().requires_grad_()
    y = F.conv2d(xx, w[name + "_weight"].double(), None, stride=s, padding=p)
    y.backward(dz.permute(0, 3, 1, 2).double())
    print(name, "dgrad rel", rel(out, xx.grad.permute(0, 2, 3, 1).float()))
    gp = torch.zeros_like(net.packed[name])
    ops.conv2d_wgrad(x, cin, dz, cout, k, k, s, p, gp, splits=mod.wgrad_splits[name], workspace=mod.ws)
    dw = torch.empty_like(w[name + "_weight"]); ops.conv2d_unpack_weight(gp, dw)
    wref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), w[name + "_weight"].shape, dz.permute(0, 3, 1, 2).double(), stride=s, padding=p)
    print(name, "wgrad rel", rel(dw, wref.float()), "splits", mod.wgrad_splits[name])
