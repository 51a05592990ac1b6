"""GEMM launch times of the split plane-GEMM kernel for the libraries tools/split_exp.sh built (one child process per library)."""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import numpy as np, torch
from lib.hip import ops
torch.manual_seed(0)
out = []
for N, H, W, Cin, Cout, tile, m in [(16, 60, 80, 256, 256, 5, 4), (16, 30, 40, 512, 512, 5, 4), (16, 120, 160, 32, 128, 4, 4), (16, 8, 10, 1024, 1024, 7, 4)]:
    x = torch.randn((N, H, W, Cin), device="cuda:0")
    w = torch.randn((Cout, Cin, 3, 3), device="cuda:0") * 0.02
    b = torch.zeros(Cout, device="cuda:0")
    wp = ops.winograd_pack_weight(w, m=m)
    ts = []
    for _ in range(6):
        ev = []
        ops.conv2d_fwd_winograd(x, Cin, wp, b, Cout, slope=1.0, tile=tile, m=m, events=ev)
        torch.cuda.synchronize()
        ts.append([s.elapsed_time(e) * 1e3 for k, s, e in ev if k == "conv"][0])
    out.append("{:.1f}".format(min(ts[2:])))
print(" ".join(out))
'''
for v in sys.argv[1:]:
    env = dict(os.environ)
    if v != "lib" and not os.environ.get("DIM_HIP_LIB_KEEP"):
        env["DIM_HIP_LIB"] = os.path.join(ROOT, "gpurun_exp", "libdeepim_hip_exp{}.so".format(v))
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=300)
    print("exp {:>4}: {}".format(v, r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED " + r.stderr[-400:]), flush=True)
