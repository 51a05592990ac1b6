"""Read the per-segment cycle sums of the stamped diagnostic build of wino_gemm_split_kernel (tools/split_exp.sh 128)."""
import ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "gpurun_exp", "libdeepim_hip_exp{}.so".format(sys.argv[1] if len(sys.argv) > 1 else "128"))
os.environ["DIM_HIP_LIB"] = LIB
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import numpy as np, torch
from lib.hip import ops

NAMES = ["U3 issue + A frags (wait)", "stage (early)", "48 MFMAs (issue)", "flush", "stage (late)", "barrier", "-", "-"]
for N, H, W, Cin, Cout, tile, m in [(16, 60, 80, 256, 256, 5, 4), (16, 30, 40, 512, 512, 5, 4), (16, 120, 160, 32, 128, 4, 4)]:
    x = torch.randn((N, H, W, Cin), device="cuda:0")
    w = torch.randn((Cout, Cin, 3, 3), device="cuda:0") * 0.02
    b = torch.zeros(Cout, device="cuda:0")
    wp = ops.winograd_pack_weight(w, m=m)
    for _ in range(4):
        ev = []
        ops.conv2d_fwd_winograd(x, Cin, wp, b, Cout, slope=1.0, tile=tile, m=m, events=ev)
    torch.cuda.synchronize()
    us = [s.elapsed_time(e) * 1e3 for k, s, e in ev if k == "conv"][0]
    n = 1024
    buf = (ctypes.c_ulonglong * (n * 8 * 11))()
    rc = ops.lib().dim_debug_split_stamps(buf, n)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8, 11).astype(np.float64)
    a = a[a[:, 0, 10] > 0]
    nw = int((a[0, :, 10] > 0).sum())
    a = a[:, :nw]
    per = a[:, :, :8] / a[:, :, 9:10]
    tot = a[:, :, 8] / a[:, :, 9]
    print("N{} {}x{} {}->{} tile {}: {:.1f} us (stamped build), {} workgroups of {} waves, {:.1f} chunks each, {:.0f} cycles per chunk".format(
        N, H, W, Cin, Cout, tile, us, len(a), nw, a[:, 0, 9].mean(), np.median(tot)))
    print("   {:<36} ".format("segment (median cycles per chunk)") + " ".join("  wave{}".format(w) for w in range(nw)))
    for k, nm in enumerate(NAMES):
        print("   {:<36} ".format(nm) + " ".join("{:7.0f}".format(np.median(per[:, w, k])) for w in range(nw)))
