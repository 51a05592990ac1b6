"""Time fc6's training-side kernels at B pairs.  usage: fc6_time.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import torch  # noqa: E402

from lib.hip import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.randn((B, 8, 10, 1024), device="cuda:0")
dz = torch.randn((B, 256), device="cuda:0")
dW = torch.empty((256, 81920), device="cuda:0")


def timed(fn, n=30):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("fc_wgrad_nhwc      %.1f us" % timed(lambda: ops.fc_wgrad_nhwc(dz, x, dW)))
gp = torch.empty(256 * 81920, device="cuda:0")
print("wgrad bf16 + unpack %.1f us" % timed(lambda: (ops.conv2d_wgrad(x, 1024, dz.view(B, 1, 1, 256), 256, 8, 10, 1, 0, gp, bf16_mfma=True),
                                                      ops.fc_unpack_weight(gp, dW, 1024, 8, 10))))
