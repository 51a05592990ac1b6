"""Time the flow / mask head convolutions (dim_conv_small_cout_fwd) at B pairs.  usage: small_cout_time.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mx-deepim_amd")]
import torch  # noqa: E402

from lib.hip import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for name, cout, cin, cpad, h, w in (("Convolution3", 2, 770, 832, 30, 40), ("mask_conv3", 1, 770, 832, 30, 40), ("Convolution2", 2, 1026, 1088, 15, 20),
                                    ("Convolution1", 2, 1024, 1024, 8, 10)):
    x = torch.randn((B, h, w, cpad), device="cuda:0")
    wp = ops.conv_small_cout_pack_weight(torch.randn((cout, cin, 3, 3), device="cuda:0") * 0.01)
    b = torch.zeros(cout, device="cuda:0")
    y = ops.conv_small_cout_fwd(x, cin, wp, b, cout)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.conv_small_cout_fwd(x, cin, wp, b, cout, out=y)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("{:14s} {:6.1f} us  {:5.2f} TB/s of the activation".format(name, us, B * h * w * cpad * 4 / us / 1e6))
