"""depth -> flow labels, reference Python face on the HIP kernel.

Drop-in for /root/reference/lib/flow_c/flow.py:17-21 (`gpu_flow_wrapper`) and the Cython binding
lib/flow_c/gpu_flow.pyx:24-41 (`gpu_flow`): C-contiguous float32 host arrays in, (flow[N,2,H,W], valid[N,1,H,W]) out.
`gpu_flow_device` is the device-resident form the training updater uses (no H2D/D2H, no malloc per call -- the
reference's _flow does cudaMalloc + 4 copies + cudaFree every call: gpu_flow_kernel.cu:94-147).
"""
import numpy as np
import torch

from lib.hip import ops


def gpu_flow_device(depth_src, depth_tgt, KT, Kinv, flow=None, valid=None):
    """CUDA tensors in/out: depth (N,1,H,W), KT (N,3,4); Kinv host 3x3."""
    return ops.depth_to_flow(depth_src, depth_tgt, KT, Kinv, flow=flow, valid=valid)


def gpu_flow(depth_src, depth_tgt, KT, Kinv, device_id=0):
    for name, a, nd in (("depth_src", depth_src, 4), ("depth_tgt", depth_tgt, 4), ("KT", KT, 3), ("Kinv", Kinv, 2)):
        if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.ndim == nd):
            raise ValueError("Buffer dtype mismatch or wrong number of dimensions for {} (expected float32, ndim {})".format(name, nd))
    dev = torch.device("cuda", device_id)
    if depth_src.shape[0] == 0:
        n, _, h, w = depth_src.shape
        return np.zeros([n, 2, h, w], dtype=np.float32), np.zeros([n, 1, h, w], dtype=np.float32)
    with torch.cuda.device(dev):
        flow, valid = ops.depth_to_flow(torch.from_numpy(np.ascontiguousarray(depth_src)).to(dev),
                                        torch.from_numpy(np.ascontiguousarray(depth_tgt)).to(dev),
                                        torch.from_numpy(np.ascontiguousarray(KT)).to(dev), Kinv)
        return flow.cpu().numpy(), valid.cpu().numpy()


def gpu_flow_wrapper(device_id):
    def _flow(depth_src, depth_tgt, KT, Kinv):
        return gpu_flow(depth_src, depth_tgt, KT, Kinv, device_id)

    return _flow
