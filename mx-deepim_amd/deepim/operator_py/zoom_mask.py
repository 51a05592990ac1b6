"""ZoomMask custom op on the HIP kernels.
Drop-in for /root/reference/deepim/operator_py/zoom_mask.py (ZoomMaskOperator :22-140, ZoomMaskProp :143-177):
same registered name, argument / output names, attrs-as-strings protocol; the numpy bbox code, the per-sample
GridGenerator loop and the BilinearSampler calls are replaced by dim_mask_bbox + dim_zoom_factor + dim_zoom_planes."""
import numpy as np
import torch

from lib.hip import ops
from .custom_op import CustomOp, CustomOpProp, parse_array, register


class ZoomMaskOperator(CustomOp):
    def __init__(self, K, height, width):
        super(ZoomMaskOperator, self).__init__()
        self.K = K
        self.height = height
        self.width = width

    def forward(self, is_train, req, in_data, out_data, aux):
        mask_real_est, mask_real_gt, mask_rendered, src_pose = in_data
        bo = ops.mask_bbox(mask_real_gt, 0.3)          # valid_real = sum(mask_gt, axis=1) > 0.3      (:35-37)
        br = ops.mask_bbox(mask_rendered, 0.2)         # rendered binarised at 0.2, then > 0.3        (:39-47)
        status = torch.zeros(src_pose.shape[0], dtype=torch.int32, device=src_pose.device)
        zf = ops.zoom_factor(bo, br, src_pose, self.K, self.height, self.width, status=status)
        st = status.cpu().numpy()
        if (st & 1).any():
            # reference: np.min(nz_x) of an empty array (:58)
            raise ValueError("zero-size array to reduction operation minimum which has no identity (empty observed mask)")
        for b in np.nonzero(st & 2)[0]:
            print("NO POINT VALID IN MASK rendered")    # (:75)
        outs = [ops.zoom_planes(mask_real_est, zf, post=1), ops.zoom_planes(mask_real_gt, zf, post=1),
                ops.zoom_planes(mask_rendered, zf, pre=1, post=1), zf]
        for i in range(4):
            self.assign(out_data[i], req[i], outs[i])

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        for i in range(4):
            self.assign(in_grad[i], req[i], 0)


@register("ZoomMask")
class ZoomMaskProp(CustomOpProp):
    def __init__(self, K, width=640, height=480):
        super(ZoomMaskProp, self).__init__(True)
        self.K = parse_array(K, (3, 3))
        self.height = int(height)
        self.width = int(width)

    def list_arguments(self):
        return ["mask_observed", "mask_gt_observed", "mask_rendered", "src_pose"]

    def list_outputs(self):
        return ["zoom_mask_observed", "zoom_mask_gt_observed", "zoom_mask_rendered", "zoom_factor"]

    def infer_shape(self, in_shape):
        batch_size = in_shape[0][0]
        out_shape = in_shape[:-1]
        out_shape.append([batch_size, 4])
        return in_shape, out_shape, []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype] * 4, [dtype] * 4, []

    def create_operator(self, ctx, shapes, dtypes):
        return ZoomMaskOperator(self.K, self.height, self.width)
