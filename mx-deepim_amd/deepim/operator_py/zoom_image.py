"""ZoomImage custom op (no-mask configs) on the HIP kernels.
Drop-in for /root/reference/deepim/operator_py/zoom_image.py (:19-161): validity = sum_c(image + mean) > 0.01."""
import numpy as np
import torch

from lib.hip import ops
from .custom_op import CustomOp, CustomOpProp, parse_array, register


class ZoomImageOperator(CustomOp):
    def __init__(self, K, height, width, pixel_means):
        super(ZoomImageOperator, self).__init__()
        self.K = K
        self.height = height
        self.width = width
        self.pixel_means = np.asarray(pixel_means, dtype=np.float32).reshape(3)

    def forward(self, is_train, req, in_data, out_data, aux):
        image_real, image_rendered, src_pose = in_data
        bo = ops.mask_bbox(image_real, 0.01, mode=1, means3=self.pixel_means)
        br = ops.mask_bbox(image_rendered, 0.01, mode=1, means3=self.pixel_means)
        status = torch.zeros(src_pose.shape[0], dtype=torch.int32, device=src_pose.device)
        zf = ops.zoom_factor(bo, br, src_pose, self.K, self.height, self.width, status=status)
        st = status.cpu().numpy()
        if (st & 1).any():
            raise ValueError("zero-size array to reduction operation minimum which has no identity (empty observed image)")
        for b in np.nonzero(st & 2)[0]:
            print("NO POINT VALID IN rendered")
        self.assign(out_data[0], req[0], ops.zoom_planes(image_real, zf, add3=self.pixel_means))
        self.assign(out_data[1], req[1], ops.zoom_planes(image_rendered, zf, add3=self.pixel_means))
        self.assign(out_data[2], req[2], zf)

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        for i in range(3):
            self.assign(in_grad[i], req[i], 0)


@register("ZoomImage")
class ZoomImageProp(CustomOpProp):
    def __init__(self, K, width=640, height=480, pixel_means="[0 0 0]"):
        super(ZoomImageProp, self).__init__(True)
        self.K = parse_array(K, (3, 3))
        self.height = int(height)
        self.width = int(width)
        self.pixel_means = parse_array(pixel_means, 3)[::-1].copy()

    def list_arguments(self):
        return ["image_observed", "image_rendered", "src_pose"]

    def list_outputs(self):
        return ["zoom_image_observed", "zoom_image_rendered", "zoom_factor"]

    def infer_shape(self, in_shape):
        batch_size = in_shape[0][0]
        return in_shape, [in_shape[0], in_shape[1], [batch_size, 4]], []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype] * 3, [dtype] * 3, []

    def create_operator(self, ctx, shapes, dtypes):
        return ZoomImageOperator(self.K, self.height, self.width, self.pixel_means)
