"""ZoomImageWithFactor custom op on the HIP kernels.
Drop-in for /root/reference/deepim/operator_py/zoom_image_with_factor.py (:20-120).  `pixel_means` arrives as the
config's PIXEL_MEANS and is reversed like the reference Prop does (:94); the optional red centre spot
(`high_light_center`, off in every shipped config) is applied with torch.maximum like :66-69."""
import numpy as np
import torch

from lib.hip import ops
from .custom_op import CustomOp, CustomOpProp, parse_array, register


class ZoomImageWithFactorOperator(CustomOp):
    def __init__(self, height, width, pixel_means, high_light_center):
        super(ZoomImageWithFactorOperator, self).__init__()
        self.height = height
        self.width = width
        self.pixel_means = np.asarray(pixel_means, dtype=np.float32).reshape(3)
        self.high_light_center = high_light_center
        self.center_map = None
        self.spot_radius = 5

    def forward(self, is_train, req, in_data, out_data, aux):
        zoom_factor, image_real, image_rendered = in_data
        zoom_real = ops.zoom_planes(image_real, zoom_factor, add3=self.pixel_means)
        zoom_rendered = ops.zoom_planes(image_rendered, zoom_factor, add3=self.pixel_means)
        if self.high_light_center:
            if self.center_map is None:
                B = image_real.shape[0]
                cm = np.zeros([B, 3, self.height, self.width], dtype=np.float32)
                sx, ex = int(np.floor(self.width / 2.0 - self.spot_radius)), int(np.ceil(self.width / 2.0 + self.spot_radius))
                sy, ey = int(np.floor(self.height / 2.0 - self.spot_radius)), int(np.ceil(self.height / 2.0 + self.spot_radius))
                cm[:, 0, sy:ey, sx:ex] = 255.0
                self.center_map = torch.from_numpy(cm).to(image_real.device)
            pm = torch.from_numpy(self.pixel_means).to(image_real.device).view(1, 3, 1, 1)
            zoom_rendered = torch.maximum(zoom_rendered + pm, self.center_map) - pm
        self.assign(out_data[0], req[0], zoom_real)
        self.assign(out_data[1], req[1], zoom_rendered)

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        for i in range(3):
            self.assign(in_grad[i], req[i], 0)


@register("ZoomImageWithFactor")
class ZoomImageWithFactorProp(CustomOpProp):
    def __init__(self, width=640, height=480, pixel_means="[0 0 0]", high_light_center="False"):
        super(ZoomImageWithFactorProp, self).__init__(True)
        self.height = int(height)
        self.width = int(width)
        self.pixel_means = parse_array(pixel_means, 3)[::-1].copy()
        self.hight_light_center = high_light_center.lower() == "true"

    def list_arguments(self):
        return ["zoom_factor", "image_observed", "image_rendered"]

    def list_outputs(self):
        return ["zoom_image_observed", "zoom_image_rendered"]

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[1], in_shape[2]], []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype] * 3, [dtype] * 2, []

    def create_operator(self, ctx, shapes, dtypes):
        return ZoomImageWithFactorOperator(self.height, self.width, self.pixel_means, self.hight_light_center)
