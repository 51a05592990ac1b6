"""ZoomTrans custom op on the HIP kernels (reference: deepim/operator_py/zoom_trans.py:15-105)."""
import torch

from lib.hip import ops
from .custom_op import CustomOp, CustomOpProp, register


class ZoomTransOperator(CustomOp):
    def __init__(self, b_inv_zoom, b_zoom_grad):
        super(ZoomTransOperator, self).__init__()
        self.b_inv_zoom = b_inv_zoom
        self.b_zoom_grad = b_zoom_grad

    def forward(self, is_train, req, in_data, out_data, aux):
        zoom_factor, trans_delta = in_data
        # zoom back multiplies (dx,dy) by wx, zoom in divides (:37-45); dz passes through
        self.assign(out_data[0], req[0], ops.zoom_trans(zoom_factor, trans_delta, 2 if self.b_inv_zoom else 1))

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        mode = 0 if not self.b_zoom_grad else (2 if self.b_inv_zoom else 1)  # (:64-72)
        self.assign(in_grad[0], req[0], 0)
        self.assign(in_grad[1], req[1], ops.zoom_trans(in_data[0], out_grad[0], mode))


@register("ZoomTrans")
class ZoomTransProp(CustomOpProp):
    def __init__(self, b_inv_zoom="False", b_zoom_grad="False"):
        super(ZoomTransProp, self).__init__(True)
        self.b_inv_zoom = b_inv_zoom.lower() == "true"
        self.b_zoom_grad = b_zoom_grad.lower() == "true"

    def list_arguments(self):
        return ["zoom_factor", "trans_delta"]

    def list_outputs(self):
        return ["zoom_trans_delta"]

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[1]], []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype] * 2, [dtype], []

    def create_operator(self, ctx, shapes, dtypes):
        return ZoomTransOperator(self.b_inv_zoom, self.b_zoom_grad)
