"""ZoomFlow custom op on the HIP kernels (reference: deepim/operator_py/zoom_flow.py:20-118).
Output arity changes with b_inv_zoom exactly like the reference Prop (:93-105)."""
import torch

from lib.hip import ops
from .custom_op import CustomOp, CustomOpProp, register


class ZoomFlowOperator(CustomOp):
    def __init__(self, height, width, b_inv_zoom):
        super(ZoomFlowOperator, self).__init__()
        self.height = height
        self.width = width
        self.b_inv_zoom = b_inv_zoom

    def forward(self, is_train, req, in_data, out_data, aux):
        zoom_factor, flow = in_data[0], in_data[1]
        zf_host = zoom_factor[:, :2].cpu()
        assert torch.equal(zf_host[:, 0], zf_host[:, 1]), "wx and wy should be equal"  # (:62)
        # inverse zoom multiplies the values by wx, forward zoom divides (:59-66)
        self.assign(out_data[0], req[0], ops.zoom_planes(flow, zoom_factor, inverse=self.b_inv_zoom, scale_mode=2 if self.b_inv_zoom else 1))
        if not self.b_inv_zoom:
            # round(zoomed weights - 0.45) (:70-77)
            self.assign(out_data[1], req[1], ops.zoom_planes(in_data[2], zoom_factor, post=2))

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        for i in range(len(in_grad)):
            self.assign(in_grad[i], req[i], 0)


@register("ZoomFlow")
class ZoomFlowProp(CustomOpProp):
    def __init__(self, width=640, height=480, b_inv_zoom="False"):
        super(ZoomFlowProp, self).__init__(True)
        self.height = int(height)
        self.width = int(width)
        self.b_inv_zoom = b_inv_zoom.lower() == "true"

    def list_arguments(self):
        return ["zoom_factor", "flow"] if self.b_inv_zoom else ["zoom_factor", "flow", "flow_weights"]

    def list_outputs(self):
        return ["zoom_flow"] if self.b_inv_zoom else ["zoom_flow", "zoom_flow_weights"]

    def infer_shape(self, in_shape):
        return in_shape, in_shape[1:], []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype] * len(in_type), [dtype] * (len(in_type) - 1), []

    def create_operator(self, ctx, shapes, dtypes):
        return ZoomFlowOperator(self.height, self.width, self.b_inv_zoom)
