"""ZoomMaskWithFactor custom op on the HIP kernels (reference: deepim/operator_py/zoom_mask_with_factor.py:22-110)."""
from lib.hip import ops
from .custom_op import CustomOp, CustomOpProp, register


class ZoomMaskWithFactorOperator(CustomOp):
    def __init__(self, height, width, b_inv_zoom):
        super(ZoomMaskWithFactorOperator, self).__init__()
        self.height = height
        self.width = width
        self.b_inv_zoom = b_inv_zoom

    def forward(self, is_train, req, in_data, out_data, aux):
        zoom_factor, mask = in_data
        # binarise at 0.2 (:35-38), sample forward / inverse (:41-63), mx.nd.round (:66)
        self.assign(out_data[0], req[0], ops.zoom_planes(mask, zoom_factor, inverse=self.b_inv_zoom, pre=1, post=1))

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        self.assign(in_grad[0], req[0], 0)
        self.assign(in_grad[1], req[1], 0)


@register("ZoomMaskWithFactor")
class ZoomMaskWithFactorProp(CustomOpProp):
    def __init__(self, width=640, height=480, b_inv_zoom="False"):
        super(ZoomMaskWithFactorProp, self).__init__(True)
        self.height = int(height)
        self.width = int(width)
        self.b_inv_zoom = b_inv_zoom.lower() == "true"

    def list_arguments(self):
        return ["zoom_factor", "mask"]

    def list_outputs(self):
        return ["zoom_mask"]

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[1]], []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype] * 2, [dtype], []

    def create_operator(self, ctx, shapes, dtypes):
        return ZoomMaskWithFactorOperator(self.height, self.width, self.b_inv_zoom)
