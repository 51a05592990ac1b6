"""ZoomDepth custom op on the HIP kernels (reference: deepim/operator_py/zoom_depth.py:18-84)."""
from lib.hip import ops
from .custom_op import CustomOp, CustomOpProp, register


class ZoomDepthOperator(CustomOp):
    def __init__(self, height, width):
        super(ZoomDepthOperator, self).__init__()
        self.height = height
        self.width = width

    def forward(self, is_train, req, in_data, out_data, aux):
        zoom_factor, depth_real, depth_rendered = in_data
        for wx, wy, tx, ty in zoom_factor.cpu().tolist():
            print("wx: {}, wy: {}, tx: {}, ty: {}".format(wx, wy, tx, ty))  # the reference prints per sample (:32)
        self.assign(out_data[0], req[0], ops.zoom_planes(depth_real, zoom_factor))
        self.assign(out_data[1], req[1], ops.zoom_planes(depth_rendered, zoom_factor))

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        for i in range(3):
            self.assign(in_grad[i], req[i], 0)


@register("ZoomDepth")
class ZoomDepthProp(CustomOpProp):
    def __init__(self, width=640, height=480):
        super(ZoomDepthProp, self).__init__(True)
        self.height = int(height)
        self.width = int(width)

    def list_arguments(self):
        return ["zoom_factor", "depth_observed", "depth_rendered"]

    def list_outputs(self):
        return ["zoom_depth_observed", "zoom_depth_rendered"]

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[1], in_shape[2]], []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype] * 3, [dtype] * 2, []

    def create_operator(self, ctx, shapes, dtypes):
        return ZoomDepthOperator(self.height, self.width)
