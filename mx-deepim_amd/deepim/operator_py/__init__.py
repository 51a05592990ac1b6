"""Custom ops of the DeepIM graph, same registered names as /root/reference/deepim/operator_py/."""
from .custom_op import Custom, CustomOp, CustomOpProp, register  # noqa: F401
from . import transform3d, zoom_depth, zoom_flow, zoom_image, zoom_image_with_factor, zoom_mask, zoom_mask_with_factor, zoom_trans  # noqa: F401
