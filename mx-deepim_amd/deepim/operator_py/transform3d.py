"""Transform3D custom op on the HIP kernels (reference: deepim/operator_py/transform3d.py:30-362).
forward: P' = R_tgt P + T_tgt with the per-sample target pose composed on device; backward: d_rot (B,4), d_trans (B,3)."""
from lib.hip import ops
from .custom_op import CustomOp, CustomOpProp, parse_array, register


class transform3dOperator(CustomOp):
    def __init__(self, T_means=None, T_stds=None, rot_coord="MODEL", projection_2d=False):
        super(transform3dOperator, self).__init__()
        self.T_means = T_means
        self.T_stds = T_stds
        self._projection_2d = projection_2d
        self.rot_coord = rot_coord
        assert not projection_2d, "NOT_IMPLEMENTED"

    def forward(self, is_train, req, in_data, out_data, aux):
        points, rotation, T_delta, pose_src = in_data
        batch_size = points.shape[0]
        assert rotation.shape[0] == batch_size and T_delta.shape[0] == batch_size, \
            "rotation.shape[0]:{} vs batch_size:{}, translation.shape[0]:{} vs batch_size:{}".format(
                rotation.shape[0], batch_size, T_delta.shape[0], batch_size)
        if rotation.shape[1] == 3:
            raise Exception("NOT_IMPLEMENTED")
        if rotation.shape[1] != 4:
            raise Exception("UNKNOWN ROTATION REPRESENTATION {}".format(rotation.shape[1]))
        out = ops.transform3d_fwd(points.reshape(batch_size, 3, -1).contiguous(), rotation, T_delta, pose_src, self.rot_coord,
                                  self.T_means, self.T_stds)
        self.assign(out_data[0], req[0], out.reshape(points.shape))

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        points, rotation, T_delta, pose_src = in_data
        B = points.shape[0]
        d_rot, d_trans = ops.transform3d_bwd(out_grad[0].reshape(B, 3, -1).contiguous(), points.reshape(B, 3, -1).contiguous(), rotation,
                                             T_delta, pose_src, self.rot_coord, self.T_means, self.T_stds)
        self.assign(in_grad[0], req[0], 0)
        self.assign(in_grad[1], req[1], d_rot)
        self.assign(in_grad[2], req[2], d_trans)
        self.assign(in_grad[3], req[3], 0)


@register("Transform3D")
class transform3dProp(CustomOpProp):
    def __init__(self, T_means=None, T_stds=None, rot_coord="MODEL", b_project_2d="False"):
        super(transform3dProp, self).__init__(True)
        self.T_means = parse_array(T_means, 3)
        self.T_stds = parse_array(T_stds, 3)
        self._b_project_2d = b_project_2d.lower() in ("true", "1", "yes", "y", "t", "on")
        self.rot_coord = rot_coord

    def list_arguments(self):
        return ["point_cloud", "rotation", "translation", "pose_src"]

    def list_outputs(self):
        return ["transformed_3d_points"]

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[0]], []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype] * 4, [dtype], []

    def create_operator(self, ctx, shapes, dtypes):
        return transform3dOperator(self.T_means, self.T_stds, self.rot_coord, self._b_project_2d)
