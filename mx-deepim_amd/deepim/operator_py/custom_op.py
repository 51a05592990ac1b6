"""Minimal stand-in for the mx.operator plugin API the reference's custom ops sit behind
(mx.operator.CustomOp / CustomOpProp / register and mx.sym.Custom, used at
/root/reference/deepim/operator_py/*.py and deepim/symbols/deepIM_flownet.py:595-665).

Same protocol, torch CUDA tensors instead of NDArrays:
  Prop:  __init__(**attrs as strings), list_arguments(), list_outputs(), infer_shape(in_shape),
         infer_type(in_type), create_operator(ctx, shapes, dtypes)
  Op:    forward(is_train, req, in_data, out_data, aux), backward(req, out_grad, in_data, out_data, in_grad, aux),
         assign(dst, req, src) with req in {"null", "write", "inplace", "add"}
`Custom(op_type=..., **kwargs)` runs an op imperatively (what mx.nd.Custom does).
"""
import numpy as np
import torch

_REGISTRY = {}


class CustomOp(object):
    def forward(self, is_train, req, in_data, out_data, aux):
        raise NotImplementedError

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        raise NotImplementedError

    def assign(self, dst, req, src):
        """mx.operator.CustomOp.assign; `src` may alias `dst` (kernels write in place)."""
        if req == "null":
            return
        if req in ("write", "inplace"):
            if isinstance(src, (int, float)):
                dst.fill_(src)
            elif src is not dst:
                dst.copy_(src)
        elif req == "add":
            dst.add_(src)
        else:
            raise ValueError("unknown req {}".format(req))


class CustomOpProp(object):
    def __init__(self, need_top_grad=True):
        self.need_top_grad_ = need_top_grad

    def list_arguments(self):
        return ["data"]

    def list_outputs(self):
        return ["output"]

    def list_auxiliary_states(self):
        return []

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[0]], []

    def infer_type(self, in_type):
        return in_type, [in_type[0]] * len(self.list_outputs()), []


def register(reg_name):
    def do_register(prop_cls):
        _REGISTRY[reg_name] = prop_cls
        return prop_cls

    return do_register


def attr_to_str(v):
    """MXNet stringifies every attr; numpy arrays become '[a b c]' (parsed back with np.fromstring)."""
    if isinstance(v, np.ndarray):
        return "[" + " ".join(repr(float(x)) for x in v.flatten()) + "]"
    return str(v)


def parse_array(s, n=None):
    a = np.array([float(x) for x in s[1:-1].replace(",", " ").split()], dtype=np.float32)
    if n is not None:
        a = a.reshape(n)
    return a


_OP_CACHE = {}


def Custom(*args, **kwargs):
    """Imperative mx.nd.Custom: tensors by keyword (argument names) or position, attrs as keywords."""
    op_type = kwargs.pop("op_type")
    kwargs.pop("name", None)
    prop_cls = _REGISTRY[op_type]
    tensors = {k: v for k, v in kwargs.items() if isinstance(v, torch.Tensor)}
    attrs = {k: attr_to_str(v) for k, v in kwargs.items() if not isinstance(v, torch.Tensor)}
    key = (op_type, tuple(sorted(attrs.items())))
    if key not in _OP_CACHE:
        prop = prop_cls(**attrs)
        _OP_CACHE[key] = (prop, None)
    prop, op = _OP_CACHE[key]
    names = prop.list_arguments()
    in_data = list(args) + [tensors[n] for n in names[len(args):]]
    in_shape = [list(t.shape) for t in in_data]
    _, out_shape, _ = prop.infer_shape(in_shape)
    if op is None:
        op = prop.create_operator(in_data[0].device, in_shape, [t.dtype for t in in_data])
        _OP_CACHE[key] = (prop, op)
    out_data = [torch.empty(tuple(s), dtype=torch.float32, device=in_data[0].device) for s in out_shape]
    op.forward(False, ["write"] * len(out_data), in_data, out_data, [])
    return out_data[0] if len(out_data) == 1 else out_data
