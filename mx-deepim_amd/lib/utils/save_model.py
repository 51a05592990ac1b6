"""Checkpoint writing with the reference's function name (lib/utils/save_model.py:10-24), on numpy / torch arrays."""
from __future__ import print_function, division

import numpy as np

from lib.utils.mx_params import nd_save


def _host(v):
    return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)


def save_checkpoint(prefix, epoch, arg_params, aux_params):
    """prefix-%04d.params with `arg:` / `aux:` key prefixes, readable by mx.nd.load (reference :10-24)."""
    save_dict = {("arg:%s" % k): _host(v) for k, v in arg_params.items()}
    save_dict.update({("aux:%s" % k): _host(v) for k, v in aux_params.items()})
    param_name = "%s-%04d.params" % (prefix, epoch)
    nd_save(param_name, save_dict)
    return param_name
