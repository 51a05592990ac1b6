"""Checkpoint loading with the reference's function names (lib/utils/load_model.py:10-67), on numpy arrays.

`prefix-%04d.params` files are MXNet NDArray lists with `arg:` / `aux:` key prefixes; parsed by lib/utils/mx_params.py."""
from __future__ import print_function, division

from lib.utils.mx_params import nd_load


def load_checkpoint(prefix, epoch):
    """-> (arg_params, aux_params): dicts of name -> numpy array  (reference :10-30)"""
    save_dict = nd_load("%s-%04d.params" % (prefix, epoch))
    arg_params = {}
    aux_params = {}
    for k, v in save_dict.items():
        tp, name = k.split(":", 1)
        if tp == "arg":
            arg_params[name] = v
        if tp == "aux":
            aux_params[name] = v
    return arg_params, aux_params


def convert_context(params, ctx):
    """reference :33-42.  Arrays stay on the host until the executor packs them for the device; ctx is accepted and ignored."""
    return dict(params)


def load_param(prefix, epoch, convert=False, ctx=None, process=False):
    """wrapper for load_checkpoint (reference :45-67): `process` renames `*_test` and `*_i2r` parameters to their plain names."""
    arg_params, aux_params = load_checkpoint(prefix, epoch)
    if convert:
        arg_params = convert_context(arg_params, ctx)
        aux_params = convert_context(aux_params, ctx)
    if process:
        tests = [k for k in arg_params.keys() if "_test" in k]
        for test in tests:
            arg_params[test.replace("_test", "")] = arg_params.pop(test)
        i2rs = [k for k in arg_params.keys() if "_i2r" in k]
        for i2r in i2rs:
            arg_params[i2r.replace("_i2r", "")] = arg_params.pop(i2r)
    return arg_params, aux_params
