"""Checkpoint reading behind the reference's function names (lib/utils/load_model.py:10-67), on host numpy arrays.

A checkpoint is `<prefix>-<epoch:04d>.params`: an MXNet NDArray list whose keys are `arg:<name>` / `aux:<name>`
(parsed without mxnet by lib/utils/mx_params.py).  Semantics kept from the reference:
  load_checkpoint -> (arg_params, aux_params); keys with another prefix are dropped silently
  load_param(process=True) strips the `_test` / `_i2r` markers from argument names (test-time renaming, :61-67)
  convert_context  is where MXNet moves arrays to a device; the HIP executors upload when they pack, so it is a shallow copy
"""
from __future__ import print_function, division

from lib.utils.mx_params import nd_load

_RENAME_MARKERS = ("_test", "_i2r")  # applied in this order, like the reference's two passes


def checkpoint_path(prefix, epoch):
    return "{}-{:04d}.params".format(prefix, int(epoch))


def load_checkpoint(prefix, epoch):
    groups = {"arg": {}, "aux": {}}
    for key, array in nd_load(checkpoint_path(prefix, epoch)).items():
        kind, _, name = key.partition(":")
        if kind in groups:
            groups[kind][name] = array
    return groups["arg"], groups["aux"]


def convert_context(params, ctx):
    return dict(params)


def _strip_marker(params, marker):
    """rename every key containing `marker` to the key without it (the renamed entry replaces a plain one of the same name)"""
    for old in [k for k in params if marker in k]:
        params[old.replace(marker, "")] = params.pop(old)


def load_param(prefix, epoch, convert=False, ctx=None, process=False):
    arg_params, aux_params = load_checkpoint(prefix, epoch)
    if convert:
        arg_params, aux_params = convert_context(arg_params, ctx), convert_context(aux_params, ctx)
    if process:
        for marker in _RENAME_MARKERS:
            _strip_marker(arg_params, marker)
    return arg_params, aux_params
