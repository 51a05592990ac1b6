"""ctypes face of the resident-loop C entry points (include/deepim_hip.h: dim_refiner_create / _run / _destroy).

This is what a host WITHOUT torch binds (INTEGRATION.md shows the same struct for C / cgo callers); torch is used here only to own
the device memory of the arguments, exactly as in the rest of lib/hip."""
import ctypes

import numpy as np
import torch

from . import capi
from .capi import check, current_stream, dptr, lib


class RefinerDesc(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("test_iter", ctypes.c_int), ("K9", ctypes.c_float * 9),
                ("pixel_means_bgr", ctypes.c_float * 3), ("T_means", ctypes.c_float * 3), ("T_stds", ctypes.c_float * 3),
                ("rot_coord", ctypes.c_int), ("znear", ctypes.c_float), ("zfar", ctypes.c_float), ("tex_bilinear", ctypes.c_int),
                ("verts", ctypes.c_void_p), ("uvs", ctypes.c_void_p), ("faces", ctypes.c_void_p), ("mesh_table", ctypes.c_void_p),
                ("n_classes", ctypes.c_int), ("vmax", ctypes.c_int), ("fmax", ctypes.c_int), ("textures", ctypes.c_void_p),
                ("tex_table", ctypes.c_void_p)]


class CRefiner(object):
    """the refinement loop driven entirely by libdeepim_hip.so: same launches as deepim.core.tester.Refiner (FAST_TEST graph)"""

    def __init__(self, config, arg_params, render_machine, batch_size, device="cuda:0"):
        cfg, rm = config, render_machine
        self.B, self.T = int(batch_size), int(cfg.TEST.test_iter)
        self.device = torch.device(device)
        names = [n for n in arg_params if n.split("_weight")[0].split("_bias")[0] in
                 ("flow_conv1", "conv2", "conv3", "conv3_1", "conv4", "conv4_1", "conv5", "conv5_1", "conv6", "conv6_1", "fc6", "fc7", "rot", "trans")]
        self.params = {n: torch.as_tensor(np.ascontiguousarray(arg_params[n]), dtype=torch.float32).to(self.device) for n in names}
        d = RefinerDesc()
        d.B, d.H, d.W, d.test_iter = self.B, rm.height, rm.width, self.T
        d.K9 = (ctypes.c_float * 9)(*np.asarray(cfg.dataset.INTRINSIC_MATRIX, np.float32).reshape(9))
        d.pixel_means_bgr = (ctypes.c_float * 3)(*np.asarray(cfg.network.PIXEL_MEANS, np.float32).reshape(3))
        d.T_means = (ctypes.c_float * 3)(*np.asarray(cfg.dataset.trans_means, np.float32).reshape(3))
        d.T_stds = (ctypes.c_float * 3)(*np.asarray(cfg.dataset.trans_stds, np.float32).reshape(3))
        d.rot_coord, d.znear, d.zfar, d.tex_bilinear = capi.rot_coord_id(cfg.network.ROT_COORD), rm.zNear, rm.zFar, int(rm.tex_bilinear)
        d.verts, d.uvs, d.faces, d.mesh_table = rm.verts.data_ptr(), rm.uvs.data_ptr(), rm.faces.data_ptr(), rm.mesh_table.data_ptr()
        d.n_classes, d.vmax, d.fmax = int(rm.mesh_table.shape[0]), rm.vmax, rm.fmax
        d.textures, d.tex_table = rm.textures.data_ptr(), rm.tex_table.data_ptr()
        self._rm = rm   # keeps the mesh table alive
        cnames = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
        cptrs = (ctypes.c_void_p * len(names))(*[self.params[n].data_ptr() for n in names])
        self._h = ctypes.c_void_p()
        check(lib().dim_refiner_create(ctypes.byref(self._h), ctypes.byref(d), cnames, cptrs, len(names), current_stream()))
        torch.cuda.synchronize(self.device)
        self.poses_iter = torch.zeros((self.T, self.B, 3, 4), dtype=torch.float32, device=self.device)
        self.se3_iter = torch.zeros((self.T, self.B, 7), dtype=torch.float32, device=self.device)
        self.status_iter = torch.zeros((self.T, self.B), dtype=torch.int32, device=self.device)

    def refine(self, image_observed, image_rendered, mask_observed, mask_rendered, src_pose, class_index):
        f32 = torch.float32
        check(lib().dim_refiner_run(self._h, dptr(image_observed, f32), dptr(image_rendered, f32), dptr(mask_observed, f32),
                                    dptr(mask_rendered, f32), dptr(src_pose, f32), dptr(class_index, torch.int32), dptr(self.poses_iter, f32),
                                    dptr(self.se3_iter, f32), dptr(self.status_iter, torch.int32), current_stream()))
        return self.poses_iter

    def close(self):
        if self._h:
            check(lib().dim_refiner_destroy(self._h))
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
