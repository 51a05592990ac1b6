"""HIP rasteriser behind the reference's renderer object API.

Drop-in for /root/reference/lib/render_glumpy/render_py_multi.py:
    Render_Py(model_dir, classes, K, width=640, height=480, zNear=0.25, zFar=6.0)
    .render(cls_idx, r, t, r_type="quat"|"mat", K=None) -> (bgr float HxWx3 0..255, depth HxW metres)
but the meshes/textures live in HBM and rendering is `dim_raster_render` (csrc/raster.hip); no GL
context, no glReadPixels.  `render_batch` is the device-resident entry the refinement loop uses
(writes the next iteration's image_rendered / mask_rendered blobs directly).
"""
import os

import numpy as np
import torch

from lib.hip import capi
from lib.hip.capi import check, current_stream, dptr, host_f32, lib


def quat2mat(q):
    """RT_transform.quat2mat (reference lib/pair_matching/RT_transform.py:393-443), host-side helper."""
    w, x, y, z = [float(v) for v in q]
    Nq = w * w + x * x + y * y + z * z
    if Nq < np.finfo(np.float64).eps:
        return np.eye(3)
    s = 2.0 / Nq
    X, Y, Z = x * s, y * s, z * s
    return np.array(
        [
            [1.0 - (y * Y + z * Z), x * Y - w * Z, x * Z + w * Y],
            [x * Y + w * Z, 1.0 - (x * X + z * Z), y * Z - w * X],
            [x * Z - w * Y, y * Z + w * X, 1.0 - (x * X + y * Y)],
        ]
    )


def load_obj(path):
    """Minimal Wavefront OBJ reader (v / vt / f with v/vt[/vn] corners) -> verts (V,3), uvs (V,2), faces (F,3).
    Vertices are split per (v, vt) pair like glumpy.data.objload does (render_py_multi.py:69-71, rescale=False)."""
    vs, vts, corners, faces = [], [], {}, []
    out_v, out_t = [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                vs.append([float(x) for x in p[1:4]])
            elif p[0] == "vt":
                vts.append([float(x) for x in p[1:3]])
            elif p[0] == "f":
                idx = []
                for c in p[1:]:
                    parts = c.split("/")
                    vi = int(parts[0]) - 1
                    ti = int(parts[1]) - 1 if len(parts) > 1 and parts[1] else -1
                    key = (vi, ti)
                    if key not in corners:
                        corners[key] = len(out_v)
                        out_v.append(vs[vi])
                        out_t.append(vts[ti] if ti >= 0 else [0.0, 0.0])
                    idx.append(corners[key])
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(out_v, np.float32), np.asarray(out_t, np.float32), np.asarray(faces, np.int32)


class Render_Py(object):
    def __init__(self, model_dir, classes, K, width=640, height=480, zNear=0.25, zFar=6.0, device="cuda:0", meshes=None,
                 tex_bilinear=False):
        """meshes: optional list of (verts, uvs, faces, texture_uint8_HxWx3) replacing the textured.obj /
        texture_map.png files under model_dir/<class>/ (render_py_multi.py:66-78)."""
        self.width, self.height, self.zNear, self.zFar = width, height, zNear, zFar
        self.K = np.asarray(K, dtype=np.float32).reshape(3, 3)
        self.model_dir = model_dir
        self.classes = list(classes)
        self.device = torch.device(device)
        self.tex_bilinear = bool(tex_bilinear)
        if meshes is None:
            from PIL import Image

            meshes = []
            for cur_class in self.classes:
                folder = os.path.join(model_dir, cur_class)
                v, t, f = load_obj("{}/textured.obj".format(folder))
                tex = np.asarray(Image.open("{}/texture_map.png".format(folder)).convert("RGB"), dtype=np.uint8)
                meshes.append((v, t, f, tex))
        assert len(meshes) == len(self.classes)
        self._upload(meshes)
        self._ws = None
        self._ws_B = 0

    def _upload(self, meshes):
        vo = fo = to = 0
        table, ttable, V, T, F, X = [], [], [], [], [], []
        seen = {}  # a texture object shared by several meshes (ModelNet's gray_texture.png) is uploaded once
        for v, t, f, tex in meshes:
            v = np.ascontiguousarray(v, np.float32)
            t = np.ascontiguousarray(t, np.float32)
            f = np.ascontiguousarray(f, np.int32)
            tex = np.ascontiguousarray(tex, np.uint8)
            assert tex.ndim == 3 and tex.shape[2] == 3
            assert f.min() >= 0 and f.max() < v.shape[0], "face index out of range"
            table.append([vo, v.shape[0], fo, f.shape[0]])
            if id(tex) in seen:
                ttable.append([seen[id(tex)], tex.shape[0], tex.shape[1]])
            else:
                seen[id(tex)] = to
                ttable.append([to, tex.shape[0], tex.shape[1]])
                X.append(tex.reshape(-1))
                to += tex.size
            V.append(v); T.append(t); F.append(f)
            vo += v.shape[0]; fo += f.shape[0]
        d = self.device
        self.verts = torch.from_numpy(np.concatenate(V)).to(d)
        self.uvs = torch.from_numpy(np.concatenate(T)).to(d)
        self.faces = torch.from_numpy(np.concatenate(F)).to(d)
        self.textures = torch.from_numpy(np.concatenate(X)).to(d)
        self.mesh_table = torch.tensor(table, dtype=torch.int32, device=d)
        self.tex_table = torch.tensor(ttable, dtype=torch.int32, device=d)
        self.vmax = max(r[1] for r in table)
        self.fmax = max(r[3] for r in table)
        self.mesh_bytes = int(self.verts.numel() * 4 + self.uvs.numel() * 4 + self.faces.numel() * 4)

    def _workspace(self, B):
        """one workspace per batch size, zero-filled when allocated: its first 256 bytes are the rasteriser's header ("the z-buffer
        behind me is clear"), which must not hold what a previous owner of the memory left there (include/deepim_hip.h); the layout
        behind the header depends on B, so two batch sizes never share one"""
        if not isinstance(self._ws, dict):
            self._ws = {}
        if B not in self._ws:
            n = lib().dim_raster_workspace_bytes(B, self.vmax, self.height, self.width)
            self._ws[B] = torch.zeros((n + 7) // 8, dtype=torch.int64, device=self.device)
        return self._ws[B]

    def reserve(self, B):
        """pre-allocate the z-buffer workspace (call before hipGraph capture)."""
        self._workspace(B)

    def render_batch(self, class_index, poses, K=None, image=None, depth=None, mask=None, bgr=None, bbox=None,
                     plane_means=None, mask_thr=0.2, status=None, clean_bbox=None):
        """class_index (B,) int32 cuda, poses (B,3,4) f32 cuda.  Any of the output tensors may be None.
        status: optional (B,) int32 cuda; DIM_STATUS_BAD_CLASS (4) / DIM_STATUS_BAD_FACE (8) are OR-ed in.
        clean_bbox: optional (B,4) int32 cuda, the bbox a PREVIOUS render_batch into the same output tensors returned (another tensor
        than `bbox`): the planes hold background outside it, and pixels out there that this render does not cover are not rewritten."""
        B = poses.shape[0]
        keep, kp = host_f32(self.K if K is None else K, 9)
        pm = host_f32(plane_means, 3) if plane_means is not None else (None, None)
        ws = self._workspace(B)
        if clean_bbox is not None and not mask_thr < self.zNear:
            clean_bbox = None   # the mask's box is the box of everything drawn only if every fragment passes the mask threshold
        check(lib().dim_raster_render_dirty(
            dptr(self.verts), None, dptr(self.uvs), dptr(self.faces), dptr(self.mesh_table), int(self.mesh_table.shape[0]), self.vmax, self.fmax,
            dptr(self.textures), dptr(self.tex_table), dptr(class_index, torch.int32), dptr(poses, torch.float32), kp, B, self.height,
            self.width, float(self.zNear), float(self.zFar), int(self.tex_bilinear), None, None, 0.0, pm[1], float(mask_thr), ws.data_ptr(),
            dptr(image), dptr(depth), dptr(mask), dptr(bgr), dptr(bbox, torch.int32) if bbox is not None else None,
            dptr(status, torch.int32) if status is not None else None, dptr(clean_bbox, torch.int32) if clean_bbox is not None else None,
            current_stream()))

    def render(self, cls_idx, r, t, r_type="quat", K=None):
        """Reference signature (render_py_multi.py:112-147); returns host numpy like glReadPixels did."""
        if r_type == "quat":
            R = quat2mat(r)
        elif r_type == "mat":
            R = np.asarray(r)
        else:
            raise Exception("Unknown r_type: {}".format(r_type))
        pose = np.zeros((1, 3, 4), dtype=np.float32)
        pose[0, :, :3] = R
        pose[0, :, 3] = np.asarray(t, dtype=np.float32).squeeze()
        d = self.device
        bgr = torch.empty((1, self.height, self.width, 3), dtype=torch.float32, device=d)
        depth = torch.empty((1, 1, self.height, self.width), dtype=torch.float32, device=d)
        self.render_batch(torch.tensor([cls_idx], dtype=torch.int32, device=d), torch.from_numpy(pose).to(d), K=K, bgr=bgr,
                          depth=depth)
        return bgr[0].cpu().numpy(), depth[0, 0].cpu().numpy()
