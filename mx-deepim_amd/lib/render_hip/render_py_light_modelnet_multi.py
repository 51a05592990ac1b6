"""HIP rasteriser behind the reference's ModelNet renderer API (lit, gray texture).

Drop-in for /root/reference/lib/render_glumpy/render_py_light_modelnet_multi.py:
    Render_Py_Light_ModelNet_Multi(model_path_list, texture_path, K, width, height, zNear, zFar, brightness_ratios=[0.7])
    .render(model_idx, r, t, light_position, light_intensity, brightness_k=0, r_type="quat") -> (bgr uint8 HxWx3, depth HxW)
All meshes (256 for the ModelNet-unseen configuration) are resident in one HBM table and selected per sample by
`class_index`; shading is the fragment shader of the reference (:36-77) evaluated in the resolve pass of
`dim_raster_render_lit` (csrc/raster.hip).
"""
import numpy as np
import torch

from lib.hip.capi import check, current_stream, dptr, host_f32, lib
from lib.render_hip.render_py_multi import Render_Py, quat2mat

LIGHT_DIRS = [[1, 0, 1], [1, 1, 1], [0, 1, 1], [-1, 1, 1], [-1, 0, 1], [0, 0, 1]]  # tester.py:206-218


def load_obj_with_normals(path, rescale=True, scale=0.1):
    """Wavefront OBJ -> verts (V,3), normals (V,3), uvs (V,2), faces (F,3); corners split per (v, vt, vn) triple.
    The reference loads with glumpy.data.objload(path, rescale=True) and then divides positions by 10 (:113-114).
    glumpy is not available here; `rescale` is restated as glumpy's documented behaviour (positions mapped so that
    the largest axis-aligned extent spans [-1, 1] around the bounding-box centre) -- loader parity is unpinned.
    Missing normals are computed as area-weighted vertex normals."""
    vs, vts, vns, corners, faces = [], [], [], {}, []
    out_v, out_t, out_n = [], [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                vs.append([float(x) for x in p[1:4]])
            elif p[0] == "vt":
                vts.append([float(x) for x in p[1:3]])
            elif p[0] == "vn":
                vns.append([float(x) for x in p[1:4]])
            elif p[0] == "f":
                idx = []
                for c in p[1:]:
                    parts = c.split("/")
                    vi = int(parts[0]) - 1
                    ti = int(parts[1]) - 1 if len(parts) > 1 and parts[1] else -1
                    ni = int(parts[2]) - 1 if len(parts) > 2 and parts[2] else -1
                    key = (vi, ti, ni)
                    if key not in corners:
                        corners[key] = len(out_v)
                        out_v.append(vs[vi])
                        out_t.append(vts[ti] if ti >= 0 else [0.0, 0.0])
                        out_n.append(vns[ni] if ni >= 0 else [0.0, 0.0, 0.0])
                    idx.append(corners[key])
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    v = np.asarray(out_v, np.float64)
    n = np.asarray(out_n, np.float64)
    fc = np.asarray(faces, np.int32)
    if not vns:
        n = vertex_normals(v, fc)
    if rescale:
        lo, hi = v.min(0), v.max(0)
        v = (v - (lo + hi) / 2.0) / ((hi - lo).max() / 2.0)
    v = v * scale
    return v.astype(np.float32), n.astype(np.float32), np.asarray(out_t, np.float32), fc


def vertex_normals(verts, faces):
    """area-weighted per-vertex normals (unit length)"""
    v = np.asarray(verts, np.float64)
    fn = np.cross(v[faces[:, 1]] - v[faces[:, 0]], v[faces[:, 2]] - v[faces[:, 0]])
    n = np.zeros_like(v)
    for k in range(3):
        np.add.at(n, faces[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    return n / np.where(ln > 0, ln, 1.0)


class Render_Py_Light_ModelNet_Multi(Render_Py):
    def __init__(self, model_path_list, texture_path, K, width=640, height=480, zNear=0.25, zFar=6.0, brightness_ratios=[0.7],
                 device="cuda:0", meshes=None, tex_bilinear=False):
        """meshes: optional list of (verts, normals, uvs, faces) replacing the .obj files; texture_path may be an
        (Ht,Wt,3) uint8 array instead of the path of gray_texture.png (tester.py:175-176)."""
        self.model_path_list = list(model_path_list) if model_path_list is not None else list(range(len(meshes)))
        self.brightness_ratios = list(brightness_ratios)
        if isinstance(texture_path, np.ndarray):
            tex = np.ascontiguousarray(texture_path, np.uint8)
        else:
            from PIL import Image

            tex = np.asarray(Image.open(texture_path).convert("RGB"), dtype=np.uint8)
        if meshes is None:
            meshes = [load_obj_with_normals(p) for p in self.model_path_list]
        self.width, self.height, self.zNear, self.zFar = width, height, zNear, zFar
        self.K = np.asarray(K, dtype=np.float32).reshape(3, 3)
        self.classes = self.model_path_list
        self.device = torch.device(device)
        self.tex_bilinear = bool(tex_bilinear)
        # one shared texture: every class points at the same bytes
        self._upload([(v, t, f, tex) for v, n, t, f in meshes])
        self.normals = torch.from_numpy(np.concatenate([np.ascontiguousarray(n, np.float32) for v, n, t, f in meshes])).to(self.device)
        assert self.normals.shape == self.verts.shape
        self._ws = None
        self._ws_B = 0

    def light_position(self, poses, idx=2, out=None):
        """tester.py:204-225 on the device: (B,3) light positions for poses (B,3,4)."""
        B = poses.shape[0]
        out = torch.empty((B, 3), dtype=torch.float32, device=self.device) if out is None else out
        d = LIGHT_DIRS[idx % 6]
        check(lib().dim_modelnet_light_position(dptr(poses, torch.float32), float(d[0]), float(d[1]), float(d[2]), dptr(out, torch.float32),
                                                B, current_stream()))
        return out

    def render_batch(self, class_index, poses, light_position=None, light_intensity=None, brightness_k=0, K=None, image=None,
                     depth=None, mask=None, bgr=None, bbox=None, plane_means=None, mask_thr=0.2, status=None, clean_bbox=None):
        """class_index (B,) int32, poses (B,3,4), light_position / light_intensity (B,3) f32, all cuda.
        light_position None = the loop's rule (idx 2); light_intensity None = white (1,1,1).  clean_bbox: as Render_Py.render_batch."""
        B = poses.shape[0]
        if light_position is None:
            light_position = self.light_position(poses)
        if light_intensity is None:
            light_intensity = torch.ones((B, 3), dtype=torch.float32, device=self.device)
        keep, kp = host_f32(self.K if K is None else K, 9)
        pm = host_f32(plane_means, 3) if plane_means is not None else (None, None)
        ws = self._workspace(B)
        if clean_bbox is not None and not mask_thr < self.zNear:
            clean_bbox = None
        check(lib().dim_raster_render_dirty(
            dptr(self.verts), dptr(self.normals), dptr(self.uvs), dptr(self.faces), dptr(self.mesh_table), int(self.mesh_table.shape[0]),
            self.vmax, self.fmax,
            dptr(self.textures), dptr(self.tex_table), dptr(class_index, torch.int32), dptr(poses, torch.float32), kp, B, self.height,
            self.width, float(self.zNear), float(self.zFar), int(self.tex_bilinear), dptr(light_position, torch.float32),
            dptr(light_intensity, torch.float32), float(self.brightness_ratios[brightness_k]), pm[1], float(mask_thr), ws.data_ptr(),
            dptr(image), dptr(depth), dptr(mask), dptr(bgr), dptr(bbox, torch.int32) if bbox is not None else None,
            dptr(status, torch.int32) if status is not None else None, dptr(clean_bbox, torch.int32) if clean_bbox is not None else None,
            current_stream()))

    def render(self, model_idx, r, t, light_position, light_intensity, brightness_k=0, r_type="quat"):
        """Reference signature (:153-235); returns host numpy (bgr uint8, depth float32) like the glReadPixels path."""
        if r_type == "quat":
            R = quat2mat(r)
        elif r_type == "mat":
            R = np.asarray(r)
        pose = np.zeros((1, 3, 4), dtype=np.float32)
        pose[0, :, :3] = R
        pose[0, :, 3] = np.asarray(t, dtype=np.float32).squeeze()
        d = self.device
        bgr = torch.empty((1, self.height, self.width, 3), dtype=torch.float32, device=d)
        depth = torch.empty((1, 1, self.height, self.width), dtype=torch.float32, device=d)
        lp = torch.tensor(np.asarray(light_position, dtype=np.float32).reshape(1, 3), device=d)
        li = torch.tensor(np.asarray(light_intensity, dtype=np.float32).reshape(1, 3), device=d)
        self.render_batch(torch.tensor([model_idx], dtype=torch.int32, device=d), torch.from_numpy(pose).to(d), lp, li,
                          brightness_k=brightness_k, bgr=bgr, depth=depth)
        return bgr[0].cpu().numpy().astype(np.uint8), depth[0, 0].cpu().numpy()
