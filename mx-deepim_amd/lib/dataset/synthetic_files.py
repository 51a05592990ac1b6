"""A synthetic LINEMOD-shaped dataset ON DISK: the file tree and pairdb records the reference's dataset classes produce
(lib/dataset/LM6D_REFINE.py:167-230: image_observed / image_rendered colour PNGs, depth_rendered / depth_gt_observed 16-bit PNGs in
mm (DEPTH_FACTOR 1000), mask_gt_observed label PNG with mask_idx, pose_observed / pose_rendered, gt_class, height, width,
img_flipped) plus `models/<class>/points.xyz` for the point-matching loss -- rendered by the HIP rasteriser from the seeded synthetic
pairs of lib/utils/synthetic.py.  No dataset exists offline; this is what lets deepim/core/loader.py's file path
(PIL decode -> pinned staging -> device blobs, pixel cache) run end to end in tests and in bench.py's `train_fresh_batch` object."""
from __future__ import print_function, division

import os

import numpy as np
import torch
from PIL import Image

from lib.utils import synthetic as syn


def write_synthetic_dataset(root, render_machine, models, class_names, n_pairs, seed=2333, chunk=16, pixel_means=syn.PIXEL_MEANS,
                            depth_factor=1000.0, compress_level=1):
    """-> pairdb (list of dicts).  Files: <root>/pairs/<i>-{color,color_r,depth,depth_r,label}.png, <root>/models/<cls>/points.xyz"""
    from deepim.core.loader import raw_from_device_batch

    dev = render_machine.device
    H, W = render_machine.height, render_machine.width
    os.makedirs(os.path.join(root, "pairs"), exist_ok=True)
    for cls, m in zip(class_names, models):
        os.makedirs(os.path.join(root, "models", cls), exist_ok=True)
        np.savetxt(os.path.join(root, "models", cls, "points.xyz"), m[0].astype(np.float64))
    pairdb = []
    for c0 in range(0, n_pairs, chunk):
        B = min(chunk, n_pairs - c0)
        b = syn.build_device_batch(render_machine, B, seed=seed + 31 * (c0 // chunk), n_classes=len(class_names), pixel_means=pixel_means,
                                   device=dev)
        depth_r = torch.empty((B, 1, H, W), device=dev)
        depth_gt = torch.empty((B, 1, H, W), device=dev)
        render_machine.render_batch(b["class_index"], b["src_pose"], depth=depth_r)
        render_machine.render_batch(b["class_index"], b["pose_gt"], depth=depth_gt)
        obs, ren, d_r, pose_r, cls, pose_o = raw_from_device_batch(b, pixel_means, depth_r, depth_factor)
        d_o = np.clip(np.rint(depth_gt.cpu().numpy()[:, 0] * depth_factor), 0, 65535).astype(np.uint16)
        for j in range(B):
            i = c0 + j
            p = {k: os.path.join(root, "pairs", "{:06d}-{}.png".format(i, k)) for k in ("color", "color_r", "depth", "depth_r", "label")}
            Image.fromarray(np.ascontiguousarray(obs[j][:, :, ::-1])).save(p["color"], compress_level=compress_level)    # files hold RGB
            Image.fromarray(np.ascontiguousarray(ren[j][:, :, ::-1])).save(p["color_r"], compress_level=compress_level)
            Image.fromarray(d_o[j]).save(p["depth"], compress_level=compress_level)
            Image.fromarray(d_r[j]).save(p["depth_r"], compress_level=compress_level)
            Image.fromarray((d_o[j] > 0).astype(np.uint8)).save(p["label"], compress_level=compress_level)
            pairdb.append({"image_observed": p["color"], "image_rendered": p["color_r"], "depth_gt_observed": p["depth"],
                           "depth_observed": p["depth"], "depth_rendered": p["depth_r"], "mask_gt_observed": p["label"], "mask_idx": 1,
                           "pose_observed": pose_o[j].astype(np.float32), "pose_rendered": pose_r[j].astype(np.float32),
                           "gt_class": class_names[int(cls[j])], "height": H, "width": W, "img_flipped": False})
    return pairdb
