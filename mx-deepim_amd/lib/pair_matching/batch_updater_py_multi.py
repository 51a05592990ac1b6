"""Between-iteration batch update of the iterative training loop, device resident.

Mirror of /root/reference/lib/pair_matching/batch_updater_py_multi.py (`batchUpdaterPyMulti.forward` :100-387): the reference
pulls src/tgt poses, predictions and the GT depth to the host, then per sample composes the pose (RT_transform), renders with
OpenGL, mean-subtracts, regenerates the labels (calc_RT_delta), builds K*se3, runs the CUDA flow kernel through host copies and
re-uploads everything.  Here every step is one batched HIP call on tensors that never leave HBM:

    src_pose'   = dim_se3_compose(src_pose, [rot_est, trans_est])                       (:216-223)
    image_rendered', depth_rendered', mask_rendered' = dim_raster_render(src_pose')     (:225-284, :316-319)
    rot', trans' = dim_se3_delta(src_pose', tgt_pose)                                   (:288-295)
    KT          = dim_pose_to_KT(src_pose', tgt_pose)                                   (:306-312)
    flow', valid = dim_depth_to_flow(depth_rendered', depth_gt_observed, KT, Kinv)      (:331-345); flow_weights' = tile(valid, 2)
"""
from __future__ import print_function, division

import numpy as np
import torch

from lib.hip import ops


class batchUpdaterPyMulti(object):
    def __init__(self, big_cfg, height, width, render_machine=None):
        self.big_cfg = big_cfg
        self.height, self.width = height, width
        self.rot_coord = big_cfg.network.ROT_COORD
        self.T_means = np.asarray(big_cfg.dataset.trans_means, dtype=np.float32)
        self.T_stds = np.asarray(big_cfg.dataset.trans_stds, dtype=np.float32)
        self.K = np.asarray(big_cfg.dataset.INTRINSIC_MATRIX, dtype=np.float32).reshape(3, 3)
        self.Kinv = np.linalg.inv(np.matrix(self.K))  # same call as the reference (:41)
        self.plane_means = np.asarray(big_cfg.network.PIXEL_MEANS, dtype=np.float32).reshape(3)[::-1].copy()  # pixel_means[[2,1,0]] (:24-25)
        # ModelNet (:232-270): the render machine is the lit one (Render_Py_Light_ModelNet_Multi) and forward() draws its intensities
        if big_cfg.dataset.dataset.startswith("ModelNet") and render_machine is not None and not hasattr(render_machine, "normals"):
            raise Exception("ModelNet batches re-render with the lit renderer (Render_Py_Light_ModelNet_Multi); got {}".format(
                type(render_machine).__name__))
        self.render_machine = render_machine
        self._bufs = None

    def forward(self, data_batch, preds, big_cfg=None):
        """data_batch: dict of CUDA blobs (data + labels, reference names); preds: dict with rot_est_norm (B,4), trans_est (B,3).
        Updates data_batch IN PLACE (image_rendered, depth_rendered if present, mask_rendered, src_pose, rot, trans, flow,
        flow_weights) and returns it."""
        cfg = self.big_cfg
        B = data_batch["src_pose"].shape[0]
        d = data_batch["src_pose"].device
        if self._bufs is None or self._bufs["pose"].shape[0] != B:
            self._bufs = {"se3": torch.empty((B, 7), device=d), "pose": torch.empty((B, 3, 4), device=d),
                          "depth": torch.empty((B, 1, self.height, self.width), device=d), "KT": torch.empty((B, 3, 4), device=d),
                          "valid": torch.empty((B, 1, self.height, self.width), device=d)}
            self.render_machine.reserve(B)
        b = self._bufs
        ops.copy_nhwc_channels(b["se3"], 0, preds["rot_est_norm"].contiguous(), 0, 4)   # strided copies as kernels of this library
        ops.copy_nhwc_channels(b["se3"], 4, preds["trans_est"].contiguous(), 0, 3)
        ops.se3_compose(data_batch["src_pose"], b["se3"], self.rot_coord, self.T_means, self.T_stds, out=b["pose"])
        depth = data_batch.get("depth_rendered", b["depth"])
        extra = {}
        if hasattr(self.render_machine, "normals"):  # ModelNet lit renderer (:232-270): light idx 2, host-drawn intensity per sample
            B = b["pose"].shape[0]
            li = np.stack([np.random.uniform(0.9, 1.1, size=(3,)) for _ in range(B)]).astype(np.float32)
            extra["light_intensity"] = torch.from_numpy(li).to(b["pose"].device)
        self.render_machine.render_batch(data_batch["class_index"], b["pose"], image=data_batch["image_rendered"], depth=depth,
                                         mask=data_batch["mask_rendered"] if cfg.network.INPUT_MASK else None, plane_means=self.plane_means,
                                         mask_thr=0.2, **extra)
        rot, trans = ops.se3_delta(b["pose"], data_batch["tgt_pose"], self.rot_coord, self.T_means, self.T_stds)
        ops.copy(data_batch["rot"], rot)
        ops.copy(data_batch["trans"], trans)
        if cfg.network.PRED_FLOW:
            ops.pose_to_KT(b["pose"], data_batch["tgt_pose"], self.K, out=b["KT"])
            ops.depth_to_flow(depth, data_batch["depth_gt_observed"], b["KT"], np.asarray(self.Kinv, dtype=np.float32), flow=data_batch["flow"],
                              valid=b["valid"])
            fw, va, hw = data_batch["flow_weights"].view(B, -1), b["valid"].view(B, -1), self.height * self.width
            ops.copy_nhwc_channels(fw, 0, va, 0, hw)    # np.tile(valid, [1, 2, 1, 1]) (:352)
            ops.copy_nhwc_channels(fw, hw, va, 0, hw)
        ops.copy(data_batch["src_pose"], b["pose"])
        return data_batch
