"""Test-time refinement loop, CPU restatement (oracle; test-only).

Restates /root/reference/deepim/core/tester.py:476-598 (pred_eval inner loop) with
lib/pair_matching/data_pair.py:75-138 (update_data_batch) per pair, using the other oracle
pieces: zoom ops + FlowNet forward (oracle.flownet.forward_test), RT_transform (oracle.se3),
the software rasteriser (oracle.native.render) in place of Render_Py.
"""
import numpy as np

from . import flownet, native, se3 as ose3


def image_transform(im_bgr, pixel_means):
    """lib/utils/image.py:709-720 transform()."""
    out = np.zeros((1, 3, im_bgr.shape[0], im_bgr.shape[1]))
    for i in range(3):
        out[0, i] = im_bgr[:, :, 2 - i] - pixel_means[2 - i]
    return out


def update_mask_observed_box_rendered(mask_rendered):
    """data_pair.py:103-114 (UPDATE_MASK == 'box_rendered').  Reference raises on an empty mask (np.min of empty)."""
    m = np.zeros(mask_rendered.shape)
    nz_x = np.nonzero(np.max(mask_rendered, 0))[0]
    nz_y = np.nonzero(np.max(mask_rendered, 1))[0]
    if len(nz_x) == 0 or len(nz_y) == 0:
        raise ValueError("empty rendered mask (reference: np.min of an empty array)")
    m[np.min(nz_y):np.max(nz_y), np.min(nz_x):np.max(nz_x)] = 1.0
    return m


def refine_pair(params, mesh, blobs, K, pixel_means, T_means, T_stds, rot_coord="CAMERA", test_iter=4, znear=0.25, zfar=6.0,
                tex_bilinear=False, fast_test=True, return_outputs=False, lit=None, forced_poses=None, **graph):
    """One (observed, rendered) pair, batch 1 like the reference.
    forced_poses: None, or (test_iter, 3, 4): "teacher forcing" for loop-parity tests -- the pose this loop computes in iteration k
    is returned as usual, but the re-render, src_pose and the next compose continue from forced_poses[k] (the pose the loop under
    test produced), so every iteration is compared on identical inputs and a silhouette pixel that flips in iteration k cannot
    masquerade as (or hide) a feedback error in iteration k+1.
    lit: None, or dict(normals=(V,3), ratio=0.7) for the ModelNet branch of `render` (tester.py:204-243): light index 2,
    one np.random.uniform(0.9, 1.1, 3) intensity per re-render drawn from numpy's global RNG like the reference.
    graph: keyword arguments of flownet.forward_test selecting the graph variant (input_mask, pred_mask, input_depth).
    blobs: image_observed (1,3,H,W), image_rendered, mask_observed (1,1,H,W), mask_rendered, src_pose (1,3,4).
    mesh: (verts, uvs, faces, tex).  Returns list of poses (test_iter x (3,4) float64) and the per-iteration se3."""
    verts, uvs, faces, tex = mesh
    batch = {k: np.array(v, dtype=np.float32) for k, v in blobs.items()}
    pose_rendered = np.array(batch["src_pose"][0], dtype=np.float64)
    out = flownet.forward_test(params, batch, K, pixel_means, fast_test=fast_test, **graph)
    poses, se3s, outs = [], [], []
    for it in range(test_iter):
        se3 = np.squeeze(out["se3"]).astype("float32")
        se3s.append(se3)
        outs.append({k: out[k] for k in ("mask_observed_pred", "zoom_mask_prob", "flow_est_crop", "zoom_factor") if k in out})
        pose_new = ose3.RT_transform(pose_rendered, se3[:-3], se3[-3:], T_means, T_stds, rot_coord)
        poses.append(pose_new)
        if forced_poses is not None:
            pose_new = np.array(forced_poses[it], dtype=np.float64)
        if it < test_iter - 1:
            if lit is None:
                bgr, depth = native.render(verts, uvs, faces, tex, pose_new[:3, :3], pose_new[:, 3], K, znear=znear, zfar=zfar,
                                           tex_bilinear=tex_bilinear)
            else:
                light_position = native.modelnet_light_position(pose_new, idx=2)
                light_intensity = np.array([1, 1, 1])[0] * np.random.uniform(0.9, 1.1, size=(3,))
                bgr, depth = native.render_lit(verts, lit["normals"], uvs, faces, tex, pose_new[:3, :3], pose_new[:3, 3], K,
                                               light_position, light_intensity, lit.get("ratio", 0.7), znear=znear, zfar=zfar,
                                               tex_bilinear=tex_bilinear)
            image_refined = bgr.astype("uint8")  # tester.py:246
            mask_r = np.zeros(depth.shape)
            mask_r[depth > 0.2] = 1  # tester.py:575-577
            batch["image_rendered"] = image_transform(image_refined.astype(np.float64), pixel_means).astype(np.float32)
            batch["mask_rendered"] = mask_r[np.newaxis, np.newaxis].astype(np.float32)
            batch["mask_observed"] = update_mask_observed_box_rendered(mask_r)[np.newaxis, np.newaxis].astype(np.float32)
            if graph.get("input_depth"):  # tester.py:573-574 + update_data_batch (data_pair.py:96-101): the rendered depth as it is
                batch["depth_rendered"] = depth[np.newaxis, np.newaxis].astype(np.float32)
            batch["src_pose"] = pose_new[np.newaxis].astype(np.float32)  # nd.array -> float32
            pose_rendered = pose_new
            out = flownet.forward_test(params, batch, K, pixel_means, fast_test=fast_test, **graph)
    if return_outputs:  # tester.py:485-491 reads the mask / flow heads every iteration when not FAST_TEST
        return poses, se3s, outs
    return poses, se3s


def update_train_batch(blobs, preds, meshes, K, pixel_means, T_means, T_stds, rot_coord="CAMERA", znear=0.25, zfar=6.0, lit=None):
    """batchUpdaterPyMulti.forward restated (lib/pair_matching/batch_updater_py_multi.py:205-365) for one GPU's blobs (numpy).
    preds: rot_est (B,4) [= rot_est_norm], trans_est (B,3).  Returns the dict of updated blobs.
    lit: None, or dict(normals=[(V,3) per mesh], ratio=0.7) for the ModelNet branch (:232-270): light index 2 moved by the refined
    translation, one np.random.uniform(0.9, 1.1, 3) intensity per sample in batch order."""
    B = blobs["src_pose"].shape[0]
    H, W = blobs["image_rendered"].shape[2:]
    K = np.asarray(K, dtype=np.float32)
    Kinv = np.linalg.inv(np.matrix(K))
    pm = np.asarray(pixel_means, dtype=np.float32).reshape(3)[[2, 1, 0]].reshape(3, 1, 1)  # self.pixel_means (:24-25)
    img = np.zeros((B, 3, H, W))
    dep = np.zeros((B, 1, H, W))
    rot_res, trans_res, poses, KT = np.zeros((B, 4)), np.zeros((B, 3)), np.zeros((B, 3, 4)), np.zeros((B, 3, 4))
    for b in range(B):
        refined = ose3.RT_transform(np.squeeze(blobs["src_pose"][b]), np.squeeze(preds["rot_est"][b]), np.squeeze(preds["trans_est"][b]),
                                    T_means, T_stds, rot_coord)
        v, t, f, tex = meshes[int(blobs["class_index"][b])]
        if lit is None:
            bgr, depth = native.render(v, t, f, tex, refined[:3, :3], refined[:3, 3], K, znear=znear, zfar=zfar)
        else:
            light_position = native.modelnet_light_position(refined, idx=2)
            light_intensity = np.array([1, 1, 1])[0] * np.random.uniform(0.9, 1.1, size=(3,))
            bgr, depth = native.render_lit(v, lit["normals"][int(blobs["class_index"][b])], t, f, tex, refined[:3, :3], refined[:3, 3], K,
                                           light_position, light_intensity, lit.get("ratio", 0.7), znear=znear, zfar=zfar)
        im = bgr[:, :, [2, 1, 0]].transpose([2, 0, 1]).astype(np.float32)
        im -= pm
        r, tr = ose3.calc_RT_delta(refined, np.squeeze(blobs["tgt_pose"][b]), T_means, T_stds, rot_coord, "QUAT")
        poses[b], img[b], dep[b, 0], rot_res[b], trans_res[b] = refined, im, depth, r, tr
        se3_m = np.zeros([3, 4])
        se3_m[:, :3], se3_m[:, 3] = ose3.calc_se3(refined, np.squeeze(blobs["tgt_pose"][b]))
        KT[b] = np.dot(K, se3_m)
    out = dict(blobs)
    mask = np.zeros(dep.shape)
    mask[dep > 0.2] = 1
    flow, valid = native.gpu_flow(dep.astype(np.float32), blobs["depth_gt_observed"].astype(np.float32), KT.astype(np.float32),
                                  np.array(Kinv).astype(np.float32))
    out.update(image_rendered=img.astype(np.float32), src_pose=poses.astype(np.float32), rot=rot_res.astype(np.float32),
               trans=trans_res.astype(np.float32), flow=flow, flow_weights=np.tile(valid, [1, 2, 1, 1]), mask_rendered=mask.astype(np.float32))
    return out
