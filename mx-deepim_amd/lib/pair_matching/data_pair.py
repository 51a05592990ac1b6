"""Batch assembly from pairdb records with the reference's function names, blob names and shapes
(lib/pair_matching/data_pair.py:22-72 test, :144-265 train; consumed by deepim/core/loader.py:35-41, :164-193).

    get_data_pair_test_batch(pairdb, config)  -> data: list of per-pair dicts, label: {}, im_info
    get_data_pair_train_batch(pairdb, config) -> {"data": {...}, "label": {...}} with every array stacked over the pairs

These are the HOST forms (float arrays in the reference's layouts): what a maintainer diffs against the reference and what the
tests use as the checker of the device path.  The product's loader (deepim/core/loader.py here) ships the compact raw pixels
(uint8 colour, uint16 depth: 2.1 MB per pair instead of 9.8 MB of float blobs) and builds the same blobs on the GPU
(dim_test_blobs_from_raw); `update_data_batch` of the reference (:75-138) is the device loop of deepim/core/tester.py.
"""
from __future__ import print_function, division

import numpy as np

from lib.pair_matching.RT_transform import calc_RT_delta
from lib.utils.image import (get_gt_observed_depth, get_pair_depth, get_pair_flow, get_pair_image, get_pair_mask, get_point_cloud_model,
                             get_point_cloud_observed, my_tensor_vstack)


def _class_index(config, rec):
    return np.array(config.dataset.class_name.index(rec["gt_class"])).reshape(1)


def get_data_pair_test_batch(pairdb, config):
    im_observed, im_rendered, scale_ind_list = get_pair_image(pairdb, config, "test")
    if config.network.INPUT_DEPTH:
        depth_observed, depth_rendered = get_pair_depth(pairdb, config, scale_ind_list, "test")
    if config.network.INPUT_MASK:
        mask_observed, _, mask_rendered = get_pair_mask(pairdb, config, scale_ind_list, "test")
    im_info = [np.array([rec["height"], rec["width"]], dtype=np.float32) for rec in pairdb]
    # (the reference rebuilds the class-index array inside a loop and ends with the LAST pair's index alone, shared by every
    # per-pair dict -- harmless there because the test loader feeds one pair per GPU; each pair carries its own index here)
    data = []
    for i, rec in enumerate(pairdb):
        cur = {"image_observed": im_observed[i], "image_rendered": im_rendered[i],
               "src_pose": np.array(rec["pose_rendered"]).reshape((1, 3, 4)), "class_index": _class_index(config, rec)}
        if config.network.INPUT_DEPTH:
            cur["depth_observed"], cur["depth_rendered"] = depth_observed[i], depth_rendered[i]
        if config.network.INPUT_MASK:
            cur["mask_observed"], cur["mask_rendered"] = mask_observed[i], mask_rendered[i]
        data.append(cur)
    return data, {}, im_info


def get_data_pair_train_batch(pairdb, config):
    n = len(pairdb)
    random_k = np.random.randint(18)
    im_observed, im_rendered, scale_ind_list = get_pair_image(pairdb, config, phase="train", random_k=random_k)
    depth_gt_observed = get_gt_observed_depth(pairdb, config, scale_ind_list, random_k=random_k)
    data = {"image_observed": my_tensor_vstack(im_observed), "image_rendered": my_tensor_vstack(im_rendered),
            "depth_gt_observed": my_tensor_vstack(depth_gt_observed)}
    label = {}
    if config.network.INPUT_DEPTH:
        d_obs, d_ren = get_pair_depth(pairdb, config, scale_ind_list, phase="train", random_k=random_k)
        data["depth_observed"], data["depth_rendered"] = my_tensor_vstack(d_obs), my_tensor_vstack(d_ren)
    if config.network.PRED_MASK or config.network.INPUT_MASK:
        m_obs, m_gt, m_ren = get_pair_mask(pairdb, config, scale_ind_list, phase="train", random_k=random_k)
        if config.network.INPUT_MASK:
            data["mask_observed"], data["mask_rendered"] = my_tensor_vstack(m_obs), my_tensor_vstack(m_ren)
        if config.network.PRED_MASK:
            label["mask_gt_observed"] = my_tensor_vstack(m_gt)
    if config.network.PRED_FLOW:
        flow, flow_w, _, _ = get_pair_flow(pairdb, config, scale_ind_list, phase="train", random_k=random_k)
        label["flow"], label["flow_weights"] = my_tensor_vstack(flow), my_tensor_vstack(flow_w)
    if config.train_iter.SE3_PM_LOSS:
        X_obj, X_w = get_point_cloud_model(config, pairdb)   # ONE sample of the first pair's class for the whole batch (:559-590)
        X_obj, X_w = my_tensor_vstack(X_obj), my_tensor_vstack(X_w)
    rot, trans, src, tgt, tgt_pts = [], [], [], [], []
    for i, rec in enumerate(pairdb):
        r, t = calc_RT_delta(rec["pose_rendered"], rec["pose_observed"], config.dataset.trans_means, config.dataset.trans_stds,
                             config.network.ROT_COORD, config.network.ROT_TYPE)
        rot.append(np.array(r).reshape((1, -1)))
        trans.append(np.array(t).reshape((1, -1)))
        src.append(np.array(rec["pose_rendered"]).reshape((1, 3, 4)))
        tgt.append(np.array(rec["pose_observed"]).reshape((1, 3, 4)))
        if config.train_iter.SE3_PM_LOSS:
            # X_obj holds one entry; the reference indexes X_obj_array[i], i.e. it trains with BATCH_PAIRS == 1 per GPU -- the model
            # sample is shared by the batch here
            tgt_pts.append(get_point_cloud_observed(config, X_obj[min(i, X_obj.shape[0] - 1)], np.array(rec["pose_observed"]))[np.newaxis])
    data.update(class_index=my_tensor_vstack([_class_index(config, rec) for rec in pairdb]).reshape(n),
                src_pose=my_tensor_vstack(src), tgt_pose=my_tensor_vstack(tgt))
    label.update(rot=my_tensor_vstack(rot), trans=my_tensor_vstack(trans))
    if config.train_iter.SE3_PM_LOSS:
        label["point_cloud_model"] = np.repeat(X_obj[:1], n, axis=0) if X_obj.shape[0] != n else X_obj
        label["point_cloud_weights"] = np.repeat(X_w[:1], n, axis=0) if X_w.shape[0] != n else X_w
        label["point_cloud_observed"] = my_tensor_vstack(tgt_pts)
    return {"data": data, "label": label}
