"""Pair loading with the reference's function names and blob conventions (lib/utils/image.py:65-553, :709-803), on PIL + numpy
(the reference decodes with cv2, which is not a dependency here).

A `pairdb` record is a dict (lib/dataset/LM6D_REFINE.py:167-230):
    image_observed, image_rendered          colour image files (read as BGR uint8, like cv2.IMREAD_COLOR)
    depth_rendered, depth_gt_observed [, depth_observed]   16-bit PNG, metres * config.dataset.DEPTH_FACTOR
    mask_gt_observed [, mask_observed, mask_observed_est, mask_syn]   8-bit label images; the object is label `mask_idx`
    pose_observed, pose_rendered            3x4 [R|t]
    gt_class, height, width, img_flipped [, data_syn]
Every getter returns a list with one (1,C,H,W) array per pair, exactly as the reference does, so deepim/core/loader.py-style code and
lib/pair_matching/data_pair.py stack them unchanged.

Conventions restated: image tensors are RGB planes of the BGR image minus PIXEL_MEANS[2-c]; masks are float
{0,1}; `mask_rendered` is the rendered DEPTH with values above 0.2 m replaced by 1 (so it is 0 on the background: image.py:485-488);
rectangles from get_min_rect are end-exclusive; flow labels are "[h, w]" unless STANDARD_FLOW_REP.

Deviation, documented: `resize` is the identity for the shipped SCALES ([[480, 640]] on 480x640 images, scale 1.0); for other scales
PIL's bilinear filter stands in for cv2.INTER_LINEAR (not bit-identical when shrinking).  JPEG backgrounds decode through PIL's libjpeg.
"""
from __future__ import print_function, division

import os
import random

import numpy as np
from PIL import Image

from lib.utils.get_min_rect import get_min_rect
from lib.utils.mask_dilate import mask_dilate

INTER_LINEAR, INTER_NEAREST = "linear", "nearest"   # stand-ins for the cv2 constants callers pass


# ---------------------------------------------------------------------------------------------------------------- file decoding
def _need(path):
    assert os.path.exists(path), "{} does not exist".format(path)
    return path


def imread_color(path):
    """cv2.imread(path, cv2.IMREAD_COLOR): (H,W,3) uint8, channels in B,G,R order"""
    with Image.open(_need(path)) as im:
        rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(rgb[:, :, ::-1])


def imread_unchanged(path):
    """cv2.imread(path, cv2.IMREAD_UNCHANGED) for what the pairdb holds: 16-bit depth -> uint16, 8-bit labels -> uint8,
    colour -> BGR uint8"""
    with Image.open(_need(path)) as im:
        if im.mode in ("I;16", "I;16B", "I;16L", "I"):
            return np.asarray(im).astype(np.uint16)
        if im.mode in ("L", "P", "1"):
            return np.asarray(im.convert("L"), dtype=np.uint8)
        return np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8)[:, :, ::-1])


def _depth_metres(path, config):
    return imread_unchanged(path).astype(np.float32) / config.dataset.DEPTH_FACTOR


# ---------------------------------------------------------------------------------------------------------------- tensors
def resize(im, target_size, max_size, stride=0, interpolation=INTER_LINEAR):
    """scale so that the short side becomes target_size unless the long side would exceed max_size (reference :680-706).
    -> (image, scale); stride > 0 zero-pads height / width up to a multiple of it."""
    h, w = im.shape[:2]
    im_scale = float(target_size) / float(min(h, w))
    if np.round(im_scale * max(h, w)) > max_size:
        im_scale = float(max_size) / float(max(h, w))
    if im_scale != 1.0:
        new_w, new_h = int(round(w * im_scale)), int(round(h * im_scale))
        flt = Image.NEAREST if interpolation == INTER_NEAREST else Image.BILINEAR
        planes = im.reshape(h, w, -1)
        out = np.stack([np.asarray(Image.fromarray(planes[:, :, c].astype(np.float32), mode="F").resize((new_w, new_h), flt))
                        for c in range(planes.shape[2])], axis=2)
        im = out.reshape((new_h, new_w) + im.shape[2:]).astype(im.dtype if im.dtype.kind == "f" else np.float32)
    if stride == 0:
        return im, im_scale
    H = int(np.ceil(im.shape[0] / float(stride)) * stride)
    W = int(np.ceil(im.shape[1] / float(stride)) * stride)
    padded = np.zeros((H, W, im.shape[2]))
    padded[:im.shape[0], :im.shape[1], :] = im
    return padded, im_scale


def transform(im, pixel_means):
    """(H,W,3) BGR -> (1,3,H,W): plane c = im[:, :, 2-c] - pixel_means[2-c]   (reference :709-720)"""
    im = np.asarray(im)
    pm = np.asarray(pixel_means, dtype=np.float64).reshape(3)
    out = np.empty((1, 3, im.shape[0], im.shape[1]))
    out[0] = im[:, :, ::-1].transpose(2, 0, 1) - pm[::-1].reshape(3, 1, 1)
    return out


def transform_inverse(im_tensor, pixel_means):
    """(1,3,H,W) -> (H,W,3) uint8 RGB (reference :736-753)"""
    assert im_tensor.shape[0] == 1 and im_tensor.shape[1] == 3
    pm = np.asarray(pixel_means, dtype=np.float64).reshape(3)
    return (im_tensor[0].transpose(1, 2, 0) + pm[[2, 1, 0]]).astype(np.uint8)


def transform_seg_gt(gt):
    return np.asarray(gt, dtype=np.float64)[np.newaxis, np.newaxis, :, :].copy()


def my_tensor_vstack(tensor_list):
    return np.concatenate(tensor_list, axis=0)


def _scale_of(config, scale_ind):
    return config.SCALES[scale_ind][0], config.SCALES[scale_ind][1]


def _plane(a):
    return a[np.newaxis, np.newaxis, :, :]


# ---------------------------------------------------------------------------------------------------------------- images
_voc_lists = {}


def _voc_backgrounds(config):
    """VOC2012 images whose 'diningtable' flag is 1 (reference :107-124); the list file is read once"""
    root = os.path.join(config.dataset.root_path, "VOCdevkit/VOC2012")
    if root not in _voc_lists:
        with open(os.path.join(root, "ImageSets/Main", "diningtable_trainval.txt")) as f:
            rows = [line.split() for line in f if line.strip()]
        _voc_lists[root] = [r[0] for r in rows if r[1] == "1"]
    return root, _voc_lists[root]


def fit_background(bg_image, height, width):
    """crop the background from its top-left corner to the observed image's aspect ratio, scale it to (height, width) and paste it at
    the top-left of a black canvas (reference :125-165; the crop may be a no-op when the background is already narrower / shorter)"""
    bg_h, bg_w = bg_image.shape[:2]
    hw_ratio = float(height) / float(width)
    same_orientation = (hw_ratio < 1) == (float(bg_h) / float(bg_w) < 1)
    if bg_h >= bg_w:
        new_h = int(np.ceil(bg_w * hw_ratio))
        crop = bg_image[:new_h, :bg_w] if (new_h < bg_h or not same_orientation) else bg_image
    else:
        new_w = int(np.ceil(bg_h / hw_ratio))
        crop = bg_image[:bg_h, :new_w] if (new_w < bg_w or not same_orientation) else bg_image
    scaled, _ = resize(crop, min(height, width), max(height, width))
    canvas = np.zeros((height, width, bg_image.shape[2]), dtype="uint8")
    h, w = min(scaled.shape[0], height), min(scaled.shape[1], width)
    canvas[:h, :w] = scaled[:h, :w]
    return canvas


def get_pair_image(pairdb, config, phase="train", random_k=18):
    """-> observed tensors, rendered tensors, the scale index drawn for every pair"""
    observed, rendered, scale_ind_list = [], [], []
    for rec in pairdb:
        scale_ind = random.randrange(len(config.SCALES))
        scale_ind_list.append(scale_ind)
        target, longest = _scale_of(config, scale_ind)
        if rec["img_flipped"]:
            raise Exception("NOT_IMPLEMENTED")
        im_obs, s_obs = resize(imread_color(rec["image_observed"]), target, longest)
        im_ren, s_ren = resize(imread_color(rec["image_rendered"]), target, longest)
        assert s_obs == s_ren, "scale mismatch"
        # synthetic observed images (and a share of the real ones) get a random VOC background behind the object
        if "data_syn" in rec and phase == "train":
            if rec["data_syn"] is True or (rec["data_syn"] is False and np.random.rand() < config.TRAIN.REPLACE_OBSERVED_BG_RATIO):
                voc_root, names = _voc_backgrounds(config)
                pick = names[random.randint(0, len(names) - 1)]
                bg = fit_background(imread_color(os.path.join(voc_root, "JPEGImages/{}.jpg".format(pick))), *im_obs.shape[:2])
                fg = imread_unchanged(rec["mask_gt_observed"]) != 0
                bg[fg] = im_obs[fg]
                im_obs = bg
        observed.append(transform(im_obs, config.network.PIXEL_MEANS))
        rendered.append(transform(im_ren, config.network.PIXEL_MEANS))
    return observed, rendered, scale_ind_list


# ---------------------------------------------------------------------------------------------------------------- depth
def get_gt_observed_depth(pairdb, config, scale_ind_list, phase="train", random_k=18):
    out = []
    for rec, scale_ind in zip(pairdb, scale_ind_list):
        d, _ = resize(imread_unchanged(rec["depth_gt_observed"]).astype(np.float32), *_scale_of(config, scale_ind))
        out.append(_plane(d / config.dataset.DEPTH_FACTOR))
    return out


def _observed_label_path(rec, config, phase, k):
    """which label image masks the observed depth (reference :243-262)"""
    if config.TRAIN.get("MASK_SYN", False) and phase == "train" and k < config.TRAIN.MASK_SYN_RATIO:
        return rec["mask_syn"]
    if config.dataset.get("MASK_GT", False) or phase == "train":
        return rec["mask_gt_observed"]
    return rec["mask_observed_est"]


def get_pair_depth(pairdb, config, scale_ind_list, phase="train", random_k=[]):
    observed, rendered = [], []
    for i, (rec, scale_ind) in enumerate(zip(pairdb, scale_ind_list)):
        size = _scale_of(config, scale_ind)
        d_obs = imread_unchanged(rec["depth_observed"]).astype(np.float32)
        if config.network.get("MASK_INPUTS", False):   # no default in config.py (the reference raises here without the key)
            k = random_k[i] if np.ndim(random_k) else random_k
            d_obs = d_obs * (imread_unchanged(_observed_label_path(rec, config, phase, k)) == rec["mask_idx"])
        d_ren = imread_unchanged(rec["depth_rendered"]).astype(np.float32)
        d_obs, _ = resize(d_obs, *size)
        d_ren, _ = resize(d_ren, *size)
        observed.append(_plane(d_obs / config.dataset.DEPTH_FACTOR))
        rendered.append(_plane(d_ren / config.dataset.DEPTH_FACTOR))
    return observed, rendered


# ---------------------------------------------------------------------------------------------------------------- masks
def _label_mask(path, mask_idx, size):
    """float {0,1} mask of label `mask_idx`, resized and re-binarised at 0.5 -> (mask, any pixel set before resizing)"""
    fg = imread_unchanged(path).astype(np.float32) == mask_idx
    m, _ = resize(fg.astype(np.float64), *size)
    m[m < 0.5] = 0.0
    return m, bool(fg.any())


def _rect_of(mask, strict):
    """filled end-exclusive bounding rectangle of `mask`; an empty mask raises (strict, training) or gives zeros (test)"""
    out = np.zeros(mask.shape)
    if np.count_nonzero(mask) == 0:
        assert not strict, "NO POINT VALID IN INIT MASK"
        print("NO POINT VALID IN INIT MASK")
        return out
    x0, y0, x1, y1 = get_min_rect(mask)
    out[y0:y1, x0:x1] = 1.0
    return out


def _rendered_fg(rec, config, size):
    d, _ = resize(imread_unchanged(rec["depth_rendered"]).astype(np.float32), *size)
    return (d / config.dataset.DEPTH_FACTOR > 0.2).astype(np.float64)


def get_pair_mask(pairdb, config, scale_ind_list, phase="train", random_k=[]):
    """-> mask_observed, mask_gt_observed, mask_rendered lists (reference :272-491).
    Train: mask_gt_observed from the label image; mask_observed by TRAIN.INIT_MASK (mask_gt | box_gt | box_rendered), optionally
    dilated.  Test: there is no ground truth, mask_gt_observed IS mask_observed; TEST.INIT_MASK in (mask_gt_observed | mask_observed |
    box_gt_observed | box_ | box_rendered); an all-zero rendered depth (object not detected) gives an empty mask."""
    observed, gt_observed, rendered = [], [], []
    for rec, scale_ind in zip(pairdb, scale_ind_list):
        size = _scale_of(config, scale_ind)
        if phase == "train":
            gt, any_fg = _label_mask(rec["mask_gt_observed"], rec["mask_idx"], size)
            assert any_fg, "NOT_VALID: {}".format(rec["mask_gt_observed"])
            kind = config.TRAIN.INIT_MASK
            if kind == "mask_gt":
                # (the reference copies the RAW label image here -- labels, not {0,1}, and un-resized: image.py:315-316)
                m = imread_unchanged(rec["mask_gt_observed"]).astype(np.float32).copy()
            elif kind == "box_gt":
                m = _rect_of(gt, strict=True)
            elif kind == "box_rendered":
                m = _rect_of(_rendered_fg(rec, config, size), strict=True)
            else:
                raise Exception("Unknown mask type: {}".format(kind))
            if config.TRAIN.MASK_DILATE:
                m = mask_dilate(m)
            observed.append(_plane(m))
            gt_observed.append(_plane(gt))
        else:
            if np.sum(imread_unchanged(rec["depth_rendered"]).astype(np.float32)) == 0:
                m = np.zeros((rec["height"], rec["width"])) if "height" in rec else np.zeros(imread_unchanged(rec["depth_rendered"]).shape)
                print("NO POINT VALID IN INIT MASK")
            else:
                kind = config.TEST.INIT_MASK
                if kind in ("mask_gt_observed", "mask_observed"):
                    m, _ = _label_mask(rec[kind], rec["mask_idx"], size)
                elif kind == "box_gt_observed":
                    m = _rect_of((imread_unchanged(rec["mask_gt_observed"]).astype(np.float32) == rec["mask_idx"]).astype(np.float64), strict=True)
                elif kind == "box_":
                    m = _rect_of((imread_unchanged(rec["mask_observed"]).astype(np.float32) == rec["mask_idx"]).astype(np.float64), strict=False)
                elif kind == "box_rendered":
                    m = _rect_of(_rendered_fg(rec, config, size), strict=False)
                else:
                    raise Exception("Unknown init mask type: {}".format(kind))
            if config.TEST.get("MASK_DILATE", False):   # set by the YAMLs, no default in config.py
                m = mask_dilate(m, max_thickness=10)
            observed.append(_plane(m))
            gt_observed.append(_plane(m))
        # mask_rendered: the depth itself with everything above 0.2 m set to 1
        d, _ = resize(imread_unchanged(rec["depth_rendered"]).astype(np.float32), *size)
        d = d / config.dataset.DEPTH_FACTOR
        d[d > 0.2] = 1
        rendered.append(_plane(d))
    return observed, gt_observed, rendered


# ---------------------------------------------------------------------------------------------------------------- labels
def get_pair_flow(pairdb, config, scale_ind_list, phase="train", random_k=[]):
    """-> flow (1,2,H,W), flow_weights (1,2,H,W), X_rendered_valid per pair, [] (reference :494-553)"""
    from lib.pair_matching.flow import calc_flow

    flows, weights, X_valid_list = [], [], []
    for rec in pairdb:
        d_ren = _depth_metres(rec["depth_rendered"], config)
        d_obs = _depth_metres(rec["depth_gt_observed"] if "depth_gt_observed" in rec else rec["depth_observed"], config)
        if config.network.PRED_FLOW or config.train_iter.SE3_PM_LOSS:
            flow, visible, X_valid = calc_flow(d_ren, rec["pose_rendered"], rec["pose_observed"], config.dataset.INTRINSIC_MATRIX, d_obs,
                                               standard_rep=config.network.STANDARD_FLOW_REP)
            flows.append(flow.transpose((2, 0, 1))[np.newaxis])
            kind = config.TRAIN.FLOW_WEIGHT_TYPE
            if kind == "all":
                w = np.ones(visible.shape, dtype=np.float32)
            elif kind == "viz":
                w = visible
            elif kind == "valid":
                w = np.logical_or(np.squeeze(d_ren == 0), visible)
            weights.append(np.tile(_plane(w), (1, 2, 1, 1)))
            X_valid_list.append(X_valid)
    return flows, weights, X_valid_list, []


point_cloud_dict = {}


def load_object_points(point_path):
    """lib/pair_matching/load_object_points.py:11-14: whitespace-separated x y z rows (points.xyz)"""
    return np.loadtxt(_need(point_path))


def get_point_cloud_model(config, pairdb):
    """NUM_3D_SAMPLE model points of the FIRST pair's class, drawn without replacement by shuffling (np.random), zero-padded with
    zero weights when the model has fewer (reference :559-590) -> [(1,3,n)], [(1,3,n)]"""
    cls = pairdb[0]["gt_class"]
    if cls not in point_cloud_dict:
        if not config.dataset.dataset.startswith("ModelNet"):
            point_cloud_dict[cls] = load_object_points(os.path.join(config.dataset.model_dir, cls, "points.xyz"))
        else:
            from lib.render_hip.render_py_light_modelnet_multi import load_obj_with_normals

            point_cloud_dict[cls] = load_obj_with_normals(os.path.join(config.dataset.model_dir, cls + ".obj"))[0].astype(np.float64)
    pts = point_cloud_dict[cls]
    n_all, n = pts.shape[0], config.train_iter.NUM_3D_SAMPLE
    keep = np.arange(n_all)
    np.random.shuffle(keep)
    keep = keep[:min(n_all, n)]
    sample, w = np.zeros((n, 3)), np.zeros((n, 3))
    sample[:len(keep)] = pts[keep]
    w[:len(keep)] = 1
    return [sample.T[np.newaxis]], [w.T[np.newaxis]]


def get_point_cloud_observed(config, points_model, pose_observed):
    pose_observed = np.asarray(pose_observed)
    return np.dot(pose_observed[:, :3], points_model) + pose_observed[:, 3].reshape((3, 1))
