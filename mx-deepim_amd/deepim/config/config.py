"""Config / flag system with the reference's schema (/root/reference/deepim/config/config.py:11-171).

`config` is a global attribute-dict with the same defaults; `update_config(yaml)` merges an
experiments/deepim/cfgs/*.yaml exactly like the reference: unknown TOP-LEVEL keys raise ValueError
(:170-171), nested keys are merged blindly (:163-164), PIXEL_MEANS / INTRINSIC_MATRIX / trans_means /
trans_stds become numpy arrays (:138-162), SCALES becomes a tuple (:166-167).
"""
from __future__ import print_function, division

import copy

import numpy as np
import yaml


class edict(dict):
    """attribute-access dict (stand-in for easydict.EasyDict, which is not installed here)."""

    def __init__(self, d=None, **kwargs):
        super(edict, self).__init__()
        d = dict(d or {}, **kwargs)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, edict):
            v = edict(v)
        super(edict, self).__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __deepcopy__(self, memo):
        return edict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _defaults():
    c = edict()
    c.ModelNet = False
    c.modelnet_root = "./data/ModelNet/"
    c.MXNET_VERSION = ""
    c.output_path = ""
    c.symbol = ""
    c.SCALES = [(480, 640)]
    c.default = edict(frequent=1000, kvstore="device")
    c.network = edict(
        FIXED_PARAMS=[], PIXEL_MEANS=np.array([0, 0, 0]), pretrained="../model/pretrained_model/flownet", pretrained_epoch=0,
        init_from_flownet=False, skip_initialize=False, INPUT_DEPTH=False, INPUT_MASK=False, PRED_MASK=False, PRED_FLOW=False,
        STANDARD_FLOW_REP=False, TRAIN_ITER=False, TRAIN_ITER_SIZE=1, REGRESSOR_NUM=1, ROT_TYPE="QUAT", ROT_COORD="CAMERA",
        TRANS_LOSS_TYPE="L2")
    c.dataset = edict(
        dataset="LINEMOD_REFINE", dataset_path="./data/LINEMOD_6D/LINEMOD_converted/LINEMOD_refine", image_set="train_ape",
        root_path="./data", test_image_set="val_ape", model_dir="", model_file="./data/ModelNet/render_v1/models.txt",
        pose_file="./data/ModelNet/render_v1/poses.txt", DEPTH_FACTOR=1000, NORMALIZE_FLOW=1.0, NORMALIZE_3D_POINT=0.1,
        INTRINSIC_MATRIX=np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]]), ZNEAR=0.25, ZFAR=6.0,
        class_name_file="", class_name=[], trans_means=np.array([0.0, 0.0, 0.0]), trans_stds=np.array([1.0, 1.0, 1.0]))
    c.TRAIN = edict(
        optimizer="sgd", warmup=False, warmup_lr=0, warmup_step=0, begin_epoch=0, end_epoch=0, lr=0.0001, lr_step="4, 6",
        momentum=0.975, wd=0.0005, model_prefix="deepim", RESUME=False, SHUFFLE=True, BATCH_PAIRS=1, FLOW_WEIGHT_TYPE="all",
        TENSORBOARD_LOG=False, INIT_MASK="box_gt", UPDATE_MASK="box_gt", MASK_DILATE=False, REPLACE_OBSERVED_BG_RATIO=0.0)
    c.TEST = edict(BATCH_PAIRS=1, test_epoch=0, VISUALIZE=False, test_iter=1, INIT_MASK="box_rendered", UPDATE_MASK="box_rendered",
                   FAST_TEST=False, PRECOMPUTED_ICP=False, BEFORE_ICP=False)
    c.train_iter = edict(SE3_DIST_LOSS=False, LW_ROT=0.0, LW_TRANS=0.0, TRANS_LOSS_TYPE="L2", TRANS_SMOOTH_L1_SCALAR=3.0,
                         SE3_PM_LOSS=False, LW_PM=0.0, SE3_PM_LOSS_TYPE="L1", SE3_PM_SL1_SCALAR=1.0, NUM_3D_SAMPLE=-1, LW_FLOW=0.0,
                         LW_MASK=0.0)
    return c


config = _defaults()


def reset_config():
    """restore defaults in place (the reference has no such call; tests need it because `config` is global)."""
    config.clear()
    for k, v in _defaults().items():
        config[k] = v
    return config


def update_config(config_file):
    with open(config_file) as f:
        # reference: yaml.load(f) without a Loader (config.py:131) -- fails on PyYAML >= 6, safe_load is equivalent here
        exp_config = edict(yaml.safe_load(f))
    for k, v in exp_config.items():
        if k not in config:
            raise ValueError("key: {} does not exist in config.py".format(k))
        if isinstance(v, dict):
            if k == "TRAIN":
                if "BBOX_WEIGHTS" in v:
                    v["BBOX_WEIGHTS"] = np.array(v["BBOX_WEIGHTS"])
            elif k == "network":
                if "PIXEL_MEANS" in v:
                    v["PIXEL_MEANS"] = np.array(v["PIXEL_MEANS"])
            elif k == "dataset":
                if "INTRINSIC_MATRIX" in v:
                    v["INTRINSIC_MATRIX"] = np.array(v["INTRINSIC_MATRIX"]).reshape([3, 3]).astype(np.float32)
                if "trans_means" in v:
                    v["trans_means"] = np.array(v["trans_means"]).flatten().astype(np.float32)
                if "trans_stds" in v:
                    v["trans_stds"] = np.array(v["trans_stds"]).flatten().astype(np.float32)
                if "class_name_file" in v and v["class_name_file"] != "":
                    with open(v["class_name_file"]) as cf:
                        v["class_name"] = [line.strip() for line in cf.readlines()]
            for vk, vv in v.items():
                config[k][vk] = vv
        else:
            if k == "SCALES":
                config[k][0] = tuple(v)
            else:
                config[k] = v
    return config
