"""CPU: config / flag system keeps the reference's semantics (deepim/config/config.py:128-171)."""
import glob
import os

import numpy as np
import pytest

from conftest import ROOT
from deepim.config.config import config, reset_config, update_config

CFG = os.path.join(ROOT, "mx-deepim_amd", "experiments", "deepim", "cfgs", "deepim_hip_LM_ape_test.yaml")


def test_defaults_and_merge():
    reset_config()
    assert config.TEST.test_iter == 1 and config.network.ROT_COORD == "CAMERA" and config.default.kvstore == "device"
    update_config(CFG)
    assert config.TEST.test_iter == 4 and config.TEST.FAST_TEST is True and config.TEST.UPDATE_MASK == "box_rendered"
    assert config.dataset.INTRINSIC_MATRIX.shape == (3, 3) and config.dataset.INTRINSIC_MATRIX.dtype == np.float32
    assert isinstance(config.network.PIXEL_MEANS, np.ndarray)
    assert config.SCALES[0] == (480, 640)
    assert config.dataset.NUM_CLASSES == 1  # nested keys are merged blindly (config.py:163-164)
    reset_config()


def test_unknown_top_level_key_raises(tmp_path):
    reset_config()
    p = tmp_path / "bad.yaml"
    p.write_text("NUM_GPUS: 4\n")  # the shipped ModelNet YAMLs carry this stale key
    with pytest.raises(ValueError):
        update_config(str(p))
    reset_config()


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference YAMLs only exist in the build container")
def test_reference_yaml_files_load():
    for f in sorted(glob.glob("/root/reference/experiments/deepim/cfgs/*LM_SIXD*.yaml")):
        reset_config()
        update_config(f)
        assert config.symbol == "deepIM_flownet" and config.network.INPUT_MASK and config.TEST.test_iter == 4
    for f in sorted(glob.glob("/root/reference/experiments/deepim/cfgs/*ModelNet*.yaml")):
        reset_config()
        with pytest.raises(ValueError):  # stale top-level NUM_GPUS key: raises in the reference too
            update_config(f)
    reset_config()


def test_param_shapes_match_survey_table():
    from deepim.symbols.deepIM_flownet import deepIM_flownet

    reset_config()
    update_config(CFG)
    sym = deepIM_flownet()
    shp = sym.infer_param_shapes(config)
    n = sum(int(np.prod(s)) for s in shp.values())
    assert n == 57749164  # 57 749 164 fp32 parameters in 46 arrays
    assert shp["flow_conv1_weight"] == (64, 8, 7, 7) and shp["fc6_weight"] == (256, 81920) and shp["deconv4_weight"] == (1026, 256, 4, 4)
    reset_config()
