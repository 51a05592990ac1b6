"""SURVEY 8(b) `dim_refine_4iter`: the resident loop as ONE C entry point (csrc/refiner.hip) for hosts without torch -- against the
Python-driven loop (same kernels, same plans: bit-identical), against the oracle, and captured into a hipGraph."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from scene import make_scene, make_test_config  # noqa: E402
from loop_parity import check_loop, moving_head, oracle_free_and_forced  # noqa: E402

DEV = "cuda:0"


def test_c_resident_loop_matches_python_loop_and_oracle(hip_lib):
    from deepim.core.tester import Predictor, Refiner
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.hip.refiner_capi import CRefiner
    from lib.render_hip.render_py_multi import Render_Py

    cfg = make_test_config(test_iter=4)
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=False)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    moving_head(params, seed=1)   # 3-12 deg / 4-42 mm per iteration (tests/loop_parity.py)
    B = 2
    scene = make_scene(B=B, seed=2333, subdiv=3, n_models=1)
    bl = scene["blobs"]
    rm = Render_Py(None, cfg.dataset.class_name, scene["K"], meshes=scene["models"])
    dev = {k: torch.as_tensor(np.ascontiguousarray(bl[k])).to(DEV) for k in
           ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose", "class_index")}
    keep = {k: v.clone() for k, v in dev.items()}
    cref = CRefiner(cfg, params, rm, B)
    poses_c = cref.refine(*[dev[k] for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose", "class_index")])
    poses_c = poses_c.cpu().numpy().copy()
    se3_c = cref.se3_iter.cpu().numpy().copy()
    assert int(cref.status_iter.abs().sum()) == 0
    for k in dev:   # the input blobs are read, never written
        assert torch.equal(dev[k], keep[k]), k
    # (1) the Python-driven loop enqueues the same launches with the same plans
    pyref = Refiner(cfg, Predictor(cfg, params, B), rm, B, capture_graph=False)
    pyref.load(bl["image_observed"], bl["image_rendered"], bl["mask_observed"], bl["mask_rendered"], bl["src_pose"], bl["class_index"])
    np.testing.assert_array_equal(pyref.refine().cpu().numpy(), poses_c)
    np.testing.assert_array_equal(pyref.se3_iter.cpu().numpy(), se3_c)
    # (2) the oracle
    for b in range(B):
        blobs_b = {k: bl[k][b:b + 1] for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose")}
        free, forced = oracle_free_and_forced(params, scene["models"][0], blobs_b, scene["K"], cfg.network.PIXEL_MEANS, poses_c[:, b])
        pts = scene["models"][0][0].astype(np.float64)
        check_loop(bl["src_pose"][b], poses_c[:, b], se3_c[:, b], free, forced, pts, np.linalg.norm(pts.max(0) - pts.min(0)),
                   tag="C loop pair {}".format(b))
    # (3) run() allocates nothing and does not synchronise: it captures into a hipGraph, and replays reproduce the eager call
    args = [dev[k] for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose", "class_index")]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        cref.refine(*args)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cref.refine(*args)
    cref.poses_iter.zero_()
    g.replay()
    np.testing.assert_array_equal(cref.poses_iter.cpu().numpy(), poses_c)
    g.replay()
    np.testing.assert_array_equal(cref.poses_iter.cpu().numpy(), poses_c)
    # (4) argument errors come back as codes + dim_last_error, not faults
    from lib.hip import capi
    import ctypes

    h = ctypes.c_void_p()
    assert capi.lib().dim_refiner_create(ctypes.byref(h), None, None, None, 0, None) == -1
    assert b"null pointer" in capi.lib().dim_last_error()
    cref.close()
