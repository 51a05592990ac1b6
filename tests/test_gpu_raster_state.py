"""The rasteriser keeps its z-buffer clear from one render to the next (csrc/raster.hip ResolveHdr: no clear pass) and may skip the
background outside a caller-named dirty box (dim_raster_render_dirty).  Every way that state could go stale must still render the
same pixels as a first render into fresh memory: recycled workspaces, another batch size on the same memory, renders without a colour
output (nothing to shade, keys still to reset), odd widths (one-pass resolve), hipGraph replays."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _setup(B=3, seed=11, subdiv=3):
    from lib.render_hip.render_py_multi import Render_Py
    from lib.utils import synthetic as syn

    models = syn.make_models(seed=seed, n_models=2, subdiv=subdiv)
    cls, gt, init = syn.sample_pairs(seed + 1, B, n_classes=2)
    rm = Render_Py(None, ["a", "b"], syn.LINEMOD_K, meshes=models)
    t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a)).to(DEV, dt)
    return rm, t(cls, torch.int32), t(gt), t(init), syn.plane_means()


def _planes(B, fill=None):
    mk = (lambda *s: torch.empty(s, device=DEV)) if fill is None else (lambda *s: torch.full(s, float(fill), device=DEV))
    return {"image": mk(B, 3, 480, 640), "depth": mk(B, 1, 480, 640), "mask": mk(B, 1, 480, 640), "bgr": mk(B, 480, 640, 3),
            "bbox": torch.zeros((B, 4), dtype=torch.int32, device=DEV)}


def _same(a, b):
    return all(torch.equal(a[k], b[k]) for k in a)


def test_recycled_workspace_and_other_batch_size(hip_lib):
    B = 3
    rm, cls, gt, init, pm = _setup(B)
    ref_gt, ref_init = _planes(B), _planes(B)
    rm.render_batch(cls, gt, plane_means=pm, **ref_gt)
    rm.render_batch(cls, init, plane_means=pm, **ref_init)          # second render on the same workspace: no clear pass ran
    assert not _same(ref_gt, ref_init) and ref_init["mask"].sum() > 1000
    ws = rm._workspace(B)
    # (1) the memory had another life: garbage everywhere, header zeroed by the owner (the contract of include/deepim_hip.h)
    ws.random_(-2 ** 62, 2 ** 62)
    ws[:32].zero_()
    out = _planes(B)
    rm.render_batch(cls, init, plane_means=pm, **out)
    assert _same(out, ref_init)
    # (2) garbage in the header too: it does not carry the magic, so the vertex pass clears
    ws.random_(-2 ** 62, 2 ** 62)
    out = _planes(B)
    rm.render_batch(cls, init, plane_means=pm, **out)
    assert _same(out, ref_init)
    # (3) the same memory under another batch size and back (the layout behind the header depends on B): the header names its
    #     geometry, a call with another one clears first
    rm._ws[2] = ws   # deliberately the SAME tensor for B = 2
    two = _planes(2)
    rm.render_batch(cls[:2], gt[:2], plane_means=pm, **two)
    assert all(torch.equal(two[k], ref_gt[k][:2]) for k in two)
    out = _planes(B)
    rm.render_batch(cls, init, plane_means=pm, **out)
    assert _same(out, ref_init)
    two = _planes(2)
    rm.render_batch(cls[:2], init[:2], plane_means=pm, **two)
    assert all(torch.equal(two[k], ref_init[k][:2]) for k in two)


def test_renders_without_colour_output_leave_the_zbuffer_clear(hip_lib):
    B = 2
    rm, cls, gt, init, pm = _setup(B, seed=23)
    ref = _planes(B)
    rm.render_batch(cls, init, plane_means=pm, **ref)
    d = torch.empty((B, 1, 480, 640), device=DEV)
    rm.render_batch(cls, gt, depth=d)                                    # depth only: nothing to shade, every key still reset
    m, bb = torch.empty_like(d), torch.zeros((B, 4), dtype=torch.int32, device=DEV)
    rm.render_batch(cls, gt, mask=m, bbox=bb)                            # mask + bbox only
    assert torch.equal(m, (d > 0.2).float())
    ys, xs = torch.nonzero(m[0, 0], as_tuple=True)
    assert bb[0].tolist() == [int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())]
    out = _planes(B)
    rm.render_batch(cls, init, plane_means=pm, **out)
    assert _same(out, ref)
    # an output plane 4 bytes off 16-byte alignment takes the one-thread-per-pixel resolve: it resets its keys too
    buf = torch.empty(B * 480 * 640 + 1, device=DEV)
    d_off = buf[1:].view(B, 1, 480, 640)
    rm.render_batch(cls, gt, depth=d_off)
    assert torch.equal(d_off, d)
    out = _planes(B)
    rm.render_batch(cls, init, plane_means=pm, **out)
    assert _same(out, ref)


def test_dirty_box_hint_renders_the_same_planes(hip_lib):
    """the refinement loop's 2nd / 3rd render: the planes hold the previous render, its bbox is the hint"""
    B = 3
    rm, cls, gt, init, pm = _setup(B, seed=31)
    full = _planes(B)
    rm.render_batch(cls, init, plane_means=pm, **full)                   # what an unhinted render of `init` writes
    out = _planes(B)
    rm.render_batch(cls, gt, plane_means=pm, **out)                      # previous render (another pose) ...
    prev_box = out["bbox"].clone()
    rm.render_batch(cls, init, plane_means=pm, clean_bbox=prev_box, **out)   # ... then the hinted re-render into the same planes
    assert _same(out, full)
    # the hint is a promise, not a check: planes that do NOT hold background outside the box keep what they held there
    junk = _planes(B, fill=7.0)
    rm.render_batch(cls, init, plane_means=pm, clean_bbox=prev_box, **junk)
    far = torch.ones((480, 640), dtype=torch.bool, device=DEV)
    for box in (prev_box[0].tolist(), full["bbox"][0].tolist()):
        far[max(box[2] - 1, 0):box[3] + 2, max(box[0] - 4, 0):box[1] + 5] = False
    assert bool((junk["mask"][0, 0][far] == 7.0).all()) and bool((junk["image"][0, 1][far] == 7.0).all())
    # ... and is rewritten inside the named box (whole quads of four pixels) and wherever the new render covers
    x0, x1, y0, y1 = prev_box[0].tolist()
    inbox = torch.zeros((480, 640), dtype=torch.bool, device=DEV)
    inbox[y0:y1 + 1, x0 // 4 * 4:x1 // 4 * 4 + 4] = True
    wrote = inbox | (full["depth"][0, 0] > 0)
    assert torch.equal(junk["mask"][0, 0][wrote], full["mask"][0, 0][wrote]) and torch.equal(junk["bgr"][0][wrote], full["bgr"][0][wrote])
    assert torch.equal(junk["image"][0][:, wrote], full["image"][0][:, wrote])
    # an empty previous box (nothing was in view): everything outside the new coverage is left alone, the object still appears
    empty = torch.tensor([[640, -1, 480, -1]] * B, dtype=torch.int32, device=DEV)
    bg = {"image": (-torch.as_tensor(pm, device=DEV).view(1, 3, 1, 1)).expand(B, 3, 480, 640).contiguous(), "depth": torch.zeros((B, 1, 480, 640), device=DEV),
          "mask": torch.zeros((B, 1, 480, 640), device=DEV), "bgr": torch.zeros((B, 480, 640, 3), device=DEV),
          "bbox": torch.zeros((B, 4), dtype=torch.int32, device=DEV)}
    rm.render_batch(cls, init, plane_means=pm, clean_bbox=empty, **bg)
    assert _same(bg, full)
    # clean_bbox and bbox must be two arrays
    from lib.hip.capi import DeepIMHipError

    with pytest.raises(DeepIMHipError):
        rm.render_batch(cls, init, plane_means=pm, clean_bbox=bg["bbox"], **bg)


def test_lit_renderer_hint_and_graph_replay(hip_lib):
    from lib.render_hip.render_py_light_modelnet_multi import Render_Py_Light_ModelNet_Multi, vertex_normals
    from lib.utils import synthetic as syn

    B = 2
    models = syn.make_models(seed=5, n_models=2, subdiv=3)
    meshes = [(v, vertex_normals(v, f).astype(np.float32), t, f) for v, t, f, _ in models]
    rm = Render_Py_Light_ModelNet_Multi(None, np.full((32, 32, 3), 180, np.uint8), syn.LINEMOD_K, 640, 480, 0.25, 6.0, brightness_ratios=[0.7],
                                        meshes=meshes, device=DEV)
    cls, gt, init = syn.sample_pairs(9, B, n_classes=2)
    t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a)).to(DEV, dt)
    cls, gt, init, pm = t(cls, torch.int32), t(gt), t(init), syn.plane_means()
    full = _planes(B)
    rm.render_batch(cls, init, plane_means=pm, **full)
    out = _planes(B)
    rm.render_batch(cls, gt, plane_means=pm, **out)
    prev = out["bbox"].clone()
    rm.render_batch(cls, init, plane_means=pm, clean_bbox=prev, **out)
    assert _same(out, full)
    # captured: two renders per replay (gt, then init hinted with gt's box), replayed three times
    a, b2 = _planes(B), torch.zeros((B, 4), dtype=torch.int32, device=DEV)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        rm.render_batch(cls, gt, plane_means=pm, image=a["image"], depth=a["depth"], mask=a["mask"], bgr=a["bgr"], bbox=b2)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rm.render_batch(cls, gt, plane_means=pm, image=a["image"], depth=a["depth"], mask=a["mask"], bgr=a["bgr"], bbox=b2)
        rm.render_batch(cls, init, plane_means=pm, clean_bbox=b2, **a)
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        assert _same(a, full)
