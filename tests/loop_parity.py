"""Shared checks for the refinement LOOP (tester.py:523-598 + data_pair.py:75-138): scenes whose pose head moves the pose the
way a trained DeepIM does, assertions on the per-iteration STEP, and fault injectors for the negative controls.

The reference's initial pose noise is 15 deg / 1 cm / 5 cm (toolkit/LM6d_1_gen_rendered_pose.py:59,98-117); a trained network
removes a good part of it per iteration.  With the reference initialisation (rot_weight rows 1..3 ~ U(0, 0.01), trans = 0:
deepIM_flownet.py:1053-1065) a random network moves the pose by 0.4 deg / 1-4 mm per iteration -- less than any useful bar --
so the loop tests scale the head: rot rows N(0, 0.2), trans N(0, 0.02) give 3-12 deg and 4-42 mm per iteration
(measured with oracle/refine.py on scene seed 2333)."""
import numpy as np

ROT_SCALE, TRANS_SCALE = 0.2, 0.02


def moving_head(params, seed=1, rot_scale=ROT_SCALE, trans_scale=TRANS_SCALE):
    """in place: a pose head that rotates ~5-10 deg and translates ~1-5 cm per iteration"""
    rng = np.random.RandomState(seed)
    params["trans_weight"] = (rng.randn(3, 256) * trans_scale).astype(np.float32)
    params["rot_weight"][1:] = (rng.randn(3, 256) * rot_scale).astype(np.float32)
    return params


def rot_angle_deg(Ra, Rb):
    c = (np.trace(np.asarray(Ra, np.float64).T @ np.asarray(Rb, np.float64)) - 1.0) / 2.0
    return float(np.degrees(np.arccos(np.clip(c, -1.0, 1.0))))


def steps_of(src_pose, poses):
    """[(rotation step in degrees, translation step in metres)] per iteration"""
    out, prev = [], np.asarray(src_pose, np.float64)
    for p in poses:
        p = np.asarray(p, np.float64)
        out.append((rot_angle_deg(prev[:, :3], p[:, :3]), float(np.linalg.norm(p[:, 3] - prev[:, 3]))))
        prev = p
    return out


def check_loop(src_pose, poses_hip, se3_hip, free, forced, pts, diam, tag="", step_tol=2e-5, add_tol=0.02, free_add_tol=None,
               min_rot_deg=1.5, min_trans_m=2e-3, mean_rot_deg=4.0, verbose=True):
    """free   = (poses, se3) of oracle.refine.refine_pair on the same blobs, running on its own;
    forced = the same with forced_poses = poses_hip: iteration k of the oracle starts from the pose the HIP loop had after
             iteration k-1, i.e. both see the same render request, the same masks, the same src_pose.
    (1) forced, every iteration: |dpose_hip - dpose_oracle| <= step_tol * max(1, |dpose|) on the 12 pose entries (dpose = the change
        of the pose in that iteration: 3-12 deg / 4-42 mm here), and the emitted quaternion direction / translation delta likewise.
        This is the check that guards the FEEDBACK: which pose is rendered, which masks are rebuilt, which pose is composed onto.
    (2) ADD(final hip pose, the oracle's final pose from the same state) < add_tol * diameter ("ADD(-S) vs reference" clause).
    (3) free: reported, and barred (free_add_tol * diameter) only where the caller says so.  A random network with a head this strong
        is an EXPANDING map: a 1e-6 pose difference moves one silhouette pixel, with it a bbox edge and the zoom window, and shows
        as 1e-5 .. 1e-2 one iteration later (measured growth 10-200x per iteration on MI355X, e.g. 1.4e-6 -> 1.1e-5 -> 1.0e-4 ->
        2.4e-2), so two correct implementations drift apart on their own; a trained DeepIM contracts toward the observed pose, and
        with the reference initialisation (tests/test_gpu_refine.py::test_pred_eval_collects_and_scores) the free-running loops
        agree to 1e-7 over four iterations and ADD < 0.02 d is asserted there.
    (4) the scene really moves (otherwise (1) guards nothing).
    Raises AssertionError; returns [(rot step deg, trans step m, forced error, free error)] per iteration."""
    from oracle import pose_error

    f_poses, f_se3 = free[0], free[1]
    t_poses, t_se3 = forced[0], forced[1]
    n_it = len(f_poses)
    st_o = steps_of(src_pose, f_poses)
    rows = []
    prev_h = np.asarray(src_pose, np.float64)
    for it in range(n_it):
        ph, pt, pf = np.asarray(poses_hip[it], np.float64), np.asarray(t_poses[it], np.float64), np.asarray(f_poses[it], np.float64)
        dh, dt = ph - prev_h, pt - prev_h     # both loops started this iteration from prev_h
        err = float(np.abs(dh - dt).max())
        bar = step_tol * max(1.0, float(np.abs(dt).max()))
        free_err = float(np.abs(ph - pf).max())
        rows.append((st_o[it][0], st_o[it][1], err, free_err))
        if verbose:
            print("{} iter {}: step {:.2f} deg / {:.1f} mm; same-input |dpose_hip - dpose_oracle| = {:.2e} (bar {:.1e}); free-running "
                  "|pose_hip - pose_oracle| = {:.2e}".format(tag, it, st_o[it][0], 1e3 * st_o[it][1], err, bar, free_err))
        assert err <= bar, (tag, it, err, bar)
        if se3_hip is not None:
            g, o = np.asarray(se3_hip[it], np.float64), np.asarray(t_se3[it], np.float64)
            # the test graph emits the UN-normalised quaternion (RT_transform.py:143 normalises on use)
            np.testing.assert_allclose(g[:4] / np.linalg.norm(g[:4]), o[:4] / np.linalg.norm(o[:4]), atol=step_tol)
            np.testing.assert_allclose(g[4:], o[4:], atol=step_tol * max(1.0, float(np.abs(o[4:]).max())))
        prev_h = ph
    assert min(r[0] for r in rows) >= min_rot_deg and min(r[1] for r in rows) >= min_trans_m, ("scene does not move enough", rows)
    assert np.mean([r[0] for r in rows]) >= mean_rot_deg, ("scene does not rotate enough", rows)
    ph, pt, po = (np.asarray(x[n_it - 1], np.float64) for x in (poses_hip, t_poses, f_poses))
    e_same = pose_error.add(ph[:, :3], ph[:, 3], pt[:, :3], pt[:, 3], pts)
    e_free = pose_error.add(ph[:, :3], ph[:, 3], po[:, :3], po[:, 3], pts)
    if verbose:
        print("{} ADD(final hip, final oracle): same state {:.2e} d, free-running {:.2e} d".format(tag, e_same / diam, e_free / diam))
    assert e_same < add_tol * diam, (tag, e_same, diam)
    assert free_add_tol is None or e_free < free_add_tol * diam, (tag, e_free, diam)
    return rows


def oracle_free_and_forced(params, mesh, blobs_b, K, pixel_means, poses_hip_b, test_iter=4, rot_coord="CAMERA", **kw):
    """the two oracle loops check_loop wants, for one pair (oracle/loop_check.py, shared with bench.py's checker leg)"""
    from oracle import loop_check

    return loop_check.oracle_free_and_forced(params, mesh, blobs_b, K, pixel_means, poses_hip_b, test_iter=test_iter, rot_coord=rot_coord, **kw)


class StaleRenderPose(object):
    """negative control 1: the render of iteration k is made with the pose of iteration k-1 (a loop that hands the wrong row of
    poses_iter to the rasteriser).  Wraps a render machine; everything else is forwarded."""

    def __init__(self, rm, pose_init):
        self._rm, self._prev = rm, pose_init.clone()

    def __getattr__(self, name):
        return getattr(self._rm, name)

    def render_batch(self, class_index, poses, **kw):
        stale = self._prev.clone()
        self._prev.copy_(poses)
        return self._rm.render_batch(class_index, stale, **kw)


def skip_box_mask(ops):
    """negative control 2: mask_observed is NOT updated from the new rendered mask (data_pair.py:103-114 skipped); the zoom window
    still gets a consistent bbox of the stale mask, so nothing but the skipped update differs.  Returns the replacement function."""

    def fake(bbox, mask, bbox_of_mask=None):
        if bbox_of_mask is not None:
            ops.mask_bbox(mask, 0.3, out=bbox_of_mask)
        return mask

    return fake
