"""Loop-level checker: the two oracle runs a refinement-loop comparison needs, and the numbers read off them.
Test infrastructure (used by tests/loop_parity.py, smoke() and the `parity` object of bench.py's checker leg -- never by the product).

free   = oracle.refine.refine_pair on its own (deepim/core/tester.py:523-598 restated), from the same initial blobs;
forced = the same loop teacher-forced onto the poses the loop under test produced: iteration k re-renders, rebuilds the masks and
         composes from the pose the tested loop had after iteration k - 1, so every iteration is compared on identical inputs."""
import numpy as np

from . import pose_error
from . import refine as orefine


def oracle_free_and_forced(params, mesh, blobs_b, K, pixel_means, poses_hip_b, test_iter=4, rot_coord="CAMERA", **kw):
    """for one pair: (free, forced), each (poses, se3s[, outputs]) of refine_pair; poses_hip_b (T,3,4) = the tested loop's poses.
    Both consume numpy's global RNG identically (lit renderer: one draw per re-render), so the state is rewound in between."""
    z3, o3 = np.zeros(3), np.ones(3)
    state = np.random.get_state()
    free = orefine.refine_pair(params, mesh, blobs_b, K, pixel_means, z3, o3, rot_coord, test_iter=test_iter, **kw)
    np.random.set_state(state)
    forced = orefine.refine_pair(params, mesh, blobs_b, K, pixel_means, z3, o3, rot_coord, test_iter=test_iter,
                                 forced_poses=np.asarray(poses_hip_b, np.float64), **kw)
    return free, forced


def rot_angle_deg(Ra, Rb):
    c = (np.trace(np.asarray(Ra, np.float64).T @ np.asarray(Rb, np.float64)) - 1.0) / 2.0
    return float(np.degrees(np.arccos(np.clip(c, -1.0, 1.0))))


def loop_numbers(src_pose, poses_hip, se3_hip, free, forced, pts, diam):
    """-> dict for one pair: per-iteration rotation step (deg) of the tested loop, same-state step error max|dpose_hip - dpose_oracle|
    relative to max(1, |dpose|), free-running max|pose_hip - pose_oracle| and max|se3_hip - se3_oracle| (quaternion normalised),
    ADD(final hip, final oracle) / diameter from the same state and free-running (lib/utils/pose_error.py:72-87 `add`)."""
    f_poses, f_se3 = free[0], free[1]
    t_poses = forced[0]
    n_it = len(f_poses)
    prev = np.asarray(src_pose, np.float64)
    rot_step, step_err, free_err, se3_err = [], [], [], []
    for it in range(n_it):
        ph, pt, pf = (np.asarray(x[it], np.float64) for x in (poses_hip, t_poses, f_poses))
        dh, dt = ph - prev, pt - prev
        rot_step.append(rot_angle_deg(prev[:, :3], ph[:, :3]))
        step_err.append(float(np.abs(dh - dt).max() / max(1.0, np.abs(dt).max())))
        free_err.append(float(np.abs(ph - pf).max()))
        if se3_hip is not None:
            g, o = np.asarray(se3_hip[it], np.float64), np.asarray(f_se3[it], np.float64)
            g = np.concatenate([g[:4] / np.linalg.norm(g[:4]), g[4:]])
            o = np.concatenate([o[:4] / np.linalg.norm(o[:4]), o[4:]])
            se3_err.append(float(np.abs(g - o).max()))
        prev = ph
    ph, pt, po = (np.asarray(x[n_it - 1], np.float64) for x in (poses_hip, t_poses, f_poses))
    return {"rot_step_deg": rot_step, "step_err": step_err, "free_pose_err": free_err, "free_se3_err": se3_err,
            "add_same_state_over_d": float(pose_error.add(ph[:, :3], ph[:, 3], pt[:, :3], pt[:, 3], pts) / diam),
            "add_free_over_d": float(pose_error.add(ph[:, :3], ph[:, 3], po[:, :3], po[:, 3], pts) / diam)}
