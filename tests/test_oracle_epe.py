"""The oracle's restatement of the test-time flow error (oracle/evaluation.calc_EPE_one_pair, deepim/core/tester.py:719-736) against
the outputs of the reference's OWN function on the reference's own calc_flow (tests/golden/epe_golden.npz, make_golden.py:epe_vectors)."""
import os

import numpy as np

from oracle import evaluation as oev


def test_calc_EPE_one_pair_vs_reference_outputs(golden_dir):
    g = np.load(os.path.join(golden_dir, "flow_golden.npz"))
    e = np.load(os.path.join(golden_dir, "epe_golden.npz"))
    for i in range(e["pred"].shape[0]):
        # the golden flow is stored as float32; the reference's sums ran on calc_flow's float64 output: <= 6e-8 relative per element
        flow, vis, d0 = g["flow"][i].astype(np.float64), g["visible"][i], g["depth_src"][i]
        r = oev.calc_EPE_one_pair({"flow": e["pred"][i].astype("float16")}, {"flow": oev.flow_gt_list(flow, vis, d0)})
        want = e["out"][i]
        assert r["num_all"] == want[1] and r["num_viz"] == want[3] and r["num_vizbg"] == want[5]
        np.testing.assert_allclose([r["epe_all"], r["epe_viz"], r["epe_vizbg"]], want[[0, 2, 4]], rtol=2e-7)
        assert 0 < want[3] < want[5] < want[1]   # the three masks differ, so the three sums are three checks
    # batch form used by the GPU test
    B = 3
    got = oev.epe_of_batch(e["pred"][:B].transpose(0, 3, 1, 2), g["flow"][:B].transpose(0, 3, 1, 2).astype(np.float64), g["visible"][:B, None],
                           g["depth_src"][:B, None])
    np.testing.assert_allclose(got[:, :3], e["out"][:B][:, [0, 2, 4]], rtol=2e-7)
    np.testing.assert_array_equal(got[:, 3:], e["out"][:B][:, [3, 5]])
