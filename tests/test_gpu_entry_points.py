"""deepim/test.py and deepim/train.py (the reference's command lines) end to end on tiny synthetic runs, as child processes."""
import glob
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mx-deepim_amd")
CFG = os.path.join(PKG, "experiments", "deepim", "cfgs", "deepim_hip_LM_ape_test.yaml")


def _run(script, extra, cwd):
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(PKG, "deepim", script), "--cfg", CFG, "--gpus", "0"] + extra, cwd=cwd, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


def test_train_then_test_entry_points(hip_lib, tmp_path):
    cwd = str(tmp_path)  # output_path in the YAML is relative: everything lands under the temporary directory
    out = _run("train.py", ["--num_pairs", "32", "--max_batches", "1", "--frequent", "1"], cwd)
    assert "Epoch[0]" in out and "Train-Flow_L2Loss=" in out and "Train-PointMatchingLoss=" in out and "saved" in out
    ckpts = sorted(glob.glob(os.path.join(cwd, "output", "deepim_hip", "*", "synthetic_train_ape", "deepim_hip_LM_ape-*.params")))
    assert len(ckpts) == 8 and ckpts[-1].endswith("-0008.params")         # end_epoch 8, one checkpoint per epoch
    assert os.path.exists(ckpts[-1].replace(".params", ".states.npz"))
    # the test script finds prefix-<test_epoch>.params written by the training run (test_epoch 8) and refines 16 pairs
    out = _run("test.py", ["--num_pairs", "16"], cwd)
    assert "loaded" in out and "-0008.params" in out
    assert "evaluating pose" in out and "add performance over 1 classes" in out and "refined 16 pairs x 4 iterations" in out
    assert glob.glob(os.path.join(cwd, "output", "deepim_hip", "*", "synthetic_val_ape", "*_results.pkl"))
