"""deepim/test.py and deepim/train.py (the reference's command lines) end to end on tiny synthetic runs, as child processes."""
import glob
import json
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


def test_train_entry_point_from_files(hip_lib, tmp_path):
    """deepim/train.py --from_files: the reference's data path (pairdb -> TrainDataLoader, train.py:136) on a synthetic dataset written to
    disk -- epoch 0 decodes the PNG files, every later epoch is served by the decoded-pixel cache in HBM"""
    import re

    cwd = str(tmp_path)
    out = _run("train.py", ["--num_pairs", "32", "--frequent", "1", "--from_files", os.path.join(cwd, "dataset")], cwd)
    assert "training from files: 32 pairs" in out and "Epoch[7]" in out and "Train-Flow_L2Loss=" in out
    assert len(glob.glob(os.path.join(cwd, "dataset", "pairs", "*-color.png"))) == 32
    m = re.findall(r"pixel cache after epoch (\d+): (\d+) files, \d+ MB, (\d+) hits / (\d+) misses", out)
    assert len(m) == 8 and int(m[0][1]) == 5 * 32 and int(m[0][3]) == 5 * 32       # first epoch: every file decoded once
    assert int(m[-1][3]) == 5 * 32 and int(m[-1][2]) == 7 * 5 * 32                 # later epochs: hits only
    assert not any(v != v for v in map(float, re.findall(r"Train-Flow_L2Loss=([-+0-9.e]+|nan)", out)))


def test_two_rank_training_replicas_stay_identical(hip_lib, tmp_path):
    """world_size 2 on ONE card (gloo carries the gradient sum; on a multi-GPU node the same code runs over RCCL): after 2 epochs x 4
    updates with different pairs per rank both replicas hold bit-identical weights (deepim/train.py asserts it and prints the digest).
    Run twice: gradient buckets overlapped with backward (default) and one flat all-reduce inside update() -- with two ranks every sum
    has two addends, so the two schedules must end in the SAME weights, digit for digit."""
    import re

    cfg2 = os.path.join(str(tmp_path), "two_epochs.yaml")
    with open(CFG) as f:
        text = f.read().replace("end_epoch: 8", "end_epoch: 2")
    with open(cfg2, "w") as f:
        f.write(text)
    digests = {}
    for overlap, port in (("1", "29517"), ("0", "29519")):
        env = dict(os.environ)
        env.update(DIM_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", DIM_OVERLAP_ALLREDUCE=overlap)
        cwd = os.path.join(str(tmp_path), "overlap" + overlap)
        os.makedirs(cwd)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", port, os.path.join(PKG, "deepim", "train.py"), "--cfg", cfg2, "--gpus", "0,0", "--num_pairs", "64",
                            "--max_batches", "1", "--frequent", "1"], cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           universal_newlines=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
        m = re.search(r"replicas identical on 2 ranks \(digest ([-+0-9.e]+)\)", r.stdout)
        assert m, r.stdout[-2000:]
        digests[overlap] = m.group(1)
    assert digests["1"] == digests["0"], digests


def test_two_rank_test_run_scores_the_union_of_the_shards(hip_lib, tmp_path):
    """deepim/test.py --gpus 0,0 (the script launches its own torch.distributed.run job), 2 ranks on one card: each rank refines its own 32 of the 64 pairs (no collective on
    the data path), the per-class pose lists are merged once (all_gather_object) and every metric is over all 64 pairs; ONE result
    cache is written."""
    import pickle

    # ONE command, like the reference's `--gpus 0,1,2,3`: the script starts its own rank per named GPU (here twice card 0)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(PKG, "deepim", "test.py"), "--cfg", CFG, "--gpus", "0,0", "--num_pairs", "64"],
                       cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    files = glob.glob(os.path.join(str(tmp_path), "output", "deepim_hip", "*", "synthetic_val_ape", "*_results.pkl"))
    assert len(files) == 1, files
    rot_err, trans_err, poses_est, poses_gt = pickle.load(open(files[0], "rb"))
    assert len(poses_est[0][3]) == 64 and len(poses_gt[0][0]) == 64


def test_bench_gpus_2_one_command_two_ranks_on_one_card(hip_lib):
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts its own 2 ranks (child torch.distributed.run job) and rank 0 prints
    ONE line with "n_gpus": 2 -- rehearsed here with both ranks on this box's single card (gloo carries the barrier, the MAX over
    ranks and the gradient sums of the `train` object; on a multi-GPU node the same command runs over RCCL)."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--train-steps", "1", "--profile-steps", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "rank 0 of 2 joined" in r.stderr and "rank 1 of 2 joined" in r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_pairs"] == 32 and out["value"] > 0
    assert "error" not in out["train"], out["train"]
    assert out["train"]["global_pairs"] == 2 * out["train"]["pairs_per_gpu"] and out["train"]["f32"]["finite"] and out["train"]["bf16"]["finite"]
    # the line says WHICH devices joined: one identity per rank (rank order), here both on this box's one card -- which the gloo
    # rehearsal allows and an RCCL run refuses (tests/test_dist_gloo.py) --, every rank's own step time next to the MAX
    assert [r["rank"] for r in out["ranks"]] == [0, 1] and out["dist_backend"] == "gloo" and out["distinct_devices"] == 1
    assert all(r["cus"] == 256 and "gfx950" in r["arch"] and r["pid"] > 0 for r in out["ranks"]) and out["ranks"][0]["pid"] != out["ranks"][1]["pid"]
    assert len(out["ms_per_step_per_rank"]) == 2 and max(out["ms_per_step_per_rank"]) == pytest.approx(out["ms_per_step"], rel=1e-3)
    # and what the gradient sum costs: bytes and time of every bucket (f32 buckets 231 MB in all, bf16 half of it)
    for dt, nbytes in (("f32", 4), ("bf16", 2)):
        bk = out["train"][dt]["allreduce_buckets"]
        assert len(bk) == 3 and all(b["ms"] > 0 and b["bytes"] == (b["end"] - b["begin"]) * nbytes for b in bk), bk
        assert out["train"][dt]["allreduce_bytes_per_update"] == sum(b["bytes"] for b in bk) and len(out["train"][dt]["iteration_ms_per_rank"]) == 2


def test_bench_line_names_the_arithmetic_and_times_the_f32_pipe_beside_it(hip_lib):
    """one rank, short run: the dominant kernel is a three-term kernel priced against the bf16 peak / 6, the same loop on the f32 pipe
    is timed in the same process, both arithmetics give the same first step to f32 rounding, and the parity object is green"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--head-epochs", "0", "--no-train",
                        "--no-train-files", "--no-fresh-batch", "--cpu-pairs", "2", "--parity-pairs", "2", "--profile-steps", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["dtype"] == "f32" and "three bf16 terms" in out["config"]["plane_gemm_arithmetic"]
    rf = out["roofline"]
    assert "_split_kernel" in rf["kernel"] and rf["peak"] == pytest.approx(2500.0 / 6, rel=1e-3) and 0.05 < rf["frac"] < 1.0
    assert rf["mfma_executed_TFLOP/s"] == pytest.approx(6 * rf["achieved"], rel=1e-2) and rf["f32_pipe_peak"] == pytest.approx(157.3)
    f32 = out["f32_pipe"]
    assert "error" not in f32 and f32["value"] > 0 and out["value"] > f32["value"]          # the three-term arithmetic is the faster one
    assert f32["first_iteration_se3_max_abs_diff_to_headline"] < 1e-4                         # ... and the same one, to f32 rounding
    assert out["parity"]["ok"] and out["parity"]["max_step_err"] <= out["parity"]["step_err_bar"]
