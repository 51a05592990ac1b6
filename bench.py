#!/usr/bin/env python
"""bench.py -- pose-refinements/sec (4 iterations, 480x640) of the HIP refinement path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one 4-iteration refinement of one batch of BATCH_PAIRS=16 synthetic LINEMOD-'ape' pairs per
GPU (BASELINE.json configs[1]: FlowNetS-backbone forward x4 + SE(3) compose x4 + HIP rasteriser x3 +
zoom x4), inputs resident in HBM, whole loop replayed as one hipGraph.  Pairs are independent, so ranks
shard them with no data-path collective ("weak" scaling: 16 pairs per GPU).

Prints ONE JSON line (rank 0) with the contract keys plus
  "roofline":     dominant kernel's algorithmic TFLOP/s vs its matrix-pipe roof -- the dense f32 MFMA peak for the f32-pipe
                  kernels; for the three-term plane GEMMs (f32 operands split into three bf16 terms, six MFMA products per
                  multiply, f32 accumulate: csrc/wino_gemm_split.hip) the dense bf16 peak / 6 -- from HIP events on the
                  launch stream, averaged over the timed launches of that kernel,
  "cpu_baseline": the CPU oracle (torch-CPU convs + numpy zoom + C rasteriser, batch 1 like the reference
                  loop) timed on this box's host cores on a bounded sample (rank 0, N == 1 only).
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "mx-deepim_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32 matrix peak (v_mfma_f32_32x32x2_f32)
HBM_PEAK_GBS = 8000.0
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak (the 5 PF headline figure includes 2:1 sparsity)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-pairs", type=int, default=None, help="pairs per GPU (default: TEST.BATCH_PAIRS of the cfg = 16)")
    ap.add_argument("--cfg", default=os.path.join(PKG, "experiments", "deepim", "cfgs", "deepim_hip_LM_ape_test.yaml"))
    ap.add_argument("--subdiv", type=int, default=5, help="icosphere subdivisions of the synthetic mesh (5 = 20480 triangles)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--full-graph", action="store_true",
                    help="not the headline: run the full test graph (decoder + flow / mask heads every iteration, TEST.FAST_TEST = False)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                    "multi-rank path on a box with fewer GPUs than ranks, together with DIM_BENCH_DEVICE)")
    ap.add_argument("--no-winograd", action="store_true", help="run every encoder layer through the direct kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-deadline", type=float, default=240.0, help="seconds the `train` object may take with several ranks before the line is printed without it")
    ap.add_argument("--no-fresh-batch", action="store_true", help="skip the (non-headline) `fresh_batch` object: a new batch uploaded every step")
    ap.add_argument("--no-variants", action="store_true", help="skip the (non-headline) `full_graph` and `modelnet_lit` objects (BASELINE configs[3] / [4] per GPU)")
    ap.add_argument("--no-train", action="store_true", help="skip the (non-headline) `train` object: timed training iterations at 16 pairs per GPU")
    ap.add_argument("--no-train-files", action="store_true", help="skip the (non-headline) `train_fresh_batch` object: training fed from image files")
    ap.add_argument("--train-steps", type=int, default=5, help="timed training iterations per phase for the `train` object")
    ap.add_argument("--autotune", action="store_true", help="time tile/split-K candidates per layer first (untimed); default: fixed plan")
    ap.add_argument("--cpu-pairs", type=int, default=32, help="bounded CPU-baseline sample (pairs refined by the oracle)")
    ap.add_argument("--head-epochs", type=int, default=250, help="untimed, before the timed region: train the network on this rank's own benchmark "
                    "pairs for this many epochs (fit_batch, TRAIN_ITER_SIZE 4, Adam 1e-4) so that the refinement loop CONTRACTS like a trained "
                    "DeepIM; 0 = seeded initialisation with a scaled random pose head that moves the pose 3-12 deg per iteration")
    ap.add_argument("--parity-pairs", type=int, default=4, help="pairs of the benchmark batch checked against the oracle loop after the timed region "
                    "(teacher-forced + free-running): the `parity` object; a value over its bar makes the exit code non-zero")
    ap.add_argument("--f32-pipe", action="store_true", help="Winograd plane GEMMs on the f32 matrix pipe (v_mfma_f32_32x32x2_f32) instead of "
                    "the default three-bf16-term arithmetic (csrc/wino_gemm_split.hip); the default run reports this variant as `f32_pipe`")
    ap.add_argument("--profile-steps", type=int, default=3, help="eager steps with per-layer HIP events for the roofline object")
    return ap.parse_args()


def host_cores():
    """cores this process may actually use: cgroup quota if any (the GPU box gives 16 per GPU), else affinity"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 16) if n > 64 else n  # no quota visible on a 256-thread host: stay within the per-GPU share


def cpu_baseline(cfg, params, models, batch, n_pairs):
    """oracle/refine.py on the first n_pairs pairs, batch 1 each (the reference loop is batch-1: tester.py:124)."""
    from oracle import native, refine as orefine  # checker only -- never the measured product path

    native.build()
    torch.set_num_threads(host_cores())
    idx = torch.arange(n_pairs, device=batch["src_pose"].device) % batch["src_pose"].shape[0]
    blobs = {k: batch[k][idx].cpu().numpy() for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose")}
    cls = batch["class_index"][idx].cpu().numpy()
    K = np.asarray(cfg.dataset.INTRINSIC_MATRIX, dtype=np.float32)
    z3, o3 = np.zeros(3), np.ones(3)
    one = {k: v[:1] for k, v in blobs.items()}
    orefine.refine_pair(params, models[int(cls[0])], one, K, cfg.network.PIXEL_MEANS, z3, o3, cfg.network.ROT_COORD, test_iter=1)  # warm
    t0 = time.time()
    finals = []
    for i in range(n_pairs):
        bi = {k: v[i:i + 1] for k, v in blobs.items()}
        poses, _ = orefine.refine_pair(params, models[int(cls[i])], bi, K, cfg.network.PIXEL_MEANS, z3, o3, cfg.network.ROT_COORD,
                                       test_iter=int(cfg.TEST.test_iter))
        finals.append(poses[-1])
    dt = time.time() - t0
    cpu_baseline.final_poses = np.array(finals)   # the checker leg scores them (ADD vs the HIP loop's final poses)
    return {"value": n_pairs / dt, "unit": "pose-refinements/sec", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": "{} pairs x {} iters, batch 1 (torch-CPU f32 convs + numpy zoom + C rasteriser), {:.1f} s".format(
                n_pairs, int(cfg.TEST.test_iter), dt)}


def encoder_roofline(net, b, test_iter, profile_steps):
    """roofline object of the dominant kernel of the network forward on the blobs `b`: HIP events on the launch stream around every
    conv launch of `profile_steps` x `test_iter` eager forwards, grouped by kernel symbol"""
    # ---- roofline of the dominant kernel: HIP events on the launch stream around every conv launch (eager)
    events = {}
    for _ in range(profile_steps):
        for it in range(test_iter):
            net.zoom(b)
            net.encoder(events=events)
            net.head()
    torch.cuda.synchronize()
    per_kernel = {}
    TILE_SYM = {1: "128, 128, 2, 2", 2: "128, 64, 2, 2", 3: "64, 64, 2, 2", 4: "128, 128, 2, 4", 5: "128, 256, 2, 4", 6: "160, 128, 1, 4",
                7: "96, 128, 1, 4"}
    SPLIT_SYM = {5: "128, 8", 4: "128, 4", 7: "96, 4"}
    from lib.hip import ops as _ops
    split_on = _ops.get_winograd_split()
    for name, evs in events.items():
        info = net.layer_info[name]
        for ev in evs:
            tag, e0, e1 = ev[:3]
            n_launch = ev[3] if len(ev) > 3 else 1  # auto mode: two conv launches (+ a reduce) inside one event pair
            if tag == "conv":  # the symbol rocprofv3 --kernel-trace reports for this launch
                if info.get("winograd"):  # batched GEMM of the Winograd path: the multiply-adds that launch really executes
                    if split_on and info["wino_tile"] in SPLIT_SYM:   # f32 operands as three bf16 terms (csrc/wino_gemm_split.hip)
                        kname = "dim::wino_gemm_split_kernel<{}>".format(SPLIT_SYM[info["wino_tile"]])
                    else:
                        kname = "dim::wino_gemm_kernel<{}>".format(TILE_SYM[info["wino_tile"]])
                    flops, nbytes = info["wino_flops"], info["wino_gemm_bytes"]
                else:
                    kname = ("dim::conv1_halo_split_kernel<7, 7>" if split_on else "dim::conv1_halo_kernel<7, 7>") if info["tile"] == 6 else \
                        "dim::conv_fwd_kernel<{}, {}>".format(TILE_SYM[info["tile"]], "true" if info["cin"] == 8 else "false")
                    flops, nbytes = info["flops"], info["min_bytes"]
            elif tag == "fc":  # fc6 weight stream (+ its partial-sum reduce inside the event pair)
                kname, flops, nbytes = ("dim::fc_stream16_kernel" if net.B <= 16 else "dim::fc_stream_kernel") + " + fc_reduce_kernel", info["flops"], info["min_bytes"]
            elif tag in ("wino_in", "wino_out"):
                kname, flops = info["wino_in_kernel" if tag == "wino_in" else "wino_out_kernel"], 0.0
                nbytes = info["wino_in_bytes" if tag == "wino_in" else "wino_out_bytes"]  # HBM-bound: read x + write V / read M + write y
            else:
                kname, flops, nbytes = "dim::splitk_reduce_kernel", 0.0, 0.0
            k = per_kernel.setdefault(kname, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "layers": []})
            k["ms"] += e0.elapsed_time(e1)
            k["flops"] += flops
            k["bytes"] += nbytes
            k["launches"] += n_launch
            if name not in k["layers"]:
                k["layers"].append(name)
    dom_name, dom = max(per_kernel.items(), key=lambda kv: kv[1]["ms"])
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    nfwd = profile_steps * test_iter
    # HBM-side bytes per launch of that kernel: cannot be read live (rocprofv3 --pmc is its own run), so the figure comes
    # from the committed PMC passes over this same command (profiles/README.md; tools/pmc_traffic.py applies the guide's
    # gfx950 correction 2*FETCH_SIZE + WRITE_SIZE); null when the summary is absent or is for another kernel
    traffic, traffic_src = None, None
    for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        rec = json.load(open(cand)).get("kernels", {}).get(dom_name)
        if rec:
            traffic, traffic_src = rec["hbm_bytes_per_launch"], os.path.relpath(cand, ROOT)
            break
    # the matrix-pipe roof of the dominant kernel.  wino_gemm_kernel / conv kernels: the dense f32 MFMA peak.  wino_gemm_split_kernel
    # multiplies f32 operands as three bf16 terms each and keeps six term products per multiply on v_mfma_f32_32x32x16_bf16: its roof
    # for ALGORITHMIC flops is the dense bf16 peak / 6 (it executes 6 x the algorithmic flops on the bf16 pipe)
    is_split = "_split_kernel" in dom_name
    peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if is_split else F32_MFMA_PEAK_TFLOPS
    roofline = {"bound": "mfma", "kernel": dom_name, "layers": dom["layers"], "achieved": round(achieved, 2),
                "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                "traffic_unit": "bytes/launch (PMC, 2*FETCH_SIZE+WRITE_SIZE)", "traffic_source": traffic_src,
                "min_bytes_per_launch_avg": round(dom["bytes"] / dom["launches"]),
                "avg_launch_ms": round(dom["ms"] / dom["launches"], 4), "launches_timed": dom["launches"],
                "gflop_per_launch_avg": round(dom["flops"] / dom["launches"] / 1e9, 3),
                "all_kernels": {k: {"TFLOP/s": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), "algorithmic_GB/s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                                    "avg_launch_ms": round(v["ms"] / v["launches"], 4),
                                    "ms_per_forward": round(v["ms"] / nfwd, 4)} for k, v in per_kernel.items()},
                "conv_stack_ms_per_forward": round(sum(v["ms"] for v in per_kernel.values()) / nfwd, 3)}
    if is_split:
        roofline.update({"arithmetic": "f32 operands, each the exact sum of three bf16 terms; six term products per multiply on "
                                       "v_mfma_f32_32x32x16_bf16, f32 accumulate (error <= 3 * 2^-27 per product; DIM_WINO_SPLIT=0 = f32 pipe)",
                         "peak_is": "dense bf16 MFMA peak {} / 6 products".format(BF16_MFMA_PEAK_TFLOPS),
                         "mfma_executed_TFLOP/s": round(6 * achieved, 1), "mfma_executed_frac_of_bf16_peak": round(6 * achieved / BF16_MFMA_PEAK_TFLOPS, 4),
                         "f32_pipe_peak": F32_MFMA_PEAK_TFLOPS, "achieved_over_f32_pipe_peak": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
                         "hbm_algorithmic_GB/s": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, 1),
                         "hbm_frac_of_peak": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})

    return roofline


def train_head(cfg, params, models, rm, B, dev, seed, epochs, lr=1e-4):
    """Untimed set-up: the behavioural test's recipe (tests/test_gpu_learns.py) on THIS rank's benchmark pairs -- fit_batch with
    TRAIN_ITER_SIZE inner iterations (forward, losses, backward, Adam update, re-render + re-label), `epochs` passes over the one
    batch -- so that the timed loop runs weights under which it contracts toward the observed pose (a trained DeepIM), not a random
    head.  No checkpoint ships with the repo (57.75 M parameters); no data leaves the GPU.  -> (params, info)"""
    from deepim.core.module import MutableModule, fit_epochs
    from lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti
    from lib.utils import synthetic as syn

    keep = cfg.TRAIN.optimizer
    cfg.TRAIN.optimizer = "adam"
    try:
        t0 = time.perf_counter()
        tb = syn.build_device_train_batch(rm, B, seed=seed, models=models, n_classes=len(models), pixel_means=cfg.network.PIXEL_MEANS,
                                          npts=int(cfg.train_iter.NUM_3D_SAMPLE), device=dev)
        mod = MutableModule(cfg, params, B, device=dev)
        upd = batchUpdaterPyMulti(cfg, 480, 640, render_machine=rm)
        hist = fit_epochs(mod, [tb], upd, lr, epochs)
        torch.cuda.synchronize()
        out = mod.get_params()
        info = {"kind": "trained in-process on the benchmark pairs (untimed)", "optimizer": "adam", "lr": lr, "epochs": int(epochs),
                "updates": int(mod.num_update), "pairs": B, "seconds": round(time.perf_counter() - t0, 1),
                "flow_pm_loss_first_epoch": round(float(hist[0, :, :2].sum()), 1), "flow_pm_loss_last_epoch": round(float(hist[-1, :, :2].sum()), 1)}
        del mod, upd, tb
        torch.cuda.empty_cache()
        return out, info
    finally:
        cfg.TRAIN.optimizer = keep


def pose_error_vs_gt(src_pose, poses_iter, pose_gt):
    """mean rotation (deg) / translation (mm) error against the ground truth: initial, after iteration 1 .. test_iter"""
    def err(p):
        R = torch.einsum("bij,bkj->bik", p[:, :, :3].double(), pose_gt[:, :, :3].double())
        c = ((R.diagonal(dim1=1, dim2=2).sum(1) - 1.0) / 2.0).clamp(-1.0, 1.0)
        return float(torch.rad2deg(torch.acos(c)).mean()), float((p[:, :, 3].double() - pose_gt[:, :, 3].double()).norm(dim=1).mean() * 1e3)

    rows = [err(src_pose)] + [err(poses_iter[i]) for i in range(poses_iter.shape[0])]
    return {"rot_deg": [round(r, 3) for r, _ in rows], "trans_mm": [round(t, 2) for _, t in rows]}


def parity_leg(cfg, params, models, batch, poses_hip, se3_hip, n_pairs, trained):
    """Checker leg, after the timed region (the metric's second clause, "ADD(-S) vs reference"): the oracle loop on the first n_pairs
    pairs of the benchmark batch, teacher-forced onto the HIP loop's poses and free-running (oracle/loop_check.py).
    Bars: from identical state every iteration's step within 2e-5 max(1, |step|) (1e-3 under trained weights, which answer a discrete
    event downstream of identical renders with up to 1e-4: tests/test_gpu_learns.py) and ADD < 0.02 d; free-running ADD < 0.02 d only
    under trained (contracting) weights -- a random moving head is an expanding map and two correct loops drift apart on their own."""
    from oracle import loop_check, native  # checker only -- never the measured product path

    native.build()
    K = np.asarray(cfg.dataset.INTRINSIC_MATRIX, dtype=np.float32)
    cls = batch["class_index"].cpu().numpy()
    host = {k: batch[k][:n_pairs].cpu().numpy() for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose")}
    rows = []
    for b in range(n_pairs):
        mesh = models[int(cls[b])]
        pts = mesh[0].astype(np.float64)
        diam = float(np.linalg.norm(pts.max(0) - pts.min(0)))
        free, forced = loop_check.oracle_free_and_forced(params, mesh, {k: v[b:b + 1] for k, v in host.items()}, K, cfg.network.PIXEL_MEANS,
                                                         poses_hip[:, b], test_iter=int(cfg.TEST.test_iter), rot_coord=cfg.network.ROT_COORD)
        rows.append(loop_check.loop_numbers(host["src_pose"][b], poses_hip[:, b], se3_hip[:, b], free, forced, pts, diam))
    step_bar = 1e-3 if trained else 2e-5
    out = {"pairs": n_pairs, "oracle": "oracle/refine.py (CPU restatement of tester.py:523-598), teacher-forced + free-running",
           "max_step_err": float(max(max(r["step_err"]) for r in rows)), "step_err_bar": step_bar,
           "add_same_state_over_d": float(max(r["add_same_state_over_d"] for r in rows)),
           "add_free_over_d": float(max(r["add_free_over_d"] for r in rows)), "add_bar_over_d": 0.02,
           "free_se3_err_median": float(np.median([r["free_se3_err"] for r in rows])),
           "free_se3_err_max": float(max(max(r["free_se3_err"]) for r in rows)),
           "rot_step_deg": [round(float(np.mean([r["rot_step_deg"][i] for r in rows])), 3) for i in range(len(rows[0]["rot_step_deg"]))],
           "free_running_barred": bool(trained)}
    out["ok"] = bool(out["max_step_err"] <= step_bar and out["add_same_state_over_d"] < 0.02 and (not trained or out["add_free_over_d"] < 0.02))
    return out


def variant_bench(kind, cfg, dev, rank, steps, warmup, profile_steps, subdiv):
    """Non-headline objects for BASELINE configs[3] / [4] at their per-GPU shape (the global sizes are 8 such ranks; ranks do not
    interact at test time):
      full_graph    FAST_TEST off: decoder + mask + flow heads produced every iteration (deepIM_flownet.py:840-954, read at
                    tester.py:485-491), 8 Occlusion-LINEMOD-like classes of 20 480 triangles resident, 16 pairs, 4 iterations
      modelnet_lit  256 gray-textured meshes resident in one HBM table, the lit renderer in the loop (tester.py:204-242,
                    render_py_light_modelnet_multi.py:82-231), 32 pairs, 4 iterations
    Same timing as the headline: hipGraph replays bracketed by device syncs; its own roofline entry from an eager event pass."""
    from deepim.core.tester import Predictor, Refiner
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.utils import synthetic as syn

    keep = (cfg.TEST.FAST_TEST, list(cfg.dataset.class_name), cfg.dataset.dataset)
    try:
        test_iter = int(cfg.TEST.test_iter)
        if kind == "full_graph":
            from lib.render_hip.render_py_multi import Render_Py

            n_cls, B = 8, 16
            cfg.TEST.FAST_TEST = False
            cfg.dataset.class_name = ["occ{:d}".format(i) for i in range(n_cls)]
            models = syn.make_models(seed=77, n_models=n_cls, subdiv=subdiv)
            rm = Render_Py(None, cfg.dataset.class_name, cfg.dataset.INTRINSIC_MATRIX, zNear=cfg.dataset.ZNEAR, zFar=cfg.dataset.ZFAR,
                           device=dev, meshes=models)
            tri = [int(m[2].shape[0]) for m in models]
        else:
            from lib.render_hip.render_py_light_modelnet_multi import Render_Py_Light_ModelNet_Multi, vertex_normals

            n_cls, B = 256, 32
            cfg.dataset.dataset = "ModelNet_v1"
            cfg.dataset.class_name = ["m{:03d}".format(i) for i in range(n_cls)]
            models = []
            for i in range(n_cls):   # 1 280 / 5 120 triangles alternating: irregular mesh_table offsets
                models += syn.make_models(seed=9000 + i, n_models=1, subdiv=3 + i % 2)
            gray = np.full((32, 32, 3), 180, np.uint8)
            meshes = [(v, vertex_normals(v, f).astype(np.float32), t, f) for v, t, f, _ in models]
            rm = Render_Py_Light_ModelNet_Multi(None, gray, cfg.dataset.INTRINSIC_MATRIX, 640, 480, cfg.dataset.ZNEAR, cfg.dataset.ZFAR,
                                                brightness_ratios=[0.7], meshes=meshes, device=dev)
            tri = [int(m[3].shape[0]) for m in meshes]
        sym = deepIM_flownet()
        sym.get_symbol(cfg, is_train=False)
        params = sym.init_weights(cfg, {}, {}, seed=0)
        _rng = np.random.RandomState(1)
        params["trans_weight"] = (_rng.randn(3, 256) * 0.02).astype(np.float32)
        params["rot_weight"][1:] = (_rng.randn(3, 256) * 0.2).astype(np.float32)
        if "mask_conv3_weight" in params:
            params["mask_conv3_weight"] = (_rng.randn(1, 770, 3, 3) * 0.05).astype(np.float32)
        pred = Predictor(cfg, params, B, device=dev)
        batch = syn.build_device_batch(rm, B, seed=3000 + rank, n_classes=n_cls, pixel_means=cfg.network.PIXEL_MEANS, device=dev)
        refiner = Refiner(cfg, pred, rm, B, capture_graph=True)
        np.random.seed(99)
        refiner.load(batch["image_observed"], batch["image_rendered"], batch["mask_observed"], batch["mask_rendered"], batch["src_pose"],
                     batch["class_index"])
        for _ in range(warmup):
            refiner.refine()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            refiner.refine()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        poses = refiner.poses_iter
        res = {"value": round(B * steps / el, 2), "unit": "pose-refinements/sec", "ms_per_step": round(el / steps * 1e3, 3), "steps": steps,
               "pairs_per_gpu": B, "test_iter": test_iter, "resident_classes": n_cls, "triangles_per_mesh": [min(tri), max(tri)],
               "classes_in_batch": int(len(set(batch["class_index"].cpu().numpy().tolist()))),
               "status_flags": int(refiner.status_iter.abs().sum().item()), "finite": bool(torch.isfinite(poses).all().item()),
               "gflop_per_refinement": round(pred.net.flops_per_forward() * test_iter / B / 1e9, 2)}
        # one render of the whole batch alone (HIP events on the launch stream): the rasteriser's share of a step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        extra = {"light_intensity": refiner.light_int[0]} if refiner.lit else {}
        kw = dict(image=refiner.batch["image_rendered"], mask=refiner.batch["mask_rendered"], bbox=refiner.bbox,
                  plane_means=pred.net.plane_means, mask_thr=0.2, **extra)
        rm.render_batch(batch["class_index"], poses[0], **kw)
        e0.record()
        for _ in range(10):
            rm.render_batch(batch["class_index"], poses[0], **kw)
        e1.record()
        e1.synchronize()
        res["render_batch_ms"] = round(e0.elapsed_time(e1) / 10.0, 4)
        # bytes a render must move: z-buffer clear + read (2 x 8 B per pixel) + the 3 image planes and the mask plane written
        rbytes = B * 480 * 640 * (16 + 16)
        res["render_roofline"] = {"bound": "hbm", "achieved": round(rbytes / (res["render_batch_ms"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": round(rbytes / (res["render_batch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "bytes_per_render_batch": rbytes}
        if kind == "full_graph":   # decoder + both heads of one forward, as one event pair
            net = pred.net
            net.forward_test(refiner.batch)
            e0.record()
            for _ in range(10):
                net.decoder()
                net.heads()
            e1.record()
            e1.synchronize()
            res["decoder_heads_ms_per_forward"] = round(e0.elapsed_time(e1) / 10.0, 4)
        res["roofline"] = encoder_roofline(pred.net, refiner.batch, test_iter, profile_steps)
        for k in ("all_kernels", "traffic_unit"):
            res["roofline"].pop(k, None)
        del refiner, pred
        torch.cuda.empty_cache()
        return res
    finally:
        cfg.TEST.FAST_TEST, cfg.dataset.class_name, cfg.dataset.dataset = keep


def fresh_batch_bench(cfg, rm, refiner, B, dev, steps, warmup):
    """Non-headline `fresh_batch` object: the same 4-iteration refinement, but every step takes a NEW batch from host memory through
    the data layer (deepim/core/loader.py): raw pixels as the image files hold them (8-bit BGR, 16-bit depth: 2.15 MB per pair) ->
    pinned staging -> copy stream -> dim_test_blobs_from_raw + dim_box_mask build the float blobs in HBM -> graph replay.  Two staging
    sets alternate, so batch k+1 is staged and uploaded while batch k is refined.  File decoding is not in the loop (the pixels of four
    distinct synthetic batches sit in RAM): this is the PCIe-inclusive rate of the refinement path."""
    from deepim.core.loader import ArraySource, TestDataLoader, raw_from_device_batch
    from lib.utils import synthetic as syn

    raws = []
    for k in range(4):
        b = syn.build_device_batch(rm, B, seed=7000 + k, n_classes=len(cfg.dataset.class_name), pixel_means=cfg.network.PIXEL_MEANS, device=dev)
        depth = torch.empty((B, 1, rm.height, rm.width), device=dev)
        rm.render_batch(b["class_index"], b["src_pose"], depth=depth)
        raws.append(raw_from_device_batch(b, cfg.network.PIXEL_MEANS, depth, float(cfg.dataset.DEPTH_FACTOR)))
    cat = [np.concatenate([r[i] for r in raws]) for i in range(6)]
    need = steps + warmup + 2
    loader = TestDataLoader(None, cfg, batch_size=B, device=dev, workers=min(8, host_cores()),
                            source=ArraySource(*cat, repeat=-(-need // 4)))

    def step():
        st = loader.next_raw()
        refiner.load_staged(loader, st)
        refiner.refine()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    loader.close()
    per_pair = 2 * rm.height * rm.width * 3 + rm.height * rm.width * 2
    return {"value": round(B * steps / el, 2), "unit": "pose-refinements/sec", "ms_per_step": round(el / steps * 1e3, 3), "steps": steps,
            "h2d_bytes_per_pair": per_pair, "h2d_bytes_per_pair_as_float_blobs": 8 * rm.height * rm.width * 4,
            "status_flags": int(refiner.status_iter.abs().sum().item()),
            "pipeline": "host RAM (uint8 BGR x2 + uint16 depth) -> pinned staging (2 sets) -> copy stream -> dim_test_blobs_from_raw + "
                        "dim_box_mask -> hipGraph replay; decode of image files not included"}


def train_fresh_batch_bench(cfg, models, rm, B, dev, epochs=5, n_batches=6):
    """Non-headline `train_fresh_batch` object: the training step of the `train` object, but every data batch comes FROM
    IMAGE FILES through deepim/core/loader.TrainDataLoader -- PNG decode on a thread pool -> pinned staging -> copy stream -> every blob
    and label of get_data_pair_train_batch built in HBM (csrc/data.hip) -> fit_batch (TRAIN_ITER_SIZE forward / backward / update +
    re-render in between).  Epoch 1 decodes the files; from epoch 2 on every file is served by the decoded-pixel cache in HBM
    (`cached_*`: epochs 3 .. 5, the steady state).
    `resident` = the same fit_batch over private copies of the same batches that never touch the loader.  A synthetic LINEMOD-shaped dataset is written to a scratch
    directory first (untimed)."""
    import shutil
    import tempfile

    from deepim.core.loader import PixelCache, TrainDataLoader
    from deepim.core.module import MutableModule, fit_batch
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.dataset.synthetic_files import write_synthetic_dataset
    from lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti
    from lib.utils import image as I

    root = tempfile.mkdtemp(prefix="dim_bench_dataset_")
    keep = (cfg.dataset.model_dir, cfg.TRAIN.INIT_MASK, cfg.TRAIN.MASK_DILATE, cfg.TRAIN.FLOW_WEIGHT_TYPE)
    try:
        t0 = time.perf_counter()
        pairdb = write_synthetic_dataset(root, rm, models, list(cfg.dataset.class_name), B * n_batches, seed=4242)
        t_write = time.perf_counter() - t0
        cfg.dataset.model_dir = os.path.join(root, "models")
        I.point_cloud_dict.clear()
        sym = deepIM_flownet()
        sym.get_symbol(cfg, is_train=True)
        params = sym.init_weights(cfg, {}, {}, seed=0)
        upd = batchUpdaterPyMulti(cfg, 480, 640, render_machine=rm)
        n_iter = int(cfg.network.TRAIN_ITER_SIZE) if cfg.network.TRAIN_ITER else 1
        res = {"pairs_per_gpu": B, "batches_per_epoch": n_batches, "train_iter_size": n_iter, "files_per_pair": 5,
               "dataset_write_s": round(t_write, 2), "init_mask": cfg.TRAIN.INIT_MASK, "mask_dilate": bool(cfg.TRAIN.MASK_DILATE),
               "pipeline": "PNG files -> PIL decode (thread pool) -> pinned staging (2 sets) -> copy stream -> dim_pair_blobs_from_raw + "
                           "dim_box_mask / dim_mask_dilate + dim_se3_delta + dim_calc_flow_labels + dim_point_clouds -> fit_batch; epoch >= 2: "
                           "decoded-pixel cache in HBM (device-to-device copies, no decode, no PCIe)"}
        for dtype in ("f32", "bf16"):
            mod = MutableModule(cfg, params, B, device=dev, compute_dtype=dtype)
            cache = PixelCache(dev, budget_bytes=8 << 30)
            loader = TrainDataLoader(None, pairdb, cfg, batch_size=B, shuffle=False, device=dev, workers=min(16, host_cores()), cache=cache)
            per_epoch = []
            for ep in range(epochs):
                loader.reset()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for batch in loader:
                    fit_batch(mod, batch, upd, 1e-5)
                torch.cuda.synchronize()
                per_epoch.append((time.perf_counter() - t0) / n_batches * 1e3)
            # the data layer alone, cached: staging (cache look-ups, device-to-device copies) + the blob kernels, no training
            loader.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for batch in loader:
                pass
            torch.cuda.synchronize()
            loader_ms = (time.perf_counter() - t0) / n_batches * 1e3
            # the same six batches as private copies that never touch the loader (one pass: fit_batch rewrites a batch in place -- poses,
            # rendered image, labels -- so only pristine copies are the same work as a loader epoch)
            loader.reset()
            resident = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()} for batch in loader]
            loader.close()
            fit_batch(mod, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in resident[0].items()}, upd, 1e-5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for batch in resident:
                fit_batch(mod, batch, upd, 1e-5)
            torch.cuda.synchronize()
            res_ms = (time.perf_counter() - t0) / len(resident) * 1e3
            # epoch 1 decodes the files; epoch 2 is the first one served by the cache (first use of the hit path: module loads,
            # allocator growth); the steady state is epochs 3 ..
            warm = float(np.mean(per_epoch[2:])) if epochs > 2 else per_epoch[-1]
            res[dtype] = {"epoch1_ms_per_batch": round(per_epoch[0], 2), "epoch2_first_cached_ms_per_batch": round(per_epoch[min(1, epochs - 1)], 2),
                          "cached_ms_per_batch": round(warm, 2),
                          "resident_ms_per_batch": round(res_ms, 2), "data_layer_alone_cached_ms_per_batch": round(loader_ms, 2),
                          "cached_pairs_per_s": round(B / warm * 1e3, 1),
                          "resident_pairs_per_s": round(B / res_ms * 1e3, 1), "cached_over_resident": round(res_ms / warm, 3),
                          "cache_hits": cache.hits, "cache_misses": cache.misses, "cache_mb": round(cache.used / 2 ** 20, 1),
                          "finite": bool(torch.isfinite(mod.flat_w).all().item())}
            del mod, loader, cache
            torch.cuda.empty_cache()
        return res
    finally:
        cfg.dataset.model_dir, cfg.TRAIN.INIT_MASK, cfg.TRAIN.MASK_DILATE, cfg.TRAIN.FLOW_WEIGHT_TYPE = keep
        shutil.rmtree(root, ignore_errors=True)


def train_bench(cfg, models, rm, B, dev, rank, world, dist, steps, progress=None):
    """Non-headline `train` object (BASELINE configs[2] per-GPU shape): one training iteration = train-graph forward (encoder +
    decoder + flow / mask / pose heads) + all losses + full backward + gradient all-reduce(SUM) over the ranks + SGD-momentum
    update + weight repack, 16 pairs per GPU, on a synthetic batch resident in HBM.  Phases are timed with HIP events on the
    launch stream; the whole iteration with a barrier + device sync on both sides, MAX over ranks."""
    from deepim.core.module import MutableModule
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti
    from lib.utils import synthetic as syn
    from lib.utils.dist_utils import gather_floats

    fast = cfg.TEST.FAST_TEST
    cfg.TRAIN.lr = 1e-4
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=True)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    batch = syn.build_device_train_batch(rm, B, seed=5 + rank, models=models, n_classes=len(models), pixel_means=cfg.network.PIXEL_MEANS,
                                         npts=int(cfg.train_iter.NUM_3D_SAMPLE), device=dev)
    upd = batchUpdaterPyMulti(cfg, 480, 640, render_machine=rm)

    def phase_ms(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / steps

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    res = {"pairs_per_gpu": B, "global_pairs": B * world, "steps": steps,
           "workload": "train graph (encoder + decoder + flow / mask / point-matching losses), backward, all-reduce(SUM) of the 57.75 M "
                       "gradients, SGD momentum + repack; synthetic batch resident in HBM"}
    # f32 = the reference's precision; bf16 = BASELINE configs[2]: convolutions on the bf16 matrix pipe (f32 accumulate, f32 master
    # weights / momentum / losses / SE(3)), gradient bucket all-reduced as bf16 -- declared tolerance: tests/test_gpu_train_bf16.py
    progress = {} if progress is None else progress
    for dtype in ("f32", "bf16"):
        mod = MutableModule(cfg, params, B, device=dev, compute_dtype=dtype)
        progress.update(module=mod, dtype=dtype, phase="phases")

        def backward_and_reduce():
            # the phase figure includes the gradient sum: every bucket this backward hands to the collective is waited for before the
            # next pass rewrites flat_g (the overlapped figure is the whole-iteration timing below)
            mod.backward(batch)
            mod._finish_allreduce()

        r = {"forward_ms": phase_ms(lambda: mod.forward(batch)), "backward_ms": phase_ms(backward_and_reduce if world > 1 else (lambda: mod.backward(batch))),
             "allreduce_update_repack_ms": phase_ms(lambda: mod.update(0.0))}
        preds = mod.forward(batch)
        r["batch_updater_ms"] = phase_ms(lambda: upd.forward(batch, preds))
        progress["phase"] = "iterations"
        mod.forward_backward(batch)
        mod.update(1e-4)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            mod.forward_backward(batch)
            mod.update(1e-4)
        sync()
        el = el_own = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        r["iteration_ms"] = el / steps * 1e3
        r["iteration_ms_per_rank"] = [round(v / steps * 1e3, 3) for v in gather_floats(el_own)]
        r["pair_iterations_per_s"] = B * world * steps / el
        if dist is not None:
            # the gradient buckets on their own: bytes that cross the links per rank and the time of a blocking all-reduce(SUM) of
            # each (no compute beside it), slowest rank; bus GB/s by the ring formula 2 (N-1)/N x bytes / time
            bk = []
            for a, b_ in mod.buckets:
                buf = mod.flat_g16[a:b_] if (mod.bf16 and mod.flat_g16 is not None) else mod.flat_g[a:b_]
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)   # warm
                sync()
                t1 = time.perf_counter()
                for _ in range(3):
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
                torch.cuda.synchronize()
                tb = torch.tensor([(time.perf_counter() - t1) / 3.0], dtype=torch.float64, device=dev)
                dist.all_reduce(tb, op=dist.ReduceOp.MAX)
                nbytes = buf.numel() * buf.element_size()
                bk.append({"begin": int(a), "end": int(b_), "bytes": int(nbytes), "dtype": str(buf.dtype).replace("torch.", ""),
                           "ms": round(float(tb.item()) * 1e3, 3),
                           "bus_GB/s": round(2.0 * (world - 1) / world * nbytes / float(tb.item()) / 1e9, 1)})
            r["allreduce_buckets"] = bk
            r["allreduce_bytes_per_update"] = int(sum(x["bytes"] for x in bk))
        r["finite"] = bool(torch.isfinite(mod.flat_w).all().item())
        if dtype == "bf16":
            # per-layer roofline of the bf16 forward convolutions (HIP events on the launch stream, 3 passes): against the dense bf16
            # MFMA peak and against HBM with the layer's compulsory bytes (fp32 activations in + out, bf16 weights); the bound is the
            # larger of the two lower limits -- the 8-channel first layer is HBM-bound in bf16 (130 FLOP/B against a ridge of 312)
            evs = {}
            for _ in range(3):
                mod.net.encoder(events=evs)
            torch.cuda.synchronize()
            layers = {}
            for name, lst in evs.items():
                info = mod.net.layer_info.get(name)
                if name == "fc6" or not info:
                    continue
                ms = sum(e[1].elapsed_time(e[2]) for e in lst) / 3.0
                nbytes = info["min_bytes"] - 2 * info["N"] * info["K"]   # weights are read as bf16
                t_mfma, t_hbm = info["flops"] / (BF16_MFMA_PEAK_TFLOPS * 1e12), nbytes / (HBM_PEAK_GBS * 1e9)
                layers[name] = {"ms": round(ms, 4), "TFLOP/s": round(info["flops"] / ms / 1e9, 1), "compulsory_GB/s": round(nbytes / ms / 1e6, 1),
                                "flop_per_byte": round(info["flops"] / nbytes, 1), "bound": "hbm" if t_hbm > t_mfma else "mfma",
                                "frac_of_bound": round(max(t_mfma, t_hbm) * 1e3 / ms, 3)}
            r["forward_conv_layers"] = layers
        res[dtype] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}
        progress.pop("module", None)
        del mod
    cfg.TEST.FAST_TEST = fast
    return res


def main():
    args = parse_args()
    # `python bench.py --gpus N` is ONE command (the reference takes `--gpus 0,1,2,3` the same way, deepim/train.py:425-438): when this
    # process is not already a rank of a torch.distributed.run job it starts the N ranks as a child job -- before anything here touches
    # the GPU -- relays their output (rank 0 prints the JSON line) and leaves with the job's exit code.  Under an external launcher
    # WORLD_SIZE must equal --gpus (also when it is 1): launch_ranks_if_needed raises otherwise.
    from lib.utils.dist_utils import launch_ranks_if_needed

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    rc = launch_ranks_if_needed(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if rc is not None:
        sys.exit(rc)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus {} but WORLD_SIZE {}".format(args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend != "nccl":   # host-side backend: the ranks meet before anything touches a GPU
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
            dist.barrier()
            sys.stderr.write("bench.py: rank {} of {} joined ({})\n".format(rank, world, args.dist_backend))
            sys.stderr.flush()
    # (torch.cuda.device_count() does not initialise the GPU on this image)
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs a GPU: the refinement path has no CPU fallback")
    if args.dist_backend == "nccl" and world > n_dev:
        raise SystemExit("--gpus {} over RCCL needs {} GPUs, {} visible (rehearse several ranks on one card with --dist-backend gloo)".format(
            args.gpus, world, n_dev))
    # gloo rehearsal: more ranks than cards share the cards round-robin (DIM_BENCH_DEVICE pins all of them to one card)
    dev_index = int(os.environ.get("DIM_BENCH_DEVICE", local_rank % n_dev))
    torch.cuda.set_device(dev_index)
    dev = "cuda:{}".format(dev_index)
    if world > 1 and args.dist_backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        dist.barrier()
        sys.stderr.write("bench.py: rank {} of {} joined (nccl = RCCL, {})\n".format(rank, world, dev))
        sys.stderr.flush()

    # which devices joined: every rank reports the card it drives; two ranks on one card under RCCL is an error, not a slow run
    from lib.utils.dist_utils import gather_floats, gather_rank_identities, rank_identity

    ranks = gather_rank_identities(rank_identity(dev_index, rank, local_rank), backend=args.dist_backend)

    from deepim.config.config import config as cfg, update_config
    from deepim.core.tester import Predictor, Refiner
    from deepim.symbols.deepIM_flownet import deepIM_flownet
    from lib.hip import ops
    from lib.render_hip.render_py_multi import Render_Py
    from lib.utils import synthetic as syn

    update_config(args.cfg)
    if args.full_graph:
        cfg.TEST.FAST_TEST = False
    B = args.batch_pairs or int(cfg.TEST.BATCH_PAIRS)
    test_iter = int(cfg.TEST.test_iter)
    sym = deepIM_flownet()
    sym.get_symbol(cfg, is_train=False)
    params = sym.init_weights(cfg, {}, {}, seed=0)
    models = syn.make_models(seed=2333, n_models=len(cfg.dataset.class_name), subdiv=args.subdiv)
    rm = Render_Py(None, cfg.dataset.class_name, cfg.dataset.INTRINSIC_MATRIX, zNear=cfg.dataset.ZNEAR, zFar=cfg.dataset.ZFAR,
                   device=dev, meshes=models)
    if args.head_epochs > 0:
        # weights under which the loop contracts toward the observed pose, like a trained DeepIM: trained here, untimed, on this
        # rank's own benchmark pairs (tests/test_gpu_learns.py is the same recipe with its assertions)
        params, weights_info = train_head(cfg, params, models, rm, B, dev, 1000 + rank, args.head_epochs)   # same seed = the same pairs
    else:
        # a pose head that moves the pose 3-12 deg / 4-42 mm per iteration (tests/loop_parity.py): every re-render covers new
        # pixels.  The reference initialisation (trans = 0, rot rows ~ U(0, 0.01)) would re-render nearly the same image.
        _rng = np.random.RandomState(1)
        params["trans_weight"] = (_rng.randn(3, 256) * 0.02).astype(np.float32)
        params["rot_weight"][1:] = (_rng.randn(3, 256) * 0.2).astype(np.float32)
        weights_info = {"kind": "seeded initialisation + scaled random pose head (3-12 deg per iteration)"}
    ops.set_winograd_split(not args.f32_pipe)   # read when the layers are planned (here) and when the graph is captured
    pred = Predictor(cfg, params, B, device=dev, winograd=not args.no_winograd)
    batch = syn.build_device_batch(rm, B, seed=1000 + rank, n_classes=len(models), pixel_means=cfg.network.PIXEL_MEANS, device=dev)
    if args.autotune:  # untimed: choose tile / split-K per layer on this GPU before the graph is captured
        pred.net.zoom({k: batch[k] for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose")})
        pred.net.autotune()
    refiner = Refiner(cfg, pred, rm, B, capture_graph=not args.no_graph)
    refiner.load(batch["image_observed"], batch["image_rendered"], batch["mask_observed"], batch["mask_rendered"], batch["src_pose"],
                 batch["class_index"])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        refiner.refine()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        refiner.refine()
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank_ms = [round(v / args.steps * 1e3, 3) for v in gather_floats(elapsed)]   # every rank's own clock, next to the MAX
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    status = int(refiner.status_iter.abs().sum().item())
    poses_hip, se3_hip = refiner.poses_iter.cpu().numpy().copy(), refiner.se3_iter.cpu().numpy().copy()
    pose_err = pose_error_vs_gt(batch["src_pose"], refiner.poses_iter, batch["pose_gt"])

    roofline = encoder_roofline(pred.net, refiner.batch, test_iter, args.profile_steps)
    net = pred.net

    total_pairs = B * world * args.steps
    value = total_pairs / elapsed
    out = {
        "metric": "pose-refinements/sec (4 iters, 480x640)", "value": round(value, 2), "unit": "pose-refinements/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("LINEMOD 'ape' batch={} per GPU, {} iters, fp32, "
                                + ("FULL test graph (zoom + FlowNetS encoder + FC heads + decoder + flow / mask heads) " if args.full_graph
                                   else "FAST_TEST graph (zoom + FlowNetS encoder + FC heads) ")
                                + "+ SE3 compose + HIP rasteriser ({} triangles) + box_rendered mask update").format(
                                   B, test_iter, models[0][2].shape[0]),
                   "pairs_per_gpu": B, "global_pairs": B * world, "test_iter": test_iter, "hipgraph": not args.no_graph, "winograd": "off" if args.no_winograd else "F(4x4,3x3): {}; phase images + F(4x4,3x3): {}{}".format(
                       " ".join(net.wino), " ".join(net.wino5),
                       "; phase images + minimal filtering F(4,1) x F(4,2): " + " ".join(net.wino3s2) if net.wino3s2 else ""),
                   "plane_gemm_arithmetic": "f32 pipe (v_mfma_f32_32x32x2_f32)" if args.f32_pipe or args.no_winograd else
                   "f32 operands as three bf16 terms, six products per multiply on v_mfma_f32_32x32x16_bf16, f32 accumulate (f32 in HBM, "
                   "f32-level error: tests/test_split_terms.py, parity object below)",
                   "gflop_per_refinement": round(net.flops_per_forward() * test_iter / B / 1e9, 2), "status_flags": status,
                   "conv_plan": {k: list(v) for k, v in net.conv_plan.items()}},
        "roofline": roofline,
        "weights": weights_info, "pose_error_vs_gt": pose_err,
        "ranks": ranks, "ms_per_step_per_rank": per_rank_ms, "dist_backend": args.dist_backend if world > 1 else None,
        "distinct_devices": len({(r["host"], r["uuid"] or r["pci_bus_id"] or r["device"]) for r in ranks}),
    }
    if world == 1 and not args.no_variants and not args.no_graph and not args.f32_pipe and not args.no_winograd:
        # the same loop, same batch, same weights with every plane GEMM on the f32 matrix pipe: the A/B of the default arithmetic in
        # the same process (timed like the headline), and how far the two arithmetics' poses lie apart
        try:
            ops.set_winograd_split(False)
            pred_f = Predictor(cfg, params, B, device=dev, winograd=True)
            ref_f = Refiner(cfg, pred_f, rm, B, capture_graph=True)
            ref_f.load(batch["image_observed"], batch["image_rendered"], batch["mask_observed"], batch["mask_rendered"], batch["src_pose"],
                       batch["class_index"])
            for _ in range(args.warmup):
                ref_f.refine()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                ref_f.refine()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            se3_f = ref_f.se3_iter.cpu().numpy()
            out["f32_pipe"] = {"value": round(B * args.steps / dt, 2), "unit": "pose-refinements/sec", "ms_per_step": round(dt / args.steps * 1e3, 3),
                               "headline_over_this": round(value / (B * args.steps / dt), 4),
                               "first_iteration_se3_max_abs_diff_to_headline": float(np.abs(se3_f[0] - se3_hip[0]).max())}
            del ref_f, pred_f
        except Exception as e:
            out["f32_pipe"] = {"error": "{}: {}".format(type(e).__name__, e)}
        finally:
            ops.set_winograd_split(True)
    if world == 1 and not args.no_fresh_batch and not args.no_graph:
        try:
            out["fresh_batch"] = fresh_batch_bench(cfg, rm, refiner, B, dev, args.steps, args.warmup)
        except Exception as e:
            out["fresh_batch"] = {"error": "{}: {}".format(type(e).__name__, e)}
    if world == 1 and not args.no_variants and not args.no_graph:
        for kind in ("full_graph", "modelnet_lit"):
            try:   # non-headline objects: a failure must not cost the headline line
                out[kind] = variant_bench(kind, cfg, dev, rank, max(5, args.steps // 2), args.warmup, 1, args.subdiv)
            except Exception as e:
                out[kind] = {"error": "{}: {}".format(type(e).__name__, e)}
    if not args.no_train:
        del refiner, pred
        torch.cuda.empty_cache()

        progress = {}

        def run_train():
            try:
                torch.cuda.set_device(dev_index)
                out["train"] = train_bench(cfg, models, rm, int(cfg.TRAIN.BATCH_PAIRS) if args.batch_pairs is None else B, dev, rank, world,
                                           dist, args.train_steps, progress=progress)
            except Exception as e:  # the non-headline object must never take the headline line down with it
                out["train"] = {"error": "{}: {}".format(type(e).__name__, e)}

        if world == 1:
            run_train()
        else:
            # With several ranks the training object is the only part of this program that posts data-path collectives (the gradient
            # all-reduce).  An exception is caught above; a collective that never completes is not an exception, so the object runs
            # under a deadline: past it rank 0 still prints the headline line and the process leaves without waiting for the thread.
            import threading

            th = threading.Thread(target=run_train, daemon=True)
            th.start()
            th.join(args.train_deadline)
            if th.is_alive():
                # a hang, not a result: rank 0 still prints the headline line (measured before the training object started), every
                # rank says what it was waiting for, and EVERY rank leaves with a non-zero code so that the launcher and the driver
                # record the failure.  No retry, no restart in-process: kernels / RCCL work may still be in flight.
                mod = progress.get("module")
                pend = None
                try:
                    pend = [{"begin": a, "end": b, "completed": c} for a, b, c in mod.pending_buckets()] if mod is not None else None
                except Exception as e:  # querying a wedged communicator may itself fail
                    pend = "unavailable: {}".format(e)
                out["train"] = {"error": "no result within {} s (a collective of the training object did not complete)".format(args.train_deadline),
                                "rank": rank, "phase": progress.get("phase"), "dtype": progress.get("dtype"), "pending_buckets": pend}
                sys.stderr.write("bench.py rank {}: train object hung: {}\n".format(rank, json.dumps(out["train"])))
                sys.stderr.flush()
                if rank == 0:
                    print(json.dumps(out), flush=True)
                os._exit(3)
    if world == 1 and not args.no_train and not args.no_train_files:
        try:
            out["train_fresh_batch"] = train_fresh_batch_bench(cfg, models, rm, int(cfg.TRAIN.BATCH_PAIRS) if args.batch_pairs is None else B, dev)
        except Exception as e:
            out["train_fresh_batch"] = {"error": "{}: {}".format(type(e).__name__, e)}
    parity_failed = False
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, params, models, batch, args.cpu_pairs)
    if rank == 0 and world == 1 and args.parity_pairs > 0:
        # the metric's second clause ("ADD(-S) vs reference"), checker leg: never inside the timed region
        trained = args.head_epochs > 0
        out["parity"] = parity_leg(cfg, params, models, batch, poses_hip, se3_hip, min(args.parity_pairs, B), trained)
        finals = getattr(cpu_baseline, "final_poses", None)
        if finals is not None:
            # the cpu_baseline leg refined the batch's pairs free-running (cyclically, if its sample is larger than the batch): ADD of
            # the HIP loop's final poses against the oracle's
            from oracle import pose_error

            adds = []
            for b in range(min(args.cpu_pairs, B)):
                pts = models[int(batch["class_index"][b])][0].astype(np.float64)
                ph, po = poses_hip[-1, b].astype(np.float64), finals[b]
                adds.append(pose_error.add(ph[:, :3], ph[:, 3], po[:, :3], po[:, 3], pts) / float(np.linalg.norm(pts.max(0) - pts.min(0))))
            out["parity"]["add_free_over_d_cpu_baseline_pairs"] = {"pairs": int(min(args.cpu_pairs, B)), "max": float(max(adds)), "median": float(np.median(adds)),
                                                                  "within_0.02": int(sum(a < 0.02 for a in adds))}
            if trained:
                out["parity"]["ok"] = bool(out["parity"]["ok"] and max(adds) < 0.02)
        parity_failed = not out["parity"]["ok"]
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if parity_failed:
        sys.stderr.write("bench.py: parity object over its bar: {}\n".format(json.dumps(out["parity"])))
        sys.exit(4)


if __name__ == "__main__":
    main()
