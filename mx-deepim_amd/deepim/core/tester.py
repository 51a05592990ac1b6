"""Test-time iterative refinement, device resident.

Mirror of /root/reference/deepim/core/tester.py: `Predictor` (:27-56) and the refinement loop of
`pred_eval` (:418-642).  The reference runs batch 1 per GPU and, per iteration, syncs the 7 floats to
the host, composes the pose in numpy (:525-532), renders with OpenGL + glReadPixels (:563-568),
rebuilds the blobs on the host (data_pair.update_data_batch) and uploads them (:590).  Here a batch of
pairs stays in HBM for all iterations:

    forward (zoom -> encoder -> heads)  ->  se3_compose  ->  rasterise (image_rendered, mask_rendered,
    bbox)  ->  box mask (mask_observed, UPDATE_MASK == "box_rendered")  ->  forward ...

and the whole loop can be captured into one hipGraph (`Refiner(capture_graph=True)`).
"""
from __future__ import print_function, division

import numpy as np
import torch

from lib.hip import ops
from deepim.symbols.deepIM_flownet import FlowNetHip


class Predictor(object):
    """Reference: binds a MutableModule and calls forward (tester.py:27-56).  Here: owns a FlowNetHip."""

    def __init__(self, config, arg_params, batch_size, device="cuda:0", conv_plan=None, winograd=True):
        self.net = FlowNetHip(config, arg_params, batch_size, device=device, conv_plan=conv_plan, winograd=winograd)
        self.data_names = ["image_observed", "image_rendered", "src_pose", "class_index", "mask_observed", "mask_rendered"]

    def predict(self, data_batch):
        """data_batch: dict blob-name -> CUDA tensor (names/shapes as deepim/core/loader.py:35-41).
        Returns [dict(se3_output, zoom_factor)] -- one entry, since one process drives one GPU."""
        return [self.net.forward_test(data_batch)]


class Refiner(object):
    def __init__(self, config, predictor, render_machine, batch_size, capture_graph=False):
        cfg = config
        if cfg.network.INPUT_MASK and cfg.network.PRED_MASK and cfg.TEST.UPDATE_MASK not in ("box_rendered", "init"):
            # same restriction as the released loop (tester.py:579-587)
            raise Exception("Unknown UPDATE_MASK type: {}".format(cfg.TEST.UPDATE_MASK))
        self.cfg = cfg
        self.predictor = predictor
        self.net = predictor.net
        self.render_machine = render_machine
        self.B = batch_size
        self.test_iter = int(cfg.TEST.test_iter)
        d = self.net.device
        B, H, W = batch_size, 480, 640
        self.batch = {
            "image_observed": torch.zeros((B, 3, H, W), dtype=torch.float32, device=d),
            "image_rendered": torch.zeros((B, 3, H, W), dtype=torch.float32, device=d),
            "mask_observed": torch.zeros((B, 1, H, W), dtype=torch.float32, device=d),
            "mask_rendered": torch.zeros((B, 1, H, W), dtype=torch.float32, device=d),
            "src_pose": torch.zeros((B, 3, 4), dtype=torch.float32, device=d),
            "class_index": torch.zeros((B,), dtype=torch.int32, device=d),
        }
        self.input_depth = bool(cfg.network.INPUT_DEPTH)
        if self.input_depth:   # the two depth planes of get_convs (:33-50); the rendered one is refreshed by every re-render (tester.py:573-574)
            self.batch["depth_observed"] = torch.zeros((B, 1, H, W), dtype=torch.float32, device=d)
            self.batch["depth_rendered"] = torch.zeros((B, 1, H, W), dtype=torch.float32, device=d)
        # pristine copies of the blobs the loop overwrites, so refine() can be replayed on the same batch
        self.init = {k: torch.zeros_like(self.batch[k]) for k in ("image_rendered", "mask_observed", "mask_rendered")
                     + (("depth_rendered",) if self.input_depth else ())}
        self.pose_init = torch.zeros((B, 3, 4), dtype=torch.float32, device=d)
        self.poses_iter = torch.zeros((self.test_iter, B, 3, 4), dtype=torch.float32, device=d)
        self.se3_iter = torch.zeros((self.test_iter, B, 7), dtype=torch.float32, device=d)
        self.bbox = torch.zeros((B, 4), dtype=torch.int32, device=d)
        self.bbox2 = torch.zeros((B, 4), dtype=torch.int32, device=d)   # consecutive renders alternate: one box is the next render's dirty-box hint
        self.bbox_obs = torch.zeros((B, 4), dtype=torch.int32, device=d)
        self.status_iter = torch.zeros((self.test_iter, B), dtype=torch.int32, device=d)
        # per-iteration head outputs of the full (not FAST_TEST) graph, read by the reference at tester.py:485-491
        self.with_heads = bool(self.net.has_decoder and not cfg.TEST.FAST_TEST)
        self.mask_pred_iter = self.flow_est_iter = None
        if self.with_heads and cfg.network.PRED_MASK:
            self.mask_pred_iter = torch.zeros((self.test_iter, B, 1, H, W), dtype=torch.float32, device=d)
        if self.with_heads and cfg.network.PRED_FLOW:
            self.flow_est_iter = torch.zeros((self.test_iter, B, 2, H, W), dtype=torch.float32, device=d)
        self.T_means = np.asarray(cfg.dataset.trans_means, dtype=np.float32)
        self.T_stds = np.asarray(cfg.dataset.trans_stds, dtype=np.float32)
        render_machine.reserve(B)
        # ModelNet: the lit renderer takes a per-render light intensity drawn on the host (tester.py:227-230)
        self.lit = hasattr(render_machine, "normals")
        self.light_int = torch.ones((max(self.test_iter - 1, 1), B, 3), dtype=torch.float32, device=d) if self.lit else None
        self.graph = None
        self._want_graph = capture_graph

    # ------------------------------------------------------------------------------------------
    def load(self, image_observed, image_rendered, mask_observed, mask_rendered, src_pose, class_index, depth_observed=None,
             depth_rendered=None):
        """copy one batch of blobs (any device) into the resident buffers (the depth planes: INPUT_DEPTH graphs only)"""
        b = self.batch
        if self.input_depth:
            assert depth_observed is not None and depth_rendered is not None, "INPUT_DEPTH: the loop needs depth_observed / depth_rendered"
            b["depth_observed"].copy_(torch.as_tensor(depth_observed))
            self.init["depth_rendered"].copy_(torch.as_tensor(depth_rendered))
        b["image_observed"].copy_(torch.as_tensor(image_observed))
        self.init["image_rendered"].copy_(torch.as_tensor(image_rendered))
        self.init["mask_observed"].copy_(torch.as_tensor(mask_observed))
        self.init["mask_rendered"].copy_(torch.as_tensor(mask_rendered))
        self.pose_init.copy_(torch.as_tensor(src_pose))
        b["class_index"].copy_(torch.as_tensor(class_index).to(torch.int32))
        if self.lit and self.test_iter > 1:
            # same draws, same order as the reference: sample by sample, one np.random.uniform(0.9,1.1,3) per re-render
            li = np.stack([[np.random.uniform(0.9, 1.1, size=(3,)) for _ in range(self.test_iter - 1)] for _ in range(self.B)])
            self.light_int.copy_(torch.from_numpy(li.transpose(1, 0, 2).astype(np.float32)))

    def load_staged(self, loader, staged):
        """take the next batch straight from a deepim.core.loader.TestDataLoader staging set: the raw pixels it uploaded are turned
        into the resident blobs by dim_test_blobs_from_raw / dim_box_mask on the current stream -- no host blobs, no extra copies"""
        loader.build_blobs(staged, out={"image_observed": self.batch["image_observed"], "image_rendered": self.init["image_rendered"],
                                        "mask_rendered": self.init["mask_rendered"], "mask_observed": self.init["mask_observed"]})
        ops.copy(self.pose_init, staged.d_pose)
        ops.copy(self.batch["class_index"], staged.d_cls)
        loader.release(staged)   # the last read of the staging set's device mirrors is enqueued
        if self.lit and self.test_iter > 1:
            li = np.stack([[np.random.uniform(0.9, 1.1, size=(3,)) for _ in range(self.test_iter - 1)] for _ in range(self.B)])
            self.light_int.copy_(torch.from_numpy(li.transpose(1, 0, 2).astype(np.float32)))

    def _loop(self):
        """tester.py:476-598 for a whole batch; everything enqueued on the current stream, no host sync."""
        cfg, net, b = self.cfg, self.net, self.batch
        # The loaded blobs are read WHERE THEY LIE by the first forward; a working plane takes over once the loop has written it (the
        # first render writes image_rendered / mask_rendered [/ depth_rendered] in full, box_mask all of mask_observed).  Round 3
        # copied every pristine blob into its working plane at the start of each replay: 100 MB and ~40 us per step at 16 pairs.
        # A plane the loop never writes (one iteration only; UPDATE_MASK 'init': mask_observed) is still copied, so that `batch` ends
        # up as the reference leaves its data batch.  (ops.copy is a kernel: no memcpy / memset node may sit in the captured graph.)
        cur = dict(b)
        cur.update(self.init)
        box_update = cfg.network.INPUT_MASK and cfg.network.PRED_MASK and cfg.TEST.UPDATE_MASK == "box_rendered"
        for k, v in self.init.items():
            if self.test_iter < 2 or (k == "mask_observed" and not box_update):
                ops.copy(b[k], v)
                cur[k] = b[k]
        bbox_ren = bbox_obs = None
        pose = self.pose_init
        for it in range(self.test_iter):
            # se3 and status land directly in their per-iteration rows; the pose of the previous iteration is read where it lies
            out = net.forward_test(cur, bbox_ren=bbox_ren, bbox_obs=bbox_obs, src_pose=pose, se3_out=self.se3_iter[it],
                                   status_out=self.status_iter[it])
            if self.mask_pred_iter is not None:
                ops.copy(self.mask_pred_iter[it], out["mask_observed_pred_output"])
            if self.flow_est_iter is not None:
                ops.copy(self.flow_est_iter[it], out["flow_est_crop_output"])
            # pose_rendered_update = RT_transform(pose_rendered, se3[:-3], se3[-3:], ...)   (:525-532)
            ops.se3_compose(pose, self.se3_iter[it], cfg.network.ROT_COORD, self.T_means, self.T_stds, out=self.poses_iter[it])
            if it < self.test_iter - 1:
                # render(render_machine, pose_rendered_update, cls_idx) + update_data_batch  (:563-590)
                extra = {"light_intensity": self.light_int[it]} if self.lit else {}
                # (the loop needs the depth only for mask_rendered = depth > 0.2, tester.py:575-577: the resolve pass writes the mask itself
                # and the depth plane is not materialised -- 1.2 MB per pair and render less to write)
                if self.input_depth:
                    extra["depth"] = b["depth_rendered"]   # INPUT_DEPTH: the rendered depth is a network input (tester.py:573-574)
                # from the second render on the planes hold the previous render: background outside ITS box, which is not written again
                bb_new, bb_prev = (self.bbox, self.bbox2) if it % 2 == 0 else (self.bbox2, self.bbox)
                self.render_machine.render_batch(b["class_index"], self.poses_iter[it], image=b["image_rendered"],
                                                 mask=b["mask_rendered"], bbox=bb_new, plane_means=net.plane_means, mask_thr=0.2,
                                                 status=self.status_iter[it], clean_bbox=bb_prev if it > 0 else None, **extra)
                cur["image_rendered"], cur["mask_rendered"] = b["image_rendered"], b["mask_rendered"]
                if self.input_depth:
                    cur["depth_rendered"] = b["depth_rendered"]
                if box_update:
                    # data_pair.py:103-114; the rectangle's own bbox comes back with it, so ZoomMask does not scan the mask again
                    ops.box_mask(bb_new, b["mask_observed"], bbox_of_mask=self.bbox_obs)
                    bbox_obs = self.bbox_obs
                    cur["mask_observed"] = b["mask_observed"]
                pose = self.poses_iter[it]
                bbox_ren = bb_new
        ops.copy(b["src_pose"], pose)  # the blob ends up as the reference leaves it: the pose the last forward used

    def refine(self):
        """run test_iter iterations on the loaded batch; returns poses_iter (test_iter,B,3,4) (device)."""
        if self._want_graph and self.graph is None:
            s = torch.cuda.Stream(device=self.net.device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._loop()  # warm-up outside capture (lazy module loads, attribute sets)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._loop()
            self.graph = g
        if self.graph is not None:
            self.graph.replay()
        else:
            self._loop()
        return self.poses_iter


class FlowEPE(object):
    """Test-time flow error, reference deepim/core/tester.py:500-512 (accumulation), :675-716 (par_generate_gt) and :719-736
    (calc_EPE_one_pair), active when `PRED_FLOW and not FAST_TEST`: the flow head's output of the FIRST forward of every pair
    (`rst_iter` is built from predictor.predict before the refinement loop, :476-494) against calc_flow(depth_rendered,
    pose_rendered, pose_observed, K, depth_gt_observed) of the initial pair.  Labels (dim_calc_flow_labels) and the three masked
    error sums (dim_flow_epe_sums) stay on the device; ONE (5,) float64 read-out at the end."""

    def __init__(self, config, batch_size, device):
        self.cfg = config
        K = np.asarray(config.dataset.INTRINSIC_MATRIX, dtype=np.float64).reshape(3, 3)
        self.K = K
        self.Kinv64 = np.linalg.inv(K)
        B, H, W = batch_size, 480, 640
        self.flow = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
        self.weights = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
        self.sums = torch.zeros((B, 5), dtype=torch.float64, device=device)
        self.work = torch.empty((ops.lib().dim_flow_epe_workspace_bytes(B) // 8,), dtype=torch.float64, device=device)
        self.P12 = torch.empty((B, 3, 4), dtype=torch.float64, device=device)
        self.num_all = 0

    def add(self, batch, flow_est, skip=None):
        """batch: depth_rendered (B,1,H,W) of the initial render, depth_gt_observed (B,1,H,W) (zero off the object: tester.py:700-704),
        src_pose, pose_observed; flow_est (B,2,H,W) = flow_est_crop_output of the first forward.  skip: (B,) bool, pairs that are not
        scored (undetected objects leave the loop before the flow error, :451-475)."""
        from lib.utils.projection import se3_inverse, se3_mul

        for k in ("depth_rendered", "depth_gt_observed", "pose_observed"):
            if k not in batch:
                raise KeyError("test-time flow error (PRED_FLOW and not FAST_TEST) needs the blob '{}' (par_generate_gt reads it from "
                               "the pair record, tester.py:681-704)".format(k))
        src = torch.as_tensor(batch["src_pose"]).cpu().numpy().astype(np.float64)
        tgt = torch.as_tensor(batch["pose_observed"]).cpu().numpy().astype(np.float64)
        P = np.stack([np.matmul(self.K, se3_mul(tgt[b], se3_inverse(src[b]))) for b in range(src.shape[0])])   # flow.py:31
        self.P12.copy_(torch.from_numpy(np.ascontiguousarray(P)))
        dr = torch.as_tensor(batch["depth_rendered"]).to(self.flow.device, torch.float32).contiguous()
        dg = torch.as_tensor(batch["depth_gt_observed"]).to(self.flow.device, torch.float32).contiguous()
        ops.calc_flow_labels(dr, dg, self.P12, self.Kinv64, self.flow, self.weights, standard_rep=bool(self.cfg.network.STANDARD_FLOW_REP),
                             weight_type="viz")
        per = ops.flow_epe_sums(flow_est, self.flow, self.weights[:, :1].contiguous(), dr, workspace=self.work)
        if skip is not None and bool(np.any(skip)):
            per = per * torch.from_numpy(~np.asarray(skip, dtype=bool)).to(per.device, torch.float64)[:, None]
        self.sums += per
        self.num_all += int(per.shape[0] - (0 if skip is None else int(np.sum(skip)))) * dr.shape[2] * dr.shape[3]

    def result(self, merge_ranks=True):
        """-> dict(epe_all, epe_vizbg, epe_viz: the three numbers the reference prints at :656-660; sums and counts next to them)"""
        import torch.distributed as dist

        t = torch.cat([self.sums.sum(0), torch.tensor([float(self.num_all)], dtype=torch.float64, device=self.sums.device)]).cpu()
        if merge_ranks and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            parts = [None] * dist.get_world_size()
            dist.all_gather_object(parts, t.numpy())
            t = torch.from_numpy(np.sum(parts, axis=0))
        s_all, s_viz, s_vizbg, n_viz, n_vizbg, n_all = (float(v) for v in t)
        return {"epe_all": s_all / max(n_all, 1.0), "epe_vizbg": s_vizbg / max(n_vizbg, 1.0), "epe_viz": s_viz / max(n_viz, 1.0),
                "sum_EPE_all": s_all, "sum_EPE_viz": s_viz, "sum_EPE_vizbg": s_vizbg, "num_inst_all": n_all, "num_inst_viz": n_viz,
                "num_inst_vizbg": n_vizbg}


def pred_eval(config, refiner, batches, evaluator, result_file=None, logger=None, merge_ranks=True):
    """The outer loop of the reference's pred_eval (deepim/core/tester.py:418-676) on device-resident batches.

    batches: iterable of dicts with the blobs Refiner.load takes plus "pose_observed" (B,3,4) ground truth.
    Collects all_poses_est[cls][iter] / all_poses_gt[cls][iter] and the rotation / translation errors per iteration exactly as
    the reference does (:497-560), writes the result cache [all_rot_err, all_trans_err, all_poses_est, all_poses_gt] with
    pickle protocol 2 (:650-654) and runs evaluate_pose / evaluate_pose_add / evaluate_pose_arp_2d (:664-673).
    evaluator: lib.dataset.evaluation.PoseEvaluator.  Returns the three result dicts."""
    import pickle

    from lib.utils.pose_error import calc_rt_dist_m

    n_cls, n_it = len(evaluator.classes), int(config.TEST.test_iter)
    all_rot_err = [[[] for _ in range(n_it)] for _ in range(n_cls)]
    all_trans_err = [[[] for _ in range(n_it)] for _ in range(n_cls)]
    all_poses_est = [[[] for _ in range(n_it)] for _ in range(n_cls)]
    all_poses_gt = [[[] for _ in range(n_it)] for _ in range(n_cls)]
    # flow error of the first forward (:500-512): only the full test graph emits the flow head's output
    epe = FlowEPE(config, refiner.B, refiner.net.device) if (config.network.PRED_FLOW and not config.TEST.FAST_TEST) else None
    for batch in batches:
        refiner.load(batch["image_observed"], batch["image_rendered"], batch["mask_observed"], batch["mask_rendered"], batch["src_pose"],
                     batch["class_index"])
        poses = refiner.refine().cpu().numpy().astype(np.float64)     # ONE device->host copy per batch: (iter, B, 3, 4)
        cls = torch.as_tensor(batch["class_index"]).cpu().numpy().astype(int)
        gt = torch.as_tensor(batch["pose_observed"]).cpu().numpy().astype(np.float64)
        src = torch.as_tensor(batch["src_pose"]).cpu().numpy().astype(np.float64)
        if epe is not None:
            epe.add(batch, refiner.flow_est_iter[0], skip=np.sum(src.reshape(src.shape[0], -1), axis=1) == -12)
        for b in range(poses.shape[1]):
            # "NO POINT VALID IN INIT POSE" (:419-445): an undetected object comes with pose_rendered = -1 everywhere (sum -12); it is
            # scored with its initial pose and 1000 deg / 1000 m at every iteration instead of being refined
            undetected = np.sum(src[b]) == -12
            for it in range(n_it):
                est = src[b] if undetected else poses[it, b]
                r_dist, t_dist = (1000, 1000) if undetected else calc_rt_dist_m(est, gt[b])
                all_poses_est[cls[b]][it].append(est)
                all_poses_gt[cls[b]][it].append(gt[b])
                all_rot_err[cls[b]][it].append(r_dist)
                all_trans_err[cls[b]][it].append(t_dist)
    # several ranks refine disjoint shards (one process per GPU): the metrics are over ALL pairs, so the per-class lists are merged in
    # rank order on every rank before scoring (the reference scores one list in one process)
    import torch.distributed as dist

    merged = False
    if merge_ranks and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, (all_rot_err, all_trans_err, all_poses_est, all_poses_gt))
        for k, mine in enumerate((all_rot_err, all_trans_err, all_poses_est, all_poses_gt)):
            for c in range(n_cls):
                for it in range(n_it):
                    mine[c][it] = [x for part in parts for x in part[k][c][it]]
        merged = True
        if dist.get_rank() != 0:
            result_file = None   # one result cache, written by rank 0
    if result_file:
        with open(result_file, "wb") as f:
            pickle.dump([np.array(all_rot_err, dtype=object), np.array(all_trans_err, dtype=object), all_poses_est, all_poses_gt], f,
                        protocol=2)
    out = {}
    if epe is not None:   # :656-660, before the pose tables like the reference
        out["epe"] = epe.result(merge_ranks=merge_ranks)
        for line in ("evaluate flow:", "EPE all: {}".format(out["epe"]["epe_all"]), "EPE ignore unvisible: {}".format(out["epe"]["epe_vizbg"]),
                     "EPE visible: {}".format(out["epe"]["epe_viz"])):
            print(line)
            if logger:
                logger.info(line)
    out["pose"] = evaluator.evaluate_pose(config, all_poses_est, all_poses_gt, logger)
    out["add"] = evaluator.evaluate_pose_add(config, all_poses_est, all_poses_gt, output_dir=None, logger=logger)
    out["arp_2d"] = evaluator.evaluate_pose_arp_2d(config, all_poses_est, all_poses_gt, output_dir=None, logger=logger)
    out["all_rot_err"], out["all_trans_err"] = all_rot_err, all_trans_err
    out["merged_over_ranks"] = merged
    return out
