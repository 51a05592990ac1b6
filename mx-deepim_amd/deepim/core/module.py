"""Training executor of the DeepIM graph on the HIP kernels.

Mirror of the reference's training path: `MutableModule.fit`'s inner loop (/root/reference/deepim/core/module.py:1205-1213:
forward_backward -> get_outputs -> update), the per-GPU executor (deepim/core/DataParallelExecutorGroup.py) and the
kvstore-based SGD (`Module.init_optimizer` :533-623, `update` :666-688), for ONE process = ONE GPU:

    forward_backward(batch)   train graph of deepim/symbols/deepIM_flownet.py:562-762 + get_loss :303-560, then the full
                              backward (loss gradients -> heads -> decoder -> encoder) on hand-written HIP kernels
    update()                  RCCL all-reduce(SUM) of the flat fp32 gradient (replaces kvstore push/pull; rescale_grad = 1.0,
                              deepim/train.py:383) + mx.optimizer.SGD(momentum, wd) on the flat parameter vector

Master parameters keep the MXNet layouts and names (one flat fp32 vector, 57.75 M elements); the kernels' packed copies
(forward + dgrad layouts) are refreshed after every update.
"""
from __future__ import print_function, division

import os

import numpy as np
import torch

from lib.hip import ops
from lib.utils.dist_utils import allreduce_sum_
from deepim.symbols.deepIM_flownet import bf16_tile, BF16_PATCH, ENCODER, FlowNetHip, deepIM_flownet

FUSED_UNPACK = os.environ.get("DIM_WGRAD_FUSED_UNPACK", "1") != "0"   # encoder weight gradients through dim_conv2d_wgrad_oihw
FOLD_LRELU = os.environ.get("DIM_BF16_FOLD_LRELU", "1") != "0"         # bf16: LeakyReLU' + bias gradient inside the input-gradient epilogue
FROZEN = ("upsampling_weight", "mask_upsampling_weight")  # attr lr_mult 0.0 (deepIM_flownet.py:334, :520)

# Order in which backward() finishes the weight gradients: heads, decoder, pose head, fc6, then the encoder top down.  The flat
# gradient vector keeps its weights in THIS order, so "everything backward has produced so far" is always one contiguous range and
# a bucket can be handed to RCCL while the rest of backward still runs.  Buckets close after the named tensor (bytes at fp32):
#   fc6_weight   heads + decoder + pose head + fc6      136 MB   (ready after ~15 % of backward)
#   conv5_weight conv6_1, conv6, conv5_1, conv5          75 MB
#   (end)        conv4_1 ... flow_conv1 + every bias      21 MB
BACKWARD_ORDER = ["Convolution3", "mask_conv3", "upsample_flow5to4", "deconv4", "Convolution2", "upsample_flow6to5", "deconv5", "Convolution1",
                  "rot", "trans", "fc7", "fc6"] + [l[0] for l in reversed(ENCODER)]
BUCKET_ENDS = ("fc6_weight", "conv5_weight")


class MutableModule(object):
    def __init__(self, config, arg_params, batch_size, device="cuda:0", process_group=None, compute_dtype="f32", overlap_allreduce=True):
        """compute_dtype "f32": the reference's precision (deepim/train.py:338-414 trains in fp32).
        "bf16": BASELINE configs[2] -- every convolution / large deconvolution forward, input gradient and weight gradient on the bf16
        matrix pipe with f32 accumulation; master weights, momentum, gradients, losses, SE(3) and every small kernel stay fp32; the
        gradient bucket crosses the ranks as bf16 (115.5 MB instead of 231 MB).  New functionality with a declared tolerance against
        the fp32 path (tests/test_gpu_bf16.py, tests/test_gpu_train_bf16.py).
        overlap_allreduce: hand the gradient to the collective in three buckets while backward is still running (see BACKWARD_ORDER);
        False = one all-reduce of the whole vector inside update(), the first version.  Same sums either way."""
        cfg = config
        assert compute_dtype in ("f32", "bf16"), compute_dtype
        self.bf16 = compute_dtype == "bf16"
        # heads of this configuration (every shipped one has both): the decoder exists when either does (deepIM_flownet.py:213)
        self.pred_flow, self.pred_mask = bool(cfg.network.PRED_FLOW), bool(cfg.network.PRED_MASK)
        self.has_decoder = self.pred_flow or self.pred_mask
        if not (cfg.train_iter.SE3_PM_LOSS or cfg.train_iter.SE3_DIST_LOSS):
            raise Exception("no pose loss: set train_iter.SE3_PM_LOSS and / or train_iter.SE3_DIST_LOSS")
        for what, kind in (("SE3_PM_LOSS_TYPE", cfg.train_iter.SE3_PM_LOSS_TYPE), ("TRANS_LOSS_TYPE", cfg.train_iter.TRANS_LOSS_TYPE)):
            if kind not in ops.LOSS_TYPE_ID:   # the reference raises for anything else too (deepIM_flownet.py:427-431, :486-491)
                raise Exception("Unknown {}: {}".format(what, kind))
        self.cfg = cfg
        self.B = batch_size
        self.device = torch.device(device)
        with torch.cuda.device(self.device):
            # compute units of THIS device, as the library reads them (conv.hip sizes its grids from the same number): the split plans
            # of the weight and input gradients fill "resident workgroup slots" = a small multiple of it
            self.n_cu = int(ops.lib().dim_device_info(None, 0))
        if self.n_cu <= 0:
            raise RuntimeError("dim_device_info: {}".format(ops.lib().dim_last_error()))
        self.pg = process_group
        d = self.device
        sym = deepIM_flownet()
        shapes = sym.infer_param_shapes(cfg)
        # ---- flat master parameters / gradients / momentum, MXNet layouts, in the shape table's order
        # grouped so that one optimizer launch covers a whole class: [weights (wd) | biases (wd_mult 0) | frozen (lr_mult 0)]
        keys = list(shapes.keys())
        wnames = [n + "_weight" for n in BACKWARD_ORDER if n + "_weight" in shapes]
        assert sorted(wnames) == sorted(n for n in keys if n not in FROZEN and n.endswith("_weight")), "BACKWARD_ORDER misses a layer"
        self.names = wnames + [n for n in keys if n not in FROZEN and not n.endswith("_weight")] + [n for n in keys if n in FROZEN]
        self.shapes = shapes
        sizes = [int(np.prod(shapes[n])) for n in self.names]
        # every tensor starts on a 128-byte boundary of the flat vectors (32 floats; 64 B in the bf16 bucket image): the vector kernels
        # (f32 <-> bf16, SGD / Adam: 16-byte accesses) and RCCL's aligned path need it for bucket and segment starts, and the sum of
        # the head + decoder + fc6 sizes is 2 mod 4.  The gaps hold zeros in all three vectors and stay zero under SGD and Adam.
        ALIGN = 32
        padded = [-(-sz // ALIGN) * ALIGN for sz in sizes]
        total = sum(padded)
        self.flat_w = torch.zeros(total, dtype=torch.float32, device=d)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=d)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=d)
        self.flat_v = None  # second Adam state, allocated on first use
        self.flat_g16 = None  # bf16 image of the gradient bucket (bf16 mode, more than one rank)
        self.force_bf16_bucket = False  # tests: round the bucket through bf16 in a single process too
        # tests: another executor (same batch, forward done) whose stored activations supply every LeakyReLU' mask of backward() -- so
        # that a bf16-vs-f32 or permuted-vs-original gradient comparison measures arithmetic, not which side of zero a pre-activation
        # within the forward noise of it fell on (tests/test_gpu_train_bf16.py)
        self.lrelu_mask_from = None
        self.w, self.g, self.m = {}, {}, {}
        self.n_weight = sum(sz for n, sz in zip(self.names, padded) if n not in FROZEN and n.endswith("_weight"))
        self.n_bias = sum(sz for n, sz in zip(self.names, padded) if n not in FROZEN and not n.endswith("_weight"))
        off = 0
        off_of = {}
        for n, sz, psz in zip(self.names, sizes, padded):
            off_of[n] = off
            self.w[n] = self.flat_w[off:off + sz].view(shapes[n])
            self.g[n] = self.flat_g[off:off + sz].view(shapes[n])
            self.m[n] = self.flat_m[off:off + sz].view(shapes[n])
            self.w[n].copy_(torch.as_tensor(np.ascontiguousarray(arg_params[n]), dtype=torch.float32))
            off += psz
        self.offset_of = off_of   # first element of every tensor in the flat vectors (tensors start on 128-byte boundaries)
        # ---- forward executor shares the master tensors (params dict = views of flat_w)
        self.net = FlowNetHip.__new__(FlowNetHip)
        # gradient buckets [begin, end) of the flat vector, in backward order; the last one also carries the biases; frozen tensors
        # (zero gradient, never updated) are not sent at all
        self.overlap_allreduce = bool(overlap_allreduce)
        ends = [-(-(off_of[n] + int(np.prod(shapes[n]))) // ALIGN) * ALIGN for n in BUCKET_ENDS if n in off_of] + [self.n_weight + self.n_bias]
        self.buckets = [(a, b) for a, b in zip([0] + ends[:-1], ends) if b > a]
        self._pending = []      # (work handle or None, begin, end) of the buckets already handed to the collective
        self._next_bucket = 0
        self._init_forward(cfg, batch_size)
        self._init_backward(batch_size)
        self.num_update = 0

    # ------------------------------------------------------------------------------------------------------------
    def _init_forward(self, cfg, B):
        net = self.net
        FlowNetHip.__init__(net, cfg, {n: self.w[n].cpu().numpy() for n in self.names}, B, device=str(self.device), winograd=True,
                            bf16=self.bf16, wino_s2=False)   # training keeps conv4 / conv5 on the direct kernel (their packed form is re-made per update)
        net.params = self.w  # the executor reads biases / small weights straight from the master vector
        d = self.device
        H, W = 480, 640
        self.zoom_mask_gt = torch.empty((B, 1, H, W), dtype=torch.float32, device=d)
        self.zoom_flow_lab = torch.empty((B, 2, H, W), dtype=torch.float32, device=d)
        self.zoom_flow_w = torch.empty((B, 2, H, W), dtype=torch.float32, device=d)
        self.flow_est_crop = torch.empty((B, 2, H, W), dtype=torch.float32, device=d)
        self.mask_logit = torch.empty((B, 1, H, W), dtype=torch.float32, device=d)
        self.mask_prob = torch.empty((B, 1, H, W), dtype=torch.float32, device=d)
        self.rot_norm = torch.empty((B, 4), dtype=torch.float32, device=d)
        self.rot_raw = torch.empty((B, 4), dtype=torch.float32, device=d)     # se3[:, :4] / se3[:, 4:] as their own arrays
        self.trans_est = torch.empty((B, 3), dtype=torch.float32, device=d)
        self.loss_sums = torch.zeros(5, dtype=torch.float32, device=d)  # flow, pm, -, rot, trans (un-scaled sums; metrics only)
        self.T_means = np.asarray(cfg.dataset.trans_means, dtype=np.float32)
        self.T_stds = np.asarray(cfg.dataset.trans_stds, dtype=np.float32)

    def _init_backward(self, B):
        d, net = self.device, self.net
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=d)  # noqa: E731
        self.dacts = {name: torch.empty_like(net.acts[name]) for name, *_ in ENCODER}
        # flow_conv1 with 6 or 10 input channels: 8-lane images of the master weight / of its gradient (spare lanes stay zero)
        nl = 0 if net.cin == 8 else (2 if net.input_mode == 3 else 1)
        self.w1_lanes = [z(64, 8, 7, 7) for _ in range(nl)]
        self.g1_lanes = [z(64, 8, 7, 7) for _ in range(nl)]
        self.dconcat3 = z(B, 30, 40, ops.pad64(770))
        self.dconcat2 = z(B, 15, 20, ops.pad64(1026))
        self.dflow4, self.dmask4 = z(B, 30, 40, 2), z(B, 30, 40, 1)
        self.dflow5, self.dflow6 = z(B, 15, 20, 2), z(B, 8, 10, 2)
        self.dflow_full, self.dlogit = z(B, 2, 480, 640), z(B, 1, 480, 640)
        self.d_rot_norm, self.d_rot, self.d_trans = z(B, 4), z(B, 4), z(B, 3)
        self.dz7, self.dz6 = z(B, 256), z(B, 256)
        self.pts_est = None
        self.dpts = None
        # wgrad plan: split the pixel range until the grid has ~4096 workgroups (measured at B = 16: 1024 -> 15.6 ms per backward,
        # 2048 -> 14.7, 4096 -> 14.3, 8192 -> 14.1; short workgroups hide the gather latency better and balance the CUs);
        # scratch = biggest slab set / packed gradient
        self.wgrad_splits, max_ws, max_pack = {}, 4, 4
        h, w, c = 480, 640, 8
        for name, cout, k, s, p in ENCODER:
            ho, wo = ops.conv_out_hw(h, w, k, k, s, p)
            nchunks = -(-k * k // 4) if c == 8 else k * k * (c // 32)
            blocks = nchunks * (cout // 128 if cout % 128 == 0 else cout // 64)
            nsteps = -(-B * ho * wo // 32)
            sp = max(1, min(-(-4096 // blocks), max(1, nsteps // 4)))
            if self.bf16:
                # the WHOLE grid must be resident at once (2-4 workgroups per CU by kernel form): a grid one workgroup over that runs a
                # second, nearly empty round in which nothing overlaps the phases of a step.  tools/wgrad_sweep.py at B = 16,
                # workgroups: us -- conv3_1 756: 155, 1044: 187, 1512: 168; conv2 767: 252, 1027: 318 (the round-2 rule aimed at ~1024
                # with a ceiling division and overshot the 768 slots on every layer).  One plan function for every caller:
                sp = ops.lib().dim_conv2d_wgrad_bf16_splits(B, h, w, c, cout, k, k, s, p, self.n_cu)
            self.wgrad_splits[name] = sp
            max_ws = max(max_ws, ops.lib().dim_conv2d_wgrad_workspace_floats(cout, c, k, k, sp + 1))   # + 1: dim_conv2d_wgrad_oihw
            if self.bf16 and c != 8:   # split-K slabs of this layer's input gradient (copies of dX), see _dgrad_splits
                ksp = self._dgrad_splits(self._bf16_gemm_tile(B * ho * wo, ops.pad64(c)), B * ho * wo, ops.pad64(c), cout)
                max_ws = max(max_ws, ops.lib().dim_conv2d_dgrad_splitk_workspace_floats(B, h, w, ops.pad64(c), ksp))
            max_pack = max(max_pack, ops.lib().dim_conv2d_packed_weight_floats(cout, c, k, k))
            h, w, c = ho, wo, cout
        max_pack = max(max_pack, 256 * 81920, 4 * 4 * 512 * 1024, 4 * 4 * 256 * ops.pad64(1026))
        max_ws = max(max_ws, ops.lib().dim_conv2d_wgrad_workspace_floats(ops.pad64(1026), 256, 4, 4, 3),   # deconv4's split gradient (_deconv_bwd)
                     ops.lib().dim_conv_small_cout_bwd_workspace_floats(B, 30, 40, 770, 2, 3, 3),
                     ops.lib().dim_conv_small_cout_bwd_workspace_floats(B, 15, 20, 1026, 2, 3, 3),
                     ops.lib().dim_conv_small_cout_bwd_workspace_floats(B, 8, 10, 1024, 2, 3, 3))
        self.ws = torch.empty(max_ws, dtype=torch.float32, device=d)
        self.gpack = torch.empty(max_pack, dtype=torch.float32, device=d)
        need_b, h, w = ops.lib().dim_bias_grad_workspace_floats(B * 240 * 320, 64) + 1024 * 64, 480, 640
        for name, cout, k, s, p in ENCODER:   # partial column sums of the fused LeakyReLU' + bias-gradient pass
            h, w = ops.conv_out_hw(h, w, k, k, s, p)
            need_b = max(need_b, ops.lib().dim_lrelu_bwd_bias_grad_workspace_floats(B * h * w, cout))
        self.bias_ws = torch.empty(need_b, dtype=torch.float32, device=d)
        self.fold_ws = None
        if self.bf16 and FOLD_LRELU:   # partial column sums of the folded input gradients (backward)
            need_f, h, w, c = 4, 480, 640, 8
            for name, cout, k, s, p in ENCODER:
                if c % 64 == 0:
                    need_f = max(need_f, ops.lib().dim_conv2d_dgrad_lrelu_workspace_floats(B, h, w, c, s))
                h, w = ops.conv_out_hw(h, w, k, k, s, p)
                c = cout
            self.fold_ws = torch.empty(need_f, dtype=torch.float32, device=d)
        # dgrad-layout weights (refreshed by repack())
        self.dgrad_packed = {}
        self.wino_dgrad = {}
        self.wino5_dgrad = {}
        self.wino_ws = None
        if net.wino or net.wino5:
            need, h, w, c = 4, 480, 640, 8
            for name, cout, k, s, p in ENCODER:
                if name in net.wino:
                    need = max(need, ops.lib().dim_winograd_workspace_floats(B, h, w, cout, c, net.wino_m[name]))
                if name in net.wino5:
                    need = max(need, ops.lib().dim_winograd5x5s2_workspace_floats(B, h, w, c, cout))
                h, w = ops.conv_out_hw(h, w, k, k, s, p)
                c = cout
            self.wino_ws = torch.empty(need, dtype=torch.float32, device=d)
        # Winograd weight gradients (3x3 / stride-1 and 5x5 / stride-2 layers whose maps are large enough: at 8x10 the 36 plane
        # products are 3 pixel steps long and their 151 MB of dM cost more than the direct kernel saves): {layer: (S, pixel splits)}
        self.wino_wgrad = {}
        if os.environ.get("DIM_WINO_WGRAD", "1") != "0" and not self.bf16:
            need, h, w, c = 4, 480, 640, 8
            for name, cout, k, s, p in ENCODER:
                ho, wo = ops.conv_out_hw(h, w, k, k, s, p)
                S = 1 if name in net.wino else 2 if name in net.wino5 else 0
                tiles = B * (-(-ho // 4)) * (-(-wo // 4))
                if S and tiles >= 256 and (c * S * S) % 64 == 0:
                    blocks = 36 * (c * S * S // 64) * (cout // 128 if cout % 128 == 0 else cout // 64)
                    sp = max(1, min(-(-2048 // blocks), max(1, (tiles // 32) // 4)))
                    self.wino_wgrad[name] = (S, sp)
                    need = max(need, ops.lib().dim_conv2d_wgrad_winograd_workspace_floats(B, h, w, c, cout, S, sp))
                h, w, c = ho, wo, cout
            self.wino_wgrad_ws = torch.empty(need, dtype=torch.float32, device=d)
        self.repack(forward=False)

    # ------------------------------------------------------------------------------------------------------------
    def repack(self, forward=True):
        """master (MXNet layout) -> the kernels' packed copies: forward layouts (FlowNetHip.packed) and dgrad layouts
        (bf16 mode: bf16 images of the same packed arrays).  (Measured and dropped in round 3: packing everything that is first read
        at fc6 or later on a second stream behind the optimizer step, overlapped with the next encoder forward -- 6.40 vs 6.41 ms per
        bf16 iteration: the packers and the now HBM-bound first layers want the same bytes per second.)"""
        net, w = self.net, self.w
        for name, cout, k, s, p in ENCODER:
            if forward:
                if name in net.wino:   # 3x3 / stride-1 layers run their forward through Winograd: re-transform the weights
                    net.wino[name] = ops.winograd_pack_weight(w[name + "_weight"], m=net.wino_m[name])
                elif name in net.wino5:  # 5x5 / stride-2 layers: phase-image Winograd forward (backward: wino5_dgrad below)
                    net.wino5[name] = ops.winograd5x5s2_pack_weight(w[name + "_weight"])
                elif name == "flow_conv1" and net.cin != 8:
                    # 6 input channels (no masks in the Concat) or 10 (depth planes and masks): the kernels take 8-lane groups
                    ops.copy_channels(self.w1_lanes[0], 0, w[name + "_weight"], 0, min(net.cin, 8))
                    net.packed[name] = net.pack_conv(self.w1_lanes[0])
                    if net.input_mode == 3:
                        ops.copy_channels(self.w1_lanes[1], 0, w[name + "_weight"], 8, 2)
                        net.packed["flow_conv1_masks"] = net.pack_conv(self.w1_lanes[1])
                else:
                    net.packed[name] = net.pack_conv(w[name + "_weight"])
            if name in net.wino:
                self.wino_dgrad[name] = ops.winograd_dgrad_pack_weight(w[name + "_weight"], m=net.wino_m[name])
            elif name in net.wino5:  # 5x5 / stride-2 layers: input gradient through Winograd too (four phase images of dX)
                self.wino5_dgrad[name] = ops.winograd5x5s2_dgrad_pack_weight(w[name + "_weight"])
            elif name != "flow_conv1":
                self.dgrad_packed[name] = ops.conv2d_dgrad_pack_weight(w[name + "_weight"], s, p, as_bf16=self.bf16)
        if forward:
            net.packed["fc6"] = ops.fc_pack_weight(w["fc6_weight"], 1024, 8, 10)
            if self.has_decoder:
                net.packed["deconv5"] = net.pack_deconv(w["deconv5_weight"])
                net.packed["deconv4"] = net.pack_deconv(w["deconv4_weight"])
            for n in ("Convolution1", "Convolution2", "Convolution3", "mask_conv3"):
                if n + "_weight" in w:
                    net.packed[n] = ops.conv_small_cout_pack_weight(w[n + "_weight"])
        self.dgrad_packed["fc6"] = ops.fc_dgrad_pack_weight(w["fc6_weight"], 1024, 8, 10, as_bf16=self.bf16)
        if self.has_decoder:
            # deconv dgrad = a plain stride-2 convolution of the output gradient with the deconv weight read as (O=Cin, I=Cout, 4, 4)
            self.dgrad_packed["deconv5"] = ops.conv2d_pack_weight_padded(w["deconv5_weight"], 1024, as_bf16=self.bf16)
            self.dgrad_packed["deconv4"] = ops.conv2d_pack_weight_padded(w["deconv4_weight"], ops.pad64(1026), as_bf16=self.bf16)

    # ------------------------------------------------------------------------------------------------------------
    def forward(self, batch):
        """train graph forward (get_train_symbol :562-762).  batch: reference blob names (data + labels), CUDA tensors."""
        cfg, net = self.cfg, self.net
        H, W = 480, 640
        if cfg.network.INPUT_MASK or self.pred_mask:
            # ZoomMask: the zoom window comes from mask_GT_observed and mask_rendered (zoom_mask.py:35-37; get_train_symbol :589-612)
            ops.mask_bbox(batch["mask_gt_observed"], 0.3, out=net.bbox_obs)
            ops.mask_bbox(batch["mask_rendered"], 0.2, out=net.bbox_ren)
        else:
            # ZoomImage (:625-640, zoom_image.py:31-37): the validity "mask" of an image is sum_c(image + mean) > 0.01
            ops.mask_bbox(batch["image_observed"], 0.01, mode=1, means3=net.plane_means, out=net.bbox_obs)
            ops.mask_bbox(batch["image_rendered"], 0.01, mode=1, means3=net.plane_means, out=net.bbox_ren)
        ops.zoom_factor(net.bbox_obs, net.bbox_ren, batch["src_pose"], net.K, H, W, out=net.zoom_factor, status=net.status)
        net.net_input(batch)   # images (+ depth planes) (+ masks): the Concat of get_convs (:33-66) for this configuration's arity
        if self.pred_mask:
            ops.zoom_planes(batch["mask_gt_observed"], net.zoom_factor, post=1, out=self.zoom_mask_gt)
        if self.pred_flow:
            ops.zoom_planes(batch["flow"], net.zoom_factor, scale_mode=1, out=self.zoom_flow_lab)          # ZoomFlow :689-698
            ops.zoom_planes(batch["flow_weights"], net.zoom_factor, post=2, out=self.zoom_flow_w)
        net.encoder()
        net.head()                                               # se3 = [rot (raw), inverse-zoomed trans]; fc7
        ops.copy_nhwc_channels(self.rot_raw, 0, net.se3, 0, 4)
        ops.quat_normalize(self.rot_raw, out=self.rot_norm)   # L2Normalization :375
        p = self.w
        if self.has_decoder:
            net.decoder()
        if self.pred_flow:
            ops.conv_small_cout_fwd(net.concat3, 770, net.packed["Convolution3"], p["Convolution3_bias"], 2, out=net.flow4)
            ops.upsample16_fwd(net.flow4, p["upsampling_weight"], H, W, crop=8, out=self.flow_est_crop)
        if self.pred_mask:
            ops.conv_small_cout_fwd(net.concat3, 770, net.packed["mask_conv3"], p["mask_conv3_bias"], 1, out=net.mask4)
            ops.upsample16_fwd(net.mask4, p["mask_upsampling_weight"], H, W, crop=8, out=self.mask_logit)
        ops.copy_nhwc_channels(self.trans_est, 0, net.se3, 4, 3)
        pts = batch["point_cloud_model"]
        self.pts_est = ops.transform3d_fwd(pts, self.rot_norm, self.trans_est, batch["src_pose"], cfg.network.ROT_COORD, self.T_means,
                                           self.T_stds, out=self.pts_est)
        out = {"rot_est_norm": self.rot_norm, "trans_est": self.trans_est, "point_cloud_observed_est": self.pts_est, "zoom_factor": net.zoom_factor}
        if self.pred_flow:
            out["flow_est_crop"] = self.flow_est_crop
        if self.pred_mask:
            out["mask_logit"] = self.mask_logit
        return out

    def backward(self, batch):
        cfg, net, w, g = self.cfg, self.net, self.w, self.g
        ti = cfg.train_iter
        B = self.B
        # collectives of a previous backward() that no update() consumed still read / write flat_g (flat_g16) in place on the
        # communicator's stream: wait for them before this pass rewrites the gradients
        for work, _, _ in self._pending:
            if work is not None:
                work.wait()
        self._pending, self._next_bucket = [], 0
        ops.fill(self.loss_sums, 0.0)
        # ---------------- loss gradients (get_loss :344-357, :446-499, :531-536)
        if self.pred_flow:
            ops.flow_loss_grad(self.flow_est_crop, self.zoom_flow_lab, self.zoom_flow_w, self.dflow_full, cfg.dataset.NORMALIZE_FLOW,
                               ti.LW_FLOW / (480.0 * 640.0), loss_sum=self.loss_sums[0:1])
        if self.pred_mask:
            ops.logistic_grad(self.mask_logit, self.zoom_mask_gt, self.dlogit, ti.LW_MASK / (480.0 * 640.0), prob=self.mask_prob)
        if ti.SE3_PM_LOSS:   # point matching (:440-499): L1 / L2 / smooth_L1 on (Transform3D(model points) - observed points) / norm
            if self.dpts is None:
                self.dpts = torch.empty_like(self.pts_est)
            ops.pm_loss_grad(self.pts_est, batch["point_cloud_observed"], batch["point_cloud_weights"], self.dpts,
                             cfg.dataset.NORMALIZE_3D_POINT, ti.LW_PM / float(ti.NUM_3D_SAMPLE), ti.SE3_PM_LOSS_TYPE, ti.SE3_PM_SL1_SCALAR,
                             loss_sum=self.loss_sums[1:2])
            d_rn, d_t = ops.transform3d_bwd(self.dpts, batch["point_cloud_model"], self.rot_norm, self.trans_est, batch["src_pose"],
                                            cfg.network.ROT_COORD, self.T_means, self.T_stds)
        else:
            d_rn = torch.zeros((B, 4), dtype=torch.float32, device=self.device)
            d_t = torch.zeros((B, 3), dtype=torch.float32, device=self.device)
        if ti.SE3_DIST_LOSS:   # :396-437; labels "rot" / "trans" = calc_RT_delta(src, gt) as the data layer delivers them (data_pair.py:201-251)
            zt_gt = ops.zoom_trans(net.zoom_factor, batch["trans"].contiguous(), 1)   # ZoomTrans, b_inv_zoom False (:659-665)
            ops.se3_dist_loss_grad(self.rot_norm, batch["rot"].contiguous(), net.fc7, w, zt_gt, d_rn, d_t, ti.LW_ROT, ti.LW_TRANS,
                                   ti.TRANS_LOSS_TYPE, ti.TRANS_SMOOTH_L1_SCALAR, loss_sums2=self.loss_sums[3:5])
        d10 = self.dacts["conv6_1"]
        if self.has_decoder:
            self._decoder_backward()
        # ---------------- pose head (fc7, rot, trans, fc6)
        fc6a = net.fc6.view(B, 256)
        msk = self.lrelu_mask_from.net if self.lrelu_mask_from is not None else net   # whose activation signs gate the gradients
        ops.pose_head_bwd(msk.fc6.view(B, 256), msk.fc7, self.rot_raw, d_rn, d_t, w, self.d_rot, self.dz7, self.dz6)
        ops.fc_wgrad(self.d_rot, net.fc7, g["rot_weight"], g["rot_bias"])
        ops.fc_wgrad(d_t, net.fc7, g["trans_weight"], g["trans_bias"])
        ops.fc_wgrad(self.dz7, fc6a, g["fc7_weight"], g["fc7_bias"])
        dz6 = self.dz6.view(B, 1, 1, 256)
        if B <= 32:   # 16 products per element: one 84 MB write in the MXNet layout (f32) instead of the packed "convolution" + conversion
            ops.fc_wgrad_nhwc(self.dz6, net.acts["conv6_1"], g["fc6_weight"])
        else:
            ops.conv2d_wgrad(net.acts["conv6_1"], 1024, dz6, 256, 8, 10, 1, 0, self.gpack, bf16_mfma=self.bf16)
            ops.fc_unpack_weight(self.gpack, g["fc6_weight"], 1024, 8, 10)
        self._bucket_ready("fc6_weight")
        ops.bias_grad(dz6, 256, g["fc6_bias"], workspace=self.bias_ws)
        # d(ReLU10) += dz6 * W6  (fc6 dgrad as a 1x1 convolution to 81920 "channels" = the NHWC feature map)
        ops.conv2d_fwd_ex(dz6, 0, 256, self.dgrad_packed["fc6"], None, d10.view(B, 1, 1, 81920), 0, 81920, 1, 1, 1, 0,
                          accumulate=self.has_decoder)  # M = B rows: tile 3; without a decoder this is d10's only term
        # ---------------- encoder, top down
        prev = {ENCODER[i][0]: (ENCODER[i - 1][0] if i else None) for i in range(len(ENCODER))}
        cin = {}
        c = 8
        for name, cout, k, s, p in ENCODER:
            cin[name] = c
            c = cout
        dz_done = set()
        for name, cout, k, s, p in reversed(ENCODER):
            dy = self.dacts[name]
            if name == "conv5_1" and self.has_decoder:
                ops.copy_nhwc_channels(dy, 0, self.dconcat2, 0, 512, add=True)   # skip connection into Concat2
            if name == "conv4_1" and self.has_decoder:
                ops.copy_nhwc_channels(dy, 0, self.dconcat3, 0, 512, add=True)   # skip connection into Concat3
            if name not in dz_done:   # else: the layer above already delivered dz and the bias gradient (folded input gradient, below)
                ops.lrelu_bwd_bias_grad(msk.acts[name], dy, cout, g[name + "_bias"], workspace=self.bias_ws)   # dz in place + bias gradient
            x = net.acts[prev[name]] if prev[name] else net.X
            if name in self.wino_wgrad:
                S, sp = self.wino_wgrad[name]
                ops.conv2d_wgrad_winograd(x, cin[name], dy, cout, g[name + "_weight"], S=S, splits=sp, workspace=self.wino_wgrad_ws)
            elif name == "flow_conv1" and net.cin != 8:
                # one weight gradient per 8-lane input group, each copied into its channel window of the (64, cin, 7, 7) gradient
                for lane, xin, c0, nch in ((0, net.X, 0, min(net.cin, 8)),) + (((1, net.X2, 8, 2),) if net.input_mode == 3 else ()):
                    ops.conv2d_wgrad(xin, 8, dy, cout, k, k, s, p, self.gpack, splits=self.wgrad_splits[name], workspace=self.ws,
                                     bf16_mfma=self.bf16)
                    ops.conv2d_unpack_weight(self.gpack, self.g1_lanes[lane])
                    ops.copy_channels(g[name + "_weight"], c0, self.g1_lanes[lane], 0, nch)
            elif FUSED_UNPACK:
                # slabs summed inside the layout converter: no packed intermediate, no reduce launch (10 launches and ~0.3 GB less per backward)
                ops.conv2d_wgrad_oihw(x, cin[name], dy, cout, k, k, s, p, g[name + "_weight"], splits=self.wgrad_splits[name], workspace=self.ws,
                                      bf16_mfma=self.bf16)
            else:
                ops.conv2d_wgrad(x, cin[name], dy, cout, k, k, s, p, self.gpack, splits=self.wgrad_splits[name], workspace=self.ws,
                                 bf16_mfma=self.bf16)
                ops.conv2d_unpack_weight(self.gpack, g[name + "_weight"])
            self._bucket_ready(name + "_weight")
            if prev[name]:
                if name in self.wino_dgrad:
                    # dX of a 3x3 / stride-1 / pad-1 convolution = the same kind of convolution of dZ with the flipped, transposed
                    # kernel: Winograd like the forward (no bias, no activation)
                    wm = self.net.wino_m[name]
                    wtiles = dy.shape[0] * (-(-dy.shape[1] // wm)) * (-(-dy.shape[2] // wm))
                    ops.conv2d_fwd_winograd(dy, cout, self.wino_dgrad[name], None, cin[name], slope=1.0,
                                            tile=4 if (cin[name] % 128 == 0 and wtiles >= 1024) else 3,
                                            out=self.dacts[prev[name]], workspace=self.wino_ws, m=wm)
                elif name in self.wino5_dgrad:
                    ops.conv2d_dgrad_winograd5x5s2(dy, cout, self.wino5_dgrad[name], self.dacts[prev[name]], cin[name], workspace=self.wino_ws)
                else:
                    # bf16: operand-traffic bound, so the widest tile the channel count allows (dX channels are the GEMM's N)
                    dg_tile = self._bf16_gemm_tile(dy.shape[0] * dy.shape[1] * dy.shape[2], ops.pad64(cin[name])) if self.bf16 else 3
                    if (dg_tile != 3 and s == 1 and k == 3 and dy.shape[1] * dy.shape[2] >= 1200 and os.environ.get("DIM_BF16_HALO", "1") != "0"):
                        dg_tile = 7   # stride-1 input gradient of a large map: LDS-halo kernel
                    if self.bf16 and BF16_PATCH and k in (3, 5) and dy.shape[1] * dy.shape[2] >= 1200:
                        dg_tile = 9   # stride-1 patch kernel: the gradient itself (stride 1) or its four phase convolutions (stride 2)
                    ksp = self._dgrad_splits(dg_tile, dy.shape[0] * dy.shape[1] * dy.shape[2], ops.pad64(cin[name]), cout) if self.bf16 else 1
                    if FOLD_LRELU and dg_tile == 9 and ksp == 1 and ((k == 5 and s == 2) or (k == 3 and s == 1)) and cin[name] % 64 == 0 and not (
                            self.has_decoder and prev[name] in ("conv4_1", "conv5_1")):
                        # every phase runs on the patch kernel and this is the only gradient that reaches the layer below: its
                        # LeakyReLU' and bias gradient ride in the epilogue instead of a 12-bytes-per-element pass of their own
                        # (flow_conv1, conv2, conv3, conv4: 80 % of the encoder's activation elements)
                        ops.conv2d_dgrad_lrelu(dy, cout, self.dgrad_packed[name], self.dacts[prev[name]], msk.acts[prev[name]], cin[name], k, k, s, p,
                                               g[prev[name] + "_bias"], workspace=self.fold_ws)
                        dz_done.add(prev[name])
                        continue
                    ops.conv2d_dgrad(dy, cout, self.dgrad_packed[name], self.dacts[prev[name]], cin[name], k, k, s, p, accumulate=False,
                                     tile=dg_tile, splits=ksp, workspace=self.ws if ksp > 1 else None)
        self._bucket_ready(None)
        return g

    # ------------------------------------------------------------------------------------------------------------
    def _bucket_ready(self, name):
        """backward() has just enqueued the last kernel that writes gradient tensor `name` (None: the end of backward): if that closes
        a bucket, start its all-reduce(SUM).  With the nccl (= RCCL) backend an async collective runs on the communicator's own stream,
        ordered after everything enqueued on the compute stream so far, so the transfer over xGMI overlaps the rest of backward;
        update() waits for the handles.  Single process: nothing to do."""
        import torch.distributed as dist

        if not self.overlap_allreduce or self.world_size() == 1 and not self.force_bf16_bucket:
            return
        k = self._next_bucket
        if k >= len(self.buckets):
            return
        closes = (name is None and k == len(self.buckets) - 1) or (name is not None and k < len(BUCKET_ENDS) and name == BUCKET_ENDS[k])
        if not closes:
            return
        a, b = self.buckets[k]
        self._next_bucket += 1
        self._pending.append((self._start_allreduce(a, b), a, b))

    def _start_allreduce(self, a, b):
        import torch.distributed as dist

        buf = self.flat_g[a:b]
        if self.bf16:
            if self.flat_g16 is None:
                self.flat_g16 = torch.empty(self.flat_g.shape, dtype=torch.bfloat16, device=self.device)
            buf = ops.to_bf16(self.flat_g[a:b], out=self.flat_g16[a:b])
        if self.world_size() == 1:
            return None
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def snapshot_lrelu_masks(self, perm=None):
        """tests: a frozen copy of the activations whose signs gate backward() (every encoder layer, the two decoder concat buffers,
        fc6 / fc7), optionally with the batch dimension permuted -- assign it to another executor's (or this one's) `lrelu_mask_from`.
        -> object with a `.net` attribute shaped like FlowNetHip for exactly those reads"""
        import types

        net = self.net
        take = (lambda t: t.clone()) if perm is None else (lambda t: t[perm].contiguous())
        snap = types.SimpleNamespace(acts={k: take(v) for k, v in net.acts.items()}, fc6=take(net.fc6), fc7=take(net.fc7),
                                     concat2=take(net.concat2) if self.has_decoder else None,
                                     concat3=take(net.concat3) if self.has_decoder else None)
        return types.SimpleNamespace(net=snap)

    def count_lrelu_flips(self, other):
        """tests: {layer: (units whose LeakyReLU branch differs between this executor's stored activations and `other`'s, units)}"""
        out = {}
        mine, theirs = self.net, other.net
        for k, a in mine.acts.items():
            b = theirs.acts[k]
            out[k] = (int(((a > 0) != (b > 0)).sum().item()), a.numel())
        for k in ("fc6", "fc7"):
            a, b = getattr(mine, k), getattr(theirs, k)
            out[k] = (int(((a > 0) != (b > 0)).sum().item()), a.numel())
        return out

    def pending_buckets(self):
        """[(begin, end, completed)] of the buckets handed to the collective and not yet consumed by update() -- for a watchdog
        that has to say WHICH transfer never finished"""
        return [(a, b, True if w is None else bool(w.is_completed())) for w, a, b in self._pending]

    def _finish_allreduce(self):
        """every bucket summed over the ranks and back in flat_g as fp32 (buckets not started yet -- overlap off, or update() without
        a backward() before it -- are reduced here)"""
        if self.world_size() > 1 or self.force_bf16_bucket:
            while self._next_bucket < len(self.buckets):
                a, b = self.buckets[self._next_bucket]
                self._next_bucket += 1
                self._pending.append((self._start_allreduce(a, b), a, b))
        for work, a, b in self._pending:
            if work is not None:
                work.wait()
            if self.bf16:
                ops.from_bf16(self.flat_g16[a:b], out=self.flat_g[a:b])
        self._pending = []
        self._next_bucket = 0

    def _decoder_backward(self):
        """heads and decoder (deepIM_flownet.py:213-299), top down: leaves d(ReLU10) in dacts["conv6_1"] (overwritten), the skip
        terms of ReLU8 / ReLU6 in dconcat2[..., :512] / dconcat3[..., :512]"""
        net, w, g = self.net, self.w, self.g
        d10 = self.dacts["conv6_1"]
        # ---------------- flow / mask heads (whichever exist; the first one overwrites d(Concat3), the second adds)
        if self.pred_flow:
            ops.upsample16_bwd(self.dflow_full, w["upsampling_weight"], self.dflow4)
            ops.conv_small_cout_bwd(net.concat3, 770, self.dflow4, w["Convolution3_weight"], self.dconcat3, g["Convolution3_weight"],
                                    g["Convolution3_bias"], accumulate_dx=False, workspace=self.ws)
        if self.pred_mask:
            ops.upsample16_bwd(self.dlogit, w["mask_upsampling_weight"], self.dmask4)
            ops.conv_small_cout_bwd(net.concat3, 770, self.dmask4, w["mask_conv3_weight"], self.dconcat3, g["mask_conv3_weight"],
                                    g["mask_conv3_bias"], accumulate_dx=self.pred_flow, workspace=self.ws)
        # ---------------- decoder level 4: Concat3 = [ReLU6 | ReLU12 (deconv4) | upsample_flow5to4]
        ops.deconv4x4s2_tiny_bwd(net.flow5, self.dconcat3, 768, w["upsample_flow5to4_weight"], self.dflow5, g["upsample_flow5to4_weight"],
                                 g["upsample_flow5to4_bias"])
        msk = self.lrelu_mask_from.net if self.lrelu_mask_from is not None else net
        ops.lrelu_bwd(msk.concat3, self.dconcat3, 256, y_coff=512, dy_coff=512)
        self._deconv_bwd("deconv4", x=net.concat2, x_c=1026, x_cpad=ops.pad64(1026), dz=self.dconcat3, dz_coff=512, cout=256, dx=self.dconcat2)
        ops.conv_small_cout_bwd(net.concat2, 1026, self.dflow5, w["Convolution2_weight"], self.dconcat2, g["Convolution2_weight"],
                                g["Convolution2_bias"], accumulate_dx=True, workspace=self.ws)
        # ---------------- decoder level 5: Concat2 = [ReLU8 | ReLU11 (deconv5) | upsample_flow6to5]
        ops.deconv4x4s2_tiny_bwd(net.flow6, self.dconcat2, 1024, w["upsample_flow6to5_weight"], self.dflow6, g["upsample_flow6to5_weight"],
                                 g["upsample_flow6to5_bias"])
        ops.lrelu_bwd(msk.concat2, self.dconcat2, 512, y_coff=512, dy_coff=512)
        self._deconv_bwd("deconv5", x=net.acts["conv6_1"], x_c=1024, x_cpad=1024, dz=self.dconcat2, dz_coff=512, cout=512, dx=d10)
        ops.conv_small_cout_bwd(net.acts["conv6_1"], 1024, self.dflow6, w["Convolution1_weight"], d10, g["Convolution1_weight"],
                                g["Convolution1_bias"], accumulate_dx=True, workspace=self.ws)

    def _deconv_bwd(self, name, x, x_c, x_cpad, dz, dz_coff, cout, dx):
        """Deconvolution(k4,s2)+Crop(1,1) backward.  x: deconv input (N,h,w,stride>=x_cpad), dz: gradient w.r.t. the pre-activation
        output = channels [dz_coff, dz_coff+cout) of a concat-gradient buffer over the crop window (N,oh,ow,stride)."""
        g, w = self.g, self.w
        N, h, wd, _ = x.shape
        # weight gradient through the convolution view: conv'(input = dz, k4, s2, pad 1) with "output gradient" = x
        # (bf16: deconv4's view is 544 workgroups of the 64-row kernel: three pixel splits fill the resident slots, 183 -> 151 us)
        sp = 3 if (self.bf16 and N * h * wd >= 4096 and x_cpad % 128 != 0) else 1
        ops.conv2d_wgrad_ex(dz, dz_coff, cout, x, 0, x_cpad, 4, 4, 2, 1, self.gpack, splits=sp, workspace=self.ws if sp > 1 else None,
                            bf16_mfma=self.bf16)
        ops.conv2d_unpack_weight(self.gpack, g[name + "_weight"], CoutPad=x_cpad)
        ops.bias_grad(dz, cout, g[name + "_bias"], dz_coff=dz_coff, workspace=self.bias_ws)
        # data gradient: the same convolution applied to dz
        ops.conv2d_fwd_ex(dz, dz_coff, cout, self.dgrad_packed[name], None, dx, 0, x_cpad, 4, 4, 2, 1, Ho=h, Wo=wd, accumulate=False,
                          tile=self._bf16_gemm_tile(N * h * wd, x_cpad) if self.bf16 else 3)

    def _dgrad_splits(self, tile, rows, cols, cout):
        """split-K count of a bf16 input gradient on the gathered-tap kernel: the small maps (conv5 .. conv6_1: 1280-4800 rows per
        launch) make 144-600 tiles of 64 x 64 for the 1280 workgroups that fit the chip (5 per CU), and each walks K = 4608-9216 alone.
        The K range is cut so that the grid fills the slots; every phase of a strided gradient must keep >= splits chunks of 32 dy
        channels (its 1-tap phase has cout / 32).  DIM_BF16_DGRAD_SPLITK=0: off."""
        if tile != 3 or os.environ.get("DIM_BF16_DGRAD_SPLITK", "1") == "0":
            return 1
        tiles = -(-rows // 64) * (cols // 64)
        for sp in (8, 4, 2):
            if sp * tiles <= 5 * self.n_cu and (cout // 32) % sp == 0:
                return sp
        return 1

    @staticmethod
    def _bf16_gemm_tile(rows, cols):
        """workgroup tile of a bf16 input-gradient GEMM with `rows` output pixels (per launch: a stride-2 gradient runs one launch per
        input phase, each with as many rows as dY has pixels) x `cols` channels: the wide tile only when it fills the chip -- conv6_1's
        gradient is 10 x 8 tiles of 128 x 128 with K = 9216, 80 workgroups on 256 CUs (149 us); 64 x 64 tiles give 320.
        DIM_BF16_SMALL_TILE=0: the wide tile wherever the channel count allows."""
        t = bf16_tile(cols)
        if t != 3 and os.environ.get("DIM_BF16_SMALL_TILE", "1") != "0" and -(-rows // 128) * (cols // 128) < 200:
            return 3
        return t

    def forward_backward(self, batch):
        out = self.forward(batch)
        self.backward(batch)
        return out

    # ------------------------------------------------------------------------------------------------------------
    def update(self, lr):
        """kvstore push/pull + optimizer (module.py:666-688; train.py:338-385): sum gradients over ranks, then the same update on
        every rank.  Two launches: weights (weight decay) and biases (MXNet sets wd_mult 0 for *_bias); frozen tensors are skipped.
        `lr` is this update's learning rate (a WarmupMultiFactorScheduler value)."""
        cfg = self.cfg
        if not self.overlap_allreduce and not self.bf16:
            allreduce_sum_(self.flat_g[:self.n_weight + self.n_bias], group=self.pg)   # one transfer (frozen tensors excluded)
            self._pending, self._next_bucket = [], 0
        else:
            self._finish_allreduce()
        self.num_update += 1
        nw, nb = self.n_weight, self.n_bias
        seg = ((0, nw, True), (nw, nw + nb, False))
        if str(cfg.TRAIN.optimizer).lower() == "adam":
            # mx.optimizer.Adam defaults (train.py:339 passes only learning_rate): beta1 .9, beta2 .999, eps 1e-8, wd 0
            if self.flat_v is None:
                self.flat_v = torch.zeros_like(self.flat_m)
            b1, b2, t = 0.9, 0.999, self.num_update
            lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
            # only the SGD branch of train.py:383 forces rescale_grad = 1.0; for Adam Module.init_optimizer (module.py:566-584)
            # fills in 1 / batch_size, batch_size = pairs over all devices
            rescale = 1.0 / float(self.B * self.world_size())
            for a, b, _ in seg:
                ops.adam(self.flat_w[a:b], self.flat_g[a:b], self.flat_m[a:b], self.flat_v[a:b], lr_t, b1, b2, 1e-8, 0.0, rescale)
        else:
            mom, wd = float(cfg.TRAIN.momentum), float(cfg.TRAIN.wd)
            for a, b, decay in seg:
                ops.sgd_momentum(self.flat_w[a:b], self.flat_g[a:b], self.flat_m[a:b], lr, mom, wd if decay else 0.0, 1.0)
        self.repack(forward=True)

    def world_size(self):
        import torch.distributed as dist

        return dist.get_world_size(self.pg) if (dist.is_available() and dist.is_initialized()) else 1

    # optimizer state checkpoint (the reference pickles MXNet updater states: module_checkpoint(save_optimizer_states=True),
    # train.py:314-316 -- that pickle needs mxnet to read, so the state travels as a plain .npz here)
    def save_optimizer_states(self, fname):
        st = {"num_update": np.int64(self.num_update)}
        for n in self.names:
            if n in FROZEN:
                continue
            st["mom:" + n] = self.m[n].cpu().numpy()
        if self.flat_v is not None:
            for n in self.names:
                off, sz = self.offset_of[n], self.w[n].numel()
                if n not in FROZEN:
                    st["var:" + n] = self.flat_v[off:off + sz].view(self.shapes[n]).cpu().numpy()
        np.savez(fname, **st)

    def load_optimizer_states(self, fname):
        st = np.load(fname)
        self.num_update = int(st["num_update"])
        for n in self.names:
            off, sz = self.offset_of[n], self.w[n].numel()
            if "mom:" + n in st:
                self.m[n].copy_(torch.as_tensor(st["mom:" + n]))
            if "var:" + n in st:
                if self.flat_v is None:
                    self.flat_v = torch.zeros_like(self.flat_m)
                self.flat_v[off:off + sz].view(self.shapes[n]).copy_(torch.as_tensor(st["var:" + n]))

    def get_params(self):
        return {n: self.w[n].cpu().numpy() for n in self.names}

    def get_grads(self):
        return {n: self.g[n].cpu().numpy() for n in self.names}


def fit_batch(module, data_batch, batch_updater, lr):
    """The inner TRAIN_ITER_SIZE loop of MutableModule.fit (reference module.py:1205-1213): every iteration is a separate
    optimizer step on refreshed inputs.  Returns the per-iteration outputs (what get_outputs feeds update_metric).
    lr: a float, or an LRScheduler -- then it is asked before EVERY update with that update's number, as mx.optimizer does
    (a step boundary inside the inner loop takes effect at once, not at the next batch)."""
    cfg = module.cfg
    n_iter = int(cfg.network.TRAIN_ITER_SIZE) if cfg.network.TRAIN_ITER else 1
    outs = []
    for iter_idx in range(n_iter):
        preds = module.forward_backward(data_batch)
        outs.append({"rot_est_norm": preds["rot_est_norm"].clone(), "trans_est": preds["trans_est"].clone(),
                     "loss_sums": module.loss_sums.clone(),
                     # what deepim/core/metric.py reads (Flow_L2Loss, PointMatchingLoss, MaskLoss)
                     "flow_loss_sum": module.loss_sums[0].clone(), "point_matching_loss_sum": module.loss_sums[1].clone(),
                     "rot_loss_sum": module.loss_sums[3].clone(), "trans_loss_sum": module.loss_sums[4].clone(),
                     "mask_prob": module.mask_prob, "mask_gt": module.zoom_mask_gt})
        module.update(lr(module.num_update + 1) if callable(lr) else lr)
        if iter_idx != n_iter - 1:
            data_batch = batch_updater.forward(data_batch, preds, cfg)
    return outs


def fit_epochs(module, batches, batch_updater, lr, epochs, on_epoch=None):
    """`epochs` passes of fit_batch over a FIXED list of device-resident training batches (the epoch loop of MutableModule.fit,
    reference module.py:1170-1260, without a data iterator).  fit_batch rewrites a batch in place (poses, rendered image, labels
    advance through the inner iterations), so every pass works on a private copy of the pristine blobs.
    -> history: array (epochs, n_inner_iterations, 5) of the loss sums (flow, point matching, -, rot, trans) over all pairs of an epoch.
    on_epoch(epoch_index, sums) -> truthy stops early."""
    hist = []
    for ep in range(int(epochs)):
        sums = None
        for b in batches:
            work = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()}
            outs = fit_batch(module, work, batch_updater, lr)
            s = torch.stack([o["loss_sums"] for o in outs])
            sums = s if sums is None else sums + s
        hist.append(sums.cpu().numpy().astype(np.float64))
        if not np.isfinite(hist[-1]).all():
            raise FloatingPointError("training diverged in epoch {}: loss sums {}".format(ep, hist[-1]))
        if on_epoch is not None and on_epoch(ep, hist[-1]):
            break
    return np.array(hist)
