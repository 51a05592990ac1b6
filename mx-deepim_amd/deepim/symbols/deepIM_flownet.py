"""DeepIM FlowNetSimple network on the HIP kernels.

Mirror of /root/reference/deepim/symbols/deepIM_flownet.py (class deepIM_flownet): the same layer
names, parameter names/shapes (MXNet layouts) and test-graph outputs (`se3`, `zoom_factor`), but the
"symbol" is executed directly: zoom (csrc/zoom.hip) -> 10 direct convolutions + fc6 (csrc/conv.hip,
f32 MFMA, NHWC) -> pose head -> se3.  Weights are packed once into the kernels' layout.

  get_convs            :32-301   -> FlowNetHip.encoder()
  get_test_symbol_share:764-980  -> FlowNetHip.forward_test()
  init_weights         :998-1124 -> deepIM_flownet.init_weights() (random init; the FlowNet
                                    checkpoint is an external download and is not available offline)
"""
import os

import numpy as np
import torch

from lib.hip import ops

# name, cout, kernel, stride, pad      (deepIM_flownet.py:67-191)
ENCODER = [
    ("flow_conv1", 64, 7, 2, 3),
    ("conv2", 128, 5, 2, 2),
    ("conv3", 256, 5, 2, 2),
    ("conv3_1", 256, 3, 1, 1),
    ("conv4", 512, 3, 2, 1),
    ("conv4_1", 512, 3, 1, 1),
    ("conv5", 512, 3, 2, 1),
    ("conv5_1", 512, 3, 1, 1),
    ("conv6", 1024, 3, 2, 1),
    ("conv6_1", 1024, 3, 1, 1),
]


# Winograd output tile edge per 3x3 / stride-1 layer (4 where not listed; measured at B = 16: F(4x4) wins on all four, also on the
# 8x10 map of conv6_1 despite 20 % tile padding: 0.108 vs 0.128 ms)
WINO_M_DEFAULT = {}
# workgroup tile of the wide-Cout bf16 layers: 4 = 128 x 128 on 8 waves (64 x 32 each), 1 = 128 x 128 on 4 waves (64 x 64 each)
BF16_BIG_TILE = int(os.environ.get("DIM_BF16_BIG_TILE", "4"))
# the stride-1 patch kernel (tile 9) for the 3x3 / stride-1 layers, their input gradients and the phases of the stride-2 input gradients
BF16_PATCH = os.environ.get("DIM_BF16_PATCH", "") != "0"
# ... and for the stride-2 forward layers (8 x 16 pixel blocks).  Measured at B = 16 against the gathered-tap kernel: conv2 (5x5) 247 vs
# 249 us, conv3 (5x5) 191 vs 223, conv4 (3x3) 95 vs 95 -- a wave of 64 pixels x 64 channels loads 4 KB of weight fragments per 8 MFMAs,
# which is the 64 B/clk a CU's vector L1 delivers; the stride-1 blocks (128 pixels per wave) need half of that.  Default: the 5x5 layers;
# DIM_BF16_PATCH=1: stride-1 layers only, =2: every stride-2 layer.
BF16_PATCH_S2_KERNELS = {"": (5,), "1": (), "2": (3, 5)}.get(os.environ.get("DIM_BF16_PATCH", ""), ())

def bf16_tile(cout):
    """workgroup tile of a bf16 layer with `cout` GEMM columns: 8 = 128 x 256 (only when DIM_BF16_BIG_TILE=8), 4 / 1 = 128 x 128, 3 = 64 x 64"""
    if BF16_BIG_TILE == 8:
        return 8 if cout % 256 == 0 else (4 if cout % 128 == 0 else 3)
    return BF16_BIG_TILE if cout % 128 == 0 else 3


def input_channels(cfg):
    """Concat arity of get_convs (reference :33-66): 6 RGB (+2 depth) (+2 masks iff INPUT_MASK and PRED_MASK)."""
    c = 6
    if cfg.network.INPUT_DEPTH:
        c += 2
    if cfg.network.INPUT_MASK and cfg.network.PRED_MASK:
        c += 2
    return c


class deepIM_flownet(object):
    def __init__(self):
        self.eps = 1e-5
        self.workspace = 4096
        self.arg_shape_dict = {}
        self.sym = None

    def infer_param_shapes(self, cfg):
        cin = input_channels(cfg)
        shp = {}
        c = cin
        for name, cout, k, s, p in ENCODER:
            shp[name + "_weight"] = (cout, c, k, k)
            shp[name + "_bias"] = (cout,)
            c = cout
        shp["fc6_weight"] = (256, 1024 * 8 * 10)
        shp["fc6_bias"] = (256,)
        shp["fc7_weight"] = (256, 256)
        shp["fc7_bias"] = (256,)
        shp["rot_weight"] = (4 if cfg.network.ROT_TYPE == "QUAT" else 3, 256)
        shp["rot_bias"] = (shp["rot_weight"][0],)
        shp["trans_weight"] = (3, 256)
        shp["trans_bias"] = (3,)
        if cfg.network.PRED_FLOW or cfg.network.PRED_MASK:
            shp["Convolution1_weight"] = (2, 1024, 3, 3)
            shp["Convolution1_bias"] = (2,)
            shp["deconv5_weight"] = (1024, 512, 4, 4)
            shp["deconv5_bias"] = (512,)
            shp["upsample_flow6to5_weight"] = (2, 2, 4, 4)
            shp["upsample_flow6to5_bias"] = (2,)
            shp["Convolution2_weight"] = (2, 1026, 3, 3)
            shp["Convolution2_bias"] = (2,)
            shp["deconv4_weight"] = (1026, 256, 4, 4)
            shp["deconv4_bias"] = (256,)
            shp["upsample_flow5to4_weight"] = (2, 2, 4, 4)
            shp["upsample_flow5to4_bias"] = (2,)
        if cfg.network.PRED_FLOW:
            shp["Convolution3_weight"] = (2, 770, 3, 3)
            shp["Convolution3_bias"] = (2,)
            shp["upsampling_weight"] = (2, 1, 32, 32)
        if cfg.network.PRED_MASK:
            shp["mask_conv3_weight"] = (1, 770, 3, 3)
            shp["mask_conv3_bias"] = (1,)
            shp["mask_upsampling_weight"] = (1, 1, 32, 32)
        self.arg_shape_dict = shp
        return shp

    def get_symbol(self, cfg, is_train=True):
        """The reference returns an mx Symbol; here the 'symbol' is the shape table + the executor class."""
        self.infer_param_shapes(cfg)
        self.sym = ("train" if is_train else "test", cfg)
        return self.sym

    @staticmethod
    def bilinear_kernel(shape):
        """mx.init.Initializer._init_bilinear (reference :1077-1099)."""
        w = np.zeros(int(np.prod(shape)), dtype=np.float32)
        f = np.ceil(shape[3] / 2.0)
        c = (2 * f - 1 - f % 2) / (2.0 * f)
        for i in range(w.size):
            x = i % shape[3]
            y = (i // shape[3]) % shape[2]
            w[i] = (1 - abs(x / f - c)) * (1 - abs(y / f - c))
        return w.reshape(shape)

    def init_weights(self, cfg, arg_params, aux_params, seed=0):
        """Fill every missing parameter.  Encoder/decoder convs: seeded He-normal stand-in for the FlowNet
        checkpoint (with the mask channels of flow_conv1 zero-padded as :1009-1025 does); heads as
        :1033-1099 (xavier fc6/fc7, rot row 0 ~U(.01,1.01), rest ~U(0,.01), trans zeros, bilinear upsampling)."""
        if not self.arg_shape_dict:
            self.infer_param_shapes(cfg)
        rng = np.random.RandomState(seed)
        # a FlowNet / RGB-only checkpoint has a 6-channel first layer: the extra (mask / depth) input channels start at zero (:1009-1025)
        w1 = arg_params.get("flow_conv1_weight")
        need = self.arg_shape_dict["flow_conv1_weight"][1]
        if w1 is not None and w1.shape[1] < need:
            w1 = np.asarray(w1, dtype=np.float32)
            pad = np.zeros((w1.shape[0], need - w1.shape[1]) + w1.shape[2:], dtype=np.float32)
            arg_params["flow_conv1_weight"] = np.concatenate([w1, pad], axis=1)
        # init_from_flownet: the pose / mask heads and the frozen upsamplers are re-initialised even if the file has them (:1031-1093)
        if arg_params and getattr(cfg.network, "init_from_flownet", False):
            for k in ("fc6_bias", "fc6_weight", "fc7_bias", "fc7_weight", "rot_bias", "rot_weight", "trans_bias", "trans_weight",
                      "upsampling_weight", "mask_conv3_bias", "mask_conv3_weight", "mask_upsampling_weight"):
                arg_params.pop(k, None)
        for k, shp in self.arg_shape_dict.items():
            if k in arg_params:
                continue
            if k.endswith("bias"):
                arg_params[k] = np.zeros(shp, dtype=np.float32) if k.startswith(("fc", "rot", "trans", "mask_conv3")) else \
                    rng.normal(0, 0.01, size=shp).astype(np.float32)
            elif k in ("upsampling_weight", "mask_upsampling_weight"):
                arg_params[k] = self.bilinear_kernel(shp)
            elif k == "rot_weight":
                w = rng.rand(*shp) * 0.01
                w[0, :] = rng.rand(shp[1]) + 0.01
                arg_params[k] = w.astype(np.float32)
            elif k == "trans_weight":
                arg_params[k] = np.zeros(shp, dtype=np.float32)
            elif k in ("fc6_weight", "fc7_weight"):
                scale = np.sqrt(3.0 / ((shp[0] + shp[1]) / 2.0))  # mx.init.Xavier(uniform, avg, 3)
                arg_params[k] = rng.uniform(-scale, scale, size=shp).astype(np.float32)
            elif k == "mask_conv3_weight":
                arg_params[k] = rng.normal(0, 0.01, size=shp).astype(np.float32)
            else:
                fan_in = int(np.prod(shp[1:])) if not k.startswith(("deconv", "upsample_flow")) else shp[0] * shp[2] * shp[3] // 4
                w = rng.normal(0, np.sqrt(2.0 / (1.01 * fan_in)), size=shp).astype(np.float32)
                if k == "flow_conv1_weight" and shp[1] > 6:
                    w[:, 6:] = 0.0
                arg_params[k] = w
        return arg_params


class FlowNetHip(object):
    """Device executor of the test graph for a fixed per-GPU batch size B (all buffers pre-allocated)."""

    H, W = 480, 640

    def __init__(self, cfg, arg_params, batch_size, device="cuda:0", conv_plan=None, winograd=True, wino_m=None, wino_tile=None,
                 bf16=False, wino_s2=True):
        """winograd: run the 3x3 / stride-1 layers (conv3_1, conv4_1, conv5_1, conv6_1) through Winograd F(4x4,3x3) / F(2x2,3x3)
        (same f32 result within 1e-4 relative, 4x / 2.25x fewer multiply-adds) and the 5x5 / stride-2 layers (conv2, conv3) through
        their four phase images and F(4x4,3x3) (2.78x fewer).  False = direct kernel for every layer.
        wino_s2: with winograd, also the 3x3 / stride-2 layers whose output map has >= 300 pixels (conv4, conv5) through their phase
        images and minimal filtering (81 plane GEMMs per 4 x 4 tile against 144 multiplies direct; csrc/wino_s2.hip).
        wino_m / wino_tile: optional {layer: output tile edge 2|4} / {layer: GEMM workgroup tile 3|4} overrides.
        bf16: the convolutions and the two large deconvolutions run on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, f32 accumulate;
        activations stay fp32 in HBM, weights are kept as bf16 copies of the packed arrays).  Direct form for every layer -- with the
        matrix pipe 16x faster the Winograd transforms would cost more than they save.  Training mode of BASELINE configs[2]; the
        inference headline stays fp32."""
        self.bf16 = bool(bf16)
        if self.bf16:
            winograd = False
        self.cfg = cfg
        self.B = batch_size
        self.device = torch.device(device)
        self.cin = input_channels(cfg)
        # first-layer input arities of get_convs (reference :33-66).  The device tensor X always has 8 NHWC channels:
        #   mode 0  images + masks            (INPUT_MASK and PRED_MASK; the shipped graph)
        #   mode 1  images only, 2 zero lanes (no masks in the Concat: Cin = 6; the weights of the two spare lanes are zero)
        #   mode 2  images + two depth planes (INPUT_DEPTH without masks: Cin = 8)
        #   mode 3  images + depth planes + masks (INPUT_DEPTH with masks: Cin = 10) as TWO 8-lane groups -- X as in mode 2 and X2 =
        #           [mask_obs, mask_ren, 0 x 6] -- and flow_conv1 as the sum of two 8-channel convolutions (weights [:, :8] and [:, 8:10]
        #           zero-padded) followed by the LeakyReLU; three launches instead of one, no shipped configuration uses it
        with_masks = bool(cfg.network.INPUT_MASK and cfg.network.PRED_MASK)
        self.input_mode = (3 if cfg.network.INPUT_DEPTH else 0) if with_masks else (2 if cfg.network.INPUT_DEPTH else 1)
        self.zoom_from_masks = bool(cfg.network.INPUT_MASK)   # ZoomMask vs ZoomImage for the zoom window (:783-819)
        d = self.device
        self.params = {k: torch.as_tensor(np.ascontiguousarray(v), dtype=torch.float32).to(d) for k, v in arg_params.items()}
        if self.cin == 6:  # zero weights for the two spare input lanes
            w1 = self.params["flow_conv1_weight"]
            self.params["flow_conv1_weight"] = torch.cat([w1, torch.zeros((w1.shape[0], 2) + tuple(w1.shape[2:]), device=d)], dim=1).contiguous()
        self.packed = {}
        if self.input_mode == 3:  # 10 input channels: [:, :8] for the images + depth group, [:, 8:10] (+ 6 zero lanes) for the mask group
            w1 = self.params["flow_conv1_weight"]
            assert w1.shape[1] == 10, w1.shape
            self.packed["flow_conv1_masks"] = self.pack_conv(
                torch.cat([w1[:, 8:], torch.zeros((w1.shape[0], 6) + tuple(w1.shape[2:]), device=d)], dim=1).contiguous())
            self.params["flow_conv1_weight"] = w1[:, :8].contiguous()
        for name, cout, k, s, p in ENCODER:
            self.packed[name] = self.pack_conv(self.params[name + "_weight"])
        self.packed["fc6"] = ops.fc_pack_weight(self.params["fc6_weight"], 1024, 8, 10)
        self.wino, self.wino_m, self.wino5, self.wino3s2 = {}, {}, {}, {}
        if winograd:
            hh, ww, cc = self.H, self.W, 8
            for name, cout, k, s, p in ENCODER:
                # conv4, conv5: phase images + minimal filtering (81 plane GEMMs); the rule lives in C (dim_winograd3x3s2_use), shared with
                # dim_refiner_create.  Inference only: the training executor keeps its forward on the direct kernel (wino_s2=False).
                if wino_s2 and k == 3 and s == 2 and p == 1 and ops.lib().dim_winograd3x3s2_use(hh, ww, cc, cout):
                    self.wino3s2[name] = ops.winograd3x3s2_pack_weight(self.params[name + "_weight"])
                hh, ww = ops.conv_out_hw(hh, ww, k, k, s, p)
                cc = cout
            for name, cout, k, s, p in ENCODER:
                if k == 5 and s == 2 and p == 2:  # conv2, conv3: four phase images through F(4x4,3x3), 36 GEMMs with K = 4 Cin
                    self.wino5[name] = ops.winograd5x5s2_pack_weight(self.params[name + "_weight"])
                if k == 3 and s == 1 and p == 1:
                    # output tile edge: F(4x4,3x3) (4x fewer multiply-adds) by default
                    self.wino_m[name] = int((wino_m or {}).get(name, WINO_M_DEFAULT.get(name, 4)))
                    self.wino[name] = ops.winograd_pack_weight(self.params[name + "_weight"], m=self.wino_m[name])
        self.K = np.asarray(cfg.dataset.INTRINSIC_MATRIX, dtype=np.float32).reshape(3, 3)
        self.plane_means = np.asarray(cfg.network.PIXEL_MEANS, dtype=np.float32).reshape(3)[::-1].copy()
        # tile / split-K plan per layer: (tile, splits); 0 = library heuristic
        self.conv_plan = {"fc6": (3, 40)}
        if self.bf16:
            # operand-traffic bound: 128 x 128 tiles (8 waves) wherever Cout allows, no split-K except on the 8 x 10 / 15 x 20 maps
            c = 8
            h, w = self.H, self.W
            for name, cout, k, s, p in ENCODER:
                h, w = ops.conv_out_hw(h, w, k, k, s, p)
                tiles = -(-batch_size * h * w // 128) * (cout // 128) if cout % 128 == 0 else 0
                # split-K on the small maps: the 8-wave 128 x 128 kernel keeps 2 workgroups per CU = 512 resident slots, and a grid just
                # over that pays a second, nearly empty round (the round-2 rule, ceil(512 / tiles) capped at 4, gave conv5 / conv5_1
                # 608 workgroups).  tools/fwd_bf16_sweep.py at B = 16, splits: us incl. the slab sum -- conv5 3: 54, 4: 62; conv5_1 3: 52,
                # 4: 59; conv6 3: 32, 4: 39, 6: 36; conv6_1 3: 50, 4: 61, 6: 49
                self.conv_plan[name] = (bf16_tile(cout) if c != 8 else 3, 1 if (tiles == 0 or tiles >= 256) else max(1, min(3, 512 // tiles)))
                # 3x3 / stride-1 layers on large maps: the LDS-halo kernel (conv.hip conv_bf16_halo_kernel, tile 7).  Measured at B = 16
                # against the gathered-tap kernel: conv3_1 0.150 vs 0.161 ms, conv4_1 0.175 vs 0.177; the stride-2 layers lose (conv2
                # 0.366 vs 0.256, conv3 0.286 vs 0.247, conv4 0.143 vs 0.102: their 45-53 KB patches + 41 KB of weight buffers leave
                # one 4-wave workgroup per CU), so they stay where they were.  DIM_BF16_HALO=0: gathered-tap kernel everywhere; =2: every
                # eligible layer on tile 7 (the A/B above)
                halo = os.environ.get("DIM_BF16_HALO", "1")
                if c % 32 == 0 and cout % 128 == 0 and h * w >= 1200 and ((halo == "1" and k == 3 and s == 1) or (halo == "2" and k in (3, 5))):
                    self.conv_plan[name] = (7, 1)
                if BF16_PATCH and c % 32 == 0 and cout % 128 == 0 and h * w >= 1200 and (
                        (k == 3 and s == 1) or (s == 2 and k in BF16_PATCH_S2_KERNELS)):
                    self.conv_plan[name] = (9, 1)   # patch kernel (conv.hip conv_bf16_patch_kernel): 3x3 / stride 1, 3x3 and 5x5 / stride 2
                c = cout
        if conv_plan:
            self.conv_plan.update(conv_plan)
        B, H, W = batch_size, self.H, self.W
        self.X = torch.empty((B, H, W, 8), dtype=torch.float32, device=d)
        self.X2 = torch.empty((B, H, W, 8), dtype=torch.float32, device=d) if self.input_mode == 3 else None   # the mask lanes of the 10-channel input
        self.acts = {}
        h, w, c = H, W, 8
        max_ws = 0
        self.layer_info = {}  # name -> dict(M, K, N, flops, tile, splits) for profiling / roofline accounting
        for name, cout, k, s, p in ENCODER:
            ho, wo = ops.conv_out_hw(h, w, k, k, s, p)
            self.acts[name] = torch.empty((B, ho, wo, cout), dtype=torch.float32, device=d)
            nchunks = -(-k * k // 4) if c == 8 else k * k * (c // 32)
            tile, splits = self.conv_plan.get(name, ops.conv_auto_plan(B * ho * wo, cout, nchunks, cin=c))
            if name == "flow_conv1" and name not in (conv_plan or {}) and os.environ.get("DIM_CONV1_HALO", "1") != "0":
                # LDS-halo first-layer kernel (conv.hip conv1_halo_kernel, on the bf16 pipe conv1_halo_bf16_kernel: 0.26 -> ~0.12 ms at
                # B = 16, the layer's HBM bytes once); DIM_CONV1_HALO=0: the gathered-tap kernel
                tile, splits = 6, 1
            self.conv_plan[name] = (tile, splits)
            self.layer_info[name] = dict(M=B * ho * wo, K=c * k * k, N=cout, flops=2 * B * ho * wo * cout * c * k * k, tile=tile,
                                         splits=splits, cin=c, min_bytes=4 * (B * h * w * c + cout * c * k * k + B * ho * wo * cout))
            if splits != 1:
                max_ws = max(max_ws, ops.lib().dim_conv2d_workspace_floats(B, h, w, c, cout, k, k, s, p, splits))
            if name in self.wino:
                # GEMM rows = tiles; 128x128 workgroup tiles once there are enough of them, 64x64 for the small maps
                m = self.wino_m[name]
                tiles = B * (-(-h // m)) * (-(-w // m))
                wt = (wino_tile or {}).get(name, self._wino_tile(cout, tiles))
                self.layer_info[name].update(self._wino_info(m, 1, wt, tiles, c, cout, B * h * w * c, B * ho * wo * cout))
                max_ws = max(max_ws, ops.lib().dim_winograd_workspace_floats(B, h, w, c, cout, m))
            if name in self.wino5:
                tiles = B * (-(-ho // 4)) * (-(-wo // 4))
                wt = (wino_tile or {}).get(name, self._wino_tile(cout, tiles))
                self.layer_info[name].update(self._wino_info(4, 2, wt, tiles, c, cout, B * h * w * c, B * ho * wo * cout))
                max_ws = max(max_ws, ops.lib().dim_winograd5x5s2_workspace_floats(B, h, w, c, cout))
            if name in self.wino3s2:
                tiles = B * (-(-ho // 4)) * (-(-wo // 4))
                wt = (wino_tile or {}).get(name, self._wino_tile(cout, tiles, 81))
                info = self._wino_info(4, 1, wt, tiles, c, cout, B * h * w * c, B * ho * wo * cout, planes=81)
                info.update(wino_in_kernel="dim::wino_s2_input_kernel", wino_out_kernel="dim::wino_s2_output_kernel")
                self.layer_info[name].update(info)
                max_ws = max(max_ws, ops.lib().dim_winograd3x3s2_workspace_floats(B, h, w, c, cout))
            h, w, c = ho, wo, cout
        assert (h, w, c) == (8, 10, 1024)
        tile, splits = self.conv_plan["fc6"]
        self.layer_info["fc6"] = dict(M=B, K=81920, N=256, flops=2 * B * 81920 * 256, tile=tile, splits=splits, cin=1024,
                                      min_bytes=4 * (B * 81920 + 256 * 81920 + B * 256))
        max_ws = max(max_ws, ops.lib().dim_conv2d_workspace_floats(B, 8, 10, 1024, 256, 8, 10, 1, 0, splits),
                     ops.lib().dim_fc_fwd_workspace_floats(1024, 8, 10, 256))
        self.workspace = torch.empty((max(max_ws, 4),), dtype=torch.float32, device=d)
        # ---- decoder + flow / mask heads (only in the graph when not FAST_TEST, reference :840-954)
        self.has_decoder = "deconv5_weight" in self.params
        if self.has_decoder:
            self.packed["deconv5"] = self.pack_deconv(self.params["deconv5_weight"])
            self.packed["deconv4"] = self.pack_deconv(self.params["deconv4_weight"])
            for n in ("Convolution1", "Convolution2", "Convolution3", "mask_conv3"):
                if n + "_weight" in self.params:
                    self.packed[n] = ops.conv_small_cout_pack_weight(self.params[n + "_weight"])
            self.flow6 = torch.empty((B, 8, 10, 2), dtype=torch.float32, device=d)          # Convolution1
            self.concat2 = torch.zeros((B, 15, 20, ops.pad64(1026)), dtype=torch.float32, device=d)  # [ReLU8 | ReLU11 | up6to5 | 0-pad]
            self.flow5 = torch.empty((B, 15, 20, 2), dtype=torch.float32, device=d)         # Convolution2
            self.concat3 = torch.zeros((B, 30, 40, ops.pad64(770)), dtype=torch.float32, device=d)   # [ReLU6 | ReLU12 | up5to4 | 0-pad]
            self.flow4 = torch.empty((B, 30, 40, 2), dtype=torch.float32, device=d)         # Convolution3
            self.mask4 = torch.empty((B, 30, 40, 1), dtype=torch.float32, device=d)         # mask_conv3
            self.zoom_flow = torch.empty((B, 2, H, W), dtype=torch.float32, device=d)
            self.flow_est = torch.empty((B, 2, H, W), dtype=torch.float32, device=d)
            self.zoom_mask_prob = torch.empty((B, 1, H, W), dtype=torch.float32, device=d)
            self.mask_pred = torch.empty((B, 1, H, W), dtype=torch.float32, device=d)
        self.fc6 = torch.empty((B, 1, 1, 256), dtype=torch.float32, device=d)
        self.fc7 = torch.empty((B, 256), dtype=torch.float32, device=d)
        self.se3 = torch.empty((B, 7), dtype=torch.float32, device=d)
        self.zoom_factor = torch.empty((B, 4), dtype=torch.float32, device=d)
        self.bbox_obs = torch.empty((B, 4), dtype=torch.int32, device=d)
        self.bbox_ren = torch.empty((B, 4), dtype=torch.int32, device=d)
        self.status = torch.zeros((B,), dtype=torch.int32, device=d)
        torch.cuda.synchronize(d)

    def pack_conv(self, w_oihw):
        """MXNet (Cout,Cin,kh,kw) -> the forward kernel's packed array (bf16 copy in bf16 mode)"""
        return ops.conv2d_pack_weight(w_oihw, as_bf16=self.bf16)

    def pack_deconv(self, w_iohw):
        return ops.deconv4x4s2_pack_weight(w_iohw, as_bf16=self.bf16)

    @staticmethod
    def _wino_tile(cout, tiles, planes=36):
        """workgroup tile of a layer's plane GEMMs: dim_winograd_gemm_tile_planes, the one copy of the rule (shared with csrc/refiner.hip)"""
        return int(ops.lib().dim_winograd_gemm_tile_planes(int(cout), int(tiles), int(planes)))

    @staticmethod
    def _wino_info(m, S, tile, tiles, cin, cout, x_floats, y_floats, planes=None):
        """accounting of one Winograd layer for bench.py: the batched GEMM (planes x [tiles x K] . [K x cout]) and the algorithmic
        bytes of the two transform kernels (read x + write V; read M + write y)"""
        planes, K = planes or (m + 2) ** 2, cin * S * S
        return dict(winograd=True, wino_m=m, wino_tile=tile, wino_planes=planes, wino_rows=tiles, wino_k=K,
                    wino_flops=2 * planes * tiles * K * cout, wino_gemm_bytes=4 * planes * (tiles * K + K * cout + tiles * cout),
                    wino_in_bytes=4 * (x_floats + planes * tiles * K), wino_out_bytes=4 * (planes * tiles * cout + y_floats),
                    wino_in_kernel="dim::wino_input_kernel" if m == 2 else "dim::wino4_input_kernel<2, %d>" % S,
                    wino_out_kernel="dim::wino_output_kernel" if m == 2 else "dim::wino4_output_kernel<2>")

    # ---- pieces -------------------------------------------------------------------------------
    def zoom(self, batch, bbox_ren=None, nchw_out=None, bbox_obs=None, src_pose=None, status=None):
        """ZoomMask + ZoomImageWithFactor + Concat (reference :783-806, :53-60).  At test time
        mask_gt_observed IS mask_observed (:779).  bbox_ren may come pre-computed from the rasteriser, bbox_obs from dim_box_mask
        (the refinement loop knows the rectangle it has just written); src_pose / status override the blob / the status buffer."""
        if self.zoom_from_masks:
            if bbox_obs is None:
                bbox_obs = ops.mask_bbox(batch["mask_observed"], 0.3, out=self.bbox_obs)
            if bbox_ren is None:
                bbox_ren = ops.mask_bbox(batch["mask_rendered"], 0.2, out=self.bbox_ren)
        else:
            # ZoomImage (zoom_image.py:31-37): the validity "mask" of an image is sum_c(image + mean) > 0.01
            bbox_obs = ops.mask_bbox(batch["image_observed"], 0.01, mode=1, means3=self.plane_means, out=self.bbox_obs)
            bbox_ren = ops.mask_bbox(batch["image_rendered"], 0.01, mode=1, means3=self.plane_means, out=self.bbox_ren)
        ops.zoom_factor(bbox_obs, bbox_ren, batch["src_pose"] if src_pose is None else src_pose, self.K, self.H, self.W,
                        out=self.zoom_factor, status=self.status if status is None else status)
        return self.net_input(batch, nchw_out=nchw_out)

    def net_input(self, batch, nchw_out=None):
        """the first layer's input from the blobs and self.zoom_factor: ZoomImageWithFactor (+ ZoomDepth) (+ the zoomed masks) in the
        Concat order of get_convs (reference :33-66), as one or two 8-lane NHWC tensors (see input_mode)"""
        if self.input_mode == 0:
            ops.zoom_net_input(batch["image_observed"], batch["image_rendered"], batch["mask_observed"], batch["mask_rendered"],
                               self.zoom_factor, self.plane_means, X=self.X, nchw_out=nchw_out)
        elif self.input_mode == 1:
            ops.zoom_net_input_ex(batch["image_observed"], batch["image_rendered"], None, None, self.zoom_factor, self.plane_means, 1, X=self.X)
        else:
            ops.zoom_net_input_ex(batch["image_observed"], batch["image_rendered"], batch["depth_observed"], batch["depth_rendered"],
                                  self.zoom_factor, self.plane_means, 2, X=self.X)
            if self.input_mode == 3:
                ops.zoom_net_input_ex(batch["image_observed"], batch["image_rendered"], batch["mask_observed"], batch["mask_rendered"],
                                      self.zoom_factor, self.plane_means, 3, X=self.X2)
        return self.X

    def encoder(self, X=None, events=None):
        """events: optional dict layer-name -> list of (tag, start, end) HIP-event triples (see ops.conv2d_fwd)."""
        x = self.X if X is None else X
        for name, cout, k, s, p in ENCODER:
            if name in self.wino:
                x = ops.conv2d_fwd_winograd(x, x.shape[-1], self.wino[name], self.params[name + "_bias"], cout, slope=0.1,
                                            tile=self.layer_info[name]["wino_tile"], out=self.acts[name], workspace=self.workspace,
                                            events=None if events is None else events.setdefault(name, []), m=self.wino_m[name])
                continue
            if name in self.wino5:
                x = ops.conv2d_fwd_winograd5x5s2(x, x.shape[-1], self.wino5[name], self.params[name + "_bias"], cout, slope=0.1,
                                                 tile=self.layer_info[name]["wino_tile"], out=self.acts[name], workspace=self.workspace,
                                                 events=None if events is None else events.setdefault(name, []))
                continue
            if name in self.wino3s2:
                x = ops.conv2d_fwd_winograd3x3s2(x, x.shape[-1], self.wino3s2[name], self.params[name + "_bias"], cout, slope=0.1,
                                                 tile=self.layer_info[name]["wino_tile"], out=self.acts[name], workspace=self.workspace,
                                                 events=None if events is None else events.setdefault(name, []))
                continue
            tile, splits = self.conv_plan[name]
            first10 = name == "flow_conv1" and self.input_mode == 3
            x = ops.conv2d_fwd(x, self.packed[name], self.params[name + "_bias"], cout, k, k, s, p, slope=1.0 if first10 else 0.1, splits=splits,
                               tile=tile, out=self.acts[name], workspace=self.workspace,
                               events=None if events is None else events.setdefault(name, []))
            if first10:   # + the mask group's convolution, then the activation (y *= y > 0 ? 1 : 0.1 is what dim_lrelu_bwd does with dy = y)
                ops.conv2d_fwd_ex(self.X2, 0, 8, self.packed["flow_conv1_masks"], None, x, 0, cout, k, k, s, p, slope=1.0, tile=3, accumulate=True)
                ops.lrelu_bwd(x, x, cout, slope=0.1)
        # fc6: a pure weight stream at these batch sizes (84 MB per forward) -> its own kernel instead of the 8x10 "convolution"
        ops.fc_fwd(x, self.packed["fc6"], self.params["fc6_bias"], 256, slope=0.1, out=self.fc6, workspace=self.workspace,
                   events=None if events is None else events.setdefault("fc6", []))
        return self.fc6

    def autotune(self, tiles=(3, 4), split_choices=(1, 2, 3, 4, 6, 8), reps=10):
        """Pick (tile, splits) per layer by timing the candidates on this GPU (HIP events on the launch stream).
        Only speed changes: split-K alters the f32 summation order (|delta| ~1e-6 relative), nothing else."""
        x = self.X
        layers = [(n, co, k, k, s, p, self.params[n + "_bias"], self.acts[n]) for n, co, k, s, p in ENCODER]
        for name, cout, kh, kw, s, p, bias, out in layers:
            N, H, W, C = x.shape
            if name in self.wino:  # Winograd layers have no (tile, splits) plan of the direct kernel: just produce their output
                x = ops.conv2d_fwd_winograd(x, C, self.wino[name], bias, cout, slope=0.1, tile=self.layer_info[name]["wino_tile"], out=out,
                                            workspace=self.workspace, m=self.wino_m[name])
                continue
            if name in self.wino5:
                x = ops.conv2d_fwd_winograd5x5s2(x, C, self.wino5[name], bias, cout, slope=0.1, tile=self.layer_info[name]["wino_tile"],
                                                 out=out, workspace=self.workspace)
                continue
            cands = [(t, sp) for t in tiles for sp in split_choices if not (t in (1, 4) and (cout % 128 or C == 8))]
            best = None
            for t, sp in cands:
                ws_need = ops.lib().dim_conv2d_workspace_floats(N, H, W, C, cout, kh, kw, s, p, sp)
                if ws_need > self.workspace.numel():
                    self.workspace = torch.empty((ws_need,), dtype=torch.float32, device=self.device)
                run = lambda: ops.conv2d_fwd(x, self.packed[name], bias, cout, kh, kw, s, p, slope=0.1, splits=sp, tile=t, out=out,  # noqa: E731
                                             workspace=self.workspace)
                run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    run()
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1) / reps
                if best is None or ms < best[0]:
                    best = (ms, t, sp)
            self.conv_plan[name] = (best[1], best[2])
            self.layer_info[name]["tile"], self.layer_info[name]["splits"] = best[1], best[2]
            ops.conv2d_fwd(x, self.packed[name], bias, cout, kh, kw, s, p, slope=0.1, splits=best[2], tile=best[1], out=out,
                           workspace=self.workspace)
            x = out
        return dict(self.conv_plan)

    def head(self, se3=None):
        return ops.pose_head_fwd(self.fc6.view(self.B, 256), self.params, self.zoom_factor, se3=self.se3 if se3 is None else se3,
                                 fc7_out=self.fc7)

    def decoder(self):
        """get_convs :213-299: Convolution1, deconv5 (+Crop, LeakyReLU), upsample_flow6to5, Concat2, Convolution2, deconv4,
        upsample_flow5to4, Concat3.  The concats are channel ranges of one NHWC buffer each (zero-padded to a multiple of 32)."""
        p = self.params
        r10, r8, r6 = self.acts["conv6_1"], self.acts["conv5_1"], self.acts["conv4_1"]
        ops.conv_small_cout_fwd(r10, 1024, self.packed["Convolution1"], p["Convolution1_bias"], 2, out=self.flow6)
        ops.copy_nhwc_channels(self.concat2, 0, r8, 0, 512)   # Concat2[..., :512] = ReLU8 (a kernel, not a memcpy node)
        ops.deconv4x4s2_fwd(r10, 1024, self.packed["deconv5"], p["deconv5_bias"], self.concat2, 512, crop=1, slope=0.1, out_coff=512,
                            tile=bf16_tile(512) if self.bf16 else 3)
        ops.deconv4x4s2_tiny_fwd(self.flow6, 2, p["upsample_flow6to5_weight"], p["upsample_flow6to5_bias"], self.concat2, 2, crop=1,
                                 out_coff=1024)
        ops.conv_small_cout_fwd(self.concat2, 1026, self.packed["Convolution2"], p["Convolution2_bias"], 2, out=self.flow5)
        ops.copy_nhwc_channels(self.concat3, 0, r6, 0, 512)   # Concat3[..., :512] = ReLU6
        ops.deconv4x4s2_fwd(self.concat2, 1026, self.packed["deconv4"], p["deconv4_bias"], self.concat3, 256, crop=1, slope=0.1,
                            out_coff=512, tile=bf16_tile(256) if self.bf16 else 3)
        ops.deconv4x4s2_tiny_fwd(self.flow5, 2, p["upsample_flow5to4_weight"], p["upsample_flow5to4_bias"], self.concat3, 2, crop=1,
                                 out_coff=768)
        return self.concat3

    def heads(self):
        """test-graph flow / mask heads (:845-954): conv -> frozen x16 bilinear deconvolution -> Crop(8,8) -> inverse zoom."""
        cfg, p, out = self.cfg, self.params, {}
        if cfg.network.PRED_MASK:
            ops.conv_small_cout_fwd(self.concat3, 770, self.packed["mask_conv3"], p["mask_conv3_bias"], 1, out=self.mask4)
            ops.upsample16_fwd(self.mask4, p["mask_upsampling_weight"], self.H, self.W, crop=8, sigmoid=True, out=self.zoom_mask_prob)
            # ZoomMaskWithFactor(b_inv_zoom) binarises at 0.2, samples, rounds; the following mx.sym.round is then a no-op
            ops.zoom_planes(self.zoom_mask_prob, self.zoom_factor, inverse=True, pre=1, post=1, out=self.mask_pred)
            out["mask_observed_pred_output"] = self.mask_pred
            out["zoom_mask_observed_prob_iter_output"] = self.zoom_mask_prob
        if cfg.network.PRED_FLOW:
            ops.conv_small_cout_fwd(self.concat3, 770, self.packed["Convolution3"], p["Convolution3_bias"], 2, out=self.flow4)
            ops.upsample16_fwd(self.flow4, p["upsampling_weight"], self.H, self.W, crop=8, scale=float(cfg.dataset.NORMALIZE_FLOW),
                               out=self.zoom_flow)
            ops.zoom_planes(self.zoom_flow, self.zoom_factor, inverse=True, scale_mode=2, out=self.flow_est)
            out["flow_est_crop_output"] = self.flow_est
        return out

    def forward_test(self, batch, bbox_ren=None, bbox_obs=None, src_pose=None, se3_out=None, status_out=None):
        """One test-graph forward.  FAST_TEST graph: zoom + encoder + FC heads; otherwise also decoder + flow / mask heads
        (reference :840-843, :913).  Returns the output dict the refinement loop reads (tester.py:483-491).
        The optional arguments let the refinement loop hand in what it already has on the device (boxes, the pose of the previous
        iteration) and receive se3 / status straight in its per-iteration buffers -- no copies between iterations."""
        self.zoom(batch, bbox_ren=bbox_ren, bbox_obs=bbox_obs, src_pose=src_pose, status=status_out)
        self.encoder()
        se3 = self.head(se3=se3_out)
        out = {"se3_output": se3, "zoom_factor": self.zoom_factor}
        cfg = self.cfg
        if self.has_decoder and not cfg.TEST.FAST_TEST and (cfg.network.PRED_MASK or cfg.network.PRED_FLOW):
            self.decoder()
            out.update(self.heads())
        return out

    def flops_per_forward(self):
        """algorithmic MACs*2 of encoder + head for this batch (SURVEY.md 8a layer table)."""
        h, w, c, total = self.H, self.W, 8, 0
        for name, cout, k, s, p in ENCODER:
            h, w = ops.conv_out_hw(h, w, k, k, s, p)
            total += h * w * cout * c * k * k
            c = cout
        total += 81920 * 256 + 256 * 256 + 256 * 7
        if self.has_decoder and not self.cfg.TEST.FAST_TEST and (self.cfg.network.PRED_MASK or self.cfg.network.PRED_FLOW):
            total += 1967511120   # decoder + flow / mask heads (SURVEY.md 8a layer table: 3.94 GFLOP per pair and forward)
        return 2 * total * self.B
