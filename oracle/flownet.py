"""DeepIM FlowNetSimple graph, torch-CPU restatement (oracle; test-only).

Restates /root/reference/deepim/symbols/deepIM_flownet.py:
  get_convs :32-301, test heads get_test_symbol_share :840-971, train heads
  get_loss :303-560.  torch.nn.functional conv2d / conv_transpose2d stand in
for mx.sym.Convolution / Deconvolution (MXNet is absent: "parity unpinned" for
those primitives; mapping documented in SURVEY.md 8(a)).

Parameters are a dict name -> numpy array with MXNet names and layouts:
  conv  *_weight (Cout,Cin,kh,kw)   deconv *_weight (Cin,Cout/g,kh,kw)
  fc    *_weight (out,in) with `in` flattened in (c,h,w) order.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import zoom as ozoom

ENCODER = [  # name, cout, k, s, p   (deepIM_flownet.py:67-191)
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


def param_shapes(cin=8, pred_flow=True, pred_mask=True):
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
    shp["rot_weight"] = (4, 256)
    shp["rot_bias"] = (4,)
    shp["trans_weight"] = (3, 256)
    shp["trans_bias"] = (3,)
    if pred_flow or pred_mask:
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
    if pred_flow:
        shp["Convolution3_weight"] = (2, 770, 3, 3)
        shp["Convolution3_bias"] = (2,)
        shp["upsampling_weight"] = (2, 1, 32, 32)
    if pred_mask:
        shp["mask_conv3_weight"] = (1, 770, 3, 3)
        shp["mask_conv3_bias"] = (1,)
        shp["mask_upsampling_weight"] = (1, 1, 32, 32)
    return shp


def bilinear_kernel(shape):
    """mx.init.Initializer._init_bilinear (used at deepIM_flownet.py:1077-1099)."""
    w = np.zeros(int(np.prod(shape)), dtype=np.float32)
    f = np.ceil(shape[3] / 2.0)
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    for i in range(w.size):
        x = i % shape[3]
        y = (i // shape[3]) % shape[2]
        w[i] = (1 - abs(x / f - c)) * (1 - abs(y / f - c))
    return w.reshape(shape)


def _t(params, name, dtype):
    return torch.from_numpy(np.ascontiguousarray(params[name])).to(dtype)


def _lrelu(x):
    return F.leaky_relu(x, 0.1)


def crop_like(x, ref_hw, offset):
    """mx.sym.Crop(a, b, offset=(oy,ox)) = a[:, :, oy:oy+Hb, ox:ox+Wb]."""
    oy, ox = offset
    return x[:, :, oy : oy + ref_hw[0], ox : ox + ref_hw[1]]


def encoder(params, data, dtype=torch.float32, return_all=False):
    """get_convs :67-208. data (B,Cin,480,640) torch. Returns relu_fc7 (B,256) [+ features]."""
    feats = {}
    x = data.to(dtype)
    for name, cout, k, s, p in ENCODER:
        x = _lrelu(F.conv2d(x, _t(params, name + "_weight", dtype), _t(params, name + "_bias", dtype), stride=s, padding=p))
        feats[name] = x
    flat = x.reshape(x.shape[0], -1)  # mx Flatten: (c,h,w) order
    fc6 = _lrelu(F.linear(flat, _t(params, "fc6_weight", dtype), _t(params, "fc6_bias", dtype)))
    fc7 = _lrelu(F.linear(fc6, _t(params, "fc7_weight", dtype), _t(params, "fc7_bias", dtype)))
    feats["fc6"] = fc6
    feats["fc7"] = fc7
    return (fc7, feats) if return_all else fc7


def decoder(params, feats, dtype=torch.float32):
    """get_convs :213-299 -> Concat3 (B,770,30,40)."""
    r10, r8, r6 = feats["conv6_1"], feats["conv5_1"], feats["conv4_1"]
    c1 = F.conv2d(r10, _t(params, "Convolution1_weight", dtype), _t(params, "Convolution1_bias", dtype), padding=1)
    d5 = F.conv_transpose2d(r10, _t(params, "deconv5_weight", dtype), _t(params, "deconv5_bias", dtype), stride=2)
    d5 = _lrelu(crop_like(d5, r8.shape[2:], (1, 1)))
    u65 = F.conv_transpose2d(c1, _t(params, "upsample_flow6to5_weight", dtype), _t(params, "upsample_flow6to5_bias", dtype), stride=2)
    u65 = crop_like(u65, r8.shape[2:], (1, 1))
    cat2 = torch.cat([r8, d5, u65], dim=1)
    c2 = F.conv2d(cat2, _t(params, "Convolution2_weight", dtype), _t(params, "Convolution2_bias", dtype), padding=1)
    d4 = F.conv_transpose2d(cat2, _t(params, "deconv4_weight", dtype), _t(params, "deconv4_bias", dtype), stride=2)
    d4 = _lrelu(crop_like(d4, r6.shape[2:], (1, 1)))
    u54 = F.conv_transpose2d(c2, _t(params, "upsample_flow5to4_weight", dtype), _t(params, "upsample_flow5to4_bias", dtype), stride=2)
    u54 = crop_like(u54, r6.shape[2:], (1, 1))
    return torch.cat([r6, d4, u54], dim=1)


def flow_head(params, cat3, dtype=torch.float32):
    """Convolution3 + frozen 32x32/s16 group-2 bilinear deconv + Crop(8,8). :317-340 / :913-937."""
    f = F.conv2d(cat3, _t(params, "Convolution3_weight", dtype), _t(params, "Convolution3_bias", dtype), padding=1)
    up = F.conv_transpose2d(f, _t(params, "upsampling_weight", dtype), None, stride=16, groups=2)
    return crop_like(up, (480, 640), (8, 8))


def mask_head(params, cat3, dtype=torch.float32):
    """mask_conv3 + 32x32/s16 deconv + Crop(8,8) (pre-sigmoid logits). :504-529 / :845-868."""
    m = F.conv2d(cat3, _t(params, "mask_conv3_weight", dtype), _t(params, "mask_conv3_bias", dtype), padding=1)
    up = F.conv_transpose2d(m, _t(params, "mask_upsampling_weight", dtype), None, stride=16)
    return crop_like(up, (480, 640), (8, 8))


def network_input(zoom_io, zoom_ir, zoom_mo=None, zoom_mr=None, zoom_do=None, zoom_dr=None):
    """Concat(img/255, img/255[, depth/255, depth/255][, masks]) :33-66."""
    parts = [np.asarray(zoom_io, np.float32) / np.float32(255.0), np.asarray(zoom_ir, np.float32) / np.float32(255.0)]
    if zoom_do is not None:
        parts += [np.asarray(zoom_do, np.float32) / np.float32(255.0), np.asarray(zoom_dr, np.float32) / np.float32(255.0)]
    if zoom_mo is not None:
        parts += [zoom_mo, zoom_mr]
    return np.concatenate(parts, axis=1).astype(np.float32)


def forward_test(params, batch, K, pixel_means, fast_test=True, input_mask=True, pred_mask=True, pred_flow=True,
                 normalize_flow=20.0, dtype=torch.float32, input_depth=False):
    """get_test_symbol_share :764-980. batch: dict of numpy blobs (image_observed, image_rendered,
    src_pose, mask_observed, mask_rendered). Returns dict(se3, zoom_factor[, mask_observed_pred,
    zoom_mask_prob, flow_est_crop])."""
    H, W = 480, 640
    # :783-838: the zoom window comes from the masks when INPUT_MASK, else from the images; get_convs (:33-66) concatenates the masks
    # only when INPUT_MASK and PRED_MASK, and the zoomed depths (ZoomDepth) when INPUT_DEPTH
    zmo = zmr = zdo = zdr = None
    if input_mask:
        zmo, _, zmr, zf = ozoom.zoom_mask(batch["mask_observed"], batch["mask_observed"], batch["mask_rendered"], batch["src_pose"], K, H, W)
        zio, zir = ozoom.zoom_image_with_factor(zf, batch["image_observed"], batch["image_rendered"], pixel_means, H, W)
    else:
        zio, zir, zf = ozoom.zoom_image(batch["image_observed"], batch["image_rendered"], batch["src_pose"], K, pixel_means, H, W)
    if input_depth:
        zdo, zdr = ozoom.zoom_depth(zf, batch["depth_observed"], batch["depth_rendered"], H, W)
    if not (input_mask and pred_mask):
        zmo = zmr = None
    data = network_input(zio, zir, zmo, zmr, zdo, zdr)
    out = {"zoom_factor": zf, "data": data}
    with torch.no_grad():
        fc7, feats = encoder(params, torch.from_numpy(data), dtype, return_all=True)
        rot = F.linear(fc7, _t(params, "rot_weight", dtype), _t(params, "rot_bias", dtype))
        tz = F.linear(fc7, _t(params, "trans_weight", dtype), _t(params, "trans_bias", dtype))
        trans = ozoom.zoom_trans(zf, tz.float().numpy(), b_inv_zoom=True)
        out["se3"] = np.concatenate([rot.float().numpy(), trans], axis=1).astype(np.float32)
        out["feats"] = feats
        if not fast_test and (pred_mask or pred_flow):
            cat3 = decoder(params, feats, dtype)
            out["concat3"] = cat3
            if pred_mask:
                prob = torch.sigmoid(mask_head(params, cat3, dtype)).float().numpy()
                out["zoom_mask_prob"] = prob
                unz = ozoom.zoom_mask_with_factor(zf, prob, b_inv_zoom=True, H=H, W=W)
                out["mask_observed_pred"] = ozoom.mx_round(unz)
            if pred_flow:
                fl = flow_head(params, cat3, dtype).float().numpy() * np.float32(normalize_flow)
                out["flow_est_crop"] = ozoom.zoom_flow(zf, fl, b_inv_zoom=True, H=H, W=W)
    return out
