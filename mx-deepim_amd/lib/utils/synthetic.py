"""Synthetic LINEMOD-shaped inputs (no dataset offline).  numpy only, no rendering here.

Definition follows SURVEY.md 8(d): procedurally generated closed meshes (displaced icospheres,
diameter 0.1-0.3 m) with a seeded texture; GT pose uniform on SO(3), z ~ U(0.6,1.2) m with the
projected centre >= 16 px inside the image; initial pose = GT (+) noise with the reference's
training-pair distribution (toolkit/LM6d_1_gen_rendered_pose.py:59,98-117: Euler N(0,15 deg) per
axis rejected above 45 deg, dx,dy ~ N(0,0.01 m), dz ~ N(0,0.05 m)).
"""
import numpy as np

LINEMOD_K = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]], dtype=np.float32)  # config.py:63-65
PIXEL_MEANS = np.array([123.68, 116.779, 103.939], dtype=np.float32)  # cfgs/*.yaml network.PIXEL_MEANS


def icosphere(subdiv):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1],
         [-t, 0, -1], [-t, 0, 1]]
    f = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
         [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]]
    v = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = nf
    return np.array(v, dtype=np.float64), np.array(f, dtype=np.int32)


def make_mesh(rng, subdiv=4, diameter=0.2):
    """closed star-shaped mesh: icosphere displaced by a few low-order lobes, anisotropically scaled."""
    v, f = icosphere(subdiv)
    r = np.ones(len(v))
    for _ in range(4):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        r += rng.uniform(0.05, 0.25) * np.cos(rng.integers(2, 5) * np.arccos(np.clip(v @ d, -1, 1)))
    p = v * r[:, None] * rng.uniform(0.6, 1.0, size=3)[None, :]
    p -= p.mean(axis=0, keepdims=True)
    ext = np.linalg.norm(p[:, None, :] - p[None, ::7, :], axis=2).max() if len(p) < 4000 else 2 * np.linalg.norm(p, axis=1).max()
    p *= diameter / ext
    # spherical texture coordinates in [0,1]
    u = 0.5 + np.arctan2(v[:, 1], v[:, 0]) / (2 * np.pi)
    w = 0.5 + np.arcsin(np.clip(v[:, 2], -1, 1)) / np.pi
    return p.astype(np.float32), np.stack([u, w], axis=1).astype(np.float32), f


def make_texture(rng, size=512, cells=32):
    """blocky random colours with a smooth gradient: distinct texels, no pure-black (mask = depth, not colour)."""
    base = rng.integers(40, 256, size=(cells, cells, 3))
    tex = np.kron(base, np.ones((size // cells, size // cells, 1), dtype=np.int64))
    g = np.linspace(0, 30, size).astype(np.int64)
    tex = np.clip(tex - g[:, None, None] + g[None, :, None] // 2, 1, 255)
    return tex.astype(np.uint8)


def make_models(seed=2333, n_models=1, subdiv=4):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_models):
        v, t, f = make_mesh(rng, subdiv=subdiv, diameter=float(rng.uniform(0.1, 0.3)))
        out.append((v, t, f, make_texture(rng)))
    return out


def random_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def euler_xyz(a, b, c):
    ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    Rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
    Ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    Rz = np.array([[cc, -sc, 0], [sc, cc, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def sample_gt_pose(rng, K=LINEMOD_K, W=640, H=480, margin=120):
    """margin keeps the whole (<=0.3 m) object inside the frame at z >= 0.6 m."""
    R = random_rotation(rng)
    z = rng.uniform(0.6, 1.2)
    u = rng.uniform(margin, W - margin)
    v = rng.uniform(margin, H - margin)
    x = (u - K[0, 2]) * z / K[0, 0]
    y = (v - K[1, 2]) * z / K[1, 1]
    return np.concatenate([R, np.array([[x], [y], [z]])], axis=1).astype(np.float32)


def perturb_pose(rng, pose, angle_std=15.0, angle_max=45.0, xy_std=0.01, z_std=0.05):
    while True:
        e = np.deg2rad(rng.normal(0, angle_std, size=3))
        Rn = euler_xyz(*e)
        ang = np.rad2deg(np.arccos(np.clip((np.trace(Rn) - 1) / 2, -1, 1)))
        if ang <= angle_max:
            break
    out = np.array(pose, dtype=np.float64)
    out[:, :3] = Rn @ out[:, :3]
    out[:, 3] += np.array([rng.normal(0, xy_std), rng.normal(0, xy_std), rng.normal(0, z_std)])
    return out.astype(np.float32)


def sample_pairs(seed, B, n_classes=1, **noise):
    """-> class_index (B,) int32, pose_gt (B,3,4), pose_init (B,3,4), background seeds.
    noise: keyword arguments of perturb_pose (angle_std, angle_max, xy_std, z_std) for pairs with another initial error."""
    rng = np.random.default_rng(seed)
    cls = rng.integers(0, n_classes, size=B).astype(np.int32)
    gt = np.stack([sample_gt_pose(rng) for _ in range(B)])
    init = np.stack([perturb_pose(rng, gt[i], **noise) for i in range(B)])
    return cls, gt, init


def bgr_to_blob(bgr_uint8, pixel_means=PIXEL_MEANS):
    """lib/utils/image.py:709-720 transform(): plane c = im[:,:,2-c] - pixel_means[2-c]  ((1,3,H,W) f32)."""
    im = np.asarray(bgr_uint8, dtype=np.float32)
    out = np.zeros((1, 3, im.shape[0], im.shape[1]), dtype=np.float32)
    for i in range(3):
        out[0, i] = im[:, :, 2 - i] - pixel_means[2 - i]
    return out


def plane_means(pixel_means=PIXEL_MEANS):
    """per-plane constants of the network blobs: PIXEL_MEANS reversed (zoom_image_with_factor.py:94)."""
    return np.asarray(pixel_means, dtype=np.float32).reshape(3)[::-1].copy()


def box_from_mask(mask):
    """filled END-EXCLUSIVE bbox rectangle of a binary mask (lib/utils/image.py:437-460, data_pair.py:103-114)."""
    out = np.zeros(mask.shape, dtype=np.float32)
    nz_x = np.nonzero(mask.max(axis=0))[0]
    nz_y = np.nonzero(mask.max(axis=1))[0]
    if len(nz_x) and len(nz_y):
        out[nz_y.min():nz_y.max(), nz_x.min():nz_x.max()] = 1.0
    return out


def compose_observed(bgr_render, depth_render, rng):
    """observed image = render at the GT pose over a seeded uniform-noise background, uint8."""
    bg = rng.integers(0, 256, size=bgr_render.shape).astype(np.float32)
    fg = (depth_render > 0)[..., None]
    return np.where(fg, bgr_render, bg).astype(np.uint8)


def build_device_batch(render_machine, B, seed, n_classes=1, pixel_means=PIXEL_MEANS, device="cuda:0", noise=None, with_depth=False):
    """Synthetic test batch built ON the GPU with the HIP rasteriser (bench / smoke inputs):
    observed = render at the GT pose over seeded uniform noise, uint8-quantised; rendered = render at the
    perturbed pose; mask_rendered = depth > 0.2; mask_observed = bbox rectangle of it (TEST.INIT_MASK box_rendered).
    Returns dict of CUDA tensors with the reference's blob names + pose_gt.
    with_depth: also depth_gt_observed (the render at the GT pose: zero off the object) and depth_rendered (the render at the initial
    pose), the two planes par_generate_gt reads for the test-time flow error (deepim/core/tester.py:681-704)."""
    import torch

    from lib.hip import ops

    d = torch.device(device)
    cls, gt, init = sample_pairs(seed, B, n_classes, **(noise or {}))
    H, W = render_machine.height, render_machine.width
    pm = plane_means(pixel_means)
    cls_t = torch.from_numpy(cls).to(d)
    img = torch.empty((B, 3, H, W), device=d)
    depth = torch.empty((B, 1, H, W), device=d)
    render_machine.render_batch(cls_t, torch.from_numpy(gt).to(d), image=img, depth=depth, plane_means=pm)
    g = torch.Generator(device=d)
    g.manual_seed(seed)
    noise = torch.randint(0, 256, (B, 3, H, W), generator=g, device=d).float() - torch.from_numpy(pm).to(d).view(1, 3, 1, 1)
    image_observed = torch.where(depth > 0, img, noise).contiguous()
    depth_gt = depth.clone() if with_depth else None
    image_rendered = torch.empty((B, 3, H, W), device=d)
    mask_rendered = torch.empty((B, 1, H, W), device=d)
    mask_observed = torch.empty((B, 1, H, W), device=d)
    bbox = torch.empty((B, 4), dtype=torch.int32, device=d)
    render_machine.render_batch(cls_t, torch.from_numpy(init).to(d), image=image_rendered, depth=depth, mask=mask_rendered, bbox=bbox,
                                plane_means=pm)
    ops.box_mask(bbox, mask_observed)
    out = {"image_observed": image_observed, "image_rendered": image_rendered, "mask_observed": mask_observed,
           "mask_rendered": mask_rendered, "src_pose": torch.from_numpy(init).to(d), "class_index": cls_t,
           "pose_gt": torch.from_numpy(gt).to(d)}
    if with_depth:
        out.update(depth_gt_observed=depth_gt, depth_rendered=depth)
    return out


def build_device_train_batch(render_machine, B, seed, models, n_classes=1, pixel_means=PIXEL_MEANS, npts=3000, device="cuda:0", noise=None):
    """build_device_batch + the training blobs/labels (reference names, deepim/core/loader.py:164-193): mask_gt_observed, tgt_pose,
    depth_gt_observed, rot/trans labels, flow/flow_weights (depth->flow kernel), point clouds."""
    import torch

    from lib.hip import ops

    b = build_device_batch(render_machine, B, seed, n_classes=n_classes, pixel_means=pixel_means, device=device, noise=noise)
    d = torch.device(device)
    H, W = render_machine.height, render_machine.width
    K = render_machine.K
    gt, init = b["pose_gt"], b["src_pose"]
    depth_gt = torch.empty((B, 1, H, W), device=d)
    depth_r = torch.empty((B, 1, H, W), device=d)
    mask_gt = torch.empty((B, 1, H, W), device=d)
    render_machine.render_batch(b["class_index"], gt, depth=depth_gt, mask=mask_gt, mask_thr=0.0)
    render_machine.render_batch(b["class_index"], init, depth=depth_r)
    z3, o3 = np.zeros(3, np.float32), np.ones(3, np.float32)
    rot, trans = ops.se3_delta(init, gt, "CAMERA", z3, o3)
    KT = ops.pose_to_KT(init, gt, K)
    flow, valid = ops.depth_to_flow(depth_r, depth_gt, KT, np.linalg.inv(K).astype(np.float32))
    rng = np.random.default_rng(seed + 17)
    cls = b["class_index"].cpu().numpy()
    gt_np = gt.cpu().numpy()
    pm, po = [], []
    for i in range(B):
        v = models[int(cls[i])][0]
        P = np.ascontiguousarray(v[rng.integers(0, v.shape[0], size=npts)].T.astype(np.float32))
        pm.append(P[None])
        po.append((gt_np[i][:, :3] @ P + gt_np[i][:, 3:4]).astype(np.float32)[None])
    b.update(mask_gt_observed=mask_gt, tgt_pose=gt.clone(), depth_gt_observed=depth_gt, rot=rot, trans=trans, flow=flow,
             flow_weights=valid.repeat(1, 2, 1, 1).contiguous(), point_cloud_model=torch.from_numpy(np.concatenate(pm)).to(d),
             point_cloud_weights=torch.ones((B, 3, npts), device=d), point_cloud_observed=torch.from_numpy(np.concatenate(po)).to(d))
    return b
