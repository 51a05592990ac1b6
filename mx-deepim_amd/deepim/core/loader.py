"""Data loaders: pairdb records -> device-resident batches, double buffered, with a decoded-pixel cache in HBM.

Counterparts of the reference's `TestDataLoader` (deepim/core/loader.py:20-117) and `TrainDataLoader` (:120-421): same constructor
arguments, `data_name` / `label_name` lists, `reset()` / iteration protocol, one batch = `batch_size` pairs per GPU.  What differs
is where the bytes are turned into blobs:

  reference   a process pool builds float32 blobs on the host (lib/pair_matching/data_pair.py:22-72 / :144-265: 9.8 MB of image
              blobs per 480x640 pair, + 7.4 MB of flow labels and masks in training) and the executor uploads them
  here        a thread pool decodes the image files into PINNED staging buffers in their file representation -- 8-bit BGR colour,
              16-bit depth, 8-bit labels: 2.1 MB per test pair, 3.1 MB per training pair --, a copy stream moves them to the GPU, and
              csrc/data.hip builds every blob in HBM: images (incl. the VOC background paste), masks by INIT_MASK kind (+ mask_dilate
              with host-drawn thicknesses), depth planes, flow labels (calc_flow), SE(3) labels, point clouds.  Two staging sets
              alternate, so decode + upload of batch k+1 overlap the work on batch k; the compute stream waits on one event per batch.
  cache       `PixelCache`: the decoded pixels of a file stay in HBM (uint8 / uint16, as decoded) keyed by path.  A hit fills the
              staging slot with a device-to-device copy -- no decode, no PCIe: every epoch after the first, and pairs that share an
              observed image inside an epoch.  LINEMOD's training set is ~3 MB per pair decoded, so a whole dataset fits the 288 GB.

Randomness (training) is drawn by the single staging thread in pair order with the reference's calls in the reference's order for one pair
(get_data_pair_train_batch([rec]): np.random.randint(18); random.randrange(len(SCALES)); [np.random.rand()]; [random.randint bg];
mask_dilate's draws; np.random.shuffle for the point sample), so a seeded run is reproducible as long as nothing else draws from
the global generators while an epoch is being staged -- the reference itself is not reproducible (its draws happen in pool processes).

Not built (raise): img_flipped (the reference raises too), TRAIN.MASK_SYN, network.MASK_INPUTS, SCALES other than the image size.
`RawPairSource` is the seam for pixels that do not come from files (bench.py's synthetic `fresh_batch` line).
"""
from __future__ import print_function, division

import os
import random
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from lib.hip import ops


# ------------------------------------------------------------------------------------------------------------------ sources (test)
class RawPairSource(object):
    """len() pairs; fill(i, obs_bgr (H,W,3) u8, ren_bgr (H,W,3) u8, depth (H,W) u16) writes pair i's pixels into the given (pinned)
    arrays and returns (pose_rendered 3x4, class_index, pose_observed 3x4 or None)."""

    def __len__(self):
        raise NotImplementedError

    def fill(self, i, obs_bgr, ren_bgr, depth):
        raise NotImplementedError


class ArraySource(RawPairSource):
    """pairs already in host memory as raw pixels (a decoded dataset cached in RAM; the bench's synthetic pairs)"""

    def __init__(self, obs_bgr, ren_bgr, depth, pose_rendered, class_index, pose_observed=None, repeat=1):
        """repeat: present the stored pairs `repeat` times over (an endless-enough stream for a throughput run)"""
        self.obs, self.ren, self.depth = obs_bgr, ren_bgr, depth
        self.pose, self.cls, self.gt = pose_rendered, class_index, pose_observed
        self.repeat = int(repeat)

    def __len__(self):
        return self.obs.shape[0] * self.repeat

    def fill(self, i, obs_bgr, ren_bgr, depth):
        i = i % self.obs.shape[0]
        obs_bgr[...] = self.obs[i]
        ren_bgr[...] = self.ren[i]
        depth[...] = self.depth[i]
        return self.pose[i], int(self.cls[i]), (self.gt[i] if self.gt is not None else None)


def raw_from_device_batch(batch, pixel_means, depth_rendered, depth_factor=1000.0):
    """device blobs of a synthetic batch (lib/utils/synthetic.build_device_batch) -> the raw host arrays an image file would hold:
    (obs_bgr u8 (B,H,W,3), ren_bgr u8, depth u16 (B,H,W), pose_rendered, class_index, pose_observed)"""
    pm = np.asarray(pixel_means, dtype=np.float32).reshape(3)   # config order B, G, R; blob plane c = BGR channel 2 - c minus pm[2 - c]

    def bgr(blob):
        rgb = blob.cpu().numpy() + pm[::-1].reshape(1, 3, 1, 1)
        return np.ascontiguousarray(np.clip(np.rint(rgb), 0, 255).astype(np.uint8).transpose(0, 2, 3, 1)[..., ::-1])

    d = np.clip(np.rint(depth_rendered.cpu().numpy()[:, 0] * depth_factor), 0, 65535).astype(np.uint16)
    return (bgr(batch["image_observed"]), bgr(batch["image_rendered"]), d, batch["src_pose"].cpu().numpy(),
            batch["class_index"].cpu().numpy().astype(np.int32), batch["pose_gt"].cpu().numpy() if "pose_gt" in batch else None)


# ------------------------------------------------------------------------------------------------------------------ pixel cache
class PixelCache(object):
    """decoded pixels of image files, resident in HBM, keyed by (path, kind).  Entries are immutable device tensors in the file's own
    representation; `budget_bytes` bounds the total (a file that does not fit is simply not cached).  Main-thread use only."""

    def __init__(self, device, budget_bytes=64 << 30):
        self.device, self.budget, self.used = torch.device(device), int(budget_bytes), 0
        self.entries = {}
        self.hits = self.misses = 0

    def get(self, key):
        e = self.entries.get(key)
        if e is None:
            self.misses += 1
        else:
            self.hits += 1
        return e

    def put(self, key, dev_tensor, meta=None):
        """clone `dev_tensor` (on the current stream) into the cache; -> True if it was stored"""
        n = dev_tensor.numel() * dev_tensor.element_size()
        if key in self.entries or self.used + n > self.budget:
            return False
        try:
            copy = dev_tensor.clone()
        except torch.cuda.OutOfMemoryError:   # a full HBM costs a cache entry, not the training run
            self.budget = self.used
            return False
        self.entries[key] = (copy, meta)
        self.used += n
        return True

    def __len__(self):
        return len(self.entries)


# ------------------------------------------------------------------------------------------------------------------ staging
_FIELD_SHAPES = {"obs": (3, torch.uint8), "ren": (3, torch.uint8), "bg": (3, torch.uint8), "depth": (0, torch.uint16),
                 "depth_gt": (0, torch.uint16), "depth_obs": (0, torch.uint16), "label": (0, torch.uint8)}


class _Staging(object):
    """one pinned host set + its device mirror + the events that order it against the copy and the compute stream"""

    def __init__(self, B, H, W, device, fields, n_points=0):
        pin = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory()  # noqa: E731
        self.h, self.d = {}, {}
        for f in fields:
            c, dt = _FIELD_SHAPES[f]
            self.h[f] = pin((B, H, W, 3) if c else (B, H, W), dt)
            self.d[f] = torch.empty_like(self.h[f], device=device)
        meta = {"pose": ((B, 3, 4), torch.float32), "cls": ((B,), torch.int32), "gt": ((B, 3, 4), torch.float32),
                "mask_idx": ((B,), torch.int32), "use_bg": ((B,), torch.int32), "thick": ((B, 4), torch.int32),
                "P12": ((B, 3, 4), torch.float64), "tab_off": ((B,), torch.int32)}
        if n_points:
            meta["pt_idx"] = ((B, n_points), torch.int32)
        self.mh = {k: pin(s, dt) for k, (s, dt) in meta.items()}
        self.md = {k: torch.empty_like(v, device=device) for k, v in self.mh.items()}
        self.ready = torch.cuda.Event()
        self.consumed = torch.cuda.Event()
        self.n = 0
        self.has_gt = False
        self.any_bg = False

    # names the refinement loop reads (deepim/core/tester.py Refiner.load_staged)
    d_obs = property(lambda self: self.d["obs"])
    d_ren = property(lambda self: self.d["ren"])
    d_depth = property(lambda self: self.d["depth"])
    d_pose = property(lambda self: self.md["pose"])
    d_cls = property(lambda self: self.md["cls"])
    d_gt = property(lambda self: self.md["gt"])


class _DeviceLoader(object):
    """shared machinery: index / shuffle, the decode pool, the copy stream, two staging sets, the cache"""

    def _setup(self, size, batch_size, shuffle, device, workers, height, width, fields, n_points=0, cache=None):
        self.size, self.batch_size, self.shuffle = int(size), int(batch_size), shuffle
        self.index = np.arange(self.size)
        self.device = torch.device(device)
        self.H, self.W = height, width
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(workers)))
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.fields = tuple(fields)
        self.sets = [_Staging(self.batch_size, self.H, self.W, self.device, self.fields, n_points) for _ in range(2)]
        self.cache = cache
        # ONE background thread stages batch k+1 (draws, cache look-ups, decode fan-out over the pool, copy-stream traffic) while the
        # caller's thread is already enqueueing the work on batch k: with staging in the caller's thread the GPU sat idle for the
        # decode time of every batch.  A single stager keeps the random draws in pair order.
        self._stager = ThreadPoolExecutor(max_workers=1)
        # The stager runs ~2 ms of Python per cached batch next to the caller's thread, which is enqueueing a GPU-bound training step:
        # with CPython's default 5 ms switch interval the caller can lose the interpreter for longer than its launch queue is deep
        # and the GPU idles (measured: 60.3 ms per batch fed from the cache against 56.9 for private copies of the same batches).
        import sys

        if sys.getswitchinterval() > 5e-4:
            sys.setswitchinterval(5e-4)
        self._lock = threading.Lock()

    def __len__(self):
        return self.size // self.batch_size   # whole batches (the resident executors are built for one batch size)

    def _fresh(self):
        """nothing consumed since the last reset(): a second reset() in a row (an epoch loop that resets before its first epoch,
        right after the constructor did) keeps the batch already being staged"""
        return getattr(self, "_handed_out", 1) == 0

    def _rewind(self):
        self._handed_out = 0
        if getattr(self, "_inflight", None) is not None:
            self._inflight.result()   # a batch staged ahead but never consumed: let it finish before the sets are reused
        self.cur = 0
        self._slot = 0
        self._inflight = None

    def _prefetch(self):
        """start staging the first batch of the epoch right away (called at the end of reset())"""
        if self._inflight is None:
            self._inflight = self._submit()

    def iter_next(self):
        return self._inflight is not None or self.cur + self.batch_size <= self.size

    def _stage_in_thread(self, st, first):
        torch.cuda.set_device(self.device)
        return self._stage(st, first)

    def _submit(self):
        """start staging the next batch in the background -> a future of its staging set, or None at the end of the epoch"""
        if self.cur + self.batch_size > self.size:
            return None
        st = self.sets[self._slot]
        self._slot ^= 1
        first, self.cur = self.cur, self.cur + self.batch_size
        return self._stager.submit(self._stage_in_thread, st, first)

    def next_raw(self):
        """-> the staging set of the next batch (device mirrors valid once `ready` has been waited for); the batch after it starts
        staging in the background before this one is handed out"""
        fut = self._inflight if self._inflight is not None else self._submit()
        if fut is None:
            raise StopIteration
        st = fut.result()
        self._handed_out += 1
        self._inflight = self._submit()
        return st

    def _fill_files(self, st, jobs):
        """jobs: [(field, slot j, cache key or None, decode() -> ndarray, meta_of(ndarray) or None)].  Cache hits become device-to-device
        copies on the copy stream, misses are decoded by the pool into the pinned set, uploaded slot by slot and then cached.
        -> {(field, j): meta} for the entries that carry one"""
        metas, todo, hit_dst, hit_src = {}, [], {}, {}
        for field, j, key, decode, meta_of in jobs:
            hit = self.cache.get(key) if (self.cache is not None and key is not None) else None
            if hit is not None:
                # (the multi-tensor copy has no uint16 kernel: 16-bit depth goes as int16, the same bytes)
                as_copyable = (lambda t: t.view(torch.int16)) if hit[0].dtype == torch.uint16 else (lambda t: t)
                hit_dst.setdefault(hit[0].dtype, []).append(as_copyable(st.d[field][j]))
                hit_src.setdefault(hit[0].dtype, []).append(as_copyable(hit[0]))
                metas[(field, j)] = hit[1]
            else:
                todo.append((field, j, key, decode, meta_of))
        with torch.cuda.stream(self.copy_stream):
            for dt, dst in hit_dst.items():   # one multi-tensor copy per element type instead of one launch per file
                torch._foreach_copy_(dst, hit_src[dt], non_blocking=True)

        def run(job):
            field, j, key, decode, meta_of = job
            a = decode()
            st.h[field].numpy()[j][...] = a
            return meta_of(a) if meta_of is not None else None

        results = list(self.pool.map(run, todo))
        with torch.cuda.stream(self.copy_stream):
            for (field, j, key, _, _), meta in zip(todo, results):
                st.d[field][j].copy_(st.h[field][j], non_blocking=True)
                metas[(field, j)] = meta
                if self.cache is not None and key is not None:
                    self.cache.put(key, st.d[field][j], meta)
        return metas

    def release(self, st):
        """everything that reads the device mirrors of `st` has been enqueued on the current stream: the set may be re-staged once
        that work is done (build_blobs records the same event; a consumer that reads st.d_* / st.md afterwards calls this again)"""
        st.consumed.record()

    def _upload_meta(self, st, names):
        with torch.cuda.stream(self.copy_stream):
            for k in names:
                st.md[k].copy_(st.mh[k], non_blocking=True)
            st.ready.record(self.copy_stream)

    def close(self):
        self._stager.shutdown(wait=True)
        self.pool.shutdown(wait=True)

    def __iter__(self):
        return self

    def __next__(self):
        return self.next()


def _DeviceLoader_reset(self):
    self._rewind()
    if self.shuffle:
        np.random.shuffle(self.index)
    self._prefetch()


def _check_common(cfg, H, W):
    scales = [tuple(int(v) for v in s) for s in cfg.SCALES]
    if any(s != (min(H, W), max(H, W)) for s in scales):
        raise NotImplementedError("device loader: SCALES must be [[{}, {}]] (resize is the identity), got {}".format(min(H, W), max(H, W), cfg.SCALES))
    if cfg.network.get("MASK_INPUTS", False):
        raise NotImplementedError("device loader: network.MASK_INPUTS")


def _imread_color(path):
    from lib.utils.image import imread_color

    return imread_color(path)


def _imread_unchanged(path):
    from lib.utils.image import imread_unchanged

    return imread_unchanged(path)


# ------------------------------------------------------------------------------------------------------------------ test loader
TEST_LABEL_OF_KIND = {"mask_gt_observed": "mask_gt_observed", "box_gt_observed": "mask_gt_observed", "mask_observed": "mask_observed",
                      "box_": "mask_observed"}


class TestDataLoader(_DeviceLoader):
    """reference deepim/core/loader.py:20-117.  Every TEST.INIT_MASK kind of get_pair_mask (image.py:367-476), TEST.MASK_DILATE and
    INPUT_DEPTH are built on the device; `source` (a RawPairSource) replaces the files for the shipped kind 'box_rendered' only."""

    def __init__(self, pairdb, config, batch_size=1, shuffle=False, device="cuda:0", workers=8, source=None, height=480, width=640,
                 cache=None):
        cfg = config
        _check_common(cfg, height, width)
        self.input_mask, self.input_depth = bool(cfg.network.INPUT_MASK), bool(cfg.network.INPUT_DEPTH)
        self.init_mask = cfg.TEST.INIT_MASK
        self.dilate = bool(cfg.TEST.get("MASK_DILATE", False))
        if self.input_mask and self.init_mask not in ("box_rendered",) + tuple(TEST_LABEL_OF_KIND):
            raise Exception("Unknown init mask type: {}".format(self.init_mask))
        self.label_key = TEST_LABEL_OF_KIND.get(self.init_mask) if self.input_mask else None
        if source is not None and (self.label_key or self.input_depth):
            raise NotImplementedError("a RawPairSource carries colour + rendered depth only: INIT_MASK 'box_rendered' without INPUT_DEPTH")
        self.source = source
        self.pairdb, self.config = pairdb, config
        fields = ["obs", "ren", "depth"] + (["label"] if self.label_key else []) + (["depth_obs"] if self.input_depth else [])
        self._setup(len(source) if source is not None else len(pairdb), batch_size, shuffle, device, workers, height, width, fields, cache=cache)
        self.data_name = ["image_observed", "image_rendered", "src_pose", "class_index"]
        if self.input_depth:
            self.data_name += ["depth_observed", "depth_rendered"]
        if self.input_mask:
            self.data_name += ["mask_observed", "mask_rendered"]
        self.label_name = None
        self.pixel_means_bgr = np.asarray(cfg.network.PIXEL_MEANS, dtype=np.float32).reshape(3)
        self.depth_factor = float(cfg.dataset.DEPTH_FACTOR)
        B, f32, d = self.batch_size, torch.float32, self.device
        plane = lambda c: torch.empty((B, c, self.H, self.W), dtype=f32, device=d)  # noqa: E731
        self.blobs = {"image_observed": plane(3), "image_rendered": plane(3), "mask_observed": plane(1), "mask_rendered": plane(1)}
        if self.input_depth:
            self.blobs["depth_observed"], self.blobs["depth_rendered"] = plane(1), plane(1)
        self.mask_tmp = plane(1) if self.dilate else None
        self.bbox = torch.empty((B, 4), dtype=torch.int32, device=d)
        self.bbox_label = torch.empty((B, 4), dtype=torch.int32, device=d)
        self._meta_out = {}
        self.reset()

    def reset(self):
        if self._fresh():
            return
        self._rewind()
        if self.shuffle:
            np.random.shuffle(self.index)
        self._prefetch()

    def _stage(self, st, first):
        ids = self.index[first:first + self.batch_size]
        st.consumed.synchronize()   # the device mirror of this set may still feed the blobs kernel of two batches ago
        hp, hc, hg, hm, ht = (st.mh[k].numpy() for k in ("pose", "cls", "gt", "mask_idx", "thick"))
        ht[...] = 0
        if self.source is not None:
            ho, hr, hd = st.h["obs"].numpy(), st.h["ren"].numpy(), st.h["depth"].numpy()
            meta = list(self.pool.map(lambda j: self.source.fill(int(ids[j]), ho[j], hr[j], hd[j]), range(len(ids))))
            st.has_gt = all(m[2] is not None for m in meta)
            for j, (pose, cls, gt) in enumerate(meta):
                hp[j], hc[j], hm[j] = pose, cls, 1
                if gt is not None:
                    hg[j] = gt
            with torch.cuda.stream(self.copy_stream):
                for f in ("obs", "ren", "depth"):
                    st.d[f].copy_(st.h[f], non_blocking=True)
        else:
            jobs = []
            for j, i in enumerate(ids):
                rec = self.pairdb[int(i)]
                if rec.get("img_flipped"):
                    raise Exception("NOT_IMPLEMENTED")
                jobs.append(("obs", j, (rec["image_observed"], "c"), lambda p=rec["image_observed"]: _imread_color(p), None))
                jobs.append(("ren", j, (rec["image_rendered"], "c"), lambda p=rec["image_rendered"]: _imread_color(p), None))
                jobs.append(("depth", j, (rec["depth_rendered"], "u"), lambda p=rec["depth_rendered"]: _imread_unchanged(p),
                             lambda a: bool(a.any())))
                if self.label_key:
                    jobs.append(("label", j, (rec[self.label_key], "u"), lambda p=rec[self.label_key]: _imread_unchanged(p), None))
                if self.input_depth:
                    jobs.append(("depth_obs", j, (rec["depth_observed"], "u"), lambda p=rec["depth_observed"]: _imread_unchanged(p), None))
            metas = self._fill_files(st, jobs)
            st.has_gt = all("pose_observed" in self.pairdb[int(i)] for i in ids)
            for j, i in enumerate(ids):
                rec = self.pairdb[int(i)]
                hp[j] = np.asarray(rec["pose_rendered"], np.float32)
                hc[j] = self.config.dataset.class_name.index(rec["gt_class"])
                # an all-zero rendered depth marks an undetected object: every INIT_MASK kind gives an empty mask (image.py:370-380);
                # label -1 matches no pixel
                hm[j] = int(rec.get("mask_idx", 1)) if metas[("depth", j)] else -1
                if st.has_gt:
                    hg[j] = np.asarray(rec["pose_observed"], np.float32)
        if self.dilate and self.input_mask:
            for j in range(len(ids)):
                ht[j] = draw_dilate_thickness(10)   # get_pair_mask: mask_dilate(cur_mask_observed, max_thickness=10), pair by pair
        st.n = len(ids)
        self._upload_meta(st, ("pose", "cls", "gt", "mask_idx", "thick"))
        return st

    def build_blobs(self, st, out=None):
        """device: raw pixels of staging set `st` -> the float blobs (into `out`, a dict of tensors, or the loader's own)"""
        out = out if out is not None else self.blobs
        torch.cuda.current_stream().wait_event(st.ready)
        by_label = self.label_key is not None
        mask_direct = self.input_mask and by_label and self.init_mask in ("mask_gt_observed", "mask_observed")
        first = self.mask_tmp if self.dilate else out.get("mask_observed")
        ops.pair_blobs_from_raw(self.batch_size, self.H, self.W, self.depth_factor, self.pixel_means_bgr, obs_bgr=st.d["obs"], ren_bgr=st.d["ren"],
                                depth_ren=st.d["depth"], depth_b=st.d.get("depth_obs"), label=st.d.get("label"),
                                mask_idx=st.md["mask_idx"] if by_label else None, image_observed=out["image_observed"],
                                image_rendered=out["image_rendered"], mask_rendered=out.get("mask_rendered") if self.input_mask else None,
                                depth_rendered=out.get("depth_rendered") if self.input_depth else None,
                                depth_b_out=out.get("depth_observed") if self.input_depth else None,
                                mask_label=first if mask_direct else None, bbox_ren=self.bbox,
                                bbox_label=self.bbox_label if (by_label and not mask_direct) else None)
        if self.input_mask:
            if not mask_direct:
                ops.box_mask(self.bbox_label if by_label else self.bbox, first)
            if self.dilate:
                ops.mask_dilate(self.mask_tmp, st.md["thick"], out=out["mask_observed"])
        st.consumed.record()
        return out

    def next(self):
        st = self.next_raw()
        b = dict(self.build_blobs(st))
        # the small per-pair arrays leave the staging set too (it is re-staged while this batch may still be in use)
        for name, src in (("src_pose", st.d_pose), ("class_index", st.d_cls)) + ((("pose_observed", st.d_gt),) if st.has_gt else ()):
            if name not in self._meta_out:
                self._meta_out[name] = torch.empty_like(src)
            ops.copy(self._meta_out[name], src)
            b[name] = self._meta_out[name]
        self.release(st)
        return b


def draw_dilate_thickness(max_thickness):
    """the draws of mask_dilate (lib/utils/mask_dilate.py:10-55) in its order -> {down, up, right, left}, 0 = side skipped"""
    direction = np.random.randint(10)
    out = [0, 0, 0, 0]
    for side, skip in enumerate(((0, 1, 4), (1, 2, 5), (2, 3, 6), (0, 3, 7))):
        if direction not in skip:
            out[side] = np.random.randint(max_thickness) + 1
    return out


# ------------------------------------------------------------------------------------------------------------------ train loader
class TrainDataLoader(_DeviceLoader):
    """reference deepim/core/loader.py:120-421 (`get_batch_parallel` :294-420 -> lib/pair_matching/data_pair.py:144-265).
    `sym` and `ctx` are accepted for signature parity (one process drives one GPU: the batch IS this GPU's BATCH_PAIRS).
    next() -> dict of device tensors: the data blobs and the label blobs under the reference's names (`data_name`, `label_name`)."""

    def __init__(self, sym, pairdb, config, batch_size=1, shuffle=False, ctx=None, work_load_list=None, device="cuda:0", workers=8,
                 height=480, width=640, cache=None):
        cfg = config
        _check_common(cfg, height, width)
        if cfg.TRAIN.get("MASK_SYN", False):
            raise NotImplementedError("device loader: TRAIN.MASK_SYN")
        self.sym, self.pairdb, self.config, self.ctx = sym, pairdb, config, ctx
        self.input_mask, self.input_depth = bool(cfg.network.INPUT_MASK), bool(cfg.network.INPUT_DEPTH)
        self.pred_mask, self.pred_flow = bool(cfg.network.PRED_MASK), bool(cfg.network.PRED_FLOW)
        self.pm_loss = bool(cfg.train_iter.SE3_PM_LOSS)
        self.init_mask, self.dilate = cfg.TRAIN.INIT_MASK, bool(cfg.TRAIN.MASK_DILATE)
        if self.init_mask not in ("mask_gt", "box_gt", "box_rendered"):
            raise Exception("Unknown mask type: {}".format(self.init_mask))
        self.n_points = int(cfg.train_iter.NUM_3D_SAMPLE) if self.pm_loss else 0
        self.may_paste = any("data_syn" in rec for rec in pairdb)
        fields = ["obs", "ren", "depth", "depth_gt", "label"] + (["depth_obs"] if self.input_depth else []) + (["bg"] if self.may_paste else [])
        self._setup(len(pairdb), batch_size, shuffle, device, workers, height, width, fields, n_points=self.n_points, cache=cache)
        self.data_name = ["image_observed", "image_rendered", "depth_gt_observed", "class_index", "src_pose", "tgt_pose"]
        if self.input_depth:
            self.data_name += ["depth_observed", "depth_rendered"]
        if self.input_mask:
            self.data_name += ["mask_observed", "mask_rendered"]
        self.label_name = ["rot", "trans"]
        if self.pred_mask:
            self.label_name.append("mask_gt_observed")
        if self.pred_flow:
            self.label_name += ["flow", "flow_weights"]
        if self.pm_loss:
            self.label_name += ["point_cloud_model", "point_cloud_weights", "point_cloud_observed"]
        self.pixel_means_bgr = np.asarray(cfg.network.PIXEL_MEANS, dtype=np.float32).reshape(3)
        self.depth_factor = float(cfg.dataset.DEPTH_FACTOR)
        self.K = np.asarray(cfg.dataset.INTRINSIC_MATRIX)
        self.Kinv64 = np.linalg.inv(np.asarray(self.K, dtype=np.float64).reshape(3, 3))
        B, f32, d = self.batch_size, torch.float32, self.device
        plane = lambda c: torch.empty((B, c, self.H, self.W), dtype=f32, device=d)  # noqa: E731

        def blob_set():
            b = {"image_observed": plane(3), "image_rendered": plane(3), "depth_gt_observed": plane(1), "mask_observed": plane(1),
                 "mask_rendered": plane(1), "mask_gt_observed": plane(1), "depth_rendered": plane(1)}
            if self.input_depth:
                b["depth_observed"] = plane(1)
            if self.pred_flow:
                b["flow"], b["flow_weights"] = plane(2), plane(2)
            if self.pm_loss:
                for k in ("point_cloud_model", "point_cloud_weights", "point_cloud_observed"):
                    b[k] = torch.empty((B, 3, self.n_points), dtype=f32, device=d)
            n_rot = {"quat": 4, "matrix": 9, "euler": 3}[str(cfg.network.ROT_TYPE).lower()]
            b.update(rot=torch.empty((B, n_rot), dtype=f32, device=d), trans=torch.empty((B, 3), dtype=f32, device=d),
                     src_pose=torch.empty((B, 3, 4), dtype=f32, device=d), tgt_pose=torch.empty((B, 3, 4), dtype=f32, device=d),
                     class_index=torch.empty((B,), dtype=torch.int32, device=d),
                     _mask_tmp=plane(1), _bbox=torch.empty((B, 4), dtype=torch.int32, device=d),
                     _bbox_label=torch.empty((B, 4), dtype=torch.int32, device=d))
            return b

        # TWO blob sets and a build stream: the blobs of batch k+1 are built (by the staging thread, on its own stream) while the
        # caller still trains on batch k, so handing out a batch costs the consumer's stream one event wait -- the ~1.5 ms of blob
        # kernels per batch no longer sit between two training steps.  `blob_free[s]`: everything the consumer enqueued on the batch
        # that last used set s (recorded when it asks for the next batch); `built` (per staging set): the blobs are complete.
        self.blob_sets = [blob_set(), blob_set()]
        self.blobs = self.blob_sets[0]
        self.build_stream = torch.cuda.Stream(device=self.device)
        self.blob_free = [torch.cuda.Event(), torch.cuda.Event()]
        self._build_count = 0
        self._last_set = None
        self._meta_out = {}
        self._point_tables = {}      # class -> (offset into the device table, n points)
        self._table = None
        # the reference's seeds (loader.py:203-208)
        random.seed(6)
        np.random.seed(3)
        self.rseed = np.random.randint(999999, size=[99999])
        np.random.seed(self.rseed[0])
        self.reset()

    # ---- point tables ------------------------------------------------------------------------------------------------------------
    def _points_of(self, cls):
        """(offset, count) of the class's points.xyz in the device table; loaded on first use (image.py:559-573)"""
        if cls not in self._point_tables:
            from lib.utils.image import load_object_points, point_cloud_dict

            if cls not in point_cloud_dict:
                cfg = self.config
                if not cfg.dataset.dataset.startswith("ModelNet"):
                    point_cloud_dict[cls] = load_object_points(os.path.join(cfg.dataset.model_dir, cls, "points.xyz"))
                else:
                    from lib.render_hip.render_py_light_modelnet_multi import load_obj_with_normals

                    point_cloud_dict[cls] = load_obj_with_normals(os.path.join(cfg.dataset.model_dir, cls + ".obj"))[0].astype(np.float64)
            pts = torch.as_tensor(np.ascontiguousarray(point_cloud_dict[cls], dtype=np.float32)).to(self.device)
            off = 0 if self._table is None else int(self._table.shape[0])
            old_table = self._table
            if old_table is not None:
                # (once per class) point_clouds of the previous batch may still be reading the old table on the build stream: it must
                # not go back to the allocator -- and from there to the training thread -- before that kernel has run
                self.build_stream.synchronize()
            self._table = pts if old_table is None else torch.cat([old_table, pts])
            torch.cuda.current_stream().synchronize()   # the new table is read on the build stream
            del old_table
            self._point_tables[cls] = (off, int(pts.shape[0]))
        return self._point_tables[cls]

    # ---- producer ----------------------------------------------------------------------------------------------------------------
    def _stage(self, st, first):
        from lib.utils.image import _voc_backgrounds, fit_background
        from lib.utils.projection import se3_inverse, se3_mul

        cfg = self.config
        ids = self.index[first:first + self.batch_size]
        st.consumed.synchronize()
        mh = {k: v.numpy() for k, v in st.mh.items()}
        mh["thick"][...] = 0
        mh["use_bg"][...] = 0
        jobs = []
        for j, i in enumerate(ids):
            rec = self.pairdb[int(i)]
            if rec.get("img_flipped"):
                raise Exception("NOT_IMPLEMENTED")
            # ---- the draws of get_data_pair_train_batch([rec]) in the reference's order
            np.random.randint(18)                          # random_k (data_pair.py:151; unused by the shipped getters)
            random.randrange(len(cfg.SCALES))              # scale_ind (image.py:83)
            if "data_syn" in rec and (rec["data_syn"] is True or (rec["data_syn"] is False and np.random.rand() < cfg.TRAIN.REPLACE_OBSERVED_BG_RATIO)):
                voc_root, names = _voc_backgrounds(cfg)
                bg_path = os.path.join(voc_root, "JPEGImages/{}.jpg".format(names[random.randint(0, len(names) - 1)]))
                mh["use_bg"][j] = 1
                jobs.append(("bg", j, (bg_path, "bg", self.H, self.W),
                             lambda p=bg_path: fit_background(_imread_color(p), self.H, self.W), None))
            if self.dilate and (self.input_mask or self.pred_mask):
                mh["thick"][j] = draw_dilate_thickness(10)
            if self.pm_loss:
                off, n_all = self._points_of(rec["gt_class"])
                keep = np.arange(n_all)
                np.random.shuffle(keep)
                keep = keep[:min(n_all, self.n_points)]
                mh["pt_idx"][j, :] = -1
                mh["pt_idx"][j, :len(keep)] = keep
                mh["tab_off"][j] = off
            # ---- files
            jobs.append(("obs", j, (rec["image_observed"], "c"), lambda p=rec["image_observed"]: _imread_color(p), None))
            jobs.append(("ren", j, (rec["image_rendered"], "c"), lambda p=rec["image_rendered"]: _imread_color(p), None))
            jobs.append(("depth", j, (rec["depth_rendered"], "u"), lambda p=rec["depth_rendered"]: _imread_unchanged(p), None))
            jobs.append(("depth_gt", j, (rec["depth_gt_observed"], "u"), lambda p=rec["depth_gt_observed"]: _imread_unchanged(p), None))
            jobs.append(("label", j, (rec["mask_gt_observed"], "u"), lambda p=rec["mask_gt_observed"]: _imread_unchanged(p), None))
            if self.input_depth:
                jobs.append(("depth_obs", j, (rec["depth_observed"], "u"), lambda p=rec["depth_observed"]: _imread_unchanged(p), None))
            # ---- poses, labels' host-side constants
            src, tgt = np.asarray(rec["pose_rendered"]), np.asarray(rec["pose_observed"])
            mh["pose"][j], mh["gt"][j] = src, tgt
            mh["cls"][j] = cfg.dataset.class_name.index(rec["gt_class"])
            mh["mask_idx"][j] = int(rec["mask_idx"])
            # calc_flow's projection K se3_mul(tgt, se3_inverse(src)) exactly as the reference forms it on the host (float32 out of the
            # float32 se3 helpers when K is float32); shipped as float64, which holds either precision exactly
            mh["P12"][j] = np.matmul(self.K, se3_mul(tgt, se3_inverse(src)))
        st.any_bg = bool(mh["use_bg"].any())
        self._fill_files(st, jobs)
        st.n, st.has_gt = len(ids), True
        self._upload_meta(st, tuple(st.mh.keys()))
        # build the blobs right away, on the build stream, into the set the consumer is not using
        st.blob_set = self._build_count & 1
        self._build_count += 1
        with torch.cuda.stream(self.build_stream):
            self.build_stream.wait_event(self.blob_free[st.blob_set])
            self.build_blobs(st, out=self.blob_sets[st.blob_set])
            st.built = torch.cuda.Event()
            st.built.record(self.build_stream)
        return st

    # ---- consumer ----------------------------------------------------------------------------------------------------------------
    def build_blobs(self, st, out=None):
        out = out if out is not None else self.blobs
        cfg = self.config
        torch.cuda.current_stream().wait_event(st.ready)
        need_mask = self.input_mask or self.pred_mask
        raw_label = need_mask and self.init_mask == "mask_gt"
        mask_tmp, bbox, bbox_label = out["_mask_tmp"], out["_bbox"], out["_bbox_label"]
        first = mask_tmp if self.dilate else out["mask_observed"]
        ops.pair_blobs_from_raw(self.batch_size, self.H, self.W, self.depth_factor, self.pixel_means_bgr, obs_bgr=st.d["obs"],
                                bg_bgr=st.d["bg"] if (self.may_paste and st.any_bg) else None, use_bg=st.md["use_bg"], ren_bgr=st.d["ren"],
                                depth_ren=st.d["depth"], depth_a=st.d["depth_gt"], depth_b=st.d.get("depth_obs"), label=st.d["label"],
                                mask_idx=st.md["mask_idx"], image_observed=out["image_observed"], image_rendered=out["image_rendered"],
                                mask_rendered=out["mask_rendered"], depth_rendered=out["depth_rendered"], depth_a_out=out["depth_gt_observed"],
                                depth_b_out=out.get("depth_observed") if self.input_depth else None, mask_label=out["mask_gt_observed"],
                                label_raw=first if raw_label else None, bbox_ren=bbox, bbox_label=bbox_label)
        if need_mask:
            if self.init_mask == "box_gt":
                ops.box_mask(bbox_label, first)
            elif self.init_mask == "box_rendered":
                ops.box_mask(bbox, first)
            if self.dilate:
                ops.mask_dilate(mask_tmp, st.md["thick"], out=out["mask_observed"])
        kind = str(cfg.network.ROT_TYPE).lower()
        delta = {"quat": ops.se3_delta, "matrix": ops.se3_delta_matrix, "euler": ops.se3_delta_euler}[kind]
        rot, trans = delta(st.md["pose"], st.md["gt"], cfg.network.ROT_COORD, cfg.dataset.trans_means, cfg.dataset.trans_stds)
        ops.copy(out["rot"], rot.reshape(self.batch_size, -1))
        ops.copy(out["trans"], trans)
        if self.pred_flow:
            ops.calc_flow_labels(out["depth_rendered"], out["depth_gt_observed"], st.md["P12"], self.Kinv64, out["flow"], out["flow_weights"],
                                 standard_rep=bool(cfg.network.STANDARD_FLOW_REP), weight_type=cfg.TRAIN.FLOW_WEIGHT_TYPE)
        if self.pm_loss:
            ops.point_clouds(self._table, st.md["tab_off"], st.md["pt_idx"], st.md["gt"], out["point_cloud_model"], out["point_cloud_weights"],
                             out["point_cloud_observed"])
        # the small per-pair arrays leave the staging set too: it is re-staged while this batch is still trained on
        ops.copy(out["src_pose"], st.md["pose"])
        ops.copy(out["tgt_pose"], st.md["gt"])
        ops.copy(out["class_index"], st.md["cls"])
        st.consumed.record()
        return out

    def reset(self):
        if self._fresh():
            return
        # the consumer has enqueued everything it will ever do on the batches of the epoch before: both blob sets may be rebuilt after it
        for e in self.blob_free:
            e.record()
        self._last_set = None
        _DeviceLoader_reset(self)

    def next(self):
        if self._last_set is not None:
            self.blob_free[self._last_set].record()   # the work on the previous batch is on the stream: its set is free after it
        st = self.next_raw()
        torch.cuda.current_stream().wait_event(st.built)
        self._last_set = st.blob_set
        b = {k: v for k, v in self.blob_sets[st.blob_set].items() if not k.startswith("_")}
        if not self.input_depth:
            b.pop("depth_rendered", None)   # built for the flow labels only
        return b
