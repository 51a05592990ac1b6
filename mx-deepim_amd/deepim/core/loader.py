"""Test-time data loader: pairdb records -> device-resident test batches, double buffered.

Counterpart of the reference's `TestDataLoader` (deepim/core/loader.py:20-160): same constructor, `data_name` list, `reset()` /
iteration protocol, one batch = `batch_size` pairs.  What differs is where the bytes are turned into blobs:

  reference   worker builds float32 blobs on the host (lib/pair_matching/data_pair.py:22-72: 9.8 MB per 480x640 pair), the executor
              uploads them, one pair per GPU per step
  here        a thread pool decodes the image files into PINNED staging buffers in their file representation -- 8-bit BGR colour,
              16-bit depth: 2.1 MB per pair --, a copy stream moves them to the GPU, and dim_test_blobs_from_raw + dim_box_mask build
              image_observed / image_rendered / mask_rendered / mask_observed in HBM.  Two staging sets alternate, so the decode and the
              H2D copy of batch k+1 overlap the refinement of batch k; the compute stream waits on one event per batch.

`RawPairSource` is the seam for data that does not come from files (the synthetic pairs of bench.py's `fresh_batch` line).
The host-side reference form of the same batch is lib.pair_matching.data_pair.get_data_pair_test_batch (used by the tests to check
this path, blob by blob).  Only the shipped test configuration is built on the device path: INPUT_MASK with TEST.INIT_MASK
'box_rendered', no INPUT_DEPTH, SCALES [[480, 640]] (anything else raises, loudly).
"""
from __future__ import print_function, division

import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from lib.hip import ops


class RawPairSource(object):
    """len() pairs; fill(i, obs_bgr (H,W,3) u8, ren_bgr (H,W,3) u8, depth (H,W) u16) writes pair i's pixels into the given (pinned)
    arrays and returns (pose_rendered 3x4, class_index, pose_observed 3x4 or None)."""

    def __len__(self):
        raise NotImplementedError

    def fill(self, i, obs_bgr, ren_bgr, depth):
        raise NotImplementedError


class PairdbSource(RawPairSource):
    """pairdb records (lib/utils/image.py header) decoded with PIL"""

    def __init__(self, pairdb, config):
        self.pairdb, self.config = pairdb, config

    def __len__(self):
        return len(self.pairdb)

    def fill(self, i, obs_bgr, ren_bgr, depth):
        from lib.utils.image import imread_color, imread_unchanged

        rec = self.pairdb[i]
        if rec.get("img_flipped"):
            raise Exception("NOT_IMPLEMENTED")
        obs_bgr[...] = imread_color(rec["image_observed"])
        ren_bgr[...] = imread_color(rec["image_rendered"])
        depth[...] = imread_unchanged(rec["depth_rendered"])
        cls = self.config.dataset.class_name.index(rec["gt_class"])
        return np.asarray(rec["pose_rendered"], np.float32), cls, (np.asarray(rec["pose_observed"], np.float32) if "pose_observed" in rec else None)


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


class _Staging(object):
    """one pinned host set + its device mirror + the event that marks the H2D copies done"""

    def __init__(self, B, H, W, device):
        pin = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory()  # noqa: E731
        self.h_obs, self.h_ren = pin((B, H, W, 3), torch.uint8), pin((B, H, W, 3), torch.uint8)
        self.h_depth = pin((B, H, W), torch.uint16)
        self.h_pose, self.h_cls, self.h_gt = pin((B, 3, 4), torch.float32), pin((B,), torch.int32), pin((B, 3, 4), torch.float32)
        dev = lambda t: torch.empty_like(t, device=device)  # noqa: E731
        self.d_obs, self.d_ren, self.d_depth = dev(self.h_obs), dev(self.h_ren), dev(self.h_depth)
        self.d_pose, self.d_cls, self.d_gt = dev(self.h_pose), dev(self.h_cls), dev(self.h_gt)
        self.ready = torch.cuda.Event()
        self.consumed = torch.cuda.Event()
        self.n = 0
        self.has_gt = False

    def pairs(self):
        return ((self.h_obs, self.d_obs), (self.h_ren, self.d_ren), (self.h_depth, self.d_depth), (self.h_pose, self.d_pose),
                (self.h_cls, self.d_cls), (self.h_gt, self.d_gt))


class TestDataLoader(object):
    def __init__(self, pairdb, config, batch_size=1, shuffle=False, device="cuda:0", workers=8, source=None, height=480, width=640):
        cfg = config
        if cfg.network.INPUT_DEPTH or not cfg.network.INPUT_MASK:
            raise NotImplementedError("device loader: the shipped test configuration (INPUT_MASK, no INPUT_DEPTH)")
        if cfg.TEST.INIT_MASK != "box_rendered" or getattr(cfg.TEST, "MASK_DILATE", False):
            raise NotImplementedError("device loader: TEST.INIT_MASK 'box_rendered' without MASK_DILATE (got {})".format(cfg.TEST.INIT_MASK))
        self.source = source if source is not None else PairdbSource(pairdb, config)
        self.pairdb, self.config, self.batch_size, self.shuffle = pairdb, config, int(batch_size), shuffle
        self.size = len(self.source)
        self.index = np.arange(self.size)
        self.data_name = ["image_observed", "image_rendered", "src_pose", "class_index", "mask_observed", "mask_rendered"]
        self.label_name = None
        self.device = torch.device(device)
        self.H, self.W = height, width
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(workers)))
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.sets = [_Staging(self.batch_size, self.H, self.W, self.device) for _ in range(2)]
        self.pixel_means_bgr = np.asarray(cfg.network.PIXEL_MEANS, dtype=np.float32).reshape(3)
        self.depth_factor = float(cfg.dataset.DEPTH_FACTOR)
        B = self.batch_size
        f32 = torch.float32
        self.blobs = {"image_observed": torch.empty((B, 3, self.H, self.W), dtype=f32, device=self.device),
                      "image_rendered": torch.empty((B, 3, self.H, self.W), dtype=f32, device=self.device),
                      "mask_observed": torch.empty((B, 1, self.H, self.W), dtype=f32, device=self.device),
                      "mask_rendered": torch.empty((B, 1, self.H, self.W), dtype=f32, device=self.device)}
        self.bbox = torch.empty((B, 4), dtype=torch.int32, device=self.device)
        self._lock = threading.Lock()
        self.reset()

    def __len__(self):
        return self.size // self.batch_size   # whole batches (the resident executors are built for one batch size)

    def reset(self):
        self.cur = 0
        if self.shuffle:
            np.random.shuffle(self.index)
        self._slot = 0
        self._inflight = None

    # ---- producer side -----------------------------------------------------------------------------------------------------------
    def _stage(self, st, first):
        """decode `batch_size` pairs into the pinned set (thread pool), then enqueue the H2D copies on the copy stream"""
        ids = self.index[first:first + self.batch_size]
        st.consumed.synchronize()   # the device mirror of this set may still feed the blobs kernel of two batches ago
        ho, hr, hd = st.h_obs.numpy(), st.h_ren.numpy(), st.h_depth.numpy()

        def one(j):
            return self.source.fill(int(ids[j]), ho[j], hr[j], hd[j])

        meta = list(self.pool.map(one, range(len(ids))))
        hp, hc, hg = st.h_pose.numpy(), st.h_cls.numpy(), st.h_gt.numpy()
        st.has_gt = all(m[2] is not None for m in meta)
        for j, (pose, cls, gt) in enumerate(meta):
            hp[j], hc[j] = pose, cls
            if gt is not None:
                hg[j] = gt
        st.n = len(ids)
        with torch.cuda.stream(self.copy_stream):
            for h, d in st.pairs():
                d.copy_(h, non_blocking=True)
            st.ready.record(self.copy_stream)
        return st

    def _submit(self):
        if self.cur + self.batch_size > self.size:
            return None
        st = self.sets[self._slot]
        self._slot ^= 1
        first, self.cur = self.cur, self.cur + self.batch_size
        return self._stage(st, first)   # runs here (its decodes fan out over the pool) while the GPU still works on the batch before

    # ---- consumer side -----------------------------------------------------------------------------------------------------------
    def iter_next(self):
        return self._inflight is not None or self.cur + self.batch_size <= self.size

    def next_raw(self):
        """-> the staging set of the next batch (device mirrors valid once `ready` has been waited for); stages the batch after it"""
        st = self._inflight if self._inflight is not None else self._submit()
        if st is None:
            raise StopIteration
        self._inflight = self._submit()   # decode + upload of batch k+1 start before batch k is consumed
        return st

    def build_blobs(self, st, out=None):
        """device: raw pixels of staging set `st` -> the four float blobs (into `out`, a dict of tensors, or the loader's own)"""
        out = out if out is not None else self.blobs
        torch.cuda.current_stream().wait_event(st.ready)
        ops.test_blobs_from_raw(st.d_obs, st.d_ren, st.d_depth, self.depth_factor, self.pixel_means_bgr, out["image_observed"],
                                out["image_rendered"], out["mask_rendered"], self.bbox)
        ops.box_mask(self.bbox, out["mask_observed"])
        st.consumed.record()
        return out

    def next(self):
        st = self.next_raw()
        b = dict(self.build_blobs(st))
        b["src_pose"], b["class_index"] = st.d_pose, st.d_cls
        if st.has_gt:
            b["pose_observed"] = st.d_gt
        return b

    __next__ = next

    def __iter__(self):
        return self

    def close(self):
        self.pool.shutdown(wait=True)
