"""Seeded synthetic (observed, rendered) pairs standing in for the LINEMOD / ModelNet pairdbs (no datasets offline):
procedural textured meshes, GT pose uniform on SO(3) with z ~ U(0.6, 1.2) m, initial pose = GT + noise, observed image = render
over uniform noise.  Batches are built ON the device by the HIP rasteriser (lib/utils/synthetic.py) with the reference's blob names
(deepim/core/loader.py:35-41, :164-193), one list entry per batch of `batch_pairs` pairs of this rank's shard."""
import numpy as np

from lib.dataset.evaluation import PoseEvaluator
from lib.render_hip.render_py_multi import Render_Py
from lib.utils import synthetic as syn
from lib.utils.dist_utils import even_shard_range, shard_range


class SyntheticPairs(object):
    def __init__(self, config, num_pairs, batch_pairs, seed=2333, subdiv=4, device="cuda:0", rank=0, world=1, equal_shards=False):
        """equal_shards: every rank gets floor(n_batches / world) batches (training: one collective per optimizer step, so the
        counts must agree); False = sizes differ by at most one (test / inference: no collective on the data path)."""
        self.config = config
        self.classes = list(config.dataset.class_name)
        self.models = syn.make_models(seed=seed, n_models=len(self.classes), subdiv=subdiv)
        self.K = np.asarray(config.dataset.INTRINSIC_MATRIX, dtype=np.float32).reshape(3, 3)
        self.render_machine = Render_Py(None, self.classes, self.K, zNear=config.dataset.ZNEAR, zFar=config.dataset.ZFAR, device=device,
                                        meshes=self.models)
        self.batch_pairs, self.device, self.seed = int(batch_pairs), device, seed
        n_batches = int(num_pairs) // self.batch_pairs  # whole batches only
        lo, hi = (even_shard_range if equal_shards else shard_range)(n_batches, rank, world)
        if equal_shards and hi == lo:
            raise ValueError("{} batches cannot be split over {} ranks for training".format(n_batches, world))
        self.batch_ids = list(range(lo, hi))
        self.num_pairs = len(self.batch_ids) * self.batch_pairs

    def __len__(self):
        return len(self.batch_ids)

    def evaluator(self):
        pts = {c: m[0].astype(np.float64) for c, m in zip(self.classes, self.models)}
        diam = {c: float(np.linalg.norm(p.max(0) - p.min(0))) for c, p in pts.items()}
        return PoseEvaluator(self.classes, pts, diam)

    def test_batches(self):
        # the full test graph also scores the flow head (tester.py:500-512): its labels need the two depth planes of the pair record
        with_depth = bool(self.config.network.PRED_FLOW and not self.config.TEST.FAST_TEST)
        for i in self.batch_ids:
            b = syn.build_device_batch(self.render_machine, self.batch_pairs, seed=self.seed + 1000 * (i + 1), n_classes=len(self.classes),
                                       pixel_means=self.config.network.PIXEL_MEANS, device=self.device, with_depth=with_depth)
            b["pose_observed"] = b["pose_gt"]
            yield b

    def train_batches(self, epoch):
        order = np.random.RandomState(self.seed + epoch).permutation(len(self.batch_ids)) if self.config.TRAIN.SHUFFLE else np.arange(len(self.batch_ids))
        for j in order:
            i = self.batch_ids[int(j)]
            yield syn.build_device_train_batch(self.render_machine, self.batch_pairs, seed=self.seed + 1000 * (i + 1), models=self.models,
                                               n_classes=len(self.classes), pixel_means=self.config.network.PIXEL_MEANS,
                                               npts=int(self.config.train_iter.NUM_3D_SAMPLE), device=self.device)
