"""Train entry point with the reference's command line (deepim/train.py:14-40):
    python deepim/train.py --cfg experiments/deepim/cfgs/<name>.yaml --gpus 0 [--frequent N]
Flow of train_net (:60-420): update_config -> logger -> symbol + init_weights (pretrained FlowNet checkpoint if present) ->
MutableModule -> metrics + Speedometer -> WarmupMultiFactorScheduler (with the resume rule) -> epochs of `fit_batch` (forward,
batch updater, backward, gradient sum over ranks, optimizer) -> epoch-end checkpoint `prefix-%04d.params` + optimizer states.
As in deepim/test.py the pairs are synthetic (`--num_pairs` per epoch); one process drives one GPU: `--gpus 0,1,2,3` starts one rank
per named GPU as a child torch.distributed.run job (gradients are summed with RCCL all-reduces per update, the kvstore='device'
semantics)."""
from __future__ import print_function, division

import _init_paths  # noqa: F401

import argparse
import os
import pprint
import sys

from deepim.config.config import config, update_config


def parse_args():
    parser = argparse.ArgumentParser(description="Train deepim network")
    parser.add_argument("--cfg", help="experiment configure file name", required=True, type=str)
    args, rest = parser.parse_known_args()
    update_config(args.cfg)
    parser.add_argument("--frequent", help="frequency of logging", default=config.default.frequent, type=int)
    parser.add_argument("--gpus", help="specify the gpu to be use", required=True, type=str)
    parser.add_argument("--temp", help="turn on visualization", action="store_true")
    parser.add_argument("--vis", help="turn on visualization", action="store_true")
    parser.add_argument("--num_pairs", help="synthetic pairs per epoch (all ranks together)", default=256, type=int)
    parser.add_argument("--max_batches", help="stop every epoch after this many batches (0 = all)", default=0, type=int)
    parser.add_argument("--from_files", help="train through the file path of the data layer: a LINEMOD-shaped synthetic dataset (PNG files, "
                        "points.xyz) is written under this directory once and read back by deepim.core.loader.TrainDataLoader "
                        "(thread-pool decode, pinned staging, blobs built on the GPU, decoded-pixel cache)", default="", type=str)
    return parser.parse_args()


def train_net(args):
    import torch

    from deepim.core import callback, metric
    from deepim.core.module import MutableModule, fit_batch
    from deepim.symbols import deepIM_flownet as symbols
    from lib.dataset.synthetic_pairs import SyntheticPairs
    from lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti
    from lib.utils.create_logger import create_logger
    from lib.utils.load_model import load_param
    from lib.utils.lr_scheduler import build_lr_schedule
    from lib.utils.save_model import save_checkpoint

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    gpu_ids = [int(i) for i in args.gpus.split(",")]
    dev_id = gpu_ids[int(os.environ.get("LOCAL_RANK", "0")) % len(gpu_ids)] if world > 1 else gpu_ids[0]
    torch.cuda.set_device(dev_id)
    device = "cuda:{}".format(dev_id)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DIM_DIST_BACKEND", "nccl")  # gloo only to rehearse several ranks on one card
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    logger, final_output_path = create_logger(config.output_path, args.cfg, config.dataset.image_set, args.temp)
    prefix = os.path.join(final_output_path, config.TRAIN.model_prefix)
    logger.info("training config:{}\n".format(pprint.pformat(config)))
    begin_epoch, end_epoch = int(config.TRAIN.begin_epoch), int(config.TRAIN.end_epoch)

    sym_instance = getattr(symbols, config.symbol)()
    sym_instance.get_symbol(config, is_train=True)
    arg_params = {}
    if config.TRAIN.RESUME and os.path.exists("%s-%04d.params" % (prefix, begin_epoch)):
        arg_params, _ = load_param(prefix, begin_epoch, convert=True)
        print("continue training from {}-{:04d}.params".format(prefix, begin_epoch))
    elif os.path.exists("%s-%04d.params" % (config.network.pretrained, config.network.pretrained_epoch)):
        arg_params, _ = load_param(config.network.pretrained, config.network.pretrained_epoch, convert=True)
        print("initialised from {}".format(config.network.pretrained))
    else:
        print("no checkpoint found ({}-{:04d}.params): seeded initialisation".format(config.network.pretrained, config.network.pretrained_epoch))
    arg_params = sym_instance.init_weights(config, arg_params, {}, seed=0)

    B = int(config.TRAIN.BATCH_PAIRS)
    data = SyntheticPairs(config, args.num_pairs, B, device=device, rank=rank, world=world, equal_shards=True)
    # DIM_TRAIN_DTYPE=bf16: convolutions on the bf16 matrix pipe (BASELINE configs[2]); DIM_OVERLAP_ALLREDUCE=0: one flat all-reduce
    # inside update() instead of three buckets overlapped with backward (same sums: tests/test_gpu_entry_points.py)
    mod = MutableModule(config, arg_params, B, device=device, compute_dtype=os.environ.get("DIM_TRAIN_DTYPE", "f32"),
                        overlap_allreduce=os.environ.get("DIM_OVERLAP_ALLREDUCE", "1") != "0")
    states = "%s-%04d.states.npz" % (prefix, begin_epoch)
    if config.TRAIN.RESUME and os.path.exists(states):
        mod.load_optimizer_states(states)
    updater = batchUpdaterPyMulti(config, 480, 640, render_machine=data.render_machine)
    file_loader = None
    if args.from_files:
        # the reference's path: pairdb records -> TrainDataLoader (train.py:136).  No dataset exists offline, so the records name the
        # files of a synthetic dataset written here once by rank 0; every rank reads its own equal share of the batches.
        import pickle

        from deepim.core.loader import PixelCache, TrainDataLoader
        from lib.dataset.synthetic_files import write_synthetic_dataset

        db_file = os.path.join(args.from_files, "pairdb_{}.pkl".format(args.num_pairs))
        if rank == 0 and not os.path.exists(db_file):
            pairdb = write_synthetic_dataset(args.from_files, data.render_machine, data.models, list(config.dataset.class_name), args.num_pairs)
            with open(db_file, "wb") as f:
                pickle.dump(pairdb, f, protocol=2)
        if dist is not None:
            dist.barrier()
        with open(db_file, "rb") as f:
            pairdb = pickle.load(f)
        per_rank = (len(pairdb) // (B * world)) * B
        pairdb = pairdb[rank * per_rank:(rank + 1) * per_rank]
        config.dataset.model_dir = os.path.join(args.from_files, "models")
        file_loader = TrainDataLoader(sym_instance, pairdb, config, batch_size=B, shuffle=bool(config.TRAIN.SHUFFLE), device=device,
                                      cache=PixelCache(device))
        print("training from files: {} pairs on this rank under {}".format(len(pairdb), args.from_files))

    eval_metrics = metric.CompositeEvalMetric()
    if config.network.PRED_FLOW:
        eval_metrics.add(metric.Flow_L2LossMetric(config))
        eval_metrics.add(metric.Flow_CurLossMetric(config))
    if config.train_iter.SE3_DIST_LOSS:   # reference train.py:293-295
        eval_metrics.add(metric.Rot_L2LossMetric(config))
        eval_metrics.add(metric.Trans_L2LossMetric(config))
    if config.train_iter.SE3_PM_LOSS:
        eval_metrics.add(metric.PointMatchingLossMetric(config))
    if config.network.PRED_MASK:
        eval_metrics.add(metric.MaskLossMetric(config))
    batch_end_callback = callback.Speedometer(B * world, frequent=args.frequent)

    # decide learning rate (train.py:318-332): len(pairdb) / batch_size iterations per epoch
    lr, lr_scheduler = build_lr_schedule(config.TRAIN.lr, config.TRAIN.lr_step, begin_epoch, args.num_pairs, B * world, config.TRAIN.warmup,
                                         config.TRAIN.warmup_lr, config.TRAIN.warmup_step)
    print("lr", lr, "lr_iters", lr_scheduler.step)

    for epoch in range(begin_epoch, end_epoch):
        eval_metrics.reset()
        if file_loader is not None:
            file_loader.reset()
        for nbatch, data_batch in enumerate(file_loader if file_loader is not None else data.train_batches(epoch)):
            if args.max_batches and nbatch >= args.max_batches:
                break
            n_iter = int(config.network.TRAIN_ITER_SIZE) if config.network.TRAIN_ITER else 1
            # one optimizer step per inner iteration (module.py:1205-1213); the scheduler sees the update count
            # the Adam branch of the reference passes only learning_rate (train.py:339): constant rate, no scheduler
            outs = fit_batch(mod, data_batch, updater, lr if str(config.TRAIN.optimizer).lower() == "adam" else lr_scheduler)
            assert len(outs) == n_iter
            for o in outs:
                eval_metrics.update(None, o)
            batch_end_callback(callback.BatchEndParam(epoch=epoch, nbatch=nbatch, eval_metric=eval_metrics, locals=None))
        if file_loader is not None and file_loader.cache is not None:
            print("pixel cache after epoch {}: {} files, {:.0f} MB, {} hits / {} misses".format(epoch, len(file_loader.cache), file_loader.cache.used / 2 ** 20,
                                                                                         file_loader.cache.hits, file_loader.cache.misses))
        names, values = eval_metrics.get()
        line = "Epoch[%d] " % epoch + " ".join("Train-%s=%f" % (n, v) for n, v in zip(names, values))
        print(line)
        logger.info(line)
        if rank == 0:  # epoch_end_callback = module_checkpoint(mod, prefix, period=1, save_optimizer_states=True)  (:314-316)
            name = save_checkpoint(prefix, epoch + 1, mod.get_params(), {})
            mod.save_optimizer_states("%s-%04d.states.npz" % (prefix, epoch + 1))
            print("saved {}".format(name))
    if dist is not None:
        # every rank applied the same summed gradient to the same weights: the replicas must be bit-identical
        digest = torch.stack([mod.flat_w.double().sum(), mod.flat_w.double().abs().sum()]).cpu()
        all_d = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(all_d, digest)
        assert all(torch.equal(all_d[0], d) for d in all_d), "parameter replicas diverged across ranks: {}".format(all_d)
        if rank == 0:
            print("replicas identical on {} ranks (digest {:.9e})".format(world, float(all_d[0][0])))
    return mod


def main():
    args = parse_args()
    print("Called with argument:", args)
    # `--gpus 0,1,2,3` in ONE command, as the reference takes it (train.py:425-438): this process starts one rank per named GPU as a
    # child torch.distributed.run job (before anything touches the GPU) and leaves with its exit code
    from lib.utils.dist_utils import launch_ranks_if_needed

    rc = launch_ranks_if_needed(len(args.gpus.split(",")), os.path.abspath(__file__), sys.argv[1:])
    if rc is not None:
        sys.exit(rc)
    train_net(args)


if __name__ == "__main__":
    main()
