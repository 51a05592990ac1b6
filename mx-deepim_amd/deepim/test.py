"""Test entry point with the reference's command line (deepim/test.py:14-44):
    python deepim/test.py --cfg experiments/deepim/cfgs/<name>.yaml --gpus 0 [--ignore_cache] [--skip_flow]
Flow of test_deepim (:58-215): update_config -> logger -> symbol -> load_param(prefix, test_epoch, process=True) -> Predictor ->
pred_eval (4-iteration refinement, result cache, evaluate_pose / _add / _arp_2d).
Differences, all forced by the environment: the LINEMOD / ModelNet dataset classes are out of scope and no data is present, so the
pairs come from lib/dataset/synthetic_pairs.py (`--num_pairs`); if `<prefix>-<epoch>.params` does not exist the network runs with
the seeded initialisation (stated in the log); one process drives ONE GPU: `--gpus 0,1,2,3` starts one rank per named GPU as a
child torch.distributed.run job (or launch the script under `python -m torch.distributed.run --nproc-per-node N` yourself) and every
rank refines its shard of the pairs."""
from __future__ import print_function, division

import _init_paths  # noqa: F401

import argparse
import os
import pprint
import sys

from deepim.config.config import config, update_config


def parse_args():
    parser = argparse.ArgumentParser(description="Test a DeepIM Network")
    parser.add_argument("--cfg", help="experiment configure file name", required=True, type=str)
    args, rest = parser.parse_known_args()
    update_config(args.cfg)
    parser.add_argument("--vis", help="turn on visualization", action="store_true")
    parser.add_argument("--vis_video", help="turn on video visualization", action="store_true")
    parser.add_argument("--vis_video_zoom", help="turn on zoom video visualization", action="store_true")
    parser.add_argument("--ignore_cache", help="ignore cached pose prediction results", action="store_true")
    parser.add_argument("--gpus", help="specify the gpu to be use", required=True, type=str)
    parser.add_argument("--skip_flow", help="whether skip flow during test", action="store_true")
    parser.add_argument("--num_pairs", help="synthetic pairs to refine (all ranks together)", default=64, type=int)
    return parser.parse_args()


def test_deepim(args):
    import torch

    from deepim.core.tester import Predictor, Refiner, pred_eval
    from deepim.symbols import deepIM_flownet as symbols
    from lib.dataset.synthetic_pairs import SyntheticPairs
    from lib.utils.create_logger import create_logger
    from lib.utils.load_model import load_param

    if args.vis or args.vis_video or args.vis_video_zoom:
        raise NotImplementedError("visualisation is outside the refinement path built here (SURVEY.md 8: out of scope)")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    gpu_ids = [int(i) for i in args.gpus.split(",")]
    dev_id = gpu_ids[int(os.environ.get("LOCAL_RANK", "0")) % len(gpu_ids)] if world > 1 else gpu_ids[0]
    if world == 1 and len(gpu_ids) > 1:
        print("note: one process drives one GPU; using gpu {} (launch under torch.distributed.run for {})".format(dev_id, args.gpus))
    torch.cuda.set_device(dev_id)
    device = "cuda:{}".format(dev_id)
    if world > 1:
        # ranks refine disjoint shards with no data-path collective; the per-class pose lists are merged once at the end
        # (pred_eval, all_gather_object) so that ADD / ARP-2D / pose accuracy are over the whole test set -- host objects: gloo
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    epoch = config.TEST.test_epoch
    image_set = config.dataset.test_image_set
    logger, final_output_path = create_logger(config.output_path, args.cfg, image_set)
    prefix = os.path.join(final_output_path, "..", "_".join([iset for iset in config.dataset.image_set.split("+")]), config.TRAIN.model_prefix)
    logger.info("testing config:{}\n".format(pprint.pformat(config)))

    sym_instance = getattr(symbols, config.symbol)()
    sym_instance.get_symbol(config, is_train=False)
    param_file = "%s-%04d.params" % (prefix, epoch)
    if os.path.exists(param_file):
        arg_params, aux_params = load_param(prefix, epoch, process=True)
        print("loaded {}".format(param_file))
    else:
        arg_params = {}
        msg = "{} not found: running with the seeded initialisation (throughput / plumbing run, poses will not converge)".format(param_file)
        print(msg)
        logger.info(msg)
    arg_params = sym_instance.init_weights(config, arg_params, {}, seed=0)

    B = int(config.TEST.BATCH_PAIRS)
    data = SyntheticPairs(config, args.num_pairs, B, device=device, rank=rank, world=world)
    predictor = Predictor(config, arg_params, B, device=device)
    refiner = Refiner(config, predictor, data.render_machine, B, capture_graph=True)
    result_file = os.path.join(final_output_path, "{}_results.pkl".format(image_set))
    out = pred_eval(config, refiner, data.test_batches(), data.evaluator(), result_file=result_file, logger=logger)
    print("refined {} pairs x {} iterations on {}; result cache: {}".format(data.num_pairs, config.TEST.test_iter, device, result_file))
    print(args.cfg, config.TEST.test_epoch)
    if world > 1:
        torch.distributed.destroy_process_group()
    return out


def main():
    args = parse_args()
    print(args)
    # `--gpus 0,1,2,3` in ONE command, as the reference takes it (train.py:425-438): this process starts one rank per named GPU as a
    # child torch.distributed.run job (before anything touches the GPU) and leaves with its exit code
    from lib.utils.dist_utils import launch_ranks_if_needed

    rc = launch_ranks_if_needed(len(args.gpus.split(",")), os.path.abspath(__file__), sys.argv[1:])
    if rc is not None:
        sys.exit(rc)
    test_deepim(args)


if __name__ == "__main__":
    main()
