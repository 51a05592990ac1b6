"""lib/utils/create_logger.py:12-41 of the reference: <output_path>/<cfg name>/<image sets>/<cfg>_<time>.log"""
import logging
import os
import time


def create_logger(root_output_path, cfg, image_set, temp_flie=False):
    os.makedirs(root_output_path, exist_ok=True)  # exist_ok: several ranks create the tree at the same time
    assert os.path.exists(root_output_path), "{} does not exist".format(root_output_path)
    cfg_name = os.path.basename(cfg).split(".")[0]
    config_output_path = os.path.join(root_output_path, "{}".format(cfg_name))
    os.makedirs(config_output_path, exist_ok=True)
    image_sets = [iset for iset in image_set.split("+")]
    final_output_path = os.path.join(config_output_path, "{}".format("_".join(image_sets)))
    os.makedirs(final_output_path, exist_ok=True)
    if temp_flie:
        log_file = "temp_{}_{}.log".format(cfg_name, time.strftime("%Y-%m-%d-%H-%M"))
    else:
        log_file = "{}_{}.log".format(cfg_name, time.strftime("%Y-%m-%d-%H-%M"))
    head = "%(asctime)-15s %(message)s"
    logging.basicConfig(filename=os.path.join(final_output_path, log_file), format=head)
    logger = logging.getLogger()
    logger.setLevel(logging.INFO)
    return logger, final_output_path
