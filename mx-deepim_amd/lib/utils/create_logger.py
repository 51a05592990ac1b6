"""Log file of a run.  Same interface and on-disk layout as the reference's lib/utils/create_logger.py:12-41 --
`<root>/<yaml stem>/<image sets joined by _>/[temp_]<yaml stem>_<YYYY-mm-dd-HH-MM>.log`, root logger at INFO -- because
deepim/test.py finds the checkpoints of a training run through that directory rule (test.py:92-99).  Written for one process
per GPU: every rank calls it at the same moment, so directories are created race-free and only rank 0 owns the file."""
import logging
import os
from datetime import datetime
from pathlib import Path


def create_logger(root_output_path, cfg, image_set, temp_flie=False):
    """-> (logger, final_output_path).  `temp_flie` keeps the reference's spelling of the keyword."""
    stem = Path(cfg).name.split(".")[0]
    out_dir = Path(root_output_path) / stem / "_".join(image_set.split("+"))
    out_dir.mkdir(parents=True, exist_ok=True)
    stamp = datetime.now().strftime("%Y-%m-%d-%H-%M")
    log_path = out_dir / "{}{}_{}.log".format("temp_" if temp_flie else "", stem, stamp)
    logger = logging.getLogger()
    logger.setLevel(logging.INFO)
    if int(os.environ.get("RANK", "0")) == 0 and not any(
            isinstance(h, logging.FileHandler) and h.baseFilename == str(log_path.resolve()) for h in logger.handlers):
        handler = logging.FileHandler(str(log_path))
        handler.setFormatter(logging.Formatter("%(asctime)-15s %(message)s"))
        logger.addHandler(handler)
    return logger, str(out_dir)
