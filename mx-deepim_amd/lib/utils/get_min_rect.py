"""Bounding rectangle of a mask's non-zero pixels, reference name and return order (lib/utils/get_min_rect.py:9-23):
(x_start, y_start, x_end, y_end) with x_end / y_end the LAST non-zero column / row -- callers slice [start:end], so their
rectangles are end-exclusive and one pixel short on the right / bottom (image.py:341, data_pair.py:103-114): kept, not fixed."""
import numpy as np


def get_min_rect(mask):
    mask = np.asarray(mask)
    cols = np.flatnonzero(mask.max(axis=0))
    rows = np.flatnonzero(mask.max(axis=1))
    return cols[0], rows[0], cols[-1], rows[-1]   # IndexError-free only for non-empty masks, like the reference's np.min of empty
