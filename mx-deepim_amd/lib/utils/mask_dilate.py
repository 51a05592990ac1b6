"""Random one-sided mask growth used as training / test augmentation (reference lib/utils/mask_dilate.py:10-55).

One draw picks which of the four sides may grow (`direction` in 0..9, each side is skipped for three of the ten values), then every
growing side draws its own thickness in 1..max_thickness.  A side "grows" by marking the pixels exactly `thickness` beyond an
inside->outside transition of the ORIGINAL mask along that axis -- a displaced copy of the boundary, not a filled band -- which is
what the reference computes and what is restated here.  Draws come from numpy's global RNG in the reference's order, so a seeded
run reproduces the reference's masks (tests/golden/data_golden.npz)."""
import numpy as np

# side -> (axis, +1: towards larger indices / -1: smaller, directions for which the side stays put), in the reference's draw order
_SIDES = ((0, +1, (0, 1, 4)), (0, -1, (1, 2, 5)), (1, +1, (2, 3, 6)), (1, -1, (0, 3, 7)))


def _shifted_boundary(inside, axis, sign, t):
    """pixels p with inside[p - sign*t] and not inside[p] along `axis` (both indices within the array)"""
    n = inside.shape[axis]
    out = np.zeros(inside.shape, dtype=bool)
    src = [slice(None)] * inside.ndim
    dst = [slice(None)] * inside.ndim
    src[axis], dst[axis] = (slice(0, n - t), slice(t, n)) if sign > 0 else (slice(t, n), slice(0, n - t))
    out[tuple(dst)] = inside[tuple(src)] & ~inside[tuple(dst)]
    return out


def mask_dilate(mask_origin, max_thickness=10):
    direction = np.random.randint(10)
    inside = np.asarray(mask_origin) != 0
    grown = np.array(mask_origin, copy=True)
    for axis, sign, skip in _SIDES:
        if direction in skip:
            continue
        thickness = np.random.randint(max_thickness) + 1
        grown = grown + _shifted_boundary(inside, axis, sign, thickness).astype(grown.dtype)
    grown[grown > 1] = 1
    return grown
