"""Estimators shared by tests/golden/make_reference_image_pins.py (applied to the reference's PNGs) and
tests/test_reference_image_pins.py (applied to our frames): the same function must measure both sides."""
import numpy as np


def box_mean(x, k):
    """Mean over the (2k + 1)^2 neighbourhood (edges replicated)."""
    n = 2 * k + 1
    c = np.cumsum(np.cumsum(np.pad(x, ((k, k), (k, k)), mode="edge"), axis=0), axis=1)
    c = np.pad(c, ((1, 0), (1, 0)))
    return (c[n:, n:] - c[:-n, n:] - c[n:, :-n] + c[:-n, :-n]) / (n * n)


CUBE_WINDOW = (250, 560, 480, 830)  # rows, columns of book2.png that hold the cluster of 1000 white spheres (world.rs:598-613)


def sphere_cube_extents(radiance_window):
    """Outline extents of the sphere cluster Translate(-100, 270, 395) o RotateY(15) o BVH(1000 spheres in [0, 165)^3, r = 10)
    in linear radiance of CUBE_WINDOW (top row first): white spheres on a dark wall.  The neighbourhood mean of the darkest
    channel, capped at 0.4 so that a firefly of a low-spp frame cannot light its neighbourhood, thresholded at 0.09 (the wall is
    ~0.02, a lit white sphere >= 0.2); an extent = first / last row or column holding >= 6 such pixels.  The marble ball hides
    the cluster's lower left, so `top` and `right` are taken right of column 560, `bottom` right of column 600 and `left` on the
    rows above the ball (280..370).  tests/test_reference_image_pins.py applies the same function to our frames."""
    r0, _, c0, _ = CUBE_WINDOW
    m = box_mean(np.minimum(radiance_window.min(axis=2), 0.4), 3) > 0.09

    def first_last(counts):
        k = np.flatnonzero(counts >= 6)
        return int(k[0]), int(k[-1])
    top = first_last(m[:, 560 - c0:].sum(axis=1))[0] + r0
    bottom = first_last(m[:, 600 - c0:].sum(axis=1))[1] + r0
    right = first_last(m.sum(axis=0))[1] + c0
    left = first_last(m[280 - r0:370 - r0, :600 - c0].sum(axis=0))[0] + c0
    return {"top_row": top, "bottom_row": bottom, "right_col": right, "left_col_above_marble": left}
