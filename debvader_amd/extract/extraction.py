"""Cutout extraction (reference: src/debvader/extract/extraction.py:4-43), the gather running on the GPU."""
import numpy as np

from debvader_amd import engine as E


def cutout_windows(field_size, galaxy_distances_to_center, cutout_size=59):
    """Window starts (row, column) the reference computes (extraction.py:26-30) and which of them fit the field.

    The reference slices `field_image[0, x_start:x_end, y_start:y_end]` and catches the ValueError numpy raises
    when the slice is not cutout_size x cutout_size (ends past the edge are truncated, a window straddling index 0
    is empty).
    """
    half = int(cutout_size / 2)
    try:                                             # all galaxies at once; int() truncates towards zero
        d = np.asarray(galaxy_distances_to_center, dtype=np.float64)
        if d.ndim != 2 or d.shape[1] < 2 or not np.isfinite(d).all():
            raise ValueError
        start = -half + np.trunc(d[:, :2]).astype(np.int64) + int(field_size / 2)
    except (ValueError, TypeError):
        return _cutout_windows_loop(field_size, galaxy_distances_to_center, cutout_size)
    end = start + 2 * half + 1
    inside = (start >= 0) & (end <= field_size)
    # numpy wraps negative indices: a window that lies entirely at negative indices (a galaxy more than half a field
    # beyond the low edge) is a full-size slice counted from the far edge.  The reference accepts it.
    wrapped = (start < 0) & (end < 0) & (field_size + start >= 0)
    starts = np.where(inside, start, np.where(wrapped, field_size + start, 0))
    ok = (inside | wrapped).all(axis=1) & (2 * half + 1 == cutout_size)
    return starts.astype(np.int32).reshape(-1, 2), ok


def _cutout_windows_loop(field_size, galaxy_distances_to_center, cutout_size=59):
    """cutout_windows galaxy by galaxy (the form that follows extraction.py:26-30 line by line; the fallback for inputs
    that are not a rectangular numeric table, and what tests compare the vector form with)."""
    half = int(cutout_size / 2)
    starts, ok = [], []

    def axis(start):
        end = start + 2 * half + 1
        if start >= 0 and end <= field_size:
            return start, True
        if start < 0 and end < 0 and field_size + start >= 0:
            return field_size + start, True
        return 0, False

    for d in galaxy_distances_to_center:
        xs, okx = axis(-half + int(d[0]) + int(field_size / 2))
        ys, oky = axis(-half + int(d[1]) + int(field_size / 2))
        starts.append((xs, ys))
        ok.append(okx and oky and 2 * half + 1 == cutout_size)
    return np.asarray(starts, dtype=np.int32).reshape(-1, 2), np.asarray(ok, dtype=bool)


def extract_cutouts(field_image, field_size, galaxy_distances_to_center, cutout_size=59, nb_of_bands=6, ctx=None):
    """
    Extract the cutouts around particular galaxies in the field
    parameters:
        field_image: image of the field to deblend, shape (1, field_size, field_size, nb_of_bands)
        field_size: size of the field
        galaxy_distances_to_center: distances of the galaxies to deblend from the center of the field. In pixels.
        cutout_size: size of the stamps
    returns (cutout_images float64 (N, cutout_size, cutout_size, nb_of_bands), list_idx): galaxies whose window
    leaves the field keep a zero stamp and are missing from list_idx, as in the reference.
    """
    field_image = np.asarray(field_image)
    n = len(galaxy_distances_to_center)
    cutout_images = np.zeros((n, cutout_size, cutout_size, nb_of_bands))
    if n == 0:
        return cutout_images, []
    starts, ok = cutout_windows(field_size, galaxy_distances_to_center, cutout_size)
    if field_image.ndim != 4 or field_image.shape[3] != nb_of_bands:
        ok[:] = False          # the reference's assignment raises ValueError for every galaxy (caught, flagged)
    list_idx = [int(i) for i in np.nonzero(ok)[0]]
    if list_idx:
        ctx = ctx or E.default_context()
        cutout_images[list_idx] = ctx.scene_extract(field_image[0], starts[ok], cutout_size)
    if not ok.all():
        print("Some galaxies are too close from the border of the field to be considered here.")
    return cutout_images, list_idx
