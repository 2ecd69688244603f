"""CPU restatement of the reference's scene compositing (TEST INFRASTRUCTURE ONLY - never imported by the product).

Follows, line by line:
  extract_cutouts        /root/reference/src/debvader/extract/extraction.py:4-43
  residual_field         /root/reference/src/debvader/deblend/field_deblender.py:46-97
  predicted_field        /root/reference/src/debvader/deblend/field_deblender.py:99-189 (one stamp list at a time)
The arithmetic that matters is scipy.ndimage.shift (third-party, scipy is installed here and on the GPU box), called
with the reference's arguments.  Pinning: extract_cutouts is checked against the reference's own function (imported
by file path in tests/golden/make_golden.py -> tests/golden/scene.npz) and against the border cases of the
reference's tests/test_extraction.py:6-62.  The two field functions have no reference test or fixture
(field_deblender.py cannot be imported here: it needs `sep`) - parity unpinned for them beyond scipy itself.
"""
import numpy as np
import scipy.ndimage


def extract_cutouts(field_image, field_size, galaxy_distances_to_center, cutout_size=59, nb_of_bands=6):
    cutout_images = np.zeros((len(galaxy_distances_to_center), cutout_size, cutout_size, nb_of_bands))
    list_idx = []
    half = int(cutout_size / 2)
    for i, d in enumerate(galaxy_distances_to_center):
        xs = -half + int(d[0]) + int(field_size / 2)
        xe = half + int(d[0]) + int(field_size / 2) + 1
        ys = -half + int(d[1]) + int(field_size / 2)
        ye = half + int(d[1]) + int(field_size / 2) + 1
        window = field_image[0, xs:xe, ys:ye]            # numpy slice semantics (negative starts wrap, ends clip)
        if window.shape != cutout_images[i].shape:       # the reference's assignment raises ValueError here
            continue
        cutout_images[i] = window
        list_idx.append(i)
    return cutout_images, list_idx


def _padded(stamp, field_size, cutout_size):
    po = int((field_size - cutout_size) / 2)
    out = np.zeros((field_size, field_size, stamp.shape[-1]))
    out[po:cutout_size + po, po:cutout_size + po, :] = stamp
    return out


def residual_field(field, stamps, positions, cutout_size):
    """field (F,F,nb) minus every stamp shifted to its position, in order."""
    out = np.array(field, dtype=np.float64, copy=True)
    for stamp, (x, y) in zip(stamps, positions):
        pad = _padded(np.asarray(stamp, np.float64), out.shape[0], cutout_size)
        for band in range(out.shape[2]):
            out[:, :, band] -= scipy.ndimage.shift(pad[:, :, band], shift=(x, y))
    return out


def predicted_field(field_size, nb_of_bands, stamps, positions, cutout_size):
    out = np.zeros((field_size, field_size, nb_of_bands))
    for stamp, (x, y) in zip(stamps, positions):
        pad = _padded(np.asarray(stamp, np.float64), field_size, cutout_size)
        for band in range(nb_of_bands):
            out[:, :, band] += scipy.ndimage.shift(pad[:, :, band], shift=(x, y))
    return out
