"""Stand-in for OpenCV (not installed here; used only by oracle/make_golden.py when the genuine reference's topology-
optimisation evaluation block runs).  `threshold` / `connectedComponents` follow the documented OpenCV semantics
(THRESH_BINARY: dst = maxval if src > thresh else 0; connectedComponents: 8-connectivity by default, returns
(number of labels INCLUDING the background label 0, label image)) on top of scipy.ndimage - third-party arithmetic the
reference never vendors, hence "parity unpinned" for the floating-material flag (see oracle/pidm_oracle.py)."""
import numpy as np

THRESH_BINARY = 0


def threshold(src, thresh, maxval, type):
    assert type == THRESH_BINARY
    src = np.asarray(src)
    return float(thresh), np.where(src > thresh, maxval, 0).astype(src.dtype)


def connectedComponents(image, connectivity=8):
    from scipy import ndimage
    structure = np.ones((3, 3), dtype=int) if connectivity == 8 else None
    labels, n = ndimage.label(np.asarray(image) != 0, structure=structure)
    return n + 1, labels.astype(np.int32)
