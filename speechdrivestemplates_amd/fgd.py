"""Frechet gesture distance between two sets of pose-encoder codes (reference: core/utils/fgd.py:6-64).
CPU / float64, once per validation epoch -- deliberately not a GPU kernel (a 32- or 64-dim matrix square root).

    FGD(A, B) = |mean_A - mean_B|^2 + tr(C_A) + tr(C_B) - 2 tr((C_A C_B)^(1/2))
"""
import numpy as np
from scipy import linalg


def _trace_sqrt_product(c_a, c_b, jitter=1e-6):
    root, _ = linalg.sqrtm(c_a @ c_b, disp=False)
    if not np.all(np.isfinite(root)):  # singular product: nudge both covariances off the boundary and retry
        eye = jitter * np.eye(c_a.shape[0])
        root = linalg.sqrtm((c_a + eye) @ (c_b + eye))
    return float(np.trace(np.real(root)))  # round-off can leave a tiny imaginary part; the reference drops it


def compute_fgd(feat_a, feat_b):
    """feat_* : (n_samples, dim) arrays of codes (mu, or mu ++ logvar)."""
    feat_a, feat_b = np.asarray(feat_a, dtype=np.float64), np.asarray(feat_b, dtype=np.float64)
    if feat_a.shape[1] != feat_b.shape[1]:
        raise ValueError('feature dimensions differ: %d vs %d' % (feat_a.shape[1], feat_b.shape[1]))
    gap = feat_a.mean(axis=0) - feat_b.mean(axis=0)
    c_a = np.atleast_2d(np.cov(feat_a, rowvar=False))
    c_b = np.atleast_2d(np.cov(feat_b, rowvar=False))
    return float(gap @ gap + np.trace(c_a) + np.trace(c_b) - 2.0 * _trace_sqrt_product(c_a, c_b))
