"""
Small host-side (numpy) finalisation steps of a PLS run -- the parts of
pyls/compute.py that touch only L- or S-sized arrays after the device has
done the resampling: p-values, percentile CIs, variance explained, the sign
convention, and the S x T x L sized score / loading products.
"""
import numpy as np


def sign_convention(x_weights, y_weights):
    """sklearn ``svd_flip`` as applied by compute.svd (pyls/compute.py:43-50):
    when T' <= B the entry of largest magnitude in each x_weights column is
    made positive, otherwise the one in each y_weights column."""
    lead = x_weights if y_weights.shape[0] <= x_weights.shape[0] else y_weights
    idx = np.argmax(np.abs(lead), axis=0)
    signs = np.sign(lead[idx, np.arange(lead.shape[1])])
    signs[signs == 0] = 1.0
    return x_weights * signs, y_weights * signs


def perm_sig(orig, perm):
    """(#{perm > orig} + 1) / (P + 1), strict '>' (pyls/compute.py:154-181).
    orig (L,), perm (L, P)."""
    orig = np.asarray(orig)
    return (np.sum(perm > orig[:, None], axis=1) + 1) / (perm.shape[-1] + 1)


def boot_ci(boot, ci=95):
    """Percentile interval over the last axis (pyls/compute.py:184-209)."""
    low = (100 - ci) / 2
    lower, upper = np.percentile(boot, [low, 100 - low], axis=-1)
    return lower, upper


def varexp(singvals):
    """s^2 / sum(s^2) (pyls/compute.py:394-414) for a 1-D vector."""
    s2 = np.asarray(singvals) ** 2
    return s2 / np.sum(s2)


def zscore_cols(A, covariance=False):
    A = np.asarray(A, dtype=float)
    Ac = A - A.mean(axis=0)
    if not covariance:
        with np.errstate(divide='ignore', invalid='ignore'):
            Ac = Ac / A.std(axis=0, ddof=1)
    return Ac


def cellwise_xcorr(scores, Y, cell_of_row, n_cells, covariance=False):
    """Stacked per-cell cross-correlation of Y columns with score columns:
    the (S, L)-sized use of gen_covcorr for ``y_loadings``
    (pyls/types/behavioral.py:195)."""
    out = []
    for c in range(n_cells):
        m = cell_of_row == c
        a, b = zscore_cols(scores[m], covariance), zscore_cols(Y[m], covariance)
        out.append(b.T @ a / (m.sum() - 1))
    return np.vstack(out)
