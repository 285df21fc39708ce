"""
Import of results saved by the Matlab PLS toolbox (``result`` struct of
``pls_analysis.m``) into :class:`~pypyls_amd.structures.PLSResults` -- the
counterpart of ``pyls.matlab.import_matlab_result``
(pyls/matlab/io.py:141-225; key tables :10-58).  Host-side only: lets results
produced elsewhere sit next to results of the accelerated front-ends.

The .mat file is read with ``scipy.io.loadmat(..., struct_as_record=False)``,
i.e. as attribute objects, and walked once; values keep their stored dtype
(Matlab's uint8 / uint16 / single) and lose their singleton axes.

Deliberate difference from the reference at this commit: its key table sends
``perm_splithalf.ucorr_ul`` AND ``ucorr_ll`` to ``ucorr_uplim`` and both
``vcorr_*`` limits to ``vcorr_lolim`` (io.py:27-30), so two of the four
split-half limits are lost; here ``*_ul`` -> ``*_uplim`` and ``*_ll`` ->
``*_lolim``.  Everything else is key-for-key what the reference returns.
"""
import numpy as np
import scipy.io as sio

from .structures import PLSResults

# Matlab field (dotted path inside ``result``) -> PLSResults / PLSInputs key
_FIELDS = {
    'u': 'x_weights', 's': 'singvals', 'v': 'y_weights', 'usc': 'x_scores', 'vsc': 'y_scores',
    'lvcorrs': 'y_loadings',
    'perm_result.sprob': 'pvals', 'perm_result.permsamp': 'permsamples',
    'boot_result.compare_u': 'x_weights_normed', 'boot_result.u_se': 'x_weights_stderr',
    'boot_result.bootsamp': 'bootsamples',
    'perm_splithalf.orig_ucorr': 'ucorr', 'perm_splithalf.orig_vcorr': 'vcorr',
    'perm_splithalf.ucorr_prob': 'ucorr_pvals', 'perm_splithalf.vcorr_prob': 'vcorr_pvals',
    'perm_splithalf.ucorr_ul': 'ucorr_uplim', 'perm_splithalf.vcorr_ul': 'vcorr_uplim',
    'perm_splithalf.ucorr_ll': 'ucorr_lolim', 'perm_splithalf.vcorr_ll': 'vcorr_lolim',
    'stacked_behavdata': 'Y', 'num_subj_lst': 'groups', 'num_conditions': 'n_cond',
    'perm_result.num_perm': 'n_perm', 'boot_result.num_boot': 'n_boot',
    'perm_splithalf.num_split': 'n_split', 'boot_result.clim': 'ci',
    'other_input.meancentering_type': 'mean_centering', 'method': 'method',
}
# bootstrap distribution fields depend on the analysis type (method 3 = behavioral)
_BEHAVIORAL = {'boot_result.orig_corr': 'y_loadings', 'boot_result.distrib': 'y_loadings_boot',
               'boot_result.ulcorr': ('y_loadings_ci', 1), 'boot_result.llcorr': ('y_loadings_ci', 0)}
_MEANCENTERED = {'boot_result.orig_usc': 'contrast', 'boot_result.distrib': 'contrast_boot',
                 'boot_result.ulusc': ('contrast_ci', 1), 'boot_result.llusc': ('contrast_ci', 0)}
_SUBSTRUCTS = ('boot_result', 'perm_result', 'perm_splithalf', 'other_input')


def _struct(obj):
    """The mat_struct inside a 1 x 1 struct array, else None."""
    if isinstance(obj, np.ndarray) and obj.dtype == object and obj.size == 1:
        obj = obj.reshape(-1)[0]
    return obj if isinstance(obj, sio.matlab.mat_struct) else None


def _value(v):
    """Scalars as numpy scalars of their stored dtype, arrays squeezed."""
    v = np.squeeze(np.asarray(v))
    return v.dtype.type(v) if v.ndim == 0 else v


def import_matlab_result(fname, datamat='datamat_lst'):
    """Matlab PLS ``result`` struct in ``fname`` -> PLSResults.  ``datamat``
    names the variable holding the cell array of per-group data matrices, if it
    was saved next to the result (stacked into ``inputs.X``)."""
    mat = sio.loadmat(fname, struct_as_record=False)
    result = _struct(mat.get('result'))
    if result is None:
        raise ValueError('Cannot get result struct from provided mat file')
    flat = {}
    for name in result._fieldnames:
        val = getattr(result, name)
        sub_struct = _struct(val) if name in _SUBSTRUCTS else None
        if sub_struct is not None:
            for sub in sub_struct._fieldnames:
                flat[name + '.' + sub] = getattr(sub_struct, sub)
        else:
            flat[name] = val
    method = int(np.asarray(flat.get('method', 0)).reshape(-1)[0])
    table = dict(_FIELDS)
    table.update(_BEHAVIORAL if method == 3 else _MEANCENTERED)
    out, pairs = {}, {}
    for path, val in flat.items():
        key = table.get(path)
        if key is None or _struct(val) is not None:
            continue
        if isinstance(key, tuple):                     # lower / upper limit -> stacked last axis
            pairs.setdefault(key[0], {})[key[1]] = _value(val)
        else:
            out[key] = _value(val)
    for key, lim in pairs.items():
        if 0 in lim and 1 in lim:
            out[key] = np.stack([lim[0], lim[1]], axis=-1)
    if datamat in mat:
        cells = np.asarray(mat[datamat])               # (n_groups, 1) cell array of data matrices
        groups = [np.atleast_2d(c) for c in (cells.reshape(-1) if cells.dtype == object else [cells])]
        out['X'] = np.vstack(groups)
    # Matlab indices start at one
    for key in ('bootsamples', 'permsamples'):
        if key in out:
            out[key] = out[key] - out[key].dtype.type(1)
    out.setdefault('n_split', None)
    return PLSResults(**out)
