"""
pypyls_amd -- MI355X-native PLS-C resampling engine behind the pyls front-ends.

    import pypyls_amd as pyls
    res = pyls.behavioral_pls(X, Y, n_perm=5000, n_boot=5000, seed=1234)

The permutation / bootstrap loops run in hand-written HIP kernels
(pypyls_amd/csrc) reached through the C ABI of include/plsx.h; there is no CPU
fallback.
"""
__version__ = '0.1.0'

from .structures import (PLSInputs, PLSResults, PLSBootResults, PLSPermResults,  # noqa: F401
                         PLSSplitHalfResults, PLSCrossValidationResults)
from .resampling import (gen_permsamp, gen_bootsamp, gen_splits, dummy_code,  # noqa: F401
                         dummy_label, permute_cols)
from .plsc import behavioral_pls, meancentered_pls  # noqa: F401
from .regression import pls_regression  # noqa: F401
from .matlab_io import import_matlab_result  # noqa: F401
from .io import save_results, load_results  # noqa: F401
from .engine import release_default_engine  # noqa: F401
