"""
ctypes binding of libplsx.so (C ABI: include/plsx.h) and a thin device-side
engine object.  PyTorch-ROCm tensors are used only as device-memory handles
(`data_ptr()`) and for host<->device copies; all arithmetic on the resampling
path runs in the hand-written HIP kernels of csrc/.

There is NO CPU fallback: without the built library or without a GPU every
entry point raises.
"""
import ctypes
import threading
import os

import numpy as np

from . import _build

_LIB = None

PLSX_BEHAVIORAL = 0
PLSX_MEANCENTERED = 1
PLSX_REGRESSION = 2
PLSX_FLAG_COVARIANCE = 1


class PlsxError(RuntimeError):
    pass


def _load():
    global _LIB
    if _LIB is not None:
        return _LIB
    # PyTorch-ROCm ships its own HIP runtime; import it first so that the
    # library binds to the runtime torch already loaded (one runtime/process).
    import torch  # noqa: F401
    path = _build.lib_path()
    if not os.path.exists(path):
        raise PlsxError(
            'libplsx.so is not built ({}). Run `python -c "import __graft_entry__ as g; '
            'g.build()"` (needs hipcc). There is no CPU fallback.'.format(path))
    lib = ctypes.CDLL(path)
    vp, i32, c_d = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    sig = {
        'plsx_version': ([], i32),
        'plsx_max_tprime': ([], i32),
        'plsx_ctx_create': ([i32, ctypes.POINTER(vp)], i32),
        'plsx_ctx_destroy': ([vp], i32),
        'plsx_last_error': ([vp], ctypes.c_char_p),
        'plsx_sync': ([vp], i32),
        'plsx_set_data': ([vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, ctypes.c_uint, vp], i32),
        'plsx_num_lv': ([vp], i32),
        'plsx_tprime': ([vp], i32),
        'plsx_crosscov_batch': ([vp, vp, vp, i32, vp, vp], i32),
        'plsx_decompose': ([vp, vp, vp, vp, vp], i32),
        'plsx_set_original': ([vp, vp, vp, vp, vp], i32),
        'plsx_project': ([vp, vp, i32, vp, vp], i32),
        'plsx_colmean': ([vp, vp, vp], i32),
        'plsx_perm_batch': ([vp, vp, i32, i32, vp, vp], i32),
        'plsx_perm_batch_y': ([vp, vp, i32, i32, vp, vp], i32),
        'plsx_crossval_batch': ([vp, vp, i32, vp, vp, vp], i32),
        'plsx_boot_batch': ([vp, vp, i32, vp, vp, vp, vp], i32),
        'plsx_boot_begin': ([vp, ctypes.c_longlong, vp], i32),
        'plsx_boot_finish': ([vp, vp, vp, vp], i32),
        'plsx_boot_route': ([vp], i32),
        'plsx_split_route': ([vp], i32),
        'plsx_split_half_batch': ([vp, vp, i32, vp, i32, vp, vp, vp], i32),
        'plsx_split_half_batch_y': ([vp, vp, vp, i32, vp, i32, vp, vp, vp], i32),
        'plsx_boot_rel': ([vp, vp, vp, vp, i32, i32, ctypes.c_longlong, vp, vp, vp], i32),
        'plsx_last_timing': ([vp, ctypes.POINTER(c_d), i32], i32),
        'plsx_set_timing': ([vp, i32], i32),
        'plsx_kernel_timing': ([vp, i32, ctypes.POINTER(c_d), ctypes.POINTER(i32)], i32),
        'plsx_kernel_class_name': ([i32], ctypes.c_char_p),
        'plsx_set_perm_path': ([vp, i32], i32),
        'plsx_mean_splits': ([vp, vp, i32, i32, i32, vp, vp], i32),
        'plsx_svd_flip': ([vp, vp, vp, vp], i32),
        'plsx_scale_columns': ([vp, vp, ctypes.c_longlong, i32, vp, vp, vp], i32),
        'plsx_transpose': ([vp, vp, i32, i32, vp, vp], i32),
        'plsx_center_rows': ([vp, vp, i32, ctypes.c_longlong, vp, vp], i32),
        'plsx_set_option': ([vp, ctypes.c_char_p, i32], i32),
        'plsx_option_name': ([i32], ctypes.c_char_p),
        'plsx_numeric_report': ([vp, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)], i32),
        'plsx_set_scratch': ([vp, c_d, i32], i32),
        'plsx_mfma_f64_peak': ([vp, ctypes.POINTER(c_d)], i32),
        'plsx_percentile_ci': ([vp, vp, ctypes.c_longlong, i32, i32, c_d, i32, c_d, vp, vp, vp], i32),
        'plsx_simpls_decompose': ([vp, vp, vp, vp, vp, vp], i32),
        'plsx_simpls_set_original': ([vp, vp, vp], i32),
        'plsx_simpls_perm_batch': ([vp, vp, i32, vp, vp], i32),
        'plsx_simpls_boot_batch': ([vp, vp, vp, i32, vp, vp, vp, vp], i32),
        'plsx_simpls_set_row_masks': ([vp, vp, vp, vp], i32),
        'plsx_gen_permsamp': ([vp, i32, i32, i32, vp, ctypes.POINTER(i32), vp], i32),
        'plsx_gen_bootsamp': ([vp, i32, i32, i32, vp, ctypes.POINTER(i32), vp], i32),
        'plsx_gen_permsamp_stream': ([vp, i32, i32, i32, vp, ctypes.POINTER(i32), vp, ctypes.POINTER(i32)], i32),
        'plsx_gen_bootsamp_stream': ([vp, i32, i32, i32, vp, ctypes.POINTER(i32), vp, ctypes.POINTER(i32)], i32),
        'plsx_gen_splits': ([vp, i32, i32, i32, c_d, vp, ctypes.POINTER(i32), vp], i32),
        'plsx_gen_splits_seeded': ([vp, i32, i32, i32, c_d, vp, i32, vp], i32),
        'plsx_comm_load': ([vp, ctypes.c_char_p], i32),
        'plsx_comm_unique_id': ([vp, vp], i32),
        'plsx_comm_init': ([vp, vp, i32, i32], i32),
        'plsx_comm_rank': ([vp, ctypes.POINTER(i32), ctypes.POINTER(i32)], i32),
        'plsx_allgather': ([vp, vp, vp, ctypes.c_longlong, vp], i32),
        'plsx_comm_destroy': ([vp], i32),
        'plsx_comm_init_all': ([ctypes.POINTER(vp), i32, i32], i32),
        'plsx_comm_transport': ([vp], i32),
        'plsx_allgather_all': ([ctypes.POINTER(vp), i32, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_longlong,
                                ctypes.POINTER(vp)], i32),
    }
    for name, (argtypes, restype) in sig.items():
        fn = getattr(lib, name)            # AttributeError if a symbol is missing
        fn.argtypes = argtypes
        fn.restype = restype
    _LIB = lib
    return lib


def exported_symbols():
    """Names declared in include/plsx.h that the loaded library exports."""
    lib = _load()
    names = ['plsx_version', 'plsx_max_tprime', 'plsx_ctx_create', 'plsx_ctx_destroy',
             'plsx_last_error', 'plsx_sync', 'plsx_set_data', 'plsx_num_lv', 'plsx_tprime',
             'plsx_crosscov_batch', 'plsx_decompose', 'plsx_set_original', 'plsx_project',
             'plsx_colmean', 'plsx_perm_batch', 'plsx_perm_batch_y', 'plsx_crossval_batch', 'plsx_boot_batch', 'plsx_boot_begin', 'plsx_boot_finish', 'plsx_boot_route', 'plsx_split_route', 'plsx_split_half_batch', 'plsx_split_half_batch_y',
             'plsx_boot_rel', 'plsx_last_timing', 'plsx_set_timing', 'plsx_kernel_timing',
             'plsx_kernel_class_name', 'plsx_set_perm_path', 'plsx_set_scratch', 'plsx_mfma_f64_peak',
             'plsx_percentile_ci', 'plsx_simpls_decompose', 'plsx_simpls_set_original', 'plsx_simpls_perm_batch',
             'plsx_simpls_boot_batch', 'plsx_simpls_set_row_masks', 'plsx_gen_permsamp', 'plsx_gen_bootsamp',
             'plsx_gen_splits', 'plsx_gen_splits_seeded', 'plsx_gen_permsamp_stream',
             'plsx_gen_bootsamp_stream', 'plsx_set_option', 'plsx_option_name', 'plsx_numeric_report',
             'plsx_svd_flip', 'plsx_scale_columns', 'plsx_transpose', 'plsx_center_rows', 'plsx_mean_splits',
             'plsx_comm_load', 'plsx_comm_unique_id', 'plsx_comm_init', 'plsx_comm_rank', 'plsx_allgather',
             'plsx_comm_destroy', 'plsx_comm_init_all', 'plsx_comm_transport', 'plsx_allgather_all']
    return [n for n in names if hasattr(lib, n)]


def _torch():
    import torch
    return torch


def option_names():
    """Keys of plsx_set_option (include/plsx.h)."""
    lib = _load()
    out, i = [], 0
    while True:
        name = lib.plsx_option_name(i)
        if not name:
            return out
        out.append(name.decode())
        i += 1


def options_from_env(environ=None):
    """``{'options': {...}, 'scratch_gb': ...}`` keyword arguments for :class:`Engine` from
    ``PLSX_<KEY>=<int>`` variables.  libplsx.so itself reads no environment variable and
    neither does the product path (``Engine()``, the front-ends): only bench.py, tools/ and
    the tests translate the environment -- so that one A/B command line or one monkeypatched
    test can select a kernel route -- through this helper."""
    environ = os.environ if environ is None else environ
    opts = {}
    for key in option_names():
        val = environ.get('PLSX_' + key.upper())
        if val is not None and val != '':
            try:
                opts[key] = int(val)
            except ValueError:
                opts[key] = 1
    kw = {'options': opts}
    if environ.get('PLSX_SCRATCH_GB'):
        kw['scratch_gb'] = float(environ['PLSX_SCRATCH_GB'])
    return kw


class GradedSpectrumWarning(UserWarning):
    """Resamples had live latent variables below 1e-5 of the largest singular value that the
    device could not refine on R (a dual-space route on data whose ORIGINAL spectrum was not graded;
    every feature-pass route refines, at every T'): their smallest LVs may miss the 1e-5 relative
    tolerance against an SVD of R (include/plsx.h, plsx_numeric_report)."""


def check_index_array(samples, S):
    """Row indices as the reference's fancy indexing ``X[inds]`` accepts them
    (pyls/base.py:569,599): integer dtype, negatives wrap like numpy's, anything
    outside [-S, S) raises IndexError.  The kernels index device memory with
    these values unchecked, so they are validated here."""
    samples = np.asarray(samples)
    if samples.dtype == bool or not np.issubdtype(samples.dtype, np.integer):
        raise IndexError('resampling arrays must hold integer row indices, got dtype {}'
                         .format(samples.dtype))
    if samples.size:
        lo, hi = int(samples.min()), int(samples.max())
        if lo < -S or hi >= S:
            raise IndexError('index {} is out of bounds for axis 0 with size {}'
                             .format(lo if lo < -S else hi, S))
        if lo < 0:
            samples = np.where(samples < 0, samples + S, samples)
    return samples


class Engine(object):
    """One device context.  All ndarray arguments / results are host numpy
    arrays unless a method says it returns a device tensor."""

    def __init__(self, device=None, scratch_gb=None, options=None):
        """``scratch_gb``: fixed super-batch scratch budget for a long-lived
        engine (steady-state throughput); None sizes the scratch per call
        (include/plsx.h, plsx_set_scratch).  ``options``: {key: int} route / layout
        switches for plsx_set_option (A/B measurements and tests; every route
        computes the same statistics)."""
        torch = _torch()
        if not torch.cuda.is_available():
            raise PlsxError('no AMD GPU visible to PyTorch-ROCm: the PLS resampling engine has '
                            'no CPU fallback')
        self.lib = _load()
        if device is None:
            device = torch.cuda.current_device()
        self.device_index = int(device)
        self.device = torch.device('cuda', self.device_index)
        ctx = ctypes.c_void_p()
        rc = self.lib.plsx_ctx_create(self.device_index, ctypes.byref(ctx))
        if rc != 0:
            raise PlsxError('plsx_ctx_create failed with status {}'.format(rc))
        self.ctx = ctx
        # one analysis at a time per context: the context is not shared across host threads (include/plsx.h);
        # the front-ends hold this lock for the whole call, so two threads that call behavioral_pls /
        # pls_regression concurrently (they share the cached default engine) run one after the other
        self.lock = threading.RLock()
        self.S = self.B = self.L = self.Tp = 0
        self.refined = self.unrefined = 0
        if scratch_gb is not None:
            self._check(self.lib.plsx_set_scratch(self.ctx, float(scratch_gb), 1))
        for key, val in (options or {}).items():
            self.set_option(key, val)

    # -- plumbing ---------------------------------------------------------
    def close(self):
        if getattr(self, 'ctx', None):
            self.lib.plsx_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = self.lib.plsx_last_error(self.ctx)
            raise PlsxError('libplsx status {}: {}'.format(rc, msg.decode() if msg else ''))

    def _stream(self):
        torch = _torch()
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, arr, dtype):
        torch = _torch()
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype))
        return t.to(self.device, non_blocking=False)

    def _empty(self, shape, dtype=None):
        torch = _torch()
        return torch.empty(shape, dtype=dtype or torch.float64, device=self.device)

    def _zeros(self, shape):
        torch = _torch()
        return torch.zeros(shape, dtype=torch.float64, device=self.device)

    def set_option(self, key, value=1):
        self._check(self.lib.plsx_set_option(self.ctx, str(key).encode(), int(value)))

    def sync(self):
        self._check(self.lib.plsx_sync(self.ctx))

    # -- the exported collective (include/plsx.h, "The collective") ---------
    def comm_load(self, path=None):
        """Bind the RCCL entry points.  Default: the librccl.so PyTorch-ROCm bundles (the copy this process
        already holds when a ``nccl`` process group exists), so that the process keeps ONE communication
        runtime."""
        if path is None:
            torch = _torch()
            cand = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
            path = cand if os.path.exists(cand) else None
        self._check(self.lib.plsx_comm_load(self.ctx, path.encode() if path else None))

    def comm_unique_id(self):
        """128 bytes for :meth:`comm_init` (rank 0 draws them, every rank receives them through the launcher's
        own rendezvous)."""
        buf = ctypes.create_string_buffer(128)
        self._check(self.lib.plsx_comm_unique_id(self.ctx, buf))
        return bytes(buf.raw)

    def comm_init(self, uid, rank, world):
        """Collective over ``world`` ranks, one per GPU (ncclCommInitRank behind the C ABI)."""
        if len(uid) != 128:
            raise PlsxError('comm_init: the unique id is 128 bytes')
        self._check(self.lib.plsx_comm_init(self.ctx, ctypes.c_char_p(bytes(uid)), int(rank), int(world)))

    def comm_rank_world(self):
        r, w = ctypes.c_int(), ctypes.c_int()
        self._check(self.lib.plsx_comm_rank(self.ctx, ctypes.byref(r), ctypes.byref(w)))
        return r.value, w.value

    def comm_destroy(self):
        self._check(self.lib.plsx_comm_destroy(self.ctx))

    def allgather_into(self, send, recv):
        """recv (world, n) <- every rank's send (n,), on the current stream (plsx_allgather); contiguous device
        tensors of one dtype.  ``send`` may be ``recv[rank]``."""
        if not (send.is_contiguous() and recv.is_contiguous()) or send.dtype != recv.dtype:
            raise PlsxError('allgather_into: contiguous tensors of one dtype')
        nbytes = send.numel() * send.element_size()
        world = self.comm_rank_world()[1]
        if recv.numel() * recv.element_size() != nbytes * world:
            raise PlsxError('allgather_into: recv holds {} bytes, world {} x {} expected'.format(
                recv.numel() * recv.element_size(), world, nbytes))
        self._check(self.lib.plsx_allgather(self.ctx, send.data_ptr(), recv.data_ptr(), nbytes, self._stream()))
        return recv

    def numeric_report(self, warn=True, stacklevel=2):
        """(refined, unrefined) resample counts of graded spectra since the last call
        (plsx_numeric_report); warns when some could not be refined (``stacklevel``: whose line the warning
        is attributed to)."""
        a, b = ctypes.c_longlong(), ctypes.c_longlong()
        self._check(self.lib.plsx_numeric_report(self.ctx, ctypes.byref(a), ctypes.byref(b)))
        self.refined += a.value
        self.unrefined += b.value
        if warn and b.value:
            import warnings
            warnings.warn('{} decomposition(s) had live latent variables below 1e-5 of the largest singular '
                          'value that could not be refined on the cross-covariance matrix (a dual-space route on '
                          'data whose original spectrum was not graded): singular values below ~6e-6 of the largest may differ from an '
                          'SVD of R by more than 1e-5 relative'.format(b.value), GradedSpectrumWarning,
                          stacklevel=stacklevel)
        return a.value, b.value

    def end_analysis(self, warn=True):
        """Close an analysis on this context, on the success AND on the error path of a front-end (it runs inside
        their ``finally``): forget the announced shard size (``expect_resamples``) and drain the refined / unrefined
        counters so that neither leaks into the next call.  It never warns itself -- a warning raised inside a
        ``finally`` would, under ``-W error``, replace the computed result: the count of unrefined
        decompositions is returned and the front-end warns AFTER its try block (:func:`warn_unrefined`), attributed to
        the user's call.  ``warn=False`` (an exception is already propagating): a failing device is not allowed to
        mask it."""
        try:
            self.set_option('expect_resamples', 0)
            return self.numeric_report(warn=False)[1]
        except PlsxError:
            if warn:
                raise
            return 0

    @staticmethod
    def warn_unrefined(n, stacklevel=3):
        """The GradedSpectrumWarning of an analysis that left ``n`` decompositions unrefined, attributed to the
        caller of the public front-end (stacklevel 3 from behavioral_pls / meancentered_pls / pls_regression's own
        frame: warn_unrefined <- run / pls_regression <- the public function <- USER)."""
        if n:
            import warnings
            warnings.warn('{} decomposition(s) had live latent variables below 1e-5 of the largest singular '
                          'value that could not be refined on the cross-covariance matrix (a dual-space route on '
                          'data whose original spectrum was not graded): singular values below ~6e-6 of the largest may differ from an '
                          'SVD of R by more than 1e-5 relative'.format(n), GradedSpectrumWarning, stacklevel=stacklevel)

    # -- data -------------------------------------------------------------
    def set_data(self, X, Y, cell_of_row, n_groups, n_cond, method, mean_centering=0,
                 covariance=False):
        """X (S, B), Y (S, T) or None: numpy arrays or device tensors."""
        torch = _torch()
        dX = X if isinstance(X, torch.Tensor) else self._dev(X, np.float64)
        dY = None
        if Y is not None:
            dY = Y if isinstance(Y, torch.Tensor) else self._dev(Y, np.float64)
        dC = self._dev(cell_of_row, np.int32)
        S, B = dX.shape
        T = 0 if dY is None else dY.shape[1]
        flags = PLSX_FLAG_COVARIANCE if covariance else 0
        self._check(self.lib.plsx_set_data(
            self.ctx, int(method), dX.data_ptr(), None if dY is None else dY.data_ptr(),
            dC.data_ptr(), S, B, T, int(n_groups), int(n_cond), int(mean_centering), flags,
            self._stream()))
        self.sync()
        self.S, self.B, self.T = S, B, T
        self.L = self.lib.plsx_num_lv(self.ctx)
        self.Tp = self.lib.plsx_tprime(self.ctx)

    def colmean(self):
        out = self._empty((self.B,))
        self._check(self.lib.plsx_colmean(self.ctx, out.data_ptr(), self._stream()))
        self.sync()
        return out.cpu().numpy()

    def _index_rows(self, samples):
        """(S, n) reference layout -> (n, S) int32 device tensor."""
        samples = np.asarray(samples)
        if samples.ndim != 2 or samples.shape[0] != self.S:
            raise ValueError('resampling array must have shape (S, n) with S = {}; got {}'
                             .format(self.S, samples.shape))
        return self._dev(check_index_array(samples, self.S).T, np.int32)

    # -- kernels ----------------------------------------------------------
    def crosscov(self, xsrc=None, ysrc=None, n=None):
        """gen_covcorr of n resamples; xsrc / ysrc (S, n) or None (identity)."""
        dx = None if xsrc is None else self._index_rows(xsrc)
        dy = None if ysrc is None else self._index_rows(ysrc)
        if n is None:
            n = (dx if dx is not None else dy).shape[0] if (dx is not None or dy is not None) else 1
        out = self._empty((n, self.Tp, self.B))
        self._check(self.lib.plsx_crosscov_batch(
            self.ctx, None if dx is None else dx.data_ptr(), None if dy is None else dy.data_ptr(),
            int(n), out.data_ptr(), self._stream()))
        self.sync()
        return out.cpu().numpy()

    def decompose(self):
        """-> x_weights (B, L), singvals (L,), y_weights (T', L), raw signs."""
        xw, sv, yw = self._empty((self.B, self.L)), self._empty((self.L,)), self._empty((self.Tp, self.L))
        self._check(self.lib.plsx_decompose(self.ctx, xw.data_ptr(), sv.data_ptr(), yw.data_ptr(),
                                            self._stream()))
        self.sync()
        return xw.cpu().numpy(), sv.cpu().numpy(), yw.cpu().numpy()

    def _as_dev(self, a):
        """numpy array or device tensor -> contiguous fp64 device tensor."""
        torch = _torch()
        if isinstance(a, torch.Tensor):
            return a if (a.is_cuda and a.is_contiguous() and a.dtype == torch.float64) else \
                a.to(self.device, dtype=torch.float64).contiguous()
        return self._dev(a, np.float64)

    def set_original(self, x_weights, singvals, y_weights):
        """Arguments: numpy arrays or device tensors (the device-resident front-end passes what
        decompose_dev / svd_flip left on the device: no host round trip of the (B, L) weights)."""
        self._orig_xw = self._as_dev(x_weights)
        sv, yw = self._as_dev(singvals), self._as_dev(y_weights)
        self._check(self.lib.plsx_set_original(self.ctx, self._orig_xw.data_ptr(), sv.data_ptr(),
                                               yw.data_ptr(), self._stream()))
        self.sync()

    # -- device-resident pieces of the front-end (tensors in, tensors out; no host copies) ------
    def colmean_dev(self):
        out = self._empty((self.B,))
        self._check(self.lib.plsx_colmean(self.ctx, out.data_ptr(), self._stream()))
        return out

    def decompose_dev(self):
        """-> x_weights (B, L), singvals (L,), y_weights (T', L) device tensors, raw signs."""
        xw, sv, yw = self._empty((self.B, self.L)), self._empty((self.L,)), self._empty((self.Tp, self.L))
        self._check(self.lib.plsx_decompose(self.ctx, xw.data_ptr(), sv.data_ptr(), yw.data_ptr(),
                                            self._stream()))
        return xw, sv, yw

    def svd_flip(self, xw, yw):
        """sklearn's svd_flip as compute.svd applies it (pyls/compute.py:43-50), in place on device tensors."""
        self._check(self.lib.plsx_svd_flip(self.ctx, xw.data_ptr(), yw.data_ptr(), self._stream()))

    def project_dev(self, W):
        """(X - colmean) @ W for a device tensor W (B, L) -> device tensor (S, L)."""
        out = self._empty((self.S, W.shape[1]))
        self._check(self.lib.plsx_project(self.ctx, W.data_ptr(), W.shape[1], out.data_ptr(), self._stream()))
        return out

    def scale_columns(self, A, scale):
        """A (rows, cols) * scale (cols,) on the device -> new tensor."""
        out = self._empty(tuple(A.shape))
        self._check(self.lib.plsx_scale_columns(self.ctx, A.data_ptr(), A.shape[0], A.shape[1], scale.data_ptr(),
                                                out.data_ptr(), self._stream()))
        return out

    def transpose_dev(self, A):
        """(rows, cols) device tensor -> (cols, rows) device tensor."""
        out = self._empty((A.shape[1], A.shape[0]))
        self._check(self.lib.plsx_transpose(self.ctx, A.data_ptr(), A.shape[0], A.shape[1], out.data_ptr(),
                                            self._stream()))
        return out

    def boot_rel_dev(self, orig, usum, usq, n_boot, add_orig=False):
        """compute.boot_rel on device tensors -> (bsr, se) device tensors, no sync."""
        bsr, se = self._empty(tuple(usum.shape)), self._empty(tuple(usum.shape))
        self._check(self.lib.plsx_boot_rel(self.ctx, orig.data_ptr(), usum.data_ptr(), usq.data_ptr(),
                                           int(n_boot), 1 if add_orig else 0, usum.numel(),
                                           bsr.data_ptr(), se.data_ptr(), self._stream()))
        return bsr, se

    def percentile_ci_dev(self, series, ci=95):
        """compute.boot_ci of (nseries, n) contiguous device series -> (lo, hi) device tensors, or None when
        n exceeds the device sort (16384): the caller falls back to numpy on the host copy."""
        n = series.shape[-1]
        if n > 16384:
            return None
        low = (100 - ci) / 2
        idx = []
        for q in (low, 100 - low):                      # numpy's virtual index of the quantile
            vi = (n - 1) * np.true_divide(q, 100)
            prev = np.floor(vi)
            idx.append((int(prev), float(vi - prev)))
        lo, hi = self._empty((series.shape[0],)), self._empty((series.shape[0],))
        self._check(self.lib.plsx_percentile_ci(self.ctx, series.data_ptr(), series.shape[0], n, idx[0][0], idx[0][1],
                                                idx[1][0], idx[1][1], lo.data_ptr(), hi.data_ptr(), self._stream()))
        return lo, hi

    def pinned_like(self, t):
        """Page-locked host tensor of the shape of device tensor ``t`` (the destination of an asynchronous D2H
        copy; 56 us per MB to map -- call it while the device is busy)."""
        torch = _torch()
        return torch.empty(tuple(t.shape), dtype=t.dtype, pin_memory=True)

    def to_host_async(self, t, pinned=None):
        """Start the D2H copy of ``t`` into page-locked memory on the current stream; the numpy view of the
        returned tensor is valid after the next sync."""
        if pinned is None:
            pinned = self.pinned_like(t)
        pinned.copy_(t, non_blocking=True)
        return pinned

    def project(self, W):
        """(X - colmean) @ W for W (B, L) -> (S, L)."""
        dW = self._dev(W, np.float64)
        out = self._empty((self.S, dW.shape[1]))
        self._check(self.lib.plsx_project(self.ctx, dW.data_ptr(), dW.shape[1], out.data_ptr(),
                                          self._stream()))
        self.sync()
        return out.cpu().numpy()

    def perm(self, permsamples, rotate=True):
        """permsamples (S, P) -> permuted singular values (L, P)."""
        idx = self._index_rows(permsamples)
        n = idx.shape[0]
        out = self._empty((n, self.L))
        self._check(self.lib.plsx_perm_batch(self.ctx, idx.data_ptr(), n, 1 if rotate else 0,
                                             out.data_ptr(), self._stream()))
        self.sync()
        return out.cpu().numpy().T.copy()

    def perm_ystack(self, ystack, rotate=True):
        """ystack (n, S, T): pre-permuted Y matrices -> permuted singular values (L, n)."""
        ystack = np.asarray(ystack, dtype=np.float64)
        if ystack.ndim != 3 or ystack.shape[1] != self.S or ystack.shape[2] != self.T:
            raise ValueError('pre-permuted Y stack must have shape (n, {}, {}); got {}'
                             .format(self.S, self.T, ystack.shape))
        d = self._dev(ystack, np.float64)
        out = self._empty((ystack.shape[0], self.L))
        self._check(self.lib.plsx_perm_batch_y(self.ctx, d.data_ptr(), ystack.shape[0],
                                               1 if rotate else 0, out.data_ptr(), self._stream()))
        self.sync()
        return out.cpu().numpy().T.copy()

    def boot(self, bootsamples, usum=None, usq=None):
        """bootsamples (S, R) -> (usum, usq) device tensors (B, L), accumulated in
        place when given, and distrib (T', L, R) numpy."""
        idx = self._index_rows(bootsamples)
        n = idx.shape[0]
        if usum is None:
            usum, usq = self._zeros((self.B, self.L)), self._zeros((self.B, self.L))
        dist = self._empty((n, self.Tp, self.L))
        self.boot_begin(n)                       # a series of one call
        self._check(self.lib.plsx_boot_batch(self.ctx, idx.data_ptr(), n, usum.data_ptr(),
                                             usq.data_ptr(), dist.data_ptr(), self._stream()))
        self.boot_finish(usum, usq)
        self.sync()
        return usum, usq, np.ascontiguousarray(dist.cpu().numpy().transpose(1, 2, 0))

    def split_half(self, masks, perms=None, ystack=None, mask_rows=False):
        """Per-split correlations.  masks (np, S, ns) bool: one (S, ns) gen_splits
        array per arrangement; perms (S, np) index array, or ystack (np, S, T)
        pre-permuted behaviour matrices, or neither (original data).
        Returns ucorr, vcorr of shape (np, L, ns)."""
        torch = _torch()
        masks = np.asarray(masks)
        if masks.ndim == 2:
            masks = masks[None]
        if mask_rows:                                   # (np, ns, S) uint8 as the device consumes them
            n_arr, ns, S = masks.shape
        else:
            n_arr, S, ns = masks.shape
        if S != self.S:
            raise ValueError('split masks must have S = {} rows'.format(self.S))
        if mask_rows:
            dm = torch.from_numpy(np.ascontiguousarray(masks, dtype=np.uint8)).to(self.device)
        else:
            dm = torch.from_numpy(np.ascontiguousarray(masks.transpose(0, 2, 1), dtype=np.uint8)).to(self.device)
        dp = None
        if perms is not None:
            dp = self._index_rows(perms)
            if dp.shape[0] != n_arr:
                raise ValueError('need one permutation per arrangement')
        dy = None
        if ystack is not None:
            dy = self._dev(ystack, np.float64)
            if tuple(dy.shape) != (n_arr, self.S, self.T) or dp is not None:
                raise ValueError('ystack must have shape ({}, {}, {}) and excludes perms'
                                 .format(n_arr, self.S, self.T))
        uc, vc = self._empty((n_arr, ns, self.L)), self._empty((n_arr, ns, self.L))
        self._check(self.lib.plsx_split_half_batch_y(
            self.ctx, None if dp is None else dp.data_ptr(), None if dy is None else dy.data_ptr(), n_arr,
            dm.data_ptr(), ns, uc.data_ptr(), vc.data_ptr(), self._stream()))
        self.sync()
        return (np.ascontiguousarray(uc.cpu().numpy().transpose(0, 2, 1)),
                np.ascontiguousarray(vc.cpu().numpy().transpose(0, 2, 1)))

    def split_route(self):
        """1 when the last split-half pass took the one-pass reader over raw first-half sums (plsx_split_route)."""
        return int(self.lib.plsx_split_route(self.ctx))

    def crossval(self, splits):
        """splits (S, m) bool, True = training row -> pearson_r, r_squared (T, m)."""
        torch = _torch()
        splits = np.asarray(splits)
        if splits.ndim != 2 or splits.shape[0] != self.S:
            raise ValueError('split masks must have shape (S, m) with S = {}'.format(self.S))
        dm = torch.from_numpy(np.ascontiguousarray(splits.T, dtype=np.uint8)).to(self.device)
        m = dm.shape[0]
        r, r2 = self._empty((m, self.T)), self._empty((m, self.T))
        self._check(self.lib.plsx_crossval_batch(self.ctx, dm.data_ptr(), m, r.data_ptr(), r2.data_ptr(),
                                                 self._stream()))
        self.sync()
        return r.cpu().numpy().T.copy(), r2.cpu().numpy().T.copy()

    # -- SIMPLS regression ---------------------------------------------------
    def set_data_regression(self, Xc, Yc, n_components):
        """Xc (S, B), Yc (S, T): globally column-centred data."""
        self.set_data(Xc, Yc, np.zeros(len(Xc), np.int32), 1, int(n_components), PLSX_REGRESSION)
        self.k = int(n_components)

    def simpls_decompose(self):
        """-> x_weights (B, k), pctvar_y (k,), cvec (T, k), y_loadings (T, k)."""
        xwT, pct = self._empty((self.k, self.B)), self._empty((self.k,))
        cv, yl = self._empty((self.T, self.k)), self._empty((self.T, self.k))
        self._check(self.lib.plsx_simpls_decompose(self.ctx, xwT.data_ptr(), pct.data_ptr(), cv.data_ptr(),
                                                   yl.data_ptr(), self._stream()))
        self.sync()
        return (np.ascontiguousarray(xwT.cpu().numpy().T), pct.cpu().numpy(), cv.cpu().numpy(),
                yl.cpu().numpy())

    def simpls_decompose_dev(self):
        """The original SIMPLS fit with everything B-sized left on the device and the sign rule of compute.svd applied
        there (plsx_svd_flip): -> x_weights (B, k) DEVICE tensor, pctvar_y (k,) numpy, y_loadings (T, k) numpy."""
        xwT, pct = self._empty((self.k, self.B)), self._empty((self.k,))
        cv, yl = self._empty((self.T, self.k)), self._empty((self.T, self.k))
        self._check(self.lib.plsx_simpls_decompose(self.ctx, xwT.data_ptr(), pct.data_ptr(), cv.data_ptr(),
                                                   yl.data_ptr(), self._stream()))
        W = self.transpose_dev(xwT)
        self.svd_flip(W, cv)
        return W, pct.cpu().numpy(), yl.cpu().numpy()

    def simpls_set_original_dev(self, W):
        """simpls_set_original for a device tensor W (B, k): transposed and column-centred on the device."""
        wT = self.transpose_dev(W)
        self._check(self.lib.plsx_center_rows(self.ctx, wT.data_ptr(), wT.shape[0], wT.shape[1], wT.data_ptr(),
                                              self._stream()))
        self._check(self.lib.plsx_simpls_set_original(self.ctx, wT.data_ptr(), self._stream()))
        self.sync()                                   # (wT is released on return)

    def simpls_set_original(self, x_weights):
        w0c = np.asarray(x_weights) - np.asarray(x_weights).mean(axis=0, keepdims=True)
        d = self._dev(w0c.T, np.float64)
        self._check(self.lib.plsx_simpls_set_original(self.ctx, d.data_ptr(), self._stream()))
        self.sync()

    def simpls_perm(self, permsamples):
        """permsamples (S, P) -> pctvar of Y per permutation (k, P)."""
        idx = self._index_rows(permsamples)
        out = self._empty((idx.shape[0], self.k))
        self._check(self.lib.plsx_simpls_perm_batch(self.ctx, idx.data_ptr(), idx.shape[0], out.data_ptr(),
                                                    self._stream()))
        self.sync()
        return out.cpu().numpy().T.copy()

    def simpls_set_row_masks(self, okx=None, oky=None):
        """okx / oky (S,) bool: usable rows of X / Y (all-NaN rows are False)."""
        torch = _torch()
        dx = None if okx is None else torch.from_numpy(np.ascontiguousarray(okx, dtype=np.uint8)).to(self.device)
        dy = None if oky is None else torch.from_numpy(np.ascontiguousarray(oky, dtype=np.uint8)).to(self.device)
        self._check(self.lib.plsx_simpls_set_row_masks(
            self.ctx, None if dx is None else dx.data_ptr(), None if dy is None else dy.data_ptr(),
            self._stream()))
        self.sync()

    def simpls_boot(self, bootsamples, usum=None, usq=None, ystack=None):
        """-> usum, usq device tensors (B, k) (accumulated in place when given)
        and y_loadings_boot (T, k, R).  ystack (R, S, T): one Y per bootstrap."""
        idx = self._index_rows(bootsamples)
        n = idx.shape[0]
        if usum is None:
            usum, usq = self._zeros((self.B, self.k)), self._zeros((self.B, self.k))
        dys = None
        if ystack is not None:
            dys = self._dev(ystack, np.float64)
            if tuple(dys.shape) != (n, self.S, self.T):
                raise ValueError('ystack must have shape ({}, {}, {})'.format(n, self.S, self.T))
        yl = self._empty((n, self.T, self.k))
        self.boot_begin(n)                       # a series of one call
        self._check(self.lib.plsx_simpls_boot_batch(
            self.ctx, idx.data_ptr(), None if dys is None else dys.data_ptr(), n, usum.data_ptr(),
            usq.data_ptr(), yl.data_ptr(), self._stream()))
        self.boot_finish(usum, usq)
        self.sync()
        return usum, usq, np.ascontiguousarray(yl.cpu().numpy().transpose(1, 2, 0))

    # -- device-resident variants (no host copies, no sync): bench / pipelines --
    def index_tensor(self, samples):
        """(S, n) host index array -> (n, S) int32 device tensor."""
        return self._index_rows(samples)

    def rows_tensor(self, rows):
        """(n, S) int32 host rows (one resample per row, already validated) -> device tensor."""
        torch = _torch()
        return torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).to(self.device)

    def perm_into(self, idx_dev, out_dev, rotate=True):
        """idx_dev (n, S) int32 device tensor, out_dev (n, L) fp64 device tensor."""
        self._check(self.lib.plsx_perm_batch(self.ctx, idx_dev.data_ptr(), idx_dev.shape[0],
                                             1 if rotate else 0, out_dev.data_ptr(), self._stream()))

    def boot_begin(self, n_total):
        """Announce a series of n_total bootstraps accumulating into one (usum, usq) (include/plsx.h,
        plsx_boot_begin).  Returns 1 when the library moved the pass over the features out of the loop (the
        quadratic-form route: the batches then leave usum / usq alone until boot_finish), 0 otherwise."""
        self._check(self.lib.plsx_boot_begin(self.ctx, int(n_total), self._stream()))
        return int(self.lib.plsx_boot_route(self.ctx))

    def boot_finish(self, usum, usq):
        """Close the series: adds what it still owes usum / usq (a no-op on the in-place route)."""
        self._check(self.lib.plsx_boot_finish(self.ctx, usum.data_ptr(), usq.data_ptr(), self._stream()))

    def boot_into(self, idx_dev, usum, usq, dist_dev):
        """idx_dev (n, S) int32; usum / usq (B, L) accumulated in place; dist_dev
        (n, T', L)."""
        self._check(self.lib.plsx_boot_batch(self.ctx, idx_dev.data_ptr(), idx_dev.shape[0],
                                             usum.data_ptr(), usq.data_ptr(), dist_dev.data_ptr(),
                                             self._stream()))

    def simpls_perm_into(self, idx_dev, out_dev):
        """idx_dev (n, S) int32, out_dev (n, k): pctvar of the permuted Y."""
        self._check(self.lib.plsx_simpls_perm_batch(self.ctx, idx_dev.data_ptr(), idx_dev.shape[0],
                                                    out_dev.data_ptr(), self._stream()))

    def simpls_boot_into(self, idx_dev, usum, usq, yl_dev, ystack=None):
        """idx_dev (n, S) int32; usum / usq (B, k) accumulated in place; yl_dev (n, T, k);
        ystack (n, S, T) device tensor: one Y per bootstrap (3-D Y), or None."""
        if ystack is not None and tuple(ystack.shape) != (idx_dev.shape[0], self.S, self.T):
            raise ValueError('ystack must have shape ({}, {}, {})'.format(idx_dev.shape[0], self.S, self.T))
        self._check(self.lib.plsx_simpls_boot_batch(self.ctx, idx_dev.data_ptr(),
                                                    None if ystack is None else ystack.data_ptr(),
                                                    idx_dev.shape[0], usum.data_ptr(), usq.data_ptr(),
                                                    yl_dev.data_ptr(), self._stream()))

    def split_half_into(self, perm_dev, masks_dev, uc_dev, vc_dev, ystack_dev=None):
        """perm_dev (np, S) int32 or None, masks_dev (np, ns, S) uint8, outputs (np, ns, L);
        ystack_dev (np, S, T): pre-permuted behaviour matrices instead of index rows."""
        n_arr, ns = masks_dev.shape[0], masks_dev.shape[1]
        self._check(self.lib.plsx_split_half_batch_y(
            self.ctx, None if perm_dev is None else perm_dev.data_ptr(),
            None if ystack_dev is None else ystack_dev.data_ptr(), n_arr, masks_dev.data_ptr(), ns,
            uc_dev.data_ptr(), vc_dev.data_ptr(), self._stream()))

    def mean_splits_into(self, corr_dev, out_dev):
        """corr_dev (np, ns, L) -> out_dev (np, L): the mean over splits of BasePLS.split_half (base.py:770)."""
        n_arr, ns, L = corr_dev.shape
        self._check(self.lib.plsx_mean_splits(self.ctx, corr_dev.data_ptr(), n_arr, ns, L, out_dev.data_ptr(),
                                              self._stream()))

    def boot_rel(self, orig, usum, usq, n_boot, add_orig=False):
        """compute.boot_rel on the device; orig numpy or tensor (B, L).  With
        add_orig the original is added back first and ``n_boot`` must already
        be R + 1 (behavioral.py:201-207)."""
        torch = _torch()
        d_orig = orig if isinstance(orig, torch.Tensor) else self._dev(orig, np.float64)
        bsr, se = self._empty(tuple(usum.shape)), self._empty(tuple(usum.shape))
        self._check(self.lib.plsx_boot_rel(self.ctx, d_orig.data_ptr(), usum.data_ptr(), usq.data_ptr(),
                                           int(n_boot), 1 if add_orig else 0, usum.numel(),
                                           bsr.data_ptr(), se.data_ptr(),
                                           self._stream()))
        self.sync()
        return bsr.cpu().numpy(), se.cpu().numpy()

    def percentile_ci(self, boot, ci=95):
        """compute.boot_ci on the device: boot (..., n) -> lower, upper (...)."""
        boot = np.ascontiguousarray(boot, dtype=np.float64)
        n = boot.shape[-1]
        if n > 16384:                                   # longer series: numpy on the host
            low = (100 - ci) / 2
            lo, hi = np.percentile(boot, [low, 100 - low], axis=-1)
            return lo, hi
        low = (100 - ci) / 2
        idx = []
        for q in (low, 100 - low):                      # numpy's virtual index of the quantile
            vi = (n - 1) * np.true_divide(q, 100)
            prev = np.floor(vi)
            idx.append((int(prev), float(vi - prev)))
        d = self._dev(boot.reshape(-1, n), np.float64)
        lo, hi = self._empty((d.shape[0],)), self._empty((d.shape[0],))
        self._check(self.lib.plsx_percentile_ci(self.ctx, d.data_ptr(), d.shape[0], n, idx[0][0], idx[0][1],
                                                idx[1][0], idx[1][1], lo.data_ptr(), hi.data_ptr(),
                                                self._stream()))
        self.sync()
        shp = boot.shape[:-1]
        return lo.cpu().numpy().reshape(shp), hi.cpu().numpy().reshape(shp)

    # -- measurement --------------------------------------------------------
    def mfma_f64_peak(self):
        """Measured fp64 MFMA TFLOP/s of this GPU (microbenchmark)."""
        v = ctypes.c_double()
        self._check(self.lib.plsx_mfma_f64_peak(self.ctx, ctypes.byref(v)))
        return v.value

    def set_timing(self, enable=True):
        self._check(self.lib.plsx_set_timing(self.ctx, 1 if enable else 0))

    def kernel_timing(self):
        """{kernel class: (summed ms, launches)} since set_timing(True)."""
        out = {}
        k = 0
        while True:
            name = self.lib.plsx_kernel_class_name(k)
            if not name:
                break
            ms, n = ctypes.c_double(), ctypes.c_int()
            self._check(self.lib.plsx_kernel_timing(self.ctx, k, ctypes.byref(ms), ctypes.byref(n)))
            if n.value:
                out[name.decode()] = (ms.value, n.value)
            k += 1
        return out

    def set_perm_path(self, dual):
        """Permutations through the S x S dual path (True) or the feature pass; None keeps
        the route and only drops the cached S x S kernel (a new analysis forms its own)."""
        rc = self.lib.plsx_set_perm_path(self.ctx, -1 if dual is None else (1 if dual else 0))
        if rc < 0:
            self._check(rc)
        return bool(rc)

    def last_timing(self):
        buf = (ctypes.c_double * 12)()
        n = self.lib.plsx_last_timing(self.ctx, buf, 12)
        keys = ['xprod_ms', 'xprod_launches', 'resamples_per_group', 'm_tiles', 'superbatch',
                'xprod_resamples', 'dual_perm', 'compact_row_fraction', 'nt_flops', 'quad_series', 'quad_m_tiles',
                'quad_blocks_per_lv']
        return {k: buf[i] for i, k in enumerate(keys[:max(n, 0)])}


# ---------------------------------------------------------------------------
# the front-ends' default engine
# ---------------------------------------------------------------------------
_DEFAULT = {}
_DEFAULT_LOCK = threading.Lock()


def default_engine(device=None, replica=0):
    """The engine the public calls use when none is passed: ONE per (process, device), created on first use and
    kept -- its context, the resident copy of the last X and the super-batch scratch stay mapped between calls.
    (``replica`` > 0: a further context on the same device, for a team that lists a device twice, team.py.)

    Why not a fresh engine per call (rounds 1-3): mapping device memory is not free on a shared MI355X.  The
    driver clears recycled VRAM lazily, so a call that maps the tens of GB its predecessor just released waits
    for the clear (25 ms per GB: 1 - 5 s measured for the c4 shape, ten times the analysis' own fixed cost),
    while a context that is re-bound (plsx_set_data) reuses what it holds.  ``release_default_engine()`` gives
    the memory back.

    **What stays mapped after a call returns**: the resident copy of X (0.8 GB at 500 x 200 000) and the
    super-batch scratch the call sized for itself (up to 48 GB) -- on purpose, see above, but it IS device memory
    that later work on the same GPU (torch or otherwise) cannot use until ``pypyls_amd.release_default_engine()``
    is called.  Thread safety: creation is guarded here, use is serialised by ``Engine.lock`` in the front-ends."""
    torch = _torch()
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    key = (int(device), int(replica))
    with _DEFAULT_LOCK:
        eng = _DEFAULT.get(key)
        if eng is None or not getattr(eng, 'ctx', None):
            if not _DEFAULT:
                import atexit
                atexit.register(release_default_engine)     # free the device memory before the runtime goes away
            eng = _DEFAULT[key] = Engine(device=key[0])
    return eng


# The cached engines keep the resident X and the super-batch scratch (up to tens of GB) mapped so that the NEXT call
# does not pay for mapping them again (25 ms per GB of recycled VRAM).  A process that has stopped calling should not
# sit on that memory for ever (a surprise-OOM for other work sharing the GPU): IDLE_RELEASE_S
# seconds after the last front-end call finished, the cached engines (and teams) nobody is using are released -- the
# next call simply builds new ones.  0 / None disables; release_default_engine() does it at once.
IDLE_RELEASE_S = 300.0
_IDLE = {'timer': None}


def touch_idle_release():
    """(Re)start the idle clock; the front-ends call this when a call ends."""
    if not IDLE_RELEASE_S:
        return
    with _DEFAULT_LOCK:
        old = _IDLE['timer']
        if old is not None:
            old.cancel()
        t = threading.Timer(float(IDLE_RELEASE_S), _idle_fire)
        t.daemon = True
        _IDLE['timer'] = t
        t.start()


def _idle_fire():
    try:
        with _DEFAULT_LOCK:
            engines = list(_DEFAULT.values())
        held = []
        for eng in engines:
            if eng.lock.acquire(blocking=False):
                held.append(eng)
        try:
            if len(held) == len(engines):               # nobody is in a call: give the memory back
                release_default_engine()
            else:
                touch_idle_release()                    # a call is running: look again later
        finally:
            for eng in held:
                eng.lock.release()
    except Exception:                                   # noqa: BLE001 -- a timer thread must never raise (interpreter exit)
        pass


def release_default_engine(device=None):
    """Destroy the cached default engine(s) and free their device memory."""
    from . import team as _team
    _team.release_teams()                                   # their communicators go before the contexts
    with _DEFAULT_LOCK:
        keys = [k for k in _DEFAULT if device is None or k[0] == int(device)]
        engines = [_DEFAULT.pop(key, None) for key in keys]
    for eng in engines:
        if eng is not None:
            with eng.lock:                                  # (a call in flight on another thread finishes first)
                eng.close()
