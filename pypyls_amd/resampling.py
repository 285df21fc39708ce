"""
Host-side resampling index generators, draw-for-draw compatible with the
reference's legacy ``numpy.random.RandomState`` usage so that a given
``seed`` produces the same permutation / bootstrap / split arrays:

  gen_permsamp  <- pyls/base.py:10-79   (conditions shuffled within subject via
                   utils.permute_cols, pyls/utils.py:200-224, then subjects
                   permuted across groups; rejection of within-group-only and
                   duplicate permutations; 500 tries)
  gen_bootsamp  <- pyls/base.py:82-159  (sorted resampling with replacement
                   inside each group, >= ceil(min cell / 2) distinct subjects)
  gen_splits    <- pyls/base.py:162-229 (ceil/floor coin flip, draw without
                   replacement per group)

Duplicate detection uses byte-string sets instead of the reference's
O(S * n^2) array comparisons; the accept / reject decisions -- and therefore
the RNG stream -- are identical.

Two implementations with identical output: the Python loops below (``_py_*``,
the readable restatement, also the checker of the native one) and the native
generators of libplsx.so (csrc/plsx_resample.h: MT19937 + numpy's legacy
sampling algorithms in C++), used by default because the Python loops would
cap multi-GPU scaling (1 s for 10 000 + 10 000 index vectors, 30 s for the
10 000 x 100 split masks of a split-half run).  ``resampling.FORCE_PYTHON = True``
(tests) forces the Python loops.  The RandomState is handed over and back through
``get_state`` / ``set_state``, so a stream shared with other draws stays in
step with the reference's.
"""
import ctypes
import numbers
import os
import warnings

import numpy as np


FORCE_PYTHON = False          # tests: take the Python loops instead of the native generators


def _native():
    """libplsx.so for the native generators, or None (forced off / not built)."""
    if FORCE_PYTHON:
        return None
    try:
        from . import engine
        return engine._load()
    except Exception:
        return None


def _native_call(fn_name, rs, groups, n_cond, n, out, *extra, tail=()):
    """Run one native generator on the stream of ``rs`` (a RandomState)."""
    lib = _native()
    state = rs.get_state()
    if lib is None or state[0] != 'MT19937':
        return False
    key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
    pos = ctypes.c_int(int(state[2]))
    g = np.ascontiguousarray(groups, dtype=np.int32)
    rc = getattr(lib, fn_name)(g.ctypes.data, len(g), int(n_cond), int(n), *extra, key.ctypes.data,
                               ctypes.byref(pos), out.ctypes.data, *tail)
    if rc < 0:
        raise ValueError('{} failed with status {}'.format(fn_name, rc))
    rs.set_state((state[0], key, pos.value) + tuple(state[3:]))
    return rc + 1                                       # 1 = ok, 2 = duplicate limit hit


def _prefaulted(shape, dtype=np.int32):
    """Zeroed output array whose pages are already mapped: ``np.zeros`` hands out
    lazily-zeroed pages and the generator then takes one page fault per 4 KB it
    writes (measured: 65 of 97 ms for a 10 000 x 500 array); a sequential fill
    maps them at memset speed (5 ms)."""
    out = np.empty(shape, dtype=dtype)
    out.fill(0)
    return out


def check_random_state(seed):
    """None -> numpy's global RandomState, int -> new RandomState, instance ->
    itself (what sklearn.utils.check_random_state does for the reference)."""
    if seed is None or seed is np.random:
        return np.random.mtrand._rand
    if isinstance(seed, numbers.Integral):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError('{!r} cannot be used to seed a numpy.random.RandomState'.format(seed))


def dummy_label(groups, n_cond=1):
    """1-based cell label of every row: group-major, then condition, then
    subject (pyls/utils.py:178-197)."""
    groups = np.asarray(groups, dtype=int)
    return np.repeat(np.arange(len(groups) * n_cond) + 1, np.repeat(groups, n_cond))


def dummy_code(groups, n_cond=1):
    """(S, J) one-hot cell membership (pyls/utils.py:155-175)."""
    labels = dummy_label(groups, n_cond)
    return (labels[:, None] == np.unique(labels)[None, :]).astype(int)


def cell_of_row(groups, n_cond=1):
    """0-based cell index of every row (int32)."""
    return (dummy_label(groups, n_cond) - 1).astype(np.int32)


def permute_cols(x, seed=None):
    """Shuffle the entries of every column of ``x`` independently
    (pyls/utils.py:200-224): one ``random_sample(x.shape)`` draw, argsort down
    the rows."""
    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError('Expected 2D array, got {}D array instead'.format(x.ndim))
    rs = check_random_state(seed)
    order = rs.random_sample(x.shape).argsort(axis=0)
    return np.take_along_axis(x, order, axis=0)


class _Design(object):
    """Row bookkeeping shared by the three generators."""

    def __init__(self, groups, n_cond):
        self.groups = [int(g) for g in groups]
        self.n_cond = int(n_cond)
        self.n_subj = int(np.sum(self.groups))
        # rows[c, s]: row index of subject s (numbered across groups) in
        # condition c
        rows = np.zeros((self.n_cond, self.n_subj), dtype=int)
        row0, s0 = 0, 0
        self.bounds = []
        for g in self.groups:
            for c in range(self.n_cond):
                rows[c, s0:s0 + g] = row0 + c * g + np.arange(g)
            self.bounds.append((s0, s0 + g))
            row0 += g * self.n_cond
            s0 += g
        self.rows = rows
        self.n_rows = row0

    def expand(self, table):
        """(n_cond, n_subj) per-subject table -> flat per-row vector in the
        canonical row order (group, condition, subject)."""
        return np.concatenate([table[:, a:b].ravel() for a, b in self.bounds])


def gen_permsamp(groups, n_cond, n_perm, seed=None, verbose=True):
    """(S, n_perm) permutation index arrays."""
    groups = [int(g) for g in (groups if isinstance(groups, (list, tuple, np.ndarray)) else [groups])]
    rs = check_random_state(seed)
    out = _prefaulted((int(n_perm), int(sum(groups)) * int(n_cond)))
    done = _native_call('plsx_gen_permsamp', rs, groups, n_cond, n_perm, out)
    if done:
        if done == 2:
            warnings.warn('WARNING: Duplicate permutations used.')
        return out.T                                    # (S, n) int32 view: rows of `out` are the resamples
    return _py_gen_permsamp(groups, n_cond, n_perm, rs)


def _py_gen_permsamp(groups, n_cond, n_perm, seed=None, verbose=True, dup_flag=None):
    # dup_flag: a list that receives True instead of the warning (IndexStream.draw runs on a thread, where
    # warnings.catch_warnings -- process-global filter state -- must not be used)
    des = _Design(groups, n_cond)
    rs = check_random_state(seed)
    out = np.zeros((des.n_rows, n_perm), dtype=int)
    seen = set()
    warned = False
    subj = np.arange(des.n_subj, dtype=int)
    for i in range(n_perm):
        count, duplicated = 0, True
        while duplicated and count < 500:
            count, duplicated = count + 1, False
            # conditions shuffled within subject, group by group (one
            # random_sample((n_cond, n_g)) draw per group)
            shuffled = np.hstack([permute_cols(des.rows[:, a:b], seed=rs)
                                  for a, b in des.bounds])
            perm = rs.permutation(subj)
            if len(des.groups) > 1:
                for a, b in des.bounds:
                    if np.array_equal(np.sort(perm[a:b]), subj[a:b]):
                        duplicated = True
            perminds = des.expand(shuffled[:, perm])
            key = perminds.tobytes()
            if key in seen:
                duplicated = True
        if count == 500 and not warned:
            if dup_flag is not None:
                dup_flag.append(True)
            else:
                warnings.warn('WARNING: Duplicate permutations used.')
            warned = True
        seen.add(key)
        out[:, i] = perminds
    return out


def gen_bootsamp(groups, n_cond, n_boot, seed=None, verbose=True):
    """(S, n_boot) bootstrap index arrays."""
    groups = [int(g) for g in (groups if isinstance(groups, (list, tuple, np.ndarray)) else [groups])]
    rs = check_random_state(seed)
    out = _prefaulted((int(n_boot), int(sum(groups)) * int(n_cond)))
    done = _native_call('plsx_gen_bootsamp', rs, groups, n_cond, n_boot, out)
    if done:
        if done == 2:
            warnings.warn('WARNING: Duplicate bootstraps used.')
        return out.T                                    # (S, n) int32 view: rows of `out` are the resamples
    return _py_gen_bootsamp(groups, n_cond, n_boot, rs)


def _py_gen_bootsamp(groups, n_cond, n_boot, seed=None, verbose=True, dup_flag=None):
    # dup_flag: a list that receives True instead of the warning (IndexStream.draw runs on a thread, where
    # warnings.catch_warnings -- process-global filter state -- must not be used)
    des = _Design(groups, n_cond)
    rs = check_random_state(seed)
    out = np.zeros((des.n_rows, n_boot), dtype=int)
    min_subj = int(np.ceil(min(des.groups) * 0.5))
    seen = [set() for _ in des.bounds]
    warned = False
    for i in range(n_boot):
        count, duplicated = 0, True
        while duplicated and count < 500:
            count, duplicated = count + 1, False
            boot = np.zeros(des.n_subj, dtype=int)
            for a, b in des.bounds:
                members = np.arange(a, b)
                while True:
                    boot[a:b] = np.sort(rs.choice(members, size=b - a, replace=True))
                    if np.unique(boot[a:b]).size >= min_subj:
                        break
            bootinds = des.expand(des.rows[:, boot])
            # the reference compares positions [a, b) of the FLAT row vector
            # (subject numbers used as row positions, base.py:145-149)
            keys = [bootinds[a:b].tobytes() for a, b in des.bounds]
            if any(k in s for k, s in zip(keys, seen)):
                duplicated = True
        if count == 500 and not warned:
            if dup_flag is not None:
                dup_flag.append(True)
            else:
                warnings.warn('WARNING: Duplicate bootstraps used.')
            warned = True
        for k, s in zip(keys, seen):
            s.add(k)
        out[:, i] = bootinds
    return out


def gen_splits(groups, n_cond, n_split, seed=None, test_size=0.5):
    """(S, n_split) boolean split masks (True = first half / training set)."""
    groups = [int(g) for g in (groups if isinstance(groups, (list, tuple, np.ndarray)) else [groups])]
    rs = check_random_state(seed)
    out = np.zeros((int(n_split), int(sum(groups)) * int(n_cond)), dtype=np.uint8)
    done = _native_call('plsx_gen_splits', rs, groups, n_cond, n_split, out, ctypes.c_double(float(test_size)))
    if done:
        if done == 2:
            warnings.warn('WARNING: Duplicate split halves used.')
        return out.T.astype(bool)
    return _py_gen_splits(groups, n_cond, n_split, rs, test_size)


def gen_splits_seeded(groups, n_cond, n_split, seeds, test_size=0.5, rows=False, warn=True):
    """Split masks of many independent streams: element i equals
    ``gen_splits(groups, n_cond, n_split, seed=seeds[i], test_size)`` -- what
    permutation ``i`` of a split-half analysis draws (pyls/base.py:705-708,
    738-742).  Returns (len(seeds), S, n_split) bool, or with ``rows`` the
    (len(seeds), n_split, S) uint8 array the device consumes (no transposes)."""
    groups = [int(g) for g in (groups if isinstance(groups, (list, tuple, np.ndarray)) else [groups])]
    seeds = np.asarray(seeds)
    S = int(sum(groups)) * int(n_cond)
    lib = _native()
    if lib is None or seeds.size == 0 or seeds.min() < 0 or seeds.max() >= 2 ** 32:
        dup = [False]
        if seeds.size == 0:
            res = np.zeros((0, S, int(n_split)), dtype=bool)
        else:
            res = np.stack([_py_gen_splits(groups, n_cond, n_split, int(sd), test_size, dup_flag=dup) for sd in seeds])
        res = np.ascontiguousarray(res.transpose(0, 2, 1), dtype=np.uint8) if rows else res
        if not warn:                                    # (thread use: the caller warns) -- same shape as the native path
            return res, dup[0]
        if dup[0]:
            warnings.warn('WARNING: Duplicate split halves used.')
        return res
    out = np.zeros((seeds.size, int(n_split), S), dtype=np.uint8)
    g = np.ascontiguousarray(groups, dtype=np.int32)
    sd = np.ascontiguousarray(seeds, dtype=np.uint32)
    rc = lib.plsx_gen_splits_seeded(g.ctypes.data, len(g), int(n_cond), int(n_split), ctypes.c_double(float(test_size)),
                                    sd.ctypes.data, int(sd.size), out.ctypes.data)
    if rc < 0:
        raise ValueError('plsx_gen_splits_seeded failed with status {}'.format(rc))
    res = out if rows else out.transpose(0, 2, 1).astype(bool)
    if not warn:                                        # (thread use: the caller warns)
        return res, rc == 1
    if rc == 1:
        warnings.warn('WARNING: Duplicate split halves used.')
    return res


def _py_gen_splits(groups, n_cond, n_split, seed=None, test_size=0.5, dup_flag=None):
    # dup_flag: a one-element list that receives the "duplicate limit hit" bit instead of a warning (the caller
    # warns from ITS thread: warnings raised on a producer thread are lost under filters or attributed wrongly)
    des = _Design(groups, n_cond)
    rs = check_random_state(seed)
    out = np.zeros((des.n_rows, n_split), dtype=bool)
    seen = set()
    warned = False
    rounders = [np.ceil, np.floor]
    for i in range(n_split):
        count, duplicated = 0, True
        while duplicated and count < 500:
            count, duplicated = count + 1, False
            split = np.zeros(des.n_subj, dtype=bool)
            for a, b in des.bounds:
                take = rounders[rs.choice(2)]
                num = int(take((b - a) * (1 - test_size)))
                split[rs.choice(np.arange(a, b), size=num, replace=False)] = True
            half = des.expand(np.repeat(split[None], des.n_cond, axis=0))
            key = half.tobytes()
            if key in seen:
                duplicated = True
        if count == 500 and not warned:
            if dup_flag is not None:
                dup_flag[0] = True
            else:
                warnings.warn('WARNING: Duplicate split halves used.')
            warned = True
        seen.add(key)
        out[:, i] = half
    return out


# ---------------------------------------------------------------------------
# streaming generation: rows become final while later rows are still drawn
# ---------------------------------------------------------------------------

class IndexStream(object):
    """One seeded index array (``kind`` 'perm' or 'boot') drawn on a host thread
    while the caller already ships finished rows to the device.

    The reference draws a whole array before its first resample runs
    (pyls/base.py:362-397, 627-633, 468-476); the ORDER in which the RandomState
    is consumed is what a seed pins, not when the rows are used.  The duplicate
    test of both generators only looks backwards (base.py:67-69, 145-149), so rows
    ``[0, available())`` never change.  ``rows`` is (n, S) int32, one resample per
    row -- the layout the device consumes; ``samples`` the reference's (S, n)."""

    def __init__(self, kind, groups, n_cond, n):
        import threading
        if kind not in ('perm', 'boot'):
            raise ValueError(kind)
        self.kind = kind
        self.groups = [int(g) for g in (groups if isinstance(groups, (list, tuple, np.ndarray)) else [groups])]
        self.n_cond, self.n = int(n_cond), int(n)
        self.S = int(sum(self.groups)) * self.n_cond
        self.rows = _prefaulted((self.n, self.S))
        self._done = ctypes.c_int(0)
        self.finished = threading.Event()
        self.error = None
        self.duplicates = False
        self.owner = None                               # the DrawThread whose job list holds draw()

    @classmethod
    def of_array(cls, samples):
        """A stream that is complete from the start (caller-supplied (S, n) array)."""
        samples = np.asarray(samples)
        st = cls.__new__(cls)
        st.kind, st.groups, st.n_cond = 'given', None, None
        st.S, st.n = samples.shape
        st.rows = np.ascontiguousarray(samples.T, dtype=np.int32)
        st._done = ctypes.c_int(st.n)
        import threading
        st.finished = threading.Event()
        st.finished.set()
        st.error, st.duplicates, st.owner = None, False, None
        return st

    def draw(self, rs):
        """Fill ``rows`` from the stream of ``rs`` (runs on the generator thread)."""
        try:
            fn = 'plsx_gen_permsamp_stream' if self.kind == 'perm' else 'plsx_gen_bootsamp_stream'
            done = _native_call(fn, rs, self.groups, self.n_cond, self.n, self.rows,
                                tail=(ctypes.byref(self._done),))
            if not done:
                dup = []
                py = _py_gen_permsamp if self.kind == 'perm' else _py_gen_bootsamp
                self.rows[:] = py(self.groups, self.n_cond, self.n, rs, dup_flag=dup).T
                self.duplicates = bool(dup)
            else:
                self.duplicates = done == 2
            self._done.value = self.n
        except BaseException as exc:                     # surfaced by wait() on the consumer side
            self.error = exc
        finally:
            self.finished.set()

    def available(self):
        return self.n if (self.finished.is_set() and self.error is None) else int(self._done.value)

    def wait(self, upto):
        """Block until rows [0, upto) are final; returns available()."""
        import time
        upto = min(int(upto), self.n)
        while self._done.value < upto and not self.finished.is_set():
            # an EARLIER job of the generator thread failed: this stream will never be drawn
            if self.owner is not None and self.owner.error is not None:
                raise self.owner.error
            time.sleep(2e-5)
        if self.error is not None:
            raise self.error
        return self.available()

    def warn(self):
        if self.duplicates:
            warnings.warn('WARNING: Duplicate {} used.'.format(
                'permutations' if self.kind == 'perm' else 'bootstraps'))

    @property
    def samples(self):
        """(S, n) index array in the reference's layout and dtype (C-contiguous int64)."""
        self.wait(self.n)
        return np.ascontiguousarray(self.rows.T, dtype=np.int64)

    def chunks(self, lo, hi, first=256, grow=4, limit=None):
        """Row ranges [a, b) covering [lo, hi), yielded as they become final: a first small
        one so that the device starts early, then ranges ``grow`` times longer (at most
        ``limit`` rows each).  The boundaries are a function of (lo, hi, first, grow, limit)
        ONLY -- not of how far the generator thread has got -- so the launch sizes, and with
        them the kernel layouts and the floating-point summation order of the accumulated
        bootstrap sums, are the same on every run of a seed (bit-reproducible results).  A
        caller-supplied array (``of_array``) is one range."""
        pos, size = int(lo), int(first)
        if self.kind == 'given' and limit is None:
            size = max(size, int(hi) - pos)
        while pos < hi:
            step = size if limit is None else min(size, int(limit))
            want = min(hi, pos + step)
            if hi - want < step // 2 and (limit is None or hi - pos <= int(limit)):
                want = hi                               # no tiny last launch: a range of 25 rows costs
            self.wait(want)                             # the device as much as one of 250
            yield pos, want
            pos, size = want, size * grow


class DrawThread(object):
    """Runs an ordered list of draws on ONE RandomState in a host thread, in the
    reference's order (run_pls: SVD seed matrix, permutation arrays, split masks
    of the original data, bootstrap arrays, cross-validation splits --
    pyls/base.py:362-397, pyls/types/behavioral.py:219-221).  ``jobs`` are
    callables taking the RandomState; their results are read after ``join``."""

    def __init__(self, rs, jobs):
        import threading
        self.rs, self.jobs, self.error = rs, list(jobs), None
        for job in self.jobs:                           # streams learn who draws them: a failure of an
            st = getattr(job, '__self__', None)         # earlier job must not leave their consumers waiting
            if isinstance(st, IndexStream):
                st.owner = self
        self.thread = threading.Thread(target=self._run, name='plsx-draws', daemon=True)

    def start(self):
        self.thread.start()
        return self

    def _run(self):
        k = 0
        try:
            for k, job in enumerate(self.jobs):
                job(self.rs)
        except BaseException as exc:
            self.error = exc
            # the jobs behind the failed one never run: fail their streams so that a consumer blocked in
            # IndexStream.wait() (and, under torch.distributed, every rank behind it) raises instead of spinning
            for job in self.jobs[k + 1:]:
                st = getattr(job, '__self__', None)
                if isinstance(st, IndexStream) and not st.finished.is_set():
                    st.error = exc
                    st.finished.set()

    def join(self):
        self.thread.join()
        if self.error is not None:
            raise self.error


class MaskStream(object):
    """Split masks of the permutations [lo, hi) of a split-half analysis, produced block by block on a host
    thread while the device works: permutation ``i`` draws its ``n_split`` masks from a fresh
    ``RandomState(i)`` (pyls/base.py:705-708, 738-742), so blocks are independent of each other and of the
    analysis' own stream.  Round 3 generated all n_perm x n_split masks (500 MB at c4) before the first split
    ran and shipped them in one piece.  Iterating yields ``(a, b, masks)`` with ``masks`` (b - a, n_split, S)
    uint8 rows as the device consumes them; ``given``: caller-supplied (n_perm, S, n_split) boolean masks
    (tests) are sliced instead.  An exception on the producer surfaces in the consumer."""

    def __init__(self, groups, n_cond, n_split, lo, hi, block=32, depth=4, given=None, test_size=0.5):
        import queue
        import threading
        self.args = (groups, n_cond, int(n_split), float(test_size))
        self.lo, self.hi, self.block, self.given = int(lo), int(hi), int(block), given
        self.q = queue.Queue(maxsize=depth)
        self.duplicates = False
        self.thread = threading.Thread(target=self._run, name='plsx-masks', daemon=True)
        self.thread.start()

    def _run(self):
        groups, n_cond, n_split, test_size = self.args
        try:
            for a in range(self.lo, self.hi, self.block):
                b = min(self.hi, a + self.block)
                if self.given is not None:
                    m = np.ascontiguousarray(np.asarray(self.given)[a:b].transpose(0, 2, 1), dtype=np.uint8)
                else:
                    m, dup = gen_splits_seeded(groups, n_cond, n_split, np.arange(a, b), test_size=test_size,
                                               rows=True, warn=False)      # (masks, duplicate limit hit)
                    self.duplicates = self.duplicates or dup
                self.q.put((a, b, m))
            self.q.put(None)
        except BaseException as exc:
            self.q.put(exc)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    def close(self):
        """Drain and join (error paths: never leave the producer blocked on a full queue)."""
        while self.thread.is_alive():
            try:
                self.q.get(timeout=0.05)
            except Exception:
                pass
        self.thread.join()
