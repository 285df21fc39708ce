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
"""
import numbers
import warnings

import numpy as np


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
            warnings.warn('WARNING: Duplicate permutations used.')
            warned = True
        seen.add(key)
        out[:, i] = perminds
    return out


def gen_bootsamp(groups, n_cond, n_boot, seed=None, verbose=True):
    """(S, n_boot) bootstrap index arrays."""
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
            warnings.warn('WARNING: Duplicate bootstraps used.')
            warned = True
        for k, s in zip(keys, seen):
            s.add(k)
        out[:, i] = bootinds
    return out


def gen_splits(groups, n_cond, n_split, seed=None, test_size=0.5):
    """(S, n_split) boolean split masks (True = first half / training set)."""
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
            warnings.warn('WARNING: Duplicate split halves used.')
            warned = True
        seen.add(key)
        out[:, i] = half
    return out
