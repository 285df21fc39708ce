"""Index generators vs the reference's seeded outputs (golden) and the
invariants of pyls/tests/test_base.py:14-154, pyls/tests/test_utils.py."""
import warnings

import numpy as np
import pytest

from conftest import load_golden
from pypyls_amd import resampling as rsmp


@pytest.fixture(scope='module')
def gold():
    return load_golden('indexgen')


@pytest.mark.parametrize('case', range(6))
def test_seeded_generators_match_reference(gold, case):
    tag = 'case{}'.format(case)
    groups, n_cond = list(gold[tag + '_groups']), int(gold[tag + '_n_cond'])
    np.testing.assert_array_equal(
        rsmp.gen_permsamp(groups, n_cond, 12, seed=1234, verbose=False), gold[tag + '_perm'])
    np.testing.assert_array_equal(
        rsmp.gen_bootsamp(groups, n_cond, 12, seed=1234, verbose=False), gold[tag + '_boot'])
    np.testing.assert_array_equal(
        rsmp.gen_splits(groups, n_cond, 6, seed=1234, test_size=0.5), gold[tag + '_split'])
    np.testing.assert_array_equal(
        rsmp.gen_splits(groups, n_cond, 6, seed=99, test_size=0.25), gold[tag + '_split25'])


def test_survey_known_answers():
    """SURVEY.md section 8c."""
    np.testing.assert_array_equal(rsmp.gen_permsamp([6], 1, 2, seed=1234).T,
                                  [[1, 4, 0, 5, 3, 2], [5, 1, 4, 3, 0, 2]])
    np.testing.assert_array_equal(rsmp.gen_bootsamp([6], 1, 2, seed=1234).T,
                                  [[0, 1, 3, 4, 4, 5], [1, 1, 2, 3, 4, 4]])


def test_permute_cols_known_answer(gold):
    """pyls/tests/test_utils.py:117-122."""
    out = rsmp.permute_cols(np.arange(9).reshape(3, 3), seed=np.random.RandomState(1234))
    np.testing.assert_array_equal(out, [[0, 1, 5], [6, 4, 2], [3, 7, 8]])
    np.testing.assert_array_equal(out, gold['permute_cols_kat'])
    with pytest.raises(ValueError):
        rsmp.permute_cols(np.arange(9))


def test_duplicate_warning_path(gold):
    """3 subjects have only 6 permutations: 500 tries then warn once
    (pyls/base.py:70-75)."""
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        out = rsmp.gen_permsamp([3], 1, 10, seed=1234, verbose=False)
    assert any('Duplicate permutations' in str(x.message) for x in w)
    np.testing.assert_array_equal(out, gold['dup_perm'])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        out = rsmp.gen_bootsamp([3], 1, 12, seed=1234, verbose=False)
    np.testing.assert_array_equal(out, gold['dup_boot'])


@pytest.mark.parametrize('groups,n_cond', [([10, 10], 1), ([10, 10], 2), ([10], 1), ([10], 2),
                                           ([5, 7, 6], 3)])
def test_invariants(groups, n_cond):
    S = sum(groups) * n_cond
    perm = rsmp.gen_permsamp(groups, n_cond, 10, seed=1234, verbose=False)
    assert perm.shape == (S, 10)
    assert len({tuple(c) for c in perm.T}) == 10                    # no duplicate columns
    for c in perm.T:
        np.testing.assert_array_equal(np.sort(c), np.arange(S))     # a permutation
    np.testing.assert_array_equal(perm, rsmp.gen_permsamp(groups, n_cond, 10, seed=1234))
    boot = rsmp.gen_bootsamp(groups, n_cond, 10, seed=1234, verbose=False)
    assert boot.shape == (S, 10)
    lab = rsmp.dummy_label(groups, n_cond)
    for c in boot.T:
        np.testing.assert_array_equal(lab[c], lab)                  # rows stay inside their cell
    # subjects keep all their conditions together (test_base.py:96-110)
    if n_cond > 1:
        g0 = groups[0]
        for c in boot.T:
            np.testing.assert_array_equal(c[:g0] + g0, c[g0:2 * g0])
    # >= 50% unique subjects per group (test_base.py:112-128)
    row0 = 0
    for g in groups:
        for c in boot.T:
            assert len(np.unique(c[row0:row0 + g])) >= int(np.ceil(g * 0.5))
        row0 += g * n_cond
    spl = rsmp.gen_splits(groups, n_cond, 5, seed=1234)
    assert spl.shape == (S, 5) and spl.dtype == bool
    assert len({tuple(c) for c in spl.T}) == 5
    row0 = 0
    for g in groups:
        for c in spl.T:
            assert c[row0:row0 + g].sum() in (int(np.ceil(g / 2)), int(np.floor(g / 2)))
        row0 += g * n_cond


def test_dummy_coding():
    np.testing.assert_array_equal(rsmp.dummy_label([2, 1], 2), [1, 1, 2, 2, 3, 4])
    assert rsmp.dummy_code([3, 4], 2).shape == (14, 4)
    np.testing.assert_array_equal(rsmp.cell_of_row([2], 2), [0, 0, 1, 1])
    with pytest.raises(ValueError):
        rsmp.check_random_state('nope')
