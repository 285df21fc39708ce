"""ONE ordinary call, several device contexts, no torch.distributed (pypyls_amd/team.py; include/plsx.h
plsx_comm_init_all / plsx_allgather_all): the counterpart of the reference's ``n_proc`` workers
(pyls/utils.py:252-279, pyls/base.py:286-292, 490-507, 644-650).

On a one-GPU box the team lists the device twice (``device_ids=[0, 0]``): two contexts, two host threads, the
gather as peer copies -- every line of the team path except ncclCommInitAll / ncclAllGather, which need one
device per rank and run in ``test_team_on_two_gpus`` as soon as two GPUs are visible."""
import ctypes
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(S=40, B=300, T=5, seed=0):
    rs = np.random.RandomState(seed)
    X = rs.randn(S, B)
    Y = rs.randn(S, T) + 0.4 * X[:, :T]
    return X, Y


def _same_analysis(a, b, sums_rtol=1e-11):
    """Everything that is per-resample is bit-identical between one device and a team (the same kernels see the same
    index rows); the bootstrap SUMS are added shard by shard and then in rank order instead of in one running sum, so
    standard errors / ratios agree to rounding."""
    assert np.array_equal(a.singvals, b.singvals) and np.array_equal(a.x_weights, b.x_weights)
    assert np.array_equal(a.permres.permsamples, b.permres.permsamples)
    assert np.array_equal(a.permres.perm_singval, b.permres.perm_singval)
    assert np.array_equal(a.permres.pvals, b.permres.pvals)
    assert np.array_equal(a.bootres.bootsamples, b.bootres.bootsamples)
    for key in ('y_loadings_boot', 'contrast_boot', 'y_loadings_ci', 'contrast_ci'):
        if key in a.bootres and a.bootres[key] is not None:
            assert np.array_equal(a.bootres[key], b.bootres[key]), key
    np.testing.assert_allclose(a.bootres.x_weights_stderr, b.bootres.x_weights_stderr, rtol=sums_rtol, atol=1e-300)
    np.testing.assert_allclose(a.bootres.x_weights_normed, b.bootres.x_weights_normed, rtol=1e-9, atol=1e-12)


def test_behavioral_team_of_two_contexts_equals_one_device():
    import pypyls_amd as pls
    X, Y = _data()
    kw = dict(n_perm=23, n_boot=19, n_split=3, test_split=5, seed=7, verbose=False)
    one = pls.behavioral_pls(X, Y, **kw)
    two = pls.behavioral_pls(X, Y, device_ids=[0, 0], **kw)
    _same_analysis(one, two)
    for key in ('ucorr', 'vcorr', 'ucorr_pvals', 'vcorr_pvals', 'ucorr_uplim', 'vcorr_lolim'):
        assert np.array_equal(one.splitres[key], two.splitres[key]), key
    assert np.array_equal(one.cvres.pearson_r, two.cvres.pearson_r)
    assert np.array_equal(one.cvres.r_squared, two.cvres.r_squared)
    # three ranks, uneven shards (23 = 8 + 8 + 7, 19 = 7 + 6 + 6)
    three = pls.behavioral_pls(X, Y, device_ids=[0, 0, 0], **kw)
    _same_analysis(one, three)
    # fewer resamples than ranks: a rank with an empty shard still meets the others in the collective
    few = dict(kw, n_perm=2, n_boot=2, n_split=0, test_split=0)
    _same_analysis(pls.behavioral_pls(X, Y, **few), pls.behavioral_pls(X, Y, device_ids=[0, 0, 0], **few))


def test_meancentered_and_regression_teams_equal_one_device():
    import pypyls_amd as pls
    X, Y = _data()
    Xm = np.random.RandomState(1).randn(36, 200)
    Xm[:12] += 1.0
    kw = dict(groups=[6, 6, 6], n_cond=2, n_perm=9, n_boot=11, seed=3, verbose=False)
    _same_analysis(pls.meancentered_pls(Xm, **kw), pls.meancentered_pls(Xm, device_ids=[0, 0], **kw))
    # a series long enough for the quadratic-form route of the bootstrap sums on the single device: the team's ranks
    # each close their own series -- same statistics
    kw2 = dict(kw, n_perm=4, n_boot=700)
    a, b = pls.meancentered_pls(Xm, **kw2), pls.meancentered_pls(Xm, device_ids=[0, 0], **kw2)
    live = a.singvals > 1e-8 * a.singvals.max()
    np.testing.assert_allclose(a.bootres.x_weights_normed[:, live], b.bootres.x_weights_normed[:, live], rtol=1e-7)
    assert np.array_equal(a.bootres.contrast_boot, b.bootres.contrast_boot)
    rk = dict(n_components=3, n_perm=8, n_boot=7, seed=5, verbose=False)
    ra, rb = pls.pls_regression(X, Y, **rk), pls.pls_regression(X, Y, device_ids=[0, 0], **rk)
    assert np.array_equal(ra.permres.perm_singval, rb.permres.perm_singval)
    assert np.array_equal(ra.permres.pvals, rb.permres.pvals)
    assert np.array_equal(ra.bootres.y_loadings_boot, rb.bootres.y_loadings_boot)
    np.testing.assert_allclose(ra.bootres.x_weights_normed, rb.bootres.x_weights_normed, rtol=1e-9)


def test_n_proc_counts_gpus():
    """``n_proc`` is the reference's worker count (pyls/structures.py:162-168); here a worker is a GPU.  With fewer
    GPUs than asked for the call uses what exists -- on a one-GPU box it IS the ordinary call, bit for bit; on a
    multi-GPU box the resamples are sharded over min(n_proc, GPUs) devices from this one process."""
    import torch
    import pypyls_amd as pls
    from pypyls_amd import team
    X, Y = _data()
    kw = dict(n_perm=12, n_boot=10, test_split=0, seed=11, verbose=False)
    one = pls.behavioral_pls(X, Y, **kw)
    two = pls.behavioral_pls(X, Y, n_proc=2, **kw)
    have = torch.cuda.device_count()
    assert team.resolve_devices(2) == (None if have < 2 else [torch.cuda.current_device(),
                                                              (torch.cuda.current_device() + 1) % have])
    assert team.resolve_devices(10 ** 6) == (None if have < 2 else
                                                                 [(torch.cuda.current_device() + i) % have
                                                                  for i in range(have)])
    if have < 2:
        for key in ('perm_singval', 'pvals'):
            assert np.array_equal(one.permres[key], two.permres[key])
        assert np.array_equal(one.bootres.x_weights_normed, two.bootres.x_weights_normed)
    else:
        _same_analysis(one, two)
    assert two.inputs.n_proc == 2
    with pytest.raises(ValueError, match='no GPU'):
        pls.behavioral_pls(X, Y, device_ids=[0, have], **kw)


def test_team_c_abi_allgather_all():
    """plsx_comm_init_all / plsx_allgather_all through ctypes on two contexts of the one device (transport: peer
    copies): rank r receives [send_0 | send_1]; a context cannot join a second team; RCCL refuses a device listed
    twice; plsx_comm_destroy returns a context to a world of one."""
    import torch
    from pypyls_amd import engine
    a, b = engine.Engine(0), engine.Engine(0)
    lib = a.lib
    vp = ctypes.c_void_p
    try:
        ctxs = (vp * 2)(a.ctx, b.ctx)
        assert lib.plsx_comm_init_all(ctxs, 2, 1) != 0              # PLSX_TRANSPORT_RCCL on one device twice
        assert b'device' in lib.plsx_last_error(a.ctx)
        assert lib.plsx_comm_transport(a.ctx) == 0
        assert lib.plsx_comm_init_all(ctxs, 2, 0) == 0, lib.plsx_last_error(a.ctx)
        assert lib.plsx_comm_transport(a.ctx) == 2 and lib.plsx_comm_transport(b.ctx) == 2
        assert a.comm_rank_world() == (0, 2) and b.comm_rank_world() == (1, 2)
        assert lib.plsx_comm_init_all(ctxs, 2, 0) != 0              # already ranks of a team
        n = 1000
        s0 = torch.arange(n, dtype=torch.float64, device='cuda:0')
        s1 = -torch.arange(n, dtype=torch.float64, device='cuda:0') - 0.5
        r0 = torch.zeros((2, n), dtype=torch.float64, device='cuda:0')
        r1 = torch.zeros((2, n), dtype=torch.float64, device='cuda:0')
        st = torch.cuda.current_stream().cuda_stream
        rc = lib.plsx_allgather_all(ctxs, 2, (vp * 2)(s0.data_ptr(), s1.data_ptr()), (vp * 2)(r0.data_ptr(), r1.data_ptr()),
                                    8 * n, (vp * 2)(st, st))
        assert rc == 0, lib.plsx_last_error(a.ctx)
        torch.cuda.synchronize()
        for r in (r0, r1):
            assert torch.equal(r[0], s0) and torch.equal(r[1], s1)
        # wrong order of the contexts is refused, not silently mis-gathered
        assert lib.plsx_allgather_all((vp * 2)(b.ctx, a.ctx), 2, (vp * 2)(s0.data_ptr(), s1.data_ptr()),
                                      (vp * 2)(r0.data_ptr(), r1.data_ptr()), 8 * n, None) != 0
        a.comm_destroy()
        b.comm_destroy()
        assert a.comm_rank_world() == (0, 1) and lib.plsx_comm_transport(a.ctx) == 0
    finally:
        a.close()
        b.close()


def test_a_failing_rank_does_not_hang_the_team(monkeypatch):
    """An exception on one rank's thread breaks the barrier of the collective: the call raises that exception
    instead of leaving the other rank waiting, and the team works again afterwards."""
    import pypyls_amd as pls
    from pypyls_amd import team
    X, Y = _data()
    kw = dict(n_perm=8, n_boot=8, test_split=0, seed=2, verbose=False)
    good = pls.behavioral_pls(X, Y, device_ids=[0, 0], **kw)
    t = team.team_for([0, 0])
    victim = t.engines[1]
    real = victim.boot_into

    def broken(*args, **kwargs):
        raise RuntimeError('rank 1 lost its device')
    monkeypatch.setattr(victim, 'boot_into', broken)
    with pytest.raises(RuntimeError, match='rank 1 lost its device'):
        pls.behavioral_pls(X, Y, device_ids=[0, 0], **kw)
    monkeypatch.setattr(victim, 'boot_into', real)
    again = pls.behavioral_pls(X, Y, device_ids=[0, 0], **kw)
    assert np.array_equal(good.permres.perm_singval, again.permres.perm_singval)
    assert np.array_equal(good.bootres.x_weights_normed, again.bootres.x_weights_normed)
    # bad input is refused on every rank before anything is launched
    Xbad = X.copy()
    Xbad[3, 5] = np.nan
    with pytest.raises(ValueError, match='NaN'):
        pls.behavioral_pls(Xbad, Y, device_ids=[0, 0], **kw)


def test_team_at_a_bench_sized_shape_matches_and_reports_its_transport():
    """A shape on the compact-block route (T' = 50: what c4 runs) through two contexts, against one device."""
    import pypyls_amd as pls
    from pypyls_amd import team
    X, Y = _data(S=120, B=6000, T=50, seed=3)
    kw = dict(n_perm=40, n_boot=40, test_split=0, seed=1234, verbose=False)
    one = pls.behavioral_pls(X, Y, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter('error')                      # no "RCCL unavailable" noise on the duplicate-device team
        two = pls.behavioral_pls(X, Y, device_ids=[0, 0], **kw)
    _same_analysis(one, two)
    t = team.team_for([0, 0])
    assert t.transport == 'peer' and 'hipMemcpyPeerAsync' in t.collective_name()


def test_team_on_two_gpus():
    """The real thing: two devices, ncclCommInitAll + grouped ncclAllGather, and the peer-copy transport over xGMI."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs')
    import pypyls_amd as pls
    from pypyls_amd import team
    X, Y = _data(S=60, B=2000, T=6)
    kw = dict(n_perm=30, n_boot=30, n_split=2, test_split=0, seed=5, verbose=False)
    one = pls.behavioral_pls(X, Y, **kw)
    two = pls.behavioral_pls(X, Y, n_proc=2, **kw)
    _same_analysis(one, two)
    assert team.team_for([0, 1]).transport == 'rccl'
    peer = pls.behavioral_pls(X, Y, device_ids=[0, 1], _transport='peer', **kw)
    _same_analysis(one, peer)
    assert np.array_equal(two.bootres.x_weights_normed, peer.bootres.x_weights_normed)


def test_team_with_prepermuted_y_and_3d_regression():
    """The less travelled inputs through the team path: pre-permuted Y stacks (``permindices=False``,
    pyls/base.py:636-639) with split-half, and pls_regression with a 3-D Y (subjects AND the third axis resampled,
    pyls/types/regression.py:208-235) and NaN rows -- each rank works on its slice of the stack / of the bootstraps."""
    import pypyls_amd as pls
    X, Y = _data(S=36, B=260, T=4, seed=2)
    rs = np.random.RandomState(3)
    ystack = np.stack([Y[rs.permutation(len(Y))] for _ in range(9)])
    kw = dict(n_perm=9, n_boot=8, n_split=2, test_split=0, permsamples=ystack, permindices=False, seed=4, verbose=False)
    one = pls.behavioral_pls(X, Y, **kw)
    two = pls.behavioral_pls(X, Y, device_ids=[0, 0], **kw)
    assert np.array_equal(one.permres.perm_singval, two.permres.perm_singval)
    assert np.array_equal(one.permres.pvals, two.permres.pvals)
    for key in ('ucorr', 'vcorr', 'ucorr_pvals', 'vcorr_pvals'):
        assert np.array_equal(one.splitres[key], two.splitres[key]), key
    np.testing.assert_allclose(one.bootres.x_weights_normed, two.bootres.x_weights_normed, rtol=1e-9, atol=1e-12)
    Y3 = np.stack([Y + 0.1 * rs.randn(*Y.shape) for _ in range(5)], axis=-1)
    Xn = X.copy()
    Xn[5] = np.nan                                           # an all-NaN row of X is masked (get_mask)
    rk = dict(n_components=3, n_perm=7, n_boot=9, seed=6, verbose=False)
    ra = pls.pls_regression(Xn, Y3, **rk)
    rb = pls.pls_regression(Xn, Y3, device_ids=[0, 0, 0], **rk)
    assert np.array_equal(ra.permres.perm_singval, rb.permres.perm_singval)
    assert np.array_equal(ra.bootres.y_loadings_boot, rb.bootres.y_loadings_boot)
    np.testing.assert_allclose(ra.bootres.x_weights_normed, rb.bootres.x_weights_normed, rtol=1e-9, atol=1e-12)
    assert np.isnan(ra.x_scores[5]).all() and np.isnan(rb.x_scores[5]).all()
