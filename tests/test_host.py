"""CPU-only host logic: C-ABI library loads and exports every declared
symbol, containers, finalisation helpers, multi-rank collection (gloo)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import cpu_ref as ref



def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port

def test_library_exports_every_declared_symbol():
    from pypyls_amd import _build, engine
    _build.build()
    hdr = open(os.path.join(ROOT, 'include', 'plsx.h')).read()
    declared = set(re.findall(r'\b(plsx_[a-z_]+)\s*\(', hdr))
    assert len(declared) >= 18
    exported = set(engine.exported_symbols())
    assert declared <= exported, declared - exported
    lib = engine._load()
    assert lib.plsx_version() >= 1000
    assert lib.plsx_max_tprime() == 1280


def test_engine_fails_loudly_without_gpu():
    import torch
    from pypyls_amd import engine
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(engine.PlsxError):
        engine.Engine()
    import pypyls_amd as pls
    with pytest.raises(engine.PlsxError):
        pls.behavioral_pls(np.random.rand(10, 4), np.random.rand(10, 2), n_perm=2, n_boot=2,
                           test_split=0)


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, 'pypyls_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert 'oracle' not in src, f


def test_structures():
    """pyls/tests/test_structures.py:9-33, test_utils.py ResDict semantics."""
    from pypyls_amd import PLSInputs, PLSResults
    inp = PLSInputs(X=1, Y=2, n_split=0, test_split=0, n_proc=-1, notakey=5)
    assert 'notakey' not in inp and inp.n_split is None and inp.test_split is None
    assert inp.n_proc == (os.cpu_count() or 1)
    assert PLSInputs(n_proc='max').n_proc == (os.cpu_count() or 1)
    assert PLSInputs(n_proc=-2).n_proc == (os.cpu_count() or 1) - 1
    with pytest.raises(ValueError):
        PLSInputs(test_size=1)
    with pytest.raises(ValueError):
        PLSInputs(test_size=-0.5)
    res = PLSResults(x_weights=np.ones(3), bogus=1)
    assert 'bogus' not in res
    res['alsobogus'] = 2
    assert 'alsobogus' not in res
    assert res == PLSResults(x_weights=np.ones(3))
    assert res != PLSResults(x_weights=np.zeros(3))
    assert str(res) == 'PLSResults(x_weights)'
    for sub in ('permres', 'bootres', 'splitres', 'cvres', 'inputs'):
        assert sub in res


def test_hostmath_against_oracle():
    from pypyls_amd import hostmath
    rs = np.random.RandomState(0)
    d = np.sort(rs.rand(5))[::-1]
    perm = rs.rand(5, 40)
    np.testing.assert_array_equal(hostmath.perm_sig(d, perm), ref.perm_sig(np.diag(d), perm))
    np.testing.assert_allclose(hostmath.varexp(d), np.diag(ref.varexp(np.diag(d))))
    b = rs.rand(3, 4, 50)
    for a, c in zip(hostmath.boot_ci(b, 90), ref.boot_ci(b, 90)):
        np.testing.assert_array_equal(a, c)
    xw, yw = rs.randn(20, 4), rs.randn(4, 4)
    a, b2 = hostmath.sign_convention(xw, yw)
    assert np.all(a[np.argmax(np.abs(a), 0), range(4)] > 0)
    xw, yw = rs.randn(3, 3), rs.randn(7, 3)
    a, b2 = hostmath.sign_convention(xw, yw)
    assert np.all(b2[np.argmax(np.abs(b2), 0), range(3)] > 0)
    X, Y = rs.randn(30, 4), rs.randn(30, 3)
    cells = ref.dummy_label([8, 7], 2) - 1
    spec = ref.Spec('behavioral', [8, 7], 2)
    np.testing.assert_allclose(hostmath.cellwise_xcorr(X, Y, cells, 4),
                               ref.gen_covcorr(spec, X, Y, spec.dummy), atol=1e-13)


def test_shard_bounds():
    from pypyls_amd.parallel import shard_bounds
    for n in (0, 1, 7, 8, 100, 10001):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_shard_chunks_partition():
    """Chunk-cyclic shards: every row owned once, chunks ascending, world 1 = one range,
    counts within parts of each other."""
    from pypyls_amd import parallel
    assert parallel.shard_chunks(17, 0, 1) == [(0, 17)]
    for n, w, parts in [(n, w, p) for n in (0, 1, 7, 64, 1000, 10001) for w in (2, 3, 8) for p in (1, 2, 4)]:
        if True:
            seen = np.concatenate([parallel.shard_rows(n, r, w, parts) for r in range(w)])
            assert sorted(seen.tolist()) == list(range(n))
            counts = [len(parallel.shard_rows(n, r, w, parts)) for r in range(w)]
            assert max(counts) - min(counts) <= parts
            for r in range(w):
                ch = parallel.shard_chunks(n, r, w, parts)
                assert all(ch[i][1] <= ch[i + 1][0] for i in range(len(ch) - 1))
                if parts == 1:
                    assert ch == ([parallel.shard_bounds(n, r, w)] if n > r else []) or n < w
            if n >= w * parts:
                # every rank has work within the first 1 / parts of the rows (+ a chunk)
                first = [parallel.shard_chunks(n, r, w, parts)[0][0] for r in range(w)]
                assert max(first) <= n // parts


_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r})
from pypyls_amd import parallel
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
rank, world = parallel.rank_world()
L, Tp, B, P, R = 3, 4, 11, 7, 5
rs = np.random.RandomState(0)
perm_all = rs.rand(L, P); dist_all = rs.rand(Tp, L, R)
usum_parts = rs.rand(world, B, L); usq_parts = rs.rand(world, B, L)
lo, hi = parallel.shard_bounds(P, rank, world)
bl, bh = parallel.shard_bounds(R, rank, world)
perm, dd, usum, usq = parallel.collect(perm_all[:, lo:hi], P, dist_all[:, :, bl:bh], R,
                                       torch.from_numpy(usum_parts[rank]), torch.from_numpy(usq_parts[rank]))
assert np.array_equal(perm, perm_all) and np.array_equal(dd, dist_all)
want = usum_parts[0].copy()
for r in range(1, world): want += usum_parts[r]
assert np.array_equal(usum.numpy(), want)
# permutation-only call
perm2, d2, u2, q2 = parallel.collect(perm_all[:, lo:hi], P, None, 0, None, None)
assert np.array_equal(perm2, perm_all) and d2 is None and u2 is None
# chunk-cyclic shards (the lever behind the bootstraps of the front-ends): rows come back in global order
parallel.BOOT_PARTS = 2
for n in (5, 29, 64):
    rows_all = rs.rand(n, 2, 3)
    mine = parallel.shard_rows(n, rank, world)
    full, _ = parallel.collect_slices([torch.from_numpy(perm_all.T[lo:hi].copy()), torch.from_numpy(rows_all[mine])],
                                      [P, n], [], cyclic=[1])
    assert np.array_equal(full[0], perm_all.T) and np.array_equal(full[1], rows_all), n
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


@pytest.mark.parametrize('world', [2, 3])
def test_collect_multiprocess_gloo(tmp_path, world):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER.format(root=ROOT))
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(port), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count('ok') == world


def test_check_index_array():
    """User-supplied resampling arrays are validated like numpy fancy indexing
    (pyls/base.py:569,599): integer dtype, negatives wrap, out of range raises."""
    from pypyls_amd.engine import check_index_array
    a = np.array([[0, 4], [-1, 2], [3, -5]])
    np.testing.assert_array_equal(check_index_array(a, 5), [[0, 4], [4, 2], [3, 0]])
    for bad in (np.array([[0, 5]]), np.array([[-6, 1]]), np.array([[2 ** 40, 0]])):
        with pytest.raises(IndexError):
            check_index_array(bad, 5)
    with pytest.raises(IndexError):
        check_index_array(np.array([[0.0, 1.0]]), 5)
    with pytest.raises(IndexError):
        check_index_array(np.array([[True, False]]), 5)
    assert check_index_array(np.zeros((5, 0), int), 5).shape == (5, 0)


def test_regression_3d_y_without_bootstraps_reaches_the_engine():
    """3-D Y with n_boot=0 died on an unbound local before any work."""
    import torch
    import pypyls_amd as pls
    from pypyls_amd import engine
    if torch.cuda.is_available():
        pytest.skip('GPU present: covered by the gpu tests')
    rs = np.random.RandomState(0)
    with pytest.raises(engine.PlsxError):
        pls.pls_regression(rs.rand(12, 6), rs.rand(12, 3, 4), n_components=2, n_perm=0, n_boot=0)


_SEED_WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, {root!r})
from pypyls_amd import parallel, resampling
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
rank = dist.get_rank()
np.random.seed(1000 + rank)                      # ranks start from DIFFERENT global states
assert parallel.shared_seed(77) == 77
for seed in (None, np.random.RandomState(5 + rank)):
    s = parallel.shared_seed(seed)
    perms = resampling.gen_permsamp([6, 7], 2, 9, seed=s, verbose=False)
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, (s, perms.tobytes()))
    assert all(b == box[0] for b in box), 'ranks drew different index arrays'
dist.destroy_process_group()
print('rank', rank, 'ok')
'''


def test_shared_seed_multiprocess_gloo(tmp_path):
    """with seed=None every rank must still draw the same arrays."""
    script = tmp_path / 'seed_worker.py'
    script.write_text(_SEED_WORKER.format(root=ROOT))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), str(script)]
    out = subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR='127.0.0.1'), capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count('ok') == 2


def test_every_c_abi_entry_is_guarded_and_reads_no_environment():
    """no C++ exception crosses the ABI, and the route switches are an interface
    (plsx_set_option), not ambient environment.  Every ``extern "C"`` definition in csrc/*.hip is either
    a function-try-block closed by PLSX_CATCH or a one-line accessor that cannot throw; nothing under
    csrc/ calls getenv."""
    import re
    csrc = os.path.join(ROOT, 'pypyls_amd', 'csrc')
    lines = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith('.hip'):
            lines += open(os.path.join(csrc, f)).read().split('\n') + ['']
    header = open(os.path.join(ROOT, 'include', 'plsx.h')).read()
    declared = set(re.findall(r'\b(plsx_\w+)\s*\(', header))
    trivial = {'plsx_version', 'plsx_max_tprime', 'plsx_last_error', 'plsx_num_lv', 'plsx_tprime',
               'plsx_kernel_class_name', 'plsx_option_name', 'plsx_boot_route', 'plsx_split_route', 'plsx_comm_rank',
               'plsx_comm_transport'}
    defined, unguarded = set(), []
    for i, ln in enumerate(lines):
        m = re.match(r'^(?:int|const char\*) (plsx_\w+)\(', ln)
        if not m or ln.rstrip().endswith(';'):
            continue
        name = m.group(1)
        j = i
        while not (lines[j].startswith('{') or lines[j].startswith('try {') or lines[j].rstrip().endswith('}')
                   or lines[j].rstrip().endswith(';')):
            j += 1
        if lines[j].rstrip().endswith(';') and '{' not in lines[j]:
            continue                                   # forward declaration
        defined.add(name)
        if name in trivial:
            continue
        if not lines[j].startswith('try {'):
            unguarded.append(name)
            continue
        k = j
        while not lines[k].startswith('}'):
            k += 1
        assert lines[k].startswith('} PLSX_CATCH('), (name, lines[k])
    assert not unguarded, unguarded
    assert declared <= defined, declared - defined
    for f in os.listdir(csrc):
        if os.path.isfile(os.path.join(csrc, f)):
            assert 'getenv' not in open(os.path.join(csrc, f)).read(), f
    # ... and the product's Python does not select routes from the environment either
    for f in ('engine.py', 'plsc.py', 'regression.py', 'parallel.py', 'resampling.py'):
        text = open(os.path.join(ROOT, 'pypyls_amd', f)).read()
        uses = re.findall(r"environ(?:\.get)?\(?\[?['\"]PLSX_", text)
        assert not uses or f == 'engine.py', (f, uses)     # engine.options_from_env: bench / tests helper only


def test_options_from_env_translates_known_keys_only():
    from pypyls_amd import engine
    kw = engine.options_from_env({'PLSX_NO_DUAL_PERM': '1', 'PLSX_MIN_BATCH': '512', 'PLSX_SCRATCH_GB': '6',
                                  'PLSX_UNKNOWN': '1', 'HOME': '/root'})
    assert kw == {'options': {'no_dual_perm': 1, 'min_batch': 512}, 'scratch_gb': 6.0}
    assert engine.options_from_env({}) == {'options': {}}


def test_collect_device_emulated_world_orders_and_sums_like_a_real_one():
    """parallel.collect_device with emulate=(rank, world) (bench.py --mode analysis --emulate-world): no peers,
    the surrogate gather replicates this rank's packed buffer; the bookkeeping -- padded shard sizes, the global
    order of chunk-cyclic shards, the rank-ordered sums -- is the code path of a real gather."""
    import torch
    from pypyls_amd import parallel
    n_perm, n_boot, world = 11, 13, 4
    for rank in range(world):
        plo, phi = parallel.shard_bounds(n_perm, rank, world)
        rows = parallel.shard_rows(n_boot, rank, world)
        perm = torch.arange(plo, phi, dtype=torch.float64)[:, None] * torch.ones(1, 3, dtype=torch.float64)
        dist = torch.from_numpy(rows.astype(np.float64))[:, None, None] * torch.ones(1, 2, 2, dtype=torch.float64)
        usum = torch.full((5, 2), float(rank + 1), dtype=torch.float64)
        full, summed = parallel.collect_device([perm, dist], [n_perm, n_boot], [usum], cyclic=[1],
                                               emulate=(rank, world))
        assert full[0].shape == (n_perm, 3) and full[1].shape == (n_boot, 2, 2)
        # this rank's own rows sit where the global order puts them (the other ranks' slots hold copies)
        np.testing.assert_array_equal(full[0][plo:phi, 0].numpy(), np.arange(plo, phi))
        np.testing.assert_array_equal(full[1][rows, 0, 0].numpy(), rows)
        np.testing.assert_array_equal(summed[0].numpy(), np.full((5, 2), world * (rank + 1.0)))
    # without a process group and without emulation nothing moves
    t = torch.ones(3, 2, dtype=torch.float64)
    full, summed = parallel.collect_device([t], [3], [t])
    assert full[0] is t and summed[0] is t


def test_team_collect_bookkeeping_on_cpu_threads():
    """parallel.collect_device(team=(rank, team)) -- the collective of ONE process driving several device contexts
    (pypyls_amd/team.py) -- with the device transport replaced by a host barrier: three rank threads with uneven
    contiguous and chunk-cyclic shards get the full arrays in global order and the rank-ordered sums, like the gloo
    ranks of test_collect_multiprocess_gloo do."""
    import threading
    import torch
    from pypyls_amd import parallel

    class Eng(object):
        device = torch.device('cpu')

    class HostTeam(object):
        def __init__(self, world):
            self.world, self.engines = world, [Eng() for _ in range(world)]
            self.barrier, self.slots = threading.Barrier(world), [None] * world

        def allgather(self, rank, flat):
            self.slots[rank] = flat
            self.barrier.wait()
            out = torch.stack(list(self.slots))
            self.barrier.wait()
            return out

    world, n_perm, n_boot = 3, 10, 11
    perm = torch.arange(n_perm * 4, dtype=torch.float64).reshape(n_perm, 4)
    dist = torch.arange(n_boot * 6, dtype=torch.float64).reshape(n_boot, 2, 3) * 0.5
    team = HostTeam(world)
    got = [None] * world

    def work(r):
        lo, hi = parallel.shard_bounds(n_perm, r, world)
        rows = parallel.shard_rows(n_boot, r, world)
        sums = [torch.full((5, 2), float(r + 1), dtype=torch.float64)]
        got[r] = parallel.collect_device([perm[lo:hi], dist[rows]], [n_perm, n_boot], sums, cyclic=[1], team=(r, team))
    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for r in range(world):
        full, summed = got[r]
        assert torch.equal(full[0], perm) and torch.equal(full[1], dist)
        assert torch.equal(summed[0], torch.full((5, 2), 6.0, dtype=torch.float64))


def test_resolve_devices_maps_n_proc_to_gpus(monkeypatch):
    """n_proc (the reference's worker count, pyls/structures.py:162-168) -> device ordinals of a team."""
    import torch
    from pypyls_amd import team
    from pypyls_amd.structures import PLSInputs
    assert team.resolve_devices(4) is None or torch.cuda.is_available()      # no GPU here: the ordinary call
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    monkeypatch.setattr(torch.cuda, 'current_device', lambda: 2)
    assert team.resolve_devices(None) is None and team.resolve_devices(1) is None
    assert team.resolve_devices(3) == [2, 3, 4]
    assert team.resolve_devices(PLSInputs(n_proc='max').n_proc) == [2, 3, 4, 5, 6, 7, 0, 1]
    assert team.resolve_devices(PLSInputs(n_proc=-1).n_proc) == [2, 3, 4, 5, 6, 7, 0, 1]
    assert team.resolve_devices(2, device_ids=[5, 6, 7]) == [5, 6, 7]        # explicit ids win
    assert team.resolve_devices(None, device_ids=[2]) is None                  # the current device: the ordinary call
    assert team.resolve_devices(None, device_ids=[4]) == [4]
    assert team.resolve_devices(None, device_ids=[0, 0]) == [0, 0]
    import pytest
    with pytest.raises(ValueError):
        team.resolve_devices(None, device_ids=[8])
