"""Two ranks (gloo) sharing the one visible GPU: exercises the resample
sharding + single-collective path of the front-end end to end and checks that
the sharded result equals the single-rank result."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, {root!r})
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
import pypyls_amd as pls
rs = np.random.RandomState(0)
X = rs.randn(40, 300); Y = rs.randn(40, 5) + 0.4 * X[:, :5]
res = pls.behavioral_pls(X, Y, n_perm=11, n_boot=9, n_split=3, test_split=0, seed=7, verbose=False)
Xm = rs.randn(36, 200); Xm[:12] += 1.0
rm = pls.meancentered_pls(Xm, groups=[6, 6, 6], n_cond=2, n_perm=7, n_boot=7, seed=3, verbose=False)
rr = pls.pls_regression(X, Y, n_components=3, n_perm=6, n_boot=5, seed=5, verbose=False)
# long series: every rank announces its share (plsx_boot_begin) and takes the quadratic-form route of the sums
rm2 = pls.meancentered_pls(Xm, groups=[6, 6, 6], n_cond=2, n_perm=4, n_boot=700, seed=3, verbose=False)
rr2 = pls.pls_regression(X, Y, n_components=3, n_perm=4, n_boot=400, seed=5, verbose=False)
if dist.get_rank() == 0:
    np.savez({out!r}, m2bsr=rm2.bootres.x_weights_normed, m2se=rm2.bootres.x_weights_stderr,
             r2bsr=rr2.bootres.x_weights_normed, r2se=rr2.bootres.x_weights_stderr,
             perm=res.permres.perm_singval, bsr=res.bootres.x_weights_normed,
             ylb=res.bootres.y_loadings_boot, uc=res.splitres.ucorr_pvals, ul=res.splitres.ucorr_uplim,
             mperm=rm.permres.perm_singval, mbsr=rm.bootres.x_weights_normed,
             rperm=rr.permres.perm_singval, rbsr=rr.bootres.x_weights_normed)
dist.barrier()
dist.destroy_process_group()
'''



def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port

def test_two_ranks_equal_one_rank(tmp_path):
    out = str(tmp_path / 'two.npz')
    script = tmp_path / 'w.py'
    script.write_text(_WORKER.format(root=ROOT, out=out))
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    two = np.load(out)
    import pypyls_amd as pls
    rs = np.random.RandomState(0)
    X = rs.randn(40, 300)
    Y = rs.randn(40, 5) + 0.4 * X[:, :5]
    res = pls.behavioral_pls(X, Y, n_perm=11, n_boot=9, n_split=3, test_split=0, seed=7, verbose=False)
    Xm = rs.randn(36, 200)
    Xm[:12] += 1.0
    rm = pls.meancentered_pls(Xm, groups=[6, 6, 6], n_cond=2, n_perm=7, n_boot=7, seed=3, verbose=False)
    rr = pls.pls_regression(X, Y, n_components=3, n_perm=6, n_boot=5, seed=5, verbose=False)
    np.testing.assert_allclose(two['perm'], res.permres.perm_singval, rtol=1e-12)
    np.testing.assert_allclose(two['ylb'], res.bootres.y_loadings_boot, rtol=1e-12)
    np.testing.assert_allclose(two['bsr'], res.bootres.x_weights_normed, rtol=1e-9)
    np.testing.assert_allclose(two['uc'], res.splitres.ucorr_pvals, rtol=1e-12)
    np.testing.assert_allclose(two['ul'], res.splitres.ucorr_uplim, rtol=1e-10)
    np.testing.assert_allclose(two['mperm'], rm.permres.perm_singval, rtol=1e-12)
    np.testing.assert_allclose(two['mbsr'], rm.bootres.x_weights_normed, rtol=1e-9)
    np.testing.assert_allclose(two['rperm'], rr.permres.perm_singval, rtol=1e-12)
    np.testing.assert_allclose(two['rbsr'], rr.bootres.x_weights_normed, rtol=1e-9)
    # series long enough for the quadratic-form route of the bootstrap sums, on one rank and on each of two
    from pypyls_amd import engine
    rm2 = pls.meancentered_pls(Xm, groups=[6, 6, 6], n_cond=2, n_perm=4, n_boot=700, seed=3, verbose=False)
    rr2 = pls.pls_regression(X, Y, n_components=3, n_perm=4, n_boot=400, seed=5, verbose=False)
    live = rm2.singvals > 1e-8 * rm2.singvals.max()
    np.testing.assert_allclose(two['m2bsr'][:, live], rm2.bootres.x_weights_normed[:, live], rtol=1e-7)
    np.testing.assert_allclose(two['m2se'][:, live], rm2.bootres.x_weights_stderr[:, live], rtol=1e-7)
    np.testing.assert_allclose(two['r2bsr'], rr2.bootres.x_weights_normed, rtol=1e-7)
    np.testing.assert_allclose(two['r2se'], rr2.bootres.x_weights_stderr, rtol=1e-7)
    eng = engine.default_engine()
    try:                                            # ... and the per-bootstrap pass agrees with both
        eng.set_option('quad_sums', -1)
        rm3 = pls.meancentered_pls(Xm, groups=[6, 6, 6], n_cond=2, n_perm=4, n_boot=700, seed=3, verbose=False)
    finally:
        eng.set_option('quad_sums', 0)
    np.testing.assert_allclose(rm3.bootres.x_weights_stderr[:, live], rm2.bootres.x_weights_stderr[:, live], rtol=1e-7)
    assert np.max(np.abs(rm3.bootres.x_weights_stderr - rm2.bootres.x_weights_stderr)[:, live]) > 0


_NCCL_WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, {root!r})
world = int(os.environ['WORLD_SIZE']); rank = int(os.environ['RANK'])
torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
dist.init_process_group('nccl', rank=rank, world_size=world,
                        device_id=torch.device('cuda', torch.cuda.current_device()))
assert dist.get_backend() == 'nccl'
import pypyls_amd as pls
from pypyls_amd import parallel
# a pure query: nothing is opened behind the caller's back (ADVICE r5) ...
assert parallel.collective_name().startswith('nccl all_gather_into_tensor')
assert parallel.native_comm() is None
# ... the data collective becomes the C ABI's plsx_allgather once every rank asks for it, explicitly
assert parallel.open_native_comm() is not None, parallel._NATIVE['why']
name = parallel.collective_name()
assert name.startswith('plsx_allgather'), (name, parallel._NATIVE['why'])
assert parallel.native_comm().comm_rank_world() == (rank, world)
rs = np.random.RandomState(0)
X = rs.randn(40, 300); Y = rs.randn(40, 5) + 0.4 * X[:, :5]
res = pls.behavioral_pls(X, Y, n_perm=11, n_boot=9, n_split=3, test_split=4, seed=7, verbose=False)
rr = pls.pls_regression(X, Y, n_components=3, n_perm=6, n_boot=5, seed=5, verbose=False)
# ... and the process group's own all-gather (the fall-back) returns the same bits
parallel.release_native_comm()
assert parallel.collective_name().startswith('nccl all_gather_into_tensor')
res2 = pls.behavioral_pls(X, Y, n_perm=11, n_boot=9, n_split=3, test_split=4, seed=7, verbose=False)
assert np.array_equal(res2.permres.perm_singval, res.permres.perm_singval)
assert np.array_equal(res2.bootres.x_weights_normed, res.bootres.x_weights_normed)
if rank == 0:
    np.savez({out!r}, perm=res.permres.perm_singval, bsr=res.bootres.x_weights_normed,
             ylb=res.bootres.y_loadings_boot, uc=res.splitres.ucorr_pvals, cv=res.cvres.pearson_r,
             rperm=rr.permres.perm_singval, rbsr=rr.bootres.x_weights_normed)
dist.barrier()
dist.destroy_process_group()
'''


def test_exported_allgather_world_of_one():
    """plsx_allgather / plsx_comm_* through ctypes (include/plsx.h, "The collective"): without a communicator a
    context is a world of one (device copy); with one opened by plsx_comm_init(id, 0, 1) the gather runs through
    RCCL's ncclAllGather, in place and out of place, fp64 and odd byte counts; a second init is refused."""
    import torch
    from pypyls_amd import engine
    eng = engine.Engine()
    try:
        a = torch.arange(1000, dtype=torch.float64, device=eng.device) * 0.5
        out = torch.zeros((1, 1000), dtype=torch.float64, device=eng.device)
        assert eng.comm_rank_world() == (0, 1)
        eng.allgather_into(a, out)
        torch.cuda.synchronize()
        assert torch.equal(out[0], a)
        eng.comm_load()
        uid = eng.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        eng.comm_init(uid, 0, 1)
        assert eng.comm_rank_world() == (0, 1)
        out.zero_()
        eng.allgather_into(a, out)
        torch.cuda.synchronize()
        assert torch.equal(out[0], a)
        eng.allgather_into(out[0], out)                 # in place
        b = torch.arange(13, dtype=torch.uint8, device=eng.device)
        ob = torch.zeros((1, 13), dtype=torch.uint8, device=eng.device)
        eng.allgather_into(b, ob)                       # 13 bytes: travels as bytes
        torch.cuda.synchronize()
        assert torch.equal(out[0], a) and torch.equal(ob[0], b)
        with pytest.raises(engine.PlsxError):
            eng.comm_init(uid, 0, 1)
        with pytest.raises(engine.PlsxError):
            eng.allgather_into(a, torch.zeros((2, 1000), dtype=torch.float64, device=eng.device))
        eng.comm_destroy()
        assert eng.comm_rank_world() == (0, 1)
    finally:
        eng.close()



def _run_nccl(tmp_path, world, port):
    out = str(tmp_path / 'nccl{}.npz'.format(world))
    script = tmp_path / 'wn{}.py'.format(world)
    script.write_text(_NCCL_WORKER.format(root=ROOT, out=out))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    got = np.load(out)
    import pypyls_amd as pls
    rs = np.random.RandomState(0)
    X = rs.randn(40, 300)
    Y = rs.randn(40, 5) + 0.4 * X[:, :5]
    res = pls.behavioral_pls(X, Y, n_perm=11, n_boot=9, n_split=3, test_split=4, seed=7, verbose=False)
    rr = pls.pls_regression(X, Y, n_components=3, n_perm=6, n_boot=5, seed=5, verbose=False)
    np.testing.assert_allclose(got['perm'], res.permres.perm_singval, rtol=1e-12)
    np.testing.assert_allclose(got['ylb'], res.bootres.y_loadings_boot, rtol=1e-12)
    np.testing.assert_allclose(got['bsr'], res.bootres.x_weights_normed, rtol=1e-9)
    np.testing.assert_allclose(got['uc'], res.splitres.ucorr_pvals, rtol=1e-12)
    np.testing.assert_allclose(got['cv'], res.cvres.pearson_r, rtol=1e-10)
    np.testing.assert_allclose(got['rperm'], rr.permres.perm_singval, rtol=1e-12)
    np.testing.assert_allclose(got['rbsr'], rr.bootres.x_weights_normed, rtol=1e-9)


def test_collective_on_rccl_one_rank(tmp_path):
    """The single all-gather of the front-ends on the RCCL backend (GPU-side pack /
    unpack, parallel.gather_device) with a 1-rank group -- what a 1-GPU box can run."""
    _run_nccl(tmp_path, 1, _free_port())


def test_collective_on_rccl_two_ranks(tmp_path):
    """Two ranks, one GPU each, RCCL: sharded result == single-process result."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs (RCCL refuses two ranks on one device)')
    _run_nccl(tmp_path, 2, _free_port())


def test_bench_launches_its_own_ranks(tmp_path):
    """`bench.py --gpus 2` outside a launcher spawns two ranks itself and reports the
    group's world size.  Dry run on one GPU: the ranks share it over gloo
    (PLSX_BENCH_SHARE_GPU; RCCL refuses two ranks on one device)."""
    import json
    env = dict(os.environ, PLSX_BENCH_SHARE_GPU='1', PLSX_BENCH_BACKEND='gloo', PLSX_SCRATCH_GB='4')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1',
           '--B', '20000', '--perms', '56', '--boots', '56', '--cpu-sample', '0']
    small = ['--strong-resamples', '300', '--split-arrangements', '5', '--n-split', '12']
    proc = subprocess.run(cmd + small, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    line = [l for l in proc.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['scaling'] == 'weak' and out['value'] > 0 and out['value_dual'] > 0
    assert out['roofline']['bound'] in ('mfma', 'hbm') and out['config']['perms_per_step'] == 56
    # the driver's one command per N also carries ONE sharded analysis (strong) and ONE sharded split-half leg, each
    # with every rank's own times and the collective
    sh = out['sharded']
    assert 'error' not in sh['strong'] and 'error' not in sh['c4split'], sh
    assert sh['strong']['scaling'] == 'strong' and sh['strong']['n_gpus'] == 2 and sh['strong']['value'] > 0
    assert [p['rank'] for p in sh['strong']['per_rank_phases_ms']] == [0, 1]
    assert all('bootstraps' in p and 'collective' in p for p in sh['strong']['per_rank_phases_ms'])
    assert sh['strong']['collective_ms'] >= 0
    assert sh['c4split']['scaling'] == 'strong' and sh['c4split']['value'] > 0
    assert [p['rank'] for p in sh['c4split']['per_rank']] == [0, 1]
    assert sh['c4split']['collective_ms_per_step'] >= 0 and '5 arrangements' in sh['c4split']['workload']
    # strong mode: one analysis split over the two ranks, index generation inside the clock
    proc = subprocess.run(cmd + ['--mode', 'strong', '--perms', '120', '--boots', '112', '--no-primal'], env=env,
                          capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    out = json.loads([l for l in proc.stdout.splitlines() if l.startswith('{')][-1])
    assert out['n_gpus'] == 2 and out['scaling'] == 'strong'
    # the END-TO-END public call under two real ranks (the front-end shards and gathers over the process group)
    proc = subprocess.run(cmd[:3] + ['2', '--mode', 'analysis', '--config', 'c2', '--perms', '200', '--boots', '200',
                                     '--steps', '1'], env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    out = json.loads([l for l in proc.stdout.splitlines() if l.startswith('{')][-1])
    assert out['n_gpus'] == 2 and out['config']['mode'] == 'analysis' and out['value'] > 0
    assert 'over 2 ranks' in out['config']['collective']
    # more ranks than GPUs without the dry-run switch: refused loudly
    env2 = dict(env)
    env2.pop('PLSX_BENCH_SHARE_GPU')
    import torch
    if torch.cuda.device_count() < 2:
        proc = subprocess.run(cmd, env=env2, capture_output=True, text=True, timeout=300)
        assert proc.returncode != 0 and 'visible' in proc.stderr


@pytest.mark.parametrize('config', ['c2'])
def test_bench_config_lines(config):
    """`bench.py --config` lines carry the contract's keys (roofline, cpu_baseline, kernel split)."""
    import json
    env = dict(os.environ, PLSX_SCRATCH_GB='8')
    env.pop('WORLD_SIZE', None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', config, '--steps', '2',
                           '--warmup', '1'], env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                       # ONE JSON line, nothing after it
    out = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in out, key
    assert out['dtype'] == 'f64' and out['roofline']['bound'] in ('hbm', 'mfma')
    assert 0 < out['roofline']['frac'] <= 1.0 and out['cpu_baseline']['kind'] == 'port'
    # `value` is the north-star pipeline (every permutation passes over X), the S x S route rides beside it
    assert out['value_dual'] >= out['value'] > 0 and 'k_xprod' in out['config']['kernel_ms_per_step']
    assert 'feature pass' in out['config']['perm_path']


def test_bench_analysis_mode_line():
    """`bench.py --mode analysis`: the end-to-end line of the public call with per-phase times and the emulated
    end-to-end critical path (small config, two emulated worlds)."""
    import json
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--mode', 'analysis', '--config', 'c2',
                           '--perms', '300', '--boots', '300', '--steps', '1', '--emulate-world', '1,2'],
                          env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out['config']['mode'] == 'analysis' and out['value'] > 0
    w = out['end_to_end_emulation']['worlds']
    assert set(w) == {'1', '2'}
    for key in ('h2d_and_bind', 'decompose', 'permutations', 'bootstraps', 'collective', 'finish_and_d2h', 'host_finish'):
        assert key in w['1']['rank0_phases_ms'], key
    assert w['2']['critical_path_ms'] > 0 and 0 < w['2']['efficiency'] <= 1.5
    assert out['fixed_cost_ms']['everything_else_ms'] >= 0


def test_bench_default_line_embeds_every_config():
    """The driver's command (`bench.py --gpus 1 --steps K --warmup W`) ends in ONE line that carries the headline
    (value = feature-pass pipeline, value_dual beside it) AND a sub-record per other BASELINE config plus the
    end-to-end call, each with value / ms_per_step / roofline.frac / cpu_baseline."""
    import json
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1'],
                          env=env, capture_output=True, text=True, timeout=1500)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out['value_dual'] > out['value'] > 0 and out['hbm_algorithmic_TBps'] <= 8.0
    assert 0 < out['roofline']['frac'] <= 1.0 and out['cpu_baseline']['kind'] == 'port'
    cfgs = out['configs']
    for c in ('c2', 'c3', 'c4split', 'c5'):
        assert 'error' not in cfgs[c], cfgs[c]
        assert cfgs[c]['value'] > 0 and cfgs[c]['ms_per_step'] > 0 and cfgs[c]['cpu_baseline']['value'] > 0
        assert 0 < cfgs[c]['roofline']['frac'] <= 1.0, (c, cfgs[c]['roofline'])
    assert 'unpinned' in cfgs['c5']['parity_note']
    assert 'error' not in cfgs['c4_analysis'] and cfgs['c4_analysis']['value'] > 0
    assert cfgs['wall_s'] < 300
