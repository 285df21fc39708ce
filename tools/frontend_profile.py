import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import pypyls_amd as pls
from pypyls_amd import plsc, engine as E
rs = np.random.RandomState(0)
S,B,T = 500,200000,50
X = rs.randn(S,B); Y = rs.randn(S,T) + 0.3*X[:,:T]
pls.behavioral_pls(X, Y, n_perm=64, n_boot=64, test_split=0, seed=1, verbose=False)
# monkeypatch timers
import functools
acc = {}
def timed(cls, name):
    f = getattr(cls, name)
    @functools.wraps(f)
    def w(*a, **k):
        torch.cuda.synchronize(); t=time.perf_counter(); r=f(*a, **k); torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter()-t; return r
    setattr(cls, name, w)
for n in ['set_data','colmean','decompose','set_original','project','boot_rel','percentile_ci','perm_into','boot_into','rows_tensor']:
    timed(E.Engine, n)
from pypyls_amd import parallel
f0 = parallel.collect_slices
def cs(*a, **k):
    torch.cuda.synchronize(); t=time.perf_counter(); r=f0(*a, **k); acc['collect_slices']=time.perf_counter()-t; return r
parallel.collect_slices = cs
torch.cuda.synchronize(); t=time.perf_counter()
res = pls.behavioral_pls(X, Y, n_perm=5000, n_boot=5000, test_split=0, seed=1234, verbose=False)
torch.cuda.synchronize(); tot=time.perf_counter()-t
print('total %.3f'%tot, {k: round(v,3) for k,v in acc.items()}, 'sum %.3f'%sum(acc.values()))
